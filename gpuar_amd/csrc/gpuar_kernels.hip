// gpuar_kernels.hip -- gfx950 (MI355X, CDNA4) kernels for the GPUAR packet codec.
//
// One lane = one packet, 64 packets per wavefront.  What the reference does
// with one CUDA thread per packet in 32-thread blocks, a 516-byte Fenwick tree
// per thread in shared memory and bit-at-a-time loops
// (/root/reference/src/gpuar_kernel.cu:894-934, 205-238, 321-367, 787-836) is
// re-derived here for 64-wide wavefronts (per-lane code: lane_codec.h):
//
//  * encode_kernel (throughput): four wavefronts per 64 packets, one role per SIMD -- a TOP MODELER (reads the input,
//    walks depths 1-4 of the 64 adaptive models: per lane a binary left-count tree in LDS, node-major / lane-minor so
//    that lane l always hits bank l & 31; software-pipelined walks that yield cumLo, cumHi and the count update), a
//    LOW MODELER (depth 0 in a register, depths 5-7, the x == 255 term, added onto the top modeler's part in place), a
//    CODER (interval narrowing by a wave-uniform reciprocal, one-clz renormalisation count, the lower bound as a 64-bit
//    window with carries, predicated dword stores) and a COURIER that carries the coder's reciprocals from memory into
//    LDS a phase ahead, so that no working role issues a scalar load.  The roles work one phase (8 symbols) apart and
//    hand a phase on in place through a three-slot LDS ring, one s_barrier per phase;
//  * encode_small_kernel (latency): the same integers cut finer for inputs that cannot fill the chip -- four tree
//    roles, an interval role, a sink role and a courier, seven wavefronts per 64 packets, phases of 16 symbols;
//  * decode_*_kernel: one wavefront per 64 packets; the symbol search reads two 16-byte subtree records per symbol
//    instead of walking eight levels, applies the increments of the 8-byte half of each that the path went through by
//    one 64-bit LDS add, and works on a scaled remainder (no division, borrow = path bit); the symbol step is a
//    hand-scheduled instruction stream; the packet stream reaches it through a per-lane ring in LDS, the per-symbol
//    reciprocal multipliers as VECTOR operands, eight at a time by loads with a wave-uniform address (no scalar load,
//    no v_readlane in the loop);
//  * compaction (scan + gather), synthetic-stream generators, a plain copy (the measured HBM roof).
//
// Bit-exact with the reference: same counts, same integer arithmetic, same
// bitstream.  No MFMA: this is integer, bit-serial
// work bounded by VALU issue and LDS operations, not by HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gpuar_hip.h"

#include "lane_codec.h"

// The GPUAR_EXP_* switches below take pieces OUT of the kernels to price them (profiles/r0N_*budget*.txt, *attribution*.txt): such a
// build produces WRONG output.  It only compiles when the build says it is one (tools/exp_build.sh defines GPUAR_EXPERIMENT_BUILD for
// any flag of that family), and the library then says so in gpuar_hip_version(), which gpuar_amd/hip.py refuses to load as the product.
#if (defined(GPUAR_EXP_NO_ADDS) || defined(GPUAR_EXP_NO_CODER) || defined(GPUAR_EXP_NO_LOW) || defined(GPUAR_EXP_NO_READS) || defined(GPUAR_EXP_NO_RING) || defined(GPUAR_EXP_NO_RING_MUL) || defined(GPUAR_EXP_NO_RING_VMCNT) || defined(GPUAR_EXP_NO_RING_WRITES) || defined(GPUAR_EXP_NO_SEARCH) || defined(GPUAR_EXP_NO_STORES) || defined(GPUAR_EXP_NO_STREAM) || defined(GPUAR_EXP_NO_STREAM_READ) || defined(GPUAR_EXP_NO_WAIT1) || defined(GPUAR_EXP_NO_WAIT2)) && !defined(GPUAR_EXPERIMENT_BUILD)
#error "a GPUAR_EXP_* switch without GPUAR_EXPERIMENT_BUILD: these builds decode / encode garbage; use tools/exp_build.sh"
#endif
namespace gpuar {

__constant__ RecipTable g_recip = RecipTable();
// The reciprocal multipliers alone, for the decoder (which fetches them eight at a time, two runs ahead, by vector loads
// with a wave-uniform address); padded so that the look-ahead behind a packet's last symbols stays inside the table.
struct MulTable {
    uint32_t m[kPacket + 64];
    constexpr MulTable() : m{} {
        const RecipTable r = RecipTable();
        for (uint32_t i = 0; i < kPacket + 64; ++i) m[i] = r.r[i < kPacket ? i : kPacket - 1u].mul;
    }
};
__device__ const MulTable g_mul = MulTable();
__constant__ DecodeConstTable g_decode = DecodeConstTable();
__device__ uint32_t g_status = 0;
__device__ uint32_t g_cu_ticket[2048];      // one arrival counter per CU (XCC, SE, SH, CU), see encode_kernel
// What the shader clock really was while the two throughput kernels ran (measurement support, gpuar_hip_clock_samples): every
// 64th workgroup notes how many shader-clock ticks (s_memtime) and how many ticks of the constant 100 MHz clock
// (s_memrealtime) passed between its start and its end, into a slot of its own -- no atomics, two scalar reads at either
// end of one wavefront in 64, nothing inside a symbol loop.  Sum of the first / sum of the second x 100 MHz is the clock the
// vector pipes ran at, which under this load is NOT the 2.4 GHz of the data sheet (bench.py: roofline_valu).
constexpr uint32_t kClockSlots = 256, kClockEvery = 64;
__device__ unsigned long long g_clock_samples[2][kClockSlots][4];      // [encode | decode][slot][shader, 100 MHz at the start; the same at the end]
// (both readings go straight to memory: nothing of this is alive across a symbol loop, so the loops' registers are what they
// were without it -- tests/test_codeobj_contract.py holds the decoder's step to its instruction budget)
__device__ __forceinline__ void clock_sample(uint32_t which, size_t group, uint32_t lane, uint32_t end) {      // group: wave-uniform
    // only the first kClockSlots x kClockEvery groups of a launch (8 GiB) sample: beyond them slots would be shared, and a slot
    // holding the start of one workgroup and the end of another -- possibly on XCDs whose counters differ -- is no measurement
    if ((group & (kClockEvery - 1u)) == 0u && group < size_t(kClockSlots) * kClockEvery && lane == 0u) {
        unsigned long long *slot = g_clock_samples[which][group / kClockEvery] + 2u * end;
        slot[0] = clock64();
        slot[1] = wall_clock64();
    }
}

// Lane's column in a tree row: lanes l and l+32 share a dword (low/high half),
// so the 32 lanes of each LDS lane-group hit 32 distinct banks whatever node
// each of them addresses.
__device__ __forceinline__ uint32_t lane_column(uint32_t lane) {
    return ((lane & 31u) << 1) | (lane >> 5);
}

__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t other = __shfl_xor(v, off);
        v = other > v ? other : v;
    }
    return __builtin_amdgcn_readfirstlane(v);
}

// ---------------------------------------------------------------------------
// Encode: replaces garCompress + arCompress (src/gpuar_kernel.cu:894-914, 487-531)
//
// Workgroup = 4 wavefronts, three of them working on the same 64 packets (lane l
// <-> packet 64*group + l in all three); which wavefront plays which role is
// decided per SIMD at run time (see encode_kernel):
//   TOP MODELER: reads the input bytes from memory (whole 128-byte lines), walks
//           depths 1..4 of the 64 adaptive models (LDS), emits its part of
//           cumLo | cumHi << 16 per symbol and hands the bytes on;
//   LOW MODELER: one phase behind: depth 0 (a register), depths 5..7 (LDS) and the
//           x == 255 term, added onto the top modeler's part in place (why 4 + 3:
//           lane_codec.h at TopModeler);
//   CODER: two phases behind: owns the interval state and the bit sink, turns the
//           sums into the packet bitstream;
//   the fourth wavefront carries the coder's reciprocals from the table in memory into LDS, a phase ahead, and meets
//           the barriers.
// They meet in a three-slot LDS ring of kPhase symbols per slot (EncodeLds), one
// s_barrier per phase.
//
// Why three: a packet's model pins 510 B of LDS, so a CU holds only 4 x 64
// packets however the work is arranged, and a lone wavefront issues at most
// one instruction per ~4.6 cycles (VALU) or ~10-12 cycles (LDS) -- measured,
// tools/valu_probe.hip, tools/lds_probe.hip.  One wavefront doing everything
// ran at 108 GB/s; modeler + coder at 290 GB/s with the modeler's serial
// stream (45 VALU + 13 LDS per symbol) as the bottleneck; cutting that stream
// in two puts three wavefronts on every SIMD for the same LDS.  32 KiB tree +
// 7 KiB of rings = 39 KiB per workgroup -> exactly 4 workgroups = 12 working wavefronts/CU
// (plus the 4 that only carry reciprocals and meet the barriers, see encode_kernel).
// ---------------------------------------------------------------------------
constexpr uint32_t kPhase = 8;
// Issue priority of the three roles (s_setprio; a SIMD hosts one wavefront of each role, from different
// groups, and a group moves at the pace of its slowest role between two barriers).  Measured on uniform 2 GiB
// (tools/kind_timing.py): all equal 6.14 ms; coder first 6.25; top modeler first 5.90; top > coder > low 5.70;
// top > low > coder 5.87; with the 4-level share of the tree moved to the low modeler the same numbers with the
// roles swapped -- whoever walks four LDS levels has to go first, the three-level modeler last.
// Round 4 (the coder in carry form, ten vector instructions shorter): top > low > coder.  With the old order the lighter coder
// bought nothing (5.14 ms for 6 + 1 and 5 + 2 depths alike); with the coder last 4.96-4.99 ms for 5 + 2, 4 + 3 and 3 + 4.
#ifndef GPUAR_PRIO_TOP
#define GPUAR_PRIO_TOP 3
#define GPUAR_PRIO_CODER 0
#define GPUAR_PRIO_LOW 2
#endif
constexpr int kPrioTop = GPUAR_PRIO_TOP, kPrioCoder = GPUAR_PRIO_CODER, kPrioLow = GPUAR_PRIO_LOW;

// The three roles work one phase apart -- the top modeler on the symbols of phase p, the low modeler on those of
// p - 1, the coder on those of p - 2 -- and hand a phase on IN PLACE: the top modeler writes its part of
// cumLo | cumHi << 16 per symbol, the low modeler adds its own, the coder reads the sum.  Three phases are alive at
// a time, so the ring has three slots.  The low modeler takes the symbols from the top one as well (the two input dwords
// of a phase, one LDS instruction on either side; it forms its row tags itself, one SDWA shift per symbol -- a u16 tag per
// symbol was eight LDS writes per phase in the top modeler's stream, the longest of the three, and eight reads in the low
// one's), so only one wavefront of a group reads the input from memory.
constexpr uint32_t kRingSlots = 3;
struct alignas(16) EncodeLds {
    uint8_t tree[kTreeRows * kLanes * 2];      // 32 KiB: 255 rows x (64 lanes x u16), in-order layout
    uint32_t sums[kRingSlots][kPhase][kLanes]; // 6 KiB: [slot][symbol][lane]
    uint32_t bytes[2][kPhase / 4][kLanes];     // 1 KiB: [phase parity][dword][lane], the input bytes of a phase's symbols as the
                                               // top modeler read them, for the low one: ONE LDS write per phase and lane
};
__device__ __forceinline__ uint32_t next_slot(uint32_t slot) { return slot == kRingSlots - 1u ? 0u : slot + 1u; }

// Which group of 64 packets a workgroup serves.  Workgroups are dealt to the eight XCDs round-robin
// (blockIdx & 7), each XCD with its own L2.  Serving groups in blockIdx order would give one XCD every
// eighth group, i.e. packets 512 apart -- and 512 slots (8704 B) or 512 packets (8192 B) apart is a
// multiple of the L2's set period, so everything an XCD has in flight would fall into a small fraction
// of its sets.  Instead each XCD walks its own contiguous eighth of the groups.  The grid is rounded up
// to a multiple of 8; a workgroup whose group does not exist returns at once.  (Measured on uniform
// 8 GiB: encoder L2 fetches 14.3 -> 9.6 GB, L2 write-backs 21.6 -> 12.4 GB, same run time.  The
// decoders, whose per-lane stores are whole 64-byte sectors, got slightly worse and keep blockIdx order.)
__device__ __forceinline__ size_t xcd_contiguous_group(uint32_t block, uint32_t grid) {
    const uint32_t per_xcd = grid >> 3;                       // grid is a multiple of 8
    return static_cast<size_t>(block & 7u) * per_xcd + (block >> 3);
}

// LDS only: the global loads/stores of every wave stay in flight across the barrier
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ uint4 load16_guarded(const uint8_t *p, size_t avail) {
    if (avail >= 16) return *reinterpret_cast<const uint4 *>(p);
    uint32_t w[4] = {0, 0, 0, 0};
    for (size_t b = 0; b < avail; ++b) w[b >> 2] |= static_cast<uint32_t>(p[b]) << (8u * (b & 3u));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// byte kByte of `word`, seven bits up: the row tag of that symbol (lane_codec.h, InorderModel::tag) in one instruction
__device__ __forceinline__ uint32_t byte_tag(uint32_t word, uint32_t kByte) {     // kByte: a constant once the caller's loop is unrolled
    uint32_t t;
    const uint32_t seven = 7u;
    if (kByte == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(t) : "v"(seven), "v"(word));
    if (kByte == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(t) : "v"(seven), "v"(word));
    if (kByte == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(t) : "v"(seven), "v"(word));
    if (kByte == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(t) : "v"(seven), "v"(word));
    return t;
}

// The top modeler's wavefront.
//
// Input fetch: 64 bytes per lane and CHUNK of eight phases, issued at least a chunk ahead of use, as
// back-to-back 16-byte loads.  Lanes sit 8192 bytes apart, so the lines all the packets of an XCD are
// reading at one moment fall into the same few L2 sets and do not survive until the lane comes back:
// every touch of a line is a fetch from memory (rocprofv3 FETCH_SIZE: 3.3x the input with 16-byte
// pieces and two modelers reading, 1.2-1.8x with 64-byte pieces, 1.01x now that one wavefront reads
// and takes the whole 128-byte line at a time).
__device__ __forceinline__ void run_top(EncodeLds &lds, const uint8_t *in, uint32_t lane, uint32_t len,
                                        uint32_t len_min, uint32_t n_phases) {
    constexpr uint32_t kChunkPhases = 8;                       // phases per fetch
    constexpr uint32_t kChunk = kChunkPhases * kPhase;         // 64 symbols = 64 bytes = 4 x 16-byte loads
    constexpr uint32_t kPieces = kChunk / 16u;
    TopModeler<7> model;
    uint32_t k = 0, slot = 0;
    {
        const uint32_t first = len ? load16_guarded(in, len).x & 0xFFu : 0u;
        model.open(lds.tree, 2u * lane_column(lane), first);
    }
    // ---- whole chunks that every lane of the wavefront owns completely ----
    const uint32_t full_chunks = len_min / kChunk;
    if (full_chunks) {
        const uint4 *src = reinterpret_cast<const uint4 *>(in);
        // A 128-byte line of the input holds two chunks.  Both halves are asked for together (at the start of every
        // odd chunk, for the two chunks behind it) so that a line is fetched from memory once: with one half per
        // chunk the line was gone from L2 by the time the lane came back for the other (8192 packets 8 KiB apart).
        auto fetch = [&](uint32_t chunk, uint4 (&into)[kPieces]) {
#pragma unroll
            for (uint32_t t = 0; t < kPieces; ++t) {
                const uint32_t at = chunk * kChunk + 16u * t;
                if (at + 16u <= len) into[t] = src[kPieces * chunk + t];
                else if (at < len) into[t] = load16_guarded(in + at, len - at);
                else into[t] = make_uint4(0, 0, 0, 0);
            }
        };
        uint4 c[kPieces], n[kPieces], nn[kPieces];            // this chunk, the next one, the one after (odd chunks only)
        fetch(0, c);
        fetch(1, n);
#pragma unroll
        for (uint32_t t = 0; t < kPieces; ++t) nn[t] = make_uint4(0, 0, 0, 0);
        for (uint32_t q = 0; q < full_chunks; ++q) {
            if (q & 1u) {                                     // wave-uniform
                fetch(q + 1u, n);
                fetch(q + 2u, nn);
            }
            uint32_t w[kChunk / 4u + 1u];
#pragma unroll
            for (uint32_t t = 0; t < kPieces; ++t) w[4 * t] = c[t].x, w[4 * t + 1] = c[t].y, w[4 * t + 2] = c[t].z, w[4 * t + 3] = c[t].w;
            w[kChunk / 4u] = n[0].x;
#pragma unroll
            for (uint32_t ph = 0; ph < kChunkPhases; ++ph) {
                uint32_t *out = &lds.sums[slot][0][lane];
                lds.bytes[ph & 1u][0][lane] = w[2 * ph];                 // k = kChunkPhases * q + ph, first term even
                lds.bytes[ph & 1u][1][lane] = w[2 * ph + 1];
#pragma unroll
                for (uint32_t j = 0; j < kPhase; ++j) {
                    const uint32_t i = ph * kPhase + j;                  // symbol index inside the chunk
                    // the successor's row tag straight out of its byte of the input word (one SDWA shift); this symbol's
                    // own tag is the one the step before formed
                    const uint32_t xn_tag = byte_tag(w[(i + 1) >> 2], (i + 1) & 3u);
                    out[j * kLanes] = model.step_tag(model.next_tag, 256u + q * kChunk + i, xn_tag);
                }
                slot = next_slot(slot);
                lds_barrier();
            }
#pragma unroll
            for (uint32_t t = 0; t < kPieces; ++t) c[t] = n[t], n[t] = nn[t];
        }
        k = kChunkPhases * full_chunks;
    }
    // ---- the rest: the phases that hold the file's ragged tail (or a partly dead wavefront), then two phases
    //      in which the other roles finish ----
    uint4 cur = make_uint4(0, 0, 0, 0), nxt = cur;
    if (k * kPhase < len) cur = load16_guarded(in + k * kPhase, len - k * kPhase);
    if (k * kPhase + 16u < len) nxt = load16_guarded(in + k * kPhase + 16u, len - (k * kPhase + 16u));
    for (; k < n_phases + 2u; ++k) {
        if (k < n_phases) {
            const uint32_t base = k * kPhase;
            const bool odd = (k & 1u) != 0u;                   // wave-uniform: second half of `cur`
            const uint32_t words[3] = {odd ? cur.z : cur.x, odd ? cur.w : cur.y, odd ? nxt.x : cur.z};
            if (odd) {                                         // `cur` is used up after this phase
                cur = nxt;
                const uint32_t ahead = base + kPhase + 16u;
                if (ahead < len) nxt = load16_guarded(in + ahead, len - ahead);
                else nxt = make_uint4(0, 0, 0, 0);
            }
            uint32_t *out = &lds.sums[slot][0][lane];
            lds.bytes[k & 1u][0][lane] = words[0];
            lds.bytes[k & 1u][1][lane] = words[1];
#pragma unroll
            for (uint32_t q = 0; q < 2; ++q) {
                uint32_t w = words[q], w_next = words[q + 1];
#pragma unroll 1
                for (uint32_t b = 0; b < 4; ++b) {
                    const uint32_t i = base + 4u * q + b;
                    const uint32_t x = w & 0xFFu;
                    w = (w >> 8) | (w_next << 24);            // next symbol now in the low byte
                    w_next >>= 8;
                    if (i < len) *out = model.step(x, 256u + i, w & 0xFFu);
                    out += kLanes;
                }
            }
            slot = next_slot(slot);
        }
        lds_barrier();
    }
}

// The low modeler: one phase behind the top one; input bytes and the top modeler's parts come through LDS, the
// sums go back to the same slot.  The last symbol of a phase does not know its successor yet (the top modeler is
// writing it in this very phase), so the node of a phase's first symbol is fetched when the phase begins.
__device__ __forceinline__ void run_low(EncodeLds &lds, uint32_t lane, uint32_t len, uint32_t len_min, uint32_t n_phases) {
    LowModeler<7> model;
    model.open(lds.tree, 2u * lane_column(lane), 0u);          // (the prefetch for symbol 0 is repeated below: harmless)
    lds_barrier();                                             // phase 0: the top modeler's first
    uint32_t slot = 0, k = 0;
    const uint32_t whole_phases = len_min / kPhase < n_phases ? len_min / kPhase : n_phases;    // (two loops: see the coder's)
    for (; k < whole_phases; ++k) {                            // the symbols of phase k, during phase k + 1
        const uint32_t base = k * kPhase;
        uint32_t *io = &lds.sums[slot][0][lane];
        {
            uint32_t part[kPhase], tag[kPhase];
            const uint32_t bytes[2] = {lds.bytes[k & 1u][0][lane], lds.bytes[k & 1u][1][lane]};
#pragma unroll
            for (uint32_t j = 0; j < kPhase; ++j) part[j] = io[j * kLanes], tag[j] = byte_tag(bytes[j >> 2], j & 3u);
            model.prime_tag(tag[0]);
#ifdef GPUAR_EXP_NO_LOW          // (timing experiments only)
            if (false)
#endif
#pragma unroll
            for (uint32_t j = 0; j < kPhase; ++j) {
                if (j + 1u < kPhase) io[j * kLanes] = model.step_tag(tag[j], 256u + base + j, tag[j + 1u], part[j]);
                else io[j * kLanes] = model.step_last_tag(tag[j], 256u + base + j, part[j]);
            }
        }
        slot = next_slot(slot);
        lds_barrier();
    }
    for (; k < n_phases; ++k) {                                // the phases that hold a ragged tail
        const uint32_t base = k * kPhase;
        uint32_t *io = &lds.sums[slot][0][lane];
        const uint32_t bytes[2] = {lds.bytes[k & 1u][0][lane], lds.bytes[k & 1u][1][lane]};
        {
#pragma unroll 1
            for (uint32_t j = 0; j < kPhase; ++j) {
                const uint32_t i = base + j;
                if (i < len) {
                    const uint32_t t = model.tree.tag((bytes[j >> 2] >> (8u * (j & 3u))) & 0xFFu);
                    model.prime_tag(t);
                    io[j * kLanes] = model.step_last_tag(t, 256u + i, io[j * kLanes]);
                }
            }
        }
        slot = next_slot(slot);
        lds_barrier();
    }
    lds_barrier();                                             // phase n_phases + 1: the coder's last
}

__global__ void __launch_bounds__(4 * kLanes)
encode_kernel(const uint8_t *__restrict__ src, size_t size, uint8_t *__restrict__ dst, uint32_t n_packets, uint32_t *__restrict__ status) {
    __shared__ EncodeLds lds;

    const size_t group = xcd_contiguous_group(blockIdx.x, gridDim.x);
    if (group * kLanes >= n_packets) return;                 // grid padding: the whole workgroup, before any barrier
    const uint32_t lane = threadIdx.x & 63u;
    // Role by SIMD, not by wavefront index.  The dispatcher puts the four wavefronts of a workgroup on
    // four different SIMDs (tools/placement_probe.hip); each workgroup draws a ticket from its CU's
    // arrival counter and leaves SIMD `ticket & 3` idle, the next three SIMDs take top modeler, low
    // modeler, coder.  Workgroups retire in arrival order (same work each), so the four resident ones
    // hold four consecutive tickets and every SIMD hosts exactly one wavefront of each role plus one
    // idle one -- three-wavefront workgroups land 4/3/3/2 on a third of the CUs instead.  Measured on
    // uniform 8 GiB: 28.0 ms (3 wavefronts) -> 27.2 (idle 4th) -> 26.7 (roles by SIMD).
    // If the four SIMD ids are ever not distinct, roles fall back to the wavefront index.
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));        // HW_ID
    const uint32_t simd = (hw >> 4) & 3u;
    uint32_t *hello = &lds.sums[0][0][0];                      // (the ring is not in use yet)
    if (lane == 0) hello[wave] = simd;
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // XCC_ID
        hello[4] = atomicAdd(&g_cu_ticket[((xcc & 7u) << 8) | ((hw >> 8) & 0xFFu)], 1u);
    }
    __syncthreads();
    const uint32_t seen = (1u << hello[0]) | (1u << hello[1]) | (1u << hello[2]) | (1u << hello[3]);
    const uint32_t by_simd = (simd - hello[4] - 1u) & 3u;
    const uint32_t role = __builtin_amdgcn_readfirstlane(seen == 0xFu ? by_simd : wave);   // 3 = idle
    __syncthreads();
    const size_t packet = group * kLanes + lane;
    const bool live = packet < n_packets;
    const size_t start = packet * kPacket;
    const uint32_t len = live ? static_cast<uint32_t>(size - start < kPacket ? size - start : kPacket) : 0u;
    const uint32_t len_max = wave_max(len);
    const uint32_t len_min = wave_max(~len) ^ 0xFFFFFFFFu;
    const uint32_t n_phases = (len_max + kPhase - 1) / kPhase;
    const uint8_t *in = src + (live ? start : 0);

    // every role meets n_phases + 2 barriers: the top modeler works in phases 0 .. n_phases - 1, the low one in
    // 1 .. n_phases, the coder in 2 .. n_phases + 1
    if (role == 0) {
        __builtin_amdgcn_s_setprio(kPrioTop);
        run_top(lds, in, lane, len, len_min, n_phases);
    } else if (role == 1) {
        __builtin_amdgcn_s_setprio(kPrioLow);
        run_low(lds, lane, len, len_min, n_phases);
    } else if (role == 3) {
        // The fourth wavefront carries the coder's reciprocals: the eight (multiplier, shift) pairs of a phase, 64 bytes of
        // the table, go into one of two slots in the tree's unused 256th row one barrier before the coder reads them --
        // lanes 0..15 load a dword each (asked for a whole interval ahead), write it, meet the barrier.
        uint32_t *courier = reinterpret_cast<uint32_t *>(lds.tree + 255u * 128u);
        const uint32_t *table = reinterpret_cast<const uint32_t *>(g_recip.r);
        const uint32_t lane16 = lane & 15u;
        clock_sample(0u, group, lane, 0u);                     // (this wavefront has the time, and lives as long as the workgroup)
        uint32_t carried = table[lane16];                      // the pairs of phase 0
        for (uint32_t k = 0; k < n_phases + 2u; ++k) {
            // interval k: the coder will work on the symbols of phase k - 1 during interval k + 1 and reads slot (k + 1) & 1
            if (lane < 16u) courier[((k + 1u) & 1u) * 16u + lane16] = carried;
            const uint32_t next_phase = k < n_phases ? k : 0u;                     // (phase k's pairs, for interval k + 1's write)
            carried = table[next_phase * 16u + lane16];
            lds_barrier();
        }
        clock_sample(0u, group, lane, 1u);
    } else {
        // ------------------------------- coder -------------------------------
        __builtin_amdgcn_s_setprio(kPrioCoder);
        // slot address = (wave-uniform base of this block's first slot) + lane * 8704
        uint8_t *block_slots = dst + group * (kLanes * kSlot);
#ifdef GPUAR_CODER_OWED_BITS            // (A/B builds only: the coder of rounds 1-3)
        CoderLane coder;
#else
        CarryCoderLane coder;                // the lower bound as a 64-bit window, carries instead of owed bits (lane_codec.h)
#endif
        coder.open(block_slots, lane * kSlot);
        lds_barrier();                                       // phases 0 and 1: the modelers' first
        lds_barrier();
        uint32_t slot = 0, k = 0;
        // Two loops, not one loop with two bodies: with both bodies in one loop the coder's eight state registers were copied
        // to the whole-phase body's own set at the top of every phase and back at its end (16 of ~280 vector instructions).
        const uint32_t whole_phases = len_min / kPhase < n_phases ? len_min / kPhase : n_phases;    // phases every lane owns completely
        for (; k < whole_phases; ++k) {                      // the symbols of phase k, during phase k + 2
            const uint32_t *in_ring = &lds.sums[slot][0][lane];
            {
                uint32_t cums[kPhase];
#pragma unroll
                for (uint32_t j = 0; j < kPhase; ++j) cums[j] = in_ring[j * kLanes];
                Recip rc[kPhase];
                {
                    // every lane reads the same 64 bytes: four broadcast reads, the pairs arrive as vector operands
                    const uint4 *slot = reinterpret_cast<const uint4 *>(lds.tree + 255u * 128u + (k & 1u) * 64u);
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q) {
                        const uint4 v = slot[q];
                        rc[2 * q] = {v.x, v.y}, rc[2 * q + 1] = {v.z, v.w};
                    }
                }
#ifdef GPUAR_CODER_OWED_BITS
#pragma unroll
                for (uint32_t j = 0; j < kPhase; ++j) coder.step(cums[j], rc[j]);
#else
                // the step in its three pieces (lane_codec.h): the NEXT symbol's two divisions sit between this symbol's
                // "who stores?" compare and the store region that reads the answer.
                // (Measured and not kept: a wave-uniform choice per phase between this and a store region that does not
                // clamp its address -- one vector instruction fewer per symbol while every lane has room for the phase's
                // eight dwords: +1 % with the reciprocals coming through LDS, +20 % while they came by scalar loads.)
                CarryCoderLane::Ahead next = coder.ahead(cums[0], rc[0]);
#ifdef GPUAR_EXP_NO_CODER        // (timing experiments only: what the kernel takes when the coder only meets its barriers)
                if (false)
#endif
#pragma unroll
                for (uint32_t j = 0; j < kPhase; ++j) {
                    const CarryCoderLane::Narrowed now = coder.narrow(next);
                    if (j + 1u < kPhase) next = coder.ahead(cums[j + 1u], rc[j + 1u]);
                    coder.settle(now);
                }
#endif
            }
            slot = next_slot(slot);
            lds_barrier();
        }
        for (; k < n_phases; ++k) {                          // the phases that hold a ragged tail
            const uint32_t base = k * kPhase;
            const uint32_t *in_ring = &lds.sums[slot][0][lane];
            {
#pragma unroll 1
                for (uint32_t j = 0; j < kPhase; ++j) {
                    const uint32_t i = base + j;
                    if (i >= len_max) break;
                    const Recip r = g_recip.r[i];
                    if (i < len) coder.step(in_ring[j * kLanes], r);
                }
            }
            slot = next_slot(slot);
            lds_barrier();
        }
        if (live) {
            bool overflowed;
            coder.finish(len, overflowed);
            if (overflowed) atomicOr(status, GPUAR_STATUS_SLOT_OVERFLOW);
        }
    }
}

// ---------------------------------------------------------------------------
// Encode, LATENCY mode: inputs too small to fill the chip (at most kSmallGroups groups of 64 packets).
//
// Such a launch takes as long as ONE packet: 8192 serial symbol steps of the slowest role, whatever the chip could do
// next to it -- the three-role kernel above needs ~355 cycles per step there (its coder's own chain), the same for 64
// packets as for 65536.  With the chip mostly idle and LDS plentiful, the step is cut finer instead: SIX working wavefronts
// per 64 packets, each one phase (16 symbols) behind the one before it.  A wavefront with few neighbours pays ~4.5
// cycles per vector instruction AND 12-20 per LDS operation, so the tree -- two LDS operations per
// level -- is what has to be spread thinnest:
//     UPPER    depths 1-2 of the tree; reads the input, hands the bytes on            13 vector + 5.5 LDS per symbol
//     MID1     depths 3-4, added onto the sums in place                               13 + 6
//     MID2     depths 5-6, added in place                                             13 + 6
//     LOW      depth 0 (register), depth 7, the x == 255 term, added in place         15 + 4
//     INTERVAL interval narrowing + renormalisation count -> one word per symbol      ~18 + 2   (lane_codec.h CarryIntervalLane)
//     SINK     the window of held bits, carries, stores                               ~19 + 1   (CarrySinkLane)
//     COURIER  (seventh wavefront) INTERVAL's reciprocals from the table into LDS, a phase ahead: no scalar load in a working role
// Same integers as the throughput kernel (lane_codec.h is shared; tests/test_lane_emulation.py pins the cut coder
// and a three-way tree against the oracle on the CPU), a different cut.  Rings: five slots of sums, four of input
// bytes, two of interval words: 64 KiB of LDS per workgroup with the tree.  Every role meets n_phases + 5 barriers.
// ---------------------------------------------------------------------------
constexpr uint32_t kSmallGroups = 512;       // up to 32768 packets = 256 MiB of input: at most two workgroups per CU
// (Measured and not kept, round 4: ONE tree level per role -- seven tree roles, ten wavefronts, phases of 8 symbols: 0.81 ms for
// 64 MiB against 0.71; every role reads and writes the sums and meets the barrier whatever it carries.)
constexpr uint32_t kSmallTreeRoles = 4;      // UPPER (depths 1-2), MID1 (3-4), MID2 (5-6), LOW (7, 0, the x == 255 term)
constexpr uint32_t kSmallLag = kSmallTreeRoles + 1u;   // the last role (SINK) works this many phases behind the first
constexpr uint32_t kSumSlots = kSmallLag;    // a slot of sums is alive from UPPER's phase to INTERVAL's, kSmallLag - 1 phases later
constexpr uint32_t kByteBufs = kSmallTreeRoles;   // the bytes of a phase are read by the other tree roles, up to kSmallTreeRoles - 1 phases later
constexpr uint32_t kSmallWaves = kSmallTreeRoles + 3u;   // + INTERVAL, SINK, COURIER
#ifndef GPUAR_SMALL_PHASE
#define GPUAR_SMALL_PHASE 16                 // (8: 0.71 ms for 64 MiB; 16: 0.68 -- half as many barriers; 64 KiB of LDS per workgroup then)
#endif
constexpr uint32_t kSmallPhase = GPUAR_SMALL_PHASE;   // symbols per phase: the roles meet at a barrier once per phase
static_assert(kSmallPhase == 8 || kSmallPhase == 16, "a phase is 2 or 4 input dwords; the courier's lanes carry one dword of reciprocals each");
struct alignas(16) EncodeSmallLds {
    uint8_t tree[kTreeRows * kLanes * 2];     // 32 KiB
    uint32_t sums[kSumSlots][kSmallPhase][kLanes];    // [slot][symbol][lane], cumLo | cumHi << 16 in the making
    uint32_t bytes[kByteBufs][kSmallPhase / 4][kLanes];    // the input bytes of a phase
    uint32_t words[2][kSmallPhase][kLanes];   // CarryIntervalLane -> CarrySinkLane: dn | n << 16 per symbol
    // (the courier's two slots -- INTERVAL's (multiplier, shift) pairs of a phase, up to 128 bytes -- are the tree's two unused
    // rows: row 127, where depth 0 would live if it were not a register, and row 255)
    __device__ uint32_t *recips(uint32_t parity) { return reinterpret_cast<uint32_t *>(tree + (parity ? 255u : 127u) * 128u); }
};

// the input bytes of one phase at in + at (a multiple of the phase length), zero beyond `len`
struct PhaseBytes {
    uint32_t w[kSmallPhase / 4];
};
__device__ __forceinline__ PhaseBytes load_phase(const uint8_t *in, uint32_t at, uint32_t len) {
    PhaseBytes r;
    if (at + kSmallPhase <= len) {
#pragma unroll
        for (uint32_t q = 0; q < kSmallPhase / 8; ++q) {
            const uint2 v = *reinterpret_cast<const uint2 *>(in + at + 8u * q);
            r.w[2 * q] = v.x, r.w[2 * q + 1] = v.y;
        }
        return r;
    }
#pragma unroll
    for (uint32_t q = 0; q < kSmallPhase / 4; ++q) {            // (static positions: the words stay in registers)
        uint32_t v = 0;
#pragma unroll
        for (uint32_t b = 0; b < 4u; ++b)
            if (at + 4u * q + b < len) v |= static_cast<uint32_t>(in[at + 4u * q + b]) << (8u * b);
        r.w[q] = v;
    }
    return r;
}
__device__ __forceinline__ uint32_t byte_of(const PhaseBytes &p, uint32_t j) {      // j static
    return (p.w[j >> 2] >> (8u * (j & 3u))) & 0xFFu;
}

template <int kUpperDepths>
__device__ __forceinline__ void small_upper(EncodeSmallLds &lds, const uint8_t *in, uint32_t lane, uint32_t len, uint32_t len_min,
                                            uint32_t n_phases) {
    PartialModeler<7, 1, kUpperDepths, 0, false> model;
    PhaseBytes cur, nxt = load_phase(in, 0, len);
    model.open(lds.tree, 2u * lane_column(lane), nxt.w[0] & 0xFFu);
    for (uint32_t k = 0; k < n_phases + kSmallLag; ++k) {
        if (k < n_phases) {
            const uint32_t base = k * kSmallPhase;
            cur = nxt;
            nxt = load_phase(in, base + kSmallPhase, len);
#pragma unroll
            for (uint32_t q = 0; q < kSmallPhase / 4; ++q) lds.bytes[k % kByteBufs][q][lane] = cur.w[q];
            uint32_t *out = &lds.sums[k % kSumSlots][0][lane];
            if (base + kSmallPhase <= len_min) {
#pragma unroll
                for (uint32_t j = 0; j < kSmallPhase; ++j)
                    out[j * kLanes] = model.step(byte_of(cur, j), 256u + base + j, j + 1u < kSmallPhase ? byte_of(cur, j + 1u) : nxt.w[0] & 0xFFu);
            } else {
#pragma unroll
                for (uint32_t j = 0; j < kSmallPhase; ++j)
                    if (base + j < len)
                        out[j * kLanes] = model.step(byte_of(cur, j), 256u + base + j, j + 1u < kSmallPhase ? byte_of(cur, j + 1u) : nxt.w[0] & 0xFFu);
            }
        }
        lds_barrier();
    }
}

// MIDDLE and LOW: `lag` phases behind UPPER; bytes and the sums so far come through LDS, the sums go back in place
template <typename Model>
__device__ __forceinline__ void small_follow(EncodeSmallLds &lds, uint32_t lane, uint32_t len, uint32_t len_min, uint32_t n_phases, uint32_t lag) {
    Model model;
    model.open(lds.tree, 2u * lane_column(lane), 0u);
    for (uint32_t b = 0; b < lag; ++b) lds_barrier();
    for (uint32_t k = 0; k < n_phases; ++k) {
        const uint32_t base = k * kSmallPhase;
        uint32_t *io = &lds.sums[k % kSumSlots][0][lane];
        PhaseBytes w;
#pragma unroll
        for (uint32_t q = 0; q < kSmallPhase / 4; ++q) w.w[q] = lds.bytes[k % kByteBufs][q][lane];
        if (base + kSmallPhase <= len_min) {
            uint32_t part[kSmallPhase];
#pragma unroll
            for (uint32_t j = 0; j < kSmallPhase; ++j) part[j] = io[j * kLanes];
            model.prime(w.w[0] & 0xFFu);
#pragma unroll
            for (uint32_t j = 0; j < kSmallPhase; ++j)
                io[j * kLanes] = j + 1u < kSmallPhase ? model.step(byte_of(w, j), 256u + base + j, byte_of(w, j + 1u), part[j])
                                                      : model.step_last(byte_of(w, j), 256u + base + j, part[j]);
        } else {
#pragma unroll
            for (uint32_t j = 0; j < kSmallPhase; ++j) {          // (unrolled: the byte's position must be static)
                const uint32_t x = byte_of(w, j);
                if (base + j < len) {
                    model.prime(x);
                    io[j * kLanes] = model.step_last(x, 256u + base + j, io[j * kLanes]);
                }
            }
        }
        lds_barrier();
    }
    for (uint32_t b = lag; b < kSmallLag; ++b) lds_barrier();
}

__global__ void __launch_bounds__(kSmallWaves * kLanes)
encode_small_kernel(const uint8_t *__restrict__ src, size_t size, uint8_t *__restrict__ dst, uint32_t n_packets, uint32_t *__restrict__ status) {
    __shared__ EncodeSmallLds lds;
    const size_t group = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t packet = group * kLanes + lane;
    const bool live = packet < n_packets;
    const size_t start = packet * kPacket;
    const uint32_t len = live ? static_cast<uint32_t>(size - start < kPacket ? size - start : kPacket) : 0u;
    const uint32_t len_max = wave_max(len);
    const uint32_t len_min = wave_max(~len) ^ 0xFFFFFFFFu;
    const uint32_t n_phases = (len_max + kSmallPhase - 1) / kSmallPhase;
    const uint8_t *in = src + (live ? start : 0);
    // The dispatcher deals a workgroup's wavefronts round the four SIMDs: wavefronts 0 and 4 share one, 1 and 5 another;
    // the four tree roles pair up there (one's LDS operations issue under the other's vector instructions), the two
    // coder roles have a SIMD each.
    constexpr uint32_t kWaveInterval = 2u, kWaveSink = 3u, kWaveCourier = 6u;
    if (wave == 1u) {
        small_upper<2>(lds, in, lane, len, len_min, n_phases);
    } else if (wave == 4u) {
        small_follow<PartialModeler<7, 3, 2, 0, false>>(lds, lane, len, len_min, n_phases, 1u);
    } else if (wave == 5u) {
        small_follow<PartialModeler<7, 5, 2, 0, false>>(lds, lane, len, len_min, n_phases, 2u);
    } else if (wave == 0u) {
        small_follow<DeepestModeler<7>>(lds, lane, len, len_min, n_phases, 3u);
    } else if (wave == kWaveCourier) {
        // The last wavefront carries INTERVAL's reciprocals (as encode_kernel's fourth carries the coder's): the eight pairs
        // of a phase go into one of two slots in the tree's unused 256th row one barrier before INTERVAL reads them, so that
        // the wavefront that owns the interval chain never waits for a scalar load.
        constexpr uint32_t kPhaseDwords = 2u * kSmallPhase;
        const uint32_t *table = reinterpret_cast<const uint32_t *>(g_recip.r);
        const uint32_t mine = lane & (kPhaseDwords - 1u);
        uint32_t carried = table[mine];
        for (uint32_t t = 0; t < n_phases + kSmallLag; ++t) {
            // INTERVAL works on phase p during interval p + kSmallLag - 1 and reads slot p & 1: the pairs of phase
            // t + 2 - kSmallLag are written during interval t (one barrier ahead), those of the phase behind it asked for
            constexpr uint32_t kAhead = kSmallLag - 2u;
            if (lane < kPhaseDwords) lds.recips((t + kAhead) & 1u)[mine] = carried;
            const uint32_t next_phase = t + 1u >= kAhead && t + 1u - kAhead < n_phases ? t + 1u - kAhead : 0u;
            carried = table[next_phase * kPhaseDwords + mine];
            lds_barrier();
        }
    } else if (wave == kWaveInterval) {
        CarryIntervalLane interval;
        interval.open();
        for (uint32_t b = 0; b < kSmallLag - 1u; ++b) lds_barrier();
        for (uint32_t k = 0; k < n_phases; ++k) {
            const uint32_t base = k * kSmallPhase;
            const uint32_t *sums = &lds.sums[k % kSumSlots][0][lane];
            uint32_t *out = &lds.words[k & 1u][0][lane];
            if (base + kSmallPhase <= len_min) {
                uint32_t cums[kSmallPhase];
#pragma unroll
                for (uint32_t j = 0; j < kSmallPhase; ++j) cums[j] = sums[j * kLanes];
                Recip rc[kSmallPhase];
                {
                    const uint4 *slot = reinterpret_cast<const uint4 *>(lds.recips(k & 1u));
#pragma unroll
                    for (uint32_t q = 0; q < kSmallPhase / 2; ++q) {
                        const uint4 v = slot[q];
                        rc[2 * q] = {v.x, v.y}, rc[2 * q + 1] = {v.z, v.w};
                    }
                }
#pragma unroll
                for (uint32_t j = 0; j < kSmallPhase; ++j) out[j * kLanes] = interval.step(cums[j], rc[j]);
            } else {
#pragma unroll 1
                for (uint32_t j = 0; j < kSmallPhase; ++j) {
                    if (base + j >= len_max) break;
                    const Recip r = g_recip.r[base + j];
                    if (base + j < len) out[j * kLanes] = interval.step(sums[j * kLanes], r);
                }
            }
            lds_barrier();
        }
        lds_barrier();
    } else if (wave == kWaveSink) {
        CarrySinkLane sink;
        sink.open(dst + group * (kLanes * kSlot), lane * kSlot);
        for (uint32_t b = 0; b < kSmallLag; ++b) lds_barrier();
        for (uint32_t k = 0; k < n_phases; ++k) {
            const uint32_t base = k * kSmallPhase;
            const uint32_t *in_words = &lds.words[k & 1u][0][lane];
            if (base + kSmallPhase <= len_min) {
                uint32_t w[kSmallPhase];
#pragma unroll
                for (uint32_t j = 0; j < kSmallPhase; ++j) w[j] = in_words[j * kLanes];
#pragma unroll
                for (uint32_t j = 0; j < kSmallPhase; ++j) sink.take(w[j]);
            } else {
#pragma unroll 1
                for (uint32_t j = 0; j < kSmallPhase; ++j)
                    if (base + j < len) sink.take(in_words[j * kLanes]);
            }
            lds_barrier();
        }
        if (live) {
            bool overflowed;
            sink.finish(len, overflowed);
            if (overflowed) atomicOr(status, GPUAR_STATUS_SLOT_OVERFLOW);
        }
    }
}

// ---------------------------------------------------------------------------
// Decode: replaces garDecompress + arDecompress (:916-934, 848-892)
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// The symbol step of the decoder, scheduled by hand for a wavefront that is ALONE on its SIMD
// (the per-packet model pins 36 KiB of LDS per wavefront, four wavefronts per CU).
//
// What tools/lat_probe.hip measures for such a wavefront (profiles/archive/r02_lat_probe.txt): every vector
// instruction costs one issue slot of 4.2-4.7 cycles whether or not it depends on its predecessor;
// `s_nop 0` costs a whole slot (4 cycles), `s_nop 1` two; a scalar instruction costs a slot as well;
// ds_read_b128 comes back after ~65 cycles and holds the issue port ~12, ds_write_b128 ~20.  So the
// step is priced in SLOTS, the two LDS round trips are free exactly when ~15 independent
// instructions sit behind each read, and anything the wavefront has to WAIT for is pure loss: a
// scalar load (thousands of cycles under load, and it can only be waited for with lgkmcnt(0)) or a
// vector load the whole wavefront waits for (~720 cycles under load, longer than the step) -- the
// loop below contains neither.  The compiler's own schedule of lane_codec.h's
// step_symbol spends ~135 slots per symbol (selects for the path bits, s_nop pads behind every lane
// mask it writes); the statements below spend 80 vector + 4.5 LDS (an even step and the odd one behind it: 82 + 4, 78 + 5).
//
//   R0 = off*total + total - 1; depths 0 and 1 (registers); READ #1 (mid record) issued
//       in its shadow: the half of the previous symbol's low record that its path took takes its increments
//       (one ds_add_u64: count +1, child +1 if left, grandchild +1 if left -- fields stay below 2^14, nothing
//       carries into a neighbour); even step: register nodes bumped, the stream reader's subtraction;
//       odd step: the stream reader moves on and reads its next dword
//   wait; mid record: 3 decisions; READ #2 (low record) issued
//       in its shadow: the mid record's half takes its increments the same way; even step: the stream window's selects and
//       refill; odd step: register nodes bumped
//   wait; low record: 3 decisions, the symbol's count for the upper bound; interval narrowed and renormalised;
//       off = ((off - dn) : window) << n   (which also steps the window over the n bits)
// What sits in which shadow was settled by timing (profiles/r06_decode_step_budget.txt): since round 6 both waits are worth
// 3-6 cycles, i.e. the step takes the time its instructions take to issue.
//
// Lane masks: v_sub_co writes "went left" as its borrow; v_min keeps the remainder; the mask is
// read two or more instructions later (path add-with-carry, selects of the next node and of the
// record update).  The LDS operands need aligned register quads / pairs and the 64-bit shift a
// pair: they are pinned (v200-v217); everything else is allocated by the compiler.  Both waits have
// the form "an LDS read, one LDS operation behind it, s_waitcnt lgkmcnt(1)" -- two and lgkmcnt(2) behind the odd step's
// first read, which has the stream read behind it as well -- (LDS operations of a wavefront complete in order).  Results are those of DecoderLane::step_symbol (same integers), which
// the CPU tests pin against the oracle; the GPU parity tests then compare this path with the oracle
// directly.
// ---------------------------------------------------------------------------
#define GPUAR_SDWA_W0 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"
#define GPUAR_SDWA_W1 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
#define GPUAR_SDWA_HALVES " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"

// One symbol of the hand-scheduled decoder (see above), in text pieces.  The step forms the whole 64-bit increment of
// its low half itself, in v204:v205, and the NEXT step applies it with one ds_add_u64 in the shadow of its first LDS read
// (profiles/archive/r03_decode_cost_attribution.txt: an LDS add costs this wavefront ~20 cycles to issue, a ds_write_b64 ~28, and
// the four vector instructions that used to rebuild the half's two dwords are gone; WAITED for, an LDS atomic is 60-440
// cycles dearer than a write, tools/lat_probe.hip).  One more instruction of a step waits for that shadow: clearing the
// top bit of lo (an instruction behind an LDS operation's issue costs this wavefront less than one in the chain; a variant
// that kept a select of the increment there needed a lane mask carried between statements in a scalar register).
// The same text serves the one wavefront of a file that holds its short last packet (lanes drop out under `if`).
// It uses decode_wave's locals by name.
// (timing experiments only, garbage out: profiles/r06_decode_step_budget.txt prices the step's LDS operations and waits by
// building the kernel without them, one kind at a time)
#ifdef GPUAR_EXP_NO_READS
#define GPUAR_LDS_READ(TEXT) ""
#else
#define GPUAR_LDS_READ(TEXT) TEXT
#endif
#ifdef GPUAR_EXP_NO_ADDS
#define GPUAR_LDS_ADD(TEXT) ""
#else
#define GPUAR_LDS_ADD(TEXT) TEXT
#endif
#ifdef GPUAR_EXP_NO_STREAM_READ
#define GPUAR_LDS_STREAM_READ(TEXT) ""
#else
#define GPUAR_LDS_STREAM_READ(TEXT) TEXT
#endif
// The model's total (256 + position, the same in every lane) reaches the step as a VECTOR register that the step itself counts up in
// an LDS shadow (round 6b).  Rounds 2-6a passed it as a scalar operand, which the compiler formed by one s_or per statement right in
// front of the step's first instruction -- on the chain, and a scalar instruction costs a lone wavefront a slot like any other.
#define GPUAR_HEAD_R0 \
            "v_mul_u32_u24_sdwa %[R0], %[off], %[totv]" GPUAR_SDWA_W0 /* off = the low half of lo : off */
#define GPUAR_HEAD_R0_ADD \
            "v_add3_u32 %[R0], %[R0], %[totv], -1\n\t" /* off*total + total - 1 */
#define GPUAR_TOTAL_STEP "v_add_u32 %[totv], 1, %[totv]\n\t"
#ifndef GPUAR_DEC_TOTAL_PLACE
#define GPUAR_DEC_TOTAL_PLACE 1          /* (A/B: 1 = counted up in the first LDS shadow, 2 = in the second; the same) */
#endif
#if GPUAR_DEC_TOTAL_PLACE == 1
#define GPUAR_TOTAL_STEP_EARLY GPUAR_TOTAL_STEP
#define GPUAR_TOTAL_STEP_LATE ""
#else
#define GPUAR_TOTAL_STEP_EARLY ""
#define GPUAR_TOTAL_STEP_LATE GPUAR_TOTAL_STEP
#endif
// A record read goes out as soon as its address is there: the remainder's minimum behind the decision in front of it (which the
// address does not need) is taken in the read's shadow (round 6b: 24.06 -> 23.87 ms; GPUAR_DEC_MIN_EARLY=1: in front of the read, as
// in rounds 2-6a).  The step's time is the vector instructions OUTSIDE the two LDS round trips + the round trips: both shadows are
// full (moving the instruction that files the symbol into the next step's first shadow changed nothing: 23.82 / 23.83 against
// 23.82 / 23.85), so only what shortens the stretch in front of a read still pays.
#ifdef GPUAR_DEC_MIN_EARLY
#define GPUAR_MIN_BEFORE_READ(TEXT) TEXT
#define GPUAR_MIN_BEHIND_READ(TEXT) ""
#else
#define GPUAR_MIN_BEFORE_READ(TEXT) ""
#define GPUAR_MIN_BEHIND_READ(TEXT) TEXT
#endif
#define GPUAR_A_HEAD \
            GPUAR_HEAD_R0 \
            "v_mul_u32_u24 %[t0], %[root], %[rng]\n\t" \
            GPUAR_HEAD_R0_ADD \
            "v_sub_co_u32 %[t1], %[m0], %[R0], %[t0]\n\t" /* borrow = went left at depth 0 */ \
            "v_min_u32 %[R], %[R0], %[t1]\n\t" \
            "v_cndmask_b32 %[t2], %[h1], %[h0], %[m0]\n\t" /* the depth-1 node on the path */ \
            "v_mul_u32_u24 %[t0], %[t2], %[rng]\n\t" \
            "v_sub_co_u32 %[t1], %[m1], %[R], %[t0]\n\t" \
            GPUAR_MIN_BEFORE_READ("v_min_u32 %[R], %[R], %[t1]\n\t") \
            "v_cndmask_b32 %[np], 0, 2, %[m0]\n\t" \
            "v_addc_co_u32 %[np], vcc, %[np], 0, %[m1]\n\t" /* complemented top two symbol bits */ \
            "v_lshl_add_u32 %[am], %[np], 10, %[col]\n\t" \
            GPUAR_LDS_READ("ds_read2_b64 v[200:203], %[am] offset1:64\n\t") /* READ #1: mid record, right half -> v200:201, left half -> v202:203 */ \
            GPUAR_MIN_BEHIND_READ("v_min_u32 %[R], %[R], %[t1]\n\t") /* (the remainder behind the second decision: the record's address does not need it) */ \

#define GPUAR_A_SHADOW_PLAIN \
         /* in its shadow: the half of the PREVIOUS symbol's low record its path went through takes its increments by ONE \
            64-bit LDS add (v204: +1 on the count, +0x10000 on the child if left; v205: the grandchild's, if left there); no field \
            can carry into its neighbour (counts stay below 2^14) */ \
            GPUAR_LDS_ADD("ds_add_u64 %[oaddr], v[204:205]\n\t")
// The same with the previous symbol's low half ADDRESSED here -- one shift-add off the end of the chain into a shadow in which
// the wavefront waits anyway (round 4: 26.64 -> 26.42 ms).  Measured and not kept: the filing of the previous symbol here
// as well (its last decision's lane mask saved by s_mov_b64 and put back into vcc in front of the SDWA add-with-carry:
// 26.83 ms) and, on top of that, the three selects that form the half's increments, from saved lane masks (27.89 ms): a
// lane mask that travels vector -> scalar -> vector costs more than the instructions it moves off the chain; the filing
// alone with the mask re-made here from the grandchild's increment (v_cmp_ne 0, v205 -- two instructions here for one on the
// chain): +3 cycles, this shadow has no room left.
#define GPUAR_A_SHADOW_DEFERRED \
            "v_lshl_add_u32 %[oaddr], %[c6], 9, %[collow]\n\t" \
            GPUAR_LDS_ADD("ds_add_u64 %[oaddr], v[204:205]\n\t")
#define GPUAR_REG_NODES \
         /* register nodes += went left: the root by the first decision's mask, of the two depth-1 nodes the one on the path by \
            the second one's -- which of them it is, is settled between the two lane masks by the scalar unit (both were written \
            a dozen instructions ago: no wait), so each node takes ONE add-with-carry (rounds 2-4: the chosen node's copy bumped, \
            then two selects to put it back) */ \
            "s_and_b64 %[sx], %[m0], %[m1]\n\t" \
            "s_andn2_b64 %[mj], %[m1], %[m0]\n\t" \
            "v_addc_co_u32 %[root], vcc, %[root], 0, %[m0]\n\t" \
            "v_addc_co_u32 %[h0], vcc, %[h0], 0, %[sx]\n\t" \
            "v_addc_co_u32 %[h1], vcc, %[h1], 0, %[mj]\n\t"
// Where a step bumps its register nodes: in its first LDS shadow (rounds 2-5) or in its second (vcc and mj are free there too).
// The even step keeps them in the first shadow; the odd step -- whose first shadow holds the stream reader's move and read and
// whose second would otherwise wait -- bumps them in the second (round 6 A/B on uniform 8 GiB: 25.13 -> 24.76 ms; both steps in
// the second shadow 25.20, only the even step 25.59).  After this both waits of a step are worth 3-6 cycles: the step's time is
// its instructions' issue time (profiles/r06_decode_step_budget.txt).
#ifndef GPUAR_DEC_NODES_PLACE
#define GPUAR_DEC_NODES_PLACE 1
#endif
#define GPUAR_NODES_NOWHERE ""
#if GPUAR_DEC_NODES_PLACE == 0
#define GPUAR_NODES_EVEN_EARLY GPUAR_REG_NODES
#define GPUAR_NODES_EVEN_LATE ""
#define GPUAR_NODES_ODD_EARLY GPUAR_REG_NODES
#define GPUAR_NODES_ODD_LATE ""
#elif GPUAR_DEC_NODES_PLACE == 1
#define GPUAR_NODES_EVEN_EARLY GPUAR_REG_NODES
#define GPUAR_NODES_EVEN_LATE ""
#define GPUAR_NODES_ODD_EARLY ""
#define GPUAR_NODES_ODD_LATE GPUAR_REG_NODES
#elif GPUAR_DEC_NODES_PLACE == 2
#define GPUAR_NODES_EVEN_EARLY ""
#define GPUAR_NODES_EVEN_LATE GPUAR_REG_NODES
#define GPUAR_NODES_ODD_EARLY ""
#define GPUAR_NODES_ODD_LATE GPUAR_REG_NODES
#elif GPUAR_DEC_NODES_PLACE == 3
#define GPUAR_NODES_EVEN_EARLY ""
#define GPUAR_NODES_EVEN_LATE GPUAR_REG_NODES
#define GPUAR_NODES_ODD_EARLY GPUAR_REG_NODES
#define GPUAR_NODES_ODD_LATE ""
#endif

// (the path after the record's first decision stays in `np` -- the two later ones go on in t3 -- so that the address of the
// half that takes the increments is formed behind read #2, next to the LDS add that uses it, and not on the chain)
#define GPUAR_MID_WRITEBACK \
            "v_addc_co_u32 %[t3], %[mj], %[t3], %[t3], %[mc]\n\t" \
            "v_lshl_add_u32 %[oaddr], %[t3], 10, %[collow]\n\t" \
            GPUAR_LDS_READ("ds_read2_b64 v[212:215], %[oaddr] offset1:64\n\t") /* READ #2: low record */ \
            GPUAR_MIN_BEHIND_READ("v_min_u32 %[R], %[R], %[t1]\n\t") \
         /* ---- the mid half takes its increments by one 64-bit LDS add in the shadow of read #2 */ \
            "v_lshl_add_u32 %[am], %[np], 9, %[col]\n\t" /* where that half lives: its index is the path up to the record's first decision */ \
            "v_cndmask_b32 v208, 1, %[k64k1], vcc\n\t" /* +1 on the half's count, +1 for bL/bR if left at the middle decision */ \
            "v_cndmask_b32 %[t2], 1, %[k64k], vcc\n\t" \
            "v_cndmask_b32 v209, 0, %[t2], %[mc]\n\t" \
            GPUAR_LDS_ADD("ds_add_u64 %[am], v[208:209]\n\t")
#ifdef GPUAR_EXP_NO_WAIT1      /* (timing experiments only: garbage out) */
#define GPUAR_WAIT1 ""
#else
#define GPUAR_WAIT1 "s_waitcnt lgkmcnt(1)\n\t"
#endif
#ifdef GPUAR_EXP_NO_WAIT2
#define GPUAR_WAIT2 ""
#else
#define GPUAR_WAIT2 "s_waitcnt lgkmcnt(1)\n\t"
#endif
#define GPUAR_BC_MID(WAIT1) \
            WAIT1 /* read #1 is back (LDS completes in order: at most the write behind it is left) */ \
         /* ---- mid record: v200 = aR | bR << 16, v201 = cRR | cRL << 16 (right half), v202 = a | bL << 16, v203 = cLR | cLL << 16 (left half) */ \
            "v_mul_u32_u24_sdwa %[t0], v202, %[rng]" GPUAR_SDWA_W0 \
            "v_sub_co_u32 %[t1], %[ma], %[R], %[t0]\n\t" \
            "v_min_u32 %[R], %[R], %[t1]\n\t" \
            "v_cndmask_b32 %[bw], v200, v202, %[ma]\n\t" /* the half the path takes: count and chosen child ... */ \
            "v_cndmask_b32 %[cc], v201, v203, %[ma]\n\t" /* ... and its two children */ \
            "v_mul_u32_u24_sdwa %[t0], %[bw], %[rng]" GPUAR_SDWA_W1 \
            "v_sub_co_u32 %[t1], vcc, %[R], %[t0]\n\t" \
            "v_min_u32 %[R], %[R], %[t1]\n\t" \
            "v_addc_co_u32 %[np], %[mj], %[np], %[np], %[ma]\n\t" \
            "v_cndmask_b32_sdwa %[t2], %[cc], %[cc], vcc" GPUAR_SDWA_HALVES \
            "v_mul_u32_u24 %[t0], %[t2], %[rng]\n\t" \
            "v_sub_co_u32 %[t1], %[mc], %[R], %[t0]\n\t" \
            GPUAR_MIN_BEFORE_READ("v_min_u32 %[R], %[R], %[t1]\n\t") \
            "v_addc_co_u32 %[t3], %[mj], %[np], %[np], vcc\n\t" \
            GPUAR_MID_WRITEBACK

// OWN_ADDRESS: the address of the step's own low half, formed at once (GPUAR_LOW_ADDRESS_NOW: the last symbol of a loop body)
// or left to the next step's first shadow (empty; GPUAR_A_SHADOW_DEFERRED)
#define GPUAR_LOW_ADDRESS_NOW \
            "v_lshl_add_u32 %[oaddr], %[c6], 9, %[collow]\n\t" /* the low half that takes the increments (in the next step's shadow) */
#define GPUAR_BC_LOW(OWN_ADDRESS, MUL) GPUAR_BC_LOW_WALK_TEXT(OWN_ADDRESS) GPUAR_BC_LOW_INTERVAL(MUL)
#define GPUAR_BC_LOW_WALK_TEXT(OWN_ADDRESS) \
            GPUAR_WAIT2 /* read #2 is back (behind it: the mid half's LDS add, perhaps the stream reader's dword) */ \
         /* ---- low record: v212 = aR | bR << 16, v213 = cRR | cRL << 16 (right half), v214 = a | bL << 16, v215 = cLR | cLL << 16 (left half). \
                 Next to the walk (three decisions on the scaled remainder) the symbol's own COUNT is picked out of the half: \
                 under the chosen side there are `count` symbols (a or aR), `child` of them left of the child node, the \
                 grandchild node holds the left one of the two leaves: count -> (lb ? child : count - child) -> (lc ? gc : that - gc). \
                 Two subtractions and two selects off the chain, then cnt * range is added to cumLo * range: one instruction fewer \
                 than carrying the scaled upper bound through the walk (round 2: Z = R - V, a sum, a product, a difference and three maxima). */ \
            "v_mul_u32_u24_sdwa %[pa], v214, %[rng]" GPUAR_SDWA_W0 \
            "v_sub_co_u32 %[t1], %[lma], %[R], %[pa]\n\t" \
            "v_min_u32 %[R], %[R], %[t1]\n\t" \
            "v_cndmask_b32 %[lbw], v212, v214, %[lma]\n\t" \
            "v_cndmask_b32 %[lcc], v213, v215, %[lma]\n\t" \
            "v_mul_u32_u24_sdwa %[pb], %[lbw], %[rng]" GPUAR_SDWA_W1 \
            "v_sub_co_u32 %[t1], vcc, %[R], %[pb]\n\t" \
            "v_sub_u32_sdwa %[ps], %[lbw], %[lbw] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t" /* count - child: right of the child node */ \
            "v_min_u32 %[R], %[R], %[t1]\n\t" \
            "v_addc_co_u32 %[c6], %[mj], %[t3], %[t3], %[lma]\n\t" /* the path after six decisions (c6, c7: kept for the next step's shadow) */ \
            OWN_ADDRESS \
         /* everything that hangs on the MIDDLE decision (vcc) comes first: the last decision's lane mask goes to vcc as well, \
            because the instruction that files the symbol takes its carry from there */ \
            "v_cndmask_b32_sdwa %[t2], %[lcc], %[lcc], vcc" GPUAR_SDWA_HALVES \
            "v_cndmask_b32_sdwa %[ps], %[ps], %[lbw], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t" /* symbols under the chosen grandchild node */ \
            "v_mul_u32_u24 %[pc], %[t2], %[rng]\n\t" \
            "v_addc_co_u32 %[c7], %[mj], %[c6], %[c6], vcc\n\t" \
            "v_cndmask_b32 v204, 1, %[k64k1], vcc\n\t" /* the increments of the low half -> v204:v205 (added in the next step's shadow) */ \
            "v_cndmask_b32 %[ti], 1, %[k64k], vcc\n\t" \
            "v_sub_co_u32 %[t1], vcc, %[R], %[pc]\n\t" \
            "v_sub_u32 %[t3], %[ps], %[t2]\n\t" /* the right leaf */ \
            "v_min_u32 %[R], %[R], %[t1]\n\t" \
            "v_cndmask_b32 %[t3], %[t3], %[t2], vcc\n\t" /* cnt(symbol) */
#define GPUAR_BC_LOW_INTERVAL(MUL) /* MUL: the name of the operand that holds this symbol's reciprocal multiplier */ \
         /* ---- applySymbolRange (:256-299) and the renormalisation (:787-836) */ \
            "v_sub_u32 %[t0], %[R0], %[R]\n\t" /* cumLo * range */ \
            "v_mad_u32_u24 %[t1], %[t3], %[rng], %[t0]\n\t" /* cumHi * range = cumLo * range + cnt * range */ \
            "v_mul_hi_u32 %[dn], %[t0], %[" MUL "]\n\t" \
            "v_mul_hi_u32 %[t1], %[t1], %[" MUL "]\n\t" \
            "v_lshrrev_b32 %[dn], %[shift], %[dn]\n\t" \
            "v_lshrrev_b32 %[t1], %[shift], %[t1]\n\t" \
            "v_mad_u32_u24 v217, %[dn], %[kffff], v217\n\t" /* lo : off -> (lo + dn) : (off - dn) in one: + dn * 0xFFFF (dn <= off, lo + dn < 2^16) */ \
            "v_sub_u32 %[wd], %[t1], %[dn]\n\t" /* new hi - new lo + 1 */ \
            "v_lshl_add_u32 %[t2], %[wd], 16, %[km32k]\n\t" /* (2 * width - 1) << 15: hi - lo with both one bit longer, at the top; its high half is width - 1 */ \
         /* (gfx950: a result written into HALF a register -- SDWA dst_sel -- may be read by the second instruction behind \
            its producer at the earliest; the assembler does not pad hand-written text, so the order below keeps that distance) */ \
            "v_add_u32_sdwa %[h], v217, %[t2] dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t" /* new hi (high half) */ \
            "v_ffbh_u32 %[e], %[t2]\n\t" /* lane_codec.h renorm_count: n = that count - 1 + [the bounds differ at that bit] */ \
            "v_xor_b32_sdwa %[kff], v217, %[h] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n\t"

// The end of the step: the grandchild's increment sits between the SDWA write of kff and its reader; the last instruction
// but one files the symbol: all eight complemented path bits = 2 * (the first seven) + the last decision's borrow, written
// straight into byte J of the output word (SDWA dst_sel, the other bytes preserved) -- no shift-or per symbol.
#define GPUAR_BC_LOW_END(N) /* N: the name of the operand this step leaves its bit count in ("ne" in an even step, "no" in an odd one) */ \
            "v_cndmask_b32 v205, 0, %[ti], vcc\n\t" \
            "v_lshlrev_b32 %[t2], %[e], %[kff]\n\t" \
            "v_lshrrev_b32 %[t2], 31, %[t2]\n\t" \
            "v_add3_u32 %[" N "], %[e], %[t2], -1\n\t" \
            "v_lshlrev_b32 %[rng], %[" N "], %[wd]\n\t"
// (not the statement's very last instruction: what reads the word behind the statement is the compiler's, and it does
// not know that the word was written by halves)
#define GPUAR_FILE_SYMBOL(J, WORD) \
            "v_addc_co_u32_sdwa %[" WORD "], vcc, %[c7], %[c7], vcc dst_sel:BYTE_" #J " dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"

// The stream reader.  What a step needs of the stream is the window v216: the next stream bits, left-aligned, of which the
// 64-bit shift at the end of the step (GPUAR_OFF_TEXT) moves the top n <= 16 into lo : off -- and leaves v216 = window << n,
// i.e. ALREADY stepped over the bits it took.  So v216 filled with 32 fresh bits lasts TWO symbols whatever they consume, and
// the reader runs at a fixed cadence, in all lanes alike, with no lane mask on exec.  (Rounds 1-5 refilled under a saved exec
// mask on every symbol -- 7 vector + 1 LDS + 2 scalar instructions per symbol, some lane needing it almost every time; now
// 10 + 1 + 0 per TWO symbols.)
//   EVEN step (symbols 0, 2, 4, ... of a block): the bits of the two symbols before it come off `rem` (unread bits of w0) in
//     one subtraction; its borrow says that w0 ran out: w1 moves up and takes the dword `ahead` (swapped to big-endian order
//     here); v216 = the 32 bits at `rem`.
//   ODD step: the lanes whose w0 ran out move their reader on by a dword, and every lane -- moved or not -- reads `ahead` again
//     from where its reader now stands (the lane's 64-byte ring in LDS, dword-major / lane-minor: dword d of lane l at ring
//     region + 256 * d + 4 * l, so the 64 lanes of a read hit 64 different banks whatever dwords they are at; `next` is the
//     byte offset of that dword times 64, i.e. already 256 * dword index; decode_wave keeps the ring filled).
// The even step's selects read the borrow as a lane mask (a scalar pair that only vector instructions read: nothing goes
// through the scalar unit); the odd step takes the same fact from the sign of the unwrapped difference, a vector register.  The read of `ahead` is never waited for by itself: it
// is older than the record reads of the next step, whose s_waitcnt lgkmcnt(1) (LDS operations complete in order) comes before
// the next even step looks at `ahead`.
// Where the pieces sit was settled by timing (profiles/r06_decode_step_budget.txt, uniform 8 GiB, one box, rounds 4-5's per-symbol
// reader 26.4-26.5 ms): the whole reader in the even step's second LDS shadow 25.8; split between the two steps' second shadows
// 25.8; the even step's subtraction moved to its first shadow 25.6; the odd step's part in ITS first shadow as well 25.2-25.3
// (kept); everything in first shadows 25.7.  A shadow hides three or four instructions, not ten.
#define GPUAR_STREAM_PART_A \
            "v_add_u32 %[pc], %[no], %[ne]\n\t" /* the bits of the two symbols since the last refill: <= 32 */ \
            "v_sub_co_u32 %[raw], %[sb], %[rem], %[pc]\n\t" /* borrow: w0 ran out */ \
            "v_and_b32 %[rem], 31, %[raw]\n\t"
#define GPUAR_STREAM_PART_B \
            "v_perm_b32 %[pa], 0, %[ahead], %[bsw]\n\t" /* big-endian order restored */ \
            "v_cndmask_b32 %[w0], %[w0], %[w1], %[sb]\n\t" \
            "v_cndmask_b32 %[w1], %[w1], %[pa], %[sb]\n\t" \
            "v_alignbit_b32 v216, %[w0], %[w1], %[rem]\n\t" /* the next 32 stream bits */
#define GPUAR_STREAM_PART_C \
            "v_lshrrev_b32 %[pb], 31, %[raw]\n\t" /* the difference before it was wrapped: negative where w0 ran out */ \
            "v_lshl_add_u32 %[next], %[pb], 8, %[next]\n\t" \
            "v_and_or_b32 %[pc], %[next], %[kf00], %[ring]\n\t" /* ring + 256 * (dword index mod 16) */ \
            GPUAR_LDS_STREAM_READ("ds_read_b32 %[ahead], %[pc]\n\t")
#ifndef GPUAR_DEC_STREAM_PLACE
#define GPUAR_DEC_STREAM_PLACE 6
#endif
#define GPUAR_STREAM_ODD_EARLY ""
#define GPUAR_STREAM_ODD_TEXT ""
#define GPUAR_WAIT1_GPUAR_STREAM_ODD_EARLY GPUAR_WAIT1
#define GPUAR_WAIT1_GPUAR_STREAM_EVEN_EARLY GPUAR_WAIT1
#ifdef GPUAR_EXP_NO_STREAM               /* (timing experiments only, garbage out: no stream instruction in any step) */
#define GPUAR_STREAM_EVEN_EARLY ""
#define GPUAR_STREAM_EVEN_TEXT ""
#elif GPUAR_DEC_STREAM_PLACE == 2        /* (A/B) all of it in the even step's second shadow */
#define GPUAR_STREAM_EVEN_EARLY ""
#define GPUAR_STREAM_EVEN_TEXT GPUAR_STREAM_PART_A GPUAR_STREAM_PART_B GPUAR_STREAM_PART_C
#elif GPUAR_DEC_STREAM_PLACE == 4        /* (A/B) the reader's move and the read of `ahead` one step later, in the ODD step's second shadow */
#define GPUAR_STREAM_EVEN_EARLY ""
#define GPUAR_STREAM_EVEN_TEXT GPUAR_STREAM_PART_A GPUAR_STREAM_PART_B
#undef GPUAR_STREAM_ODD_TEXT
#define GPUAR_STREAM_ODD_TEXT GPUAR_STREAM_PART_C
#elif GPUAR_DEC_STREAM_PLACE == 5        /* (A/B) as 4, with the subtraction in the even step's first shadow */
#define GPUAR_STREAM_EVEN_EARLY GPUAR_STREAM_PART_A
#define GPUAR_STREAM_EVEN_TEXT GPUAR_STREAM_PART_B
#undef GPUAR_STREAM_ODD_TEXT
#define GPUAR_STREAM_ODD_TEXT GPUAR_STREAM_PART_C
#elif GPUAR_DEC_STREAM_PLACE == 6        /* the subtraction in the even step's first shadow, the selects and the window in its second; the odd step's part in ITS first shadow (two LDS operations behind read #1) */
#define GPUAR_STREAM_EVEN_EARLY GPUAR_STREAM_PART_A
#define GPUAR_STREAM_EVEN_TEXT GPUAR_STREAM_PART_B
#undef GPUAR_STREAM_ODD_EARLY
#define GPUAR_STREAM_ODD_EARLY GPUAR_STREAM_PART_C
#undef GPUAR_WAIT1_GPUAR_STREAM_ODD_EARLY
#define GPUAR_WAIT1_GPUAR_STREAM_ODD_EARLY "s_waitcnt lgkmcnt(2)\n\t"
#elif GPUAR_DEC_STREAM_PLACE == 10       /* (A/B) as 6, the byte swap in the even step's first shadow as well */
#define GPUAR_STREAM_EVEN_EARLY GPUAR_STREAM_PART_A "v_perm_b32 %[pa], 0, %[ahead], %[bsw]\n\t"
#define GPUAR_STREAM_EVEN_TEXT \
            "v_cndmask_b32 %[w0], %[w0], %[w1], %[sb]\n\t" \
            "v_cndmask_b32 %[w1], %[w1], %[pa], %[sb]\n\t" \
            "v_alignbit_b32 v216, %[w0], %[w1], %[rem]\n\t"
#undef GPUAR_STREAM_ODD_EARLY
#define GPUAR_STREAM_ODD_EARLY GPUAR_STREAM_PART_C
#undef GPUAR_WAIT1_GPUAR_STREAM_ODD_EARLY
#define GPUAR_WAIT1_GPUAR_STREAM_ODD_EARLY "s_waitcnt lgkmcnt(2)\n\t"
#elif GPUAR_DEC_STREAM_PLACE == 7        /* (A/B) everything in first shadows */
#define GPUAR_STREAM_EVEN_EARLY GPUAR_STREAM_PART_A GPUAR_STREAM_PART_B
#define GPUAR_STREAM_EVEN_TEXT ""
#undef GPUAR_STREAM_ODD_EARLY
#define GPUAR_STREAM_ODD_EARLY GPUAR_STREAM_PART_C
#undef GPUAR_WAIT1_GPUAR_STREAM_ODD_EARLY
#define GPUAR_WAIT1_GPUAR_STREAM_ODD_EARLY "s_waitcnt lgkmcnt(2)\n\t"
#elif GPUAR_DEC_STREAM_PLACE == 9        /* (A/B) the even step's part in its second shadow, the odd step's in its first */
#define GPUAR_STREAM_EVEN_EARLY ""
#define GPUAR_STREAM_EVEN_TEXT GPUAR_STREAM_PART_A GPUAR_STREAM_PART_B
#undef GPUAR_STREAM_ODD_EARLY
#define GPUAR_STREAM_ODD_EARLY GPUAR_STREAM_PART_C
#undef GPUAR_WAIT1_GPUAR_STREAM_ODD_EARLY
#define GPUAR_WAIT1_GPUAR_STREAM_ODD_EARLY "s_waitcnt lgkmcnt(2)\n\t"
#endif

// lo' : off' = (((lo + dn) : (off - dn)) : window) << n, upper half, with lo's top bit cleared: ONE 64-bit shift moves both
// (v217 = lo << 16 | off, the window in v216).  What leaves lo at the top falls off the register; (off - dn + 1) << n
// <= width << n = range' <= 2^16 keeps the lower half inside its 16 bits; bit 31 is the last underflow position
// (lo' = (a << n) & 0x7FFF) and is cleared in the NEXT step's LDS shadow (GPUAR_A_SHADOW; decode_wave clears it once
// more behind the last step).
#define GPUAR_OFF_TEXT(N) \
            "v_lshlrev_b64 v[216:217], %[" N "], v[216:217]\n\t"

#define GPUAR_STEP_OPERANDS_COMMON \
              [R0] "=&v"(R0), [R] "=&v"(R), [np] "=&v"(np), [am] "=&v"(am), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), \
              [m0] "=&s"(m0), [m1] "=&s"(m1), [ma] "=&s"(ma), [mc] "=&s"(mc), [mj] "=&s"(mj), [sx] "=&s"(sx), [sb] "=&s"(sb), [raw] "+v"(rem_raw), \
              [root] "+v"(dec.model.root), [h0] "+v"(dec.model.half0), [h1] "+v"(dec.model.half1), \
              [rng] "+v"(dec.range), [off] "+v"(offr), [kff] "+v"(kff), [oaddr] "+v"(oaddr), \
              [rem] "+v"(dec.rem), [w0] "+v"(dec.w0), [w1] "+v"(dec.w1), [ahead] "+v"(dec.ahead), [next] "+v"(next64), "+v"(window), \
              [dn] "=&v"(dn), [bw] "=&v"(bw), [cc] "=&v"(cc), [pa] "=&v"(pa), [pb] "=&v"(pb), [pc] "=&v"(pc), [ps] "=&v"(ps), \
              [wd] "=&v"(wd), [h] "=&v"(h), [e] "=&v"(e)

#define GPUAR_STEP_LOCALS \
        uint32_t R0, R, np, am, t0, t1, t2, t3, dn, bw, cc, pa, pb, pc, ps, wd, h, e; \
        unsigned long long m0, m1, ma, mc, mj, sx, sb;

// The kinds of step in a loop body of 32 symbols.  By position: all but the last leave the address of their low half to the next
// step's first LDS shadow (the path after six decisions stays in a register of its own, `path6`), all but the first form it
// there for their predecessor.  By parity: even steps refill the stream window, odd steps live on what the even one left.
// J: the byte of its output word the step files its symbol in; MUL, WORD: the NAMES of the operands that hold the symbol's
// reciprocal multiplier and its output word (a run of eight steps is ONE asm statement, below); the step leaves the number of stream
// bits it took in "ne" (even steps) or "no" (odd steps): the even step's refill needs both.
#define GPUAR_N_EVEN "ne"
#define GPUAR_N_ODD "no"
#ifdef GPUAR_EXP_NO_SEARCH      /* (timing experiments only, garbage out: the step WITHOUT its symbol search -- no decision, no record read, no
                                   increment, no register-node update: what every decoder of this format pays per symbol whatever finds the symbol,
                                   profiles/r06_latency_decoder_prototype.txt) */
#define GPUAR_STEP_TEXT(SHADOW, PARITY, OWN_ADDRESS, J, MUL, WORD) \
            GPUAR_HEAD_R0 GPUAR_HEAD_R0_ADD GPUAR_TOTAL_STEP \
            "v_lshrrev_b32 %[R], 1, %[R0]\n\t" \
            "v_mov_b32 %[t3], 1\n\t" \
            "v_mov_b32 %[c7], %[c6]\n\t" \
            GPUAR_STREAM_##PARITY##_EARLY GPUAR_STREAM_##PARITY##_TEXT GPUAR_BC_LOW_INTERVAL(MUL) GPUAR_BC_LOW_END(GPUAR_N_##PARITY) \
            GPUAR_FILE_SYMBOL(J, WORD) GPUAR_OFF_TEXT(GPUAR_N_##PARITY)
#else
#define GPUAR_STEP_TEXT(SHADOW, PARITY, OWN_ADDRESS, J, MUL, WORD) \
    GPUAR_A_HEAD SHADOW GPUAR_TOTAL_STEP_EARLY GPUAR_NODES_##PARITY##_EARLY GPUAR_STREAM_##PARITY##_EARLY GPUAR_BC_MID(GPUAR_WAIT1_GPUAR_STREAM_##PARITY##_EARLY) \
    GPUAR_STREAM_##PARITY##_TEXT GPUAR_NODES_##PARITY##_LATE GPUAR_TOTAL_STEP_LATE GPUAR_BC_LOW(OWN_ADDRESS, MUL) GPUAR_BC_LOW_END(GPUAR_N_##PARITY) \
    GPUAR_FILE_SYMBOL(J, WORD) GPUAR_OFF_TEXT(GPUAR_N_##PARITY)
#endif
#ifdef GPUAR_DEC_NO_DEFER      /* (A/B builds: every step forms its own address, as in rounds 2-4a) */
#define GPUAR_STEP_FIRST(J, MUL, WORD) GPUAR_STEP_TEXT(GPUAR_A_SHADOW_PLAIN, EVEN, GPUAR_LOW_ADDRESS_NOW, J, MUL, WORD)
#define GPUAR_STEP_EVEN(J, MUL, WORD) GPUAR_STEP_FIRST(J, MUL, WORD)
#define GPUAR_STEP_ODD(J, MUL, WORD) GPUAR_STEP_TEXT(GPUAR_A_SHADOW_PLAIN, ODD, GPUAR_LOW_ADDRESS_NOW, J, MUL, WORD)
#define GPUAR_STEP_LAST(J, MUL, WORD) GPUAR_STEP_ODD(J, MUL, WORD)
#else
#define GPUAR_STEP_FIRST(J, MUL, WORD) GPUAR_STEP_TEXT(GPUAR_A_SHADOW_PLAIN, EVEN, , J, MUL, WORD)
#define GPUAR_STEP_EVEN(J, MUL, WORD) GPUAR_STEP_TEXT(GPUAR_A_SHADOW_DEFERRED, EVEN, , J, MUL, WORD)
#define GPUAR_STEP_ODD(J, MUL, WORD) GPUAR_STEP_TEXT(GPUAR_A_SHADOW_DEFERRED, ODD, , J, MUL, WORD)
#define GPUAR_STEP_LAST(J, MUL, WORD) GPUAR_STEP_TEXT(GPUAR_A_SHADOW_DEFERRED, ODD, GPUAR_LOW_ADDRESS_NOW, J, MUL, WORD)
#endif

// LDS of a decoder workgroup (one wavefront): the 64 models and the 64 stream rings, 40 KiB -> four per CU.
constexpr uint32_t kRingPieces = 4;                        // 16-byte pieces per lane: 64 bytes of stream
constexpr uint32_t kDecodeLdsQuads = (kDecodeRecords + kRingPieces) * kLanes;

// `base` is the same in every lane (4-byte aligned); lane offsets are 32-bit.  `col` = this lane's 8-byte
// column of the workgroup's LDS (72 half-records, 512 bytes apart); `ring` = this lane's dword 0 in the 4 KiB
// stream-ring region behind the records (16 dwords per lane, 256 bytes apart; the region is 4 KiB-aligned).
__device__ __forceinline__ void decode_wave(uint8_t *col, uint8_t *ring, const uint8_t *base, uint32_t pkt_off, uint32_t limit_off,
                                            uint8_t *out, bool live, uint32_t *status) {
    DecoderLane<9> dec;
    dec.open(col, base, pkt_off, limit_off, live);
    const uint32_t len_max = wave_max(dec.ulen);
    // Whole blocks of 64 symbols: the 64 output bytes gather in 16 registers and leave as four
    // back-to-back 16-byte stores, i.e. one whole 64-byte sector of this lane's output line
    // at a time (dword-at-a-time stores from 64 lanes at an 8 KiB stride were
    // measured to cost ~10x the output bytes in HBM writes: every partial
    // sector left L2 before its neighbours arrived).  A lane that does not own the whole block -- a
    // dead lane of the last wavefront, the file's short last packet -- sits the block out with its
    // state untouched (plain SIMT divergence), so one such lane no longer slows the other 63 down.
    uint32_t i = 0;

    // ---- state of the hand-scheduled step ----
    const uint32_t col_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(col));   // LDS byte address of this lane's column
    const uint32_t ring_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(ring));  // ... and of dword 0 of its stream ring
    // the refill's address arithmetic ORs a dword index into bits 8-11: the ring region must start on a 4 KiB
    // boundary of LDS (the kernels' only __shared__ array is declared that way); anything else is a build error
    // that would decode garbage, so the wavefront flags every packet bad and decodes nothing instead
    if ((ring_lds & 0xF00u) != 0u) {
        atomicOr(status, GPUAR_STATUS_BAD_PACKET);
        return;
    }
    register uint32_t o0 asm("v204");          // the increments the previous symbol's low half still has to take (v204:v205)
    register uint32_t o1 asm("v205");
    register uint32_t offr asm("v217");        // lo << 16 | (code - lo): v216:v217 is the pair the 64-bit shift works on, and
    offr = dec.off | (dec.lo << 16);           // lower bound and code offset move through the step as ONE register
    // the stream bits the last two symbols took and the reader has not stepped over yet: the even step's and the odd step's, a
    // register each (GPUAR_STREAM_EVEN_EARLY takes both off `rem` at once)
    uint32_t n_even = 0, n_odd = dec.owed_bits;
    uint32_t total_v = 256u;                   // the model's total as a vector register (the same in every lane): GPUAR_TOTAL_STEP
    uint32_t rem_raw = 0;                      // the even step's `rem` before it was wrapped into 0..31: negative where w0 ran out -- the odd step
                                               // behind it moves those lanes' reader on (a vector register: nothing crosses statements as a lane mask)
    register uint32_t window asm("v216");      // the stream bits in front of the reader, left-aligned: filled by every even step, shifted
    asm volatile("v_mov_b32 %0, 0" : "=v"(window));       // on by every step (the low half of the pair the 64-bit shift works on)
    const uint32_t col_low_lds = col_lds + SubtreeModel<9>::kLowBase;               // ... and of its first low half
    uint32_t oaddr = col_lds + dec.model.owed.at;                                   // where the half owed goes
    // (No check for "a code value no symbol owns" -- off >= range, where the reference stops decoding, :873-877 -- in
    // this loop: it cannot happen, whatever the bits are.  off < range holds at the start (off < 2^16 = range); the walk
    // finds s with cumLo*range <= R0 < cumHi*range for R0 = (off + 1)*total - 1 < range*total; then
    // dn = floor(cumLo*range/total) <= off and (off + 1)*total <= cumHi*range gives off + 1 <= up, so
    // 0 <= off - dn < up - dn = width, and appending n stream bits to both keeps ((off - dn) : bits) << n below
    // width << n.  Round 2 carried a per-symbol minimum for it; 65 million symbols of garbage never raised it.)
    uint32_t kff = 0xFFFFu;                    // low half stays 0xFFFF, high half is scratch of the renormalisation
    const uint32_t k64k = 0x10000u, k64k1 = 0x10001u;
    const uint32_t minus_half = 0xFFFF8000u;               // (2 * width - 1) << 15 = (width << 16) + this: see renorm_count
    uint32_t bswap_sel, ring_wrap, low_half;
    asm volatile("s_mov_b32 %0, 0xffff" : "=s"(low_half));
    asm volatile("s_mov_b32 %0, 0x00010203" : "=s"(bswap_sel));     // (through asm: a known constant would be spliced in as a literal)
    asm volatile("s_movk_i32 %0, 0xf00" : "=s"(ring_wrap));        // 256 * 15: the ring's dword index, scaled

    // ---- the stream ring ----
    // The step takes its stream dwords from a per-lane ring of 64 bytes in LDS, NOT from memory: a load from
    // memory that every symbol waits for (whichever lane asked for it) makes the symbol as long as a
    // round trip to L2, which at full load is LONGER than the step itself (~740 against ~600 cycles).
    // The ring is refilled here, every eight symbols, in pieces of 16 bytes that are asked for one phase
    // (eight symbols) before they are written to LDS: the vector-memory wait is for something issued
    // ~4000 cycles ago.  Offsets are counted from base16 = base rounded down to 16 bytes, so pieces are
    // aligned (an aligned piece that holds one readable byte never crosses a page); past the end of what
    // may be read the last such piece is repeated (a well-formed packet decodes the same whatever follows).
    // A lane consumes at most 16 bits per symbol (n = e + u <= 16: range' = width << n <= 2^16, whatever the
    // bits are) = 16 bytes per phase of EIGHT symbols (rounds 1-3 reckoned with 31 bits and ran the phase every
    // four), a phase brings 16.  The reader's position A is the offset of the dword `ahead` holds, as of the last even step
    // (which has taken off the bits of all symbols before it): that dword is read AGAIN at every even step until the reader
    // moves on, so everything from A on must stay in the ring.  Asking for piece P = `fill` once fill - A <= 48: (1) the piece
    // P overwrites, P - 64, ends at or below A when P is written a phase later (A only grows); (2) the gap fill - A at a phase
    // never falls below 33 (49 or more and nothing is asked for: the next phase sees 16 less at most; below that a piece is asked
    // for and the gap keeps its size), while the even steps up to the next phase read dwords below A + 16 + 4 <= fill, all
    // written by then.  The ring starts full from the piece that holds `ahead` (gap >= 49).
    const uint32_t skew16 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(base) & 15u);
    const uint8_t *base16 = base - skew16;
    const uint32_t next16 = dec.next + skew16;                   // offset from base16 of the dword after `ahead` ...
    // ... and, times 64 (bits 8-11 = 256 * (dword index mod 16)), of the dword `ahead` holds itself: the even step reads
    // `ahead` again from where the reader stands whether or not it has moved
    uint32_t next64 = (next16 - 4u) << 6;
    const uint32_t last_piece = (dec.last + skew16) & ~15u;
    const uint32_t first_piece = (next16 - 4u) & ~15u;           // the piece that holds `ahead`'s dword: the ring starts there
    uint32_t fill = first_piece + 16u * kRingPieces;             // offset of the next piece to ask for
    // piece `at` (a multiple of 16) = dwords 4p .. 4p+3 of the ring, p = (at / 16) mod 4: two ds_write2st64_b32
    register uint32_t q0 asm("v220");          // the piece asked for last (in flight, or already in the ring:
    register uint32_t q1 asm("v221");          // writing it a second time is harmless) ...
    register uint32_t q2 asm("v222");
    register uint32_t q3 asm("v223");
    uint32_t slot_lds = ring_lds;              // ... and the LDS address of its dword 0
#pragma unroll
    for (uint32_t k = 0; k < kRingPieces; ++k) {
        const uint32_t at = first_piece + 16u * k;
        const Quad q = load128(base16 + (at < last_piece ? at : last_piece));
        uint32_t *slot = reinterpret_cast<uint32_t *>(ring + ((at & 0x30u) << 6));
        slot[0] = q.w[0], slot[64] = q.w[1], slot[128] = q.w[2], slot[192] = q.w[3];
        if (k == kRingPieces - 1u) {
            q0 = q.w[0], q1 = q.w[1], q2 = q.w[2], q3 = q.w[3];
            slot_lds = ring_lds + ((at & 0x30u) << 6);
        }
    }
    // One phase, by hand (the compiler's version of it, two divergent branches and their bookkeeping, was 24 issue
    // slots per four symbols): the piece asked for last is (re)written to its place, then the lanes whose reader
    // is within 48 bytes of `fill` ask for the next piece -- a predicated region without a branch around it.
    // s_waitcnt vmcnt(0): the piece was asked for a phase ago; the block's output stores drain here as well.
// ... and, since round 4, the reciprocal multipliers of the run AFTER NEXT: eight dwords by two loads with a wave-uniform
// address (`mulbase` = the half block's first multiplier, a scalar pair; OFF = byte offset of that run) that put the same
// value into every lane of the eight registers of set SET (v224 + 8 * SET ..., named in the text as two tuples; the compiler
// knows them as the pinned variables m<SET>0..7) -- where the step's two v_mul_hi_u32 take it from directly.  The
// loads land before the next phase's s_waitcnt vmcnt(0), one whole run before they are used.  (Rounds 2-3 kept symbol j's
// multiplier in lane j of ONE register per block of 64 and fetched it with a v_readlane per symbol.)
// (timing experiments only, garbage out: the ring phase without its wait for the vector memory, without its LDS writes, without
// the multipliers' loads -- profiles/r06_decode_step_budget.txt, section 3b)
#ifdef GPUAR_EXP_NO_RING_VMCNT
#define GPUAR_RING_VMCNT ""
#else
#define GPUAR_RING_VMCNT "s_waitcnt vmcnt(0)\n\t"
#endif
#ifdef GPUAR_EXP_NO_RING_WRITES
#define GPUAR_RING_WRITES(TEXT) ""
#else
#define GPUAR_RING_WRITES(TEXT) TEXT
#endif
#ifdef GPUAR_EXP_NO_RING_MUL
#define GPUAR_RING_MUL(TEXT) ""
#else
#define GPUAR_RING_MUL(TEXT) TEXT
#endif
// (the phase is the tail of its run's asm statement -- GPUAR_DECODE_RUN8 --: it shares the run's operands `next`, `ring`, `t2`, `sx`)
#ifdef GPUAR_EXP_NO_RING        /* (timing experiments only, garbage out: no ring phase at all) */
#define GPUAR_RING_PHASE_TEXT(TUPLE_LO, TUPLE_HI) ""
#else
#define GPUAR_RING_PHASE_TEXT(TUPLE_LO, TUPLE_HI)                                                                    \
            "v_lshrrev_b32 %[rt], 6, %[next]\n\t"                                                                    \
            "v_sub_u32 %[rt], %[fill], %[rt]\n\t"                                                                    \
            "v_cmp_gt_u32 vcc, 49, %[rt]\n\t" /* (first: the scalar unit reads this mask five instructions later) */ \
            GPUAR_RING_VMCNT                                                                                         \
            GPUAR_RING_WRITES("ds_write2st64_b32 %[slot], v220, v221 offset1:1\n\t"                                  \
                              "ds_write2st64_b32 %[slot], v222, v223 offset0:2 offset1:3\n\t")                        \
            GPUAR_RING_MUL("global_load_dwordx4 " TUPLE_LO ", %[zero], %[mulbase] offset:%[roff]\n\t"                \
                           "global_load_dwordx4 " TUPLE_HI ", %[zero], %[mulbase] offset:%[roff]+16\n\t")              \
            "s_and_saveexec_b64 %[sx], vcc\n\t"                                                                      \
            "v_min_u32 %[rt], %[fill], %[lastp]\n\t"                                                                 \
            "global_load_dwordx4 v[220:223], %[rt], %[base]\n\t"                                                     \
            "v_and_b32 %[t2], 48, %[fill]\n\t"                                                                       \
            "v_lshl_add_u32 %[slot], %[t2], 6, %[ring]\n\t"                                                          \
            "v_add_u32 %[fill], 16, %[fill]\n\t"                                                                     \
            "s_or_b64 exec, exec, %[sx]"
#endif

    // The per-symbol reciprocal multipliers (wave-uniform) reach the symbol step WITHOUT scalar loads and -- since round 4 --
    // without a v_readlane: the ring phase above fetches the eight of the run after next into registers by vector loads
    // with a wave-uniform address (every lane gets the same dword), four sets of eight registers taking turns; the shift
    // that goes with a multiplier is the same for all 64 symbols of a block (RecipTable) and follows from the block's first
    // total; the model total is simply counted up.  Scalar loads were the costliest thing in this loop: a scalar load that
    // misses its cache goes to L2 like everything else, and while 1024 wavefronts stream their packets in and their
    // output out that round trip is thousands of cycles; it counts in lgkmcnt with the LDS operations, can
    // only be waited for with lgkmcnt(0), and even issued eight symbols ahead of its use it cost ~80 of
    // ~720 cycles per symbol (one group ahead: ~175 of 830; measured by taking the table walk out,
    // tools/kind_timing.py).  The waits of the symbol step have the form "an LDS read, one LDS operation
    // behind it, s_waitcnt lgkmcnt(1)", which holds whatever else is in flight.
    // four sets of eight pinned registers, v224 + 8 * set + j: set s holds the multipliers of run s of a half block
    register uint32_t m00 asm("v224"); register uint32_t m01 asm("v225"); register uint32_t m02 asm("v226"); register uint32_t m03 asm("v227"); register uint32_t m04 asm("v228"); register uint32_t m05 asm("v229"); register uint32_t m06 asm("v230"); register uint32_t m07 asm("v231");
    register uint32_t m10 asm("v232"); register uint32_t m11 asm("v233"); register uint32_t m12 asm("v234"); register uint32_t m13 asm("v235"); register uint32_t m14 asm("v236"); register uint32_t m15 asm("v237"); register uint32_t m16 asm("v238"); register uint32_t m17 asm("v239");
    register uint32_t m20 asm("v240"); register uint32_t m21 asm("v241"); register uint32_t m22 asm("v242"); register uint32_t m23 asm("v243"); register uint32_t m24 asm("v244"); register uint32_t m25 asm("v245"); register uint32_t m26 asm("v246"); register uint32_t m27 asm("v247");
    register uint32_t m30 asm("v248"); register uint32_t m31 asm("v249"); register uint32_t m32 asm("v250"); register uint32_t m33 asm("v251"); register uint32_t m34 asm("v252"); register uint32_t m35 asm("v253"); register uint32_t m36 asm("v254"); register uint32_t m37 asm("v255");
    uint32_t vzero;                                             // (a zero the compiler does not know: the loads' vector offset)
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
// Eight symbols (two output words) and the ring phase behind them, as ONE asm statement; RUN (0..3) is the run's static position
// in the half block.  (Rounds 2-6a: a statement per symbol.  Between two statements the compiler puts what it likes -- the s_or
// that formed the symbol's total, the halves of an s_add_u32 / s_addc_u32 pair, a wait state for a store hazard the statement
// next door cannot have -- and every scalar instruction or s_nop there is a slot on the chain: 24.9 -> 24.4 ms for the total as a
// self-counting vector register, and the rest with the statements joined.)
// scc is clobbered: s_and_b64 / s_andn2_b64 / s_and_saveexec_b64 / s_or_b64 write it, and the compiler DOES keep a carry alive
// across a statement when it has such a pair to spread (a build without the per-statement s_or faulted on exactly that).
#define GPUAR_DECODE_RUN8(RUN, FIRST_KIND, LAST_KIND, WORD_A, WORD_B, SET_AHEAD, TUPLE_LO, TUPLE_HI)                         \
        {                                                                                                            \
            GPUAR_STEP_LOCALS                                                                                        \
            uint32_t lbw_, lcc_, ti_, path7_, rt_;                                                                   \
            unsigned long long lma_;                                                                                 \
            asm volatile(                                                                                            \
                FIRST_KIND(0, "mul0", "wa") GPUAR_STEP_ODD(1, "mul1", "wa") GPUAR_STEP_EVEN(2, "mul2", "wa") GPUAR_STEP_ODD(3, "mul3", "wa") \
                GPUAR_STEP_EVEN(0, "mul4", "wb") GPUAR_STEP_ODD(1, "mul5", "wb") GPUAR_STEP_EVEN(2, "mul6", "wb") LAST_KIND(3, "mul7", "wb") \
                /* ... and the phase fetches the multipliers of the run after next: 8 * (RUN + 2) dwords behind the half block's first */ \
                GPUAR_RING_PHASE_TEXT(TUPLE_LO, TUPLE_HI)                                                            \
                : GPUAR_STEP_OPERANDS_COMMON,                                                                        \
                  [lbw] "=&v"(lbw_), [lcc] "=&v"(lcc_), [ti] "=&v"(ti_), [lma] "=&s"(lma_), [wa] "+v"(WORD_A), [wb] "+v"(WORD_B),     \
                  "+v"(o0), "+v"(o1), [c6] "+v"(path6), [c7] "=&v"(path7_), [ne] "+v"(n_even), [no] "+v"(n_odd), [totv] "+v"(total_v), \
                  [slot] "+v"(slot_lds), [fill] "+v"(fill), [rt] "=&v"(rt_), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3),               \
                  "=v"(m##SET_AHEAD##0), "=v"(m##SET_AHEAD##1), "=v"(m##SET_AHEAD##2), "=v"(m##SET_AHEAD##3),                     \
                  "=v"(m##SET_AHEAD##4), "=v"(m##SET_AHEAD##5), "=v"(m##SET_AHEAD##6), "=v"(m##SET_AHEAD##7)                      \
                : [mul0] "v"(m##RUN##0), [mul1] "v"(m##RUN##1), [mul2] "v"(m##RUN##2), [mul3] "v"(m##RUN##3),                     \
                  [mul4] "v"(m##RUN##4), [mul5] "v"(m##RUN##5), [mul6] "v"(m##RUN##6), [mul7] "v"(m##RUN##7),                     \
                  [shift] "s"(block_shift), [col] "v"(col_lds), [collow] "v"(col_low_lds), [ring] "v"(ring_lds),                  \
                  [k64k] "v"(k64k), [k64k1] "v"(k64k1), [bsw] "s"(bswap_sel), [kf00] "s"(ring_wrap), [km32k] "s"(minus_half), [kffff] "s"(low_half), \
                  [lastp] "v"(last_piece), [base] "s"(base16), [zero] "v"(vzero), [mulbase] "s"(mul_base), [roff] "n"(32 * ((RUN) + 2))   \
                : "vcc", "scc", "memory", "v200", "v201", "v202", "v203", "v208", "v209", "v212", "v213", "v214", "v215");         \
        }
// A block of 64 symbols as two half blocks of 32: the loop body is 32 symbols long, so every output word has a register
// of its own by name -- rounds 1-3 looped over runs of eight and filed each word into a register array by index (a
// v_not, an s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off and a scalar add per word, 1.5 issue slots per symbol, plus
// the loop's own six per eight symbols).
#ifdef GPUAR_EXP_NO_STORES      /* (timing experiments only: the decoded bytes never leave; a condition the compiler cannot fold) */
#define GPUAR_EXP_STORES (len_max > 0x7FFFFFF0u)
#else
#define GPUAR_EXP_STORES true
#endif
#define GPUAR_DECODE_BLOCK                                                                                           \
    {                                                                                                                \
        /* the shift that goes with the multipliers: floor(log2(total)) - 1, the same for all 64 totals of a block */ \
        const uint32_t block_shift = 30u - static_cast<uint32_t>(__builtin_clz(256u + i));                           \
        total_v = 256u + i; /* the model's total at the block's first symbol; every step counts it up */              \
        asm volatile("" : "+v"(total_v));                                                                            \
        _Pragma("unroll 1") for (uint32_t half = 0; half < 2u; ++half) {                                             \
            const uint32_t *mul_base = g_mul.m + i + 32u * half; /* wave-uniform: a scalar pair */                  \
            /* what a step leaves to the next one's first shadow: the path after six decisions (written by every step of   */ \
            /* the body, read by all but the first; defined here, for the compiler, without an instruction)                */ \
            uint32_t path6;                                                                                          \
            asm volatile("" : "=v"(path6));                                                                          \
            GPUAR_DECODE_RUN8(0, GPUAR_STEP_FIRST, GPUAR_STEP_ODD, w0, w1, 2, "v[240:243]", "v[244:247]")            \
            GPUAR_DECODE_RUN8(1, GPUAR_STEP_EVEN, GPUAR_STEP_ODD, w2, w3, 3, "v[248:251]", "v[252:255]")             \
            GPUAR_DECODE_RUN8(2, GPUAR_STEP_EVEN, GPUAR_STEP_ODD, w4, w5, 0, "v[224:227]", "v[228:231]")             \
            GPUAR_DECODE_RUN8(3, GPUAR_STEP_EVEN, GPUAR_STEP_LAST, w6, w7, 1, "v[232:235]", "v[236:239]")            \
            /* the block's 64 bytes leave TOGETHER, as four back-to-back 16-byte stores (a whole 64-byte sector: with two */ \
            /* stores per half block the L2 wrote 5 % and fetched 9 % more than the bytes): the first half's words wait,    */ \
            /* complemented, in k0..k7 (the path bits are the COMPLEMENTED symbol bits)                                     */ \
            if (half == 0u) {                                                                                        \
                k0 = ~w0, k1 = ~w1, k2 = ~w2, k3 = ~w3, k4 = ~w4, k5 = ~w5, k6 = ~w6, k7 = ~w7;                      \
            } else if (GPUAR_EXP_STORES) {                                                                           \
                uint4 *dst = reinterpret_cast<uint4 *>(out + i);                                                     \
                dst[0] = make_uint4(k0, k1, k2, k3);                                                                 \
                dst[1] = make_uint4(k4, k5, k6, k7);                                                                 \
                dst[2] = make_uint4(~w0, ~w1, ~w2, ~w3);                                                             \
                dst[3] = make_uint4(~w4, ~w5, ~w6, ~w7);                                                             \
            }                                                                                                        \
        }                                                                                                            \
    }

    // ---- blocks that every lane of the wavefront owns: uniform control flow ----
    const uint32_t len_min = wave_max(~dec.ulen) ^ 0xFFFFFFFFu;
    // v204:v205 = the 64-bit increment the previous symbol's low half still has to take (added in the shadow of the
    // next step's first read); nothing is owed yet: an add of zero to the half reset() named.
    // (Initial values go through asm: a known constant would be spliced into the statements as an immediate.)
    asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0" : "=v"(o0), "=v"(o1));
    // the eight output words of a half block (every byte of each is written before it is read; defined once for the compiler)
    uint32_t w0, w1, w2, w3, w4, w5, w6, w7;
    uint32_t k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0, k5 = 0, k6 = 0, k7 = 0;     // the first half block's words, until the second half's are there
    asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0\n\tv_mov_b32 %4, 0\n\tv_mov_b32 %5, 0\n\tv_mov_b32 %6, 0\n\tv_mov_b32 %7, 0"
                 : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7));
    // the multipliers of the packet's first two runs (symbols 0..15); from then on every run fetches those of the run after next
    asm volatile("global_load_dwordx4 v[224:227], %[zero], %[mulbase]\n\t"
                 "global_load_dwordx4 v[228:231], %[zero], %[mulbase] offset:16\n\t"
                 "global_load_dwordx4 v[232:235], %[zero], %[mulbase] offset:32\n\t"
                 "global_load_dwordx4 v[236:239], %[zero], %[mulbase] offset:48\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=v"(m00), "=v"(m01), "=v"(m02), "=v"(m03), "=v"(m04), "=v"(m05), "=v"(m06), "=v"(m07),
                   "=v"(m10), "=v"(m11), "=v"(m12), "=v"(m13), "=v"(m14), "=v"(m15), "=v"(m16), "=v"(m17)
                 : [zero] "v"(vzero), [mulbase] "s"(static_cast<const uint32_t *>(g_mul.m))
                 : "memory");
    for (; i + 64u <= len_min; i += 64u) {
        GPUAR_DECODE_BLOCK
    }
    // ---- the remaining whole blocks of a wavefront whose lanes differ in length (the file's short last packet,
    //      dead lanes of the last wavefront): lanes that do not own the block sit it out ----
    for (; i + 64u <= len_max; i += 64u) {
        if (i + 64u <= dec.ulen) GPUAR_DECODE_BLOCK
    }
#undef GPUAR_DECODE_BLOCK
#undef GPUAR_DECODE_RUN8

#undef GPUAR_RING_PHASE_TEXT
    // hand the state back to the plain step (the tail below, finish()); `ahead` may still be on its way from the ring
    // (and a piece the last ring phase asked for may still be on its way into v220-v223)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(dec.ahead), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : : "memory");
    dec.next = (next64 >> 6) + 4u - skew16;    // (next64 is the offset of `ahead`'s own dword, dec.next of the one behind it)
    dec.off = offr & 0xFFFFu;
    dec.lo = (offr >> 16) & 0x7FFFu;           // (the step leaves lo's top bit to the next step's shadow)
    dec.owed_bits = n_even + n_odd;            // what the last two symbols took is still to be stepped over (skip() takes up to 32 bits at once)
    // the increment still owed goes in now; what the plain step is then handed as "owed" is a rewrite of that half with
    // the values it holds (its write_back stores, it does not add)
    asm volatile("ds_add_u64 %[oaddr], v[204:205]\n\ts_waitcnt lgkmcnt(0)" : "+v"(o0), "+v"(o1) : [oaddr] "v"(oaddr) : "memory");
    dec.model.owed.at = oaddr - col_lds;
    {
        const Pair now = load64(col + dec.model.owed.at);
        dec.model.owed.w0 = now.w[0], dec.model.owed.w1 = now.w[1];
    }
    // the last, partial block of a packet whose length is not a multiple of 64 (at most one per file,
    // unless the packets are malformed): symbol by symbol, only the lanes that are inside such a block
    const uint32_t part_from = dec.ulen & ~63u;
    const uint32_t part_end = wave_max((dec.ulen & 63u) ? dec.ulen : 0u);
    const uint32_t part_begin = wave_max((dec.ulen & 63u) ? ~part_from : 0u) ^ 0xFFFFFFFFu;   // min over those lanes
    for (i = part_begin; i < part_end; ++i) {
        const DecodeConst k = g_decode.c[i];
        if (i >= part_from && i < dec.ulen) dec.step(i, k, out);
    }
    if (live) {
        dec.finish(out);
        if (dec.bad) atomicOr(status, GPUAR_STATUS_BAD_PACKET);
    }
}

__global__ void __launch_bounds__(kLanes)
decode_slots_kernel(const uint8_t *__restrict__ slots, uint32_t n_packets, size_t n_bytes, uint8_t *__restrict__ out, uint32_t *__restrict__ status) {
    __shared__ __attribute__((aligned(4096))) uint4 lds[kDecodeLdsQuads];    // 40 KiB: 72 half-records x 64 lanes x 8 B, then 4 KiB of stream rings
    const uint32_t lane = threadIdx.x;
    const size_t packet = static_cast<size_t>(blockIdx.x) * kLanes + lane;
    const bool live = packet < n_packets;
    const size_t group_at = static_cast<size_t>(blockIdx.x) * (kLanes * kSlot);
    const uint8_t *group_slots = slots + group_at;                                                // wave-uniform
    // what may be read: the lane's slot, cut short where the caller's buffer ends (the last slot of garDecompressExecutor's
    // `size` bytes may be a partial one, src/gpuar_kernel.cu:916-934)
    const size_t group_left = n_bytes - group_at;
    const uint32_t slot_end = (lane + 1u) * kSlot;
    const uint32_t limit_off = group_left < slot_end ? static_cast<uint32_t>(group_left) : slot_end;
    clock_sample(1u, blockIdx.x, lane, 0u);
    decode_wave(reinterpret_cast<uint8_t *>(lds) + 8u * lane, reinterpret_cast<uint8_t *>(lds + kDecodeRecords * kLanes) + 4u * lane, group_slots, lane * kSlot, limit_off,
                out + (live ? packet : 0) * static_cast<size_t>(kPacket), live, status);
    clock_sample(1u, blockIdx.x, lane, 1u);
}

// Decode from a back-to-back packet stream (the bytes after the 20-byte .gip
// header): lane p starts at stream + offsets[p].
__global__ void __launch_bounds__(kLanes)
decode_stream_kernel(const uint8_t *__restrict__ stream, const uint64_t *__restrict__ offsets,
                     uint32_t n_packets, uint8_t *__restrict__ out, uint32_t *__restrict__ status) {
    __shared__ __attribute__((aligned(4096))) uint4 lds[kDecodeLdsQuads];
    const uint32_t lane = threadIdx.x;
    const size_t packet = static_cast<size_t>(blockIdx.x) * kLanes + lane;
    const bool live = packet < n_packets;
    // wave-uniform base: the (4-byte aligned) start of this group's first packet
    const uint64_t first = offsets[static_cast<size_t>(blockIdx.x) * kLanes] & ~3ull;
    const uint32_t first_hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(first >> 32))));
    const uint32_t first_lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(first))));
    const uint8_t *group_stream = stream + ((static_cast<uint64_t>(first_hi) << 32) | first_lo);
    const uint64_t left = offsets[n_packets] - first;                       // bytes from the base to the end of the stream
    const uint32_t limit_off = left < 0x7FFFFFFFull ? static_cast<uint32_t>(left) : 0x7FFFFFFFu;
    const uint32_t pkt_off = live ? static_cast<uint32_t>(offsets[packet] - first) : 0u;
    clock_sample(1u, blockIdx.x, lane, 0u);
    decode_wave(reinterpret_cast<uint8_t *>(lds) + 8u * lane, reinterpret_cast<uint8_t *>(lds + kDecodeRecords * kLanes) + 4u * lane, group_stream, pkt_off, limit_off,
                out + (live ? packet : 0) * static_cast<size_t>(kPacket), live, status);
    clock_sample(1u, blockIdx.x, lane, 1u);
}

// ---------------------------------------------------------------------------
// Compaction: exclusive scan of the packet lengths, then a gather of the
// defined bytes of every slot into one back-to-back stream.
// ---------------------------------------------------------------------------
constexpr uint32_t kScanThreads = 256;
constexpr uint32_t kScanItems = 16;                        // packets per thread
constexpr uint32_t kScanTile = kScanThreads * kScanItems;  // 4096 packets per block
// Per-tile sums/prefixes live at the start of the OUTPUT stream buffer until the gather overwrites
// it (one u64 per 4096 packets, and the stream holds >= 4 bytes per packet): no global scratch,
// so compactions on different streams or devices never share state.

__device__ __forceinline__ uint32_t slot_clen(const uint8_t *slots, size_t p) {
    return *reinterpret_cast<const uint16_t *>(slots + p * kSlot);
}

template <typename T>
__device__ __forceinline__ T block_exclusive_scan(T v, T &block_total) {
    __shared__ T wave_sums[kScanThreads / kLanes];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    T incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T other = __shfl_up(incl, off);
        if (lane >= static_cast<uint32_t>(off)) incl += other;
    }
    if (lane == 63u) wave_sums[wave] = incl;
    __syncthreads();
    T before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kScanThreads / kLanes; ++w) {
        const T s = wave_sums[w];
        before += (w < wave) ? s : T(0);
        total += s;
    }
    __syncthreads();
    block_total = total;
    return before + incl - v;
}

__global__ void __launch_bounds__(kScanThreads)
scan_tile_sums_kernel(const uint8_t *__restrict__ slots, uint32_t n_packets, uint64_t *__restrict__ tile_prefix) {
    const size_t first = static_cast<size_t>(blockIdx.x) * kScanTile + threadIdx.x * kScanItems;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < kScanItems; ++k)
        if (first + k < n_packets) sum += slot_clen(slots, first + k);
    uint32_t total;
    block_exclusive_scan(sum, total);
    if (threadIdx.x == 0) tile_prefix[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kScanThreads)
scan_tile_prefix_kernel(uint32_t n_tiles, uint64_t *__restrict__ tile_prefix) {   // one block
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles; base += kScanThreads) {
        const uint32_t t = base + threadIdx.x;
        const uint64_t v = t < n_tiles ? tile_prefix[t] : 0;
        // a tile sum fits 32 bits (4096 * 8704); sums over tiles need 64
        uint64_t total;
        const uint64_t excl = block_exclusive_scan<uint64_t>(v, total);
        const uint64_t start = carry;
        if (t < n_tiles) tile_prefix[t] = start + excl;
        __syncthreads();
        if (threadIdx.x == 0) carry = start + total;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(kScanThreads)
scan_offsets_kernel(const uint8_t *__restrict__ slots, uint32_t n_packets, uint64_t *__restrict__ offsets,
                    const uint64_t *__restrict__ tile_prefix) {
    const size_t first = static_cast<size_t>(blockIdx.x) * kScanTile + threadIdx.x * kScanItems;
    uint32_t lens[kScanItems];
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < kScanItems; ++k) {
        lens[k] = (first + k < n_packets) ? slot_clen(slots, first + k) : 0u;
        sum += lens[k];
    }
    uint32_t total;
    uint64_t run = tile_prefix[blockIdx.x] + block_exclusive_scan(sum, total);
#pragma unroll
    for (uint32_t k = 0; k < kScanItems; ++k) {
        if (first + k < n_packets) offsets[first + k] = run;
        run += lens[k];
        if (first + k + 1 == n_packets) offsets[n_packets] = run;
    }
}

// One workgroup of 256 threads moves one packet, 16 bytes per lane and round (three rounds for a packet of uniform data,
// two for text): destination-aligned stores, the source read through byte-exact unaligned loads.  Rounds 1-3 gave
// every packet ONE wavefront that looped over it (four packets per workgroup): 3.43 ms for the 8.66 GB stream of the
// bench workload, 5.0 TB/s read + write; with a workgroup per packet (measured, uniform / text 8 GiB: 192 threads 3.37 /
// 2.14 ms, 256: 3.00 / 2.03, 320: 2.90 / 2.15, 384: 2.93 / 2.17, 448: 3.03 / 2.25, 576 = one quad per thread: 3.26 /
// 2.70) it is 5.76 TB/s = 72 % of the datasheet's 8 TB/s, 93 % of what a plain copy reaches on this part (6.18 TB/s).
#ifndef GPUAR_GATHER_THREADS
#define GPUAR_GATHER_THREADS 256
#endif
constexpr uint32_t kGatherThreads = GPUAR_GATHER_THREADS;
__global__ void __launch_bounds__(kGatherThreads)
gather_kernel(const uint8_t *__restrict__ slots, const uint64_t *__restrict__ offsets, uint32_t n_packets,
              uint8_t *__restrict__ stream) {
    const size_t packet = blockIdx.x;
    const uint8_t *src = slots + packet * kSlot;
    const uint64_t off = offsets[packet];
    const uint32_t len = static_cast<uint32_t>(offsets[packet + 1] - off);
    uint8_t *dst = stream + off;
    // head: bytes up to the first 16-byte boundary of dst
    uint32_t head = static_cast<uint32_t>((16u - (reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u);
    if (head > len) head = len;
    const uint32_t body = (len - head) & ~15u;
    const uint32_t tail = len - head - body;
    for (uint32_t at = threadIdx.x * 16u; at < body; at += kGatherThreads * 16u) {
        uint4 v;
        __builtin_memcpy(&v, src + head + at, 16);
        *reinterpret_cast<uint4 *>(dst + head + at) = v;
    }
    // the ragged ends: 32 lanes of the last wavefront
    const uint32_t spare = threadIdx.x - (kGatherThreads - 32u);
    if (spare < head) dst[spare] = src[spare];
    else if (spare >= 16u && spare - 16u < tail) dst[head + body + spare - 16u] = src[head + body + spare - 16u];
}

// ---------------------------------------------------------------------------
// Synthetic input streams; bench/test support.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix_word(uint64_t seed, uint64_t k) {
    uint64_t z = seed + k * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256)
generate_uniform_kernel(uint64_t seed, uint64_t first_word, size_t n, uint8_t *__restrict__ out) {
    const size_t w = static_cast<size_t>(blockIdx.x) * 256u + threadIdx.x;
    const size_t at = w * 8u;
    if (at >= n) return;
    const uint64_t v = splitmix_word(seed, first_word + w + 1u);
    if (at + 8u <= n) {
        *reinterpret_cast<uint64_t *>(out + at) = v;
    } else {
        for (size_t b = 0; at + b < n; ++b) out[at + b] = static_cast<uint8_t>(v >> (8u * b));
    }
}

struct ZipfTable {
    uint32_t cum[256];
    uint8_t sym[256];
};

// K ranks, weight floor(2^24 / r); each thread makes 8 bytes from 4 words
__global__ void __launch_bounds__(256)
generate_zipf_kernel(uint64_t seed, uint64_t first_word, size_t n, uint8_t *__restrict__ out,
                     ZipfTable table, uint32_t K) {
    __shared__ uint32_t cum[256];
    __shared__ uint8_t sym[256];
    cum[threadIdx.x] = threadIdx.x < K ? table.cum[threadIdx.x] : 0xFFFFFFFFu;
    sym[threadIdx.x] = table.sym[threadIdx.x];
    __syncthreads();
    const uint64_t W = table.cum[K - 1];
    const size_t g = static_cast<size_t>(blockIdx.x) * 256u + threadIdx.x;  // group of 4 words = 8 bytes
    const size_t at = g * 8u;
    if (at >= n) return;
    uint64_t packed = 0;
    for (uint32_t j = 0; j < 4; ++j) {
        const uint64_t word = splitmix_word(seed, first_word + g * 4u + j + 1u);
        for (uint32_t h = 0; h < 2; ++h) {
            const uint64_t u = h ? (word >> 32) : (word & 0xFFFFFFFFull);
            const uint32_t t = static_cast<uint32_t>((u * W) >> 32);
            uint32_t lo = 0, hi = K;   // first index with cum[idx] > t
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (cum[mid] > t) hi = mid; else lo = mid + 1;
            }
            packed |= static_cast<uint64_t>(sym[lo]) << (8u * (2u * j + h));
        }
    }
    if (at + 8u <= n) {
        *reinterpret_cast<uint64_t *>(out + at) = packed;
    } else {
        for (size_t b = 0; at + b < n; ++b) out[at + b] = static_cast<uint8_t>(packed >> (8u * b));
    }
}

// ---------------------------------------------------------------------------
// The roof bench.py quotes next to the datasheet's: a plain device-to-device copy, 16 bytes per lane, ONE quad per thread
// and one 4 KiB tile per workgroup ("confirm a practical peak on the box with a device-to-device
// copy and report both").  Of the shapes tried (tools/copy_probe.hip, profiles/r04_copy_probe.txt: hipMemcpyAsync 4.8 TB/s,
// grid-stride loops with 4-8 loads in flight 4.7-5.2, tiles of 4 or 8 quads per thread 3.7-5.6) this, the simplest one,
// is the fastest: 6.18 TB/s read + write on 8 GiB.  Measurement support, not part of the codec path.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n_quads) {
    const size_t q = static_cast<size_t>(blockIdx.x) * 256u + threadIdx.x;
    if (q < n_quads) dst[q] = src[q];
}

}  // namespace gpuar

// ===========================================================================
// C ABI (include/gpuar_hip.h)
// ===========================================================================
namespace {

thread_local int t_last_error = GPUAR_OK;

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_launch() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? GPUAR_OK : static_cast<int>(e);
}

// Where a launch reports SLOT_OVERFLOW / BAD_PACKET: the caller's own device word, or -- for callers that pass
// none (the reference-named executors) -- the current device's fallback word, which gpuar_hip_status() reads.
uint32_t *status_word(uint32_t *d_status) {
    if (d_status) return d_status;
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(gpuar::g_status)) != hipSuccess) return nullptr;
    return static_cast<uint32_t *>(p);
}

}  // namespace

extern "C" {

size_t gpuar_hip_packet_count(size_t n_bytes) { return (n_bytes + GPUAR_PACKET_BYTES - 1) / GPUAR_PACKET_BYTES; }

int gpuar_hip_encode_mode(const uint8_t *d_in, size_t n_bytes, uint8_t *d_slots, uint32_t *d_status, void *stream, int mode) {
    if (mode != GPUAR_MODE_AUTO && mode != GPUAR_MODE_THROUGHPUT && mode != GPUAR_MODE_LATENCY) return GPUAR_ERR_ARGUMENT;
    if (n_bytes == 0) return GPUAR_OK;
    if (!d_in || !d_slots) return GPUAR_ERR_ARGUMENT;
    if (!aligned16(d_in) || !aligned16(d_slots) || (reinterpret_cast<uintptr_t>(d_status) & 3u)) return GPUAR_ERR_ALIGNMENT;
    uint32_t *status = status_word(d_status);
    if (!status) return GPUAR_ERR_NO_DEVICE;
    const size_t n_packets = gpuar_hip_packet_count(n_bytes);
    if (n_packets > 0xFFFFFFFFull) return GPUAR_ERR_ARGUMENT;
    const uint32_t groups = static_cast<uint32_t>((n_packets + gpuar::kLanes - 1) / gpuar::kLanes);
    // Small inputs cannot fill the chip and take as long as one packet: left to itself (GPUAR_MODE_AUTO) such a launch
    // goes to the latency-mode kernel (six working roles and a courier, a shorter step).  The slots are the same bytes either way; the
    // caller's `mode` is the only switch (no environment is read here).
    const bool latency = mode == GPUAR_MODE_LATENCY || (mode == GPUAR_MODE_AUTO && groups <= gpuar::kSmallGroups);
    if (latency) {
        gpuar::encode_small_kernel<<<groups, gpuar::kSmallWaves * gpuar::kLanes, 0, static_cast<hipStream_t>(stream)>>>(
            d_in, n_bytes, d_slots, static_cast<uint32_t>(n_packets), status);
        return check_launch();
    }
    const uint32_t blocks = (groups + 7u) & ~7u;              // see xcd_contiguous_group
    gpuar::encode_kernel<<<blocks, 4 * gpuar::kLanes, 0, static_cast<hipStream_t>(stream)>>>(
        d_in, n_bytes, d_slots, static_cast<uint32_t>(n_packets), status);
    return check_launch();
}

int gpuar_hip_encode(const uint8_t *d_in, size_t n_bytes, uint8_t *d_slots, uint32_t *d_status, void *stream) {
    return gpuar_hip_encode_mode(d_in, n_bytes, d_slots, d_status, stream, GPUAR_MODE_AUTO);
}

// n_bytes: how much of d_slots may be read (n_packets * 8704, or less when the last slot is a partial one)
static int launch_decode_slots(const uint8_t *d_slots, size_t n_packets, size_t n_bytes, uint8_t *d_out, uint32_t *d_status, void *stream) {
    if (n_packets == 0) return GPUAR_OK;
    if (!d_slots || !d_out || n_packets > 0xFFFFFFFFull) return GPUAR_ERR_ARGUMENT;
    if (!aligned16(d_slots) || !aligned16(d_out) || (reinterpret_cast<uintptr_t>(d_status) & 3u)) return GPUAR_ERR_ALIGNMENT;
    uint32_t *status = status_word(d_status);
    if (!status) return GPUAR_ERR_NO_DEVICE;
    const uint32_t blocks = static_cast<uint32_t>((n_packets + gpuar::kLanes - 1) / gpuar::kLanes);
    gpuar::decode_slots_kernel<<<blocks, gpuar::kLanes, 0, static_cast<hipStream_t>(stream)>>>(
        d_slots, static_cast<uint32_t>(n_packets), n_bytes, d_out, status);
    return check_launch();
}

int gpuar_hip_decode(const uint8_t *d_slots, size_t n_packets, uint8_t *d_out, uint32_t *d_status, void *stream) {
    return launch_decode_slots(d_slots, n_packets, n_packets * static_cast<size_t>(GPUAR_SLOT_BYTES), d_out, d_status, stream);
}

int gpuar_hip_compact(const uint8_t *d_slots, size_t n_packets, uint8_t *d_stream, uint64_t *d_offsets, void *stream) {
    if (!d_offsets) return GPUAR_ERR_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_packets == 0) {
        const hipError_t e = hipMemsetAsync(d_offsets, 0, sizeof(uint64_t), s);
        return e == hipSuccess ? GPUAR_OK : static_cast<int>(e);
    }
    if (!d_slots || !d_stream) return GPUAR_ERR_ARGUMENT;
    if (!aligned16(d_slots) || (reinterpret_cast<uintptr_t>(d_offsets) & 7u) || (reinterpret_cast<uintptr_t>(d_stream) & 7u))
        return GPUAR_ERR_ALIGNMENT;
    // the gather launches one workgroup per packet, and gridDim.x * blockDim.x must stay below 2^32 threads: 16.7 M packets
    // = 128 GiB of input per call with 256-thread workgroups (a device holds 288 GB; the CLI compacts 512 MiB chunks)
    if (n_packets * static_cast<size_t>(gpuar::kGatherThreads) > 0xFFFFFFFFull) return GPUAR_ERR_ARGUMENT;
    const size_t tiles = (n_packets + gpuar::kScanTile - 1) / gpuar::kScanTile;
    const uint32_t np = static_cast<uint32_t>(n_packets);
    uint64_t *tile_prefix = reinterpret_cast<uint64_t *>(d_stream);   // scratch until the gather overwrites it
    gpuar::scan_tile_sums_kernel<<<static_cast<uint32_t>(tiles), gpuar::kScanThreads, 0, s>>>(d_slots, np, tile_prefix);
    gpuar::scan_tile_prefix_kernel<<<1, gpuar::kScanThreads, 0, s>>>(static_cast<uint32_t>(tiles), tile_prefix);
    gpuar::scan_offsets_kernel<<<static_cast<uint32_t>(tiles), gpuar::kScanThreads, 0, s>>>(d_slots, np, d_offsets, tile_prefix);
    gpuar::gather_kernel<<<np, gpuar::kGatherThreads, 0, s>>>(d_slots, d_offsets, np, d_stream);
    return check_launch();
}

int gpuar_hip_decode_stream(const uint8_t *d_stream, const uint64_t *d_offsets, size_t n_packets,
                            uint8_t *d_out, uint32_t *d_status, void *stream) {
    if (n_packets == 0) return GPUAR_OK;
    if (!d_stream || !d_offsets || !d_out || n_packets > 0xFFFFFFFFull) return GPUAR_ERR_ARGUMENT;
    if (!aligned16(d_out) || (reinterpret_cast<uintptr_t>(d_stream) & 3u) || (reinterpret_cast<uintptr_t>(d_status) & 3u)) return GPUAR_ERR_ALIGNMENT;
    uint32_t *status = status_word(d_status);
    if (!status) return GPUAR_ERR_NO_DEVICE;
    const uint32_t blocks = static_cast<uint32_t>((n_packets + gpuar::kLanes - 1) / gpuar::kLanes);
    gpuar::decode_stream_kernel<<<blocks, gpuar::kLanes, 0, static_cast<hipStream_t>(stream)>>>(
        d_stream, d_offsets, static_cast<uint32_t>(n_packets), d_out, status);
    return check_launch();
}

int gpuar_hip_status(uint32_t *flags) {
    if (!flags) return GPUAR_ERR_ARGUMENT;
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return static_cast<int>(e);
    uint32_t v = 0;
    e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(gpuar::g_status), sizeof v);
    if (e != hipSuccess) return static_cast<int>(e);
    const uint32_t zero = 0;
    e = hipMemcpyToSymbol(HIP_SYMBOL(gpuar::g_status), &zero, sizeof zero);
    *flags = v;
    return e == hipSuccess ? GPUAR_OK : static_cast<int>(e);
}

int gpuar_hip_last_error(void) {
    const int e = t_last_error;
    t_last_error = GPUAR_OK;
    return e;
}

const char *gpuar_hip_error_string(int code) {
    switch (code) {
        case GPUAR_OK: return "ok";
        case GPUAR_ERR_ALIGNMENT: return "device pointer is not suitably aligned (16 bytes)";
        case GPUAR_ERR_ARGUMENT: return "invalid argument";
        case GPUAR_ERR_NO_DEVICE: return "no HIP device";
        default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "unknown gpuar error";
    }
}

#ifdef GPUAR_EXPERIMENT_BUILD
const char *gpuar_hip_version(void) { return "gpuar-hip 0.2 gfx950 EXPERIMENT BUILD (timing switches: output may be garbage)"; }
#else
const char *gpuar_hip_version(void) { return "gpuar-hip 0.2 gfx950"; }
#endif

int gpuar_hip_abi_version(void) { return GPUAR_HIP_ABI_VERSION; }

int gpuar_hip_generate(int kind, uint64_t seed, uint64_t offset, size_t n, uint8_t *d_out, void *stream) {
    if (n == 0) return GPUAR_OK;
    if (!d_out || (offset & 7u) || kind < 0 || kind > 2) return GPUAR_ERR_ARGUMENT;
    if (reinterpret_cast<uintptr_t>(d_out) & 7u) return GPUAR_ERR_ALIGNMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t groups = (n + 7) / 8;
    const uint32_t blocks = static_cast<uint32_t>((groups + 255) / 256);
    if (kind == 0) {
        gpuar::generate_uniform_kernel<<<blocks, 256, 0, s>>>(seed, offset / 8, n, d_out);
    } else {
        static const char rank[] =
            " etaoinshrdlcumwfgypbvkjxqz\n.,ETAOINSHRDLCUMWFGYPBVKJXQZ0123456789-'\"()/:;=_<>[]{}!?#$%&*+@\\^`|~";
        gpuar::ZipfTable t;
        memset(&t, 0, sizeof t);
        const uint32_t K = kind == 1 ? 256u : 96u;
        uint32_t run = 0;
        for (uint32_t r = 1; r <= K; ++r) {
            run += (1u << 24) / r;
            t.cum[r - 1] = run;
            t.sym[r - 1] = kind == 1 ? static_cast<uint8_t>(((r - 1) * 167u + 13u) & 255u) : static_cast<uint8_t>(rank[r - 1]);
        }
        gpuar::generate_zipf_kernel<<<blocks, 256, 0, s>>>(seed, offset / 2, n, d_out, t, K);
    }
    return check_launch();
}

int gpuar_hip_copy(const uint8_t *d_src, uint8_t *d_dst, size_t n_bytes, void *stream) {
    if (n_bytes == 0) return GPUAR_OK;
    if (!d_src || !d_dst || (n_bytes & 15u)) return GPUAR_ERR_ARGUMENT;
    if (!aligned16(d_src) || !aligned16(d_dst)) return GPUAR_ERR_ALIGNMENT;
    const size_t n_quads = n_bytes / 16u;
    const size_t want = (n_quads + 255u) / 256u;                    // one quad per thread
    // HIP rejects a launch once gridDim.x * blockDim.x reaches 2^32 threads: 64 GiB less one tile for this kernel (bench.py
    // copies its 8 GiB workload); beyond that the caller gets an argument error instead of a launch error
    if (want * 256u > 0xFFFFFFFFull) return GPUAR_ERR_ARGUMENT;
    const uint32_t blocks = static_cast<uint32_t>(want);
    gpuar::copy_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(
        reinterpret_cast<const uint4 *>(d_src), reinterpret_cast<uint4 *>(d_dst), n_quads);
    return check_launch();
}

int gpuar_hip_clock_samples(int which, uint64_t *ticks, int reset) {
    if ((which != 0 && which != 1) || !ticks) return GPUAR_ERR_ARGUMENT;
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return static_cast<int>(e);
    const size_t bytes = sizeof(unsigned long long) * gpuar::kClockSlots * 4u, at = static_cast<size_t>(which) * bytes;
    e = hipMemcpyFromSymbol(ticks, HIP_SYMBOL(gpuar::g_clock_samples), bytes, at);
    if (e != hipSuccess) return static_cast<int>(e);
    if (reset) {
        static const unsigned long long zeros[gpuar::kClockSlots * 4u] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(gpuar::g_clock_samples), zeros, bytes, at);
    }
    return e == hipSuccess ? GPUAR_OK : static_cast<int>(e);
}

// ---- reference-named shims (src/gpuar.h:74,77,78) --------------------------
void initConstantRange(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        t_last_error = GPUAR_ERR_NO_DEVICE;
        return;
    }
    (void)hipFree(nullptr);  // forces context creation on the current device
}

void garCompressExecutor(const uint8_t *source, size_t size, uint8_t *destination, uint32_t numBlocks) {
    (void)numBlocks;
    const int e = gpuar_hip_encode(source, size, destination, nullptr, nullptr);
    if (e != GPUAR_OK) {
        t_last_error = e;
        fprintf(stderr, "garCompressExecutor: %s\n", gpuar_hip_error_string(e));
    }
}

void garDecompressExecutor(const uint8_t *source, size_t size, uint8_t *destination, uint32_t numBlocks) {
    (void)numBlocks;
    // the reference's kernel decodes every packet whose slot STARTS inside `size` (index * 8704 < size,
    // src/gpuar_kernel.cu:916-934): a ceiling, not a floor
    const int e = launch_decode_slots(source, (size + GPUAR_SLOT_BYTES - 1) / GPUAR_SLOT_BYTES, size, destination, nullptr, nullptr);
    if (e != GPUAR_OK) {
        t_last_error = e;
        fprintf(stderr, "garDecompressExecutor: %s\n", gpuar_hip_error_string(e));
    }
}

}  // extern "C"
