// lane_codec.h -- what ONE lane does for ONE packet, written once.
//
// Compiled by hipcc into the gfx950 kernels (gpuar_kernels.hip), where 64
// lanes run it in lock step over 64 packets.  The same source also compiles
// with a host compiler: tests/lane_emulation.cpp checks the closed forms below
// against the oracle on a machine without a GPU, and host/cpu_compressor.cpp
// uses it for the explicit `--host` mode of the CLI (the reference's
// CPUCompressor, src/cpu_compressor.cpp).  The GPU entry points never fall
// back to it.
//
// Semantics follow SURVEY.md section 8(a); reference lines are cited at each
// function (/root/reference/src/gpuar_kernel.cu unless noted).
#ifndef GPUAR_LANE_CODEC_H
#define GPUAR_LANE_CODEC_H

#include <stdint.h>
#include <string.h>

#include "gpuar_hip.h"
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define GPUAR_LANE __device__ __forceinline__
#define GPUAR_CLZ32(x) static_cast<uint32_t>(__clz(static_cast<int>(x)))
// leading zeros of a value that is never 0 (plain v_ffbh_u32: no guard for the all-zero case)
#define GPUAR_CLZ32_NZ(x) static_cast<uint32_t>(__builtin_clz(x))
#define GPUAR_MULHI(a, b) __umulhi((a), (b))
#define GPUAR_MUL24(a, b) __umul24((a), (b))     // both factors < 2^24: one full-rate multiply
// e = clz(((a ^ h) << 16) | 0xFFFF) and span = ~a | h as ONE statement: the xor lands in the high half of a register whose
// low half is preset to 0xFFFF (kff: low half 0xFFFF, high half scratch), so the count needs no guard for a zero argument.
// gfx950 forwards a result written into HALF a register (SDWA dst_sel) to the second instruction behind its producer at
// the earliest; the compiler pads its own code for that but cannot see into an asm statement, so the count is not left
// for it to place: the other bit-wise term of the renormalisation sits between the two.
#define GPUAR_AGREE_COUNT(kff, a, h, span) ([](uint32_t &k_, uint32_t a_, uint32_t h_, uint32_t &s_) { uint32_t e_; asm("v_xor_b32_sdwa %0, %3, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\tv_bfi_b32 %1, %3, %4, -1\n\tv_ffbh_u32 %2, %0" : "+v"(k_), "=&v"(s_), "=&v"(e_) : "v"(a_), "v"(h_)); return e_; }((kff), (a), (h), (span)))
// bit-field mask ((1 << w) - 1) << off, w and off taken mod 32: one instruction
#define GPUAR_BFM(w, off) ([](uint32_t w_, uint32_t o_) { uint32_t r_; asm("v_bfm_b32 %0, %1, %2" : "=v"(r_) : "v"(w_), "v"(o_)); return r_; }((w), (off)))
// the same where hipcc cannot see the 24-bit bound by itself (it would emit and + v_mul_lo_u32)
#define GPUAR_MUL24_VV(a, b) ([](uint32_t a_, uint32_t b_) { uint32_t r_; asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r_) : "v"(a_), "v"(b_)); return r_; }((a), (b)))
// a * b + c with 24-bit factors in ONE instruction (hipcc prefers separate multiplies and three-operand
// adds to shorten the dependency chain; these kernels are bound by instruction count, not by latency)
#define GPUAR_MAD24(a, b, c) ([](uint32_t a_, uint32_t b_, uint32_t c_) { uint32_t r_; asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r_) : "v"(a_), "v"(b_), "v"(c_)); return r_; }((a), (b), (c)))
// the same with the second factor uniform over the wavefront (a scalar register: no copy into a vector register)
#define GPUAR_MAD24_VS(a, b, c) ([](uint32_t a_, uint32_t b_, uint32_t c_) { uint32_t r_; asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r_) : "v"(a_), "s"(b_), "v"(c_)); return r_; }((a), (b), (c)))
// c | (b where bit `bit` of a is set, else 0): the bit spread over the register (v_bfe_i32, width 1), then and-or; b uniform
#define GPUAR_OR_WHERE_BIT(a, bit, b, c) ([](uint32_t a_, uint32_t b_, uint32_t c_) { uint32_t m_, r_; asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m_) : "v"(a_), "n"(bit)); asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r_) : "v"(m_), "s"(b_), "v"(c_)); return r_; }((a), (b), (c)))
// (a ^ 1) + b in one instruction; callers use only the low 16 bits
#define GPUAR_XOR1_ADD(a, b) ([](uint32_t a_, uint32_t b_) { uint32_t r_; asm("v_xad_u32 %0, %1, 1, %2" : "=v"(r_) : "v"(a_), "v"(b_)); return r_; }((a), (b)))
// low 32 bits of (hi:lo) >> (s & 31)
#define GPUAR_ALIGNBIT(hi, lo, s) __builtin_amdgcn_alignbit((hi), (lo), (s))
// the 16-bit field of v that starts at bit (off & 31); (a << (s & 31)) + c -- one instruction each
#define GPUAR_BFE16(v, off) ([](uint32_t v_, uint32_t o_) { uint32_t r_; asm("v_bfe_u32 %0, %1, %2, 16" : "=v"(r_) : "v"(v_), "v"(o_)); return r_; }((v), (off)))
#define GPUAR_LSHL_ADD(a, s, c) ([](uint32_t a_, uint32_t s_, uint32_t c_) { uint32_t r_; asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r_) : "v"(a_), "v"(s_), "v"(c_)); return r_; }((a), (s), (c)))
// x as a value the compiler cannot see through (keeps it from re-associating the expression x feeds)
#define GPUAR_OPAQUE_V(x) ([](uint32_t x_) { asm("" : "+v"(x_)); return x_; }((x)))
// the load that produces q is issued here, before any later store (no wait is implied)
#define GPUAR_PIN_LOAD(q) asm volatile("" : : : "memory")
// materialise x here and keep memory operations on their side of this point
#define GPUAR_PIN_ORDER(x) asm volatile("" : "+v"(x) : : "memory")
#else
#define GPUAR_LANE inline
#define GPUAR_CLZ32(x) ((x) ? static_cast<uint32_t>(__builtin_clz(x)) : 32u)
#define GPUAR_CLZ32_NZ(x) static_cast<uint32_t>(__builtin_clz(x))
#define GPUAR_AGREE_COUNT(kff, a, h, span) ((span) = ~(a) | (h), static_cast<uint32_t>(__builtin_clz(((((a) ^ (h)) << 16) | 0xFFFFu))))
#define GPUAR_MULHI(a, b) static_cast<uint32_t>((static_cast<uint64_t>(a) * (b)) >> 32)
#define GPUAR_MUL24(a, b) ((a) * (b))
#define GPUAR_MUL24_VV(a, b) ((a) * (b))
#define GPUAR_BFM(w, off) (((1u << ((w) & 31u)) - 1u) << ((off) & 31u))
#define GPUAR_XOR1_ADD(a, b) ((((a) ^ 1u)) + (b))
#define GPUAR_MAD24(a, b, c) ((a) * (b) + (c))
#define GPUAR_MAD24_VS(a, b, c) ((a) * (b) + (c))
#define GPUAR_OR_WHERE_BIT(a, bit, b, c) (((((a) >> (bit)) & 1u) ? (b) : 0u) | (c))
#define GPUAR_ALIGNBIT(hi, lo, s) static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | (lo)) >> ((s) & 31u))
#define GPUAR_BFE16(v, off) (((v) >> ((off) & 31u)) & 0xFFFFu)
#define GPUAR_LSHL_ADD(a, s, c) (((a) << ((s) & 31u)) + (c))
#define GPUAR_PIN_ORDER(x) ((void)0)
#define GPUAR_PIN_LOAD(q) ((void)0)
#define GPUAR_OPAQUE_V(x) (x)
#endif

namespace gpuar {

constexpr uint32_t kPacket = GPUAR_PACKET_BYTES;   // 8192
constexpr uint32_t kSlot = GPUAR_SLOT_BYTES;       // 8704
constexpr uint32_t kHdr = GPUAR_PACKET_HEADER_BYTES;
constexpr uint32_t kLanes = 64;
constexpr uint32_t kTreeRows = 256;                // rows of the encoder's node table (255 used)
constexpr uint32_t kDecodeRecords = 36;            // 16-byte subtree records of the decoder's model

// ---------------------------------------------------------------------------
// Exact division by the wave-uniform model total d = 256 + i, i in [0, 8192).
// For n < 2^30 and 2^(l-1) <= d < 2^l:  floor(n/d) == (n * m) >> (30 + l)
// with m = ceil(2^(30+l) / d) <= 2^31  (Granlund & Montgomery, N = 30: exact because m * d - 2^(30+l) < d <= 2^l).
// Numerators here are cum * range <= d * 65536 < 2^30 because the model total
// never exceeds 256 + 8192 < 2^14 (the guard of src/compressor.cpp:13-16).
// ---------------------------------------------------------------------------
struct Recip {
    uint32_t mul;
    uint32_t shift;  // applied to the high 32 bits of n * mul
};

struct RecipTable {
    Recip r[kPacket];
    constexpr RecipTable() : r{} {
        for (uint32_t i = 0; i < kPacket; ++i) {
            const uint64_t d = 256u + i;
            uint32_t l = 8;
            while ((1ull << l) <= d) ++l;   // 2^(l-1) <= d < 2^l: a power of two takes the shift of the totals after it,
                                            // so the shift is the same for all 64 symbols of a block (d = 256 + i, i = 64 b + j)
            const uint64_t m = ((1ull << (30 + l)) + d - 1) / d;
            r[i].mul = static_cast<uint32_t>(m);
            r[i].shift = l - 2;  // (n*m) >> (30+l) == hi32(n*m) >> (l-2)
        }
    }
};

GPUAR_LANE uint32_t div_total(uint32_t n, Recip rc) { return GPUAR_MULHI(n, rc.mul) >> rc.shift; }

// Closed form of the renormalisation loops (writeEncodedBits :321-367,
// readEncodedBits :787-836), used by CoderLane::step and DecoderLane::step_symbol.
// The loop first shifts out e = clz16(lo ^ hi) agreeing MSBs; after that the
// MSBs differ (lo: 0, hi: 1).  An underflow shift leaves lo's MSB 0 and hi's
// MSB 1 again, so no "agree" case can follow an underflow case: the loop is
// always e agree-shifts, then u underflow shifts, u = length of the run from
// bit 14 down where lo has 1 and hi has 0.  With hi kept as nh = 0xFFFF - hi,
// zeros enter both bounds, "lo ^ hi agrees" reads as leading ones of lo ^ nh,
// and the underflow run is the leading ones of lo & nh below bit 15.

//
// The decoder needs only n = e + u, and that is ONE count of leading zeros: write both bounds one bit longer, the
// way the loop extends them (a1 = 2a, h1 = 2h + 1, so a1 < h1 always), d1 = h1 - a1 = 2 * wd - 1.  With r = 16 - n
// positions left below the run, d1 = 2^r + X - Y (X, Y: what h1 and a1 hold there, X - Y > -2^(r-1) because the
// position behind the run is not another underflow position), so d1's top bit is at r -- where a1 and h1 differ (the
// run's last position, or the first differing one) -- or at r - 1, where they do not:
//     n = clz17(d1) - 1 + [a1 and h1 differ at d1's top bit].
// renorm_count() is that in 32-bit registers: d1 << 15 and (a ^ h) << 16 | 0xFFFF put both 17-bit numbers at the top.
// Checked against the loop for every a <= h < 65536 (tests/test_lane_emulation.py, emu_check_renorm_count).
GPUAR_LANE uint32_t renorm_count(uint32_t a, uint32_t wd) {
    const uint32_t h = a + wd - 1u;
    const uint32_t d1 = (wd << 16) - 0x8000u;                     // (2 * wd - 1) << 15, never 0 (wd >= 1)
    const uint32_t differ = (((a ^ h) & 0xFFFFu) << 16) | 0xFFFFu;  // (a1 ^ h1) << 15, ones behind it
    const uint32_t c = GPUAR_CLZ32_NZ(d1);
    return c + ((differ << (c & 31u)) >> 31) - 1u;
}

GPUAR_LANE uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

// typed accesses to addresses that are suitably aligned by construction
struct Quad {
    uint32_t w[4];
};
GPUAR_LANE Quad load128(const uint8_t *at) {
    Quad q;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4 v = *reinterpret_cast<const uint4 *>(at);
    q.w[0] = v.x, q.w[1] = v.y, q.w[2] = v.z, q.w[3] = v.w;
#else
    memcpy(q.w, at, 16);
#endif
    return q;
}
GPUAR_LANE void store128(uint8_t *at, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
#if defined(__HIP_DEVICE_COMPILE__)
    *reinterpret_cast<uint4 *>(at) = make_uint4(a, b, c, d);
#else
    const uint32_t w[4] = {a, b, c, d};
    memcpy(at, w, 16);
#endif
}
struct Pair {
    uint32_t w[2];
};
GPUAR_LANE Pair load64(const uint8_t *at) {
    Pair q;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint2 v = *reinterpret_cast<const uint2 *>(at);
    q.w[0] = v.x, q.w[1] = v.y;
#else
    memcpy(q.w, at, 8);
#endif
    return q;
}
GPUAR_LANE void store64(uint8_t *at, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    *reinterpret_cast<uint2 *>(at) = make_uint2(a, b);
#else
    const uint32_t w[2] = {a, b};
    memcpy(at, w, 8);
#endif
}
GPUAR_LANE void store16(uint8_t *at, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    *reinterpret_cast<uint16_t *>(at) = static_cast<uint16_t>(v);
#else
    const uint16_t h = static_cast<uint16_t>(v);
    memcpy(at, &h, 2);
#endif
}

// 32-bit load from an address that is 4-byte aligned on the GPU (any alignment on the host)
GPUAR_LANE uint32_t load32(const uint8_t *at) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<const uint32_t *>(at);
#else
    uint32_t v;
    memcpy(&v, at, 4);
    return v;
#endif
}

// 32-bit store to an address that is 4-byte aligned by construction
GPUAR_LANE void store32(uint8_t *at, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    *reinterpret_cast<uint32_t *>(at) = v;
#else
    memcpy(at, &v, 4);
#endif
}

// ===========================================================================
// ENCODER.  The per-symbol work of arCompress (:487-531) as three lane
// programs -- two partial modelers and a coder -- joined by 32-bit words
// (their sum per symbol is cumLo | cumHi << 16).  On the GPU they run in three
// wavefronts of a workgroup serving the same 64 packets through an LDS ring;
// on the host (tests, --host) one thread simply calls them in turn.
//
// Model = complete binary tree over the 256 symbols, each node holding the
// count of symbols in its LEFT subtree (u16).  For a symbol x the eight nodes
// on its path give cumLo = sum of the nodes where the path turns right, the
// same sum for x + 1 gives cumHi (the path of x + 1 leaves the path of x at
// x's lowest zero bit and has only zero bits below, so it can be taken over
// x's own nodes), and adding 1 to the nodes where the path turns left counts
// x.  Exactly the reference's counts (Fenwick getRange/update :215-238 and the
// all-ones start :403-419), in a container that needs one walk for all three.
// ===========================================================================

// Node addressing, IN-ORDER layout: the node at depth k on the path of symbol x
// sits in row (x & topmask_k) | ((1 << (7-k)) - 1), so its address is one
// AND-OR of the pre-shifted symbol plus a compile-time offset.  kRowShift =
// log2(bytes between rows): 7 on the GPU (64 lanes x u16, lane-minor, so lane
// l always hits LDS bank l & 31), 1 on the host.
template <uint32_t kRowShift>
struct InorderModel {
    uint8_t *table;      // first byte of the node table (same pointer in every lane of a wavefront)
    uint32_t lane_bits;  // this lane's byte offset inside a row (< 1 << kRowShift)

    // x_tag = x << kRowShift, formed once per symbol (on the GPU straight out of the input word: one SDWA shift of its
    // byte); the node address is then ONE and-or -- the tag's bits above the node's depth, this lane's column -- plus a
    // compile-time offset the LDS instruction carries as an immediate.  (Rounds 1-3 kept the column inside the tag, for an
    // address by a two-operand AND -- which needed an OR per symbol to put it there.  Issued between other wavefronts'
    // instructions the two- and the three-operand forms cost the same)
    GPUAR_LANE uint32_t tag(uint32_t x) const { return x << kRowShift; }
    GPUAR_LANE uint16_t *node(uint32_t x_tag, int k) const {
        const uint32_t keep = ((0xFF00u >> k) & 0xFFu) << kRowShift;
        const uint32_t fixed = ((1u << (7 - k)) - 1u) << kRowShift;
        return reinterpret_cast<uint16_t *>(table + ((x_tag & keep) | lane_bits) + fixed);
    }
};

// One partial modeler owns depths [kFirst, kFirst + kDepths) of the tree (its
// rows of the shared table are disjoint from the other's), optionally the
// kHead register-resident depths (0, 1 or 2 of depths 0 and 1) and the x == 255 term (kTail), and
// returns its part of cumLo | cumHi << 16 (both halves stay below 2^16, so the
// packed add of the parts cannot carry across).  Software-pipelined: while
// symbol i is being accounted, this part's nodes of symbol i+1 are already
// being fetched -- each fetch is issued right AFTER the store to the same depth
// for symbol i, and LDS operations of a wavefront complete in order, so a node
// shared by both symbols is read with symbol i's increment applied.  The node
// addresses travel with the prefetched values.
template <uint32_t kRowShift, int kFirst, int kDepths, int kHead, bool kTail>
struct PartialModeler {
    InorderModel<kRowShift> tree;
    uint32_t root, half0, half1;            // kHead == 1, 2: depth 0; kHead == 2: depth 1 as well
    uint32_t halves;                        // kHead == 3: depth 1 ALONE in a register, both nodes packed: x < 128 | x >= 128 << 16
    uint32_t left[kDepths];                 // this part's nodes of the NEXT symbol to account
    uint16_t *where[kDepths];

    // table: first byte of the shared node table; lane_bits: this lane's byte offset inside a row
    GPUAR_LANE void open(uint8_t *table, uint32_t lane_bits, uint32_t first_symbol) {
        tree.table = table;
        tree.lane_bits = lane_bits;
        // initial counts, own rows only (the other part initialises its own)
#pragma unroll 1
        for (uint32_t row = 0; row < 255u; ++row) {
            const uint32_t trailing_ones = 31u - GPUAR_CLZ32((row ^ (row + 1u)));
            const int depth = 7 - static_cast<int>(trailing_ones);
            if (depth >= kFirst && depth < kFirst + kDepths)
                *reinterpret_cast<uint16_t *>(table + (row << kRowShift) + lane_bits) = static_cast<uint16_t>(1u << trailing_ones);
        }
        root = 128u;
        half0 = half1 = 64u;
        halves = 64u * 0x10001u;
        prime(first_symbol);
    }

    // fetch this part's nodes of symbol x, the next one to account (what step() does for its x_next)
    GPUAR_LANE void prime(uint32_t x) { prime_tag(tree.tag(x)); }
    GPUAR_LANE void prime_tag(uint32_t x_tag) {
        next_tag = x_tag;
#pragma unroll
        for (int k = 0; k < kDepths; ++k) {
            where[k] = tree.node(x_tag, kFirst + k);
            left[k] = *where[k];
        }
    }

    // The path bits of x (low half) and of x + 1 (high half) side by side, kShift bits up: from the symbol itself
    // (kShift = 0) or from its row tag (x << kRowShift) | lane_bits (kShift = kRowShift; the lane bits stay below the
    // bits that are looked at, and adding 1 << kRowShift to the high half does not reach them).
    GPUAR_LANE static uint32_t paths_of_symbol(uint32_t x) { return GPUAR_MUL24(x, 0x10001u) + 0x10000u; }
    GPUAR_LANE static uint32_t paths_of_tag(uint32_t x_tag) { return GPUAR_MUL24(x_tag, 0x10001u) + (0x10000u << kRowShift); }

    // step() for a symbol whose successor is not known yet: nothing is fetched ahead, prime() has to follow
    GPUAR_LANE uint32_t step_last(uint32_t x, uint32_t total, uint32_t onto = 0) { return account<0, false>(paths_of_symbol(x), total, 0u, onto); }
    GPUAR_LANE uint32_t step_last_tag(uint32_t x_tag, uint32_t total, uint32_t onto = 0) {
        return account<static_cast<int>(kRowShift), false>(paths_of_tag(x_tag), total, 0u, onto);
    }

    // `onto`: what the sums are added to (the other modeler's part, when this one runs behind it)
    GPUAR_LANE uint32_t step(uint32_t x, uint32_t total, uint32_t x_next, uint32_t onto = 0) {
        return account<0, true>(paths_of_symbol(x), total, tree.tag(x_next), onto);
    }
    // the same for a caller that has the row tags already (both modelers of encode_kernel form them straight out of the input
    // words, one SDWA shift per symbol; the path bits come out of the tag by the same multiply-add)
    GPUAR_LANE uint32_t step_tag(uint32_t x_tag, uint32_t total, uint32_t xn_tag, uint32_t onto = 0) {
        return account<static_cast<int>(kRowShift), true>(paths_of_tag(x_tag), total, xn_tag, onto);
    }

    uint32_t next_tag;                      // row tag of the symbol whose nodes are held in left[] (the next to account)

  private:
    // z: path bits of x | path bits of x + 1, kShift bits up; kAhead: fetch the nodes of the symbol tagged xn behind the stores
    template <int kShift, bool kAhead>
    GPUAR_LANE uint32_t account(uint32_t z, uint32_t total, uint32_t xn, uint32_t onto) {
        uint32_t acc = onto;
        // x == 255: cumHi is the whole total.  Every OTHER term of cumHi is a count times a path bit of (x + 1) & 255 = 0 then,
        // in this part and in the one `onto` came from: the high half is empty and the total can be OR-ed in (two instructions
        // where shift, mask and multiply-add were three)
        if (kTail) acc = GPUAR_OR_WHERE_BIT(z, kShift + 24, total << 16, onto);
        if (kHead == 1 || kHead == 2) {
            const uint32_t pick0 = (z >> (kShift + 7)) & 0x10001u;
            acc = GPUAR_MAD24(root, pick0, acc);
            root = GPUAR_XOR1_ADD(pick0, root) & 0xFFFFu;
        }
        if (kHead == 3) {
            // Depth 1 in ONE register (round 5): its two nodes side by side, the one on the path picked by a bit-field extract
            // whose offset is 16 * (bit 7 of x), the count of x added by a shift-add with the same offset -- eight vector
            // instructions where the LDS-resident level costs five and two LDS operations (a read and a write that hold a lone
            // wavefront's issue for ~11 and ~19 cycles, profiles/r05_ldsbw_probe.txt).
            const uint32_t pick1 = (z >> (kShift + 6)) & 0x10001u;
            const uint32_t where1 = (z >> (kShift + 3)) & 16u;
            acc = GPUAR_MAD24(GPUAR_BFE16(halves, where1), pick1, acc);
            halves = GPUAR_LSHL_ADD(~pick1 & 1u, where1, halves);             // +1 where x goes left
        }
        if (kHead == 2) {
            static_assert(kHead != 2 || (kShift == 0 && kAhead), "depths 0 and 1 in registers: symbol-based step() only");
            const uint32_t x = z & 0xFFu;
            const bool upper_half = x >= 128u;
            const uint32_t pick1 = (z >> 6) & 0x10001u;
            acc = GPUAR_MAD24(upper_half ? half1 : half0, pick1, acc);
            const uint32_t quarter = x >> 6;
            half0 += quarter == 0u ? 1u : 0u;
            half1 += quarter == 2u ? 1u : 0u;
        }
#pragma unroll
        for (int k = 0; k < kDepths; ++k) {
            const uint32_t pick = (z >> (kShift + 7 - (kFirst + k))) & 0x10001u;
            const uint32_t l = left[k];
            acc = GPUAR_MAD24(l, pick, acc);
            *where[k] = static_cast<uint16_t>(GPUAR_XOR1_ADD(pick, l));   // +1 where x goes left
            if (kAhead) {
                where[k] = tree.node(xn, kFirst + k);
                left[k] = *where[k];
            }
        }
        if (kAhead) next_tag = xn;
        return acc;
    }
};

// How the tree is dealt to the two modelers (measured on the GPU, uniform 2 GiB).  Rounds 1-3, with the
// 43-instruction coder: 4 + 3 depths 5.71 ms, 5 + 2 5.43, 6 + 1 5.36, 7 + 0 5.75 -- the SIMDs issued a vector instruction in 98 %
// of their slots and the top modeler, which went first, carried what it could.  Round 4: the coder in carry form is ten vector
// instructions shorter and now goes LAST (issue priorities top > low > coder, gpuar_kernels.hip); 6 + 1 stays at 5.14 ms whatever
// the coder weighs (the top modeler's own stream -- 33 vector instructions and 14 LDS operations per symbol, each LDS operation
// holding its wavefront's issue for 12-20 cycles -- is as long as the phase), 5 + 2, 4 + 3 and 3 + 4 all take 4.96-4.99.
#ifndef GPUAR_TOP_DEPTHS
#define GPUAR_TOP_DEPTHS 4          // LDS-resident depths the top modeler walks (1 .. GPUAR_TOP_DEPTHS); the low modeler takes the rest
#endif
// (Round 5's timing experiments on this stream -- dummy instructions in the top modeler's step, an LDS add-with-return per level,
// depth 1 walked by nobody: profiles/r05_encoder_attribution.txt -- produced WRONG output and are no longer in this header, which the
// host build shares; they are in the history at cfd435d.)
#if defined(GPUAR_TOP_D1_REGS)    // depth 1 in a register of the top modeler, depths 2 .. GPUAR_TOP_DEPTHS in LDS
template <uint32_t kRowShift>
using TopModeler = PartialModeler<kRowShift, 2, GPUAR_TOP_DEPTHS - 1, 3, false>;
#else
template <uint32_t kRowShift>
using TopModeler = PartialModeler<kRowShift, 1, GPUAR_TOP_DEPTHS, 0, false>;                              // depths 1..
#endif
template <uint32_t kRowShift>
using LowModeler = PartialModeler<kRowShift, 1 + GPUAR_TOP_DEPTHS, 7 - GPUAR_TOP_DEPTHS, 1, true>;        // depth 0 (register), the deepest ones and the x == 255 term
// the last role of the latency kernel's four-way tree: depth 0 (register), depth 7 and the x == 255 term, whatever the split above
template <uint32_t kRowShift>
using DeepestModeler = PartialModeler<kRowShift, 7, 1, 1, true>;

// Range coder of one packet, fed with cumLo | cumHi << 16 per symbol.
// State: the interval as its lower bound and its WIDTH (lo, range = hi - lo + 1): both renormalise
// with one left shift each -- every step of writeEncodedBits (:321-367) takes the same constant off lo
// and hi and doubles them, so hi - lo + 1 just doubles; lo' = (lo << n) & 0x7FFF (closed form above bswap32).
// Output bits gather in a 32-bit accumulator; a full dword leaves with one (predicated) store.
// (A 64-bit accumulator needs fewer instructions but 64-bit shifts on every symbol: measured 4 % slower.)
struct CoderLane {
    uint32_t lo;         // lower bound (< 2^15 between symbols)
    uint32_t range;      // hi - lo + 1  (2^14 < range <= 2^16 between symbols)
    uint32_t pending;    // underflow bits owed
    uint32_t acc;        // the n (< 32) waiting output bits, right-aligned
    uint32_t n;
    uint32_t at;         // offset from `base` of the next dword to store
    uint32_t last;       // offset of the last dword of the slot: stores beyond it land there
    uint32_t kff;        // low half 0xFFFF, high half scratch (GPUAR_AGREE_COUNT)
    uint8_t *base;       // same pointer in every lane of a wavefront (scalar register on the GPU)
    uint32_t body_off;   // this lane's byte offset of slot + 4

    GPUAR_LANE void open(uint8_t *uniform_base, uint32_t slot_offset) {
        lo = 0;           // lo = 0, hi = 0xFFFF  (:492-494)
        range = 0x10000u;
        pending = 0;
        acc = 0;
        n = 0;
        kff = 0xFFFFu;
        base = uniform_base;
        body_off = slot_offset + kHdr;
        at = body_off;
        last = slot_offset + kSlot - 4u;
    }

    // append `count` (<= 32) bits, MSB first.  Straight-line except for the
    // one predicated region around the store.  A packet that outgrows its slot
    // keeps overwriting the slot's last dword (never beyond it); finish() sees
    // at > last and reports the overflow.
    GPUAR_LANE void put(uint32_t bits, uint32_t count) {
        const uint32_t total = n + count;                       // <= 63
        // right-aligned accumulator; the oldest 32 bits of acc:bits by one funnel shift
        const uint32_t left_aligned = bits << ((32u - count) & 31u);
        const uint32_t word = GPUAR_ALIGNBIT(acc, left_aligned, n);
#if defined(__HIP_DEVICE_COMPILE__)
        acc = (acc << (count & 31u)) | bits;                    // right if nothing leaves
        {
            // the predicated region by hand: lanes with a full dword store it; NO branch around it (some lane
            // has a full dword on almost every call, and a branch costs this wavefront more than the region)
            const unsigned long long full = __builtin_amdgcn_ballot_w64(total >= 32u);
            unsigned long long saved;
            uint32_t t_addr, t_word, t_mask;
            asm volatile(
                "s_and_saveexec_b64 %[sx], %[m]\n\t"
                "v_min_u32 %[ta], %[at], %[last]\n\t"
                "v_perm_b32 %[tw], 0, %[word], %[sel]\n\t"
                "global_store_dword %[ta], %[tw], %[base]\n\t"
                "v_add_u32 %[at], 4, %[at]\n\t"
                "v_bfm_b32 %[tm], %[total], 0\n\t"
                "v_and_b32 %[acc], %[bits], %[tm]\n\t"
                "s_or_b64 exec, exec, %[sx]"
                : [at] "+v"(at), [acc] "+v"(acc), [ta] "=&v"(t_addr), [tw] "=&v"(t_word), [tm] "=&v"(t_mask), [sx] "=&s"(saved)
                : [m] "s"(full), [last] "v"(last), [word] "v"(word), [sel] "s"(0x00010203u), [base] "s"(base), [total] "v"(total), [bits] "v"(bits)
                : "scc", "memory");       // (s_and_saveexec_b64 / s_or_b64 write scc; the region stores)
        }
#else
        acc = (acc << (count & 31u)) | bits;                    // right if nothing leaves
        if (total >= 32u) {
            store32(base + (at < last ? at : last), bswap32(word));
            at += 4u;
            acc = bits & GPUAR_BFM(total, 0u);                  // the total-32 youngest bits stay, all from `bits`
        }
#endif
        n = total & 31u;
    }

    // `first` (one bit), then `count` copies of its complement; count is
    // unbounded in principle.  One put() call site, deliberately not unrolled:
    // this is the rare path.
    GPUAR_LANE void put_bit_then_run(uint32_t first, uint32_t count) {
        const uint32_t ones = first - 1u;                       // complement of `first`, replicated
        uint32_t remaining = count + 1u;
        uint32_t lead = first;                                  // the very first bit emitted
#pragma clang loop unroll(disable) vectorize(disable)
        while (remaining) {
            const uint32_t c = remaining < 32u ? remaining : 32u;
            const uint32_t mask = c == 32u ? 0xFFFFFFFFu : ((1u << c) - 1u);
            // chunk = [lead or complement] followed by c-1 complements
            const uint32_t chunk = ((ones & mask) & ~(1u << (c - 1u))) | (((remaining == count + 1u) ? lead : (first ^ 1u)) << (c - 1u));
            put(chunk, c);
            remaining -= c;
        }
    }

    GPUAR_LANE void step(uint32_t cums, Recip rc) {
        const uint32_t up = div_total(GPUAR_MUL24(cums >> 16, range), rc);
        const uint32_t dn = div_total(GPUAR_MUL24(cums & 0xFFFFu, range), rc);
        const uint32_t a = lo + dn;                           // new lo
        const uint32_t wd = up - dn;                          // new hi - new lo + 1
        const uint32_t h = a + wd - 1u;                       // new hi (<= 0xFFFF: the interval only ever shrinks)
        // e agreeing MSBs leave, then a run of u underflow positions (closed form, see above bswap32);
        // neither argument of the two counts can be 0 (ones are shifted in behind the bits that matter)
        uint32_t span;                                        // ~a | h
        const uint32_t e = GPUAR_AGREE_COUNT(kff, a, h, span);
        const uint32_t u = GPUAR_CLZ32_NZ(GPUAR_ALIGNBIT(span, 0xFFFFFFFFu, 15u - e));
        const uint32_t shift = e + u;
        lo = (a << shift) & 0x7FFFu;
        range = wd << shift;
        // The e agreed bits (same in lo and hi) leave as: their MSB, then `pending`
        // copies of its complement, then the rest.  Inserting p complement bits
        // behind the MSB of an e-bit number A is the number A + (2^p - 1) * 2^(e-1)
        // (MSB 1: the carry pushes it up p places; MSB 0: p ones appear below
        // it), so the usual case pending <= 16 is one add of a bit-field mask.
        const uint32_t agreed = a >> (16u - e);                // 0 when e == 0 (a < 2^16)
        const uint32_t em1 = e - 1u;
        const bool shift_out = e != 0u;
        uint32_t bits = agreed + GPUAR_BFM(pending, em1);
        uint32_t count = e + pending;
        const bool long_run = shift_out & (pending > 16u);    // rare: more than 16 underflow bits owed (one mask, one branch)
        if (long_run) {
            const uint32_t top = (agreed >> (em1 & 31u)) & 1u;
            put_bit_then_run(top, pending);
            bits = agreed & GPUAR_BFM(em1, 0u);
            count = em1;
        }
        put(shift_out ? bits : 0u, shift_out ? count : 0u);
        pending = (shift_out ? 0u : pending) + u;
    }
    GPUAR_LANE uint32_t finish(uint32_t ulen, bool &overflowed) {
        uint8_t *body = base + body_off;
        const uint32_t bit14 = (lo >> 14) & 1u;               // bit 14 of lo (writeRemaining :379-388)
        put_bit_then_run(bit14, pending + 1u);
        // zero-pad to a byte boundary and store the tail (writeClose :430-439)
        const uint32_t pos = at - body_off;                   // bytes stored (or attempted) after the header
        const uint32_t tail_bytes = (n + 7u) >> 3;
        const uint32_t word = n ? (acc << (32u - n)) : 0u;
        uint32_t clen = pos + tail_bytes + kHdr;
        overflowed = clen > kSlot;
        if (overflowed) {
            clen = kSlot;
        } else {
            for (uint32_t k = 0; k < tail_bytes; ++k) body[pos + k] = static_cast<uint8_t>(word >> (24u - 8u * k));
        }
        const uint32_t hdr = clen | (ulen << 16);             // u16 LE clen, u16 LE ulen (:525-528)
        memcpy(body - kHdr, &hdr, 4);
        return clen;
    }
};

// ---------------------------------------------------------------------------
// The same coder in CARRY form (round 4; the throughput kernel's coder role).
//
// writeEncodedBits (:321-367) keeps lower and upper as 16-bit registers and never lets a carry out of them: while the
// interval straddles the midpoint (lower = 01.., upper = 10..) it takes 0x4000 off both, doubles them and OWES one bit
// ("underflow"), to be emitted -- as the complement of the next agreed bit -- once the bounds agree again.  CoderLane
// above does exactly that in closed form (agreed bits with `pending` complements spliced in behind their first one), and
// that splice is most of its 43 instructions.  Here the lower bound is kept as what it really is -- the low end of ONE
// long binary fraction --: a 64-bit window w whose bits [0, 16) are the live lower bound, bits [16, 16 + held) the `held`
// newest output bits still in the register, and the bit above them a carry out of them.  A symbol then costs
//     w += dn  (a carry, if any, runs through the held bits by itself),   w <<= n,   held += n
// -- n = e + u, ONE count of leading zeros (renorm_count) -- and the 32 oldest held bits leave as a dword when held >= 32.
//
// Why the bytes are the same.  Taking 0x4000 off a 16-bit register and doubling it (mod 2^16) is doubling it and
// flipping bit 15, so between symbols the reference's lower = w[15:0] ^ K with K = 0x8000 while bits are owed, 0
// otherwise; the widths agree (range = upper - lower + 1 is the same number in both forms), hence dn, up, and -- K
// cancels in a ^ h -- renorm_count.  The first underflow shifts a true 0 out of w, every further one a true 1 (bit 15 of
// w is the flipped 0 of lower): p owed bits stand in the stream as 0 1..1 (p - 1 ones) plus the set bit 15.  When the
// bounds next agree on b, the reference emits b and p complements: b = 0 is 0 1..1 1 -- what w shifts out anyway; b = 1
// means lower crossed 0x8000, i.e. w[15:0] crossed 2^16: the carry turns 0 1..1 into 1 0..0 and a 0 follows.  The end of
// the packet (writeRemaining :379-388: bit 14 of lower, then pending + 1 complements) is  w += 0x4000  and two more bits.
// n <= 16 (range' = width << n <= 2^16), held < 32 between symbols, so 16 + held + n <= 64: the window never overflows.
//
// A run of owed bits does not care where a dword ends: its leading 0 may have left the window while its ones are still
// in it (6 % of the dwords of a uniform stream), so a dword that leaves is not final yet.  It is kept back one store
// (`cache`) and takes the carry found above the NEXT dword when that one leaves -- cache + carry, one add in front of
// the store; a dword can be carried into once at most (after the carry everything below it is zero, and the interval is
// far too narrow to reach the next multiple).  A leaving dword of 32 ones cannot be kept back that way -- a carry would
// run through it -- so it is only counted (`nff`) and written, as ones or as zeros, when the next decided dword leaves:
// the classic carry-counting range coder at dword grain.  That is the rare path (one leaving dword in 2^32 on random
// data; the packet that owes 2396 bits, tests/golden/adversarial_midpoint, counts 74 of them).  The very first `cache`
// is a placeholder that lands on the packet's header dword, which finish() overwrites.
// Pinned against the oracle on the CPU like every other lane program (tests/test_lane_emulation.py).
// ---------------------------------------------------------------------------
// Cache policy bits of the coder's dword stores (A/B builds -DGPUAR_STORE_POLICY_ID=1, 2, 3: " nt", " sc1", " sc0 sc1"; profiles/r05_encoder_attribution.txt section 3:
// none of them brings the L2's write-backs nearer to the bytes written, the default is kept)
#if !defined(GPUAR_STORE_POLICY_ID) || GPUAR_STORE_POLICY_ID == 0
#define GPUAR_STORE_POLICY ""
#elif GPUAR_STORE_POLICY_ID == 1
#define GPUAR_STORE_POLICY " nt"
#elif GPUAR_STORE_POLICY_ID == 2
#define GPUAR_STORE_POLICY " sc1"
#else
#define GPUAR_STORE_POLICY " sc0 sc1"
#endif
struct CarryCoderLane {
    uint32_t wl, wh;     // the window w = wh:wl
    uint32_t range;      // hi - lo + 1  (2^14 < range <= 2^16 between symbols)
    uint32_t held;       // output bits in the window (< 32 between symbols)
    uint32_t cache;      // the dword that left the window last: written (plus a carry) when the next one leaves
    uint32_t nff;        // dwords of 32 ones that left behind `cache` and wait with it
    uint32_t at;         // offset from `base` where `cache` will be written
    uint32_t last;       // offset of the last dword of the slot: stores beyond it land there
    uint32_t kff;        // low half 0xFFFF, high half scratch of the renormalisation count
    uint8_t *base;       // same pointer in every lane of a wavefront (scalar register on the GPU)
    uint32_t body_off;   // this lane's byte offset of slot + 4

    GPUAR_LANE void open(uint8_t *uniform_base, uint32_t slot_offset) {
        wl = 0;           // lo = 0, hi = 0xFFFF  (:492-494)
        wh = 0;
        range = 0x10000u;
#if defined(__HIP_DEVICE_COMPILE__)
        held = 16u;       // the GPU's step keeps this count 16 too high (its shifts then take it as it is); finish() undoes that
#else
        held = 0;
#endif
        cache = 0;        // the placeholder: its "store" lands on the header dword
        nff = 0;
        kff = 0xFFFFu;
        base = uniform_base;
        body_off = slot_offset + kHdr;
        at = slot_offset;
        last = slot_offset + kSlot - 4u;
    }

    GPUAR_LANE void store_clamped(uint32_t word) {           // a packet that outgrows its slot keeps overwriting the slot's last dword
        store32(base + (at < last ? at : last), bswap32(word));
        at += 4u;
    }

    // the 32 oldest held bits leave the window (held >= 32); returns them and, through `over`, the carry found above them
    GPUAR_LANE uint32_t take_top(uint32_t &over) {
        const uint32_t s = held - 16u;                               // 16 <= s < 32: the dword is bits [s, s + 32) of w
        const uint32_t word = GPUAR_ALIGNBIT(wh, wl, s);
        over = wh >> s;
        wl &= GPUAR_BFM(s, 0u);
        wh = 0;
        held -= 32u;
        return word;
    }

    // a dword has left the window with `over` above it: the complete rule (the GPU's common path is its first line)
    GPUAR_LANE void leave(uint32_t word, uint32_t over) {
        if (word != 0xFFFFFFFFu || over) {
            store_clamped(cache + over);
            for (; nff; --nff) store_clamped(over ? 0u : 0xFFFFFFFFu);
            cache = word;
        } else {
            ++nff;
        }
    }

    // w <<= n, held += n, and the dword on top of the held bits leaves once there are 32 (the GPU's version of
    // take_top() + leave(); shared by the throughput kernel's coder and the latency kernel's sink)
#if defined(__HIP_DEVICE_COMPILE__)
    // `full`: the lanes with 32 bits held once this symbol's n are in (settled_mask(): formed well before this is called)
    GPUAR_LANE void shift_and_store(uint32_t n, unsigned long long full) {
        // The store region, predicated by hand, NO branch around it (some lane has a dword leaving on almost every symbol):
        // the common line of leave() -- cache + carry goes to memory, the leaving dword becomes the cache.  Lanes for which
        // that is not the whole story (a leaving dword of 32 ones, or such dwords waiting) are named in `rare` and put
        // right behind the region, from what the region left: `sent` (what was stored), `word`, `over`.
        // (Every shift in the region takes `held` itself as its count: the hardware looks at the low five bits only, and
        // held - 16 = held + 16 (mod 32) -- so the device keeps `held` 16 too high, see open().)
        // A lane mask that a vector instruction has just written is not yet there for the scalar unit: s_and_saveexec right
        // behind its v_cmp stalls the wavefront (the decoder's stream window lost 12 cycles per symbol that way,
        // profiles/r04_decode_cost_attribution.txt).  So the compare comes FIRST, the window's 64-bit shift behind it, the
        // region last; and the compare that finds the rare lanes sits inside the region, ten instructions before the
        // scalar test of its result.  One compare finds both kinds: `key` is 0xFFFFFFFF (word >= key: 32 ones) or, while
        // dwords wait, 0 (always).
        // The rare path is part of the same statement, behind a scalar branch: written in C++ behind the region, every
        // symbol paid two register copies for it (what it changes -- nff, key -- came back in other registers, and the
        // common path was the one that got the copies).  Lanes of `rare`:
        //     32 ones and no carry: not decided yet -- the dword waits (++nff) and so does the cache again (at -= 4)
        //     otherwise: decided -- the waiting dwords go out behind what was just stored (zeros behind a carry, else ones)
        //     key = nff ? 0 : ~0
        unsigned long long saved, rare, undecided;
        uint32_t over, sent, t_addr, t_swapped;
        {
            const uint64_t w = ((static_cast<uint64_t>(wh) << 32) | wl) << n;
            wl = static_cast<uint32_t>(w), wh = static_cast<uint32_t>(w >> 32);
        }
        asm volatile(
            "s_and_saveexec_b64 %[sx], %[m]\n\t"
            "v_lshrrev_b32 %[over], %[held], %[wh]\n\t"
            "v_add_u32 %[sent], %[cache], %[over]\n\t"                   /* what goes to memory: the dword kept back + the carry found now */
            "v_alignbit_b32 %[cache], %[wh], %[wl], %[held]\n\t"         /* the leaving dword is the one kept back from here on */
            "v_cmp_ge_u32 %[rare], %[cache], %[key]\n\t"                 /* (bits of lanes that do not store stay 0) */
            "v_min_u32 %[ta], %[at], %[last]\n\t"
            "v_perm_b32 %[tw], 0, %[sent], %[sel]\n\t"
            "global_store_dword %[ta], %[tw], %[base]" GPUAR_STORE_POLICY "\n\t"
            "v_bfe_u32 %[wl], %[wl], 0, %[held]\n\t"                     /* the window keeps what is below the dword that left */
            "v_mov_b32 %[wh], 0\n\t"
            "v_add_u32 %[at], 4, %[at]\n\t"
            "v_add_u32 %[held], -32, %[held]\n\t"
            "s_or_b64 exec, exec, %[sx]\n\t"
#ifndef GPUAR_CARRY_NO_RARE        // (timing experiments only: without the rare path the kernel is WRONG for dwords of 32 ones)
            "s_cmp_eq_u64 %[rare], 0\n\t"
            "s_cbranch_scc1 .Lgpuar_common_%=\n\t"
            "s_mov_b64 %[sx], exec\n\t"
            "s_mov_b64 exec, %[rare]\n\t"
            "v_cmp_eq_u32 vcc, -1, %[cache]\n\t"
            "v_cmp_eq_u32 %[und], 0, %[over]\n\t"
            "v_cndmask_b32 %[tw], 0, -1, %[und]\n\t"                    /* what waiting dwords turn into: ones, zeros behind a carry */
            "s_and_b64 %[und], %[und], vcc\n\t"
            "s_and_b64 exec, %[rare], %[und]\n\t"                       /* -- undecided */
            "v_add_u32 %[at], -4, %[at]\n\t"
            "v_mov_b32 %[cache], %[sent]\n\t"
            "v_add_u32 %[nff], 1, %[nff]\n\t"
            "s_andn2_b64 exec, %[rare], %[und]\n\t"                     /* -- decided */
            ".Lgpuar_fill_%=:\n\t"
            "v_cmp_ne_u32 vcc, 0, %[nff]\n\t"
            "s_and_b64 exec, exec, vcc\n\t"
            "s_cbranch_execz .Lgpuar_filled_%=\n\t"
            "v_min_u32 %[ta], %[at], %[last]\n\t"
            "global_store_dword %[ta], %[tw], %[base]" GPUAR_STORE_POLICY "\n\t"
            "v_add_u32 %[at], 4, %[at]\n\t"
            "v_add_u32 %[nff], -1, %[nff]\n\t"
            "s_branch .Lgpuar_fill_%=\n\t"
            ".Lgpuar_filled_%=:\n\t"
            "s_mov_b64 exec, %[rare]\n\t"
            "v_cmp_eq_u32 vcc, 0, %[nff]\n\t"
            "v_cndmask_b32 %[key], 0, -1, vcc\n\t"
            "s_mov_b64 exec, %[sx]\n\t"
            ".Lgpuar_common_%=:"
#endif
            : [wl] "+v"(wl), [wh] "+v"(wh), [at] "+v"(at), [held] "+v"(held), [cache] "+v"(cache), [nff] "+v"(nff), [key] "+v"(key),
              [over] "=&v"(over), [sent] "=&v"(sent), [ta] "=&v"(t_addr), [tw] "=&v"(t_swapped),
              [sx] "=&s"(saved), [rare] "=&s"(rare), [und] "=&s"(undecided)
            : [m] "s"(full), [last] "v"(last), [sel] "s"(0x00010203u), [base] "s"(base)
            : "memory", "vcc", "scc");
    }
#endif

#if defined(__HIP_DEVICE_COMPILE__)
    // held += n, and which lanes then hold a whole dword -- asked HERE, several instructions before the store region's
    // s_and_saveexec reads the answer (see shift_and_store)
    // (`wd`: the new width; range' = wd << n is formed here as well, behind the compare: one more instruction between the
    // compare and the region)
    GPUAR_LANE unsigned long long settled_mask(uint32_t n, uint32_t wd) {
        unsigned long long full;
        asm volatile("v_add_u32 %[held], %[held], %[n]\n\t"
                     "v_cmp_le_u32 %[m], 48, %[held]\n\t"
                     "v_lshlrev_b32 %[rng], %[n], %[wd]"
                     : [held] "+v"(held), [m] "=&s"(full), [rng] "=v"(range)
                     : [n] "v"(n), [wd] "v"(wd));
        return full;
    }
#endif

    // The step in three pieces, so that a caller with the next symbol's sums at hand (encode_kernel's unrolled phase) can put
    // that symbol's two divisions between this symbol's compare and its store region:
    //     ahead(cums, rc)  the two bounds of the symbol's interval (needs `range` as the symbol before it left it)
    //     narrow(a)        w += dn, the renormalisation count, range', held -- and the mask of the lanes that will store
    //     settle(r)        w <<= n, the store region, the rare path
    // step() is the three in a row.
    struct Ahead {
        uint32_t dn, wd;
    };
    struct Narrowed {
        uint32_t n;
        unsigned long long full;
    };
    GPUAR_LANE Ahead ahead(uint32_t cums, Recip rc) const {
        // A volatile statement on the GPU: it stays where the caller puts it -- between the "who stores?" compare of the symbol
        // before and the store region that reads the compare's lane mask (seven more instructions for the scalar unit to wait
        // behind; left to the compiler the divisions end up behind that region).  -0.5 %.  (While the coder still fetched its
        // reciprocals by scalar loads this very statement cost 24 %, 18.4 -> 23.0 ms: it pulled the wait for the scalar load --
        // which can only be lgkmcnt(0) -- in front of the phase's LDS reads.  The reciprocals now arrive through LDS.)
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t up, dn;
        asm volatile("v_mul_u32_u24_sdwa %[up], %[c], %[r] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
                     "v_mul_u32_u24_sdwa %[dn], %[c], %[r] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"
                     "v_mul_hi_u32 %[up], %[up], %[m]\n\t"
                     "v_mul_hi_u32 %[dn], %[dn], %[m]\n\t"
                     "v_lshrrev_b32 %[up], %[s], %[up]\n\t"
                     "v_lshrrev_b32 %[dn], %[s], %[dn]\n\t"
                     "v_sub_u32 %[up], %[up], %[dn]"
                     : [up] "=&v"(up), [dn] "=&v"(dn)
                     : [c] "v"(cums), [r] "v"(range), [m] "v"(rc.mul), [s] "v"(rc.shift));
        return {dn, up};
#else
        const uint32_t up = div_total(GPUAR_MUL24(cums >> 16, range), rc);
        const uint32_t dn = div_total(GPUAR_MUL24(cums & 0xFFFFu, range), rc);
        return {dn, up - dn};                                 // up - dn = new hi - new lo + 1
#endif
    }
    GPUAR_LANE Narrowed narrow(Ahead a) {
        const uint32_t dn = a.dn, wd = a.wd;
#if defined(__HIP_DEVICE_COMPILE__)
        // w += dn; n = renorm_count(w[15:0], wd) -- one statement: the xor lands in HALF a register (SDWA) and may be read
        // by the second instruction behind it at the earliest, so the order is fixed here
        uint32_t h, t2, c, t, n;
        asm("v_add_co_u32 %[wl], vcc, %[wl], %[dn]\n\t"
            "v_addc_co_u32 %[wh], vcc, 0, %[wh], vcc\n\t"
            "v_add3_u32 %[h], %[wl], %[wd], -1\n\t"                   /* new hi, low 16 bits */
            "v_lshl_add_u32 %[t2], %[wd], 16, %[km]\n\t"              /* (2 * width - 1) << 15 */
            "v_xor_b32_sdwa %[kff], %[wl], %[h] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n\t"
            "v_ffbh_u32 %[c], %[t2]\n\t"
            "v_lshlrev_b32 %[t], %[c], %[kff]\n\t"
            "v_lshrrev_b32 %[t], 31, %[t]\n\t"
            "v_add3_u32 %[n], %[c], %[t], -1"
            : [wl] "+v"(wl), [wh] "+v"(wh), [kff] "+v"(kff), [h] "=&v"(h), [t2] "=&v"(t2), [c] "=&v"(c), [t] "=&v"(t), [n] "=&v"(n)
            : [dn] "v"(dn), [wd] "v"(wd), [km] "s"(0xFFFF8000u)
            : "vcc");
        const unsigned long long full = settled_mask(n, wd);
        return {n, full};
#else
        const uint64_t sum = ((static_cast<uint64_t>(wh) << 32) | wl) + dn;
        wl = static_cast<uint32_t>(sum), wh = static_cast<uint32_t>(sum >> 32);
        const uint32_t n = renorm_count(wl & 0xFFFFu, wd);
        range = wd << n;
        held += n;
        return {n, held >= 32u ? 1ull : 0ull};
#endif
    }
    GPUAR_LANE void settle(Narrowed r) {
#if defined(__HIP_DEVICE_COMPILE__)
        shift_and_store(r.n, r.full);
#else
        const uint64_t w = ((static_cast<uint64_t>(wh) << 32) | wl) << r.n;     // 16 + held <= 64 (+ the carry bit above the held ones)
        wl = static_cast<uint32_t>(w), wh = static_cast<uint32_t>(w >> 32);
        if (r.full) {
            uint32_t over;
            const uint32_t word = take_top(over);
            leave(word, over);
        }
#endif
    }

    GPUAR_LANE void step(uint32_t cums, Recip rc) { settle(narrow(ahead(cums, rc))); }
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t key = 0xFFFFFFFFu;       // what a leaving dword is compared with to find the rare cases: 0 while nff != 0
#endif

    // writeRemaining (:379-388) + writeClose (:430-439) in carry form (derivation in the header comment): + 0x4000, two
    // more bits, zero padding to a byte.  Once per packet: plain code.
    GPUAR_LANE uint32_t finish(uint32_t ulen, bool &overflowed) {
#if defined(__HIP_DEVICE_COMPILE__)
        held -= 16u;
#endif
        const uint64_t w = ((((static_cast<uint64_t>(wh) << 32) | wl) + 0x4000u) << 2);
        wl = static_cast<uint32_t>(w), wh = static_cast<uint32_t>(w >> 32);
        held += 2u;                                           // < 34
        if (held >= 32u) {
            uint32_t over;
            const uint32_t word = take_top(over);
            leave(word, over);
        }
        // what is left: `held` (< 32) bits at [16, 16 + held), perhaps a carry above them: everything is decided now
        const uint32_t top = GPUAR_ALIGNBIT(wh, wl, 16u);     // bits [16, 48) of w
        const uint32_t bits = held ? (top & GPUAR_BFM(held, 0u)) : 0u;
        const uint32_t over = held ? (top >> held) : top;     // (held < 32: the carry, if any, is inside `top`)
        store_clamped(cache + over);
        for (; nff; --nff) store_clamped(over ? 0u : 0xFFFFFFFFu);
        uint8_t *body = base + body_off;
        const uint32_t pos = at - body_off;                   // bytes stored (or attempted) after the header
        const uint32_t tail_bytes = (held + 7u) >> 3;
        const uint32_t word = held ? (bits << (32u - held)) : 0u;
        uint32_t clen = pos + tail_bytes + kHdr;
        overflowed = clen > kSlot;
        if (overflowed) {
            clen = kSlot;
        } else {
            for (uint32_t k = 0; k < tail_bytes; ++k) body[pos + k] = static_cast<uint8_t>(word >> (24u - 8u * k));
        }
        const uint32_t hdr = clen | (ulen << 16);             // u16 LE clen, u16 LE ulen (:525-528)
        memcpy(body - kHdr, &hdr, 4);
        return clen;
    }
};

// ---------------------------------------------------------------------------
// The carry-form coder cut in two, for the LATENCY-mode encoder (encode_small_kernel; round 4: the owed-bits form's cut,
// IntervalLane | SinkLane below, made the sink the longest role at ~34 instructions).  CarryIntervalLane owns the interval
// -- applySymbolRange (:256-299) and the renormalisation count; of the lower bound it needs the live 16 bits only --,
// CarrySinkLane owns the window, the held bits and the stores.  They are joined by ONE word per symbol: dn | n << 16
// (what is added to the window, and by how much it then moves); nothing else, not even at the end of the packet
// (the sink's window holds the true lower bound).  Same integers as CarryCoderLane::step.
// ---------------------------------------------------------------------------
struct CarryIntervalLane {
    uint32_t lo, range, kff;       // lo: the live lower bound in its low 16 bits (what is above them is ignored)
    GPUAR_LANE void open() {
        lo = 0;
        range = 0x10000u;
        kff = 0xFFFFu;
    }
    GPUAR_LANE uint32_t step(uint32_t cums, Recip rc) {
        const uint32_t up = div_total(GPUAR_MUL24(cums >> 16, range), rc);
        const uint32_t dn = div_total(GPUAR_MUL24(cums & 0xFFFFu, range), rc);
        const uint32_t wd = up - dn;
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t a, h, t2, c, t, n;
        asm("v_add_u32 %[a], %[lo], %[dn]\n\t"
            "v_add3_u32 %[h], %[a], %[wd], -1\n\t"
            "v_lshl_add_u32 %[t2], %[wd], 16, %[km]\n\t"
            "v_xor_b32_sdwa %[kff], %[a], %[h] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n\t"
            "v_ffbh_u32 %[c], %[t2]\n\t"
            "v_lshlrev_b32 %[t], %[c], %[kff]\n\t"
            "v_lshrrev_b32 %[t], 31, %[t]\n\t"
            "v_add3_u32 %[n], %[c], %[t], -1\n\t"
            "v_lshlrev_b32_sdwa %[lo], %[n], %[a] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"   /* (a & 0xFFFF) << n */
            : [lo] "+v"(lo), [kff] "+v"(kff), [a] "=&v"(a), [h] "=&v"(h), [t2] "=&v"(t2), [c] "=&v"(c), [t] "=&v"(t), [n] "=&v"(n)
            : [dn] "v"(dn), [wd] "v"(wd), [km] "s"(0xFFFF8000u));
#else
        const uint32_t a = (lo + dn) & 0xFFFFu;
        const uint32_t n = renorm_count(a, wd);
        lo = a << n;
#endif
        range = wd << n;
        return dn | (n << 16);
    }
};

struct CarrySinkLane : CarryCoderLane {
    GPUAR_LANE void take(uint32_t packed) {
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t n;
        asm("v_add_co_u32_sdwa %[wl], vcc, %[wl], %[p] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
            "v_addc_co_u32 %[wh], vcc, 0, %[wh], vcc\n\t"
            "v_lshrrev_b32 %[n], 16, %[p]"
            : [wl] "+v"(wl), [wh] "+v"(wh), [n] "=&v"(n)
            : [p] "v"(packed)
            : "vcc");
        const unsigned long long full = settled_mask(n, 0u);     // (the sink has no range of its own: the shift of 0 is dead weight of one instruction)
        shift_and_store(n, full);
#else
        const uint32_t n = packed >> 16;
        const uint64_t w = ((((static_cast<uint64_t>(wh) << 32) | wl) + (packed & 0xFFFFu)) << n);
        wl = static_cast<uint32_t>(w), wh = static_cast<uint32_t>(w >> 32);
        held += n;
        if (held >= 32u) {
            uint32_t over;
            const uint32_t word = take_top(over);
            leave(word, over);
        }
#endif
    }
};

// ===========================================================================
// DECODER (arDecompress :848-892).  The symbol search
// (getSymbolFromProbability :727-763) touches LDS in two round trips of one
// 16-byte read each instead of walking eight levels.  Depths 0 and 1 of the
// left-count tree live in registers; depths 2..4 and 5..7 are stored as
// 3-level subtrees, each one RECORD of two 8-byte halves (the seven nodes and the
// count under the root's right child; exact order at decide3) so a single
// ds_read2_b64 fetches everything the next three decisions can need, those
// decisions are then taken in registers, and ONE 8-byte store puts back the half
// the path went through.  (Measured with tools/lds_probe.hip: at these occupancies
// a u16 LDS read costs the CU about as much as a 16-byte one.)  Same left-count
// tree as the encoder's, same counts, same sums, same symbols.
//   records  0..3   : subtrees rooted at the depth-2 nodes (index = complemented top 2 symbol bits)
//   records  4..35  : subtrees rooted at the depth-5 nodes (index = complemented top 5 symbol bits)
//   half 2r + h of record r sits at col + ((2r + h) << kHalfShift): its index is the path so far, one bit longer
// kHalfShift = log2(bytes between consecutive halves of one lane): 9 on the
// GPU (64 lanes x 8 B, lane-minor: the lanes of a 64-bit access cover all banks
// whatever halves they address), 3 on the host.
//
// THE WALK WORKS ON A SCALED REMAINDER.  The reference forms
//     unscaled = (((code - lower) + 1) * total - 1) / range          (getUnscaledCode :703-716)
// and looks for the symbol s with cum(s) <= unscaled < cum(s+1).  With
// R0 = (code - lower + 1) * total - 1 that is  cum(s)*range <= R0 < cum(s+1)*range
// (all integers), so the quotient is never needed: keep R = R0 - below*range,
// where `below` counts the symbols left of the current subtree; at a node whose
// left subtree holds `a` symbols the walk goes left iff R < a*range, and going
// right subtracts a*range.  One decision = one 24-bit multiply, one
// subtract-with-borrow (the borrow IS "went left") and one unsigned minimum
// (R < a*range leaves R, otherwise R - a*range is the smaller of the two) --
// no lane-mask select on the critical chain.  At the leaf R0 - R = cumLo*range
// is the very numerator applySymbolRange (:256-299) divides by total, and
// cnt*range for the upper bound comes from S and the three products of the low
// record: width(left child) = a, width(right child) = width(parent) - a;
// S = a + aR, the two counts at the head of a record's halves.
// ===========================================================================
template <uint32_t kHalfShift>
struct SubtreeModel {
    // What three decisions inside one record leave to be written back: the HALF of the record the first
    // decision chose, with +1 on its count (a or aR: every symbol that lands in the record lands in one of
    // the two halves), on the child if the second decision went left and on the grandchild if the third
    // did -- ONE 8-byte store at an address that is one shift-add of the path, no select.
    // (The GPU's hand-scheduled step, gpuar_kernels.hip, applies the same three increments to the half in place with
    // one 64-bit LDS add instead of rebuilding and storing it: no field can carry into its neighbour.)
    struct Path {
        uint32_t at;                    // byte offset of the half from `col`
        uint32_t w0, w1;
    };

    uint8_t *col;                       // this lane's 8-byte column
    uint32_t root, half0, half1;        // depth 0; depth 1 under root's left / right child
    Path owed;                          // write-back of the previous symbol's low record, not yet issued

    static constexpr uint32_t kHalves = 2u * kDecodeRecords;
    static constexpr uint32_t kHalf = 1u << kHalfShift;        // bytes between the two halves of a record
    static constexpr uint32_t kLowBase = 8u << kHalfShift;     // byte offset of low record 0 (behind the 4 mid records)

    GPUAR_LANE void reset() {
        root = 128u;
        half0 = half1 = 64u;
        // nothing owed yet: a write-back that rewrites the last half with its initial values
        owed.at = (kHalves - 1u) << kHalfShift;
        owed.w0 = 4u | (2u << 16);
        owed.w1 = 1u | (1u << 16);
#pragma unroll 1
        for (uint32_t h = 0; h < kHalves; ++h) {
            const uint32_t top = h < 8u ? 32u : 4u;       // count under a child of a depth-2 / depth-5 node
            store64(col + (h << kHalfShift), top | ((top >> 1) << 16), (top >> 2) * 0x10001u);
        }
    }

    // one decision on the scaled remainder: went left iff R < prod; going right takes prod off R
    static GPUAR_LANE bool decide(uint32_t &R, uint32_t prod) {
        const uint32_t d = R - prod;
        const bool left = R < prod;
        R = d < R ? d : R;                                 // == left ? R : d  (prod >= 1)
        return left;
    }

    // Three decisions inside the record whose halves are `right` and `left`.  `npath` = the
    // COMPLEMENTED symbol bits decided so far, MSB first: every decision is kept
    // as "went LEFT" -- that is what the borrow of the subtraction says, what the
    // node update adds, and npath = 2 * npath + left is one add-with-carry of it;
    // records, halves and the nodes inside them are simply stored in complemented
    // order so that npath indexes them directly.  A record is two halves of 8 bytes,
    // one per child of its root (L/R = left/right child of the node before):
    //     right half:  aR | bR << 16,  cRR | cRL << 16        (half 2r of the lane's column)
    //     left half:   a  | bL << 16,  cLR | cLL << 16        (half 2r + 1)
    // a = symbols counted in the root's LEFT subtree (the node of the tree proper), aR = in its right one;
    // b, c: the left-counts of the child and of the grandchildren on that side.  The first decision needs a,
    // the next two only the half it chose -- and only that half changes.
    // `base`: byte offset of the record's right half.  kLow: the record is a low one -- also returns
    // W = cnt(symbol) * range through `width`.
    template <bool kLow>
    GPUAR_LANE Path decide3(uint32_t base, const Pair &right, const Pair &left, uint32_t range, uint32_t &R, uint32_t &npath,
                            uint32_t &width) {
        const uint32_t pa = GPUAR_MUL24_VV(left.w[0] & 0xFFFFu, range);
        const bool la = decide(R, pa);
        const uint32_t bw = la ? left.w[0] : right.w[0];      // the chosen child sits in its high half
        const uint32_t cc = la ? left.w[1] : right.w[1];      // both grandchildren under the chosen child
        const uint32_t pb = GPUAR_MUL24_VV(bw >> 16, range);
        const bool lb = decide(R, pb);
        const uint32_t pc = GPUAR_MUL24_VV(lb ? cc >> 16 : cc & 0xFFFFu, range);
        const bool lc = decide(R, pc);
        npath = 8u * npath + (la ? 4u : 0u) + (lb ? 2u : 0u) + (lc ? 1u : 0u);
        if (kLow) {
            const uint32_t ps = GPUAR_MUL24_VV((left.w[0] & 0xFFFFu) + (right.w[0] & 0xFFFFu), range);   // all eight symbols
            const uint32_t gw = la ? pa : ps - pa;             // width of the chosen child, scaled
            const uint32_t pw = lb ? pb : gw - pb;
            width = lc ? pc : pw - pc;
        }
        Path p;
        p.at = base + (la ? kHalf : 0u);
        p.w0 = bw + (lb ? 0x10001u : 1u);
        p.w1 = cc + (lc ? (lb ? 0x10000u : 1u) : 0u);
        return p;
    }
    GPUAR_LANE void write_back(const Path &p) { store64(col + p.at, p.w0, p.w1); }

    // The symbol s with cum(s)*range <= R0 < cum(s+1)*range.  On return R = R0 - cum(s)*range
    // and width = cnt(s)*range.  Memory-safe for any R0 (a value beyond the model's total simply
    // walks right).
    // Order of LDS traffic (LDS operations of a wavefront complete in order):
    //   write back the PREVIOUS symbol's low half -> read mid record -> read low record (a later
    //   read of the same low record comes after that write) -> write back the mid half ->
    //   (the low half's write-back is owed to the next call / flush()).
    // `in_shadow()` is called right after the first record read has been issued:
    // work that the symbol search does not depend on goes there.
    template <typename Shadow>
    GPUAR_LANE uint32_t decode_step(uint32_t R0, uint32_t range, uint32_t &R, uint32_t &width, Shadow &&in_shadow) {
        R = R0;
        const bool l0 = decide(R, GPUAR_MUL24_VV(root, range));
        const uint32_t h = l0 ? half0 : half1;
        const bool l1 = decide(R, GPUAR_MUL24_VV(h, range));
        uint32_t npath = (l0 ? 2u : 0u) + (l1 ? 1u : 0u);     // complemented top two symbol bits
        write_back(owed);
        const uint32_t rec_mid = (2u * npath) << kHalfShift;
        Pair mid_r = load64(col + rec_mid), mid_l = load64(col + rec_mid + kHalf);   // ds_read2_b64 #1
        GPUAR_PIN_LOAD(mid_l);
        root += l0 ? 1u : 0u;                                 // register nodes: in the shadow of read #1
        const uint32_t h_new = h + (l1 ? 1u : 0u);            // the depth-1 node on the path
        half0 = l0 ? h_new : half0;
        half1 = l0 ? half1 : h_new;
        in_shadow();
        uint32_t unused = 0;
        const Path p_mid = decide3<false>(rec_mid, mid_r, mid_l, range, R, npath, unused);
        const uint32_t rec_low = kLowBase + ((2u * npath) << kHalfShift);   // npath = complemented top five symbol bits
        Pair low_r = load64(col + rec_low), low_l = load64(col + rec_low + kHalf);   // ds_read2_b64 #2 ...
        GPUAR_PIN_LOAD(low_l);
        write_back(p_mid);                                    // ... with the mid half's write-back behind it
        owed = decide3<true>(rec_low, low_r, low_l, range, R, npath, width);
        return npath ^ 255u;
    }

    // issue the write-back still owed (call once after the last symbol; harmless if repeated)
    GPUAR_LANE void flush() { write_back(owed); }
};

// Per-symbol constants of the decoder, wave-uniform, one 16-byte scalar load: the reciprocal of
// the model total (Recip above), the total itself and total - 1.
struct DecodeConst {
    uint32_t mul, shift, total, total_m1;
};
struct DecodeConstTable {
    DecodeConst c[kPacket];
    constexpr DecodeConstTable() : c{} {
        const RecipTable r = RecipTable();
        for (uint32_t i = 0; i < kPacket; ++i) {
            c[i].mul = r.r[i].mul;
            c[i].shift = r.r[i].shift;
            c[i].total = 256u + i;
            c[i].total_m1 = 255u + i;
        }
    }
};

// Decoder state of one packet: SubtreeModel, the interval as its lower bound
// and its WIDTH (lo, range = hi - lo + 1), the code value as its OFFSET above
// lo, and a bit reader.
//
// Why the offset: every renormalisation step of readEncodedBits (:787-836)
// subtracts the same constant (0, 0x8000 or 0x4000) from lo, hi and code and
// then doubles them, pulling one stream bit into code.  code - lo therefore
// just doubles and takes the bit, and hi - lo + 1 just doubles: after
// n = e + u steps
//     off' = (off << n) | next n stream bits,        range' = width << n,
// one 64-bit shift of off:window, with none of the masks and the conditional
// complement the absolute code value needs; getUnscaledCode's numerator
// ((code - lower) + 1) * total - 1 (:703-716) is off * total + (total - 1).
// Only e and u themselves are read off the bit patterns of the new bounds
// a = lo' and h = hi' (closed form above bswap32): e = agreeing MSBs of a and h,
// u = the run below them where a has 1 and h has 0, and lo'' = (a << n) & 0x7FFF.
//
// Bit reader: two aligned big-endian dwords (w0:w1) hold the stream at the
// current position, `rem` (0..31) bits of w0 are still unread, a third dword
// is in flight; alignbit(w0, w1, rem) is the next 32 stream bits.
template <uint32_t kRowShift>
struct DecoderLane {
    SubtreeModel<kRowShift> model;
    uint32_t w0, w1;           // two consecutive stream dwords, big-endian order restored
    uint32_t ahead;            // the dword after w1, still as loaded (swapped only when it moves up,
                               // so the wait for its load lands a whole dword of bits later)
    uint32_t rem;              // unread bits of w0 (0..31); 0: the window starts at w1
    uint32_t owed_bits;        // bits consumed by the previous symbol, not yet skipped (done in the
                               // shadow of the next symbol's first record read)
    const uint8_t *base;       // start of the bytes this wavefront reads (the same in every lane: a scalar on the GPU)
    uint32_t next;             // offset from base of the dword after `ahead` (base + next is 4-byte aligned)
    uint32_t last;             // offset of the last dword that still holds a readable byte
    uint32_t lo;               // lower bound of the interval (< 2^15 between symbols)
    uint32_t range;            // hi - lo + 1  (2^14 < range <= 2^16 between symbols)
    uint32_t off;              // code - lo
    uint32_t ulen;
    uint32_t outword;
    bool bad;

    // An aligned dword that holds at least one readable byte never crosses a
    // page, so it is loaded whole; past the last such dword the reader simply
    // keeps re-reading it (a well-formed packet decodes the same whatever
    // follows it), which costs one v_min instead of a compare and a branch.
    GPUAR_LANE uint32_t fetch() {
        const uint32_t w = load32(base + (next < last ? next : last));
        next += 4;
        return w;
    }

    // the next 32 stream bits, left-aligned
    GPUAR_LANE uint32_t peek() const { return GPUAR_ALIGNBIT(w0, w1, rem); }
    GPUAR_LANE void skip(uint32_t count) {      // count <= 32
        const bool refill = rem < count;
        rem = (rem - count) & 31u;
        if (refill) {
            w0 = w1;
            w1 = bswap32(ahead);
            // consume the OLD prefetched dword before the new load is issued:
            // otherwise the wait for the old one (vmcnt) also waits for the new one
            GPUAR_PIN_ORDER(w1);
            ahead = fetch();
        }
    }

    // The packet starts `pkt_off` bytes after `uniform_base`; bytes up to `limit_off` (exclusive, > pkt_off
    // for a live lane) may be read.  A dead lane (live == false) reads the dword at uniform_base.
    GPUAR_LANE void open(uint8_t *col, const uint8_t *uniform_base, uint32_t pkt_off, uint32_t limit_off, bool live) {
        model.col = col;
        model.reset();
        base = uniform_base;
        ulen = 0;
        bad = false;
        outword = 0;
        uint32_t body = 0;
        last = 0;
        if (live) {
            const uint8_t *pkt = base + pkt_off;
            const uint32_t clen = pkt[0] | (static_cast<uint32_t>(pkt[1]) << 8);
            ulen = pkt[2] | (static_cast<uint32_t>(pkt[3]) << 8);
            if (ulen > kPacket || clen < kHdr) {   // the reference would run off its buffers here
                bad = true;
                ulen = 0;
            }
            body = pkt_off + kHdr;
            // offset of the dword holding byte limit_off - 1, counted from an aligned address
            const uint32_t skew = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(base) & 3u);
            last = ((limit_off - 1u + skew) & ~3u) - skew;
        }
        const uint32_t misalign = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(base + body) & 3u);
        next = body - misalign;
        w0 = bswap32(fetch());
        w1 = bswap32(fetch());
        ahead = fetch();
        // the first 16 bits of the body are the initial code value (initializeDecoder :582-603)
        const uint32_t used = 8u * misalign + 16u;            // 16, 24, 32 or 40 bits of w0:w1 are behind us
        const uint64_t both = (static_cast<uint64_t>(w0) << 32) | w1;
        lo = 0;                                               // lo = 0, hi = 0xFFFF
        range = 0x10000u;
        off = static_cast<uint32_t>(both >> (64u - used)) & 0xFFFFu;
        rem = (32u - used) & 31u;                             // used == 32: w0 is spent, the window starts at w1
        if (used > 32u) {
            w0 = w1;
            w1 = bswap32(ahead);
            ahead = fetch();
        }
        owed_bits = 0;
    }

    // applySymbolRange (:256-299) and the renormalisation (:787-836) on (lo, range, off), given the
    // two numerators cumLo*range and cumHi*range.  Shared by the plain and the hand-scheduled step.
    GPUAR_LANE void narrow(uint32_t num_lo, uint32_t num_hi, const DecodeConst &k) {
        const Recip rc = {k.mul, k.shift};
        const uint32_t dn = div_total(num_lo, rc);
        const uint32_t up = div_total(num_hi, rc);
        const uint32_t a = lo + dn;                           // new lo
        const uint32_t wd = up - dn;                          // new hi - new lo + 1
        // e agreeing MSBs leave, then a run of u underflow positions; only their sum matters here (see above bswap32)
        const uint32_t n = renorm_count(a, wd);               // <= 31 fresh bits
        lo = (a << n) & 0x7FFFu;
        range = wd << n;
        off = static_cast<uint32_t>((((static_cast<uint64_t>(off - dn) << 32) | peek()) << n) >> 32);
        owed_bits = n;
    }

    // decodes symbol i and returns it; the caller places it (see put_symbol / flush)
    GPUAR_LANE uint32_t step_symbol(const DecodeConst &k) {
        const uint32_t R0 = GPUAR_MUL24(off, k.total) + k.total_m1;   // getUnscaledCode's numerator (:703-716)
        // No symbol owns a code value with floor(R0 / range) >= total, i.e. off >= range (:873-877,
        // where the reference stops decoding the packet).  It cannot happen -- off < range is an invariant of
        // the step for any bit stream (argument in gpuar_kernels.hip, decode_wave) -- so the GPU's hand-scheduled
        // step does not look; this compiled path keeps the comparison as a guard of its own arithmetic.
        bad = bad || off >= range;
        uint32_t R, width;
        const uint32_t sym = model.decode_step(R0, range, R, width, [this]() {
            skip(owed_bits);          // the stream window is next needed at the end of this symbol
        });
        const uint32_t num_lo = R0 - R;                       // cumLo * range
        narrow(num_lo, num_lo + width, k);
        return sym;
    }

    // simple placement: one dword store per four symbols
    GPUAR_LANE void step(uint32_t i, const DecodeConst &k, uint8_t *out) {
        const uint32_t sym = step_symbol(k);
        outword |= sym << (8u * (i & 3u));
        if ((i & 3u) == 3u) {
            store32(out + (i & ~3u), outword);
            outword = 0;
        }
    }

    GPUAR_LANE void finish(uint8_t *out) {
        model.flush();
        for (uint32_t b = ulen & ~3u; b < ulen; ++b) out[b] = static_cast<uint8_t>(outword >> (8u * (b & 3u)));
    }
};

}  // namespace gpuar
#endif  // GPUAR_LANE_CODEC_H
