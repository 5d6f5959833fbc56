// Prototype of the SYMBOL SEARCH of a latency-mode decoder with eight lanes per packet ("a one-day prototype of
// the symbol search alone"; the cut profiles/r04_latency_decoder_study.txt found best on paper: 16 x 16 cumulative counts, two
// entries per lane, one wavefront per SIMD = 8 packets).  The exact per-symbol instruction stream such a decoder's search and
// model update would issue -- real multiplies, borrows, DPP all-reduces over 8 lanes, the dependent LDS row read and the row's
// write-back -- on synthetic counts, timed by the kernel's own clock, one wavefront per SIMD on the whole chip (1024 workgroups
// with 40 KiB of LDS each), next to the lane-per-packet search's cost taken from the real kernel (a build of decode_slots_kernel
// WITHOUT its search, -DGPUAR_EXP_NO_SEARCH).  Nothing here decodes anything: it prices the stream.
//
// Model held per packet (8 lanes, lane j of the group):
//   level 1: P1a, P1b = number of symbols in buckets below 2j, 2j + 1 (16 buckets of 16 symbols), in registers;
//   level 2: 16 rows of 16 cumulative counts (u16) in LDS, lane j owns entries 2j, 2j + 1 of every row (one dword).
// One symbol:
//   L1  thresholds P1 * range against the scaled remainder R0: two multiplies, two subtractions with borrow; the remainder
//       below the hit = min of the differences (a borrowed one wraps to a huge number), the hit's index = 16 - borrows, the
//       distance up to the next threshold = min of the reversed differences; three all-reduces over the 8 lanes, interleaved
//       (quad_perm, quad_perm, row_half_mirror); every prefix above the hit takes +1 by its own borrow (two add-with-carry);
//   row address from the index, DEPENDENT ds_read_b32 of the lane's two entries of that row; in its shadow the stand-in for the
//       part of the step that can sit there (the stream window's selects: 4 instructions), and the previous row's write-back;
//   L2  the same on the row; write-back of the row (+1 above the hit: add-with-carry, select, add, ds_write_b32);
//   symbol = 16 * index1 + index2; cnt * range = remainder + min(the two distances).
// Variants: K = 0 the stream as described; K = 1 without the upper-distance reduce (a decoder that finds cumHi some cheaper way);
// K = 2 the LDS read not waited for (what the search costs when the row read were free); K = 3 the yardstick (64 dependent
// v_add_u32, 4.63 cycles each on a lone wavefront).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/latsearch_probe.bin tools/latsearch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

#define DPP3(A, B, C, PERM)                                                        \
    "v_min_u32_dpp " A ", " A ", " A " " PERM " row_mask:0xf bank_mask:0xf\n"      \
    "v_add_u32_dpp " B ", " B ", " B " " PERM " row_mask:0xf bank_mask:0xf\n"      \
    "v_min_u32_dpp " C ", " C ", " C " " PERM " row_mask:0xf bank_mask:0xf\n"
#define DPP2(A, B, PERM)                                                           \
    "v_min_u32_dpp " A ", " A ", " A " " PERM " row_mask:0xf bank_mask:0xf\n"      \
    "v_add_u32_dpp " B ", " B ", " B " " PERM " row_mask:0xf bank_mask:0xf\n"      \
    "s_nop 0\n" /* two chains: one wait state short of what a DPP read of a fresh result needs */
#define REDUCE3(A, B, C) DPP3(A, B, C, "quad_perm:[1,0,3,2]") DPP3(A, B, C, "quad_perm:[2,3,0,1]") DPP3(A, B, C, "row_half_mirror")
#define REDUCE2(A, B) DPP2(A, B, "quad_perm:[1,0,3,2]") DPP2(A, B, "quad_perm:[2,3,0,1]") DPP2(A, B, "row_half_mirror")

// operands: %0 R0 (scaled remainder; rewritten at the end so that the next symbol depends on this one), %1 rng, %2 P1a, %3 P1b,
// %4 ta, %5 tb, %6 d (remainder), %7 cnt, %8 u (distance up), %9 addr, %10 row, %11 t, %12 sym, %13 rowbase, %14 k64k, %15 zero,
// %16 w0, %17 w1 (stand-ins for the stream window's registers), %18 prevaddr, %19 prevrow
#define LEVEL_COMPARE(PA, PB, R, SDWA_A, SDWA_B, WITH_UPPER)                                                    \
    "v_mul_u32_u24" SDWA_A "\n"                                                                                   \
    "v_mul_u32_u24" SDWA_B "\n"                                                                                   \
    "v_sub_co_u32 %6, s[20:21], " R ", %4\n"                                                                       \
    "v_sub_co_u32 %11, s[22:23], " R ", %5\n"                                                                      \
    "v_min_u32 %6, %6, %11\n"                                                                                      \
    "v_addc_co_u32 %7, vcc, %15, %15, s[20:21]\n"                                                                  \
    "v_addc_co_u32 %7, vcc, %7, %15, s[22:23]\n"                                                                   \
    WITH_UPPER
#define UPPER(R)                                                                                                  \
    "v_sub_u32 %8, %4, " R "\n"                                                                                    \
    "v_sub_u32 %11, %5, " R "\n"                                                                                   \
    "v_min_u32 %8, %8, %11\n"

#define SEARCH(WITH_UPPER1, WITH_UPPER2, REDUCE_1, REDUCE_2, WAIT, FINAL_UPPER)                                  \
    /* ---- level 1 ---- */                                                                                        \
    LEVEL_COMPARE("%2", "%3", "%0", " %4, %2, %1", " %5, %3, %1", WITH_UPPER1)                                     \
    REDUCE_1                                                                                                       \
    "v_addc_co_u32 %2, vcc, %2, %15, s[20:21]\n" /* every prefix above the hit: + 1 */                             \
    "v_addc_co_u32 %3, vcc, %3, %15, s[22:23]\n"                                                                   \
    "v_lshlrev_b32 %11, 5, %7\n"                                                                                   \
    "v_sub_u32 %9, %13, %11\n" /* the row of the hit: rowbase - 32 * borrows */                                    \
    "v_and_b32 %9, 0x7fc, %9\n" /* (the probe's counts are synthetic: keep the address inside the rows) */         \
    "ds_read_b32 %10, %9\n"                                                                                        \
    /* in the shadow: the previous symbol's row goes back, and what else of the step can sit here */               \
    "ds_write_b32 %18, %19\n"                                                                                      \
    "v_cndmask_b32_e64 %16, %16, %17, s[20:21]\n"                                                                  \
    "v_cndmask_b32_e64 %17, %17, %16, s[22:23]\n"                                                                  \
    "v_alignbit_b32 %12, %16, %17, %7\n"                                                                           \
    "v_perm_b32 %17, %15, %17, %16\n"                                                                              \
    WAIT                                                                                                           \
    /* ---- level 2 on the remainder ---- */                                                                       \
    LEVEL_COMPARE("%10", "%10", "%6",                                                                              \
                  "_sdwa %4, %10, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD",          \
                  "_sdwa %5, %10, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD", WITH_UPPER2) \
    "v_lshlrev_b32 %12, 4, %7\n" /* (index 1 before the second reduce overwrites the count) */                     \
    REDUCE_2                                                                                                       \
    "v_addc_co_u32 %19, vcc, %10, %15, s[20:21]\n" /* the row, + 1 above the hit */                                \
    "v_cndmask_b32_e64 %11, %15, %14, s[22:23]\n"                                                                  \
    "v_add_u32 %19, %19, %11\n"                                                                                    \
    "v_mov_b32 %18, %9\n"                                                                                          \
    "v_add_u32 %12, %12, %7\n" /* the symbol's (complemented) index */                                             \
    FINAL_UPPER                                                                                                    \
    /* the next symbol's remainder depends on this one's results (stand-in for R0 = off * total + total - 1) */    \
    "v_add_u32 %0, %6, %12\n"                                                                                      \
    "v_and_b32 %0, 0xffffff, %0\n"

#define OPS                                                                                                        \
    "+v"(R0), "+v"(rng), "+v"(p1a), "+v"(p1b), "+v"(ta), "+v"(tb), "+v"(d), "+v"(cnt), "+v"(u), "+v"(addr), "+v"(row), "+v"(t), "+v"(sym) \
    : "v"(rowbase), "v"(k64k), "v"(zero), "v"(w0), "v"(w1), "v"(prevaddr), "v"(prevrow)
#define CLOBBER "vcc", "s20", "s21", "s22", "s23", "memory"

template <int K>
__global__ void __launch_bounds__(64) probe(unsigned long long *out, int iters) {
    __shared__ uint4 buf[64 * 40];          // 40 KiB: at most four of these workgroups per CU, one per SIMD
    for (int i = threadIdx.x; i < 64 * 40; i += 64) buf[i] = make_uint4(i & 0x3fff, (i * 7) & 0x3fff, i & 0xfff, 3);
    __syncthreads();
    const uint32_t lane = threadIdx.x;
    uint32_t R0 = 123456 + lane * 977, rng = 40000 + lane, p1a = 16 * (2 * (lane & 7)), p1b = 16 * (2 * (lane & 7) + 1);
    uint32_t ta = 0, tb = 0, d = 0, cnt = 0, u = 0, addr = 0, row = 0, t = 0, sym = 0;
    uint32_t rowbase = 16 * 32 + 4 * (lane & 7) + 512 * (lane >> 3), k64k = 0x10000, zero = 0, w0 = lane, w1 = lane * 3, prevaddr = 4 * lane, prevrow = 0;
    asm volatile("" : "+v"(zero), "+v"(k64k), "+v"(rowbase), "+v"(w0), "+v"(w1), "+v"(prevaddr), "+v"(prevrow));
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (K == 0) asm volatile(REP8(SEARCH(UPPER("%0"), UPPER("%6"), REDUCE3("%6", "%7", "%8"), REDUCE3("%6", "%7", "%8"), "s_waitcnt lgkmcnt(1)\n",
                                            "v_add_u32 %8, %8, %6\n")) : OPS : CLOBBER);
        if (K == 1) asm volatile(REP8(SEARCH("", "", "s_nop 0\n" REDUCE2("%6", "%7"), "s_nop 0\n" REDUCE2("%6", "%7"), "s_waitcnt lgkmcnt(1)\n", "")) : OPS : CLOBBER);
        if (K == 2) asm volatile(REP8(SEARCH(UPPER("%0"), UPPER("%6"), REDUCE3("%6", "%7", "%8"), REDUCE3("%6", "%7", "%8"), "", "v_add_u32 %8, %8, %6\n")) : OPS : CLOBBER);
        if (K == 3) asm volatile(REP64("v_add_u32 %0, %0, %1\n") : OPS : CLOBBER);
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (R0 + d + cnt + u + sym + row + p1a + p1b + addr + t + ta + tb == 0x12345u) out[1] = R0;
}

template <int K>
static double run(unsigned long long *d_out, int iters, double *ms_out) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    probe<K><<<1024, 64>>>(d_out, 10);
    (void)hipEventRecord(a);
    probe<K><<<1024, 64>>>(d_out, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    *ms_out = ms;
    unsigned long long ticks = 0;
    (void)hipMemcpy(&ticks, d_out, sizeof ticks, hipMemcpyDeviceToHost);
    return static_cast<double>(ticks);
}

int main() {
    unsigned long long *d_out;
    (void)hipMalloc(&d_out, 16);
    const int iters = 4000;
    double ms = 0;
    // clock64() ticks at 100 MHz on this part: the yardstick (a dependent v_add_u32 = 4.63 shader cycles on a lone wavefront,
    // tools/lat_probe.hip) converts ticks into shader cycles; the wall time of the launch cross-checks it
    const double yard = run<3>(d_out, iters, &ms) / (iters * 64.0);
    const double cycles_per_tick = 4.63 / yard;
    printf("# one wavefront per SIMD on the whole chip (1024 workgroups x 40 KiB of LDS); yardstick: %d x 64 dependent v_add_u32 in %.2f ms -> %.0f MHz if one takes 4.63 cycles\n",
           iters, ms, iters * 64.0 * 4.63 / (ms * 1e-3) / 1e6);
#define ROW(K, NAME)                                                                                              \
    {                                                                                                              \
        const double ticks = run<K>(d_out, iters, &ms);                                                            \
        printf("%-118s %7.1f cycles per symbol  (%.2f ms for %d symbols = %.1f cycles at 2.4 GHz)\n", NAME, ticks / (iters * 8.0) * cycles_per_tick, ms, \
               iters * 8, ms * 1e-3 * 2.4e9 / (iters * 8.0));                                                      \
    }
    ROW(0, "8 lanes per packet: search + model update, 16 x 16 cumulative counts, three reduces per level (remainder, index, distance up)")
    ROW(1, "... two reduces per level (no distance up: cumHi found some cheaper way), one wait state per round")
    ROW(2, "... as the first, the row read not waited for (garbage: what the dependent LDS read costs)")
    return 0;
}
