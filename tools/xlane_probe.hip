// What it costs ONE wavefront alone on a gfx950 SIMD to let the lanes of a packet talk to each other -- the
// primitives a "several lanes per packet" latency-mode decoder would be made of, next to the
// three-instruction binary decision the lane-per-packet decoder is made of.  Exact inline-asm streams; the shader
// clock is read inside the kernel.  One workgroup of 64 threads on an otherwise idle chip.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/xlane_probe.bin tools/xlane_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

#define CLOBBER "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "m0", "memory"
#define OPS "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(r) : "v"(lds)

// one round of a rotate-and-min all-reduce inside a row of 16 lanes: x = min(x, x rotated by K)
#define ROR_MIN(X, K) "v_min_u32_dpp " X ", " X ", " X " row_ror:" #K " row_mask:0xf bank_mask:0xf\n"
#define ROR_MAX(X, K) "v_max_u32_dpp " X ", " X ", " X " row_ror:" #K " row_mask:0xf bank_mask:0xf\n"
// gfx9: a VALU write of a VGPR needs two wait states before a DPP instruction reads it
#define NOP2 "s_nop 1\n"

template <int K>
__global__ void __launch_bounds__(64) probe(unsigned long long *out, int iters) {
    __shared__ uint32_t buf[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) buf[i] = i * 2654435761u;
    __syncthreads();
    uint32_t a = threadIdx.x * 2654435761u + 1000, b = threadIdx.x * 3 + 7, c = 7, d = 9, e = 11, f = 13, g = 17, h = 19, r = 40000;
    uint32_t lds = threadIdx.x * 4;
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        // ---- the yardsticks ----
        if (K == 0) asm volatile(REP64("v_add_u32 %0, %0, %2\n") : OPS : CLOBBER);                                   // dependent VALU
        if (K == 1) asm volatile(REP64("v_mul_u32_u24 %4, %2, %3\n v_sub_co_u32 %5, s[20:21], %8, %4\n v_min_u32 %8, %8, %5\n") : OPS : CLOBBER);   // one binary decision
        // ---- DPP ----
        if (K == 10) asm volatile(REP64(ROR_MIN("%0", 8) NOP2) : OPS : CLOBBER);                                     // dependent DPP chain, padded as the hardware asks
        if (K == 11) asm volatile(REP64(ROR_MIN("%0", 8) ROR_MIN("%1", 8) ROR_MAX("%3", 8)) : OPS : CLOBBER);        // three independent chains interleaved: no padding
        if (K == 12) asm volatile(REP16(ROR_MIN("%0", 8) NOP2 ROR_MIN("%0", 4) NOP2 ROR_MIN("%0", 2) NOP2 ROR_MIN("%0", 1) NOP2) : OPS : CLOBBER);   // one 16-lane all-reduce
        if (K == 13) asm volatile(REP16(ROR_MIN("%0", 8) ROR_MIN("%1", 8) ROR_MAX("%3", 8) ROR_MIN("%0", 4) ROR_MIN("%1", 4) ROR_MAX("%3", 4)
                                        ROR_MIN("%0", 2) ROR_MIN("%1", 2) ROR_MAX("%3", 2) ROR_MIN("%0", 1) ROR_MIN("%1", 1) ROR_MAX("%3", 1)) : OPS : CLOBBER);   // three all-reduces interleaved
        if (K == 14) asm volatile(REP16(ROR_MIN("%0", 8) ROR_MAX("%3", 8) "s_nop 0\n" ROR_MIN("%0", 4) ROR_MAX("%3", 4) "s_nop 0\n"
                                        ROR_MIN("%0", 2) ROR_MAX("%3", 2) "s_nop 0\n" ROR_MIN("%0", 1) ROR_MAX("%3", 1) "s_nop 0\n") : OPS : CLOBBER);            // two interleaved, one pad each
        // 8-lane groups: quad_perm xor 1, xor 2, then row_half_mirror
        if (K == 15) asm volatile(REP16("v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" NOP2
                                        "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n" NOP2
                                        "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n" NOP2) : OPS : CLOBBER);
        // ---- the LDS crossbar without memory: permute / swizzle (dependent round trips) ----
        if (K == 20) asm volatile(REP16("ds_bpermute_b32 %0, %9, %0\n s_waitcnt lgkmcnt(0)\n") : OPS : CLOBBER);
        if (K == 21) asm volatile(REP16("ds_swizzle_b32 %0, %0 offset:swizzle(BROADCAST,16,3)\n s_waitcnt lgkmcnt(0)\n") : OPS : CLOBBER);
        // ---- through LDS memory: the owning lane writes, everybody reads it back (a data-dependent broadcast) ----
        if (K == 22) asm volatile(REP16("ds_write_b32 %9, %0 offset:8192\n ds_read_b32 %0, %9 offset:8192\n s_waitcnt lgkmcnt(0)\n") : OPS : CLOBBER);
        if (K == 23) asm volatile(REP16("ds_read_b32 %0, %9\n s_waitcnt lgkmcnt(0)\n v_and_b32 %9, 0xfc, %0\n") : OPS : CLOBBER);   // plain dependent LDS read
        // ---- through the scalar unit ----
        // compare -> lane mask -> count of set bits -> that lane's value -> back into a vector register (ONE group per wavefront)
        if (K == 30) asm volatile(REP16("v_cmp_lt_u32_e64 s[20:21], %0, %1\n s_bcnt1_i32_b64 s22, s[20:21]\n v_readlane_b32 s23, %2, s22\n v_mov_b32 %0, s23\n") : OPS : CLOBBER);
        // the same for FOUR groups of 16 lanes: a field extract and a count per group, four readlanes, four writelanes' worth of selects
        if (K == 31) asm volatile(REP16("v_cmp_lt_u32_e64 s[20:21], %0, %1\n"
                                        "s_bfe_u32 s22, s20, 0x100000\n s_bcnt1_i32_b32 s22, s22\n v_readlane_b32 s24, %2, s22\n"
                                        "s_bfe_u32 s22, s20, 0x100010\n s_bcnt1_i32_b32 s22, s22\n s_add_i32 s22, s22, 16\n v_readlane_b32 s25, %2, s22\n"
                                        "s_bfe_u32 s22, s21, 0x100000\n s_bcnt1_i32_b32 s22, s22\n s_add_i32 s22, s22, 32\n v_readlane_b32 s26, %2, s22\n"
                                        "s_bfe_u32 s22, s21, 0x100010\n s_bcnt1_i32_b32 s22, s22\n s_add_i32 s22, s22, 48\n v_readlane_b32 s27, %2, s22\n"
                                        "v_mov_b32 %0, s24\n v_mov_b32 %3, s25\n v_cndmask_b32_e64 %0, %0, %3, s[20:21]\n v_mov_b32 %3, s26\n v_cndmask_b32_e64 %0, %0, %3, s[20:21]\n v_mov_b32 %3, s27\n v_cndmask_b32_e64 %0, %0, %3, s[20:21]\n") : OPS : CLOBBER);
        // the lane mask counted per lane in vector registers: bits below me (mbcnt), then row broadcasts of lane 15 and lane 0
        if (K == 32) asm volatile(REP16("v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_mbcnt_lo_u32_b32 %3, s20, 0\n v_mbcnt_hi_u32_b32 %3, s21, %3\n" NOP2
                                        "v_mov_b32_dpp %4, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
                                        "v_sub_u32_dpp %0, %3, %4 row_newbcast:15 row_mask:0xf bank_mask:0xf\n") : OPS : CLOBBER);
        // a v_readlane whose scalar result feeds the next vector instruction (what the decoder did per symbol until round 4)
        if (K == 33) asm volatile(REP16("v_readlane_b32 s22, %2, 5\n v_mul_hi_u32 %0, %0, s22\n") : OPS : CLOBBER);
        if (K == 34) asm volatile(REP16("v_mul_hi_u32 %0, %0, %2\n") : OPS : CLOBBER);
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (a + b + c + d + e + f + g + h + r == 0x12345u) out[1] = a;
}

struct Row {
    int k;
    const char *name;
    double units;      // what one REP block holds (cycles are divided by iters * units)
};

template <int K>
static double run(unsigned long long *d_out, int iters) {
    probe<K><<<1, 64>>>(d_out, 10);
    (void)hipDeviceSynchronize();
    probe<K><<<1, 64>>>(d_out, iters);
    (void)hipDeviceSynchronize();
    unsigned long long t = 0;
    (void)hipMemcpy(&t, d_out, sizeof t, hipMemcpyDeviceToHost);
    return static_cast<double>(t);
}

int main() {
    unsigned long long *d_out;
    (void)hipMalloc(&d_out, 16);
    const int iters = 2000;
    // clock64() counts at 100 MHz on this part; the shader clock under one wavefront is ~2.1-2.4 GHz: calibrate on the yardstick
    const double base = run<0>(d_out, iters) / (iters * 64.0);
    printf("# cycles are in units of the dependent v_add_u32 (= 1.00; a lone wavefront issues one per ~4.6 shader cycles)\n");
#define ROW(K, UNITS, NAME) printf("%-110s %7.2f\n", NAME, run<K>(d_out, iters) / (iters * (UNITS)) / base);
    ROW(0, 64.0, "dependent v_add_u32 (the yardstick)")
    ROW(1, 64.0, "one binary decision of the lane-per-packet walk (mul, sub -> borrow, min), per decision")
    ROW(10, 64.0, "v_min_u32_dpp row_ror, dependent, with the two wait states the hardware asks for, per step")
    ROW(11, 64.0 * 3, "three independent DPP chains interleaved (no padding needed), per DPP instruction")
    ROW(12, 16.0, "all-reduce over 16 lanes (4 rotate-and-min steps, padded), per all-reduce")
    ROW(13, 16.0, "THREE all-reduces over 16 lanes interleaved (12 DPP instructions), per group of three")
    ROW(14, 16.0, "TWO all-reduces over 16 lanes interleaved, one pad per round, per pair")
    ROW(15, 16.0, "all-reduce over 8 lanes (quad_perm, quad_perm, row_half_mirror; padded), per all-reduce")
    ROW(20, 16.0, "ds_bpermute_b32 + wait (data-dependent gather across the wavefront), per round trip")
    ROW(21, 16.0, "ds_swizzle_b32 broadcast in 16 + wait, per round trip")
    ROW(22, 16.0, "LDS write + read back + wait (broadcast through memory), per round trip")
    ROW(23, 16.0, "dependent ds_read_b32 + wait + address, per round trip")
    ROW(30, 16.0, "compare -> mask -> s_bcnt1 -> v_readlane -> v_mov, ONE group per wavefront, per search level")
    ROW(31, 16.0, "the same for FOUR groups of 16 lanes (4 x extract/count/readlane + selects), per search level")
    ROW(32, 16.0, "compare -> mask -> v_mbcnt x2 -> two row broadcasts (index of the hit in every lane), per search level")
    ROW(33, 16.0, "v_readlane -> scalar register -> v_mul_hi_u32 (the decoder's multiplier fetch until round 4), per pair")
    ROW(34, 16.0, "v_mul_hi_u32 with a vector multiplier (since round 4), per instruction")
    return 0;
}
