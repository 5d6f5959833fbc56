// Latency / issue model of ONE wavefront alone on a gfx950 SIMD, for the instruction patterns of the
// decoder's symbol step.  Every probe is an exact inline-asm stream; the shader clock (s_memtime) is
// read before and after inside the kernel, so launch overhead does not enter.  One workgroup of 64
// threads on an otherwise idle chip.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/lat_probe.bin tools/lat_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

#define CLOBBER "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "memory"
#define OPS "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(r) : "v"(lds)

struct Probe {
    const char *name;
    int per_rep;        // "units" per REP64 block (what the cycles are divided by)
};

template <int K>
__global__ void __launch_bounds__(64) probe(unsigned long long *out, int iters) {
    __shared__ uint4 buf[64 * 40];          // 40 KiB: at most four of these workgroups per CU, one per SIMD
    for (int i = threadIdx.x; i < 64 * 40; i += 64) buf[i] = make_uint4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    uint32_t a = threadIdx.x + 1000, b = threadIdx.x * 3 + 7, c = 7, d = 9, e = 11, f = 13, g = 17, h = 19, r = 40000;
    uint32_t lds = threadIdx.x * 16;
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        // ---- plain VALU chains ----
        if (K == 0) asm volatile(REP64("v_add_u32 %0, %0, %2\n") : OPS : CLOBBER);                                   // distance 1
        if (K == 1) asm volatile(REP64("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n") : OPS : CLOBBER);              // distance 2
        if (K == 2) asm volatile(REP64("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %3, %3, %2\n v_add_u32 %4, %4, %2\n") : OPS : CLOBBER);   // distance 4
        if (K == 3) asm volatile(REP64("v_mul_u32_u24 %0, %0, %2\n") : OPS : CLOBBER);
        if (K == 4) asm volatile(REP64("v_mul_u32_u24 %0, %0, %2\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %3, %3, %2\n v_mul_u32_u24 %4, %4, %2\n") : OPS : CLOBBER);
        if (K == 5) asm volatile(REP64("v_cndmask_b32_e64 %0, %0, %2, s[20:21]\n") : OPS : CLOBBER);
        if (K == 6) asm volatile(REP64("v_add_u32 %0, %0, %2\n v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %1, %1, %2\n") : OPS : CLOBBER);  // pairs
        // ---- decisions ----
        // new: prod = node * range (node independent), d = R - prod -> borrow, R = min(R, d)
        if (K == 10) asm volatile(REP64("v_mul_u32_u24 %4, %2, %3\n v_sub_co_u32 %5, s[20:21], %8, %4\n v_min_u32 %8, %8, %5\n") : OPS : CLOBBER);
        // new with the dependent select of the next node in the chain (node = mask ? x : y) and the path add-with-carry
        if (K == 11) asm volatile(REP64("v_mul_u32_u24 %4, %2, %3\n v_sub_co_u32 %5, s[20:21], %8, %4\n v_min_u32 %8, %8, %5\n v_addc_co_u32 %6, s[22:23], %6, %6, s[20:21]\n v_cndmask_b32_e64 %2, %0, %1, s[20:21]\n") : OPS : CLOBBER);
        // same, the select directly behind the subtraction (no wait state in between)
        if (K == 12) asm volatile(REP64("v_mul_u32_u24 %4, %2, %3\n v_sub_co_u32 %5, s[20:21], %8, %4\n v_cndmask_b32_e64 %2, %0, %1, s[20:21]\n v_min_u32 %8, %8, %5\n v_addc_co_u32 %6, s[22:23], %6, %6, s[20:21]\n") : OPS : CLOBBER);
        // old: s = below + node, prod = s * range, cmp -> mask, below = mask ? below : s, upper = mask ? s : upper, path addc; next node select
        if (K == 13) asm volatile(REP64("v_add_u32 %4, %8, %2\n v_mul_u32_u24 %5, %4, %3\n v_cmp_le_u32_e64 s[20:21], %7, %5\n s_nop 1\n v_cndmask_b32_e64 %8, %4, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %4, s[20:21]\n v_addc_co_u32 %0, s[22:23], %0, %0, s[20:21]\n v_cndmask_b32_e64 %2, %0, %1, s[20:21]\n") : OPS : CLOBBER);
        // ---- mask produce -> consume distance ----
        if (K == 20) asm volatile(REP64("v_cmp_lt_u32_e64 s[20:21], %0, %2\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n") : OPS : CLOBBER);
        if (K == 21) asm volatile(REP64("v_cmp_lt_u32_e64 s[20:21], %0, %2\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n") : OPS : CLOBBER);
        if (K == 22) asm volatile(REP64("v_cmp_lt_u32_e64 s[20:21], %0, %2\n v_add_u32 %3, %3, %2\n v_add_u32 %4, %4, %2\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n") : OPS : CLOBBER);
        if (K == 23) asm volatile(REP64("v_cmp_lt_u32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %1, vcc\n") : OPS : CLOBBER);
        if (K == 24) asm volatile(REP64("v_cmp_lt_u32 vcc, %0, %2\n v_add_u32 %3, %3, %2\n v_add_u32 %4, %4, %2\n v_cndmask_b32 %0, %0, %1, vcc\n") : OPS : CLOBBER);
        if (K == 25) asm volatile(REP64("v_sub_co_u32 %3, vcc, %0, %2\n v_add_u32 %4, %4, %2\n v_add_u32 %5, %5, %2\n v_cndmask_b32_sdwa %0, %1, %1, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n") : OPS : CLOBBER);
        if (K == 26) asm volatile(REP64("v_cndmask_b32 %0, %0, %1, vcc\n") : OPS : CLOBBER);       // VOP2 reading a stale vcc
        if (K == 27) asm volatile(REP64("v_cndmask_b32_e64 %0, %0, %1, vcc\n") : OPS : CLOBBER);   // VOP3 encoding, vcc as the mask
        // ---- fillers ----
        if (K == 30) asm volatile(REP64("v_add_u32 %0, %0, %2\n s_nop 0\n") : OPS : CLOBBER);
        if (K == 31) asm volatile(REP64("v_add_u32 %0, %0, %2\n s_nop 1\n") : OPS : CLOBBER);
        if (K == 32) asm volatile(REP64("v_add_u32 %0, %0, %2\n s_or_b64 s[24:25], s[24:25], s[26:27]\n") : OPS : CLOBBER);
        if (K == 33) asm volatile(REP64("v_add_u32 %0, %0, %2\n s_nop 0\n s_nop 0\n s_nop 0\n") : OPS : CLOBBER);
        // ---- SDWA ----
        if (K == 40) asm volatile(REP64("v_mul_u32_u24_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n") : OPS : CLOBBER);
        if (K == 41) asm volatile(REP64("v_and_b32 %3, 0xffff, %0\n v_mul_u32_u24 %0, %3, %2\n") : OPS : CLOBBER);
        // ---- LDS round trips ----
        if (K == 50) asm volatile(REP16("ds_read_b128 v[100:103], %9\n s_waitcnt lgkmcnt(0)\n v_and_b32 %9, 0x3f0, v100\n") : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 51) asm volatile(REP16("ds_read_b128 v[100:103], %9\n" REP16("v_add_u32 %0, %0, %2\n") "s_waitcnt lgkmcnt(0)\n v_and_b32 %9, 0x3f0, v100\n") : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 52) asm volatile(REP16("ds_read_b128 v[100:103], %9\n ds_write_b128 %9, v[104:107] offset:8192\n" REP16("v_add_u32 %0, %0, %2\n") "s_waitcnt lgkmcnt(1)\n v_and_b32 %9, 0x3f0, v100\n") : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 53) asm volatile(REP16("ds_write_b128 %9, v[104:107] offset:8192\n ds_read_b128 v[100:103], %9\n" REP16("v_add_u32 %0, %0, %2\n") "s_waitcnt lgkmcnt(0)\n v_and_b32 %9, 0x3f0, v100\n") : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 54) asm volatile(REP16("ds_read_b128 v[100:103], %9\n" REP16("v_add_u32 %0, %0, %2\n") REP16("v_add_u32 %1, %1, %2\n") "s_waitcnt lgkmcnt(0)\n v_and_b32 %9, 0x3f0, v100\n") : OPS : CLOBBER, "v100", "v101", "v102", "v103");
#define RW_ROUND(N_ADDS) "ds_read_b128 v[100:103], %9\n ds_write_b128 %9, v[104:107] offset:8192\n" N_ADDS "s_waitcnt lgkmcnt(1)\n v_and_b32 %9, 0x3f0, v100\n"
#define R_ROUND(N_ADDS) "ds_read_b128 v[100:103], %9\n" N_ADDS "s_waitcnt lgkmcnt(0)\n v_and_b32 %9, 0x3f0, v100\n"
#define ADD1 "v_add_u32 %0, %0, %2\n"
#define ADD8 REP4(ADD1) REP4(ADD1)
        if (K == 60) asm volatile(REP16(RW_ROUND(ADD8)) : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 61) asm volatile(REP16(RW_ROUND(ADD8 ADD8)) : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 62) asm volatile(REP16(RW_ROUND(ADD8 ADD8 ADD8)) : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 63) asm volatile(REP16(RW_ROUND(ADD8 ADD8 ADD8 ADD8)) : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 64) asm volatile(REP16(RW_ROUND(ADD8 ADD8 ADD8 ADD8 ADD8 ADD8)) : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 65) asm volatile(REP16(R_ROUND(ADD8)) : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 66) asm volatile(REP16(R_ROUND(ADD8 ADD8)) : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 67) asm volatile(REP16(R_ROUND(ADD8 ADD8 ADD8)) : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 68) asm volatile(REP16(R_ROUND(ADD8 ADD8 ADD8 ADD8)) : OPS : CLOBBER, "v100", "v101", "v102", "v103");
#define RW2(WR, N_ADDS) "ds_read2_b64 v[100:103], %9 offset1:64\n " WR " %9, v[104:105] offset:8192\n" N_ADDS "s_waitcnt lgkmcnt(1)\n v_and_b32 %9, 0x1f8, v100\n"
        if (K == 70) asm volatile(REP16(RW2("ds_write_b64", ADD8)) : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 71) asm volatile(REP16(RW2("ds_add_u64", ADD8)) : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 72) asm volatile(REP16(RW2("ds_write_b64", ADD8 ADD8)) : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 73) asm volatile(REP16(RW2("ds_add_u64", ADD8 ADD8)) : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 74) asm volatile(REP16(RW2("ds_write_b64", ADD8 ADD8 ADD8)) : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 75) asm volatile(REP16(RW2("ds_add_u64", ADD8 ADD8 ADD8)) : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 76) asm volatile(REP16("ds_read2_b64 v[100:103], %9 offset1:64\n" ADD8 ADD8 ADD8 "s_waitcnt lgkmcnt(0)\n v_and_b32 %9, 0x1f8, v100\n") : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 77) asm volatile(REP16("ds_read2_b64 v[100:103], %9 offset1:64\n ds_write_b64 %9, v[104:105] offset:8192\n ds_read_b32 v106, %9 offset:16384\n" ADD8 ADD8 ADD8 "s_waitcnt lgkmcnt(1)\n v_and_b32 %9, 0x1f8, v100\n") : OPS : CLOBBER, "v100", "v101", "v102", "v103", "v106");
        if (K == 55) asm volatile(REP16("ds_read_b32 v100, %9\n s_waitcnt lgkmcnt(0)\n v_and_b32 %9, 0x3f0, v100\n") : OPS : CLOBBER, "v100", "v101", "v102", "v103");
        if (K == 56) asm volatile(REP16("ds_read_b64 v[100:101], %9\n s_waitcnt lgkmcnt(0)\n v_and_b32 %9, 0x3f0, v100\n") : OPS : CLOBBER, "v100", "v101", "v102", "v103");
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (a + b + c + d + e + f + g + h + r + lds == 0x12345) out[1] = a;
}

template <int K>
void run(unsigned long long *d, const char *name, int units_per_iter) {
    const int iters = 200;
    probe<K><<<1, 64>>>(d, 10);
    probe<K><<<1, 64>>>(d, iters);
    unsigned long long cyc = 0;
    (void)hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
    printf("%-100s %8.2f cycles per unit\n", name, double(cyc) / (double(iters) * units_per_iter));
}

// the same stream on every SIMD of the chip at once (1024 workgroups, one wavefront per SIMD): cycles per
// unit as the wavefront's own clock counts them, and the clock rate that count implies against wall time
template <int K>
void run_full(unsigned long long *d, const char *name, int units_per_iter) {
    const int iters = 20000;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    probe<K><<<1024, 64>>>(d, 10);
    (void)hipEventRecord(a);
    probe<K><<<1024, 64>>>(d, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long cyc = 0;
    (void)hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
    printf("FULL CHIP %-90s %8.2f cycles per unit, %7.1f ms wall -> s_memtime ticks at %6.0f MHz\n", name,
           double(cyc) / (double(iters) * units_per_iter), ms, double(cyc) / (ms * 1e3));
}

int main() {
    unsigned long long *d;
    (void)hipMalloc(&d, 64);
    run<0>(d, "v_add_u32 chain, distance 1 (per instruction)", 64);
    run<1>(d, "v_add_u32 two chains interleaved, distance 2 (per instruction)", 128);
    run<2>(d, "v_add_u32 four chains, distance 4 (per instruction)", 256);
    run<6>(d, "v_add_u32 two chains in pairs aabb (per instruction)", 256);
    run<3>(d, "v_mul_u32_u24 chain, distance 1 (per instruction)", 64);
    run<4>(d, "v_mul_u32_u24 four chains, distance 4 (per instruction)", 256);
    run<5>(d, "v_cndmask_b32_e64 chain, distance 1 (per instruction)", 64);
    run<10>(d, "decision NEW: mul(indep), sub_co, min            (per decision, 3 instr)", 64);
    run<11>(d, "decision NEW + addc + select of next node, select 2 behind sub_co (per decision, 5 instr)", 64);
    run<12>(d, "decision NEW + addc + select of next node, select right behind sub_co (per decision, 5 instr)", 64);
    run<13>(d, "decision OLD: add, mul, cmp, s_nop 1, 3 cndmask, addc (per decision, 8 instr)", 64);
    run<20>(d, "v_cmp_e64 -> v_cndmask_e64, back to back (per pair)", 64);
    run<21>(d, "v_cmp_e64 -> s_nop 1 -> v_cndmask_e64 (per pair)", 64);
    run<22>(d, "v_cmp_e64 -> 2 independent adds -> v_cndmask_e64 (per group of 4)", 64);
    run<23>(d, "v_cmp vcc -> v_cndmask vcc (VOP2), back to back (per pair)", 64);
    run<24>(d, "v_cmp vcc -> 2 independent adds -> v_cndmask vcc (VOP2) (per group of 4)", 64);
    run<25>(d, "v_sub_co vcc -> 2 adds -> v_cndmask_b32_sdwa vcc (per group of 4)", 64);
    run<26>(d, "v_cndmask_b32 VOP2, vcc not written in the loop (per instruction)", 64);
    run<27>(d, "v_cndmask_b32_e64 with vcc as mask (per instruction)", 64);
    run<30>(d, "v_add_u32 chain + s_nop 0 (per pair)", 64);
    run<31>(d, "v_add_u32 chain + s_nop 1 (per pair)", 64);
    run<32>(d, "v_add_u32 chain + s_or_b64 (per pair)", 64);
    run<33>(d, "v_add_u32 chain + 3 x s_nop 0 (per group)", 64);
    run<40>(d, "v_mul_u32_u24_sdwa chain (per instruction)", 64);
    run<41>(d, "v_and + v_mul_u32_u24 chain (per pair)", 64);
    run<50>(d, "ds_read_b128 -> wait -> address from data (per round trip)", 16);
    run<51>(d, "ds_read_b128, 16 dependent adds in its shadow, wait (per round trip)", 16);
    run<52>(d, "ds_read_b128 + ds_write_b128 behind it, 16 adds, wait lgkmcnt(1) (per round trip)", 16);
    run<53>(d, "ds_write_b128 in FRONT of ds_read_b128, 16 adds, wait (per round trip)", 16);
    run<54>(d, "ds_read_b128, 32 adds in its shadow, wait (per round trip)", 16);
    run<55>(d, "ds_read_b32 -> wait -> address from data (per round trip)", 16);
    run<56>(d, "ds_read_b64 -> wait -> address from data (per round trip)", 16);
    run<70>(d, "ds_read2_b64 + ds_write_b64 behind it,  8 adds, wait(1)", 16);
    run<71>(d, "ds_read2_b64 + ds_add_u64   behind it,  8 adds, wait(1)", 16);
    run<72>(d, "ds_read2_b64 + ds_write_b64 behind it, 16 adds, wait(1)", 16);
    run<73>(d, "ds_read2_b64 + ds_add_u64   behind it, 16 adds, wait(1)", 16);
    run<74>(d, "ds_read2_b64 + ds_write_b64 behind it, 24 adds, wait(1)", 16);
    run<75>(d, "ds_read2_b64 + ds_add_u64   behind it, 24 adds, wait(1)", 16);
    run<76>(d, "ds_read2_b64 alone, 24 adds, wait(0)", 16);
    run<77>(d, "ds_read2_b64 + ds_write_b64 + ds_read_b32 behind it, 24 adds, wait(1)", 16);
    run_full<72>(d, "ds_read2_b64 + ds_write_b64 behind it, 16 adds, wait(1)", 16);
    run_full<73>(d, "ds_read2_b64 + ds_add_u64   behind it, 16 adds, wait(1)", 16);
    run_full<0>(d, "v_add_u32 chain, distance 1 (per instruction)", 64);
    run_full<2>(d, "v_add_u32 four chains (per instruction)", 256);
    run_full<4>(d, "v_mul_u32_u24 four chains (per instruction)", 256);
    run_full<11>(d, "decision NEW + addc + select (per decision, 5 instr)", 64);
    run_full<52>(d, "ds_read_b128 + ds_write_b128 behind it, 16 adds, wait (per round trip)", 16);
    printf("-- LDS round trip with N adds in the shadow: 1 wavefront alone | 4 wavefronts per CU on the whole chip\n");
    run<60>(d, "read+write,  8 adds, wait(1)", 16);   run_full<60>(d, "read+write,  8 adds, wait(1)", 16);
    run<61>(d, "read+write, 16 adds, wait(1)", 16);   run_full<61>(d, "read+write, 16 adds, wait(1)", 16);
    run<62>(d, "read+write, 24 adds, wait(1)", 16);   run_full<62>(d, "read+write, 24 adds, wait(1)", 16);
    run<63>(d, "read+write, 32 adds, wait(1)", 16);   run_full<63>(d, "read+write, 32 adds, wait(1)", 16);
    run<64>(d, "read+write, 48 adds, wait(1)", 16);   run_full<64>(d, "read+write, 48 adds, wait(1)", 16);
    run<65>(d, "read only,   8 adds, wait(0)", 16);   run_full<65>(d, "read only,   8 adds, wait(0)", 16);
    run<66>(d, "read only,  16 adds, wait(0)", 16);   run_full<66>(d, "read only,  16 adds, wait(0)", 16);
    run<67>(d, "read only,  24 adds, wait(0)", 16);   run_full<67>(d, "read only,  24 adds, wait(0)", 16);
    run<68>(d, "read only,  32 adds, wait(0)", 16);   run_full<68>(d, "read only,  32 adds, wait(0)", 16);
    return 0;
}
