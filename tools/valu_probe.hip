// How fast does one SIMD of gfx950 retire wave64 INTEGER vector instructions, with one or two
// wavefronts resident on it?  Exact instruction streams via inline asm (64 instructions per block,
// independent or chained), one or two 64-thread workgroups' worth per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_probe.bin tools/valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND>
__global__ void __launch_bounds__(64) probe(uint32_t *out, int iters, int dyn_unused) {
    extern __shared__ uint8_t pad[];
    uint32_t a = threadIdx.x, b = blockIdx.x + 1, c = 7, d = 9;
    unsigned long long w = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) asm volatile(REP64("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 1) asm volatile(REP64("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 2) asm volatile(REP64("v_and_b32 %0, %0, %2\n v_and_b32 %1, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 3) asm volatile(REP64("v_and_or_b32 %0, %0, %2, %3\n v_and_or_b32 %1, %1, %2, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 4) asm volatile(REP64("v_lshl_or_b32 %0, %0, 1, %3\n v_lshl_or_b32 %1, %1, 1, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 5) asm volatile(REP64("v_xad_u32 %0, %0, 1, %3\n v_xad_u32 %1, %1, 1, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 6) asm volatile(REP64("v_add3_u32 %0, %0, %2, %3\n v_add3_u32 %1, %1, %2, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 7) asm volatile(REP64("v_bfe_u32 %0, %0, 1, 20\n v_bfe_u32 %1, %1, 1, 20\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 8) asm volatile(REP64("v_mul_u32_u24 %0, %0, %2\n v_mul_u32_u24 %1, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 9) asm volatile(REP64("v_mad_u32_u24 %0, %0, %2, %3\n v_mad_u32_u24 %1, %1, %2, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 10) asm volatile(REP64("v_mul_hi_u32 %0, %0, %2\n v_mul_hi_u32 %1, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 11) asm volatile(REP64("v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 12) asm volatile(REP64("v_cndmask_b32_e64 %0, %0, %2, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 13) asm volatile(REP64("v_cmp_lt_u32 vcc, %0, %2\n v_cmp_lt_u32 vcc, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 14) asm volatile(REP64("v_cmp_lt_u32_e64 s[22:23], %0, %2\n v_cmp_lt_u32_e64 s[22:23], %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 15) asm volatile(REP64("v_ffbh_u32 %0, %0\n v_ffbh_u32 %1, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 16) asm volatile(REP64("v_not_b32 %0, %0\n v_not_b32 %1, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 17) asm volatile(REP64("v_perm_b32 %0, %0, %2, %3\n v_perm_b32 %1, %1, %2, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 18) asm volatile(REP64("v_alignbit_b32 %0, %0, %2, %3\n v_alignbit_b32 %1, %1, %2, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 19) asm volatile(REP64("v_lshrrev_b64 %4, 3, %4\n v_lshrrev_b64 %4, 3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 20) asm volatile(REP64("v_sub_u32_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_sub_u32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 21) asm volatile(REP64("v_bfm_b32 %0, %0, %2\n v_bfm_b32 %1, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 22) asm volatile(REP64("v_subb_co_u32 %0, vcc, %0, %2, vcc\n v_subb_co_u32 %1, vcc, %1, %2, vcc\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 23) asm volatile(REP64("v_and_b32 %0, 0x407f, %0\n v_and_b32 %1, 0x607f, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 24) asm volatile(REP64("v_and_b32 %0, s20, %0\n v_and_b32 %1, s21, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 25) asm volatile(REP64("v_bitop3_b32 %0, %0, s20, %2 bitop3:0xc8\n v_bitop3_b32 %1, %1, s20, %2 bitop3:0xc8\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 26) asm volatile(REP64("v_or_b32 %0, %0, %2\n v_or_b32 %1, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 27) asm volatile(REP64("v_xor_b32 %0, %0, %2\n v_xor_b32 %1, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 28) asm volatile(REP64("v_sub_u32 %0, %0, %2\n v_sub_u32 %1, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 29) asm volatile(REP64("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 30) asm volatile(REP64("v_min_u32 %0, %0, %2\n v_min_u32 %1, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 31) asm volatile(REP64("v_max_u32 %0, %0, %2\n v_max_u32 %1, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 32) asm volatile(REP64("v_mov_b32 %0, %2\n v_mov_b32 %1, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 33) asm volatile(REP64("v_lshl_add_u32 %0, %0, 1, %2\n v_lshl_add_u32 %1, %1, 1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 34) asm volatile(REP64("v_sub_co_u32 %0, vcc, %0, %2\n v_sub_co_u32 %1, vcc, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 35) asm volatile(REP64("v_sub_co_u32 %0, s[22:23], %0, %2\n v_sub_co_u32 %1, s[22:23], %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 36) asm volatile(REP64("v_addc_co_u32 %0, vcc, %0, %2, vcc\n v_addc_co_u32 %1, vcc, %1, %2, vcc\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 37) asm volatile(REP64("v_mul_u32_u24_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n v_mul_u32_u24_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 38) asm volatile(REP64("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 39) asm volatile(REP64("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 40) asm volatile(REP64("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 41) asm volatile(REP64("v_pk_add_u16 %0, %0, %2\n v_pk_add_u16 %1, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 42) asm volatile(REP64("v_pk_mul_lo_u16 %0, %0, %2\n v_pk_mul_lo_u16 %1, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 43) asm volatile(REP64("v_mul_lo_u32 %0, %0, %2\n v_mul_lo_u32 %1, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 44) asm volatile(REP64("v_mad_u64_u32 %4, vcc, %0, %2, %4\n v_mad_u64_u32 %4, vcc, %1, %2, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 45) asm volatile(REP64("v_add_u32_dpp %0, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
        if (KIND == 46) asm volatile(REP64("v_and_or_b32 %0, %0, %2, %3\n v_and_or_b32 %1, %1, %2, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(w) : : "vcc", "s20", "s21", "s22", "s23");
    }
    if (a + b + (uint32_t)w == 0x12345) out[blockIdx.x] = a + pad[0];
}
template <int KIND>
void run(const char *name) {
    uint32_t *d;
    hipMalloc(&d, 1 << 20);
    const int iters = 2000;
    for (int waves_per_simd = 1; waves_per_simd <= 4; ++waves_per_simd) {
        const int wg_per_cu = 4 * waves_per_simd;
        const size_t dyn = 160 * 1024 / wg_per_cu - 512;      // LDS footprint pins the residency
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        probe<KIND><<<256 * wg_per_cu, 64, dyn>>>(d, 10, 0);
        hipEventRecord(a);
        probe<KIND><<<256 * wg_per_cu, 64, dyn>>>(d, iters, 0);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        const double cycles = ms * 1e-3 * 2.4e9;
        printf("%-44s %d wave/SIMD: %5.2f cycles per instruction per wave, %5.2f per SIMD slot\n", name, waves_per_simd,
               cycles / (iters * 128.0), cycles / (iters * 128.0 * waves_per_simd));
    }
    hipFree(d);
}
int main() {
    run<0>("v_add_u32 (VOP2)");
    run<1>("v_lshrrev_b32 (VOP2)");
    run<2>("v_and_b32 (VOP2)");
    run<3>("v_and_or_b32 (VOP3)");
    run<4>("v_lshl_or_b32 (VOP3)");
    run<5>("v_xad_u32 (VOP3)");
    run<6>("v_add3_u32 (VOP3)");
    run<7>("v_bfe_u32 (VOP3)");
    run<8>("v_mul_u32_u24 (VOP2)");
    run<9>("v_mad_u32_u24 (VOP3)");
    run<10>("v_mul_hi_u32 (VOP3)");
    run<11>("v_cndmask_b32 vcc (VOP2)");
    run<12>("v_cndmask_b32 sgpr (VOP3)");
    run<13>("v_cmp_lt_u32 -> vcc (VOPC)");
    run<14>("v_cmp_lt_u32 -> sgpr (VOP3)");
    run<15>("v_ffbh_u32 (VOP1)");
    run<16>("v_not_b32 (VOP1)");
    run<17>("v_perm_b32 (VOP3)");
    run<18>("v_alignbit_b32 (VOP3)");
    run<19>("v_lshrrev_b64 (VOP3)");
    run<20>("v_sub_u32_sdwa");
    run<21>("v_bfm_b32 (VOP3)");
    run<22>("v_subb_co_u32 (VOP2)");
    run<23>("v_and_b32 literal (VOP2 + 32-bit literal)");
    run<24>("v_and_b32 sgpr (VOP2)");
    run<25>("v_bitop3_b32 (VOP3)");
    run<26>("v_or_b32 (VOP2)");
    run<27>("v_xor_b32 (VOP2)");
    run<28>("v_sub_u32 (VOP2)");
    run<29>("v_lshlrev_b32 (VOP2)");
    run<30>("v_min_u32 (VOP2)");
    run<31>("v_max_u32 (VOP2)");
    run<32>("v_mov_b32 (VOP1)");
    run<33>("v_lshl_add_u32 (VOP3)");
    run<34>("v_sub_co_u32 -> vcc (VOP2)");
    run<35>("v_sub_co_u32 -> sgpr (VOP3)");
    run<36>("v_addc_co_u32 vcc (VOP2)");
    run<37>("v_mul_u32_u24_sdwa");
    run<38>("v_fma_f32 (VOP3)");
    run<39>("v_mul_f32 (VOP2)");
    run<40>("v_cvt_f32_u32 (VOP1)");
    run<41>("v_pk_add_u16 (VOP3P)");
    run<42>("v_pk_mul_lo_u16 (VOP3P)");
    run<43>("v_mul_lo_u32 (VOP3)");
    run<44>("v_mad_u64_u32 (VOP3)");
    run<45>("v_add_u32 DPP row_shr:1");
    run<46>("v_and_or_b32 literal (VOP3 + literal)");
    return 0;
}
