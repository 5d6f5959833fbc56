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
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) asm volatile(REP64("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %3\n") : "+v"(a), "+v"(b) : "v"(c), "v"(d));          // 128 adds, 2 chains
        if (KIND == 1) asm volatile(REP64("v_add_u32 %0, %0, %2\n v_add_u32 %0, %0, %3\n") : "+v"(a), "+v"(b) : "v"(c), "v"(d));          // 128 adds, 1 chain
        if (KIND == 2) asm volatile(REP64("v_and_or_b32 %0, %0, %2, %3\n v_lshrrev_b32 %1, 1, %1\n") : "+v"(a), "+v"(b) : "v"(c), "v"(d));
        if (KIND == 3) asm volatile(REP64("v_mul_u32_u24 %0, %0, %2\n v_mad_u32_u24 %1, %1, %2, %3\n") : "+v"(a), "+v"(b) : "v"(c), "v"(d));
        if (KIND == 4) asm volatile(REP64("v_mul_hi_u32 %0, %0, %2\n v_add_u32 %1, %1, %3\n") : "+v"(a), "+v"(b) : "v"(c), "v"(d));
        if (KIND == 5) asm volatile(REP64("v_cmp_lt_u32 vcc, %0, %2\n v_cndmask_b32 %1, %1, %3, vcc\n") : "+v"(a), "+v"(b) : "v"(c), "v"(d) : "vcc");
        if (KIND == 6) asm volatile(REP64("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n") : "+v"(a), "+v"(b) : "v"(c), "v"(d));
        if (KIND == 7) asm volatile(REP64("s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, %3\n") : "+v"(a), "+v"(b) : "v"(c), "v"(d) : "s20", "scc");
    }
    if (a + b == 0x12345) out[blockIdx.x] = a + pad[0];
}
template <int KIND>
void run(const char *name) {
    uint32_t *d;
    hipMalloc(&d, 1 << 20);
    const int iters = 2000;
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
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
    run<0>("v_add_u32 x2 independent chains");
    run<1>("v_add_u32 single dependent chain");
    run<2>("v_and_or_b32 + v_lshrrev_b32");
    run<3>("v_mul_u32_u24 + v_mad_u32_u24");
    run<4>("v_mul_hi_u32 + v_add_u32");
    run<5>("v_cmp_lt_u32 + v_cndmask_b32");
    run<6>("v_fma_f32 x2 (reference point)");
    run<7>("s_add_u32 + v_add_u32");
    return 0;
}
