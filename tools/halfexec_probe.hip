// Does a wave64 vector instruction cost a gfx950 SIMD less when only half of its lanes are enabled?
// (If it did, two half-filled wavefronts per SIMD would cost the vector pipe what one full one does, and the decoder
// could keep two wavefronts per SIMD in the LDS it has.)  Streams of one opcode, EXEC = all 64 lanes / the low 32 /
// the low 16 / the even lanes, with 1 and 2 one-wavefront workgroups per SIMD (LDS-limited residency).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/halfexec_probe.bin tools/halfexec_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND>
__global__ void __launch_bounds__(64) probe(uint32_t *out, int iters, unsigned long long mask) {
    extern __shared__ uint8_t pad[];
    uint32_t a = threadIdx.x, b = blockIdx.x + 1, c = 7, d = 9;
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n s_and_b64 exec, exec, %1" : "=s"(saved) : "s"(mask));
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) asm volatile(REP64("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (KIND == 1) asm volatile(REP64("v_mad_u32_u24 %0, %0, %2, %3\n v_mad_u32_u24 %1, %1, %2, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (KIND == 2) asm volatile(REP64("v_add_u32 %0, %0, %2\n v_mad_u32_u24 %1, %1, %2, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (KIND == 3) asm volatile(REP64("v_mul_hi_u32 %0, %0, %2\n v_mul_hi_u32 %1, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
    asm volatile("s_mov_b64 exec, %0" : : "s"(saved));
    if (a + b == 0x12345) out[blockIdx.x] = a + pad[0];
}
template <int KIND>
void run(const char *name, uint32_t *d) {
    const int iters = 2000;
    const unsigned long long masks[4] = {~0ull, 0xFFFFFFFFull, 0xFFFFull, 0x5555555555555555ull};
    const char *mname[4] = {"all 64 lanes", "low 32 lanes", "low 16 lanes", "even lanes  "};
    for (int waves = 1; waves <= 2; ++waves)
        for (int m = 0; m < 4; ++m) {
            const int wg_per_cu = 4 * waves;
            const size_t dyn = 160 * 1024 / wg_per_cu - 512;
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            probe<KIND><<<256 * wg_per_cu, 64, dyn>>>(d, 10, masks[m]);
            hipEventRecord(a);
            probe<KIND><<<256 * wg_per_cu, 64, dyn>>>(d, iters, masks[m]);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double cycles = ms * 1e-3 * 2.4e9;
            printf("%-28s %d wavefront(s)/SIMD, EXEC = %s: %6.2f cycles per instruction per wavefront, %5.2f per SIMD slot\n", name, waves, mname[m],
                   cycles / (iters * 128.0), cycles / (iters * 128.0 * waves));
        }
}
int main() {
    uint32_t *d;
    hipMalloc(&d, 1 << 20);
    run<0>("v_add_u32", d);
    run<1>("v_mad_u32_u24", d);
    run<2>("add + mad alternating", d);
    run<3>("v_mul_hi_u32", d);
    return 0;
}
