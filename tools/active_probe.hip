// How does a gfx950 SIMD retire integer vector instructions when it hosts FOUR wavefronts of which only K are
// issuing (the others wait at an s_barrier)?  This is the encoder's situation: workgroups of four wavefronts, one
// per SIMD, 40 KiB of LDS each -> four workgroups per CU -> every SIMD hosts four wavefronts; roles are dealt by
// SIMD through a per-CU ticket exactly as encode_kernel does, so every SIMD hosts exactly one wavefront of each
// role.  Roles < K run the instruction stream, the others go straight to the barrier.
// tools/valu_probe.hip measured RESIDENT wavefronts (1..4 one-wavefront workgroups per SIMD): three resident ones
// retire VOP3 instructions at 7.0 cycles per SIMD slot, four at 4.35.  Question: do three ACTIVE + one parked
// behave like three or like four?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/active_probe.bin tools/active_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
__device__ uint32_t g_ticket[2048];

template <int KIND>
__global__ void __launch_bounds__(256) probe(uint32_t *out, int iters, int active) {
    __shared__ uint32_t lds[40 * 1024 / 4 - 64];
    __shared__ uint32_t hello[8];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    const uint32_t simd = (hw >> 4) & 3u;
    if (lane == 0) hello[wave] = simd;
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        hello[4] = atomicAdd(&g_ticket[((xcc & 7u) << 8) | ((hw >> 8) & 0xFFu)], 1u);
    }
    __syncthreads();
    const uint32_t seen = (1u << hello[0]) | (1u << hello[1]) | (1u << hello[2]) | (1u << hello[3]);
    const uint32_t by_simd = (simd - hello[4] - 1u) & 3u;
    const uint32_t role = __builtin_amdgcn_readfirstlane(seen == 0xFu ? by_simd : wave);
    __syncthreads();
    uint32_t a = threadIdx.x, b = blockIdx.x + 1, c = 7, d = 9;
    lds[threadIdx.x] = a;
    if (role < static_cast<uint32_t>(active)) {
        for (int i = 0; i < iters; ++i) {
            if (KIND == 0) asm volatile(REP64("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (KIND == 1) asm volatile(REP64("v_mad_u32_u24 %0, %0, %2, %3\n v_mad_u32_u24 %1, %1, %2, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (KIND == 2) asm volatile(REP64("v_add_u32 %0, %0, %2\n v_mad_u32_u24 %1, %1, %2, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (KIND == 3) asm volatile(REP64("v_mul_u32_u24 %0, %0, %2\n v_mul_u32_u24 %1, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        }
    }
    __syncthreads();
    if (a + b == 0x12345) out[blockIdx.x] = a + lds[0];
}

template <int KIND>
void run(const char *name, uint32_t *d) {
    const int iters = 2000;
    for (int active = 1; active <= 4; ++active) {
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        probe<KIND><<<256 * 4, 256>>>(d, 10, active);
        hipEventRecord(a);
        probe<KIND><<<256 * 4, 256>>>(d, iters, active);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        const double cycles = ms * 1e-3 * 2.4e9;
        printf("%-34s 4 resident / %d issuing per SIMD: %6.2f cycles per instruction per wavefront, %5.2f per SIMD slot\n", name, active,
               cycles / (iters * 128.0), cycles / (iters * 128.0 * active));
    }
}
int main() {
    uint32_t *d;
    hipMalloc(&d, 1 << 20);
    run<0>("v_add_u32 (fast class)", d);
    run<1>("v_mad_u32_u24 (VOP3, slow class)", d);
    run<2>("add + mad alternating", d);
    run<3>("v_mul_u32_u24 (VOP2, slow class)", d);
    return 0;
}
