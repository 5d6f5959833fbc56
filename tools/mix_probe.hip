// Does the double rate of the "fast class" (v_add_u32, v_and_b32, v_lshrrev_b32 ...: 2.2 cycles per wave64 instruction
// per SIMD against 4.3 for everything else, tools/valu_probe.hip) survive in MIXED instruction streams?  Four
// wavefronts per SIMD as in the encoder (tools/active_probe.hip's set-up: roles dealt by SIMD), every one issuing;
// what differs is how fast-class (add) and slow-class (mad) instructions are arranged in and across the wavefronts.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mix_probe.bin tools/mix_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP2(x) x x
#define REP4(x) x x x x
#define REP8(x) x x x x x x x x
#define REP16(x) REP8(x) REP8(x)
#define REP32(x) REP16(x) REP16(x)
#define REP64(x) REP8(REP8(x))
#define ADD "v_add_u32 %0, %0, %2\n"
#define ADD2 "v_add_u32 %1, %1, %2\n"
#define MAD "v_mad_u32_u24 %1, %1, %2, %3\n"
#define MAD2 "v_mad_u32_u24 %0, %0, %2, %3\n"
#define OPS : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
__device__ uint32_t g_ticket[2048];

template <int KIND>
__global__ void __launch_bounds__(256) probe(uint32_t *out, int iters) {
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
    const uint32_t role = __builtin_amdgcn_readfirstlane(seen == 0xFu ? by_simd : wave);   // every SIMD hosts one of each
    __syncthreads();
    uint32_t a = threadIdx.x, b = blockIdx.x + 1, c = 7, d = 9;
    lds[threadIdx.x] = a;
    for (int i = 0; i < iters; ++i) {          // 128 instructions per iteration in every variant
        if (KIND == 0) asm volatile(REP64(ADD MAD) OPS);                                     // 1 : 1 alternating
        if (KIND == 1) asm volatile(REP8(REP8(ADD) REP8(MAD)) OPS);                           // runs of 8
        if (KIND == 2) asm volatile(REP2(REP32(ADD) REP32(MAD)) OPS);                         // runs of 32
        if (KIND == 3) {                                                                       // two roles all-add, two all-mad
            if (role < 2) asm volatile(REP64(ADD ADD2) OPS); else asm volatile(REP64(MAD MAD2) OPS);
        }
        if (KIND == 4) {
            if (role & 1) asm volatile(REP64(ADD ADD2) OPS); else asm volatile(REP64(MAD MAD2) OPS);
        }
        if (KIND == 5) asm volatile(REP32(ADD ADD2 ADD MAD) OPS);                             // 3 : 1
        if (KIND == 6) asm volatile(REP32(ADD MAD MAD2 MAD) OPS);                             // 1 : 3
        if (KIND == 7) {                                                                       // three roles all-add, one all-mad
            if (role < 3) asm volatile(REP64(ADD ADD2) OPS); else asm volatile(REP64(MAD MAD2) OPS);
        }
        if (KIND == 8) asm volatile(REP64(ADD ADD2) OPS);                                      // all add
        if (KIND == 9) asm volatile(REP64(MAD MAD2) OPS);                                      // all mad
    }
    __syncthreads();
    if (a + b == 0x12345) out[blockIdx.x] = a + lds[0];
}

template <int KIND>
void run(const char *name, uint32_t *d, double expect) {
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    probe<KIND><<<256 * 4, 256>>>(d, 10);
    hipEventRecord(a);
    probe<KIND><<<256 * 4, 256>>>(d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-58s %5.2f cycles per instruction per SIMD   (if the classes simply added up: %4.2f)\n", name, cycles / (iters * 128.0 * 4), expect);
}
int main() {
    uint32_t *d;
    hipMalloc(&d, 1 << 20);
    const double f = 2.4, s = 4.4;
    run<8>("all add (fast class)", d, f);
    run<9>("all mad (slow class)", d, s);
    run<0>("add, mad alternating in every wavefront", d, (f + s) / 2);
    run<1>("8 adds, 8 mads, ... in every wavefront", d, (f + s) / 2);
    run<2>("32 adds, 32 mads, ... in every wavefront", d, (f + s) / 2);
    run<3>("wavefronts 0,1 only add / 2,3 only mad", d, (f + s) / 2);
    run<4>("wavefronts 1,3 only add / 0,2 only mad", d, (f + s) / 2);
    run<5>("3 adds : 1 mad in every wavefront", d, (3 * f + s) / 4);
    run<6>("1 add : 3 mads in every wavefront", d, (f + 3 * s) / 4);
    run<7>("wavefronts 0,1,2 only add / 3 only mad", d, (3 * f + s) / 4);
    return 0;
}
