// What does an LDS instruction cost on gfx950 in the codec kernels' regime (few wavefronts per CU,
// every lane at its own row of a lane-minor table, i.e. no bank conflicts)?  Exact instruction
// streams via inline asm: per loop iteration 64 vector adds, optionally interleaved with 16 LDS
// reads and/or 16 LDS writes at per-lane addresses that do not depend on loaded data.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/lds_probe.bin tools/lds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
template <int KIND>
__global__ void __launch_bounds__(64) probe(uint32_t *out, int iters) {
    extern __shared__ uint8_t lds[];      // 32 KiB used + padding that pins the residency
    const uint32_t lane = threadIdx.x;
    // lane-minor u16 column (as the codec), 4 different rows per lane
    uint32_t a0 = (((lane * 37u) & 63u) << 7) + ((lane & 31u) << 2) + ((lane >> 5) << 1);
    uint32_t a1 = a0 + 64 * 128, a2 = a0 + 128 * 128, a3 = a0 + 192 * 128;
    uint32_t x = lane, y = blockIdx.x, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    for (uint32_t i = lane; i < 8192; i += 64) reinterpret_cast<uint32_t *>(lds)[i] = i;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0)   // 64 adds only
            asm volatile(REP16("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0\n v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0\n") : "+v"(x), "+v"(y));
        if (KIND == 1)   // + 16 u16 reads (4 per 16 adds), results consumed at the end of the block
            asm volatile(REP4("ds_read_u16 %2, %6\n" REP4("v_add_u32 %0, %0, %1\n") "ds_read_u16 %3, %7\n" REP4("v_add_u32 %1, %1, %0\n")
                              "ds_read_u16 %4, %8\n" REP4("v_add_u32 %0, %0, %1\n") "ds_read_u16 %5, %9\n" REP4("v_add_u32 %1, %1, %0\n"))
                         "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(x), "+v"(y), "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
        if (KIND == 2)   // + 16 u16 writes
            asm volatile(REP4("ds_write_b16 %2, %0\n" REP4("v_add_u32 %0, %0, %1\n") "ds_write_b16 %3, %1\n" REP4("v_add_u32 %1, %1, %0\n")
                              "ds_write_b16 %4, %0\n" REP4("v_add_u32 %0, %0, %1\n") "ds_write_b16 %5, %1\n" REP4("v_add_u32 %1, %1, %0\n"))
                         : "+v"(x), "+v"(y) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
        if (KIND == 3)   // + 16 reads + 16 writes
            asm volatile(REP4("ds_read_u16 %2, %6\n ds_write_b16 %7, %0\n" REP4("v_add_u32 %0, %0, %1\n") "ds_read_u16 %3, %7\n ds_write_b16 %8, %1\n" REP4("v_add_u32 %1, %1, %0\n")
                              "ds_read_u16 %4, %8\n ds_write_b16 %9, %0\n" REP4("v_add_u32 %0, %0, %1\n") "ds_read_u16 %5, %9\n ds_write_b16 %6, %1\n" REP4("v_add_u32 %1, %1, %0\n"))
                         "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(x), "+v"(y), "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
        if (KIND == 4)   // + 16 b128 reads
            asm volatile(REP4("ds_read_b128 %2, %3\n" REP4("v_add_u32 %0, %0, %1\n") "ds_read_b128 %2, %4\n" REP4("v_add_u32 %1, %1, %0\n")
                              "ds_read_b128 %2, %5\n" REP4("v_add_u32 %0, %0, %1\n") "ds_read_b128 %2, %6\n" REP4("v_add_u32 %1, %1, %0\n"))
                         "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(x), "+v"(y), "=v"(*reinterpret_cast<uint4 *>(&r0)) : "v"(lane * 16), "v"(lane * 16 + 1024), "v"(lane * 16 + 2048), "v"(lane * 16 + 3072) : "memory");
        x += r0 + r1 + r2 + r3;
    }
    if (x + y == 0x12345) out[blockIdx.x] = x;
}
template <int KIND>
void run(const char *name, int lds_ops) {
    uint32_t *d;
    hipMalloc(&d, 1 << 20);
    const int iters = 4000;
    float base = 0;
    for (int wg_per_cu = 4; wg_per_cu <= 8; wg_per_cu *= 2) {
        const size_t dyn = 160 * 1024 / wg_per_cu - 512;
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        probe<KIND><<<256 * wg_per_cu, 64, dyn>>>(d, 10);
        hipEventRecord(a);
        probe<KIND><<<256 * wg_per_cu, 64, dyn>>>(d, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        const double cyc = ms * 1e-3 * 2.4e9 / iters;
        printf("%-36s %d wave/SIMD: %7.1f cycles per iteration (64 adds + %d LDS ops)\n", name, wg_per_cu / 4, cyc, lds_ops);
        (void)base;
    }
    hipFree(d);
}
int main() {
    run<0>("64 v_add_u32", 0);
    run<1>("+ 16 ds_read_u16", 16);
    run<2>("+ 16 ds_write_b16", 16);
    run<3>("+ 16 ds_read_u16 + 16 ds_write_b16", 32);
    run<4>("+ 16 ds_read_b128", 16);
    return 0;
}
