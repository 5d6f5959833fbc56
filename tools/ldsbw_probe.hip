// How many LDS instructions per cycle does a CU of gfx950 take, with nothing else going on?  Every lane at its own
// row of a lane-minor u16 table (lanes l and l+32 share a dword: no bank conflicts inside a half-wave), 32 LDS
// instructions back to back per iteration, 4 / 8 / 16 wavefronts per CU.  Answers whether the encoder's ~75 LDS
// instructions per CU and symbol step (~335 cycles) are anywhere near what the LDS can take.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ldsbw_probe.bin tools/ldsbw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
template <int KIND>
__global__ void __launch_bounds__(64) probe(uint32_t *out, int iters) {
    extern __shared__ uint8_t lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t col = (((lane & 31u) << 1) | (lane >> 5)) << 1;       // the codec's lane_column, in bytes
    uint32_t a0 = (((lane * 5u) & 15u) << 7) + col;
    uint32_t a1 = a0 + 16 * 128, a2 = a0 + 32 * 128, a3 = a0 + 48 * 128;
    uint32_t x = lane, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    for (uint32_t i = lane; i < 2048; i += 64) reinterpret_cast<uint32_t *>(lds)[i] = i;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) asm volatile(REP8("ds_read_u16 %0, %4\n ds_read_u16 %1, %5\n ds_read_u16 %2, %6\n ds_read_u16 %3, %7\n") "s_waitcnt lgkmcnt(0)\n"
                                    : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
        if (KIND == 1) asm volatile(REP8("ds_write_b16 %0, %4\n ds_write_b16 %1, %4\n ds_write_b16 %2, %4\n ds_write_b16 %3, %4\n") "s_waitcnt lgkmcnt(0)\n"
                                    : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(x) : "memory");
        if (KIND == 2) asm volatile(REP8("ds_read_u16 %0, %4\n ds_write_b16 %5, %8\n ds_read_u16 %2, %6\n ds_write_b16 %7, %8\n") "s_waitcnt lgkmcnt(0)\n"
                                    : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(x) : "memory");
        if (KIND == 3) asm volatile(REP8("ds_read_b32 %0, %4\n ds_read_b32 %1, %5\n ds_read_b32 %2, %6\n ds_read_b32 %3, %7\n") "s_waitcnt lgkmcnt(0)\n"
                                    : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0 & ~3u), "v"(a1 & ~3u), "v"(a2 & ~3u), "v"(a3 & ~3u) : "memory");
        if (KIND == 4) asm volatile(REP8("ds_write_b32 %0, %4\n ds_write_b32 %1, %4\n ds_write_b32 %2, %4\n ds_write_b32 %3, %4\n") "s_waitcnt lgkmcnt(0)\n"
                                    : : "v"(lane * 4u), "v"(lane * 4u + 256u), "v"(lane * 4u + 512u), "v"(lane * 4u + 768u), "v"(x) : "memory");
        if (KIND == 5) asm volatile(REP8("ds_read_b32 %0, %4\n ds_read_b32 %1, %5\n ds_read_b32 %2, %6\n ds_read_b32 %3, %7\n") "s_waitcnt lgkmcnt(0)\n"
                                    : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(lane * 4u), "v"(lane * 4u + 256u), "v"(lane * 4u + 512u), "v"(lane * 4u + 768u) : "memory");
        x += r0 + r1 + r2 + r3;
    }
    if (x == 0x12345) out[blockIdx.x] = x;
}
template <int KIND>
void run(const char *name) {
    uint32_t *d;
    hipMalloc(&d, 1 << 20);
    const int iters = 4000;
    for (int wg_per_cu = 4; wg_per_cu <= 16; wg_per_cu *= 2) {
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
        const double cyc = ms * 1e-3 * 2.4e9 / iters;             // per iteration = 32 LDS instructions of every wavefront
        printf("%-44s %2d wavefronts/CU: %7.1f cycles per 32 instructions of a wavefront, %5.2f cycles per instruction and CU\n", name, wg_per_cu,
               cyc, cyc / (32.0 * wg_per_cu));
    }
    hipFree(d);
}
int main() {
    run<0>("ds_read_u16, a row per lane");
    run<1>("ds_write_b16, a row per lane");
    run<2>("ds_read_u16 + ds_write_b16 alternating");
    run<3>("ds_read_b32, a row per lane");
    run<4>("ds_write_b32, consecutive dwords");
    run<5>("ds_read_b32, consecutive dwords");
    return 0;
}
