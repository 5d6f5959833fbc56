// What does one u16 LDS read / write cost a CU in the access pattern of the codec kernels
// (row-major u16 tables, row chosen per lane, lane-minor columns, few wavefronts per CU)?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/lds_probe.bin tools/lds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NR, int NW, int NV, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) probe(uint32_t *out, int iters) {
    __shared__ uint16_t tab[16000];   // 32000 B
    extern __shared__ uint8_t pad[];                            // dynamic: sets workgroups per CU
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint16_t *col = tab + (((lane & 31) << 1) | (lane >> 5));
    for (int r = wave; r < 256; r += WAVES) col[r * 64] = r;
    __syncthreads();
    // addresses come from `a` (never touched by loaded data: reads can be in flight while
    // the arithmetic runs, as in the kernels); the arithmetic chain on `x` is not foldable
    uint32_t x = lane * 2654435761u + blockIdx.x, a = x ^ 0x9E3779B9u, acc = 0;
    if (wave == 0 || WAVES == 1) {
        for (int i = 0; i < iters; ++i) {
            a = a * 1664525u + 1013904223u;
            uint32_t v[NR > 0 ? NR : 1];
#pragma unroll
            for (int k = 0; k < NR; ++k) v[k] = col[(((a >> (k + 1)) + 37 * k) & 249) * 64];
#pragma unroll
            for (int k = 0; k < NV; ++k) x = (x ^ (x >> 7)) + (x << 3) + k;
#pragma unroll
            for (int k = 0; k < NW; ++k) col[(((a >> (k + 9)) + 91 * k) & 249) * 64] = (uint16_t)(x + k);
#pragma unroll
            for (int k = 0; k < NR; ++k) acc += v[k];
            acc += x;
        }
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < NV; ++k) x = (x ^ (x >> 7)) + (x << 3) + k;
            acc += x;
        }
    }
    if (acc == 0x12345) out[blockIdx.x] = acc + pad[0];
}
template <int NR, int NW, int NV, int WAVES>
void run(const char *name, int wg_per_cu) {
    uint32_t *d;
    hipMalloc(&d, 1 << 20);
    const int iters = 20000, blocks = 256 * wg_per_cu;
    const size_t dyn = 160 * 1024 / wg_per_cu - 32000 - 256;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    probe<NR, NW, NV, WAVES><<<blocks, 64 * WAVES, dyn>>>(d, 100);
    hipEventRecord(a);
    probe<NR, NW, NV, WAVES><<<blocks, 64 * WAVES, dyn>>>(d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-34s wg/cu=%d waves/wg=%d : %7.1f cycles per iteration (at 2.4 GHz)\n", name, wg_per_cu, WAVES, ms * 1e-3 * 2.4e9 / iters);
    hipFree(d);
}
int main() {
    // NV counts 4-instruction groups (xor-shift, shift, add, add)
    run<0, 0, 11, 2>("2 waves x 44 valu", 4);
    run<7, 0, 11, 2>("44 valu + 7 rd | 44 valu", 4);
    run<0, 7, 11, 2>("44 valu + 7 wr | 44 valu", 4);
    run<7, 7, 11, 2>("44 valu + 7 rd + 7 wr | 44 valu", 4);
    run<0, 0, 37, 1>("148 valu (1 wave)", 5);
    run<14, 6, 37, 1>("148 valu + 14 rd + 6 wr (1 wave)", 5);
    run<2, 6, 37, 1>("148 valu + 2 rd + 6 wr (1 wave)", 5);
    run<0, 0, 37, 1>("148 valu (1 wave)", 4);
    run<14, 6, 37, 1>("148 valu + 14 rd + 6 wr (1 wave)", 4);
    return 0;
}
