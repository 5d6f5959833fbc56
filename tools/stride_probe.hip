// How expensive are the decoder's memory patterns by themselves?  One wavefront per SIMD (LDS-pinned),
// lane l works at base + (64*block + l) * STRIDE + i:
//   store: four 16-byte stores (one 64-byte sector) per step, steps 64 bytes apart  (decoded output, stride 8192)
//   load : one dword per step, 4 bytes apart                                        (packet slots, stride 8704)
// with a configurable amount of dependent VALU work between steps.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/stride_probe.bin tools/stride_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <int MODE>
__global__ void __launch_bounds__(64) probe(uint8_t *buf, size_t stride, int steps, int work, uint32_t *sink) {
    __shared__ uint4 pin[64 * 36];                       // 36 KiB: four workgroups per CU
    pin[threadIdx.x] = make_uint4(1, 2, 3, 4);
    __syncthreads();
    uint8_t *p = buf + (static_cast<size_t>(blockIdx.x) * 64 + threadIdx.x) * stride;
    uint32_t a = threadIdx.x + pin[threadIdx.x].x, acc = 0;
    for (int s = 0; s < steps; ++s) {
        for (int w = 0; w < work; ++w) asm volatile("v_mad_u32_u24 %0, %0, 3, 1" : "+v"(a));
        if (MODE == 0) {
            uint4 *d = reinterpret_cast<uint4 *>(p + static_cast<size_t>(s) * 64);
            d[0] = make_uint4(a, a, a, a);
            d[1] = make_uint4(a, a, a, a);
            d[2] = make_uint4(a, a, a, a);
            d[3] = make_uint4(a, a, a, a);
        } else {
            acc += *reinterpret_cast<const uint32_t *>(p + static_cast<size_t>(s) * 4);
            asm volatile("" : "+v"(acc));
        }
    }
    if (a + acc == 0x12345) sink[0] = a;
}

int main() {
    const size_t max_stride = 8704 + 128;
    const size_t n = 1024ull * 64 * max_stride;
    uint8_t *buf;
    uint32_t *sink;
    (void)hipMalloc(&buf, n);
    (void)hipMalloc(&sink, 64);
    (void)hipMemset(buf, 1, n);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const size_t strides[] = {8192, 8192 + 64, 8192 + 128, 8192 + 256, 8704};
    for (int mode = 0; mode < 2; ++mode)
        for (int work : {0, 1000, 10000})
            for (size_t stride : strides) {
                const int steps = mode == 0 ? 128 : 2048;
                for (int rep = 0; rep < 2; ++rep) {
                    (void)hipEventRecord(a);
                    if (mode == 0) probe<0><<<1024, 64>>>(buf, stride, steps, work, sink);
                    else probe<1><<<1024, 64>>>(buf, stride, steps, work, sink);
                    (void)hipEventRecord(b);
                    (void)hipEventSynchronize(b);
                }
                float ms = 0;
                (void)hipEventElapsedTime(&ms, a, b);
                const double cyc = ms * 1e-3 * 2.4e9 / steps;
                printf("%s stride %5zu, %5d VALU between steps: %8.1f cycles per step (%.0f for the VALU alone) -> %7.1f GB/s\n",
                       mode == 0 ? "store 64 B/lane" : "load   4 B/lane", stride, work, cyc, work * 4.63,
                       1024.0 * 64 * steps * (mode == 0 ? 64 : 4) / (ms * 1e-3) / 1e9);
            }
    return 0;
}
