// Which plain device-to-device copy reaches the practical HBM roof on this part (bench.py quotes the winner, gpuar_hip_copy).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/copy_probe.bin tools/copy_probe.hip ; run: tools/copy_probe.bin [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int kInFlight>
__global__ void __launch_bounds__(256) copy_strided(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * 256u;
    size_t q = static_cast<size_t>(blockIdx.x) * 256u + threadIdx.x;
    for (; q + (kInFlight - 1) * stride < n; q += kInFlight * stride) {
        uint4 v[kInFlight];
#pragma unroll
        for (int k = 0; k < kInFlight; ++k) v[k] = src[q + k * stride];
#pragma unroll
        for (int k = 0; k < kInFlight; ++k) dst[q + k * stride] = v[k];
    }
    for (; q < n; q += stride) dst[q] = src[q];
}

// every workgroup copies one contiguous tile of kTile quads (kTile / 256 per thread), tiles dealt to workgroups in order
template <int kPerThread, bool kNontemporal>
__global__ void __launch_bounds__(256) copy_tiles(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
    const size_t base = static_cast<size_t>(blockIdx.x) * (256u * kPerThread) + threadIdx.x;
    uint4 v[kPerThread];
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        const size_t q = base + k * 256u;
        if (q < n) {
            if (kNontemporal) {
                const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + q));
                v[k] = make_uint4(t.x, t.y, t.z, t.w);
            } else {
                v[k] = src[q];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        const size_t q = base + k * 256u;
        if (q < n) {
            if (kNontemporal) {
                u32x4 t = {v[k].x, v[k].y, v[k].z, v[k].w};
                __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(dst + q));
            } else {
                dst[q] = v[k];
            }
        }
    }
}

template <typename F>
static void time_it(const char *name, size_t bytes, F launch) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) launch();
    float best = 1e9f, sum = 0;
    for (int r = 0; r < 10; ++r) {
        (void)hipEventRecord(a);
        launch();
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
        sum += ms;
    }
    printf("%-58s avg %7.3f ms  %7.1f GB/s   best %7.1f GB/s (read + write)\n", name, sum / 10, 2.0 * bytes / (sum / 10 * 1e-3) / 1e9, 2.0 * bytes / (best * 1e-3) / 1e9);
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 8.0;
    const size_t bytes = static_cast<size_t>(gib * (1ull << 30)) / 4096 * 4096, n = bytes / 16;
    uint4 *src, *dst;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess) return 1;
    (void)hipMemset(src, 0x5a, bytes);
    (void)hipMemset(dst, 0, bytes);
    printf("# %.2f GiB, 16 bytes per lane\n", gib);
    time_it("hipMemcpyAsync device to device", bytes, [&] { (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, nullptr); });
    for (int blocks : {2048, 8192, 32768})
        for (int fl : {4, 8}) {
            char name[96];
            snprintf(name, sizeof name, "grid-stride, %d in flight per lane, %d workgroups", fl, blocks);
            if (fl == 4) time_it(name, bytes, [&] { copy_strided<4><<<blocks, 256>>>(src, dst, n); });
            else time_it(name, bytes, [&] { copy_strided<8><<<blocks, 256>>>(src, dst, n); });
        }
    time_it("one tile per workgroup, 1 quad per thread", bytes, [&] { copy_tiles<1, false><<<static_cast<unsigned>((n + 255) / 256), 256>>>(src, dst, n); });
    time_it("one tile per workgroup, 4 quads per thread", bytes, [&] { copy_tiles<4, false><<<static_cast<unsigned>((n + 1023) / 1024), 256>>>(src, dst, n); });
    time_it("one tile per workgroup, 8 quads per thread", bytes, [&] { copy_tiles<8, false><<<static_cast<unsigned>((n + 2047) / 2048), 256>>>(src, dst, n); });
    time_it("one tile per workgroup, 4 quads per thread, nontemporal", bytes, [&] { copy_tiles<4, true><<<static_cast<unsigned>((n + 1023) / 1024), 256>>>(src, dst, n); });
    time_it("one tile per workgroup, 8 quads per thread, nontemporal", bytes, [&] { copy_tiles<8, true><<<static_cast<unsigned>((n + 2047) / 2048), 256>>>(src, dst, n); });
    return 0;
}
