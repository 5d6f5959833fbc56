// Where does the dispatcher put the wavefronts of a workgroup of W wavefronts with L bytes of LDS, when the chip
// is filled with exactly as many workgroups as fit?  Prints, per configuration, how many wavefronts each SIMD of a
// CU ends up hosting and which SIMD each wavefront index of a workgroup lands on.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/placement_probe.bin tools/placement_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void probe(uint32_t *out, int spin) {
    extern __shared__ uint32_t pad[];
    const uint32_t waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
    uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_ID
    uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));     // XCC_ID
    uint32_t acc = hw;
    for (int i = 0; i < spin; ++i) acc = acc * 1664525u + 1013904223u;
    pad[threadIdx.x] = acc;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * waves + wave) * 2 + 0] = hw;
        out[(blockIdx.x * waves + wave) * 2 + 1] = xcc | (pad[(threadIdx.x + 1) % blockDim.x] & 0);
    }
}
static void run(int waves, int lds_bytes, int wg_per_cu) {
    const int blocks = 256 * wg_per_cu;
    uint32_t *d;
    hipMalloc(&d, blocks * waves * 2 * sizeof(uint32_t));
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    probe<<<blocks, waves * 64, lds_bytes>>>(d, 300000);
    std::vector<uint32_t> h(blocks * waves * 2);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<uint32_t, std::vector<uint32_t>> cu;
    std::vector<std::vector<int>> simd_of_wave(waves, std::vector<int>(4, 0));
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) {
            uint32_t hw = h[(b * waves + w) * 2], xcc = h[(b * waves + w) * 2 + 1];
            uint32_t simd = (hw >> 4) & 3, cuid = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            simd_of_wave[w][simd]++;
            cu[(xcc << 16) | (se << 8) | (sh << 4) | cuid].push_back((simd << 8) | w);
        }
    printf("== %d wavefronts/workgroup, %d B LDS, %d workgroups/CU launched (%d waves/CU): CUs seen %zu\n", waves, lds_bytes,
           wg_per_cu, waves * wg_per_cu, cu.size());
    for (int w = 0; w < waves; ++w)
        printf("   wavefront %d of a workgroup lands on SIMD0..3: %d %d %d %d\n", w, simd_of_wave[w][0], simd_of_wave[w][1],
               simd_of_wave[w][2], simd_of_wave[w][3]);
    std::map<std::vector<int>, int> patterns;                 // sorted per-SIMD occupancy -> number of CUs
    for (auto &kv : cu) {
        std::vector<int> occ(4, 0);
        for (uint32_t v : kv.second) occ[v >> 8]++;
        patterns[occ]++;
    }
    for (auto &kv : patterns) printf("   SIMD occupancy %d %d %d %d on %d CUs\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
    hipFree(d);
}
int main() {
    run(1, 160 * 1024 / 12 - 512, 12);      // what tools/valu_probe.hip does at "3 wave/SIMD"
    run(1, 160 * 1024 / 16 - 512, 16);
    run(2, 40 * 1024, 4);                   // the two-role encoder shape
    run(3, 40 * 1024, 4);                   // the three-role encoder shape (encode_kernel)
    run(4, 40 * 1024, 4);                   // a four-role shape
    run(1, 36 * 1024, 4);                   // decode kernels
    return 0;
}
