// Where does the dispatcher put the two wavefronts of a 128-thread, 40 KiB-LDS workgroup?
// Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/hwid_probe tools/hwid_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(128) probe(uint32_t *out, int spin) {
    __shared__ uint32_t pad[10240];
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
    uint32_t acc = hw;
    for (int i = 0; i < spin; ++i) acc = acc * 1664525u + 1013904223u;
    pad[threadIdx.x] = acc;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 2 + wave) * 2 + 0] = hw;
        out[(blockIdx.x * 2 + wave) * 2 + 1] = xcc | (pad[(threadIdx.x + 1) & 127] & 0);
    }
}
int main() {
    const int blocks = 1024;   // exactly 4 per CU
    uint32_t *d;
    hipMalloc(&d, blocks * 4 * sizeof(uint32_t));
    probe<<<blocks, 128>>>(d, 200000);
    std::vector<uint32_t> h(blocks * 4);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // per (xcc, se, sh, cu): list of (simd, slot, wave-in-wg)
    std::map<uint32_t, std::vector<uint32_t>> cu;
    int simd_of_wave[2][4] = {{0}};
    int slotpar[2][2] = {{0}};
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 2; ++w) {
            uint32_t hw = h[(b * 2 + w) * 2], xcc = h[(b * 2 + w) * 2 + 1];
            uint32_t slot = hw & 15, simd = (hw >> 4) & 3, cuid = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            simd_of_wave[w][simd]++;
            slotpar[w][slot & 1]++;
            cu[(xcc << 16) | (se << 8) | (sh << 4) | cuid].push_back((simd << 8) | (slot << 4) | w);
        }
    printf("CUs seen: %zu\n", cu.size());
    printf("wave0 per SIMD: %d %d %d %d   wave1 per SIMD: %d %d %d %d\n", simd_of_wave[0][0], simd_of_wave[0][1], simd_of_wave[0][2], simd_of_wave[0][3],
           simd_of_wave[1][0], simd_of_wave[1][1], simd_of_wave[1][2], simd_of_wave[1][3]);
    printf("slot parity: wave0 even %d odd %d ; wave1 even %d odd %d\n", slotpar[0][0], slotpar[0][1], slotpar[1][0], slotpar[1][1]);
    int shown = 0;
    for (auto &kv : cu) {
        if (shown++ >= 6) break;
        printf("CU %06x:", kv.first);
        for (uint32_t v : kv.second) printf(" [simd%u slot%u w%u]", v >> 8, (v >> 4) & 15, v & 1);
        printf("\n");
    }
    // how many SIMDs host two wave0s / two wave1s / one of each
    int same = 0, mixed = 0, other = 0;
    for (auto &kv : cu) {
        int cnt[4][2] = {{0}};
        for (uint32_t v : kv.second) cnt[v >> 8][v & 1]++;
        for (int s = 0; s < 4; ++s) {
            if (cnt[s][0] + cnt[s][1] != 2) other++;
            else if (cnt[s][0] == 1) mixed++;
            else same++;
        }
    }
    printf("SIMDs with two same-index waves: %d, one of each: %d, other occupancy: %d\n", same, mixed, other);
    return 0;
}
