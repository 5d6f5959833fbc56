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
        // ---- round 5: LDS atomics with return, for "one ds_add_rtn_u32 per tree level instead of read + write" ----
        // 6: lanes l and l + 32 on the SAME dword (the tree's layout: two u16 counts per dword), addend 1 << 16 * (lane >> 5)
        if (KIND == 6) asm volatile(REP8("ds_add_rtn_u32 %0, %4, %8\n ds_add_rtn_u32 %1, %5, %8\n ds_add_rtn_u32 %2, %6, %8\n ds_add_rtn_u32 %3, %7, %8\n") "s_waitcnt lgkmcnt(0)\n"
                                    : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0 & ~3u), "v"(a1 & ~3u), "v"(a2 & ~3u), "v"(a3 & ~3u), "v"(1u << (16u * (lane >> 5))) : "memory");
        // 7: every lane its own dword (what a 64 KiB tree would allow)
        if (KIND == 7) asm volatile(REP8("ds_add_rtn_u32 %0, %4, %8\n ds_add_rtn_u32 %1, %5, %8\n ds_add_rtn_u32 %2, %6, %8\n ds_add_rtn_u32 %3, %7, %8\n") "s_waitcnt lgkmcnt(0)\n"
                                    : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(lane * 4u), "v"(lane * 4u + 256u), "v"(lane * 4u + 512u), "v"(lane * 4u + 768u), "v"(1u) : "memory");
        // 8: the same shared dwords without a return value
        if (KIND == 8) asm volatile(REP8("ds_add_u32 %0, %4\n ds_add_u32 %1, %4\n ds_add_u32 %2, %4\n ds_add_u32 %3, %4\n") "s_waitcnt lgkmcnt(0)\n"
                                    : : "v"(a0 & ~3u), "v"(a1 & ~3u), "v"(a2 & ~3u), "v"(a3 & ~3u), "v"(1u << (16u * (lane >> 5))) : "memory");
        // 9: RETURN LATENCY: 32 atomics on shared dwords, each waited for and its result folded into the next one's addend
        if (KIND == 9) asm volatile(REP8("ds_add_rtn_u32 %0, %4, %8\n s_waitcnt lgkmcnt(0)\n v_and_or_b32 %1, %0, 0, %8\n ds_add_rtn_u32 %0, %5, %1\n s_waitcnt lgkmcnt(0)\n v_and_or_b32 %1, %0, 0, %8\n"
                                         "ds_add_rtn_u32 %0, %6, %1\n s_waitcnt lgkmcnt(0)\n v_and_or_b32 %1, %0, 0, %8\n ds_add_rtn_u32 %0, %7, %1\n s_waitcnt lgkmcnt(0)\n v_and_or_b32 %1, %0, 0, %8\n")
                                    : "=&v"(r0), "=&v"(r1), "=v"(r2), "=v"(r3) : "v"(a0 & ~3u), "v"(a1 & ~3u), "v"(a2 & ~3u), "v"(a3 & ~3u), "v"(1u << (16u * (lane >> 5))) : "memory");
        // 10: the round trip the modelers pay now: ds_read_u16, wait, xor-add, ds_write_b16 (32 reads + 32 writes)
        if (KIND == 10) asm volatile(REP8("ds_read_u16 %0, %4\n s_waitcnt lgkmcnt(0)\n v_xad_u32 %1, %8, 1, %0\n ds_write_b16 %4, %1\n ds_read_u16 %0, %5\n s_waitcnt lgkmcnt(0)\n v_xad_u32 %1, %8, 1, %0\n ds_write_b16 %5, %1\n"
                                          "ds_read_u16 %0, %6\n s_waitcnt lgkmcnt(0)\n v_xad_u32 %1, %8, 1, %0\n ds_write_b16 %6, %1\n ds_read_u16 %0, %7\n s_waitcnt lgkmcnt(0)\n v_xad_u32 %1, %8, 1, %0\n ds_write_b16 %7, %1\n")
                                     : "=&v"(r0), "=&v"(r1), "=v"(r2), "=v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(x & 1u) : "memory");
        // 11: the modelers' pattern without waiting for each read (software-pipelined: 4 reads in flight, then 4 writes): read + write per level
        if (KIND == 11) asm volatile(REP8("ds_read_u16 %0, %4\n ds_write_b16 %5, %8\n ds_read_u16 %1, %5\n ds_write_b16 %6, %8\n ds_read_u16 %2, %6\n ds_write_b16 %7, %8\n ds_read_u16 %3, %7\n ds_write_b16 %4, %8\n") "s_waitcnt lgkmcnt(0)\n"
                                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(x) : "memory");
        x += r0 + r1 + r2 + r3;
    }
    if (x == 0x12345) out[blockIdx.x] = x;
}
template <int KIND>
void run(const char *name) {
    uint32_t *d;
    hipMalloc(&d, 1 << 20);
    const int iters = 4000;
    for (int wg_per_cu : {4, 8, 12, 16}) {
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
    run<6>("ds_add_rtn_u32, lanes l / l+32 share a dword");
    run<7>("ds_add_rtn_u32, a dword per lane");
    run<8>("ds_add_u32 (no return), shared dwords");
    run<9>("ds_add_rtn_u32 shared, each WAITED for + 1 valu");
    run<10>("ds_read_u16, wait, xad, ds_write_b16 (x32 each)");
    run<11>("ds_read_u16 + ds_write_b16, 64 instr, no waits");
    return 0;
}
