// Does a lone wavefront's vector instruction cost depend on WHICH registers it names?  The decoder's step runs at 4.5 cycles per
// vector instruction where four independent chains run at 4.16 (tools/lat_probe.hip); lat_probe's dependent chain (4.63) lets the
// compiler pick the registers.  Here every stream names its registers: source operands in the same VGPR bank (index mod 4) or in
// different ones, dependent or not, two and three sources, the result into a source's own bank or another.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bank_probe.bin tools/bank_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "vcc", "s20", "s21"

template <int K>
__global__ void __launch_bounds__(64) probe(unsigned long long *out, int iters) {
    __shared__ uint4 buf[64 * 40];
    buf[threadIdx.x] = make_uint4(1, 2, 3, 4);
    __syncthreads();
    asm volatile("v_mov_b32 v10, 1\n v_mov_b32 v11, 2\n v_mov_b32 v12, 3\n v_mov_b32 v13, 4\n v_mov_b32 v14, 5\n v_mov_b32 v15, 6\n v_mov_b32 v16, 7\n"
                 "v_mov_b32 v17, 8\n v_mov_b32 v18, 9\n v_mov_b32 v19, 10\n v_mov_b32 v20, 11\n v_mov_b32 v21, 12\n v_mov_b32 v22, 13\n v_mov_b32 v23, 14\n" ::: CLOB);
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (K == 0) asm volatile(REP64("v_add_u32 v10, v10, v14\n") ::: CLOB);                          // chain, sources in the same bank
        if (K == 1) asm volatile(REP64("v_add_u32 v10, v10, v11\n") ::: CLOB);                          // chain, neighbouring banks
        if (K == 2) asm volatile(REP64("v_add_u32 v10, v10, v12\n") ::: CLOB);                          // chain, banks 2 and 0
        if (K == 3) asm volatile(REP64("v_add_u32 v10, v11, v15\n") ::: CLOB);                          // no dependence, sources in the same bank
        if (K == 4) asm volatile(REP64("v_add_u32 v10, v11, v12\n") ::: CLOB);                          // no dependence, different banks
        if (K == 5) asm volatile(REP64("v_add3_u32 v10, v10, v14, v18\n") ::: CLOB);                    // chain, three sources in one bank
        if (K == 6) asm volatile(REP64("v_add3_u32 v10, v10, v11, v12\n") ::: CLOB);                    // chain, three banks
        if (K == 7) asm volatile(REP64("v_add3_u32 v10, v11, v15, v19\n") ::: CLOB);                    // no dependence, one bank
        if (K == 8) asm volatile(REP64("v_add3_u32 v10, v11, v12, v13\n") ::: CLOB);                    // no dependence, three banks
        if (K == 9) asm volatile(REP64("v_add_u32 v11, v10, v12\n v_add_u32 v10, v11, v13\n") ::: CLOB);     // chain through two registers (ping-pong), different banks
        if (K == 10) asm volatile(REP64("v_add_u32 v14, v10, v18\n v_add_u32 v10, v14, v22\n") ::: CLOB);    // the same, everything in bank 2
        if (K == 11) asm volatile(REP64("v_mul_u32_u24 v10, v10, v14\n") ::: CLOB);
        if (K == 12) asm volatile(REP64("v_mul_u32_u24 v10, v10, v11\n") ::: CLOB);
        if (K == 13) asm volatile(REP64("v_sub_co_u32 v11, s[20:21], v10, v12\n v_min_u32 v10, v10, v11\n") ::: CLOB);   // a decision's tail, different banks
        if (K == 14) asm volatile(REP64("v_sub_co_u32 v14, s[20:21], v10, v18\n v_min_u32 v10, v10, v14\n") ::: CLOB);   // the same in one bank
        if (K == 15) asm volatile(REP64("v_add_u32 v10, v10, v11\n v_add_u32 v12, v12, v13\n v_add_u32 v14, v14, v15\n v_add_u32 v16, v16, v17\n") ::: CLOB);   // four chains
        if (K == 16) asm volatile(REP64("v_cndmask_b32_e64 v10, v10, v11, s[20:21]\n") ::: CLOB);
        if (K == 17) asm volatile(REP64("v_cndmask_b32_e64 v10, v10, v14, s[20:21]\n") ::: CLOB);
        if (K == 18) asm volatile(REP64("v_add_u32 v10, 1, v10\n") ::: CLOB);                           // chain with an inline constant: one register source
        if (K == 19) asm volatile(REP64("v_add_u32 v10, s20, v10\n") ::: CLOB);                         // chain with a scalar source
    }
    unsigned long long t1 = clock64();
    uint32_t r;
    asm volatile("v_mov_b32 %0, v10" : "=v"(r)::CLOB);
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (r == 0x12345u) out[1] = r;
}

template <int K>
static double run(unsigned long long *d, int iters) {
    probe<K><<<1, 64>>>(d, 10);
    (void)hipDeviceSynchronize();
    probe<K><<<1, 64>>>(d, iters);
    (void)hipDeviceSynchronize();
    unsigned long long t = 0;
    (void)hipMemcpy(&t, d, sizeof t, hipMemcpyDeviceToHost);
    return static_cast<double>(t);
}

int main() {
    unsigned long long *d;
    (void)hipMalloc(&d, 16);
    const int iters = 3000;
    const double four = run<15>(d, iters) / (iters * 256.0);      // four independent chains: 4.16 cycles per instruction (tools/lat_probe.hip)
    printf("# unit: cycles per instruction if an instruction of four independent chains takes 4.16 (tools/lat_probe.hip)\n");
#define ROW(K, N, NAME) printf("%-100s %6.2f\n", NAME, run<K>(d, iters) / (iters * 64.0 * (N)) / four * 4.16);
    ROW(0, 1, "v_add_u32 chain, both sources in ONE bank (v10, v14)")
    ROW(1, 1, "v_add_u32 chain, sources in neighbouring banks (v10, v11)")
    ROW(2, 1, "v_add_u32 chain, sources two banks apart (v10, v12)")
    ROW(18, 1, "v_add_u32 chain, one register source + inline constant")
    ROW(19, 1, "v_add_u32 chain, one register source + scalar register")
    ROW(3, 1, "v_add_u32 independent, both sources in ONE bank")
    ROW(4, 1, "v_add_u32 independent, different banks")
    ROW(5, 1, "v_add3_u32 chain, three sources in ONE bank")
    ROW(6, 1, "v_add3_u32 chain, three banks")
    ROW(7, 1, "v_add3_u32 independent, ONE bank")
    ROW(8, 1, "v_add3_u32 independent, three banks")
    ROW(9, 2, "v_add_u32 chain through two registers, different banks (per instruction)")
    ROW(10, 2, "v_add_u32 chain through two registers, everything in ONE bank (per instruction)")
    ROW(11, 1, "v_mul_u32_u24 chain, ONE bank")
    ROW(12, 1, "v_mul_u32_u24 chain, neighbouring banks")
    ROW(13, 2, "v_sub_co_u32 -> v_min_u32 (a decision's tail), different banks (per instruction)")
    ROW(14, 2, "v_sub_co_u32 -> v_min_u32, ONE bank (per instruction)")
    ROW(16, 1, "v_cndmask_b32_e64 chain, neighbouring banks")
    ROW(17, 1, "v_cndmask_b32_e64 chain, ONE bank")
    ROW(15, 4, "four independent v_add_u32 chains (the unit)")
    return 0;
}
