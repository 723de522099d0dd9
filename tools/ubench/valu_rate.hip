// VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction per SIMD for the ops the
// W4A16 GEMV is made of.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b = 0x3c003c00u + threadIdx.x, c = 0x2c002c00u;
    uint32_t m = __builtin_amdgcn_readfirstlane(0x000f000fu);
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP8(asm volatile("v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %1, %1, %8, %9\n v_pk_fma_f16 %2, %2, %8, %9\n v_pk_fma_f16 %3, %3, %8, %9\n v_pk_fma_f16 %4, %4, %8, %9\n v_pk_fma_f16 %5, %5, %8, %9\n v_pk_fma_f16 %6, %6, %8, %9\n v_pk_fma_f16 %7, %7, %8, %9" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c));) }
        if (OP == 1) { REP8(asm volatile("v_and_or_b32 %0, %0, %10, %9\n v_and_or_b32 %1, %1, %10, %9\n v_and_or_b32 %2, %2, %10, %9\n v_and_or_b32 %3, %3, %10, %9\n v_and_or_b32 %4, %4, %10, %9\n v_and_or_b32 %5, %5, %10, %9\n v_and_or_b32 %6, %6, %10, %9\n v_and_or_b32 %7, %7, %10, %9" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c),"s"(m));) }
        if (OP == 2) { REP8(asm volatile("v_pk_add_f16 %0, %0, %8\n v_pk_add_f16 %1, %1, %8\n v_pk_add_f16 %2, %2, %8\n v_pk_add_f16 %3, %3, %8\n v_pk_add_f16 %4, %4, %8\n v_pk_add_f16 %5, %5, %8\n v_pk_add_f16 %6, %6, %8\n v_pk_add_f16 %7, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c));) }
        if (OP == 3) { REP8(asm volatile("v_fma_mix_f32 %0, %8, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, 1.0, %1 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %8, 1.0, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, 1.0, %3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %8, 1.0, %4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, 1.0, %5 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %8, 1.0, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, 1.0, %7 op_sel_hi:[1,0,0]" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c));) }
        if (OP == 4) { REP8(asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3\n v_lshrrev_b32 %4, 1, %4\n v_lshrrev_b32 %5, 1, %5\n v_lshrrev_b32 %6, 1, %6\n v_lshrrev_b32 %7, 1, %7" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c));) }
        if (OP == 5) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c));) }
        if (OP == 6) { REP8(asm volatile("v_dot2_f32_f16 %0, %8, %9, %0\n v_dot2_f32_f16 %1, %8, %9, %1\n v_dot2_f32_f16 %2, %8, %9, %2\n v_dot2_f32_f16 %3, %8, %9, %3\n v_dot2_f32_f16 %4, %8, %9, %4\n v_dot2_f32_f16 %5, %8, %9, %5\n v_dot2_f32_f16 %6, %8, %9, %6\n v_dot2_f32_f16 %7, %8, %9, %7" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c));) }
        if (OP == 7) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c));) }
        if (OP == 8) { REP8(asm volatile("v_pk_fma_f16 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %10, %9\n v_pk_fma_f16 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %10, %9\n v_pk_fma_f16 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %10, %9\n v_pk_fma_f16 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %10, %9" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c),"s"(m));) }
        if (OP == 9) { REP8(asm volatile("v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %0, %0, %8, %9" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP>
double run(int wgs_per_cu, uint32_t* out) {
    int iters = 200;
    dim3 grid(256 * wgs_per_cu), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<grid, block>>>(out, iters, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<grid, block>>>(out, iters, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: waves per SIMD = wgs_per_cu (4 waves per WG over 4 SIMDs)
    double instr_per_simd = (double)wgs_per_cu * iters * 64.0;
    return ms * 1e-3 / instr_per_simd;  // seconds per instruction per SIMD
}

int main() {
    uint32_t* out;
    hipMalloc(&out, 256 * 16 * 256 * 4);
    const char* names[] = {"v_pk_fma_f16", "v_and_or_b32", "v_pk_add_f16", "v_fma_mix_f32", "v_lshrrev_b32", "v_fma_f32", "v_dot2_f32_f16", "v_mad_u32_u24", "mix pk_fma/and_or", "pk_fma dependent chain"};
    for (int w : {1, 2, 4, 8}) {
        printf("waves/SIMD=%d:", w);
        double t[10];
        t[0] = run<0>(w, out); t[1] = run<1>(w, out); t[2] = run<2>(w, out); t[3] = run<3>(w, out); t[4] = run<4>(w, out);
        t[5] = run<5>(w, out); t[6] = run<6>(w, out); t[7] = run<7>(w, out); t[8] = run<8>(w, out); t[9] = run<9>(w, out);
        for (int i = 0; i < 10; ++i) printf("  %s=%.2fns", names[i], t[i] * 1e9);
        printf("\n");
    }
    return 0;
}
