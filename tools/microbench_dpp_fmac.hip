// Microbenchmark: issue / dependency cost of fp32 fma chains as the per-XCD decoder's chain waves run them, per wave, at
// 1, 2 and 3 waves per SIMD (256 / 512 / 768 threads per workgroup, one workgroup per CU, all CUs busy):
//   dep_fmac       one dependent chain of v_fmac_f32 (plain VGPR operands)
//   dep_fmac_dpp   the same with v_fmac_f32_dpp quad_perm (operand broadcast inside the quad)
//   2x_fmac_dpp    two independent dpp chains interleaved
//   4x_fma         four independent v_fma_f32 chains interleaved
//   dep_pk_fma     one dependent chain of v_pk_fma_f32 (two fmas per instruction)
// Prints shader cycles (s_memtime) and ns (s_memrealtime) per instruction per wave.
// hipcc --offload-arch=gfx950 -O3 tools/microbench_dpp_fmac.hip -o build/mb_dpp && build/mb_dpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define QP "quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf"
typedef float f2 __attribute__((ext_vector_type(2)));
#define R8(X) X X X X X X X X
template <int MODE>
__global__ void k(float *out, const float *in, int iters, unsigned long long *cyc) {
    float w[8], h[8], a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    f2 pa = {0.f, 0.f}, pw = {in[threadIdx.x], in[threadIdx.x + 1]}, ph = {in[threadIdx.x + 2], in[threadIdx.x + 3]};
    for (int i = 0; i < 8; ++i) { w[i] = in[threadIdx.x * 8 + i]; h[i] = in[4096 + threadIdx.x * 8 + i]; }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) asm volatile(R8("v_fmac_f32 %0, %1, %2\n\t") R8("v_fmac_f32 %0, %3, %4\n\t") : "+v"(a0) : "v"(h[0]), "v"(w[0]), "v"(h[1]), "v"(w[1]));
        if (MODE == 1) asm volatile(R8("v_fmac_f32_dpp %0, %1, %2 " QP "\n\t") R8("v_fmac_f32_dpp %0, %3, %4 " QP "\n\t") : "+v"(a0) : "v"(h[0]), "v"(w[0]), "v"(h[1]), "v"(w[1]));
        if (MODE == 2) asm volatile(R8("v_fmac_f32_dpp %0, %2, %3 " QP "\n\tv_fmac_f32_dpp %1, %4, %5 " QP "\n\t") : "+v"(a0), "+v"(a1) : "v"(h[0]), "v"(w[0]), "v"(h[1]), "v"(w[1]));
        if (MODE == 3) asm volatile(R8("v_fma_f32 %0, %4, %5, %0\n\tv_fma_f32 %1, %6, %7, %1\n\t") R8("v_fma_f32 %2, %4, %7, %2\n\tv_fma_f32 %3, %6, %5, %3\n\t")
                                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(h[0]), "v"(w[0]), "v"(h[1]), "v"(w[1]));
        if (MODE == 4) asm volatile(R8("v_pk_fma_f32 %0, %1, %2, %0\n\t") R8("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t") : "+v"(pa) : "v"(pw), "v"(ph));
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + pa.x + pa.y;
    if (threadIdx.x == 0 && blockIdx.x == 7) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}
int main() {
    float *in, *out; unsigned long long *cyc, h[2];
    CK(hipMalloc(&in, 16384 * 4)); CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&cyc, 16));
    CK(hipMemset(in, 0, 16384 * 4));
    const int iters = 4000;
    const char *names[] = {"dep_fmac", "dep_fmac_dpp", "2x_fmac_dpp", "4x_fma", "dep_pk_fma"};
    printf("mode,waves_per_simd,shader_cycles_per_instruction_per_wave,ns_per_instruction_per_wave\n");
    for (int mode = 0; mode < 5; ++mode)
        for (int thr = 256; thr <= 768; thr += 256) {
            for (int rep = 0; rep < 2; ++rep) {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(thr), 0, 0, out, in, iters, cyc); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(thr), 0, 0, out, in, iters, cyc); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(thr), 0, 0, out, in, iters, cyc); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(thr), 0, 0, out, in, iters, cyc); break;
                    default: hipLaunchKernelGGL(k<4>, dim3(256), dim3(thr), 0, 0, out, in, iters, cyc); break;
                }
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
            const double n = iters * (mode == 3 ? 32.0 : 16.0);
            printf("%s,%d,%.2f,%.2f\n", names[mode], thr / 256, (double)h[0] / n, (double)h[1] * 10.0 / n);
        }
    return 0;
}
