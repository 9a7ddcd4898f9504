// Microbenchmark: one W_hh chain pass of ar_xcd.hip's chain waves (csrc/ar_chain.h chain_regs<112>: 112 v_fmac_f32_dpp on pinned
// weights, 7 ds_read_b128 of operands, the 8-chain combine, one LDS write) exactly as the decoder runs it, for 4 slots per round,
// at 1 / 2 / 3 waves per SIMD on every CU -- what a pass costs a wave when it shares its SIMD, with nothing else going on.
//   variant 0: the decoder's pass (one accumulator, slot after slot)
//   variant 1: two slots per pass, their phases interleaved (two accumulators)
//   variant 2: the four slots at once on the matrix pipe: 112 dependent v_mfma_f32_4x4x1_16B_f32 (block = quad: A = the quad's four rows,
//              B = the four slots; lane 4 b + j reads slot j's 112 operands itself: 28 ds_read_b128), four combines
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vectorquantizedcpc_amd/csrc -o /tmp/mb_chain tools/microbench_chain.hip && /tmp/mb_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ar_shared.h"
#include "ar_chain.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NT>
__device__ __forceinline__ void chain_regs2(const float *w, const float *opA, const float *opB, float &accA, float &accB) {
    constexpr int NPH = (NT + 31) / 32;
    float4 cA[2], nA[2], cB[2], nB[2];
    load_phase(opA, 0, 32, cA); load_phase(opB, 0, 32, cB);
    accA = 0.f; accB = 0.f;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        const int len = NT - 32 * ph < 32 ? NT - 32 * ph : 32;
        if (ph + 1 < NPH) {
            const int nlen = NT - 32 * (ph + 1) < 32 ? NT - 32 * (ph + 1) : 32;
            load_phase(opA, ph + 1, nlen, nA); load_phase(opB, ph + 1, nlen, nB);
        }
        const float hA[8] = {cA[0].x, cA[0].y, cA[0].z, cA[0].w, cA[1].x, cA[1].y, cA[1].z, cA[1].w};
        const float hB[8] = {cB[0].x, cB[0].y, cB[0].z, cB[0].w, cB[1].x, cB[1].y, cB[1].z, cB[1].w};
        const float *wp = w + 32 * ph;
        if (len == 32) {
            fmac8<0>(accA, hA, wp); fmac8<0>(accB, hB, wp); fmac8<1>(accA, hA, wp + 8); fmac8<1>(accB, hB, wp + 8);
            fmac8<2>(accA, hA, wp + 16); fmac8<2>(accB, hB, wp + 16); fmac8<3>(accA, hA, wp + 24); fmac8<3>(accB, hB, wp + 24);
        } else {
            fmac4<0>(accA, hA, wp); fmac4<0>(accB, hB, wp); fmac4<1>(accA, hA, wp + 4); fmac4<1>(accB, hB, wp + 4);
            fmac4<2>(accA, hA, wp + 8); fmac4<2>(accB, hB, wp + 8); fmac4<3>(accA, hA, wp + 12); fmac4<3>(accB, hB, wp + 12);
        }
        cA[0] = nA[0]; cA[1] = nA[1]; cB[0] = nB[0]; cB[1] = nB[1];
    }
}

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int MS = 136, MHS = 1092;          // chain stride / slot stride of the MFMA variant's operand copy: lane (kw, j) starts at bank 16 kw + 4 j
__device__ __forceinline__ v4f chain_mfma(const float *w, const float *op) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    float4 cur[2], nxt[2];
    cur[0] = *(const float4 *)op; cur[1] = *(const float4 *)(op + 4);
#pragma unroll
    for (int g = 0; g < 14; ++g) {                                // 14 groups of 8 terms, the next group's operands requested first
        if (g + 1 < 14) { nxt[0] = *(const float4 *)(op + 8 * (g + 1)); nxt[1] = *(const float4 *)(op + 8 * (g + 1) + 4); }
        __builtin_amdgcn_sched_barrier(0);
        const float hv[8] = {cur[0].x, cur[0].y, cur[0].z, cur[0].w, cur[1].x, cur[1].y, cur[1].z, cur[1].w};
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w[8 * g + i], hv[i], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        cur[0] = nxt[0]; cur[1] = nxt[1];
    }
    return acc;
}

template <int VAR>
__global__ void __launch_bounds__(768) k(const float *in, float *out, int rounds, unsigned long long *ticks) {
    __shared__ float hc[4 * MHS];
    __shared__ float gsum[4 * 96];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (unsigned i = tid; i < 4 * MHS; i += blockDim.x) hc[i] = in[i & 4095];
    float w[NT_H];
#pragma unroll
    for (int i = 0; i < NT_H; ++i) w[i] = in[4096 + ((tid * 7 + i) & 4095)];
    const unsigned j = lane & 3u, kw = (lane >> 2) & 3u, R = lane >> 4, c0 = R & 1u, cid = 2 * kw + c0;
    const float *opnd = hc + cid * NT_H + 4 * j;
    const bool sum_lane = kw == 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < rounds; ++r) {
        if (VAR == 0) {
            for (int b = 0; b < 4; ++b) {
                const float acc = chain_regs<NT_H>(w, opnd + b * HR);
                const float v = chain_combine(acc);
                if (sum_lane) gsum[b * 96 + 8 * (wave % 12) + (lane >> 4) * 2 + (lane & 1u)] = v;
            }
        } else if (VAR == 2) {
            const v4f a4 = chain_mfma(w, hc + j * MHS + cid * MS);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = chain_combine(a4[i]);
                if (sum_lane) gsum[j * 96 + 8 * (wave % 12) + (lane >> 5) * 4 + i] = v;      // rows 4 rq + i of slot j (either c0 lane row holds it)
            }
        } else {
            for (int b = 0; b < 4; b += 2) {
                float accA, accB;
                chain_regs2<NT_H>(w, opnd + b * HR, opnd + (b + 1) * HR, accA, accB);
                const float v = chain_combine(accA), v2 = chain_combine(accB);
                if (sum_lane) { gsum[b * 96 + 8 * (wave % 12) + (lane >> 4) * 2 + (lane & 1u)] = v; gsum[(b + 1) * 96 + 8 * (wave % 12) + (lane >> 4) * 2 + (lane & 1u)] = v2; }
            }
        }
        asm volatile("" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) ticks[blockIdx.x * 12 + wave] = t1 - t0;
    __syncthreads();
    out[blockIdx.x * blockDim.x + tid] = gsum[tid % 384];
}

int main() {
    float *in, *out; unsigned long long *ticks;
    CK(hipMalloc(&in, 8192 * 4)); CK(hipMalloc(&out, 256 * 768 * 4)); CK(hipMalloc(&ticks, 256 * 12 * 8));
    CK(hipMemset(in, 0, 8192 * 4));
    const int rounds = 2000;
    static unsigned long long h[256 * 12];
    printf("variant,waves_per_simd,us_per_pass_of_one_slot_mean,slowest_wave,fastest_wave\n");
    for (int var = 0; var < 3; ++var)
        for (int thr = 256; thr <= 768; thr += 256) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(ticks, 0, 256 * 12 * 8));
                if (var == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(thr), 0, 0, in, out, rounds, ticks);
                else if (var == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(thr), 0, 0, in, out, rounds, ticks);
                else hipLaunchKernelGGL(k<2>, dim3(256), dim3(thr), 0, 0, in, out, rounds, ticks);
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost));
            double sum = 0, mx = 0, mn = 1e30; int n = 0;
            for (int b = 0; b < 256; ++b) for (int wv = 0; wv < thr / 64; ++wv) { const double us = h[b * 12 + wv] * 0.01 / (rounds * 4.0); sum += us; ++n; if (us > mx) mx = us; if (us < mn) mn = us; }
            printf("%d,%d,%.3f,%.3f,%.3f\n", var, thr / 256, sum / n, mx, mn);
        }
    return 0;
}
