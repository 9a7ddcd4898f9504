// Microbenchmark: what v_mfma_f32_32x32x2_f32 sustains on MI355X in short and long bursts, and at what
// shader clock -- to price the encoder GEMM (28 us dispatches) honestly.
// hipcc --offload-arch=gfx950 -O3 tools/microbench_mfma.hip -o build/mbm
// Each wave runs `n` MFMAs on `chains` independent accumulators (no memory traffic); wave 0 of every block
// stamps s_memtime (shader cycles) and wall_clock64 (100 MHz) at both ends.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ void k_mfma(int n, float seed, float *sink, unsigned long long *stamps) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    const float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + threadIdx.x * 1e-4f;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < n; i += CH)
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    float s = 0.f;
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float *sink; unsigned long long *st;
    const int maxb = 2048;
    CK(hipMalloc(&sink, 4)); CK(hipMalloc(&st, maxb * 16));
    std::vector<unsigned long long> h(2 * maxb);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("config,us_per_launch,TFLOPs,cycles_per_mfma_per_simd,clock_GHz\n");
    struct Cfg { int blocks, threads, n, chains, reps; };
    const Cfg cfgs[] = {
        {256, 256, 256, 1, 200},  {512, 256, 256, 1, 200},  {256, 512, 256, 1, 200},      // encoder-GEMM-sized bursts (~7-14 us)
        {512, 256, 512, 1, 200},                                                          // = one 512x512x4096 layer's MFMA count
        {256, 256, 8192, 1, 20},  {512, 256, 8192, 1, 20},  {256, 256, 8192, 4, 20},      // long runs
        {512, 256, 65536, 1, 5},                                                          // ~4 ms
    };
    for (const Cfg &c : cfgs) {
        auto launch = [&]() {
            if (c.chains == 1) hipLaunchKernelGGL(k_mfma<1>, dim3(c.blocks), dim3(c.threads), 0, s, c.n, 1.0f, sink, st);
            else hipLaunchKernelGGL(k_mfma<4>, dim3(c.blocks), dim3(c.threads), 0, s, c.n, 1.0f, sink, st);
        };
        for (int i = 0; i < 3; ++i) launch();
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < c.reps; ++i) launch();
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), st, c.blocks * 16, hipMemcpyDeviceToHost));
        double cyc = 0, wall = 0;
        for (int b = 0; b < c.blocks; ++b) { cyc += h[2 * b]; wall += h[2 * b + 1]; }
        const double us = ms * 1e3 / c.reps;
        const double waves_per_simd = (double)c.blocks * (c.threads / 64) / 1024.0;
        const double flop = (double)c.blocks * (c.threads / 64) * c.n * 4096.0;
        printf("blocks=%d threads=%d mfma/wave=%d chains=%d,%.2f,%.1f,%.1f,%.3f\n", c.blocks, c.threads, c.n, c.chains, us,
               flop / us * 1e-6, cyc / c.blocks / (c.n * (waves_per_simd < 1 ? 1 : waves_per_simd)), cyc / wall * 0.1);
    }
    return 0;
}
