// v_mfma_f32_4x4x1_16B_f32 as a batched fp32 fma: what ar_xcd.hip's chain waves need to know before they use it.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb4 tools/microbench_mfma4x4.hip && /tmp/mb4
// (1) operand layout: which lane supplies A / B of which block, where D[i][j] lands;
// (2) D = fmaf(A, B, C) bit for bit (one rounding), also for denormal / huge operands;
// (3) cost of a DEPENDENT chain of 112 of them (the accumulator of one is srcC of the next), 1..3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const float *a, const float *b, const float *c, float *d) {
    const int l = threadIdx.x;
    v4f acc = {c[4 * l], c[4 * l + 1], c[4 * l + 2], c[4 * l + 3]};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[4 * l + i] = acc[i];
}

// B lane-group pattern 1: the upper 32 lanes take their B operand from the lower 32 (blocks b + 8 use B of block b)
__global__ void blgp_kernel(const float *a, const float *b, float *d) {
    const int l = threadIdx.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 1);
    for (int i = 0; i < 4; ++i) d[4 * l + i] = acc[i];
}

template <int N>
__global__ __launch_bounds__(768) void chain_kernel(const float *a, const float *b, float *d, long long *cycles, int reps) {
    const int l = threadIdx.x;
    float w[N], h[8];                  // 112 pinned weights as in the kernel; 8 operand registers (a phase), so that nothing spills under the 168-VGPR cap
    for (int i = 0; i < N; ++i) w[i] = a[(l * 131 + i * 7) % 4096];
    for (int i = 0; i < 8; ++i) h[i] = b[(l * 17 + i * 3) % 4096];
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w[i], h[i & 7], acc, 0, 0, 0);
    }
    const long long t1 = clock64();
    if ((l & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + l / 64] = t1 - t0;
    for (int i = 0; i < 4; ++i) d[(blockIdx.x * blockDim.x + l) * 4 + i] = acc[i];
}

typedef float v4f_ __attribute__((ext_vector_type(4)));
// the same for v_mfma_f32_16x16x4_f32 (the large-batch kernels' instruction): FOUR independent accumulators per wave, as a
// wave of those kernels has, so that the pipe -- not the dependency -- is what is measured
template <int N, int NACC, bool LDSOP>
__global__ __launch_bounds__(768) void chain16_kernel(const float *a, const float *b, float *d, long long *cycles, int reps) {
    __shared__ __attribute__((aligned(16))) float hs[16 * 900];
    const int l = threadIdx.x;
    float w[N], h[8];
    for (int i = 0; i < N; ++i) w[i] = a[(l * 131 + i * 7) % 4096];
    for (int i = 0; i < 8; ++i) h[i] = b[(l * 17 + i * 3) % 4096];
    for (int i = l; i < 16 * 900; i += blockDim.x) hs[i] = b[i % 4096];
    v4f acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    __syncthreads();
    const float4 *hb = (const float4 *)(hs + (l & 15) * 900 + 4 * ((l >> 4) & 3));
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
        if (LDSOP) {                       // as ar_xcm.hip: one 16-byte B fragment read per four MFMAs, two blocks ahead
            float4 q[3];
            q[0] = hb[0]; q[1] = hb[4];
#pragma unroll
            for (int blk = 0; blk < N / 4; ++blk) {
                if (blk + 2 < N / 4) q[(blk + 2) % 3] = hb[4 * (blk + 2)];
                __builtin_amdgcn_sched_barrier(0);
                const float4 hv = q[blk % 3];
                acc[(4 * blk + 0) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[4 * blk + 0], hv.x, acc[(4 * blk + 0) % NACC], 0, 0, 0);
                acc[(4 * blk + 1) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[4 * blk + 1], hv.y, acc[(4 * blk + 1) % NACC], 0, 0, 0);
                acc[(4 * blk + 2) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[4 * blk + 2], hv.z, acc[(4 * blk + 2) % NACC], 0, 0, 0);
                acc[(4 * blk + 3) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[4 * blk + 3], hv.w, acc[(4 * blk + 3) % NACC], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i], h[i & 7], acc[i % NACC], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    if ((l & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + l / 64] = t1 - t0;
    for (int i = 0; i < 4; ++i) d[(blockIdx.x * blockDim.x + l) * 4 + i] = acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i];
}
template <int N>
__global__ __launch_bounds__(768) void chain4_wall_kernel(const float *a, const float *b, float *d, long long *cycles, int reps) {
    const int l = threadIdx.x;
    float w[N], h[8];
    for (int i = 0; i < N; ++i) w[i] = a[(l * 131 + i * 7) % 4096];
    for (int i = 0; i < 8; ++i) h[i] = b[(l * 17 + i * 3) % 4096];
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w[i], h[i & 7], acc, 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    if ((l & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + l / 64] = t1 - t0;
    for (int i = 0; i < 4; ++i) d[(blockIdx.x * blockDim.x + l) * 4 + i] = acc[i];
}

int main() {
    float *a, *b, *c, *d;
    hipMalloc(&a, 4096 * 4); hipMalloc(&b, 4096 * 4); hipMalloc(&c, 4096 * 4); hipMalloc(&d, 256 * 768 * 16);
    std::vector<float> ha(4096), hb(4096), hc(4096), hd(256);
    // ---- (1) layout: A = 1 in ONE lane, B = 1 in ONE lane -> which D entries become 1?
    int a_blk[64], a_row[64], b_blk[64], b_col[64];
    memset(a_blk, -1, sizeof a_blk);
    bool layout_ok = true;
    for (int la = 0; la < 64; la += 1) {
        for (int lb = 4 * (la / 4); lb < 4 * (la / 4) + 4; ++lb) {
            std::fill(ha.begin(), ha.end(), 0.f); std::fill(hb.begin(), hb.end(), 0.f); std::fill(hc.begin(), hc.end(), 0.f);
            ha[la] = 1.f; hb[lb] = 1.f;
            hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
            hipMemcpy(c, hc.data(), 1024, hipMemcpyHostToDevice);
            layout_kernel<<<1, 64>>>(a, b, c, d);
            hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
            int hits = 0, where = -1;
            for (int e = 0; e < 256; ++e) if (hd[e] != 0.f) { ++hits; where = e; }
            // expectation: block = la / 4 = lb / 4; D[row la % 4][col lb % 4] in register (la % 4) of lane lb
            const int want = 4 * lb + (la % 4);
            if (hits != 1 || where != want) { layout_ok = false; printf("A lane %d x B lane %d -> %d hits, entry %d (lane %d reg %d), expected lane %d reg %d\n", la, lb, hits, where, where / 4, where % 4, lb, la % 4); }
        }
    }
    // A in block x, B in block y != x: nothing
    {
        std::fill(ha.begin(), ha.end(), 0.f); std::fill(hb.begin(), hb.end(), 0.f);
        ha[5] = 1.f; hb[9] = 1.f;
        hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
        layout_kernel<<<1, 64>>>(a, b, c, d);
        hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
        for (int e = 0; e < 256; ++e) if (hd[e] != 0.f) { layout_ok = false; printf("cross-block product at entry %d\n", e); }
    }
    printf("layout: lane 4 b + i supplies A_b[i], lane 4 b + j supplies B_b[j], D_b[i][j] = register i of lane 4 b + j: %s\n", layout_ok ? "CONFIRMED" : "NOT AS EXPECTED");
    {
        for (int e = 0; e < 64; ++e) { ha[e] = 1.f + e; hb[e] = 100.f + e; }
        hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
        blgp_kernel<<<1, 64>>>(a, b, d);
        hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
        bool ok = true;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) ok &= hd[4 * l + i] == ha[4 * (l / 4) + i] * hb[l & 31];
        printf("blgp = 1: lanes 32..63 use the B operand of lanes 0..31 (A unchanged): %s\n", ok ? "CONFIRMED" : "NOT AS EXPECTED");
        layout_ok &= ok;
    }
    // ---- (2) exactness against fmaf
    srand(7);
    long bad = 0, n = 0;
    for (int round = 0; round < 200; ++round) {
        for (int e = 0; e < 64; ++e) {
            auto rnd = [&]() {
                const int k = rand() % 10;
                float v = (float)rand() / RAND_MAX * 2.f - 1.f;
                if (k == 0) v *= 1e-38f; if (k == 1) v *= 1e30f; if (k == 2) v *= 1e-20f; if (k == 3) v = 0.f;
                return v;
            };
            ha[e] = rnd(); hb[e] = rnd();
        }
        for (int e = 0; e < 256; ++e) { hc[e] = ((float)rand() / RAND_MAX * 2.f - 1.f) * (rand() % 4 == 0 ? 1e-6f : 1.f); }
        hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(c, hc.data(), 1024, hipMemcpyHostToDevice);
        layout_kernel<<<1, 64>>>(a, b, c, d);
        hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) {
                const float want = fmaf(ha[4 * (l / 4) + i], hb[l], hc[4 * l + i]);
                ++n;
                if (memcmp(&want, &hd[4 * l + i], 4) != 0) { if (bad < 5) printf("mismatch: %a * %a + %a = %a, mfma %a\n", ha[4 * (l / 4) + i], hb[l], hc[4 * l + i], want, hd[4 * l + i]); ++bad; }
            }
    }
    printf("exactness: %ld of %ld results differ from fmaf\n", bad, n);
    // ---- (3) dependent chain of 112
    for (int e = 0; e < 4096; ++e) { ha[e] = (float)rand() / RAND_MAX - 0.5f; hb[e] = (float)rand() / RAND_MAX - 0.5f; }
    hipMemcpy(a, ha.data(), 4096 * 4, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 4096 * 4, hipMemcpyHostToDevice);
    long long *cyc;
    hipMalloc(&cyc, 256 * 12 * 8);
    std::vector<long long> hcyc(256 * 12);
    const int reps = 200;
    for (int waves : {4, 8, 12}) {
        chain_kernel<112><<<256, 64 * waves>>>(a, b, d, cyc, reps);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        chain_kernel<112><<<256, 64 * waves>>>(a, b, d, cyc, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hcyc.data(), cyc, 256 * waves * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256 * waves; ++i) s += hcyc[i];
        printf("chain of 112 dependent mfma 4x4x1, %d waves per CU (%d per SIMD): %.1f shader cycles per instruction per wave, %.3f us per chain (kernel %.3f ms)\n",
               waves, waves / 4, s / (256.0 * waves) / reps / 112, ms * 1e3 / reps, ms);
    }
    // ---- (4) wall clock (100 MHz counter inside the kernel), whole chip (256 workgroups) and one workgroup alone
    for (int grid : {1, 256})
        for (int waves : {4, 8, 12}) {
            for (int which = 0; which < 2; ++which) {
                if (which == 0) chain4_wall_kernel<112><<<grid, 64 * waves>>>(a, b, d, cyc, reps);
                else chain16_kernel<112, 4, false><<<grid, 64 * waves>>>(a, b, d, cyc, reps);
                hipDeviceSynchronize();
                hipMemcpy(hcyc.data(), cyc, grid * waves * 8, hipMemcpyDeviceToHost);
                // The SIMD serves its waves strictly oldest first (see (5)): the pipe's rate is total instructions / time of the LAST
                // wave to finish, not the mean over waves (which counts the early finishers' idle time as throughput).
                double mx = 0; for (int i = 0; i < grid * waves; ++i) mx = hcyc[i] > mx ? hcyc[i] : mx;
                const double ns = mx * 10.0 / reps / 112 / (waves / 4);               // per instruction per SIMD
                printf("%s, %3d workgroup(s) x %2d waves (%d per SIMD): %.2f ns per instruction per SIMD (last wave to finish) = %.1f TFLOP/s on 256 CUs\n",
                       which == 0 ? "4x4x1 (one dependent chain) " : "16x16x4 (4 accumulators)    ", grid, waves, waves / 4, ns,
                       (which == 0 ? 512.0 : 2048.0) / ns * 1024 * 1e-3);
            }
        }
    // ---- (5) what ar_xcm.hip's MFMA phase is made of, 3 waves per SIMD on the whole chip: accumulators per wave (a row's K
    // quarter has two fma chains, a K half four) and B fragments read from LDS
    {
        const int grid = 256, waves = 12;
        auto run = [&](int which, const char *name) {
            if (which == 0) chain16_kernel<112, 4, false><<<grid, 64 * waves>>>(a, b, d, cyc, reps);
            if (which == 1) chain16_kernel<112, 2, false><<<grid, 64 * waves>>>(a, b, d, cyc, reps);
            if (which == 2) chain16_kernel<112, 4, true><<<grid, 64 * waves>>>(a, b, d, cyc, reps);
            if (which == 3) chain16_kernel<112, 2, true><<<grid, 64 * waves>>>(a, b, d, cyc, reps);
            hipDeviceSynchronize();
            hipMemcpy(hcyc.data(), cyc, grid * waves * 8, hipMemcpyDeviceToHost);
            double sm = 0, mx = 0; for (int i = 0; i < grid * waves; ++i) { sm += hcyc[i]; if (hcyc[i] > mx) mx = hcyc[i]; }
            printf("16x16x4, 256 workgroups x 12 waves, %s: last wave done after %.2f us per 112 instructions = %.2f ns per instruction per SIMD (mean finishing time over waves %.2f us)\n",
                   name, mx * 10.0 / reps / 1e3, mx * 10.0 / reps / 112 / 3, sm / (grid * waves) * 10.0 / reps / 1e3);
            if (which == 0) {                // how the time is spread: per workgroup (= CU), the slowest of its 12 waves
                double wmin = 1e30, wmax = 0, hist[8] = {0};
                for (int g = 0; g < grid; ++g) {
                    double m = 0; for (int w2 = 0; w2 < waves; ++w2) m = hcyc[g * waves + w2] > m ? hcyc[g * waves + w2] : m;
                    m = m * 10.0 / reps / 1e3;
                    wmin = m < wmin ? m : wmin; wmax = m > wmax ? m : wmax;
                    hist[g % 8] += m / (grid / 8);
                }
                printf("   slowest wave of a workgroup, us per 112: min %.2f max %.2f over the 256 workgroups; mean by workgroup id %% 8 (= XCD):", wmin, wmax);
                for (int x = 0; x < 8; ++x) printf(" %.2f", hist[x]);
                printf("\n   the 12 waves of workgroup 0:");
                for (int w2 = 0; w2 < waves; ++w2) printf(" %.2f", hcyc[w2] * 10.0 / reps / 1e3);
                printf("\n");
            }
        };
        run(0, "4 accumulators, operands in registers");
        run(1, "2 accumulators, operands in registers");
        run(2, "4 accumulators, B fragments from LDS ");
        run(3, "2 accumulators, B fragments from LDS ");
    }
    return layout_ok && bad == 0 ? 0 : 1;
}
