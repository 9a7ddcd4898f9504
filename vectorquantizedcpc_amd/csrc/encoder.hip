// encoder.hip -- Encoder.encode / Encoder.forward on gfx950 (reference: model.py:43-155).
//
// Every kernel here reproduces the rounding sequence of the reference's PyTorch-CPU path
// (oracle/vqcpc_oracle.c documents each order and tests pin it against the reference):
//   * contractions are fp32 MFMA chains (v_mfma_f32_32x32x2_f32 == a k-ordered fmaf chain),
//     restarted at the K-block boundaries the reference's MKL / oneDNN kernels use;
//   * LayerNorm moments follow ATen's 8-lane Welford cascade;
//   * |x|^2 follows ATen's cascade_sum order; the VQ distance is one fma per (row, code).
// Build with -ffp-contract=off: every fused multiply-add below is written explicitly.
#include "common.h"
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>

// ------------------------------------------------------------------------------------------
// error plumbing + device query
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void vq_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
extern "C" const char *vqcpc_last_error(void) { return g_err; }
extern "C" int vqcpc_abi_version(void) { return VQCPC_ABI_VERSION; }
extern "C" int vqcpc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}
static int require_gfx950() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        vq_set_error("no HIP device: libvqcpc_hip has no CPU fallback");
        return VQCPC_ERR_NO_DEVICE;
    }
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, dev));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        vq_set_error("device %d is %s; this library is built for gfx950 (MI355X) only", dev, p.gcnArchName);
        return VQCPC_ERR_NO_DEVICE;
    }
    return VQCPC_OK;
}
int vq_require_gfx950() { return require_gfx950(); }

// ------------------------------------------------------------------------------------------
// Exact-chain GEMM: 64x64 output tile per 256-thread workgroup, 4 waves as 2x2 of 32x32,
// one v_mfma_f32_32x32x2_f32 per two k.  LDS tiles are k-major ([k][row], stride 65) so the
// MFMA operand reads are conflict-free row-contiguous b32 reads.
// ------------------------------------------------------------------------------------------
#define GT_BM 64
#define GT_BN 64
#define GT_BK 32
#define GT_LD 65

struct GemmP {
    const float *A; int lda;
    const float *W;            // (N, K) row-major
    const float *bias;         // (N) or null
    float *Y; int ldy;
    int M, N, K, KC;
    // im2col source (AMODE 1/2): mel (B, C, T)
    const float *x; int C, T, To;
    // epilogue extras (vq_gemm_chain_ex): ReLU; output row m -> (m / ydiv) * ystride + yoff + m % ydiv, rows whose
    // yoff + m % ydiv >= ylim are not stored (ydiv == 0: row m)
    int relu, ydiv, ystride, yoff, ylim;
};

template <int AMODE>
__device__ __forceinline__ void fetch_a(const GemmP &p, int m0, int k0, int tid, float (&v)[8]) {
    const int row = tid >> 2, kq = (tid & 3) * 8;
    const int m = m0 + row;
    if (AMODE == 0) {
        if (m < p.M) {
            const float4 *src = (const float4 *)(p.A + (size_t)m * p.lda + k0 + kq);
            float4 a = src[0], b = src[1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = 0.f;
        }
    } else {
        int b = 0, tt = 0;
        if (m < p.M) { b = m / p.To; tt = m - b * p.To; }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int kidx = k0 + kq + i;
            int c, tap;
            if (AMODE == 1) { c = kidx >> 2; tap = kidx & 3; }
            else { const int rem = kidx & 63; tap = rem >> 4; c = (kidx >> 6) * 16 + (rem & 15); }
            const int ti = 2 * tt + tap - 1;
            v[i] = (m < p.M && ti >= 0 && ti < p.T) ? p.x[((size_t)b * p.C + c) * p.T + ti] : 0.f;
        }
    }
}
__device__ __forceinline__ void fetch_w(const GemmP &p, int n0, int k0, int tid, float (&v)[8]) {
    const int row = tid >> 2, kq = (tid & 3) * 8;
    const float4 *src = (const float4 *)(p.W + (size_t)(n0 + row) * p.K + k0 + kq);
    float4 a = src[0], b = src[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void stage(float (*T)[GT_LD], int tid, const float (&v)[8]) {
    const int row = tid >> 2, kq = (tid & 3) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) T[kq + i][row] = v[i];
}

__device__ __forceinline__ void gemm_epilogue(const GemmP &p, const f32x16 &tot, int m0, int wm, int half, int col) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m >= p.M) continue;
        size_t row = (size_t)m;
        if (p.ydiv > 0) {
            const int q = m / p.ydiv, t = p.yoff + (m - q * p.ydiv);
            if (t >= p.ylim) continue;
            row = (size_t)q * p.ystride + t;
        }
        const float v = tot[r];
        p.Y[row * p.ldy + col] = (p.relu && v < 0.f) ? 0.f : v;
    }
}

// Software pipeline.  A loop that does, per k tile, {wait for global loads, 16 LDS stores, barrier, 32 LDS
// operand reads} and then its 16 MFMAs pays for both halves: measured on MI355X, 17 us of MFMA + 16 us of
// LDS work for a 4096 x 512 x 512 layer that ran 28.6 us (tools/microbench_mfma.hip).  Here the
// LDS tiles are double-buffered and every MFMA gap carries its share of the other work, in program order
// (pinned with sched_barrier): MFMAs 0-7 of tile k are interleaved with the LDS stores of tile k+1, then the
// global loads of tile k+2 are requested, one barrier, and MFMAs 8-15 are interleaved with the operand
// reads of tile k+1 into a second register set.  22.5 us for the same layer, same k order, same bits.
template <int AMODE, int C>
__device__ __forceinline__ void pipe_body(const GemmP &p, float (*As)[GT_BK][GT_LD], float (*Ws)[GT_BK][GT_LD], int tid,
                                          int m0, int n0, int k2, float (&va)[8], float (&vw)[8], float (&oa)[2][16],
                                          float (&ob)[2][16], f32x16 &acc, int acol, int bcol, int half) {
    const int row = tid >> 2, kq = (tid & 3) * 8;
    float (*An)[GT_LD] = As[C ^ 1];
    float (*Wn)[GT_LD] = Ws[C ^ 1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[C][j], ob[C][j], acc, 0, 0, 0);
        An[kq + j][row] = va[j];
        Wn[kq + j][row] = vw[j];
        __builtin_amdgcn_sched_barrier(0);
    }
    fetch_a<AMODE>(p, m0, k2, tid, va);
    fetch_w(p, n0, k2, tid, vw);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[C][8 + j], ob[C][8 + j], acc, 0, 0, 0);
        oa[C ^ 1][2 * j] = An[4 * j + half][acol];
        oa[C ^ 1][2 * j + 1] = An[4 * j + 2 + half][acol];
        ob[C ^ 1][2 * j] = Wn[4 * j + half][bcol];
        ob[C ^ 1][2 * j + 1] = Wn[4 * j + 2 + half][bcol];
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void fold_chain(const GemmP &p, f32x16 &acc, f32x16 &tot, bool &first, float bv) {
    if (first) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[r] = p.bias ? bv + acc[r] : acc[r];
        first = false;
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[r] = tot[r] + acc[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
}

template <int AMODE>
__global__ __launch_bounds__(256) void gemm_chain_kernel(GemmP p) {
    __shared__ float As[2][GT_BK][GT_LD];
    __shared__ float Ws[2][GT_BK][GT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, li = lane & 31;
    const int n0 = blockIdx.x * GT_BN, m0 = blockIdx.y * GT_BM;
    const int col = n0 + wn * 32 + li, acol = wm * 32 + li, bcol = wn * 32 + li;
    const int nt = p.K / GT_BK, tpb = p.KC / GT_BK;

    f32x16 acc, tot;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; tot[r] = 0.f; }
    bool first = true;
    const float bv = p.bias ? p.bias[col] : 0.f;

    float va[8], vw[8], oa[2][16], ob[2][16];
    fetch_a<AMODE>(p, m0, 0, tid, va);
    fetch_w(p, n0, 0, tid, vw);
    stage(As[0], tid, va);
    stage(Ws[0], tid, vw);
    const int k1 = nt > 1 ? GT_BK : 0;
    fetch_a<AMODE>(p, m0, k1, tid, va);
    fetch_w(p, n0, k1, tid, vw);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        oa[0][kk] = As[0][2 * kk + half][acol];
        ob[0][kk] = Ws[0][2 * kk + half][bcol];
    }

    int k = 0;
    for (; k + 2 < nt; k += 2) {
        pipe_body<AMODE, 0>(p, As, Ws, tid, m0, n0, (k + 2) * GT_BK, va, vw, oa, ob, acc, acol, bcol, half);
        if ((k + 1) % tpb == 0) fold_chain(p, acc, tot, first, bv);
        pipe_body<AMODE, 1>(p, As, Ws, tid, m0, n0, (k + 3 < nt ? k + 3 : nt - 1) * GT_BK, va, vw, oa, ob, acc, acol, bcol, half);
        if ((k + 2) % tpb == 0) fold_chain(p, acc, tot, first, bv);
    }
    if (k + 1 < nt) {                                                 // two tiles left
        pipe_body<AMODE, 0>(p, As, Ws, tid, m0, n0, (nt - 1) * GT_BK, va, vw, oa, ob, acc, acol, bcol, half);
        if ((k + 1) % tpb == 0) fold_chain(p, acc, tot, first, bv);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[1][kk], ob[1][kk], acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[0][kk], ob[0][kk], acc, 0, 0, 0);
    }
    fold_chain(p, acc, tot, first, bv);

    gemm_epilogue(p, tot, m0, wm, half, col);
}

template <int AMODE>
static int launch_gemm(const GemmP &p, hipStream_t s) {
    VQ_REQUIRE(p.N % GT_BN == 0 && p.K % GT_BK == 0 && p.KC % GT_BK == 0 && p.M > 0,
               "gemm_chain: unsupported shape M=%d N=%d K=%d KC=%d", p.M, p.N, p.K, p.KC);
    dim3 grid(p.N / GT_BN, (p.M + GT_BM - 1) / GT_BM);
    hipLaunchKernelGGL((gemm_chain_kernel<AMODE>), grid, dim3(256), 0, s, p);
    HIP_TRY(hipGetLastError());
    return VQCPC_OK;
}

int vq_gemm_chain(const float *A, int lda, const float *W, const float *bias, float *Y, int ldy,
                  int M, int N, int K, int KC, hipStream_t s) {
    VQ_REQUIRE(lda % 4 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
               "gemm_chain: operands must be 16-byte aligned");
    GemmP p{};
    p.A = A; p.lda = lda; p.W = W; p.bias = bias; p.Y = Y; p.ldy = ldy;
    p.M = M; p.N = N; p.K = K; p.KC = KC;
    return launch_gemm<0>(p, s);
}

int vq_gemm_chain_ex(const float *A, int lda, const float *W, const float *bias, float *Y, int ldy,
                     int M, int N, int K, int KC, int relu, int ydiv, int ystride, int yoff, int ylim, hipStream_t s) {
    VQ_REQUIRE(lda % 4 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
               "gemm_chain: operands must be 16-byte aligned");
    GemmP p{};
    p.A = A; p.lda = lda; p.W = W; p.bias = bias; p.Y = Y; p.ldy = ldy;
    p.M = M; p.N = N; p.K = K; p.KC = KC;
    p.relu = relu; p.ydiv = ydiv; p.ystride = ystride; p.yoff = yoff; p.ylim = ylim;
    return launch_gemm<0>(p, s);
}

// ------------------------------------------------------------------------------------------
// LayerNorm(512) + optional ReLU, ATen CPU order (see oracle orc_ln_moments).
// One half-wave (32 lanes) per row: lane = chunk*8 + l runs the Welford chain of vector lane
// l over chunk `chunk` (16 vectors of 8), merges by shuffles, then all lanes normalise.
// ------------------------------------------------------------------------------------------
struct LnConst { float inv[16]; float sc[8]; };

__global__ __launch_bounds__(256) void ln512_kernel(const float *__restrict__ X, const float *__restrict__ g,
                                                    const float *__restrict__ b, float *__restrict__ Y, int M,
                                                    float eps, int relu, LnConst k) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, ll = lane & 31;
    const int row = (blockIdx.x * 4 + wave) * 2 + half;
    const int rr = row < M ? row : M - 1;
    const float *x = X + (size_t)rr * 512;
    const int chunk = ll >> 3, l = ll & 7;
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float xv = x[(chunk * 16 + j) * 8 + l];
        const float d = xv - m1;
        m1 = __builtin_fmaf(d, k.inv[j], m1);
        m2 = __builtin_fmaf(d, xv - m1, m2);
    }
    // chunk 1 -> chunk 0, chunk 3 -> chunk 2   (AddMomentsVec, 16 + 16 vectors, c = 1/2)
    float a1 = __shfl_down(m1, 8), a2 = __shfl_down(m2, 8);
    {
        const float delta = a1 - m1;
        const float n1 = __builtin_fmaf(0.5f, delta, m1);
        m2 = __builtin_fmaf((0.5f * 16.0f) * delta, delta, m2 + a2);
        m1 = n1;
    }
    // (chunks 2,3) -> (chunks 0,1)              (32 + 32 vectors)
    a1 = __shfl_down(m1, 16); a2 = __shfl_down(m2, 16);
    {
        const float delta = a1 - m1;
        const float n1 = __builtin_fmaf(0.5f, delta, m1);
        m2 = __builtin_fmaf((0.5f * 32.0f) * delta, delta, m2 + a2);
        m1 = n1;
    }
    // the 8 vector lanes, serially (AddMoments, scalar)
    float M1 = 0.f, M2 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float s1 = __shfl(m1, half * 32 + q), s2 = __shfl(m2, half * 32 + q);
        const float delta = s1 - M1;
        M1 = __builtin_fmaf(k.sc[q], delta, M1);
        M2 = M2 + __builtin_fmaf((delta * delta) * k.sc[q], (float)(64 * q), s2);
    }
    const float mean = M1, var = M2 / 512.0f;
    // 1 / sqrt(var + eps), both correctly rounded in fp32 (via fp64: innocuous double rounding)
    const float sd = (float)sqrt((double)(var + eps));
    const float rstd = (float)(1.0 / (double)sd);
    if (row < M) {
        float *y = Y + (size_t)row * 512;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = q * 128 + ll * 4;
            const float4 xv = *(const float4 *)(x + e);
            const float4 gv = *(const float4 *)(g + e);
            const float4 bv = *(const float4 *)(b + e);
            float4 o;
            o.x = __builtin_fmaf((xv.x - mean) * rstd, gv.x, bv.x);
            o.y = __builtin_fmaf((xv.y - mean) * rstd, gv.y, bv.y);
            o.z = __builtin_fmaf((xv.z - mean) * rstd, gv.z, bv.z);
            o.w = __builtin_fmaf((xv.w - mean) * rstd, gv.w, bv.w);
            if (relu) {
                o.x = o.x < 0.f ? 0.f : o.x; o.y = o.y < 0.f ? 0.f : o.y;
                o.z = o.z < 0.f ? 0.f : o.z; o.w = o.w < 0.f ? 0.f : o.w;
            }
            *(float4 *)(y + e) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------
// torch.sum(v**2, dim=1) over 64 floats, ATen cascade order (oracle orc_sumsq64).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float sumsq64(const float *v) {
    float sq[64];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float4 t = *(const float4 *)(v + 4 * i);
        sq[4 * i + 0] = t.x * t.x; sq[4 * i + 1] = t.y * t.y; sq[4 * i + 2] = t.z * t.z; sq[4 * i + 3] = t.w * t.w;
    }
    float acc = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const float q0 = sq[l] + sq[32 + l], q1 = sq[8 + l] + sq[40 + l];
        const float q2 = sq[16 + l] + sq[48 + l], q3 = sq[24 + l] + sq[56 + l];
        acc += ((q0 + q1) + q2) + q3;
    }
    return acc;
}
__global__ void rowsumsq64_kernel(const float *__restrict__ X, float *__restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sumsq64(X + (size_t)i * 64);
}

// ------------------------------------------------------------------------------------------
// VQEmbeddingEMA.encode (model.py:103-115) in one kernel: |x|^2, the distance matrix row block,
// first-index argmin and the F.embedding gather.  One 256-thread workgroup per 16 rows; each wave owns a
// quarter of the codes and runs them as 16x16 tiles on v_mfma_f32_16x16x4_f32 with ONE accumulator per
// tile, so a dot product is the k-ascending fmaf chain of the reference's addmm (K = 64 is a single MKL
// block); two tiles are interleaved to cover the MFMA's dependent latency.
// Codebook fragments: Ef[(t*4 + q)*64 + lane] (float4) = E[16t + (lane&15)][16q + 4c + (lane>>4)], c = .x.y.z.w
// ------------------------------------------------------------------------------------------
__global__ void vq_build_frag_kernel(const float *__restrict__ E, float4 *__restrict__ Ef, int n_emb) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_emb * 16) return;
    const int lane = id & 63, q = (id >> 6) & 3, t = id >> 8;
    const float *r = E + (size_t)(16 * t + (lane & 15)) * 64 + 16 * q + (lane >> 4);
    Ef[id] = make_float4(r[0], r[4], r[8], r[12]);
}

__device__ __forceinline__ void vq_load_tile(const float4 *__restrict__ Ef, int t, int lane, float4 (&f)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) f[q] = Ef[(size_t)(t * 4 + q) * 64 + lane];
}
__device__ __forceinline__ void vq_take(const f32x4 &acc, int code, float e2, const float (&x2)[4], float (&bd)[4], int (&bj)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float d = __builtin_fmaf(-2.0f, acc[r], e2 + x2[r]);
        if (d < bd[r]) { bd[r] = d; bj[r] = code; }
    }
}

// Shared tail of the VQ search: xs / x2s hold the 16 rows (and their |x|^2) of the block starting at row r0.
struct VqSmem {
    float xs[16][68];
    float x2s[16];
    float cd[8][16];
    int cj[8][16];
    int best[16];
};
__device__ __forceinline__ void vq_rows16(VqSmem &sm, int r0, int N, const float4 *__restrict__ Ef, const float *__restrict__ E,
                                          const float *__restrict__ e2, int n_emb, int64_t *__restrict__ idx,
                                          float *__restrict__ zq, int tid, const float4 (&f0_in)[4], const float4 (&f1_in)[4],
                                          int nwv = 4) {
    // nwv = waves that search (4, or 8 of a 512-thread workgroup when the tiles divide); waves beyond only keep the two
    // barriers company
    const bool active = (tid >> 6) < nwv;                              // wave-uniform
    const int lane = tid & 63, wave = active ? tid >> 6 : 0;
    const int tpw = active ? n_emb / (16 * nwv) : 0, t0 = wave * tpw;  // 16-code tiles per wave, this wave's first
    float4 f0[4], f1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { f0[q] = f0_in[q]; f1[q] = f1_in[q]; }
    float a[16], x2[4], bd[4];
    int bj[4];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = sm.xs[lane & 15][4 * j + (lane >> 4)];
#pragma unroll
    for (int r = 0; r < 4; ++r) { x2[r] = sm.x2s[4 * (lane >> 4) + r]; bd[r] = __builtin_inff(); bj[r] = 0; }

    int tt = 0;
    for (; tt + 1 < tpw; tt += 2) {
        const int t = t0 + tt, c0 = 16 * t + (lane & 15);
        const float ea = e2[c0], eb = e2[c0 + 16];
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q + 0], f0[q].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q + 0], f1[q].x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q + 1], f0[q].y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q + 1], f1[q].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q + 2], f0[q].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q + 2], f1[q].z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q + 3], f0[q].w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q + 3], f1[q].w, acc1, 0, 0, 0);
        }
        if (tt + 2 < tpw) vq_load_tile(Ef, t + 2, lane, f0);           // next pair streams in under the compares
        if (tt + 3 < tpw) vq_load_tile(Ef, t + 3, lane, f1);
        vq_take(acc0, c0, ea, x2, bd, bj);
        vq_take(acc1, c0 + 16, eb, x2, bd, bj);
    }
    if (tt < tpw) {                                                    // odd tile count: last tile alone
        const int c0 = 16 * (t0 + tt) + (lane & 15);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q + 0], f0[q].x, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q + 1], f0[q].y, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q + 2], f0[q].z, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * q + 3], f0[q].w, acc0, 0, 0, 0);
        }
        vq_take(acc0, c0, e2[c0], x2, bd, bj);
    }
    // first-index argmin: over the 16 codes of a lane group, then over the four waves (ascending codes).  The lexicographic
    // (distance, index) minimum is idempotent, so four DPP exchanges inside the row of 16 lanes (pairs, quads, the mirrored
    // half, the mirrored row) leave it in every lane -- no ds_bpermute round trips.
#define VQ_DPP_MIN(ctrl)                                                                                        \
    {                                                                                                           \
        const float od = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bd[r]), (ctrl), 0xF, 0xF, false)); \
        const int oj = __builtin_amdgcn_update_dpp(0, bj[r], (ctrl), 0xF, 0xF, false);                          \
        if (od < bd[r] || (od == bd[r] && oj < bj[r])) { bd[r] = od; bj[r] = oj; }                              \
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        VQ_DPP_MIN(0xB1)            // quad_perm [1,0,3,2]
        VQ_DPP_MIN(0x4E)            // quad_perm [2,3,0,1]
        VQ_DPP_MIN(0x141)           // row_half_mirror
        VQ_DPP_MIN(0x140)           // row_mirror
        if (active && (lane & 15) == 0) { sm.cd[wave][4 * (lane >> 4) + r] = bd[r]; sm.cj[wave][4 * (lane >> 4) + r] = bj[r]; }
    }
#undef VQ_DPP_MIN
    __syncthreads();
    if (tid < 16) {
        float d = sm.cd[0][tid];
        int j = sm.cj[0][tid];
        for (int w = 1; w < nwv; ++w)
            if (sm.cd[w][tid] < d) { d = sm.cd[w][tid]; j = sm.cj[w][tid]; }
        sm.best[tid] = j;
        if (r0 + tid < N) idx[r0 + tid] = j;
    }
    __syncthreads();
    if (tid < 256) {
        const int row = tid >> 4, c4 = tid & 15;                       // F.embedding gather (model.py:113)
        if (r0 + row < N) ((float4 *)zq)[(size_t)(r0 + row) * 16 + c4] = ((const float4 *)E)[(size_t)sm.best[row] * 16 + c4];
    }
}

__global__ __launch_bounds__(256) void vq_encode_kernel(const float *__restrict__ X, int N, const float4 *__restrict__ Ef,
                                                        const float *__restrict__ E, const float *__restrict__ e2, int n_emb,
                                                        int64_t *__restrict__ idx, float *__restrict__ zq) {
    __shared__ VqSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r0 = blockIdx.x * 16;
    {
        const int row = tid >> 4, c4 = tid & 15;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + row < N) v = ((const float4 *)X)[(size_t)(r0 + row) * 16 + c4];
        sm.xs[row][4 * c4 + 0] = v.x; sm.xs[row][4 * c4 + 1] = v.y; sm.xs[row][4 * c4 + 2] = v.z; sm.xs[row][4 * c4 + 3] = v.w;
    }
    if (tid < 16) sm.x2s[tid] = r0 + tid < N ? sumsq64(X + (size_t)(r0 + tid) * 64) : 0.f;
    const int tpw = n_emb / 64, t0 = wave * tpw;
    float4 f0[4], f1[4];
    vq_load_tile(Ef, t0, lane, f0);
    if (tpw > 1) vq_load_tile(Ef, t0 + 1, lane, f1);
    else vq_load_tile(Ef, t0, lane, f1);
    __syncthreads();
    vq_rows16(sm, r0, N, Ef, E, e2, n_emb, idx, zq, tid, f0, f1);
}

// ------------------------------------------------------------------------------------------
// Fused front end: conv -> LN/ReLU -> 4 x (FC -> LN/ReLU) -> FC 512->64 -> VQ in ONE launch, 16 rows per workgroup.
//
// The layered path above moves every activation through HBM ten times and spends 22 % of a C2 call in stand-alone
// LayerNorm passes (DESIGN "Encoder").  Here a 256-thread workgroup owns 16 whole rows: the activation tile
// (16 x 512 fp32) lives in LDS from the im2col gather to the VQ search, LayerNorm runs on it in place (the same
// half-wave-per-row Welford cascade as ln512_kernel, so the same bits), and each wave multiplies it with 128 output
// columns at a time on v_mfma_f32_16x16x4_f32: weights are pre-arranged in fragment order
//   Wf[(ct * K/16 + q) * 64 + lane] (float4) = W[16 ct + (lane & 15)][16 q + 4 c + (lane >> 4)], c = .x .y .z .w
// and go from L2 straight to registers (16 B per lane per 4 MFMAs, requested 3 q-steps ahead); one MFMA covers 4
// consecutive k, one accumulator per 16 x 16 tile, so a dot product is the same k-ascending fmaf chain -- restarted at
// the reference's K-block boundaries -- that gemm_chain_kernel runs on 32x32x2 tiles.  Same bits, one launch, no
// activation traffic; per workgroup the stream is all 5 MB of weights (L2-resident: every workgroup reads the same).
// ------------------------------------------------------------------------------------------
#define FE_LD 514                       // LDS row stride of the activation tile: bank = 2 row + k, conflict-free A reads
#define FE_WIN_C 128                    // channels of the staged mel window (the fused schedule runs for 4 C <= 512)

__global__ void frag16_build_kernel(const float *__restrict__ W, int N, int K, float4 *__restrict__ Wf) {
    const int nq = K / 16;
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)(N / 16) * nq * 64) return;
    const int lane = (int)(id & 63);
    const int q = (int)((id >> 6) % nq), ct = (int)((id >> 6) / nq);
    const float *r = W + (size_t)(16 * ct + (lane & 15)) * K + 16 * q + (lane >> 4);
    Wf[id] = make_float4(r[0], r[4], r[8], r[12]);
}

struct FusedP {
    const float *mel; int C, T, To, N;
    int conv_mode;                      // 1 im2col order (one chain), 2 direct order (chain restarted every 64 k)
    const float4 *conv_f;               // conv weight fragments in that order: [32 ct][K/16][64]
    const float *ln_g[5], *ln_b[5];
    const float4 *fc_f[4];              // [32 ct][32][64]
    const float4 *out_f; const float *out_b;   // [4 ct][32][64]
    const float4 *Ef; const float *E, *e2; int n_emb;
    float *z_pre; float *z_q; int64_t *idx;
    float *stage_out; int stage;        // stage dump (vqcpc_encoder_stage): -1 = none
    float eps; LnConst lnc;
};

// One GEMM stage for this wave: tile (LDS, 16 rows x K) x NT column tiles starting at ct0.  tot = fold over K blocks of
// kcq q-steps (16 k each) of zero-started chains: first block (bias ? bias + chain : chain), later blocks tot + chain.
// The first D-1 q-steps' fragments of a stage, requested ahead of time (under the previous stage's epilogue and
// LayerNorm, when the MFMA pipe would otherwise wait for the first bytes of the next weight matrix).
// D = depth of the fragment ring (a power of two): fragments are requested D-1 q-steps (of 0.43 us of MFMAs at NT = 8)
// before their MFMAs.  The weight stream is latency x bytes-in-flight bound: with D = 4 a wave has 24 KB in flight and
// a workgroup pulls 44 GB/s -- alone on the chip as well as among 255 others -- against the 75 GB/s its MFMAs consume.
// q_off / nq_all: the stage covers q-steps [q_off, q_off + nq) of a fragment array holding nq_all per column tile (0: nq).
template <int NT, int D>
__device__ __forceinline__ void rows16_prefetch(const float4 *__restrict__ Wf, int ct0, int nq, float4 (&fr)[D][NT], int lane,
                                                int q_off = 0, int nq_all = 0) {
    const int qs = nq_all ? nq_all : nq;
#pragma unroll
    for (int u = 0; u < D - 1; ++u)
#pragma unroll
        for (int j = 0; j < NT; ++j) fr[u][j] = Wf[(((size_t)(ct0 + j) * qs) + q_off + (u < nq ? u : nq - 1)) * 64 + lane];
}

template <int NT, int D>
__device__ __forceinline__ void rows16_gemm(const float *tile, const float4 *__restrict__ Wf, int ct0, int nq, int kcq,
                                            const float *__restrict__ bias, f32x4 (&tot)[NT], int lane, float4 (&fr)[D][NT],
                                            int q_off = 0, int nq_all = 0) {
    const float4 *wp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wp[j] = Wf + ((size_t)(ct0 + j) * (nq_all ? nq_all : nq) + q_off) * 64 + lane;
    const float *arow = tile + (lane & 15) * FE_LD + (lane >> 4) + 16 * q_off;
    float an[4] = {arow[0], arow[4], arow[8], arow[12]};
    f32x4 acc[NT];
    float bv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        tot[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bv[j] = bias ? bias[16 * (ct0 + j) + (lane & 15)] : 0.f;
    }
    bool first = true;
    for (int q0 = 0; q0 < nq; q0 += D) {                             // nq is a multiple of 4; D = 8 walks it in halves
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int q = q0 + u;
            if (D > 4 && q >= nq) break;                             // wave-uniform
            const int qp = q + D - 1 < nq ? q + D - 1 : nq - 1;
#pragma unroll
            for (int j = 0; j < NT; ++j) fr[(u + D - 1) & (D - 1)][j] = wp[j][(size_t)qp * 64];
            const float a0 = an[0], a1 = an[1], a2 = an[2], a3 = an[3];
            const int qn = q + 1 < nq ? q + 1 : q;
            an[0] = arow[16 * qn]; an[1] = arow[16 * qn + 4]; an[2] = arow[16 * qn + 8]; an[3] = arow[16 * qn + 12];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, fr[u][j].x, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, fr[u][j].y, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, fr[u][j].z, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, fr[u][j].w, acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if ((u & 3) == 3 && (q + 1) % kcq == 0) {                // end of a K block (kcq is a multiple of 4)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        tot[j][r] = first ? (bias ? bv[j] + acc[j][r] : acc[j][r]) : tot[j][r] + acc[j][r];
                        acc[j][r] = 0.f;
                    }
                }
                first = false;
            }
        }
    }
}

// D layout of the 16x16x4 MFMA: register r of a lane = row 4 (lane >> 4) + r, column lane & 15.
template <int NT>
__device__ __forceinline__ void rows16_store(float *tile, const f32x4 (&tot)[NT], int ct0, int lane) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[(4 * (lane >> 4) + r) * FE_LD + 16 * (ct0 + j) + (lane & 15)] = tot[j][r];
}

// LayerNorm(512) + ReLU of the tile in place: the half-wave-per-row procedure of ln512_kernel (ATen's order).  A half-wave
// owns rows r and r + 8 and runs their two (serial, latency-bound) moment cascades interleaved; it then normalises both
// (R = 2, 256-thread workgroups); with 512 threads every half-wave has one row (R = 1).
template <int R>
__device__ __forceinline__ void rows16_layernorm(float *tile, const float *__restrict__ g, const float *__restrict__ b,
                                                 float eps, const LnConst &k, int tid) {
    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, ll = lane & 31;
    const int chunk = ll >> 3, l = ll & 7;
    float *x[2] = {tile + (wave * 2 + half) * FE_LD, tile + ((R == 2 ? 8 : 0) + wave * 2 + half) * FE_LD};
    float m1[2] = {0.f, 0.f}, m2[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float xv = x[r][(chunk * 16 + j) * 8 + l];
            const float d = xv - m1[r];
            m1[r] = __builtin_fmaf(d, k.inv[j], m1[r]);
            m2[r] = __builtin_fmaf(d, xv - m1[r], m2[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {                                       // chunk 1 -> 0, 3 -> 2 (16 + 16 vectors)
        const float a1 = __shfl_down(m1[r], 8), a2 = __shfl_down(m2[r], 8);
        const float delta = a1 - m1[r];
        const float n1 = __builtin_fmaf(0.5f, delta, m1[r]);
        m2[r] = __builtin_fmaf((0.5f * 16.0f) * delta, delta, m2[r] + a2);
        m1[r] = n1;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {                                       // (chunks 2, 3) -> (chunks 0, 1) (32 + 32 vectors)
        const float a1 = __shfl_down(m1[r], 16), a2 = __shfl_down(m2[r], 16);
        const float delta = a1 - m1[r];
        const float n1 = __builtin_fmaf(0.5f, delta, m1[r]);
        m2[r] = __builtin_fmaf((0.5f * 32.0f) * delta, delta, m2[r] + a2);
        m1[r] = n1;
    }
    float M1[2] = {0.f, 0.f}, M2[2] = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 8; ++q) {                                       // the 8 vector lanes, serially
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float s1 = __shfl(m1[r], half * 32 + q), s2 = __shfl(m2[r], half * 32 + q);
            const float delta = s1 - M1[r];
            M1[r] = __builtin_fmaf(k.sc[q], delta, M1[r]);
            M2[r] = M2[r] + __builtin_fmaf((delta * delta) * k.sc[q], (float)(64 * q), s2);
        }
    }
    float mean[2], rstd[2];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mean[r] = M1[r];
        const float var = M2[r] / 512.0f;
        const float sd = (float)sqrt((double)(var + eps));              // both correctly rounded in fp32 (via fp64)
        rstd[r] = (float)(1.0 / (double)sd);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = q * 64 + ll * 2;                                  // 8-byte LDS accesses (the row stride is 8 mod 16)
        const float2 gv = *(const float2 *)(g + e);
        const float2 bb = *(const float2 *)(b + e);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float2 xv = *(const float2 *)(x[r] + e);
            float2 o;
            o.x = __builtin_fmaf((xv.x - mean[r]) * rstd[r], gv.x, bb.x);
            o.y = __builtin_fmaf((xv.y - mean[r]) * rstd[r], gv.y, bb.y);
            o.x = o.x < 0.f ? 0.f : o.x;
            o.y = o.y < 0.f ? 0.f : o.y;
            *(float2 *)(x[r] + e) = o;
        }
    }
}

__device__ __forceinline__ bool rows16_dump(const FusedP &p, const float *tile, int stage, int r0, int tid) {
    if (p.stage != stage) return false;
    for (int e = tid; e < 16 * 512; e += (int)blockDim.x) {
        const int row = e >> 9, col = e & 511;
        if (r0 + row < p.N) p.stage_out[(size_t)(r0 + row) * 512 + col] = tile[row * FE_LD + col];
    }
    return true;
}

// 512 threads: eight waves of four column tiles, TWO per SIMD.  With one wave per SIMD (256 threads x eight tiles) the MFMA
// pipe idles while that wave requests the next fragments, reads its A operands and folds K blocks -- each Linear took 18.5 us
// for 13.7 us of MFMAs (tools/encoder_stage_times.py); a second wave fills those gaps, and LayerNorm has a half-wave per row.
__global__ __launch_bounds__(512) void enc_fused_kernel(FusedP p) {
    __shared__ __attribute__((aligned(16))) float tile[16 * FE_LD];
    __shared__ VqSmem sm;
    __shared__ float mel_win[FE_WIN_C * 36];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r0 = blockIdx.x * 16;
    const int K0 = 4 * p.C;
    constexpr int NT = 4, RD = 4;                        // column tiles per wave; fragment ring: 3 q-steps ahead (a ring of 8 --
    f32x4 tot[NT];                                       // 7 ahead -- measured the same: the stream is not what the MFMAs wait for)
    float4 fr[RD][NT];
    rows16_prefetch<NT, RD>(p.conv_f, NT * wave, K0 / 16, fr, lane);      // the first conv fragments fly under the gather

    // ---- im2col of the 16 rows (model.py:65), k order of the reference back end (fetch_a).  When the 16 rows are consecutive
    // frames of ONE utterance (always at configs[1]: 64 frames per utterance) their taps are a window of 34 mel frames per
    // channel: it is loaded ONCE, coalesced along T (136 contiguous bytes per channel instead of 16 x 4 scalar loads 8 bytes
    // apart), into LDS and the tile is built from there.  A tile that straddles two utterances (or the end) gathers from global.
    const int b0 = r0 / p.To, tt0 = r0 - b0 * p.To;
    const bool one_utt = r0 + 15 < p.N && tt0 + 15 < p.To && p.C <= FE_WIN_C;
    if (one_utt) {
        float *win = mel_win;                                 // [C][36]: frames 2 tt0 - 1 ... 2 tt0 + 32 (zero outside the utterance)
        const float *src = p.mel + (size_t)b0 * p.C * p.T;
        const int ti0 = 2 * tt0 - 1;
        for (int e = tid; e < p.C * 36; e += 512) {
            const int c = e / 36, o = e - c * 36, ti = ti0 + o;
            win[e] = (o < 34 && ti >= 0 && ti < p.T) ? src[(size_t)c * p.T + ti] : 0.f;
        }
        __syncthreads();
        for (int e = tid; e < 16 * K0; e += 512) {
            const int i = e & 15, kidx = e >> 4;
            int c, tap;
            if (p.conv_mode == 1) { c = kidx >> 2; tap = kidx & 3; }
            else { const int rem = kidx & 63; tap = rem >> 4; c = (kidx >> 6) * 16 + (rem & 15); }
            tile[i * FE_LD + kidx] = win[c * 36 + 2 * i + tap];
        }
    } else {
        for (int e = tid; e < 16 * K0; e += 512) {
            const int i = e & 15, kidx = e >> 4, m = r0 + i;
            int c, tap;
            if (p.conv_mode == 1) { c = kidx >> 2; tap = kidx & 3; }
            else { const int rem = kidx & 63; tap = rem >> 4; c = (kidx >> 6) * 16 + (rem & 15); }
            float v = 0.f;
            if (m < p.N) {
                const int b = m / p.To, tt = m - b * p.To, ti = 2 * tt + tap - 1;
                if (ti >= 0 && ti < p.T) v = p.mel[((size_t)b * p.C + c) * p.T + ti];
            }
            tile[i * FE_LD + kidx] = v;
        }
    }
    __syncthreads();

    rows16_gemm<NT, RD>(tile, p.conv_f, NT * wave, K0 / 16, p.conv_mode == 1 ? K0 / 16 : 4, nullptr, tot, lane, fr);
    if (p.stage != 0) rows16_prefetch<NT, RD>(p.fc_f[0], NT * wave, 32, fr, lane);    // under the store + LayerNorm below
    __syncthreads();                                     // every wave has read its A operands
    rows16_store<NT>(tile, tot, NT * wave, lane);
    __syncthreads();
    if (rows16_dump(p, tile, 0, r0, tid)) return;

    // ---- seg-FC stack (model.py:46-55)
    rows16_layernorm<1>(tile, p.ln_g[0], p.ln_b[0], p.eps, p.lnc, tid);
    __syncthreads();
    if (rows16_dump(p, tile, 1, r0, tid)) return;
    for (int l = 0; l < 4; ++l) {
        rows16_gemm<NT, RD>(tile, p.fc_f[l], NT * wave, 32, 16, nullptr, tot, lane, fr);
        if (l < 3) rows16_prefetch<NT, RD>(p.fc_f[l + 1], NT * wave, 32, fr, lane);
        __syncthreads();
        rows16_store<NT>(tile, tot, NT * wave, lane);
        __syncthreads();
        if (rows16_dump(p, tile, 2 + 2 * l, r0, tid)) return;
        rows16_layernorm<1>(tile, p.ln_g[l + 1], p.ln_b[l + 1], p.eps, p.lnc, tid);
        __syncthreads();
        if (rows16_dump(p, tile, 3 + 2 * l, r0, tid)) return;
    }

    // ---- encoder.14: 512 -> 64 with bias: four 16-column tiles.  The reference's fold (bias + c0) + c1 has two independent
    // zero-started chains (k < 256, k >= 256): wave w runs c0 of tile w, wave w + 4 runs c1 -- half the dependent MFMA chain
    // each -- and hands it over through LDS.  The VQ codebook tiles stream in underneath.
    const int ctw = wave & 3, kh = wave >> 2;
    const int nwv = p.n_emb % 128 == 0 ? 8 : 4;          // waves of the VQ search
    const int tpw = p.n_emb / (16 * nwv), t0 = (wave < nwv ? wave : 0) * tpw;
    float4 f0[4] = {}, f1[4] = {};
    if (wave < nwv) {
        vq_load_tile(p.Ef, t0, lane, f0);
        vq_load_tile(p.Ef, tpw > 1 ? t0 + 1 : t0, lane, f1);
    }
    f32x4 zt[1];
    float4 fr1[4][1];
    rows16_prefetch<1, 4>(p.out_f, ctw, 16, fr1, lane, 16 * kh, 32);
    rows16_gemm<1, 4>(tile, p.out_f, ctw, 16, 16, kh == 0 ? p.out_b : nullptr, zt, lane, fr1, 16 * kh, 32);
    float (*c1s)[68] = (float (*)[68])tile;              // the activation tile is dead once every wave is past its chain
    __syncthreads();
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) c1s[4 * (lane >> 4) + r][16 * ctw + (lane & 15)] = zt[0][r];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (lane >> 4) + r, col = 16 * ctw + (lane & 15);
            const float z = zt[0][r] + c1s[row][col];
            sm.xs[row][col] = z;
            if (p.z_pre && r0 + row < p.N) p.z_pre[(size_t)(r0 + row) * 64 + col] = z;
        }
    }
    __syncthreads();
    if (p.stage == 10) return;                           // z_pre is the stage output
    // ---- VQ (model.py:103-115)
    if (tid < 16) sm.x2s[tid] = r0 + tid < p.N ? sumsq64(&sm.xs[tid][0]) : 0.f;
    __syncthreads();
    vq_rows16(sm, r0, p.N, p.Ef, p.E, p.e2, p.n_emb, p.idx, p.z_q, tid, f0, f1, nwv);
}

// ------------------------------------------------------------------------------------------
// Small calls (a single utterance: BASELINE configs[0], encode.py:44-46): too few 16-row tiles to fill the chip with
// whole-row workgroups, and each of those would still stream all 5 MB of weights.  Here the columns are split over
// several workgroups per row tile and one launch covers one Linear:
//   launch 0: im2col + conv (8 column groups: 4 waves x one 16-column tile)                     -> raw rows
//   launch l = 1..4: LayerNorm_{l-1} + ReLU of the raw rows ON LOAD (every column workgroup repeats it for its 16 rows:
//                    16 x 512 elements, nothing next to the launch boundary it saves) + Linear_l, 16 column groups of
//                    2 column tiles x the 2 K halves of the chain fold                          -> raw rows
//   launch 5: LayerNorm_4 + ReLU on load + encoder.14 + VQ search (one workgroup per row tile, as the fused tail)
// Six launches instead of fourteen, same chains, same bits.  The K blocks of a chain fold are independent
// zero-started chains, so a wave runs them interleaved (NBLK accumulators) instead of one dependent MFMA sequence.
// ------------------------------------------------------------------------------------------
// A wave's whole weight slice (one 16-column tile, NQ q-steps) is requested up front -- 4 NQ registers -- so that the
// loads fly under the A-tile load and the LayerNorm, and the MFMAs then run back to back (a 1-deep prefetch left every
// step waiting ~0.3 us on L2: 10.4 us per launch, rocprofv3).
template <int NQ>
__device__ __forceinline__ void rows16_load_w(const float4 *__restrict__ Wf, int ct, int lane, float4 (&wf)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) wf[q] = Wf[((size_t)ct * NQ + q) * 64 + lane];
}
template <int NBLK, int KCQ>
__device__ __forceinline__ f32x4 rows16_gemm_pre(const float *tile, const float4 (&wf)[NBLK * KCQ], int ct,
                                                 const float *__restrict__ bias, int lane) {
    const float *arow = tile + (lane & 15) * FE_LD + (lane >> 4);
    f32x4 acc[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KCQ; ++s) {
        float a[NBLK][4];
#pragma unroll
        for (int b = 0; b < NBLK; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c) a[b][c] = arow[16 * (b * KCQ + s) + 4 * c];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[b][0], wf[b * KCQ + s].x, acc[b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[b][1], wf[b * KCQ + s].y, acc[b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[b][2], wf[b * KCQ + s].z, acc[b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[b][3], wf[b * KCQ + s].w, acc[b], 0, 0, 0);
    }
    const float bv = bias ? bias[16 * ct + (lane & 15)] : 0.f;
    f32x4 tot;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        tot[r] = bias ? bv + acc[0][r] : acc[0][r];
#pragma unroll
        for (int b = 1; b < NBLK; ++b) tot[r] = tot[r] + acc[b][r];
    }
    return tot;
}

// The three layer kinds of the split schedule for column group `cg` of the row tile at r0 (bodies of the enc_split_*_kernel launches):
// conv (layer 0), Linear l = 1..4 (LayerNorm l-1 on load); encoder.14 + VQ (LayerNorm 4 on load) has a launch shape of its own.
__device__ __forceinline__ void split_conv(const FusedP &p, float *__restrict__ out, int cg, int r0, float *tile) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K0 = 4 * p.C;
    const int ct = 4 * cg + wave, nq = K0 / 16;
    // vmcnt retires in order: the gather's loads go FIRST, the weight slice behind them, so that waiting for the
    // gather leaves the weights in flight
    float gv[20];
#pragma unroll
    for (int n = 0; n < 20; ++n) {
        const int e = tid + 256 * n, i = e & 15, kidx = e >> 4, m = r0 + i;
        int c, tap;
        if (p.conv_mode == 1) { c = kidx >> 2; tap = kidx & 3; }
        else { const int rem = kidx & 63; tap = rem >> 4; c = (kidx >> 6) * 16 + (rem & 15); }
        float v = 0.f;
        if (e < 16 * K0 && m < p.N) {
            const int b = m / p.To, tt = m - b * p.To, ti = 2 * tt + tap - 1;
            if (ti >= 0 && ti < p.T) v = p.mel[((size_t)b * p.C + c) * p.T + ti];
        }
        gv[n] = v;
    }
    float4 wc[20];
    if (nq == 20) rows16_load_w<20>(p.conv_f, ct, lane, wc);          // C = 80 (the reference): whole slice up front
#pragma unroll
    for (int n = 0; n < 20; ++n) {
        const int e = tid + 256 * n;
        if (e < 16 * K0) tile[(e & 15) * FE_LD + (e >> 4)] = gv[n];
    }
    for (int e = tid + 256 * 20; e < 16 * K0; e += 256) {             // more than 80 channels: the rest, plainly
        const int i = e & 15, kidx = e >> 4, m = r0 + i;
        int c, tap;
        if (p.conv_mode == 1) { c = kidx >> 2; tap = kidx & 3; }
        else { const int rem = kidx & 63; tap = rem >> 4; c = (kidx >> 6) * 16 + (rem & 15); }
        float v = 0.f;
        if (m < p.N) {
            const int b = m / p.To, tt = m - b * p.To, ti = 2 * tt + tap - 1;
            if (ti >= 0 && ti < p.T) v = p.mel[((size_t)b * p.C + c) * p.T + ti];
        }
        tile[i * FE_LD + kidx] = v;
    }
    __syncthreads();
    f32x4 tot;
    if (nq == 20 && p.conv_mode == 1) tot = rows16_gemm_pre<1, 20>(tile, wc, ct, nullptr, lane);
    else if (nq == 20) tot = rows16_gemm_pre<5, 4>(tile, wc, ct, nullptr, lane);
    else {                                                            // other channel counts: the generic fold
        f32x4 t1[1];
        float4 fr1[4][1];
        rows16_prefetch<1, 4>(p.conv_f, ct, nq, fr1, lane);
        rows16_gemm<1, 4>(tile, p.conv_f, ct, nq, p.conv_mode == 1 ? nq : 4, nullptr, t1, lane, fr1);
        tot = t1[0];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = r0 + 4 * (lane >> 4) + r;
        if (m < p.N) out[(size_t)m * 512 + 16 * ct + (lane & 15)] = tot[r];
    }
}

// 512 threads: 8 column groups per row tile; a workgroup = 4 column tiles x the 2 K halves of the chain fold (they are
// independent zero-started chains: tot = c0 + c1), so a wave streams 16 KB of weights and issues 64 MFMAs; every column
// workgroup repeats the LayerNorm of its row tile, a half-wave per row.
__device__ __forceinline__ void split_fc(const FusedP &p, int layer, const float *__restrict__ in, float *__restrict__ out, int cg,
                                         int r0, float *tile, float (*part)[16][17]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (vmcnt retires in order: the rows' loads go first, the weight slice behind them)
    float4 av[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int e = tid + 512 * n, row = e >> 7, c4 = e & 127;
        av[n] = r0 + row < p.N ? ((const float4 *)in)[(size_t)(r0 + row) * 128 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int cl = wave & 3, ct = 4 * cg + cl, kh = wave >> 2;
    float4 wh[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) wh[q] = p.fc_f[layer - 1][((size_t)ct * 32 + 16 * kh + q) * 64 + lane];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int e = tid + 512 * n, row = e >> 7, c4 = e & 127;
        float *d = tile + row * FE_LD + 4 * c4;
        *(float2 *)d = make_float2(av[n].x, av[n].y);
        *(float2 *)(d + 2) = make_float2(av[n].z, av[n].w);
    }
    __syncthreads();
    rows16_layernorm<1>(tile, p.ln_g[layer - 1], p.ln_b[layer - 1], p.eps, p.lnc, tid);
    __syncthreads();
    const f32x4 acc = rows16_gemm_pre<1, 16>(tile + 256 * kh, wh, ct, nullptr, lane);      // k in [256 kh, 256 kh + 256)
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) part[cl][4 * (lane >> 4) + r][lane & 15] = acc[r];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = r0 + 4 * (lane >> 4) + r;
            if (m < p.N) out[(size_t)m * 512 + 16 * ct + (lane & 15)] = acc[r] + part[cl][4 * (lane >> 4) + r][lane & 15];
        }
    }
}

// The last launch, one 512-thread workgroup per row tile: LayerNorm 4 + ReLU on load (a half-wave per row), encoder.14 with
// its two K-block chains on wave pairs (as the fused launch's tail), VQ search on 8 waves.
__global__ __launch_bounds__(512) void enc_split_tail_kernel(FusedP p, const float *__restrict__ in) {
    __shared__ __attribute__((aligned(16))) float tile[16 * FE_LD];
    __shared__ VqSmem sm;
    __shared__ float part[4][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r0 = blockIdx.y * 16;
    // (vmcnt retires in order: the rows' loads go first, the weight slice behind them)
    float4 av[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int e = tid + 512 * n, row = e >> 7, c4 = e & 127;
        av[n] = r0 + row < p.N ? ((const float4 *)in)[(size_t)(r0 + row) * 128 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int ct = wave & 3, kh = wave >> 2;
    float4 wh[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) wh[q] = p.out_f[((size_t)ct * 32 + 16 * kh + q) * 64 + lane];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int e = tid + 512 * n, row = e >> 7, c4 = e & 127;
        float *d = tile + row * FE_LD + 4 * c4;
        *(float2 *)d = make_float2(av[n].x, av[n].y);
        *(float2 *)(d + 2) = make_float2(av[n].z, av[n].w);
    }
    __syncthreads();
    rows16_layernorm<1>(tile, p.ln_g[4], p.ln_b[4], p.eps, p.lnc, tid);
    __syncthreads();
    const int nwv = p.n_emb % 128 == 0 ? 8 : 4;          // waves of the VQ search
    const int tpw = p.n_emb / (16 * nwv), t0 = (wave < nwv ? wave : 0) * tpw;
    float4 f0[4] = {}, f1[4] = {};
    if (wave < nwv) {
        vq_load_tile(p.Ef, t0, lane, f0);
        vq_load_tile(p.Ef, tpw > 1 ? t0 + 1 : t0, lane, f1);
    }
    // (bias + c0) + c1: wave w < 4 runs c0 of column tile w, wave w + 4 runs c1
    const f32x4 zt = rows16_gemm_pre<1, 16>(tile + 256 * kh, wh, ct, kh == 0 ? p.out_b : nullptr, lane);
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) part[ct][4 * (lane >> 4) + r][lane & 15] = zt[r];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (lane >> 4) + r, col = 16 * ct + (lane & 15);
            const float z = zt[r] + part[ct][row][lane & 15];
            sm.xs[row][col] = z;
            if (p.z_pre && r0 + row < p.N) p.z_pre[(size_t)(r0 + row) * 64 + col] = z;
        }
    }
    __syncthreads();
    if (tid < 16) sm.x2s[tid] = r0 + tid < p.N ? sumsq64(&sm.xs[tid][0]) : 0.f;
    __syncthreads();
    vq_rows16(sm, r0, p.N, p.Ef, p.E, p.e2, p.n_emb, p.idx, p.z_q, tid, f0, f1, nwv);
}

__global__ __launch_bounds__(256) void enc_split_conv_kernel(FusedP p, float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float tile[16 * FE_LD];
    split_conv(p, out, blockIdx.x, blockIdx.y * 16, tile);
}
__global__ __launch_bounds__(512) void enc_split_fc_kernel(FusedP p, int layer, const float *__restrict__ in, float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float tile[16 * FE_LD];
    __shared__ float part[4][16][17];
    split_fc(p, layer, in, out, blockIdx.x, blockIdx.y * 16, tile, part);
}

// (A resident one-launch form of this schedule -- the 16 column workgroups of a row tile on one XCD, handing the rows over
// in-kernel through that XCD's L2 -- was built and measured slower, 62 vs 50.5 us: profiles/r02_encoder_resident_timeline.txt.)

// Eval-branch statistics of VQEmbeddingEMA.forward (model.py:147-153): deterministic two-level
// reductions (per-block partials, then one block in fixed order).
__global__ __launch_bounds__(256) void vq_stats_partial_kernel(const float *__restrict__ x, const float *__restrict__ q,
                                                               const int64_t *__restrict__ idx, int n_rows,
                                                               float *__restrict__ zst, double *__restrict__ part_se,
                                                               unsigned *__restrict__ hist) {
    __shared__ double red[256];
    const size_t n = (size_t)n_rows * 64;
    double se = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float xv = x[i], qv = q[i];
        const float d = xv - qv;
        se += (double)d * (double)d;
        if (zst) zst[i] = xv + (qv - xv);
    }
    for (int r = blockIdx.x * 256 + threadIdx.x; r < n_rows; r += gridDim.x * 256) atomicAdd(&hist[idx[r]], 1u);
    red[threadIdx.x] = se;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part_se[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void vq_stats_final_kernel(const double *__restrict__ part_se, int nparts,
                                                             const unsigned *__restrict__ hist, int n_emb, int n_rows,
                                                             float *__restrict__ loss, float *__restrict__ ppl) {
    __shared__ double red[256];
    double ent = 0.0;
    for (int j = threadIdx.x; j < n_emb; j += 256) {
        const float p = (float)((double)hist[j] / (double)n_rows);
        ent += (double)(p * logf(p + 1e-10f));
    }
    red[threadIdx.x] = ent;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double se = 0.0;
        for (int i = 0; i < nparts; ++i) se += part_se[i];
        *loss = (float)(0.25 * se / ((double)n_rows * 64.0));
        *ppl = expf((float)-red[0]);
    }
}

// conv.weight (O, C, 4) -> GEMM operand (O, 4C) in the k order of each reference back-end.
__global__ void conv_weight_permute_kernel(const float *__restrict__ w, float *__restrict__ w1,
                                           float *__restrict__ w2, int O, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int K = 4 * C;
    if (i >= O * K) return;
    const int o = i / K, kidx = i - o * K;
    w1[i] = w[i];                                          // mode 1: kidx = c*4 + tap (native layout)
    const int rem = kidx & 63, tap = rem >> 4, c = (kidx >> 6) * 16 + (rem & 15);
    w2[i] = w[((size_t)o * C + c) * 4 + tap];              // mode 2: [block of 16 c][tap][c in block]
}

// ------------------------------------------------------------------------------------------
// handle + ABI
// ------------------------------------------------------------------------------------------
struct vqcpc_encoder {
    int in_channels, channels, n_emb, z_dim, c_dim;
    float *conv_w1 = nullptr, *conv_w2 = nullptr;
    float *ln_g[5] = {}, *ln_b[5] = {};
    float *fc_w[4] = {};
    float *out_w = nullptr, *out_b = nullptr;
    float *codebook = nullptr, *e2 = nullptr, *cbfrag = nullptr;
    LstmPlan *lstm = nullptr;
    int dbg_drop = -1, dbg_timeout_ms = 1000;      // tests of the resident scan's abort path
    LnConst lnc;
    DevBuf bufA, bufB, zpre, stats;
    // fused front end (enc_fused_kernel): weights in 16x16x4 fragment order
    float4 *conv_f[2] = {nullptr, nullptr}, *fc_f[4] = {}, *out_f = nullptr;
    int fused = -1;                      // -1 auto (split below split_max_tiles row tiles, else fused), 0 layered kernels,
                                         // 1 one-launch fused kernel, 2 six-launch column-split kernels
    int split_max_tiles = 80;            // auto: calls of up to this many 16-row tiles take the column-split launches
};

static int dev_copy(float **dst, const float *src, size_t n) {
    HIP_TRY(hipMalloc((void **)dst, n * sizeof(float)));
    HIP_TRY(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice));
    return VQCPC_OK;
}
#define TRY(x) do { int rc_ = (x); if (rc_ != VQCPC_OK) return rc_; } while (0)

extern "C" void vqcpc_encoder_destroy(vqcpc_encoder *e) {
    if (!e) return;
    float *ptrs[] = {e->conv_w1, e->conv_w2, e->out_w, e->out_b, e->codebook, e->e2, e->cbfrag};
    for (float *p : ptrs) if (p) (void)hipFree(p);
    for (int i = 0; i < 5; ++i) { if (e->ln_g[i]) (void)hipFree(e->ln_g[i]); if (e->ln_b[i]) (void)hipFree(e->ln_b[i]); }
    for (int i = 0; i < 4; ++i) if (e->fc_w[i]) (void)hipFree(e->fc_w[i]);
    if (e->lstm) vq_lstm_plan_destroy(e->lstm);
    float4 *fr[] = {e->conv_f[0], e->conv_f[1], e->fc_f[0], e->fc_f[1], e->fc_f[2], e->fc_f[3], e->out_f};
    for (float4 *q : fr) if (q) (void)hipFree(q);
    e->bufA.release(); e->bufB.release();
    e->zpre.release(); e->stats.release();
    delete e;
}

static int build_frag16(const float *W, int N, int K, float4 **out) {
    VQ_REQUIRE(N % 16 == 0 && K % 64 == 0, "build_frag16: unsupported shape (%d, %d)", N, K);
    const size_t n4 = (size_t)(N / 16) * (K / 16) * 64;
    HIP_TRY(hipMalloc((void **)out, n4 * sizeof(float4)));
    hipLaunchKernelGGL(frag16_build_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, W, N, K, *out);
    HIP_TRY(hipGetLastError());
    return VQCPC_OK;
}

static int encoder_create_impl(const vqcpc_encoder_weights *w, vqcpc_encoder *e) {
    const int C = w->in_channels, CH = w->channels;
    e->in_channels = C; e->channels = CH; e->n_emb = w->n_embeddings; e->z_dim = w->z_dim; e->c_dim = w->c_dim;
    for (int j = 0; j < 16; ++j) e->lnc.inv[j] = 1.0f / (float)(j + 1);
    for (int q = 0; q < 8; ++q) e->lnc.sc[q] = (float)64 / (float)(64 * (q + 1));
    const size_t nconv = (size_t)CH * C * 4;
    HIP_TRY(hipMalloc((void **)&e->conv_w1, nconv * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&e->conv_w2, nconv * sizeof(float)));
    hipLaunchKernelGGL(conv_weight_permute_kernel, dim3((unsigned)((nconv + 255) / 256)), dim3(256), 0, 0,
                       w->conv_weight, e->conv_w1, e->conv_w2, CH, C);
    HIP_TRY(hipGetLastError());
    for (int i = 0; i < 5; ++i) { TRY(dev_copy(&e->ln_g[i], w->ln_weight[i], CH)); TRY(dev_copy(&e->ln_b[i], w->ln_bias[i], CH)); }
    for (int i = 0; i < 4; ++i) TRY(dev_copy(&e->fc_w[i], w->fc_weight[i], (size_t)CH * CH));
    TRY(dev_copy(&e->out_w, w->out_weight, (size_t)w->z_dim * CH));
    TRY(dev_copy(&e->out_b, w->out_bias, w->z_dim));
    TRY(dev_copy(&e->codebook, w->codebook, (size_t)w->n_embeddings * w->z_dim));
    HIP_TRY(hipMalloc((void **)&e->e2, w->n_embeddings * sizeof(float)));
    hipLaunchKernelGGL(rowsumsq64_kernel, dim3((w->n_embeddings + 63) / 64), dim3(64), 0, 0, e->codebook, e->e2,
                       w->n_embeddings);
    HIP_TRY(hipMalloc((void **)&e->cbfrag, (size_t)w->n_embeddings * 64 * sizeof(float)));
    hipLaunchKernelGGL(vq_build_frag_kernel, dim3((w->n_embeddings * 16 + 255) / 256), dim3(256), 0, 0, e->codebook,
                       (float4 *)e->cbfrag, w->n_embeddings);
    HIP_TRY(hipGetLastError());
    TRY(vq_lstm_plan_create(w->rnn_w_ih, w->rnn_w_hh, w->rnn_b_ih, w->rnn_b_hh, w->z_dim, w->c_dim, &e->lstm));

    // fragment-ordered copies for the fused front end (the activation tile is 512 wide: 4 C <= 512)
    if (4 * C <= 512) {
        TRY(build_frag16(e->conv_w1, CH, 4 * C, &e->conv_f[0]));
        TRY(build_frag16(e->conv_w2, CH, 4 * C, &e->conv_f[1]));
        for (int i = 0; i < 4; ++i) TRY(build_frag16(e->fc_w[i], CH, CH, &e->fc_f[i]));
        TRY(build_frag16(e->out_w, w->z_dim, CH, &e->out_f));
    }
    HIP_TRY(hipDeviceSynchronize());
    return VQCPC_OK;
}

extern "C" int vqcpc_encoder_create(const vqcpc_encoder_weights *w, vqcpc_encoder **out) {
    VQ_REQUIRE(w && out, "vqcpc_encoder_create: null argument");
    *out = nullptr;
    TRY(require_gfx950());
    VQ_REQUIRE(w->channels == 512 && w->z_dim == 64, "encoder: channels must be 512 and z_dim 64 (got %d, %d)",
               w->channels, w->z_dim);
    VQ_REQUIRE(w->in_channels % 16 == 0 && w->in_channels > 0 && w->in_channels <= 256,
               "encoder: in_channels must be a multiple of 16 in (0, 256] (got %d)", w->in_channels);
    VQ_REQUIRE(w->n_embeddings % 64 == 0 && w->n_embeddings > 0 && w->n_embeddings <= 4096,
               "encoder: n_embeddings must be a multiple of 64 in (0, 4096] (got %d)", w->n_embeddings);
    VQ_REQUIRE(w->c_dim % 64 == 0 && w->c_dim > 0 && w->c_dim <= 1024, "encoder: c_dim must be a multiple of 64 (got %d)", w->c_dim);
    vqcpc_encoder *e = new vqcpc_encoder();
    int rc = encoder_create_impl(w, e);
    if (rc != VQCPC_OK) { vqcpc_encoder_destroy(e); return rc; }
    *out = e;
    return VQCPC_OK;
}

// conv + seg-FC stack up to `stop_stage` (0 conv, 1 LN0+ReLU, 2+2l FC_l, 3+2l LN_l+ReLU, 10 z_pre).
// Returns the device buffer holding that stage's rows in *stage_out.
static int encoder_front(vqcpc_encoder *e, const float *mel, int B, int T, int conv_mode, int stop_stage,
                         float *zp, const float **stage_out, hipStream_t s) {
    const int C = e->in_channels, CH = e->channels, To = (T - 2) / 2 + 1, N = B * To;
    if (conv_mode == VQCPC_CONV_AUTO)
        conv_mode = (B > 1 || (long)B * C * T > 20480) ? VQCPC_CONV_DIRECT : VQCPC_CONV_IM2COL;
    TRY(e->bufA.reserve((size_t)N * CH * sizeof(float)));
    TRY(e->bufB.reserve((size_t)N * CH * sizeof(float)));
    float *a = e->bufA.as<float>(), *b = e->bufB.as<float>();

    // conv (model.py:65) as an im2col GEMM in the reference back-end's summation order
    GemmP p{};
    p.W = conv_mode == VQCPC_CONV_IM2COL ? e->conv_w1 : e->conv_w2;
    p.Y = a; p.ldy = CH; p.M = N; p.N = CH; p.K = 4 * C;
    p.KC = conv_mode == VQCPC_CONV_IM2COL ? 4 * C : 64;
    p.x = mel; p.C = C; p.T = T; p.To = To;
    if (conv_mode == VQCPC_CONV_IM2COL) TRY((launch_gemm<1>(p, s)));
    else TRY((launch_gemm<2>(p, s)));
    *stage_out = a;
    if (stop_stage == 0) return VQCPC_OK;

    // seg-FC stack (model.py:46-55, :67)
    const dim3 lng((N + 7) / 8), lnb(256);
    hipLaunchKernelGGL(ln512_kernel, lng, lnb, 0, s, a, e->ln_g[0], e->ln_b[0], b, N, 1e-5f, 1, e->lnc);
    *stage_out = b;
    if (stop_stage == 1) return VQCPC_OK;
    for (int l = 0; l < 4; ++l) {
        TRY(vq_gemm_chain(b, CH, e->fc_w[l], nullptr, a, CH, N, CH, CH, 256, s));
        *stage_out = a;
        if (stop_stage == 2 + 2 * l) return VQCPC_OK;
        hipLaunchKernelGGL(ln512_kernel, lng, lnb, 0, s, a, e->ln_g[l + 1], e->ln_b[l + 1], b, N, 1e-5f, 1, e->lnc);
        *stage_out = b;
        if (stop_stage == 3 + 2 * l) return VQCPC_OK;
    }
    TRY(vq_gemm_chain(b, CH, e->out_w, e->out_b, zp, 64, N, 64, CH, 256, s));
    *stage_out = zp;
    HIP_TRY(hipGetLastError());
    return VQCPC_OK;
}

static bool use_fused(const vqcpc_encoder *e) { return e->fused != 0 && e->conv_f[0] != nullptr; }

// The whole front end + VQ in one launch (stage < 0), or up to `stage` with that stage's rows in stage_out.
static int encoder_fused(vqcpc_encoder *e, const float *mel, int B, int T, int conv_mode, float *z_pre, float *z_q,
                         int64_t *idx, int stage, float *stage_out, hipStream_t s) {
    const int C = e->in_channels, To = (T - 2) / 2 + 1, N = B * To;
    if (conv_mode == VQCPC_CONV_AUTO)
        conv_mode = (B > 1 || (long)B * C * T > 20480) ? VQCPC_CONV_DIRECT : VQCPC_CONV_IM2COL;
    FusedP p{};
    p.mel = mel; p.C = C; p.T = T; p.To = To; p.N = N; p.conv_mode = conv_mode;
    p.conv_f = e->conv_f[conv_mode == VQCPC_CONV_IM2COL ? 0 : 1];
    for (int i = 0; i < 5; ++i) { p.ln_g[i] = e->ln_g[i]; p.ln_b[i] = e->ln_b[i]; }
    for (int i = 0; i < 4; ++i) p.fc_f[i] = e->fc_f[i];
    p.out_f = e->out_f; p.out_b = e->out_b;
    p.Ef = (const float4 *)e->cbfrag; p.E = e->codebook; p.e2 = e->e2; p.n_emb = e->n_emb;
    p.z_pre = z_pre; p.z_q = z_q; p.idx = idx; p.stage_out = stage_out; p.stage = stage;
    p.eps = 1e-5f; p.lnc = e->lnc;
    const int ntiles = (N + 15) / 16;
    const bool split = stage < 0 && (e->fused == 2 || (e->fused != 1 && ntiles <= e->split_max_tiles));
    if (split) {                                          // small call: six column-split launches (enc_split_*_kernel)
        TRY(e->bufA.reserve((size_t)N * 512 * sizeof(float)));
        TRY(e->bufB.reserve((size_t)N * 512 * sizeof(float)));
        float *a = e->bufA.as<float>(), *b = e->bufB.as<float>();
        hipLaunchKernelGGL(enc_split_conv_kernel, dim3(8, ntiles), dim3(256), 0, s, p, a);
        for (int l = 1; l <= 4; ++l) {
            hipLaunchKernelGGL(enc_split_fc_kernel, dim3(8, ntiles), dim3(512), 0, s, p, l, (const float *)a, b);
            float *t = a; a = b; b = t;
        }
        hipLaunchKernelGGL(enc_split_tail_kernel, dim3(1, ntiles), dim3(512), 0, s, p, (const float *)a);
    } else {
        hipLaunchKernelGGL(enc_fused_kernel, dim3(ntiles), dim3(512), 0, s, p);
    }
    HIP_TRY(hipGetLastError());
    return VQCPC_OK;
}

extern "C" int vqcpc_encoder_set_option(vqcpc_encoder *e, const char *name, int value) {
    VQ_REQUIRE(e && name, "vqcpc_encoder_set_option: null argument");
    if (!strcmp(name, "fused")) {
        VQ_REQUIRE(value >= -1 && value <= 2, "fused must be -1 (auto), 0, 1 or 2");
        VQ_REQUIRE(value < 1 || e->conv_f[0], "fused front end needs 4 * in_channels <= 512");
        e->fused = value;
        return VQCPC_OK;
    }
    if (!strcmp(name, "split_max_tiles")) {
        VQ_REQUIRE(value >= 0, "split_max_tiles must be >= 0");
        e->split_max_tiles = value;
        return VQCPC_OK;
    }
    if (!strcmp(name, "persistent_context")) return vq_lstm_set_persistent(e->lstm, value == 2 ? 2 : (value != 0 ? -1 : 0));
    if (!strcmp(name, "context_debug_drop_step")) { e->dbg_drop = value; return vq_lstm_set_debug(e->lstm, e->dbg_drop, e->dbg_timeout_ms); }
    if (!strcmp(name, "context_timeout_ms")) {
        VQ_REQUIRE(value >= 1 && value <= 10000, "context_timeout_ms must be in [1, 10000]");
        e->dbg_timeout_ms = value;
        return vq_lstm_set_debug(e->lstm, e->dbg_drop, e->dbg_timeout_ms);
    }
    vq_set_error("unknown option %s", name);
    return VQCPC_ERR_INVALID;
}

extern "C" int vqcpc_encoder_encode(vqcpc_encoder *e, const float *mel, int B, int T, int conv_mode,
                                    float *z_q, float *c, int64_t *idx, float *z_pre, void *stream) {
    VQ_REQUIRE(e && mel && z_q && idx, "vqcpc_encoder_encode: null argument");
    VQ_REQUIRE(B > 0 && T >= 2, "encoder.encode: need B > 0 and T >= 2 (got B=%d T=%d)", B, T);
    VQ_REQUIRE(conv_mode >= 0 && conv_mode <= 2, "encoder.encode: conv_mode must be 0, 1 or 2");
    hipStream_t s = (hipStream_t)stream;
    const int To = (T - 2) / 2 + 1, N = B * To;
    float *zp = z_pre;
    if (!zp && !use_fused(e)) { TRY(e->zpre.reserve((size_t)N * 64 * sizeof(float))); zp = e->zpre.as<float>(); }
    VQ_REQUIRE(((uintptr_t)zp & 15) == 0 && ((uintptr_t)z_q & 15) == 0, "encoder.encode: outputs must be 16-byte aligned");
    if (use_fused(e)) {
        TRY(encoder_fused(e, mel, B, T, conv_mode, z_pre, z_q, idx, -1, nullptr, s));     // z_pre only if asked for
    } else {
        const float *unused = nullptr;
        TRY(encoder_front(e, mel, B, T, conv_mode, 10, zp, &unused, s));
        // VQ (model.py:103-115)
        hipLaunchKernelGGL(vq_encode_kernel, dim3((N + 15) / 16), dim3(256), 0, s, zp, N, (const float4 *)e->cbfrag, e->codebook,
                           e->e2, e->n_emb, idx, z_q);
        HIP_TRY(hipGetLastError());
    }
    if (c) TRY(vq_lstm_run(e->lstm, z_q, B, To, c, s));
    return VQCPC_OK;
}

extern "C" int vqcpc_encoder_check(vqcpc_encoder *e) {
    VQ_REQUIRE(e, "vqcpc_encoder_check: null argument");
    return vq_lstm_check(e->lstm);
}

extern "C" int vqcpc_encoder_vq_encode(vqcpc_encoder *e, const float *x, int n_rows, float *z_q, int64_t *idx,
                                       void *stream) {
    VQ_REQUIRE(e && x && z_q && idx && n_rows > 0, "vqcpc_encoder_vq_encode: bad argument");
    VQ_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)z_q & 15) == 0, "vqcpc_encoder_vq_encode: rows must be 16-byte aligned");
    hipLaunchKernelGGL(vq_encode_kernel, dim3((n_rows + 15) / 16), dim3(256), 0, (hipStream_t)stream, x, n_rows,
                       (const float4 *)e->cbfrag, e->codebook, e->e2, e->n_emb, idx, z_q);
    HIP_TRY(hipGetLastError());
    return VQCPC_OK;
}

extern "C" int vqcpc_encoder_stage(vqcpc_encoder *e, const float *mel, int B, int T, int conv_mode, int stage,
                                   float *out, void *stream) {
    VQ_REQUIRE(e && mel && out, "vqcpc_encoder_stage: null argument");
    VQ_REQUIRE(B > 0 && T >= 2 && stage >= 0 && stage <= 10, "vqcpc_encoder_stage: bad shape or stage");
    hipStream_t s = (hipStream_t)stream;
    const int N = B * ((T - 2) / 2 + 1);
    if (use_fused(e))                                    // the fused kernel stops after `stage` and dumps its tile
        return encoder_fused(e, mel, B, T, conv_mode, stage == 10 ? out : nullptr, nullptr, nullptr, stage,
                             stage == 10 ? nullptr : out, s);
    const float *src = nullptr;
    float *zp = nullptr;
    if (stage == 10) {
        VQ_REQUIRE(((uintptr_t)out & 15) == 0, "vqcpc_encoder_stage: out must be 16-byte aligned");
        zp = out;
    }
    TRY(encoder_front(e, mel, B, T, conv_mode, stage, zp, &src, s));
    if (stage != 10)
        HIP_TRY(hipMemcpyAsync(out, src, (size_t)N * e->channels * sizeof(float), hipMemcpyDeviceToDevice, s));
    return VQCPC_OK;
}

extern "C" int vqcpc_encoder_context(vqcpc_encoder *e, const float *z, int B, int Tz, float *c, void *stream) {
    VQ_REQUIRE(e && z && c && B > 0 && Tz > 0, "vqcpc_encoder_context: bad argument");
    return vq_lstm_run(e->lstm, z, B, Tz, c, (hipStream_t)stream);
}

extern "C" int vqcpc_encoder_forward_stats(vqcpc_encoder *e, const float *z_pre, const float *z_q,
                                           const int64_t *idx, int n_rows, float *z_st, float *loss,
                                           float *perplexity, void *stream) {
    VQ_REQUIRE(e && z_pre && z_q && idx && loss && perplexity && n_rows > 0, "vqcpc_encoder_forward_stats: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = 64;
    const size_t hist_bytes = (size_t)e->n_emb * sizeof(unsigned);
    TRY(e->stats.reserve(hist_bytes + nblk * sizeof(double) + 64));
    unsigned *hist = e->stats.as<unsigned>();
    double *part = (double *)((char *)e->stats.p + ((hist_bytes + 15) / 16) * 16);
    HIP_TRY(hipMemsetAsync(hist, 0, hist_bytes, s));
    hipLaunchKernelGGL(vq_stats_partial_kernel, dim3(nblk), dim3(256), 0, s, z_pre, z_q, idx, n_rows, z_st, part, hist);
    hipLaunchKernelGGL(vq_stats_final_kernel, dim3(1), dim3(256), 0, s, part, nblk, hist, e->n_emb, n_rows, loss, perplexity);
    HIP_TRY(hipGetLastError());
    return VQCPC_OK;
}
