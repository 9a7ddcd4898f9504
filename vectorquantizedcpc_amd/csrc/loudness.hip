// loudness.hip -- the loudness steps of convert.py on gfx950.
// Replaces pyloudnorm.Meter(sr).integrated_loudness (convert.py:50,57,79) and
// pyloudnorm.normalize.loudness (convert.py:80) for mono audio: K-weighting (two biquads derived at the
// stream's rate, direct form II transposed like scipy.signal.lfilter), 400 ms blocks with 75 % overlap,
// absolute gate -70 LUFS, relative gate -10 LU, -0.691 + 10 log10(mean); one gain 10^((target - measured)/20).
// All arithmetic fp64 (the high-pass poles sit 0.007 from the unit circle at 16 kHz).
//
// The recurrence is serial in time, so it is cut into chunks of LC samples: the cascade is a linear
// system with a 4-vector state, state_end = P * state_start + r, P = (one-sample transition)^LC.
// Pass 1 runs every chunk from a zero state (r), a short scan over the chunks of an utterance fixes the
// start states, pass 2 re-runs the chunks from them and stores the squared output; one workgroup per
// gating block sums its 0.4 s of squares, one thread per utterance applies the two gates.
// pyloudnorm is absent offline: checked against oracle/loudness_ref.py (float64) -- parity unpinned.
#include "common.h"
#include <math.h>
#include <vector>

int vq_require_gfx950();
#define TRY(x) do { int rc_ = (x); if (rc_ != VQCPC_OK) return rc_; } while (0)

constexpr int LC = 128;                    // samples per chunk
constexpr double BLOCK_S = 0.400, STEP = 1.0 - 0.75, GAMMA_ABS = -70.0;

struct KW {                                // the K-weighting cascade, normalised by a0
    double b[2][3], a[2][2];
    double P[16];                          // LC-sample homogeneous transition of the 4-vector state
};

struct vqcpc_loudness {
    int rate;
    KW k;
    DevBuf lens, r, sq, z, bl, bu, bb, boff, gain;
};

__device__ __host__ __forceinline__ double kw_step(const KW &k, double x, double *s) {
    const double y1 = k.b[0][0] * x + s[0];
    s[0] = k.b[0][1] * x - k.a[0][0] * y1 + s[1];
    s[1] = k.b[0][2] * x - k.a[0][1] * y1;
    const double y2 = k.b[1][0] * y1 + s[2];
    s[2] = k.b[1][1] * y1 - k.a[1][0] * y2 + s[3];
    s[3] = k.b[1][2] * y1 - k.a[1][1] * y2;
    return y2;
}

// APPLY = 0: zero start state, writes the end state r[b][c][4].  APPLY = 1: start state s0[b][c][4], writes y^2.
template <int APPLY>
__global__ void ld_chunk_kernel(const float *__restrict__ wav, const int *__restrict__ len, int Lmax, int nchmax, KW k,
                                double *__restrict__ st, double *__restrict__ sq) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    const int L = len[b];
    if (c >= nchmax || c * LC >= L) return;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    double *rec = st + ((size_t)b * nchmax + c) * 4;
    if (APPLY) { s[0] = rec[0]; s[1] = rec[1]; s[2] = rec[2]; s[3] = rec[3]; }
    const int i0 = c * LC, i1 = min(L, i0 + LC);
    const float *x = wav + (size_t)b * Lmax;
    for (int i = i0; i < i1; ++i) {
        const double y = kw_step(k, (double)x[i], s);
        if (APPLY) sq[(size_t)b * Lmax + i] = y * y;
    }
    if (!APPLY) { rec[0] = s[0]; rec[1] = s[1]; rec[2] = s[2]; rec[3] = s[3]; }
}

// One thread per utterance: start state of chunk c = P * start(c-1) + r(c-1).  r is overwritten by the start states.
__global__ void ld_scan_kernel(const int *__restrict__ len, int B, int nchmax, KW k, double *__restrict__ st) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int nch = (len[b] + LC - 1) / LC;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int c = 0; c < nch; ++c) {
        double *rec = st + ((size_t)b * nchmax + c) * 4;
        const double r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
        rec[0] = s[0]; rec[1] = s[1]; rec[2] = s[2]; rec[3] = s[3];
        double n[4];
        for (int i = 0; i < 4; ++i)
            n[i] = ((k.P[i * 4 + 0] * s[0] + k.P[i * 4 + 1] * s[1]) + (k.P[i * 4 + 2] * s[2] + k.P[i * 4 + 3] * s[3]));
        s[0] = n[0] + r0; s[1] = n[1] + r1; s[2] = n[2] + r2; s[3] = n[3] + r3;
    }
}

// One 256-thread workgroup per gating block: z = sum(sq[l:u]) * inv, fixed summation order.
__global__ void ld_block_kernel(const double *__restrict__ sq, const int *__restrict__ bl, const int *__restrict__ bu,
                                const int *__restrict__ bb, int Lmax, double inv, double *__restrict__ z) {
    __shared__ double part[256];
    const int j = blockIdx.x;
    const double *x = sq + (size_t)bb[j] * Lmax;
    double acc = 0.0;
    for (int i = bl[j] + threadIdx.x; i < bu[j]; i += 256) acc += x[i];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) z[j] = inv * part[0];
}

// One thread per utterance: the two gates.  Mirrors the empty-set behaviour of the reference's meter
// (mean of nothing -> NaN -> every comparison false -> 0 -> -inf).
__global__ void ld_gate_kernel(const double *__restrict__ z, const int *__restrict__ boff, int B, double *__restrict__ lufs) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int j0 = boff[b], j1 = boff[b + 1];
    double sum = 0.0; int n = 0;
    for (int j = j0; j < j1; ++j) {
        const double l = -0.691 + 10.0 * log10(z[j]);
        if (l >= GAMMA_ABS) { sum += z[j]; ++n; }
    }
    const double gamma_r = n ? -0.691 + 10.0 * log10(sum / n) - 10.0 : __longlong_as_double(0x7ff8000000000000ll);
    sum = 0.0; n = 0;
    for (int j = j0; j < j1; ++j) {
        const double l = -0.691 + 10.0 * log10(z[j]);
        if (l > gamma_r && l > GAMMA_ABS) { sum += z[j]; ++n; }
    }
    lufs[b] = -0.691 + 10.0 * log10(n ? sum / n : 0.0);
}

__global__ void ld_gain_kernel(const double *__restrict__ measured, const double *__restrict__ target, int B, double *__restrict__ gain) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) gain[b] = pow(10.0, (target[b] - measured[b]) / 20.0);
}

__global__ void ld_scale_kernel(float *__restrict__ wav, const int *__restrict__ len, int Lmax, const double *__restrict__ gain) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len[b]) wav[(size_t)b * Lmax + i] = (float)(gain[b] * (double)wav[(size_t)b * Lmax + i]);
}

static void biquad(bool shelf, double G, double Q, double fc, double rate, double *b, double *a) {
    const double A = pow(10.0, G / 40.0), w0 = 2.0 * M_PI * (fc / rate), alpha = sin(w0) / (2.0 * Q), c = cos(w0);
    double b0, b1, b2, a0, a1, a2;
    if (shelf) {
        b0 = A * ((A + 1) + (A - 1) * c + 2 * sqrt(A) * alpha);
        b1 = -2 * A * ((A - 1) + (A + 1) * c);
        b2 = A * ((A + 1) + (A - 1) * c - 2 * sqrt(A) * alpha);
        a0 = (A + 1) - (A - 1) * c + 2 * sqrt(A) * alpha;
        a1 = 2 * ((A - 1) - (A + 1) * c);
        a2 = (A + 1) - (A - 1) * c - 2 * sqrt(A) * alpha;
    } else {
        b0 = (1 + c) / 2; b1 = -(1 + c); b2 = (1 + c) / 2;
        a0 = 1 + alpha; a1 = -2 * c; a2 = 1 - alpha;
    }
    b[0] = b0 / a0; b[1] = b1 / a0; b[2] = b2 / a0; a[0] = a1 / a0; a[1] = a2 / a0;
}

// Gating blocks of an n-sample signal: count, and (when l/u are given) their sample bounds.
static int blocks_of(int n, int rate, std::vector<int> *l, std::vector<int> *u, int b, std::vector<int> *bb) {
    if ((double)n < BLOCK_S * rate) return 0;
    const double T = (double)n / rate;
    const int nb = (int)(nearbyint((T - BLOCK_S) / (BLOCK_S * STEP)) + 1);     // np.round: half to even
    if (l)
        for (int j = 0; j < nb; ++j) {
            const int lo = (int)(BLOCK_S * (j * STEP) * rate), hi = (int)(BLOCK_S * (j * STEP + 1) * rate);
            l->push_back(lo < n ? lo : n); u->push_back(hi < n ? hi : n); bb->push_back(b);
        }
    return nb;
}

extern "C" void vqcpc_loudness_destroy(vqcpc_loudness *m) {
    if (!m) return;
    DevBuf *bufs[] = {&m->lens, &m->r, &m->sq, &m->z, &m->bl, &m->bu, &m->bb, &m->boff, &m->gain};
    for (DevBuf *b : bufs) b->release();
    delete m;
}

extern "C" int vqcpc_loudness_create(int rate, vqcpc_loudness **out) {
    VQ_REQUIRE(out, "vqcpc_loudness_create: null argument");
    *out = nullptr;
    TRY(vq_require_gfx950());
    VQ_REQUIRE(rate >= 4000 && rate <= 768000, "vqcpc_loudness_create: rate %d outside [4000, 768000]", rate);
    vqcpc_loudness *m = new vqcpc_loudness();
    m->rate = rate;
    biquad(true, 4.0, 1.0 / sqrt(2.0), 1500.0, rate, m->k.b[0], m->k.a[0]);
    biquad(false, 0.0, 0.5, 38.0, rate, m->k.b[1], m->k.a[1]);
    for (int col = 0; col < 4; ++col) {                      // P column = LC zero-input steps from a unit state
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        s[col] = 1.0;
        for (int i = 0; i < LC; ++i) (void)kw_step(m->k, 0.0, s);
        for (int row = 0; row < 4; ++row) m->k.P[row * 4 + col] = s[row];
    }
    *out = m;
    return VQCPC_OK;
}

extern "C" int vqcpc_loudness_blocks(const vqcpc_loudness *m, int n_samples) {
    return (m && n_samples > 0) ? blocks_of(n_samples, m->rate, nullptr, nullptr, 0, nullptr) : 0;
}

static int upload_lens(vqcpc_loudness *m, const int *lens, int B, hipStream_t s) {
    TRY(m->lens.reserve(B * sizeof(int)));
    HIP_TRY(hipMemcpyAsync(m->lens.p, lens, B * sizeof(int), hipMemcpyHostToDevice, s));
    return VQCPC_OK;
}

extern "C" int vqcpc_loudness_integrated(vqcpc_loudness *m, const float *wav, const int *lens, int B, int Lmax, double *lufs,
                                         double *block_energy, void *stream) {
    VQ_REQUIRE(m && wav && lens && lufs && B > 0 && Lmax > 0, "vqcpc_loudness_integrated: bad argument");
    hipStream_t s = (hipStream_t)stream;
    std::vector<int> bl, bu, bb, boff(1, 0);
    for (int b = 0; b < B; ++b) {
        VQ_REQUIRE(lens[b] >= 0 && lens[b] <= Lmax, "vqcpc_loudness_integrated: lens[%d] = %d outside [0, %d]", b, lens[b], Lmax);
        VQ_REQUIRE(blocks_of(lens[b], m->rate, &bl, &bu, b, &bb) > 0,
                   "vqcpc_loudness_integrated: utterance %d has %d samples: audio must have length greater than the block size "
                   "(%.0f samples)", b, lens[b], BLOCK_S * m->rate);
        boff.push_back((int)bl.size());
    }
    const int nblk = (int)bl.size(), nchmax = (Lmax + LC - 1) / LC;
    TRY(upload_lens(m, lens, B, s));
    TRY(m->r.reserve((size_t)B * nchmax * 4 * sizeof(double)));
    TRY(m->sq.reserve((size_t)B * Lmax * sizeof(double)));
    TRY(m->z.reserve(nblk * sizeof(double)));
    TRY(m->bl.reserve(nblk * sizeof(int)));
    TRY(m->bu.reserve(nblk * sizeof(int)));
    TRY(m->bb.reserve(nblk * sizeof(int)));
    TRY(m->boff.reserve((B + 1) * sizeof(int)));
    HIP_TRY(hipMemcpyAsync(m->bl.p, bl.data(), nblk * sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(m->bu.p, bu.data(), nblk * sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(m->bb.p, bb.data(), nblk * sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(m->boff.p, boff.data(), (B + 1) * sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));                          // the tables above are host temporaries
    const int *dl = m->lens.as<int>();
    const dim3 cg((nchmax + 63) / 64, B);
    hipLaunchKernelGGL(ld_chunk_kernel<0>, cg, dim3(64), 0, s, wav, dl, Lmax, nchmax, m->k, m->r.as<double>(), (double *)nullptr);
    hipLaunchKernelGGL(ld_scan_kernel, dim3((B + 63) / 64), dim3(64), 0, s, dl, B, nchmax, m->k, m->r.as<double>());
    hipLaunchKernelGGL(ld_chunk_kernel<1>, cg, dim3(64), 0, s, wav, dl, Lmax, nchmax, m->k, m->r.as<double>(), m->sq.as<double>());
    hipLaunchKernelGGL(ld_block_kernel, dim3(nblk), dim3(256), 0, s, m->sq.as<double>(), m->bl.as<int>(), m->bu.as<int>(),
                       m->bb.as<int>(), Lmax, 1.0 / (BLOCK_S * m->rate), m->z.as<double>());
    hipLaunchKernelGGL(ld_gate_kernel, dim3((B + 63) / 64), dim3(64), 0, s, m->z.as<double>(), m->boff.as<int>(), B, lufs);
    if (block_energy)
        HIP_TRY(hipMemcpyAsync(block_energy, m->z.p, nblk * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipGetLastError());
    return VQCPC_OK;
}

extern "C" int vqcpc_loudness_normalize(vqcpc_loudness *m, float *wav, const int *lens, int B, int Lmax, const double *measured,
                                        const double *target, void *stream) {
    VQ_REQUIRE(m && wav && lens && measured && target && B > 0 && Lmax > 0, "vqcpc_loudness_normalize: bad argument");
    hipStream_t s = (hipStream_t)stream;
    for (int b = 0; b < B; ++b)
        VQ_REQUIRE(lens[b] >= 0 && lens[b] <= Lmax, "vqcpc_loudness_normalize: lens[%d] = %d outside [0, %d]", b, lens[b], Lmax);
    TRY(upload_lens(m, lens, B, s));
    HIP_TRY(hipStreamSynchronize(s));                          // lens is the caller's host buffer
    TRY(m->gain.reserve(B * sizeof(double)));
    hipLaunchKernelGGL(ld_gain_kernel, dim3((B + 63) / 64), dim3(64), 0, s, measured, target, B, m->gain.as<double>());
    hipLaunchKernelGGL(ld_scale_kernel, dim3((Lmax + 255) / 256, B), dim3(256), 0, s, wav, m->lens.as<int>(), Lmax,
                       m->gain.as<double>());
    HIP_TRY(hipGetLastError());
    return VQCPC_OK;
}
