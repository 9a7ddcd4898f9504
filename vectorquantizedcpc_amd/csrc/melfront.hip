// melfront.hip -- the reference's mel front-end on gfx950.
// Replaces wave_to_mel (preprocess.py:53-75) / convert.py:54-70: peak-normalise to 0.999,
// pre-emphasis y[n] = x[n] - a x[n-1] (preprocess.py:16-17), centered STFT (n_fft 2048, periodic
// Hann 400 zero-padded and centred, hop 160, reflect padding = librosa 0.8), magnitude,
// Slaney mel filterbank (80 bands from 50 Hz), amplitude_to_db (amin 1e-5, top_db 80 against the
// utterance maximum), / top_db + 1.  Parameters config.py:103-112.
//
// The window is only 400 samples wide, so a frame's DFT is a 400-term sum: both contractions
// (frames x [cos|sin] matrix, magnitudes x mel filterbank) run on the fp32 MFMA GEMM of encoder.hip.
// librosa is absent offline: checked against oracle/mel_ref.py (numpy float64) -- parity unpinned.
#include "common.h"
#include <math.h>
#include <vector>

int vq_require_gfx950();
#define TRY(x) do { int rc_ = (x); if (rc_ != VQCPC_OK) return rc_; } while (0)

struct vqcpc_melfront {
    int sr, n_fft, n_mels, hop, win;
    float fmin, preemph, top_db;
    int Kp, nbins, nbp, nmp;              // padded window, bins, padded bins, padded mels
    float *dftW = nullptr, *melW = nullptr;
    DevBuf frames, spec, mag, melraw, peak, maxdb, lens;
};

__global__ void mf_peak_kernel(const float *__restrict__ wav, const int *__restrict__ len, int Lmax, unsigned *peak) {
    const int b = blockIdx.y;
    float m = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len[b]; i += gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(wav[(size_t)b * Lmax + i]));
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(&peak[b], __float_as_uint(m));       // non-negative floats order like uints
}

__global__ void mf_frame_kernel(const float *__restrict__ wav, const int *__restrict__ len, const unsigned *__restrict__ peak,
                                int Lmax, int Tmax, int hop, int win, int Kp, float preemph, float *__restrict__ F, size_t rows) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= rows * Kp) return;                               // the grid is rounded up to 256 threads
    const int j = (int)(id % Kp);
    const size_t row = id / Kp;
    const int n = (int)(row % Tmax), b = (int)(row / Tmax);
    const int L = len[b];
    float v = 0.f;
    if (j < win && L >= 2 && n < 1 + L / hop) {
        int i = n * hop + j - win / 2;                         // centred frame, centred window
        while (i < 0 || i >= L) { if (i < 0) i = -i; if (i >= L) i = 2 * (L - 1) - i; }   // reflect padding
        const float *x = wav + (size_t)b * Lmax;
        const float s = 0.999f / __uint_as_float(peak[b]);
        v = x[i] * s - (i > 0 ? preemph * (x[i - 1] * s) : 0.f);
    }
    F[id] = v;
}

__global__ void mf_mag_kernel(const float *__restrict__ spec, int nbins, int nbp, float *__restrict__ mag, size_t rows) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= rows * nbp) return;
    const int k = (int)(id % nbp);
    const size_t row = id / nbp;
    float m = 0.f;
    if (k < nbins) {
        const float re = spec[row * 2 * nbp + k], im = spec[row * 2 * nbp + nbp + k];
        m = sqrtf(re * re + im * im);
    }
    mag[id] = m;
}

__device__ __forceinline__ unsigned ordered_key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_val(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__global__ void mf_db_kernel(float *__restrict__ melraw, const int *__restrict__ len, int Tmax, int hop, int n_mels, int nmp,
                             unsigned *maxdb, size_t rows) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = id < rows * nmp;
    int b = -1;
    unsigned key = 0u;
    if (valid) {
        const int m = (int)(id % nmp);
        const size_t row = id / nmp;
        const int n = (int)(row % Tmax);
        b = (int)(row / Tmax);
        valid = m < n_mels && n < 1 + len[b] / hop && len[b] >= 2;
        if (valid) {
            const float a = melraw[id];
            const float l = 10.0f * log10f(fmaxf(1e-10f, a * a));      // amplitude_to_db(amin=1e-5, ref=1)
            melraw[id] = l;
            key = ordered_key(l);
        }
    }
    // the utterance maximum: one atomic per wave instead of one per element (every element of a 2 s utterance hammering the
    // same word took 8.5 ms per call: profiles/r03_bench_default_kernel_stats.csv) -- a wave almost always covers one utterance
    const unsigned long long act = __ballot(valid);
    if (act == 0ull) return;
    const int first = __ffsll((long long)act) - 1;
    const int b0 = __shfl(b, first);
    if (__all(!valid || b == b0)) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) key = max(key, (unsigned)__shfl_xor((int)key, off));
        if ((int)(threadIdx.x & 63) == first) atomicMax(&maxdb[b0], key);
    } else if (valid) atomicMax(&maxdb[b], key);
}

__global__ void mf_final_kernel(const float *__restrict__ melraw, const int *__restrict__ len, const unsigned *__restrict__ maxdb,
                                int Tmax, int hop, int n_mels, int nmp, float top_db, float *__restrict__ out, int B) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)B * n_mels * Tmax) return;
    const int n = (int)(id % Tmax);
    const int m = (int)((id / Tmax) % n_mels), b = (int)(id / ((size_t)Tmax * n_mels));
    float v = 0.f;
    if (len[b] >= 2 && n < 1 + len[b] / hop) {
        const float l = melraw[((size_t)b * Tmax + n) * nmp + m];
        v = fmaxf(l, ordered_val(maxdb[b]) - top_db) / top_db + 1.0f;
    }
    out[id] = v;                                               // (B, n_mels, Tmax): the encoder's input layout
}

static double hz2mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel2hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}

extern "C" void vqcpc_melfront_destroy(vqcpc_melfront *f) {
    if (!f) return;
    if (f->dftW) (void)hipFree(f->dftW);
    if (f->melW) (void)hipFree(f->melW);
    DevBuf *bufs[] = {&f->frames, &f->spec, &f->mag, &f->melraw, &f->peak, &f->maxdb, &f->lens};
    for (DevBuf *b : bufs) b->release();
    delete f;
}

extern "C" int vqcpc_melfront_create(int sr, int n_fft, int n_mels, int hop, int win, float fmin, float preemph,
                                     float top_db, vqcpc_melfront **out) {
    VQ_REQUIRE(out, "vqcpc_melfront_create: null argument");
    *out = nullptr;
    TRY(vq_require_gfx950());
    VQ_REQUIRE(sr > 0 && n_fft >= 64 && n_fft % 2 == 0 && win > 0 && win <= n_fft && hop > 0 && n_mels > 0 && n_mels <= 1024 &&
               top_db > 0, "vqcpc_melfront_create: bad parameter");
    vqcpc_melfront *f = new vqcpc_melfront();
    f->sr = sr; f->n_fft = n_fft; f->n_mels = n_mels; f->hop = hop; f->win = win; f->fmin = fmin; f->preemph = preemph; f->top_db = top_db;
    f->Kp = (win + 31) / 32 * 32; f->nbins = n_fft / 2 + 1; f->nbp = (f->nbins + 31) / 32 * 32; f->nmp = (n_mels + 63) / 64 * 64;
    const int Kp = f->Kp, nb = f->nbins, nbp = f->nbp, nmp = f->nmp;
    // [cos | sin] matrix with the periodic Hann window folded in (magnitudes do not see the frame-offset phase)
    std::vector<float> W((size_t)2 * nbp * Kp, 0.f);
    for (int k = 0; k < nb; ++k)
        for (int j = 0; j < win; ++j) {
            const double w = 0.5 - 0.5 * cos(2.0 * M_PI * j / win);
            const double ph = 2.0 * M_PI * (double)((long long)k * j % n_fft) / n_fft;
            W[(size_t)k * Kp + j] = (float)(w * cos(ph));
            W[(size_t)(nbp + k) * Kp + j] = (float)(-w * sin(ph));
        }
    // Slaney mel filterbank (librosa.filters.mel, htk=False, norm='slaney'), float32 like librosa's
    std::vector<float> Mw((size_t)nmp * nbp, 0.f);
    std::vector<double> mf(n_mels + 2);
    const double m0 = hz2mel(fmin), m1 = hz2mel(sr / 2.0);
    for (int i = 0; i < n_mels + 2; ++i) mf[i] = mel2hz(m0 + (m1 - m0) * i / (n_mels + 1));
    for (int i = 0; i < n_mels; ++i) {
        const double enorm = 2.0 / (mf[i + 2] - mf[i]);
        for (int k = 0; k < nb; ++k) {
            const double fr = (sr / 2.0) * k / (nb - 1);
            const double lower = (fr - mf[i]) / (mf[i + 1] - mf[i]), upper = (mf[i + 2] - fr) / (mf[i + 2] - mf[i + 1]);
            const double v = fmax(0.0, lower < upper ? lower : upper);
            Mw[(size_t)i * nbp + k] = (float)(v * enorm);
        }
    }
    int rc = VQCPC_OK;
    if (hipMalloc((void **)&f->dftW, W.size() * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&f->melW, Mw.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(f->dftW, W.data(), W.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(f->melW, Mw.data(), Mw.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        vq_set_error("vqcpc_melfront_create: device allocation or copy failed");
        rc = VQCPC_ERR_ALLOC;
    }
    if (rc != VQCPC_OK) { vqcpc_melfront_destroy(f); return rc; }
    *out = f;
    return VQCPC_OK;
}

extern "C" int vqcpc_melfront_frames(const vqcpc_melfront *f, int n_samples) {
    return (f && n_samples >= 2) ? 1 + n_samples / f->hop : 0;
}

extern "C" int vqcpc_melfront_run(vqcpc_melfront *f, const float *wav, const int *lens, int B, int Lmax, float *mel,
                                  void *stream) {
    VQ_REQUIRE(f && wav && lens && mel && B > 0 && Lmax >= 2, "vqcpc_melfront_run: bad argument");
    hipStream_t s = (hipStream_t)stream;
    for (int b = 0; b < B; ++b) VQ_REQUIRE(lens[b] >= 0 && lens[b] <= Lmax, "vqcpc_melfront_run: lens[%d] = %d outside [0, %d]", b, lens[b], Lmax);
    const int Tmax = 1 + Lmax / f->hop, Kp = f->Kp, nbp = f->nbp, nmp = f->nmp;
    const size_t rows = (size_t)B * Tmax;
    VQ_REQUIRE(rows < (1u << 30), "vqcpc_melfront_run: batch too large");
    TRY(f->lens.reserve(B * sizeof(int)));
    TRY(f->peak.reserve(B * sizeof(unsigned)));
    TRY(f->maxdb.reserve(B * sizeof(unsigned)));
    TRY(f->frames.reserve(rows * Kp * sizeof(float)));
    TRY(f->spec.reserve(rows * 2 * nbp * sizeof(float)));
    TRY(f->mag.reserve(rows * nbp * sizeof(float)));
    TRY(f->melraw.reserve(rows * nmp * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(f->lens.p, lens, B * sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));                          // lens is the caller's host buffer
    HIP_TRY(hipMemsetAsync(f->peak.p, 0, B * sizeof(unsigned), s));
    HIP_TRY(hipMemsetAsync(f->maxdb.p, 0, B * sizeof(unsigned), s));
    const int *dl = f->lens.as<int>();
    hipLaunchKernelGGL(mf_peak_kernel, dim3(64, B), dim3(256), 0, s, wav, dl, Lmax, f->peak.as<unsigned>());
    hipLaunchKernelGGL(mf_frame_kernel, dim3((unsigned)((rows * Kp + 255) / 256)), dim3(256), 0, s, wav, dl,
                       f->peak.as<unsigned>(), Lmax, Tmax, f->hop, f->win, Kp, f->preemph, f->frames.as<float>(), rows);
    TRY(vq_gemm_chain(f->frames.as<float>(), Kp, f->dftW, nullptr, f->spec.as<float>(), 2 * nbp, (int)rows, 2 * nbp, Kp, Kp, s));
    hipLaunchKernelGGL(mf_mag_kernel, dim3((unsigned)((rows * nbp + 255) / 256)), dim3(256), 0, s, f->spec.as<float>(),
                       f->nbins, nbp, f->mag.as<float>(), rows);
    TRY(vq_gemm_chain(f->mag.as<float>(), nbp, f->melW, nullptr, f->melraw.as<float>(), nmp, (int)rows, nmp, nbp, nbp, s));
    hipLaunchKernelGGL(mf_db_kernel, dim3((unsigned)((rows * nmp + 255) / 256)), dim3(256), 0, s, f->melraw.as<float>(), dl,
                       Tmax, f->hop, f->n_mels, nmp, f->maxdb.as<unsigned>(), rows);
    const size_t nout = (size_t)B * f->n_mels * Tmax;
    hipLaunchKernelGGL(mf_final_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, s, f->melraw.as<float>(), dl,
                       f->maxdb.as<unsigned>(), Tmax, f->hop, f->n_mels, nmp, f->top_db, mel, B);
    HIP_TRY(hipGetLastError());
    return VQCPC_OK;
}
