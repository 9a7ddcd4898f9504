// resample.hip -- the resampling inside librosa.load(path, sr=...) (reference: convert.py:54-56) on gfx950.
//
// librosa ^0.8 resamples with res_type "kaiser_best" = resampy's band-limited sinc interpolation: a half sinc window
// (64 zero crossings, 512 table entries per crossing) tapered by a Kaiser window, linearly interpolated between table
// entries, stretched and scaled by the rate ratio when down-sampling.  oracle/resample_ref.py restates the algorithm
// and its constants (resampy is absent offline: parity unpinned).  Arithmetic in fp64 like resampy, result rounded to
// fp32 like librosa.  One thread per output sample; the two filter wings are gathers from an L2-resident table.
#include "common.h"
#include <math.h>
#include <vector>

int vq_require_gfx950();
#define TRY(x) do { int rc_ = (x); if (rc_ != VQCPC_OK) return rc_; } while (0)

struct vqcpc_resampler {
    int sr_in, sr_out;
    double ratio, scale, time_increment;
    int num_table, index_step, nwin;
    double *win = nullptr, *delta = nullptr;     // device, nwin entries each
    // resampy advances its output clock by repeated fp64 addition (time_register += time_increment); t * increment
    // differs from that sum in the last bits, and where the clock lands within rounding of an integer the two pick
    // different filter phases (resampy's integer index step makes that a 5e-4 jump).  So the clock values are produced
    // by the same sequential additions on the host, once, and kept on the device (grow-only).
    DevBuf treg;
    int treg_len = 0;
};

#define RS_MAXB 256
struct RsLens { int n_in[RS_MAXB]; };

__global__ __launch_bounds__(256) void resample_kernel(const float *__restrict__ x, float *__restrict__ y, RsLens lens,
                                                       int Lin_max, int Lout_max, const double *__restrict__ win,
                                                       const double *__restrict__ delta, int nwin, int num_table, int index_step,
                                                       double ratio, double scale, const double *__restrict__ treg) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (t >= Lout_max) return;
    const int n_orig = lens.n_in[b];
    const int n_res = (int)((double)n_orig * ratio);            // resampy's output length; librosa zero-pads up to ceil
    float out = 0.f;
    if (t < n_res) {
        const float *xb = x + (size_t)b * Lin_max;
        const double time_register = treg[t];
        const int n = (int)time_register;
        double acc = 0.0;
        double frac = scale * (time_register - (double)n);
        double index_frac = frac * (double)num_table;
        int offset = (int)index_frac;
        double eta = index_frac - (double)offset;
        int i_max = (nwin - offset) / index_step;
        i_max = i_max < n + 1 ? i_max : n + 1;
        for (int i = 0; i < i_max; ++i) {                       // left wing
            const int k = offset + i * index_step;
            acc += (win[k] + eta * delta[k]) * (double)xb[n - i];
        }
        frac = scale - frac;
        index_frac = frac * (double)num_table;
        offset = (int)index_frac;
        eta = index_frac - (double)offset;
        int k_max = (nwin - offset) / index_step;
        k_max = k_max < n_orig - n - 1 ? k_max : n_orig - n - 1;
        for (int i = 0; i < k_max; ++i) {                       // right wing
            const int k = offset + i * index_step;
            acc += (win[k] + eta * delta[k]) * (double)xb[n + i + 1];
        }
        out = (float)acc;
    }
    y[(size_t)b * Lout_max + t] = out;
}

static double bessel_i0(double x) {                              // power series; x <= ~20 here
    double sum = 1.0, term = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 200; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}

extern "C" void vqcpc_resampler_destroy(vqcpc_resampler *r) {
    if (!r) return;
    if (r->win) (void)hipFree(r->win);
    if (r->delta) (void)hipFree(r->delta);
    r->treg.release();
    delete r;
}

extern "C" int vqcpc_resampler_create(int sr_in, int sr_out, vqcpc_resampler **out) {
    VQ_REQUIRE(out, "vqcpc_resampler_create: null argument");
    *out = nullptr;
    TRY(vq_require_gfx950());
    VQ_REQUIRE(sr_in > 0 && sr_out > 0 && sr_in <= 768000 && sr_out <= 768000, "vqcpc_resampler_create: bad sample rate");
    // resampy "kaiser_best": see oracle/resample_ref.py
    const int num_zeros = 64, precision = 9;
    const double beta = 14.769656459379492, rolloff = 0.9475937167399596;
    vqcpc_resampler *r = new vqcpc_resampler();
    r->sr_in = sr_in; r->sr_out = sr_out;
    r->ratio = (double)sr_out / (double)sr_in;
    r->scale = r->ratio < 1.0 ? r->ratio : 1.0;
    r->time_increment = 1.0 / r->ratio;
    r->num_table = 1 << precision;
    r->index_step = (int)(r->scale * r->num_table);
    const int n = r->num_table * num_zeros;
    r->nwin = n + 1;
    if (r->index_step < 1) { delete r; vq_set_error("vqcpc_resampler_create: rate ratio %g too small", r->ratio); return VQCPC_ERR_INVALID; }
    std::vector<double> win(n + 1), delta(n + 1, 0.0);
    const double i0b = bessel_i0(beta);
    for (int i = 0; i <= n; ++i) {
        const double xz = (double)num_zeros * (double)i / (double)n;           // np.linspace(0, num_zeros, n + 1)
        const double a = rolloff * xz;
        const double sinc = a == 0.0 ? 1.0 : sin(M_PI * a) / (M_PI * a);       // np.sinc
        const double u = (double)i / (double)n;                                 // np.kaiser(2n + 1, beta)[n + i]
        const double taper = bessel_i0(beta * sqrt(1.0 - u * u > 0.0 ? 1.0 - u * u : 0.0)) / i0b;
        win[i] = taper * rolloff * sinc;
        if (r->ratio < 1.0) win[i] *= r->ratio;
    }
    for (int i = 0; i < n; ++i) delta[i] = win[i + 1] - win[i];
    int rc = VQCPC_OK;
    if (hipMalloc((void **)&r->win, win.size() * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&r->delta, delta.size() * sizeof(double)) != hipSuccess ||
        hipMemcpy(r->win, win.data(), win.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(r->delta, delta.data(), delta.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        vq_set_error("vqcpc_resampler_create: device allocation or copy failed");
        rc = VQCPC_ERR_ALLOC;
    }
    if (rc != VQCPC_OK) { vqcpc_resampler_destroy(r); return rc; }
    *out = r;
    return VQCPC_OK;
}

extern "C" int vqcpc_resampler_out_len(const vqcpc_resampler *r, int n_in) {
    if (!r || n_in < 0) return 0;
    return (int)ceil((double)n_in * r->ratio);
}

extern "C" int vqcpc_resampler_run(vqcpc_resampler *r, const float *wav_in, const int *lens_in, int B, int Lin_max,
                                   float *wav_out, int Lout_max, void *stream) {
    VQ_REQUIRE(r && wav_in && lens_in && wav_out && B > 0 && Lin_max > 0 && Lout_max > 0, "vqcpc_resampler_run: bad argument");
    for (int b = 0; b < B; ++b) {
        VQ_REQUIRE(lens_in[b] >= 0 && lens_in[b] <= Lin_max, "vqcpc_resampler_run: lens[%d] = %d outside [0, %d]", b, lens_in[b], Lin_max);
        VQ_REQUIRE(vqcpc_resampler_out_len(r, lens_in[b]) <= Lout_max, "vqcpc_resampler_run: output row too short for utterance %d", b);
    }
    hipStream_t s = (hipStream_t)stream;
    if (Lout_max > r->treg_len) {                               // grow the clock table (rare: synchronises once)
        int want = 1 << 16;
        while (want < Lout_max) want <<= 1;
        std::vector<double> tr((size_t)want);
        double acc = 0.0;
        for (int t = 0; t < want; ++t) { tr[t] = acc; acc += r->time_increment; }
        HIP_TRY(hipStreamSynchronize(s));                       // earlier launches may still read the old table
        TRY(r->treg.reserve((size_t)want * sizeof(double)));
        HIP_TRY(hipMemcpy(r->treg.p, tr.data(), (size_t)want * sizeof(double), hipMemcpyHostToDevice));
        r->treg_len = want;
    }
    for (int b0 = 0; b0 < B; b0 += RS_MAXB) {                    // lengths travel as kernel arguments: no host-table upload, no sync
        const int nb = B - b0 < RS_MAXB ? B - b0 : RS_MAXB;
        RsLens lens{};
        for (int b = 0; b < nb; ++b) lens.n_in[b] = lens_in[b0 + b];
        hipLaunchKernelGGL(resample_kernel, dim3((Lout_max + 255) / 256, nb), dim3(256), 0, s, wav_in + (size_t)b0 * Lin_max,
                           wav_out + (size_t)b0 * Lout_max, lens, Lin_max, Lout_max, r->win, r->delta, r->nwin, r->num_table,
                           r->index_step, r->ratio, r->scale, r->treg.as<double>());
    }
    HIP_TRY(hipGetLastError());
    return VQCPC_OK;
}
