// Internal helpers shared by the HIP translation units of libvqcpc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/vqcpc.h"

void vq_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            vq_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,  \
                         __LINE__);                                                        \
            return VQCPC_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)

#define VQ_REQUIRE(cond, ...)                   \
    do {                                        \
        if (!(cond)) {                          \
            vq_set_error(__VA_ARGS__);          \
            return VQCPC_ERR_INVALID;           \
        }                                       \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Grow-only device buffer.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return VQCPC_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            vq_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
            return VQCPC_ERR_ALLOC;
        }
        cap = bytes;
        return VQCPC_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return (T *)p; }
};

// ---- exact-chain GEMM (encoder.hip), also used for the hoisted projections of the vocoder.
// Y[m, n] = fold over K blocks of KC of an fp32 fma chain (k ascending, from 0) of
// A[m, k] * W[n, k]; first block: (bias ? bias[n] + chain : chain), later: tot + chain.
// A dense row-major (lda) ; W (N, K) row-major ; N % 64 == 0 ; K % 32 == 0 ; KC % 32 == 0.
int vq_gemm_chain(const float *A, int lda, const float *W, const float *bias, float *Y, int ldy,
                  int M, int N, int K, int KC, hipStream_t s);

// The same with an epilogue: optional ReLU, and an optional output row map for chunked scans -- row m is stored at
// (m / ydiv) * ystride + yoff + m % ydiv, or not at all when yoff + m % ydiv >= ylim (ydiv == 0: plain row m).
int vq_gemm_chain_ex(const float *A, int lda, const float *W, const float *bias, float *Y, int ldy,
                     int M, int N, int K, int KC, int relu, int ydiv, int ystride, int yoff, int ylim, hipStream_t s);

// ---- recurrent machinery (vocoder.hip), shared with the encoder's LSTM.
struct LstmPlan;   // opaque, owns fragment-ordered weights
int vq_lstm_plan_create(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                        int D, int H, LstmPlan **out);
void vq_lstm_plan_destroy(LstmPlan *p);
// x (B, T, D) device -> out (B, T, H) device, zero initial state.
int vq_lstm_run(LstmPlan *p, const float *x, int B, int T, float *out, hipStream_t s);
int vq_lstm_set_persistent(LstmPlan *p, int value);                  // -1 auto (one utterance: the resident scan), 0 one launch per time step
int vq_lstm_set_debug(LstmPlan *p, int drop_step, int timeout_ms);   // tests of the abort path
int vq_lstm_check(LstmPlan *p);            // after the caller's sync: did the resident scan of the last call time out?
