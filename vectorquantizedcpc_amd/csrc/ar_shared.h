// Device helpers shared by the decode kernels of vocoder.hip (launch-per-step and 64-workgroup persistent decoders)
// and ar_xcd.hip (one resident decoder per XCD).  Everything that decides a BIT of the result lives here once, so that
// an utterance decoded on any of the paths gives the same samples: the Philox stream of the sampling protocol, the
// gate non-linearities, and the order in which a row's fp32 fma chains are loaded and combined.
#pragma once
#include "common.h"
#include <math.h>

typedef unsigned long long u64;

__device__ __forceinline__ u64 ps_load(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ps_store(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// workgroup barrier that also orders LDS traffic around it
__device__ __forceinline__ void ps_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// Philox4x32-10, word `k & 3` of counter (t, utt, k >> 2, 0): the sampling protocol's stream.
__device__ __forceinline__ unsigned philox_word(unsigned c0, unsigned c1, unsigned c2, unsigned k0, unsigned k1, int w) {
    unsigned c[4] = {c0, c1, c2, 0u};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return w == 0 ? c[0] : w == 1 ? c[1] : w == 2 ? c[2] : c[3];
}
// Gumbel noise of (class, utterance, sample): 23 random bits + 0.5 -- every value is exact in fp32 and strictly inside
// (0, 1) (a 24-bit form rounds to 1.0f at the top word, i.e. +inf noise that wins whatever the logit is).
__device__ __forceinline__ float gumbel_from_word(unsigned w) {
    return -logf(-logf(((float)(w >> 9) + 0.5f) * (1.0f / 8388608.0f)));
}

// A row's dot product over K = 64 NS runs as 8 fp32 fma chains, exactly those of the v_mfma_f32_16x16x4_f32 schedule of
// the launch-per-step kernels: K quarter kw (0..3) x accumulator c0 (0: the x/z fragment components, 1: y/w).  Term
// n = 8 s + 4 ci + q of chain (kw, c0) multiplies column k = 16 (kw NS + s) + 4 q + c0 + 2 ci; the row sum is
// ((q0 + q1) + q2) + q3 with q_kw = chain(kw, 0) + chain(kw, 1).
__device__ __forceinline__ int chain_col(int NS, int kw, int c0, int n) {
    return 16 * (kw * NS + (n >> 3)) + 4 * (n & 3) + c0 + 2 * ((n >> 2) & 1);
}
template <int NS>
__device__ __forceinline__ void ps_load_weights(const float *Wrow, int kw, int c0, float (&w)[8 * NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) w[8 * s + 4 * ci + kq] = Wrow[16 * (kw * NS + s) + 4 * kq + c0 + 2 * ci];
}

#define PS_DPP(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xF, 0xF, false))

// ---- in-kernel exchanges of the resident decoders (ar_xcd.hip, ar_xcm.hip): 8-byte {tag, value} granules -- the data is the
// flag.  Published with a store addressed as (uniform 64-bit base in SGPRs) + (32-bit byte offset in a VGPR): workgroup scope
// (sc0: the write-through L1 leaves the granule in this XCD's L2) or agent scope (sc1).  hipcc does not pad the hazard between
// a VALU write of the base SGPRs (v_readfirstlane) and a vector-memory instruction inside an asm statement reading them:
// every such statement opens with the five wait states itself.
__device__ __forceinline__ void xd_put(u64 *base, unsigned byte_off, u64 v, int agent) {
    if (agent) asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2 sc1" :: "v"(byte_off), "v"(v), "s"(base) : "memory");
    else asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2 sc0" :: "v"(byte_off), "v"(v), "s"(base) : "memory");   // stays in this XCD's L2
}

struct Waiter {                      // bounded spinning shared by all sweeps of the kernel
    unsigned *status;
    unsigned ticks;
    u64 t0;
    // The wall clock (s_memrealtime: a scalar memory read, ~0.3 us) is only consulted once a wait has spun 64 times: a wait
    // that succeeds quickly never pays for it.  t0 = 0: not taken yet.
    __device__ __forceinline__ void start() { t0 = 0; }
    // true: give up (deadline passed -- status bit 0 is then set -- or somebody else already gave up).  `tagp`: an LDS word with the
    // call's epoch << 8, so that the status word says WHICH call of the handle gave up; it is read only when the deadline has
    // passed (kept in a register for the whole call, the tag cost ar_xcd_kernel<2> a VGPR spill inside its sample loop)
    __device__ __forceinline__ bool expired(unsigned spins, int lane, const int *tagp) {
        if ((spins & 63) != 63) return false;
        const u64 now = __builtin_amdgcn_s_memrealtime();
        if (t0 == 0) t0 = now;
        const bool late = now - t0 > (u64)ticks;
        if (late && lane == 0) __hip_atomic_store(status, (unsigned)*tagp | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // host-mapped: a plain store, no PCIe atomic
        return late || (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u);
    }
};
