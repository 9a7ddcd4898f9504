// Device helpers of the per-XCD resident decoders on the vector ALU (ar_xcd.hip; the pipelined variant of round 4, measured and
// dropped: profiles/r04_xcp_experiment.md): dimensions, granule loads, and a row's fp32 fma chains on v_fmac_f32_dpp -- the same 8
// chains per row as the MFMA schedule of the launch-per-step kernels (ar_shared.h), combined in the same order.
#pragma once
#include "ar_xcd.h"
#include "ar_shared.h"

namespace {

constexpr int HR = 896, HF = 256, NC = 256;
constexpr int NW = 32;                 // workgroups per XCD
constexpr int UPB = 28;                // hidden units per workgroup
constexpr int ROWS = 3 * UPB;          // W_hh rows per workgroup
constexpr int FPB = 8;                 // fc1 rows / fc2 classes per workgroup
constexpr int THREADS = 768;
constexpr int NT_H = HR / 8;           // terms of a chain over h: 112
constexpr int NT_A = HF / 8;           // terms of a chain over a: 32

// exchange area of one XCD, in granules
__host__ __device__ constexpr int xg_h(int bxt) { return NW * bxt * 32; }         // [rank][slot][32] (28 used: two whole lines)
__host__ __device__ constexpr int xg_a(int bxt) { return NW * bxt * FPB; }        // [rank][slot][8]
__host__ __device__ constexpr int xg_c() { return NW * 16; }                      // [slot < 16][rank]: a slot's 32 candidates are contiguous
__host__ __device__ constexpr int xg_region(int bxt) { return xg_h(bxt) + xg_a(bxt) + xg_c(); }
constexpr int CTL_WORDS = 64;          // u32: arrivals per XCC [0..7], total [8]



// Granule traffic is addressed as (uniform 64-bit base in SGPRs) + (32-bit byte offset in a VGPR) + immediate: a 64-bit
// address per lane and granule costs two VGPRs each, and the chain waves have none to spare.  The loads are sc1 (served by
// L2, not by this CU's L1); each helper issues its loads together and returns when they have landed (hipcc does not count
// the memory operations of an asm statement, so the wait is part of it; nor does it pad the hazard between a VALU write of
// the base SGPRs (v_readfirstlane) and a vector-memory instruction inside the statement reading them: every statement
// opens with the five wait states itself -- without them the stamped build and the 2-slot instantiation faulted).
template <int STEP>
__device__ __forceinline__ void gran_load1(u64 (&v)[1], const u64 *base, unsigned off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v[0]) : "v"(off), "s"(base) : "memory");
}
template <int STEP>
__device__ __forceinline__ void gran_load2(u64 (&v)[2], const u64 *base, unsigned off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %2, %3 sc1\n\tglobal_load_dwordx2 %1, %2, %3 offset:%4 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]) : "v"(off), "s"(base), "i"(STEP) : "memory");
}
template <int STEP>
__device__ __forceinline__ void gran_load4(u64 (&v)[4], const u64 *base, unsigned off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %4, %5 sc1\n\tglobal_load_dwordx2 %1, %4, %5 offset:%6 sc1\n\t"
                 "global_load_dwordx2 %2, %4, %5 offset:%7 sc1\n\tglobal_load_dwordx2 %3, %4, %5 offset:%8 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(off), "s"(base), "i"(STEP), "i"(2 * STEP), "i"(3 * STEP) : "memory");
}
template <int STEP>     // granules at off, off + STEP (from base) and the same two from base2
__device__ __forceinline__ void gran_load4b(u64 (&v)[4], const u64 *base, const u64 *base2, unsigned off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %4, %5 sc1\n\tglobal_load_dwordx2 %1, %4, %5 offset:%7 sc1\n\t"
                 "global_load_dwordx2 %2, %4, %6 sc1\n\tglobal_load_dwordx2 %3, %4, %6 offset:%7 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(off), "s"(base), "s"(base2), "i"(STEP) : "memory");
}
template <int STEP>     // the same for two slots (second slot SLOT2 bytes further): eight granules in flight together
__device__ __forceinline__ void gran_load8b(u64 (&v)[2][4], const u64 *base, const u64 *base2, unsigned off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %8, %9 sc1\n\tglobal_load_dwordx2 %1, %8, %9 offset:%11 sc1\n\t"
                 "global_load_dwordx2 %2, %8, %10 sc1\n\tglobal_load_dwordx2 %3, %8, %10 offset:%11 sc1\n\t"
                 "global_load_dwordx2 %4, %8, %9 offset:128 sc1\n\tglobal_load_dwordx2 %5, %8, %9 offset:%12 sc1\n\t"
                 "global_load_dwordx2 %6, %8, %10 offset:128 sc1\n\tglobal_load_dwordx2 %7, %8, %10 offset:%12 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[0][2]), "=&v"(v[0][3]), "=&v"(v[1][0]), "=&v"(v[1][1]), "=&v"(v[1][2]), "=&v"(v[1][3])
                 : "v"(off), "s"(base), "s"(base2), "i"(STEP), "i"(STEP + 128) : "memory");
}
// 2 N granules: N at `off` + i STEP and N at `off2` + i STEP, all in flight together
template <int N, int STEP>
__device__ __forceinline__ void gran_load_pair(u64 (&a)[N], u64 (&b)[N], const u64 *base, unsigned off, unsigned off2) {
    if constexpr (N == 1)
        asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %2, %4 sc1\n\tglobal_load_dwordx2 %1, %3, %4 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(a[0]), "=&v"(b[0]) : "v"(off), "v"(off2), "s"(base) : "memory");
    if constexpr (N == 2)
        asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %4, %6 sc1\n\tglobal_load_dwordx2 %1, %4, %6 offset:%7 sc1\n\t"
                     "global_load_dwordx2 %2, %5, %6 sc1\n\tglobal_load_dwordx2 %3, %5, %6 offset:%7 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(a[0]), "=&v"(a[1]), "=&v"(b[0]), "=&v"(b[1]) : "v"(off), "v"(off2), "s"(base), "i"(STEP) : "memory");
    if constexpr (N == 4)
        asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %8, %10 sc1\n\tglobal_load_dwordx2 %1, %8, %10 offset:%11 sc1\n\t"
                     "global_load_dwordx2 %2, %8, %10 offset:%12 sc1\n\tglobal_load_dwordx2 %3, %8, %10 offset:%13 sc1\n\t"
                     "global_load_dwordx2 %4, %9, %10 sc1\n\tglobal_load_dwordx2 %5, %9, %10 offset:%11 sc1\n\t"
                     "global_load_dwordx2 %6, %9, %10 offset:%12 sc1\n\tglobal_load_dwordx2 %7, %9, %10 offset:%13 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
                     : "v"(off), "v"(off2), "s"(base), "i"(STEP), "i"(2 * STEP), "i"(3 * STEP) : "memory");
}
template <int N, int STEP>
__device__ __forceinline__ void gran_load(u64 (&v)[N], const u64 *base, unsigned off) {
    if constexpr (N == 1) gran_load1<STEP>(v, base, off);
    if constexpr (N == 2) gran_load2<STEP>(v, base, off);
    if constexpr (N == 4) gran_load4<STEP>(v, base, off);
}

// acc += (value of `h` in lane J of this lane's quad) * w -- one fp32 fma, as the MFMA chain does it.  Eight (four)
// consecutive terms of a chain go into ONE asm statement: hipcc pads every asm statement with an s_nop, which at one
// fmac per statement doubled the instruction count of a chain (13 cycles per term measured; the terms of one statement
// need no padding among themselves: the accumulator is an ordinary operand, the DPP operand comes from LDS loads).
#define XD_QP(J) "quad_perm:[" #J "," #J "," #J "," #J "] row_mask:0xf bank_mask:0xf"
#define XD_FMAC8(J)                                                                                                      \
    asm("v_fmac_f32_dpp %0, %1, %9 " XD_QP(J) "\n\tv_fmac_f32_dpp %0, %2, %10 " XD_QP(J) "\n\t"                        \
        "v_fmac_f32_dpp %0, %3, %11 " XD_QP(J) "\n\tv_fmac_f32_dpp %0, %4, %12 " XD_QP(J) "\n\t"                       \
        "v_fmac_f32_dpp %0, %5, %13 " XD_QP(J) "\n\tv_fmac_f32_dpp %0, %6, %14 " XD_QP(J) "\n\t"                       \
        "v_fmac_f32_dpp %0, %7, %15 " XD_QP(J) "\n\tv_fmac_f32_dpp %0, %8, %16 " XD_QP(J)                              \
        : "+v"(acc)                                                                                                      \
        : "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]), "v"(h[4]), "v"(h[5]), "v"(h[6]), "v"(h[7]),                       \
          "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]))
#define XD_FMAC4(J)                                                                                                      \
    asm("v_fmac_f32_dpp %0, %1, %5 " XD_QP(J) "\n\tv_fmac_f32_dpp %0, %2, %6 " XD_QP(J) "\n\t"                         \
        "v_fmac_f32_dpp %0, %3, %7 " XD_QP(J) "\n\tv_fmac_f32_dpp %0, %4, %8 " XD_QP(J)                                \
        : "+v"(acc)                                                                                                      \
        : "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]))
// terms 8 J .. 8 J + 7 of a 32-term phase: operands h[0..7] as held by quad lane J, weights w[0..7]
template <int J>
__device__ __forceinline__ void fmac8(float &acc, const float *h, const float *w) {
    if constexpr (J == 0) XD_FMAC8(0);
    if constexpr (J == 1) XD_FMAC8(1);
    if constexpr (J == 2) XD_FMAC8(2);
    if constexpr (J == 3) XD_FMAC8(3);
}
template <int J>
__device__ __forceinline__ void fmac4(float &acc, const float *h, const float *w) {
    if constexpr (J == 0) XD_FMAC4(0);
    if constexpr (J == 1) XD_FMAC4(1);
    if constexpr (J == 2) XD_FMAC4(2);
    if constexpr (J == 3) XD_FMAC4(3);
}
// one phase: len = 32 (8 terms per quad lane) or 16 (4 per lane); w = the phase's weights in term order
template <int LEN>
__device__ __forceinline__ void chain_phase(float &acc, const float (&hv)[8], const float *w) {
    if constexpr (LEN == 32) { fmac8<0>(acc, hv, w); fmac8<1>(acc, hv, w + 8); fmac8<2>(acc, hv, w + 16); fmac8<3>(acc, hv, w + 24); }
    else { fmac4<0>(acc, hv, w); fmac4<1>(acc, hv, w + 4); fmac4<2>(acc, hv, w + 8); fmac4<3>(acc, hv, w + 12); }
}

// Lane layout of a chain wave: lane = 16 R + 4 kw + j; R = 2 rq + c0.  The quad (j = 0..3) holds chain (kw, c0) of rows
// 4 rq + j.  Operand values of a chain sit in LDS chain by chain ([cid = 2 kw + c0][...]); they are used in PHASES of 32
// terms: in phase ph quad lane j holds terms 32 ph + 8 j .. + 7 (two 16-byte LDS words), so term n comes from quad lane
// (n % 32) / 8, register n % 8 -- 8 operand registers per phase instead of NT / 4 for the whole chain.  A last phase of 16
// terms (NT = 112) gives every lane 4.
// LDS order inside a phase: the four lanes' FIRST words, then their SECOND words (term 32 ph + 8 j + r sits at 32 ph + 16 (r / 4)
// + 4 j + r % 4), so that one ds_read_b128 of a quad covers 16 contiguous floats and the four chains of a 16-lane group (chain
// stride 112 or 48 floats = 48 mod 64 banks) cover all 64 banks: with the terms in plain order (lane j at 8 j) a quad's read had
// holes and chains cid / cid + 4 shared banks -- rocprofv3 counted 59 % of the LDS cycles of ar_xcd_kernel<4> as bank conflicts
// (profiles/r04_pmc_sq_xcd32.json, round 4's first pass).
// (the 16-term last phase of a 112-term chain keeps its plain order: lane j's one word at 4 j)
template <int NT>
__device__ __forceinline__ int phase_pos_t(int n) { return (NT % 32 != 0 && n >= NT - NT % 32) ? n : ((n & ~31) + ((n >> 2) & 1) * 16 + ((n >> 3) & 3) * 4 + (n & 3)); }
template <int NT>
__device__ __forceinline__ float chain_regs(const float *w, const float *opnd) {
    float acc = 0.f;
    constexpr int NPH = (NT + 31) / 32;
    float4 cur[2], nxt[2];                              // opnd = base + cid * stride + 4 j: this lane's first word of phase 0
    cur[0] = *(const float4 *)opnd; cur[1] = *(const float4 *)(opnd + 16);
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        const int len = NT - 32 * ph < 32 ? NT - 32 * ph : 32;          // 32, or 16 in the last phase of 112
        if (ph + 1 < NPH) {
            const int nlen = NT - 32 * (ph + 1) < 32 ? NT - 32 * (ph + 1) : 32;
            nxt[0] = *(const float4 *)(opnd + 32 * (ph + 1));
            // a 16-term phase: lane j holds terms 4 j .. 4 j + 3 = its first word; there is no second
            nxt[1] = nlen == 32 ? *(const float4 *)(opnd + 32 * (ph + 1) + 16) : nxt[0];
        }
        const float hv[8] = {cur[0].x, cur[0].y, cur[0].z, cur[0].w, cur[1].x, cur[1].y, cur[1].z, cur[1].w};
        if (len == 32) chain_phase<32>(acc, hv, w + 32 * ph);
        else chain_phase<16>(acc, hv, w + 32 * ph);
        cur[0] = nxt[0]; cur[1] = nxt[1];
    }
    return acc;
}
// The same chain for ONE or TWO operand vectors (two decode slots) with its weights streamed from LDS a phase ahead of their use: a
// service wave must not hold 112 weights next to everything else it keeps.  The weights of a lane's chain are NT / 4 16-byte words,
// word i at wp[i * WS]: the words of the WS lane-chains of a wave are interleaved ([word][lane-chain]), so that a wave's read of word
// i is one contiguous run -- with a lane's chain contiguous ([lane-chain][NT], 448 bytes apart) the sixteen lanes of a read group hit
// four bank windows four ways each.  w0 = the weights of phase 0, already in registers (requested before the barrier the chain waits
// behind).
__device__ __forceinline__ void load_phase(const float *opnd, int ph, int nlen, float4 (&d)[2]) {
    d[0] = *(const float4 *)(opnd + 32 * ph);
    d[1] = nlen == 32 ? *(const float4 *)(opnd + 32 * ph + 16) : d[0];     // 16-term phase: lane j holds terms 4 j .. 4 j + 3 = the first word
}
template <int NT, int WS>
__device__ __forceinline__ void chain_lds2(const float4 *wp, const float4 (&w0)[8], const float *opA, const float *opB, bool two,
                                           float &accA, float &accB) {
    constexpr int NPH = (NT + 31) / 32;
    float4 curA[2], nxtA[2], curB[2], nxtB[2], wc[8], wn[8];
    load_phase(opA, 0, NT < 32 ? NT : 32, curA);
    if (two) load_phase(opB, 0, NT < 32 ? NT : 32, curB);
    else { curB[0] = curA[0]; curB[1] = curA[1]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) wc[i] = w0[i];
    accA = 0.f; accB = 0.f;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        const int len = NT - 32 * ph < 32 ? NT - 32 * ph : 32;
        if (ph + 1 < NPH) {
            const int nlen = NT - 32 * (ph + 1) < 32 ? NT - 32 * (ph + 1) : 32;
            load_phase(opA, ph + 1, nlen, nxtA);
            if (two) load_phase(opB, ph + 1, nlen, nxtB);
#pragma unroll
            for (int i = 0; i < nlen / 4; ++i) wn[i] = wp[(8 * (ph + 1) + i) * WS];
        }
        __builtin_amdgcn_sched_barrier(0);
        float wv[32];
#pragma unroll
        for (int i = 0; i < 8; ++i) { wv[4 * i] = wc[i].x; wv[4 * i + 1] = wc[i].y; wv[4 * i + 2] = wc[i].z; wv[4 * i + 3] = wc[i].w; }
        {
            const float hv[8] = {curA[0].x, curA[0].y, curA[0].z, curA[0].w, curA[1].x, curA[1].y, curA[1].z, curA[1].w};
            if (len == 32) chain_phase<32>(accA, hv, wv);
            else chain_phase<16>(accA, hv, wv);
        }
        if (two) {
            const float hv[8] = {curB[0].x, curB[0].y, curB[0].z, curB[0].w, curB[1].x, curB[1].y, curB[1].z, curB[1].w};
            if (len == 32) chain_phase<32>(accB, hv, wv);
            else chain_phase<16>(accB, hv, wv);
        }
        __builtin_amdgcn_sched_barrier(0);
        curA[0] = nxtA[0]; curA[1] = nxtA[1]; curB[0] = nxtB[0]; curB[1] = nxtB[1];
#pragma unroll
        for (int i = 0; i < 8; ++i) wc[i] = wn[i];
    }
}

// the lane of a chain wave that holds chain cc (= 2 kw + c0) of the wave's row rr (0..7): inverse of the lane layout above
__device__ __forceinline__ unsigned xd_lane_of(unsigned rr, unsigned cc) { return 16u * (2u * (rr >> 2) + (cc & 1u)) + 4u * (cc >> 1) + (rr & 3u); }

// Row sum from the 8 chain lanes of a row: a0 + a1 across the two 16-lane rows of an rq pair (lane ^ 16), then
// ((q0 + q1) + q2) + q3 along the K quarters (lane + 4, + 8, + 12 inside the row) -- the order of the MFMA kernels.
// Meaningful in lanes with kw == 0 of the c0 = 0 rows (lanes 0..3 and 32..35: sum_lane).  The partner row comes through
// v_permlane16_swap_b32 (gfx950: the second operand's rows 0 and 2 receive the first operand's rows 1 and 3), not ds_swizzle: a
// swizzle is an LDS round trip, ~0.1 us behind the other waves' reads -- and four in a row, one per accumulator of a matrix-pipe
// chain, were 0.4 us of the step's critical path (fc1 -> a_t).
__device__ __forceinline__ float chain_combine(float acc) {
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc), __float_as_uint(acc), false, false);
    const float o = __uint_as_float(sw[1]);                                                        // rows 0 and 2: lane + 16
    const float q = acc + o;
    // ((q + q1) + q2) + q3, q_k = q of lane + 4 k: three v_add_f32_dpp (the shifted operand read in place) instead of three dpp moves and
    // three adds; s_nop 1: the two wait states between the add that wrote q and its first dpp read (hipcc does not look into the statement)
    float s;                                            // early clobber: q must stay what the second and third add shift
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_shl:12 row_mask:0xf bank_mask:0xf" : "=&v"(s) : "v"(q));
    return s;
}

// position of operand column k in the LDS copy (inverse of chain_col): chain cid = 2 kw + c0 at cid * stride, term n at phase_pos(n)
template <int NS>
__device__ __forceinline__ int chain_pos(int k, int stride) {
    const int S = k >> 4, kw = S / NS, s = S - kw * NS;
    const int q = (k >> 2) & 3, c0 = k & 1, ci = (k >> 1) & 1;
    return (2 * kw + c0) * stride + phase_pos_t<8 * NS>(8 * s + 4 * ci + q);
}

// ---- the same chains on the matrix pipe, for the FOUR slots of an XCD at once (ar_xcd_kernel<4>).  v_mfma_f32_4x4x1_16B_f32 is
// sixteen independent 4x4 outer products, one per quad: lane 4 b + i supplies A_b[i], lane 4 b + j supplies B_b[j], and
// D_b[i][j] += A_b[i] * B_b[j] (one rounding: bit for bit fmaf -- tools/microbench_mfma4x4.hip) lands in register i of lane
// 4 b + j.  With the chain waves' lane layout (quad = chain (kw, c0) of rows 4 rq + 0..3) A is the lane's pinned weight of term
// n -- exactly what v_fmac_f32_dpp used -- and B is term n's operand of slot j = lane & 3: one instruction advances the chains of
// 4 rows x 4 slots, where the vector ALU needed 4 (one per slot).  Why: with three waves per SIMD a wave's 112-term dpp chain
// costs the SIMD 0.27 us whoever waits for it (tools/microbench_chain.hip: 0.80 us per pass for the last of three waves), the
// twelve passes of a 4-slot step 3.3 us -- the step was vector-ALU bound behind barrier A (tools/xcd_barriers.py: the youngest
// wave of every SIMD reaches barrier B 3.6 us after A, with or without the exchange waits); the matrix pipe does the four
// slots' chains of three waves in 1.5 us and leaves the vector ALU to the serial work.
// Operands: the lane reads ITS slot's copy of the chain itself, terms in plain order (28 ds_read_b128 per step -- as many as the
// dpp form's 4 x 7): hc[j * HS4 + cid * NT_H + n].  Slot stride HS4 = HR + 4: the four lanes of a quad start one 16-byte unit
// apart, the four quads of a ds_read_b128 lane group (chains {0,6,3,5} or {2,4,1,7}, 28 units apart) 4 units apart -- 64 banks.
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int HS4 = HR + 4;
constexpr int NG_H = NT_H / 8;         // groups of 8 terms (two 16-byte operand words)
template <int NS>
__device__ __forceinline__ int chain_pos_plain(int k, int stride) {
    const int S = k >> 4, kw = S / NS, s = S - kw * NS;
    const int q = (k >> 2) & 3, c0 = k & 1, ci = (k >> 1) & 1;
    return (2 * kw + c0) * stride + 8 * s + 4 * ci + q;
}
// groups [G0, G1) of the chain, weights pinned (w[n] = term n); cur = the operands of group G0, already requested (it leaves
// with those of group G1: a chain can pause between two calls without losing its prefetch)
template <int G0, int G1>
__device__ __forceinline__ void chain_mfma_regs(v4f &acc, const float *w, const float *op, float4 (&cur)[2]) {
    float4 nxt[2];
#pragma unroll
    for (int g = G0; g < G1; ++g) {
        nxt[0] = cur[0]; nxt[1] = cur[1];
        const float hv[8] = {cur[0].x, cur[0].y, cur[0].z, cur[0].w, cur[1].x, cur[1].y, cur[1].z, cur[1].w};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w[8 * g + i], hv[i], acc, 0, 0, 0);
        // the next group's operands are requested BEHIND this group's terms (in front of them: 3.73 instead of 3.69 us per step at 32
        // utterances): three waves share the pipe, the read comes back while the other two run their groups
        if (g + 1 < NG_H) { nxt[0] = *(const float4 *)(op + 8 * (g + 1)); nxt[1] = *(const float4 *)(op + 8 * (g + 1) + 4); }
        __builtin_amdgcn_sched_barrier(0);
        cur[0] = nxt[0]; cur[1] = nxt[1];
    }
}
// the whole chain with its weights streamed from the word-interleaved LDS copy (word i of the lane's chain at wp[i * WS]), D groups
// ahead of their use -- weights AND operands: behind barrier A a ds_read_b128 comes back after ~300 cycles (twelve waves read), a
// group's 8 dependent matrix instructions take ~140, and with one group of prefetch the chain ran at the LDS latency (fc1, the first
// thing on the step's critical path: 1.9 us; profiles/r04_mfma_chains.txt).  wpre = words 0 .. 2 D - 1, requested before the barrier
// the chain waits behind.
template <int WS, int D>
__device__ __forceinline__ v4f chain_mfma_lds(const float4 *wp, const float4 *wpre, const float *op) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    float4 hb[D][2], wb[D][2];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        hb[d][0] = *(const float4 *)(op + 8 * d); hb[d][1] = *(const float4 *)(op + 8 * d + 4);
        wb[d][0] = wpre[2 * d]; wb[d][1] = wpre[2 * d + 1];
    }
#pragma unroll
    for (int g = 0; g < NG_H; ++g) {
        const int s = g % D;
        __builtin_amdgcn_sched_barrier(0);
        const float hv[8] = {hb[s][0].x, hb[s][0].y, hb[s][0].z, hb[s][0].w, hb[s][1].x, hb[s][1].y, hb[s][1].z, hb[s][1].w};
        const float wv[8] = {wb[s][0].x, wb[s][0].y, wb[s][0].z, wb[s][0].w, wb[s][1].x, wb[s][1].y, wb[s][1].z, wb[s][1].w};
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[i], hv[i], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (g + D < NG_H) {                                     // refill the buffer just used with group g + D
            hb[s][0] = *(const float4 *)(op + 8 * (g + D)); hb[s][1] = *(const float4 *)(op + 8 * (g + D) + 4);
            wb[s][0] = wp[(2 * (g + D)) * WS]; wb[s][1] = wp[(2 * (g + D) + 1) * WS];
        }
    }
    return acc;
}

}  // namespace
