// ar_xcd.hip -- the WaveRNN-style sample loop of Vocoder.generate (network_vocoder.py:78; RNN_MS spec: DESIGN.md 2.2) as
// EIGHT independent, weight-stationary, resident decoders: one per XCD.
//
// Utterances are independent, so they are dealt over the 8 XCDs (decode slot s lives on XCD s % 8) and each XCD runs its
// own copy of the recurrence on its 32 CUs, with a full copy of the weights held ON the CUs for the whole call:
//   * workgroup `rank` (0..31, one per CU, 768 threads) owns hidden units 28 rank .. 28 rank + 27 = 84 gate rows of W_hh
//     (80 of them in VGPRs, 112 weights per lane = one fp32 fma chain of the row; the last 4 in LDS), 8 rows of fc1 and
//     8 classes of fc2 (36.9 KB, LDS) and its slice of the sample-embedding table Gemb (86 KB, LDS);
//   * nothing is re-fetched per sample: after the prologue the only global traffic is the three all-to-all exchanges of a
//     sample step (h_t, a_t, the draw candidates) as 8-byte {tag, value} granules (the data is the flag), which stay
//     inside the XCD: published with workgroup-scope stores (the write-through L1 leaves them in the XCD's L2) and swept
//     with sc1 loads that bypass L1 -- 0.41 us per exchange against 1.2 us across XCDs
//     (profiles/r02_xcd_exchange_microbench.csv).  The placement is CHECKED, not assumed: every workgroup reads its
//     XCC_ID and takes a ticket; unless each XCD got exactly 32 workgroups the launch gives up (status 2) before it
//     touches any output, and the host falls back to the launch-per-step kernels.
// Arithmetic is bit-identical to those kernels: a row's dot product is the same 8 fp32 fma chains (K quarter x x/z|y/w
// accumulator: ar_shared.h), combined in the same order; cell update, fc epilogues and the Gumbel-max draw are the
// same expressions.  A chain lives in ONE lane; the four lanes of a quad hold the same chain of four different rows.
// With ONE or TWO slots per XCD (ar_xcd_kernel<1>, <2>) the quad shares a chain's operand values (each lane loads a quarter of
// h from LDS, v_fmac_f32_dpp quad_perm broadcasts them): a quarter of the LDS operand traffic, one vector instruction per term,
// row and slot.  With FOUR slots per XCD (ar_xcd_kernel<4>, 17..32 utterances) the same chains run on the matrix pipe:
// v_mfma_f32_4x4x1_16B_f32 advances a quad's four rows for the four slots in one instruction (ar_chain.h) -- the vector form was
// vector-ALU bound between the barriers at four slots (3.5 of 5.1 us; profiles/r04_ablation.txt, r04_mfma_chains.txt).
//
// Roles of the 12 waves of a workgroup (the two service waves are waves 0 and 1, the OLDEST of their SIMDs: the instruction
// arbiter serves the oldest wave first -- ar_xcm.hip has the measurements -- and what they do is the critical path; as waves
// 11 / 10 the single-utterance step took 2.90 us, as waves 0 / 1 it takes 2.82):
//   waves 2..11  (chain waves 0..9) W_hh h_t chains of rows 0..79 for every slot of the XCD (8 rows per wave, weights pinned
//                in VGPRs) -> gsum
//   chain wave b < bx  also: sweep of a_t of slot b, fc2 + Gumbel-max candidate of the 8 owned classes -> publish (between its
//                W_hh chains: one slot per wave, in parallel)
//   wave 1       with ONE slot per XCD (up to 8 utterances): fc1 with its 8 rows pinned like a chain wave's (2.82 -> 2.54 us per step)
//   wave 0 / 1   what else is serial in a sample step, for the even / odd slots: cell update of the 28 owned units; publish
//                h_t; the slot's state for the next step in the shadow of the h_t exchange; fc1 (weights streamed from LDS)
//                -> publish a_t; W_hh rows 80..83 for the OTHER wave's slots (one chain pass: a half wave per slot); the
//                next step's Gumbel noise (Philox + two logs per class); x_t from the slot's 32 candidates, picked up
//                before barrier B
//   all waves    sweep h_t into LDS
// Four slots per XCD, what differs: every wave runs ONE dependent chain of 112 matrix instructions per step (waves 2..11 rows
// 0..79 with pinned weights, wave 0 fc1 for all four slots and wave 1 W_hh rows 80..83, both with weights streamed from LDS);
// fc2 + draw of slots 0..3 on waves 2, 3, 6, 7; waves 4 and 8 -- fc1's SIMD mates -- start their chains when a_t is out (two
// dependent-chain waves keep a SIMD's issue port ~80 % busy, whatever s_setprio says); wave 11 draws the next step's noise
// behind its chain from the slots' clocks the service waves post in LDS.
// Two workgroup barriers per sample.  Every wait is wall-clock bounded; a timeout sets status bit 0, every workgroup
// leaves, and the call's outputs are incomplete (vqcpc_vocoder_check reports it).
#include "ar_xcd.h"
#include "ar_shared.h"

// Timeline stamps of worker 5 of XCD 0 (100 MHz wall clock), steps 256..383, for tools/xcd_timeline.py: compiled in only
// with -DVQCPC_XD_STAMPS (a debug build under build/stamps/, never the shipped library).
#ifdef VQCPC_XD_STAMPS
__device__ unsigned long long g_xd_stamps[128 * 16];
#define XD_STAMP(wv, i) do { if (rank == 5 && xcc == 0 && wave == (wv) && lane == 0 && t >= 256 && t < 384) \
        g_xd_stamps[(t - 256) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int vqcpc_debug_xd_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xd_stamps), sizeof(g_xd_stamps)) == hipSuccess ? 0 : -1;
}
// per-worker stamps of XCD 0 (slot 0's events on every worker): [event][worker][step 256..383]
__device__ unsigned long long g_xd_workers[6 * 32 * 128];
#define XD_WSTAMP(ev) do { if (xcc == 0 && lane == 0 && t >= 256 && t < 384) \
        g_xd_workers[((ev) * 32 + rank) * 128 + (t - 256)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int vqcpc_debug_xd_workers(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xd_workers), sizeof(g_xd_workers)) == hipSuccess ? 0 : -1;
}
#else
#define XD_STAMP(wv, i) do { } while (0)
#define XD_WSTAMP(ev) do { } while (0)
#endif
// -DVQCPC_XD_BARS (alone: the stamps above perturb what these measure)
#ifdef VQCPC_XD_BARS
// every wave of worker 5 of XCD 0 at the two barriers of a step: [step 256..383][wave][arrives at A, leaves A, arrives at B, leaves B]
__device__ unsigned long long g_xd_bars[128 * 12 * 4];
// (the arrival time waits in a register and is stored behind the barrier: a store in front of it would be waited for by the
// barrier's release fence, and the barrier would look as long as a store takes)
#define XD_BARRIVE() do { xd_arrived = __builtin_amdgcn_s_memrealtime(); } while (0)
#define XD_BLEAVE(i) do { if (rank == 5 && xcc == 0 && lane == 0 && t >= 256 && t < 384) { \
        g_xd_bars[((t - 256) * 12 + wave) * 4 + (i)] = xd_arrived; \
        g_xd_bars[((t - 256) * 12 + wave) * 4 + (i) + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
extern "C" int vqcpc_debug_xd_bars(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xd_bars), sizeof(g_xd_bars)) == hipSuccess ? 0 : -1;
}
#else
#define XD_BARRIVE() do { } while (0)
#define XD_BLEAVE(i) do { } while (0)
#endif

#include "ar_chain.h"

// Ablation of the exchange WAITS, for timing only (results are garbage: stale granules are used as they are found): bit 0 the
// candidate poll, bit 1 the a_t poll, bit 2 the h_t sweep take whatever their first load returns.  What a step still costs then
// is what the workgroup's own instruction streams, LDS traffic, L2 round trips and barriers cost -- profiles/r04_ablation.txt.
// Never set in the shipped library (tools/build_variant.sh A7 "-DXD_ABLATE=7"; tools/ab_libs.py times the builds).
#ifndef XD_ABLATE
#define XD_ABLATE 0
#endif
// four slots per XCD: groups of 8 terms (of 14) a chain wave runs before it turns to fc2 of its slot
#ifndef XD_GSPLIT
#define XD_GSPLIT 10
#endif
// four slots per XCD: groups of 8 terms the service waves' chains request ahead (weights and operands both come from LDS)
#ifndef XD_DEPTH
#define XD_DEPTH 3
#endif
// four slots per XCD: the two chain waves on fc1's SIMD start their chains when a_t is out (0: at once -- A/B builds, profiles/r04_mfma_chains.txt)
#ifndef XD_HOLD
#define XD_HOLD 1
#endif

namespace {

// LDS carve, in floats (ints behind them)
template <int BXT> struct Lds {
    static constexpr int gemb = 0;                        // [NC][3][UPB]   slice of the sample-embedding table
    static constexpr int fc1w = gemb + NC * ROWS;         // [28 words][64 lane-chains (row r8, chain cid)][4]: word-interleaved (chain_lds2)
    static constexpr int fc2w = fc1w + FPB * HR;          // [8 words][64 lane-chains][4]
    static constexpr int whx = fc2w + FPB * HF;           // [28 words][32 lane-chains][4]  W_hh rows 80..83 (the chain waves hold rows 0..79)
    static constexpr int HS = BXT == 4 ? HS4 : HR;        // slot stride of h_t (four slots: the matrix-pipe chains' copy, ar_chain.h)
    static constexpr int hc = whx + 4 * HR;               // [BXT][HS]  h_t, chain by chain (dpp chains: phase order; matrix pipe: term order)
    static constexpr int ac = hc + BXT * HS;              // [BXT][8 chains][AS]  a_t (32 terms per chain, chain stride AS = 48: bank windows)
    static constexpr int gsum = ac + BXT * 8 * 48;            // [BXT][96]  W_hh h of the owned rows [gate][unit]
    static constexpr int noise = gsum + BXT * 96;         // [2][BXT][8] Gumbel noise of the step in flight / the next one
    static constexpr int mtab = noise + 2 * BXT * 8;      // [NC] mu-law decode table
    static constexpr int bq = mtab + NC;                  // [3][32] b_hh of the owned units
    static constexpr int seg = bq + 96;                   // int [BXT][8] {index, row, t0, len, utt, samples into / index of the conditioning frame, first Gcond row}
    static constexpr int sinfo = seg + BXT * 8;           // int [BXT][2] {lt, utt} of the NEXT step, posted by advance() (four slots: chain wave 9 draws the noise)
    static constexpr int ctl = sinfo + BXT * 2;           // int [8] {xcc, rank, ok, abort, the call's status tag}
    static constexpr int total = ctl + 8;
};

template <int BXT>
__global__ __launch_bounds__(THREADS) void ar_xcd_kernel(XdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = Lds<BXT>;
    float *gemb = smem + L::gemb, *fc1w = smem + L::fc1w, *fc2w = smem + L::fc2w, *whx = smem + L::whx, *hc = smem + L::hc,
          *ac = smem + L::ac, *gsum = smem + L::gsum;
    float *noise = smem + L::noise, *mtab = smem + L::mtab, *c_bq = smem + L::bq;
    int *seg_st = (int *)(smem + L::seg), *sinfo = (int *)(smem + L::sinfo), *s_ctl = (int *)(smem + L::ctl);

    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));

    // ---- placement: which XCD am I on, which of its 32 workers am I?
    if (tid == 0) {
        unsigned xid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xid));
        xid &= 7u;
        if (p.dbg_misplace && blockIdx.x == 0) xid = (xid + 1u) & 7u;      // tests: one workgroup reports the wrong XCD
        unsigned *ctl = (unsigned *)p.xg;
        const unsigned r = __hip_atomic_fetch_add(ctl + xid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(ctl + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        const u64 t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned spins = 0; __hip_atomic_load(ctl + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x; ++spins) {
            if ((spins & 63) == 63 && (__builtin_amdgcn_s_memrealtime() - t0 > (u64)p.timeout_ticks ||
                                       __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (ok)
            for (int x = 0; x < 8; ++x)
                if (__hip_atomic_load(ctl + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)NW) ok = 0;
        if (!ok) __hip_atomic_store(p.status, p.status_tag | 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        s_ctl[0] = (int)xid; s_ctl[1] = (int)r; s_ctl[2] = ok; s_ctl[3] = 0; s_ctl[4] = (int)p.status_tag; s_ctl[5] = 0;
    }
    __syncthreads();
    const int xcc = __builtin_amdgcn_readfirstlane(s_ctl[0]), rank = __builtin_amdgcn_readfirstlane(s_ctl[1]);
    if (__builtin_amdgcn_readfirstlane(s_ctl[2]) == 0) return;
    int bx = (p.n_slots - xcc + 7) / 8;                // slots of this XCD: xcc, xcc + 8, ...
    bx = bx < 0 ? 0 : (bx > BXT ? BXT : bx);
    const int n_steps = p.n_steps[xcc];
    if (bx == 0 || n_steps <= 0) return;
    const int agent = p.agent_stores;

    u64 *gh = p.xg + CTL_WORDS / 2 + (size_t)xcc * xg_region(BXT);
    u64 *ga = gh + xg_h(BXT), *gc = ga + xg_a(BXT);

    // ---- lane geometry of the chain waves
    const unsigned R = lane >> 4, kw = (lane >> 2) & 3u, j = lane & 3u, rq = R >> 1, c0 = R & 1u, cid = 2u * kw + c0;
    const unsigned r8 = 4u * rq + j;                                   // row of the wave's 8
    const bool sum_lane = (lane & 0x1Cu) == 0;                         // kw == 0, c0 == 0: holds the row sum after chain_combine

    // ---- resident LDS state
    for (unsigned e = tid; e < NC * ROWS; e += THREADS) {
        const unsigned cls = e / ROWS, rem = e - cls * ROWS, g = rem / UPB, ul = rem - g * UPB;
        gemb[e] = p.Gemb[(size_t)cls * 3 * HR + g * HR + UPB * rank + ul];
    }
    for (unsigned e = tid; e < FPB * HR; e += THREADS) {
        const unsigned rr = e / HR, rem = e - rr * HR, cc = rem / NT_H, n = rem - cc * NT_H;
        fc1w[((n >> 2) * 64 + xd_lane_of(rr, cc)) * 4 + (n & 3)] = p.w_fc1[(size_t)(FPB * rank + rr) * HR + chain_col(HR / 64, cc >> 1, cc & 1, n)];
    }
    for (unsigned e = tid; e < FPB * HF; e += THREADS) {
        const unsigned rr = e / HF, rem = e - rr * HF, cc = rem / NT_A, n = rem - cc * NT_A;
        fc2w[((n >> 2) * 64 + xd_lane_of(rr, cc)) * 4 + (n & 3)] = p.w_fc2[(size_t)(FPB * rank + rr) * HF + chain_col(HF / 64, cc >> 1, cc & 1, n)];
    }
    for (unsigned e = tid; e < 4 * HR; e += THREADS) {                 // W_hh rows 80..83 = gate 2 (n), units 24..27
        const unsigned rr = e / HR, rem = e - rr * HR, cc = rem / NT_H, n = rem - cc * NT_H;
        whx[((n >> 2) * 32 + (xd_lane_of(rr, cc) & 31u)) * 4 + (n & 3)] = p.w_hh[(size_t)(2 * HR + UPB * rank + 24 + rr) * HR + chain_col(HR / 64, cc >> 1, cc & 1, n)];
    }
    for (unsigned e = tid; e < NC; e += THREADS) mtab[e] = p.mulaw_tab[e];
    for (unsigned e = tid; e < 96; e += THREADS) { const unsigned g = e >> 5, u = e & 31; c_bq[e] = u < UPB ? p.b_hh[g * HR + UPB * rank + u] : 0.f; }
    for (unsigned e = tid; e < BXT * 96; e += THREADS) gsum[e] = 0.f;
    for (unsigned e = tid; e < BXT * L::HS; e += THREADS) hc[e] = 0.f;     // four slots: the matrix instruction multiplies all four columns, also those of slots this XCD does not run
    for (unsigned e = tid; e < BXT; e += THREADS) {
        const XdSeg sg = (int)e < bx ? p.segs[(size_t)(xcc + 8 * e) * p.max_seg] : XdSeg{-1, 0, 0, 0u};
        seg_st[e * 8 + 0] = 0; seg_st[e * 8 + 1] = sg.len > 0 ? sg.row : -1; seg_st[e * 8 + 2] = sg.t0; seg_st[e * 8 + 3] = sg.len;
        seg_st[e * 8 + 4] = (int)sg.utt; seg_st[e * 8 + 5] = 0; seg_st[e * 8 + 6] = 0;      // samples into / index of the conditioning frame
        seg_st[e * 8 + 7] = sg.len > 0 ? ((const int *)(p.segs + (size_t)8 * BXT * p.max_seg))[sg.row] : 0;      // the utterance's first Gcond row
    }
    __syncthreads();

    Waiter wt{p.status, p.timeout_ticks, 0};
#ifdef VQCPC_XD_BARS
    unsigned long long xd_arrived = 0;
#endif
    int *s_abort = s_ctl + 3;

    // h_t gather: every thread takes column tid of every slot, the first 128 threads also column 768 + tid
    const unsigned hk1 = tid, hk2 = THREADS + (tid & 127u);
    const unsigned hoff1 = ((((hk1 / UPB) * BXT) << 5) + hk1 % UPB) * 8u, hoff2 = ((((hk2 / UPB) * BXT) << 5) + hk2 % UPB) * 8u;
    const unsigned hdst1 = BXT == 4 ? chain_pos_plain<HR / 64>(hk1, NT_H) : chain_pos<HR / 64>(hk1, NT_H),
                   hdst2 = BXT == 4 ? chain_pos_plain<HR / 64>(hk2, NT_H) : chain_pos<HR / 64>(hk2, NT_H);
    const bool two = tid < HR - THREADS;
#define XD_SWEEP_H()                                                                                              \
    do {                                                                                                          \
        u64 v1[BXT], v2[BXT];                                                                                     \
        wt.start();                                                                                               \
        for (unsigned spins = 0;; ++spins) {                                                                      \
            bool ok = true;                                                                                       \
            if (wave < 2) {                                                                                       \
                gran_load_pair<BXT, 256>(v1, v2, gh, hoff1, hoff2);                                               \
                _Pragma("unroll") for (int b = 0; b < BXT; ++b) ok &= b >= bx || (unsigned)(v2[b] >> 32) == tag;  \
            } else gran_load<BXT, 256>(v1, gh, hoff1);                                                            \
            _Pragma("unroll") for (int b = 0; b < BXT; ++b) ok &= b >= bx || (unsigned)(v1[b] >> 32) == tag;      \
            if ((XD_ABLATE & 4) || __all(ok)) break;                                                         \
            if (wt.expired(spins, lane, s_abort + 1)) { *s_abort = 1; break; }                                                 \
            __builtin_amdgcn_s_sleep(1);                                                                          \
        }                                                                                                         \
        _Pragma("unroll") for (int b = 0; b < BXT; ++b) {                                                         \
            if (b < bx) { hc[b * L::HS + hdst1] = __uint_as_float((unsigned)v1[b]); if (two) hc[b * L::HS + hdst2] = __uint_as_float((unsigned)v2[b]); } \
        }                                                                                                         \
    } while (0)

    const float *opnd = hc + cid * NT_H + 4 * j;                       // dpp chains: the quad's lane j reads word j of a phase
    const float *opm = hc + j * L::HS + cid * NT_H;                     // matrix-pipe chains (four slots): lane j reads slot j's chain
    bool fc1_role = false;
    if constexpr (BXT == 1) fc1_role = wave == 1;
    if (fc1_role) {
        // =====================================================================================  one slot per XCD: wave 1 is fc1
        // With one slot the second service wave has no slot of its own; it holds the 8 fc1 rows like a chain wave holds W_hh rows
        // (112 weights per lane, pinned) and runs fc1 -- the first thing on the step's critical path -- in one chain pass of
        // 0.4 us instead of the 0.75 us of the LDS-streamed pass on wave 0, which takes W_hh rows 80..83 in exchange.
        if constexpr (BXT == 1) {
            float w1[NT_H];
            ps_load_weights<HR / 64>(p.w_fc1 + (size_t)(FPB * rank + r8) * HR, kw, c0, w1);
            const float b1 = p.b_fc1[FPB * rank + r8];
            ps_barrier();                                                  // state and noise of step 0 posted
            for (int t = 0; t < n_steps; ++t) {
                const unsigned tag = (unsigned)t + 1u;
                XD_SWEEP_H();
                XD_BARRIVE();
                ps_barrier();                                            // A: h_t in LDS
                XD_BLEAVE(0);
                if (*s_abort) break;
                float v = chain_combine(chain_regs<NT_H>(w1, opnd));
                v += b1;
                v = v > 0.f ? v : 0.f;
                if (sum_lane) xd_put(ga, (((unsigned)(rank * BXT) << 3) + r8) * 8u, ((u64)tag << 32) | __float_as_uint(v), agent);
                XD_STAMP(1, 10);
                XD_BARRIVE();
                ps_barrier();                                            // B
                XD_BLEAVE(2);
                if (*s_abort) break;
            }
        }
    } else if (wave >= 2) {
        const int cw = wave - 2;                                           // chain wave 0..9
        // =====================================================================================  chain waves: W_hh rows 0..79
        const unsigned row_local = 8 * cw + r8;                          // gate * UPB + unit
        float w[NT_H];
        {
            const unsigned gate = row_local / UPB, ul = row_local - gate * UPB;
            ps_load_weights<HR / 64>(p.w_hh + (size_t)(gate * HR + UPB * rank + ul) * HR, kw, c0, w);
        }
        // wave b < bx also runs fc2 + the draw of slot b (one slot per wave, in parallel), between its W_hh chains: a_t of the
        // slot arrives while the first chains run
        // which slot's fc2 + draw this wave runs.  Four slots: waves 2, 3, 6, 7 (SIMDs 2 and 3) take slots 0..3 -- the chain waves that share
        // a SIMD with a service wave stay pure chains (their matrix instructions starve the vector instructions of whoever shares the SIMD:
        // profiles/r04_mfma_chains.txt), and wave 11 draws the next step's noise once its chain is through
        const int fs = BXT == 4 ? ((cw & 2) == 0 && cw < 6 ? 2 * (cw >> 2) + (cw & 1) : BXT) : cw;
        const bool fc2_wave = fs < bx;
        const int fc2_after = bx < 3 ? bx : 3;                             // chains done before it looks for a_t
        const float b2 = p.b_fc2[FPB * rank + r8];
        const float4 *wp2 = (const float4 *)fc2w + lane;                   // word i of this lane's chain at wp2[64 i]
        const float *opnd2 = ac + (fs < BXT ? fs : 0) * (8 * 48) + cid * 48 + 4 * j;
        ps_barrier();                                                      // state and noise of step 0 posted
        for (int t = 0; t < n_steps; ++t) {
            const unsigned tag = (unsigned)t + 1u;
            XD_SWEEP_H();
            XD_BARRIVE();
            ps_barrier();                                                // A: h_t in LDS
            XD_BLEAVE(0);
            if (*s_abort) break;
            auto fc2_and_draw = [&]() {
                // ---- a_t of slot `wave` (256 granules, 4 per lane) -> fc2 -> Gumbel-max candidate of the 8 owned classes
                const unsigned aoff = (((((lane >> 3) * BXT) + (unsigned)fs) << 3) + (lane & 7u)) * 8u;
                const unsigned adst = (unsigned)chain_pos<HF / 64>((int)lane, 48);      // a_t[lane + 64 i]: chain 2 i + (lane & 1), i.e. 96 i further
                float4 wa = wp2[0], wb = wp2[64];                    // first weights and the noise: on their way during the sweep
                unsigned lno = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));      // the lane id, from the hardware
                asm volatile("" : "+v"(lno));      // opaque, and rebuilt: the hoisted address of this read -- then r8, then the lane id
                const unsigned r8o = ((lno >> 5) << 2) | (lno & 3u);      // itself -- was spilled to scratch and reloaded at every step
                const float nz = noise[(t & 1) * (BXT * 8) + fs * 8 + r8o];
                u64 va[4];
                wt.start();
                for (unsigned spins = 0;; ++spins) {
                    gran_load4b<512 * BXT>(va, ga, ga + 128 * BXT, aoff);
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < 4; ++i) ok &= (unsigned)(va[i] >> 32) == tag;
                    if ((XD_ABLATE & 2) || __all(ok)) break;
                    if (wt.expired(spins, lane, s_abort + 1)) { *s_abort = 1; break; }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) ac[fs * (8 * 48) + 96 * i + adst] = __uint_as_float((unsigned)va[i]);
                XD_STAMP(2, 8);
                const float4 a0 = *(const float4 *)opnd2, a1 = *(const float4 *)(opnd2 + 16);
                const float hv[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                float acc2 = 0.f;
#pragma unroll
                for (int J = 0; J < 4; ++J) {                            // 32 terms: 8 per quad lane, weights two 16-byte words at a time
                    float4 na = wa, nb2 = wb;
                    if (J < 3) { na = wp2[64 * (2 * J + 2)]; nb2 = wp2[64 * (2 * J + 3)]; }
                    const float w8[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
                    if (J == 0) fmac8<0>(acc2, hv, w8);
                    if (J == 1) fmac8<1>(acc2, hv, w8);
                    if (J == 2) fmac8<2>(acc2, hv, w8);
                    if (J == 3) fmac8<3>(acc2, hv, w8);
                    wa = na; wb = nb2;
                }
                float v2 = chain_combine(acc2);
                v2 += b2;
                const float sc = v2 + nz;                                // classes 0..3 of the 8 in lanes 0..3, 4..7 in lanes 32..35
                float best = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc), 0));
                int kb = 0;
#pragma unroll
                for (int k = 1; k < 8; ++k) {
                    const float sk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc), k < 4 ? k : 28 + k));
                    if (sk > best) { best = sk; kb = k; }
                }
                const bool drop = p.dbg_drop_step >= 0 && t == p.dbg_drop_step && rank == 3 && xcc == 0;
                if (lane == 0 && !drop)
                    xd_put(gc, ((unsigned)fs * NW + (unsigned)rank) * 8u, ((u64)((tag << 8) | (unsigned)(FPB * rank + kb)) << 32) | __float_as_uint(best), agent);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the chain wave has slack here; draining its store now measured 0.035 us/step (profiles/r04_ab_*, section 5)
                XD_STAMP(2, 9); if (fs == 0) XD_WSTAMP(3);
            };
            if constexpr (BXT == 4) {
                // ---- the four slots at once on the matrix pipe (ar_chain.h); the wave of slot cw looks for a_t after XD_GSPLIT of the
                // chain's 14 groups: fc1 is out about then
                v4f a4 = {0.f, 0.f, 0.f, 0.f};
                float4 cur[2];
                cur[0] = *(const float4 *)opm; cur[1] = *(const float4 *)(opm + 4);
                // waves 4 and 8 share fc1's SIMD (wave 0): they stand back until a_t is out.  Three dependent-chain waves keep a SIMD's issue port
                // busy; fc1 -- the head of the step's critical path -- ran at a third of the matrix pipe, and the 60 vector instructions between its
                // last term and the a_t stores took 0.4..0.7 us among the others' matrix instructions (profiles/r04_mfma_chains.txt)
#if XD_HOLD
                if ((wave & 3) == 0)
                    while (__hip_atomic_load(s_ctl + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != (int)tag) __builtin_amdgcn_s_sleep(2);
#endif
                chain_mfma_regs<0, XD_GSPLIT>(a4, w, opm, cur);
                if (fc2_wave) fc2_and_draw();
                chain_mfma_regs<XD_GSPLIT, NG_H>(a4, w, opm, cur);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = chain_combine(a4[i]);                // row 8 cw + 4 rq + i of slot j
                    if (sum_lane) gsum[j * 96 + 8 * cw + 4 * rq + i] = v;
                }
                if (wave == 11) {
                    // ---- the Gumbel noise of step t + 1's draw, all four slots (class lane & 7 of slot lane >> 3): it depends on nothing but the
                    // slots' clocks, which advance() posted in front of barrier A
                    unsigned ln = lane;
                    asm volatile("" : "+v"(ln));
                    const int b = (int)(ln >> 3);
                    if (ln < 32 && b < bx) {
                        const unsigned cls = FPB * rank + (ln & 7u);
                        const unsigned wd = philox_word((unsigned)sinfo[b * 2], (unsigned)sinfo[b * 2 + 1], cls >> 2, (unsigned)p.seed, (unsigned)(p.seed >> 32), (int)(cls & 3u));
                        noise[((t + 1) & 1) * (BXT * 8) + b * 8 + (ln & 7u)] = gumbel_from_word(wd);
                    }
                }
            } else {
                for (int b = 0; b < bx; ++b) {
                    const float acc = chain_regs<NT_H>(w, opnd + b * HR);
                    const float v = chain_combine(acc);
                    if (sum_lane) gsum[b * 96 + row_local] = v;
                    if (fc2_wave && b + 1 == fc2_after) fc2_and_draw();
                }
            }
            XD_STAMP(2, 6);
            XD_BARRIVE();
            ps_barrier();                                                // B: gsum of step t complete; hc free for h_{t+1}
            XD_BLEAVE(2);
            if (*s_abort) break;
        }
    } else {
        // =====================================================================================  service waves
        // Wave 0 (sv 0) owns the even slots of the XCD, wave 1 (sv 1) the odd ones, for what is serial in a sample step
        // and not a chain wave's: x_{t-1} from the slot's 32 candidates, the cell update of the 28 owned units, h_t
        // published, fc1 -> a_t published; and, behind those, W_hh rows 80..83 for the OTHER wave's slots and the next
        // step's Gumbel noise.  Half wave hw of the cell update takes slot sv + 2 hw.
        const int sv = wave;
        const int hw = (int)(lane >> 5);
        const int cb = sv + 2 * hw;                                     // slot of this half wave in the cell update
        const unsigned cu = lane & 31u;                                 // unit
        const int n_own = bx > sv ? (bx - sv + 1) / 2 : 0;              // slots sv, sv + 2 < bx
        const bool cell_on = cb < bx;
        const float b1 = p.b_fc1[FPB * rank + r8];
        float b1q[4] = {0.f, 0.f, 0.f, 0.f};                              // four slots: the lane's accumulators are rows 4 rq + 0..3
        if constexpr (BXT == 4) { for (int i = 0; i < 4; ++i) b1q[i] = p.b_fc1[FPB * rank + 4 * rq + i]; }
        const float4 *wp1 = (const float4 *)fc1w + lane;                   // word i of this lane's chain at wp1[64 i]
        const float4 *wpx = (const float4 *)whx + (lane & 31u);            // rows 80..83: both half waves read the same 32 lane-chains, word i at wpx[32 i]
        const float bq0 = c_bq[cu], bq1 = c_bq[32 + cu], bq2 = c_bq[64 + cu];
        // the slot's 32 candidates are contiguous ([slot][worker]: a half wave's poll reads two lines, not one line per worker -- 1.6 % of a
        // single-utterance step, 0.7 % at 32 utterances: profiles/r04_ab_lds_layout_and_encoder_pingpong.txt)
        const u64 *csrc = gc + ((unsigned)(cb < BXT ? cb : 0) * NW + (lane & 31u));
        // W_hh rows 80..83 for the other wave's slots 1 - sv and 3 - sv: the lower half wave takes the first, the upper half the second
        const int xb0 = BXT == 1 ? 0 : 1 - sv;                            // one slot per XCD: wave 0 takes them itself (wave 1 is fc1)
        const bool x_two = xb0 + 2 < bx;
        const int xslot = xb0 + ((hw && x_two) ? 2 : 0);
        const float *opndx = opnd + xslot * HR;

        // ---- slot state, one step ahead: what the cell update of step `tn` will need that does not depend on the data
        bool st_active = false, st_first = false, st_emit = false;
        int st_erow = 0, st_eidx = 0, st_lt = 0;
        unsigned st_utt = 0u;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, hprev = 0.f;
        auto advance = [&](int tn) {
            st_active = false; st_first = false; st_emit = false;
            if (!cell_on) return;
            int si = seg_st[cb * 8 + 0], row = seg_st[cb * 8 + 1], t0 = seg_st[cb * 8 + 2], len = seg_st[cb * 8 + 3];
            int fpos = seg_st[cb * 8 + 5], fidx = seg_st[cb * 8 + 6], gb = seg_st[cb * 8 + 7];
            unsigned utt = (unsigned)seg_st[cb * 8 + 4];
            int lt = tn - t0;
            if (row >= 0 && lt >= 1 && lt <= len) { st_emit = true; st_erow = row; st_eidx = lt - 1; }    // x_{tn-1} is sample lt - 1 of `row`
            if (row >= 0 && lt >= len) {                             // next utterance of this slot (uniform per half wave)
                si += 1;
                XdSeg sg = XdSeg{-1, 0, 0, 0u};
                if (si < p.max_seg) sg = p.segs[(size_t)(xcc + 8 * cb) * p.max_seg + si];
                row = sg.len > 0 ? sg.row : -1; t0 = sg.t0; len = sg.len; utt = sg.utt;
                lt = tn - t0;
                fpos = 0; fidx = 0;
                gb = row >= 0 ? ((const int *)(p.segs + (size_t)8 * BXT * p.max_seg))[row] : 0;
                if (cu == 0) { seg_st[cb * 8 + 0] = si; seg_st[cb * 8 + 1] = row; seg_st[cb * 8 + 2] = t0; seg_st[cb * 8 + 3] = len; seg_st[cb * 8 + 4] = (int)utt;
                               seg_st[cb * 8 + 5] = 0; seg_st[cb * 8 + 6] = 0; seg_st[cb * 8 + 7] = gb; }
            }
            st_active = row >= 0 && lt >= 0 && lt < len;
            st_first = lt == 0;
            st_lt = lt; st_utt = utt;
            if (BXT == 4 && cu == 0) { sinfo[cb * 2] = lt; sinfo[cb * 2 + 1] = (int)utt; }
            if (st_active) {
                if (fpos == p.upsample) { fpos = 0; fidx += 1; }
                if (fpos == 0 && cu < UPB) {                         // next conditioning frame (once per hop)
                    const int f = fidx < p.F ? fidx : p.F - 1;
                    const float *gcp = p.Gcond + ((size_t)gb + f) * 3 * HR + UPB * rank + cu;
                    g0 = gcp[0]; g1 = gcp[HR]; g2 = gcp[2 * HR];
                }
                fpos += 1;
                if (cu == 0) { seg_st[cb * 8 + 5] = fpos; seg_st[cb * 8 + 6] = fidx; }
            }
        };
        // the Gumbel noise of step `tn`'s draw (it does not depend on the data): class lane & 7 of slot sv + 2 ((lane >> 3) & 1)
        auto draw_noise = [&](int tn) {
            const int lt_b = __builtin_amdgcn_readlane(st_lt, 0), lt_b2 = __builtin_amdgcn_readlane(st_lt, 32);
            const unsigned ut_b = __builtin_amdgcn_readlane(st_utt, 0), ut_b2 = __builtin_amdgcn_readlane(st_utt, 32);
            unsigned ln = lane;
            asm volatile("" : "+v"(ln));           // an opaque copy: hipcc hoisted the store address out of the sample loop and spilled it
            if (ln < 16) {
                const int which = (int)(ln >> 3), b = sv + 2 * which;
                if (b < bx) {
                    const unsigned cls = FPB * rank + (ln & 7u);
                    const unsigned wd = philox_word((unsigned)(which ? lt_b2 : lt_b), which ? ut_b2 : ut_b, cls >> 2,
                                                    (unsigned)p.seed, (unsigned)(p.seed >> 32), (int)(cls & 3u));
                    noise[(tn & 1) * (BXT * 8) + b * 8 + (ln & 7u)] = gumbel_from_word(wd);
                }
            }
        };
        advance(0);
        draw_noise(0);
        ps_barrier();                                                    // state and noise of step 0 posted

        int x = NC / 2;
        for (int t = 0; t < n_steps; ++t) {
            const unsigned tag = (unsigned)t + 1u;
            __builtin_amdgcn_s_setprio(3);
            XD_STAMP(0, 0);
            // ---- cell update: gsum of step t-1 is complete (barrier B), x_{t-1} was picked up before it
            float s0 = 0.f, s1 = 0.f, sn = 0.f, hold = 0.f;
            if (st_active && !st_first) {
                s0 = gsum[cb * 96 + cu]; s1 = gsum[cb * 96 + UPB + cu]; sn = gsum[cb * 96 + 2 * UPB + cu];
                hold = hprev;
            }
            s0 += bq0; s1 += bq1; sn += bq2;
            if (cell_on) {
                float hn = 0.f;
                if (st_active && cu < UPB) {
                    const int xe = st_first ? NC / 2 : x;
                    const float e0 = gemb[(xe * 3 + 0) * UPB + cu], e1 = gemb[(xe * 3 + 1) * UPB + cu], e2 = gemb[(xe * 3 + 2) * UPB + cu];
                    const float r = sigmoidf_((e0 + g0) + s0);
                    const float z = sigmoidf_((e1 + g1) + s1);
                    const float nn = tanhf((e2 + g2) + r * sn);
                    hn = (1.0f - z) * nn + z * hold;
                    hprev = hn;
                }
                if (cu < UPB) xd_put(gh, (((unsigned)(rank * BXT + cb) << 5) + cu) * 8u, ((u64)tag << 32) | __float_as_uint(hn), agent);
            }
            XD_STAMP(0, 2); if (wave == 0) XD_WSTAMP(0);
            // ---- in the shadow of the h_t exchange: the sample x_{t-1} goes out (network_vocoder.py:78 output), the slot's state
            // and the noise for step t + 1, the first phase of the fc1 weights
            if (st_emit && cu == 0 && rank == (cb & 31)) {
                if (p.wav) p.wav[(size_t)st_erow * p.Lout + st_eidx] = mtab[x];
                if (p.mulaw) p.mulaw[(size_t)st_erow * p.Lout + st_eidx] = x;
            }
            advance(t + 1);
            float4 w1p[BXT == 4 ? 2 * XD_DEPTH : 8];                      // the first weights of what follows barrier A, requested in front of it
#pragma unroll
            for (int i = 0; i < (BXT == 4 ? 2 * XD_DEPTH : 8); ++i) w1p[i] = (BXT == 4 && sv == 1) ? wpx[32 * i] : wp1[64 * i];      // four slots: wave 1 runs W_hh rows 80..83
            XD_SWEEP_H();
            XD_STAMP(0, 3);
            XD_BARRIVE();
            ps_barrier();                                                // A: h_t in LDS
            XD_BLEAVE(0);
            if (*s_abort) break;
            XD_STAMP(0, 4); if (wave == 0) XD_WSTAMP(1);
            // ---- fc1 of the own slots (both together: every weight is used for both and then dropped); one slot per XCD: wave 1
            if constexpr (BXT == 4) {
                // ---- four slots: the matrix pipe takes them together (ar_chain.h), so the two waves split the ROWS instead of the slots:
                // wave 0 fc1 (the step's critical path: a_t goes out first), wave 1 W_hh rows 80..83
                if (sv == 0) {
                    const v4f a4 = chain_mfma_lds<64, XD_DEPTH>(wp1, w1p, opm);
                    XD_STAMP(0, 14);
#ifdef VQCPC_XD_STAMPS
                    { const int done = __builtin_amdgcn_readfirstlane(__float_as_int(a4[0])); asm volatile("" :: "s"(done)); }      // the last term's RESULT is there
                    XD_STAMP(0, 15);
#endif
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = chain_combine(a4[i]);                  // fc1 row 4 rq + i of slot j
                        v += b1q[i];
                        v = v > 0.f ? v : 0.f;
                        if (sum_lane && (int)j < bx) xd_put(ga, (((unsigned)(rank * BXT) + j) << 3) * 8u + (4u * rq + (unsigned)i) * 8u, ((u64)tag << 32) | __float_as_uint(v), agent);
                    }
#if XD_HOLD
                    if (lane == 0) __hip_atomic_store(s_ctl + 5, (int)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // a_t is out: waves 4 and 8 may start
#endif
                } else {
                    const v4f a4 = chain_mfma_lds<32, XD_DEPTH>(wpx, w1p, opm);     // both half waves run the same 32 lane-chains; the lower half's results count
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = chain_combine(a4[i]);
                        if (sum_lane && lane < 32) gsum[j * 96 + 80 + i] = v;
                    }
                }
            } else if (BXT > 1 && n_own > 0) {
                float accA, accB;
                chain_lds2<NT_H, 64>(wp1, w1p, opnd + sv * HR, opnd + (sv + 2 < BXT ? sv + 2 : sv) * HR, n_own > 1, accA, accB);
                float v = chain_combine(accA);
                v += b1;
                v = v > 0.f ? v : 0.f;
                if (sum_lane) xd_put(ga, (((unsigned)(rank * BXT + sv) << 3) + r8) * 8u, ((u64)tag << 32) | __float_as_uint(v), agent);
                if (n_own > 1) {
                    float v2 = chain_combine(accB);
                    v2 += b1;
                    v2 = v2 > 0.f ? v2 : 0.f;
                    if (sum_lane) xd_put(ga, (((unsigned)(rank * BXT + sv + 2) << 3) + r8) * 8u, ((u64)tag << 32) | __float_as_uint(v2), agent);
                }
            }
            XD_STAMP(0, 5); if (wave == 0) XD_WSTAMP(2);
            XD_STAMP(1, 10);
            __builtin_amdgcn_s_setprio(1);
            // ---- W_hh rows 80..83 of the OTHER wave's slots, behind the a_t exchange and the chain waves' fc2: one chain pass,
            // the lower half wave on slot 1 - sv, the upper half on slot 3 - sv
            if (BXT != 4 && xb0 < bx) {
                float4 wx0[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) wx0[i] = wpx[32 * i];
                float acc, unused;
                chain_lds2<NT_H, 32>(wpx, wx0, opndx, opndx, false, acc, unused);
                const float v = chain_combine(acc);
                if (sum_lane && (lane < 32 || x_two)) gsum[xslot * 96 + 80 + (r8 & 3u)] = v;
            }
            XD_STAMP(0, 7);
            XD_STAMP(1, 11);
            if (BXT != 4) draw_noise(t + 1);                             // idle time: the candidates are still on their way (four slots: wave 11 draws)
            XD_STAMP(0, 13);
            // ---- x_t: the slot's 32 candidates (tag t + 1, from the chain waves' fc2), picked up BEFORE barrier B
            __builtin_amdgcn_s_setprio(3);
            if (n_own > 0) {
                u64 g = 0;
                wt.start();
                for (unsigned spins = 0;; ++spins) {
                    g = ps_load(csrc);
                    if ((XD_ABLATE & 1) || __all(!cell_on || (unsigned)(g >> 40) == tag)) break;
                    if (wt.expired(spins, lane, s_abort + 1)) { *s_abort = 1; break; }
                }
                // first argmax per half of 32 lanes (classes ascend with the rank): order-preserving integer image of the score
                unsigned u = (unsigned)g;
                u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                unsigned m = u;
                m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, false));
                m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, false));
                m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xF, 0xF, false));
                m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x140, 0xF, 0xF, false));
                const unsigned m0 = __builtin_amdgcn_readlane(m, 0), m1 = __builtin_amdgcn_readlane(m, 16),
                               m2 = __builtin_amdgcn_readlane(m, 32), m3 = __builtin_amdgcn_readlane(m, 48);
                const unsigned b01 = max(m0, m1), b23 = max(m2, m3);
                const unsigned long long hit = __ballot(u == (lane < 32 ? b01 : b23));
                const int f0 = __ffs((int)(unsigned)hit) - 1, f1 = __ffs((int)(unsigned)(hit >> 32)) - 1;
                const int cls = (int)((g >> 32) & 255u);
                const int x0 = __builtin_amdgcn_readlane(cls, f0 < 0 ? 0 : f0), x1 = __builtin_amdgcn_readlane(cls, 32 + (f1 < 0 ? 0 : f1));
                x = lane < 32 ? x0 : x1;
            }
            XD_STAMP(0, 1); if (wave == 0) XD_WSTAMP(4);
            __builtin_amdgcn_s_setprio(0);
            XD_BARRIVE();
            ps_barrier();                                                // B: gsum of step t complete; hc free for h_{t+1}
            XD_BLEAVE(2);
            if (*s_abort) break;
            XD_STAMP(0, 12); if (wave == 0) XD_WSTAMP(5);
        }
        // ---- the last step's x has nowhere to go: every utterance ended at least one step before n_steps
    }
#undef XD_SWEEP_H
}

template <int BXT>
constexpr size_t lds_bytes() { return sizeof(float) * (size_t)Lds<BXT>::total; }

template <int BXT>
int launch_t(const XdParams &p, hipStream_t s) {
    constexpr size_t lds = lds_bytes<BXT>();
    static_assert(lds <= 160 * 1024, "LDS budget");
    // per launch, not once per process: the attribute belongs to the current device, and a process may hold handles on several
    HIP_TRY(hipFuncSetAttribute((const void *)ar_xcd_kernel<BXT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((ar_xcd_kernel<BXT>), dim3(8 * NW), dim3(THREADS), lds, s, p);
    HIP_TRY(hipGetLastError());
    return VQCPC_OK;
}

}  // namespace

size_t xd_exchange_bytes(int bxt) { return (size_t)CTL_WORDS * 4 + (size_t)8 * xg_region(bxt) * sizeof(u64); }
bool xd_supported(int Hr, int Hf, int n_cls) { return Hr == HR && Hf == HF && n_cls == NC; }
int xd_pick_bxt(int n) { return n <= 1 ? 1 : n <= 2 ? 2 : n <= XD_MAX_BX ? 4 : 0; }

int xd_launch(const XdParams &p, hipStream_t s) {
    VQ_REQUIRE(p.bxt == 1 || p.bxt == 2 || p.bxt == 4, "xd_launch: bxt %d", p.bxt);
    VQ_REQUIRE(p.n_slots >= 1 && p.n_slots <= 8 * p.bxt, "xd_launch: %d slots do not fit 8 x %d", p.n_slots, p.bxt);
    HIP_TRY(hipMemsetAsync(p.xg, 0, xd_exchange_bytes(p.bxt), s));
    switch (p.bxt) {
        case 1: return launch_t<1>(p, s);
        case 2: return launch_t<2>(p, s);
        case 4: return launch_t<4>(p, s);
        default: return launch_t<4>(p, s);
    }
}
