// ar_xcp.hip -- the per-XCD resident decoders of ar_xcd.hip with the decode slots of an XCD PIPELINED through the workgroup
// instead of marching in lockstep (network_vocoder.py:78's sample loop; RNN_MS spec: DESIGN.md 2.2).
//
// Why.  One sample step of one slot is a latency chain of three all-to-all exchanges inside the XCD:
//     cell update -> h_t -> [W_hh; W_fc1] h_t -> a_t -> fc2 + draw -> candidates -> x_t -> cell update ...
// about 2.6 us end to end, of which a CU's vector ALUs work 0.3 us.  ar_xcd.hip runs the (up to four) slots of an XCD through
// that chain together, between two workgroup barriers per step, with two "service" waves doing everything that is not a
// pinned W_hh row -- fc1 for two slots each (weights streamed from LDS), the four left-over W_hh rows, the cell updates, the
// Gumbel noise, the candidate sweeps: 3.4 us of instructions per step on those two waves, 4.87 us per step at 32 utterances.
// Here
//   * EVERY wave holds 8 rows for the whole call (112 weights per lane): waves 0..9 W_hh rows 0..79 of the workgroup's 28
//     hidden units, wave 10 the 8 fc1 rows, wave 11 W_hh rows 80..83 -- [W_hh; W_fc1] h_t is ONE chain pass of all twelve
//     waves (92 of 96 row slots used); nothing is streamed from LDS any more;
//   * what is left of a step's serial work is dealt out, one duty per wave: waves 0..3 fc2 + noise + draw of slot (wave),
//     waves 4..7 the cell update + bookkeeping of slot (wave - 4), waves 8..11 the sweep of h_t into LDS;
//   * there is NO workgroup barrier in the sample loop.  The slots go round robin (for t: for slot: ...) in every wave's
//     program, and a wave waits only for what its next piece of work needs: a counter in LDS ("h_t of this slot is in LDS":
//     the four sweeping waves add to it; "the row sums of this slot are in LDS": all twelve add to it) or a granule tag in
//     the XCD's L2.  So while slot 0's a_t is on its way to the fc2 waves, the chain waves multiply slot 1's h_t, slot 2's
//     candidates are being merged and slot 3's h_t is being published: the slots settle a quarter of a step apart and the
//     step time tends to max(latency chain of one slot, slots x (sweep + chain pass)).
// Arithmetic, sampling stream, granule format, placement check, bounded waits and status word are ar_xcd.hip's (ar_chain.h,
// ar_shared.h): the samples are the same bits (tests/test_gpu_xcp.py checks against the launch path and the C oracle).
//
// Deadlock freedom: events are ordered (t, slot, stage) with stage = cell < publish h < sweep < chain pass < publish a < fc2 <
// publish candidate; every wave's program visits them in that order and every wait is for an event that is earlier in it
// (x_{t-1} of the slot, row sums of (t-1, slot), h_t / a_t of (t, slot)), so the earliest unfinished event can always run.
// Single-buffered granules are safe as before: nobody can produce step t+1's value of a word before every consumer has
// taken step t's (the chain of dependencies goes through every worker's sweep).  h_t of a slot in LDS is overwritten by the
// sweep of step t+1 only after all twelve waves have added to that slot's row-sum counter for step t.
#include "ar_chain.h"

#ifdef VQCPC_XD_STAMPS
__device__ unsigned long long g_xp_stamps[128 * 32];
// stamp k of slot s at step t (whichever wave writes it): 1 candidates in (cell update starts), 2 h_t published, 3 wave 10's part of
// h_t swept, 4 h_t complete in LDS (wave 10 starts its chain pass), 5 a_t published, 6 a_t gathered by the fc2 wave, 7 candidate published
#define XP_STAMP(s, k) do { if (rank == 5 && xcc == 0 && lane == 0 && t >= 256 && t < 384) \
        g_xp_stamps[(t - 256) * 32 + (s) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
__device__ unsigned long long g_xp_workers[6 * 32 * 128];      // [event: h / a / candidate published][worker][step 256..383], slot 0 of XCD 0
#define XP_WSTAMP(ev, s) do { if ((s) == 0 && xcc == 0 && lane == 0 && t >= 256 && t < 384) \
        g_xp_workers[((ev) * 32 + rank) * 128 + (t - 256)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int vqcpc_debug_xp_workers(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xp_workers), sizeof(g_xp_workers)) == hipSuccess ? 0 : -1;
}
__device__ unsigned long long g_xp_waves[2 * 12 * 32 * 128];      // [event][wave][worker][step]: 0 sweep of slot 1 done (sweepers) / 1 chain pass over slot 1 starts
#define XP_VSTAMP(ev, s) do { if ((s) == 1 && xcc == 0 && lane == 0 && t >= 256 && t < 384) \
        g_xp_waves[(((ev) * 12 + wave) * 32 + rank) * 128 + (t - 256)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int vqcpc_debug_xp_waves(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xp_waves), sizeof(g_xp_waves)) == hipSuccess ? 0 : -1;
}
__device__ unsigned g_xp_hwid[32];
extern "C" int vqcpc_debug_xp_hwid(unsigned *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xp_hwid), sizeof(g_xp_hwid)) == hipSuccess ? 0 : -1; }
#define XP_HWID() do { if (xcc == 0 && tid == 0) { unsigned h_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h_)); g_xp_hwid[rank] = h_; } } while (0)
__device__ unsigned g_xp_polls[4];
#define XP_COUNT(s) do { if (rank == 5 && xcc == 0 && lane == 0 && wave == XP_FC1 && t >= 256 && t < 384) g_xp_polls[s] += 1; } while (0)
extern "C" int vqcpc_debug_xp_polls(unsigned *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xp_polls), sizeof(g_xp_polls)) == hipSuccess ? 0 : -1;
}
extern "C" int vqcpc_debug_xp_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xp_stamps), sizeof(g_xp_stamps)) == hipSuccess ? 0 : -1;
}
#else
#define XP_STAMP(wv, i) do { } while (0)
#define XP_COUNT(s) do { } while (0)
#define XP_HWID() do { } while (0)
#define XP_VSTAMP(ev, s) do { } while (0)
#define XP_WSTAMP(ev, s) do { } while (0)
#endif

namespace {

constexpr int XP_FC1 = 10;             // the wave that holds the 8 fc1 rows
constexpr int XP_WHX = 11;             // the wave that holds W_hh rows 80..83
constexpr int XP_SWEEPERS = 4;         // waves 8..11 sweep h_t
constexpr int XP_WAVES = THREADS / 64;
#ifndef XP_OWN_LINES
#define XP_OWN_LINES 0            // 1: every worker publishes into 128-byte lines of its own; 0: what one wave sweeps is contiguous (workers share lines)
#endif
constexpr unsigned XP_POLL_SPINS = 3;  // looks at an LDS counter (~60 ns each) between issuing a poll of L2 and reading its answer

// exchange area of one XCD, in granules: h [slot][worker][32] (28 used: two whole lines per worker), a [slot][worker][8],
// candidates [slot][worker] -- what ONE wave sweeps is contiguous (a_t of a slot 2 KB, its candidates 256 B): a sweep costs the
// CU's memory pipeline a handful of lines, not one line per worker (workers share lines; all stores stay in this XCD's L2)
__host__ __device__ constexpr int xp_h(int bxt) { return bxt * NW * 32; }
__host__ __device__ constexpr int xp_a(int bxt) { return bxt * NW * FPB; }
__host__ __device__ constexpr int xp_region(int bxt) { return xp_h(bxt) + xp_a(bxt) + 16 * NW; }

template <int BXT> struct LdsP {
    static constexpr int gemb = 0;                        // [NC][3][UPB]   slice of the sample-embedding table
    static constexpr int fc2w = gemb + NC * ROWS;         // [FPB][8 chains][32]
    static constexpr int hc = fc2w + FPB * HF;            // [BXT][HR]  h_t, chain order
    static constexpr int ac = hc + BXT * HR;              // [BXT][HF]  a_t, chain order (private to the slot's fc2 wave)
    static constexpr int gsum = ac + BXT * HF;            // [BXT][96]  W_hh h of the owned rows [gate][unit]
    static constexpr int mtab = gsum + BXT * 96;          // [NC] mu-law decode table
    static constexpr int bq = mtab + NC;                  // [3][32] b_hh of the owned units
    static constexpr int seg = bq + 96;                   // int [BXT][8] {index, row, t0, len, utt, samples into / index of the conditioning frame}
    static constexpr int sinfo = seg + BXT * 8;           // int [BXT][4] {lt, utt} of the step whose h_t is on its way: for the slot's noise
    static constexpr int cnt = sinfo + BXT * 4;           // int [2][16]: h_t of slot s swept (4 per step), row sums of slot s stored (12 per step)
    static constexpr int ctl = cnt + 32;                  // int [4] {xcc, rank, ok, abort}
    static constexpr int par = ctl + 4;                   // rarely used kernel arguments (XpPar): read from LDS where they are needed instead of
    static constexpr int total = par + 24;                // sitting in SGPRs for the whole call (the sample loop has none to spare: 11 were spilled)
};

struct XpPar {                     // 96 bytes
    float *wav; int64_t *mulaw; const XdSeg *segs; const float *Gcond;
    unsigned long long seed;
    int Lout, max_seg, F, upsample, dbg_drop_step, pad;
};
static_assert(sizeof(XpPar) <= 24 * 4, "XpPar");

__device__ __forceinline__ int lds_peek(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// one wave's LDS instructions are performed in order: a counter bumped after the data is seen after the data (ar_xcm.hip); the
// compiler is kept from moving LDS accesses across it
__device__ __forceinline__ void lds_bump(int *p, unsigned lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned xp_ordered(unsigned u) { return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

template <int BXT>
__global__ __launch_bounds__(THREADS) void ar_xcp_kernel(XdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = LdsP<BXT>;
    float *gemb = smem + L::gemb, *fc2w = smem + L::fc2w, *hc = smem + L::hc, *ac = smem + L::ac, *gsum = smem + L::gsum;
    float *mtab = smem + L::mtab, *c_bq = smem + L::bq;
    int *seg_st = (int *)(smem + L::seg), *sinfo = (int *)(smem + L::sinfo), *cnt = (int *)(smem + L::cnt), *s_ctl = (int *)(smem + L::ctl);
    volatile XpPar *par = (volatile XpPar *)(smem + L::par);

    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const int pwave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int wave = ((p.xp_cell_lag >= 0 && (p.xp_cell_lag & 4096)) ? (pwave + 8) % 12 : pwave);      // role of this wave (tuning: sweepers as the oldest waves)

    // ---- placement: which XCD am I on, which of its 32 workers am I?  (as ar_xcd.hip)
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
        if (!ok) __hip_atomic_store(p.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        s_ctl[0] = (int)xid; s_ctl[1] = (int)(r & 31u); s_ctl[2] = ok; s_ctl[3] = 0;
        par->wav = p.wav; par->mulaw = p.mulaw; par->segs = p.segs; par->Gcond = p.Gcond; par->seed = p.seed;
        par->Lout = p.Lout; par->max_seg = p.max_seg; par->F = p.F; par->upsample = p.upsample; par->dbg_drop_step = p.dbg_drop_step;
    }
    __syncthreads();
    const int xcc = __builtin_amdgcn_readfirstlane(s_ctl[0]), rank = __builtin_amdgcn_readfirstlane(s_ctl[1]);
    if (__builtin_amdgcn_readfirstlane(s_ctl[2]) == 0) return;
    int bx = (p.n_slots - xcc + 7) / 8;                // slots of this XCD: xcc, xcc + 8, ...
    bx = bx < 0 ? 0 : (bx > BXT ? BXT : bx);
    const int n_steps = p.n_steps[xcc];
    if (bx == 0 || n_steps <= 0) return;
    const int agent = p.agent_stores;
    XP_HWID();

    u64 *gh = p.xg + CTL_WORDS / 2 + (size_t)xcc * xp_region(BXT);
    u64 *ga = gh + xp_h(BXT), *gc = ga + xp_a(BXT);

    // ---- lane geometry of a chain pass (ar_chain.h)
    const unsigned R = lane >> 4, kw = (lane >> 2) & 3u, j = lane & 3u, rq = R >> 1, c0 = R & 1u, cid = 2u * kw + c0;
    const unsigned r8 = 4u * rq + j;                                   // row of the wave's 8
    const bool sum_lane = (lane & 0x1Cu) == 0;                         // kw == 0, c0 == 0: holds the row sum after chain_combine

    // ---- resident LDS state
    for (unsigned e = tid; e < NC * ROWS; e += THREADS) {
        const unsigned cls = e / ROWS, rem = e - cls * ROWS, g = rem / UPB, ul = rem - g * UPB;
        gemb[e] = p.Gemb[(size_t)cls * 3 * HR + g * HR + UPB * rank + ul];
    }
    for (unsigned e = tid; e < FPB * HF; e += THREADS) {
        const unsigned rr = e / HF, rem = e - rr * HF, cc = rem / NT_A, n = rem - cc * NT_A;
        fc2w[e] = p.w_fc2[(size_t)(FPB * rank + rr) * HF + chain_col(HF / 64, cc >> 1, cc & 1, n)];
    }
    for (unsigned e = tid; e < NC; e += THREADS) mtab[e] = p.mulaw_tab[e];
    for (unsigned e = tid; e < 96; e += THREADS) { const unsigned g = e >> 5, u = e & 31; c_bq[e] = u < UPB ? p.b_hh[g * HR + UPB * rank + u] : 0.f; }
    for (unsigned e = tid; e < BXT * 96; e += THREADS) gsum[e] = 0.f;
    for (unsigned e = tid; e < 32; e += THREADS) cnt[e] = 0;
    for (unsigned e = tid; e < BXT; e += THREADS) {
        const XdSeg sg = (int)e < bx ? p.segs[(size_t)(xcc + 8 * e) * p.max_seg] : XdSeg{-1, 0, 0, 0u};
        seg_st[e * 8 + 0] = 0; seg_st[e * 8 + 1] = sg.len > 0 ? sg.row : -1; seg_st[e * 8 + 2] = sg.t0; seg_st[e * 8 + 3] = sg.len;
        seg_st[e * 8 + 4] = (int)sg.utt; seg_st[e * 8 + 5] = 0; seg_st[e * 8 + 6] = 0;      // samples into / index of the conditioning frame
        sinfo[e * 4 + 0] = 0; sinfo[e * 4 + 1] = 0;
    }

    // ---- this wave's 8 rows, pinned for the whole call
    float w[NT_H];
    {
        const float *Wrow;
        if (wave < 10) {
            const unsigned row_local = 8 * wave + r8, gate = row_local / UPB, ul = row_local - gate * UPB;      // gate * UPB + unit
            Wrow = p.w_hh + (size_t)(gate * HR + UPB * rank + ul) * HR;
        } else if (wave == XP_FC1) {
            Wrow = p.w_fc1 + (size_t)(FPB * rank + r8) * HR;
        } else {
            Wrow = p.w_hh + (size_t)(2 * HR + UPB * rank + 24 + (r8 & 3u)) * HR;      // gate 2 (n), units 24..27; rows 4..7 of this wave repeat them (unused)
        }
        ps_load_weights<HR / 64>(Wrow, kw, c0, w);
    }
    __syncthreads();

    const int tune = p.xp_cell_lag < 0 ? 0 : p.xp_cell_lag;
    auto setprio = [&](int v) {
        if (v == 0) __builtin_amdgcn_s_setprio(0);
        else if (v == 1) __builtin_amdgcn_s_setprio(1);
        else if (v == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
    };
    const int tSP = tune & 3, tWP = (tune >> 2) & 3, tFS = (tune >> 4) & 3, tDP = (tune >> 6) & 3, tSS = (tune >> 8) & 3, tCS = (tune >> 10) & 3;
    auto nap = [&](int v) {
        if (v == 0) __builtin_amdgcn_s_sleep(1);
        else if (v == 1) __builtin_amdgcn_s_sleep(3);
        else if (v == 2) __builtin_amdgcn_s_sleep(6);
        else __builtin_amdgcn_s_sleep(12);
    };
    Waiter wt{p.status, p.timeout_ticks, 0};
    int *s_abort = s_ctl + 3;
    int *hcnt = cnt, *gcnt = cnt + 16;
    bool dead = false;                                     // a wait of this wave gave up (its own deadline, or somebody else's abort)
    // one more look at the clock / the abort word of a wait that has spun `spins` times; true: give up
    auto give_up = [&](unsigned spins) {
        if (wt.expired(spins, (int)lane) || lds_peek(s_abort) != 0) { *s_abort = 1; dead = true; return true; }
        return false;
    };
    // wait until counter *c has reached `target` (counters only grow)
    auto wait_cnt = [&](const int *c, int target) {
        if (lds_peek(c) >= target) { asm volatile("" ::: "memory"); return; }
        wt.start();
        for (unsigned spins = 0; lds_peek(c) < target; ++spins) {
            if (give_up(spins)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
    };
    const float *opnd = hc + cid * NT_H + 8 * j;

    if (wave >= XP_WAVES - XP_SWEEPERS) {
        // =====================================================================================  waves 8..11: h_t -> LDS, then their rows
        // 256 lanes x 4 granules: lane (g = 0..7, u = 0..31) takes unit u (< 28) of workers 4 g .. 4 g + 3
        const unsigned Lq = (unsigned)(wave - (XP_WAVES - XP_SWEEPERS)) * 64u + lane, g = Lq >> 5, u = Lq & 31u;
        const bool sw_on = u < (unsigned)UPB;
        const unsigned uu = sw_on ? u : 0u;
        unsigned hdst[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) hdst[q] = (unsigned)chain_pos(HR / 64, UPB * (int)(4u * g + q) + (int)uu);
        const float b1 = p.b_fc1[FPB * rank + r8];
        const u64 *hsrc = gh + (4u * g) * 32u + uu;
        setprio(wave == XP_FC1 ? tWP : tSP);
        // A sweep is a round trip to L2 (0.4 us and more) even when the granules have long been there, and this wave's program is
        // sweep, chain pass, sweep, chain pass ...: so the loads of the NEXT slot's sweep are issued before the chain pass and
        // read behind it (one sweep + one pass in series per slot made the sweepers the bottleneck: 4 x 1.1 us per step).
        auto h_poll = [&](u64 (&v)[4], int s) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = ps_load(hsrc + (size_t)s * (NW * 32) + q * 32);
        };
        auto h_there = [&](const u64 (&v)[4], unsigned tag) {
            bool ok = true;
#pragma unroll
            for (int q = 0; q < 4; ++q) ok &= (unsigned)(v[q] >> 32) == tag;
            return (bool)__all(ok);
        };
        u64 v[4];
        h_poll(v, 0);
        for (int t = 0; t < n_steps && !dead; ++t) {
            const unsigned tag = (unsigned)t + 1u;
            for (int s = 0; s < bx; ++s) {
                if (wave == XP_FC1) XP_STAMP(s, 0);
                if (!h_there(v, tag)) {
                    wt.start();
                    for (unsigned spins = 0;; ++spins) {
                        nap(tSS);
                        h_poll(v, s);
                        if (h_there(v, tag)) break;
                        if (give_up(spins)) break;
                        XP_COUNT(s);
                    }
                }
                wait_cnt(gcnt + s, XP_WAVES * t);                      // every wave is through its chain pass over h_{t-1} of this slot
                if (sw_on) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) hc[s * HR + hdst[q]] = __uint_as_float((unsigned)v[q]);
                }
                lds_bump(hcnt + s, lane);
                XP_VSTAMP(0, s);
                if (wave == XP_FC1) XP_STAMP(s, 3);
                wait_cnt(hcnt + s, XP_SWEEPERS * (t + 1));
                XP_VSTAMP(1, s);
                if (wave == XP_FC1) { XP_STAMP(s, 4); __builtin_amdgcn_s_setprio(3); }
                h_poll(v, s + 1 < bx ? s + 1 : 0);                      // the next sweep: in flight during the chain pass
                asm volatile("" ::: "memory");
                float vv = chain_combine(chain_regs<NT_H>(w, opnd + s * HR));
                if (wave == XP_FC1) {
                    vv += b1;
                    vv = vv > 0.f ? vv : 0.f;
                    if (sum_lane) xd_put(ga, (((unsigned)(XP_OWN_LINES ? rank * BXT + s : s * NW + rank) << 3) + r8) * 8u, ((u64)tag << 32) | __float_as_uint(vv), agent);
                    setprio(tWP);
                } else if (wave == XP_WHX) {
                    if (sum_lane && r8 < 4u) gsum[s * 96 + 80 + r8] = vv;
                } else {
                    if (sum_lane) gsum[s * 96 + 8 * wave + r8] = vv;
                }
                lds_bump(gcnt + s, lane);
                if (wave == XP_FC1) { XP_STAMP(s, 5); XP_WSTAMP(1, s); }
                if (dead) break;
            }
            if (((tune >> 17) & 1) && (t & 15) == 15) __syncthreads();
        }
    } else if (wave < 4) {
        // =====================================================================================  waves 0..3: their rows; fc2 + noise + draw of slot (wave)
        // The duty of (t, f) -- a_t of the slot gathered, fc2, Gumbel-max candidate published -- is pending once this wave's own
        // chain pass over h_t of the slot is done, and is the next thing the wave does.
        const int f = wave;
        const bool duty_on = f < bx;
        const float b2 = p.b_fc2[FPB * rank + r8];
        const float4 *wp2 = (const float4 *)(fc2w + (r8 * 8 + cid) * NT_A);
        const unsigned cls = (unsigned)(FPB * rank) + r8;              // the class whose score ends up in this lane (sum lanes)
        const unsigned adst = (lane & 1u) * NT_A + 8 * (lane >> 4) + 4 * ((lane >> 1) & 1u) + ((lane >> 2) & 3u);
        // a_t of the slot: 256 granules, lane takes k = lane + 64 i = row (lane & 7) of worker (lane >> 3) + 8 i
        const u64 *asrc = XP_OWN_LINES ? ga + ((size_t)(lane >> 3) * BXT + f) * FPB + (lane & 7u) : ga + (size_t)f * (NW * FPB) + lane;
        float nz = 0.f;                                                // the slot's noise for the pending duty
        int duty_t = 0, own_done = 0;                                  // next duty step; own-slot chain passes done
        auto a_poll = [&](u64 (&va)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) va[i] = ps_load(asrc + (XP_OWN_LINES ? 64 * BXT : 64) * i);
        };
        auto a_there = [&](const u64 (&va)[4]) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < 4; ++i) ok &= (unsigned)(va[i] >> 32) == (unsigned)duty_t + 1u;
            return (bool)__all(ok);
        };
        auto duty = [&](const u64 (&va)[4]) {
            const int t = duty_t;                                       // (stamps)
            (void)t;
            const int s = f;
            const unsigned tag = (unsigned)duty_t + 1u;
            setprio(3 - tDP);
#pragma unroll
            for (int i = 0; i < 4; ++i) ac[s * HF + 64 * i + adst] = __uint_as_float((unsigned)va[i]);
            XP_STAMP(s, 6);
            const float *opnd2 = ac + s * HF + cid * NT_A + 8 * j;
            float4 wa = wp2[0], wb = wp2[1];
            const float4 a0 = ((const float4 *)opnd2)[0], a1 = ((const float4 *)opnd2)[1];
            const float hv[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            float acc2 = 0.f;
#pragma unroll
            for (int J = 0; J < 4; ++J) {                            // 32 terms: 8 per quad lane, weights two 16-byte words at a time
                float4 na = wa, nb2 = wb;
                if (J < 3) { na = wp2[2 * J + 2]; nb2 = wp2[2 * J + 3]; }
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
            const int dstep = par->dbg_drop_step;
            const bool drop = dstep >= 0 && duty_t == dstep && rank == 3 && xcc == 0;
            if (lane == 0 && !drop)
                xd_put(gc, (XP_OWN_LINES ? (unsigned)rank * 16 + (unsigned)s : (unsigned)s * NW + (unsigned)rank) * 8u, ((u64)((tag << 8) | (unsigned)(FPB * rank + kb)) << 32) | __float_as_uint(best), agent);
            XP_STAMP(s, 7); XP_WSTAMP(2, s);
            __builtin_amdgcn_s_setprio(0);
            duty_t += 1;
        };
        for (int t = 0; t < n_steps && !dead; ++t) {
            for (int s = 0; s < bx; ++s) {
                // ---- the pending duty first: it is on the step's critical path (a_t is on its way: ~0.4 us), this wave's chain passes
                // are not (their row sums are needed a cell update later)
                if (duty_on && duty_t < own_done) {
                    u64 va[4];
                    wt.start();
                    for (unsigned spins = 0;; ++spins) {
                        a_poll(va);
                        if (a_there(va)) { duty(va); break; }
                        if (give_up(spins)) break;
                        if (tFS) nap(tFS - 1);
                    }
                    if (dead) break;
                }
                wait_cnt(hcnt + s, XP_SWEEPERS * (t + 1));
                if (dead) break;
                XP_VSTAMP(1, s);
                const float vv = chain_combine(chain_regs<NT_H>(w, opnd + s * HR));
                if (sum_lane) gsum[s * 96 + 8 * wave + r8] = vv;
                lds_bump(gcnt + s, lane);
                if (s == f) {
                    // own pass done: the duty of (t, f) is pending; its Gumbel noise now (the cell wave posted the step's record
                    // before it published h_t)
                    unsigned so = (unsigned)s;
                    asm volatile("" : "+v"(so));
                    const unsigned long long seed = par->seed;
                    const int lt = sinfo[so * 4 + 0];
                    const unsigned utt = (unsigned)sinfo[so * 4 + 1];
                    nz = gumbel_from_word(philox_word((unsigned)lt, utt, cls >> 2, (unsigned)seed, (unsigned)(seed >> 32), (int)(cls & 3u)));
                    own_done += 1;
                }
            }
            if (((tune >> 17) & 1) && (t & 15) == 15) __syncthreads();
        }
        // the duty of the last step has no taker: x of step n_steps - 1 goes nowhere
    } else {
        // =====================================================================================  waves 4..7: their rows; cell update + books of slot (wave - 4)
        const int c = wave - 4;
        const unsigned cu = lane & 31u;                                 // unit (the upper half wave mirrors the lower one)
        const bool cell_on = c < bx;
        const float bq0 = c_bq[cu], bq1 = c_bq[32 + cu], bq2 = c_bq[64 + cu];
        const u64 *csrc = XP_OWN_LINES ? gc + (cu * 16 + (unsigned)c) : gc + ((unsigned)c * NW + cu);       // the slot's 32 candidates

        // ---- slot state, one step ahead: what the cell update of step `tn` will need that does not depend on the data
        bool st_active = false, st_first = false, st_emit = false;
        int st_erow = 0, st_eidx = 0, st_lt = 0;
        unsigned st_utt = 0u;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, hprev = 0.f;
        auto advance = [&](int tn) {
            st_active = false; st_first = false; st_emit = false;
            if (!cell_on) return;
            int si = seg_st[c * 8 + 0], row = seg_st[c * 8 + 1], t0 = seg_st[c * 8 + 2], len = seg_st[c * 8 + 3];
            int fpos = seg_st[c * 8 + 5], fidx = seg_st[c * 8 + 6];
            unsigned utt = (unsigned)seg_st[c * 8 + 4];
            int lt = tn - t0;
            if (row >= 0 && lt >= 1 && lt <= len) { st_emit = true; st_erow = row; st_eidx = lt - 1; }    // x_{tn-1} is sample lt - 1 of `row`
            if (row >= 0 && lt >= len) {                             // next utterance of this slot
                si += 1;
                XdSeg sg = XdSeg{-1, 0, 0, 0u};
                const int max_seg = par->max_seg;
                if (si < max_seg) sg = par->segs[(size_t)(xcc + 8 * c) * max_seg + si];
                row = sg.len > 0 ? sg.row : -1; t0 = sg.t0; len = sg.len; utt = sg.utt;
                lt = tn - t0;
                fpos = 0; fidx = 0;
                if (lane == 0) { seg_st[c * 8 + 0] = si; seg_st[c * 8 + 1] = row; seg_st[c * 8 + 2] = t0; seg_st[c * 8 + 3] = len; seg_st[c * 8 + 4] = (int)utt;
                                 seg_st[c * 8 + 5] = 0; seg_st[c * 8 + 6] = 0; }
            }
            st_active = row >= 0 && lt >= 0 && lt < len;
            st_first = lt == 0;
            st_lt = lt; st_utt = utt;
            if (st_active) {
                if (fpos == par->upsample) { fpos = 0; fidx += 1; }
                if (fpos == 0 && cu < UPB) {                         // next conditioning frame (once per hop)
                    const int F = par->F;
                    const int fr = fidx < F ? fidx : F - 1;
                    const float *gcp = par->Gcond + ((size_t)row * F + fr) * 3 * HR + UPB * rank + cu;
                    g0 = gcp[0]; g1 = gcp[HR]; g2 = gcp[2 * HR];
                }
                fpos += 1;
                if (lane == 0) { seg_st[c * 8 + 5] = fpos; seg_st[c * 8 + 6] = fidx; }
            }
        };
        advance(0);

        int x = NC / 2;
        int cell_t = 0, own_done = 0;                                   // next cell update; own-slot chain passes done
        // ---- cell update of step cell_t.  From step 1 on it is pending once this wave's own chain pass over h_{cell_t - 1} of the slot
        // is done, and ready when all twelve waves have posted that step's row sums (LDS counter) and the slot's 32 candidates
        // (tag cell_t, from the workers' fc2 waves) are there: g = this lane's candidate; x_{t-1} = their first argmax.
        auto cell = [&](u64 g) {
            const int t = cell_t;
            const int s = c;
            const unsigned tag = (unsigned)t + 1u;
            setprio(3 - tDP);
            XP_STAMP(s, 1); XP_WSTAMP(3, s);
            if (t > 0) {
                const unsigned uo = xp_ordered((unsigned)g);
                unsigned m = uo;
                m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, false));
                m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, false));
                m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xF, 0xF, false));
                m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x140, 0xF, 0xF, false));
                const unsigned m0 = __builtin_amdgcn_readlane(m, 0), m1 = __builtin_amdgcn_readlane(m, 16);
                const unsigned bb = max(m0, m1);
                const unsigned hit = (unsigned)__ballot(uo == bb);          // lanes 0..31: workers in class order -- the first maximum wins
                const int f0 = __ffs((int)hit) - 1;
                const int cl = (int)((g >> 32) & 255u);
                x = __builtin_amdgcn_readlane(cl, f0 < 0 ? 0 : f0);
            }
            if (lane == 0) { sinfo[s * 4 + 0] = st_lt; sinfo[s * 4 + 1] = (int)st_utt; }       // for the slot's fc2 wave: this step's noise
            float s0 = 0.f, s1 = 0.f, sn = 0.f, hold = 0.f;
            if (st_active && !st_first) {
                s0 = gsum[s * 96 + cu]; s1 = gsum[s * 96 + UPB + cu]; sn = gsum[s * 96 + 2 * UPB + cu];
                hold = hprev;
            }
            s0 += bq0; s1 += bq1; sn += bq2;
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
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the noise record is in LDS before h_t can be seen anywhere
            if (lane < (unsigned)UPB) xd_put(gh, ((unsigned)(s * NW + rank) * 32u + cu) * 8u, ((u64)tag << 32) | __float_as_uint(hn), agent);
            XP_STAMP(s, 2); XP_WSTAMP(0, s);
            __builtin_amdgcn_s_setprio(0);
            // ---- behind the publish: the sample x_{t-1} goes out (network_vocoder.py:78 output), the slot's state for step t + 1
            if (st_emit && lane == 0 && rank == (s & 31) && !((tune >> 16) & 1)) {
                float *wav = par->wav;
                int64_t *mulaw = par->mulaw;
                const size_t at = (size_t)st_erow * par->Lout + st_eidx;
                if (wav) wav[at] = mtab[x];
                if (mulaw) mulaw[at] = x;
            }
            advance(t + 1);
            cell_t += 1;
        };
        auto cell_pending = [&]() { return cell_on && cell_t < n_steps && cell_t <= own_done; };
        auto cand_there = [&](u64 g) { return (bool)__all((unsigned)(g >> 40) == (unsigned)cell_t) && lds_peek(gcnt + c) >= XP_WAVES * cell_t; };
        if (cell_on) cell(0);                                           // step 0 needs nothing: all slots start together
        // A poll is a load with a round trip of 0.3 .. 1 us, and the update is pending for half a step: the wave does not sit on the
        // poll (its chain passes would fall behind and hold up everybody: measured, 6.3 us per step).  It keeps ONE poll in flight,
        // goes on looking at the LDS counter of its next chain pass, and reads the poll's answer after XP_POLL_SPINS looks, or after
        // the pass if that became ready first.
        bool infl = false;
        u64 gp = 0;
        const bool c_static = (tune >> 15) & 1;
        const int c_lag = (tune >> 13) & 3;
        int since_own = 0;
        for (int t = 0; t < n_steps && !dead; ++t) {
            for (int s = 0; s < bx; ++s) {
                if (c_static) {
                    // ---- static schedule: the cell update sits c_lag chain passes behind the wave's own pass, and the wave waits there
                    if (cell_pending() && (since_own >= c_lag || s == c)) {
                        wt.start();
                        for (unsigned spins = 0;; ++spins) {
                            gp = ps_load(csrc);
                            if (cand_there(gp)) { cell(gp); break; }
                            if (give_up(spins)) break;
                            if (tCS) nap(tCS);
                        }
                        if (dead) break;
                    }
                    wait_cnt(hcnt + s, XP_SWEEPERS * (t + 1));
                    if (dead) break;
                    if (c == 0 && s == 1) XP_WSTAMP(4, 0);
                    XP_VSTAMP(1, s);
                    const float vv = chain_combine(chain_regs<NT_H>(w, opnd + s * HR));
                    if (sum_lane) gsum[s * 96 + 8 * wave + r8] = vv;
                    lds_bump(gcnt + s, lane);
                    if (c == 0 && s == 1) XP_WSTAMP(5, 0);
                    since_own += 1;
                    if (s == c) { own_done += 1; since_own = 0; }
                    continue;
                }
                // ---- until h_t of slot s is in LDS: the pending cell update, if its candidates arrive first (before this wave's
                // own pass over h_t of its slot they must: nobody publishes that h_t before this very update)
                wt.start();
                unsigned since = 0;
                for (unsigned spins = 0; lds_peek(hcnt + s) < XP_SWEEPERS * (t + 1); ++spins) {
                    if (cell_pending()) {
                        if (!infl) { gp = ps_load(csrc); infl = true; since = 0; }
                        else if (++since >= XP_POLL_SPINS) {
                            infl = false;
                            if (cand_there(gp)) { cell(gp); continue; }
                        }
                    }
                    if (give_up(spins)) break;
                    nap(tCS);
                }
                asm volatile("" ::: "memory");
                if (dead) break;
                if (cell_pending() && !infl) { gp = ps_load(csrc); infl = true; }           // in flight during the chain pass
                asm volatile("" ::: "memory");
                const float vv = chain_combine(chain_regs<NT_H>(w, opnd + s * HR));
                if (sum_lane) gsum[s * 96 + 8 * wave + r8] = vv;
                lds_bump(gcnt + s, lane);
                if (s == c) own_done += 1;
                if (infl) {
                    infl = false;
                    if (cand_there(gp)) cell(gp);
                }
            }
            if (((tune >> 17) & 1) && (t & 15) == 15) __syncthreads();
        }
        // ---- the last step's x has nowhere to go: every utterance ended at least one step before n_steps
    }
}

template <int BXT>
int launch_p(const XdParams &p, hipStream_t s) {
    constexpr size_t lds = sizeof(float) * (size_t)LdsP<BXT>::total;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert(BXT % 2 == 0 && BXT <= 4, "slots per XCD");
    // per launch, not once per process: the attribute belongs to the current device, and a process may hold handles on several
    HIP_TRY(hipFuncSetAttribute((const void *)ar_xcp_kernel<BXT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((ar_xcp_kernel<BXT>), dim3(8 * NW), dim3(THREADS), lds, s, p);
    HIP_TRY(hipGetLastError());
    return VQCPC_OK;
}

}  // namespace

size_t xp_exchange_bytes(int bxt) { return (size_t)CTL_WORDS * 4 + (size_t)8 * xp_region(bxt) * sizeof(u64); }
int xp_pick_bxt(int n) { return n <= 2 ? 2 : n <= XP_MAX_BX ? 4 : 0; }

int xp_launch(const XdParams &p, hipStream_t s) {
    VQ_REQUIRE(p.bxt == 2 || p.bxt == 4, "xp_launch: bxt %d", p.bxt);
    VQ_REQUIRE(p.n_slots >= 1 && p.n_slots <= 8 * p.bxt, "xp_launch: %d slots do not fit 8 x %d", p.n_slots, p.bxt);
    HIP_TRY(hipMemsetAsync(p.xg, 0, xp_exchange_bytes(p.bxt), s));
    switch (p.bxt) {
        case 2: return launch_p<2>(p, s);
        default: return launch_p<4>(p, s);
    }
}
