// ar_xcm.hip -- the per-XCD resident decoders of ar_xcd.hip for LARGE batches: 16 decode slots per XCD (128 per GPU) on
// the matrix cores (network_vocoder.py:78's sample loop; RNN_MS spec: DESIGN.md 2.2).
//
// Same partition as ar_xcd.hip: decode slot s lives on XCD s % 8; each XCD runs its own copy of the recurrence on its 32 CUs
// with a full copy of the weights held ON the CUs; workgroup `rank` owns hidden units 28 rank .. + 27 (84 gate rows of W_hh),
// 8 rows of fc1 and 8 classes of fc2.  What changes is how a row meets the state: the 84 + 8 rows (+ 4 of padding) are six
// 16-row tiles of v_mfma_f32_16x16x4_f32 against the 16 slots' h_t -- [W_hh; W_fc1] h_t is ONE product, fc1 rides along --
// with the A fragments of two (tile, K quarter) pairs pinned in the 112 VGPRs of one wave (12 waves x 2 = 6 tiles x 4 K
// quarters), so that,
// as in ar_xcd.hip, no weight is fetched again after the prologue (the launch-per-step kernel this replaces pulls 172 KB
// of W_hh into every workgroup at every sample step).  The arithmetic is the launch kernels': a row's dot product is the
// same 8 fp32 fma chains (K quarter x x/z | y/w accumulator, ar_shared.h), combined ((q0 + q1) + q2) + q3.
//
// A sample step (all 16 slots of the XCD together):
//   all waves     cell update of the 28 owned units x 16 slots (thread = (unit, slot)) from the previous step's row sums
//                 and the embedding rows of x_{t-1} (requested before the barrier that ended the previous step)
//                 -> h_t published; h_t of all 32 workgroups swept into LDS                               -> barrier A
//   all waves     2 x 56 MFMAs each: [W_hh; W_fc1] h_t, the K quarter sums of the wave's two (tile, quarter) pairs -> LDS
//   waves 0..3    FIRST the four K quarters of tile 5 (W_hh rows 80..83 + the 8 fc1 rows; the oldest waves of their SIMDs: see
//                 below) -> wave 0: fc1 + ReLU -> a_t published; then their quarter of tile 4; then, a K quarter of fc2 each:
//                 a_t swept, fc2 on the matrix pipe (A fragments from LDS), wave 0: Gumbel-max candidate of the 8 owned
//                 classes per slot published; the 32 candidates of four slots each swept, x_t = their first argmax; the
//                 sample goes out
//   wave 11       slot bookkeeping, conditioning rows and Gumbel noise of step t + 1
//   waves 0..6    the embedding rows of x_t requested                                                    -> barrier B
// Variants measured on the way and dropped (same bits each): K halves instead of quarters, tile 5 on waves 5 / 11 (11.3 us per
// step) or 0 / 1 (11.3), a_t swept by all waves behind one more barrier (10.9), fc2's eight chains on eight waves behind yet
// another (11.1); with the quarters (10.3): the other waves held back until a_t is published (10.4) or swept (11.1) -- whenever
// they multiply, waves 0..3 crawl (the sweep of a_t: 0.6 us alone, 2.5 us next to them).  hipcc sinks every B-fragment read
// down to its four MFMAs (read, lgkmcnt(0), 4 MFMAs, ...); pinning a 3-block look-ahead with sched_barrier makes the twelve
// waves advance evenly instead of oldest first -- and tile 5 later: 10.4 us; look-ahead depths 2..13 without it: no change.
// The second round of the h_t sweep in flight under an "early" quarter (K quarter 0 or 1) of every wave, a barrier, then the
// late quarters: 11.3 us (tile 5 is complete later, and the extra barrier sits in the middle of the MFMA phase); wave 0's
// quarter of tile 4 behind x_t instead of behind the fc1 publish: 11.2 us.
// Exchanges are the 8-byte {tag, value} granules of ar_xcd.hip, two per 16-byte load, laid out so that every sweep reads
// a linear array.  Every wait is wall-clock bounded (status bit 0, vqcpc_vocoder_check); placement is checked as there.
#include "ar_xcd.h"
#include "ar_shared.h"

// Timeline stamps of worker 5 of XCD 0 (100 MHz wall clock), steps 256..383, for tools/xcm_timeline.py: compiled in only
// with -DVQCPC_XD_STAMPS (a debug build under build/stamps/, never the shipped library).
#ifdef VQCPC_XD_STAMPS
__device__ unsigned long long g_xm_stamps[128 * 24];
#define XM_STAMP(wv, i) do { if (rank == 5 && xcc == 0 && wave == (wv) && (threadIdx.x & 63u) == 0 && t >= 256 && t < 384) \
        g_xm_stamps[(t - 256) * 24 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int vqcpc_debug_xm_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xm_stamps), sizeof(g_xm_stamps)) == hipSuccess ? 0 : -1;
}
#else
#define XM_STAMP(wv, i) do { } while (0)
#endif

namespace {

constexpr int HR = 896, HF = 256, NC = 256;
constexpr int NW = 32;                 // workgroups per XCD
constexpr int UPB = 28;                // hidden units per workgroup
constexpr int FPB = 8;                 // fc1 rows / fc2 classes per workgroup
constexpr int THREADS = 768;
constexpr int BX = XM_BX;              // decode slots per XCD = columns of an MFMA tile
constexpr int LROWS = 96;              // rows of a workgroup: 84 of W_hh [gate][unit], 8 of fc1, 4 of padding
// Slot strides of h_t / a_t in LDS (floats): 2 units of 16 bytes mod 16.  A B fragment is read by ds_read_b128 with lane = (slot, aq) at
// 16-byte unit S slot + aq (+ const); the instruction's four lane groups ({0-3, 12-15, 20-27}, ...) each hold all 16 slots, half of them
// with aq one higher: with S = 1 (stride HR + 4, rounds 3 and 4 until the SQ counters were read) two lanes of every group met on one
// unit -- SQ_LDS_BANK_CONFLICT 0.43 of the LDS cycles of the whole kernel (profiles/r04_pmc_sq_xcm128.json); S = 2 puts the sixteen lanes
// on sixteen units.  The sweeps store slots s and s + 8 from one exchange chunk (XM_SLOT: the exchange arrays pair them), 8 and 264 floats
// = 8 banks apart: 2-way on ds_write_b32, which costs nothing.
constexpr int HSD = HR + 8;
constexpr int ASD = HF + 8;
#define XM_SLOT(s) (2u * ((s) & 7u) + ((s) >> 3))     // position of slot s in a 16-slot row of the h_t / a_t exchange arrays
constexpr int CELLS = UPB * BX;        // cell-update threads: (unit, slot)
constexpr int BOOK = 11;               // the wave that keeps the slots' books (the youngest: it has nothing else to do behind its MFMAs)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// exchange area of one XCD, in granules
constexpr int XM_H = HR * BX;          // [column = 28 rank + unit][slot]
constexpr int XM_A = HF * BX;          // [row = 8 rank + f][slot]
constexpr int XM_C = BX * NW;          // [slot][rank]
constexpr int XM_REGION = XM_H + XM_A + XM_C;
constexpr int CTL_WORDS = 64;          // u32: arrivals per XCC [0..7], total [8]

// LDS carve, in floats (ints behind them)
struct Lds {
    static constexpr int hT = 0;                          // [BX][HSD]   h_t
    static constexpr int aT = hT + BX * HSD;              // [BX][ASD]   a_t
    static constexpr int f2a = aT + BX * ASD;             // [16 blocks][64 lanes][4]  fc2 A fragments (8 classes + 8 zero rows)
    static constexpr int part = f2a + 16 * 64 * 4;        // [4][LROWS][BX]  the K quarters q0 .. q3 of every row
    static constexpr int fcx = part + 4 * LROWS * BX;     // [4][8][BX]  fc2: the K quarters of the 8 owned classes
    static constexpr int gcl = fcx + 4 * 8 * BX;         // [BX][84]  conditioning rows in use [gate][unit]
    static constexpr int noise = gcl + BX * 84;           // [2][BX][8]
    static constexpr int mtab = noise + 2 * BX * 8;       // [NC]
    static constexpr int sinfo = mtab + NC;               // int [2][BX][8] {active, first, lt, utt, row, frame to load or -1, the utterance's first Gcond row}
    static constexpr int xs = sinfo + 2 * BX * 8;         // int [BX]  x_t
    static constexpr int segst = xs + BX;                 // int [BX][8] {index, row, t0, len, utt, samples into / index of the conditioning frame}
    static constexpr int bqs = segst + BX * 8;            // [3][32] b_hh of the owned units, then b_fc1 [8], b_fc2 [8] of the owned rows
    static constexpr int par = bqs + 112;                   // XmPar: kernel arguments needed once per step or less, read from LDS where they are used (51 SGPRs were spilled)
    static constexpr int ctl = par + 24;                   // int [12] {xcc, rank, ok, abort, [4] fc1 halves of wave 1 in LDS, [5] its fc2 halves, [7] x_t posted (counts both fc waves), [10] the call's status tag}
    static constexpr int total = ctl + 12;
};

// five 16-byte chunks (two granules each) at byte offset `off` from five uniform bases, all in flight together; sc1: served by
// L2.  (The bases are SGPR pairs: see ar_shared.h for the wait states in front.)
__device__ __forceinline__ void chunks5(u32x4 (&v)[5], const u64 *b0, const u64 *b1, const u64 *b2, const u64 *b3, const u64 *b4, unsigned off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %5, %6 sc1\n\tglobal_load_dwordx4 %1, %5, %7 sc1\n\tglobal_load_dwordx4 %2, %5, %8 sc1\n\t"
                 "global_load_dwordx4 %3, %5, %9 sc1\n\tglobal_load_dwordx4 %4, %5, %10 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4])
                 : "v"(off), "s"(b0), "s"(b1), "s"(b2), "s"(b3), "s"(b4) : "memory");
}
// eight chunks 1 KB apart from two bases 4 KB apart
__device__ __forceinline__ void chunks8(u32x4 (&v)[8], const u64 *b, unsigned off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %8, %9 sc1\n\tglobal_load_dwordx4 %1, %8, %9 offset:1024 sc1\n\t"
                 "global_load_dwordx4 %2, %8, %9 offset:2048 sc1\n\tglobal_load_dwordx4 %3, %8, %9 offset:3072 sc1\n\t"
                 "global_load_dwordx4 %4, %8, %10 sc1\n\tglobal_load_dwordx4 %5, %8, %10 offset:1024 sc1\n\t"
                 "global_load_dwordx4 %6, %8, %10 offset:2048 sc1\n\tglobal_load_dwordx4 %7, %8, %10 offset:3072 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(off), "s"(b), "s"(b + 512) : "memory");
}
__device__ __forceinline__ void chunks1(u32x4 (&v)[1], const u64 *b, unsigned off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v[0]) : "v"(off), "s"(b) : "memory");
}
// three chunks from three uniform bases
__device__ __forceinline__ void chunks3(u32x4 (&v)[3], const u64 *b0, const u64 *b1, const u64 *b2, unsigned off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %3, %4 sc1\n\tglobal_load_dwordx4 %1, %3, %5 sc1\n\tglobal_load_dwordx4 %2, %3, %6 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]) : "v"(off), "s"(b0), "s"(b1), "s"(b2) : "memory");
}
// four chunks 1 KB apart from one base
__device__ __forceinline__ void chunks4(u32x4 (&v)[4], const u64 *b, unsigned off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %5 sc1\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024 sc1\n\t"
                 "global_load_dwordx4 %2, %4, %5 offset:2048 sc1\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(off), "s"(b) : "memory");
}
__device__ __forceinline__ void chunks2(u32x4 (&v)[2], const u64 *b, unsigned off0, unsigned off1) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %4 sc1\n\tglobal_load_dwordx4 %1, %3, %4 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]) : "v"(off0), "v"(off1), "s"(b) : "memory");
}

// max / min over the 16 lanes of a DPP row, result in all of them
__device__ __forceinline__ unsigned row_max(unsigned m) {
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, false));      // quad_perm [1,0,3,2]
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, false));      // quad_perm [2,3,0,1]
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xF, 0xF, false));     // row_half_mirror
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x140, 0xF, 0xF, false));     // row_mirror
    return m;
}
__device__ __forceinline__ unsigned row_min(unsigned m) {
    m = min(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, false));
    m = min(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, false));
    m = min(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xF, 0xF, false));
    m = min(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x140, 0xF, 0xF, false));
    return m;
}
__device__ __forceinline__ unsigned ordered(unsigned u) { return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }     // order-preserving image of a float

// A value the optimiser must recompute behind this point: the sample loop keeps 112 weight registers per lane for the whole
// call, and every loop-invariant address hipcc hoists out of it (it finds dozens) is spilled to scratch.  Each phase of a
// step derives its addressing from an opaque copy of the thread index instead.
__device__ __forceinline__ unsigned opq(unsigned x) { asm volatile("" : "+v"(x)); return x; }

#define XM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Kernel arguments that are needed once per sample step or less live in LDS (ints) and are read where they are used: held in
// SGPRs for the whole call they were spilled to VGPR lanes.  (Read through the __shared__ array's own address space: a struct
// pointer cast from it made hipcc emit FLAT loads with system scope, 0.25 us per step slower than the spills.)
enum { PAR_WAV = 0, PAR_MULAW = 2, PAR_SEGS = 4, PAR_GCOND = 6, PAR_SEED = 8, PAR_LOUT = 10, PAR_MAXSEG, PAR_F, PAR_UPS, PAR_DROP, PAR_GEMB = 16, PAR_WORDS = 24 };
__device__ __forceinline__ int par_i(const int *par, int i) { return __hip_atomic_load(par + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ unsigned long long par_q(const int *par, int i) { return (unsigned long long)(unsigned)par_i(par, i) | ((unsigned long long)(unsigned)par_i(par, i + 1) << 32); }
// a pointer rebuilt from those words, in the GLOBAL address space (from a plain integer cast hipcc would emit flat_ loads / stores)
#define PAR_GLOBAL(T, par, i) ((T __attribute__((address_space(1))) *)par_q(par, i))
__device__ __forceinline__ void par_set(int *par, int i, unsigned long long v) { par[i] = (int)(unsigned)v; par[i + 1] = (int)(unsigned)(v >> 32); }

__global__ __launch_bounds__(THREADS) void ar_xcm_kernel(XdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *hT = smem + Lds::hT, *aT = smem + Lds::aT, *f2a = smem + Lds::f2a, *part = smem + Lds::part, *fcx = smem + Lds::fcx;
    float *gcl = smem + Lds::gcl, *noise = smem + Lds::noise, *mtab = smem + Lds::mtab;
    int *sinfo = (int *)(smem + Lds::sinfo), *xs = (int *)(smem + Lds::xs), *s_ctl = (int *)(smem + Lds::ctl), *segst = (int *)(smem + Lds::segst);
    float *bqs = smem + Lds::bqs;
    int *par = (int *)(smem + Lds::par);

    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));

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
        if (!ok) __hip_atomic_store(p.status, p.status_tag | 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        par_set(par, PAR_WAV, (unsigned long long)p.wav); par_set(par, PAR_MULAW, (unsigned long long)p.mulaw); par_set(par, PAR_SEGS, (unsigned long long)p.segs);
        par_set(par, PAR_GCOND, (unsigned long long)p.Gcond); par_set(par, PAR_SEED, p.seed); par_set(par, PAR_GEMB, (unsigned long long)p.Gemb);
        par[PAR_LOUT] = p.Lout; par[PAR_MAXSEG] = p.max_seg; par[PAR_F] = p.F; par[PAR_UPS] = p.upsample; par[PAR_DROP] = p.dbg_drop_step;
        s_ctl[0] = (int)xid; s_ctl[1] = (int)r; s_ctl[2] = ok; s_ctl[3] = 0; s_ctl[4] = 0; s_ctl[5] = 0; s_ctl[6] = 0; s_ctl[7] = 0; s_ctl[8] = 0; s_ctl[9] = 0; s_ctl[10] = (int)p.status_tag;
    }
    __syncthreads();
    const int xcc = __builtin_amdgcn_readfirstlane(s_ctl[0]), rank = __builtin_amdgcn_readfirstlane(s_ctl[1]);
    if (__builtin_amdgcn_readfirstlane(s_ctl[2]) == 0) return;
    int bx = (p.n_slots - xcc + 7) / 8;                // slots of this XCD: xcc, xcc + 8, ...
    bx = bx < 0 ? 0 : (bx > BX ? BX : bx);
    const int n_steps = p.n_steps[xcc];
    if (bx == 0 || n_steps <= 0) return;
    const int agent = p.agent_stores;

    u64 *gh = p.xg + CTL_WORDS / 2 + (size_t)xcc * XM_REGION;
    u64 *ga = gh + XM_H, *gc = ga + XM_A;

    // ---- this wave's two (tile, K quarter) pairs; their A fragments (2 x 56 registers) stay for the whole call.
    // The SIMD's instruction arbiter serves its OLDEST wave first (tools/microbench_mfma4x4.hip: of three waves in the same MFMA
    // loop the lowest wave id is through its 112 MFMAs after 1.69 us, the next after 3.37, the youngest after 5.03 -- one MFMA per
    // 15 ns per SIMD whatever the number of waves, so a step's 336 MFMAs per SIMD are 5.1 us of matrix pipe; and a young wave doing
    // ordinary work next to older waves in their MFMA loop gets next to no issue slots: ~30 instructions took up to 2.7 us;
    // s_setprio changes neither).  Everything that is serial in a step hangs on tile 5 (W_hh rows 80..83 + the 8 fc1 rows +
    // padding: fc1 -> a_t -> fc2 -> draw -> x_t), so its four K quarters are the FIRST work of waves 0..3 -- the oldest wave
    // of each SIMD: 56 MFMAs instead of 112 until a_t can go out -- and those waves keep the serial path (one K quarter of fc2
    // each); their second pair is a quarter of tile 4.  Waves 4..11 hold the two quarters of a K half of tiles 0..3.
    const bool fcw = wave < 4;
    const int tlA = fcw ? 5 : (wave - 4) >> 1, kwA = fcw ? wave : 2 * ((wave - 4) & 1);
    const int tlB = fcw ? 4 : tlA, kwB = fcw ? wave : kwA + 1;
    const unsigned arow = lane & 15u, aq = lane >> 4;
    float wr[112];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int lr = 16 * (h ? tlB : tlA) + (int)arow, kwq = h ? kwB : kwA;
        const float *Wrow = lr < 84 ? p.w_hh + (size_t)((lr / UPB) * HR + UPB * rank + lr % UPB) * HR
                          : lr < 92 ? p.w_fc1 + (size_t)(FPB * rank + lr - 84) * HR : nullptr;
#pragma unroll
        for (int b = 0; b < 14; ++b) {                                 // 16-column block 14 kw + b of the K quarter
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (Wrow) v = *(const float4 *)(Wrow + 16 * (14 * kwq + b) + 4 * aq);
            wr[56 * h + 4 * b] = v.x; wr[56 * h + 4 * b + 1] = v.y; wr[56 * h + 4 * b + 2] = v.z; wr[56 * h + 4 * b + 3] = v.w;
        }
    }

    // ---- resident LDS state
    for (unsigned e = tid; e < 16 * 64 * 4; e += THREADS) {
        const unsigned blk = e >> 8, ln = (e >> 2) & 63u, comp = e & 3u, row = ln & 15u, qq = ln >> 4;
        f2a[e] = row < FPB ? p.w_fc2[(size_t)(FPB * rank + row) * HF + 16 * blk + 4 * qq + comp] : 0.f;
    }
    for (unsigned e = tid; e < NC; e += THREADS) mtab[e] = p.mulaw_tab[e];
    for (unsigned e = tid; e < 4 * LROWS * BX; e += THREADS) part[e] = 0.f;
    for (unsigned e = tid; e < BX * 84; e += THREADS) gcl[e] = 0.f;
    for (unsigned e = tid; e < 2 * BX * 8; e += THREADS) sinfo[e] = 0;
    for (unsigned e = tid; e < BX; e += THREADS) xs[e] = NC / 2;
    __syncthreads();

    Waiter wt{p.status, p.timeout_ticks, 0};
    int *s_abort = s_ctl + 3;

    for (unsigned e = tid; e < 96; e += THREADS) { const unsigned g = e >> 5, u = e & 31u; bqs[e] = u < UPB ? p.b_hh[g * HR + UPB * rank + u] : 0.f; }
    for (unsigned e = tid; e < 16; e += THREADS) bqs[96 + e] = e < 8 ? p.b_fc1[FPB * rank + e] : p.b_fc2[FPB * rank + e - 8];
    for (unsigned e = tid; e < BX; e += THREADS) {
        const XdSeg sg = (int)e < bx ? p.segs[(size_t)(xcc + 8 * e) * p.max_seg] : XdSeg{-1, 0, 0, 0u};
        int *st = segst + e * 8;
        st[0] = 0; st[1] = sg.len > 0 ? sg.row : -1; st[2] = sg.t0; st[3] = sg.len; st[4] = (int)sg.utt; st[5] = 0; st[6] = 0;
        st[7] = sg.len > 0 ? ((const int *)(p.segs + (size_t)8 * BX * p.max_seg))[sg.row] : 0;      // the utterance's first Gcond row
    }
    __syncthreads();

    // ---- slot bookkeeping (lane b < 16 of wave BOOK = slot b; the segment state lives in LDS): what step `tn` needs that does
    // not depend on the data: sinfo[tn & 1][slot] = {active, first, lt, utt, row, conditioning frame to load or -1}
    auto advance = [&](int tn, unsigned ln) {
        if (ln >= (unsigned)BX) return;
        int *st = segst + ln * 8;
        int sg_i = st[0], sg_row = st[1], sg_t0 = st[2], sg_len = st[3], sg_fpos = st[5], sg_fidx = st[6], sg_gb = st[7];
        unsigned sg_utt = (unsigned)st[4];
        int lt = tn - sg_t0;
        if (sg_row >= 0 && lt >= sg_len) {                        // next utterance of this slot
            sg_i += 1;
            XdSeg sg = XdSeg{-1, 0, 0, 0u};
            const int max_seg = par_i(par, PAR_MAXSEG);
            if (sg_i < max_seg) {
                auto sp = PAR_GLOBAL(const int, par, PAR_SEGS) + 4 * ((size_t)(xcc + 8 * ln) * max_seg + sg_i);      // XdSeg = 4 ints
                sg = XdSeg{sp[0], sp[1], sp[2], (unsigned)sp[3]};
            }
            sg_row = sg.len > 0 ? sg.row : -1; sg_t0 = sg.t0; sg_len = sg.len; sg_utt = sg.utt;
            lt = tn - sg_t0;
            sg_fpos = 0; sg_fidx = 0;
            sg_gb = sg_row >= 0 ? (PAR_GLOBAL(const int, par, PAR_SEGS) + 4 * ((size_t)8 * BX * max_seg))[sg_row] : 0;
            st[0] = sg_i; st[1] = sg_row; st[2] = sg_t0; st[3] = sg_len; st[4] = (int)sg_utt; st[7] = sg_gb;
        }
        const bool active = sg_row >= 0 && lt >= 0 && lt < sg_len;
        int frame = -1;
        if (active) {
            if (sg_fpos == par_i(par, PAR_UPS)) { sg_fpos = 0; sg_fidx += 1; }
            if (sg_fpos == 0) { const int F = par_i(par, PAR_F); frame = sg_fidx < F ? sg_fidx : F - 1; }      // next conditioning frame (once per hop)
            sg_fpos += 1;
        }
        st[5] = sg_fpos; st[6] = sg_fidx;
        int *si = sinfo + ((tn & 1) * BX + (int)ln) * 8;
        si[0] = active ? 1 : 0; si[1] = lt == 0 ? 1 : 0; si[2] = lt; si[3] = (int)sg_utt; si[4] = sg_row; si[5] = frame; si[6] = sg_gb;
    };
    // conditioning rows of the slots that enter a new frame at step tn, and the Gumbel noise of that step's draw (wave BOOK)
    auto prepare = [&](int tn, unsigned ln) {
        const int *sn = sinfo + (tn & 1) * BX * 8;
        unsigned long long fresh = __ballot(ln < (unsigned)bx && sn[(ln & 15u) * 8 + 5] >= 0);       // slots entering a new frame (once per hop each)
        while (fresh) {
            const int b = __ffsll((long long)fresh) - 1;
            fresh &= fresh - 1;
            const int f = sn[b * 8 + 5], gb = sn[b * 8 + 6];
            for (unsigned e = ln; e < 84; e += 64) {
                const unsigned g = e / UPB, u = e - g * UPB;
                gcl[b * 84 + e] = PAR_GLOBAL(const float, par, PAR_GCOND)[((size_t)gb + f) * 3 * HR + g * HR + UPB * rank + u];
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int b = (int)(ln >> 3) + 8 * h;
            if (b < bx) {
                const unsigned cls = FPB * rank + (ln & 7u);
                const unsigned long long seed = par_q(par, PAR_SEED);
                const unsigned wd = philox_word((unsigned)sn[b * 8 + 2], (unsigned)sn[b * 8 + 3], cls >> 2, (unsigned)seed, (unsigned)(seed >> 32), (int)(cls & 3u));
                noise[((tn & 1) * BX + b) * 8 + (ln & 7u)] = gumbel_from_word(wd);
            }
        }
    };
    // LDS hand-offs inside the workgroup are ordered by construction: one wave's LDS instructions are performed in order, so a
    // flag written after the data is seen after the data, and a read issued after the flag was seen sees the data.  A
    // release fence would also wait for the wave's outstanding global stores (published granules, samples going out to HBM):
    // ~1 us, on the critical path.  So: relaxed accesses, lgkmcnt waits, and the compiler kept from reordering.
    auto lds_fence = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    auto xm_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto aborted = [&]() { return __hip_atomic_load(s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0; };
    if (wave == BOOK) { advance(0, lane); lds_fence(); prepare(0, lane); }
    ps_barrier();

    float hprev = 0.f, e0 = 0.f, e1 = 0.f, e2 = 0.f;     // cell threads: h_{t-1} of (unit, slot); embedding rows of x_{t-1}
    for (int t = 0; t < n_steps; ++t) {
        const unsigned tag = (unsigned)t + 1u;
        // ---- cell update (nn.GRU cell of RNN_MS): thread (unit cu, slot cs); row sums of step t - 1 from LDS
        XM_STAMP(0, 0); XM_STAMP(2, 12);
        {
            const unsigned td = opq(tid);
            if (td < (unsigned)CELLS) {
                const unsigned cu = td >> 4, cs = td & 15u;
                const int *si = sinfo + ((t & 1) * BX + (int)cs) * 8;
                const bool active = (int)cs < bx && si[0] != 0, first = si[1] != 0;
                // the embedding rows of the sample fed in were requested before barrier B; an utterance's first step takes class NC / 2
                if (active && first) {
                    auto gp = PAR_GLOBAL(const float, par, PAR_GEMB) + (size_t)(NC / 2) * 3 * HR + UPB * rank + cu;
                    e0 = gp[0]; e1 = gp[HR]; e2 = gp[2 * HR];
                }
                float s0 = 0.f, s1 = 0.f, sn = 0.f, hold = 0.f;
                if (active && !first) {
                    const float *pp = part + cu * BX + cs;
                    constexpr int Q = LROWS * BX;
                    s0 = ((pp[0] + pp[Q]) + pp[2 * Q]) + pp[3 * Q];
                    s1 = ((pp[UPB * BX] + pp[Q + UPB * BX]) + pp[2 * Q + UPB * BX]) + pp[3 * Q + UPB * BX];
                    sn = ((pp[2 * UPB * BX] + pp[Q + 2 * UPB * BX]) + pp[2 * Q + 2 * UPB * BX]) + pp[3 * Q + 2 * UPB * BX];
                    hold = hprev;
                }
                s0 += bqs[cu]; s1 += bqs[32 + cu]; sn += bqs[64 + cu];
                float hn = 0.f;
                if (active) {
                    const float *cp = gcl + cs * 84 + cu;
                    const float r = sigmoidf_((e0 + cp[0]) + s0);
                    const float z = sigmoidf_((e1 + cp[UPB]) + s1);
                    const float nn = tanhf((e2 + cp[2 * UPB]) + r * sn);
                    hn = (1.0f - z) * nn + z * hold;
                    hprev = hn;
                }
                xd_put(gh, ((unsigned)(UPB * rank + cu) * BX + XM_SLOT(cs)) * 8u, ((u64)tag << 32) | __float_as_uint(hn), agent);
            }
        }
        XM_STAMP(0, 1);
        // ---- h_t of all 32 workers: 7168 16-byte chunks (column, slot pair), a linear array; ten per thread in two rounds (all
        // ten in flight together would need 40 registers next to the 112 weights: measured, spills)
#pragma unroll 1
        for (int rnd = 0; rnd < 2; ++rnd) {
            const unsigned td = opq(tid);
            u32x4 v[5];
            const u64 *b0 = gh + (size_t)(5 * rnd) * 1536;
            wt.start();
            for (unsigned spins = 0;; ++spins) {
                chunks5(v, b0, b0 + 1536, b0 + 2 * 1536, b0 + 3 * 1536, b0 + 4 * 1536, td * 16u);
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 5; ++i) ok &= (td + 768u * (5 * rnd + i) >= (unsigned)(XM_H / 2)) || (v[i].y == tag && v[i].w == tag);
                if (__all(ok)) break;
                if (wt.expired(spins, (int)(td & 63u), s_abort + 7)) { *s_abort = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const unsigned L = td + 768u * (5 * rnd + i);
                if (L < (unsigned)(XM_H / 2)) {
                    const unsigned k = L >> 3, pr = L & 7u;
                    hT[pr * HSD + k] = __uint_as_float(v[i].x);                 // chunk pr of a column: slots pr and pr + 8
                    hT[(pr + 8) * HSD + k] = __uint_as_float(v[i].z);
                }
            }
        }
        XM_STAMP(0, 2); XM_STAMP(2, 13);
        xm_barrier();                                                // A: h_t in LDS
        if (*s_abort) break;
        XM_STAMP(0, 3); XM_STAMP(2, 14);

        // ---- [W_hh; W_fc1] h_t: a K quarter of a tile = 14 blocks of 16 columns, four MFMAs each (x, z -> accumulator 0; y, w ->
        // accumulator 1); the quarter sum (D fragment: register i of lane (aq, arow) = row 4 aq + i of the tile, slot arow) -> LDS
        auto quarter = [&](const float *wq, int tlq, int kwq) -> v4f {
            const unsigned ln = opq(lane);
            const float4 *hb = (const float4 *)(hT + (ln & 15u) * HSD + 224 * kwq + 4 * (ln >> 4));     // B fragments: h_t[slot][16 (14 kw + b) + 4 aq ..]
            v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            float4 q[3];
            q[0] = hb[0]; q[1] = hb[4];
#pragma unroll
            for (int b = 0; b < 14; ++b) {
                if (b + 2 < 14) q[(b + 2) % 3] = hb[4 * (b + 2)];
                const float4 hv = q[b % 3];
                acc0 = XM_MFMA(wq[4 * b + 0], hv.x, acc0);
                acc1 = XM_MFMA(wq[4 * b + 1], hv.y, acc1);
                acc0 = XM_MFMA(wq[4 * b + 2], hv.z, acc0);
                acc1 = XM_MFMA(wq[4 * b + 3], hv.w, acc1);
            }
            const v4f qs = acc0 + acc1;
            float *pw = part + ((size_t)kwq * LROWS + 16 * tlq + 4 * (ln >> 4)) * BX + (ln & 15u);
#pragma unroll
            for (int i = 0; i < 4; ++i) pw[i * BX] = qs[i];
            return qs;
        };
        const v4f qA = quarter(wr, tlA, kwA);
        XM_STAMP(0, 4); XM_STAMP(1, 16); XM_STAMP(4, 15);
        if (fcw) {
            // ================================================================  waves 0..3: tile 5 first, then everything behind fc1
            if (wave != 0) {
                asm volatile("" ::: "memory");
                if (opq(lane) == 0) __hip_atomic_fetch_add(s_ctl + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     // this quarter of tile 5 is in LDS
                XM_STAMP(1, 19);
            } else {
                // ---- fc1 + ReLU -> a_t published: tile rows 4..11 = fc1 rows 0..7 (lanes aq = 1, 2); quarter 0 is in registers
                const unsigned ln = opq(lane), aq = ln >> 4, arow = ln & 15u;
                wt.start();
                for (unsigned spins = 0; __hip_atomic_load(s_ctl + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 3 * (int)tag; ++spins)
                    if (wt.expired(spins, (int)ln, s_abort + 7) || aborted()) { *s_abort = 1; break; }
                asm volatile("" ::: "memory");
                XM_STAMP(0, 18);
                if (aq == 1 || aq == 2) {
                    const float *pr = part + (80 + 4 * aq) * BX + arow;
                    constexpr int Q = LROWS * BX;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int f = 4 * ((int)aq - 1) + i;
                        float v = ((qA[i] + pr[Q + i * BX]) + pr[2 * Q + i * BX]) + pr[3 * Q + i * BX];
                        v += bqs[96 + f];
                        v = v > 0.f ? v : 0.f;
                        xd_put(ga, ((unsigned)(FPB * rank + f) * BX + XM_SLOT(arow)) * 8u, ((u64)tag << 32) | __float_as_uint(v), agent);
                    }
                }
                XM_STAMP(0, 5);
            }
        }
        (void)quarter(wr + 56, tlB, kwB);
        XM_STAMP(0, 22); XM_STAMP(1, 23);
        if (wave == BOOK) {
            // ---- bookkeeping, conditioning rows and Gumbel noise of step t + 1
            const unsigned ln = opq(lane);
            advance(t + 1, ln);
            lds_fence();
            prepare(t + 1, ln);
        }
        if (fcw) {
            // ---- a_t: wave kw takes the rows of fc2's K quarter kw (64 kw .. + 63) of all 16 slots: 512 chunks, 8 per lane
            {
                const unsigned ln = opq(lane);
                u32x4 v[8];
                const u64 *b0 = ga + (size_t)(64 * wave) * BX;
                wt.start();
                for (unsigned spins = 0;; ++spins) {
                    chunks8(v, b0, ln * 16u);
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < 8; ++i) ok &= v[i].y == tag && v[i].w == tag;
                    if (__all(ok)) break;
                    if (wt.expired(spins, (int)ln, s_abort + 7) || aborted()) { *s_abort = 1; break; }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned c = ln + 64u * i;
                    const unsigned r = 64u * wave + (c >> 3), pr = c & 7u;
                    aT[pr * ASD + r] = __uint_as_float(v[i].x);
                    aT[(pr + 8) * ASD + r] = __uint_as_float(v[i].z);
                }
            }
            XM_STAMP(0, 6);
            XM_STAMP(0, 7);
            // ---- fc2 on the matrix pipe: this wave's K quarter = 4 blocks of 16 columns; rows 0..7 = the owned classes
            v4f fq;
            {
                const unsigned ln = opq(lane);
                v4f a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                const float4 *ab = (const float4 *)(aT + (ln & 15u) * ASD + 64 * wave + 4 * (ln >> 4));
                const float4 *wb = (const float4 *)f2a + (4 * wave) * 64 + ln;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float4 av = ab[4 * b], wv = wb[64 * b];
                    a0 = XM_MFMA(wv.x, av.x, a0);
                    a1 = XM_MFMA(wv.y, av.y, a1);
                    a0 = XM_MFMA(wv.z, av.z, a0);
                    a1 = XM_MFMA(wv.w, av.w, a1);
                }
                fq = a0 + a1;
            }
            XM_STAMP(0, 20);
            {
                const unsigned ln = opq(lane), aq = ln >> 4, arow = ln & 15u;
                if (wave != 0) {
                    if (aq < 2) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) fcx[(wave * 8 + 4 * aq + i) * BX + arow] = fq[i];
                    }
                    asm volatile("" ::: "memory");
                    if (ln == 0) __hip_atomic_fetch_add(s_ctl + 5, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    wt.start();
                    for (unsigned spins = 0; __hip_atomic_load(s_ctl + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 3 * (int)tag; ++spins)
                        if (wt.expired(spins, (int)ln, s_abort + 7) || aborted()) { *s_abort = 1; break; }
                    asm volatile("" ::: "memory");
                    XM_STAMP(0, 21);
                    // ---- Gumbel-max candidate of the 8 owned classes for slot arow: lanes aq = 0 (classes 0..3), 1 (4..7)
                    float best = 0.f;
                    int kb = 0;
                    if (aq < 2) {
                        const float *nz = noise + ((t & 1) * BX + arow) * 8 + 4 * aq;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float *fp = fcx + (4 * aq + i) * BX + arow;
                            float v = ((fq[i] + fp[8 * BX]) + fp[16 * BX]) + fp[24 * BX];
                            v += bqs[104 + 4 * aq + i];
                            const float sc = v + nz[i];
                            if (i == 0 || sc > best) { best = sc; kb = 4 * (int)aq + i; }
                        }
                    }
                    const float ob = __shfl(best, (int)ln + 16);
                    const int ok2 = __shfl(kb, (int)ln + 16);
                    if (ob > best) { best = ob; kb = ok2; }                  // lanes 0..15: first maximum over the 8 classes
                    const int dstep = par_i(par, PAR_DROP);
                    const bool drop = dstep >= 0 && t == dstep && rank == 3 && xcc == 0;
                    if (ln < (unsigned)BX && !drop)
                        xd_put(gc, (ln * NW + (unsigned)rank) * 8u, ((u64)((tag << 8) | (unsigned)(FPB * rank + kb)) << 32) | __float_as_uint(best), agent);
                }
            }
            XM_STAMP(0, 8);
            // ---- x_t: the 32 candidates of a slot = 16 chunks = one DPP row; wave w takes slots 4 w .. 4 w + 3
            {
                const unsigned ln = opq(lane);
                const int *si = sinfo + (t & 1) * BX * 8;
                u32x4 v[1];
                const unsigned c0 = 64u * wave + ln;                              // chunk = slot * 16 + rank pair
                wt.start();
                for (unsigned spins = 0;; ++spins) {
                    chunks1(v, gc, c0 * 16u);
                    const bool ok = (v[0].y >> 8) == tag && (v[0].w >> 8) == tag;
                    if (__all(ok)) break;
                    if (wt.expired(spins, (int)ln, s_abort + 7) || aborted()) { *s_abort = 1; break; }
                }
                {
                    const unsigned u0 = ordered(v[0].x), u1 = ordered(v[0].z);
                    const unsigned u = u1 > u0 ? u1 : u0;                 // classes ascend with the rank: the first maximum wins
                    const unsigned cl = (u1 > u0 ? v[0].w : v[0].y) & 255u;
                    const unsigned m = row_max(u);
                    const unsigned x = row_min(u == m ? cl : 0xFFFFu);
                    const int b = (int)(c0 >> 4);
                    if ((ln & 15u) == 0 && b < bx) {
                        xs[b] = (int)x;
                        if (si[b * 8 + 0] && rank == (b & 31)) {          // the sample goes out (network_vocoder.py:78 output)
                            const size_t at = (size_t)si[b * 8 + 4] * par_i(par, PAR_LOUT) + si[b * 8 + 2];
                            auto wav = PAR_GLOBAL(float, par, PAR_WAV);
                            auto mulaw = PAR_GLOBAL(int64_t, par, PAR_MULAW);
                            if (wav) wav[at] = mtab[x];
                            if (mulaw) mulaw[at] = (int64_t)x;
                        }
                    }
                }
                lds_fence();
                if (ln == 0) __hip_atomic_fetch_add(s_ctl + 7, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     // x_t of this wave's slots is in LDS
                XM_STAMP(0, 9);
            }
            XM_STAMP(0, 10); XM_STAMP(1, 17);
        }
        // ---- the embedding rows of x_t, requested before barrier B (both fc waves have posted their slots' x_t: word 7 counts
        // them) and used by the next cell update; an utterance's first step takes class NC / 2 instead (decided behind the barrier)
        if (tid < (unsigned)CELLS) {
            const unsigned td = opq(tid), cu = td >> 4, cs = td & 15u;
            wt.start();
            for (unsigned spins = 0; __hip_atomic_load(s_ctl + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 4 * (int)tag; ++spins) {
                if (wt.expired(spins, (int)(td & 63u), s_abort + 7) || aborted()) { *s_abort = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            asm volatile("" ::: "memory");
            auto gp = PAR_GLOBAL(const float, par, PAR_GEMB) + (size_t)xs[cs] * 3 * HR + UPB * rank + cu;
            e0 = gp[0]; e1 = gp[HR]; e2 = gp[2 * HR];
        }
        xm_barrier();                                                // B: row sums, x_t, step t + 1's bookkeeping in LDS
        if (*s_abort) break;
    }
}

}  // namespace

size_t xm_exchange_bytes() { return (size_t)CTL_WORDS * 4 + (size_t)8 * XM_REGION * sizeof(u64); }

int xm_launch(const XdParams &p, hipStream_t s) {
    constexpr size_t lds = sizeof(float) * (size_t)Lds::total;
    static_assert(lds <= 160 * 1024, "LDS budget");
    VQ_REQUIRE(p.n_slots >= 1 && p.n_slots <= 8 * XM_BX, "xm_launch: %d slots do not fit 8 x %d", p.n_slots, XM_BX);
    // per launch, not once per process: the attribute belongs to the current device, and a process may hold handles on several
    HIP_TRY(hipFuncSetAttribute((const void *)ar_xcm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipMemsetAsync(p.xg, 0, xm_exchange_bytes(), s));
    hipLaunchKernelGGL(ar_xcm_kernel, dim3(8 * NW), dim3(THREADS), lds, s, p);
    HIP_TRY(hipGetLastError());
    return VQCPC_OK;
}
