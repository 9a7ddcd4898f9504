// One resident, weight-stationary sample-loop decoder per XCD (ar_xcd.hip) -- interface towards vocoder.hip.
#pragma once
#include "common.h"

// What a decode slot does over time: utterance `row` of the call's inputs/outputs produces `len` samples starting at the
// XCD's step t0 (its sampling-stream id is `utt`).  A slot's list ends with len == 0.
struct XdSeg { int row, t0, len; unsigned utt; };
static_assert(sizeof(XdSeg) == 16, "XdSeg is read as four ints");

#define XD_MAX_BX 4           // decode slots per XCD the kernel is instantiated for (two service waves x two slots)

struct XdParams {
    const float *w_hh;        // (3Hr, Hr) plain
    const float *w_fc1, *b_fc1, *w_fc2, *b_fc2;     // (Hf, Hr), (n_cls, Hf) plain
    const float *Gemb;        // [n_cls][3Hr]   sample embedding . W_ih[:, :de]^T
    const float *b_hh;        // [3Hr]
    const float *Gcond;       // [sum of the utterances' frames][3Hr] conditioning rows (W_ih[:, de:] cond + b_ih), ragged: utterance `row`
                              // starts at row gbase[row], gbase = (const int *)(segs + 8 * slots per XCD * max_seg): behind the table
    const float *mulaw_tab;   // [n_cls]
    const XdSeg *segs;        // [8 * bxt slots][max_seg]; slot s lives on XCD s % 8 as its local slot s / 8; then int gbase[rows]
    unsigned long long *xg;   // exchange area (xd_exchange_bytes), zeroed by xd_launch
    unsigned *status;         // host-mapped word: status_tag | 1 = an exchange timed out, | 2 = the workgroups were not dealt 32 per XCD
    unsigned status_tag;      // the call's epoch << 8 (which call of the handle reported)
    float *wav;               // (rows, Lout) or null
    int64_t *mulaw;           // (rows, Lout) or null
    unsigned long long seed;
    int max_seg, n_slots, bxt;          // bxt: slots per XCD the launch is laid out for (1, 2 or 4)
    int n_steps[8];           // steps XCD x runs (its last slot end + 1: the last sample is emitted one step later)
    int Lout, F, upsample;
    int agent_stores;         // 1: publish with agent-scope (sc1) stores instead of workgroup-scope ones (tests / A-B)
    unsigned timeout_ticks;   // bound of every in-kernel wait, 100 MHz ticks
    int dbg_drop_step;        // >= 0: rank 3 of XCD 0 skips its candidate publish at that step (exercises the abort path)
    int dbg_misplace;         // != 0: workgroup 0 reports the XCD next to its own (exercises the placement check: status 2, nothing written)
};

size_t xd_exchange_bytes(int bxt);
// Dimensions this decoder is built for (the reference's: size_h_rnn 896, size_h_fc 256, 8-bit mu-law).
bool xd_supported(int Hr, int Hf, int n_cls);
int xd_pick_bxt(int slots_per_xcd);            // 1, 2, 4 (0: too many)
int xd_launch(const XdParams &p, hipStream_t s);

// The same decoders for large batches (ar_xcm.hip): 16 decode slots per XCD on the matrix cores.  Same XdParams (bxt is
// ignored: the layout is fixed at XM_BX slots per XCD), same schedule table, same status word.
#define XM_BX 16
size_t xm_exchange_bytes();
int xm_launch(const XdParams &p, hipStream_t s);
