// search.hpp - the reference's two search loops with their state on the device.
//   * rate search (/root/reference/src/ebcc_codec.c:545-596, called at :728 and :836): halve / double / bisect the
//     JPEG 2000 rate until the fraction of samples within the error target crosses the wanted quantile;
//   * truncation bisection of the SPIHT stream (:765-796).
// The decision sequence is the reference's, evaluated unchanged (same float / double expressions in the same
// order); what moves is where it runs: a one-thread-per-chunk kernel reads the statistics the previous probe left on
// the device, advances the state machine and writes the next rate (or cut point) and the active mask, so a whole
// search is enqueued as a fixed number of rounds without the host in the loop (batch_codec.hip: device_rate_search,
// device_truncation).  Rounds in which no chunk is active cost their launches only - every probe kernel returns at
// once for inactive frames.
#pragma once

#include "j2k.hpp"

namespace ebcc {

struct DevProbe {                  // outcome of the base layer coded at one rate (a function of (chunk, rate) only)
    float cr;
    int stream_bytes;
    unsigned long long nbad;       // count(|x - d| > target)
    double err_sum;                // sum(x - d)
    int complete;                  // 0: the probe may have stopped counting early (J2kFrame::bad_limit): nbad is a lower bound that
    int pad;                       //    already settles "infeasible", err_sum is partial - a search's RESULT never rests on such a record
};

// :545-596 as a resumable state machine (phases 0-4 as in the reference's three loops and final encode)
struct DevRateSearch {
    float lo, hi, cr, result, pending;
    int phase;                     // 0 halving, 1 doubling, 2 bisect, 3 final probe, 4 search done, 5 its decode is being restored, 6 finished
    double q, q0, qt;
    int want;                      // waits for the probe at want_cr
    float want_cr;
    DevProbe last;                 // the probe the result rests on
};

constexpr int kMaxProbes = 64;     // probes on record per chunk (both searches share them); further ones are simply made again

struct DevChunk {                  // per chunk (= frame, or the frames coded as the tiles of one image)
    DevRateSearch rs[2];           // [0] error-bounded search (:728), [1] pure base layer (:836)
    float state_cr;                // rate of the decode the engine holds for this chunk (-1: none)
    int n_probes;
    int const_field;
    int pad0;
    double q;                      // quantile of the last probe search 0 looked at
    DevProbe probes[kMaxProbes];
    // truncation bisection (:765-796)
    double t_hi, t_lo, t_best, mean_err;
    float best_err, target;
    int trunc_active, trunc_pending;
};

// One round of rate search k for every chunk: take in the probe made for it in the previous round (if any), walk the
// state machine through probes already on record, and either ask for the next probe (jf[tile].cr, active[tile] = 1)
// or finish.  `unfinished` (device int) is incremented by every chunk that still wants a probe.
// cand_cr / cand_sel (J2kBuffers, may be null): speculative rate allocation - the round also says which of the previous round's
// two candidate rates is the one asked for now (cand_sel) and writes the two rates the search can ask for next (cand_cr).
// limit_qt > 0: probes that only have to tell feasible from infeasible may stop counting early (J2kFrame::bad_limit); the
// quantile target the limit is derived from is the smaller of limit_qt (the error-bounded search's, whose probes the pure
// base-layer search re-uses) and the running search's own.  0: every probe counts the whole frame.
void launch_search_advance(DevChunk *chunks, J2kFrame *jf, int *d_active, int n_chunks, int tiles, int k, double n_pix,
                           int *unfinished, hipStream_t s, float *cand_cr = nullptr, int *cand_sel = nullptr, double limit_qt = 0.0);

// One round of the truncation bisection: take in the statistics of the cut made in the previous round, choose the next
// cut (trunc_bits[f], active[f] = 1) or finish.
void launch_trunc_advance(DevChunk *chunks, FrameState *fs, unsigned long long *trunc_bits, int *d_active, int n_chunks,
                          double n_pix, int *unfinished, hipStream_t s);
// One round of the bisection with look-ahead: `levels` (1..3) iterations of :777-795 per round from the outcomes of the
// 2^levels - 1 cuts the previous round proposed (cut slots, residual.hpp); chunk f uses slots rank[f] * K .. (rank null: f;
// rank[f] < 0: the chunk is not part of this launch).  Writes cs.bits / active / frame_of / fs of its slots.
void launch_trunc_advance_multi(DevChunk *chunks, const FrameState *fs, const CutSlots &cs, int n_chunks, double n_pix, int levels,
                                const int *rank, int *unfinished, hipStream_t s);

}  // namespace ebcc
