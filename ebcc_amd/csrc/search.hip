// search.hip - device-side state machines of the rate search and the truncation bisection (search.hpp).
#include "search.hpp"

namespace ebcc {
namespace {

constexpr int kMainHeaderBytes = 135;      // SOC, SIZ, COD, QCD, COM of every codestream the codec writes

// ---- /root/reference/src/ebcc_codec.c:545-596, one probe per call --------------------------------------
// returns true and sets `out` if a probe at rate `out` is needed next
__device__ bool rs_next(DevRateSearch &r, float &out)
{
    for (;;) {
        if (r.phase == 0) {
            if (r.q < r.qt && r.lo >= 1. / 2) { r.lo /= 2; out = r.pending = r.lo; return true; }          // :559-563
            r.q = r.q0; r.phase = 1;
        } else if (r.phase == 1) {
            if (r.q >= r.qt && r.hi <= 1000) { r.hi *= 2; out = r.pending = r.hi; return true; }           // :565-569
            if (r.q >= r.qt) { r.result = r.hi; r.phase = 4; return false; }                             // :571-574
            r.q = r.q0; r.phase = 2;
        } else if (r.phase == 2) {
            const double eps = 1e-8;
            if ((fabs(r.q - r.qt) > eps || r.q == 1.0) && r.hi - r.lo > 1.) {                             // :579-588
                r.cr = (r.lo + r.hi) / 2; out = r.pending = r.cr; return true;
            }
            r.phase = 3; out = r.pending = r.lo; return true;                                             // :590
        } else {
            return false;
        }
    }
}
__device__ void rs_feed(DevRateSearch &r, double quantile)
{
    r.q = quantile;
    if (r.phase == 2) { if (r.q < r.qt) r.hi = r.cr; else r.lo = r.cr; }
    else if (r.phase == 3) { r.result = r.lo; r.phase = 4; }
}
__device__ const DevProbe *find_probe(const DevChunk &c, float cr)
{
    for (int i = 0; i < c.n_probes; i++) if (c.probes[i].cr == cr) return &c.probes[i];
    return nullptr;
}

// Only the probes the residual layer can be computed from store their decoded field: the final probe of search 0 (:590)
// and the probe that restores it (phase 5); the others are asked for the statistics alone (J2kFrame::keep), which spares
// the frame a full-field write per probe.
__device__ bool keeps_field(int k, int phase) { return k == 0 && (phase == 3 || phase == 5); }

// The rate the search would ask for next if the probe it is waiting for (at R.want_cr) came back with quantile q_hyp
// (1.0: "feasible", 0.0: "not feasible" - the two ways a step of :545-596 can go); -1 if it would ask for none.  Works
// on copies: a guess for the speculative rate allocation, the search itself is advanced by k_search_advance alone.
__device__ float next_rate_if(const DevChunk &C, const DevRateSearch &R, int k, double q_hyp, double n_pix)
{
    if (R.phase > 3) return -1.0f;                                      // (the probe restoring the decode is the last one)
    DevRateSearch T = R;
    rs_feed(T, q_hyp);
    const float scr = k != 0 ? C.state_cr : (keeps_field(k, R.phase) ? R.want_cr : -1.0f);     // the state the engine will hold then
    while (T.phase < 4) {
        float cr;
        if (!rs_next(T, cr)) break;
        const DevProbe *rec = find_probe(C, cr);
        const bool known = rec != nullptr || cr == R.want_cr;
        const bool needs_state = (k == 0 && T.phase == 3 && scr != cr) || (T.phase == 3 && rec && !rec->complete);
        if (!known || needs_state) return cr;
        rs_feed(T, rec ? 1. - ((double) rec->nbad / n_pix) : q_hyp);
    }
    if (T.phase == 4 && k == 0 && scr != T.result) return T.result;
    return -1.0f;
}

// The count of samples above the target from which a probe is infeasible for both searches beyond the bisection's
// tolerance: q(m) = 1. - m / n_pix (:512) falls with m, so there is a largest feasible count T (q(T) >= qt), and from
// T + 2 + ceil(1e-8 n_pix) on q(m) < qt and |q(m) - qt| > 1e-8 hold - every test :559-588 makes of q comes out as it
// does for the exact count.  0: no such shortcut (the count is needed, or the arithmetic here is not trusted).
__device__ unsigned int infeasible_from(double qt, double n_pix)
{
    if (!(qt > 0.0 && qt <= 1.0) || !(n_pix >= 1.0 && n_pix < 2e9)) return 0;
    double m = floor((1.0 - qt) * n_pix) - 2.0;
    if (m < 0) m = 0;
    if (!(1. - (m / n_pix) >= qt)) return 0;                           // (m must be feasible to start from)
    int guard = 0;
    while (1. - ((m + 1.0) / n_pix) >= qt) { m += 1.0; if (++guard > 16) return 0; }
    const double from = m + 2.0 + ceil(1e-8 * n_pix);
    if (!(1. - (from / n_pix) < qt) || !(fabs((1. - (from / n_pix)) - qt) > 1e-8) || from > 4.0e9) return 0;
    return (unsigned int) from;
}

__global__ void k_search_advance(DevChunk *chunks, J2kFrame *jf, int *active, int n_chunks, int tiles, int k, double n_pix,
                                 int *unfinished, float *cand_cr, int *cand_sel, double limit_qt)
{
    const int ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= n_chunks) return;
    DevChunk &C = chunks[ci];
    DevRateSearch &R = C.rs[k];
    const int t0 = ci * tiles;
    // speculative rate allocation: which of the previous round's two candidates (if any) is the rate asked for now, and the
    // two candidates for the next round
    const float old0 = cand_cr ? cand_cr[2 * t0] : -1.0f, old1 = cand_cr ? cand_cr[2 * t0 + 1] : -1.0f;
    auto set_active = [&](int v) {
        for (int t = 0; t < tiles; t++) active[t0 + t] = v;
        if (cand_cr) {
            int sel = -1;
            float c0 = -1.0f, c1 = -1.0f;
            if (v) {
                sel = R.want_cr == old0 ? 0 : (R.want_cr == old1 ? 1 : -1);
                c0 = next_rate_if(C, R, k, 1.0, n_pix);
                c1 = next_rate_if(C, R, k, 0.0, n_pix);
                if (c1 == c0) c1 = -1.0f;
            }
            for (int t = 0; t < tiles; t++) { cand_sel[t0 + t] = sel; cand_cr[2 * (t0 + t)] = c0; cand_cr[2 * (t0 + t) + 1] = c1; }
        }
    };
    if (C.const_field || R.phase == 6) { set_active(0); return; }
    auto feed = [&](const DevProbe &rec) {
        if (R.phase == 3) R.last = rec;
        const double q = 1. - ((double) rec.nbad / n_pix);                                                // :512
        if (k == 0) C.q = q;
        rs_feed(R, q);
    };
    if (R.want) {                                                        // the probe asked for in the previous round
        DevProbe rec{jf[t0].cr, 0, 0ull, 0.0, 1, 0};
        int body = 0;
        for (int t = 0; t < tiles; t++) { rec.nbad += jf[t0 + t].nbad; rec.err_sum += jf[t0 + t].err_sum; body += jf[t0 + t].body_bytes; }
        rec.stream_bytes = kMainHeaderBytes + tiles * 14 + body + 2;     // main header, SOT + SOD per tile, EOC
        rec.complete = jf[t0].bad_limit == 0 || rec.nbad < jf[t0].bad_limit;   // (a count below the limit: no piece left early)
        // search 0's probes change the layer assignment the codestream is written from; only those that also stored
        // their field leave the engine in the state of one rate (decode and assignment), the others in none
        if (k == 0) C.state_cr = keeps_field(k, R.phase) ? rec.cr : -1.0f;
        {
            DevProbe *have = nullptr;
            for (int i = 0; i < C.n_probes; i++) if (C.probes[i].cr == rec.cr) have = &C.probes[i];
            if (have) { if (!have->complete && rec.complete) *have = rec; }      // (a partial record gives way to the whole one)
            else if (C.n_probes < kMaxProbes) C.probes[C.n_probes++] = rec;
        }
        R.want = 0;
        if (R.phase == 5) { R.last = rec; R.phase = 6; set_active(0); return; }
        feed(rec);
    }
    // walk through the probes already on record (the other search, or the first encode, usually made them).  The
    // final probe of search 0 (:590) must leave its decode in the engine - the residual layer is computed from it -
    // so it is made again unless the engine's last decode of the chunk was at exactly that rate.
    while (!R.want && R.phase < 4) {
        float cr;
        if (!rs_next(R, cr)) break;
        const DevProbe *rec = find_probe(C, cr);
        // ... and the final probe of either search is the search's result (size, count, error sum): a record that may be
        // partial does not do
        const bool needs_state = (k == 0 && R.phase == 3 && C.state_cr != cr) || (R.phase == 3 && rec && !rec->complete);
        if (rec && !needs_state) feed(*rec);
        else { R.want = 1; R.want_cr = cr; }
    }
    if (!R.want && R.phase == 4) {
        // a search that leaves through the rate > 1000 exit (:571-574) makes no final probe: its result is the last
        // doubling step; for search 0 that step's decode has to be in the engine
        if (R.last.cr != R.result) { const DevProbe *rec = find_probe(C, R.result); if (rec) R.last = *rec; }
        if (k == 0 && C.state_cr != R.result) { R.want = 1; R.want_cr = R.result; R.phase = 5; }
        else R.phase = 6;
    }
    set_active(R.want);
    if (R.want) {
        // the final probe of a search (:590: its size, error sum and count are the search's result) and the probe that
        // restores a decode are exact; every other probe only has to tell feasible from infeasible
        // (so is a doubling step beyond rate 1000: if it is feasible the search ends on it, :571-574, without a final probe)
        const bool exact = R.phase == 3 || R.phase == 5 || (R.phase == 1 && R.want_cr > 1000.0f) || tiles > 1 || limit_qt <= 0.0;
        const unsigned int limit = exact ? 0u : infeasible_from(limit_qt < R.qt ? limit_qt : R.qt, n_pix);
        for (int t = 0; t < tiles; t++) { jf[t0 + t].cr = R.want_cr; jf[t0 + t].keep = keeps_field(k, R.phase); jf[t0 + t].bad_limit = limit; }
        atomicAdd(unfinished, 1);
    }
}

// ---- :765-796 ------------------------------------------------------------------------------------------
__global__ void k_trunc_advance(DevChunk *chunks, FrameState *fs, unsigned long long *trunc_bits, int *active, int n_chunks,
                                double n_pix, int *unfinished)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_chunks) return;
    DevChunk &C = chunks[f];
    if (!C.trunc_active) { active[f] = 0; return; }
    if (C.trunc_pending) {
        const double tb = (double) trunc_bits[f];
        const float cur = __uint_as_float(fs[f].maxerr_bits);
        if (cur > C.target) C.t_lo = tb;
        else {
            C.t_hi = tb;
            if (cur >= C.best_err) { C.best_err = cur; C.t_best = tb; C.mean_err = fs[f].err_sum / n_pix; }
        }
        C.trunc_pending = 0;
    }
    const double eps = 1e-8;
    if (((C.target - C.best_err) / C.target > eps) && (C.t_hi - C.t_lo > 8 * 4)) {
        trunc_bits[f] = (unsigned long long) ceil((C.t_hi + C.t_lo) / 2 / 8) * 8ull;
        fs[f].exit_above = C.target;                                  // (cur > target is all that is asked of an infeasible cut, above)
        C.trunc_pending = 1;
        active[f] = 1;
        atomicAdd(unfinished, 1);
    } else {
        C.trunc_active = 0;
        active[f] = 0;
    }
}

}  // namespace

void launch_search_advance(DevChunk *chunks, J2kFrame *jf, int *d_active, int n_chunks, int tiles, int k, double n_pix,
                           int *unfinished, hipStream_t s, float *cand_cr, int *cand_sel, double limit_qt)
{
    hipLaunchKernelGGL(k_search_advance, dim3(ceil_div(n_chunks, 64)), dim3(64), 0, s, chunks, jf, d_active, n_chunks, tiles, k, n_pix,
                       unfinished, cand_cr, cand_sel, limit_qt);
    EBCC_HIP_LAUNCH_CHECK();
}

void launch_trunc_advance(DevChunk *chunks, FrameState *fs, unsigned long long *trunc_bits, int *d_active, int n_chunks,
                          double n_pix, int *unfinished, hipStream_t s)
{
    hipLaunchKernelGGL(k_trunc_advance, dim3(ceil_div(n_chunks, 64)), dim3(64), 0, s, chunks, fs, trunc_bits, d_active, n_chunks, n_pix,
                       unfinished);
    EBCC_HIP_LAUNCH_CHECK();
}

// ---- the same with look-ahead (cut slots, residual.hpp): a round probes the bisection tree `levels` deep below the current
//      interval - node 0 the cut :779 would choose now, node 2 i + 1 the cut it would choose next if node i turns out
//      feasible (t_hi = cut), node 2 i + 2 if not (t_lo = cut) - and this kernel then walks the tree with the real
//      outcomes, statement for statement what `levels` iterations of :777-795 do: same cuts consumed in the same order,
//      same t_lo / t_hi / best error / mean error; the probes off the path are simply not looked at.
__global__ void k_trunc_advance_multi(DevChunk *chunks, const FrameState *fs, CutSlots cs, int n_chunks, double n_pix, int levels,
                                      const int *rank, int *unfinished)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_chunks) return;
    const int r = rank ? rank[f] : f;
    if (r < 0) return;                                                   // (not in this launch: not a word of it is touched)
    const int K = (1 << levels) - 1, v0 = r * K;
    DevChunk &C = chunks[f];
    if (!C.trunc_active) { for (int i = 0; i < K; i++) cs.active[v0 + i] = 0; return; }
    const double eps = 1e-8;
    auto goes_on = [&]() { return ((C.target - C.best_err) / C.target > eps) && (C.t_hi - C.t_lo > 8 * 4); };   // :777
    if (C.trunc_pending) {
        int node = 0;
        for (int depth = 0; depth < levels && goes_on(); depth++) {
            // (cs.bits[v0 + node] is the cut the loop chooses at this point: the proposals below follow its arithmetic)
            const double tb = (double) cs.bits[v0 + node];
            const float cur = __uint_as_float(cs.fs[v0 + node].maxerr_bits);
            if (cur > C.target) { C.t_lo = tb; node = 2 * node + 2; }
            else {
                C.t_hi = tb;
                if (cur >= C.best_err) { C.best_err = cur; C.t_best = tb; C.mean_err = cs.fs[v0 + node].err_sum / n_pix; }
                node = 2 * node + 1;
            }
        }
        C.trunc_pending = 0;
    }
    if (!goes_on()) {
        C.trunc_active = 0;
        for (int i = 0; i < K; i++) cs.active[v0 + i] = 0;
        return;
    }
    double lo[7], hi[7];
    bool on[7];
    lo[0] = C.t_lo; hi[0] = C.t_hi;
    for (int i = 0; i < K; i++) {
        on[i] = (i == 0 || on[(i - 1) >> 1]) && (hi[i] - lo[i] > 8 * 4);
        cs.active[v0 + i] = on[i] ? 1 : 0;
        if (!on[i]) continue;
        const unsigned long long cut = (unsigned long long) ceil((hi[i] + lo[i]) / 2 / 8) * 8ull;     // :779
        cs.bits[v0 + i] = cut;
        cs.frame_of[v0 + i] = f;
        cs.fs[v0 + i] = fs[f];                                           // (what a probe reads of the frame's state)
        cs.fs[v0 + i].exit_above = C.target;                             // (cur > target is all that is asked of an infeasible cut)
        if (2 * i + 2 < K) { lo[2 * i + 1] = lo[i]; hi[2 * i + 1] = (double) cut; lo[2 * i + 2] = (double) cut; hi[2 * i + 2] = hi[i]; }
    }
    C.trunc_pending = 1;
    atomicAdd(unfinished, 1);
}

void launch_trunc_advance_multi(DevChunk *chunks, const FrameState *fs, const CutSlots &cs, int n_chunks, double n_pix, int levels,
                                const int *rank, int *unfinished, hipStream_t s)
{
    hipLaunchKernelGGL(k_trunc_advance_multi, dim3(ceil_div(n_chunks, 64)), dim3(64), 0, s, chunks, fs, cs, n_chunks, n_pix, levels, rank, unfinished);
    EBCC_HIP_LAUNCH_CHECK();
}

}  // namespace ebcc
