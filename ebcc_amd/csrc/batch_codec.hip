// batch_codec.hip - one device batch of the frame codec (host.hpp): /root/reference/src/ebcc_codec.c:607-918 as
// encode_batch, :1215-1320 as decode_batch / decode_tiled, with the rate search (:545-596, twice per frame) and the
// truncation bisection (:765-796) either as host loops or as device-side state machines (search.hpp).  The host only
// steers: per-frame scalars come back from the device after every search, every per-sample operation runs in the kernels.
#include "host.hpp"

namespace ebcc {
namespace {

// ================================================================================================
// rate search of src/ebcc_codec.c:545-596 as a resumable state machine (one probe per step)
// ================================================================================================
struct RateSearch {
    float lo = 0, hi = 0, cr = 0, result = 0;
    double q = 0, q0 = 0, qt = 0;
    int phase = 4;           // 0 halving, 1 doubling, 2 bisect, 3 final probe, 4 done
    float pending = 0;
    void start(float cr0, double q_init, double q_target)
    {
        lo = hi = cr = cr0; q = q0 = q_init; qt = q_target; phase = 0;
    }
    bool done() const { return phase == 4; }
    // returns true and sets `out` if a probe at rate `out` is needed next
    bool next(float &out)
    {
        for (;;) {
            if (phase == 0) {
                if (q < qt && lo >= 1. / 2) { lo /= 2; out = pending = lo; return true; }      // :559-563
                q = q0; phase = 1;
            } else if (phase == 1) {
                if (q >= qt && hi <= 1000) { hi *= 2; out = pending = hi; return true; }       // :565-569
                if (q >= qt) { result = hi; phase = 4; return false; }                         // :571-574
                q = q0; phase = 2;
            } else if (phase == 2) {
                const double eps = 1e-8;
                if ((std::fabs(q - qt) > eps || q == 1.0) && hi - lo > 1.) {                   // :579-588
                    cr = (lo + hi) / 2; out = pending = cr; return true;
                }
                phase = 3; out = pending = lo; return true;                                    // :590
            } else {
                return false;
            }
        }
    }
    void feed(double quantile)
    {
        q = quantile;
        if (phase == 2) { if (q < qt) hi = cr; else lo = cr; }
        else if (phase == 3) { result = lo; phase = 4; }
    }
};


struct ProbeRec { float cr = -1; unsigned long long nbad = 0; int stream_bytes = 0; double err_sum = 0; bool complete = true; };

struct Job {                     // host-side state of one frame being encoded
    bool const_field = false;
    float minv = 0, maxv = 0, target = -1, cr = -1;
    double mean_err = 0, q = 0, q_first = 0;
    float rmin = 0, rmax = 0;
    bool skip = true, need_pure = false;
    float best_err = -1;
    size_t coeffs_orig = 0, coeffs_size = 0, len1 = 0;
    double t_hi = 0, t_lo = 0, t_best = 0;
    bool trunc_active = false;
    std::vector<uint8_t> tail, zbytes;
    // rate searches: [0] error-bounded (:728), [1] pure base layer (:836).  A probe's outcome depends only on
    // (frame, rate), so both searches share one record of the probes made so far.
    RateSearch rs[2];
    bool want[2] = {false, false};    // search k waits for the probe at want_cr[k]
    float want_cr[2] = {0, 0};
    ProbeRec last[2];                 // the final probe of search k (phase 3)
    std::vector<ProbeRec> probes;
    const ProbeRec *find_probe(float cr) const
    {
        for (const ProbeRec &r : probes) if (r.cr == cr) return &r;
        return nullptr;
    }
};

// The base layer of a batch of chunks.  A chunk is one frame, or `tiles` frames stacked along the row axis that
// the reference codes as ONE JPEG 2000 image with one tile per frame (src/ebcc_codec.c:105-180, n_tiles > 1).
// Every tile is a frame of the engine `ctx` (tile t of chunk c at index c * tiles + t); the arrays below are per
// CHUNK - rate, target, error statistics, codestream size - and are expanded to / gathered from the tiles here:
//   * all tiles of a chunk are probed at the same rate; a tile's byte budget subtracts its share of the main
//     header only (opj_j2k_update_rates: 135 / tiles, J2kFrame::hdr_share);
//   * nbad / err_sum add up; the codestream is main header (SIZ rewritten for the stacked image) + the tile-parts
//     (SOT with the tile index) + EOC, byte-identical to OpenJPEG's opj_write_tile sequence.
// The residual layer works on whole chunks in the engine `rc` (== ctx for one-frame chunks).
struct Batch {
    ebcc_hip_ctx *ctx;
    J2kBuffers &jb;
    const float *d_frames;
    size_t n, tiles, nt;           // chunks, tiles per chunk, n * tiles
    std::vector<J2kFrame> jf;      // per chunk
    J2kFrame *tjf;                 // per tile (device image; pinned: ctx->h_jf)
    std::vector<int> active;       // per chunk
    int *tactive, *ractive;        // pinned: ctx->h_act
    std::vector<float> state_cr;   // rate of the decode the engine holds for every chunk (-1: none)
    int *d_active;                 // per tile, on ctx
    hipStream_t s;
    ebcc_hip_ctx *rc;              // residual engine (chunk-sized frames)
    hipStream_t rs;
    Batch(ebcc_hip_ctx *c, const float *d, size_t n_, size_t tiles_ = 1, ebcc_hip_ctx *rc_ = nullptr)
        : ctx(c), jb(*static_cast<J2kBuffers *>(c->j2k)), d_frames(d), n(n_), tiles(tiles_), nt(n_ * tiles_), jf(n_),
          tjf(static_cast<J2kFrame *>(c->h_jf)), active(n_, 0), tactive(c->h_act), ractive(c->h_act + c->max_frames), state_cr(n_, -1.f),
          d_active(c->d_active), s(c->stream), rc(rc_ ? rc_ : c), rs((rc_ ? rc_ : c)->stream)
    {
        memset(tjf, 0, sizeof(J2kFrame) * nt);
    }
    void fetch_jf(hipStream_t on = nullptr)
    {
        if (!on) on = s;
        EBCC_HIP_CHECK(hipMemcpyAsync(tjf, jb.jf, sizeof(J2kFrame) * nt, hipMemcpyDeviceToHost, on));
        wait_stream(on);
        for (size_t c = 0; c < n; c++) {
            J2kFrame &o = jf[c];
            o.nbad = 0; o.err_sum = 0; o.overflow = 0; o.body_bytes = 0;
            for (size_t t = c * tiles; t < (c + 1) * tiles; t++) {
                o.nbad += tjf[t].nbad; o.err_sum += tjf[t].err_sum; o.overflow |= tjf[t].overflow; o.body_bytes += tjf[t].body_bytes;
            }
            o.stream_bytes = kJ2kMainHeaderBytes + (int) tiles * 14 + o.body_bytes + 2;     // main header, SOT + SOD per tile, EOC
        }
    }
    void push_jf()
    {
        for (size_t c = 0; c < n; c++)
            for (size_t t = c * tiles; t < (c + 1) * tiles; t++) {
                tjf[t].cr = jf[c].cr; tjf[t].target = jf[c].target;
                tjf[t].hdr_share = tiles > 1 ? (float) kJ2kMainHeaderBytes / (float) tiles : 0.0f;
            }
        EBCC_HIP_CHECK(hipMemcpyAsync(jb.jf, tjf, sizeof(J2kFrame) * nt, hipMemcpyHostToDevice, s));
    }
    void push_active()
    {
        for (size_t c = 0; c < n; c++)
            for (size_t t = c * tiles; t < (c + 1) * tiles; t++) tactive[t] = active[c];
        EBCC_HIP_CHECK(hipMemcpyAsync(d_active, tactive, sizeof(int) * nt, hipMemcpyHostToDevice, s));
    }
    // the chunk mask for the residual engine's kernels
    void push_ractive()
    {
        for (size_t c = 0; c < n; c++) ractive[c] = active[c];
        EBCC_HIP_CHECK(hipMemcpyAsync(rc->d_active, ractive, sizeof(int) * n, hipMemcpyHostToDevice, rs));
    }
    // one probe of the base layer for the active chunks: rate allocation at jf[c].cr (+ decode and statistics)
    // keep_field = false: only the statistics are wanted, jb.DEC stays what it was
    void launch_probe(bool decode, bool keep_field = true)
    {
        push_jf();
        push_active();
        launch_j2k_rate(jb, (int) nt, d_active, s);
        if (decode) launch_j2k_probe_decode(d_frames, jb, (int) nt, d_active, s, keep_field);
    }
    void probe(bool decode, bool keep_field = true) { launch_probe(decode, keep_field); fetch_jf(); }
    // codestream of the current layer assignment of the active chunks -> jobs[c].tail
    template <class Jobs>
    void collect_tails(Jobs &jobs)
    {
        push_active();
        launch_j2k_write(jb, (int) nt, d_active, s);
        fetch_jf();
        if (tiles == 1) {
            // all codestreams of the batch in one packed download (engine.hip: stage_download)
            std::vector<size_t> len(n, 0), off(n, 0);
            for (size_t f = 0; f < n; f++) if (active[f]) len[f] = (size_t) jf[f].stream_bytes;
            stage_download(ctx, jb.stream, jb.stream_cap, len.data(), off.data(), n, s);
            for (size_t f = 0; f < n; f++)
                if (active[f]) jobs[f].tail.assign(ctx->h_stage + off[f], ctx->h_stage + off[f] + len[f]);
            return;
        }
        // every tile was written as a one-tile codestream into its slot: [main header 135][SOT 12][SOD 2][packets][EOC 2]
        for (size_t c = 0; c < n; c++) {
            if (!active[c]) continue;
            std::vector<uint8_t> &o = jobs[c].tail;
            o.resize((size_t) jf[c].stream_bytes);
            size_t at = kJ2kMainHeaderBytes;
            for (size_t k = 0; k < tiles; k++) {
                const size_t t = c * tiles + k, part = 14 + (size_t) tjf[t].body_bytes;
                const uint8_t *slot = jb.stream + t * jb.stream_cap;
                if (k == 0) EBCC_HIP_CHECK(hipMemcpyAsync(o.data(), slot, kJ2kMainHeaderBytes, hipMemcpyDeviceToHost, s));
                EBCC_HIP_CHECK(hipMemcpyAsync(o.data() + at, slot + kJ2kMainHeaderBytes, part, hipMemcpyDeviceToHost, s));
                at += part;
            }
        }
        wait_stream(s);
        const unsigned H = (unsigned) jb.geom.H;
        for (size_t c = 0; c < n; c++) {
            if (!active[c]) continue;
            std::vector<uint8_t> &o = jobs[c].tail;
            auto put32 = [&](size_t at, unsigned v) { o[at] = (uint8_t) (v >> 24); o[at + 1] = (uint8_t) (v >> 16); o[at + 2] = (uint8_t) (v >> 8); o[at + 3] = (uint8_t) v; };
            put32(12, H * (unsigned) tiles);                             // SIZ: Ysiz (tile size XTsiz/YTsiz stays W x H)
            size_t at = kJ2kMainHeaderBytes;
            for (size_t k = 0; k < tiles; k++) {
                o[at + 4] = (uint8_t) (k >> 8); o[at + 5] = (uint8_t) k;  // SOT: Isot
                at += 14 + (size_t) tjf[c * tiles + k].body_bytes;
            }
            o[at] = 0xFF; o[at + 1] = 0xD9;                               // EOC
        }
    }
};

// Drive rate search k (0: error-bounded :728, 1: pure base layer :836) of every frame to completion; every
// round runs at most one probe per frame.  A search first advances through the probes already on record for its
// frame (the other search, or the first encode, usually made them) and only asks the GPU for rates not seen yet.
// The final probe of search 0 (:590) must leave its decode in the engine - the residual layer is computed from
// it - so it is re-run unless the engine's last decode of the frame was at exactly that rate.
template <class Jobs>
void run_search(Batch &b, int k, Jobs &jobs, size_t n_pix)
{
    const size_t n = b.n;
    auto needs_state = [&](const Job &j, size_t f, float cr) { return k == 0 && j.rs[0].phase == 3 && b.state_cr[f] != cr; };
    auto feed = [&](Job &j, const ProbeRec &rec) {
        RateSearch &rs = j.rs[k];
        if (rs.phase == 3) j.last[k] = rec;
        const double q = 1. - ((double) rec.nbad / (double) n_pix);                            // :512
        if (k == 0) j.q = q;
        rs.feed(q);
    };
    for (;;) {
        bool any = false;
        for (size_t f = 0; f < n; f++) {
            Job &j = jobs[f];
            b.active[f] = 0;
            if (j.const_field) continue;
            while (!j.want[k] && !j.rs[k].done()) {
                float cr;
                if (!j.rs[k].next(cr)) break;
                const ProbeRec *rec = j.find_probe(cr);
                if (rec && !needs_state(j, f, cr)) feed(j, *rec);
                else { j.want[k] = true; j.want_cr[k] = cr; }
            }
            if (j.want[k]) { b.active[f] = 1; b.jf[f].cr = j.want_cr[k]; any = true; }
        }
        if (!any) break;
        b.probe(true, k == 0);                                   // (search 1 uses the statistics only: the field of search 0 stays)
        for (size_t f = 0; f < n; f++) {
            if (!b.active[f]) continue;
            Job &j = jobs[f];
            const J2kFrame &r = b.jf[f];
            if (k == 0) b.state_cr[f] = r.cr;
            if (!j.find_probe(r.cr)) j.probes.push_back(ProbeRec{r.cr, r.nbad, r.stream_bytes, r.err_sum});
            log_trace("frame %zu (search %d): cr %f 1-quantile %.1e jp2_length %d", f, k, r.cr, (double) r.nbad / (double) n_pix,
                      r.stream_bytes);
            feed(j, *j.find_probe(r.cr));
            j.want[k] = false;
        }
    }
    // A search that leaves through the rate > 1000 exit (:571-574) makes no final probe: its result is the last
    // doubling step.  Take that probe's record, and for search 0 make sure its decode is in the engine.
    bool redo = false;
    for (size_t f = 0; f < n; f++) {
        Job &j = jobs[f];
        b.active[f] = 0;
        if (j.const_field) continue;
        if (j.last[k].cr != j.rs[k].result)
            if (const ProbeRec *rec = j.find_probe(j.rs[k].result)) j.last[k] = *rec;
        if (k == 0 && b.state_cr[f] != j.rs[0].result) { b.active[f] = 1; b.jf[f].cr = j.rs[0].result; redo = true; }
    }
    if (redo) {
        b.probe(true);
        for (size_t f = 0; f < n; f++)
            if (b.active[f]) {
                const J2kFrame &r = b.jf[f];
                b.state_cr[f] = r.cr;
                jobs[f].last[0] = ProbeRec{r.cr, r.nbad, r.stream_bytes, r.err_sum};
            }
    }
}

// The same search with its state machine on the device (search.hpp): the rounds are enqueued back to back - advance,
// rate allocation, probe decode - without a host synchronisation in between; the host looks at the states once after
// `rounds` of them (EBCC_HIP_SEARCH_ROUNDS, default 16: more than the usual search needs) and only enqueues more if a
// chunk is still searching.  Same probes, same decisions, same result as run_search (EBCC_HIP_HOST_SEARCH=1 selects that).
int search_rounds()
{
    if (const char *e = getenv("EBCC_HIP_SEARCH_ROUNDS")) return std::max(1, atoi(e));
    return 16;
}
constexpr int kSearchAll = 0, kSearchStart = 1, kSearchFinish = 2;
template <class Jobs>
void device_rate_search(Batch &b, int k, Jobs &jobs, size_t n_pix, unsigned slices, int lane = 0, int part = kSearchAll)
{
    // lane 1: the search runs on the engine's second stream with its own state, counters and active mask, beside whatever
    // the first stream does (search #2 beside the residual layer).  part: enqueue the first batch of rounds only
    // (kSearchStart: no host synchronisation), or take the search up from there (kSearchFinish), or both.
    ebcc_hip_ctx *ctx = b.ctx;
    const size_t n = b.n;
    DevChunk *h = static_cast<DevChunk *>(ctx->h_search) + (size_t) lane * ctx->max_frames, *d = static_cast<DevChunk *>(ctx->d_search) + (size_t) lane * ctx->max_frames;
    int *const d_counter = ctx->d_counter + 4 * lane, *const h_counter = ctx->h_counter + 4 * lane;
    int *const d_active = lane ? ctx->d_active + ctx->max_frames : b.d_active;
    hipStream_t s = lane ? second_stream(ctx) : b.s;
    J2kBuffers &jb = b.jb;
    const int forced = getenv("EBCC_HIP_SPECULATION") ? atoi(getenv("EBCC_HIP_SPECULATION")) != 0 : -1;
    const bool speculate = lane == 0 && (forced >= 0 ? forced != 0 : slices <= 1);   // (lane 1 runs on the stream the candidates would use)
    hipStream_t s2 = nullptr;
    // probes that only steer the search stop counting once they are certainly infeasible (search.hpp); EBCC_HIP_NO_SHORTCUTS=1
    // and TRACE logging (which prints every probe's count) keep every probe exact
    double jobs_qt0 = 0.0;                                               // the error-bounded search's quantile target (0: it never ran)
    for (size_t f = 0; f < n; f++) if (!jobs[f].const_field) { jobs_qt0 = jobs[f].rs[0].qt; break; }
    const double limit_qt = getenv("EBCC_HIP_NO_SHORTCUTS") || g_log_level <= 0 ? 0.0 : std::min(jobs_qt0, 1.0);
    auto advance = [&]() {
        launch_search_advance(d, jb.jf, d_active, (int) n, (int) b.tiles, k, (double) n_pix, d_counter, s,
                              speculate ? jb.cand_cr : nullptr, speculate ? jb.cand_sel : nullptr, limit_qt);
        if (speculate) launch_j2k_rate_publish(jb, (int) b.nt, s);
    };
    auto enqueue_rounds = [&](int rounds) {
        for (int r = 0; r < rounds; r++) {
            launch_j2k_rate(jb, (int) b.nt, d_active, s, speculate ? jb.have_rate : nullptr);
            if (speculate) {
                // the candidates read the record of bisection steps this k_rate may have extended, and the masks / rates of
                // the advance: after both; the next advance reads their results: after them
                EBCC_HIP_CHECK(hipEventRecord(ctx->ev_a, s));
                EBCC_HIP_CHECK(hipStreamWaitEvent(s2, ctx->ev_a, 0));
                launch_j2k_rate_candidates(jb, (int) b.nt, d_active, s2);
                EBCC_HIP_CHECK(hipEventRecord(ctx->ev_b, s2));
            }
            launch_j2k_probe_decode(b.d_frames, jb, (int) b.nt, d_active, s, k == 0 ? 2 : 0);     // (the field is stored where the advance asked for it: search.hip keeps_field)
            if (speculate) EBCC_HIP_CHECK(hipStreamWaitEvent(s, ctx->ev_b, 0));
            advance();
        }
    };
    if (part != kSearchFinish) {
    for (size_t f = 0; f < n; f++) {
        const Job &j = jobs[f];
        DevChunk &c = h[f];
        c.const_field = j.const_field ? 1 : 0;
        c.state_cr = b.state_cr[f];
        c.q = j.q;
        c.n_probes = (int) std::min<size_t>(j.probes.size(), kMaxProbes);
        for (int i = 0; i < c.n_probes; i++) c.probes[i] = DevProbe{j.probes[i].cr, j.probes[i].stream_bytes, j.probes[i].nbad, j.probes[i].err_sum, j.probes[i].complete ? 1 : 0, 0};
        const RateSearch &r = j.rs[k];
        DevRateSearch &o = c.rs[k];
        o.lo = r.lo; o.hi = r.hi; o.cr = r.cr; o.result = r.result; o.pending = r.pending; o.phase = j.const_field ? 6 : r.phase;
        o.q = r.q; o.q0 = r.q0; o.qt = r.qt; o.want = 0; o.want_cr = 0;
        o.last = DevProbe{j.last[k].cr, j.last[k].stream_bytes, j.last[k].nbad, j.last[k].err_sum, 1, 0};
    }
    EBCC_HIP_CHECK(hipMemcpyAsync(d, h, sizeof(DevChunk) * n, hipMemcpyHostToDevice, s));
    EBCC_HIP_CHECK(hipMemsetAsync(d_counter, 0, sizeof(int) * 4, s));
    // a round = the probe the previous advance asked for (rate allocation + decode of the active chunks), then the advance
    // that takes it in and asks for the next one.  Speculative rate allocation: a
    // step of the search can go two ways, so the layers of both rates it may ask for next are worked out on the engine's
    // second stream while the first stream decodes the current probe; the advance then takes the matching one over
    // (k_rate_publish) and the round's own k_rate only runs for the frames whose rate was not among the guesses.
    // It shortens a slice's chain (search #1 of 256 frames in one slice: 33 -> 29 ms) at the price of two more k_rate per
    // round; with several slices in flight the chip has no idle issue slots left to pay with (four slices: encode 7.7 GB/s
    // without, 6.7 with) - so it is on for a batch that runs as one slice, off otherwise; EBCC_HIP_SPECULATION=1 / 0 forces it.
    if (speculate) {
        s2 = second_stream(ctx);
        if (!ctx->ev_a) {
            EBCC_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_a, hipEventDisableTiming));
            EBCC_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_b, hipEventDisableTiming));
        }
        EBCC_HIP_CHECK(hipMemsetAsync(jb.cand_cr, 0xFF, sizeof(float) * 2 * b.nt, s));       // (NaN: no candidate matches)
        EBCC_HIP_CHECK(hipMemsetAsync(jb.have_rate, 0, sizeof(int) * b.nt, s));
    }
    advance();
    enqueue_rounds(search_rounds());
    }
    if (part == kSearchStart) return;
    if (speculate && !s2) s2 = second_stream(ctx);
    for (;;) {
        EBCC_HIP_CHECK(hipMemcpyAsync(h, d, sizeof(DevChunk) * n, hipMemcpyDeviceToHost, s));
        EBCC_HIP_CHECK(hipMemcpyAsync(h_counter, d_counter, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
        wait_stream(s);
        bool done = true;
        for (size_t f = 0; f < n; f++) done &= h[f].rs[k].phase == 6;
        if (done) break;
        enqueue_rounds(6);
    }
    log_trace("rate search %d: %d probes of chunks over the rounds", k, h_counter[0]);
    if (getenv("EBCC_HIP_PHASE_TIMING")) {
        int most = 0; long long sum = 0;
        for (size_t f = 0; f < n; f++) { most = std::max(most, h[f].n_probes); sum += h[f].n_probes; }
        fprintf(stderr, "ebcc-mi355x rate search %d: %d probes over the rounds, probes on record per chunk: mean %.1f, most %d\n", k, h_counter[0], (double) sum / (double) n, most);
    }
    if (getenv("EBCC_HIP_T1_STATS") && slices <= 1) j2k_probe_hist_dump(k == 0 ? "search #1" : "search #2");
    for (size_t f = 0; f < n; f++) {
        Job &j = jobs[f];
        if (j.const_field) continue;
        const DevChunk &c = h[f];
        const DevRateSearch &o = c.rs[k];
        RateSearch &r = j.rs[k];
        r.lo = o.lo; r.hi = o.hi; r.cr = o.cr; r.result = o.result; r.pending = o.pending; r.phase = 4; r.q = o.q; r.q0 = o.q0; r.qt = o.qt;
        j.last[k] = ProbeRec{o.last.cr, o.last.nbad, o.last.stream_bytes, o.last.err_sum};
        if (k == 0) j.q = c.q;
        j.probes.clear();
        for (int i = 0; i < c.n_probes; i++) j.probes.push_back(ProbeRec{c.probes[i].cr, c.probes[i].nbad, c.probes[i].stream_bytes, c.probes[i].err_sum, c.probes[i].complete != 0});
        b.state_cr[f] = c.state_cr;
    }
    b.fetch_jf(s);                                                        // (the host mirror of the per-frame scalars follows the device again)
}
template <class Jobs>
void rate_search(Batch &b, int k, Jobs &jobs, size_t n_pix, unsigned slices)
{
    const bool host_loop = getenv("EBCC_HIP_HOST_SEARCH") != nullptr;
    if (host_loop) run_search(b, k, jobs, n_pix); else device_rate_search(b, k, jobs, n_pix, slices);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// ebcc_encode for a batch of device-resident single-frame chunks.  Returns 0, 1 (error) or 2 (NaN/Inf).
// ------------------------------------------------------------------------------------------------
// `n` chunks of `tiles` frames each (tiles == 1: the frame-per-chunk case); `rctx`: residual engine for the stacked
// chunk image when tiles > 1.
int encode_batch(ebcc_hip_ctx *ctx, const float *d_frames, size_t n, const codec_config_t *cfg, uint8_t **outs, size_t *sizes,
                 SliceGate *next, size_t tiles, ebcc_hip_ctx *rctx, unsigned slices, PhaseNote *note)
{
    struct Release { SliceGate *g; ~Release() { if (g) g->release(); } } release_on_exit{next};   // (error paths too)
    struct NoteOnce { PhaseNote *n; void tell() { if (n) { n->slice_done(); n = nullptr; } } ~NoteOnce() { tell(); } } gpu_phase_over{note};   // (every path reports once)
    const EncodeEnv env;
    const double q_target = 1 - env.base_error_quantile;
    const int mode = (int) cfg->residual_compression_type;
    const bool searching = mode == MAX_ERROR || mode == RELATIVE_ERROR;
    const size_t n_pix = ctx->n_pix * tiles;                           // pixels of a chunk
    const size_t nt = n * tiles;
    Batch b(ctx, d_frames, n, tiles, rctx);
    J2kBuffers &jb = b.jb;
    hipStream_t s = b.s;
    ebcc_hip_ctx *rc = b.rc;                                           // residual engine and its stream (== ctx, s for one-frame chunks)
    hipStream_t rs = b.rs;
    std::vector<Job> jobs(n);
    PhaseTimer pt;

    // ---- statistics, scaling, transform, tier-1: once per frame
    launch_input_stats(d_frames, (int) nt, ctx->n_pix, ctx->rb.fs, s);
    if (tiles > 1) {
        // the reference scales the whole chunk with one (min, max) (:686-689): combine the tiles' statistics before
        // the transform reads them; a tile that happens to be constant inside a varying chunk is coded normally
        fetch_frame_states(ctx, nt);
        for (size_t c = 0; c < n; c++) {
            FrameState *t0 = ctx->h_fs + c * tiles;
            float mn = t0[0].minv, mx = t0[0].maxv;
            int bad = 0;
            for (size_t k = 0; k < tiles; k++) { mn = std::min(mn, t0[k].minv); mx = std::max(mx, t0[k].maxv); bad |= t0[k].has_nonfinite; }
            for (size_t k = 0; k < tiles; k++) { t0[k].minv = mn; t0[k].maxv = mx; t0[k].has_nonfinite = bad; t0[k].const_field = mn == mx; }
        }
        push_frame_states(ctx, nt);
    }
    launch_j2k_analysis(d_frames, jb, (int) nt, s);
    if (next) { next->release(); release_on_exit.g = nullptr; }       // the next slice may start: this one's first stage is queued
    fetch_frame_states(ctx, nt);
    b.fetch_jf();
    if (j2k_tier1_retry(jb, (int) nt, b.tjf, s)) b.fetch_jf();        // (a group's decisions outgrew the segmented encoder's buffer)
    for (size_t f = 0; f < n; f++) {
        const FrameState &t0 = ctx->h_fs[f * tiles];                   // (all tiles of a chunk carry the chunk's statistics)
        if (t0.has_nonfinite) { log_fatal("NaN or Inf found in data of frame %zu", f); return 2; }
        if (b.jf[f].overflow) { log_fatal("code-block byte slot overflow in frame %zu", f); return 1; }
        jobs[f].const_field = t0.const_field != 0;
        jobs[f].minv = t0.minv;
        jobs[f].maxv = t0.maxv;
        b.jf[f].cr = cfg->base_cr;
        b.jf[f].target = 0;
        b.active[f] = jobs[f].const_field ? 0 : 1;
    }
    if (rc != ctx) {                                                   // chunk-level frame states of the residual engine
        for (size_t f = 0; f < n; f++) {
            FrameState &r = rc->h_fs[f];
            r = FrameState{};
            r.minv = jobs[f].minv; r.maxv = jobs[f].maxv; r.const_field = jobs[f].const_field;
        }
        push_frame_states(rc, n);
    }
    pt.mark("analysis (dwt, tier-1, ckpt)");
    const bool need_decode = mode != NONE;
    if (need_decode)
        for (size_t f = 0; f < n; f++) {
            float target = cfg->error;                                                        // :723-726
            if (mode == RELATIVE_ERROR) target *= jobs[f].maxv - jobs[f].minv;
            jobs[f].target = target;
            b.jf[f].target = target;
        }
    // ---- first encode at base_cr (:693) and, unless NONE, its decode (:707-709)
    b.probe(need_decode);
    if (mode == NONE) {
        b.collect_tails(jobs);
    } else {
        for (size_t f = 0; f < n; f++) {
            if (jobs[f].const_field) continue;
            jobs[f].mean_err = b.jf[f].err_sum / (double) n_pix;                              // :709
            jobs[f].q = jobs[f].q_first = 1. - ((double) b.jf[f].nbad / (double) n_pix);
            jobs[f].cr = cfg->base_cr;
            jobs[f].probes.push_back(ProbeRec{b.jf[f].cr, b.jf[f].nbad, b.jf[f].stream_bytes, b.jf[f].err_sum});
            b.state_cr[f] = b.jf[f].cr;
        }
        // residual range of the first decode: only the header fields survive when no search runs (:716)
        launch_residual_minmax(d_frames, jb.DEC, (int) n, n_pix, rc->rb.fs, rs);
        fetch_frame_states(rc, n);
        for (size_t f = 0; f < n; f++) { jobs[f].rmin = rc->h_fs[f].rmin; jobs[f].rmax = rc->h_fs[f].rmax; }
        if (!searching) {                       // stale enum values fall through to a base-only stream (quirk Q2)
            for (size_t f = 0; f < n; f++) b.active[f] = jobs[f].const_field ? 0 : 1;
            b.collect_tails(jobs);
        }
    }

    pt.mark("first probe");
    if (searching) {
        // ---- rate search #1 (:728)
        const bool pure_done = q_target == 1.0;                                               // :738
        const bool want_pure = !pure_done && !env.no_fallback;
        for (size_t f = 0; f < n; f++)
            if (!jobs[f].const_field) jobs[f].rs[0].start(cfg->base_cr, jobs[f].q, q_target);
        rate_search(b, 0, jobs, n_pix, slices);
        for (size_t f = 0; f < n; f++) {
            b.active[f] = jobs[f].const_field ? 0 : 1;
            if (!jobs[f].const_field) { jobs[f].cr = jobs[f].rs[0].result; jobs[f].len1 = (size_t) jobs[f].last[0].stream_bytes; }
        }
        pt.mark("rate search 1");
        // base layer of search #1.  (Sending the codestreams off without waiting for them - written and packed on the second
        // search's stream, fetched at the assembly - was measured: the slice's next stages are queued 2 ms earlier and the step
        // gets 1 - 5 ms LONGER, three alternating runs on two boxes; the wait stays.)
        b.collect_tails(jobs);
        auto start_search2 = [&]() {
            for (size_t f = 0; f < n; f++) {
                if (jobs[f].const_field) continue;
                if (env.no_consistency) jobs[f].rs[1].start(jobs[f].cr, jobs[f].q, 1.0);      // from search #1's state
                else jobs[f].rs[1].start(cfg->base_cr, jobs[f].q_first, 1.0);                 // :829-833 == the first probe
            }
        };
        // ---- the pure base-layer search (:819-836) depends on nothing the residual layer produces: its rounds are queued on the
        //      engine's second stream now (own state, counters and mask: device_rate_search lane 1) and run beside the residual
        //      layer and the truncation search; it is taken up again where the reference runs it (below).  Round 2 measured
        //      this slower - the search was hidden behind the level-22 zstd of every prefix then; with the entropy stage cut
        //      down to the prefixes whose size can matter, the search was what the slice waited for, and its sizes are what
        //      decides which prefixes those are.  (With EBCC_HIP_HOST_SEARCH=1 it runs in the reference's place.)
        const bool overlap2 = want_pure && tiles == 1 && rc == ctx && !getenv("EBCC_HIP_HOST_SEARCH");
        struct DrainSecond {               // an error return between here and the take-up must not leave rounds in flight
            ebcc_hip_ctx *c; bool armed;
            ~DrainSecond() { if (armed && c->stream2) hipStreamSynchronize(c->stream2); }
        } drain2{ctx, false};
        if (overlap2) {
            start_search2();
            device_rate_search(b, 1, jobs, n_pix, slices, 1, kSearchStart);
            drain2.armed = true;
        }
        launch_residual_minmax(d_frames, jb.DEC, (int) n, n_pix, rc->rb.fs, rs);             // :730-733
        fetch_frame_states(rc, n);
        bool any_resid = false;
        for (size_t f = 0; f < n; f++) {
            Job &j = jobs[f];
            b.active[f] = 0;
            if (j.const_field) continue;
            j.rmin = rc->h_fs[f].rmin; j.rmax = rc->h_fs[f].rmax;
            float cur = fmaxf(fabsf(j.rmin), fabsf(j.rmax));                                  // :735
            j.skip = cur <= j.target;                                                         // :737
            if (!j.skip) { b.active[f] = 1; any_resid = true; }
        }
        pt.mark("tails + residual range");

        if (!zstd().ok) { log_fatal("libzstd not available"); return 1; }
        // ---- the entropy stage's state (the stage itself follows the truncation search; the frames whose search ends first
        //      - the early group, below - enter it while the others still search).
        //      jobs on the process-wide pool (HostPool): every slice of a batch feeds the same workers, so the host is never
        //      oversubscribed however many slices run
        std::vector<const uint8_t *> coeff_ptr(n, nullptr);             // the kept prefix of a frame in pinned host memory
        std::atomic<long long> zstd_us{0}, zstd_max_us{0}, zstd_bytes{0}, bound_us{0};    // core time, longest job, bytes
        enum : uint8_t { kZNone = 0, kZQueued = 1, kZRunning = 2, kZSkipped = 3 };
        std::unique_ptr<std::atomic<uint8_t>[]> zstate(new std::atomic<uint8_t>[n]);
        for (size_t f = 0; f < n; f++) zstate[f] = kZNone;
        std::vector<size_t> zfloor(n, 0);                               // lower bound of z (0: none)
        std::vector<char> floor_done(n, 0);
        using PoolBatches = std::vector<std::shared_ptr<HostPool::Batch>>;
        PoolBatches zbatches, fbatches;                                 // level-22 jobs; lower bounds
        struct WaitOnExit { PoolBatches &v; ~WaitOnExit() { for (auto &b : v) if (b) b->wait(); } } wait_on_exit{zbatches}, wait_on_exit_f{fbatches};   // (error paths too: the jobs point into this frame)
        // level-22 zstd of the frames in `list`, in the order given; a frame that was decided in the meantime (kZSkipped)
        // is passed over
        const bool trace_jobs = pt.on && getenv("EBCC_HIP_ZSTD_TRACE");
        auto submit_zstd = [&](std::vector<size_t> list) {
            if (trace_jobs) {
                static const auto epoch0 = std::chrono::steady_clock::now(); (void) epoch0;
                fprintf(stderr, "zstd-submit batch %p jobs %zu\n", (const void *) &jobs, list.size());
            }
            for (size_t f : list) zstate[f] = kZQueued;
            auto order = std::make_shared<std::vector<size_t>>(std::move(list));
            zbatches.push_back(HostPool::instance().submit(order->size(), entropy_threads(slices), [&, order](size_t i) {
                const size_t f = (*order)[i];
                uint8_t expect = kZQueued;
                if (!zstate[f].compare_exchange_strong(expect, kZRunning)) return;
                Job &j = jobs[f];
                const auto z0 = std::chrono::steady_clock::now();
                j.zbytes.resize(zstd().bound(j.coeffs_size));
                const size_t z = zstd().compress(j.zbytes.data(), j.zbytes.size(), coeff_ptr[f], j.coeffs_size, env.zstd_level);
                if ((zstd().is_error && zstd().is_error(z)) || z > j.zbytes.size()) throw std::runtime_error("ZSTD_compress failed on a residual prefix");
                j.zbytes.resize(z);
                const long long us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - z0).count();
                if (trace_jobs) {                                           // (EBCC_HIP_ZSTD_TRACE: when each job ran, on which CPU)
                    static const auto epoch = std::chrono::steady_clock::now();
                    const long long a = std::chrono::duration_cast<std::chrono::microseconds>(z0 - epoch).count();
                    fprintf(stderr, "zstd-job batch %p bytes %zu start %lld us end %lld us cpu %d\n", (const void *) &jobs, j.coeffs_size, a, a + us, sched_getcpu());
                }
                zstd_us += us; zstd_bytes += (long long) j.coeffs_size;
                long long m = zstd_max_us.load(); while (us > m && !zstd_max_us.compare_exchange_weak(m, us)) {}
            }));
        };
        // the lower bounds of z for the frames in `list` (zstd_size_lower_bound)
        auto submit_floors = [&](const std::vector<size_t> &frames_) {
            auto list = std::make_shared<std::vector<size_t>>(frames_);
            fbatches.push_back(HostPool::instance().submit(list->size(), entropy_threads(slices), [&, list](size_t i) {
                const size_t f = (*list)[i];
                const auto z0 = std::chrono::steady_clock::now();
                zfloor[f] = zstd_size_lower_bound(coeff_ptr[f], jobs[f].coeffs_size);
                floor_done[f] = 1;
                bound_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - z0).count();
            }));
        };
        // longest first: level 22 takes ~0.2 ms per KB on one core and a batch has frames whose prefix is ten times the
        // average - started last, such a frame alone decides when the slice can go on
        auto longest_first = [&](std::vector<size_t> &list) {
            std::stable_sort(list.begin(), list.end(), [&](size_t a, size_t c) { return jobs[a].coeffs_size > jobs[c].coeffs_size; });
        };
        long long wait_us = 0;
        auto join = [&](PoolBatches &v) -> bool {
            const auto w0 = std::chrono::steady_clock::now();
            bool ok = true;
            std::string why;
            for (auto &b : v) if (b && !b->wait()) { ok = false; if (why.empty()) why = b->error; }
            v.clear();
            wait_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - w0).count();
            if (!ok) { log_fatal("entropy stage failed: %s", why.c_str()); set_error("%s", why.c_str()); }
            return ok;
        };
        auto zjoin = [&]() -> bool { const bool a = join(fbatches), c = join(zbatches); return a && c; };
        const bool use_floor = want_pure && zstd_floor_usable() && !getenv("EBCC_HIP_NO_SHORTCUTS");
        if (any_resid) {
            // ---- residual layer: SPIHT with a budget of the base layer's size (:744-754)
            b.push_ractive();
            launch_pad_and_dc(d_frames, jb.DEC, rc->rb, (int) n, rc->d_active, rs);
            launch_analysis(rc->rb, (int) n, rc->d_active, rs);
            // (budgets, the encoder, the cut "everything" and its probe are queued without a look at the frame states in
            //  between: the budget follows from the base layer's size, the whole stream's length stays on the device)
            for (size_t f = 0; f < n; f++) rc->h_u64a[f] = (unsigned long long) jobs[f].len1 * 8 + 128;   // bits0 = trunc_bits + 128
            EBCC_HIP_CHECK(hipMemcpyAsync(rc->d_u64a, rc->h_u64a, n * sizeof(unsigned long long), hipMemcpyHostToDevice, rs));
            launch_residual_budget(rc->rb, (int) n, rc->d_u64a, rc->d_active, rs);
            launch_spiht_encode(rc->rb, (int) n, rc->d_u64a, rc->d_active, rs);
            launch_whole_stream_cut(rc->rb, (int) n, rc->d_u64b, rc->d_active, rs);
            launch_prefix_synthesis_stats(d_frames, jb.DEC, rc->rb, (int) n, rc->d_u64b, rc->d_active, rs);   // full decode, :749
            fetch_frame_states(rc, n);
            for (size_t f = 0; f < n; f++) {
                if (!b.active[f]) continue;
                jobs[f].coeffs_orig = rc->h_fs[f].stream_bytes;
                jobs[f].coeffs_size = jobs[f].coeffs_orig;
                rc->h_u64b[f] = (unsigned long long) jobs[f].coeffs_orig * 8;
            }
            auto probe_residual = [&]() {
                EBCC_HIP_CHECK(hipMemcpyAsync(rc->d_u64b, rc->h_u64b, n * sizeof(unsigned long long), hipMemcpyHostToDevice, rs));
                b.push_ractive();
                launch_prefix_synthesis_stats(d_frames, jb.DEC, rc->rb, (int) n, rc->d_u64b, rc->d_active, rs);
                fetch_frame_states(rc, n);
            };
            pt.mark("residual: analysis, SPIHT, whole-stream probe");
            for (size_t f = 0; f < n; f++) {
                Job &j = jobs[f];
                if (!b.active[f]) continue;
                float cur = u2f(rc->h_fs[f].maxerr_bits);                                    // :754
                if (cur > j.target) {                                                         // :755-759
                    log_info("frame %zu: could not reach error target %f (%f instead); retry with pure base compression", f, j.target, cur);
                    j.skip = true; j.need_pure = true;
                } else {
                    j.best_err = cur;
                    j.mean_err = rc->h_fs[f].err_sum / (double) n_pix;                       // :762
                    j.t_hi = (double) j.coeffs_size * 8; j.t_lo = 112.0; j.t_best = j.t_hi;   // :766-776
                    j.trunc_active = true;
                }
            }
            // ---- truncation bisection (:777-795): all frames advance one cut per round
            const bool host_loop = getenv("EBCC_HIP_HOST_SEARCH") != nullptr;
            if (!host_loop) {
                // state machine on the device (search.hpp): advance, reconstruct the decoder state at the cut, synthesis +
                // statistics - enqueued back to back, one look at the states after `rounds` of them
                DevChunk *h = static_cast<DevChunk *>(rc->h_search), *d = static_cast<DevChunk *>(rc->d_search);
                for (size_t f = 0; f < n; f++) {
                    const Job &j = jobs[f];
                    DevChunk &c = h[f];
                    c.t_hi = j.t_hi; c.t_lo = j.t_lo; c.t_best = j.t_best; c.mean_err = j.mean_err; c.best_err = j.best_err;
                    c.target = j.target; c.trunc_active = j.trunc_active ? 1 : 0; c.trunc_pending = 0;
                }
                EBCC_HIP_CHECK(hipMemcpyAsync(d, h, sizeof(DevChunk) * n, hipMemcpyHostToDevice, rs));
                EBCC_HIP_CHECK(hipMemsetAsync(rc->d_counter, 0, sizeof(int) * 4, rs));
                // rounds a chunk can still need: a cut halves the interval (rounded up to a byte: + 8 bits at most) until it is
                // 32 bits wide (:777), then one more advance sees that nothing is left
                auto rounds_left = [](const DevChunk &c) {
                    if (!c.trunc_active) return 0;
                    double w = c.t_hi - c.t_lo;
                    int r = 1;
                    while (w > 32 && r < 64) { w = w / 2 + 8; r++; }
                    return r;
                };
                const bool forced_rounds = getenv("EBCC_HIP_SEARCH_ROUNDS") != nullptr;
                int cuts_left = 1;                                       // cuts the longest search still visits, + the advance that ends it
                for (size_t f = 0; f < n; f++) cuts_left = std::max(cuts_left, rounds_left(h[f]));
                auto take_result = [&](const DevChunk &c, Job &j) {
                    j.t_hi = c.t_hi; j.t_lo = c.t_lo; j.t_best = c.t_best; j.mean_err = c.mean_err; j.best_err = c.best_err;
                    j.trunc_active = false;
                };
                // ---- Look-ahead (search.hpp: launch_trunc_advance_multi): a round probes the cut :779 chooses now and the cuts
                //      either outcome leads to - `levels` levels of the bisection tree, 2^levels - 1 cut slots per frame - so the
                //      search takes 1 / levels of the rounds.  The rounds are latency (a chain of six small launches beside the
                //      other slices' work), the probes off the path mostly stop early (a cut shorter than an infeasible one is
                //      infeasible too: its first wave over the target ends it).  EBCC_HIP_TRUNC_LEVELS=1: one cut per round.
                int levels = getenv("EBCC_HIP_TRUNC_LEVELS") ? std::min(3, std::max(1, atoi(getenv("EBCC_HIP_TRUNC_LEVELS")))) : 2;
                while (levels > 1 && !ensure_cut_slots(rc, (int) n * ((1 << levels) - 1))) levels--;
                if (levels > 1) {
                    const CutSlots &cs = rc->cut;
                    const int n_slots = (int) n * ((1 << levels) - 1);
                    launch_trunc_advance_multi(d, rc->rb.fs, cs, (int) n, (double) n_pix, levels, nullptr, rc->d_counter, rs);
                    int rounds = forced_rounds ? search_rounds() : (cuts_left + levels - 1) / levels + 1;
                    for (;;) {
                        for (int r = 0; r < rounds; r++) {
                            launch_prefix_synthesis_slots(d_frames, jb.DEC, rc->rb, cs, n_slots, rs);
                            launch_trunc_advance_multi(d, rc->rb.fs, cs, (int) n, (double) n_pix, levels, nullptr, rc->d_counter, rs);
                        }
                        EBCC_HIP_CHECK(hipMemcpyAsync(h, d, sizeof(DevChunk) * n, hipMemcpyDeviceToHost, rs));
                        wait_stream(rs);
                        bool done = true;
                        for (size_t f = 0; f < n; f++) done &= !h[f].trunc_active;
                        if (done) break;
                        rounds = 3;
                    }
                    for (size_t f = 0; f < n; f++) if (jobs[f].trunc_active) take_result(h[f], jobs[f]);
                } else {
                    launch_trunc_advance(d, rc->rb.fs, rc->d_u64b, rc->d_active, (int) n, (double) n_pix, rc->d_counter, rs);
                    int rounds = forced_rounds ? search_rounds() : cuts_left + 1;
                    for (;;) {
                        for (int r = 0; r < rounds; r++) {
                            launch_prefix_synthesis_stats(d_frames, jb.DEC, rc->rb, (int) n, rc->d_u64b, rc->d_active, rs);
                            launch_trunc_advance(d, rc->rb.fs, rc->d_u64b, rc->d_active, (int) n, (double) n_pix, rc->d_counter, rs);
                        }
                        EBCC_HIP_CHECK(hipMemcpyAsync(h, d, sizeof(DevChunk) * n, hipMemcpyDeviceToHost, rs));
                        wait_stream(rs);
                        bool done = true;
                        for (size_t f = 0; f < n; f++) done &= !h[f].trunc_active;
                        if (done) break;
                        rounds = 6;
                    }
                    for (size_t f = 0; f < n; f++) if (jobs[f].trunc_active) take_result(h[f], jobs[f]);
                }
            } else
            for (;;) {
                const double eps = 1e-8;
                bool any = false;
                for (size_t f = 0; f < n; f++) {
                    Job &j = jobs[f];
                    b.active[f] = 0;
                    if (!j.trunc_active) continue;
                    if (((j.target - j.best_err) / j.target > eps) && (j.t_hi - j.t_lo > 8 * 4)) {
                        size_t tb = ((size_t) ceill((long double) ((j.t_hi + j.t_lo) / 2 / 8))) * 8;
                        rc->h_u64b[f] = tb;
                        b.active[f] = 1;
                        any = true;
                    } else {
                        j.trunc_active = false;
                    }
                }
                if (!any) break;
                probe_residual();
                for (size_t f = 0; f < n; f++) {
                    Job &j = jobs[f];
                    if (!b.active[f]) continue;
                    const double tb = (double) rc->h_u64b[f];
                    float cur = u2f(rc->h_fs[f].maxerr_bits);
                    if (cur > j.target) j.t_lo = tb;
                    else {
                        j.t_hi = tb;
                        if (cur >= j.best_err) { j.best_err = cur; j.t_best = tb; j.mean_err = rc->h_fs[f].err_sum / (double) n_pix; }
                    }
                    log_trace("frame %zu: trunc_lo %.1f trunc_hi %.1f max error %f", f, j.t_lo, j.t_hi, cur);
                }
            }
            for (size_t f = 0; f < n; f++) {
                Job &j = jobs[f];
                if (j.const_field || j.skip) { if (j.need_pure) j.coeffs_size = j.coeffs_orig; else j.coeffs_size = 0; }
                else j.coeffs_size = (size_t) (j.t_best / 8.);                                // :796
            }
        }
        pt.mark("truncation search");
        // ---- entropy stage of the kept SPIHT prefix on host cores (:811-817) and the pure base-layer fallback (:819-854).
        //      Level-22 zstd is by far the longest host step (~160 ns per byte on one core: 1.3 core-seconds per 256 frames
        //      of the bench workload on a box whose container has 16 CPUs), and the reference throws most of it away: the
        //      compressed size z is compared with what the pure base-layer search gives (:838: len2 < z + len1), and for
        //      ~95 % of ERA5-like frames the base layer alone wins.  So z is only worked out where it can matter: a frame
        //      whose z is PROVABLY above len2 - len1 (zstd_size_lower_bound: the literals no match can cover cost at least
        //      their entropy) takes the pure base layer without being compressed - the same decision, bytes unchanged.
        // the kept SPIHT prefixes of the batch (but those of the early group, which left during the truncation search) in one
        // packed download; the workers read them where they land (the staging buffer of the residual engine is not touched
        // again before they are done)
        std::vector<size_t> coeff_len(n, 0), coeff_off(n, 0);
        for (size_t f = 0; f < n; f++) {
            Job &j = jobs[f];
            if (j.coeffs_size <= 16) j.coeffs_size = 0;
            if (!coeff_ptr[f]) coeff_len[f] = j.coeffs_size;
        }
        stage_download(rc, (const uint8_t *) rc->rb.stream, rc->rb.stream_words * sizeof(uint32_t), coeff_len.data(), coeff_off.data(), n, rs);
        for (size_t f = 0; f < n; f++) if (coeff_len[f]) coeff_ptr[f] = rc->h_stage + coeff_off[f];
        std::vector<size_t> with_prefix;
        for (size_t f = 0; f < n; f++) if (jobs[f].coeffs_size > 0) with_prefix.push_back(f);
        std::vector<size_t> cand;
        auto not_started = [&](std::vector<size_t> v) {                 // (the early group's jobs are on their way)
            v.erase(std::remove_if(v.begin(), v.end(), [&](size_t f) { return zstate[f] != kZNone; }), v.end());
            return v;
        };
        if (!want_pure) {
            std::vector<size_t> rest = not_started(with_prefix);
            longest_first(rest);
            submit_zstd(rest);                                          // no fallback: every prefix is part of its stream
        } else {
            // frames whose residual layer could not reach the target are coded by the base layer whatever z is (:838 need_pure)
            std::vector<size_t> now;
            for (size_t f : with_prefix) {
                if (jobs[f].need_pure) { zstate[f] = kZSkipped; continue; }
                if (use_floor && jobs[f].coeffs_size <= kZstdFloorMaxBytes) cand.push_back(f); else now.push_back(f);
            }
            now = not_started(now);
            longest_first(now);
            if (!now.empty()) submit_zstd(now);
            std::vector<size_t> want_floor;
            for (size_t f : cand) if (!floor_done[f]) want_floor.push_back(f);
            if (!want_floor.empty()) submit_floors(want_floor);
        }
        pt.mark("zstd: queued");
        if (want_pure) {
            // The pure-base-layer search restarts from base_cr with the quantile of a re-encode at base_cr
            // (:829-833), i.e. of the first probe above (unless that consistency step is disabled), and re-uses
            // every probe search #1 made; it runs here, while host cores work on the prefixes: first the floors (a few
            // microseconds per KB), then - the search still running on the GPU, len2 not known yet - zstd of the
            // candidates in the order in which they are likely to need it (lowest floor per byte first); the moment the
            // search's sizes are in, the candidates they decide are struck from the queue.
            const bool host_loop = getenv("EBCC_HIP_HOST_SEARCH") != nullptr;
            if (overlap2) {
                // the search has been running beside the residual layer: its sizes are (nearly) there, the floors take a
                // millisecond - the prefixes that are still open after that are compressed, longest first
                device_rate_search(b, 1, jobs, n_pix, slices, 1, kSearchFinish);              // :836
                drain2.armed = false;
                pt.mark("rate search 2");
                gpu_phase_over.tell();                                                        // (what follows is host work and one small launch)
                if (!join(fbatches)) return 1;                                                // (the floors)
            } else {
                start_search2();
                if (!host_loop) device_rate_search(b, 1, jobs, n_pix, slices, 0, kSearchStart);
                if (!cand.empty()) {
                    if (!join(fbatches)) return 1;                                            // (the floors)
                    std::vector<size_t> spec = not_started(cand);
                    std::stable_sort(spec.begin(), spec.end(), [&](size_t a, size_t c) {
                        return (double) zfloor[a] * (double) jobs[c].coeffs_size < (double) zfloor[c] * (double) jobs[a].coeffs_size; });
                    submit_zstd(spec);
                }
                if (host_loop) run_search(b, 1, jobs, n_pix); else device_rate_search(b, 1, jobs, n_pix, slices, 0, kSearchFinish);   // :836
                pt.mark("rate search 2");
                gpu_phase_over.tell();
            }
            long long skipped_bytes = 0, skipped = 0;
            for (size_t f : cand) {
                const Job &j = jobs[f];
                const size_t len2 = (size_t) j.last[1].stream_bytes;
                // z >= zfloor: len2 < zfloor + len1 implies len2 < z + len1 - the base layer alone wins (:838)
                if (!(zfloor[f] > 0 && len2 < zfloor[f] + j.len1)) continue;
                // (decided before a worker took it up: not queued yet, or queued - by the early group or the speculative list)
                uint8_t expect = kZNone;
                bool struck = zstate[f].compare_exchange_strong(expect, kZSkipped);
                if (!struck) { expect = kZQueued; struck = zstate[f].compare_exchange_strong(expect, kZSkipped); }
                if (struck) { skipped++; skipped_bytes += (long long) j.coeffs_size; }
            }
            if (overlap2) {
                std::vector<size_t> open;
                for (size_t f : cand) if (zstate[f] == kZNone) open.push_back(f);
                longest_first(open);
                if (!open.empty()) submit_zstd(open);
            }
            if (!zjoin()) return 1;
            pt.mark("zstd: wait for the workers");
            for (size_t f : with_prefix) if (jobs[f].need_pure) { skipped++; skipped_bytes += (long long) jobs[f].coeffs_size; }
            if (pt.on) fprintf(stderr, "ebcc-mi355x zstd: %.1f ms of core time for %lld bytes, longest job %.1f ms; floors %.1f ms; %lld of %zu prefixes (%lld bytes) not compressed\n",
                               zstd_us.load() / 1e3, zstd_bytes.load(), zstd_max_us.load() / 1e3, bound_us.load() / 1e3, skipped, with_prefix.size(), skipped_bytes);
            host_stats().skipped_bytes += skipped_bytes;
            if (pt.on && getenv("EBCC_HIP_ZSTD_TRACE"))
                for (size_t f : with_prefix) {
                    const Job &j = jobs[f];
                    const long long x = (long long) j.last[1].stream_bytes - (long long) j.len1;
                    fprintf(stderr, "zstd-trace c %zu floor %zu X %lld z %zu state %d need_pure %d nbad1 %llu len1 %zu orig %zu\n", j.coeffs_size, zfloor[f], x, j.zbytes.size(), (int) zstate[f].load(), (int) j.need_pure,
                            (unsigned long long) j.last[0].nbad, j.len1, j.coeffs_orig);
                }
            bool any_pure = false;
            for (size_t f = 0; f < n; f++) {
                Job &j = jobs[f];
                b.active[f] = 0;
                if (j.const_field) continue;
                const size_t len2 = (size_t) j.last[1].stream_bytes;
                const bool decided = zfloor[f] > 0 && len2 < zfloor[f] + j.len1;              // (z may not be known - only that it loses)
                if (decided || len2 < j.zbytes.size() + j.len1 || j.need_pure) {              // :838
                    if (decided) log_info("frame %zu: pure base compression (%zu) beats base (%zu) + residual (at least %zu)", f, len2, j.len1, zfloor[f]);
                    else if (len2 < j.zbytes.size() + j.len1)
                        log_info("frame %zu: pure base compression (%zu) beats base (%zu) + residual (%zu)", f, len2, j.len1, j.zbytes.size());
                    j.mean_err = j.last[1].err_sum / (double) n_pix;                          // :843
                    j.zbytes.clear(); j.coeffs_size = 0;
                    b.active[f] = 1; b.jf[f].cr = j.rs[1].result; any_pure = true;
                }
            }
            if (any_pure) {
                // the layer assignment of that rate again (allocation only, no decode), then its codestream
                b.push_jf();
                b.push_active();
                launch_j2k_rate(jb, (int) nt, b.d_active, s);
                b.collect_tails(jobs);
            }
        }
        if (!zjoin()) return 1;
        host_stats().add(zstd_us.load() + bound_us.load(), wait_us, zstd_bytes.load());
    }

    pt.mark("fallback search + tails");
    // ---- assemble (:863-907)
    for (size_t f = 0; f < n; f++) {
        Job &j = jobs[f];
        float minv = j.minv, maxv = j.maxv;
        log_info("frame %zu: mean of compression error %e", f, j.mean_err);
        if (!env.no_mean_adjust && std::fabs(j.mean_err) > 1e-18) {
            minv += j.mean_err;
            maxv += j.mean_err;
        }
        const size_t codec_size = j.const_field ? sizeof(uint64_t) : j.tail.size();
        const size_t total = sizeof(FrameHeader) + j.zbytes.size() + codec_size;
        uint8_t *o = (uint8_t *) malloc(total), *p = o;
        if (!o) { log_fatal("out of memory"); return 1; }
        FrameHeader hd;
        memset(&hd, 0, sizeof hd);
        memcpy(hd.magic, EBCC_HEADER_MAGIC, 4);
        hd.version = EBCC_HEADER_VERSION;
        if (j.const_field) hd.flags |= EBCC_HEADER_FLAG_CONST_FIELD;
        hd.minval_bits = f2u(minv); hd.maxval_bits = f2u(maxv);
        hd.coeffs_size = j.coeffs_size;
        hd.rmin_bits = f2u(j.const_field ? 0.0f : j.rmin); hd.rmax_bits = f2u(j.const_field ? 0.0f : j.rmax);
        hd.compressed_size = j.zbytes.size(); hd.tail_size = codec_size;
        memcpy(p, &hd, sizeof hd); p += sizeof hd;
        if (!j.zbytes.empty()) { memcpy(p, j.zbytes.data(), j.zbytes.size()); p += j.zbytes.size(); }
        if (j.const_field) { uint64_t cnt = n_pix; memcpy(p, &cnt, 8); }
        else memcpy(p, j.tail.data(), j.tail.size());
        log_info("frame %zu: coeffs_size %zu compressed_size %zu jp2_length %zu ratio %f", f, j.coeffs_size, j.zbytes.size(),
                 codec_size, (double) (n_pix * 4) / (double) total);
        outs[f] = o;
        sizes[f] = total;
    }
    pt.mark("assemble");
    return 0;
}

// One frame stream, either format: the 48-byte "EBCC" header (:190-202, :1234-1260) or the legacy header-less
// prefix `f32 min, f32 max, u64 coeffs_size, f32 rmin, f32 rmax, u64 compressed_size` (ebcc_decode_legacy,
// :1147-1213), where a constant field is signalled by min == max.
bool parse_frame(const uint8_t *d, size_t len, ParsedFrame &pf)
{
    if (len >= sizeof(FrameHeader) && memcmp(d, EBCC_HEADER_MAGIC, 4) == 0) {
        FrameHeader hd;
        memcpy(&hd, d, sizeof hd);
        if (hd.version != EBCC_HEADER_VERSION) { log_fatal("Unsupported EBCC header version: %u", hd.version); return false; }
        size_t used = sizeof hd;
        if (hd.compressed_size > len - used) { log_fatal("Invalid encoded data: truncated payload"); return false; }   // :1249
        used += hd.compressed_size;
        if (hd.tail_size > len - used) { log_fatal("Invalid encoded data: truncated payload"); return false; }         // :1254
        used += hd.tail_size;
        if (used != len) { log_fatal("Invalid encoded data: payload size mismatch"); return false; }                  // :1314
        pf.minv = u2f(hd.minval_bits); pf.maxv = u2f(hd.maxval_bits);
        pf.rmin = u2f(hd.rmin_bits); pf.rmax = u2f(hd.rmax_bits);
        pf.const_field = (hd.flags & EBCC_HEADER_FLAG_CONST_FIELD) != 0;
        pf.coeffs_size = hd.coeffs_size; pf.compressed_size = hd.compressed_size; pf.tail_size = hd.tail_size;
        pf.z = d + sizeof hd; pf.tail = pf.z + hd.compressed_size;
        if (pf.const_field && hd.tail_size != sizeof(uint64_t)) {
            log_fatal("Invalid encoded data: const-field payload must contain uint64_t length");
            return false;
        }
    } else {
        const size_t prefix = 4 + 4 + 8 + 4 + 4 + 8;
        if (len < prefix) { log_fatal("Invalid legacy encoded data: truncated header"); return false; }
        uint64_t cs, zs;
        memcpy(&pf.minv, d, 4); memcpy(&pf.maxv, d + 4, 4); memcpy(&cs, d + 8, 8);
        memcpy(&pf.rmin, d + 16, 4); memcpy(&pf.rmax, d + 20, 4); memcpy(&zs, d + 24, 8);
        if (zs > len - prefix) { log_fatal("Invalid legacy encoded data: truncated residual payload"); return false; }
        pf.coeffs_size = cs; pf.compressed_size = zs;
        pf.z = d + prefix; pf.tail = pf.z + zs; pf.tail_size = len - prefix - zs;
        pf.const_field = pf.minv == pf.maxv;
        if (pf.const_field && pf.tail_size < sizeof(uint64_t)) { log_fatal("Invalid legacy encoded data: missing const-field length"); return false; }
    }
    if (pf.const_field && pf.compressed_size > 0 && pf.coeffs_size > 0) {
        log_fatal("Invalid encoded data: residual data cannot be applied to const field");
        return false;
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// ebcc_decode for a batch of single-frame EBCC streams -> device buffer d_out [n][H*W]
// ------------------------------------------------------------------------------------------------
int decode_batch(ebcc_hip_ctx *ctx, const uint8_t *const *streams, const size_t *sizes, size_t n, float *d_out,
                 SliceGate *next)
{
    struct Release { SliceGate *g; ~Release() { if (g) g->release(); } } release_on_exit{next};
    J2kBuffers &jb = *static_cast<J2kBuffers *>(ctx->j2k);
    hipStream_t s = ctx->stream;
    const size_t n_pix = ctx->n_pix;
    const J2kGeom &g = jb.geom;
    int *const table = ctx->h_table;                                  // (pinned)
    const size_t table_ints = n * (size_t) g.stride * 4;
    memset(table, 0, table_ints * sizeof(int));
    // pieces to upload: codestream k = f, SPIHT bytes k = n + f - staged in pinned memory and sent as one copy
    std::vector<size_t> piece(2 * n, 0), piece_off(2 * n, 0);
    std::vector<ParsedFrame> heads(n);
    PhaseTimer pt;
    fetch_frame_states(ctx, n);
    // the host side of a batch - frame headers, packet headers, zstd of the residual streams - is per frame and runs on a
    // few host threads (decode is one slice: nothing else hides it); a frame's failure fails the batch
    std::atomic<bool> failed{false}, resid{false};
    // (the reason a frame was rejected is written on the worker's thread - set_error's text is per thread; the first one is
    //  carried over to the calling thread, where ebcc_hip_last_error is read)
    auto for_frames = [&](auto body) {
        const unsigned width = (unsigned) std::min<size_t>({(size_t) 16, (size_t) entropy_threads(1), (n + 7) / 8});
        auto batch = HostPool::instance().submit(n, width, [&](size_t f) {
            if (failed.load(std::memory_order_relaxed)) return;
            clear_error();
            if (!body(f)) {
                failed = true;
                const char *why = ebcc_hip_last_error();
                throw std::runtime_error(why && *why ? why : "invalid encoded data");
            }
        });
        if (!batch->wait()) { failed = true; set_error("%s", batch->error.c_str()); }
    };
    for_frames([&](size_t f) -> bool {
        const uint8_t *d = streams[f];
        const size_t len = sizes[f];
        ctx->h_active[f] = 0;
        ParsedFrame &hd = heads[f];
        if (!parse_frame(d, len, hd)) return false;
        FrameState &fs = ctx->h_fs[f];
        fs.minv = hd.minv; fs.maxv = hd.maxv;
        fs.rmin = hd.rmin; fs.rmax = hd.rmax;
        fs.const_field = hd.const_field ? 1 : 0;
        if (fs.const_field) {
            uint64_t cnt = 0;
            memcpy(&cnt, hd.tail, 8);
            if (cnt != n_pix) { log_fatal("const-field length %llu does not match the frame", (unsigned long long) cnt); return false; }
        } else {
            if (hd.tail_size > jb.stream_cap) { log_fatal("codestream larger than the device slot"); return false; }
            if (!j2k_parse_codestream(hd.tail, hd.tail_size, g, table + f * g.stride * 4)) return false;
            piece[f] = hd.tail_size;
            if (hd.compressed_size > 0 && hd.coeffs_size > 0) {                                                    // :1294-1304
                if (!zstd().ok) { log_fatal("libzstd not available"); return false; }
                if (hd.coeffs_size > ctx->rb.stream_words * 4 - 64) { log_fatal("residual stream larger than the device slot"); return false; }
                piece[n + f] = hd.coeffs_size;
                ctx->h_active[f] = 1;
                resid = true;
            }
        }
        return true;
    });
    if (failed) return 1;
    const bool any_resid = resid;
    stage_reserve(ctx, piece.data(), piece_off.data(), 2 * n);
    for_frames([&](size_t f) -> bool {
        const ParsedFrame &hd = heads[f];
        if (piece[f]) memcpy(ctx->h_stage + piece_off[f], hd.tail, hd.tail_size);
        if (piece[n + f]) {
            // the residual stream: exactly coeffs_size bytes (the staging buffer holds whatever an earlier call left),
            // a SPIHT header for this grid and a bit budget the decoder can work with (:1294-1304)
            const size_t got = zstd().decompress(ctx->h_stage + piece_off[n + f], hd.coeffs_size, hd.z, hd.compressed_size);
            if ((zstd().is_error && zstd().is_error(got)) || got != hd.coeffs_size) { log_fatal("Invalid encoded data: residual payload does not decompress to %zu bytes", hd.coeffs_size); return false; }
            if (check_ims_header(ctx, ctx->h_stage + piece_off[n + f], hd.coeffs_size, hd.coeffs_size * 8)) { log_fatal("Invalid encoded data: %s", ebcc_hip_last_error()); return false; }
        }
        return true;
    });
    if (failed) return 1;
    stage_send(ctx, 2 * n, s);
    stage_scatter(ctx, jb.stream, jb.stream_cap, 0, n, s);
    pt.mark("decode: parse, zstd, uploads");
    push_frame_states(ctx, n);
    // The residual layer (SPIHT decode + synthesis: one wave per frame, latency-bound) does not depend on the
    // base layer until the final addition, so it runs on the engine's second stream beside the tier-1 decode.
    hipStream_t s2 = s;
    if (any_resid) {
        s2 = second_stream(ctx);
        if (!ctx->ev_a) {
            EBCC_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_a, hipEventDisableTiming));
            EBCC_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_b, hipEventDisableTiming));
        }
        EBCC_HIP_CHECK(hipEventRecord(ctx->ev_a, s));                       // frame states and the staged pieces are on the device
        EBCC_HIP_CHECK(hipStreamWaitEvent(s2, ctx->ev_a, 0));
    }
    // the residual stream is fed first: its one-wave-per-frame kernel has to find a wave slot on every CU, and once the
    // tier-1 decoder's ~10^4 workgroups (longest code-blocks first) hold the slots none frees up for milliseconds
    if (any_resid) {
        const size_t slot = ctx->rb.stream_words * 4;
        for (size_t f = 0; f < n; f++) {
            ctx->h_u64a[f] = piece[n + f];
            ctx->h_u64b[f] = piece[n + f] * 8;
        }
        stage_scatter(ctx, (uint8_t *) ctx->rb.stream, slot, n, n, s2);
        EBCC_HIP_CHECK(hipMemcpyAsync(ctx->d_u64a, ctx->h_u64a, n * sizeof(unsigned long long), hipMemcpyHostToDevice, s2));
        EBCC_HIP_CHECK(hipMemcpyAsync(ctx->d_u64b, ctx->h_u64b, n * sizeof(unsigned long long), hipMemcpyHostToDevice, s2));
        EBCC_HIP_CHECK(hipMemcpyAsync(ctx->d_active, ctx->h_active, n * sizeof(int), hipMemcpyHostToDevice, s2));
        launch_spiht_decode((const uint8_t *) ctx->rb.stream, slot, ctx->d_u64a, ctx->d_u64b, ctx->rb, (int) n, ctx->d_active, s2);
        launch_synthesis_head(ctx->rb, (int) n, ctx->d_active, s2);
        if (s2 != s) EBCC_HIP_CHECK(hipEventRecord(ctx->ev_b, s2));
    }
    EBCC_HIP_CHECK(hipMemcpyAsync(jb.dec_table, table, table_ints * sizeof(int), hipMemcpyHostToDevice, s));
    // the decoded field is written where the caller wants it (the engine's own field buffer and a 1 GB device-to-device copy
    // per 256 frames only for an output that is not aligned the way the engine's buffers are)
    const bool direct = ((uintptr_t) d_out & 255u) == 0;
    J2kBuffers view = jb;
    if (direct) view.DEC = d_out;
    launch_j2k_decode(view, (int) n, s, table);
    if (next) { next->release(); release_on_exit.g = nullptr; }     // host parsing done, kernels queued
    if (any_resid) {
        if (s2 != s) EBCC_HIP_CHECK(hipStreamWaitEvent(s, ctx->ev_b, 0));
        launch_synthesis_tail_add(view.DEC, ctx->rb, (int) n, ctx->d_active, s);     // last row pass: field += residual
    }
    if (!direct) EBCC_HIP_CHECK(hipMemcpyAsync(d_out, jb.DEC, n * n_pix * sizeof(float), hipMemcpyDeviceToDevice, s));
    // constant fields: fill on the host side of the copy (rare path)
    for (size_t f = 0; f < n; f++)
        if (ctx->h_fs[f].const_field) {
            std::vector<float> v(n_pix, ctx->h_fs[f].minv);
            EBCC_HIP_CHECK(hipMemcpyAsync(d_out + f * n_pix, v.data(), n_pix * sizeof(float), hipMemcpyHostToDevice, s));
            wait_stream(s);
        }
    wait_stream(s);
    pt.mark("decode: kernels");
    return 0;
}

// Chunks of several frames: the tail is one codestream with a tile per frame (reference :121-125); the tiles are
// decoded as frames of `ctx`, the residual of the whole chunk image in `rc`.
// Frame heights a chunk of several frames can have: OpenJPEG cannot set up 6 resolutions on smaller tiles (the
// reference crashes on them), and the chunk image itself is bounded by the reference's 2047-row limit.
bool tile_height_supported(size_t h) { return h >= 32 && h <= 1023; }
// Heights for which every tile has the geometry of a tile at the origin (sub-band extents, parity and code-block
// partition repeat): the context then needs a single geometry for all tile positions.
bool tile_geometry_uniform(size_t h) { return h >= 32 && h <= 1024 && (h & (h - 1)) == 0; }

int decode_tiled(ebcc_hip_ctx *ctx, ebcc_hip_ctx *rc, const uint8_t *const *streams, const size_t *sizes, size_t n, size_t tiles,
                 float *d_out)
{
    J2kBuffers &jb = *static_cast<J2kBuffers *>(ctx->j2k);
    hipStream_t s = ctx->stream, rs = rc->stream;
    const J2kGeom &g = jb.geom;
    const size_t tile_pix = ctx->n_pix, n_pix = tile_pix * tiles, nt = n * tiles;
    std::vector<int> table(nt * (size_t) g.stride * 4, 0);
    std::vector<std::vector<uint8_t>> coeffs(n);
    std::vector<size_t> off(tiles), len(tiles);
    bool any_resid = false;
    for (size_t c = 0; c < n; c++) {
        ParsedFrame hd;
        if (!parse_frame(streams[c], sizes[c], hd)) return 1;
        rc->h_active[c] = 0;
        FrameState &r = rc->h_fs[c];
        r = FrameState{};
        r.minv = hd.minv; r.maxv = hd.maxv; r.rmin = hd.rmin; r.rmax = hd.rmax; r.const_field = hd.const_field ? 1 : 0;
        for (size_t k = 0; k < tiles; k++) {
            FrameState &t = ctx->h_fs[c * tiles + k];
            t = FrameState{};
            t.minv = hd.minv; t.maxv = hd.maxv; t.const_field = r.const_field;
        }
        if (hd.const_field) {
            uint64_t cnt = 0;
            memcpy(&cnt, hd.tail, 8);
            if (cnt != n_pix) { log_fatal("const-field length %llu does not match the chunk", (unsigned long long) cnt); return 1; }
            continue;
        }
        if (!j2k_parse_tiled(hd.tail, hd.tail_size, jb, (int) tiles, table.data() + c * tiles * g.stride * 4, off.data(), len.data())) {
            log_fatal("Invalid encoded data: %s", ebcc_hip_last_error());
            return 1;
        }
        for (size_t k = 0; k < tiles; k++) {
            if (len[k] > jb.stream_cap) { log_fatal("tile-part larger than the device slot"); return 1; }
            EBCC_HIP_CHECK(hipMemcpyAsync(jb.stream + (c * tiles + k) * jb.stream_cap, hd.tail + off[k], len[k], hipMemcpyHostToDevice, s));
        }
        if (hd.compressed_size > 0 && hd.coeffs_size > 0) {
            if (hd.coeffs_size > rc->rb.stream_words * 4 - 64) { log_fatal("residual stream larger than the device slot"); return 1; }
            coeffs[c].assign(hd.coeffs_size, 0);
            const size_t got = zstd().decompress(coeffs[c].data(), hd.coeffs_size, hd.z, hd.compressed_size);
            if ((zstd().is_error && zstd().is_error(got)) || got != hd.coeffs_size) { log_fatal("Invalid encoded data: residual payload does not decompress to %zu bytes", hd.coeffs_size); return 1; }
            if (check_ims_header(rc, coeffs[c].data(), hd.coeffs_size, hd.coeffs_size * 8)) { log_fatal("Invalid encoded data: %s", ebcc_hip_last_error()); return 1; }
            rc->h_active[c] = 1;
            any_resid = true;
        }
    }
    push_frame_states(ctx, nt);
    EBCC_HIP_CHECK(hipMemcpyAsync(jb.dec_table, table.data(), table.size() * sizeof(int), hipMemcpyHostToDevice, s));
    launch_j2k_decode(jb, (int) nt, s, table.data());
    wait_stream(s);
    if (any_resid) {
        push_frame_states(rc, n);
        const size_t slot = rc->rb.stream_words * 4;
        for (size_t c = 0; c < n; c++) {
            rc->h_u64a[c] = coeffs[c].size();
            rc->h_u64b[c] = coeffs[c].size() * 8;
            if (rc->h_active[c])
                EBCC_HIP_CHECK(hipMemcpyAsync((uint8_t *) rc->rb.stream + c * slot, coeffs[c].data(), coeffs[c].size(), hipMemcpyHostToDevice, rs));
        }
        EBCC_HIP_CHECK(hipMemcpyAsync(rc->d_u64a, rc->h_u64a, n * sizeof(unsigned long long), hipMemcpyHostToDevice, rs));
        EBCC_HIP_CHECK(hipMemcpyAsync(rc->d_u64b, rc->h_u64b, n * sizeof(unsigned long long), hipMemcpyHostToDevice, rs));
        EBCC_HIP_CHECK(hipMemcpyAsync(rc->d_active, rc->h_active, n * sizeof(int), hipMemcpyHostToDevice, rs));
        launch_spiht_decode((const uint8_t *) rc->rb.stream, slot, rc->d_u64a, rc->d_u64b, rc->rb, (int) n, rc->d_active, rs);
        launch_synthesis_head(rc->rb, (int) n, rc->d_active, rs);
        launch_synthesis_tail_add(jb.DEC, rc->rb, (int) n, rc->d_active, rs);
        wait_stream(rs);
    }
    EBCC_HIP_CHECK(hipMemcpyAsync(d_out, jb.DEC, n * n_pix * sizeof(float), hipMemcpyDeviceToDevice, s));
    for (size_t c = 0; c < n; c++)
        if (rc->h_fs[c].const_field) {
            std::vector<float> v(n_pix, rc->h_fs[c].minv);
            EBCC_HIP_CHECK(hipMemcpyAsync(d_out + c * n_pix, v.data(), n_pix * sizeof(float), hipMemcpyHostToDevice, s));
            wait_stream(s);
        }
    wait_stream(s);
    return 0;
}

}  // namespace ebcc
