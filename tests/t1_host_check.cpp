// Host build of the product's tier-1 core (ebcc_amd/csrc/t1_core.hpp) checked block-by-block against the
// oracle's tier-1 (oracle/j2k_oracle.c, itself pinned to OpenJPEG 2.4.0).  Test-only binary.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../ebcc_amd/csrc/t1_core.hpp"

extern "C" {
int orc_j2k_t1_encode(const int32_t *q, int w, int h, int orient, int level, float stepsize, uint8_t *out, int out_cap,
                      int *numbps, int *rates, double *disto);
void orc_j2k_t1_decode(const uint8_t *data, int len, int numbps, int npasses, int w, int h, int orient, int32_t *out);
}

using namespace ebcc::t1;

struct HostStore {
    u64 s[66], neg[64], vis[64], ref[64], sps[64], sgn[64];
    std::vector<u64> bp;          // [plane][64]
    int32_t *out = nullptr;       // decoder output (half units), row stride w
    int w = 0;
    HostStore() { memset(s, 0, sizeof s); memset(neg, 0, sizeof neg); memset(vis, 0, sizeof vis); memset(ref, 0, sizeof ref); memset(sps, 0, sizeof sps); memset(sgn, 0, sizeof sgn); }
    u64 &S(int y) { return s[y + 1]; }
    u64 &NEG(int y) { return neg[y]; }
    u64 &VIS(int y) { return vis[y]; }
    u64 &REF(int y) { return ref[y]; }
    u64 &SPS(int y) { return sps[y]; }
    u64 SGN(int y) { return y < 64 ? sgn[y] : 0; }
    u64 BP(int plane, int y) { return y < 64 ? bp[(size_t) plane * 64 + y] : 0; }
    void set_sig(int x, int y, int negv, int plane)
    {
        int one = 1 << (plane + 1), v = one | (one >> 1);
        out[y * w + x] = negv ? -v : v;
    }
    void refine(int x, int y, int bit, int plane, int negv)
    {
        int half = 1 << plane;
        out[y * w + x] += (bit ^ negv) ? half : -half;
    }
};
struct VecSink { std::vector<uint8_t> *v; void put(int i, uint8_t b) { if ((int) v->size() <= i) v->resize(i + 1); (*v)[i] = b; } };
struct VecSink2 { std::vector<uint8_t> *v; void put(int i, uint32_t b) { if (i < 0) return; if ((int) v->size() <= i) v->resize(i + 1); (*v)[i] = (uint8_t) b; } void finish() {} };
struct VecAt { std::vector<uint8_t> *v; uint8_t operator()(int i) const { return i < (int) v->size() ? (*v)[i] : 0; } };
struct Obs {                      // decoder registers at the start of every (pass, stripe)
    MqCheckpoint ck[120 * 16];
    u64 visp[40][64];
    int cur = 0, nstr = 16;
    template <class Mq> void pass_start(int p, const Mq &) { cur = p; }
    template <class Mq> void stripe_start(int y0, const Mq &m) { ck[cur * nstr + (y0 >> 2)] = MqCheckpoint{m.a, m.c, m.ct, m.pos, m.cx.w0, m.cx.w1, m.cx.w2}; }
    template <class Store> void sigprop_done(int bp, Store &st) { for (int y = 0; y < 64; y++) visp[bp][y] = st.VIS(y); }
};
struct EncObs {                   // the same, derived from the encoder
    MqCheckpoint ck[120 * 16];
    u64 visp[40][64];
    int cur = 0, nstr = 16;
    template <class Mq> void pass_start(int p, const Mq &) { cur = p; }
    template <class Mq> void stripe_start(int y0, const Mq &m) { ck[cur * nstr + (y0 >> 2)] = encoder_checkpoint(m); }
    template <class Store> void sigprop_done(int bp, Store &st) { for (int y = 0; y < 64; y++) visp[bp][y] = st.VIS(y); }
};
struct SymPut { std::vector<uint8_t> *v; void operator()(uint32_t i, uint32_t sym) { if (v->size() <= i) v->resize(i + 1); (*v)[i] = (uint8_t) sym; }
                void put_if(bool on, uint32_t i, uint32_t sym) { if (on) (*this)(i, sym); } };
struct SymGet { const std::vector<uint8_t> *v; uint32_t get(uint32_t i) const { return (*v)[i]; } };
struct EmitObs {                  // phase 1 observer (the stripe markers go into the stream itself)
    u64 visp[40][64];
    template <class Em> void pass_start(int, const Em &) {}
    template <class Store> void sigprop_done(int bp, Store &st) { for (int y = 0; y < 64; y++) visp[bp][y] = st.VIS(y); }
};
struct CtxBytes { uint8_t *b; uint32_t ld(int c) const { return b[c]; } void st(int c, uint32_t v) { b[c] = (uint8_t) v; }
                  void words(uint32_t x[5]) const { for (int j = 0; j < 5; j++) x[j] = 0; for (int i = 0; i < NCTX; i++) x[i >> 2] |= (uint32_t) b[i] << (8 * (i & 3)); } };
struct CkView {                   // CkArray over a plain [pass * nstr + stripe] array
    MqCheckpoint *ck; int nstr;
    uint32_t shifts(int p, int s) const { return (uint32_t) ck[p * nstr + s].pos; }
    uint32_t c16(int p, int s) const { return ck[p * nstr + s].c; }
    void finish(int p, int s, uint32_t c, int ct, int pos) { MqCheckpoint &k = ck[p * nstr + s]; k.c = c; k.ct = ct; k.pos = pos; }
    void store(int p, int s, uint32_t a, uint32_t c16, uint32_t shifts, const uint32_t x[5])
    {
        const Contexts c = Contexts::from_bytes(x);
        ck[p * nstr + s] = MqCheckpoint{a, c16, 0, (int) shifts, c.w0, c.w1, c.w2};
    }
    void store_interval(int p, int s, uint32_t a, uint32_t shifts, const uint32_t x[5])
    {
        const Contexts c = Contexts::from_bytes(x);
        MqCheckpoint &k = ck[p * nstr + s];
        k.a = a; k.ct = 0; k.pos = (int) shifts; k.w0 = c.w0; k.w1 = c.w1; k.w2 = c.w2;
    }
    void store_code(int p, int s, uint32_t c16) { ck[p * nstr + s].c = c16; }
};

// ---- segmented encoder (scan_block / emit_stripe_segments / mq_encode_segments) ----
struct HostMasks {                 // the encoder's masks of one code-block
    std::vector<u64> bp, suf, visp;    // [plane][64], suf/visp with two spare planes
    u64 sgn[64], sps[64];
    int P;
    u64 BP(int p, int y) { return p < P ? bp[(size_t) p * 64 + y] : 0; }
    u64 SUF(int p, int y) { return suf[(size_t) p * 64 + y]; }
    u64 SGN(int y) { return sgn[y]; }
    u64 VISP(int p, int y) { return visp[(size_t) p * 64 + y]; }
};
struct ScanOut {
    HostMasks *m; std::vector<uint32_t> *lens;
    void visp(int p, int y, u64 v) { m->visp[(size_t) p * 64 + y] = v; }
    void sps_or(int y, u64 v) { m->sps[y] |= v; }
    void len(int seg, uint32_t n) { (*lens)[seg] = n; }
};
struct SegEm {
    std::vector<std::vector<uint8_t>> *segs; int cur = -1;
    void begin(int seg) { cur = seg; (*segs)[seg].clear(); }
    void emit_if(bool on, uint32_t ctx, uint32_t d) { if (on) (*segs)[cur].push_back((uint8_t) (ctx | (d << 5))); }
    void column_end() {}
    void end() {}
};
struct RowSrcHost {               // one code-block's row stream (t1_core.hpp: row format)
    const std::vector<uint32_t> *w;
    uint32_t rows() const { return (uint32_t) (w->size() / 4); }
    uint32_t wave_rows() const { return rows() + 3; }                     // (a wave runs as long as its longest lane: extra rows must be ignored)
    void load(uint32_t row, uint32_t o[4]) const { for (int k = 0; k < 4; k++) o[k] = row < rows() ? (*w)[(size_t) row * 4 + k] : 0x12345678u; }
    bool any(bool b) const { return b; }
    void sync(uint32_t) const {}
    void finish() const {}
};
struct HostHand { std::vector<uint32_t> *v; void put(uint32_t row, int j, uint32_t word) { size_t i = (size_t) row * 16 + j; if (v->size() <= i) v->resize(i + 1); (*v)[i] = word; } };
struct CtxSlots { uint32_t *b; uint32_t handle(uint32_t c) const { return c; } uint32_t ld(uint32_t h) const { return b[h]; } void st(uint32_t h, uint32_t v) { b[h] = v; }
    void words(uint32_t x[5]) const { for (int j = 0; j < 5; j++) { uint32_t v = 0; for (int k = 0; k < 4; k++) if (4 * j + k < NCTX) v |= mq_code_state(b[4 * j + k] >> 16) << (8 * k); x[j] = v; } } };
struct VecSink3 { std::vector<uint8_t> *v; void put(int i, uint32_t b) { if (i < 0) return; if ((int) v->size() <= i) v->resize(i + 1); (*v)[i] = (uint8_t) b; } void row_end(int) {} void finish(int n) { v->resize((size_t) n); } };
struct BufSrc { const uint8_t *p; int n; uint32_t get(int i) const { return i < n ? p[i] : 0xFFu; } };
// a source of the sequential kind (t1_core.hpp: source_is_sequential; the device's is DecSrcSeq of j2k_rate.hip): the
// decoder only asks for the byte it stands on, the one after it, and a step forward
struct SeqSrc {
    static constexpr bool kSequential = true;
    const uint8_t *p; int n; int o = 0;
    uint32_t at0() const { return o < n ? p[o] : 0xFFu; }
    uint32_t at1() const { return o + 1 < n ? p[o + 1] : 0xFFu; }
    void step() { o++; }
    uint32_t get(int) const { return 0u; }            // (must not be used: a wrong byte would show in the decode)
};
struct HostEnv { BufSrc src; bool any(bool b) const { return b; } BufSrc &bytes() { return src; } bool starved(int) const { return false; } void refill_point(int) {} void pass_point(int) {} };

int main(int argc, char **argv)
{
    int trials = argc > 1 ? atoi(argv[1]) : 300;
    std::mt19937 rng(12345);
    int bad = 0;
    for (int t = 0; t < trials; t++) {
        int w = (t % 5 == 0) ? 64 : 1 + rng() % 64, h = (t % 7 == 0) ? 64 : 1 + rng() % 64;
        int orient = rng() % 4;
        int maxbits = 7 + rng() % 16;                  // q6 magnitude bits (6 fractional)
        int style = rng() % 5;                        // 4: a few coefficients in an empty code-block (long run-length runs)
        std::vector<int32_t> q((size_t) w * h);
        std::vector<int32_t> &qvals = q;
        for (int i = 0; i < w * h; i++) {
            int32_t m;
            if (style == 0) m = (int32_t) (rng() % (1u << maxbits));
            else if (style == 1) m = (rng() % 8 == 0) ? (int32_t) (rng() % (1u << maxbits)) : (int32_t) (rng() % 64);
            else if (style == 4) m = (rng() % 300 == 0) ? (int32_t) (rng() % (1u << maxbits)) : 0;
            else if (style == 2) { int b = rng() % (maxbits + 1); m = (int32_t) (rng() % (1u << b)); }
            else { int x = i % w, y = i / w; m = (int32_t) ((((x * x + 3 * y * y) % 4099) * (1u << maxbits)) / 4099); }
            q[i] = (rng() & 1) ? -m : m;
        }
        // ---- oracle
        std::vector<uint8_t> ob((size_t) w * h * 8 + 64);
        int onumbps = 0, orates[200];
        double odisto[200];
        int opasses = orc_j2k_t1_encode(q.data(), w, h, orient, 0, 1.0f, ob.data(), (int) ob.size(), &onumbps, orates, odisto);
        // ---- product core
        uint32_t mx = 0;
        for (auto v : q) { uint32_t a = (uint32_t) (v < 0 ? -v : v); if (a > mx) mx = a; }
        int numbps = 0;
        if (mx) { int fl = 31 - __builtin_clz(mx); numbps = fl + 1 - 6; }
        if (numbps != onumbps && !(numbps <= 0 && opasses == 0)) { printf("trial %d numbps %d vs %d\n", t, numbps, onumbps); bad++; continue; }
        if (numbps <= 0) { if (opasses != 0) { printf("trial %d expected no passes\n", t); bad++; } continue; }
        HostStore st;
        st.bp.assign((size_t) numbps * 64, 0);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int32_t v = q[(size_t) y * w + x];
                uint32_t a = (uint32_t) (v < 0 ? -v : v) >> 6;
                if (v < 0) st.sgn[y] |= 1ull << x;
                for (int p = 0; p < numbps; p++) if ((a >> p) & 1) st.bp[(size_t) p * 64 + y] |= 1ull << x;
            }
        std::vector<uint8_t> bytes;
        int rates[kMaxPasses];
        static EncObs eobs;
        memset(eobs.visp, 0, sizeof eobs.visp);
        const int nstr = (h + 3) / 4;
        eobs.nstr = nstr;
        EncodeResult r = encode_block_observed(st, VecSink{&bytes}, VecAt{&bytes}, w, h, orient, numbps, rates, eobs);
        CkView ckv{eobs.ck, nstr};
        finalize_checkpoints(ckv, r.totalpasses, nstr, BufSrc{bytes.data(), r.length});
        // ---- the two-phase encoder must give the same bytes, rates and checkpoints
        {
            HostStore st2 = st;
            memset(st2.s, 0, sizeof st2.s); memset(st2.neg, 0, sizeof st2.neg); memset(st2.vis, 0, sizeof st2.vis);
            memset(st2.ref, 0, sizeof st2.ref); memset(st2.sps, 0, sizeof st2.sps);
            std::vector<uint8_t> syms, bytes2;
            static EmitObs mobs;
            memset(mobs.visp, 0, sizeof mobs.visp);
            SymbolEmitter<SymPut> em{0, SymPut{&syms}};
            const int np2 = emit_block<true>(st2, em, w, h, orient, numbps, mobs);
            {
                HostStore st3 = st;
                memset(st3.s, 0, sizeof st3.s); memset(st3.neg, 0, sizeof st3.neg); memset(st3.vis, 0, sizeof st3.vis);
                memset(st3.ref, 0, sizeof st3.ref); memset(st3.sps, 0, sizeof st3.sps);
                std::vector<uint8_t> syms3;
                static EmitObs mobs3;
                memset(mobs3.visp, 0, sizeof mobs3.visp);
                SymbolEmitter<SymPut> em3{0, SymPut{&syms3}};
                const int np3 = emit_block<false>(st3, em3, w, h, orient, numbps, mobs3);
                bool eq = np3 == np2 && em3.n == em.n && memcmp(syms3.data(), syms.data(), em.n) == 0 && memcmp(mobs3.visp, mobs.visp, sizeof mobs.visp) == 0;
                if (!eq) {
                    size_t k = 0; while (k < em.n && k < em3.n && syms[k] == syms3[k]) k++;
                    printf("trial %d BIT-PARALLEL EMISSION differs at %zu of %u/%u (w %d h %d orient %d P %d): %02x vs %02x\n", t, k, em.n, em3.n, w, h, orient, numbps,
                           k < em.n ? syms[k] : 0, k < em3.n ? syms3[k] : 0);
                    bad++; continue;
                }
            }
            // ---- the segmented encoder: scan (masks + counts), per-stripe emission, uniform-segment MQ pass
            {
                HostMasks hm;
                hm.P = numbps;
                hm.bp = st.bp;
                hm.suf.assign((size_t) (kSegPlanes + 2) * 64, 0);
                hm.visp.assign((size_t) (kSegPlanes + 2) * 64, 0xDEADBEEFDEADBEEFull);      // (unwritten rows must never matter)
                memcpy(hm.sgn, st.sgn, sizeof hm.sgn);
                memset(hm.sps, 0, sizeof hm.sps);
                for (int y = 0; y < 64; y++) { u64 acc = 0; for (int pl = numbps - 1; pl >= 0; pl--) { acc |= st.bp[(size_t) pl * 64 + y]; hm.suf[(size_t) pl * 64 + y] = acc; } }
                std::vector<uint32_t> lens(kSegCount, 0xFFFFFFFFu);
                ScanOut so{&hm, &lens};
                scan_block(hm, so, numbps, w, h, orient);
                bool okm = memcmp(hm.sps, st.sps, sizeof st.sps) == 0;
                for (int pl = 0; okm && pl < numbps - 1; pl++)
                    for (int yy = 0; yy < h; yy++) if (hm.visp[(size_t) pl * 64 + yy] != mobs.visp[pl][yy]) { okm = false; printf("trial %d scan visp plane %d row %d\n", t, pl, yy); break; }
                // segments of the marker stream
                std::vector<std::vector<uint8_t>> want(kSegCount);
                {
                    int pass = -1, stripe = 0, segi = -1;
                    for (uint32_t i = 0; i < em.n; i++) {
                        const uint8_t v = syms[i];
                        if (v & kSymStripe) {
                            if (v & kSymPass) { pass++; stripe = 0; } else stripe++;
                            segi = seg_index(plane_of_pass(numbps, pass), type_of_pass(pass), stripe);
                        } else want[segi].push_back(v);
                    }
                }
                std::vector<std::vector<uint8_t>> got(kSegCount);
                SegEm sem{&got};
                for (int pl = numbps - 1; pl >= 0; pl--)
                    for (int s_ = 0; s_ < nstr; s_++) emit_stripe_segments(hm, sem, numbps, pl, s_, w, h, orient);
                for (int sg = 0; okm && sg < kSegCount; sg++) {
                    const bool valid = seg_valid(numbps, nstr, seg_plane(sg), seg_type(sg), sg & 15);
                    if (!valid) { if (!want[sg].empty() || !got[sg].empty()) { okm = false; printf("trial %d decisions in invalid segment %d\n", t, sg); } continue; }
                    if (got[sg] != want[sg]) { okm = false; printf("trial %d segment %d (plane %d type %d stripe %d) differs: %zu vs %zu decisions\n", t, sg, seg_plane(sg), seg_type(sg), sg & 15, got[sg].size(), want[sg].size()); }
                    else if (lens[sg] != want[sg].size()) { okm = false; printf("trial %d segment %d count %u vs %zu\n", t, sg, lens[sg], want[sg].size()); }
                }
                if (!okm) { printf("trial %d SEGMENTED EMISSION mismatch (w %d h %d orient %d P %d)\n", t, w, h, orient, numbps); bad++; continue; }
                static MqCheckpoint ck3[120 * 16];
                memset(ck3, 0, sizeof ck3);
                CkView ckv3{ck3, nstr};
                int rates3[kMaxPasses];
                uint32_t ctxb3[32] = {0};
                std::vector<uint8_t> bytes3;
                std::vector<uint32_t> rowsw;                                // the segments as rows, in coding order
                for (int sg = 0; sg < kSegCount; sg++) {
                    if (!seg_valid(numbps, nstr, seg_plane(sg), seg_type(sg), sg & 15)) continue;
                    const std::vector<uint8_t> &v = got[sg];
                    const uint32_t nr = seg_rows((uint32_t) v.size());
                    for (uint32_t rr = 0; rr < nr; rr++)
                        for (int k = 0; k < 4; k++) {
                            uint32_t word = 0;
                            for (int bb = 0; bb < 4; bb++) {
                                const size_t i = (size_t) rr * 16 + k * 4 + bb;
                                uint32_t byte = i < v.size() ? v[i] : kRowPad;
                                if (rr == 0 && k == 0 && bb == 0) byte |= kRowStart;
                                word |= byte << (8 * bb);
                            }
                            rowsw.push_back(word);
                        }
                }
                std::vector<uint32_t> handv;
                const uint32_t a_end = mq_rows_interval(RowSrcHost{&rowsw}, CtxSlots{ctxb3}, numbps, nstr, HostHand{&handv}, ckv3);
                MqCodeChain chain;
                VecSink3 sink3{&bytes3};
                for (size_t rr = 0; rr * 16 < handv.size(); rr++) {
                    uint32_t hw[16];
                    for (int j = 0; j < 16; j++) hw[j] = handv[rr * 16 + j];
                    chain.row(hw, nstr, rates3, sink3, ckv3, [](bool b_) { return b_; });
                }
                EncodeResult r3 = chain.finish(a_end, numbps > 0 ? 3 * numbps - 2 : 0, rates3, sink3, VecAt{&bytes3});
                finalize_checkpoints(ckv3, r3.totalpasses, nstr, BufSrc{bytes3.data(), r3.length});
                bool same = r3.totalpasses == r.totalpasses && r3.length == r.length && memcmp(bytes3.data(), bytes.data(), (size_t) r.length) == 0;
                for (int p_ = 0; same && p_ < r.totalpasses; p_++) same = rates3[p_] == rates[p_];
                for (int i = 0; same && i < r.totalpasses * nstr; i++) same = memcmp(&ck3[i], &eobs.ck[i], sizeof(MqCheckpoint)) == 0;
                if (!same) { printf("trial %d SEGMENTED MQ mismatch (passes %d/%d len %d/%d)\n", t, r3.totalpasses, r.totalpasses, r3.length, r.length); bad++; continue; }
            }
            static MqCheckpoint ck2[120 * 16];
            CkView ckv2{ck2, nstr};
            int rates2[kMaxPasses];
            uint8_t ctxb[32] = {0};
            EncodeResult r2 = mq_encode_stream(SymGet{&syms}, CtxBytes{ctxb}, em.n, np2, VecSink2{&bytes2}, VecAt{&bytes2}, rates2, ckv2);
            finalize_checkpoints(ckv2, np2, nstr, BufSrc{bytes2.data(), r2.length});
            bool same = np2 == r.totalpasses && r2.length == r.length && memcmp(bytes2.data(), bytes.data(), (size_t) r.length) == 0 &&
                        memcmp(st2.sps, st.sps, sizeof st.sps) == 0;
            for (int p = 0; same && p < np2; p++) same = rates2[p] == rates[p];
            for (int i = 0; same && i < np2 * nstr; i++) same = memcmp(&ck2[i], &eobs.ck[i], sizeof(MqCheckpoint)) == 0;
            if (!same) { printf("trial %d TWO-PHASE ENCODER mismatch (passes %d/%d len %d/%d)\n", t, np2, r.totalpasses, r2.length, r.length); bad++; continue; }
        }
        bool ok = r.totalpasses == opasses && r.length == (opasses ? orates[opasses - 1] >= 0 ? r.length : 0 : 0);
        int olen = 0;
        // oracle's len is mq numbytes; recover it as the max rate (last pass rate equals it unless trimmed for FF)
        if (ok) for (int p = 0; p < opasses; p++) if (rates[p] != orates[p]) { ok = false; break; }
        (void) olen;
        if (ok && memcmp(bytes.data(), ob.data(), (size_t) r.length) != 0) ok = false;
        if (!ok) {
            printf("trial %d ENCODE mismatch w %d h %d orient %d numbps %d passes %d/%d len %d\n", t, w, h, orient, numbps,
                   r.totalpasses, opasses, r.length);
            bad++;
            continue;
        }
        // sigprop-significance bookkeeping must be consistent: subset of nonzero coefficients
        // ---- decode at several truncation points with both decoders
        for (int k = 0; k < 4; k++) {
            int np = k == 0 ? opasses : 1 + (int) (rng() % opasses);
            int len = rates[np - 1];
            std::vector<int32_t> d1((size_t) w * h, 0), d2((size_t) w * h, 0);
            orc_j2k_t1_decode(bytes.data(), len, numbps, np, w, h, orient, d1.data());
            HostStore ds;
            ds.out = d2.data(); ds.w = w;
            decode_block(ds, BufSrc{bytes.data(), len}, w, h, orient, numbps, np);
            if (d1 != d2) { printf("trial %d DECODE mismatch np %d/%d\n", t, np, opasses); bad++; break; }
            {
                std::vector<int32_t> d3((size_t) w * h, 0);
                HostStore ss;
                ss.out = d3.data(); ss.w = w;
                decode_block(ss, SeqSrc{bytes.data(), len}, w, h, orient, numbps, np);
                if (d1 != d3) { printf("trial %d DECODE (sequential source) mismatch np %d/%d\n", t, np, opasses); bad++; break; }
            }
            // ---- resume at the last coded plane from a checkpoint of the FULL-segment decode
            static Obs obs;
            obs.nstr = nstr;
            {
                std::vector<int32_t> dummy((size_t) w * h, 0);
                HostStore fs;
                fs.out = dummy.data(); fs.w = w;
                decode_block_observed(fs, BufSrc{bytes.data(), r.length}, w, h, orient, numbps, opasses, obs);
            }
            // the encoder-derived checkpoints (no decoding) must equal the decoder's own registers
            if (k == 0) {
                bool same = true;
                for (int i = 0; i < opasses * nstr && same; i++) {
                    const int p = i / nstr;
                    const MqCheckpoint &x = obs.ck[i], &y = eobs.ck[i];
                    same = x.a == y.a && x.c == y.c && x.ct == y.ct && x.pos == y.pos && x.w0 == y.w0 && x.w1 == y.w1 && x.w2 == y.w2;
                    if (!same) printf("trial %d pass %d ckpt dec {a %x c %x ct %d pos %d} enc {a %x c %x ct %d pos %d}\n", t, p, x.a, x.c, x.ct, x.pos, y.a, y.c, y.ct, y.pos);
                    if (same && i % nstr == 0 && type_of_pass(p) == 0) {
                        const int pl = plane_of_pass(numbps, p);
                        for (int yy = 0; yy < 64; yy++) if (obs.visp[pl][yy] != eobs.visp[pl][yy]) { same = false; printf("trial %d visp plane %d row %d\n", t, pl, yy); break; }
                    }
                }
                if (!same) { printf("trial %d ENCODER CHECKPOINT mismatch\n", t); bad++; break; }
            }
            // restart at the latest (pass, stripe) whose checkpoint is still valid for this truncation
            int idx = np * nstr - 1;
            while (idx > 0 && eobs.ck[idx].pos + 1 >= len) idx--;
            const int r = idx / nstr, rs_ = idx % nstr;
            std::vector<int32_t> d3((size_t) w * h, 0);
            HostStore rs;
            rs.out = d3.data(); rs.w = w;
            for (int y = 0; y < h; y++) {
                const int q_ = y < 4 * rs_ ? r + 1 : r;                 // rows above the restart stripe have finished pass r
                const int pq = plane_of_pass(numbps, q_), tq = q_ == 0 ? 0 : type_of_pass(q_);
                for (int x = 0; x < w; x++) {
                    int32_t src6 = qvals[(size_t) y * w + x];
                    uint32_t a = (uint32_t) (src6 < 0 ? -src6 : src6) >> 6;
                    if (!a) continue;
                    int bs = 31 - __builtin_clz(a);
                    bool sps = (st.sps[y] >> x) & 1;
                    int ps = bs == numbps - 1 ? 0 : 3 * (numbps - 1 - bs) - (sps ? 2 : 0);     // pass of first significance
                    if (ps >= q_) continue;
                    rs.s[y + 1] |= 1ull << x;
                    if (src6 < 0) rs.neg[y] |= 1ull << x;
                    int v = 3 << bs;
                    for (int pl = bs - 1; pl >= 0; pl--) {
                        if (3 * (numbps - 1 - pl) - 1 >= q_) break;
                        v += ((a >> pl) & 1) ? (1 << pl) : -(1 << pl);
                        rs.ref[y] |= 1ull << x;
                    }
                    d3[(size_t) y * w + x] = src6 < 0 ? -v : v;
                }
                if (tq != 0 && q_ > 0 && pq >= 0) rs.vis[y] = eobs.visp[pq][y];
            }
            decode_resume(rs, BufSrc{bytes.data(), len}, w, h, orient, numbps, np, r, rs_, eobs.ck[idx]);
            int qp = r;
            if (d1 != d3) { printf("trial %d RESUME mismatch np %d/%d q %d P %d\n", t, np, opasses, qp, numbps); bad++; break; }
        }
    }
    printf("t1_host_check: %d trials, %d failures\n", trials, bad);
    return bad != 0;
}
