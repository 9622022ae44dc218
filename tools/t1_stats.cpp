// Work statistics of the tier-1 coder on a real frame (development tool, not part of the product or the tests):
// symbols by kind, column iterations of the current kernels, per-pass balance across the 64 code-blocks a wave holds.
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
static long g_stat[8];
#define T1_STAT(i) (g_stat[i]++)
#include "../ebcc_amd/csrc/t1_core.hpp"
extern "C" {
void orc_j2k_set_block_sink(void (*fn)(const int32_t *, int, int, int, void *), void *user);
int orc_j2k_analysis(const uint16_t *img, size_t height, size_t width, int *numbps, int *totalpasses, int *lens, int *rates, double *disto, unsigned *hashes);
}
using namespace ebcc::t1;
struct Blk { std::vector<int32_t> q; int w, h, orient; };
static std::vector<Blk> blocks;
static void sink(const int32_t *q, int w, int h, int orient, void *) { blocks.push_back(Blk{std::vector<int32_t>(q, q + w * h), w, h, orient}); }

struct HostStore {
    u64 s[66], neg[64], vis[64], ref[64], sps[64], sgn[64];
    std::vector<u64> bp;
    HostStore() { memset(this, 0, offsetof(HostStore, bp)); }
    u64 &S(int y) { return s[y + 1]; } u64 &NEG(int y) { return neg[y]; } u64 &VIS(int y) { return vis[y]; } u64 &REF(int y) { return ref[y]; }
    u64 &SPS(int y) { return sps[y]; } u64 SGN(int y) { return y < 64 ? sgn[y] : 0; } u64 BP(int p, int y) { return y < 64 ? bp[(size_t) p * 64 + y] : 0; }
};
struct CountCoder;
struct StripeObs {               // symbol count at every stripe start
    std::vector<long> *marks; CountCoder *cc;
    template <class Mq> void pass_start(int, const Mq &) {}
    template <class Mq> void stripe_start(int, const Mq &m);
    template <class St> void sigprop_done(int, St &) {}
};
struct CountCoder {      // counts decisions by kind; no arithmetic coding
    long n[5] = {0, 0, 0, 0, 0};   // zc, sc, mr, agg, uni
    void encode_zc(int c, int d) { encode(c, d); } void encode_sc(int c, int d) { encode(c, d); } void encode_mag(int c, int d) { encode(c, d); }
    void encode_agg(int d) { encode(CTX_AGG, d); } void encode_uni(int d) { encode(CTX_UNI, d); }
    void encode(int ctx, int) { if (ctx <= CTX_ZC0 + 8 && ctx >= CTX_ZC0) n[0]++; else if (ctx >= CTX_SC0 && ctx < CTX_SC0 + 5) n[1]++; else if (ctx >= CTX_MAG0 && ctx < CTX_MAG0 + 3) n[2]++; else if (ctx == CTX_AGG) n[3]++; else n[4]++; }
};
template <class Mq> void StripeObs::stripe_start(int, const Mq &m) { marks->push_back(m.n[0] + m.n[1] + m.n[2] + m.n[3] + m.n[4]); }
int main(int argc, char **argv)
{
    const int H = 721, W = 1440;
    std::vector<uint16_t> img((size_t) H * W);
    FILE *f = fopen(argv[1], "rb"); if (!f || fread(img.data(), 2, img.size(), f) != img.size()) return 1; fclose(f);
    orc_j2k_set_block_sink(sink, nullptr);
    std::vector<int> nb(4096), tp(4096), ln(4096), rt(4096 * 100); std::vector<double> ds(4096 * 100); std::vector<unsigned> hs(4096);
    int n = orc_j2k_analysis(img.data(), H, W, nb.data(), tp.data(), ln.data(), rt.data(), ds.data(), hs.data());
    printf("%d blocks\n", n);
    long tot[5] = {0}, iters[3] = {0};
    // per block, per pass: symbols and iterations
    std::vector<std::vector<long>> sym(n), it(n), marks(n);
    long bytes = 0;
    for (int b = 0; b < n; b++) {
        Blk &B = blocks[b];
        uint32_t mx = 0; for (auto v : B.q) mx = std::max(mx, (uint32_t) abs(v));
        int P = mx ? 32 - __builtin_clz(mx) - 6 : 0;
        if (P <= 0) continue;
        bytes += ln[b];
        HostStore st; st.bp.assign((size_t) P * 64, 0);
        for (int y = 0; y < B.h; y++) for (int x = 0; x < B.w; x++) { int32_t v = B.q[y * B.w + x]; uint32_t a = (uint32_t) abs(v) >> 6; if (v < 0) st.sgn[y] |= 1ull << x; for (int p = 0; p < P; p++) if ((a >> p) & 1) st.bp[(size_t) p * 64 + y] |= 1ull << x; }
        CountCoder cc;
        StripeObs so{&marks[b], &cc};
        Passes<true, HostStore, CountCoder, StripeObs> ps(st, cc, B.w, B.h, B.orient, &so);
        int type = 2;
        for (int bp = P - 1; bp >= 0;) {
            long before = cc.n[0] + cc.n[1] + cc.n[2] + cc.n[3] + cc.n[4], ib = g_stat[0] + g_stat[1] + g_stat[2];
            if (type == 0) ps.sigprop(bp); else if (type == 1) ps.refine(bp); else ps.cleanup(bp);
            sym[b].push_back(cc.n[0] + cc.n[1] + cc.n[2] + cc.n[3] + cc.n[4] - before);
            it[b].push_back(g_stat[0] + g_stat[1] + g_stat[2] - ib);
            if (++type == 3) { type = 0; bp--; }
        }
        for (int k = 0; k < 5; k++) tot[k] += cc.n[k];
        marks[b].push_back(cc.n[0] + cc.n[1] + cc.n[2] + cc.n[3] + cc.n[4]);
    }
    for (int k = 0; k < 3; k++) iters[k] = g_stat[k];
    long S = tot[0] + tot[1] + tot[2] + tot[3] + tot[4];
    printf("symbols: zc %ld sc %ld mr %ld agg %ld uni %ld  total %ld  (%.1f per sample), bytes %ld (%.2f symbols/byte)\n", tot[0], tot[1], tot[2], tot[3], tot[4], S, (double) S / (H * W), bytes, (double) S / bytes);
    printf("column iterations: sigprop %ld refine %ld cleanup %ld total %ld (symbols/iteration %.2f)\n", iters[0], iters[1], iters[2], iters[0] + iters[1] + iters[2], (double) S / (iters[0] + iters[1] + iters[2]));
    // lock-step cost of a wave of G consecutive blocks: sum over passes of the max over lanes
    for (int G : {8, 16, 32, 64}) {
        double lock_it = 0, lock_sym = 0, sum_sym = 0, max_tot = 0;
        for (int b0 = 0; b0 < n; b0 += G) {
            size_t np = 0; for (int b = b0; b < std::min(n, b0 + G); b++) np = std::max(np, sym[b].size());
            long mt = 0;
            for (int b = b0; b < std::min(n, b0 + G); b++) { long t = 0; for (auto v : sym[b]) t += v; mt = std::max(mt, t); }
            max_tot += mt;
            for (size_t p = 0; p < np; p++) {
                long mi = 0, ms = 0;
                for (int b = b0; b < std::min(n, b0 + G); b++) if (p < sym[b].size()) { mi = std::max(mi, it[b][p]); ms = std::max(ms, sym[b][p]); sum_sym += sym[b][p]; }
                lock_it += mi; lock_sym += ms;
            }
        }
        printf("G=%2d: waves %d, pass-locked iterations/wave %.0f, pass-locked symbols/wave %.0f, free-running max symbols/wave %.0f, mean symbols/lane %.0f\n",
               G, (n + G - 1) / G, lock_it / ((n + G - 1) / G), lock_sym / ((n + G - 1) / G), max_tot / ((n + G - 1) / G), sum_sym / n);
    }
    // slot-locked (pass, stripe) cost: blocks of a wave have the same number of stripes mostly; align by slot index
    for (int G : {32, 64}) {
        double tot_lock = 0; int waves = 0;
        for (int b0 = 0; b0 < n; b0 += G, waves++) {
            size_t ns = 0; for (int b = b0; b < std::min(n, b0 + G); b++) ns = std::max(ns, marks[b].size());
            for (size_t k = 0; k + 1 < ns; k++) {
                long m = 0;
                for (int b = b0; b < std::min(n, b0 + G); b++) if (k + 1 < marks[b].size()) m = std::max(m, marks[b][k + 1] - marks[b][k]);
                tot_lock += m;
            }
        }
        printf("G=%d: stripe-locked symbols/wave %.0f\n", G, tot_lock / waves);
    }
    return 0;
}
