// Work statistics of the tier-1 DECODER on the code-block segments of a real codestream (development tool, not part of the
// product or the tests): decisions by kind and pass, column iterations of the pass loops, per code-block.
//   g++ -O2 -std=c++17 -o /tmp/t1ds tools/t1_decode_stats.cpp -Loracle -lebcc_oracle -Wl,-rpath,$PWD/oracle && /tmp/t1ds stream.j2k
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
void orc_j2k_set_decode_sink(void (*fn)(const uint8_t *, int, int, int, int, int, int, void *), void *user);
size_t orc_j2k_decode(const uint8_t *cs, size_t n, int32_t **samples, size_t *height, size_t *width);
}
using namespace ebcc::t1;
struct Seg { std::vector<uint8_t> d; int P, np, w, h, orient; };
static std::vector<Seg> segs;
static void sink(const uint8_t *d, int len, int P, int np, int w, int h, int orient, void *) { segs.push_back(Seg{std::vector<uint8_t>(d, d + len), P, np, w, h, orient}); }

struct HostDecStore {
    u64 s[66], neg[64], vis[64], ref[64];
    HostDecStore() { memset(this, 0, sizeof *this); }
    u64 &S(int y) { return s[y + 1]; } u64 &NEG(int y) { return neg[y]; } u64 &VIS(int y) { return vis[y]; } u64 &REF(int y) { return ref[y]; }
    void set_sig(int, int, int, int) {}
    void refine(int, int, int, int, int) {}
};
struct BufSrc { const uint8_t *p; int n; uint32_t get(int i) const { return i < n ? p[i] : 0xFFu; } };
static long g_kind[6];                       // zc, sc, mr, agg, uni, agg-run decisions taken in bulk
struct CountingDecoder : MqDecoder<BufSrc> {
    int decode_zc(int c) { g_kind[0]++; return MqDecoder<BufSrc>::decode_zc(c); }
    int decode_sc(int c) { g_kind[1]++; return MqDecoder<BufSrc>::decode_sc(c); }
    int decode_mag(int c) { g_kind[2]++; return MqDecoder<BufSrc>::decode_mag(c); }
    int decode_agg() { g_kind[3]++; return MqDecoder<BufSrc>::decode_agg(); }
    int decode_uni() { g_kind[4]++; return MqDecoder<BufSrc>::decode_uni(); }
    int agg_zero_run(int n) { const int k = MqDecoder<BufSrc>::agg_zero_run(n); g_kind[5] += k; return k; }
};
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb"); if (!f) return 1;
    std::vector<uint8_t> cs; int c; while ((c = fgetc(f)) != EOF) cs.push_back((uint8_t) c); fclose(f);
    orc_j2k_set_decode_sink(sink, nullptr);
    int32_t *smp = nullptr; size_t H = 0, W = 0;
    if (!orc_j2k_decode(cs.data(), cs.size(), &smp, &H, &W)) { fprintf(stderr, "decode failed\n"); return 1; }
    printf("%zu x %zu, %zu code-block segments, %zu bytes\n", H, W, segs.size(), cs.size());
    long pass_dec[3] = {0, 0, 0}, pass_it[3] = {0, 0, 0}, stripes[3] = {0, 0, 0}, bytes = 0;
    std::vector<std::pair<long, int>> per;
    for (size_t b = 0; b < segs.size(); b++) {
        Seg &S = segs[b];
        if (S.np <= 0 || S.P <= 0) continue;
        bytes += (long) S.d.size();
        HostDecStore st;
        CountingDecoder mq{};
        mq.src = BufSrc{S.d.data(), (int) S.d.size()};
        mq.init();
        Passes<false, HostDecStore, CountingDecoder> ps(st, mq, S.w, S.h, S.orient);
        long before_all = g_kind[0] + g_kind[1] + g_kind[2] + g_kind[3] + g_kind[4] + g_kind[5];
        for (int i = 0; i < S.np; i++) {
            const int bp = plane_of_pass(S.P, i), ty = type_of_pass(i);
            const long d0 = g_kind[0] + g_kind[1] + g_kind[2] + g_kind[3] + g_kind[4] + g_kind[5], i0 = g_stat[0] + g_stat[1] + g_stat[2];
            if (ty == 0) ps.sigprop(bp); else if (ty == 1) ps.refine(bp); else ps.cleanup(bp);
            pass_dec[ty] += g_kind[0] + g_kind[1] + g_kind[2] + g_kind[3] + g_kind[4] + g_kind[5] - d0;
            pass_it[ty] += g_stat[0] + g_stat[1] + g_stat[2] - i0;
            stripes[ty] += (S.h + 3) / 4;
        }
        per.push_back({g_kind[0] + g_kind[1] + g_kind[2] + g_kind[3] + g_kind[4] + g_kind[5] - before_all, (int) S.d.size()});
    }
    const long tot = g_kind[0] + g_kind[1] + g_kind[2] + g_kind[3] + g_kind[4] + g_kind[5];
    printf("decisions: zc %ld sc %ld mr %ld agg %ld uni %ld agg-in-bulk %ld  total %ld (%.2f per sample, %.2f per byte of %ld)\n", g_kind[0], g_kind[1], g_kind[2], g_kind[3],
           g_kind[4], g_kind[5], tot, (double) tot / (H * W), (double) tot / bytes, bytes);
    const char *nm[3] = {"propagation", "refinement", "cleanup"};
    for (int t = 0; t < 3; t++)
        printf("  %-12s decisions %9ld  column iterations %9ld (%.2f decisions per iteration)  stripe visits %ld\n", nm[t], pass_dec[t], pass_it[t],
               pass_it[t] ? (double) pass_dec[t] / pass_it[t] : 0.0, stripes[t]);
    std::sort(per.begin(), per.end());
    printf("per code-block decisions: median %ld, p90 %ld, max %ld (%d bytes)\n", per[per.size() / 2].first, per[per.size() * 9 / 10].first, per.back().first, per.back().second);
    return 0;
}
