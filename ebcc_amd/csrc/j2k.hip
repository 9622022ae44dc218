// j2k.hip - base-layer buffers of a context, host-side codestream parsing for the decode path and the
// unit-level C-ABI entry points of the JPEG 2000 layer (include/ebcc_hip.h).
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/ebcc_hip.h"
#include "engine.hpp"
#include "j2k.hpp"

namespace ebcc {

bool j2k_create(ebcc_hip_ctx *ctx)
{
    J2kBuffers *jb = new J2kBuffers();
    ctx->j2k = jb;
    // one geometry per tile position (j2k.hpp); plain frames: one
    const int period = ctx->tile_period;
    std::vector<std::vector<J2kBlock>> tile_blocks((size_t) period);
    int stride = 0;
    for (int k = 0; k < period; k++) {
        jb->geoms.push_back(make_j2k_geom(ctx->height, ctx->width, tile_blocks[(size_t) k], k * ctx->height));
        stride = std::max(stride, jb->geoms.back().nblocks);
    }
    for (J2kGeom &t : jb->geoms) { t.period = period; t.stride = stride; }
    jb->geom = jb->geoms[0];
    const J2kGeom &g = jb->geom;
    const size_t n_pix = ctx->n_pix, F = ctx->max_frames;
    const size_t total = F * (size_t) stride, groups = (total + 63) / 64;
    jb->max_frames = (int) F;
    jb->fs = ctx->rb.fs;
    // the largest codestream keeps every pass: bounded by the code-block slots; 2 bytes/sample is far above
    // anything the 9/7 + MQ coder emits for 16-bit data, and the writer never exceeds the slots it copies from
    jb->stream_cap = ((n_pix * 3 + 4096 + 63) / 64) * 64;
    bool ok = true;
    ok &= (jb->d_geom = (J2kGeom *) ctx_alloc<uint8_t>(ctx, sizeof(J2kGeom) * (size_t) period)) != nullptr;
    ok &= (jb->d_blocks = (J2kBlock *) ctx_alloc<uint8_t>(ctx, sizeof(J2kBlock) * (size_t) period * (size_t) stride)) != nullptr;
    ok &= (jb->d_blkmap = ctx_alloc<uint16_t>(ctx, (size_t) period * n_pix)) != nullptr;
    ok &= (jb->B = ctx_alloc<float>(ctx, F * n_pix)) != nullptr;
    ok &= (jb->B2 = ctx_alloc<float>(ctx, F * n_pix)) != nullptr;
    ok &= (jb->Q6 = ctx_alloc<int32_t>(ctx, F * n_pix)) != nullptr;
    ok &= (jb->DEC = ctx_alloc<float>(ctx, F * n_pix)) != nullptr;
    ok &= (jb->V = ctx_alloc<int32_t>(ctx, F * n_pix)) != nullptr;
    ok &= (jb->BP = ctx_alloc<unsigned long long>(ctx, groups * kJ2kMaxPlanes * 64 * 64)) != nullptr;
    ok &= (jb->SGN = ctx_alloc<unsigned long long>(ctx, groups * 64 * 64)) != nullptr;
    ok &= (jb->SUF = ctx_alloc<unsigned long long>(ctx, groups * (kJ2kMaxPlanes + 2) * 64 * 64)) != nullptr;
    ok &= (jb->SPS = ctx_alloc<unsigned long long>(ctx, groups * 64 * 64)) != nullptr;
    ok &= (jb->VISP = ctx_alloc<unsigned long long>(ctx, groups * kJ2kMaxPlanes * 64 * 64)) != nullptr;
    ok &= (jb->ckpt = ctx_alloc<uint8_t>(ctx, groups * 64 * j2k_ckpt_block_bytes())) != nullptr;
    // decision streams of the two-phase tier-1 encoder (default; EBCC_T1_TWO_PHASE=0 selects the single-kernel encoder
    // and saves this buffer)
    jb->SYM = nullptr;
    {
        const char *e = getenv("EBCC_T1_TWO_PHASE");
        jb->sym_rows = kJ2kSymRows;
        if (const char *r = getenv("EBCC_HIP_SYM_ROWS")) jb->sym_rows = std::max(1, std::min(kJ2kSymRows, atoi(r)));
        if (!e || atoi(e)) ok &= (jb->SYM = ctx_alloc<uint8_t>(ctx, groups * (size_t) ((jb->sym_rows + 1) & ~1) * 1024 + 256)) != nullptr;
    }
    ok &= (jb->seglen = ctx_alloc<uint16_t>(ctx, groups * (size_t) kJ2kSegCount * 64)) != nullptr;
    ok &= (jb->lanerows = ctx_alloc<uint32_t>(ctx, groups * 64)) != nullptr;
    ok &= (jb->qplane = (int *) ctx_alloc<int32_t>(ctx, groups * 64)) != nullptr;
    ok &= (jb->lastnp = (int *) ctx_alloc<int32_t>(ctx, groups * 64)) != nullptr;
    ok &= (jb->T1S = ctx_alloc<unsigned long long>(ctx, groups * kT1StateWords * 64)) != nullptr;
    ok &= (jb->blkmax = (int *) ctx_alloc<int32_t>(ctx, groups * 64)) != nullptr;
    ok &= (jb->numbps = (int *) ctx_alloc<int32_t>(ctx, groups * 64)) != nullptr;
    ok &= (jb->totalpasses = (int *) ctx_alloc<int32_t>(ctx, groups * 64)) != nullptr;
    ok &= (jb->cblk_len = (int *) ctx_alloc<int32_t>(ctx, groups * 64)) != nullptr;
    ok &= (jb->npass = (int *) ctx_alloc<int32_t>(ctx, groups * 64)) != nullptr;
    ok &= (jb->cand_cr = ctx_alloc<float>(ctx, F * 2)) != nullptr;
    ok &= (jb->cand_out = (int *) ctx_alloc<int32_t>(ctx, F * 6)) != nullptr;
    ok &= (jb->cand_npass = (int *) ctx_alloc<int32_t>(ctx, 2 * F * (size_t) stride)) != nullptr;
    ok &= (jb->cand_sel = (int *) ctx_alloc<int32_t>(ctx, F)) != nullptr;
    ok &= (jb->have_rate = (int *) ctx_alloc<int32_t>(ctx, F)) != nullptr;
    ok &= (jb->rate_path = (int *) ctx_alloc<int32_t>(ctx, F * 1536 * 3)) != nullptr;
    ok &= (jb->rate_path_n = (int *) ctx_alloc<int32_t>(ctx, 2 * F)) != nullptr;          // (second half: RateCache::ok, reset together)
    {
        int nodes = 0;
        for (const J2kGeom &t : jb->geoms) nodes = std::max(nodes, t.tree_nodes);
        J2kBuffers::RateCache &rc = jb->rate_cache;
        rc.nodes_cap = (nodes + 3) & ~3;
        rc.cap = stride * kJ2kMaxPasses;
        rc.ok = jb->rate_path_n ? jb->rate_path_n + F : nullptr;
        ok &= (rc.mnmx = ctx_alloc<double>(ctx, F * 2)) != nullptr;
        ok &= (rc.mval = (short *) ctx_alloc<uint16_t>(ctx, F * (size_t) rc.nodes_cap)) != nullptr;
        ok &= (rc.off = (int *) ctx_alloc<int32_t>(ctx, F * (size_t) (stride + 1))) != nullptr;
        ok &= (rc.crate = ctx_alloc<uint16_t>(ctx, F * (size_t) rc.cap)) != nullptr;
        ok &= (rc.cdisto = ctx_alloc<double>(ctx, F * (size_t) rc.cap)) != nullptr;
    }
    ok &= (jb->rates = (int *) ctx_alloc<int32_t>(ctx, groups * 64 * kJ2kMaxPasses)) != nullptr;
    ok &= (jb->disto = ctx_alloc<double>(ctx, groups * 64 * kJ2kMaxPasses)) != nullptr;
    ok &= (jb->cblk_bytes = ctx_alloc<uint8_t>(ctx, groups * 64 * kJ2kCblkBytes)) != nullptr;
    ok &= (jb->stream = ctx_alloc<uint8_t>(ctx, F * jb->stream_cap)) != nullptr;
    ok &= (jb->dec_table = (int *) ctx_alloc<int32_t>(ctx, groups * 64 * 4)) != nullptr;
    ok &= (jb->dec_order = (int *) ctx_alloc<int32_t>(ctx, groups * 64 + 128)) != nullptr;
    ok &= (jb->mq_order = (int *) ctx_alloc<int32_t>(ctx, groups * 64 + 256)) != nullptr;
    ok &= (jb->jf = (J2kFrame *) ctx_alloc<uint8_t>(ctx, sizeof(J2kFrame) * F)) != nullptr;
    ok &= (jb->partial = ctx_alloc<double>(ctx, F * kPartials)) != nullptr;
    ok &= (jb->partial_u = ctx_alloc<unsigned long long>(ctx, F * kPartials)) != nullptr;
    if (!ok) return false;
    if (stride >= 65535) { set_error("too many code-blocks"); return false; }
    std::vector<uint16_t> map((size_t) period * n_pix, 0);
    std::vector<J2kBlock> blocks((size_t) period * (size_t) stride, J2kBlock{});
    for (int k = 0; k < period; k++) {
        const std::vector<J2kBlock> &tb = tile_blocks[(size_t) k];
        std::copy(tb.begin(), tb.end(), blocks.begin() + (size_t) k * (size_t) stride);
        for (size_t b = 0; b < tb.size(); b++)
            for (int y = 0; y < tb[b].h; y++)
                for (int x = 0; x < tb[b].w; x++) map[(size_t) k * n_pix + (size_t) (tb[b].y + y) * g.W + tb[b].x + x] = (uint16_t) b;
    }
    hipStream_t s = ctx->stream;
    EBCC_HIP_CHECK(hipMemcpyAsync(jb->d_geom, jb->geoms.data(), sizeof(J2kGeom) * (size_t) period, hipMemcpyHostToDevice, s));
    EBCC_HIP_CHECK(hipMemcpyAsync(jb->d_blocks, blocks.data(), sizeof(J2kBlock) * blocks.size(), hipMemcpyHostToDevice, s));
    EBCC_HIP_CHECK(hipMemcpyAsync(jb->d_blkmap, map.data(), map.size() * sizeof(uint16_t), hipMemcpyHostToDevice, s));
    EBCC_HIP_CHECK(hipHostMalloc(&ctx->h_jf, sizeof(J2kFrame) * F));
    EBCC_HIP_CHECK(hipHostMalloc((void **) &ctx->h_table, sizeof(int) * 4 * total));
    EBCC_HIP_CHECK(hipMemsetAsync(jb->jf, 0, sizeof(J2kFrame) * F, s));
    EBCC_HIP_CHECK(hipMemsetAsync(jb->rate_path_n, 0, sizeof(int) * 2 * F, s));
    wait_stream(s);
    return true;
}

void j2k_destroy(ebcc_hip_ctx *ctx)
{
    delete static_cast<J2kBuffers *>(ctx->j2k);
    ctx->j2k = nullptr;
    if (ctx->stream2) { hipStreamSynchronize(ctx->stream2); hipStreamDestroy(ctx->stream2); ctx->stream2 = nullptr; }
}

// ================================================================================================
// host-side codestream parsing (main header + tier-2 packet headers, T.800 Annex A / B.10) for the
// decode path: fills {offset, length, numbps, npasses} per code-block.  Serial and tiny (~300
// code-blocks per frame); tier-1 and everything per-sample runs on the device.
// ================================================================================================
namespace {

struct BitReader {
    const uint8_t *p, *end;
    unsigned buf = 0;
    int ct = 0;
    int bit()
    {
        if (ct == 0) {
            buf = (buf << 8) & 0xFFFF;
            ct = buf == 0xFF00 ? 7 : 8;
            if (p < end) buf |= *p++;
        }
        ct--;
        return (buf >> ct) & 1;
    }
    int bits(int n) { int v = 0; for (int i = n - 1; i >= 0; i--) v |= bit() << i; return v; }
    void align() { if ((buf & 0xFF) == 0xFF && p < end) p++; ct = 0; }
};

struct HostTree {
    const J2kBand *bd;
    std::vector<int> val, low;
    explicit HostTree(const J2kBand &b) : bd(&b)
    {
        int n = 0;
        for (int l = 0; l < b.tree_levels; l++) n += b.lvl_w[l] * b.lvl_h[l];
        val.assign(n, 999); low.assign(n, 0);
    }
    bool decode(BitReader &br, int cx, int cy, int threshold)
    {
        int lowv = 0, idx = 0;
        for (int l = bd->tree_levels - 1; l >= 0; l--) {
            idx = bd->lvl_off[l] + (cy >> l) * bd->lvl_w[l] + (cx >> l);
            if (lowv > low[idx]) low[idx] = lowv; else lowv = low[idx];
            while (lowv < threshold && lowv < val[idx]) {
                if (br.bit()) val[idx] = lowv; else ++lowv;
            }
            low[idx] = lowv;
        }
        return val[idx] < threshold;
    }
};

unsigned be16(const uint8_t *p) { return ((unsigned) p[0] << 8) | p[1]; }
unsigned be32(const uint8_t *p) { return ((unsigned) p[0] << 24) | ((unsigned) p[1] << 16) | ((unsigned) p[2] << 8) | p[3]; }
int flog2(int a) { int l = 0; while (a > 1) { a >>= 1; l++; } return l; }

}  // namespace

// returns false on a malformed / unsupported codestream
namespace {

struct MainHeader {
    int W = 0, H = 0, tile_w = 0, tile_h = 0, nres = 0, qsty = -1, guard = 0;
    int expn[kJ2kBands] = {0}, mant[kJ2kBands] = {0};
    size_t first_sot = 0;          // position of the first SOT marker
};

// main header up to the first tile-part (A.5, A.6)
bool parse_main_header(const uint8_t *cs, size_t n, MainHeader &h)
{
    if (n < 4 || be16(cs) != 0xFF4F) { set_error("J2K: missing SOC"); return false; }
    size_t pos = 2;
    while (pos + 4 <= n) {
        const unsigned mk = be16(cs + pos), len = be16(cs + pos + 2);
        const uint8_t *p = cs + pos + 4;
        if (mk == 0xFF90) { h.first_sot = pos; return true; }
        // a segment is its length field (2 bytes) + payload, all inside the stream; the fixed-offset reads below stay
        // inside the payload (A.5.1: SIZ has 36 bytes + 3 per component, A.6.1: COD 10 or more, A.6.4: QCD 1 + 2 per band)
        if (len < 2 || pos + 2 + (size_t) len > n) { set_error("J2K: truncated marker segment"); return false; }
        if (mk == 0xFF51) {
            if (len < 38) { set_error("J2K: SIZ segment too short"); return false; }
            h.W = (int) (be32(p + 2) - be32(p + 10)); h.H = (int) (be32(p + 6) - be32(p + 14));
            h.tile_w = (int) be32(p + 18); h.tile_h = (int) be32(p + 22);
        } else if (mk == 0xFF52) {
            if (len < 8) { set_error("J2K: COD segment too short"); return false; }
            h.nres = p[5] + 1;
        } else if (mk == 0xFF5C) {
            if (len < 3) { set_error("J2K: QCD segment too short"); return false; }
            h.qsty = p[0] & 0x1F; h.guard = p[0] >> 5;
            const int nb = (int) (len - 3) / 2;
            for (int i = 0; i < nb && i < kJ2kBands; i++) { unsigned v = be16(p + 1 + 2 * i); h.expn[i] = (int) (v >> 11); h.mant[i] = (int) (v & 0x7FF); }
        }
        pos += 2 + len;
    }
    set_error("J2K: no tile-part");
    return false;
}

bool header_matches(const MainHeader &h, const J2kGeom &g)
{
    if (h.nres != kJ2kRes || h.qsty != 2 || h.guard != 2) return false;
    for (int b = 0; b < g.nbands; b++)
        if (h.expn[b] != g.bands[b].expn || h.mant[b] != g.bands[b].mant) return false;
    return true;
}

// One tile-part starting at the SOT marker at cs + sot: packet headers -> table [nblocks][4] = {offset of the
// code-block's bytes relative to `origin`, length, numbps, passes}.  Returns the position after the tile-part.
bool parse_tile_part(const uint8_t *cs, size_t n, size_t sot, const uint8_t *origin, const J2kGeom &g, int *table, int *isot,
                     size_t *next)
{
    if (sot + 12 + 2 > n || be16(cs + sot) != 0xFF90) { set_error("J2K: missing SOT"); return false; }
    const uint8_t *p = cs + sot + 4;
    *isot = (int) be16(p);
    const unsigned psot = be32(p + 2);
    if (p[6] != 0) { set_error("J2K: tiles split into several tile-parts are not supported"); return false; }
    size_t pos = sot + 2 + be16(cs + sot + 2);
    if (pos + 2 > n || be16(cs + pos) != 0xFF93) { set_error("J2K: missing SOD"); return false; }
    const uint8_t *tile_data = cs + pos + 2;
    const uint8_t *tile_end = psot ? cs + sot + psot : cs + n - 2;
    if (tile_end > cs + n || tile_end < tile_data) { set_error("J2K: tile-part overruns the stream"); return false; }
    std::memset(table, 0, sizeof(int) * 4 * (size_t) g.stride);
    p = tile_data;
    for (int r = 0; r < kJ2kRes; r++) {
        BitReader br{p, tile_end};
        std::vector<int> included;
        if (br.bit()) {
            for (int bi = 0; bi < g.nbands; bi++) {
                const J2kBand &bd = g.bands[bi];
                if (bd.res != r || bd.ncw * bd.nch == 0) continue;
                HostTree incl(bd), imsb(bd);
                for (int cy = 0; cy < bd.nch; cy++)
                    for (int cx = 0; cx < bd.ncw; cx++) {
                        if (!incl.decode(br, cx, cy, 1)) continue;
                        int blk = bd.first_block + cy * bd.ncw + cx;
                        // everything read from the packet header is checked before it is used as a count, a shift or
                        // an offset: the tables steer device kernels (OpenJPEG does the same checks for the reference)
                        int i = 1;
                        while (!imsb.decode(br, cx, cy, i)) { if (++i > bd.numbps + 1) { set_error("J2K: bad zero-bit-plane count"); return false; } }
                        const int planes = bd.numbps + 1 - i;
                        if (planes < 1 || planes > kJ2kMaxPlanes) { set_error("J2K: code-block with %d bit-planes", planes); return false; }
                        int np;
                        if (!br.bit()) np = 1;
                        else if (!br.bit()) np = 2;
                        else { int v = br.bits(2); if (v != 3) np = 3 + v; else { v = br.bits(5); np = v != 31 ? 6 + v : 37 + br.bits(7); } }
                        if (np > 3 * planes - 2) { set_error("J2K: %d coding passes for %d bit-planes", np, planes); return false; }
                        int lblock = 3;
                        while (br.bit()) { if (++lblock > 24) { set_error("J2K: bad Lblock"); return false; } }
                        const int lenbits = lblock + flog2(np);
                        if (lenbits > 30) { set_error("J2K: bad segment length field"); return false; }
                        int len = br.bits(lenbits);
                        if (len < 0 || (size_t) len > (size_t) (tile_end - tile_data)) { set_error("J2K: segment longer than the tile-part"); return false; }
                        table[4 * blk + 1] = len;
                        table[4 * blk + 2] = planes;
                        table[4 * blk + 3] = np;
                        included.push_back(blk);
                    }
            }
        }
        br.align();
        p = br.p;
        for (int blk : included) {
            if (p > tile_end || table[4 * blk + 1] > tile_end - p) { set_error("J2K: packet body overruns the tile-part"); return false; }
            table[4 * blk + 0] = (int) (p - origin);
            p += table[4 * blk + 1];
        }
    }
    *next = (size_t) (tile_end - cs);
    return true;
}

}  // namespace

bool j2k_parse_codestream(const uint8_t *cs, size_t n, const J2kGeom &g, int *table /* [nblocks][4] */)
{
    MainHeader h;
    if (!parse_main_header(cs, n, h)) return false;
    if (h.W != g.W || h.H != g.H || !header_matches(h, g)) {
        set_error("J2K: codestream (%dx%d, %d resolutions, qsty %d) does not match the context (%dx%d)", h.W, h.H, h.nres, h.qsty, g.W, g.H);
        return false;
    }
    int isot;
    size_t next;
    return parse_tile_part(cs, n, h.first_sot, cs, g, table, &isot, &next);      // offsets relative to the whole stream
}

// Image / tile extents of a codestream (A.5.1): false if it has no usable main header
bool j2k_peek_dims(const uint8_t *cs, size_t n, int *W, int *H, int *tile_w, int *tile_h)
{
    MainHeader h;
    if (!parse_main_header(cs, n, h)) return false;
    *W = h.W; *H = h.H; *tile_w = h.tile_w; *tile_h = h.tile_h;
    return true;
}

// A codestream of `tiles` tiles of the context's frame size stacked along y (what the reference writes for a
// chunk of several frames): tables[t] as above with offsets relative to the tile-part's SOT, and the extent of
// every tile-part inside cs.  Tile t is parsed with the context's geometry for that tile position.
bool j2k_parse_tiled(const uint8_t *cs, size_t n, const J2kBuffers &jb, int tiles, int *tables /* [tiles][stride][4] */,
                     size_t *part_off, size_t *part_len)
{
    const J2kGeom &g = jb.geom;
    if (g.period > 1 && g.period != tiles) { set_error("J2K: the context holds %d tile positions, the chunk %d", g.period, tiles); return false; }
    MainHeader h;
    if (!parse_main_header(cs, n, h)) return false;
    if (h.W != g.W || h.H != g.H * tiles || h.tile_w != g.W || h.tile_h != g.H || !header_matches(h, g)) {
        set_error("J2K: codestream (%dx%d in tiles of %dx%d) does not match %d tiles of %dx%d", h.W, h.H, h.tile_w, h.tile_h, tiles, g.W, g.H);
        return false;
    }
    std::vector<char> seen((size_t) tiles, 0);
    size_t pos = h.first_sot;
    for (int k = 0; k < tiles; k++) {
        int isot;
        size_t next;
        if (pos + 12 > n || be16(cs + pos) != 0xFF90) { set_error("J2K: fewer tile-parts than tiles"); return false; }
        const int peek = (int) be16(cs + pos + 4);
        if (peek < 0 || peek >= tiles || seen[(size_t) peek]) { set_error("J2K: bad tile index %d", peek); return false; }
        if (!parse_tile_part(cs, n, pos, cs + pos, jb.geoms[(size_t) j2k_geom_index(&g, peek)], tables + (size_t) peek * g.stride * 4, &isot, &next)) return false;
        seen[(size_t) peek] = 1;
        part_off[peek] = pos;
        part_len[peek] = next - pos;
        pos = next;
    }
    return true;
}

}  // namespace ebcc

using namespace ebcc;

extern "C" {

// Host-only check of the codestream parser (no device work): parses `cs` as a one-tile codestream of height x width
// and verifies that every code-block entry it would hand to the device kernels lies inside the stream.  0 = accepted,
// 1 = rejected (message in ebcc_hip_last_error), 2 = ACCEPTED WITH AN ENTRY OUT OF BOUNDS (a parser bug).
__attribute__((visibility("default"))) int ebcc_hip_j2k_parse_check(const uint8_t *cs, size_t n, size_t height, size_t width)
{
    EBCC_API_TRY
    if (height < 1 || width < 1 || height > 2047 || width > 2047) { set_error("bad geometry"); return 1; }
    std::vector<J2kBlock> blocks;
    J2kGeom g = make_j2k_geom((int) height, (int) width, blocks);
    g.period = 1; g.stride = g.nblocks;
    std::vector<int> table((size_t) g.stride * 4, 0);
    if (!j2k_parse_codestream(cs, n, g, table.data())) return 1;
    for (int b = 0; b < g.nblocks; b++) {
        const int off = table[4 * b], len = table[4 * b + 1], planes = table[4 * b + 2], np = table[4 * b + 3];
        if (np == 0 && len == 0) continue;
        if (off < 0 || len < 0 || (size_t) off + (size_t) len > n || planes < 1 || planes > kJ2kMaxPlanes || np < 1 || np > 3 * planes - 2) {
            set_error("parser accepted code-block %d with offset %d length %d planes %d passes %d", b, off, len, planes, np);
            return 2;
        }
    }
    return 0;
    EBCC_API_CATCH(1)
}

// ---- unit-level entry points of the base layer (parity tests) -------------------------------------
// j2k_encode_internal (src/ebcc_codec.c:105-180) for a batch: frames are scaled to u16 with their own
// min/max as ebcc_encode does (:675-689), then coded at rate cr[f].
__attribute__((visibility("default"))) int ebcc_hip_j2k_encode(ebcc_hip_ctx *ctx, const float *d_frames, size_t n_frames,
                                                               const float *cr, uint8_t **out_streams, size_t *out_sizes,
                                                               float *minmax)
{
    EBCC_API_TRY
    if (!ctx || n_frames < 1 || n_frames > ctx->max_frames) { set_error("ebcc_hip_j2k_encode: bad batch"); return 1; }
    EBCC_HIP_CHECK(hipSetDevice(ctx->device));
    J2kBuffers &jb = *static_cast<J2kBuffers *>(ctx->j2k);
    hipStream_t s = ctx->stream;
    const int n = (int) n_frames;
    launch_input_stats(d_frames, n, ctx->n_pix, ctx->rb.fs, s);
    launch_j2k_analysis(d_frames, jb, n, s);
    std::vector<J2kFrame> jf(n_frames);
    EBCC_HIP_CHECK(hipMemcpyAsync(jf.data(), jb.jf, sizeof(J2kFrame) * n_frames, hipMemcpyDeviceToHost, s));
    wait_stream(s);
    if (j2k_tier1_retry(jb, n, jf.data(), s)) {
        EBCC_HIP_CHECK(hipMemcpyAsync(jf.data(), jb.jf, sizeof(J2kFrame) * n_frames, hipMemcpyDeviceToHost, s));
        wait_stream(s);
    }
    for (size_t f = 0; f < n_frames; f++) { jf[f].cr = cr[f]; jf[f].target = 0; }
    EBCC_HIP_CHECK(hipMemcpyAsync(jb.jf, jf.data(), sizeof(J2kFrame) * n_frames, hipMemcpyHostToDevice, s));
    launch_j2k_rate(jb, n, nullptr, s);
    launch_j2k_write(jb, n, nullptr, s);
    EBCC_HIP_CHECK(hipMemcpyAsync(jf.data(), jb.jf, sizeof(J2kFrame) * n_frames, hipMemcpyDeviceToHost, s));
    fetch_frame_states(ctx, n_frames);
    for (size_t f = 0; f < n_frames; f++) {
        if (ctx->h_fs[f].const_field) { out_sizes[f] = 0; out_streams[f] = (uint8_t *) malloc(1); continue; }
        if (jf[f].overflow) { set_error("code-block byte slot overflow"); return 1; }
        out_sizes[f] = (size_t) jf[f].stream_bytes;
        out_streams[f] = (uint8_t *) malloc(out_sizes[f]);
        EBCC_HIP_CHECK(hipMemcpyAsync(out_streams[f], jb.stream + f * jb.stream_cap, out_sizes[f], hipMemcpyDeviceToHost, s));
        if (minmax) { minmax[2 * f] = ctx->h_fs[f].minv; minmax[2 * f + 1] = ctx->h_fs[f].maxv; }
    }
    wait_stream(s);
    return 0;
    EBCC_API_CATCH(1)
}

// After ebcc_hip_j2k_encode: the field j2k_decode_internal (:1092-1136) would return for those streams,
// decoded in place from the encoder's code-block slots; nbad[f] = count(|x - d| > target[f]).
__attribute__((visibility("default"))) int ebcc_hip_j2k_emulated_decode(ebcc_hip_ctx *ctx, const float *d_frames, size_t n_frames,
                                                                         const float *target, float *d_out,
                                                                         unsigned long long *nbad, double *err_sum)
{
    EBCC_API_TRY
    if (!ctx || n_frames < 1 || n_frames > ctx->max_frames) { set_error("ebcc_hip_j2k_emulated_decode: bad batch"); return 1; }
    EBCC_HIP_CHECK(hipSetDevice(ctx->device));
    J2kBuffers &jb = *static_cast<J2kBuffers *>(ctx->j2k);
    hipStream_t s = ctx->stream;
    std::vector<J2kFrame> jf(n_frames);
    EBCC_HIP_CHECK(hipMemcpyAsync(jf.data(), jb.jf, sizeof(J2kFrame) * n_frames, hipMemcpyDeviceToHost, s));
    wait_stream(s);
    for (size_t f = 0; f < n_frames; f++) jf[f].target = target[f];
    EBCC_HIP_CHECK(hipMemcpyAsync(jb.jf, jf.data(), sizeof(J2kFrame) * n_frames, hipMemcpyHostToDevice, s));
    launch_j2k_probe_decode(d_frames, jb, (int) n_frames, nullptr, s);
    EBCC_HIP_CHECK(hipMemcpyAsync(d_out, jb.DEC, n_frames * ctx->n_pix * sizeof(float), hipMemcpyDeviceToDevice, s));
    EBCC_HIP_CHECK(hipMemcpyAsync(jf.data(), jb.jf, sizeof(J2kFrame) * n_frames, hipMemcpyDeviceToHost, s));
    wait_stream(s);
    for (size_t f = 0; f < n_frames; f++) { nbad[f] = jf[f].nbad; err_sum[f] = jf[f].err_sum; }
    return 0;
    EBCC_API_CATCH(1)
}

// j2k_decode_internal for a batch of codestreams with the given (minval, maxval)
__attribute__((visibility("default"))) int ebcc_hip_j2k_decode(ebcc_hip_ctx *ctx, const uint8_t *const *streams, const size_t *sizes,
                                                               size_t n_frames, const float *minmax, float *d_out)
{
    EBCC_API_TRY
    if (!ctx || n_frames < 1 || n_frames > ctx->max_frames) { set_error("ebcc_hip_j2k_decode: bad batch"); return 1; }
    EBCC_HIP_CHECK(hipSetDevice(ctx->device));
    J2kBuffers &jb = *static_cast<J2kBuffers *>(ctx->j2k);
    hipStream_t s = ctx->stream;
    const J2kGeom &g = jb.geom;
    std::vector<int> table((size_t) n_frames * g.stride * 4);
    fetch_frame_states(ctx, n_frames);
    for (size_t f = 0; f < n_frames; f++) {
        if (sizes[f] > jb.stream_cap) { set_error("codestream larger than the slot"); return 1; }
        if (!j2k_parse_codestream(streams[f], sizes[f], g, table.data() + f * g.stride * 4)) return 1;
        EBCC_HIP_CHECK(hipMemcpyAsync(jb.stream + f * jb.stream_cap, streams[f], sizes[f], hipMemcpyHostToDevice, s));
        ctx->h_fs[f].minv = minmax[2 * f];
        ctx->h_fs[f].maxv = minmax[2 * f + 1];
        ctx->h_fs[f].const_field = 0;
    }
    push_frame_states(ctx, n_frames);
    EBCC_HIP_CHECK(hipMemcpyAsync(jb.dec_table, table.data(), table.size() * sizeof(int), hipMemcpyHostToDevice, s));
    launch_j2k_decode(jb, (int) n_frames, s, table.data());
    EBCC_HIP_CHECK(hipMemcpyAsync(d_out, jb.DEC, n_frames * ctx->n_pix * sizeof(float), hipMemcpyDeviceToDevice, s));
    wait_stream(s);
    return 0;
    EBCC_API_CATCH(1)
}

}  // extern "C"
