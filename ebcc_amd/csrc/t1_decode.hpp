// t1_decode.hpp - JPEG 2000 tier-1 DECODER for SIMT execution: one code-block per lane, all lanes of a wave inside
// the same (bit-plane, coding pass, stripe) SEGMENT, one MQ decision per lane and loop iteration.
//
// The per-sample decoder of t1_core.hpp (Passes<false, ...>) follows the data: which sample is coded next, in
// which pass, with which context depends on the decisions so far, so 64 code-blocks in one wave execute the union
// of 64 different paths.  Here the wave walks the segments in lock-step (planes aligned at the least significant
// one, lanes whose code-block does not have a segment sit it out), which makes the pass type, the stripe loads and
// stores and the loop structure uniform; inside a segment every lane runs the same small state machine:
//     find the next sample the pass codes (mask arithmetic on the stripe's row masks) -> context -> one decision
// with the MQ decoder written without data-dependent branches (C.3.2 as selects, renormalisation by count, a
// 64-bit code register that already holds the next bytes so that BYTEIN is a rare refill, not part of a decision).
// Results are identical to decode_block() (tests/t1_host_check.cpp runs both on the host).
// Replaces the opj_t1_decode_cblk calls behind /root/reference/src/ebcc_codec.c:1116.
#pragma once

#include "t1_core.hpp"

namespace ebcc {
namespace t1 {

// ---------------------------------------------------------------------------------------------------------------
// MQ decoder, look-ahead form.  The standard's register C (32 bits: Chigh | Clow) receives a byte at bits 8..15 (or
// 9..16 after a 0xFF byte) whenever its down-counter reaches 0.  All updates of C are linear (shifts, additions,
// the subtraction of Qe << 16), so a byte may be added EARLIER at a position that many bits lower: here C sits in
// the upper half of a 64-bit word and the bytes due within the next 25..32 shifts are already in the lower half.
// `due` = shifts until the next byte that is not yet in the word would be added by the standard's BYTEIN.
//   Bytes: uint32_t get(int index) returning 0xFF past the end of the segment (C.3.4 end-of-data behaviour)
// ---------------------------------------------------------------------------------------------------------------
struct MqLookahead {
    u64 c;
    uint32_t a;
    int due, pos;
    template <class Bytes>
    T1_HD void refill(Bytes &src)
    {
        // BYTEIN (C.3.4) for every byte due within the next 32 shifts; a byte due in `due` shifts with offset s (8, or 9
        // after 0xFF) lands at bit 32 + s - due of the word
        while (due <= 24) {
            const uint32_t cur = src.get(pos), nxt = src.get(pos + 1);
            u64 add;
            int width;
            if (cur == 0xFFu) {
                if (nxt > 0x8Fu) { add = 0xFF00ull; width = 8; }
                else { pos++; add = (u64) nxt << 9; width = 7; }
            } else {
                pos++; add = (u64) nxt << 8; width = 8;
            }
            c += add << (32 - due);
            due += width;
        }
    }
    template <class Bytes>
    T1_HD void init(Bytes &src)
    {
        pos = 0;
        c = (u64) (src.get(0) << 16) << 32;
        due = 0;
        refill(src);
        c <<= 7; due -= 7;
        a = 0x8000;
    }
    // one decision in the context whose state code (t1_core.hpp: mq_code) is `code`; returns the decision, `code` becomes the new state
    template <class Table>
    T1_HD uint32_t decode(uint32_t &code, const Table &tab)
    {
        uint32_t qe, nxt;
        tab(code, qe, nxt);
        const uint32_t mps = (code >> 9) & 1u;
        const uint32_t chigh = (uint32_t) (c >> 48);
        const bool low = chigh < qe;                                     // the decision lies in the lower (LPS) sub-interval
        const uint32_t a1 = a - qe;
        const bool small = a1 < qe;
        const bool ren = low || (a1 & 0x8000u) == 0;
        const bool flip = low ? !small : small;                          // conditional exchange (C.3.2, figures C.16 / C.17)
        const uint32_t lps = (ren && flip) ? 1u : 0u;
        c -= low ? 0ull : (u64) qe << 48;
        const uint32_t a2 = low ? qe : a1;
        code = ren ? (lps ? nxt >> 16 : nxt & 0xFFFFu) : code;
        const int k = renorm_shifts(a2);                                 // 0 when no renormalisation is due
        a = a2 << k;
        c <<= k;
        due -= k;
        return mps ^ lps;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// Segment decoder.  DStore: the decoder's block state and output
//   u64 &S(int y) (y in [-1, 64]), &NEG(int y), &VIS(int y), &REF(int y)
//   void set_sig(int x, int y, int neg, int plane), void refine(int x, int y, int bit, int plane, int neg)
// Ctx: uint32_t handle(ctx), ld(handle), void st(handle, code) as for the row coder (state codes)
// Env: bool any(bool) (true if the argument is true for any lane of the wave), Bytes &bytes(), bool starved(int pos):
//      the byte source cannot serve position pos (+1) yet - the lane then waits for a refill of its ring (device) -
//      void refill_point(int pos) called by all lanes when some lane starved (uniform; pos = the lane's byte position:
//      ring bytes before it may be overwritten), void pass_point(int pos) at the start of every coding pass (uniform: a
//      good moment to top the byte rings up)
// ---------------------------------------------------------------------------------------------------------------
template <class DStore, class Ctx, class Table, class Env>
struct SegmentDecoder {
    DStore &st;
    Ctx &cx;
    const Table &tab;
    Env &env;
    MqLookahead mq;
    int w, h, orient;

    T1_HD static uint32_t bit(u64 m, int x) { return (uint32_t) (m >> x) & 1u; }
    T1_HD static u64 neighbours(u64 up, u64 mid, u64 dn)
    {
        const u64 all = up | mid | dn;
        return up | dn | (all << 1) | (all >> 1);
    }
    T1_HD uint32_t decide(int ctx)
    {
        const uint32_t hdl = cx.handle((uint32_t) ctx);
        uint32_t code = cx.ld(hdl);
        const uint32_t d = mq.decode(code, tab);
        cx.st(hdl, code);
        return d;
    }
    // a lane may decode while the bytes its decoder could touch next are there (two per BYTEIN, up to four BYTEINs a refill)
    T1_HD bool can_decode() { return !env.starved(mq.pos + 6); }
    T1_HD void top_up() { mq.refill(env.bytes()); }

    // ---- one stripe of a pass; `on`: this lane's code-block has the segment
    T1_HD void sigprop(bool on, int plane, int y0)
    {
        u64 s[6] = {0, 0, 0, 0, 0, 0}, neg[6] = {0, 0, 0, 0, 0, 0}, vis[4] = {0, 0, 0, 0}, cand[4] = {0, 0, 0, 0};
        const u64 wmask = w >= 64 ? ~0ull : ((1ull << w) - 1);
        const int nrows = h - y0 < 4 ? h - y0 : 4;
        if (on) {
#pragma unroll
            for (int r = 0; r < 6; r++) { const int y = y0 - 1 + r; s[r] = st.S(y); neg[r] = (y >= 0 && y < 64) ? st.NEG(y) : 0ull; }
#pragma unroll
            for (int r = 0; r < 4; r++) cand[r] = r < nrows ? ~s[r + 1] & neighbours(s[r], s[r + 1], s[r + 2]) & wmask : 0ull;
        }
        int x = 0, r = 0;                                                 // position of the pending sign decision
        bool sign_due = false;
        for (;;) {
            const u64 pend = cand[0] | cand[1] | cand[2] | cand[3];
            const bool more = on && (sign_due || pend != 0);
            if (!env.any(more)) break;
            const bool go = more && can_decode();
            if (go) {
                if (!sign_due) {                                          // next candidate in scan order: lowest column, then lowest row
                    x = ctz64(pend);
                    const uint32_t nib = bit(cand[0], x) | (bit(cand[1], x) << 1) | (bit(cand[2], x) << 2) | (bit(cand[3], x) << 3);
                    r = nib & 1u ? 0 : (nib & 2u ? 1 : (nib & 4u ? 2 : 3));
                }
                const u64 b = 1ull << x;
                const u64 su = r == 0 ? s[0] : (r == 1 ? s[1] : (r == 2 ? s[2] : s[3])), sm = r == 0 ? s[1] : (r == 1 ? s[2] : (r == 2 ? s[3] : s[4])),
                          sd = r == 0 ? s[2] : (r == 1 ? s[3] : (r == 2 ? s[4] : s[5]));
                int ctx;
                int xb = 0;
                if (!sign_due) ctx = ctx_zc(tri(su, x), tri(sm, x), tri(sd, x), orient);
                else {
                    const u64 nu = r == 0 ? neg[0] : (r == 1 ? neg[1] : (r == 2 ? neg[2] : neg[3])), nm = r == 0 ? neg[1] : (r == 1 ? neg[2] : (r == 2 ? neg[3] : neg[4])),
                              nd = r == 0 ? neg[2] : (r == 1 ? neg[3] : (r == 2 ? neg[4] : neg[5]));
                    ctx = ctx_sc(tri(su, x), tri(sm, x), tri(sd, x), tri(nu, x), tri(nm, x), tri(nd, x), xb);
                }
                const uint32_t d = decide(ctx);
                if (!sign_due) {
                    // visited; a 1 makes the sample significant: its sign follows
#pragma unroll
                    for (int q = 0; q < 4; q++) if (q == r) { cand[q] &= ~b; vis[q] |= b; }
                    sign_due = d != 0;
                } else {
                    const int ng = (int) d ^ xb;
                    st.set_sig(x, y0 + r, ng, plane);
#pragma unroll
                    for (int q = 0; q < 4; q++) if (q == r) { s[q + 1] |= b; if (ng) neg[q + 1] |= b; }
                    // samples later in the scan that now have a significant neighbour: the row below in this column, the next column
                    const u64 bn = x + 1 < 64 ? b << 1 : 0ull;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        u64 add = 0;
                        if (q == r + 1) add = b | bn;
                        else if (q == r || q == r - 1) add = bn;
                        if (q < nrows) cand[q] |= add & ~s[q + 1] & ~vis[q] & wmask;
                    }
                    sign_due = false;
                }
                top_up();
            }
            if (env.any(more && !go)) env.refill_point(mq.pos);
        }
        if (on) {
#pragma unroll
            for (int q = 0; q < 4; q++) { st.S(y0 + q) = s[q + 1]; st.NEG(y0 + q) = neg[q + 1]; st.VIS(y0 + q) = vis[q]; }
        }
    }

    T1_HD void refine(bool on, int plane, int y0)
    {
        u64 m[4] = {0, 0, 0, 0}, ref[4] = {0, 0, 0, 0}, nb[4] = {0, 0, 0, 0}, ng[4] = {0, 0, 0, 0}, done[4] = {0, 0, 0, 0};
        if (on) {
            u64 s[6];
#pragma unroll
            for (int r = 0; r < 6; r++) s[r] = st.S(y0 - 1 + r);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                m[r] = s[r + 1] & ~st.VIS(y0 + r);                        // significant before this plane (significant implies inside the block)
                ref[r] = st.REF(y0 + r);
                nb[r] = neighbours(s[r], s[r + 1], s[r + 2]);
                ng[r] = st.NEG(y0 + r);
                done[r] = m[r];
            }
        }
        for (;;) {
            const u64 pend = m[0] | m[1] | m[2] | m[3];
            const bool more = on && pend != 0;
            if (!env.any(more)) break;
            const bool go = more && can_decode();
            if (go) {
                const int x = ctz64(pend);
                const uint32_t nib = bit(m[0], x) | (bit(m[1], x) << 1) | (bit(m[2], x) << 2) | (bit(m[3], x) << 3);
                const int r = nib & 1u ? 0 : (nib & 2u ? 1 : (nib & 4u ? 2 : 3));
                const u64 b = 1ull << x;
                const u64 rr = r == 0 ? ref[0] : (r == 1 ? ref[1] : (r == 2 ? ref[2] : ref[3])), nn = r == 0 ? nb[0] : (r == 1 ? nb[1] : (r == 2 ? nb[2] : nb[3])),
                          gg = r == 0 ? ng[0] : (r == 1 ? ng[1] : (r == 2 ? ng[2] : ng[3]));
                const int ctx = (rr & b) ? CTX_MAG0 + 2 : CTX_MAG0 + (int) bit(nn, x);
                const uint32_t d = decide(ctx);
                st.refine(x, y0 + r, (int) d, plane, (int) bit(gg, x));
#pragma unroll
                for (int q = 0; q < 4; q++) if (q == r) m[q] &= ~b;
                top_up();
            }
            if (env.any(more && !go)) env.refill_point(mq.pos);
        }
        if (on) {
#pragma unroll
            for (int q = 0; q < 4; q++) st.REF(y0 + q) = ref[q] | done[q];
        }
    }

    T1_HD void cleanup(bool on, int plane, int y0)
    {
        u64 s[6] = {0, 0, 0, 0, 0, 0}, neg[6] = {0, 0, 0, 0, 0, 0}, m[4] = {0, 0, 0, 0};
        const u64 wmask = w >= 64 ? ~0ull : ((1ull << w) - 1);
        const int nrows = h - y0 < 4 ? h - y0 : 4;
        const bool full = y0 + 3 < h;
        u64 visany = 0;
        if (on) {
#pragma unroll
            for (int r = 0; r < 6; r++) { const int y = y0 - 1 + r; s[r] = st.S(y); neg[r] = (y >= 0 && y < 64) ? st.NEG(y) : 0ull; }
#pragma unroll
            for (int r = 0; r < 4; r++) { const u64 v = st.VIS(y0 + r); visany |= v; m[r] = r < nrows ? ~(s[r + 1] | v) & wmask : 0ull; }
        }
        // per-lane state machine: 0 column entry (run-length test), 1 / 2 the two UNIFORM bits of a run, 3 zero coding, 4 sign
        int state = 0, x = 0, r = 0, run = 0;
        for (;;) {
            const u64 pend = m[0] | m[1] | m[2] | m[3];
            const bool more = on && (state != 0 || pend != 0);
            if (!env.any(more)) break;
            const bool go = more && can_decode();
            if (go) {
                int ctx = CTX_UNI, xb = 0;
                bool agg = false;
                if (state == 0) {
                    x = ctz64(pend);
                    const u64 b0 = 1ull << x;
                    // run-length mode: the whole column is insignificant, unvisited and has an all-zero neighbourhood
                    if (full && !(visany & b0) && (m[0] & m[1] & m[2] & m[3] & b0)) {
                        uint32_t any = 0;
#pragma unroll
                        for (int q = 0; q < 6; q++) any |= tri(s[q], x);
                        agg = any == 0;
                    }
                    if (agg) ctx = CTX_AGG;
                    else {
                        const uint32_t nib = bit(m[0], x) | (bit(m[1], x) << 1) | (bit(m[2], x) << 2) | (bit(m[3], x) << 3);
                        r = nib & 1u ? 0 : (nib & 2u ? 1 : (nib & 4u ? 2 : 3));
                        state = 3;
                    }
                }
                const u64 b = 1ull << x;
                const u64 su = r == 0 ? s[0] : (r == 1 ? s[1] : (r == 2 ? s[2] : s[3])), sm = r == 0 ? s[1] : (r == 1 ? s[2] : (r == 2 ? s[3] : s[4])),
                          sd = r == 0 ? s[2] : (r == 1 ? s[3] : (r == 2 ? s[4] : s[5]));
                if (state == 3) ctx = ctx_zc(tri(su, x), tri(sm, x), tri(sd, x), orient);
                else if (state == 4) {
                    const u64 nu = r == 0 ? neg[0] : (r == 1 ? neg[1] : (r == 2 ? neg[2] : neg[3])), nm = r == 0 ? neg[1] : (r == 1 ? neg[2] : (r == 2 ? neg[3] : neg[4])),
                              nd = r == 0 ? neg[2] : (r == 1 ? neg[3] : (r == 2 ? neg[4] : neg[5]));
                    ctx = ctx_sc(tri(su, x), tri(sm, x), tri(sd, x), tri(nu, x), tri(nm, x), tri(nd, x), xb);
                }
                const uint32_t d = decide(ctx);
                if (state == 0) {                                         // (agg)
                    if (d) state = 1;
                    else { m[0] &= ~b; m[1] &= ~b; m[2] &= ~b; m[3] &= ~b; }      // the four samples stay insignificant
                } else if (state == 1) { run = (int) d << 1; state = 2; }
                else if (state == 2) {
                    run |= (int) d;
                    // rows above the run's end are insignificant and done; its end is significant without a zero-coding decision
                    r = run;
#pragma unroll
                    for (int q = 0; q < 4; q++) if (q <= run) m[q] &= ~b;
                    state = 4;
                } else if (state == 3) {
#pragma unroll
                    for (int q = 0; q < 4; q++) if (q == r) m[q] &= ~b;
                    state = d ? 4 : 0;
                } else {
                    const int ng = (int) d ^ xb;
                    st.set_sig(x, y0 + r, ng, plane);
#pragma unroll
                    for (int q = 0; q < 4; q++) if (q == r) { s[q + 1] |= b; if (ng) neg[q + 1] |= b; }
                    state = 0;
                }
                top_up();
            }
            if (env.any(more && !go)) env.refill_point(mq.pos);
        }
        if (on) {
#pragma unroll
            for (int q = 0; q < 4; q++) { st.S(y0 + q) = s[q + 1]; st.NEG(y0 + q) = neg[q + 1]; st.VIS(y0 + q) = 0; }   // the visited flags die with the plane
        }
    }
};

// Decodes `npasses` coding passes of a code-block with `numbps` bit-planes; `pmax` (uniform) = bit-planes of the wave's
// deepest code-block: every lane walks the segments of planes pmax - 1 .. 0 and takes part in those its code-block has.
template <class DStore, class Ctx, class Table, class Env>
T1_HD void decode_block_segments(DStore &st, Ctx &cx, const Table &tab, Env &env, int w, int h, int orient, int numbps, int npasses, int pmax)
{
    SegmentDecoder<DStore, Ctx, Table, Env> sd{st, cx, tab, env, MqLookahead{0, 0, 0, 0}, w, h, orient};
    const bool live = numbps > 0 && npasses > 0;
    for (uint32_t i = 0; i < 32; i++)
        cx.st(cx.handle(i), i >= (uint32_t) NCTX ? kNullCode : mq_code(i == CTX_UNI ? 46u : (i == CTX_AGG ? 3u : (i == CTX_ZC0 ? 4u : 0u))));
    if (live) sd.mq.init(env.bytes());
    const int nstr = (h + 3) >> 2;
    for (int p = pmax - 1; p >= 0; p--)
        for (int t = 0; t < 3; t++) {
            // pass index of (plane p, type t) in this lane's code-block; its top plane has a cleanup pass only
            const int pass = p == numbps - 1 ? 0 : 3 * (numbps - 1 - p) - 2 + t;
            const bool has = live && p < numbps && (p < numbps - 1 || t == 2) && pass < npasses;
            if (!env.any(has)) continue;
            env.pass_point(sd.mq.pos);
            for (int s = 0; s < 16; s++) {
                const bool on = has && s < nstr;
                if (!env.any(on)) break;
                if (t == 0) sd.sigprop(on, p, 4 * s);
                else if (t == 1) sd.refine(on, p, 4 * s);
                else sd.cleanup(on, p, 4 * s);
            }
        }
}

}  // namespace t1
}  // namespace ebcc
