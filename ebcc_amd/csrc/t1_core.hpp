// t1_core.hpp - JPEG 2000 tier-1 (T.800 Annex C MQ coder + Annex D bit-plane coding passes) as scalar code
// on 64-bit ROW MASKS, one code-block per lane.
//
// A code-block is at most 64x64, so one u64 holds one property of one row (bit x = column x):
// significance S, sign NEG, "visited in this plane's significance-propagation pass" VIS, "already
// refined once" REF and - encoder side - the bit-plane masks BP[plane][row] that a data-parallel
// kernel extracts with wave ballots.  Neighbourhood tests become shifts of three row masks, columns
// without candidates are skipped with ctz, and the only per-symbol work left is the context lookup and
// the MQ coder itself.  Every lane of a wavefront runs this code on its own code-block; the state
// arrays are interleaved across lanes (element i of lane l at [i * 64 + l]) so that lanes in lock-step
// touch one contiguous 512-byte line.
//
// The same source compiles for the host (tests/t1_host_check.cpp checks it against the oracle) and for
// gfx950.  Behaviour follows OpenJPEG 2.4.0's t1.c/mqc.c (the reference's dependency, called at
// /root/reference/src/ebcc_codec.c:173,1116): pass order, run-length mode, contexts, pass-rate
// conventions (+3 for unterminated passes, no 0xFF as last byte of a pass, final FLUSH).
#pragma once

#include <cstdint>
#include <type_traits>

#if defined(__HIPCC__)
#define T1_HD __host__ __device__ inline
#else
#define T1_HD inline
#endif
#ifndef T1_STAT
#define T1_STAT(i)            // work counters of the host statistics tool (tools/t1_stats.cpp)
#endif

namespace ebcc {
namespace t1 {

typedef unsigned long long u64;

enum { CTX_ZC0 = 0, CTX_SC0 = 9, CTX_MAG0 = 14, CTX_AGG = 17, CTX_UNI = 18, NCTX = 19 };
constexpr int kMaxPasses = 3 * 30;

// T.800 table C-2 packed: qe | nmps << 16 | nlps << 22 | switch << 28
T1_HD uint32_t mq_entry(int i)
{
    constexpr uint32_t T[47] = {
#define E(q, m, l, s) ((uint32_t) (q) | ((uint32_t) (m) << 16) | ((uint32_t) (l) << 22) | ((uint32_t) (s) << 28))
        E(0x5601, 1, 1, 1),   E(0x3401, 2, 6, 0),   E(0x1801, 3, 9, 0),   E(0x0AC1, 4, 12, 0),  E(0x0521, 5, 29, 0),
        E(0x0221, 38, 33, 0), E(0x5601, 7, 6, 1),   E(0x5401, 8, 14, 0),  E(0x4801, 9, 14, 0),  E(0x3801, 10, 14, 0),
        E(0x3001, 11, 17, 0), E(0x2401, 12, 18, 0), E(0x1C01, 13, 20, 0), E(0x1601, 29, 21, 0), E(0x5601, 15, 14, 1),
        E(0x5401, 16, 14, 0), E(0x5101, 17, 15, 0), E(0x4801, 18, 16, 0), E(0x3801, 19, 17, 0), E(0x3401, 20, 18, 0),
        E(0x3001, 21, 19, 0), E(0x2801, 22, 19, 0), E(0x2401, 23, 20, 0), E(0x2201, 24, 21, 0), E(0x1C01, 25, 22, 0),
        E(0x1801, 26, 23, 0), E(0x1601, 27, 24, 0), E(0x1401, 28, 25, 0), E(0x1201, 29, 26, 0), E(0x1101, 30, 27, 0),
        E(0x0AC1, 31, 28, 0), E(0x09C1, 32, 29, 0), E(0x08A1, 33, 30, 0), E(0x0521, 34, 31, 0), E(0x0441, 35, 32, 0),
        E(0x02A1, 36, 33, 0), E(0x0221, 37, 34, 0), E(0x0141, 38, 35, 0), E(0x0111, 39, 36, 0), E(0x0085, 40, 37, 0),
        E(0x0049, 41, 38, 0), E(0x0025, 42, 39, 0), E(0x0015, 43, 40, 0), E(0x0009, 44, 41, 0), E(0x0005, 45, 42, 0),
        E(0x0001, 45, 43, 0), E(0x5601, 46, 46, 0)};
#undef E
    return T[i];
}

// Where the state table lives: the constexpr array above ends up in constant memory, and a per-lane dynamic
// index into it is a global load on the critical path of EVERY symbol.  Device kernels copy the 47 words to
// LDS once and pass an LdsTable; the host build uses ConstTable.
struct ConstTable {
    T1_HD uint32_t operator()(int i) const { return mq_entry(i); }
};

struct Contexts {
    // 19 contexts x 7 bits (6-bit state index | mps << 6) packed 9 per word: no dynamically indexed
    // array, so the state stays in registers on the GPU.
    u64 w0, w1, w2;
    T1_HD void reset()
    {
        w0 = w1 = w2 = 0;
        set(CTX_UNI, 46); set(CTX_AGG, 3); set(CTX_ZC0, 4);              // T.800 table D-7
    }
    // Run-time context index: pure value arithmetic on the three words (no address-of-member selects, which
    // would force the whole state into scratch memory on the GPU).
    T1_HD uint32_t get(int i) const
    {
        const int hi1 = i >= 9, hi2 = i >= 18;
        const int k = i - 9 * (hi1 + hi2);
        const u64 m2 = 0ull - (u64) hi2, m1 = (0ull - (u64) hi1) & ~m2, m0 = ~(m1 | m2);
        const u64 w = (w0 & m0) | (w1 & m1) | (w2 & m2);
        return (uint32_t) (w >> (7 * k)) & 0x7Fu;
    }
    T1_HD void set(int i, uint32_t v)
    {
        const int hi1 = i >= 9, hi2 = i >= 18;
        const int k = i - 9 * (hi1 + hi2);
        const u64 m2 = 0ull - (u64) hi2, m1 = (0ull - (u64) hi1) & ~m2, m0 = ~(m1 | m2);
        const u64 clr = 0x7Full << (7 * k), nv = (u64) (v & 0x7Fu) << (7 * k);
        w0 = (w0 & ~(clr & m0)) | (nv & m0);
        w1 = (w1 & ~(clr & m1)) | (nv & m1);
        w2 = (w2 & ~(clr & m2)) | (nv & m2);
    }
    // byte-per-context form (context i in byte i & 3 of word i >> 2): what the checkpoints store and what the
    // stream coder keeps in LDS
    T1_HD void to_bytes(uint32_t x[5]) const
    {
        for (int j = 0; j < 5; j++) x[j] = 0;
        for (int i = 0; i < NCTX; i++) x[i >> 2] |= get(i) << (8 * (i & 3));
    }
    T1_HD static Contexts from_bytes(const uint32_t x[5])
    {
        Contexts c{0, 0, 0};
        for (int i = 0; i < NCTX; i++) c.set(i, (x[i >> 2] >> (8 * (i & 3))) & 0x7Fu);
        return c;
    }
    // The caller usually knows the word at compile time (zero-coding contexts live in w0, sign / magnitude /
    // run-length in w1, UNIFORM in w2): no selects between the three words then.
    template <int W>
    T1_HD u64 &word() { if constexpr (W == 0) return w0; else if constexpr (W == 1) return w1; else return w2; }
    template <int W>
    T1_HD uint32_t get_in(int k) { return (uint32_t) (word<W>() >> (7 * k)) & 0x7Fu; }
    // the same with the field's bit position (7 k) given: a context table can hold the position itself
    template <int W>
    T1_HD uint32_t get_at(int sh) { return (uint32_t) (word<W>() >> sh) & 0x7Fu; }
    template <int W>
    T1_HD void set_at(int sh, uint32_t v)
    {
        u64 &w = word<W>();
        w = (w & ~(0x7Full << sh)) | ((u64) (v & 0x7Fu) << sh);
    }
    template <int W>
    T1_HD void set_in(int k, uint32_t v)
    {
        u64 &w = word<W>();
        w = (w & ~(0x7Full << (7 * k))) | ((u64) (v & 0x7Fu) << (7 * k));
    }
};

T1_HD int renorm_shifts(uint32_t a);

// ------------------------------------------------------------------------------------------------
// MQ encoder (C.2).  Sink: void put(int index, uint8_t byte)
// ------------------------------------------------------------------------------------------------
template <class Sink, class Table = ConstTable>
struct MqEncoder {
    uint32_t a, c;
    int ct;
    int n;            // index of the byte being formed (== opj_mqc_numbytes)
    uint32_t cur;     // its value; index -1 is the non-FF byte that precedes the segment
    Contexts cx;
    Sink sink;
    Table tab;
    uint32_t shifts;  // renormalisation shifts so far (the decoder performs exactly the same number)

    T1_HD void init()
    {
        cx.reset();
        a = 0x8000; c = 0; ct = 12; n = -1; cur = 0; shifts = 0;
    }
    T1_HD void emit() { if (n >= 0) sink.put(n, (uint8_t) cur); }
    T1_HD void byteout()
    {
        if (cur == 0xFF) {
            emit(); n++; cur = c >> 20; c &= 0xFFFFF; ct = 7;
        } else if ((c & 0x8000000) == 0) {
            emit(); n++; cur = c >> 19; c &= 0x7FFFF; ct = 8;
        } else {
            cur++;
            if (cur == 0xFF) {
                c &= 0x7FFFFFF;
                emit(); n++; cur = c >> 20; c &= 0xFFFFF; ct = 7;
            } else {
                emit(); n++; cur = c >> 19; c &= 0x7FFFF; ct = 8;
            }
        }
    }
    T1_HD void renorm()
    {
        // all shifts at once (C.2.6 does them bit by bit): BYTEOUT whenever the down-counter runs out
        int n = renorm_shifts(a);
        shifts += (uint32_t) n;
        while (n >= ct) { a <<= ct; c <<= ct; n -= ct; byteout(); }
        a <<= n; c <<= n; ct -= n;
    }
    // W: word of the context set that holds the context, k: its index inside that word
    template <int W>
    T1_HD void encode_in(int k, int d)
    {
        uint32_t st = cx.template get_in<W>(k);
        uint32_t e = tab((int) (st & 0x3F));
        uint32_t qe = e & 0xFFFF;
        int mps = st >> 6;
        a -= qe;
        if (d == mps) {
            if ((a & 0x8000) == 0) {
                if (a < qe) a = qe; else c += qe;
                cx.template set_in<W>(k, ((e >> 16) & 0x3F) | (mps << 6));
                renorm();
            } else {
                c += qe;
            }
        } else {
            if (a < qe) c += qe; else a = qe;
            if (e >> 28) mps ^= 1;
            cx.template set_in<W>(k, ((e >> 22) & 0x3F) | (mps << 6));
            renorm();
        }
    }
    // any context, chosen at run time (selects between the three words): the stream coder's form
    T1_HD void encode(int ctx, int d)
    {
        uint32_t st = cx.get(ctx);
        uint32_t e = tab((int) (st & 0x3F));
        uint32_t qe = e & 0xFFFF;
        int mps = st >> 6;
        a -= qe;
        if (d == mps) {
            if ((a & 0x8000) == 0) {
                if (a < qe) a = qe; else c += qe;
                cx.set(ctx, ((e >> 16) & 0x3F) | (mps << 6));
                renorm();
            } else {
                c += qe;
            }
        } else {
            if (a < qe) c += qe; else a = qe;
            if (e >> 28) mps ^= 1;
            cx.set(ctx, ((e >> 22) & 0x3F) | (mps << 6));
            renorm();
        }
    }
    T1_HD void encode_zc(int ctx, int d) { encode_in<0>(ctx - CTX_ZC0, d); }
    T1_HD void encode_sc(int ctx, int d) { encode_in<1>(ctx - 9, d); }
    T1_HD void encode_mag(int ctx, int d) { encode_in<1>(ctx - 9, d); }
    T1_HD void encode_agg(int d) { encode_in<1>(CTX_AGG - 9, d); }
    T1_HD void encode_uni(int d) { encode_in<2>(0, d); }
    T1_HD int numbytes() const { return n; }
    T1_HD void flush()
    {
        uint32_t tempc = c + a;                                           // SETBITS
        c |= 0xFFFF;
        if (c >= tempc) c -= 0x8000;
        c <<= ct; byteout();
        c <<= ct; byteout();
        emit();
        if (cur != 0xFF) n++;                                             // a trailing FF is not part of the segment
    }
};

// ------------------------------------------------------------------------------------------------
// MQ decoder (C.3).  Source: uint32_t get(int index) returning 0xFF past the end
// ------------------------------------------------------------------------------------------------
#ifdef T1_RUN_STATS
static unsigned long t1_run_stats[64];
#endif
// x / q for x < 2^16, 0 < q < 2^16 (the GPU has no integer divider: a reciprocal, then the two possible corrections)
T1_HD uint32_t div_u16(uint32_t x, uint32_t q)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t k = (uint32_t) ((float) x * __frcp_rn((float) q));
    if (k * q > x) k--;
    if ((k + 1) * q <= x) k++;
    return k;
#else
    return x / q;
#endif
}

template <class Source, class = void>
struct source_is_sequential : std::false_type {};
template <class Source>
struct source_is_sequential<Source, std::enable_if_t<Source::kSequential>> : std::true_type {};

template <class Source, class Table = ConstTable>
struct MqDecoder {
    uint32_t a, c;
    int ct, pos;
    Contexts cx;
    Source src;
    Table tab;

    // BYTEIN looks at the byte the decoder stands on and the one after it, and steps forward by at most one: a source that
    // keeps those two at hand (Source::kSequential: at(0), at(1), step()) saves the two random accesses of get()
    T1_HD void bytein()
    {
        uint32_t cur, nxt;
        if constexpr (source_is_sequential<Source>::value) { cur = src.at0(); nxt = src.at1(); }
        else { cur = src.get(pos); nxt = src.get(pos + 1); }
        if (cur == 0xFF) {
            if (nxt > 0x8F) { c += 0xFF00; ct = 8; }
            else { step(); c += nxt << 9; ct = 7; }
        } else {
            step(); c += nxt << 8; ct = 8;
        }
    }
    T1_HD void step()
    {
        pos++;
        if constexpr (source_is_sequential<Source>::value) src.step();
    }
    T1_HD void init()
    {
        cx.reset();
        pos = 0;
        if constexpr (source_is_sequential<Source>::value) c = src.at0() << 16;
        else c = src.get(0) << 16;
        bytein();
        c <<= 7; ct -= 7; a = 0x8000;
    }
    T1_HD void renorm()
    {
        // all shifts at once (C.3.3 does them bit by bit): BYTEIN only when another shift needs a fresh bit
        int n = renorm_shifts(a);
        do {
            if (ct == 0) bytein();
            const int k = n < ct ? n : ct;
            a <<= k; c <<= k; ct -= k; n -= k;
        } while (n);
    }
    template <int W>
    T1_HD int decode_in(int k) { return decode_at<W>(7 * k); }
    // sh: bit position of the context's field in its word (7 x index inside the word)
    template <int W>
    T1_HD int decode_at(int sh)
    {
        uint32_t st = cx.template get_at<W>(sh);
        uint32_t e = tab((int) (st & 0x3F));
        uint32_t qe = e & 0xFFFF;
        int mps = st >> 6, d;
        a -= qe;
        if ((c >> 16) < qe) {
            if (a < qe) { d = mps; cx.template set_at<W>(sh, ((e >> 16) & 0x3F) | (mps << 6)); }
            else { d = 1 - mps; if (e >> 28) mps ^= 1; cx.template set_at<W>(sh, ((e >> 22) & 0x3F) | (mps << 6)); }
            a = qe;
            renorm();
        } else {
            c -= qe << 16;
            if ((a & 0x8000) == 0) {
                if (a < qe) { d = 1 - mps; if (e >> 28) mps ^= 1; cx.template set_at<W>(sh, ((e >> 22) & 0x3F) | (mps << 6)); }
                else { d = mps; cx.template set_at<W>(sh, ((e >> 16) & 0x3F) | (mps << 6)); }
                renorm();
            } else {
                d = mps;
            }
        }
        return d;
    }
    // Up to n decisions in a row on the run-length context that all come out 0, in one step.  While the context's MPS is 0
    // a decision that takes neither the exchange nor the renormalisation branch of decode_in() is `a -= qe, c -= qe << 16`
    // and leaves the context alone, so the next k of them are one update of the two registers: decision j (from 0)
    // stays on that branch while (c >> 16) - j * qe >= qe and a - (j + 1) * qe >= 0x8000.  Returns k (0: the next
    // decision needs decode_agg()).  The cleanup pass of a sparse code-block is almost only such decisions - one per
    // all-zero column of four - which made it the bulk of the decoder's instructions.
    T1_HD int agg_zero_run(int n)
    {
        const uint32_t st = cx.template get_in<1>(CTX_AGG - 9);
        if (st >> 6) return 0;
        const uint32_t qe = tab((int) (st & 0x3F)) & 0xFFFF;
        const uint32_t ka = div_u16(a - 0x8000u, qe), kc = div_u16(c >> 16, qe);
        uint32_t k = ka < kc ? ka : kc;
        if (k > (uint32_t) n) k = (uint32_t) n;
#ifdef T1_RUN_STATS
        t1_run_stats[k > 63 ? 63 : k]++;
#endif
        a -= k * qe;
        c -= (k * qe) << 16;
        return (int) k;
    }
    T1_HD int decode_zc(int ctx) { return decode_in<0>(ctx - CTX_ZC0); }
    T1_HD int decode_sc(int ctx) { return decode_in<1>(ctx - 9); }
    T1_HD int decode_zc_at(int sh) { return decode_at<0>(sh); }           // (a context policy with kFieldPositions hands these out)
    T1_HD int decode_sc_at(int sh) { return decode_at<1>(sh); }
    T1_HD int decode_mag(int ctx) { return decode_in<1>(ctx - 9); }
    T1_HD int decode_agg() { return decode_in<1>(CTX_AGG - 9); }
    T1_HD int decode_uni() { return decode_in<2>(0); }
};

// ------------------------------------------------------------------------------------------------
// contexts from row masks
// ------------------------------------------------------------------------------------------------
// three bits {x-1, x, x+1} of a row mask, at bit positions 0..2
T1_HD uint32_t tri(u64 m, int x) { return (uint32_t) ((x ? (m >> (x - 1)) : (m << 1)) & 7u); }

// zero-coding context, T.800 table D-1.  up/mid/dn: tri() of significance of rows y-1, y, y+1.
T1_HD int ctx_zc(uint32_t up, uint32_t mid, uint32_t dn, int orient)
{
    int h = (int) (mid & 1) + (int) ((mid >> 2) & 1);
    int v = (int) ((up >> 1) & 1) + (int) ((dn >> 1) & 1);
    int d = (int) (up & 1) + (int) ((up >> 2) & 1) + (int) (dn & 1) + (int) ((dn >> 2) & 1);
    int n;
    if (orient == 1) { int t = h; h = v; v = t; }
    if (orient == 3) {
        int hv = h + v;
        if (d == 0) n = hv == 0 ? 0 : (hv == 1 ? 1 : 2);
        else if (d == 1) n = hv == 0 ? 3 : (hv == 1 ? 4 : 5);
        else if (d == 2) n = hv == 0 ? 6 : 7;
        else n = 8;
    } else {
        if (h == 0) {
            if (v == 0) n = d == 0 ? 0 : (d == 1 ? 1 : 2);
            else if (v == 1) n = 3;
            else n = 4;
        } else if (h == 1) {
            if (v == 0) n = d == 0 ? 5 : 6;
            else n = 7;
        } else n = 8;
    }
    return CTX_ZC0 + n;
}

// sign-coding context and XOR bit, tables D-2/D-3.  s*: tri() of significance, n*: tri() of sign masks
T1_HD int ctx_sc(uint32_t sup, uint32_t smid, uint32_t sdn, uint32_t nup, uint32_t nmid, uint32_t ndn, int &xorbit)
{
    int hc = 0, vc = 0;
    if (smid & 1) hc += (nmid & 1) ? -1 : 1;
    if (smid & 4) hc += (nmid & 4) ? -1 : 1;
    if (sup & 2) vc += (nup & 2) ? -1 : 1;
    if (sdn & 2) vc += (ndn & 2) ? -1 : 1;
    hc = hc > 1 ? 1 : (hc < -1 ? -1 : hc);
    vc = vc > 1 ? 1 : (vc < -1 ? -1 : vc);
    int n, xb = 0;
    if (hc == 1) n = vc == 1 ? 4 : (vc == 0 ? 3 : 2);
    else if (hc == 0) { if (vc == 1) n = 1; else if (vc == 0) n = 0; else { n = 1; xb = 1; } }
    else { xb = 1; n = vc == 1 ? 2 : (vc == 0 ? 3 : 4); }
    xorbit = xb;
    return CTX_SC0 + n;
}

// The serial passes form a column's contexts from WINDOWS: the 3-bit windows {x-1, x, x+1} of the six significance rows
// around the stripe in one word, row r in bits 3r .. 3r+2 (Passes::windows).  The zero-coding context of stripe row R is a
// function of the nine bits from 3R on - rows R (above), R+1 (own), R+2 (below) - and the sign context of eight bits
// picked from the significance and the sign windows:
//   bit 0 above significant, 1 above negative, 2 left significant, 3 left negative, 4 right significant, 5 right negative,
//   6 below significant, 7 below negative.
// A context policy turns the index into the context: DirectCtx evaluates tables D-1 .. D-3 (host builds), the device
// kernels read a 512-byte table per orientation and a 256-byte sign table from LDS (t1_device.hpp: LdsCtx) - what was
// ~25 + ~30 vector instructions per decision.
T1_HD uint32_t sc_index(uint32_t si, uint32_t ni)                        // si / ni: nine window bits of significance / sign
{
    return (((si >> 1) & 0x41u) | (((ni >> 1) & 0x41u) << 1)) | ((((si >> 3) & 5u) | (((ni >> 3) & 5u) << 1)) << 2);
}
struct DirectCtx {
    static constexpr bool kFieldPositions = false;   // zc() / sc() return context numbers (a table policy may return 7 x (index in the word) instead)
    int orient = 0;
    T1_HD void bind(int o) { orient = o; }
    T1_HD int zc(uint32_t idx) const { return ctx_zc(idx & 7u, (idx >> 3) & 7u, (idx >> 6) & 7u, orient); }
    T1_HD int sc(uint32_t idx, int &xb) const
    {
        const uint32_t sup = (idx & 1u) << 1, nup = ((idx >> 1) & 1u) << 1, sdn = ((idx >> 6) & 1u) << 1, ndn = ((idx >> 7) & 1u) << 1;
        const uint32_t smid = ((idx >> 2) & 1u) | (((idx >> 4) & 1u) << 2), nmid = ((idx >> 3) & 1u) | (((idx >> 5) & 1u) << 2);
        return ctx_sc(sup, smid, sdn, nup, nmid, ndn, xb);
    }
};

// renormalisation shift count of the interval register: a in [1, 0x7FFF] -> shifts until bit 15 is set
T1_HD int renorm_shifts(uint32_t a)
{
    return __builtin_clz(a) - 16;                                        // (a is never 0)
}

T1_HD int ctz64(u64 v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((long long) v) - 1;
#else
    return __builtin_ctzll(v);
#endif
}

// ------------------------------------------------------------------------------------------------
// Code-block state accessors (a "Store" provides these; indices are logical)
//   u64 &S(int y)    y in [-1, 64]          significance
//   u64 &NEG(int y), &VIS(int y), &REF(int y)   y in [0, 64)
// Encoder store additionally:  u64 BP(int plane, int y) (bit-plane masks), u64 SGN(int y) (sign of every
//   coefficient), u64 &SPS(int y) (output: became significant in a significance-propagation pass)
// Decoder store additionally:  void set_sig(int x, int y, int neg, int plane), void refine(int x, int y, int bit, int plane, int neg)
// ------------------------------------------------------------------------------------------------

// Observer of a coder run (checkpointing): every hook is a no-op here.
struct NoObserver {
    template <class Mq>
    T1_HD void pass_start(int, const Mq &) {}
    template <class Mq>
    T1_HD void stripe_start(int, const Mq &) {}
    template <class Store>
    T1_HD void sigprop_done(int, Store &) {}
};

// One stripe worth of state held in locals (fully unrolled accesses keep it in registers).
struct Stripe {
    u64 s[6];        // significance rows y0-1 .. y0+4
    u64 neg[6];      // signs of those rows
    u64 vis[4];
    u64 wmask;       // columns inside the code-block
    int nrows;       // rows of this stripe inside the code-block (1..4)
    u64 sgn[4];      // encoder: sign of every coefficient of the stripe's rows
    u64 sps[4];      // encoder: became significant in a propagation pass (accumulated, stored with the stripe)
};

template <bool ENC, class Store, class Coder, class Obs = NoObserver, class CtxP = DirectCtx>
struct Passes {
    Store &st;
    Coder &mq;
    int w, h, orient;
    Obs *obs;
    CtxP cp;

    T1_HD Passes(Store &s, Coder &c, int w_, int h_, int o, Obs *ob = nullptr, CtxP cp_ = CtxP()) : st(s), mq(c), w(w_), h(h_), orient(o), obs(ob), cp(cp_) { cp.bind(o); }
    // the windows of six row masks at column x (see DirectCtx)
    T1_HD static uint32_t windows(const u64 *m, int x)
    {
        // (one 64-bit shift per row by x - 1; column 0 has no left neighbour: every field moves up by one bit instead)
        const int sh = x > 0 ? x - 1 : 0;
        uint32_t v = 0;
#pragma unroll
        for (int r = 0; r < 6; r++) v |= ((uint32_t) (m[r] >> sh) & 7u) << (3 * r);
        return x > 0 ? v : (v << 1) & 0x36DB6u;
    }
    T1_HD void stripe_hook(int y0)
    {
        if constexpr (!std::is_same<Obs, NoObserver>::value) obs->stripe_start(y0, mq);   // (may write to the coder: stream markers)
    }

    T1_HD void load(Stripe &sp, int y0)
    {
        sp.wmask = w >= 64 ? ~0ull : ((1ull << w) - 1);
        sp.nrows = h - y0 < 4 ? h - y0 : 4;
#pragma unroll
        for (int r = 0; r < 6; r++) {
            int y = y0 - 1 + r;
            sp.s[r] = st.S(y);
            sp.neg[r] = (y >= 0 && y < 64) ? st.NEG(y) : 0ull;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            sp.vis[r] = st.VIS(y0 + r);
            if constexpr (ENC) { sp.sgn[r] = st.SGN(y0 + r); sp.sps[r] = 0; }
        }
    }
    T1_HD void store(const Stripe &sp, int y0)
    {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            st.S(y0 + r) = sp.s[r + 1];
            st.NEG(y0 + r) = sp.neg[r + 1];
            st.VIS(y0 + r) = sp.vis[r];
            if constexpr (ENC) { if (sp.sps[r]) st.SPS(y0 + r) |= sp.sps[r]; }
        }
    }

    // sign coding of (x, row r) and state update; returns nothing.  R is a compile-time row.
    // sw / nw: the column's significance / sign windows (nw is taken from the masks when the first sign of the column is
    // coded: have_n); both are kept up to date for the rows below
    template <int R>
    T1_HD void code_sign(Stripe &sp, uint32_t &sw, uint32_t &nw, bool &have_n, int x, int y0, int plane, bool from_sigprop)
    {
        if (!have_n) { nw = windows(sp.neg, x); have_n = true; }
        int xb;
        int cx = cp.sc(sc_index((sw >> (3 * R)) & 0x1FFu, (nw >> (3 * R)) & 0x1FFu), xb);
        int neg;
        if constexpr (ENC) {
            neg = (int) ((sp.sgn[R] >> x) & 1);
            mq.encode_sc(cx, neg ^ xb);
        } else if constexpr (CtxP::kFieldPositions) {
            neg = mq.decode_sc_at(cx) ^ xb;
        } else {
            neg = mq.decode_sc(cx) ^ xb;
        }
        sp.s[R + 1] |= 1ull << x;
        sw |= 0x10u << (3 * R);
        if (neg) { sp.neg[R + 1] |= 1ull << x; nw |= 0x10u << (3 * R); }
        if constexpr (ENC) { if (from_sigprop) sp.sps[R] |= 1ull << x; }
        else st.set_sig(x, y0 + R, neg, plane);
    }

    // ---------------- significance propagation pass over one stripe
    template <int R>
    T1_HD bool sigprop_cell(Stripe &sp, uint32_t &sw, uint32_t &nw, bool &have_n, int x, int y0, int plane, u64 bp)
    {
        const uint32_t idx = (sw >> (3 * R)) & 0x1FFu;                // (x comes from a width-masked column set)
        if (R >= sp.nrows || (idx & 0x10u) || !(idx & 0x1EFu)) return false;   // outside, significant, or no significant neighbour
        int cx = cp.zc(idx);
        int v;
        if constexpr (ENC) { v = (int) ((bp >> x) & 1); mq.encode_zc(cx, v); }
        else if constexpr (CtxP::kFieldPositions) v = mq.decode_zc_at(cx);
        else v = mq.decode_zc(cx);
        sp.vis[R] |= 1ull << x;
        if (v) { code_sign<R>(sp, sw, nw, have_n, x, y0, plane, true); return true; }
        return false;
    }

    T1_HD void sigprop(int plane, int ystart = 0)
    {
        for (int y0 = ystart; y0 < h; y0 += 4) {
            stripe_hook(y0);
            Stripe sp;
            load(sp, y0);
            u64 b0 = 0, b1 = 0, b2 = 0, b3 = 0;
            if constexpr (ENC) { b0 = st.BP(plane, y0); b1 = st.BP(plane, y0 + 1); b2 = st.BP(plane, y0 + 2); b3 = st.BP(plane, y0 + 3); }
            // columns that can hold a candidate given the significance known so far
            u64 nb = 0;
#pragma unroll
            for (int r = 0; r < 6; r++) nb |= sp.s[r];
            u64 pending = (nb | (nb << 1) | (nb >> 1)) & sp.wmask;
            if (!pending) continue;                                      // nothing significant in or around the stripe: no candidate, nothing to store
            while (pending) {
                T1_STAT(0);
                int x = ctz64(pending);
                pending &= pending - 1;
                uint32_t sw = windows(sp.s, x), nw = 0;
                bool have_n = false, grew = false;
                grew |= sigprop_cell<0>(sp, sw, nw, have_n, x, y0, plane, b0);
                grew |= sigprop_cell<1>(sp, sw, nw, have_n, x, y0, plane, b1);
                grew |= sigprop_cell<2>(sp, sw, nw, have_n, x, y0, plane, b2);
                grew |= sigprop_cell<3>(sp, sw, nw, have_n, x, y0, plane, b3);
                if (grew && x + 1 < w) pending |= 1ull << (x + 1);       // a new neighbour to the right
            }
            store(sp, y0);
        }
    }

    // ---------------- magnitude refinement pass
    // nb: "some neighbour is significant" for every column of row R (nothing becomes significant in this pass,
    // so it is computed once per stripe with a handful of mask operations)
    template <int R>
    T1_HD void refine_cell(Stripe &sp, u64 &ref, u64 nb, int x, int y0, int plane, u64 bp)
    {
        const u64 bit = 1ull << x;
        if (!(sp.s[R + 1] & bit) || (sp.vis[R] & bit)) return;           // significant implies inside the block
        const int cx = (ref & bit) ? CTX_MAG0 + 2 : CTX_MAG0 + (int) ((nb >> x) & 1);
        int v;
        if constexpr (ENC) { v = (int) ((bp >> x) & 1); mq.encode_mag(cx, v); }
        else { v = mq.decode_mag(cx); st.refine(x, y0 + R, v, plane, (int) ((sp.neg[R + 1] >> x) & 1)); }
        ref |= bit;
    }
    T1_HD static u64 neighbours(u64 up, u64 mid, u64 dn)
    {
        const u64 all = up | mid | dn;
        return up | dn | (all << 1) | (all >> 1);
    }

    T1_HD void refine(int plane, int ystart = 0)
    {
        for (int y0 = ystart; y0 < h; y0 += 4) {
            stripe_hook(y0);
            Stripe sp;
            load(sp, y0);
            u64 b0 = 0, b1 = 0, b2 = 0, b3 = 0;
            if constexpr (ENC) { b0 = st.BP(plane, y0); b1 = st.BP(plane, y0 + 1); b2 = st.BP(plane, y0 + 2); b3 = st.BP(plane, y0 + 3); }
            u64 pending = (sp.s[1] & ~sp.vis[0]) | (sp.s[2] & ~sp.vis[1]) | (sp.s[3] & ~sp.vis[2]) | (sp.s[4] & ~sp.vis[3]);
            if (!pending) continue;
            u64 r0 = st.REF(y0), r1 = st.REF(y0 + 1), r2 = st.REF(y0 + 2), r3 = st.REF(y0 + 3);
            const u64 n0 = neighbours(sp.s[0], sp.s[1], sp.s[2]), n1 = neighbours(sp.s[1], sp.s[2], sp.s[3]),
                      n2 = neighbours(sp.s[2], sp.s[3], sp.s[4]), n3 = neighbours(sp.s[3], sp.s[4], sp.s[5]);
            while (pending) {
                T1_STAT(1);
                int x = ctz64(pending);
                pending &= pending - 1;
                refine_cell<0>(sp, r0, n0, x, y0, plane, b0);
                refine_cell<1>(sp, r1, n1, x, y0, plane, b1);
                refine_cell<2>(sp, r2, n2, x, y0, plane, b2);
                refine_cell<3>(sp, r3, n3, x, y0, plane, b3);
            }
            st.REF(y0) = r0; st.REF(y0 + 1) = r1; st.REF(y0 + 2) = r2; st.REF(y0 + 3) = r3;
        }
    }

    // ---------------- cleanup pass
    template <int R>
    T1_HD void cleanup_cell(Stripe &sp, uint32_t &sw, uint32_t &nw, bool &have_n, int x, int y0, int plane, u64 bp, bool skip_zc)
    {
        const u64 bit = 1ull << x;
        const uint32_t idx = (sw >> (3 * R)) & 0x1FFu;
        if (!skip_zc) {
            if (R >= sp.nrows || (idx & 0x10u) || (sp.vis[R] & bit)) return;
        }
        int v = 1;
        if (!skip_zc) {
            int cx = cp.zc(idx);
            if constexpr (ENC) { v = (int) ((bp >> x) & 1); mq.encode_zc(cx, v); }
            else if constexpr (CtxP::kFieldPositions) v = mq.decode_zc_at(cx);
            else v = mq.decode_zc(cx);
        }
        if (v) code_sign<R>(sp, sw, nw, have_n, x, y0, plane, false);
    }

    T1_HD void cleanup(int plane, int ystart = 0)
    {
        for (int y0 = ystart; y0 < h; y0 += 4) {
            stripe_hook(y0);
            Stripe sp;
            load(sp, y0);
            u64 b0 = 0, b1 = 0, b2 = 0, b3 = 0;
            if constexpr (ENC) { b0 = st.BP(plane, y0); b1 = st.BP(plane, y0 + 1); b2 = st.BP(plane, y0 + 2); b3 = st.BP(plane, y0 + 3); }
            const bool full = y0 + 3 < h;
            u64 pending = ~(sp.s[1] | sp.vis[0]);
            if (sp.nrows > 1) pending |= ~(sp.s[2] | sp.vis[1]);
            if (sp.nrows > 2) pending |= ~(sp.s[3] | sp.vis[2]);
            if (sp.nrows > 3) pending |= ~(sp.s[4] | sp.vis[3]);
            pending &= sp.wmask;
            // decoder: the run-length columns as of the start of the stripe.  What is coded in column x changes the
            // neighbourhood of columns x and x + 1 only, so when the scan reaches column x the bits above x still hold.
            u64 aggmask = 0;
            if constexpr (!ENC) {
                if (full) {
                    u64 anys = sp.s[0] | sp.s[1] | sp.s[2] | sp.s[3] | sp.s[4] | sp.s[5];
                    anys |= (anys << 1) | (anys >> 1);
                    aggmask = ~(anys | sp.vis[0] | sp.vis[1] | sp.vis[2] | sp.vis[3]) & sp.wmask;
                }
            }
            while (pending) {
                T1_STAT(2);
                int x = ctz64(pending);
                pending &= pending - 1;
                const u64 bit = 1ull << x;
                uint32_t sw = windows(sp.s, x), nw = 0;
                bool have_n = false;
                // run-length mode: the whole column is insignificant, unvisited and has an all-zero neighbourhood
                const bool agg = full && !((sp.vis[0] | sp.vis[1] | sp.vis[2] | sp.vis[3]) & bit) && sw == 0;
                int start = 0;
                if (agg) {
                    if constexpr (ENC) {
                        int run = (b0 >> x) & 1 ? 0 : ((b1 >> x) & 1 ? 1 : ((b2 >> x) & 1 ? 2 : ((b3 >> x) & 1 ? 3 : 4)));
                        mq.encode_agg(run != 4);
                        if (run == 4) continue;
                        mq.encode_uni(run >> 1);
                        mq.encode_uni(run & 1);
                        start = run;
                    } else {
                        // this column and the run-length columns right after it, as many as come out 0 in one step
                        const u64 after = x < 63 ? aggmask >> (x + 1) : 0ull;
                        const int k = mq.agg_zero_run(1 + ctz64(~after));
                        if (k) {
                            if (k > 1) pending &= ~((k >= 64 ? ~0ull : (1ull << k) - 1) << x);
                            continue;
                        }
                        if (!mq.decode_agg()) continue;
                        start = mq.decode_uni();
                        start = (start << 1) | mq.decode_uni();
                    }
                }
                if (start <= 0) cleanup_cell<0>(sp, sw, nw, have_n, x, y0, plane, b0, agg && start == 0);
                if (start <= 1) cleanup_cell<1>(sp, sw, nw, have_n, x, y0, plane, b1, agg && start == 1);
                if (start <= 2) cleanup_cell<2>(sp, sw, nw, have_n, x, y0, plane, b2, agg && start == 2);
                if (start <= 3) cleanup_cell<3>(sp, sw, nw, have_n, x, y0, plane, b3, agg && start == 3);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) sp.vis[r] = 0;                   // the visited flags die with the plane
            store(sp, y0);
        }
    }

    // =============================================================================================
    // Encoder passes computed on the row masks (decision emitter only).  For the encoder every outcome is known
    // in advance, so which samples a pass codes, which of them become significant and every context follow
    // from mask arithmetic over the whole stripe; only writing the decisions out in scan order stays a loop
    // over columns - one code path for all lanes, no per-sample branching.
    //   * membership of the propagation pass: a sample is coded if it is insignificant and has a significant
    //     neighbour WHEN IT IS VISITED, i.e. counting samples that became significant earlier in this pass
    //     (same column above, previous column).  Along a row this is a carry chain - one addition
    //     (flood()) - and the coupling between rows converges in a few sweeps.
    //   * contexts: the eight neighbours' significance at visit time as eight masks, their sums bit-sliced
    //     (T.800 tables D-1..D-3 as boolean functions).
    // =============================================================================================
    T1_HD static u64 flood(u64 p, u64 g)                                 // n(x) = g(x) | (p(x) & n(x-1)),  g subset of p
    {
        const u64 sum = p + g;
        return g | (p & (sum ^ p ^ g));
    }
    struct RowCtx { u64 n0, n1, n2, n3; };                                // zero-coding context number, bit-sliced
    struct RowSgn { u64 c0, c1, c2, sb; };                                // sign context number and the coded bit (sign ^ xor bit)

    // at-visit significance of the eight neighbours of every column of a row -> table D-1
    T1_HD RowCtx zc_bits(u64 l, u64 r, u64 u, u64 d, u64 ul, u64 ur, u64 dl, u64 dr) const
    {
        u64 h0 = l ^ r, h1 = l & r, v0 = u ^ d, v1 = u & d;
        const u64 a = ul ^ ur, b = ul & ur, c = dl ^ dr, e = dl & dr, k = a & c;
        const u64 d0 = a ^ c, d1 = b ^ e ^ k, d2 = b & e;
        const u64 dge1 = d0 | d1 | d2, dge2 = d1 | d2, dge3 = (d1 & d0) | d2;
        RowCtx o;
        if (orient == 3) {
            const u64 hvge1 = h0 | h1 | v0 | v1, hvge2 = h1 | v1 | (h0 & v0);
            const u64 e0 = ~dge1, e1 = dge1 & ~dge2, e2 = dge2 & ~dge3;
            o.n3 = dge3;
            o.n2 = e2 | (e1 & hvge1);
            o.n1 = e2 | (e1 & ~hvge1) | (e0 & hvge2);
            o.n0 = (e2 & hvge1) | (e1 & hvge2) | (e1 & ~hvge1) | (e0 & hvge1 & ~hvge2);
        } else {
            if (orient == 1) { u64 t = h0; h0 = v0; v0 = t; t = h1; h1 = v1; v1 = t; }
            const u64 hz = ~(h0 | h1), vz = ~(v0 | v1);
            o.n3 = h1;
            o.n2 = h0 | (hz & v1);
            o.n1 = (h0 & (~vz | dge1)) | (hz & (v0 | (vz & dge2)));
            o.n0 = (h0 & (~vz | ~dge1)) | (hz & (v0 | (vz & dge1 & ~dge2)));
        }
        return o;
    }
    // horizontal / vertical neighbours: significance at visit time and their signs -> tables D-2, D-3
    T1_HD static RowSgn sc_bits(u64 l, u64 ln, u64 r, u64 rn, u64 u, u64 un, u64 d, u64 dn, u64 sgn)
    {
        const u64 lp = l & ~ln, lm = l & ln, rp = r & ~rn, rm = r & rn;
        const u64 up = u & ~un, um = u & un, dp = d & ~dn, dm = d & dn;
        const u64 hp = (lp | rp) & ~(lm | rm), hn = (lm | rm) & ~(lp | rp), hz = ~(hp | hn);
        const u64 vp = (up | dp) & ~(um | dm), vn = (um | dm) & ~(up | dp), vz = ~(vp | vn);
        const u64 n4 = (hp & vp) | (hn & vn), n3 = (hp | hn) & vz, n2 = (hp & vn) | (hn & vp), n1 = hz & (vp | vn);
        RowSgn o;
        o.c2 = n4; o.c1 = n3 | n2; o.c0 = n3 | n1;
        o.sb = sgn ^ (hn | (hz & vn));
        return o;
    }
    T1_HD static uint32_t bit_at(u64 m, int x) { return (uint32_t) (m >> x) & 1u; }

    // contexts of all four rows given the samples that become significant in this pass (n[r], signs from sgn)
    T1_HD void row_contexts(const Stripe &sp, const u64 n[4], RowCtx zc[4], RowSgn sc[4]) const
    {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const u64 nu = r > 0 ? n[r - 1] : 0ull, nd = r < 3 ? n[r + 1] : 0ull;     // new significance of the rows above / below
            const u64 up = sp.s[r] | nu, mid = sp.s[r + 1] | n[r], dn = sp.s[r + 2];
            // earlier in the scan: the row above in this column, everything in the previous column
            zc[r] = zc_bits(mid << 1, sp.s[r + 1] >> 1, up, dn, up << 1, sp.s[r] >> 1, (dn | nd) << 1, dn >> 1);
            const u64 upn = sp.neg[r] | (nu & (r > 0 ? sp.sgn[r > 0 ? r - 1 : 0] : 0ull));
            const u64 midn = sp.neg[r + 1] | (n[r] & sp.sgn[r]);
            sc[r] = sc_bits(mid << 1, midn << 1, sp.s[r + 1] >> 1, sp.neg[r + 1] >> 1, up, upn, dn, sp.neg[r + 2], sp.sgn[r]);
        }
    }

    // ---- per-stripe pieces of the bit-parallel passes: membership masks, decision counts, column emission.
    // The column loops work on 32-bit halves of the row masks (a 64-bit variable shift is a quarter-rate
    // instruction on gfx950; v_bfe_u32 on a half is full rate).
    struct Lo { T1_HD static uint32_t of(u64 v) { return (uint32_t) v; } };
    struct Hi { T1_HD static uint32_t of(u64 v) { return (uint32_t) (v >> 32); } };
    T1_HD static uint32_t bit32(uint32_t m, int x) { return (m >> x) & 1u; }
    T1_HD static int ctz32(uint32_t v)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        return __ffs((int) v) - 1;
#else
        return __builtin_ctz(v);
#endif
    }
    T1_HD static int popc64(u64 v)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        return __popcll(v);
#else
        return __builtin_popcountll(v);
#endif
    }

    // propagation pass over one stripe: m = samples coded, n = samples that become significant
    T1_HD void sigprop_members(const Stripe &sp, const u64 b[4], u64 m[4], u64 n[4]) const
    {
        u64 cand[4], p[4], nbo[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            cand[r] = ~sp.s[r + 1] & (r < sp.nrows ? sp.wmask : 0ull);
            p[r] = cand[r] & b[r];
            nbo[r] = neighbours(sp.s[r], sp.s[r + 1], sp.s[r + 2]);
            n[r] = 0;
        }
        bool changed;
        do {
            changed = false;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const u64 ext = (r > 0 ? n[r - 1] | (n[r - 1] << 1) : 0ull) | (r < 3 ? n[r + 1] << 1 : 0ull);
                const u64 v = flood(p[r], p[r] & (nbo[r] | ext));
                changed |= v != n[r];
                n[r] = v;
            }
        } while (changed);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const u64 ext = (r > 0 ? n[r - 1] | (n[r - 1] << 1) : 0ull) | (r < 3 ? n[r + 1] << 1 : 0ull);
            m[r] = cand[r] & (nbo[r] | ext | (n[r] << 1));
        }
    }
    T1_HD static uint32_t sigprop_count(const u64 m[4], const u64 n[4])
    {
        return (uint32_t) (popc64(m[0]) + popc64(m[1]) + popc64(m[2]) + popc64(m[3]) + popc64(n[0]) + popc64(n[1]) + popc64(n[2]) + popc64(n[3]));
    }
    template <class H, class Em>
    T1_HD static void sigprop_cols(Em &em, const u64 m[4], const u64 b[4], const RowCtx zc[4], const RowSgn sc[4])
    {
        uint32_t m_[4], b_[4], z0[4], z1[4], z2[4], z3[4], c0[4], c1[4], c2[4], sb[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            m_[r] = H::of(m[r]); b_[r] = H::of(b[r]);
            z0[r] = H::of(zc[r].n0); z1[r] = H::of(zc[r].n1); z2[r] = H::of(zc[r].n2); z3[r] = H::of(zc[r].n3);
            c0[r] = H::of(sc[r].c0); c1[r] = H::of(sc[r].c1); c2[r] = H::of(sc[r].c2); sb[r] = H::of(sc[r].sb);
        }
        uint32_t pending = m_[0] | m_[1] | m_[2] | m_[3];
        while (pending) {
            T1_STAT(0);
            const int x = ctz32(pending);
            pending &= pending - 1;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t on = bit32(m_[r], x), d = bit32(b_[r], x);
                em.emit_if(on != 0, bit32(z0[r], x) | (bit32(z1[r], x) << 1) | (bit32(z2[r], x) << 2) | (bit32(z3[r], x) << 3), d);
                em.emit_if((on & d) != 0, (uint32_t) CTX_SC0 + (bit32(c0[r], x) | (bit32(c1[r], x) << 1) | (bit32(c2[r], x) << 2)), bit32(sb[r], x));
            }
            em.column_end();
        }
    }
    template <class Em>
    T1_HD void sigprop_emit(Em &em, const Stripe &sp, const u64 b[4], const u64 m[4], const u64 n[4]) const
    {
        if ((m[0] | m[1] | m[2] | m[3]) == 0) return;
        RowCtx zc[4];
        RowSgn sc[4];
        row_contexts(sp, n, zc, sc);
        sigprop_cols<Lo>(em, m, b, zc, sc);
        sigprop_cols<Hi>(em, m, b, zc, sc);
    }

    // cleanup pass over one stripe: m = samples coded here, n = those that become significant, agg = columns in
    // run-length mode (all four samples coded here, nothing significant in the 3 x 6 neighbourhood on entry)
    T1_HD void cleanup_members(const Stripe &sp, const u64 b[4], bool full, u64 m[4], u64 n[4], u64 &agg) const
    {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            m[r] = ~(sp.s[r + 1] | sp.vis[r]) & (r < sp.nrows ? sp.wmask : 0ull);
            n[r] = m[r] & b[r];
        }
        agg = 0;
        if (full) {
            u64 any = (n[0] | n[1] | n[2] | n[3]) << 1;
#pragma unroll
            for (int r = 0; r < 6; r++) any |= sp.s[r] | (sp.s[r] << 1) | (sp.s[r] >> 1);
            agg = m[0] & m[1] & m[2] & m[3] & ~any;
        }
    }
    // decisions of the stripe: AGG per run-length column, two UNIFORM bits when the run ends inside it, a zero-coding
    // decision for every coded sample that is not covered by the run, a sign for every newly significant one
    T1_HD static uint32_t cleanup_count(const u64 m[4], const u64 n[4], const u64 b[4], u64 agg)
    {
        const u64 hit = agg & (b[0] | b[1] | b[2] | b[3]);
        uint32_t c = (uint32_t) (popc64(agg) + 2 * popc64(hit));
        u64 prior = 0;                                                   // a row above in this column has its bit set
#pragma unroll
        for (int r = 0; r < 4; r++) {
            c += (uint32_t) (popc64(m[r] & ~(agg & ~prior)) + popc64(n[r]));
            prior |= b[r];
        }
        return c;
    }
    template <class H, class Em>
    T1_HD static void cleanup_cols(Em &em, const u64 m[4], const u64 b[4], u64 agg, const RowCtx zc[4], const RowSgn sc[4])
    {
        uint32_t m_[4], b_[4], z0[4], z1[4], z2[4], z3[4], c0[4], c1[4], c2[4], sb[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            m_[r] = H::of(m[r]); b_[r] = H::of(b[r]);
            z0[r] = H::of(zc[r].n0); z1[r] = H::of(zc[r].n1); z2[r] = H::of(zc[r].n2); z3[r] = H::of(zc[r].n3);
            c0[r] = H::of(sc[r].c0); c1[r] = H::of(sc[r].c1); c2[r] = H::of(sc[r].c2); sb[r] = H::of(sc[r].sb);
        }
        const uint32_t agg_ = H::of(agg);
        uint32_t pending = m_[0] | m_[1] | m_[2] | m_[3];
        while (pending) {
            T1_STAT(2);
            const int x = ctz32(pending);
            pending &= pending - 1;
            const uint32_t a = bit32(agg_, x);
            const uint32_t d0 = bit32(b_[0], x), d1 = bit32(b_[1], x), d2 = bit32(b_[2], x), d3 = bit32(b_[3], x);
            const uint32_t run = d0 ? 0u : (d1 ? 1u : (d2 ? 2u : (d3 ? 3u : 4u)));
            const bool hit = a && run != 4u;
            em.emit_if(a != 0, (uint32_t) CTX_AGG, run != 4u ? 1u : 0u);
            em.emit_if(hit, (uint32_t) CTX_UNI, run >> 1);
            em.emit_if(hit, (uint32_t) CTX_UNI, run & 1u);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t on = bit32(m_[r], x), d = bit32(b_[r], x);
                const bool zc_on = on && !(a && (uint32_t) r <= run);          // rows up to the run's end carry no own decision
                em.emit_if(zc_on, bit32(z0[r], x) | (bit32(z1[r], x) << 1) | (bit32(z2[r], x) << 2) | (bit32(z3[r], x) << 3), d);
                em.emit_if((on & d) != 0, (uint32_t) CTX_SC0 + (bit32(c0[r], x) | (bit32(c1[r], x) << 1) | (bit32(c2[r], x) << 2)), bit32(sb[r], x));
            }
            em.column_end();
        }
    }
    template <class Em>
    T1_HD void cleanup_emit(Em &em, const Stripe &sp, const u64 b[4], const u64 m[4], const u64 n[4], u64 agg) const
    {
        if ((m[0] | m[1] | m[2] | m[3]) == 0) return;
        RowCtx zc[4];
        RowSgn sc[4];
        row_contexts(sp, n, zc, sc);
        cleanup_cols<Lo>(em, m, b, agg, zc, sc);
        cleanup_cols<Hi>(em, m, b, agg, zc, sc);
    }

    // refinement pass over one stripe: m = samples significant before this plane, ref = refined before, nb = "has a
    // significant neighbour"
    template <class H, class Em>
    T1_HD static void refine_cols(Em &em, const u64 m[4], const u64 b[4], const u64 ref[4], const u64 nb[4])
    {
        uint32_t m_[4], b_[4], r_[4], n_[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { m_[r] = H::of(m[r]); b_[r] = H::of(b[r]); r_[r] = H::of(ref[r]); n_[r] = H::of(nb[r]); }
        uint32_t pending = m_[0] | m_[1] | m_[2] | m_[3];
        while (pending) {
            T1_STAT(1);
            const int x = ctz32(pending);
            pending &= pending - 1;
#pragma unroll
            for (int r = 0; r < 4; r++)
                em.emit_if(bit32(m_[r], x) != 0, (uint32_t) CTX_MAG0 + (bit32(r_[r], x) ? 2u : bit32(n_[r], x)), bit32(b_[r], x));
            em.column_end();
        }
    }
    template <class Em>
    T1_HD void refine_emit(Em &em, const Stripe &sp, const u64 b[4], const u64 ref[4], u64 m[4]) const
    {
        u64 nb[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            m[r] = sp.s[r + 1] & ~sp.vis[r];                              // significant before this plane
            nb[r] = neighbours(sp.s[r], sp.s[r + 1], sp.s[r + 2]);
        }
        if ((m[0] | m[1] | m[2] | m[3]) == 0) return;
        refine_cols<Lo>(em, m, b, ref, nb);
        refine_cols<Hi>(em, m, b, ref, nb);
    }

    T1_HD void sigprop_bits(int plane, int ystart = 0)
    {
        for (int y0 = ystart; y0 < h; y0 += 4) {
            stripe_hook(y0);
            Stripe sp;
            load(sp, y0);
            const u64 b[4] = {st.BP(plane, y0), st.BP(plane, y0 + 1), st.BP(plane, y0 + 2), st.BP(plane, y0 + 3)};
            u64 n[4], m[4];
            sigprop_members(sp, b, m, n);
            sigprop_emit(mq, sp, b, m, n);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                sp.s[r + 1] |= n[r];
                sp.neg[r + 1] |= n[r] & sp.sgn[r];
                sp.vis[r] |= m[r];
                sp.sps[r] |= n[r];
            }
            store(sp, y0);
        }
    }

    T1_HD void cleanup_bits(int plane, int ystart = 0)
    {
        for (int y0 = ystart; y0 < h; y0 += 4) {
            stripe_hook(y0);
            Stripe sp;
            load(sp, y0);
            const u64 b[4] = {st.BP(plane, y0), st.BP(plane, y0 + 1), st.BP(plane, y0 + 2), st.BP(plane, y0 + 3)};
            u64 m[4], n[4], agg;
            cleanup_members(sp, b, y0 + 3 < h, m, n, agg);
            cleanup_emit(mq, sp, b, m, n, agg);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                sp.s[r + 1] |= n[r];
                sp.neg[r + 1] |= n[r] & sp.sgn[r];
                sp.vis[r] = 0;                                           // the visited flags die with the plane
            }
            store(sp, y0);
        }
    }

    T1_HD void refine_bits(int plane, int ystart = 0)
    {
        for (int y0 = ystart; y0 < h; y0 += 4) {
            stripe_hook(y0);
            Stripe sp;
            load(sp, y0);
            const u64 b[4] = {st.BP(plane, y0), st.BP(plane, y0 + 1), st.BP(plane, y0 + 2), st.BP(plane, y0 + 3)};
            const u64 ref[4] = {st.REF(y0), st.REF(y0 + 1), st.REF(y0 + 2), st.REF(y0 + 3)};
            u64 m[4];
            refine_emit(mq, sp, b, ref, m);
#pragma unroll
            for (int r = 0; r < 4; r++) st.REF(y0 + r) = ref[r] | m[r];
        }
    }
};

// ------------------------------------------------------------------------------------------------
// whole code-block drivers
// ------------------------------------------------------------------------------------------------
struct EncodeResult {
    int totalpasses;
    int length;                 // bytes in the segment
};

// Encodes every pass of a code-block with `numbps` magnitude bit-planes (numbps >= 1).
// rates[p] follow OpenJPEG: bytes completed + 3 for unterminated passes, clipped to be non-decreasing,
// never ending on 0xFF (needs read access to the bytes: ByteAt(int) -> uint8_t).
template <class Store, class Sink, class ByteAt, class Observer, class Table = ConstTable>
T1_HD EncodeResult encode_block_observed(Store &st, Sink sink, ByteAt bytes, int w, int h, int orient, int numbps, int *rates,
                                         Observer &obs, Table tab = Table())
{
    MqEncoder<Sink, Table> mq{0, 0, 0, 0, 0, {0, 0, 0}, sink, tab, 0};
    mq.init();
    Passes<true, Store, MqEncoder<Sink, Table>, Observer> ps(st, mq, w, h, orient, &obs);
    int passno = 0, passtype = 2;
    for (int bp = numbps - 1; bp >= 0; passno++) {
        obs.pass_start(passno, mq);
        if (passtype == 0) { ps.sigprop(bp); obs.sigprop_done(bp, st); }
        else if (passtype == 1) ps.refine(bp);
        else ps.cleanup(bp);
        if (passtype == 2 && bp == 0) { mq.flush(); rates[passno] = mq.numbytes(); }
        else rates[passno] = (int) ((uint32_t) mq.numbytes() + 3u);
        if (++passtype == 3) { passtype = 0; bp--; }
    }
    int last = mq.numbytes();
    for (int p = passno; p > 0;) {
        --p;
        if (rates[p] > last) rates[p] = last; else last = rates[p];
    }
    for (int p = 0; p < passno; p++)
        if (rates[p] > 0 && bytes(rates[p] - 1) == 0xFF) rates[p]--;
    EncodeResult r;
    r.totalpasses = passno;
    r.length = mq.numbytes();
    return r;
}

template <class Store, class Sink, class ByteAt, class Table = ConstTable>
T1_HD EncodeResult encode_block(Store &st, Sink sink, ByteAt bytes, int w, int h, int orient, int numbps, int *rates,
                                Table tab = Table())
{
    NoObserver obs;
    return encode_block_observed(st, sink, bytes, w, h, orient, numbps, rates, obs, tab);
}

// MQ decoder registers at the start of a coding pass (taken while decoding the complete segment)
struct MqCheckpoint {
    uint32_t a, c;
    int ct, pos;
    u64 w0, w1, w2;
};

// ------------------------------------------------------------------------------------------------
// Decoder checkpoints without decoding.  The MQ decoder's registers at a pass boundary follow from the
// ENCODER's registers there plus the finished bytes:
//   * A and the context states evolve identically in both coders (C.2.2 / C.3.2 mirror each other);
//   * the decoder's C register only ever sees linear updates mod 2^32: "c <<= 1" per renormalisation shift,
//     "c += byte << 8|9" per BYTEIN, "c -= Qe << 16" whenever the symbol lies in the upper sub-interval - and
//     those are exactly the symbols for which the encoder does "c += Qe".  Hence
//         C_dec = Inject(shifts, bytes) - ((C_enc & 0xFFFF) << 16)   (mod 2^32)
//     where Inject is the decoder's C with the subtractions left out: a function of the number of shifts so
//     far (equal in both coders) and of the segment bytes only;
//   * CT and the byte position are the same function of (shifts, bytes).
// The encoder therefore stores {A, C_enc & 0xFFFF, shifts, contexts} at every pass start and one sweep over
// the bytes afterwards (O(bytes + passes), no symbols decoded) turns them into decoder registers.
// ------------------------------------------------------------------------------------------------
template <class Mq>
T1_HD MqCheckpoint encoder_checkpoint(const Mq &m)
{
    return MqCheckpoint{m.a, m.c & 0xFFFFu, 0, (int) m.shifts, m.cx.w0, m.cx.w1, m.cx.w2};
}

// CkArray: checkpoint storage of one code-block addressed by (pass, stripe):
//   uint32_t shifts(p, s), c16(p, s)            what encoder_checkpoint() left in .pos / .c
//   void finish(p, s, uint32_t c, int ct, int pos)   overwrite them with the decoder's registers
template <class CkArray, class Source>
T1_HD void finalize_checkpoints(CkArray &ck, int npasses, int nstripes, Source src)
{
    uint32_t c;
    int ct, pos = 0;
    auto bytein = [&]() {
        uint32_t cur = src.get(pos), nxt = src.get(pos + 1);
        if (cur == 0xFF) {
            if (nxt > 0x8F) { c += 0xFF00; ct = 8; }
            else { pos++; c += nxt << 9; ct = 7; }
        } else {
            pos++; c += nxt << 8; ct = 8;
        }
    };
    c = src.get(0) << 16;
    bytein();
    c <<= 7; ct -= 7;
    uint32_t done = 0;
    for (int p = 0; p < npasses; p++)
        for (int s = 0; s < nstripes; s++) {
            const uint32_t target = ck.shifts(p, s);
            uint32_t rem = target - done;
            done = target;
            while (rem) {
                if (ct == 0) bytein();
                uint32_t k = rem < (uint32_t) ct ? rem : (uint32_t) ct;
                c <<= k; ct -= (int) k; rem -= k;
            }
            ck.finish(p, s, c - (ck.c16(p, s) << 16), ct, pos);
        }
}

// ------------------------------------------------------------------------------------------------
// Two-phase encoder.  The coding passes are branchy (which samples are coded depends on the data) while the
// MQ coder is a long dependent chain; run together, a wave executes the union of both for all its lanes.
//   phase 1 (emit_block): the passes with a "coder" that only appends the decisions (context, bit) to a byte
//                         stream and notes where every (pass, stripe) starts in it;
//   phase 2 (mq_encode_stream): one tight loop per code-block that reads decisions and arithmetic-codes them,
//                         the same instruction sequence for every lane and every symbol kind.
// Bytes, rates and checkpoints are identical to encode_block_observed().
// ------------------------------------------------------------------------------------------------
// Stream bytes: a decision is context | bit << 5 (< 0x40); 0x80 marks the start of a stripe, 0xC0 the start of
// the first stripe of a coding pass - the stream coder checkpoints there and closes the previous pass.
// Put: void operator()(uint32_t index, uint32_t byte)
constexpr uint32_t kSymStripe = 0x80, kSymPass = 0x40;
template <class Put>
struct SymbolEmitter {
    uint32_t n;
    Put put;
    T1_HD void emit(int ctx, int d) { put(n, (uint32_t) ctx | ((uint32_t) d << 5)); n++; }
    // the same when `on`, nothing otherwise - without a branch around it (bit-parallel passes below)
    T1_HD void emit_if(bool on, uint32_t ctx, uint32_t d) { put.put_if(on, n, ctx | (d << 5)); n += on ? 1u : 0u; }
    T1_HD void column_end() {}
    T1_HD void mark(bool new_pass) { put(n, kSymStripe | (new_pass ? kSymPass : 0u)); n++; }
    T1_HD void encode_zc(int ctx, int d) { emit(ctx, d); }
    T1_HD void encode_sc(int ctx, int d) { emit(ctx, d); }
    T1_HD void encode_mag(int ctx, int d) { emit(ctx, d); }
    T1_HD void encode_agg(int d) { emit(CTX_AGG, d); }
    T1_HD void encode_uni(int d) { emit(CTX_UNI, d); }
};

// Observer of emit_block(): stripe markers into the stream; Inner gets the remaining hooks
template <class Inner>
struct MarkingObserver {
    Inner &in;
    template <class Em>
    T1_HD void pass_start(int p, const Em &e) { in.pass_start(p, e); }
    template <class Em>
    T1_HD void stripe_start(int y0, Em &e) { e.mark(y0 == 0); }
    template <class Store>
    T1_HD void sigprop_done(int bp, Store &st) { in.sigprop_done(bp, st); }
};

#ifdef EBCC_T1_PROFILE
__device__ long long t1_profile[4];
#endif
// Returns the number of coding passes; the length of the stream is in em.n afterwards.
template <bool BITS = true, class Store, class Put, class Observer>
T1_HD int emit_block(Store &st, SymbolEmitter<Put> &em, int w, int h, int orient, int numbps, Observer &inner)
{
    MarkingObserver<Observer> obs{inner};
    Passes<true, Store, SymbolEmitter<Put>, MarkingObserver<Observer>> ps(st, em, w, h, orient, &obs);
    int passno = 0, passtype = 2;
    for (int bp = numbps - 1; bp >= 0; passno++) {
        obs.pass_start(passno, em);
#ifdef EBCC_T1_PROFILE
        const long long t0_ = wall_clock64();
#endif
        if (passtype == 0) { if constexpr (BITS) ps.sigprop_bits(bp); else ps.sigprop(bp); obs.sigprop_done(bp, st); }
        else if (passtype == 1) { if constexpr (BITS) ps.refine_bits(bp); else ps.refine(bp); }
        else { if constexpr (BITS) ps.cleanup_bits(bp); else ps.cleanup(bp); }
#ifdef EBCC_T1_PROFILE
        t1_profile[passtype] += wall_clock64() - t0_;
#endif
        if (++passtype == 3) { passtype = 0; bp--; }
    }
    return passno;
}

// SymSrc: uint32_t get(uint32_t index), called with index = 0, 1, 2, ... (every lane of a wave is at the same
//         index, which lets a device source refill a staging buffer for all lanes at once);
// CtxMem: uint32_t ld(int ctx), void st(int ctx, uint32_t state), void words(uint32_t out[5]) - the 19 context
//         states, one byte each (6-bit table index | mps << 6); LDS on the device;
// Sink:   void put(int index, uint32_t byte), ignored for index < 0, indices come in increasing order (the last
//         one possibly twice); void finish() once at the end;
// CkArray additionally: void store(p, s, a, c16, shifts, const uint32_t cx[5]).
// The coder is written for SIMT execution: one code path for MPS and LPS, renormalisation by count, a
// branch-free BYTEOUT, and the next decision / context state fetched while the current one is coded.
template <class SymSrc, class CtxMem, class Sink, class ByteAt, class CkArray, class Table = ConstTable>
T1_HD EncodeResult mq_encode_stream(SymSrc sym, CtxMem cx, uint32_t nsym, int npasses, Sink sink, ByteAt bytes, int *rates,
                                    CkArray &ck, Table tab = Table())
{
    uint32_t a = 0x8000, c = 0, cur = 0, shifts = 0;
    int ct = 12, n = -1;
    for (int i = 0; i < NCTX; i++) cx.st(i, i == CTX_UNI ? 46u : (i == CTX_AGG ? 3u : (i == CTX_ZC0 ? 4u : 0u)));
    auto byteout = [&]() {
        // C.2.7 without branches: a carry goes into the byte being held unless that byte is 0xFF; a byte that is
        // (or becomes) 0xFF is followed by a 7-bit byte
        const uint32_t carry = cur != 0xFFu ? (c >> 27) & 1u : 0u;
        cur += carry;
        c &= ~(carry << 27);
        const bool stuff = cur == 0xFFu;
        sink.put(n, cur);
        n++;
        const int sh = stuff ? 20 : 19;
        cur = (c >> sh) & 0xFFu;
        c &= (1u << sh) - 1u;
        ct = stuff ? 7 : 8;
    };
    int p = 0, s = -1;
    uint32_t v = nsym ? sym.get(0) : 0u;
    uint32_t st = cx.ld((int) (v & 31u));
    for (uint32_t pos = 0; pos < nsym; pos++) {
        const uint32_t vn = pos + 1 < nsym ? sym.get(pos + 1) : 0u;
        if (v & kSymStripe) {
            if (v & kSymPass) {
                if (s >= 0) { rates[p] = (int) ((uint32_t) n + 3u); p++; }
                s = 0;
            } else {
                s++;
            }
            uint32_t x[5];
            cx.words(x);
            ck.store(p, s, a, c & 0xFFFFu, shifts, x);
        } else {
            const uint32_t e = tab((int) (st & 0x3Fu)), qe = e & 0xFFFFu;
            const uint32_t mps = st >> 6, lps = (v >> 5) ^ mps;
            a -= qe;
            const bool renorm = lps || (a & 0x8000u) == 0;
            const bool lower = (a < qe) != (lps != 0);                    // which sub-interval the symbol is coded in
            if (lower) a = qe; else c += qe;
            if (renorm) {
                const uint32_t nidx = lps ? (e >> 22) & 0x3Fu : (e >> 16) & 0x3Fu;
                cx.st((int) (v & 31u), nidx | ((mps ^ (lps & (e >> 28))) << 6));
            }
            int k = renorm_shifts(a);                                     // 0 when bit 15 is set
            shifts += (uint32_t) k;
            while (k >= ct) { a <<= ct; c <<= ct; k -= ct; byteout(); }
            a <<= k; c <<= k; ct -= k;
        }
        v = vn;
        st = cx.ld((int) (v & 31u));
    }
    {                                                                     // FLUSH (C.2.9)
        const uint32_t tempc = c + a;
        c |= 0xFFFFu;
        if (c >= tempc) c -= 0x8000u;
        c <<= ct; byteout();
        c <<= ct; byteout();
        sink.put(n, cur);
        if (cur != 0xFFu) n++;
        sink.finish();                                                    // (a buffering sink writes out its tail)
    }
    rates[p] = n;
    int last = n;
    for (int q = npasses; q > 0;) {
        --q;
        if (rates[q] > last) rates[q] = last; else last = rates[q];
    }
    for (int q = 0; q < npasses; q++)
        if (rates[q] > 0 && bytes(rates[q] - 1) == 0xFF) rates[q]--;
    EncodeResult r;
    r.totalpasses = npasses;
    r.length = n;
    return r;
}

// ------------------------------------------------------------------------------------------------
// Segmented two-phase encoder: the decision stream of a code-block cut into SEGMENTS, one per (bit-plane,
// pass type, stripe).  Everything a segment needs follows from the encoder's masks alone:
//     significant before plane p          SUF[p + 1]          (OR of the bit-plane masks above p)
//     coded by plane p's propagation pass VISP[p]             (scan_block below: mask arithmetic, no decisions)
//     significant after it                SUF[p + 1] | (VISP[p] & BP[p])
//     refined before plane p              SUF[p + 2]
// so the segments of a code-block can be produced in any order - one wave per (64 code-blocks, plane) on the
// device.  A segment is padded to whole rows of 16 decisions and flagged in-band, so the MQ pass is a loop over
// rows in which stripe checkpoints and pass boundaries are handled once per row, not per decision; the inner loop
// is the arithmetic coder and nothing else, the same instruction stream for all 64 code-blocks of a wave.
//   scan_block            -> VISP, SPS and the number of decisions of every segment
//   emit_stripe_segments  -> the decisions of the three segments of one (plane, stripe)
//   mq_rows_interval + MqCodeChain -> bytes, pass rates and checkpoints (identical to encode_block_observed)
// ------------------------------------------------------------------------------------------------
constexpr int kSegPlanes = 26;                                            // == kJ2kMaxPlanes (j2k.hpp)
constexpr int kSegCount = kSegPlanes * 3 * 16;
T1_HD int seg_index(int p, int t, int s) { return ((kSegPlanes - 1 - p) * 3 + t) * 16 + s; }
T1_HD int seg_plane(int seg) { return kSegPlanes - 1 - seg / 48; }
T1_HD int seg_type(int seg) { return (seg >> 4) % 3; }
// does code-block (P planes, nstr stripes) have segment (p, t, s)?  (its top plane has a cleanup pass only)
T1_HD bool seg_valid(int P, int nstr, int p, int t, int s) { return p < P && (p < P - 1 || t == 2) && s < nstr; }

// Masks: u64 BP(int plane, int y), SUF(int plane, int y), SGN(int y), VISP(int plane, int y) for y in [0, 64) and
// planes in [0, kSegPlanes + 1]; rows at or beyond the block height are never asked for.
template <class Masks>
struct MaskRows {
    Masks &M;
    int h;
    T1_HD u64 bp(int p, int y) const { return (y >= 0 && y < h) ? M.BP(p, y) : 0ull; }
    T1_HD u64 suf(int p, int y) const { return (y >= 0 && y < h) ? M.SUF(p, y) : 0ull; }
    T1_HD u64 sgn(int y) const { return (y >= 0 && y < h) ? M.SGN(y) : 0ull; }
    T1_HD u64 visp(int p, int y) const { return (y >= 0 && y < h) ? M.VISP(p, y) : 0ull; }
};
struct NullCoder {};

// Out: void visp(int plane, int y, u64 m), void sps_or(int y, u64 n), void len(int seg, uint32_t decisions)
template <class Masks, class Out>
T1_HD void scan_block(Masks &M, Out &out, int P, int w, int h, int orient)
{
    NullCoder nc;
    Passes<true, Masks, NullCoder> ps(M, nc, w, h, orient);
    const MaskRows<Masks> R{M, h};
    const int nstr = (h + 3) >> 2;
    for (int p = P - 1; p >= 0; p--) {
        const bool top = p == P - 1;
        Stripe sp;
        sp.wmask = w >= 64 ? ~0ull : ((1ull << w) - 1);
        if (!top) {
            u64 carry = 0;                                                // new significance of the row above the stripe
            for (int s = 0; s < nstr; s++) {
                const int y0 = 4 * s;
                sp.nrows = h - y0 < 4 ? h - y0 : 4;
#pragma unroll
                for (int r = 0; r < 6; r++) sp.s[r] = R.suf(p + 1, y0 - 1 + r);
                const u64 b[4] = {R.bp(p, y0), R.bp(p, y0 + 1), R.bp(p, y0 + 2), R.bp(p, y0 + 3)};
                const uint32_t nref = (uint32_t) (ps.popc64(sp.s[1]) + ps.popc64(sp.s[2]) + ps.popc64(sp.s[3]) + ps.popc64(sp.s[4]));
                sp.s[0] |= carry;
                u64 m[4], n[4];
                ps.sigprop_members(sp, b, m, n);
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (r < sp.nrows) { out.visp(p, y0 + r, m[r]); if (n[r]) out.sps_or(y0 + r, n[r]); }
                out.len(seg_index(p, 0, s), ps.sigprop_count(m, n));
                out.len(seg_index(p, 1, s), nref);
                carry = n[3];
            }
        }
        for (int s = 0; s < nstr; s++) {
            const int y0 = 4 * s;
            sp.nrows = h - y0 < 4 ? h - y0 : 4;
            sp.s[0] = R.suf(p, y0 - 1);                                   // the row above has been through this pass
#pragma unroll
            for (int r = 1; r < 6; r++) {
                const int y = y0 - 1 + r;
                sp.s[r] = R.suf(p + 1, y) | (top ? 0ull : (R.visp(p, y) & R.bp(p, y)));
            }
#pragma unroll
            for (int r = 0; r < 4; r++) sp.vis[r] = top ? 0ull : R.visp(p, y0 + r);
            const u64 b[4] = {R.bp(p, y0), R.bp(p, y0 + 1), R.bp(p, y0 + 2), R.bp(p, y0 + 3)};
            u64 m[4], n[4], agg;
            ps.cleanup_members(sp, b, y0 + 3 < h, m, n, agg);
            out.len(seg_index(p, 2, s), ps.cleanup_count(m, n, b, agg));
        }
    }
}

// Em: void begin(int seg), void emit_if(bool on, uint32_t ctx, uint32_t d), void column_end(), void end()
template <class Masks, class Em>
T1_HD void emit_stripe_segments(Masks &M, Em &em, int P, int p, int s, int w, int h, int orient)
{
    Passes<true, Masks, Em> ps(M, em, w, h, orient);
    const MaskRows<Masks> R{M, h};
    const bool top = p == P - 1;
    const int y0 = 4 * s;
    u64 s1[6], A[6], sg[6], vp[6];
#pragma unroll
    for (int r = 0; r < 6; r++) {
        const int y = y0 - 1 + r;
        s1[r] = R.suf(p + 1, y);
        sg[r] = R.sgn(y);
        vp[r] = top ? 0ull : R.visp(p, y);
        A[r] = s1[r] | (vp[r] & R.bp(p, y));
    }
    const u64 b[4] = {R.bp(p, y0), R.bp(p, y0 + 1), R.bp(p, y0 + 2), R.bp(p, y0 + 3)};
    Stripe sp;
    sp.wmask = w >= 64 ? ~0ull : ((1ull << w) - 1);
    sp.nrows = h - y0 < 4 ? h - y0 : 4;
#pragma unroll
    for (int r = 0; r < 4; r++) { sp.sgn[r] = sg[r + 1]; sp.sps[r] = 0; }
    u64 m[4], n[4];
    if (!top) {
        // propagation pass: the row above is through it, the rest as before the plane; nothing visited yet
        sp.s[0] = A[0];
#pragma unroll
        for (int r = 1; r < 6; r++) sp.s[r] = s1[r];
#pragma unroll
        for (int r = 0; r < 6; r++) sp.neg[r] = sp.s[r] & sg[r];
#pragma unroll
        for (int r = 0; r < 4; r++) sp.vis[r] = 0;
        em.begin(seg_index(p, 0, s));
        ps.sigprop_members(sp, b, m, n);
        ps.sigprop_emit(em, sp, b, m, n);
        em.end();
        // refinement pass: everything as after the propagation pass
#pragma unroll
        for (int r = 0; r < 6; r++) { sp.s[r] = A[r]; sp.neg[r] = A[r] & sg[r]; }
#pragma unroll
        for (int r = 0; r < 4; r++) sp.vis[r] = vp[r + 1];
        const u64 ref[4] = {R.suf(p + 2, y0), R.suf(p + 2, y0 + 1), R.suf(p + 2, y0 + 2), R.suf(p + 2, y0 + 3)};
        em.begin(seg_index(p, 1, s));
        ps.refine_emit(em, sp, b, ref, m);
        em.end();
    }
    // cleanup pass: the row above is through it (significant at or above this plane)
    sp.s[0] = R.suf(p, y0 - 1);
#pragma unroll
    for (int r = 1; r < 6; r++) sp.s[r] = A[r];
#pragma unroll
    for (int r = 0; r < 6; r++) sp.neg[r] = sp.s[r] & sg[r];
#pragma unroll
    for (int r = 0; r < 4; r++) sp.vis[r] = vp[r + 1];
    u64 agg;
    em.begin(seg_index(p, 2, s));
    ps.cleanup_members(sp, b, y0 + 3 < h, m, n, agg);
    ps.cleanup_emit(em, sp, b, m, n, agg);
    em.end();
}

// MQ state table of the row coder.  A context's state is kept as the CODE 8 * (6-bit state | mps << 6) - the byte
// offset of its entry in a table of 8-byte entries - and an entry holds Qe and BOTH successor codes, so a decision
// costs one table read, no sense/switch arithmetic and no address arithmetic:
//     word 0: qe     word 1: code(nmps, mps) | code(nlps, mps ^ switch) << 16
// Code kNullCode (state 47, unused by T.800) has Qe = 0 and itself as successor: coding its more probable symbol
// leaves every register of the coder unchanged.  Padding bytes of a row are decisions "0 in context kCtxNull",
// whose state is kNullCode for ever - so the coder needs no test for padding at all.
constexpr uint32_t kNullCode = 8u * 47u;
constexpr uint32_t kCtxNull = 19;
T1_HD uint32_t mq_code(uint32_t st7) { return st7 << 3; }
T1_HD uint32_t mq_code_state(uint32_t code) { return code >> 3; }
T1_HD void mq_entry2(int st7, uint32_t &qe, uint32_t &next)
{
    const int i = st7 & 63, mps = st7 >> 6;
    if (i >= 47) { qe = 0; next = kNullCode | (kNullCode << 16); return; }
    const uint32_t e = mq_entry(i);
    const uint32_t nm = ((e >> 16) & 0x3Fu) | ((uint32_t) mps << 6), nl = ((e >> 22) & 0x3Fu) | (((uint32_t) mps ^ (e >> 28)) << 6);
    qe = e & 0xFFFFu;
    next = mq_code(nm) | (mq_code(nl) << 16);
}
struct ConstTable2 {
    T1_HD void operator()(uint32_t code, uint32_t &qe, uint32_t &next) const { mq_entry2((int) mq_code_state(code), qe, next); }
};
// The row coder keeps every context as ONE word, qe | code << 16 - Qe of its current state is in a register the moment the
// slot is read - and a table entry holds what a transition needs: both successor codes and both successors' Qe:
//     entry(code): n[0] = code(nmps) | code(nlps) << 16     n[1] = qe(nmps) | qe(nlps) << 16
T1_HD uint32_t mq_slot_word(uint32_t code)
{
    uint32_t qe, nxt;
    mq_entry2((int) mq_code_state(code), qe, nxt);
    return qe | (code << 16);
}
T1_HD void mq_entry_next(uint32_t code, uint32_t &nxt, uint32_t &nqes)
{
    uint32_t qe, qm, ql, t;
    mq_entry2((int) mq_code_state(code), qe, nxt);
    mq_entry2((int) mq_code_state(nxt & 0xFFFFu), qm, t);
    mq_entry2((int) mq_code_state(nxt >> 16), ql, t);
    nqes = qm | (ql << 16);
}
struct ConstTableNext {
    T1_HD void operator()(uint32_t code, uint32_t &nxt, uint32_t &nqes) const { mq_entry_next(code, nxt, nqes); }
};

// Row format of the decision streams: a code-block's decisions, segment after segment in coding order, in ROWS of
// 16 bytes; every segment starts a new row (at least one, even when it has no decision), so a row never straddles
// two segments.  Byte = context | bit << 5; padding = kRowPad (context kCtxNull, bit 0: a no-op for the coder, see
// above); bit 6 of byte 0 set = first row of a segment (the coder checkpoints there and, at the first stripe of a
// pass, closes the previous pass).
constexpr uint32_t kRowPad = 0x80u | kCtxNull, kRowStart = 0x40u;
T1_HD uint32_t seg_rows(uint32_t decisions) { return decisions ? (decisions + 15u) >> 4 : 1u; }

// The MQ pass is TWO dependency chains per decision that meet nowhere:
//   * the INTERVAL chain: context state -> table entry (Qe, successors) -> A -> renormalisation shift count k, and
//     whether Qe is added to the code register (the decision falls into the upper sub-interval);
//   * the CODE chain: C += addend, k shifts, output bytes whenever the down-counter runs out (C.2.6, C.2.7).
// The code chain needs from the interval chain 12 bits per decision - the context's state before the decision (Qe
// follows from it), the "no addend" flag and k - so the two run as two waves of one workgroup, a few rows apart, with
// a 16-bit word per decision handed over through LDS (mq_rows_interval / MqCodeChain).  A lone wave issues one
// instruction every ~4-6 cycles whatever it does; two waves halve what each has to issue per decision.
//
// Hand-over word (32 bits): bits 0..15 Qe of the decision, bits 16..19 = k, bit 20 = nothing is added to C, bit 21 (first
// decision of a row only) = the row starts a segment.  (Round 2 handed the state over in 16 bits and the code chain
// looked Qe up again: a table read and its wait per decision on that wave too.)
constexpr uint32_t kHandNoAdd = 1u << 20, kHandStart = 1u << 21;

// RowSrc: uint32_t rows()                      rows of this lane's code-block
//         uint32_t wave_rows()                 the most rows any lane of the wave has (uniform)
//         void sync(uint32_t row)              called by every lane before row `row` is loaded (a uniform point)
//         void load(uint32_t row, uint32_t w[4])
// Slots:  uint32_t handle(uint32_t ctx) (an address on the device), uint32_t ld(handle) / void st(handle, word): the slot
//         words (qe | code << 16) of contexts 0 .. 19 (19 is the null context), void words(uint32_t out[5]): the 19
//         context states as bytes (state | mps << 6), four per word - the checkpoint format
// Table:  void operator()(code, nxt, nqes): mq_entry_next
// Hand:   void put(uint32_t row, int j, uint32_t word)
// CkArray: store_interval(p, s, a, shifts, cx[5])
//
// The wave is an in-order machine: whatever a decision waits for, it waits for in full.  Round 2 read the context's state
// and then its table entry, one after the other, at the head of every decision (~650 cycles per decision for a lone
// wave).  Now nothing a decision needs at its head comes out of LDS at that moment: Qe and the code are one slot word
// that was asked for a decision earlier (or is forwarded from the decision before when it used the same context), the
// table entry - only needed at the END of the decision, for the transition - is asked for at the end of the decision
// before, and the slot store that follows a transition depends on no read at all (the successor's Qe is in the entry).
template <class RowSrc, class Slots, class Hand, class CkArray, class Table = ConstTableNext>
T1_HD uint32_t mq_rows_interval(RowSrc src, Slots cx, int P, int nstr, Hand hand, CkArray &ck, Table tab = Table())
{
    uint32_t a = 0x8000, shifts = 0;
    for (uint32_t i = 0; i <= kCtxNull; i++)
        cx.st(cx.handle(i), mq_slot_word(i >= (uint32_t) NCTX ? kNullCode : mq_code(i == CTX_UNI ? 46u : (i == CTX_AGG ? 3u : (i == CTX_ZC0 ? 4u : 0u)))));
    const uint32_t nrows = P > 0 ? src.rows() : 0u, wrows = src.wave_rows();
    int pass = 0, stripe = 0;
    for (uint32_t row = 0; row < wrows; row++) {
        src.sync(row);                                                    // (uniform: the device source swaps its staging buffers here)
        if (row < nrows) {
            uint32_t w[4];
            src.load(row, w);
            const bool start = (w[0] & kRowStart) != 0;
            if (start) {
                uint32_t x[5];
                cx.words(x);
                ck.store_interval(pass, stripe, a, shifts, x);
                if (++stripe == nstr) { stripe = 0; pass++; }
            }
            uint32_t h = cx.handle(w[0] & 31u);
            uint32_t s0 = cx.ld(h);                                       // qe | code << 16 of this decision's context
            uint32_t nxt, nqes;
            tab(s0 >> 16, nxt, nqes);                                     // its transition (needed at the end of the decision)
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t wj = w[j >> 2] >> (8 * (j & 3));           // this decision's byte in bits 0..7
                uint32_t hn = 0, sn = 0;
                if (j < 15) {
                    hn = cx.handle((w[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 31u);
                    sn = cx.ld(hn);                                       // (issued before this decision's slot store)
                }
                const uint32_t qe = s0 & 0xFFFFu, code = s0 >> 16;
                const bool lps = (((wj << 4) ^ code) & 0x200u) != 0;      // decision bit (bit 5) against the mps (bit 9 of the code)
                a -= qe;
                const bool small = a < 0x8000u;
                const bool lower = (a < qe) != lps;                       // which sub-interval the symbol is coded in
                a = lower ? qe : a;
                const uint32_t ncode = lps ? nxt >> 16 : (small ? nxt & 0xFFFFu : code);
                const uint32_t nqe = lps ? nqes >> 16 : (small ? nqes & 0xFFFFu : qe);
                const uint32_t nword = nqe | (ncode << 16);
                cx.st(h, nword);
                if (j < 15) {
                    if (hn == h) sn = nword;                              // the next decision uses the same context
                    tab(sn >> 16, nxt, nqes);                             // the next decision's transition: a whole decision to arrive
                }
                const int k = renorm_shifts(a);                           // 0 when bit 15 is set
                shifts += (uint32_t) k;
                a <<= k;
                hand.put(row, j, qe | ((uint32_t) k << 16) | (lower ? kHandNoAdd : 0u) | ((j == 0 && start) ? kHandStart : 0u));
                h = hn; s0 = sn;
            }
        }
    }
    src.finish();
    return a;                                                             // (the code chain's FLUSH needs it)
}

// The code chain of one lane (Qe comes with the hand-over word).
// Sink:   void put(int index, uint32_t byte) (index -1 ignored; an index may be written again until a higher one
//         has been), void row_end(int n) (a uniform point: bytes below n are final and may leave), void finish(int n)
// CkArray: store_code(p, s, c16)
struct MqCodeChain {
    uint32_t c = 0, cur = 0;
    int ct = 12, n = -1, pass = 0, stripe = 0;
    template <class Sink>
    T1_HD void byteout(Sink &sink)
    {
        // C.2.7 without branches: a carry goes into the byte being held unless that byte is 0xFF; a byte that is
        // (or becomes) 0xFF is followed by a 7-bit byte
        const uint32_t carry = cur != 0xFFu ? (c >> 27) & 1u : 0u;
        cur += carry;
        c -= carry << 27;
        const bool stuff = cur == 0xFFu;
        sink.put(n, cur);
        n++;
        const int sh = stuff ? 20 : 19;
        cur = (c >> sh) & 0xFFu;
        c &= (1u << sh) - 1u;
        ct = stuff ? 7 : 8;
    }
    // one row of hand-over words (hw[j], j = 0..15); any(b): true if b holds for any lane of the wave
    template <class Sink, class CkArray, class Any>
    T1_HD void row(const uint32_t hw[16], int nstr, int *rates, Sink &sink, CkArray &ck, Any any)
    {
        if (hw[0] & kHandStart) {
            if (stripe == 0 && pass > 0) rates[pass - 1] = (int) ((uint32_t) n + 3u);
            ck.store_code(pass, stripe, c & 0xFFFFu);
            if (++stripe == nstr) { stripe = 0; pass++; }
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t qe = hw[j] & 0xFFFFu;
            const int k = (int) ((hw[j] >> 16) & 15u);
            c += (hw[j] & kHandNoAdd) ? 0u : qe;
            const bool need = k >= ct;                                    // a byte leaves when the down-counter runs out (selects, no branch)
            const int sh1 = need ? ct : k;
            int k2 = k - sh1;
            c <<= sh1;
            const uint32_t carry = (need && cur != 0xFFu) ? (c >> 27) & 1u : 0u;
            const uint32_t curc = cur + carry;
            c -= carry << 27;
            const bool stuff = curc == 0xFFu;
            sink.put(n, curc);                                            // (written again later unless `need`)
            n += need ? 1 : 0;
            const int shn = stuff ? 20 : 19;
            cur = need ? (c >> shn) & 0xFFu : cur;
            c = need ? c & ((1u << shn) - 1u) : c;
            ct = need ? (stuff ? 7 : 8) : ct - sh1;
            if (any(k2 >= ct)) {                                          // rare: a second byte (k <= 15: never a third)
                if (k2 >= ct) { c <<= ct; k2 -= ct; byteout(sink); }
            }
            c <<= k2; ct -= k2;
        }
        sink.row_end(n);
    }
    // FLUSH (C.2.9) and the pass rates (OpenJPEG: bytes completed + 3 for unterminated passes, clipped to be
    // non-decreasing, never ending on 0xFF); `a` = the interval register at the end (from the interval chain)
    template <class Sink, class ByteAt>
    T1_HD EncodeResult finish(uint32_t a, int npasses, int *rates, Sink &sink, ByteAt bytes)
    {
        EncodeResult r;
        r.totalpasses = npasses;
        r.length = 0;
        if (npasses <= 0) return r;
        const uint32_t tempc = c + a;
        c |= 0xFFFFu;
        if (c >= tempc) c -= 0x8000u;
        c <<= ct; byteout(sink);
        c <<= ct; byteout(sink);
        sink.put(n, cur);
        if (cur != 0xFFu) n++;
        sink.finish(n);                                                   // (a buffering sink writes out its tail)
        rates[npasses - 1] = n;
        int last = n;
        for (int q = npasses; q > 0;) {
            --q;
            if (rates[q] > last) rates[q] = last; else last = rates[q];
        }
        for (int q = 0; q < npasses; q++)
            if (rates[q] > 0 && bytes(rates[q] - 1) == 0xFF) rates[q]--;
        r.length = n;
        return r;
    }
};

// index of the first coding pass of bit-plane `bp` in a code-block with P planes; plane / type of pass i
T1_HD int first_pass_of_plane(int P, int bp) { return bp == P - 1 ? 0 : 3 * (P - 1 - bp) - 2; }
T1_HD int plane_of_pass(int P, int i) { return i == 0 ? P - 1 : P - 2 - (i - 1) / 3; }
T1_HD int type_of_pass(int i) { return i == 0 ? 2 : (i - 1) % 3; }          // 0 propagation, 1 refinement, 2 cleanup

template <class Store, class Source, class Observer, class Table = ConstTable, class CtxP = DirectCtx>
T1_HD void decode_block_observed(Store &st, Source src, int w, int h, int orient, int numbps, int npasses, Observer &obs,
                                 Table tab = Table(), CtxP cp = CtxP())
{
    MqDecoder<Source, Table> mq{0, 0, 0, 0, {0, 0, 0}, src, tab};
    mq.init();
    Passes<false, Store, MqDecoder<Source, Table>, Observer, CtxP> ps(st, mq, w, h, orient, &obs, cp);
    int passtype = 2, bp = numbps - 1;
    for (int p = 0; p < npasses && bp >= 0; p++) {
        obs.pass_start(p, mq);
        if (passtype == 0) { ps.sigprop(bp); obs.sigprop_done(bp, st); }
        else if (passtype == 1) ps.refine(bp);
        else ps.cleanup(bp);
        if (++passtype == 3) { passtype = 0; bp--; }
    }
}

template <class Store, class Source, class Table = ConstTable, class CtxP = DirectCtx>
T1_HD void decode_block(Store &st, Source src, int w, int h, int orient, int numbps, int npasses, Table tab = Table(), CtxP cp = CtxP())
{
    NoObserver obs;
    decode_block_observed(st, src, w, h, orient, numbps, npasses, obs, tab, cp);
}

// Decode passes [r, npasses) only.  The caller has put the store into the state the decoder has at the start
// of pass r (S/NEG/VIS/REF masks and the values of everything already significant); `ck` holds the MQ
// registers there (ignored for r == 0, where decoding starts afresh).  Valid whenever the checkpoint was taken
// with no byte at or beyond the (truncated) segment length consumed: ck.pos + 1 < length of src.
template <class Store, class Source, class Table = ConstTable, class CtxP = DirectCtx>
T1_HD void decode_resume(Store &st, Source src, int w, int h, int orient, int numbps, int npasses, int r, int stripe,
                         const MqCheckpoint &ck, Table tab = Table(), CtxP cp = CtxP())
{
    MqDecoder<Source, Table> mq{0, 0, 0, 0, {0, 0, 0}, src, tab};
    if (r == 0 && stripe == 0) mq.init();
    else { mq.a = ck.a; mq.c = ck.c; mq.ct = ck.ct; mq.pos = ck.pos; mq.cx.w0 = ck.w0; mq.cx.w1 = ck.w1; mq.cx.w2 = ck.w2; }
    Passes<false, Store, MqDecoder<Source, Table>, NoObserver, CtxP> ps(st, mq, w, h, orient, nullptr, cp);
    int bp = plane_of_pass(numbps, r), passtype = type_of_pass(r);
    int ystart = 4 * stripe;                                             // only the first pass starts mid-way
    for (int p = r; p < npasses && bp >= 0; p++) {
        if (passtype == 0) ps.sigprop(bp, ystart);
        else if (passtype == 1) ps.refine(bp, ystart);
        else ps.cleanup(bp, ystart);
        ystart = 0;
        if (++passtype == 3) { passtype = 0; bp--; }
    }
}

}  // namespace t1
}  // namespace ebcc
