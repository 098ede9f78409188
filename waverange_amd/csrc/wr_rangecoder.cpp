// wr_rangecoder.cpp -- see wr_rangecoder.h.
//
// Same arithmetic as rngcod13 (32-bit low/range, byte-wise renormalisation, carry through a
// run of 0xff bytes), organised differently:
//   * bytes go straight to the output; a carry walks back through the 0xff run already
//     written instead of being deferred in a (buffer, help) pair -- identical bytes, because
//     the run it increments is exactly the deferred run of rangecod.c:182-207;
//   * per coding block a 256-entry {cum, freq} table is built once; blocks of exactly 60000
//     symbols (all but the last) use a compile-time divisor so range/tot is a multiply;
//   * no intermediate copies (wrappers.cpp:119-121,137-139 copy every byte twice).
#include "wr_rangecoder.h"

#include <string.h>

#include <vector>

namespace wrrc {

namespace {

constexpr uint32_t kTop = 0x80000000u;     // rangecod.c:121  1 << (CODE_BITS-1)
constexpr uint32_t kBottom = 0x00800000u;  // rangecod.c:129  Top >> 8
constexpr int kShift = 23;                 // rangecod.c:127  CODE_BITS - 9
constexpr int kExtra = 7;                  // rangecod.c:128  (CODE_BITS-2) % 8 + 1

struct Enc {
    uint32_t low = 0, range = kTop, nbytes = 0;
    uint8_t* out;
    size_t pos;

    explicit Enc(uint8_t* o) : out(o), pos(1) { out[0] = 0; }  // start_encoding(rc, 0, 0)

    // add one to the byte before `pos` with ripple: rangecod.c:191-195 ("carry now")
    inline void carry()
    {
        size_t p = pos - 1;
        while (++out[p] == 0) p--;  // out[0] == 0 and the code value < 1: never runs off the front
    }
    // rangecod.c:182-207
    inline void renorm()
    {
        while (range <= kBottom) {
            if (low & kTop) carry();
            out[pos++] = (uint8_t)(low >> kShift);
            range <<= 8;
            low = (low << 8) & (kTop - 1);
            nbytes++;
        }
    }
    // rangecod.c:217-229
    inline void freq(uint32_t sy, uint32_t lt, uint32_t tot)
    {
        renorm();
        uint32_t r = range / tot, t = r * lt;
        low += t;
        range -= t;
        if (lt + sy < tot) range = r * sy;
    }
    // rangecod.c:231-245
    inline void shift(uint32_t sy, uint32_t lt, uint32_t sh)
    {
        renorm();
        uint32_t r = range >> sh, t = r * lt;
        low += t;
        if ((lt + sy) >> sh) range -= t; else range = r * sy;
    }
    // rangecod.c:254-276
    size_t finish()
    {
        renorm();
        nbytes += 5;
        uint32_t t = low >> kShift;
        if (!((low & (kBottom - 1)) < ((nbytes & 0xffffffu) >> 1))) t += 1;
        if (t > 0xff) carry();
        out[pos++] = (uint8_t)(t & 0xff);
        out[pos++] = (uint8_t)(nbytes >> 16);
        out[pos++] = (uint8_t)(nbytes >> 8);
        out[pos++] = (uint8_t)nbytes;
        return pos;
    }
};

struct SymEntry {
    uint32_t lt;    // cumulative count of smaller symbols
    uint32_t sy;    // count of this symbol
};

// symbols of one block; TOT > 0 selects the compile-time divisor
template <uint32_t TOT>
inline void encode_symbols(Enc& e, const uint8_t* s, uint32_t bs, const SymEntry* tab, uint32_t top_sym)
{
    // The renormalisation test is data dependent and badly predicted, so the common case (at
    // most one byte shifted out per symbol) is written branch-free: the byte is stored
    // unconditionally and `pos` advances by 0 or 1.  Two rare cases keep a branch: a pending
    // carry at the moment a byte leaves, and a second shift (symbol probability < 1/256).
    uint32_t low = e.low, range = e.range, nbytes = e.nbytes;
    uint8_t* out = e.out;
    size_t pos = e.pos;
    const uint32_t tot = TOT ? TOT : bs;
    for (uint32_t i = 0; i < bs; i++) {
        const uint32_t c = s[i];
        const uint32_t sh = range <= kBottom;  // 0 or 1
        if (__builtin_expect(sh & (low >> 31), 0)) {
            size_t p = pos - 1;
            while (++out[p] == 0) p--;
        }
        out[pos] = (uint8_t)(low >> kShift);
        pos += sh;
        nbytes += sh;
        low = sh ? (low << 8) & (kTop - 1) : low;
        range = sh ? range << 8 : range;
        while (__builtin_expect(range <= kBottom, 0)) {
            if (low & kTop) {
                size_t p = pos - 1;
                while (++out[p] == 0) p--;
            }
            out[pos++] = (uint8_t)(low >> kShift);
            range <<= 8;
            low = (low << 8) & (kTop - 1);
            nbytes++;
        }
        const uint32_t r = range / tot;
        const uint32_t t = r * tab[c].lt;
        low += t;
        // lt + sy < tot holds for every symbol except the largest one present (rangecod.c:227)
        range = (c != top_sym) ? r * tab[c].sy : range - t;
    }
    e.low = low; e.range = range; e.nbytes = nbytes; e.pos = pos;
}

inline void histogram(const uint8_t* s, uint32_t bs, uint32_t* h)
{
    uint32_t h1[256], h2[256], h3[256];
    memset(h, 0, 256 * sizeof(uint32_t));
    memset(h1, 0, sizeof h1); memset(h2, 0, sizeof h2); memset(h3, 0, sizeof h3);
    uint32_t i = 0;
    for (; i + 4 <= bs; i += 4) { h[s[i]]++; h1[s[i + 1]]++; h2[s[i + 2]]++; h3[s[i + 3]]++; }
    for (; i < bs; i++) h[s[i]]++;
    for (int b = 0; b < 256; b++) h[b] += h1[b] + h2[b] + h3[b];
}

}  // namespace

size_t encode_bound(size_t n)
{
    const size_t blocks = n / kBlock + 2;
    return n + n / 32 + blocks * 520 + 1024;
}

size_t encode_plane(const uint8_t* sym, size_t n, uint8_t* out, const uint16_t* hists)
{
    // block loop of wrappers.cpp:85-128: a full final block is followed by an empty one
    Enc e(out);
    size_t done = 0, blk = 0;
    for (;; blk++) {
        const size_t left = n - done;
        const uint32_t bs = left < kBlock ? (uint32_t)left : kBlock;
        const uint8_t* s = sym + done;
        e.freq(1, 1, 2);  // "a block follows"
        uint32_t h[256];
        if (hists) { for (int b = 0; b < 256; b++) h[b] = hists[blk * 256 + b]; }
        else histogram(s, bs, h);
        SymEntry tab[256];
        uint32_t cum = 0, top_sym = 0;
        for (int b = 0; b < 256; b++) {
            e.shift(1, h[b], 16);  // encode_short(count), rangecod.h:155
            tab[b].lt = cum; tab[b].sy = h[b];
            cum += h[b];
            if (h[b]) top_sym = (uint32_t)b;
        }
        if (bs == kBlock) encode_symbols<kBlock>(e, s, bs, tab, top_sym);
        else if (bs) encode_symbols<0>(e, s, bs, tab, top_sym);
        done += bs;
        if (bs < kBlock) break;
    }
    e.freq(1, 0, 2);  // "no more blocks"
    return e.finish();
}

namespace {

struct Dec {
    uint32_t low, range, help = 0;
    uint8_t held;
    const uint8_t* in;
    size_t len, pos = 0;

    inline uint32_t get() { return pos < len ? in[pos++] : (pos++, 0u); }
    // rangecod.c:282-291
    Dec(const uint8_t* i, size_t l) : in(i), len(l)
    {
        (void)get();
        held = (uint8_t)get();
        low = held >> (8 - kExtra);
        range = 1u << kExtra;
    }
    // rangecod.c:294-302
    inline void renorm()
    {
        while (range <= kBottom) {
            low = (low << 8) | (((uint32_t)held << kExtra) & 0xff);
            held = (uint8_t)get();
            low |= held >> (8 - kExtra);
            range <<= 8;
        }
    }
    inline uint32_t culfreq(uint32_t tot)  // rangecod.c:309-319
    {
        renorm();
        help = range / tot;
        uint32_t t = low / help;
        return t >= tot ? tot - 1 : t;
    }
    inline uint32_t culshift(uint32_t sh)  // rangecod.c:321-331
    {
        renorm();
        help = range >> sh;
        uint32_t t = low / help;
        return (t >> sh) ? (1u << sh) - 1 : t;
    }
    inline void update(uint32_t sy, uint32_t lt, uint32_t tot)  // rangecod.c:339-351
    {
        uint32_t t = help * lt;
        low -= t;
        if (lt + sy < tot) range = help * sy; else range -= t;
    }
};

template <uint32_t TOT>
inline size_t decode_symbols(Dec& d, uint8_t* dst, size_t room, uint32_t bs, const SymEntry* tab,
                             const uint8_t* lookup, uint32_t top_sym)
{
    const uint32_t tot = TOT ? TOT : bs;
    const uint32_t nout = bs < room ? bs : (uint32_t)room;
    uint32_t low = d.low, range = d.range;
    uint32_t held = d.held;
    const uint8_t* in = d.in;
    size_t pos = d.pos;
    const size_t len = d.len;
    for (uint32_t i = 0; i < bs; i++) {
        // first renormalisation step branch-free (see encode_symbols); the input byte is read
        // speculatively, which needs one readable byte at in[pos]: guaranteed while pos < len
        if (__builtin_expect(pos < len, 1)) {
            const uint32_t sh = range <= kBottom;
            const uint32_t nb = in[pos];
            const uint32_t l2 = (low << 8) | ((held << kExtra) & 0xff) | (nb >> (8 - kExtra));
            low = sh ? l2 : low;
            held = sh ? nb : held;
            range = sh ? range << 8 : range;
            pos += sh;
        }
        while (__builtin_expect(range <= kBottom, 0)) {
            low = (low << 8) | ((held << kExtra) & 0xff);
            held = pos < len ? in[pos] : 0;
            pos++;
            low |= held >> (8 - kExtra);
            range <<= 8;
        }
        const uint32_t help = range / tot;
        // low < range = help*tot + (range % tot)  =>  cf < tot + tot/help <= tot + tot^2/2^23 < tot + 430
        // for tot <= 60000.  rangecod.c:317 clamps cf to tot-1, which only ever selects the largest
        // symbol present; the lookup table is padded with that symbol instead (no clamp on the
        // per-symbol dependency chain: +5 % decode speed).
        const uint32_t cf = low / help;
        const uint32_t c = lookup[cf];
        const uint32_t t = help * tab[c].lt;
        low -= t;
        range = (c != top_sym) ? help * tab[c].sy : range - t;
        if (i < nout) dst[i] = (uint8_t)c;
    }
    d.low = low; d.range = range; d.held = (uint8_t)held; d.pos = pos;
    return bs;
}

}  // namespace

size_t decode_plane(const uint8_t* in, size_t len, uint8_t* sym, size_t n)
{
    // wrappers.cpp:153-224
    Dec d(in, len);
    size_t produced = 0;
    constexpr uint32_t kPad = 512;
    std::vector<uint8_t> lookup(kBlock + kPad);
    while (d.culfreq(2)) {
        d.update(1, 1, 2);
        SymEntry tab[256];
        uint32_t bs = 0, top_sym = 0;
        for (int b = 0; b < 256; b++) {
            uint32_t c = d.culshift(16) & 0xffffu;  // decode_short, rangecod.c:362-366
            d.update(1, c, 1u << 16);
            tab[b].lt = bs; tab[b].sy = c;
            bs += c;
            if (c) top_sym = (uint32_t)b;
        }
        if (bs > kBlock) return (size_t)-1;  // not a WaveRange stream (blocks hold at most 60000 symbols, defs.h:36)
        for (int b = 0; b < 256; b++)
            if (tab[b].sy) memset(lookup.data() + tab[b].lt, b, tab[b].sy);
        memset(lookup.data() + bs, (int)top_sym, kPad);  // see decode_symbols
        const size_t room = produced < n ? n - produced : 0;
        uint8_t* dst = sym + (produced < n ? produced : n);
        if (bs == kBlock) decode_symbols<kBlock>(d, dst, room, bs, tab, lookup.data(), top_sym);
        else if (bs) decode_symbols<0>(d, dst, room, bs, tab, lookup.data(), top_sym);
        produced += bs;
        if (d.pos > len + 8) return (size_t)-1;  // ran far past the end: corrupt stream
    }
    d.renorm();  // done_decoding, rangecod.c:371-373
    return produced;
}

}  // namespace wrrc
