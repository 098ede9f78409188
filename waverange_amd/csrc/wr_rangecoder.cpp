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
#include "wr_rangecoder_vec.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <deque>
#include <memory>
#include <new>
#include <thread>
#include <utility>
#include <vector>

namespace wrrc {

#ifdef WR_PROBE_HDR
// tools/native/hdr_probe.cpp: cycles the 16-lane sessions spend in the block headers (scalar, 257 coder steps per block and
// stream) against the cycles of whole steps
unsigned long long g_probe_hdr[4];  // encoder headers, encoder steps, decoder headers, decoder steps
struct ProbeScope {
    const int i;
    const unsigned long long t0 = __builtin_ia32_rdtsc();
    explicit ProbeScope(int k) : i(k) {}
    ~ProbeScope() { g_probe_hdr[i] += __builtin_ia32_rdtsc() - t0; }
};
#define WR_PROBE(i) ProbeScope probe_scope_##i(i)
#else
#define WR_PROBE(i) do { } while (0)
#endif

// CPU dispatch for the AVX-512 loops.  Lives here, not next to them: wr_rangecoder_avx512.cpp is built with
// -mavx512*, so the compiler may use those instructions anywhere in that file -- also in a function whose job is to
// find out whether the CPU has them.
bool vec_available()
{
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
                           __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl") &&
                           !(getenv("WR_NO_AVX512") && atoi(getenv("WR_NO_AVX512")));
    return ok;
}

namespace {

constexpr uint32_t kTop = 0x80000000u;     // rangecod.c:121  1 << (CODE_BITS-1)
constexpr uint32_t kBottom = 0x00800000u;  // rangecod.c:129  Top >> 8
constexpr int kShift = 23;                 // rangecod.c:127  CODE_BITS - 9
constexpr int kExtra = 7;                  // rangecod.c:128  (CODE_BITS-2) % 8 + 1

// cond ? a : b as a conditional move, whatever the compiler thinks of the predictability.  Used where the
// condition is "which of the two dominant symbols came" (a coin flip on a two-symbol plane); the
// "largest symbol present" tests stay branches, which predict well and keep the chain shorter
// (measured on EPYC 9575F: cmov there costs the encoder 12 %).
static inline uint32_t select_u32(uint32_t cond, uint32_t a, uint32_t b)
{
#if defined(__x86_64__)
    __asm__("testl %1, %1\n\tcmovzl %2, %0" : "+r"(a) : "r"(cond), "r"(b) : "cc");
    return a;
#else
    return cond ? a : b;
#endif
}
// keeps the compiler from splitting `if (x | y)` into two branches
static inline uint32_t opaque_u32(uint32_t v)
{
#if defined(__x86_64__)
    __asm__("" : "+r"(v));
#endif
    return v;
}

struct Enc {
    uint32_t low = 0, range = kTop;  // rangecod.c's bytecount is pos - 1 here
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
        // (range - 1 < Bottom: as range <= Bottom, except that a range of ZERO -- which only a block model that does not
        // belong to the symbols can produce, see block_failed -- ends the loop instead of feeding it for ever)
        while (range - 1u < kBottom) {
            if (low & kTop) carry();
            out[pos++] = (uint8_t)(low >> kShift);
            range <<= 8;
            low = (low << 8) & (kTop - 1);
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
        const uint32_t nbytes = (uint32_t)(pos - 1) + 5;
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

// The (up to) four most frequent symbols of a block in one pass over its table, most frequent first, the smaller symbol
// first among equals; best[e] = -1 where the block holds fewer.  Returns the number of distinct symbols.
inline uint32_t top4_symbols(const SymEntry* tab, int best[4])
{
    uint32_t cnt[4] = {0, 0, 0, 0}, distinct = 0;
    best[0] = best[1] = best[2] = best[3] = -1;
    for (int v = 0; v < 256; v++) {
        const uint32_t c = tab[v].sy;
        if (!c) continue;
        distinct++;
        if (c <= cnt[3]) continue;
        int e = 3;
        while (e > 0 && c > cnt[e - 1]) { cnt[e] = cnt[e - 1]; best[e] = best[e - 1]; e--; }
        cnt[e] = c; best[e] = v;
    }
    return distinct;
}

// symbols of one block; TOT > 0 selects the compile-time divisor
// TOPSEL: the largest symbol present is frequent in this block (the two-valued trailing plane of a field
// often is {254, 255}): "is it the largest symbol" is then a coin flip and becomes a conditional move;
// otherwise it stays a well-predicted branch, which keeps the chain shorter (+9 % on EPYC 9575F).
template <uint32_t TOT, bool TOPSEL>
inline void encode_symbols(Enc& e, const uint8_t* s, uint32_t bs, const SymEntry* tab, uint32_t top_sym)
{
    // The renormalisation test is data dependent and badly predicted, so the common case (at
    // most one byte shifted out per symbol) is written branch-free: the byte is stored
    // unconditionally and `pos` advances by 0 or 1.  Two rare cases keep a branch: a pending
    // carry at the moment a byte leaves, and a second shift (symbol probability < 1/256).
    uint32_t low = e.low, range = e.range;
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
        low = sh ? (low << 8) & (kTop - 1) : low;
        range = sh ? range << 8 : range;
        while (__builtin_expect(range - 1u < kBottom, 0)) {  // (range 0: see Enc::renorm)
            if (low & kTop) {
                size_t p = pos - 1;
                while (++out[p] == 0) p--;
            }
            out[pos++] = (uint8_t)(low >> kShift);
            range <<= 8;
            low = (low << 8) & (kTop - 1);
        }
        const uint32_t r = range / tot;
        const uint32_t t = r * tab[c].lt;
        low += t;
        // lt + sy < tot holds for every symbol except the largest one present (rangecod.c:227)
        range = TOPSEL ? select_u32(c ^ top_sym, r * tab[c].sy, range - t) : ((c != top_sym) ? r * tab[c].sy : range - t);
    }
    e.low = low; e.range = range; e.pos = pos;
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

// Upper bound on the stream length of a plane whose per-block histograms are known (the GPU counts them next to the
// quantizer): what lets the planes of a field be coded side by side straight into the caller's buffer, each at an offset
// that is known before its neighbours have ended, with gaps of a fraction of a per cent to close afterwards.
// The bound is rigorous.  Let P = 8 * pos - log2(range) for an encoder (Enc above).  A renormalisation step leaves P
// alone (one byte out, range x 256); a coding step raises it by log2(range before / range after); P starts at 8 - 31 and
// the stream ends with pos <= (P + 31) / 8 plus the 4 bytes of done_encoding (rangecod.c:254-276).  A coding step happens
// at range > 2^23 (rangecod.c:182-207), so with r = range / tot rounded down (rangecod.c:221-227)
//   symbol of count c in a block of tot:  range after >= r * c >= range * c / tot * (1 - tot / 2^23)
//   encode_shift(1, count, 16):           range after  = range >> 16 >= range / 2^16 * (1 - 2^16 / 2^23)
//   encode_freq(1, 1, 2) / (1, 0, 2):     range after >= range / 2 * (1 - 2 / 2^23)
// (the last symbol present gets the rounding remainder on top: rangecod.c:227).  Summed: the entropy of every block under
// its own histogram, 513 header bytes per block, and at most 0.0104 bit per symbol of rounding loss.
// The same sum taken over the first b blocks bounds the stream's position after block b, so a coder that is given the bound
// (PlaneJob::dst_limit) can tell after every block whether the histograms were the plane's: a position beyond it is proof
// that they were not, found at most one block late -- and a block cannot put out more than 2 bytes per symbol plus its
// header (a coding step with a count of 1 or more costs at most log2(60000) + 0.01 bits), which is kFailedBlockSlack.
size_t encode_bound_hist(const uint16_t* hists, size_t n)
{
    static const std::vector<double> lg = []() {
        std::vector<double> t(kBlock + 1, 0.0);
        for (uint32_t c = 1; c <= kBlock; c++) t[c] = log2((double)c);
        return t;
    }();
    const double loss16 = -log2(1.0 - 65536.0 / 8388608.0), loss2 = -log2(1.0 - 2.0 / 8388608.0);
    const size_t nblocks = n / kBlock + 1;  // (a plane of a multiple of 60000 symbols ends with an empty block: wrappers.cpp:85-128)
    double bits = 1.0 + loss2;             // "no more blocks"
    for (size_t b = 0; b < nblocks; b++) {
        const size_t left = n - b * (size_t)kBlock;
        const uint32_t bs = left < kBlock ? (uint32_t)left : kBlock;
        bits += 1.0 + loss2 + 256.0 * (16.0 + loss16);
        if (!bs) continue;
        const uint16_t* h = hists + b * 256;
        const double loss = -log2(1.0 - (double)bs / 8388608.0);
        double e = 0;
        uint32_t seen = 0;
        for (int v = 0; v < 256; v++) { const uint32_t c = h[v]; e += (double)c * lg[c]; seen += c; }
        if (seen != bs) return encode_bound(n);  // not this plane's histograms: the plain bound holds whatever the symbols are
        bits += (double)bs * (lg[bs] + loss) - e;
    }
    // (+ 5: see above; + 64: the loops store a byte, the 16-lane loop four, ahead of the stream's end; 1e-9: the table's rounding)
    const double bytes = bits * (1.0 + 1e-9) / 8.0 + 5.0 + 64.0;
    const size_t est = (size_t)bytes + 1;
    const size_t plain = encode_bound(n);
    return est < plain ? est : plain;
}

namespace {

// Interleaved symbol loop for NS planes (full 60000-symbol blocks).  One plane's coder is a
// serial dependency chain of ~11 cycles per symbol that leaves most of a core idle; NS
// independent chains in one loop fill it.  Same statements per plane as encode_symbols.
template <int NS, bool TOPSEL>
inline void encode_symbols_multi(Enc* const* es, const uint8_t* const* ss, const SymEntry (*tabs)[256], const uint32_t* tops)
{
    uint32_t low[NS], range[NS];
    uint8_t* out[NS];
    size_t pos[NS];
    // local copies of the per-plane pointers and constants: the byte stores below may alias anything reached through
    // a pointer (they are char stores), and would otherwise force ss[k] and tops[k] to be reloaded for every symbol
    const uint8_t* sym[NS];
    uint32_t top[NS];
    for (int k = 0; k < NS; k++) {
        low[k] = es[k]->low; range[k] = es[k]->range; out[k] = es[k]->out; pos[k] = es[k]->pos;
        sym[k] = ss[k]; top[k] = tops[k];
    }
    for (uint32_t i = 0; i < kBlock; i++) {
#pragma GCC unroll 8
        for (int k = 0; k < NS; k++) {
            const uint32_t c = sym[k][i];
            const uint32_t sh = range[k] <= kBottom;
            if (__builtin_expect(sh & (low[k] >> 31), 0)) {
                size_t p = pos[k] - 1;
                while (++out[k][p] == 0) p--;
            }
            out[k][pos[k]] = (uint8_t)(low[k] >> kShift);
            pos[k] += sh;
            // conditional moves by hand: left to itself the compiler turns these selects into branches in
            // this loop, and on a plane of ~1 bit per symbol "a byte leaves now" is badly predicted
            low[k] = select_u32(sh, (low[k] << 8) & (kTop - 1), low[k]);
            range[k] = select_u32(sh, range[k] << 8, range[k]);
            while (__builtin_expect(range[k] - 1u < kBottom, 0)) {  // (range 0: see Enc::renorm)
                if (low[k] & kTop) {
                    size_t p = pos[k] - 1;
                    while (++out[k][p] == 0) p--;
                }
                out[k][pos[k]++] = (uint8_t)(low[k] >> kShift);
                range[k] <<= 8;
                low[k] = (low[k] << 8) & (kTop - 1);
            }
            const uint32_t r = range[k] / kBlock;
            const uint32_t t = r * tabs[k][c].lt;
            low[k] += t;
            range[k] = TOPSEL ? select_u32(c ^ top[k], r * tabs[k][c].sy, range[k] - t)
                              : ((c != top[k]) ? r * tabs[k][c].sy : range[k] - t);
        }
    }
    for (int k = 0; k < NS; k++) { es[k]->low = low[k]; es[k]->range = range[k]; es[k]->pos = pos[k]; }
}

// block header of wrappers.cpp:85-113: "a block follows", then the 256 counts.
// false: the histogram that came with the plane (the GPU counts it next to the quantizer) does not add up to the block --
// it is not this block's, the stream is given up (nothing has been written for the block).  A histogram that adds up and is
// still not the symbols' shows while the block is coded: a symbol with a count of zero leaves the coder with a range of
// zero (Enc::renorm ends on it, the 16-lane loop retires the lane), which the callers test for after every block
// (block_failed).  Either way the stream ends at once and its job reports (size_t)-1; at most one byte per symbol of the
// block has been written by then (kFailedBlockSlack: what a buffer sized by encode_bound_hist needs on top).
inline bool encode_block_header(Enc& e, const uint8_t* s, uint32_t bs, const uint16_t* hist, SymEntry* tab, uint32_t* top_sym)
{
    uint32_t h[256];
    if (hist) {
        uint32_t sum = 0;
        for (int b = 0; b < 256; b++) { h[b] = hist[b]; sum += h[b]; }
        if (sum != bs) return false;
    } else
        histogram(s, bs, h);
    e.freq(1, 1, 2);
    uint32_t cum = 0, top = 0;
    for (int b = 0; b < 256; b++) {
        e.shift(1, h[b], 16);  // encode_short(count), rangecod.h:155
        tab[b].lt = cum; tab[b].sy = h[b];
        cum += h[b];
        if (h[b]) top = (uint32_t)b;
    }
    *top_sym = top;
    return true;
}
// after a block: the coder's state says that the block model was not the symbols' (see encode_block_header)
inline bool block_failed(const Enc& e) { return e.range == 0; }

}  // namespace

// The encoder's loop keeps five values per plane in registers; beyond three planes the spills cost more
// than the interleaving gains (EPYC 9575F: 1 plane 380-500 Msym/s, 2: 700, 3: 680, 4: 560 in total).
constexpr int kMaxEncStreams = 3;

namespace {
// Where the symbols of a plane are: the whole plane in host memory, or the window of it that a PlaneWindow handed out.
struct SymCursor {
    uint8_t* base = nullptr;
    size_t first = 0, count = 0;  // extent of what `base` points at
    const PlaneWindow* io = nullptr;
    bool refused = false;         // the plane's owner turned a window request down (PlaneWindow): the stream is to end at once
    void set(const uint8_t* whole, size_t n, const PlaneWindow* w)
    {
        io = w; first = 0; refused = false;
        if (w) { base = nullptr; count = 0; } else { base = const_cast<uint8_t*>(whole); count = n; }
    }
    // symbol `pos` of a plane of n symbols; moves the window on when pos has run out of it.  nullptr (and `refused`): the
    // request was turned down -- the caller gives the stream up without touching a symbol
    uint8_t* at(size_t pos, size_t n)
    {
        if (refused) return nullptr;
        if (io && pos >= first + count && pos < n) {
            size_t c = n - pos;
            uint8_t* const w = io->window(io->user, pos, &c);
            if (!w || !c) { refused = true; base = nullptr; first = pos; count = 0; return nullptr; }
            base = w; first = pos; count = c;
        }
        return base + (pos - first);
    }
    size_t room(size_t pos) const { return pos < first + count ? first + count - pos : 0; }  // symbols from pos on that exist here
    void end(size_t pos) { if (io && !refused) { size_t z = 0; (void)io->window(io->user, pos, &z); } }  // (a refused stream has nothing to end)
};
}  // namespace

namespace {

// A set of up to kMaxEncStreams plane streams that advance block by block in lockstep on one thread
// (block loop of wrappers.cpp:85-128: a full final block is followed by an empty one).  Streams may join at
// any block boundary and leave when they end; slots stay compact so that the interleaved loop always runs on
// slots 0 .. count-1.
class EncGroup {
public:
    struct Stream {
        SymCursor sym; size_t n, done, blk; const uint16_t* hist;
        void* tag;
        bool failed = false;  // a refused window or a block model that is not the symbols': ends with (size_t)-1
        size_t limit = 0;     // what a stream coded with this plane's own histograms never exceeds (0: not known, not checked)
    };
    int count() const { return count_; }
    bool full() const { return count_ == kMaxEncStreams; }
    void add(const uint8_t* sym, size_t n, uint8_t* out, const uint16_t* hist, void* tag, const PlaneWindow* io = nullptr, size_t limit = 0)
    {
        const int k = count_++;
        es_[k] = new (store_[k]) Enc(out);
        st_[k] = Stream{SymCursor(), n, 0, 0, hist, tag, false, hist ? limit : 0};
        st_[k].sym.set(sym, n, io);
    }
    // one block of every stream; on_end(tag, stream length or (size_t)-1) for the streams that ended with it
    template <class OnEnd>
    void step(OnEnd on_end) { step_streams(count_, es_, store_, st_, tabs_, tops_, on_end); }
    // the same on the slots of another holder of at most kMaxEncStreams streams (a 16-lane session that is down to a few)
    template <class OnEnd>
    static void step_streams(int& count, Enc** es, unsigned char (*store)[sizeof(Enc)], Stream* st, SymEntry (*tabs)[256], uint32_t* tops,
                             OnEnd on_end)
    {
        uint32_t bs[kMaxEncStreams];
        const uint8_t* ss[kMaxEncStreams];
        bool all_full = true, topsel = false;  // topsel: some plane's largest symbol holds > 2 % of this block
        for (int k = 0; k < count; k++) {
            Stream& s = st[k];
            const size_t left = s.n - s.done;
            bs[k] = left < kBlock ? (uint32_t)left : kBlock;
            ss[k] = s.sym.at(s.done, s.n);
            if ((!ss[k] && bs[k]) || !encode_block_header(*es[k], ss[k], bs[k], s.hist ? s.hist + s.blk * 256 : nullptr, tabs[k], &tops[k])) {
                s.failed = true; bs[k] = 0; all_full = false;
                continue;
            }
            all_full = all_full && bs[k] == kBlock;
            topsel = topsel || (uint64_t)tabs[k][tops[k]].sy * 50 > bs[k];
        }
        if (all_full) {
            switch (count) {
            case 1: topsel ? encode_symbols<kBlock, true>(*es[0], ss[0], kBlock, tabs[0], tops[0]) : encode_symbols<kBlock, false>(*es[0], ss[0], kBlock, tabs[0], tops[0]); break;
            case 2: topsel ? encode_symbols_multi<2, true>(es, ss, tabs, tops) : encode_symbols_multi<2, false>(es, ss, tabs, tops); break;
            default: topsel ? encode_symbols_multi<3, true>(es, ss, tabs, tops) : encode_symbols_multi<3, false>(es, ss, tabs, tops); break;
            }
        } else {
            for (int k = 0; k < count; k++) {
                if (st[k].failed) continue;
                if (bs[k] == kBlock) encode_symbols<kBlock, true>(*es[k], ss[k], kBlock, tabs[k], tops[k]);
                else if (bs[k]) encode_symbols<0, true>(*es[k], ss[k], bs[k], tabs[k], tops[k]);
            }
        }
        for (int k = 0; k < count;) {
            const bool failed = st[k].failed || block_failed(*es[k]) || (st[k].limit && es[k]->pos > st[k].limit);
            st[k].done += bs[k];
            st[k].blk++;
            if (!failed && bs[k] == kBlock) { k++; continue; }
            if (failed) on_end(st[k].tag, (size_t)-1);
            else {
                es[k]->freq(1, 0, 2);  // "no more blocks"
                on_end(st[k].tag, es[k]->finish());
            }
            // the last slot moves into the hole (its block size with it: it has not been looked at yet)
            const int last = --count;
            if (k != last) {
                es[k] = new (store[k]) Enc(*es[last]);
                st[k] = st[last];
                bs[k] = bs[last];
            }
        }
    }

private:
    int count_ = 0;
    Enc* es_[kMaxEncStreams];
    alignas(Enc) unsigned char store_[kMaxEncStreams][sizeof(Enc)];
    Stream st_[kMaxEncStreams];
    SymEntry tabs_[kMaxEncStreams][256];
    uint32_t tops_[kMaxEncStreams];
};

// A 16-lane session that holds no more streams than a scalar loop takes runs them through that loop: per stream and per
// thread it is the faster one there (EPYC 9575F, dominant-symbol planes, Msym/s per stream: encoder 1 / 2 / 3 streams scalar
// ~300 / ~250 / 190 against ~110 in the vector loop at <= 4 lanes; decoder 304 / 214 / 142 against ~110).  Sessions get
// that small when a run fills or drains, with a lone caller, and after idle workers have taken over half of a session's
// streams.  (Two 16-lane groups per session with interleaved steps, 2.3 Gsym/s per thread at 72 Msym/s per stream, lost in
// the pipeline and is gone: profiles/r03/NOTES.md.)

}  // namespace

namespace {

// An encoder stream between two blocks, as it moves from one worker's session to another's (Pool: an idle worker takes
// over half of the streams of the fullest session)
struct EncStream {
    Enc e;
    EncGroup::Stream st;
};

// Up to 16 encoder streams advancing block by block in lockstep on one thread: the full blocks that at most four
// symbols hold (>= 99 %) are coded 16 lanes at a time by the AVX-512 loop (wr_rangecoder_vec.h), block headers,
// other blocks and the final partial block by the scalar code of their stream.
class VecEncGroup {
public:
    static constexpr int kCap = kVecLanes;
    VecEncGroup() { memset(tabs_, 0, sizeof tabs_); }
    int count() const { return count_; }
    bool full() const { return count_ == kCap; }
    void add(const uint8_t* sym, size_t n, uint8_t* out, const uint16_t* hist, void* tag, const PlaneWindow* io = nullptr, size_t limit = 0)
    {
        const int k = count_++;
        es_[k] = new (store_[k]) Enc(out);
        st_[k] = EncGroup::Stream{SymCursor(), n, 0, 0, hist, tag, false, hist ? limit : 0};
        st_[k].sym.set(sym, n, io);
    }
    // hand the last stream over (call between steps) / adopt one
    EncStream give()
    {
        const int k = --count_;
        return EncStream{*es_[k], st_[k]};
    }
    void take(const EncStream& m)
    {
        const int k = count_++;
        es_[k] = new (store_[k]) Enc(m.e);
        st_[k] = m.st;
    }
    void* tag_of_last() const { return st_[count_ - 1].tag; }
    void set_tag_of_last(void* t) { st_[count_ - 1].tag = t; }
    template <class OnEnd>
    void step(OnEnd on_end)
    {
        if (count_ <= kMaxEncStreams) { EncGroup::step_streams(count_, es_, store_, st_, tabs_, tops_, on_end); return; }
        WR_PROBE(1);
        uint32_t bs[kCap];
        VecEncBlock vb;
        vb.active = 0;
        vb.failed = 0;
        vb.gather = 0;
        vb.tab = &tabs_[0][0].lt;
        vb.packed = &packed_[0][0];
        const uint8_t* ss[kCap];
        for (int k = 0; k < count_; k++) {
            EncGroup::Stream& s = st_[k];
            const size_t left = s.n - s.done;
            bs[k] = left < kBlock ? (uint32_t)left : kBlock;
            ss[k] = s.sym.at(s.done, s.n);
            bool header_ok;
            { WR_PROBE(0); header_ok = !(!ss[k] && bs[k]) && encode_block_header(*es_[k], ss[k], bs[k], s.hist ? s.hist + s.blk * 256 : nullptr, tabs_[k], &tops_[k]); }
            if (!header_ok) {
                s.failed = true; bs[k] = 0;
                continue;
            }
            if (bs[k] != kBlock) {  // the final, partial block of a stream: scalar
                if (bs[k]) encode_symbols<0, true>(*es_[k], ss[k], bs[k], tabs_[k], tops_[k]);
                continue;
            }
            // the four most frequent symbols of the block; if they hold >= 99 % of every full block of this step the
            // lanes pick {lt, sy} by comparing with them, otherwise all lanes gather them from their tables
            static_assert(kVecCand == 4, "top4_symbols");
            uint32_t cand[kVecCand], covered = 0;
            int best[4];
            top4_symbols(tabs_[k], best);
            for (int e = 0; e < kVecCand; e++) {
                cand[e] = best[e] < 0 ? 0x100u : (uint32_t)best[e];
                if (best[e] >= 0) covered += tabs_[k][best[e]].sy;
            }
            if ((uint64_t)covered * 100 < (uint64_t)kBlock * 99) vb.gather = 1;
            vb.active |= 1u << k;
            vb.low[k] = es_[k]->low; vb.range[k] = es_[k]->range;
            vb.sym[k] = ss[k]; vb.out[k] = es_[k]->out; vb.pos[k] = es_[k]->pos; vb.top[k] = tops_[k];
            for (int e = 0; e < kVecCand; e++) {
                vb.cand[e][k] = cand[e];
                vb.lt[e][k] = cand[e] < 256 ? tabs_[k][cand[e]].lt : 0;
                vb.sy[e][k] = cand[e] < 256 ? tabs_[k][cand[e]].sy : 0;
            }
        }
        if (vb.active) {
            for (int k = 0; k < kCap; k++)
                if (!(vb.active >> k & 1)) {
                    vb.low[k] = 0; vb.range[k] = 0; vb.sym[k] = nullptr; vb.out[k] = nullptr; vb.pos[k] = 0; vb.top[k] = 0xffffffffu;
                    for (int e = 0; e < kVecCand; e++) { vb.cand[e][k] = 0x100; vb.lt[e][k] = 0; vb.sy[e][k] = 0; }
                }
            if (vb.gather)
                for (int k = 0; k < kCap; k++)
                    if (vb.active >> k & 1)
                        for (int b = 0; b < 256; b++) packed_[k][b] = tabs_[k][b].lt | (((uint32_t)b == tops_[k] ? kVecTopMark : tabs_[k][b].sy) << 16);
            vec_encode_block(&vb);
            for (int k = 0; k < count_; k++) {
                if (!(vb.active >> k & 1)) continue;
                if (vb.failed >> k & 1) { st_[k].failed = true; continue; }  // (the lane was retired inside the block)
                es_[k]->low = vb.low[k]; es_[k]->range = vb.range[k]; es_[k]->pos = vb.pos[k];
            }
        }
        for (int k = 0; k < count_;) {
            const bool failed = st_[k].failed || block_failed(*es_[k]) || (st_[k].limit && es_[k]->pos > st_[k].limit);
            st_[k].done += bs[k];
            st_[k].blk++;
            if (!failed && bs[k] == kBlock) { k++; continue; }
            if (failed) on_end(st_[k].tag, (size_t)-1);
            else {
                es_[k]->freq(1, 0, 2);  // "no more blocks"
                on_end(st_[k].tag, es_[k]->finish());
            }
            const int last = --count_;
            if (k != last) {
                es_[k] = new (store_[k]) Enc(*es_[last]);
                st_[k] = st_[last];
                bs[k] = bs[last];
            }
        }
    }

private:
    int count_ = 0;
    Enc* es_[kCap];
    alignas(Enc) unsigned char store_[kCap][sizeof(Enc)];
    EncGroup::Stream st_[kCap];
    SymEntry tabs_[kCap][256];  // contiguous: the vector loop gathers {lt, sy} at (lane * 256 + symbol)
    alignas(64) uint32_t packed_[kCap][256];  // the same as lt | sy << 16 (sy = kVecTopMark: the largest symbol present) for the per-lane look-ups
    uint32_t tops_[kCap];
};

}  // namespace

// `count` planes (any lengths, any statistics) on the calling thread through the 16-lane encoder loop: test and
// measurement hook; the coder pool is the product path.  False if the CPU lacks AVX-512.
bool encode_planes_vec(int count, const uint8_t* const* sym, const size_t* n, uint8_t* const* out, size_t* lens,
                       const PlaneWindow* const* io)
{
    if (!vec_available()) return false;
    std::unique_ptr<VecEncGroup> g(new VecEncGroup);
    int next = 0;
    while (next < count || g->count()) {
        while (next < count && !g->full()) {
            g->add(sym[next], n[next], out[next], nullptr, lens + next, io ? io[next] : nullptr);
            next++;
        }
        g->step([](void* tag, size_t len) { *static_cast<size_t*>(tag) = len; });
    }
    return true;
}

void encode_planes(int count, const uint8_t* const* sym, size_t n, uint8_t* const* out, const uint16_t* const* hists, size_t* lens,
                   const PlaneWindow* const* io, const size_t* limits)
{
    // up to kMaxEncStreams planes at a time in one symbol loop; further planes join as earlier ones end
    EncGroup g;
    int next = 0;
    while (next < count || g.count()) {
        while (next < count && !g.full()) {
            g.add(sym[next], n, out[next], (hists && hists[next]) ? hists[next] : nullptr, lens + next, io ? io[next] : nullptr, limits ? limits[next] : 0);
            next++;
        }
        g.step([](void* tag, size_t len) { *static_cast<size_t*>(tag) = len; });
    }
}

size_t encode_plane(const uint8_t* sym, size_t n, uint8_t* out, const uint16_t* hists)
{
    size_t len = 0;
    encode_planes(1, &sym, n, &out, &hists, &len);
    return len;
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

namespace {

constexpr uint32_t kPad = 512;                          // see decode_symbols
constexpr uint32_t kBucketShift = 4;                    // 16 cumulative-frequency values per bucket
constexpr uint32_t kBuckets = ((kBlock + kPad) >> kBucketShift) + 1;

// Everything the symbol loop needs about one block of one plane.
struct BlockModel {
    SymEntry tab[256];
    uint32_t top;       // largest symbol present: its interval is open-ended (rangecod.c:317,345-348)
    uint32_t bs;
    // the two most probable symbols, when together they hold >= 90 % of the block: they are recognised
    // by comparing `low` with the ends of their intervals, without the division and the look-ups
    // (mps_on == false: test disabled; sy[1] == 0: one symbol only)
    bool mps_on;
    uint32_t mps[2], mps_lt[2], mps_sy[2];
    bool mps_is_top[2];
    // blocks of three or four distinct symbols that the test above does not cover (mps_on is set as well,
    // it selects the loop variant): the symbol is the number of interval
    // starts help * lt at or below `low`; absent entries carry an lt no cumulative frequency reaches
    uint32_t few;                              // number of distinct symbols if <= 4, else 0
    uint32_t few_sym[4], few_lt[4], few_sy[4];  // ascending symbols
    // bucket table: symbol of all 16 values [16 j, 16 j + 15] if they agree, else kMixed.  15 KB for four
    // planes stays in L1, the four 60 KB lookup tables do not.  Used when < 2 % of the block
    // falls into mixed buckets (each escape is a branch miss).
    // the four most probable symbols for the 16-lane vector loop (unused entries: sy 0), and whether they hold
    // >= WR_RC_MPS_PCT of the block
    uint32_t cand[4], cand_lt[4], cand_sy[4];
    bool cand_ok;
    bool use_buckets;
    bool tables_ready;  // bucket / lookup are filled in (the vector loop builds them only when a rare symbol turns up)
    uint16_t bucket[kBuckets];
    uint8_t lookup[kBlock + kPad];
};
constexpr uint16_t kMixed = 0x100;
#ifndef WR_RC_MPS_PCT
#define WR_RC_MPS_PCT 90
#endif
constexpr uint32_t kMixedPct = 2;  // (1 .. 8 measured on EPYC 9575F: profiles/r04/t_rc_noise_bucket_threshold_epyc9575f.txt)

// the look-up side of a block model (60 KB + buckets): what the division path needs
void finish_model_tables(BlockModel& m)
{
    uint32_t present = 0, rare = 0;  // rare: symbols of fewer than a bucket's 16 values (they may share a bucket with a neighbour's boundary)
    for (int b = 0; b < 256; b++)
        if (m.tab[b].sy) { memset(m.lookup + m.tab[b].lt, b, m.tab[b].sy); present++; rare += m.tab[b].sy < (1u << kBucketShift); }
    memset(m.lookup + m.bs, (int)m.top, kPad);
    // Two boundaries between symbols of 16 values or more never share a bucket, and a boundary makes its bucket mixed unless
    // it falls on a multiple of 16: with that many mixed buckets all but certain (noise planes: ~250 symbols of ~235 values
    // each, 6.8 % of the block; the test asks for 5/4 of the threshold) the table below would be built (3782 entries, ~2 % of
    // the block's decoding time) only to be switched off.  (Which look-up a block takes never changes what it decodes.)
    if (present > 2 * rare + 1 && (uint64_t)(present - 2 * rare - 1) * (80u << kBucketShift) >= (uint64_t)m.bs * kMixedPct) {
        m.use_buckets = false;
        m.tables_ready = true;
        return;
    }
    uint32_t mixed = 0;
    for (uint32_t j = 0; j < kBuckets; j++) {
        const uint32_t lo = j << kBucketShift, hi = lo + (1u << kBucketShift) - 1;
        const uint8_t a = m.lookup[lo < kBlock + kPad ? lo : kBlock + kPad - 1], z = m.lookup[hi < kBlock + kPad ? hi : kBlock + kPad - 1];
        m.bucket[j] = (a == z) ? a : kMixed;  // symbols ascend with the cumulative frequency: equal ends = equal throughout
        if (a != z && lo < m.bs) mixed += 1u << kBucketShift;
    }
    m.use_buckets = (uint64_t)mixed * 100 < (uint64_t)m.bs * kMixedPct;
    m.tables_ready = true;
}

// the cheap side: dominant symbols, few-symbol form
void finish_model_stats(BlockModel& m)
{
    m.tables_ready = false;
    int best[4];
    const uint32_t distinct = top4_symbols(m.tab, best);
    const uint32_t b1 = best[0] < 0 ? 0u : (uint32_t)best[0], b2 = best[1] < 0 ? 256u : (uint32_t)best[1];
    const uint32_t sy2 = b2 < 256 ? m.tab[b2].sy : 0;
    m.mps[0] = b1; m.mps_lt[0] = m.tab[b1].lt; m.mps_sy[0] = m.tab[b1].sy; m.mps_is_top[0] = b1 == m.top;
    m.mps[1] = b2 & 255; m.mps_lt[1] = sy2 ? m.tab[b2].lt : 0; m.mps_sy[1] = sy2; m.mps_is_top[1] = sy2 && b2 == m.top;
    m.mps_on = m.bs && ((uint64_t)(m.tab[b1].sy + sy2) * 100 >= (uint64_t)m.bs * WR_RC_MPS_PCT);
    // (only where the two-symbol test above does not already cover the block: that one is shorter)
    m.few = (!m.mps_on && distinct >= 3 && distinct <= 4) ? distinct : 0;
    if (distinct <= 4) {  // the symbols in ascending order (the four most frequent ones are all there are)
        int asc[4] = {best[0], best[1], best[2], best[3]};
        for (int i = 1; i < 4; i++)
            for (int j = i; j > 0 && asc[j] >= 0 && (asc[j - 1] < 0 || asc[j] < asc[j - 1]); j--) { const int t = asc[j]; asc[j] = asc[j - 1]; asc[j - 1] = t; }
        for (uint32_t j = 0; j < distinct; j++) { m.few_sym[j] = (uint32_t)asc[j]; m.few_lt[j] = m.tab[asc[j]].lt; m.few_sy[j] = m.tab[asc[j]].sy; }
    }
    for (uint32_t j = distinct < 4 ? distinct : 4; j < 4; j++) { m.few_sym[j] = m.top; m.few_lt[j] = 0xffff; m.few_sy[j] = 0; }  // help * 0xffff > low, always
    if (m.few) m.mps_on = true;
    // candidates of the vector loop
    uint32_t covered = 0;
    for (int e = 0; e < 4; e++) {
        if (best[e] < 0) { m.cand[e] = 0; m.cand_lt[e] = 0; m.cand_sy[e] = 0; continue; }
        m.cand[e] = (uint32_t)best[e]; m.cand_lt[e] = m.tab[best[e]].lt; m.cand_sy[e] = m.tab[best[e]].sy;
        covered += m.tab[best[e]].sy;
    }
    m.cand_ok = m.bs && (uint64_t)covered * 100 >= (uint64_t)m.bs * WR_RC_MPS_PCT;
}

void finish_model(BlockModel& m)
{
    finish_model_stats(m);
    finish_model_tables(m);
}

// Symbol loop of NS planes, interleaved; needs, per plane, a full block, room for 60000 symbols
// and at least 3 * 60000 + 8 unread stream bytes (a symbol pulls in at most 3 bytes), so the
// loop carries no bounds checks.  One plane's decoder is a serial chain of ~45 cycles per symbol
// (renormalise, range / tot, low / help, table look-ups, multiply) that leaves most of a core
// idle; NS independent chains in one loop fill it.  Arithmetic per plane as in decode_symbols.
// (The stream feeds `low` a continuous bit string that starts 7 bits into a byte:
// held << EXTRA | next >> (8 - EXTRA), rangecod.c:297-299.)
#ifndef WR_RC_SPEC_DIV
#define WR_RC_SPEC_DIV 1
#endif
constexpr bool kSpecDiv = WR_RC_SPEC_DIV != 0;

template <int NS, unsigned MPS>  // bit k of MPS: plane k's block has dominant symbols (BlockModel::mps_on)
void decode_symbols_multi(Dec* const* ds, uint8_t* const* dst, const BlockModel* const* ms)
{
    // Only low, range and the stream pointer live in registers per plane; everything else is read
    // through the model pointer where it is needed (hoisting it costs more in spills than it saves).
    uint32_t low[NS], range[NS];
    const uint8_t* p[NS];
    // local copies of the per-plane pointers: the symbol stores below are char stores, which may alias anything
    // reached through a pointer, and would otherwise force ms[k] and dst[k] to be reloaded for every symbol
    const BlockModel* mod[NS];
    uint8_t* sym[NS];
    for (int k = 0; k < NS; k++) {
        low[k] = ds[k]->low; range[k] = ds[k]->range; p[k] = ds[k]->in + ds[k]->pos;
        mod[k] = ms[k]; sym[k] = dst[k];
    }
    for (uint32_t i = 0; i < kBlock; i++) {
#pragma GCC unroll 8
        for (int k = 0; k < NS; k++) {
            const BlockModel* const m = mod[k];
            uint32_t lw = low[k], rg = range[k];
            const uint8_t* q = p[k];
            uint32_t help;
            {   // first renormalisation step without a branch
                // (shifts by 0 or 8 rather than selects: compilers turn selects into branches here, and
                // this one is badly predicted on planes of medium entropy)
                const uint32_t sh = rg <= kBottom;
                // range / 60000 of either outcome beside the compare (kSpecDiv): compare -> shift -> divide in a row is
                // a tenth of the symbol's chain
                const uint32_t h_as_is = rg / kBlock, h_shifted = kSpecDiv ? (rg << 8) / kBlock : 0;
                const uint32_t bits = ((((uint32_t)q[-1] << 8) | q[0]) >> (8 - kExtra)) & 0xff;
                lw = (lw << (8 * sh)) | (bits & (0u - sh));
                rg <<= 8 * sh;
                q += sh;
                help = kSpecDiv ? select_u32(sh, h_shifted, h_as_is) : rg / kBlock;
            }
            if (__builtin_expect(rg <= kBottom, 0)) {
                do {
                    lw = (lw << 8) | (((((uint32_t)q[-1] << 8) | q[0]) >> (8 - kExtra)) & 0xff);
                    rg <<= 8;
                    q++;
                } while (rg <= kBottom);
                help = rg / kBlock;
            }
            uint32_t c;
            bool hit = false;
            if ((MPS >> k & 1) && m->few) {
                // <= 4 symbols in the block: index = how many interval starts lie at or below low
                const uint32_t t1 = help * m->few_lt[1], t2 = help * m->few_lt[2], t3 = help * m->few_lt[3];
                const uint32_t idx = (uint32_t)(lw >= t1) + (uint32_t)(lw >= t2) + (uint32_t)(lw >= t3);
                const uint32_t t = select_u32(idx, select_u32(idx ^ 1, select_u32(idx ^ 2, t3, t2), t1), 0);
                hit = true;
                c = m->few_sym[idx];
                lw -= t;
                rg = select_u32(idx ^ (m->few - 1), help * m->few_sy[idx], rg - t);  // rangecod.c:345-348
            } else if (MPS >> k & 1) {
                // cf = low / help lies in a symbol's interval [lt, lt + sy)  <=>  low - help * lt < help * sy;
                // the new range is that width, or what is left of range for the largest symbol (rangecod.c:345-348)
                const uint32_t a0 = help * m->mps_lt[0], a1 = help * m->mps_lt[1];
                const uint32_t w0 = m->mps_is_top[0] ? rg - a0 : help * m->mps_sy[0];
                const uint32_t w1 = m->mps_is_top[1] ? rg - a1 : help * m->mps_sy[1];
                const uint32_t in0 = lw - a0 < w0, in1 = lw - a1 < w1;
                if (__builtin_expect(opaque_u32(in0 | in1), 1)) {
                    hit = true;
                    c = select_u32(in0, m->mps[0], m->mps[1]);
                    lw -= select_u32(in0, a0, a1);
                    rg = select_u32(in0, w0, w1);
                }
            }
            if (!hit) {
                const uint32_t cf = lw / help;
                if (m->use_buckets) {
                    const uint32_t e = m->bucket[cf >> kBucketShift];
                    c = e;
                    if (__builtin_expect(e == kMixed, 0)) c = m->lookup[cf];
                } else
                    c = m->lookup[cf];
                const uint32_t t = help * m->tab[c].lt;
                lw -= t;
                rg = (c != m->top) ? help * m->tab[c].sy : rg - t;
            }
            sym[k][i] = (uint8_t)c;
            low[k] = lw; range[k] = rg; p[k] = q;
        }
    }
    for (int k = 0; k < NS; k++) { ds[k]->low = low[k]; ds[k]->range = range[k]; ds[k]->pos = (size_t)(p[k] - ds[k]->in); ds[k]->held = p[k][-1]; }
}

// The same loop for blocks without dominant symbols and without a usable bucket table -- noise planes, a third of the
// pool's worker-seconds -- on a different representation of the decoder's state.  rangecod.c feeds `low` from a bit string
// that starts 7 bits into a byte (rangecod.c:297-299: low = low << 8 | (buffer << 7 & 0xff); buffer = next byte; low |=
// buffer >> 1), so every renormalisation step looks at two stream bytes and shifts.  Carry the one bit of `buffer` that
// `low` has not taken yet along with it: L2 = low << 1 | (buffer & 1).  Then the step is byte-aligned,
//     L2' = (low' << 1) | (next & 1) = (low << 9) | ((buffer & 1) << 8) | next = (L2 << 8) | next
// -- one byte load, one shift, one or; low = L2 >> 1 where the arithmetic wants it, and low -= t becomes L2 -= 2 t.
// (L2 < 2 * range <= 2^32, and < 2^25 whenever a step shifts it.)  15 instructions less per symbol than the loop above.
template <int NS>
void decode_symbols_noise(Dec* const* ds, uint8_t* const* dst, const BlockModel* const* ms)
{
    uint32_t low2[NS], range[NS];
    const uint8_t* p[NS];
    const BlockModel* mod[NS];
    uint8_t* sym[NS];
    for (int k = 0; k < NS; k++) {
        low2[k] = (ds[k]->low << 1) | (ds[k]->held & 1u); range[k] = ds[k]->range; p[k] = ds[k]->in + ds[k]->pos;
        mod[k] = ms[k]; sym[k] = dst[k];
    }
    for (uint32_t i = 0; i < kBlock; i++) {
#pragma GCC unroll 8
        for (int k = 0; k < NS; k++) {
            const BlockModel* const m = mod[k];
            uint32_t l2 = low2[k], rg = range[k];
            const uint8_t* q = p[k];
            // range / 60000 of either outcome beside the compare (see above)
            uint32_t help = rg / kBlock;
            const uint32_t h_shifted = (rg << 8) / kBlock, rg_shifted = rg << 8;
            const uint32_t fed = (l2 << 8) | q[0];
#if defined(__x86_64__)
            // one compare decides all four: the state, the range, its quotient and whether the stream pointer moves on
            __asm__("cmpl $0x800001, %[rg]\n\t"
                    "cmovbl %[fed], %[l2]\n\t"
                    "cmovbl %[hs], %[help]\n\t"
                    "cmovbl %[rgs], %[rg]\n\t"
                    "adcq $0, %[q]"
                    : [l2] "+r"(l2), [help] "+r"(help), [rg] "+r"(rg), [q] "+r"(q)
                    : [fed] "r"(fed), [hs] "r"(h_shifted), [rgs] "r"(rg_shifted)
                    : "cc");
#else
            const uint32_t sh = rg <= kBottom;
            l2 = sh ? fed : l2; help = sh ? h_shifted : help; rg = sh ? rg_shifted : rg; q += sh;
#endif
            if (__builtin_expect(rg <= kBottom, 0)) {
                do {
                    l2 = (l2 << 8) | q[0];
                    rg <<= 8;
                    q++;
                } while (rg <= kBottom);
                help = rg / kBlock;
            }
            const uint32_t cf = (l2 >> 1) / help;
            const uint32_t c = m->lookup[cf];
            const uint32_t t = help * m->tab[c].lt;
            l2 -= t + t;
            rg = (c != m->top) ? help * m->tab[c].sy : rg - t;
            sym[k][i] = (uint8_t)c;
            low2[k] = l2; range[k] = rg; p[k] = q;
        }
    }
    for (int k = 0; k < NS; k++) { ds[k]->low = low2[k] >> 1; ds[k]->range = range[k]; ds[k]->pos = (size_t)(p[k] - ds[k]->in); ds[k]->held = p[k][-1]; }
}
#if defined(__x86_64__) && defined(__GNUC__)
// The four-stream form of the loop above with the registers allotted by hand.  Four streams carry 4 x {state, range,
// stream pointer} + the loop counter = 13 live values besides the temporaries, two of which the division fixes (eax,
// edx): the compiler keeps the states in registers and passes all four RANGES through the stack in every iteration
// (a store and a forwarded load on each stream's dependency chain) -- 45 instructions per symbol.  Here states and
// ranges own eight registers, the stream pointers live in memory (they are only read, and bumped by `adc $0, mem`: a
// pointer is needed again a whole iteration later), models and output pointers are loaded where they are used: 32
// instructions per symbol, nothing of the chain on the stack.  Same arithmetic, step by step, as decode_symbols_noise.
// EPYC 9575F, one thread, four noise streams: 437 against 423 Msym/s (profiles/r04/m_*): the loop is bound by neither its
// chains nor its instruction count alone -- five to eight streams with state and range packed into one register each (37
// instructions per symbol, built and measured, not kept) reach 440 / 467 / 478 / 469 Msym/s per thread at 88 .. 59 per stream,
// and six streams to a loop in the pool 14.9 against 15.2 GB/s on the same box (n_bench_k8_dec*.json).
struct NoiseCtx {
    const uint8_t* q[4];
    const BlockModel* mod[4];
    uint8_t* sym[4];
};
#define WR_NOISE_STREAM(K, L2, RG)                                                                          \
    "movq   " #K "*8(%[ctx]), %%rsi\n\t"                                                                     \
    "movzbl (%%rsi), %%esi\n\t"                                                                              \
    "movl   %" L2 ", %%eax\n\t"                                                                              \
    "shll   $8, %%eax\n\t"                                                                                   \
    "orl    %%esi, %%eax\n\t"                                                                                \
    "movl   %" RG ", %%edx\n\t"                                                                              \
    "shll   $8, %%edx\n\t"                                                                                   \
    "cmpl   $0x800001, %" RG "\n\t"                                                                          \
    "cmovbl %%eax, %" L2 "\n\t"                                                                              \
    "cmovbl %%edx, %" RG "\n\t"                                                                              \
    "adcq   $0, " #K "*8(%[ctx])\n\t"                                                                        \
    "cmpl   $0x800000, %" RG "\n\t"                                                                          \
    "jbe    2" #K "f\n"                                                                                      \
    "1" #K ":\n\t"                                                                                           \
    "movl   %" RG ", %%eax\n\t"                                                                              \
    "imulq  $1172812403, %%rax, %%rcx\n\t" /* range / 60000 by multiply-shift (exact for 32 bits) */         \
    "shrq   $46, %%rcx\n\t"                                                                                  \
    "movl   %" L2 ", %%eax\n\t"                                                                              \
    "shrl   %%eax\n\t"                                                                                       \
    "xorl   %%edx, %%edx\n\t"                                                                                \
    "divl   %%ecx\n\t"                                                                                       \
    "movq   32+" #K "*8(%[ctx]), %%rsi\n\t"                                                                  \
    "movzbl %c[lk](%%rsi,%%rax), %%eax\n\t"                                                                  \
    "movl   (%%rsi,%%rax,8), %%edx\n\t"                                                                      \
    "imull  %%ecx, %%edx\n\t"                                                                                \
    "cmpl   %%eax, %c[tp](%%rsi)\n\t"                                                                        \
    "je     3" #K "f\n\t"                                                                                    \
    "imull  4(%%rsi,%%rax,8), %%ecx\n\t"                                                                     \
    "movl   %%ecx, %" RG "\n"                                                                                \
    "4" #K ":\n\t"                                                                                           \
    "addl   %%edx, %%edx\n\t"                                                                                \
    "subl   %%edx, %" L2 "\n\t"                                                                              \
    "movq   64+" #K "*8(%[ctx]), %%rsi\n\t"                                                                  \
    "movb   %%al, (%%rsi,%[i])\n\t"
#define WR_NOISE_TAILS(K, L2, RG)                                                                           \
    "2" #K ":\n\t" /* one byte was not enough (rare): the reference's loop, rangecod.c:294-302 */            \
    "movq   " #K "*8(%[ctx]), %%rsi\n"                                                                       \
    "5" #K ":\n\t"                                                                                           \
    "shll   $8, %" L2 "\n\t"                                                                                 \
    "movzbl (%%rsi), %%eax\n\t"                                                                              \
    "orl    %%eax, %" L2 "\n\t"                                                                              \
    "shll   $8, %" RG "\n\t"                                                                                 \
    "incq   %%rsi\n\t"                                                                                       \
    "cmpl   $0x800000, %" RG "\n\t"                                                                          \
    "jbe    5" #K "b\n\t"                                                                                    \
    "movq   %%rsi, " #K "*8(%[ctx])\n\t"                                                                     \
    "jmp    1" #K "b\n"                                                                                      \
    "3" #K ":\n\t" /* the largest symbol present takes what is left of the range (rangecod.c:345-348) */     \
    "subl   %%edx, %" RG "\n\t"                                                                              \
    "jmp    4" #K "b\n"
void decode_symbols_noise4_asm(Dec* const* ds, uint8_t* const* dst, const BlockModel* const* ms)
{
    NoiseCtx ctx;
    for (int k = 0; k < 4; k++) { ctx.q[k] = ds[k]->in + ds[k]->pos; ctx.mod[k] = ms[k]; ctx.sym[k] = dst[k]; }
    register uint32_t l0 asm("r8") = (ds[0]->low << 1) | (ds[0]->held & 1u), l1 asm("r9") = (ds[1]->low << 1) | (ds[1]->held & 1u);
    register uint32_t l2 asm("r10") = (ds[2]->low << 1) | (ds[2]->held & 1u), l3 asm("r11") = (ds[3]->low << 1) | (ds[3]->held & 1u);
    register uint32_t r0 asm("r12") = ds[0]->range, r1 asm("r13") = ds[1]->range, r2 asm("r14") = ds[2]->range, r3 asm("r15") = ds[3]->range;
    register uint64_t i asm("rbx") = 0;
    register NoiseCtx* c asm("rdi") = &ctx;
    __asm__ volatile(
        ".p2align 5\n"
        "9:\n\t"
        WR_NOISE_STREAM(0, "k[l0]", "k[r0]")
        WR_NOISE_STREAM(1, "k[l1]", "k[r1]")
        WR_NOISE_STREAM(2, "k[l2]", "k[r2]")
        WR_NOISE_STREAM(3, "k[l3]", "k[r3]")
        "incq   %[i]\n\t"
        "cmpq   %[n], %[i]\n\t"
        "jne    9b\n\t"
        "jmp    8f\n"
        WR_NOISE_TAILS(0, "k[l0]", "k[r0]")
        WR_NOISE_TAILS(1, "k[l1]", "k[r1]")
        WR_NOISE_TAILS(2, "k[l2]", "k[r2]")
        WR_NOISE_TAILS(3, "k[l3]", "k[r3]")
        "8:\n"
        : [l0] "+r"(l0), [l1] "+r"(l1), [l2] "+r"(l2), [l3] "+r"(l3), [r0] "+r"(r0), [r1] "+r"(r1), [r2] "+r"(r2), [r3] "+r"(r3), [i] "+r"(i)
        : [ctx] "r"(c), [n] "i"((int)kBlock), [lk] "i"(offsetof(BlockModel, lookup)), [tp] "i"(offsetof(BlockModel, top))
        : "rax", "rcx", "rdx", "rsi", "cc", "memory");
    const uint32_t lw[4] = {l0, l1, l2, l3}, rg[4] = {r0, r1, r2, r3};
    for (int k = 0; k < 4; k++) {
        ds[k]->low = lw[k] >> 1; ds[k]->range = rg[k]; ds[k]->pos = (size_t)(ctx.q[k] - ds[k]->in); ds[k]->held = ctx.q[k][-1];
    }
}
#undef WR_NOISE_STREAM
#undef WR_NOISE_TAILS

constexpr bool kNoiseAsm = true;
#else
constexpr bool kNoiseAsm = false;
inline void decode_symbols_noise4_asm(Dec* const*, uint8_t* const*, const BlockModel* const*) {}
#endif

using MultiFn = void (*)(Dec* const*, uint8_t* const*, const BlockModel* const*);
template <int NS, unsigned... M>
constexpr MultiFn multi_entry(unsigned mask, std::integer_sequence<unsigned, M...>)
{
    constexpr MultiFn table[] = {&decode_symbols_multi<NS, M>...};
    return table[mask];
}
inline void decode_block_multi(int count, Dec* const* ds, uint8_t* const* dst, const BlockModel* const* ms)
{
    unsigned mask = 0, buckets = 0;
    for (int k = 0; k < count; k++) { mask |= (ms[k]->mps_on ? 1u : 0u) << k; buckets |= ms[k]->use_buckets ? 1u : 0u; }
    if (!mask && !buckets) {
        switch (count) {
        case 1: decode_symbols_noise<1>(ds, dst, ms); break;
        case 2: decode_symbols_noise<2>(ds, dst, ms); break;
        case 3: decode_symbols_noise<3>(ds, dst, ms); break;
        default: if (kNoiseAsm) decode_symbols_noise4_asm(ds, dst, ms); else decode_symbols_noise<4>(ds, dst, ms); break;
        }
        return;
    }
    switch (count) {
    case 1: multi_entry<1>(mask, std::make_integer_sequence<unsigned, 2>())(ds, dst, ms); break;
    case 2: multi_entry<2>(mask, std::make_integer_sequence<unsigned, 4>())(ds, dst, ms); break;
    case 3: multi_entry<3>(mask, std::make_integer_sequence<unsigned, 8>())(ds, dst, ms); break;
    default: multi_entry<4>(mask, std::make_integer_sequence<unsigned, 16>())(ds, dst, ms); break;
    }
}

// A decoder stream between two blocks, as it moves from one worker's session to another's.  `d.in` may point into
// `tail` (the zero-padded copy of the stream's end): moving the vector keeps its buffer where it is.
struct DecStream {
    Dec d;
    SymCursor cur;
    size_t n, produced;
    bool failed;
    void* tag;
    std::vector<uint8_t> tail;
};

// Slow-path block of one stream (partial block, window or stream about to end): decoded straight into what is left
// of the stream's window, or -- a block that straddles the end of a window although the plane goes on, which only
// happens behind a block of fewer than 60000 symbols in mid-stream: the reference's encoder never writes one, the
// format allows it -- into a bounce buffer whose parts go to this window and the following one(s).
inline void decode_block_checked(Dec& d, const BlockModel& m, SymCursor& cur, uint8_t* dst, size_t at, size_t n, std::vector<uint8_t>& bounce)
{
    if (!m.bs) return;
    const size_t room = cur.room(at);  // symbols beyond it are dropped (a stream that holds more than the plane has)
    const bool split = room < m.bs && at + room < n;
    if (split) { if (bounce.size() < kBlock) bounce.resize(kBlock); dst = bounce.data(); }
    if (m.bs == kBlock) decode_symbols<kBlock>(d, dst, split ? m.bs : room, m.bs, m.tab, m.lookup, m.top);
    else decode_symbols<0>(d, dst, split ? m.bs : room, m.bs, m.tab, m.lookup, m.top);
    if (!split) return;
    size_t done = 0;
    while (done < m.bs && at + done < n) {
        uint8_t* const w = cur.at(at + done, n);  // moves the window on when `at + done` has run out of it
        if (!w) break;  // refused: the caller sees cur.refused
        size_t r = cur.room(at + done);
        if (r > m.bs - done) r = m.bs - done;
        if (!r) break;
        memcpy(w, bounce.data() + done, r);
        done += r;
    }
}

// A set of up to kMaxDecStreams plane streams advancing block by block in lockstep on one thread
// (wrappers.cpp:153-224 per stream).  Streams join at any block boundary and leave when they end.
class DecGroup {
public:
    explicit DecGroup(int cap) : cap_(cap < 1 ? 1 : (cap > kMaxDecStreams ? kMaxDecStreams : cap)), models_((size_t)cap_)
    {
        for (int k = 0; k < cap_; k++) ms_[k] = &models_[k];
    }
    int count() const { return count_; }
    bool full() const { return count_ == cap_; }
    void add(const uint8_t* in, size_t len, uint8_t* sym, size_t n, void* tag, const PlaneWindow* io = nullptr)
    {
        const int k = count_++;
        ds_[k] = new (store_[k]) Dec(in, len);
        cur_[k].set(sym, n, io); n_[k] = n; produced_[k] = 0; failed_[k] = false; tag_[k] = tag;
        tails_[k].clear();
    }
    // hand the last stream over (call between steps) / adopt one
    DecStream give()
    {
        const int k = --count_;
        DecStream m{*ds_[k], cur_[k], n_[k], produced_[k], failed_[k], tag_[k], std::move(tails_[k])};
        tails_[k].clear();
        return m;
    }
    void take(DecStream&& m)
    {
        const int k = count_++;
        ds_[k] = new (store_[k]) Dec(m.d);
        cur_[k] = m.cur; n_[k] = m.n; produced_[k] = m.produced; failed_[k] = m.failed; tag_[k] = m.tag;
        tails_[k] = std::move(m.tail);
    }
    void set_tag_of_last(void* t) { tag_[count_ - 1] = t; }
    // one block of every stream; on_end(tag, symbols the stream held or (size_t)-1) for the streams that ended
    template <class OnEnd>
    void step(OnEnd on_end)
    {
        constexpr size_t kMargin = 3 * (size_t)kBlock + 8;  // a symbol pulls in at most 3 bytes
        bool fast = true;
        for (int k = 0; k < count_;) {
            Dec& d = *ds_[k];
            BlockModel& m = *ms_[k];
            bool ended = false;
            if (!d.culfreq(2)) { d.renorm(); ended = true; }  // done_decoding, rangecod.c:371-373
            else {
                d.update(1, 1, 2);
                uint32_t bs = 0, top_sym = 0;
                for (int b = 0; b < 256; b++) {
                    uint32_t c = d.culshift(16) & 0xffffu;  // decode_short, rangecod.c:362-366
                    d.update(1, c, 1u << 16);
                    m.tab[b].lt = bs; m.tab[b].sy = c;
                    bs += c;
                    if (c) top_sym = (uint32_t)b;
                }
                if (bs > kBlock) { failed_[k] = true; ended = true; }  // not a WaveRange stream (defs.h:36)
                else {
                    m.top = top_sym; m.bs = bs;
                    finish_model(m);
                }
            }
            if (ended) { retire(k, on_end); continue; }
            const size_t at = produced_[k] < n_[k] ? produced_[k] : n_[k];
            dst_[k] = cur_[k].at(at, n_[k]);
            if (cur_[k].refused) { failed_[k] = true; retire(k, on_end); continue; }  // the plane's owner turned the window down
            if (d.pos + kMargin > d.len && tails_[k].empty() && d.pos >= 1 && d.pos <= d.len) {
                // near the end of the stream: continue on a zero-padded copy of the rest (reading past
                // the end yields zeros, Dec::get), so that the unchecked loop stays usable
                tails_[k].assign(d.len - (d.pos - 1) + kMargin, 0);
                memcpy(tails_[k].data(), d.in + (d.pos - 1), d.len - (d.pos - 1));
                d.in = tails_[k].data(); d.len = tails_[k].size(); d.pos = 1;
            }
            if (m.bs != kBlock || cur_[k].room(at) < kBlock || d.pos + kMargin > d.len) fast = false;
            k++;
        }
        if (!count_) return;
        if (fast) {
            decode_block_multi(count_, ds_, dst_, ms_);
            for (int k = 0; k < count_; k++) produced_[k] += kBlock;
            return;
        }
        for (int k = 0; k < count_;) {
            const BlockModel& m = *ms_[k];
            decode_block_checked(*ds_[k], m, cur_[k], dst_[k], produced_[k] < n_[k] ? produced_[k] : n_[k], n_[k], bounce_);
            produced_[k] += m.bs;
            if (ds_[k]->pos > ds_[k]->len + 8 || cur_[k].refused) { failed_[k] = true; retire(k, on_end); continue; }  // ran far past the end: corrupt stream
            k++;
        }
    }

private:
    template <class OnEnd>
    void retire(int k, OnEnd on_end)
    {
        cur_[k].end(produced_[k] < n_[k] ? produced_[k] : n_[k]);
        on_end(tag_[k], failed_[k] ? (size_t)-1 : produced_[k]);
        const int last = --count_;
        if (k != last) {  // the last slot moves into the hole
            ds_[k] = new (store_[k]) Dec(*ds_[last]);
            std::swap(ms_[k], ms_[last]);
            cur_[k] = cur_[last]; n_[k] = n_[last]; produced_[k] = produced_[last]; failed_[k] = failed_[last]; tag_[k] = tag_[last];
            dst_[k] = dst_[last];
            // Dec may point into its tail copy: the vector's buffer moves with it
            tails_[k].swap(tails_[last]);
        }
        tails_[last].clear();
    }

    int cap_, count_ = 0;
    std::vector<BlockModel> models_;
    BlockModel* ms_[kMaxDecStreams];
    Dec* ds_[kMaxDecStreams];
    alignas(Dec) unsigned char store_[kMaxDecStreams][sizeof(Dec)];
    SymCursor cur_[kMaxDecStreams];
    uint8_t* dst_[kMaxDecStreams];
    size_t n_[kMaxDecStreams], produced_[kMaxDecStreams];
    bool failed_[kMaxDecStreams];
    void* tag_[kMaxDecStreams];
    std::vector<uint8_t> tails_[kMaxDecStreams];
    std::vector<uint8_t> bounce_;  // decode_block_checked
};

}  // namespace

namespace {

// a symbol outside the two dominant ones, for one lane of the vector loop: the look-up path of decode_symbols
uint32_t vec_other_symbol(const void* model, uint32_t* low, uint32_t* range, uint32_t help)
{
    BlockModel& m = *const_cast<BlockModel*>(static_cast<const BlockModel*>(model));
    if (!m.tables_ready) finish_model_tables(m);
    const uint32_t cf = *low / help;
    const uint32_t c = m.lookup[cf];
    const uint32_t t = help * m.tab[c].lt;
    *low -= t;
    *range = (c != m.top) ? help * m.tab[c].sy : *range - t;
    return c;
}

// Up to 16 plane streams whose blocks are mostly held by at most four symbols, advancing block by block in lockstep on
// one thread: the blocks of that kind are decoded 16 lanes at a time by the AVX-512 loop, the occasional other
// block by the scalar loop of its stream.
class VecDecGroup {
public:
    static constexpr int kCap = kVecLanes;
    VecDecGroup() : models_((size_t)kCap)
    {
        for (int k = 0; k < kCap; k++) ms_[k] = &models_[k];
    }
    int count() const { return count_; }
    bool full() const { return count_ == kCap; }
    void add(const uint8_t* in, size_t len, uint8_t* sym, size_t n, void* tag, const PlaneWindow* io = nullptr)
    {
        const int k = count_++;
        ds_[k] = new (store_[k]) Dec(in, len);
        cur_[k].set(sym, n, io); n_[k] = n; produced_[k] = 0; failed_[k] = false; tag_[k] = tag;
        tails_[k].clear();
    }
    DecStream give()
    {
        const int k = --count_;
        DecStream m{*ds_[k], cur_[k], n_[k], produced_[k], failed_[k], tag_[k], std::move(tails_[k])};
        tails_[k].clear();
        return m;
    }
    void take(DecStream&& m)
    {
        const int k = count_++;
        ds_[k] = new (store_[k]) Dec(m.d);
        cur_[k] = m.cur; n_[k] = m.n; produced_[k] = m.produced; failed_[k] = m.failed; tag_[k] = m.tag;
        tails_[k] = std::move(m.tail);
    }
    void set_tag_of_last(void* t) { tag_[count_ - 1] = t; }
    template <class OnEnd>
    void step(OnEnd on_end)
    {
        WR_PROBE(3);
        VecBlock vb;
        if (!prepare(on_end, vb)) return;
        if (vb.active) vec_decode_block(&vb, vec_other_symbol);
        finish(vb, on_end);
    }
    // A step in two halves.  prepare: block headers of all streams (streams that end leave), the lanes of the vector loop in
    // vb (vb.active may be 0); false: nothing left to do in this step (no streams, or they went through another loop
    // already).  finish: lane state back to the streams, the other streams' blocks through the scalar loop.
    template <class OnEnd>
    bool prepare(OnEnd on_end, VecBlock& vb)
    {
        constexpr size_t kMargin = 3 * (size_t)kBlock + 32;  // a symbol pulls in at most 3 bytes; the vector loop reads two windows ahead
        bool* vec = vec_;
        bool all_fast = true;
        vb.active = 0;
        for (int k = 0; k < count_;) {
            Dec& d = *ds_[k];
            BlockModel& m = *ms_[k];
            bool ended = false;
            {
            WR_PROBE(2);
            if (!d.culfreq(2)) { d.renorm(); ended = true; }
            else {
                d.update(1, 1, 2);
                uint32_t bs = 0, top_sym = 0;
                for (int b = 0; b < 256; b++) {
                    uint32_t c = d.culshift(16) & 0xffffu;
                    d.update(1, c, 1u << 16);
                    m.tab[b].lt = bs; m.tab[b].sy = c;
                    bs += c;
                    if (c) top_sym = (uint32_t)b;
                }
                if (bs > kBlock) { failed_[k] = true; ended = true; }
                else {
                    m.top = top_sym; m.bs = bs;
                    finish_model_stats(m);
                }
            }
            }
            if (ended) { retire(k, on_end); continue; }
            const size_t at = produced_[k] < n_[k] ? produced_[k] : n_[k];
            dst_[k] = cur_[k].at(at, n_[k]);
            if (cur_[k].refused) { failed_[k] = true; retire(k, on_end); continue; }
            if (d.pos + kMargin > d.len && tails_[k].empty() && d.pos >= 1 && d.pos <= d.len) {
                tails_[k].assign(d.len - (d.pos - 1) + kMargin, 0);
                memcpy(tails_[k].data(), d.in + (d.pos - 1), d.len - (d.pos - 1));
                d.in = tails_[k].data(); d.len = tails_[k].size(); d.pos = 1;
            }
            const bool fast = m.bs == kBlock && cur_[k].room(at) >= kBlock && d.pos + kMargin <= d.len;
            vec[k] = fast && m.cand_ok;
            all_fast = all_fast && fast;
            k++;
        }
        if (!count_) return false;
        if (all_fast && count_ < kMaxDecStreams) {  // down to a few streams: the scalar loop of up to three
            for (int k = 0; k < count_; k++)
                if (!ms_[k]->tables_ready) finish_model_tables(*ms_[k]);
            decode_block_multi(count_, ds_, dst_, ms_);
            for (int k = 0; k < count_; k++) produced_[k] += kBlock;
            return false;
        }
        for (int k = 0; k < count_; k++) {
            if (!vec[k]) continue;
            const Dec& d = *ds_[k];
            const BlockModel& m = *ms_[k];
            vb.active |= 1u << k;
            vb.low[k] = d.low; vb.range[k] = d.range; vb.ptr[k] = d.in + d.pos; vb.dst[k] = dst_[k];
            // the candidates that exist, sorted by interval start; a lane with fewer than four repeats its last one
            // (vec_decode_block picks the last candidate whose start is at or below `low`)
            int order[kVecCand], nc = 0;
            for (int e = 0; e < kVecCand; e++)
                if (m.cand_sy[e]) {
                    int i = nc++;
                    while (i > 0 && m.cand_lt[order[i - 1]] > m.cand_lt[e]) { order[i] = order[i - 1]; i--; }
                    order[i] = e;
                }
            for (int e = 0; e < kVecCand; e++) {
                const int o = order[e < nc ? e : nc - 1];
                vb.lt[e][k] = m.cand_lt[o]; vb.sy[e][k] = m.cand_sy[o]; vb.sym[e][k] = m.cand[o];
                vb.is_top[e][k] = m.cand[o] == m.top;
            }
            vb.model[k] = &m;
        }
        if (vb.active)
            for (int k = 0; k < kCap; k++)
                if (!(vb.active >> k & 1)) { for (int e = 0; e < kVecCand; e++) vb.is_top[e][k] = 0; vb.ptr[k] = nullptr; vb.dst[k] = nullptr; }
        return true;
    }
    template <class OnEnd>
    void finish(const VecBlock& vb, OnEnd on_end)
    {
        bool* vec = vec_;
        if (vb.active)
            for (int k = 0; k < count_; k++) {
                if (!vec[k]) continue;
                Dec& d = *ds_[k];
                d.low = vb.low[k]; d.range = vb.range[k]; d.pos = (size_t)(vb.ptr[k] - d.in); d.held = vb.ptr[k][-1];
                produced_[k] += kBlock;
            }
        finish_slow(vec, on_end);
    }

private:
    // the lanes the vector loop did not take in this step (partial block, window or stream about to end; no dominant
    // symbols in the candidate mode): scalar, checked
    template <class OnEnd>
    void finish_slow(bool* vec, OnEnd on_end)
    {
        for (int k = 0; k < count_;) {
            if (vec[k]) { k++; continue; }
            BlockModel& m = *ms_[k];
            if (!m.tables_ready) finish_model_tables(m);
            decode_block_checked(*ds_[k], m, cur_[k], dst_[k], produced_[k] < n_[k] ? produced_[k] : n_[k], n_[k], bounce_);
            produced_[k] += m.bs;
            if (ds_[k]->pos > ds_[k]->len + 8 || cur_[k].refused) {
                failed_[k] = true;
                const int last = count_ - 1;
                retire(k, on_end);
                if (k != last) vec[k] = vec[last];  // the slot that moved in: done already if it was a vector lane
                continue;
            }
            k++;
        }
    }
    template <class OnEnd>
    void retire(int k, OnEnd on_end)
    {
        cur_[k].end(produced_[k] < n_[k] ? produced_[k] : n_[k]);
        on_end(tag_[k], failed_[k] ? (size_t)-1 : produced_[k]);
        const int last = --count_;
        if (k != last) {
            ds_[k] = new (store_[k]) Dec(*ds_[last]);
            std::swap(ms_[k], ms_[last]);
            cur_[k] = cur_[last]; n_[k] = n_[last]; produced_[k] = produced_[last]; failed_[k] = failed_[last]; tag_[k] = tag_[last];
            dst_[k] = dst_[last];
            tails_[k].swap(tails_[last]);
        }
        tails_[last].clear();
    }

    bool vec_[kCap];  // this step: the stream's block goes through the vector loop
    int count_ = 0;
    std::vector<BlockModel> models_;
    BlockModel* ms_[kCap];
    Dec* ds_[kCap];
    alignas(Dec) unsigned char store_[kCap][sizeof(Dec)];
    SymCursor cur_[kCap];
    uint8_t* dst_[kCap];
    size_t n_[kCap], produced_[kCap];
    bool failed_[kCap];
    void* tag_[kCap];
    std::vector<uint8_t> tails_[kCap];
    std::vector<uint8_t> bounce_;  // decode_block_checked
};

}  // namespace

// `count` streams of dominant-symbol planes (any lengths) on the calling thread through the 16-lane loop: test
// and measurement hook; the coder pool is the product path.  False if the CPU lacks AVX-512.
bool decode_planes_vec(int count, const uint8_t* const* in, const size_t* len, uint8_t* const* sym, const size_t* n, size_t* produced,
                       const PlaneWindow* const* io)
{
    if (!vec_available()) return false;
    auto run = [&](auto& g) {
        int next = 0;
        while (next < count || g.count()) {
            while (next < count && !g.full()) {
                produced[next] = 0;
                g.add(in[next], len[next], sym[next], n[next], produced + next, io ? io[next] : nullptr);
                next++;
            }
            g.step([](void* tag, size_t got) { *static_cast<size_t*>(tag) = got; });
        }
    };
    std::unique_ptr<VecDecGroup> g(new VecDecGroup);
    run(*g);
    return true;
}

void decode_planes(int count, const uint8_t* const* in, const size_t* len, uint8_t* const* sym, size_t n, size_t* produced,
                   const PlaneWindow* const* io)
{
    // up to kMaxStreams planes at a time in one symbol loop; further planes join as earlier ones end
    DecGroup g(kMaxStreams);
    int next = 0;
    while (next < count || g.count()) {
        while (next < count && !g.full()) {
            produced[next] = 0;
            g.add(in[next], len[next], sym[next], n, produced + next, io ? io[next] : nullptr);
            next++;
        }
        g.step([](void* tag, size_t got) { *static_cast<size_t*>(tag) = got; });
    }
}

size_t decode_plane(const uint8_t* in, size_t len, uint8_t* sym, size_t n)
{
    size_t produced = 0;
    decode_planes(1, &in, &len, &sym, n, &produced);
    return produced;
}

// =====================================================================================
// Process-wide coder pool
// =====================================================================================
namespace {

class Pool {
public:
    static Pool& get() { static Pool p; return p; }

    int threads()
    {
        std::lock_guard<std::mutex> lk(mu_);
        return (int)workers_.size();
    }
    void resize(int nthreads, int dec_streams)
    {
        std::unique_lock<std::mutex> lk(mu_);
        if (dec_streams >= 1) dec_streams_ = dec_streams > kMaxDecStreams ? kMaxDecStreams : dec_streams;
        if (nthreads < 0) nthreads = 0;
        if (nthreads == (int)workers_.size()) return;
        // stop the present workers (they finish what they hold and what is queued), then start the new set
        std::vector<std::thread> old;
        old.swap(workers_);
        stop_ = true;
        cv_.notify_all();
        lk.unlock();
        for (auto& t : old) t.join();
        lk.lock();
        stop_ = false;
        lanes_.assign((size_t)nthreads, 0);
        for (int i = 0; i < nthreads; i++) workers_.emplace_back([this, i] { run(i); });
    }
    // false: the pool has no workers (never started, or stopped meanwhile by another thread) -- nothing was queued and
    // the caller codes the planes itself
    bool submit(PlaneJob* jobs, int count, JobBatch* batch)
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (workers_.empty()) return false;
        const double t_sub = now_s();
        for (int i = 0; i < count; i++) {
            jobs[i].batch = batch;
            jobs[i].submitted = t_sub;
            // planes below 2 bits per symbol are the dominant-symbol kind: they go to the 16-lane vector loop
            const bool vec = jobs[i].kind == PlaneJob::kDecode && vec_ok_ && jobs[i].n >= 4 * (size_t)kBlock &&
                             8 * jobs[i].src_len < 2 * jobs[i].n;
            // encoder: every plane takes the vector loop (candidate compares while all lanes of a session hold
            // dominant-symbol blocks, gathers from the lanes' tables otherwise)
            const bool venc = jobs[i].kind == PlaneJob::kEncode && vec_ok_ && vec_enc_ && jobs[i].n >= 4 * (size_t)kBlock;
            (vec ? vec_q_ : venc ? venc_q_ : jobs[i].kind == PlaneJob::kDecode ? dec_q_ : enc_q_).push_back(&jobs[i]);
        }
        { std::lock_guard<std::mutex> bl(batch->mu); batch->remaining += count; }
        if (count > 1) cv_.notify_all(); else cv_.notify_one();
        return true;
    }
    ~Pool() { resize(0, 0); }

private:
    enum Want { kAny = -1, kEnc = 0, kDec = 1, kVec = 2, kVecEnc = 3 };
    // kAny: vector-decode jobs first (one worker absorbs up to 16 of them), then decode, then encode
    PlaneJob* pop(bool block, int want, int* got = nullptr)
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            // at most vec_sessions_max() / venc_sessions_max() workers run a vector session of either kind at a time
            // (5/16 and 3/8 of the workers): a session is worth its core
            // with many lanes filled, but every stream in it advances the slower the fuller it is, so their number is
            // what balances CPU time against the time a field waits for its planes (profiles/r02/NOTES.md); what is
            // queued beyond that joins a running session at its next block boundary
            if (want == kVec && !vec_q_.empty()) { PlaneJob* j = vec_q_.front(); vec_q_.pop_front(); return j; }
            if (want == kAny && !vec_q_.empty() && vec_sessions_ < vec_sessions_max()) {
                PlaneJob* j = vec_q_.front(); vec_q_.pop_front(); vec_sessions_++; if (got) *got = kVec; return j;
            }
            if (want == kVecEnc && !venc_q_.empty()) { PlaneJob* j = venc_q_.front(); venc_q_.pop_front(); return j; }
            if (want == kAny && !venc_q_.empty() && venc_sessions_ < venc_sessions_max()) {
                PlaneJob* j = venc_q_.front(); venc_q_.pop_front(); venc_sessions_++; if (got) *got = kVecEnc; return j;
            }
            if ((want == kAny || want == kDec) && !dec_q_.empty()) { PlaneJob* j = dec_q_.front(); dec_q_.pop_front(); if (got) *got = kDec; return j; }
            if ((want == kAny || want == kEnc) && !enc_q_.empty()) { PlaneJob* j = enc_q_.front(); enc_q_.pop_front(); if (got) *got = kEnc; return j; }
            if (!block || stop_) return nullptr;
            const double t = now_s();
            cv_.wait(lk);
            idle_s_ += now_s() - t;
        }
    }
public:
    double idle_seconds() { std::lock_guard<std::mutex> lk(mu_); return idle_s_; }
    void queued_decode(int* scalar_jobs, int* vector_jobs) { std::lock_guard<std::mutex> lk(mu_); *scalar_jobs = (int)dec_q_.size(); *vector_jobs = (int)vec_q_.size(); }
    void set_steal_idle_test(int workers) { std::lock_guard<std::mutex> lk(mu_); steal_idle_test_ = workers < 0 ? 0 : workers; }
    double queue_seconds() { return queued_ns_.load() * 1e-9; }
    unsigned long streams_moved() { return moved_.load(); }
    // per loop kind (scalar encoder, scalar decoder, vector decoder, vector encoder): worker seconds spent in its block
    // steps and stream-blocks (60000 symbols each, the last block of a stream counted whole) they advanced
    void loop_stats(double seconds[kLoopKinds], double blocks[kLoopKinds])
    {
        for (int k = 0; k < kLoopKinds; k++) { seconds[k] = loop_ns_[k].load() * 1e-9; blocks[k] = (double)loop_blocks_[k].load(); }
    }
private:
    std::atomic<unsigned long long> loop_ns_[kLoopKinds] = {}, loop_blocks_[kLoopKinds] = {};
    std::atomic<unsigned long long> queued_ns_{0};  // jobs waiting in the queues for a worker, summed
    void account(int kind, double s, int nstreams)
    {
        loop_ns_[kind] += (unsigned long long)(s * 1e9);
        loop_blocks_[kind] += (unsigned long long)nstreams;
    }
    double idle_s_ = 0;  // worker time spent waiting for a job, all workers
    static void finish(PlaneJob* j, size_t result, double t0)
    {
        j->result = result;
        j->seconds = now_s() - t0;
        JobBatch* b = j->batch;
        std::lock_guard<std::mutex> lk(b->mu);
        if (--b->remaining == 0) b->cv.notify_all();
    }
    static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    // ---- streams change workers.  The fields in flight are bounded by memory, so what sets the whole-job rate is how fast
    // the slowest plane stream of every field advances, and a stream advances the faster the fewer streams share its
    // worker's loop.  A worker that finds nothing queued therefore takes over half of the streams of the fullest running
    // session, at that session's next block boundary (every 60000 symbols per stream: a millisecond or so): the session's
    // owner packs them up (offer), the idle worker adopts them (next) -- same kind of loop, same bytes, the streams only
    // change threads between two blocks.
    struct Tag { PlaneJob* job; double t0; };
    static constexpr int kSessionTags = kVecLanes;  // streams a session can hold
    struct Handoff {
        int kind = kAny;
        std::vector<DecStream> dec;
        std::vector<EncStream> enc;
        std::vector<Tag> tags;  // job and start time of every stream, in the same order
        bool empty() const { return dec.empty() && enc.empty(); }
    };
    std::deque<Handoff> handoff_;
    int thieves_ = 0;            // workers waiting with nothing queued for them (mu_)
    std::vector<int> lanes_;     // streams in every worker's running session, by worker index (mu_)
    int steal_idle_test_ = 0;    // tests: this many idle workers are enough for a hand-over (pool_test_steal_idle_min; 0: the rule below)
    int steal_idle_min() const  // mu_ held
    {
        if (steal_idle_test_ > 0) return steal_idle_test_;
        const int w = (int)workers_.size();
        // (an eighth of the workers: two of 16.  A quarter until the end of round 5 -- from when the workers stopped blocking on
        // window copies, idle workers are idle for want of streams, and two of them taking over half a session pay:
        // 16.90 / 17.00 -> 17.54 / 17.61 GB/s same box, 1.05 -> 0.63-0.9 workers idle; one idle worker: 17.62 / 17.37:
        // profiles/r05/z_ab_steal_threshold_k10_same_box.txt)
        return w < 4 ? 1 : (w + 7) / 8;
    }
    std::atomic<unsigned long> moved_{0};

    // what a worker does next: a queued job (*j, the session kind in *kind) or streams handed over by another worker (*h);
    // false: the pool is stopping and nothing is left
    bool next(int* kind, PlaneJob** j, Handoff* h)
    {
        {
            std::unique_lock<std::mutex> lk(mu_);
            if (!handoff_.empty()) { *h = std::move(handoff_.front()); handoff_.pop_front(); *kind = h->kind; *j = nullptr; return true; }
        }
        for (;;) {
            *j = pop(false, kAny, kind);
            if (*j) return true;
            std::unique_lock<std::mutex> lk(mu_);
            if (!handoff_.empty()) { *h = std::move(handoff_.front()); handoff_.pop_front(); *kind = h->kind; return true; }
            if (!(vec_q_.empty() && venc_q_.empty() && dec_q_.empty() && enc_q_.empty()) && startable()) continue;
            if (stop_) return false;
            const double t = now_s();
            thieves_++;
            cv_.wait(lk);
            thieves_--;
            idle_s_ += now_s() - t;
        }
    }
    // something queued that a worker without a session may start now (mu_ held)
    bool startable() const
    {
        return (!vec_q_.empty() && vec_sessions_ < vec_sessions_max()) ||
               (!venc_q_.empty() && venc_sessions_ < venc_sessions_max()) || !dec_q_.empty() || !enc_q_.empty();
    }
    // A session's owner, between two blocks: publishes how many streams it holds and, if a worker is waiting and no
    // session is fuller, hands over half of them.
    template <class G, class Pack>
    void offer(int id, int kind, G& g, Pack pack)
    {
        std::lock_guard<std::mutex> lk(mu_);
        lanes_[(size_t)id] = g.count();
        // Only when part of the pool has nothing to do (steal_idle_min: an eighth of the workers): a
        // lone caller, a short batch, the drain of a run.  In steady state with every worker busy most of the time, a
        // worker that is idle for a moment would split a well-filled session into two half-filled ones, and a 16-lane
        // loop at 8 lanes does 70 % of the work per second: measured 10.7 GB/s against 11.8 without (profiles/r03).
        if (thieves_ <= (int)handoff_.size() || thieves_ < steal_idle_min() || g.count() < 2) return;
        for (int c : lanes_) if (c > g.count()) return;  // a fuller session does it at its next boundary
        Handoff h;
        h.kind = kind;
        for (int k = g.count() / 2; k > 0; k--) pack(g, &h);
        if (h.tags.empty()) return;  // (a kind whose streams stay put)
        lanes_[(size_t)id] = g.count();
        moved_ += (unsigned long)h.tags.size();
        handoff_.push_back(std::move(h));
        cv_.notify_one();
    }
    void session_over(int id, int* counter)
    {
        { std::lock_guard<std::mutex> lk(mu_); lanes_[(size_t)id] = 0; if (counter) (*counter)--; }
        if (counter) cv_.notify_all();  // jobs of that kind queued meanwhile may start a session of their own now
    }

    // One session: the streams of one kind this worker interleaves, topped up from the kind's queue and thinned out by
    // offer() at block boundaries.  Starts with one queued job or with the streams of a hand-over.
    template <class G, class Add, class Adopt, class Pack>
    void session(int id, int kind, int stat, G& g, PlaneJob* j, Handoff& h, Tag* tags, Add add, Adopt adopt, Pack pack)
    {
        // A session that began with handed-over streams only carries those to their ends: the number of sessions that
        // live on by topping up from the queues stays what the caps say (they balance CPU time against the time a field
        // waits for its planes; a session that tops up never ends while jobs keep coming).
        const bool tops_up = j != nullptr;
        auto free_tag = [&]() -> Tag* { for (int i = 0; i < kSessionTags; i++) if (!tags[i].job) return &tags[i]; return nullptr; };
        auto on_end = [](void* tag, size_t result) { Tag* t = static_cast<Tag*>(tag); finish(t->job, result, t->t0); t->job = nullptr; };
        for (size_t i = 0; i < h.tags.size(); i++) {
            Tag* t = free_tag();
            *t = h.tags[i];
            adopt(g, h, i, t);
        }
        h = Handoff();
        while (j || g.count()) {
            while (j) {
                Tag* t = free_tag();
                t->job = j; t->t0 = now_s();
                queued_ns_ += (unsigned long long)((t->t0 - j->submitted) * 1e9);
                add(g, j, t);
                j = g.full() ? nullptr : pop(false, kind);
            }
            { const int nstreams = g.count(); const double ts = now_s(); g.step(on_end); account(stat, now_s() - ts, nstreams); }
            offer(id, kind, g, pack);
            if (tops_up && !g.full()) j = pop(false, kind);
        }
    }

    void run(int id)
    {
        {   // the workers carry a name (wr-coder-<id>): a caller that places threads on cores finds them in /proc/self/task
            char name[16];
            snprintf(name, sizeof name, "wr-coder-%d", id);
            (void)pthread_setname_np(pthread_self(), name);
        }
        int dec_streams;
        { std::lock_guard<std::mutex> lk(mu_); dec_streams = dec_streams_; }
        std::unique_ptr<DecGroup> dg;
        std::unique_ptr<VecDecGroup> vg;
        std::unique_ptr<VecEncGroup> veg;
        EncGroup eg;
        Tag tags[kSessionTags];
        auto add_dec = [](auto& g, PlaneJob* j, Tag* t) { g.add(j->src, j->src_len, j->dst, j->n, t, j->io); };
        auto add_enc = [](auto& g, PlaneJob* j, Tag* t) { g.add(j->src, j->n, j->dst, j->hist, t, j->io, j->dst_limit); };
        auto adopt_dec = [](auto& g, Handoff& h, size_t i, Tag* t) { g.take(std::move(h.dec[i])); g.set_tag_of_last(t); };
        auto adopt_enc = [](auto& g, Handoff& h, size_t i, Tag* t) { g.take(h.enc[i]); g.set_tag_of_last(t); };
        auto pack_dec = [](auto& g, Handoff* h) {
            DecStream m = g.give();
            Tag* t = static_cast<Tag*>(m.tag);
            h->tags.push_back(*t); t->job = nullptr;
            h->dec.push_back(std::move(m));
        };
        auto pack_enc = [](auto& g, Handoff* h) {
            EncStream m = g.give();
            Tag* t = static_cast<Tag*>(m.st.tag);
            h->tags.push_back(*t); t->job = nullptr;
            h->enc.push_back(m);
        };
        auto pack_none = [](auto&, Handoff*) {};
        for (;;) {
            int kind = kAny;
            PlaneJob* j = nullptr;
            Handoff h;
            if (!next(&kind, &j, &h)) return;
            for (Tag& t : tags) t.job = nullptr;
            const bool counted = j != nullptr;  // a session started from a queue counts against its kind's cap (pop)
            if (kind == kVec) {
                if (!vg) vg.reset(new VecDecGroup);
                session(id, kVec, 2, *vg, j, h, tags, add_dec, adopt_dec, pack_dec);
                session_over(id, counted ? &vec_sessions_ : nullptr);
            } else if (kind == kVecEnc) {
                if (!veg) veg.reset(new VecEncGroup);
                session(id, kVecEnc, 3, *veg, j, h, tags, add_enc, adopt_enc, pack_enc);
                session_over(id, counted ? &venc_sessions_ : nullptr);
            } else if (kind == kDec) {
                if (!dg) dg.reset(new DecGroup(dec_streams));
                session(id, kDec, 1, *dg, j, h, tags, add_dec, adopt_dec, pack_dec);
                session_over(id, nullptr);
            } else {
                // (scalar encoder loops of three: only without AVX-512 or with WR_VEC_ENCODE=0; their streams stay put)
                session(id, kEnc, 0, eg, j, h, tags, add_enc, [](EncGroup&, Handoff&, size_t, Tag*) {}, pack_none);
                session_over(id, nullptr);
            }
        }
    }

    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<PlaneJob*> enc_q_, dec_q_, vec_q_, venc_q_;
    // (A 16-lane decoder loop for planes of ANY statistics -- range / 60000 by multiply-shift, low / help by vdivpd, the two
    // table look-ups as scalar loads per lane -- was built, bit-exact, and lost: 438 Msym/s per worker against 323 in the scalar
    // loop of four on noise planes, but every stream in it advances at 27 Msym/s against 81, and with the fields in flight
    // bounded by memory it is the slowest stream of a field that sets the rate: 4.4 against 11.8 GB/s.  Removed in round 5;
    // profiles/r03/h_rc_any_epyc9575f.txt, i_bench_*)
    // The encoder's vector loop takes every plane (WR_VEC_ENCODE=0: scalar loops of three instead): 16 planes at 1.0-1.3
    // Gsym/s per worker in the pipeline against 0.56 in the scalar loops.  Per stream it advances at 75-150 Msym/s against
    // ~190, so the sessions are kept as many as leave a field's encode no longer than its decode (3/8 of the workers:
    // 16 workers, 24 fields in flight: 6 sessions 11.3 GB/s, 4 sessions 9.4, scalar 9.4-9.7; 4 + 5, 4 + 6, 5 + 5 encoder +
    // decoder sessions instead of 6 + 5: the same within the run-to-run spread, profiles/r03/be_sessions_*).
    const bool vec_enc_ = !(getenv("WR_VEC_ENCODE") && !atoi(getenv("WR_VEC_ENCODE")));
    int venc_sessions_ = 0;
    int venc_sessions_max() const  // call with mu_ held
    {
        const int w = (int)workers_.size();
        return w < 3 ? 1 : (w * 3 + 4) / 8;
    }
    const bool vec_ok_ = vec_available();
    int vec_sessions_ = 0;
    int vec_sessions_max() const  // call with mu_ held
    {
        const int w = (int)workers_.size();
        return w < 4 ? 1 : (w * 5 + 8) / 16;  // 16 workers: 5 sessions (with the vector encoder: 11.3 GB/s; 3 sessions: 10.1)
    }
    std::vector<std::thread> workers_;
    bool stop_ = false;
    int dec_streams_ = kMaxDecStreams;
};

}  // namespace

void pool_configure(int nthreads, int dec_streams) { Pool::get().resize(nthreads, dec_streams); }
void pool_test_steal_idle_min(int workers) { Pool::get().set_steal_idle_test(workers); }
int pool_threads() { return Pool::get().threads(); }
double pool_idle_seconds() { return Pool::get().idle_seconds(); }
double pool_queue_seconds() { return Pool::get().queue_seconds(); }
void pool_queued_decode(int* scalar_jobs, int* vector_jobs) { Pool::get().queued_decode(scalar_jobs, vector_jobs); }
unsigned long pool_streams_moved() { return Pool::get().streams_moved(); }
void pool_loop_stats(double seconds[kLoopKinds], double blocks[kLoopKinds]) { Pool::get().loop_stats(seconds, blocks); }
bool pool_submit(PlaneJob* jobs, int count, JobBatch* batch) { return Pool::get().submit(jobs, count, batch); }
void pool_wait(JobBatch* batch)
{
    std::unique_lock<std::mutex> lk(batch->mu);
    batch->cv.wait(lk, [&] { return batch->remaining == 0; });
}

}  // namespace wrrc
