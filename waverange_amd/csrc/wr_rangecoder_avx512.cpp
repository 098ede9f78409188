// wr_rangecoder_avx512.cpp -- see wr_rangecoder_vec.h.  Compiled with -mavx512f -mavx512bw -mavx512dq -mavx512vl;
// entered only when vec_available() (wr_rangecoder.cpp: the dispatcher is built without those flags).  Nothing but
// the block kernels lives here.
#include "wr_rangecoder_vec.h"

#include <immintrin.h>
#include <stdlib.h>
#include <string.h>

namespace wrrc {

namespace {

constexpr uint32_t kTop = 0x80000000u;     // rangecod.c:121
constexpr uint32_t kBottom = 0x00800000u;  // rangecod.c:129
constexpr uint32_t kBlockSyms = 60000;     // defs.h:36

// transpose of a 16 x 16 byte matrix held in 16 xmm registers (rows in, columns out)
inline void transpose16x16(__m128i r[16])
{
    __m128i t[16];
    for (int i = 0; i < 8; i++) { t[2 * i] = _mm_unpacklo_epi8(r[i], r[i + 8]); t[2 * i + 1] = _mm_unpackhi_epi8(r[i], r[i + 8]); }
    for (int i = 0; i < 8; i++) { r[2 * i] = _mm_unpacklo_epi8(t[i], t[i + 8]); r[2 * i + 1] = _mm_unpackhi_epi8(t[i], t[i + 8]); }
    for (int i = 0; i < 8; i++) { t[2 * i] = _mm_unpacklo_epi8(r[i], r[i + 8]); t[2 * i + 1] = _mm_unpackhi_epi8(r[i], r[i + 8]); }
    for (int i = 0; i < 8; i++) { r[2 * i] = _mm_unpacklo_epi8(t[i], t[i + 8]); r[2 * i + 1] = _mm_unpackhi_epi8(t[i], t[i + 8]); }
}

}  // namespace

namespace {

// The lanes of one VecBlock: state and one symbol step of all of them (rangecod.c:294-351 per lane).
// floor(x / 60000) in every 32-bit lane = ((x >> 5) * 146601551) >> 38 (checked exhaustively over x >> 5 < 2^27).  Even
// lanes: the product's bits 38.. land in the low half; odd lanes: x >> 37 of the 64-bit lane IS the odd lane's x >> 5, and
// the product's bits 38.. move to the high half by a shift of 6 and a blend.
inline __m512i div60000(__m512i x)
{
    const __m512i magic = _mm512_set1_epi64(146601551);
    const __m512i ev = _mm512_srli_epi64(_mm512_mul_epu32(_mm512_srli_epi32(x, 5), magic), 38);
    const __m512i od = _mm512_srli_epi64(_mm512_mul_epu32(_mm512_srli_epi64(x, 37), magic), 6);
    return _mm512_mask_mov_epi32(ev, (__mmask16)0xAAAA, od);
}

// what a step indexes by lane number: kept apart from the vector state so that the latter lives in registers
struct DecMem {
    VecBlock* b;
    VecOther other;
    const uint8_t* pw[kVecLanes];  // stream position the lane's current window was loaded at
    uint8_t* d[kVecLanes];
    __m128i rows[16];
};

struct DecLanes {
    __mmask16 act;
    __m512i low, range, win, win2, used;
    __m512i lt[kVecCand], sy[kVecCand], sym[kVecCand];
    __mmask16 top[kVecCand];
    // Byte feed.  The stream enters `low` as a bit string that starts 7 bits into the byte held back
    // (rangecod.c:297-299): the byte a lane shifts in at pointer q is ((q[-1] << 8 | q[0]) >> 1) & 0xff.  Every lane
    // keeps the next eight such bytes in two registers (win: the next four, top byte first; win2: the four after them)
    // that shift as one when the lane takes a byte, and counts the bytes taken.  A step takes at most one byte per lane
    // without looking (a second one in the same step -- a symbol of probability < 1/256 -- refills first), so looking
    // every fourth step is enough: lanes that have taken four or more get eight fresh bytes (scalar load, byte swap,
    // general register -> vector lane, ~25 cycles, off the loop's dependency chain).  No per-step counter test, no
    // branch per step: the test and its mask sat on the chain of every step (tools/native/vstep_probe.cpp: 2.6 of 10 ns).

    static inline uint32_t window_at(const uint8_t* q)
    {
        uint64_t v;
        __builtin_memcpy(&v, q - 1, 8);
        return (uint32_t)(__builtin_bswap64(v) >> 25);
    }
    __attribute__((always_inline)) inline void init(DecMem& mem, VecBlock* b, VecOther oth)
    {
        mem.b = b; mem.other = oth;
        const uint8_t** pw = mem.pw;
        uint8_t** d = mem.d;
        act = (__mmask16)b->active;
        // inactive lanes idle on a state that never renormalises: all four candidates {lt 0, sy 60000}
        low = _mm512_maskz_loadu_epi32(act, b->low);
        range = _mm512_mask_loadu_epi32(_mm512_set1_epi32((int)kTop), act, b->range);
        for (int e = 0; e < kVecCand; e++) {
            lt[e] = _mm512_maskz_loadu_epi32(act, b->lt[e]);
            sy[e] = _mm512_mask_loadu_epi32(_mm512_set1_epi32((int)kBlockSyms), act, b->sy[e]);
            sym[e] = _mm512_maskz_loadu_epi32(act, b->sym[e]);
            top[e] = _mm512_mask_cmpneq_epu32_mask(act, _mm512_loadu_si512(b->is_top[e]), _mm512_setzero_si512());
        }
        alignas(64) uint32_t w0[kVecLanes], w1[kVecLanes];
        for (int j = 0; j < kVecLanes; j++) {
            pw[j] = b->ptr[j]; d[j] = b->dst[j]; w0[j] = w1[j] = 0;
            if (act >> j & 1) { w0[j] = window_at(pw[j]); w1[j] = window_at(pw[j] + 4); }
        }
        win = _mm512_load_si512(w0); win2 = _mm512_load_si512(w1);
        used = _mm512_setzero_si512();
    }
    // eight fresh bytes for the lanes in m
    __attribute__((always_inline)) inline void refill(DecMem& mem, unsigned m)
    {
        const uint8_t** pw = mem.pw;
        alignas(64) uint32_t u[kVecLanes];
        _mm512_store_si512(u, used);
        used = _mm512_maskz_mov_epi32((__mmask16)~m, used);
        m &= act;
        while (m) {
            const int j = __builtin_ctz(m);
            m &= m - 1;
            pw[j] += u[j];
            const __mmask16 bit = (__mmask16)(1u << j);
            win = _mm512_mask_set1_epi32(win, bit, (int)window_at(pw[j]));
            win2 = _mm512_mask_set1_epi32(win2, bit, (int)window_at(pw[j] + 4));
        }
    }
    __attribute__((always_inline)) inline void step(DecMem& mem, uint32_t i)
    {
        VecBlock* const b = mem.b;
        uint8_t** d = mem.d;
        __m128i* rows = mem.rows;
        const __m512i vbottom = _mm512_set1_epi32((int)kBottom);
        const __m512i one = _mm512_set1_epi32(1);
        // ---- renormalise (rangecod.c:294-302): lanes with range <= Bottom shift one byte in; help = range / 60000
        // (rangecod.c:312) of the range after it.  The quotient is worked out for the range as it is AND for the range
        // shifted, beside the compare, and the compare's mask picks one: compare -> mask -> shift -> divide in a row was a
        // third of the step's dependency chain.
        const __mmask16 sh = _mm512_cmple_epu32_mask(range, vbottom);
        const __m512i h_as_is = div60000(range), h_shifted = div60000(_mm512_slli_epi32(range, 8));
        low = _mm512_mask_or_epi32(low, sh, _mm512_slli_epi32(low, 8), _mm512_srli_epi32(win, 24));
        range = _mm512_mask_slli_epi32(range, sh, range, 8);
        win = _mm512_mask_or_epi32(win, sh, _mm512_slli_epi32(win, 8), _mm512_srli_epi32(win2, 24));
        win2 = _mm512_mask_slli_epi32(win2, sh, win2, 8);
        used = _mm512_mask_add_epi32(used, sh, used, one);
        __m512i help = _mm512_mask_mov_epi32(h_as_is, sh, h_shifted);
        __mmask16 again = _mm512_cmple_epu32_mask(range, vbottom);  // a second byte: symbol probability < 1/256, rare
        if (__builtin_expect(!_kortestz_mask16_u8(again, again), 0)) {
            do {
                refill(mem, 0xffffu);
                low = _mm512_mask_or_epi32(low, again, _mm512_slli_epi32(low, 8), _mm512_srli_epi32(win, 24));
                range = _mm512_mask_slli_epi32(range, again, range, 8);
                win = _mm512_mask_or_epi32(win, again, _mm512_slli_epi32(win, 8), _mm512_srli_epi32(win2, 24));
                win2 = _mm512_mask_slli_epi32(win2, again, win2, 8);
                used = _mm512_mask_add_epi32(used, again, used, one);
                again = _mm512_cmple_epu32_mask(range, vbottom);
            } while (!_kortestz_mask16_u8(again, again));
            help = div60000(range);
        }
        if ((i & 3) == 3) {
            const __mmask16 need = _mm512_cmpge_epu32_mask(used, _mm512_set1_epi32(4));
            if (need) refill(mem, need);
        }
        // ---- which candidate (rangecod.c:313-319, 339-351).  The lane's candidates come sorted by interval start (a lane
        // with fewer than four repeats its last one), so the symbol is the LAST candidate whose start help * lt is at or
        // below `low` -- three compares whose masks nest, and a two-level tree of selects -- provided `low` lies inside
        // that candidate's interval; whether it does (else: a symbol outside the candidates, below the first one or in a
        // gap) shows in low - start < width afterwards, off the chain that the next step waits for.
        static_assert(kVecCand == 4, "select tree");
        __m512i a[kVecCand], w[kVecCand];
        for (int e = 0; e < kVecCand; e++) {
            a[e] = _mm512_mullo_epi32(help, lt[e]);
            w[e] = _mm512_mask_sub_epi32(_mm512_mullo_epi32(help, sy[e]), top[e], range, a[e]);
        }
        const __mmask16 ge1 = _mm512_cmpge_epu32_mask(low, a[1]), ge2 = _mm512_cmpge_epu32_mask(low, a[2]), ge3 = _mm512_cmpge_epu32_mask(low, a[3]);
        __m512i c = _mm512_mask_mov_epi32(_mm512_mask_mov_epi32(sym[0], ge1, sym[1]), ge2, _mm512_mask_mov_epi32(sym[2], ge3, sym[3]));
        const __m512i sa = _mm512_mask_mov_epi32(_mm512_mask_mov_epi32(a[0], ge1, a[1]), ge2, _mm512_mask_mov_epi32(a[2], ge3, a[3]));
        __m512i nrange = _mm512_mask_mov_epi32(_mm512_mask_mov_epi32(w[0], ge1, w[1]), ge2, _mm512_mask_mov_epi32(w[2], ge3, w[3]));
        __m512i nlow = _mm512_sub_epi32(low, sa);
        const __mmask16 miss = _mm512_mask_cmpge_epu32_mask(act, nlow, nrange);
        if (__builtin_expect(!_kortestz_mask16_u8(miss, miss), 0)) {  // some other symbol: scalar look-up path for those lanes
            alignas(64) uint32_t tl[kVecLanes], tr[kVecLanes], th[kVecLanes];
            _mm512_store_si512(tl, low); _mm512_store_si512(tr, range); _mm512_store_si512(th, help);
            unsigned m = miss;
            do {
                const int j = __builtin_ctz(m);
                m &= m - 1;
                const uint32_t cj = mem.other(b->model[j], &tl[j], &tr[j], th[j]);
                const __mmask16 bit = (__mmask16)(1u << j);
                nlow = _mm512_mask_set1_epi32(nlow, bit, (int)tl[j]);
                nrange = _mm512_mask_set1_epi32(nrange, bit, (int)tr[j]);
                c = _mm512_mask_set1_epi32(c, bit, (int)cj);
            } while (m);
        }
        low = nlow; range = nrange;
        // ---- symbols out: 16 steps are collected and transposed into 16 bytes per lane
        rows[i & 15] = _mm512_cvtepi32_epi8(c);
        if ((i & 15) == 15) {
            transpose16x16(rows);
            unsigned m = act;
            while (m) {
                const int j = __builtin_ctz(m);
                m &= m - 1;
                _mm_storeu_si128(reinterpret_cast<__m128i*>(d[j] + (i - 15)), rows[j]);
            }
        }
    }
    __attribute__((always_inline)) inline void fini(DecMem& mem)
    {
        VecBlock* const b = mem.b;
        const uint8_t** pw = mem.pw;
        _mm512_mask_storeu_epi32(b->low, act, low);
        _mm512_mask_storeu_epi32(b->range, act, range);
        alignas(64) uint32_t left[kVecLanes];
        _mm512_store_si512(left, used);
        for (int j = 0; j < kVecLanes; j++)
            if (act >> j & 1) b->ptr[j] = pw[j] + left[j];
    }
};

}  // namespace

void vec_decode_block(VecBlock* b, VecOther other)
{
    DecLanes s;
    DecMem m;
    s.init(m, b, other);
    for (uint32_t i = 0; i < kBlockSyms; i++) s.step(m, i);
    s.fini(m);
}

// MODE 0: {lt, sy} by comparing with the lane's candidates; 2: a scalar load per lane from the lanes' packed tables, returned
// to a vector by inserts (the symbols then come straight from the lanes' streams, no transposes).  (1 was two 8-lane gathers
// from the lanes' tables: AMD's gathers are microcoded, EPYC 9575F, 16 noise planes, one thread: 0.93 against 1.09 Gsym/s,
// profiles/r03; removed.)
template <bool ALWAYS, int MODE>
static void vec_encode_block_t(VecEncBlock* b)
{
    static_assert(MODE == 0 || MODE == 2, "candidate compares or per-lane look-ups");
    __mmask16 act = (__mmask16)b->active;
    const __m512i vbottom = _mm512_set1_epi32((int)kBottom), vtopm1 = _mm512_set1_epi32((int)(kTop - 1));
    __m512i low = _mm512_maskz_loadu_epi32(act, b->low);
    __m512i range = _mm512_mask_loadu_epi32(_mm512_set1_epi32((int)kTop), act, b->range);
    const __m512i top = _mm512_loadu_si512(b->top);
    __m512i cand[kVecCand], clt[kVecCand], csy[kVecCand];
    for (int e = 0; e < kVecCand; e++) {
        cand[e] = _mm512_mask_loadu_epi32(_mm512_set1_epi32(0x100), act, b->cand[e]);
        clt[e] = _mm512_maskz_loadu_epi32(act, b->lt[e]);
        // (an idle lane codes its candidate 0 -- no symbol matches it, the miss test leaves idle lanes out -- with {lt 0,
        // sy 60000} for ever: its range stays where it is, above Bottom, so no mask of the loop needs an "and active")
        csy[e] = e == 0 ? _mm512_mask_loadu_epi32(_mm512_set1_epi32((int)kBlockSyms), act, b->sy[e]) : _mm512_maskz_loadu_epi32(act, b->sy[e]);
    }
    const __m512i one = _mm512_set1_epi32(1), four = _mm512_set1_epi32(4), v255 = _mm512_set1_epi32(0xff);
    // byte swap inside every 32-bit lane (vpshufb works per 128-bit lane; the pattern repeats)
    const __m512i bswap = _mm512_broadcast_i32x4(_mm_set_epi8(12, 13, 14, 15, 8, 9, 10, 11, 4, 5, 6, 7, 0, 1, 2, 3));

    // Bytes leave the way rangecod.c:182-207 lets them: the last byte shifted out of `low` is held back (`held`), together
    // with the number of 0xff bytes that followed it (`ffs`), until the next byte shows whether a carry reaches them; so
    // nothing that has been written is ever touched again.  The scalar code around this loop writes every byte at once
    // and walks back on a carry; on entry the lane's last byte (and the 0xff bytes behind it, if any) are therefore
    // taken back from the stream, and on exit they are written out again.
    // Final bytes collect in the lane's 32-bit `pend` (count in `cnt`); when a lane holds four, all lanes store theirs
    // with one unaligned 4-byte store each (what lies beyond a lane's `cnt` is overwritten by its next store).
    const uint8_t* in[kVecLanes];
    uint32_t step_of[kVecLanes];  // MODE 2: index mask, all ones for a lane with a stream, 0 for an idle one (it reads zeros[0] in every step)
    static const uint8_t zeros[16] = {0};
    alignas(64) uint8_t dummy[64];  // idle lanes store here (and never advance)
    alignas(64) uint64_t addr[kVecLanes];  // next write address of every lane
    alignas(64) uint32_t hbuf[kVecLanes], fbuf[kVecLanes], vbuf[kVecLanes], tc[kVecLanes];
    for (int j = 0; j < kVecLanes; j++) {
        const bool on = act >> j & 1;
        hbuf[j] = fbuf[j] = 0;
        addr[j] = (uint64_t)(uintptr_t)dummy;
        in[j] = zeros;
        step_of[j] = on ? ~0u : 0u;
        // an idle lane codes symbol 0 with {lt 0, sy 60000} for ever: its range stays where it is, above Bottom
        if (MODE == 2 && !on) b->packed[j * 256] = kBlockSyms << 16;
        if (!on) continue;
        size_t p = b->pos[j] - 1;
        while (p > 0 && b->out[j][p] == 0xff) { fbuf[j]++; p--; }
        hbuf[j] = b->out[j][p];
        addr[j] = (uint64_t)(uintptr_t)(b->out[j] + p);
        in[j] = b->sym[j];
    }
    __m512i alo = _mm512_load_si512(addr), ahi = _mm512_load_si512(addr + 8);
    __m512i held = _mm512_load_si512(hbuf), ffs = _mm512_load_si512(fbuf);
    __m512i pend = _mm512_setzero_si512(), cnt = _mm512_setzero_si512();
    __m128i rows[16];

    // Lanes whose range has gone to ZERO: their table counted the symbol they just coded zero times -- it is not the table
    // of these symbols (VecEncBlock::failed).  Such a lane would shift bytes out for ever; it is retired instead: from here
    // on it is an idle lane (symbol 0 with {lt 0, sy 60000} for ever, stores into `dummy`), and the caller gives its stream up.
    auto retire_lanes = [&](unsigned m) {
        const __mmask16 k = (__mmask16)m;
        b->failed |= m;
        act = (__mmask16)(act & ~m);
        range = _mm512_mask_mov_epi32(range, k, _mm512_set1_epi32((int)kTop));
        low = _mm512_maskz_mov_epi32((__mmask16)~k, low);
        held = _mm512_maskz_mov_epi32((__mmask16)~k, held);
        ffs = _mm512_maskz_mov_epi32((__mmask16)~k, ffs);
        pend = _mm512_maskz_mov_epi32((__mmask16)~k, pend);
        cnt = _mm512_maskz_mov_epi32((__mmask16)~k, cnt);
        const __m512i vdummy = _mm512_set1_epi64((long long)(uintptr_t)dummy);
        alo = _mm512_mask_mov_epi64(alo, (__mmask8)(m & 0xff), vdummy);
        ahi = _mm512_mask_mov_epi64(ahi, (__mmask8)(m >> 8), vdummy);
        for (int e = 0; e < kVecCand; e++) {
            cand[e] = _mm512_mask_mov_epi32(cand[e], k, _mm512_set1_epi32(0x100));
            clt[e] = _mm512_maskz_mov_epi32((__mmask16)~k, clt[e]);
            csy[e] = e == 0 ? _mm512_mask_mov_epi32(csy[e], k, _mm512_set1_epi32((int)kBlockSyms)) : _mm512_maskz_mov_epi32((__mmask16)~k, csy[e]);
        }
        while (m) {
            const int j = __builtin_ctz(m);
            m &= m - 1;
            in[j] = zeros; step_of[j] = 0;
            if (MODE == 2) b->packed[j * 256] = kBlockSyms << 16;
        }
    };

    auto flush_all = [&]() {
        // first byte out in the most significant position, then byte-swapped: it lands at the lowest address
        const __m512i val = _mm512_shuffle_epi8(_mm512_sllv_epi32(pend, _mm512_slli_epi32(_mm512_sub_epi32(four, cnt), 3)), bswap);
        _mm512_store_si512(vbuf, val);
        _mm512_store_si512(addr, alo);
        _mm512_store_si512(addr + 8, ahi);
#pragma GCC unroll 16
        for (int j = 0; j < kVecLanes; j++) memcpy(reinterpret_cast<void*>((uintptr_t)addr[j]), &vbuf[j], 4);  // one unaligned 4-byte store
        alo = _mm512_add_epi64(alo, _mm512_cvtepu32_epi64(_mm512_castsi512_si256(cnt)));
        ahi = _mm512_add_epi64(ahi, _mm512_cvtepu32_epi64(_mm512_extracti64x4_epi64(cnt, 1)));
        pend = _mm512_setzero_si512();
        cnt = _mm512_setzero_si512();
    };
    // lanes that put a byte out while 0xff bytes are held back (one byte in 256 is 0xff): held + carry, then the 0xff
    // bytes (0x00 after a carry), straight to memory
    auto emit_with_ffs = [&](unsigned lanes, __m512i carrybit) {
        flush_all();
        _mm512_store_si512(addr, alo);
        _mm512_store_si512(addr + 8, ahi);
        _mm512_store_si512(hbuf, held);
        _mm512_store_si512(fbuf, ffs);
        _mm512_store_si512(tc, carrybit);
        do {
            const int j = __builtin_ctz(lanes);
            lanes &= lanes - 1;
            uint8_t* p = reinterpret_cast<uint8_t*>((uintptr_t)addr[j]);
            *p++ = (uint8_t)(hbuf[j] + tc[j]);
            for (uint32_t k = 0; k < fbuf[j]; k++) *p++ = tc[j] ? 0x00 : 0xff;
            addr[j] = (uint64_t)(uintptr_t)p;
        } while (lanes);
        alo = _mm512_load_si512(addr);
        ahi = _mm512_load_si512(addr + 8);
    };

    for (uint32_t i = 0; i < kBlockSyms; i++) {
        // ---- symbols in: every 16 steps, 16 bytes of each plane, transposed to one row per step
        __m512i c = _mm512_setzero_si512();
        if (MODE != 2) {
            if ((i & 15) == 0) {
                for (int j = 0; j < kVecLanes; j++) rows[j] = _mm_loadu_si128(reinterpret_cast<const __m128i*>(in[j] + ((act >> j & 1) ? i : 0)));
                transpose16x16(rows);
            }
            c = _mm512_cvtepu8_epi32(rows[i & 15]);
        }
        // ---- {lt, sy} of the symbol: one of the lane's candidates, else the lane's table; or (planes of any
        // statistics) looked up per lane
        __m512i lt, sy;
        __mmask16 is_top_m = 0;
        if (MODE == 2) {
            __m128i x[4];
#pragma GCC unroll 4
            for (int g = 0; g < 4; g++) {
                uint32_t e[4];
#pragma GCC unroll 4
                for (int k = 0; k < 4; k++) {
                    const int j = 4 * g + k;
                    e[k] = b->packed[j * 256 + in[j][i & step_of[j]]];
                }
                x[g] = _mm_insert_epi32(_mm_insert_epi32(_mm_insert_epi32(_mm_cvtsi32_si128((int)e[0]), (int)e[1], 1), (int)e[2], 2), (int)e[3], 3);
            }
            const __m512i ent = _mm512_inserti64x4(_mm512_castsi256_si512(_mm256_inserti128_si256(_mm256_castsi128_si256(x[0]), x[1], 1)),
                                                   _mm256_inserti128_si256(_mm256_castsi128_si256(x[2]), x[3], 1), 1);
            lt = _mm512_and_si512(ent, _mm512_set1_epi32(0xffff));
            sy = _mm512_srli_epi32(ent, 16);
            is_top_m = _mm512_cmpeq_epu32_mask(sy, _mm512_set1_epi32((int)kVecTopMark));
        } else {
            const __mmask16 k1 = _mm512_cmpeq_epu32_mask(c, cand[1]), k2 = _mm512_cmpeq_epu32_mask(c, cand[2]), k3 = _mm512_cmpeq_epu32_mask(c, cand[3]);
            const __mmask16 k0 = _mm512_cmpeq_epu32_mask(c, cand[0]);
            lt = _mm512_mask_mov_epi32(_mm512_mask_mov_epi32(_mm512_mask_mov_epi32(clt[0], k1, clt[1]), k2, clt[2]), k3, clt[3]);
            sy = _mm512_mask_mov_epi32(_mm512_mask_mov_epi32(_mm512_mask_mov_epi32(csy[0], k1, csy[1]), k2, csy[2]), k3, csy[3]);
            unsigned miss = act & ~(k0 | k1 | k2 | k3);
            if (__builtin_expect(miss != 0, 0)) {
                _mm512_store_si512(tc, c);
                do {
                    const int j = __builtin_ctz(miss);
                    miss &= miss - 1;
                    const uint32_t* e = b->tab + ((size_t)j * 256 + tc[j]) * 2;
                    lt = _mm512_mask_set1_epi32(lt, (__mmask16)(1u << j), (int)e[0]);
                    sy = _mm512_mask_set1_epi32(sy, (__mmask16)(1u << j), (int)e[1]);
                } while (miss);
            }
        }
        // ---- renormalise (rangecod.c:182-207): lanes with range <= Bottom shift the top 9 bits out of `low`
        // (ALWAYS: the step's renormalisation runs whether or not a lane shifts -- on planes of a bit per symbol some lane
        // does in most steps, unpredictably, and the branch costs more than the dozen masked instructions)
        // The masks stay in mask registers: `& act` on them went through a general register and back, on the range's
        // chain in every step.  Idle lanes never renormalise (above).
        __mmask16 sh = _mm512_cmple_epu32_mask(range, vbottom);
        // r = range / 60000 of the range after renormalisation is worked out for the range as it is and for the range
        // shifted, beside the compare; the compare's mask picks one (see DecLanes::step).  A second round: divide afterwards.
        const __mmask16 sh0 = sh;
        const __m512i r_as_is = div60000(range), r_shifted = div60000(_mm512_slli_epi32(range, 8));
        int rounds = 0;
        while (ALWAYS || sh) {
            // a range of 1 or more is above Bottom after three bytes: a lane that wants a fourth has a range of zero
            if (__builtin_expect(rounds == 3, 0)) { retire_lanes((unsigned)sh); break; }
            rounds++;
            if (__builtin_expect(_mm512_cmpeq_epu32_mask(cnt, four) != 0, 0)) flush_all();
            const __m512i v9 = _mm512_srli_epi32(low, 23);                 // carry bit (bit 8) + byte
            const __mmask16 isff = _mm512_mask_cmpeq_epu32_mask(sh, v9, v255);
            __mmask16 emit = _kandn_mask16(isff, sh);
            const __m512i carrybit = _mm512_srli_epi32(v9, 8);
            const __mmask16 with_ffs = _mm512_mask_test_epi32_mask(emit, ffs, ffs);
            if (__builtin_expect(!_kortestz_mask16_u8(with_ffs, with_ffs), 0)) {
                emit_with_ffs((unsigned)with_ffs, carrybit);
                emit = _kandn_mask16(with_ffs, emit);
                ffs = _mm512_maskz_mov_epi32(_knot_mask16(with_ffs), ffs);
                held = _mm512_mask_and_epi32(held, with_ffs, v9, v255);
            }
            const __m512i outb = _mm512_and_si512(_mm512_add_epi32(held, carrybit), v255);
            pend = _mm512_mask_or_epi32(pend, emit, _mm512_slli_epi32(pend, 8), outb);
            cnt = _mm512_mask_add_epi32(cnt, emit, cnt, one);
            held = _mm512_mask_and_epi32(held, emit, v9, v255);
            ffs = _mm512_mask_add_epi32(ffs, isff, ffs, one);
            low = _mm512_mask_and_epi32(low, sh, _mm512_slli_epi32(low, 8), vtopm1);
            range = _mm512_mask_slli_epi32(range, sh, range, 8);
            sh = _mm512_cmple_epu32_mask(range, vbottom);  // a second byte: symbol probability < 1/256
            if (ALWAYS && __builtin_expect(_kortestz_mask16_u8(sh, sh), 1)) break;
        }
        // ---- r = range / 60000; low += r * lt; range = r * sy, or what is left for the largest symbol (rangecod.c:217-229)
        __m512i r = _mm512_mask_mov_epi32(r_as_is, sh0, r_shifted);
        if (__builtin_expect(rounds > 1, 0)) r = div60000(range);
        const __m512i t = _mm512_mullo_epi32(r, lt);
        low = _mm512_add_epi32(low, t);
        const __mmask16 is_top = MODE == 2 ? is_top_m : _mm512_cmpeq_epu32_mask(c, top);
        range = _mm512_mask_sub_epi32(_mm512_mullo_epi32(r, sy), is_top, range, t);
    }
    // ---- back to the form the scalar code keeps: everything written, the held byte and its 0xff bytes included
    flush_all();
    _mm512_mask_storeu_epi32(b->low, act, low);
    _mm512_mask_storeu_epi32(b->range, act, range);
    _mm512_store_si512(addr, alo);
    _mm512_store_si512(addr + 8, ahi);
    _mm512_store_si512(hbuf, held);
    _mm512_store_si512(fbuf, ffs);
    for (int j = 0; j < kVecLanes; j++) {
        if (!(act >> j & 1)) continue;  // (idle and retired lanes)
        uint8_t* p = reinterpret_cast<uint8_t*>((uintptr_t)addr[j]);
        *p++ = (uint8_t)hbuf[j];
        for (uint32_t k = 0; k < fbuf[j]; k++) *p++ = 0xff;
        b->pos[j] = (size_t)(p - b->out[j]);
    }
}

void vec_encode_block(VecEncBlock* b)
{
    // how often does some lane shift a byte out?  A lane whose most probable symbol holds less than 97 % of the block
    // puts out more than a byte per 40 symbols.
    int busy = 0;
    for (int j = 0; j < kVecLanes; j++) {
        if (!(b->active >> j & 1)) continue;
        uint32_t best = 0;
        for (int e = 0; e < kVecCand; e++) if (b->sy[e][j] > best) best = b->sy[e][j];
        if ((uint64_t)best * 100 < (uint64_t)kBlockSyms * 97) busy++;
    }
    b->failed = 0;
    if (b->gather) vec_encode_block_t<true, 2>(b);
    else if (busy >= 2) vec_encode_block_t<true, 0>(b);
    else vec_encode_block_t<false, 0>(b);
}

}  // namespace wrrc
