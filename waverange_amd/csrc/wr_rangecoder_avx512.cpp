// wr_rangecoder_avx512.cpp -- see wr_rangecoder_vec.h.  Compiled with -mavx512f -mavx512bw -mavx512dq -mavx512vl;
// entered only when vec_available().
#include "wr_rangecoder_vec.h"

#include <immintrin.h>
#include <stdlib.h>

namespace wrrc {

namespace {

constexpr uint32_t kTop = 0x80000000u;     // rangecod.c:121
constexpr uint32_t kBottom = 0x00800000u;  // rangecod.c:129
constexpr int kExtra = 7;                  // rangecod.c:128
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

bool vec_available()
{
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
                           __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl") &&
                           !(getenv("WR_NO_AVX512") && atoi(getenv("WR_NO_AVX512")));
    return ok;
}

void vec_decode_block(VecBlock* b, VecOther other)
{
    const __mmask16 act = (__mmask16)b->active;
    const __m512i vbottom = _mm512_set1_epi32((int)kBottom);
    // inactive lanes idle on a state that never renormalises and always "hits" candidate 0
    __m512i low = _mm512_maskz_loadu_epi32(act, b->low);
    __m512i range = _mm512_mask_loadu_epi32(_mm512_set1_epi32((int)kTop), act, b->range);
    __m512i lt[kVecCand], sy[kVecCand], sym[kVecCand];
    __mmask16 top[kVecCand];
    for (int e = 0; e < kVecCand; e++) {
        lt[e] = _mm512_maskz_loadu_epi32(act, b->lt[e]);
        sy[e] = e == 0 ? _mm512_mask_loadu_epi32(_mm512_set1_epi32((int)kBlockSyms), act, b->sy[e]) : _mm512_maskz_loadu_epi32(act, b->sy[e]);
        sym[e] = _mm512_maskz_loadu_epi32(act, b->sym[e]);
        top[e] = _mm512_mask_cmpneq_epu32_mask(act, _mm512_loadu_si512(b->is_top[e]), _mm512_setzero_si512());
    }
    // floor(x / 60000) = ((x >> 5) * 146601551) >> 38 for every 32-bit x (checked exhaustively over x >> 5 < 2^27)
    const __m512i magic = _mm512_set1_epi64(146601551);
    const __m512i hi32 = _mm512_set1_epi64((long long)0xFFFFFFFF00000000ull);

    // Byte feed.  The stream enters `low` as a bit string that starts 7 bits into the byte held back
    // (rangecod.c:297-299): the byte a lane shifts in at pointer q is ((q[-1] << 8 | q[0]) >> 1) & 0xff.  Every lane
    // keeps a window of the next four such bytes (top byte first), the count of those still unused, and the
    // window after that, already loaded: when a window runs empty the next one takes its place with one vector
    // move, and the scalar reload of the window after it (load, byte swap, general register -> vector lane:
    // ~25 cycles) is not needed before the lane has consumed four more bytes -- it stays off the loop's
    // dependency chain, which the whole-register update of a just-in-time reload would sit on.
    const uint8_t* pw[kVecLanes];  // stream position the lane's current window was loaded at
    uint8_t* d[kVecLanes];
    alignas(64) uint32_t w0[kVecLanes], w1[kVecLanes];
    auto window_at = [](const uint8_t* q) -> uint32_t {
        uint64_t v;
        __builtin_memcpy(&v, q - 1, 8);
        return (uint32_t)(__builtin_bswap64(v) >> 25);
    };
    for (int j = 0; j < kVecLanes; j++) {
        pw[j] = b->ptr[j]; d[j] = b->dst[j]; w0[j] = w1[j] = 0;
        if (act >> j & 1) { w0[j] = window_at(pw[j]); w1[j] = window_at(pw[j] + 4); }
    }
    __m512i win = _mm512_load_si512(w0), nxt = _mm512_load_si512(w1);
    __m512i cnt = _mm512_set1_epi32(4);
    const __m512i one = _mm512_set1_epi32(1), four = _mm512_set1_epi32(4);
    alignas(64) uint32_t tl[kVecLanes], tr[kVecLanes], th[kVecLanes];
    __m128i rows[16];

    for (uint32_t i = 0; i < kBlockSyms; i++) {
        // ---- renormalise (rangecod.c:294-302): lanes with range <= Bottom shift one byte in
        __mmask16 sh = _mm512_cmple_epu32_mask(range, vbottom);
        for (;;) {
            low = _mm512_mask_or_epi32(low, sh, _mm512_slli_epi32(low, 8), _mm512_srli_epi32(win, 24));
            range = _mm512_mask_slli_epi32(range, sh, range, 8);
            win = _mm512_mask_slli_epi32(win, sh, win, 8);
            cnt = _mm512_mask_sub_epi32(cnt, sh, cnt, one);
            const __mmask16 dry = _mm512_cmpeq_epu32_mask(cnt, _mm512_setzero_si512());
            if (dry) {
                win = _mm512_mask_mov_epi32(win, dry, nxt);
                cnt = _mm512_mask_mov_epi32(cnt, dry, four);
                unsigned m = dry & act;
                while (m) {
                    const int j = __builtin_ctz(m);
                    m &= m - 1;
                    pw[j] += 4;
                    nxt = _mm512_mask_set1_epi32(nxt, (__mmask16)(1u << j), (int)window_at(pw[j] + 4));
                }
            }
            sh = _mm512_cmple_epu32_mask(range, vbottom);  // a second byte: symbol probability < 1/256, rare
            if (__builtin_expect(sh == 0, 1)) break;
        }
        // ---- help = range / 60000 (rangecod.c:312)
        const __m512i n5 = _mm512_srli_epi32(range, 5);
        const __m512i ev = _mm512_srli_epi64(_mm512_mul_epu32(n5, magic), 38);
        const __m512i od = _mm512_and_si512(_mm512_srli_epi64(_mm512_mul_epu32(_mm512_srli_epi64(n5, 32), magic), 6), hi32);
        const __m512i help = _mm512_or_si512(ev, od);
        // ---- which candidate: low - help*lt < width of its interval (rangecod.c:313-319, 339-351).  The intervals
        // are disjoint, so at most one test holds; an unused entry has width 0.
        __m512i a[kVecCand], w[kVecCand];
        __mmask16 in[kVecCand];
        for (int e = 0; e < kVecCand; e++) {
            a[e] = _mm512_mullo_epi32(help, lt[e]);
            w[e] = _mm512_mask_sub_epi32(_mm512_mullo_epi32(help, sy[e]), top[e], range, a[e]);
            in[e] = _mm512_cmplt_epu32_mask(_mm512_sub_epi32(low, a[e]), w[e]);
        }
        __m512i c = sym[0], sa = a[0], nrange = w[0];
        for (int e = 1; e < kVecCand; e++) {
            c = _mm512_mask_mov_epi32(c, in[e], sym[e]);
            sa = _mm512_mask_mov_epi32(sa, in[e], a[e]);
            nrange = _mm512_mask_mov_epi32(nrange, in[e], w[e]);
        }
        __m512i nlow = _mm512_sub_epi32(low, sa);
        const __mmask16 in0 = in[0], in1 = (__mmask16)(in[1] | in[2] | in[3]);
        const __mmask16 miss = act & ~(in0 | in1);
        if (__builtin_expect(miss != 0, 0)) {  // some other symbol: scalar look-up path for those lanes
            _mm512_store_si512(tl, low); _mm512_store_si512(tr, range); _mm512_store_si512(th, help);
            unsigned m = miss;
            do {
                const int j = __builtin_ctz(m);
                m &= m - 1;
                const uint32_t cj = other(b->model[j], &tl[j], &tr[j], th[j]);
                const __mmask16 one = (__mmask16)(1u << j);
                nlow = _mm512_mask_set1_epi32(nlow, one, (int)tl[j]);
                nrange = _mm512_mask_set1_epi32(nrange, one, (int)tr[j]);
                c = _mm512_mask_set1_epi32(c, one, (int)cj);
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
    _mm512_mask_storeu_epi32(b->low, act, low);
    _mm512_mask_storeu_epi32(b->range, act, range);
    alignas(64) uint32_t left[kVecLanes];
    _mm512_store_si512(left, cnt);
    for (int j = 0; j < kVecLanes; j++)
        if (act >> j & 1) b->ptr[j] = pw[j] + (4 - left[j]);
}

void vec_encode_block(VecEncBlock* b)
{
    const __mmask16 act = (__mmask16)b->active;
    const __m512i vbottom = _mm512_set1_epi32((int)kBottom), vtopm1 = _mm512_set1_epi32((int)(kTop - 1));
    __m512i low = _mm512_maskz_loadu_epi32(act, b->low);
    __m512i range = _mm512_mask_loadu_epi32(_mm512_set1_epi32((int)kTop), act, b->range);
    const __m512i top = _mm512_loadu_si512(b->top);
    __m512i cand[kVecCand], clt[kVecCand], csy[kVecCand];
    for (int e = 0; e < kVecCand; e++) {
        cand[e] = _mm512_mask_loadu_epi32(_mm512_set1_epi32(0x100), act, b->cand[e]);
        clt[e] = _mm512_maskz_loadu_epi32(act, b->lt[e]);
        csy[e] = _mm512_maskz_loadu_epi32(act, b->sy[e]);
    }
    const __m512i magic = _mm512_set1_epi64(146601551);  // see vec_decode_block
    const __m512i hi32 = _mm512_set1_epi64((long long)0xFFFFFFFF00000000ull);

    uint8_t* out[kVecLanes];
    size_t pos[kVecLanes];
    const uint8_t* in[kVecLanes];
    static const uint8_t zeros[16] = {0};
    uint8_t dummy[8];  // idle lanes store their byte here (and never advance)
    for (int j = 0; j < kVecLanes; j++) {
        const bool on = act >> j & 1;
        out[j] = on ? b->out[j] : dummy; pos[j] = on ? b->pos[j] : 0; in[j] = on ? b->sym[j] : zeros;
    }
    alignas(16) uint8_t bytes[16];
    alignas(64) uint32_t tc[kVecLanes];
    __m128i rows[16];

    for (uint32_t i = 0; i < kBlockSyms; i++) {
        // ---- symbols in: every 16 steps, 16 bytes of each plane, transposed to one row per step
        if ((i & 15) == 0) {
            for (int j = 0; j < kVecLanes; j++) rows[j] = _mm_loadu_si128(reinterpret_cast<const __m128i*>((act >> j & 1) ? in[j] + i : zeros));
            transpose16x16(rows);
        }
        const __m512i c = _mm512_cvtepu8_epi32(rows[i & 15]);
        // ---- {lt, sy} of the symbol: one of the lane's candidates, else the lane's table
        const __mmask16 k1 = _mm512_cmpeq_epu32_mask(c, cand[1]), k2 = _mm512_cmpeq_epu32_mask(c, cand[2]), k3 = _mm512_cmpeq_epu32_mask(c, cand[3]);
        const __mmask16 k0 = _mm512_cmpeq_epu32_mask(c, cand[0]);
        __m512i lt = _mm512_mask_mov_epi32(_mm512_mask_mov_epi32(_mm512_mask_mov_epi32(clt[0], k1, clt[1]), k2, clt[2]), k3, clt[3]);
        __m512i sy = _mm512_mask_mov_epi32(_mm512_mask_mov_epi32(_mm512_mask_mov_epi32(csy[0], k1, csy[1]), k2, csy[2]), k3, csy[3]);
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
        // ---- renormalise (rangecod.c:182-207): lanes with range <= Bottom put a byte out.  Branch-free for the
        // usual single byte: every lane stores its candidate byte at its write position and advances by 0 or 1
        // (as the scalar loop does), so that no data-dependent branch or scalar loop count sits between one
        // step's range and the next; a pending carry and a second byte (symbol probability < 1/256) are rare.
        __mmask16 sh = _mm512_cmple_epu32_mask(range, vbottom) & act;
        for (;;) {
            _mm_store_si128(reinterpret_cast<__m128i*>(bytes), _mm512_cvtepi32_epi8(_mm512_srli_epi32(low, 23)));
            unsigned carry = _mm512_mask_test_epi32_mask(sh, low, _mm512_set1_epi32((int)kTop));
            while (__builtin_expect(carry != 0, 0)) {  // "carry now", rangecod.c:191-195
                const int j = __builtin_ctz(carry);
                carry &= carry - 1;
                size_t p = pos[j] - 1;
                while (++out[j][p] == 0) p--;
            }
            const unsigned m = sh;
#pragma GCC unroll 16
            for (int j = 0; j < kVecLanes; j++) {
                out[j][pos[j]] = bytes[j];
                pos[j] += m >> j & 1;
            }
            low = _mm512_mask_and_epi32(low, sh, _mm512_slli_epi32(low, 8), vtopm1);
            range = _mm512_mask_slli_epi32(range, sh, range, 8);
            sh = _mm512_cmple_epu32_mask(range, vbottom) & act;
            if (__builtin_expect(sh == 0, 1)) break;
        }
        // ---- r = range / 60000; low += r * lt; range = r * sy, or what is left for the largest symbol (rangecod.c:217-229)
        const __m512i n5 = _mm512_srli_epi32(range, 5);
        const __m512i ev = _mm512_srli_epi64(_mm512_mul_epu32(n5, magic), 38);
        const __m512i od = _mm512_and_si512(_mm512_srli_epi64(_mm512_mul_epu32(_mm512_srli_epi64(n5, 32), magic), 6), hi32);
        const __m512i r = _mm512_or_si512(ev, od);
        const __m512i t = _mm512_mullo_epi32(r, lt);
        low = _mm512_add_epi32(low, t);
        const __mmask16 is_top = _mm512_cmpeq_epu32_mask(c, top);
        range = _mm512_mask_sub_epi32(_mm512_mullo_epi32(r, sy), is_top, range, t);
    }
    _mm512_mask_storeu_epi32(b->low, act, low);
    _mm512_mask_storeu_epi32(b->range, act, range);
    for (int j = 0; j < kVecLanes; j++) if (act >> j & 1) b->pos[j] = pos[j];
}

}  // namespace wrrc
