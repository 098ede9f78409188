// vstep_probe.cpp -- where the cycles of one step of the 16-lane decoder loop go (wr_rangecoder_avx512.cpp, DecLanes::step),
// on the CPU it runs on: the step's dataflow on synthetic lane state, with parts taken out.  Timing tool, decodes nothing.
//   g++ -O3 -mavx512f -mavx512bw -mavx512dq -mavx512vl tools/native/vstep_probe.cpp -o vstep_probe && ./vstep_probe
#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <chrono>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

constexpr uint32_t kBottom = 0x00800000u, kTop = 0x80000000u;
constexpr int kSteps = 60000 * 200;

static inline __m512i div60000(__m512i x)
{
    const __m512i magic = _mm512_set1_epi64(146601551);
    const __m512i ev = _mm512_srli_epi64(_mm512_mul_epu32(_mm512_srli_epi32(x, 5), magic), 38);
    const __m512i od = _mm512_srli_epi64(_mm512_mul_epu32(_mm512_srli_epi64(x, 37), magic), 6);
    return _mm512_mask_mov_epi32(ev, (__mmask16)0xAAAA, od);
}

// MODE bits: 1 renormalise, 2 four candidates (else one), 4 symbol output (transposes + stores), 8 the byte feed's
// bookkeeping, 16 divide by multiply-shift (else: help = range >> 16), 32 the round-3 form of the step: eight bytes of
// look-ahead looked at every fourth step instead of a per-step "window ran dry" test, the quotient worked out beside the
// renormalisation compare for both outcomes, candidates sorted by interval start with nested lower-bound compares, a select
// tree and the in-interval test after the selection
template <int MODE>
__attribute__((noinline)) uint32_t run(uint8_t* out)
{
    alignas(64) uint32_t r0[16], l0[16], lt1[16], sy0[16], sy1[16];
    for (int j = 0; j < 16; j++) { r0[j] = kTop - 12345u * j; l0[j] = 77777u * j; sy0[j] = 48000 - 100 * j; lt1[j] = sy0[j]; sy1[j] = 60000 - sy0[j]; }
    __m512i range = _mm512_load_si512(r0), low = _mm512_load_si512(l0);
    constexpr bool NEW = (MODE & 32) != 0;
    // (round-3 form: a lane with two candidates repeats its last one)
    __m512i lt[4] = {_mm512_setzero_si512(), _mm512_load_si512(lt1), NEW ? _mm512_load_si512(lt1) : _mm512_set1_epi32(0xffff), NEW ? _mm512_load_si512(lt1) : _mm512_set1_epi32(0xffff)};
    __m512i sy[4] = {_mm512_load_si512(sy0), _mm512_load_si512(sy1), NEW ? _mm512_load_si512(sy1) : _mm512_setzero_si512(), NEW ? _mm512_load_si512(sy1) : _mm512_setzero_si512()};
    __m512i sym[4] = {_mm512_set1_epi32(127), _mm512_set1_epi32(128), _mm512_set1_epi32(NEW ? 128 : 3), _mm512_set1_epi32(NEW ? 128 : 4)};
    const __mmask16 top[4] = {0, 0xffff, (__mmask16)(NEW ? 0xffff : 0), (__mmask16)(NEW ? 0xffff : 0)};
    __m512i win = _mm512_set1_epi32(0x5a3c9671), nxt = _mm512_set1_epi32(0x1e2d3c4b), cnt = _mm512_set1_epi32(4);
    __m512i used = _mm512_setzero_si512();
    const __m512i vbottom = _mm512_set1_epi32((int)kBottom), one = _mm512_set1_epi32(1), four = _mm512_set1_epi32(4);
    const __m512i magic = _mm512_set1_epi64(146601551);
    __m128i rows[16];
    uint32_t miss_count = 0;
    for (int i = 0; i < 16; i++) rows[i] = _mm_setzero_si128();
    for (uint32_t i = 0; i < (uint32_t)kSteps; i++) {
        asm volatile("" : "+v"(range), "+v"(low));  // the loop stays a loop
        __m512i help;
        if (NEW) {
            const __mmask16 sh = _mm512_cmple_epu32_mask(range, vbottom);
            const __m512i h0 = (MODE & 16) ? div60000(range) : _mm512_srli_epi32(range, 16);
            const __m512i h1 = (MODE & 16) ? div60000(_mm512_slli_epi32(range, 8)) : _mm512_srli_epi32(range, 8);
            if (MODE & 1) {
                low = _mm512_mask_or_epi32(low, sh, _mm512_slli_epi32(low, 8), _mm512_srli_epi32(win, 24));
                range = _mm512_mask_slli_epi32(range, sh, range, 8);
                if (MODE & 8) {
                    win = _mm512_mask_or_epi32(win, sh, _mm512_slli_epi32(win, 8), _mm512_srli_epi32(nxt, 24));
                    nxt = _mm512_mask_slli_epi32(nxt, sh, nxt, 8);
                    used = _mm512_mask_add_epi32(used, sh, used, one);
                } else win = _mm512_or_si512(win, one);
                help = _mm512_mask_mov_epi32(h0, sh, h1);
                __mmask16 again = _mm512_cmple_epu32_mask(range, vbottom);
                if (__builtin_expect(!_kortestz_mask16_u8(again, again), 0)) {
                    do { range = _mm512_mask_slli_epi32(range, again, range, 8); again = _mm512_cmple_epu32_mask(range, vbottom); } while (again);
                    help = (MODE & 16) ? div60000(range) : _mm512_srli_epi32(range, 16);
                }
                if ((MODE & 8) && (i & 3) == 3) {
                    const __mmask16 need = _mm512_cmpge_epu32_mask(used, four);
                    if (need) { used = _mm512_maskz_mov_epi32((__mmask16)~need, used); nxt = _mm512_mask_add_epi32(nxt, need, nxt, _mm512_set1_epi32(0x01010101)); win = _mm512_mask_mov_epi32(win, need, nxt); }
                }
            } else {
                range = _mm512_or_si512(range, _mm512_set1_epi32((int)kTop));
                help = h0;
            }
        } else {
            if (MODE & 1) {
                __mmask16 sh = _mm512_cmple_epu32_mask(range, vbottom);
                for (;;) {
                    low = _mm512_mask_or_epi32(low, sh, _mm512_slli_epi32(low, 8), _mm512_srli_epi32(win, 24));
                    range = _mm512_mask_slli_epi32(range, sh, range, 8);
                    win = _mm512_mask_slli_epi32(win, sh, win, 8);
                    if (MODE & 8) {
                        cnt = _mm512_mask_sub_epi32(cnt, sh, cnt, one);
                        const __mmask16 dry = _mm512_cmpeq_epu32_mask(cnt, _mm512_setzero_si512());
                        if (dry) {
                            win = _mm512_mask_mov_epi32(win, dry, nxt);
                            cnt = _mm512_mask_mov_epi32(cnt, dry, four);
                            nxt = _mm512_add_epi32(nxt, _mm512_set1_epi32(0x01010101));
                        }
                    } else win = _mm512_or_si512(win, one);
                    sh = _mm512_cmple_epu32_mask(range, vbottom);
                    if (__builtin_expect(sh == 0, 1)) break;
                }
            } else {
                range = _mm512_or_si512(range, _mm512_set1_epi32((int)kTop));  // keeps the range up without a compare
            }
            if (MODE & 16) {
                const __m512i ev = _mm512_srli_epi64(_mm512_mul_epu32(_mm512_srli_epi32(range, 5), magic), 38);
                const __m512i od = _mm512_srli_epi64(_mm512_mul_epu32(_mm512_srli_epi64(range, 37), magic), 6);
                help = _mm512_mask_mov_epi32(ev, (__mmask16)0xAAAA, od);
            } else help = _mm512_srli_epi32(range, 16);
        }
        constexpr int NC = (MODE & 2) ? 4 : 1;
        __m512i a[4], w[4];
        __m512i c, sa, nrange;
        if (NEW && NC == 4) {
            for (int e = 0; e < 4; e++) {
                a[e] = _mm512_mullo_epi32(help, lt[e]);
                w[e] = _mm512_mask_sub_epi32(_mm512_mullo_epi32(help, sy[e]), top[e], range, a[e]);
            }
            const __mmask16 ge1 = _mm512_cmpge_epu32_mask(low, a[1]), ge2 = _mm512_cmpge_epu32_mask(low, a[2]), ge3 = _mm512_cmpge_epu32_mask(low, a[3]);
            c = _mm512_mask_mov_epi32(_mm512_mask_mov_epi32(sym[0], ge1, sym[1]), ge2, _mm512_mask_mov_epi32(sym[2], ge3, sym[3]));
            sa = _mm512_mask_mov_epi32(_mm512_mask_mov_epi32(a[0], ge1, a[1]), ge2, _mm512_mask_mov_epi32(a[2], ge3, a[3]));
            nrange = _mm512_mask_mov_epi32(_mm512_mask_mov_epi32(w[0], ge1, w[1]), ge2, _mm512_mask_mov_epi32(w[2], ge3, w[3]));
            low = _mm512_sub_epi32(low, sa);
            const __mmask16 miss = _mm512_cmpge_epu32_mask(low, nrange);
            if (__builtin_expect(!_kortestz_mask16_u8(miss, miss), 0)) miss_count++;
        } else {
            __mmask16 in[4] = {0, 0, 0, 0};
            for (int e = 0; e < NC; e++) {
                a[e] = _mm512_mullo_epi32(help, lt[e]);
                w[e] = _mm512_mask_sub_epi32(_mm512_mullo_epi32(help, sy[e]), top[e], range, a[e]);
                in[e] = _mm512_cmplt_epu32_mask(_mm512_sub_epi32(low, a[e]), w[e]);
            }
            c = sym[0]; sa = a[0]; nrange = w[0];
            for (int e = 1; e < NC; e++) {
                c = _mm512_mask_mov_epi32(c, in[e], sym[e]);
                sa = _mm512_mask_mov_epi32(sa, in[e], a[e]);
                nrange = _mm512_mask_mov_epi32(nrange, in[e], w[e]);
            }
            low = _mm512_sub_epi32(low, sa);
        }
        range = nrange;
        if (MODE & 4) {
            rows[i & 15] = _mm512_cvtepi32_epi8(c);
            if ((i & 15) == 15) {
                __m128i t[16];
                for (int k = 0; k < 8; k++) { t[2 * k] = _mm_unpacklo_epi8(rows[k], rows[k + 8]); t[2 * k + 1] = _mm_unpackhi_epi8(rows[k], rows[k + 8]); }
                for (int k = 0; k < 8; k++) { rows[2 * k] = _mm_unpacklo_epi8(t[k], t[k + 8]); rows[2 * k + 1] = _mm_unpackhi_epi8(t[k], t[k + 8]); }
                for (int k = 0; k < 8; k++) { t[2 * k] = _mm_unpacklo_epi8(rows[k], rows[k + 8]); t[2 * k + 1] = _mm_unpackhi_epi8(rows[k], rows[k + 8]); }
                for (int k = 0; k < 8; k++) { rows[2 * k] = _mm_unpacklo_epi8(t[k], t[k + 8]); rows[2 * k + 1] = _mm_unpackhi_epi8(t[k], t[k + 8]); }
                for (int j = 0; j < 16; j++) _mm_storeu_si128(reinterpret_cast<__m128i*>(out + j * 65536 + ((i - 15) & 0xfff0)), rows[j]);
            }
        }
    }
    return (uint32_t)_mm512_reduce_add_epi32(_mm512_add_epi32(range, low)) + (uint32_t)_mm512_reduce_add_epi32(win) + miss_count;
}

// a chain of dependent one-cycle vector adds: the clock
__attribute__((noinline)) uint32_t clock_probe()
{
    __m512i x = _mm512_set1_epi32(1);
    const __m512i y = _mm512_set1_epi32(3);
    for (int i = 0; i < kSteps; i++) {
#define ADD1 x = _mm512_add_epi32(x, y); asm volatile("" : "+v"(x));
        ADD1 ADD1 ADD1 ADD1 ADD1 ADD1 ADD1 ADD1
    }
    return (uint32_t)_mm512_reduce_add_epi32(x);
}

template <int MODE>
void report(const char* what, uint8_t* out, double ghz)
{
    double best = 1e9;
    uint32_t sink = 0;
    for (int r = 0; r < 3; r++) { const double t = now(); sink += run<MODE>(out); const double dt = now() - t; if (dt < best) best = dt; }
    printf("%-78s %6.2f ns/step  %5.1f cycles  (%u)\n", what, best / kSteps * 1e9, best / kSteps * ghz * 1e9, sink);
}

int main()
{
    static uint8_t out[16 * 65536 + 64];
    double best = 1e9;
    uint32_t s = 0;
    for (int r = 0; r < 3; r++) { const double t = now(); s += clock_probe(); const double dt = now() - t; if (dt < best) best = dt; }
    const double ghz = 8.0 * kSteps / best * 1e-9;
    printf("clock from a chain of dependent vpaddd zmm: %.2f GHz (%u)\n", ghz, s);
    printf("-- the step as it was at the start of round 3\n");
    report<16>("division by multiply-shift + one candidate (2 multiplies, compare, subtract)", out, ghz);
    report<0>("the same with help = range >> 16", out, ghz);
    report<16 + 2>("+ four candidates and the selects", out, ghz);
    report<16 + 2 + 1>("+ renormalisation (compare -> mask -> shifts, second compare, branch)", out, ghz);
    report<16 + 2 + 1 + 8>("+ window counter and the ran-dry test", out, ghz);
    report<16 + 2 + 1 + 8 + 4>("+ symbol output = the whole step", out, ghz);
    report<16 + 1 + 8 + 4>("the whole step with one candidate", out, ghz);
    printf("-- the step as it is now\n");
    report<32 + 16>("division (both outcomes) + one candidate", out, ghz);
    report<32 + 16 + 2>("+ four sorted candidates, nested compares, select tree, in-interval test", out, ghz);
    report<32 + 16 + 2 + 1>("+ renormalisation (mask picks the quotient)", out, ghz);
    report<32 + 16 + 2 + 1 + 8>("+ byte feed: two registers shifting as one, counter, a look every fourth step", out, ghz);
    report<32 + 16 + 2 + 1 + 8 + 4>("+ symbol output = the whole step", out, ghz);
    return 0;
}
