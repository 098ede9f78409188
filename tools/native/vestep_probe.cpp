// vestep_probe.cpp -- where the time of one step of the 16-lane ENCODER loop goes (wr_rangecoder_avx512.cpp,
// vec_encode_block_t<true, 0>: candidate compares, renormalisation in every step), on the CPU it runs on: the step's
// dataflow on synthetic two-symbol planes, with parts taken out.  Timing tool; what it writes is not a stream.
//   g++ -O3 -mavx512f -mavx512bw -mavx512dq -mavx512vl tools/native/vestep_probe.cpp -o vestep_probe && ./vestep_probe
#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

constexpr uint32_t kBottom = 0x00800000u, kTop = 0x80000000u;
constexpr int kBlock = 60000, kBlocks = 100;

static inline void transpose16x16(__m128i r[16])
{
    __m128i t[16];
    for (int i = 0; i < 8; i++) { t[2 * i] = _mm_unpacklo_epi8(r[i], r[i + 8]); t[2 * i + 1] = _mm_unpackhi_epi8(r[i], r[i + 8]); }
    for (int i = 0; i < 8; i++) { r[2 * i] = _mm_unpacklo_epi8(t[i], t[i + 8]); r[2 * i + 1] = _mm_unpackhi_epi8(t[i], t[i + 8]); }
    for (int i = 0; i < 8; i++) { t[2 * i] = _mm_unpacklo_epi8(r[i], r[i + 8]); t[2 * i + 1] = _mm_unpackhi_epi8(r[i], r[i + 8]); }
    for (int i = 0; i < 8; i++) { r[2 * i] = _mm_unpacklo_epi8(t[i], t[i + 8]); r[2 * i + 1] = _mm_unpackhi_epi8(t[i], t[i + 8]); }
}

// PART bits: 1 symbols in (loads + transposes; else a constant), 2 candidate compares and selects (else lt/sy of candidate
// 0), 4 bytes out: held byte / carry / 0xff run / pending word logic, 8 the stores of the pending words
template <int PART>
__attribute__((noinline)) uint32_t run(const uint8_t* syms, uint8_t* outbuf)
{
    const __m512i vbottom = _mm512_set1_epi32((int)kBottom), vtopm1 = _mm512_set1_epi32((int)(kTop - 1));
    alignas(64) uint32_t r0[16], l0[16];
    for (int j = 0; j < 16; j++) { r0[j] = kTop - 12345u * j; l0[j] = 77777u * j; }
    __m512i low = _mm512_load_si512(l0), range = _mm512_load_si512(r0);
    const __m512i top = _mm512_set1_epi32(128);
    __m512i cand[4] = {_mm512_set1_epi32(127), _mm512_set1_epi32(128), _mm512_set1_epi32(0x100), _mm512_set1_epi32(0x100)};
    __m512i clt[4] = {_mm512_setzero_si512(), _mm512_set1_epi32(48000), _mm512_setzero_si512(), _mm512_setzero_si512()};
    __m512i csy[4] = {_mm512_set1_epi32(48000), _mm512_set1_epi32(12000), _mm512_setzero_si512(), _mm512_setzero_si512()};
    const __m512i magic = _mm512_set1_epi64(146601551);
    const __m512i hi32 = _mm512_set1_epi64((long long)0xFFFFFFFF00000000ull);
    const __m512i one = _mm512_set1_epi32(1), four = _mm512_set1_epi32(4), v255 = _mm512_set1_epi32(0xff);
    const __m512i bswap = _mm512_broadcast_i32x4(_mm_set_epi8(12, 13, 14, 15, 8, 9, 10, 11, 4, 5, 6, 7, 0, 1, 2, 3));
    alignas(64) uint64_t addr[16];
    alignas(64) uint32_t vbuf[16];
    for (int j = 0; j < 16; j++) addr[j] = (uint64_t)(uintptr_t)(outbuf + (size_t)j * (1 << 20));
    __m512i alo = _mm512_load_si512(addr), ahi = _mm512_load_si512(addr + 8);
    __m512i held = _mm512_setzero_si512(), ffs = _mm512_setzero_si512(), pend = _mm512_setzero_si512(), cnt = _mm512_setzero_si512();
    __m128i rows[16];
    const __mmask16 act = 0xffff;
    auto flush_all = [&]() {
        const __m512i val = _mm512_shuffle_epi8(_mm512_sllv_epi32(pend, _mm512_slli_epi32(_mm512_sub_epi32(four, cnt), 3)), bswap);
        if (PART & 8) {
            _mm512_store_si512(vbuf, val);
            _mm512_store_si512(addr, alo);
            _mm512_store_si512(addr + 8, ahi);
#pragma GCC unroll 16
            for (int j = 0; j < 16; j++) memcpy(reinterpret_cast<void*>((uintptr_t)addr[j]), &vbuf[j], 4);
            alo = _mm512_add_epi64(alo, _mm512_cvtepu32_epi64(_mm512_castsi512_si256(cnt)));
            ahi = _mm512_add_epi64(ahi, _mm512_cvtepu32_epi64(_mm512_extracti64x4_epi64(cnt, 1)));
        }
        pend = _mm512_setzero_si512();
        cnt = _mm512_setzero_si512();
    };
    for (int blk = 0; blk < kBlocks; blk++) {
        for (int j = 0; j < 16; j++) addr[j] = (uint64_t)(uintptr_t)(outbuf + (size_t)j * (1 << 20));
        alo = _mm512_load_si512(addr); ahi = _mm512_load_si512(addr + 8);
        for (uint32_t i = 0; i < (uint32_t)kBlock; i++) {
            asm volatile("" : "+v"(range), "+v"(low));
            __m512i c = _mm512_set1_epi32(127);
            if (PART & 1) {
                if ((i & 15) == 0) {
                    for (int j = 0; j < 16; j++) rows[j] = _mm_loadu_si128(reinterpret_cast<const __m128i*>(syms + (size_t)j * kBlock + i));
                    transpose16x16(rows);
                }
                c = _mm512_cvtepu8_epi32(rows[i & 15]);
            }
            __m512i lt = clt[0], sy = csy[0];
            if (PART & 2) {
                const __mmask16 k1 = _mm512_cmpeq_epu32_mask(c, cand[1]), k2 = _mm512_cmpeq_epu32_mask(c, cand[2]), k3 = _mm512_cmpeq_epu32_mask(c, cand[3]);
                lt = _mm512_mask_mov_epi32(_mm512_mask_mov_epi32(_mm512_mask_mov_epi32(clt[0], k1, clt[1]), k2, clt[2]), k3, clt[3]);
                sy = _mm512_mask_mov_epi32(_mm512_mask_mov_epi32(_mm512_mask_mov_epi32(csy[0], k1, csy[1]), k2, csy[2]), k3, csy[3]);
            }
            __mmask16 sh = _mm512_cmple_epu32_mask(range, vbottom) & act;
            for (;;) {
                if (PART & 4) {
                    if (__builtin_expect(_mm512_cmpeq_epu32_mask(cnt, four) != 0, 0)) flush_all();
                    const __m512i v9 = _mm512_srli_epi32(low, 23);
                    const __mmask16 isff = _mm512_mask_cmpeq_epu32_mask(sh, v9, v255);
                    __mmask16 emit = sh & ~isff;
                    const __m512i carrybit = _mm512_srli_epi32(v9, 8);
                    const unsigned with_ffs = _mm512_mask_test_epi32_mask(emit, ffs, ffs);
                    if (__builtin_expect(with_ffs != 0, 0)) {  // (the probe just drops the run)
                        emit &= (__mmask16)~with_ffs;
                        ffs = _mm512_maskz_mov_epi32((__mmask16)~with_ffs, ffs);
                        held = _mm512_mask_and_epi32(held, (__mmask16)with_ffs, v9, v255);
                    }
                    const __m512i outb = _mm512_and_si512(_mm512_add_epi32(held, carrybit), v255);
                    pend = _mm512_mask_or_epi32(pend, emit, _mm512_slli_epi32(pend, 8), outb);
                    cnt = _mm512_mask_add_epi32(cnt, emit, cnt, one);
                    held = _mm512_mask_and_epi32(held, emit, v9, v255);
                    ffs = _mm512_mask_add_epi32(ffs, isff, ffs, one);
                }
                low = _mm512_mask_and_epi32(low, sh, _mm512_slli_epi32(low, 8), vtopm1);
                range = _mm512_mask_slli_epi32(range, sh, range, 8);
                sh = _mm512_cmple_epu32_mask(range, vbottom) & act;
                if (__builtin_expect(sh == 0, 1)) break;
            }
            const __m512i n5 = _mm512_srli_epi32(range, 5);
            const __m512i ev = _mm512_srli_epi64(_mm512_mul_epu32(n5, magic), 38);
            const __m512i od = _mm512_and_si512(_mm512_srli_epi64(_mm512_mul_epu32(_mm512_srli_epi64(n5, 32), magic), 6), hi32);
            const __m512i r = _mm512_or_si512(ev, od);
            const __m512i t = _mm512_mullo_epi32(r, lt);
            low = _mm512_add_epi32(low, t);
            const __mmask16 is_top = _mm512_cmpeq_epu32_mask(c, top);
            range = _mm512_mask_sub_epi32(_mm512_mullo_epi32(r, sy), is_top, range, t);
        }
    }
    return (uint32_t)_mm512_reduce_add_epi32(_mm512_add_epi32(range, low)) + (uint32_t)_mm512_reduce_add_epi32(_mm512_add_epi32(pend, held));
}

template <int PART>
void report(const char* what, const uint8_t* syms, uint8_t* out)
{
    double best = 1e9;
    uint32_t sink = 0;
    for (int r = 0; r < 3; r++) { const double t = now(); sink += run<PART>(syms, out); const double dt = now() - t; if (dt < best) best = dt; }
    printf("%-86s %6.2f ns/step  (%u)\n", what, best / ((double)kBlock * kBlocks) * 1e9, sink);
}

int main()
{
    uint8_t* syms = static_cast<uint8_t*>(aligned_alloc(64, 16 * kBlock + 64));
    uint8_t* out = static_cast<uint8_t*>(aligned_alloc(64, (size_t)16 << 20));
    uint64_t x = 88172645463325252ull;
    for (int i = 0; i < 16 * kBlock + 64; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; syms[i] = (x % 10) < 8 ? 127 : 128; }
    report<0>("range chain: renormalise, range / 60000, two multiplies (constant symbol)", syms, out);
    report<1>("+ symbols in (16 x 16 bytes loaded and transposed every 16 steps)", syms, out);
    report<1 + 2>("+ four candidate compares and the selects of {lt, sy}", syms, out);
    report<1 + 2 + 4>("+ bytes out: held byte, carry, 0xff run, pending word", syms, out);
    report<1 + 2 + 4 + 8>("+ the stores of the pending words = the whole step", syms, out);
    report<4 + 8>("bytes out alone on the range chain (constant symbol)", syms, out);
    return 0;
}
