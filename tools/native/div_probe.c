/* div_probe.c -- what a division costs on the host CPU: the 32-bit integer divide of the range decoder's step (low / help),
 * as a dependent chain (latency) and as 8 independent ones (throughput), against the same quotient through the FP divider
 * (cvtsi2sd x 2, divsd, cvttsd2si: exact for low < 2^32, help < 2^16).
 * gcc -O2 tools/native/div_probe.c -o div_probe && ./div_probe */
#include <stdint.h>
#include <stdio.h>
#include <time.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(void)
{
    const long N = 200000000L;
    volatile uint32_t seed = 2000000011u;
    /* 1. dependent chain of 32-bit divides: x = x / d + big  (keeps operands in the decoder's ranges) */
    {
        uint32_t x = seed, d = 12345;
        double t = now();
        for (long i = 0; i < N; i++) { x = x / d; x += 1900000000u; d = (d & 0x7fff) | 139; }
        t = now() - t;
        printf("int div r32, dependent chain:      %.2f ns per divide (x=%u)\n", t / N * 1e9, x);
    }
    /* 2. eight independent chains */
    {
        uint32_t x[8], d = 12345;
        for (int k = 0; k < 8; k++) x[k] = seed + k;
        double t = now();
        for (long i = 0; i < N / 8; i++)
            for (int k = 0; k < 8; k++) { x[k] = x[k] / (d + k); x[k] += 1900000000u; }
        t = now() - t;
        uint32_t s = 0; for (int k = 0; k < 8; k++) s += x[k];
        printf("int div r32, 8 independent chains: %.2f ns per divide (s=%u)\n", t / N * 1e9, s);
    }
    /* 3. the same through the FP divider, dependent chain */
    {
        uint32_t x = seed, d = 12345;
        double t = now();
        for (long i = 0; i < N; i++) { x = (uint32_t)(int64_t)((double)x / (double)d); x += 1900000000u; d = (d & 0x7fff) | 139; }
        t = now() - t;
        printf("cvt + divsd + cvt, dependent chain: %.2f ns per divide (x=%u)\n", t / N * 1e9, x);
    }
    /* 4. FP, eight independent chains */
    {
        uint32_t x[8], d = 12345;
        for (int k = 0; k < 8; k++) x[k] = seed + k;
        double t = now();
        for (long i = 0; i < N / 8; i++)
            for (int k = 0; k < 8; k++) { x[k] = (uint32_t)(int64_t)((double)x[k] / (double)(d + k)); x[k] += 1900000000u; }
        t = now() - t;
        uint32_t s = 0; for (int k = 0; k < 8; k++) s += x[k];
        printf("cvt + divsd + cvt, 8 independent:   %.2f ns per divide (s=%u)\n", t / N * 1e9, s);
    }
    /* 5. four integer + four FP chains together */
    {
        uint32_t x[8], d = 12345;
        for (int k = 0; k < 8; k++) x[k] = seed + k;
        double t = now();
        for (long i = 0; i < N / 8; i++) {
            for (int k = 0; k < 4; k++) { x[k] = x[k] / (d + k); x[k] += 1900000000u; }
            for (int k = 4; k < 8; k++) { x[k] = (uint32_t)(int64_t)((double)x[k] / (double)(d + k)); x[k] += 1900000000u; }
        }
        t = now() - t;
        uint32_t s = 0; for (int k = 0; k < 8; k++) s += x[k];
        printf("4 integer + 4 FP chains together:   %.2f ns per divide (s=%u)\n", t / N * 1e9, s);
    }
    return 0;
}
