/*
 * wr_oracle.c -- CPU oracle (plain C restatement) of the WaveRange hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- see wr_oracle.h.  Build with -ffp-contract=off:
 * the canonical arithmetic is "multiply, round, add, round" (SURVEY.md 8c).
 *
 * Each function cites the reference lines it restates (paths relative to
 * /root/reference/).  Pinned against the compiled reference and the golden
 * vectors by tests/test_oracle_vs_ref.py and tests/test_oracle_golden.py.
 */
#include "wr_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* CDF-9/7 lifting                                                          */
/* ------------------------------------------------------------------------ */

/* lifting and scale constants, src/waveletcdf97_3d/waveletcdf97_3d.c:41-45 */
static const double K_ALPHA = -1.5861343420693648;
static const double K_BETA = -0.0529801185718856;
static const double K_GAMMA = 0.8829110755411875;
static const double K_DELTA = 0.4435068520511142;
static const double K_ZETA = 1.1496043988602418;
static const double K_IZETA = 1.0 / 1.1496043988602418;

static size_t half_up(size_t n) { return n / 2 + (n % 2 ? 1 : 0); }

/* odd-length extrapolation weights, waveletcdf97_3d.c:55-58 (evaluated left to right) */
static void odd_ext_coefs(double e[3])
{
    e[0] = -2 * K_ALPHA * K_BETA * K_GAMMA / (1 + 2 * K_BETA * K_GAMMA);
    e[1] = -2 * K_BETA * K_GAMMA / (1 + 2 * K_BETA * K_GAMMA);
    e[2] = -2 * (K_ALPHA + K_GAMMA + 3 * K_ALPHA * K_BETA * K_GAMMA) / (1 + 2 * K_BETA * K_GAMMA);
}

/* "predict"-type step on the odd samples: d[i] += c*(s[i+1]+s[i]), mirrored at the
 * right edge as d[m-1] += (c*2)*s[m-1]   (waveletcdf97_3d.c:112-113,120-121) */
static void lift_odd(double *d, const double *s, size_t m, double c, double sign)
{
    size_t i;
    for (i = 0; i + 1 < m; i++) d[i] = d[i] + sign * (c * (s[i + 1] + s[i]));
    d[m - 1] = d[m - 1] + sign * (c * 2 * s[m - 1]);
}

/* "update"-type step on the even samples: s[i] += c*(d[i]+d[i-1]), mirrored at the
 * left edge as s[0] += (c*2)*d[0]   (waveletcdf97_3d.c:116-117,124-125) */
static void lift_even(double *s, const double *d, size_t m, double c, double sign)
{
    size_t i;
    s[0] = s[0] + sign * (c * 2 * d[0]);
    for (i = 1; i < m; i++) s[i] = s[i] + sign * (c * (d[i] + d[i - 1]));
}

/* forward transform of one strided line of length n >= 2, waveletcdf97_3d.c:98-135 */
static void line_forward(double *p, size_t stride, size_t n, double *s, double *d, const double e[3])
{
    size_t m = half_up(n), i;
    for (i = 0; i < m; i++) {
        s[i] = p[(2 * i) * stride];
        if (2 * i + 1 < n) d[i] = p[(2 * i + 1) * stride];
    }
    if (n % 2) d[m - 1] = s[m - 2] * e[0] + d[m - 2] * e[1] + s[m - 1] * e[2];
    lift_odd(d, s, m, K_ALPHA, 1.0);
    lift_even(s, d, m, K_BETA, 1.0);
    lift_odd(d, s, m, K_GAMMA, 1.0);
    lift_even(s, d, m, K_DELTA, 1.0);
    for (i = 0; i < m; i++) {
        p[i * stride] = s[i] * K_ZETA;
        if (2 * i + 1 < n) p[(i + m) * stride] = d[i] * K_IZETA;
    }
}

/* inverse transform of one strided line of length m >= 2, waveletcdf97_3d.c:308-340 */
static void line_inverse(double *p, size_t stride, size_t m, double *s, double *d)
{
    size_t q = half_up(m), i;
    for (i = 0; i < q; i++) s[i] = p[i * stride] * K_IZETA;
    for (i = 0; i < m - q; i++) d[i] = p[(i + q) * stride] * K_ZETA;
    if (m % 2) d[q - 1] = 0;
    lift_even(s, d, q, K_DELTA, -1.0);
    lift_odd(d, s, q, K_GAMMA, -1.0);
    lift_even(s, d, q, K_BETA, -1.0);
    lift_odd(d, s, q, K_ALPHA, -1.0);
    for (i = 0; i < q; i++) {
        p[(2 * i) * stride] = s[i];
        if (2 * i + 1 < m) p[(2 * i + 1) * stride] = d[i];
    }
}

/* NOTE on `sign * (c * (...))`: sign is exactly +1 or -1, so the product is exact and
 * `a + (-1)*t` == `a - t` bit for bit; this keeps one code path for both directions. */

void wro_cdf97_3d(int n1in, int n2in, int n3in, int lvl, double *x)
{
    const size_t L1 = (size_t)n1in, L2 = (size_t)n2in, L3 = (size_t)n3in;
    const size_t sx = 1, sy = L1, sz = L1 * L2;
    size_t big = L1 > L2 ? L1 : L2;
    double e[3], *s, *d;
    size_t a, b;
    if (L3 > big) big = L3;
    s = (double *)malloc((big / 2 + 2) * sizeof(double));
    d = (double *)malloc((big / 2 + 2) * sizeof(double));
    odd_ext_coefs(e);

    if (lvl >= 0) {
        /* forward: x, y, z passes on a corner box that halves every level (:73-276) */
        size_t c1 = L1, c2 = L2, c3 = L3;
        int k;
        for (k = 0; k < lvl; k++) {
            if (c1 > 1)
                for (b = 0; b < c3; b++)
                    for (a = 0; a < c2; a++) line_forward(x + a * sy + b * sz, sx, c1, s, d, e);
            if (c2 > 1)
                for (b = 0; b < c3; b++)
                    for (a = 0; a < c1; a++) line_forward(x + a * sx + b * sz, sy, c2, s, d, e);
            if (c3 > 1)
                for (b = 0; b < c2; b++)
                    for (a = 0; a < c1; a++) line_forward(x + a * sx + b * sy, sz, c3, s, d, e);
            c1 = half_up(c1);
            c2 = half_up(c2);
            c3 = half_up(c3);
        }
    } else {
        /* inverse: coarsest box first, z, y, x order (:281-466) */
        int k;
        for (k = -lvl - 1; k >= 0; k--) {
            size_t p2 = (size_t)1 << k;
            size_t m1 = L1 / p2 + (L1 % p2 ? 1 : 0);
            size_t m2 = L2 / p2 + (L2 % p2 ? 1 : 0);
            size_t m3 = L3 / p2 + (L3 % p2 ? 1 : 0);
            if (m3 > 1)
                for (b = 0; b < m2; b++)
                    for (a = 0; a < m1; a++) line_inverse(x + a * sx + b * sy, sz, m3, s, d);
            if (m2 > 1)
                for (b = 0; b < m3; b++)
                    for (a = 0; a < m1; a++) line_inverse(x + a * sx + b * sz, sy, m2, s, d);
            if (m1 > 1)
                for (b = 0; b < m3; b++)
                    for (a = 0; a < m2; a++) line_inverse(x + a * sy + b * sz, sx, m1, s, d);
        }
    }
    free(s);
    free(d);
}

void wro_ind_p2w_3d(int lvlin, int n1, int n2, int n3, int i1in, int i2in, int i3in,
                    int *lvl, int *i1, int *i2, int *i3)
{
    /* waveletcdf97_3d.c:473-553.  Note the `touched` flag is sticky across levels there. */
    int c1 = n1, c2 = n2, c3 = n3, k, touched = 0;
    *lvl = 0;
    *i1 = i1in;
    *i2 = i2in;
    *i3 = i3in;
    for (k = 0; k < lvlin; k++) {
        int m1 = c1 / 2 + (c1 % 2 > 0), m2 = c2 / 2 + (c2 % 2 > 0), m3 = c3 / 2 + (c3 % 2 > 0);
        if (c1 > 1 && *i3 < c3 && *i2 < c2 && *i1 < c1) {
            *i1 = (*i1 % 2) ? *i1 / 2 + m1 : *i1 / 2;
            touched = 1;
        }
        if (c2 > 1 && *i3 < c3 && *i2 < c2 && *i1 < c1) {
            *i2 = (*i2 % 2) ? *i2 / 2 + m2 : *i2 / 2;
            touched = 1;
        }
        if (c3 > 1 && *i3 < c3 && *i2 < c2 && *i1 < c1) {
            *i3 = (*i3 % 2) ? *i3 / 2 + m3 : *i3 / 2;
            touched = 1;
        }
        c1 = m1;
        c2 = m2;
        c3 = m3;
        if (touched) *lvl += 1;
    }
}

/* ------------------------------------------------------------------------ */
/* rngcod13 range coder (32-bit, byte-wise renormalisation)                 */
/* ------------------------------------------------------------------------ */

/* src/rangecod/rangecod.c:120-129 */
#define RC_TOP 0x80000000u    /* 1 << 31 */
#define RC_BOTTOM 0x00800000u /* TOP >> 8 */
#define RC_SHIFT 23
#define RC_EXTRA 7

typedef struct {
    uint32_t low, range, pending, nbytes;
    uint8_t held;
    uint8_t *out;
    size_t pos;
} rc_enc;

static void enc_put(rc_enc *e, unsigned v) { e->out[e->pos++] = (uint8_t)v; }

/* rangecod.c:170-176 */
static void enc_start(rc_enc *e, uint8_t first, uint8_t *out)
{
    e->low = 0;
    e->range = RC_TOP;
    e->held = first;
    e->pending = 0;
    e->nbytes = 0;
    e->out = out;
    e->pos = 0;
}

/* rangecod.c:182-207 */
static void enc_renorm(rc_enc *e)
{
    while (e->range <= RC_BOTTOM) {
        if (e->low < ((uint32_t)0xff << RC_SHIFT)) {
            enc_put(e, e->held);
            for (; e->pending; e->pending--) enc_put(e, 0xff);
            e->held = (uint8_t)(e->low >> RC_SHIFT);
        } else if (e->low & RC_TOP) {
            enc_put(e, e->held + 1u);
            for (; e->pending; e->pending--) enc_put(e, 0x00);
            e->held = (uint8_t)(e->low >> RC_SHIFT);
        } else {
            e->pending++;
        }
        e->range <<= 8;
        e->low = (e->low << 8) & (RC_TOP - 1);
        e->nbytes++;
    }
}

/* rangecod.c:217-229 */
static void enc_freq(rc_enc *e, uint32_t sy, uint32_t lt, uint32_t tot)
{
    uint32_t r, t;
    enc_renorm(e);
    r = e->range / tot;
    t = r * lt;
    e->low += t;
    e->range -= t;
    if (lt + sy < tot) e->range = r * sy;
}

/* rangecod.c:231-245 */
static void enc_shift(rc_enc *e, uint32_t sy, uint32_t lt, uint32_t shift)
{
    uint32_t r, t;
    enc_renorm(e);
    r = e->range >> shift;
    t = r * lt;
    e->low += t;
    if ((lt + sy) >> shift)
        e->range -= t;
    else
        e->range = r * sy;
}

/* rangecod.c:254-276 */
static void enc_finish(rc_enc *e)
{
    uint32_t t;
    enc_renorm(e);
    e->nbytes += 5;
    if ((e->low & (RC_BOTTOM - 1)) < ((e->nbytes & 0xffffffu) >> 1))
        t = e->low >> RC_SHIFT;
    else
        t = (e->low >> RC_SHIFT) + 1;
    if (t > 0xff) {
        enc_put(e, e->held + 1u);
        for (; e->pending; e->pending--) enc_put(e, 0x00);
    } else {
        enc_put(e, e->held);
        for (; e->pending; e->pending--) enc_put(e, 0xff);
    }
    enc_put(e, t & 0xff);
    enc_put(e, (e->nbytes >> 16) & 0xff);
    enc_put(e, (e->nbytes >> 8) & 0xff);
    enc_put(e, e->nbytes & 0xff);
}

size_t wro_range_encode(const uint8_t *sym, size_t n, uint8_t *out)
{
    /* src/core/wrappers.cpp:68-149.  Blocks of up to 60000 symbols; a full final block
     * is followed by an empty one (consequence of the fill loop at :88-92). */
    rc_enc e;
    size_t done = 0;
    enc_start(&e, 0, out);
    for (;;) {
        uint32_t hist[257], bs, i;
        size_t left = n - done;
        bs = left < WRO_BLOCKSIZE ? (uint32_t)left : WRO_BLOCKSIZE;
        enc_freq(&e, 1, 1, 2); /* "a block follows" */
        memset(hist, 0, sizeof hist);
        for (i = 0; i < bs; i++) hist[sym[done + i]]++;
        for (i = 0; i < 256; i++) enc_shift(&e, 1, hist[i], 16);
        /* turn counts into exclusive prefix sums, hist[256] = bs (:108-110) */
        hist[256] = bs;
        for (i = 256; i; i--) hist[i - 1] = hist[i] - hist[i - 1];
        for (i = 0; i < bs; i++) {
            unsigned c = sym[done + i];
            enc_freq(&e, hist[c + 1] - hist[c], hist[c], bs);
        }
        done += bs;
        if (bs < WRO_BLOCKSIZE) break;
    }
    enc_freq(&e, 1, 0, 2); /* "no more blocks" */
    enc_finish(&e);
    return e.pos;
}

typedef struct {
    uint32_t low, range, help;
    uint8_t held;
    const uint8_t *in;
    size_t len, pos;
} rc_dec;

/* the reference reads without bounds checks (NOWARN, rangecod.c:105,163-171);
 * the oracle feeds zeros past the end instead of reading out of bounds */
static unsigned dec_get(rc_dec *d) { return d->pos < d->len ? d->in[d->pos++] : (d->pos++, 0u); }

/* rangecod.c:282-291 */
static void dec_start(rc_dec *d, const uint8_t *in, size_t len)
{
    d->in = in;
    d->len = len;
    d->pos = 0;
    d->help = 0;
    (void)dec_get(d); /* the byte given to start_encoding */
    d->held = (uint8_t)dec_get(d);
    d->low = d->held >> (8 - RC_EXTRA);
    d->range = (uint32_t)1 << RC_EXTRA;
}

/* rangecod.c:294-302 */
static void dec_renorm(rc_dec *d)
{
    while (d->range <= RC_BOTTOM) {
        d->low = (d->low << 8) | (((uint32_t)d->held << RC_EXTRA) & 0xff);
        d->held = (uint8_t)dec_get(d);
        d->low |= d->held >> (8 - RC_EXTRA);
        d->range <<= 8;
    }
}

/* rangecod.c:309-319 */
static uint32_t dec_culfreq(rc_dec *d, uint32_t tot)
{
    uint32_t t;
    dec_renorm(d);
    d->help = d->range / tot;
    t = d->low / d->help;
    return t >= tot ? tot - 1 : t;
}

/* rangecod.c:321-331 */
static uint32_t dec_culshift(rc_dec *d, uint32_t shift)
{
    uint32_t t;
    dec_renorm(d);
    d->help = d->range >> shift;
    t = d->low / d->help;
    return (t >> shift) ? ((uint32_t)1 << shift) - 1 : t;
}

/* rangecod.c:339-351 */
static void dec_update(rc_dec *d, uint32_t sy, uint32_t lt, uint32_t tot)
{
    uint32_t t = d->help * lt;
    d->low -= t;
    if (lt + sy < tot)
        d->range = d->help * sy;
    else
        d->range -= t;
}

size_t wro_range_decode(const uint8_t *in, size_t len, uint8_t *sym, size_t cap)
{
    /* src/core/wrappers.cpp:153-224 */
    rc_dec d;
    size_t produced = 0;
    uint16_t *lookup = (uint16_t *)malloc((WRO_BLOCKSIZE + 70000u) * sizeof(uint16_t));
    dec_start(&d, in, len);
    while (dec_culfreq(&d, 2)) {
        uint32_t cum[257], bs = 0, i, b;
        dec_update(&d, 1, 1, 2);
        for (i = 0; i < 256; i++) {
            uint32_t c = dec_culshift(&d, 16) & 0xffffu; /* decode_short, rangecod.c:362-366 */
            dec_update(&d, 1, c, (uint32_t)1 << 16);
            cum[i] = c;
        }
        for (i = 0; i < 256; i++) {
            uint32_t c = cum[i];
            cum[i] = bs;
            bs += c;
        }
        cum[256] = bs;
        if (bs > WRO_BLOCKSIZE + 65535u) break; /* corrupt stream guard (not in the reference) */
        for (b = 0; b < 256; b++)
            for (i = cum[b]; i < cum[b + 1]; i++) lookup[i] = (uint16_t)b;
        for (i = 0; i < bs; i++) {
            uint32_t cf = dec_culfreq(&d, bs), s = lookup[cf];
            dec_update(&d, cum[s + 1] - cum[s], cum[s], bs);
            if (produced < cap) sym[produced] = (uint8_t)s;
            produced++;
        }
    }
    dec_renorm(&d); /* done_decoding, rangecod.c:371-373 */
    free(lookup);
    return produced;
}

/* ------------------------------------------------------------------------ */
/* quantizer and the encode / decode wrappers                               */
/* ------------------------------------------------------------------------ */

void wro_minmax(const double *x, size_t n, double *mn, double *mx)
{
    /* wrappers.cpp:244-250: running libm fmin/fmax.  On the x86-64 glibc the reference was
     * built and run against here (2.35, minsd/maxsd based), fmin(a,b) returns a only when
     * a<b and b otherwise, so among equal values -- -0 vs +0 is the only case where it
     * shows -- the LAST one scanned wins; a NaN operand yields the other operand.
     * Spelled out so the oracle does not depend on which libm it is linked against. */
    double lo = x[0], hi = x[0];
    size_t j;
    for (j = 1; j < n; j++) {
        double v = x[j];
        if (v != v) continue;
        if (!(lo < v)) lo = v;
        if (!(hi > v)) hi = v;
    }
    *mn = lo;
    *mx = hi;
}

void wro_quantize_plane(double *x, size_t n, double deps, double minval, uint8_t *q)
{
    /* wrappers.cpp:339-340, 384-389, 397-398 */
    double aopt = 1.0 / deps;
    double bopt = -minval * aopt + 0.5;
    size_t j;
    for (j = 0; j < n; j++) {
        double fq = aopt * x[j] + bopt;
        q[j] = (uint8_t)fq;
    }
    for (j = 0; j < n; j++) x[j] = x[j] - (q[j] * deps + minval);
}

void wro_dequant_accum(double *acc, size_t n, const uint8_t *q, double deps, double minval)
{
    /* wrappers.cpp:513-514 */
    size_t j;
    for (j = 0; j < n; j++) acc[j] = acc[j] + (q[j] * deps + minval);
}

void wro_setup(int nx, int ny, int nz, unsigned char *nlaymax, unsigned long *ntot_enc_max)
{
    /* wrappers.cpp:531-541 */
    unsigned long ntot = (unsigned long)nx * (unsigned long)ny * (unsigned long)nz;
    *nlaymax = (unsigned char)WRO_NLAYMAX;
    *ntot_enc_max = WRO_NLAYMAX * (ntot < 1024ul ? 1024ul : ntot);
}

void wro_encode(int nx, int ny, int nz, double *fld, int wtflag, int mx, int my, int mz,
                const double *cutoffvec, double *tolabs, double *midval, double *halfspanval,
                unsigned char *wlev, unsigned char *nlay, unsigned long *ntot_enc,
                double *deps_vec, double *minval_vec, unsigned long *len_enc_vec,
                unsigned char *data_enc)
{
    /* wrappers.cpp:228-452 */
    const size_t ntot = (size_t)nx * (size_t)ny * (size_t)nz;
    const unsigned mtot = (unsigned)(mx * my * mz);
    double lo, hi, tolrel, tol;
    uint8_t *q, *stream;
    size_t total = 0;
    unsigned ilay = 0, k;
    int last = 0;

    *wlev = wtflag ? WRO_WAV_LVL : 0;
    wro_minmax(fld, ntot, &lo, &hi);
    *halfspanval = (hi - lo) / 2;
    *midval = lo + *halfspanval;
    if (*halfspanval <= 2 * DBL_MIN) { /* constant field: nothing is coded (:256-266) */
        *ntot_enc = 0;
        *nlay = 0;
        *tolabs = 0;
        return;
    }
    wro_cdf97_3d(nx, ny, nz, (int)*wlev, fld);

    tolrel = cutoffvec[0];
    for (k = 1; k < mtot; k++)
        if (cutoffvec[k] < tolrel) tolrel = cutoffvec[k];
    tol = tolrel * fmax(fabs(lo), fabs(hi)); /* extrema of the ORIGINAL field (:296) */
    tol /= WRO_WAV_ACC_COEF;
    *tolabs = tol;

    q = (uint8_t *)malloc(ntot);
    stream = (uint8_t *)malloc(2 * (ntot < 1024 ? 1024 : ntot));
    while (!last) {
        double deps;
        size_t len;
        wro_minmax(fld, ntot, &lo, &hi);
        minval_vec[ilay] = lo;
        deps = (hi - lo) / 255.0;
        if (deps < tol) {
            deps = tol;
            last = 1;
        }
        if (ilay >= WRO_NLAYMAX - 1) last = 1;
        deps_vec[ilay] = deps;
        if (mtot > 1) {
            /* local-cutoff branch, wrappers.cpp:343-379 */
            double aopt = 1.0 / deps, bopt = -lo * aopt + 0.5;
            size_t jp;
            for (jp = 0; jp < ntot; jp++) {
                int l, wx, wy, wz;
                int px = (int)(jp % (size_t)nx), py = (int)((jp / (size_t)nx) % (size_t)ny);
                int pz = (int)(jp / (size_t)nx / (size_t)ny);
                double mask = tol;
                size_t jw;
                wro_ind_p2w_3d(*wlev, nx, ny, nz, px, py, pz, &l, &wx, &wy, &wz);
                if (l <= 1) { /* LOC_CUTOFF_LVL, defs.h:42; lcl_prec, wrappers.cpp:55-64 */
                    int kx = (int)((double)px / (double)nx * (double)mx);
                    int ky = (int)((double)py / (double)ny * (double)my);
                    int kz = (int)((double)pz / (double)nz * (double)mz);
                    mask = tol / tolrel * cutoffvec[kx + mx * ky + mx * my * kz];
                }
                jw = (size_t)wx + (size_t)nx * (size_t)wy + (size_t)nx * (size_t)ny * (size_t)wz;
                if (hi - lo < mask) {
                    q[jw] = 0;
                    fld[jw] = lo;
                } else {
                    double fq = aopt * fld[jw] + bopt;
                    q[jw] = (uint8_t)fq;
                }
            }
            for (jp = 0; jp < ntot; jp++) fld[jp] = fld[jp] - (q[jp] * deps + lo);
        } else {
            wro_quantize_plane(fld, ntot, deps, lo, q);
        }
        len = wro_range_encode(q, ntot, stream);
        len_enc_vec[ilay] = (unsigned long)len;
        memcpy(data_enc + total, stream, len); /* caller sized data_enc via wro_setup */
        total += len;
        ilay++;
    }
    *nlay = (unsigned char)ilay;
    *ntot_enc = (unsigned long)total;
    free(q);
    free(stream);
}

void wro_decode(int nx, int ny, int nz, double *fld, double midval, unsigned char wlev,
                unsigned char nlay, unsigned long ntot_enc, const double *deps_vec,
                const double *minval_vec, const unsigned long *len_enc_vec,
                const unsigned char *data_enc)
{
    /* wrappers.cpp:456-527 */
    const size_t ntot = (size_t)nx * (size_t)ny * (size_t)nz;
    size_t j, off = 0;
    unsigned ilay;
    uint8_t *q;
    if (ntot_enc == 0) {
        for (j = 0; j < ntot; j++) fld[j] = midval;
        return;
    }
    q = (uint8_t *)malloc(ntot);
    for (j = 0; j < ntot; j++) fld[j] = 0;
    for (ilay = 0; ilay < nlay; ilay++) {
        wro_range_decode(data_enc + off, len_enc_vec[ilay], q, ntot);
        off += len_enc_vec[ilay];
        wro_dequant_accum(fld, ntot, q, deps_vec[ilay], minval_vec[ilay]);
    }
    wro_cdf97_3d(nx, ny, nz, -(int)wlev, fld);
    free(q);
}
