/*
 * wr_oracle.h -- CPU oracle for the WaveRange encode/decode hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker / reported baseline.
 *
 * Parity pin: this restatement is checked bit-for-bit against the reference
 * itself (oracle/_ref/libwaverange_ref.so, built by oracle/Makefile from the
 * sources where they lie under /root/reference with -ffp-contract=off) by
 * tests/test_oracle_vs_ref.py, and against the committed golden vectors under
 * tests/golden/ (generated from that reference build by
 * tools/make_golden.py) by tests/test_oracle_golden.py.
 *
 * Canonical arithmetic: strict IEEE-754 double, no FMA contraction.
 */
#ifndef WR_ORACLE_H
#define WR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* format constants, reference src/core/defs.h:34-50 */
#define WRO_BLOCKSIZE 60000u
#define WRO_NLAYMAX 8u
#define WRO_WAV_LVL 4
#define WRO_WAV_ACC_COEF 1.75

/* 3-D CDF-9/7 lifting transform, in place; lvl>0 forward, lvl<0 inverse.
 * follows src/waveletcdf97_3d/waveletcdf97_3d.c:38-468 */
void wro_cdf97_3d(int n1, int n2, int n3, int lvl, double *x);

/* physical -> wavelet-space index map, waveletcdf97_3d.c:473-553 */
void wro_ind_p2w_3d(int lvlin, int n1, int n2, int n3, int i1in, int i2in, int i3in,
                    int *lvl, int *i1, int *i2, int *i3);

/* plane bit-stream coder, src/core/wrappers.cpp:68-149 + src/rangecod/rangecod.c:170-276.
 * out must hold at least 2*max(1024,n) bytes; returns the stream length. */
size_t wro_range_encode(const uint8_t *sym, size_t n, uint8_t *out);

/* plane bit-stream decoder, wrappers.cpp:153-224 + rangecod.c:282-404.
 * returns the number of symbols produced (never more than cap are stored). */
size_t wro_range_decode(const uint8_t *in, size_t len, uint8_t *sym, size_t cap);

/* min / max with the reference's libm fmin/fmax scan semantics (first of equal
 * values wins, NaNs skipped), wrappers.cpp:244-250, 308-314 */
void wro_minmax(const double *x, size_t n, double *mn, double *mx);

/* one quantizer plane: q = (uchar)(aopt*x + bopt), wrappers.cpp:339-340,384-389;
 * residual x -= q*deps + minval, :397-398 (always applied, as in the reference) */
void wro_quantize_plane(double *x, size_t n, double deps, double minval, uint8_t *q);

/* decoder accumulate: acc += q*deps + minval, wrappers.cpp:513-514 */
void wro_dequant_accum(double *acc, size_t n, const uint8_t *q, double deps, double minval);

/* encoding_wrap, wrappers.cpp:228-452 (uniform-cutoff branch; mtot>1 follows :343-379) */
void wro_encode(int nx, int ny, int nz, double *fld, int wtflag, int mx, int my, int mz,
                const double *cutoffvec, double *tolabs, double *midval, double *halfspanval,
                unsigned char *wlev, unsigned char *nlay, unsigned long *ntot_enc,
                double *deps_vec, double *minval_vec, unsigned long *len_enc_vec,
                unsigned char *data_enc);

/* decoding_wrap, wrappers.cpp:456-527 */
void wro_decode(int nx, int ny, int nz, double *fld, double midval, unsigned char wlev,
                unsigned char nlay, unsigned long ntot_enc, const double *deps_vec,
                const double *minval_vec, const unsigned long *len_enc_vec,
                const unsigned char *data_enc);

/* setup_wr, wrappers.cpp:531-541 */
void wro_setup(int nx, int ny, int nz, unsigned char *nlaymax, unsigned long *ntot_enc_max);

#ifdef __cplusplus
}
#endif
#endif
