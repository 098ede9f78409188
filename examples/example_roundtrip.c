/* example_roundtrip.c -- library usage in plain C, the flow of the reference's Fortran example
 * (examples/fortran/example_fort.f90:74-129): setup_wr -> allocate -> encoding_wrap -> decoding_wrap ->
 * print the L-infinity error.  Build against this repository's library under the reference's name:
 *
 *   gcc -O2 -I include examples/example_roundtrip.c -o example_roundtrip \
 *       -L waverange_amd -lwaverange -Wl,-rpath,$PWD/waverange_amd -Wl,-rpath,/opt/rocm/lib -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "waverange_amd.h"

int main(int argc, char** argv)
{
    const int nx = 64, ny = 64, nz = 64;
    const double tolrel = argc > 1 ? atof(argv[1]) : 1e-6;
    const size_t n = (size_t)nx * ny * nz;
    double* fld = malloc(n * sizeof(double));
    double* work = malloc(n * sizeof(double)); /* encoding_wrap overwrites its input (reference README.md:197) */
    double* rec = malloc(n * sizeof(double));
    /* smooth field + a ripple, x fastest (idx = ix + nx*(iy + ny*iz)), cf. example_fort.f90:82-91 */
    for (int iz = 0; iz < nz; iz++)
        for (int iy = 0; iy < ny; iy++)
            for (int ix = 0; ix < nx; ix++) {
                const double x = (double)ix / nx, y = (double)iy / ny, z = (double)iz / nz;
                fld[ix + (size_t)nx * (iy + (size_t)ny * iz)] =
                    10.0 * (4 * x * (1 - x)) * (4 * y * (1 - y)) * (4 * y * (1 - y)) * (1 - 2 * z) + 0.05 * (x - 0.5) * (y * z);
            }
    double fmax_abs = 0;
    for (size_t j = 0; j < n; j++) fmax_abs = fmax(fmax_abs, fabs(fld[j]));

    unsigned char nlaymax, wlev, nlay;
    unsigned long ntot_enc_max, ntot_enc, len_enc_vec[8];
    double tolabs, midval, halfspanval, deps_vec[8], minval_vec[8], cutoff = tolrel;
    setup_wr(nx, ny, nz, &nlaymax, &ntot_enc_max);
    unsigned char* data_enc = malloc(ntot_enc_max);

    for (size_t j = 0; j < n; j++) work[j] = fld[j];
    encoding_wrap(nx, ny, nz, work, 1, 1, 1, 1, &cutoff, &tolabs, &midval, &halfspanval, &wlev, &nlay, &ntot_enc,
                  deps_vec, minval_vec, len_enc_vec, data_enc);
    decoding_wrap(nx, ny, nz, rec, &tolabs, &midval, &halfspanval, &wlev, &nlay, &ntot_enc, deps_vec, minval_vec,
                  len_enc_vec, data_enc);

    double linf = 0;
    for (size_t j = 0; j < n; j++) linf = fmax(linf, fabs(rec[j] - fld[j]));
    printf("planes=%u coded=%lu bytes ratio=%.2f Linf_abs=%.3e Linf_rel=%.3e (tol %.1e)\n", (unsigned)nlay, ntot_enc,
           (double)(n * sizeof(double)) / (double)ntot_enc, linf, linf / fmax_abs, tolrel);
    free(data_enc); free(rec); free(work); free(fld);
    return linf / fmax_abs <= 1.05 * tolrel ? 0 : 1;
}
