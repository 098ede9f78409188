/* shim_test.c -- plain C client of the drop-in symbols, driven by tests/test_dropin_native.py.
 *
 *   shim_test f  IN NX NY NZ TOL OUT        the Fortran shims setup_wr_f / encoding_wrap_f / decoding_wrap_f
 *                                           (reference src/core/wrappers.cpp:545-594; call pattern of
 *                                           examples/fortran/example_fort.f90:74-121): every scalar by
 *                                           pointer, lengths in signed long arrays of NLAYMAX = 8
 *   shim_test t  IN NX NY NZ TOL OUT  IN2 NX2 NY2 NZ2 TOL2 OUT2  REPS
 *                                           two threads calling encoding_wrap / decoding_wrap at the same
 *                                           time on their own buffers, REPS times each (the reference is
 *                                           re-entrant on distinct buffers)
 * IN: raw doubles, x fastest.  OUT: one record per call, see write_record().
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "waverange_amd.h"

static double* read_field(const char* path, size_t n)
{
    double* f = malloc(n * sizeof(double));
    FILE* fh = fopen(path, "rb");
    if (!f || !fh || fread(f, sizeof(double), n, fh) != n) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
    fclose(fh);
    return f;
}

/* record: 3 doubles (tolabs, midval, halfspanval), 2 x uint64 (wlev, nlay), uint64 ntot_enc, 8 doubles deps,
 * 8 doubles minval, 8 x int64 len_enc_vec, ntot_enc bytes, n doubles residual (fld after encode), n doubles rec */
static void write_record(FILE* fh, double tolabs, double midval, double halfspanval, unsigned char wlev, unsigned char nlay,
                         long ntot_enc, const double* deps, const double* minval, const long* lens,
                         const unsigned char* data, const double* resid, const double* rec, size_t n)
{
    const double s[3] = {tolabs, midval, halfspanval};
    const unsigned long long u[3] = {wlev, nlay, (unsigned long long)ntot_enc};
    long long l8[8];
    for (int j = 0; j < 8; j++) l8[j] = lens[j];
    fwrite(s, sizeof s, 1, fh);
    fwrite(u, sizeof u, 1, fh);
    fwrite(deps, sizeof(double), 8, fh);
    fwrite(minval, sizeof(double), 8, fh);
    fwrite(l8, sizeof l8, 1, fh);
    fwrite(data, 1, (size_t)ntot_enc, fh);
    fwrite(resid, sizeof(double), n, fh);
    fwrite(rec, sizeof(double), n, fh);
}

static int run_fortran_shims(char** a)
{
    int nx = atoi(a[1]), ny = atoi(a[2]), nz = atoi(a[3]);
    double tolrel = atof(a[4]);
    const size_t n = (size_t)nx * ny * nz;
    double* fld = read_field(a[0], n);
    double* rec = malloc(n * sizeof(double));
    int nlaymax = 0, wtflag = 1;
    long ntot_enc_max = 0, ntot_enc = -1, len_enc_vec[8];
    for (int j = 0; j < 8; j++) len_enc_vec[j] = -7;  /* the shim writes all 8 entries (wrappers.cpp:561-562) */
    setup_wr_f(&nx, &ny, &nz, &nlaymax, &ntot_enc_max);
    if (nlaymax != 8 || ntot_enc_max != 8L * (long)(n < 1024 ? 1024 : n)) { fprintf(stderr, "setup_wr_f: %d %ld\n", nlaymax, ntot_enc_max); return 1; }
    unsigned char* data_enc = malloc((size_t)ntot_enc_max);
    unsigned char wlev = 0, nlay = 0;
    double tolabs = 0, midval = 0, halfspanval = 0, deps_vec[8] = {0}, minval_vec[8] = {0};
    encoding_wrap_f(&nx, &ny, &nz, fld, &wtflag, &tolrel, &tolabs, &midval, &halfspanval, &wlev, &nlay, &ntot_enc, deps_vec,
                    minval_vec, len_enc_vec, data_enc);
    for (int j = nlay; j < 8; j++)
        if (len_enc_vec[j] != 0) { fprintf(stderr, "len_enc_vec[%d] = %ld, expected 0\n", j, len_enc_vec[j]); return 1; }
    decoding_wrap_f(&nx, &ny, &nz, rec, &midval, &halfspanval, &wlev, &nlay, &ntot_enc, deps_vec, minval_vec, len_enc_vec, data_enc);
    FILE* fh = fopen(a[5], "wb");
    if (!fh) return 2;
    write_record(fh, tolabs, midval, halfspanval, wlev, nlay, ntot_enc, deps_vec, minval_vec, len_enc_vec, data_enc, fld, rec, n);
    fclose(fh);
    free(data_enc); free(rec); free(fld);
    return 0;
}

struct job {
    char** a;
    int reps;
    int rc;
};

static void* thread_main(void* p)
{
    struct job* jb = p;
    char** a = jb->a;
    const int nx = atoi(a[1]), ny = atoi(a[2]), nz = atoi(a[3]);
    const size_t n = (size_t)nx * ny * nz;
    double* orig = read_field(a[0], n);
    double* fld = malloc(n * sizeof(double));
    double* rec = malloc(n * sizeof(double));
    unsigned char nlaymax;
    unsigned long cap;
    setup_wr(nx, ny, nz, &nlaymax, &cap);
    unsigned char* data_enc = malloc(cap);
    FILE* fh = fopen(a[5], "wb");
    if (!fh) { jb->rc = 2; return NULL; }
    for (int r = 0; r < jb->reps; r++) {
        double cutoff = atof(a[4]), tolabs, midval, halfspanval, deps_vec[8] = {0}, minval_vec[8] = {0};
        unsigned char wlev, nlay;
        unsigned long ntot_enc, lens[8] = {0};
        long slens[8];
        memcpy(fld, orig, n * sizeof(double));
        encoding_wrap(nx, ny, nz, fld, 1, 1, 1, 1, &cutoff, &tolabs, &midval, &halfspanval, &wlev, &nlay, &ntot_enc, deps_vec,
                      minval_vec, lens, data_enc);
        decoding_wrap(nx, ny, nz, rec, &tolabs, &midval, &halfspanval, &wlev, &nlay, &ntot_enc, deps_vec, minval_vec, lens, data_enc);
        for (int j = 0; j < 8; j++) slens[j] = (long)lens[j];
        write_record(fh, tolabs, midval, halfspanval, wlev, nlay, (long)ntot_enc, deps_vec, minval_vec, slens, data_enc, fld, rec, n);
    }
    fclose(fh);
    free(data_enc); free(rec); free(fld); free(orig);
    jb->rc = 0;
    return NULL;
}

int main(int argc, char** argv)
{
    if (argc == 8 && argv[1][0] == 'f') return run_fortran_shims(argv + 2);
    if (argc == 15 && argv[1][0] == 't') {
        struct job jobs[2] = {{argv + 2, atoi(argv[14]), -1}, {argv + 8, atoi(argv[14]), -1}};
        pthread_t th[2];
        for (int i = 0; i < 2; i++) pthread_create(&th[i], NULL, thread_main, &jobs[i]);
        for (int i = 0; i < 2; i++) pthread_join(th[i], NULL);
        return jobs[0].rc | jobs[1].rc;
    }
    fprintf(stderr, "usage: see the header of shim_test.c\n");
    return 2;
}
