/* h5tool -- test helper (plain C + libhdf5): creates FluSI-style HDF5 inputs from raw float64 files and
 * dumps HDF5 files as text + raw dataset bytes, so that the Python tests need no h5py.
 *   h5tool make OUT.h5 regular|backup NBYTES NX NY NZ  NAME RAW.bin [NAME RAW.bin ...]
 *   h5tool dump FILE.h5 OUTDIR        (one line per dataset / attribute on stdout, data in OUTDIR/NAME.bin)
 */
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void attr_d(hid_t ds, const char* name, const double* v, hsize_t n)
{
    hid_t sp = H5Screate_simple(1, &n, NULL);
    hid_t a = H5Acreate2(ds, name, H5T_NATIVE_DOUBLE, sp, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, H5T_NATIVE_DOUBLE, v);
    H5Aclose(a); H5Sclose(sp);
}

static int do_make(int argc, char** argv)
{
    const char* out = argv[2];
    const int backup = strcmp(argv[3], "backup") == 0;
    const int nbytes = atoi(argv[4]);
    const int nx = atoi(argv[5]), ny = atoi(argv[6]), nz = atoi(argv[7]);
    const size_t n = (size_t)nx * ny * nz;
    hid_t f = H5Fcreate(out, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    double* buf = (double*)malloc(n * sizeof(double));
    for (int i = 8; i + 1 < argc; i += 2) {
        FILE* fh = fopen(argv[i + 1], "rb");
        if (!fh || fread(buf, sizeof(double), n, fh) != n) { fprintf(stderr, "cannot read %s\n", argv[i + 1]); return 1; }
        fclose(fh);
        hsize_t dims[3] = {(hsize_t)nz, (hsize_t)ny, (hsize_t)nx};
        hid_t sp = H5Screate_simple(3, dims, NULL);
        hid_t ds = H5Dcreate2(f, argv[i], nbytes == 4 ? H5T_NATIVE_FLOAT : H5T_NATIVE_DOUBLE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        H5Dwrite(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf);  /* converted to fp32 by the library if asked */
        if (backup) {
            double b[8] = {1.25, 1e-3, 1.1e-3, 1.0, 4200.0, (double)nx, (double)ny, (double)nz};
            attr_d(ds, "bckp", b, 8);
        } else {
            double t = 3.5, nu = 1e-4, eps = 1e-3, dom[3] = {6.28, 3.14, 1.57};
            int nxyz[3] = {nx, ny, nz};
            attr_d(ds, "time", &t, 1); attr_d(ds, "viscosity", &nu, 1); attr_d(ds, "epsi", &eps, 1); attr_d(ds, "domain_size", dom, 3);
            hsize_t three = 3; hid_t s3 = H5Screate_simple(1, &three, NULL);
            hid_t a = H5Acreate2(ds, "nxyz", H5T_NATIVE_INT, s3, H5P_DEFAULT, H5P_DEFAULT);
            H5Awrite(a, H5T_NATIVE_INT, nxyz); H5Aclose(a); H5Sclose(s3);
        }
        H5Dclose(ds); H5Sclose(sp);
    }
    free(buf);
    H5Fclose(f);
    return 0;
}

static const char* outdir_g;

static herr_t attr_cb(hid_t obj, const char* name, const H5A_info_t* info, void* data)
{
    (void)info;
    hid_t a = H5Aopen(obj, name, H5P_DEFAULT);
    hid_t t = H5Aget_type(a), sp = H5Aget_space(a);
    hssize_t n = H5Sget_simple_extent_npoints(sp);
    H5T_class_t cls = H5Tget_class(t);
    size_t sz = H5Tget_size(t);
    printf("attr %s %s %s size=%zu n=%lld :", (const char*)data, name, cls == H5T_FLOAT ? "float" : "int", sz, (long long)n);
    if (cls == H5T_FLOAT) {
        double v[64]; if (n > 0 && n <= 64) { H5Aread(a, H5T_NATIVE_DOUBLE, v); for (hssize_t i = 0; i < n; i++) printf(" %a", v[i]); }
    } else {
        unsigned long long v[64]; if (n > 0 && n <= 64) { H5Aread(a, H5T_NATIVE_ULLONG, v); for (hssize_t i = 0; i < n; i++) printf(" %llu", v[i]); }
    }
    printf("\n");
    H5Sclose(sp); H5Tclose(t); H5Aclose(a);
    return 0;
}

static herr_t link_cb(hid_t g, const char* name, const H5L_info_t* info, void* data)
{
    (void)info; (void)data;
    hid_t ds = H5Dopen2(g, name, H5P_DEFAULT);
    if (ds < 0) return 0;
    hid_t t = H5Dget_type(ds), sp = H5Dget_space(ds);
    int rank = H5Sget_simple_extent_ndims(sp);
    hsize_t dims[8] = {0};
    H5Sget_simple_extent_dims(sp, dims, NULL);
    hssize_t n = H5Sget_simple_extent_npoints(sp);
    H5T_class_t cls = H5Tget_class(t);
    size_t sz = H5Tget_size(t);
    printf("dataset %s %s size=%zu rank=%d dims=", name, cls == H5T_FLOAT ? "float" : "int", sz, rank);
    for (int i = 0; i < rank; i++) printf("%s%llu", i ? "x" : "", (unsigned long long)dims[i]);
    printf(" n=%lld\n", (long long)n);
    char path[4096];
    snprintf(path, sizeof path, "%s/%s.bin", outdir_g, name);
    FILE* fh = fopen(path, "wb");
    if (n > 0) {
        void* buf = malloc((size_t)n * sz);
        hid_t nt = H5Tget_native_type(t, H5T_DIR_ASCEND);
        H5Dread(ds, nt, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf);
        fwrite(buf, sz, (size_t)n, fh);
        H5Tclose(nt); free(buf);
    }
    fclose(fh);
    H5Aiterate2(ds, H5_INDEX_NAME, H5_ITER_INC, NULL, attr_cb, (void*)name);
    H5Sclose(sp); H5Tclose(t); H5Dclose(ds);
    return 0;
}

int main(int argc, char** argv)
{
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    if (argc >= 10 && strcmp(argv[1], "make") == 0) return do_make(argc, argv);
    if (argc == 4 && strcmp(argv[1], "dump") == 0) {
        outdir_g = argv[3];
        hid_t f = H5Fopen(argv[2], H5F_ACC_RDONLY, H5P_DEFAULT);
        if (f < 0) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
        H5Literate(f, H5_INDEX_NAME, H5_ITER_INC, NULL, link_cb, NULL);
        H5Fclose(f);
        return 0;
    }
    fprintf(stderr, "usage: h5tool make|dump ...\n");
    return 2;
}
