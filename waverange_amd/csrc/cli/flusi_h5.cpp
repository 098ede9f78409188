// flusi_h5.cpp -- see flusi_h5.h.  Plain HDF5 C API (1.8+), serial, one open/close per call like
// the reference; every function aborts with a message on an HDF5 failure.
#include "flusi_h5.h"

#include <dlfcn.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

// The HDF5 C library is bound at run time (dlopen) instead of at link time: the only HDF5 on this
// image lives under /opt/conda/lib next to an older libstdc++, and putting that directory on the
// link or run path breaks the HIP runtime's C++ dependencies.  Minimal declarations of the 1.10+
// ABI are restated here so that no HDF5 header (whose macros call H5open/H5check_version) is needed.
typedef int64_t hid_t;
typedef int herr_t;
typedef int htri_t;
typedef unsigned long long hsize_t;
typedef long long hssize_t;
namespace {
constexpr unsigned H5F_ACC_RDONLY = 0u, H5F_ACC_RDWR = 1u, H5F_ACC_TRUNC = 2u;
constexpr hid_t H5P_DEFAULT = 0, H5S_ALL = 0;
constexpr int H5D_ALLOC_TIME_EARLY = 1, H5_INDEX_NAME = 0, H5_ITER_INC = 0;
typedef herr_t (*iter_cb)(hid_t, const char*, const void*, void*);

struct H5 {
    void* so = nullptr;
    herr_t (*open)() = nullptr;
    herr_t (*get_libversion)(unsigned*, unsigned*, unsigned*) = nullptr;
    hid_t (*Fcreate)(const char*, unsigned, hid_t, hid_t) = nullptr;
    hid_t (*Fopen)(const char*, unsigned, hid_t) = nullptr;
    herr_t (*Fclose)(hid_t) = nullptr;
    hid_t (*Dopen2)(hid_t, const char*, hid_t) = nullptr;
    herr_t (*Dclose)(hid_t) = nullptr;
    hid_t (*Dcreate2)(hid_t, const char*, hid_t, hid_t, hid_t, hid_t, hid_t) = nullptr;
    herr_t (*Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void*) = nullptr;
    herr_t (*Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void*) = nullptr;
    hid_t (*Dget_space)(hid_t) = nullptr;
    hid_t (*Screate_simple)(int, const hsize_t*, const hsize_t*) = nullptr;
    herr_t (*Sclose)(hid_t) = nullptr;
    hssize_t (*Sget_simple_extent_npoints)(hid_t) = nullptr;
    htri_t (*Aexists)(hid_t, const char*) = nullptr;
    hid_t (*Aopen)(hid_t, const char*, hid_t) = nullptr;
    hid_t (*Acreate2)(hid_t, const char*, hid_t, hid_t, hid_t, hid_t) = nullptr;
    herr_t (*Aread)(hid_t, hid_t, void*) = nullptr;
    herr_t (*Awrite)(hid_t, hid_t, const void*) = nullptr;
    herr_t (*Aclose)(hid_t) = nullptr;
    hid_t (*Aget_space)(hid_t) = nullptr;
    hid_t (*Pcreate)(hid_t) = nullptr;
    herr_t (*Pclose)(hid_t) = nullptr;
    herr_t (*Pset_alloc_time)(hid_t, int) = nullptr;
    htri_t (*Lexists)(hid_t, const char*, hid_t) = nullptr;
    herr_t (*Literate)(hid_t, int, int, hsize_t*, iter_cb, void*) = nullptr;
    herr_t (*Eset_auto2)(hid_t, void*, void*) = nullptr;
    hid_t *t_double = nullptr, *t_float = nullptr, *t_int = nullptr, *t_uchar = nullptr, *t_ulong = nullptr, *p_dcreate = nullptr;
};

H5& h5()
{
    static H5 h;
    if (h.so) return h;
    std::string tried;
    const char* root = getenv("HDF5_ROOT");
    const std::string cands[] = {root ? std::string(root) + "/lib/libhdf5.so" : std::string(), "libhdf5.so", "libhdf5_serial.so",
                                 "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so"};
    for (const std::string& c : cands) {
        if (c.empty()) continue;
        h.so = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (h.so) break;
        tried += " " + c;
    }
    if (!h.so) { fprintf(stderr, "flusi hdf5: no HDF5 C library found (tried:%s); set HDF5_ROOT\n", tried.c_str()); exit(2); }
    auto sym = [&](const char* n) { void* p = dlsym(h.so, n); if (!p) { fprintf(stderr, "flusi hdf5: symbol %s missing\n", n); exit(2); } return p; };
#define WR_H5(field, name) h.field = reinterpret_cast<decltype(h.field)>(sym(name))
    WR_H5(open, "H5open"); WR_H5(get_libversion, "H5get_libversion"); WR_H5(Fcreate, "H5Fcreate"); WR_H5(Fopen, "H5Fopen");
    WR_H5(Fclose, "H5Fclose"); WR_H5(Dopen2, "H5Dopen2"); WR_H5(Dclose, "H5Dclose"); WR_H5(Dcreate2, "H5Dcreate2");
    WR_H5(Dread, "H5Dread"); WR_H5(Dwrite, "H5Dwrite"); WR_H5(Dget_space, "H5Dget_space"); WR_H5(Screate_simple, "H5Screate_simple");
    WR_H5(Sclose, "H5Sclose"); WR_H5(Sget_simple_extent_npoints, "H5Sget_simple_extent_npoints"); WR_H5(Aexists, "H5Aexists");
    WR_H5(Aopen, "H5Aopen"); WR_H5(Acreate2, "H5Acreate2"); WR_H5(Aread, "H5Aread"); WR_H5(Awrite, "H5Awrite"); WR_H5(Aclose, "H5Aclose");
    WR_H5(Aget_space, "H5Aget_space"); WR_H5(Pcreate, "H5Pcreate"); WR_H5(Pclose, "H5Pclose"); WR_H5(Pset_alloc_time, "H5Pset_alloc_time");
    WR_H5(Lexists, "H5Lexists"); WR_H5(Literate, "H5Literate"); WR_H5(Eset_auto2, "H5Eset_auto2");
    WR_H5(t_double, "H5T_NATIVE_DOUBLE_g"); WR_H5(t_float, "H5T_NATIVE_FLOAT_g"); WR_H5(t_int, "H5T_NATIVE_INT_g");
    WR_H5(t_uchar, "H5T_NATIVE_UCHAR_g"); WR_H5(t_ulong, "H5T_NATIVE_ULONG_g"); WR_H5(p_dcreate, "H5P_CLS_DATASET_CREATE_ID_g");
#undef WR_H5
    unsigned maj = 0, min = 0, rel = 0;
    h.get_libversion(&maj, &min, &rel);
    if (maj != 1 || min < 10) { fprintf(stderr, "flusi hdf5: HDF5 %u.%u.%u found, 1.10 or newer needed (64-bit hid_t)\n", maj, min, rel); exit(2); }
    h.open();
    h.Eset_auto2(0, nullptr, nullptr);  // quiet probing (H5Dopen2 on a group, missing attributes)
    return h;
}
// the spellings the rest of this file uses
#define H5Fcreate h5().Fcreate
#define H5Fopen h5().Fopen
#define H5Fclose h5().Fclose
#define H5Dopen2 h5().Dopen2
#define H5Dclose h5().Dclose
#define H5Dcreate2 h5().Dcreate2
#define H5Dread h5().Dread
#define H5Dwrite h5().Dwrite
#define H5Dget_space h5().Dget_space
#define H5Screate_simple h5().Screate_simple
#define H5Sclose h5().Sclose
#define H5Sget_simple_extent_npoints h5().Sget_simple_extent_npoints
#define H5Aexists h5().Aexists
#define H5Aopen h5().Aopen
#define H5Acreate2 h5().Acreate2
#define H5Aread h5().Aread
#define H5Awrite h5().Awrite
#define H5Aclose h5().Aclose
#define H5Aget_space h5().Aget_space
#define H5Pcreate h5().Pcreate
#define H5Pclose h5().Pclose
#define H5Pset_alloc_time h5().Pset_alloc_time
#define H5Lexists h5().Lexists
#define H5Literate h5().Literate
#define H5T_NATIVE_DOUBLE (*h5().t_double)
#define H5T_NATIVE_FLOAT (*h5().t_float)
#define H5T_NATIVE_INT (*h5().t_int)
#define H5T_NATIVE_UCHAR (*h5().t_uchar)
#define H5T_NATIVE_ULONG (*h5().t_ulong)
#define H5P_DATASET_CREATE (*h5().p_dcreate)
}  // namespace

namespace flusi {

const char* const kBackupNames[50] = {
    "ux", "uy", "uz", "nlkx0", "nlky0", "nlkz0", "nlkx1", "nlky1", "nlkz1", "bx", "by", "bz", "bnlkx0", "bnlky0",
    "bnlkz0", "bnlkx1", "bnlky1", "bnlkz1", "scalar1", "scalar1_nlk0", "scalar1_nlk1", "scalar2", "scalar2_nlk0",
    "scalar2_nlk1", "scalar3", "scalar3_nlk0", "scalar3_nlk1", "scalar4", "scalar4_nlk0", "scalar4_nlk1", "scalar5",
    "scalar5_nlk0", "scalar5_nlk1", "scalar6", "scalar6_nlk0", "scalar6_nlk1", "scalar7", "scalar7_nlk0",
    "scalar7_nlk1", "scalar8", "scalar8_nlk0", "scalar8_nlk1", "scalar9", "scalar9_nlk0", "scalar9_nlk1", "uavgx",
    "uavgy", "uavgz", "ekinavg", "Z_avg"};

namespace {

[[noreturn]] void die(const std::string& what)
{
    fprintf(stderr, "flusi hdf5: %s\n", what.c_str());
    exit(2);
}

struct File {
    hid_t id;
    File(const std::string& path, bool rw) : id(H5Fopen(path.c_str(), rw ? H5F_ACC_RDWR : H5F_ACC_RDONLY, H5P_DEFAULT))
    {
        if (id < 0) die("cannot open " + path);
    }
    ~File() { H5Fclose(id); }
};

struct Dset {
    hid_t id;
    Dset(hid_t file, const std::string& name) : id(H5Dopen2(file, name.c_str(), H5P_DEFAULT))
    {
        if (id < 0) die("cannot open dataset " + name);
    }
    ~Dset() { H5Dclose(id); }
};

void write_attr(hid_t dset, const char* name, hid_t type, const void* v, hsize_t n)
{
    hid_t attr;
    if (H5Aexists(dset, name) > 0) {
        attr = H5Aopen(dset, name, H5P_DEFAULT);
    } else {
        hid_t sp = H5Screate_simple(1, &n, NULL);
        attr = H5Acreate2(dset, name, type, sp, H5P_DEFAULT, H5P_DEFAULT);
        H5Sclose(sp);
    }
    if (attr < 0 || (n > 0 && H5Awrite(attr, type, v) < 0)) die(std::string("cannot write attribute ") + name);
    H5Aclose(attr);
}

bool read_attr(hid_t dset, const char* name, hid_t type, void* out, hsize_t n)
{
    if (H5Aexists(dset, name) <= 0) return false;
    hid_t attr = H5Aopen(dset, name, H5P_DEFAULT);
    hid_t sp = H5Aget_space(attr);
    const hssize_t have = H5Sget_simple_extent_npoints(sp);
    H5Sclose(sp);
    bool ok = have == (hssize_t)n && (n == 0 || H5Aread(attr, type, out) >= 0);
    H5Aclose(attr);
    return ok;
}

herr_t collect(hid_t, const char* name, const void*, void* data)
{
    static_cast<std::vector<std::string>*>(data)->push_back(name);
    return 0;
}

}  // namespace

void create_file(const std::string& path)
{
    hid_t f = H5Fcreate(path.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);  // main_enc.cpp:214
    if (f < 0) die("cannot create " + path);
    H5Fclose(f);
}

std::vector<std::string> dataset_names(const std::string& path)
{
    File f(path, false);
    std::vector<std::string> all, out;
    H5Literate(f.id, H5_INDEX_NAME, H5_ITER_INC, NULL, collect, &all);
    for (const std::string& n : all) {  // a root link is a dataset iff H5Dopen2 accepts it
        hid_t d = H5Dopen2(f.id, n.c_str(), H5P_DEFAULT);
        if (d >= 0) { H5Dclose(d); out.push_back(n); }
    }
    return out;
}

bool has_dataset(const std::string& path, const std::string& name)
{
    File f(path, false);
    return H5Lexists(f.id, name.c_str(), H5P_DEFAULT) > 0;  // main_enc.cpp:457
}

bool read_attr_double(const std::string& path, const std::string& dset, const char* attr, double* out, int n)
{
    File f(path, false); Dset d(f.id, dset);
    return read_attr(d.id, attr, H5T_NATIVE_DOUBLE, out, (hsize_t)n);
}

bool read_attr_int(const std::string& path, const std::string& dset, const char* attr, int* out, int n)
{
    File f(path, false); Dset d(f.id, dset);
    return read_attr(d.id, attr, H5T_NATIVE_INT, out, (hsize_t)n);
}

void write_attr_double(const std::string& path, const std::string& dset, const char* attr, const double* v, int n)
{
    File f(path, true); Dset d(f.id, dset);
    write_attr(d.id, attr, H5T_NATIVE_DOUBLE, v, (hsize_t)n);
}

void write_attr_int(const std::string& path, const std::string& dset, const char* attr, const int* v, int n)
{
    File f(path, true); Dset d(f.id, dset);
    write_attr(d.id, attr, H5T_NATIVE_INT, v, (hsize_t)n);
}

void read_field(const std::string& path, const std::string& dset, std::vector<double>& fld, size_t expect)
{
    File f(path, false); Dset d(f.id, dset);
    hid_t sp = H5Dget_space(d.id);
    const hssize_t have = H5Sget_simple_extent_npoints(sp);
    H5Sclose(sp);
    if ((size_t)have != expect) die("dataset " + dset + " does not hold nx*ny*nz values");
    fld.resize(expect);
    // fp32 inputs are widened by the library, as in the reference (hdf5_interfaces.cpp:730)
    if (H5Dread(d.id, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, fld.data()) < 0) die("cannot read " + dset);
}

void write_field(const std::string& path, const std::string& dset, const double* fld, int nx, int ny, int nz, bool single)
{
    File f(path, true);  // hdf5_interfaces.cpp:671-701
    const hsize_t dims[3] = {(hsize_t)nz, (hsize_t)ny, (hsize_t)nx};
    const hid_t type = single ? H5T_NATIVE_FLOAT : H5T_NATIVE_DOUBLE;
    hid_t sp = H5Screate_simple(3, dims, NULL);
    hid_t pl = H5Pcreate(H5P_DATASET_CREATE);
    H5Pset_alloc_time(pl, H5D_ALLOC_TIME_EARLY);
    hid_t ds = H5Dcreate2(f.id, dset.c_str(), type, sp, H5P_DEFAULT, pl, H5P_DEFAULT);
    H5Pclose(pl);
    if (ds < 0 || H5Dwrite(ds, H5T_NATIVE_DOUBLE, H5S_ALL, sp, H5P_DEFAULT, fld) < 0) die("cannot write " + dset);
    H5Dclose(ds);
    H5Sclose(sp);
}

void write_coded(const std::string& path, const std::string& dset, const unsigned char* data, const wr_enc_info& info)
{
    File f(path, true);
    const hsize_t n = info.ntot_enc;  // hdf5_interfaces.cpp:741-771 (a zero-length dataset for a trivial field)
    hid_t sp = H5Screate_simple(1, &n, NULL);
    hid_t ds = H5Dcreate2(f.id, dset.c_str(), H5T_NATIVE_UCHAR, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (ds < 0 || (n > 0 && H5Dwrite(ds, H5T_NATIVE_UCHAR, sp, sp, H5P_DEFAULT, data) < 0)) die("cannot write " + dset);
    H5Sclose(sp);
    // coding attributes, hdf5_interfaces.cpp:283-441 (the vectors are always written there: its
    // `if (ntot_enc > 0)` tests a pointer)
    const int cv = kCoderVersion;
    const unsigned long ne = info.ntot_enc;
    write_attr(ds, "coder_version", H5T_NATIVE_INT, &cv, 1);
    write_attr(ds, "tolabs", H5T_NATIVE_DOUBLE, &info.tolabs, 1);
    write_attr(ds, "midval", H5T_NATIVE_DOUBLE, &info.midval, 1);
    write_attr(ds, "halfspanval", H5T_NATIVE_DOUBLE, &info.halfspanval, 1);
    write_attr(ds, "wlev", H5T_NATIVE_UCHAR, &info.wlev, 1);
    write_attr(ds, "nlay", H5T_NATIVE_UCHAR, &info.nlay, 1);
    write_attr(ds, "ntot_enc", H5T_NATIVE_ULONG, &ne, 1);
    write_attr(ds, "deps_vec", H5T_NATIVE_DOUBLE, info.deps_vec, info.nlay);
    write_attr(ds, "minval_vec", H5T_NATIVE_DOUBLE, info.minval_vec, info.nlay);
    write_attr(ds, "len_enc_vec", H5T_NATIVE_ULONG, info.len_enc_vec, info.nlay);
    H5Dclose(ds);
}

void read_coded(const std::string& path, const std::string& dset, std::vector<unsigned char>& data, wr_enc_info& info)
{
    File f(path, false); Dset d(f.id, dset);
    info = wr_enc_info();
    unsigned long ne = 0;
    int cv = 0;
    if (!read_attr(d.id, "coder_version", H5T_NATIVE_INT, &cv, 1)) die(dset + ": no coder_version attribute (not a WaveRange file)");
    if (cv / 10000 != kCoderVersion / 10000) die(dset + ": incompatible coder version");  // MAJOR differs (defs.h:33)
    bool ok = read_attr(d.id, "tolabs", H5T_NATIVE_DOUBLE, &info.tolabs, 1) &&
              read_attr(d.id, "midval", H5T_NATIVE_DOUBLE, &info.midval, 1) &&
              read_attr(d.id, "halfspanval", H5T_NATIVE_DOUBLE, &info.halfspanval, 1) &&
              read_attr(d.id, "wlev", H5T_NATIVE_UCHAR, &info.wlev, 1) && read_attr(d.id, "nlay", H5T_NATIVE_UCHAR, &info.nlay, 1) &&
              read_attr(d.id, "ntot_enc", H5T_NATIVE_ULONG, &ne, 1);
    if (!ok || info.nlay > WR_NLAYMAX) die(dset + ": coding attributes missing or malformed");
    info.ntot_enc = ne;
    if (ne > 0) {
        ok = read_attr(d.id, "deps_vec", H5T_NATIVE_DOUBLE, info.deps_vec, info.nlay) &&
             read_attr(d.id, "minval_vec", H5T_NATIVE_DOUBLE, info.minval_vec, info.nlay) &&
             read_attr(d.id, "len_enc_vec", H5T_NATIVE_ULONG, info.len_enc_vec, info.nlay);
        if (!ok) die(dset + ": coding vectors missing");
        hid_t sp = H5Dget_space(d.id);
        const hssize_t have = H5Sget_simple_extent_npoints(sp);
        H5Sclose(sp);
        if ((unsigned long)have != ne) die(dset + ": dataset length differs from ntot_enc");
        data.resize(ne);
        if (H5Dread(d.id, H5T_NATIVE_UCHAR, H5S_ALL, H5S_ALL, H5P_DEFAULT, data.data()) < 0) die("cannot read " + dset);
    } else {
        data.clear();
    }
}

}  // namespace flusi
