// wr_dropin.cpp -- Part 1 of include/waverange_amd.h: the reference's own entry points (same unmangled symbols, argument
// order and meaning as libwaverange, src/core/wrappers.h:53,70,75,95,111,119; waveletcdf97_3d) on host pointers, on top of
// wr_encode_host / wr_decode_host / wr_transform_host.  "void + fatal" error behaviour as the reference's.
#include "wr_internal.h"

using namespace wri;

namespace {

// The reference's entry points are re-entrant on distinct buffers (wrappers.cpp works on locals only).
// Here every call borrows a context from a free list (created on demand, kept for reuse), so concurrent
// callers never share staging; their device stages serialise on the per-GPU stage locks.
std::mutex g_free_mu;
std::vector<wr_ctx*> g_free_ctx;

[[noreturn]] void fatal(const char* where)
{
    fprintf(stderr, "libwaverange_amd: %s: %s\n", where, last_error().c_str());
    abort();
}

struct ImplicitCtx {
    wr_ctx* c = nullptr;
    explicit ImplicitCtx(const char* where)
    {
        {
            std::lock_guard<std::mutex> lk(g_free_mu);
            if (!g_free_ctx.empty()) { c = g_free_ctx.back(); g_free_ctx.pop_back(); }
        }
        if (!c) {
            int dev = 0;
            if (const char* e = getenv("WR_DEVICE")) dev = atoi(e);
            if (wr_ctx_create(&c, dev, nullptr) != WR_OK) fatal(where);
        }
    }
    ~ImplicitCtx()
    {
        std::lock_guard<std::mutex> lk(g_free_mu);
        g_free_ctx.push_back(c);
    }
};

}  // namespace

extern "C" {

void setup_wr(int nx, int ny, int nz, unsigned char* nlaymax, unsigned long* ntot_enc_max)
{
    const unsigned long ntot = (unsigned long)nx * (unsigned long)ny * (unsigned long)nz;
    *nlaymax = WR_NLAYMAX;
    *ntot_enc_max = kSafetyBufferFactor * WR_NLAYMAX * (ntot < 1024ul ? 1024ul : ntot);
}

void encoding_wrap(int nx, int ny, int nz, double* fld_1d, int wtflag, int mx, int my, int mz, double* cutoffvec,
                   double* tolabs, double* midval, double* halfspanval, unsigned char* wlev, unsigned char* nlay,
                   unsigned long* ntot_enc, double* deps_vec, double* minval_vec, unsigned long* len_enc_vec,
                   unsigned char* data_enc)
{
    if (mx < 1 || my < 1 || mz < 1) { last_error() = "mx, my, mz must be >= 1"; fatal("encoding_wrap"); }
    ImplicitCtx ic("encoding_wrap");
    unsigned char nl; unsigned long cap;
    setup_wr(nx, ny, nz, &nl, &cap);
    wr_enc_info info;
    ic.c->keep_residual = writeback_residual() != 0;  // fld_1d ends up holding the residual (wrappers.cpp:397-398)
    if (wr_encode_host(ic.c, fld_1d, nx, ny, nz, wtflag, mx, my, mz, cutoffvec, &info, data_enc, cap, nullptr))
        fatal("encoding_wrap");
    *tolabs = info.tolabs; *midval = info.midval; *halfspanval = info.halfspanval;
    *wlev = info.wlev; *nlay = info.nlay; *ntot_enc = info.ntot_enc;
    for (int l = 0; l < info.nlay; l++) {
        deps_vec[l] = info.deps_vec[l];
        minval_vec[l] = info.minval_vec[l];
        len_enc_vec[l] = info.len_enc_vec[l];
    }
}

void decoding_wrap(int nx, int ny, int nz, double* fld_1d, double* tolabs, double* midval, double* halfspanval,
                   unsigned char* wlev, unsigned char* nlay, unsigned long* ntot_enc, double* deps_vec,
                   double* minval_vec, unsigned long* len_enc_vec, unsigned char* data_enc)
{
    (void)tolabs; (void)halfspanval;  // unused by the reference too (wrappers.h:62-64)
    ImplicitCtx ic("decoding_wrap");
    wr_enc_info info;
    memset(&info, 0, sizeof info);
    info.midval = *midval; info.wlev = *wlev; info.nlay = *nlay; info.ntot_enc = *ntot_enc;
    if (info.nlay > WR_NLAYMAX) { last_error() = "nlay > 8"; fatal("decoding_wrap"); }
    for (int l = 0; l < info.nlay; l++) {
        info.deps_vec[l] = deps_vec[l];
        info.minval_vec[l] = minval_vec[l];
        info.len_enc_vec[l] = len_enc_vec[l];
    }
    if (wr_decode_host(ic.c, fld_1d, nx, ny, nz, &info, data_enc, 0, nullptr)) fatal("decoding_wrap");
}

void setup_wr_f(int* nx, int* ny, int* nz, int* nlaymax, long* ntot_enc_max)
{
    const long ntot = (long)(*nx) * (long)(*ny) * (long)(*nz);
    *nlaymax = WR_NLAYMAX;
    *ntot_enc_max = (long)kSafetyBufferFactor * WR_NLAYMAX * (ntot < 1024L ? 1024L : ntot);
}

void encoding_wrap_f(int* nx, int* ny, int* nz, double* fld, int* wtflag, double* tolrel, double* tolabs,
                     double* midval, double* halfspanval, unsigned char* wlev, unsigned char* nlay, long* ntot_enc,
                     double* deps_vec, double* minval_vec, long* len_enc_vec, unsigned char* data_enc)
{
    unsigned long ne = 0, lens[WR_NLAYMAX] = {0};
    double cutoff = *tolrel;
    encoding_wrap(*nx, *ny, *nz, fld, *wtflag, 1, 1, 1, &cutoff, tolabs, midval, halfspanval, wlev, nlay, &ne,
                  deps_vec, minval_vec, lens, data_enc);
    *ntot_enc = (long)ne;
    for (int j = 0; j < WR_NLAYMAX; j++) len_enc_vec[j] = (long)lens[j];  // all 8, as wrappers.cpp:561-562
}

void decoding_wrap_f(int* nx, int* ny, int* nz, double* fld, double* midval, double* halfspanval,
                     unsigned char* wlev, unsigned char* nlay, long* ntot_enc, double* deps_vec,
                     double* minval_vec, long* len_enc_vec, unsigned char* data_enc)
{
    double tolabs = 0;
    unsigned long ne = (unsigned long)*ntot_enc, lens[WR_NLAYMAX];
    for (int j = 0; j < WR_NLAYMAX; j++) lens[j] = (unsigned long)len_enc_vec[j];
    decoding_wrap(*nx, *ny, *nz, fld, &tolabs, midval, halfspanval, wlev, nlay, &ne, deps_vec, minval_vec, lens, data_enc);
}

void waveletcdf97_3d(int n1, int n2, int n3, int lvl, double* x)
{
    ImplicitCtx ic("waveletcdf97_3d");
    if (wr_transform_host(ic.c, x, n1, n2, n3, lvl)) fatal("waveletcdf97_3d");
}

}  // extern "C"
