// wr_api.cpp -- host side of libwaverange_amd: context, encode/decode pipeline, C ABI.
//
// Mirrors the reference's codec layer (src/core/wrappers.cpp) for the path
//   encoding_wrap : min/max -> forward transform -> bit-plane quantizer loop -> range coder
//   decoding_wrap : range decoder -> dequantise-accumulate -> inverse transform
// with the field resident in HBM, the quantized planes streamed to pinned host memory on a
// copy stream, and host range-coder threads (one per plane, or fewer with the planes of a field
// interleaved in one loop: wr_set_threads).  Compiled with hipcc, strict IEEE
// (-ffp-contract=off): the scalar arithmetic on deps/aopt/bopt/tolabs below must round
// exactly as wrappers.cpp:292-340 does.
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/waverange_amd.h"
#include "wr_kernels.h"
#include "wr_rangecoder.h"

#pragma clang fp contract(off)

namespace {

// reference src/core/defs.h:34-50
constexpr int kWavLvl = 4;
constexpr double kWavAccCoef = 1.75;
constexpr unsigned long kSafetyBufferFactor = 1;

thread_local std::string g_err;
// Device phases of the contexts on one GPU are serialised by the lock of that GPU's shared work space
// (DevPool below; they are tens of milliseconds); what overlaps between concurrent encode/decode calls
// is the host range coding.
int g_verbose = -1;  // -1: not initialised from the environment yet
int g_threads = -1;  // -1: not initialised from the environment yet (WR_THREADS, default one per plane)

int g_enc_threads = 0;  // 0: same as g_threads (wr_set_encoder_threads)

int coder_threads()
{
    if (g_threads < 0) {
        const char* e = getenv("WR_THREADS");
        const int k = e ? atoi(e) : 0;
        g_threads = k >= 1 ? k : WR_NLAYMAX;
    }
    return g_threads;
}
int encoder_threads() { return g_enc_threads > 0 ? g_enc_threads : coder_threads(); }

int verbose()
{
    if (g_verbose < 0) {
        const char* q = getenv("WR_QUIET");
        g_verbose = (q && *q && *q != '0') ? 0 : 1;
    }
    return g_verbose;
}

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(WR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// Device work space of the transform / quantizer phase, one per GPU and shared by all contexts on
// it: a device phase holds `phase` from its first kernel to its last copy (phases of different
// contexts would only fight over HBM bandwidth), so nothing in here is live outside the lock and
// 1024^3 jobs cost 2 x 8.6 GB of HBM each (field in, field out) instead of 5 x.
struct DevPool {
    std::mutex phase;
    int users = 0;
    double* scratch = nullptr; size_t scratch_elems = 0;  // coefficient array (out-of-place fused transform)
    uint8_t* planes = nullptr; size_t planes_bytes = 0;   // quantized planes
    double* lowbuf = nullptr; size_t lowbuf_elems = 0;    // compact low-pass boxes (fused transform)
    uint16_t* hist = nullptr; size_t hist_elems = 0;      // per-block byte histograms, all planes
};

struct wr_ctx {
    int device = 0;
    DevPool* pool = nullptr;
    hipStream_t stream = nullptr, copy = nullptr;
    bool own_stream = false;
    bool keep_residual = false;
    double* d_field = nullptr; size_t field_elems = 0;  // staging for the host-pointer API
    double* d_cutoff = nullptr; size_t cutoff_elems = 0;  // local cutoff vector (mx*my*mz > 1 only)
    double* d_partial = nullptr; double* d_result = nullptr;
    unsigned long long* d_idx = nullptr;
    // pinned host
    double* h_result = nullptr;  // [0..1] min/max, [2] probe value, [3] index
    uint8_t* h_plane[WR_NLAYMAX] = {nullptr}; size_t h_plane_bytes[WR_NLAYMAX] = {0};  // pinned, one per plane, on demand
    bool h_plane_pinned[WR_NLAYMAX] = {false};
    uint16_t* h_hist = nullptr; size_t h_hist_elems = 0;  // pinned: per-block byte histograms, all planes
    // host coded-stream staging, one per plane
    uint8_t* enc_buf[WR_NLAYMAX] = {nullptr}; size_t enc_buf_bytes[WR_NLAYMAX] = {0};  // malloc'd: only coded bytes get touched
    hipEvent_t ev_plane[WR_NLAYMAX], ev_copy[WR_NLAYMAX], ev_a, ev_b, ev_c, ev_d;
    std::mutex mu;
};

namespace {

int ctx_bind(wr_ctx* c) { HIPCHK(hipSetDevice(c->device)); return WR_OK; }

constexpr int kMaxDevices = 64;
DevPool g_pools[kMaxDevices];
std::mutex g_pools_mu;

using PhaseLock = std::unique_lock<std::mutex>;

// pool buffers: call with pool->phase held
int ensure_scratch(wr_ctx* c, size_t n)
{
    DevPool* p = c->pool;
    if (p->scratch_elems >= n) return WR_OK;
    if (p->scratch) HIPCHK(hipFree(p->scratch));
    p->scratch = nullptr; p->scratch_elems = 0;
    HIPCHK(hipMalloc(&p->scratch, n * sizeof(double)));
    p->scratch_elems = n;
    return WR_OK;
}

int ensure_planes(wr_ctx* c, size_t bytes)
{
    DevPool* p = c->pool;
    if (p->planes_bytes >= bytes) return WR_OK;
    if (p->planes) HIPCHK(hipFree(p->planes));
    p->planes = nullptr; p->planes_bytes = 0;
    HIPCHK(hipMalloc(&p->planes, bytes));
    p->planes_bytes = bytes;
    return WR_OK;
}

// pinned staging of plane l, allocated the first time a field needs that many planes (a 1024^3
// field at tol 1e-3 needs 3 GiB here, not 8)
int ensure_host_plane(wr_ctx* c, int l, size_t bytes)
{
    if (c->h_plane_bytes[l] >= bytes) return WR_OK;
    if (c->h_plane[l]) { if (c->h_plane_pinned[l]) HIPCHK(hipHostFree(c->h_plane[l])); else free(c->h_plane[l]); }
    c->h_plane[l] = nullptr; c->h_plane_bytes[l] = 0;
    if (hipHostMalloc(&c->h_plane[l], bytes, hipHostMallocDefault) == hipSuccess) {
        c->h_plane_pinned[l] = true;
    } else {
        // no pinned memory left (many contexts of many ranks on one host): pageable staging works,
        // the copy is then staged by the runtime and slower
        (void)hipGetLastError();
        c->h_plane[l] = static_cast<uint8_t*>(aligned_alloc(4096, (bytes + 4095) / 4096 * 4096));
        c->h_plane_pinned[l] = false;
        if (!c->h_plane[l]) return fail(WR_ERR_ARG, "out of host memory for the plane staging buffer");
    }
    c->h_plane_bytes[l] = bytes;
    return WR_OK;
}

int ensure_enc_buf(wr_ctx* c, int l, size_t bytes)
{
    if (c->enc_buf_bytes[l] >= bytes) return WR_OK;
    free(c->enc_buf[l]);
    c->enc_buf[l] = static_cast<uint8_t*>(malloc(bytes));
    c->enc_buf_bytes[l] = c->enc_buf[l] ? bytes : 0;
    return c->enc_buf[l] ? WR_OK : fail(WR_ERR_ARG, "out of host memory for the coded stream");
}

int ensure_hist(wr_ctx* c, size_t elems)
{
    DevPool* p = c->pool;
    if (p->hist_elems < elems) {
        if (p->hist) HIPCHK(hipFree(p->hist));
        p->hist = nullptr; p->hist_elems = 0;
        HIPCHK(hipMalloc(&p->hist, elems * sizeof(uint16_t)));
        p->hist_elems = elems;
    }
    if (c->h_hist_elems < elems) {
        if (c->h_hist) HIPCHK(hipHostFree(c->h_hist));
        c->h_hist = nullptr; c->h_hist_elems = 0;
        HIPCHK(hipHostMalloc(&c->h_hist, elems * sizeof(uint16_t), hipHostMallocDefault));
        c->h_hist_elems = elems;
    }
    return WR_OK;
}

int ensure_lowbuf(wr_ctx* c, size_t n)
{
    DevPool* p = c->pool;
    if (p->lowbuf_elems >= n) return WR_OK;
    if (p->lowbuf) HIPCHK(hipFree(p->lowbuf));
    p->lowbuf = nullptr; p->lowbuf_elems = 0;
    HIPCHK(hipMalloc(&p->lowbuf, n * sizeof(double)));
    p->lowbuf_elems = n;
    return WR_OK;
}

// Forward transform of d_fld.  The fused path is out of place: the coefficients land in the
// context's scratch buffer and *coef points there; the generic path works in place.
int forward_transform(wr_ctx* c, double* d_fld, int nx, int ny, int nz, int lvl, double** coef)
{
    const size_t n = (size_t)nx * ny * nz;
    if (int rc = ensure_scratch(c, n)) return rc;
    *coef = d_fld;
    if (wrk::fused_ok(nx, ny, nz, lvl) && !getenv("WR_NO_FUSED")) {
        if (int rc = ensure_lowbuf(c, wrk::fused_lowbuf_elems(nx, ny, nz))) return rc;
        if (lvl > 0) wrk::transform_fwd_fused(d_fld, c->pool->scratch, c->pool->lowbuf, nx, ny, nz, c->stream);
        else wrk::transform_inv_fused(d_fld, c->pool->scratch, c->pool->lowbuf, nx, ny, nz, c->stream);
        *coef = c->pool->scratch;
    } else {
        wrk::transform(d_fld, c->pool->scratch, nx, ny, nz, lvl, c->stream);
    }
    return WR_OK;
}

// decoder back end: acc = sum of planes, then the inverse transform, result in d_fld.
// Fused path: accumulate into the scratch buffer and transform out of place into d_fld.
// Records ev_a / ev_b / ev_c around the two stages (for the timings) when tm is given.
int inverse_from_planes(wr_ctx* c, double* d_fld, int nx, int ny, int nz, int wlev, const wrk::DequantParams& p,
                        wr_timings* tm)
{
    const size_t n = (size_t)nx * ny * nz;
    if (int rc = ensure_scratch(c, n)) return rc;
    const bool fused = wlev == 4 && wrk::fused_ok(nx, ny, nz, -4) && !getenv("WR_NO_FUSED");
    if (fused) if (int rc = ensure_lowbuf(c, wrk::fused_lowbuf_elems(nx, ny, nz))) return rc;
    if (tm) HIPCHK(hipEventRecord(c->ev_a, c->stream));
    wrk::dequant_accum(fused ? c->pool->scratch : d_fld, n, p, c->stream);
    if (tm) HIPCHK(hipEventRecord(c->ev_b, c->stream));
    if (fused) wrk::transform_inv_fused(c->pool->scratch, d_fld, c->pool->lowbuf, nx, ny, nz, c->stream);
    else wrk::transform(d_fld, c->pool->scratch, nx, ny, nz, -wlev, c->stream);
    if (tm) HIPCHK(hipEventRecord(c->ev_c, c->stream));
    return WR_OK;
}

int ensure_field(wr_ctx* c, size_t n)
{
    if (c->field_elems >= n) return WR_OK;
    if (c->d_field) HIPCHK(hipFree(c->d_field));
    c->d_field = nullptr; c->field_elems = 0;
    HIPCHK(hipMalloc(&c->d_field, n * sizeof(double)));
    c->field_elems = n;
    return WR_OK;
}

// min/max of a device array with the reference's scan semantics (wrappers.cpp:244-250):
// values from the reduction; if the minimum is a zero, its sign is that of the LAST zero in
// memory order (glibc fmin keeps the later of equal operands -- oracle/wr_oracle.c:wro_minmax).
// `pending` = the reduction has already been enqueued into d_result by a fused kernel.
int read_minmax(wr_ctx* c, const double* d_x, size_t n, bool pending, double* mn, double* mx)
{
    if (!pending) wrk::minmax(d_x, n, c->d_partial, c->d_result, c->stream);
    HIPCHK(hipMemcpyAsync(c->h_result, c->d_result, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    double lo = c->h_result[0], hi = c->h_result[1];
    if (lo == 0.0) {
        wrk::last_zero_index(d_x, n, c->d_idx, c->stream);
        unsigned long long* hidx = reinterpret_cast<unsigned long long*>(c->h_result + 3);
        HIPCHK(hipMemcpyAsync(hidx, c->d_idx, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (*hidx) {
            HIPCHK(hipMemcpyAsync(c->h_result + 2, d_x + (*hidx - 1), sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            lo = c->h_result[2];
        }
    }
    *mn = lo; *mx = hi;
    return WR_OK;
}

struct PlaneStep {
    double deps, minval, aopt, bopt;
    bool last;
};

// scalar side of one quantizer iteration, wrappers.cpp:316-340
PlaneStep plane_step(double lo, double hi, double tolabs, unsigned ilay)
{
    PlaneStep s;
    s.minval = lo;
    s.deps = (hi - lo) / (double)(256 - 1);
    s.last = false;
    if (s.deps < tolabs) { s.deps = tolabs; s.last = true; }
    if (ilay >= WR_NLAYMAX - 1u) s.last = true;
    s.aopt = 1.0 / s.deps;
    s.bopt = -lo * s.aopt + 0.5;
    return s;
}

// what the prologue of encoding_wrap computes, wrappers.cpp:235-266, 292-299
struct Prologue {
    bool trivial;
    double lo, hi;
};

int prologue(wr_ctx* c, const double* d_fld, size_t n, int wtflag, wr_enc_info* info, Prologue* p)
{
    memset(info, 0, sizeof(*info));
    info->wlev = wtflag ? kWavLvl : 0;
    int rc = read_minmax(c, d_fld, n, false, &p->lo, &p->hi);
    if (rc) return rc;
    if (p->lo != p->lo || p->hi != p->hi) return fail(WR_ERR_ARG, "field is all NaN");
    info->halfspanval = (p->hi - p->lo) / 2;
    info->midval = p->lo + info->halfspanval;
    p->trivial = info->halfspanval <= 2 * DBL_MIN;
    return WR_OK;
}

double abs_tolerance(double tolrel, const Prologue& p)
{
    double tolabs = tolrel * fmax(fabs(p.lo), fabs(p.hi));
    tolabs /= kWavAccCoef;
    return tolabs;
}

int check_dims(int nx, int ny, int nz, const void* ptr)
{
    if (nx < 1 || ny < 1 || nz < 1) return fail(WR_ERR_ARG, "non-positive dimension");
    if (ny > 65535 || nz > 65535) return fail(WR_ERR_UNSUPPORTED, "ny, nz must be <= 65535");
    if (((uintptr_t)ptr) & 15) return fail(WR_ERR_ARG, "device field pointer must be 16-byte aligned");
    return WR_OK;
}

struct Sem {  // tiny counting semaphore limiting concurrent range-coder threads
    std::mutex m; std::condition_variable cv; int n;
    explicit Sem(int k) : n(k) {}
    void acquire() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return n > 0; }); n--; }
    void release() { { std::lock_guard<std::mutex> l(m); n++; } cv.notify_one(); }
};

}  // namespace

// =====================================================================================
// Part 2: device-resident API
// =====================================================================================
extern "C" {

const char* wr_last_error(void) { return g_err.c_str(); }

int wr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void wr_set_verbosity(int level) { g_verbose = level ? 1 : 0; }
void wr_set_threads(int nthreads) { g_threads = nthreads < 1 ? 1 : nthreads; g_enc_threads = 0; }
void wr_set_encoder_threads(int nthreads) { g_enc_threads = nthreads < 0 ? 0 : nthreads; }

int wr_ctx_create(wr_ctx** out, int device, void* hip_stream)
{
    if (!out) return fail(WR_ERR_ARG, "null ctx pointer");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail(WR_ERR_HIP, std::string("no usable HIP device (") + hipGetErrorString(e) +
                                    "): libwaverange_amd has no CPU fallback");
    if (device < 0 || device >= ndev || device >= kMaxDevices) return fail(WR_ERR_ARG, "device index out of range");
    wr_ctx* c = new wr_ctx;
    c->device = device;
    c->pool = &g_pools[device];
    { std::lock_guard<std::mutex> lk(g_pools_mu); c->pool->users++; }
    HIPCHK(hipSetDevice(device));
    if (hip_stream) c->stream = (hipStream_t)hip_stream;
    else { HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    HIPCHK(hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
    HIPCHK(hipMalloc(&c->d_partial, 2 * wrk::minmax_partials() * sizeof(double)));
    HIPCHK(hipMalloc(&c->d_result, 4 * sizeof(double)));
    HIPCHK(hipMalloc(&c->d_idx, sizeof(unsigned long long)));
    HIPCHK(hipHostMalloc(&c->h_result, 8 * sizeof(double), hipHostMallocDefault));
    for (int i = 0; i < WR_NLAYMAX; i++) {
        HIPCHK(hipEventCreateWithFlags(&c->ev_plane[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_copy[i], hipEventDisableTiming));
    }
    HIPCHK(hipEventCreate(&c->ev_a)); HIPCHK(hipEventCreate(&c->ev_b));
    HIPCHK(hipEventCreate(&c->ev_c)); HIPCHK(hipEventCreate(&c->ev_d));
    *out = c;
    return WR_OK;
}

void wr_ctx_destroy(wr_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->copy);
    (void)hipFree(c->d_field); (void)hipFree(c->d_cutoff);
    (void)hipFree(c->d_partial); (void)hipFree(c->d_result); (void)hipFree(c->d_idx);
    (void)hipHostFree(c->h_result); (void)hipHostFree(c->h_hist);
    for (int l = 0; l < WR_NLAYMAX; l++) {
        if (c->h_plane[l]) { if (c->h_plane_pinned[l]) (void)hipHostFree(c->h_plane[l]); else free(c->h_plane[l]); }
        free(c->enc_buf[l]);
    }
    for (int i = 0; i < WR_NLAYMAX; i++) { (void)hipEventDestroy(c->ev_plane[i]); (void)hipEventDestroy(c->ev_copy[i]); }
    (void)hipEventDestroy(c->ev_a); (void)hipEventDestroy(c->ev_b);
    (void)hipEventDestroy(c->ev_c); (void)hipEventDestroy(c->ev_d);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    (void)hipStreamDestroy(c->copy);
    {   // the last context on a device releases the shared work space
        std::lock_guard<std::mutex> lk(g_pools_mu);
        DevPool* p = c->pool;
        if (--p->users == 0) {
            std::lock_guard<std::mutex> ph(p->phase);
            (void)hipFree(p->scratch); (void)hipFree(p->planes); (void)hipFree(p->lowbuf); (void)hipFree(p->hist);
            p->scratch = nullptr; p->planes = nullptr; p->lowbuf = nullptr; p->hist = nullptr;
            p->scratch_elems = p->planes_bytes = p->lowbuf_elems = p->hist_elems = 0;
        }
    }
    delete c;
}

int wr_ctx_sync(wr_ctx* c)
{
    if (int rc = ctx_bind(c)) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(c->copy));
    return WR_OK;
}

void wr_ctx_set_keep_residual(wr_ctx* c, int keep) { c->keep_residual = keep != 0; }

int wr_dev_alloc(wr_ctx* c, void** ptr, size_t bytes)
{
    if (int rc = ctx_bind(c)) return rc;
    HIPCHK(hipMalloc(ptr, bytes ? bytes : 16));
    return WR_OK;
}

int wr_dev_free(wr_ctx* c, void* ptr)
{
    if (int rc = ctx_bind(c)) return rc;
    HIPCHK(hipFree(ptr));
    return WR_OK;
}

int wr_dev_upload(wr_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (int rc = ctx_bind(c)) return rc;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return WR_OK;
}

int wr_dev_download(wr_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (int rc = ctx_bind(c)) return rc;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return WR_OK;
}

int wr_dev_copy(wr_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (int rc = ctx_bind(c)) return rc;
    PhaseLock ph(c->pool->phase);  // a device phase like any other: keeps it off other contexts' transforms
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return WR_OK;
}

int wr_dev_linf(wr_ctx* c, const double* d_a, const double* d_b, size_t n, double* max_abs_diff, double* max_abs_a)
{
    if (int rc = ctx_bind(c)) return rc;
    if (!n) return fail(WR_ERR_ARG, "empty array");
    wrk::linf_diff(d_a, d_b, n, c->d_partial, c->d_result, c->stream);
    HIPCHK(hipMemcpyAsync(c->h_result, c->d_result, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *max_abs_diff = c->h_result[0];
    *max_abs_a = c->h_result[1];
    return WR_OK;
}

int wr_dev_transform(wr_ctx* c, double* d_fld, int nx, int ny, int nz, int lvl)
{
    if (int rc = ctx_bind(c)) return rc;
    if (int rc = check_dims(nx, ny, nz, d_fld)) return rc;
    double* coef = nullptr;
    PhaseLock ph(c->pool->phase);
    if (int rc = forward_transform(c, d_fld, nx, ny, nz, lvl, &coef)) return rc;  // handles lvl < 0 too
    if (coef != d_fld)
        HIPCHK(hipMemcpyAsync(d_fld, coef, (size_t)nx * ny * nz * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));  // the shared work space is released with the lock
    return WR_OK;
}

int wr_dev_minmax(wr_ctx* c, const double* d_x, size_t n, double* mn, double* mx)
{
    if (int rc = ctx_bind(c)) return rc;
    if (!n) return fail(WR_ERR_ARG, "empty array");
    return read_minmax(c, d_x, n, false, mn, mx);
}

int wr_dev_quantize_plane(wr_ctx* c, double* d_x, size_t n, double deps, double minval, unsigned char* d_q,
                          double* next_min, double* next_max)
{
    if (int rc = ctx_bind(c)) return rc;
    if (((uintptr_t)d_x & 15) || ((uintptr_t)d_q & 1)) return fail(WR_ERR_ARG, "misaligned device pointer");
    const double aopt = 1.0 / deps;
    const double bopt = -minval * aopt + 0.5;
    wrk::quantize_plane(d_x, n, aopt, bopt, deps, minval, d_q, true, c->d_partial, c->d_result, c->stream);
    HIPCHK(hipGetLastError());
    return read_minmax(c, d_x, n, true, next_min, next_max);
}

int wr_dev_dequant_accum(wr_ctx* c, double* d_acc, size_t n, int nlay, const unsigned char* const* d_planes,
                         const double* deps, const double* minval)
{
    if (int rc = ctx_bind(c)) return rc;
    if (nlay < 0 || nlay > WR_NLAYMAX) return fail(WR_ERR_ARG, "nlay out of range");
    wrk::DequantParams p;
    memset(&p, 0, sizeof p);
    p.nlay = nlay;
    for (int l = 0; l < nlay; l++) { p.q[l] = d_planes[l]; p.deps[l] = deps[l]; p.minval[l] = minval[l]; }
    wrk::dequant_accum(d_acc, n, p, c->stream);
    HIPCHK(hipGetLastError());
    return WR_OK;
}

int wr_dev_synth_field(wr_ctx* c, double* d_out, int nx, int ny, int nz, unsigned long long seed)
{
    if (int rc = ctx_bind(c)) return rc;
    wrk::synth_field(d_out, nx, ny, nz, seed, 0, nz, c->stream);
    HIPCHK(hipGetLastError());
    return WR_OK;
}

size_t wr_plane_pitch(size_t n) { return (n + 255) & ~(size_t)255; }

}  // extern "C"

namespace {

// Device part of the encoder.  on_plane(l, last) is called right after plane l's kernel has
// been enqueued and ev_plane[l] recorded (the full pipeline hooks its D2H + coder thread in).
// local cutoff description (mx*my*mz == 1: uniform cutoff, the benchmark path)
struct Cutoff {
    int mx = 1, my = 1, mz = 1;
    const double* vec = nullptr;  // host, mx*my*mz entries
    int count() const { return mx * my * mz; }
};

template <class OnPlane>
int encode_planes_core(wr_ctx* c, double* d_fld, int nx, int ny, int nz, int wtflag, const Cutoff& cut,
                       uint8_t* d_planes, wr_enc_info* info, wr_timings* tm, OnPlane on_plane)
{
    // minimum cutoff = the global relative tolerance (wrappers.cpp:288-290)
    double tolrel = cut.vec[0];
    for (int k = 1; k < cut.count(); k++) if (cut.vec[k] < tolrel) tolrel = cut.vec[k];
    const bool local = cut.count() > 1;
    if (local) {
        if (c->cutoff_elems < (size_t)cut.count()) {
            if (c->d_cutoff) HIPCHK(hipFree(c->d_cutoff));
            c->d_cutoff = nullptr; c->cutoff_elems = 0;
            HIPCHK(hipMalloc(&c->d_cutoff, cut.count() * sizeof(double)));
            c->cutoff_elems = cut.count();
        }
        HIPCHK(hipMemcpyAsync(c->d_cutoff, cut.vec, cut.count() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    const size_t n = (size_t)nx * ny * nz;
    const size_t pitch = wr_plane_pitch(n);
    if (int rc = ensure_scratch(c, n)) return rc;
    Prologue p;
    HIPCHK(hipEventRecord(c->ev_a, c->stream));
    if (int rc = prologue(c, d_fld, n, wtflag, info, &p)) return rc;
    if (verbose()) printf("Wavelet decomposition...\n");
    if (p.trivial) {  // wrappers.cpp:256-266
        info->ntot_enc = 0; info->nlay = 0; info->tolabs = 0;
        return WR_OK;
    }
    HIPCHK(hipEventRecord(c->ev_b, c->stream));
    double* const d_in = d_fld;
    if (int rc = forward_transform(c, d_in, nx, ny, nz, (int)info->wlev, &d_fld)) return rc;  // d_fld := coefficients
    HIPCHK(hipEventRecord(c->ev_c, c->stream));
    if (verbose()) printf("Range encoding...\n");
    info->tolabs = abs_tolerance(tolrel, p);

    double lo, hi;
    if (int rc = read_minmax(c, d_fld, n, false, &lo, &hi)) return rc;
    HIPCHK(hipEventRecord(c->ev_d, c->stream));
    float ms = 0;
    HIPCHK(hipEventSynchronize(c->ev_d));
    if (tm) {
        HIPCHK(hipEventElapsedTime(&ms, c->ev_b, c->ev_c)); tm->transform_ms = ms;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_a, c->ev_b)); tm->minmax_ms = ms;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_c, c->ev_d)); tm->minmax_ms += ms;
    }
    unsigned ilay = 0;
    float quant_ms = 0;
    for (;;) {
        PlaneStep s = plane_step(lo, hi, info->tolabs, ilay);
        info->minval_vec[ilay] = s.minval;
        info->deps_vec[ilay] = s.deps;
        if (verbose()) { printf("min=%g max=%g\n", lo, hi); printf("ilay=%u deps=%g\n", ilay, s.deps); }
        const bool resid = !s.last || c->keep_residual;
        HIPCHK(hipEventRecord(c->ev_a, c->stream));
        if (local) {
            wrk::LocalCutoff lc;
            lc.nx = nx; lc.ny = ny; lc.nz = nz; lc.wlev = info->wlev;
            lc.mx = cut.mx; lc.my = cut.my; lc.mz = cut.mz;
            lc.cutoff = c->d_cutoff;
            lc.tol_scale = info->tolabs / tolrel;
            lc.tolabs = info->tolabs;
            lc.span = hi - lo;
            wrk::quantize_plane_local(d_fld, n, s.aopt, s.bopt, s.deps, s.minval, d_planes + ilay * pitch, lc,
                                      c->d_partial, c->d_result, c->stream);
        } else
        wrk::quantize_plane(d_fld, n, s.aopt, s.bopt, s.deps, s.minval, d_planes + ilay * pitch, resid,
                            c->d_partial, c->d_result, c->stream);
        HIPCHK(hipEventRecord(c->ev_b, c->stream));
        HIPCHK(hipEventRecord(c->ev_plane[ilay], c->stream));
        HIPCHK(hipGetLastError());
        if (int rc = on_plane(ilay, s.last)) return rc;
        ilay++;
        if (s.last) {
            HIPCHK(hipEventSynchronize(c->ev_b));
            HIPCHK(hipEventElapsedTime(&ms, c->ev_a, c->ev_b)); quant_ms += ms;
            break;
        }
        if (int rc = read_minmax(c, d_fld, n, true, &lo, &hi)) return rc;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_a, c->ev_b)); quant_ms += ms;
    }
    info->nlay = (unsigned char)ilay;
    if (tm) tm->quant_ms = quant_ms;
    if (c->keep_residual && d_fld != d_in)  // leave the residual where the reference leaves it
        HIPCHK(hipMemcpyAsync(d_in, d_fld, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return WR_OK;
}

}  // namespace

extern "C" {

int wr_dev_encode_planes(wr_ctx* c, double* d_fld, int nx, int ny, int nz, int wtflag, double tolrel,
                         unsigned char* d_planes, wr_enc_info* info)
{
    if (int rc = ctx_bind(c)) return rc;
    if (int rc = check_dims(nx, ny, nz, d_fld)) return rc;
    if ((uintptr_t)d_planes & 15) return fail(WR_ERR_ARG, "plane buffer must be 16-byte aligned");
    std::lock_guard<std::mutex> lk(c->mu);
    PhaseLock ph(c->pool->phase);
    Cutoff cut; cut.vec = &tolrel;
    int rc = encode_planes_core(c, d_fld, nx, ny, nz, wtflag, cut, d_planes, info, nullptr,
                                [](unsigned, bool) { return WR_OK; });
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return WR_OK;
}

int wr_dev_decode_planes(wr_ctx* c, double* d_fld, int nx, int ny, int nz, const unsigned char* d_planes,
                         const wr_enc_info* info)
{
    if (int rc = ctx_bind(c)) return rc;
    if (int rc = check_dims(nx, ny, nz, d_fld)) return rc;
    const size_t n = (size_t)nx * ny * nz;
    std::lock_guard<std::mutex> lk(c->mu);
    if (info->ntot_enc == 0 && info->nlay == 0) {  // trivial field, wrappers.cpp:462-469
        wrk::fill(d_fld, n, info->midval, c->stream);
        HIPCHK(hipStreamSynchronize(c->stream));
        return WR_OK;
    }
    if (info->nlay > WR_NLAYMAX) return fail(WR_ERR_ARG, "nlay out of range");
    PhaseLock ph(c->pool->phase);
    if (int rc = ensure_scratch(c, n)) return rc;
    wrk::DequantParams p;
    memset(&p, 0, sizeof p);
    p.nlay = info->nlay;
    for (int l = 0; l < p.nlay; l++) {
        p.q[l] = d_planes + l * wr_plane_pitch(n);
        p.deps[l] = info->deps_vec[l];
        p.minval[l] = info->minval_vec[l];
    }
    if (int rc = inverse_from_planes(c, d_fld, nx, ny, nz, (int)info->wlev, p, nullptr)) return rc;
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return WR_OK;
}

static int encode_device_impl(wr_ctx* c, double* d_fld, int nx, int ny, int nz, int wtflag, const Cutoff& cut,
                              wr_enc_info* info, unsigned char* data_enc, size_t cap, wr_timings* tm);

int wr_encode_device(wr_ctx* c, double* d_fld, int nx, int ny, int nz, int wtflag, double tolrel,
                     wr_enc_info* info, unsigned char* data_enc, size_t cap, wr_timings* tm)
{
    Cutoff cut; cut.vec = &tolrel;
    return encode_device_impl(c, d_fld, nx, ny, nz, wtflag, cut, info, data_enc, cap, tm);
}

int wr_encode_device_local(wr_ctx* c, double* d_fld, int nx, int ny, int nz, int wtflag, int mx, int my, int mz,
                           const double* cutoffvec, wr_enc_info* info, unsigned char* data_enc, size_t cap,
                           wr_timings* tm)
{
    if (mx < 1 || my < 1 || mz < 1 || !cutoffvec) return fail(WR_ERR_ARG, "bad local cutoff description");
    Cutoff cut; cut.mx = mx; cut.my = my; cut.mz = mz; cut.vec = cutoffvec;
    return encode_device_impl(c, d_fld, nx, ny, nz, wtflag, cut, info, data_enc, cap, tm);
}

static int encode_device_impl(wr_ctx* c, double* d_fld, int nx, int ny, int nz, int wtflag, const Cutoff& cut,
                              wr_enc_info* info, unsigned char* data_enc, size_t cap, wr_timings* tm)
{
    if (int rc = ctx_bind(c)) return rc;
    if (int rc = check_dims(nx, ny, nz, d_fld)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    const double t0 = now();
    const size_t n = (size_t)nx * ny * nz;
    const size_t pitch = wr_plane_pitch(n);
    wr_timings local; memset(&local, 0, sizeof local);
    // per-60000-symbol-block byte histograms, counted on the GPU next to the quantizer and shipped
    // with the plane, so that the host coder starts every block with its model ready
    const size_t hist_per_plane = (n / wrrc::kBlock + 1) * 256;

    std::vector<std::thread> workers;
    size_t lens[WR_NLAYMAX] = {0};
    double coder_s[WR_NLAYMAX] = {0};
    std::string logs[WR_NLAYMAX];
    Sem sem(encoder_threads());
    const int dev = c->device;
    double t_gpu_done = 0;

    // With a coder thread for every possible plane, plane l's thread starts as soon as the plane is
    // on the host.  With fewer (wr_set_threads), the planes are split into that many groups once
    // their number is known and each thread codes its group with the symbol loops interleaved.
    const bool per_plane = encoder_threads() >= WR_NLAYMAX;
    auto code_group = [&](unsigned l0, unsigned l1) {
        (void)hipSetDevice(dev);
        (void)hipEventSynchronize(c->ev_copy[l1 - 1]);  // copies complete in plane order
        sem.acquire();
        const double t = now();
        const uint8_t* syms[WR_NLAYMAX];
        uint8_t* outs[WR_NLAYMAX];
        const uint16_t* hs[WR_NLAYMAX];
        for (unsigned l = l0; l < l1; l++) { syms[l - l0] = c->h_plane[l]; outs[l - l0] = c->enc_buf[l]; hs[l - l0] = c->h_hist + l * hist_per_plane; }
        wrrc::encode_planes((int)(l1 - l0), syms, n, outs, hs, lens + l0);
        for (unsigned l = l0; l < l1; l++) coder_s[l] = now() - t;
        sem.release();
        if (verbose())  // wrappers.cpp:401-409, 430
            for (unsigned l = l0; l < l1; l++) {
                const uint8_t* q = c->h_plane[l];
                unsigned lo = q[0], hi = q[0];
                for (size_t j = 1; j < n; j++) { unsigned v = q[j]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
                char b[256];
                snprintf(b, sizeof b, "imin=%u imax=%u med=%g\nlen_out_q=%lu ntot=%lu\n", lo, hi,
                         q[n / 2] * info->deps_vec[l] + info->minval_vec[l], (unsigned long)lens[l], (unsigned long)n);
                logs[l] = b;
            }
    };
    auto on_plane = [&](unsigned l, bool) -> int {
        // plane l: device -> pinned host on the copy stream
        HIPCHK(hipStreamWaitEvent(c->copy, c->ev_plane[l], 0));
        wrk::block_histograms(c->pool->planes + l * pitch, n, c->pool->hist + l * hist_per_plane, c->copy);
        HIPCHK(hipMemcpyAsync(c->h_hist + l * hist_per_plane, c->pool->hist + l * hist_per_plane, hist_per_plane * sizeof(uint16_t),
                              hipMemcpyDeviceToHost, c->copy));
        if (int rc = ensure_host_plane(c, (int)l, pitch)) return rc;
        if (int rc = ensure_enc_buf(c, (int)l, wrrc::encode_bound(n))) return rc;
        HIPCHK(hipMemcpyAsync(c->h_plane[l], c->pool->planes + l * pitch, n, hipMemcpyDeviceToHost, c->copy));
        HIPCHK(hipEventRecord(c->ev_copy[l], c->copy));
        if (per_plane) workers.emplace_back(code_group, l, l + 1);
        return WR_OK;
    };
    int rc;
    double t_phase = 0;
    {
        PhaseLock gpu(c->pool->phase);
        t_phase = now();
        rc = ensure_planes(c, pitch * WR_NLAYMAX);
        if (!rc) rc = ensure_hist(c, hist_per_plane * WR_NLAYMAX);
        if (!rc) rc = encode_planes_core(c, d_fld, nx, ny, nz, wtflag, cut, c->pool->planes, info, &local, on_plane);
        // The phase ends when the planes are on the host: the plane buffer is shared with the other
        // contexts on this device, and with many contexts in a process the runtime runs the copies as blit kernels, which are
        // better kept off other contexts' transforms (tens of ms; coder threads start per plane regardless)
        (void)hipStreamSynchronize(c->copy);
        (void)hipStreamSynchronize(c->stream);
    }
    t_gpu_done = now();
    if (!per_plane && rc == WR_OK && info->nlay) {
        const unsigned groups = std::min<unsigned>(info->nlay, (unsigned)encoder_threads());
        for (unsigned g = 0; g < groups; g++)
            workers.emplace_back(code_group, g * info->nlay / groups, (g + 1) * info->nlay / groups);
    }
    for (auto& w : workers) w.join();
    if (rc) return rc;
    const double t_coded = now();
    // concatenate the plane streams (wrappers.cpp:412-427); gigabytes at 1024^3, so one copier per plane
    size_t total = 0, offs[WR_NLAYMAX] = {0};
    for (unsigned l = 0; l < info->nlay; l++) {
        offs[l] = total;
        total += lens[l];
        info->len_enc_vec[l] = lens[l];
        if (coder_s[l] > local.rangecoder) local.rangecoder = coder_s[l];
    }
    if (total > cap) return fail(WR_ERR_OVERFLOW, "Error: encoded array is too large. Use larger SAFETY_BUFFER_FACTOR");
    {
        std::vector<std::thread> copiers;
        for (unsigned l = 1; l < info->nlay; l++)
            copiers.emplace_back([&, l]() { memcpy(data_enc + offs[l], c->enc_buf[l], lens[l]); });
        if (info->nlay) memcpy(data_enc, c->enc_buf[0], lens[0]);
        for (auto& t : copiers) t.join();
    }
    if (verbose())
        for (unsigned l = 0; l < info->nlay; l++) fputs(logs[l].c_str(), stdout);
    info->ntot_enc = total;
    local.total = now() - t0;
    local.gpu = t_gpu_done - t_phase;  // without the wait for the device
    local.transfer = (t_coded - t_gpu_done) - local.rangecoder;
    if (local.transfer < 0) local.transfer = 0;
    if (tm) *tm = local;
    return WR_OK;
}

int wr_decode_device(wr_ctx* c, double* d_fld, int nx, int ny, int nz, const wr_enc_info* info,
                     const unsigned char* data_enc, wr_timings* tm)
{
    if (int rc = ctx_bind(c)) return rc;
    if (int rc = check_dims(nx, ny, nz, d_fld)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    const double t0 = now();
    const size_t n = (size_t)nx * ny * nz;
    const size_t pitch = wr_plane_pitch(n);
    wr_timings local; memset(&local, 0, sizeof local);
    if (info->ntot_enc == 0) {  // wrappers.cpp:462-469
        wrk::fill(d_fld, n, info->midval, c->stream);
        HIPCHK(hipStreamSynchronize(c->stream));
        local.total = now() - t0;
        if (tm) *tm = local;
        return WR_OK;
    }
    const int nlay = info->nlay;
    if (nlay < 1 || nlay > WR_NLAYMAX) return fail(WR_ERR_ARG, "nlay out of range");
    if (verbose()) printf("Range decoding...\n");
    for (int l = 0; l < nlay; l++) if (int rc = ensure_host_plane(c, l, pitch)) return rc;

    size_t off[WR_NLAYMAX + 1] = {0};
    for (int l = 0; l < nlay; l++) off[l + 1] = off[l] + info->len_enc_vec[l];
    if (off[nlay] > info->ntot_enc) return fail(WR_ERR_STREAM, "len_enc_vec exceeds ntot_enc");
    size_t got[WR_NLAYMAX] = {0};
    double coder_s[WR_NLAYMAX] = {0};
    std::vector<std::thread> workers;
    Sem sem(coder_threads());
    // one thread per plane, or (wr_set_threads) fewer threads with their planes interleaved
    const int groups = std::min(nlay, coder_threads());
    for (int g = 0; g < groups; g++)
        workers.emplace_back([&, g]() {
            const int l0 = g * nlay / groups, l1 = (g + 1) * nlay / groups;
            sem.acquire();
            const double t = now();
            const uint8_t* ins[WR_NLAYMAX];
            uint8_t* syms[WR_NLAYMAX];
            for (int l = l0; l < l1; l++) { ins[l - l0] = data_enc + off[l]; syms[l - l0] = c->h_plane[l]; }
            wrrc::decode_planes(l1 - l0, ins, info->len_enc_vec + l0, syms, n, got + l0);
            for (int l = l0; l < l1; l++) coder_s[l] = now() - t;
            sem.release();
        });
    for (auto& w : workers) w.join();
    int bad = -1;
    for (int l = 0; l < nlay; l++) {
        if (got[l] != n) bad = l;
        if (coder_s[l] > local.rangecoder) local.rangecoder = coder_s[l];
    }
    if (bad >= 0)
        return fail(WR_ERR_STREAM, "plane " + std::to_string(bad) + ": stream does not decode to nx*ny*nz symbols");
    const double t_coded = now();
    if (verbose()) {  // wrappers.cpp:489, 503-510
        for (int l = 0; l < nlay; l++) {
            const uint8_t* q = c->h_plane[l];
            unsigned lo = q[0], hi = q[0];
            for (size_t j = 1; j < n; j++) { unsigned v = q[j]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
            printf("ilay=%d\nimin=%u imax=%u med=%g\n", l, lo, hi, q[n / 2] * info->deps_vec[l] + info->minval_vec[l]);
        }
        printf("Wavelet reconstruction...\n");
    }
    wrk::DequantParams p;
    memset(&p, 0, sizeof p);
    p.nlay = nlay;
    for (int l = 0; l < nlay; l++) { p.deps[l] = info->deps_vec[l]; p.minval[l] = info->minval_vec[l]; }
    double t_phase = 0;
    {
        PhaseLock gpu(c->pool->phase);
        t_phase = now();
        if (int rc = ensure_planes(c, pitch * nlay)) return rc;
        if (int rc = ensure_scratch(c, n)) return rc;
        for (int l = 0; l < nlay; l++) p.q[l] = c->pool->planes + l * pitch;
        for (int l = 0; l < nlay; l++)  // planes: pinned host -> device
            HIPCHK(hipMemcpyAsync(c->pool->planes + l * pitch, c->h_plane[l], n, hipMemcpyHostToDevice, c->stream));
        if (int rc = inverse_from_planes(c, d_fld, nx, ny, nz, (int)info->wlev, p, &local)) return rc;
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev_a, c->ev_b)); local.quant_ms = ms;
    HIPCHK(hipEventElapsedTime(&ms, c->ev_b, c->ev_c)); local.transform_ms = ms;
    local.total = now() - t0;
    local.gpu = now() - t_phase;  // without the wait for the device
    local.transfer = (t_coded - t0) - local.rangecoder;
    if (local.transfer < 0) local.transfer = 0;
    if (tm) *tm = local;
    return WR_OK;
}

size_t wr_range_encode_bound(size_t n) { return wrrc::encode_bound(n); }
size_t wr_range_encode(const unsigned char* sym, size_t n, unsigned char* out) { return wrrc::encode_plane(sym, n, out, nullptr); }
size_t wr_range_decode(const unsigned char* in, size_t len, unsigned char* sym, size_t n) { return wrrc::decode_plane(in, len, sym, n); }
void wr_range_encode_multi(int count, const unsigned char* const* sym, size_t n, unsigned char* const* out, size_t* lens)
{
    wrrc::encode_planes(count, sym, n, out, nullptr, lens);
}
void wr_range_decode_multi(int count, const unsigned char* const* in, const size_t* len, unsigned char* const* sym, size_t n, size_t* produced)
{
    wrrc::decode_planes(count, in, len, sym, n, produced);
}

int wr_bench_transform(wr_ctx* c, double* d_fld, int nx, int ny, int nz, int lvl, int reps, double* ms_out)
{
    if (int rc = ctx_bind(c)) return rc;
    if (int rc = check_dims(nx, ny, nz, d_fld)) return rc;
    if (reps < 1) return fail(WR_ERR_ARG, "reps < 1");
    PhaseLock ph(c->pool->phase);
    if (int rc = ensure_scratch(c, (size_t)nx * ny * nz)) return rc;
    double* coef = nullptr;
    HIPCHK(hipEventRecord(c->ev_a, c->stream));
    for (int r = 0; r < reps; r++)
        if (int rc = forward_transform(c, d_fld, nx, ny, nz, lvl, &coef)) return rc;  // fused: result stays in scratch
    HIPCHK(hipEventRecord(c->ev_b, c->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventSynchronize(c->ev_b));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev_a, c->ev_b));
    *ms_out = (double)ms / reps;
    return WR_OK;
}

}  // extern "C"

// =====================================================================================
// Part 1: libwaverange drop-in symbols (host pointers)
// =====================================================================================
namespace {

wr_ctx* g_ctx = nullptr;
std::mutex g_ctx_mu;

[[noreturn]] void fatal(const char* where)
{
    fprintf(stderr, "libwaverange_amd: %s: %s\n", where, g_err.c_str());
    abort();
}

wr_ctx* default_ctx()
{
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (!g_ctx) {
        int dev = 0;
        if (const char* e = getenv("WR_DEVICE")) dev = atoi(e);
        if (wr_ctx_create(&g_ctx, dev, nullptr) != WR_OK) fatal("no GPU context");
    }
    return g_ctx;
}

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
    if (mx < 1 || my < 1 || mz < 1) { g_err = "mx, my, mz must be >= 1"; fatal("encoding_wrap"); }
    wr_ctx* c = default_ctx();
    const size_t n = (size_t)nx * ny * nz;
    unsigned char nl; unsigned long cap;
    setup_wr(nx, ny, nz, &nl, &cap);
    if (ctx_bind(c) || ensure_field(c, n) || wr_dev_upload(c, c->d_field, fld_1d, n * sizeof(double))) fatal("encoding_wrap");
    wr_enc_info info;
    const bool wb = getenv("WR_WRITEBACK_RESIDUAL") && atoi(getenv("WR_WRITEBACK_RESIDUAL"));
    c->keep_residual = wb;
    if (wr_encode_device_local(c, c->d_field, nx, ny, nz, wtflag, mx, my, mz, cutoffvec, &info, data_enc, cap, nullptr))
        fatal("encoding_wrap");
    if (wb && info.nlay && wr_dev_download(c, fld_1d, c->d_field, n * sizeof(double))) fatal("encoding_wrap");
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
    wr_ctx* c = default_ctx();
    const size_t n = (size_t)nx * ny * nz;
    wr_enc_info info;
    memset(&info, 0, sizeof info);
    info.midval = *midval; info.wlev = *wlev; info.nlay = *nlay; info.ntot_enc = *ntot_enc;
    if (info.nlay > WR_NLAYMAX) { g_err = "nlay > 8"; fatal("decoding_wrap"); }
    for (int l = 0; l < info.nlay; l++) {
        info.deps_vec[l] = deps_vec[l];
        info.minval_vec[l] = minval_vec[l];
        info.len_enc_vec[l] = len_enc_vec[l];
    }
    if (ctx_bind(c) || ensure_field(c, n)) fatal("decoding_wrap");
    if (wr_decode_device(c, c->d_field, nx, ny, nz, &info, data_enc, nullptr)) fatal("decoding_wrap");
    if (wr_dev_download(c, fld_1d, c->d_field, n * sizeof(double))) fatal("decoding_wrap");
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
    wr_ctx* c = default_ctx();
    const size_t n = (size_t)n1 * n2 * n3;
    if (ctx_bind(c) || ensure_field(c, n) || wr_dev_upload(c, c->d_field, x, n * sizeof(double)) ||
        wr_dev_transform(c, c->d_field, n1, n2, n3, lvl) || wr_dev_download(c, x, c->d_field, n * sizeof(double)))
        fatal("waveletcdf97_3d");
}

}  // extern "C"
