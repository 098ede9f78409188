// wr_kernels.h -- launchers for the gfx950 kernels of the WaveRange hot path.
//
// Everything here works on DEVICE pointers and enqueues on the given HIP stream; nothing
// synchronises.  Canonical arithmetic is strict IEEE double without FMA contraction
// (the library is compiled with -ffp-contract=off; see DESIGN.md "Arithmetic").
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace wrk {

// ---- 3-D CDF-9/7 lifting transform (reference src/waveletcdf97_3d/waveletcdf97_3d.c:38-468)
// In place on `fld` (nx*ny*nz doubles, x fastest); `scratch` is a second buffer of the same
// size.  lvl > 0 forward, lvl < 0 inverse, 0 identity.
void transform(double* fld, double* scratch, int nx, int ny, int nz, int lvl, hipStream_t st);
// one level of the above: the box ceil(n / 2^k), in place on `fld`
void transform_level(double* fld, double* scratch, int nx, int ny, int nz, int k, bool inverse, hipStream_t st);

// ---- reductions (reference src/core/wrappers.cpp:244-250, 308-314)
// partial[] needs 2*minmax_partials() doubles; result (min, max) lands in result[0..1] (device).
int minmax_partials();
void minmax(const double* x, size_t n, double* partial, double* result, hipStream_t st);
// sign bit of the LAST element equal to zero (the reference's fmin scan keeps the last of
// equal values): out[0] = index+1 of that element (0 if none).  Rare path (min == 0 only).
void last_zero_index(const double* x, size_t n, unsigned long long* out, hipStream_t st);

// A quantized plane as the kernels see it: one array, or up to kPlaneChunks chunks of 2^shift bytes each (a large plane
// of a call in flight lives in chunks that come and go as the host coder drains or fills it: wr_pipeline.cpp).  Byte i of
// the plane is chunk[i >> shift][i & (2^shift - 1)]; a chunk is a multiple of 4096 bytes, so the 4096-element groups the
// kernels work in, and their 16-byte loads, never straddle two chunks.
constexpr int kPlaneChunks = 16;
struct PlaneRef {
    uint8_t* chunk[kPlaneChunks];
    unsigned shift;
    __host__ __device__ uint8_t* at(size_t i) const { return chunk[i >> shift] + (i & (((size_t)1 << shift) - 1)); }
};
// every chunk that the bytes [0, n) of the plane fall into is there (host-side check in front of a launch: a kernel handed
// a table with a hole in it would write through a null or stale pointer)
inline bool plane_ref_covers(const PlaneRef& r, size_t n)
{
    if (!n) return true;
    const size_t last = r.shift >= 8 * sizeof(size_t) - 1 ? 0 : (n - 1) >> r.shift;
    if (last >= (size_t)kPlaneChunks) return false;
    for (size_t k = 0; k <= last; k++) if (!r.chunk[k]) return false;
    return true;
}
inline PlaneRef plane_ref(const uint8_t* base)  // a plane that is one array
{
    PlaneRef r;
    for (int k = 0; k < kPlaneChunks; k++) r.chunk[k] = const_cast<uint8_t*>(base);
    r.shift = 63;
    return r;
}

// ---- quantizer plane (wrappers.cpp:339-340, 384-398) fused with the min/max of the residual
// q[j] = (uchar)(aopt*x[j] + bopt); if write_resid: x[j] -= q[j]*deps + minval and
// result[0..1] = min/max of the new residual.
void quantize_plane(double* x, size_t n, double aopt, double bopt, double deps, double minval,
                    const PlaneRef& q, bool write_resid, double* partial, double* result, hipStream_t st);

// ---- the same plane without a residual array (k_quant_blk): x holds the COEFFICIENTS and stays as it is; the residual this
// plane is cut from is recomputed from them and the scalars of the planes before (prev, in order); want_minmax: result[0..1]
// = min/max of the residual after this plane; write_resid: x := that residual (the last plane of a caller who wants it);
// hist != nullptr: the per-60000-symbol-block byte histograms of the plane (block_histograms' output) are written too.
// quantize_plane_blk_ok: the pointers are 16-byte aligned (and WR_QUANT_INPLACE is not set).
constexpr int kQuantPrevMax = 7;
struct QuantPrev {
    int n = 0;
    double aopt[kQuantPrevMax], bopt[kQuantPrevMax], deps[kQuantPrevMax], minval[kQuantPrevMax];
    void push(double a, double b, double d, double m) { aopt[n] = a; bopt[n] = b; deps[n] = d; minval[n] = m; n++; }
};
bool quantize_plane_blk_ok(const double* x, const PlaneRef& q);
void quantize_plane_blk(double* x, size_t n, const QuantPrev& prev, double aopt, double bopt, double deps, double minval, const PlaneRef& q,
                        bool write_resid, bool want_minmax, uint16_t* hist, double* partial, double* result, hipStream_t st);
// x := the residual after the planes in prev
void residual_apply(double* x, size_t n, const QuantPrev& prev, hipStream_t st);

// ---- quantizer plane with the non-uniform (local) cutoff mask (wrappers.cpp:343-379, 397-398):
// loops over PHYSICAL positions, maps each to its wavelet-space index (ind_p2w_3d,
// waveletcdf97_3d.c:473-553) and quantizes there; coefficients of the finest level whose plane
// range is below the local precision are zeroed.  Residual update and its min/max are fused.
struct LocalCutoff {
    int nx, ny, nz, wlev;
    int mx, my, mz;
    const double* cutoff;  // device array, mx*my*mz entries
    double tol_scale;      // tolabs / tolrel
    double tolabs, span;   // span = maxval - minval of the plane
};
void quantize_plane_local(double* x, size_t n, double aopt, double bopt, double deps, double minval,
                          uint8_t* q, const LocalCutoff& lc, double* partial, double* result, hipStream_t st);

// ---- decoder accumulation (wrappers.cpp:480, 513-514): acc = 0; acc += q_l*deps_l + min_l, l in order
struct DequantParams {
    PlaneRef q[8];
    double deps[8];
    double minval[8];
    int nlay;
};
void dequant_accum(double* acc, size_t n, const DequantParams& p, hipStream_t st);

// ---- helpers
// result[0] = max|a-b|, result[1] = max|a|  (partial: 2*minmax_partials() doubles)
void linf_diff(const double* a, const double* b, size_t n, double* partial, double* result, hipStream_t st);
void fill(double* x, size_t n, double v, hipStream_t st);
// synthetic field of waverange_amd/synth.py, planes z0..z1-1, written at out[0..]
void synth_field(double* out, int nx, int ny, int nz, unsigned long long seed, int z0, int z1,
                 hipStream_t st);
// per-60000-symbol-block byte histograms of a plane (feeds the host range coder's model):
// hist[b*256 + v] = count of value v in block b (uint16, block size < 65536)
void block_histograms(const PlaneRef& q, size_t n, uint16_t* hist, hipStream_t st);
// keeps `workgroups` workgroups of 256 lanes on the device for `ms` milliseconds (mode 0: fp64 arithmetic, 1: asleep)
void burn(double ms, int mode, int workgroups, double* sink, hipStream_t st);
// bytes (a multiple of 16, 16-byte aligned pointers) copied by `workgroups` workgroups; src / dst may be pinned host memory
void copy_kernel(void* dst, const void* src, size_t bytes, int workgroups, hipStream_t st);

}  // namespace wrk

namespace wrk {
// ---- fused single-pass-per-level transform (wr_fused.hip).  The finest levels whose boxes are even
// in all three directions (and, for the inverse, whose x extent is a multiple of 4) run fused, the
// remaining coarser levels on the general kernels: fused_levels().  A 1024^3 field runs all four
// levels fused, 1000^3 three forward / two inverse, 500^3 two / one.
// Out of place: the result lands in `dst`; `src` is CONSUMED (the general levels use the forward's
// input array as scratch, the inverse rewrites the coarse corner of its coefficient array);
// `lowbuf` holds the compact low-pass boxes between fused levels (fused_lowbuf_elems() doubles).
int fused_levels(int nx, int ny, int nz, bool inverse);
bool fused_ok(int nx, int ny, int nz, int lvl);
size_t fused_lowbuf_elems(int nx, int ny, int nz);
// one-time set-up of the fused kernels (their dynamic LDS limit); nullptr, or why they cannot run on this device
const char* fused_prepare();
// mm_partial (fused_minmax_records() x 4 doubles, device) and mm_result (4 doubles, device): when given -- and all
// four levels run fused -- the forward kernels also reduce min/max of the field read and of the coefficient array
// written; mm_result = {field min, field max, coefficient min, coefficient max} (NaNs skipped, the sign of a zero
// minimum is not defined here: see read_minmax in wr_pipeline.cpp).
size_t fused_minmax_records(int nx, int ny, int nz);
void transform_fwd_fused(double* src, double* dst, double* lowbuf, int nx, int ny, int nz, hipStream_t st,
                         double* mm_partial = nullptr, double* mm_result = nullptr);
void transform_inv_fused(double* src, double* dst, double* lowbuf, int nx, int ny, int nz, hipStream_t st);
}  // namespace wrk
