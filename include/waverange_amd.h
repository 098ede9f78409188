/*
 * waverange_amd.h -- C ABI of libwaverange_amd.so: the MI355X (gfx950) implementation of
 * WaveRange's encode/decode hot path (3-D CDF-9/7 transform + bit-plane quantizer on the
 * GPU, rngcod13 range coder on the host).
 *
 * Part 1 are the drop-in entry points: the SAME unmangled symbols, argument order and
 * meaning as the reference's libwaverange (src/core/wrappers.h:53,70,75,95,111,119).  The
 * reference declares the scalar outputs as C++ references; at the SysV ABI level a
 * reference is a pointer, so the declarations below are call-compatible with code compiled
 * against the reference header.
 *
 * Part 2 is the device-resident API used by the tests, bench.py and multi-field callers:
 * plain pointers and sizes only, HIP stream handles passed as void*.
 *
 * Error convention: Part 1 keeps the reference's "void + fatal" behaviour (a message on
 * stderr and abort(); the reference throws through the extern "C" frame, wrappers.cpp:425,
 * or exit(1)s, :170).  Part 2 functions return 0 on success and a negative code on error;
 * wr_last_error() returns the message.  There is NO CPU fallback: without a usable GPU
 * every compute entry point fails loudly.
 */
#ifndef WAVERANGE_AMD_H
#define WAVERANGE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* Everything declared here -- and nothing else -- is exported by libwaverange_amd.so (built with -fvisibility=hidden):
 * the 24 symbols of the reference's libwaverange.so plus the wr_* functions. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* ----------------------------------------------------------------------------------- */
/* Part 1: libwaverange drop-in symbols                                                 */
/* ----------------------------------------------------------------------------------- */

/* replaces setup_wr, reference src/core/wrappers.cpp:531-541 (wrappers.h:75) */
void setup_wr(int nx, int ny, int nz, unsigned char *nlaymax, unsigned long *ntot_enc_max);

/* replaces encoding_wrap, reference src/core/wrappers.cpp:228-452 (wrappers.h:53).
 * fld_1d: host double[nx*ny*nz], x fastest.  data_enc: host buffer of ntot_enc_max bytes.
 * mx*my*mz > 1 selects the reference's non-uniform (local) cutoff branch, wrappers.cpp:343-379.
 * As in the reference (wrappers.cpp:397-398, README.md:197) fld_1d is overwritten with the residual
 * in wavelet space; WR_WRITEBACK_RESIDUAL=0 or wr_set_writeback_residual(0) skips that download.
 * Thread safety: like the reference, encoding_wrap / decoding_wrap / waveletcdf97_3d may be called
 * concurrently from several threads on distinct buffers (every call borrows its own context). */
void encoding_wrap(int nx, int ny, int nz, double *fld_1d, int wtflag, int mx, int my, int mz,
                   double *cutoffvec, double *tolabs, double *midval, double *halfspanval,
                   unsigned char *wlev, unsigned char *nlay, unsigned long *ntot_enc,
                   double *deps_vec, double *minval_vec, unsigned long *len_enc_vec,
                   unsigned char *data_enc);

/* replaces decoding_wrap, reference src/core/wrappers.cpp:456-527 (wrappers.h:70) */
void decoding_wrap(int nx, int ny, int nz, double *fld_1d, double *tolabs, double *midval,
                   double *halfspanval, unsigned char *wlev, unsigned char *nlay,
                   unsigned long *ntot_enc, double *deps_vec, double *minval_vec,
                   unsigned long *len_enc_vec, unsigned char *data_enc);

/* Fortran shims, reference src/core/wrappers.cpp:545-594 (wrappers.h:95,111,119) */
void setup_wr_f(int *nx, int *ny, int *nz, int *nlaymax, long *ntot_enc_max);
void encoding_wrap_f(int *nx, int *ny, int *nz, double *fld, int *wtflag, double *tolrel,
                     double *tolabs, double *midval, double *halfspanval, unsigned char *wlev,
                     unsigned char *nlay, long *ntot_enc, double *deps_vec, double *minval_vec,
                     long *len_enc_vec, unsigned char *data_enc);
void decoding_wrap_f(int *nx, int *ny, int *nz, double *fld, double *midval, double *halfspanval,
                     unsigned char *wlev, unsigned char *nlay, long *ntot_enc, double *deps_vec,
                     double *minval_vec, long *len_enc_vec, unsigned char *data_enc);

/* replaces waveletcdf97_3d, reference src/waveletcdf97_3d/waveletcdf97_3d.c:38 (exported by
 * the reference's .so; host buffer, in place; lvl>0 forward, lvl<0 inverse) */
void waveletcdf97_3d(int n1, int n2, int n3, int lvl, double *x);

/* Part 1b: the other symbols the reference's .so exports (host-only integer code) */
/* rangecoder state, layout of reference src/rangecod/rangecod.h:110-131 */
typedef struct {
    unsigned int low, range, help;
    unsigned char buffer;
    unsigned int bytecount;
    unsigned char *databuf;
    unsigned long datalen, datapos;
} rangecoder;
extern char coderversion[];
/* replace the rngcod13 primitives of reference src/rangecod/rangecod.c:170-404 */
void start_encoding(rangecoder *rc, char c, unsigned long initlength);
void encode_freq(rangecoder *rc, unsigned int sy_f, unsigned int lt_f, unsigned int tot_f);
void encode_shift(rangecoder *rc, unsigned int sy_f, unsigned int lt_f, unsigned int shift);
unsigned int done_encoding(rangecoder *rc);
int start_decoding(rangecoder *rc);
unsigned int decode_culfreq(rangecoder *rc, unsigned int tot_f);
unsigned int decode_culshift(rangecoder *rc, unsigned int shift);
void decode_update(rangecoder *rc, unsigned int sy_f, unsigned int lt_f, unsigned int tot_f);
unsigned char decode_byte(rangecoder *rc);
unsigned short decode_short(rangecoder *rc);
void done_decoding(rangecoder *rc);
void init_databuf(rangecoder *rc, unsigned long maxlen);
void free_databuf(rangecoder *rc);
void countblock(int *buffer, unsigned int length, unsigned int *counters);
void readcounts(rangecoder *rc, unsigned int *counters);
/* replaces ind_p2w_3d, reference src/waveletcdf97_3d/waveletcdf97_3d.c:473-553 */
void ind_p2w_3d(int lvlin, int n1, int n2, int n3, int i1in, int i2in, int i3in, int *lvl, int *i1,
                int *i2, int *i3);

/* ----------------------------------------------------------------------------------- */
/* Part 2: device-resident API                                                          */
/* ----------------------------------------------------------------------------------- */

#define WR_NLAYMAX 8 /* reference src/core/defs.h:38 */
#define WR_OK 0
#define WR_ERR_ARG (-1)
#define WR_ERR_HIP (-2)
#define WR_ERR_UNSUPPORTED (-3)
#define WR_ERR_STREAM (-4)
#define WR_ERR_OVERFLOW (-5)

typedef struct wr_ctx wr_ctx;

/* outputs of one encode = the header record of a field (reference .wrh fields) */
typedef struct wr_enc_info {
    double tolabs, midval, halfspanval;
    unsigned char wlev, nlay;
    unsigned long ntot_enc;
    double deps_vec[WR_NLAYMAX];
    double minval_vec[WR_NLAYMAX];
    unsigned long len_enc_vec[WR_NLAYMAX];
} wr_enc_info;

/* per-call stage timings in seconds (host wall clock around the stages) */
typedef struct wr_timings {
    double total;      /* whole call */
    double gpu;        /* the call's device phase: from getting a work-space slot to its last copy (upload,
                          min/max + transform + quantizer or dequant + inverse, downloads) */
    double transfer;   /* host time outside the range coder and the device phase (stream concatenation etc.) */
    double rangecoder; /* host range coder, wall time of the slowest plane thread */
    /* HIP-event durations on the context's stream, milliseconds */
    float transform_ms; /* all launches of the forward or inverse transform */
    float quant_ms;     /* all quantizer-plane (or the dequantise-accumulate) launches */
    float minmax_ms;    /* stand-alone min/max reductions */
    double wait;        /* waiting for a free work-space slot of the device */
    float h2d_ms;       /* host entry points: upload of the field (encode) / the planes (decode) */
    float d2h_ms;       /* host entry points: sum of the plane downloads (encode) / download of the field
                           (decode); engine timestamps for DMA copies */
    double plane_coder_s[WR_NLAYMAX]; /* host range coder, per plane: from the moment a coder took the plane's stream to its end (a
                           call is as long as its SLOWEST plane stream: `rangecoder` is the maximum of these) */
} wr_timings;

const char *wr_last_error(void);
int wr_device_count(void);
/* 0 = silent, 1 = the reference's progress lines on stdout (default; WR_QUIET=1 silences) */
void wr_set_verbosity(int level);
/* host range-coder threads per encode / decode call (default: one per plane; WR_THREADS=k in the
 * environment sets the default).  With fewer threads than planes a thread codes several planes
 * with interleaved symbol loops: less CPU time per field, more wall time for a single field. */
void wr_set_threads(int nthreads);
/* a different count for the encoder alone (0 = follow wr_set_threads, which also resets this): the
 * encoder interleaves 2 planes as efficiently as 3-4, the decoder is at its best with 4 per thread */
void wr_set_encoder_threads(int nthreads);
/* Process-wide coder pool for callers that keep several fields in flight: nthreads > 0 starts that many worker
 * threads which code the plane streams of ALL concurrent encode / decode calls (wr_set_threads is then
 * ignored); 0 stops it (default: every call runs its own coder threads).  A worker interleaves up to 3 encoder
 * or decoder_streams (1..4, default 4; < 1 keeps the setting) decoder streams in one symbol loop, whichever
 * fields they belong to, so a decoder loop is not limited to the 3-4 planes of one field, and the number of
 * running coder threads never exceeds nthreads.  Same bytes either way. */
void wr_set_coder_pool(int nthreads, int decoder_streams);
/* whether the drop-in encoding_wrap leaves the residual in fld_1d as the reference does (default 1,
 * WR_WRITEBACK_RESIDUAL in the environment): callers that discard the array save a field download */
void wr_set_writeback_residual(int on);
/* Work-space slots of a device (1..4, default 3 or WR_SLOTS): how many device phases may be in
 * flight at once -- one uploading, one in its kernels, one downloading.  A slot holds a staging
 * field, the coefficient array and the low-pass boxes of the transform (2.2 x the field size) and is
 * only populated when concurrent callers need it; if the device runs out of memory the library keeps
 * to the slots it has.  The quantized planes are not in the slot: see wr_ctx_create. */
int wr_set_device_slots(int device, int nslots);
/* process-wide event counters (diagnostics and tests) */
#define WR_STAT_EARLY_DECODES 0   /* decode calls whose planes went to the device window by window under the decoder */
#define WR_STAT_SLOTS_POPULATED 1 /* work-space slots that received device buffers */
#define WR_STAT_DEVICE_PLANE_BYTES 2 /* device memory of quantized planes allocated right now (in use + idle), all devices */
#define WR_STAT_POOL_IDLE_MS 3  /* milliseconds the coder pool's workers have waited for a job, summed over the workers */
#define WR_STAT_POOL_STREAMS_MOVED 4  /* plane streams that changed pool workers between two blocks: an idle worker takes over half
                                        of the streams of the fullest running session */
#define WR_STAT_POOL_QUEUE_MS 5  /* milliseconds plane jobs have waited in the coder pool's queues before a worker took them, summed over jobs */
#define WR_STAT_PLANE_WAIT_MS 6  /* milliseconds calls have waited for device memory for their quantized planes (a decoder also: for its turn
                                    to gather them), summed over calls */
#define WR_STAT_HANDOVER_ERRORS 7  /* window requests of a host coder that were refused: the plane stream they named had moved on to another
                                      call, they came out of order, or two coders were inside one stream (always 0 in a correct run; the
                                      call concerned fails) */
#define WR_STAT_CLOCK_WARMUP_MS 8  /* milliseconds of clock warm-up load put in front of kernel stages (WR_CLOCK_WARMUP_MS, a measurement
                                      hook that is off by default: always 0 then) */
#define WR_STAT_WINDOW_WAIT_MS 10  /* milliseconds host coders have waited inside their window requests for a window's DMA copy (a worker of the
                                    * coder pool is blocked then, not idle), summed over coders */
#define WR_STAT_DECODE_GATE_MS 9   /* milliseconds decode calls have waited for admission to the coder pool, holding no device memory yet
                                      (before: the same time in the pool's queues with their planes allocated), summed over calls */
unsigned long wr_stat(int what);
/* Hands the idle buffers of the device's plane pool back to the device (the pool keeps the plane memory of finished calls for
 * the next ones: after a burst of concurrent calls that can be most of the HBM).  Buffers in use are not touched. */
int wr_ctx_trim(wr_ctx *c);
/* coder pool, per loop kind {scalar encoder, scalar decoder, 16-lane decoder for dominant-symbol planes, 16-lane encoder}
 * -- WR_POOL_LOOP_KINDS entries each: seconds the workers have spent in block steps of that loop and stream-blocks (60000
 * symbols) advanced: symbols per worker-second in the pipeline */
#define WR_POOL_LOOP_KINDS 4
void wr_pool_loop_stats(double *seconds, double *blocks);

/* One context per concurrent caller: (device, kernel stream, coded-stream buffers, and per plane a ring
 * of two 15 MB pinned chunks), grown on demand and kept.  The device work space (wr_set_device_slots)
 * and the device buffers of the quantized planes are shared between the contexts of a GPU: the planes
 * stay in device memory -- a call borrows one buffer per plane while the plane exists and returns it
 * -- and the host coder reads or writes them through the ring, chunk by chunk, so no whole plane is
 * ever staged in host memory.  stream == NULL makes the context create its own. */
int wr_ctx_create(wr_ctx **ctx, int device, void *hip_stream);
void wr_ctx_destroy(wr_ctx *ctx);
int wr_ctx_sync(wr_ctx *ctx);
/* keep_residual != 0: also apply the residual update on the last plane, so that the device
 * field ends up bit-identical to what the reference leaves in fld_1d (wrappers.cpp:397-398) */
void wr_ctx_set_keep_residual(wr_ctx *ctx, int keep_residual);

/* device memory helpers (thin hipMalloc/hipMemcpy wrappers so callers need no HIP headers) */
int wr_dev_alloc(wr_ctx *ctx, void **ptr, size_t bytes);
int wr_dev_free(wr_ctx *ctx, void *ptr);
int wr_dev_upload(wr_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int wr_dev_download(wr_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* pinned host memory: field and coded-stream buffers allocated here move over PCIe by DMA without
 * a staging copy (pageable buffers work everywhere too, at roughly half the rate) */
int wr_host_alloc(void **ptr, size_t bytes);
int wr_host_free(void *ptr);
/* Pin a buffer the caller already owns (hipHostRegister): its copies then move by DMA without the runtime's staging
 * copy.  They go through hipMemcpyAsync, not through the library's own SDMA path (the GPU sees registered memory at
 * another address than the host does, which only HIP's copy translates): under many concurrent calls wr_host_alloc'd
 * buffers are the faster choice.  Unregister before freeing the buffer. */
int wr_host_register(void *ptr, size_t bytes);
int wr_host_unregister(void *ptr);

int wr_dev_copy(wr_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes); /* ctx stream, waits */
/* measurement hook: copies through the compute units with `workgroups` workgroups on the context's stream, without
 * waiting (either side may be pinned host memory; 16-byte granularity) */
int wr_dev_copy_kernel(wr_ctx *ctx, void *dst, const void *src, size_t bytes, int workgroups);
/* measurement hook: `workgroups` workgroups stay on the device for `ms` milliseconds on the context's stream, without waiting
 * (mode 0: a chain of fp64 arithmetic, 1: asleep) -- what it takes to bring the shader clock up before a kernel stage */
int wr_dev_burn(wr_ctx *ctx, double ms, int mode, int workgroups);
/* max|a-b| and max|a| over n doubles (accuracy check of a reconstruction, "L-inf vs tol") */
int wr_dev_linf(wr_ctx *ctx, const double *d_a, const double *d_b, size_t n, double *max_abs_diff,
                double *max_abs_a);

/* --- stage-level entry points on device pointers (d_ prefix = device memory, 16-B aligned) */
/* transform in place: a1/a2 of SURVEY.md 8a */
int wr_dev_transform(wr_ctx *ctx, double *d_fld, int nx, int ny, int nz, int lvl);
/* min/max with the reference's scan semantics incl. the sign of a zero minimum: a4/a5 */
int wr_dev_minmax(wr_ctx *ctx, const double *d_x, size_t n, double *mn, double *mx);
/* one quantizer plane + residual update; next_min/next_max = extrema of the new residual */
int wr_dev_quantize_plane(wr_ctx *ctx, double *d_x, size_t n, double deps, double minval,
                          unsigned char *d_q, double *next_min, double *next_max);
/* acc = sum over planes of (q*deps + minval), planes are nlay device arrays of n bytes */
int wr_dev_dequant_accum(wr_ctx *ctx, double *d_acc, size_t n, int nlay,
                         const unsigned char *const *d_planes, const double *deps,
                         const double *minval);
/* synthetic field generator (waverange_amd/synth.py) straight into device memory */
int wr_dev_synth_field(wr_ctx *ctx, double *d_out, int nx, int ny, int nz,
                       unsigned long long seed);

/* --- device-only part of the codec (no range coder): transform + all quantizer planes.
 * d_fld is CONSUMED: with wr_ctx_set_keep_residual(ctx, 1) it holds the residual in wavelet space afterwards (what the
 * reference leaves in fld_1d, wrappers.cpp:397-398); without, its contents are unspecified (since round 4 the planes are cut
 * from residuals recomputed from the coefficient array, which nobody writes back: the array then holds the coefficients).
 * The same goes for d_fld of wr_encode_device.  d_planes receives nlay planes at a pitch of wr_plane_pitch(n) bytes.
 * Fills tolabs/midval/halfspanval/wlev/nlay/deps/minval. */
size_t wr_plane_pitch(size_t n);
int wr_dev_encode_planes(wr_ctx *ctx, double *d_fld, int nx, int ny, int nz, int wtflag,
                         double tolrel, unsigned char *d_planes, wr_enc_info *info);
int wr_dev_decode_planes(wr_ctx *ctx, double *d_fld, int nx, int ny, int nz,
                         const unsigned char *d_planes, const wr_enc_info *info);

/* --- whole hot path with the field resident in HBM: encode -> host byte stream, and back.
 * data_enc: host buffer of at least setup_wr()'s ntot_enc_max bytes (cap is checked). */
int wr_encode_device(wr_ctx *ctx, double *d_fld, int nx, int ny, int nz, int wtflag,
                     double tolrel, wr_enc_info *info, unsigned char *data_enc, size_t cap,
                     wr_timings *tm);
/* same with the reference's local cutoff vector (mx*my*mz entries, host memory) */
int wr_encode_device_local(wr_ctx *ctx, double *d_fld, int nx, int ny, int nz, int wtflag, int mx,
                           int my, int mz, const double *cutoffvec, wr_enc_info *info,
                           unsigned char *data_enc, size_t cap, wr_timings *tm);
/* data_len: bytes readable at data_enc (0 = trust info->ntot_enc, as the reference does) */
int wr_decode_device(wr_ctx *ctx, double *d_fld, int nx, int ny, int nz,
                     const wr_enc_info *info, const unsigned char *data_enc, size_t data_len,
                     wr_timings *tm);

/* --- whole hot path host buffer to host buffer: what encoding_wrap / decoding_wrap run on, with an
 * explicit context (one per concurrent caller), error codes and timings.  The field is staged through
 * the device's work-space slot: upload on the device's upload stream, kernels, planes / field back on
 * the download stream, so that concurrent calls overlap their copies with one another's kernels and
 * host range coding.  h_fld may be pinned (wr_host_alloc) or pageable.  Encode leaves h_fld untouched
 * unless wr_ctx_set_keep_residual(ctx, 1) asks for the reference's residual write-back. */
int wr_encode_host(wr_ctx *ctx, double *h_fld, int nx, int ny, int nz, int wtflag, int mx, int my,
                   int mz, const double *cutoffvec, wr_enc_info *info, unsigned char *data_enc,
                   size_t cap, wr_timings *tm);
int wr_decode_host(wr_ctx *ctx, double *h_fld, int nx, int ny, int nz, const wr_enc_info *info,
                   const unsigned char *data_enc, size_t data_len, wr_timings *tm);
/* Decode in two calls, for callers that want to bound their output buffers: the host range decoding takes seconds
 * and needs no field buffer -- wr_decode_begin runs it, every decoded window going straight to the planes' device
 * buffers, which stay parked in the context -- the field buffer is only touched by the ~0.2 s of kernels and
 * download that wr_decode_finish_host / _device run.  (A streaming decoder with many fields in flight holds one
 * output field per finish in progress instead of one per field.)  One begin may be pending per context (another
 * begin, a whole decode or an encode on the context discards it); data_enc is not needed after begin has returned. */
int wr_decode_begin(wr_ctx *ctx, int nx, int ny, int nz, const wr_enc_info *info,
                    const unsigned char *data_enc, size_t data_len, wr_timings *tm);
int wr_decode_finish_host(wr_ctx *ctx, double *h_fld, wr_timings *tm);
int wr_decode_finish_device(wr_ctx *ctx, double *d_fld, wr_timings *tm);
/* waveletcdf97_3d on a host array, in place */
int wr_transform_host(wr_ctx *ctx, double *h_fld, int nx, int ny, int nz, int lvl);

/* --- host range coder alone (one plane stream), rows a6/a7/a10 of SURVEY.md 8a */
size_t wr_range_encode_bound(size_t n);
/* the same for a plane whose byte histograms per 60000-symbol block are known (unsigned short[256] per block, n/60000+1
 * blocks): block entropies + headers + the coder's worst-case rounding loss (0.0104 bit per symbol), a rigorous bound
 * within ~0.2 % of the stream's length.  wr_encode_* use it to code the planes of a field side by side straight into
 * data_enc (the reference codes each plane into a buffer of its own and copies, wrappers.cpp:412-427). */
size_t wr_range_encode_bound_hist(const unsigned short *hists, size_t n);
size_t wr_range_encode(const unsigned char *sym, size_t n, unsigned char *out);
size_t wr_range_decode(const unsigned char *in, size_t len, unsigned char *sym, size_t n);
/* `count` planes of n symbols each coded on the calling thread, their symbol loops interleaved
 * (up to 4 at a time): the same bytes as `count` calls of the functions above, at a fraction of
 * the CPU time, because one plane's coder is a serial dependency chain that leaves most of a
 * core idle.  out[k] holds wr_range_encode_bound(n) bytes; produced[k] as wr_range_decode. */
void wr_range_encode_multi(int count, const unsigned char *const *sym, size_t n,
                           unsigned char *const *out, size_t *lens);
void wr_range_decode_multi(int count, const unsigned char *const *in, const size_t *len,
                           unsigned char *const *sym, size_t n, size_t *produced);

/* the same through the coder pool (wr_set_coder_pool must have started it): `count` planes of their own lengths
 * n[k], coded by the pool's workers next to whatever else is queued; returns when all of them are done */
int wr_range_encode_pool(int count, const unsigned char *const *sym, const size_t *n,
                         unsigned char *const *out, size_t *lens);
int wr_range_decode_pool(int count, const unsigned char *const *in, const size_t *len,
                         unsigned char *const *sym, const size_t *n, size_t *produced);

/* `count` planes on the calling thread through the 16-lane AVX-512 loops: the encoder's takes planes of any kind,
 * the decoder's gains on dominant-symbol planes (any plane decodes correctly, the others just gain nothing);
 * WR_ERR_UNSUPPORTED on a CPU without AVX-512.  The coder pool routes all encoder planes and the decoder planes
 * below 2 bits per symbol there by itself. */
int wr_range_encode_vec(int count, const unsigned char *const *sym, const size_t *n,
                        unsigned char *const *out, size_t *lens);
int wr_range_decode_vec(int count, const unsigned char *const *in, const size_t *len,
                        unsigned char *const *sym, const size_t *n, size_t *produced);

/* Test hooks for the windowed symbol path: planes that live in device memory reach the host coder through a small
 * pinned ring, window by window (wr_encode_host / wr_decode_*); here the windows are `chunk` symbols (a multiple of
 * 60000) of plain host buffers.  mode 0: interleaved loops on the calling thread, 1: the coder pool, 2: the 16-lane
 * loops.  Same bytes / symbols as the whole-plane functions above. */
int wr_range_encode_windowed(int mode, int count, const unsigned char *const *sym, size_t n, size_t chunk,
                             unsigned char *const *out, size_t *lens);
int wr_range_decode_windowed(int mode, int count, const unsigned char *const *in, const size_t *len,
                             unsigned char *const *sym, size_t n, size_t chunk, size_t *produced);

/* Test hook for the plane hand-over (wr_handover.h): replays the window handle of a finished call against the plane of the
 * next call on the same context; 0 if the request was refused and left that plane untouched. */
int wr_test_stale_window(wr_ctx *ctx, size_t n);

/* --- for callers with a batch of independent fields (the wrenc / wrdec / FluSI tools): starts the coder pool with one
 * worker per CPU this process may use (affinity mask, cgroup quota) when nfields > 1, and returns how many
 * encoding_wrap / decoding_wrap (or wr_*_host) calls on fields of field_elems elements to keep in flight at once:
 * 1.5 per CPU, fewer if host memory or device memory are short, never more than nfields.  The calls themselves are
 * unchanged (same bytes); this only sizes the concurrency around them. */
int wr_autotune_batch(size_t field_elems, int nfields);

/* --- measurement hook for bench.py: runs `reps` forward (lvl>0) or inverse transforms of an
 * nx*ny*nz field back to back on the context's stream and returns the average duration of
 * one transform in milliseconds measured with HIP events on that stream. */
int wr_bench_transform(wr_ctx *ctx, double *d_fld, int nx, int ny, int nz, int lvl, int reps,
                       double *ms_per_transform);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
