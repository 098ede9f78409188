// wr_codec.cpp -- the codec drivers of libwaverange_amd: a field through the stages of wr_pipeline.cpp.
//
// Mirrors the reference's codec layer (src/core/wrappers.cpp) for the path
//   encoding_wrap : min/max -> forward transform -> bit-plane quantizer loop -> range coder     (wrappers.cpp:228-452)
//   decoding_wrap : range decoder -> dequantise-accumulate -> inverse transform                  (wrappers.cpp:456-527)
// Compiled with hipcc, strict IEEE (-ffp-contract=off): the scalar arithmetic on deps/aopt/bopt/tolabs below must round
// exactly as wrappers.cpp:292-340 does.
#include "wr_internal.h"

using namespace wri;

namespace {

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

}  // namespace

namespace {

// local cutoff description (mx*my*mz == 1: uniform cutoff, the benchmark path)
struct Cutoff {
    int mx = 1, my = 1, mz = 1;
    const double* vec = nullptr;  // host, mx*my*mz entries
    int count() const { return mx * my * mz; }
};

// Kernel stage of the encoder (call with the slot leased and DevPool::cu_mu held).  Hooks for the full
// pipeline: hist_buf(l) says where plane l's block histograms go on the device (nullptr: nobody wants them);
// after_quant(l, hist_done) is called right after plane l's quantizer kernel and the read-back of the next
// plane's min/max have been enqueued (hist_done: the quantizer has written the histograms on its way; otherwise they
// are enqueued there, behind the read-back);
// plane_ready(l, last) is called from the host once everything enqueued for plane l has completed on the
// device (the download starts there).  *resid = where the coefficient array / residual lives afterwards.
template <class PlaneBuf, class HistBuf, class AfterQuant, class PlaneReady>
int encode_planes_core(wr_ctx* c, Slot* slot, double* d_fld, int nx, int ny, int nz, int wtflag, const Cutoff& cut,
                       PlaneBuf plane_buf, HistBuf hist_buf, wr_enc_info* info, wr_timings* tm, AfterQuant after_quant, PlaneReady plane_ready,
                       double** resid)
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
    Prologue p;
    *resid = d_fld;
    double lo, hi;
    float ms = 0;
    // When all four levels run on the fused kernels, those reduce min/max of the field and of the coefficient
    // array on the way (no stand-alone passes, one host round trip instead of two).  The transform then runs
    // before it is known whether the field is trivial; it is out of place, so nothing is lost if it is.
    const size_t mm_records = (wtflag && use_fused(nx, ny, nz, kWavLvl)) ? wrk::fused_minmax_records(nx, ny, nz) : 0;
    if (mm_records) {
        if (const char* why = wrk::fused_prepare()) return fail(WR_ERR_HIP, why);
        if (c->mm_records < mm_records) {
            if (c->d_mm) HIPCHK(hipFree(c->d_mm));
            c->d_mm = nullptr; c->mm_records = 0;
            HIPCHK(hipMalloc(&c->d_mm, mm_records * 4 * sizeof(double)));
            c->mm_records = mm_records;
        }
        memset(info, 0, sizeof(*info));
        info->wlev = kWavLvl;
        double* const d_in = d_fld;
        HIPCHK(hipEventRecord(c->ev_b, c->stream));
        wrk::transform_fwd_fused(d_in, slot->scratch, slot->lowbuf, nx, ny, nz, c->stream, c->d_mm, c->h_result_dev + 4);
        HIPCHK(hipEventRecord(c->ev_c, c->stream));
        HIPCHK(hipEventRecord(c->ev_mm, c->stream));
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventSynchronize(c->ev_mm));
        p.lo = c->h_result[4]; p.hi = c->h_result[5];
        lo = c->h_result[6]; hi = c->h_result[7];
        if (p.lo == 0.0)  // sign of a zero minimum: the reference's scan semantics, rare path
            if (int rc = read_minmax(c, d_in, n, false, &p.lo, &p.hi)) return rc;
        if (p.lo != p.lo || p.hi != p.hi) return fail(WR_ERR_ARG, "field is all NaN");
        info->halfspanval = (p.hi - p.lo) / 2;
        info->midval = p.lo + info->halfspanval;
        p.trivial = info->halfspanval <= 2 * DBL_MIN;
        if (verbose()) printf("Wavelet decomposition...\n");
        if (p.trivial) {  // wrappers.cpp:256-266
            info->ntot_enc = 0; info->nlay = 0; info->tolabs = 0;
            return WR_OK;
        }
        d_fld = slot->scratch;  // d_fld := coefficients
        *resid = d_fld;
        if (verbose()) printf("Range encoding...\n");
        info->tolabs = abs_tolerance(tolrel, p);
        if (lo == 0.0)
            if (int rc = read_minmax(c, d_fld, n, false, &lo, &hi)) return rc;
        if (tm) { HIPCHK(hipEventElapsedTime(&ms, c->ev_b, c->ev_c)); tm->transform_ms = ms; tm->minmax_ms = 0; }
    } else {
    HIPCHK(hipEventRecord(c->ev_a, c->stream));
    if (int rc = prologue(c, d_fld, n, wtflag, info, &p)) return rc;
    if (verbose()) printf("Wavelet decomposition...\n");
    if (p.trivial) {  // wrappers.cpp:256-266
        info->ntot_enc = 0; info->nlay = 0; info->tolabs = 0;
        return WR_OK;
    }
    HIPCHK(hipEventRecord(c->ev_b, c->stream));
    if (int rc = run_transform(c, slot, d_fld, nx, ny, nz, (int)info->wlev, &d_fld)) return rc;  // d_fld := coefficients
    *resid = d_fld;
    HIPCHK(hipEventRecord(c->ev_c, c->stream));
    if (verbose()) printf("Range encoding...\n");
    info->tolabs = abs_tolerance(tolrel, p);

    if (int rc = read_minmax(c, d_fld, n, false, &lo, &hi)) return rc;
    HIPCHK(hipEventRecord(c->ev_d, c->stream));
    HIPCHK(hipEventSynchronize(c->ev_d));
    if (tm) {
        HIPCHK(hipEventElapsedTime(&ms, c->ev_b, c->ev_c)); tm->transform_ms = ms;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_a, c->ev_b)); tm->minmax_ms = ms;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_c, c->ev_d)); tm->minmax_ms += ms;
    }
    }
    unsigned ilay = 0;
    float quant_ms = 0;
    // The residual is not kept in memory between planes (wr_kernels.hip: k_quant_blk): d_fld stays the coefficient array, every
    // plane's kernel redoes the subtractions of the planes in `prev`.  The local-cutoff quantizer and unaligned caller
    // pointers keep the reference's form (the array is updated in place after every plane).
    wrk::QuantPrev prev;
    for (;;) {
        PlaneStep s = plane_step(lo, hi, info->tolabs, ilay);
        info->minval_vec[ilay] = s.minval;
        info->deps_vec[ilay] = s.deps;
        if (verbose()) { printf("min=%g max=%g\n", lo, hi); printf("ilay=%u deps=%g\n", ilay, s.deps); }
        const wrk::PlaneRef* const d_plane = plane_buf(ilay);  // device memory of this plane (the error is set if there is none)
        if (!d_plane) return WR_ERR_HIP;
        if (!wrk::plane_ref_covers(*d_plane, n)) return fail(WR_ERR_HIP, "internal: the device buffer of plane " + std::to_string(ilay) + " has a hole");
        // (a chunked plane is only ever indexed through its table: the direct-form kernels for unaligned pointers take one array)
        if (d_plane->shift < 63 && (((uintptr_t)d_fld | (uintptr_t)d_plane->chunk[0]) & 15)) return fail(WR_ERR_ARG, "internal: chunked plane with an unaligned pointer");
        const bool blk = !local && (prev.n > 0 || ilay == 0) && wrk::quantize_plane_blk_ok(d_fld, *d_plane);
        const bool resid_upd = blk ? (s.last && c->keep_residual) : (!s.last || c->keep_residual);
        // (the histograms by a kernel of their own behind the quantizer: 10.7 + 0.9 ms per field against 6.7,
        // profiles/r04/x_quantizer_with_and_without_fused_histograms.txt)
        uint16_t* const d_hist = blk ? hist_buf(ilay) : nullptr;
        launch_note(c, local ? "quant_local" : blk ? (resid_upd ? "quant_blk<1>" : "quant_blk<0>") : resid_upd ? "quant<1>" : "quant<0>", (int)ilay, d_fld, n,
                    c->d_partial, *d_plane);
        HIPCHK(hipEventRecord(c->ev_a, c->stream));
        if (local) {
            wrk::LocalCutoff lc;
            lc.nx = nx; lc.ny = ny; lc.nz = nz; lc.wlev = info->wlev;
            lc.mx = cut.mx; lc.my = cut.my; lc.mz = cut.mz;
            lc.cutoff = c->d_cutoff;
            lc.tol_scale = info->tolabs / tolrel;
            lc.tolabs = info->tolabs;
            lc.span = hi - lo;
            wrk::quantize_plane_local(d_fld, n, s.aopt, s.bopt, s.deps, s.minval, d_plane->chunk[0], lc,  // (one array: plane_prepare(contiguous))
                                      c->d_partial, c->h_result_dev, c->stream);
        } else if (blk)
            wrk::quantize_plane_blk(d_fld, n, prev, s.aopt, s.bopt, s.deps, s.minval, *d_plane, resid_upd, !s.last, d_hist, c->d_partial,
                                    c->h_result_dev, c->stream);
        else
            wrk::quantize_plane(d_fld, n, s.aopt, s.bopt, s.deps, s.minval, *d_plane, resid_upd, c->d_partial, c->h_result_dev, c->stream);
        HIPCHK(hipEventRecord(c->ev_b, c->stream));
        if (hipGetLastError() != hipSuccess) return fail(WR_ERR_HIP, "quantizer launch failed" + launch_describe(c));
        // the next plane's min/max is in host memory when this event fires (the reduction stores it there); what
        // after_quant enqueues runs behind it
        HIPCHK(hipEventRecord(c->ev_mm, c->stream));
        if (int rc = after_quant(ilay, d_hist != nullptr)) return rc;
        HIPCHK(hipEventRecord(c->ev_plane[ilay], c->stream));
        HIPCHK(hipEventSynchronize(c->ev_mm));
        HIPCHK(hipEventElapsedTime(&ms, c->ev_a, c->ev_b)); quant_ms += ms;
        // plane ilay-1 is complete: what after_quant enqueued for it ran before this plane's quantizer
        if (ilay > 0) if (int rc = plane_ready(ilay - 1, false)) return rc;
        if (s.last) {
            HIPCHK(hipEventSynchronize(c->ev_plane[ilay]));
            if (int rc = plane_ready(ilay, true)) return rc;
            ilay++;
            break;
        }
        if (blk) prev.push(s.aopt, s.bopt, s.deps, s.minval);
        ilay++;
        lo = c->h_result[0]; hi = c->h_result[1];
        // (WR_TEST_ZERO_MIN_PATH=1: tests take the rare path below after every plane)
        static const bool force_rare = getenv("WR_TEST_ZERO_MIN_PATH") && atoi(getenv("WR_TEST_ZERO_MIN_PATH"));
        if (lo == 0.0 || force_rare) {  // sign of a zero minimum: rare path, goes through the full read-back -- of the residual, in memory
            if (prev.n) { wrk::residual_apply(d_fld, n, prev, c->stream); prev.n = 0; }  // (the planes from here on update it in place)
            if (int rc = read_minmax(c, d_fld, n, true, &lo, &hi)) return rc;
        }
    }
    info->nlay = (unsigned char)ilay;
    if (tm) tm->quant_ms = quant_ms;
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
    SlotNeed need;
    transform_need(nx, ny, nz, wtflag ? kWavLvl : 0, &need);
    SlotLease slot;
    if (int rc = slot.acquire(c, need)) return rc;
    StageLock cu(c->pool->cu_mu);
    Cutoff cut; cut.vec = &tolrel;
    double* resid = nullptr;
    const size_t pitch = wr_plane_pitch((size_t)nx * ny * nz);
    wrk::PlaneRef refs[WR_NLAYMAX];
    for (int l = 0; l < WR_NLAYMAX; l++) refs[l] = wrk::plane_ref(d_planes + l * pitch);
    int rc = encode_planes_core(c, slot.get(), d_fld, nx, ny, nz, wtflag, cut, [&](unsigned l) { return &refs[l]; },
                                [](unsigned) { return (uint16_t*)nullptr; }, info, nullptr, [](unsigned, bool) { return WR_OK; },
                                [](unsigned, bool) { return WR_OK; }, &resid);
    if (rc == WR_OK && resid != d_fld && info->nlay)  // d_fld holds the residual afterwards (header contract)
        if (hipMemcpyAsync(d_fld, resid, (size_t)nx * ny * nz * sizeof(double), hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
            rc = fail(WR_ERR_HIP, "residual copy failed");
    (void)hipStreamSynchronize(c->stream);
    return rc;
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
    SlotNeed need;
    transform_need(nx, ny, nz, info->wlev ? -kWavLvl : 0, &need);
    SlotLease slot;
    if (int rc = slot.acquire(c, need)) return rc;
    StageLock cu(c->pool->cu_mu);
    wrk::DequantParams p;
    memset(&p, 0, sizeof p);
    p.nlay = info->nlay;
    for (int l = 0; l < p.nlay; l++) {
        p.q[l] = wrk::plane_ref(d_planes + l * wr_plane_pitch(n));
        p.deps[l] = info->deps_vec[l];
        p.minval[l] = info->minval_vec[l];
    }
    int rc = inverse_from_planes(c, slot.get(), d_fld, nx, ny, nz, (int)info->wlev, p);
    if (rc == WR_OK && hipGetLastError() != hipSuccess) rc = fail(WR_ERR_HIP, "kernel launch failed");
    (void)hipStreamSynchronize(c->stream);
    return rc;
}

}  // extern "C"

namespace {

// where the field of an encode call comes from / the field of a decode call goes to
struct FieldRef {
    double* dev = nullptr;   // device-resident (caller's buffer), or
    double* host = nullptr;  // host buffer (pinned or pageable): staged through the slot
};

int encode_impl(wr_ctx* c, FieldRef fld, int nx, int ny, int nz, int wtflag, const Cutoff& cut, wr_enc_info* info,
                unsigned char* data_enc, size_t cap, wr_timings* tm)
{
    if (int rc = ctx_bind(c)) return rc;
    if (int rc = check_dims(nx, ny, nz, fld.dev)) return rc;
    if (!fld.dev && !fld.host) return fail(WR_ERR_ARG, "null field pointer");
    if (cut.mx < 1 || cut.my < 1 || cut.mz < 1 || !cut.vec) return fail(WR_ERR_ARG, "bad local cutoff description");
    std::lock_guard<std::mutex> lk(c->mu);
    ActiveCall active(c->pool);
    if (tm) wrdma::enable_timing();
    const double t0 = now();
    const size_t n = (size_t)nx * ny * nz;
    wr_timings local; memset(&local, 0, sizeof local);
    // per-60000-symbol-block byte histograms, counted on the GPU next to the quantizer and shipped
    // with the plane, so that the host coder starts every block with its model ready
    const size_t hist_per_plane = (n / wrrc::kBlock + 1) * 256;
    DevPool* const pool = c->pool;

    size_t lens[WR_NLAYMAX] = {0};
    double coder_s[WR_NLAYMAX] = {0};
    int copy_failed[WR_NLAYMAX] = {0};
    std::string logs[WR_NLAYMAX];
    Sem sem(encoder_threads());
    const int dev = c->device;
    c->pend_valid = false;  // planes a wr_decode_begin parked in this context do not survive an encode on it
    PlaneHold planes(c);  // before the workers: they are joined first when the call unwinds
    Workers workers;

    // The planes stay in device memory; a coder reads its plane through the plane's ring of pinned chunks
    // (PlaneStream).  With a coder thread for every possible plane, plane l's thread starts as soon as the plane is
    // complete on the device and its histograms are on the host.  With fewer (wr_set_threads), the planes are split into that many groups once
    // their number is known and each thread codes its group with the symbol loops interleaved.
    const bool pooled = wrrc::pool_threads() > 0;  // the process-wide coder pool codes the planes (wr_set_coder_pool)
    const bool per_plane = !pooled && encoder_threads() >= WR_NLAYMAX;
    wrrc::PlaneJob jobs[WR_NLAYMAX];
    wrrc::JobBatch batch;
    unsigned pool_mask = 0;  // planes the pool took
    // Where a plane's coder writes.  The reference codes every plane into a buffer of its own and copies the streams
    // together (wrappers.cpp:412-427); here the planes are coded at the same time, so a plane's place in data_enc is not
    // known when its coder starts -- but an upper bound on the length of every plane before it is, from their block
    // histograms (wrrc::encode_bound_hist: rigorous, ~0.2 % above the stream).  Plane l is coded straight into data_enc at
    // the sum of the bounds of planes 0 .. l-1, and the gaps are closed afterwards by moving the planes down in order (a few
    // megabytes each).  The coded bytes of a field then exist once in host memory, not twice (a 1e-7 field at 1024^3: 2 GB),
    // and the context keeps no per-plane output buffers.  A plane whose bound does not fit under `cap` takes the old way
    // through c->enc_buf (only then can total <= cap < sum of bounds happen).  The coder is told the bound (dst_limit): a
    // stream that outgrows it was coded against histograms that are not its plane's and is given up, at most one block
    // (wrrc::kFailedBlockSlack) late -- which is why that much room lies behind every plane's place.
    std::mutex place_mu;
    unsigned placed = 0;
    size_t est_off[WR_NLAYMAX + 1] = {0}, est_len[WR_NLAYMAX] = {0};
    uint8_t* plane_out[WR_NLAYMAX] = {nullptr};
    bool direct[WR_NLAYMAX] = {false};
    // (callable from the coder threads: the histograms of plane k and of the planes before it have been sent off)
    auto place_plane = [&](unsigned k) -> uint8_t* {
        std::lock_guard<std::mutex> lk(place_mu);
        for (unsigned j = placed; j <= k; j++) {
            if (xfer_wait(&c->x_plane[j]) != WR_OK) copy_failed[j] = 1;
            est_len[j] = copy_failed[j] ? wrrc::encode_bound(n) : wrrc::encode_bound_hist(c->h_hist + j * hist_per_plane, n);
            est_off[j + 1] = est_off[j] + est_len[j];
            direct[j] = data_enc && est_off[j + 1] + wrrc::kFailedBlockSlack <= cap;
            if (direct[j]) plane_out[j] = data_enc + est_off[j];
            else plane_out[j] = ensure_enc_buf(c, (int)j, est_len[j] + wrrc::kFailedBlockSlack) == WR_OK ? c->enc_buf[j] : nullptr;
            placed = j + 1;
        }
        return plane_out[k];
    };
    auto code_group = [&](unsigned l0, unsigned l1) {
        (void)hipSetDevice(dev);
        for (unsigned l = l0; l < l1; l++)
            if (xfer_wait(&c->x_plane[l]) != WR_OK) copy_failed[l] = 1;
        for (unsigned l = l0; l < l1; l++)
            if (!place_plane(l)) { copy_failed[l] = 1; return; }
        sem.acquire();
        const double t = now();
        const uint8_t* syms[WR_NLAYMAX];
        uint8_t* outs[WR_NLAYMAX];
        const uint16_t* hs[WR_NLAYMAX];
        const wrrc::PlaneWindow* ios[WR_NLAYMAX];
        size_t limits[WR_NLAYMAX];
        for (unsigned l = l0; l < l1; l++) {
            syms[l - l0] = nullptr; outs[l - l0] = plane_out[l]; hs[l - l0] = c->h_hist + l * hist_per_plane; ios[l - l0] = &c->ps[l].io;
            limits[l - l0] = est_len[l];
        }
        wrrc::encode_planes((int)(l1 - l0), syms, n, outs, hs, lens + l0, ios, limits);
        for (unsigned l = l0; l < l1; l++) coder_s[l] = now() - t;
        sem.release();
    };

    SlotNeed need;
    transform_need(nx, ny, nz, wtflag ? kWavLvl : 0, &need);
    if (fld.host) need.field_elems = n;
    need.hist_elems = hist_per_plane * WR_NLAYMAX;
    if (int rc = ensure_host_hist(c, hist_per_plane * WR_NLAYMAX)) return rc;

    int rc = WR_OK;
    double t_phase = 0, t_gpu_done = 0;
    unsigned planes_started = 0;
    try {
        SlotLease slot;
        if ((rc = slot.acquire(c, need)) != WR_OK) return rc;
        t_phase = now();
        double* d_fld = fld.dev;
        if (fld.host) {
            // ---- stage "up": the field goes host -> device; the kernel stage is only claimed once it has
            // arrived, so other calls compute meanwhile
            d_fld = slot->field;
            if ((rc = xfer_field(c, &c->x_field, d_fld, fld.host, n * sizeof(double), kUp)) != WR_OK) return rc;
            local.h2d_ms = (float)c->x_field.ms;
        }
        double* resid = d_fld;
        auto hist_buf = [&](unsigned l) { return slot->hist + l * hist_per_plane; };
        auto after_quant = [&](unsigned l, bool hist_done) -> int {
            if (hist_done) return WR_OK;  // the quantizer wrote them on its way (k_quant_blk)
            // block histograms of plane l on the kernel stream, behind the read-back of the next plane's min/max
            launch_note(c, "hist", (int)l, slot->hist + l * hist_per_plane, n, nullptr, c->ps[l].ref);
            wrk::block_histograms(c->ps[l].ref, n, slot->hist + l * hist_per_plane, c->stream);
            return WR_OK;
        };
        // (a plane that has to wait for device memory gives the kernel stage up meanwhile: cu)
        StageLock cu(pool->cu_mu, std::defer_lock);
        // (the local-cutoff quantizer scatters into the plane by wavelet-space index: it wants one array)
        const bool one_array = cut.count() > 1;
        // pool: plane k goes to the workers once its histograms are on the host (they set off when it completed)
        unsigned handed = 0;  // planes that have a coder (a pool job or, if the pool refused, a thread of this call)
        unsigned ready = 0;   // planes whose histograms and first window are on their way to the host (plane_ready has run)
        auto submit_plane = [&](unsigned k) {
            if (handed >> k & 1) return;
            if (xfer_wait(&c->x_plane[k]) != WR_OK) { copy_failed[k] = 1; return; }
            uint8_t* const dst = place_plane(k);
            if (!dst || copy_failed[k]) { copy_failed[k] = 1; return; }
            handed |= 1u << k;
            wrrc::PlaneJob& j = jobs[k];
            j.kind = wrrc::PlaneJob::kEncode;
            j.src = nullptr; j.io = &c->ps[k].io; j.dst = dst; j.n = n; j.hist = c->h_hist + k * hist_per_plane; j.dst_limit = est_len[k];
            if (wrrc::pool_submit(&j, 1, &batch)) pool_mask |= 1u << k;
            else workers.v.emplace_back(code_group, k, k + 1);  // the pool was stopped meanwhile: a thread of this call codes the plane
        };
        auto plane_ready = [&](unsigned l, bool) -> int {
            // plane l and its histograms are complete on the device: the histograms go to pinned host memory, the
            // plane's first chunk sets off into its ring, and a coder thread waits for them
            if (ready >> l & 1) return WR_OK;  // (done early, by a later plane that had to wait for device memory: before_wait)
            ready |= 1u << l;
            const Piece pc = {c->h_hist + l * hist_per_plane, slot->hist + l * hist_per_plane, hist_per_plane * sizeof(uint16_t)};
            if (int r = xfer_start(c, &c->x_plane[l], &pc, 1, kDown)) return r;
            plane_prefetch(c, (int)l);
            planes_started = l + 1;
            if (per_plane) workers.v.emplace_back(code_group, l, l + 1);
            // the plane before this one is handed to the pool now (its histograms have had a quantizer launch's time to
            // arrive): its coder drains it while the stage goes on -- what a later plane of this call may be waiting for
            // if device memory is short (plane_prepare)
            if (pooled && l > 0) submit_plane(l - 1);
            return WR_OK;
        };
        // A plane that finds no device memory waits for chunks to come back -- and the planes this call has quantized before
        // it are chunks that can: with the stream synchronised (plane_buffer_wait) every one of them is complete, so they
        // all go to their coders before the wait begins instead of after the next quantizer launch.  (With fewer coder
        // threads than planes and no pool the coders only start when the number of planes is known: nothing drains early.)
        unsigned preparing = 0;
        const std::function<void()> before_wait = [&]() {
            for (unsigned k = 0; k < preparing; k++)
                if (plane_ready(k, false) != WR_OK) return;
            if (pooled) for (unsigned k = 0; k < preparing; k++) submit_plane(k);
        };
        auto plane_buf = [&](unsigned l) -> const wrk::PlaneRef* {
            preparing = l;
            return plane_prepare(c, (int)l, n, false, one_array, &cu, &before_wait) == WR_OK ? &c->ps[l].ref : nullptr;
        };
        {
            // ---- stage "kernels"
            cu.lock();
            clock_warmup(c, n);
            rc = encode_planes_core(c, slot.get(), d_fld, nx, ny, nz, wtflag, cut, plane_buf, hist_buf, info, &local, after_quant, plane_ready, &resid);
            if (hipStreamSynchronize(c->stream) != hipSuccess && rc == WR_OK) rc = fail(WR_ERR_HIP, "the encoder's kernel stage failed on the device" + launch_describe(c));
            if (rc == WR_OK && c->keep_residual && info->nlay && !fld.host && resid != fld.dev) {  // leave the residual where the reference leaves it
                if (hipMemcpyAsync(fld.dev, resid, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
                    hipStreamSynchronize(c->stream) != hipSuccess)
                    rc = fail(WR_ERR_HIP, "residual copy failed");
            }
            pool->last_stage_end.store(now());
            cu.unlock();
        }
        // ---- stage "down": the histograms were sent off as the planes completed; the residual follows them
        if (rc == WR_OK && c->keep_residual && info->nlay && fld.host) {
            rc = xfer_field(c, &c->x_field, fld.host, resid, n * sizeof(double), kDown);
        }
        // The slot's histogram buffer must not be reused before its downloads are done (the coder threads wait
        // for the same transfers; xfer_wait is safe to call from both sides).  With the coder pool, every plane
        // is handed over the moment its histograms are on the host.
        for (unsigned l = 0; l < planes_started; l++) {
            if (xfer_wait(&c->x_plane[l]) != WR_OK) copy_failed[l] = 1;
            if (pooled && rc == WR_OK && !copy_failed[l]) submit_plane(l);
        }
        t_gpu_done = now();
        // the slot goes back here: the planes are in device buffers of their own
    } catch (const std::exception& e) {
        workers.join();
        wrrc::pool_wait(&batch);
        return fail(WR_ERR_ARG, std::string("encode: ") + e.what());
    }
    if (pooled) {
        wrrc::pool_wait(&batch);
        for (unsigned l = 0; l < WR_NLAYMAX; l++) if (pool_mask >> l & 1) { lens[l] = jobs[l].result; coder_s[l] = jobs[l].seconds; }
    } else if (rc == WR_OK && !per_plane && info->nlay) {
        try {
            const unsigned groups = std::min<unsigned>(info->nlay, (unsigned)encoder_threads());
            for (unsigned g = 0; g < groups; g++)
                workers.v.emplace_back(code_group, g * info->nlay / groups, (g + 1) * info->nlay / groups);
        } catch (const std::exception& e) {
            workers.join();
            return fail(WR_ERR_ARG, std::string("encode: ") + e.what());
        }
    }
    workers.join();
    if (rc) return rc;
    for (unsigned l = 0; l < info->nlay; l++) {
        if (copy_failed[l] || c->ps[l].err) return fail(WR_ERR_HIP, "download of plane " + std::to_string(l) + " failed");
        if (lens[l] == (size_t)-1)  // (a refused window sets ps[l].err as well; what is left: histograms that are not the plane's)
            return fail(WR_ERR_HIP, "internal: the coder gave plane " + std::to_string(l) + " up (its block histograms are not its symbols')");
        local.d2h_ms += (float)(c->x_plane[l].ms + c->ps[l].copy_ms);
        if (verbose()) logs[l] = plane_log(c, (int)l, n, info, true, lens[l]);  // wrappers.cpp:401-409, 430
    }
    const double t_coded = now();
    // concatenate the plane streams (wrappers.cpp:412-427)
    size_t total = 0, offs[WR_NLAYMAX] = {0};
    for (unsigned l = 0; l < info->nlay; l++) {
        offs[l] = total;
        total += lens[l];
        info->len_enc_vec[l] = lens[l];
        local.plane_coder_s[l] = coder_s[l];
        if (coder_s[l] > local.rangecoder) local.rangecoder = coder_s[l];
    }
    for (unsigned l = 0; l < info->nlay; l++)  // (cannot happen: the bound is rigorous; if it did, a neighbour's bytes are gone)
        if (lens[l] > est_len[l]) return fail(WR_ERR_OVERFLOW, "internal: plane " + std::to_string(l) + " outgrew the bound computed from its histograms");
    if (total > cap) return fail(WR_ERR_OVERFLOW, "Error: encoded array is too large. Use larger SAFETY_BUFFER_FACTOR");
    // close the gaps: plane l moves down by what the planes before it stayed below their bounds.  In plane order, one after
    // the other: plane l's new place may still hold the end of plane l-1's old one.
    for (unsigned l = 0; l < info->nlay; l++) {
        if (!direct[l]) memcpy(data_enc + offs[l], c->enc_buf[l], lens[l]);
        else if (est_off[l] != offs[l]) memmove(data_enc + offs[l], data_enc + est_off[l], lens[l]);
    }
    // (a plane that went through c->enc_buf: hand its pages back, a noise plane's gigabyte would otherwise stay resident in
    // every context that once coded one)
    for (unsigned l = 0; l < info->nlay; l++) {
        if (direct[l]) continue;
        const uintptr_t a = ((uintptr_t)c->enc_buf[l] + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)c->enc_buf[l] + lens[l]) & ~(uintptr_t)4095;
        if (e > a && e - a >= ((size_t)64 << 20)) (void)madvise(reinterpret_cast<void*>(a), e - a, MADV_DONTNEED);
    }
    if (verbose())
        for (unsigned l = 0; l < info->nlay; l++) fputs(logs[l].c_str(), stdout);
    info->ntot_enc = total;
    local.total = now() - t0;
    local.wait = t_phase - t0;
    local.gpu = t_gpu_done - t_phase;  // without the wait for a slot
    local.transfer = (t_coded - t_gpu_done) - local.rangecoder;
    if (local.transfer < 0) local.transfer = 0;
    if (tm) *tm = local;
    return WR_OK;
}

// mode: the whole decode; or only its host half (range decoding, every decoded chunk going straight to the plane's device
// buffer: no field buffer and no slot needed, wr_decode_begin); or only its device half on planes decoded before
// (wr_decode_finish_*)
enum DecodeMode { kDecodeWhole, kDecodeBegin, kDecodeFinish };
// admission gate of the pooled decoder (decode_impl): queued jobs of a kind at which a further decode waits
constexpr int kGateScalarJobs = 3, kGateVectorJobs = 6;

int decode_impl(wr_ctx* c, FieldRef fld, int nx, int ny, int nz, const wr_enc_info* info, const unsigned char* data_enc,
                size_t data_len, wr_timings* tm, DecodeMode mode = kDecodeWhole)
{
    if (int rc = ctx_bind(c)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    ActiveCall active(c->pool);
    if (tm) wrdma::enable_timing();
    if (mode == kDecodeFinish) {
        if (!c->pend_valid) return fail(WR_ERR_ARG, "wr_decode_finish without a wr_decode_begin on this context");
        info = &c->pend_info; nx = c->pend_nx; ny = c->pend_ny; nz = c->pend_nz;
    }
    // (a finish that is refused for its arguments leaves the begin pending: the caller may try again with a usable
    // pointer, and the parked planes are not orphaned)
    if (int rc = check_dims(nx, ny, nz, fld.dev)) return rc;
    if (mode != kDecodeBegin && !fld.dev && !fld.host) return fail(WR_ERR_ARG, "null field pointer");
    if (mode == kDecodeFinish) c->pend_valid = false;
    // From here on the context's device planes go back on every way out, unless a begin parks them (a finish finds the
    // planes its begin parked; a begin or a whole decode discards what an earlier begin left).
    PlaneHold planes(c);
    const double t0 = now();
    const size_t n = (size_t)nx * ny * nz;
    wr_timings local; memset(&local, 0, sizeof local);
    if (mode == kDecodeFinish) local = c->pend_tm;
    DevPool* const pool = c->pool;
    if (mode == kDecodeBegin) { c->pend_info = *info; c->pend_nx = nx; c->pend_ny = ny; c->pend_nz = nz; }
    if (info->ntot_enc == 0) {  // wrappers.cpp:462-469
        if (mode == kDecodeBegin) { c->pend_tm = local; c->pend_valid = true; if (tm) *tm = local; return WR_OK; }
        if (fld.host) for (size_t j = 0; j < n; j++) fld.host[j] = info->midval;
        else { wrk::fill(fld.dev, n, info->midval, c->stream); HIPCHK(hipStreamSynchronize(c->stream)); }
        local.total += now() - t0;
        if (tm) *tm = local;
        return WR_OK;
    }
    const int nlay = info->nlay;
    if (nlay < 1 || nlay > WR_NLAYMAX) return fail(WR_ERR_ARG, "nlay out of range");
    if (info->wlev != 0 && info->wlev != kWavLvl) return fail(WR_ERR_ARG, "wlev must be 0 or 4");
    const bool host_half = mode != kDecodeFinish, device_half = mode != kDecodeBegin;
    if (host_half && verbose()) printf("Range decoding...\n");
    size_t off[WR_NLAYMAX + 1] = {0};
    for (int l = 0; l < nlay; l++) off[l + 1] = off[l] + info->len_enc_vec[l];
    if (off[nlay] > info->ntot_enc) return fail(WR_ERR_STREAM, "len_enc_vec exceeds ntot_enc");
    if (host_half && data_len && info->ntot_enc > data_len) return fail(WR_ERR_STREAM, "ntot_enc exceeds the length of the coded buffer");

    // Admission to the coder pool.  A decoder's planes take device memory from the moment they are prepared, and with every
    // session of the pool full a decode's jobs sat in the queues for seconds (4 of a decode's 15 s at 32 lanes) -- a third of
    // the decoders' plane memory was held by planes nobody was writing yet, and plane memory is what bounds the fields in
    // flight.  So a decode waits HERE, holding nothing, until the queues of its kind are short (a free lane of a session is
    // refilled at the next block boundary, half a millisecond away: a couple of queued jobs keep every session topped up),
    // and only then prepares its planes and submits: the same wait, without the memory.  One decode at a time passes, from
    // the look at the queues to the submit (else all that wait would pass together).
    std::unique_lock<std::mutex> gate(pool->planes.gate_mu, std::defer_lock);
    if (host_half && wrrc::pool_threads() > 0) {
        gate.lock();
        const double t_gate = now();
        for (;;) {
            int qs = 0, qv = 0;
            wrrc::pool_queued_decode(&qs, &qv);
            if ((qs < kGateScalarJobs && qv < kGateVectorJobs) || now() - t_gate > 120.0) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
        g_stat[WR_STAT_DECODE_GATE_MS] += (unsigned long)((now() - t_gate) * 1e3);
    }
    if (host_half) {
        c->pend_valid = false;  // whatever an earlier begin parked here is overwritten now
        // one decode at a time gathers its planes: a decoder keeps them all until its field is done, so two that each hold
        // half of theirs and wait for the other half would never finish
        const double t_turn = now();
        std::lock_guard<std::mutex> gather(pool->planes.gather_mu);
        const double t_got = now();
        if (t_got - t_turn > 1e-3) g_stat[WR_STAT_PLANE_WAIT_MS] += (unsigned long)((t_got - t_turn) * 1e3);
        for (int l = 0; l < nlay; l++) if (int rc = plane_prepare(c, l, n, true)) return rc;
    }
    for (int l = 0; l < nlay; l++)
        if (!c->ps[l].dev || c->ps[l].n != n) return fail(WR_ERR_ARG, "wr_decode_finish: the planes of the begin are gone");

    SlotNeed need;
    transform_need(nx, ny, nz, info->wlev ? -kWavLvl : 0, &need);
    if (fld.host) need.field_elems = n;

    size_t got[WR_NLAYMAX] = {0};
    double coder_s[WR_NLAYMAX] = {0};
    Sem sem(coder_threads());
    // one thread per plane, or (wr_set_threads) fewer threads with their planes interleaved, or the process-wide
    // coder pool (wr_set_coder_pool), whose workers interleave planes of several fields
    bool pooled = wrrc::pool_threads() > 0;
    int rc = WR_OK;
    double t_phase = 0, t_coded = t0;
    try {
        SlotLease slot;
        if (pooled && host_half) {
            wrrc::PlaneJob jobs[WR_NLAYMAX];
            wrrc::JobBatch batch;
            for (int l = 0; l < nlay; l++) {
                jobs[l].kind = wrrc::PlaneJob::kDecode;
                jobs[l].src = data_enc + off[l]; jobs[l].src_len = info->len_enc_vec[l]; jobs[l].dst = nullptr; jobs[l].io = &c->ps[l].io; jobs[l].n = n;
            }
            const bool queued = wrrc::pool_submit(jobs, nlay, &batch);
            if (gate.owns_lock()) gate.unlock();  // the next decode may look at the queues now
            if (queued) {
                wrrc::pool_wait(&batch);
                for (int l = 0; l < nlay; l++) { got[l] = jobs[l].result; coder_s[l] = jobs[l].seconds; }
            } else
                pooled = false;  // the pool was stopped meanwhile: this call's own threads decode the planes
        }
        if (gate.owns_lock()) gate.unlock();
        const int groups = (pooled || !host_half) ? 0 : std::min(nlay, coder_threads());
        // Every decoded window of a plane goes to the plane's device buffer while the decoder fills the next one
        // (chunk by chunk: wrappers.cpp:492-516 reorganised); the accumulate kernel consumes the
        // planes in plane order afterwards.
        if (host_half) g_stat[WR_STAT_EARLY_DECODES]++;
        {
            Workers workers;
            for (int g = 0; g < groups; g++)
                workers.v.emplace_back([&, g]() {
                    const int l0 = g * nlay / groups, l1 = (g + 1) * nlay / groups;
                    sem.acquire();
                    const double t = now();
                    const uint8_t* ins[WR_NLAYMAX];
                    uint8_t* syms[WR_NLAYMAX];
                    const wrrc::PlaneWindow* ios[WR_NLAYMAX];
                    for (int l = l0; l < l1; l++) { ins[l - l0] = data_enc + off[l]; syms[l - l0] = nullptr; ios[l - l0] = &c->ps[l].io; }
                    wrrc::decode_planes(l1 - l0, ins, info->len_enc_vec + l0, syms, n, got + l0, ios);
                    for (int l = l0; l < l1; l++) coder_s[l] = now() - t;
                    sem.release();
                });
        }
        int bad = -1;
        for (int l = 0; l < nlay && host_half; l++) {
            if (got[l] != n) bad = l;
            local.plane_coder_s[l] = coder_s[l];
            if (coder_s[l] > local.rangecoder) local.rangecoder = coder_s[l];
        }
        if (bad >= 0) return fail(WR_ERR_STREAM, "plane " + std::to_string(bad) + ": stream does not decode to nx*ny*nz symbols");
        if (host_half) {
            for (int l = 0; l < nlay; l++) {
                if (c->ps[l].err) return fail(WR_ERR_HIP, "upload of plane " + std::to_string(l) + " failed");
                local.h2d_ms += (float)c->ps[l].copy_ms;
            }
            t_coded = now();
            local.transfer = (t_coded - t0) - local.rangecoder;
            if (local.transfer < 0) local.transfer = 0;
        }
        if (!device_half) {  // the planes wait in the context's device buffers for wr_decode_finish_*
            local.total = now() - t0;
            c->pend_tm = local;
            c->pend_valid = true;
            planes.keep = true;
            if (tm) *tm = local;
            return WR_OK;
        }
        if (host_half && verbose()) {  // wrappers.cpp:489, 503-510
            for (int l = 0; l < nlay; l++) fputs(plane_log(c, l, n, info, false, 0).c_str(), stdout);
            printf("Wavelet reconstruction...\n");
        }
        if ((rc = slot.acquire(c, need)) != WR_OK) return rc;  // the planes are on the device already: no "up" stage
        t_phase = now();
        wrk::DequantParams p;
        memset(&p, 0, sizeof p);
        p.nlay = nlay;
        for (int l = 0; l < nlay; l++) {
            p.deps[l] = info->deps_vec[l]; p.minval[l] = info->minval_vec[l]; p.q[l] = c->ps[l].ref;
            if (!wrk::plane_ref_covers(p.q[l], n)) return fail(WR_ERR_HIP, "internal: the device buffer of plane " + std::to_string(l) + " has a hole");
        }
        launch_note(c, "dequant", nlay - 1, fld.host ? slot->field : fld.dev, n, nullptr, p.q[nlay - 1]);
        double* d_fld = fld.host ? slot->field : fld.dev;
        {
            // ---- stage "kernels"
            StageLock cu(pool->cu_mu);
            clock_warmup(c, n);
            rc = inverse_from_planes(c, slot.get(), d_fld, nx, ny, nz, (int)info->wlev, p);
            if (rc == WR_OK && hipGetLastError() != hipSuccess) rc = fail(WR_ERR_HIP, "kernel launch failed");
            if (hipStreamSynchronize(c->stream) != hipSuccess && rc == WR_OK) rc = fail(WR_ERR_HIP, "the decoder's kernel stage failed on the device" + launch_describe(c));
            pool->last_stage_end.store(now());
        }
        if (rc) return rc;
        if (fld.host) {
            // ---- stage "down": the reconstructed field, device -> host
            if ((rc = xfer_field(c, &c->x_field, fld.host, d_fld, n * sizeof(double), kDown)) != WR_OK) return rc;
            local.d2h_ms = (float)c->x_field.ms;
        }
    } catch (const std::exception& e) {
        return fail(WR_ERR_ARG, std::string("decode: ") + e.what());
    }
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev_a, c->ev_b)); local.quant_ms = ms;
    HIPCHK(hipEventElapsedTime(&ms, c->ev_b, c->ev_c)); local.transform_ms = ms;
    local.total += now() - t0;  // (a finish adds to what its begin took)
    local.gpu = now() - t_phase;  // without the wait for a slot
    local.wait = t_phase - t_coded;
    if (tm) *tm = local;
    return WR_OK;
}

}  // namespace

extern "C" {

int wr_encode_device(wr_ctx* c, double* d_fld, int nx, int ny, int nz, int wtflag, double tolrel,
                     wr_enc_info* info, unsigned char* data_enc, size_t cap, wr_timings* tm)
{
    Cutoff cut; cut.vec = &tolrel;
    FieldRef f; f.dev = d_fld;
    if (!d_fld) return fail(WR_ERR_ARG, "null device field pointer");
    return encode_impl(c, f, nx, ny, nz, wtflag, cut, info, data_enc, cap, tm);
}

int wr_encode_device_local(wr_ctx* c, double* d_fld, int nx, int ny, int nz, int wtflag, int mx, int my, int mz,
                           const double* cutoffvec, wr_enc_info* info, unsigned char* data_enc, size_t cap,
                           wr_timings* tm)
{
    Cutoff cut; cut.mx = mx; cut.my = my; cut.mz = mz; cut.vec = cutoffvec;
    FieldRef f; f.dev = d_fld;
    if (!d_fld) return fail(WR_ERR_ARG, "null device field pointer");
    return encode_impl(c, f, nx, ny, nz, wtflag, cut, info, data_enc, cap, tm);
}

int wr_decode_device(wr_ctx* c, double* d_fld, int nx, int ny, int nz, const wr_enc_info* info,
                     const unsigned char* data_enc, size_t data_len, wr_timings* tm)
{
    FieldRef f; f.dev = d_fld;
    if (!d_fld) return fail(WR_ERR_ARG, "null device field pointer");
    return decode_impl(c, f, nx, ny, nz, info, data_enc, data_len, tm);
}

int wr_encode_host(wr_ctx* c, double* h_fld, int nx, int ny, int nz, int wtflag, int mx, int my, int mz,
                   const double* cutoffvec, wr_enc_info* info, unsigned char* data_enc, size_t cap, wr_timings* tm)
{
    Cutoff cut; cut.mx = mx; cut.my = my; cut.mz = mz; cut.vec = cutoffvec;
    FieldRef f; f.host = h_fld;
    return encode_impl(c, f, nx, ny, nz, wtflag, cut, info, data_enc, cap, tm);
}

int wr_decode_host(wr_ctx* c, double* h_fld, int nx, int ny, int nz, const wr_enc_info* info,
                   const unsigned char* data_enc, size_t data_len, wr_timings* tm)
{
    FieldRef f; f.host = h_fld;
    return decode_impl(c, f, nx, ny, nz, info, data_enc, data_len, tm);
}

int wr_decode_begin(wr_ctx* c, int nx, int ny, int nz, const wr_enc_info* info, const unsigned char* data_enc, size_t data_len,
                    wr_timings* tm)
{
    FieldRef none;
    return decode_impl(c, none, nx, ny, nz, info, data_enc, data_len, tm, kDecodeBegin);
}

int wr_decode_finish_host(wr_ctx* c, double* h_fld, wr_timings* tm)
{
    FieldRef f; f.host = h_fld;
    if (!h_fld) return fail(WR_ERR_ARG, "null field pointer");
    return decode_impl(c, f, 0, 0, 0, nullptr, nullptr, 0, tm, kDecodeFinish);
}

int wr_decode_finish_device(wr_ctx* c, double* d_fld, wr_timings* tm)
{
    FieldRef f; f.dev = d_fld;
    if (!d_fld) return fail(WR_ERR_ARG, "null device field pointer");
    return decode_impl(c, f, 0, 0, 0, nullptr, nullptr, 0, tm, kDecodeFinish);
}

int wr_transform_host(wr_ctx* c, double* h_fld, int nx, int ny, int nz, int lvl)
{
    if (int rc = ctx_bind(c)) return rc;
    if (int rc = check_dims(nx, ny, nz, nullptr)) return rc;
    if (!h_fld) return fail(WR_ERR_ARG, "null field pointer");
    std::lock_guard<std::mutex> lk(c->mu);
    const size_t n = (size_t)nx * ny * nz;
    SlotNeed need;
    transform_need(nx, ny, nz, lvl, &need);
    need.field_elems = n;
    SlotLease slot;
    if (int rc = slot.acquire(c, need)) return rc;
    DevPool* const pool = c->pool;
    {
        if (int rc = xfer_field(c, &c->x_field, slot->field, h_fld, n * sizeof(double), kUp)) return rc;
    }
    double* res = nullptr;
    {
        StageLock cu(pool->cu_mu);
        if (int rc = run_transform(c, slot.get(), slot->field, nx, ny, nz, lvl, &res)) return rc;
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return xfer_field(c, &c->x_field, h_fld, res, n * sizeof(double), kDown);
}

}  // extern "C"
namespace {
// CPUs this process may use: its affinity mask, cut down to a cgroup CPU quota if there is one
int usable_cpus()
{
    cpu_set_t set;
    int n = 0;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    if (n < 1) n = (int)std::thread::hardware_concurrency();
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "max 100000" or "<quota> <period>"
        char q[64]; double period = 0;
        if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const int k = (int)(atof(q) / period + 0.5);
            if (k >= 1 && k < n) n = k;
        }
        fclose(f);
    }
    return n < 1 ? 1 : n;
}

size_t host_mem_available()
{
    size_t avail = 0;
    if (FILE* f = fopen("/proc/meminfo", "r")) {
        char line[256];
        while (fgets(line, sizeof line, f))
            if (strncmp(line, "MemAvailable:", 13) == 0) { avail = (size_t)strtoull(line + 13, nullptr, 10) * 1024; break; }
        fclose(f);
    }
    if (FILE* f = fopen("/sys/fs/cgroup/memory.max", "r")) {
        char q[64];
        if (fscanf(f, "%63s", q) == 1 && strcmp(q, "max") != 0) {
            const size_t lim = (size_t)strtoull(q, nullptr, 10);
            if (lim && (!avail || lim < avail)) avail = lim;
        }
        fclose(f);
    }
    return avail;
}
}  // namespace
extern "C" {

int wr_autotune_batch(size_t field_elems, int nfields)
{
    if (nfields < 1) nfields = 1;
    const int cpus = usable_cpus();
    if (nfields > 1 && cpus >= 2) wr_set_coder_pool(cpus, 0);
    // 1.5 fields in flight per CPU keep the pool's workers busy (a field spends part of its time in copies, kernels and
    // waiting for its slowest plane)
    long fit = (3L * cpus + 1) / 2;
    const double fb = 8.0 * (double)(field_elems ? field_elems : 1);
    // host: the caller's field and coded buffers plus the coder's output while it is produced: ~2.5 field sizes per call
    if (const size_t mem = host_mem_available()) { const long k = (long)(0.6 * (double)mem / (2.5 * fb)); if (k < fit) fit = k; }
    // device: three work-space slots of 2.2 field sizes; per call its quantized planes (1 byte per element and plane, 4-5
    // planes at the usual tolerances, 8 at most)
    int dev = 0;
    if (const char* e = getenv("WR_DEVICE")) dev = atoi(e);
    size_t free_b = 0, total_b = 0;
    if (hipSetDevice(dev) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const long k = (long)((0.9 * (double)free_b - 3 * 2.2 * fb) / (0.75 * fb));
        if (k < fit) fit = k;
    } else
        (void)hipGetLastError();
    if (fit > nfields) fit = nfields;
    return fit < 1 ? 1 : (int)fit;
}

int wr_bench_transform(wr_ctx* c, double* d_fld, int nx, int ny, int nz, int lvl, int reps, double* ms_out)
{
    if (int rc = ctx_bind(c)) return rc;
    if (int rc = check_dims(nx, ny, nz, d_fld)) return rc;
    if (reps < 1) return fail(WR_ERR_ARG, "reps < 1");
    SlotNeed need;
    transform_need(nx, ny, nz, lvl, &need);
    SlotLease slot;
    if (int rc = slot.acquire(c, need)) return rc;
    StageLock cu(c->pool->cu_mu);
    double* res = nullptr;
    HIPCHK(hipEventRecord(c->ev_a, c->stream));
    for (int r = 0; r < reps; r++)
        if (int rc = run_transform(c, slot.get(), d_fld, nx, ny, nz, lvl, &res)) return rc;  // fused: result stays in scratch
    HIPCHK(hipEventRecord(c->ev_b, c->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventSynchronize(c->ev_b));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev_a, c->ev_b));
    *ms_out = (double)ms / reps;
    return WR_OK;
}

}  // extern "C"
