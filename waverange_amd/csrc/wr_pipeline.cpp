// wr_pipeline.cpp -- host side of libwaverange_amd: contexts, work-space slots, transfers, device-resident planes (see wr_internal.h for the other files).
//
// A call (wr_codec.cpp) moves through three device stages on one of a few work-space SLOTS per GPU:
//   up      host -> device   field (encode)                                     SDMA engine (wr_dma.h)
//   kernels min/max, transform, quantizer / dequantizer                         the context's stream, one call
//                                                                               at a time (DevPool::cu_mu)
//   down    device -> host   block histograms, residual (encode) or field       SDMA engine
// The quantized planes are not part of the slot: they live in device buffers of their own (DevPlanes) and the host
// range coder (one thread per plane, fewer with the planes of a field interleaved in one loop: wr_set_threads, or the
// process-wide pool: wr_set_coder_pool) reads or writes them through a ring of two pinned 15 MB windows per plane
// (PlaneStream, wrrc::PlaneWindow) while the slot already serves the next field.  Copies and kernels are ordered from
// the host (HIP event of the producing kernel -> start the copy; signal of the copy -> launch the consumer); pageable
// caller memory goes through hipMemcpyAsync on a copy stream instead.
#include "wr_internal.h"

#include <signal.h>
#include <unistd.h>

namespace wri {

thread_local std::string g_err;
std::string& last_error() { return g_err; }
std::atomic<int> g_verbose{-1};      // -1: not initialised from the environment yet
std::atomic<int> g_threads{-1};      // -1: not initialised from the environment yet (WR_THREADS, default one per plane)
std::atomic<int> g_enc_threads{0};   // 0: same as g_threads (wr_set_encoder_threads)
std::atomic<unsigned long> g_stat[12];  // see wr_stat()
std::atomic<int> g_writeback{-1};    // drop-in encoding_wrap leaves the residual in fld_1d (-1: from WR_WRITEBACK_RESIDUAL, default 1)

int coder_threads()
{
    int t = g_threads.load();
    if (t < 0) {
        const char* e = getenv("WR_THREADS");
        const int k = e ? atoi(e) : 0;
        t = k >= 1 ? k : WR_NLAYMAX;
        g_threads.store(t);
    }
    return t;
}
int encoder_threads() { const int e = g_enc_threads.load(); return e > 0 ? e : coder_threads(); }

int verbose()
{
    int v = g_verbose.load();
    if (v < 0) {
        const char* q = getenv("WR_QUIET");
        v = (q && *q && *q != '0') ? 0 : 1;
        g_verbose.store(v);
    }
    return v;
}

int writeback_residual()
{
    int v = g_writeback.load();
    if (v < 0) {
        const char* e = getenv("WR_WRITEBACK_RESIDUAL");
        v = (e && *e) ? (atoi(e) ? 1 : 0) : 1;
        g_writeback.store(v);
    }
    return v;
}

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int ctx_bind(wr_ctx* c) { HIPCHK(hipSetDevice(c->device)); return WR_OK; }

DevPool g_pools[kMaxDevices];
std::mutex g_pools_mu;

hipError_t slot_ensure(Slot* s, const SlotNeed& need)
{
    hipError_t e;
    if ((e = grow(&s->field, &s->field_elems, need.field_elems)) != hipSuccess) return e;
    if ((e = grow(&s->scratch, &s->scratch_elems, need.scratch_elems)) != hipSuccess) return e;
    if ((e = grow(&s->lowbuf, &s->lowbuf_elems, need.lowbuf_elems)) != hipSuccess) return e;
    if ((e = grow(&s->hist, &s->hist_elems, need.hist_elems)) != hipSuccess) return e;
    return hipSuccess;
}

}  // namespace wri

// ---- device memory of the quantized planes ---------------------------------------------------------------------------
bool DevPlanes::alloc_fails_now()
{
    static const struct Hook {
        long first = -1, count = 0;
        Hook()
        {
            if (const char* e = getenv("WR_TEST_PLANE_ALLOC_FAIL")) {
                char* end = nullptr;
                first = strtol(e, &end, 10);
                count = (end && *end == ':') ? strtol(end + 1, nullptr, 10) : 1;
                if (first < 0 || count < 1) first = -1;
            }
        }
    } hook;
    static std::atomic<long> seq{0};
    if (hook.first < 0) return false;
    const long k = seq++;
    return k >= hook.first && k < hook.first + hook.count;
}

void* DevPlanes::device_alloc(size_t bytes, bool others_hold_planes)
{
    if (alloc_fails_now()) return nullptr;
    // The reserve only makes sense while somebody can give memory back: a caller whose own data fills the device to within
    // the reserve, with no other plane outstanding, would wait for nothing (and fail after five minutes where the hipMalloc
    // below succeeds).
    if (reserve_bytes && others_hold_planes) {  // (the calling thread has the context's device bound: ctx_bind)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) (void)hipGetLastError();
        else if (free_b < bytes + reserve_bytes) return nullptr;
    }
    void* q = nullptr;
    if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return q;
}

DevPlanes::Buf DevPlanes::take(size_t bytes, size_t held_by_caller)
{
    for (int attempt = 0; attempt < 2; attempt++) {
        bool any_idle = false, booked = false, others = false;
        {
            std::lock_guard<std::mutex> lk(mu);
            int best = -1;
            for (int i = 0; i < (int)idle.size(); i++)
                if (idle[i].bytes >= bytes && idle[i].bytes / 2 <= bytes && (best < 0 || idle[i].bytes < idle[best].bytes)) best = i;
            if (best >= 0) { Buf b = idle[best]; idle[best] = idle.back(); idle.pop_back(); return b; }
            any_idle = !idle.empty();
            if (!chunk_limit || allocated + bytes <= chunk_limit) {
                others = allocated > held_by_caller;  // planes of OTHER calls (in use or idle): something that can come back
                allocated += bytes;
                wri::g_stat[WR_STAT_DEVICE_PLANE_BYTES] += bytes;
                booked = true;
            }
        }
        if (booked) {
            if (void* q = device_alloc(bytes, others)) { Buf b; b.p = static_cast<uint8_t*>(q); b.bytes = bytes; return b; }
            std::lock_guard<std::mutex> lk(mu);
            allocated -= bytes;
            wri::g_stat[WR_STAT_DEVICE_PLANE_BYTES] -= bytes;
            any_idle = !idle.empty();
        }
        // no room: idle buffers of another size (another field size before) may be holding it
        if (!any_idle) break;
        drop_idle();
    }
    return Buf();
}

namespace wri {

// ---- what the plane kernels were launched with -----------------------------------------------------------------------
// A GPU memory fault ends the process inside the runtime (the queue's error callback calls abort()), so no error path of
// this library ever sees it.  Every launch of a kernel that works through a plane's chunk table is therefore noted in a
// process-wide ring beforehand; with WR_FAULT_LOG=1 a SIGABRT handler writes the ring to stderr before the process dies, so
// that a fault can be attributed to a launch, its pointers and the chunk table it was given.
namespace {
struct LaunchRecord {
    double t; const void* ctx; int device; char what[12]; int plane; const void* x; size_t n; const void* partial;
    unsigned shift; const void* chunk[wrk::kPlaneChunks];
};
constexpr unsigned kLaunchRing = 128;
LaunchRecord g_launches[kLaunchRing];
std::atomic<unsigned> g_launch_seq{0};

// (async-signal-safe: no stdio, no allocation -- digits by hand into a stack buffer, write(2))
struct Line {
    char b[768];
    size_t n = 0;
    void s(const char* t) { while (*t && n < sizeof b - 1) b[n++] = *t++; }
    void u(unsigned long long v) { char t[24]; int k = 0; do { t[k++] = (char)('0' + v % 10); v /= 10; } while (v); while (k && n < sizeof b - 1) b[n++] = t[--k]; }
    void x(const void* p) { s("0x"); const unsigned long long v = (unsigned long long)(uintptr_t)p; bool on = false; for (int sh = 60; sh >= 0; sh -= 4) { const unsigned d = (unsigned)(v >> sh) & 15u; if (d || on || !sh) { on = true; if (n < sizeof b - 1) b[n++] = "0123456789abcdef"[d]; } } }
    void out() { if (n < sizeof b) b[n++] = '\n'; (void)!write(2, b, n); n = 0; }
};
struct sigaction g_prev_abort;  // what the host application (or the runtime) had installed before us: it still runs

void fault_log_dump(int sig)
{
    Line l;
    const unsigned end = g_launch_seq.load();
    const unsigned begin = end > kLaunchRing ? end - kLaunchRing : 0;
    l.s("libwaverange_amd: abort -- the last "); l.u(end - begin); l.s(" plane-kernel launches (oldest first), planes hold ");
    l.u(g_stat[WR_STAT_DEVICE_PLANE_BYTES].load()); l.s(" bytes of device memory");
    l.out();
    for (unsigned k = begin; k < end; k++) {
        const LaunchRecord& r = g_launches[k % kLaunchRing];
        l.s("  #"); l.u(k); l.s(" t_us="); l.u((unsigned long long)(r.t * 1e6)); l.s(" ctx="); l.x(r.ctx); l.s(" dev="); l.u((unsigned)r.device);
        l.s(" "); l.s(r.what); l.s(" plane="); l.u((unsigned)r.plane); l.s(" x="); l.x(r.x); l.s(" n="); l.u(r.n); l.s(" partial="); l.x(r.partial);
        l.s(" shift="); l.u(r.shift); l.s(" chunks:");
        const size_t nch = r.shift >= 63 ? 1 : ((r.n ? r.n - 1 : 0) >> r.shift) + 1;
        for (size_t i = 0; i < nch && i < (size_t)wrk::kPlaneChunks; i++) { l.s(" "); l.x(r.chunk[i]); }
        l.out();
    }
    // then whoever was there before: the application's own handler, or the default action (the core dump)
    if (g_prev_abort.sa_handler != SIG_DFL && g_prev_abort.sa_handler != SIG_IGN && !(g_prev_abort.sa_flags & SA_SIGINFO)) {
        sigaction(SIGABRT, &g_prev_abort, nullptr);
        g_prev_abort.sa_handler(sig);
        return;
    }
    sigaction(SIGABRT, &g_prev_abort, nullptr);  // (an SA_SIGINFO handler wants arguments we do not have: it gets the re-raised signal)
    raise(SIGABRT);
}

const bool g_fault_log_installed = []() {
    const char* e = getenv("WR_FAULT_LOG");
    if (!(e && atoi(e))) return false;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = fault_log_dump;
    sigemptyset(&sa.sa_mask);
    return sigaction(SIGABRT, &sa, &g_prev_abort) == 0;
}();
}  // namespace

void launch_note(wr_ctx* c, const char* what, int plane, const void* x, size_t n, const void* partial, const wrk::PlaneRef& q)
{
    wr_ctx::LastLaunch& l = c->last_launch;
    l.what = what; l.plane = plane; l.x = x; l.n = n; l.partial = partial; l.q = q;
    LaunchRecord& r = g_launches[g_launch_seq.fetch_add(1) % kLaunchRing];
    r.t = now(); r.ctx = c; r.device = c->device; r.plane = plane; r.x = x; r.n = n; r.partial = partial; r.shift = q.shift;
    snprintf(r.what, sizeof r.what, "%s", what);
    for (int k = 0; k < wrk::kPlaneChunks; k++) r.chunk[k] = q.chunk[k];
}

std::string launch_describe(const wr_ctx* c)
{
    const wr_ctx::LastLaunch& l = c->last_launch;
    if (!l.what) return std::string();
    char b[512];
    int len = snprintf(b, sizeof b, " [last plane kernel: %s plane=%d x=%p n=%zu partial=%p shift=%u chunks", l.what, l.plane, l.x, l.n, l.partial, l.q.shift);
    const size_t nch = l.q.shift >= 63 ? 1 : ((l.n ? l.n - 1 : 0) >> l.q.shift) + 1;
    for (size_t i = 0; i < nch && i < (size_t)wrk::kPlaneChunks && len < (int)sizeof b - 24; i++) len += snprintf(b + len, sizeof b - (size_t)len, " %p", l.q.chunk[i]);
    return std::string(b, (size_t)len) + "]";
}

int ensure_enc_buf(wr_ctx* c, int l, size_t bytes)
{
    if (c->enc_buf_bytes[l] >= bytes) return WR_OK;
    free(c->enc_buf[l]);
    c->enc_buf[l] = static_cast<uint8_t*>(malloc(bytes));
    c->enc_buf_bytes[l] = c->enc_buf[l] ? bytes : 0;
    return c->enc_buf[l] ? WR_OK : fail(WR_ERR_ARG, "out of host memory for the coded stream");
}

int ensure_host_hist(wr_ctx* c, size_t elems)
{
    if (c->h_hist_elems >= elems) return WR_OK;
    if (c->h_hist) HIPCHK(hipHostFree(c->h_hist));
    c->h_hist = nullptr; c->h_hist_elems = 0;
    HIPCHK(hipHostMalloc(&c->h_hist, elems * sizeof(uint16_t), hipHostMallocDefault));
    c->h_hist_elems = elems;
    return WR_OK;
}

bool dma_enabled()
{
    static const bool on = !(getenv("WR_NO_DMA") && atoi(getenv("WR_NO_DMA")));
    return on;
}

// Starts the pieces of one transfer: on a DMA engine where ROCr knows the host memory (pinned), through
// hipMemcpyAsync on the device's copy stream of that direction otherwise.  Never blocks on the copies.
int xfer_start(wr_ctx* c, wr_ctx::Xfer* x, const Piece* pc, int count, Dir dir)
{
    bool able[4] = {false, false, false, false};
    int ndma = 0;
    const bool use = x->sig && dma_enabled();
    for (int k = 0; k < count; k++) { able[k] = use && wrdma::can_copy(pc[k].dst, pc[k].src); ndma += able[k]; }
    std::lock_guard<std::mutex> lk(x->mu);
    x->t_start = now();
    x->ms = 0;
    if (ndma) {
        wrdma::signal_arm(x->sig, ndma);
        x->dma_pending = true;
        int started = 0;
        for (int k = 0; k < count; k++) {
            if (!able[k]) continue;
            if (wrdma::copy_async(pc[k].dst, pc[k].src, pc[k].bytes, x->sig) != 0) {
                wrdma::signal_cancel(x->sig, ndma - started);
                return fail(WR_ERR_HIP, "DMA copy could not be queued");
            }
            started++;
        }
    }
    if (ndma < count) {
        DevPool* p = c->pool;
        hipStream_t st = dir == kUp ? p->up : p->down;
        StageLock lk(dir == kUp ? p->up_mu : p->down_mu);
        for (int k = 0; k < count; k++)
            if (!able[k])
                HIPCHK(hipMemcpyAsync(pc[k].dst, pc[k].src, pc[k].bytes, dir == kUp ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(x->ev, st));
        x->hip_pending = true;
    }
    return WR_OK;
}

// Waits for a transfer (no-op if none is pending).  May be called from any thread, any number of times.
int xfer_wait(wr_ctx::Xfer* x)
{
    std::lock_guard<std::mutex> lk(x->mu);
    int rc = WR_OK;
    const bool timed_by_host = x->hip_pending;
    if (x->hip_pending) {
        if (hipEventSynchronize(x->ev) != hipSuccess) rc = fail(WR_ERR_HIP, "copy failed");
        x->hip_pending = false;
    }
    if (x->dma_pending) {
        if (wrdma::wait(x->sig) != 0) rc = fail(WR_ERR_HIP, "DMA copy failed");
        x->dma_pending = false;
        const double ms = wrdma::last_copy_ms(x->sig);
        x->ms = (!timed_by_host && ms >= 0) ? ms : (now() - x->t_start) * 1e3;
    } else if (timed_by_host) {
        x->ms = (now() - x->t_start) * 1e3;
    }
    return rc;
}

// A whole field host <-> device: in pieces of 128 MB, one in flight.  The DMA engine of a direction serves its queue in
// order, so a field queued as one 8.6 GB copy (150 ms) holds up every 15 MB plane window queued behind it -- and a coder
// whose next window is late blocks with all the streams of its session: 0.87 of the pool's 16 workers were blocked that way
// on average (profiles/r05/u_*).  In pieces, a window waits for 2 ms of the field at most; the field pays a wake-up per
// piece (~2 %).
int xfer_field(wr_ctx* c, wr_ctx::Xfer* x, void* dst, const void* src, size_t bytes, Dir dir)
{
    constexpr size_t kFieldPiece = (size_t)128 << 20;
    double ms = 0;
    for (size_t off = 0; off < bytes; off += kFieldPiece) {
        const Piece pc = {static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, bytes - off < kFieldPiece ? bytes - off : kFieldPiece};
        if (int rc = xfer_start(c, x, &pc, 1, dir)) return rc;
        if (int rc = xfer_wait(x)) return rc;
        ms += x->ms;
    }
    x->ms = ms;
    return WR_OK;
}

// ---- device-resident planes and their host windows -------------------------------------------------------------------
// a window: 256 coder blocks, 15.36 MB, ~0.3 ms on a DMA engine (WR_WINDOW_BLOCKS: 1..256 blocks, for tests that want
// many windows on small fields)
const size_t kChunkSyms = []() {
    int blocks = 256;
    if (const char* e = getenv("WR_WINDOW_BLOCKS")) { const int v = atoi(e); if (v >= 1 && v <= 256) blocks = v; }
    return (size_t)blocks * wrrc::kBlock;
}();
const size_t kChunkBytes = 256 * (size_t)wrrc::kBlock + 64;  // ring buffers are always full-size

void plane_release(wr_ctx* c, int l)
{
    PlaneStream& s = c->ps[l];
    if (!s.dev) return;
    if (!s.chunks.empty()) {
        for (DevPlanes::Buf& b : s.chunks)
            if (b.p) { c->pool->planes.give(b); b.p = nullptr; }
        s.chunks.clear();
    } else {
        DevPlanes::Buf b; b.p = s.dev; b.bytes = s.dev_bytes;
        c->pool->planes.give(b);
    }
    s.dev = nullptr; s.dev_bytes = 0;
    s.released_chunks = 0; s.drain = false;
    s.ho.retire();  // window tickets cut for this plane are void from here on
    s.io.user = nullptr; s.io.window = nullptr;
}

namespace {

// `count` bytes of the plane from byte `first` on, against the host buffer `host`: one piece, or two where the range
// straddles a chunk boundary (a window is shorter than a chunk).  to_host: device -> host.
int plane_pieces(const PlaneStream& s, size_t first, size_t count, uint8_t* host, bool to_host, Piece out[4])
{
    int k = 0;
    while (count && k < 4) {
        const size_t room = s.chunks.empty() ? count : (((first >> s.ref.shift) + 1) << s.ref.shift) - first;
        const size_t len = count < room ? count : room;
        uint8_t* const dev = s.ref.at(first);
        out[k++] = to_host ? Piece{host, dev, len} : Piece{dev, host, len};
        first += len; host += len; count -= len;
    }
    return k;
}

void planes_configure(DevPlanes& dp)
{
    std::lock_guard<std::mutex> lk(dp.mu);
    if (dp.chunk_bytes) return;
    size_t mb = 32;
    if (const char* e = getenv("WR_PLANE_CHUNK_MB")) { const long v = atol(e); if (v >= 1 && v <= 65536) mb = (size_t)v; }
    size_t c = 1;
    while (c < (mb << 20)) c <<= 1;   // a power of two: the kernels find a byte's chunk by a shift
    dp.chunk_bytes = c;
    if (const char* e = getenv("WR_PLANE_LIMIT_MB")) { const long v = atol(e); if (v >= 1) dp.chunk_limit = (size_t)v << 20; }
    if (const char* e = getenv("WR_PLANE_RESERVE_MB")) { const long v = atol(e); if (v >= 0) dp.reserve_bytes = (size_t)v << 20; }
}

// A plane buffer of `bytes`; if the device (or WR_PLANE_LIMIT_MB) has no room, waits for one to come back -- other calls'
// encoders return their chunks as their coders advance, decoders when their field is done -- without the caller's kernel-stage
// lock, which those other calls may need to get there.
DevPlanes::Buf plane_buffer_wait(wr_ctx* c, size_t bytes, std::unique_lock<std::mutex>* unlock_while_waiting, bool* unlocked,
                                 const std::function<void()>* before_wait)
{
    DevPlanes& dp = c->pool->planes;
    const double t0 = now();
    bool waited = false;
    for (;;) {
        // (the cap is checked and booked there, idle buffers of other sizes make room; what this context holds itself does not
        // count as "somebody can give memory back": the reserve never makes a lone caller wait for its own planes)
        size_t mine = 0;
        for (int l = 0; l < WR_NLAYMAX; l++) {
            const PlaneStream& s = c->ps[l];
            if (!s.chunks.empty()) { for (const DevPlanes::Buf& q : s.chunks) if (q.p) mine += q.bytes; }
            else if (s.dev) mine += s.dev_bytes;
        }
        const DevPlanes::Buf b = dp.take(bytes, mine);
        if (b.p) { if (waited) g_stat[WR_STAT_PLANE_WAIT_MS] += (unsigned long)((now() - t0) * 1e3); return b; }
        if (!waited) {
            // what this call has queued must not straddle the gap in its kernel stage
            if (unlock_while_waiting || before_wait) (void)hipStreamSynchronize(c->stream);
            // the planes this call has quantized so far go to their coders now: they drain while it waits -- they may be
            // exactly what it is waiting for
            if (before_wait && *before_wait) (*before_wait)();
            if (unlock_while_waiting && !*unlocked) { unlock_while_waiting->unlock(); *unlocked = true; }
        }
        waited = true;
        std::unique_lock<std::mutex> lk(dp.mu);
        dp.cv.wait_for(lk, std::chrono::milliseconds(100));
        if (now() - t0 > 300.0) return DevPlanes::Buf();
    }
}

}  // namespace

namespace {

// A window callback that must not be served -- its ticket is of an earlier generation of the plane (a coder that outlived
// its call), it is out of order, it comes after the stream's end, or another callback is inside the same stream -- gets
// NOTHING: a null window tells the coder to give its stream up at once (wrrc::PlaneWindow; its job ends with (size_t)-1
// and the call it belongs to fails), and the plane of the call that owns the stream now is not touched.  (Until round 4
// it got a zeroed scratch window "to run on harmlessly" -- against the GPU's block histograms, in which the count of
// symbol 0 is normally zero, an encoder on that window had a range of zero and never came back.)
uint8_t* refused_window(PlaneStream* s, const char* who, const char* why, size_t first, size_t* count)
{
    static std::mutex mu;
    g_stat[WR_STAT_HANDOVER_ERRORS]++;
    if (s) s->err = 1;
    {
        std::lock_guard<std::mutex> lk(mu);
        static int reported = 0;
        if (reported++ < 8)
            fprintf(stderr, "libwaverange_amd: plane hand-over violated in %s: %s (first=%zu count=%zu plane stream %p) -- window refused\n", who, why,
                    first, *count, (void*)s);
    }
    return nullptr;
}

}  // namespace

namespace {
// xfer_wait on behalf of a host coder inside its window request: what it waits is booked (WR_STAT_WINDOW_WAIT_MS; kept in
// microseconds, so that many short waits add up)
std::atomic<unsigned long long> g_window_wait_us{0};
int window_xfer_wait(wr_ctx::Xfer* x)
{
    const double t0 = now();
    const int rc = xfer_wait(x);
    const double us = (now() - t0) * 1e6;
    if (us >= 20.0) g_window_wait_us += (unsigned long long)us;
    return rc;
}
}  // namespace

// Encoder side: the symbols [first, first + count) of the plane, fetched into the ring; the following chunk is
// started into the buffer the coder has just left, so that it arrives while this one is being coded.
uint8_t* plane_window_encode(void* user, size_t first, size_t* count)
{
    const PlaneStream::Ticket* const tk = static_cast<const PlaneStream::Ticket*>(user);
    PlaneStream& s = *tk->s;
    if (!s.ho.current(tk->gen)) return refused_window(nullptr, "plane_window_encode", "the ticket is of an earlier generation of this plane", first, count);
    HandoverGuard guard(s.ho);
    if (!guard.alone) return refused_window(&s, "plane_window_encode", "two coders inside one plane stream", first, count);
    if (const char* why = s.ho.check_encode(first, *count)) return refused_window(&s, "plane_window_encode", why, first, count);
    wr_ctx* const c = s.c;
    (void)hipSetDevice(c->device);  // coder threads: the pageable-copy fallback of xfer_start needs the device bound
    const size_t want = *count < kChunkSyms ? *count : kChunkSyms;
    const int b = s.cur ^ 1;
    Piece pc[4];
    if (!(s.ahead && s.ahead_first == first)) {
        if (s.ahead) (void)xfer_wait(&s.x[b]);
        const int np = plane_pieces(s, first, want, s.buf[b], true, pc);
        if (xfer_start(c, &s.x[b], pc, np, kDown) != WR_OK) s.err = 1;
    }
    if (window_xfer_wait(&s.x[b]) != WR_OK) s.err = 1;
    s.copy_ms += s.x[b].ms;
    s.cur = b;
    s.ahead = false;
    if (s.drain) {
        // everything below `first` is coded, the window from `first` on is on the host: the chunks that lie wholly below `first`
        // go back to the pool (the copy that is started next reads from first + want on)
        const size_t upto = first >> s.ref.shift;
        for (size_t k = s.released_chunks; k < upto && k < s.chunks.size(); k++)
            if (s.chunks[k].p) { c->pool->planes.give(s.chunks[k]); s.chunks[k].p = nullptr; s.ref.chunk[k] = nullptr; }
        if (upto > s.released_chunks) s.released_chunks = upto;
    }
    const size_t next = first + want;
    if (next < s.n) {
        const int np = plane_pieces(s, next, s.n - next < kChunkSyms ? s.n - next : kChunkSyms, s.buf[b ^ 1], true, pc);
        if (xfer_start(c, &s.x[b ^ 1], pc, np, kDown) == WR_OK) { s.ahead = true; s.ahead_first = next; }
        else s.err = 1;
    }
    s.ho.served(first, want, true);
    *count = want;
    return s.buf[b];
}

// Decoder side: room for the symbols from `first` on; the window handed out before goes to the device meanwhile.
// *count == 0 ends the stream: both uploads are waited for.
uint8_t* plane_window_decode(void* user, size_t first, size_t* count)
{
    const PlaneStream::Ticket* const tk = static_cast<const PlaneStream::Ticket*>(user);
    PlaneStream& s = *tk->s;
    if (!s.ho.current(tk->gen)) return refused_window(nullptr, "plane_window_decode", "the ticket is of an earlier generation of this plane", first, count);
    HandoverGuard guard(s.ho);
    if (!guard.alone) return refused_window(&s, "plane_window_decode", "two coders inside one plane stream", first, count);
    if (const char* why = s.ho.check_decode(first, *count)) return refused_window(&s, "plane_window_decode", why, first, count);
    wr_ctx* const c = s.c;
    (void)hipSetDevice(c->device);
    if (s.win_count) {
        Piece pc[4];
        const int np = plane_pieces(s, s.win_first, s.win_count, s.buf[s.cur], false, pc);
        if (xfer_start(c, &s.x[s.cur], pc, np, kUp) != WR_OK) s.err = 1;
        s.win_count = 0;
    }
    if (*count == 0) {
        for (int b = 0; b < 2; b++) { if (xfer_wait(&s.x[b]) != WR_OK) s.err = 1; s.copy_ms += s.x[b].ms; s.x[b].ms = 0; }
        s.ho.end();
        return nullptr;
    }
    const int b = s.cur ^ 1;
    if (window_xfer_wait(&s.x[b]) != WR_OK) s.err = 1;  // the upload of two windows ago
    s.copy_ms += s.x[b].ms; s.x[b].ms = 0;
    s.cur = b;
    s.win_first = first;
    s.win_count = *count < kChunkSyms ? *count : kChunkSyms;
    s.ho.served(first, s.win_count, false);
    *count = s.win_count;
    return s.buf[b];
}

// plane l of n symbols for this call: a device buffer (kept if the context holds one that fits: a finish after a
// begin), the ring, and the window callbacks of the direction
int plane_prepare(wr_ctx* c, int l, size_t n, bool decode, bool contiguous, std::unique_lock<std::mutex>* unlock_while_waiting,
                  const std::function<void()>* before_wait)
{
    PlaneStream& s = c->ps[l];
    const size_t bytes = wr_plane_pitch(n);
    DevPlanes& dp = c->pool->planes;
    planes_configure(dp);
    // a plane of two chunks or more lives in chunks: of the configured size, or larger so that kPlaneChunks of them hold it
    size_t cb = dp.chunk_bytes;
    while (cb < bytes && cb * wrk::kPlaneChunks < bytes) cb <<= 1;
    const bool want_chunks = !contiguous && bytes >= 2 * cb && cb >= kChunkSyms;  // (a window never spans more than two chunks)
    const bool have_chunks = !s.chunks.empty();
    const bool reusable = s.dev && s.dev_bytes >= bytes && have_chunks == want_chunks && !s.released_chunks;
    if (!reusable) {
        plane_release(c, l);
        bool unlocked = false;
        const char* const no_room = "out of device memory for a quantized plane: nothing came back in five minutes (fewer calls in flight need less)";
        if (want_chunks) {
            const size_t nch = (bytes + cb - 1) / cb;
            unsigned shift = 0;
            while (((size_t)1 << shift) < cb) shift++;
            s.ref.shift = shift;
            for (int k = 0; k < wrk::kPlaneChunks; k++) s.ref.chunk[k] = nullptr;
            s.chunks.assign(nch, DevPlanes::Buf());
            for (size_t k = 0; k < nch; k++) {
                const DevPlanes::Buf b = plane_buffer_wait(c, cb, unlock_while_waiting, &unlocked, before_wait);
                if (!b.p) {
                    for (DevPlanes::Buf& q : s.chunks) if (q.p) dp.give(q);
                    s.chunks.clear();
                    if (unlocked) unlock_while_waiting->lock();
                    return fail(WR_ERR_HIP, no_room);
                }
                s.chunks[k] = b; s.ref.chunk[k] = b.p;
            }
            s.dev = s.chunks[0].p; s.dev_bytes = nch * cb;
        } else {
            const DevPlanes::Buf b = plane_buffer_wait(c, bytes, unlock_while_waiting, &unlocked, before_wait);
            if (!b.p) { if (unlocked) unlock_while_waiting->lock(); return fail(WR_ERR_HIP, no_room); }
            s.dev = b.p; s.dev_bytes = b.bytes;
            s.ref = wrk::plane_ref(b.p);
        }
        if (unlocked) unlock_while_waiting->lock();
    }
    // an encoder's plane drains: its chunks go back as the coder has fetched the windows they hold -- unless the plane is
    // looked at again afterwards (the verbose mode's per-plane diagnostics, plane_log)
    s.drain = !s.chunks.empty() && !decode && !verbose();
    for (int b = 0; b < 2; b++) {
        if (s.buf[b]) continue;
        if (hipHostMalloc(reinterpret_cast<void**>(&s.buf[b]), kChunkBytes, hipHostMallocDefault) == hipSuccess) { s.buf_pinned[b] = true; continue; }
        // no pinned memory left: a pageable window works (the copies are then staged by the runtime, xfer_start)
        (void)hipGetLastError();
        s.buf[b] = static_cast<uint8_t*>(aligned_alloc(4096, (kChunkBytes + 4095) / 4096 * 4096));
        s.buf_pinned[b] = false;
        if (!s.buf[b]) return fail(WR_ERR_ARG, "out of host memory for a plane window");
    }
    s.c = c; s.n = n;
    s.cur = 1; s.win_first = s.win_count = 0; s.ahead = false; s.ahead_first = 0; s.err = 0; s.copy_ms = 0;
    s.x[0].ms = s.x[1].ms = 0;
    // a new generation of the plane: the coder's handle is a ticket for exactly this one
    const uint64_t g = s.ho.begin(n);
    PlaneStream::Ticket& tk = s.tickets[s.ticket_seq++ % 8];  // (a coder would have to be eight calls late to meet its ticket re-cut)
    tk.s = &s; tk.gen = g;
    s.io.window = decode ? plane_window_decode : plane_window_encode;
    s.io.user = &tk;
    return WR_OK;
}

// encode: the plane is complete on the device -- its first chunk sets off for the host before a coder asks for it
void plane_prefetch(wr_ctx* c, int l)
{
    PlaneStream& s = c->ps[l];
    if (!s.n) return;
    Piece pc[4];
    const int np = plane_pieces(s, 0, s.n < kChunkSyms ? s.n : kChunkSyms, s.buf[s.cur ^ 1], true, pc);
    if (xfer_start(c, &s.x[s.cur ^ 1], pc, np, kDown) == WR_OK) { s.ahead = true; s.ahead_first = 0; }
    else s.err = 1;
}

// a whole plane on the host, for the diagnostics of the verbose mode (wrappers.cpp:401-409, 503-510)
std::string plane_log(wr_ctx* c, int l, size_t n, const wr_enc_info* info, bool encode, size_t len)
{
    std::vector<uint8_t> q(n);
    {
        const PlaneStream& s = c->ps[l];
        for (size_t at = 0; at < n;) {  // chunk by chunk (the verbose mode keeps an encoder's plane whole: plane_prepare)
            const size_t room = s.chunks.empty() ? n - at : (((at >> s.ref.shift) + 1) << s.ref.shift) - at;
            const size_t len = n - at < room ? n - at : room;
            if (hipMemcpy(q.data() + at, s.ref.at(at), len, hipMemcpyDeviceToHost) != hipSuccess) return std::string();
            at += len;
        }
    }
    unsigned lo = q[0], hi = q[0];
    for (size_t j = 1; j < n; j++) { unsigned v = q[j]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
    char b[256];
    if (encode)
        snprintf(b, sizeof b, "imin=%u imax=%u med=%g\nlen_out_q=%lu ntot=%lu\n", lo, hi, q[n / 2] * info->deps_vec[l] + info->minval_vec[l],
                 (unsigned long)len, (unsigned long)n);
    else
        snprintf(b, sizeof b, "ilay=%d\nimin=%u imax=%u med=%g\n", l, lo, hi, q[n / 2] * info->deps_vec[l] + info->minval_vec[l]);
    return b;
}

bool use_fused(int nx, int ny, int nz, int lvl)
{
    return wrk::fused_ok(nx, ny, nz, lvl) && !getenv("WR_NO_FUSED");
}

// work space a transform of this shape needs next to the field itself
void transform_need(int nx, int ny, int nz, int lvl, SlotNeed* need)
{
    need->scratch_elems = (size_t)nx * ny * nz;
    if (use_fused(nx, ny, nz, lvl)) need->lowbuf_elems = wrk::fused_lowbuf_elems(nx, ny, nz);
}

// Forward (lvl > 0) or inverse (lvl < 0) transform of d_fld.  The fused path is out of place: the result
// lands in the slot's scratch buffer and *out points there; the general path works in place.
int run_transform(wr_ctx* c, Slot* s, double* d_fld, int nx, int ny, int nz, int lvl, double** out)
{
    *out = d_fld;
    if (use_fused(nx, ny, nz, lvl)) {
        if (const char* why = wrk::fused_prepare()) return fail(WR_ERR_HIP, why);
        if (lvl > 0) wrk::transform_fwd_fused(d_fld, s->scratch, s->lowbuf, nx, ny, nz, c->stream);
        else wrk::transform_inv_fused(d_fld, s->scratch, s->lowbuf, nx, ny, nz, c->stream);
        *out = s->scratch;
    } else {
        wrk::transform(d_fld, s->scratch, nx, ny, nz, lvl, c->stream);
    }
    return WR_OK;
}

// decoder back end: acc = sum of planes, then the inverse transform, result in d_fld.
// Fused path: accumulate into the scratch buffer and transform out of place into d_fld.
// Records ev_a / ev_b / ev_c around the two stages (for the timings).
int inverse_from_planes(wr_ctx* c, Slot* s, double* d_fld, int nx, int ny, int nz, int wlev, const wrk::DequantParams& p)
{
    const size_t n = (size_t)nx * ny * nz;
    const bool fused = wlev == 4 && use_fused(nx, ny, nz, -4);
    if (fused) if (const char* why = wrk::fused_prepare()) return fail(WR_ERR_HIP, why);
    HIPCHK(hipEventRecord(c->ev_a, c->stream));
    wrk::dequant_accum(fused ? s->scratch : d_fld, n, p, c->stream);
    HIPCHK(hipEventRecord(c->ev_b, c->stream));
    if (fused) wrk::transform_inv_fused(s->scratch, d_fld, s->lowbuf, nx, ny, nz, c->stream);
    else wrk::transform(d_fld, s->scratch, nx, ny, nz, -wlev, c->stream);
    HIPCHK(hipEventRecord(c->ev_c, c->stream));
    return WR_OK;
}

// min/max of a device array with the reference's scan semantics (wrappers.cpp:244-250):
// values from the reduction; if the minimum is a zero, its sign is that of the LAST zero in
// memory order (glibc fmin keeps the later of equal operands -- oracle/wr_oracle.c:wro_minmax).
// `pending` = a fused kernel has already been enqueued that stores the reduction into h_result[0..1].
// Waits on an event recorded right behind the read-back, so kernels enqueued afterwards do not
// delay the answer.
int read_minmax(wr_ctx* c, const double* d_x, size_t n, bool pending, double* mn, double* mx)
{
    // the final reduction kernel stores min and max straight into pinned host memory: no copy command
    if (!pending) wrk::minmax(d_x, n, c->d_partial, c->h_result_dev, c->stream);
    HIPCHK(hipEventRecord(c->ev_mm, c->stream));
    HIPCHK(hipEventSynchronize(c->ev_mm));
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

// The shader clock of an idle GPU is down, and it takes ~30 ms of load to come all the way up -- longer than a whole
// kernel stage; the transforms (half VALU issue) run 20-30 % slower at the start of a stage than back to back
// (tools/clock_burn.py, profiles/r04/c_clock_burner_before_the_transform.txt: forward 4.91 ms cold, 3.77 after 10 ms of fp64
// arithmetic on every CU, 3.68 after 20 ms, 3.64 warm; waves that merely occupy the CUs asleep do nothing for the clock).
// MEASUREMENT HOOK, OFF BY DEFAULT (WR_CLOCK_WARMUP_MS=k, k <= 200): a stage that finds the GPU idle puts k ms of fp64 load on
// every CU in front of its first kernel.  It shows what the transform kernels do at full clock inside the pipeline (4.69 /
// 5.09 -> 4.14 / 4.64 ms at k = 15) and it buys NOTHING: the whole-job rate is set by the host coder and did not move
// (14.56 against 14.54 GB/s, profiles/r04/b_ and d_bench_k8_*), while the burner was 55-75 % of all GPU kernel time and
// stretched a 17 ms kernel stage to 57 ms (round 4 shipped it on at 40 ms; the review was right to call that buying a
// fraction with a busy-loop).  Called with DevPool::cu_mu held: nothing else computes on the device meanwhile.
void clock_warmup(wr_ctx* c, size_t n)
{
    static const double ms = []() { const char* e = getenv("WR_CLOCK_WARMUP_MS"); const double v = e ? atof(e) : 0.0; return v < 0 ? 0.0 : (v > 200 ? 200.0 : v); }();
    if (ms <= 0) return;
    DevPool* p = c->pool;
    (void)n;
    // (profiles/r04/c_clock_burner_before_the_transform.txt: a forward transform after half a second of idle 4.91 ms, behind
    // 10 / 20 / 40 / 80 ms of load 3.77 / 3.68 / 3.64 / 3.65)
    if (now() - p->last_stage_end.load() < 0.004) return;  // a stage has just ended: the clock is up
    wrk::burn(ms, 0, 1024, c->d_partial, c->stream);
    (void)hipGetLastError();
    g_stat[WR_STAT_CLOCK_WARMUP_MS] += (unsigned long)ms;
}

int check_dims(int nx, int ny, int nz, const void* dev_ptr)
{
    if (nx < 1 || ny < 1 || nz < 1) return fail(WR_ERR_ARG, "non-positive dimension");
    if (((uintptr_t)dev_ptr) & 15) return fail(WR_ERR_ARG, "device field pointer must be 16-byte aligned");
    return WR_OK;
}

}  // namespace wri

using namespace wri;

// =====================================================================================
// Part 2: device-resident API -- settings, contexts, memory helpers, stage-level entry points
// =====================================================================================
extern "C" {

const char* wr_last_error(void) { return last_error().c_str(); }

int wr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void wr_set_verbosity(int level) { g_verbose.store(level ? 1 : 0); }
void wr_set_threads(int nthreads) { g_threads.store(nthreads < 1 ? 1 : nthreads); g_enc_threads.store(0); }
void wr_set_encoder_threads(int nthreads) { g_enc_threads.store(nthreads < 0 ? 0 : nthreads); }
void wr_set_writeback_residual(int on) { g_writeback.store(on ? 1 : 0); }
void wr_pool_loop_stats(double* seconds, double* blocks) { wrrc::pool_loop_stats(seconds, blocks); }

int wr_ctx_trim(wr_ctx* c)
{
    if (int rc = ctx_bind(c)) return rc;
    c->pool->planes.drop_idle();
    return WR_OK;
}

unsigned long wr_stat(int what)
{
    if (what == WR_STAT_POOL_IDLE_MS) return (unsigned long)(wrrc::pool_idle_seconds() * 1e3);
    if (what == WR_STAT_POOL_STREAMS_MOVED) return wrrc::pool_streams_moved();
    if (what == WR_STAT_POOL_QUEUE_MS) return (unsigned long)(wrrc::pool_queue_seconds() * 1e3);
    if (what == WR_STAT_WINDOW_WAIT_MS) return (unsigned long)(g_window_wait_us.load() / 1000);
    return (what >= 0 && what < 12) ? g_stat[what].load() : 0;
}
void wr_set_coder_pool(int nthreads, int decoder_streams)
{
    wrrc::pool_configure(nthreads < 0 ? 0 : nthreads, decoder_streams);
}

int wr_set_device_slots(int device, int nslots)
{
    if (device < 0 || device >= kMaxDevices) return fail(WR_ERR_ARG, "device index out of range");
    if (nslots < 1) nslots = 1;
    if (nslots > kMaxSlots) nslots = kMaxSlots;
    DevPool* p = &g_pools[device];
    std::lock_guard<std::mutex> lk(p->mu);
    for (int i = nslots; i < kMaxSlots; i++)
        if (p->slots[i].busy || p->slots[i].allocated()) return fail(WR_ERR_ARG, "slots beyond the new count are in use");
    p->max_slots = nslots;
    for (Slot& s : p->slots) s.disabled = false;
    return WR_OK;
}

static int ctx_init(wr_ctx* c, int device, void* hip_stream)
{
    HIPCHK(hipSetDevice(device));
    {
        std::lock_guard<std::mutex> lk(g_pools_mu);
        DevPool* p = c->pool;
        if (!p->up) {
            if (const char* e = getenv("WR_SLOTS")) { const int k = atoi(e); if (k >= 1) p->max_slots = k > kMaxSlots ? kMaxSlots : k; }
            HIPCHK(hipStreamCreateWithFlags(&p->up, hipStreamNonBlocking));
        }
        if (!p->down) HIPCHK(hipStreamCreateWithFlags(&p->down, hipStreamNonBlocking));
    }
    if (hip_stream) c->stream = (hipStream_t)hip_stream;
    else { HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    HIPCHK(hipMalloc(&c->d_partial, 2 * wrk::minmax_partials() * sizeof(double)));
    HIPCHK(hipMalloc(&c->d_idx, sizeof(unsigned long long)));
    HIPCHK(hipHostMalloc(&c->h_result, 8 * sizeof(double), hipHostMallocDefault));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_result_dev), c->h_result, 0));
    for (int i = 0; i < WR_NLAYMAX; i++) {
        HIPCHK(hipEventCreateWithFlags(&c->ev_plane[i], hipEventDisableTiming));
    }
    HIPCHK(hipEventCreate(&c->ev_a)); HIPCHK(hipEventCreate(&c->ev_b));
    HIPCHK(hipEventCreate(&c->ev_c)); HIPCHK(hipEventCreate(&c->ev_d));
    HIPCHK(hipEventCreateWithFlags(&c->ev_mm, hipEventDisableTiming));
    std::vector<wr_ctx::Xfer*> xs = {&c->x_field};
    for (int l = 0; l < WR_NLAYMAX; l++) { xs.push_back(&c->x_plane[l]); xs.push_back(&c->ps[l].x[0]); xs.push_back(&c->ps[l].x[1]); }
    for (wr_ctx::Xfer* x : xs) {
        x->sig = wrdma::signal_create();  // 0 if ROCr is not usable: every copy then goes through hipMemcpyAsync
        HIPCHK(hipEventCreateWithFlags(&x->ev, hipEventDisableTiming));
    }
    return WR_OK;
}

int wr_ctx_create(wr_ctx** out, int device, void* hip_stream)
{
    if (!out) return fail(WR_ERR_ARG, "null ctx pointer");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail(WR_ERR_HIP, std::string("no usable HIP device (") + hipGetErrorString(e) +
                                    "): libwaverange_amd has no CPU fallback");
    if (device < 0 || device >= ndev || device >= kMaxDevices) return fail(WR_ERR_ARG, "device index out of range");
    wr_ctx* c = new (std::nothrow) wr_ctx;
    if (!c) return fail(WR_ERR_ARG, "out of host memory");
    c->device = device;
    c->pool = &g_pools[device];
    { std::lock_guard<std::mutex> lk(g_pools_mu); c->pool->users++; }
    const int rc = ctx_init(c, device, hip_stream);
    if (rc != WR_OK) {
        const std::string msg = g_err;
        wr_ctx_destroy(c);  // releases whatever was created, and the pool reference
        g_err = msg;
        return rc;
    }
    *out = c;
    return WR_OK;
}

void wr_ctx_destroy(wr_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->d_cutoff);
    (void)hipFree(c->d_partial); (void)hipFree(c->d_idx); (void)hipFree(c->d_mm);
    if (c->h_result) (void)hipHostFree(c->h_result);
    if (c->h_hist) (void)hipHostFree(c->h_hist);
    for (int l = 0; l < WR_NLAYMAX; l++) {
        (void)xfer_wait(&c->ps[l].x[0]); (void)xfer_wait(&c->ps[l].x[1]);
        plane_release(c, l);
        for (int b = 0; b < 2; b++)
            if (c->ps[l].buf[b]) { if (c->ps[l].buf_pinned[b]) (void)hipHostFree(c->ps[l].buf[b]); else free(c->ps[l].buf[b]); }
        free(c->enc_buf[l]);
    }
    for (int i = 0; i < WR_NLAYMAX; i++) {
        if (c->ev_plane[i]) (void)hipEventDestroy(c->ev_plane[i]);
    }
    for (hipEvent_t ev : {c->ev_a, c->ev_b, c->ev_c, c->ev_d, c->ev_mm})
        if (ev) (void)hipEventDestroy(ev);
    std::vector<wr_ctx::Xfer*> xs = {&c->x_field};
    for (int l = 0; l < WR_NLAYMAX; l++) { xs.push_back(&c->x_plane[l]); xs.push_back(&c->ps[l].x[0]); xs.push_back(&c->ps[l].x[1]); }
    for (wr_ctx::Xfer* x : xs) {
        wrdma::signal_destroy(x->sig);
        if (x->ev) (void)hipEventDestroy(x->ev);
    }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    {   // the last context on a device releases the shared work space
        std::lock_guard<std::mutex> lk(g_pools_mu);
        DevPool* p = c->pool;
        if (--p->users == 0) {
            std::lock_guard<std::mutex> sl(p->mu);
            if (p->up) { (void)hipStreamSynchronize(p->up); (void)hipStreamDestroy(p->up); p->up = nullptr; }
            if (p->down) { (void)hipStreamSynchronize(p->down); (void)hipStreamDestroy(p->down); p->down = nullptr; }
            for (Slot& s : p->slots) s.release_buffers();
            p->planes.drop_idle();
        }
    }
    delete c;
}

int wr_ctx_sync(wr_ctx* c)
{
    if (int rc = ctx_bind(c)) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
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

int wr_host_alloc(void** ptr, size_t bytes)
{
    HIPCHK(hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocDefault));
    return WR_OK;
}

int wr_host_free(void* ptr)
{
    HIPCHK(hipHostFree(ptr));
    return WR_OK;
}

int wr_host_register(void* ptr, size_t bytes)
{
    if (!ptr || !bytes) return fail(WR_ERR_ARG, "wr_host_register: empty range");
    HIPCHK(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return WR_OK;
}

int wr_host_unregister(void* ptr)
{
    HIPCHK(hipHostUnregister(ptr));
    return WR_OK;
}

int wr_dev_upload(wr_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (int rc = ctx_bind(c)) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));  // ordered behind what the context has queued, like a stream copy
    if (bytes >= (1u << 20) && c->x_field.sig && wrdma::can_copy(dst, src)) {  // pinned host memory: SDMA engine
        std::lock_guard<std::mutex> lk(c->mu);
        wrdma::signal_arm(c->x_field.sig, 1);
        if (wrdma::copy_async(dst, src, bytes, c->x_field.sig) != 0) { wrdma::signal_cancel(c->x_field.sig, 1); return fail(WR_ERR_HIP, "DMA copy could not be queued"); }
        return wrdma::wait(c->x_field.sig) == 0 ? WR_OK : fail(WR_ERR_HIP, "DMA copy failed");
    }
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return WR_OK;
}

int wr_dev_download(wr_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (int rc = ctx_bind(c)) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (bytes >= (1u << 20) && c->x_field.sig && wrdma::can_copy(dst, src)) {
        std::lock_guard<std::mutex> lk(c->mu);
        wrdma::signal_arm(c->x_field.sig, 1);
        if (wrdma::copy_async(dst, src, bytes, c->x_field.sig) != 0) { wrdma::signal_cancel(c->x_field.sig, 1); return fail(WR_ERR_HIP, "DMA copy could not be queued"); }
        return wrdma::wait(c->x_field.sig) == 0 ? WR_OK : fail(WR_ERR_HIP, "DMA copy failed");
    }
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return WR_OK;
}

int wr_dev_copy_kernel(wr_ctx* c, void* dst, const void* src, size_t bytes, int workgroups)
{
    if (int rc = ctx_bind(c)) return rc;
    if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 15) || workgroups < 1) return fail(WR_ERR_ARG, "copy kernel: 16-byte granularity");
    wrk::copy_kernel(dst, src, bytes, workgroups, c->stream);
    HIPCHK(hipGetLastError());
    return WR_OK;
}

int wr_dev_burn(wr_ctx* c, double ms, int mode, int workgroups)
{
    if (int rc = ctx_bind(c)) return rc;
    if (ms < 0 || ms > 1000 || workgroups < 1) return fail(WR_ERR_ARG, "burn: 0..1000 ms, at least one workgroup");
    wrk::burn(ms, mode, workgroups, c->d_partial, c->stream);
    HIPCHK(hipGetLastError());
    return WR_OK;
}

int wr_dev_copy(wr_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (int rc = ctx_bind(c)) return rc;
    StageLock cu(c->pool->cu_mu);  // a kernel stage like any other: keeps it off other contexts' transforms
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return WR_OK;
}

int wr_dev_linf(wr_ctx* c, const double* d_a, const double* d_b, size_t n, double* max_abs_diff, double* max_abs_a)
{
    if (int rc = ctx_bind(c)) return rc;
    if (!n) return fail(WR_ERR_ARG, "empty array");
    wrk::linf_diff(d_a, d_b, n, c->d_partial, c->h_result_dev, c->stream);
    HIPCHK(hipStreamSynchronize(c->stream));
    *max_abs_diff = c->h_result[0];
    *max_abs_a = c->h_result[1];
    return WR_OK;
}

int wr_dev_transform(wr_ctx* c, double* d_fld, int nx, int ny, int nz, int lvl)
{
    if (int rc = ctx_bind(c)) return rc;
    if (int rc = check_dims(nx, ny, nz, d_fld)) return rc;
    SlotNeed need;
    transform_need(nx, ny, nz, lvl, &need);
    SlotLease slot;
    if (int rc = slot.acquire(c, need)) return rc;
    StageLock cu(c->pool->cu_mu);
    double* res = nullptr;
    if (int rc = run_transform(c, slot.get(), d_fld, nx, ny, nz, lvl, &res)) return rc;
    if (res != d_fld)
        HIPCHK(hipMemcpyAsync(d_fld, res, (size_t)nx * ny * nz * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));  // the work space goes back with the lease
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
    wrk::quantize_plane(d_x, n, aopt, bopt, deps, minval, wrk::plane_ref(d_q), true, c->d_partial, c->h_result_dev, c->stream);
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
    for (int l = 0; l < nlay; l++) { p.q[l] = wrk::plane_ref(d_planes[l]); p.deps[l] = deps[l]; p.minval[l] = minval[l]; }
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

// Test hook: a host coder that outlives its call.  Plane 0 of the context is prepared as an encoder's plane of n symbols
// (call A), released (A is over) and prepared again (call B); then A's window handle asks for its first window, as a pool
// worker that was still holding A's stream would.  The request must be refused: B's chunk table, ring and window order stay
// as plane_prepare left them and WR_STAT_HANDOVER_ERRORS goes up by one.
int wr_test_stale_window(wr_ctx* c, size_t n)
{
    if (int rc = ctx_bind(c)) return rc;
    if (!n) return fail(WR_ERR_ARG, "empty plane");
    std::lock_guard<std::mutex> lk(c->mu);
    if (int rc = plane_prepare(c, 0, n, false)) return rc;
    PlaneStream& s = c->ps[0];
    const wrrc::PlaneWindow late = s.io;
    plane_release(c, 0);
    if (int rc = plane_prepare(c, 0, n, false)) return rc;
    const wrk::PlaneRef before = s.ref;
    const unsigned long refused0 = g_stat[WR_STAT_HANDOVER_ERRORS].load();
    size_t count = n;
    uint8_t* const w = late.window(late.user, 0, &count);
    (void)xfer_wait(&s.x[0]); (void)xfer_wait(&s.x[1]);
    const bool intact = memcmp(&before, &s.ref, sizeof before) == 0 && s.ho.next_first == 0 && !s.ho.ended && !s.ahead && !s.err &&
                        w != s.buf[0] && w != s.buf[1];
    const unsigned long refused = g_stat[WR_STAT_HANDOVER_ERRORS].load() - refused0;
    s.ahead = false;
    plane_release(c, 0);
    if (!intact) return fail(WR_ERR_HIP, "a window request with the handle of an earlier call reached the plane of the present one");
    if (refused != 1) return fail(WR_ERR_HIP, "the stale window request was not counted as refused");
    return WR_OK;
}

}  // extern "C"
