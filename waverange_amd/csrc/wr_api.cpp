// wr_api.cpp -- host side of libwaverange_amd: context, encode/decode pipeline, C ABI.
//
// Mirrors the reference's codec layer (src/core/wrappers.cpp) for the path
//   encoding_wrap : min/max -> forward transform -> bit-plane quantizer loop -> range coder
//   decoding_wrap : range decoder -> dequantise-accumulate -> inverse transform
// A call moves through three device stages on one of a few work-space SLOTS per GPU:
//   up      host -> device   field (encode)                                     SDMA engine (wr_dma.h)
//   kernels min/max, transform, quantizer / dequantizer                         the context's stream, one call
//                                                                               at a time (DevPool::cu_mu)
//   down    device -> host   block histograms, residual (encode) or field       SDMA engine
// The quantized planes are not part of the slot: they live in device buffers of their own (DevPlanes) and the host
// range coder (one thread per plane, fewer with the planes of a field interleaved in one loop: wr_set_threads, or the
// process-wide pool: wr_set_coder_pool) reads or writes them through a ring of two pinned 15 MB windows per plane
// (PlaneStream, wrrc::PlaneWindow) while the slot already serves the next field.  Copies and kernels are ordered from
// the host (HIP event of the producing kernel -> start the copy; signal of the copy -> launch the consumer); pageable
// caller memory goes through hipMemcpyAsync on a copy stream instead.  Compiled with hipcc, strict IEEE
// (-ffp-contract=off): the scalar arithmetic on deps/aopt/bopt/tolabs below must round exactly as
// wrappers.cpp:292-340 does.
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sched.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/waverange_amd.h"
#include "wr_dma.h"
#include "wr_kernels.h"
#include "wr_rangecoder.h"

#pragma clang fp contract(off)

namespace {

// reference src/core/defs.h:34-50
constexpr int kWavLvl = 4;
constexpr double kWavAccCoef = 1.75;
constexpr unsigned long kSafetyBufferFactor = 1;

thread_local std::string g_err;
std::atomic<int> g_verbose{-1};      // -1: not initialised from the environment yet
std::atomic<int> g_threads{-1};      // -1: not initialised from the environment yet (WR_THREADS, default one per plane)
std::atomic<int> g_enc_threads{0};   // 0: same as g_threads (wr_set_encoder_threads)
std::atomic<unsigned long> g_stat[4];  // see wr_stat()
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

// One set of device work space.  A device phase owns a slot from its first copy to its last; the
// kernels of different slots serialise on DevPool::cu_mu, their copies run under one another.
struct Slot {
    bool busy = false;
    bool disabled = false;  // the device had no memory left for it: never handed out again
    double* field = nullptr; size_t field_elems = 0;      // staging of a host caller's field
    double* scratch = nullptr; size_t scratch_elems = 0;  // coefficient array (out-of-place fused transform)
    double* lowbuf = nullptr; size_t lowbuf_elems = 0;    // compact low-pass boxes (fused transform)
    uint16_t* hist = nullptr; size_t hist_elems = 0;      // per-block byte histograms, all planes
    bool allocated() const { return field || scratch || lowbuf || hist; }
    void release_buffers()
    {
        (void)hipFree(field); (void)hipFree(scratch); (void)hipFree(lowbuf); (void)hipFree(hist);
        field = scratch = lowbuf = nullptr; hist = nullptr;
        field_elems = scratch_elems = lowbuf_elems = hist_elems = 0;
    }
};

constexpr int kMaxSlots = 4;

}  // namespace

// Per-GPU state shared by all contexts on it: the work-space slots (a 1024^3 slot is 8.6 GB each for
// field staging and coefficients, 1.2 GB of low-pass boxes; slots are populated on demand, so a
// lone caller uses one), the pool of plane buffers, one copy stream per direction (copies of all contexts queue on them in call
// order and stay off the streams that run kernels) and the stage locks.
// Quantized planes live in DEVICE memory, in buffers shared by the contexts of one device: a call borrows one per
// plane for as long as the plane exists (encode: quantized until coded; decode: from the first decoded symbol until the
// inverse transform has read it, parked in the context between wr_decode_begin and wr_decode_finish_*) and hands it
// back.  The host coder sees a plane only through a ring of two pinned chunks (PlaneStream below), so 16 fields in
// flight cost ~100 GiB of the 288 GB of HBM instead of that much pinned host memory.
struct DevPlanes {
    struct Buf { uint8_t* p = nullptr; size_t bytes = 0; };
    std::mutex mu;
    std::vector<Buf> idle;
    void drop_idle()
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const Buf& b : idle) { g_stat[WR_STAT_DEVICE_PLANE_BYTES] -= b.bytes; (void)hipFree(b.p); }
        idle.clear();
    }
    // smallest idle buffer that holds `bytes` without being more than twice as large, else a new one
    Buf take(size_t bytes)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            int best = -1;
            for (int i = 0; i < (int)idle.size(); i++)
                if (idle[i].bytes >= bytes && idle[i].bytes / 2 <= bytes && (best < 0 || idle[i].bytes < idle[best].bytes)) best = i;
            if (best >= 0) { Buf b = idle[best]; idle[best] = idle.back(); idle.pop_back(); return b; }
        }
        Buf b;
        void* q = nullptr;
        if (hipMalloc(&q, bytes) != hipSuccess) {
            (void)hipGetLastError();
            drop_idle();  // buffers of another field size may be holding the memory
            if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); q = nullptr; }
        }
        if (q) { b.p = static_cast<uint8_t*>(q); b.bytes = bytes; g_stat[WR_STAT_DEVICE_PLANE_BYTES] += bytes; }
        return b;
    }
    void give(const Buf& b)
    {
        std::lock_guard<std::mutex> lk(mu);
        idle.push_back(b);
    }
};

struct DevPool {
    std::mutex mu; std::condition_variable cv;  // slot hand-out
    Slot slots[kMaxSlots];
    int max_slots = 3;
    int users = 0;
    hipStream_t up = nullptr, down = nullptr;  // pageable fallback copies only
    std::mutex up_mu, cu_mu, down_mu;
    DevPlanes planes;
};

struct wr_ctx {
    int device = 0;
    DevPool* pool = nullptr;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool keep_residual = false;
    double* d_cutoff = nullptr; size_t cutoff_elems = 0;  // local cutoff vector (mx*my*mz > 1 only)
    double* d_partial = nullptr;
    double* d_mm = nullptr; size_t mm_records = 0;  // min/max records of the fused forward transform
    unsigned long long* d_idx = nullptr;
    // pinned host
    double* h_result = nullptr;  // [0..1] min/max, [2] probe value, [3] index, [4..7] fused min/max
    double* h_result_dev = nullptr;  // the same block as the device sees it: reductions write their result straight to the host
    uint16_t* h_hist = nullptr; size_t h_hist_elems = 0;  // pinned: per-block byte histograms, all planes
    // host coded-stream staging, one per plane
    uint8_t* enc_buf[WR_NLAYMAX] = {nullptr}; size_t enc_buf_bytes[WR_NLAYMAX] = {0};  // malloc'd: only coded bytes get touched
    hipEvent_t ev_plane[WR_NLAYMAX] = {nullptr};
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_d = nullptr, ev_mm = nullptr;
    // A host <-> device transfer in flight: on a DMA engine through ROCr (signal) and / or, for pageable host
    // memory, staged by the HIP runtime on the device's copy stream of that direction (event).
    struct Xfer {
        wrdma::Signal sig = 0;
        hipEvent_t ev = nullptr;
        bool dma_pending = false, hip_pending = false;
        double t_start = 0, ms = 0;  // duration: engine timestamps of the last DMA copy, else host clock
        std::mutex mu;               // several threads may wait for the same transfer
    };
    Xfer x_field, x_plane[WR_NLAYMAX];  // x_plane: the block histograms of an encode's plane
    // A quantized plane of this context: in device memory (borrowed from DevPool::planes), and the ring of two pinned
    // chunks through which the host coder reads (encode) or writes (decode) it, window by window (wrrc::PlaneWindow)
    struct PlaneStream {
        wr_ctx* c = nullptr;
        uint8_t* dev = nullptr; size_t dev_bytes = 0;
        size_t n = 0;
        uint8_t* buf[2] = {nullptr, nullptr};  // pinned (pageable if the host has no pinned memory left), allocated at first use, kept
        bool buf_pinned[2] = {false, false};
        Xfer x[2];
        int cur = 1;                            // buffer of the window handed out last
        size_t win_first = 0, win_count = 0;    // decode: the window being filled
        bool ahead = false; size_t ahead_first = 0;  // encode: the chunk in flight into buf[cur ^ 1]
        int err = 0;
        double copy_ms = 0;
        wrrc::PlaneWindow io;
    };
    PlaneStream ps[WR_NLAYMAX];
    // two-phase decode (wr_decode_begin / wr_decode_finish_*): the planes are decoded and wait in ps[].dev
    bool pend_valid = false;
    wr_enc_info pend_info;
    int pend_nx = 0, pend_ny = 0, pend_nz = 0;
    wr_timings pend_tm;
    std::mutex mu;
};

namespace {

int ctx_bind(wr_ctx* c) { HIPCHK(hipSetDevice(c->device)); return WR_OK; }

constexpr int kMaxDevices = 64;
DevPool g_pools[kMaxDevices];
std::mutex g_pools_mu;

// what a phase needs from its slot (0 = not needed)
struct SlotNeed {
    size_t field_elems = 0, scratch_elems = 0, lowbuf_elems = 0, hist_elems = 0;
};

template <class T>
hipError_t grow(T** buf, size_t* have, size_t want)
{
    if (*have >= want) return hipSuccess;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr; *have = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(buf), want * sizeof(T));
    if (e == hipSuccess) *have = want;
    return e;
}

hipError_t slot_ensure(Slot* s, const SlotNeed& need)
{
    hipError_t e;
    if ((e = grow(&s->field, &s->field_elems, need.field_elems)) != hipSuccess) return e;
    if ((e = grow(&s->scratch, &s->scratch_elems, need.scratch_elems)) != hipSuccess) return e;
    if ((e = grow(&s->lowbuf, &s->lowbuf_elems, need.lowbuf_elems)) != hipSuccess) return e;
    if ((e = grow(&s->hist, &s->hist_elems, need.hist_elems)) != hipSuccess) return e;
    return hipSuccess;
}

// RAII ownership of a slot.  Free slots that already hold buffers are handed out first; a further
// slot is populated only when all of those are busy.  If the device has no memory left for another
// slot, the pool shrinks to the slots it has and the caller waits for one of them.
class SlotLease {
public:
    SlotLease() = default;
    SlotLease(const SlotLease&) = delete;
    SlotLease& operator=(const SlotLease&) = delete;
    ~SlotLease() { release(); }

    // nowait: return 1 instead of blocking; spare: only succeed if that many further slots stay free
    int acquire(wr_ctx* c, const SlotNeed& need, bool nowait = false, int spare = 0)
    {
        DevPool* p = c->pool;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(p->mu);
                for (;;) {
                    Slot* pick = nullptr;
                    int nfree = 0;
                    for (int i = 0; i < p->max_slots; i++) {
                        Slot& s = p->slots[i];
                        if (s.busy || s.disabled) continue;
                        nfree++;
                        // populated slots first; a further one is populated only when all of those are busy
                        if (!pick || (s.allocated() && !pick->allocated())) pick = &s;
                    }
                    if (pick && nfree > spare) { pick->busy = true; slot_ = pick; pool_ = p; break; }
                    if (nowait) return 1;
                    p->cv.wait(lk);
                }
            }
            const bool fresh = !slot_->allocated();
            const hipError_t e = slot_ensure(slot_, need);
            if (e == hipSuccess) { if (fresh) g_stat[WR_STAT_SLOTS_POPULATED]++; return WR_OK; }
            (void)hipGetLastError();
            std::unique_lock<std::mutex> lk(p->mu);
            int others = 0;
            for (int i = 0; i < p->max_slots; i++)
                if (&p->slots[i] != slot_ && !p->slots[i].disabled && p->slots[i].allocated()) others++;
            slot_->release_buffers();
            // out of device memory with other slots populated: keep to those and wait for one of them
            const bool retry = e == hipErrorOutOfMemory && others > 0;
            if (retry) slot_->disabled = true;
            slot_->busy = false; slot_ = nullptr;
            p->cv.notify_all();
            if (!retry) return fail(WR_ERR_HIP, std::string("device work space: ") + hipGetErrorString(e));
            if (nowait) return 1;
        }
    }
    void release()
    {
        if (!slot_) return;
        { std::lock_guard<std::mutex> lk(pool_->mu); slot_->busy = false; }
        pool_->cv.notify_all();
        slot_ = nullptr;
    }
    Slot* get() const { return slot_; }
    Slot* operator->() const { return slot_; }
    explicit operator bool() const { return slot_ != nullptr; }

private:
    Slot* slot_ = nullptr;
    DevPool* pool_ = nullptr;
};

using StageLock = std::unique_lock<std::mutex>;

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

enum Dir { kUp = 0, kDown = 1 };
struct Piece { void* dst; const void* src; size_t bytes; };

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

// ---- device-resident planes and their host windows -------------------------------------------------------------------
// a window: 256 coder blocks, 15.36 MB, ~0.3 ms on a DMA engine (WR_WINDOW_BLOCKS: 1..256 blocks, for tests that want
// many windows on small fields)
const size_t kChunkSyms = []() {
    int blocks = 256;
    if (const char* e = getenv("WR_WINDOW_BLOCKS")) { const int v = atoi(e); if (v >= 1 && v <= 256) blocks = v; }
    return (size_t)blocks * wrrc::kBlock;
}();
const size_t kChunkBytes = 256 * (size_t)wrrc::kBlock + 64;  // ring buffers are always full-size

using PlaneStream = wr_ctx::PlaneStream;

void plane_release(wr_ctx* c, int l)
{
    PlaneStream& s = c->ps[l];
    if (!s.dev) return;
    DevPlanes::Buf b; b.p = s.dev; b.bytes = s.dev_bytes;
    c->pool->planes.give(b);
    s.dev = nullptr; s.dev_bytes = 0;
}

// Encoder side: the symbols [first, first + count) of the plane, fetched into the ring; the following chunk is
// started into the buffer the coder has just left, so that it arrives while this one is being coded.
uint8_t* plane_window_encode(void* user, size_t first, size_t* count)
{
    PlaneStream& s = *static_cast<PlaneStream*>(user);
    wr_ctx* const c = s.c;
    (void)hipSetDevice(c->device);  // coder threads: the pageable-copy fallback of xfer_start needs the device bound
    const size_t want = *count < kChunkSyms ? *count : kChunkSyms;
    const int b = s.cur ^ 1;
    if (!(s.ahead && s.ahead_first == first)) {
        if (s.ahead) (void)xfer_wait(&s.x[b]);
        const Piece pc = {s.buf[b], s.dev + first, want};
        if (xfer_start(c, &s.x[b], &pc, 1, kDown) != WR_OK) s.err = 1;
    }
    if (xfer_wait(&s.x[b]) != WR_OK) s.err = 1;
    s.copy_ms += s.x[b].ms;
    s.cur = b;
    s.ahead = false;
    const size_t next = first + want;
    if (next < s.n) {
        const Piece pc = {s.buf[b ^ 1], s.dev + next, s.n - next < kChunkSyms ? s.n - next : kChunkSyms};
        if (xfer_start(c, &s.x[b ^ 1], &pc, 1, kDown) == WR_OK) { s.ahead = true; s.ahead_first = next; }
        else s.err = 1;
    }
    *count = want;
    return s.buf[b];
}

// Decoder side: room for the symbols from `first` on; the window handed out before goes to the device meanwhile.
// *count == 0 ends the stream: both uploads are waited for.
uint8_t* plane_window_decode(void* user, size_t first, size_t* count)
{
    PlaneStream& s = *static_cast<PlaneStream*>(user);
    wr_ctx* const c = s.c;
    (void)hipSetDevice(c->device);
    if (s.win_count) {
        const Piece pc = {s.dev + s.win_first, s.buf[s.cur], s.win_count};
        if (xfer_start(c, &s.x[s.cur], &pc, 1, kUp) != WR_OK) s.err = 1;
        s.win_count = 0;
    }
    if (*count == 0) {
        for (int b = 0; b < 2; b++) { if (xfer_wait(&s.x[b]) != WR_OK) s.err = 1; s.copy_ms += s.x[b].ms; s.x[b].ms = 0; }
        return nullptr;
    }
    const int b = s.cur ^ 1;
    if (xfer_wait(&s.x[b]) != WR_OK) s.err = 1;  // the upload of two windows ago
    s.copy_ms += s.x[b].ms; s.x[b].ms = 0;
    s.cur = b;
    s.win_first = first;
    s.win_count = *count < kChunkSyms ? *count : kChunkSyms;
    *count = s.win_count;
    return s.buf[b];
}

// plane l of n symbols for this call: a device buffer (kept if the context holds one that fits: a finish after a
// begin), the ring, and the window callbacks of the direction
int plane_prepare(wr_ctx* c, int l, size_t n, bool decode)
{
    PlaneStream& s = c->ps[l];
    const size_t bytes = wr_plane_pitch(n);
    if (!s.dev || s.dev_bytes < bytes) {
        plane_release(c, l);
        const DevPlanes::Buf b = c->pool->planes.take(bytes);
        if (!b.p) return fail(WR_ERR_HIP, "out of device memory for a quantized plane (fewer calls in flight need less)");
        s.dev = b.p; s.dev_bytes = b.bytes;
    }
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
    s.io.window = decode ? plane_window_decode : plane_window_encode;
    s.io.user = &s;
    return WR_OK;
}

// encode: the plane is complete on the device -- its first chunk sets off for the host before a coder asks for it
void plane_prefetch(wr_ctx* c, int l)
{
    PlaneStream& s = c->ps[l];
    if (!s.n) return;
    const Piece pc = {s.buf[s.cur ^ 1], s.dev, s.n < kChunkSyms ? s.n : kChunkSyms};
    if (xfer_start(c, &s.x[s.cur ^ 1], &pc, 1, kDown) == WR_OK) { s.ahead = true; s.ahead_first = 0; }
    else s.err = 1;
}

// Declared before anything that may still touch the planes when the call unwinds (coder threads are joined, pool
// jobs waited for by then): hands the context's device planes back unless the call parks them (wr_decode_begin keeps
// the decoded planes for wr_decode_finish_*).
struct PlaneHold {
    wr_ctx* c;
    bool keep = false;
    explicit PlaneHold(wr_ctx* ctx) : c(ctx) {}
    ~PlaneHold()
    {
        if (keep) return;
        for (int l = 0; l < WR_NLAYMAX; l++) {
            PlaneStream& s = c->ps[l];
            if (s.dev) { (void)xfer_wait(&s.x[0]); (void)xfer_wait(&s.x[1]); }  // a prefetch nobody consumed (error paths)
            s.ahead = false;
            plane_release(c, l);
        }
    }
};

// a whole plane on the host, for the diagnostics of the verbose mode (wrappers.cpp:401-409, 503-510)
std::string plane_log(wr_ctx* c, int l, size_t n, const wr_enc_info* info, bool encode, size_t len)
{
    std::vector<uint8_t> q(n);
    if (hipMemcpy(q.data(), c->ps[l].dev, n, hipMemcpyDeviceToHost) != hipSuccess) return std::string();
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

int check_dims(int nx, int ny, int nz, const void* dev_ptr)
{
    if (nx < 1 || ny < 1 || nz < 1) return fail(WR_ERR_ARG, "non-positive dimension");
    if (((uintptr_t)dev_ptr) & 15) return fail(WR_ERR_ARG, "device field pointer must be 16-byte aligned");
    return WR_OK;
}

struct Sem {  // tiny counting semaphore limiting concurrent range-coder threads
    std::mutex m; std::condition_variable cv; int n;
    explicit Sem(int k) : n(k) {}
    void acquire() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return n > 0; }); n--; }
    void release() { { std::lock_guard<std::mutex> l(m); n++; } cv.notify_one(); }
};

// joins what it holds on every way out of a scope (a std::thread that is destroyed while joinable
// terminates the process)
struct Workers {
    std::vector<std::thread> v;
    ~Workers() { join(); }
    void join() { for (auto& t : v) if (t.joinable()) t.join(); v.clear(); }
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

void wr_set_verbosity(int level) { g_verbose.store(level ? 1 : 0); }
void wr_set_threads(int nthreads) { g_threads.store(nthreads < 1 ? 1 : nthreads); g_enc_threads.store(0); }
void wr_set_encoder_threads(int nthreads) { g_enc_threads.store(nthreads < 0 ? 0 : nthreads); }
void wr_set_writeback_residual(int on) { g_writeback.store(on ? 1 : 0); }
void wr_pool_loop_stats(double* seconds4, double* blocks4) { wrrc::pool_loop_stats(seconds4, blocks4); }

unsigned long wr_stat(int what)
{
    if (what == WR_STAT_POOL_IDLE_MS) return (unsigned long)(wrrc::pool_idle_seconds() * 1e3);
    return (what >= 0 && what < 4) ? g_stat[what].load() : 0;
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
    wrk::quantize_plane(d_x, n, aopt, bopt, deps, minval, d_q, true, c->d_partial, c->h_result_dev, c->stream);
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

// local cutoff description (mx*my*mz == 1: uniform cutoff, the benchmark path)
struct Cutoff {
    int mx = 1, my = 1, mz = 1;
    const double* vec = nullptr;  // host, mx*my*mz entries
    int count() const { return mx * my * mz; }
};

// Kernel stage of the encoder (call with the slot leased and DevPool::cu_mu held).  Two hooks for the full
// pipeline: after_quant(l) is called right after plane l's quantizer kernel and the read-back of the next
// plane's min/max have been enqueued (the block histograms are enqueued there, behind the read-back);
// plane_ready(l, last) is called from the host once everything enqueued for plane l has completed on the
// device (the download starts there).  *resid = where the coefficient array / residual lives afterwards.
template <class PlaneBuf, class AfterQuant, class PlaneReady>
int encode_planes_core(wr_ctx* c, Slot* slot, double* d_fld, int nx, int ny, int nz, int wtflag, const Cutoff& cut,
                       PlaneBuf plane_buf, wr_enc_info* info, wr_timings* tm, AfterQuant after_quant, PlaneReady plane_ready,
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
    const size_t mm_records = (wtflag && use_fused(nx, ny, nz, kWavLvl) && !getenv("WR_NO_FUSED_MINMAX")) ? wrk::fused_minmax_records(nx, ny, nz) : 0;
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
    for (;;) {
        PlaneStep s = plane_step(lo, hi, info->tolabs, ilay);
        info->minval_vec[ilay] = s.minval;
        info->deps_vec[ilay] = s.deps;
        if (verbose()) { printf("min=%g max=%g\n", lo, hi); printf("ilay=%u deps=%g\n", ilay, s.deps); }
        const bool resid_upd = !s.last || c->keep_residual;
        uint8_t* const d_plane = plane_buf(ilay);  // device memory of this plane (the error is set if there is none)
        if (!d_plane) return WR_ERR_HIP;
        HIPCHK(hipEventRecord(c->ev_a, c->stream));
        if (local) {
            wrk::LocalCutoff lc;
            lc.nx = nx; lc.ny = ny; lc.nz = nz; lc.wlev = info->wlev;
            lc.mx = cut.mx; lc.my = cut.my; lc.mz = cut.mz;
            lc.cutoff = c->d_cutoff;
            lc.tol_scale = info->tolabs / tolrel;
            lc.tolabs = info->tolabs;
            lc.span = hi - lo;
            wrk::quantize_plane_local(d_fld, n, s.aopt, s.bopt, s.deps, s.minval, d_plane, lc,
                                      c->d_partial, c->h_result_dev, c->stream);
        } else
        wrk::quantize_plane(d_fld, n, s.aopt, s.bopt, s.deps, s.minval, d_plane, resid_upd,
                            c->d_partial, c->h_result_dev, c->stream);
        HIPCHK(hipEventRecord(c->ev_b, c->stream));
        HIPCHK(hipGetLastError());
        // the next plane's min/max is in host memory when this event fires (the reduction stores it there); what
        // after_quant enqueues runs behind it
        HIPCHK(hipEventRecord(c->ev_mm, c->stream));
        if (int rc = after_quant(ilay)) return rc;
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
        ilay++;
        lo = c->h_result[0]; hi = c->h_result[1];
        if (lo == 0.0)  // sign of a zero minimum: rare path, goes through the full read-back
            if (int rc = read_minmax(c, d_fld, n, true, &lo, &hi)) return rc;
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
    int rc = encode_planes_core(c, slot.get(), d_fld, nx, ny, nz, wtflag, cut, [&](unsigned l) { return d_planes + l * pitch; }, info, nullptr,
                                [](unsigned) { return WR_OK; }, [](unsigned, bool) { return WR_OK; }, &resid);
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
        p.q[l] = d_planes + l * wr_plane_pitch(n);
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
    auto code_group = [&](unsigned l0, unsigned l1) {
        (void)hipSetDevice(dev);
        for (unsigned l = l0; l < l1; l++)
            if (xfer_wait(&c->x_plane[l]) != WR_OK) copy_failed[l] = 1;
        sem.acquire();
        const double t = now();
        const uint8_t* syms[WR_NLAYMAX];
        uint8_t* outs[WR_NLAYMAX];
        const uint16_t* hs[WR_NLAYMAX];
        const wrrc::PlaneWindow* ios[WR_NLAYMAX];
        for (unsigned l = l0; l < l1; l++) { syms[l - l0] = nullptr; outs[l - l0] = c->enc_buf[l]; hs[l - l0] = c->h_hist + l * hist_per_plane; ios[l - l0] = &c->ps[l].io; }
        wrrc::encode_planes((int)(l1 - l0), syms, n, outs, hs, lens + l0, ios);
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
            const Piece pc = {d_fld, fld.host, n * sizeof(double)};
            if ((rc = xfer_start(c, &c->x_field, &pc, 1, kUp)) != WR_OK) return rc;
            if ((rc = xfer_wait(&c->x_field)) != WR_OK) return rc;
            local.h2d_ms = (float)c->x_field.ms;
        }
        double* resid = d_fld;
        auto after_quant = [&](unsigned l) -> int {
            // block histograms of plane l on the kernel stream, behind the read-back of the next plane's min/max
            wrk::block_histograms(c->ps[l].dev, n, slot->hist + l * hist_per_plane, c->stream);
            return WR_OK;
        };
        auto plane_buf = [&](unsigned l) -> uint8_t* { return plane_prepare(c, (int)l, n, false) == WR_OK ? c->ps[l].dev : nullptr; };
        auto plane_ready = [&](unsigned l, bool) -> int {
            // plane l and its histograms are complete on the device: the histograms go to pinned host memory, the
            // plane's first chunk sets off into its ring, and a coder thread waits for them
            if (int r = ensure_enc_buf(c, (int)l, wrrc::encode_bound(n))) return r;
            const Piece pc = {c->h_hist + l * hist_per_plane, slot->hist + l * hist_per_plane, hist_per_plane * sizeof(uint16_t)};
            if (int r = xfer_start(c, &c->x_plane[l], &pc, 1, kDown)) return r;
            plane_prefetch(c, (int)l);
            planes_started = l + 1;
            if (per_plane) workers.v.emplace_back(code_group, l, l + 1);
            return WR_OK;
        };
        {
            // ---- stage "kernels"
            StageLock cu(pool->cu_mu);
            rc = encode_planes_core(c, slot.get(), d_fld, nx, ny, nz, wtflag, cut, plane_buf, info, &local, after_quant, plane_ready, &resid);
            (void)hipStreamSynchronize(c->stream);
            if (rc == WR_OK && c->keep_residual && info->nlay && !fld.host && resid != fld.dev) {  // leave the residual where the reference leaves it
                if (hipMemcpyAsync(fld.dev, resid, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
                    hipStreamSynchronize(c->stream) != hipSuccess)
                    rc = fail(WR_ERR_HIP, "residual copy failed");
            }
        }
        // ---- stage "down": the histograms were sent off as the planes completed; the residual follows them
        if (rc == WR_OK && c->keep_residual && info->nlay && fld.host) {
            const Piece pc = {fld.host, resid, n * sizeof(double)};
            if ((rc = xfer_start(c, &c->x_field, &pc, 1, kDown)) == WR_OK) rc = xfer_wait(&c->x_field);
        }
        // The slot's histogram buffer must not be reused before its downloads are done (the coder threads wait
        // for the same transfers; xfer_wait is safe to call from both sides).  With the coder pool, every plane
        // is handed over the moment its histograms are on the host.
        for (unsigned l = 0; l < planes_started; l++) {
            if (xfer_wait(&c->x_plane[l]) != WR_OK) copy_failed[l] = 1;
            if (pooled && rc == WR_OK && !copy_failed[l]) {
                wrrc::PlaneJob& j = jobs[l];
                j.kind = wrrc::PlaneJob::kEncode;
                j.src = nullptr; j.io = &c->ps[l].io; j.dst = c->enc_buf[l]; j.n = n; j.hist = c->h_hist + l * hist_per_plane;
                if (wrrc::pool_submit(&j, 1, &batch)) pool_mask |= 1u << l;
                else workers.v.emplace_back(code_group, l, l + 1);  // the pool was stopped meanwhile: a thread of this call codes the plane
            }
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
        local.d2h_ms += (float)(c->x_plane[l].ms + c->ps[l].copy_ms);
        if (verbose()) logs[l] = plane_log(c, (int)l, n, info, true, lens[l]);  // wrappers.cpp:401-409, 430
    }
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
    try {
        Workers copiers;
        for (unsigned l = 1; l < info->nlay; l++)
            copiers.v.emplace_back([&, l]() { memcpy(data_enc + offs[l], c->enc_buf[l], lens[l]); });
        if (info->nlay) memcpy(data_enc, c->enc_buf[0], lens[0]);
    } catch (const std::exception&) {  // no thread to be had: copy here
        for (unsigned l = 0; l < info->nlay; l++) memcpy(data_enc + offs[l], c->enc_buf[l], lens[l]);
    }
    // The per-plane coder output has been copied out: hand its pages back (only coded bytes were ever touched, but a
    // noise plane's gigabyte would otherwise stay resident in every context that once coded one)
    for (unsigned l = 0; l < info->nlay; l++) {
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

int decode_impl(wr_ctx* c, FieldRef fld, int nx, int ny, int nz, const wr_enc_info* info, const unsigned char* data_enc,
                size_t data_len, wr_timings* tm, DecodeMode mode = kDecodeWhole)
{
    if (int rc = ctx_bind(c)) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
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

    if (host_half) {
        c->pend_valid = false;  // whatever an earlier begin parked here is overwritten now
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
            if (wrrc::pool_submit(jobs, nlay, &batch)) {
                wrrc::pool_wait(&batch);
                for (int l = 0; l < nlay; l++) { got[l] = jobs[l].result; coder_s[l] = jobs[l].seconds; }
            } else
                pooled = false;  // the pool was stopped meanwhile: this call's own threads decode the planes
        }
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
        for (int l = 0; l < nlay; l++) { p.deps[l] = info->deps_vec[l]; p.minval[l] = info->minval_vec[l]; p.q[l] = c->ps[l].dev; }
        double* d_fld = fld.host ? slot->field : fld.dev;
        {
            // ---- stage "kernels"
            StageLock cu(pool->cu_mu);
            rc = inverse_from_planes(c, slot.get(), d_fld, nx, ny, nz, (int)info->wlev, p);
            if (rc == WR_OK && hipGetLastError() != hipSuccess) rc = fail(WR_ERR_HIP, "kernel launch failed");
            (void)hipStreamSynchronize(c->stream);
        }
        if (rc) return rc;
        if (fld.host) {
            // ---- stage "down": the reconstructed field, device -> host
            const Piece pc = {fld.host, d_fld, n * sizeof(double)};
            if ((rc = xfer_start(c, &c->x_field, &pc, 1, kDown)) != WR_OK) return rc;
            if ((rc = xfer_wait(&c->x_field)) != WR_OK) return rc;
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
        const Piece pc = {slot->field, h_fld, n * sizeof(double)};
        if (int rc = xfer_start(c, &c->x_field, &pc, 1, kUp)) return rc;
        if (int rc = xfer_wait(&c->x_field)) return rc;
    }
    double* res = nullptr;
    {
        StageLock cu(pool->cu_mu);
        if (int rc = run_transform(c, slot.get(), slot->field, nx, ny, nz, lvl, &res)) return rc;
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    const Piece pc = {h_fld, res, n * sizeof(double)};
    if (int rc = xfer_start(c, &c->x_field, &pc, 1, kDown)) return rc;
    return xfer_wait(&c->x_field);
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

int wr_range_decode_vec(int count, const unsigned char* const* in, const size_t* len, unsigned char* const* sym, const size_t* n,
                        size_t* produced)
{
    if (!wrrc::decode_planes_vec(count, in, len, sym, n, produced)) return fail(WR_ERR_UNSUPPORTED, "this CPU has no AVX-512");
    return WR_OK;
}

int wr_range_encode_vec(int count, const unsigned char* const* sym, const size_t* n, unsigned char* const* out, size_t* lens)
{
    if (!wrrc::encode_planes_vec(count, sym, n, out, lens)) return fail(WR_ERR_UNSUPPORTED, "this CPU has no AVX-512");
    return WR_OK;
}

int wr_range_encode_pool(int count, const unsigned char* const* sym, const size_t* n, unsigned char* const* out, size_t* lens)
{
    if (wrrc::pool_threads() < 1) return fail(WR_ERR_ARG, "the coder pool is not running (wr_set_coder_pool)");
    if (count < 1) return WR_OK;
    std::vector<wrrc::PlaneJob> jobs((size_t)count);
    wrrc::JobBatch batch;
    for (int k = 0; k < count; k++) { jobs[k].kind = wrrc::PlaneJob::kEncode; jobs[k].src = sym[k]; jobs[k].n = n[k]; jobs[k].dst = out[k]; }
    if (!wrrc::pool_submit(jobs.data(), count, &batch)) return fail(WR_ERR_ARG, "the coder pool is not running (wr_set_coder_pool)");
    wrrc::pool_wait(&batch);
    for (int k = 0; k < count; k++) lens[k] = jobs[k].result;
    return WR_OK;
}

int wr_range_decode_pool(int count, const unsigned char* const* in, const size_t* len, unsigned char* const* sym, const size_t* n,
                         size_t* produced)
{
    if (wrrc::pool_threads() < 1) return fail(WR_ERR_ARG, "the coder pool is not running (wr_set_coder_pool)");
    if (count < 1) return WR_OK;
    std::vector<wrrc::PlaneJob> jobs((size_t)count);
    wrrc::JobBatch batch;
    for (int k = 0; k < count; k++) {
        jobs[k].kind = wrrc::PlaneJob::kDecode; jobs[k].src = in[k]; jobs[k].src_len = len[k]; jobs[k].dst = sym[k]; jobs[k].n = n[k];
    }
    if (!wrrc::pool_submit(jobs.data(), count, &batch)) return fail(WR_ERR_ARG, "the coder pool is not running (wr_set_coder_pool)");
    wrrc::pool_wait(&batch);
    for (int k = 0; k < count; k++) produced[k] = jobs[k].result;
    return WR_OK;
}

namespace {
// memory-backed PlaneWindow for the windowed test hooks: the symbol side passes through two alternating buffers of
// `chunk` symbols, as a device-resident plane does through its pinned ring; the buffer not in use is poisoned
struct MemWindow {
    const uint8_t* plane = nullptr;  // encode: source plane
    uint8_t* out = nullptr;          // decode: destination plane
    size_t n = 0, chunk = 0;
    std::vector<uint8_t> buf[2];
    int cur = 1;
    size_t last_first = 0, last_count = 0;
    wrrc::PlaneWindow io;
    static uint8_t* enc_window(void* user, size_t first, size_t* count)
    {
        MemWindow* w = static_cast<MemWindow*>(user);
        const size_t c = *count < w->chunk ? *count : w->chunk;
        memset(w->buf[w->cur].data(), 0xA5, w->buf[w->cur].size());  // the window handed out before is dead now
        w->cur ^= 1;
        memcpy(w->buf[w->cur].data(), w->plane + first, c);
        *count = c;
        return w->buf[w->cur].data();
    }
    static uint8_t* dec_window(void* user, size_t first, size_t* count)
    {
        MemWindow* w = static_cast<MemWindow*>(user);
        if (w->last_count) memcpy(w->out + w->last_first, w->buf[w->cur].data(), w->last_count);  // the previous window is complete
        w->last_count = 0;
        if (*count == 0) return nullptr;
        const size_t c = *count < w->chunk ? *count : w->chunk;
        w->cur ^= 1;
        memset(w->buf[w->cur].data(), 0x5A, w->buf[w->cur].size());
        w->last_first = first; w->last_count = c;
        *count = c;
        return w->buf[w->cur].data();
    }
    void init(size_t n_, size_t chunk_, bool decode)
    {
        n = n_; chunk = chunk_;
        buf[0].assign(chunk, 0); buf[1].assign(chunk, 0);
        io.window = decode ? dec_window : enc_window;
        io.user = this;
    }
};
}  // namespace

int wr_range_encode_windowed(int mode, int count, const unsigned char* const* sym, size_t n, size_t chunk, unsigned char* const* out, size_t* lens)
{
    if (count < 1) return WR_OK;
    if (chunk == 0 || chunk % wrrc::kBlock) return fail(WR_ERR_ARG, "the window length must be a multiple of 60000");
    std::vector<MemWindow> w((size_t)count);
    std::vector<const wrrc::PlaneWindow*> io((size_t)count);
    std::vector<size_t> ns((size_t)count, n);
    std::vector<const unsigned char*> none((size_t)count, nullptr);
    for (int k = 0; k < count; k++) { w[k].plane = sym[k]; w[k].init(n, chunk, false); io[k] = &w[k].io; }
    if (mode == 0) wrrc::encode_planes(count, none.data(), n, out, nullptr, lens, io.data());
    else if (mode == 2) { if (!wrrc::encode_planes_vec(count, none.data(), ns.data(), out, lens, io.data())) return fail(WR_ERR_UNSUPPORTED, "this CPU has no AVX-512"); }
    else {
        if (wrrc::pool_threads() < 1) return fail(WR_ERR_ARG, "the coder pool is not running (wr_set_coder_pool)");
        std::vector<wrrc::PlaneJob> jobs((size_t)count);
        wrrc::JobBatch batch;
        for (int k = 0; k < count; k++) { jobs[k].kind = wrrc::PlaneJob::kEncode; jobs[k].n = n; jobs[k].dst = out[k]; jobs[k].io = io[k]; }
        if (!wrrc::pool_submit(jobs.data(), count, &batch)) return fail(WR_ERR_ARG, "the coder pool is not running (wr_set_coder_pool)");
        wrrc::pool_wait(&batch);
        for (int k = 0; k < count; k++) lens[k] = jobs[k].result;
    }
    return WR_OK;
}

int wr_range_decode_windowed(int mode, int count, const unsigned char* const* in, const size_t* len, unsigned char* const* sym, size_t n,
                             size_t chunk, size_t* produced)
{
    if (count < 1) return WR_OK;
    if (chunk == 0 || chunk % wrrc::kBlock) return fail(WR_ERR_ARG, "the window length must be a multiple of 60000");
    std::vector<MemWindow> w((size_t)count);
    std::vector<const wrrc::PlaneWindow*> io((size_t)count);
    std::vector<size_t> ns((size_t)count, n);
    std::vector<unsigned char*> none((size_t)count, nullptr);
    for (int k = 0; k < count; k++) { w[k].out = sym[k]; w[k].init(n, chunk, true); io[k] = &w[k].io; }
    if (mode == 0) wrrc::decode_planes(count, in, len, none.data(), n, produced, io.data());
    else if (mode == 2) { if (!wrrc::decode_planes_vec(count, in, len, none.data(), ns.data(), produced, io.data())) return fail(WR_ERR_UNSUPPORTED, "this CPU has no AVX-512"); }
    else {
        if (wrrc::pool_threads() < 1) return fail(WR_ERR_ARG, "the coder pool is not running (wr_set_coder_pool)");
        std::vector<wrrc::PlaneJob> jobs((size_t)count);
        wrrc::JobBatch batch;
        for (int k = 0; k < count; k++) {
            jobs[k].kind = wrrc::PlaneJob::kDecode; jobs[k].src = in[k]; jobs[k].src_len = len[k]; jobs[k].n = n; jobs[k].io = io[k];
        }
        if (!wrrc::pool_submit(jobs.data(), count, &batch)) return fail(WR_ERR_ARG, "the coder pool is not running (wr_set_coder_pool)");
        wrrc::pool_wait(&batch);
        for (int k = 0; k < count; k++) produced[k] = jobs[k].result;
    }
    return WR_OK;
}

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

// =====================================================================================
// Part 1: libwaverange drop-in symbols (host pointers)
// =====================================================================================
namespace {

// The reference's entry points are re-entrant on distinct buffers (wrappers.cpp works on locals only).
// Here every call borrows a context from a free list (created on demand, kept for reuse), so concurrent
// callers never share staging; their device stages serialise on the per-GPU stage locks.
std::mutex g_free_mu;
std::vector<wr_ctx*> g_free_ctx;

[[noreturn]] void fatal(const char* where)
{
    fprintf(stderr, "libwaverange_amd: %s: %s\n", where, g_err.c_str());
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
    if (mx < 1 || my < 1 || mz < 1) { g_err = "mx, my, mz must be >= 1"; fatal("encoding_wrap"); }
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
    if (info.nlay > WR_NLAYMAX) { g_err = "nlay > 8"; fatal("decoding_wrap"); }
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
