// wr_internal.h -- what the translation units of libwaverange_amd's host side share (not installed, no C ABI here):
//   wr_pipeline.cpp  contexts, work-space slots, transfers, device-resident planes and their host windows, stage-level
//                    entry points (wr_dev_*), settings
//   wr_codec.cpp     the codec drivers: encode / decode through the stages (wr_encode_*, wr_decode_*)
//   wr_coder_hooks.cpp  the host range coder alone behind the C ABI (wr_range_*)
//   wr_dropin.cpp    the reference's entry points (encoding_wrap, decoding_wrap, ...) on top of the above
#pragma once
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
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/waverange_amd.h"
#include "wr_dma.h"
#include "wr_handover.h"
#include "wr_kernels.h"
#include "wr_rangecoder.h"

#pragma clang fp contract(off)

namespace wri {

// reference src/core/defs.h:34-50
constexpr int kWavLvl = 4;
constexpr double kWavAccCoef = 1.75;
constexpr unsigned long kSafetyBufferFactor = 1;
constexpr int kMaxDevices = 64;

extern std::atomic<unsigned long> g_stat[12];  // see wr_stat()
std::string& last_error();            // this thread's message (wr_last_error)
int coder_threads();                  // wr_set_threads / WR_THREADS, default one per plane
int encoder_threads();
int verbose();
int writeback_residual();
int fail(int code, const std::string& msg);
double now();

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            /* the runtime keeps the error as the thread's "last error" until somebody asks for it: left there, it would fail   \
               the NEXT call's check of a kernel launch (a refused 8.6 GB hipMalloc once failed a later encode that way) */   \
            (void)hipGetLastError();                                                         \
            return ::wri::fail(WR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
        }                                                                                    \
    } while (0)

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

}  // namespace wri

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
    size_t allocated = 0;         // bytes this device's planes hold right now, in use + idle + reserved for an allocation under way (mu)
    std::condition_variable cv;   // a buffer came back (calls waiting for device memory: plane_prepare)
    std::mutex gather_mu;         // held by the one decode that is gathering its planes (wr_codec.cpp)
    std::mutex gate_mu;           // decodes pass the pool's admission gate one at a time: from the look at the queues to the submit
    size_t chunk_bytes = 0;       // large planes live in chunks of this size or more (WR_PLANE_CHUNK_MB, default 32 MiB; 0: not read yet)
    size_t chunk_limit = 0;       // WR_PLANE_LIMIT_MB: device memory the planes of all calls may take together (0: what the device gives)
    // What the planes leave of the device's memory to everybody else (WR_PLANE_RESERVE_MB, default 2 GiB): the planes are
    // the one consumer that allocates until the device says no, and the HIP runtime allocates too, lazily and at moments of
    // its own (kernel-argument and signal pools, code objects of kernels used for the first time, staging for pageable
    // copies): a device whose last megabyte went to a plane chunk leaves those allocations to fail inside the runtime,
    // where nothing reports them (profiles/r04/NOTES.md, the round-3 fault).
    size_t reserve_bytes = (size_t)2 << 30;
    // Test hook (WR_TEST_PLANE_ALLOC_FAIL="first:count"): the device allocations number first .. first+count-1 of this
    // process (0-based, counted over all devices) fail as if the device were full, so that the path "hipMalloc fails ->
    // idle buffers are dropped -> the call waits without its kernel-stage lock" runs on a device with memory to spare.
    static bool alloc_fails_now();

    void drop_idle()
    {
        std::lock_guard<std::mutex> lk(mu);
        drop_idle_locked();
    }
    // smallest idle buffer that holds `bytes` without being more than twice as large, else a new one; {nullptr, 0} if the
    // device (or the cap) has no room right now.  The cap is checked and the bytes are booked in ONE critical section, so
    // callers that arrive together cannot all pass the check and overshoot it.
    Buf take(size_t bytes, size_t held_by_caller = 0);
    void give(const Buf& b)
    {
        { std::lock_guard<std::mutex> lk(mu); idle.push_back(b); }
        cv.notify_all();
    }

private:
    void drop_idle_locked()
    {
        for (const Buf& b : idle) { allocated -= b.bytes; wri::g_stat[WR_STAT_DEVICE_PLANE_BYTES] -= b.bytes; (void)hipFree(b.p); }
        idle.clear();
    }
    void* device_alloc(size_t bytes, bool others_hold_planes);
};

struct DevPool {
    std::mutex mu; std::condition_variable cv;  // slot hand-out
    wri::Slot slots[wri::kMaxSlots];
    int max_slots = 3;
    int users = 0;
    hipStream_t up = nullptr, down = nullptr;  // pageable fallback copies only
    std::mutex up_mu, cu_mu, down_mu;
    std::atomic<double> last_stage_end{0};  // when the last kernel stage of a host call gave up cu_mu (clock_warmup)
    std::atomic<int> active_calls{0};  // wr_encode_* / wr_decode_* calls inside the library on this device
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
        // Where the plane is: one buffer, or -- a large plane -- chunks of 2^ref.shift bytes each that are borrowed together
        // and, on the encoder's side, go back one by one as the host coder has fetched the windows they hold: a plane is
        // complete the moment the quantizer has run and then drains for seconds, so on average half of it is gone and a third
        // more fields in flight fit into the same HBM.  (Chunks as plain buffers with a table the kernels index,
        // wrk::PlaneRef: backing one address range with chunks that come and go -- hipMemMap / hipMemUnmap -- leaves kernels
        // reading the chunks that were there before, profiles/r03/t_vmm_probe.txt.)
        uint8_t* dev = nullptr; size_t dev_bytes = 0;   // dev: the first byte's address (non-null = the plane has storage)
        wrk::PlaneRef ref{};
        // Hand-over check (wr_handover.h): every plane_prepare / plane_release starts a new generation of the plane, the
        // handle the coder gets (io.user) is a ticket for one generation, and a window request that is stale, out of
        // order or not alone in the stream gets no window at all: its coder gives the stream up, the call it belongs to fails
        // (wr_pipeline.cpp: refused_window)
        struct Ticket { PlaneStream* s = nullptr; uint64_t gen = 0; };
        Ticket tickets[8];
        size_t ticket_seq = 0;
        wri::HandoverCheck ho;
        std::vector<DevPlanes::Buf> chunks;             // chunked form: chunk k, p = nullptr once it went back
        size_t released_chunks = 0;
        bool drain = false;                             // hand chunks back as the windows behind them have been fetched
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
    // what the context's last plane kernel (quantizer, histograms, dequantizer) was launched with: printed when the stream
    // reports an error, and kept in a process-wide ring that WR_FAULT_LOG=1 dumps if the runtime aborts (launch_note)
    struct LastLaunch { const char* what = nullptr; int plane = -1; const void* x = nullptr; size_t n = 0; const void* partial = nullptr; wrk::PlaneRef q{}; };
    LastLaunch last_launch;
    // two-phase decode (wr_decode_begin / wr_decode_finish_*): the planes are decoded and wait in ps[].dev
    bool pend_valid = false;
    wr_enc_info pend_info;
    int pend_nx = 0, pend_ny = 0, pend_nz = 0;
    wr_timings pend_tm;
    std::mutex mu;
};

namespace wri {

int ctx_bind(wr_ctx* c);
extern DevPool g_pools[kMaxDevices];
extern std::mutex g_pools_mu;

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

hipError_t slot_ensure(Slot* s, const SlotNeed& need);

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

int ensure_enc_buf(wr_ctx* c, int l, size_t bytes);
int ensure_host_hist(wr_ctx* c, size_t elems);

enum Dir { kUp = 0, kDown = 1 };
struct Piece { void* dst; const void* src; size_t bytes; };

// Starts the pieces of one transfer (SDMA engine for pinned memory, hipMemcpyAsync otherwise) / waits for it
int xfer_start(wr_ctx* c, wr_ctx::Xfer* x, const Piece* pc, int count, Dir dir);
int xfer_wait(wr_ctx::Xfer* x);
int xfer_field(wr_ctx* c, wr_ctx::Xfer* x, void* dst, const void* src, size_t bytes, Dir dir);  // a whole field, in pieces (blocks until done)

using PlaneStream = wr_ctx::PlaneStream;
void plane_release(wr_ctx* c, int l);
// contiguous: the plane must be one array (the local-cutoff quantizer).  unlock_while_waiting: a stage lock the caller holds
// and that must not be held while waiting for device memory that other calls' kernel stages have to free (nullptr: none)
// before_wait: called once, with the context's stream synchronised and the lock still held, before the call starts to wait:
// an encoder hands the planes it has already quantized to their coders there, so that they drain while it waits (nullptr: none)
int plane_prepare(wr_ctx* c, int l, size_t n, bool decode, bool contiguous = false, std::unique_lock<std::mutex>* unlock_while_waiting = nullptr,
                  const std::function<void()>* before_wait = nullptr);
// remembers what a plane kernel of the context is about to be launched with (wr_ctx::last_launch and the fault log's ring)
void launch_note(wr_ctx* c, const char* what, int plane, const void* x, size_t n, const void* partial, const wrk::PlaneRef& q);
std::string launch_describe(const wr_ctx* c);  // " [last plane kernel: ...]" for error messages
void plane_prefetch(wr_ctx* c, int l);
std::string plane_log(wr_ctx* c, int l, size_t n, const wr_enc_info* info, bool encode, size_t len);

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

bool use_fused(int nx, int ny, int nz, int lvl);
void transform_need(int nx, int ny, int nz, int lvl, SlotNeed* need);
int run_transform(wr_ctx* c, Slot* s, double* d_fld, int nx, int ny, int nz, int lvl, double** out);
int inverse_from_planes(wr_ctx* c, Slot* s, double* d_fld, int nx, int ny, int nz, int wlev, const wrk::DequantParams& p);
int read_minmax(wr_ctx* c, const double* d_x, size_t n, bool pending, double* mn, double* mx);
struct ActiveCall {  // RAII: a codec call is inside the library
    DevPool* p;
    explicit ActiveCall(DevPool* pool) : p(pool) { p->active_calls++; }
    ~ActiveCall() { p->active_calls--; }
    ActiveCall(const ActiveCall&) = delete;
    ActiveCall& operator=(const ActiveCall&) = delete;
};
void clock_warmup(wr_ctx* c, size_t n);  // first thing in a host call's kernel stage (cu_mu held); n: elements of the field
int check_dims(int nx, int ny, int nz, const void* dev_ptr);

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

}  // namespace wri
