// wr_rangecoder.h -- host-side plane bit-stream coder (product code).
//
// Bit-exact with the reference's per-plane stream: Schindler's rngcod13 primitives
// (reference src/rangecod/rangecod.c:170-404) driven by the 60000-symbol block model of
// src/core/wrappers.cpp:68-224.  The range coder stays on the host by design (it is one
// serial recurrence per plane); planes are coded concurrently: one thread each, or several
// planes interleaved in one thread's loop when cores are scarcer than planes.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <condition_variable>
#include <mutex>

namespace wrrc {

constexpr uint32_t kBlock = 60000;  // reference src/core/defs.h:36
constexpr int kMaxStreams = 4;      // planes interleaved in one symbol loop by encode_planes / decode_planes
constexpr int kMaxDecStreams = 4;   // ... by a pool worker's decoder loop (5 and 6 were measured: slower in the compiler's loops, which run out of
                                    // registers, and no gain in hand-written ones with packed state: profiles/r04/NOTES.md)

// upper bound on the stream length for n symbols
size_t encode_bound(size_t n);
// the same for a plane whose per-block byte histograms are known (uint16[256] per block, n/60000+1 blocks, as encode_plane
// takes them): the blocks' entropy plus headers plus the coder's worst-case rounding loss -- within ~0.2 % of the stream
size_t encode_bound_hist(const uint16_t* hists, size_t n);

// Windowed access to the symbol side of a plane that is not in host memory as a whole (it lives in device memory and
// passes through a small pinned ring, wr_pipeline.cpp).  `count` comes in as what is left of the plane from `first` on and
// goes out as the length of the window handed back: a multiple of kBlock unless the plane ends in it.
//   encoder: the symbols [first, first + count); the pointer stays valid until the next call.
//   decoder: room for the symbols [first, first + count); the window handed out before is complete by then.  A call
//            with *count == 0 ends the stream (the last window is complete; nothing is handed back).
// REFUSAL: a request for symbols (*count != 0 on entry) that comes back with a null pointer has been turned down by the
// plane's owner (a stale handle, windows out of order, two coders in one stream: wr_handover.h).  The coder then gives the
// stream up at once -- it touches no symbol and writes nothing further -- and its result is (size_t)-1.
struct PlaneWindow {
    uint8_t* (*window)(void* user, size_t first, size_t* count);
    void* user;
};

// Encode n symbols; `out` must hold encode_bound(n) bytes.  `hists`, when non-null, holds
// per-block byte histograms (uint16[256] per block, n/60000+1 blocks) computed on the GPU.
// Returns the stream length, (size_t)-1 if the stream was given up (a refused window; histograms that are not the plane's:
// they do not add up to a block, or a symbol they count zero times turns up -- at most kFailedBlockSlack bytes beyond what
// encode_bound_hist says have been written by then).
constexpr size_t kFailedBlockSlack = 131072;  // one block at 2 bytes per symbol + its header
size_t encode_plane(const uint8_t* sym, size_t n, uint8_t* out, const uint16_t* hists);

// `count` planes of n symbols each on the calling thread, their symbol loops interleaved in
// groups of up to kMaxStreams (same bytes as encode_plane on each).  hists may be null, and so
// may its entries.
void encode_planes(int count, const uint8_t* const* sym, size_t n, uint8_t* const* out, const uint16_t* const* hists, size_t* lens,
                   const PlaneWindow* const* io = nullptr,  // io[k] non-null: plane k comes through windows, sym[k] is ignored
                   const size_t* limits = nullptr);         // limits[k]: encode_bound_hist of plane k's histograms (see PlaneJob::dst_limit)

// Decode a stream into exactly n symbols.  Returns the number of symbols the stream held
// (== n for a well-formed stream; never writes more than n symbols, never reads past len).
size_t decode_plane(const uint8_t* in, size_t len, uint8_t* sym, size_t n);

// `count` streams of n symbols each on the calling thread, interleaved like encode_planes;
// produced[k] as decode_plane's return value, (size_t)-1 for a stream that is not decodable.
void decode_planes(int count, const uint8_t* const* in, const size_t* len, uint8_t* const* sym, size_t n, size_t* produced,
                   const PlaneWindow* const* io = nullptr);  // io[k] non-null: plane k goes out through windows, sym[k] is ignored

// `count` dominant-symbol planes (any lengths) on the calling thread, up to 16 at a time in the AVX-512 loop
// (wr_rangecoder_vec.h); false if the CPU lacks AVX-512.  Same symbols as decode_plane on each.
bool decode_planes_vec(int count, const uint8_t* const* in, const size_t* len, uint8_t* const* sym, const size_t* n, size_t* produced,
                       const PlaneWindow* const* io = nullptr);

// `count` planes of any kind on the calling thread, up to 16 at a time in the AVX-512 encoder loop; false if the CPU
// lacks AVX-512.  Same bytes as encode_plane on each.
bool encode_planes_vec(int count, const uint8_t* const* sym, const size_t* n, uint8_t* const* out, size_t* lens,
                       const PlaneWindow* const* io = nullptr);

// ---- process-wide coder pool: plane streams of ALL concurrent encode / decode calls are coded by a fixed set
// of worker threads.  A worker interleaves up to 3 encoder or up to `dec_streams` (<= kMaxDecStreams) decoder
// streams in one symbol loop, whichever fields they belong to; streams join at block boundaries as others
// end.  (Per call there are only 3-4 planes: fewer than a decoder loop can keep in flight.)
struct JobBatch {  // completion of a group of jobs (owned by the submitter)
    std::mutex mu;
    std::condition_variable cv;
    int remaining = 0;
};
struct PlaneJob {
    enum Kind { kEncode = 0, kDecode = 1 };
    int kind = kEncode;
    const uint8_t* src = nullptr;  // encode: n symbols; decode: the stream
    size_t src_len = 0;            // decode: stream length
    uint8_t* dst = nullptr;        // encode: encode_bound(n) bytes; decode: n symbols
    size_t n = 0;
    const uint16_t* hist = nullptr;  // encode: per-block histograms from the GPU, or null
    size_t dst_limit = 0;            // encode with `hist`: encode_bound_hist(hist, n) -- the stream is given up once it is longer than its own
                                     // histograms allow (they are not this plane's); dst then holds dst_limit + kFailedBlockSlack bytes.  0: no check
    const PlaneWindow* io = nullptr; // the symbol side (encode: src, decode: dst) comes / goes through windows instead
    size_t result = 0;             // encode: stream length, (size_t)-1 if given up; decode: symbols the stream held, (size_t)-1 if undecodable or given up
    double seconds = 0;            // from the moment a worker took the job to its end
    double submitted = 0;          // (pool) when the job was queued
    JobBatch* batch = nullptr;
};
void pool_configure(int nthreads, int dec_streams);  // nthreads = 0 stops the pool; dec_streams < 1 keeps the setting
void pool_test_steal_idle_min(int workers);  // native tests: a hand-over of streams needs only this many idle workers (0: the product's rule)
int pool_threads();
constexpr int kLoopKinds = 4;
// per loop kind {scalar enc, scalar dec, vector dec (dominant symbols), vector enc}: worker seconds in block steps, stream-blocks advanced
void pool_loop_stats(double seconds[kLoopKinds], double blocks[kLoopKinds]);
unsigned long pool_streams_moved();  // streams that changed workers between two blocks (an idle worker took over half of the fullest session)
// decoder jobs that are queued and no worker has taken yet: {planes for the scalar loops (noise), planes for the 16-lane sessions}
void pool_queued_decode(int* scalar_jobs, int* vector_jobs);
double pool_queue_seconds();  // time jobs have waited in the pool's queues before a worker took them, summed over jobs
double pool_idle_seconds();  // time the workers have spent waiting for a job since the process started, summed over workers
// The jobs must stay valid until pool_wait returns.  False (nothing queued) if the pool has no workers -- it may have
// been stopped by another thread since the caller looked at pool_threads(): the caller then codes the planes itself.
bool pool_submit(PlaneJob* jobs, int count, JobBatch* batch);
void pool_wait(JobBatch* batch);

}  // namespace wrrc
