// wr_handover.h -- the rules of the plane hand-over between a codec call and the host coder, in checkable form.
//
// A quantized plane of a call in flight lives in device memory (wr_pipeline.cpp: PlaneStream); the host coder that reads
// (encode) or writes (decode) it is another thread -- a thread of the call, or a worker of the process-wide pool that may
// pass the stream on to another worker between two blocks -- and sees the plane only through window requests
// (wrrc::PlaneWindow).  The reference has no such hand-over: a plane is quantized, then coded, then never touched again,
// all on one thread (src/core/wrappers.cpp:381-447); here the same order has to hold across threads:
//   * the stream belongs to the calling thread from plane_prepare until the coder is started, and again once the coder has
//     ended; in between ONLY the coder's window requests touch it, one at a time;
//   * an encoder's windows come in ascending order without gaps and end with the plane; a decoder's likewise, and nothing
//     comes after its end-of-stream request;
//   * a request made with the handle of an earlier call on the same context (a coder that outlived its call) must not
//     reach the plane of the call that owns the stream now.
// Every plane_prepare / plane_release starts a new GENERATION; the handle a coder gets is a ticket for one generation.
// Host-only, no HIP: tests/native/handover_tsan.cpp runs the coder pool against it under ThreadSanitizer.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <atomic>

namespace wri {

struct HandoverCheck {
    std::atomic<uint64_t> gen{0};
    std::atomic<int> inside{0};  // window requests inside the stream right now
    size_t n = 0;                // symbols of the plane
    size_t next_first = 0;       // where the next window must start
    bool ended = false;          // encode: the last window has been handed out; decode: the coder has said so

    // a new generation for a plane of `symbols` symbols (calling thread, no coder attached); returns it for the ticket
    uint64_t begin(size_t symbols)
    {
        n = symbols; next_first = 0; ended = symbols == 0;
        return gen.fetch_add(1) + 1;
    }
    // the plane's storage goes away: tickets cut so far are void
    void retire() { gen.fetch_add(1); }

    bool current(uint64_t ticket_gen) const { return ticket_gen == gen.load(); }

    // why a window request must be refused (nullptr: serve it).  count: what the coder says is left from `first` on.
    const char* check_encode(size_t first, size_t count) const
    {
        if (ended) return "a window after the last one";
        if (first != next_first || first >= n || count != n - first) return "windows out of order";
        return nullptr;
    }
    // (a decoder may end early -- a stream that does not hold n symbols -- but never goes back, skips or overruns the plane)
    const char* check_decode(size_t first, size_t count) const
    {
        if (ended) return "a window after the end of the stream";
        if (count && (first != next_first || first >= n || count > n - first)) return "windows out of order";
        return nullptr;
    }
    void served(size_t first, size_t handed, bool encode)
    {
        next_first = first + handed;
        if (encode && next_first >= n) ended = true;
    }
    void end() { ended = true; }
};

// counts the requests inside a stream: `alone` is false if another one is in there already (two coders on one plane)
struct HandoverGuard {
    HandoverCheck& h;
    const bool alone;
    explicit HandoverGuard(HandoverCheck& hc) : h(hc), alone(hc.inside.fetch_add(1) == 0) {}
    ~HandoverGuard() { h.inside.fetch_sub(1); }
    HandoverGuard(const HandoverGuard&) = delete;
    HandoverGuard& operator=(const HandoverGuard&) = delete;
};

}  // namespace wri
