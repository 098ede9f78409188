// ThreadSanitizer / AddressSanitizer harness for the plane hand-over (CPU build only; sanitizers are not available on the
// GPU pool): the product's coder pool (wr_rangecoder.cpp, streams changing workers forced: pool_test_steal_idle_min(1)) codes
// planes that it sees only through window requests, against a mock of wr_pipeline.cpp's PlaneStream that keeps the
// product's rules (wr_handover.h: generations and tickets, one request at a time, windows in order, nothing after the end)
// and the product's storage behaviour: a plane lives in chunks that are FREED as an encoder's windows pass them (so ASan
// sees a request that reaches behind them) and that the calling thread frees and re-allocates for the next call the
// moment pool_wait returns (so TSan sees a worker that still touches the stream then).  Several calling threads, several
// rounds, planes of every kind; then the violations one by one: each must be refused and leave the plane untouched.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <memory>
#include <thread>
#include <vector>

#include "wr_handover.h"
#include "wr_rangecoder.h"

static std::atomic<unsigned long> g_refused{0};
static std::atomic<unsigned long> g_served{0};

struct MockPlane {
    static constexpr size_t kChunk = (size_t)60000 * 3;  // "device" chunk: 3 coder blocks
    static constexpr size_t kWindow = (size_t)60000 * 2;  // ring window: 2 coder blocks (a window straddles chunks)
    struct Ticket { MockPlane* s; uint64_t gen; };
    wri::HandoverCheck ho;
    Ticket tickets[8];
    size_t ticket_seq = 0;
    std::vector<std::unique_ptr<uint8_t[]>> chunks;  // the plane in "device memory"; reset() = handed back
    size_t n = 0;
    std::unique_ptr<uint8_t[]> ring[2];
    int cur = 1;
    size_t win_first = 0, win_count = 0;
    bool drain = false;
    wrrc::PlaneWindow io{nullptr, nullptr};
    uint8_t scratch[16];

    uint8_t& at(size_t i) { return chunks[i / kChunk][i % kChunk]; }

    void prepare(size_t symbols, bool decode, bool draining)
    {
        n = symbols;
        chunks.clear();
        for (size_t at0 = 0; at0 < n; at0 += kChunk) chunks.emplace_back(new uint8_t[kChunk]);
        for (int b = 0; b < 2; b++) ring[b].reset(new uint8_t[kWindow]);  // fresh: a late writer into the old ring is a use after free
        cur = 1; win_first = win_count = 0; drain = draining;
        const uint64_t g = ho.begin(n);
        Ticket& tk = tickets[ticket_seq++ % 8];
        tk.s = this; tk.gen = g;
        io.window = decode ? window_decode : window_encode;
        io.user = &tk;
    }
    void release()
    {
        chunks.clear();
        ho.retire();
        io.window = nullptr; io.user = nullptr;
    }
    static uint8_t* refuse(size_t*)
    {
        g_refused++;
        return nullptr;  // a refused request hands nothing out (wrrc::PlaneWindow): the coder gives the stream up
    }
    static uint8_t* window_encode(void* user, size_t first, size_t* count)
    {
        const Ticket* tk = static_cast<const Ticket*>(user);
        MockPlane& s = *tk->s;
        if (!s.ho.current(tk->gen)) return refuse(count);
        wri::HandoverGuard guard(s.ho);
        if (!guard.alone) return refuse(count);
        if (s.ho.check_encode(first, *count)) return refuse(count);
        const size_t want = *count < kWindow ? *count : kWindow;
        const int b = s.cur ^ 1;
        for (size_t i = 0; i < want; i++) s.ring[b][i] = s.at(first + i);
        s.cur = b;
        if (s.drain)  // the chunks wholly below `first` go back (wr_pipeline.cpp: plane_window_encode)
            for (size_t k = 0; k < first / kChunk && k < s.chunks.size(); k++) s.chunks[k].reset();
        s.ho.served(first, want, true);
        g_served++;
        *count = want;
        return s.ring[b].get();
    }
    static uint8_t* window_decode(void* user, size_t first, size_t* count)
    {
        const Ticket* tk = static_cast<const Ticket*>(user);
        MockPlane& s = *tk->s;
        if (!s.ho.current(tk->gen)) return refuse(count);
        wri::HandoverGuard guard(s.ho);
        if (!guard.alone) return refuse(count);
        if (s.ho.check_decode(first, *count)) return refuse(count);
        if (s.win_count) {  // the window handed out before is complete: it goes to the "device"
            for (size_t i = 0; i < s.win_count; i++) s.at(s.win_first + i) = s.ring[s.cur][i];
            s.win_count = 0;
        }
        if (*count == 0) { s.ho.end(); return nullptr; }
        const int b = s.cur ^ 1;
        s.cur = b;
        s.win_first = first;
        s.win_count = *count < kWindow ? *count : kWindow;
        s.ho.served(first, s.win_count, false);
        g_served++;
        *count = s.win_count;
        return s.ring[b].get();
    }
};

static unsigned long long rng_state = 88172645463325252ull;
static unsigned rnd(unsigned long long& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 11); }

static void fill_plane(std::vector<uint8_t>& p, int kind, unsigned long long& s)
{
    for (size_t i = 0; i < p.size(); i++) {
        const unsigned r = rnd(s);
        p[i] = kind == 0 ? (uint8_t)(r & 255) : kind == 1 ? ((r & 7) ? 127 : 128) : kind == 2 ? (uint8_t)(100 + (r & 3)) : ((r & 1023) ? 7 : (uint8_t)(r >> 12));
    }
}

int main()
{
    wrrc::pool_test_steal_idle_min(1);  // any idle worker takes over half of the fullest session at its next block boundary
    constexpr int kCallers = 4, kPlanes = 5, kRounds = 3;
    const size_t sizes[kPlanes] = {(size_t)60000 * 7 + 123, (size_t)60000 * 6, (size_t)60000 * 9 + 1, (size_t)60000 * 4 + 59999, (size_t)60000 * 8};
    wrrc::pool_configure(6, 4);
    std::atomic<int> failures{0};
    auto caller = [&](int id) {
        unsigned long long seed = 0x9E3779B97F4A7C15ull * (unsigned long long)(id + 1);
        std::vector<MockPlane> ps(kPlanes);
        for (int round = 0; round < kRounds && !failures.load(); round++) {
            std::vector<std::vector<uint8_t>> plane(kPlanes), want(kPlanes), got(kPlanes);
            std::vector<size_t> want_len(kPlanes);
            for (int l = 0; l < kPlanes; l++) {
                plane[l].resize(sizes[l]);
                fill_plane(plane[l], (l + round + id) % 4, seed);
                want[l].resize(wrrc::encode_bound(sizes[l]));
                want_len[l] = wrrc::encode_plane(plane[l].data(), sizes[l], want[l].data(), nullptr);
                got[l].assign(wrrc::encode_bound(sizes[l]), 0);
            }
            // ---- encode: the planes become complete one after the other and are handed to the pool one by one, like
            // submit_plane in wr_codec.cpp; the chunks drain under the coder
            {
                wrrc::PlaneJob jobs[kPlanes];
                wrrc::JobBatch batch;
                for (int l = 0; l < kPlanes; l++) {
                    ps[l].prepare(sizes[l], false, true);
                    for (size_t i = 0; i < sizes[l]; i++) ps[l].at(i) = plane[l][i];  // "the quantizer has run"
                    jobs[l].kind = wrrc::PlaneJob::kEncode;
                    jobs[l].src = nullptr; jobs[l].io = &ps[l].io; jobs[l].dst = got[l].data(); jobs[l].n = sizes[l];
                    if (!wrrc::pool_submit(&jobs[l], 1, &batch)) { failures++; return; }
                }
                wrrc::pool_wait(&batch);
                for (int l = 0; l < kPlanes; l++) {
                    if (jobs[l].result != want_len[l] || memcmp(got[l].data(), want[l].data(), want_len[l])) { printf("caller %d round %d: encoded plane %d differs\n", id, round, l); failures++; }
                    ps[l].release();  // PlaneHold: the storage goes back the moment the coders are done
                }
            }
            // ---- decode through windows into fresh planes; a stale writer would land in freed chunks
            {
                wrrc::PlaneJob jobs[kPlanes];
                wrrc::JobBatch batch;
                for (int l = 0; l < kPlanes; l++) {
                    ps[l].prepare(sizes[l], true, false);
                    jobs[l].kind = wrrc::PlaneJob::kDecode;
                    jobs[l].src = want[l].data(); jobs[l].src_len = want_len[l]; jobs[l].dst = nullptr; jobs[l].io = &ps[l].io; jobs[l].n = sizes[l];
                }
                if (!wrrc::pool_submit(jobs, kPlanes, &batch)) { failures++; return; }
                wrrc::pool_wait(&batch);
                for (int l = 0; l < kPlanes; l++) {
                    bool same = jobs[l].result == sizes[l] && ps[l].ho.ended;
                    for (size_t i = 0; same && i < sizes[l]; i++) same = ps[l].at(i) == plane[l][i];
                    if (!same) { printf("caller %d round %d: decoded plane %d differs\n", id, round, l); failures++; }
                    ps[l].release();
                }
            }
        }
    };
    {
        std::vector<std::thread> ts;
        for (int k = 0; k < kCallers; k++) ts.emplace_back(caller, k);
        for (auto& t : ts) t.join();
    }
    printf("window requests served: %lu, refused: %lu, streams that changed workers: %lu\n", g_served.load(), g_refused.load(), wrrc::pool_streams_moved());
    if (failures.load() || g_refused.load()) { printf("hand-over test FAILED\n"); return 1; }
    if (wrrc::pool_streams_moved() == 0) printf("(no stream changed workers in this run)\n");
    wrrc::pool_configure(0, 0);

    // ---- the violations, one by one: each is refused and does not reach the plane
    {
        const size_t n = (size_t)60000 * 5;
        MockPlane p;
        p.prepare(n, false, true);
        for (size_t i = 0; i < n; i++) p.at(i) = (uint8_t)i;
        const wrrc::PlaneWindow stale = p.io;  // the handle of this call ...
        size_t c = n;
        uint8_t* w = p.io.window(p.io.user, 0, &c);
        if (!w || c != MockPlane::kWindow || g_refused.load() != 0) { printf("first window refused\n"); return 1; }
        c = n - 2 * MockPlane::kWindow;  // a gap
        (void)p.io.window(p.io.user, 2 * MockPlane::kWindow, &c);
        if (g_refused.load() != 1) { printf("out-of-order window served\n"); return 1; }
        c = n;  // going back
        (void)p.io.window(p.io.user, 0, &c);
        if (g_refused.load() != 2) { printf("repeated window served\n"); return 1; }
        p.ho.inside.fetch_add(1);  // another coder inside
        c = n - MockPlane::kWindow;
        (void)p.io.window(p.io.user, MockPlane::kWindow, &c);
        p.ho.inside.fetch_sub(1);
        if (g_refused.load() != 3) { printf("second coder inside the stream served\n"); return 1; }
        for (size_t first = MockPlane::kWindow; first < n; first += MockPlane::kWindow) { c = n - first; (void)p.io.window(p.io.user, first, &c); }
        if (g_refused.load() != 3 || !p.ho.ended) { printf("in-order windows refused\n"); return 1; }
        c = 1;  // after the last window
        (void)p.io.window(p.io.user, n - 1, &c);
        if (g_refused.load() != 4) { printf("window after the end served\n"); return 1; }
        p.release();
        p.prepare(n, false, true);  // ... the next call on the same stream: the old handle is void, the new plane intact
        for (size_t i = 0; i < n; i++) p.at(i) = 0x5a;
        c = n;
        (void)stale.window(stale.user, 0, &c);
        if (g_refused.load() != 5 || p.ho.next_first != 0 || !p.chunks[0]) { printf("stale ticket served\n"); return 1; }
        // decoder: nothing after the end-of-stream request
        p.release();
        p.prepare(n, true, false);
        c = n; (void)p.io.window(p.io.user, 0, &c);
        c = 0; (void)p.io.window(p.io.user, MockPlane::kWindow, &c);
        c = n - MockPlane::kWindow; (void)p.io.window(p.io.user, MockPlane::kWindow, &c);
        if (g_refused.load() != 6) { printf("decoder window after the end served\n"); return 1; }
    }
    // ---- a real coder on a refused window: the stream ends at once with (size_t)-1, nothing is written or read beyond what
    // was served (ASan: the output buffers are exact-size; the advisor's case: an encoder that "ran on" on a zeroed scratch
    // window against the GPU's histograms had a range of zero and never came back)
    {
        const size_t n = (size_t)60000 * 6 + 77;
        MockPlane p;
        p.prepare(n, false, true);
        unsigned long long seed = 12345;
        std::vector<uint8_t> plane(n);
        fill_plane(plane, 1, seed);
        for (size_t i = 0; i < n; i++) p.at(i) = plane[i];
        std::vector<uint16_t> hist((n / 60000 + 1) * 256, 0);
        for (size_t i = 0; i < n; i++) hist[(i / 60000) * 256 + plane[i]]++;
        const wrrc::PlaneWindow stale = p.io;
        p.release();
        p.prepare(n, false, true);  // the next call owns the stream; the coder below still holds the old handle
        const unsigned long refused0 = g_refused.load();
        wrrc::pool_configure(2, 4);
        std::vector<uint8_t> out(wrrc::encode_bound_hist(hist.data(), n) + wrrc::kFailedBlockSlack);
        wrrc::PlaneJob job;
        wrrc::JobBatch batch;
        job.kind = wrrc::PlaneJob::kEncode; job.src = nullptr; job.io = &stale; job.dst = out.data(); job.n = n; job.hist = hist.data();
        if (!wrrc::pool_submit(&job, 1, &batch)) { printf("pool refused the job\n"); return 1; }
        wrrc::pool_wait(&batch);
        if (job.result != (size_t)-1 || g_refused.load() == refused0 || p.ho.next_first != 0) { printf("encoder on a refused window: result %zu\n", job.result); return 1; }
        std::vector<uint8_t> stream(wrrc::encode_bound(n));
        const size_t len = wrrc::encode_plane(plane.data(), n, stream.data(), nullptr);
        MockPlane q;
        q.prepare(n, true, false);
        const wrrc::PlaneWindow stale_d = q.io;
        q.release();
        q.prepare(n, true, false);
        wrrc::PlaneJob dj;
        wrrc::JobBatch db;
        dj.kind = wrrc::PlaneJob::kDecode; dj.src = stream.data(); dj.src_len = len; dj.dst = nullptr; dj.io = &stale_d; dj.n = n;
        if (!wrrc::pool_submit(&dj, 1, &db)) { printf("pool refused the job\n"); return 1; }
        wrrc::pool_wait(&db);
        if (dj.result != (size_t)-1 || q.ho.next_first != 0 || q.ho.ended) { printf("decoder on a refused window: result %zu\n", dj.result); return 1; }
        wrrc::pool_configure(0, 0);
    }
    (void)rng_state;
    printf("hand-over run OK\n");
    return 0;
}
