// wr_coder_hooks.cpp -- the host range coder alone behind the C ABI (rows a6/a7/a10 of SURVEY.md 8a): whole planes,
// interleaved planes, the coder pool, the 16-lane loops, and the windowed symbol path with host buffers standing in for
// device-resident planes (test hooks).
#include "wr_internal.h"

using namespace wri;

extern "C" {

size_t wr_range_encode_bound(size_t n) { return wrrc::encode_bound(n); }
size_t wr_range_encode_bound_hist(const unsigned short* hists, size_t n) { return wrrc::encode_bound_hist(hists, n); }
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

}  // extern "C"
