// batch.h -- what the command-line tools (wrenc / wrdec, FluSI, MSSG) share to keep several independent fields in
// flight around the drop-in entry points: how many (asked of the library when it is libwaverange_amd; the tools also
// link against the reference's libwaverange, which has no such symbol), an ordering gate between the reading thread and
// the writing thread, buffers that are not zero-filled, and a min/max scan with fmin / fmax semantics.
#pragma once
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <new>
#include <stdexcept>

// Weak: resolved when the tool runs on libwaverange_amd, null on the reference's library (tests/test_cli.py builds the
// same sources against both).
extern "C" int wr_autotune_batch(size_t field_elems, int nfields) __attribute__((weak));

namespace wrcli {

// Fields to keep in flight: WR_CLI_PIPELINE if set (0 = strictly serial, the reference's behaviour and log order), else
// what the library suggests for fields of this size (it also starts its coder pool), else 2; never more than nf - 1
// ... 0 for a single field.
inline int fields_in_flight(size_t field_elems, int nf)
{
    if (nf <= 1) return 0;
    int depth = -1;
    if (const char* e = getenv("WR_CLI_PIPELINE")) depth = atoi(e);
    if (depth < 0) depth = wr_autotune_batch ? wr_autotune_batch(field_elems, nf) : 2;
    else if (depth > 1 && wr_autotune_batch) (void)wr_autotune_batch(field_elems, nf);  // the pool, sized to the CPUs
    if (depth > nf) depth = nf;
    return depth < 0 ? 0 : depth;
}

// malloc'd bytes: the worst-case coded buffer of a field is 8 bytes per element, of which a fraction is ever touched
class RawBuffer {
public:
    RawBuffer() = default;
    RawBuffer(const RawBuffer&) = delete;
    RawBuffer& operator=(const RawBuffer&) = delete;
    ~RawBuffer() { release(); }
    void allocate(size_t bytes)
    {
        release();
        p_ = static_cast<unsigned char*>(malloc(bytes ? bytes : 1));
        if (!p_) throw std::bad_alloc();
    }
    void release() { free(p_); p_ = nullptr; }
    unsigned char* data() const { return p_; }

private:
    unsigned char* p_ = nullptr;
};

// Reader -> writer hand-over in field order with at most `depth` fields between the two.
class InFlight {
public:
    explicit InFlight(int depth) : free_(depth) {}
    bool enter()  // reader: before it touches the next field; false after abort()
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return free_ > 0 || aborted_; });
        if (aborted_) return false;
        free_--;
        return true;
    }
    void launched(int it) { { std::lock_guard<std::mutex> lk(mu_); launched_ = it + 1; } cv_.notify_all(); }
    void wait_launched(int it)  // writer
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return launched_ > it || aborted_; });
        if (launched_ <= it) throw std::runtime_error("field pipeline aborted");
    }
    void leave() { { std::lock_guard<std::mutex> lk(mu_); free_++; } cv_.notify_all(); }
    void abort() { { std::lock_guard<std::mutex> lk(mu_); aborted_ = true; } cv_.notify_all(); }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    int free_, launched_ = 0;
    bool aborted_ = false;
};

// min / max as a scan with libm fmin / fmax gives them (NaNs skipped; of equal values -- only +0 / -0 can tell --
// the LAST one wins, glibc x86-64), four elements at a time
inline void minmax(const double* v, size_t n, double* lo_out, double* hi_out)
{
    double lo = v[0], hi = v[0];
    size_t j = 1;
    if (n >= 16) {
        double l[4] = {lo, lo, lo, lo}, h[4] = {hi, hi, hi, hi};
        for (; j + 4 <= n; j += 4)
            for (int k = 0; k < 4; k++) {
                const double x = v[j + k];
                l[k] = (x <= l[k] || l[k] != l[k]) ? x : l[k];  // NaN in x: both comparisons false, l[k] stays
                h[k] = (x >= h[k] || h[k] != h[k]) ? x : h[k];
            }
        lo = l[0]; hi = h[0];
        for (int k = 1; k < 4; k++) { lo = std::fmin(lo, l[k]); hi = std::fmax(hi, h[k]); }
    }
    for (; j < n; j++) { lo = std::fmin(lo, v[j]); hi = std::fmax(hi, v[j]); }
    if (lo == 0.0 || hi == 0.0) {  // the sign of a zero extremum: the plain scan decides
        lo = hi = v[0];
        for (size_t k = 1; k < n; k++) { lo = std::fmin(lo, v[k]); hi = std::fmax(hi, v[k]); }
    }
    *lo_out = lo; *hi_out = hi;
}

}  // namespace wrcli
