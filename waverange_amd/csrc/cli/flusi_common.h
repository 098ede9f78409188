// flusi_common.h -- shared bits of the FluSI wrenc / wrdec front-ends: parameter sources
// (reference src/flusi/main_enc.cpp:95-199, main_dec.cpp:80-116) and a two-slot device pipeline.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <future>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/waverange_amd.h"

namespace flusi {

inline std::string trim(const std::string& s)
{
    const char* ws = " \t\v\r\n";
    size_t a = s.find_first_not_of(ws);
    if (a == std::string::npos) return "";
    return s.substr(a, s.find_last_not_of(ws) - a + 1);
}

// Four string parameters in the reference's order.  Sources by priority: a file named `inmeta`
// in the working directory (new "&key = value" or old one-value-per-line format), 4 arguments,
// interactive prompts.  keys/prompts are given by the caller.  Returns false on a parse error.
inline bool get_params(int argc, char** argv, bool use_inmeta, const char* const keys[4], const char* const prompts[4],
                       const char* usage, std::string out[4])
{
    std::ifstream meta("inmeta");
    if (use_inmeta && !meta.fail()) {
        std::cout << "==== inmeta exists. ====" << std::endl;
        std::vector<std::string> lines;
        for (std::string l; std::getline(meta, l);) lines.push_back(l);
        bool keyed = false;
        for (const std::string& raw : lines) {
            std::string l = trim(raw);
            if (l.empty() || l[0] != '&') continue;
            if (std::count(l.begin(), l.end(), '=') != 1 || trim(l.substr(l.find('=') + 1)).empty()) {
                std::cout << "==== Error : malformed sentence :" << l << " ====" << std::endl;
                return false;
            }
            keyed = true;
            std::string k = trim(l.substr(0, l.find('='))), v = trim(l.substr(l.find('=') + 1));
            std::transform(k.begin(), k.end(), k.begin(), ::tolower);
            for (int i = 0; i < 4; i++) if (k == keys[i]) out[i] = v;
        }
        if (!keyed) {
            std::cout << "==== read parameters from inmeta as old format. ====" << std::endl;
            for (int i = 0; i < 4 && i < (int)lines.size(); i++) out[i] = lines[i];
        }
        for (int i = 0; i < 4; i++) std::cout << (keys[i] + 1) << " = " << out[i] << std::endl;
        return true;
    }
    std::cout << usage;
    if (argc == 5) {
        std::cout << "automatic mode.";
        for (int i = 0; i < 4; i++) out[i] = argv[i + 1];
    } else {
        for (int i = 0; i < 4; i++) { std::cout << prompts[i]; std::getline(std::cin, out[i]); }
    }
    return true;
}

// Datasets in flight on one GPU: while dataset k is in its host range-coding phase on one context, datasets k+1, ...
// are read, uploaded and transformed through others (wr_encode_host / wr_decode_host: the library overlaps the copies
// of one call with the kernels and the host coding of the others; with more than one dataset its coder pool codes the
// plane streams of all of them on one worker per CPU).  How many: wr_autotune_batch for datasets of this size
// (WR_CLI_PIPELINE overrides; at least 1, at most the number of datasets).  HDF5 stays on the main thread.
struct Pipeline {
    std::vector<wr_ctx*> ctx;
    bool open(size_t field_elems, int ndatasets)
    {
        int dev = 0;
        if (const char* e = getenv("WR_DEVICE")) dev = atoi(e);
        int depth = ndatasets > 1 ? wr_autotune_batch(field_elems, ndatasets) : 1;
        if (const char* e = getenv("WR_CLI_PIPELINE")) { const int k = atoi(e); if (k >= 1) depth = k; }
        if (depth > ndatasets) depth = ndatasets;
        if (depth < 1) depth = 1;
        for (int i = 0; i < depth; i++) {
            wr_ctx* c = nullptr;
            if (wr_ctx_create(&c, dev, nullptr) != WR_OK) { std::cerr << "wrenc/wrdec: " << wr_last_error() << std::endl; return false; }
            ctx.push_back(c);
        }
        return true;
    }
    int depth() const { return (int)ctx.size(); }
    ~Pipeline()
    {
        for (wr_ctx* c : ctx) wr_ctx_destroy(c);
    }
};

// min / max as a scan with libm fmin / fmax gives them (NaNs skipped), four elements at a time (the log lines only)
inline void minmax(const double* v, size_t n, double* lo_out, double* hi_out)
{
    double l[4] = {v[0], v[0], v[0], v[0]}, h[4] = {v[0], v[0], v[0], v[0]};
    size_t j = 1;
    for (; j + 4 <= n; j += 4)
        for (int k = 0; k < 4; k++) {
            const double x = v[j + k];
            l[k] = (x <= l[k] || l[k] != l[k]) ? x : l[k];
            h[k] = (x >= h[k] || h[k] != h[k]) ? x : h[k];
        }
    double lo = l[0], hi = h[0];
    for (int k = 1; k < 4; k++) { lo = fmin(lo, l[k]); hi = fmax(hi, h[k]); }
    for (; j < n; j++) { lo = fmin(lo, v[j]); hi = fmax(hi, v[j]); }
    *lo_out = lo; *hi_out = hi;
}

// WR_CLI_TIMING=1: one line per dataset with its phases in seconds since the tool started (stderr) -- when it was read,
// when its codec call ran and how much of that was the wait for a work-space slot, the device phase (copies + kernels)
// and the slowest plane's range coder, when it was written: shows dataset k+1's upload and kernels inside dataset k's call
struct Clock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    bool on = getenv("WR_CLI_TIMING") && atoi(getenv("WR_CLI_TIMING"));
    double now() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
struct Phases {
    double read0 = 0, read1 = 0, call0 = 0, call1 = 0, write0 = 0, write1 = 0;
    wr_timings tm = {};
    void print(const std::string& name) const
    {
        fprintf(stderr, "timing dset=%s read %.3f-%.3f call %.3f-%.3f (slot wait %.3f, device phase %.3f, slowest plane coder %.3f, "
                        "h2d %.1f ms, d2h %.1f ms, transform %.2f ms, quantizer %.2f ms) write %.3f-%.3f\n",
                name.c_str(), read0, read1, call0, call1, tm.wait, tm.gpu, tm.rangecoder, tm.h2d_ms, tm.d2h_ms, tm.transform_ms, tm.quant_ms,
                write0, write1);
    }
};

// malloc'd bytes (a coded buffer's worst case is 8 bytes per element, of which a fraction is ever touched)
struct RawBytes {
    unsigned char* p = nullptr; size_t bytes = 0;
    RawBytes() = default;
    RawBytes(const RawBytes&) = delete;
    RawBytes& operator=(const RawBytes&) = delete;
    ~RawBytes() { free(p); }
    bool allocate(size_t n) { free(p); p = static_cast<unsigned char*>(malloc(n ? n : 1)); bytes = p ? n : 0; return p != nullptr; }
    void release() { free(p); p = nullptr; bytes = 0; }
};

}  // namespace flusi
