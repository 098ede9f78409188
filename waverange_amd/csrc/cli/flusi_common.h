// flusi_common.h -- shared bits of the FluSI wrenc / wrdec front-ends: parameter sources
// (reference src/flusi/main_enc.cpp:95-199, main_dec.cpp:80-116) and a two-slot device pipeline.
#pragma once
#include <algorithm>
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

// Two contexts on one GPU: while dataset k is in its host range-coding phase on one context, dataset
// k+1 is uploaded and transformed through the other (wr_encode_host / wr_decode_host: the library
// overlaps the copies of one call with the kernels and the host coding of the other).
struct Pipeline {
    wr_ctx* ctx[2] = {nullptr, nullptr};
    bool open()
    {
        int dev = 0;
        if (const char* e = getenv("WR_DEVICE")) dev = atoi(e);
        for (int i = 0; i < 2; i++)
            if (wr_ctx_create(&ctx[i], dev, nullptr) != WR_OK) { std::cerr << "wrenc/wrdec: " << wr_last_error() << std::endl; return false; }
        return true;
    }
    ~Pipeline()
    {
        for (int i = 0; i < 2; i++)
            if (ctx[i]) wr_ctx_destroy(ctx[i]);
    }
};

}  // namespace flusi
