// wr_rangecoder_vec.h -- AVX-512 decoder loop for blocks with two dominant symbols, 16 plane streams at once
// (internal interface between wr_rangecoder.cpp and wr_rangecoder_avx512.cpp).
//
// The leading bit planes of a smooth field hold two symbols almost exclusively (p = 0.8 / 0.2 on the synthetic
// field): their decoder step needs no division and no table look-up -- renormalise, range / 60000, two
// multiplies, two interval tests (wr_rangecoder.cpp, decode_symbols_multi, the `mps_on` path).  That is pure
// 32-bit lane arithmetic, so 16 streams of 16 different planes advance in one vector step; the scalar loop
// spends ~60 instructions per symbol on it, this one ~4.  Arithmetic per lane is rangecod.c:294-351 unchanged.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace wrrc {

constexpr int kVecLanes = 16;

// One full block (60000 symbols) of up to 16 streams.  Lane state in, lane state out.
struct VecBlock {
    uint32_t active;                 // lane mask
    uint32_t low[kVecLanes], range[kVecLanes];
    const uint8_t* ptr[kVecLanes];   // next unread stream byte; ptr[-1] is the byte held back (rangecod.c:297-299)
    uint8_t* dst[kVecLanes];         // 60000 symbols each
    // the two dominant symbols of the lane's block: interval start, width, "is the largest symbol present"
    uint32_t lt[2][kVecLanes], sy[2][kVecLanes], is_top[2][kVecLanes], sym[2][kVecLanes];
    const void* model[kVecLanes];    // handed to `other` for a symbol outside the two
};
// symbol outside the two dominant ones (rare): the scalar look-up path on that lane's model
typedef uint32_t (*VecOther)(const void* model, uint32_t* low, uint32_t* range, uint32_t help);

bool vec_available();  // the CPU has AVX-512 F/BW/DQ/VL and WR_NO_AVX512 is not set
void vec_decode_two_symbol_block(VecBlock* b, VecOther other);

}  // namespace wrrc
