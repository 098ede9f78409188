// wr_rangecoder_vec.h -- AVX-512 decoder loop for blocks with a few dominant symbols, 16 plane streams at once
// (internal interface between wr_rangecoder.cpp and wr_rangecoder_avx512.cpp).
//
// The leading bit planes of a smooth field hold two symbols almost exclusively (p = 0.8 / 0.2 on the synthetic
// field), the trailing plane of a loosely coded one three or four: their decoder step needs no division and no
// table look-up -- renormalise, range / 60000, then per candidate symbol a multiply and an interval test
// (wr_rangecoder.cpp, decode_symbols_multi, the `mps_on` / `few` paths).  That is pure 32-bit lane
// arithmetic, so 16 streams of 16 different planes advance in one vector step; the scalar loop spends ~60
// instructions per symbol on it, this one ~6.  Arithmetic per lane is rangecod.c:294-351 unchanged.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace wrrc {

constexpr int kVecLanes = 16;
constexpr int kVecCand = 4;
constexpr uint32_t kVecTopMark = 0xffffu;  // VecEncBlock::packed: the count field of the largest symbol present (counts are <= 60000)

// One full block (60000 symbols) of up to 16 streams.  Lane state in, lane state out.
struct VecBlock {
    uint32_t active;                 // lane mask
    uint32_t low[kVecLanes], range[kVecLanes];
    const uint8_t* ptr[kVecLanes];   // next unread stream byte; ptr[-1] is the byte held back (rangecod.c:297-299)
    uint8_t* dst[kVecLanes];         // 60000 symbols each
    // the (up to) four most probable symbols of the lane's block, SORTED BY INTERVAL START, a lane with fewer repeating
    // its last one: interval start, width, "is the largest symbol present" (its interval is open-ended,
    // rangecod.c:345-348)
    uint32_t lt[kVecCand][kVecLanes], sy[kVecCand][kVecLanes], is_top[kVecCand][kVecLanes], sym[kVecCand][kVecLanes];
    const void* model[kVecLanes];    // handed to `other` for a symbol outside the candidates
};
// symbol outside the candidates (rare): the scalar look-up path on that lane's model
typedef uint32_t (*VecOther)(const void* model, uint32_t* low, uint32_t* range, uint32_t help);

// One full block (60000 symbols) of up to 16 ENCODER streams.  The encoder's dependency chain is only renormalise ->
// range / 60000 -> new range (rangecod.c:182-229).  The symbol's {lt, sy} come from four compares against the lane's
// most probable symbols while every lane's block is held by at most four symbols (all but a per cent or less; a symbol
// outside them takes a scalar table look-up on that lane), or (`gather`) from a scalar load per lane out of the lanes' packed
// tables, returned to a vector by inserts: planes of any statistics (AMD's gathers are microcoded and slower).  The last byte shifted out and the 0xff bytes behind it are held back per lane until the
// next byte decides about a carry (rangecod.c:182-207); final bytes leave in packed 4-byte stores.
struct VecEncBlock {
    uint32_t active;                 // lane mask
    uint32_t failed;                 // out: lanes retired inside the block -- their table counted a symbol zero times (the lane's
                                     // range went to zero): nothing of them is handed back, at most a byte per symbol was written
    uint32_t low[kVecLanes], range[kVecLanes];
    const uint8_t* sym[kVecLanes];   // 60000 symbols each
    uint8_t* out[kVecLanes];         // stream buffers
    size_t pos[kVecLanes];           // bytes written so far (>= 1: a carry walks back from out[pos - 1])
    uint32_t cand[kVecCand][kVecLanes], lt[kVecCand][kVecLanes], sy[kVecCand][kVecLanes];  // unused entries: cand = 0x100
    const uint32_t* tab;             // [16 lanes][256 symbols]{lt, sy}: symbols outside the candidates
    uint32_t top[kVecLanes];         // largest symbol present: its interval is open-ended (rangecod.c:227)
    int gather;                      // != 0: {lt, sy} of every symbol looked up per lane (lanes with any statistics); else candidates
    uint32_t* packed;                // gather mode: [16 lanes][256 symbols] lt | sy << 16, sy = kVecTopMark marking the largest symbol
                                     // present (a count of ZERO is a symbol the table does not know: coding one retires the lane);
                                     // filled by the caller for the active lanes; entry 0 of idle lanes is overwritten
};
void vec_encode_block(VecEncBlock* b);

bool vec_available();  // the CPU has AVX-512 F/BW/DQ/VL and WR_NO_AVX512 is not set
void vec_decode_block(VecBlock* b, VecOther other);

}  // namespace wrrc
