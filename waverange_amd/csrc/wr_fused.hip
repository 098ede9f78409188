// wr_fused.hip -- one launch per level: the 3-D CDF-9/7 lifting of a whole level in a single
// pass over HBM (reference src/waveletcdf97_3d/waveletcdf97_3d.c:73-276 forward, :281-466 inverse).
//
// Decomposition (forward): a workgroup of 512 threads owns an xy tile of 128 x 32 samples
// (64 x-pairs x 16 y-pairs; thread = one x-pair x two y-pairs = 8 output points) and marches
// along z over a segment of z-pairs:
//   * the two input planes of z-pair t (tile + halo: 4 samples before, 3 after, in x and y)
//     stream from global memory straight into LDS (global_load_lds, no register staging) while
//     the previous pair is being computed; at the domain edges the halo is filled by
//     whole-sample mirroring, which reproduces the reference's doubled boundary terms bit
//     for bit (c*(v+v) == (2c)*v), so the tile code has no edge branches;
//   * x lifting: every lane recomputes the 4-stage chain of two adjacent x-pairs of a row from 6
//     raw pairs in LDS (no barrier between stages; 14 lifting steps for the two pairs), and
//     writes the [low | high] halves of the row to a second buffer;
//   * y lifting: the same recomputation down the columns, 11 rows -> two y-pairs, result in
//     registers: 8 values (LL, HL, LH, HH of both pairs) per plane per thread;
//   * z lifting: streaming register pipeline (as k_stream), 5 doubles of state per point;
//     a segment that starts mid-volume warms the pipeline up on 2 extra z-pairs;
//   * 16 coalesced stores per thread and step, straight to the final octant positions; the
//     low-pass octant goes to a compact buffer that is the next level's input, so no level
//     reads what another workgroup of the same launch writes (no in-place hazard).
// Halo recomputation is bit-safe: every output is a fixed expression tree of its inputs.
#include "wr_kernels.h"

#include <stdlib.h>
#include <string.h>

#include <mutex>

#pragma clang fp contract(off)

namespace wrk {

#define WR_ALPHA (-0x1.960ce676401a2p+0)
#define WR_BETA (-0x1.b2035c9357a96p-5)
#define WR_GAMMA (0x1.c40ceba5738p-1)
#define WR_DELTA (0x1.c626a904721eep-2)
#define WR_ZETA (0x1.264c795071464p+0)
#define WR_IZETA (0x1.bd5edf975ce17p-1)

#ifdef WR_STAMP
unsigned long long* g_stamp_buf = nullptr;
#endif

namespace {

constexpr int TXP = 64;             // x-pairs per tile
#ifndef WR_FTYP
#define WR_FTYP 16
#endif
constexpr int TYP = WR_FTYP;        // y-pairs per tile
constexpr int RX = TXP + 4;         // raw pairs per staged row: 2 + 64 + 2
// Pitches of the two LDS arrays of the forward kernel.  In the x lifting thread tid takes the pairs 5 (tid % 13) .. of row
// tid / 13: within a row the lanes are 5 chunks (80 bytes) apart.  A ds_read_b128 is served in four groups of 16 lanes on
// 64 banks (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27}, ...: every group holds all 16 residues of the lane number), i.e.
// conflict-free when the 16 lanes of a group fall into the 16 different 16-byte slots of a 256-byte row: 5 k mod 16 does that
// -- as long as the pattern carries on across the rows: 13 lanes x 5 = 65 = 1 (mod 16), so the row pitch must be 1 (mod 16)
// chunks.  69 (= 5) made the lane groups that straddle two rows collide: SQ_LDS_BANK_CONFLICT 2.2e8 cycles per level-0
// launch, 7.2e7 with 73, profiles/r04/y_sq_*, z_*.  Likewise the ten ds_write_b64 of a lane (40 bytes apart, 16 contiguous
// lanes served together on 32 banks): 65 = 1 (mod 16) doubles.
constexpr int RXP = RX + 13;        // staged row pitch in 16-byte chunks (81 = 1 mod 16; the last twelve are never loaded)
constexpr int XLP = 2 * TXP + 1;    // xl row pitch in doubles (129 = 1 mod 16)
constexpr int RROWS = 2 * TYP + 7;  // staged rows: 4 + 32 + 3
constexpr int NCHUNK = RROWS * RXP; // 16-byte chunks per plane (2691)
constexpr int NTHR = 32 * TYP;      // one thread = one x-pair x two y-pairs
constexpr int NWAVE = NTHR / 64;
constexpr int KCH = (NCHUNK + NTHR - 1) / NTHR;  // chunks per thread (6)
// x lifting: a lane takes XPL adjacent x-pairs of one staged row, so that rows x lane groups fill the workgroup in ONE
// round: 39 rows x 13 groups of 5 pairs = 507 tasks on 512 lanes (with four pairs per lane the 39 x 16 tasks took two
// rounds on two of the eight waves while the other six waited at the barrier)
constexpr int XPL = (RROWS * TXP + NTHR - 1) / NTHR;   // x-pairs per lane (5)
constexpr int XG = (TXP + XPL - 1) / XPL;              // lane groups per row (13)
static_assert(RROWS * XG <= NTHR, "x lifting: one round");
static_assert(XG * XPL + 4 <= RX + 1, "x lifting: the last group's chunks are staged ones (RX of the tile and one more)");
constexpr size_t LDS_BYTES = (size_t)2 * NCHUNK * 16 + (size_t)RROWS * XLP * 8;

// blockIdx.x -> (tile column, tile row).  Workgroups are dealt round-robin over the 8 XCDs (b % 8 says which
// share an L2), and the j-th workgroups of all XCDs run at the same time.  Tiles next to each other read the same
// halo lines (x: the two 128-B lines a 68-column row segment sticks out into; y: 4 rows), which cost HBM traffic
// unless the neighbour sits on the same L2.  An XCD therefore gets whole tile ROWS: tiles_y / 8 consecutive rows,
// all columns -- no x halo leaves the XCD, and only the 4 halo rows at the band's two edges are fetched twice
// (1024^3, level 0: 8 x 4 tiles per XCD, fetch 1.06 x the payload; a band 2 tiles wide and 16 tall: 1.26 x,
// measured 10.84 GB for 8.59; one tile column per XCD, the plain order: 1.5 x).
__device__ inline void tile_of_block(int b, int tiles_x, int tiles_y, int& tx, int& ty)
{
#ifndef WR_NO_XCD_BLOCK
    if ((tiles_y & 7) == 0) {
        const int xcd = b & 7, j = b >> 3;  // j-th workgroup of this XCD
        tx = j % tiles_x;
        ty = xcd * (tiles_y >> 3) + j / tiles_x;
        return;
    }
#endif
    tx = b % tiles_x;
    ty = b / tiles_x;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup fence, and
// with global_load_lds in flight the compiler turns that into `s_waitcnt vmcnt(0)` in front of
// the s_barrier: every mid-step barrier would drain the DMA that was issued to run BEHIND it.
// Here the wave only waits for its own LDS reads/writes; the DMA is waited for explicitly at
// the top of the next step (vmcnt(0) + barrier) by every wave that issued it.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// min / max that skip quiet NaNs as fmin / fmax do, as the bare instructions: fmin() / fmax() first canonicalize every operand
// that comes from memory (a v_max_f64 x, x, x each, to quiet a signalling NaN) -- 72 of the 168 min/max instructions of the
// encoder's forward kernel.  A signalling NaN in a field is not something the reference's scan survives either (glibc's
// fmin returns x + y for one); the sign of a zero minimum is settled elsewhere (k_last_zero).
__device__ inline double vmin64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ inline double vmax64(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

__device__ inline int mirror(int v, int n)
{
    if (v < 0) v = -v;
    if (v >= n) v = 2 * (n - 1) - v;
    return v < 0 ? 0 : (v >= n ? n - 1 : v);  // far outside (partial tiles): any valid index
}

// forward lifting of two adjacent pairs from s[-2..3], d[-2..2]  (waveletcdf97_3d.c:112-132):
// 14 lifting steps instead of 2 x 10 for two separately recomputed pairs
__device__ inline void lift_fwd_two(const double s[6], const double d[5], double& lo0, double& hi0, double& lo1,
                                    double& hi1)
{
    const double d1a = d[0] + WR_ALPHA * (s[1] + s[0]);
    const double d1b = d[1] + WR_ALPHA * (s[2] + s[1]);
    const double d1c = d[2] + WR_ALPHA * (s[3] + s[2]);
    const double d1d = d[3] + WR_ALPHA * (s[4] + s[3]);
    const double d1e = d[4] + WR_ALPHA * (s[5] + s[4]);
    const double s1b = s[1] + WR_BETA * (d1b + d1a);
    const double s1c = s[2] + WR_BETA * (d1c + d1b);
    const double s1d = s[3] + WR_BETA * (d1d + d1c);
    const double s1e = s[4] + WR_BETA * (d1e + d1d);
    const double d2b = d1b + WR_GAMMA * (s1c + s1b);
    const double d2c = d1c + WR_GAMMA * (s1d + s1c);
    const double d2d = d1d + WR_GAMMA * (s1e + s1d);
    const double s2c = s1c + WR_DELTA * (d2c + d2b);
    const double s2d = s1d + WR_DELTA * (d2d + d2c);
    lo0 = s2c * WR_ZETA; hi0 = d2c * WR_IZETA;
    lo1 = s2d * WR_ZETA; hi1 = d2d * WR_IZETA;
}

// forward lifting of N adjacent pairs from s[-2..N+1], d[-2..N]: 4 N + 6 lifting steps (N = 5: 5.2 per pair; the same
// expression tree per output as lift_fwd_two and as the reference's line loop: the outputs are bit-identical)
template <int N>
__device__ inline void lift_fwd_n(const double s[N + 4], const double d[N + 3], double lo[N], double hi[N])
{
    double d1[N + 3], s1[N + 3], d2[N + 2];
#pragma unroll
    for (int k = 0; k < N + 3; k++) d1[k] = d[k] + WR_ALPHA * (s[k + 1] + s[k]);
#pragma unroll
    for (int k = 1; k < N + 3; k++) s1[k] = s[k] + WR_BETA * (d1[k] + d1[k - 1]);
#pragma unroll
    for (int k = 1; k < N + 2; k++) d2[k] = d1[k] + WR_GAMMA * (s1[k + 1] + s1[k]);
#pragma unroll
    for (int k = 2; k < N + 2; k++) {
        const double s2 = s1[k] + WR_DELTA * (d2[k] + d2[k - 1]);
        lo[k - 2] = s2 * WR_ZETA;
        hi[k - 2] = d2[k] * WR_IZETA;
    }
}

}  // namespace

// MM_IN / MM_OUT: also reduce min/max of the samples read (the whole level input: level 0 = the field) / of the
// coefficients stored to their final positions (mm_lll: including the low-pass octant, i.e. this is the last
// level), one {in lo, in hi, out lo, out hi} record per wave in mm_partial -- replaces the two stand-alone
// min/max passes of the encoder (wrappers.cpp:244-250, 308-314) at ~3 % more vector instructions.
template <bool MM_IN, bool MM_OUT>
__global__ __launch_bounds__(NTHR, 2) void k_fwd_fused(
    const double* __restrict__ src, size_t s_sy, size_t s_sz,  // level input (x stride 1)
    double* __restrict__ dst, size_t d_sy, size_t d_sz,        // coefficient array (final positions)
    double* __restrict__ low, size_t l_sy, size_t l_sz,        // low-pass octant destination
    int n1, int n2, int n3, int zps, double* __restrict__ mm_partial, int mm_lll
#ifdef WR_STAMP
    , unsigned long long* __restrict__ stamp_out  // diagnostic build: per-workgroup phase cycle sums
#endif
    )
{
#ifdef WR_STAMP
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long tn_ = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_s_waitcnt(0xC07F); ph[i] += tn_ - tprev; tprev = tn_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
    extern __shared__ double2 lds2[];
    double2* raw = lds2;                                          // [2][RROWS][RX]
    double* xl = reinterpret_cast<double*>(lds2 + 2 * NCHUNK);   // [RROWS][XLP]: [low 64 | high 64 | 2 unused]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m1 = n1 >> 1, m2 = n2 >> 1, m3 = n3 >> 1;
    const int tiles_x = (m1 + TXP - 1) / TXP;
    int tile_x, tile_y;
    tile_of_block(blockIdx.x, tiles_x, (m2 + TYP - 1) / TYP, tile_x, tile_y);
    const int px0 = tile_x * TXP, py0 = tile_y * TYP;
    const int z0 = blockIdx.y * zps;
    const int z1 = (z0 + zps < m3) ? z0 + zps : m3;
    const int tb = z0 >= 2 ? z0 - 2 : 0, te = z1 + 1;

    const double mm_nan = __builtin_nan("");
    double in_lo = mm_nan, in_hi = mm_nan, out_lo = mm_nan, out_hi = mm_nan;  // fmin/fmax skip NaNs, as the reference's scan does

    // per-thread source offsets of its staged 16-byte chunks (same for every plane).  Rows are
    // mirrored at the y edges by choosing the source row; a chunk that would need x mirroring
    // loads any valid address and is patched in LDS by the x-lifting wave (see xlift).
    int offa[KCH];
#pragma unroll
    for (int k = 0; k < KCH; k++) {
        const int c = tid + NTHR * k;
        const int row = c / RXP, pr = c - row * RXP;
        const int gy = mirror(2 * py0 - 4 + (row < RROWS ? row : 0), n2);
        int gx = 2 * (px0 - 2 + pr);
        gx = gx < 0 ? 0 : (gx > n1 - 2 ? n1 - 2 : gx);
        // pr > RX: padding of the row pitch, never read.  pr == RX: read by the last lane group of a row (its fifth pair,
        // whose results are dropped): any sample of the input will do
        offa[k] = (c < NCHUNK && pr <= RX) ? (int)(gy * s_sy) + gx : -1;
    }
    // global -> LDS without a register round trip: each lane supplies its own 16-byte source,
    // the wave's 64 chunks land contiguously at a wave-uniform LDS base (global_load_lds_dwordx4)
    auto fetch = [&](int t, int p) {
        const double* pl = src + (size_t)(2 * t + p) * s_sz;
#pragma unroll
        for (int k = 0; k < KCH; k++) {
            if (offa[k] >= 0) {
                double2* l = raw + p * NCHUNK + NTHR * k + (w << 6);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pl + offa[k]),
                                                 (__attribute__((address_space(3))) void*)l, 16, 0, 0);
            }
        }
    };
    const bool left_edge = px0 == 0, right_edge = px0 + TXP >= m1;
    const int iL = m1 - 1 - px0;  // local index of the last x-pair of the domain (right-edge tiles)

    // x lifting of every staged row of plane p -> xl.  Thread tid takes row tid / XG, pairs XPL * (tid % XG) and the
    // following (the last group of a row is one pair short).  First the mirrored x halo of the rows this wave reads is
    // patched (edge tiles only; a row two waves share gets the same values from both), THEN the row is read.
    const int xrow = tid / XG, xg = tid - xrow * XG;
    const bool xact = xrow < RROWS;
    const int wrow0 = (w << 6) / XG;  // first staged row this wave reads
    auto xlift = [&](int p) {
        double2* rp = raw + p * NCHUNK;
        if (left_edge | right_edge) {
            // whole-sample mirror of the x halo, in pair terms: pair -1 = (s[1], d[0]),
            // pair -2 = (s[2], d[1]); pair m = (s[m-1], d[m-2]), pair m+1 = (s[m-2], d[m-3]).
            const int row = wrow0 + (lane >> 1);
            if (lane < 2 * ((63 + XG) / XG + 1) && row < RROWS && row <= ((w << 6) + 63) / XG) {
                double2* e = rp + row * RXP;
                if (left_edge) e[lane & 1] = (lane & 1) ? make_double2(e[3].x, e[2].y) : make_double2(e[4].x, e[3].y);
                if (right_edge) {
                    double2* f = e + iL + 3 + (lane & 1);
                    f[0] = (lane & 1) ? make_double2(f[-3].x, f[-4].y) : make_double2(f[-1].x, f[-2].y);
                }
            }
        }
        if (!xact) return;
        const int jq = xg * XPL;  // first of this lane's x-pairs; its chunks: jq .. jq + XPL + 3 (lane stride 80 bytes: the
                                  // eight lanes a ds_read_b128 serves together hit eight different quarters of the banks)
        const double2* r = rp + xrow * RXP + jq;
        double2 v[XPL + 4];
#pragma unroll
        for (int k = 0; k < XPL + 4; k++) v[k] = r[k];
        if (MM_IN) {  // v[2 .. XPL + 1] are this lane's own pairs (the last group's last one is its neighbour's: still a
                      // sample of the level input, like every halo value): all lanes together cover every sample read
#pragma unroll
            for (int k = 2; k < XPL + 2; k++) {
                in_lo = vmin64(in_lo, vmin64(v[k].x, v[k].y));
                in_hi = vmax64(in_hi, vmax64(v[k].x, v[k].y));
            }
        }
        double sv[XPL + 4], dv[XPL + 3], lo[XPL], hi[XPL];
#pragma unroll
        for (int k = 0; k < XPL + 4; k++) sv[k] = v[k].x;
#pragma unroll
        for (int k = 0; k < XPL + 3; k++) dv[k] = v[k].y;
        lift_fwd_n<XPL>(sv, dv, lo, hi);
        double* o = xl + xrow * XLP + jq;
#pragma unroll
        for (int k = 0; k < XPL; k++)
            if (k < TXP - (XG - 1) * XPL || xg < XG - 1) { o[k] = lo[k]; o[TXP + k] = hi[k]; }
    };
    // y lifting of this thread's two y-pairs (2w, 2w+1) for its two x columns:
    // out[4*yp + {0,1,2,3}] = {LL, HL, LH, HH} of y-pair yp
    auto ylift = [&](double out[8]) {
        const double* c0 = xl + (4 * w) * XLP + lane;
        constexpr int P = XLP;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const double* c = c0 + h * TXP;
            const double s[6] = {c[0], c[2 * P], c[4 * P], c[6 * P], c[8 * P], c[10 * P]};
            const double d[5] = {c[1 * P], c[3 * P], c[5 * P], c[7 * P], c[9 * P]};
            lift_fwd_two(s, d, out[h], out[2 + h], out[4 + h], out[6 + h]);
        }
    };

    // z pipeline state of the 8 points (see k_stream in wr_kernels.hip)
    double sr1[8], dr1[8], p1[8], q1[8], p2[8];
#pragma unroll
    for (int q = 0; q < 8; q++) sr1[q] = dr1[q] = p1[q] = q1[q] = p2[q] = 0.0;
    const int ox = px0 + lane, oy = py0 + 2 * w;
    const bool own_x = ox < m1;
    // element offsets of the points inside a z-plane, all 32-bit on top of wave-uniform bases
    const unsigned pos0 = (unsigned)ox + (unsigned)oy * (unsigned)d_sy;
    const unsigned lpos = (unsigned)ox + (unsigned)oy * (unsigned)l_sy;
    const size_t oct_x = (size_t)m1, oct_y = (size_t)m2 * d_sy;

    if (tb < m3) { fetch(tb, 0); fetch(tb, 1); }
    for (int t = tb; t <= te; t++) {
        double a[8], b[8];
#pragma unroll
        for (int q = 0; q < 8; q++) a[q] = b[q] = 0.0;
        if (t < m3) {  // block-uniform
            const bool more = t + 1 <= te && t + 1 < m3;
            STAMP(7);
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's chunks of planes 2t, 2t+1 have landed
            STAMP(0);
            __syncthreads();                      // ... and everybody else's
            STAMP(1);
#ifdef WR_FWD_MEMONLY
            // DIAGNOSTIC (not a transform): the kernel's loads (LDS-DMA) and stores alone, with its barriers; the x and y lifting
            // and their LDS traffic are gone, the z step works on constants
            lds_barrier();
            if (more) fetch(t + 1, 0);
            lds_barrier();
            lds_barrier();
            if (more) fetch(t + 1, 1);
#pragma unroll
            for (int q = 0; q < 8; q++) { a[q] = 1.0 + q; b[q] = 2.0 + t; }
#else
            xlift(0);
            STAMP(2);
            lds_barrier();
            STAMP(6);
            if (more) fetch(t + 1, 0);  // raw[0] is free again: next even plane streams in behind the compute
            STAMP(3);
            ylift(a);
            STAMP(4);
            lds_barrier();
            STAMP(6);
            xlift(1);
            STAMP(2);
            lds_barrier();
            STAMP(6);
            if (more) fetch(t + 1, 1);
            STAMP(3);
            ylift(b);
            STAMP(4);
#endif
        }
        // ---- z step: (a, b) is z-pair t  (waveletcdf97_3d.c:228-262).  Pair t-1 gets its first half
        // (last1 / first1: it is the last / first pair), pair t-2 its second half and leaves.
        const bool last1 = t - 1 >= m3 - 1, first1 = t - 1 <= 0, last2 = t - 2 >= m3 - 1, first2 = t - 2 <= 0;
        if (last1) {  // the pair after the last one mirrors it
#pragma unroll
            for (int q = 0; q < 8; q++) a[q] = sr1[q];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            // Boundary forms by mirroring (c * (v + v) has the bits of (2 c) * v): one block-uniform select
            // on an operand instead of the reference's second expression (waveletcdf97_3d.c:232,238,245,
            // 251).  Pipeline fill / drain steps compute on zeros or stale values that never reach a store.
            double D1 = 0, S1 = 0;
            if (t >= 1 && t <= m3) {
                D1 = dr1[q] + WR_ALPHA * (a[q] + sr1[q]);
                S1 = sr1[q] + WR_BETA * (D1 + (first1 ? D1 : p1[q]));
            }
            if (t >= 2) {
                const int j = t - 2;
                const double D2 = p1[q] + WR_GAMMA * ((last2 ? q1[q] : S1) + q1[q]);
                const double S2 = q1[q] + WR_DELTA * (D2 + (first2 ? D2 : p2[q]));
                const int yp = q >> 2;  // which of the two y-pairs
                // wave-uniform plane bases + per-lane 32-bit offsets (saddr form).  Trading values
                // between lane pairs for 16-byte stores was measured 3 % SLOWER (per-lane 64-bit
                // addresses, DPP + selects) -- profiles/r01/NOTES.md.
                if (own_x && oy + yp < m2 && j >= z0 && j < z1) {
                    double* base = dst + ((q & 1) ? oct_x : 0) + ((q & 2) ? oct_y : 0) + (size_t)yp * d_sy;
                    const double clo = S2 * WR_ZETA, chi = D2 * WR_IZETA;
                    if ((q & 3) == 0) (low + (size_t)j * l_sz + (size_t)yp * l_sy)[lpos] = clo;
                    else (base + (size_t)j * d_sz)[pos0] = clo;
                    (base + (size_t)(m3 + j) * d_sz)[pos0] = chi;
                    if (MM_OUT) {
                        out_lo = vmin64(out_lo, chi); out_hi = vmax64(out_hi, chi);
                        if ((q & 3) != 0 || mm_lll) { out_lo = vmin64(out_lo, clo); out_hi = vmax64(out_hi, clo); }
                    }
                }
                p2[q] = D2;
            }
            p1[q] = D1; q1[q] = S1; sr1[q] = a[q]; dr1[q] = b[q];
        }
        STAMP(5);
    }
    if (MM_IN || MM_OUT) {
        for (int o = 32; o > 0; o >>= 1) {
            if (MM_IN) { in_lo = fmin(in_lo, __shfl_down(in_lo, o, 64)); in_hi = fmax(in_hi, __shfl_down(in_hi, o, 64)); }
            out_lo = fmin(out_lo, __shfl_down(out_lo, o, 64)); out_hi = fmax(out_hi, __shfl_down(out_hi, o, 64));
        }
        if (lane == 0) {
            double* rec = mm_partial + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NWAVE + w) * 4;
            rec[0] = in_lo; rec[1] = in_hi; rec[2] = out_lo; rec[3] = out_hi;
        }
    }
#ifdef WR_STAMP
    if (stamp_out && lane == 0)
        for (int i = 0; i < 8; i++) stamp_out[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NWAVE + w) * 8 + i] = ph[i];
#endif
}


// =====================================================================================
// Inverse level (waveletcdf97_3d.c:281-466): z, then y, then x.
//   * the coefficient planes of z-pair t (low-z plane t, high-z plane q3+t; per plane the four
//     xy quadrants LL/HL/LH/HH of the tile + 2 halo pairs on every side, i.e. 4 x 20 x 68
//     doubles) go from global memory straight into registers; halo ROWS are mirrored by choosing
//     the source row;
//   * z: pointwise streaming register pipeline on every staged point (tile + halo), 4 doubles
//     of state per point; both output planes of the pair go to LDS;
//   * y: each wave rebuilds the 4 output rows of its two y-pairs for all 136 coefficient
//     columns by recomputation from 11 rows (wave-private LDS rows, no workgroup barrier);
//   * x: mirrored halo COLUMNS are patched in those rows (x mirroring commutes with the
//     pointwise z and y steps), then every lane rebuilds FOUR adjacent x-pairs of one of the four
//     rows (22 lifting steps where two lanes with two pairs each need 28) and stores 4 x 16 B.
// The scaling that opens every 1-D inverse pass (s * 1/zeta, d * zeta, waveletcdf97_3d.c:312-313) is applied where a
// value is PRODUCED for the next pass -- by the z step for the y pass, by the y pass for the x pass -- instead of where
// it is consumed: the same single multiplication of the same operands, so the same bits, but once per value and not
// once per lane that re-reads it (a value is read by 5.5 lanes of the next pass on average).
// =====================================================================================
namespace {
#ifndef WR_ITYP
#define WR_ITYP 16
#endif
// Tile geometry of the inverse kernel: ITXP x ITYP coefficient pairs (2 ITXP x 2 ITYP output samples per z-plane), a wave
// owns YPW y-pairs (RW = 2 YPW output rows).  64 x 16 with 2 y-pairs per wave is ONE workgroup of 512 threads per CU (125 KB
// of LDS): its eight waves go through the phases of a step together; 32 x 16 with 4 y-pairs per wave is TWO workgroups of 256
// threads per CU (2 x 67 KB), which drift apart -- one computes while the other waits at its barrier or for LDS -- at the
// price of 6 % more halo (36 / 32 against 68 / 64 columns).  Same registers per thread either way (six chunk slots).
#ifndef WR_ITXP
#define WR_ITXP 64
#endif
// DIAGNOSTIC (results are NOT a transform): leaves out one class of the inverse kernel's LDS instructions, to attribute the
// SQ_LDS_* counters.  1: the y stage's pass over columns 64..127, 2: its leftover columns, 4: the x stage's reads, 8: the z
// step's writes, 16: the y stage's writes, 32: the y stage's pass over columns 0..63.
#ifndef WR_INV_DIAG_SKIP
#define WR_INV_DIAG_SKIP 0
#endif
#ifndef WR_YPW
#define WR_YPW 2
#endif
constexpr int ITXP = WR_ITXP;            // x-pairs per tile
constexpr int ITYP = WR_ITYP;            // y-pairs per tile
constexpr int YPW = WR_YPW;              // y-pairs per wave
constexpr int RW = 2 * YPW;              // output rows per wave and z-plane
constexpr int YG = YPW / 2;              // groups of two y-pairs per wave (lift_inv_two rebuilds two pairs = four rows)
constexpr int INWAVE = ITYP / YPW;
constexpr int INTHR = 64 * INWAVE;
constexpr int XCG = ITXP / 4;            // lane column groups of the x stage (four x-pairs per lane)
static_assert(YPW % 2 == 0 && ITYP % YPW == 0 && ITXP % 4 == 0 && RW * XCG == 64 && 64 % ITXP == 0, "x stage: one lane per (row, four pairs)");
constexpr int HX = ITXP + 4;             // coefficient columns per quadrant row (68)
constexpr int HY = ITYP + 4;             // coefficient rows per quadrant (20)
constexpr int CROW = HX / 2;             // 16-byte chunks per row (34)
[[maybe_unused]] constexpr int NCI = 4 * HY * CROW;       // chunks per plane without padding (2720)
// A thread's chunk slots: the first KH cover the y-low quadrants (LL, HL), the last KH the y-high ones (LH, HH), so
// that the scale a slot's values leave the z step with is a compile-time constant
constexpr int NCH = 2 * HY * CROW;       // chunks per y-half (1360)
constexpr int KH = (NCH + INTHR - 1) / INTHR;   // chunk slots per thread and half (3)
constexpr int KCI = 2 * KH;              // chunk slots per thread (6)
// Wave-private rows between the y and the x stage: 4 rows of [xlow 68 | xhigh 68].  The x stage's ds_read_b128 are served in
// groups of 16 lanes (lane = 4 * lane column + row: a group holds the four rows of four lane columns, {0, 3, 5, 6} or
// {1, 2, 4, 7} (+ 8)), conflict-free when their 16-byte pieces fall into the 16 slots of a 256-byte bank row: the columns
// give {0, 6, 10, 12} resp. {2, 4, 8, 14} slots, which the row offsets {0, 8, 1, 9} tile -- rows 9 x 128 bytes apart, the
// upper two shifted by one piece.  (The write-back of the results, ds_write_b128 in groups of 8 lanes on 32 banks, stays
// two-way with any offsets that serve the reads: 16 against the 13 cycles the store takes anyway.)
// 32-pair tiles: eight rows of [xlow 36 | xhigh 36], lane = 8 * lane column + row.  By the same rules a ds_read_b128 group
// holds rows 0-3 of lane columns {0, 3} and rows 4-7 of lane columns {1, 2} (or the other way round): with X, Y the slots
// (16-byte pieces mod 16) rows 0-3 resp. 4-7 start in, X, Y + 2, Y + 4, X + 6 must tile the 16 slots, and so must Y, X + 2,
// X + 4, Y + 6: X = Y = {0, 1, 8, 9} does -- row starts at 44 r + 5 (r & 1) pieces.
#if WR_ITXP == 64
constexpr int YPITCH = 144;              // doubles between rows
constexpr int YWAVE = 4 * YPITCH + 16;   // doubles per wave
__device__ inline int yrow_off(int r) { return r * YPITCH + (r >> 1) * 2; }
#else
static_assert(WR_ITXP == 32 && WR_YPW == 4, "row offsets of the wave-private rows are worked out for 64 x 16 / 2 and 32 x 16 / 4");
constexpr int YPITCH = 88;
constexpr int YWAVE = 8 * YPITCH;        // (7 x 88 + 10 + 72 = 698 doubles used)
__device__ inline int yrow_off(int r) { return r * YPITCH + (r & 1) * 10; }
#endif
// The z buffer's strides, padded for the y stage's column reads (8 bytes per lane, consecutive columns; served 16 or 32 lanes
// at a time by the compiler's choice of ds_read2_b64 / ds_read_b64, a lane's bank = its double's index mod 16 resp. 32):
//  * the pass over columns 64..127 holds the last four x-low columns and the first sixty x-high ones: conflict-free when the
//    quadrants lie 4 doubles apart mod 32 (unpadded: 1360 = 16 mod 32, lanes 0-3 against 20-23; mod 16: against 4-7);
//  * the leftover pass holds the same eight columns of the even and of the odd plane: conflict-free when the planes lie 8
//    doubles apart mod 16 (unpadded: 0).
// Both cost one extra LDS cycle on every one of the pass's eleven reads: 33 cycles per wave and z-pair, 40 % of what was
// left of the kernel's bank-conflict cycles (profiles/r05/s_lds_conflicts_by_instruction_class.txt).  A quadrant begins on
// a multiple of eight chunks, so the z step's ds_write_b128 (eight consecutive lanes, eight consecutive chunks) never straddles one.
#if WR_ITXP == 64
constexpr int QPAD = 20;                 // doubles between two quadrants of a plane
constexpr int PPAD = 8;                  // doubles between the two planes of a step
#else
constexpr int QPAD = 0, PPAD = 0;
#endif
constexpr int QS = HY * HX + QPAD;       // quadrant stride, doubles
constexpr int PSC = (4 * QS + PPAD) / 2; // plane stride, 16-byte chunks
static_assert(QPAD % 2 == 0 && PPAD % 2 == 0 && (HY * CROW) % 8 == 0, "chunks stay 16-byte aligned; quadrants begin on a multiple of eight chunks");
static_assert(WR_ITXP != 64 || (QS % 32 == 4 && (2 * PSC) % 16 == 8), "strides of the z buffer against the y stage's reads");
constexpr size_t LDS_INV = (size_t)2 * PSC * 16 + (size_t)INWAVE * YWAVE * 8;

// A double of another lane of the quad (lanes 4q .. 4q + 3), by DPP: `v_mov_b32 ... quad_perm:[..]` on either half.
// 0xB1 = quad_perm:[1,0,3,2] (the lane one over), 0x4E = quad_perm:[2,3,0,1] (the lane two over).
template <int CTRL>
__device__ inline double quad_lane(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// One stage of the 4 x 4 transpose of {a, b} held by the lanes of a quad: the lanes with `odd` set hand their `a` to their
// partner and take its `b`, the others the other way round.
template <int CTRL>
__device__ inline void quad_exchange(bool odd, double& a, double& b)
{
    const double got = quad_lane<CTRL>(odd ? a : b);
    a = odd ? got : a;
    b = odd ? b : got;
}

// whole-sample symmetric extension in coefficient space (even length 2M):
//   low-pass  s[-k] = s[k],    s[M-1+k] = s[M-k]
//   high-pass d[-k] = d[k-1],  d[M-1+k] = d[M-1-k]
__device__ inline int mirror_s(int k, int M) { if (k < 0) k = -k; if (k >= M) k = 2 * M - 1 - k; return k < 0 ? 0 : (k >= M ? M - 1 : k); }
__device__ inline int mirror_d(int k, int M) { if (k < 0) k = -k - 1; if (k >= M) k = 2 * M - 2 - k; return k < 0 ? 0 : (k >= M ? M - 1 : k); }

// inverse lifting of two adjacent pairs from s[-1..3], d[-2..3], both already scaled (s * 1/zeta, d * zeta)  (:314-337):
// 14 lifting steps
__device__ inline void lift_inv_two(const double s[5], const double d[6], double out[4])
{
    double s1[5], d1[4], s2[3];
#pragma unroll
    for (int k = 0; k < 5; k++) s1[k] = s[k] - WR_DELTA * (d[k + 1] + d[k]);
#pragma unroll
    for (int k = 0; k < 4; k++) d1[k] = d[k + 1] - WR_GAMMA * (s1[k + 1] + s1[k]);
#pragma unroll
    for (int k = 0; k < 3; k++) s2[k] = s1[k + 1] - WR_BETA * (d1[k + 1] + d1[k]);
    out[0] = s2[0];
    out[1] = d1[1] - WR_ALPHA * (s2[1] + s2[0]);
    out[2] = s2[1];
    out[3] = d1[2] - WR_ALPHA * (s2[2] + s2[1]);
}

// inverse lifting of four adjacent pairs from s[-1..5], d[-2..5] (scaled): 22 lifting steps, 5.5 per pair; every output
// is the expression tree of lift_inv_two and of the reference's line loop
__device__ inline void lift_inv_four(const double s[7], const double d[8], double ev[4], double od[4])
{
    double s1[7], d1[6], s2[5];
#pragma unroll
    for (int k = 0; k < 7; k++) s1[k] = s[k] - WR_DELTA * (d[k + 1] + d[k]);
#pragma unroll
    for (int k = 0; k < 6; k++) d1[k] = d[k + 1] - WR_GAMMA * (s1[k + 1] + s1[k]);
#pragma unroll
    for (int k = 0; k < 5; k++) s2[k] = s1[k + 1] - WR_BETA * (d1[k + 1] + d1[k]);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ev[k] = s2[k];
        od[k] = d1[k + 1] - WR_ALPHA * (s2[k + 1] + s2[k]);
    }
}
}  // namespace

// (A form of this kernel that dequantized the finest level's detail octants from the bit planes on the way -- no accumulate
// pass over the whole array, 16 GB of traffic less per decode -- was built in round 4, bit-exact, and measured EQUAL: 6.6 ms
// against 2.5 + 4.2 = 6.7 ms, its ~480 extra vector instructions per thread and z step sitting in the z phase of a kernel
// whose eight waves per CU move through their phases together.  Removed in round 5: profiles/r04/j_inverse_from_planes.txt.)
__global__ __launch_bounds__(INTHR, 2) void k_inv_fused(
    const double* __restrict__ src, size_t s_sy, size_t s_sz,  // coefficient array (detail octants)
    const double* __restrict__ low, size_t l_sy, size_t l_sz,  // low-pass octant (previous level's output)
    double* __restrict__ out, size_t o_sy, size_t o_sz,        // reconstructed box of this level
    int n1, int n2, int n3, int zps
#ifdef WR_STAMP
    , unsigned long long* __restrict__ stamp_out
#endif
    )
{
#ifdef WR_STAMP
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#endif
#if defined(WR_INV_MEMONLY) && defined(WR_INV_MEMPITCH)
    // DIAGNOSTIC: the detail octants read with a row pitch that is not a power of two (what a padded coefficient array
    // would give); planes wrap so that the padded strides stay inside the array
    const int pad_planes = (int)(((size_t)n1 * n2 * n3) / ((size_t)WR_INV_MEMPITCH * n2)) - 1;
    s_sy = WR_INV_MEMPITCH; s_sz = (size_t)WR_INV_MEMPITCH * n2;
#endif
    extern __shared__ double2 lds2[];
    double2* zb = lds2;                   // [2][PSC]   the two z-reconstructed planes of a step: four quadrants, QS doubles apart
    double* yb = reinterpret_cast<double*>(lds2 + 2 * PSC);  // [INWAVE][YWAVE] wave-private rows
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m1 = n1 >> 1, m2 = n2 >> 1, m3 = n3 >> 1;
    const int tiles_x = (m1 + ITXP - 1) / ITXP;
    int tile_x, tile_y;
    tile_of_block(blockIdx.x, tiles_x, (m2 + ITYP - 1) / ITYP, tile_x, tile_y);
    const int px0 = tile_x * ITXP, py0 = tile_y * ITYP;
    const int z0 = blockIdx.y * zps;
    const int z1 = (z0 + zps < m3) ? z0 + zps : m3;
    const int tb = z0 >= 2 ? z0 - 2 : 0, te = z1 + 1;

    // source offsets of this thread's chunk slots inside a z-plane; zi: where the slot's chunk lives in a zb plane
    int offL[KCI], offH[KCI], zi[KCI];
    unsigned lll_mask = 0, valid_mask = 0;
#pragma unroll
    for (int k = 0; k < KCI; k++) {
        const int half = k / KH;                       // 0: y-low quadrants, 1: y-high quadrants
        const int ch = tid + INTHR * (k - half * KH);  // chunk inside the half
        if (ch < NCH) valid_mask |= 1u << k;
        // (Plain row order: a quadrant row is 544 bytes that start 16 bytes before a cache line.  Taking the 512-byte cores of
        // all rows first -- a wave's load = two whole rows = eight whole lines -- and the two halo chunks of every row
        // afterwards was measured 5 % SLOWER, profiles/r05/c_ab_inverse_aligned_loads_same_box.txt.)
        const int c = half * NCH + (ch < NCH ? ch : 0);
        const int q = c / (HY * CROW), rem = c - q * (HY * CROW);
        zi[k] = c + q * (QPAD / 2);
        const int row = rem / CROW, cc = rem - row * CROW;
        const int gyp = (q & 2) ? mirror_d(py0 - 2 + row, m2) : mirror_s(py0 - 2 + row, m2);
        int gxp = px0 - 2 + 2 * cc;
        gxp = gxp < 0 ? 0 : (gxp > m1 - 2 ? m1 - 2 : gxp);  // x-mirrored columns are patched later
        const int so = (int)(((q & 2) ? m2 + gyp : gyp) * s_sy) + ((q & 1) ? m1 : 0) + gxp;
        offH[k] = so;
        offL[k] = (q == 0) ? (int)(gyp * l_sy) + gxp : so;
        if (q == 0) lll_mask |= 1u << k;
#if defined(WR_INV_MEMONLY) && defined(WR_INV_MEMPAT)
        // DIAGNOSTIC: the same number of loads, but every wave's load instruction takes 1 KB of consecutive, aligned bytes from
        // a 32 KB region of the plane that belongs to this tile alone (what a tile-blocked coefficient layout with the halo
        // stored twice would give): the price of the row-segment shape of the real pattern is the difference
        {
            const int tile_lin = blockIdx.x % ((int)((size_t)n1 * n2 / 4096) > 0 ? (int)((size_t)n1 * n2 / 4096) : 1);
            const int lin = (c % (NCI / 2 < 2048 ? NCI / 2 : 2048)) * 2;
            offH[k] = tile_lin * 4096 + lin;
            offL[k] = offH[k];
            lll_mask = 0;
        }
#endif
    }
    // the coefficient chunks of z-pair t go straight into registers: the z step is the only reader,
    // it runs first in a step, and the registers are free again for the next pair right after it.
    double2 rl[KCI], rh[KCI];
    auto fetch = [&](int t) {
        const double* pll = low + (size_t)t * l_sz;
#if defined(WR_INV_MEMONLY) && defined(WR_INV_MEMPITCH)
        const double* pl = src + (size_t)(t % pad_planes) * s_sz;
        const double* ph = src + (size_t)((m3 + t) % pad_planes) * s_sz;
#else
        const double* pl = src + (size_t)t * s_sz;
        const double* ph = src + (size_t)(m3 + t) * s_sz;
#endif
        // (non-temporal loads: 5.10 against 4.15 ms -- the halo lines a neighbour tile fetches again then come from HBM; all
        // low-z chunks before all high-z ones, non-temporal stores: within 1 %: profiles/r05/c_ab_inverse_nontemporal_*.txt)
#pragma unroll
        for (int k = 0; k < KCI; k++) {
            if ((valid_mask >> k) & 1) {
                rl[k] = *reinterpret_cast<const double2*>((((lll_mask >> k) & 1) ? pll : pl) + offL[k]);
                rh[k] = *reinterpret_cast<const double2*>(ph + offH[k]);
            }
        }
    };
    const bool left_edge = px0 == 0, right_edge = px0 + ITXP >= m1;
    const int iL = m1 - 1 - px0;
    double* ybw = yb + w * YWAVE;  // this wave's RW rows of [xlow HX | xhigh HX] (yrow_off)
    const int J = YPW * w;         // first of this wave's local y-pairs

    // x stage on the wave's RW staged rows (scaled by the y stage): patch mirrored halo columns, rebuild, store
    auto xstage = [&](int zplane, int yrow0) {
        if ((left_edge | right_edge) && lane < 2 * RW) {
            double* r = ybw + yrow_off(lane >> 1);
            if (left_edge) {
                if (lane & 1) { r[0] = r[4]; r[1] = r[3]; }              // s[-2] = s[2], s[-1] = s[1]
                else { r[HX + 0] = r[HX + 3]; r[HX + 1] = r[HX + 2]; }   // d[-2] = d[1], d[-1] = d[0]
            }
            if (right_edge) {
                if (lane & 1) { r[iL + 3] = r[iL + 2]; r[iL + 4] = r[iL + 1]; }            // s[m] = s[m-1], s[m+1] = s[m-2]
                else { r[HX + iL + 3] = r[HX + iL + 1]; r[HX + iL + 4] = r[HX + iL]; }     // d[m] = d[m-2], d[m+1] = d[m-3]
            }
        }
        // four adjacent x-pairs per lane: lane = RW * (lane column) + row
        const int r = lane % RW, i = (lane / RW) * 4;
        const double* row = ybw + yrow_off(r);
        // local column of pair k is k + 2:  s[i-1..i+5] -> i+1..i+7,  d[i-2..i+5] -> i..i+7
        const double2* ps = reinterpret_cast<const double2*>(row + i);
        const double2* pd = reinterpret_cast<const double2*>(row + HX + i);
        double2 s0, s1, s2, s3, d0, d1, d2, d3;
        if (!(WR_INV_DIAG_SKIP & 4)) {
            s0 = ps[0]; s1 = ps[1]; s2 = ps[2]; s3 = ps[3]; d0 = pd[0]; d1 = pd[1]; d2 = pd[2]; d3 = pd[3];
        }
        else s0 = s1 = s2 = s3 = d0 = d1 = d2 = d3 = make_double2((double)lane, (double)zplane);
        const double sv[7] = {s0.y, s1.x, s1.y, s2.x, s2.y, s3.x, s3.y};
        const double dv[8] = {d0.x, d0.y, d1.x, d1.y, d2.x, d2.y, d3.x, d3.y};
        double ev[4], od[4];
        lift_inv_four(sv, dv, ev, od);
        // s0.x is no operand of the lifting, and without a use the compiler narrows the four 16-byte reads of the s side to
        // ds_read2_b64 / ds_read_b64 of the seven doubles it needs -- which are served 16 / 32 consecutive lanes at a time on
        // 32 / 64 banks, where the rows' offsets (worked out for ds_read_b128's lane groups) put rows 0 and 1, 2 and 3 on the
        // same banks: 26 extra LDS cycles per plane, 60 % of what was left of the kernel's conflict cycles
        // (profiles/r05/s_lds_conflicts_by_instruction_class.txt).  The use sits behind the lifting so that nothing waits for it.
        asm volatile("" :: "v"(s0.x));
        // The lane's eight results are 64 consecutive bytes of its row: stored as they are, every store instruction would
        // write 16 bytes out of every 64.  They leave row by row instead, one x-pair per lane: 1 KB of consecutive bytes per
        // instruction.
        // (a store instruction takes 64 / ITXP whole rows of the tile: 1 KB, or two pieces of 512 bytes)
        constexpr int RPS = 64 / ITXP;
        const int srow = lane / ITXP, spair = lane % ITXP;
        const bool own = px0 + spair < m1;
        double* dstp = out + (size_t)zplane * o_sz + (size_t)(yrow0 + srow) * o_sy + 2 * (px0 + spair);
#if WR_ITXP == 64 && !defined(WR_INV_LDS_STORE)
        // 64-pair tiles: a quad of lanes IS the four rows of one lane column (lane = 4 * column + row), and lane 4 c + p of the
        // row-wise store wants pair 4 c + p of every row: a 4 x 4 transpose of {ev, od} pairs inside the quad -- two DPP
        // exchanges (the lane one over, then the lane two over), no LDS.  (Through the wave's LDS rows, as the 32-pair tiles
        // still do it, the write-back is two-way bank-conflicted by construction: it was all that was left of the kernel's
        // conflict cycles, 1.6e8 per level-0 launch, profiles/r04/y_sq_*; -DWR_INV_LDS_STORE rebuilds that form.)
        {
            const bool b0 = lane & 1, b1 = lane & 2;
            quad_exchange<0xB1>(b0, ev[0], ev[1]); quad_exchange<0xB1>(b0, od[0], od[1]);
            quad_exchange<0xB1>(b0, ev[2], ev[3]); quad_exchange<0xB1>(b0, od[2], od[3]);
            quad_exchange<0x4E>(b1, ev[0], ev[2]); quad_exchange<0x4E>(b1, od[0], od[2]);
            quad_exchange<0x4E>(b1, ev[1], ev[3]); quad_exchange<0x4E>(b1, od[1], od[3]);
        }
        static_assert(WR_ITXP != 64 || (RW == 4 && RPS == 1), "a quad of lanes holds the four rows of a lane column");
#pragma unroll
        for (int st = 0; st < RW / RPS; st++)
            if (own && yrow0 + st < n2) *reinterpret_cast<double2*>(dstp + (size_t)st * o_sy) = make_double2(ev[st & 3], od[st & 3]);
#else
        // through the wave's rows (everything has been read: LDS runs a wave's instructions in order)
        double* o = ybw + yrow_off(r) + 2 * i;
#pragma unroll
        for (int k = 0; k < 4; k++) *reinterpret_cast<double2*>(o + 2 * k) = make_double2(ev[k], od[k]);
#pragma unroll
        for (int st = 0; st < RW / RPS; st++) {
            const int rr = st * RPS + srow;
            const double2 v = *reinterpret_cast<const double2*>(ybw + yrow_off(rr) + 2 * spair);
            if (own && yrow0 + rr < n2) *reinterpret_cast<double2*>(dstp + (size_t)(st * RPS) * o_sy) = v;
        }
#endif
    };
    // y pass of one coefficient column cid of the z-plane in zbuf (scaled by the z step): the wave's four output rows,
    // scaled for the x pass
    auto ycolumn = [&](const double2* zbuf, int cid, int J, double (&o)[4]) {  // J: first of the two local y-pairs rebuilt
        const int xh = cid >= HX;             // 0: x-low column, 1: x-high column
        const int col = cid - xh * HX;
        const double* zl = reinterpret_cast<const double*>(zbuf) + (size_t)(xh) * QS + col;        // y-low quadrant
        const double* zh = reinterpret_cast<const double*>(zbuf) + (size_t)(2 + xh) * QS + col;    // y-high quadrant
        // local row of y-pair k is k + 2:  s[J-1..J+3] -> rows J+1..J+5,  d[J-2..J+3] -> rows J..J+5
        const double sr[5] = {zl[(J + 1) * HX], zl[(J + 2) * HX], zl[(J + 3) * HX], zl[(J + 4) * HX], zl[(J + 5) * HX]};
        const double dr[6] = {zh[(J + 0) * HX], zh[(J + 1) * HX], zh[(J + 2) * HX], zh[(J + 3) * HX], zh[(J + 4) * HX],
                              zh[(J + 5) * HX]};
        lift_inv_two(sr, dr, o);
        const double sc = xh ? WR_ZETA : WR_IZETA;  // the x pass's opening scale (x-low columns are its s, x-high its d)
#pragma unroll
        for (int r = 0; r < 4; r++) o[r] *= sc;
    };
    // The 2 HX columns (136) are NP full passes of the wave (two) and YT = 8 columns more: those of BOTH planes of the z-pair
    // (and of all the wave's groups of y-pairs) share one pass -- lanes 0..7 the even plane, 8..15 the odd one: five passes
    // per z-pair and group instead of six.
    constexpr int NP = (2 * HX) / 64, YT = 2 * HX - 64 * NP;
    static_assert(YT > 0 && (YT & (YT - 1)) == 0 && 2 * YG * YT <= 64, "y stage: the leftover columns of two planes fit one pass");
    const int tail_plane = lane / (YG * YT), tail_g = (lane / YT) % YG;  // (lanes below 2 YG YT)
    // y + x stages of the z-plane held in zbuf; tail: this lane's leftover column (mine: of this plane)
    auto yxstage = [&](int zplane, const double2* zbuf, const double (&tail)[4], bool mine) {
#pragma unroll
        for (int g = 0; g < YG; g++) {
#pragma unroll
            for (int p = 0; p < NP; p++) {
                double o[4] = {1.0, 2.0, 3.0, 4.0};
                if (!((WR_INV_DIAG_SKIP & 1) && p == 1) && !((WR_INV_DIAG_SKIP & 32) && p == 0)) ycolumn(zbuf, lane + 64 * p, J + 2 * g, o);
                if (!(WR_INV_DIAG_SKIP & 16)) {
#pragma unroll
                    for (int r = 0; r < 4; r++) ybw[yrow_off(4 * g + r) + lane + 64 * p] = o[r];
                } else asm volatile("" :: "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));
            }
        }
        if (mine && !(WR_INV_DIAG_SKIP & 2)) {
#pragma unroll
            for (int r = 0; r < 4; r++) ybw[yrow_off(4 * tail_g + r) + 64 * NP + (lane & (YT - 1))] = tail[r];
        }
        xstage(zplane, 2 * (py0 + J));
    };

    // z pipeline state per staged point: 2 points per chunk slot
    double dprev[KCI][2], s1prev[KCI][2], d1prev[KCI][2], s2prev[KCI][2];
#pragma unroll
    for (int k = 0; k < KCI; k++)
#pragma unroll
        for (int e = 0; e < 2; e++) dprev[k][e] = s1prev[k][e] = d1prev[k][e] = s2prev[k][e] = 0.0;

    if (tb < m3) fetch(tb);
    for (int t = tb; t <= te; t++) {
        const int j = t - 2;
        const bool emit = j >= z0 && j < z1;  // block-uniform
        // pair t enters (first: it has no left neighbour), pair t-1 gets its second half (last1 / first1:
        // it is the last / the first pair), pair t-2 leaves (last2: it is the last pair)
        const bool first = t == 0, last1 = t - 1 >= m3 - 1, first1 = t - 1 <= 0, last2 = j >= m3 - 1;
        STAMP(7);
#ifdef WR_INV_MEMONLY
        // DIAGNOSTIC (not a transform): the kernel's loads and stores alone -- same addresses, same instructions, same order,
        // WR_INV_MEMONLY=2 also the two barriers -- with one addition per loaded value in place of the three lifting passes
        // and no LDS traffic: what the access pattern costs by itself.
        {
            if (WR_INV_MEMONLY == 2) lds_barrier();
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < KCI; k++)
                if ((valid_mask >> k) & 1) acc += (rl[k].x + rl[k].y) + (rh[k].x + rh[k].y);
            if (t + 1 <= te && t + 1 < m3) fetch(t + 1);
            if (WR_INV_MEMONLY == 2) lds_barrier();
            if (emit) {
                constexpr int RPS = 64 / ITXP;
                const int srow = lane / ITXP, spair = lane % ITXP;
                const bool own = px0 + spair < m1;
#pragma unroll
                for (int pl2 = 0; pl2 < 2; pl2++) {
                    const int yrow0 = 2 * (py0 + J);
                    double* dstp = out + (size_t)(2 * j + pl2) * o_sz + (size_t)(yrow0 + srow) * o_sy + 2 * (px0 + spair);
#pragma unroll
                    for (int st = 0; st < RW / RPS; st++) {
                        const int rr = st * RPS + srow;
                        if (own && yrow0 + rr < n2) *reinterpret_cast<double2*>(dstp + (size_t)(st * RPS) * o_sy) = make_double2(acc, acc + pl2);
                    }
                }
            }
            continue;
        }
#endif
        lds_barrier();  // the previous step's readers of zb are done
        STAMP(1);
        // ---- z step on every staged point  (waveletcdf97_3d.c:312-337 along z)
#pragma unroll
        for (int k = 0; k < KCI; k++) {
            if ((valid_mask >> k) & 1) {
                // Boundary forms by mirroring: c * (v + v) has the bits of (2 c) * v, so "the missing neighbour
                // is the other one" replaces the reference's second expression (waveletcdf97_3d.c:316,323,
                // 329,335) with one block-uniform select on an operand.  Steps outside [0, m3) of the
                // pipeline's fill and drain compute on stale operands; nothing they produce reaches an
                // emitted plane (the selects below cut exactly those dependencies).
                const double lo[2] = {rl[k].x, rl[k].y}, hi[2] = {rh[k].x, rh[k].y};
                // the y pass's opening scale: y-low quadrants are its s (* 1/zeta), y-high quadrants its d (* zeta)
                const double sc = (k < KH) ? WR_IZETA : WR_ZETA;
                double ev[2], od[2];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const double s0 = lo[e] * WR_IZETA;
                    const double d0 = hi[e] * WR_ZETA;
                    const double S1 = s0 - WR_DELTA * (d0 + (first ? d0 : dprev[k][e]));
                    const double D1 = dprev[k][e] - WR_GAMMA * ((last1 ? s1prev[k][e] : S1) + s1prev[k][e]);
                    const double S2 = s1prev[k][e] - WR_BETA * (D1 + (first1 ? D1 : d1prev[k][e]));
                    ev[e] = s2prev[k][e] * sc;
                    od[e] = (d1prev[k][e] - WR_ALPHA * ((last2 ? s2prev[k][e] : S2) + s2prev[k][e])) * sc;
                    dprev[k][e] = d0; s1prev[k][e] = S1; d1prev[k][e] = D1; s2prev[k][e] = S2;
                }
                if (emit && !(WR_INV_DIAG_SKIP & 8)) { zb[zi[k]] = make_double2(ev[0], ev[1]); zb[PSC + zi[k]] = make_double2(od[0], od[1]); }
                else if (WR_INV_DIAG_SKIP & 8) asm volatile("" :: "v"(ev[0]), "v"(ev[1]), "v"(od[0]), "v"(od[1]));
            }
        }
        STAMP(2);
        if (t + 1 <= te && t + 1 < m3) fetch(t + 1);  // in flight behind the y/x stages
        STAMP(3);
        lds_barrier();  // both planes complete
        STAMP(6);
        if (emit) {
            double tail[4] = {0.0, 0.0, 0.0, 0.0};
            if (lane < 2 * YG * YT && !(WR_INV_DIAG_SKIP & 2)) ycolumn(zb + (tail_plane ? PSC : 0), 64 * NP + (lane & (YT - 1)), J + 2 * tail_g, tail);
            yxstage(2 * j, zb, tail, lane < YG * YT);
            yxstage(2 * j + 1, zb + PSC, tail, lane >= YG * YT && lane < 2 * YG * YT);
            STAMP(4);
        }
    }
#ifdef WR_STAMP
    if (stamp_out && lane == 0)
        for (int i = 0; i < 8; i++) stamp_out[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * INWAVE + w) * 8 + i] = ph[i];
#endif
}

// Number of leading levels (finest first) the fused kernels can take; the remaining, coarser
// levels run on the general kernels.  A level is fusable when its box is even in every direction
// (no odd-length synthesis in the fused kernels), at least 8 long, and -- inverse only -- when the
// x-high half of a row starts 16-byte aligned (n1 % 4 == 0: its rows are fetched in 16-byte chunks).
int fused_levels(int nx, int ny, int nz, bool inverse)
{
    if ((size_t)nx * ny >= (1u << 30)) return 0;  // per-plane offsets are 32-bit
    int l = 0;
    for (; l < 4; l++) {
        const int n1 = nx >> l, n2 = ny >> l, n3 = nz >> l;
        if ((n1 | n2 | n3) & 1) break;
        if (n1 < 8 || n2 < 8 || n3 < 8) break;
        if (inverse && (n1 & 3)) break;
    }
    return l;
}

bool fused_ok(int nx, int ny, int nz, int lvl)
{
    if (lvl != 4 && lvl != -4) return false;
    // worth it from two fused levels on (98 % of the bytes), or one level of a big field
    const int f = fused_levels(nx, ny, nz, lvl < 0);
    return f >= 2 || (f == 1 && (size_t)nx * ny * nz >= (1u << 21));
}

size_t fused_lowbuf_elems(int nx, int ny, int nz)
{
    size_t tot = 0;
    for (int l = 1; l <= 3; l++) tot += (size_t)(nx >> l) * (ny >> l) * (nz >> l);
    return tot + 64;
}

static int pick_zps(int tiles, int m3, int per_round)
{
    // One workgroup is resident per CU (LDS), so the grid should be a small whole number of
    // rounds of 256 workgroups; every segment pays 4 extra steps (2 warm-up + 2 drain).
    // Measured at 1024^3 (level 0, 256 tiles): one round of 512 z-pairs 4.57 ms, 2 x 256 the same,
    // 4 x 128 4.72 ms, 8 x 64 4.69 ms.
    int best = m3;
    double best_cost = 1e300;
    for (int segs = 1; segs <= m3; segs++) {
        const int zps = (m3 + segs - 1) / segs;
        if (zps < 4) break;
        const long long wgs = (long long)tiles * ((m3 + zps - 1) / zps);
        const long long rounds = (wgs + per_round - 1) / per_round;
        const double cost = (double)rounds * (zps + 4);  // steps on the critical path
        if (cost < best_cost - 1e-9) { best_cost = cost; best = zps; }
    }
    return best;
}

#ifdef WR_STAMP
// diagnostic build only: phase stamps of the level-0 launch land here (8 x u64 per wave)
extern "C" __attribute__((visibility("default"))) unsigned long long* wr_stamp_buffer(size_t nwaves)
{
    if (!g_stamp_buf) { (void)hipMalloc(&g_stamp_buf, nwaves * 8 * sizeof(unsigned long long)); (void)hipMemset(g_stamp_buf, 0, nwaves * 64); }
    return g_stamp_buf;
}
#endif

// grid of one fused forward level
static dim3 fwd_grid(int n1, int n2, int n3, int* zps_out)
{
    const int m1 = n1 / 2, m2 = n2 / 2, m3 = n3 / 2;
    const int tiles = ((m1 + TXP - 1) / TXP) * ((m2 + TYP - 1) / TYP);
    const int zps = pick_zps(tiles, m3, 256 * (int)(160 * 1024 / LDS_BYTES));
    *zps_out = zps;
    return dim3(tiles, (m3 + zps - 1) / zps);
}

// min/max records (4 doubles each) transform_fwd_fused writes when it reduces min/max on the way; 0 if the
// shape does not run all four levels fused (the stand-alone reductions are used then)
size_t fused_minmax_records(int nx, int ny, int nz)
{
    if (fused_levels(nx, ny, nz, false) != 4) return 0;
    size_t tot = 0;
    for (int l = 0; l < 4; l++) {
        int zps;
        const dim3 g = fwd_grid(nx >> l, ny >> l, nz >> l, &zps);
        tot += (size_t)g.x * g.y * NWAVE;
    }
    return tot;
}

// 4-wide records -> result[0..3] = {in lo, in hi, out lo, out hi}.  The input range comes from the first
// `n_in` records only (level 0 reads the field; deeper levels read low-pass boxes).
__global__ __launch_bounds__(256) void k_minmax_final4(const double* __restrict__ rec, int n_in, int n_all, double* __restrict__ result)
{
    const double nan = __builtin_nan("");
    double v[4] = {nan, nan, nan, nan};
    for (int i = threadIdx.x; i < n_all; i += blockDim.x) {
        if (i < n_in) { v[0] = fmin(v[0], rec[4 * i]); v[1] = fmax(v[1], rec[4 * i + 1]); }
        v[2] = fmin(v[2], rec[4 * i + 2]); v[3] = fmax(v[3], rec[4 * i + 3]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        v[0] = fmin(v[0], __shfl_down(v[0], o, 64)); v[1] = fmax(v[1], __shfl_down(v[1], o, 64));
        v[2] = fmin(v[2], __shfl_down(v[2], o, 64)); v[3] = fmax(v[3], __shfl_down(v[3], o, 64));
    }
    __shared__ double sh[4][4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 4; k++) sh[w][k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; i++) {
            v[0] = fmin(v[0], sh[i][0]); v[1] = fmax(v[1], sh[i][1]); v[2] = fmin(v[2], sh[i][2]); v[3] = fmax(v[3], sh[i][3]);
        }
        for (int k = 0; k < 4; k++) result[k] = v[k];
    }
}

// The fused kernels need more dynamic LDS than a kernel gets by default: raised once per process, and the one call whose
// failure would make every later launch fail with "invalid argument" is checked here, where the cause is still known.
const char* fused_prepare()
{
    // (per device: a process that holds contexts on several GPUs raises the limit on each of them -- the attribute belongs to
    // the function as loaded on ONE device; the calling thread has its context's device bound)
    static std::mutex mu;
    static char msg[256];
    static int state[64];  // 0: not tried, 1: ok, 2: failed
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); dev = 0; }
    std::lock_guard<std::mutex> lk(mu);
    if (state[dev] == 1) return nullptr;
    if (state[dev] == 2) return msg;
    struct { const void* fn; const char* name; int bytes; } ks[] = {
        {(const void*)k_fwd_fused<false, false>, "k_fwd_fused<false,false>", (int)LDS_BYTES},
        {(const void*)k_fwd_fused<true, true>, "k_fwd_fused<true,true>", (int)LDS_BYTES},
        {(const void*)k_fwd_fused<false, true>, "k_fwd_fused<false,true>", (int)LDS_BYTES},
        {(const void*)k_inv_fused, "k_inv_fused", (int)LDS_INV}};
    for (const auto& k : ks) {
        const hipError_t e = hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, k.bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            snprintf(msg, sizeof msg, "fused transform: %s cannot have %d bytes of dynamic LDS on device %d (%s); this build is for gfx950 (160 KB of LDS per CU)",
                     k.name, k.bytes, dev, hipGetErrorString(e));
            state[dev] = 2;
            return msg;
        }
    }
    state[dev] = 1;
    return nullptr;
}

// mm_partial != nullptr (needs fused_minmax_records() > 0): min/max of the field and of the coefficient array
// are reduced on the way and land in mm_result[0..3].
void transform_fwd_fused(double* src, double* dst, double* lowbuf, int nx, int ny, int nz, hipStream_t st,
                         double* mm_partial, double* mm_result)
{
    (void)fused_prepare();  // (callers check it before they choose this path: wr_pipeline.cpp, wr_codec.cpp)
    const size_t d_sy = (size_t)nx, d_sz = (size_t)nx * ny;
    const int nfused = fused_levels(nx, ny, nz, false);
    const bool mm = mm_partial != nullptr && nfused == 4;
    const double* in = src;
    size_t in_sy = d_sy, in_sz = d_sz;
    double* lb = lowbuf;
    double* rec = mm_partial;
    int n_in = 0, n_all = 0;
    for (int l = 0; l < nfused; l++) {
        const int n1 = nx >> l, n2 = ny >> l, n3 = nz >> l;
        const int m1 = n1 / 2, m2 = n2 / 2, m3 = n3 / 2;
        double* lo;
        size_t lo_sy, lo_sz;
        // the last fused level leaves its low-pass octant in the coefficient array itself
        if (l < nfused - 1) { lo = lb; lo_sy = (size_t)m1; lo_sz = (size_t)m1 * m2; }
        else { lo = dst; lo_sy = d_sy; lo_sz = d_sz; }
        int zps;
        const dim3 grid = fwd_grid(n1, n2, n3, &zps);
        const int lll = l == nfused - 1;
#ifdef WR_STAMP
        hipLaunchKernelGGL((k_fwd_fused<false, false>), grid, dim3(NTHR), LDS_BYTES, st, in, in_sy, in_sz, dst, d_sy, d_sz, lo, lo_sy,
                           lo_sz, n1, n2, n3, zps, (double*)nullptr, 0, l == 0 ? g_stamp_buf : nullptr);
#else
        if (!mm)
            hipLaunchKernelGGL((k_fwd_fused<false, false>), grid, dim3(NTHR), LDS_BYTES, st, in, in_sy, in_sz, dst, d_sy, d_sz, lo, lo_sy,
                               lo_sz, n1, n2, n3, zps, (double*)nullptr, 0);
        else if (l == 0)
            hipLaunchKernelGGL((k_fwd_fused<true, true>), grid, dim3(NTHR), LDS_BYTES, st, in, in_sy, in_sz, dst, d_sy, d_sz, lo, lo_sy,
                               lo_sz, n1, n2, n3, zps, rec, lll);
        else
            hipLaunchKernelGGL((k_fwd_fused<false, true>), grid, dim3(NTHR), LDS_BYTES, st, in, in_sy, in_sz, dst, d_sy, d_sz, lo, lo_sy,
                               lo_sz, n1, n2, n3, zps, rec, lll);
#endif
        const int nrec = (int)(grid.x * grid.y * NWAVE);
        if (l == 0) n_in = nrec;
        n_all += nrec;
        rec += (size_t)nrec * 4;
        in = lo; in_sy = lo_sy; in_sz = lo_sz;
        lb += (size_t)m1 * m2 * m3;
    }
    if (mm) hipLaunchKernelGGL(k_minmax_final4, dim3(1), dim3(256), 0, st, mm_partial, n_in, n_all, mm_result);
    // coarser levels whose boxes are odd somewhere: general kernels, in place on the corner box of
    // dst; the input array has been fully consumed by level 0 and serves as their ping-pong scratch
    for (int k = nfused; k < 4; k++) transform_level(dst, src, nx, ny, nz, k, false, st);
}

// src holds the coefficient array; the reconstruction lands in dst
void transform_inv_fused(double* src, double* dst, double* lowbuf, int nx, int ny, int nz, hipStream_t st)
{
    (void)fused_prepare();
    const size_t f_sy = (size_t)nx, f_sz = (size_t)nx * ny;
    const int nfused = fused_levels(nx, ny, nz, true);
    // compact reconstruction buffers: C1 = (n/2)^3, C2 = (n/4)^3, C3 = (n/8)^3, laid out as in the forward pass
    double* cbuf[4] = {nullptr, lowbuf, nullptr, nullptr};
    cbuf[2] = cbuf[1] + (size_t)(nx >> 1) * (ny >> 1) * (nz >> 1);
    cbuf[3] = cbuf[2] + (size_t)(nx >> 2) * (ny >> 2) * (nz >> 2);
    // coarsest levels the fused kernel cannot take (odd boxes, or an x-high half that is not 16-byte
    // aligned): general kernels, in place on the corner box of the coefficient array; dst has not
    // been written yet and serves as their ping-pong scratch
    for (int k = 3; k >= nfused; k--) transform_level(src, dst, nx, ny, nz, k, true, st);
    for (int l = nfused - 1; l >= 0; l--) {
        const int n1 = nx >> l, n2 = ny >> l, n3 = nz >> l;
        const int m1 = n1 / 2, m2 = n2 / 2, m3 = n3 / 2;
        const double* lo;
        size_t lo_sy, lo_sz;
        // the coarsest fused level finds its low-pass octant in the coefficient array itself
        if (l == nfused - 1) { lo = src; lo_sy = f_sy; lo_sz = f_sz; }
        else { lo = cbuf[l + 1]; lo_sy = (size_t)m1; lo_sz = (size_t)m1 * m2; }
        double* o;
        size_t o_sy, o_sz;
        if (l == 0) { o = dst; o_sy = f_sy; o_sz = f_sz; }
        else { o = cbuf[l]; o_sy = (size_t)n1; o_sz = (size_t)n1 * n2; }
        const int tiles = ((m1 + ITXP - 1) / ITXP) * ((m2 + ITYP - 1) / ITYP);
        const int zps = pick_zps(tiles, m3, 256 * (int)(160 * 1024 / LDS_INV));
        dim3 grid(tiles, (m3 + zps - 1) / zps);
#ifdef WR_STAMP
        hipLaunchKernelGGL(k_inv_fused, grid, dim3(INTHR), LDS_INV, st, src, f_sy, f_sz, lo, lo_sy, lo_sz, o, o_sy, o_sz, n1,
                           n2, n3, zps, l == 0 ? g_stamp_buf : nullptr);
#else
        hipLaunchKernelGGL(k_inv_fused, grid, dim3(INTHR), LDS_INV, st, src, f_sy, f_sz, lo, lo_sy, lo_sz, o, o_sy, o_sz, n1,
                           n2, n3, zps);
#endif
    }
}

}  // namespace wrk
