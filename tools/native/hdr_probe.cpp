// hdr_probe.cpp -- what share of a 16-lane coder session goes to the block headers (257 scalar coder steps per block and
// stream: wrappers.cpp:96-110,178-190 / rangecod.c:231-245,321-337) on the CPU this runs on: DESIGN.md 9's lead "block
// headers as steps of the 16-lane loops" is worth at most that share.  The product's coder with cycle counters around the
// header code and around whole session steps (-DWR_PROBE_HDR), 32 planes of a kind through a pool of one worker, block
// histograms supplied as the GPU supplies them.
//   g++ -O3 -std=c++17 -march=x86-64-v3 -DWR_PROBE_HDR -Iwaverange_amd/csrc tools/native/hdr_probe.cpp waverange_amd/csrc/wr_rangecoder.cpp \
//       vec.o -o hdr_probe -lpthread     (vec.o: wr_rangecoder_avx512.cpp with -mavx512f -mavx512bw -mavx512dq -mavx512vl)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "wr_rangecoder.h"

namespace wrrc { extern unsigned long long g_probe_hdr[4]; }
static unsigned long long s = 88172645463325252ull;
static unsigned rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 11); }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const size_t nb = argc > 1 ? (size_t)atoi(argv[1]) : 60;
    const size_t n = 60000 * nb;
    const int count = 32;
    const char* kinds[3] = {"two-symbol planes (p = 0.8 / 0.2)", "one dominant symbol (p = 0.9997)", "noise planes (7.4 bits per symbol)"};
    wrrc::pool_configure(1, 4);
    for (int kind = 0; kind < 3; kind++) {
        std::vector<std::vector<uint8_t>> p(count), out(count), back(count);
        std::vector<std::vector<uint16_t>> hist(count);
        for (int k = 0; k < count; k++) {
            p[k].resize(n); back[k].resize(n);
            for (size_t i = 0; i < n; i++) {
                const unsigned r = rnd();
                p[k][i] = kind == 0 ? ((r % 10) < 8 ? 127 : 128) : kind == 1 ? ((r % 10000) < 9997 ? 128 : (uint8_t)(120 + (r >> 16) % 16)) : (uint8_t)(((r & 255) < (64 + (r >> 8) % 256)) ? (r & 255) : (64 + (r >> 8) % 256) & 255);
            }
            hist[k].assign((n / 60000 + 1) * 256, 0);
            for (size_t i = 0; i < n; i++) hist[k][(i / 60000) * 256 + p[k][i]]++;
            out[k].resize(wrrc::encode_bound(n));
        }
        memset(wrrc::g_probe_hdr, 0, sizeof wrrc::g_probe_hdr);
        std::vector<wrrc::PlaneJob> jobs(count);
        {
            wrrc::JobBatch batch;
            for (int k = 0; k < count; k++) { jobs[k].kind = wrrc::PlaneJob::kEncode; jobs[k].src = p[k].data(); jobs[k].n = n; jobs[k].dst = out[k].data(); jobs[k].hist = hist[k].data(); }
            const double t = now();
            wrrc::pool_submit(jobs.data(), count, &batch);
            wrrc::pool_wait(&batch);
            printf("%-36s encode %7.1f Msym/s per worker, headers %5.2f %% of the session steps' cycles\n", kinds[kind], count * n / (now() - t) / 1e6,
                   100.0 * wrrc::g_probe_hdr[0] / (double)(wrrc::g_probe_hdr[1] ? wrrc::g_probe_hdr[1] : 1));
        }
        std::vector<size_t> len(count);
        for (int k = 0; k < count; k++) len[k] = jobs[k].result;
        {
            std::vector<wrrc::PlaneJob> dj(count);
            wrrc::JobBatch batch;
            for (int k = 0; k < count; k++) { dj[k].kind = wrrc::PlaneJob::kDecode; dj[k].src = out[k].data(); dj[k].src_len = len[k]; dj[k].dst = back[k].data(); dj[k].n = n; }
            const double t = now();
            wrrc::pool_submit(dj.data(), count, &batch);
            wrrc::pool_wait(&batch);
            const double dt = now() - t;
            for (int k = 0; k < count; k++) if (dj[k].result != n || memcmp(back[k].data(), p[k].data(), n)) { printf("decode mismatch\n"); return 1; }
            if (wrrc::g_probe_hdr[3])
                printf("%-36s decode %7.1f Msym/s per worker, headers %5.2f %% of the session steps' cycles\n", kinds[kind], count * n / dt / 1e6,
                       100.0 * wrrc::g_probe_hdr[2] / (double)wrrc::g_probe_hdr[3]);
            else
                printf("%-36s decode %7.1f Msym/s per worker (scalar loops of four: no 16-lane session)\n", kinds[kind], count * n / dt / 1e6);
        }
    }
    wrrc::pool_configure(0, 0);
    return 0;
}
