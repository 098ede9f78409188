// ASan/UBSan harness for the host range coder (CPU build only; sanitizers are not available on the GPU pool)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "wr_rangecoder.h"
static unsigned long long s = 88172645463325252ull;
static unsigned rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 11); }
int main() {
    size_t sizes[] = {0, 1, 2, 3, 255, 59999, 60000, 60001, 120000, 180001, 400003};
    for (size_t n : sizes) for (int kind = 0; kind < 5; kind++) {
        std::vector<uint8_t> p(n ? n : 1), out(wrrc::encode_bound(n)), back(n ? n : 1);
        for (size_t i = 0; i < n; i++) {
            unsigned r = rnd();
            p[i] = kind == 0 ? r & 255 : kind == 1 ? ((r & 255) < 200 ? 0 : r >> 8 & 7) : kind == 2 ? 255 : kind == 3 ? (i % 251) : ((r & 1023) == 0 ? 255 : 1);
        }
        size_t len = wrrc::encode_plane(p.data(), n, out.data(), nullptr);
        if (len > out.size()) { printf("bound exceeded n=%zu\n", n); return 1; }
        // exact-size copy so that any over-read of the stream is caught by ASan
        std::vector<uint8_t> exact(out.begin(), out.begin() + len);
        size_t got = wrrc::decode_plane(exact.data(), len, back.data(), n);
        if (got != n || memcmp(back.data(), p.data(), n)) { printf("roundtrip failed n=%zu kind=%d got=%zu\n", n, kind, got); return 1; }
        // truncated and corrupted streams must not crash or over-read / over-write
        for (int trial = 0; trial < 6 && len > 8; trial++) {
            std::vector<uint8_t> bad(exact);
            if (trial < 3) bad.resize(len * (trial + 1) / 4);
            else for (int k = 0; k < 16; k++) bad[rnd() % bad.size()] ^= (uint8_t)(1 + rnd() % 255);
            std::vector<uint8_t> dst(n ? n : 1);
            (void)wrrc::decode_plane(bad.data(), bad.size(), dst.data(), n);
        }
    }
    printf("range coder sanitizer run OK\n");
    return 0;
}
