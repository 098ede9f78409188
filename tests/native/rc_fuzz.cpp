// ASan/UBSan harness for the host range coder (CPU build only; sanitizers are not available on the GPU pool)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "wr_rangecoder.h"
#include "waverange_amd.h"  // the exported rngcod13 primitives (wr_compat.cpp): an encoder with block sizes of its own
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
    // four noise planes in one loop: the hand-allotted form of the byte-aligned decoder loop (decode_symbols_noise4_asm), also
    // with one stream of the four damaged or cut short
    for (int round = 0; round < 3; round++) {
        const size_t n = (size_t)60000 * (5 + round) + 1234 * round;
        const int count = 4;
        std::vector<std::vector<uint8_t>> p(count), enc(count), back(count);
        std::vector<const uint8_t*> ip(count);
        std::vector<uint8_t*> bp(count);
        std::vector<size_t> len(count), got(count);
        for (int k = 0; k < count; k++) {
            p[k].resize(n); enc[k].resize(wrrc::encode_bound(n)); back[k].resize(n);
            for (size_t i = 0; i < n; i++) { const unsigned r = rnd(); p[k][i] = (uint8_t)(round == 2 ? (r & 255) : ((r & 255) < (r >> 8 & 255) ? (r & 255) : (r >> 8 & 255))); }
            len[k] = wrrc::encode_plane(p[k].data(), n, enc[k].data(), nullptr);
            enc[k].resize(len[k]); enc[k].shrink_to_fit();
            ip[k] = enc[k].data(); bp[k] = back[k].data();
        }
        wrrc::decode_planes(count, ip.data(), len.data(), bp.data(), n, got.data());
        for (int k = 0; k < count; k++)
            if (got[k] != n || memcmp(back[k].data(), p[k].data(), n)) { printf("four noise planes: decode failed round=%d k=%d\n", round, k); return 1; }
        for (int trial = 0; trial < 4; trial++) {
            const int victim = (int)(rnd() % count);
            std::vector<uint8_t> bad(enc[victim]);
            if (trial < 2) bad.resize(bad.size() * (trial + 1) / 3);
            else for (int j = 0; j < 64; j++) bad[rnd() % bad.size()] ^= (uint8_t)(1 + rnd() % 255);
            std::vector<const uint8_t*> ip2(ip); std::vector<size_t> l2(len);
            ip2[victim] = bad.data(); l2[victim] = bad.size();
            wrrc::decode_planes(count, ip2.data(), l2.data(), bp.data(), n, got.data());
            for (int k = 0; k < count; k++)
                if (k != victim && (got[k] != n || memcmp(back[k].data(), p[k].data(), n))) { printf("four noise planes: healthy stream disturbed by a damaged one\n"); return 1; }
        }
    }
    // several planes interleaved in one loop (encode_planes / decode_planes): same bytes as one by one,
    // streams long enough for the unchecked fast loop (needs 3 * 60000 + 8 unread bytes per stream)
    for (size_t n : {(size_t)7, (size_t)60000 * 2, (size_t)60000 * 9 + 17}) for (int count = 1; count <= 6; count++) {
        std::vector<std::vector<uint8_t>> p(count), one(count), multi(count), back(count);
        std::vector<const uint8_t*> pp(count), ip(count);
        std::vector<uint8_t*> op(count), bp(count);
        std::vector<size_t> len1(count), lenm(count), got(count);
        for (int k = 0; k < count; k++) {
            p[k].resize(n); one[k].resize(wrrc::encode_bound(n)); multi[k].resize(wrrc::encode_bound(n)); back[k].resize(n);
            for (size_t i = 0; i < n; i++) { unsigned r = rnd(); p[k][i] = k % 3 == 0 ? r & 255 : k % 3 == 1 ? (r & 255) % 23 + 100 : ((r & 63) ? 128 : r >> 8 & 255); }
            len1[k] = wrrc::encode_plane(p[k].data(), n, one[k].data(), nullptr);
            pp[k] = p[k].data(); op[k] = multi[k].data();
        }
        wrrc::encode_planes(count, pp.data(), n, op.data(), nullptr, lenm.data());
        for (int k = 0; k < count; k++) {
            if (lenm[k] != len1[k] || memcmp(multi[k].data(), one[k].data(), len1[k])) { printf("interleaved encode differs n=%zu count=%d k=%d\n", n, count, k); return 1; }
            multi[k].resize(len1[k]); multi[k].shrink_to_fit();  // exact size: over-reads are caught
            ip[k] = multi[k].data(); bp[k] = back[k].data();
        }
        wrrc::decode_planes(count, ip.data(), len1.data(), bp.data(), n, got.data());
        for (int k = 0; k < count; k++)
            if (got[k] != n || memcmp(back[k].data(), p[k].data(), n)) { printf("interleaved decode failed n=%zu count=%d k=%d\n", n, count, k); return 1; }
        for (int trial = 0; trial < 4; trial++) {  // one damaged stream among healthy ones
            const int victim = (int)(rnd() % count);
            std::vector<uint8_t> bad(multi[victim]);
            if (trial < 2) bad.resize(bad.size() * (trial + 1) / 3);
            else for (int j = 0; j < 16; j++) bad[rnd() % bad.size()] ^= (uint8_t)(1 + rnd() % 255);
            std::vector<const uint8_t*> ip2(ip); std::vector<size_t> l2(len1);
            ip2[victim] = bad.data(); l2[victim] = bad.size();
            wrrc::decode_planes(count, ip2.data(), l2.data(), bp.data(), n, got.data());
            for (int k = 0; k < count; k++)
                if (k != victim && (got[k] != n || memcmp(back[k].data(), p[k].data(), n))) { printf("healthy stream disturbed by a damaged one\n"); return 1; }
        }
    }
    // the 16-lane AVX-512 loop (dominant-symbol planes) and the coder pool: 20 planes of mixed kinds and lengths,
    // exact-size stream copies, damaged streams among healthy ones
    {
        const int count = 20;
        std::vector<std::vector<uint8_t>> p(count), enc(count), back(count);
        std::vector<const uint8_t*> ip(count);
        std::vector<uint8_t*> bp(count);
        std::vector<size_t> n(count), len(count), got(count);
        for (int k = 0; k < count; k++) {
            n[k] = (size_t)60000 * (5 + k % 4) + (k % 3) * 777;
            p[k].resize(n[k]); back[k].resize(n[k]);
            for (size_t i = 0; i < n[k]; i++) { unsigned r = rnd(); p[k][i] = k % 5 == 4 ? r & 255 : ((r & 7) ? 254 : 255 - (k % 2) * ((r >> 8 & 255) == 0 ? 200 : 0)); }
            std::vector<uint8_t> out(wrrc::encode_bound(n[k]));
            len[k] = wrrc::encode_plane(p[k].data(), n[k], out.data(), nullptr);
            enc[k].assign(out.begin(), out.begin() + len[k]);
            ip[k] = enc[k].data(); bp[k] = back[k].data();
        }
        if (wrrc::decode_planes_vec(count, ip.data(), len.data(), bp.data(), n.data(), got.data())) {
            for (int k = 0; k < count; k++)
                if (got[k] != n[k] || memcmp(back[k].data(), p[k].data(), n[k])) { printf("vector decode failed k=%d\n", k); return 1; }
            for (int trial = 0; trial < 4; trial++) {
                const int victim = (int)(rnd() % count);
                std::vector<uint8_t> bad(enc[victim]);
                if (trial < 2) bad.resize(bad.size() * (trial + 1) / 3);
                else for (int j = 0; j < 16; j++) bad[rnd() % bad.size()] ^= (uint8_t)(1 + rnd() % 255);
                std::vector<const uint8_t*> ip2(ip); std::vector<size_t> l2(len);
                ip2[victim] = bad.data(); l2[victim] = bad.size();
                wrrc::decode_planes_vec(count, ip2.data(), l2.data(), bp.data(), n.data(), got.data());
                for (int k = 0; k < count; k++)
                    if (k != victim && (got[k] != n[k] || memcmp(back[k].data(), p[k].data(), n[k]))) { printf("vector: healthy stream disturbed\n"); return 1; }
            }
            // the decoder's loop for planes of any statistics: the same planes, exact-size streams, damaged ones among healthy ones
            for (int k = 0; k < count; k++) memset(back[k].data(), 0xEE, n[k]);
            wrrc::decode_planes_vec(count, ip.data(), len.data(), bp.data(), n.data(), got.data(), nullptr, true);
            for (int k = 0; k < count; k++)
                if (got[k] != n[k] || memcmp(back[k].data(), p[k].data(), n[k])) { printf("any-statistics vector decode failed k=%d\n", k); return 1; }
            for (int trial = 0; trial < 6; trial++) {
                const int victim = (int)(rnd() % count);
                std::vector<uint8_t> bad(enc[victim]);
                if (trial < 2) bad.resize(bad.size() * (trial + 1) / 3);
                else for (int j = 0; j < 16; j++) bad[rnd() % bad.size()] ^= (uint8_t)(1 + rnd() % 255);
                std::vector<const uint8_t*> ip2(ip); std::vector<size_t> l2(len);
                ip2[victim] = bad.data(); l2[victim] = bad.size();
                wrrc::decode_planes_vec(count, ip2.data(), l2.data(), bp.data(), n.data(), got.data(), nullptr, true);
                for (int k = 0; k < count; k++)
                    if (k != victim && (got[k] != n[k] || memcmp(back[k].data(), p[k].data(), n[k]))) { printf("any-statistics vector: healthy stream disturbed\n"); return 1; }
            }
            // the encoder's vector loop: exact-size output is not possible (the length is the result), so the
            // buffers have the documented bound and the bytes are compared
            std::vector<std::vector<uint8_t>> venc(count);
            std::vector<const uint8_t*> sp(count);
            std::vector<uint8_t*> vp(count);
            std::vector<size_t> vlen(count);
            for (int k = 0; k < count; k++) { venc[k].resize(wrrc::encode_bound(n[k])); sp[k] = p[k].data(); vp[k] = venc[k].data(); }
            if (!wrrc::encode_planes_vec(count, sp.data(), n.data(), vp.data(), vlen.data())) { printf("vector encode unavailable\n"); return 1; }
            for (int k = 0; k < count; k++)
                if (vlen[k] != len[k] || memcmp(venc[k].data(), enc[k].data(), len[k])) { printf("vector encode differs k=%d\n", k); return 1; }
        } else printf("(no AVX-512 here: vector loops not exercised)\n");
        wrrc::pool_configure(3, 4);
        std::vector<wrrc::PlaneJob> jobs(2 * count);
        std::vector<std::vector<uint8_t>> out2(count), back2(count);
        wrrc::JobBatch batch;
        for (int k = 0; k < count; k++) {
            out2[k].resize(wrrc::encode_bound(n[k])); back2[k].resize(n[k]);
            jobs[k].kind = wrrc::PlaneJob::kEncode; jobs[k].src = p[k].data(); jobs[k].n = n[k]; jobs[k].dst = out2[k].data();
            jobs[count + k].kind = wrrc::PlaneJob::kDecode; jobs[count + k].src = enc[k].data(); jobs[count + k].src_len = len[k];
            jobs[count + k].dst = back2[k].data(); jobs[count + k].n = n[k];
        }
        wrrc::pool_submit(jobs.data(), 2 * count, &batch);
        wrrc::pool_wait(&batch);
        for (int k = 0; k < count; k++) {
            if (jobs[k].result != len[k] || memcmp(out2[k].data(), enc[k].data(), len[k])) { printf("pool encode differs k=%d\n", k); return 1; }
            if (jobs[count + k].result != n[k] || memcmp(back2[k].data(), p[k].data(), n[k])) { printf("pool decode failed k=%d\n", k); return 1; }
        }
        wrrc::pool_configure(0, 0);
    }
    // streams changing workers: long planes of every kind submitted at once to a pool of five -- the first workers take
    // them all into 16-lane sessions, the idle ones take over halves at block boundaries (Pool::offer / next)
    {
        const int count = 22;
        const size_t n = (size_t)60000 * 9 + 1234;
        std::vector<std::vector<uint8_t>> p(count), enc(count), out2(count), back2(count);
        std::vector<size_t> len(count);
        for (int k = 0; k < count; k++) {
            p[k].resize(n);
            for (size_t i = 0; i < n; i++) { unsigned r = rnd(); p[k][i] = k % 3 == 0 ? r & 255 : k % 3 == 1 ? ((r & 7) ? 254 : 255) : ((r & 1023) ? 7 : r >> 12 & 255); }
            std::vector<uint8_t> o(wrrc::encode_bound(n));
            len[k] = wrrc::encode_plane(p[k].data(), n, o.data(), nullptr);
            enc[k].assign(o.begin(), o.begin() + len[k]);
        }
        const unsigned long moved0 = wrrc::pool_streams_moved();
        wrrc::pool_configure(5, 4);
        for (int rep = 0; rep < 3; rep++) {
            std::vector<wrrc::PlaneJob> jobs(2 * count);
            wrrc::JobBatch batch;
            for (int k = 0; k < count; k++) {
                out2[k].assign(wrrc::encode_bound(n), 0); back2[k].assign(n, 0xEE);
                jobs[k].kind = wrrc::PlaneJob::kEncode; jobs[k].src = p[k].data(); jobs[k].n = n; jobs[k].dst = out2[k].data();
                jobs[count + k].kind = wrrc::PlaneJob::kDecode; jobs[count + k].src = enc[k].data(); jobs[count + k].src_len = len[k];
                jobs[count + k].dst = back2[k].data(); jobs[count + k].n = n;
            }
            if (!wrrc::pool_submit(jobs.data(), 2 * count, &batch)) { printf("pool refused the jobs\n"); return 1; }
            wrrc::pool_wait(&batch);
            for (int k = 0; k < count; k++) {
                if (jobs[k].result != len[k] || memcmp(out2[k].data(), enc[k].data(), len[k])) { printf("moved streams: pool encode differs k=%d\n", k); return 1; }
                if (jobs[count + k].result != n || memcmp(back2[k].data(), p[k].data(), n)) { printf("moved streams: pool decode failed k=%d\n", k); return 1; }
            }
        }
        wrrc::pool_configure(0, 0);
        printf("streams that changed workers: %lu\n", wrrc::pool_streams_moved() - moved0);
    }
    // windowed symbol access (PlaneWindow): exact-size window buffers allocated afresh for every window, so ASan sees
    // any access outside the window or to a window that was handed back
    {
        struct Win {
            const uint8_t* plane; uint8_t* out; size_t n, chunk; uint8_t* cur; size_t cur_first, cur_count; bool dec;
            static uint8_t* fn(void* u, size_t first, size_t* count)
            {
                Win* w = (Win*)u;
                if (w->dec && w->cur) memcpy(w->out + w->cur_first, w->cur, w->cur_count);
                delete[] w->cur; w->cur = nullptr;
                if (*count == 0) return nullptr;
                const size_t c = *count < w->chunk ? *count : w->chunk;
                w->cur = new uint8_t[c]; w->cur_first = first; w->cur_count = c;
                if (!w->dec) memcpy(w->cur, w->plane + first, c);
                *count = c;
                return w->cur;
            }
        };
        for (size_t n : {(size_t)1, (size_t)60000, (size_t)60000 * 3, (size_t)60000 * 5 + 77}) {
            const int count = 6;
            std::vector<std::vector<uint8_t>> p(count), enc(count), wenc(count), back(count);
            std::vector<Win> we(count), wd(count);
            std::vector<wrrc::PlaneWindow> ioe(count), iod(count);
            std::vector<const wrrc::PlaneWindow*> pe(count), pd(count);
            std::vector<const uint8_t*> none(count, nullptr), ip(count);
            std::vector<uint8_t*> op(count), noned(count, nullptr);
            std::vector<size_t> len(count), wlen(count), got(count), ns(count, n);
            for (int k = 0; k < count; k++) {
                p[k].resize(n); back[k].assign(n, 0xEE);
                for (size_t i = 0; i < n; i++) { unsigned r = rnd(); p[k][i] = k % 3 == 0 ? r & 255 : k % 3 == 1 ? ((r & 15) ? 254 : 255) : (uint8_t)(100 + (r & 31)); }
                std::vector<uint8_t> o(wrrc::encode_bound(n));
                len[k] = wrrc::encode_plane(p[k].data(), n, o.data(), nullptr);
                enc[k].assign(o.begin(), o.begin() + len[k]);
                wenc[k].resize(wrrc::encode_bound(n));
                we[k] = Win{p[k].data(), nullptr, n, (size_t)60000 * (1 + k % 2), nullptr, 0, 0, false};
                wd[k] = Win{nullptr, back[k].data(), n, (size_t)60000 * (1 + k % 2), nullptr, 0, 0, true};
                ioe[k] = wrrc::PlaneWindow{Win::fn, &we[k]}; iod[k] = wrrc::PlaneWindow{Win::fn, &wd[k]};
                pe[k] = &ioe[k]; pd[k] = &iod[k]; ip[k] = enc[k].data(); op[k] = wenc[k].data();
            }
            for (int mode = 0; mode < 3; mode++) {  // 2: the decoder's loop for planes of any statistics
                if (mode == 0) wrrc::encode_planes(count, none.data(), n, op.data(), nullptr, wlen.data(), pe.data());
                else if (!wrrc::encode_planes_vec(count, none.data(), ns.data(), op.data(), wlen.data(), pe.data())) continue;
                for (int k = 0; k < count; k++) {
                    delete[] we[k].cur; we[k].cur = nullptr;
                    if (wlen[k] != len[k] || memcmp(wenc[k].data(), enc[k].data(), len[k])) { printf("windowed encode differs n=%zu mode=%d k=%d\n", n, mode, k); return 1; }
                }
                for (int k = 0; k < count; k++) back[k].assign(n, 0xEE);
                if (mode == 0) wrrc::decode_planes(count, ip.data(), len.data(), noned.data(), n, got.data(), pd.data());
                else wrrc::decode_planes_vec(count, ip.data(), len.data(), noned.data(), ns.data(), got.data(), pd.data(), mode == 2);
                for (int k = 0; k < count; k++)
                    if (got[k] != n || memcmp(back[k].data(), p[k].data(), n)) { printf("windowed decode failed n=%zu mode=%d k=%d\n", n, mode, k); return 1; }
                // a truncated and a corrupted stream among healthy ones
                for (int trial = 0; trial < 2 && n > 60000; trial++) {
                    std::vector<uint8_t> bad(enc[1]);
                    if (trial == 0) bad.resize(bad.size() / 2); else for (int j = 0; j < 16; j++) bad[rnd() % bad.size()] ^= (uint8_t)(1 + rnd() % 255);
                    std::vector<const uint8_t*> ip2(ip); std::vector<size_t> l2(len);
                    ip2[1] = bad.data(); l2[1] = bad.size();
                    for (int k = 0; k < count; k++) back[k].assign(n, 0xEE);
                    if (mode == 0) wrrc::decode_planes(count, ip2.data(), l2.data(), noned.data(), n, got.data(), pd.data());
                    else wrrc::decode_planes_vec(count, ip2.data(), l2.data(), noned.data(), ns.data(), got.data(), pd.data(), mode == 2);
                    for (int k = 0; k < count; k++)
                        if (k != 1 && (got[k] != n || memcmp(back[k].data(), p[k].data(), n))) { printf("windowed: healthy stream disturbed\n"); return 1; }
                }
            }
        }
    }
    // Streams whose blocks are shorter than 60000 symbols in MID-stream (the format allows them, the reference's encoder
    // never writes them: built here from the exported rngcod13 primitives, block model of wrappers.cpp:85-128 with
    // block sizes of our own).  Blocks then straddle window ends: whole-plane and windowed decoders must agree.
    {
        auto ragged_encode = [](const std::vector<uint8_t>& p, const std::vector<unsigned>& blocks, std::vector<uint8_t>* out) {
            rangecoder rc;
            init_databuf(&rc, 2 * p.size() + 600 * (blocks.size() + 2) + 1024);
            start_encoding(&rc, 0, 0);
            size_t at = 0;
            for (unsigned bs : blocks) {
                encode_freq(&rc, 1, 1, 2);
                unsigned cnt[256] = {0}, cum[257];
                for (unsigned i = 0; i < bs; i++) cnt[p[at + i]]++;
                cum[0] = 0;
                for (int b = 0; b < 256; b++) { encode_shift(&rc, 1, cnt[b], 16); cum[b + 1] = cum[b] + cnt[b]; }
                for (unsigned i = 0; i < bs; i++) encode_freq(&rc, cnt[p[at + i]], cum[p[at + i]], bs);
                at += bs;
            }
            encode_freq(&rc, 1, 0, 2);
            done_encoding(&rc);
            out->assign(rc.databuf, rc.databuf + rc.datapos);
            free_databuf(&rc);
        };
        struct Win {
            uint8_t* out; size_t n, chunk; uint8_t* cur; size_t cur_first, cur_count;
            static uint8_t* fn(void* u, size_t first, size_t* count)
            {
                Win* w = (Win*)u;
                if (w->cur) memcpy(w->out + w->cur_first, w->cur, w->cur_count);
                delete[] w->cur; w->cur = nullptr;
                if (*count == 0) return nullptr;
                const size_t c = *count < w->chunk ? *count : w->chunk;
                w->cur = new uint8_t[c]; w->cur_first = first; w->cur_count = c;
                memset(w->cur, 0xEE, c);
                *count = c;
                return w->cur;
            }
        };
        for (int trial = 0; trial < 6; trial++) {
            const int count = 5;
            const size_t n = (size_t)60000 * (4 + trial) + (trial % 2) * 4321;
            std::vector<std::vector<uint8_t>> p(count), enc(count), whole(count), back(count);
            std::vector<Win> wd(count);
            std::vector<wrrc::PlaneWindow> iod(count);
            std::vector<const wrrc::PlaneWindow*> pd(count);
            std::vector<const uint8_t*> ip(count);
            std::vector<uint8_t*> noned(count, nullptr);
            std::vector<size_t> len(count), got(count), ns(count, n);
            for (int k = 0; k < count; k++) {
                p[k].resize(n); whole[k].assign(n, 0xEE); back[k].assign(n, 0xEE);
                for (size_t i = 0; i < n; i++) { unsigned r = rnd(); p[k][i] = k % 3 == 0 ? r & 255 : k % 3 == 1 ? ((r & 15) ? 254 : 255) : (uint8_t)(100 + (r & 31)); }
                std::vector<unsigned> blocks;
                size_t left = n;
                while (left) {  // full blocks, short ones (down to one symbol) and empty ones mixed
                    unsigned r = rnd() % 8, bs = r < 4 ? 60000 : r == 4 ? 0 : 1 + rnd() % 60000;
                    if (bs > left) bs = (unsigned)left;
                    blocks.push_back(bs);
                    left -= bs;
                }
                ragged_encode(p[k], blocks, &enc[k]);
                len[k] = enc[k].size(); ip[k] = enc[k].data();
                if (wrrc::decode_plane(enc[k].data(), len[k], whole[k].data(), n) != n || memcmp(whole[k].data(), p[k].data(), n)) {
                    printf("ragged blocks: whole-plane decode failed trial=%d k=%d\n", trial, k); return 1;
                }
                wd[k] = Win{back[k].data(), n, (size_t)60000 * (1 + (k + trial) % 3), nullptr, 0, 0};
                iod[k] = wrrc::PlaneWindow{Win::fn, &wd[k]}; pd[k] = &iod[k];
            }
            for (int mode = 0; mode < 3; mode++) {
                for (int k = 0; k < count; k++) back[k].assign(n, 0xEE);
                if (mode == 0) wrrc::decode_planes(count, ip.data(), len.data(), noned.data(), n, got.data(), pd.data());
                else if (!wrrc::decode_planes_vec(count, ip.data(), len.data(), noned.data(), ns.data(), got.data(), pd.data(), mode == 2)) continue;
                for (int k = 0; k < count; k++)
                    if (got[k] != n || memcmp(back[k].data(), p[k].data(), n)) { printf("ragged blocks: windowed decode differs trial=%d mode=%d k=%d got=%zu\n", trial, mode, k, got[k]); return 1; }
            }
        }
    }
    printf("range coder sanitizer run OK\n");
    return 0;
}
