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
            for (int mode = 0; mode < 2; mode++) {
                if (mode == 0) wrrc::encode_planes(count, none.data(), n, op.data(), nullptr, wlen.data(), pe.data());
                else if (!wrrc::encode_planes_vec(count, none.data(), ns.data(), op.data(), wlen.data(), pe.data())) continue;
                for (int k = 0; k < count; k++) {
                    delete[] we[k].cur; we[k].cur = nullptr;
                    if (wlen[k] != len[k] || memcmp(wenc[k].data(), enc[k].data(), len[k])) { printf("windowed encode differs n=%zu mode=%d k=%d\n", n, mode, k); return 1; }
                }
                for (int k = 0; k < count; k++) back[k].assign(n, 0xEE);
                if (mode == 0) wrrc::decode_planes(count, ip.data(), len.data(), noned.data(), n, got.data(), pd.data());
                else wrrc::decode_planes_vec(count, ip.data(), len.data(), noned.data(), ns.data(), got.data(), pd.data());
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
                    else wrrc::decode_planes_vec(count, ip2.data(), l2.data(), noned.data(), ns.data(), got.data(), pd.data());
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
            for (int mode = 0; mode < 2; mode++) {
                for (int k = 0; k < count; k++) back[k].assign(n, 0xEE);
                if (mode == 0) wrrc::decode_planes(count, ip.data(), len.data(), noned.data(), n, got.data(), pd.data());
                else if (!wrrc::decode_planes_vec(count, ip.data(), len.data(), noned.data(), ns.data(), got.data(), pd.data())) continue;
                for (int k = 0; k < count; k++)
                    if (got[k] != n || memcmp(back[k].data(), p[k].data(), n)) { printf("ragged blocks: windowed decode differs trial=%d mode=%d k=%d got=%zu\n", trial, mode, k, got[k]); return 1; }
            }
        }
    }
    // An encoder whose block model is not the symbols' (the GPU delivers the histograms; a wrong one must end the stream, not
    // the process): (a) histograms that do not add up to the block, (b) histograms that add up but count a symbol that turns
    // up zero times -- the coder's range goes to zero there, which would feed the renormalisation loop for ever.  The stream
    // must end with (size_t)-1 within the buffer (bound from those histograms + kFailedBlockSlack, exact size: ASan sees an
    // overrun), and the healthy streams in the same loop must be untouched.  Scalar loops (pool without AVX-512 is not
    // reachable here: encode_planes), the 16-lane loop through the pool.
    {
        const size_t n = (size_t)60000 * 6 + 4321;
        const size_t nblk = n / 60000 + 1;
        const int count = 20;   // more than a scalar loop holds: the pool's sessions run the 16-lane loop
        std::vector<std::vector<uint8_t>> p(count), want(count), out(count);
        std::vector<std::vector<uint16_t>> hist(count);
        std::vector<size_t> wlen(count);
        for (int k = 0; k < count; k++) {
            p[k].resize(n);
            for (size_t i = 0; i < n; i++) { unsigned r = rnd(); p[k][i] = k % 2 ? (uint8_t)((r & 7) ? 127 : 128 + (r >> 8 & 1)) : (uint8_t)(r & 255); }
            hist[k].assign(nblk * 256, 0);
            for (size_t i = 0; i < n; i++) hist[k][(i / 60000) * 256 + p[k][i]]++;
            want[k].resize(wrrc::encode_bound(n));
            wlen[k] = wrrc::encode_plane(p[k].data(), n, want[k].data(), hist[k].data());
        }
        for (int variant = 0; variant < 3; variant++) {
            // the victims' histograms are damaged in block 2: (0) one count too many: the sum is off; (1) the count of a symbol
            // that occurs moved to one that does not; (2) all counts moved to symbol 0 of a plane that holds other symbols
            const int victims[3] = {1, 4, 17};
            std::vector<std::vector<uint16_t>> h2(hist);
            for (int v : victims) {
                uint16_t* h = h2[v].data() + 2 * 256;
                int present = -1, absent = -1;
                for (int b = 255; b >= 0; b--) { if (h[b] && present < 0) present = b; if (!h[b] && absent < 0) absent = b; }
                if (variant == 0) h[present]++;
                else if (variant == 1 && absent >= 0) { h[absent] = h[present]; h[present] = 0; }
                else { unsigned sum = 0; for (int b = 0; b < 256; b++) { sum += h[b]; h[b] = 0; } h[p[v][2 * 60000] == 0 ? 1 : 0] = (uint16_t)sum; }
            }
            for (int pass = 0; pass < 2; pass++) {  // 0: scalar interleaved loops on this thread, 1: the pool (16-lane sessions)
                std::vector<uint8_t*> op(count);
                std::vector<const uint8_t*> sp(count);
                std::vector<const uint16_t*> hp(count);
                std::vector<size_t> len(count, 0);
                for (int k = 0; k < count; k++) {
                    out[k].assign(wrrc::encode_bound_hist(h2[k].data(), n) + wrrc::kFailedBlockSlack, 0xEE);
                    out[k].shrink_to_fit();
                    op[k] = out[k].data(); sp[k] = p[k].data(); hp[k] = h2[k].data();
                }
                if (pass == 0) wrrc::encode_planes(count, sp.data(), n, op.data(), hp.data(), len.data());
                else {
                    wrrc::pool_configure(2, 4);
                    std::vector<wrrc::PlaneJob> jobs(count);
                    wrrc::JobBatch batch;
                    for (int k = 0; k < count; k++) { jobs[k].kind = wrrc::PlaneJob::kEncode; jobs[k].src = sp[k]; jobs[k].n = n; jobs[k].dst = op[k]; jobs[k].hist = hp[k]; }
                    if (!wrrc::pool_submit(jobs.data(), count, &batch)) { printf("pool refused\n"); return 1; }
                    wrrc::pool_wait(&batch);
                    for (int k = 0; k < count; k++) len[k] = jobs[k].result;
                    wrrc::pool_configure(0, 0);
                }
                for (int k = 0; k < count; k++) {
                    const bool victim = k == victims[0] || k == victims[1] || k == victims[2];
                    if (victim) { if (len[k] != (size_t)-1) { printf("wrong histograms: stream %d did not fail (variant %d pass %d len %zu)\n", k, variant, pass, len[k]); return 1; } }
                    else if (len[k] != wlen[k] || memcmp(out[k].data(), want[k].data(), wlen[k])) { printf("wrong histograms next door: healthy stream %d differs (variant %d pass %d)\n", k, variant, pass); return 1; }
                }
            }
        }
        // a refused window (PlaneWindow: null with a count): the encoder and the decoder give the stream up without touching a symbol
        struct Refuser {
            const uint8_t* plane; size_t allow, calls;  // serves windows below `allow`, refuses from there on
            static uint8_t* fn(void* u, size_t first, size_t* count)
            {
                Refuser* r = (Refuser*)u;
                r->calls++;
                if (*count == 0 || first >= r->allow) return nullptr;
                if (*count > 60000) *count = 60000;
                return const_cast<uint8_t*>(r->plane) + first;
            }
        };
        for (size_t allow : {(size_t)0, (size_t)120000}) {
            Refuser r[3] = {{p[0].data(), allow, 0}, {p[1].data(), (size_t)-1, 0}, {p[2].data(), allow, 0}};
            wrrc::PlaneWindow io[3] = {{Refuser::fn, &r[0]}, {Refuser::fn, &r[1]}, {Refuser::fn, &r[2]}};
            const wrrc::PlaneWindow* pio[3] = {&io[0], &io[1], &io[2]};
            const uint8_t* none[3] = {nullptr, nullptr, nullptr};
            uint8_t* op[3];
            const uint16_t* hp[3] = {hist[0].data(), hist[1].data(), hist[2].data()};
            size_t len[3] = {0, 0, 0};
            for (int k = 0; k < 3; k++) { out[k].assign(wrrc::encode_bound_hist(hist[k].data(), n) + wrrc::kFailedBlockSlack, 0xEE); op[k] = out[k].data(); }
            wrrc::encode_planes(3, none, n, op, hp, len, pio);
            if (len[0] != (size_t)-1 || len[2] != (size_t)-1 || len[1] != wlen[1] || memcmp(out[1].data(), want[1].data(), wlen[1])) {
                printf("refused window: encoder lens %zu %zu %zu (allow %zu)\n", len[0], len[1], len[2], allow); return 1;
            }
            // decoder: the output window is refused
            std::vector<uint8_t> sink(n);
            struct DRef { uint8_t* out; size_t allow; static uint8_t* fn(void* u, size_t first, size_t* count) { DRef* d = (DRef*)u; if (*count == 0 || first >= d->allow) return nullptr; if (*count > 60000) *count = 60000; return d->out + first; } };
            DRef d[2] = {{sink.data(), allow}, {sink.data(), (size_t)-1}};
            wrrc::PlaneWindow dio[2] = {{DRef::fn, &d[0]}, {DRef::fn, &d[1]}};
            const wrrc::PlaneWindow* pdio[2] = {&dio[0], &dio[1]};
            const uint8_t* ins[2] = {want[0].data(), want[0].data()};
            size_t lens2[2] = {wlen[0], wlen[0]}, got2[2] = {0, 0};
            uint8_t* noned[2] = {nullptr, nullptr};
            wrrc::decode_planes(1, ins, lens2, noned, n, got2, pdio);
            if (got2[0] != (size_t)-1) { printf("refused window: decoder result %zu (allow %zu)\n", got2[0], allow); return 1; }
            wrrc::decode_planes(1, ins + 1, lens2 + 1, noned, n, got2 + 1, pdio + 1);
            if (got2[1] != n || memcmp(sink.data(), p[0].data(), n)) { printf("served windows: decoder failed\n"); return 1; }
        }
    }
    printf("range coder sanitizer run OK\n");
    return 0;
}
