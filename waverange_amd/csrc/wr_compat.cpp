// wr_compat.cpp -- the remaining symbols the reference's libwaverange.so exports besides the
// codec API ("also visible, keep for safety"): the rngcod13 primitives on the
// reference's `rangecoder` struct (src/rangecod/rangecod.h:110-131, rangecod.c:132-404) and
// ind_p2w_3d (src/waveletcdf97_3d/waveletcdf97_3d.c:473-553).  Host-only integer code; the
// product's own plane coder is wr_rangecoder.cpp, these exist so that a client which linked the
// primitives directly keeps linking.
#include <stdlib.h>

#include "../../include/waverange_amd.h"

namespace {
constexpr unsigned kTop = 0x80000000u, kBottom = 0x00800000u;
constexpr int kShift = 23, kExtra = 7;

inline void put(rangecoder* rc, unsigned v) { rc->databuf[rc->datapos++] = (unsigned char)v; }
inline unsigned get(rangecoder* rc) { return rc->databuf[rc->datapos++]; }

void enc_normalize(rangecoder* rc)  // rangecod.c:182-207
{
    while (rc->range <= kBottom) {
        if (rc->low < (0xffu << kShift)) {
            put(rc, rc->buffer);
            for (; rc->help; rc->help--) put(rc, 0xff);
            rc->buffer = (unsigned char)(rc->low >> kShift);
        } else if (rc->low & kTop) {
            put(rc, rc->buffer + 1u);
            for (; rc->help; rc->help--) put(rc, 0);
            rc->buffer = (unsigned char)(rc->low >> kShift);
        } else {
            rc->help++;
        }
        rc->range <<= 8;
        rc->low = (rc->low << 8) & (kTop - 1);
        rc->bytecount++;
    }
}

void dec_normalize(rangecoder* rc)  // rangecod.c:294-302
{
    while (rc->range <= kBottom) {
        rc->low = (rc->low << 8) | (((unsigned)rc->buffer << kExtra) & 0xff);
        rc->buffer = (unsigned char)get(rc);
        rc->low |= rc->buffer >> (8 - kExtra);
        rc->range <<= 8;
    }
}
}  // namespace

extern "C" {

char coderversion[] = "rangecoder 1.3 NOWARN (c) 1997-2000 Michael Schindler";  // rangecod.c:132

void start_encoding(rangecoder* rc, char c, unsigned long initlength)
{
    rc->low = 0; rc->range = kTop; rc->buffer = (unsigned char)c; rc->help = 0; rc->bytecount = (unsigned)initlength;
}

void encode_freq(rangecoder* rc, unsigned sy_f, unsigned lt_f, unsigned tot_f)
{
    enc_normalize(rc);
    const unsigned r = rc->range / tot_f, tmp = r * lt_f;
    rc->low += tmp;
    rc->range -= tmp;
    if (lt_f + sy_f < tot_f) rc->range = r * sy_f;
}

void encode_shift(rangecoder* rc, unsigned sy_f, unsigned lt_f, unsigned shift)
{
    enc_normalize(rc);
    const unsigned r = rc->range >> shift, tmp = r * lt_f;
    rc->low += tmp;
    if ((lt_f + sy_f) >> shift) rc->range -= tmp; else rc->range = r * sy_f;
}

unsigned done_encoding(rangecoder* rc)
{
    enc_normalize(rc);
    rc->bytecount += 5;
    unsigned tmp = rc->low >> kShift;
    if (!((rc->low & (kBottom - 1)) < ((rc->bytecount & 0xffffffu) >> 1))) tmp += 1;
    if (tmp > 0xff) {
        put(rc, rc->buffer + 1u);
        for (; rc->help; rc->help--) put(rc, 0);
    } else {
        put(rc, rc->buffer);
        for (; rc->help; rc->help--) put(rc, 0xff);
    }
    put(rc, tmp & 0xff);
    put(rc, (rc->bytecount >> 16) & 0xff);
    put(rc, (rc->bytecount >> 8) & 0xff);
    put(rc, rc->bytecount & 0xff);
    return rc->bytecount;
}

int start_decoding(rangecoder* rc)
{
    const int c = (int)get(rc);
    rc->buffer = (unsigned char)get(rc);
    rc->low = rc->buffer >> (8 - kExtra);
    rc->range = 1u << kExtra;
    return c;
}

unsigned decode_culfreq(rangecoder* rc, unsigned tot_f)
{
    dec_normalize(rc);
    rc->help = rc->range / tot_f;
    const unsigned tmp = rc->low / rc->help;
    return tmp >= tot_f ? tot_f - 1 : tmp;
}

unsigned decode_culshift(rangecoder* rc, unsigned shift)
{
    dec_normalize(rc);
    rc->help = rc->range >> shift;
    const unsigned tmp = rc->low / rc->help;
    return (tmp >> shift) ? (1u << shift) - 1 : tmp;
}

void decode_update(rangecoder* rc, unsigned sy_f, unsigned lt_f, unsigned tot_f)
{
    const unsigned tmp = rc->help * lt_f;
    rc->low -= tmp;
    if (lt_f + sy_f < tot_f) rc->range = rc->help * sy_f; else rc->range -= tmp;
}

unsigned char decode_byte(rangecoder* rc)
{
    const unsigned char tmp = (unsigned char)decode_culshift(rc, 8);
    decode_update(rc, 1, tmp, 1u << 8);
    return tmp;
}

unsigned short decode_short(rangecoder* rc)
{
    const unsigned short tmp = (unsigned short)decode_culshift(rc, 16);
    decode_update(rc, 1, tmp, 1u << 16);
    return tmp;
}

void done_decoding(rangecoder* rc) { dec_normalize(rc); }

void init_databuf(rangecoder* rc, unsigned long maxlen)
{
    rc->datalen = maxlen;
    rc->datapos = 0;
    rc->databuf = (unsigned char*)calloc(maxlen ? maxlen : 1, 1);
}

void free_databuf(rangecoder* rc) { free(rc->databuf); rc->databuf = nullptr; }

void countblock(int* buffer, unsigned length, unsigned* counters)
{
    for (int i = 0; i < 257; i++) counters[i] = 0;
    for (unsigned i = 0; i < length; i++) counters[buffer[i]]++;
}

void readcounts(rangecoder* rc, unsigned* counters)
{
    for (int i = 0; i < 256; i++) counters[i] = decode_short(rc);
}

void ind_p2w_3d(int lvlin, int n1, int n2, int n3, int i1in, int i2in, int i3in, int* lvl, int* i1, int* i2, int* i3)
{
    int c1 = n1, c2 = n2, c3 = n3, touched = 0;
    *lvl = 0; *i1 = i1in; *i2 = i2in; *i3 = i3in;
    for (int k = 0; k < lvlin; k++) {
        const int m1 = c1 / 2 + (c1 % 2 > 0), m2 = c2 / 2 + (c2 % 2 > 0), m3 = c3 / 2 + (c3 % 2 > 0);
        if (c1 > 1 && *i3 < c3 && *i2 < c2 && *i1 < c1) { *i1 = (*i1 % 2) ? *i1 / 2 + m1 : *i1 / 2; touched = 1; }
        if (c2 > 1 && *i3 < c3 && *i2 < c2 && *i1 < c1) { *i2 = (*i2 % 2) ? *i2 / 2 + m2 : *i2 / 2; touched = 1; }
        if (c3 > 1 && *i3 < c3 && *i2 < c2 && *i1 < c1) { *i3 = (*i3 % 2) ? *i3 / 2 + m3 : *i3 / 2; touched = 1; }
        c1 = m1; c2 = m2; c3 = m3;
        if (touched) *lvl += 1;
    }
}

}  // extern "C"
