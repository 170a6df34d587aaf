// DEFLATE (RFC 1951) decoder for one raw stream, written for one GPU thread per stream: BGZF files (bgzip, BAM) are a
// sequence of independent <= 64 KiB deflate streams, so a file inflates with one thread per block (k_inflate in
// mlst_engine.hip).  The same code compiles for the host, where tests/test_inflate.py checks it against zlib through
// mlst_selftest_inflate (the decoder's own test hook; the product path runs it on the device only).
//
// Plain and small on purpose: canonical Huffman codes decoded bit by bit from (count per length, symbols in code order)
// tables -- 700 bytes per stream (LDS in k_inflate), no look-up tables to build, every read and write bounds-checked,
// every loop consumes input or output, so a corrupt block ends in an error code, never in a fault or a hang.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MLST_HD __host__ __device__
#else
#define MLST_HD
#endif

namespace mlst_inflate {

enum { OK = 0, E_INPUT = -1, E_BLOCKTYPE = -2, E_STORED = -3, E_LENGTHS = -4, E_OUTPUT = -5, E_DISTANCE = -6, E_SYMBOL = -7, E_SHORT = -8 };

struct Bits {
    const uint8_t* in; uint64_t n, pos;
    uint64_t buf; int cnt;
};
// n <= 16 bits, least significant bit first.  Refills take eight bytes with one batch of independent loads (a thread of
// k_inflate waits a full memory latency for every dependent load): the bytes are ORed in above the cnt valid bits, the
// position moves by the whole bytes that fit, and a byte that was ORed in only in part is ORed in again by the next
// refill -- with the same bits, so nothing has to be masked.
MLST_HD inline int take(Bits& b, int n, uint32_t& out) {
    if (b.cnt < n) {
        if (b.pos + 8 <= b.n) {
            uint64_t w = 0;
            for (int k = 0; k < 8; k++) w |= (uint64_t)b.in[b.pos + k] << (8 * k);
            b.buf |= w << b.cnt;
            b.pos += (uint64_t)((63 - b.cnt) >> 3);
            b.cnt |= 56;
        } else {
            while (b.cnt <= 56 && b.pos < b.n) { b.buf |= (uint64_t)b.in[b.pos++] << b.cnt; b.cnt += 8; }
            if (b.cnt < n) return E_INPUT;
        }
    }
    out = (uint32_t)(b.buf & ((1ull << n) - 1ull)); b.buf >>= n; b.cnt -= n;
    return OK;
}

struct Huff { uint16_t count[16]; uint16_t symbol[288]; };      // literal / length code (also the 19-symbol code-length code)
struct HuffD { uint16_t count[16]; uint16_t symbol[30]; };      // distance code
struct Tables { Huff lc; HuffD dc; };                           // 700 bytes per stream: LDS in k_inflate, the stack on the host

// canonical code from code lengths; returns 0 for a complete code, > 0 for an incomplete one, < 0 for an over-subscribed one
template <typename H>
MLST_HD inline int build(H& h, const uint16_t* length, int n) {
    for (int l = 0; l <= 15; l++) h.count[l] = 0;
    for (int s = 0; s < n; s++) h.count[length[s]]++;
    if (h.count[0] == n) return 0;                 // no codes: complete, but decoding anything with it fails
    int left = 1;
    for (int l = 1; l <= 15; l++) { left <<= 1; left -= h.count[l]; if (left < 0) return left; }
    uint16_t offs[16]; offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + h.count[l]);
    for (int s = 0; s < n; s++) if (length[s] != 0) h.symbol[offs[length[s]]++] = (uint16_t)s;
    return left;
}
template <typename H>
MLST_HD inline int decode(Bits& b, const H& h) {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; l++) {
        uint32_t bit; if (take(b, 1, bit) != OK) return E_INPUT;
        code |= (int)bit;
        const int count = h.count[l];
        if (code - count < first) return h.symbol[index + (code - first)];
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return E_SYMBOL;
}

// literal / length and distance codes until the end-of-block symbol
MLST_HD inline int codes(Bits& b, const Huff& lc, const HuffD& dc, uint8_t* out, uint64_t cap, uint64_t& op) {
    const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
        int sym = decode(b, lc);
        if (sym < 0) return sym;
        if (sym < 256) {
            if (op >= cap) return E_OUTPUT;
            out[op++] = (uint8_t)sym;
        } else if (sym == 256) return OK;
        else {
            sym -= 257;
            if (sym >= 29) return E_SYMBOL;
            uint32_t x; int rc = take(b, lext[sym], x); if (rc != OK) return rc;
            const uint32_t len = lbase[sym] + x;
            const int ds = decode(b, dc);
            if (ds < 0) return ds;
            if (ds >= 30) return E_SYMBOL;
            rc = take(b, dext[ds] > 8 ? 8 : dext[ds], x); if (rc != OK) return rc;       // up to 13 extra bits: two takes of <= 8
            uint32_t dist = x;
            if (dext[ds] > 8) { rc = take(b, dext[ds] - 8, x); if (rc != OK) return rc; dist |= x << 8; }
            dist += dbase[ds];
            if ((uint64_t)dist > op) return E_DISTANCE;
            if (op + len > cap) return E_OUTPUT;
            uint32_t left = len;
            if (dist >= 8) {          // source and destination of a group of eight do not overlap: eight loads in one batch
                while (left >= 8) {
                    uint8_t t[8];
                    for (int k = 0; k < 8; k++) t[k] = out[op - dist + k];
                    for (int k = 0; k < 8; k++) out[op + k] = t[k];
                    op += 8; left -= 8;
                }
            }
            while (left--) { out[op] = out[op - dist]; op++; }
        }
    }
}

// one raw deflate stream -> out[0 .. cap); *produced = bytes written.  The stream has to fill exactly what the caller
// expects only if it says so: the caller compares *produced with the size it knows (ISIZE of the BGZF block).
MLST_HD inline int inflate_raw(const uint8_t* in, uint64_t n_in, uint8_t* out, uint64_t cap, uint64_t* produced, Tables* tb) {
    Bits b; b.in = in; b.n = n_in; b.pos = 0; b.buf = 0; b.cnt = 0;
    uint64_t op = 0;
    Huff& lc = tb->lc; HuffD& dc = tb->dc;
    uint16_t lengths[320];
    for (;;) {
        uint32_t last, type; int rc;
        if ((rc = take(b, 1, last)) != OK || (rc = take(b, 2, type)) != OK) { *produced = op; return rc; }
        if (type == 0) {
            b.pos -= (uint64_t)(b.cnt >> 3); b.buf = 0; b.cnt = 0;   // stored: back to the byte boundary (whole bytes still in the buffer are unread)
            if (b.pos + 4 > b.n) { *produced = op; return E_INPUT; }
            const uint32_t len = (uint32_t)b.in[b.pos] | ((uint32_t)b.in[b.pos + 1] << 8);
            const uint32_t nlen = (uint32_t)b.in[b.pos + 2] | ((uint32_t)b.in[b.pos + 3] << 8);
            b.pos += 4;
            if (len != (~nlen & 0xFFFFu)) { *produced = op; return E_STORED; }
            if (b.pos + len > b.n) { *produced = op; return E_INPUT; }
            if (op + len > cap) { *produced = op; return E_OUTPUT; }
            for (uint32_t k = 0; k < len; k++) out[op++] = b.in[b.pos++];
        } else if (type == 1) {
            for (int s = 0; s < 144; s++) lengths[s] = 8;
            for (int s = 144; s < 256; s++) lengths[s] = 9;
            for (int s = 256; s < 280; s++) lengths[s] = 7;
            for (int s = 280; s < 288; s++) lengths[s] = 8;
            build(lc, lengths, 288);
            for (int s = 0; s < 30; s++) lengths[s] = 5;
            build(dc, lengths, 30);
            rc = codes(b, lc, dc, out, cap, op);
            if (rc != OK) { *produced = op; return rc; }
        } else if (type == 2) {
            const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint32_t nlen, ndist, ncode;
            if ((rc = take(b, 5, nlen)) != OK || (rc = take(b, 5, ndist)) != OK || (rc = take(b, 4, ncode)) != OK) { *produced = op; return rc; }
            nlen += 257; ndist += 1; ncode += 4;
            if (nlen > 286 || ndist > 30) { *produced = op; return E_LENGTHS; }
            for (int i = 0; i < 19; i++) lengths[i] = 0;
            for (uint32_t i = 0; i < ncode; i++) { uint32_t x; if ((rc = take(b, 3, x)) != OK) { *produced = op; return rc; } lengths[order[i]] = (uint16_t)x; }
            if (build(lc, lengths, 19) != 0) { *produced = op; return E_LENGTHS; }      // the code-length code must be complete
            uint32_t idx = 0;
            while (idx < nlen + ndist) {
                int sym = decode(b, lc);
                if (sym < 0) { *produced = op; return sym; }
                if (sym < 16) lengths[idx++] = (uint16_t)sym;
                else {
                    uint32_t rep, x; uint16_t val = 0;
                    if (sym == 16) {
                        if (idx == 0) { *produced = op; return E_LENGTHS; }
                        val = lengths[idx - 1];
                        if ((rc = take(b, 2, x)) != OK) { *produced = op; return rc; }
                        rep = 3 + x;
                    } else if (sym == 17) { if ((rc = take(b, 3, x)) != OK) { *produced = op; return rc; } rep = 3 + x; }
                    else { if ((rc = take(b, 7, x)) != OK) { *produced = op; return rc; } rep = 11 + x; }
                    if (idx + rep > nlen + ndist) { *produced = op; return E_LENGTHS; }
                    while (rep--) lengths[idx++] = val;
                }
            }
            if (lengths[256] == 0) { *produced = op; return E_LENGTHS; }                 // no end-of-block code
            int e = build(lc, lengths, (int)nlen);
            if (e < 0 || (e > 0 && nlen != (uint32_t)(lc.count[0] + lc.count[1]))) { *produced = op; return E_LENGTHS; }
            e = build(dc, lengths + nlen, (int)ndist);
            if (e < 0 || (e > 0 && ndist != (uint32_t)(dc.count[0] + dc.count[1]))) { *produced = op; return E_LENGTHS; }
            rc = codes(b, lc, dc, out, cap, op);
            if (rc != OK) { *produced = op; return rc; }
        } else { *produced = op; return E_BLOCKTYPE; }
        if (last) break;
    }
    *produced = op;
    return OK;
}

}  // namespace mlst_inflate
