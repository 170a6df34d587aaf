// DEFLATE (RFC 1951) decoder for one raw stream.  BGZF files (bgzip, BAM) are a sequence of independent <= 64 KiB
// deflate streams, so a file inflates with one GPU wave per block (k_inflate in mlst_engine.hip: every lane runs this
// code on the same stream, the output policy spreads the copies over the lanes).  The same code compiles for the host,
// where tests/test_inflate.py checks it against zlib through mlst_selftest_inflate (the decoder's own test hook; the
// product path runs it on the device only).
//
// Canonical Huffman codes as (count per length, symbols in code order) + a look-up table over the next 8 (6) bits;
// 1.3 KB of tables per stream, every read and write bounds-checked, every loop consumes input or output, so a corrupt
// block ends in an error code, never in a fault or a hang.
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
#if defined(__HIP_DEVICE_COMPILE__)
    // one lane per stream (inflate_lane.h): a window of WIN bytes of the stream in LDS, refilled WIN bytes at a time by WIN / 16
    // independent 16-byte loads (one memory round trip per ~50 symbols instead of one per three).  128 bytes since round 5
    // (256 before): k_inflate_tok then leaves 10 KB of a CU's LDS to the kernels that run beside it (with 2 KB left the
    // parser's workgroups, 16 bytes of LDS each, found room for two per CU and took 2.0 ms instead of 0.3).
    enum { WIN = 128 };
    __attribute__((address_space(3))) uint64_t* win;      // (an LDS pointer by type: a generic one makes every read a FLAT load, which waits for the lane's outstanding global stores)
                                 // element j of this lane's window at win[j * 64] (lane-interleaved); nullptr: no window
    const uint8_t* win_at;       // the 16-byte aligned address the window starts at
    const uint8_t* buf_end;      // end of the (padded) compressed buffer: no load reaches beyond it
#endif
};
// n <= 16 bits, least significant bit first.  Refills take eight bytes with one batch of independent loads (a thread of
// k_inflate waits a full memory latency for every dependent load): the bytes are ORed in above the cnt valid bits, the
// position moves by the whole bytes that fit, and a byte that was ORed in only in part is ORed in again by the next
// refill -- with the same bits, so nothing has to be masked.
MLST_HD inline void refill(Bits& b) {
    if (b.pos + 8 <= b.n) {
        uint64_t w = 0;
#if defined(__HIP_DEVICE_COMPILE__)
        // (device: eight byte loads per lane are 512 requests per wave; the eight bytes as two aligned 8-byte words and a funnel
        // shift.  The second word may lie up to 7 bytes behind the stream: the engine's compressed buffers end with 16 bytes of
        // padding.)
        const uint8_t* p = b.in + b.pos;
        uint64_t w0, w1; unsigned sh;
        {   // (device code always brings a window -- inflate_lane.h is its only user; a second path that loads from global memory
            // here made the compiler wait at the join for EVERYTHING the lane had in flight, its token stores included)
            const bool need = p + 16 > b.win_at + Bits::WIN || p < b.win_at;
            if (__ballot(need)) {                      // every lane that is here moves its window up: the waits coincide
                const uint8_t* a0 = p - ((uintptr_t)p & 15u);
                typedef unsigned int v4 __attribute__((ext_vector_type(4)));
                v4 v[Bits::WIN / 16];
                #pragma unroll
                for (int j = 0; j < Bits::WIN / 16; j++) {
                    const uint8_t* a = a0 + 16 * j;
                    if (a + 16 > b.buf_end) a = b.buf_end - 16;      // (behind the buffer: bytes no code of the stream reaches)
                    v[j] = *reinterpret_cast<const v4*>(a);
                }
                #pragma unroll
                for (int j = 0; j < Bits::WIN / 16; j++) {
                    b.win[(2 * j) * 64] = (uint64_t)v[j].x | ((uint64_t)v[j].y << 32);
                    b.win[(2 * j + 1) * 64] = (uint64_t)v[j].z | ((uint64_t)v[j].w << 32);
                }
                b.win_at = a0;
            }
            const unsigned o = (unsigned)(p - b.win_at);
            w0 = b.win[(o >> 3) * 64]; w1 = b.win[((o >> 3) + 1) * 64];
            sh = (o & 7u) * 8u;
        }
        w = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0;
#else
        for (int k = 0; k < 8; k++) w |= (uint64_t)b.in[b.pos + k] << (8 * k);
#endif
        b.buf |= w << b.cnt;
        b.pos += (uint64_t)((63 - b.cnt) >> 3);
        b.cnt |= 56;
    } else {
        while (b.cnt <= 56 && b.pos < b.n) { b.buf |= (uint64_t)b.in[b.pos++] << b.cnt; b.cnt += 8; }
    }
}
MLST_HD inline int take(Bits& b, int n, uint32_t& out) {
    if (b.cnt < n) { refill(b); if (b.cnt < n) return E_INPUT; }
    out = (uint32_t)(b.buf & ((1ull << n) - 1ull)); b.buf >>= n; b.cnt -= n;
    return OK;
}

// A code = (count per length, symbols in code order) + a look-up table over the next LB input bits for the codes of at
// most LB bits: entry = (symbol << SH) | length, 0 = longer code (decoded bit by bit).
// (table bits: with one lane per stream -- inflate_lane.h -- a code longer than the table sends its lane through the canonical walk
// while the other 63 wait, and with 64 streams some lane meets one in nearly every step at 8 / 6 bits: 9 / 8 bits are what 160 KB
// of LDS hold for 64 lanes beside the input windows)
struct Huff  { enum { LB = 9, SH = 4 }; uint16_t count[16]; uint16_t symbol[288]; uint16_t lut[1 << LB]; };   // literal / length code (also the code-length code)
struct HuffD { enum { LB = 8, SH = 4 }; uint16_t count[16]; uint16_t symbol[30]; uint16_t lut[1 << LB]; };    // distance code
struct Tables { Huff lc; HuffD dc; uint16_t offs[16]; };        // 2.2 KB per stream: LDS in k_inflate_tok, the stack on the host
// No array of the decoder lives on a thread's stack (round 5; k_inflate_tok had 656 bytes of scratch per lane): the code
// lengths of a block are kept in lc.lut until the literal / length code is built from them (build() reads every length before it
// writes the first table entry), the code-length code is decoded with dc's tables (19 symbols of at most 7 bits fit HuffD), and
// build()'s running offsets sit in Tables::offs.

// canonical code from code lengths; returns 0 for a complete code, > 0 for an incomplete one, < 0 for an over-subscribed one
// (length may be h.lut itself: every length is read before the first table entry is written)
template <typename H>
MLST_HD inline int build(H& h, const uint16_t* length, int n, uint16_t* offs) {
    for (int l = 0; l <= 15; l++) h.count[l] = 0;
    for (int s = 0; s < n; s++) h.count[length[s]]++;
    int left = 1;
    if (h.count[0] != n) for (int l = 1; l <= 15; l++) { left <<= 1; left -= h.count[l]; if (left < 0) break; }
    if (h.count[0] == n || left < 0) {             // no codes (complete, but decoding anything with it fails) or an over-subscribed set
        const int none = h.count[0] == n;
        for (int i = 0; i < (1 << H::LB); i++) h.lut[i] = 0;
        return none ? 0 : left;
    }
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + h.count[l]);
    for (int s = 0; s < n; s++) if (length[s] != 0) h.symbol[offs[length[s]]++] = (uint16_t)s;
    for (int i = 0; i < (1 << H::LB); i++) h.lut[i] = 0;
    // look-up table: the canonical code of the j-th symbol of length l is first(l) + j, sent most significant bit first
    int code = 0, idx = 0;
    for (int l = 1; l <= H::LB; l++) {
        for (int j = 0; j < h.count[l]; j++, idx++) {
            const int c = code + j; int rev = 0;
            for (int k = 0; k < l; k++) rev |= ((c >> k) & 1) << (l - 1 - k);
            const unsigned entry = ((unsigned)h.symbol[idx] << H::SH) | (unsigned)l;
            for (int i = rev; i < (1 << H::LB); i += 1 << l) h.lut[i] = entry;
        }
        code = (code + h.count[l]) << 1;
    }
    return left;
}
template <typename H>
MLST_HD inline int decode_slow(Bits& b, const H& h) {
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

template <typename H>
MLST_HD inline int decode(Bits& b, const H& h) {
    if (b.cnt < 15) refill(b);
    const uint32_t e = h.lut[b.buf & ((1u << H::LB) - 1u)];
    if (e) {
        const int l = (int)(e & ((1u << H::SH) - 1u));
        if (l > b.cnt) return E_INPUT;
        b.buf >>= l; b.cnt -= l;
        return (int)(e >> H::SH);
    }
    return decode_slow(b, h);
}

// Where the decoded bytes go.  OutSerial: one thread, one byte at a time (the host, the tests).  The engine's k_inflate
// runs one WAVE per stream -- every lane decodes the same symbols, so the wave never diverges -- with its own policy
// whose copies are spread over the lanes (mlst_engine.hip).  put / copy / raw are only called with room checked.
struct OutSerial {
    uint8_t* out; uint64_t op;
    MLST_HD void put(uint8_t c) { out[op++] = c; }
    MLST_HD void copy(uint32_t dist, uint32_t len) { for (uint32_t k = 0; k < len; k++) { out[op] = out[op - dist]; op++; } }
    MLST_HD void raw(const uint8_t* src, uint32_t len) { for (uint32_t k = 0; k < len; k++) out[op++] = src[k]; }
};

// literal / length and distance codes until the end-of-block symbol
template <typename Out>
MLST_HD inline int codes(Bits& b, const Huff& lc, const HuffD& dc, Out& o, uint64_t cap) {
    // base values and extra-bit counts of the length and distance symbols are computed, not looked up: a table in
    // memory would be one more dependent load per match
    for (;;) {
        int sym = decode(b, lc);
        if (sym < 0) return sym;
        if (sym < 256) {
            if (o.op >= cap) return E_OUTPUT;
            o.put((uint8_t)sym);
        } else if (sym == 256) return OK;
        else {
            sym -= 257;
            if (sym >= 29) return E_SYMBOL;
            const int le = sym < 8 || sym == 28 ? 0 : (sym - 4) >> 2;
            const uint32_t lb = sym < 8 ? 3u + (uint32_t)sym : (sym == 28 ? 258u : ((4u + ((uint32_t)sym & 3u)) << le) + 3u);
            uint32_t x; int rc = take(b, le, x); if (rc != OK) return rc;
            const uint32_t len = lb + x;
            const int ds = decode(b, dc);
            if (ds < 0) return ds;
            if (ds >= 30) return E_SYMBOL;
            const int de = ds < 4 ? 0 : (ds - 2) >> 1;
            const uint32_t db = ds < 4 ? 1u + (uint32_t)ds : ((2u + ((uint32_t)ds & 1u)) << de) + 1u;
            rc = take(b, de, x); if (rc != OK) return rc;                                // up to 13 extra bits
            const uint32_t dist = db + x;
            if ((uint64_t)dist > o.op) return E_DISTANCE;
            if (o.op + len > cap) return E_OUTPUT;
            o.copy(dist, len);
        }
    }
}

// one raw deflate stream -> at most cap bytes through o (o.op = bytes written, also after an error).  Whether the stream
// filled what the caller expected is the caller's check (ISIZE of the BGZF block).
// OWN_CODES: the output policy brings its own symbol loop (Out::codes, same contract as codes() above) -- inflate_lane.h's
// loop for one lane per stream, laid out so that the lanes of a wave walk few different paths
template <typename Out, bool OWN_CODES = false>
MLST_HD inline int inflate_stream(const uint8_t* in, uint64_t n_in, Out& o, uint64_t cap, Tables* tb, uint64_t* dev_win = nullptr, const uint8_t* dev_buf_end = nullptr) {
    Bits b; b.in = in; b.n = n_in; b.pos = 0; b.buf = 0; b.cnt = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    b.win = (__attribute__((address_space(3))) uint64_t*)dev_win; b.win_at = nullptr; b.buf_end = dev_buf_end;
#else
    (void)dev_win; (void)dev_buf_end;
#endif
    uint64_t& op = o.op;
    Huff& lc = tb->lc; HuffD& dc = tb->dc;
    uint16_t* const lengths = lc.lut;                  // 512 entries; at most 320 lengths (see Tables)
    uint16_t* const offs = tb->offs;
    for (;;) {
        uint32_t last, type; int rc;
        if ((rc = take(b, 1, last)) != OK || (rc = take(b, 2, type)) != OK) return rc;
        if (type == 0) {
            b.pos -= (uint64_t)(b.cnt >> 3); b.buf = 0; b.cnt = 0;   // stored: back to the byte boundary (whole bytes still in the buffer are unread)
            if (b.pos + 4 > b.n) return E_INPUT;
            const uint32_t len = (uint32_t)b.in[b.pos] | ((uint32_t)b.in[b.pos + 1] << 8);
            const uint32_t nlen = (uint32_t)b.in[b.pos + 2] | ((uint32_t)b.in[b.pos + 3] << 8);
            b.pos += 4;
            if (len != (~nlen & 0xFFFFu)) return E_STORED;
            if (b.pos + len > b.n) return E_INPUT;
            if (op + len > cap) return E_OUTPUT;
            o.raw(b.in + b.pos, len); b.pos += len;
        } else if (type == 1) {
            for (int s = 0; s < 30; s++) lengths[s] = 5;
            build(dc, lengths, 30, offs);
            for (int s = 0; s < 144; s++) lengths[s] = 8;
            for (int s = 144; s < 256; s++) lengths[s] = 9;
            for (int s = 256; s < 280; s++) lengths[s] = 7;
            for (int s = 280; s < 288; s++) lengths[s] = 8;
            build(lc, lengths, 288, offs);
            if constexpr (OWN_CODES) rc = Out::codes(b, lc, dc, o, cap); else rc = codes(b, lc, dc, o, cap);
            if (rc != OK) return rc;
        } else if (type == 2) {
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint32_t nlen, ndist, ncode;
            if ((rc = take(b, 5, nlen)) != OK || (rc = take(b, 5, ndist)) != OK || (rc = take(b, 4, ncode)) != OK) return rc;
            nlen += 257; ndist += 1; ncode += 4;
            if (nlen > 286 || ndist > 30) return E_LENGTHS;
            for (int i = 0; i < 19; i++) lengths[i] = 0;
            for (uint32_t i = 0; i < ncode; i++) { uint32_t x; if ((rc = take(b, 3, x)) != OK) return rc; lengths[order[i]] = (uint16_t)x; }
            if (build(dc, lengths, 19, offs) != 0) return E_LENGTHS;      // the code-length code (in dc's tables) must be complete
            uint32_t idx = 0;
            while (idx < nlen + ndist) {
                int sym = decode(b, dc);
                if (sym < 0) return sym;
                if (sym < 16) lengths[idx++] = (uint16_t)sym;
                else {
                    uint32_t rep, x; uint16_t val = 0;
                    if (sym == 16) {
                        if (idx == 0) return E_LENGTHS;
                        val = lengths[idx - 1];
                        if ((rc = take(b, 2, x)) != OK) return rc;
                        rep = 3 + x;
                    } else if (sym == 17) { if ((rc = take(b, 3, x)) != OK) return rc; rep = 3 + x; }
                    else { if ((rc = take(b, 7, x)) != OK) return rc; rep = 11 + x; }
                    if (idx + rep > nlen + ndist) return E_LENGTHS;
                    while (rep--) lengths[idx++] = val;
                }
            }
            if (lengths[256] == 0) return E_LENGTHS;                 // no end-of-block code
            int e = build(dc, lengths + nlen, (int)ndist, offs);      // (the distance code first: building lc overwrites the lengths)
            if (e < 0 || (e > 0 && ndist != (uint32_t)(dc.count[0] + dc.count[1]))) return E_LENGTHS;
            e = build(lc, lengths, (int)nlen, offs);
            if (e < 0 || (e > 0 && nlen != (uint32_t)(lc.count[0] + lc.count[1]))) return E_LENGTHS;
            if constexpr (OWN_CODES) rc = Out::codes(b, lc, dc, o, cap); else rc = codes(b, lc, dc, o, cap);
            if (rc != OK) return rc;
        } else return E_BLOCKTYPE;
        if (last) break;
    }
    return OK;
}

MLST_HD inline int inflate_raw(const uint8_t* in, uint64_t n_in, uint8_t* out, uint64_t cap, uint64_t* produced, Tables* tb) {
    OutSerial o; o.out = out; o.op = 0;
    const int rc = inflate_stream(in, n_in, o, cap, tb);
    *produced = o.op;
    return rc;
}

}  // namespace mlst_inflate
