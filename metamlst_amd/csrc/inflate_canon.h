// inflate_canon.h -- phase 1 of the two-kernel inflate with 576 bytes of state per stream (round 5; 800 in its first version).
//
// inflate_lane.h's k_inflate_tok decodes one BGZF block per LANE with look-up tables of 9 / 8 bits: 2.3 KB per stream, 64
// streams = one wave fill a CU's LDS, and the kernel is bound by the latency of that ONE wave per CU (three SIMDs of four
// idle; a piece of 16,384 blocks takes 4.4 ms however few of the GPU's lanes it uses).  Blocks in flight = LDS / bytes per
// stream, so the tables go: a canonical Huffman code is decoded from the next 15 stream bits P (first bit most
// significant) with 15 left-justified LIMITS -- limit[l] = (first code of length l + codes of length l) << (15 - l), non-
// decreasing in l -- by counting the limits that P has reached (length = 1 + count; no table look-up, no long-code walk, one
// straight line for every lane), the symbol index is base[length] + (P >> (15 - length)), and the symbols sit in code order.
// Per stream: two codes x (16 limits + 16 bases) x 2 B, 192 + 32 symbols x 1 B, the code lengths of a block as nibbles
// (160 B, only while its tables are built) and a 64-byte window of the stream: 144 words.  Four waves per CU instead of one.
// A block whose literal / length code has more than 192 symbols in use (text never does; binary data may) is flagged like a
// block with too many tokens and left to inflate_wave.h's kernel.
//
// The decoder is written once over two small policies -- where a stream's 200 words live (host: an array; device: LDS,
// interleaved by lane, so that the same index in every lane is conflict-free and any index is a plain 32-bit access) and
// where the stream's bytes come from -- so that tests/test_inflate.py runs the very same code on the host against zlib
// (mlst_selftest_inflate_canon) before the device kernel is trusted.
#pragma once
#include <stdint.h>
#include "inflate_dev.h"

namespace inflate_canon {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;
typedef unsigned char u8;

enum : u32 { NSYM_L = 192, NSYM_D = 32, WIN = 64 };
// Symbols are kept as BYTES (second half of round 5; 16 bits each before: 800 bytes per stream, three waves per CU): a literal below
// 224 as itself, the end-of-block and length symbols 256 .. 287 as 224 .. 255; a block that uses a literal of 224 or more (text
// never does) is left to the other kernel like one with more than NSYM_L symbols.  144 words = 576 bytes per stream: FOUR waves per
// CU, one per SIMD.
enum : u32 { SYM_ESC = 224 };
enum : u32 { W_LIM_L = 0, W_BASE_L = 8, W_SYM_L = 16, W_LIM_D = W_SYM_L + NSYM_L / 4, W_BASE_D = W_LIM_D + 8, W_SYM_D = W_BASE_D + 8,
             W_LEN = W_SYM_D + NSYM_D / 4, W_WIN = W_LEN + 40, W_TOTAL = W_WIN + WIN / 4 };      // 32-bit words: 144 = 576 bytes
enum : u32 { TAG_LIT = 0u, TAG_RAW = 1u, TAG_MATCH = 2u, TAG_OPERAND = 3u };      // token tags of inflate_lane.h

#if defined(__HIP_DEVICE_COMPILE__)
#define CANON_BREV32(x) __brev(x)
#else
static inline u32 canon_brev32_(u32 x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1); x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4); x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
}
#define CANON_BREV32(x) canon_brev32_(x)
#endif

// ---- 16-bit and 4-bit elements of a stream's words
template <class M> MLST_HD inline u32 ld16(const M& m, u32 arr, u32 i) { return (m.ld(arr + (i >> 1)) >> (16u * (i & 1u))) & 0xFFFFu; }
template <class M> MLST_HD inline void st16(M& m, u32 arr, u32 i, u32 v) {
    const u32 d = arr + (i >> 1), sh = 16u * (i & 1u); m.st(d, (m.ld(d) & ~(0xFFFFu << sh)) | ((v & 0xFFFFu) << sh));
}
template <class M> MLST_HD inline u32 ld8(const M& m, u32 arr, u32 i) { return (m.ld(arr + (i >> 2)) >> (8u * (i & 3u))) & 0xFFu; }
template <class M> MLST_HD inline void st8(M& m, u32 arr, u32 i, u32 v) {
    const u32 d = arr + (i >> 2), sh = 8u * (i & 3u); m.st(d, (m.ld(d) & ~(0xFFu << sh)) | ((v & 0xFFu) << sh));
}
template <class M> MLST_HD inline u32 ldn(const M& m, u32 i) { return (m.ld(W_LEN + (i >> 3)) >> (4u * (i & 7u))) & 15u; }
template <class M> MLST_HD inline void stn(M& m, u32 i, u32 v) {
    const u32 d = W_LEN + (i >> 3), sh = 4u * (i & 7u); m.st(d, (m.ld(d) & ~(15u << sh)) | ((v & 15u) << sh));
}

// ---- canonical code from the nibble lengths [start, start + n): limits, bases, symbols in code order.
// returns 0 for a complete code, > 0 for an incomplete one, < 0 for an over-subscribed one; *zeros = symbols without a code,
// *ones = codes of one bit; over |= more symbols in use than the table holds
template <class M>
MLST_HD inline int build(M& m, const u32 w_lim, const u32 w_base, const u32 w_sym, const u32 cap, const u32 start, const u32 n, u32* zeros, u32* ones, bool& over) {
    for (u32 j = 0; j < 8; j++) m.st(w_lim + j, 0u);
    for (u32 s = 0; s < n; s++) { const u32 l = ldn(m, start + s); st16(m, w_lim, l, ld16(m, w_lim, l) + 1u); }
    u32 cnt[16];
    #pragma unroll
    for (u32 l = 0; l < 16; l++) cnt[l] = ld16(m, w_lim, l);
    *zeros = cnt[0]; *ones = cnt[1];
    int left = 1; bool neg = false;
    #pragma unroll
    for (u32 l = 1; l < 16; l++) { left = (left << 1) - (int)cnt[l]; neg = neg || left < 0; }
    if (cnt[0] == n) left = 0;                       // no codes: complete, and decoding anything with it fails (every limit is 0)
    u32 offs = 0;
    #pragma unroll
    for (u32 l = 1; l < 16; l++) { st16(m, w_base, l, offs); offs += cnt[l]; }
    if (offs > cap) over = true;
    if (!neg) for (u32 s = 0; s < n; s++) {
        const u32 l = ldn(m, start + s);
        if (l) {
            const u32 j = ld16(m, w_base, l); st16(m, w_base, l, j + 1u);
            if (s >= SYM_ESC && s < 256u) over = true;                               // a literal the byte encoding has no room for
            if (j < cap) st8(m, w_sym, j, s < 256u ? s : s - 256u + SYM_ESC);
        }
    }
    u32 first = 0; offs = 0;
    #pragma unroll
    for (u32 l = 1; l < 16; l++) {
        st16(m, w_lim, l, (first + cnt[l]) << (15u - l));
        st16(m, w_base, l, offs - first);            // (two's complement in 16 bits: base + code is taken modulo 2^16)
        offs += cnt[l]; first = (first + cnt[l]) << 1;
    }
    st16(m, w_lim, 0, 0u);
    return neg ? -1 : left;
}

// next symbol of a code from the next 15 stream bits P (first bit most significant); len = its code length.  < 0: no such code.
// (L: the code's 16 limits, two per word -- the symbol loop keeps them in registers for the length of a deflate block)
template <class M>
MLST_HD inline int decode_l(const M& m, const u32 (&L)[8], const u32 w_base, const u32 w_sym, const u32 cap, const u32 P, int& len) {
    // count of the limits P has reached, without a compare: limit - (P + 1) is negative exactly then, and the sign bits add up
    // (a compare per limit went through VCC: compare, hazard nop, select, add)
    const u32 p1 = P + 1u;
#if defined(__HIP_DEVICE_COMPILE__)
    // (device: two limits per instruction -- the words hold them as 16-bit pairs: packed subtract, packed shift, packed add.  All 16
    // limits go through; limit[0] = 0 always counts (p1 >= 1) and limit[15]'s turn is the `none` test below: both taken off again.)
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const us2 pp = {(unsigned short)p1, (unsigned short)p1};
    us2 acc = {0, 0};
    #pragma unroll
    for (u32 j = 0; j < 8; j++) { const us2 d = __builtin_bit_cast(us2, L[j]) - pp; acc += d >> (unsigned short)15; }
    const u32 c = (u32)acc.x + (u32)acc.y - 1u - ((((L[7] >> 16) - p1) >> 31) & 1u);
#else
    u32 c = 0;
    for (u32 l = 1; l < 15; l++) c += (((L[l >> 1] >> (16u * (l & 1u))) & 0xFFFFu) - p1) >> 31;
#endif
    len = 1 + (int)c;
    const bool none = P >= (L[7] >> 16);               // beyond the last code of 15 bits: an incomplete code's gap
    const u32 idx = (ld16(m, w_base, (u32)len) + (P >> (15u - (u32)len))) & 0xFFFFu;
    const bool bad = none || idx >= cap;
    const u32 v = ld8(m, w_sym, bad ? 0u : idx);
    const u32 sym = v + (v >= SYM_ESC ? 256u - SYM_ESC : 0u);
    return bad ? (int)mlst_inflate::E_SYMBOL : (int)sym;
}
template <class M>
MLST_HD inline int decode(const M& m, const u32 w_lim, const u32 w_base, const u32 w_sym, const u32 cap, const u32 P, int& len) {
    u32 L[8];
    #pragma unroll
    for (u32 j = 0; j < 8; j++) L[j] = m.ld(w_lim + j);
    return decode_l(m, L, w_base, w_sym, cap, P, len);
}

// ---- the bits of a stream.  S supplies bytes: fetch8(pos) = the eight bytes at pos (pos + 8 <= n), byte(pos) = one.
template <class S>
struct Bits {
    S src; u64 buf; int cnt; u32 pos, n;
    MLST_HD void refill() {
        if (pos + 8u <= n) { const u64 w = src.fetch8(pos); buf |= w << cnt; pos += (u32)((63 - cnt) >> 3); cnt |= 56; }
        else { while (cnt <= 56 && pos < n) { buf |= (u64)src.byte(pos++) << cnt; cnt += 8; } }
    }
    MLST_HD int take(int k, u32& out) {      // k <= 16 bits, least significant bit first
        if (cnt < k) { refill(); if (cnt < k) return mlst_inflate::E_INPUT; }
        out = (u32)(buf & ((1ull << k) - 1ull)); buf >>= k; cnt -= k;
        return mlst_inflate::OK;
    }
};

// where the tokens go: tok[0 .. cap); over = more than cap
struct Tok {
    u32* tok; u32 nt, cap; bool over;
    MLST_HD void emit(u32 t) { if (nt < cap) tok[nt] = t; else over = true; nt++; }
};

// one raw deflate stream -> tokens.  *produced = bytes the tokens stand for.  o.over: the block is left to the other kernel
// (too many tokens, or a literal / length code with more symbols in use than the table holds).
template <class M, class S>
MLST_HD inline int tok_stream(M& m, Bits<S>& b, Tok& o, const u32 want, u32* produced) {
    using namespace mlst_inflate;
    u32 op = 0; int err = OK;
    *produced = 0;
    for (;;) {
        u32 last, type; int rc;
        if ((rc = b.take(1, last)) != OK || (rc = b.take(2, type)) != OK) return rc;
        if (type == 0) {
            b.pos -= (u32)(b.cnt >> 3); b.buf = 0; b.cnt = 0;   // stored: back to the byte boundary (whole bytes still in the buffer are unread)
            if (b.pos + 4u > b.n) return E_INPUT;
            const u32 len = (u32)b.src.byte(b.pos) | ((u32)b.src.byte(b.pos + 1) << 8), nlen = (u32)b.src.byte(b.pos + 2) | ((u32)b.src.byte(b.pos + 3) << 8);
            b.pos += 4;
            if (len != (~nlen & 0xFFFFu)) return E_STORED;
            if (b.pos + len > b.n) return E_INPUT;
            if (op + len > want) return E_OUTPUT;
            if (len) { o.emit((TAG_RAW << 30) | len); o.emit((TAG_OPERAND << 30) | b.pos); op += len; }
            b.pos += len;
        } else if (type == 1 || type == 2) {
            u32 nlen = 288, ndist = 30;
            if (type == 1) {
                for (u32 s = 0; s < 144; s++) stn(m, s, 8);
                for (u32 s = 144; s < 256; s++) stn(m, s, 9);
                for (u32 s = 256; s < 280; s++) stn(m, s, 7);
                for (u32 s = 280; s < 288; s++) stn(m, s, 8);
                for (u32 s = 288; s < 318; s++) stn(m, s, 5);
            } else {
                u32 ncode;
                if ((rc = b.take(5, nlen)) != OK || (rc = b.take(5, ndist)) != OK || (rc = b.take(4, ncode)) != OK) return rc;
                nlen += 257; ndist += 1; ncode += 4;
                if (nlen > 286 || ndist > 30) return E_LENGTHS;
                for (u32 i = 0; i < 19; i++) stn(m, i, 0);
                for (u32 i = 0; i < ncode; i++) {
                    u32 x; if ((rc = b.take(3, x)) != OK) return rc;
                    static const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                    stn(m, order[i], x);
                }
                u32 z, o1; bool ov = false;
                if (build(m, W_LIM_D, W_BASE_D, W_SYM_D, NSYM_D, 0, 19, &z, &o1, ov) != 0) return E_LENGTHS;      // the code-length code (in the distance code's tables) must be complete
                u32 idx = 0;
                while (idx < nlen + ndist) {
                    if (b.cnt < 24) b.refill();
                    int l; const int sym = decode(m, W_LIM_D, W_BASE_D, W_SYM_D, NSYM_D, CANON_BREV32((u32)b.buf) >> 17, l);
                    if (sym < 0) return sym;
                    if (l > b.cnt) return E_INPUT;
                    b.buf >>= l; b.cnt -= l;
                    if (sym < 16) stn(m, idx++, (u32)sym);
                    else {
                        u32 rep, x, val = 0;
                        if (sym == 16) {
                            if (idx == 0) return E_LENGTHS;
                            val = ldn(m, idx - 1);
                            if ((rc = b.take(2, x)) != OK) return rc;
                            rep = 3 + x;
                        } else if (sym == 17) { if ((rc = b.take(3, x)) != OK) return rc; rep = 3 + x; }
                        else { if ((rc = b.take(7, x)) != OK) return rc; rep = 11 + x; }
                        if (idx + rep > nlen + ndist) return E_LENGTHS;
                        while (rep--) stn(m, idx++, val);
                    }
                }
                if (ldn(m, 256) == 0) return E_LENGTHS;                 // no end-of-block code
            }
            u32 z, o1; bool ov = false;
            // (the fixed code's 30 distance codes of 5 bits are incomplete by definition: only a transmitted code is checked)
            int e = build(m, W_LIM_D, W_BASE_D, W_SYM_D, NSYM_D, nlen, ndist, &z, &o1, ov);
            if (type == 2 && (e < 0 || (e > 0 && ndist != z + o1))) return E_LENGTHS;
            e = build(m, W_LIM_L, W_BASE_L, W_SYM_L, NSYM_L, 0, nlen, &z, &o1, ov);
            if (type == 2 && (e < 0 || (e > 0 && nlen != z + o1))) return E_LENGTHS;
            if (ov) { o.over = true; *produced = want; return OK; }      // (the caller leaves the block to the other kernel)
            // ---- the symbol loop: one straight line per step (see inflate_lane.h's codes_simt for why); both codes' limits in registers
            u32 LL[8], LD[8];
            #pragma unroll
            for (u32 j = 0; j < 8; j++) { LL[j] = m.ld(W_LIM_L + j); LD[j] = m.ld(W_LIM_D + j); }
            for (;;) {
                if (b.cnt < 48) b.refill();
                int l; int sym = decode_l(m, LL, W_BASE_L, W_SYM_L, NSYM_L, CANON_BREV32((u32)b.buf) >> 17, l);
                err = sym < 0 ? sym : err; sym = sym < 0 ? 256 : sym;
                err = l > b.cnt ? (int)E_INPUT : err;
                b.buf >>= l; b.cnt -= l;
                u32 token = (u32)sym, n_out = 1;
                if (sym > 256) {
                    int ls = sym - 257;
                    err = ls >= 29 ? (int)E_SYMBOL : err; ls = ls >= 29 ? 0 : ls;
                    const int le = ls < 8 || ls == 28 ? 0 : (ls - 4) >> 2;
                    const u32 lb = ls < 8 ? 3u + (u32)ls : (ls == 28 ? 258u : ((4u + ((u32)ls & 3u)) << le) + 3u);
                    const u32 len = lb + ((u32)b.buf & ((1u << le) - 1u));
                    b.buf >>= le; b.cnt -= le;
                    int dl; int ds = decode_l(m, LD, W_BASE_D, W_SYM_D, NSYM_D, CANON_BREV32((u32)b.buf) >> 17, dl);
                    err = ds < 0 ? ds : err; ds = ds < 0 ? 0 : ds;
                    b.buf >>= dl; b.cnt -= dl;
                    err = ds >= 30 ? (int)E_SYMBOL : err; ds = ds >= 30 ? 0 : ds;
                    const int de = ds < 4 ? 0 : (ds - 2) >> 1;
                    const u32 db = ds < 4 ? 1u + (u32)ds : ((2u + ((u32)ds & 1u)) << de) + 1u;
                    const u32 dist = db + ((u32)b.buf & ((1u << de) - 1u));
                    b.buf >>= de; b.cnt -= de;
                    err = b.cnt < 0 ? (int)E_INPUT : err;                       // (the three fields were cut without a look at the count)
                    err = dist > op ? (int)E_DISTANCE : err;
                    token = (TAG_MATCH << 30) | ((len - 3u) << 16) | (dist - 1u); n_out = len;
                }
                if (sym == 256 || err != OK) break;
                if (op + n_out > want) { err = E_OUTPUT; break; }
                o.emit(token); op += n_out;
            }
            if (err != OK) return err;
        } else return E_BLOCKTYPE;
        if (last) break;
    }
    *produced = op;
    return OK;
}

// ---- host: a stream's words in an array, its bytes in memory; tokens replayed into bytes (the test hook's second half)
struct MemHost { u32 w[W_TOTAL]; u32 ld(u32 d) const { return w[d]; } void st(u32 d, u32 v) { w[d] = v; } };
struct SrcHost {
    const u8* in;
    u64 fetch8(u32 pos) const { u64 w = 0; for (int k = 0; k < 8; k++) w |= (u64)in[pos + k] << (8 * k); return w; }
    u8 byte(u32 pos) const { return in[pos]; }
};
inline int inflate_raw_host(const u8* in, u64 n_in, u8* out, u64 cap, u64* produced_out, bool* over_out) {
    using namespace mlst_inflate;
    *produced_out = 0; *over_out = false;
    if (n_in >= (1ull << 31) || cap > 65536) return E_INPUT;
    MemHost m; for (u32 i = 0; i < W_TOTAL; i++) m.w[i] = 0;
    Bits<SrcHost> b; b.src.in = in; b.buf = 0; b.cnt = 0; b.pos = 0; b.n = (u32)n_in;
    static thread_local u32 tokbuf[65536 + 8];
    Tok o; o.tok = tokbuf; o.nt = 0; o.cap = 65536 + 8; o.over = false;
    u32 produced = 0;
    const int rc = tok_stream(m, b, o, (u32)cap, &produced);
    if (rc != OK) return rc;
    if (o.over) { *over_out = true; return OK; }
    u64 op = 0;
    for (u32 i = 0; i < o.nt; i++) {
        const u32 t = tokbuf[i], tag = t >> 30;
        if (tag == TAG_LIT) { if (op >= cap) return E_OUTPUT; out[op++] = (u8)t; }
        else if (tag == TAG_MATCH) {
            const u32 len = ((t >> 16) & 0xFFu) + 3u, dist = (t & 0x7FFFu) + 1u;
            if (dist > op) return E_DISTANCE;
            if (op + len > cap) return E_OUTPUT;
            for (u32 k = 0; k < len; k++) { out[op] = out[op - dist]; op++; }
        } else if (tag == TAG_RAW) {
            const u32 len = t & 0xFFFFu;
            if (i + 1 >= o.nt || (tokbuf[i + 1] >> 30) != TAG_OPERAND) return E_STORED;
            const u32 src = tokbuf[i + 1] & 0x3FFFFFFFu;
            if ((u64)src + len > n_in || op + len > cap) return E_STORED;
            for (u32 k = 0; k < len; k++) out[op++] = in[src + k];
            i++;
        }
    }
    *produced_out = op;
    return OK;
}

#if defined(__HIP_DEVICE_COMPILE__)
// ---- device: words in LDS, word d of lane's stream at lds[d * 64 + lane]; bytes through a WIN-byte window of the stream in
// the same words, refilled WIN bytes at a time (four 16-byte loads; every lane that is in refill() when one needs it moves up)
struct MemLds {
    __attribute__((address_space(3))) u32* base;      // = lds + lane
    __device__ __attribute__((always_inline)) u32 ld(u32 d) const { return base[d * 64u]; }
    __device__ __attribute__((always_inline)) void st(u32 d, u32 v) { base[d * 64u] = v; }
};
struct SrcLds {
    __attribute__((address_space(3))) u32* win;       // = lds + W_WIN * 64 + lane
    const u8* in; const u8* buf_end; u32 win_at; bool have;
    __device__ __attribute__((always_inline)) u64 fetch8(u32 pos) {
        const bool need = !have || (pos - win_at) + 12u > WIN;      // (unsigned: a position in front of the window is a huge difference)
        if (__ballot(need)) {
            const u8* p = in + pos;
            const u8* a0 = p - ((uintptr_t)p & 15u);
            typedef unsigned int v4 __attribute__((ext_vector_type(4)));
            v4 v[WIN / 16];
            #pragma unroll
            for (u32 j = 0; j < WIN / 16; j++) {
                const u8* a = a0 + 16u * j;
                if (a + 16 > buf_end) a = buf_end - 16;      // (behind the buffer: bytes no code of the stream reaches)
                v[j] = *reinterpret_cast<const v4*>(a);
            }
            #pragma unroll
            for (u32 j = 0; j < WIN / 16; j++) { win[(4 * j) * 64u] = v[j].x; win[(4 * j + 1) * 64u] = v[j].y; win[(4 * j + 2) * 64u] = v[j].z; win[(4 * j + 3) * 64u] = v[j].w; }
            win_at = (u32)(a0 - in); have = true;      // (a0 may lie up to 15 bytes in front of the stream: win_at wraps, the differences below do not)
        }
        const u32 o = pos - win_at, d = o >> 2, sh = (o & 3u) * 8u;
        const u32 w0 = win[d * 64u], w1 = win[(d + 1) * 64u], w2 = win[(d + 2) * 64u];
        const u64 lo = (u64)w0 | ((u64)w1 << 32);
        return sh ? (lo >> sh) | ((u64)w2 << (64u - sh)) : lo;
    }
    __device__ __attribute__((always_inline)) u8 byte(u32 pos) const { return in[pos]; }
};
#endif

}  // namespace inflate_canon
