// inflate_wave.h -- DEFLATE (RFC 1951) on the GPU, one WAVE per BGZF block (device code only).
//
// A deflate stream is a chain: every code's position depends on the lengths of all codes before it.  So the wave
// decodes ONE stream, in lockstep, and the 64 lanes are used for everything that is not the chain:
//   * the compressed bytes live in registers: lane k holds bytes 8k .. 8k+7 of a 512-byte window of the stream (one
//     coalesced load per window, the next window requested a window ahead); taking the next 32 bits is a v_readlane
//     with a uniform lane number -- no memory access on the chain;
//   * the decoder's state (bit buffer, counts, positions) is wave-uniform and kept uniform explicitly
//     (readfirstlane / readlane results), so the compiler keeps it in scalar registers and the chain runs on the
//     scalar unit; per symbol the vector unit sees one LDS look-up (10-bit table, symbol << 4 | length);
//   * the Huffman tables are built by all lanes together in LDS: counts and canonical ranks by ballot, the look-up
//     table by entry (lane = table index, walking the canonical code lengths) -- no per-lane arrays, no scratch;
//   * literals collect in one register (lane k = byte k) and leave 64 at a time as one coalesced store;
//     matches are copied by all lanes, 64 bytes per round.
// Round 2's kernel ran the host's serial decoder on all 64 lanes redundantly: ~40 vector instructions and a one-byte
// store per LITERAL, the code-length arrays in scratch memory (688 B per lane, 157 spills): 28 GB/s of text.
//
// Every loop consumes input bits or produces output bytes and both are bounded (total input bits, output capacity), so a
// corrupt block ends in an error code, never in a fault or a hang.  Error codes: those of inflate_dev.h.
#pragma once
#include "inflate_dev.h"

namespace inflate_wave {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;
typedef unsigned char u8;
using mlst_inflate::OK; using mlst_inflate::E_INPUT; using mlst_inflate::E_BLOCKTYPE; using mlst_inflate::E_STORED; using mlst_inflate::E_LENGTHS;
using mlst_inflate::E_OUTPUT; using mlst_inflate::E_DISTANCE; using mlst_inflate::E_SYMBOL;

enum { LB = 10, DBITS = 8, SYM_D = 288 };
struct Tabs {                       // per wave, in LDS (3.6 KB)
    u16 lut[1 << LB];               // literal / length code (and the code-length code): symbol << 4 | length, 0 = longer code
    u16 dlut[1 << DBITS];           // distance code
    u16 sym[SYM_D + 32];            // symbols in canonical order: literal / length, then (from SYM_D) distance
    u16 cnt[2][16];                 // codes per length
    u8 len[320];                    // code lengths: literal / length symbols followed by distance symbols
};

// the order in which the code-length code's own lengths are sent (RFC 1951 3.2.7), five bits each, in two words
constexpr u64 order_word(int from, int to) {
    const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    u64 w = 0;
    for (int i = from; i < to; i++) w |= (u64)order[i] << (5 * (i - from));
    return w;
}
constexpr u64 ORDER_LO = order_word(0, 12), ORDER_HI = order_word(12, 19);

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((always_inline)) inline u32 uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __attribute__((always_inline)) inline void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// ---- input: 512-byte windows of the stream in registers
struct In {
    const u64* base;                // 8-byte aligned address at or below the stream's first byte
    u64 last_chunk;                 // index (from base) of the last 8-byte chunk that may be read
    u32 wlo, whi, nlo, nhi;         // this lane's chunk of the current and of the next window
    u32 nd;                         // dwords taken so far (uniform)
    u64 buf; u32 cnt;               // bit buffer (uniform): cnt valid bits
    u32 limit_bits;                 // bits of the stream, counted from base
    int lane;
    __device__ __attribute__((always_inline)) void load(u32 w, u32& lo, u32& hi) const {
        u64 c = (u64)w * 64 + (u32)lane; c = c < last_chunk ? c : last_chunk;
        const u64 v = __builtin_nontemporal_load(base + c);
        lo = (u32)v; hi = (u32)(v >> 32);
    }
    __device__ __attribute__((always_inline)) void open(const u8* p, u64 n, const u8* buf_end, int lane_) {
        lane = lane_;
        const u64 a = (u64)(uintptr_t)p, a0 = a & ~7ull;
        base = reinterpret_cast<const u64*>(a0);
        const u64 e = ((u64)(uintptr_t)buf_end - a0) >> 3;      // whole chunks inside the caller's buffer
        last_chunk = e ? e - 1 : 0;
        limit_bits = (u32)((a - a0 + n) * 8);
        load(0, wlo, whi); load(1, nlo, nhi);
        nd = 0; buf = 0; cnt = 0;
        refill(); refill();                                       // 64 bits: up to 56 of them lie in front of the stream
        const u32 skip = (u32)(a - a0) * 8;                       // bytes in front of the stream
        buf >>= skip; cnt -= skip;
    }
    __device__ __attribute__((always_inline)) u32 next32() {
        const u32 d = nd & 127u, l = d >> 1;
        const u32 v = (d & 1u) ? (u32)__builtin_amdgcn_readlane((int)whi, (int)l) : (u32)__builtin_amdgcn_readlane((int)wlo, (int)l);
        nd++;
        if ((nd & 127u) == 0) { wlo = nlo; whi = nhi; load((nd >> 7) + 1, nlo, nhi); }
        return v;
    }
    __device__ __attribute__((always_inline)) void refill() { if (cnt <= 32) { buf |= (u64)next32() << cnt; cnt += 32; } }      // afterwards cnt >= 32
    __device__ __attribute__((always_inline)) u32 peek(int n) const { return (u32)buf & ((1u << n) - 1u); }
    __device__ __attribute__((always_inline)) void drop(u32 n) { buf >>= n; cnt -= n; }
    __device__ __attribute__((always_inline)) u32 take(int n) { const u32 v = peek(n); drop((u32)n); return v; }                 // n <= 16, cnt >= n assured by the caller's refill
    __device__ __attribute__((always_inline)) u32 used_bits() const { return nd * 32u - cnt; }
    __device__ __attribute__((always_inline)) bool overrun() const { return used_bits() > limit_bits; }
    // byte position (from base) of the next unread whole byte, and a restart there
    __device__ __attribute__((always_inline)) void seek_bytes(u64 byte_from_base) {
        const u32 w = (u32)(byte_from_base >> 9);
        load(w, wlo, whi); load(w + 1, nlo, nhi);
        nd = w * 128u + (u32)((byte_from_base & 511u) >> 2);
        buf = 0; cnt = 0;
        refill();
        const u32 skip = (u32)(byte_from_base & 3u) * 8;
        buf >>= skip; cnt -= skip;
    }
};

// ---- output: pending literals in a register, matches copied by all lanes
struct Out {
    u8* out; u32 op, cap;           // op = bytes produced (pending literals included)
    u32 lit, nlit;                  // lane k holds pending byte k; nlit of them (uniform)
    u32 safe;                       // bytes below this position were stored before the last fence
    int lane;
    __device__ __attribute__((always_inline)) void flush() {
        if (nlit) { if ((u32)lane < nlit) out[op - nlit + (u32)lane] = (u8)lit; nlit = 0; }
    }
    __device__ __attribute__((always_inline)) void put(u32 c) {
        lit = (u32)lane == nlit ? c : lit;          // (a compare and a select; this compiler has no writelane builtin)
        nlit++; op++;
        if (nlit == 64) flush();
    }
    __device__ __attribute__((always_inline)) void copy(u32 dist, u32 len) {
        flush();
        // source bytes that this wave stored since the last fence have to have landed before they are read back
        if (op - dist + (dist < len ? dist : len) > safe) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); safe = op; }
        const u8* src = out + op - dist;
        if (dist >= len) { for (u32 k = (u32)lane; k < len; k += 64) out[op + k] = src[k]; }
        else             { for (u32 k = (u32)lane; k < len; k += 64) out[op + k] = src[k % dist]; }
        op += len;
    }
    __device__ __attribute__((always_inline)) void raw(const u8* src, u32 len) { flush(); for (u32 k = (u32)lane; k < len; k += 64) out[op + k] = src[k]; op += len; }
};

// ---- canonical code from the lengths T.len[off .. off + n): counts, symbols in code order, look-up table.
// Returns 0 for a complete code, > 0 for an incomplete one, < 0 for an over-subscribed one (as inflate_dev.h's build).
template <int BITS>
__device__ __attribute__((always_inline)) inline int build(Tabs& T, int which, u32 off, u32 n, u16* lut, int lane) {
    u16* const cnt = T.cnt[which];
    u16* const sym = T.sym + (which ? SYM_D : 0);
    const u64 lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    u32 mylen[5];
    #pragma unroll
    for (int j = 0; j < 5; j++) { const u32 s = (u32)j * 64 + (u32)lane; mylen[j] = s < n ? (u32)T.len[off + s] : 0u; }
    int left = 1; u32 offs = 0, total = 0;
    for (u32 l = 1; l <= 15; l++) {                               // uniform
        u32 c = 0;
        #pragma unroll
        for (int j = 0; j < 5; j++) {
            if ((u32)j * 64 < n) {
                const bool mine = mylen[j] == l;
                const u64 m = __ballot(mine);
                if (mine) sym[offs + c + (u32)__popcll(m & lt)] = (u16)((u32)j * 64 + (u32)lane);
                c += (u32)__popcll(m);
            }
        }
        if (lane == 0) cnt[l] = (u16)c;
        left = (left << 1) - (int)c;
        if (left < 0) return left;
        offs += c; total += c;
    }
    if (lane == 0) cnt[0] = (u16)(n - total);
    wave_sync();
    // look-up table by entry: walk the canonical code along the bits of the index (first bit of the stream = bit 0)
    for (u32 i = (u32)lane; i < (1u << BITS); i += 64) {
        u32 code = 0, first = 0, index = 0, entry = 0;
        #pragma unroll
        for (u32 l = 1; l <= (u32)BITS; l++) {
            code |= (i >> (l - 1)) & 1u;
            const u32 count = cnt[l];
            if (entry == 0 && code < first + count && code >= first) entry = ((u32)sym[index + (code - first)] << 4) | l;
            index += count; first += count; first <<= 1; code <<= 1;
        }
        lut[i] = (u16)entry;
    }
    wave_sync();
    if (total == 0) return 0;                                     // no codes: complete, but decoding anything with it fails
    return left;
}

// one code of more than the table's bits (rare): the canonical walk, bit by bit; -> symbol or error
__device__ __attribute__((always_inline)) inline int decode_long(In& in, const Tabs& T, int which) {
    const u16* cnt = T.cnt[which]; const u16* sym = T.sym + (which ? SYM_D : 0);
    u32 code = 0, first = 0, index = 0;
    for (u32 l = 1; l <= 15; l++) {
        in.refill();
        code |= in.take(1);
        const u32 count = uni(cnt[l]);
        if (code < first + count && code >= first) return (int)uni(sym[index + (code - first)]);
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return E_SYMBOL;
}
template <int BITS>
__device__ __attribute__((always_inline)) inline int decode(In& in, const Tabs& T, const u16* lut, int which) {
    in.refill();
    const u32 e = uni(lut[in.peek(BITS)]);
    if (e) { in.drop(e & 15u); return (int)(e >> 4); }
    return decode_long(in, T, which);
}

// literal / length and distance codes until the end-of-block symbol
__device__ __attribute__((always_inline)) inline int codes(In& in, Tabs& T, Out& o) {
    for (;;) {
        int sym = decode<LB>(in, T, T.lut, 0);
        if (sym < 0) return sym;
        if (sym < 256) {
            if (o.op >= o.cap) return E_OUTPUT;
            o.put((u32)sym);
        } else if (sym == 256) return in.overrun() ? E_INPUT : OK;
        else {
            sym -= 257;
            if (sym >= 29) return E_SYMBOL;
            const int le = sym < 8 || sym == 28 ? 0 : (sym - 4) >> 2;
            const u32 lb = sym < 8 ? 3u + (u32)sym : (sym == 28 ? 258u : ((4u + ((u32)sym & 3u)) << le) + 3u);
            in.refill();
            const u32 len = lb + in.take(le);
            const int ds = decode<DBITS>(in, T, T.dlut, 1);
            if (ds < 0) return ds;
            if (ds >= 30) return E_SYMBOL;
            const int de = ds < 4 ? 0 : (ds - 2) >> 1;
            const u32 db = ds < 4 ? 1u + (u32)ds : ((2u + ((u32)ds & 1u)) << de) + 1u;
            in.refill();
            const u32 dist = db + in.take(de);                  // up to 13 extra bits
            if (dist > o.op) return E_DISTANCE;
            if (o.op + len > o.cap) return E_OUTPUT;
            if (in.overrun()) return E_INPUT;
            o.copy(dist, len);
        }
    }
}

// one raw deflate stream of n_in bytes at `in_p` -> at most cap bytes at `out_p`; *produced = bytes written
__device__ __attribute__((always_inline)) inline int inflate_stream(const u8* in_p, u64 n_in, const u8* buf_end, u8* out_p, u32 cap, Tabs& T, int lane, u32* produced) {
    In in; in.open(in_p, n_in, buf_end, lane);
    Out o; o.out = out_p; o.op = 0; o.cap = cap; o.lit = 0; o.nlit = 0; o.safe = 0; o.lane = lane;
    int rc = OK;
    for (;;) {
        in.refill();
        const u32 last = in.take(1), type = in.take(2);
        if (in.overrun()) { rc = E_INPUT; break; }
        if (type == 0) {
            const u64 byte0 = ((u64)in.used_bits() + 7) >> 3;                // next whole byte, from in.base
            const u8* p = reinterpret_cast<const u8*>(in.base) + byte0;
            if ((byte0 + 4) * 8 > in.limit_bits) { rc = E_INPUT; break; }
            const u32 len = (u32)p[0] | ((u32)p[1] << 8), nlen = (u32)p[2] | ((u32)p[3] << 8);
            if (uni(len) != (~uni(nlen) & 0xFFFFu)) { rc = E_STORED; break; }
            const u32 ln = uni(len);
            if ((byte0 + 4 + ln) * 8 > in.limit_bits) { rc = E_INPUT; break; }
            if (o.op + ln > o.cap) { rc = E_OUTPUT; break; }
            o.raw(p + 4, ln);
            in.seek_bytes(byte0 + 4 + ln);
        } else if (type == 1 || type == 2) {
            u32 nlen = 288, ndist = 30;
            if (type == 1) {
                for (u32 s = (u32)lane; s < 320; s += 64) T.len[s] = (u8)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
                wave_sync();
            } else {
                in.refill();
                nlen = in.take(5) + 257; ndist = in.take(5) + 1;
                const u32 ncode = in.take(4) + 4;
                if (nlen > 286 || ndist > 30) { rc = E_LENGTHS; break; }
                if (lane < 19) T.len[lane] = 0;
                wave_sync();
                for (u32 i = 0; i < ncode; i++) {
                    in.refill();
                    const u32 x = in.take(3);
                    const u32 ord = (u32)((i < 12 ? ORDER_LO >> (5 * i) : ORDER_HI >> (5 * (i - 12))) & 31ull);
                    if (lane == 0) T.len[ord] = (u8)x;
                }
                wave_sync();
                if (build<7>(T, 0, 0, 19, T.lut, lane) != 0) { rc = E_LENGTHS; break; }      // the code-length code must be complete
                // (the code-length code is in its tables now: T.len is free for the code lengths proper)
                u32 idx = 0, prev = 0; const u32 want = nlen + ndist;
                bool bad = false;
                while (idx < want) {
                    const int sym = decode<7>(in, T, T.lut, 0);
                    if (sym < 0) { rc = sym; bad = true; break; }
                    if (sym < 16) { if (lane == 0) T.len[idx] = (u8)sym; prev = (u32)sym; idx++; }
                    else {
                        u32 rep, val = 0;
                        in.refill();
                        if (sym == 16) { if (idx == 0) { rc = E_LENGTHS; bad = true; break; } val = prev; rep = 3 + in.take(2); }
                        else if (sym == 17) rep = 3 + in.take(3);
                        else rep = 11 + in.take(7);
                        if (idx + rep > want) { rc = E_LENGTHS; bad = true; break; }
                        for (u32 k = (u32)lane; k < rep; k += 64) T.len[idx + k] = (u8)val;
                        prev = val; idx += rep;
                    }
                    if (in.overrun()) { rc = E_INPUT; bad = true; break; }
                }
                if (bad) break;
                wave_sync();
                if (uni(T.len[256]) == 0) { rc = E_LENGTHS; break; }                        // no end-of-block code
            }
            // (the fixed distance code is incomplete by definition, 30 of 32 codes: only a dynamic block's codes are checked)
            int e = build<LB>(T, 0, 0, nlen, T.lut, lane);
            if (type == 2 && (e < 0 || (e > 0 && nlen != (u32)uni(T.cnt[0][0]) + (u32)uni(T.cnt[0][1])))) { rc = E_LENGTHS; break; }
            e = build<DBITS>(T, 1, nlen, ndist, T.dlut, lane);
            if (type == 2 && (e < 0 || (e > 0 && ndist != (u32)uni(T.cnt[1][0]) + (u32)uni(T.cnt[1][1])))) { rc = E_LENGTHS; break; }
            rc = codes(in, T, o);
            if (rc != OK) break;
        } else { rc = E_BLOCKTYPE; break; }
        if (last) break;
    }
    o.flush();
    *produced = o.op;
    return rc;
}
#endif  // __HIP_DEVICE_COMPILE__

}  // namespace inflate_wave
