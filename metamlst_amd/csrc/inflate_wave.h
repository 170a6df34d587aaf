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

#if !defined(MLST_INFLATE_PB)
#define MLST_INFLATE_PB 4         /* far matches copied per memory round trip (a multiple of four) */
#endif
enum { LB = 10, DBITS = 8, SYM_D = 288, RING = 1024 };
struct Tabs {                       // per wave, in LDS (5.1 KB: LDS decides how many blocks a CU decodes at a time -- 31)
    u8 ring[RING];                  // the last RING bytes of output: near matches are copied from here (see Out::copy)
    u16 lut[1 << LB];               // literal / length code.  While it is built (and for the code-length code): symbol << 4 | bits, 0 =
                                    // longer code.  Then converted in place (encode_lut): bits 0-3 = bits consumed; bit 15 clear: bits 4-11 =
                                    // the literal, bit 12 = end of block, bit 13 = invalid symbol; bit 15 set: bits 4-11 = base length - 3,
                                    // bits 12-14 = extra bits of the length code
    u32 dlut[1 << DBITS];           // distance code: bits 0-3 = bits consumed, 4-7 = extra bits, 8-22 = base distance, bit 23 = invalid symbol;
                                    // 0 = longer code (built as a single-symbol table in its own upper half, then converted in place)
    u16 sym[SYM_D + 32];            // symbols in canonical order: literal / length, then (from SYM_D) distance
    u16 cnt[2][16];                 // codes per length
    u8 len[320];                    // code lengths: literal / length symbols followed by distance symbols
    u32 pq[3][MLST_INFLATE_PB];     // far matches waiting for their copy (Out::copy): source, destination, length
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
#if !defined(MLST_INFLATE_VALU)
#define MLST_INFLATE_VALU 1
#endif
// MLST_INFLATE_VALU = 0: the chain's state is made uniform explicitly and runs on the scalar unit (one per CU, shared by
// all its waves); 1: nothing is made uniform, every lane carries the state and the chain runs on the vector units (four
// per CU).  Same results; which is faster is a measurement (profiles/round3/README.md).
// MLST_INFLATE_GROUP = lanes per stream: 64 (one block per wave) or 16 (four blocks per wave, each decoded by sixteen lanes:
// every vector instruction then advances four streams; needs MLST_INFLATE_VALU = 1).  In the structures below `lane` is the
// lane's number INSIDE its group and `gbase` the wave lane of the group's first.
#if !defined(MLST_INFLATE_GROUP)
#define MLST_INFLATE_GROUP 64
#endif
enum { GS = MLST_INFLATE_GROUP, GLOG = (GS == 64 ? 6 : (GS == 32 ? 5 : 4)) };
static_assert(GS == 64 || MLST_INFLATE_VALU, "groups of fewer than 64 lanes keep their state in vector registers");
__device__ __attribute__((always_inline)) inline u32 uni(u32 v) { return MLST_INFLATE_VALU ? v : (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __attribute__((always_inline)) inline void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// ---- input: 512-byte windows of the stream in registers
struct In {
    const u64* base;                // 8-byte aligned address at or below the stream's first byte
    u64 last_chunk;                 // index (from base) of the last 8-byte chunk that may be read
    u32 wlo, whi, nlo, nhi;         // this lane's chunk of the current and of the next window
    u32 nd;                         // dwords taken so far (uniform)
    u64 buf; u32 cnt;               // bit buffer (uniform): cnt valid bits
    u32 limit_bits;                 // bits of the stream, counted from base
    int lane, gbase;
    __device__ __attribute__((always_inline)) void load(u32 w, u32& lo, u32& hi) const {
        u64 c = (u64)w * GS + (u32)lane; c = c < last_chunk ? c : last_chunk;
        const u64 v = __builtin_nontemporal_load(base + c);
        lo = (u32)v; hi = (u32)(v >> 32);
    }
    __device__ __attribute__((always_inline)) void open(const u8* p, u64 n, const u8* buf_end, int lane_, int gbase_) {
        lane = lane_; gbase = gbase_;
        const u64 a = (u64)(uintptr_t)p, a0 = a & ~7ull;
        base = reinterpret_cast<const u64*>(p - (a - a0));          // (arithmetic on the pointer itself: it stays a global-memory pointer)
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
        const u32 d = nd & (2u * GS - 1u), l = d >> 1;
        u32 v;
        // (one stream per wave: the lane number is the same in every lane whichever unit carries the state -- a v_readlane
        // behind a v_readfirstlane instead of a trip through the LDS crossbar, on the chain once per 32 bits of input)
        if (MLST_INFLATE_VALU && GS != 64) v = (u32)__shfl((int)((d & 1u) ? whi : wlo), gbase + (int)l);
        else { const int ls = __builtin_amdgcn_readfirstlane((int)l); const u32 sel = (d & 1u) ? whi : wlo; v = (u32)__builtin_amdgcn_readlane((int)sel, ls); }
        nd++;
        if ((nd & (2u * GS - 1u)) == 0) { wlo = nlo; whi = nhi; load((nd >> (GLOG + 1)) + 1, nlo, nhi); }
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
        const u32 w = (u32)(byte_from_base >> (GLOG + 3));
        load(w, wlo, whi); load(w + 1, nlo, nhi);
        nd = w * (2u * GS) + (u32)((byte_from_base & (8u * GS - 1u)) >> 2);
        buf = 0; cnt = 0;
        refill();
        const u32 skip = (u32)(byte_from_base & 3u) * 8;
        buf >>= skip; cnt -= skip;
    }
};

// ---- output: pending literals in a register, matches copied by all lanes.
// A match reads bytes this wave has just written.  Read back from global memory they have to have LANDED first: a fence
// (s_waitcnt vmcnt(0)) per match whose source was written since the last fence -- and in FASTQ nearly every match is
// such a one (a quality run is a literal followed by a match at distance 1; the same read name, one record back): the
// first version of this kernel spent ~2,000 cycles per symbol waiting for its own stores.  So every byte also goes into
// a ring of the last RING bytes in LDS (in order within a wave, no wait), and a match whose source lies inside the ring
// (distance + length <= RING) is copied from there; only far matches read global memory, and those need a fence at most
// once per ~RING bytes of output.
struct Stats { u32 lookups = 0, lits = 0, near_ = 0, far_def = 0, far_sync = 0, far_flush = 0, fences = 0, builds = 0; unsigned long long t_build = 0, t_codes = 0; };
#if defined(MLST_INFLATE_STATS)      // diagnostic builds only (hipcc -DMLST_INFLATE_STATS): a dozen more scalar registers and adds per symbol
#define ISTAT(...) __VA_ARGS__
#else
#define ISTAT(...)
#endif
struct Out {
    enum { PB = MLST_INFLATE_PB };  // far matches that may wait for their copy
    ISTAT(Stats st;)
    u8* out; u8* ring; u32* pq; u32 op, cap; // op = bytes produced (pending literals and waiting matches included)
    u32 lit, nlit;                  // lane k holds pending byte k; nlit of them (uniform)
    u32 safe;                       // bytes below this position were stored before the last fence
    u32 npend, pend_start, pend_maxsrc;      // waiting far matches: how many, the first one's destination, the largest source end
    int lane;
    __device__ __attribute__((always_inline)) void flush() {
        if (nlit) {
            // (a waiting match writes its ring slots when it is copied: nothing RING bytes further on may be written before that)
            if (npend && op - pend_start >= RING) flush_far();
            if ((u32)lane < nlit) { const u32 at = op - nlit + (u32)lane; out[at] = (u8)lit; ring[at & (RING - 1)] = (u8)lit; }
            nlit = 0;
        }
    }
    // n = 1..3 literals, first in the low byte (room in the output checked by the caller)
    __device__ __attribute__((always_inline)) void put_n(u32 bytes, u32 n) {
        if (nlit + n > GS) flush();
        const u32 k = (u32)lane - nlit;             // (wraps to a large number for the lanes that hold older bytes)
        lit = k < n ? (bytes >> (8 * k)) & 0xFFu : lit;
        nlit += n; op += n;
    }
    // The far matches that wait: all their loads, ONE wait, all their stores.  Copied one by one, every far match cost the
    // wave a memory round trip (the sequence lines of a FASTQ file are coded as short matches tens of kilobytes back:
    // ~11,000 per block, 7.5 ms per block); eight at a time share one.
    __device__ __attribute__((always_inline)) void flush_far() {
        if (npend == 0) return;
        ISTAT(st.far_flush++;)
        if (pend_maxsrc > safe) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); safe = pend_start; ISTAT(st.fences++;) }
        // every load of the batch is issued before the first wait: no branch around a load (a branch makes the compiler wait for
        // the load inside it), inactive slots and lanes read byte 0 of the output
        u32 sp[PB], dp[PB], l[PB], c[PB];
        #pragma unroll
        for (int i = 0; i < PB; i++) { sp[i] = pq[i]; dp[i] = pq[PB + i]; l[i] = pq[2 * PB + i]; }
        #pragma unroll
        for (int i = 0; i < PB; i++) { sp[i] = uni(sp[i]); dp[i] = uni(dp[i]); l[i] = (u32)i < npend ? uni(l[i]) : 0u; }
        #pragma unroll
        for (int i = 0; i < PB; i++) c[i] = out[(u32)lane < l[i] ? sp[i] + (u32)lane : 0u];
        #pragma unroll
        for (int i = 0; i + 4 <= PB; i += 4) asm volatile("" : "+v"(c[i]), "+v"(c[i + 1]), "+v"(c[i + 2]), "+v"(c[i + 3]));
        #pragma unroll
        for (int i = 0; i < PB; i++) {
            if ((u32)lane < l[i]) { const u32 d = dp[i] + (u32)lane; out[d] = (u8)c[i]; ring[d & (RING - 1)] = (u8)c[i]; }
        }
        npend = 0;
    }
    __device__ __attribute__((always_inline)) void copy(u32 dist, u32 len) {
        flush();
        const u32 src_end = op - dist + (dist < len ? dist : len);
        if (npend && (src_end > pend_start || op + len - pend_start >= RING)) flush_far();      // the source holds bytes of a match that has not been copied yet (or the ring would wrap onto one)
        if (dist + len <= RING) {                   // near: from the ring (LDS is in order within a wave)
            ISTAT(st.near_++;)
            if (dist >= len) {
                for (u32 k = (u32)lane; k < len; k += GS) { const u8 c = ring[(op - dist + k) & (RING - 1)]; out[op + k] = c; ring[(op + k) & (RING - 1)] = c; }
            } else if (dist == 1) {                 // a run of one byte
                const u8 c = ring[(op - 1) & (RING - 1)];
                for (u32 k = (u32)lane; k < len; k += GS) { out[op + k] = c; ring[(op + k) & (RING - 1)] = c; }
            } else {                                // the match overlaps its own output: source byte k is byte k mod dist of the period
                for (u32 k = (u32)lane; k < len; k += GS) { const u8 c = ring[(op - dist + k % dist) & (RING - 1)]; out[op + k] = c; ring[(op + k) & (RING - 1)] = c; }
            }
        } else if (len <= GS) {                     // far and short (dist > len: no overlap with itself): it waits for its turn
            ISTAT(st.far_def++;)
            if ((u32)lane == 0) { pq[npend] = op - dist; pq[PB + npend] = op; pq[2 * PB + npend] = len; }
            if (npend == 0) { pend_start = op; pend_maxsrc = 0; }
            pend_maxsrc = src_end > pend_maxsrc ? src_end : pend_maxsrc;
            npend++; op += len;
            if (npend == PB) flush_far();
            return;
        } else {
            // far and long: at once, from global memory; source bytes stored since the last fence have to have landed
            flush_far();
            ISTAT(st.far_sync++;)
            if (src_end > safe) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); safe = op; ISTAT(st.fences++;) }
            const u8* src = out + op - dist;
            for (u32 k = (u32)lane; k < len; k += GS) { const u8 c = src[k]; out[op + k] = c; ring[(op + k) & (RING - 1)] = c; }
        }
        op += len;
    }
    __device__ __attribute__((always_inline)) void raw(const u8* src, u32 len) {
        flush();
        for (u32 k = (u32)lane; k < len; k += GS) { const u8 c = src[k]; out[op + k] = c; if (k + RING >= len) ring[(op + k) & (RING - 1)] = c; }
        op += len;
    }
};

// ---- canonical code from the lengths T.len[off .. off + n): counts, symbols in code order, look-up table.
// Returns 0 for a complete code, > 0 for an incomplete one, < 0 for an over-subscribed one (as inflate_dev.h's build).
template <int BITS>
__device__ __attribute__((always_inline)) inline int build(Tabs& T, int which, u32 off, u32 n, u16* lut, int lane, int gbase) {
    u16* const cnt = T.cnt[which];
    u16* const sym = T.sym + (which ? SYM_D : 0);
    const u64 lt = (1ull << lane) - 1ull, gmask = GS == 64 ? ~0ull : ((1ull << (GS & 63)) - 1ull);
    int left = 1; u32 offs = 0, total = 0;
    for (u32 l = 1; l <= 15; l++) {                               // (the same trip count for every group of the wave)
        u32 c = 0;
        for (u32 s0 = 0; s0 < n; s0 += GS) {
            const u32 sI = s0 + (u32)lane;
            const bool mine = sI < n && (u32)T.len[off + sI] == l;
            const u64 m = (__ballot(mine) >> gbase) & gmask;
            if (mine) sym[offs + c + (u32)__popcll(m & lt)] = (u16)sI;
            c += (u32)__popcll(m);
        }
        if (lane == 0) cnt[l] = (u16)c;
        left = (left << 1) - (int)c;
        if (left < 0) return left;
        offs += c; total += c;
    }
    if (lane == 0) cnt[0] = (u16)(n - total);
    wave_sync();
    // look-up table by entry: walk the canonical code along the bits of the index (first bit of the stream = bit 0)
    for (u32 i = (u32)lane; i < (1u << BITS); i += GS) {
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

// base value and extra-bit count of a length symbol (0..28 = symbols 257..285) and of a distance symbol (0..29)
__device__ __attribute__((always_inline)) inline void len_base(u32 sym, u32& lb, u32& le) {
    le = (sym < 8 || sym == 28) ? 0u : (sym - 4) >> 2;
    lb = sym < 8 ? 3u + sym : (sym == 28 ? 258u : ((4u + (sym & 3u)) << le) + 3u);
}
__device__ __attribute__((always_inline)) inline void dist_base(u32 ds, u32& db, u32& de) {
    de = ds < 4 ? 0u : (ds - 2) >> 1;
    db = ds < 4 ? 1u + ds : ((2u + (ds & 1u)) << de) + 1u;
}
// the literal / length table with base and extra bits worked out per entry (one look-up per symbol and no arithmetic on
// the chain), converted in place.  (A table of up to three literals per look-up was tried: no gain -- the bases of a FASTQ
// file are coded as short matches, not as literals -- and 4 KB more LDS per block.)
__device__ __attribute__((always_inline)) inline void encode_lut(Tabs& T, int lane) {
    for (u32 i = (u32)lane; i < (1u << LB); i += GS) {
        const u32 e = T.lut[i];
        u32 m = 0;
        if (e) {
            const u32 sy = e >> 4, l = e & 15u;
            if (sy < 256) m = l | (sy << 4);
            else if (sy == 256) m = l | (1u << 12);
            else if (sy - 257 >= 29) m = l | (1u << 13);
            else { u32 lb, le; len_base(sy - 257, lb, le); m = l | ((lb - 3u) << 4) | (le << 12) | 0x8000u; }
        }
        T.lut[i] = (u16)m;
    }
    wave_sync();
}
// the distance table likewise: the single-symbol table sits in the upper half of T.dlut (entry i at 16-bit index 256 + i);
// converting entry i overwrites single-symbol entries 2 i - 256 and 2 i - 255, which have been read by then (i >= 128, and
// the 64 reads of a sweep are issued before its writes)
__device__ __attribute__((always_inline)) inline void encode_dlut(Tabs& T, int lane) {
    const u16* single = reinterpret_cast<const u16*>(T.dlut) + (1 << DBITS);
    for (u32 i = (u32)lane; i < (1u << DBITS); i += GS) {
        const u32 e = single[i];
        u32 m = 0;
        if (e) {
            const u32 ds = e >> 4;
            if (ds >= 30) m = (e & 15u) | (1u << 23);
            else { u32 db, de; dist_base(ds, db, de); m = (e & 15u) | (de << 4) | (db << 8); }
        }
        wave_sync();
        T.dlut[i] = m;
        wave_sync();
    }
}

// literal / length and distance codes until the end-of-block symbol
__device__ __attribute__((always_inline)) inline int codes(In& in, Tabs& T, Out& o) {
    for (;;) {
        in.refill();
        const u32 e = uni(T.lut[in.peek(LB)]);
        ISTAT(o.st.lookups++;)
        u32 lb, le;
        if (e) {
            in.drop(e & 15u);
            if (!(e & 0x8000u)) {
                if (e & 0x3000u) { if (e & 0x2000u) return E_SYMBOL; return in.overrun() ? E_INPUT : OK; }
                if (o.op >= o.cap) return E_OUTPUT;
                o.put_n((e >> 4) & 0xFFu, 1);
                ISTAT(o.st.lits++;)
                continue;
            }
            lb = ((e >> 4) & 0xFFu) + 3u; le = (e >> 12) & 7u;
        } else {
            const int sym = decode_long(in, T, 0);
            if (sym < 0) return sym;
            if (sym < 256) {
                if (o.op >= o.cap) return E_OUTPUT;
                o.put_n((u32)sym, 1);
                continue;
            }
            if (sym == 256) return in.overrun() ? E_INPUT : OK;
            if (sym - 257 >= 29) return E_SYMBOL;
            len_base((u32)sym - 257u, lb, le);
            in.refill();
        }
        const u32 len = lb + in.take((int)le);                // (>= 17 bits are left behind a table code, >= 12 behind these)
        in.refill();
        const u32 d = uni(T.dlut[in.peek(DBITS)]);
        u32 db, de;
        if (d) {
            if (d & (1u << 23)) return E_SYMBOL;
            in.drop(d & 15u);
            de = (d >> 4) & 15u; db = (d >> 8) & 0x7FFFu;
        } else {
            const int ds = decode_long(in, T, 1);
            if (ds < 0) return ds;
            if (ds >= 30) return E_SYMBOL;
            dist_base((u32)ds, db, de);
            in.refill();
        }
        const u32 dist = db + in.take((int)de);             // up to 13 extra bits (>= 17 are left behind a table code)
        if (dist > o.op) return E_DISTANCE;
        if (o.op + len > o.cap) return E_OUTPUT;
        o.copy(dist, len);      // (a stream that runs past its end is caught at its end-of-block symbol or by the output bound: every symbol writes at least a byte)
    }
}

// one raw deflate stream of n_in bytes at `in_p` -> at most cap bytes at `out_p`; *produced = bytes written
__device__ __attribute__((always_inline)) inline int inflate_stream(const u8* in_p, u64 n_in, const u8* buf_end, u8* out_p, u32 cap, Tabs& T, int lane, int gbase, u32* produced, Stats* stats_out = nullptr) {
    In in; in.open(in_p, n_in, buf_end, lane, gbase);
    Out o; o.out = out_p; o.ring = T.ring; o.pq = &T.pq[0][0]; o.op = 0; o.cap = cap; o.lit = 0; o.nlit = 0; o.safe = 0; o.npend = 0; o.pend_start = 0; o.pend_maxsrc = 0; o.lane = lane;
    ISTAT(o.st = Stats();)
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
                for (u32 s = (u32)lane; s < 320; s += GS) T.len[s] = (u8)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
                wave_sync();
            } else {
                in.refill();
                nlen = in.take(5) + 257; ndist = in.take(5) + 1;
                const u32 ncode = in.take(4) + 4;
                if (nlen > 286 || ndist > 30) { rc = E_LENGTHS; break; }
                for (u32 s2 = (u32)lane; s2 < 19; s2 += GS) T.len[s2] = 0;
                wave_sync();
                for (u32 i = 0; i < ncode; i++) {
                    in.refill();
                    const u32 x = in.take(3);
                    const u32 ord = (u32)((i < 12 ? ORDER_LO >> (5 * i) : ORDER_HI >> (5 * (i - 12))) & 31ull);
                    if (lane == 0) T.len[ord] = (u8)x;
                }
                wave_sync();
                if (build<7>(T, 0, 0, 19, T.lut, lane, gbase) != 0) { rc = E_LENGTHS; break; }      // the code-length code must be complete
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
                        for (u32 k = (u32)lane; k < rep; k += GS) T.len[idx + k] = (u8)val;
                        prev = val; idx += rep;
                    }
                    if (in.overrun()) { rc = E_INPUT; bad = true; break; }
                }
                if (bad) break;
                wave_sync();
                if (uni(T.len[256]) == 0) { rc = E_LENGTHS; break; }                        // no end-of-block code
            }
            // (the fixed distance code is incomplete by definition, 30 of 32 codes: only a dynamic block's codes are checked)
            ISTAT(const unsigned long long tb0 = __builtin_readcyclecounter(); o.st.builds++;)
            int e = build<LB>(T, 0, 0, nlen, T.lut, lane, gbase);
            if (type == 2 && (e < 0 || (e > 0 && nlen != (u32)uni(T.cnt[0][0]) + (u32)uni(T.cnt[0][1])))) { rc = E_LENGTHS; break; }
            encode_lut(T, lane);
            e = build<DBITS>(T, 1, nlen, ndist, reinterpret_cast<u16*>(T.dlut) + (1 << DBITS), lane, gbase);
            if (type == 2 && (e < 0 || (e > 0 && ndist != (u32)uni(T.cnt[1][0]) + (u32)uni(T.cnt[1][1])))) { rc = E_LENGTHS; break; }
            encode_dlut(T, lane);
            ISTAT(const unsigned long long tb1 = __builtin_readcyclecounter();)
            rc = codes(in, T, o);
            ISTAT(o.st.t_build += tb1 - tb0; o.st.t_codes += __builtin_readcyclecounter() - tb1;)
            if (rc != OK) break;
        } else { rc = E_BLOCKTYPE; break; }
        if (last) break;
    }
    o.flush();
    o.flush_far();
    ISTAT(if (stats_out) *stats_out = o.st;)
    *produced = o.op;
    return rc;
}
#endif  // __HIP_DEVICE_COMPILE__

}  // namespace inflate_wave
