// inflate_lane.h -- DEFLATE on the GPU in two kernels (device code only; round 4).
//
// inflate_wave.h decodes ONE stream per wave: 134 instructions per Huffman symbol of which 63 of 64 lanes repeat the first
// lane's arithmetic, and the kernel is bound by instruction issue (profiles/round4/inflate.md: 85 cycles per symbol and CU).
// The chain of a deflate stream is in its BITS -- where a code starts depends on every code before it -- not in its
// bytes: what a match copies does not decide where the next code starts.  So the two are taken apart:
//
//   phase 1  k_inflate_tok   one LANE per BGZF block: the serial decoder of inflate_dev.h (the one the host tests run
//            against zlib), its tables in LDS (1.3 KB per lane), emits one 32-bit TOKEN per symbol -- a literal, a match
//            (length, distance) or a stored run -- and copies nothing.  Every vector instruction advances 64 streams.
//   phase 2  k_inflate_ptr   one 1,024-thread workgroup per block: the tokens' output offsets by a prefix sum, then every
//            byte of the block gets a 16-bit POINTER in LDS (128 KB): a literal points at itself, byte k of a match at the
//            byte `distance` before it.  Pointer jumping (ptr[p] = ptr[ptr[p]], in place, until nothing moves: the
//            depth of the longest chain halves per round) turns every pointer into the position of the literal the byte
//            comes from, and one gather writes the block.  No byte waits for another one's store; nothing leaves the LDS
//            between the fill and the gather.
//
// A block whose tokens do not fit its share of the token buffer (more than TOK_CAP symbols: data that hardly compresses) is
// flagged and left to inflate_wave.h's kernel.  Errors: the codes of inflate_dev.h; every loop is bounded as there.
#pragma once
#include "inflate_dev.h"

namespace inflate_lane {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;
typedef unsigned char u8;

struct Blk { u64 in_off, out_off; u32 in_len, out_len; };      // one BGZF block: its deflate stream in the compressed buffer, its text in the output
enum : u32 { TAG_LIT = 0u, TAG_RAW = 1u, TAG_MATCH = 2u, TAG_OPERAND = 3u, TOK_OVERFLOW = 0xFFFFFFFFu };
enum : u32 { LIT_BASE = 0xFF00u };      // phase 2: a 16-bit pointer >= LIT_BASE is a literal, its byte in the low 8 bits
// token (tag in bits 30-31): literal = the byte; match = (length - 3) << 16 | (distance - 1); stored run = its length (16 bits),
// followed by an OPERAND word: where the run starts, in bytes from the start of the block's deflate stream

#if defined(__HIP_DEVICE_COMPILE__)
struct OutTok;
__device__ inline int codes_simt(mlst_inflate::Bits& b, const mlst_inflate::Huff& lc, const mlst_inflate::HuffD& dc, OutTok& o, uint64_t cap);
struct OutTok {
    u32* tok; u32 nt, cap; uint64_t op; const u8* in_base; bool over;
    __device__ static int codes(mlst_inflate::Bits& b, const mlst_inflate::Huff& lc, const mlst_inflate::HuffD& dc, OutTok& o, uint64_t cap) { return codes_simt(b, lc, dc, o, cap); }
    __device__ __attribute__((always_inline)) void emit(u32 t) { if (nt < cap) tok[nt] = t; else over = true; nt++; }
    __device__ __attribute__((always_inline)) void put(u8 c) { emit((u32)c); op++; }
    __device__ __attribute__((always_inline)) void copy(u32 dist, u32 len) { emit((TAG_MATCH << 30) | ((len - 3u) << 16) | (dist - 1u)); op += len; }
    __device__ __attribute__((always_inline)) void raw(const u8* src, u32 len) {
        if (len == 0) return;
        emit((TAG_RAW << 30) | len); emit((TAG_OPERAND << 30) | (u32)(src - in_base)); op += len;      // (len <= 65535: LEN is a 16-bit field)
    }
};

// The symbol loop for one lane per stream.  inflate_dev.h's codes() is written for one thread: a literal, a match, a long code,
// a refill are branches, and 64 lanes that each take their own walk all of them one after the other (measured: 1,057
// instructions per step of the wave, 4.6 cycles each with one wave per SIMD).  Here a step is one straight line that every
// lane walks: refill when fewer than 48 bits are left (a symbol takes at most 15 + 5 + 15 + 13), look the literal / length code
// up, and -- only the lanes whose code is longer than the table's 8 bits -- find it by the canonical walk over lengths 9..15
// on the bit-reversed window (no bit-by-bit loop); a literal's token leaves, the match lanes go on through length extra bits,
// distance code and distance extra bits, all cut from the same 64-bit buffer without another refill.
// (Branches cost a wave twice: the lanes that do not take one wait, and every divergent `if` is a saved and restored execution
// mask -- the first version of this loop, with the serial decoder's early returns, spent 290 scalar instructions per step on
// them.  So: errors are a sticky per-lane code looked at once per step, the canonical walk has a fixed trip count and selects
// its hit, and what is left are four `if`s: refill, long literal / length code, match, long distance code.)
template <int FROM>
__device__ __attribute__((always_inline)) inline int long_code(const uint16_t* count, const uint16_t* symbol, u32 first, u32 index, u32 rev15, int& len_out) {
    // rev15: the next 15 stream bits, first bit most significant (canonical codes are sent most significant bit first)
    int found_l = 0; u32 found = 0;
    #pragma unroll
    for (int l = FROM; l <= 15; l++) {
        const u32 code = rev15 >> (15 - l), c = count[l];
        const bool hit = found_l == 0 && code - first < c;
        found = hit ? index + (code - first) : found; found_l = hit ? l : found_l;
        index += c; first = (first + c) << 1;
    }
    len_out = found_l;
    return found_l ? (int)symbol[found] : (int)mlst_inflate::E_SYMBOL;
}
__device__ inline int codes_simt(mlst_inflate::Bits& b, const mlst_inflate::Huff& lc, const mlst_inflate::HuffD& dc, OutTok& o, uint64_t cap) {
    using namespace mlst_inflate;
    // where the canonical walks start behind the tables' bits: first code and symbol index of length LB + 1
    u32 lf = 0, li = 0, df = 0, di = 0;
    for (int l = 1; l <= Huff::LB; l++) { const u32 c = lc.count[l]; li += c; lf = (lf + c) << 1; }
    for (int l = 1; l <= HuffD::LB; l++) { const u32 c = dc.count[l]; di += c; df = (df + c) << 1; }
    u32 op = (u32)o.op; const u32 cap32 = (u32)cap;          // (a block holds at most 65,536 bytes: 32-bit arithmetic on the path)
    int err = OK;
    for (;;) {
        if (b.cnt < 48) refill(b);
        const u32 lo = (u32)b.buf;
        const u32 e = lc.lut[lo & ((1u << Huff::LB) - 1u)];
        int sym = (int)(e >> Huff::SH), l = (int)(e & ((1u << Huff::SH) - 1u));
        if (e == 0) {
            sym = long_code<Huff::LB + 1>(lc.count, lc.symbol, lf, li, __brev(lo) >> 17, l);
            err = sym < 0 ? sym : err; sym = sym < 0 ? 256 : sym;
        }
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
            const u32 lo2 = (u32)b.buf;
            const u32 ed = dc.lut[lo2 & ((1u << HuffD::LB) - 1u)];
            int ds = (int)(ed >> HuffD::SH), dl = (int)(ed & ((1u << HuffD::SH) - 1u));
            if (ed == 0) {
                ds = long_code<HuffD::LB + 1>(dc.count, dc.symbol, df, di, __brev(lo2) >> 17, dl);
                err = ds < 0 ? ds : err; ds = ds < 0 ? 0 : ds;
            }
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
        if (op + n_out > cap32) { err = E_OUTPUT; break; }
        o.emit(token); op += n_out;
    }
    o.op = op;
    return err;
}

// ---- phase 1: lane = block
__device__ __attribute__((always_inline)) inline void tok_body(const u8* comp, const u8* comp_end, const Blk* blk, u32 n_blk, u32 err_base,
                                                               u32* tok, u32 tok_cap, u32* n_tok, u32* err, mlst_inflate::Tables* tabs, unsigned long* win) {
    const u32 lane = threadIdx.x & 63u;
    for (u32 i0 = blockIdx.x * 64u; i0 < n_blk; i0 += gridDim.x * 64u) {
        const u32 i = i0 + lane;
        if (i >= n_blk) continue;
        const Blk B = blk[i];
        const u32 want = B.out_len;
        if (want > LIT_BASE) { n_tok[i] = (u32)TOK_OVERFLOW; continue; }      // (phase 2 keeps literals in the pointers' top 256 values; bgzip's blocks hold 65,280 bytes)
        OutTok o; o.tok = tok + (u64)i * tok_cap; o.nt = 0; o.cap = tok_cap; o.op = 0; o.in_base = comp + B.in_off; o.over = false;
        int rc = mlst_inflate::inflate_stream<OutTok, true>(comp + B.in_off, (u64)B.in_len, o, (u64)want, &tabs[lane], win + lane, comp_end);
        if (rc == mlst_inflate::OK && o.op != (u64)want) rc = mlst_inflate::E_SHORT;
        if (rc != mlst_inflate::OK) { if (atomicCAS(&err[0], 0u, err_base + i + 1u) == 0u) err[1] = (u32)(-rc); n_tok[i] = 0; }
        else n_tok[i] = o.over ? (u32)TOK_OVERFLOW : o.nt;
    }
}

// ---- phase 2: workgroup = block.  LDS: ptr[65536] (128 KB) + the scan's partial sums.
// Round 5: a literal is kept IN its pointer (LIT_BASE | byte: the top 256 pointer values, free because a block that takes
// this path holds at most LIT_BASE bytes), so a pointer that reaches a literal takes the byte with it and is final at once;
// nothing is stored to or gathered from global memory before the end, where the text leaves the LDS in 16-byte stores, and
// the newlines of every 2^nl_shift-byte cell of the text buffer are counted on the way (the FASTQ parser's first pass).
__device__ __attribute__((always_inline)) inline u32 wave_incl_scan(u32 v) {      // DPP lane moves: no trip through the LDS crossbar (__shfl_up was six of them per tile)
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);      // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);      // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);      // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);      // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);      // row_bcast:15
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);      // row_bcast:31
    return v;
}
__device__ __attribute__((always_inline)) inline u32 nl_count4(u32 w) {      // newlines among the four bytes of w
    const u32 y = w ^ 0x0A0A0A0Au;
    return (u32)__popc(~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu));
}
template <int NT>      // NT = threads per workgroup (a multiple of 64, at most 1024)
__device__ __attribute__((always_inline)) inline void ptr_body(const u8* comp, const Blk* blk, u32 n_blk, u32 err_base,
                                                               const u32* tok, u32 tok_cap, const u32* n_tok, u8* out, u32* err,
                                                               u16* ptr, u32* s_part, u32* s_nl /* 20 */, u32* nl, u32 nl_shift) {
    constexpr int NW = NT / 64;
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (u32 b = blockIdx.x; b < n_blk; b += gridDim.x) {
        const u32 nt = n_tok[b];
        if (nt == (u32)TOK_OVERFLOW || nt == 0) continue;              // block-uniform (left to the other kernel / failed in phase 1)
        const u32* T = tok + (u64)b * tok_cap;
        const Blk B = blk[b];
        u8* O = out + B.out_off;
        const u8* I = comp + B.in_off;
        const u32 total = B.out_len;
#if defined(MLST_PTR_TRACE)      // profiling builds: cycles per phase, summed over the blocks, in err[2..5] (fill, jumping, write-out, rounds)
        const unsigned long long tr0 = __builtin_readcyclecounter();
#endif
        if (tid < 20) s_nl[tid] = 0;
        // ---- fill: every wave takes a contiguous run of the tokens.  Pass A adds up the run's output bytes (one barrier for all
        // sixteen sums), pass B walks the run again -- the tokens come from L2 the second time -- with a scan inside the wave and
        // writes the pointers (or literal bytes).  (Until round 5 the workgroup went through the tokens in tiles of NT with two
        // barriers per tile: 44 barriers per block, 68 k cycles of the kernel's 140 k.)
        const u32 per_wave = ((nt + (u32)NT - 1u) / (u32)NT) * 64u;
        const u32 w_lo = wave * per_wave < nt ? wave * per_wave : nt, w_hi = w_lo + per_wave < nt ? w_lo + per_wave : nt;
        auto tok_len = [](u32 t) -> u32 {
            const u32 tag = t >> 30;
            return tag == TAG_MATCH ? ((t >> 16) & 0xFFu) + 3u : (tag == TAG_RAW ? (t & 0xFFFFu) : (tag == TAG_LIT ? 1u : 0u));
        };
        bool bad = false;
        {
            u32 sum = 0;
            for (u32 ti = w_lo + lane; ti < w_hi; ti += 64u) sum += tok_len(T[ti]);
            sum = wave_incl_scan(sum);
            if (lane == 63) s_part[wave] = sum;
        }
        __syncthreads();
        u32 base = 0, all = 0;
        for (u32 w = 0; w < (u32)NW; w++) { const u32 v = s_part[w]; all += v; if (w < wave) base += v; }
        u32 t_next = w_lo + lane < w_hi ? T[w_lo + lane] : 0u;            // (the next 64 tokens are on their way while these are worked on)
        for (u32 t0 = w_lo; t0 < w_hi; t0 += 64u) {
            const u32 ti = t0 + lane;
            u32 t = 0, len = 0, tag = TAG_OPERAND;
            const u32 t_cur = t_next;
            t_next = ti + 64u < w_hi ? T[ti + 64u] : 0u;
            if (ti < w_hi) { t = t_cur; tag = t >> 30; len = tok_len(t); }
            const u32 inc = wave_incl_scan(len);
            const u32 dst = base + inc - len;
            // per token: the first bytes by its own lane, the rest of a long one by the whole wave
            u32 done = 0;
            if (len && dst + len <= total) {
                if (tag == TAG_MATCH) {
                    const u32 dist = (t & 0x7FFFu) + 1u;
                    const u32 n0 = len < 16u ? len : 16u;      // (longer ones -- quality runs -- are finished by the whole wave below; 8 / 16 / 24 / 32: fill 16.5 / 13.7 / 14.4 / 15.0)
                    if (dist > dst) { done = len; bad = true; if (atomicCAS(&err[0], 0u, err_base + b + 1u) == 0u) err[1] = (u32)(-mlst_inflate::E_DISTANCE); }
                    else {
                        // (a match that overlaps itself -- a run of one quality value: dist 1, up to 258 long -- is left to the wave whole:
                        // its bytes repeat the `dist` bytes in front of it and every one of them is pointed INTO that period, one link
                        // instead of len / dist; per lane that is a wrap test per byte, measured slower: profiles/round5/inflate.md 3)
                        if (dist >= len) { for (u32 k = 0; k < n0; k++) ptr[dst + k] = (u16)(dst + k - dist); done = n0; }
                    }
                } else if (tag == TAG_LIT) { ptr[dst] = (u16)(LIT_BASE | (t & 0xFFu)); done = 1; }
            } else if (len) { done = len; bad = true; if (atomicCAS(&err[0], 0u, err_base + b + 1u) == 0u) err[1] = (u32)(-mlst_inflate::E_OUTPUT); }
            const u32 nxt = (tag == TAG_RAW && ti + 1 < nt) ? T[ti + 1] : 0u;      // (the operand of a stored run)
            u64 todo = __ballot(done < len);
            while (todo) {
                const int k = __ffsll((long long)todo) - 1; todo &= todo - 1;
                // (k is wave-uniform: v_readlane, not __shfl -- five trips through the LDS crossbar per long token were most of the fill phase)
                const u32 tt = (u32)__builtin_amdgcn_readlane((int)t, k), tl = (u32)__builtin_amdgcn_readlane((int)len, k), td = (u32)__builtin_amdgcn_readlane((int)dst, k),
                          t_done = (u32)__builtin_amdgcn_readlane((int)done, k);
                const u32 t_nxt = (u32)__builtin_amdgcn_readlane((int)nxt, k);
                if ((tt >> 30) == TAG_MATCH) {
                    const u32 dist = (tt & 0x7FFFu) + 1u;
                    if (dist < tl) {      // periodic (see above): j mod dist by the reciprocal (j < 65,536, dist < 65,536: one correction step either way is enough)
                        const float rd = 1.0f / (float)dist;
                        for (u32 j = t_done + lane; j < tl; j += 64) {
                            int r = (int)j - (int)((u32)((float)j * rd) * dist);
                            r += r < 0 ? (int)dist : 0; r -= r >= (int)dist ? (int)dist : 0;
                            ptr[td + j] = (u16)(td - dist + (u32)r);
                        }
                    } else for (u32 j = t_done + lane; j < tl; j += 64) ptr[td + j] = (u16)(td + j - dist);
                } else {      // stored run: its bytes straight from the compressed buffer
                    const u32 src = t_nxt & 0x3FFFFFFFu;
                    if ((t_nxt >> 30) == TAG_OPERAND && (u64)src + tl <= (u64)B.in_len)
                        for (u32 j = lane; j < tl; j += 64) ptr[td + j] = (u16)(LIT_BASE | I[src + j]);
                    else { bad = true; if (lane == 0 && atomicCAS(&err[0], 0u, err_base + b + 1u) == 0u) err[1] = (u32)(-mlst_inflate::E_STORED); }
                }
            }
            base += (u32)__builtin_amdgcn_readlane((int)inc, 63);
        }
        __syncthreads();
        base = all;
        if (base != total) bad = true;
        // a block with an inconsistent token (flagged above) is not jumped or written out: its pointers may be stale
        if (__syncthreads_or(bad)) { if (tid == 0 && atomicCAS(&err[0], 0u, err_base + b + 1u) == 0u) err[1] = (u32)(-mlst_inflate::E_SHORT); continue; }
        // ---- pointer jumping: every pointer ends as a literal.  Thread t owns the byte PAIRS 2 (t + NT j) (one 32-bit read and
        // write for two pointers; only their targets are read one by one); `live` has a bit per j whose pair still holds a
        // pointer.  (The quality lines of a FASTQ block copy each other record after record: half the bytes of a block sit
        // ~200 links deep and stay for 8-9 rounds.)
#if defined(MLST_PTR_TRACE)
        const unsigned long long tr1 = __builtin_readcyclecounter(); int tr_rounds = 0;
#endif
        u32* const ptr2 = reinterpret_cast<u32*>(ptr);
        const u32 n_pairs = (total + 1u) >> 1;
        if (tid == 0) { ptr[total] = (u16)LIT_BASE; ptr[total + 1] = (u16)LIT_BASE; }      // (the odd block's last pair: its second half is no pointer; total <= LIT_BASE)
        __syncthreads();
        u32 live = 0;
        for (u32 j = 0; tid + j * NT < n_pairs; j++) {
            const u32 w = ptr2[tid + j * NT];
            if ((w & 0xFFFFu) < LIT_BASE || (w >> 16) < LIT_BASE) live |= 1u << j;
        }
        for (int round = 0; round < 18; round++) {
            u32 next = 0;
            constexpr int PJ = 4;      // pairs per turn (8 measured no faster: profiles/round5/inflate.md)
            for (u32 m = live; __ballot(m != 0); ) {                    // (every lane walks its own set bits, PJ pairs at a time: their LDS reads are independent)
                u32 pj[PJ], w[PJ], r0[PJ], r1[PJ]; bool on[PJ];
                #pragma unroll
                for (int k = 0; k < PJ; k++) {
                    on[k] = m != 0;
                    pj[k] = on[k] ? (u32)__ffs((int)m) - 1u : 0u;
                    m &= m - 1;
                }
                #pragma unroll
                for (int k = 0; k < PJ; k++) w[k] = on[k] ? ptr2[tid + pj[k] * NT] : (LIT_BASE | (LIT_BASE << 16));
                #pragma unroll
                for (int k = 0; k < PJ; k++) {      // (a half that is a literal already reads slot 0 and keeps its value)
                    const u32 q0 = w[k] & 0xFFFFu, q1 = w[k] >> 16;
                    r0[k] = ptr[q0 < LIT_BASE ? q0 : 0u]; r1[k] = ptr[q1 < LIT_BASE ? q1 : 0u];
                }
                #pragma unroll
                for (int k = 0; k < PJ; k++) {
                    const u32 q0 = w[k] & 0xFFFFu, q1 = w[k] >> 16;
                    const u32 n0 = q0 < LIT_BASE ? r0[k] : q0, n1 = q1 < LIT_BASE ? r1[k] : q1;
                    if (on[k]) {
                        ptr2[tid + pj[k] * NT] = n0 | (n1 << 16);
                        if (n0 < LIT_BASE || n1 < LIT_BASE) next |= 1u << pj[k];
                    }
                }
            }
            live = next;
            const int any = __syncthreads_or(live != 0);
#if defined(MLST_PTR_TRACE)
            tr_rounds++;
#endif
            if (!any) break;
        }
#if defined(MLST_PTR_TRACE)
        const unsigned long long tr2 = __builtin_readcyclecounter();
#endif
        // ---- write-out: sixteen bytes per thread and turn (two 16-byte LDS reads, one 16-byte store) where the block's text starts
        // on a 16-byte boundary (every block of a bgzip'd file does), byte by byte otherwise; newlines counted per cell
        typedef unsigned int v4 __attribute__((ext_vector_type(4)));
        const u64 cell0 = B.out_off >> nl_shift;
        u32 n16 = 0;
        if ((((uintptr_t)O) & 15u) == 0) {
            n16 = total >> 4;
            const v4* p4 = reinterpret_cast<const v4*>(ptr);
            for (u32 c = tid; c < n16; c += NT) {
                const v4 a = p4[2 * c], e = p4[2 * c + 1];
                v4 q;
                q.x = __builtin_amdgcn_perm(a.y, a.x, 0x06040200u); q.y = __builtin_amdgcn_perm(a.w, a.z, 0x06040200u);
                q.z = __builtin_amdgcn_perm(e.y, e.x, 0x06040200u); q.w = __builtin_amdgcn_perm(e.w, e.z, 0x06040200u);
                __builtin_nontemporal_store(q, reinterpret_cast<v4*>(O + 16 * c));
                if (nl) {
                    const u32 cnt = nl_count4(q.x) + nl_count4(q.y) + nl_count4(q.z) + nl_count4(q.w);
                    if (cnt) atomicAdd(&s_nl[(u32)(((B.out_off + 16ull * c) >> nl_shift) - cell0)], cnt);
                }
            }
        }
        for (u32 p = n16 * 16 + tid; p < total; p += NT) {
            const u8 c = (u8)ptr[p]; O[p] = c;
            if (nl && c == (u8)'\n') atomicAdd(&s_nl[(u32)(((B.out_off + p) >> nl_shift) - cell0)], 1u);
        }
        __syncthreads();
        if (nl && tid < 20 && s_nl[tid]) atomicAdd(&nl[cell0 + tid], s_nl[tid]);
        __syncthreads();
#if defined(MLST_PTR_TRACE)
        if (tid == 0) { const unsigned long long tr3 = __builtin_readcyclecounter();
                        atomicAdd(&err[2], (u32)((tr1 - tr0) >> 6)); atomicAdd(&err[3], (u32)((tr2 - tr1) >> 6)); atomicAdd(&err[4], (u32)((tr3 - tr2) >> 6)); atomicAdd(&err[5], (u32)tr_rounds); }
#endif
    }
}
#endif  // __HIP_DEVICE_COMPILE__

}  // namespace inflate_lane
