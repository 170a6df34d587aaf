// mlst_engine.hip -- MI355X (gfx950) MLST-typing engine: HIP kernels + the C-ABI of include/mlst.h.
//
// Path (reference file:line in include/mlst.h and DESIGN.md):
//   K0 k_pack        ASCII reads -> 2-bit rows + Phred rows                (FASTQ/SAM fields)
//   K1 k_sieve       streaming seed sieve over ALL reads (HBM bound)       (bowtie2 seeding)
//   K2 k_seed        exact 20-mer seeds -> (read, locus, strand, diag)     (bowtie2 seeding)
//   K3 k_extend      ungapped XOR/popcount extension vs every allele       (bowtie2 -a extension)
//   K4 k_banded      banded affine Smith-Waterman for indel-broken pairs   (bowtie2 gapped DP)
//   K5 k_accumulate  per-allele {sum AS, hits}, per-locus read length      (metamlst.py:101-130)
//   K6 k_pileup*     base counts per column of the chosen alleles          (cmseq/pysam pileup)
//   K7 k_hamming     stringDiff scan over the alleles of one locus         (metamlst-merge.py:177-181)
// Integer work only; no MFMA.  Wave = 64 lanes.  There is no CPU fallback in this file.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mlst.h"
#include "mlst_policy.h"

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;
typedef unsigned char u8;
typedef long long i64;

#define RW  (MLST_MAX_READ_LEN / 16)     // words of a retained read row (20)
#define RQ  MLST_MAX_READ_LEN            // bytes of a retained quality row
#define KEY_EMPTY 0xFFFFFFFFFFFFFFFFull
#define MAX_W 16                          // largest supported band half width
#define NEGP MLST_P_NEG
#define P0   MLST_P0

// result word of one (item, allele) pair
#define R_REC   0x80000000u
#define R_NEEDDP 0x40000000u
#define R_USEDDP 0x20000000u
__host__ __device__ inline u32 pack_result(int score, int xm, int xo) {
    return (u32)(score & 0x3FF) | ((u32)(xm & 0xFF) << 10) | ((u32)(xo & 0x7F) << 18);
}

// ------------------------------------------------------------------ hashing (host + device)
__host__ __device__ inline u32 sieve_bucket_hash(u32 lo, u32 hi) {
    u32 h = (lo ^ (hi * 0x9E3779B1u)) * 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__host__ __device__ inline u32 sieve_fp(u32 lo, u32 hi) {
    u32 h = (lo * 0x27D4EB2Fu) ^ (hi * 0x165667B1u);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    u32 fp = h & 0xFFFFu;
    return fp ? fp : 1u;
}
__host__ __device__ inline u32 table_hash(u32 lo, u32 hi) {
    u32 h = (lo + hi * 0x7FEB352Du) * 0x846CA68Bu;
    h ^= h >> 15; h *= 0x9E3779B1u; h ^= h >> 14;
    return h;
}

// ------------------------------------------------------------------ device-side views
struct LocusDev {
    u64 arena_off;      // word offset of this locus in the transposed 2-bit arena
    u64 nmask_off;      // word offset in the N-mask arena (valid iff has_n)
    u32 a_begin, n_alleles, n_pad;   // n_pad = n_alleles rounded up to 64 (row stride of the transposed arena)
    u32 words;          // 2-bit words per allele (ceil(max_len/16) + 2 zero words)
    u32 nwords;         // N-mask words per allele (ceil(max_len/32) + 1)
    u32 has_n, species, max_len;
};
struct ItemDev {         // one (read, locus, strand, diagonal) unit of extension work
    u64 res_off;         // offset of its result row in the pair-result arena
    u32 ret;             // retained-read slot
    u32 locus;
    int diag;
    u16 strand, votes;
};
struct Counters {
    u64 n_cand, n_ret, n_items, n_res, n_dp, items_done, dp_done, n_pl_dp;
    u64 err;             // bit0 retained overflow, bit1 item overflow, bit2 result overflow, bit3 dp overflow
    u64 cnt[MLST_CNT_N];
};
struct KParams {
    int minscore, max_xm, min_read_len, minqual, match_bonus, n_penalty, open_p, ext_p, gbar, band_w, trig, quirk;
};
struct EngineDev {
    // reference
    const u32* arena; const u32* nmask; const u16* allele_len; const u32* allele_locus; const LocusDev* loci;
    const uint4* sieve; u32 sieve_mask;
    const u64* keys; const u32* vals; const u32* posts; u32 table_mask;
    const int* floor_tab; const u8* pen_tab;
    u32 n_alleles, n_loci;
    // sample state
    long long* sum_score; u32* n_hits; u64* locus_len; u64* locus_first;
    Counters* ctr;
    u32* ret_bases; u8* ret_quals; u16* ret_len; u64* ret_ridx; u32* ret_nrec;
    ItemDev* items; u32* res; u64* dp_list;
    u64 cap_ret, cap_items, cap_res, cap_dp;
};

// ------------------------------------------------------------------ K0: pack
// One thread packs 16 bases (one 32-bit word) and their 16 Phred bytes.  lens[r] bit 15 = read has a non-ACGT base.
__global__ __launch_bounds__(256) void k_pack(const u8* __restrict__ bases, const u8* __restrict__ quals,
                                               const u64* __restrict__ off, u64 n_reads, u32* __restrict__ packed,
                                               u8* __restrict__ qrows, u16* __restrict__ lens, u32 wpr, u32 qstride) {
    u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 total = n_reads * wpr;
    for (; gid < total; gid += (u64)gridDim.x * blockDim.x) {
        u64 r = gid / wpr; u32 w = (u32)(gid - r * wpr);
        u64 o = off[r]; u32 n = (u32)(off[r + 1] - o);
        u32 word = 0; u32 anyn = 0;
        for (int k = 0; k < 16; k++) {
            u32 i = w * 16 + k;
            if (i < n) {
                u8 c = bases[o + i]; u32 b; u32 isn = 0;
                switch (c) { case 'A': case 'a': b = 0; break; case 'C': case 'c': b = 1; break;
                             case 'G': case 'g': b = 2; break; case 'T': case 't': b = 3; break; default: b = 0; isn = 1; }
                word |= b << (2 * k);
                int q = (int)quals[o + i] - 33; q = q < 0 ? 0 : (q > 127 ? 127 : q);
                if (i < qstride) qrows[r * qstride + i] = (u8)q | (u8)(isn << 7);
                anyn |= isn;
            } else if (i < qstride) qrows[r * qstride + i] = 0;
        }
        packed[gid] = word;
        // lens[r] was stored by k_pack_lens (earlier launch on the same stream); OR in the "has N" flag
        if (anyn) atomicOr((u32*)(lens + (r & ~1ull)), (r & 1) ? 0x80000000u : 0x00008000u);
    }
}
// first pass of the pack: plain lengths (d_lens must hold n_reads rounded up to an even count)
__global__ __launch_bounds__(256) void k_pack_lens(const u64* __restrict__ off, u64 n_reads, u16* __restrict__ lens) {
    u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; r < n_reads; r += (u64)gridDim.x * blockDim.x) lens[r] = (u16)(off[r + 1] - off[r]);
}

// ------------------------------------------------------------------ K1: seed sieve (the streaming kernel)
// Each 256-thread block stages 256 packed read rows through LDS with coalesced 16-byte loads; each lane then
// owns one read: a seed is the 20-mer at every 16th base = word t plus the low byte of word t+1.  A seed is
// looked up in the sieve: 16-byte buckets of eight 16-bit fingerprints (one 16-byte load per seed).  Reads with
// any hit are compacted into the candidate list with a wave ballot and one atomic per wave.
__device__ inline bool bucket_has(uint4 b, u32 fp, bool& full) {
    u32 pat = fp * 0x00010001u;
    u32 x0 = b.x ^ pat, x1 = b.y ^ pat, x2 = b.z ^ pat, x3 = b.w ^ pat;
    // zero-halfword test
    u32 z = ((x0 - 0x00010001u) & ~x0) | ((x1 - 0x00010001u) & ~x1) | ((x2 - 0x00010001u) & ~x2) | ((x3 - 0x00010001u) & ~x3);
    u32 e = ((b.x - 0x00010001u) & ~b.x) | ((b.y - 0x00010001u) & ~b.y) | ((b.z - 0x00010001u) & ~b.z) | ((b.w - 0x00010001u) & ~b.w);
    full = (e & 0x80008000u) == 0;
    return (z & 0x80008000u) != 0;
}

#define SIEVE_MAXP 10        // pairs of words per row (RW / 2)
__global__ __launch_bounds__(256) void k_sieve(const u32* __restrict__ packed, const u16* __restrict__ lens, u64 n_reads,
                                                u32 wpr /* even */, const uint4* __restrict__ sieve, u32 smask,
                                                u32* __restrict__ cand, Counters* __restrict__ ctr) {
    extern __shared__ __attribute__((aligned(16))) u32 s_rows[];
    const int tid = threadIdx.x;
    const u32 S2 = wpr >> 1;
    u64 n_blocks = (n_reads + 255) / 256;
    for (u64 blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        u64 r0 = blk * 256;
        u32 nr = (u32)((n_reads - r0) < 256 ? (n_reads - r0) : 256);
        // stage rows: nr*wpr words, 16-byte vectors (r0*wpr*4 is 16-byte aligned because r0 % 256 == 0)
        u32 nvec = (nr * wpr + 3) >> 2;
        const uint4* g4 = reinterpret_cast<const uint4*>(packed + r0 * wpr);
        uint4* s4 = reinterpret_cast<uint4*>(s_rows);
        u32 full_vec = (nr * wpr) >> 2;
        for (u32 v = tid; v < nvec; v += 256) {
            if (v < full_vec) s4[v] = g4[v];
            else { // ragged tail (only when nr*wpr % 4 != 0)
                const u32* g = packed + r0 * wpr; u32 base = v * 4, lim = nr * wpr;
                uint4 t; t.x = base < lim ? g[base] : 0; t.y = base + 1 < lim ? g[base + 1] : 0;
                t.z = base + 2 < lim ? g[base + 2] : 0; t.w = 0; s4[v] = t;
            }
        }
        __syncthreads();
        bool hit = false;
        if ((u32)tid < nr) {
            u32 n = lens[r0 + tid] & 0x7FFFu;
            int nseeds = n >= MLST_SEED_LEN ? (int)((n - MLST_SEED_LEN) / MLST_SEED_STEP) + 1 : 0;
            const uint2* row = reinterpret_cast<const uint2*>(s_rows + (u32)tid * wpr);
            u32 pending = 0;          // seeds whose first bucket was full without a match (rare)
            u32 prev = 0;
            #pragma unroll
            for (int t2 = 0; t2 < SIEVE_MAXP; t2++) {
                if ((u32)t2 < S2) {
                    uint2 w = row[t2];
                    // seed t = 2*t2-1 : words (prev, w.x) ; seed t = 2*t2 : words (w.x, w.y)
                    if (t2 > 0 && 2 * t2 - 1 < nseeds) {
                        u32 lo = prev, hi = w.x & 0xFFu; bool full;
                        uint4 b = sieve[sieve_bucket_hash(lo, hi) & smask];
                        bool f = bucket_has(b, sieve_fp(lo, hi), full);
                        hit |= f; if (!f && full) pending |= 1u << (2 * t2 - 1);
                    }
                    if (2 * t2 < nseeds) {
                        u32 lo = w.x, hi = w.y & 0xFFu; bool full;
                        uint4 b = sieve[sieve_bucket_hash(lo, hi) & smask];
                        bool f = bucket_has(b, sieve_fp(lo, hi), full);
                        hit |= f; if (!f && full) pending |= 1u << (2 * t2);
                    }
                    prev = w.y;
                }
            }
            while (pending && !hit) {   // overflow chain: the key may sit in a following bucket
                int t = __ffs(pending) - 1; pending &= pending - 1;
                const u32* rw = s_rows + (u32)tid * wpr;
                u32 lo = rw[t], hi = rw[t + 1] & 0xFFu; u32 fp = sieve_fp(lo, hi);
                u32 bi = sieve_bucket_hash(lo, hi) & smask;
                for (int step = 0; step < 64; step++) {
                    bi = (bi + 1) & smask; bool full; uint4 b = sieve[bi];
                    if (bucket_has(b, fp, full)) { hit = true; break; }
                    if (!full) break;
                }
            }
        }
        u64 mask = __ballot(hit);
        if (mask) {
            int lane = tid & 63;
            u64 base = 0;
            if (lane == (__ffsll((long long)mask) - 1)) base = atomicAdd(&ctr->n_cand, (u64)__popcll(mask));
            base = __shfl(base, __ffsll((long long)mask) - 1);
            if (hit) cand[base + __popcll(mask & ((1ull << lane) - 1))] = (u32)(r0 + tid);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ K2: exact seeds -> work items
struct Bin { u32 locus; int diag; u16 strand, votes; };

__device__ inline bool table_find(const EngineDev& E, u32 lo, u32 hi, u32& val) {
    u64 key = (u64)lo | ((u64)hi << 32);
    u32 h = table_hash(lo, hi) & E.table_mask;
    for (u32 step = 0; step <= E.table_mask; step++) {
        u64 k = E.keys[h];
        if (k == key) { val = E.vals[h]; return true; }
        if (k == KEY_EMPTY) return false;
        h = (h + 1) & E.table_mask;
    }
    return false;
}

__global__ __launch_bounds__(256) void k_seed(EngineDev E, const u32* __restrict__ packed, const u8* __restrict__ qrows,
                                               const u16* __restrict__ lens, u32 wpr, u32 qstride, u64 read_base,
                                               const u32* __restrict__ cand) {
    __shared__ Bin s_bins[256][MLST_MAX_CAND];
    __shared__ Bin s_items[256][MLST_MAX_CAND];
    const int tid = threadIdx.x;
    u64 n_cand = E.ctr->n_cand;
    for (u64 c = (u64)blockIdx.x * 256 + tid; c < n_cand; c += (u64)gridDim.x * 256) {
        u32 r = cand[c];
        u32 lw = lens[r]; u32 n = lw & 0x7FFFu; bool has_n = (lw & 0x8000u) != 0;
        const u32* row = packed + (u64)r * wpr;
        const u8* qrow = qrows + (u64)r * qstride;
        Bin* bins = s_bins[tid]; int nb = 0;
        int nseeds = n >= MLST_SEED_LEN ? (int)((n - MLST_SEED_LEN) / MLST_SEED_STEP) + 1 : 0;
        for (int t = 0; t < nseeds; t++) {
            int o = t * MLST_SEED_STEP;
            if (has_n) { bool bad = false; for (int k = 0; k < MLST_SEED_LEN; k++) bad |= (qrow[o + k] & 0x80) != 0; if (bad) continue; }
            u32 lo = row[t], hi = row[t + 1] & 0xFFu, val;
            if (!table_find(E, lo, hi, val)) continue;
            u32 pstart, pcount; u32 single = 0;
            if (val & 0x80000000u) { single = val & 0x7FFFFFFFu; pstart = 0; pcount = 1; }
            else { pstart = val >> 5; pcount = val & 31u; }
            for (u32 p = 0; p < pcount; p++) {
                u32 post = (val & 0x80000000u) ? single : E.posts[pstart + p];
                u32 locus = post >> 13, strand = (post >> 12) & 1; int pos = (int)(post & 0xFFFu);
                int diag = strand ? pos + MLST_SEED_LEN + o - (int)n : pos - o;
                int k; for (k = 0; k < nb; k++) if (bins[k].locus == locus && bins[k].strand == strand && bins[k].diag == diag) break;
                if (k < nb) bins[k].votes++;
                else if (nb < MLST_MAX_CAND) { bins[nb].locus = locus; bins[nb].strand = (u16)strand; bins[nb].diag = diag; bins[nb].votes = 1; nb++; }
            }
        }
        if (nb == 0) continue;
        // one item per (locus, strand): most votes, then the smaller diagonal; first-seen order
        Bin* items = s_items[tid]; int ni = 0;
        for (int k = 0; k < nb; k++) {
            int u; for (u = 0; u < ni; u++) if (items[u].locus == bins[k].locus && items[u].strand == bins[k].strand) break;
            if (u == ni) items[ni++] = bins[k];
            else if (bins[k].votes > items[u].votes || (bins[k].votes == items[u].votes && bins[k].diag < items[u].diag)) items[u] = bins[k];
        }
        int no = 0;
        for (int u = 0; u < ni; u++) if (items[u].votes >= MLST_MIN_VOTES) items[no++] = items[u];
        if (no == 0) continue;
        u64 slot = atomicAdd(&E.ctr->n_ret, 1ull);
        if (slot >= E.cap_ret) { atomicOr(&E.ctr->err, 1ull); continue; }
        for (u32 w = 0; w < RW; w++) E.ret_bases[slot * RW + w] = w < wpr ? row[w] : 0u;
        for (u32 i = 0; i < RQ; i++) E.ret_quals[slot * RQ + i] = (i < n && i < qstride) ? qrow[i] : (u8)0;
        E.ret_len[slot] = (u16)lw; E.ret_ridx[slot] = read_base + r; E.ret_nrec[slot] = 0;
        u64 ib = atomicAdd(&E.ctr->n_items, (u64)no);
        for (int u = 0; u < no; u++) {
            if (ib + u >= E.cap_items) { atomicOr(&E.ctr->err, 2ull); break; }
            u64 ro = atomicAdd(&E.ctr->n_res, (u64)E.loci[items[u].locus].n_pad);
            if (ro + E.loci[items[u].locus].n_pad > E.cap_res) { atomicOr(&E.ctr->err, 4ull); }
            ItemDev it; it.res_off = ro; it.ret = (u32)slot; it.locus = items[u].locus; it.diag = items[u].diag;
            it.strand = items[u].strand; it.votes = items[u].votes;
            E.items[ib + u] = it;
        }
    }
}

// ------------------------------------------------------------------ shared read-orientation helpers
// Oriented read i (after reverse-complement when strand = 1) lives at source position s = strand ? n-1-i : i.
__device__ inline u32 src_base(const u32* rb, int s) { return (rb[s >> 4] >> (2 * (s & 15))) & 3u; }

// Build the oriented read of an item in LDS: 2-bit words, N bits (1 per base) and the per-position
// mismatch penalty (bowtie2 --mp 6,2 quality-aware; --np 1 for N).
__device__ inline void stage_read(const EngineDev& E, const KParams& P, const ItemDev& it, int n,
                                  u32* s_rw, u32* s_rn, u8* s_pen, u8* s_q, int tid, int nthreads) {
    const u32* rb = E.ret_bases + (u64)it.ret * RW;
    const u8* rq = E.ret_quals + (u64)it.ret * RQ;
    for (int w = tid; w < RW; w += nthreads) {
        u32 word = 0;
        for (int k = 0; k < 16; k++) { int i = w * 16 + k; if (i < n) { int s = it.strand ? n - 1 - i : i; u32 b = src_base(rb, s); if (it.strand) b ^= 3u; word |= b << (2 * k); } }
        s_rw[w] = word;
    }
    for (int w = tid; w < RW / 2; w += nthreads) {
        u32 word = 0;
        for (int k = 0; k < 32; k++) { int i = w * 32 + k; if (i < n) { int s = it.strand ? n - 1 - i : i; word |= (u32)(rq[s] >> 7) << k; } }
        s_rn[w] = word;
    }
    for (int i = tid; i < n; i += nthreads) {
        int s = it.strand ? n - 1 - i : i; u8 qb = rq[s];
        s_pen[i] = (qb & 0x80) ? (u8)P.n_penalty : E.pen_tab[qb & 0x7F];
        if (s_q) s_q[i] = qb;
    }
}

__device__ inline u32 arena_word(const EngineDev& E, const LocusDev& L, int q, u32 a_local) {
    return (q >= 0 && q < (int)L.words) ? E.arena[L.arena_off + (u64)q * L.n_pad + a_local] : 0u;
}
__device__ inline u32 nmask_word(const EngineDev& E, const LocusDev& L, int q, u32 a_local) {
    return (q >= 0 && q < (int)L.nwords) ? E.nmask[L.nmask_off + (u64)q * L.n_pad + a_local] : 0u;
}
__device__ inline bool allele_is_n(const EngineDev& E, const LocusDev& L, int j, u32 a_local) {
    return L.has_n && ((nmask_word(E, L, j >> 5, a_local) >> (j & 31)) & 1u);
}
// spread the low 16 bits of x to the even bit positions
__device__ inline u32 spread16(u32 x) {
    x &= 0xFFFFu; x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u; return x;
}

// Ungapped local alignment of the staged read against allele a_local on diagonal d (Kadane over the mismatch
// positions of the XOR of 2-bit words).  Returns the packed best value; mm_total = mismatching columns of the
// whole overlap; [bs,be) = aligned read span.  Same recurrence as oracle align_ungapped.
__device__ inline int ungapped(const EngineDev& E, const KParams& P, const LocusDev& L, u32 a_local, int m, int n, int d,
                               const u32* s_rw, const u32* s_rn, const u8* s_pen, bool read_has_n,
                               int& mm_total, int& bs, int& be) {
    int i0 = d < 0 ? -d : 0, i1 = (m - d) < n ? (m - d) : n;
    mm_total = 0; bs = be = i0;
    if (i1 <= i0) return P0;
    const int MA = P.match_bonus << MLST_P_SHIFT;
    int cur = P0, best = P0, cs = i0, last = i0;
    int t0 = i0 >> 4, t1 = (i1 + 15) >> 4;
    // allele word holding allele base (16*t0 + d)
    int g = 16 * t0 + d; int q = g >> 4; int r2 = (g & 15) * 2;   // g >> 4 floors for negatives (arithmetic shift)
    u32 A0 = arena_word(E, L, q, a_local);
    for (int t = t0; t < t1; t++, q++) {
        u32 A1 = arena_word(E, L, q + 1, a_local);
        u32 ash = r2 ? ((A0 >> r2) | (A1 << (32 - r2))) : A0;
        A0 = A1;
        u32 x = s_rw[t] ^ ash;
        u32 mmw = (x | (x >> 1)) & 0x55555555u;
        u32 anw = 0;
        if (read_has_n) mmw |= spread16(s_rn[t >> 1] >> ((t & 1) * 16));
        if (L.has_n) {   // allele N bits for allele bases g..g+15 (g = 16t + d)
            int gg = 16 * t + d; int nq = gg >> 5, nr = gg & 31;
            u32 n0 = nmask_word(E, L, nq, a_local), n1 = nmask_word(E, L, nq + 1, a_local);
            u32 nb = nr ? ((n0 >> nr) | (n1 << (32 - nr))) : n0;
            anw = spread16(nb); mmw |= anw;
        }
        int lo = i0 - 16 * t; lo = lo < 0 ? 0 : lo;
        int hi = i1 - 16 * t; hi = hi > 16 ? 16 : hi;
        u32 vm = (hi >= 16 ? 0xFFFFFFFFu : ((1u << (2 * hi)) - 1u)) & ~((1u << (2 * lo)) - 1u);
        mmw &= vm;
        mm_total += __popc(mmw);
        while (mmw) {
            int bit = __ffs(mmw) - 1; mmw &= mmw - 1;
            int i = 16 * t + (bit >> 1);
            cur += (i - last) * MA;
            if (cur > best) { best = cur; bs = cs; be = i; }
            int pen = ((anw >> bit) & 1u) ? P.n_penalty : (int)s_pen[i];
            cur -= (pen << MLST_P_SHIFT) + 1;
            if (cur <= P0) { cur = P0; cs = i + 1; }
            last = i + 1;
        }
    }
    cur += (i1 - last) * MA;
    if (cur > best) { best = cur; bs = cs; be = i1; }
    return best;
}

// ------------------------------------------------------------------ K3: extension of every item against every allele of its locus
__global__ __launch_bounds__(256) void k_extend(EngineDev E, KParams P) {
    __shared__ u32 s_rw[RW + 2]; __shared__ u32 s_rn[RW / 2 + 1]; __shared__ u8 s_pen[RQ]; __shared__ u32 s_cnt[4];
    const int tid = threadIdx.x;
    const u64 begin = E.ctr->items_done, end = E.ctr->n_items < E.cap_items ? E.ctr->n_items : E.cap_items;
    for (u64 ii = begin + blockIdx.x; ii < end; ii += gridDim.x) {
        ItemDev it = E.items[ii];
        const LocusDev L = E.loci[it.locus];
        u32 lw = E.ret_len[it.ret]; int n = (int)(lw & 0x7FFFu); bool read_has_n = (lw & 0x8000u) != 0;
        __syncthreads();
        stage_read(E, P, it, n, s_rw, s_rn, s_pen, nullptr, tid, 256);
        if (tid == 0) { s_rw[RW] = s_rw[RW + 1] = 0; }
        __syncthreads();
        if (it.res_off + L.n_pad > E.cap_res) continue;      // flagged by k_seed
        const int floor_n = E.floor_tab[n];
        u32 nrec = 0;
        for (u32 a = tid; a < L.n_alleles; a += 256) {
            int m = (int)E.allele_len[L.a_begin + a];
            int mm, bs, be;
            int best = ungapped(E, P, L, a, m, n, it.diag, s_rw, s_rn, s_pen, read_has_n, mm, bs, be);
            int score = best >> MLST_P_SHIFT, xm = 255 - (best & 0xFF), xo = 127 - ((best >> 8) & 0x7F);
            bool need_dp = P.trig < 0 ? true : (mm > P.trig && score >= floor_n);
            u32 r = pack_result(score, xm, xo);
            if (need_dp) {
                r |= R_NEEDDP;
                u64 slot = atomicAdd(&E.ctr->n_dp, 1ull);
                if (slot < E.cap_dp) E.dp_list[slot] = (ii << 20) | (u64)a; else atomicOr(&E.ctr->err, 8ull);
            } else if (score >= floor_n && score > 0) { r |= R_REC; nrec++; }
            E.res[it.res_off + a] = r;
        }
        // records of this item -> per-read record count (decides XM vs XO column, Q1)
        for (int o = 32; o > 0; o >>= 1) nrec += __shfl_down(nrec, o);
        if ((tid & 63) == 0) s_cnt[tid >> 6] = nrec;
        __syncthreads();
        if (tid == 0) { u32 tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]; if (tot) atomicAdd(&E.ret_nrec[it.ret], tot); }
    }
}

// ------------------------------------------------------------------ K4: banded affine Smith-Waterman (one lane per pair)
// Band arrays live in registers (fully unrolled to 2*MAX_W+1 with a uniform guard).  Same recurrence and tie
// rules as oracle align_banded.  TB != nullptr additionally stores the traceback byte of every cell.
template <bool TRACE>
__device__ inline int banded(const EngineDev& E, const KParams& P, const ItemDev& it, const LocusDev& L, u32 a_local,
                             int n, u8* TB, int& bi, int& bb) {
    const int W = P.band_w, BW = 2 * W + 1, G = P.gbar, d = it.diag;
    const int m = (int)E.allele_len[L.a_begin + a_local];
    const int OPEN = P.open_p, EXT = P.ext_p, MA = P.match_bonus << MLST_P_SHIFT;
    const u32* rb = E.ret_bases + (u64)it.ret * RW;
    const u8* rq = E.ret_quals + (u64)it.ret * RQ;
    int Hp[2 * MAX_W + 2], Fp[2 * MAX_W + 2];
    #pragma unroll
    for (int b = 0; b < 2 * MAX_W + 2; b++) { Hp[b] = P0; Fp[b] = NEGP; }
    int best = P0; bi = -1; bb = -1;
    for (int i = 0; i < n; i++) {
        int s = it.strand ? n - 1 - i : i;
        u8 qb = rq[s]; bool rn = (qb & 0x80) != 0;
        u32 rbase = src_base(rb, s); if (it.strand) rbase ^= 3u;
        int pen = rn ? P.n_penalty : (int)E.pen_tab[qb & 0x7F];
        bool gap_ok = (i >= G && i < n - G);
        int Hleft = P0, Eleft = NEGP;
        int jb = i + d - W;
        #pragma unroll
        for (int b = 0; b < 2 * MAX_W + 1; b++) {
            if (b < BW) {
                int j = jb + b;
                bool exists = (j >= 0 && j < m);
                int jj = exists ? j : 0;
                u32 ab = (arena_word(E, L, jj >> 4, a_local) >> (2 * (jj & 15))) & 3u;
                bool an = exists && allele_is_n(E, L, jj, a_local);
                int delta = (!rn && !an && rbase == ab) ? MA : -(((rn || an) ? P.n_penalty : pen) << MLST_P_SHIFT) - 1;
                int diag = Hp[b] + delta;
                int e = NEGP, f = NEGP; u8 tb = 0;
                if (gap_ok && b > 0) { int e1 = Hleft - OPEN, e2 = Eleft - EXT; if (e2 > e1) { e = e2; tb |= 4; } else e = e1; }
                if (gap_ok && b < BW - 1) { int f1 = Hp[b + 1] - OPEN, f2 = Fp[b + 1] - EXT; if (f2 > f1) { f = f2; tb |= 8; } else f = f1; }
                int h = P0; u8 src = 0;
                if (diag > h) { h = diag; src = 1; }
                if (e > h) { h = e; src = 2; }
                if (f > h) { h = f; src = 3; }
                if (!exists) { h = P0; e = NEGP; f = NEGP; src = 0; tb = 0; }
                if (TRACE) TB[i * (2 * MAX_W + 1) + b] = tb | src;
                if (h > best) { best = h; bi = i; bb = b; }
                Hp[b] = h; Fp[b] = f; Hleft = h; Eleft = e;
            }
        }
    }
    return best;
}

__global__ __launch_bounds__(64) void k_banded(EngineDev E, KParams P) {
    const u64 begin = E.ctr->dp_done, end = E.ctr->n_dp < E.cap_dp ? E.ctr->n_dp : E.cap_dp;
    for (u64 k = begin + (u64)blockIdx.x * 64 + threadIdx.x; k < end; k += (u64)gridDim.x * 64) {
        u64 e = E.dp_list[k]; u64 ii = e >> 20; u32 a = (u32)(e & 0xFFFFFu);
        ItemDev it = E.items[ii];
        const LocusDev L = E.loci[it.locus];
        int n = (int)(E.ret_len[it.ret] & 0x7FFFu);
        int bi, bb;
        int best = banded<false>(E, P, it, L, a, n, nullptr, bi, bb);
        int score = best >> MLST_P_SHIFT, xm = 255 - (best & 0xFF), xo = 127 - ((best >> 8) & 0x7F);
        u32 r = pack_result(score, xm, xo) | R_USEDDP;
        if (score >= E.floor_tab[n] && score > 0) { r |= R_REC; atomicAdd(&E.ret_nrec[it.ret], 1u); }
        E.res[it.res_off + a] = r;
    }
}

// ------------------------------------------------------------------ K5: accumulate (metamlst.py:101-130)
__global__ __launch_bounds__(256) void k_accumulate(EngineDev E, KParams P) {
    __shared__ u32 s_red[4][4];
    const int tid = threadIdx.x;
    const u64 begin = E.ctr->items_done, end = E.ctr->n_items < E.cap_items ? E.ctr->n_items : E.cap_items;
    for (u64 ii = begin + blockIdx.x; ii < end; ii += gridDim.x) {
        ItemDev it = E.items[ii];
        const LocusDev L = E.loci[it.locus];
        if (it.res_off + L.n_pad > E.cap_res) continue;
        int n = (int)(E.ret_len[it.ret] & 0x7FFFu);
        // column 15 of the SAM line is XM when the read has a second record (XS:i present), else XO (Q1)
        bool use_xo = P.quirk && E.ret_nrec[it.ret] == 1;
        u32 tot = 0, ign = 0, acc = 0, dp = 0;
        for (u32 a = tid; a < L.n_alleles; a += 256) {
            u32 r = E.res[it.res_off + a];
            if (r & R_USEDDP) dp++;
            if (!(r & R_REC)) continue;
            int score = (int)(r & 0x3FF), xm = (int)((r >> 10) & 0xFF), xo = (int)((r >> 18) & 0x7F);
            int f15 = use_xo ? xo : xm;
            tot++;
            if (score >= P.minscore && n >= P.min_read_len && f15 <= P.max_xm) {
                atomicAdd((u64*)&E.sum_score[L.a_begin + a], (u64)score);
                atomicAdd(&E.n_hits[L.a_begin + a], 1u);
                acc++;
            } else ign++;
        }
        for (int o = 32; o > 0; o >>= 1) { tot += __shfl_down(tot, o); ign += __shfl_down(ign, o); acc += __shfl_down(acc, o); dp += __shfl_down(dp, o); }
        __syncthreads();
        if ((tid & 63) == 0) { s_red[tid >> 6][0] = tot; s_red[tid >> 6][1] = ign; s_red[tid >> 6][2] = acc; s_red[tid >> 6][3] = dp; }
        __syncthreads();
        if (tid == 0) {
            u32 T = 0, I = 0, A = 0, D = 0;
            for (int w = 0; w < 4; w++) { T += s_red[w][0]; I += s_red[w][1]; A += s_red[w][2]; D += s_red[w][3]; }
            if (T) atomicAdd(&E.ctr->cnt[MLST_CNT_TOTAL_RECORDS], (u64)T);
            if (I) atomicAdd(&E.ctr->cnt[MLST_CNT_IGNORED], (u64)I);
            if (D) atomicAdd(&E.ctr->cnt[MLST_CNT_DP_PAIRS], (u64)D);
            if (A) {   // sequenceBank[locus][QNAME] = len(SEQ), first-seen order (Q6)
                atomicAdd(&E.locus_len[it.locus], (u64)n);
                atomicMin(&E.locus_first[it.locus], E.ret_ridx[it.ret]);
            }
        }
    }
}

__global__ void k_advance(Counters* c, u64 n_reads) {
    u64 ni = c->n_items, nd = c->n_dp;
    c->cnt[MLST_CNT_CANDIDATES] += c->n_cand;
    c->cnt[MLST_CNT_READS_SEEN] += n_reads;
    c->items_done = ni; c->dp_done = nd; c->n_cand = 0;
}

// ------------------------------------------------------------------ K6: pileup against the chosen allele of each locus
// One lane per item.  Ungapped pairs are piled up here; pairs that trigger the banded SW go to k_pileup_dp.
__device__ inline void pile_base(const EngineDev& E, const KParams& P, const ItemDev& it, int n, int i, int j,
                                 u32* counts, u64 colbase) {
    int s = it.strand ? n - 1 - i : i;
    u8 qb = E.ret_quals[(u64)it.ret * RQ + s];
    if ((qb & 0x80) || (int)(qb & 0x7F) < P.minqual) return;
    u32 b = src_base(E.ret_bases + (u64)it.ret * RW, s); if (it.strand) b ^= 3u;
    atomicAdd(&counts[(colbase + (u64)j) * 4 + b], 1u);
}

__global__ __launch_bounds__(64) void k_pileup(EngineDev E, KParams P, const int* __restrict__ locus_chosen,
                                               const u64* __restrict__ locus_colbase, u32* __restrict__ counts,
                                               u64* __restrict__ pl_list) {
    // per-lane staging area in LDS: oriented read words, N bits, penalties
    __shared__ u32 s_rw[64][RW + 2]; __shared__ u32 s_rn[64][RW / 2 + 1]; __shared__ u8 s_pen[64][RQ];
    const int tid = threadIdx.x;
    const u64 end = E.ctr->n_items < E.cap_items ? E.ctr->n_items : E.cap_items;
    for (u64 ii = (u64)blockIdx.x * 64 + tid; ii < end; ii += (u64)gridDim.x * 64) {
        ItemDev it = E.items[ii];
        int ca = locus_chosen[it.locus];
        if (ca < 0) continue;
        const LocusDev L = E.loci[it.locus];
        u32 a = (u32)ca - L.a_begin;
        u32 lw = E.ret_len[it.ret]; int n = (int)(lw & 0x7FFFu);
        stage_read(E, P, it, n, s_rw[tid], s_rn[tid], s_pen[tid], nullptr, 0, 1);
        s_rw[tid][RW] = s_rw[tid][RW + 1] = 0;
        int m = (int)E.allele_len[ca];
        int mm, bs, be;
        int best = ungapped(E, P, L, a, m, n, it.diag, s_rw[tid], s_rn[tid], s_pen[tid], (lw & 0x8000u) != 0, mm, bs, be);
        int score = best >> MLST_P_SHIFT, xm = 255 - (best & 0xFF);
        int floor_n = E.floor_tab[n];
        bool need_dp = P.trig < 0 ? true : (mm > P.trig && score >= floor_n);
        if (need_dp) { u64 slot = atomicAdd(&E.ctr->n_pl_dp, 1ull); pl_list[slot] = ii; continue; }
        if (score < floor_n || score <= 0 || score < P.minscore || xm > P.max_xm) continue;   // BAM_tagFilter AS, XM
        for (int i = bs; i < be; i++) pile_base(E, P, it, n, i, i + it.diag, counts, locus_colbase[it.locus]);
    }
}

__global__ __launch_bounds__(64) void k_pileup_dp(EngineDev E, KParams P, const int* __restrict__ locus_chosen,
                                                  const u64* __restrict__ locus_colbase, u32* __restrict__ counts,
                                                  const u64* __restrict__ pl_list, u8* __restrict__ tb_scratch) {
    const u64 end = E.ctr->n_pl_dp;
    const int BWMAX = 2 * MAX_W + 1;
    u8* TB = tb_scratch + ((u64)blockIdx.x * 64 + threadIdx.x) * (u64)(MLST_MAX_READ_LEN * BWMAX);
    for (u64 k = (u64)blockIdx.x * 64 + threadIdx.x; k < end; k += (u64)gridDim.x * 64) {
        ItemDev it = E.items[pl_list[k]];
        int ca = locus_chosen[it.locus];
        const LocusDev L = E.loci[it.locus];
        u32 a = (u32)ca - L.a_begin;
        int n = (int)(E.ret_len[it.ret] & 0x7FFFu);
        int bi, bb;
        int best = banded<true>(E, P, it, L, a, n, TB, bi, bb);
        int score = best >> MLST_P_SHIFT, xm = 255 - (best & 0xFF);
        if (score < E.floor_tab[n] || score <= 0 || score < P.minscore || xm > P.max_xm) continue;
        const int W = P.band_w, BW = 2 * W + 1;
        int i = bi, b = bb, state = 0;
        u64 colbase = locus_colbase[it.locus];
        while (i >= 0 && b >= 0 && b < BW) {
            u8 t = TB[i * BWMAX + b];
            if (state == 0) {
                int src = t & 3;
                if (src == 0) break;
                if (src == 1) { pile_base(E, P, it, n, i, i + it.diag - W + b, counts, colbase); i--; }
                else if (src == 2) state = 1; else state = 2;
            } else if (state == 1) { int ext = t & 4; b--; state = ext ? 1 : 0; }
            else { int ext = t & 8; i--; b++; state = ext ? 2 : 0; }
        }
    }
}

// ------------------------------------------------------------------ K7: stringDiff over the alleles of one locus
__global__ __launch_bounds__(256) void k_hamming(const u8* __restrict__ ascii, const u64* __restrict__ aoff, u32 a_begin,
                                                 u32 n_alleles, const u8* __restrict__ query, u32 qlen, u32* __restrict__ dist) {
    extern __shared__ u8 s_q[];
    for (u32 i = threadIdx.x; i < qlen; i += blockDim.x) s_q[i] = query[i];
    __syncthreads();
    u32 a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_alleles) return;
    u64 o = aoff[a_begin + a]; u32 m = (u32)(aoff[a_begin + a + 1] - o);
    u32 len = m < qlen ? m : qlen, c = 0;
    for (u32 i = 0; i < len; i++) c += (ascii[o + i] != s_q[i]);     // zip(s1, s2): the shorter length bounds it (Q10)
    dist[a] = c;
}

// ------------------------------------------------------------------ stats export / import (multi-GPU all-reduce)
__global__ void k_export(EngineDev E, long long* d_sum, long long* d_min) {
    u64 nA = E.n_alleles, nL = E.n_loci;
    u64 total = 2 * nA + nL + MLST_CNT_N;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total + nL; i += (u64)gridDim.x * blockDim.x) {
        if (i < nA) d_sum[i] = E.sum_score[i];
        else if (i < 2 * nA) d_sum[i] = (long long)E.n_hits[i - nA];
        else if (i < 2 * nA + nL) d_sum[i] = (long long)E.locus_len[i - 2 * nA];
        else if (i < total) d_sum[i] = (long long)E.ctr->cnt[i - 2 * nA - nL];
        else { u64 l = i - total; u64 f = E.locus_first[l]; d_min[l] = f > 0x7FFFFFFFFFFFFFFFull ? 0x7FFFFFFFFFFFFFFFll : (long long)f; }
    }
}
__global__ void k_import(EngineDev E, const long long* d_sum, const long long* d_min) {
    u64 nA = E.n_alleles, nL = E.n_loci;
    u64 total = 2 * nA + nL + MLST_CNT_N;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total + nL; i += (u64)gridDim.x * blockDim.x) {
        if (i < nA) E.sum_score[i] = d_sum[i];
        else if (i < 2 * nA) E.n_hits[i - nA] = (u32)d_sum[i];
        else if (i < 2 * nA + nL) E.locus_len[i - 2 * nA] = (u64)d_sum[i];
        else if (i < total) E.ctr->cnt[i - 2 * nA - nL] = (u64)d_sum[i];
        else { u64 l = i - total; long long f = d_min[l]; E.locus_first[l] = f == 0x7FFFFFFFFFFFFFFFll ? 0xFFFFFFFFFFFFFFFFull : (u64)f; }
    }
}
__global__ void k_fill_u64(u64* p, u64 n, u64 v) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) p[i] = v;
}

// ================================================================== host side
struct EvPair { hipEvent_t a, b; int which; };

struct mlst_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    mlst_params prm;
    KParams kp;
    std::string err;
    // reference (host copies needed later)
    std::vector<LocusDev> loci;
    std::vector<u64> aoff;
    u32 n_alleles = 0, n_loci = 0;
    std::vector<u32> allele_locus;
    // device memory
    u32* d_arena = nullptr; u32* d_nmask = nullptr; u16* d_allele_len = nullptr; u32* d_allele_locus = nullptr;
    LocusDev* d_loci = nullptr; uint4* d_sieve = nullptr; u64* d_keys = nullptr; u32* d_vals = nullptr; u32* d_posts = nullptr;
    int* d_floor = nullptr; u8* d_pen = nullptr; u8* d_ascii = nullptr; u64* d_aoff = nullptr;
    u64 bytes_arena = 0, bytes_sieve = 0, bytes_table = 0;
    EngineDev E;
    bool have_ref = false, have_state = false;
    // batch scratch
    u32* d_cand = nullptr; u64 cap_cand = 0;
    u8* d_in_bases = nullptr; u8* d_in_quals = nullptr; u64* d_in_off = nullptr; u64 cap_in_bytes = 0, cap_in_reads = 0;
    u32* d_packed = nullptr; u8* d_qrows = nullptr; u16* d_lens = nullptr; u64 cap_packed_words = 0, cap_qrow_bytes = 0, cap_lens = 0;
    u64 reads_seen = 0;
    // pileup scratch
    int* d_locus_chosen = nullptr; u64* d_locus_colbase = nullptr; u64* d_pl_list = nullptr; u8* d_tb = nullptr;
    u32* d_counts = nullptr; u64 cap_counts = 0;
    u32* d_dist = nullptr; u8* d_query = nullptr; u64 cap_dist = 0, cap_query = 0;
    // profiling
    bool profiling = false;
    std::vector<EvPair> events;
    double k_ms[8] = {0}; u64 k_n[8] = {0};
};

static std::string g_create_err;

static int fail(mlst_handle* h, int code, const char* fmt, ...) {
    char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (h) h->err = buf; else g_create_err = buf;
    return code;
}
#define HIPCHK(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(h, MLST_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); } while (0)

template <typename T> static hipError_t dmalloc(T** p, u64 n) { return hipMalloc((void**)p, (n ? n : 1) * sizeof(T)); }

struct Prof {
    mlst_handle* h; int which; hipEvent_t a = nullptr, b = nullptr;
    Prof(mlst_handle* h_, int w) : h(h_), which(w) {
        if (h->profiling) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, h->stream); }
    }
    ~Prof() { if (h->profiling) { hipEventRecord(b, h->stream); h->events.push_back({a, b, which}); } }
};
static void drain_events(mlst_handle* h) {
    if (h->events.empty()) return;
    hipStreamSynchronize(h->stream);
    for (auto& e : h->events) { float ms = 0; hipEventElapsedTime(&ms, e.a, e.b); h->k_ms[e.which] += ms; h->k_n[e.which]++; hipEventDestroy(e.a); hipEventDestroy(e.b); }
    h->events.clear();
}

extern "C" void mlst_default_params(mlst_params* p) {
    memset(p, 0, sizeof *p);
    p->minscore = MLST_DEF_MINSCORE; p->max_xm = MLST_DEF_MAX_XM; p->min_read_len = MLST_DEF_MIN_READ_LEN;
    p->minqual = MLST_DEF_MINQUAL; p->mincov = MLST_DEF_MINCOV; p->match_bonus = MLST_DEF_MATCH_BONUS;
    p->mm_max = MLST_DEF_MM_MAX; p->mm_min = MLST_DEF_MM_MIN; p->n_penalty = MLST_DEF_N_PENALTY;
    p->gap_open = MLST_DEF_GAP_OPEN; p->gap_ext = MLST_DEF_GAP_EXT; p->gbar = MLST_DEF_GBAR; p->band_w = MLST_DEF_BAND_W;
    p->gap_trigger_mm = MLST_DEF_GAP_TRIGGER_MM; p->xm_field_quirk = MLST_DEF_XM_FIELD_QUIRK;
    p->minscore_const = MLST_DEF_MINSCORE_CONST; p->minscore_coef = MLST_DEF_MINSCORE_COEF;
}

extern "C" const char* mlst_last_error(const mlst_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

extern "C" int mlst_create(int device, const mlst_params* p, mlst_handle** out) {
    if (!out) return fail(nullptr, MLST_E_INVALID, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, MLST_E_NOGPU, "no HIP device visible: this library has no CPU path");
    if (device < 0 || device >= ndev) return fail(nullptr, MLST_E_INVALID, "device %d out of range (%d devices)", device, ndev);
    mlst_params prm; if (p) prm = *p; else mlst_default_params(&prm);
    if (prm.band_w < 1 || prm.band_w > MAX_W) return fail(nullptr, MLST_E_INVALID, "band_w must be in 1..%d", MAX_W);
    if (prm.max_xm > 254 || prm.minscore > 1000) return fail(nullptr, MLST_E_INVALID, "max_xm/minscore out of range");
    if (!prm.max_retained_reads) prm.max_retained_reads = 4ull << 20;
    if (!prm.max_items) prm.max_items = 8ull << 20;
    if (!prm.max_pair_results) prm.max_pair_results = 256ull << 20;
    mlst_handle* h = new mlst_handle();
    h->device = device; h->prm = prm;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess) { delete h; return fail(nullptr, MLST_E_HIP, "cannot initialise device %d", device); }
    KParams& k = h->kp;
    k.minscore = prm.minscore; k.max_xm = prm.max_xm; k.min_read_len = prm.min_read_len; k.minqual = prm.minqual;
    k.match_bonus = prm.match_bonus; k.n_penalty = prm.n_penalty;
    k.open_p = ((prm.gap_open + prm.gap_ext) << MLST_P_SHIFT) + (1 << 8); k.ext_p = prm.gap_ext << MLST_P_SHIFT;
    k.gbar = prm.gbar; k.band_w = prm.band_w; k.trig = prm.gap_trigger_mm; k.quirk = prm.xm_field_quirk;
    memset(&h->E, 0, sizeof h->E);
    *out = h;
    return MLST_OK;
}

static void free_ref(mlst_handle* h) {
    hipFree(h->d_arena); hipFree(h->d_nmask); hipFree(h->d_allele_len); hipFree(h->d_allele_locus); hipFree(h->d_loci);
    hipFree(h->d_sieve); hipFree(h->d_keys); hipFree(h->d_vals); hipFree(h->d_posts); hipFree(h->d_floor); hipFree(h->d_pen);
    hipFree(h->d_ascii); hipFree(h->d_aoff);
    h->d_arena = h->d_nmask = nullptr; h->d_allele_len = nullptr; h->d_allele_locus = nullptr; h->d_loci = nullptr; h->d_sieve = nullptr;
    h->d_keys = nullptr; h->d_vals = h->d_posts = nullptr; h->d_floor = nullptr; h->d_pen = nullptr; h->d_ascii = nullptr; h->d_aoff = nullptr;
    h->have_ref = false;
}
static void free_state(mlst_handle* h) {
    EngineDev& E = h->E;
    hipFree(E.sum_score); hipFree(E.n_hits); hipFree(E.locus_len); hipFree(E.locus_first); hipFree(E.ctr);
    hipFree(E.ret_bases); hipFree(E.ret_quals); hipFree(E.ret_len); hipFree(E.ret_ridx); hipFree(E.ret_nrec);
    hipFree(E.items); hipFree(E.res); hipFree(E.dp_list);
    hipFree(h->d_locus_chosen); hipFree(h->d_locus_colbase); hipFree(h->d_pl_list); hipFree(h->d_tb);
    E.sum_score = nullptr; E.n_hits = nullptr; E.locus_len = E.locus_first = nullptr; E.ctr = nullptr; E.ret_bases = nullptr; E.ret_quals = nullptr;
    E.ret_len = nullptr; E.ret_ridx = nullptr; E.ret_nrec = nullptr; E.items = nullptr; E.res = nullptr; E.dp_list = nullptr;
    h->d_locus_chosen = nullptr; h->d_locus_colbase = nullptr; h->d_pl_list = nullptr; h->d_tb = nullptr;
    h->have_state = false;
}

extern "C" void mlst_destroy(mlst_handle* h) {
    if (!h) return;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    for (auto& e : h->events) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    free_ref(h); free_state(h);
    hipFree(h->d_cand); hipFree(h->d_in_bases); hipFree(h->d_in_quals); hipFree(h->d_in_off);
    hipFree(h->d_packed); hipFree(h->d_qrows); hipFree(h->d_lens); hipFree(h->d_counts); hipFree(h->d_dist); hipFree(h->d_query);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

static inline int base_code(u8 c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

static int reset_sample_state(mlst_handle* h) {
    EngineDev& E = h->E;
    HIPCHK(h, hipMemsetAsync(E.sum_score, 0, sizeof(long long) * (h->n_alleles ? h->n_alleles : 1), h->stream));
    HIPCHK(h, hipMemsetAsync(E.n_hits, 0, sizeof(u32) * (h->n_alleles ? h->n_alleles : 1), h->stream));
    HIPCHK(h, hipMemsetAsync(E.locus_len, 0, sizeof(u64) * (h->n_loci ? h->n_loci : 1), h->stream));
    HIPCHK(h, hipMemsetAsync(E.locus_first, 0xFF, sizeof(u64) * (h->n_loci ? h->n_loci : 1), h->stream));
    HIPCHK(h, hipMemsetAsync(E.ctr, 0, sizeof(Counters), h->stream));
    h->reads_seen = 0;
    return MLST_OK;
}

// Build the device-resident reference: transposed 2-bit allele arena, N masks, seed sieve and exact seed table.
extern "C" int mlst_load_reference(mlst_handle* h, const uint8_t* ascii, const uint64_t* off, const uint32_t* locus_id,
                                   const uint32_t* species_id, const int32_t* allele_no, uint32_t n_alleles) {
    if (!h || !off || !locus_id) return fail(h, MLST_E_INVALID, "NULL argument");
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    free_ref(h); free_state(h);
    h->n_alleles = n_alleles;
    // ---- locus table
    u32 n_loci = 0;
    for (u32 a = 0; a < n_alleles; a++) {
        if (locus_id[a] > MLST_MAX_LOCI) return fail(h, MLST_E_LIMIT, "locus id %u exceeds %d", locus_id[a], MLST_MAX_LOCI);
        n_loci = std::max(n_loci, locus_id[a] + 1);
        if (off[a + 1] - off[a] > MLST_MAX_ALLELE_LEN) return fail(h, MLST_E_LIMIT, "allele %u longer than %d", a, MLST_MAX_ALLELE_LEN);
    }
    h->n_loci = n_loci;
    std::vector<LocusDev> loci(n_loci); for (auto& L : loci) memset(&L, 0, sizeof L);
    std::vector<char> seen(n_loci, 0);
    for (u32 a = 0; a < n_alleles; a++) {
        LocusDev& L = loci[locus_id[a]];
        if (!seen[locus_id[a]]) { seen[locus_id[a]] = 1; L.a_begin = a; L.species = species_id ? species_id[a] : 0; }
        else if (L.a_begin + L.n_alleles != a) return fail(h, MLST_E_INVALID, "alleles of locus %u are not contiguous", locus_id[a]);
        L.n_alleles++; L.max_len = std::max(L.max_len, (u32)(off[a + 1] - off[a]));
    }
    u64 arena_words = 0, nmask_words = 0;
    std::vector<u16> alen(n_alleles); h->allele_locus.assign(locus_id, locus_id + n_alleles);
    for (u32 a = 0; a < n_alleles; a++) {
        alen[a] = (u16)(off[a + 1] - off[a]);
        for (u64 i = off[a]; i < off[a + 1]; i++) if (base_code(ascii[i]) > 3) { loci[locus_id[a]].has_n = 1; break; }
    }
    for (auto& L : loci) {
        if (L.n_alleles >= (1u << 20)) return fail(h, MLST_E_LIMIT, "locus with %u alleles exceeds 2^20", L.n_alleles);
        L.n_pad = (L.n_alleles + 63) & ~63u; L.words = (L.max_len + 15) / 16 + 2; L.nwords = (L.max_len + 31) / 32 + 1;
        L.arena_off = arena_words; arena_words += (u64)L.words * L.n_pad;
        if (L.has_n) { L.nmask_off = nmask_words; nmask_words += (u64)L.nwords * L.n_pad; }
    }
    std::vector<u32> arena(arena_words ? arena_words : 1, 0), nmask(nmask_words ? nmask_words : 1, 0);
    // ---- arena fill + seed pairs
    struct KP { u64 key; u32 post; };
    std::vector<KP> kp;
    { u64 tot = 0; for (u32 a = 0; a < n_alleles; a++) if (alen[a] >= MLST_SEED_LEN) tot += 2ull * (alen[a] - MLST_SEED_LEN + 1); kp.reserve(tot); }
    std::vector<u8> code;
    for (u32 a = 0; a < n_alleles; a++) {
        const LocusDev& L = loci[locus_id[a]]; u32 al = a - L.a_begin; u32 len = alen[a];
        code.resize(len);
        for (u32 i = 0; i < len; i++) {
            int c = base_code(ascii[off[a] + i]); code[i] = (u8)c;
            if (c < 4) arena[L.arena_off + (u64)(i >> 4) * L.n_pad + al] |= (u32)c << (2 * (i & 15));
            else nmask[L.nmask_off + (u64)(i >> 5) * L.n_pad + al] |= 1u << (i & 31);
        }
        if (len < MLST_SEED_LEN) continue;
        // rolling forward key and reverse-complement key of the window [p, p+20)
        int bad = 0; u64 fk = 0, rk = 0; const u64 mask40 = (1ull << 40) - 1;
        for (u32 i = 0; i < len; i++) {
            int c = code[i];
            if (c > 3) { bad = MLST_SEED_LEN; c = 0; } else if (bad) bad--;
            fk = (fk >> 2) | ((u64)c << 38);                 // base t of the window at bits 2t
            rk = ((rk << 2) | (u64)(3 - c)) & mask40;        // rc base t = 3 - code[p+19-t]
            if (i + 1 >= MLST_SEED_LEN && !bad) {
                u32 p = i + 1 - MLST_SEED_LEN;
                kp.push_back({fk, (locus_id[a] << 13) | p});
                kp.push_back({rk, (locus_id[a] << 13) | (1u << 12) | p});
            }
        }
    }
    std::sort(kp.begin(), kp.end(), [](const KP& x, const KP& y) { return x.key != y.key ? x.key < y.key : x.post < y.post; });
    kp.erase(std::unique(kp.begin(), kp.end(), [](const KP& x, const KP& y) { return x.key == y.key && x.post == y.post; }), kp.end());
    // group, drop repetitive seeds
    std::vector<u64> ukeys; std::vector<u32> uval; std::vector<u32> posts;
    for (size_t i = 0; i < kp.size();) {
        size_t j = i; while (j < kp.size() && kp[j].key == kp[i].key) j++;
        size_t cnt = j - i;
        if (cnt <= MLST_MAX_POSTINGS) {
            ukeys.push_back(kp[i].key);
            if (cnt == 1) uval.push_back(0x80000000u | kp[i].post);
            else {
                if (posts.size() >= (1ull << 26)) return fail(h, MLST_E_LIMIT, "posting list exceeds 2^26 entries");
                uval.push_back(((u32)posts.size() << 5) | (u32)cnt);
                for (size_t t = i; t < j; t++) posts.push_back(kp[t].post);
            }
        }
        i = j;
    }
    std::vector<KP>().swap(kp);
    const u64 nk = ukeys.size();
    // ---- exact table: open addressing, load <= 0.5
    u64 tcap = 1024; while (tcap < 2 * nk) tcap <<= 1;
    if (tcap > (1ull << 32)) return fail(h, MLST_E_LIMIT, "seed table too large");
    std::vector<u64> tkeys(tcap, KEY_EMPTY); std::vector<u32> tvals(tcap, 0);
    u32 tmask = (u32)(tcap - 1);
    for (u64 i = 0; i < nk; i++) {
        u32 lo = (u32)ukeys[i], hi = (u32)(ukeys[i] >> 32);
        u32 hh = table_hash(lo, hi) & tmask;
        while (tkeys[hh] != KEY_EMPTY) hh = (hh + 1) & tmask;
        tkeys[hh] = ukeys[i]; tvals[hh] = uval[i];
    }
    // ---- sieve: 16-byte buckets of eight 16-bit fingerprints, mean fill <= 4
    u64 nb = 256; while (nb * 4 < nk) nb <<= 1;
    if (nb > (1ull << 31)) return fail(h, MLST_E_LIMIT, "sieve too large");
    std::vector<u16> sv(nb * 8, 0); u32 smask = (u32)(nb - 1);
    for (u64 i = 0; i < nk; i++) {
        u32 lo = (u32)ukeys[i], hi = (u32)(ukeys[i] >> 32);
        u32 fp = sieve_fp(lo, hi); u32 b = sieve_bucket_hash(lo, hi) & smask;
        for (u64 step = 0; step < nb; step++) {
            u16* B = &sv[(u64)b * 8]; int k; bool done = false;
            for (k = 0; k < 8; k++) { if (B[k] == fp) { done = true; break; } if (B[k] == 0) { B[k] = (u16)fp; done = true; break; } }
            if (done) break;
            b = (b + 1) & smask;
        }
    }
    // ---- tables derived from the parameters
    std::vector<int> floor_tab(MLST_MAX_READ_LEN + 1);
    for (int n = 0; n <= MLST_MAX_READ_LEN; n++) {
        double f = h->prm.minscore_const + h->prm.minscore_coef * log((double)(n > 0 ? n : 1));
        long v = (long)f; if (v < 0) v = 0; floor_tab[n] = (int)v;
    }
    std::vector<u8> pen_tab(256);
    for (int q = 0; q < 256; q++) { int qq = q > 40 ? 40 : q; pen_tab[q] = (u8)(h->prm.mm_min + ((h->prm.mm_max - h->prm.mm_min) * qq) / 40); }
    // ---- upload
    HIPCHK(h, dmalloc(&h->d_arena, arena.size())); HIPCHK(h, hipMemcpy(h->d_arena, arena.data(), arena.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_nmask, nmask.size())); HIPCHK(h, hipMemcpy(h->d_nmask, nmask.data(), nmask.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_allele_len, (u64)n_alleles)); HIPCHK(h, hipMemcpy(h->d_allele_len, alen.data(), (u64)n_alleles * 2, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_allele_locus, (u64)n_alleles)); HIPCHK(h, hipMemcpy(h->d_allele_locus, locus_id, (u64)n_alleles * 4, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_loci, (u64)n_loci)); HIPCHK(h, hipMemcpy(h->d_loci, loci.data(), (u64)n_loci * sizeof(LocusDev), hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_sieve, nb)); HIPCHK(h, hipMemcpy(h->d_sieve, sv.data(), nb * 16, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_keys, tcap)); HIPCHK(h, hipMemcpy(h->d_keys, tkeys.data(), tcap * 8, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_vals, tcap)); HIPCHK(h, hipMemcpy(h->d_vals, tvals.data(), tcap * 4, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_posts, (u64)posts.size())); if (!posts.empty()) HIPCHK(h, hipMemcpy(h->d_posts, posts.data(), posts.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_floor, (u64)floor_tab.size())); HIPCHK(h, hipMemcpy(h->d_floor, floor_tab.data(), floor_tab.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_pen, (u64)256)); HIPCHK(h, hipMemcpy(h->d_pen, pen_tab.data(), 256, hipMemcpyHostToDevice));
    u64 abytes = off[n_alleles];
    HIPCHK(h, dmalloc(&h->d_ascii, abytes)); if (abytes) HIPCHK(h, hipMemcpy(h->d_ascii, ascii, abytes, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_aoff, (u64)n_alleles + 1)); HIPCHK(h, hipMemcpy(h->d_aoff, off, ((u64)n_alleles + 1) * 8, hipMemcpyHostToDevice));
    h->bytes_arena = arena.size() * 4 + nmask.size() * 4; h->bytes_sieve = nb * 16; h->bytes_table = tcap * 12 + posts.size() * 4;
    h->loci = loci; h->aoff.assign(off, off + n_alleles + 1);
    // ---- sample state
    EngineDev& E = h->E;
    E.arena = h->d_arena; E.nmask = h->d_nmask; E.allele_len = h->d_allele_len; E.allele_locus = h->d_allele_locus; E.loci = h->d_loci;
    E.sieve = h->d_sieve; E.sieve_mask = smask; E.keys = h->d_keys; E.vals = h->d_vals; E.posts = h->d_posts; E.table_mask = tmask;
    E.floor_tab = h->d_floor; E.pen_tab = h->d_pen; E.n_alleles = n_alleles; E.n_loci = n_loci;
    E.cap_ret = h->prm.max_retained_reads; E.cap_items = h->prm.max_items; E.cap_res = h->prm.max_pair_results; E.cap_dp = h->prm.max_items * 4;
    HIPCHK(h, dmalloc(&E.sum_score, (u64)n_alleles)); HIPCHK(h, dmalloc(&E.n_hits, (u64)n_alleles));
    HIPCHK(h, dmalloc(&E.locus_len, (u64)n_loci)); HIPCHK(h, dmalloc(&E.locus_first, (u64)n_loci)); HIPCHK(h, dmalloc(&E.ctr, (u64)1));
    HIPCHK(h, dmalloc(&E.ret_bases, E.cap_ret * RW)); HIPCHK(h, dmalloc(&E.ret_quals, E.cap_ret * RQ));
    HIPCHK(h, dmalloc(&E.ret_len, E.cap_ret)); HIPCHK(h, dmalloc(&E.ret_ridx, E.cap_ret)); HIPCHK(h, dmalloc(&E.ret_nrec, E.cap_ret));
    HIPCHK(h, dmalloc(&E.items, E.cap_items)); HIPCHK(h, dmalloc(&E.res, E.cap_res)); HIPCHK(h, dmalloc(&E.dp_list, E.cap_dp));
    HIPCHK(h, dmalloc(&h->d_locus_chosen, (u64)n_loci)); HIPCHK(h, dmalloc(&h->d_locus_colbase, (u64)n_loci));
    HIPCHK(h, dmalloc(&h->d_pl_list, E.cap_items));
    HIPCHK(h, dmalloc(&h->d_tb, (u64)64 * 64 * MLST_MAX_READ_LEN * (2 * MAX_W + 1)));
    h->have_ref = h->have_state = true;
    int rc = reset_sample_state(h); if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MLST_OK;
}

extern "C" int mlst_reset_sample(mlst_handle* h) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    hipSetDevice(h->device);
    drain_events(h);
    return reset_sample_state(h);
}

static int grid_for(u64 n_units, int per_block, int cap = 2048) {
    u64 g = (n_units + per_block - 1) / per_block; if (g < 1) g = 1; if (g > (u64)cap) g = cap; return (int)g;
}

extern "C" int mlst_pack_reads_device(mlst_handle* h, const uint8_t* d_bases, const uint8_t* d_quals, const uint64_t* d_off,
                                      uint64_t n_reads, uint32_t* d_packed, uint8_t* d_qrows, uint16_t* d_lens,
                                      uint32_t wpr, uint32_t qstride) {
    if (!h) return MLST_E_INVALID;
    hipSetDevice(h->device);
    if (wpr == 0 || wpr > RW || (wpr & 1)) return fail(h, MLST_E_INVALID, "words_per_read must be even and in 2..%d", RW);
    if (qstride < 1 || qstride > RQ) return fail(h, MLST_E_INVALID, "qual_stride must be in 1..%d", RQ);
    if (n_reads == 0) return MLST_OK;
    Prof pf(h, 6);
    hipLaunchKernelGGL(k_pack_lens, dim3(grid_for(n_reads, 256)), dim3(256), 0, h->stream, (const u64*)d_off, (u64)n_reads, d_lens);
    hipLaunchKernelGGL(k_pack, dim3(grid_for(n_reads * wpr, 256, 8192)), dim3(256), 0, h->stream, d_bases, d_quals, (const u64*)d_off, (u64)n_reads,
                       d_packed, d_qrows, d_lens, wpr, qstride);
    HIPCHK(h, hipGetLastError());
    return MLST_OK;
}

extern "C" int mlst_submit_packed_device(mlst_handle* h, const uint32_t* d_packed, const uint8_t* d_qrows, const uint16_t* d_lens,
                                         uint64_t n_reads, uint32_t wpr, uint32_t qstride, int paired) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    hipSetDevice(h->device);
    if (wpr == 0 || wpr > RW || (wpr & 1)) return fail(h, MLST_E_INVALID, "words_per_read must be even and in 2..%d", RW);
    if (qstride < 1 || qstride > RQ) return fail(h, MLST_E_INVALID, "qual_stride must be in 1..%d", RQ);
    if (n_reads >= (1ull << 32)) return fail(h, MLST_E_LIMIT, "a batch holds at most 2^32-1 reads");
    if (((uintptr_t)d_packed & 15) != 0) return fail(h, MLST_E_INVALID, "packed rows must be 16-byte aligned");
    if (n_reads == 0) return MLST_OK;
    (void)paired;   // mates are typed independently; they share a QNAME only for the coverage figure
    if (h->cap_cand < n_reads) { hipStreamSynchronize(h->stream); hipFree(h->d_cand); h->d_cand = nullptr; HIPCHK(h, dmalloc(&h->d_cand, n_reads)); h->cap_cand = n_reads; }
    EngineDev& E = h->E;
    { Prof pf(h, 0);
      u64 nblk = (n_reads + 255) / 256;
      hipLaunchKernelGGL(k_sieve, dim3(grid_for(nblk, 1, 256 * 8)), dim3(256), 256 * wpr * 4, h->stream, d_packed, d_lens, n_reads, wpr,
                         E.sieve, E.sieve_mask, h->d_cand, E.ctr); }
    { Prof pf(h, 1);
      hipLaunchKernelGGL(k_seed, dim3(512), dim3(256), 0, h->stream, E, d_packed, d_qrows, d_lens, wpr, qstride, h->reads_seen, h->d_cand); }
    { Prof pf(h, 2); hipLaunchKernelGGL(k_extend, dim3(2048), dim3(256), 0, h->stream, E, h->kp); }
    { Prof pf(h, 3); hipLaunchKernelGGL(k_banded, dim3(1024), dim3(64), 0, h->stream, E, h->kp); }
    { Prof pf(h, 4); hipLaunchKernelGGL(k_accumulate, dim3(2048), dim3(256), 0, h->stream, E, h->kp); }
    hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, h->stream, E.ctr, n_reads);
    HIPCHK(h, hipGetLastError());
    h->reads_seen += n_reads;
    return MLST_OK;
}

static int ensure_pack_buffers(mlst_handle* h, u64 n_reads, u32 wpr, u32 qstride) {
    if (h->cap_packed_words < n_reads * wpr + 4 || h->cap_qrow_bytes < n_reads * qstride || h->cap_lens < n_reads + 2) {
        hipStreamSynchronize(h->stream);
        hipFree(h->d_packed); hipFree(h->d_qrows); hipFree(h->d_lens); h->d_packed = nullptr; h->d_qrows = nullptr; h->d_lens = nullptr;
        HIPCHK(h, dmalloc(&h->d_packed, n_reads * wpr + 4)); HIPCHK(h, dmalloc(&h->d_qrows, n_reads * qstride)); HIPCHK(h, dmalloc(&h->d_lens, n_reads + 2));
        h->cap_packed_words = n_reads * wpr + 4; h->cap_qrow_bytes = n_reads * qstride; h->cap_lens = n_reads + 2;
    }
    return MLST_OK;
}

extern "C" int mlst_submit_reads_device(mlst_handle* h, const uint8_t* d_bases, const uint8_t* d_quals, const uint64_t* d_off,
                                        uint64_t n_reads, uint32_t max_len, int paired) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    hipSetDevice(h->device);
    if (n_reads == 0) return MLST_OK;
    if (max_len > MLST_MAX_READ_LEN) return fail(h, MLST_E_LIMIT, "read longer than %d bases", MLST_MAX_READ_LEN);
    u32 wpr = (max_len + 15) / 16; if (wpr < 2) wpr = 2; wpr = (wpr + 1) & ~1u;
    u32 qstride = (max_len + 7) & ~7u; if (qstride < 8) qstride = 8;
    int rc = ensure_pack_buffers(h, n_reads, wpr, qstride); if (rc) return rc;
    rc = mlst_pack_reads_device(h, d_bases, d_quals, d_off, n_reads, h->d_packed, h->d_qrows, h->d_lens, wpr, qstride); if (rc) return rc;
    return mlst_submit_packed_device(h, h->d_packed, h->d_qrows, h->d_lens, n_reads, wpr, qstride, paired);
}

extern "C" int mlst_submit_reads(mlst_handle* h, const uint8_t* bases, const uint8_t* quals, const uint64_t* off,
                                 uint64_t n_reads, int paired) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    if (n_reads == 0) return MLST_OK;
    if (!bases || !quals || !off) return fail(h, MLST_E_INVALID, "NULL argument");
    hipSetDevice(h->device);
    u64 nbytes = off[n_reads] - off[0]; u32 max_len = 0;
    for (u64 r = 0; r < n_reads; r++) { u64 l = off[r + 1] - off[r]; if (l > MLST_MAX_READ_LEN) return fail(h, MLST_E_LIMIT, "read %llu longer than %d bases", (unsigned long long)r, MLST_MAX_READ_LEN); max_len = std::max(max_len, (u32)l); }
    if (h->cap_in_bytes < nbytes || h->cap_in_reads < n_reads + 1) {
        hipStreamSynchronize(h->stream);
        hipFree(h->d_in_bases); hipFree(h->d_in_quals); hipFree(h->d_in_off); h->d_in_bases = h->d_in_quals = nullptr; h->d_in_off = nullptr;
        HIPCHK(h, dmalloc(&h->d_in_bases, nbytes)); HIPCHK(h, dmalloc(&h->d_in_quals, nbytes)); HIPCHK(h, dmalloc(&h->d_in_off, n_reads + 1));
        h->cap_in_bytes = nbytes; h->cap_in_reads = n_reads + 1;
    }
    hipStreamSynchronize(h->stream);     // the previous batch may still read the staging buffers
    std::vector<u64> rel(n_reads + 1); for (u64 r = 0; r <= n_reads; r++) rel[r] = off[r] - off[0];
    HIPCHK(h, hipMemcpy(h->d_in_bases, bases + off[0], nbytes, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_in_quals, quals + off[0], nbytes, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_in_off, rel.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice));
    return mlst_submit_reads_device(h, h->d_in_bases, h->d_in_quals, (const uint64_t*)h->d_in_off, n_reads, max_len, paired);
}

static int check_overflow(mlst_handle* h) {
    Counters c; HIPCHK(h, hipMemcpyAsync(&c, h->E.ctr, sizeof c, hipMemcpyDeviceToHost, h->stream)); HIPCHK(h, hipStreamSynchronize(h->stream));
    if (c.err) return fail(h, MLST_E_CAPACITY, "capacity exceeded (flags 0x%llx: 1=retained reads %llu/%llu, 2=items %llu/%llu, 4=pair results %llu/%llu, 8=banded-SW list); raise mlst_params.max_*",
                           (unsigned long long)c.err, (unsigned long long)c.n_ret, (unsigned long long)h->E.cap_ret, (unsigned long long)c.n_items, (unsigned long long)h->E.cap_items,
                           (unsigned long long)c.n_res, (unsigned long long)h->E.cap_res);
    return MLST_OK;
}

extern "C" int mlst_get_allele_stats(mlst_handle* h, int64_t* sum_score, uint32_t* n_hits, uint64_t* locus_len,
                                     uint64_t* locus_first, uint64_t* counters) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    hipSetDevice(h->device);
    int rc = check_overflow(h); if (rc) return rc;
    EngineDev& E = h->E;
    if (sum_score) HIPCHK(h, hipMemcpyAsync(sum_score, E.sum_score, (u64)h->n_alleles * 8, hipMemcpyDeviceToHost, h->stream));
    if (n_hits) HIPCHK(h, hipMemcpyAsync(n_hits, E.n_hits, (u64)h->n_alleles * 4, hipMemcpyDeviceToHost, h->stream));
    if (locus_len) HIPCHK(h, hipMemcpyAsync(locus_len, E.locus_len, (u64)h->n_loci * 8, hipMemcpyDeviceToHost, h->stream));
    if (locus_first) HIPCHK(h, hipMemcpyAsync(locus_first, E.locus_first, (u64)h->n_loci * 8, hipMemcpyDeviceToHost, h->stream));
    Counters c; HIPCHK(h, hipMemcpyAsync(&c, E.ctr, sizeof c, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (counters) { for (int i = 0; i < MLST_CNT_N; i++) counters[i] = c.cnt[i]; counters[MLST_CNT_RETAINED] = c.n_ret; counters[MLST_CNT_ITEMS] = c.n_items; }
    return MLST_OK;
}

extern "C" int mlst_stats_flat_sizes(mlst_handle* h, uint64_t* n_sum, uint64_t* n_min) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    if (n_sum) *n_sum = 2ull * h->n_alleles + h->n_loci + MLST_CNT_N;
    if (n_min) *n_min = h->n_loci;
    return MLST_OK;
}
extern "C" int mlst_export_stats_device(mlst_handle* h, int64_t* d_sum, int64_t* d_min) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    hipSetDevice(h->device);
    int rc = check_overflow(h); if (rc) return rc;
    hipLaunchKernelGGL(k_export, dim3(256), dim3(256), 0, h->stream, h->E, (long long*)d_sum, (long long*)d_min);
    HIPCHK(h, hipGetLastError()); HIPCHK(h, hipStreamSynchronize(h->stream));
    return MLST_OK;
}
extern "C" int mlst_import_stats_device(mlst_handle* h, const int64_t* d_sum, const int64_t* d_min) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    hipSetDevice(h->device);
    hipLaunchKernelGGL(k_import, dim3(256), dim3(256), 0, h->stream, h->E, (const long long*)d_sum, (const long long*)d_min);
    HIPCHK(h, hipGetLastError()); HIPCHK(h, hipStreamSynchronize(h->stream));
    return MLST_OK;
}

extern "C" int mlst_pileup_device(mlst_handle* h, const uint32_t* chosen, uint32_t n, uint32_t* d_counts, uint64_t* n_cols_out) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    hipSetDevice(h->device);
    std::vector<int> lc(h->n_loci, -1); std::vector<u64> cb(h->n_loci, 0); u64 ncols = 0;
    for (u32 k = 0; k < n; k++) {
        u32 a = chosen[k]; if (a >= h->n_alleles) return fail(h, MLST_E_INVALID, "chosen allele %u out of range", a);
        u32 L = h->allele_locus[a]; if (lc[L] >= 0) return fail(h, MLST_E_INVALID, "two chosen alleles for locus %u", L);
        lc[L] = (int)a; cb[L] = ncols; ncols += h->aoff[a + 1] - h->aoff[a];
    }
    if (n_cols_out) *n_cols_out = ncols;
    HIPCHK(h, hipMemcpyAsync(h->d_locus_chosen, lc.data(), (u64)h->n_loci * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_locus_colbase, cb.data(), (u64)h->n_loci * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(d_counts, 0, (ncols ? ncols : 1) * 16, h->stream));
    HIPCHK(h, hipMemsetAsync(&h->E.ctr->n_pl_dp, 0, 8, h->stream));
    { Prof pf(h, 5);
      hipLaunchKernelGGL(k_pileup, dim3(1024), dim3(64), 0, h->stream, h->E, h->kp, h->d_locus_chosen, h->d_locus_colbase, d_counts, h->d_pl_list);
      hipLaunchKernelGGL(k_pileup_dp, dim3(64), dim3(64), 0, h->stream, h->E, h->kp, h->d_locus_chosen, h->d_locus_colbase, d_counts, h->d_pl_list, h->d_tb); }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));   // lc / cb are stack-owned host buffers
    return MLST_OK;
}

extern "C" int mlst_pileup(mlst_handle* h, const uint32_t* chosen, uint32_t n, uint32_t* counts) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    hipSetDevice(h->device);
    u64 ncols = 0;
    for (u32 k = 0; k < n; k++) { if (chosen[k] >= h->n_alleles) return fail(h, MLST_E_INVALID, "chosen allele out of range"); ncols += h->aoff[chosen[k] + 1] - h->aoff[chosen[k]]; }
    if (h->cap_counts < ncols * 4 + 4) { hipStreamSynchronize(h->stream); hipFree(h->d_counts); h->d_counts = nullptr; HIPCHK(h, dmalloc(&h->d_counts, ncols * 4 + 4)); h->cap_counts = ncols * 4 + 4; }
    uint64_t nc2 = 0; int rc = mlst_pileup_device(h, chosen, n, h->d_counts, &nc2); if (rc) return rc;
    if (ncols) HIPCHK(h, hipMemcpy(counts, h->d_counts, ncols * 16, hipMemcpyDeviceToHost));
    return MLST_OK;
}

extern "C" int mlst_hamming_all(mlst_handle* h, uint32_t locus, const uint8_t* query, uint32_t len, uint32_t* dist) {
    if (!h || !h->have_ref) return fail(h, MLST_E_INVALID, "no reference loaded");
    if (locus >= h->n_loci) return fail(h, MLST_E_INVALID, "locus %u out of range", locus);
    hipSetDevice(h->device);
    const LocusDev& L = h->loci[locus];
    if (h->cap_dist < L.n_alleles) { hipFree(h->d_dist); h->d_dist = nullptr; HIPCHK(h, dmalloc(&h->d_dist, (u64)L.n_alleles)); h->cap_dist = L.n_alleles; }
    if (h->cap_query < (u64)len + 1) { hipFree(h->d_query); h->d_query = nullptr; HIPCHK(h, dmalloc(&h->d_query, (u64)len + 1)); h->cap_query = (u64)len + 1; }
    if (len > 60000) return fail(h, MLST_E_LIMIT, "query longer than 60000 bytes");
    if (len) HIPCHK(h, hipMemcpyAsync(h->d_query, query, len, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_hamming, dim3((L.n_alleles + 255) / 256), dim3(256), len ? len : 1, h->stream, h->d_ascii, h->d_aoff, L.a_begin,
                       L.n_alleles, h->d_query, len, h->d_dist);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(dist, h->d_dist, (u64)L.n_alleles * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MLST_OK;
}

extern "C" int mlst_hamming_le(mlst_handle* h, uint32_t locus, const uint8_t* query, uint32_t len, uint32_t z,
                               int32_t* first_allele_idx, uint32_t* n_within) {
    if (!h || !h->have_ref) return fail(h, MLST_E_INVALID, "no reference loaded");
    if (locus >= h->n_loci) return fail(h, MLST_E_INVALID, "locus %u out of range", locus);
    std::vector<u32> d(h->loci[locus].n_alleles);
    int rc = mlst_hamming_all(h, locus, query, len, d.data()); if (rc) return rc;
    int first = -1; u32 nw = 0;
    for (u32 a = 0; a < d.size(); a++) if (d[a] <= z) { if (first < 0) first = (int)(h->loci[locus].a_begin + a); nw++; }
    if (first_allele_idx) *first_allele_idx = first;
    if (n_within) *n_within = nw;
    return MLST_OK;
}

extern "C" int mlst_get_items(mlst_handle* h, mlst_item* out, uint64_t cap, uint64_t* n) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    hipSetDevice(h->device);
    Counters c; HIPCHK(h, hipMemcpy(&c, h->E.ctr, sizeof c, hipMemcpyDeviceToHost));
    u64 ni = std::min<u64>(c.n_items, h->E.cap_items);
    if (n) *n = ni;
    u64 k = std::min<u64>(ni, cap);
    if (!k || !out) return MLST_OK;
    std::vector<ItemDev> items(k); HIPCHK(h, hipMemcpy(items.data(), h->E.items, k * sizeof(ItemDev), hipMemcpyDeviceToHost));
    u64 nr = std::min<u64>(c.n_ret, h->E.cap_ret);
    std::vector<u64> ridx(nr ? nr : 1); if (nr) HIPCHK(h, hipMemcpy(ridx.data(), h->E.ret_ridx, nr * 8, hipMemcpyDeviceToHost));
    for (u64 i = 0; i < k; i++) {
        out[i].read_index = items[i].ret < nr ? ridx[items[i].ret] : ~0ull; out[i].locus = items[i].locus; out[i].diag = items[i].diag;
        out[i].strand = items[i].strand; out[i].votes = items[i].votes; out[i].reserved = 0;
    }
    return MLST_OK;
}

extern "C" int mlst_set_profiling(mlst_handle* h, int on) { if (!h) return MLST_E_INVALID; drain_events(h); h->profiling = on != 0; return MLST_OK; }
extern "C" int mlst_get_kernel_time(mlst_handle* h, int which, double* total_ms, uint64_t* launches) {
    if (!h || which < 0 || which >= 8) return MLST_E_INVALID;
    hipSetDevice(h->device); drain_events(h);
    if (total_ms) *total_ms = h->k_ms[which];
    if (launches) *launches = h->k_n[which];
    return MLST_OK;
}
extern "C" int mlst_reset_kernel_time(mlst_handle* h) { if (!h) return MLST_E_INVALID; drain_events(h); for (int i = 0; i < 8; i++) { h->k_ms[i] = 0; h->k_n[i] = 0; } return MLST_OK; }
extern "C" int mlst_get_index_bytes(mlst_handle* h, uint64_t out[4]) {
    if (!h || !out) return MLST_E_INVALID;
    out[0] = h->bytes_arena; out[1] = h->bytes_sieve; out[2] = h->bytes_table; out[3] = 0; return MLST_OK;
}
extern "C" int mlst_synchronize(mlst_handle* h) { if (!h) return MLST_E_INVALID; hipSetDevice(h->device); HIPCHK(h, hipStreamSynchronize(h->stream)); return MLST_OK; }
