// mlst_engine.hip -- MI355X (gfx950) MLST-typing engine: HIP kernels + the C-ABI of include/mlst.h.
//
// Path (reference file:line in include/mlst.h and DESIGN.md):
//   K0 k_pack        ASCII reads -> 2-bit rows + Phred rows                (FASTQ/SAM fields)
//   K1 k_sieve_q     streaming seed sieve over ALL reads                   (bowtie2 seeding)
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
#include <cstddef>
#include <functional>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <chrono>
#include <unistd.h>

#include "mlst.h"
#include "mlst_debug.h"
#include "inflate_dev.h"
#include "inflate_wave.h"
#include "inflate_lane.h"
#include "inflate_canon.h"
#include "mlst_policy.h"

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;
typedef unsigned char u8;
typedef long long i64;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));   // native vector: accepted by the nontemporal builtins

#define RW  (MLST_MAX_READ_LEN / 16)     // words of a retained read row (20)
#define RQ  MLST_MAX_READ_LEN            // bytes of a retained quality row
#define KEY_EMPTY 0xFFFFFFFFFFFFFFFFull
#define MAX_W 15                          // largest supported band half width (band <= 31 cells: one 64-bit base window)
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
// The sieve's three indices (bitmap bit, bucket, fingerprint) are multiplicative hashes of one pre-mixed word:
// four integer multiplies per seed (32-bit multiplies are quarter rate).  Top bits of a product are the well
// mixed ones.  The pre-mix folds 40 key bits into 32, so a key has 255 aliases (one per other value of its top
// byte) with the same bucket; the fingerprint therefore mixes the top byte in a second, different way.
__host__ __device__ inline u32 seed_premix(u32 lo, u32 hi) { return lo ^ (hi * 0x01010101u); }
__host__ __device__ inline u32 sieve_bucket_hash(u32 lo, u32 hi) { return seed_premix(lo, hi) * 0x9E3779B1u; }   // callers keep the TOP log2(buckets) bits: h >> sshift
__host__ __device__ inline u32 sieve_fp(u32 lo, u32 hi) {
    u32 t = seed_premix(lo, hi) ^ (hi << 11); t ^= t >> 15;   // differs between bucket aliases; the xor-shift breaks the linear relation to the bucket hash
    u32 fp = (t * 0x85EBCA6Bu) >> 16;
    return fp ? fp : 1u;
}
__host__ __device__ inline u32 table_hash(u32 lo, u32 hi) {
    u32 h = (lo + hi * 0x7FEB352Du) * 0x846CA68Bu;
    h ^= h >> 15; h *= 0x9E3779B1u; h ^= h >> 14;
    return h;
}

// Canonical seed: the smaller of a 20-mer and its reverse complement (40-bit keys, base t at bits 2t).  One table
// entry serves both strands, which halves the sieve and the seed table.  flag = 1 when the reverse complement is the
// canonical form.  A posting stores the flag of the ALLELE k-mer; strand = posting flag XOR read-seed flag.
__host__ __device__ inline u64 brev64_(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
    return (x >> 32) | (x << 32);
#endif
}
__host__ __device__ inline u64 revcomp40(u64 s) {
    u64 y = brev64_(s);                                                                  // group t -> pair 31-t, bits swapped
    y = ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);      // restore bit order inside each base
    return (~(y >> 24)) & 0xFFFFFFFFFFull;                                               // pair 31-t -> 19-t, complement
}
__host__ __device__ inline u64 canon40(u64 s, u32& flag) {
    u64 r = revcomp40(s); flag = r < s ? 1u : 0u; return flag ? r : s;
}
#define BITMAP_BITS 20                  // LDS first-level filter: 2^20 bits = 128 KiB
__host__ __device__ inline u32 bitmap_hash_bits(u32 lo, u32 hi, u32 bits) { u32 t = seed_premix(lo, hi); t ^= t >> 13; return (t * 0xC2B2AE35u) >> (32 - bits); }
__host__ __device__ inline u32 bitmap_hash(u32 lo, u32 hi) { return bitmap_hash_bits(lo, hi, BITMAP_BITS); }

// ------------------------------------------------------------------ device-side views
struct LocusDev {
    u64 arena_off;      // word offset of this locus in the transposed 2-bit arena
    u64 nmask_off;      // word offset in the N-mask arena (valid iff has_n)
    u32 a_begin, n_alleles, n_pad;   // n_pad = n_alleles rounded up to 64 (row stride of the transposed arena)
    u32 words;          // 2-bit words per allele (ceil(max_len/16) + 2 zero words)
    u32 nwords;         // N-mask words per allele (ceil(max_len/32) + 1)
    u32 has_n, species, max_len;
    u64 plane_off;      // word offset of this locus in the bit-plane arena
    u32 pblocks;        // 32-base blocks per allele in the bit-plane arena (ceil(max_len/32) + 1)
    u32 hap_ok;         // block-haplotype tables present (every block of the locus has fewer than 65,536 haplotypes)
    // block-haplotype tables (k_extend): the distinct (planes, N mask, length) tuples of every 32-base allele block
    u64 hid_off;        // word offset of the locus in hap_id: hap_id[hid_off + (q >> 1) * n_pad + allele] = haplotypes of blocks q (even; low half) and q + 1
    u32 hap_off;        // record offset of the locus in hap_rec
    u32 hblk_off;       // offset in hap_blk of the locus' pblocks + 1 prefix counts: the records of block q are
                        // hap_rec[hap_off + hap_blk[hblk_off + q] ... hap_off + hap_blk[hblk_off + q + 1])
    u32 hap_win[2];     // most haplotype records in any run of 6 / of 11 consecutive blocks (what a read of <= 160 / <= 320 bases covers)
};
// One distinct 32-base block of the alleles of a locus: bit planes, N mask and how many of the 32 columns exist
// (the last block of an allele is shorter; blocks behind an allele's end have len = 0 and act as the identity).
struct HapRec { u32 lo, hi, nm, len; };
struct ItemDev {         // one (read, locus, strand, diagonal) unit of extension work
    u64 res_off;         // offset of its result row in the pair-result arena
    u32 ret;             // retained-read slot
    u32 locus;
    int diag;
    u16 strand, votes;
};
#define EXT_Q 32
struct Counters {
    u64 n_cand, n_ret, n_items, n_res, n_dp, items_done, dp_done, n_pl_dp, ret_done;
    // k_extend work queues: queue q hands out the items begin + q + EXT_Q * t of this submission.  One counter would
    // see ~1 returning atomic per item, and a single word sustains only ~90 of those per microsecond; the queues
    // sit in separate 128-byte lines.
    u64 ext_q[EXT_Q][16];
    u64 err;             // bit0 retained overflow, bit1 item overflow, bit2 result overflow, bit3 dp overflow
    u64 sv_t0n, sv_t1;   // sieve execution window in wall-clock ticks: max over workgroups of ~start and of end (profiling)
    u64 sv_wgmax;        // longest residency of one sieve workgroup (end - start), wall-clock ticks
    u64 cnt[MLST_CNT_N];
    // (new fields go behind this point: the arrays that follow the counters in the statistics block keep their offsets --
    // with these two in front of cnt[], k_extend, whose code had not changed, ran 0.62 instead of 0.55 ms)
    u64 rt_next;         // routed sieve: tiles handed out beyond every producer's first one (zero between submissions)
    u64 rt_parked;       // routed sieve: entries that passed the filter and wait in R.parked for k_route_verify (zero between submissions)
    u64 pad_[14];        // the block stays a multiple of 128 bytes longer than it was in round 2
    u64 ext_q2[EXT_Q][16];      // work queues of k_extend_pairs (as ext_q; 4 KB: the arrays behind keep their alignment)
};
// item_state bits
#define IS_SINGLE 1   /* the read has exactly one work item */
#define IS_DONE   2   /* accumulated by k_extend (fused path) */
#define IS_ACC    4   /* at least one accepted record */
struct KParams {
    int minscore, max_xm, min_read_len, minqual, match_bonus, n_penalty, open_p, ext_p, gbar, band_w, trig, quirk, clip;
};
// A pointer field of the device descriptor.  The fields are loaded from memory at run time, so the compiler cannot
// prove that they point to global memory and would emit flat_* accesses -- slower, and they count on lgkmcnt as
// well as vmcnt, which ties every LDS wait to the outstanding global loads.  On the device the accessor
// round-trips through address space 1, from which the address-space inference makes every use a global_* access.
#if defined(__HIP_DEVICE_COMPILE__)
#define GLOBAL_AS __attribute__((address_space(1)))
#else
#define GLOBAL_AS                  /* host pass: device functions are only type-checked */
#endif
template <typename T> struct GP {
    T* p;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(1))) T G;
#else
    typedef T G;                   // host pass: device functions are only type-checked
#endif
    __host__ __device__ GP& operator=(T* q) { p = q; return *this; }
    template <typename I> __device__ G& operator[](I i) const { return ((G*)p)[i]; }
    __device__ G* operator->() const { return (G*)p; }
    __host__ __device__ operator T*() const { return p; }     // generic pointer: host code, pointer arithmetic
    __device__ G* g() const { return (G*)p; }
};

struct EngineDev {
    // reference
    GP<const u32> arena; GP<const u32> planes; GP<const u32> nmask; GP<const u16> allele_len; GP<const u32> allele_locus; GP<const LocusDev> loci;
    GP<const uint4> sieve; u32 sieve_mask;
    GP<const u32> bitmap;     // first-level 2^20-bit filter kept in LDS (nullptr when the database is too large for it to be selective)
    GP<const u32> gbitmap; u32 gbitmap_bits;   // larger first-level filter in global memory (L2 resident) for big databases
    GP<const u64> keys; GP<const u32> vals; GP<const u32> posts; u32 table_mask;
    GP<const int> floor_tab; GP<const u8> pen_tab;
    u32 n_alleles, n_loci;
    GP<const HapRec> hap_rec; GP<const u32> hap_blk; GP<const u32> hap_id;      // block-haplotype tables (see LocusDev)
    GP<u64> acc64;            // additions of k_extend's fast pass: count << 40 | sum per allele, ONE device addition per accepted record; k_accumulate hands them on (zero between submissions)
    // sample state
    GP<long long> sum_score; GP<u32> n_hits; GP<u64> locus_len; GP<u64> locus_first;
    GP<Counters> ctr;
    GP<u32> ret_bases; GP<u8> ret_quals; GP<u16> ret_len; GP<u64> ret_ridx; GP<u32> ret_nrec;
    GP<u32> ret_cpos;         // position of the read in the submission's candidate list (quality rows that hold the candidates only: mlst_submit_packed_host)
    GP<u32> ret_mate; GP<u32> ret_item0; GP<u8> ret_nitems;      // Q3: slot of the mate (or ~0), first work item and number of items of the read
    GP<ItemDev> items; GP<u8> item_state; GP<u32> res; GP<u64> dp_list;
    u64 cap_ret, cap_items, cap_res, cap_dp;
};

// Batched loads.  hipcc sinks a plain load to its first use, which turns N independent probes into N dependent
// round trips.  An empty asm statement that takes the loaded values as in/out operands is a use the optimiser
// cannot move or split: every load of the batch has to be issued before it, and the compiler itself places the
// (correct, spill-safe) s_waitcnt.  Inline-asm *loads* are not used: their results may be spilled before they land.
#define TIE1(a)             asm volatile("" : "+v"(a))
#define TIE4(a, b, c, d)    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define TIE7(a, b, c, d, e, f, g) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g))
template <int N, typename T> __device__ inline void tie_all(T (&v)[N]) {
    #pragma unroll
    for (int i = 0; i + 4 <= N; i += 4) TIE4(v[i], v[i + 1], v[i + 2], v[i + 3]);
    #pragma unroll
    for (int i = N - (N % 4); i < N; i++) TIE1(v[i]);
}

__device__ inline u64 uniform_u64(u64 v) {
    return ((u64)(u32)__builtin_amdgcn_readfirstlane((u32)(v >> 32)) << 32) | (u64)(u32)__builtin_amdgcn_readfirstlane((u32)v);   // the builtin returns int: cast before widening
}

// ------------------------------------------------------------------ resident read format
// 2-bit rows are stored in groups of 64 reads, transposed in 8-byte units: unit u (words 2u, 2u+1) of read i of a group
// sits at 8-byte index group*64*(wpr/2) + u*64 + i.  A wave whose lanes own the 64 reads of a group reads every unit
// with one fully coalesced 512-byte access (row-major 40-byte rows cost three L2 requests per 128-byte line and
// streamed at 4.3 TB/s; this layout streams at 6.4 TB/s).  A buffer holds ceil(n/64)*64 rows.
__host__ __device__ inline u64 packed_index(u64 r, u32 wpr, u32 c) {
    return (r >> 6) * 64 * wpr + ((((u64)(c >> 1)) * 64 + (r & 63)) << 1) + (c & 1);
}
__host__ __device__ inline u64 packed_words(u64 n_reads, u32 wpr) { return ((n_reads + 63) & ~63ull) * wpr; }

// ------------------------------------------------------------------ K0: pack
// One workgroup per group of 64 reads.  Pass 1: thread = one 16-base word of one read, consecutive threads = consecutive
// words of the same read, so a wave reads its sequence and quality bytes with 16-byte loads from contiguous memory and
// writes contiguous quality rows; the 2-bit word goes to LDS.  Pass 2: the group's words leave LDS in the resident
// (transposed) order as fully coalesced dword stores, and one thread per read stores its length (bit 15 = the read holds a
// non-ACGT base).  No atomics, no scattered global stores (the first version wrote every word to its own line:
// 0.18 TB/s).
// seq_off / qual_off: byte offset of every read's bases / qualities inside `bases` / `quals` (the same array for ASCII
// reads given by one offset table; FASTQ text has separate ones); len_of(r) gives the length.
template <typename LenFn>
__device__ inline void pack_group(const u8* __restrict__ bases, const u8* __restrict__ quals, const u64* __restrict__ seq_off,
                                  const u64* __restrict__ qual_off, LenFn len_of, u64 n_reads, u64 grp,
                                  u32* __restrict__ packed, u8* __restrict__ qrows, u16* __restrict__ lens, u32 wpr, u32 qstride,
                                  u32* s_words /* 64 * wpr */, u32* s_anyn /* 2 */) {
    const int tid = threadIdx.x;
    if (tid < 2) s_anyn[tid] = 0;
    __syncthreads();
    const u32 total = 64u * wpr;
    for (u32 idx = (u32)tid; idx < total; idx += blockDim.x) {
        const u32 i = idx / wpr, w = idx - i * wpr; const u64 r = grp * 64 + i;
        u32 word = 0;
        if (r < n_reads) {
            const u64 so = seq_off[r], qo = qual_off[r]; const u32 n = len_of(r);
            u8 cs[16], cq[16];
            if (w * 16 + 16 <= n) {      // whole word inside the read: two 16-byte copies (the compiler picks the widest loads the target allows unaligned)
                __builtin_memcpy(cs, bases + so + w * 16, 16); __builtin_memcpy(cq, quals + qo + w * 16, 16);
            } else {
                #pragma unroll
                for (int k = 0; k < 16; k++) { const u32 p = w * 16 + k; const bool in = p < n; cs[k] = in ? bases[so + p] : (u8)'A'; cq[k] = in ? quals[qo + p] : (u8)33; }
            }
            // sixteen bases and qualities, four at a time in 32-bit words (round 5; a compare chain per character before: ~350
            // instructions per word).  Bytes beyond the read are 'A' / '!' here (see above) and come out as zeros, as before.
            u32 anyn = 0, qw[4] = {0, 0, 0, 0};
            #pragma unroll
            for (int j = 0; j < 4; j++) {
                u32 b4, q4; __builtin_memcpy(&b4, cs + 4 * j, 4); __builtin_memcpy(&q4, cq + 4 * j, 4);
                const u32 y = (b4 >> 1) & 0x03030303u, y0 = y & 0x01010101u, y1 = (y >> 1) & 0x01010101u;      // bits 1-2 of a letter: A 0, C 1, T 2, G 3 (either case)
                u32 code = y ^ y1;                                                                       // A 0, C 1, G 2, T 3
                const u32 expect = 0x41414141u + y0 * 2u + y1 * 0x13u - (y0 & y1) * 0x0Fu;                  // the upper-case letter those bits stand for
                const u32 diff = (b4 & 0xDFDFDFDFu) ^ expect;                                              // non-zero byte: not one of ACGT
                const u32 isn = ((diff | ((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu)) >> 7) & 0x01010101u;
                code &= ~(isn * 3u);                                                                      // (packed as A)
                word |= ((code * 0x01041040u) >> 24) << (8 * j);                                          // four 2-bit codes -> one byte
                // Phred = character - 33, kept in 0..127
                u32 qv;
                const u32 lowc = (q4 | 0x80808080u) - 0x21212121u;                                         // bit 7 of a byte survives iff its character is >= 33
                const u32 okl = ((lowc & 0x80808080u) >> 7) | ((q4 & 0x80808080u) >> 7);                    // 1: character >= 33
                if (okl == 0x01010101u && !(q4 & 0x80808080u)) qv = q4 - 0x21212121u;                      // the usual case: 33..127 everywhere
                else {
                    qv = 0;
                    #pragma unroll
                    for (int k = 0; k < 4; k++) { int q = (int)((q4 >> (8 * k)) & 0xFFu) - 33; q = q < 0 ? 0 : (q > 127 ? 127 : q); qv |= (u32)q << (8 * k); }
                }
                qw[j] = qv | (isn << 7);
                anyn |= isn;
            }
            anyn = anyn ? 1u : 0u;
            #pragma unroll
            for (int j = 0; j < 4; j++) if (w * 16 + 4 * j < qstride) reinterpret_cast<u32*>(qrows + r * qstride)[w * 4 + j] = qw[j];     // qstride is a multiple of 4
            if (anyn) atomicOr(&s_anyn[i >> 5], 1u << (i & 31));      // LDS
        }
        s_words[i * wpr + w] = word;
    }
    __syncthreads();
    // resident order: word o of the group = unit o >> 7, read (o >> 1) & 63, half o & 1
    u32* out = packed + grp * total;
    for (u32 o = (u32)tid; o < total; o += blockDim.x) out[o] = s_words[((o >> 1) & 63u) * wpr + ((o >> 7) << 1) + (o & 1u)];
    if (tid < 64) { const u64 r = grp * 64 + tid; if (r < n_reads) lens[r] = (u16)(len_of(r) | (((s_anyn[tid >> 5] >> (tid & 31)) & 1u) << 15)); }
}
__global__ __launch_bounds__(256) void k_pack(const u8* __restrict__ bases, const u8* __restrict__ quals,
                                               const u64* __restrict__ off, u64 n_reads, u32* __restrict__ packed,
                                               u8* __restrict__ qrows, u16* __restrict__ lens, u32 wpr, u32 qstride) {
    __shared__ u32 s_words[64 * RW]; __shared__ u32 s_anyn[2];
    const u64 n_groups = (n_reads + 63) >> 6;
    for (u64 grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        pack_group(bases, quals, off, off, [&](u64 r) { return (u32)(off[r + 1] - off[r]); }, n_reads, grp, packed, qrows, lens, wpr, qstride, s_words, s_anyn);
        __syncthreads();
    }
}

// ------------------------------------------------------------------ K1: seed sieve (the streaming kernel)
// lane = read, wave = one 64-read group of the transposed layout.  A seed is the 20-mer at every 16th base = word t plus
// the low byte of word t+1.  Two levels: a first-level bitmap test for every seed (LDS half-seed bitmaps for small
// databases, a hashed global bitmap for big ones), then the seeds that pass, compacted per wave into an LDS queue,
// against the sieve proper: 16-byte buckets of eight 16-bit fingerprints of the canonical seed (one 16-byte load per
// queue entry).  Reads with any hit go to the candidate list with one atomic per wave.
__device__ inline bool bucket_has(uint4 b, u32 fp, bool& full) {
    u32 pat = fp * 0x00010001u;
    u32 x0 = b.x ^ pat, x1 = b.y ^ pat, x2 = b.z ^ pat, x3 = b.w ^ pat;
    // zero-halfword test
    u32 z = ((x0 - 0x00010001u) & ~x0) | ((x1 - 0x00010001u) & ~x1) | ((x2 - 0x00010001u) & ~x2) | ((x3 - 0x00010001u) & ~x3);
    full = (b.w >> 16) != 0;        // slots are filled in order: the bucket is full iff its last slot is occupied
    return (z & 0x80008000u) != 0;
}

// Sieve with an LDS-resident first level.  One 1024-thread workgroup per CU keeps two 2^19-bit bitmaps in LDS (128 KiB),
// each indexed DIRECTLY by one 10-base half of the seed (top bit dropped), read in an orientation that does not depend
// on the strand:
//   A = bases 0..9, B = bases 10..19, rcA / rcB their reverse complements.  The seed reads (A, B), its reverse
//   complement reads (rcB, rcA); sieve_half_flip(A, rcB) chooses one of the two readings by a circular comparison
//   (which keeps the chosen halves uniformly distributed); L = its first half, R = its second half.
// A seed passes when bit L of the first bitmap and bit R of the second are set.  Indexing by half of the bases makes
// the alleles' SNP variants collapse -- a SNP changes 20 seeds but only ~10 distinct halves -- so each bitmap of an
// MLST database is far emptier than a hashed bitmap of whole seeds would be, no multiply is needed, and the two tests
// multiply (7 x 1430-allele database: 326 k seeds, each half bitmap ~23 % full, ~5 % of random seeds pass, against
// 27 % for one hashed 2^20-bit bitmap).  The builder sets the bits of BOTH readings of every database seed, so the
// choice never has to be consistent across strands.
// Seeds that pass are compacted per wave into an LDS queue and only the queue is checked against the fingerprint
// sieve -- about half a round of 64 probes per 64 reads instead of nine.  Rows are read straight from global memory
// (lane = read, coalesced 8-byte non-temporal loads, the next tile requested a whole tile ahead).
#define SV_CAP 224            // queue entries per wave (8 bytes each); drained early when it could overflow
#define SV_HALF_BITS (1u << (BITMAP_BITS - 1))      // bits per half bitmap (2^19 each: first half | second half = 128 KiB)
__host__ __device__ inline bool sieve_half_flip(u32 A, u32 rcB) { return ((A - rcB) & 0x80000u) != 0; }
// host restatement for the builder: s = 40-bit seed, base t at bits 2t -> the two bitmap indices of s read in the
// orientation the pick chooses (L = chosen first half, R = the half that follows it in that orientation)
__host__ inline void sieve_halves_of(u64 s, u32& L, u32& R) {
    u32 A = (u32)(s & 0xFFFFFu), B = (u32)((s >> 20) & 0xFFFFFu), rcA = 0, rcB = 0;
    for (int q = 0; q < 10; q++) {
        rcB |= (u32)(3u - (u32)((s >> (2 * (19 - q))) & 3u)) << (2 * q);
        rcA |= (u32)(3u - (u32)((s >> (2 * (9 - q))) & 3u)) << (2 * q);
    }
    const bool flip = sieve_half_flip(A, rcB);
    L = (flip ? rcB : A) & (SV_HALF_BITS - 1u); R = (flip ? rcA : B) & (SV_HALF_BITS - 1u);
}
struct SvProbe { v4u bv; u32 fp, bi, src; bool act; };
// read queue entry e (if any), canonical key, request its sieve bucket
template <bool NT>
__device__ inline void sv_issue(SvProbe& P, const u64* queue, u32 e, u32 cnt, const uint4* __restrict__ sieve, u32 sshift) {
    P.act = e < cnt;
    const u64 ent = P.act ? queue[e] : 0ull;
    u32 fl; const u64 c = canon40(ent & 0xFFFFFFFFFFull, fl);
    const u32 klo = (u32)c, khi = (u32)(c >> 32);
    P.src = (u32)(ent >> 40) & 63u;
    P.fp = sieve_fp(klo, khi);
    P.bi = P.act ? sieve_bucket_hash(klo, khi) >> sshift : 0u;
    // a sieve that does not fit L2 is probed with non-temporal loads, which keeps the first-level bitmap resident there
    if (NT) P.bv = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(sieve) + P.bi);
    else P.bv = reinterpret_cast<const v4u*>(sieve)[P.bi];
}
// examine a requested bucket; a hit sets the bit of the seed's read (= lane of the tile) in the wave's hit mask
__device__ inline void sv_check(const SvProbe& P, const uint4* __restrict__ sieve, u32 smask, u32* hitw) {
    bool full; bool hit = bucket_has(make_uint4(P.bv.x, P.bv.y, P.bv.z, P.bv.w), P.fp, full) && P.act;
    if (P.act && !hit && full) {   // overflow chain: the key may sit in a following bucket
        u32 bi = P.bi;
        for (int step = 0; step < 64; step++) {
            bi = (bi + 1) & smask; bool f2; uint4 bb = sieve[bi];
            if (bucket_has(bb, P.fp, f2)) { hit = true; break; }
            if (!f2) break;
        }
    }
    if (hit) atomicOr(&hitw[P.src >> 5], 1u << (P.src & 31));
}
// append the reads of one tile whose bit is set in the wave's hit mask to the candidate list; clears the mask
// paired: mates (reads 2k, 2k+1: neighbouring lanes) become candidates together, so that they sit side by side in the list and
// k_seed can link them (Q3: sequenceBank is keyed by QNAME, metamlst.py:127); a superfluous candidate costs one exact lookup
__device__ inline void sv_emit(u32* hitw, u32* __restrict__ cand, Counters* __restrict__ ctr, u64 r, int lane, int paired) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    u64 mask = uniform_u64((u64)hitw[0] | ((u64)hitw[1] << 32));
    if (paired) mask |= ((mask & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((mask & 0x5555555555555555ull) << 1);
    if (mask) {
        if (lane < 2) hitw[lane] = 0;
        u64 base = 0;
        if (lane == 0) base = atomicAdd(&ctr->n_cand, (u64)__popcll(mask));
        base = __shfl(base, 0);
        if ((mask >> lane) & 1ull) cand[base + __popcll(mask & ((1ull << lane) - 1))] = (u32)r;
    }
}
// LDSBM = true : the first level is the pair of half-seed bitmaps in LDS (small databases), 16 waves per workgroup.
// LDSBM = false: big databases, whose half seeds saturate any LDS-sized bitmap: the first level is a hashed bitmap of
//   the canonical seed in global memory (2^gbm_bits bits, mostly L2 resident); 4 waves per workgroup, several
//   workgroups per CU; three bucket requests stay in flight (a third of the seeds pass this level).
template <int WPR, bool LDSBM>
__global__ __launch_bounds__(LDSBM ? 1024 : 256) void k_sieve_q(const u32* __restrict__ packed, const u16* __restrict__ lens, u64 n_reads,
                                                               const uint4* __restrict__ sieve, u32 smask, const u32* __restrict__ bitmap,
                                                               u32 gbm_bits, u32* __restrict__ cand, Counters* __restrict__ ctr, int paired) {
    constexpr int NW = LDSBM ? 16 : 4;                 // waves per workgroup; a tile = NW groups of 64 reads
    constexpr int NP = LDSBM ? 2 : 3;                  // bucket requests kept in flight across the first level
    constexpr u32 TILE = NW * 64;
    // one LDS block, queue first so that its addresses fit the DS offset field: [queues NW x SV_CAP x 8 B][hit masks][bitmap]
    constexpr u32 Q_WORDS = NW * SV_CAP * 2, H_WORDS = 2 * NW, BM_OFF = Q_WORDS + H_WORDS;
    __shared__ __attribute__((aligned(16))) u32 s_all[BM_OFF + (LDSBM ? (1u << BITMAP_BITS) / 32 : 4)];
    u32* const s_bm = s_all + BM_OFF;
    const u32 sshift = (u32)__clz((int)smask);       // buckets = smask + 1 = 2^(32 - sshift)
    constexpr int NT = WPR - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u64* const queue = reinterpret_cast<u64*>(s_all) + wave * SV_CAP;
    u32* const hitw = s_all + Q_WORDS + wave * 2;
    const u64 wg_t0 = (u64)wall_clock64();
    if (tid == 0) atomicMax(&ctr->sv_t0n, ~wg_t0);      // execution window (excludes queueing behind other streams)
    if (LDSBM) {   // 32768 words, 16-byte vectors
        const v4u* g4 = reinterpret_cast<const v4u*>(bitmap); v4u* s4 = reinterpret_cast<v4u*>(s_bm);
        #pragma unroll
        for (int k = 0; k < 8; k++) s4[tid + 1024 * k] = g4[tid + 1024 * k];
    }
    if (tid < (int)H_WORDS) s_all[Q_WORDS + tid] = 0;
    __syncthreads();
    const u64 n_tiles = (n_reads + TILE - 1) / TILE;
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    v2u xn[WPR / 2]; u16 len_raw = 0; bool live_next = false;
    const u64 n_groups = (n_reads + 63) >> 6;      // the wave's 64 lanes own one group of the transposed layout
    {
        u64 r = (u64)blockIdx.x * TILE + tid; u64 grp = (u64)blockIdx.x * NW + wave;
        live_next = blockIdx.x < n_tiles && r < n_reads;
        const v2u* row = reinterpret_cast<const v2u*>(packed) + (grp < n_groups ? grp : 0) * (32 * WPR) + lane;
        #pragma unroll
        for (int t2 = 0; t2 < WPR / 2; t2++) xn[t2] = __builtin_nontemporal_load(row + t2 * 64);
        len_raw = lens[live_next ? r : 0];
    }
    // Order inside an iteration (tile k): request the sieve buckets of the seeds queued by tile k-1, request the rows of
    // tile k+1, run the first level of tile k (which hides both latencies), examine the buckets and emit the candidates
    // of tile k-1, queue the passing seeds of tile k.  The only wait for the buckets sits behind the first level, and
    // the row requests are younger than the bucket requests, so waiting for the buckets leaves them in flight.
    u32 qcnt = 0;                 // seeds queued by the previous tile (entries 0..qcnt-1 of the wave's queue)
    u64 last_tile = blockIdx.x;
    for (u64 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const bool live_cur = live_next;
        // explicit register copies: xn's registers are free again right here, so the loads below can write into them
        // directly (left to itself the compiler copies at the loop latch instead, and that copy waits for the loads)
        u32 w[WPR]; u32 len_cur;
        #pragma unroll
        for (int t2 = 0; t2 < WPR / 2; t2++) {
            asm volatile("v_mov_b32 %0, %1" : "=v"(w[2 * t2]) : "v"(xn[t2].x));
            asm volatile("v_mov_b32 %0, %1" : "=v"(w[2 * t2 + 1]) : "v"(xn[t2].y));
        }
        asm volatile("v_mov_b32 %0, %1" : "=v"(len_cur) : "v"((u32)len_raw));
        const u32 n = live_cur ? (len_cur & 0x7FFFu) : 0u;
        const int nseeds = n >= MLST_SEED_LEN ? (int)((n - MLST_SEED_LEN) / MLST_SEED_STEP) + 1 : 0;
        // ---- buckets of the previous tile's queue: NP requests stay in flight across the first level (a request is
        // made only when the queue reaches it: wave-uniform); a longer queue is examined on the spot
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        SvProbe PR[NP];
        #pragma unroll
        for (int j = 0; j < NP; j++) {
            if (j == 0 || qcnt > 64u * j) sv_issue<!LDSBM>(PR[j], queue, 64u * j + (u32)lane, qcnt, sieve, sshift);
            else { PR[j].act = false; PR[j].bv = v4u{0u, 0u, 0u, 0u}; PR[j].fp = 1u; PR[j].bi = 0u; PR[j].src = 0u; }
        }
        for (u32 base = 64u * NP; base < qcnt; base += 64) {
            SvProbe Ps; sv_issue<!LDSBM>(Ps, queue, base + (u32)lane, qcnt, sieve, sshift);
            sv_check(Ps, sieve, smask, hitw);
        }
        asm volatile("" ::: "memory");      // the bucket requests are older than the row requests below: waiting for them leaves the rows in flight
        {   // rows of the next tile.  The memory clobber keeps the compiler from sinking the loads towards their first use.
            u64 tn = tile + gridDim.x; u64 rn = tn * TILE + tid;
            live_next = tn < n_tiles && rn < n_reads;
            const u64 gn = tn * NW + wave;
            const v2u* row = reinterpret_cast<const v2u*>(packed) + (gn < n_groups ? gn : 0) * (32 * WPR) + lane;
            #pragma unroll
            for (int t2 = 0; t2 < WPR / 2; t2++) xn[t2] = __builtin_nontemporal_load(row + t2 * 64);
            len_raw = lens[live_next ? rn : 0];
            asm volatile("" ::: "memory");
        }
        const bool all_full = __all(nseeds >= NT) != 0;
        u64 pm[NT]; u32 elo[NT], ehi[NT];            // pass masks; the queue entry of every seed (40 key bits)
        if (LDSBM) {
            // ---- first level in LDS: per word the bit-reversed swapped complement (rb[i] base p = complement of w[i]
            // base 15-p), per seed the two halves, the pick and two LDS bits.  Entry = the seed as it stands.
            u32 rb[WPR];
            #pragma unroll
            for (int i = 0; i < WPR; i++) {
                u32 x = w[i];
                rb[i] = __brev(~(((x >> 1) & 0x55555555u) | ((x + x) & 0xAAAAAAAAu)));
            }
            #pragma unroll
            for (int t = 0; t < NT; t++) {
                const u32 A = w[t] & 0xFFFFFu, B = __builtin_amdgcn_alignbit(w[t + 1], w[t], 20) & 0xFFFFFu;
                const u32 rcB = __builtin_amdgcn_alignbit(rb[t], rb[t + 1], 24) & 0xFFFFFu, rcA = rb[t] >> 12;
                const bool flip = sieve_half_flip(A, rcB);
                const u32 L = (flip ? rcB : A) & (SV_HALF_BITS - 1u), R = (flip ? rcA : B) & (SV_HALF_BITS - 1u);
                const u32 wl = s_bm[L >> 5], wr = s_bm[SV_HALF_BITS / 32 + (R >> 5)];
                pm[t] = __ballot((((wl >> (L & 31)) & (wr >> (R & 31))) & 1u) != 0);
                elo[t] = w[t]; ehi[t] = w[t + 1] & 0xFFu;
            }
        } else {
            // ---- first level in global memory: canonical keys, then every bitmap word requested before any is looked
            // at.  Entry = the canonical key (canon40 is idempotent, so sv_issue treats it like any seed).
            u32 gi[NT], gw[NT];
            #pragma unroll
            for (int t = 0; t < NT; t++) {
                u32 fl; const u64 c = canon40((u64)w[t] | ((u64)(w[t + 1] & 0xFFu) << 32), fl);
                elo[t] = (u32)c; ehi[t] = (u32)(c >> 32);
                gi[t] = bitmap ? bitmap_hash_bits(elo[t], ehi[t], gbm_bits) : 0u;
                gw[t] = bitmap ? bitmap[gi[t] >> 5] : 0xFFFFFFFFu;
            }
            tie_all<NT>(gw);
            #pragma unroll
            for (int t = 0; t < NT; t++) pm[t] = __ballot(((gw[t] >> (gi[t] & 31)) & 1u) != 0);
        }
        if (!all_full) {
            #pragma unroll
            for (int t = 0; t < NT; t++) pm[t] &= __ballot(t < nseeds);
        }
        // ---- examine the buckets requested above; candidates of the previous tile
        #pragma unroll
        for (int j = 0; j < NP; j++) if (j == 0 || qcnt > 64u * j) sv_check(PR[j], sieve, smask, hitw);
        sv_emit(hitw, cand, ctr, last_tile * TILE + tid, lane, paired);
        // ---- queue the passing seeds of this tile (probed during the next iteration).  If the queue could overflow
        // (dense on-locus data) it is drained on the spot and filling continues.
        u32 cnt = 0; int t0 = 0;
        for (;;) {
            int t_next = NT;
            #pragma unroll
            for (int t = 0; t < NT; t++) {
                if (t < t0 || t >= t_next) continue;
                const u64 m = pm[t];
                if (m == 0) continue;
                if (cnt > SV_CAP - 64) { t_next = t; continue; }
                u32 pos = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, cnt));
                if (__builtin_amdgcn_inverse_ballot_w64(m))
                    queue[pos] = (u64)elo[t] | ((u64)(ehi[t] | ((u32)lane << 8)) << 32);
                cnt += (u32)__popcll(m);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (t_next == NT) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (u32 base = 0; base < cnt; base += 64) {
                SvProbe Ps; sv_issue<!LDSBM>(Ps, queue, base + (u32)lane, cnt, sieve, sshift);
                sv_check(Ps, sieve, smask, hitw);
            }
            cnt = 0; t0 = t_next;
        }
        qcnt = cnt; last_tile = tile;
    }
    // the last tile's queue
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (u32 base = 0; base < qcnt; base += 64) {
        SvProbe Ps; sv_issue<!LDSBM>(Ps, queue, base + (u32)lane, qcnt, sieve, sshift);
        sv_check(Ps, sieve, smask, hitw);
    }
    sv_emit(hitw, cand, ctr, last_tile * TILE + tid, lane, paired);
    __syncthreads();
    if (tid == 0) { const u64 t1 = (u64)wall_clock64(); atomicMax(&ctr->sv_t1, t1); atomicMax(&ctr->sv_wgmax, t1 - wg_t0); }
}

// ---- DPP lane moves (gfx9 encodings).  update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl): a lane whose
// source is out of range, or whose row is masked off, keeps `old`.
#define DPP_ROW_SHR(n)  (0x110 + (n))
#define DPP_WAVE_SHL1   0x130      /* lane i <- lane i+1 */
#define DPP_WAVE_SHR1   0x138      /* lane i <- lane i-1 */
#define DPP_ROW_BCAST15 0x142      /* lane 15 of each 16-lane row -> every lane of the next row */
#define DPP_ROW_BCAST31 0x143      /* lane 31 -> every lane of rows 2 and 3 */
// ------------------------------------------------------------------ wave helpers
__device__ inline u32 wave_excl_scan_u32(u32 v, u32& total) {      // exclusive prefix sum over the 64 lanes
    int lane = threadIdx.x & 63; u32 x = v;
    #pragma unroll
    for (int o = 1; o < 64; o <<= 1) { u32 y = __shfl_up(x, o); if (lane >= o) x += y; }
    total = __shfl(x, 63);
    return x - v;
}
__device__ inline u64 wave_sum_u64(u64 v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ inline u64 wave_min_u64(u64 v) { for (int o = 32; o > 0; o >>= 1) { u64 w = __shfl_xor(v, o); v = w < v ? w : v; } return v; }
__device__ inline u32 wave_sum_u32(u32 v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
// exact check of one seed: canonical key -> fingerprint sieve; true = candidate
__device__ inline bool bin_exact(u32 w0, u32 w1, const uint4* __restrict__ sieve, u32 smask, u32 sshift) {
    u32 fl; const u64 c = canon40((u64)w0 | ((u64)(w1 & 0xFFu) << 32), fl);
    const u32 klo = (u32)c, khi = (u32)(c >> 32), fp = sieve_fp(klo, khi);
    u32 bi = sieve_bucket_hash(klo, khi) >> sshift;
    for (int step = 0; step < 65; step++) {
        bool full; const uint4 bb = sieve[bi];
        if (bucket_has(bb, fp, full)) return true;
        if (!full) return false;
        bi = (bi + 1) & smask;
    }
    return false;
}
// ------------------------------------------------------------------ K1c: CU-routed sieve (databases beyond the LDS half-seed bitmaps)
// A filter for millions of seeds needs ~16 MiB.  The only on-chip memories of that aggregate size are the eight L2s
// (4 MiB each, ~270 G random requests/s in all: the 450 M seeds of a 50 M-read batch are 1.7 ms of L2 requests alone --
// round 1's XCD-binned sieve, 4.25 ms per 50 M reads, was bound by exactly that; profiles/microbench/xcd_probe.hip)
// and the 256 LDS (128 KiB usable each, 32 lanes per clock per CU: ~60 x the L2 request rate).  So
// the filter is cut into 256 slices by key hash, one slice per CU, and every seed is ROUTED to the CU that owns its slice:
//   k_route        streams the reads once (1024-thread workgroup = tile of 16 groups of 64 reads); per seed the canonical
//                  key and 33 hash bits: 8 choose the owner, 25 address the owner's filter.  The tile's seeds are
//                  counting-sorted by (owner, wave) in LDS and each owner's run is appended to the region
//                  (owner, this workgroup) of the arena -- contiguous stores of ~150 bytes that the L2 merges into whole
//                  lines (the open lines of an XCD's workgroups are ~2 MiB).  A 4-byte entry = first-of-(owner, wave)
//                  flag | lane | 25 hash bits; the read index is implied by the position in the region: the
//                  consumer counts the flags (every (owner, wave) pair of a tile contributes at least a dummy entry).
//   k_route_probe  one workgroup per owner holds the owner's 128 KiB filter slice in LDS (three bits in each of two 32-bit
//                  words per key, rt_filter_addr: ~0.3 % of foreign seeds pass, 0.2 % of them collisions of the 33 hash bits), streams the owner's regions 16 bytes per lane, and sends the
//                  entries that pass the exact way: the read's seeds are re-hashed, the one(s) equal to the entry's hash
//                  probe the fingerprint sieve, a hit sets the read's candidate flag.
//   k_flag_compact candidate flags -> candidate list.
// A tile that would overflow a region or the sort buffer (only degenerate data: low-complexity reads crowd one owner) is
// not routed at all: its reads become candidates outright (k_seed looks every candidate up exactly), and the producer's
// list of emitted tiles tells the consumer which tile a flag count belongs to.
#define RT_OWNERS 256
#define RT_FWORDS 32768                // 32-bit words of one owner's filter slice (128 KiB)
#define RT_MAXP  2048                  // most producer workgroups (regions per owner) a submission may use
#define RT_FLAG  0x80000000u           // entry = RT_FLAG | lane << 25 | 25 hash bits; RT_DUMMY in the hash bits = "no seed here"
#define RT_HMASK 0x01FFFFFFu
#define RT_DUMMY 0x01FFFFFFu
// 33 hash bits of a canonical key: 8 choose the owner, 25 travel in the entry (a real key that hashes to RT_DUMMY takes
// the value below it -- in the filter build, in k_route and in the re-hash of k_route_probe alike).
// One 32 x 32 -> 64-bit multiply of the key's low word; 32 bits from the MIDDLE of the product (bits 16..47: each of them
// depends on the key bits below it and, through the carries, on those above), XORed with a multiple of the key's top byte,
// plus product bit 15.  Simulated on 16 M random keys: 0.189 % of random seeds share all 33 bits with a key (ideal
// 2^-33 per pair: 0.186 %).  A first version folded the 40 key bits to 32 BEFORE the multiply: 33 bits cut out of a
// 32-bit quantity collide seven times as often (1.3 %, 7.7 M instead of 2.9 M entries to examine: the consumer ran 0.3 ms
// longer).  Round 2 ran table_hash here (two multiplies, two xor-shifts) plus a parity: ~22 VALU instructions per seed,
// three of them quarter rate; this is 9 with one multiply, and k_route hashes 450 M seeds per batch.
__host__ __device__ inline void rt_hash(u32 lo, u32 hi, u32& owner, u32& h25) {
    const u64 P = (u64)lo * 0x9E3779B1u;
    const u32 x = (u32)(P >> 16) ^ (hi * 0x9E3779u);     // hi < 256: a 24-bit multiply (full rate)
    owner = x >> 24;
    h25 = ((x & 0xFFFFFFu) << 1) | ((u32)(P >> 15) & 1u);
    h25 = h25 < RT_DUMMY - 1u ? h25 : RT_DUMMY - 1u;
}
// Filter slice = 2^15 words of 32 bits (128 KiB), 16 bits per key.  A key sets three independently chosen bits in each
// of TWO words: 0.19 % of foreign seeds pass (measured and simulated), beside the 0.19 % that share all 33 hash bits with
// a database key -- every pass costs the consumer ~0.7 KB of row and bucket traffic.  Cheaper masks were tried in round
// 3 (a rotated pair of bits + one bit per word: two instructions fewer per entry, 0.44 % pass: the consumer ran 0.3 ms
// LONGER; a rotated triple: 1.8 %); with v_lshl_or the three bit positions cost six instructions per word as they are.
__host__ __device__ inline void rt_filter_addr(u32 h25, u32& word0, u32& mask0, u32& word1, u32& mask1) {
    const u32 m = h25 * 0x9E3779B1u;               // top bits of a product are the well mixed ones
    word0 = h25 >> 10;
    mask0 = (1u << (m >> 27)) | (1u << ((m >> 22) & 31u)) | (1u << ((m >> 17) & 31u));
    word1 = ((m >> 9) ^ h25) & 0x7FFFu;
    mask1 = (1u << ((m >> 12) & 31u)) | (1u << ((m >> 7) & 31u)) | (1u << ((m >> 2) & 31u));
}
struct RouteDev {
    GP<u32> arena;                   // [owner][producer][cap] entries
    GP<u32> counts;                  // [owner][producer] entries written
    GP<u32> emitted;                 // [producer][1 + tiles_max]: number of tiles routed, then their iteration numbers
    GP<const u32> filter;            // [owner][RT_FWORDS]
    GP<u32> flags;                   // candidate flag per read (zeroed per submission)
    GP<u64> trace;                   // diagnostics (mlst_get_route_trace), NULL when off: four words per workgroup
    GP<u64> parked; u64 parked_cap;  // entries that passed the filter (rt_park), examined by k_route_verify
    u32 cap, n_prod, tiles_max, nw;  // nw = waves per producer workgroup = groups of 64 reads per tile
    u32 dbg;                         // profiling builds of the launch: bit 0 = entries that pass the filter are not examined (timing only, results wrong)
};
// where and when a workgroup ran: XCC id | HW_ID << 32, wall clock at its start; the end is stored by rt_trace_end
__device__ inline void rt_trace_begin(const RouteDev& R, u32 slot) {
    if (R.trace.p) {
        const u32 xcc = (u32)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // HW_REG_XCC_ID[3:0]
        const u32 hw = (u32)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID
        R.trace[(u64)slot * 4] = (u64)xcc | ((u64)hw << 32); R.trace[(u64)slot * 4 + 1] = (u64)wall_clock64();
    }
}
__device__ inline void rt_trace_end(const RouteDev& R, u32 slot) { if (R.trace.p) R.trace[(u64)slot * 4 + 2] = (u64)wall_clock64(); }
// inclusive prefix sum over the 64 lanes with DPP lane moves (no LDS traffic, unlike __shfl_up)
__device__ inline u32 wave_incl_scan_dpp(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR(1), 0xF, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR(2), 0xF, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR(4), 0xF, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR(8), 0xF, 0xF, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_BCAST15, 0xA, 0xF, false);      // rows 1 and 3 take the total of rows 0 and 2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_BCAST31, 0xC, 0xF, false);      // rows 2 and 3 take the total of the first half
    return v;
}
// NW = waves per workgroup = groups of 64 reads per tile: 16 (one or two workgroups per CU) or 8 (up to four)
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global-memory
// operation of the wave (it is a workgroup-scope release): inside k_route's tile loop that would stall every barrier on
// the prefetched rows of the next tile and on the stores of the segment write-out, which no other wave of the
// workgroup ever reads.  What the waves share is LDS: lgkmcnt(0) before s_barrier makes a wave's LDS writes visible to
// the waves that pass the barrier.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int WPR, int NW>
__attribute__((amdgpu_waves_per_eu(WPR <= 10 ? 8 : 4, WPR <= 10 ? 8 : 4)))      // reads up to 160 bases: 64 VGPRs, so that the LDS decides how many workgroups share a CU
__global__ __launch_bounds__(NW * 64) void k_route(const u32* __restrict__ packed, const u16* __restrict__ lens, u64 n_reads,
                                                   const RouteDev R, Counters* __restrict__ ctr) {
    constexpr int NT = WPR - 1;
    constexpr int QN = NW / 4;                    // thread slices of 256: slice q holds the runs of waves 4q .. 4q+3
    constexpr u32 TILE = NW * 64u;
    constexpr u32 SCAP = TILE * NT + (NW == 16 ? 1024u : 768u);       // + dummy entries of empty runs (typically ~5 %)
    constexpr u32 CW = 16;                        // entries per 64-byte chunk: what leaves for a region is whole, aligned chunks
    constexpr int OPW = RT_OWNERS / NW;           // owners whose segments a wave writes out
    __shared__ u32 s_cnt[NW][RT_OWNERS];          // per (wave, owner): count, later the start of the run in s_sorted
    __shared__ __attribute__((aligned(16))) u32 s_sorted[SCAP + 4];      // + a spare word for slots without a seed
    __shared__ u32 s_part[QN][RT_OWNERS];         // entries of an owner's runs per slice of the waves
    __shared__ u32 s_off[RT_OWNERS + 1];          // start of each owner's segment in s_sorted (multiples of four)
    __shared__ u32 s_cur[RT_OWNERS];              // where region (owner, this workgroup) continues, as the number of the 64-byte chunk in the ARENA (one shift away from the address: the region's base was three 64-bit multiply-adds per owner and tile)
    __shared__ u32 s_carry[RT_OWNERS][CW];        // the entries of an owner that did not fill a chunk yet (fewer than CW), oldest first
    __shared__ u32 s_cn[RT_OWNERS];               // how many
    __shared__ u32 s_wsum[4]; __shared__ u32 s_over;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // the wave number in a scalar register
    const u32 p = blockIdx.x, P = gridDim.x;
    const u32 so = (u32)tid & 255u, sq = (u32)tid >> 8;
    if (tid == 0) { atomicMax(&ctr->sv_t0n, ~(u64)wall_clock64()); rt_trace_begin(R, p); }
    #pragma unroll
    for (int v = 0; v < 4; v++) s_cnt[sq * 4 + v][so] = 0;
    const u32 cap_ch = R.cap / CW;                // chunks per region (the capacity is a multiple of 32 entries)
    if (tid < RT_OWNERS) { s_cur[tid] = ((u32)tid * P + p) * cap_ch; s_cn[tid] = 0; }
    if (tid == 0) s_over = 0;
    __syncthreads();
    const u64 n_groups = (n_reads + 63) >> 6, n_tiles = (n_reads + TILE - 1) / TILE;
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    v2u xn[WPR / 2]; u16 len_raw = 0;
    {
        const u64 g0 = (u64)p * NW + wave, gc = g0 < n_groups ? g0 : 0, r0 = gc * 64 + lane;
        const v2u* row = reinterpret_cast<const v2u*>(packed) + gc * (32 * WPR) + lane;
        #pragma unroll
        for (int t2 = 0; t2 < WPR / 2; t2++) xn[t2] = __builtin_nontemporal_load(row + t2 * 64);
        len_raw = lens[r0 < n_reads ? r0 : 0];
    }
    // Tiles are handed out by a counter (ctr->rt_next), not dealt in a fixed stride: with a fixed share the workgroups of one
    // launch finished between 0.91 and 1.33 ms (profiles/round3/route_modes.md: same work, different CUs and partners) and
    // the launch took as long as the slowest.  A workgroup's first tile is its own number; while it works on tile i it holds
    // the number of tile i+1 (whose rows are being prefetched) and has asked for that of tile i+2 (thread 0, answer parked in
    // LDS by the end of the iteration), so the counter's latency never shows.  The consumer learns the tile numbers from
    // the producer's list (R.emitted).
    __shared__ u32 s_tile;
    u32 n_emit = 0;                               // tiles routed so far (thread 0 keeps the list)
    u64 t_nxt;
    {
        if (tid == 0) s_tile = P + (u32)atomicAdd(&ctr->rt_next, 1ull);
        __syncthreads();
        t_nxt = s_tile;
        __syncthreads();
    }
    u32 n_taken = 2;                              // tiles this workgroup has claimed (thread 0's copy decides)
    for (u64 tile = p; tile < n_tiles; ) {
        // Per-thread addresses (LDS slots, rows, regions) are all functions of the thread index.  Left alone, the compiler
        // computes dozens of them once, keeps them across the tile loop and spills them at 64 registers; every scratch
        // reload then carries an s_waitcnt vmcnt(0), which (vmcnt is in order) also waits for the prefetched rows and for
        // the segment stores.  The thread index is re-read through an opaque copy per tile instead: a few integer
        // operations per phase, no spill.
        int td = tid; asm volatile("" : "+v"(td));
        const int ln = td & 63; const u32 so_ = (u32)td & 255u, sq_ = (u32)td >> 8;
        // the tile after next: asked for now, parked in LDS at the end of the iteration.  A workgroup whose list of tiles or
        // whose regions are nearly full stops asking (the others take what is left).
        u32 t_nn = 0xFFFFFFFFu;
        if (td == 0 && n_taken < R.tiles_max) { t_nn = P + (u32)atomicAdd(&ctr->rt_next, 1ull); n_taken++; }      // (four tiles per visit to the counter: no faster, 1.03-1.04 against 1.01-1.02 ms)
        auto prefetch_rows = [&](u64 tn) {          // rows of this workgroup's next tile into xn / len_raw
            const u64 gn = tn * NW + wave, gc = gn < n_groups ? gn : 0, rn = gc * 64 + ln;
            const v2u* row = reinterpret_cast<const v2u*>(packed) + gc * (32 * WPR) + ln;
            #pragma unroll
            for (int t2 = 0; t2 < WPR / 2; t2++) xn[t2] = __builtin_nontemporal_load(row + t2 * 64);
            len_raw = lens[rn < n_reads ? rn : 0];
            asm volatile("" ::: "memory");
        };
        const u64 g = tile * NW + wave, r = g * 64 + ln;
        const bool live = g < n_groups && r < n_reads;
        u32 w[WPR]; u32 len_cur;
        #pragma unroll
        for (int t2 = 0; t2 < WPR / 2; t2++) {     // register copies free xn[] for the loads below (see k_sieve_q)
            asm volatile("v_mov_b32 %0, %1" : "=v"(w[2 * t2]) : "v"(xn[t2].x));
            asm volatile("v_mov_b32 %0, %1" : "=v"(w[2 * t2 + 1]) : "v"(xn[t2].y));
        }
        asm volatile("v_mov_b32 %0, %1" : "=v"(len_cur) : "v"((u32)len_raw));
        const u32 n = live ? (len_cur & 0x7FFFu) : 0u;
        const int nseeds = n >= MLST_SEED_LEN ? (int)((n - MLST_SEED_LEN) / MLST_SEED_STEP) + 1 : 0;
        // ---- hash every seed, count per (wave, owner); the returned count is the seed's rank inside its run.
        // canonical key = the smaller of the seed and its reverse complement.  Per word the bit-reversed swapped
        // complement once (revc(x) base p = complement of x base 15-p); the reverse complement of the seed at word t
        // is then two funnel shifts (bases 0..3 from word t+1, 4..19 from word t) -- the same 40 bits canon40() returns.
        auto revc = [](u32 x) { return __brev(~(((x >> 1) & 0x55555555u) | ((x + x) & 0xAAAAAAAAu))); };
        u32 hv[NT], rk[NT];                        // the 25 hash bits; owner << 16 | rank inside the (wave, owner) run
        u32 rb_cur = revc(w[0]);
        #pragma unroll
        for (int t = 0; t < NT; t++) {
            const u32 rb_nxt = revc(w[t + 1]);
            const u32 slo = w[t], shi = w[t + 1] & 0xFFu;
            const u32 rlo = __builtin_amdgcn_alignbit(rb_cur, rb_nxt, 24), rhi = rb_cur >> 24;
            const bool rc_less = rhi < shi || (rhi == shi && rlo < slo);
            u32 ow; rt_hash(rc_less ? rlo : slo, rc_less ? rhi : shi, ow, hv[t]);
            rk[t] = ow;
            rb_cur = rb_nxt;
            asm volatile("" : "+v"(hv[t]), "+v"(rk[t]), "+v"(rb_cur));      // one hash after the other: side by side they need more than 64 registers
        }
        // the counting atomics in two batches, one wait per batch (one after the other, each waited for, they were a chain of
        // nine LDS round trips per tile; all nine at once need more than the 64 registers of this kernel)
        {
            constexpr int H = (NT + 1) / 2;
            u32 ra[H];
            #pragma unroll
            for (int t = 0; t < H; t++) { ra[t] = 0; if (t < nseeds) ra[t] = atomicAdd(&s_cnt[wave][rk[t]], 1u); }      // a run holds at most 64 * NT entries
            tie_all<H>(ra);
            #pragma unroll
            for (int t = 0; t < H; t++) rk[t] = t < nseeds ? ((rk[t] << 16) | ra[t]) : 0xFFFFFFFFu;
            u32 rb[NT - H > 0 ? NT - H : 1];
            #pragma unroll
            for (int t = H; t < NT; t++) { rb[t - H] = 0; if (t < nseeds) rb[t - H] = atomicAdd(&s_cnt[wave][rk[t]], 1u); }
            tie_all<(NT - H > 0 ? NT - H : 1)>(rb);
            #pragma unroll
            for (int t = H; t < NT; t++) rk[t] = t < nseeds ? ((rk[t] << 16) | rb[t - H]) : 0xFFFFFFFFu;
        }
        lds_barrier();
        // ---- run lengths and starts.  Thread (sq_, so_) owns the runs of waves 4 sq_ .. 4 sq_ + 3 for owner so_.  An empty run
        // holds one dummy entry (the consumer counts runs).
        u32 c4[4], s4 = 0;
        #pragma unroll
        for (int v = 0; v < 4; v++) { c4[v] = s_cnt[sq_ * 4 + v][so_]; s4 += c4[v] ? c4[v] : 1u; }
        s_part[sq_][so_] = s4;
        lds_barrier();
        u32 pq[QN], tot = 0;
        #pragma unroll
        for (int v = 0; v < QN; v++) { pq[v] = s_part[v][so_]; tot += pq[v]; }
        const u32 inc = wave_incl_scan_dpp(tot);           // every wave scans the 64 owners of its chunk (so_ >> 6)
        if (sq_ == 0 && ln == 63) s_wsum[so_ >> 6] = inc;
        lds_barrier();
        {
            u32 off = inc - tot;
            for (u32 v = 0; v < (so_ >> 6); v++) off += s_wsum[v];
            if (sq_ == 0) {
                s_off[so_] = off;
                if (so_ == RT_OWNERS - 1) s_off[RT_OWNERS] = off + tot;
                const u32 written = (s_cur[so_] - (so_ * P + p) * cap_ch) * CW;
                if (written + s_cn[so_] + tot + CW > R.cap || off + tot > SCAP) s_over = 1;      // (+ CW: the last chunk is padded at the end)
            }
            #pragma unroll
            for (int v = 0; v < QN; v++) off += (u32)v < sq_ ? pq[v] : 0u;
            #pragma unroll
            for (int v = 0; v < 4; v++) {
                s_cnt[sq_ * 4 + v][so_] = off;
                if (c4[v] == 0) { if (off < SCAP) s_sorted[off] = RT_FLAG | RT_DUMMY; off += 1; }      // dummy: the run exists, it holds no seed
                else off += c4[v];
            }
        }
        lds_barrier();
        if (s_over) {      // block-uniform: the tile is not routed, its reads are candidates
            if ((u32)td < TILE / 32) {
                const u64 wi = tile * (TILE / 32) + td, first = wi * 32;
                if (first < n_reads) { const u64 left = n_reads - first; R.flags[wi] = left >= 32 ? 0xFFFFFFFFu : ((1u << left) - 1u); }
            }
            #pragma unroll
            for (int v = 0; v < 4; v++) s_cnt[sq_ * 4 + v][so_] = 0;
            prefetch_rows(t_nxt);
            if (td == 0) s_tile = t_nn;
            lds_barrier();
            if (td == 0) s_over = 0;
            tile = t_nxt; t_nxt = s_tile;
            lds_barrier();
            continue;
        }
        // ---- scatter into (owner, wave) order
        {   // the run starts of all seeds first (independent LDS reads), then the writes; a slot without a seed writes to a
            // spare word behind the buffer
            u32 st[NT];
            #pragma unroll
            for (int t = 0; t < NT; t++) st[t] = s_cnt[wave][(rk[t] >> 16) & 255u];
            tie_all<NT>(st);
            #pragma unroll
            for (int t = 0; t < NT; t++) {
                const bool valid = rk[t] != 0xFFFFFFFFu;
                s_sorted[valid ? st[t] + (rk[t] & 0xFFFFu) : SCAP] = ((rk[t] & 0xFFFFu) == 0 ? RT_FLAG : 0u) | ((u32)ln << 25) | hv[t];
            }
        }
        lds_barrier();
        prefetch_rows(t_nxt);         // requested here, when the seeds' registers are free again; in flight during the write-out
        // ---- append to the regions in whole 64-byte chunks.  An owner's stream = what it carried over from the tiles before
        // (fewer than CW entries, in LDS) followed by this tile's segment; the chunks that are full leave as aligned 16-byte
        // stores, the rest is carried on.  (Round 2 appended every segment as it was, ~150 bytes at a 16-byte boundary: the
        // lines were completed in L2 by the next tile's segment -- or left it half written; 2.41 GB reached HBM for 2.05 GB
        // of entries.)  Wave v serves owners OPW v .. OPW v + OPW - 1, four at a time: 16 lanes per owner, four consecutive
        // entries of the stream per lane and pass.
        {
            constexpr int NI = 2;                 // owners-of-four handled together (OPW / 4 = 4 or 8 in all)
            const u32 sub = (u32)ln & 15u, grp = (u32)ln >> 4;
            #pragma unroll
            for (int i0 = 0; i0 < OPW / 4; i0 += NI) {
                u32 oo[NI], sb[NI], se[NI], sc[NI], cn[NI];
                #pragma unroll
                for (int i = 0; i < NI; i++) {
                    oo[i] = (u32)wave * OPW + (u32)(i0 + i) * 4 + grp;
                    asm volatile("" : "+v"(oo[i]));      // keeps the region addresses out of the loop-invariant (spilled) set
                    sb[i] = s_off[oo[i]]; se[i] = s_off[oo[i] + 1]; sc[i] = s_cur[oo[i]]; cn[i] = s_cn[oo[i]];
                }
                tie_all<NI>(sb); tie_all<NI>(se); tie_all<NI>(sc); tie_all<NI>(cn);
                #pragma unroll
                for (int i = 0; i < NI; i++) {
                    const u32 T = cn[i] + (se[i] - sb[i]), out = T & ~(CW - 1u);
                    auto dst = reinterpret_cast<v4u GLOBAL_AS*>(R.arena.g()) + (u64)sc[i] * (CW / 4);
                    for (u32 base = sub * 4; base < T; base += 64) {      // one pass for streams up to 64 entries, seldom two
                        u32 v[4];
                        #pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const u32 idx = base + (u32)e;
                            const u32* src = idx < cn[i] ? &s_carry[oo[i]][idx] : &s_sorted[sb[i] + idx - cn[i]];
                            v[e] = idx < T ? *src : RT_DUMMY;
                        }
                        tie_all<4>(v);
                        if (base < out) { v4u q4 = {v[0], v[1], v[2], v[3]}; dst[base >> 2] = q4; }      // (base and out are multiples of 4 and 16: whole or not at all)
                        else {
                            #pragma unroll
                            for (int e = 0; e < 4; e++) if (base + (u32)e < T) s_carry[oo[i]][base + (u32)e - out] = v[e];
                        }
                    }
                    if (sub == 0) { s_cur[oo[i]] = sc[i] + out / CW; s_cn[oo[i]] = T - out; }
                }
            }
        }
        #pragma unroll
        for (int v = 0; v < 4; v++) s_cnt[sq_ * 4 + v][so_] = 0;
        if (td == 0) { R.emitted[(u64)p * (R.tiles_max + 1) + 1 + n_emit] = (u32)tile; n_emit++; s_tile = t_nn; }
        lds_barrier();
        tile = t_nxt; t_nxt = s_tile;      // (s_tile is next written behind the first barrier of the following iteration)
    }
    if (tid < RT_OWNERS) {      // the last, partly filled chunk of every region: padded with entries that are neither a seed nor a run start
        const u32 ch = s_cur[tid], cn = s_cn[tid];
        u32 done = (ch - ((u32)tid * P + p) * cap_ch) * CW;
        if (cn) {
            auto dst = reinterpret_cast<v4u GLOBAL_AS*>(R.arena.g()) + (u64)ch * (CW / 4);
            #pragma unroll
            for (int k = 0; k < (int)CW / 4; k++) {
                v4u q4;
                q4.x = (u32)(4 * k) < cn ? s_carry[tid][4 * k] : RT_DUMMY; q4.y = (u32)(4 * k + 1) < cn ? s_carry[tid][4 * k + 1] : RT_DUMMY;
                q4.z = (u32)(4 * k + 2) < cn ? s_carry[tid][4 * k + 2] : RT_DUMMY; q4.w = (u32)(4 * k + 3) < cn ? s_carry[tid][4 * k + 3] : RT_DUMMY;
                dst[k] = q4;
            }
            done += CW;
        }
        R.counts[(u64)tid * P + p] = done;
    }
    if (tid == 0) { R.emitted[(u64)p * (R.tiles_max + 1)] = n_emit; rt_trace_end(R, p); }
}

// examine up to 64 parked survivors of one wave: re-hash the read's seeds, probe the fingerprint sieve with those whose
// hash is the entry's.  Two dependent round trips (the whole row, the buckets), each a batch of independent loads: a lane
// that walked its seeds one load at a time held its wave for a dozen memory latencies per survivor.  The read's length is
// not fetched: rows hold zeros beyond the read, and a window of them that happened to reproduce the entry's 33 hash
// bits AND sat in the fingerprint sieve would add a candidate, which k_seed looks up exactly like every other.
// A parked entry: hash (25 bits) | lane << 25 | producer wave << 31 | position of the tile in the producer's list << 35 |
// producer << 51.  The tile's number is looked up when the entry is examined, not where it passed the filter: a
// dependent global load there stalled the streaming loop in three iterations out of four (0.90 -> 1.00 ms).
__device__ inline u64 rt_park(u32 entry, u32 wv, u32 jt, u32 p) {
    return (u64)(entry & (RT_HMASK | (63u << 25))) | ((u64)wv << 31) | ((u64)jt << 35) | ((u64)p << 51);
}
// Examine one parked entry: the read's row is fetched, its seeds are re-hashed, the one whose 25 hash bits are the
// entry's probes the fingerprint sieve; a hit sets the read's candidate flag.  Three dependent round trips (tile number,
// row, bucket), each a batch of independent loads.  A read with several seeds of that hash becomes a candidate without
// the exact check (k_seed looks every seed of every candidate up exactly; round 2 probed inside the loop over the seeds:
// up to nine dependent bucket walks per round, ~17 us per 64 entries).  The read's length is not fetched: rows hold
// zeros beyond the read, and a window of them that reproduced the hash AND sat in the sieve would add a candidate.
template <int WPR, bool CHECK_FLAG>
__device__ inline void rt_examine_one(u64 e, const u32* __restrict__ packed, u64 n_reads, const RouteDev& R, u32 NWP,
                                      const uint4* __restrict__ sieve, u32 smask, u32 sshift) {
    constexpr int NT = WPR - 1;
    const u32 want = (u32)e & RT_HMASK, ln = ((u32)e >> 25) & 63u, wv = (u32)(e >> 31) & 15u, jt = (u32)(e >> 35) & 0xFFFFu, p = (u32)(e >> 51);
    const u32 tile = R.emitted[(u64)p * (R.tiles_max + 1) + 1 + jt];
    const u64 rr = ((u64)tile * NWP + wv) * 64 + ln;
    if (rr >= n_reads) return;
    // experiments of round 5 (MLST_RT_DEBUG, profiles/round5/sieve.md): 4 = every second entry is dropped unexamined (TIMING ONLY,
    // candidates are lost: what the examination would cost if a check on the entry's 33 hash bits alone removed the filter's false
    // positives first); 8 = the read becomes a candidate without its row being fetched (results unchanged -- k_seed looks every
    // seed of a candidate up exactly -- but k_seed sees every parked read: what leaving the examination out would cost there)
    if ((R.dbg & 4u) && (((u32)e ^ (u32)(e >> 7)) & 1u)) return;
    if (R.dbg & 8u) { atomicOr(&R.flags.p[rr >> 5], 1u << (rr & 31)); return; }
    // a read on a locus arrives here nine times, from nine owners: once its flag is up the other eight need no row (640
    // bytes each).  A stale look (the flag words are written by atomics of other XCDs) only costs the fetch it would have saved.
    if (CHECK_FLAG && ((__hip_atomic_load(&R.flags.p[rr >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (rr & 31)) & 1u)) return;
    const u32* row = packed + packed_index(rr, WPR, 0);      // word c of the row: row[(c >> 1) * 128 + (c & 1)]
    u32 w[WPR];
    #pragma unroll
    for (int c = 0; c < WPR; c++) w[c] = row[(c >> 1) * 128 + (c & 1)];
    tie_all<WPR>(w);
    u32 k0 = 0, k1 = 0, nm = 0;
    #pragma unroll
    for (int t = 0; t < NT; t++) {
        u32 fl; const u64 c = canon40((u64)w[t] | ((u64)(w[t + 1] & 0xFFu) << 32), fl);
        u32 ow, hh; rt_hash((u32)c, (u32)(c >> 32), ow, hh);
        if (hh == want) { if (nm == 0) { k0 = w[t]; k1 = w[t + 1]; } nm++; }
    }
    bool hit = nm > 1;
    if (nm == 1) hit = bin_exact(k0, k1, sieve, smask, sshift);
    if (hit) atomicOr(&R.flags.p[rr >> 5], 1u << (rr & 31));
}
// up to 64 parked entries of one wave leave for k_route_verify (one atomic, one coalesced store); when the list is full
// they are examined here
template <int WPR>
__device__ inline void rt_flush(const u64* q, u32 cnt, int lane, const u32* __restrict__ packed, u64 n_reads, const RouteDev& R, u32 NWP,
                                const uint4* __restrict__ sieve, u32 smask, u32 sshift, Counters* __restrict__ ctr) {
    if (cnt == 0 || (R.dbg & 1u)) return;
    u64 base = 0;
    if (lane == 0) base = atomicAdd(&ctr->rt_parked, (u64)cnt);
    base = uniform_u64(base);
    u64 e = 0;
    if ((u32)lane < cnt) {      // the queue holds entry | (run | region << 20) << 32: packed here, 64 at a time, not where an entry passed the filter
        const u64 raw = q[lane]; const u32 hi = (u32)(raw >> 32), run = hi & 0xFFFFFu, p = hi >> 20;
        e = rt_park((u32)raw, run & (NWP - 1u), NWP == 16 ? run >> 4 : run >> 3, p);
    }
    if (base + cnt <= R.parked_cap) { if ((u32)lane < cnt) R.parked[base + (u32)lane] = e; }
    else if ((u32)lane < cnt) rt_examine_one<WPR, false>(e, packed, n_reads, R, NWP, sieve, smask, sshift);
}
// the entries that passed the LDS filter, examined at full occupancy (inside k_route_probe, whose 147 KB of LDS allow 16
// waves per CU, the three round trips of an examination stalled the streaming waves: 0.5 of 1.2 ms at 7.7 M entries)
template <int WPR>
__global__ __launch_bounds__(256) void k_route_verify(const u32* __restrict__ packed, u64 n_reads, const uint4* __restrict__ sieve, u32 smask, const RouteDev R,
                                                      Counters* __restrict__ ctr) {
    const u32 sshift = (u32)__clz((int)smask);
    u64 n = ctr->rt_parked; n = n < R.parked_cap ? n : R.parked_cap;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        rt_examine_one<WPR, true>(R.parked[i], packed, n_reads, R, R.nw, sieve, smask, sshift);
}
template <int WPR, int PF>
__global__ __launch_bounds__(1024) void k_route_probe(const u32* __restrict__ packed, u64 n_reads,
                                                      const uint4* __restrict__ sieve, u32 smask, const RouteDev R, Counters* __restrict__ ctr) {
    __shared__ __attribute__((aligned(16))) u32 s_f[RT_FWORDS];
    __shared__ u64 s_q[16][128];                  // per-wave queue of entries that passed the filter: read | hash << 32
    // PF = 16-byte loads per lane in flight: 2 (2 KiB per wave, 32 KiB per CU) or 4
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 owner = blockIdx.x, P = R.n_prod;
    const u32 sshift = (u32)__clz((int)smask);
    if (tid == 0) rt_trace_begin(R, P + owner);
    {
        const v4u* g4 = reinterpret_cast<const v4u*>(R.filter.p + (u64)owner * RT_FWORDS); v4u* s4 = reinterpret_cast<v4u*>(s_f);
        #pragma unroll
        for (int j = 0; j < RT_FWORDS / 4 / 1024; j++) s4[tid + 1024 * j] = g4[tid + 1024 * j];
    }
    __syncthreads();
    const u32 NWP = R.nw;                          // runs per tile in a region (one per producer wave)
    u64* const q = s_q[wave]; u32 qn = 0, n_pass = 0;      // wave-uniform: parked entries; entries that passed the filter so far
    const u64 lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    const v4u none4 = {RT_DUMMY, RT_DUMMY, RT_DUMMY, RT_DUMMY};      // lanes beyond a region's end: neither a seed nor a run start
    // the entry counts of this wave's regions (lane i: region wave + 16 i), fetched once: read region by region they cost a
    // dependent round trip in front of every region's first load
    u32 my_n[RT_MAXP / 1024];
    #pragma unroll
    for (int c = 0; c < RT_MAXP / 1024; c++) {
        const u32 pp = (u32)wave + 16u * ((u32)lane + 64u * c);
        my_n[c] = pp < P ? R.counts[(u64)owner * P + pp] : 0u;
    }
    for (u32 p = (u32)wave, pi = 0; p < P; p += 16, pi++) {      // this wave's regions
        u32 n = 0;
        #pragma unroll
        for (int c = 0; c < RT_MAXP / 1024; c++) if ((pi >> 6) == (u32)c) n = (u32)__shfl((int)my_n[c], (int)(pi & 63));
        n = (u32)__builtin_amdgcn_readfirstlane((int)n);
        const auto ent4 = reinterpret_cast<const v4u GLOBAL_AS*>(R.arena.g() + ((u64)owner * P + p) * R.cap);
        int seq = -1;                               // flags seen so far - 1 = sequence number of the current (tile, wave) run
        const u32 pshift = p << 20;                 // a queued entry: entry | (run | region << 20) << 32 (runs per region < 2^20: tiles_max <= 0xFFFF, 16 runs each)
        v4u en[PF];
        #pragma unroll
        for (int u = 0; u < PF; u++) { const u32 i = (u32)u * 256 + (u32)lane * 4; en[u] = i < n ? __builtin_nontemporal_load(ent4 + (i >> 2)) : none4; }
        for (u32 i0 = 0; i0 < n; i0 += 256 * PF) {
            u32 ev[PF][4];
            #pragma unroll
            for (int u = 0; u < PF; u++) {         // register copies free en[] for the loads below (see k_sieve_q)
                asm volatile("v_mov_b32 %0, %1" : "=v"(ev[u][0]) : "v"(en[u].x)); asm volatile("v_mov_b32 %0, %1" : "=v"(ev[u][1]) : "v"(en[u].y));
                asm volatile("v_mov_b32 %0, %1" : "=v"(ev[u][2]) : "v"(en[u].z)); asm volatile("v_mov_b32 %0, %1" : "=v"(ev[u][3]) : "v"(en[u].w));
            }
            #pragma unroll
            for (int u = 0; u < PF; u++) { const u32 i = i0 + 256 * PF + (u32)u * 256 + (u32)lane * 4; en[u] = i < n ? __builtin_nontemporal_load(ent4 + (i >> 2)) : none4; }
            asm volatile("" ::: "memory");
            #pragma unroll
            for (int u = 0; u < PF; u++) {
                const u32 ib = i0 + (u32)u * 256;
                if (ib >= n) break;                 // wave-uniform
                // (a region's length is a multiple of four: a lane's four entries are all inside it or all the no-seed filler)
                bool fg[4]; u64 B[4]; u32 before = 0;
                #pragma unroll
                for (int j = 0; j < 4; j++) {
                    fg[j] = (ev[u][j] & RT_FLAG) != 0;
                    B[j] = __ballot(fg[j]);
                    before = __builtin_amdgcn_mbcnt_hi((u32)(B[j] >> 32), __builtin_amdgcn_mbcnt_lo((u32)B[j], before));      // run starts in the lanes below
                }
                // the filter words of the four entries in one batch of independent LDS reads
                u32 f0[4], f1[4], k0[4], k1[4];
                #pragma unroll
                for (int j = 0; j < 4; j++) {
                    u32 b0, b1; rt_filter_addr(ev[u][j] & RT_HMASK, b0, k0[j], b1, k1[j]);
                    f0[j] = s_f[b0]; f1[j] = s_f[b1];      // (an entry without a seed reads the words of the value RT_DUMMY and fails below)
                }
                tie_all<4>(f0); tie_all<4>(f1);
                const u32 run0 = (u32)(seq + (int)before);      // the run of entry j: run0 + run starts among the lane's entries 0 .. j (worked out only where an entry passes)
                #pragma unroll
                for (int j = 0; j < 4; j++) {
                    // missing filter bits | "no seed" (RT_DUMMY is all ones: + 1 carries into bit 25), without a branch
                    const u32 miss = ((f0[j] & k0[j]) ^ k0[j]) | ((f1[j] & k1[j]) ^ k1[j]) | (((ev[u][j] & RT_HMASK) + 1u) >> 25);
                    const bool pass = miss == 0u;
                    const u64 pm = __ballot(pass);
                    if (pm) {
                        if (pass) {
                            u32 run = run0;
                            #pragma unroll
                            for (int jj = 0; jj <= j; jj++) run += ev[u][jj] >> 31;
                            q[qn + (u32)__popcll(pm & lt)] = (u64)ev[u][j] | ((u64)(run | pshift) << 32);      // raw: rt_flush makes the parked form of it
                        }
                        qn += (u32)__popcll(pm);
                        if (qn >= 64) {
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            rt_flush<WPR>(q + (qn - 64), 64, lane, packed, n_reads, R, NWP, sieve, smask, sshift, ctr);
                            qn -= 64; n_pass += 64;
                        }
                    }
                }
                seq += (int)(__popcll(B[0]) + __popcll(B[1]) + __popcll(B[2]) + __popcll(B[3]));
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    rt_flush<WPR>(q, qn, lane, packed, n_reads, R, NWP, sieve, smask, sshift, ctr);
    n_pass += qn;
    if (lane == 0 && n_pass) atomicAdd(&ctr->cnt[MLST_CNT_SIEVE_PASS], (u64)n_pass);
    __syncthreads();
    if (tid == 0) { atomicMax(&ctr->sv_t1, (u64)wall_clock64()); rt_trace_end(R, P + owner); }
}

// candidate flags -> candidate list (one atomic per 1024-thread workgroup that holds candidates)
#define FLAG_U 8      /* flag words per thread and turn of k_flag_compact */
// between the slices of a sub-batched routed sieve (MLST_RT_SLICE): the producers' tile counter and the parked list start over
__global__ void k_rt_reset(Counters* ctr) { ctr->rt_next = 0; ctr->rt_parked = 0; }
__global__ __launch_bounds__(1024) void k_flag_compact(u32* __restrict__ flags, u64 n_reads, u32* __restrict__ cand, Counters* __restrict__ ctr, int paired) {
    __shared__ u32 s_cnt[16]; __shared__ u64 s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u64 n_words = (n_reads + 31) >> 5;
    if (blockIdx.x == 0 && tid == 0) { ctr->rt_next = 0; ctr->rt_parked = 0; }      // the producers' tile counter and the parked list, for the next submission
    // FLAG_U words per thread, loaded together: one scan and one returning atomic per 8,192 words (a word per thread and turn
    // was six turns per workgroup on cfg3, each a load, two barriers and an atomic one after the other: 22 us)
    for (u64 w0 = (u64)blockIdx.x * (1024 * FLAG_U); w0 < n_words; w0 += (u64)gridDim.x * (1024 * FLAG_U)) {
        u32 f[FLAG_U]; u32 c = 0;
        #pragma unroll
        for (int k = 0; k < FLAG_U; k++) {      // (a word that held flags is left zero for the next submission: no fill of the 6 MB in between)
            const u64 wi = w0 + (u64)k * 1024 + tid; f[k] = wi < n_words ? flags[wi] : 0u;
            if (f[k]) flags[wi] = 0u;
        }
        #pragma unroll
        for (int k = 0; k < FLAG_U; k++) {
            if (paired) {      // mates become candidates together (see sv_emit); the last word is clipped to the reads that exist
                const u64 first = (w0 + (u64)k * 1024 + tid) * 32;
                f[k] |= ((f[k] & 0xAAAAAAAAu) >> 1) | ((f[k] & 0x55555555u) << 1);
                if (first + 32 > n_reads) f[k] &= first < n_reads ? ((1u << (n_reads - first)) - 1u) : 0u;
            }
            c += (u32)__popc(f[k]);
        }
        u32 tot;
        const u32 pre = wave_excl_scan_u32(c, tot);
        if (lane == 0) s_cnt[wv] = tot;
        __syncthreads();
        u32 before = 0, all = 0;
        for (int k = 0; k < 16; k++) { if (k < wv) before += s_cnt[k]; all += s_cnt[k]; }
        if (all) {
            if (tid == 0) s_base = atomicAdd(&ctr->n_cand, (u64)all);
            __syncthreads();
            u64 at = s_base + before + pre;
            #pragma unroll
            for (int k = 0; k < FLAG_U; k++) {
                u32 m = f[k]; const u64 r0 = (w0 + (u64)k * 1024 + tid) * 32;
                while (m) { int b = __ffs(m) - 1; m &= m - 1; cand[at++] = (u32)(r0 + b); }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ BGZF -> text (one wave per <= 64 KiB deflate block)
typedef inflate_lane::Blk BgzfBlk;      // { u64 in_off, out_off; u32 in_len, out_len; }
#if !defined(MLST_INFLATE_GROUP)
#define MLST_INFLATE_GROUP 64
#endif
#define INFLATE_NG (64 / MLST_INFLATE_GROUP)      /* streams per wave (csrc/inflate_wave.h) */
// one wave per block (csrc/inflate_wave.h); comp_bytes = size of the compressed buffer (the input windows stop there)
__global__ __launch_bounds__(64) void k_inflate(const u8* __restrict__ comp, u64 comp_bytes, const BgzfBlk* __restrict__ blk, u32 n_blk, u8* __restrict__ out,
                                                u32* __restrict__ err /* [0] = 1 + first bad block, [1] = its code */, unsigned long long* __restrict__ stats /* optional: 12 sums */,
                                                const u32* __restrict__ only /* optional: decode block i only where only[i] == TOK_OVERFLOW (what the two-kernel path left over) */,
                                                u32 err_base /* number of blk[0] in the caller's list (for err[0]) */) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int GS = inflate_wave::GS, NG = 64 / GS;      // lanes per stream, streams per wave
    __shared__ inflate_wave::Tabs s_tb[NG];        // Huffman tables, output ring and match queue of every stream
    const int lane = (int)threadIdx.x % GS, grp = (int)threadIdx.x / GS, gbase = grp * GS;
    for (u32 i = blockIdx.x * NG + (u32)grp; i < n_blk; i += gridDim.x * NG) {     // the same for the lanes of a group
        if (only && only[i] != (u32)inflate_lane::TOK_OVERFLOW) continue;
        const BgzfBlk B = blk[i];
        u32 produced = 0;
        inflate_wave::Stats st;
        int rc = inflate_wave::inflate_stream(comp + B.in_off, (u64)B.in_len, comp + comp_bytes, out + B.out_off, B.out_len, s_tb[grp], lane, gbase, &produced, &st);
#if defined(MLST_INFLATE_STATS)
        if (stats && lane == 0) {
            const unsigned long long v[10] = {st.lookups, st.lits, st.near_, st.far_def, st.far_sync, st.far_flush, st.fences, st.builds, st.t_build, st.t_codes};
            for (int k = 0; k < 10; k++) atomicAdd(&stats[k], v[k]);
        }
#endif
        if (rc == mlst_inflate::OK && produced != B.out_len) rc = mlst_inflate::E_SHORT;
        if (rc != mlst_inflate::OK && lane == 0 && atomicCAS(&err[0], 0u, err_base + i + 1u) == 0u) err[1] = (u32)(-rc);
        inflate_wave::wave_sync();                // the tables are rebuilt for the next stream
    }
#endif
}
// the two-kernel inflate (csrc/inflate_lane.h): lane = block -> tokens; workgroup = block -> bytes by pointer jumping in LDS
#define INFL_TOK_CAP 24576u          /* tokens a block may have (96 KB of the token buffer per block); more: left to k_inflate */
__global__ __launch_bounds__(64) void k_inflate_tok(const u8* __restrict__ comp, u64 comp_bytes, const BgzfBlk* __restrict__ blk, u32 n_blk, u32 err_base, u32* __restrict__ tok,
                                                    u32* __restrict__ n_tok, u32* __restrict__ err) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ mlst_inflate::Tables s_tb[64];
    __shared__ unsigned long s_win[(mlst_inflate::Bits::WIN / 8) * 64];       // WIN bytes of every lane's stream, lane-interleaved 8-byte words (inflate_dev.h: Bits::win)
    inflate_lane::tok_body(comp, comp + comp_bytes, blk, n_blk, err_base, tok, INFL_TOK_CAP, n_tok, err, s_tb, s_win);
#endif
}
// phase 1 with 576 bytes of state per stream (csrc/inflate_canon.h): four waves per CU, one per SIMD, instead of one
__global__ __launch_bounds__(64) void k_inflate_tok2(const u8* __restrict__ comp, u64 comp_bytes, const BgzfBlk* __restrict__ blk, u32 n_blk, u32 err_base, u32* __restrict__ tok,
                                                     u32* __restrict__ n_tok, u32* __restrict__ err) {
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace inflate_canon;
    __shared__ u32 s_w[W_TOTAL * 64];
    const u32 lane = threadIdx.x & 63u;
    for (u32 i0 = blockIdx.x * 64u; i0 < n_blk; i0 += gridDim.x * 64u) {
        const u32 i = i0 + lane;
        if (i >= n_blk) continue;
        const BgzfBlk B = blk[i];
        const u32 want = B.out_len;
        if (want > inflate_lane::LIT_BASE) { n_tok[i] = (u32)inflate_lane::TOK_OVERFLOW; continue; }
        MemLds m; m.base = (__attribute__((address_space(3))) u32*)s_w + lane;
        Bits<SrcLds> b; b.src.win = (__attribute__((address_space(3))) u32*)s_w + W_WIN * 64u + lane; b.src.in = comp + B.in_off; b.src.buf_end = comp + comp_bytes;
        b.src.win_at = 0; b.src.have = false; b.buf = 0; b.cnt = 0; b.pos = 0; b.n = B.in_len;
        Tok o; o.tok = tok + (u64)i * INFL_TOK_CAP; o.nt = 0; o.cap = INFL_TOK_CAP; o.over = false;
        u32 produced = 0;
        int rc = tok_stream(m, b, o, want, &produced);
        if (rc == mlst_inflate::OK && !o.over && produced != want) rc = mlst_inflate::E_SHORT;
        if (rc != mlst_inflate::OK) { if (atomicCAS(&err[0], 0u, err_base + i + 1u) == 0u) err[1] = (u32)(-rc); n_tok[i] = 0; }
        else n_tok[i] = o.over ? (u32)inflate_lane::TOK_OVERFLOW : o.nt;
    }
#endif
}
// nl (optional): newlines per 2^nl_shift bytes of the text buffer (cell = offset in `out` >> nl_shift; nl_shift >= 12), added up as the text is written
__global__ __launch_bounds__(1024) void k_inflate_ptr(const u8* __restrict__ comp, const BgzfBlk* __restrict__ blk, u32 n_blk, u32 err_base, const u32* __restrict__ tok,
                                                      const u32* __restrict__ n_tok, u8* __restrict__ out, u32* __restrict__ err, u32* __restrict__ nl, u32 nl_shift) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) u16 s_ptr[65536];
    __shared__ u32 s_part[16]; __shared__ u32 s_nl[20];
    inflate_lane::ptr_body<1024>(comp, blk, n_blk, err_base, tok, INFL_TOK_CAP, n_tok, out, err, s_ptr, s_part, s_nl, nl, nl_shift);
#endif
}
// ------------------------------------------------------------------ FASTQ text -> packed reads (GPU parser)
#define FQ_BLOCK 4096        // bytes of text per workgroup
// newline flags of 16 bytes of text at byte p (a multiple of 16; the text buffer is 256-byte aligned): bit k = byte p + k is '\n'.
// One 16-byte load per thread instead of sixteen byte loads (the passes over the text were bound by their load instructions:
// 330-540 GB/s; the parse of a bgzip'd piece took 7 ms beside 9.5 ms of inflate).
__device__ inline u32 fq_nl16(const u8* __restrict__ text, u64 p, u64 n_bytes) {
    u32 m = 0;
    if (p + 16 <= n_bytes) {
        const uint4 v = *reinterpret_cast<const uint4*>(text + p);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
        #pragma unroll
        for (int d = 0; d < 4; d++) {
            const u32 y = w[d] ^ 0x0A0A0A0Au;                                                  // zero bytes where the text holds '\n'
            const u32 t = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);              // 0x80 in exactly those bytes
            m |= (((t >> 7) & 1u) | ((t >> 14) & 2u) | ((t >> 21) & 4u) | ((t >> 28) & 8u)) << (4 * d);
        }
    } else {
        for (u32 k = 0; k < 16 && p + k < n_bytes; k++) if (text[p + k] == '\n') m |= 1u << k;
    }
    return m;
}
// pass A: newlines per FQ_BLOCK bytes
__global__ __launch_bounds__(256) void k_fq_count(const u8* __restrict__ text, u64 n_bytes, u32* __restrict__ blk_count) {
    __shared__ u32 s_c[4];
    const u64 p = (u64)blockIdx.x * FQ_BLOCK + (u64)threadIdx.x * 16;
    u32 c = p < n_bytes ? (u32)__popc(fq_nl16(text, p, n_bytes)) : 0u;
    c = wave_sum_u32(c);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_count[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}
// pass B: exclusive scan of the block counts, in place (one workgroup; 4 GB of text = 1 M blocks).  4,096 counts per turn:
// four per thread, a wave scan, the waves' totals through LDS.  (Until round 5 every thread walked a range of its own:
// 64 cache lines per load instruction, 2.1 ms for the 262 k blocks of a 1 GB piece.)
__global__ __launch_bounds__(1024) void k_fq_scan(u32* __restrict__ blk_count, u32 n_blocks, u64* __restrict__ n_lines_out, u64 n_bytes, const u8* __restrict__ text, int count_partial) {
    __shared__ u32 s_w[16]; __shared__ u64 s_carry;
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (u32 i0 = 0; i0 < n_blocks; i0 += 4096) {
        const u32 i = i0 + tid * 4;
        u32 v[4];
        #pragma unroll
        for (int k = 0; k < 4; k++) v[k] = i + (u32)k < n_blocks ? blk_count[i + k] : 0u;
        const u32 sum = v[0] + v[1] + v[2] + v[3], inc = wave_incl_scan_dpp(sum);
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        u32 before = 0, tile = 0;
        for (u32 w = 0; w < 16; w++) { const u32 x = s_w[w]; if (w < wv) before += x; tile += x; }
        u64 run = s_carry + before + inc - sum;
        #pragma unroll
        for (int k = 0; k < 4; k++) { if (i + (u32)k < n_blocks) blk_count[i + k] = (u32)run; run += v[k]; }      // < 2^32 lines per chunk (checked on the host)
        __syncthreads();
        if (tid == 0) s_carry += tile;
        __syncthreads();
    }
    // a last line without a trailing newline still counts as a line (unless more text follows in the next chunk)
    if (tid == 0) *n_lines_out = s_carry + ((count_partial && n_bytes > 0 && text[n_bytes - 1] != '\n') ? 1 : 0);
}
// newlines of the blocks the two-kernel inflate left to k_inflate (only[b] == TOK_OVERFLOW), added to the cells k_inflate_ptr counts into
__global__ __launch_bounds__(256) void k_nl_blocks(const BgzfBlk* __restrict__ blk, u32 n_blk, const u32* __restrict__ only, const u8* __restrict__ out, u32* __restrict__ nl, u32 nl_shift) {
    for (u32 b = blockIdx.x; b < n_blk; b += gridDim.x) {
        if (only[b] != (u32)inflate_lane::TOK_OVERFLOW) continue;
        const BgzfBlk B = blk[b];
        for (u32 p = threadIdx.x; p < B.out_len; p += 256) if (out[B.out_off + p] == (u8)'\n') atomicAdd(&nl[(B.out_off + p) >> nl_shift], 1u);
    }
}
// pass C: start offset of every line: line 0 starts at 0, line k+1 starts after the k-th newline.  A thread takes 16 bytes
// (FQ_BLOCK = 256 threads x 16), the newline counts are scanned over the workgroup once.
// first: where line 0 starts (the bytes in front of it are filler without a newline: a chunk whose text was placed behind a partial record of unknown length)
__global__ __launch_bounds__(256) void k_fq_lines(const u8* __restrict__ text, u64 n_bytes, const u32* __restrict__ blk_excl, u64* __restrict__ line_start, u64 first) {
    __shared__ u32 s_w[4];
    const u64 p = (u64)blockIdx.x * FQ_BLOCK + (u64)threadIdx.x * 16;
    const int wv = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0) line_start[0] = first;
    u32 m = p < n_bytes ? fq_nl16(text, p, n_bytes) : 0u;
    const u32 cnt = (u32)__popc(m), inc = wave_incl_scan_dpp(cnt);
    if ((threadIdx.x & 63) == 63) s_w[wv] = inc;
    __syncthreads();
    u64 line = (u64)blk_excl[blockIdx.x] + (inc - cnt) + 1;
    for (int w = 0; w < wv; w++) line += s_w[w];
    while (m) { const int k = __ffs((int)m) - 1; m &= m - 1; line_start[line++] = p + (u64)k + 1; }
}
// pass D: per record the byte ranges of the sequence line (4r+1) and the quality line (4r+3); CR stripped
// pair_k != 0: the text is two files of mates one after the other, pair_k records each; record j of the first becomes read
// 2j, record j of the second read 2j + 1 (mates side by side, the order a paired submission expects)
__global__ __launch_bounds__(256) void k_fq_records(const u8* __restrict__ text, u64 n_bytes, const u64* __restrict__ line_start, u64 n_lines, u64 n_reads,
                                                     u64* __restrict__ seq_off, u64* __restrict__ qual_off, u16* __restrict__ lens, u32* __restrict__ flags /* [0]=max len, [1]=errors */,
                                                     u64 pair_k) {
    u32 mx = 0, err = 0;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (u64)gridDim.x * blockDim.x) {
        u64 s0 = line_start[4 * r + 1], s1 = line_start[4 * r + 2], q0 = line_start[4 * r + 3];
        u64 se = s1 - 1;
        u64 qe = (4 * r + 4 < n_lines) ? line_start[4 * r + 4] - 1 : n_bytes;      // last record: up to the end of the chunk ...
        if (qe > q0 && text[qe - 1] == '\n') qe--;                                 // ... minus its newline when there is one
        if (se > s0 && text[se - 1] == '\r') se--;
        if (qe > q0 && text[qe - 1] == '\r') qe--;
        u64 ls = se - s0, lq = qe - q0;
        if (text[line_start[4 * r]] != '@' || ls != lq) err = 1;
        if (ls > MLST_MAX_READ_LEN) { err |= 2; ls = MLST_MAX_READ_LEN; }
        const u64 o = pair_k ? (r < pair_k ? 2 * r : 2 * (r - pair_k) + 1) : r;
        seq_off[o] = s0; qual_off[o] = q0; lens[o] = (u16)ls;
        if ((u32)ls > mx) mx = (u32)ls;
    }
    if (mx) atomicMax(&flags[0], mx);
    if (err) atomicOr(&flags[1], err);
}
// pack from text: same output format as k_pack, reads addressed by separate sequence / quality offsets; lens holds the
// plain lengths on entry (k_fq_records)
__global__ __launch_bounds__(256) void k_pack_text(const u8* __restrict__ text, const u64* __restrict__ seq_off, const u64* __restrict__ qual_off,
                                                    u16* __restrict__ lens, u64 n_reads, u32* __restrict__ packed, u8* __restrict__ qrows, u32 wpr, u32 qstride) {
    __shared__ u32 s_words[64 * RW]; __shared__ u32 s_anyn[2]; __shared__ u16 s_len[64];
    const u64 n_groups = (n_reads + 63) >> 6;
    for (u64 grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        if (threadIdx.x < 64) { const u64 r = grp * 64 + threadIdx.x; s_len[threadIdx.x] = r < n_reads ? (u16)(lens[r] & 0x7FFFu) : (u16)0; }
        __syncthreads();
        pack_group(text, text, seq_off, qual_off, [&](u64 r) { return (u32)s_len[r & 63]; }, n_reads, grp, packed, qrows, lens, wpr, qstride, s_words, s_anyn);
        __syncthreads();
    }
}

// ------------------------------------------------------------------ K2: exact seeds -> work items
struct Bin { u32 locus; int diag; u16 strand, votes; };

// Linear probing, four slots per round trip: a wave walks for its worst lane (about a dozen slots among the 640 look-ups of
// a wave at load 0.5).  Variant builds cut after each stage put k_seed's 132 us on cfg3 at ~15 rows / 12 home slots / 43 values
// and walks / 4 votes / ~55 item emission: every stage is a level of ~2.5 M scattered requests, not a latency chain.
__device__ inline bool table_find(const EngineDev& E, u32 lo, u32 hi, u32& val) {
    const u64 key = (u64)lo | ((u64)hi << 32);
    u32 h = table_hash(lo, hi) & E.table_mask;
    for (u64 step = 0; step <= (u64)E.table_mask; step += 4) {
        u64 k[4];
        #pragma unroll
        for (int j = 0; j < 4; j++) k[j] = E.keys[(h + (u32)j) & E.table_mask];
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            if (k[j] == key) { val = E.vals[(h + (u32)j) & E.table_mask]; return true; }
            if (k[j] == KEY_EMPTY) return false;
        }
        h = (h + 4) & E.table_mask;
    }
    return false;
}

__global__ __launch_bounds__(256) void k_seed(const EngineDev* __restrict__ Ep, const u32* __restrict__ packed, const u8* __restrict__ qrows,
                                               const u16* __restrict__ lens, u32 wpr, u32 qstride, u64 read_base,
                                               const u32* __restrict__ cand, int paired, int qcompact) {
    const EngineDev& E = *Ep;     // device-resident descriptor: fields are scalar-loaded on demand
    // a lane's eight 12-byte bins = 24 words: left at that stride the 64 lanes share 4 of the 32 LDS banks (16-way conflicts
    // on every access of the vote loops); one word of padding per lane makes the stride odd
    // (the per-(locus, strand) items are compacted in place over the bins: slot ni <= k is free by the time bin k is read)
    __shared__ u32 s_bins[256][MLST_MAX_CAND * 3 + 1];
    __shared__ u32 s_tot[4][3]; __shared__ u64 s_base[3];
    const int tid = threadIdx.x, lane = tid & 63;
    u64 n_cand = E.ctr->n_cand;
    for (u64 c0 = (u64)blockIdx.x * 256; c0 < n_cand; c0 += (u64)gridDim.x * 256) {
        u64 c = c0 + tid;
        u32 r = 0, lw = 0, n = 0; int no = 0;
        Bin* items = reinterpret_cast<Bin*>(s_bins[tid]);
        if (c < n_cand) {
            r = cand[c];
            lw = lens[r]; n = lw & 0x7FFFu; bool has_n = (lw & 0x8000u) != 0;
            const u32* row = packed + packed_index(r, wpr, 0);      // word c of the row: row[(c >> 1) * 128 + (c & 1)]
            const u8* qrow = qrows + (qcompact ? c : (u64)r) * qstride;      // qcompact: row k belongs to candidate k
            Bin* bins = reinterpret_cast<Bin*>(s_bins[tid]); int nb = 0;
            int nseeds = n >= MLST_SEED_LEN ? (int)((n - MLST_SEED_LEN) / MLST_SEED_STEP) + 1 : 0;
            // The lane's work is a chain of dependent look-ups (key -> value -> postings); a kernel over ~10^4
            // candidates is as long as one lane's chain.  So the look-ups of up to SEED_CHUNK seeds are issued
            // together: all keys, then all values; only probe chains longer than one slot are walked serially.
            constexpr int SEED_CHUNK = 10;
            for (int t0 = 0; t0 < nseeds; t0 += SEED_CHUNK) {
                u32 wv[SEED_CHUNK + 1];
                #pragma unroll
                for (int u = 0; u <= SEED_CHUNK; u++) { wv[u] = 0; const u32 c = (u32)(t0 + u); if (c < wpr) wv[u] = row[(c >> 1) * 128 + (c & 1)]; }
                tie_all<SEED_CHUNK + 1>(wv);
                u32 klo[SEED_CHUNK], khi[SEED_CHUNK], slot[SEED_CHUNK], sfl[SEED_CHUNK], k0[SEED_CHUNK], k1[SEED_CHUNK]; u32 okm = 0;
                #pragma unroll
                for (int u = 0; u < SEED_CHUNK; u++) {
                    int t = t0 + u; bool ok = t < nseeds;
                    klo[u] = khi[u] = slot[u] = sfl[u] = 0; k0[u] = k1[u] = 0xFFFFFFFFu;
                    if (ok && has_n) { int o = t * MLST_SEED_STEP; bool bad = false; for (int k = 0; k < MLST_SEED_LEN; k++) bad |= (qrow[o + k] & 0x80) != 0; ok = !bad; }
                    if (ok) {
                        u32 sflag; u64 ck = canon40((u64)wv[u] | ((u64)(wv[u + 1] & 0xFFu) << 32), sflag);
                        klo[u] = (u32)ck; khi[u] = (u32)(ck >> 32); sfl[u] = sflag;
                        slot[u] = table_hash(klo[u], khi[u]) & E.table_mask;
                        u64 kk = E.keys[slot[u]]; k0[u] = (u32)kk; k1[u] = (u32)(kk >> 32);
                        okm |= 1u << u;
                    }
                }
                tie_all<SEED_CHUNK>(k0); tie_all<SEED_CHUNK>(k1);
                u32 vals[SEED_CHUNK]; u32 found = 0;
                #pragma unroll
                for (int u = 0; u < SEED_CHUNK; u++) {
                    vals[u] = 0;
                    if (!((okm >> u) & 1u)) continue;
                    if (k0[u] == klo[u] && k1[u] == khi[u]) { vals[u] = E.vals[slot[u]]; found |= 1u << u; }
                    else if (!(k0[u] == 0xFFFFFFFFu && k1[u] == 0xFFFFFFFFu)) {      // occupied by another key: walk the chain
                        u32 v; if (table_find(E, klo[u], khi[u], v)) { vals[u] = v; found |= 1u << u; }
                    }
                }
                tie_all<SEED_CHUNK>(vals);
                #pragma unroll
                for (int u = 0; u < SEED_CHUNK; u++) {
                    if (!((found >> u) & 1u)) continue;
                    const int o = (t0 + u) * MLST_SEED_STEP;
                    const u32 val = vals[u], sflag = sfl[u];
                    u32 pstart, pcount; u32 single = 0;
                    if (val & 0x80000000u) { single = val & 0x7FFFFFFFu; pstart = 0; pcount = 1; }
                    else { pstart = val >> 5; pcount = val & 31u; }
                    // postings are stored sorted by (locus, flag, pos); visit them in (locus, strand, pos) order with
                    // strand = flag XOR sflag -- the order the non-canonical index of the oracle has
                    u32 last_key = 0; bool first = true;
                    for (u32 k = 0; k < pcount; k++) {
                        u32 post;
                        if (val & 0x80000000u) post = single ^ (sflag << 12);
                        else if (!sflag) post = E.posts[pstart + k];
                        else {      // k-th smallest of the flag-flipped postings (pcount <= 16)
                            u32 bestp = 0xFFFFFFFFu;
                            for (u32 pp = 0; pp < pcount; pp++) { u32 x = E.posts[pstart + pp] ^ (1u << 12); if ((first || x > last_key) && x < bestp) bestp = x; }
                            post = bestp; last_key = bestp; first = false;
                        }
                        u32 locus = post >> 13, strand = (post >> 12) & 1; int pos = (int)(post & 0xFFFu);
                        int diag = strand ? pos + MLST_SEED_LEN + o - (int)n : pos - o;
                        int kk; for (kk = 0; kk < nb; kk++) if (bins[kk].locus == locus && bins[kk].strand == strand && bins[kk].diag == diag) break;
                        if (kk < nb) bins[kk].votes++;
                        else if (nb < MLST_MAX_CAND) { bins[nb].locus = locus; bins[nb].strand = (u16)strand; bins[nb].diag = diag; bins[nb].votes = 1; nb++; }
                    }
                }
            }
            // one item per (locus, strand): most votes, then the smaller diagonal; first-seen order
            int ni = 0;
            for (int k = 0; k < nb; k++) {
                int u; for (u = 0; u < ni; u++) if (items[u].locus == bins[k].locus && items[u].strand == bins[k].strand) break;
                if (u == ni) items[ni++] = bins[k];
                else if (bins[k].votes > items[u].votes || (bins[k].votes == items[u].votes && bins[k].diag < items[u].diag)) items[u] = bins[k];
            }
            for (int u = 0; u < ni; u++) if (items[u].votes >= MLST_MIN_VOTES) items[no++] = items[u];
        }
        // retain the read.  The three counters (retained reads, items, result rows) are bumped once per WORKGROUP and
        // turn: a contended word serves only ~88 returning atomics per microsecond, and once per wave they were 3 x 4,096
        // of them -- most of the ~55 us this part of the kernel took on cfg3.  Two rounds: a read that finds no room
        // in the retained arena must not reserve item slots (nothing would fill them).
        u64 keep = __ballot(no > 0);
        if (lane == 0) s_tot[tid >> 6][0] = (u32)__popcll(keep);
        __syncthreads();
        if (tid == 0) {
            const u32 t0 = s_tot[0][0] + s_tot[1][0] + s_tot[2][0] + s_tot[3][0];
            s_base[0] = t0 ? atomicAdd(&E.ctr->n_ret, (u64)t0) : 0ull;
        }
        __syncthreads();
        u64 slot = ~0ull;
        if (no > 0) {
            u64 ret0 = s_base[0];
            for (int w = 0; w < (tid >> 6); w++) ret0 += s_tot[w][0];
            slot = ret0 + __popcll(keep & ((1ull << lane) - 1));
            if (slot >= E.cap_ret) { atomicOr(&E.ctr->err, 1ull); slot = ~0ull; no = 0; }
        }
        u32 my_res = 0;
        for (int u = 0; u < no; u++) my_res += E.loci[items[u].locus].n_pad;
        u32 tot_items, tot_res;
        const u32 pre_items = wave_excl_scan_u32((u32)no, tot_items);
        const u32 pre_res = wave_excl_scan_u32(my_res, tot_res);
        if (lane == 0) { s_tot[tid >> 6][1] = tot_items; s_tot[tid >> 6][2] = tot_res; }
        __syncthreads();
        if (tid == 0) {
            u32 t1 = 0, t2 = 0;
            for (int w = 0; w < 4; w++) { t1 += s_tot[w][1]; t2 += s_tot[w][2]; }
            u64 b1 = 0, b2 = 0;
            if (t1) { b1 = atomicAdd(&E.ctr->n_items, (u64)t1); b2 = atomicAdd(&E.ctr->n_res, (u64)t2); }
            s_base[1] = b1; s_base[2] = b2;
        }
        __syncthreads();
        u64 ib0 = s_base[1], ro0 = s_base[2];
        for (int w = 0; w < (tid >> 6); w++) { ib0 += s_tot[w][1]; ro0 += s_tot[w][2]; }
        // the 400-byte copy of each kept read is left to k_retain (one half-block per read, all reads in parallel);
        // doing it here, read after read inside the wave, was the longest chain of this kernel
        if (no > 0) { E.ret_len[slot] = (u16)lw; E.ret_ridx[slot] = read_base + r; E.ret_nrec[slot] = 0; E.ret_cpos[slot] = (u32)c; }
        // Q3: the mate (the neighbouring candidate when the two are reads 2k, 2k+1) and where this read's items start
        u32 mate_slot = 0xFFFFFFFFu;
        if (paired) {
            const u32 other_r = (u32)__shfl_xor((int)r, 1); const u64 other_slot = (u64)__shfl_xor((long long)slot, 1); const bool other_in = __shfl_xor((int)(c < n_cand), 1) != 0;
            if (other_in && (other_r ^ 1u) == r && other_slot != ~0ull) mate_slot = (u32)other_slot;
        }
        // item slots and result rows (reserved above)
        u64 ib = ib0 + pre_items, ro = ro0 + pre_res;
        if (no > 0) { E.ret_mate[slot] = mate_slot; E.ret_item0[slot] = (u32)ib; E.ret_nitems[slot] = (u8)no; }
        for (int u = 0; u < no; u++) {
            u32 np = E.loci[items[u].locus].n_pad;
            if (ib + u >= E.cap_items) { atomicOr(&E.ctr->err, 2ull); break; }
            if (ro + np > E.cap_res) atomicOr(&E.ctr->err, 4ull);
            ItemDev itd; itd.res_off = ro; itd.ret = (u32)slot; itd.locus = items[u].locus; itd.diag = items[u].diag;
            itd.strand = items[u].strand; itd.votes = items[u].votes;
            E.items[ib + u] = itd;
            E.item_state[ib + u] = (u8)(no == 1 ? IS_SINGLE : 0);
            ro += np;
        }
    }
}

// Copy the reads k_seed decided to keep into the retained-read arena: 128 lanes per read (20 base words + 80 quality
// words).  Each copy is two dependent loads (slot -> read index -> row); a half-block that takes one read per turn spends
// the kernel waiting for them 30 times over (67 us on cfg3), so it takes eight reads per turn and has their loads in flight
// together.
__global__ __launch_bounds__(256) void k_retain(const EngineDev* __restrict__ Ep, const u32* __restrict__ packed, const u8* __restrict__ qrows,
                                                 u32 wpr, u32 qstride, u64 read_base, int qcompact) {
    const EngineDev& E = *Ep;
    constexpr int U = 8;
    const u64 begin = E.ctr->ret_done, end = E.ctr->n_ret < E.cap_ret ? E.ctr->n_ret : E.cap_ret;
    const u32 sub = threadIdx.x & 127;
    if (sub >= RW + RQ / 4) return;
    const bool is_base = sub < RW; const u32 w = is_base ? sub : sub - RW;
    for (u64 s0 = begin + ((u64)blockIdx.x * 2 + (threadIdx.x >> 7)) * U; s0 < end; s0 += (u64)gridDim.x * 2 * U) {
        u64 rr[U], qr[U]; u32 nn[U];
        #pragma unroll
        for (int u = 0; u < U; u++) {
            const u64 sl = s0 + u; rr[u] = 0; nn[u] = 0; qr[u] = 0;
            if (sl < end) { rr[u] = E.ret_ridx[sl] - read_base; nn[u] = E.ret_len[sl] & 0x7FFFu; qr[u] = qcompact ? (u64)E.ret_cpos[sl] : rr[u]; }
        }
        u32 val[U];
        #pragma unroll
        for (int u = 0; u < U; u++) {
            val[u] = 0;
            if (s0 + u >= end) continue;
            if (is_base) { if (w < wpr) val[u] = packed[packed_index(rr[u], wpr, w)]; }
            else {
                const u32 nq = nn[u] < qstride ? nn[u] : qstride;        // bytes to keep; rows hold zeros beyond the read length
                if (w * 4 < nq) val[u] = reinterpret_cast<const u32*>(qrows + qr[u] * qstride)[w];
            }
        }
        tie_all<U>(val);
        #pragma unroll
        for (int u = 0; u < U; u++) {
            const u64 sl = s0 + u;
            if (sl >= end) continue;
            if (is_base) E.ret_bases[sl * RW + w] = val[u];
            else reinterpret_cast<GP<u32>::G*>(E.ret_quals.g() + sl * RQ)[w] = val[u];
        }
    }
}

// ------------------------------------------------------------------ shared read-orientation helpers
// Oriented read i (after reverse-complement when strand = 1) lives at source position s = strand ? n-1-i : i.
__device__ inline u32 src_base(const u32* rb, int s) { return (rb[s >> 4] >> (2 * (s & 15))) & 3u; }

__device__ inline u32 arena_word(const EngineDev& E, const LocusDev& L, int q, u32 a_local) {
    return (q >= 0 && q < (int)L.words) ? E.arena[L.arena_off + (u64)q * L.n_pad + a_local] : 0u;
}
__device__ inline u32 nmask_word(const EngineDev& E, const LocusDev& L, int q, u32 a_local) {
    return (q >= 0 && q < (int)L.nwords) ? E.nmask[L.nmask_off + (u64)q * L.n_pad + a_local] : 0u;
}
__device__ inline bool allele_is_n(const EngineDev& E, const LocusDev& L, int j, u32 a_local) {
    return L.has_n && ((nmask_word(E, L, j >> 5, a_local) >> (j & 31)) & 1u);
}
// gather the even bits of x into the low 16 bits
__device__ inline u32 compress16(u32 x) {
    x &= 0x55555555u; x = (x | (x >> 1)) & 0x33333333u; x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu; x = (x | (x >> 8)) & 0xFFFFu; return x;
}

// ---- Ungapped local alignment in bit-plane form (k_extend).  A base is two bits; kept as two planes of 32 bases per
// word (low bits, high bits), the mismatch mask of 32 columns is (rl ^ al) | (rh ^ ah): no bit gathering, and the
// read planes come straight out of ballots.  k_extend is VALU-bound (a wave64 VALU op takes 4 cycles on a SIMD16),
// so what counts here is instructions per (read, allele) pair.
// Returns the read's default mismatch penalty: the penalty of its middle base (reads are mostly one quality value --
// Phred 40 in simulations, the top bin of binned instruments -- and only positions whose penalty differs from the
// default cost an LDS lookup in the Kadane pass).  Any choice gives the same result; this one is only the fast one.
__device__ inline int stage_read_planes(const EngineDev& E, const KParams& P, const ItemDev& it, int n,
                                        u32* s_rl, u32* s_rh, u32* s_rn, u32* s_odd, u8* s_pen, const u8* s_pentab, int tid, int nthreads) {
    auto rb = E.ret_bases.g() + (u64)it.ret * RW;
    auto rq = E.ret_quals.g() + (u64)it.ret * RQ;
    const u8 qmid = rq[n >> 1];
    const u8 pen_def = (qmid >> 7) ? (u8)P.n_penalty : s_pentab[qmid & 0x7F];
    // (issuing the loads of all passes in one batch was measured: k_extend 554 -> 610 us on cfg3 -- the kernel is bound by
    // instruction issue, not by this chain; the same change in stage_read, whose kernel waits on it, stays)
    for (int i0 = 0; i0 < RQ; i0 += nthreads) {
        int i = i0 + tid; u32 b = 0, isn = 0, odd = 0;
        if (i < n) {
            int s = it.strand ? n - 1 - i : i; u8 qb = rq[s];
            b = (rb[s >> 4] >> (2 * (s & 15))) & 3u; if (it.strand) b ^= 3u;
            isn = qb >> 7;
            u8 pen = isn ? (u8)P.n_penalty : s_pentab[qb & 0x7F];
            s_pen[i] = pen;
            odd = pen != pen_def;
        }
        u64 bl = __ballot(b & 1u), bh = __ballot(b >> 1), bn = __ballot(isn != 0), bo = __ballot(odd != 0);
        if ((tid & 63) == 0 && i < RQ) {
            int w = i >> 5;
            s_rl[w] = (u32)bl; s_rl[w + 1] = (u32)(bl >> 32); s_rh[w] = (u32)bh; s_rh[w + 1] = (u32)(bh >> 32);
            s_rn[w] = (u32)bn; s_rn[w + 1] = (u32)(bn >> 32); s_odd[w] = (u32)bo; s_odd[w + 1] = (u32)(bo >> 32);
        }
    }
    return (int)pen_def;
}

// a * b + c with 24-bit factors, b wave-uniform (full-rate v_mad_i32_i24; the host pass only type-checks device code)
__device__ inline int mad24(int a, int b, int c) {
#if defined(__HIP_DEVICE_COMPILE__)
    int r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c)); return r;      // b: wave-uniform
#else
    return a * b + c;
#endif
}
// a + b + c, b wave-uniform (one v_add3_u32; left to itself the compiler re-associates the Kadane update into three adds)
__device__ inline int add3(int a, int b, int c) {
#if defined(__HIP_DEVICE_COMPILE__)
    int r; asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c)); return r;
#else
    return a + b + c;
#endif
}
// uniform row pointer (scalar registers) + 32-bit byte offset of the lane: the saddr form of global_load
__device__ inline u32 ld_row(GP<const u32>::G* row, u32 byte_off) {
    return *reinterpret_cast<GP<const u32>::G*>(reinterpret_cast<GP<const char>::G*>(row) + byte_off);
}

// NB = 32-base read blocks the instantiation supports.  TRACK = also report the aligned span [bs, be) (only the
// gap-trigger policy needs it, and only for pairs with many mismatches).  Same recurrence as oracle align_ungapped (Kadane over the mismatch columns).
// rl/rh/od/rn = read planes, non-default-penalty mask and N mask, block-uniform (scalar registers); s_pen stays in LDS
// because they are needed only for reads with N / for the rare non-default penalty.
template <int NB, bool TRACK, bool HASN>
__device__ inline int ungapped_planes(const EngineDev& E, const KParams& P, const LocusDev& L, u32 a_local, int m, int n, int d,
                                      const u32 (&rl)[NB], const u32 (&rh)[NB], const u32 (&od)[NB], const u32 (&rn)[NB], const u8* s_pen,
                                      int pen_def, bool read_has_n, int& mm_total, int& bs, int& be) {
    const int i0 = d < 0 ? -d : 0;                             // block-uniform
    const int MA = P.match_bonus << MLST_P_SHIFT;
    const int q0 = d >> 5, r = d & 31;                         // allele block of read position 0 (floor), bit shift
    auto pbase = E.planes.g() + L.plane_off;
    auto nbase = E.nmask.g() + L.nmask_off;
    const u32 aoff = a_local * 4u;                             // byte offset of the lane's allele inside a row
    // One batch of independent loads: uniform row pointer (scalar unit) + the lane's allele index.  Blocks outside
    // the allele are clamped; they only ever meet read positions outside [i0, i1), which the masks below remove.
    u32 A[2 * (NB + 1)], AW[NB + 1];
    #pragma unroll
    for (int t = 0; t <= NB; t++) {
        int q = q0 + t; u32 qc = (u32)(q < 0 ? 0 : (q >= (int)L.pblocks ? (int)L.pblocks - 1 : q));
        // one base pointer for every load + a 32-bit byte offset (scalar row offset added to the lane's): the saddr
        // form, one 32-bit VALU add per load instead of a 64-bit one (arenas are far below 4 GB, checked at load time)
        const u32 rowoff = qc * 2u * (L.n_pad * 4u);
        A[2 * t] = ld_row(pbase, rowoff + aoff);
        A[2 * t + 1] = ld_row(pbase, rowoff + L.n_pad * 4u + aoff);
        AW[t] = 0;
        if (HASN) AW[t] = ld_row(nbase, qc * (L.n_pad * 4u) + aoff);
    }
    tie_all<2 * (NB + 1)>(A);
    if (HASN) tie_all<NB + 1>(AW);
    // m (the caller's allele_len load) is first needed here, behind the batch: one round trip, not two.  An allele
    // that does not overlap the read (i1 <= i0) falls out of the masks: M = 0 and the final value is <= P0.
    const int i1 = (m - d) < n ? (m - d) : n;                  // per allele
    mm_total = 0; bs = be = i0;
    u32 M[NB], AN[NB];
    const bool lane_mask = __any(i1 < n);                      // some allele of this wave ends inside the read
    #pragma unroll
    for (int w = 0; w < NB; w++) {
        u32 lo = __builtin_amdgcn_alignbit(A[2 * w + 2], A[2 * w], r), hi = __builtin_amdgcn_alignbit(A[2 * w + 3], A[2 * w + 1], r);
        M[w] = (lo ^ rl[w]) | (hi ^ rh[w]);
        M[w] |= rn[w];                                          // read N mask (zero for reads without N), scalar
        AN[w] = 0;
        if (HASN) { AN[w] = __builtin_amdgcn_alignbit(AW[w + 1], AW[w], r); M[w] |= AN[w]; }
        int lo_i = i0 - 32 * w; lo_i = lo_i < 0 ? 0 : (lo_i > 32 ? 32 : lo_i);            // uniform: scalar unit
        int hi_u = n - 32 * w; hi_u = hi_u < 0 ? 0 : (hi_u > 32 ? 32 : hi_u);
        M[w] &= (hi_u >= 32 ? 0xFFFFFFFFu : ((1u << hi_u) - 1u)) & ~(lo_i >= 32 ? 0xFFFFFFFFu : ((1u << lo_i) - 1u));
    }
    if (lane_mask) {                                           // one branch for all blocks
        #pragma unroll
        for (int w = 0; w < NB; w++) {
            int hi_i = i1 - 32 * w; hi_i = hi_i < 0 ? 0 : (hi_i > 32 ? 32 : hi_i);
            M[w] &= hi_i >= 32 ? 0xFFFFFFFFu : ((1u << hi_i) - 1u);
        }
    }
    #pragma unroll
    for (int w = 0; w < NB; w++) mm_total += __popc(M[w]);
    const int PD = (pen_def << MLST_P_SHIFT) + 1;
    if (!TRACK) {
        // g = cur - last*MA, so the running value just before column i is g + i*MA; gw = g + 32*w*MA is the same with
        // the column counted inside word w.  One mismatch = nine full-rate VALU operations: the value before it is one
        // 24-bit multiply-add (v_mad_i32_i24; a 32-bit multiply runs at a quarter of the rate), the new offset one
        // three-operand add.
        int gw = P0 - i0 * MA, best = P0;
        const int negMA = -MA;
        // does any mismatch of this wave sit on a column with a non-default penalty?  decided once for all words
        u32 spany = 0;
        #pragma unroll
        for (int w = 0; w < NB; w++) spany |= M[w] & (od[w] | AN[w]);
        if (__any(spany != 0)) {
            #pragma unroll
            for (int w = 0; w < NB; w++) {
                u32 Mw = M[w];
                const u32 special = od[w] | AN[w];
                while (Mw) {
                    const int bit = __ffs(Mw) - 1; Mw &= Mw - 1;
                    const int t = mad24(bit, MA, gw);
                    best = t > best ? t : best;
                    int dec = PD;
                    if ((special >> bit) & 1u) dec = ((((AN[w] >> bit) & 1u) ? P.n_penalty : (int)s_pen[32 * w + bit]) << MLST_P_SHIFT) + 1;
                    int dd = P0 - t; dd = dd > -dec ? dd : -dec;       // (value after the mismatch, floored at P0) - t
                    gw = add3(gw, negMA, dd);
                }
                gw += 32 * MA;
            }
        } else {
            #pragma unroll
            for (int w = 0; w < NB; w++) {
                u32 Mw = M[w];
                while (Mw) {
                    const int bit = __ffs(Mw) - 1; Mw &= Mw - 1;
                    const int t = mad24(bit, MA, gw);
                    best = t > best ? t : best;
                    int dd = P0 - t; dd = dd > -PD ? dd : -PD;         // (value after the mismatch, floored at P0) - t
                    gw = add3(gw, negMA, dd);
                }
                gw += 32 * MA;
            }
        }
        const int t = gw + (i1 - 32 * NB) * MA;
        return t > best ? t : best;
    }
    if (i1 <= i0) return P0;
    int cur = P0, best = P0, cs = i0, last = i0, blen = 0, bend = i0;
    #pragma unroll
    for (int w = 0; w < NB; w++) {
        u32 Mw = M[w];
        const u32 special = od[w] | AN[w];
        while (Mw) {
            int bit = __ffs(Mw) - 1; Mw &= Mw - 1;
            int i = 32 * w + bit;
            cur += (i - last) * MA;
            if (cur > best) { best = cur; blen = i - cs; bend = i; }
            int dec = PD;
            if ((special >> bit) & 1u) dec = ((((AN[w] >> bit) & 1u) ? P.n_penalty : (int)s_pen[i]) << MLST_P_SHIFT) + 1;
            cur -= dec;
            if (cur <= P0) { cur = P0; cs = i + 1; }
            last = i + 1;
        }
    }
    cur += (i1 - last) * MA;
    if (cur > best) { best = cur; blen = i1 - cs; bend = i1; }
    be = bend; bs = bend - blen;
    return best;
}
// the policy deciding whether the banded Smith-Waterman runs for a pair (same expression as oracle align_pair)
__device__ inline bool gap_trigger(const KParams& P, int mm, int xm, int score, int floor_n, int m, int n, int d, int bs, int be) {
    if (P.trig < 0) return true;
    int i0 = d < 0 ? -d : 0, i1 = (m - d) < n ? (m - d) : n;
    int overlap = i1 > i0 ? i1 - i0 : 0;
    int clipped = overlap - (be - bs);          // overlap columns the ungapped alignment left out
    return mm > P.trig && score >= floor_n && clipped >= P.clip && 2 * (mm - xm) >= clipped;
}

// ------------------------------------------------------------------ K3: extension of every item against every allele of its locus
// accept test of metamlst.py:115 on one result word; f15 = XM, or XO when the read has a single record (Q1)
__device__ inline bool accept_rec(const KParams& P, u32 r, int n, bool use_xo) {
    int score = (int)(r & 0x3FF), xm = (int)((r >> 10) & 0xFF), xo = (int)((r >> 18) & 0x7F);
    return score >= P.minscore && n >= P.min_read_len && (use_xo ? xo : xm) <= P.max_xm;
}

// Take a ticket from queue q unless a plain look shows it already drained (a stale look only costs one atomic).
__device__ inline u64 ext_steal(const EngineDev& E, u32 q, u64 begin, u64 end) {
    u64 seen = __hip_atomic_load((u64*)&E.ctr->ext_q[q][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (begin + q + (u64)EXT_Q * seen >= end) return end;
    return begin + q + (u64)EXT_Q * atomicAdd(&E.ctr->ext_q[q][0], 1ull);
}

// ---- The pair-by-pair form of the extension (rounds 1-3): every (item, allele) pair aligned on its own, lanes = alleles.
// It keeps the loci the block-haplotype kernel does not take (more alleles than MLST_EXT_HAP_MAX, default 512: an item of such
// a locus wants a workgroup of several waves, and with one the haplotype form gains nothing -- profiles/round4/README.md; or
// no tables); both kernels walk the items of a submission and skip the other one's loci (LocusDev::hap_ok).
__device__ inline u64 ext_steal2(const EngineDev& E, u32 q, u64 begin, u64 end) {
    u64 seen = __hip_atomic_load((u64*)&E.ctr->ext_q2[q][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (begin + q + (u64)EXT_Q * seen >= end) return end;
    return begin + q + (u64)EXT_Q * atomicAdd(&E.ctr->ext_q2[q][0], 1ull);
}
template <int NB>
__device__ __forceinline__ void extend_pairs_body(const EngineDev* __restrict__ Ep, const KParams& P) {
    const EngineDev& E = *Ep;     // device-resident descriptor: fields are scalar-loaded on demand
    __shared__ u32 s_rl[RW / 2 + 2]; __shared__ u32 s_rh[RW / 2 + 2]; __shared__ u32 s_rn[RW / 2 + 2]; __shared__ u32 s_odd[RW / 2 + 2];
    __shared__ u8 s_pen[RQ]; __shared__ u8 s_pentab[128];
    __shared__ u32 s_cnt[16][3];
    const int tid = threadIdx.x, nthr = blockDim.x, nwv = blockDim.x >> 6;      // 64..1024 threads per work item
    for (int i = tid; i < 128; i += nthr) s_pentab[i] = E.pen_tab[i];
    u64 c_tot = 0, c_ign = 0;                     // block-level counters, flushed once at the end (thread 0)
    const u64 begin = E.ctr->items_done, end = E.ctr->n_items < E.cap_items ? E.ctr->n_items : E.cap_items;
    // Work queue: items cost different amounts (mismatch density, allele count of the locus), so blocks take the
    // next item from a counter instead of a fixed stride.  The ticket for item k+1 is drawn while item k is staged.
    __shared__ u64 s_next;
    u32 myq = blockIdx.x % EXT_Q, tried = 0;      // thread 0 only: current queue, exhausted queues seen in a row
    if (tid == 0) s_next = begin + myq + (u64)EXT_Q * atomicAdd(&E.ctr->ext_q2[myq][0], 1ull);
    __syncthreads();
    u64 ii = uniform_u64(s_next);                 // readfirstlane: keeps the per-item descriptor loads and index math scalar
    if (ii >= end) {                              // own queue already empty: steal (block-uniform branch)
        if (tid == 0) {
            u64 nx = ii;
            while (nx >= end && ++tried < EXT_Q) { myq = (myq + 1) % EXT_Q; nx = ext_steal2(E, myq, begin, end); }
            s_next = nx; tried = 0;
        }
        __syncthreads();
        ii = uniform_u64(s_next);
    }
    while (ii < end) {                            // block-uniform
        u64 ticket = 0;
        if (tid == 0) ticket = atomicAdd(&E.ctr->ext_q2[myq][0], 1ull);
        ItemDev it = E.items[ii];
        const LocusDev L = E.loci[it.locus];
        if (L.hap_ok) {                               // the locus belongs to k_extend_160 / _320 (block-uniform)
            if (tid == 0) {
                u64 nx = begin + myq + (u64)EXT_Q * ticket;
                while (nx >= end && ++tried < EXT_Q) { myq = (myq + 1) % EXT_Q; nx = ext_steal2(E, myq, begin, end); }
                s_next = nx; tried = 0;
            }
            __syncthreads();
            ii = uniform_u64(s_next);
            continue;
        }
        u32 lw = E.ret_len[it.ret]; int n = (int)(lw & 0x7FFFu); bool read_has_n = (lw & 0x8000u) != 0;
        u8 state = E.item_state[ii];
        __syncthreads();
        const int pen_def = __builtin_amdgcn_readfirstlane(stage_read_planes(E, P, it, n, s_rl, s_rh, s_rn, s_odd, s_pen, s_pentab, tid, nthr));
        __syncthreads();
        const bool res_ok = it.res_off + L.n_pad <= E.cap_res;      // else flagged by k_seed
        const int floor_n = E.floor_tab[n];
        u32 rl[NB], rh[NB], od[NB], rn[NB];       // block-uniform read planes, held in scalar registers
        #pragma unroll
        for (int w = 0; w < NB; w++) {
            rl[w] = __builtin_amdgcn_readfirstlane(s_rl[w]); rh[w] = __builtin_amdgcn_readfirstlane(s_rh[w]);
            od[w] = __builtin_amdgcn_readfirstlane(s_odd[w]);
            rn[w] = read_has_n ? __builtin_amdgcn_readfirstlane(s_rn[w]) : 0u;
        }
        u32 nrec = 0, ndp = 0;
        for (u32 a = tid; res_ok && a < L.n_alleles; a += nthr) {
            int m = (int)E.allele_len[L.a_begin + a];
            int mm, bs, be;
            int best = L.has_n ? ungapped_planes<NB, false, true>(E, P, L, a, m, n, it.diag, rl, rh, od, rn, s_pen, pen_def, read_has_n, mm, bs, be)
                               : ungapped_planes<NB, false, false>(E, P, L, a, m, n, it.diag, rl, rh, od, rn, s_pen, pen_def, read_has_n, mm, bs, be);
            int score = best >> MLST_P_SHIFT, xm = 255 - (best & 0xFF), xo = 127 - ((best >> 8) & 0x7F);
            bool need_dp = P.trig < 0;
            if (!need_dp && mm > P.trig && score >= floor_n) {       // rare: the policy needs the aligned span
                if (L.has_n) ungapped_planes<NB, true, true>(E, P, L, a, m, n, it.diag, rl, rh, od, rn, s_pen, pen_def, read_has_n, mm, bs, be);
                else ungapped_planes<NB, true, false>(E, P, L, a, m, n, it.diag, rl, rh, od, rn, s_pen, pen_def, read_has_n, mm, bs, be);
                need_dp = gap_trigger(P, mm, xm, score, floor_n, m, n, it.diag, bs, be);
            }
            u32 r = pack_result(score, xm, xo);
            if (need_dp) { r |= R_NEEDDP; ndp++; }
            else if (score >= floor_n && score > 0) { r |= R_REC; nrec++; }
            // banded-SW worklist: one returning atomic per wave, not per pair
            u64 wm = __ballot(need_dp);
            if (wm) {
                int lane = tid & 63, leader = __ffsll((long long)wm) - 1; u64 base = 0;
                if (lane == leader) base = atomicAdd(&E.ctr->n_dp, (u64)__popcll(wm));
                base = __shfl(base, leader);
                if (need_dp) { u64 slot = base + __popcll(wm & ((1ull << lane) - 1));
                               if (slot < E.cap_dp) E.dp_list[slot] = (ii << 20) | (u64)a; else atomicOr(&E.ctr->err, 8ull); }
            }
            E.res[it.res_off + a] = r;
        }
        nrec = wave_sum_u32(nrec); ndp = wave_sum_u32(ndp);
        if ((tid & 63) == 0) { s_cnt[tid >> 6][0] = nrec; s_cnt[tid >> 6][1] = ndp; }
        __syncthreads();
        u32 tot_rec = 0, tot_dp = 0;
        for (int w = 0; w < nwv; w++) { tot_rec += s_cnt[w][0]; tot_dp += s_cnt[w][1]; }
        if (tid == 0 && tot_rec) atomicAdd(&E.ret_nrec[it.ret], tot_rec);   // per-read record count (Q1)
        // Fused accumulation (metamlst.py:101-130) when everything about this read is known here: no pair is waiting for the
        // banded SW, and whether the read has exactly one record (Q1: column 15 is then XO) does not depend on its other work
        // items -- it has none, or this item alone holds two records or more (then it has not), or none at all.  (On databases
        // with near-duplicate loci most reads hold several items, each with hundreds of records: left to k_accumulate, which
        // walks open items one per workgroup turn, they were 0.39 ms of the skewed database's step.)
        if (res_ok && tot_dp == 0 && ((state & IS_SINGLE) || !P.quirk || tot_rec != 1)) {
            bool use_xo = P.quirk && tot_rec == 1;
            const bool packed_acc = E.cap_items < (1ull << 24);      // count << 40 | sum in one 64-bit addition (see extend_body)
            u32 acc = 0, ign = 0;
            for (u32 a = tid; a < L.n_alleles; a += nthr) {
                u32 r = E.res[it.res_off + a];
                if (!(r & R_REC)) continue;
                if (accept_rec(P, r, n, use_xo)) {
                    if (packed_acc) atomicAdd(&E.acc64[L.a_begin + a], (1ull << 40) | (u64)(r & 0x3FF));
                    else { atomicAdd((u64*)&E.sum_score[L.a_begin + a], (u64)(r & 0x3FF)); atomicAdd(&E.n_hits[L.a_begin + a], 1u); }
                    acc++;
                } else ign++;
            }
            acc = wave_sum_u32(acc); ign = wave_sum_u32(ign);
            __syncthreads();
            if ((tid & 63) == 0) { s_cnt[tid >> 6][0] = acc; s_cnt[tid >> 6][1] = ign; }
            __syncthreads();
            if (tid == 0) {
                u32 A = 0, I = 0;
                for (int w = 0; w < nwv; w++) { A += s_cnt[w][0]; I += s_cnt[w][1]; }
                c_tot += tot_rec; c_ign += I;
                E.item_state[ii] = (u8)(state | IS_DONE | (A ? IS_ACC : 0));
            }
        }
        if (tid == 0) {
            u64 nx = begin + myq + (u64)EXT_Q * ticket;
            while (nx >= end && ++tried < EXT_Q) { myq = (myq + 1) % EXT_Q; nx = ext_steal2(E, myq, begin, end); }
            s_next = nx; tried = 0;
        }
        __syncthreads();
        ii = uniform_u64(s_next);
    }
    if (tid == 0) {
        if (c_tot) atomicAdd(&E.ctr->cnt[MLST_CNT_TOTAL_RECORDS], c_tot);
        if (c_ign) atomicAdd(&E.ctr->cnt[MLST_CNT_IGNORED], c_ign);
    }
}

// Two instantiations: reads up to 160 bases (five 32-base blocks; held to 72 VGPRs = 7 waves per SIMD, which measured
// 3 % faster than the 80 the allocator takes when left alone) and up to MLST_MAX_READ_LEN.
__attribute__((amdgpu_waves_per_eu(7, 7)))
__global__ __launch_bounds__(1024) void k_extend_pairs_160(const EngineDev* __restrict__ Ep, KParams P) { extend_pairs_body<5>(Ep, P); }
__global__ __launch_bounds__(1024) void k_extend_pairs_320(const EngineDev* __restrict__ Ep, KParams P) { extend_pairs_body<RW / 2>(Ep, P); }

// ---- item records (k_ext_prep -> k_extend).  Everything k_extend needs to know about a work item, gathered into one
// contiguous record by a kernel whose lanes are items: read on its own, a workgroup of k_extend followed a chain of ~8
// dependent round trips per item (ticket -> item -> locus, read length -> quality and base rows -> ballots -> block
// table), ~25 with the six record loads and five id loads behind it, and at 7 waves per SIMD that chain, not the vector
// units, set its pace (PMC, profiles/round4/README.md).  Words of a record (NBA = NB + 1 blocks):
//   0 flags   1 n | pen_def << 16   2 diag   3 ret   4 locus   5,6 res_off   7 a_begin   8 n_alleles   9 n_pad
//   10 hap_off   11,12 hid_off   13 floor_n   14 pblocks   15 hap_win (of this instantiation)
//   16 .. 16+NBA      first record of block q0 + t in the locus' table (clamped: empty outside the allele), t = 0 .. NBA
//   31 work item
//   32 + 4 t          the read's planes funnel-shifted onto block t: low, high, N mask, non-default-penalty mask
//   32 + 4 NBA + t    read columns that exist in block t
#define XF_SINGLE 1u
#define XF_RESOK 2u
#define XF_HAPOK 4u
#define XF_READN 8u
#define XF_SPECIAL 16u      /* a column with a non-default penalty or an N in the read, or a locus with N columns: s_pen is needed */
#define XF_STRAND 32u
template <int NB> struct XRec {
    static constexpr int NBA = NB + 1, HB = 16, PL = 32, VR = 32 + 4 * NBA, WORDS = (VR + NBA + 15) & ~15;
};

template <int NB>
__global__ __launch_bounds__(256) void k_ext_prep(const EngineDev* __restrict__ Ep, KParams P, u32* __restrict__ xrec, u64 cap_xrec) {
    // A wave takes FOUR items per turn: an item's record hangs on a chain of three round trips (item -> locus, length ->
    // rows), and with one item per wave and turn that chain, 15 times per wave, was the kernel (0.09 ms for the 121 k items of
    // cfg3).  Lanes 0-3 walk the first two links for the four items side by side; the rows of all four are requested in
    // one batch; 16 lanes per item then shift its planes onto the allele's block grid.  LDS is private to the wave: no barriers.
    typedef XRec<NB> X;
    constexpr int B = 4, KT = RQ / 64;
    const EngineDev& E = *Ep;
    __shared__ u32 s_pl[4][B][4][RW / 2 + 2];      // per wave and item: low plane, high plane, N mask, non-default-penalty mask
    __shared__ u8 s_pentab[128]; __shared__ __attribute__((aligned(16))) u32 s_w[4][B][X::WORDS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 128; i += 256) s_pentab[i] = E.pen_tab[i];
    __syncthreads();
    const u64 begin = E.ctr->items_done, end0 = E.ctr->n_items < E.cap_items ? E.ctr->n_items : E.cap_items;
    const u64 end = end0 - begin > cap_xrec ? begin + cap_xrec : end0;
    for (u64 i0 = begin + ((u64)blockIdx.x * 4 + wv) * B; i0 < end; i0 += (u64)gridDim.x * 4 * B) {      // wave-uniform
        const int n_live = (int)(end - i0 < (u64)B ? end - i0 : (u64)B);
        // ---- links one and two of the chain, lane j for item j
        const bool mine = lane < n_live;
        ItemDev it; it.res_off = 0; it.ret = 0; it.locus = 0; it.diag = 0; it.strand = 0; it.votes = 0;
        if (mine) it = E.items[i0 + lane];
        TIE1(it.ret); TIE1(it.locus);
        u32 lw = 0, st = 0; LocusDev L; memset(&L, 0, sizeof L);
        if (mine) { L = E.loci[it.locus]; lw = E.ret_len[it.ret]; st = E.item_state[i0 + lane]; }
        TIE1(lw); TIE1(L.hap_ok); TIE1(L.hblk_off); TIE1(L.pblocks);
        const int n_l = (int)(lw & 0x7FFFu);
        const int floor_l = mine ? E.floor_tab[n_l] : 0;
        // ---- the rows of the four items, one batch: lane = read position modulo 64, KT turns per item
        u32 qv[B][KT], bw[B][KT], qmid[B];
        #pragma unroll
        for (int k = 0; k < B; k++) {
            const u32 ret_k = __shfl(it.ret, k); const int n_k = __shfl(n_l, k); const int strand_k = __shfl((int)it.strand, k);
            auto rb = E.ret_bases.g() + (u64)ret_k * RW; auto rq = E.ret_quals.g() + (u64)ret_k * RQ;
            qmid[k] = k < n_live ? (u32)rq[n_k >> 1] : 0u;
            #pragma unroll
            for (int t = 0; t < KT; t++) {
                const int i = lane + 64 * t, sp = (k < n_live && i < n_k) ? (strand_k ? n_k - 1 - i : i) : 0;
                qv[k][t] = rq[sp]; bw[k][t] = rb[sp >> 4];
            }
        }
        // the block table: lane = t + 16 k, block t of item k
        const int kk = lane >> 4, tt = lane & 15;
        const int diag_g = __shfl(it.diag, kk); const u32 pbl_g = __shfl(L.pblocks, kk), hoff_g = __shfl(L.hblk_off, kk), hok_g = __shfl(L.hap_ok, kk);
        u32 hbv = 0;
        if (kk < n_live && tt <= X::NBA && hok_g) { int q = (diag_g >> 5) + tt; q = q < 0 ? 0 : (q > (int)pbl_g ? (int)pbl_g : q); hbv = E.hap_blk[hoff_g + (u32)q]; }
        #pragma unroll
        for (int k = 0; k < B; k++) { tie_all<KT>(qv[k]); tie_all<KT>(bw[k]); }
        tie_all<B>(qmid); TIE1(hbv);
        // ---- planes of each item (ballots), to the wave's LDS
        int pen_def_k[B];
        #pragma unroll
        for (int k = 0; k < B; k++) {
            const int n_k = __shfl(n_l, k); const int strand_k = __shfl((int)it.strand, k);
            const u8 pd = (qmid[k] >> 7) ? (u8)P.n_penalty : s_pentab[qmid[k] & 0x7Fu];
            pen_def_k[k] = (int)pd;
            #pragma unroll
            for (int t = 0; t < KT; t++) {
                const int i = lane + 64 * t; u32 b = 0, isn = 0, odd = 0;
                if (k < n_live && i < n_k) {
                    const int sp = strand_k ? n_k - 1 - i : i;
                    b = (bw[k][t] >> (2 * (sp & 15))) & 3u; if (strand_k) b ^= 3u;
                    isn = qv[k][t] >> 7;
                    const u8 pen = isn ? (u8)P.n_penalty : s_pentab[qv[k][t] & 0x7Fu];
                    odd = pen != pd;
                }
                const u64 bl = __ballot(b & 1u), bh = __ballot(b >> 1), bn = __ballot(isn != 0), bo = __ballot(odd != 0);
                if (lane == 0) {
                    s_pl[wv][k][0][2 * t] = (u32)bl; s_pl[wv][k][0][2 * t + 1] = (u32)(bl >> 32); s_pl[wv][k][1][2 * t] = (u32)bh; s_pl[wv][k][1][2 * t + 1] = (u32)(bh >> 32);
                    s_pl[wv][k][2][2 * t] = (u32)bn; s_pl[wv][k][2][2 * t + 1] = (u32)(bn >> 32); s_pl[wv][k][3][2 * t] = (u32)bo; s_pl[wv][k][3][2 * t + 1] = (u32)(bo >> 32);
                }
            }
        }
        // ---- the records: 16 lanes per item (lane t of them = block t), lane 16 k its header
        u32 any = 0;
        const int n_g = __shfl(n_l, kk);
        if (kk < n_live) {
            u32* w = s_w[wv][kk];
            const u32 rs = 32u - ((u32)diag_g & 31u);      // 1..32
            if (tt <= X::NBA) w[X::HB + tt] = hbv;
            if (tt <= NB) {
                auto sh = [&](const u32* A) { const u64 v2 = ((u64)(tt < NB ? A[tt] : 0u) << 32) | (u64)(tt > 0 ? A[tt - 1] : 0u); return (u32)(v2 >> rs); };
                auto vw = [&](int j) { const int c = n_g - 32 * j; return (j < 0 || j >= NB) ? 0u : (c >= 32 ? 0xFFFFFFFFu : (c > 0 ? ((1u << c) - 1u) : 0u)); };
                const u32 rn_t = sh(s_pl[wv][kk][2]), od_t = sh(s_pl[wv][kk][3]);
                w[X::PL + 4 * tt] = sh(s_pl[wv][kk][0]); w[X::PL + 4 * tt + 1] = sh(s_pl[wv][kk][1]); w[X::PL + 4 * tt + 2] = rn_t; w[X::PL + 4 * tt + 3] = od_t;
                w[X::VR + tt] = (u32)((((u64)vw(tt) << 32) | (u64)vw(tt - 1)) >> rs);
                any = rn_t | od_t;
            }
        }
        // (does any of the item's 16 lanes see a special column?  an OR over the item's quarter of the wave)
        #pragma unroll
        for (int o = 8; o > 0; o >>= 1) any |= __shfl_xor(any, o);
        const u32 special_l = __shfl(any, lane * 16);      // lane j < 4: the OR of item j's lanes (lane 16 j holds it)
        if (mine) {
            u32* w = s_w[wv][lane];
            if (!L.hap_ok) { for (int j = 0; j < 32; j++) w[j] = 0; }      // the other kernel's item: a record without XF_HAPOK
            else {
                const bool res_ok = it.res_off + L.n_pad <= E.cap_res;
                w[0] = ((st & IS_SINGLE) ? XF_SINGLE : 0u) | (res_ok ? XF_RESOK : 0u) | XF_HAPOK | ((lw & 0x8000u) ? XF_READN : 0u)
                     | ((special_l || L.has_n) ? XF_SPECIAL : 0u) | (it.strand ? XF_STRAND : 0u);
                int pd = pen_def_k[0];
                #pragma unroll
                for (int k = 1; k < B; k++) pd = lane == k ? pen_def_k[k] : pd;
                w[1] = (u32)n_l | ((u32)pd << 16); w[2] = (u32)it.diag; w[3] = it.ret; w[4] = it.locus; w[5] = (u32)it.res_off; w[6] = (u32)(it.res_off >> 32);
                w[7] = L.a_begin; w[8] = L.n_alleles; w[9] = L.n_pad; w[10] = L.hap_off; w[11] = (u32)L.hid_off; w[12] = (u32)(L.hid_off >> 32);
                w[13] = (u32)floor_l; w[14] = L.pblocks; w[15] = L.hap_win[NB > 5 ? 1 : 0];
            }
            w[31] = (u32)(i0 + lane);                        // (max_items < 2^32)
        }
        #pragma unroll
        for (int k = 0; k < B; k++) {
            if (k >= n_live) break;
            u32* dst = xrec + (i0 + k - begin) * X::WORDS;
            for (int j = lane; j < X::WORDS; j += 64) dst[j] = s_w[wv][k][j];
        }
    }
}

// ---- Block-haplotype form of the ungapped extension (round 4).  The consumer, metamlst.py:133-151, needs one score per
// allele, and the alleles of a locus differ by a few SNPs (metaMLST_functions.py:149-161 dumps them one by one): a 32-base
// block of a 300-allele locus has ~55 distinct contents, of a 1,430-allele locus ~180 (profiles/round4/hap_counts.md).
// The Kadane recurrence on the packed value is max-plus linear: a block maps the running value v to max(v + T, R) and
// the best value seen so far to max(best, v + Pm, B) -- four integers, exact (sums and maxima of the same terms the
// column-by-column walk adds and compares).  So a work item (1) has the READ shifted onto the allele's block grid, once
// (k_ext_prep); (2) scores every distinct haplotype of every covered block (lanes = haplotypes, summaries to LDS);
// (3) composes each allele from its 5-6 (10-11) summaries: one 16-byte LDS read and five integer operations per block.
// The mismatch count of the full overlap is minus the low 16 bits of the summed T (every mismatch subtracts
// (penalty << 16) + 1).  Pairs whose gap-trigger test needs the aligned span go through ungapped_planes<TRACK> as before.
__device__ inline int4 hap_summary(const KParams& P, const HapRec h, u32 rl, u32 rh, u32 rn, u32 od, u32 vr, const u8* s_pen, int pen_base,
                                   int pen_def, bool special_any) {
    const int MA = P.match_bonus << MLST_P_SHIFT, negMA = -MA;
    const u32 lm = h.len >= 32u ? 0xFFFFFFFFu : ((1u << h.len) - 1u);
    const u32 valid = vr & lm;                                  // columns that exist in the read and in the haplotype: one run of bits
    u32 M = ((h.lo ^ rl) | (h.hi ^ rh) | rn | h.nm) & valid;
    const int lo_c = __ffs((int)valid) - 1, hi_c = lo_c + __popc(valid);      // valid = 0: lo_c = hi_c = -1
    // gu / gf: value on the path that never restarted (relative to v) / on the best path that did (absolute), both
    // minus MA * (columns counted so far), so that the value just before column c is g + c * MA
    int gu = mad24(lo_c, negMA, 0), gf = NEGP, Pm = NEGP, B = NEGP;
    const int PDM = (pen_def << MLST_P_SHIFT) + 1 + MA, floorv = P0 - MA;
    if (special_any) {
        const u32 special = od | h.nm;
        while (M) {
            const int bit = __ffs((int)M) - 1; M &= M - 1;
            const int tu = mad24(bit, MA, gu), tf = mad24(bit, MA, gf);
            Pm = tu > Pm ? tu : Pm; B = tf > B ? tf : B;
            int dm = PDM;
            if ((special >> bit) & 1u) dm = ((((h.nm >> bit) & 1u) ? P.n_penalty : (int)s_pen[pen_base + bit]) << MLST_P_SHIFT) + 1 + MA;
            gu -= dm;
            int x = tf - dm; x = x > floorv ? x : floorv;          // max(value after the mismatch, P0) - MA
            gf = mad24(bit, negMA, x);
        }
    } else {
        while (M) {
            const int bit = __ffs((int)M) - 1; M &= M - 1;
            const int tu = mad24(bit, MA, gu), tf = mad24(bit, MA, gf);
            Pm = tu > Pm ? tu : Pm; B = tf > B ? tf : B;
            gu -= PDM;
            int x = tf - PDM; x = x > floorv ? x : floorv;
            gf = mad24(bit, negMA, x);
        }
    }
    return make_int4(mad24(hi_c, MA, gu), mad24(hi_c, MA, gf), Pm, B);
}

// Profiling build only (-DMLST_EXT_TRACE, profiles/extend_phases.sh): shader-clock cycles a wave of k_extend spends in each
// phase of an item, summed over all waves.  0 record, 1 requests + s_pen, 2 summaries, 3 composition, 4 counts + fused
// accumulation, 5 hand-over to the next item, 6 items, 7 waves.
#ifdef MLST_EXT_TRACE
__device__ u32 g_ext_skip;      // bit 0: no summaries, bit 1: no composition (wrong results: instruction counts by difference)
#define XSKIP(b) (xskip_ & (b))
extern "C" int mlst_debug_ext_skip(uint32_t v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_ext_skip), &v, 4) == hipSuccess ? 0 : -1; }
__device__ u64 g_ext_trace[8];
__device__ u64 g_ext_cnt[8];      // 0 items through the fast pass, 1 items that fell back, 2 pairs whose policy test needs the span, 3 turns with such a pair
extern "C" int mlst_debug_ext_cnt(uint64_t out[8], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ext_cnt), 64) != hipSuccess) return -1;
    if (reset) { u64 z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_ext_cnt), z, 64) != hipSuccess) return -1; }
    return 0;
}
#define XC(k, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_ext_cnt[k], (u64)(v)); } while (0)
#if MLST_EXT_TRACE > 1      /* cycle stamps: they serialise the wave (the kernel runs 2-3 x longer); the phases' shares only */
#define XT_DECL u64 xt_[6] = {0, 0, 0, 0, 0, 0}; u64 xt_items = 0; u64 xt_t = __builtin_readcyclecounter(); const u32 xskip_ = __builtin_amdgcn_readfirstlane(g_ext_skip)
#define XT(k) do { const u64 now_ = __builtin_readcyclecounter(); xt_[k] += now_ - xt_t; xt_t = now_; } while (0)
#define XT_ITEM xt_items++
#define XT_FLUSH do { if ((threadIdx.x & 63) == 0) { for (int k_ = 0; k_ < 6; k_++) atomicAdd(&g_ext_trace[k_], xt_[k_]); atomicAdd(&g_ext_trace[6], xt_items); atomicAdd(&g_ext_trace[7], 1ull); } } while (0)
#else
#define XT_DECL const u32 xskip_ = __builtin_amdgcn_readfirstlane(g_ext_skip)
#define XT(k)
#define XT_ITEM
#define XT_FLUSH
#endif
extern "C" int mlst_debug_ext_trace(uint64_t out[8], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ext_trace), 64) != hipSuccess) return -1;
    if (reset) { u64 z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_ext_trace), z, 64) != hipSuccess) return -1; }
    return 0;
}
#else
#define XC(k, v)
#define XSKIP(b) false
#define XT_DECL
#define XT(k)
#define XT_ITEM
#define XT_FLUSH
#endif
#define R_TRACK 0x10000000u      /* k_extend, inside one item only: the pair's gap-trigger test needs the aligned span */

template <int NB>
__device__ __forceinline__ void extend_body(const EngineDev* __restrict__ Ep, const KParams& P, const u32 lds_recs, const u32 acc_cap, const u32* __restrict__ xrec, const u64 cap_xrec) {
    typedef XRec<NB> X;
    const EngineDev& E = *Ep;     // device-resident descriptor: fields are scalar-loaded on demand
    extern __shared__ int4 s_hap[];                // haplotype summaries of the current item (lds_recs of them + the identity)
    __shared__ u32 s_rl[RW / 2 + 2]; __shared__ u32 s_rh[RW / 2 + 2]; __shared__ u32 s_rn[RW / 2 + 2]; __shared__ u32 s_odd[RW / 2 + 2];
    __shared__ u8 s_pen[RQ]; __shared__ u8 s_pentab[128];
    __shared__ u32 s_cnt[16][3];
    __shared__ __attribute__((aligned(16))) u32 s_x[2][X::WORDS];      // the records of the current and of the next item
    __shared__ u64 s_ii[2];
    const int tid = threadIdx.x, nthr = blockDim.x, nwv = blockDim.x >> 6;      // 64..1024 threads per work item
    for (int i = tid; i < 128; i += nthr) s_pentab[i] = E.pen_tab[i];
    u64 c_tot = 0, c_ign = 0;                     // block-level counters, flushed once at the end (thread 0)
    u64 N;                                        // records of this submission, in item order (k_ext_prep)
    { const u64 b0 = E.ctr->items_done, e0 = E.ctr->n_items < E.cap_items ? E.ctr->n_items : E.cap_items; N = e0 - b0; if (N > cap_xrec) N = cap_xrec; }
    // Work queue: items cost different amounts (mismatch density, allele count of the locus), so workgroups take the next
    // record from a counter instead of a fixed stride.  32 counters in separate lines; counter q hands out the records
    // q B .. (q + 1) B - 1 in order (B = N / 32 rounded up).  (The records are in item order, i.e. in read order: NOT grouped by
    // locus -- round 5 tried holding a workgroup's additions in LDS across items of one locus and found no runs to hold; with the
    // additions switched off altogether k_extend is no faster on cfg2, profiles/round5/extend.md.)  Own counter first, then the others'.  Tickets are
    // drawn two records ahead: the one after the current is known when the current one starts and is fetched meanwhile.
    const u64 QB = (N + EXT_Q - 1) / EXT_Q;
    u32 myq = blockIdx.x % EXT_Q, tried = 0;      // thread 0 only: current queue, exhausted queues seen in a row
    auto place = [&](u32 q, u64 ticket) -> u64 { const u64 pl = (u64)q * QB + ticket; return (ticket < QB && pl < N) ? pl : N; };
    auto resolve = [&](u64 ticket) -> u64 {       // thread 0: record of a ticket of queue myq; moves on to other queues when that one is drained
        u64 nx = place(myq, ticket);
        while (nx >= N && ++tried < EXT_Q) {
            myq = (myq + 1) % EXT_Q;
            const u64 seen = __hip_atomic_load((u64*)&E.ctr->ext_q[myq][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // a stale look only costs one atomic
            nx = place(myq, seen) < N ? place(myq, atomicAdd(&E.ctr->ext_q[myq][0], 1ull)) : N;
        }
        if (nx < N) tried = 0;
        return nx;
    };
    if (tid == 0) {
        s_ii[0] = resolve(atomicAdd(&E.ctr->ext_q[myq][0], 1ull));
        s_ii[1] = tried < EXT_Q ? resolve(atomicAdd(&E.ctr->ext_q[myq][0], 1ull)) : N;
    }
    u16* const s_pend = reinterpret_cast<u16*>(s_hap + lds_recs + 1);      // what the current item would add per allele, until its counts are known
    const bool packed_acc = E.cap_items < (1ull << 24);      // count << 40 | sum in one 64-bit addition (else two additions)
    __syncthreads();
    u64 pp = uniform_u64(s_ii[0]);
    if (pp < N) for (int j = tid; j < X::WORDS; j += nthr) s_x[0][j] = xrec[pp * X::WORDS + j];
    __syncthreads();
    int cur = 0;
    XT_DECL;
    while (pp < N) {                              // block-uniform
        XT_ITEM;
        const u64 pp_next = uniform_u64(s_ii[cur ^ 1]);
        // the next record: requested now, parked in LDS once a wait for loads has passed anyway
        u32 pf[(X::WORDS + 63) / 64];
        #pragma unroll
        for (int k = 0; k < (X::WORDS + 63) / 64; k++) { pf[k] = 0; if (pp_next < N && tid + 64 * k < X::WORDS && tid < 64) pf[k] = xrec[pp_next * X::WORDS + tid + 64 * k]; }
        u64 ticket = 0;
        if (tid == 0 && tried < EXT_Q) ticket = atomicAdd(&E.ctr->ext_q[myq][0], 1ull);
        asm volatile("" ::: "memory");
        const u32* x = s_x[cur];
        const uint4 h0 = *reinterpret_cast<const uint4*>(x), h1 = *reinterpret_cast<const uint4*>(x + 4), h2 = *reinterpret_cast<const uint4*>(x + 8), h3 = *reinterpret_cast<const uint4*>(x + 12);
        const u32 flags = __builtin_amdgcn_readfirstlane(h0.x);
        const u64 ii = (u64)(u32)__builtin_amdgcn_readfirstlane(x[31]);      // the work item of this record
        const int n = __builtin_amdgcn_readfirstlane(h0.y & 0xFFFFu), pen_def = __builtin_amdgcn_readfirstlane(h0.y >> 16);
        const int diag = __builtin_amdgcn_readfirstlane((int)h0.z);
        const u32 ret = __builtin_amdgcn_readfirstlane(h0.w);
        const u64 res_off = ((u64)(u32)__builtin_amdgcn_readfirstlane(h1.z) << 32) | (u32)__builtin_amdgcn_readfirstlane(h1.y);
        const u32 a_begin = __builtin_amdgcn_readfirstlane(h1.w), n_alleles = __builtin_amdgcn_readfirstlane(h2.x), n_pad = __builtin_amdgcn_readfirstlane(h2.y);
        const int floor_n = __builtin_amdgcn_readfirstlane((int)h3.y);
        const bool res_ok = flags & XF_RESOK, read_has_n = flags & XF_READN;
        const bool use_hap = res_ok && (flags & XF_HAPOK) && (u32)__builtin_amdgcn_readfirstlane(h3.w) <= lds_recs;      // block-uniform
        u32 nrec = 0, ndp = 0, ntrack = 0;        // per wave (counted with ballots: scalar registers)
        // hand-over to the next item: its record goes to the free LDS slot, the ticket drawn above names the item after it.
        // Done as soon as a wait for loads has passed anyway (behind the haplotype records): at the end of the item the
        // additions of the item are in flight, and a wait for these two values would wait for all of them (vmcnt is in order).
        bool parked = false;
        auto park = [&]() {
            #pragma unroll
            for (int k = 0; k < (X::WORDS + 63) / 64; k++) if (tid < 64 && tid + 64 * k < X::WORDS) s_x[cur ^ 1][tid + 64 * k] = pf[k];
            if (tid == 0) s_ii[cur] = tried < EXT_Q ? resolve(ticket) : N;
            parked = true;
        };
        // what follows an alignment, for both forms of it: the gap-trigger policy, the result word, the banded-SW worklist
        // (called by every lane of a turn, `valid` or not: the counts are kept per wave, in every lane's copy)
        auto emit_pair = [&](const u32 a, const int score, const int xm, const int xo, bool need_dp, const bool valid) {
            u32 r = pack_result(score, xm, xo);
            need_dp = need_dp && valid;
            const bool is_rec = valid && !need_dp && score >= floor_n && score > 0;
            if (need_dp) r |= R_NEEDDP; else if (is_rec) r |= R_REC;
            nrec += (u32)__popcll(__ballot(is_rec));
            const u64 wm = __ballot(need_dp);      // banded-SW worklist: one returning atomic per wave, not per pair
            if (wm) {
                ndp += (u32)__popcll(wm);
                int lane = tid & 63, leader = __ffsll((long long)wm) - 1; u64 base = 0;
                if (lane == leader) base = atomicAdd(&E.ctr->n_dp, (u64)__popcll(wm));
                base = __shfl(base, leader);
                if (need_dp) { u64 slot = base + __popcll(wm & ((1ull << lane) - 1));
                               if (slot < E.cap_dp) E.dp_list[slot] = (ii << 20) | (u64)a; else atomicOr(&E.ctr->err, 8ull); }
            }
            if (valid) E.res[res_off + a] = r;
        };
        // the read in its own coordinates (planes in LDS, s_pen): what rounds 1-3 staged for every item, now only for the
        // items that take the pair-by-pair path or hold a pair whose policy test needs the aligned span
        auto stage_old = [&]() {
            ItemDev it; it.res_off = res_off; it.ret = ret; it.locus = __builtin_amdgcn_readfirstlane(h1.x); it.diag = diag; it.strand = (flags & XF_STRAND) ? 1 : 0; it.votes = 0;
            __syncthreads();
            stage_read_planes(E, P, it, n, s_rl, s_rh, s_rn, s_odd, s_pen, s_pentab, tid, nthr);
            __syncthreads();
        };
        auto pairs_loop = [&](const bool only_marked) {
            const LocusDev L = E.loci[__builtin_amdgcn_readfirstlane(h1.x)];
            u32 rl[NB], rh[NB], od[NB], rn[NB];       // block-uniform read planes, held in scalar registers
            #pragma unroll
            for (int w = 0; w < NB; w++) {
                rl[w] = __builtin_amdgcn_readfirstlane(s_rl[w]); rh[w] = __builtin_amdgcn_readfirstlane(s_rh[w]);
                od[w] = __builtin_amdgcn_readfirstlane(s_odd[w]);
                rn[w] = read_has_n ? __builtin_amdgcn_readfirstlane(s_rn[w]) : 0u;
            }
            for (u32 a = tid; a < n_alleles; a += nthr) {
                const bool valid = !only_marked || (E.res[res_off + a] & R_TRACK);
                int m = (int)E.allele_len[a_begin + a];
                int mm = 0, bs = 0, be = 0, best = P0;
                if (valid) best = L.has_n ? ungapped_planes<NB, false, true>(E, P, L, a, m, n, diag, rl, rh, od, rn, s_pen, pen_def, read_has_n, mm, bs, be)
                                         : ungapped_planes<NB, false, false>(E, P, L, a, m, n, diag, rl, rh, od, rn, s_pen, pen_def, read_has_n, mm, bs, be);
                const int score = best >> MLST_P_SHIFT, xm = 255 - (best & 0xFF), xo = 127 - ((best >> 8) & 0x7F);
                bool need_dp = P.trig < 0;
                if (valid && !need_dp && mm > P.trig && score >= floor_n) {       // rare: the policy needs the aligned span
                    if (L.has_n) ungapped_planes<NB, true, true>(E, P, L, a, m, n, diag, rl, rh, od, rn, s_pen, pen_def, read_has_n, mm, bs, be);
                    else ungapped_planes<NB, true, false>(E, P, L, a, m, n, diag, rl, rh, od, rn, s_pen, pen_def, read_has_n, mm, bs, be);
                    need_dp = gap_trigger(P, mm, xm, score, floor_n, m, n, diag, bs, be);
                }
                emit_pair(a, score, xm, xo, need_dp, valid);
            }
        };
        XT(0);
        bool done_fast = false;                   // the item was accumulated in the composition pass itself (below)
        if (use_hap) {
            // ---- block-haplotype path: block t of the item is allele block (diag >> 5) + t
            const int q0 = diag >> 5;
            const u32 pblocks = __builtin_amdgcn_readfirstlane(h3.z);
            const u64 hid_off = ((u64)(u32)__builtin_amdgcn_readfirstlane(h3.x) << 32) | (u32)__builtin_amdgcn_readfirstlane(h2.w);
            // haplotype ids: rows of block PAIRS (low half = even block), lanes = alleles; the item's NB + 1 blocks lie in NR
            // rows from row q0 >> 1 on.  Uniform row pointers + the lane's offset: no vector address arithmetic per load.
            constexpr int NR = (NB + 3) / 2;
            GP<const u32>::G* idrow[NR];
            #pragma unroll
            for (int j = 0; j < NR; j++) { int r = (q0 >> 1) + j; const int nrows = (int)((pblocks + 1) >> 1); r = r < 0 ? 0 : (r >= nrows ? nrows - 1 : r); idrow[j] = E.hap_id.g() + hid_off + (u64)(u32)r * n_pad; }
            const u32 odd16 = ((u32)q0 & 1u) * 16u;
            auto load_ids = [&](const u32 a, u32 (&w)[NR]) {      // unconditional (a clamped): the compiler can count the loads in flight
                const u32 ac = a < n_alleles ? a : n_alleles - 1;
                if (XSKIP(8u)) { for (int j = 0; j < NR; j++) w[j] = 0; return; }
                #pragma unroll
                for (int j = 0; j < NR; j++) w[j] = ld_row(idrow[j], ac * 4u);
            };
            u32 hb[NB + 2];
            #pragma unroll
            for (int t = 0; t <= NB + 1; t++) hb[t] = __builtin_amdgcn_readfirstlane(x[X::HB + t]);
            auto recs = E.hap_rec.g() + __builtin_amdgcn_readfirstlane(h2.z);
            u32 maxcnt = 0;
            #pragma unroll
            for (int t = 0; t <= NB; t++) { const u32 c = hb[t + 1] - hb[t]; maxcnt = c > maxcnt ? c : maxcnt; }
            const bool special_any = flags & XF_SPECIAL;
            if (special_any) {                     // per-column penalties of the read (Phred of a column that is not the read's usual one, N)
                auto rq = E.ret_quals.g() + (u64)ret * RQ;
                for (int i = tid; i < n; i += nthr) { const u8 qb = rq[(flags & XF_STRAND) ? n - 1 - i : i]; s_pen[i] = (qb >> 7) ? (u8)P.n_penalty : s_pentab[qb & 0x7F]; }
            }
            u32 Wn[NR], Wn2[NR], Wn3[NR], Wn4[NR];   // ids of this thread's first four alleles: they do not depend on the summaries
            load_ids((u32)tid, Wn); load_ids((u32)tid + nthr, Wn2); load_ids((u32)tid + 2 * nthr, Wn3); load_ids((u32)tid + 3 * nthr, Wn4);
            const u32 ident = hb[NB + 1] - hb[0];   // one slot behind the summaries: the identity, for blocks outside the allele
            u32 sbase[NB + 1], wid[NB + 1];        // per block: first summary, width of an id (0 = block outside the allele: identity)
            #pragma unroll
            for (int t = 0; t <= NB; t++) { const bool present = hb[t + 1] != hb[t]; sbase[t] = present ? hb[t] - hb[0] : ident; wid[t] = present ? 16u : 0u; }
            if (tid == 0) s_hap[ident] = make_int4(0, NEGP, NEGP, NEGP);
            if (special_any) lds_barrier();
            XT(1);
            for (u32 hb0 = 0; hb0 < maxcnt && !XSKIP(1u); hb0 += nthr) {
                const u32 hh = hb0 + tid;
                uint4 rc[NB + 1];
                #pragma unroll
                for (int t = 0; t <= NB; t++) {
                    rc[t] = make_uint4(0, 0, 0, 0);
                    if (hh < hb[t + 1] - hb[t]) rc[t] = *reinterpret_cast<GP<const uint4>::G*>(recs + hb[t] + hh);
                }
                #pragma unroll
                for (int t = 0; t <= NB; t++) { TIE4(rc[t].x, rc[t].y, rc[t].z, rc[t].w); }
                #pragma unroll
                for (int t = 0; t <= NB; t++) {
                    if (hh < hb[t + 1] - hb[t]) {
                        const uint4 sp = *reinterpret_cast<const uint4*>(x + X::PL + 4 * t); HapRec hr; hr.lo = rc[t].x; hr.hi = rc[t].y; hr.nm = rc[t].z; hr.len = rc[t].w;
                        s_hap[(hb[t] - hb[0]) + hh] = hap_summary(P, hr, sp.x, sp.y, sp.z, sp.w, x[X::VR + t], s_pen, 32 * t - (int)((u32)diag & 31u), pen_def, special_any);
                    }
                }
            }
            park();
            lds_barrier();                         // (not __syncthreads(): that would also wait for the loads in flight)
            XT(2);
            // composition of one allele from the summaries of its blocks -> packed best value, mismatches of the full overlap
            auto compose = [&](const u32 (&w)[NR], int& best, int& mm) {
                u32 wp[NB / 2 + 1];
                #pragma unroll
                for (int k = 0; k <= NB / 2; k++) wp[k] = __builtin_amdgcn_alignbit(k + 1 < NR ? w[k + 1 < NR ? k + 1 : 0] : 0u, w[k], odd16);
                int4 S[NB + 1];
                #pragma unroll
                for (int t = 0; t <= NB; t++) S[t] = s_hap[sbase[t] + __builtin_amdgcn_ubfe(wp[t >> 1], 16u * (t & 1), wid[t])];
                int v = P0, ts = 0; best = P0;
                #pragma unroll
                for (int t = 0; t <= NB; t++) {
                    const int c1 = v + S[t].z; best = c1 > best ? c1 : best; best = S[t].w > best ? S[t].w : best;
                    const int c2 = v + S[t].x; v = c2 > S[t].y ? c2 : S[t].y; ts += S[t].x;
                }
                best = v > best ? v : best;
                mm = (-ts) & 0xFFFF;
            };
            // The gap-trigger policy of a pair with many mismatches (rare per pair, but one item in six holds such a pair on a
            // database 3 % apart): length of the ungapped local alignment by the column-by-column walk of ungapped_planes<TRACK>,
            // on the allele's block grid -- haplotype records by id, the read's planes from the item record.
            auto span_needs_dp = [&](const u32 (&w)[NR], const int mm, const int xm, const int score) -> bool {
                const int MA = P.match_bonus << MLST_P_SHIFT, PD = (pen_def << MLST_P_SHIFT) + 1;
                int cur = P0, best = P0, cs = 0, last = 0, blen = 0, overlap = 0, i1 = 0; bool first = true;
                #pragma unroll
                for (int t = 0; t <= NB; t++) {
                    if (hb[t + 1] == hb[t]) continue;                      // uniform
                    const u32 k = (u32)t + (odd16 >> 4);
                    const u32 id = (w[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu;
                    const uint4 rc = *reinterpret_cast<GP<const uint4>::G*>(recs + hb[t] + id);
                    const uint4 sp = *reinterpret_cast<const uint4*>(x + X::PL + 4 * t);
                    const u32 lm = rc.w >= 32u ? 0xFFFFFFFFu : ((1u << rc.w) - 1u), valid = x[X::VR + t] & lm;
                    if (!valid) continue;
                    u32 M = ((rc.x ^ sp.x) | (rc.y ^ sp.y) | sp.z | rc.z) & valid;
                    const int lo_c = __ffs((int)valid) - 1, nv = __popc(valid);
                    if (first) { cs = last = 32 * t + lo_c; first = false; }
                    overlap += nv; i1 = 32 * t + lo_c + nv;
                    const u32 special = sp.w | rc.z;
                    while (M) {
                        const int bit = __ffs((int)M) - 1; M &= M - 1;
                        const int i = 32 * t + bit;
                        cur += (i - last) * MA;
                        if (cur > best) { best = cur; blen = i - cs; }
                        int dec = PD;
                        if ((special >> bit) & 1u) dec = ((((rc.z >> bit) & 1u) ? P.n_penalty : (int)s_pen[32 * t - (int)((u32)diag & 31u) + bit]) << MLST_P_SHIFT) + 1;
                        cur -= dec;
                        if (cur <= P0) { cur = P0; cs = i + 1; }
                        last = i + 1;
                    }
                }
                cur += (i1 - last) * MA;
                if (cur > best) blen = i1 - cs;
                const int clipped = overlap - blen;          // overlap columns the ungapped alignment left out (gap_trigger)
                return mm > P.trig && score >= floor_n && clipped >= P.clip && 2 * (mm - xm) >= clipped;
            };
            const bool single = (flags & XF_SINGLE) != 0;
            bool slow = P.trig < 0 || n_alleles > acc_cap;
            if (!slow) {
                // A read with one work item (nearly all): what metamlst.py:101-130 would add for each record is noted beside the
                // summaries (LDS) while the alleles are composed, and added once the item's counts say that no pair needs the
                // aligned span or the banded SW and that the read has not exactly one record (Q1: column 15 would then be XO).
                // No result word is stored, none is read back; the additions leave in one burst at the end of the item (inside
                // the pass every turn waited for the additions of the turn before: vmcnt counts them with the loads, in order).
                u16* const s_acc = s_pend;
                const bool n_ok = n >= P.min_read_len;
                u32 nacc = 0;
                for (u32 a = tid; a < n_alleles && !XSKIP(2u); a += nthr) {
                    u32 W[NR];
                    #pragma unroll
                    for (int j = 0; j < NR; j++) { W[j] = Wn[j]; Wn[j] = Wn2[j]; Wn2[j] = Wn3[j]; Wn3[j] = Wn4[j]; }
                    load_ids(a + 4 * nthr, Wn4);            // the ids four turns ahead: a load under this kernel's traffic takes several turns of composition
                    asm volatile("" ::: "memory");
                    int best, mm; compose(W, best, mm);
                    const int score = best >> MLST_P_SHIFT, xm = 255 - (best & 0xFF);
                    const bool cand = score >= floor_n && score > 0;
                    bool track = cand && mm > P.trig;
                    if (track) track = span_needs_dp(W, mm, xm, score);      // now: the pair goes to the banded SW (rare)
                    const bool rec = cand && !track, ok = rec && n_ok && score >= P.minscore && xm <= P.max_xm;
                    nrec += (u32)__popcll(__ballot(rec)); nacc += (u32)__popcll(__ballot(ok)); ntrack += (u32)__popcll(__ballot(track));
                    s_acc[a] = ok ? (u16)score : (u16)0;
                }
                // (a lane's copy of a count holds the ballots of the turns in which the lane was active: lane 0 has them all)
                u32 tot_rec = __builtin_amdgcn_readfirstlane(nrec), tot_acc = __builtin_amdgcn_readfirstlane(nacc), tot_track = __builtin_amdgcn_readfirstlane(ntrack);
                if (nwv > 1) {
                    if ((tid & 63) == 0) { s_cnt[tid >> 6][0] = tot_rec; s_cnt[tid >> 6][1] = tot_acc; s_cnt[tid >> 6][2] = tot_track; }
                    __syncthreads();
                    tot_rec = tot_acc = tot_track = 0;
                    for (int w = 0; w < nwv; w++) { tot_rec += s_cnt[w][0]; tot_acc += s_cnt[w][1]; tot_track += s_cnt[w][2]; }
                    __syncthreads();
                }
                // (a read with several work items: this item alone settles Q1 unless it holds exactly one record -- see extend_pairs_body)
                if (tot_track == 0 && !(P.quirk && tot_rec == 1)) {      // block-uniform
                    // The device performs ~95 G additions a second (the rate that binds k_pileup too), and two per accepted record --
                    // sum and count -- were 0.42 of this kernel's 0.46 ms however few instructions it issued: count and sum travel in
                    // ONE 64-bit addition to acc64 (count << 40 | sum: a submission holds < 2^24 items, checked on the host).
                    if (tot_acc) for (u32 a = tid; a < n_alleles; a += nthr) {      // (each thread reads back what it wrote)
                        const u32 sc = s_acc[a];
                        if (sc && !XSKIP(4u)) {
                            if (packed_acc) atomicAdd(&E.acc64[a_begin + a], (1ull << 40) | (u64)sc);
                            else { atomicAdd((u64*)&E.sum_score[a_begin + a], (u64)sc); atomicAdd(&E.n_hits[a_begin + a], 1u); }
                        }
                    }
                    if (tid == 0) {
                        if (tot_rec) atomicAdd(&E.ret_nrec[ret], tot_rec);      // per-read record count (Q1)
                        c_tot += tot_rec; c_ign += tot_rec - tot_acc;
                        E.item_state[ii] = (u8)((single ? IS_SINGLE : 0) | IS_DONE | (tot_acc ? IS_ACC : 0));
                    }
                    done_fast = true;
                } else { slow = true; nrec = 0; ntrack = 0; }
            }
            if (slow) {
                load_ids((u32)tid, Wn);
                for (u32 a = tid; a < n_alleles; a += nthr) {
                    u32 W[NR];
                    #pragma unroll
                    for (int j = 0; j < NR; j++) W[j] = Wn[j];
                    load_ids(a + nthr, Wn);
                    asm volatile("" ::: "memory");
                    int best, mm; compose(W, best, mm);
                    const int score = best >> MLST_P_SHIFT, xm = 255 - (best & 0xFF), xo = 127 - ((best >> 8) & 0x7F);
                    const bool track = P.trig >= 0 && mm > P.trig && score >= floor_n;      // rare: the policy needs the aligned span
                    const u64 tm = __ballot(track);
                    if (tm) { ntrack += (u32)__popcll(tm); if (track) E.res[res_off + a] = R_TRACK; }
                    emit_pair(a, score, xm, xo, P.trig < 0, !track);
                }
                ntrack = __builtin_amdgcn_readfirstlane(ntrack);      // (lane 0 took part in every turn: its copy is the wave's count)
                if (nwv > 1) {
                    if ((tid & 63) == 0) s_cnt[tid >> 6][2] = ntrack;
                    __syncthreads();
                    ntrack = 0; for (int w = 0; w < nwv; w++) ntrack += s_cnt[w][2];
                }
                if (ntrack) { stage_old(); pairs_loop(true); }      // block-uniform
            }
        }                                         // (items of other loci: k_extend_pairs)
        XT(3);
        if (!done_fast && use_hap) {              // block-uniform
        if ((tid & 63) == 0) { s_cnt[tid >> 6][0] = nrec; s_cnt[tid >> 6][1] = ndp; }
        __syncthreads();
        u32 tot_rec = 0, tot_dp = 0;
        for (int w = 0; w < nwv; w++) { tot_rec += s_cnt[w][0]; tot_dp += s_cnt[w][1]; }
        if (tid == 0 && tot_rec) atomicAdd(&E.ret_nrec[ret], tot_rec);   // per-read record count (Q1)
        // Fused accumulation (metamlst.py:101-130) when everything about this read is known here:
        // it has a single work item and no pair is waiting for the banded SW.
        if (res_ok && tot_dp == 0 && ((flags & XF_SINGLE) || !P.quirk || tot_rec != 1)) {
            bool use_xo = P.quirk && tot_rec == 1;
            u32 acc = 0, ign = 0;
            for (u32 a = tid; a < n_alleles; a += nthr) {
                u32 r = E.res[res_off + a];
                const bool rec = r & R_REC, ok = rec && accept_rec(P, r, n, use_xo);
                if (ok) {
                    if (packed_acc) atomicAdd(&E.acc64[a_begin + a], (1ull << 40) | (u64)(r & 0x3FF));
                    else { atomicAdd((u64*)&E.sum_score[a_begin + a], (u64)(r & 0x3FF)); atomicAdd(&E.n_hits[a_begin + a], 1u); }
                }
                acc += (u32)__popcll(__ballot(ok)); ign += (u32)__popcll(__ballot(rec && !ok));
            }
            __syncthreads();
            if ((tid & 63) == 0) { s_cnt[tid >> 6][0] = acc; s_cnt[tid >> 6][1] = ign; }
            __syncthreads();
            if (tid == 0) {
                u32 A = 0, I = 0;
                for (int w = 0; w < nwv; w++) { A += s_cnt[w][0]; I += s_cnt[w][1]; }
                c_tot += tot_rec; c_ign += I;
                const u8 state = (u8)((flags & XF_SINGLE) ? IS_SINGLE : 0);
                E.item_state[ii] = (u8)(state | IS_DONE | (A ? IS_ACC : 0));
            }
        }
        }
        XT(4);
        if (!parked) park();
        lds_barrier();                            // (LDS only: the additions of this item are still on their way)
        pp = pp_next; cur ^= 1;
        XT(5);
    }
    XT_FLUSH;
    if (tid == 0) {
        if (c_tot) atomicAdd(&E.ctr->cnt[MLST_CNT_TOTAL_RECORDS], c_tot);
        if (c_ign) atomicAdd(&E.ctr->cnt[MLST_CNT_IGNORED], c_ign);
    }
}

// Two instantiations: reads up to 160 bases (five 32-base blocks) and up to MLST_MAX_READ_LEN.
__global__ __launch_bounds__(1024) void k_extend_160(const EngineDev* __restrict__ Ep, KParams P, u32 lds_recs, u32 acc_cap, const u32* __restrict__ xrec, u64 cap_xrec) { extend_body<5>(Ep, P, lds_recs, acc_cap, xrec, cap_xrec); }
// (at most 512 threads per workgroup for the long instantiation: with 1,024 the compiler has 128 registers per lane and spilled five of them -- 48 B of scratch)
__global__ __launch_bounds__(512) void k_extend_320(const EngineDev* __restrict__ Ep, KParams P, u32 lds_recs, u32 acc_cap, const u32* __restrict__ xrec, u64 cap_xrec) { extend_body<RW / 2>(Ep, P, lds_recs, acc_cap, xrec, cap_xrec); }

template <int CTRL, int ROW_MASK = 0xF> __device__ inline int dpp_i32(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xF, false);
}
// inclusive prefix maximum inside each 32-lane half of the wave (identity NEGP); max is exact, so any order of
// combination gives the same integers as the serial recurrence
__device__ inline int seg32_prefix_max(int v) {
    int t;
    t = dpp_i32<DPP_ROW_SHR(1)>(NEGP, v); v = t > v ? t : v;
    t = dpp_i32<DPP_ROW_SHR(2)>(NEGP, v); v = t > v ? t : v;
    t = dpp_i32<DPP_ROW_SHR(4)>(NEGP, v); v = t > v ? t : v;
    t = dpp_i32<DPP_ROW_SHR(8)>(NEGP, v); v = t > v ? t : v;
    t = dpp_i32<DPP_ROW_BCAST15, 0xA>(NEGP, v); v = t > v ? t : v;      // rows 1 and 3 take the total of rows 0 and 2
    return v;
}

// ------------------------------------------------------------------ K4: banded affine Smith-Waterman (one lane per pair)
// Band arrays live in registers (fully unrolled to 2*MAX_W+1 with a uniform guard).  Same recurrence and tie
// rules as oracle align_banded.  TB != nullptr additionally stores the traceback byte of every cell.
template <bool TRACE>
__device__ inline int banded(const EngineDev& E, const KParams& P, const ItemDev& it, const LocusDev& L, u32 a_local,
                             int n, const u8* s_pentab, u8* TB, int& bi, int& bb) {
    const int W = P.band_w, BW = 2 * W + 1, G = P.gbar, d = it.diag;
    const int m = (int)E.allele_len[L.a_begin + a_local];
    const int OPEN = P.open_p, EXT = P.ext_p, MA = P.match_bonus << MLST_P_SHIFT;
    auto rb = E.ret_bases.g() + (u64)it.ret * RW;
    auto rq = E.ret_quals.g() + (u64)it.ret * RQ;
    int Hp[2 * MAX_W + 2], Fp[2 * MAX_W + 2];
    #pragma unroll
    for (int b = 0; b < 2 * MAX_W + 2; b++) { Hp[b] = P0; Fp[b] = NEGP; }
    int best = P0; bi = -1; bb = -1;
    // allele bases of the band live in a sliding window: three arena words = allele bases [16q, 16q+48)
    int jb = d - W;                               // allele position of band cell 0 in row 0
    int q = jb >> 4;                              // floor(jb / 16)
    u32 w0 = arena_word(E, L, q, a_local), w1 = arena_word(E, L, q + 1, a_local), w2 = arena_word(E, L, q + 2, a_local);
    u8 qb_next = rq[it.strand ? n - 1 : 0];
    u32 rword = 0; int rword_idx = -1;
    for (int i = 0; i < n; i++, jb++) {
        int s = it.strand ? n - 1 - i : i;
        u8 qb = qb_next;
        if (i + 1 < n) qb_next = rq[it.strand ? n - 2 - i : i + 1];      // prefetch the next row's quality byte
        if ((s >> 4) != rword_idx) { rword_idx = s >> 4; rword = rb[rword_idx]; }
        bool rn = (qb & 0x80) != 0;
        u32 rbase = (rword >> (2 * (s & 15))) & 3u; if (it.strand) rbase ^= 3u;
        int pen = rn ? P.n_penalty : (int)s_pentab[qb & 0x7F];
        bool gap_ok = (i >= G && i < n - G);
        if ((jb >> 4) != q) { q++; w0 = w1; w1 = w2; w2 = arena_word(E, L, q + 2, a_local); }
        int sh = 2 * (jb - 16 * q);               // 0..30
        u64 lo64 = (u64)w0 | ((u64)w1 << 32);
        u64 rowbits = sh ? ((lo64 >> sh) | ((u64)w2 << (64 - sh))) : lo64;   // bases jb .. jb+31, 2 bits each
        int Hleft = P0, Eleft = NEGP;
        #pragma unroll
        for (int b = 0; b < 2 * MAX_W + 1; b++) {
            if (b < BW) {
                int j = jb + b;
                bool exists = (j >= 0 && j < m);
                u32 ab = (u32)(rowbits >> (2 * b)) & 3u;
                bool an = L.has_n && exists && allele_is_n(E, L, j, a_local);
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

// Score-only banded SW, row-parallel: 32 lanes per (item, allele) pair, lane b owns band cell b of the current row.
//   diagonal predecessor (i-1, j-1) = the lane's own previous H      (band coordinates shift by one per row)
//   F predecessor        (i-1, j)   = lane b+1's previous H / F      (one shuffle)
//   E within the row: E_b = max_{b'<b} (H'_{b'} - OPEN - EXT*(b-1-b')) with H' = max(fresh, diag, F): an exclusive
//   prefix-max scan over the lanes (a gap is never opened from a cell that was itself reached through E, because
//   OPEN > EXT), so every H equals the sequential recurrence of banded<>() / oracle align_banded.
// Two pairs per wave; XM / XO come out of the packed maximum, no traceback.
__global__ __launch_bounds__(256) void k_banded(const EngineDev* __restrict__ Ep, KParams P) {
    const EngineDev& E = *Ep;     // device-resident descriptor: fields are scalar-loaded on demand
    __shared__ u8 s_pentab[128];
    // per pair (8 pairs per block) and oriented read position: mismatch penalty | N flag << 8 | base << 9, so that a DP
    // row costs one LDS read (fetched a row ahead) and no global-memory round trip
    __shared__ u16 s_info[8][RQ];
    // the allele words the band can touch (32 words of 16 bases from the word of band cell 0 in row 0: 512 bases, the longest
    // read + the band + the slack of the three-word window), fetched once per pair, one word per lane.  Fetched inside the
    // row loop (a word every 16 rows, requested three words ahead) each of them was waited for at the top of the NEXT row --
    // the compiler cannot count loads across the loop's branches and writes vmcnt(0): ten exposed round trips per pair.
    __shared__ u32 s_aw[8][32];
    if (threadIdx.x < 128) s_pentab[threadIdx.x] = E.pen_tab[threadIdx.x];
    __syncthreads();
    const int b = threadIdx.x & 31, gl = threadIdx.x >> 5;
    const u64 grp = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 5, ngrp = ((u64)gridDim.x * blockDim.x) >> 5;
    const u64 begin = E.ctr->dp_done, end = E.ctr->n_dp < E.cap_dp ? E.ctr->n_dp : E.cap_dp;
    const int W = P.band_w, BW = 2 * W + 1, G = P.gbar;
    const int OPEN = P.open_p, EXT = P.ext_p, MA = P.match_bonus << MLST_P_SHIFT;
    const u64 n_iter = (end - begin + ngrp - 1) / ngrp;
    for (u64 itx = 0; itx < n_iter; itx++) {
        u64 k = begin + itx * ngrp + grp;
        bool live = k < end;
        u64 e = live ? E.dp_list[k] : 0; u64 ii = e >> 20; u32 a_local = (u32)(e & 0xFFFFFu);
        ItemDev it; it.res_off = 0; it.ret = 0; it.locus = 0; it.diag = 0; it.strand = 0; it.votes = 0;
        if (live) it = E.items[ii];
        const LocusDev L = E.loci[it.locus];
        int n = live ? (int)(E.ret_len[it.ret] & 0x7FFFu) : 0;
        int nmax = n; { int o = __shfl_xor(nmax, 32); nmax = o > nmax ? o : nmax; }    // both halves run the same trip count
        const int d = it.diag;
        const int m = live ? (int)E.allele_len[L.a_begin + a_local] : 0;
        __syncthreads();                      // previous pair's rows are no longer read (trip count is block-uniform)
        {
            auto gq = E.ret_quals.g() + (u64)it.ret * RQ;
            auto gb = E.ret_bases.g() + (u64)it.ret * RW;
            for (int i = b; i < n; i += 32) {
                int s = it.strand ? n - 1 - i : i;
                u8 qb = gq[s]; u32 isn = qb >> 7;
                u32 rbase = (gb[s >> 4] >> (2 * (s & 15))) & 3u; if (it.strand) rbase ^= 3u;
                u32 pen = isn ? (u32)P.n_penalty : (u32)s_pentab[qb & 0x7F];
                s_info[gl][i] = (u16)(pen | (isn << 8) | (rbase << 9));
            }
            s_aw[gl][b] = arena_word(E, L, ((d - W) >> 4) + b, a_local);
        }
        __syncthreads();
        int Hp = P0, Fp = NEGP, best = P0;
        int jb = d - W, q = jb >> 4;
        const int q0 = q;
        // allele bases of the band: words q..q+2 are in use, q+3 is taken from the staged words at the next word crossing
        u32 w0 = s_aw[gl][0], w1 = s_aw[gl][1], w2 = s_aw[gl][2], w3 = s_aw[gl][3];
        u32 info_next = n > 0 ? (u32)s_info[gl][0] : 0u;
        for (int i = 0; i < nmax; i++, jb++) {
            bool row = i < n;
            u32 info = info_next;
            if (i + 1 < n) info_next = s_info[gl][i + 1];
            bool rn = (info >> 8) & 1u;
            u32 rbase = (info >> 9) & 3u;
            int pen = (int)(info & 0xFFu);
            bool gap_ok = row && (i >= G && i < n - G);
            if (row && (jb >> 4) != q) { q++; w0 = w1; w1 = w2; w2 = w3; w3 = s_aw[gl][(q + 3 - q0) & 31]; }
            int sh = 2 * (jb - 16 * q);
            u64 lo64 = (u64)w0 | ((u64)w1 << 32);
            u64 rowbits = sh ? ((lo64 >> sh) | ((u64)w2 << (64 - sh))) : lo64;
            int j = jb + b;
            bool exists = row && b < BW && j >= 0 && j < m;
            u32 ab = (u32)(rowbits >> (2 * b)) & 3u;
            bool an = L.has_n && exists && allele_is_n(E, L, j, a_local);
            int delta = (!rn && !an && rbase == ab) ? MA : -(((rn || an) ? P.n_penalty : pen) << MLST_P_SHIFT) - 1;
            int diag = Hp + delta;
            // neighbours through DPP (register-to-register lane moves), not ds_bpermute: a row is a serial chain of
            // ~8 lane exchanges, and the kernel is that chain times the read length.  Lane 31/63 of wave_shl is never
            // consumed (b < BW - 1 <= 30 below).
            int Hup = dpp_i32<DPP_WAVE_SHL1>(Hp, Hp), Fup = dpp_i32<DPP_WAVE_SHL1>(Fp, Fp);
            int f = NEGP;
            if (gap_ok && b < BW - 1) { int f1 = Hup - OPEN, f2 = Fup - EXT; f = f2 > f1 ? f2 : f1; }
            int h1 = P0; if (diag > h1) h1 = diag; if (f > h1) h1 = f;
            if (!exists) { h1 = P0; f = NEGP; }
            // exclusive prefix max of g = H' + EXT*b over the 32 lanes of the pair
            int g = h1 + EXT * b;
            int x = dpp_i32<DPP_WAVE_SHR1>(NEGP, g); if (b < 1) x = NEGP;
            x = seg32_prefix_max(x);
            int ee = (gap_ok && b > 0) ? x - OPEN - EXT * (b - 1) : NEGP;
            int h = h1; if (ee > h) h = ee;
            if (!exists) h = P0;
            if (row) { if (h > best) best = h; Hp = h; Fp = f; }
        }
        #pragma unroll
        for (int o = 16; o > 0; o >>= 1) { int y = __shfl_xor(best, o, 32); best = y > best ? y : best; }
        if (live && b == 0) {
            int score = best >> MLST_P_SHIFT, xm = 255 - (best & 0xFF), xo = 127 - ((best >> 8) & 0x7F);
            u32 r = pack_result(score, xm, xo) | R_USEDDP;
            if (score >= E.floor_tab[n] && score > 0) { r |= R_REC; atomicAdd(&E.ret_nrec[it.ret], 1u); }
            E.res[it.res_off + a_local] = r;
        }
    }
}

// ------------------------------------------------------------------ K5: accumulate (metamlst.py:101-130)
__global__ __launch_bounds__(256) void k_accumulate(const EngineDev* __restrict__ Ep, KParams P) {
    const EngineDev& E = *Ep;     // device-resident descriptor: fields are scalar-loaded on demand
    // items k_extend could not finish: reads with several work items, or pairs that went through the banded SW
    // (they are few: the states of 256 items are read at once and the open ones listed, instead of one dependent
    // state load per item and block)
    __shared__ u32 s_red[4][4]; __shared__ u32 s_list[256]; __shared__ u32 s_nlist;
    const int tid = threadIdx.x;
    // what k_extend added in packed form (count << 40 | sum per allele) goes on to the per-allele sums; zero again afterwards
    for (u64 a = (u64)blockIdx.x * 256 + tid; a < E.n_alleles; a += (u64)gridDim.x * 256) {
        const u64 v = E.acc64[a];
        if (v) { atomicAdd((u64*)&E.sum_score[a], v & ((1ull << 40) - 1)); atomicAdd(&E.n_hits[a], (u32)(v >> 40)); E.acc64[a] = 0; }
    }
    u64 c_tot = 0, c_ign = 0, c_dp = 0;
    const u64 begin = E.ctr->items_done, end = E.ctr->n_items < E.cap_items ? E.ctr->n_items : E.cap_items;
    for (u64 i0 = begin + (u64)blockIdx.x * 256; i0 < end; i0 += (u64)gridDim.x * 256) {
      if (tid == 0) s_nlist = 0;
      __syncthreads();
      if (i0 + tid < end && !(E.item_state[i0 + tid] & IS_DONE)) s_list[atomicAdd(&s_nlist, 1u)] = (u32)tid;
      __syncthreads();
      const u32 n_open = s_nlist;
      for (u32 q = 0; q < n_open; q++) {
        const u64 ii = i0 + s_list[q];
        u8 state = E.item_state[ii];
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
            tot++;
            if (accept_rec(P, r, n, use_xo)) {
                atomicAdd((u64*)&E.sum_score[L.a_begin + a], (u64)(r & 0x3FF));
                atomicAdd(&E.n_hits[L.a_begin + a], 1u);
                acc++;
            } else ign++;
        }
        tot = wave_sum_u32(tot); ign = wave_sum_u32(ign); acc = wave_sum_u32(acc); dp = wave_sum_u32(dp);
        __syncthreads();
        if ((tid & 63) == 0) { s_red[tid >> 6][0] = tot; s_red[tid >> 6][1] = ign; s_red[tid >> 6][2] = acc; s_red[tid >> 6][3] = dp; }
        __syncthreads();
        if (tid == 0) {
            u32 T = 0, I = 0, A = 0, D = 0;
            for (int w = 0; w < 4; w++) { T += s_red[w][0]; I += s_red[w][1]; A += s_red[w][2]; D += s_red[w][3]; }
            c_tot += T; c_ign += I; c_dp += D;
            E.item_state[ii] = (u8)(state | IS_DONE | (A ? IS_ACC : 0));
        }
      }
      __syncthreads();
    }
    if (tid == 0) {
        if (c_tot) atomicAdd(&E.ctr->cnt[MLST_CNT_TOTAL_RECORDS], c_tot);
        if (c_ign) atomicAdd(&E.ctr->cnt[MLST_CNT_IGNORED], c_ign);
        if (c_dp) atomicAdd(&E.ctr->cnt[MLST_CNT_DP_PAIRS], c_dp);
    }
}

// sequenceBank[locus][QNAME] = len(SEQ) (metamlst.py:127) and first-seen order (Q6): one lane per item, items of the
// same locus inside a wave are combined before touching the per-locus words.  The dictionary holds ONE length per
// (locus, QNAME) -- that of the last accepted record (Q3) -- so a read that matches both strands of a locus counts once,
// and of two mates that share a QNAME (paired) the second one's length replaces the first one's.
// The per-locus words are hot: an isolate's items sit on 7 of them, a metagenome's on ~140, and 2 x 121 k device atomics
// on those few addresses were the kernel (61 of 76 us on cfg3).  Items are first combined in an LDS table per workgroup
// (1024 items wide: one sweep of 128 workgroups covers cfg3), and each workgroup then touches every locus it saw once.
#define LOCUS_LT 1024
__global__ __launch_bounds__(1024) void k_locus(const EngineDev* __restrict__ Ep, int paired) {
    const EngineDev& E = *Ep;     // device-resident descriptor: fields are scalar-loaded on demand
    __shared__ u32 s_key[LOCUS_LT]; __shared__ u64 s_sum[LOCUS_LT]; __shared__ u64 s_first[LOCUS_LT];
    for (u32 i = threadIdx.x; i < LOCUS_LT; i += blockDim.x) { s_key[i] = 0xFFFFFFFFu; s_sum[i] = 0; s_first[i] = ~0ull; }
    __syncthreads();
    const u64 begin = E.ctr->items_done, end = E.ctr->n_items < E.cap_items ? E.ctr->n_items : E.cap_items;
    for (u64 i0 = begin + (u64)blockIdx.x * blockDim.x; i0 < end; i0 += (u64)gridDim.x * blockDim.x) {
        u64 ii = i0 + threadIdx.x;
        if (!(ii < end && (E.item_state[ii] & IS_ACC))) continue;
        ItemDev it = E.items[ii];
        bool counts = true; const u32 locus = it.locus; const u64 n = (u64)(E.ret_len[it.ret] & 0x7FFFu), ridx = E.ret_ridx[it.ret];
        // an earlier item of the same read on the same locus (the other strand) already stands for this QNAME
        const u64 first = E.ret_item0[it.ret];
        for (u64 j = first; j < ii; j++) if (E.items[j].locus == locus && (E.item_state[j] & IS_ACC)) counts = false;
        if (counts && paired && !(ridx & 1ull)) {      // first mate: superseded when the second mate has an accepted record here
            const u32 ms = E.ret_mate[it.ret];
            if (ms != 0xFFFFFFFFu) {
                const u64 m0 = E.ret_item0[ms], m1 = m0 + E.ret_nitems[ms];
                for (u64 j = m0; j < m1; j++) if (E.items[j].locus == locus && (E.item_state[j] & IS_ACC)) counts = false;
            }
        }
        u32 slot = (locus * 0x9E3779B1u) >> 22;
        bool placed = false;
        for (int probe = 0; probe < 8 && !placed; probe++) {
            const u32 prev = atomicCAS(&s_key[slot], 0xFFFFFFFFu, locus);
            if (prev == 0xFFFFFFFFu || prev == locus) {
                if (counts) atomicAdd((unsigned long long*)&s_sum[slot], (unsigned long long)n);
                atomicMin((unsigned long long*)&s_first[slot], (unsigned long long)ridx);
                placed = true;
            } else slot = (slot + 1) & (LOCUS_LT - 1);
        }
        if (!placed) { if (counts) atomicAdd(&E.locus_len[locus], n); atomicMin(&E.locus_first[locus], ridx); }     // table crowded
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < LOCUS_LT; i += blockDim.x) {
        const u32 locus = s_key[i];
        if (locus == 0xFFFFFFFFu) continue;
        if (s_sum[i]) atomicAdd(&E.locus_len[locus], s_sum[i]);
        atomicMin(&E.locus_first[locus], s_first[i]);
    }
}

__global__ void k_advance(Counters* c, u64 n_reads) {
    u64 ni = c->n_items, nd = c->n_dp; c->ret_done = c->n_ret;
    c->cnt[MLST_CNT_CANDIDATES] += c->n_cand;
    c->cnt[MLST_CNT_READS_SEEN] += n_reads;
    c->items_done = ni; c->dp_done = nd; c->n_cand = 0;
    c->n_res = 0;        // the result rows of a submission have been consumed by k_accumulate: the arena is reused by the next one
    for (int q = 0; q < EXT_Q; q++) { c->ext_q[q][0] = 0; c->ext_q2[q][0] = 0; }
}

// ------------------------------------------------------------------ K6: pileup against the chosen allele of each locus
// One lane per item.  Ungapped pairs are piled up here; pairs that trigger the banded SW go to k_pileup_dp.
__device__ inline void pile_base(const EngineDev& E, const KParams& P, const ItemDev& it, int n, int i, int j,
                                 u32* counts, u64 colbase) {
    int s = it.strand ? n - 1 - i : i;
    u8 qb = E.ret_quals[(u64)it.ret * RQ + s];
    if ((qb & 0x80) || (int)(qb & 0x7F) < P.minqual) return;
    u32 b = src_base(E.ret_bases.g() + (u64)it.ret * RW, s); if (it.strand) b ^= 3u;
    atomicAdd(&counts[(colbase + (u64)j) * 4 + b], 1u);
}

// Two phases per batch of 64 items.
//  (1) one LANE per item: the item's descriptors, its single ungapped alignment against the chosen allele of its locus
//      (value-identical to ungapped_planes<., true, .>), the gap-trigger policy and the tag filter.  The read stays in
//      source order; for the reverse strand it is the allele window that is fetched mirrored (words in descending order,
//      bit-reversed, complemented), and the Kadane pass walks the mismatches from the top bit down.
//  (2) one WAVE per item that passed: its aligned columns are piled up 64 at a time.
// One-wave workgroups, 128 bytes of LDS, device atomics into the counters -- on purpose: a version that kept the counters of a
// locus in LDS (one 1024-thread workgroup per locus and slice of the item list, 49 KB of LDS) was no faster alone (116-121 us)
// and cost the pipelined step 0.24 ms: a CU that holds one of its workgroups cannot take a k_route_probe workgroup (147 KB).
// History of the 207 us this kernel took on cfg3: LDS-staged reads with 64 redundant lanes, bit planes, batched loads, one
// lane per item -- all 190-215 us, until builds that stop after one stage each showed 84 us in the Kadane loops of the 1.4 %
// of items that disagree with the chosen allele in dozens of columns (see `hopeless` below).
// CAP = the depth-capped pile-up (mlst_set_depth_cap; policy MLST_DEPTH_CAP, oracle: orc_pileup_capped).  A record's key is
// read index << 1 | strand; thr[column] is a key, and a record is seen by a column iff its key <= thr[column].
//   mode & 3 == 1: `counts` is ONE word per column and receives the number of records with key <= thr[column] that SPAN it
//                  (every record the aligner reports, tag filter or not -- that is what sits in a pysam pile-up column);
//   mode & 3 == 2: the pile-up proper (four words per column), restricted to the records a column sees;
//   mode & 4     : this pass does not append to the banded-SW list (an earlier pass of the same search did).
// The host finds thr[column] = the cap-th smallest key of the records that span the column by a bitwise search over
// mode-1 passes (pile_all).
template <int NB, bool CAP>
__device__ __forceinline__ void pileup_body(const EngineDev* __restrict__ Ep, const KParams& P, const int* __restrict__ locus_chosen,
                                            const u64* __restrict__ locus_colbase, u32* __restrict__ counts, u64* __restrict__ pl_list,
                                            const u64* __restrict__ thr, u32 mode) {
    const EngineDev& E = *Ep;     // device-resident descriptor: fields are scalar-loaded on demand
    __shared__ u8 s_pentab[128];
    const int lane = threadIdx.x;
    int pmin = P.n_penalty;                                            // smallest penalty a mismatch can cost (score bound below)
    for (int i = lane; i < 128; i += 64) { const u8 v = E.pen_tab[i]; s_pentab[i] = v; pmin = (int)v < pmin ? (int)v : pmin; }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(pmin, o); pmin = y < pmin ? y : pmin; }
    __syncthreads();
    const u64 end = E.ctr->n_items < E.cap_items ? E.ctr->n_items : E.cap_items;
    constexpr int NI = (32 * NB + 63) / 64;                            // sweeps of 64 columns that cover one read
    const int MA = P.match_bonus << MLST_P_SHIFT;
    for (u64 base = blockIdx.x; base < end; base += (u64)gridDim.x * 64) {       // the item of lane k: base + k * gridDim.x
        const u64 my = base + (u64)lane * gridDim.x;
        u32 f_ret = 0, f_strand = 0; int f_diag = 0, f_n = 0, f_bs = 0, f_be = 0; u64 f_col = 0, f_key = 0; bool f_pile = false;
        if (my < end) {
            const ItemDev it = E.items[my];
            const int ca = locus_chosen[it.locus];
            if (ca >= 0) {
                const LocusDev L = E.loci[it.locus];
                const u32 lw = E.ret_len[it.ret];
                const int n = (int)(lw & 0x7FFFu), d = it.diag; const bool read_has_n = (lw & 0x8000u) != 0, rev = it.strand != 0;
                const int m = (int)E.allele_len[ca]; const int floor_n = E.floor_tab[n];
                f_ret = it.ret; f_strand = it.strand; f_diag = d; f_n = n; f_col = locus_colbase[it.locus];
                if (CAP) f_key = (E.ret_ridx[it.ret] << 1) | (u64)(it.strand != 0);
                const u32 a = (u32)ca - L.a_begin;
                auto rb = E.ret_bases.g() + (u64)it.ret * RW;
                auto rq = E.ret_quals.g() + (u64)it.ret * RQ;
                // allele window: block w of the read (source positions 32w..32w+31) meets allele bits pbit - 32w .. +31 going
                // down (reverse) or pbit + 32w .. +31 going up (forward); NB + 1 consecutive words cover all blocks
                const int pbit = rev ? n + d - 32 : d;
                const int q0 = pbit >> 5, r = pbit & 31, qlow = rev ? q0 - (NB - 1) : q0;
                auto pbase = E.planes.g() + L.plane_off; auto nbase = E.nmask.g() + L.nmask_off;
                u32 Wl[NB + 1], Wh[NB + 1], Wn[NB + 1];
                #pragma unroll
                for (int t = 0; t <= NB; t++) {
                    const int q = qlow + t; const u32 qc = (u32)(q < 0 ? 0 : (q >= (int)L.pblocks ? (int)L.pblocks - 1 : q));
                    Wl[t] = pbase[(u64)(2u * qc) * L.n_pad + a]; Wh[t] = pbase[(u64)(2u * qc + 1u) * L.n_pad + a];
                    Wn[t] = 0;
                    if (L.has_n) Wn[t] = nbase[(u64)qc * L.n_pad + a];
                }
                u32 xw[2 * NB];
                #pragma unroll
                for (int w = 0; w < 2 * NB; w++) xw[w] = rb[w];
                tie_all<NB + 1>(Wl); tie_all<NB + 1>(Wh); tie_all<2 * NB>(xw);
                // columns of the overlap, in source coordinates
                const int i0 = d < 0 ? -d : 0, i1 = (m - d) < n ? (m - d) : n;
                const int slo = rev ? n - i1 : i0, shi = rev ? n - i0 : i1;
                u32 M[NB], AN[NB]; int mm = 0;
                #pragma unroll
                for (int w = 0; w < NB; w++) {
                    const u32 rl = compress16(xw[2 * w]) | (compress16(xw[2 * w + 1]) << 16);
                    const u32 rh = compress16(xw[2 * w] >> 1) | (compress16(xw[2 * w + 1] >> 1) << 16);
                    const int t = rev ? NB - 1 - w : w;
                    u32 al = __builtin_amdgcn_alignbit(Wl[t + 1], Wl[t], r), ah = __builtin_amdgcn_alignbit(Wh[t + 1], Wh[t], r);
                    u32 an = __builtin_amdgcn_alignbit(Wn[t + 1], Wn[t], r);
                    if (rev) { al = ~__brev(al); ah = ~__brev(ah); an = __brev(an); }
                    AN[w] = an;
                    u32 rn = 0;
                    if (read_has_n) {                                   // rare: N positions of the read, from the quality rows
                        for (int k = 0; k < 32; k++) { const int sp = 32 * w + k; if (sp < n && (rq[sp] & 0x80)) rn |= 1u << k; }
                    }
                    int lo_i = slo - 32 * w; lo_i = lo_i < 0 ? 0 : (lo_i > 32 ? 32 : lo_i);
                    int hi_i = shi - 32 * w; hi_i = hi_i < 0 ? 0 : (hi_i > 32 ? 32 : hi_i);
                    const u32 vm = (hi_i >= 32 ? 0xFFFFFFFFu : ((1u << hi_i) - 1u)) & ~(lo_i >= 32 ? 0xFFFFFFFFu : ((1u << lo_i) - 1u));
                    M[w] = ((al ^ rl) | (ah ^ rh) | rn | an) & vm;
                    mm += __popc(M[w]);
                }
                // A few items in a hundred (1,644 of 121 k on cfg3) disagree with the chosen allele in dozens of columns; one such
                // lane kept its whole wave in the loops below for a hundred rounds, each waiting for a quality byte (84 of the
                // kernel's 190 us).  For a lane with many mismatches an upper bound of its score comes first -- the same Kadane
                // pass with every mismatch at the smallest penalty of the table, no memory access -- and a lane whose bound
                // stays under the floor is dropped here: below the floor an item is neither piled up nor sent to the banded SW.
                bool hopeless = false;
                if (mm > 8 && P.trig >= 0) {
                    const int ma = P.match_bonus;
                    int c2 = 0, b2 = 0, last2 = slo;
                    #pragma unroll
                    for (int w = 0; w < NB; w++) {
                        u32 Mw = M[w];
                        while (Mw) {
                            const int bit = __ffs(Mw) - 1; Mw &= Mw - 1;
                            c2 += (32 * w + bit - last2) * ma; b2 = c2 > b2 ? c2 : b2;
                            c2 -= pmin; c2 = c2 < 0 ? 0 : c2; last2 = 32 * w + bit + 1;
                        }
                    }
                    c2 += (shi - last2) * ma; b2 = c2 > b2 ? c2 : b2;
                    hopeless = b2 < floor_n;
                }
                // The quality byte under a mismatch decides its penalty.  Fetched inside the Kadane loops, every mismatch of every
                // word is a round trip of its own (the loops run for the worst lane of the wave); so the bytes under the first two
                // mismatches of each word -- in walking order -- are requested here, all in one batch, and only a third mismatch
                // in the same 32 columns still waits for its own.
                u32 qa[NB], qb2[NB];
                #pragma unroll
                for (int w = 0; w < NB; w++) {
                    qa[w] = 0; qb2[w] = 0;
                    u32 Mw = hopeless ? 0u : M[w];
                    if (Mw) {
                        const int b1 = rev ? 31 - __clz(Mw) : __ffs(Mw) - 1;
                        qa[w] = rq[32 * w + b1];
                        Mw &= ~(1u << b1);
                        if (Mw) { const int b2 = rev ? 31 - __clz(Mw) : __ffs(Mw) - 1; qb2[w] = rq[32 * w + b2]; }
                    }
                }
                tie_all<NB>(qa); tie_all<NB>(qb2);
                int cur = P0, best = P0, cs = i0, last = i0, blen = 0, bend = i0;
                if (i1 > i0 && !hopeless) {
                    // one mismatch at oriented position i (qb: the read's quality byte there; an_bit: the allele has N there)
                    auto step = [&](int i, u32 qb, u32 an_bit) {
                        cur += (i - last) * MA;
                        if (cur > best) { best = cur; blen = i - cs; bend = i; }
                        int pen = P.n_penalty;
                        if (!an_bit && !(qb & 0x80u)) pen = (int)s_pentab[qb & 0x7Fu];
                        cur -= (pen << MLST_P_SHIFT) + 1;
                        if (cur <= P0) { cur = P0; cs = i + 1; }
                        last = i + 1;
                    };
                    if (!rev) {
                        #pragma unroll
                        for (int w = 0; w < NB; w++) {
                            u32 Mw = M[w]; int seen = 0;
                            while (Mw) {
                                const int bit = __ffs(Mw) - 1; Mw &= Mw - 1;
                                const u32 qb = seen == 0 ? qa[w] : (seen == 1 ? qb2[w] : (u32)rq[32 * w + bit]); seen++;
                                step(32 * w + bit, qb, (AN[w] >> bit) & 1u);
                            }
                        }
                    } else {
                        #pragma unroll
                        for (int w = NB - 1; w >= 0; w--) {
                            u32 Mw = M[w]; int seen = 0;
                            while (Mw) {
                                const int bit = 31 - __clz(Mw); Mw &= ~(1u << bit);
                                const u32 qb = seen == 0 ? qa[w] : (seen == 1 ? qb2[w] : (u32)rq[32 * w + bit]); seen++;
                                step(n - 1 - (32 * w + bit), qb, (AN[w] >> bit) & 1u);
                            }
                        }
                    }
                    cur += (i1 - last) * MA;
                    if (cur > best) { best = cur; blen = i1 - cs; bend = i1; }
                }
                const int be = bend, bs = bend - blen;
                const int score = best >> MLST_P_SHIFT, xm = 255 - (best & 0xFF);
                if (hopeless) { }
                else if (gap_trigger(P, mm, xm, score, floor_n, m, n, d, bs, be)) {
                    if (!CAP || !(mode & 4u)) { const u64 slot = atomicAdd(&E.ctr->n_pl_dp, 1ull); pl_list[slot] = my; }
                }
                else if (CAP && (mode & 3u) == 1u) {                     // the record exists: it counts towards the depth of its span
                    if (!(score < floor_n || score <= 0)) { f_pile = be > bs; f_bs = bs; f_be = be; }
                }
                else if (!(score < floor_n || score <= 0 || score < P.minscore || xm > P.max_xm)) {      // BAM_tagFilter AS, XM
                    f_pile = be > bs; f_bs = bs; f_be = be;
                }
            }
        }
        // ---- phase 2: the wave piles up the columns of one item after the other; the rows of the next item are requested
        // before the atomics of this one go out (on gfx9 the loads behind an atomic wait for its acknowledgement)
        u64 todo = __ballot(f_pile);
        u32 qv[NI], wv[NI]; int c_n = 0, c_bs = 0, c_be = 0, c_d = 0; u32 c_strand = 0; u64 c_col = 0, c_key = 0;
        auto fetch = [&](int k) {
            if (CAP) c_key = (u64)(u32)__builtin_amdgcn_readlane((int)(u32)f_key, k) | ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(f_key >> 32), k) << 32);
            c_n = __builtin_amdgcn_readlane(f_n, k); c_bs = __builtin_amdgcn_readlane(f_bs, k); c_be = __builtin_amdgcn_readlane(f_be, k);
            c_d = __builtin_amdgcn_readlane(f_diag, k); c_strand = (u32)__builtin_amdgcn_readlane((int)f_strand, k);
            c_col = (u64)(u32)__builtin_amdgcn_readlane((int)(u32)f_col, k) | ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(f_col >> 32), k) << 32);
            const u32 ret = (u32)__builtin_amdgcn_readlane((int)f_ret, k);
            auto rb = E.ret_bases.g() + (u64)ret * RW; auto rq = E.ret_quals.g() + (u64)ret * RQ;
            #pragma unroll
            for (int t = 0; t < NI; t++) {
                const int i = c_bs + 64 * t + lane; qv[t] = 0x80u; wv[t] = 0;
                if (i < c_be) { const int sp = c_strand ? c_n - 1 - i : i; qv[t] = rq[sp]; wv[t] = rb[sp >> 4]; }
            }
        };
        if (todo) { const int k = __ffsll((long long)todo) - 1; todo &= todo - 1; fetch(k); }
        else continue;
        for (;;) {                                                      // block-uniform
            u32 q0v[NI], w0v[NI];
            #pragma unroll
            for (int t = 0; t < NI; t++) { q0v[t] = qv[t]; w0v[t] = wv[t]; }
            const int p_n = c_n, p_bs = c_bs, p_be = c_be, p_d = c_d; const u32 p_strand = c_strand; const u64 p_col = c_col, p_key = c_key;
            const bool more = todo != 0;
            if (more) { const int k = __ffsll((long long)todo) - 1; todo &= todo - 1; fetch(k); }
            #pragma unroll
            for (int t = 0; t < NI; t++) {
                const int i = p_bs + 64 * t + lane;
                if (i >= p_be) continue;
                if (CAP) {
                    if (p_key > thr[p_col + (u64)(i + p_d)]) continue;                    // this column does not see the record
                    if ((mode & 3u) == 1u) { atomicAdd(&counts[p_col + (u64)(i + p_d)], 1u); continue; }
                }
                const u32 qb = q0v[t];
                if ((qb & 0x80u) || (int)(qb & 0x7Fu) < P.minqual) continue;
                const int sp = p_strand ? p_n - 1 - i : i;
                u32 b = (w0v[t] >> (2 * (sp & 15))) & 3u; if (p_strand) b ^= 3u;
                atomicAdd(&counts[(p_col + (u64)(i + p_d)) * 4 + b], 1u);
            }
            if (!more) break;
        }
    }
}
// reads up to 160 bases / up to MLST_MAX_READ_LEN (the host knows the longest row width submitted for the sample)
__global__ __launch_bounds__(64) void k_pileup_160(const EngineDev* __restrict__ Ep, KParams P, const int* __restrict__ locus_chosen,
                                                   const u64* __restrict__ locus_colbase, u32* __restrict__ counts, u64* __restrict__ pl_list) {
    pileup_body<5, false>(Ep, P, locus_chosen, locus_colbase, counts, pl_list, nullptr, 0u);
}
__global__ __launch_bounds__(64) void k_pileup_320(const EngineDev* __restrict__ Ep, KParams P, const int* __restrict__ locus_chosen,
                                                   const u64* __restrict__ locus_colbase, u32* __restrict__ counts, u64* __restrict__ pl_list) {
    pileup_body<RW / 2, false>(Ep, P, locus_chosen, locus_colbase, counts, pl_list, nullptr, 0u);
}
__global__ __launch_bounds__(64) void k_pileup_cap_160(const EngineDev* __restrict__ Ep, KParams P, const int* __restrict__ locus_chosen,
                                                       const u64* __restrict__ locus_colbase, u32* __restrict__ counts, u64* __restrict__ pl_list,
                                                       const u64* __restrict__ thr, u32 mode) {
    pileup_body<5, true>(Ep, P, locus_chosen, locus_colbase, counts, pl_list, thr, mode);
}
__global__ __launch_bounds__(64) void k_pileup_cap_320(const EngineDev* __restrict__ Ep, KParams P, const int* __restrict__ locus_chosen,
                                                       const u64* __restrict__ locus_colbase, u32* __restrict__ counts, u64* __restrict__ pl_list,
                                                       const u64* __restrict__ thr, u32 mode) {
    pileup_body<RW / 2, true>(Ep, P, locus_chosen, locus_colbase, counts, pl_list, thr, mode);
}

// thr / mode: the depth-capped pile-up (see pileup_body); thr == NULL: the plain one
__global__ __launch_bounds__(64) void k_pileup_dp(const EngineDev* __restrict__ Ep, KParams P, const int* __restrict__ locus_chosen,
                                                  const u64* __restrict__ locus_colbase, u32* __restrict__ counts,
                                                  const u64* __restrict__ pl_list, u8* __restrict__ tb_scratch,
                                                  const u64* __restrict__ thr, u32 mode) {
    const EngineDev& E = *Ep;     // device-resident descriptor: fields are scalar-loaded on demand
    __shared__ u8 s_pentab[128];
    for (int i = threadIdx.x; i < 128; i += 64) s_pentab[i] = E.pen_tab[i];
    __syncthreads();
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
        int best = banded<true>(E, P, it, L, a, n, s_pentab, TB, bi, bb);
        int score = best >> MLST_P_SHIFT, xm = 255 - (best & 0xFF);
        if (score < E.floor_tab[n] || score <= 0) continue;                                   // no record
        const bool tags_ok = !(score < P.minscore || xm > P.max_xm);                          // BAM_tagFilter AS, XM
        const bool count_pass = thr && (mode & 3u) == 1u;
        if (!tags_ok && !count_pass) continue;
        const int W = P.band_w, BW = 2 * W + 1;
        int i = bi, b = bb, state = 0;
        u64 colbase = locus_colbase[it.locus];
        const u64 key = thr ? ((E.ret_ridx[it.ret] << 1) | (u64)(it.strand != 0)) : 0ull;
        if (count_pass) {      // the record's span: from the last aligned column of the walk back to the first one, deletions inside included
            int j1 = -1, j0 = 0;
            while (i >= 0 && b >= 0 && b < BW) {
                u8 t = TB[i * BWMAX + b];
                if (state == 0) {
                    int src = t & 3;
                    if (src == 0) break;
                    if (src == 1) { const int j = i + it.diag - W + b; if (j1 < 0) j1 = j; j0 = j; i--; }
                    else if (src == 2) state = 1; else state = 2;
                } else if (state == 1) { int ext = t & 4; b--; state = ext ? 1 : 0; }
                else { int ext = t & 8; i--; b++; state = ext ? 2 : 0; }
            }
            for (int j = j0; j1 >= 0 && j <= j1; j++) if (key <= thr[colbase + (u64)j]) atomicAdd(&counts[colbase + (u64)j], 1u);
            continue;
        }
        while (i >= 0 && b >= 0 && b < BW) {
            u8 t = TB[i * BWMAX + b];
            if (state == 0) {
                int src = t & 3;
                if (src == 0) break;
                if (src == 1) { const int j = i + it.diag - W + b; if (!thr || key <= thr[colbase + (u64)j]) pile_base(E, P, it, n, i, j, counts, colbase); i--; }
                else if (src == 2) state = 1; else state = 2;
            } else if (state == 1) { int ext = t & 4; b--; state = ext ? 1 : 0; }
            else { int ext = t & 8; i--; b++; state = ext ? 2 : 0; }
        }
    }
}

// The bitwise search of the depth-capped pile-up, one step per launch: lo / hi bracket the smallest key t of a column with
// #(records that span it, key <= t) >= cap; thr = the probe of the next counting pass; cnt = what the last pass counted.
//   step 0: lo = 0, hi = CAP_KEY_MAX, thr = the first probe;  step 1: narrow by cnt, next probe;
//   step 2: narrow by cnt, then thr = lo (a column that fewer than cap records span ends at CAP_KEY_MAX: it sees them all)
#define CAP_KEY_BITS 41                 /* read index < 2^40, one strand bit */
#define CAP_KEY_MAX ((1ull << CAP_KEY_BITS) - 1ull)
__global__ __launch_bounds__(256) void k_cap_search(u64* __restrict__ lo, u64* __restrict__ hi, u64* __restrict__ thr, u32* __restrict__ cnt,
                                                    u64 n_cols, u32 cap, int step) {
    for (u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x; c < n_cols; c += (u64)gridDim.x * blockDim.x) {
        u64 l = 0, h = CAP_KEY_MAX;
        if (step) {
            l = lo[c]; h = hi[c];
            if (l < h) { const u64 mid = l + ((h - l) >> 1); if (cnt[c] >= cap) h = mid; else l = mid + 1; }
        }
        lo[c] = l; hi[c] = h; cnt[c] = 0;
        thr[c] = step == 2 ? l : (l < h ? l + ((h - l) >> 1) : l);
    }
}
// majority base per column (cmseq reference_free_consensus [NOT IN TREE]: ties alphabetical, < mincov -> none_char)
// Zero-fill as a kernel of its own.  The launch sequences that are replayed as hipGraphs hold kernel nodes only: a replayed
// graph with memset / memcpy nodes faulted just past the end of the counts buffer once foreign copies and fills (torch
// tensors moved by the same process) had run between its capture and its replay (profiles/check_batch.py, three samples on
// one engine); with MLST_GRAPHS=0 the same sequence was clean.
__global__ __launch_bounds__(256) void k_zero(u32* __restrict__ p, u64 n_words) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (u64)gridDim.x * blockDim.x) p[i] = 0u;
}
__global__ __launch_bounds__(256) void k_consensus(const u32* __restrict__ counts, u64 n_cols, u32 mincov, u8 none_char, u8* __restrict__ out) {
    for (u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x; c < n_cols; c += (u64)gridDim.x * blockDim.x) {
        const uint4 v = reinterpret_cast<const uint4*>(counts)[c];
        u32 best = v.x; u8 ch = 'A';
        if (v.y > best) { best = v.y; ch = 'C'; }
        if (v.z > best) { best = v.z; ch = 'G'; }
        if (v.w > best) { best = v.w; ch = 'T'; }
        out[c] = (v.x + v.y + v.z + v.w) >= mincov ? ch : none_char;
    }
}

// ---- compact column layout for the multi-GPU exchange of the pileup counts (mlst_typing_choose_pileup_compact).
// After the first exchange every rank holds the same statistics and chooses the same alleles, so every rank derives the
// same layout: the loci WITH a chosen allele get their slots (of the locus' longest allele, as in the fixed layout) one
// after the other; on cfg3 that is 140 of 1,050 loci, 1.1 MB of counts to all-reduce instead of 8.4 MB.  The caller fixes
// the capacity before it knows the need (a collective's size is a host decision): when the slots do not fit, nothing is
// piled up (chosen_pl = -1 everywhere), info says so and the caller repeats the phase with a larger buffer.
// One block; info[0] = columns needed, info[1] = 1 when they exceed cap_cols.
__global__ __launch_bounds__(1024) void k_layout_compact(const int* __restrict__ chosen, const u64* __restrict__ fixed_colbase, u32 n_loci, u64 cap_cols,
                                                         u64* __restrict__ colbase_c, int* __restrict__ chosen_pl, u64* __restrict__ info) {
    __shared__ u32 s_w[16]; __shared__ u64 s_run;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (u32 l0 = 0; l0 < n_loci; l0 += 1024) {
        const u32 l = l0 + (u32)tid;
        const u32 w = (l < n_loci && chosen[l] >= 0) ? (u32)(fixed_colbase[l + 1] - fixed_colbase[l]) : 0u;
        u32 inc = w;
        for (int o = 1; o < 64; o <<= 1) { const u32 y = __shfl_up(inc, o); if (lane >= o) inc += y; }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        u32 before = 0, all = 0;
        for (int v = 0; v < 16; v++) { const u32 t = s_w[v]; if (v < wv) before += t; all += t; }
        const u64 run = s_run;
        if (l < n_loci) colbase_c[l] = run + before + inc - w;
        __syncthreads();
        if (tid == 0) s_run = run + all;
        __syncthreads();
    }
    const u64 need = s_run; const bool over = need > cap_cols;
    for (u32 l = (u32)tid; l < n_loci; l += 1024) chosen_pl[l] = over ? -1 : chosen[l];
    if (tid == 0) { info[0] = need; info[1] = over ? 1ull : 0ull; }
}
// majority letters of the compact counts, written in the FIXED layout mlst_typing_fetch hands out (one block per locus);
// a locus without a slot gets what k_consensus makes of zero counts
__global__ __launch_bounds__(256) void k_consensus_expand(const u32* __restrict__ counts, const int* __restrict__ chosen_pl, const u64* __restrict__ colbase_c,
                                                          const u64* __restrict__ fixed_colbase, u32 mincov, u8 none_char, u8* __restrict__ out) {
    const u32 l = blockIdx.x; const u64 fb = fixed_colbase[l], w = fixed_colbase[l + 1] - fb, cb = colbase_c[l];
    const bool have = chosen_pl[l] >= 0;
    for (u64 c = threadIdx.x; c < w; c += 256) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (have) v = reinterpret_cast<const uint4*>(counts)[cb + c];
        u32 best = v.x; u8 ch = 'A';
        if (v.y > best) { best = v.y; ch = 'C'; }
        if (v.z > best) { best = v.z; ch = 'G'; }
        if (v.w > best) { best = v.w; ch = 'T'; }
        out[fb + c] = (v.x + v.y + v.z + v.w) >= mincov ? ch : none_char;
    }
}

// ------------------------------------------------------------------ pileup of ready-made alignments (SAM / BAM input)
// One thread per record: CIGAR walk over the chosen contig's columns.  A base counts when the record's true tags pass
// AS >= minscore and XM <= max_xm (cmseq BAM_tagFilter), its Phred is >= minqual and it is A/C/G/T
// (metaMLST_functions.py:255-259 -> cmseq get_base_stats [NOT IN TREE]; policy constants as in k_pileup).
__global__ __launch_bounds__(256) void k_pileup_aln(u64 n_rec, const u32* __restrict__ rec_allele, const int* __restrict__ rec_pos,
                                                    const int* __restrict__ rec_as, const int* __restrict__ rec_xm,
                                                    const u64* __restrict__ cig_off, const u32* __restrict__ cig,
                                                    const u64* __restrict__ seq_off, const u8* __restrict__ seq, const u8* __restrict__ qual,
                                                    const int* __restrict__ allele_slot /* n_alleles: column base or -1 */,
                                                    const u64* __restrict__ aoff, int minscore, int max_xm, int minqual,
                                                    u32* __restrict__ counts) {
    for (u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x; k < n_rec; k += (u64)gridDim.x * blockDim.x) {
        const u32 a = rec_allele[k];
        const int base = allele_slot[a];
        if (base < 0 || rec_as[k] < minscore || rec_xm[k] > max_xm) continue;
        const long long alen = (long long)(aoff[a + 1] - aoff[a]);
        const u64 s0 = seq_off[k], sn = seq_off[k + 1] - s0;
        long long r = rec_pos[k]; u64 q = 0;
        for (u64 c = cig_off[k]; c < cig_off[k + 1]; c++) {
            const u32 ln = cig[c] >> 4, op = cig[c] & 15u;
            if (op == 0 || op == 7 || op == 8) {                       // M = X
                for (u32 t = 0; t < ln && q + t < sn; t++) {
                    const u8 ch = seq[s0 + q + t] & 0xDF;              // upper case
                    const int b = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : -1;
                    const long long col = r + t;
                    if (b >= 0 && (int)qual[s0 + q + t] >= minqual && col >= 0 && col < alen)
                        atomicAdd(&counts[((u64)base + (u64)col) * 4 + b], 1u);
                }
                r += ln; q += ln;
            } else if (op == 1 || op == 4) q += ln;                    // I S
            else if (op == 2 || op == 3) r += ln;                      // D N
        }
    }
}

// ------------------------------------------------------------------ allele choice on the device (metamlst.py:133-151, 244)
// Python's round(float(p) / float(q), 1) as an exact integer number of tenths.  round() of a float is the correctly
// rounded decimal of the binary double (ties of the DOUBLE go to the even digit).  The double d = fl(p/q) lies within
// one ulp of p/q, and p/q is at least 1/(20q) away from every rounding boundary (2m+1)/20 it does not hit exactly, so
// away from rational ties the exact integer arithmetic decides; on a rational tie the sign of 20*d - (2m+1), which one
// fma gives exactly, says on which side of the boundary the double fell.
__host__ __device__ inline long long round_tenths(long long p, u32 q) {
    const long long p10 = 10 * p, qq = (long long)q;
    long long m = p10 / qq; if ((p10 % qq) < 0) m -= 1;                 // floor division
    const long long rem = p10 - m * qq;                                  // 0 <= rem < q
    if (2 * rem < qq) return m;
    if (2 * rem > qq) return m + 1;
    const double d = (double)p / (double)q;
    const double sgn = fma(d, 20.0, -(double)(2 * m + 1));
    if (sgn > 0.0) return m + 1;
    if (sgn < 0.0) return m;
    return (m & 1) ? m + 1 : m;                                          // the double IS the tie: even digit
}
// One block per locus: chosen[l] = the allele with the highest rounded penalised average, ties to the lowest allele
// number (Q5), or -1 when the locus has no accepted record.
__global__ __launch_bounds__(256) void k_choose(const EngineDev* __restrict__ Ep, const int* __restrict__ allele_no, int penalty,
                                                int* __restrict__ chosen) {
    const EngineDev& E = *Ep;
    __shared__ u32 s_max[4]; __shared__ long long s_r[4]; __shared__ int s_no[4]; __shared__ int s_a[4];
    const u32 l = blockIdx.x; const LocusDev L = E.loci[l];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (l == 0 && threadIdx.x == 0) E.ctr->n_pl_dp = 0;      // the pile-up that follows starts its banded-SW list here (one launch less than a fill)
    u32 mx = 0;
    for (u32 k = threadIdx.x; k < L.n_alleles; k += 256) { u32 v = E.n_hits[L.a_begin + k]; mx = v > mx ? v : mx; }
    for (int o = 32; o > 0; o >>= 1) { u32 y = __shfl_xor(mx, o); mx = y > mx ? y : mx; }
    if (lane == 0) s_max[wv] = mx;
    __syncthreads();
    mx = s_max[0]; for (int k = 1; k < 4; k++) mx = s_max[k] > mx ? s_max[k] : mx;
    long long br = 0; int bno = 0x7FFFFFFF, ba = -1;
    for (u32 k = threadIdx.x; k < L.n_alleles; k += 256) {
        const u32 a = L.a_begin + k; const u32 nh = E.n_hits[a];
        if (!nh) continue;
        const long long local = E.sum_score[a] - (long long)(mx - nh) * (long long)penalty;     // metamlst.py:146-147
        const long long r = round_tenths(local, nh); const int no = allele_no[a];
        if (ba < 0 || r > br || (r == br && no < bno)) { br = r; bno = no; ba = (int)a; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        long long r2 = __shfl_xor(br, o); int no2 = __shfl_xor(bno, o); int a2 = __shfl_xor(ba, o);
        if (a2 >= 0 && (ba < 0 || r2 > br || (r2 == br && no2 < bno))) { br = r2; bno = no2; ba = a2; }
    }
    if (lane == 0) { s_r[wv] = br; s_no[wv] = bno; s_a[wv] = ba; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) if (s_a[k] >= 0 && (ba < 0 || s_r[k] > br || (s_r[k] == br && s_no[k] < bno))) { br = s_r[k]; bno = s_no[k]; ba = s_a[k]; }
        chosen[l] = ba;
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
__global__ void k_export(const EngineDev* __restrict__ Ep, long long* d_sum, long long* d_min) {
    const EngineDev& E = *Ep;     // device-resident descriptor: fields are scalar-loaded on demand
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
__global__ void k_import(const EngineDev* __restrict__ Ep, const long long* d_sum, const long long* d_min) {
    const EngineDev& E = *Ep;     // device-resident descriptor: fields are scalar-loaded on demand
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
    u32* d_arena = nullptr; u32* d_planes = nullptr; u32* d_nmask = nullptr; u16* d_allele_len = nullptr; u32* d_allele_locus = nullptr;
    LocusDev* d_loci = nullptr; uint4* d_sieve = nullptr; u32* d_bitmap = nullptr; u32* d_gbitmap = nullptr; u64* d_keys = nullptr; u32* d_vals = nullptr; u32* d_posts = nullptr;
    int* d_floor = nullptr; u8* d_pen = nullptr; u8* d_ascii = nullptr; u64* d_aoff = nullptr;
    HapRec* d_hap_rec = nullptr; u32* d_hap_blk = nullptr; u32* d_hap_id = nullptr; u64 bytes_hap = 0, n_hap_rec = 0; u32 hap_win_max[2] = {0, 0}, hap_loci = 0;
    u32 ext_lds_recs[2] = {0, 0};                // k_extend_160 / _320: haplotype summaries (16 B each) the launch keeps in LDS
    u32 ext_acc_cap = 0;                         // alleles per locus for which k_extend keeps the pending additions of an item in LDS (2 B each)
    u8* h_qc = nullptr; u8* d_qc = nullptr; u64 cap_hostq = 0, cap_qc = 0;      // mlst_submit_packed_host: the candidates' Phred rows (pinned staging, device)
    u64* d_acc64 = nullptr;                      // k_extend's additions of a submission, count << 40 | sum of scores per allele (k_accumulate hands them on)
    u32* d_xrec[2] = {nullptr, nullptr}; u64 cap_xrec[2] = {0, 0};      // item records of k_ext_prep (one per work item of a submission), per instantiation
    u64 bytes_arena = 0, bytes_sieve = 0, bytes_table = 0; double bitmap_fill = 0.0;
    EngineDev E; EngineDev* d_E = nullptr;      // host copy and its device-resident twin
    bool have_ref = false, have_state = false;
    // batch scratch
    u32* d_cand = nullptr; u64 cap_cand = 0;
    // CU-routed sieve (K1c): filter slices (reference) and the per-submission arena
    int sieve_kind = 0; u32 sieve_chain = 0; u64 n_keys = 0;
    u32* d_rfilter = nullptr; u32* d_rt_arena = nullptr; u64 cap_rt_arena = 0; u32* d_rt_counts = nullptr; u32* d_rt_emitted = nullptr; u64 cap_rt_emitted = 0;
    u32 rt_prod = 0, rt_cap = 0, rt_tiles_max = 0, rt_nw = 16, rt_pf = 2; u64 rt_slice = 0;
    u64* d_rt_parked = nullptr; u64 cap_rt_parked = 0;      // entries that passed the LDS filter (k_route_probe -> k_route_verify)
    u64* d_rt_trace = nullptr; bool rt_trace_on = false; const void* rt_last_packed = nullptr;      // mlst_get_route_trace
    std::vector<void*> dbg_pads;                 // mlst_debug_route_realloc: allocations kept to move the arena elsewhere
    u32* d_bin_flags = nullptr; u64 cap_bin_flags = 0;      // candidate flag per read of the current submission
    u8* d_in_bases = nullptr; u8* d_in_quals = nullptr; u64* d_in_off = nullptr; u64 cap_in_bytes = 0, cap_in_reads = 0;
    u32* d_packed = nullptr; u8* d_qrows = nullptr; u16* d_lens = nullptr; u64 cap_packed_words = 0, cap_qrow_bytes = 0, cap_lens = 0;
    u64 reads_seen = 0;
    u32 max_wpr = 0;                             // widest read rows submitted since the last reset (picks the k_pileup instantiation)
    int ext_threads = 64, ext_blocks = 5120;     // k_extend launch shape (set in mlst_load_reference)
    int extp_threads = 256, extp_blocks = 1792; u32 pair_loci = 0;      // k_extend_pairs: launch shape, loci it takes
    int sieve_g_blocks = 256 * 5;                // k_sieve_q<.,false> grid (MLST_SIEVE_BLOCKS overrides it)
    u8* d_fq_text = nullptr;                    // the text buffer of the chunk being parsed: one of d_fq_slot[] (not owned)
    u32* d_fq_blk = nullptr; u64 cap_fq_blk = 0;
    // BGZF input: compressed bytes + block descriptors on the device; the partial record at the end of a chunk (carry)
    u8* d_bgzf = nullptr; u64 cap_bgzf = 0; void* d_bgzf_blk = nullptr; u64 cap_bgzf_blk = 0; u8* d_fq_carry = nullptr; u64 cap_fq_carry = 0, fq_carry_len = 0;
    u64* d_fq_lines = nullptr; u64 cap_fq_lines = 0; u64* d_fq_soff = nullptr; u64* d_fq_qoff = nullptr; u64 cap_fq_reads = 0; u64* d_fq_meta = nullptr;
    // pileup scratch
    int* d_locus_chosen = nullptr; u64* d_locus_colbase = nullptr; u64* d_pl_list = nullptr; u8* d_tb = nullptr;
    u32* d_itok = nullptr; u32* d_intok = nullptr; u64 cap_itok_blocks = 0; int inflate_mode = 0;      // two-kernel inflate: token buffer (INFL_TOK_CAP words per block of a pass), tokens per block
    // BGZF input in three stages on three streams (mlst_submit_fastq_bgzf, round 5): the copy of piece k (copy_stream), the inflate
    // of piece k (infl_stream) and the parse + pass 1 of piece k - 1 (the engine's stream) run side by side.  Two slots of
    // compressed bytes / block descriptors / error words used in turn; `bz_pend` is the piece whose text is being inflated
    // (or has been) and has not been parsed yet.
    struct BzSlot { void* d_blk = nullptr; void* h_blk = nullptr; u64 cap_blk = 0; hipEvent_t ev_copied = nullptr, ev_inflated = nullptr; u32* d_err = nullptr; u32* h_err = nullptr; };
    BzSlot bz[2]; int bz_slot = 0, bz_mode = -1; hipStream_t infl_stream = nullptr;
    // the compressed bytes of a CHUNK (one call of mlst_submit_fastq_bgzf), two buffers used in turn: the copy is queued in BZ_SUB parts
    // before the chunk's block headers are walked (a cache miss per block: 8 ms for 49,152 blocks, now beside the copy), a piece's
    // inflate waits for the part that holds its last byte; ev_used: the last inflate that read the buffer
    enum { BZ_SUB = 8 };
    struct BzChunk { u8* d = nullptr; u64 cap = 0; hipEvent_t ev[BZ_SUB] = {}; u64 upto[BZ_SUB] = {}; int n_ev = 0; hipEvent_t ev_used = nullptr; bool used = false; };
    BzChunk bzc[2]; int bz_chunk = 0;
    struct { bool on = false, counted = false; int slot = 0, tslot = 0, paired = 0; u64 text_bytes = 0; } bz_pend;
    u32 depth_cap = 0; u64* d_capbuf = nullptr; u64 cap_capcols = 0;      // depth-capped pile-up (mlst_set_depth_cap): lo, hi, thr (u64 each) and cnt (u32) per column
    u32* d_counts = nullptr; u64 cap_counts = 0;
    // device-side typing (mlst_typing_enqueue / mlst_typing_fetch): fixed column layout, one slot of loc_maxlen columns per locus
    int* d_allele_no = nullptr; int* d_auto_chosen = nullptr; u64* d_fixed_colbase = nullptr; std::vector<u64> fixed_colbase; u64 fixed_cols = 0;
    u32* d_auto_counts = nullptr; u8* d_auto_letters = nullptr; bool auto_pending = false;
    // results of the typing tail in pinned memory, two slots written in turn: the host can queue the engine's next step
    // (mlst_typing_wait, then submit + mlst_typing_enqueue) before it copies the results of the step just finished out of
    // theirs (mlst_typing_fetch_waited)
    u8* h_tstats[2] = {nullptr, nullptr}; u8* h_tauto[2] = {nullptr, nullptr}; int t_slot = 0, t_last = 0, t_ready = -1;
    u64* d_compact_colbase = nullptr; int* d_compact_chosen = nullptr; u64* d_compact_info = nullptr; u64 off_compact_info = 0; bool compact_pending = false;      // mlst_typing_choose_pileup_compact
    // host-to-device copies of the FASTQ entries: a copy stream, two text buffers used in turn, the event that says a
    // buffer's last chunk has been packed (h2d_overlapped)
    hipStream_t copy_stream = nullptr; hipEvent_t stage_done = nullptr;
    u8* d_fq_slot[2] = {nullptr, nullptr}; u64 cap_fq_slot[2] = {0, 0}; hipEvent_t ev_packed[2] = {nullptr, nullptr}; int fq_slot = 0;
    u32* d_fq_nl[2] = {nullptr, nullptr};      // newlines per FQ_BLOCK bytes of a text slot, counted by the inflate (k_inflate_ptr) for the parser
    int cu_split = 1, cu_part = 0;              // mlst_set_cu_partition: the engine's own stream runs on CUs [part, part + 1) * n_cu / split
    hipStream_t own_stream = nullptr;           // the stream created by mlst_create (h->stream may be a caller's stream: mlst_set_stream)
    u32* d_dist = nullptr; u8* d_query = nullptr; u64 cap_dist = 0, cap_query = 0;
    // one contiguous device block [sum_score | locus_len | Counters | n_hits | pad][locus_first] with a pinned mirror
    u8* d_stats = nullptr; u8* h_stats = nullptr; u64 stats_bytes = 0, stats_zero_bytes = 0;
    u64 off_sum = 0, off_len = 0, off_ctr = 0, off_hits = 0, off_first = 0;
    u8* h_pin = nullptr; u64 cap_pin = 0;          // pinned staging for pileup tables / counts
    // profiling
    bool profiling = false;                     // HIP events around the kernel groups (mlst_set_profiling(1))
    bool window = false;                        // in-kernel sieve window only (mlst_set_profiling(1 or 2))
    // the launch sequence of one submission / one typing tail, captured once per argument set and replayed (hipGraph)
    struct GraphSlot { hipGraphExec_t exec = nullptr; std::vector<u64> sig; };
    GraphSlot g_submit, g_typing; bool use_graphs = true;
    std::vector<EvPair> events;
    std::vector<hipEvent_t> ev_pool;
    double k_ms[16] = {0}; u64 k_n[16] = {0};      // see mlst_get_kernel_time
    double wall_khz = 100000.0;                  // wall_clock64 rate (hipDeviceAttributeWallClockRate)
};

static std::string g_create_err;

static int fail(mlst_handle* h, int code, const char* fmt, ...) {
    char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (h) h->err = buf; else g_create_err = buf;
    return code;
}
#define HIPCHK(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(h, MLST_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); } while (0)

static int bz_flush(mlst_handle* h);      // BGZF input: the piece still being inflated is parsed and submitted (defined with mlst_submit_fastq_bgzf)
static void bz_free(mlst_handle* h);
template <typename T> static hipError_t dmalloc(T** p, u64 n) { return hipMalloc((void**)p, (n ? n : 1) * sizeof(T)); }
template <typename T> static hipError_t dmalloc(GP<T>* p, u64 n) { return hipMalloc((void**)&p->p, (n ? n : 1) * sizeof(T)); }

static hipEvent_t ev_get(mlst_handle* h) {
    if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
    // timing events only: without the system-scope release a record carries by default (the write-back of what the kernel
    // before it left in the L2s would otherwise be counted into that kernel's interval)
    hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) hipEventCreate(&e); return e;
}
// Event profiling (mlst_set_profiling): the start event of a group is recorded behind an empty kernel.  Recorded on a
// stream that has been idle (the first group of a submission), the interval also held the wake-up of the queue: 40-55 us
// on k_route once no buffer fill ran in front of it any more, against 12 us before -- rocprofv3's kernel durations do
// not include it, and bench.py's rooflines are meant to be the same numbers (profiles/round3/README.md).
__global__ void k_nop() {}
struct Prof {
    mlst_handle* h; int which; hipEvent_t a = nullptr, b = nullptr;
    Prof(mlst_handle* h_, int w) : h(h_), which(w) {
        if (h->profiling) { a = ev_get(h); b = ev_get(h); hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, h->stream); hipEventRecord(a, h->stream); }
    }
    ~Prof() { if (h->profiling) { hipEventRecord(b, h->stream); h->events.push_back({a, b, which}); } }
};
static void drain_events(mlst_handle* h) {
    if (h->events.empty()) return;
    hipStreamSynchronize(h->stream);
    for (auto& e : h->events) { float ms = 0; hipEventElapsedTime(&ms, e.a, e.b); h->k_ms[e.which] += ms; h->k_n[e.which]++; h->ev_pool.push_back(e.a); h->ev_pool.push_back(e.b); }
    h->events.clear();
}

// hipGraph replay of a fixed launch sequence.  graph_enter returns 1 when a cached graph with the same signature was
// launched (nothing left to do), 2 when stream capture has begun (the caller issues its launches, then graph_leave
// instantiates and launches the graph), 0 when the caller should launch directly (profiling with events, a caller's
// stream, graphs disabled or capture unavailable).
static int graph_enter(mlst_handle* h, mlst_handle::GraphSlot& g, const std::vector<u64>& sig) {
    if (!h->use_graphs || h->profiling || h->stream != h->own_stream) return 0;
    if (g.exec && g.sig == sig) return hipGraphLaunch(g.exec, h->stream) == hipSuccess ? 1 : 0;
    if (g.exec) { hipGraphExecDestroy(g.exec); g.exec = nullptr; }
    // capture + instantiate cost more than one direct submission: a graph is built only for an argument set that
    // comes a second time in a row (the first call launches directly and remembers the arguments)
    if (g.sig != sig) { g.sig = sig; return 0; }
    if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { h->use_graphs = false; return 0; }
    return 2;
}
static int graph_leave(mlst_handle* h, mlst_handle::GraphSlot& g) {
    hipGraph_t graph = nullptr;
    if (hipStreamEndCapture(h->stream, &graph) != hipSuccess || !graph) { h->use_graphs = false; return fail(h, MLST_E_HIP, "hipStreamEndCapture failed"); }
    hipError_t e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) { g.exec = nullptr; h->use_graphs = false; return fail(h, MLST_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
    HIPCHK(h, hipGraphLaunch(g.exec, h->stream));
    return MLST_OK;
}

extern "C" void mlst_default_params(mlst_params* p) {
    memset(p, 0, sizeof *p);
    p->minscore = MLST_DEF_MINSCORE; p->max_xm = MLST_DEF_MAX_XM; p->min_read_len = MLST_DEF_MIN_READ_LEN;
    p->minqual = MLST_DEF_MINQUAL; p->mincov = MLST_DEF_MINCOV; p->match_bonus = MLST_DEF_MATCH_BONUS;
    p->mm_max = MLST_DEF_MM_MAX; p->mm_min = MLST_DEF_MM_MIN; p->n_penalty = MLST_DEF_N_PENALTY;
    p->gap_open = MLST_DEF_GAP_OPEN; p->gap_ext = MLST_DEF_GAP_EXT; p->gbar = MLST_DEF_GBAR; p->band_w = MLST_DEF_BAND_W;
    p->gap_trigger_mm = MLST_DEF_GAP_TRIGGER_MM; p->xm_field_quirk = MLST_DEF_XM_FIELD_QUIRK; p->gap_trigger_clip = MLST_DEF_GAP_TRIGGER_CLIP;
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
    if (prm.match_bonus < 1 || prm.match_bonus > 127) return fail(nullptr, MLST_E_INVALID, "match_bonus must be in 1..127");
    if (!prm.max_retained_reads) prm.max_retained_reads = 4ull << 20;
    if (!prm.max_items) prm.max_items = 8ull << 20;
    if (prm.max_items >= (1ull << 32)) return fail(nullptr, MLST_E_INVALID, "max_items must be below 2^32");
    if (!prm.max_pair_results) prm.max_pair_results = 256ull << 20;
    mlst_handle* h = new mlst_handle();
    h->device = device; h->prm = prm;
    if (hipSetDevice(device) != hipSuccess) { delete h; return fail(nullptr, MLST_E_HIP, "cannot initialise device %d", device); }
    if (hipStreamCreate(&h->stream) != hipSuccess) { delete h; return fail(nullptr, MLST_E_HIP, "cannot initialise device %d", device); }
    h->own_stream = h->stream;
    { const char* g = getenv("MLST_GRAPHS"); if (g && g[0] == '0') h->use_graphs = false; }
    { int khz = 0; if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) h->wall_khz = (double)khz; }
    KParams& k = h->kp;
    k.minscore = prm.minscore; k.max_xm = prm.max_xm; k.min_read_len = prm.min_read_len; k.minqual = prm.minqual;
    k.match_bonus = prm.match_bonus; k.n_penalty = prm.n_penalty;
    k.open_p = ((prm.gap_open + prm.gap_ext) << MLST_P_SHIFT) + (1 << 8); k.ext_p = prm.gap_ext << MLST_P_SHIFT;
    k.gbar = prm.gbar; k.band_w = prm.band_w; k.trig = prm.gap_trigger_mm; k.quirk = prm.xm_field_quirk; k.clip = prm.gap_trigger_clip;
    memset(&h->E, 0, sizeof h->E);
    *out = h;
    return MLST_OK;
}

static void free_ref(mlst_handle* h) {
    hipFree(h->d_arena); hipFree(h->d_planes); hipFree(h->d_nmask); hipFree(h->d_allele_len); hipFree(h->d_allele_locus); hipFree(h->d_loci);
    hipFree(h->d_sieve); hipFree(h->d_bitmap); h->d_bitmap = nullptr; hipFree(h->d_gbitmap); h->d_gbitmap = nullptr; hipFree(h->d_keys); hipFree(h->d_vals); hipFree(h->d_posts); hipFree(h->d_floor); hipFree(h->d_pen);
    hipFree(h->d_ascii); hipFree(h->d_aoff);
    hipFree(h->d_hap_rec); hipFree(h->d_hap_blk); hipFree(h->d_hap_id); h->d_hap_rec = nullptr; h->d_hap_blk = nullptr; h->d_hap_id = nullptr;
    hipFree(h->d_rfilter); h->d_rfilter = nullptr;
    hipFree(h->d_allele_no); hipFree(h->d_auto_chosen); hipFree(h->d_fixed_colbase); hipFree(h->d_auto_counts); hipFree(h->d_auto_letters);
    hipFree(h->d_compact_colbase); hipFree(h->d_compact_chosen); hipFree(h->d_compact_info); h->d_compact_colbase = nullptr; h->d_compact_chosen = nullptr; h->d_compact_info = nullptr;
    for (int k = 0; k < 2; k++) if (h->h_tauto[k]) { hipHostFree(h->h_tauto[k]); h->h_tauto[k] = nullptr; }
    h->d_allele_no = h->d_auto_chosen = nullptr; h->d_fixed_colbase = nullptr; h->d_auto_counts = nullptr; h->d_auto_letters = nullptr; h->auto_pending = false;
    h->d_arena = h->d_planes = h->d_nmask = nullptr; h->d_allele_len = nullptr; h->d_allele_locus = nullptr; h->d_loci = nullptr; h->d_sieve = nullptr;
    h->d_keys = nullptr; h->d_vals = h->d_posts = nullptr; h->d_floor = nullptr; h->d_pen = nullptr; h->d_ascii = nullptr; h->d_aoff = nullptr;
    h->have_ref = false;
}
static void free_state(mlst_handle* h) {
    EngineDev& E = h->E;
    hipFree(h->d_bin_flags); h->d_bin_flags = nullptr; h->cap_bin_flags = 0;
    hipFree(h->d_rt_arena); hipFree(h->d_rt_counts); hipFree(h->d_rt_emitted); hipFree(h->d_rt_trace); h->d_rt_trace = nullptr;
    hipFree(h->d_rt_parked); h->d_rt_parked = nullptr; h->cap_rt_parked = 0;
    for (void* q : h->dbg_pads) hipFree(q);
    h->dbg_pads.clear();
    h->d_rt_arena = nullptr; h->d_rt_counts = nullptr; h->d_rt_emitted = nullptr; h->cap_rt_arena = 0; h->cap_rt_emitted = 0; h->rt_prod = 0;
    hipFree(h->d_E); h->d_E = nullptr;
    hipFree(h->d_stats); h->d_stats = nullptr; if (h->h_stats) { hipHostFree(h->h_stats); h->h_stats = nullptr; }
    for (int k = 0; k < 2; k++) if (h->h_tstats[k]) { hipHostFree(h->h_tstats[k]); h->h_tstats[k] = nullptr; }
    hipFree(E.ret_bases); hipFree(E.ret_quals); hipFree(E.ret_len); hipFree(E.ret_ridx); hipFree(E.ret_nrec);
    hipFree(E.ret_mate); hipFree(E.ret_item0); hipFree(E.ret_nitems); hipFree(E.ret_cpos); E.ret_mate = nullptr; E.ret_item0 = nullptr; E.ret_nitems = nullptr; E.ret_cpos = nullptr;
    hipFree(E.items); hipFree(E.item_state); hipFree(E.res); hipFree(E.dp_list);
    for (int k = 0; k < 2; k++) { hipFree(h->d_xrec[k]); h->d_xrec[k] = nullptr; h->cap_xrec[k] = 0; }
    hipFree(h->d_acc64); h->d_acc64 = nullptr;
    hipFree(h->d_qc); h->d_qc = nullptr; h->cap_qc = 0; if (h->h_qc) { hipHostFree(h->h_qc); h->h_qc = nullptr; h->cap_hostq = 0; }
    hipFree(h->d_locus_chosen); hipFree(h->d_locus_colbase); hipFree(h->d_pl_list); hipFree(h->d_tb);
    hipFree(h->d_capbuf); h->d_capbuf = nullptr; h->cap_capcols = 0;
    hipFree(h->d_itok); hipFree(h->d_intok); h->d_itok = nullptr; h->d_intok = nullptr; h->cap_itok_blocks = 0;
    E.sum_score = nullptr; E.n_hits = nullptr; E.locus_len = E.locus_first = nullptr; E.ctr = nullptr; E.ret_bases = nullptr; E.ret_quals = nullptr;
    E.ret_len = nullptr; E.ret_ridx = nullptr; E.ret_nrec = nullptr; E.items = nullptr; E.item_state = nullptr; E.res = nullptr; E.dp_list = nullptr;
    h->d_locus_chosen = nullptr; h->d_locus_colbase = nullptr; h->d_pl_list = nullptr; h->d_tb = nullptr;
    h->have_state = false;
}

extern "C" void mlst_destroy(mlst_handle* h) {
    if (!h) return;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    bz_free(h);
    for (auto& e : h->events) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    for (auto& e : h->ev_pool) hipEventDestroy(e);
    free_ref(h); free_state(h);
    if (h->h_pin) hipHostFree(h->h_pin);
    for (int k = 0; k < 2; k++) { if (h->ev_packed[k]) hipEventDestroy(h->ev_packed[k]); }
    if (h->stage_done) hipEventDestroy(h->stage_done);
    if (h->copy_stream) hipStreamDestroy(h->copy_stream);
    hipFree(h->d_cand); hipFree(h->d_in_bases); hipFree(h->d_in_quals); hipFree(h->d_in_off);
    hipFree(h->d_fq_slot[0]); hipFree(h->d_fq_slot[1]); hipFree(h->d_fq_nl[0]); hipFree(h->d_fq_nl[1]); hipFree(h->d_fq_blk); hipFree(h->d_fq_lines); hipFree(h->d_fq_soff); hipFree(h->d_fq_qoff); hipFree(h->d_fq_meta);
    hipFree(h->d_bgzf); hipFree(h->d_bgzf_blk); hipFree(h->d_fq_carry); h->d_bgzf = nullptr; h->d_bgzf_blk = nullptr; h->d_fq_carry = nullptr; h->cap_bgzf = h->cap_bgzf_blk = h->cap_fq_carry = h->fq_carry_len = 0;
    hipFree(h->d_packed); hipFree(h->d_qrows); hipFree(h->d_lens); hipFree(h->d_counts); hipFree(h->d_dist); hipFree(h->d_query);
    for (auto* g : {&h->g_submit, &h->g_typing}) if (g->exec) hipGraphExecDestroy(g->exec);
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    delete h;
}

static inline int base_code(u8 c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

static int reset_sample_state(mlst_handle* h) {
    HIPCHK(h, hipMemsetAsync(h->d_stats, 0, h->stats_zero_bytes, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_stats + h->off_first, 0xFF, h->stats_bytes - h->off_first, h->stream));
    if (h->d_acc64 && h->n_alleles) HIPCHK(h, hipMemsetAsync(h->d_acc64, 0, (u64)h->n_alleles * 8, h->stream));      // (zero after every complete submission already: k_accumulate; this is for one that was cut short)
    h->reads_seen = 0; h->fq_carry_len = 0; h->max_wpr = 0;
    return MLST_OK;
}

// Build the device-resident reference: transposed 2-bit allele arena, N masks, seed sieve and exact seed table.
// ---- host-side index (everything of mlst_load_reference that does not depend on the device)
struct HostIndex {
    std::vector<LocusDev> loci; std::vector<u16> alen;
    std::vector<u32> arena, nmask, planes;
    std::vector<HapRec> hap_rec; std::vector<u32> hap_blk; std::vector<u32> hap_id;      // block-haplotype tables
    u32 hap_win_max[2] = {0, 0}; u32 hap_loci = 0;
    std::vector<u64> tkeys; std::vector<u32> tvals, posts; u32 tmask = 0;
    std::vector<u16> sv; u64 nb = 0; u32 smask = 0, sieve_chain = 0;      // fingerprint sieve; longest overflow walk of any key
    std::vector<u32> bitmap; double bitmap_fill = 0.0;                   // LDS half-seed bitmaps (small databases)
    std::vector<u32> gbitmap; u32 gbits = 0;                             // hashed global bitmap (MLST_SIEVE=global)
    std::vector<u32> rfilter;                                            // CU-routed filter slices (big databases)
    u64 n_keys = 0; u32 n_loci = 0; int kind = 0;                        // kind: MLST_SIEVE_* below
    std::string err; int err_code = 0;
};
enum { MLST_SIEVE_LDS = 0, MLST_SIEVE_GLOBAL = 1, MLST_SIEVE_ROUTED = 3 };      // (2 was round 1's XCD-binned sieve, replaced by the routed one)
struct KP { u64 key; u32 post; };

static u64 fnv1a(const void* p, u64 n, u64 hsh) {       // eight bytes at a time (a cache key, not a checksum of record)
    const u8* b = (const u8*)p; u64 i = 0;
    for (; i + 8 <= n; i += 8) { u64 v; memcpy(&v, b + i, 8); hsh = (hsh ^ v) * 0x100000001B3ull; hsh ^= hsh >> 29; }
    for (; i < n; i++) hsh = (hsh ^ b[i]) * 0x100000001B3ull;
    return hsh;
}

// Builds the index.  The alleles of a locus repeat most of their seeds (they differ by a few SNPs) and a posting names
// (locus, strand, position), so seeds are made unique locus by locus -- loci on as many host threads as there are -- and
// an allele only emits the windows in which it differs from the allele before it (alleles arrive in similarity order);
// the one sort of the whole index then handles ~4 % of the raw pairs of a many-species database.
static std::shared_ptr<HostIndex> build_host_index(const uint8_t* ascii, const uint64_t* off, const uint32_t* locus_id,
                                                   const uint32_t* species_id, uint32_t n_alleles, int want_kind) {
    auto H = std::make_shared<HostIndex>();
    auto bad = [&](int code, const char* fmt, ...) { char buf[256]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap); H->err = buf; H->err_code = code; return H; };
    // ---- locus table
    u32 n_loci = 0;
    for (u32 a = 0; a < n_alleles; a++) {
        if (locus_id[a] > MLST_MAX_LOCI) return bad(MLST_E_LIMIT, "locus id %u exceeds %d", locus_id[a], MLST_MAX_LOCI);
        n_loci = std::max(n_loci, locus_id[a] + 1);
        if (off[a + 1] - off[a] > MLST_MAX_ALLELE_LEN) return bad(MLST_E_LIMIT, "allele %u longer than %d", a, MLST_MAX_ALLELE_LEN);
    }
    H->n_loci = n_loci;
    std::vector<LocusDev>& loci = H->loci; loci.resize(n_loci); for (auto& L : loci) memset(&L, 0, sizeof L);
    std::vector<char> seen(n_loci, 0);
    for (u32 a = 0; a < n_alleles; a++) {
        LocusDev& L = loci[locus_id[a]];
        if (!seen[locus_id[a]]) { seen[locus_id[a]] = 1; L.a_begin = a; L.species = species_id ? species_id[a] : 0; }
        else if (L.a_begin + L.n_alleles != a) return bad(MLST_E_INVALID, "alleles of locus %u are not contiguous", locus_id[a]);
        L.n_alleles++; L.max_len = std::max(L.max_len, (u32)(off[a + 1] - off[a]));
    }
    u64 arena_words = 0, nmask_words = 0, plane_words = 0;
    std::vector<u16>& alen = H->alen; alen.resize(n_alleles);
    for (u32 a = 0; a < n_alleles; a++) alen[a] = (u16)(off[a + 1] - off[a]);
    for (auto& L : loci) {
        if (L.n_alleles >= (1u << 20)) return bad(MLST_E_LIMIT, "locus with %u alleles exceeds 2^20", L.n_alleles);
        L.n_pad = (L.n_alleles + 63) & ~63u; L.words = (L.max_len + 15) / 16 + 2; L.nwords = (L.max_len + 31) / 32 + 1;
        if ((u64)L.words * L.n_pad * 4 >= (1ull << 32)) return bad(MLST_E_LIMIT, "a locus arena exceeds 4 GB");
        L.arena_off = arena_words; arena_words += (u64)L.words * L.n_pad;
        L.pblocks = (L.max_len + 31) / 32 + 1; L.plane_off = plane_words; plane_words += (u64)L.pblocks * 2 * L.n_pad;
    }
    // N-mask rows only for loci that hold a non-ACGT character: found per locus below, laid out afterwards
    std::vector<u32>& arena = H->arena; std::vector<u32>& planes = H->planes;
    arena.assign(arena_words ? arena_words : 1, 0); planes.assign(plane_words ? plane_words : 1, 0);
    if (planes.size() * 4 >= (1ull << 32)) return bad(MLST_E_LIMIT, "allele bit-plane arena exceeds 4 GB (32-bit offsets in k_extend)");
    // ---- per locus, in parallel: arena rows, seed pairs (unique within the locus)
    std::vector<std::vector<KP>> lkp(n_loci);
    std::vector<std::vector<std::pair<u32, u32>>> lnpos(n_loci);      // (allele, position) of non-ACGT characters
    std::vector<std::vector<HapRec>> lhrec(n_loci); std::vector<std::vector<u32>> lhblk(n_loci); std::vector<std::vector<u32>> lhid(n_loci);
    {
        unsigned nthr = std::thread::hardware_concurrency(); if (nthr < 1) nthr = 1; if (nthr > 64) nthr = 64; if (nthr > n_loci) nthr = n_loci ? n_loci : 1;
        std::atomic<u32> next(0);
        auto work = [&]() {
            std::vector<u8> code; std::vector<u64> prev;
            struct HK { HapRec r; u32 al; };
            std::vector<HK> hk; std::vector<std::pair<u64, u32>> nmv;      // N-mask words of the locus, keyed (allele << 8) | block
            for (;;) {
                const u32 l = next.fetch_add(1); if (l >= n_loci) break;
                const LocusDev& L = loci[l]; std::vector<KP>& out = lkp[l];
                prev.assign(L.max_len + 1, ~0ull);
                for (u32 a = L.a_begin; a < L.a_begin + L.n_alleles; a++) {
                    const u32 al = a - L.a_begin, len = alen[a];
                    code.resize(len);
                    for (u32 i = 0; i < len; i++) {
                        const int c = base_code(ascii[off[a] + i]); code[i] = (u8)c;
                        if (c < 4) {
                            arena[L.arena_off + (u64)(i >> 4) * L.n_pad + al] |= (u32)c << (2 * (i & 15));
                            planes[L.plane_off + (u64)((i >> 5) * 2) * L.n_pad + al] |= (u32)(c & 1) << (i & 31);
                            planes[L.plane_off + (u64)((i >> 5) * 2 + 1) * L.n_pad + al] |= (u32)(c >> 1) << (i & 31);
                        } else lnpos[l].push_back({al, i});
                    }
                    if (len < MLST_SEED_LEN) continue;
                    // rolling forward key and reverse-complement key of the window [p, p+20)
                    int badw = 0; u64 fk = 0, rk = 0; const u64 mask40 = (1ull << 40) - 1;
                    for (u32 i = 0; i < len; i++) {
                        int c = code[i];
                        if (c > 3) { badw = MLST_SEED_LEN; c = 0; } else if (badw) badw--;
                        fk = (fk >> 2) | ((u64)c << 38);                 // base t of the window at bits 2t
                        rk = ((rk << 2) | (u64)(3 - c)) & mask40;        // rc base t = 3 - code[p+19-t]
                        if (i + 1 >= MLST_SEED_LEN) {
                            const u32 p = i + 1 - MLST_SEED_LEN;
                            if (badw) { prev[p] = ~0ull; continue; }
                            if (prev[p] == fk) continue;                  // the allele before this one had the same window here
                            prev[p] = fk;
                            // canonical entry: flag = 1 when the reverse complement is the smaller (canonical) form
                            const u64 ck = fk < rk ? fk : rk; const u32 fl = rk < fk ? 1u : 0u;
                            out.push_back({ck, (l << 13) | (fl << 12) | p});
                            if (fk == rk) out.push_back({ck, (l << 13) | (1u << 12) | p});   // palindrome: both strands
                        }
                    }
                    for (u32 pz = len >= MLST_SEED_LEN ? len - MLST_SEED_LEN + 1 : 0; pz < prev.size(); pz++) prev[pz] = ~0ull;   // windows this allele does not have
                    if (out.size() > (1u << 18)) {
                        std::sort(out.begin(), out.end(), [](const KP& x, const KP& y) { return x.key != y.key ? x.key < y.key : x.post < y.post; });
                        out.erase(std::unique(out.begin(), out.end(), [](const KP& x, const KP& y) { return x.key == y.key && x.post == y.post; }), out.end());
                    }
                }
                std::sort(out.begin(), out.end(), [](const KP& x, const KP& y) { return x.key != y.key ? x.key < y.key : x.post < y.post; });
                out.erase(std::unique(out.begin(), out.end(), [](const KP& x, const KP& y) { return x.key == y.key && x.post == y.post; }), out.end());
                // ---- block-haplotype tables: the alleles of a locus differ by a few SNPs, so a 32-base block has far
                // fewer distinct contents than the locus has alleles (profiles/round4/hap_counts.md); k_extend scores a
                // read against every distinct block once and composes the alleles from those summaries
                {
                    nmv.clear();
                    for (auto& ap : lnpos[l]) {
                        const u64 key = ((u64)ap.first << 8) | (ap.second >> 5);
                        if (nmv.empty() || nmv.back().first != key) nmv.push_back({key, 0u});
                        nmv.back().second |= 1u << (ap.second & 31);
                    }
                    std::vector<HapRec>& hr = lhrec[l]; std::vector<u32>& hb = lhblk[l]; std::vector<u32>& hi_ = lhid[l];
                    hb.assign(L.pblocks + 1, 0); hi_.assign((u64)((L.pblocks + 1) / 2) * L.n_pad, 0);
                    bool ok = true;
                    for (u32 q = 0; q < L.pblocks && ok; q++) {
                        hk.resize(L.n_alleles);
                        for (u32 al = 0; al < L.n_alleles; al++) {
                            const int left = (int)alen[L.a_begin + al] - 32 * (int)q;
                            HK& k = hk[al]; k.al = al; k.r.len = (u32)(left < 0 ? 0 : (left > 32 ? 32 : left));
                            k.r.lo = planes[L.plane_off + (u64)(2 * q) * L.n_pad + al]; k.r.hi = planes[L.plane_off + (u64)(2 * q + 1) * L.n_pad + al]; k.r.nm = 0;
                            if (!nmv.empty()) {
                                const u64 key = ((u64)al << 8) | q;
                                auto itn = std::lower_bound(nmv.begin(), nmv.end(), std::make_pair(key, 0u), [](const std::pair<u64, u32>& x, const std::pair<u64, u32>& y) { return x.first < y.first; });
                                if (itn != nmv.end() && itn->first == key) k.r.nm = itn->second;
                            }
                        }
                        auto less = [](const HK& x, const HK& y) {
                            if (x.r.lo != y.r.lo) return x.r.lo < y.r.lo; if (x.r.hi != y.r.hi) return x.r.hi < y.r.hi;
                            if (x.r.nm != y.r.nm) return x.r.nm < y.r.nm; return x.r.len < y.r.len; };
                        std::sort(hk.begin(), hk.end(), less);
                        u32 nh = 0;
                        for (u32 i = 0; i < L.n_alleles; i++) {
                            if (i == 0 || less(hk[i - 1], hk[i])) { if (nh >= 65535u) { ok = false; break; } hr.push_back(hk[i].r); nh++; }
                            hi_[(u64)(q >> 1) * L.n_pad + hk[i].al] |= (nh - 1) << (16 * (q & 1));
                        }
                        hb[q + 1] = hb[q] + nh;
                    }
                    if (!ok) { hr.clear(); hb.clear(); hi_.clear(); }
                }
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nthr; t++) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
    }
    {   // lay the block-haplotype tables of the loci out one behind the other
        u64 nrec = 0, nblk = 0, nid = 0;
        for (u32 l = 0; l < n_loci; l++) { nrec += lhrec[l].size(); nblk += lhblk[l].size(); nid += lhid[l].size(); }
        if (nrec >= (1ull << 28)) return bad(MLST_E_LIMIT, "block-haplotype table exceeds 2^28 records");
        H->hap_rec.reserve(nrec ? nrec : 1); H->hap_blk.reserve(nblk ? nblk : 1); H->hap_id.reserve(nid ? nid : 1);
        for (u32 l = 0; l < n_loci; l++) {
            LocusDev& L = loci[l];
            L.hap_ok = lhblk[l].empty() ? 0u : 1u; L.hap_off = (u32)H->hap_rec.size(); L.hblk_off = (u32)H->hap_blk.size(); L.hid_off = H->hap_id.size();
            L.hap_win[0] = L.hap_win[1] = 0;
            if (!L.hap_ok) continue;
            const std::vector<u32>& hb = lhblk[l];
            for (u32 q = 0; q < L.pblocks; q++) for (int k = 0; k < 2; k++) {
                const u32 qe = std::min(L.pblocks, q + (k ? 11u : 6u)); L.hap_win[k] = std::max(L.hap_win[k], hb[qe] - hb[q]);
            }
            for (int k = 0; k < 2; k++) H->hap_win_max[k] = std::max(H->hap_win_max[k], L.hap_win[k]);
            H->hap_loci++;
            H->hap_rec.insert(H->hap_rec.end(), lhrec[l].begin(), lhrec[l].end()); H->hap_blk.insert(H->hap_blk.end(), hb.begin(), hb.end());
            H->hap_id.insert(H->hap_id.end(), lhid[l].begin(), lhid[l].end());
            std::vector<HapRec>().swap(lhrec[l]); std::vector<u32>().swap(lhid[l]);
        }
        if (H->hap_rec.empty()) H->hap_rec.push_back(HapRec{0, 0, 0, 0}); if (H->hap_blk.empty()) H->hap_blk.push_back(0); if (H->hap_id.empty()) H->hap_id.push_back(0);
    }
    for (u32 l = 0; l < n_loci; l++) if (!lnpos[l].empty()) { loci[l].has_n = 1; loci[l].nmask_off = nmask_words; nmask_words += (u64)loci[l].nwords * loci[l].n_pad; }
    std::vector<u32>& nmask = H->nmask; nmask.assign(nmask_words ? nmask_words : 1, 0);
    if (nmask.size() * 4 >= (1ull << 32)) return bad(MLST_E_LIMIT, "N-mask arena exceeds 4 GB (32-bit offsets in k_extend)");
    for (u32 l = 0; l < n_loci; l++) for (auto& ap : lnpos[l]) nmask[loci[l].nmask_off + (u64)(ap.second >> 5) * loci[l].n_pad + ap.first] |= 1u << (ap.second & 31);
    std::vector<KP> kp;
    { u64 tot = 0; for (auto& v : lkp) tot += v.size(); kp.reserve(tot); for (auto& v : lkp) { kp.insert(kp.end(), v.begin(), v.end()); std::vector<KP>().swap(v); } }
    std::sort(kp.begin(), kp.end(), [](const KP& x, const KP& y) { return x.key != y.key ? x.key < y.key : x.post < y.post; });
    // group, drop repetitive seeds
    std::vector<u64> ukeys; std::vector<u32> uval; std::vector<u32>& posts = H->posts;
    for (size_t i = 0; i < kp.size();) {
        size_t j = i; while (j < kp.size() && kp[j].key == kp[i].key) j++;
        size_t cnt = j - i;
        if (cnt <= MLST_MAX_POSTINGS) {
            ukeys.push_back(kp[i].key);
            if (cnt == 1) uval.push_back(0x80000000u | kp[i].post);
            else {
                if (posts.size() >= (1ull << 26)) return bad(MLST_E_LIMIT, "posting list exceeds 2^26 entries");
                uval.push_back(((u32)posts.size() << 5) | (u32)cnt);
                for (size_t t = i; t < j; t++) posts.push_back(kp[t].post);
            }
        }
        i = j;
    }
    std::vector<KP>().swap(kp);
    const u64 nk = ukeys.size(); H->n_keys = nk;
    // ---- exact table: open addressing, load <= 0.5
    u64 tcap = 1024; while (tcap < 2 * nk) tcap <<= 1;
    if (tcap > (1ull << 32)) return bad(MLST_E_LIMIT, "seed table too large");
    std::vector<u64>& tkeys = H->tkeys; std::vector<u32>& tvals = H->tvals; tkeys.assign(tcap, KEY_EMPTY); tvals.assign(tcap, 0);
    const u32 tmask = (u32)(tcap - 1); H->tmask = tmask;
    for (u64 i = 0; i < nk; i++) {
        u32 lo = (u32)ukeys[i], hi = (u32)(ukeys[i] >> 32);
        u32 hh = table_hash(lo, hi) & tmask;
        while (tkeys[hh] != KEY_EMPTY) hh = (hh + 1) & tmask;
        tkeys[hh] = ukeys[i]; tvals[hh] = uval[i];
    }
    // ---- sieve: 16-byte buckets of eight 16-bit fingerprints, mean fill <= 4.  A key whose home bucket is full moves on
    // to the next one; the kernels follow such a chain for at most 64 buckets, so the build checks the longest walk and
    // doubles the sieve until it is far below that (it never is in practice: mean fill <= 4 of 8 slots).
    u64 nb = 256; while (nb * 4 < nk) nb <<= 1;
    for (;;) {
        if (nb > (1ull << 31)) return bad(MLST_E_LIMIT, "sieve too large");
        std::vector<u16>& sv = H->sv; sv.assign(nb * 8, 0); const u32 smask = (u32)(nb - 1);
        u32 sshift_h = 32; for (u64 t = nb; t > 1; t >>= 1) sshift_h--;      // nb = 2^(32 - sshift_h)
        u32 longest = 0;
        for (u64 i = 0; i < nk; i++) {
            u32 lo = (u32)ukeys[i], hi = (u32)(ukeys[i] >> 32);
            u32 fp = sieve_fp(lo, hi); u32 b = sieve_bucket_hash(lo, hi) >> sshift_h;
            for (u64 step = 0; step < nb; step++) {
                u16* B = &sv[(u64)b * 8]; int k; bool done = false;
                for (k = 0; k < 8; k++) { if (B[k] == fp) { done = true; break; } if (B[k] == 0) { B[k] = (u16)fp; done = true; break; } }
                if (done) { if ((u32)step > longest) longest = (u32)step; break; }
                b = (b + 1) & smask;
            }
        }
        H->nb = nb; H->smask = smask; H->sieve_chain = longest;
        if (longest <= 32) break;
        nb <<= 1;
    }
    // ---- first-level filter.  Small databases: two half-seed bitmaps kept in LDS by k_sieve_q<., true>, used when they
    // turn out at most half full.  Everything else: the CU-routed filter slices (K1c).  MLST_SIEVE = lds / routed /
    // global forces a kind (tests, A/B measurements; "lds" still falls back when the bitmaps are not selective).
    int kind = want_kind;
    if (kind < 0 || kind == MLST_SIEVE_LDS) {
        const u64 nbits = 1ull << BITMAP_BITS;
        if (nk <= 4 * nbits) {
            std::vector<u32>& bitmap = H->bitmap; bitmap.assign(nbits / 32, 0);
            u64 set = 0;
            for (u64 i = 0; i < nk; i++) {      // both orientations of every seed (see k_sieve_q)
                for (int o = 0; o < 2; o++) {
                    u32 L, R; sieve_halves_of(o ? revcomp40(ukeys[i]) : ukeys[i], L, R);
                    R += SV_HALF_BITS;
                    if (!((bitmap[L >> 5] >> (L & 31)) & 1u)) { bitmap[L >> 5] |= 1u << (L & 31); set++; }
                    if (!((bitmap[R >> 5] >> (R & 31)) & 1u)) { bitmap[R >> 5] |= 1u << (R & 31); set++; }
                }
            }
            H->bitmap_fill = (double)set / (double)nbits;       // mean fill of the two halves; a seed passes with ~fill^2
            if (set * 2 > nbits) { bitmap.clear(); H->bitmap_fill = 0.0; }   // more than half full: not selective
        }
        kind = H->bitmap.empty() ? MLST_SIEVE_ROUTED : MLST_SIEVE_LDS;
    }
    if (nk == 0) kind = H->bitmap.empty() ? MLST_SIEVE_GLOBAL : MLST_SIEVE_LDS;      // nothing to look up: the plain kernel with no first level
    if (kind == MLST_SIEVE_GLOBAL && nk > 0) {
        const char* gsw = getenv("MLST_GBM_BITS");            // tuning switch: 0 disables, default 25 (4 MiB)
        u32 gbits = gsw ? (u32)atoi(gsw) : 25u;
        if (gbits >= 16 && gbits <= 31 && nk <= (1ull << gbits)) {      // keep the expected fill below ~63 %
            H->gbitmap.assign((1ull << gbits) / 32, 0); H->gbits = gbits;
            for (u64 i = 0; i < nk; i++) { u32 bi = bitmap_hash_bits((u32)ukeys[i], (u32)(ukeys[i] >> 32), gbits); H->gbitmap[bi >> 5] |= 1u << (bi & 31); }
        }
    }
    if (kind == MLST_SIEVE_ROUTED) {
        H->rfilter.assign((u64)RT_OWNERS * RT_FWORDS, 0u);
        for (u64 i = 0; i < nk; i++) {
            u32 ow, hh, b0, b1, k0, k1; rt_hash((u32)ukeys[i], (u32)(ukeys[i] >> 32), ow, hh); rt_filter_addr(hh, b0, k0, b1, k1);
            H->rfilter[(u64)ow * RT_FWORDS + b0] |= k0; H->rfilter[(u64)ow * RT_FWORDS + b1] |= k1;
        }
    }
    H->kind = kind;
    return H;
}

// The last host index built in this process: a second engine that loads the same database (several engines per GPU,
// several GPUs per process) uploads it without building it again.
static std::mutex g_index_mu;
static std::shared_ptr<HostIndex> g_index_last; static u64 g_index_key[5] = {0, 0, 0, 0, 0};      // hashes of the inputs + their sizes

// ---- The built host index on disk (round 5; the reference keeps `<idx>.1.bt2` next to its FASTA dump and skips bowtie2-build
// when it is there: metamlst-index.py:224-225).  mlst_set_reference_cache names a file; mlst_load_reference reads the index
// from it when its header carries the key of the inputs (hashes of the allele text, offsets, locus and species ids + the
// sieve switches -- the same key the in-process cache uses) and writes it after a build.  A file that does not fit is ignored.
static std::mutex g_refcache_mu; static std::string g_refcache_path;
extern "C" int mlst_set_reference_cache(const char* path) { std::lock_guard<std::mutex> lk(g_refcache_mu); g_refcache_path = path ? path : ""; return MLST_OK; }
#define REFCACHE_MAGIC 0x3146455254534C4Dull      /* "MLSTREF1" */
#define REFCACHE_VERSION 3u
template <typename T> static bool rc_put(FILE* f, const std::vector<T>& v) { const u64 n = v.size(); return fwrite(&n, 8, 1, f) == 1 && (!n || fwrite(v.data(), sizeof(T), n, f) == n); }
template <typename T> static bool rc_get(FILE* f, std::vector<T>& v) {
    u64 n = 0; if (fread(&n, 8, 1, f) != 1 || n > (1ull << 36) / sizeof(T)) return false;
    v.resize(n); return !n || fread(v.data(), sizeof(T), n, f) == n;
}
struct RefCacheHead { u64 magic; u32 version, sz_locus, sz_hap, pad; u64 key[5]; u32 hap_win_max[2], hap_loci, tmask, smask, sieve_chain, gbits, n_loci; int kind; u32 pad2; u64 nb, n_keys; double bitmap_fill; };
static bool refcache_store(const std::string& path, const u64 key[5], const HostIndex& H) {
    const std::string tmp = path + "." + std::to_string((long)getpid()) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb"); if (!f) return false;
    RefCacheHead hd; memset(&hd, 0, sizeof hd);
    hd.magic = REFCACHE_MAGIC; hd.version = REFCACHE_VERSION; hd.sz_locus = (u32)sizeof(LocusDev); hd.sz_hap = (u32)sizeof(HapRec); memcpy(hd.key, key, sizeof hd.key);
    hd.hap_win_max[0] = H.hap_win_max[0]; hd.hap_win_max[1] = H.hap_win_max[1]; hd.hap_loci = H.hap_loci; hd.tmask = H.tmask; hd.smask = H.smask; hd.sieve_chain = H.sieve_chain;
    hd.gbits = H.gbits; hd.n_loci = H.n_loci; hd.kind = H.kind; hd.nb = H.nb; hd.n_keys = H.n_keys; hd.bitmap_fill = H.bitmap_fill;
    bool ok = fwrite(&hd, sizeof hd, 1, f) == 1 && rc_put(f, H.loci) && rc_put(f, H.alen) && rc_put(f, H.arena) && rc_put(f, H.nmask) && rc_put(f, H.planes) && rc_put(f, H.hap_rec)
              && rc_put(f, H.hap_blk) && rc_put(f, H.hap_id) && rc_put(f, H.tkeys) && rc_put(f, H.tvals) && rc_put(f, H.posts) && rc_put(f, H.sv) && rc_put(f, H.bitmap)
              && rc_put(f, H.gbitmap) && rc_put(f, H.rfilter);
    ok = (fclose(f) == 0) && ok;
    if (ok) ok = rename(tmp.c_str(), path.c_str()) == 0;
    if (!ok) remove(tmp.c_str());
    return ok;
}
static std::shared_ptr<HostIndex> refcache_load(const std::string& path, const u64 key[5]) {
    FILE* f = fopen(path.c_str(), "rb"); if (!f) return nullptr;
    auto H = std::make_shared<HostIndex>();
    RefCacheHead hd;
    bool ok = fread(&hd, sizeof hd, 1, f) == 1 && hd.magic == REFCACHE_MAGIC && hd.version == REFCACHE_VERSION && hd.sz_locus == sizeof(LocusDev) && hd.sz_hap == sizeof(HapRec)
              && memcmp(hd.key, key, sizeof hd.key) == 0;
    ok = ok && rc_get(f, H->loci) && rc_get(f, H->alen) && rc_get(f, H->arena) && rc_get(f, H->nmask) && rc_get(f, H->planes) && rc_get(f, H->hap_rec) && rc_get(f, H->hap_blk)
         && rc_get(f, H->hap_id) && rc_get(f, H->tkeys) && rc_get(f, H->tvals) && rc_get(f, H->posts) && rc_get(f, H->sv) && rc_get(f, H->bitmap) && rc_get(f, H->gbitmap) && rc_get(f, H->rfilter);
    fclose(f);
    if (!ok || H->loci.size() != hd.n_loci) return nullptr;
    H->hap_win_max[0] = hd.hap_win_max[0]; H->hap_win_max[1] = hd.hap_win_max[1]; H->hap_loci = hd.hap_loci; H->tmask = hd.tmask; H->smask = hd.smask; H->sieve_chain = hd.sieve_chain;
    H->gbits = hd.gbits; H->n_loci = hd.n_loci; H->kind = hd.kind; H->nb = hd.nb; H->n_keys = hd.n_keys; H->bitmap_fill = hd.bitmap_fill;
    return H;
}

static int sieve_kind_from_env() {
    const char* s = getenv("MLST_SIEVE");
    if (!s || !s[0]) return -1;
    if (!strcmp(s, "lds")) return MLST_SIEVE_LDS;
    if (!strcmp(s, "global")) return MLST_SIEVE_GLOBAL;
    if (!strcmp(s, "routed")) return MLST_SIEVE_ROUTED;
    return -1;
}

// dynamic LDS of k_extend_160 (k = 0) / _320: the haplotype summaries + the identity, the pending additions of the item (2 B per allele)
static size_t ext_lds_bytes(const mlst_handle* h, int k) { return (size_t)(h->ext_lds_recs[k] + 1) * 16 + (size_t)h->ext_acc_cap * 2; }
// Build the device-resident reference: transposed 2-bit allele arena, N masks, seed sieve and exact seed table.
extern "C" int mlst_load_reference(mlst_handle* h, const uint8_t* ascii, const uint64_t* off, const uint32_t* locus_id,
                                   const uint32_t* species_id, const int32_t* allele_no, uint32_t n_alleles) {
    if (!h || !off || !locus_id) return fail(h, MLST_E_INVALID, "NULL argument");
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    free_ref(h); free_state(h);
    for (auto* g : {&h->g_submit, &h->g_typing}) { if (g->exec) { hipGraphExecDestroy(g->exec); g->exec = nullptr; } g->sig.clear(); }
    h->n_alleles = n_alleles;
    const int want_kind = sieve_kind_from_env();
    std::shared_ptr<HostIndex> HI;
    {
        u64 key[5];
        key[3] = n_alleles; key[4] = off[n_alleles];
        key[0] = fnv1a(off, ((u64)n_alleles + 1) * 8, 0xCBF29CE484222325ull ^ n_alleles);
        key[1] = fnv1a(ascii, off[n_alleles], fnv1a(locus_id, (u64)n_alleles * 4, 0x9E3779B97F4A7C15ull));
        key[2] = (u64)(want_kind + 2);
        for (const char* v : {"MLST_GBM_BITS"}) { const char* e = getenv(v); if (e) key[2] = fnv1a(e, strlen(e), key[2]); }
        if (species_id) key[2] = fnv1a(species_id, (u64)n_alleles * 4, key[2]);
        std::lock_guard<std::mutex> lk(g_index_mu);
        const char* nocache = getenv("MLST_INDEX_CACHE");
        if (nocache && nocache[0] == '0') g_index_last.reset();
        if (g_index_last && memcmp(g_index_key, key, sizeof key) == 0) HI = g_index_last;
        else {
            std::string rc_path; { std::lock_guard<std::mutex> lk2(g_refcache_mu); rc_path = g_refcache_path; }
            if (!rc_path.empty()) HI = refcache_load(rc_path, key);
            if (!HI) {
                HI = build_host_index(ascii, off, locus_id, species_id, n_alleles, want_kind);
                if (HI->err_code) return fail(h, HI->err_code, "%s", HI->err.c_str());
                if (!rc_path.empty()) refcache_store(rc_path, key, *HI);      // (a directory that cannot be written: no cache, no error)
            }
            if (!(nocache && nocache[0] == '0')) { g_index_last = HI; memcpy(g_index_key, key, sizeof key); }
        }
    }
    const u32 n_loci = HI->n_loci; h->n_loci = n_loci;
    const std::vector<LocusDev>& loci = HI->loci; const std::vector<u16>& alen = HI->alen;
    const std::vector<u32>&arena = HI->arena, &nmask = HI->nmask, &planes = HI->planes, &posts = HI->posts, &tvals = HI->tvals, &bitmap = HI->bitmap, &gbitmap = HI->gbitmap;
    const std::vector<u64>& tkeys = HI->tkeys; const std::vector<u16>& sv = HI->sv;
    const u64 tcap = tkeys.size(), nb = HI->nb; const u32 tmask = HI->tmask, smask = HI->smask, gbits = HI->gbits;
    h->allele_locus.assign(locus_id, locus_id + n_alleles);
    h->sieve_kind = HI->kind; h->sieve_chain = HI->sieve_chain; h->n_keys = HI->n_keys;
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
    HIPCHK(h, dmalloc(&h->d_planes, planes.size())); HIPCHK(h, hipMemcpy(h->d_planes, planes.data(), planes.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_nmask, nmask.size())); HIPCHK(h, hipMemcpy(h->d_nmask, nmask.data(), nmask.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_allele_len, (u64)n_alleles)); HIPCHK(h, hipMemcpy(h->d_allele_len, alen.data(), (u64)n_alleles * 2, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_allele_locus, (u64)n_alleles)); HIPCHK(h, hipMemcpy(h->d_allele_locus, locus_id, (u64)n_alleles * 4, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_loci, (u64)n_loci)); HIPCHK(h, hipMemcpy(h->d_loci, loci.data(), (u64)n_loci * sizeof(LocusDev), hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_sieve, nb)); HIPCHK(h, hipMemcpy(h->d_sieve, sv.data(), nb * 16, hipMemcpyHostToDevice));
    if (!bitmap.empty()) { HIPCHK(h, dmalloc(&h->d_bitmap, (u64)bitmap.size())); HIPCHK(h, hipMemcpy(h->d_bitmap, bitmap.data(), bitmap.size() * 4, hipMemcpyHostToDevice)); }
    if (!HI->rfilter.empty()) { HIPCHK(h, dmalloc(&h->d_rfilter, (u64)HI->rfilter.size())); HIPCHK(h, hipMemcpy(h->d_rfilter, HI->rfilter.data(), HI->rfilter.size() * 4, hipMemcpyHostToDevice)); }
    if (!gbitmap.empty()) { HIPCHK(h, dmalloc(&h->d_gbitmap, (u64)gbitmap.size())); HIPCHK(h, hipMemcpy(h->d_gbitmap, gbitmap.data(), gbitmap.size() * 4, hipMemcpyHostToDevice)); }
    HIPCHK(h, dmalloc(&h->d_keys, tcap)); HIPCHK(h, hipMemcpy(h->d_keys, tkeys.data(), tcap * 8, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_vals, tcap)); HIPCHK(h, hipMemcpy(h->d_vals, tvals.data(), tcap * 4, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_posts, (u64)posts.size())); if (!posts.empty()) HIPCHK(h, hipMemcpy(h->d_posts, posts.data(), posts.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_floor, (u64)floor_tab.size())); HIPCHK(h, hipMemcpy(h->d_floor, floor_tab.data(), floor_tab.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_pen, (u64)256)); HIPCHK(h, hipMemcpy(h->d_pen, pen_tab.data(), 256, hipMemcpyHostToDevice));
    u64 abytes = off[n_alleles];
    HIPCHK(h, dmalloc(&h->d_ascii, abytes)); if (abytes) HIPCHK(h, hipMemcpy(h->d_ascii, ascii, abytes, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_aoff, (u64)n_alleles + 1)); HIPCHK(h, hipMemcpy(h->d_aoff, off, ((u64)n_alleles + 1) * 8, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_hap_rec, (u64)HI->hap_rec.size())); HIPCHK(h, hipMemcpy(h->d_hap_rec, HI->hap_rec.data(), HI->hap_rec.size() * sizeof(HapRec), hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_hap_blk, (u64)HI->hap_blk.size())); HIPCHK(h, hipMemcpy(h->d_hap_blk, HI->hap_blk.data(), HI->hap_blk.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(h, dmalloc(&h->d_hap_id, (u64)HI->hap_id.size())); HIPCHK(h, hipMemcpy(h->d_hap_id, HI->hap_id.data(), HI->hap_id.size() * 4, hipMemcpyHostToDevice));
    h->bytes_hap = HI->hap_rec.size() * sizeof(HapRec) + HI->hap_blk.size() * 4 + HI->hap_id.size() * 4; h->n_hap_rec = HI->hap_rec.size();
    h->hap_win_max[0] = HI->hap_win_max[0]; h->hap_win_max[1] = HI->hap_win_max[1]; h->hap_loci = HI->hap_loci;
    h->bytes_arena = arena.size() * 4 + planes.size() * 4 + nmask.size() * 4; h->bytes_sieve = nb * 16 + bitmap.size() * 4 + gbitmap.size() * 4 + HI->rfilter.size() * 4; h->bytes_table = tcap * 12 + posts.size() * 4;
    h->aoff.assign(off, off + n_alleles + 1);
    {   // device-side typing: allele numbers, one slot of max_len columns per locus
        HIPCHK(h, dmalloc(&h->d_allele_no, (u64)n_alleles)); HIPCHK(h, hipMemcpy(h->d_allele_no, allele_no, (u64)n_alleles * 4, hipMemcpyHostToDevice));
        h->fixed_colbase.assign(n_loci + 1, 0);
        for (u32 l = 0; l < n_loci; l++) h->fixed_colbase[l + 1] = h->fixed_colbase[l] + loci[l].max_len;
        h->fixed_cols = h->fixed_colbase[n_loci];
        HIPCHK(h, dmalloc(&h->d_fixed_colbase, (u64)n_loci + 1)); HIPCHK(h, hipMemcpy(h->d_fixed_colbase, h->fixed_colbase.data(), ((u64)n_loci + 1) * 8, hipMemcpyHostToDevice));
        HIPCHK(h, dmalloc(&h->d_auto_chosen, (u64)n_loci));
        HIPCHK(h, dmalloc(&h->d_auto_counts, h->fixed_cols * 4 + 4)); HIPCHK(h, dmalloc(&h->d_auto_letters, h->fixed_cols + 16));
        h->off_compact_info = (((u64)n_loci * 4 + 15) & ~15ull) + ((h->fixed_cols + 15) & ~15ull);
        for (int k = 0; k < 2; k++) { if (h->h_tauto[k]) hipHostFree(h->h_tauto[k]); HIPCHK(h, hipHostMalloc((void**)&h->h_tauto[k], h->off_compact_info + 64, hipHostMallocDefault)); }
        HIPCHK(h, dmalloc(&h->d_compact_colbase, (u64)n_loci + 1)); HIPCHK(h, dmalloc(&h->d_compact_chosen, (u64)n_loci + 1)); HIPCHK(h, dmalloc(&h->d_compact_info, 2));
    }
    // ---- sample state
    EngineDev& E = h->E;
    E.arena = h->d_arena; E.planes = h->d_planes; E.nmask = h->d_nmask; E.allele_len = h->d_allele_len; E.allele_locus = h->d_allele_locus; E.loci = h->d_loci;
    h->bitmap_fill = HI->bitmap_fill;
    E.sieve = h->d_sieve; E.sieve_mask = smask; E.bitmap = h->d_bitmap; E.gbitmap = h->d_gbitmap; E.gbitmap_bits = gbitmap.empty() ? 0 : gbits; E.keys = h->d_keys; E.vals = h->d_vals; E.posts = h->d_posts; E.table_mask = tmask;
    E.floor_tab = h->d_floor; E.pen_tab = h->d_pen; E.n_alleles = n_alleles; E.n_loci = n_loci;
    E.hap_rec = h->d_hap_rec; E.hap_blk = h->d_hap_blk; E.hap_id = h->d_hap_id;
    E.cap_ret = h->prm.max_retained_reads; E.cap_items = h->prm.max_items; E.cap_res = h->prm.max_pair_results; E.cap_dp = h->prm.max_items * 4;
    {   // statistics live in ONE device block so that a sample needs one memset pair and one D2H copy
        u64 o = 0;
        h->off_sum = o; o += (u64)n_alleles * 8;
        h->off_len = o; o += (u64)n_loci * 8;
        h->off_ctr = o; o += (sizeof(Counters) + 7) & ~7ull;
        h->off_hits = o; o += ((u64)n_alleles * 4 + 7) & ~7ull;
        h->stats_zero_bytes = o;
        h->off_first = o; o += (u64)n_loci * 8;
        h->stats_bytes = o ? o : 8;
        HIPCHK(h, dmalloc(&h->d_stats, h->stats_bytes));
        HIPCHK(h, hipHostMalloc((void**)&h->h_stats, h->stats_bytes, hipHostMallocDefault));
        for (int k = 0; k < 2; k++) { if (h->h_tstats[k]) hipHostFree(h->h_tstats[k]); HIPCHK(h, hipHostMalloc((void**)&h->h_tstats[k], h->stats_bytes, hipHostMallocDefault)); }
        h->t_slot = 0; h->t_ready = -1;
        E.sum_score = (long long*)(h->d_stats + h->off_sum); E.locus_len = (u64*)(h->d_stats + h->off_len);
        E.ctr = (Counters*)(h->d_stats + h->off_ctr); E.n_hits = (u32*)(h->d_stats + h->off_hits);
        E.locus_first = (u64*)(h->d_stats + h->off_first);
    }
    HIPCHK(h, dmalloc(&E.ret_bases, E.cap_ret * RW)); HIPCHK(h, dmalloc(&E.ret_quals, E.cap_ret * RQ));
    HIPCHK(h, dmalloc(&E.ret_len, E.cap_ret)); HIPCHK(h, dmalloc(&E.ret_ridx, E.cap_ret)); HIPCHK(h, dmalloc(&E.ret_nrec, E.cap_ret));
    HIPCHK(h, dmalloc(&E.ret_mate, E.cap_ret)); HIPCHK(h, dmalloc(&E.ret_item0, E.cap_ret)); HIPCHK(h, dmalloc(&E.ret_nitems, E.cap_ret)); HIPCHK(h, dmalloc(&E.ret_cpos, E.cap_ret));
    HIPCHK(h, dmalloc(&E.items, E.cap_items)); HIPCHK(h, dmalloc(&E.item_state, E.cap_items)); HIPCHK(h, dmalloc(&E.res, E.cap_res)); HIPCHK(h, dmalloc(&E.dp_list, E.cap_dp));
    HIPCHK(h, dmalloc(&h->d_locus_colbase, (u64)n_loci * 2 + 2));   // [colbase u64 x L][chosen int x L]
    HIPCHK(h, dmalloc(&h->d_pl_list, E.cap_items));
    HIPCHK(h, dmalloc(&h->d_tb, (u64)64 * 64 * MLST_MAX_READ_LEN * (2 * MAX_W + 1)));
    HIPCHK(h, dmalloc(&h->d_acc64, (u64)n_alleles)); HIPCHK(h, hipMemset(h->d_acc64, 0, (u64)(n_alleles ? n_alleles : 1) * 8)); E.acc64 = h->d_acc64;
    HIPCHK(h, dmalloc(&h->d_E, (u64)1)); HIPCHK(h, hipMemcpy(h->d_E, &h->E, sizeof(EngineDev), hipMemcpyHostToDevice));
    {   // Which extension kernel takes a locus (this engine's decision, in its own copy of the locus table):
        // k_extend (block-haplotype summaries, one-wave workgroups) the loci that have the tables, at most MLST_EXT_HAP_MAX
        // alleles (default 512: eight turns of one wave) and windows that fit the LDS budget (MLST_EXT_LDS_KB, default 96 of
        // the CU's 160 KB; 0 = no locus); k_extend_pairs (every pair on its own, rounds 1-3) the others.  Measured on
        // cfg2 (1,430 alleles per locus, 9 k items): haplotype form 0.40 / 0.25 / 0.19 ms with 64 / 128 / 256 threads per
        // item, pair form 0.18 ms: few items, and an item's chain of phases does not get shorter with its width.
        u32 hap_max = 512; const char* e0 = getenv("MLST_EXT_HAP_MAX"); if (e0 && atoi(e0) >= 0) hap_max = (u32)atoi(e0);
        u32 budget_kb = 96; const char* e4 = getenv("MLST_EXT_LDS_KB"); if (e4 && atoi(e4) >= 0 && atoi(e4) <= 150) budget_kb = (u32)atoi(e4);
        std::vector<LocusDev> lc = loci;
        u32 mx_hap = 0, mx_pair = 0, n_hap = 0, n_pair = 0; h->hap_win_max[0] = h->hap_win_max[1] = 0;
        for (auto& L : lc) {
            const bool take = L.hap_ok && L.n_alleles <= hap_max && (u64)L.hap_win[1] * 16 <= (u64)budget_kb * 1024;
            L.hap_ok = take ? 1u : 0u;
            if (take) { n_hap++; mx_hap = std::max(mx_hap, L.n_alleles); for (int k = 0; k < 2; k++) h->hap_win_max[k] = std::max(h->hap_win_max[k], L.hap_win[k]); }
            else { n_pair++; mx_pair = std::max(mx_pair, L.n_alleles); }
        }
        h->loci = lc; h->hap_loci = n_hap; h->pair_loci = n_pair;
        HIPCHK(h, hipMemcpy(h->d_loci, lc.data(), (u64)n_loci * sizeof(LocusDev), hipMemcpyHostToDevice));
        // launch shapes: threads per work item (lanes = alleles) and workgroups (the work queues balance the rest).
        // MLST_EXT_THREADS / MLST_EXT_BLOCKS override those of the haplotype kernel, MLST_EXTP_THREADS / _BLOCKS the other's.
        int thr = mx_hap <= 512 ? 64 : 256;
        const char* e1 = getenv("MLST_EXT_THREADS"); if (e1 && atoi(e1) >= 64 && atoi(e1) <= 1024 && atoi(e1) % 64 == 0) thr = atoi(e1);
        int blocks = 1280 * 256 / thr;     // 5 waves per SIMD at ~100 VGPRs (one-wave workgroups: 5,120)
        const char* e2 = getenv("MLST_EXT_BLOCKS"); if (e2 && atoi(e2) > 0) blocks = atoi(e2);
        h->ext_threads = thr; h->ext_blocks = blocks;
        // pair kernel: one-wave workgroups up to 512 alleles per locus, 256 threads beyond (cfg2: 0.18 ms against 0.31 with one wave)
        int thr2 = mx_pair <= 512 ? 64 : 256;
        const char* e5 = getenv("MLST_EXTP_THREADS"); if (e5 && atoi(e5) >= 64 && atoi(e5) <= 1024 && atoi(e5) % 64 == 0) thr2 = atoi(e5);
        int blocks2 = 1792 * 256 / thr2;   // 7 waves per SIMD (k_extend_pairs_160 is held to 72 VGPRs)
        const char* e6 = getenv("MLST_EXTP_BLOCKS"); if (e6 && atoi(e6) > 0) blocks2 = atoi(e6);
        h->extp_threads = thr2; h->extp_blocks = blocks2;
        // dynamic LDS of the haplotype kernel: the summaries of the widest window of its loci, the pending additions of an item
        h->ext_acc_cap = std::min<u32>((mx_hap + 63u) & ~63u, 16384u);
        for (int k = 0; k < 2; k++) {
            h->ext_lds_recs[k] = h->hap_win_max[k];
            if (ext_lds_bytes(h, k) > 48u * 1024u)
                HIPCHK(h, hipFuncSetAttribute(k ? (const void*)k_extend_320 : (const void*)k_extend_160, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ext_lds_bytes(h, k)));
        }
        const char* e3 = getenv("MLST_SIEVE_BLOCKS"); if (e3 && atoi(e3) > 0) h->sieve_g_blocks = atoi(e3);
        const char* e7 = getenv("MLST_RT_SLICE"); h->rt_slice = (e7 && atoll(e7) > 0) ? (u64)atoll(e7) : 0;
    }
    h->have_ref = h->have_state = true;
    int rc = reset_sample_state(h); if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MLST_OK;
}

extern "C" int mlst_set_read_index_base(mlst_handle* h, uint64_t base) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    h->reads_seen = base;
    return MLST_OK;
}

extern "C" int mlst_reset_sample(mlst_handle* h) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    hipSetDevice(h->device);
    drain_events(h);
    if (h->bz_pend.on) { if (h->infl_stream) hipStreamSynchronize(h->infl_stream); h->bz_pend.on = false; }      // a BGZF piece still open belongs to the sample that is dropped
    return reset_sample_state(h);
}

static int grid_for(u64 n_units, int per_block, int cap = 2048) {
    u64 g = (n_units + per_block - 1) / per_block; if (g < 1) g = 1; if (g > (u64)cap) g = cap; return (int)g;
}
static void zero_words(mlst_handle* h, void* p, u64 n_words) {
    if (!n_words) return;
    hipLaunchKernelGGL(k_zero, dim3(grid_for((n_words + 255) / 256, 1, 1024)), dim3(256), 0, h->stream, (u32*)p, n_words);
}

extern "C" int mlst_pack_reads_device(mlst_handle* h, const uint8_t* d_bases, const uint8_t* d_quals, const uint64_t* d_off,
                                      uint64_t n_reads, uint32_t* d_packed, uint8_t* d_qrows, uint16_t* d_lens,
                                      uint32_t wpr, uint32_t qstride) {
    if (!h) return MLST_E_INVALID;
    hipSetDevice(h->device);
    if (wpr == 0 || wpr > RW || (wpr & 1)) return fail(h, MLST_E_INVALID, "words_per_read must be even and in 2..%d", RW);
    // (k_pack stores quality rows as 32-bit words: a stride that is not a multiple of four would tear neighbouring rows)
    if (qstride < 4 || qstride > RQ || (qstride & 3)) return fail(h, MLST_E_INVALID, "qual_stride must be a multiple of 4 in 4..%d", RQ);
    if (((uintptr_t)d_qrows & 3) != 0) return fail(h, MLST_E_INVALID, "quality rows must be 4-byte aligned");
    if (n_reads == 0) return MLST_OK;
    Prof pf(h, 6);
    hipLaunchKernelGGL(k_pack, dim3(grid_for((n_reads + 63) / 64, 1, 8192)), dim3(256), 0, h->stream, d_bases, d_quals, (const u64*)d_off, (u64)n_reads,
                       d_packed, d_qrows, d_lens, wpr, qstride);
    HIPCHK(h, hipGetLastError());
    return MLST_OK;
}

// arena / list buffers of the CU-routed sieve for a batch of n_reads (grow-only; must run outside stream capture)
static int ensure_route_buffers(mlst_handle* h, u64 n_reads, u32 wpr, u64 n_reads_flags = 0) {
    u32 nw = 16;                                        // waves per producer workgroup (tile = nw groups of 64 reads)
    { const char* e = getenv("MLST_ROUTE_WAVES"); if (e && atoi(e) == 8) nw = 8; }
    { const char* e = getenv("MLST_PROBE_PF"); h->rt_pf = (e && atoi(e) == 4) ? 4u : 2u; }
    const u64 tile = (u64)nw * 64, n_tiles = (n_reads + tile - 1) / tile;
    u32 prod = (wpr <= 10 ? 512u : 256u) * (16u / nw) / (u32)h->cu_split;  // as many workgroups as the LDS lets share the CUs
    { const char* e = getenv("MLST_ROUTE_BLOCKS"); if (e && atoi(e) > 0) prod = (u32)atoi(e); }
    if (prod > RT_MAXP) prod = RT_MAXP;
    if (prod > n_tiles) prod = (u32)(n_tiles ? n_tiles : 1);
    // tiles are handed out dynamically: a workgroup may take up to half as many again as its even share (the spread seen
    // is +-20 %); the capacities below (list of tiles, regions) are sized for that and a workgroup stops asking at the bound
    u64 tiles_max = (n_tiles + prod - 1) / prod * 3 / 2 + 8;
    // expected entries per (owner, tile): tile * seeds / 256 + dummies (~10 % of nw) + ~2 of padding; 25 % and a constant on top
    u64 cap = (u64)((double)tiles_max * ((double)tile / 256.0 * (wpr - 1) + 0.15 * nw + 2.0) * 1.25) + 256; cap = (cap + 31) & ~31ull;      // regions start on 128-byte boundaries, whole 64-byte chunks are written
    // A layout that is large enough stays (round 5): the capacities are upper bounds, and a submission of fewer reads than the one
    // before -- the pieces of a bgzip'd file differ in size -- used to re-allocate (a device-wide synchronisation) for its smaller one.
    if (h->rt_prod == prod && h->rt_nw == nw && h->rt_cap >= cap && h->rt_tiles_max >= tiles_max) { cap = h->rt_cap; tiles_max = h->rt_tiles_max; }
    if (cap >= (1ull << 31) || tiles_max > 0xFFFFull || (u64)RT_OWNERS * prod * (cap / 16) >= (1ull << 32)) return fail(h, MLST_E_LIMIT, "batch too large for the routed sieve (%llu tiles per producer workgroup)", (unsigned long long)tiles_max);
    const u64 need = (u64)RT_OWNERS * prod * cap;
    if (h->cap_rt_arena < need || h->rt_prod != prod || h->rt_cap != (u32)cap) {
        hipStreamSynchronize(h->stream);
        if (h->cap_rt_arena < need) { hipFree(h->d_rt_arena); h->d_rt_arena = nullptr; HIPCHK(h, dmalloc(&h->d_rt_arena, need)); h->cap_rt_arena = need; }
        hipFree(h->d_rt_counts); h->d_rt_counts = nullptr; HIPCHK(h, dmalloc(&h->d_rt_counts, (u64)RT_OWNERS * prod));
        h->rt_prod = prod; h->rt_cap = (u32)cap;
    }
    const u64 need_e = (u64)prod * (tiles_max + 1);
    if (h->cap_rt_emitted < need_e) { hipStreamSynchronize(h->stream); hipFree(h->d_rt_emitted); h->d_rt_emitted = nullptr; HIPCHK(h, dmalloc(&h->d_rt_emitted, need_e)); h->cap_rt_emitted = need_e; }
    h->rt_tiles_max = (u32)tiles_max; h->rt_nw = nw;
    if (h->rt_trace_on && !h->d_rt_trace) { hipStreamSynchronize(h->stream); HIPCHK(h, dmalloc(&h->d_rt_trace, (u64)(RT_MAXP + RT_OWNERS) * 4)); HIPCHK(h, hipMemsetAsync(h->d_rt_trace, 0, (u64)(RT_MAXP + RT_OWNERS) * 32, h->stream)); }
    // ~0.6 % of the entries pass the filter (1.5 x the reads' real hits + 0.4 % of chance): room for a quarter of the reads,
    // beyond which the consumer examines in place
    const u64 need_p = std::max<u64>(n_reads / 4, 1ull << 16);
    if (h->cap_rt_parked < need_p) { hipStreamSynchronize(h->stream); hipFree(h->d_rt_parked); h->d_rt_parked = nullptr; HIPCHK(h, dmalloc(&h->d_rt_parked, need_p)); h->cap_rt_parked = need_p; }
    const u64 n_flag_words = ((n_reads_flags > n_reads ? n_reads_flags : n_reads) + 31) >> 5;
    if (h->cap_bin_flags < n_flag_words) { hipStreamSynchronize(h->stream); hipFree(h->d_bin_flags); h->d_bin_flags = nullptr; HIPCHK(h, dmalloc(&h->d_bin_flags, n_flag_words)); h->cap_bin_flags = n_flag_words;
                                            // (on the engine's stream, not hipMemset: the legacy stream waits for every blocking stream, and ANOTHER engine of the
                                            // process may be capturing its graph on one right now -- "would make the legacy stream depend on a capturing blocking stream")
                                            HIPCHK(h, hipMemsetAsync(h->d_bin_flags, 0, n_flag_words * 4, h->stream)); }
    return MLST_OK;
}

// phase 0 = a whole submission (the only one replayed as a hipGraph); 1 = the sieve alone (candidate list), 2 = everything behind
// the sieve with d_qrows holding the rows of the CANDIDATES only, in candidate order (mlst_submit_packed_host)
static int submit_impl(mlst_handle* h, const uint32_t* d_packed, const uint8_t* d_qrows, const uint16_t* d_lens,
                       uint64_t n_reads, uint32_t wpr, uint32_t qstride, int paired, int phase);
extern "C" int mlst_submit_packed_device(mlst_handle* h, const uint32_t* d_packed, const uint8_t* d_qrows, const uint16_t* d_lens,
                                         uint64_t n_reads, uint32_t wpr, uint32_t qstride, int paired) {
    return submit_impl(h, d_packed, d_qrows, d_lens, n_reads, wpr, qstride, paired, 0);
}
static int submit_impl(mlst_handle* h, const uint32_t* d_packed, const uint8_t* d_qrows, const uint16_t* d_lens,
                       uint64_t n_reads, uint32_t wpr, uint32_t qstride, int paired, int phase) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    hipSetDevice(h->device);
    if (wpr == 0 || wpr > RW || (wpr & 1)) return fail(h, MLST_E_INVALID, "words_per_read must be even and in 2..%d", RW);
    if (qstride < 4 || qstride > RQ || (qstride & 3)) return fail(h, MLST_E_INVALID, "qual_stride must be a multiple of 4 in 4..%d", RQ);
    if (phase != 1 && (!d_qrows || ((uintptr_t)d_qrows & 3) != 0)) return fail(h, MLST_E_INVALID, "quality rows must be 4-byte aligned");
    if (n_reads >= (1ull << 32)) return fail(h, MLST_E_LIMIT, "a batch holds at most 2^32-1 reads");
    if (((uintptr_t)d_packed & 15) != 0) return fail(h, MLST_E_INVALID, "packed rows must be 16-byte aligned");
    if (n_reads == 0) return MLST_OK;
    // mates are aligned and typed independently (the documented pipeline is bowtie2 -U); sharing a QNAME matters for the
    // per-locus read-length sums only (Q3, k_locus)
    paired = paired ? 1 : 0;
    if (paired && ((n_reads | h->reads_seen) & 1ull)) return fail(h, MLST_E_INVALID, "paired submissions hold whole pairs (reads 2k, 2k+1) and start at an even read index");
    if (h->cap_cand < n_reads) { hipStreamSynchronize(h->stream); hipFree(h->d_cand); h->d_cand = nullptr; HIPCHK(h, dmalloc(&h->d_cand, n_reads)); h->cap_cand = n_reads; }
    EngineDev& E = h->E;
    if (wpr > h->max_wpr) h->max_wpr = wpr;
    // Sub-batched routed sieve (experiment b of VERDICT r3 item 2; MLST_RT_SLICE = reads per slice, 0 = off): route -> probe ->
    // verify slice by slice, every slice's entries into the same small arena, which then never leaves the 256 MiB
    // Infinity Cache -- measured: profiles/round4/sieve_experiments.md
    u64 rt_slice = 0;
    if (h->sieve_kind == MLST_SIEVE_ROUTED && h->rt_slice) { rt_slice = (h->rt_slice + 1023) & ~1023ull; if (rt_slice >= n_reads) rt_slice = 0; }
    if (h->sieve_kind == MLST_SIEVE_ROUTED) { int rc = ensure_route_buffers(h, rt_slice ? rt_slice : n_reads, wpr, n_reads); if (rc) return rc; }
    {   // item records of k_ext_prep: one per work item a sample may hold (allocated on first use of an instantiation)
        const int k = wpr <= 10 ? 0 : 1; const u64 words = k ? (u64)XRec<RW / 2>::WORDS : (u64)XRec<5>::WORDS;
        if (h->hap_loci && !h->d_xrec[k]) { hipStreamSynchronize(h->stream); HIPCHK(h, dmalloc(&h->d_xrec[k], h->E.cap_items * words)); h->cap_xrec[k] = h->E.cap_items; }
    }
    const int gs = phase ? 0 : graph_enter(h, h->g_submit, {(u64)(uintptr_t)d_packed, (u64)(uintptr_t)d_qrows, (u64)(uintptr_t)d_lens, (u64)n_reads, (u64)wpr,
                                               (u64)qstride, (u64)h->reads_seen, (u64)(uintptr_t)h->d_cand,
                                               (u64)(uintptr_t)h->d_bin_flags, (u64)(uintptr_t)h->d_rt_arena, (u64)(uintptr_t)h->d_rt_counts,
                                               (u64)(uintptr_t)h->d_rt_emitted, (u64)h->rt_cap, (u64)h->rt_prod, (u64)h->rt_nw, (u64)paired, (u64)(uintptr_t)h->d_rt_trace, (u64)(uintptr_t)h->d_rt_parked, (u64)h->cap_rt_parked, (u64)h->rt_pf, rt_slice});
    if (gs == 1) { h->reads_seen += n_reads; return MLST_OK; }
    if (phase != 2) { Prof pf(h, 0);
      if (h->sieve_kind == MLST_SIEVE_LDS) {      // LDS first level: one 1024-thread workgroup per CU
        dim3 grid(grid_for((n_reads + 1023) / 1024, 1, 256)), block(1024);
#define SIEVE_CASE(W) case W: hipLaunchKernelGGL((k_sieve_q<W, true>), grid, block, 0, h->stream, d_packed, d_lens, (u64)n_reads, E.sieve, E.sieve_mask, E.bitmap, 0u, h->d_cand, E.ctr, paired); break;
        switch (wpr) { SIEVE_CASE(2) SIEVE_CASE(4) SIEVE_CASE(6) SIEVE_CASE(8) SIEVE_CASE(10) SIEVE_CASE(12) SIEVE_CASE(14)
                       SIEVE_CASE(16) SIEVE_CASE(18) SIEVE_CASE(20) default: return fail(h, MLST_E_INVALID, "words_per_read %u unsupported", wpr); }
#undef SIEVE_CASE
      } else if (h->sieve_kind == MLST_SIEVE_ROUTED) {      // seeds routed to the CU that owns their filter slice (K1c)
        const u64 n_flag_words = (n_reads + 31) >> 5;      // (all zero: cleared at allocation and again by every k_flag_compact)
        for (u64 sl_first = 0; sl_first < n_reads; sl_first += (rt_slice ? rt_slice : n_reads)) {
        const u64 sl_n = rt_slice ? std::min<u64>(rt_slice, n_reads - sl_first) : n_reads;
        const u32* sl_packed = d_packed + (sl_first >> 6) * 64 * wpr; const u16* sl_lens = d_lens + sl_first;
        if (sl_first) hipLaunchKernelGGL(k_rt_reset, dim3(1), dim3(1), 0, h->stream, E.ctr);
        RouteDev R; R.arena = h->d_rt_arena; R.counts = h->d_rt_counts; R.emitted = h->d_rt_emitted; R.filter = h->d_rfilter; R.flags = h->d_bin_flags + (sl_first >> 5); R.trace = h->d_rt_trace; R.parked = h->d_rt_parked; R.parked_cap = h->cap_rt_parked;
        h->rt_last_packed = d_packed;
        R.cap = h->rt_cap; R.n_prod = h->rt_prod; R.tiles_max = h->rt_tiles_max; R.nw = h->rt_nw;
        { const char* e = getenv("MLST_RT_DEBUG"); R.dbg = e ? (u32)atoi(e) : 0u; if (R.dbg & 2u) R.parked_cap = 0; }      // 2 = examine in place (no k_route_verify work)
        { Prof pa(h, 9);
#define SIEVE_CASE(W) case W: if (h->rt_nw == 8) hipLaunchKernelGGL((k_route<W, 8>), dim3(h->rt_prod), dim3(512), 0, h->stream, sl_packed, sl_lens, (u64)sl_n, R, E.ctr); \
                              else hipLaunchKernelGGL((k_route<W, 16>), dim3(h->rt_prod), dim3(1024), 0, h->stream, sl_packed, sl_lens, (u64)sl_n, R, E.ctr); break;
        switch (wpr) { SIEVE_CASE(2) SIEVE_CASE(4) SIEVE_CASE(6) SIEVE_CASE(8) SIEVE_CASE(10) SIEVE_CASE(12) SIEVE_CASE(14)
                       SIEVE_CASE(16) SIEVE_CASE(18) SIEVE_CASE(20) default: return fail(h, MLST_E_INVALID, "words_per_read %u unsupported", wpr); }
#undef SIEVE_CASE
        }
        { Prof pb(h, 10);
#define SIEVE_CASE(W) case W: if (h->rt_pf == 4) hipLaunchKernelGGL((k_route_probe<W, 4>), dim3(RT_OWNERS), dim3(1024), 0, h->stream, sl_packed, (u64)sl_n, E.sieve, E.sieve_mask, R, E.ctr); \
                              else hipLaunchKernelGGL((k_route_probe<W, 2>), dim3(RT_OWNERS), dim3(1024), 0, h->stream, sl_packed, (u64)sl_n, E.sieve, E.sieve_mask, R, E.ctr); break;
        switch (wpr) { SIEVE_CASE(2) SIEVE_CASE(4) SIEVE_CASE(6) SIEVE_CASE(8) SIEVE_CASE(10) SIEVE_CASE(12) SIEVE_CASE(14)
                       SIEVE_CASE(16) SIEVE_CASE(18) SIEVE_CASE(20) default: return fail(h, MLST_E_INVALID, "words_per_read %u unsupported", wpr); }
#undef SIEVE_CASE
        }
        { Prof pc(h, 11);
#define SIEVE_CASE(W) case W: hipLaunchKernelGGL(k_route_verify<W>, dim3(2048), dim3(256), 0, h->stream, sl_packed, (u64)sl_n, E.sieve, E.sieve_mask, R, E.ctr); break;
        switch (wpr) { SIEVE_CASE(2) SIEVE_CASE(4) SIEVE_CASE(6) SIEVE_CASE(8) SIEVE_CASE(10) SIEVE_CASE(12) SIEVE_CASE(14)
                       SIEVE_CASE(16) SIEVE_CASE(18) SIEVE_CASE(20) default: return fail(h, MLST_E_INVALID, "words_per_read %u unsupported", wpr); }
#undef SIEVE_CASE
        }
        }
        hipLaunchKernelGGL(k_flag_compact, dim3(grid_for((n_flag_words + FLAG_U - 1) / FLAG_U, 1024, 512)), dim3(1024), 0, h->stream, h->d_bin_flags, (u64)n_reads, h->d_cand, E.ctr, paired);
      } else {      // hashed first-level bitmap in global memory, 256-thread workgroups (MLST_SIEVE=global; databases without seeds)
        dim3 grid(grid_for((n_reads + 255) / 256, 1, h->sieve_g_blocks)), block(256);
#define SIEVE_CASE(W) case W: hipLaunchKernelGGL((k_sieve_q<W, false>), grid, block, 0, h->stream, d_packed, d_lens, (u64)n_reads, E.sieve, E.sieve_mask, E.gbitmap.p, E.gbitmap_bits, h->d_cand, E.ctr, paired); break;
        switch (wpr) { SIEVE_CASE(2) SIEVE_CASE(4) SIEVE_CASE(6) SIEVE_CASE(8) SIEVE_CASE(10) SIEVE_CASE(12) SIEVE_CASE(14)
                       SIEVE_CASE(16) SIEVE_CASE(18) SIEVE_CASE(20) default: return fail(h, MLST_E_INVALID, "words_per_read %u unsupported", wpr); }
#undef SIEVE_CASE
      }
    }
    if (phase == 1) { HIPCHK(h, hipGetLastError()); return MLST_OK; }
    { Prof pf(h, 1);
      hipLaunchKernelGGL(k_seed, dim3(1024), dim3(256), 0, h->stream, h->d_E, d_packed, d_qrows, d_lens, wpr, qstride, h->reads_seen, h->d_cand, paired, phase == 2 ? 1 : 0);
      // (one pair of reads per workgroup and sweep, every sweep a chain of dependent scattered loads: a large grid keeps the
      // sweeps few -- 1024 workgroups took 59 sweeps = 112 us for the 121 k retained reads of cfg3)
      hipLaunchKernelGGL(k_retain, dim3(2048), dim3(256), 0, h->stream, h->d_E, d_packed, d_qrows, wpr, qstride, h->reads_seen, phase == 2 ? 1 : 0); }
    if (h->hap_loci) {
      { Prof pf(h, 12);     // item records (the read on the allele's block grid, the locus' fields, the block table)
        if (wpr <= 10) hipLaunchKernelGGL(k_ext_prep<5>, dim3(2048), dim3(256), 0, h->stream, h->d_E, h->kp, h->d_xrec[0], h->cap_xrec[0]);
        else hipLaunchKernelGGL(k_ext_prep<RW / 2>, dim3(2048), dim3(256), 0, h->stream, h->d_E, h->kp, h->d_xrec[1], h->cap_xrec[1]); }
      { Prof pf(h, 2);     // register arrays sized for the batch's read words: 160 bp and 320 bp instantiations
        const int thr = h->ext_threads, blocks = h->ext_blocks;
        if (wpr <= 10) hipLaunchKernelGGL(k_extend_160, dim3(blocks), dim3(thr), ext_lds_bytes(h, 0), h->stream, h->d_E, h->kp, h->ext_lds_recs[0], h->ext_acc_cap, h->d_xrec[0], h->cap_xrec[0]);
        else hipLaunchKernelGGL(k_extend_320, dim3(blocks), dim3(std::min(thr, 512)), ext_lds_bytes(h, 1), h->stream, h->d_E, h->kp, h->ext_lds_recs[1], h->ext_acc_cap, h->d_xrec[1], h->cap_xrec[1]); }
    }
    if (h->pair_loci) { Prof pf(h, 2);      // the loci the haplotype kernel does not take
      const int thr = h->extp_threads, blocks = h->extp_blocks;
      if (wpr <= 10) hipLaunchKernelGGL(k_extend_pairs_160, dim3(blocks), dim3(thr), 0, h->stream, h->d_E, h->kp);
      else hipLaunchKernelGGL(k_extend_pairs_320, dim3(blocks), dim3(thr), 0, h->stream, h->d_E, h->kp); }
    { Prof pf(h, 3); hipLaunchKernelGGL(k_banded, dim3(1024), dim3(256), 0, h->stream, h->d_E, h->kp); }
    { Prof pf(h, 4); hipLaunchKernelGGL(k_accumulate, dim3(1024), dim3(256), 0, h->stream, h->d_E, h->kp);
      hipLaunchKernelGGL(k_locus, dim3(128), dim3(1024), 0, h->stream, h->d_E, paired); }
    hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, h->stream, E.ctr, n_reads);
    if (gs == 2) { int rc = graph_leave(h, h->g_submit); if (rc) return rc; }
    HIPCHK(h, hipGetLastError());
    h->reads_seen += n_reads;
    return MLST_OK;
}

static int ensure_pack_buffers(mlst_handle* h, u64 n_reads, u32 wpr, u32 qstride) {
    if (h->cap_packed_words < packed_words(n_reads, wpr) + 4 || h->cap_qrow_bytes < n_reads * qstride || h->cap_lens < n_reads + 2) {
        hipStreamSynchronize(h->stream);
        hipFree(h->d_packed); hipFree(h->d_qrows); hipFree(h->d_lens); h->d_packed = nullptr; h->d_qrows = nullptr; h->d_lens = nullptr;
        HIPCHK(h, dmalloc(&h->d_packed, packed_words(n_reads, wpr) + 4)); HIPCHK(h, dmalloc(&h->d_qrows, n_reads * qstride)); HIPCHK(h, dmalloc(&h->d_lens, n_reads + 2));
        h->cap_packed_words = packed_words(n_reads, wpr) + 4; h->cap_qrow_bytes = n_reads * qstride; h->cap_lens = n_reads + 2;
    }
    return MLST_OK;
}

// Host -> device copy on the engine's copy stream, ordered in front of whatever the engine's stream does next.  The runtime
// moves a caller's pageable buffer at the speed of the link already (56 GB/s against 57 GB/s from pinned memory on the
// MI355X box, profiles/h2d_rate.py; a ring of pinned slices filled by worker threads was measured and is slower), so what
// is left to gain is overlap: the copy runs on its own stream while the engine's stream still works on the chunk before
// (the FASTQ entries alternate between two device text buffers).  `before` (optional) is an event the copy has to wait
// for: the last reader of the destination buffer.  The source has been read completely when the call returns.
static int h2d_overlapped(mlst_handle* h, void* d_dst, const void* src, u64 n, hipEvent_t before) {
    if (n == 0) return MLST_OK;
    if (!h->copy_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    if (!h->stage_done) HIPCHK(h, hipEventCreateWithFlags(&h->stage_done, hipEventDisableTiming));
    if (before) HIPCHK(h, hipStreamWaitEvent(h->copy_stream, before, 0));
    HIPCHK(h, hipMemcpyAsync(d_dst, src, n, hipMemcpyHostToDevice, h->copy_stream));
    HIPCHK(h, hipEventRecord(h->stage_done, h->copy_stream));
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->stage_done, 0));
    HIPCHK(h, hipStreamSynchronize(h->copy_stream));      // the caller owns src again on return
    return MLST_OK;
}

extern "C" int mlst_submit_reads_device(mlst_handle* h, const uint8_t* d_bases, const uint8_t* d_quals, const uint64_t* d_off,
                                        uint64_t n_reads, uint32_t max_len, int paired) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    { int rc_ = bz_flush(h); if (rc_) return rc_; }
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
    { int rc_ = bz_flush(h); if (rc_) return rc_; }
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
    HIPCHK(h, hipMemcpyAsync(h->d_in_bases, bases + off[0], nbytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_in_quals, quals + off[0], nbytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_in_off, rel.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));      // (`rel` and the caller's arrays are read until here; on the engine's stream, not the legacy one: DESIGN.md 4a)
    return mlst_submit_reads_device(h, h->d_in_bases, h->d_in_quals, (const uint64_t*)h->d_in_off, n_reads, max_len, paired);
}

// The next text buffer (the two are used in turn), grown to `bytes`; h->d_fq_text points at it.  The chunk that used it
// last has been packed when ev_packed[slot] fires.
static int next_text_slot(mlst_handle* h, u64 bytes) {
    const int slot = h->fq_slot ^= 1;
    if (!h->ev_packed[slot]) HIPCHK(h, hipEventCreateWithFlags(&h->ev_packed[slot], hipEventDisableTiming));
    if (h->cap_fq_slot[slot] < bytes) {
        HIPCHK(h, hipEventSynchronize(h->ev_packed[slot]));      // (an event never recorded counts as complete)
        hipFree(h->d_fq_slot[slot]); h->d_fq_slot[slot] = nullptr; HIPCHK(h, dmalloc(&h->d_fq_slot[slot], bytes + 16)); h->cap_fq_slot[slot] = bytes;
        hipFree(h->d_fq_nl[slot]); h->d_fq_nl[slot] = nullptr; HIPCHK(h, dmalloc(&h->d_fq_nl[slot], bytes / 4096 + 8));
    }
    h->d_fq_text = h->d_fq_slot[slot];
    return MLST_OK;
}

// FASTQ text in h->d_fq_text[0 .. n_bytes) -> packed reads -> pass 1.  whole: the text consists of whole records; else
// the partial record at its end is kept (h->d_fq_carry) for the next chunk.
// pair_boundary != 0 (whole text only): the text is two mate files back to back, the second starting at that byte; both
// must hold the same number of records, which are then interleaved (k_fq_records) and submitted as pairs.
// tslot: the text slot the chunk sits in (-1: the current one); skip: bytes of filler in front of the first line (see k_fq_lines).
// ext_blk (optional): the newline counts per FQ_BLOCK bytes come with the text (the inflate counted them while writing it) except for
// the first count_bytes bytes (a multiple of FQ_BLOCK: the filler and the partial record in front of an inflated piece).
static int fastq_pipeline(mlst_handle* h, u64 n_bytes, int paired, bool whole, uint64_t* n_reads_out, u64 pair_boundary = 0, int tslot = -1, u64 skip = 0,
                          u32* ext_blk = nullptr, u64 count_bytes = 0) {
    if (tslot < 0) tslot = h->fq_slot;
    const u64 n_blocks = (n_bytes + FQ_BLOCK - 1) / FQ_BLOCK;
    if (!ext_blk && h->cap_fq_blk < n_blocks) { hipFree(h->d_fq_blk); h->d_fq_blk = nullptr; HIPCHK(h, dmalloc(&h->d_fq_blk, n_blocks + 1)); h->cap_fq_blk = n_blocks; }
    u32* const blk = ext_blk ? ext_blk : h->d_fq_blk;
    HIPCHK(h, hipMemsetAsync(h->d_fq_meta, 0, 16, h->stream));
    Prof pf(h, 6);
    if (!ext_blk) hipLaunchKernelGGL(k_fq_count, dim3((u32)n_blocks), dim3(256), 0, h->stream, h->d_fq_text, (u64)n_bytes, blk);
    else if (count_bytes) hipLaunchKernelGGL(k_fq_count, dim3((u32)(count_bytes / FQ_BLOCK)), dim3(256), 0, h->stream, h->d_fq_text, (u64)count_bytes, blk);
    hipLaunchKernelGGL(k_fq_scan, dim3(1), dim3(1024), 0, h->stream, blk, (u32)n_blocks, h->d_fq_meta, (u64)n_bytes, h->d_fq_text, whole ? 1 : 0);
    u64 n_lines = 0;
    HIPCHK(h, hipMemcpyAsync(&n_lines, h->d_fq_meta, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (n_lines >= (1ull << 32)) return fail(h, MLST_E_LIMIT, "more than 2^32 lines in one FASTQ chunk");
    if (whole && n_lines % 4 != 0) return fail(h, MLST_E_INVALID, "FASTQ chunk holds %llu lines: not a whole number of 4-line records", (unsigned long long)n_lines);
    const u64 n_reads = n_lines / 4;
    if (h->cap_fq_lines < n_lines + 2) { hipFree(h->d_fq_lines); h->d_fq_lines = nullptr; HIPCHK(h, dmalloc(&h->d_fq_lines, n_lines + 2)); h->cap_fq_lines = n_lines + 2; }
    hipLaunchKernelGGL(k_fq_lines, dim3((u32)n_blocks), dim3(256), 0, h->stream, h->d_fq_text, (u64)n_bytes, blk, h->d_fq_lines, skip);
    h->fq_carry_len = 0;
    if (!whole) {      // text behind the last whole record waits for the next chunk
        u64 end_off = 0;
        HIPCHK(h, hipMemcpyAsync(&end_off, h->d_fq_lines + 4 * n_reads, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (end_off > n_bytes) return fail(h, MLST_E_HIP, "FASTQ line table inconsistent");
        const u64 keep = n_bytes - end_off;
        if (keep) {
            if (h->cap_fq_carry < keep) { hipFree(h->d_fq_carry); h->d_fq_carry = nullptr; HIPCHK(h, dmalloc(&h->d_fq_carry, keep + 16)); h->cap_fq_carry = keep; }
            HIPCHK(h, hipMemcpyAsync(h->d_fq_carry, h->d_fq_text + end_off, keep, hipMemcpyDeviceToDevice, h->stream));
            h->fq_carry_len = keep;
        }
        n_bytes = end_off; n_lines = 4 * n_reads;
    }
    if (n_reads == 0) { HIPCHK(h, hipEventRecord(h->ev_packed[tslot], h->stream)); return MLST_OK; }
    u64 pair_k = 0;
    if (pair_boundary) {
        if (n_reads & 1) return fail(h, MLST_E_INVALID, "mate files hold different numbers of records (%llu records in all)", (unsigned long long)n_reads);
        pair_k = n_reads / 2;
        u64 second = 0;
        HIPCHK(h, hipMemcpyAsync(&second, h->d_fq_lines + 4 * pair_k, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (second != pair_boundary) return fail(h, MLST_E_INVALID, "mate files hold different numbers of records (record %llu starts at byte %llu, the second file at %llu)",
                                                 (unsigned long long)pair_k, (unsigned long long)second, (unsigned long long)pair_boundary);
    }
    if (h->cap_fq_reads < n_reads) { hipFree(h->d_fq_soff); hipFree(h->d_fq_qoff); h->d_fq_soff = h->d_fq_qoff = nullptr;
                                     HIPCHK(h, dmalloc(&h->d_fq_soff, n_reads)); HIPCHK(h, dmalloc(&h->d_fq_qoff, n_reads)); h->cap_fq_reads = n_reads; }
    // lengths are needed before the packed buffers can be sized: worst-case row width first, then the real one
    int rc = ensure_pack_buffers(h, n_reads, 2, 8); if (rc) return rc;
    u32* d_flags = reinterpret_cast<u32*>(h->d_fq_meta + 1);
    hipLaunchKernelGGL(k_fq_records, dim3(grid_for(n_reads, 256)), dim3(256), 0, h->stream, h->d_fq_text, (u64)n_bytes, h->d_fq_lines, n_lines, n_reads,
                       h->d_fq_soff, h->d_fq_qoff, h->d_lens, d_flags, pair_k);
    u32 flags[2] = {0, 0};
    HIPCHK(h, hipMemcpyAsync(flags, d_flags, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (flags[1] & 2) return fail(h, MLST_E_LIMIT, "a FASTQ read is longer than %d bases", MLST_MAX_READ_LEN);
    if (flags[1] & 1) return fail(h, MLST_E_INVALID, "malformed FASTQ: a record does not start with '@' or its sequence and quality lengths differ");
    u32 max_len = flags[0];
    u32 wpr = (max_len + 15) / 16; if (wpr < 2) wpr = 2; wpr = (wpr + 1) & ~1u;
    u32 qstride = (max_len + 7) & ~7u; if (qstride < 8) qstride = 8;
    {   // k_fq_records wrote the lengths into d_lens; growing the pack buffers must keep them
        std::vector<u16> keep;
        if (h->cap_packed_words < packed_words(n_reads, wpr) + 4 || h->cap_qrow_bytes < n_reads * qstride) {
            // (copies on the engine's stream, not hipMemcpy: the legacy stream waits for every blocking stream, and another engine of the
            // process -- the folder mode runs six, a feeder thread each -- may be capturing its graph on one: DESIGN.md 4a)
            keep.resize(n_reads); HIPCHK(h, hipMemcpyAsync(keep.data(), h->d_lens, n_reads * 2, hipMemcpyDeviceToHost, h->stream)); HIPCHK(h, hipStreamSynchronize(h->stream));
            rc = ensure_pack_buffers(h, n_reads, wpr, qstride); if (rc) return rc;
            HIPCHK(h, hipMemcpyAsync(h->d_lens, keep.data(), n_reads * 2, hipMemcpyHostToDevice, h->stream)); HIPCHK(h, hipStreamSynchronize(h->stream));
        }
    }
    hipLaunchKernelGGL(k_pack_text, dim3(grid_for((n_reads + 63) / 64, 1, 8192)), dim3(256), 0, h->stream, h->d_fq_text, h->d_fq_soff, h->d_fq_qoff,
                       h->d_lens, n_reads, h->d_packed, h->d_qrows, wpr, qstride);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev_packed[tslot], h->stream));      // the text buffer may be overwritten from here on
    if (n_reads_out) *n_reads_out = n_reads;
    return mlst_submit_packed_device(h, h->d_packed, h->d_qrows, h->d_lens, n_reads, wpr, qstride, paired);
}

// ---- Host-packed input (VERDICT r3 missing 8).  FASTQ text costs the link 2 L + ~16 bytes per read, and all of
// it crosses although the sieve reads the bases only and 0.24 % of the reads ever need their Phred values.  Here the HOST
// packs (mlst_pack_fastq_host: all host threads, the resident 2-bit layout + raw Phred rows + lengths, byte for byte what
// k_pack_text makes of the same text) and only bases and lengths cross -- 42 bytes per 150-base read; the sieve runs; the
// candidate list comes back (4 bytes per candidate); the host gathers the candidates' Phred rows and sends those: 152 bytes for
// one read in 400.  One host synchronisation in the middle of the submission, outside any hipGraph.
static void par_for(u64 n, int threads, const std::function<void(u64, u64, int)>& fn) {
    if (threads < 1) threads = 1;
    if ((u64)threads > n) threads = (int)(n ? n : 1);
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(fn, n * t / threads, n * (t + 1) / threads, t);
    fn(0, n / threads, 0);
    for (auto& th : pool) th.join();
}
extern "C" int mlst_pack_fastq_host(const uint8_t* text, uint64_t n_bytes, uint32_t wpr, uint32_t qstride, uint32_t* packed, uint8_t* qrows,
                                    uint16_t* lens, uint64_t cap_reads, uint64_t* n_reads_out, int threads) {
    if (!text || !packed || !qrows || !lens || !n_reads_out) return MLST_E_INVALID;
    if (wpr == 0 || wpr > RW || (wpr & 1) || qstride < 4 || qstride > RQ || (qstride & 3)) return MLST_E_INVALID;
    if (threads <= 0) { threads = (int)std::thread::hardware_concurrency(); if (threads < 1) threads = 1; if (threads > 128) threads = 128; }
    *n_reads_out = 0;
    // line starts: per-thread lists of the newlines of a slice of the text, then one table
    std::vector<std::vector<u64>> nl(threads);
    par_for(n_bytes, threads, [&](u64 lo, u64 hi, int t) {
        std::vector<u64>& v = nl[t]; v.reserve((hi - lo) / 64 + 16);
        const u8* p = text + lo; const u8* e = text + hi;
        while (p < e) { const u8* q = (const u8*)memchr(p, '\n', (size_t)(e - p)); if (!q) break; v.push_back((u64)(q - text) + 1); p = q + 1; }
    });
    std::vector<u64> first(threads + 1, 0);
    for (int t = 0; t < threads; t++) first[t + 1] = first[t] + nl[t].size();
    u64 n_lines = first[threads] + ((n_bytes && text[n_bytes - 1] != '\n') ? 1 : 0);      // an unterminated last line counts
    if (n_lines % 4) return MLST_E_INVALID;
    const u64 n = n_lines / 4;
    if (n > cap_reads) return MLST_E_CAPACITY;
    std::vector<u64> ls(n_lines + 1); ls[0] = 0;
    par_for((u64)threads, threads, [&](u64 lo, u64 hi, int) { for (u64 t = lo; t < hi; t++) if (!nl[t].empty()) memcpy(&ls[1 + first[t]], nl[t].data(), nl[t].size() * 8); });
    if (n_lines > first[threads]) ls[n_lines] = n_bytes + 1;      // (as if a newline followed the text)
    std::atomic<int> bad(0);
    u8 lut[256]; for (int c = 0; c < 256; c++) { const int u = c & 0xDF; lut[c] = (u8)(u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : u == 'T' ? 3 : 4); }      // bit 2 = not ACGT (packed as A)
    const u64 n_groups = (n + 63) >> 6;
    par_for(n_groups, threads, [&](u64 glo, u64 ghi, int) {
        u32 words[64 * RW];
        for (u64 grp = glo; grp < ghi; grp++) {
            memset(words, 0, sizeof(u32) * 64 * wpr);
            for (u32 i = 0; i < 64; i++) {
                const u64 r = grp * 64 + i; if (r >= n) break;
                u64 s0 = ls[4 * r + 1], se = ls[4 * r + 2] - 1, q0 = ls[4 * r + 3], qe = ls[4 * r + 4] - 1;
                if (se > s0 && text[se - 1] == '\r') se--;
                if (qe > q0 && text[qe - 1] == '\r') qe--;
                u64 L = se - s0;
                if (text[ls[4 * r]] != '@' || L != qe - q0) { bad.store(1); L = L < qe - q0 ? L : qe - q0; }
                if (L > MLST_MAX_READ_LEN || L > (u64)wpr * 16) { bad.store(2); L = std::min<u64>(MLST_MAX_READ_LEN, (u64)wpr * 16); }
                u8* qr = qrows + r * qstride;
                const u8* sp = text + s0; const u8* qp = text + q0;
                const u64 Lq = L < qstride ? L : qstride;
                u32 anyn = 0; u32* wrow = words + i * wpr;
                for (u64 w0 = 0; w0 < L; w0 += 16) {                      // sixteen bases = one packed word
                    const u64 e = L - w0 < 16 ? L - w0 : 16; u32 word = 0;
                    for (u64 k = 0; k < e; k++) { const u32 c = lut[sp[w0 + k]]; word |= (c & 3u) << (2 * k); anyn |= c >> 2; }
                    wrow[w0 >> 4] = word;
                }
                for (u64 p2 = 0; p2 < Lq; p2++) { int q = (int)qp[p2] - 33; q = q < 0 ? 0 : (q > 127 ? 127 : q); qr[p2] = (u8)(q | ((lut[sp[p2]] >> 2) << 7)); }
                if (Lq < qstride) memset(qr + Lq, 0, qstride - Lq);
                lens[r] = (u16)(L | (anyn << 15));
            }
            u32* out = packed + grp * 64 * wpr;      // resident order: word o of the group = unit o >> 7, read (o >> 1) & 63, half o & 1
            for (u32 o = 0; o < 64 * wpr; o++) out[o] = words[((o >> 1) & 63u) * wpr + ((o >> 7) << 1) + (o & 1u)];
        }
    });
    if (bad.load()) return MLST_E_INVALID;
    *n_reads_out = n;
    return MLST_OK;
}
extern "C" int mlst_submit_packed_host(mlst_handle* h, const uint32_t* packed, const uint8_t* qrows, const uint16_t* lens, uint64_t n_reads,
                                       uint32_t wpr, uint32_t qstride, int paired) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    { int rc_ = bz_flush(h); if (rc_) return rc_; }
    if (!packed || !qrows || !lens) return fail(h, MLST_E_INVALID, "NULL argument");
    hipSetDevice(h->device);
    if (wpr == 0 || wpr > RW || (wpr & 1)) return fail(h, MLST_E_INVALID, "words_per_read must be even and in 2..%d", RW);
    if (qstride < 4 || qstride > RQ || (qstride & 3)) return fail(h, MLST_E_INVALID, "qual_stride must be a multiple of 4 in 4..%d", RQ);
    if (n_reads == 0) return MLST_OK;
    // bases + lengths cross the link; the quality buffer on the device only ever holds candidates' rows
    if (h->cap_packed_words < packed_words(n_reads, wpr) + 4 || h->cap_lens < n_reads + 2) {
        hipStreamSynchronize(h->stream);
        hipFree(h->d_packed); hipFree(h->d_lens); h->d_packed = nullptr; h->d_lens = nullptr;
        HIPCHK(h, dmalloc(&h->d_packed, packed_words(n_reads, wpr) + 4)); HIPCHK(h, dmalloc(&h->d_lens, n_reads + 2));
        h->cap_packed_words = packed_words(n_reads, wpr) + 4; h->cap_lens = n_reads + 2;
        if (h->cap_qrow_bytes) { hipFree(h->d_qrows); h->d_qrows = nullptr; h->cap_qrow_bytes = 0; }      // (sized with the other two elsewhere: start over)
    }
    HIPCHK(h, hipMemcpyAsync(h->d_packed, packed, packed_words(n_reads, wpr) * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_lens, lens, n_reads * 2, hipMemcpyHostToDevice, h->stream));
    int rc = submit_impl(h, h->d_packed, nullptr, h->d_lens, n_reads, wpr, qstride, paired, 1);
    if (rc) return rc;
    // the candidate list
    u64 n_cand = 0;
    HIPCHK(h, hipMemcpyAsync(&n_cand, (const u8*)h->E.ctr.p + offsetof(Counters, n_cand), 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (n_cand > n_reads) n_cand = n_reads;
    std::vector<u32> cand(n_cand ? n_cand : 1);
    if (n_cand) { HIPCHK(h, hipMemcpyAsync(cand.data(), h->d_cand, n_cand * 4, hipMemcpyDeviceToHost, h->stream)); HIPCHK(h, hipStreamSynchronize(h->stream)); }
    const u64 qbytes = n_cand * qstride;
    if (h->cap_hostq < qbytes) { if (h->h_qc) hipHostFree(h->h_qc); h->h_qc = nullptr; HIPCHK(h, hipHostMalloc((void**)&h->h_qc, qbytes + 64, hipHostMallocDefault)); h->cap_hostq = qbytes + 64; }
    if (h->cap_qc < qbytes) { hipFree(h->d_qc); h->d_qc = nullptr; HIPCHK(h, dmalloc(&h->d_qc, qbytes + 64)); h->cap_qc = qbytes + 64; }
    int thr = (int)std::thread::hardware_concurrency(); if (thr < 1) thr = 1; if (thr > 32) thr = 32; if (n_cand < 4096) thr = 1;
    par_for(n_cand, thr, [&](u64 lo, u64 hi, int) { for (u64 k = lo; k < hi; k++) memcpy(h->h_qc + k * qstride, qrows + (u64)cand[k] * qstride, qstride); });
    if (qbytes) HIPCHK(h, hipMemcpyAsync(h->d_qc, h->h_qc, qbytes, hipMemcpyHostToDevice, h->stream));
    return submit_impl(h, h->d_packed, h->d_qc, h->d_lens, n_reads, wpr, qstride, paired, 2);
}

// BGZF blocks -> text.  Two ways (MLST_INFLATE_MODE): 2 = the two-kernel path of csrc/inflate_lane.h (one lane per block decodes
// the Huffman codes into tokens, one workgroup per block turns tokens into bytes by pointer jumping in LDS), in passes of at
// most INFL_PASS blocks (the token buffer holds 96 KB per block of a pass), with the blocks whose tokens did not fit left to
// 1 = the one-wave-per-block kernel of csrc/inflate_wave.h.
#define INFL_PASS 16384u          /* blocks per pass with k_inflate_tok: one wave (64 blocks) per CU is all its 160 KB of tables allow */
#define INFL_PASS2 65536u         /* with k_inflate_tok2: four waves per CU (the token buffer holds 96 KB per block of a pass: 6.3 GB) */
// d_nl (optional; two-kernel path only): newline counts per FQ_BLOCK bytes of d_out, added up while the text is written (zeroed by the caller)
static int launch_inflate(mlst_handle* h, const u8* d_comp, u64 comp_bytes_padded, const BgzfBlk* d_blk, u32 n_blk, u8* d_out, u32* d_err, unsigned long long* d_st, hipStream_t st = nullptr, u32* d_nl = nullptr) {
    if (n_blk == 0) return MLST_OK;
    if (!st) st = h->stream;
    if (!h->inflate_mode) { const char* e = getenv("MLST_INFLATE_MODE"); h->inflate_mode = e ? atoi(e) : 2; if (h->inflate_mode != 1 && h->inflate_mode != 2) h->inflate_mode = 2; }
    if (h->inflate_mode == 1 || d_st) {
        if (d_nl) return fail(h, MLST_E_INVALID, "newline counts come with the two-kernel inflate only");
        hipLaunchKernelGGL(k_inflate, dim3((u32)std::min<u64>(((u64)n_blk + INFLATE_NG - 1) / INFLATE_NG, 1u << 20)), dim3(64), 0, st, d_comp, comp_bytes_padded, d_blk, n_blk, d_out, d_err, d_st, (const u32*)nullptr, 0u);
        return MLST_OK;
    }
    // Phase 1: k_inflate_tok (tables of 9 / 8 bits, 2.3 KB per stream: ONE wave of 64 blocks per CU, 0.58 us per symbol step) or
    // k_inflate_tok2 (canonical limits, 576 B per stream: four waves per CU, 0.9 us per step).  Both are bound by the latency of a
    // wave, so what counts is how many blocks are in flight: up to 16,384 blocks the first finishes in one turn (4.4 ms) and wins;
    // a larger piece costs it a turn per 16,384 blocks (49,152 blocks: 13.2 ms) while the second still takes one (9 ms, up to 65,536 blocks).
    // MLST_INFLATE_TOK = 1 / 2 forces one of them; default: by the number of blocks.
    const char* tok_e = getenv("MLST_INFLATE_TOK"); const int tok_env = tok_e ? atoi(tok_e) : 0;
    const int tok_kind = tok_env == 1 || tok_env == 2 ? tok_env : (n_blk > 18432u ? 2 : 1);      // (24,576 blocks: 1.5 turns of the first = 8.8 ms, one of the second ~7.5)
    const u32 pass = std::min(n_blk, tok_kind == 1 ? INFL_PASS : INFL_PASS2);
    if (h->cap_itok_blocks < pass) {
        HIPCHK(h, hipStreamSynchronize(st)); HIPCHK(h, hipStreamSynchronize(h->stream));
        hipFree(h->d_itok); hipFree(h->d_intok); h->d_itok = nullptr; h->d_intok = nullptr; h->cap_itok_blocks = 0;
        HIPCHK(h, dmalloc(&h->d_itok, (u64)pass * INFL_TOK_CAP)); HIPCHK(h, dmalloc(&h->d_intok, (u64)pass));
        h->cap_itok_blocks = pass;
    }
    for (u32 at = 0; at < n_blk; at += pass) {
        const u32 n = std::min(pass, n_blk - at);
        if (tok_kind == 1) hipLaunchKernelGGL(k_inflate_tok, dim3((n + 63) / 64), dim3(64), 0, st, d_comp, comp_bytes_padded, d_blk + at, n, at, h->d_itok, h->d_intok, d_err);
        else hipLaunchKernelGGL(k_inflate_tok2, dim3((n + 63) / 64), dim3(64), 0, st, d_comp, comp_bytes_padded, d_blk + at, n, at, h->d_itok, h->d_intok, d_err);
        hipLaunchKernelGGL(k_inflate_ptr, dim3(std::min(n, 4096u)), dim3(1024), 0, st, d_comp, d_blk + at, n, at, (const u32*)h->d_itok, (const u32*)h->d_intok, d_out, d_err, d_nl, 12u);
        hipLaunchKernelGGL(k_inflate, dim3((u32)std::min<u64>(((u64)n + INFLATE_NG - 1) / INFLATE_NG, 1u << 20)), dim3(64), 0, st, d_comp, comp_bytes_padded, d_blk + at, n, d_out, d_err, (unsigned long long*)nullptr, (const u32*)h->d_intok, at);
        if (d_nl) hipLaunchKernelGGL(k_nl_blocks, dim3(std::min(n, 2048u)), dim3(256), 0, st, d_blk + at, n, (const u32*)h->d_intok, (const u8*)d_out, d_nl, 12u);
    }
    return MLST_OK;
}

// Page-locked host memory for a caller's input buffers (FASTQ text read from files): a copy from such a buffer is one DMA
// transfer at the link's rate; from ordinary memory the runtime first copies through its own staging buffers on a host
// thread.  Process-wide (no engine needed); released by mlst_free_host.
extern "C" int mlst_alloc_host(uint64_t n_bytes, void** out) {
    if (!out) return MLST_E_INVALID;
    *out = nullptr;
    if (n_bytes == 0) return MLST_E_INVALID;
    void* p = nullptr;
    if (hipHostMalloc(&p, n_bytes, hipHostMallocDefault) != hipSuccess || !p) { (void)hipGetLastError(); return MLST_E_HIP; }
    *out = p;
    return MLST_OK;
}
extern "C" int mlst_free_host(void* p) {
    if (!p) return MLST_OK;
    return hipHostFree(p) == hipSuccess ? MLST_OK : MLST_E_HIP;
}

extern "C" int mlst_submit_fastq(mlst_handle* h, const uint8_t* text, uint64_t n_bytes, int paired, uint64_t* n_reads_out) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    { int rc_ = bz_flush(h); if (rc_) return rc_; }
    if (n_reads_out) *n_reads_out = 0;
    if (n_bytes == 0) return MLST_OK;
    if (!text) return fail(h, MLST_E_INVALID, "NULL argument");
    if (n_bytes >= (1ull << 40)) return fail(h, MLST_E_LIMIT, "FASTQ chunk too large");
    if (h->fq_carry_len) return fail(h, MLST_E_INVALID, "a BGZF stream is open (its last chunk was not marked final)");
    hipSetDevice(h->device);
    // no wait for the chunk before: its text sits in the other buffer, and this buffer's last reader is named by an event
    { int rc = next_text_slot(h, n_bytes); if (rc) return rc; }
    if (!h->d_fq_meta) HIPCHK(h, dmalloc(&h->d_fq_meta, (u64)4));
    { int rc = h2d_overlapped(h, h->d_fq_text, text, n_bytes, h->ev_packed[h->fq_slot]); if (rc) return rc; }
    return fastq_pipeline(h, n_bytes, paired, true, n_reads_out);
}

extern "C" int mlst_submit_fastq_stream(mlst_handle* h, const uint8_t* text, uint64_t n_bytes, int final_chunk, int paired, uint64_t* n_reads_out) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    { int rc_ = bz_flush(h); if (rc_) return rc_; }
    if (n_reads_out) *n_reads_out = 0;
    if (n_bytes && !text) return fail(h, MLST_E_INVALID, "NULL argument");
    const u64 total = h->fq_carry_len + n_bytes;
    if (total == 0) return MLST_OK;
    if (total >= (1ull << 40)) return fail(h, MLST_E_LIMIT, "FASTQ chunk too large");
    hipSetDevice(h->device);
    { int rc = next_text_slot(h, total); if (rc) return rc; }
    if (!h->d_fq_meta) HIPCHK(h, dmalloc(&h->d_fq_meta, (u64)4));
    const u64 carry = h->fq_carry_len;
    if (carry) HIPCHK(h, hipMemcpyAsync(h->d_fq_text, h->d_fq_carry, carry, hipMemcpyDeviceToDevice, h->stream));
    { int rc = h2d_overlapped(h, h->d_fq_text + carry, text, n_bytes, h->ev_packed[h->fq_slot]); if (rc) return rc; }
    return fastq_pipeline(h, total, paired, final_chunk != 0, n_reads_out);
}

extern "C" int mlst_submit_fastq_pair(mlst_handle* h, const uint8_t* text1, uint64_t n1, const uint8_t* text2, uint64_t n2, uint64_t* n_reads_out) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    { int rc_ = bz_flush(h); if (rc_) return rc_; }
    if (n_reads_out) *n_reads_out = 0;
    if (n1 == 0 && n2 == 0) return MLST_OK;
    if (!text1 || !text2 || n1 == 0 || n2 == 0) return fail(h, MLST_E_INVALID, "mate files hold different numbers of records (one chunk is empty)");
    if (h->fq_carry_len) return fail(h, MLST_E_INVALID, "a FASTQ stream is open (its last chunk was not marked final)");
    const u64 pad = text1[n1 - 1] != '\n' ? 1 : 0, total = n1 + pad + n2;
    if (total >= (1ull << 40)) return fail(h, MLST_E_LIMIT, "FASTQ chunk too large");
    hipSetDevice(h->device);
    { int rc = next_text_slot(h, total); if (rc) return rc; }
    if (!h->d_fq_meta) HIPCHK(h, dmalloc(&h->d_fq_meta, (u64)4));
    { int rc = h2d_overlapped(h, h->d_fq_text, text1, n1, h->ev_packed[h->fq_slot]); if (rc) return rc; }
    if (pad) HIPCHK(h, hipMemsetAsync(h->d_fq_text + n1, '\n', 1, h->stream));
    { int rc = h2d_overlapped(h, h->d_fq_text + n1 + pad, text2, n2, nullptr); if (rc) return rc; }
    return fastq_pipeline(h, total, 1, true, n_reads_out, n1 + pad);
}

struct BzHdr { u64 off; u32 total, coff, clen, isize; };      // one BGZF block of a chunk: where it starts, its size, where its deflate data lies, the bytes it inflates to
// BGZF framing (SAM spec 4.1): gzip member with an extra subfield 'B','C' holding the block size - 1; deflate data; CRC32, ISIZE
static bool bgzf_block(const u8* p, u64 left, u64& total, u64& cdata_off, u64& cdata_len, u32& isize) {
    if (left < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return false;
    const u32 xlen = (u32)p[10] | ((u32)p[11] << 8);
    if (left < 12ull + xlen) return false;
    u32 bsize = 0; bool found = false;
    for (u32 o = 0; o + 4 <= xlen; ) {
        const u8* sf = p + 12 + o; const u32 slen = (u32)sf[2] | ((u32)sf[3] << 8);
        if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && o + 6 <= xlen) { bsize = (u32)sf[4] | ((u32)sf[5] << 8); found = true; }
        o += 4 + slen;
    }
    if (!found) return false;
    total = (u64)bsize + 1;
    if (total > left || total < 12ull + xlen + 8) return false;
    cdata_off = 12ull + xlen; cdata_len = total - cdata_off - 8;
    isize = (u32)p[total - 4] | ((u32)p[total - 3] << 8) | ((u32)p[total - 2] << 16) | ((u32)p[total - 1] << 24);
    return true;
}

// The walk over a chunk's block headers is a chain of cache (and TLB) misses, ~190 ns per block (the next header's place is in this
// one): a chunk of 32 MB or more is walked by four threads, each from a block start it FINDS behind its quarter mark (the header's
// fixed bytes, parsed, and the block behind it parsed too); a list is taken only where the chain of the list before it lands exactly
// on its first block -- so a false start (header bytes inside deflate data) costs the time, never the result -- and the caller's
// serial loop goes on from `walked`, where the accepted lists end (all errors are its).  lists_taken: how many of the four counted.
static void bgzf_walk_parallel(const u8* data, u64 n_bytes, std::vector<BzHdr>& hdr, u64& walked, int* lists_taken) {
    hdr.clear(); walked = 0; if (lists_taken) *lists_taken = 0;
    static const bool one = [] { const char* e = getenv("MLST_BGZF_WALK"); return e && e[0] == '1'; }();      // MLST_BGZF_WALK=1: the serial walk (A/B)
    if (n_bytes < (32ull << 20) || one) return;
    enum { WT = 4 };
    std::vector<BzHdr> part[WT]; u64 stop[WT] = {0}, cand[WT] = {0}; bool have[WT] = {false};
    auto work = [&](int k) {
        u64 from = 0;
        if (k) {
            const u64 s0 = n_bytes * (u64)k / WT, s1 = std::min<u64>(s0 + 131072, n_bytes - 18);
            bool ok = false;
            for (u64 q = s0; q < s1 && !ok; q++) {
                if (data[q] != 0x1f || data[q + 1] != 0x8b || data[q + 2] != 8 || !(data[q + 3] & 4)) continue;
                u64 t, co, cl; u32 is;
                if (!bgzf_block(data + q, n_bytes - q, t, co, cl, is)) continue;
                u64 t2, co2, cl2; u32 is2;
                if (q + t != n_bytes && !bgzf_block(data + q + t, n_bytes - q - t, t2, co2, cl2, is2)) continue;
                from = q; ok = true;
            }
            if (!ok) return;
            cand[k] = from; have[k] = true;
        }
        const u64 limit = k + 1 < WT ? std::min<u64>(n_bytes * (u64)(k + 1) / WT + 262144, n_bytes) : n_bytes;
        part[k].reserve((size_t)((limit - from) / 8192 + 64));
        u64 off = from;
        while (off < limit) {
            u64 t, co, cl; u32 is;
            if (!bgzf_block(data + off, n_bytes - off, t, co, cl, is)) break;
            part[k].push_back(BzHdr{off, (u32)t, (u32)co, (u32)cl, is});
            off += t;
        }
        stop[k] = off;
    };
    std::thread th[WT - 1];
    for (int k = 1; k < WT; k++) th[k - 1] = std::thread(work, k);
    work(0);
    for (auto& t : th) t.join();
    hdr.swap(part[0]); walked = stop[0];
    int taken = 1;
    for (int k = 1; k < WT; k++) {
        if (!have[k]) break;
        size_t i = hdr.size();
        while (i > 0 && hdr[i - 1].off > cand[k]) i--;
        if (i == 0 || hdr[i - 1].off != cand[k]) break;      // the chain does not pass through the start this thread found: its list is dropped, and those behind it
        hdr.resize(i - 1);
        hdr.insert(hdr.end(), part[k].begin(), part[k].end());
        walked = stop[k]; taken++;
    }
    if (lists_taken) *lists_taken = taken;
}
// test hook (include/mlst_debug.h): the blocks of a chunk as mlst_submit_fastq_bgzf lists them -- count, text bytes, how many of the four threads' lists counted
extern "C" int mlst_debug_bgzf_walk(const uint8_t* data, uint64_t n_bytes, uint64_t* n_blocks, uint64_t* text_bytes, int* lists_taken) {
    if (!data || !n_blocks || !text_bytes) return MLST_E_INVALID;
    std::vector<BzHdr> hdr; u64 walked = 0;
    bgzf_walk_parallel(data, n_bytes, hdr, walked, lists_taken);
    u64 nb = 0, tb = 0; size_t hi_ = 0;
    for (u64 off = 0; off < n_bytes; ) {
        u64 total, coff, clen; u32 isize;
        if (hi_ < hdr.size()) { const BzHdr& q = hdr[hi_++]; off = q.off; total = q.total; isize = q.isize; }
        else if (off < walked) { off = walked; continue; }
        else if (!bgzf_block(data + off, n_bytes - off, total, coff, clen, isize)) return MLST_E_INVALID;
        if (isize) { nb++; tb += isize; }
        off += total;
    }
    *n_blocks = nb; *text_bytes = tb;
    return MLST_OK;
}

// ---- BGZF input, three stages on three streams (round 5).  Until round 4 a piece went copy -> inflate -> parse -> pass 1 on ONE
// stream with four host synchronisations on the way: 3.8 + 9.5 + 4 ms per 1.07 GB of text one after the other.  Now call k
//   (1) queues the copy of piece k on the copy stream and its inflate on the inflate stream (into a text slot, FQ_HEAD bytes
//       from its start: the partial record the piece before leaves is not known yet and goes in front of it later),
//   (2) finishes piece k - 1: waits for ITS inflate, parses it and queues its pass 1 on the engine's stream -- the host
//       synchronisations of the parser wait for the parser's kernels only, while the inflate of piece k runs beside them,
//   (3) waits for the copy of piece k (the caller owns `data` again on return).
// The last call (final_chunk) finishes its own piece as well; every other entry that looks at the sample's state finishes a
// piece that is still open first (bz_flush).  n_reads_out counts the records completed by the call (those of piece k - 1).
#define FQ_HEAD (1ull << 20)      /* room in front of an inflated piece for the partial record of the piece before it */
static int bz_mode(mlst_handle* h) {
    if (h->bz_mode < 0) { const char* e = getenv("MLST_BGZF_PIPE"); h->bz_mode = (e && e[0] == '0') ? 0 : 1; }
    return h->bz_mode;
}
// parse + pass 1 of the piece whose inflate was queued last; whole: no text follows it
static int bz_finish(mlst_handle* h, bool whole, uint64_t* n_reads_out) {
    if (n_reads_out) *n_reads_out = 0;
    if (!h->bz_pend.on) return MLST_OK;
    h->bz_pend.on = false;
    mlst_handle::BzSlot& B = h->bz[h->bz_pend.slot];
    HIPCHK(h, hipEventSynchronize(B.ev_inflated));
    if (B.h_err[0]) { h->fq_carry_len = 0; return fail(h, MLST_E_INVALID, "corrupt deflate data in BGZF block %u of the chunk (code %u)", B.h_err[0] - 1, B.h_err[1]); }
    const u64 carry = h->fq_carry_len;
    if (carry + 256 > FQ_HEAD) { h->fq_carry_len = 0; return fail(h, MLST_E_LIMIT, "a FASTQ record of more than %llu bytes", (unsigned long long)(FQ_HEAD - 256)); }
    u8* slot = h->d_fq_slot[h->bz_pend.tslot];
    // the parser's blocks are the cells the inflate counted newlines in (FQ_BLOCK bytes from the slot's start on): filler up to the carry
    const u64 s0 = FQ_HEAD - carry, base = s0 & ~(u64)(FQ_BLOCK - 1), skip = s0 - base;
    if (carry) HIPCHK(h, hipMemcpyAsync(slot + s0, h->d_fq_carry, carry, hipMemcpyDeviceToDevice, h->stream));
    if (skip) HIPCHK(h, hipMemsetAsync(slot + base, 'X', skip, h->stream));
    h->d_fq_text = slot + base;
    u32* counts = h->bz_pend.counted ? h->d_fq_nl[h->bz_pend.tslot] + base / FQ_BLOCK : nullptr;
    return fastq_pipeline(h, skip + carry + h->bz_pend.text_bytes, h->bz_pend.paired, whole, n_reads_out, 0, h->bz_pend.tslot, skip, counts, FQ_HEAD - base);
}
// a piece that is still open is finished (its trailing partial record stays in the carry, as after any non-final chunk)
static int bz_flush(mlst_handle* h) { return h->bz_pend.on ? bz_finish(h, false, nullptr) : MLST_OK; }
static void bz_free(mlst_handle* h) {
    if (h->infl_stream) hipStreamSynchronize(h->infl_stream);
    for (auto& B : h->bz) {
        hipFree(B.d_blk); hipFree(B.d_err); if (B.h_err) hipHostFree(B.h_err); if (B.h_blk) hipHostFree(B.h_blk);
        if (B.ev_copied) hipEventDestroy(B.ev_copied); if (B.ev_inflated) hipEventDestroy(B.ev_inflated);
        B = mlst_handle::BzSlot();
    }
    if (h->copy_stream) hipStreamSynchronize(h->copy_stream);
    for (auto& C : h->bzc) {
        hipFree(C.d);
        for (auto& e : C.ev) if (e) hipEventDestroy(e);
        if (C.ev_used) hipEventDestroy(C.ev_used);
        C = mlst_handle::BzChunk();
    }
    if (h->infl_stream) { hipStreamDestroy(h->infl_stream); h->infl_stream = nullptr; }
    h->bz_pend.on = false;
}
// The inflate stream is NOT tied to the engine's CU share (mlst_set_cu_partition): k_inflate_tok decodes one block per LANE and
// is bound by the latency of one wave per CU -- on a quarter of the CUs a piece of 16,384 blocks takes four turns instead of one.
static int bz_stream(mlst_handle* h) {
    if (h->infl_stream) return MLST_OK;
    HIPCHK(h, hipStreamCreateWithFlags(&h->infl_stream, hipStreamNonBlocking));
    return MLST_OK;
}

// One piece through the three stages: its copy and inflate are queued, the piece before it is finished meanwhile (and, for the last
// piece of a stream, the piece itself).  blks: the piece's blocks, in_off relative to `data`, out_off from FQ_HEAD on.
static int bz_piece(mlst_handle* h, mlst_handle::BzChunk& C, u64 lo, u64 hi, const std::vector<BgzfBlk>& blks, u64 text_end, int paired, bool final_piece, uint64_t* done) {
    static const bool bz_trace = getenv("MLST_BGZF_TRACE") != nullptr;      // host-side time stamps of a piece's steps (stderr)
    auto bz_now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double bt0 = bz_trace ? bz_now() : 0.0;
    uint64_t n1 = 0;
    const int sl = h->bz_slot ^= 1;
    mlst_handle::BzSlot& B = h->bz[sl];
    if (!B.ev_copied) { HIPCHK(h, hipEventCreateWithFlags(&B.ev_copied, hipEventDisableTiming)); HIPCHK(h, hipEventCreateWithFlags(&B.ev_inflated, hipEventDisableTiming));
                        HIPCHK(h, dmalloc(&B.d_err, (u64)16)); HIPCHK(h, hipHostMalloc((void**)&B.h_err, 64, hipHostMallocDefault)); }
    // (the slot's last user, piece k - 2, has been finished: its inflate is over)
    if (B.cap_blk < blks.size()) {
        hipFree(B.d_blk); B.d_blk = nullptr; if (B.h_blk) { hipHostFree(B.h_blk); B.h_blk = nullptr; }
        const u64 cap = blks.size() + blks.size() / 8;
        BgzfBlk* pb = nullptr; HIPCHK(h, dmalloc(&pb, cap)); B.d_blk = pb;
        HIPCHK(h, hipHostMalloc(&B.h_blk, cap * sizeof(BgzfBlk), hipHostMallocDefault));
        B.cap_blk = cap;
    }
    { int rc = next_text_slot(h, text_end + text_end / 16); if (rc) return rc; }
    const int tslot = h->fq_slot;
    // (the block list through page-locked memory: a copy from the vector's pageable memory is staged by the runtime on the calling
    // thread; and on the INFLATE stream, in front of the kernels that read it -- the copy stream is busy with the chunk's bytes)
    memcpy(B.h_blk, blks.data(), blks.size() * sizeof(BgzfBlk));
    HIPCHK(h, hipMemcpyAsync(B.d_blk, B.h_blk, blks.size() * sizeof(BgzfBlk), hipMemcpyHostToDevice, h->infl_stream));
    { int k = 0; while (k + 1 < C.n_ev && C.upto[k] < hi) k++;                // the part of the chunk's copy that holds the piece's last byte (parts are queued in order)
      HIPCHK(h, hipStreamWaitEvent(h->infl_stream, C.ev[k], 0)); }
    HIPCHK(h, hipStreamWaitEvent(h->infl_stream, h->ev_packed[tslot], 0));      // the text slot's last reader (an event never recorded counts as complete)
    HIPCHK(h, hipMemsetAsync(B.d_err, 0, 64, h->infl_stream));
    if (!h->inflate_mode) { const char* e = getenv("MLST_INFLATE_MODE"); h->inflate_mode = e ? atoi(e) : 2; if (h->inflate_mode != 1 && h->inflate_mode != 2) h->inflate_mode = 2; }
    const bool counted = h->inflate_mode == 2 && !getenv("MLST_BGZF_NOCOUNT");      // the two-kernel inflate counts the newlines of the text it writes
    if (counted) HIPCHK(h, hipMemsetAsync(h->d_fq_nl[tslot], 0, (text_end / FQ_BLOCK + 2) * 4, h->infl_stream));
    { int rc = launch_inflate(h, C.d + lo, C.cap + 16 - lo, (const BgzfBlk*)B.d_blk, (u32)blks.size(), h->d_fq_slot[tslot], B.d_err, nullptr, h->infl_stream,
                              counted ? h->d_fq_nl[tslot] : nullptr); if (rc) return rc; }
    HIPCHK(h, hipMemcpyAsync(B.h_err, B.d_err, 8, hipMemcpyDeviceToHost, h->infl_stream));
    HIPCHK(h, hipEventRecord(B.ev_inflated, h->infl_stream));
    HIPCHK(h, hipEventRecord(C.ev_used, h->infl_stream)); C.used = true;
    // piece k - 1 while the GPU inflates piece k
    int rc = MLST_OK;
    const double bt1 = bz_trace ? bz_now() : 0.0;
    if (h->bz_pend.on) { rc = bz_finish(h, false, &n1); *done += n1; }
    const double bt2 = bz_trace ? bz_now() : 0.0;
    h->bz_pend.on = true; h->bz_pend.counted = counted; h->bz_pend.slot = sl; h->bz_pend.tslot = tslot; h->bz_pend.paired = paired; h->bz_pend.text_bytes = text_end - FQ_HEAD;
    if (!rc && final_piece) { rc = bz_finish(h, true, &n1); *done += n1; }
    if (bz_trace) fprintf(stderr, "bgzf piece of %zu blocks at %.3f: queueing %.3f ms, piece before %.3f ms, own piece (final) %.3f ms\n",
                          blks.size(), bt0, bt1 - bt0, bt2 - bt1, bz_now() - bt2);
    if (rc) { if (h->infl_stream) hipStreamSynchronize(h->infl_stream); h->bz_pend.on = false; return rc; }
    return MLST_OK;
}

extern "C" int mlst_submit_fastq_bgzf(mlst_handle* h, const uint8_t* data, uint64_t n_bytes, int final_chunk, int paired, uint64_t* n_reads_out,
                                      uint64_t* n_consumed_out) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    if (n_reads_out) *n_reads_out = 0;
    if (n_consumed_out) *n_consumed_out = 0;
    if (n_bytes && !data) return fail(h, MLST_E_INVALID, "NULL argument");
    if (n_bytes >= (1ull << 36)) return fail(h, MLST_E_LIMIT, "BGZF chunk too large");
    hipSetDevice(h->device);
    const bool piped = bz_mode(h) != 0;
    static const bool bz_trace = getenv("MLST_BGZF_TRACE") != nullptr;
    auto bz_now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double ct0 = bz_trace ? bz_now() : 0.0;
    // (piped) the chunk's bytes are on their way before its block headers are looked at: the walk below is a chain of cache misses,
    // one per block (the next header's place is in this one), 8 ms for 49,152 blocks -- time the inflate stream sat idle for
    mlst_handle::BzChunk* C = nullptr;
    if (piped && n_bytes) {
        { int rc = bz_stream(h); if (rc) return rc; }
        if (!h->copy_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        C = &h->bzc[h->bz_chunk ^= 1];
        if (!C->ev_used) { HIPCHK(h, hipEventCreateWithFlags(&C->ev_used, hipEventDisableTiming)); for (auto& e : C->ev) HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
        if (C->cap < n_bytes) {
            if (C->used) HIPCHK(h, hipEventSynchronize(C->ev_used));
            hipFree(C->d); C->d = nullptr; C->cap = 0;
            HIPCHK(h, dmalloc(&C->d, n_bytes + n_bytes / 8 + 32)); C->cap = n_bytes + n_bytes / 8;
        }
        if (C->used) HIPCHK(h, hipStreamWaitEvent(h->copy_stream, C->ev_used, 0));      // the buffer's last reader (the chunk before the last: finished long ago, unless a sample was dropped half way)
        const int n_sub = n_bytes >= (64ull << 20) ? (int)mlst_handle::BZ_SUB : 1;
        u64 at = 0;
        for (int k = 0; k < n_sub; k++) {
            const u64 end = k + 1 == n_sub ? n_bytes : ((n_bytes * (u64)(k + 1) / (u64)n_sub) & ~(u64)4095);
            HIPCHK(h, hipMemcpyAsync(C->d + at, data + at, end - at, hipMemcpyHostToDevice, h->copy_stream));
            HIPCHK(h, hipEventRecord(C->ev[k], h->copy_stream));
            C->upto[k] = end; at = end;
        }
        C->n_ev = n_sub;
    }
    // (an error return below leaves the copy running: the caller's buffer is read until the copy stream is idle -- waited for here)
    struct CopyGuard { mlst_handle* h; bool on; ~CopyGuard() { if (on && h->copy_stream) hipStreamSynchronize(h->copy_stream); } } copy_guard{h, C != nullptr};
    std::vector<BgzfBlk> blks;
    std::vector<u64> blk_start;                                   // (piped) where every listed block begins in `data`
    const u64 text_at = piped ? FQ_HEAD : h->fq_carry_len;      // where the first block's text goes in its slot
    u64 text_bytes = text_at;
    std::vector<BzHdr> hdr; u64 walked = 0;
    bgzf_walk_parallel(data, n_bytes, hdr, walked, nullptr);
    size_t hi_ = 0;      // (headers of `hdr` first, then block by block from `walked`)
    for (u64 off = 0; off < n_bytes; ) {
        u64 total, coff, clen; u32 isize;
        if (hi_ < hdr.size()) { const BzHdr& q = hdr[hi_++]; off = q.off; total = q.total; coff = q.coff; clen = q.clen; isize = q.isize; }
        else if (off < walked) { off = walked; continue; }
        else if (!bgzf_block(data + off, n_bytes - off, total, coff, clen, isize)) {
            // a block cut off by the end of the buffer is left to the caller when it asked how much was taken (and more is to come)
            const bool cut = n_bytes - off < 18 || (data[off] == 0x1f && data[off + 1] == 0x8b && data[off + 2] == 8 && (data[off + 3] & 4));
            if (n_consumed_out && !final_chunk && cut) { n_bytes = off; break; }
            return fail(h, MLST_E_INVALID, "not a whole BGZF block at byte %llu of the chunk", (unsigned long long)off);
        }
        if (isize > 65536) return fail(h, MLST_E_INVALID, "BGZF block at byte %llu claims %u bytes of data", (unsigned long long)off, isize);
        if (isize) { BgzfBlk b; b.in_off = off + coff; b.in_len = (u32)clen; b.out_off = text_bytes; b.out_len = isize; blks.push_back(b); if (piped) blk_start.push_back(off); text_bytes += isize; }
        off += total;
    }
    if (text_bytes >= (1ull << 40)) return fail(h, MLST_E_LIMIT, "FASTQ chunk too large");
    if (n_consumed_out) *n_consumed_out = n_bytes;
    if (!h->d_fq_meta) HIPCHK(h, dmalloc(&h->d_fq_meta, (u64)4));
    if (piped) {
        uint64_t done = 0, n1 = 0;
        if (blks.empty()) {      // nothing new: the open piece (if any) is finished; a last call also types what the carry holds
            if (h->bz_pend.on) { int rc = bz_finish(h, final_chunk != 0, &n1); if (rc) return rc; done += n1; }
            else if (final_chunk && h->fq_carry_len) {
                { int rc = next_text_slot(h, h->fq_carry_len); if (rc) return rc; }
                const u64 carry = h->fq_carry_len;
                HIPCHK(h, hipMemcpyAsync(h->d_fq_text, h->d_fq_carry, carry, hipMemcpyDeviceToDevice, h->stream));
                int rc = fastq_pipeline(h, carry, paired, true, &n1); if (rc) return rc; done += n1;
            }
            if (n_reads_out) *n_reads_out = done;
            return MLST_OK;
        }
        // The chunk goes through the pipeline in pieces.  What the three streams cannot hide is the FIRST piece's copy + inflate (nothing
        // to parse beside them yet) and the LAST piece's parse + pass 1 (nothing inflating beside them any more): so a chunk that
        // meets an empty pipeline leads with a piece of 16,384 blocks (one turn of k_inflate_tok), a last chunk ends with one of
        // 8,192, and what lies between goes in pieces of up to 65,536 (one turn of k_inflate_tok2).  MLST_BGZF_SPLIT=0: the chunk as one piece.
        static const bool split = [] { const char* e = getenv("MLST_BGZF_SPLIT"); return !(e && e[0] == '0'); }();
        const size_t nb = blks.size();
        std::vector<size_t> cuts; cuts.push_back(0);
        if (split) {
            size_t lo = 0, hi = nb;
            if (!h->bz_pend.on && nb > 24576) { lo = 16384; cuts.push_back(lo); }
            const size_t tail = (final_chunk && hi - lo > 16384) ? 8192 : 0;
            while (hi - tail - lo > INFL_PASS2) { lo += INFL_PASS2; cuts.push_back(lo); }
            if (tail) cuts.push_back(hi - tail);
        }
        cuts.push_back(nb);
        for (size_t c = 0; c + 1 < cuts.size(); c++) {
            const size_t b0 = cuts[c], b1 = cuts[c + 1];
            const u64 lo = b0 ? blk_start[b0] : 0, hi = b1 < nb ? blk_start[b1] : n_bytes;
            std::vector<BgzfBlk> sub(blks.begin() + b0, blks.begin() + b1);
            const u64 t0 = sub[0].out_off - FQ_HEAD;
            for (auto& q : sub) { q.in_off -= lo; q.out_off -= t0; }
            const u64 text_end = sub.back().out_off + sub.back().out_len;
            const int rc = bz_piece(h, *C, lo, hi, sub, text_end, paired, final_chunk != 0 && b1 == nb, &done);
            if (rc) return rc;
        }
        if (n_reads_out) *n_reads_out = done;
        const double ct1 = bz_trace ? bz_now() : 0.0;
        copy_guard.on = false;
        HIPCHK(h, hipStreamSynchronize(h->copy_stream));      // `data` may be released by the caller after this
        if (bz_trace) fprintf(stderr, "bgzf chunk of %zu blocks at %.3f: %.3f ms in the call, of which %.3f ms waiting for the copy at its end\n", nb, ct0, bz_now() - ct0, bz_now() - ct1);
        return MLST_OK;
    }
    if (text_bytes == 0) return MLST_OK;
    { int rc = next_text_slot(h, text_bytes); if (rc) return rc; }      // (the inflate kernel writes it on the engine's stream: ordered behind its last reader)
    if (h->fq_carry_len) HIPCHK(h, hipMemcpyAsync(h->d_fq_text, h->d_fq_carry, h->fq_carry_len, hipMemcpyDeviceToDevice, h->stream));
    if (!blks.empty()) {
        if (h->cap_bgzf < n_bytes) { hipFree(h->d_bgzf); h->d_bgzf = nullptr; HIPCHK(h, dmalloc(&h->d_bgzf, n_bytes + 16)); h->cap_bgzf = n_bytes; }
        if (h->cap_bgzf_blk < blks.size()) { hipFree(h->d_bgzf_blk); h->d_bgzf_blk = nullptr; BgzfBlk* pb = nullptr; HIPCHK(h, dmalloc(&pb, (u64)blks.size())); h->d_bgzf_blk = pb; h->cap_bgzf_blk = blks.size(); }
        HIPCHK(h, hipMemcpyAsync(h->d_bgzf, data, n_bytes, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_bgzf_blk, blks.data(), blks.size() * sizeof(BgzfBlk), hipMemcpyHostToDevice, h->stream));
        u32* d_err = reinterpret_cast<u32*>(h->d_fq_meta + 2);
        HIPCHK(h, hipMemsetAsync(d_err, 0, 8, h->stream));
        { int rc = launch_inflate(h, h->d_bgzf, (u64)h->cap_bgzf + 16, (const BgzfBlk*)h->d_bgzf_blk, (u32)blks.size(), h->d_fq_text, d_err, nullptr); if (rc) return rc; }
        u32 err[2] = {0, 0};
        HIPCHK(h, hipMemcpyAsync(err, d_err, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));      // also: blks / data may be released by the caller after this
        if (err[0]) return fail(h, MLST_E_INVALID, "corrupt deflate data in BGZF block %u of the chunk (code %u)", err[0] - 1, err[1]);
    }
    return fastq_pipeline(h, text_bytes, paired, final_chunk != 0, n_reads_out);
}

// k_inflate itself on whole BGZF blocks, text back to the host: the test hook of the DEVICE decoder (tests/test_inflate.py
// compares it with zlib block by block on the GPU box)
extern "C" int mlst_selftest_inflate_device(mlst_handle* h, const uint8_t* data, uint64_t n_bytes, uint8_t* out, uint64_t cap, uint64_t* produced, double* kernel_ms) {
    if (!h) return MLST_E_INVALID;
    if (produced) *produced = 0;
    if (!data || !out) return fail(h, MLST_E_INVALID, "NULL argument");
    hipSetDevice(h->device);
    { int rc_ = bz_flush(h); if (rc_) return rc_; }
    std::vector<BgzfBlk> blks; u64 text_bytes = 0;
    for (u64 off = 0; off < n_bytes; ) {
        u64 total, coff, clen; u32 isize;
        if (!bgzf_block(data + off, n_bytes - off, total, coff, clen, isize)) return fail(h, MLST_E_INVALID, "not a whole BGZF block at byte %llu", (unsigned long long)off);
        if (isize > 65536) return fail(h, MLST_E_INVALID, "BGZF block claims %u bytes of data", isize);
        if (isize) { BgzfBlk b; b.in_off = off + coff; b.in_len = (u32)clen; b.out_off = text_bytes; b.out_len = isize; blks.push_back(b); text_bytes += isize; }
        off += total;
    }
    if (text_bytes > cap) return fail(h, MLST_E_LIMIT, "output buffer too small (%llu bytes needed)", (unsigned long long)text_bytes);
    if (produced) *produced = text_bytes;
    if (blks.empty()) return MLST_OK;
    u8* d_in = nullptr; u8* d_out = nullptr; BgzfBlk* d_blk = nullptr; u32* d_err = nullptr; unsigned long long* d_st = nullptr;
    int rc = MLST_OK; u32 err[2] = {0, 0};
#if defined(MLST_INFLATE_STATS)
    const bool want_stats = getenv("MLST_INFLATE_STATS") != nullptr;      // (diagnostic builds: hipcc -DMLST_INFLATE_STATS)
#else
    const bool want_stats = false;
#endif
    if (want_stats && (dmalloc(&d_st, (u64)16) != hipSuccess || hipMemset(d_st, 0, 128) != hipSuccess)) d_st = nullptr;
    if (dmalloc(&d_in, n_bytes + 16) != hipSuccess || dmalloc(&d_out, text_bytes + 16) != hipSuccess || dmalloc(&d_blk, (u64)blks.size()) != hipSuccess || dmalloc(&d_err, (u64)8) != hipSuccess)
        rc = fail(h, MLST_E_HIP, "device allocation failed");
    if (!rc && (hipMemcpy(d_in, data, n_bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_blk, blks.data(), blks.size() * sizeof(BgzfBlk), hipMemcpyHostToDevice) != hipSuccess
                || hipMemset(d_err, 0, 32) != hipSuccess || hipMemset(d_out, 0xEE, text_bytes) != hipSuccess)) rc = fail(h, MLST_E_HIP, "copy to the device failed");
    if (!rc) {
        hipEvent_t e0 = ev_get(h), e1 = ev_get(h);
        hipEventRecord(e0, h->stream);
        rc = launch_inflate(h, d_in, (u64)n_bytes + 16, (const BgzfBlk*)d_blk, (u32)blks.size(), d_out, d_err, d_st);
        hipEventRecord(e1, h->stream);
        hipError_t se = hipStreamSynchronize(h->stream);
        float ms = 0; if (se == hipSuccess) hipEventElapsedTime(&ms, e0, e1);
        if (kernel_ms) *kernel_ms = (double)ms;
        h->ev_pool.push_back(e0); h->ev_pool.push_back(e1);
        if (se != hipSuccess || hipMemcpy(err, d_err, 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(out, d_out, text_bytes, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(h, MLST_E_HIP, "k_inflate failed: %s", hipGetErrorString(hipGetLastError()));
        else if (err[0]) rc = fail(h, MLST_E_INVALID, "corrupt deflate data in BGZF block %u (code %u)", err[0] - 1, err[1]);
#if defined(MLST_PTR_TRACE)
        { u32 tr[8] = {0}; if (hipMemcpy(tr, d_err, 32, hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "k_inflate_ptr phases over %zu blocks (units of 64 cycles): fill %u, pointer jumping %u (%u rounds), gather %u\n", blks.size(), tr[2], tr[3], tr[5], tr[4]); }
#endif
    }
    if (d_st && !rc) {
        unsigned long long v[10] = {0};
        if (hipMemcpy(v, d_st, sizeof v, hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "k_inflate stats over %zu blocks: look-ups %llu, literal bytes %llu, near matches %llu, far matches deferred %llu / at once %llu, far flushes %llu, fences %llu, "
                            "table builds %llu, cycles in builds %llu / in codes %llu (sums over waves)\n", blks.size(), v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9]);
    }
    hipFree(d_in); hipFree(d_out); hipFree(d_blk); hipFree(d_err); hipFree(d_st);
    return rc;
}

// the decoder of k_inflate_tok2 (csrc/inflate_canon.h) run on the host, its tokens replayed into bytes: a test hook
// (*left_to_other_kernel = 1: the block's literal / length code holds more symbols than the 192-entry table: k_inflate takes it)
extern "C" int mlst_selftest_inflate_canon(const uint8_t* in, uint64_t n_in, uint8_t* out, uint64_t cap, uint64_t* produced, int* left_to_other_kernel) {
    unsigned long long p = 0; bool over = false;
    const int rc = inflate_canon::inflate_raw_host(in, n_in, out, cap, &p, &over);
    if (produced) *produced = p;
    if (left_to_other_kernel) *left_to_other_kernel = over ? 1 : 0;
    return rc;
}
// the decoder of k_inflate run on the host: a test hook (tests/test_inflate.py compares it with zlib without a GPU)
extern "C" int mlst_selftest_inflate(const uint8_t* in, uint64_t n_in, uint8_t* out, uint64_t cap, uint64_t* produced) {
    uint64_t p = 0; mlst_inflate::Tables tb; const int rc = mlst_inflate::inflate_raw(in, n_in, out, cap, &p, &tb);
    if (produced) *produced = p;
    return rc;
}

// one D2H copy of the whole statistics block into pinned memory, one synchronisation
static int fetch_stats(mlst_handle* h, Counters** c_out) {
    HIPCHK(h, hipMemcpyAsync(h->h_stats, h->d_stats, h->stats_bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    Counters* c = (Counters*)(h->h_stats + h->off_ctr);
    if (c_out) *c_out = c;
    if (c->err) return fail(h, MLST_E_CAPACITY, "capacity exceeded (flags 0x%llx: 1=retained reads %llu/%llu, 2=items %llu/%llu, 4=pair results %llu/%llu, 8=banded-SW list); raise mlst_params.max_*",
                            (unsigned long long)c->err, (unsigned long long)c->n_ret, (unsigned long long)h->E.cap_ret, (unsigned long long)c->n_items, (unsigned long long)h->E.cap_items,
                            (unsigned long long)c->n_res, (unsigned long long)h->E.cap_res);
    return MLST_OK;
}
static int check_overflow(mlst_handle* h) { return fetch_stats(h, nullptr); }
// in-kernel execution window of the sample's sieve launch (one submission per sample), slot 7 of the kernel times
static void note_sieve_window(mlst_handle* h, const Counters* c) {
    if (!h->window || !c->sv_t1 || !c->sv_t0n) return;
    const u64 t0 = ~c->sv_t0n;
    if (c->sv_t1 > t0) { h->k_ms[7] += (double)(c->sv_t1 - t0) / h->wall_khz; h->k_n[7]++; }
    if (c->sv_wgmax) { h->k_ms[8] += (double)c->sv_wgmax / h->wall_khz; h->k_n[8]++; }
}

extern "C" int mlst_get_allele_stats(mlst_handle* h, int64_t* sum_score, uint32_t* n_hits, uint64_t* locus_len,
                                     uint64_t* locus_first, uint64_t* counters) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    { int rc_ = bz_flush(h); if (rc_) return rc_; }
    hipSetDevice(h->device);
    Counters* c = nullptr;
    int rc = fetch_stats(h, &c); if (rc) return rc;
    note_sieve_window(h, c);
    if (sum_score) memcpy(sum_score, h->h_stats + h->off_sum, (u64)h->n_alleles * 8);
    if (n_hits) memcpy(n_hits, h->h_stats + h->off_hits, (u64)h->n_alleles * 4);
    if (locus_len) memcpy(locus_len, h->h_stats + h->off_len, (u64)h->n_loci * 8);
    if (locus_first) memcpy(locus_first, h->h_stats + h->off_first, (u64)h->n_loci * 8);
    if (counters) { for (int i = 0; i < MLST_CNT_N; i++) counters[i] = c->cnt[i]; counters[MLST_CNT_RETAINED] = c->n_ret; counters[MLST_CNT_ITEMS] = c->n_items; }
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
    { int rc_ = bz_flush(h); if (rc_) return rc_; }
    hipSetDevice(h->device);
    int rc = check_overflow(h); if (rc) return rc;
    hipLaunchKernelGGL(k_export, dim3(256), dim3(256), 0, h->stream, h->d_E, (long long*)d_sum, (long long*)d_min);
    HIPCHK(h, hipGetLastError()); HIPCHK(h, hipStreamSynchronize(h->stream));
    return MLST_OK;
}
extern "C" int mlst_import_stats_device(mlst_handle* h, const int64_t* d_sum, const int64_t* d_min) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    hipSetDevice(h->device);
    hipLaunchKernelGGL(k_import, dim3(256), dim3(256), 0, h->stream, h->d_E, (const long long*)d_sum, (const long long*)d_min);
    HIPCHK(h, hipGetLastError()); HIPCHK(h, hipStreamSynchronize(h->stream));
    return MLST_OK;
}

static int ensure_pin(mlst_handle* h, u64 bytes) {
    if (h->cap_pin >= bytes) return MLST_OK;
    hipStreamSynchronize(h->stream);
    if (h->h_pin) hipHostFree(h->h_pin);
    h->h_pin = nullptr; h->cap_pin = 0;
    HIPCHK(h, hipHostMalloc((void**)&h->h_pin, bytes, hipHostMallocDefault));
    h->cap_pin = bytes;
    return MLST_OK;
}

// k_pileup: one wave per workgroup, 8192 of them (64 VGPRs-class kernels fill the chip's wave slots with that many); the
// instantiation follows the widest read rows submitted for the sample.
static void launch_pileup(mlst_handle* h, const int* d_chosen, const u64* d_colbase, u32* d_counts) {
    if (h->max_wpr <= 10) hipLaunchKernelGGL(k_pileup_160, dim3(8192), dim3(64), 0, h->stream, h->d_E, h->kp, d_chosen, d_colbase, d_counts, h->d_pl_list);
    else hipLaunchKernelGGL(k_pileup_320, dim3(8192), dim3(64), 0, h->stream, h->d_E, h->kp, d_chosen, d_colbase, d_counts, h->d_pl_list);
}
// Ungapped + banded pile-up of every item against its locus' chosen allele into d_counts (zeroed by the caller, n_pl_dp
// too).  With a depth cap set (mlst_set_depth_cap; default off) the pile-up is the last of 42 passes: 41 counting passes
// of a bitwise search find, per column, the key of the cap-th record that spans it (k_cap_search), and the pile-up proper
// is restricted to the records each column sees.  ~7 ms instead of ~0.2: not a fast path, and an order-free APPROXIMATION of
// pysam's max_depth (htslib drops whole reads at their start position in coordinate-sorted order: DESIGN.md section 6), not bit-identical to it.
static int pile_all(mlst_handle* h, const int* d_chosen, const u64* d_colbase, u32* d_counts, u64 ncols) {
    if (!h->depth_cap) {
        launch_pileup(h, d_chosen, d_colbase, d_counts);
        hipLaunchKernelGGL(k_pileup_dp, dim3(64), dim3(64), 0, h->stream, h->d_E, h->kp, d_chosen, d_colbase, d_counts, h->d_pl_list, h->d_tb, (const u64*)nullptr, 0u);
        return MLST_OK;
    }
    if (ncols == 0) return MLST_OK;
    if (h->cap_capcols < ncols) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        hipFree(h->d_capbuf); h->d_capbuf = nullptr; h->cap_capcols = 0;
        HIPCHK(h, hipMalloc((void**)&h->d_capbuf, ncols * 28));
        h->cap_capcols = ncols;
    }
    u64* lo = h->d_capbuf; u64* hi = lo + h->cap_capcols; u64* thr = hi + h->cap_capcols; u32* cnt = reinterpret_cast<u32*>(thr + h->cap_capcols);
    const unsigned gs = grid_for(ncols, 256, 1024);
    auto pass = [&](u32* out, u32 mode) {
        if (h->max_wpr <= 10) hipLaunchKernelGGL(k_pileup_cap_160, dim3(8192), dim3(64), 0, h->stream, h->d_E, h->kp, d_chosen, d_colbase, out, h->d_pl_list, (const u64*)thr, mode);
        else hipLaunchKernelGGL(k_pileup_cap_320, dim3(8192), dim3(64), 0, h->stream, h->d_E, h->kp, d_chosen, d_colbase, out, h->d_pl_list, (const u64*)thr, mode);
        hipLaunchKernelGGL(k_pileup_dp, dim3(64), dim3(64), 0, h->stream, h->d_E, h->kp, d_chosen, d_colbase, out, h->d_pl_list, h->d_tb, (const u64*)thr, mode);
    };
    hipLaunchKernelGGL(k_cap_search, dim3(gs), dim3(256), 0, h->stream, lo, hi, thr, cnt, (u64)ncols, h->depth_cap, 0);
    for (int it = 0; it < CAP_KEY_BITS; it++) {
        pass(cnt, 1u | (it ? 4u : 0u));
        hipLaunchKernelGGL(k_cap_search, dim3(gs), dim3(256), 0, h->stream, lo, hi, thr, cnt, (u64)ncols, h->depth_cap, it == CAP_KEY_BITS - 1 ? 2 : 1);
    }
    pass(d_counts, 2u | 4u);
    HIPCHK(h, hipGetLastError());
    return MLST_OK;
}
// Policy MLST_DEPTH_CAP as a switch (pysam pileup(max_depth=8000), metaMLST_functions.py:255-259): 0 = off (the default: every
// record counts), n = a column sees the first n records that span it, in (read index, strand) order.  Takes effect with the
// next pile-up; single-engine samples only (a rank of a sharded sample sees only its own reads' records).
extern "C" int mlst_set_depth_cap(mlst_handle* h, uint32_t cap) {
    if (!h) return MLST_E_INVALID;
    hipSetDevice(h->device);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->depth_cap = cap;
    if (h->g_typing.exec) { hipGraphExecDestroy(h->g_typing.exec); h->g_typing.exec = nullptr; }
    h->g_typing.sig.clear();
    return MLST_OK;
}

// launches pass 2; tables go through pinned memory (laid out [colbase u64 x L][chosen int x L]); no sync here
static int pileup_launch(mlst_handle* h, const uint32_t* chosen, uint32_t n, uint32_t* d_counts, u64 counts_tail_bytes, uint64_t* n_cols_out) {
    const u64 nl = h->n_loci, tab_bytes = nl * 12;
    int rc = ensure_pin(h, tab_bytes + counts_tail_bytes + 128); if (rc) return rc;
    u64* cb = (u64*)h->h_pin; int* lc = (int*)(h->h_pin + nl * 8);
    for (u64 l = 0; l < nl; l++) { lc[l] = -1; cb[l] = 0; }
    u64 ncols = 0;
    for (u32 k = 0; k < n; k++) {
        u32 a = chosen[k]; if (a >= h->n_alleles) return fail(h, MLST_E_INVALID, "chosen allele %u out of range", a);
        u32 L = h->allele_locus[a]; if (lc[L] >= 0) return fail(h, MLST_E_INVALID, "two chosen alleles for locus %u", L);
        lc[L] = (int)a; cb[L] = ncols; ncols += h->aoff[a + 1] - h->aoff[a];
    }
    if (n_cols_out) *n_cols_out = ncols;
    HIPCHK(h, hipMemcpyAsync(h->d_locus_colbase, h->h_pin, tab_bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(d_counts, 0, (ncols ? ncols : 1) * 16, h->stream));
    HIPCHK(h, hipMemsetAsync(&h->E.ctr.p->n_pl_dp, 0, 8, h->stream));
    const int* d_lc = (const int*)((u8*)h->d_locus_colbase + nl * 8);
    { Prof pf(h, 5);
      rc = pile_all(h, d_lc, h->d_locus_colbase, d_counts, ncols); if (rc) return rc; }
    HIPCHK(h, hipGetLastError());
    return MLST_OK;
}

extern "C" int mlst_pileup_device(mlst_handle* h, const uint32_t* chosen, uint32_t n, uint32_t* d_counts, uint64_t* n_cols_out) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    { int rc_ = bz_flush(h); if (rc_) return rc_; }
    hipSetDevice(h->device);
    int rc = pileup_launch(h, chosen, n, d_counts, 0, n_cols_out); if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MLST_OK;
}

extern "C" int mlst_pileup(mlst_handle* h, const uint32_t* chosen, uint32_t n, uint32_t* counts) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    { int rc_ = bz_flush(h); if (rc_) return rc_; }
    hipSetDevice(h->device);
    u64 ncols = 0;
    for (u32 k = 0; k < n; k++) { if (chosen[k] >= h->n_alleles) return fail(h, MLST_E_INVALID, "chosen allele out of range"); ncols += h->aoff[chosen[k] + 1] - h->aoff[chosen[k]]; }
    if (h->cap_counts < ncols * 4 + 4) { hipStreamSynchronize(h->stream); hipFree(h->d_counts); h->d_counts = nullptr; HIPCHK(h, dmalloc(&h->d_counts, ncols * 4 + 4)); h->cap_counts = ncols * 4 + 4; }
    uint64_t nc2 = 0; int rc = pileup_launch(h, chosen, n, h->d_counts, ncols * 16, &nc2); if (rc) return rc;
    u8* stage = h->h_pin + (u64)h->n_loci * 12 + 16; stage = (u8*)(((uintptr_t)stage + 15) & ~(uintptr_t)15);
    if (ncols) HIPCHK(h, hipMemcpyAsync(stage, h->d_counts, ncols * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (ncols) memcpy(counts, stage, ncols * 16);
    return MLST_OK;
}

extern "C" int mlst_consensus(mlst_handle* h, const uint32_t* chosen, uint32_t n, uint32_t mincov, char none_char,
                              uint8_t* out_seq, uint32_t* counts) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    { int rc_ = bz_flush(h); if (rc_) return rc_; }
    if (!out_seq) return fail(h, MLST_E_INVALID, "out_seq is NULL");
    hipSetDevice(h->device);
    u64 ncols = 0;
    for (u32 k = 0; k < n; k++) { if (chosen[k] >= h->n_alleles) return fail(h, MLST_E_INVALID, "chosen allele out of range"); ncols += h->aoff[chosen[k] + 1] - h->aoff[chosen[k]]; }
    const u64 need = ncols * 4 + 4 + (ncols + 15) / 4 + 8;       // counts + letters (u32 units)
    if (h->cap_counts < need) { hipStreamSynchronize(h->stream); hipFree(h->d_counts); h->d_counts = nullptr; HIPCHK(h, dmalloc(&h->d_counts, need)); h->cap_counts = need; }
    uint64_t nc2 = 0; int rc = pileup_launch(h, chosen, n, h->d_counts, ncols * 17 + 64, &nc2); if (rc) return rc;
    u8* d_letters = reinterpret_cast<u8*>(h->d_counts + ncols * 4 + 4);
    if (ncols) hipLaunchKernelGGL(k_consensus, dim3(grid_for(ncols, 256, 256)), dim3(256), 0, h->stream, h->d_counts, (u64)ncols, mincov, (u8)none_char, d_letters);
    HIPCHK(h, hipGetLastError());
    u8* stage = h->h_pin + (u64)h->n_loci * 12 + 16; stage = (u8*)(((uintptr_t)stage + 15) & ~(uintptr_t)15);
    if (ncols) HIPCHK(h, hipMemcpyAsync(stage, d_letters, ncols, hipMemcpyDeviceToHost, h->stream));
    if (ncols && counts) HIPCHK(h, hipMemcpyAsync(stage + ((ncols + 15) & ~15ull), h->d_counts, ncols * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (ncols) memcpy(out_seq, stage, ncols);
    if (ncols && counts) memcpy(counts, stage + ((ncols + 15) & ~15ull), ncols * 16);
    return MLST_OK;
}

extern "C" int mlst_consensus_from_counts_device(mlst_handle* h, const uint32_t* d_counts, uint64_t n_cols, uint32_t mincov,
                                                 char none_char, uint8_t* out_seq) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    if (n_cols == 0) return MLST_OK;
    if (!d_counts || !out_seq) return fail(h, MLST_E_INVALID, "NULL argument");
    hipSetDevice(h->device);
    const u64 need = (n_cols + 15) / 4 + 8;
    if (h->cap_counts < need) { hipStreamSynchronize(h->stream); hipFree(h->d_counts); h->d_counts = nullptr; HIPCHK(h, dmalloc(&h->d_counts, need)); h->cap_counts = need; }
    int rc = ensure_pin(h, (u64)h->n_loci * 12 + n_cols + 128); if (rc) return rc;
    u8* d_letters = reinterpret_cast<u8*>(h->d_counts);
    hipLaunchKernelGGL(k_consensus, dim3(grid_for(n_cols, 256, 256)), dim3(256), 0, h->stream, d_counts, (u64)n_cols, mincov, (u8)none_char, d_letters);
    HIPCHK(h, hipGetLastError());
    u8* stage = h->h_pin + (u64)h->n_loci * 12 + 16; stage = (u8*)(((uintptr_t)stage + 15) & ~(uintptr_t)15);
    HIPCHK(h, hipMemcpyAsync(stage, d_letters, n_cols, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    memcpy(out_seq, stage, n_cols);
    return MLST_OK;
}

extern "C" int mlst_pileup_alignments(mlst_handle* h, const uint32_t* chosen, uint32_t n_chosen, uint64_t n_rec,
                                      const uint32_t* rec_allele, const int32_t* rec_pos0, const int32_t* rec_as, const int32_t* rec_xm,
                                      const uint64_t* cigar_off, const uint32_t* cigar, const uint64_t* seq_off,
                                      const uint8_t* seq, const uint8_t* qual, int32_t minscore, int32_t max_xm, int32_t minqual,
                                      uint32_t* counts) {
    if (!h || !h->have_ref) return fail(h, MLST_E_INVALID, "no reference loaded");
    hipSetDevice(h->device);
    std::vector<int> slot(h->n_alleles, -1); u64 ncols = 0;
    for (u32 k = 0; k < n_chosen; k++) {
        if (chosen[k] >= h->n_alleles) return fail(h, MLST_E_INVALID, "chosen allele %u out of range", chosen[k]);
        if (slot[chosen[k]] >= 0) return fail(h, MLST_E_INVALID, "allele %u chosen twice", chosen[k]);
        if (ncols > 0x7FFFFFFFull) return fail(h, MLST_E_LIMIT, "more than 2^31 pileup columns");
        slot[chosen[k]] = (int)ncols; ncols += h->aoff[chosen[k] + 1] - h->aoff[chosen[k]];
    }
    for (u64 k = 0; k < n_rec; k++) if (rec_allele[k] >= h->n_alleles) return fail(h, MLST_E_INVALID, "record %llu: allele out of range", (unsigned long long)k);
    const u64 n_cig = n_rec ? cigar_off[n_rec] : 0, n_seq = n_rec ? seq_off[n_rec] : 0;
    u32 *d_ra = nullptr, *d_cig = nullptr, *d_cnt = nullptr; int *d_pos = nullptr, *d_as = nullptr, *d_xm = nullptr, *d_slot = nullptr;
    u64 *d_co = nullptr, *d_so = nullptr; u8 *d_seq = nullptr, *d_q = nullptr;
    auto cleanup = [&]() { hipFree(d_ra); hipFree(d_cig); hipFree(d_cnt); hipFree(d_pos); hipFree(d_as); hipFree(d_xm); hipFree(d_slot);
                           hipFree(d_co); hipFree(d_so); hipFree(d_seq); hipFree(d_q); };
#define UP(dst, src, n, T) do { if (dmalloc(&dst, (u64)(n)) != hipSuccess || ((n) && hipMemcpyAsync(dst, src, (u64)(n) * sizeof(T), hipMemcpyHostToDevice, h->stream) != hipSuccess)) \
                                { cleanup(); return fail(h, MLST_E_HIP, "upload of the alignment arrays failed"); } } while (0)
    UP(d_ra, rec_allele, n_rec, u32); UP(d_pos, rec_pos0, n_rec, int); UP(d_as, rec_as, n_rec, int); UP(d_xm, rec_xm, n_rec, int);
    UP(d_co, cigar_off, n_rec + 1, u64); UP(d_cig, cigar, n_cig, u32); UP(d_so, seq_off, n_rec + 1, u64);
    UP(d_seq, seq, n_seq, u8); UP(d_q, qual, n_seq, u8); UP(d_slot, slot.data(), h->n_alleles, int);
#undef UP
    if (dmalloc(&d_cnt, (ncols ? ncols : 1) * 4) != hipSuccess) { cleanup(); return fail(h, MLST_E_HIP, "out of device memory"); }
    hipMemsetAsync(d_cnt, 0, (ncols ? ncols : 1) * 16, h->stream);
    if (n_rec && ncols)
        hipLaunchKernelGGL(k_pileup_aln, dim3(grid_for(n_rec, 256, 4096)), dim3(256), 0, h->stream, (u64)n_rec, d_ra, d_pos, d_as, d_xm, d_co, d_cig,
                           d_so, d_seq, d_q, d_slot, h->d_aoff, (int)minscore, (int)max_xm, (int)minqual, d_cnt);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && ncols) e = hipMemcpyAsync(counts, d_cnt, ncols * 16, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    cleanup();
    if (e != hipSuccess) return fail(h, MLST_E_HIP, "mlst_pileup_alignments: %s", hipGetErrorString(e));
    return MLST_OK;
}

extern "C" long long mlst_round_tenths(long long p, uint32_t q) { return q ? round_tenths(p, q) : 0; }

extern "C" int mlst_typing_layout(mlst_handle* h, uint64_t* colbase, uint64_t* total_cols) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    if (colbase) memcpy(colbase, h->fixed_colbase.data(), ((u64)h->n_loci + 1) * 8);
    if (total_cols) *total_cols = h->fixed_cols;
    return MLST_OK;
}

// Asynchronous, on the engine's stream, behind whatever pass 1 has been submitted.  Phase 1: allele choice and pileup
// against the chosen alleles into d_counts (NULL = the engine's own buffer; a caller's buffer of total_cols*4 uint32 lets
// a multi-GPU caller all-reduce the counts between the phases).  Phase 2: majority consensus over those counts and the
// copies of statistics, choice and consensus into pinned memory.
extern "C" int mlst_typing_choose_pileup(mlst_handle* h, int32_t penalty, uint32_t* d_counts) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    { int rc_ = bz_flush(h); if (rc_) return rc_; }
    hipSetDevice(h->device);
    const u64 nl = h->n_loci, ncols = h->fixed_cols;
    u32* cnt = d_counts ? d_counts : h->d_auto_counts;
    if (nl) hipLaunchKernelGGL(k_choose, dim3((unsigned)nl), dim3(256), 0, h->stream, h->d_E, h->d_allele_no, (int)penalty, h->d_auto_chosen);
    zero_words(h, cnt, (ncols ? ncols : 1) * 4);
    { Prof pf(h, 5);
      const int rc = pile_all(h, h->d_auto_chosen, h->d_fixed_colbase, cnt, ncols); if (rc) return rc; }
    HIPCHK(h, hipGetLastError());
    return MLST_OK;
}
// the kernels of phase 2 (inside the replayed graph) and its copies into pinned memory (issued directly, every time)
static int typing_finish_kernels(mlst_handle* h, uint32_t mincov, char none_char, const uint32_t* d_counts) {
    const u64 ncols = h->fixed_cols;
    const u32* cnt = d_counts ? d_counts : h->d_auto_counts;
    if (ncols) hipLaunchKernelGGL(k_consensus, dim3(grid_for(ncols, 256, 256)), dim3(256), 0, h->stream, cnt, (u64)ncols, mincov, (u8)none_char, h->d_auto_letters);
    HIPCHK(h, hipGetLastError());
    return MLST_OK;
}
static int typing_finish_copies(mlst_handle* h) {
    const u64 nl = h->n_loci, ncols = h->fixed_cols;
    u8* hs = h->h_tstats[h->t_slot]; u8* ha = h->h_tauto[h->t_slot];
    HIPCHK(h, hipMemcpyAsync(hs, h->d_stats, h->stats_bytes, hipMemcpyDeviceToHost, h->stream));
    if (nl) HIPCHK(h, hipMemcpyAsync(ha, h->d_auto_chosen, nl * 4, hipMemcpyDeviceToHost, h->stream));
    if (ncols) HIPCHK(h, hipMemcpyAsync(ha + ((nl * 4 + 15) & ~15ull), h->d_auto_letters, ncols, hipMemcpyDeviceToHost, h->stream));
    h->auto_pending = true; h->t_last = h->t_slot; h->t_slot ^= 1;
    return MLST_OK;
}
extern "C" int mlst_typing_finish(mlst_handle* h, uint32_t mincov, char none_char, const uint32_t* d_counts) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    hipSetDevice(h->device);
    int rc = typing_finish_kernels(h, mincov, none_char, d_counts);
    return rc ? rc : typing_finish_copies(h);
}
// The same two halves with the counts in the compact layout of k_layout_compact (what a multi-GPU caller all-reduces
// between them): d_counts holds cap_cols * 4 uint32.  mlst_typing_compact_info, after mlst_typing_fetch, says how many
// columns were needed and whether they fitted; if not, no letter of the fetch is valid and the caller runs both halves
// again with a buffer of at least that many columns (statistics and choice are not touched by the repeat).
extern "C" int mlst_typing_choose_pileup_compact(mlst_handle* h, int32_t penalty, uint32_t* d_counts, uint64_t cap_cols) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    { int rc_ = bz_flush(h); if (rc_) return rc_; }
    if (!d_counts || cap_cols == 0) return fail(h, MLST_E_INVALID, "mlst_typing_choose_pileup_compact needs a counts buffer");
    hipSetDevice(h->device);
    const u64 nl = h->n_loci;
    if (nl) hipLaunchKernelGGL(k_choose, dim3((unsigned)nl), dim3(256), 0, h->stream, h->d_E, h->d_allele_no, (int)penalty, h->d_auto_chosen);
    hipLaunchKernelGGL(k_layout_compact, dim3(1), dim3(1024), 0, h->stream, h->d_auto_chosen, h->d_fixed_colbase, (u32)nl, (u64)cap_cols,
                       h->d_compact_colbase, h->d_compact_chosen, h->d_compact_info);
    zero_words(h, d_counts, cap_cols * 4);
    { Prof pf(h, 5);
      const int rc = pile_all(h, h->d_compact_chosen, h->d_compact_colbase, d_counts, cap_cols); if (rc) return rc; }
    HIPCHK(h, hipGetLastError());
    return MLST_OK;
}
extern "C" int mlst_typing_finish_compact(mlst_handle* h, uint32_t mincov, char none_char, const uint32_t* d_counts) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    if (!d_counts) return fail(h, MLST_E_INVALID, "mlst_typing_finish_compact needs the counts buffer");
    hipSetDevice(h->device);
    if (h->n_loci) hipLaunchKernelGGL(k_consensus_expand, dim3((unsigned)h->n_loci), dim3(256), 0, h->stream, d_counts, h->d_compact_chosen, h->d_compact_colbase,
                                      h->d_fixed_colbase, mincov, (u8)none_char, h->d_auto_letters);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(h->h_tauto[h->t_slot] + h->off_compact_info, h->d_compact_info, 16, hipMemcpyDeviceToHost, h->stream));      // (the slot typing_finish_copies is about to fill)
    h->compact_pending = true;
    return typing_finish_copies(h);
}
extern "C" int mlst_typing_compact_info(mlst_handle* h, uint64_t* need_cols, uint32_t* overflow) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    if (!h->compact_pending || h->t_ready < 0) return fail(h, MLST_E_INVALID, "mlst_typing_compact_info follows mlst_typing_finish_compact and mlst_typing_fetch");
    const u64* info = (const u64*)(h->h_tauto[h->t_ready] + h->off_compact_info);
    if (need_cols) *need_cols = info[0];
    if (overflow) *overflow = (uint32_t)info[1];
    return MLST_OK;
}
extern "C" int mlst_typing_enqueue(mlst_handle* h, int32_t penalty, uint32_t mincov, char none_char) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    { int rc_ = bz_flush(h); if (rc_) return rc_; }
    hipSetDevice(h->device);
    // (the depth-capped pile-up may size its buffers on the way and is not a fast path: launched directly)
    const int gs = h->depth_cap ? 0 : graph_enter(h, h->g_typing, {(u64)(u32)penalty, (u64)mincov, (u64)(u8)none_char, (u64)(h->max_wpr <= 10)});
    int rc = MLST_OK;
    if (gs != 1) {
        rc = mlst_typing_choose_pileup(h, penalty, nullptr);
        if (!rc) rc = typing_finish_kernels(h, mincov, none_char, nullptr);
        if (gs == 2) { int rc2 = graph_leave(h, h->g_typing); if (!rc) rc = rc2; }
    }
    return rc ? rc : typing_finish_copies(h);        // the copies stay outside the graph (kernel nodes only, see k_zero)
}

// CU partitions.  The engine's own stream is re-created with a CU mask: part `part` of `n_parts` equal shares of the device's
// CUs (n_parts = 1: the whole device again).  Several engines of one process, each on its own share, run their launch
// sequences side by side -- every kernel of the path that waits (for memory round trips, for the device's addition rate, for
// one long DP chain) waits beside the other engines' kernels instead of in front of them, and four engines on a quarter
// each type 6 % more reads per second than four engines that take turns on the whole device.
// Everything queued so far is waited for first; a caller's stream (mlst_set_stream) is left alone.
extern "C" int mlst_set_cu_partition(mlst_handle* h, uint32_t part, uint32_t n_parts) {
    if (!h) return MLST_E_INVALID;
    if (n_parts == 0 || part >= n_parts) return fail(h, MLST_E_INVALID, "CU partition %u of %u", part, n_parts);
    if (n_parts == 1 && h->cu_split == 1) return MLST_OK;      // the whole device already: keep the stream (a fresh one may share a hardware queue with another engine's)
    if (h->cu_split == (int)n_parts && h->cu_part == (int)part && h->own_stream) return MLST_OK;      // this share already (a second run over the same engines: a masked stream costs ~5 ms to make)
    hipSetDevice(h->device);
    { int rc_ = bz_flush(h); if (rc_) return rc_; }
    drain_events(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int n_cu_i = 0; HIPCHK(h, hipDeviceGetAttribute(&n_cu_i, hipDeviceAttributeMultiprocessorCount, h->device));      // (hipGetDeviceProperties takes milliseconds)
    const u32 n_cu = (u32)n_cu_i;
    if (n_parts > n_cu) return fail(h, MLST_E_INVALID, "more CU partitions (%u) than CUs (%u)", n_parts, n_cu);
    hipStream_t ns = nullptr;
    if (n_parts == 1) HIPCHK(h, hipStreamCreate(&ns));
    else {
        std::vector<uint32_t> mask((n_cu + 31) / 32, 0u);
        // (shares of whole XCDs -- 2, 4 or 8 parts of the 8 x 32 CUs -- are the ones that pay: the dispatcher deals workgroups
        // to the XCDs in turn, and a share with 32 CUs on one XCD and 19 on the next runs at the pace of the 19: thirds of
        // the device were 15 % slower than no partition at all, fifths 85 %; shares that overlap their neighbours 5-8 %)
        const u32 lo = (u32)((u64)n_cu * part / n_parts), hi = (u32)((u64)n_cu * (part + 1) / n_parts);
        for (u32 c = lo; c < hi; c++) mask[c >> 5] |= 1u << (c & 31);
        HIPCHK(h, hipExtStreamCreateWithCUMask(&ns, (uint32_t)mask.size(), mask.data()));
    }
    const bool was_own = h->stream == h->own_stream;
    // (events last recorded on the stream that goes away go with it: hipStreamWaitEvent looks the recording stream of an event up --
    // for capture bookkeeping -- and that of a destroyed stream is a dangling pointer: "dependency created on uncaptured work in
    // another stream" from the inflate stream's wait for ev_packed, once in a dozen bench runs.  They are complete: the stream was waited for.)
    for (auto& e : h->ev_packed) if (e) { hipEventDestroy(e); e = nullptr; }
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    h->own_stream = ns;
    if (was_own) h->stream = ns;
    h->cu_split = (int)n_parts; h->cu_part = (int)part;
    for (auto* g : {&h->g_submit, &h->g_typing}) { if (g->exec) { hipGraphExecDestroy(g->exec); g->exec = nullptr; } g->sig.clear(); }
    return MLST_OK;
}
// the engine's own stream (for a caller that orders other work against it, e.g. torch.cuda.ExternalStream)
extern "C" int mlst_get_stream(mlst_handle* h, void** stream) {
    if (!h || !stream) return MLST_E_INVALID;
    *stream = (void*)h->own_stream;
    return MLST_OK;
}
// 1 while work queued on the engine's stream has not finished, 0 when it has (no waiting)
extern "C" int mlst_busy(mlst_handle* h) {
    if (!h) return MLST_E_INVALID;
    hipSetDevice(h->device);
    const hipError_t e = hipStreamQuery(h->stream);
    if (e == hipSuccess) return 0;
    if (e == hipErrorNotReady) return 1;
    return fail(h, MLST_E_HIP, "hipStreamQuery: %s", hipGetErrorString(e));
}

// Run the engine on a caller's HIP stream (e.g. the stream a torch.distributed collective is ordered against), or
// back on its own (stream = NULL).  Everything queued so far is waited for first.
extern "C" int mlst_set_stream(mlst_handle* h, void* stream) {
    if (!h) return MLST_E_INVALID;
    hipSetDevice(h->device);
    { int rc_ = bz_flush(h); if (rc_) return rc_; }
    drain_events(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->stream = stream ? (hipStream_t)stream : h->own_stream;
    return MLST_OK;
}
// mlst_export_stats_device / mlst_import_stats_device without the host synchronisation (capacity errors surface in
// mlst_typing_fetch / mlst_get_allele_stats)
extern "C" int mlst_export_stats_device_async(mlst_handle* h, int64_t* d_sum, int64_t* d_min) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    { int rc_ = bz_flush(h); if (rc_) return rc_; }
    hipSetDevice(h->device);
    hipLaunchKernelGGL(k_export, dim3(256), dim3(256), 0, h->stream, h->d_E, (long long*)d_sum, (long long*)d_min);
    HIPCHK(h, hipGetLastError());
    return MLST_OK;
}
extern "C" int mlst_import_stats_device_async(mlst_handle* h, const int64_t* d_sum, const int64_t* d_min) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    hipSetDevice(h->device);
    hipLaunchKernelGGL(k_import, dim3(256), dim3(256), 0, h->stream, h->d_E, (const long long*)d_sum, (const long long*)d_min);
    HIPCHK(h, hipGetLastError());
    return MLST_OK;
}

// Waits for mlst_typing_enqueue and hands everything over: the statistics of mlst_get_allele_stats, chosen[n_loci]
// (allele index or -1) and the consensus letters in the fixed layout of mlst_typing_layout.
static int typing_hand_over(mlst_handle* h, int slot, int64_t* sum_score, uint32_t* n_hits, uint64_t* locus_len, uint64_t* locus_first,
                            uint64_t* counters, int32_t* chosen, uint8_t* letters) {
    const u8* hs = h->h_tstats[slot]; const u8* ha = h->h_tauto[slot];
    Counters* c = (Counters*)(hs + h->off_ctr);
    if (c->err) return fail(h, MLST_E_CAPACITY, "capacity exceeded (flags 0x%llx); raise mlst_params.max_*", (unsigned long long)c->err);
    const u64 nl = h->n_loci;
    if (sum_score) memcpy(sum_score, hs + h->off_sum, (u64)h->n_alleles * 8);
    if (n_hits) memcpy(n_hits, hs + h->off_hits, (u64)h->n_alleles * 4);
    if (locus_len) memcpy(locus_len, hs + h->off_len, nl * 8);
    if (locus_first) memcpy(locus_first, hs + h->off_first, nl * 8);
    note_sieve_window(h, c);
    if (counters) { for (int i = 0; i < MLST_CNT_N; i++) counters[i] = c->cnt[i]; counters[MLST_CNT_RETAINED] = c->n_ret; counters[MLST_CNT_ITEMS] = c->n_items; }
    if (chosen) memcpy(chosen, ha, nl * 4);
    if (letters) memcpy(letters, ha + ((nl * 4 + 15) & ~15ull), h->fixed_cols);
    return MLST_OK;
}
extern "C" int mlst_typing_fetch(mlst_handle* h, int64_t* sum_score, uint32_t* n_hits, uint64_t* locus_len, uint64_t* locus_first,
                                 uint64_t* counters, int32_t* chosen, uint8_t* letters) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    if (!h->auto_pending) return fail(h, MLST_E_INVALID, "mlst_typing_fetch without mlst_typing_enqueue");
    hipSetDevice(h->device);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->auto_pending = false; h->t_ready = h->t_last;
    return typing_hand_over(h, h->t_ready, sum_score, n_hits, locus_len, locus_first, counters, chosen, letters);
}
// The two halves of mlst_typing_fetch.  mlst_typing_wait waits for the step queued by mlst_typing_enqueue (or the _finish
// entries); its results stay where they are, in one of two pinned slots, while the caller queues the engine's NEXT step
// (mlst_reset_sample, submit, mlst_typing_enqueue -- that one writes the other slot); mlst_typing_fetch_waited then copies
// them out without waiting for anything.  An engine on its own share of the CUs (mlst_set_cu_partition) is idle from the
// end of one step to the submission of the next: this keeps the copies (4 MB per step on cfg3) out of that gap.
extern "C" int mlst_typing_wait(mlst_handle* h) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    if (!h->auto_pending) return fail(h, MLST_E_INVALID, "mlst_typing_wait without mlst_typing_enqueue");
    hipSetDevice(h->device);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->auto_pending = false; h->t_ready = h->t_last;
    return MLST_OK;
}
extern "C" int mlst_typing_fetch_waited(mlst_handle* h, int64_t* sum_score, uint32_t* n_hits, uint64_t* locus_len, uint64_t* locus_first,
                                        uint64_t* counters, int32_t* chosen, uint8_t* letters) {
    if (!h || !h->have_state) return fail(h, MLST_E_INVALID, "no reference loaded");
    if (h->t_ready < 0) return fail(h, MLST_E_INVALID, "mlst_typing_fetch_waited without mlst_typing_wait");
    return typing_hand_over(h, h->t_ready, sum_score, n_hits, locus_len, locus_first, counters, chosen, letters);
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

extern "C" int mlst_set_profiling(mlst_handle* h, int on) { if (!h) return MLST_E_INVALID; drain_events(h); h->profiling = on == 1; h->window = on != 0; return MLST_OK; }
extern "C" int mlst_get_kernel_time(mlst_handle* h, int which, double* total_ms, uint64_t* launches) {
    if (!h || which < 0 || which >= 16) return MLST_E_INVALID;
    hipSetDevice(h->device); drain_events(h);
    if (total_ms) *total_ms = h->k_ms[which];
    if (launches) *launches = h->k_n[which];
    return MLST_OK;
}
extern "C" int mlst_reset_kernel_time(mlst_handle* h) { if (!h) return MLST_E_INVALID; drain_events(h); for (int i = 0; i < 16; i++) { h->k_ms[i] = 0; h->k_n[i] = 0; } return MLST_OK; }
extern "C" int mlst_get_index_bytes(mlst_handle* h, uint64_t out[4]) {
    if (!h || !out) return MLST_E_INVALID;
    out[0] = h->bytes_arena + h->bytes_hap; out[1] = h->bytes_sieve; out[2] = h->bytes_table; out[3] = (u64)(h->bitmap_fill * 1e6); return MLST_OK;
}
extern "C" int mlst_get_sieve_info(mlst_handle* h, uint64_t out[4]) {
    if (!h || !out || !h->have_ref) return fail(h, MLST_E_INVALID, "no reference loaded");
    out[0] = (u64)h->sieve_kind; out[1] = h->n_keys; out[2] = h->sieve_chain; out[3] = (u64)h->E.sieve_mask + 1; return MLST_OK;
}
extern "C" int mlst_get_extend_info(mlst_handle* h, uint64_t out[8]) {
    if (!h || !out || !h->have_ref) return fail(h, MLST_E_INVALID, "no reference loaded");
    out[0] = h->n_hap_rec; out[1] = h->bytes_hap; out[2] = h->hap_loci; out[3] = h->hap_win_max[0]; out[4] = h->hap_win_max[1];
    out[5] = h->ext_lds_recs[0] ? (u64)ext_lds_bytes(h, 0) : 0; out[6] = h->ext_lds_recs[1] ? (u64)ext_lds_bytes(h, 1) : 0; out[7] = (u64)h->ext_threads; return MLST_OK;
}
extern "C" void mlst_release_index_cache(void) { std::lock_guard<std::mutex> lk(g_index_mu); g_index_last.reset(); memset(g_index_key, 0, sizeof g_index_key); }
// ---- diagnostics of the routed sieve (profiles/route_modes.py; not a data path)
extern "C" int mlst_get_route_trace(mlst_handle* h, uint64_t* out, uint64_t cap_words, uint64_t* n_words) {
    if (!h) return MLST_E_INVALID;
    hipSetDevice(h->device);
    if (n_words) *n_words = 0;
    if (!h->rt_trace_on) { h->rt_trace_on = true; return MLST_OK; }      // first call: switch the trace on (allocated at the next submission)
    if (!h->d_rt_trace || !h->rt_prod) return MLST_OK;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const u64 n_wg = (u64)h->rt_prod + RT_OWNERS, need = 8 + n_wg * 4;
    if (n_words) *n_words = need;
    if (!out || cap_words < need) return MLST_OK;
    out[0] = h->rt_prod; out[1] = (u64)(uintptr_t)h->d_rt_arena; out[2] = (u64)(uintptr_t)h->rt_last_packed; out[3] = (u64)h->wall_khz;
    out[4] = h->rt_cap; out[5] = (u64)(uintptr_t)h->d_rfilter; out[6] = (u64)(uintptr_t)h->d_bin_flags; out[7] = h->cap_rt_arena;
    HIPCHK(h, hipMemcpy(out + 8, h->d_rt_trace, n_wg * 32, hipMemcpyDeviceToHost));
    return MLST_OK;
}
extern "C" int mlst_debug_route_realloc(mlst_handle* h, uint64_t pad_bytes) {
    if (!h) return MLST_E_INVALID;
    hipSetDevice(h->device);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->g_submit.exec) { hipGraphExecDestroy(h->g_submit.exec); h->g_submit.exec = nullptr; h->g_submit.sig.clear(); }
    if (pad_bytes == ~0ull) { if (h->d_rt_arena) h->dbg_pads.push_back(h->d_rt_arena); }      // keep the old arena allocated: the new one is other memory for certain
    else hipFree(h->d_rt_arena);
    h->d_rt_arena = nullptr; h->cap_rt_arena = 0;
    if (pad_bytes && pad_bytes != ~0ull) { void* q = nullptr; HIPCHK(h, hipMalloc(&q, pad_bytes)); h->dbg_pads.push_back(q); }
    return MLST_OK;
}
extern "C" int mlst_synchronize(mlst_handle* h) { if (!h) return MLST_E_INVALID; hipSetDevice(h->device); { int rc_ = bz_flush(h); if (rc_) return rc_; } HIPCHK(h, hipStreamSynchronize(h->stream)); return MLST_OK; }
