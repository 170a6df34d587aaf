/*
 * mlst_oracle.c -- CPU restatement of the MLST-typing hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product
 * (metamlst_amd/) never links, imports or calls it.
 *
 * What it restates:
 *   a1  read alignment against every allele (bowtie2 --very-sensitive-local -a --no-unal;
 *       [NOT IN TREE] -- user-run command documented via /root/reference/README.md:20).
 *       PARITY UNPINNED: bowtie2's source is not under /root/reference and the binary is
 *       not in the image, so this file restates bowtie2's documented local-mode scoring
 *       (constants in include/mlst_policy.h) with this engine's own deterministic
 *       seed-and-extend, not bowtie2's heuristics.  ST / allele concordance is the claim.
 *   a2  hit accumulation            /root/reference/metamlst.py:101-130
 *   a7  pileup base counts          /root/reference/metaMLST_functions.py:255-259 -> cmseq
 *       [NOT IN TREE, empty submodule].  PARITY UNPINNED for the same reason.
 *   a10 stringDiff                  /root/reference/metaMLST_functions.py:230-234
 *       (pinned: tests/golden holds vectors produced by importing the reference function).
 * Rows a3-a6, a8, a9, a11, a12 are host Python in the reference and are restated in Python
 * (metamlst_amd/typing.py, merge.py), pinned by tests/golden.
 *
 * Style: clarity over speed.  One read at a time, per-base loops, full banded matrices.
 * OpenMP over reads is used only so the CPU baseline can use every host core.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/mlst_policy.h"
#include "../include/mlst.h"   /* mlst_params, mlst_item: struct layouts only */

#define K  MLST_SEED_LEN
#define ST MLST_SEED_STEP

/* ------------------------------------------------------------------ reference */

typedef struct { uint64_t key; uint32_t post; } kp_t;

typedef struct orc_ref {
    uint32_t n_alleles, n_loci;
    uint8_t** code;          /* per allele: 0..3 = ACGT, 4 = other */
    uint8_t** ascii;         /* per allele: bytes as given */
    uint32_t* len;
    uint32_t* locus_of;      /* per allele */
    uint32_t* locus_begin;   /* per locus: first allele */
    uint32_t* locus_count;
    /* seed index: sorted unique keys with posting ranges */
    uint64_t n_keys;
    uint64_t* keys;
    uint64_t* pstart;        /* n_keys+1 */
    uint32_t* posts;
    mlst_params prm;
    int32_t floor_tab[MLST_MAX_READ_LEN + 1];   /* bowtie2 --score-min per read length */
    uint8_t pen_tab[256];                       /* mismatch penalty per Phred */
} orc_ref;

static uint8_t base_code(uint8_t c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1;
                 case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

static int kp_cmp(const void* a, const void* b) {
    const kp_t* x = (const kp_t*)a; const kp_t* y = (const kp_t*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    if (x->post != y->post) return x->post < y->post ? -1 : 1;
    return 0;
}

/* seed key: base t of the 20-mer at bits 2t (same as the packed read word layout) */
static int kmer_key(const uint8_t* code, uint64_t* key) {
    uint64_t k = 0;
    for (int t = 0; t < K; t++) { if (code[t] > 3) return 0; k |= (uint64_t)code[t] << (2 * t); }
    *key = k; return 1;
}

void orc_ref_free(orc_ref* r) {
    if (!r) return;
    for (uint32_t a = 0; a < r->n_alleles; a++) { free(r->code ? r->code[a] : NULL); free(r->ascii ? r->ascii[a] : NULL); }
    free(r->code); free(r->ascii); free(r->len); free(r->locus_of); free(r->locus_begin);
    free(r->locus_count); free(r->keys); free(r->pstart); free(r->posts); free(r);
}

/* Index every 20-mer of every allele on both strands.  Counterpart of bowtie2-build over
 * the FASTA dump (metamlst-index.py:222-247, metaMLST_functions.py:149-161). */
orc_ref* orc_ref_build(const uint8_t* ascii_concat, const uint64_t* off, const uint32_t* locus_id,
                       uint32_t n_alleles, const mlst_params* prm) {
    orc_ref* r = (orc_ref*)calloc(1, sizeof(orc_ref));
    r->prm = *prm;
    r->n_alleles = n_alleles;
    r->code = (uint8_t**)calloc(n_alleles, sizeof(uint8_t*));
    r->ascii = (uint8_t**)calloc(n_alleles, sizeof(uint8_t*));
    r->len = (uint32_t*)calloc(n_alleles, sizeof(uint32_t));
    r->locus_of = (uint32_t*)calloc(n_alleles, sizeof(uint32_t));
    uint32_t n_loci = 0; uint64_t n_pairs_max = 0;
    for (uint32_t a = 0; a < n_alleles; a++) {
        uint32_t L = (uint32_t)(off[a + 1] - off[a]);
        if (L > MLST_MAX_ALLELE_LEN) { orc_ref_free(r); return NULL; }
        r->len[a] = L; r->locus_of[a] = locus_id[a];
        if (locus_id[a] + 1 > n_loci) n_loci = locus_id[a] + 1;
        r->code[a] = (uint8_t*)malloc(L + 1); r->ascii[a] = (uint8_t*)malloc(L + 1);
        for (uint32_t i = 0; i < L; i++) { r->ascii[a][i] = ascii_concat[off[a] + i]; r->code[a][i] = base_code(r->ascii[a][i]); }
        if (L >= K) n_pairs_max += 2ull * (L - K + 1);
    }
    r->n_loci = n_loci;
    r->locus_begin = (uint32_t*)calloc(n_loci + 1, sizeof(uint32_t));
    r->locus_count = (uint32_t*)calloc(n_loci + 1, sizeof(uint32_t));
    for (uint32_t a = 0; a < n_alleles; a++) {
        uint32_t l = locus_id[a];
        if (r->locus_count[l] == 0) r->locus_begin[l] = a;
        else if (r->locus_begin[l] + r->locus_count[l] != a) { orc_ref_free(r); return NULL; } /* not contiguous */
        r->locus_count[l]++;
    }
    /* A posting is (locus, strand, position): the alleles of a locus repeat most of them (they differ by a few SNPs),
     * and the index keeps each distinct (key, posting) pair once.  So the pairs are made unique locus by locus
     * (a posting names its locus: pairs of different loci never coincide) -- loci in parallel, each a small sort --
     * before the one sort of the whole index.  A many-species database has 25 x fewer distinct pairs than pairs. */
    (void)n_pairs_max;
    kp_t** lkp = (kp_t**)calloc(n_loci ? n_loci : 1, sizeof(kp_t*));
    uint64_t* lnp = (uint64_t*)calloc(n_loci ? n_loci : 1, sizeof(uint64_t));
    #pragma omp parallel for schedule(dynamic, 1)
    for (int64_t l = 0; l < (int64_t)n_loci; l++) {
        uint64_t cap = 0;
        for (uint32_t a = r->locus_begin[l]; a < r->locus_begin[l] + r->locus_count[l]; a++) if (r->len[a] >= K) cap += 2ull * (r->len[a] - K + 1);
        kp_t* v = (kp_t*)malloc((cap + 1) * sizeof(kp_t)); uint64_t n = 0;
        uint8_t rc[K];
        for (uint32_t a = r->locus_begin[l]; a < r->locus_begin[l] + r->locus_count[l]; a++) {
            uint32_t L = r->len[a]; if (L < K) continue;
            for (uint32_t p = 0; p + K <= L; p++) {
                uint64_t key;
                if (!kmer_key(r->code[a] + p, &key)) continue;
                v[n].key = key; v[n].post = ((uint32_t)l << 13) | (0u << 12) | p; n++;
                for (int t = 0; t < K; t++) rc[t] = 3 - r->code[a][p + K - 1 - t];
                kmer_key(rc, &key);
                v[n].key = key; v[n].post = ((uint32_t)l << 13) | (1u << 12) | p; n++;
            }
            if (n > (1u << 16) || a + 1 == r->locus_begin[l] + r->locus_count[l]) {   /* keep the working set small: unique as we go */
                qsort(v, n, sizeof(kp_t), kp_cmp);
                uint64_t u = 0;
                for (uint64_t i = 0; i < n; i++) if (i == 0 || kp_cmp(&v[i], &v[i - 1]) != 0) v[u++] = v[i];
                n = u;
            }
        }
        lkp[l] = v; lnp[l] = n;
    }
    uint64_t np = 0;
    for (uint32_t l = 0; l < n_loci; l++) np += lnp[l];
    kp_t* kp = (kp_t*)malloc((np + 1) * sizeof(kp_t)); np = 0;
    for (uint32_t l = 0; l < n_loci; l++) { memcpy(kp + np, lkp[l], lnp[l] * sizeof(kp_t)); np += lnp[l]; free(lkp[l]); }
    free(lkp); free(lnp);
    qsort(kp, np, sizeof(kp_t), kp_cmp);
    /* unique pairs, group by key, drop repetitive keys */
    uint64_t nu = 0;
    for (uint64_t i = 0; i < np; i++) if (i == 0 || kp_cmp(&kp[i], &kp[i - 1]) != 0) kp[nu++] = kp[i];
    r->keys = (uint64_t*)malloc((nu + 1) * sizeof(uint64_t));
    r->pstart = (uint64_t*)malloc((nu + 2) * sizeof(uint64_t));
    r->posts = (uint32_t*)malloc((nu + 1) * sizeof(uint32_t));
    uint64_t nk = 0, npost = 0;
    for (uint64_t i = 0; i < nu;) {
        uint64_t j = i; while (j < nu && kp[j].key == kp[i].key) j++;
        if (j - i <= MLST_MAX_POSTINGS) {
            r->keys[nk] = kp[i].key; r->pstart[nk] = npost; nk++;
            for (uint64_t t = i; t < j; t++) r->posts[npost++] = kp[t].post;
        }
        i = j;
    }
    r->pstart[nk] = npost; r->n_keys = nk;
    free(kp);
    /* bowtie2 --score-min G,20,8: minimum AS for a record to exist [NOT IN TREE] */
    for (int n = 0; n <= MLST_MAX_READ_LEN; n++) {
        double f = prm->minscore_const + prm->minscore_coef * log((double)(n > 0 ? n : 1));
        long v = (long)f; if (v < 0) v = 0;
        r->floor_tab[n] = (int32_t)v;
    }
    /* bowtie2 --mp MX,MN quality-aware mismatch penalty [NOT IN TREE] */
    for (int q = 0; q < 256; q++) {
        int qq = q > 40 ? 40 : q;
        r->pen_tab[q] = (uint8_t)(prm->mm_min + ((prm->mm_max - prm->mm_min) * qq) / 40);
    }
    return r;
}

uint32_t orc_ref_n_loci(const orc_ref* r) { return r->n_loci; }
uint64_t orc_ref_n_keys(const orc_ref* r) { return r->n_keys; }

static int64_t find_key(const orc_ref* r, uint64_t key) {
    int64_t lo = 0, hi = (int64_t)r->n_keys - 1;
    while (lo <= hi) { int64_t mid = (lo + hi) >> 1;
        if (r->keys[mid] == key) return mid;
        if (r->keys[mid] < key) lo = mid + 1; else hi = mid - 1; }
    return -1;
}

/* ------------------------------------------------------------------ alignment */

typedef struct {
    int32_t P;            /* packed best value */
    int score, xm, xo;
    int mm_total;         /* mismatching columns of the full overlap on the seed diagonal */
    int used_dp;
    int n_cols;           /* aligned (read i, allele j) columns, for the pileup */
    int16_t ci[MLST_MAX_READ_LEN], cj[MLST_MAX_READ_LEN];
} aln_t;

static inline int32_t col_delta(const orc_ref* r, uint8_t rb, uint8_t pen, uint8_t ab) {
    if (rb < 4 && ab < 4 && rb == ab) return (int32_t)r->prm.match_bonus << MLST_P_SHIFT;
    int p = (rb > 3 || ab > 3) ? r->prm.n_penalty : pen;
    return -((int32_t)p << MLST_P_SHIFT) - 1;          /* xm field counts down */
}

/* Ungapped local alignment on diagonal d (allele j = read i + d): Kadane on packed values.
 * The end is the first position reaching the maximum; the start is where the running value
 * last restarted from the empty alignment. */
static void align_ungapped(const orc_ref* r, const uint8_t* rb, const uint8_t* pen, int n,
                           const uint8_t* ab, int m, int d, aln_t* o) {
    int i0 = d < 0 ? -d : 0, i1 = (m - d) < n ? (m - d) : n;
    int32_t cur = MLST_P0, best = MLST_P0; int cs = i0, bs = i0, be = i0, mm = 0;
    for (int i = i0; i < i1; i++) {
        int32_t dl = col_delta(r, rb[i], pen[i], ab[i + d]);
        if (dl < 0) mm++;
        cur += dl;
        if (cur <= MLST_P0) { cur = MLST_P0; cs = i + 1; }     /* equality cannot occur, see DESIGN.md */
        if (cur > best) { best = cur; bs = cs; be = i + 1; }
    }
    o->P = best; o->score = best >> MLST_P_SHIFT; o->xm = 255 - (best & 0xFF); o->xo = 127 - ((best >> 8) & 0x7F);
    o->mm_total = mm; o->used_dp = 0; o->n_cols = 0;
    for (int i = bs; i < be; i++) { o->ci[o->n_cols] = (int16_t)i; o->cj[o->n_cols] = (int16_t)(i + d); o->n_cols++; }
}

/* Banded affine-gap local alignment (Gotoh) around diagonal d, half width W.
 * E = gap in the read (consumes an allele base), F = gap in the allele (consumes a read base).
 * Tie rules: H prefers fresh > diagonal > E > F; E and F prefer open over extend;
 * the end cell is the first maximum in row-major order. */
static void align_banded(const orc_ref* r, const uint8_t* rb, const uint8_t* pen, int n,
                         const uint8_t* ab, int m, int d, aln_t* o) {
    const int W = r->prm.band_w, BW = 2 * W + 1, G = r->prm.gbar;
    const int32_t OPEN = ((int32_t)(r->prm.gap_open + r->prm.gap_ext) << MLST_P_SHIFT) + (1 << 8);
    const int32_t EXT = (int32_t)r->prm.gap_ext << MLST_P_SHIFT;
    int32_t* H = (int32_t*)malloc(sizeof(int32_t) * (size_t)n * BW);
    int32_t* E = (int32_t*)malloc(sizeof(int32_t) * (size_t)n * BW);
    int32_t* F = (int32_t*)malloc(sizeof(int32_t) * (size_t)n * BW);
    uint8_t* T = (uint8_t*)malloc((size_t)n * BW);   /* bits 0-1 H source, bit 2 E extend, bit 3 F extend */
    int32_t best = MLST_P0; int bi = -1, bb = -1;
    for (int i = 0; i < n; i++) {
        int gap_ok = (i >= G && i < n - G);
        for (int b = 0; b < BW; b++) {
            int j = i + d - W + b; size_t c = (size_t)i * BW + b;
            if (j < 0 || j >= m) { H[c] = E[c] = F[c] = MLST_P_NEG; T[c] = 0; continue; }
            int32_t hd = (i > 0 && j > 0) ? H[c - BW] : MLST_P0;
            int32_t diag = hd + col_delta(r, rb[i], pen[i], ab[j]);
            int32_t e = MLST_P_NEG, f = MLST_P_NEG; uint8_t t = 0;
            if (gap_ok && b > 0 && j - 1 >= 0) {
                int32_t e1 = H[c - 1] - OPEN, e2 = E[c - 1] - EXT;
                if (e2 > e1) { e = e2; t |= 4; } else e = e1;
            }
            if (gap_ok && i > 0 && b < BW - 1) {
                int32_t f1 = H[c - BW + 1] - OPEN, f2 = F[c - BW + 1] - EXT;
                if (f2 > f1) { f = f2; t |= 8; } else f = f1;
            }
            int32_t h = MLST_P0; uint8_t src = 0;
            if (diag > h) { h = diag; src = 1; }
            if (e > h) { h = e; src = 2; }
            if (f > h) { h = f; src = 3; }
            H[c] = h; E[c] = e; F[c] = f; T[c] = t | src;
            if (h > best) { best = h; bi = i; bb = b; }
        }
    }
    o->P = best; o->score = best >> MLST_P_SHIFT; o->xm = 255 - (best & 0xFF); o->xo = 127 - ((best >> 8) & 0x7F);
    o->used_dp = 1; o->n_cols = 0;
    /* traceback */
    int i = bi, b = bb, state = 0;  /* 0 = H, 1 = E, 2 = F */
    int16_t ti[MLST_MAX_READ_LEN], tj[MLST_MAX_READ_LEN]; int nt = 0;
    while (i >= 0 && b >= 0 && b < BW) {
        size_t c = (size_t)i * BW + b; uint8_t t = T[c];
        if (state == 0) {
            int src = t & 3;
            if (src == 0) break;
            if (src == 1) { ti[nt] = (int16_t)i; tj[nt] = (int16_t)(i + d - W + b); nt++; i--; /* b unchanged */ if (i < 0) break; }
            else if (src == 2) state = 1; else state = 2;
        } else if (state == 1) {            /* E at (i,b): came from (i,b-1) */
            int ext = t & 4; b--; state = ext ? 1 : 0;
        } else {                            /* F at (i,b): came from (i-1,b+1) */
            int ext = t & 8; i--; b++; state = ext ? 2 : 0;
        }
    }
    for (int k = nt - 1; k >= 0; k--) { o->ci[o->n_cols] = ti[k]; o->cj[o->n_cols] = tj[k]; o->n_cols++; }
    free(H); free(E); free(F); free(T);
}

/* The engine's ALIGN(read, allele, diagonal): ungapped first, banded SW when the seed
 * diagonal looks broken by an indel (policy MLST_DEF_GAP_TRIGGER_MM). */
static void align_pair(const orc_ref* r, const uint8_t* rb, const uint8_t* pen, int n,
                       uint32_t allele, int d, aln_t* o) {
    const uint8_t* ab = r->code[allele]; int m = (int)r->len[allele];
    align_ungapped(r, rb, pen, n, ab, m, d, o);
    int trig = r->prm.gap_trigger_mm;
    int i0 = d < 0 ? -d : 0, i1 = (m - d) < n ? (m - d) : n;
    int overlap = i1 > i0 ? i1 - i0 : 0;
    int clipped = overlap - o->n_cols;            /* overlap columns the ungapped alignment left out */
    int run_dp = trig < 0 ? 1 : (o->mm_total > trig && o->score >= r->floor_tab[n] &&
                                 clipped >= r->prm.gap_trigger_clip && 2 * (o->mm_total - o->xm) >= clipped);
    if (run_dp) { int mm = o->mm_total; align_banded(r, rb, pen, n, ab, m, d, o); o->mm_total = mm; }
}

/* test hook: one (read, allele, strand, diag) alignment.  mode 0 = policy, 1 = ungapped, 2 = banded */
int orc_align_one(const orc_ref* r, const uint8_t* bases, const uint8_t* quals, int n, uint32_t allele,
                  int strand, int d, int mode, int32_t* out /* score,xm,xo,mm_total,used_dp,n_cols */,
                  int16_t* cols_i, int16_t* cols_j) {
    if (n > MLST_MAX_READ_LEN || allele >= r->n_alleles) return -1;
    uint8_t rb[MLST_MAX_READ_LEN], pen[MLST_MAX_READ_LEN];
    for (int i = 0; i < n; i++) {
        int s = strand ? n - 1 - i : i; uint8_t c = base_code(bases[s]);
        rb[i] = strand ? (c < 4 ? 3 - c : 4) : c;
        int q = (int)quals[s] - 33; if (q < 0) q = 0;
        pen[i] = r->pen_tab[q];
    }
    aln_t* a = (aln_t*)malloc(sizeof(aln_t));
    if (mode == 0) align_pair(r, rb, pen, n, allele, d, a);
    else if (mode == 1) align_ungapped(r, rb, pen, n, r->code[allele], (int)r->len[allele], d, a);
    else { align_ungapped(r, rb, pen, n, r->code[allele], (int)r->len[allele], d, a); int mm = a->mm_total;
           align_banded(r, rb, pen, n, r->code[allele], (int)r->len[allele], d, a); a->mm_total = mm; }
    out[0] = a->score; out[1] = a->xm; out[2] = a->xo; out[3] = a->mm_total; out[4] = a->used_dp; out[5] = a->n_cols;
    if (cols_i && cols_j) for (int k = 0; k < a->n_cols; k++) { cols_i[k] = a->ci[k]; cols_j[k] = a->cj[k]; }
    free(a); return 0;
}

/* ------------------------------------------------------------------ seeding */

typedef struct { uint32_t locus; int32_t diag; uint16_t strand, votes; } cand_t;

/* Exact 20-mer seeds every 16 read bases against both strands of every allele; votes per
 * (locus, strand, diagonal); one work item per (locus, strand): most votes, then smaller diagonal. */
static int seed_read(const orc_ref* r, const uint8_t* code, int n, cand_t* items /* MLST_MAX_CAND */) {
    cand_t bins[MLST_MAX_CAND]; int nb = 0;
    for (int o = 0; o + K <= n; o += ST) {
        uint64_t key; if (!kmer_key(code + o, &key)) continue;
        int64_t ki = find_key(r, key); if (ki < 0) continue;
        for (uint64_t p = r->pstart[ki]; p < r->pstart[ki + 1]; p++) {
            uint32_t post = r->posts[p];
            uint32_t locus = post >> 13, strand = (post >> 12) & 1; int pos = (int)(post & 0xFFF);
            int diag = strand ? pos + K + o - n : pos - o;
            int t; for (t = 0; t < nb; t++) if (bins[t].locus == locus && bins[t].strand == strand && bins[t].diag == diag) break;
            if (t < nb) bins[t].votes++;
            else if (nb < MLST_MAX_CAND) { bins[nb].locus = locus; bins[nb].strand = (uint16_t)strand; bins[nb].diag = diag; bins[nb].votes = 1; nb++; }
        }
    }
    int ni = 0;
    for (int t = 0; t < nb; t++) {
        int u; for (u = 0; u < ni; u++) if (items[u].locus == bins[t].locus && items[u].strand == bins[t].strand) break;
        if (u == ni) items[ni++] = bins[t];
        else if (bins[t].votes > items[u].votes || (bins[t].votes == items[u].votes && bins[t].diag < items[u].diag)) items[u] = bins[t];
    }
    int no = 0;
    for (int u = 0; u < ni; u++) if (items[u].votes >= MLST_MIN_VOTES) items[no++] = items[u];
    return no;
}

static void orient_read(const orc_ref* r, const uint8_t* code, const uint8_t* phred, int n, int strand,
                        uint8_t* rb, uint8_t* pen, uint8_t* q) {
    for (int i = 0; i < n; i++) {
        int s = strand ? n - 1 - i : i;
        rb[i] = strand ? (code[s] < 4 ? 3 - code[s] : 4) : code[s];
        q[i] = phred[s]; pen[i] = r->pen_tab[phred[s]];
    }
}

/* ------------------------------------------------------------------ pass 1 */

typedef struct { uint32_t allele; int score, xm, xo; uint32_t locus; int item; } rec_t;

typedef struct {
    int64_t* sum; uint32_t* hits; uint64_t* len; uint64_t* first; uint64_t cnt[MLST_CNT_N];
} acc_t;

/* metamlst.py:101-130 for the records of ONE read (its SAM lines are consecutive):
 *   score = AS (column 12); xM = column 15 read BY POSITION (metamlst.py:109-110): with bowtie2's
 *   tag order AS,[XS],XN,XM,XO,... that is XM when XS:i is present -- i.e. when the read has a
 *   second record -- and XO when it is not (Q1);
 *   accept iff score >= minscore and len(SEQ) >= min_read_len and xM <= max_xM  (:115);
 *   cel[sp][gene][allele].append(score) (:125); sequenceBank[sp_gene][QNAME] = len(SEQ) (:127);
 *   ignoredReads / totalReads count records (:129-130, Q13).
 * `item` groups the records of one (read, locus, strand) extension. */
/* sequenceBank[species_gene][readCode] = len(sequence) (metamlst.py:127) for the reads that share ONE readCode (QNAME): a
 * dictionary entry per locus, overwritten by every accepted record, so that the locus is credited with the length of the
 * LAST accepted record of the QNAME (Q3) -- once for a read that matches both strands of the locus, and with the second
 * mate's length when both mates of a pair (same QNAME) have accepted records there. */
typedef struct { uint32_t locus[2 * MLST_MAX_CAND]; int len[2 * MLST_MAX_CAND]; int n; } bank_t;
static void bank_set(bank_t* b, uint32_t locus, int len) {
    for (int i = 0; i < b->n; i++) if (b->locus[i] == locus) { b->len[i] = len; return; }
    b->locus[b->n] = locus; b->len[b->n] = len; b->n++;
}
static void bank_flush(bank_t* b, acc_t* A) { for (int i = 0; i < b->n; i++) A->len[b->locus[i]] += (uint64_t)b->len[i]; b->n = 0; }

static void accumulate_read(const orc_ref* r, int n, uint64_t gi, const rec_t* recs, size_t nrec, acc_t* A, bank_t* bank) {
    int use_xo = r->prm.xm_field_quirk && nrec == 1;
    for (size_t k = 0; k < nrec; k++) {
        int f15 = use_xo ? recs[k].xo : recs[k].xm;
        A->cnt[MLST_CNT_TOTAL_RECORDS]++;
        if (recs[k].score >= r->prm.minscore && n >= r->prm.min_read_len && f15 <= r->prm.max_xm) {
            A->sum[recs[k].allele] += recs[k].score; A->hits[recs[k].allele]++;
            bank_set(bank, recs[k].locus, n);
            if (gi < A->first[recs[k].locus]) A->first[recs[k].locus] = gi;
        } else A->cnt[MLST_CNT_IGNORED]++;
    }
}

/* Pass 1: for every read, every record bowtie2 -a would emit (AS >= --score-min), filtered and
 * accumulated as metamlst.py:101-130 does: accept iff AS >= minscore and len(SEQ) >= min_read_len
 * and field15 <= max_xM; cel[sp][gene][allele].append(AS); sequenceBank[locus][QNAME] = len(SEQ);
 * totalReads / ignoredReads count records.
 * Outputs are zeroed here.  items_out may be NULL. */
int orc_pass1(const orc_ref* r, const uint8_t* bases, const uint8_t* quals, const uint64_t* off,
              uint64_t n_reads, uint64_t read_index_base,
              int64_t* sum_score, uint32_t* n_hits, uint64_t* locus_len_sum, uint64_t* locus_first,
              uint64_t* counters, mlst_item* items_out, uint64_t items_cap, uint64_t* n_items_out,
              int n_threads, int paired /* reads 2k, 2k+1 are mates sharing a QNAME (Q3) */) {
    const uint32_t nA = r->n_alleles, nL = r->n_loci;
    memset(sum_score, 0, sizeof(int64_t) * nA); memset(n_hits, 0, sizeof(uint32_t) * nA);
    memset(locus_len_sum, 0, sizeof(uint64_t) * nL);
    for (uint32_t l = 0; l < nL; l++) locus_first[l] = UINT64_MAX;
    memset(counters, 0, sizeof(uint64_t) * MLST_CNT_N);
    uint64_t n_items = 0; int bad = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    #pragma omp parallel
    {
        int64_t* t_sum = (int64_t*)calloc(nA, sizeof(int64_t));
        uint32_t* t_hits = (uint32_t*)calloc(nA, sizeof(uint32_t));
        uint64_t* t_len = (uint64_t*)calloc(nL, sizeof(uint64_t));
        uint64_t* t_first = (uint64_t*)malloc(nL * sizeof(uint64_t));
        for (uint32_t l = 0; l < nL; l++) t_first[l] = UINT64_MAX;
        acc_t A; A.sum = t_sum; A.hits = t_hits; A.len = t_len; A.first = t_first; memset(A.cnt, 0, sizeof(A.cnt));
        uint64_t* t_cnt = A.cnt;
        rec_t* recs = NULL; size_t recs_cap = 0;
        aln_t* al = (aln_t*)malloc(sizeof(aln_t));
        const int64_t unit = paired ? 2 : 1;          /* reads that share a QNAME are accumulated together */
        bank_t bank; bank.n = 0;
        #pragma omp for schedule(dynamic, 2048)
        for (int64_t ui = 0; ui < ((int64_t)n_reads + unit - 1) / unit; ui++) {
          for (int64_t ri = ui * unit; ri < (ui + 1) * unit && ri < (int64_t)n_reads; ri++) {
            int n = (int)(off[ri + 1] - off[ri]);
            t_cnt[MLST_CNT_READS_SEEN]++;
            if (n > MLST_MAX_READ_LEN) { bad = 1; continue; }
            if (n < K) continue;
            uint8_t code[MLST_MAX_READ_LEN], phred[MLST_MAX_READ_LEN];
            for (int i = 0; i < n; i++) {
                code[i] = base_code(bases[off[ri] + i]);
                int q = (int)quals[off[ri] + i] - 33; phred[i] = (uint8_t)(q < 0 ? 0 : (q > 127 ? 127 : q));
            }
            cand_t items[MLST_MAX_CAND];
            int ni = seed_read(r, code, n, items);
            if (ni == 0) continue;
            t_cnt[MLST_CNT_RETAINED]++;
            size_t nrec = 0;
            for (int it = 0; it < ni; it++) {
                uint8_t rb[MLST_MAX_READ_LEN], pen[MLST_MAX_READ_LEN], q[MLST_MAX_READ_LEN];
                orient_read(r, code, phred, n, items[it].strand, rb, pen, q);
                uint32_t L = items[it].locus;
                t_cnt[MLST_CNT_ITEMS]++;
                if (items_out) {
                    uint64_t slot;
                    #pragma omp atomic capture
                    slot = n_items++;
                    if (slot < items_cap) { mlst_item* m = &items_out[slot];
                        m->read_index = read_index_base + (uint64_t)ri; m->locus = L; m->diag = items[it].diag;
                        m->strand = items[it].strand; m->votes = items[it].votes; m->reserved = 0; }
                }
                for (uint32_t a = r->locus_begin[L]; a < r->locus_begin[L] + r->locus_count[L]; a++) {
                    align_pair(r, rb, pen, n, a, items[it].diag, al);
                    if (al->used_dp) t_cnt[MLST_CNT_DP_PAIRS]++;
                    if (al->score < r->floor_tab[n] || al->score <= 0) continue;   /* bowtie2 emits no record */
                    if (nrec == recs_cap) { recs_cap = recs_cap ? recs_cap * 2 : 4096; recs = (rec_t*)realloc(recs, recs_cap * sizeof(rec_t)); }
                    recs[nrec].allele = a; recs[nrec].score = al->score; recs[nrec].xm = al->xm; recs[nrec].xo = al->xo; recs[nrec].locus = L; recs[nrec].item = it; nrec++;
                }
            }
            accumulate_read(r, n, read_index_base + (uint64_t)ri, recs, nrec, &A, &bank);
          }
          bank_flush(&bank, &A);
        }
        #pragma omp critical
        {
            for (uint32_t a = 0; a < nA; a++) { sum_score[a] += t_sum[a]; n_hits[a] += t_hits[a]; }
            for (uint32_t l = 0; l < nL; l++) { locus_len_sum[l] += t_len[l]; if (t_first[l] < locus_first[l]) locus_first[l] = t_first[l]; }
            for (int c = 0; c < MLST_CNT_N; c++) counters[c] += t_cnt[c];
        }
        free(t_sum); free(t_hits); free(t_len); free(t_first); free(recs); free(al);
    }
    counters[MLST_CNT_CANDIDATES] = counters[MLST_CNT_RETAINED];   /* the oracle has no sieve */
    if (n_items_out) *n_items_out = n_items;
    return bad ? -5 : 0;
}

/* Records-level entry: accumulate explicit alignment records (the SAM lines the reference
 * parses) exactly as pass 1 does.  Records of one read must be consecutive.  Used by the golden
 * tests to pin accumulate_read against /root/reference/metamlst.py run on the same records. */
int orc_accumulate_records(const orc_ref* r, uint64_t n_recs, const uint64_t* read_index, const uint32_t* allele,
                           const int32_t* as, const int32_t* xm, const int32_t* xo, const int32_t* seqlen,
                           int64_t* sum_score, uint32_t* n_hits, uint64_t* locus_len_sum, uint64_t* locus_first,
                           uint64_t* counters) {
    const uint32_t nA = r->n_alleles, nL = r->n_loci;
    memset(sum_score, 0, sizeof(int64_t) * nA); memset(n_hits, 0, sizeof(uint32_t) * nA);
    memset(locus_len_sum, 0, sizeof(uint64_t) * nL);
    for (uint32_t l = 0; l < nL; l++) locus_first[l] = UINT64_MAX;
    acc_t A; A.sum = sum_score; A.hits = n_hits; A.len = locus_len_sum; A.first = locus_first; memset(A.cnt, 0, sizeof(A.cnt));
    rec_t* recs = (rec_t*)malloc(sizeof(rec_t) * (n_recs + 1));
    for (uint64_t i = 0; i < n_recs;) {
        uint64_t j = i; size_t nrec = 0; uint32_t loci_seen[MLST_MAX_CAND]; int nls = 0;
        while (j < n_recs && read_index[j] == read_index[i]) {
            if (allele[j] >= nA) { free(recs); return -1; }
            uint32_t L = r->locus_of[allele[j]]; int it;
            for (it = 0; it < nls; it++) if (loci_seen[it] == L) break;
            if (it == nls) { if (nls == MLST_MAX_CAND) { free(recs); return -4; } loci_seen[nls++] = L; }
            recs[nrec].allele = allele[j]; recs[nrec].score = as[j]; recs[nrec].xm = xm[j]; recs[nrec].xo = xo[j];
            recs[nrec].locus = L; recs[nrec].item = it; nrec++; j++;
        }
        { bank_t bank; bank.n = 0; accumulate_read(r, seqlen[i], read_index[i], recs, nrec, &A, &bank); bank_flush(&bank, &A); }
        i = j;
    }
    memcpy(counters, A.cnt, sizeof(A.cnt));
    free(recs);
    return 0;
}

/* ------------------------------------------------------------------ pass 2 */

/* Pass 2: base counts per column of each chosen allele.  Restates cmseq get_base_stats over a
 * pysam pileup with stepper 'nofilter' (metaMLST_functions.py:255-259; cmseq [NOT IN TREE]):
 * every record on the contig (secondaries included) whose tags satisfy AS >= minscore and
 * XM <= max_xM contributes its aligned bases with Phred >= minqual and base in ACGT;
 * deletions and inserted / soft-clipped read bases contribute nothing.
 * chosen[k] = allele index; counts = uint32[sum len][4] (A,C,G,T) in the order given. */
int orc_pileup(const orc_ref* r, const uint8_t* bases, const uint8_t* quals, const uint64_t* off,
               uint64_t n_reads, const uint32_t* chosen, uint32_t n_chosen, uint32_t* counts, int n_threads) {
    const uint32_t nL = r->n_loci;
    int64_t* col_base = (int64_t*)malloc(sizeof(int64_t) * nL);
    uint32_t* chosen_of = (uint32_t*)malloc(sizeof(uint32_t) * nL);
    for (uint32_t l = 0; l < nL; l++) { col_base[l] = -1; chosen_of[l] = 0; }
    uint64_t ncols = 0;
    for (uint32_t k = 0; k < n_chosen; k++) {
        uint32_t a = chosen[k]; if (a >= r->n_alleles) { free(col_base); free(chosen_of); return -1; }
        uint32_t L = r->locus_of[a]; if (col_base[L] >= 0) { free(col_base); free(chosen_of); return -1; }
        col_base[L] = (int64_t)ncols; chosen_of[L] = a; ncols += r->len[a];
    }
    memset(counts, 0, sizeof(uint32_t) * 4 * ncols);
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    #pragma omp parallel
    {
        aln_t* al = (aln_t*)malloc(sizeof(aln_t));
        #pragma omp for schedule(dynamic, 4096)
        for (int64_t ri = 0; ri < (int64_t)n_reads; ri++) {
            int n = (int)(off[ri + 1] - off[ri]);
            if (n > MLST_MAX_READ_LEN || n < K) continue;
            uint8_t code[MLST_MAX_READ_LEN], phred[MLST_MAX_READ_LEN];
            for (int i = 0; i < n; i++) {
                code[i] = base_code(bases[off[ri] + i]);
                int q = (int)quals[off[ri] + i] - 33; phred[i] = (uint8_t)(q < 0 ? 0 : (q > 127 ? 127 : q));
            }
            cand_t items[MLST_MAX_CAND];
            int ni = seed_read(r, code, n, items);
            for (int it = 0; it < ni; it++) {
                uint32_t L = items[it].locus; if (col_base[L] < 0) continue;
                uint8_t rb[MLST_MAX_READ_LEN], pen[MLST_MAX_READ_LEN], q[MLST_MAX_READ_LEN];
                orient_read(r, code, phred, n, items[it].strand, rb, pen, q);
                align_pair(r, rb, pen, n, chosen_of[L], items[it].diag, al);
                if (al->score < r->floor_tab[n] || al->score <= 0) continue;
                if (al->score < r->prm.minscore || al->xm > r->prm.max_xm) continue;   /* BAM_tagFilter AS, XM */
                for (int k = 0; k < al->n_cols; k++) {
                    int i = al->ci[k], j = al->cj[k];
                    if (rb[i] > 3 || q[i] < r->prm.minqual) continue;
                    uint32_t* c = &counts[((uint64_t)col_base[L] + (uint64_t)j) * 4 + rb[i]];
                    #pragma omp atomic
                    (*c)++;
                }
            }
        }
        free(al);
    }
    free(col_base); free(chosen_of);
    return 0;
}

/* Pass 2 under a per-column depth cap (policy MLST_DEPTH_CAP of include/mlst_policy.h; pysam's pileup(max_depth=8000),
 * metaMLST_functions.py:255-259, stated as a rule that does not depend on the order of a BAM file): a column of a
 * chosen allele sees only the first `cap` records that SPAN it -- records ordered by (read index, strand), the span of a
 * record being its aligned allele columns from the first to the last, deleted columns inside included (they sit in a pysam
 * pile-up column as is_del entries and count towards its depth).  Every record of the BAM file counts towards the depth
 * (score >= the aligner's floor), whether or not it passes the AS / XM tag filter; the filters (tags, Phred, ACGT) apply to
 * what was seen.  cap = 0: orc_pileup.  Serial in read order: this is the statement of the rule, not a fast path. */
int orc_pileup_capped(const orc_ref* r, const uint8_t* bases, const uint8_t* quals, const uint64_t* off,
                      uint64_t n_reads, const uint32_t* chosen, uint32_t n_chosen, uint32_t cap, uint32_t* counts, uint32_t* depth_seen) {
    if (cap == 0) return orc_pileup(r, bases, quals, off, n_reads, chosen, n_chosen, counts, 0);
    const uint32_t nL = r->n_loci;
    int64_t* col_base = (int64_t*)malloc(sizeof(int64_t) * nL);
    uint32_t* chosen_of = (uint32_t*)malloc(sizeof(uint32_t) * nL);
    for (uint32_t l = 0; l < nL; l++) { col_base[l] = -1; chosen_of[l] = 0; }
    uint64_t ncols = 0;
    for (uint32_t k = 0; k < n_chosen; k++) {
        uint32_t a = chosen[k]; if (a >= r->n_alleles) { free(col_base); free(chosen_of); return -1; }
        uint32_t L = r->locus_of[a]; if (col_base[L] >= 0) { free(col_base); free(chosen_of); return -1; }
        col_base[L] = (int64_t)ncols; chosen_of[L] = a; ncols += r->len[a];
    }
    memset(counts, 0, sizeof(uint32_t) * 4 * ncols);
    uint32_t* depth = (uint32_t*)calloc(ncols ? ncols : 1, sizeof(uint32_t));      /* records that span the column, all of them */
    uint8_t* seen = (uint8_t*)malloc(MLST_MAX_ALLELE_LEN + 1);
    aln_t* al = (aln_t*)malloc(sizeof(aln_t));
    for (uint64_t ri = 0; ri < n_reads; ri++) {
        int n = (int)(off[ri + 1] - off[ri]);
        if (n > MLST_MAX_READ_LEN || n < K) continue;
        uint8_t code[MLST_MAX_READ_LEN], phred[MLST_MAX_READ_LEN];
        for (int i = 0; i < n; i++) {
            code[i] = base_code(bases[off[ri] + i]);
            int q = (int)quals[off[ri] + i] - 33; phred[i] = (uint8_t)(q < 0 ? 0 : (q > 127 ? 127 : q));
        }
        cand_t items[MLST_MAX_CAND];
        int ni = seed_read(r, code, n, items);
        for (int strand = 0; strand < 2; strand++)      /* the order of a read's records: forward strand first (a read has one item per locus and strand) */
        for (int it = 0; it < ni; it++) {
            if (items[it].strand != strand) continue;
            uint32_t L = items[it].locus; if (col_base[L] < 0) continue;
            uint8_t rb[MLST_MAX_READ_LEN], pen[MLST_MAX_READ_LEN], q[MLST_MAX_READ_LEN];
            orient_read(r, code, phred, n, items[it].strand, rb, pen, q);
            align_pair(r, rb, pen, n, chosen_of[L], items[it].diag, al);
            if (al->score < r->floor_tab[n] || al->score <= 0 || al->n_cols <= 0) continue;      /* no record */
            int j0 = al->cj[0], j1 = al->cj[0];
            for (int k = 1; k < al->n_cols; k++) { if (al->cj[k] < j0) j0 = al->cj[k]; if (al->cj[k] > j1) j1 = al->cj[k]; }
            for (int j = j0; j <= j1; j++) {
                uint32_t* dp = &depth[(uint64_t)col_base[L] + (uint64_t)j];
                seen[j] = *dp < cap; (*dp)++;
            }
            if (al->score < r->prm.minscore || al->xm > r->prm.max_xm) continue;                 /* BAM_tagFilter AS, XM */
            for (int k = 0; k < al->n_cols; k++) {
                int i = al->ci[k], j = al->cj[k];
                if (!seen[j] || rb[i] > 3 || q[i] < r->prm.minqual) continue;
                counts[((uint64_t)col_base[L] + (uint64_t)j) * 4 + rb[i]]++;
            }
        }
    }
    if (depth_seen) memcpy(depth_seen, depth, sizeof(uint32_t) * ncols);
    free(al); free(seen); free(depth); free(col_base); free(chosen_of);
    return 0;
}

/* ------------------------------------------------------------------ exhaustive mode (measurement of the seeding policy)
 *
 * What bowtie2 -a would report if it examined EVERYTHING: for every read, every allele and both strands the best
 * local alignment under the same scoring (match bonus, quality-aware mismatch, N, affine gaps, --gbar), found by a full
 * (unbanded, unseeded) Gotoh recurrence over the whole read x allele matrix.  Same packed values, so XM / XO come out of
 * the maximum.  O(read x allele) per pair: small databases only.  tests/test_seeding_deviation.py compares it with the
 * seeded specification (orc_pass1_dense) to put a number on what seeding every 16th base / one diagonal per
 * (locus, strand) / band 8 leaves out (bowtie2 itself [NOT IN TREE] seeds more densely: -L 20 -i S,1,0.50
 * via /root/reference/README.md:20).
 * out_score / out_xm / out_xo: [n_reads][n_alleles], score 0 = no alignment reaches a positive score. */
static int32_t full_local(const orc_ref* r, const uint8_t* rb, const uint8_t* pen, int n, const uint8_t* ab, int m) {
    const int G = r->prm.gbar;
    const int32_t OPEN = ((int32_t)(r->prm.gap_open + r->prm.gap_ext) << MLST_P_SHIFT) + (1 << 8);
    const int32_t EXT = (int32_t)r->prm.gap_ext << MLST_P_SHIFT;
    int32_t* H = (int32_t*)malloc(sizeof(int32_t) * (size_t)(m + 1) * 2);      /* two rows: previous read base, this one */
    int32_t* F = (int32_t*)malloc(sizeof(int32_t) * (size_t)(m + 1) * 2);
    for (int j = 0; j <= m; j++) { H[j] = MLST_P0; F[j] = MLST_P_NEG; }
    int32_t best = MLST_P0;
    for (int i = 0; i < n; i++) {
        int32_t* Hp = H + (size_t)(i & 1) * (m + 1); int32_t* Hc = H + (size_t)((i + 1) & 1) * (m + 1);
        int32_t* Fp = F + (size_t)(i & 1) * (m + 1); int32_t* Fc = F + (size_t)((i + 1) & 1) * (m + 1);
        const int gap_ok = (i >= G && i < n - G);
        int32_t e = MLST_P_NEG; Hc[0] = MLST_P0; Fc[0] = MLST_P_NEG;
        for (int j = 1; j <= m; j++) {           /* cell (read i, allele j-1) */
            int32_t diag = Hp[j - 1] + col_delta(r, rb[i], pen[i], ab[j - 1]);
            int32_t f = MLST_P_NEG;
            if (gap_ok) { int32_t e1 = Hc[j - 1] - OPEN, e2 = e - EXT; e = e2 > e1 ? e2 : e1;      /* gap in the read: allele base skipped */
                          int32_t f1 = Hp[j] - OPEN, f2 = Fp[j] - EXT; f = f2 > f1 ? f2 : f1; }     /* gap in the allele: read base skipped */
            else e = MLST_P_NEG;
            int32_t h = MLST_P0;
            if (diag > h) h = diag;
            if (e > h) h = e;
            if (f > h) h = f;
            Hc[j] = h; Fc[j] = f;
            if (h > best) best = h;
        }
    }
    free(H); free(F);
    return best;
}

int orc_exhaustive(const orc_ref* r, const uint8_t* bases, const uint8_t* quals, const uint64_t* off, uint64_t n_reads,
                   int16_t* out_score, uint8_t* out_xm, uint8_t* out_xo, int n_threads) {
    const uint32_t nA = r->n_alleles;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    int bad = 0;
    #pragma omp parallel for schedule(dynamic, 8)
    for (int64_t ri = 0; ri < (int64_t)n_reads; ri++) {
        int n = (int)(off[ri + 1] - off[ri]);
        if (n > MLST_MAX_READ_LEN) { bad = 1; continue; }
        uint8_t code[MLST_MAX_READ_LEN], phred[MLST_MAX_READ_LEN], rb[MLST_MAX_READ_LEN], pen[MLST_MAX_READ_LEN], q[MLST_MAX_READ_LEN];
        for (int i = 0; i < n; i++) {
            code[i] = base_code(bases[off[ri] + i]);
            int qq = (int)quals[off[ri] + i] - 33; phred[i] = (uint8_t)(qq < 0 ? 0 : (qq > 127 ? 127 : qq));
        }
        for (uint32_t a = 0; a < nA; a++) {
            int32_t best = MLST_P0;
            for (int strand = 0; strand < 2; strand++) {
                orient_read(r, code, phred, n, strand, rb, pen, q);
                int32_t v = n > 0 ? full_local(r, rb, pen, n, r->code[a], (int)r->len[a]) : MLST_P0;
                if (v > best) best = v;
            }
            size_t o = (size_t)ri * nA + a;
            out_score[o] = (int16_t)(best >> MLST_P_SHIFT); out_xm[o] = (uint8_t)(255 - (best & 0xFF)); out_xo[o] = (uint8_t)(127 - ((best >> 8) & 0x7F));
        }
    }
    return bad ? -5 : 0;
}

/* The seeded specification (what orc_pass1 accumulates) as dense per-(read, allele) tables, for the comparison above:
 * score 0 = the pair was never extended or did not reach a positive score.  A read with work items on both strands of
 * one locus keeps the better record per allele (exhaustive mode reports one value per pair too). */
int orc_pass1_dense(const orc_ref* r, const uint8_t* bases, const uint8_t* quals, const uint64_t* off, uint64_t n_reads,
                    int16_t* out_score, uint8_t* out_xm, uint8_t* out_xo, int n_threads) {
    const uint32_t nA = r->n_alleles;
    memset(out_score, 0, sizeof(int16_t) * (size_t)n_reads * nA); memset(out_xm, 0, (size_t)n_reads * nA); memset(out_xo, 0, (size_t)n_reads * nA);
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    #pragma omp parallel
    {
        aln_t* al = (aln_t*)malloc(sizeof(aln_t));
        #pragma omp for schedule(dynamic, 64)
        for (int64_t ri = 0; ri < (int64_t)n_reads; ri++) {
            int n = (int)(off[ri + 1] - off[ri]);
            if (n > MLST_MAX_READ_LEN || n < K) continue;
            uint8_t code[MLST_MAX_READ_LEN], phred[MLST_MAX_READ_LEN];
            for (int i = 0; i < n; i++) {
                code[i] = base_code(bases[off[ri] + i]);
                int qq = (int)quals[off[ri] + i] - 33; phred[i] = (uint8_t)(qq < 0 ? 0 : (qq > 127 ? 127 : qq));
            }
            cand_t items[MLST_MAX_CAND];
            int ni = seed_read(r, code, n, items);
            for (int it = 0; it < ni; it++) {
                uint8_t rb[MLST_MAX_READ_LEN], pen[MLST_MAX_READ_LEN], q[MLST_MAX_READ_LEN];
                orient_read(r, code, phred, n, items[it].strand, rb, pen, q);
                uint32_t L = items[it].locus;
                for (uint32_t a = r->locus_begin[L]; a < r->locus_begin[L] + r->locus_count[L]; a++) {
                    align_pair(r, rb, pen, n, a, items[it].diag, al);
                    size_t o = (size_t)ri * nA + a;
                    if (al->score > out_score[o]) { out_score[o] = (int16_t)al->score; out_xm[o] = (uint8_t)al->xm; out_xo[o] = (uint8_t)al->xo; }
                }
            }
        }
        free(al);
    }
    return 0;
}

/* ------------------------------------------------------------------ allele match */

/* stringDiff (metaMLST_functions.py:230-234): mismatches over zip(s1, s2) -- the shorter
 * length bounds the comparison and the length difference is not counted (Q10). */
uint32_t orc_string_diff(const uint8_t* s1, uint32_t n1, const uint8_t* s2, uint32_t n2) {
    uint32_t n = n1 < n2 ? n1 : n2, c = 0;
    for (uint32_t i = 0; i < n; i++) if (s1[i] != s2[i]) c++;
    return c;
}

/* The stringDiff scan of metamlst-merge.py:177-181 over every allele of one locus. */
int orc_hamming_all(const orc_ref* r, uint32_t locus, const uint8_t* query, uint32_t len, uint32_t* dist) {
    if (locus >= r->n_loci) return -1;
    for (uint32_t k = 0; k < r->locus_count[locus]; k++) {
        uint32_t a = r->locus_begin[locus] + k;
        dist[k] = orc_string_diff(query, len, r->ascii[a], r->len[a]);
    }
    return 0;
}
