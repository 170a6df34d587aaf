/*
 * mlst_policy.h -- named policy constants of the MLST-typing hot path.
 *
 * Constants only (no code): included by the HIP engine (metamlst_amd/csrc) and by the
 * CPU oracle (oracle/) so that both state the same specification.  Every value marked
 * [NOT IN TREE] restates the documented default of a dependency whose source is not under
 * /root/reference (bowtie2, cmseq, htslib); SURVEY.md section 8(c) lists them as
 * "parity unpinned".  Values marked with a reference file:line are pinned by the tree.
 */
#ifndef MLST_POLICY_H
#define MLST_POLICY_H

/* ---- MetaMLST filter defaults (metamlst.py:38-42, metaMLST_functions.py:258) ---- */
#define MLST_DEF_MINSCORE      80   /* --minscore      metamlst.py:39 */
#define MLST_DEF_MAX_XM         5   /* --max_xM        metamlst.py:40 */
#define MLST_DEF_MIN_READ_LEN  50   /* --min_read_len  metamlst.py:41 */
#define MLST_DEF_PENALTY      100   /* --penalty       metamlst.py:38 (host side only) */
#define MLST_DEF_MINQUAL       20   /* minqual=20      metaMLST_functions.py:258 */
#define MLST_DEF_MINCOV         1   /* mincov=1        metaMLST_functions.py:258 */

/* ---- bowtie2 --very-sensitive-local scoring [NOT IN TREE; wiki command via README.md:20] ---- */
#define MLST_DEF_MATCH_BONUS    2   /* --ma 2 (local mode) */
#define MLST_DEF_MM_MAX         6   /* --mp 6,2 : penalty = MN + floor((MX-MN)*min(Q,40)/40) */
#define MLST_DEF_MM_MIN         2
#define MLST_DEF_N_PENALTY      1   /* --np 1 ; an N column also counts in XM */
#define MLST_DEF_GAP_OPEN       5   /* --rdg 5,3 and --rfg 5,3 : a gap of length g costs 5+3g */
#define MLST_DEF_GAP_EXT        3
#define MLST_DEF_GBAR           4   /* --gbar 4 : no gap within 4 positions of either read end */
#define MLST_DEF_MINSCORE_CONST 20.0 /* --score-min G,20,8 : record exists iff AS >= (long)(20+8 ln L) */
#define MLST_DEF_MINSCORE_COEF   8.0

/* ---- engine seeding / extension policy (this engine's own; replaces bowtie2's heuristics) ---- */
#define MLST_SEED_LEN          20   /* exact seed length (bowtie2 -L 20 in --very-sensitive-local) */
#define MLST_SEED_STEP         16   /* seed every 16 read bases = one packed 32-bit word            */
#define MLST_MAX_POSTINGS      16   /* a seed with more distinct (locus,strand,pos) is repetitive: dropped */
#define MLST_MAX_CAND           8   /* distinct (locus,strand,diag) vote bins kept per read, first-seen order */
#define MLST_MIN_VOTES          1   /* seeds needed on the winning diagonal */
#define MLST_DEF_BAND_W         8   /* banded SW half-width around the voted diagonal */
#define MLST_DEF_GAP_TRIGGER_MM 12  /* banded SW runs iff the seed diagonal looks broken by an indel:
                                       ungapped full-overlap mismatches > this, the ungapped local score
                                       reaches the bowtie2 floor, and ... (next constant);
                                       a negative value means "always run banded SW" */
#define MLST_DEF_GAP_TRIGGER_CLIP 8 /* ... the ungapped local alignment leaves at least this many overlap
                                       columns unaligned (an indel clips the alignment; scattered SNPs do not),
                                       and at least half of those unaligned columns are mismatches (past an indel
                                       the diagonal looks random, ~75 % mismatches; a clipped SNP cluster does not) */
#define MLST_DEF_XM_FIELD_QUIRK 1   /* metamlst.py:110 reads SAM column 15 by position: it is XM only
                                       when XS:i is present (read has >= 2 records), else XO (Q1) */

/* ---- cmseq / pysam pileup policy [NOT IN TREE] ---- */
#define MLST_TIE_ORDER "ACGT"       /* majority-base ties resolve alphabetically (host side) */
#define MLST_DEPTH_CAP          0   /* default of mlst_set_depth_cap: pysam's max_depth=8000 depends on the order of the BAM
                                       file; 0 = not applied.  As a switch it is "the first n records that span a column,
                                       in (read index, strand) order" -- oracle: orc_pileup_capped */

/* ---- hard limits of the packed formats ---- */
#define MLST_MAX_READ_LEN     320   /* 20 packed words; xm field of the packed score is 8 bits */
#define MLST_MAX_ALLELE_LEN  4095   /* 12-bit position in a seed posting */
#define MLST_MAX_LOCI      262143   /* 18-bit locus id in a seed posting */

/* ---- packed DP value: (score << 16) | ((127 - xo) << 8) | (255 - xm) ----
 * One signed 32-bit max() orders by score, then fewer gap opens, then fewer mismatches. */
#define MLST_P_SHIFT   16
#define MLST_P0        0x7FFF              /* empty alignment: score 0, xo 0, xm 0 */
#define MLST_P_NEG     (-(1 << 29))

#endif
