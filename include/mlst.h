/*
 * mlst.h -- C-ABI of the MI355X MLST-typing engine (libmlst_hip.so).
 *
 * The reference (SegataLab/metamlst) has no FFI for this path: the seams it offers are a
 * process pipe and Python functions.  Each entry point below names the
 * reference site it replaces.  Plain pointers and sizes only; the caller owns every
 * buffer; the library never returns owned memory.  Every function returns 0 on success
 * or a negative MLST_E_* code, with text available from mlst_last_error().  A handle is
 * bound to one GPU and is not thread-safe; distinct handles are independent.
 *
 * Binding from the reference's language (Python) is ctypes: see INTEGRATION.md.
 */
#ifndef MLST_H
#define MLST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLST_OK              0
#define MLST_E_INVALID      -1   /* bad argument / bad state */
#define MLST_E_NOGPU        -2   /* no usable HIP device: the product path has no CPU fallback */
#define MLST_E_HIP          -3   /* a HIP runtime call failed */
#define MLST_E_CAPACITY     -4   /* retained-read / item / worklist capacity exceeded */
#define MLST_E_LIMIT        -5   /* input exceeds a packed-format limit (mlst_policy.h) */

typedef struct mlst_handle mlst_handle;

/* Engine parameters.  Defaults (mlst_default_params) are the reference's:
 * minscore/max_xm/min_read_len = argparse defaults metamlst.py:39-41; minqual/mincov =
 * cmseq call metaMLST_functions.py:258; scoring = bowtie2 --very-sensitive-local [NOT IN TREE]. */
typedef struct mlst_params {
    int32_t minscore;          /* accept record iff AS >= minscore            metamlst.py:115 */
    int32_t max_xm;            /* ... and field15 <= max_xm                   metamlst.py:115 */
    int32_t min_read_len;      /* ... and len(SEQ) >= min_read_len            metamlst.py:115 */
    int32_t minqual;           /* pileup base quality floor                   metaMLST_functions.py:258 */
    int32_t mincov;            /* informational (applied host side)           metaMLST_functions.py:258 */
    int32_t match_bonus;       /* bowtie2 --ma */
    int32_t mm_max, mm_min;    /* bowtie2 --mp MX,MN */
    int32_t n_penalty;         /* bowtie2 --np */
    int32_t gap_open, gap_ext; /* bowtie2 --rdg/--rfg (symmetric) */
    int32_t gbar;              /* bowtie2 --gbar */
    int32_t band_w;            /* banded Smith-Waterman half width */
    int32_t gap_trigger_mm;    /* see mlst_policy.h; <0 = always banded SW */
    int32_t xm_field_quirk;    /* 1 = emulate metamlst.py:110 positional parse (Q1) */
    int32_t gap_trigger_clip;  /* see mlst_policy.h */
    double  minscore_const;    /* bowtie2 --score-min G,const,coef */
    double  minscore_coef;
    uint64_t max_retained_reads; /* capacity of the on-locus read store (0 = default) */
    uint64_t max_items;          /* capacity of the (read,locus,strand,diag) item list (0 = default) */
    uint64_t max_pair_results;   /* capacity of the per-(item,allele) result arena (0 = default) */
} mlst_params;

/* counters[] layout of mlst_get_allele_stats */
enum {
    MLST_CNT_TOTAL_RECORDS = 0,  /* totalReads   metamlst.py:130 (alignment records, Q13) */
    MLST_CNT_IGNORED       = 1,  /* ignoredReads metamlst.py:129 */
    MLST_CNT_READS_SEEN    = 2,  /* reads submitted */
    MLST_CNT_CANDIDATES    = 3,  /* reads that passed the seed sieve */
    MLST_CNT_RETAINED      = 4,  /* reads with at least one exact seed (kept for pass 2) */
    MLST_CNT_ITEMS         = 5,  /* (read,locus,strand,diag) work items */
    MLST_CNT_DP_PAIRS      = 6,  /* (item,allele) pairs sent to banded SW */
    MLST_CNT_SIEVE_PASS    = 7,  /* routed sieve: entries that passed the LDS filter and were examined exactly (a tuning figure) */
    MLST_CNT_N             = 8
};

void mlst_default_params(mlst_params* p);

/* Create / destroy an engine bound to HIP device `device`.  Fails with MLST_E_NOGPU when
 * no device is present (there is no CPU path in this library). */
int  mlst_create(int device, const mlst_params* p, mlst_handle** out);
void mlst_destroy(mlst_handle* h);
const char* mlst_last_error(const mlst_handle* h);   /* h may be NULL: last create error */

/* Load the allele reference.  Replaces dump_db_to_fasta + bowtie2-build
 * (metaMLST_functions.py:149-161, metamlst-index.py:222-247).
 *   ascii_concat : all allele sequences concatenated (bytes as stored in alleles.sequence)
 *   off[n+1]     : byte offsets into ascii_concat
 *   locus_id[n]  : dense locus index 0..L-1; alleles of one locus must be contiguous
 *   species_id[n], allele_no[n] : carried for the caller (alleles.bacterium / alleleVariant)
 */
int mlst_load_reference(mlst_handle* h, const uint8_t* ascii_concat, const uint64_t* off,
                        const uint32_t* locus_id, const uint32_t* species_id,
                        const int32_t* allele_no, uint32_t n_alleles);

/* The built host index on disk.  The reference keeps `<idx>.1.bt2` next to its FASTA dump and skips bowtie2-build when the
 * file is there (metamlst-index.py:224-225); here mlst_load_reference reads the index it would build (2-bit arena,
 * block-haplotype tables, seed table, sieves) from `path` when the file's header carries the key of the inputs (hashes of
 * the allele text, offsets, locus and species ids, the sieve switches), and writes the file after a build otherwise.
 * Process-wide; NULL or "" switches it off (the default).  A file that does not fit the inputs is ignored and replaced. */
int mlst_set_reference_cache(const char* path);

/* The host-side index built by the last mlst_load_reference of the process is kept (a second engine on the same
 * database only uploads it: 0.5 s instead of 13 s for the full database); this releases it (~0.6 GB for the full
 * database).  MLST_INDEX_CACHE=0 in the environment disables the cache. */
void mlst_release_index_cache(void);

/* Pass 1 over one batch of reads held in HOST memory (FASTQ fields as read from the file):
 * seed sieve -> exact seeds -> extension against every allele of the hit locus ->
 * per-allele {sum AS, hits}.  Replaces bowtie2 -a ... | samtools view + the hit
 * accumulator loop metamlst.py:96-130.  On-locus reads are retained on the GPU for pass 2.
 *   bases  : ASCII bases concatenated;  quals : ASCII Phred+33 concatenated (same offsets)
 *   off[n+1]; paired != 0 means reads 2k and 2k+1 are mates (share a QNAME, Q3). */
int mlst_submit_reads(mlst_handle* h, const uint8_t* bases, const uint8_t* quals,
                      const uint64_t* off, uint64_t n_reads, int paired);

/* Page-locked host memory for the caller's input buffers (FASTQ text read from files, packed arrays): host-to-device copies
 * from it are DMA transfers at the link's rate, without the runtime's staging copy.  Process-wide; no engine needed. */
int mlst_alloc_host(uint64_t n_bytes, void** out);
int mlst_free_host(void* p);

/* Pass 1 straight from FASTQ TEXT (uncompressed, 4 lines per record, LF or CRLF) held in host memory: the bytes
 * cross PCIe once and are parsed on the GPU (line starts by block newline counts + scan, then packed).  The chunk
 * must hold whole records (cut it after a multiple of four lines).  Replaces the FASTQ reader in front of
 * bowtie2 [NOT IN TREE].  n_reads_out (optional) receives the number of records found. */
int mlst_submit_fastq(mlst_handle* h, const uint8_t* text, uint64_t n_bytes, int paired, uint64_t* n_reads_out);

/* One chunk of an OPEN FASTQ stream: the text continues what the calls before left over (a partial record at the end of
 * a chunk is kept on the device and completed by the next call, whichever of mlst_submit_fastq_stream /
 * mlst_submit_fastq_bgzf it is: text and BGZF chunks of one stream may alternate); a chunk may be cut anywhere.  The last
 * chunk is passed with final_chunk != 0 and must end with a whole record (n_bytes may be 0 then).  Used for the head and
 * tail of a BGZF byte range, which the host inflates itself to find the record boundary (metamlst_amd/fastq.py). */
int mlst_submit_fastq_stream(mlst_handle* h, const uint8_t* text, uint64_t n_bytes, int final_chunk, int paired, uint64_t* n_reads_out);

/* Two chunks of FASTQ text from the two files of a paired-end sample, holding the SAME number of whole records each:
 * record k of text1 and record k of text2 are mates.  They are interleaved on the GPU (reads 2k, 2k+1) and submitted as
 * pairs, i.e. one QNAME per pair for sequenceBank (metamlst.py:127, Q3) -- bowtie2 itself sees them as unpaired reads
 * (-U r1,r2, README.md:20).  n_reads_out receives the number of reads (2 x records per file). */
int mlst_submit_fastq_pair(mlst_handle* h, const uint8_t* text1, uint64_t n1, const uint8_t* text2, uint64_t n2, uint64_t* n_reads_out);

/* The same from BGZF-compressed FASTQ (bgzip; a series of independent <= 64 KiB deflate blocks): the COMPRESSED bytes
 * cross PCIe, the blocks are inflated on the GPU (csrc/inflate_lane.h: one lane per block decodes the codes into tokens,
 * one workgroup per block turns tokens into bytes; csrc/inflate_wave.h for oversized blocks), the text is parsed as above.  A chunk is
 * a run of whole BGZF blocks cut anywhere between blocks; a record that straddles two chunks is completed by the next
 * call; the last chunk of a file is passed with final_chunk != 0 and must end with a whole record.  Block CRCs are
 * not verified (the inflated size is).  With n_consumed_out != NULL a non-final buffer may also end inside a block:
 * the call takes the whole blocks, reports their size, and the caller passes the rest again in front of the next
 * buffer (so a reader never has to walk the block headers itself).  Plain gzip has no block structure to parallelise:
 * inflate it on the host and use mlst_submit_fastq.
 * Three stages on three streams (round 5): a call queues the copy of ITS chunk and the inflate of it, then parses and submits
 * the chunk of the call BEFORE it while the GPU inflates -- so the reads of a non-final chunk enter the statistics with the
 * next call on this handle (whichever entry that is: every entry that looks at or adds to the sample's state finishes an open
 * chunk first), n_reads_out counts the records COMPLETED by the call (their sum over a file is the file's record count), and a
 * corrupt block is reported by the call that finishes its chunk.  `data` may be released when the call returns.
 * MLST_BGZF_PIPE=0 restores the serial behaviour (copy, inflate, parse and submit inside the call). */
int mlst_submit_fastq_bgzf(mlst_handle* h, const uint8_t* data, uint64_t n_bytes, int final_chunk, int paired, uint64_t* n_reads_out,
                           uint64_t* n_consumed_out);

/* Host-packed input: what crosses the link is 2-bit bases + lengths (42 bytes per 150-base read instead of the 316 of its
 * FASTQ text); the Phred rows stay on the host and only those of the reads that pass the seed sieve (one in ~400 of a
 * metagenome) follow.  mlst_pack_fastq_host: FASTQ text (whole 4-line records) -> `packed` in the engine's resident layout
 * (ceil(n / 64) * 64 * words_per_read words: groups of 64 reads, transposed in 8-byte units), `qrows` (n x qual_stride raw
 * Phred, bit 7 = non-ACGT base), `lens` (bit 15 = the read holds such a base) -- byte for byte what mlst_submit_fastq's
 * device-side parser makes of the same text; all host threads (threads <= 0: as many as the machine has, at most 128).
 * No handle: pure host code.  mlst_submit_packed_host: pass 1 from such arrays in HOST memory; one host synchronisation in
 * the middle (the candidate list comes back before the quality rows go out), so this entry is not replayed as a graph.
 * Replaces, like mlst_submit_fastq, the user-run `bowtie2 ... -U <fastq>` of /root/reference/README.md:20. */
int mlst_pack_fastq_host(const uint8_t* text, uint64_t n_bytes, uint32_t words_per_read, uint32_t qual_stride, uint32_t* packed,
                         uint8_t* qrows, uint16_t* lens, uint64_t cap_reads, uint64_t* n_reads_out, int threads);
int mlst_submit_packed_host(mlst_handle* h, const uint32_t* packed, const uint8_t* qrows, const uint16_t* lens, uint64_t n_reads,
                            uint32_t words_per_read, uint32_t qual_stride, int paired);

/* Same, with the three arrays already in DEVICE memory (GPU-side FASTQ decode feeds this). */
int mlst_submit_reads_device(mlst_handle* h, const uint8_t* d_bases, const uint8_t* d_quals,
                             const uint64_t* d_off, uint64_t n_reads, uint32_t max_len, int paired);

/* Pack ASCII reads (device) into the resident format (device): 2-bit bases in rows of
 * words_per_read uint32 (even; base k at bits 2(k%16) of word k/16, A=0 C=1 G=2 T=3), quality rows
 * of qual_stride bytes holding raw Phred with bit 7 set for a non-ACGT base, and uint16 lengths
 * (bit 15 = the read holds a non-ACGT base).  This is the layout SURVEY.md 8(d) prices at 38+150 B
 * per 150 bp read.
 * The 2-bit rows are stored in groups of 64 reads, transposed in 8-byte units so that 64 lanes
 * owning the 64 reads of a group load every unit with one coalesced 512-byte access: words 2u and
 * 2u+1 of read r are the uint32 at
 *     (r / 64) * 64 * words_per_read  +  ((u * 64 + r % 64) * 2)   and the one after it.
 * qual_stride must be a multiple of 4 (rows are written as 32-bit words) and d_qual_rows 4-byte aligned.
 * d_packed must hold ceil(n_reads / 64) * 64 * words_per_read words (the rows that pad the last
 * group are written as zeros) and be 16-byte aligned; a batch that is packed in pieces must cut the
 * pieces at multiples of 64 reads. */
int mlst_pack_reads_device(mlst_handle* h, const uint8_t* d_bases, const uint8_t* d_quals,
                           const uint64_t* d_off, uint64_t n_reads,
                           uint32_t* d_packed, uint8_t* d_qual_rows, uint16_t* d_lens,
                           uint32_t words_per_read, uint32_t qual_stride);

/* Pass 1 over a batch already resident in the packed format (the benchmark's timed entry). */
int mlst_submit_packed_device(mlst_handle* h, const uint32_t* d_packed, const uint8_t* d_qual_rows,
                              const uint16_t* d_lens, uint64_t n_reads,
                              uint32_t words_per_read, uint32_t qual_stride, int paired);

/* Read back pass-1 statistics.  Replaces the `cel` / `sequenceBank` dictionaries of
 * metamlst.py:116-127.  Any pointer may be NULL.
 *   sum_score[n_alleles]  int64  : sum of AS over accepted records of the allele
 *   n_hits[n_alleles]     uint32 : number of accepted records
 *   locus_read_len_sum[L] uint64 : sum of len(SEQ) over reads with an accepted record on the locus
 *   locus_first_read[L]   uint64 : smallest read index with an accepted record (UINT64_MAX if none; Q6)
 *   counters[MLST_CNT_N]  uint64 */
int mlst_get_allele_stats(mlst_handle* h, int64_t* sum_score, uint32_t* n_hits,
                          uint64_t* locus_read_len_sum, uint64_t* locus_first_read,
                          uint64_t* counters);

/* Multi-GPU: copy the additive statistics to / from a caller-owned DEVICE buffer so the
 * host can all-reduce them with torch.distributed (RCCL).  Layout of d_sum (int64):
 * [sum_score(n_alleles) | n_hits(n_alleles) | locus_read_len_sum(L) | counters(MLST_CNT_N)];
 * d_min (int64, reduce with MIN): [locus_first_read(L)] (INT64_MAX if none). */
int mlst_stats_flat_sizes(mlst_handle* h, uint64_t* n_sum, uint64_t* n_min);
int mlst_export_stats_device(mlst_handle* h, int64_t* d_sum, int64_t* d_min);
int mlst_import_stats_device(mlst_handle* h, const int64_t* d_sum, const int64_t* d_min);

/* Pass 2: pileup of the retained reads against one chosen allele per locus.  Replaces
 * cmseq get_base_stats over pysam pileup (metaMLST_functions.py:255-259).
 *   chosen_allele_idx[n] : indices into the loaded allele list (at most one per locus)
 *   counts : uint32[sum(len(chosen))][4] A,C,G,T, alleles in the order given.
 * A base is counted iff its record has AS >= minscore and XM <= max_xm (true XM),
 * Phred >= minqual and base in ACGT. */
int mlst_pileup(mlst_handle* h, const uint32_t* chosen_allele_idx, uint32_t n, uint32_t* counts);
int mlst_pileup_device(mlst_handle* h, const uint32_t* chosen_allele_idx, uint32_t n,
                       uint32_t* d_counts /* device, n_cols*4, zeroed by the call */, uint64_t* n_cols);
/* pysam's pileup(max_depth = 8000) (metaMLST_functions.py:255-259; policy MLST_DEPTH_CAP of mlst_policy.h) as a switch.
 * cap = 0 (default): every record counts.  cap = n: a column of a chosen allele sees only the first n records that span it
 * (aligned columns first to last, deleted columns inside included), records ordered by (read index, strand); every record
 * of the aligner counts towards the depth, the AS / XM / Phred / ACGT filters apply to what a column saw.  Applies to every
 * pile-up that follows (mlst_pileup*, mlst_consensus, mlst_typing_*); ~40 extra passes over the work items, so a
 * literal-parity mode.  Single-engine samples only: a rank of a sharded sample knows its own reads' records only. */
int mlst_set_depth_cap(mlst_handle* h, uint32_t cap);

/* Pass 2 with the majority rule applied on the GPU: the string cmseq's
 * reference_free_consensus(mincov, noneCharacter, ...) returns for each chosen contig
 * (metaMLST_functions.py:258-259), concatenated in the order given.  A column with fewer than
 * mincov counted bases is none_char; ties resolve A < C < G < T (MLST_TIE_ORDER).
 *   out_seq : sum(len(chosen)) bytes;  counts (optional, may be NULL): as mlst_pileup. */
int mlst_consensus(mlst_handle* h, const uint32_t* chosen_allele_idx, uint32_t n, uint32_t mincov,
                   char none_char, uint8_t* out_seq, uint32_t* counts);

/* Multi-GPU variant: majority rule over pileup counts that already sit in DEVICE memory (the all-reduced
 * counts of every rank); out_seq is host memory, n_cols bytes. */
int mlst_consensus_from_counts_device(mlst_handle* h, const uint32_t* d_counts, uint64_t n_cols, uint32_t mincov,
                                      char none_char, uint8_t* out_seq);

/* Pass 2 for ready-made alignments (a SAM / BAM produced by the documented bowtie2 command):
 * counts as mlst_pileup, over the records given.  Replaces the cmseq / pysam pileup of the BAM
 * (metaMLST_functions.py:255-259 [cmseq NOT IN TREE]): a base counts when its record's AS >= minscore and
 * XM <= max_xm (true tags), its Phred >= minqual and it is A/C/G/T; secondary records count (stepper 'nofilter').
 *   rec_allele  : allele index of the record's contig      rec_pos0 : leftmost reference position, 0-based
 *   cigar       : len << 4 | op with the BAM operation codes (M I D N S H P = X), cigar_off[n_rec+1] delimits records
 *   seq / qual  : ASCII bases and raw Phred (not +33), seq_off[n_rec+1] delimits records (SEQ as stored in SAM,
 *                 i.e. already on the reference strand)
 *   counts      : host, sum(len(chosen)) * 4 uint32. */
int mlst_pileup_alignments(mlst_handle* h, const uint32_t* chosen_allele_idx, uint32_t n_chosen, uint64_t n_rec,
                           const uint32_t* rec_allele, const int32_t* rec_pos0, const int32_t* rec_as, const int32_t* rec_xm,
                           const uint64_t* cigar_off, const uint32_t* cigar, const uint64_t* seq_off,
                           const uint8_t* seq, const uint8_t* qual, int32_t minscore, int32_t max_xm, int32_t minqual,
                           uint32_t* counts);

/* ---- whole typing tail on the device, without a host round trip between the passes ----------------------
 * mlst_typing_enqueue queues, behind the pass-1 work already submitted on the engine's stream:
 *   the allele choice of metamlst.py:133-151 + :244 (per locus the allele with the highest
 *   round((sum AS - (maxHits - hits) * penalty) / hits, 1), ties to the lowest allele number, computed exactly
 *   as Python's round() does -- see mlst_round_tenths), pass 2 against the chosen alleles, the majority
 *   consensus of mlst_consensus, and the copies to the host.  It returns at once.
 * mlst_typing_fetch waits for it and returns the statistics of mlst_get_allele_stats, chosen[n_loci] (allele
 *   index, -1 = locus without an accepted record) and the consensus letters; the letters of locus l start at
 *   colbase[l] of mlst_typing_layout (one slot of the locus' longest allele per locus; only the first
 *   len(chosen allele) bytes of a slot are meaningful).
 * The choice is part of the host logic of the reference (float round + tie-break); the Python host keeps its
 * own statement of it (typing.pick_alleles_fast) and the tests compare the two. */
int mlst_typing_layout(mlst_handle* h, uint64_t* colbase /* n_loci + 1 */, uint64_t* total_cols);
int mlst_typing_enqueue(mlst_handle* h, int32_t penalty, uint32_t mincov, char none_char);
/* The two halves of mlst_typing_enqueue, for a multi-GPU caller that all-reduces the pileup counts in between:
 * choice + pileup into d_counts (device, total_cols * 4 uint32, zeroed by the call; NULL = internal buffer), then
 * consensus over d_counts + the copies to the host. */
int mlst_typing_choose_pileup(mlst_handle* h, int32_t penalty, uint32_t* d_counts);
int mlst_typing_finish(mlst_handle* h, uint32_t mincov, char none_char, const uint32_t* d_counts);
/* The same halves with the counts in a COMPACT layout, for the all-reduce in between: only the loci with a chosen
 * allele get a slot (of the locus' longest allele), in locus order -- every rank holds the same statistics after the
 * first exchange, chooses the same alleles and so derives the same layout.  d_counts: device, cap_cols * 4 uint32,
 * zeroed by the call.  The capacity is fixed by the caller before the need is known (the size of a collective is a
 * host decision): mlst_typing_compact_info, after mlst_typing_fetch, returns the columns needed and whether they
 * fitted.  If they did not, nothing was piled up, no letter of that fetch is valid, and the caller repeats both halves
 * with at least need_cols columns (mlst_typing_layout's total always suffices); statistics and choice are unaffected.
 * mlst_typing_fetch returns the letters in the fixed layout of mlst_typing_layout either way.  (No reference
 * counterpart: the reference is one process; what is exchanged are the counts behind cmseq's consensus,
 * metaMLST_functions.py:255-259.) */
int mlst_typing_choose_pileup_compact(mlst_handle* h, int32_t penalty, uint32_t* d_counts, uint64_t cap_cols);
int mlst_typing_finish_compact(mlst_handle* h, uint32_t mincov, char none_char, const uint32_t* d_counts);
int mlst_typing_compact_info(mlst_handle* h, uint64_t* need_cols, uint32_t* overflow);
int mlst_typing_fetch(mlst_handle* h, int64_t* sum_score, uint32_t* n_hits, uint64_t* locus_read_len_sum,
                      uint64_t* locus_first_read, uint64_t* counters, int32_t* chosen, uint8_t* letters);
/* mlst_typing_fetch in two halves: mlst_typing_wait waits for the queued typing step; its results stay in pinned memory (two
 * slots, written in turn) while the caller queues the engine's next step; mlst_typing_fetch_waited copies them out without
 * waiting.  (An engine on its own share of the CUs idles between the end of a step and the submission of the next.) */
int mlst_typing_wait(mlst_handle* h);
int mlst_typing_fetch_waited(mlst_handle* h, int64_t* sum_score, uint32_t* n_hits, uint64_t* locus_read_len_sum,
                             uint64_t* locus_first_read, uint64_t* counters, int32_t* chosen, uint8_t* letters);
/* round(float(p) / float(q), 1) of Python as an exact integer number of tenths (host function, the same code
 * the device uses); 0 when q == 0. */
long long mlst_round_tenths(long long p, uint32_t q);

/* Allele match: Hamming distance of `query` against every allele of `locus`, semantics of
 * stringDiff (metaMLST_functions.py:230-234: zip truncates, length difference not counted),
 * as used by metamlst-merge.py:177-181.  Outputs the first allele (load order) within z,
 * or -1, and the number of alleles within z. */
int mlst_hamming_le(mlst_handle* h, uint32_t locus, const uint8_t* query, uint32_t len,
                    uint32_t z, int32_t* first_allele_idx, uint32_t* n_within);
/* Full distance vector (dist[n_alleles_of_locus]) for tests and reports. */
int mlst_hamming_all(mlst_handle* h, uint32_t locus, const uint8_t* query, uint32_t len,
                     uint32_t* dist);

/* Multi-GPU: index of this rank's first read in the whole sample, so that locus_first_read (the
 * first-seen order of metamlst.py's dicts, Q6) is global.  Call after mlst_reset_sample / before submitting. */
int mlst_set_read_index_base(mlst_handle* h, uint64_t base);

/* Forget reads and statistics, keep the reference (next sample). */
int mlst_reset_sample(mlst_handle* h);

/* ---- introspection (tests, bench) ---- */
typedef struct mlst_item {     /* one (read, locus, strand, diagonal) unit of extension work */
    uint64_t read_index;       /* index of the read in submission order */
    uint32_t locus;
    int32_t  diag;             /* allele position minus (oriented) read position */
    uint16_t strand;           /* 1 = read reverse-complemented */
    uint16_t votes;
    uint32_t reserved;
} mlst_item;
int mlst_get_items(mlst_handle* h, mlst_item* out, uint64_t cap, uint64_t* n);

/* Run the engine on the caller's HIP stream (hipStream_t; NULL = back on the engine's own stream).  Work queued so
 * far is waited for.  With the engine on the stream a torch.distributed collective is ordered against, a multi-GPU
 * step needs no host synchronisation between its kernels and its collectives (metamlst_amd/dist.py). */
int mlst_set_stream(mlst_handle* h, void* stream);
/* The engine's own stream restricted to share `part` of `n_parts` equal shares of the device's CUs (hipExtStreamCreateWithCUMask;
 * n_parts = 1: the whole device).  Engines of one process on disjoint shares run side by side instead of taking turns;
 * mlst_get_stream hands the stream out (e.g. for torch.cuda.ExternalStream), mlst_busy asks without waiting whether work
 * queued on the engine's current stream is still running (1) or not (0).  No reference counterpart (one process, one CPU). */
int mlst_set_cu_partition(mlst_handle* h, uint32_t part, uint32_t n_parts);
int mlst_get_stream(mlst_handle* h, void** stream);
int mlst_busy(mlst_handle* h);
/* mlst_export_stats_device / mlst_import_stats_device without the host synchronisation. */
int mlst_export_stats_device_async(mlst_handle* h, int64_t* d_sum, int64_t* d_min);
int mlst_import_stats_device_async(mlst_handle* h, const int64_t* d_sum, const int64_t* d_min);

int mlst_set_profiling(mlst_handle* h, int on);   /* 0 = off, 1 = events + sieve window, 2 = sieve window only (keeps the hipGraph replay of the launch sequences, which event profiling turns off) */
/* Per-kernel device time measured with HIP events on the engine's stream.
 * which: 0=sieve (all its kernels) 1=seed 2=extend (k_extend + k_extend_pairs) 3=banded-SW 4=accumulate 5=pileup 6=pack 12=k_ext_prep (the item records of k_extend); 9 = k_route, 10 =
 * k_route_probe and 11 = k_route_verify, the three kernels of the routed sieve (inside 0) (events bracket the launch on the engine's
 * stream, so with several engines on one GPU they include the time a kernel queues behind another stream's kernel);
 * 7 = the sieve's execution window measured inside the kernel (wall clock at the first workgroup's start and the last
 * one's end; one submission per sample), added up when the sample's statistics are fetched;
 * 8 = the longest residency of one workgroup of that launch (LDS sieve; one workgroup per CU, each doing an equal share):
 * what the launch takes once its workgroups run -- when it shares the GPU with another stream's kernel its workgroups
 * start one by one as CUs free up, which stretches the window (7) without the kernel being any slower. */
int mlst_get_kernel_time(mlst_handle* h, int which, double* total_ms, uint64_t* launches);
int mlst_reset_kernel_time(mlst_handle* h);
/* Bytes of the device-resident index structures: [0]=allele arena [1]=sieve [2]=seed table;
 * [3]=fill of the LDS first-level bitmap in parts per million (0 when the plain sieve kernel is in use) */
int mlst_get_index_bytes(mlst_handle* h, uint64_t out[4]);
/* The seed sieve chosen for the loaded database: [0] = kind (0 = half-seed bitmaps in LDS, 1 = hashed bitmap in global
 * memory, 3 = CU-routed filter slices in LDS; chosen by database size, MLST_SIEVE=lds / global / routed forces one), [1] = distinct canonical seeds, [2] = longest overflow walk of a key in the
 * fingerprint sieve (the kernels follow a chain for 64 buckets; the build keeps it <= 32), [3] = sieve buckets. */
int mlst_get_sieve_info(mlst_handle* h, uint64_t out[4]);
/* The block-haplotype tables k_extend scores against (SURVEY 8 f1 "locus backbone / variant-column table"; they replace
 * aligning a read against every allele one by one, which is what a bowtie2 -a run over the FASTA of
 * metaMLST_functions.py:149-161 does): [0] = distinct 32-base block haplotypes of all loci, [1] = bytes of the tables
 * (also counted in mlst_get_index_bytes [0]), [2] = loci that have them, [3] / [4] = most haplotypes in any run of 6 / 11
 * blocks (what a read of <= 160 / <= 320 bases covers), [5] / [6] = bytes of LDS a work item of k_extend_160 / _320 gets
 * for the summaries (MLST_EXT_LDS_KB bounds it; loci that need more are aligned pair by pair), [7] = threads per work item. */
int mlst_get_extend_info(mlst_handle* h, uint64_t out[8]);
/* Block until all work queued on the engine's stream is done. */
int mlst_synchronize(mlst_handle* h);

#ifdef __cplusplus
}
#endif
#endif
