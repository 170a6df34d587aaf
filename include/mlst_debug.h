/*
 * mlst_debug.h -- diagnostics and test hooks of libmlst_hip.so.
 *
 * NOT part of the drop-in boundary (include/mlst.h): nothing here has a counterpart in the reference, no data path calls
 * these, and a binding of the product needs none of them.  They are exported by the same library because the tests
 * (tests/test_inflate.py) and the profiling scripts (profiles/route_modes.py, profiles/inflate_rate.py) drive the device
 * code through them.
 */
#ifndef MLST_DEBUG_H
#define MLST_DEBUG_H

#include "mlst.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: the deflate decoder of the call above run on the HOST on one raw deflate stream (returns 0 or a negative
 * code of csrc/inflate_dev.h; *produced = bytes written).  Not a data path. */
int mlst_selftest_inflate(const uint8_t* in, uint64_t n_in, uint8_t* out, uint64_t cap, uint64_t* produced);

/* Test hook of the DEVICE decoder (csrc/inflate_wave.h): whole BGZF blocks in, their inflated text out (host buffers);
 * kernel_ms (optional) receives the duration of the inflate kernel alone (HIP events). */
int mlst_selftest_inflate_device(mlst_handle* h, const uint8_t* data, uint64_t n_bytes, uint8_t* out, uint64_t cap, uint64_t* produced, double* kernel_ms);

/* The decoder of k_inflate_tok2 (csrc/inflate_canon.h: canonical limits, 800 bytes of state per stream) run on the HOST on one
 * raw deflate stream, its tokens replayed into bytes.  *left_to_other_kernel = 1: the literal / length code of a block holds
 * more symbols than the decoder's 192-entry table (the device leaves such a block to k_inflate); nothing is produced then. */
int mlst_selftest_inflate_canon(const uint8_t* in, uint64_t n_in, uint8_t* out, uint64_t cap, uint64_t* produced, int* left_to_other_kernel);

/* The BGZF blocks of a chunk as mlst_submit_fastq_bgzf lists them (host code only, no device): blocks with data, the bytes they
 * inflate to, and -- a chunk of 32 MB or more is walked by four threads, three of them from a block start they find behind their
 * quarter mark -- how many of the four lists counted (a list counts only where the chain of the one before it lands on its first
 * block).  MLST_E_INVALID where the serial walk meets something that is not a whole block.  tests/test_inflate.py. */
int mlst_debug_bgzf_walk(const uint8_t* data, uint64_t n_bytes, uint64_t* n_blocks, uint64_t* text_bytes, int* lists_taken);

/* Diagnostics of the routed sieve (profiles/route_modes.py; no reference counterpart, not a data path).
 * mlst_get_route_trace: the first call switches the trace on; later calls wait for the stream and return, for the last
 * submission, out[0] = producer workgroups P, [1] = arena address, [2] = packed-row address, [3] = wall-clock kHz,
 * [4] = region capacity, [5] = filter address, [6] = flag address, [7] = arena capacity in entries, then four words per
 * workgroup (P producers, then the 256 consumers): XCC_ID | HW_ID << 32, wall clock at start, at end, 0.
 * *n_words = words needed (0 while nothing has been traced).
 * mlst_debug_route_realloc: free the routing arena (the next submission allocates it again), keeping pad_bytes of
 * device memory allocated in between so that the new arena lands elsewhere; pad_bytes = UINT64_MAX keeps the old arena itself
 * allocated (until mlst_destroy), so that the new one is different memory for certain. */
int mlst_get_route_trace(mlst_handle* h, uint64_t* out, uint64_t cap_words, uint64_t* n_words);
int mlst_debug_route_realloc(mlst_handle* h, uint64_t pad_bytes);

#ifdef __cplusplus
}
#endif
#endif
