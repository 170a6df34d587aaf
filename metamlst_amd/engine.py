"""ctypes binding of libmlst_hip.so (include/mlst.h) -- the only compute path of the package.

There is no CPU fallback: if the HIP library is missing or no GPU is present the
constructor raises.  The oracle under oracle/ is test infrastructure and is never
imported from here.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .index import AlleleIndex
from .typing import SampleStats

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MLST_LIB", os.path.join(_HERE, "libmlst_hip.so"))   # MLST_LIB: profiling builds only

MLST_CNT_N = 8
CNT_TOTAL_RECORDS, CNT_IGNORED, CNT_READS_SEEN, CNT_CANDIDATES, CNT_RETAINED, CNT_ITEMS, CNT_DP_PAIRS = range(7)
KERNELS = ("sieve", "seed", "extend", "banded_sw", "accumulate", "pileup", "pack", "sieve_inkernel", "sieve_wg_longest",
           "sieve_route", "sieve_probe", "sieve_verify", "extend_prep")
SIEVE_KINDS = ("lds", "global", "binned (round 1, removed)", "routed")


class MlstParams(C.Structure):
    """struct mlst_params (include/mlst.h)."""
    _fields_ = [("minscore", C.c_int32), ("max_xm", C.c_int32), ("min_read_len", C.c_int32),
                ("minqual", C.c_int32), ("mincov", C.c_int32), ("match_bonus", C.c_int32),
                ("mm_max", C.c_int32), ("mm_min", C.c_int32), ("n_penalty", C.c_int32),
                ("gap_open", C.c_int32), ("gap_ext", C.c_int32), ("gbar", C.c_int32),
                ("band_w", C.c_int32), ("gap_trigger_mm", C.c_int32), ("xm_field_quirk", C.c_int32),
                ("gap_trigger_clip", C.c_int32), ("minscore_const", C.c_double), ("minscore_coef", C.c_double),
                ("max_retained_reads", C.c_uint64), ("max_items", C.c_uint64), ("max_pair_results", C.c_uint64)]


class MlstItem(C.Structure):
    """struct mlst_item (include/mlst.h)."""
    _fields_ = [("read_index", C.c_uint64), ("locus", C.c_uint32), ("diag", C.c_int32),
                ("strand", C.c_uint16), ("votes", C.c_uint16), ("reserved", C.c_uint32)]


def default_params() -> MlstParams:
    """Defaults of mlst_policy.h, usable without loading the library (tests, oracle)."""
    p = MlstParams()
    p.minscore, p.max_xm, p.min_read_len, p.minqual, p.mincov = 80, 5, 50, 20, 1
    p.match_bonus, p.mm_max, p.mm_min, p.n_penalty = 2, 6, 2, 1
    p.gap_open, p.gap_ext, p.gbar, p.band_w = 5, 3, 4, 8
    p.gap_trigger_mm, p.xm_field_quirk, p.gap_trigger_clip = 12, 1, 8
    p.minscore_const, p.minscore_coef = 20.0, 8.0
    p.max_retained_reads = p.max_items = p.max_pair_results = 0
    return p


class MlstError(RuntimeError):
    pass


_lib = None


def load_library(path: str | None = None):
    """Load libmlst_hip.so and declare every prototype of include/mlst.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    # The PyTorch ROCm wheel bundles its own HIP / HSA runtime, this library links the system one.  Both can live in
    # one process only if torch's runtime initialises first ("No HIP GPUs are available" otherwise), so when torch
    # is installed it is imported before the engine is loaded.  torch is not needed by the engine itself.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(path):
        raise MlstError("HIP engine library not built: %s is missing (run __graft_entry__.build()); "
                        "there is no CPU fallback" % path)
    lib = C.CDLL(path)
    vp, u8p, u64p, u32p, i32p, i64p, u16p = (C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
    H = C.c_void_p
    sig = {
        "mlst_default_params": (None, [C.POINTER(MlstParams)]),
        "mlst_create": (C.c_int, [C.c_int, C.POINTER(MlstParams), C.POINTER(H)]),
        "mlst_destroy": (None, [H]),
        "mlst_last_error": (C.c_char_p, [H]),
        "mlst_load_reference": (C.c_int, [H, u8p, u64p, u32p, u32p, i32p, C.c_uint32]),
        "mlst_set_reference_cache": (C.c_int, [C.c_char_p]),
        "mlst_submit_reads": (C.c_int, [H, u8p, u8p, u64p, C.c_uint64, C.c_int]),
        "mlst_submit_fastq": (C.c_int, [H, u8p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]),
        "mlst_submit_fastq_bgzf": (C.c_int, [H, u8p, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "mlst_selftest_inflate": (C.c_int, [u8p, C.c_uint64, u8p, C.c_uint64, C.POINTER(C.c_uint64)]),
        "mlst_selftest_inflate_canon": (C.c_int, [u8p, C.c_uint64, u8p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
        "mlst_debug_bgzf_walk": (C.c_int, [u8p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
        "mlst_selftest_inflate_device": (C.c_int, [H, u8p, C.c_uint64, u8p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
        "mlst_submit_reads_device": (C.c_int, [H, u8p, u8p, u64p, C.c_uint64, C.c_uint32, C.c_int]),
        "mlst_pack_reads_device": (C.c_int, [H, u8p, u8p, u64p, C.c_uint64, u32p, u8p, u16p, C.c_uint32, C.c_uint32]),
        "mlst_submit_packed_device": (C.c_int, [H, u32p, u8p, u16p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]),
        "mlst_pack_fastq_host": (C.c_int, [u8p, C.c_uint64, C.c_uint32, C.c_uint32, u32p, u8p, u16p, C.c_uint64, C.POINTER(C.c_uint64), C.c_int]),
        "mlst_submit_packed_host": (C.c_int, [H, u32p, u8p, u16p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]),
        "mlst_alloc_host": (C.c_int, [C.c_uint64, C.POINTER(C.c_void_p)]),
        "mlst_free_host": (C.c_int, [C.c_void_p]),
        "mlst_get_allele_stats": (C.c_int, [H, i64p, u32p, u64p, u64p, u64p]),
        "mlst_stats_flat_sizes": (C.c_int, [H, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "mlst_export_stats_device": (C.c_int, [H, i64p, i64p]),
        "mlst_import_stats_device": (C.c_int, [H, i64p, i64p]),
        "mlst_pileup": (C.c_int, [H, u32p, C.c_uint32, u32p]),
        "mlst_set_depth_cap": (C.c_int, [H, C.c_uint32]),
        "mlst_pileup_device": (C.c_int, [H, u32p, C.c_uint32, u32p, C.POINTER(C.c_uint64)]),
        "mlst_consensus": (C.c_int, [H, u32p, C.c_uint32, C.c_uint32, C.c_char, u8p, u32p]),
        "mlst_consensus_from_counts_device": (C.c_int, [H, u32p, C.c_uint64, C.c_uint32, C.c_char, u8p]),
        "mlst_pileup_alignments": (C.c_int, [H, u32p, C.c_uint32, C.c_uint64, u32p, i32p, i32p, i32p, u64p, u32p, u64p, u8p, u8p,
                                             C.c_int32, C.c_int32, C.c_int32, u32p]),
        "mlst_typing_layout": (C.c_int, [H, u64p, C.POINTER(C.c_uint64)]),
        "mlst_typing_enqueue": (C.c_int, [H, C.c_int32, C.c_uint32, C.c_char]),
        "mlst_typing_choose_pileup": (C.c_int, [H, C.c_int32, u32p]),
        "mlst_typing_finish": (C.c_int, [H, C.c_uint32, C.c_char, u32p]),
        "mlst_typing_choose_pileup_compact": (C.c_int, [H, C.c_int32, u32p, C.c_uint64]),
        "mlst_typing_finish_compact": (C.c_int, [H, C.c_uint32, C.c_char, u32p]),
        "mlst_typing_compact_info": (C.c_int, [H, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
        "mlst_set_stream": (C.c_int, [H, C.c_void_p]),
        "mlst_set_cu_partition": (C.c_int, [H, C.c_uint32, C.c_uint32]),
        "mlst_get_stream": (C.c_int, [H, C.POINTER(C.c_void_p)]),
        "mlst_busy": (C.c_int, [H]),
        "mlst_export_stats_device_async": (C.c_int, [H, i64p, i64p]),
        "mlst_import_stats_device_async": (C.c_int, [H, i64p, i64p]),
        "mlst_typing_fetch": (C.c_int, [H, i64p, u32p, u64p, u64p, u64p, i32p, u8p]),
        "mlst_typing_wait": (C.c_int, [H]),
        "mlst_typing_fetch_waited": (C.c_int, [H, i64p, u32p, u64p, u64p, u64p, i32p, u8p]),
        "mlst_round_tenths": (C.c_longlong, [C.c_longlong, C.c_uint32]),
        "mlst_hamming_le": (C.c_int, [H, C.c_uint32, u8p, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]),
        "mlst_hamming_all": (C.c_int, [H, C.c_uint32, u8p, C.c_uint32, u32p]),
        "mlst_reset_sample": (C.c_int, [H]),
        "mlst_set_read_index_base": (C.c_int, [H, C.c_uint64]),
        "mlst_get_items": (C.c_int, [H, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
        "mlst_set_profiling": (C.c_int, [H, C.c_int]),
        "mlst_get_kernel_time": (C.c_int, [H, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
        "mlst_reset_kernel_time": (C.c_int, [H]),
        "mlst_get_index_bytes": (C.c_int, [H, C.POINTER(C.c_uint64)]),
        "mlst_get_sieve_info": (C.c_int, [H, C.POINTER(C.c_uint64)]),
        "mlst_get_extend_info": (C.c_int, [H, C.POINTER(C.c_uint64)]),
        "mlst_release_index_cache": (None, []),
        "mlst_submit_fastq_stream": (C.c_int, [H, u8p, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
        "mlst_submit_fastq_pair": (C.c_int, [H, u8p, C.c_uint64, u8p, C.c_uint64, C.POINTER(C.c_uint64)]),
        "mlst_get_route_trace": (C.c_int, [H, u64p, C.c_uint64, C.POINTER(C.c_uint64)]),
        "mlst_debug_route_realloc": (C.c_int, [H, C.c_uint64]),
        "mlst_synchronize": (C.c_int, [H]),
    }
    tolerant = bool(os.environ.get("MLST_LIB_ALLOW_MISSING"))      # A/B runs against an older build (profiles/ab.sh)
    for name, (res, args) in sig.items():
        if tolerant and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)          # AttributeError here = the library does not export the ABI
        fn.restype, fn.argtypes = res, args
    lib._mlst_symbols = tuple(sig)
    if path == LIB_PATH:
        _lib = lib
    return lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def pack_fastq_host(text, read_len_max: int = 160, threads: int = 0):
    """FASTQ text (bytes / uint8 array of whole records) -> (packed, qrows, lens, n_reads, words_per_read, qual_stride): the engine's
    resident read format made on the host by all its threads (mlst_pack_fastq_host), for Engine.submit_packed_host."""
    lib = load_library()
    buf = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else text
    wpr = ((int(read_len_max) + 31) // 32) * 2
    qstride = (int(read_len_max) + 7) // 8 * 8
    cap = int(buf.size // 8 + 64)                       # a record is at least 8 bytes
    cap = min(cap, int(np.count_nonzero(buf[:1 << 20] == 10) * (buf.size / max(1, min(buf.size, 1 << 20))) / 4 * 1.1) + 4096)
    packed = np.zeros(((cap + 63) // 64) * 64 * wpr, np.uint32)
    qrows = np.empty((cap, qstride), np.uint8)
    lens = np.empty(cap, np.uint16)
    n = C.c_uint64()
    rc = lib.mlst_pack_fastq_host(_ptr(buf), buf.size, wpr, qstride, _ptr(packed), _ptr(qrows), _ptr(lens), cap, C.byref(n), int(threads))
    if rc == -4:                                        # MLST_E_CAPACITY: the estimate from the first megabyte was too low
        cap = int(buf.size // 8 + 64)
        packed = np.zeros(((cap + 63) // 64) * 64 * wpr, np.uint32); qrows = np.empty((cap, qstride), np.uint8); lens = np.empty(cap, np.uint16)
        rc = lib.mlst_pack_fastq_host(_ptr(buf), buf.size, wpr, qstride, _ptr(packed), _ptr(qrows), _ptr(lens), cap, C.byref(n), int(threads))
    if rc != 0:
        raise MlstError("mlst_pack_fastq_host failed (%d): not whole 4-line FASTQ records, or a read longer than %d bases" % (rc, read_len_max))
    return packed, qrows, lens, int(n.value), wpr, qstride


def pinned_array(n_bytes: int) -> np.ndarray:
    """uint8 array of n_bytes in page-locked host memory (mlst_alloc_host); the memory is released when the array is collected.
    Needs the GPU runtime: raises MlstError where there is none."""
    import weakref
    lib = load_library()
    p = C.c_void_p()
    rc = lib.mlst_alloc_host(int(n_bytes), C.byref(p))
    if rc != 0 or not p.value:
        raise MlstError("mlst_alloc_host(%d) failed (%d)" % (n_bytes, rc))
    arr = np.frombuffer((C.c_uint8 * int(n_bytes)).from_address(p.value), np.uint8)
    weakref.finalize(arr, lib.mlst_free_host, C.c_void_p(p.value)).atexit = False      # (at interpreter exit the runtime may be gone: the OS takes the pages)
    return arr


class Engine:
    """One GPU's typing engine.  Mirrors the reference's per-sample flow:
    load_reference (index) -> submit_reads* (alignment + hit accumulation) -> stats ->
    pileup (consensus counts) -> hamming_le (allele match)."""

    def __init__(self, device: int = 0, params: MlstParams | None = None):
        self.lib = load_library()
        self.params = params or default_params()
        self._h = C.c_void_p()
        rc = self.lib.mlst_create(int(device), C.byref(self.params), C.byref(self._h))
        if rc != 0:
            msg = self.lib.mlst_last_error(None)
            raise MlstError("mlst_create failed (%d): %s" % (rc, msg.decode() if msg else "?"))
        self.device = device
        self.depth_cap = 0
        self.index: AlleleIndex | None = None

    def close(self):
        if self._h:
            self.lib.mlst_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.mlst_last_error(self._h)
            raise MlstError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))

    # ---- reference ----
    def load_reference(self, index: AlleleIndex, cache_path: str | None = None):
        """cache_path: where the built host index is kept between runs (mlst_set_reference_cache: `<database>.mlstref` next to the
        database, as the reference keeps `<idx>.1.bt2`, metamlst-index.py:224-225); None leaves the process-wide setting alone."""
        self.index = index
        if cache_path is not None:
            self.lib.mlst_set_reference_cache(cache_path.encode() if cache_path else None)
        self._check(self.lib.mlst_load_reference(self._h, _ptr(index.ascii_concat), _ptr(index.off), _ptr(index.locus_id),
                                                 _ptr(index.species_id), _ptr(index.allele_no), index.n_alleles),
                    "mlst_load_reference")

    # ---- pass 1 ----
    def submit_reads(self, bases: np.ndarray, quals: np.ndarray, off: np.ndarray, paired: bool = False):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        self._check(self.lib.mlst_submit_reads(self._h, _ptr(bases), _ptr(quals), _ptr(off), len(off) - 1, int(paired)),
                    "mlst_submit_reads")

    def submit_fastq(self, text, paired: bool = False) -> int:
        """Pass 1 straight from FASTQ text (bytes / bytearray / uint8 array holding whole 4-line records); parsed on the GPU."""
        buf = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, np.uint8)
        n = C.c_uint64()
        self._check(self.lib.mlst_submit_fastq(self._h, _ptr(buf) if buf.size else None, buf.size, int(paired), C.byref(n)), "mlst_submit_fastq")
        return int(n.value)

    def submit_fastq_bgzf(self, data, final: bool, paired: bool = False) -> int:
        """Pass 1 from BGZF-compressed FASTQ: a run of whole BGZF blocks (fastq.bgzf_chunks), inflated and parsed on the GPU.
        final marks the last chunk of the file.  Returns the number of records completed by this chunk."""
        buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, np.uint8)
        n = C.c_uint64()
        self._check(self.lib.mlst_submit_fastq_bgzf(self._h, _ptr(buf) if buf.size else None, buf.size, int(final), int(paired), C.byref(n), None),
                    "mlst_submit_fastq_bgzf")
        return int(n.value)

    def _submit_bgzf_pieces(self, path: str, lo: int, hi: int, chunk_bytes: int, paired: bool, final_last: bool) -> int:
        """Compressed bytes [lo, hi) of a BGZF file (whole blocks from lo on) into mlst_submit_fastq_bgzf, piece by piece: a
        reader thread stays two pieces ahead (fastq.raw_chunks: positional reads of a few threads into pooled buffers), the
        library takes the whole blocks of a piece and says how many bytes that was, and the cut-off block is copied in front
        of the next piece (the buffers keep a margin for it).  Until round 4 every piece went through f.read() and
        `carry + block`: ~0.15 s of one host thread per 256 MB piece against ~0.04 s of GPU work."""
        from .fastq import prefetch, raw_chunks, release_buffers
        margin = 1 << 17
        total, carry = 0, None
        n, used = C.c_uint64(), C.c_uint64()
        ring: list = []
        at = lo
        if hi <= lo:
            self._check(self.lib.mlst_submit_fastq_bgzf(self._h, None, 0, int(final_last), int(paired), C.byref(n), None), "mlst_submit_fastq_bgzf")
            return int(n.value)
        for buf, got in prefetch(raw_chunks(path, chunk_bytes, lo, hi, margin, reuse=True, ring=ring)):
            at += got
            c = 0 if carry is None else carry.size
            if c > margin:
                raise MlstError("a BGZF block of more than %d bytes?" % margin)
            if c:
                buf[margin - c:margin] = carry
            view = buf[margin - c:margin + got]
            last = at >= hi
            self._check(self.lib.mlst_submit_fastq_bgzf(self._h, _ptr(view), view.size, int(last and final_last), int(paired),
                                                        C.byref(n), None if last else C.byref(used)), "mlst_submit_fastq_bgzf")
            total += int(n.value)
            carry = None if last else view[int(used.value):].copy()      # (a cut-off block: under 64 KB)
        release_buffers(ring)
        return total

    def submit_fastq_bgzf_file(self, path: str, paired: bool = False, chunk_bytes: int = (512 << 20) - (1 << 18)) -> int:
        """A whole bgzip'd FASTQ file (see _submit_bgzf_pieces).  Chunks of just under 512 MB compressed (~44 k BGZF blocks; with the
        reader's margin a chunk fills a 512 MB page-locked buffer, four of them per walk): one pass of the inflate kernels with
        three waves per CU (k_inflate_tok2 holds four: 65,536 blocks a turn), and the chunk behind it is copied and inflated
        meanwhile; the library cuts a first chunk's head and a last chunk's tail off as pieces of their own (mlst_submit_fastq_bgzf)."""
        return self._submit_bgzf_pieces(path, 0, os.path.getsize(path), chunk_bytes, paired, True)

    def inflate_bgzf(self, data) -> bytes:
        """Test hook: whole BGZF blocks -> their text, inflated by the device kernel."""
        buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, np.uint8)
        out = np.empty(max(1, (buf.size // 18 + 1) * 65536), np.uint8)
        n, ms = C.c_uint64(), C.c_double()
        self._check(self.lib.mlst_selftest_inflate_device(self._h, _ptr(buf), buf.size, _ptr(out), out.size, C.byref(n), C.byref(ms)), "mlst_selftest_inflate_device")
        self.last_inflate_ms = float(ms.value)
        return out[:int(n.value)].tobytes()

    def submit_fastq_stream(self, text, final: bool, paired: bool = False) -> int:
        """One chunk of an open FASTQ stream (cut anywhere; a partial record at its end is completed by the next chunk, text or
        BGZF); final marks the last one.  Returns the number of records completed."""
        buf = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, np.uint8)
        n = C.c_uint64()
        self._check(self.lib.mlst_submit_fastq_stream(self._h, _ptr(buf) if buf.size else None, buf.size, int(final), int(paired), C.byref(n)),
                    "mlst_submit_fastq_stream")
        return int(n.value)

    def submit_fastq_pair(self, text1, text2) -> int:
        """Two FASTQ chunks with the same number of whole records (fastq.pair_chunks): mates are interleaved on the GPU and
        submitted as pairs.  Returns the number of reads."""
        b1 = np.frombuffer(text1, dtype=np.uint8) if not isinstance(text1, np.ndarray) else np.ascontiguousarray(text1, np.uint8)
        b2 = np.frombuffer(text2, dtype=np.uint8) if not isinstance(text2, np.ndarray) else np.ascontiguousarray(text2, np.uint8)
        n = C.c_uint64()
        self._check(self.lib.mlst_submit_fastq_pair(self._h, _ptr(b1) if b1.size else None, b1.size, _ptr(b2) if b2.size else None, b2.size, C.byref(n)),
                    "mlst_submit_fastq_pair")
        return int(n.value)

    def submit_fastq_bgzf_range(self, path: str, lo: int, hi: int, chunk_bytes: int = 256 << 20) -> int:
        """The records of a bgzip'd FASTQ that start in the BGZF blocks starting in compressed bytes [lo, hi): the boundary
        blocks are inflated on the host to find the record boundaries (fastq.bgzf_range_plan), everything between goes to
        the GPU compressed.  N ranks given consecutive ranges read every record exactly once."""
        from .fastq import bgzf_range_plan
        plan = bgzf_range_plan(path, lo, hi)
        total = 0
        first, end = plan["mid"]
        if plan["head"]:
            total += self.submit_fastq_stream(plan["head"], final=(first >= end and not plan["tail"]))
        if end > first:
            total += self._submit_bgzf_pieces(path, first, end, chunk_bytes, False, not plan["tail"])
        if plan["tail"]:
            total += self.submit_fastq_stream(plan["tail"], final=True)
        return total

    def submit_reads_device(self, d_bases: int, d_quals: int, d_off: int, n_reads: int, max_len: int, paired: bool = False):
        self._check(self.lib.mlst_submit_reads_device(self._h, d_bases, d_quals, d_off, n_reads, max_len, int(paired)),
                    "mlst_submit_reads_device")

    def pack_reads_device(self, d_bases: int, d_quals: int, d_off: int, n_reads: int, d_packed: int, d_qual_rows: int,
                          d_lens: int, words_per_read: int, qual_stride: int):
        self._check(self.lib.mlst_pack_reads_device(self._h, d_bases, d_quals, d_off, n_reads, d_packed, d_qual_rows, d_lens,
                                                    words_per_read, qual_stride), "mlst_pack_reads_device")

    def submit_packed_device(self, d_packed: int, d_qual_rows: int, d_lens: int, n_reads: int, words_per_read: int,
                             qual_stride: int, paired: bool = False):
        self._check(self.lib.mlst_submit_packed_device(self._h, d_packed, d_qual_rows, d_lens, n_reads, words_per_read,
                                                       qual_stride, int(paired)), "mlst_submit_packed_device")

    def submit_packed_host(self, packed: np.ndarray, qrows: np.ndarray, lens: np.ndarray, n_reads: int, words_per_read: int, qual_stride: int,
                           paired: bool = False):
        """Pass 1 from host-packed arrays (pack_fastq_host): bases and lengths cross the link, Phred rows of the sieve's
        candidates only."""
        self._check(self.lib.mlst_submit_packed_host(self._h, _ptr(packed), _ptr(qrows), _ptr(lens), int(n_reads), int(words_per_read),
                                                     int(qual_stride), 1 if paired else 0), "mlst_submit_packed_host")

    def stats(self) -> SampleStats:
        nA, nL = self.index.n_alleles, self.index.n_loci
        s = SampleStats(np.zeros(nA, np.int64), np.zeros(nA, np.uint32), np.zeros(nL, np.uint64),
                        np.zeros(nL, np.uint64), np.zeros(MLST_CNT_N, np.uint64))
        self._check(self.lib.mlst_get_allele_stats(self._h, _ptr(s.sum_score), _ptr(s.n_hits), _ptr(s.locus_len_sum),
                                                   _ptr(s.locus_first), _ptr(s.counters)), "mlst_get_allele_stats")
        return s

    def pileup_alignments(self, chosen, rec_allele, rec_pos0, rec_as, rec_xm, cigar_off, cigar, seq_off, seq, qual,
                          minscore: int = 80, max_xm: int = 5, minqual: int = 20) -> dict[int, np.ndarray]:
        """Pass 2 over ready-made alignment records (metamlst_amd.samin); {allele idx: uint32[len, 4]}."""
        ch = np.ascontiguousarray(chosen, np.uint32)
        lens = [int(self.index.off[a + 1] - self.index.off[a]) for a in ch]
        counts = np.zeros((max(1, sum(lens)), 4), np.uint32)
        arr = [np.ascontiguousarray(x, t) for x, t in ((rec_allele, np.uint32), (rec_pos0, np.int32), (rec_as, np.int32), (rec_xm, np.int32),
                                                       (cigar_off, np.uint64), (cigar, np.uint32), (seq_off, np.uint64), (seq, np.uint8), (qual, np.uint8))]
        self._check(self.lib.mlst_pileup_alignments(self._h, _ptr(ch), len(ch), len(arr[0]), *[_ptr(x) for x in arr],
                                                    int(minscore), int(max_xm), int(minqual), _ptr(counts)), "mlst_pileup_alignments")
        out, at = {}, 0
        for a, L in zip(ch, lens):
            out[int(a)] = counts[at:at + L]
            at += L
        return out

    # ---- typing tail on the device ----
    def typing_enqueue(self, penalty: int = 100, mincov: int = 1, none_char: str = "N"):
        """Queue allele choice + pileup + consensus + host copies behind the submitted pass 1 (returns at once)."""
        self._check(self.lib.mlst_typing_enqueue(self._h, int(penalty), int(mincov), none_char.encode()), "mlst_typing_enqueue")

    def typing_choose_pileup(self, penalty: int = 100, d_counts: int = 0):
        self._check(self.lib.mlst_typing_choose_pileup(self._h, int(penalty), d_counts or None), "mlst_typing_choose_pileup")

    def typing_finish(self, mincov: int = 1, none_char: str = "N", d_counts: int = 0):
        self._check(self.lib.mlst_typing_finish(self._h, int(mincov), none_char.encode(), d_counts or None), "mlst_typing_finish")

    def typing_choose_pileup_compact(self, penalty: int, d_counts: int, cap_cols: int):
        """Allele choice + pileup into the compact layout (loci with a chosen allele only) of a buffer of cap_cols columns."""
        self._check(self.lib.mlst_typing_choose_pileup_compact(self._h, int(penalty), d_counts, int(cap_cols)), "mlst_typing_choose_pileup_compact")

    def typing_finish_compact(self, mincov: int, none_char: str, d_counts: int):
        self._check(self.lib.mlst_typing_finish_compact(self._h, int(mincov), none_char.encode(), d_counts), "mlst_typing_finish_compact")

    def typing_compact_info(self) -> tuple[int, bool]:
        """After typing_fetch: (columns the compact layout needed, True when they did not fit the buffer)."""
        need, over = C.c_uint64(), C.c_uint32()
        self._check(self.lib.mlst_typing_compact_info(self._h, C.byref(need), C.byref(over)), "mlst_typing_compact_info")
        return int(need.value), bool(over.value)

    def typing_total_cols(self) -> int:
        tot = C.c_uint64()
        self._check(self.lib.mlst_typing_layout(self._h, None, C.byref(tot)), "mlst_typing_layout")
        return int(tot.value)

    def set_stream(self, stream: int = 0):
        """Run the engine on a caller's HIP stream (e.g. torch.cuda.Stream.cuda_stream); 0 = its own stream again."""
        self._check(self.lib.mlst_set_stream(self._h, stream or None), "mlst_set_stream")

    def set_cu_partition(self, part: int, n_parts: int):
        """The engine's own stream on share `part` of `n_parts` equal shares of the CUs (n_parts = 1: the whole device)."""
        self._check(self.lib.mlst_set_cu_partition(self._h, int(part), int(n_parts)), "mlst_set_cu_partition")

    def own_stream(self) -> int:
        p = C.c_void_p()
        self._check(self.lib.mlst_get_stream(self._h, C.byref(p)), "mlst_get_stream")
        return int(p.value or 0)

    def busy(self) -> bool:
        rc = self.lib.mlst_busy(self._h)
        if rc < 0:
            self._check(rc, "mlst_busy")
        return rc == 1

    def export_stats_device_async(self, d_sum: int, d_min: int):
        self._check(self.lib.mlst_export_stats_device_async(self._h, d_sum, d_min), "mlst_export_stats_device_async")

    def import_stats_device_async(self, d_sum: int, d_min: int):
        self._check(self.lib.mlst_import_stats_device_async(self._h, d_sum, d_min), "mlst_import_stats_device_async")

    def typing_wait(self):
        """Wait for the queued typing step; its results stay in the engine (typing_fetch(waited=True)) while the next step is queued."""
        self._check(self.lib.mlst_typing_wait(self._h), "mlst_typing_wait")

    def typing_fetch(self, per_allele: bool = True, waited: bool = False):
        """-> (SampleStats, {locus: chosen allele idx}, {allele idx: consensus bytes}) of the last typing_enqueue.
        per_allele=False leaves sum_score / n_hits out (empty arrays): a caller that takes choice and consensus from the
        device needs the per-locus figures only, and 12 bytes per allele of a large database are 4 MB to copy per sample."""
        nA, nL = self.index.n_alleles, self.index.n_loci
        if getattr(self, "_colbase", None) is None or len(self._colbase) != nL + 1:
            self._colbase = np.zeros(nL + 1, np.uint64)
            tot = C.c_uint64()
            self._check(self.lib.mlst_typing_layout(self._h, _ptr(self._colbase), C.byref(tot)), "mlst_typing_layout")
            self._cb_list = [int(x) for x in self._colbase]
        nA_out = nA if per_allele else 0
        s = SampleStats(np.empty(nA_out, np.int64), np.empty(nA_out, np.uint32), np.empty(nL, np.uint64),
                        np.empty(nL, np.uint64), np.empty(MLST_CNT_N, np.uint64))
        chosen = np.empty(nL, np.int32)
        letters = np.empty(self._cb_list[-1], np.uint8)
        fn = self.lib.mlst_typing_fetch_waited if waited else self.lib.mlst_typing_fetch
        self._check(fn(self._h, _ptr(s.sum_score) if per_allele else None, _ptr(s.n_hits) if per_allele else None,
                       _ptr(s.locus_len_sum), _ptr(s.locus_first), _ptr(s.counters), _ptr(chosen), _ptr(letters)), "mlst_typing_fetch")
        raw = letters.tobytes()
        ch, let = {}, {}
        off = self.index.off
        for l in np.nonzero(chosen >= 0)[0].tolist():
            a = int(chosen[l]); ch[l] = a
            b = self._cb_list[l]
            let[a] = raw[b:b + int(off[a + 1] - off[a])]
        return s, ch, let

    def flat_sizes(self) -> tuple[int, int]:
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self.lib.mlst_stats_flat_sizes(self._h, C.byref(a), C.byref(b)), "mlst_stats_flat_sizes")
        return int(a.value), int(b.value)

    def export_stats_device(self, d_sum: int, d_min: int):
        self._check(self.lib.mlst_export_stats_device(self._h, d_sum, d_min), "mlst_export_stats_device")

    def import_stats_device(self, d_sum: int, d_min: int):
        self._check(self.lib.mlst_import_stats_device(self._h, d_sum, d_min), "mlst_import_stats_device")

    # ---- pass 2 ----
    def set_depth_cap(self, cap: int) -> None:
        """Policy MLST_DEPTH_CAP as a switch (pysam max_depth, metaMLST_functions.py:255-259): 0 = off; n = a column sees
        the first n records that span it in (read index, strand) order.  Every pile-up that follows."""
        self._check(self.lib.mlst_set_depth_cap(self._h, int(cap)), "mlst_set_depth_cap")
        self.depth_cap = int(cap)

    def pileup(self, chosen: list[int]) -> dict[int, np.ndarray]:
        """{allele idx: uint32[len, 4]} for the chosen alleles (A,C,G,T columns)."""
        ch = np.ascontiguousarray(chosen, dtype=np.uint32)
        if any(int(a) >= self.index.n_alleles for a in chosen):
            raise MlstError("mlst_pileup: chosen allele out of range")
        lens = [int(self.index.off[a + 1] - self.index.off[a]) for a in chosen]
        counts = np.zeros((sum(lens), 4), np.uint32)
        self._check(self.lib.mlst_pileup(self._h, _ptr(ch), len(chosen), _ptr(counts)), "mlst_pileup")
        out, at = {}, 0
        for a, L in zip(chosen, lens):
            out[int(a)] = counts[at:at + L]
            at += L
        return out

    def consensus(self, chosen: list[int], mincov: int = 1, none_char: str = "N") -> dict[int, bytes]:
        """{allele idx: consensus bytes}: what cmseq's reference_free_consensus returns per contig (majority base,
        none_char below mincov), computed on the GPU from the pileup."""
        ch = np.ascontiguousarray(chosen, dtype=np.uint32)
        if any(int(a) >= self.index.n_alleles for a in chosen):
            raise MlstError("mlst_consensus: chosen allele out of range")
        lens = [int(self.index.off[a + 1] - self.index.off[a]) for a in chosen]
        out = np.zeros(max(1, sum(lens)), np.uint8)
        self._check(self.lib.mlst_consensus(self._h, _ptr(ch), len(chosen), int(mincov), none_char.encode(), _ptr(out), None), "mlst_consensus")
        res, at = {}, 0
        buf = out.tobytes()
        for a, L in zip(chosen, lens):
            res[int(a)] = buf[at:at + L]
            at += L
        return res

    def consensus_from_counts_device(self, d_counts: int, n_cols: int, mincov: int = 1, none_char: str = "N") -> bytes:
        out = np.zeros(max(1, n_cols), np.uint8)
        self._check(self.lib.mlst_consensus_from_counts_device(self._h, d_counts, n_cols, int(mincov), none_char.encode(), _ptr(out)),
                    "mlst_consensus_from_counts_device")
        return out.tobytes()[:n_cols]

    def pileup_device(self, chosen: list[int], d_counts: int) -> int:
        ch = np.ascontiguousarray(chosen, dtype=np.uint32)
        n = C.c_uint64()
        self._check(self.lib.mlst_pileup_device(self._h, _ptr(ch), len(chosen), d_counts, C.byref(n)), "mlst_pileup_device")
        return int(n.value)

    # ---- allele match ----
    def hamming_le(self, locus: int, query: bytes, z: int) -> tuple[int, int]:
        q = np.frombuffer(query, dtype=np.uint8)
        first, nw = C.c_int32(), C.c_uint32()
        self._check(self.lib.mlst_hamming_le(self._h, locus, _ptr(q) if len(q) else None, len(q), z, C.byref(first), C.byref(nw)),
                    "mlst_hamming_le")
        return int(first.value), int(nw.value)

    def hamming_all(self, locus: int, query: bytes) -> np.ndarray:
        q = np.frombuffer(query, dtype=np.uint8)
        if not 0 <= locus < self.index.n_loci:
            raise MlstError("mlst_hamming_all: locus %d out of range" % locus)
        d = np.zeros(int(self.index.locus_count[locus]), np.uint32)
        self._check(self.lib.mlst_hamming_all(self._h, locus, _ptr(q) if len(q) else None, len(q), _ptr(d)), "mlst_hamming_all")
        return d

    # ---- misc ----
    def reset_sample(self):
        self._check(self.lib.mlst_reset_sample(self._h), "mlst_reset_sample")

    def set_read_index_base(self, base: int):
        self._check(self.lib.mlst_set_read_index_base(self._h, int(base)), "mlst_set_read_index_base")

    def items(self, cap: int = 1 << 20) -> np.ndarray:
        buf = (MlstItem * cap)()
        n = C.c_uint64()
        self._check(self.lib.mlst_get_items(self._h, C.cast(buf, C.c_void_p), cap, C.byref(n)), "mlst_get_items")
        k = min(cap, int(n.value))
        return np.array([(b.read_index, b.locus, b.strand, b.diag, b.votes) for b in buf[:k]], dtype=np.int64).reshape(-1, 5)

    def set_profiling(self, on):
        """0 / False = off, 1 / True = HIP events + sieve window, 2 = sieve window only (hipGraph replay stays on)."""
        self._check(self.lib.mlst_set_profiling(self._h, int(on)), "mlst_set_profiling")

    def kernel_time(self, which: int | str) -> tuple[float, int]:
        w = KERNELS.index(which) if isinstance(which, str) else which
        ms, n = C.c_double(), C.c_uint64()
        self._check(self.lib.mlst_get_kernel_time(self._h, w, C.byref(ms), C.byref(n)), "mlst_get_kernel_time")
        return float(ms.value), int(n.value)

    def reset_kernel_time(self):
        self._check(self.lib.mlst_reset_kernel_time(self._h), "mlst_reset_kernel_time")

    def index_bytes(self) -> list[int]:
        out = (C.c_uint64 * 4)()
        self._check(self.lib.mlst_get_index_bytes(self._h, out), "mlst_get_index_bytes")
        return [int(x) for x in out]

    def sieve_info(self) -> dict:
        """{kind, n_seeds, longest_chain, buckets} of the loaded database's seed sieve."""
        out = (C.c_uint64 * 4)()
        self._check(self.lib.mlst_get_sieve_info(self._h, out), "mlst_get_sieve_info")
        return {"kind": SIEVE_KINDS[int(out[0])], "n_seeds": int(out[1]), "longest_chain": int(out[2]), "buckets": int(out[3])}

    def extend_info(self) -> dict:
        """The block-haplotype tables of the loaded database (mlst_get_extend_info)."""
        out = (C.c_uint64 * 8)()
        self._check(self.lib.mlst_get_extend_info(self._h, out), "mlst_get_extend_info")
        keys = ("haplotypes", "bytes", "loci", "window6", "window11", "lds_bytes_160", "lds_bytes_320", "threads")
        return {k: int(v) for k, v in zip(keys, out)}

    def route_trace(self):
        """Diagnostics of the routed sieve's last submission (None until the trace, switched on by the first call, has
        seen one): {P, arena, packed, khz, cap, filter, flags, arena_entries, wg: uint64[P + 256][4]}."""
        n = C.c_uint64()
        self._check(self.lib.mlst_get_route_trace(self._h, None, 0, C.byref(n)), "mlst_get_route_trace")
        if not n.value:
            return None
        out = np.zeros(int(n.value), np.uint64)
        self._check(self.lib.mlst_get_route_trace(self._h, _ptr(out), out.size, C.byref(n)), "mlst_get_route_trace")
        keys = ("P", "arena", "packed", "khz", "cap", "filter", "flags", "arena_entries")
        d = {k: int(out[i]) for i, k in enumerate(keys)}
        d["wg"] = out[8:].reshape(-1, 4)
        return d

    def debug_route_realloc(self, pad_bytes: int = 0):
        """pad_bytes = -1: keep the old arena allocated (the new one is other memory for certain)."""
        self._check(self.lib.mlst_debug_route_realloc(self._h, int(pad_bytes) & 0xFFFFFFFFFFFFFFFF), "mlst_debug_route_realloc")

    def synchronize(self):
        self._check(self.lib.mlst_synchronize(self._h), "mlst_synchronize")
