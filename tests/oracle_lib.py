"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE.  Only tests/, smoke() and
bench.py's cpu_baseline leg import this module; the product package never does."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from metamlst_amd.engine import MLST_CNT_N, MlstItem, MlstParams, default_params
from metamlst_amd.typing import SampleStats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "liboracle.so")


def build_oracle():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(ROOT, "oracle", "mlst_oracle.c")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle())
        vp = C.c_void_p
        _lib.orc_ref_build.restype = vp
        _lib.orc_ref_build.argtypes = [vp, vp, vp, C.c_uint32, C.POINTER(MlstParams)]
        _lib.orc_ref_free.argtypes = [vp]
        _lib.orc_ref_n_keys.restype = C.c_uint64
        _lib.orc_ref_n_keys.argtypes = [vp]
        _lib.orc_pass1.restype = C.c_int
        _lib.orc_pass1.argtypes = [vp, vp, vp, vp, C.c_uint64, C.c_uint64, vp, vp, vp, vp, vp, vp, C.c_uint64,
                                   C.POINTER(C.c_uint64), C.c_int, C.c_int]
        _lib.orc_accumulate_records.restype = C.c_int
        _lib.orc_accumulate_records.argtypes = [vp, C.c_uint64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        _lib.orc_pileup.restype = C.c_int
        _lib.orc_pileup.argtypes = [vp, vp, vp, vp, C.c_uint64, vp, C.c_uint32, vp, C.c_int]
        _lib.orc_pileup_capped.restype = C.c_int
        _lib.orc_pileup_capped.argtypes = [vp, vp, vp, vp, C.c_uint64, vp, C.c_uint32, C.c_uint32, vp, vp]
        _lib.orc_align_one.restype = C.c_int
        _lib.orc_align_one.argtypes = [vp, vp, vp, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, vp, vp, vp]
        for fn in ("orc_exhaustive", "orc_pass1_dense"):
            getattr(_lib, fn).restype = C.c_int
            getattr(_lib, fn).argtypes = [vp, vp, vp, vp, C.c_uint64, vp, vp, vp, C.c_int]
        _lib.orc_string_diff.restype = C.c_uint32
        _lib.orc_string_diff.argtypes = [vp, C.c_uint32, vp, C.c_uint32]
        _lib.orc_hamming_all.restype = C.c_int
        _lib.orc_hamming_all.argtypes = [vp, C.c_uint32, vp, C.c_uint32, vp]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """CPU restatement of the path, same call shape as metamlst_amd.engine.Engine."""

    def __init__(self, index, params: MlstParams | None = None, threads: int = 0):
        self.index = index
        self.params = params or default_params()
        self.threads = threads
        self._r = lib().orc_ref_build(_p(index.ascii_concat), _p(index.off), _p(index.locus_id), index.n_alleles,
                                      C.byref(self.params))
        if not self._r:
            raise RuntimeError("orc_ref_build failed")
        self._reads = None

    def __del__(self):
        try:
            if self._r:
                lib().orc_ref_free(self._r)
        except Exception:
            pass

    def n_keys(self) -> int:
        return int(lib().orc_ref_n_keys(self._r))

    def submit_reads(self, bases, quals, off, paired=False, read_base: int = 0):
        self._reads = (np.ascontiguousarray(bases, np.uint8), np.ascontiguousarray(quals, np.uint8),
                       np.ascontiguousarray(off, np.uint64))
        self._read_base = read_base
        self._paired = bool(paired)

    def stats(self, want_items: int = 0):
        b, q, off = self._reads
        nA, nL = self.index.n_alleles, self.index.n_loci
        s = SampleStats(np.zeros(nA, np.int64), np.zeros(nA, np.uint32), np.zeros(nL, np.uint64),
                        np.zeros(nL, np.uint64), np.zeros(MLST_CNT_N, np.uint64))
        items = (MlstItem * max(1, want_items))()
        n_items = C.c_uint64()
        rc = lib().orc_pass1(self._r, _p(b), _p(q), _p(off), len(off) - 1, getattr(self, '_read_base', 0), _p(s.sum_score), _p(s.n_hits),
                             _p(s.locus_len_sum), _p(s.locus_first), _p(s.counters),
                             C.cast(items, C.c_void_p) if want_items else None, want_items, C.byref(n_items), self.threads,
                             int(getattr(self, "_paired", False)))
        if rc != 0:
            raise RuntimeError("orc_pass1 rc=%d" % rc)
        if want_items:
            k = min(want_items, int(n_items.value))
            arr = np.array([(x.read_index, x.locus, x.strand, x.diag, x.votes) for x in items[:k]], dtype=np.int64).reshape(-1, 5)
            return s, arr
        return s

    def _dense(self, fn):
        b, q, off = self._reads
        n = len(off) - 1
        sc = np.zeros((n, self.index.n_alleles), np.int16)
        xm = np.zeros((n, self.index.n_alleles), np.uint8)
        xo = np.zeros((n, self.index.n_alleles), np.uint8)
        rc = getattr(lib(), fn)(self._r, _p(b), _p(q), _p(off), n, _p(sc), _p(xm), _p(xo), self.threads)
        if rc != 0:
            raise RuntimeError("%s rc=%d" % (fn, rc))
        return sc, xm, xo

    def exhaustive(self):
        """(score, xm, xo)[n_reads, n_alleles]: best local alignment of every read against every allele, no seeding, no band."""
        return self._dense("orc_exhaustive")

    def pass1_dense(self):
        """The same tables under the seeded specification (what stats() accumulates)."""
        return self._dense("orc_pass1_dense")

    def accumulate_records(self, read_index, allele, AS, XM, XO, seqlen) -> SampleStats:
        nA, nL = self.index.n_alleles, self.index.n_loci
        s = SampleStats(np.zeros(nA, np.int64), np.zeros(nA, np.uint32), np.zeros(nL, np.uint64),
                        np.zeros(nL, np.uint64), np.zeros(MLST_CNT_N, np.uint64))
        ri = np.ascontiguousarray(read_index, np.uint64)
        al = np.ascontiguousarray(allele, np.uint32)
        a, m, o, sl = (np.ascontiguousarray(x, np.int32) for x in (AS, XM, XO, seqlen))
        rc = lib().orc_accumulate_records(self._r, len(ri), _p(ri), _p(al), _p(a), _p(m), _p(o), _p(sl), _p(s.sum_score),
                                          _p(s.n_hits), _p(s.locus_len_sum), _p(s.locus_first), _p(s.counters))
        if rc != 0:
            raise RuntimeError("orc_accumulate_records rc=%d" % rc)
        return s

    def pileup(self, chosen, depth_cap: int = 0, depth_out: dict = None):
        """depth_cap > 0: orc_pileup_capped (the first `depth_cap` records per column in (read index, strand) order);
        depth_out, when given, receives {allele: records that span each column} (uncapped)."""
        b, q, off = self._reads
        ch = np.ascontiguousarray(chosen, np.uint32)
        lens = [int(self.index.off[a + 1] - self.index.off[a]) for a in chosen]
        counts = np.zeros((sum(lens), 4), np.uint32)
        if depth_cap:
            depth = np.zeros(sum(lens), np.uint32)
            rc = lib().orc_pileup_capped(self._r, _p(b), _p(q), _p(off), len(off) - 1, _p(ch), len(ch), int(depth_cap), _p(counts), _p(depth))
            if depth_out is not None:
                at = 0
                for a, L in zip(chosen, lens):
                    depth_out[int(a)] = depth[at:at + L]
                    at += L
        else:
            rc = lib().orc_pileup(self._r, _p(b), _p(q), _p(off), len(off) - 1, _p(ch), len(ch), _p(counts), self.threads)
        if rc != 0:
            raise RuntimeError("orc_pileup rc=%d" % rc)
        out, at = {}, 0
        for a, L in zip(chosen, lens):
            out[int(a)] = counts[at:at + L]
            at += L
        return out

    def align_one(self, bases: bytes, quals: bytes, allele: int, strand: int, diag: int, mode: int = 0):
        b = np.frombuffer(bases, np.uint8)
        q = np.frombuffer(quals, np.uint8)
        out = np.zeros(6, np.int32)
        ci = np.zeros(320, np.int16)
        cj = np.zeros(320, np.int16)
        rc = lib().orc_align_one(self._r, _p(b), _p(q), len(b), allele, strand, diag, mode, _p(out), _p(ci), _p(cj))
        if rc != 0:
            raise RuntimeError("orc_align_one rc=%d" % rc)
        n = int(out[5])
        return dict(score=int(out[0]), xm=int(out[1]), xo=int(out[2]), mm_total=int(out[3]), used_dp=int(out[4]),
                    cols=list(zip(ci[:n].tolist(), cj[:n].tolist())))

    def hamming_all(self, locus: int, query: bytes) -> np.ndarray:
        q = np.frombuffer(query, np.uint8)
        d = np.zeros(int(self.index.locus_count[locus]), np.uint32)
        rc = lib().orc_hamming_all(self._r, locus, _p(q) if len(q) else None, len(q), _p(d))
        if rc != 0:
            raise RuntimeError("orc_hamming_all rc=%d" % rc)
        return d

    def hamming_le(self, locus: int, query: bytes, z: int):
        d = self.hamming_all(locus, query)
        w = np.nonzero(d <= z)[0]
        return (int(self.index.locus_begin[locus]) + int(w[0]) if len(w) else -1), int(len(w))


def string_diff(s1: bytes, s2: bytes) -> int:
    a = np.frombuffer(s1, np.uint8)
    b = np.frombuffer(s2, np.uint8)
    return int(lib().orc_string_diff(_p(a) if len(a) else None, len(a), _p(b) if len(b) else None, len(b)))
