"""SQLite allele table -> flat arrays for mlst_load_reference.

Counterpart of dump_db_to_fasta (metaMLST_functions.py:149-161: every allele with
sequence <> '', contig name bacterium_gene_alleleVariant) feeding bowtie2-build
(metamlst-index.py:222-247).  Here the "index" is built on the GPU side by
mlst_load_reference; this module only orders the alleles so that each locus is contiguous
and keeps the maps back to (species, gene, alleleVariant).
"""
from __future__ import annotations

import os
import sqlite3
from dataclasses import dataclass

import numpy as np


@dataclass
class AlleleIndex:
    species: list[str]                      # species_id -> organism key
    loci: list[tuple[str, str]]             # locus id -> (species, gene)
    locus_id: np.ndarray                    # uint32[n_alleles]
    species_id: np.ndarray                  # uint32[n_alleles]
    allele_no: np.ndarray                   # int32[n_alleles]  (alleles.alleleVariant)
    rec_id: np.ndarray                      # int64[n_alleles]  (alleles.recID)
    off: np.ndarray                         # uint64[n_alleles+1]
    ascii_concat: np.ndarray                # uint8
    locus_begin: np.ndarray                 # int64[n_loci]
    locus_count: np.ndarray                 # int64[n_loci]
    locus_species: np.ndarray               # uint32[n_loci]
    locus_maxlen: np.ndarray                # int64[n_loci]  (SELECT LENGTH(sequence) ... ORDER BY L DESC LIMIT 1, metamlst.py:225)

    @property
    def n_alleles(self) -> int:
        return int(self.locus_id.shape[0])

    @property
    def n_loci(self) -> int:
        return len(self.loci)

    def sequence(self, a: int) -> str:
        s = self._seq_cache.get(a)
        if s is None:
            s = self.ascii_concat[int(self.off[a]):int(self.off[a + 1])].tobytes().decode()
            if len(self._seq_cache) < 65536:     # the alleles a run keeps choosing (a typing step asks for ~140 of them, again and again)
                self._seq_cache[a] = s
        return s

    def label(self, a: int) -> str:
        s = self._label_cache.get(a)
        if s is None:
            sp, gene = self.loci[int(self.locus_id[a])]
            s = "%s_%s_%d" % (sp, gene, int(self.allele_no[a]))
            if len(self._label_cache) < 65536:
                self._label_cache[a] = s
        return s

    def locus_index(self, species: str, gene: str) -> int:
        return self._locus_map[(species, gene)]

    def __post_init__(self):
        self._locus_map = {k: i for i, k in enumerate(self.loci)}
        self._seq_cache: dict = {}
        self._label_cache: dict = {}
        self.locus_id_ip = self.locus_id.astype(np.intp)       # fancy-index form (uint32 indices are converted on every use)
        self.locus_begin_ip = self.locus_begin.astype(np.intp)


def similarity_order(seqs: list[bytes]) -> list[int]:
    """Order the alleles of one locus so that neighbours are similar.  The engine extends a read against the alleles
    of a locus 64 at a time (one per lane); lanes with similar mismatch patterns diverge less.  Purely a layout
    choice: every output is keyed by alleleVariant, not by position in the index.

    Sub-quadratic (real PubMLST loci hold > 10 k alleles): the alleles are compared with the column-wise majority
    sequence, the variant columns are ranked by how many alleles differ there, and the alleles are sorted
    lexicographically on their differences read in that rank order -- alleles that share the locus' common variants
    (a clade) end up adjacent, then split by the next most common variant, and so on: O(n L) to build the keys and
    one sort.  Ties (identical difference patterns cannot occur for distinct equal-length alleles; alleles of other
    lengths differ in the tail) keep the input order."""
    n = len(seqs)
    if n <= 2:
        return list(range(n))
    L = max(len(s) for s in seqs)
    mat = np.full((n, L), ord("-"), np.uint8)
    for i, s in enumerate(seqs):
        mat[i, :len(s)] = np.frombuffer(s, np.uint8)
    mat &= 0xDF                                    # upper case ('-' becomes 0x0D: still a distinct symbol)
    # majority symbol per column among the five that matter; everything else counts as a difference
    cnt = np.stack([(mat == c).sum(axis=0) for c in b"ACGT"])
    major = np.frombuffer(b"ACGT", np.uint8)[cnt.argmax(axis=0)]
    diff = mat != major[None, :]
    freq = diff.sum(axis=0)
    cols = np.argsort(-freq, kind="stable")
    cols = cols[freq[cols] > 0]
    if len(cols) == 0:
        return list(range(n))
    # key = the differing symbol (0 where the allele has the majority base) per ranked column: not just "differs",
    # so that the alleles of a multi-allelic site split by base
    key = np.where(diff[:, cols], mat[:, cols], 0).astype(np.uint8)
    rows = np.ascontiguousarray(key).view(np.dtype((np.void, key.shape[1]))).ravel()
    return [int(i) for i in np.argsort(rows, kind="stable")]


_CACHE_VERSION = 1


def _db_fingerprint(db_path: str) -> str:
    """What names one state of the database file: size, mtime, SQLite's own change counter (header bytes 24-27: incremented by
    every write transaction) and a hash of the file's first and last 64 KB.  (A hash of the whole file would cost more than the
    cache saves on a 200 MB database.)"""
    import hashlib
    st = os.stat(db_path)
    h = hashlib.blake2b(digest_size=16)
    with open(db_path, "rb") as f:
        head = f.read(65536)
        h.update(head)
        if st.st_size > 131072:
            f.seek(st.st_size - 65536)
            h.update(f.read(65536))
    return "%d-%d-%s-%s-v%d" % (st.st_size, st.st_mtime_ns, head[24:28].hex(), h.hexdigest(), _CACHE_VERSION)


def _cache_path(db_path: str) -> str:
    return db_path + ".mlstidx"


_ARRAYS = ("locus_id", "species_id", "allele_no", "rec_id", "off", "ascii_concat", "locus_begin", "locus_count", "locus_species", "locus_maxlen")


def _load_cached(db_path: str):
    """One file: a JSON header line (fingerprint, names, where every array sits), then the arrays as they are in memory,
    mapped rather than read (the 150 MB of allele text is paged in when the engine walks it)."""
    import json
    try:
        path = _cache_path(db_path)
        if not os.path.exists(path):
            return None
        with open(path, "rb") as f:
            head = json.loads(f.readline().decode())
        if head.get("fingerprint") != _db_fingerprint(db_path):
            return None
        arr = {}
        for name, (dtype, n, offset) in head["arrays"].items():
            arr[name] = np.memmap(path, dtype=np.dtype(dtype), mode="r", offset=int(offset), shape=(int(n),)) if int(n) else np.zeros(0, np.dtype(dtype))
        species = head["species"]
        loci = [(species[int(a)], g) for a, g in head["loci"]]
        return AlleleIndex(species, loci, *[arr[k] for k in _ARRAYS])
    except Exception:      # noqa: BLE001 -- an unreadable or foreign cache file is no cache
        return None


def _store_cached(db_path: str, ix: AlleleIndex) -> None:
    """Next to the database, as `<idx>.1.bt2` sits next to the FASTA dump (metamlst-index.py:224-225); silently skipped where
    the directory cannot be written."""
    import json
    tmp = None
    try:
        path = _cache_path(db_path)
        tmp = "%s.%d.tmp" % (path, os.getpid())
        sp_of = {s: i for i, s in enumerate(ix.species)}
        arrays = {k: np.ascontiguousarray(getattr(ix, k)) for k in _ARRAYS}
        head = {"fingerprint": _db_fingerprint(db_path), "species": list(ix.species), "loci": [[sp_of[s], g] for s, g in ix.loci], "arrays": {}}
        # offsets depend on the header's length, which holds the offsets: fixed-width numbers, two passes
        for k, a in arrays.items():
            head["arrays"][k] = [a.dtype.str, int(a.size), "%020d" % 0]
        line_len = len(json.dumps(head).encode()) + 1
        at = (line_len + 63) & ~63
        for k, a in arrays.items():
            head["arrays"][k][2] = "%020d" % at
            at = (at + a.nbytes + 63) & ~63
        line = json.dumps(head).encode() + b"\n"
        assert len(line) == line_len
        with open(tmp, "wb") as f:
            f.write(line)
            for k, a in arrays.items():
                f.seek(int(head["arrays"][k][2]))
                a.tofile(f)
        os.replace(tmp, path)
    except Exception:      # noqa: BLE001
        try:
            if tmp:
                os.unlink(tmp)
        except Exception:  # noqa: BLE001
            pass


def load_index(db_path: str, species_filter: list[str] | None = None, cluster: bool = True, cache: bool | None = None) -> AlleleIndex:
    """Read alleles as dump_db_to_fasta does (sequence <> ''), optionally restricted to
    the --filter species (metamlst.py:114 applies the filter per record; not loading the
    other species' alleles is equivalent for every output of the path).

    cache (default: on unless MLST_INDEX_CACHE=0): the ordered arrays are kept in `<database>.mlstidx`, named by the state
    of the database file, the way the reference keeps `<idx>.1.bt2` and skips bowtie2-build when it is there
    (metamlst-index.py:224-225).  A second command on a 315 k-allele database then loads in ~0.1 s instead of walking the
    SQLite table and ordering every locus again (1.3 s)."""
    if cache is None:
        cache = os.environ.get("MLST_INDEX_CACHE", "1") != "0"
    use_cache = cache and cluster and not species_filter and db_path != ":memory:" and os.path.isfile(db_path)
    if use_cache:
        ix = _load_cached(db_path)
        if ix is not None:
            return ix
    ix = _load_index_sql(db_path, species_filter, cluster)
    if use_cache:
        _store_cached(db_path, ix)
    return ix


def _load_index_sql(db_path: str, species_filter: list[str] | None, cluster: bool) -> AlleleIndex:
    conn = sqlite3.connect(db_path)
    q = "SELECT recID,bacterium,gene,alleleVariant,sequence FROM alleles WHERE sequence <> ''"
    rows = conn.execute(q).fetchall()
    conn.close()
    if species_filter:
        keep = set(species_filter)
        rows = [r for r in rows if r[1] in keep]
    # contiguous loci; allele order inside a locus = (alleleVariant, recID), then optionally similarity order
    rows.sort(key=lambda r: (r[1], r[2], int(r[3]), r[0]))
    if cluster and rows:
        out, i = [], 0
        while i < len(rows):
            j = i
            while j < len(rows) and rows[j][1] == rows[i][1] and rows[j][2] == rows[i][2]:
                j += 1
            grp = rows[i:j]
            out += [grp[k] for k in similarity_order([r[4].encode() for r in grp])]
            i = j
        rows = out
    species, loci = [], []
    sp_map, lo_map = {}, {}
    n = len(rows)
    locus_id = np.zeros(n, np.uint32)
    species_id = np.zeros(n, np.uint32)
    allele_no = np.zeros(n, np.int32)
    rec_id = np.zeros(n, np.int64)
    off = np.zeros(n + 1, np.uint64)
    chunks = []
    for i, (rid, sp, gene, av, seq) in enumerate(rows):
        if sp not in sp_map:
            sp_map[sp] = len(species)
            species.append(sp)
        key = (sp, gene)
        if key not in lo_map:
            lo_map[key] = len(loci)
            loci.append(key)
        locus_id[i] = lo_map[key]
        species_id[i] = sp_map[sp]
        allele_no[i] = int(av)
        rec_id[i] = rid
        b = seq.encode()
        chunks.append(b)
        off[i + 1] = off[i] + np.uint64(len(b))
    ascii_concat = np.frombuffer(b"".join(chunks), dtype=np.uint8).copy() if chunks else np.zeros(0, np.uint8)
    nl = len(loci)
    locus_begin = np.zeros(nl, np.int64)
    locus_count = np.zeros(nl, np.int64)
    locus_species = np.zeros(nl, np.uint32)
    locus_maxlen = np.zeros(nl, np.int64)
    lens = (off[1:] - off[:-1]).astype(np.int64)
    for l in range(nl):
        idx = np.nonzero(locus_id == l)[0]
        locus_begin[l] = idx[0]
        locus_count[l] = len(idx)
        locus_species[l] = species_id[idx[0]]
        locus_maxlen[l] = lens[idx].max()
    return AlleleIndex(species, loci, locus_id, species_id, allele_no, rec_id, off, ascii_concat,
                       locus_begin, locus_count, locus_species, locus_maxlen)
