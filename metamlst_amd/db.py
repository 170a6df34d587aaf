"""SQLite access helpers of the MLST hot path.

Host-side counterparts of the thin SQL wrappers in metaMLST_functions.py (same names,
argument meaning and return values) so the typing / merge logic reads like the reference.
Not accelerated: these are a handful of indexed lookups per sample (SURVEY.md section 2,
"DB access helpers").  Schema: metamlst-index.py:62-65.
"""
from __future__ import annotations

import sqlite3


class metaMLST_db:
    """metaMLST_functions.py:428-481 (connection holder + defineProfile method)."""

    def __init__(self, dbPath: str):
        # metaMLST_functions.py:433-440
        self.conn = sqlite3.connect(dbPath)
        self.conn.row_factory = sqlite3.Row
        self.cursor = self.conn.cursor()

    def closeConnection(self) -> None:
        self.conn.close()

    def getGeneNames(self, profile: str) -> list[str]:
        # metaMLST_functions.py:465-469
        return [row["geneName"] for row in self.cursor.execute("SELECT geneName FROM genes WHERE bacterium = ?", (profile,))]

    def defineProfile(self, geneList):
        # metaMLST_functions.py:472-481
        recs = []
        result = None
        for allele in geneList:
            self.cursor.execute("SELECT recID FROM alleles WHERE bacterium||'_'||gene||'_'||alleleVariant = ?", (allele,))
            result = self.cursor.fetchone()
            if result:
                recs.append(str(result["recID"]))
        return _profile_query(self.cursor, recs) if result else [(0, 0)]


def _profile_query(cursor, recs):
    """The profile-matching SQL shared by both defineProfile variants
    (metaMLST_functions.py:216 and :481): profiles sharing the maximum number of the given
    alleles, with the share of alleles matched as an int percentage."""
    inlist = ",".join(recs)
    q = ("SELECT profileCode, COUNT(*) as T FROM profiles WHERE alleleCode IN (" + inlist + ") GROUP BY profileCode "
         "HAVING T = (SELECT COUNT(*)  FROM profiles WHERE alleleCode IN (" + inlist + ") GROUP BY profileCode "
         "ORDER BY COUNT(*) DESC LIMIT 1) ORDER BY T DESC")
    return [(row["profileCode"], int((float(row["T"]) / float(len(recs))) * 100)) for row in cursor.execute(q)]


def defineProfile(conn, geneList):
    """metaMLST_functions.py:205-216 (module-level variant used by metamlst-merge.py:205).
    Quirk Q9: only the LAST label's lookup decides the [(0,0)] fallback."""
    recs = []
    e = None
    result = None
    for allele in geneList:
        e = conn.cursor()
        e.execute("SELECT recID FROM alleles WHERE bacterium||'_'||gene||'_'||alleleVariant = ?", (allele,))
        result = e.fetchone()
        if result:
            recs.append(str(result["recID"]))
    return _profile_query(e, recs) if result else [(0, 0)]


def sequenceExists(conn, bacterium, sequence) -> bool:
    # metaMLST_functions.py:168-172
    e = conn.cursor()
    e.execute("SELECT 1 FROM alleles WHERE sequence = ? AND bacterium = ?", (str(sequence), bacterium))
    return len(e.fetchall()) > 0


def sequenceFind(conn, bacterium, sequence):
    # metaMLST_functions.py:196-203 -- returns the GENE name (display only) or 0
    e = conn.cursor()
    e.execute("SELECT gene,alleleVariant FROM alleles WHERE sequence = ? AND bacterium = ?", (str(sequence), bacterium))
    res = e.fetchone()
    return res["gene"] if res else 0


def sequenceLocate(conn, bacterium, sequence) -> str:
    # metaMLST_functions.py:218-222
    e = conn.cursor()
    e.execute("SELECT alleleVariant FROM alleles WHERE sequence = ? AND bacterium = ?", (str(sequence), bacterium))
    return str(e.fetchone()["alleleVariant"])


def sequencesGetAll(conn, bacterium, gene) -> dict:
    # metaMLST_functions.py:224-228
    e = conn.cursor()
    e.execute("SELECT sequence,alleleVariant FROM alleles WHERE gene = ? AND bacterium = ?", (gene, bacterium))
    return dict((x["alleleVariant"], x["sequence"]) for x in e.fetchall())


def db_getUnalSequence(conn, bacterium, gene, allele):
    # metaMLST_functions.py:186-194
    e = conn.cursor()
    unalobj = e.execute("SELECT sequence FROM alleles WHERE bacterium = ? AND gene = ? AND alleleVariant = ?",
                        (bacterium, gene, allele)).fetchone()
    return unalobj["sequence"] if unalobj is not None else None


def db_getOrganisms(conn, bacterium=None):
    # metaMLST_functions.py:422-426
    e = conn.cursor()
    t = dict((elem["organismkey"], (elem["label"] if elem["label"] is not None else "(" + elem["organismkey"] + ")"))
             for elem in e.execute("SELECT label,organismkey,COUNT(DISTINCT profileCode) AS totalProfiles FROM organisms,profiles "
                                   "WHERE organismkey = bacterium GROUP BY label,organismkey"))
    return t[bacterium] if bacterium else t


def stringDiff(s1, s2) -> int:
    """metaMLST_functions.py:230-234, restated for host-side checks of short strings.
    The allele-match scan itself runs on the GPU (mlst_hamming_le)."""
    c = 0
    for a, b in zip(s1, s2):
        if a != b:
            c += 1
    return c


class DbCache:
    """In-memory lookups with the same answers as the SQL helpers above.  The reference issues un-indexed full-table scans
    per locus per sample (sequenceExists / sequenceLocate / defineProfile); with the alignment on the GPU those scans would
    dominate a typing pass.  Equivalence with the SQL versions is tested in tests/test_golden_functions.py.  Rows are
    visited in rowid order, which is the order SQLite returns them for these un-ordered queries, so `fetchone()` semantics
    (first row wins) are kept.

    Nothing is built before it is asked for (round 5; until round 4 the constructor walked every allele row of the
    database -- 0.7 s of every `cli type` command on 315 k alleles, the whole prologue of the folder mode): the sequence
    table of ONE species when a consensus of that species is looked up (from `index`, the AlleleIndex the engine was loaded
    from, when it holds the species: no SQL at all), the label and profile tables when an ST is first called."""

    def __init__(self, conn, index=None):
        self.conn = conn
        self.index = index
        self._seq_sp: dict = {}         # bacterium -> {sequence: (gene, alleleVariant) of the first row}
        self._seq_all = None            # {(bacterium, sequence): ...} of every row (species the index does not hold)
        self._label_rec = None          # 'bacterium_gene_alleleVariant' -> recID of the first row
        self._prof_by_allele = None     # alleleCode -> [profileCode, ...] (one per profiles row)
        self._genes: dict = {}          # bacterium -> [geneName, ...]

    # ---- tables, built on first use
    def _seqs_of(self, bacterium) -> dict:
        d = self._seq_sp.get(bacterium)
        if d is not None:
            return d
        idx = self.index
        if idx is not None and bacterium in getattr(idx, "species", ()) and not getattr(idx, "filtered", False):
            import numpy as np
            sid = idx.species.index(bacterium)
            al = np.nonzero(idx.species_id == sid)[0]
            al = al[np.argsort(idx.rec_id[al], kind="stable")]          # rowid order: the first row of a sequence wins
            d = {}
            for a in al:
                a = int(a)
                d.setdefault(idx.ascii_concat[int(idx.off[a]):int(idx.off[a + 1])].tobytes().decode(), (idx.loci[int(idx.locus_id[a])][1], str(int(idx.allele_no[a]))))
            # (rows with an empty sequence are not in the index, and alleleVariant comes back from SQLite as it is stored:
            # both only matter for the look-ups below that fall through to SQL)
            self._seq_sp[bacterium] = d
            return d
        if self._seq_all is None:
            self._seq_all = {}
            for row in self.conn.execute("SELECT bacterium,gene,sequence,alleleVariant FROM alleles ORDER BY recID"):
                self._seq_all.setdefault((row["bacterium"], row["sequence"]), (row["gene"], row["alleleVariant"]))
        d = {seq: v for (sp, seq), v in self._seq_all.items() if sp == bacterium}
        self._seq_sp[bacterium] = d
        return d

    def _first(self, bacterium, sequence):
        sequence = str(sequence)
        if sequence == "":              # (the index holds no empty sequences: the reference's own query answers)
            row = self.conn.execute("SELECT gene,alleleVariant FROM alleles WHERE bacterium = ? AND sequence = ?", (bacterium, sequence)).fetchone()
            return (row["gene"], row["alleleVariant"]) if row else None
        return self._seqs_of(bacterium).get(sequence)

    @property
    def label_rec(self) -> dict:
        if self._label_rec is None:
            self._label_rec = {}
            for row in self.conn.execute("SELECT recID,bacterium,gene,alleleVariant FROM alleles ORDER BY recID"):
                self._label_rec.setdefault("%s_%s_%s" % (row["bacterium"], row["gene"], row["alleleVariant"]), row["recID"])
        return self._label_rec

    @property
    def prof_by_allele(self) -> dict:
        if self._prof_by_allele is None:
            self._prof_by_allele = {}
            for row in self.conn.execute("SELECT profileCode,alleleCode FROM profiles ORDER BY recID"):
                self._prof_by_allele.setdefault(row["alleleCode"], []).append(row["profileCode"])
        return self._prof_by_allele

    def genes(self, bacterium) -> list:
        """geneName rows of `SELECT geneName FROM genes WHERE bacterium = ?` (metamlst.py:185), in rowid order."""
        g = self._genes.get(bacterium)
        if g is None:
            g = [row["geneName"] for row in self.conn.execute("SELECT geneName FROM genes WHERE bacterium = ?", (bacterium,))]
            self._genes[bacterium] = g
        return g

    def sequenceExists(self, bacterium, sequence) -> bool:
        return self._first(bacterium, sequence) is not None

    def sequenceFind(self, bacterium, sequence):
        r = self._first(bacterium, sequence)
        return r[0] if r else 0

    def sequenceLocate(self, bacterium, sequence) -> str:
        return str(self._first(bacterium, sequence)[1])

    def defineProfile(self, geneList):
        """Module-level defineProfile (metaMLST_functions.py:205-216) including Q9: the LAST label
        decides the [(0,0)] fallback.  Profiles tied on the match count come out in ascending
        profileCode order (SQLite's GROUP BY order for this query)."""
        recs = []
        result = None
        for allele in geneList:
            result = self.label_rec.get(allele)
            if result is not None:
                recs.append(result)
        if result is None:
            return [(0, 0)]
        count: dict = {}
        for r in set(recs):            # SQL `IN (...)` ignores duplicate recIDs
            for pc in self.prof_by_allele.get(r, ()):
                count[pc] = count.get(pc, 0) + 1
        if not count:
            return []
        top = max(count.values())
        return [(pc, int((float(top) / float(len(recs))) * 100)) for pc in sorted(count) if count[pc] == top]
