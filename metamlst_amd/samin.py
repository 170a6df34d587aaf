"""Alignment input: a SAM / BAM made by the documented bowtie2 command goes through the same
typing tail as reads aligned on the GPU.

The reference reads `samtools view -h` text (metamlst.py:96-130) and later lets cmseq / pysam pile the same BAM up
(metaMLST_functions.py:255-259).  Here

  * `read_alignments` decodes SAM text (plain or gzip) or BAM (BGZF is multi-member gzip; the record layout is the
    SAM/BAM specification's) into the fields the reference touches -- no samtools, no pysam;
  * `AlignmentSample.add` is metamlst.py:101-130 line for line: RNAME split into species / gene / allele, the score
    taken from the 12th column and "xM" from the 15th BY POSITION (quirk Q1), the species filter, the accept test of
    :115, `cel[...].append(score)`, `sequenceBank[species_gene][QNAME] = len(SEQ)` (a dict: one entry per read name per
    locus, last write wins -- quirk Q3 exactly), the two counters;
  * `AlignmentSample.stats()` hands the result over as the SampleStats the rest of the host logic consumes;
  * `AlignmentSample.pileup(engine, chosen)` piles the records of the chosen contigs up on the GPU
    (mlst_pileup_alignments): CIGAR walk, a base counts when Phred >= minqual, it is A/C/G/T and the record's TRUE
    tags pass AS >= minscore, XM <= max_xM (cmseq's BAM_tagFilter looks tags up by name, unlike :110).

pysam's per-column depth cap (max_depth = 8000) is not applied, as everywhere in this package.
"""
from __future__ import annotations

import gzip
import struct
from dataclasses import dataclass

import numpy as np

from .index import AlleleIndex
from .typing import NO_READ, SampleStats, TypingArgs

CIGAR_OPS = "MIDNSHP=X"
_SEQ16 = "=ACMGRSVTWYHKDBN"


@dataclass
class Alignment:
    """The SAM fields the reference and cmseq touch.  `tags` are the optional fields as text, in file order."""
    qname: str
    flag: int
    rname: str
    pos: int            # 1-based, as in SAM
    cigar: str
    seq: str
    qual: str           # Phred+33 text, or '*'
    tags: list


# ------------------------------------------------------------------ SAM text
def _iter_sam_lines(fh):
    for line in fh:
        if not line or line[0] == "@":
            continue
        f = line.rstrip("\r\n").split("\t")
        if len(f) < 11:
            raise ValueError("malformed SAM record (fewer than 11 columns): %r" % line[:80])
        yield Alignment(f[0], int(f[1]), f[2], int(f[3]), f[5], f[9], f[10], f[11:])


# ------------------------------------------------------------------ BAM
def _aux_text(buf: bytes, at: int, end: int) -> list:
    """Optional fields of one BAM record rendered the way `samtools view` prints them (integers of every width as :i:)."""
    out = []
    while at < end:
        tag = buf[at:at + 2].decode("ascii"); t = chr(buf[at + 2]); at += 3
        if t == "A":
            out.append("%s:A:%s" % (tag, chr(buf[at]))); at += 1
        elif t in "cCsSiI":
            fmt, n = {"c": ("<b", 1), "C": ("<B", 1), "s": ("<h", 2), "S": ("<H", 2), "i": ("<i", 4), "I": ("<I", 4)}[t]
            out.append("%s:i:%d" % (tag, struct.unpack_from(fmt, buf, at)[0])); at += n
        elif t == "f":
            out.append("%s:f:%g" % (tag, struct.unpack_from("<f", buf, at)[0])); at += 4
        elif t in "ZH":
            z = buf.index(b"\0", at)
            out.append("%s:%s:%s" % (tag, t, buf[at:z].decode("ascii"))); at = z + 1
        elif t == "B":
            sub = chr(buf[at]); cnt = struct.unpack_from("<i", buf, at + 1)[0]; at += 5
            fmt, n = {"c": ("b", 1), "C": ("B", 1), "s": ("h", 2), "S": ("H", 2), "i": ("i", 4), "I": ("I", 4), "f": ("f", 4)}[sub]
            vals = struct.unpack_from("<%d%s" % (cnt, fmt), buf, at); at += cnt * n
            out.append("%s:B:%s%s" % (tag, sub, "".join(",%g" % v if sub == "f" else ",%d" % v for v in vals)))
        else:
            raise ValueError("unknown BAM aux type %r" % t)
    return out


def _iter_bam(fh):
    def need(n):
        b = fh.read(n)
        if len(b) != n:
            raise ValueError("truncated BAM")
        return b
    if need(4) != b"BAM\1":
        raise ValueError("not a BAM file")
    need(struct.unpack("<i", need(4))[0])                       # header text
    refs = []
    for _ in range(struct.unpack("<i", need(4))[0]):
        ln = struct.unpack("<i", need(4))[0]
        refs.append(need(ln)[:-1].decode("ascii")); need(4)
    while True:
        head = fh.read(4)
        if not head:
            return
        if len(head) != 4:
            raise ValueError("truncated BAM")
        rec = need(struct.unpack("<i", head)[0])
        ref_id, pos, l_name, _mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", rec, 0)
        at = 32
        qname = rec[at:at + l_name - 1].decode("ascii"); at += l_name
        ops = struct.unpack_from("<%dI" % n_cig, rec, at); at += 4 * n_cig
        cigar = "".join("%d%s" % (o >> 4, CIGAR_OPS[o & 15]) for o in ops) or "*"
        packed = rec[at:at + (l_seq + 1) // 2]; at += (l_seq + 1) // 2
        seq = "".join(_SEQ16[b >> 4] + _SEQ16[b & 15] for b in packed)[:l_seq] or "*"
        q = rec[at:at + l_seq]; at += l_seq
        qual = "*" if (l_seq == 0 or q[0] == 0xFF) else bytes(x + 33 for x in q).decode("ascii")
        yield Alignment(qname, flag, refs[ref_id] if ref_id >= 0 else "*", pos + 1, cigar, seq, qual, _aux_text(rec, at, len(rec)))


def read_alignments(path: str):
    """Iterate the records of a SAM (plain / gzip) or BAM file in file order."""
    with open(path, "rb") as raw:
        magic = raw.read(2)
    if magic == b"\x1f\x8b":
        with gzip.open(path, "rb") as z:
            is_bam = z.read(4) == b"BAM\1"
        if is_bam:
            with gzip.open(path, "rb") as z:
                yield from _iter_bam(z)
        else:
            with gzip.open(path, "rt", newline="") as z:
                yield from _iter_sam_lines(z)
    else:
        with open(path, "r", newline="") as fh:
            yield from _iter_sam_lines(fh)


def parse_cigar(cigar: str) -> list:
    """'5S100M2D45M' -> [len << 4 | op, ...] with the BAM operation codes (MIDNSHP=X = 0..8)."""
    if cigar == "*" or not cigar:
        return []
    out, n = [], 0
    for ch in cigar:
        if ch.isdigit():
            n = n * 10 + ord(ch) - 48
        else:
            out.append((n << 4) | CIGAR_OPS.index(ch)); n = 0
    return out


# ------------------------------------------------------------------ metamlst.py:101-130
class AlignmentSample:
    """One sample's alignments: the accumulation of metamlst.py:101-130 plus what the pileup needs."""

    def __init__(self, index: AlleleIndex, args: TypingArgs | None = None):
        self.index, self.args = index, args or TypingArgs()
        self.label2a = {index.label(a): a for a in range(index.n_alleles)}
        self.cel: dict = {}              # cel[species][gene][allele] = [score, ...]
        self.sequenceBank: dict = {}     # sequenceBank[species_gene][QNAME] = len(SEQ)
        self.first_seen: dict = {}       # species_gene -> index of the first accepted record (dict order of the reference)
        self.totalReads = self.ignoredReads = 0
        self.n_records = 0
        self._rec = []                   # (allele idx, pos0, AS, XM, cigar ops, seq, qual) of records on loaded contigs

    def add(self, al: Alignment):
        a = self.args
        fields = [al.qname, al.flag, al.rname, al.pos, 255, al.cigar, "*", 0, 0, al.seq, al.qual] + list(al.tags)
        species, gene, allele = fields[2].split("_")                     # metamlst.py:106 (ValueError as in the reference)
        score = int(fields[11].split(":")[2])                            # :109
        xM = int(fields[14].split(":")[2])                               # :110, 15th column BY POSITION (Q1)
        sequence = fields[9]
        idx_rec = self.n_records
        self.n_records += 1
        if (a.filter and species in a.filter.split(",")) or not a.filter:   # :114
            if score >= a.minscore and len(sequence) >= a.min_read_len and xM <= a.max_xM:      # :115
                self.cel.setdefault(species, {}).setdefault(gene, {}).setdefault(allele, []).append(score)
                self.sequenceBank.setdefault(species + "_" + gene, {})[fields[0]] = len(sequence)   # :127
                self.first_seen.setdefault(species + "_" + gene, idx_rec)
            else:
                self.ignoredReads += 1                                   # :129
            self.totalReads += 1                                         # :130
        ai = self.label2a.get(al.rname)
        if ai is not None:
            tags = {t.split(":")[0]: t.split(":")[2] for t in al.tags if t.count(":") >= 2}
            self._rec.append((ai, al.pos - 1, int(tags.get("AS", -(1 << 30))), int(tags.get("XM", 1 << 30)),
                              parse_cigar(al.cigar), al.seq, al.qual))

    def add_file(self, path: str):
        for al in read_alignments(path):
            self.add(al)
        return self

    def stats(self) -> SampleStats:
        """`cel` and `sequenceBank` as the exact-integer arrays the typing tail works on."""
        ix = self.index
        s = SampleStats(np.zeros(ix.n_alleles, np.int64), np.zeros(ix.n_alleles, np.uint32), np.zeros(ix.n_loci, np.uint64),
                        np.full(ix.n_loci, NO_READ, np.uint64), np.zeros(8, np.uint64))
        for sp, genes in self.cel.items():
            for g, alleles in genes.items():
                for al, scores in alleles.items():
                    a = self.label2a.get("%s_%s_%s" % (sp, g, al))
                    if a is None:
                        continue                                         # contig that is not in the loaded database
                    s.sum_score[a] = sum(scores); s.n_hits[a] = len(scores)
        for key, names in self.sequenceBank.items():
            sp, g = key.split("_")
            try:
                l = ix.locus_index(sp, g)
            except Exception:
                continue
            s.locus_len_sum[l] = sum(names.values())
            s.locus_first[l] = self.first_seen[key]
        s.counters[0], s.counters[1] = self.totalReads, self.ignoredReads
        return s

    def pileup(self, engine, chosen, minqual: int = 20) -> dict:
        """{allele idx: uint32[len, 4]} for the chosen contigs, counted on the GPU (cmseq get_base_stats restated)."""
        a = self.args
        n = len(self._rec)
        rec_allele = np.fromiter((r[0] for r in self._rec), np.uint32, n)
        rec_pos = np.fromiter((r[1] for r in self._rec), np.int32, n)
        rec_as = np.fromiter((max(-(1 << 30), min(1 << 30, r[2])) for r in self._rec), np.int32, n)
        rec_xm = np.fromiter((max(-(1 << 30), min(1 << 30, r[3])) for r in self._rec), np.int32, n)
        cig_off = np.zeros(n + 1, np.uint64); seq_off = np.zeros(n + 1, np.uint64)
        cig_off[1:] = np.cumsum([len(r[4]) for r in self._rec]); seq_off[1:] = np.cumsum([0 if r[5] == "*" else len(r[5]) for r in self._rec])
        cig = np.fromiter((o for r in self._rec for o in r[4]), np.uint32, int(cig_off[-1]))
        seq = np.frombuffer("".join(r[5] for r in self._rec if r[5] != "*").encode("ascii"), np.uint8)
        qual = np.zeros(int(seq_off[-1]), np.uint8)
        for k, r in enumerate(self._rec):
            if r[5] != "*" and r[6] != "*":
                q = np.frombuffer(r[6].encode("ascii"), np.uint8)
                qual[int(seq_off[k]):int(seq_off[k]) + len(q)] = q - 33
        return engine.pileup_alignments(chosen, rec_allele, rec_pos, rec_as, rec_xm, cig_off, cig, seq_off, seq, qual,
                                        a.minscore, a.max_xM, minqual)
