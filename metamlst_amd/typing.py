"""Host side of the typing run: counterpart of metamlst.py:133-289.

The GPU engine replaces bowtie2 + samtools + pysam + cmseq and hands back exact integer
statistics; everything with floating point or text formatting stays here in Python so that
`round()` and `str(float)` behave exactly as in the reference (SURVEY.md 8a rows a3-a8 and
quirks Q5-Q8, Q13, Q14).  Console colouring (metamlst_print, bcolors) is out of scope.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

from . import db as mdb
from .index import AlleleIndex


@dataclass
class TypingArgs:
    """argparse defaults of metamlst.py:34-49."""
    penalty: int = 100
    minscore: int = 80
    max_xM: int = 5
    min_read_len: int = 50
    min_accuracy: float = 0.90
    nloci: int = 100
    a: bool = False
    quiet: bool = True
    filter: str | None = None
    log: bool = False
    debug: bool = False


@dataclass
class SampleStats:
    """What pass 1 returns (mlst_get_allele_stats): the content of `cel` and `sequenceBank`
    (metamlst.py:116-127) as exact integers."""
    sum_score: np.ndarray      # int64[n_alleles]
    n_hits: np.ndarray         # uint32[n_alleles]
    locus_len_sum: np.ndarray  # uint64[n_loci]
    locus_first: np.ndarray    # uint64[n_loci]
    counters: np.ndarray       # uint64[MLST_CNT_N]


class SeqRecordLite:
    """The three SeqRecord fields buildConsensus fills (metaMLST_functions.py:276)."""
    __slots__ = ("seq", "id", "description", "seqLen", "ci", "sp")

    def __init__(self, seq: str, id: str, description: str, ci: int | None = None, sp: int | None = None):
        self.seq, self.id, self.description, self.seqLen = seq, id, description, None
        self.ci, self.sp = ci, sp                # the two numbers inside `description`, when the maker had them as integers


def _ci_sp(rec) -> tuple[int, int]:
    """(holes, SNPs) of a consensus record: parsed out of 'CI::<n>_SP::<m>' as metamlst.py:254-255 does, unless the record
    carries them already."""
    if rec.ci is not None:
        return rec.ci, rec.sp
    d = rec.description.split("_")
    return int(d[0].split("::")[1]), int(d[1].split("::")[1])


NO_READ = np.uint64(0xFFFFFFFFFFFFFFFF)


def compile_cel(index: AlleleIndex, st: SampleStats, penalty: int) -> dict:
    """metamlst.py:133-151.  cel[species][gene][allele] = (localScore, nHits, round(avg, 1)).
    An allele is present iff it has an accepted record; dict orders follow first appearance in
    the read stream (Q6): species by their earliest locus, genes by locus_first, alleles by number."""
    loci_hit = [l for l in range(index.n_loci) if st.locus_first[l] != NO_READ]
    loci_hit.sort(key=lambda l: (int(st.locus_first[l]), l))
    cel: dict = {}
    for l in loci_hit:
        sp, gene = index.loci[l]
        b, c = int(index.locus_begin[l]), int(index.locus_count[l])
        nh = st.n_hits[b:b + c]
        hit = np.nonzero(nh)[0]
        if hit.size == 0:
            continue
        maxLen = int(nh[hit].max())
        geneInfo = {}
        for k in hit:
            a = b + int(k)
            geneLen = int(nh[k])
            localScore = int(st.sum_score[a])
            if geneLen != maxLen:
                localScore = localScore - (maxLen - geneLen) * penalty
            averageScore = float(localScore) / float(geneLen)
            geneInfo[str(int(index.allele_no[a]))] = (localScore, geneLen, round(averageScore, 1))
        cel.setdefault(sp, {})[gene] = geneInfo
    return cel


def pick_alleles(cel_species: dict, speciesKey: str) -> list[tuple[str, str]]:
    """metamlst.py:244 without the SQL: per gene the label of the allele with the highest
    rounded average, ties resolved by the lowest integer allele number (Q5)."""
    out = []
    for g1, g2 in cel_species.items():
        best = max(avg1 for (_, _, avg1) in g2.values())
        cands = [k for k, (_, _, avg) in g2.items() if avg == best]
        k = sorted(cands, key=lambda x: int(x))[0]
        out.append((g1, k))
    return out


def pick_alleles_fast(index: AlleleIndex, st: SampleStats, penalty: int) -> dict[int, int]:
    """Same decision as compile_cel + pick_alleles for every locus at once: {locus: allele idx}.
    Exactness: Python's round(x, 1) is correctly rounded and monotone, so the winning rounded
    average is round(max x); only alleles with x within 0.11 of the maximum can tie with it, and
    those few are rounded with Python's own round().  Array operations run over all loci together
    (alleles of a locus are contiguous in the index)."""
    if st.n_hits.size == 0:
        return {}
    hl = np.nonzero(st.locus_first != NO_READ)[0]                  # loci with an accepted record
    if hl.size == 0:
        return {}
    if hl.size == index.n_loci:                                    # every locus hit: work on the arrays as they are
        sel = None
        nh = st.n_hits.astype(np.int64)
        ssum = st.sum_score
        begins, lid = index.locus_begin_ip, index.locus_id_ip
    else:                                                          # metagenome against a big database: only the hit loci
        counts = index.locus_count[hl].astype(np.intp)
        begins = np.concatenate(([0], np.cumsum(counts)[:-1])).astype(np.intp)
        lid = np.repeat(np.arange(hl.size, dtype=np.intp), counts)
        sel = np.repeat(index.locus_begin_ip[hl] - begins, counts) + np.arange(int(counts.sum()), dtype=np.intp)
        nh = st.n_hits[sel].astype(np.int64)
        ssum = st.sum_score[sel]
    maxlen = np.maximum.reduceat(nh, begins)                       # per locus: max hits over its alleles
    local = ssum - (maxlen[lid] - nh) * penalty                    # metamlst.py:146-147
    x = local / np.maximum(nh, 1)                                  # same IEEE division as float(a)/float(b); alleles without hits excluded below
    x[nh == 0] = -np.inf
    xmax = np.maximum.reduceat(x, begins)
    near = np.nonzero(x >= xmax[lid] - 0.11)[0]
    near = near[nh[near] > 0]
    best: dict = {}
    for k in near.tolist():
        r = round(float(int(local[k])) / float(int(nh[k])), 1)
        a = k if sel is None else int(sel[k])
        l = int(index.locus_id[a])
        no = int(index.allele_no[a])
        cur = best.get(l)
        if cur is None or r > cur[0] or (r == cur[0] and no < cur[1]):
            best[l] = (r, no, a)
    return {l: v[2] for l, v in best.items()}


def _consensus_bytes(counts: np.ndarray, mincov: int = 1, none_char: str = "N") -> np.ndarray:
    tot = counts.sum(axis=1)
    arg = counts.argmax(axis=1)          # first maximum = alphabetical order of "ACGT"
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)[arg]
    return np.where(tot >= mincov, letters, ord(none_char)).astype(np.uint8)


def consensus_from_counts(counts: np.ndarray, mincov: int = 1, none_char: str = "N") -> list[str]:
    """cmseq reference_free_consensus over get_base_stats [NOT IN TREE]: a column with fewer
    than mincov counted bases is none_char, else the majority base; ties resolve in the order
    A < C < G < T (policy MLST_TIE_ORDER).  dominant_frq_thrsh=0.4 has no effect on the string."""
    return list(_consensus_bytes(counts, mincov, none_char).tobytes().decode())


def build_consensus(chromosomeList: dict, counts_by_label: dict, mincov: int = 1) -> list[SeqRecordLite]:
    """buildConsensus (metaMLST_functions.py:249-281) with the cmseq call replaced by the
    engine's pileup counts.  Gap-fill (:265-273): 'N' -> lower-cased database base (CI += 1); a
    consensus base differing from the database base counts as a SNP (Q11).  The per-position loop
    of the reference is done with array operations over all contigs at once;
    build_consensus_loop below is the literal restatement the tests compare it with."""
    labels = list(chromosomeList)
    if not labels:
        return []
    lens = [len(chromosomeList[l]) for l in labels]
    if any(counts_by_label[l].shape[0] != n for l, n in zip(labels, lens)) or min(lens) == 0:
        return build_consensus_loop(chromosomeList, counts_by_label, mincov)
    counts = counts_by_label[labels[0]] if len(labels) == 1 else np.concatenate([counts_by_label[l] for l in labels])
    dbarr = np.frombuffer("".join(chromosomeList[l] for l in labels).encode("latin-1"), dtype=np.uint8)
    cons = _consensus_bytes(counts, mincov)
    isN = cons == ord("N")
    upper = (dbarr >= 65) & (dbarr <= 90)
    out = np.where(isN, np.where(upper, dbarr + 32, dbarr), cons).astype(np.uint8).tobytes().decode("latin-1")
    offs = np.concatenate(([0], np.cumsum(lens)))
    cI = np.add.reduceat(isN.astype(np.int64), offs[:-1])
    sn = np.add.reduceat(((cons != dbarr) & ~isN).astype(np.int64), offs[:-1])
    return [SeqRecordLite(out[int(offs[k]):int(offs[k + 1])], l, "CI::" + str(int(cI[k])) + "_SP::" + str(int(sn[k])))
            for k, l in enumerate(labels)]


def build_consensus_from_letters(chromosomeList: dict, letters_by_label: dict) -> list[SeqRecordLite]:
    """The tail of buildConsensus (metaMLST_functions.py:260-276) when the consensus string itself comes from
    the engine (mlst_consensus = cmseq's reference_free_consensus): gap-fill and SNP count only."""
    labels = list(chromosomeList)
    if not labels:
        return []
    lens = [len(chromosomeList[l]) for l in labels]
    if any(len(letters_by_label[l]) != n for l, n in zip(labels, lens)) or min(lens) == 0:
        out = []
        for l in labels:      # literal loop for odd cases
            rSequen = list(letters_by_label[l].decode("latin-1")); dbSequen = chromosomeList[l]; cIndex = SNPs = 0
            for i, ch in enumerate(rSequen):
                if ch == "N":
                    rSequen[i] = dbSequen[i].lower(); cIndex += 1
                elif rSequen[i] != dbSequen[i]:
                    SNPs += 1
            out.append(SeqRecordLite("".join(rSequen), l, "CI::" + str(cIndex) + "_SP::" + str(SNPs)))
        return out
    cons = np.frombuffer(b"".join(letters_by_label[l] for l in labels), dtype=np.uint8)
    dbarr = np.frombuffer("".join(chromosomeList[l] for l in labels).encode("latin-1"), dtype=np.uint8)
    isN = cons == ord("N")
    upper = (dbarr >= 65) & (dbarr <= 90)
    out = np.where(isN, np.where(upper, dbarr + 32, dbarr), cons).astype(np.uint8).tobytes().decode("latin-1")
    offs = np.concatenate(([0], np.cumsum(lens)))
    cI = np.add.reduceat(isN.astype(np.int64), offs[:-1])
    sn = np.add.reduceat(((cons != dbarr) & ~isN).astype(np.int64), offs[:-1])
    return [SeqRecordLite(out[int(offs[k]):int(offs[k + 1])], l, "CI::" + str(int(cI[k])) + "_SP::" + str(int(sn[k])))
            for k, l in enumerate(labels)]


def build_consensus_from_letters_many(jobs: list[tuple[dict, dict]]) -> list[list[SeqRecordLite]]:
    """build_consensus_from_letters for every species of a sample in ONE pass over the concatenated loci (a sample of 20
    species is 20 x ~15 numpy calls on 3 KB arrays otherwise: call overhead, 0.6 of the 1.3 ms of a typing step's host
    tail).  Same records, same order; species with an odd case (lengths that differ, an empty locus) go the single way."""
    out: list = [None] * len(jobs)
    flat = []                                    # (job, label) of the regular ones
    for k, (chromosomeList, by_label) in enumerate(jobs):
        labels = list(chromosomeList)
        if not labels:
            out[k] = []
        elif any(len(by_label[l]) != len(chromosomeList[l]) or len(chromosomeList[l]) == 0 for l in labels):
            out[k] = build_consensus_from_letters(chromosomeList, by_label)
        else:
            out[k] = []
            flat.extend((k, l) for l in labels)
    if not flat:
        return out
    lens = [len(jobs[k][0][l]) for k, l in flat]
    cons = np.frombuffer(b"".join(jobs[k][1][l] for k, l in flat), dtype=np.uint8)
    dbarr = np.frombuffer("".join(jobs[k][0][l] for k, l in flat).encode("latin-1"), dtype=np.uint8)
    isN = cons == ord("N")
    upper = (dbarr >= 65) & (dbarr <= 90)
    text = np.where(isN, np.where(upper, dbarr + 32, dbarr), cons).astype(np.uint8).tobytes().decode("latin-1")
    offs = np.concatenate(([0], np.cumsum(lens)))
    cI = np.add.reduceat(isN.astype(np.int64), offs[:-1]).tolist()
    sn = np.add.reduceat(((cons != dbarr) & ~isN).astype(np.int64), offs[:-1]).tolist()
    offs = offs.tolist()
    for i, (k, l) in enumerate(flat):
        out[k].append(SeqRecordLite(text[offs[i]:offs[i + 1]], l, "CI::" + str(cI[i]) + "_SP::" + str(sn[i]), cI[i], sn[i]))
    return out


def build_consensus_loop(chromosomeList: dict, counts_by_label: dict, mincov: int = 1) -> list[SeqRecordLite]:
    """Literal per-position form of metaMLST_functions.py:257-276."""
    seqRec = []
    for chromo, nucleots in chromosomeList.items():
        rSequen = consensus_from_counts(counts_by_label[chromo], mincov)
        dbSequen = chromosomeList[chromo]
        cIndex = 0
        SNPs = 0
        for i, ch in enumerate(rSequen):
            if ch == "N":
                rSequen[i] = dbSequen[i].lower()
                cIndex += 1
            elif rSequen[i] != dbSequen[i]:
                SNPs += 1
        seqRec.append(SeqRecordLite("".join(rSequen), chromo, "CI::" + str(cIndex) + "_SP::" + str(SNPs)))
    return seqRec


@dataclass
class SpeciesResult:
    species: str
    detected: list[str]
    missing: list[str]
    passed_nloci: bool
    closest: dict = field(default_factory=dict)     # gene -> (avg, hits, [allele strs], coverage)
    chosen: list[tuple[str, str]] = field(default_factory=list)   # (label, sequence)
    loci_report: list[dict] = field(default_factory=list)
    written: bool = False
    nfo_line: str | None = None
    newProfile: int = 0


def nfo_line(speciesKey: str, fileName: str, consenSeq: list[SeqRecordLite]) -> str:
    """metamlst.py:285, byte for byte (float formatting quirks such as 98.50999999999999 included)."""
    return (speciesKey + "\t" + fileName + "\t" + "\t".join(
        [recd.id + "::" + str(recd.seq) + "::"
         + str(round(1 - float(_ci_sp(recd)[0]) / float(recd.seqLen), 4) * 100) + "::"
         + str(round(float(_ci_sp(recd)[1]) / float(recd.seqLen), 4) * 100)
         for recd in consenSeq]) + "\r\n")


def sample_name(path: str) -> str:
    """metamlst.py:89: basename up to the first '.' (Q13)."""
    return path.split("/")[-1].split(".")[0]


def _detected_loci(index: AlleleIndex, st: SampleStats) -> dict:
    """{species: {gene: None}} in the same first-appearance order as compile_cel, without the
    per-allele tuples (fast path)."""
    hit = np.nonzero(st.locus_first != NO_READ)[0]
    loci_hit = hit[np.lexsort((hit, st.locus_first[hit]))].tolist()
    out: dict = {}
    for l in loci_hit:
        sp, gene = index.loci[l]
        out.setdefault(sp, {})[gene] = None
    return out


def type_sample(index: AlleleIndex, st: SampleStats, pileup_fn, database: mdb.metaMLST_db, fileName: str,
                args: TypingArgs | None = None, out_dir: str | None = None, fast: bool = False,
                cache: mdb.DbCache | None = None, consensus_fn=None, typed=None) -> list[SpeciesResult]:
    """metamlst.py:133-289 for one sample.

    pileup_fn(list of allele indices) -> {allele idx: uint32[len, 4]} is pass 2 of the engine
    (mlst_pileup).  When out_dir is given the .nfo line is appended to <out_dir>/<fileName>.nfo
    (append mode as metamlst.py:284).  fast=True skips the per-allele `cel` table and the
    closest-allele listing (display only) and picks alleles with pick_alleles_fast; the .nfo
    line is identical (tests/test_typing_host.py)."""
    args = args or TypingArgs()
    cursor = database.cursor
    cel = _detected_loci(index, st) if fast else compile_cel(index, st, args.penalty)
    # typed = (chosen {locus: allele idx}, letters {allele idx: bytes}) from Engine.typing_fetch: choice and consensus
    # were made on the device (mlst_typing_enqueue, penalty = args.penalty); implies the fast path
    fast_choice = (typed[0] if typed is not None else pick_alleles_fast(index, st, args.penalty)) if fast else None
    results: list[SpeciesResult] = []
    plan = []
    for speciesKey, species in cel.items():
        # metamlst.py:184-206 locus presence gate
        tVar = (dict.fromkeys(cache.genes(speciesKey), 0) if cache is not None else
                dict([(row["geneName"], 0) for row in cursor.execute("SELECT geneName FROM genes WHERE bacterium = ?", (speciesKey,))]))
        if len(tVar) < len(species.keys()):
            raise SystemExit("Database is broken for " + speciesKey)      # metamlst.py:188-190 exits
        for sk in species.keys():
            tVar[sk] = 1
        vals = sum(tVar.values())
        res = SpeciesResult(speciesKey,
                            detected=[sk for (sk, v) in sorted(tVar.items()) if v == 1],
                            missing=[sk for (sk, v) in sorted(tVar.items()) if v == 0],
                            passed_nloci=int((float(vals) / float(len(tVar))) * 100) >= args.nloci)
        results.append(res)
        if not res.passed_nloci:
            continue
        if fast:
            for g1 in species.keys():
                a = fast_choice[index.locus_index(speciesKey, g1)]
                res.chosen.append((index.label(a), index.sequence(a)))
                plan.append((res, a))
            continue
        # metamlst.py:213-230 closest alleles + coverage
        for geneKey, geneInfo in sorted(species.items(), key=lambda x: x[0]):
            minValue = max([avg for (val, leng, avg) in geneInfo.values()])
            aElements = {k: v for k, v in geneInfo.items() if v[2] == minValue}
            l = index.locus_index(speciesKey, geneKey)
            genL = int(index.locus_maxlen[l])
            coverage = int(st.locus_len_sum[l])
            res.closest[geneKey] = (minValue, list(aElements.values())[0][1],
                                    sorted(aElements.keys(), key=lambda x: int(x)),
                                    round(float(coverage) / float(genL), 2))
        # metamlst.py:244 choice
        for g1, k in pick_alleles(species, speciesKey):
            l = index.locus_index(speciesKey, g1)
            b = int(index.locus_begin[l])
            a = b + int(np.nonzero(index.allele_no[b:b + int(index.locus_count[l])] == int(k))[0][0])
            res.chosen.append((speciesKey + "_" + g1 + "_" + k, index.sequence(a)))
            plan.append((res, a))
    # pass 2 once for every species that passed (identical to one buildConsensus per species)
    # consensus_fn(list of allele indices) -> {allele idx: consensus bytes} (mlst_consensus) replaces counts + majority
    if typed is not None and not fast:
        # the listing above is the host's (metamlst.py:213-230 prints every allele's figures); choice and consensus came from
        # the device (mlst_typing_enqueue).  The two statements of metamlst.py:244 are tested equal; should they ever differ,
        # that is an error to be seen, not to be papered over with the host's pile-up
        for _, a in plan:
            if a not in typed[1]:
                raise RuntimeError("device-side allele choice differs from metamlst.py:244 for %s" % index.label(a))
    letters = (typed[1] if typed is not None else
               consensus_fn([a for _, a in plan]) if (plan and consensus_fn is not None) else None)
    counts = pileup_fn([a for _, a in plan]) if (plan and letters is None) else {}
    # gap-fill and SNP counts (the tail of buildConsensus) for all species at once when the letters come from the engine
    passed = [res for res in results if res.passed_nloci]
    by_res = {id(res): {} for res in passed}
    for (r2, a) in plan:
        by_res[id(r2)][index.label(a)] = counts[a] if letters is None else letters[a]
    pre = (dict(zip((id(r) for r in passed), build_consensus_from_letters_many([(dict(r.chosen), by_res[id(r)]) for r in passed])))
           if letters is not None else {})
    for res in results:
        if not res.passed_nloci:
            continue
        chromosomeList = dict(res.chosen)
        by_label = by_res[id(res)]
        consenSeq = build_consensus(chromosomeList, by_label, mincov=1) if letters is None else pre[id(res)]
        finWrite = 1
        for l in sorted(consenSeq, key=lambda x: x.id):
            ci, snps = _ci_sp(l)
            holes = str(ci)
            leng = str(len(l.seq))
            leng_ns = str(round(1 - float(holes) / float(leng), 4) * 100) + " %"
            l.seqLen = len(l.seq)
            if (1 - float(holes) / float(leng)) <= args.min_accuracy:      # metamlst.py:262 (Q7: <=)
                finWrite = 0
            if snps > 0:
                seqFind = (cache.sequenceFind(res.species, l.seq) if cache else
                           mdb.sequenceFind(database.conn, res.species, l.seq))   # mixed case (Q8)
                if seqFind:
                    newAllele = seqFind
                else:
                    newAllele = "NEW"
                    res.newProfile = 1
            else:
                newAllele = "--"
                if not args.a:
                    l.seq = ""
            res.loci_report.append(dict(locus=l.id.split("_")[1], ref=l.id.split("_")[2], length=leng, ns=holes,
                                        snps=snps, confidence=leng_ns, notes=newAllele))
        if finWrite:
            res.written = True
            res.nfo_line = nfo_line(res.species, fileName, consenSeq)
            if out_dir is not None:
                if not os.path.isdir(out_dir):
                    os.mkdir(out_dir)
                with open(out_dir + "/" + fileName + ".nfo", "a", newline="") as profil:
                    profil.write(res.nfo_line)
    return results


def log_table(index: AlleleIndex, st: SampleStats, args: TypingArgs, sample_path: str) -> str:
    """The --log table of metamlst.py:159-172 (a cheap per-allele parity probe)."""
    cel = compile_cel(index, st, args.penalty)
    out = ["SAMPLE:\t\t\t\t\t" + sample_path + "\r\n", "VERSION:\t\t\t\t\t1.1\r\n",
           "PENALTY:\t\t\t\t" + repr(args.penalty) + "\r\n", "MIN-THRESHOLD SCORE:\t\t\t\t" + repr(args.minscore) + "\r\n",
           "TOTAL ALIGNED READS:\t\t\t\t" + repr(int(st.counters[0])) + "\r\n",
           " - OF WHICH IGNORED:\t\t\t\t" + repr(int(st.counters[1])) + " BAM READS\r\n\r\n"
           "------------------------------  RESULTS ------------------------------\r\n"]
    for speciesKey, species in cel.items():
        for geneKey, geneInfo in species.items():
            for geneInfoKey, (score, geneLen, average) in sorted(geneInfo.items(), key=lambda x: x[1]):
                out.append("\t".join(map(str, [speciesKey, geneKey, geneInfoKey, score, geneLen, average])) + "\r\n")
    return "".join(out)
