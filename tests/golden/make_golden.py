#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own Python.

Run in the authoring container only (needs /root/reference, which never travels):

    python tests/golden/make_golden.py

Part A imports /root/reference/metaMLST_functions.py (with the stubs of tests/golden/stubs for
pysam / Bio / cmseq) and records inputs -> outputs of stringDiff, the SQL helpers, both
defineProfile variants and buildConsensus' gap-fill tail.
Part B runs /root/reference/metamlst.py as a script on synthetic SAM records (a fake `samtools`
on PATH pipes the SAM text through) and keeps the .nfo / --log files it writes.
Part C runs /root/reference/metamlst-merge.py on a folder of .nfo lines and keeps merged/*.txt.
The vectors are data: inputs and the reference's outputs.  No reference source is copied.
"""
from __future__ import annotations

import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
STUBS = os.path.join(HERE, "stubs")
sys.path.insert(0, ROOT)

from metamlst_amd import synth  # noqa: E402


# ------------------------------------------------------------------ database shared by every case
def build_db(path):
    loci = {"spA": [("g1", 120), ("g2", 150), ("g3", 100)], "spB": [("h1", 110), ("h2", 90)]}
    db = synth.make_db(path, loci, alleles_per_locus=12, n_profiles=15, seed=4242, max_div=0.10)
    conn = sqlite3.connect(path)
    with open(os.path.join(HERE, "db.sql"), "w") as f:
        for line in conn.iterdump():
            if "sqlite_sequence" in line:
                continue
            f.write(line + "\n")
    conn.close()
    return db


def seq_of(dbpath, sp, gene, allele):
    return synth.allele_sequence(dbpath, sp, gene, allele)


# ------------------------------------------------------------------ part A: importable functions
def part_a(dbpath):
    sys.path.insert(0, STUBS)
    sys.path.insert(0, REF)
    import metaMLST_functions as F  # the reference module itself
    out = {}
    rng = np.random.default_rng(1)
    cases = [("", ""), ("A", ""), ("ACGT", "ACGT"), ("ACGT", "ACGA"), ("ACGTAC", "ACG"), ("acgt", "ACGT"), ("NNNN", "ACGT")]
    for _ in range(40):
        n1, n2 = int(rng.integers(0, 60)), int(rng.integers(0, 60))
        cases.append(("".join(rng.choice(list("ACGTN"), n1)), "".join(rng.choice(list("ACGTN"), n2))))
    out["stringDiff"] = [[a, b, F.stringDiff(a, b)] for a, b in cases]

    conn = sqlite3.connect(dbpath)
    conn.row_factory = sqlite3.Row
    mdb = F.metaMLST_db(dbpath)
    s_known = seq_of(dbpath, "spA", "g2", 5)
    s_mut = s_known[:10] + ("A" if s_known[10] != "A" else "C") + s_known[11:]
    q = []
    for sp, s in [("spA", s_known), ("spB", s_known), ("spA", s_mut), ("spA", s_known.lower()), ("spB", seq_of(dbpath, "spB", "h1", 2))]:
        ex = F.sequenceExists(conn, sp, s)
        q.append({"species": sp, "seq": s, "exists": ex, "find": F.sequenceFind(conn, sp, s),
                  "locate": F.sequenceLocate(conn, sp, s) if ex else None})
    out["sequence_queries"] = q
    out["sequencesGetAll"] = {"spA|g1": {str(k): v for k, v in F.sequencesGetAll(conn, "spA", "g1").items()}}
    out["db_getUnalSequence"] = [["spA", "g3", "7", F.db_getUnalSequence(conn, "spA", "g3", "7")]]
    out["getGeneNames"] = {"spA": mdb.getGeneNames("spA"), "spB": mdb.getGeneNames("spB")}
    # defineProfile: exact profile, partial, unknown label last / first (Q9), duplicates
    prof = {sp: conn.execute("SELECT profileCode,gene,alleleVariant FROM profiles,alleles WHERE alleleCode=alleles.recID AND alleles.bacterium=? ORDER BY profileCode,gene", (sp,)).fetchall() for sp in ("spA", "spB")}
    p3 = ["spA_%s_%d" % (r["gene"], r["alleleVariant"]) for r in prof["spA"] if r["profileCode"] == 3]
    p7 = ["spA_%s_%d" % (r["gene"], r["alleleVariant"]) for r in prof["spA"] if r["profileCode"] == 7]
    lists = [p3, p7, p3[:2] + [p7[2]], p3[:2] + ["spA_g3_999"], ["spA_g1_999"] + p3[1:], [p3[0], p3[0], p3[1]], ["spA_g9_1"], p3[::-1]]
    dp = []
    for L in lists:
        dp.append({"labels": L, "module": [list(t) for t in F.defineProfile(conn, L)], "method": [list(t) for t in mdb.defineProfile(L)]})
    out["defineProfile"] = dp
    # buildConsensus tail with the stub cmseq (consensus supplied through the counts file)
    bc = []
    tmp = tempfile.mkdtemp()
    for k in range(6):
        label = "spA_g3_%d" % (k + 1)
        dbs = seq_of(dbpath, "spA", "g3", k + 1)
        counts = np.zeros((len(dbs), 4), int)
        for i, ch in enumerate(dbs):
            counts[i, "ACGT".index(ch)] = int(rng.integers(1, 9))
        holes = rng.choice(len(dbs), size=int(rng.integers(0, 15)), replace=False)
        counts[holes] = 0
        for i in rng.choice(len(dbs), size=int(rng.integers(0, 4)), replace=False):
            counts[i] = 0
            counts[i, ("ACGT".index(dbs[i]) + 1) % 4] = 3
        tie = int(rng.integers(len(dbs)))
        counts[tie] = [2, 2, 0, 2]
        bam = os.path.join(tmp, "x%d.bam" % k)
        json.dump({"spA_g3": counts.tolist()}, open(bam + ".counts.json", "w"))
        recs = F.buildConsensus(bam, {label: dbs}, 80, 5, False)
        bc.append({"label": label, "db": dbs, "counts": counts.tolist(), "seq": str(recs[0].seq), "id": recs[0].id,
                   "description": recs[0].description})
    out["buildConsensus"] = bc
    out["cmseq_call"] = json.load(open(bam + ".cmseq_calls.json"))[0]["kwargs"]
    json.dump(out, open(os.path.join(HERE, "functions.json"), "w"), indent=1)
    conn.close()
    sys.path.remove(STUBS)
    sys.path.remove(REF)


# ------------------------------------------------------------------ part B: metamlst.py on synthetic SAM
def sam_line(qname, rname, seqlen, AS, XM, XO, has_xs, flag=0):
    tags = ["AS:i:%d" % AS] + (["XS:i:%d" % max(0, AS - 3)] if has_xs else []) + ["XN:i:0", "XM:i:%d" % XM, "XO:i:%d" % XO,
                                                                                "XG:i:%d" % XO, "NM:i:%d" % (XM + XO), "YT:Z:UU"]
    return "\t".join([qname, str(flag), rname, "1", "255", "%dM" % seqlen, "*", "0", "0", "A" * seqlen, "*"] + tags)


def gen_sam(rng, dbpath, spec):
    """spec: list of (species, gene, true_allele, n_reads, alt_allele or None).  Every read gets a record on
    every allele of the locus (as bowtie2 -a does); AS falls with a made-up distance to the true allele."""
    lines = ["@HD\tVN:1.0\tSO:unsorted"]
    rid = 0
    order = []
    for sp, gene, true, n_reads, twin in spec:
        order += [(sp, gene, true, twin)] * n_reads
    perm = rng.permutation(len(order))
    for p in perm:
        sp, gene, true, twin = order[p]
        rid += 1
        L = int(rng.choice([150, 150, 150, 120, 75, 49, 40]))
        single = rng.random() < 0.08
        alleles = [true] if single else list(range(1, 13))
        for a in alleles:
            dist = 0 if a == true or a == twin else 1 + (a * 7 + true) % 6
            xm = dist + int(rng.integers(0, 2)) * (rng.random() < 0.2)
            xo = 0
            if single:   # no XS:i -> column 15 is XO (quirk Q1): make both outcomes appear
                xm, xo = (7, 0) if rid % 2 else (0, 6)
            AS = max(20, 2 * L - 12 * dist - int(rng.integers(0, 6)) - 8 * xo)
            lines.append(sam_line("r%d" % rid, "%s_%s_%d" % (sp, gene, a), L, AS, xm, xo, has_xs=not single, flag=0 if a == alleles[0] else 256))
    return "\n".join(lines) + "\n"


def counts_for(rng, dbpath, sp, gene, allele, n_holes, n_snps, ties=1):
    dbs = seq_of(dbpath, sp, gene, allele)
    counts = np.zeros((len(dbs), 4), int)
    for i, ch in enumerate(dbs):
        counts[i, "ACGT".index(ch)] = int(rng.integers(2, 30))
        counts[i, ("ACGT".index(ch) + 2) % 4] = int(rng.integers(0, 2))
    pos = rng.permutation(len(dbs))
    counts[pos[:n_holes]] = 0
    for i in pos[n_holes:n_holes + n_snps]:
        b = "ACGT".index(dbs[i])
        counts[i] = 0
        counts[i, (b + 1) % 4] = 9
        counts[i, b] = 3
    for i in pos[n_holes + n_snps:n_holes + n_snps + ties]:
        b = "ACGT".index(dbs[i])
        counts[i] = 0
        counts[i, b] = 4
        counts[i, (b + 1) % 4] = 4       # a tie: alphabetical order decides
    return counts.tolist()


def run_metamlst(case_dir, dbpath, sam_text, counts, extra_args):
    work = tempfile.mkdtemp()
    bam = os.path.join(work, "sampleX.fake.bam")
    open(bam, "w").write(sam_text)
    json.dump(counts, open(bam + ".counts.json", "w"))
    env = dict(os.environ)
    env["PATH"] = STUBS + os.pathsep + env["PATH"]
    env["PYTHONPATH"] = STUBS + os.pathsep + REF
    out = os.path.join(work, "out")
    cmd = [sys.executable, os.path.join(REF, "metamlst.py"), bam, "-o", out, "-d", dbpath] + extra_args
    r = subprocess.run(cmd, env=env, cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        raise RuntimeError("reference metamlst.py failed: " + r.stderr.decode()[-2000:])
    os.makedirs(case_dir, exist_ok=True)
    open(os.path.join(case_dir, "input.sam"), "w").write(sam_text)
    json.dump(counts, open(os.path.join(case_dir, "counts.json"), "w"))
    json.dump(extra_args, open(os.path.join(case_dir, "args.json"), "w"))
    nfo = os.path.join(out, "sampleX.nfo")
    data = open(nfo, "rb").read() if os.path.exists(nfo) else b""
    open(os.path.join(case_dir, "expected.nfo"), "wb").write(data)
    logs = glob.glob(os.path.join(out, "sampleX_*.out"))
    if logs:
        txt = open(logs[0], "rb").read()
        # first line carries the temp path of the fake BAM: keep everything after it
        open(os.path.join(case_dir, "expected_log.out"), "wb").write(txt.split(b"\r\n", 1)[1])
    calls = bam + ".cmseq_calls.json"
    if os.path.exists(calls):
        shutil.copy(calls, os.path.join(case_dir, "cmseq_calls.json"))
    shutil.rmtree(work)


def part_b(dbpath):
    rng = np.random.default_rng(77)
    base = os.path.join(HERE, "typing")
    if os.path.isdir(base):
        shutil.rmtree(base)
    # case 1: one species passes, the other misses a locus (nloci gate); ties + quirk records; --log
    spec = [("spA", "g1", 4, 40, 9), ("spA", "g2", 5, 45, None), ("spA", "g3", 7, 35, 2), ("spB", "h1", 3, 20, None)]
    counts = {"spA_g1": counts_for(rng, dbpath, "spA", "g1", 4, 0, 0), "spA_g2": counts_for(rng, dbpath, "spA", "g2", 5, 3, 2),
              "spA_g3": counts_for(rng, dbpath, "spA", "g3", 2, 5, 1), "spB_h1": counts_for(rng, dbpath, "spB", "h1", 3, 0, 0),
              "spB_h2": counts_for(rng, dbpath, "spB", "h2", 1, 0, 0)}
    run_metamlst(os.path.join(base, "case1_basic"), dbpath, gen_sam(rng, dbpath, spec), counts, ["--log"])
    # case 2: a locus with exactly 10 % holes -> species dropped (Q7: <=); no .nfo is written
    counts2 = dict(counts)
    counts2["spA_g3"] = counts_for(rng, dbpath, "spA", "g3", 2, 10, 0)
    run_metamlst(os.path.join(base, "case2_gate90"), dbpath, gen_sam(rng, dbpath, spec[:3]), counts2, [])
    # case 3: non-default flags
    run_metamlst(os.path.join(base, "case3_flags"), dbpath, gen_sam(rng, dbpath, spec), counts,
                 ["-a", "--nloci", "50", "--penalty", "50", "--minscore", "100", "--max_xM", "3", "--min_read_len", "60", "--log"])
    # case 4: two species, both complete
    spec4 = spec + [("spB", "h2", 6, 25, None)]
    run_metamlst(os.path.join(base, "case4_two_species"), dbpath, gen_sam(rng, dbpath, spec4), counts, [])
    # case 5: species filter
    run_metamlst(os.path.join(base, "case5_filter"), dbpath, gen_sam(rng, dbpath, spec4), counts, ["--filter", "spB,spZ"])
    # case 6: 9 holes of 100 (kept) and float-format quirks on odd lengths
    counts6 = dict(counts)
    counts6["spA_g3"] = counts_for(rng, dbpath, "spA", "g3", 2, 9, 1)
    counts6["spA_g1"] = counts_for(rng, dbpath, "spA", "g1", 4, 7, 3)
    run_metamlst(os.path.join(base, "case6_formats"), dbpath, gen_sam(rng, dbpath, spec[:3]), counts6, [])


# ------------------------------------------------------------------ part C: metamlst-merge.py
def mutate(s, k, rng):
    s = list(s)
    for p in rng.choice(len(s), size=k, replace=False):
        s[p] = "ACGT"[("ACGT".index(s[p]) + 1 + int(rng.integers(0, 3))) % 4]
    return "".join(s)


def part_c(dbpath):
    rng = np.random.default_rng(99)
    base = os.path.join(HERE, "merge")
    if os.path.isdir(base):
        shutil.rmtree(base)
    conn = sqlite3.connect(dbpath)
    conn.row_factory = sqlite3.Row

    def profile(sp, code):
        return {r["gene"]: r["alleleVariant"] for r in conn.execute(
            "SELECT gene,alleleVariant FROM profiles,alleles WHERE alleleCode=alleles.recID AND profiles.bacterium=? AND profileCode=?", (sp, code))}

    def line(sp, sample, loci):   # loci: {gene: (ref_allele, seq, acc, snp)}
        return sp + "\t" + sample + "\t" + "\t".join("%s_%s_%d::%s::%s::%s" % (sp, g, a, s, acc, snp) for g, (a, s, acc, snp) in loci.items()) + "\r\n"

    p3, p5 = profile("spA", 3), profile("spA", 5)
    novel_ok = mutate(seq_of(dbpath, "spA", "g2", p3["g2"]), 2, rng)
    novel_bad = mutate(seq_of(dbpath, "spA", "g2", p3["g2"]), 40, rng)
    lines = []
    lines.append(line("spA", "s1", {g: (a, "", "100.0", "0.0") for g, a in p3.items()}))                                  # known ST 3
    l2 = {g: (a, "", "100.0", "0.0") for g, a in p3.items()}
    l2["g1"] = (p3["g1"], seq_of(dbpath, "spA", "g1", p5["g1"]), "99.17", "0.83")                                           # located allele
    lines.append(line("spA", "s2", l2))
    l3 = {g: (a, "", "100.0", "0.0") for g, a in p3.items()}
    l3["g2"] = (p3["g2"], novel_ok, "98.0", "1.33")                                                                         # new allele, accepted
    lines.append(line("spA", "s3", l3))
    lines.append(line("spA", "s4", l3))                                                                                     # recurring
    l5 = {g: (a, "", "100.0", "0.0") for g, a in p3.items()}
    l5["g2"] = (p3["g2"], novel_bad, "97.0", "20.0")                                                                        # rejected
    lines.append(line("spA", "s5", l5))
    l6 = {"g1": (p3["g1"], "", "100.0", "0.0"), "g2": (p5["g2"], "", "100.0", "0.0"), "g3": (p3["g3"], "", "100.0", "0.0")}  # old alleles, new combo
    lines.append(line("spA", "s6.fna", l6))
    lines.append(line("spA", "s7", l6))
    l8 = {g: (a, "", "100.0", "0.0") for g, a in p5.items()}
    l8["g3"] = (p5["g3"], seq_of(dbpath, "spA", "g3", p5["g3"]).lower(), "91.0", "0.0")                                     # lower-case -> upper()
    lines.append(line("spA", "s8", l8))
    pb = profile("spB", 2)
    lines.append(line("spB", "s1", {g: (a, "", "100.0", "0.0") for g, a in pb.items()}))
    lb = {g: (a, "", "100.0", "0.0") for g, a in pb.items()}
    lb["h1"] = (pb["h1"], seq_of(dbpath, "spB", "h1", pb["h1"])[:-12], "95.5", "0.0")                                       # truncated: zip() hides it (Q10)
    lines.append(line("spB", "s9", lb))
    meta_text = ("sampleID\tcountry\tage\n" + "s1\tIT\t31\n" + "s2\tIT\t7\n" + "s3\tDE\t52\n" + "s4\tDE\t52\n" + "s6\tFR\t19\n"
                 + "s7\tFR\n" + "s8\tUK\t44\n")        # s5 absent (rejected anyway), s7 malformed (dropped, merge:316)
    for case, z in (("case1_z5", ["-z", "5"]), ("case2_z1", ["-z", "1"]), ("case3_filter", ["--filter", "spA"]),
                    ("case4_seqB", ["--outseqformat", "B"]), ("case5_seqBplus", ["--outseqformat", "B+", "--filter", "spA"]),
                    ("case6_seqC", ["--outseqformat", "C"]), ("case7_seqCplus", ["--outseqformat", "C+"]),
                    ("case8_seqA_meta", ["--outseqformat", "A", "--filter", "spA", "--meta", "META"]),
                    ("case9_seqAplus_j", ["--outseqformat", "A+", "--filter", "spA", "--meta", "META", "-j", "country,age"]),
                    ("case10_seqA_jgroup", ["--outseqformat", "A", "--filter", "spA", "--meta", "META", "--idField", "0", "-j", "country", "--jgroup"]),
                    ("case11_seqA_nometa", ["--outseqformat", "A", "--filter", "spA", "-z", "1"])):
        work = tempfile.mkdtemp()
        open(os.path.join(work, "all.nfo"), "w", newline="").write("".join(lines))
        meta_dir = tempfile.mkdtemp()
        if "META" in z:
            open(os.path.join(meta_dir, "meta.tsv"), "w").write(meta_text)
        z_run = [os.path.join(meta_dir, "meta.tsv") if t == "META" else t for t in z]
        env = dict(os.environ)
        env["PYTHONPATH"] = STUBS + os.pathsep + REF
        r = subprocess.run([sys.executable, os.path.join(REF, "metamlst-merge.py"), work, "-d", dbpath] + z_run, env=env, cwd=work,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode != 0:
            raise RuntimeError("reference metamlst-merge.py failed: " + r.stderr.decode()[-2000:])
        cd = os.path.join(base, case)
        os.makedirs(os.path.join(cd, "expected"))
        shutil.copy(os.path.join(work, "all.nfo"), os.path.join(cd, "all.nfo"))
        json.dump(z, open(os.path.join(cd, "args.json"), "w"))
        for f in glob.glob(os.path.join(work, "merged", "*")):
            shutil.copy(f, os.path.join(cd, "expected", os.path.basename(f)))
        if "META" in z:
            shutil.copy(os.path.join(meta_dir, "meta.tsv"), os.path.join(cd, "meta.tsv"))
        shutil.rmtree(meta_dir)
        shutil.rmtree(work)
    conn.close()


# ------------------------------------------------------------------ part D: metamlst-index.py (FASTA + typings ingest)
def dump_tables(dbpath):
    conn = sqlite3.connect(dbpath)
    out = {t: [list(r) for r in conn.execute("SELECT * FROM %s ORDER BY rowid" % t)] for t in ("organisms", "genes", "alleles", "profiles")}
    conn.close()
    return out


def part_d(dbpath):
    rng = np.random.default_rng(5)
    base = os.path.join(HERE, "dbbuild")
    if os.path.isdir(base):
        shutil.rmtree(base)
    os.makedirs(base)

    def rnd(n):
        return "".join(rng.choice(list("ACGT"), n))

    recs = []
    for g, L in (("g1", 60), ("g2", 75)):
        for a in range(1, 5):
            recs.append(("spC_%s_%d" % (g, a), rnd(L)))
    recs.append(("sp-D_hk_1 some description", rnd(40)))
    recs.append(("spC_g1", rnd(30)))                 # malformed: two parts
    recs.append(("spC_g+1_3", rnd(30)))              # invalid character in the gene
    recs.append(("spC_g1_x3", rnd(30)))              # allele not numeric
    recs.append(("spC_g2_4", rnd(75)))               # same id again inside one file: both are inserted (quirk)
    recs.append(("spC_g2_7", rnd(75).lower()))       # lower case kept as is
    with open(os.path.join(base, "a.fasta"), "w") as f:
        for k, (i, sq) in enumerate(recs):
            f.write(">" + i + "\n")
            f.write((sq[:25] + "\n" + sq[25:] + "\n") if k % 2 else sq + "\n")       # multi-line records
    with open(os.path.join(base, "b.fasta"), "w") as f:
        f.write(">spC_g1_2\n" + rnd(60) + "\n>spC_g1_9\n" + rnd(60) + "\n>spE_q_1\n" + rnd(50) + "\n")   # first one already present
    with open(os.path.join(base, "typ.txt"), "w") as f:
        f.write("@ a comment line\n#spC|Species C label\nST\tg1\tg2\tclonal_complex\n1\t1\t1\tCC1\n2\t2\t1\tCC1\n3\t5\t1\t-\n4\t1\t2\n5\t9\t7\tCC9\n")
    with open(os.path.join(base, "typ2.txt"), "w") as f:
        f.write("#sp_E\nST\tq\tspecies\n1\t1\tsomething\n2\t2\tsomething\n")
    work = tempfile.mkdtemp()
    for fn in ("a.fasta", "b.fasta", "typ.txt", "typ2.txt"):
        shutil.copy(os.path.join(base, fn), work)
    env = dict(os.environ)
    env["PYTHONPATH"] = STUBS + os.pathsep + REF
    r = subprocess.run([sys.executable, os.path.join(REF, "metamlst-index.py"), "-d", "new.db", "-s", "a.fasta,b.fasta", "-t", "typ.txt,typ2.txt"],
                       env=env, cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        raise RuntimeError("reference metamlst-index.py failed: " + r.stderr.decode()[-2000:])
    json.dump(dump_tables(os.path.join(work, "new.db")), open(os.path.join(base, "expected_tables.json"), "w"), indent=0)
    lf = os.path.join(work, "metamlst_logfile.log")
    open(os.path.join(base, "expected_logfile.log"), "wb").write(open(lf, "rb").read() if os.path.exists(lf) else b"")
    # second run on the same database: everything is already present, profiles are replaced
    r = subprocess.run([sys.executable, os.path.join(REF, "metamlst-index.py"), "-d", "new.db", "-s", "a.fasta", "-t", "typ.txt"],
                       env=env, cwd=work, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        raise RuntimeError("reference metamlst-index.py (2nd run) failed: " + r.stderr.decode()[-2000:])
    json.dump(dump_tables(os.path.join(work, "new.db")), open(os.path.join(base, "expected_tables_second_run.json"), "w"), indent=0)
    shutil.rmtree(work)


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs /root/reference (authoring container only)")
    tmp = tempfile.mkdtemp()
    dbpath = os.path.join(tmp, "golden.db")
    build_db(dbpath)
    part_a(dbpath)
    part_b(dbpath)
    part_c(dbpath)
    part_d(dbpath)
    print("golden vectors written under", HERE)


if __name__ == "__main__":
    main()
