"""Opportunistic LITERAL parity with the reference's aligner.

The reference's read alignment is a user-run `bowtie2 --very-sensitive-local -a --no-unal` (wiki via README.md:20);
neither bowtie2 nor samtools ships with the reference or with this image, which is why row a1 is "parity unpinned".
On a machine that HAS both tools this module runs the documented command on a set of reads, feeds the BAM to this
build's `--alignments` path (metamlst.py:101-130 restated in samin.py, pileup on the GPU) and compares it with the
FASTQ path (alignment on the GPU): the ST-relevant outcome (chosen alleles, .nfo line) and, record by record in
aggregate, every allele's (hits, sum of AS).  Nothing here is on the product path; without the tools it reports
{"skipped": ...}."""
from __future__ import annotations

import os
import shutil
import subprocess
import tempfile

import numpy as np


def tools() -> dict | None:
    t = {k: shutil.which(k) for k in ("bowtie2", "bowtie2-build", "samtools")}
    return t if all(t.values()) else None


def literal_parity(engine, index, database, db_path: str, fastq_path: str, threads: int = 8, keep_dir: str | None = None) -> dict:
    """engine: an Engine with `index` loaded; fastq_path: plain FASTQ of single reads.  -> comparison dict."""
    from . import dbbuild
    from .cli import submit_sample_files
    from .samin import AlignmentSample
    from .typing import TypingArgs, type_sample
    t = tools()
    if t is None:
        return {"skipped": "bowtie2 / bowtie2-build / samtools not on PATH (the reference ships none of them)"}
    d = keep_dir or tempfile.mkdtemp(prefix="mlst_literal_")
    fa, bam = os.path.join(d, "alleles.fa"), os.path.join(d, "sample.bam")
    conn = dbbuild.open_db(db_path)
    dbbuild.dump_db_to_fasta(conn, fa)                       # metaMLST_functions.py:149-161, as metamlst-index.py -i does
    conn.close()
    try:
        subprocess.run([t["bowtie2-build"], "-q", "--threads", str(threads), fa, os.path.join(d, "idx")], check=True, capture_output=True, timeout=3600)
        # README.md:20 -> wiki: bowtie2 --threads N --very-sensitive-local -a --no-unal -x INDEX -U READS | samtools view -bS - > BAM
        p1 = subprocess.Popen([t["bowtie2"], "--threads", str(threads), "--very-sensitive-local", "-a", "--no-unal", "-x", os.path.join(d, "idx"), "-U", fastq_path],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        with open(bam, "wb") as f:
            p2 = subprocess.run([t["samtools"], "view", "-bS", "-"], stdin=p1.stdout, stdout=f, stderr=subprocess.DEVNULL, timeout=36000)
        p1.wait()
        if p1.returncode or p2.returncode:
            return {"error": "bowtie2 | samtools failed (%s, %s)" % (p1.returncode, p2.returncode)}
    except Exception as e:      # noqa: BLE001 -- an optional leg must not take the bench down
        return {"error": "%s: %s" % (type(e).__name__, e)}
    targs = TypingArgs()
    smp = AlignmentSample(index, targs).add_file(bam)
    s_bam = smp.stats()
    r_bam = type_sample(index, s_bam, lambda chosen: smp.pileup(engine, chosen), database, "sample", targs, out_dir=None)
    engine.reset_sample()
    submit_sample_files(engine, [fastq_path], False, 256 << 20)
    s_gpu = engine.stats()
    r_gpu = type_sample(index, s_gpu, engine.pileup, database, "sample", targs, out_dir=None)
    hit = (s_bam.n_hits > 0) | (s_gpu.n_hits > 0)
    same = (s_bam.n_hits == s_gpu.n_hits) & (s_bam.sum_score == s_gpu.sum_score)
    rec_bam, rec_gpu = int(s_bam.n_hits.sum()), int(s_gpu.n_hits.sum())
    return {"tools": t, "records_accepted": {"bowtie2": rec_bam, "gpu": rec_gpu},
            "alleles_with_hits": int(hit.sum()), "alleles_identical_hits_and_score": int((same & hit).sum()),
            "records_missing_on_gpu": int(np.clip(s_bam.n_hits.astype(np.int64) - s_gpu.n_hits.astype(np.int64), 0, None).sum()),
            "records_extra_on_gpu": int(np.clip(s_gpu.n_hits.astype(np.int64) - s_bam.n_hits.astype(np.int64), 0, None).sum()),
            "sum_score": {"bowtie2": int(s_bam.sum_score.sum()), "gpu": int(s_gpu.sum_score.sum())},
            "chosen_alleles_equal": [r.chosen for r in r_bam] == [r.chosen for r in r_gpu],
            "nfo_lines_equal": [r.nfo_line for r in r_bam] == [r.nfo_line for r in r_gpu],
            "nfo_lines": {"bowtie2": sum(1 for r in r_bam if r.written), "gpu": sum(1 for r in r_gpu if r.written)}}
