"""SAM / BAM input.  The accumulation of metamlst.py:101-130 restated in metamlst_amd/samin.py is
pinned by the golden cases the reference itself produced from these very SAM files (tests/golden/typing); the BAM
decoder is checked against SAM text through a BAM written by tests/bam_writer.py."""
import glob
import json
import os

import numpy as np
import pytest

import bam_writer
import golden_util as gu
import oracle_lib
from metamlst_amd import db as mdb
from metamlst_amd import samin
from metamlst_amd.index import load_index
from metamlst_amd.typing import log_table, type_sample
from test_golden_typing import parse_args

CASES = sorted(glob.glob(os.path.join(gu.GOLD, "typing", "case*")))


@pytest.mark.parametrize("case", CASES, ids=[os.path.basename(c) for c in CASES])
def test_sam_accumulation_reproduces_the_reference_nfo(case):
    targs, prm = parse_args(json.load(open(os.path.join(case, "args.json"))))
    dbp = gu.golden_db()
    idx = load_index(dbp, targs.filter.split(",") if targs.filter else None)
    smp = samin.AlignmentSample(idx, targs).add_file(os.path.join(case, "input.sam"))
    st = smp.stats()
    so = oracle_lib.Oracle(idx, prm).accumulate_records(*gu.parse_sam(os.path.join(case, "input.sam"), idx))
    assert np.array_equal(st.sum_score, so.sum_score) and np.array_equal(st.n_hits, so.n_hits)
    assert np.array_equal(st.locus_len_sum, so.locus_len_sum)
    assert int(st.counters[0]) == int(so.counters[0]) and int(st.counters[1]) == int(so.counters[1])
    counts = json.load(open(os.path.join(case, "counts.json")))

    def pileup_fn(chosen):
        return {a: np.array(counts["%s_%s" % idx.loci[int(idx.locus_id[a])]], np.uint32) for a in chosen}

    res = type_sample(idx, st, pileup_fn, mdb.metaMLST_db(dbp), "sampleX", targs)
    assert "".join(r.nfo_line for r in res if r.written).encode() == open(os.path.join(case, "expected.nfo"), "rb").read()
    logf = os.path.join(case, "expected_log.out")
    if os.path.exists(logf):
        assert log_table(idx, st, targs, "x").encode().split(b"\r\n", 1)[1] == open(logf, "rb").read()


def _records(n=400, seed=5):
    rng = np.random.default_rng(seed)
    refs = [("spA_g%d_%d" % (g, a), 400 + 10 * g) for g in (1, 2) for a in (1, 2, 3)]
    recs = []
    for k in range(n):
        L = int(rng.integers(1, 160))
        seq = "".join(rng.choice(list("ACGTN"), size=L, p=[.24, .24, .24, .24, .04]))
        qual = "".join(chr(33 + int(x)) for x in rng.integers(0, 42, size=L))
        kind = k % 5
        if kind == 0:
            cigar = "%dM" % L
        elif kind == 1 and L > 20:
            cigar = "5S%dM2D%dM3S" % ((L - 8) // 2, L - 8 - (L - 8) // 2)
        elif kind == 2 and L > 20:
            cigar = "%d=1X%dM1I%dM" % (4, L - 4 - 1 - 1 - 5, 5)
        elif kind == 3 and L > 30:
            cigar = "2H10M100N%dM" % (L - 10)
        else:
            cigar = "%dM" % L
        tags = ["AS:i:%d" % int(rng.integers(-5, 300))] + (["XS:i:%d" % int(rng.integers(0, 300))] if k % 3 else []) + \
               ["XN:i:0", "XM:i:%d" % int(rng.integers(0, 9)), "XO:i:%d" % int(rng.integers(0, 3)), "XG:i:0", "NM:i:70000", "YT:Z:UU", "ZA:A:x", "ZF:f:1.5"]
        if k % 17 == 0:
            seq, qual = seq, "*"
        recs.append(("read%d" % (k // 2), [0, 16, 256, 272][k % 4], refs[k % len(refs)][0], int(rng.integers(1, 380)), 255, cigar, seq, qual, tags))
    return refs, recs


def test_bam_decoder_equals_sam_text(tmp_path):
    refs, recs = _records()
    hdr = "@HD\tVN:1.0\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    sam = tmp_path / "x.sam"
    with open(sam, "w") as f:
        f.write(hdr)
        for r in recs:
            f.write("\t".join([r[0], str(r[1]), r[2], str(r[3]), str(r[4]), r[5], "*", "0", "0", r[6], r[7]] + r[8]) + "\n")
    bam = tmp_path / "x.bam"
    bam_writer.write_bam(str(bam), hdr, refs, recs)
    a, b = list(samin.read_alignments(str(sam))), list(samin.read_alignments(str(bam)))
    assert len(a) == len(b) == len(recs)
    for x, y in zip(a, b):
        assert x == y
    import gzip, shutil
    with open(sam, "rb") as fi, gzip.open(str(sam) + ".gz", "wb") as fo:
        shutil.copyfileobj(fi, fo)
    assert list(samin.read_alignments(str(sam) + ".gz")) == a


def test_cigar_codes():
    assert samin.parse_cigar("5S100M2D45M") == [(5 << 4) | 4, (100 << 4) | 0, (2 << 4) | 2, (45 << 4) | 0]
    assert samin.parse_cigar("*") == []


def test_unaligned_record_raises_like_the_reference():
    idx = load_index(gu.golden_db())
    with pytest.raises(ValueError):      # RNAME '*' does not split into three parts (metamlst.py:106; hence --no-unal)
        samin.AlignmentSample(idx).add(samin.Alignment("r", 4, "*", 0, "*", "ACGT", "IIII", ["YT:Z:UP"]))


@pytest.mark.gpu
def test_alignment_pileup_on_the_gpu_equals_the_literal_loop():
    """mlst_pileup_alignments against the per-base Python loop of tests/samin_ref.py: soft / hard clips, insertions,
    deletions, skips, = / X, N bases, low qualities, missing qualities, records failing the AS / XM tag filter,
    records hanging over the contig's ends, contigs that were not chosen."""
    import samin_ref
    from metamlst_amd.engine import Engine
    idx = load_index(gu.golden_db())
    labels = [idx.label(a) for a in range(idx.n_alleles)]
    _, recs = _records(n=3000, seed=9)
    smp = samin.AlignmentSample(idx)
    rng = np.random.default_rng(1)
    for r in recs:
        lab = labels[int(rng.integers(0, len(labels)))]
        ln = int(idx.off[idx.n_alleles and labels.index(lab) + 1] - idx.off[labels.index(lab)])
        pos = int(rng.integers(1, max(2, ln)))
        smp.add(samin.Alignment(r[0], r[1], lab, pos, r[5], r[6], r[7], r[8]))
    eng = Engine(0)
    eng.load_reference(idx)
    chosen = sorted({int(idx.locus_begin[l]) + (l % int(idx.locus_count[l])) for l in range(idx.n_loci)})
    got = smp.pileup(eng, chosen)
    want = samin_ref.pileup_python(idx, smp, chosen)
    assert set(got) == set(want) and sum(int(v.sum()) for v in want.values()) > 1000
    for a in want:
        assert np.array_equal(got[a], want[a]), a
    assert all(int(v.sum()) == 0 for v in smp.pileup(eng, chosen, minqual=99).values())
    assert smp.pileup(eng, []) == {}
