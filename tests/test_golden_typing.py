"""metamlst.py:96-289 pinned: the reference script was run on synthetic SAM records
(tests/golden/make_golden.py, part B); here the same records go through the oracle's
accumulate_read (C) and the package's typing host logic, and the .nfo / --log bytes must match."""
import glob
import json
import os

import numpy as np
import pytest

import golden_util as gu
import oracle_lib
from metamlst_amd import db as mdb
from metamlst_amd.engine import default_params
from metamlst_amd.index import load_index
from metamlst_amd.typing import TypingArgs, log_table, type_sample

CASES = sorted(glob.glob(os.path.join(gu.GOLD, "typing", "case*")))


def parse_args(argv):
    t, p = TypingArgs(), default_params()
    it = iter(argv)
    for a in it:
        if a == "-a":
            t.a = True
        elif a == "--log":
            t.log = True
        elif a == "--nloci":
            t.nloci = int(next(it))
        elif a == "--penalty":
            t.penalty = int(next(it))
        elif a == "--minscore":
            t.minscore = p.minscore = int(next(it))
        elif a == "--max_xM":
            t.max_xM = p.max_xm = int(next(it))
        elif a == "--min_read_len":
            t.min_read_len = p.min_read_len = int(next(it))
        elif a == "--filter":
            t.filter = next(it)
    return t, p


@pytest.mark.parametrize("case", CASES, ids=[os.path.basename(c) for c in CASES])
@pytest.mark.parametrize("fast", [False, True])
def test_reference_nfo_bytes(case, fast):
    targs, prm = parse_args(json.load(open(os.path.join(case, "args.json"))))
    dbp = gu.golden_db()
    idx = load_index(dbp, targs.filter.split(",") if targs.filter else None)
    orc = oracle_lib.Oracle(idx, prm)
    st = orc.accumulate_records(*gu.parse_sam(os.path.join(case, "input.sam"), idx))
    counts = json.load(open(os.path.join(case, "counts.json")))

    def pileup_fn(chosen):
        return {a: np.array(counts["%s_%s" % idx.loci[int(idx.locus_id[a])]], np.uint32) for a in chosen}

    cache = mdb.DbCache(mdb.metaMLST_db(dbp).conn) if fast else None
    res = type_sample(idx, st, pileup_fn, mdb.metaMLST_db(dbp), "sampleX", targs, fast=fast, cache=cache)
    got = "".join(r.nfo_line for r in res if r.written).encode()
    assert got == open(os.path.join(case, "expected.nfo"), "rb").read()
    logf = os.path.join(case, "expected_log.out")
    if os.path.exists(logf) and not fast:
        mine = log_table(idx, st, targs, "x").encode().split(b"\r\n", 1)[1]
        assert mine == open(logf, "rb").read()


def test_cases_cover_the_quirks():
    names = [os.path.basename(c) for c in CASES]
    assert len(names) >= 6
    assert open(os.path.join(gu.GOLD, "typing", "case2_gate90", "expected.nfo"), "rb").read() == b""      # Q7: exactly 90 % dropped
    assert b"1.3299999999999998" in open(os.path.join(gu.GOLD, "typing", "case1_basic", "expected.nfo"), "rb").read()   # float format quirk
