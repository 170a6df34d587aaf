"""Helpers for the golden-vector tests: load the committed SQL dump, parse committed SAM text."""
from __future__ import annotations

import json
import os
import sqlite3
import tempfile

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_TMP = tempfile.mkdtemp(prefix="mlst_gold_")


def golden_db() -> str:
    path = os.path.join(_TMP, "golden.db")
    if not os.path.exists(path):
        conn = sqlite3.connect(path)
        conn.executescript(open(os.path.join(GOLD, "db.sql")).read())
        conn.commit()
        conn.close()
    return path


def functions():
    return json.load(open(os.path.join(GOLD, "functions.json")))


def parse_sam(path: str, index):
    """SAM text -> record arrays the way the engine sees them: one record per line, reads numbered by
    first appearance, true tag values (AS, XM, XO) looked up BY NAME.  Records whose contig is not in
    the loaded index (species filter) are dropped, as the engine never aligns to alleles it did not load."""
    label2a = {index.label(a): a for a in range(index.n_alleles)}
    rid, qmap = [], {}
    allele, AS, XM, XO, slen = [], [], [], [], []
    for line in open(path):
        if line[0] == "@":
            continue
        f = line.rstrip("\n").split("\t")
        if f[0] not in qmap:
            qmap[f[0]] = len(qmap)
        if f[2] not in label2a:
            continue
        tags = {t.split(":")[0]: t.split(":")[2] for t in f[11:]}
        rid.append(qmap[f[0]])
        allele.append(label2a[f[2]])
        AS.append(int(tags["AS"]))
        XM.append(int(tags["XM"]))
        XO.append(int(tags["XO"]))
        slen.append(len(f[9]))
    return (np.array(rid, np.uint64), np.array(allele, np.uint32), np.array(AS, np.int32), np.array(XM, np.int32),
            np.array(XO, np.int32), np.array(slen, np.int32))
