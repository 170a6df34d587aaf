#!/usr/bin/env python3
"""`cli type folder/` on bgzip'd samples in memory-backed storage, with a host-side trace of who waits for what:
    python profiles/folder_bgzf.py [samples] [reads per sample] [engines] [text]"""
import json
import os
import struct
import sys
import tempfile
import threading
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from metamlst_amd import engine as eng_mod, fastq, synth  # noqa: E402
from metamlst_amd import db as mdb  # noqa: E402
from metamlst_amd.index import load_index  # noqa: E402
from metamlst_amd.multigpu import type_many_samples  # noqa: E402
from metamlst_amd.pipeline import make_engines  # noqa: E402
from metamlst_amd.typing import TypingArgs  # noqa: E402


def block(data):
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp
            + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


NS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
PER = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
NE = int(sys.argv[3]) if len(sys.argv) > 3 else 4
TEXT = len(sys.argv) > 4 and sys.argv[4] == "text"      # plain FASTQ text files instead (the other folder leg of bench.py)
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
sdb = synth.make_ecoli_db(d + "/e.db", alleles_per_locus=300, n_profiles=50)
idx = load_index(sdb.path)
database = mdb.metaMLST_db(sdb.path)
g, _ = synth.make_genome(sdb, "ecoli", sdb.profiles["ecoli"][3], size=1_000_000)
L = 150
rec = 16 + 2 * L
files = []
with ThreadPoolExecutor(max(1, min(64, os.cpu_count() or 1))) as ex:
    for s in range(NS):
        rows = np.empty((PER, rec), np.uint8)
        for at in range(0, PER, 1 << 20):
            c = min(1 << 20, PER - at)
            b, q = synth.sample_reads(g, c, seed=synth.SEED + 7 * s + at)
            rows[at:at + c, 12:12 + L] = b
            rows[at:at + c, 15 + L:15 + 2 * L] = q
        rows[:, :12] = np.frombuffer(b"@r000000000\n", np.uint8)
        num = np.arange(PER)
        for k in range(9):
            rows[:, 10 - k] = 48 + (num // 10 ** k) % 10
        rows[:, 12 + L] = 10; rows[:, 13 + L] = ord("+"); rows[:, 14 + L] = 10; rows[:, 15 + 2 * L] = 10
        if TEXT:
            f = "%s/s%02d.fastq" % (d, s)
            rows.tofile(f)
            files.append([f])
            continue
        raw = rows.tobytes()
        parts = list(ex.map(block, [raw[at:at + 65280] for at in range(0, len(raw), 65280)]))
        f = "%s/s%02d.fastq.gz" % (d, s)
        with open(f, "wb") as fh:
            fh.write(b"".join(parts) + block(b""))
        files.append([f])
print("files ready: %d x %.1f MB" % (NS, os.path.getsize(files[0][0]) / 1e6), file=sys.stderr, flush=True)

T0 = time.perf_counter()
trace = []
lock = threading.Lock()


def wrap(obj, name, label):
    orig = getattr(obj, name)

    def f(*a, **k):
        t0 = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            with lock:
                trace.append((threading.current_thread().name, label, (t0 - T0) * 1e3, (time.perf_counter() - T0) * 1e3))
    setattr(obj, name, f)


if os.environ.get("TRACE"):
    wrap(fastq, "_pread_into", "read")
    for nm in ("submit_fastq_bgzf_file", "submit_fastq", "typing_enqueue", "typing_wait", "typing_fetch", "reset_sample"):
        wrap(eng_mod.Engine, nm, nm)

engines = make_engines(idx, 0, NE)
out = {"samples": NS, "reads_per_sample": PER, "engines": NE, "runs": []}
for r in range(4):
    od = "%s/out%d" % (d, r)
    tm = {}
    if r == 3:
        trace.clear(); T0 = time.perf_counter()
    t0 = time.perf_counter()
    rc = type_many_samples(engines, idx, database, TypingArgs(quiet=True), files, 0, 1, od, False, 256 << 20, timing=tm)
    t = time.perf_counter() - t0
    assert rc == 0
    out["runs"].append({"s": round(t, 4), "prologue_s": round(tm["prologue_s"], 4), "samples_s": round(tm["samples_s"], 4),
                        "Mreads_per_s": round(NS * PER / tm["samples_s"] / 1e6, 1), "host_ms": {k: round(v, 2) for k, v in tm["host_ms"].items()}})
print(json.dumps(out))
if trace:
    for th, lab, a, b in sorted(trace, key=lambda x: x[2]):
        print("%-14s %-24s %8.2f -> %8.2f (%.2f)" % (th[:14], lab, a, b, b - a), file=sys.stderr)
import shutil
shutil.rmtree(d, ignore_errors=True)
