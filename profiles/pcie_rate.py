#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry (mlst_submit_reads): FASTQ fields in pageable host memory ->
H2D copy -> pack kernel -> pass 1.  Reported in DESIGN.md next to the resident-in-HBM figure of bench.py."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from metamlst_amd import synth  # noqa: E402
from metamlst_amd.engine import Engine  # noqa: E402
from metamlst_amd.index import load_index  # noqa: E402

d = tempfile.mkdtemp()
db = synth.make_ecoli_db(d + "/e.db", alleles_per_locus=1430, n_profiles=100)
idx = load_index(db.path)
g, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][3])
b, q = synth.sample_reads(g, 4_000_000)
fb, fq, off = synth.flatten_reads(b, q)
eng = Engine(0)
eng.load_reference(idx)
eng.submit_reads(fb, fq, off)
eng.stats()
ts = []
for _ in range(5):
    eng.reset_sample()
    t0 = time.perf_counter()
    eng.submit_reads(fb, fq, off)
    eng.stats()
    ts.append(time.perf_counter() - t0)
t = float(np.median(ts))
print(json.dumps({"entry": "mlst_submit_reads (pageable host ASCII bases + quals, 300 B/read over PCIe)", "reads": 4_000_000,
                  "ms": round(t * 1e3, 2), "Mreads_per_s": round(4.0 / t, 1), "GB_per_s_host_to_device": round(4e6 * 300 / t / 1e9, 1)}))

# FASTQ text parsed on the GPU (mlst_submit_fastq): the bytes of the file cross PCIe once
text = b"".join(b"@r%d\n" % k + b[k].tobytes() + b"\n+\n" + q[k].tobytes() + b"\n" for k in range(1_000_000))
buf = np.frombuffer(text, np.uint8)
eng.reset_sample()
eng.submit_fastq(buf)
eng.stats()
ts = []
for _ in range(5):
    eng.reset_sample()
    t0 = time.perf_counter()
    eng.submit_fastq(buf)
    eng.stats()
    ts.append(time.perf_counter() - t0)
t = float(np.median(ts))
print(json.dumps({"entry": "mlst_submit_fastq (FASTQ text in pageable host memory, parsed on the GPU)", "reads": 1_000_000, "text_bytes": len(text),
                  "ms": round(t * 1e3, 2), "Mreads_per_s": round(1.0 / t, 1), "GB_per_s_host_to_device": round(len(text) / t / 1e9, 1)}))
