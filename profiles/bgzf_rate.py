#!/usr/bin/env python3
"""Rate of the compressed-FASTQ entries: a bgzip'd FASTQ file read from disk (page cache) to pass-1 statistics,
(a) inflated on the host (gzip module) and parsed on the GPU, (b) inflated and parsed on the GPU (mlst_submit_fastq_bgzf)."""
import gzip
import json
import os
import struct
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from metamlst_amd import synth  # noqa: E402
from metamlst_amd.engine import Engine  # noqa: E402
from metamlst_amd.fastq import bgzf_chunks, text_chunks  # noqa: E402
from metamlst_amd.index import load_index  # noqa: E402


def bgzf_block(data: bytes) -> bytes:
    c = zlib.compressobj(4, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp
            + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


N = int(os.environ.get("N_READS", "2000000"))
d = tempfile.mkdtemp()
db = synth.make_ecoli_db(d + "/e.db", alleles_per_locus=1430, n_profiles=100)
idx = load_index(db.path)
g, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][3])
b, q = synth.sample_reads(g, N)
rows = np.empty((N, 150 + 1 + 2 + 150 + 1), np.uint8)
rows[:, :150] = b; rows[:, 150] = 10; rows[:, 151] = ord("+"); rows[:, 152] = 10; rows[:, 153:303] = q; rows[:, 303] = 10
names = [b"@r%d\n" % k for k in range(N)]
text = b"".join(n + r.tobytes() for n, r in zip(names, rows))
path = d + "/s.fastq.gz"
with open(path, "wb") as f:
    for at in range(0, len(text), 65280):
        f.write(bgzf_block(text[at:at + 65280]))
    f.write(bgzf_block(b""))
comp_bytes = os.path.getsize(path)
eng = Engine(0)
eng.load_reference(idx)
out = {"reads": N, "text_bytes": len(text), "bgzf_bytes": comp_bytes}
for name, fn in (("host_inflate_gpu_parse", lambda: sum(eng.submit_fastq(c) for c in text_chunks(path))),
                 ("gpu_inflate_gpu_parse", lambda: eng.submit_fastq_bgzf_file(path))):
    ts, stats = [], None
    for _ in range(3):
        eng.reset_sample()
        t0 = time.perf_counter()
        n = fn()
        st = eng.stats()
        ts.append(time.perf_counter() - t0)
        assert n == N
        stats = st
    t = float(np.median(ts))
    out[name] = {"s": round(t, 3), "Mreads_per_s": round(N / t / 1e6, 2), "text_GB_per_s": round(len(text) / t / 1e9, 2), "records": int(stats.counters[0])}
assert out["host_inflate_gpu_parse"]["records"] == out["gpu_inflate_gpu_parse"]["records"]
print(json.dumps(out))
