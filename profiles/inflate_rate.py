#!/usr/bin/env python3
"""Throughput of the device inflate kernel alone (HIP events around k_inflate): bgzip'd FASTQ of N reads, level 1 and 6.
    python profiles/inflate_rate.py [reads] [levels, e.g. 6 or 1,6]"""
import os, struct, sys, time, zlib
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build()
from metamlst_amd.engine import Engine

def block(args):
    data, level = args
    c = zlib.compressobj(level, zlib.DEFLATED, -15); comp = c.compress(data) + c.flush()
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(1)
L = 150
bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, L))]
quals = np.full((n, L), 73, np.uint8); err = rng.random((n, L)) < 0.01; quals[err] = 48
recs = [b"@r%d x\n" % k + bases[k].tobytes() + b"\n+\n" + quals[k].tobytes() + b"\n" for k in range(n)]
text = b"".join(recs)
eng = Engine(0)
for level in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,6").split(",")]:
    with ThreadPoolExecutor(16) as ex:
        blocks = list(ex.map(block, [(text[i:i + 65280], level) for i in range(0, len(text), 65280)]))
    comp = b"".join(blocks)
    got = eng.inflate_bgzf(comp)
    assert got == text
    ms = []
    for _ in range(3):
        eng.inflate_bgzf(comp); ms.append(eng.last_inflate_ms)
    print("level %d: %d blocks, %.1f MB -> %.1f MB, k_inflate %.3f ms = %.1f GB/s of text" % (level, len(blocks), len(comp) / 1e6, len(text) / 1e6, min(ms), len(text) / min(ms) / 1e6), flush=True)
