#!/usr/bin/env python3
"""bgzip'd FASTQ in host memory -> pass-1 statistics, fed in pieces of 16,384 BGZF blocks the way a file is:
the three-stream path of round 5 (MLST_BGZF_PIPE=1, default) against the serial one (MLST_BGZF_PIPE=0), same statistics.
    python profiles/e2e_bgzf.py [reads] [level]"""
import json
import os
import struct
import sys
import tempfile
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from metamlst_amd import synth  # noqa: E402
from metamlst_amd.engine import Engine  # noqa: E402
from metamlst_amd.index import load_index  # noqa: E402


def block(args):
    data, level = args
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp
            + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
LEVEL = int(sys.argv[2]) if len(sys.argv) > 2 else 6
d = tempfile.mkdtemp()
db = synth.make_ecoli_db(d + "/e.db", alleles_per_locus=300, n_profiles=50)
idx = load_index(db.path)
g, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][3], size=1_000_000)
L = 150
rec = 16 + 2 * L
rows = np.empty((N, rec), np.uint8)
for at in range(0, N, 1 << 20):      # (a million reads at a time: sample_reads builds an index array of 8 bytes per base)
    c = min(1 << 20, N - at)
    b, q = synth.sample_reads(g, c, seed=synth.SEED + at)
    rows[at:at + c, 12:12 + L] = b
    rows[at:at + c, 15 + L:15 + 2 * L] = q
rows[:, :12] = np.frombuffer(b"@r000000000\n", np.uint8)      # fixed-width names (12 bytes with the newline)
num = np.arange(N)
for k in range(9):
    rows[:, 10 - k] = 48 + (num // 10 ** k) % 10
rows[:, 12 + L] = 10; rows[:, 13 + L] = ord("+"); rows[:, 14 + L] = 10; rows[:, 15 + 2 * L] = 10
raw = rows.tobytes()
t0 = time.perf_counter()
with ThreadPoolExecutor(max(1, min(64, os.cpu_count() or 1))) as ex:
    parts = list(ex.map(block, [(raw[at:at + 65280], LEVEL) for at in range(0, len(raw), 65280)]))
comp = np.frombuffer(b"".join(parts) + block((b"", LEVEL)), np.uint8)
print("compressed %d blocks in %.1f s: %.1f MB -> %.1f MB" % (len(parts), time.perf_counter() - t0, len(raw) / 1e6, comp.size / 1e6), file=sys.stderr, flush=True)
per_piece = int(os.environ.get("PER_PIECE", "16384"))
cuts = np.concatenate([[0], np.cumsum([len(x) for x in parts])])
pieces = [comp[int(cuts[a]):int(cuts[min(a + per_piece, len(parts))])] for a in range(0, len(parts), per_piece)]
pieces[-1] = comp[int(cuts[(len(pieces) - 1) * per_piece]):]
if os.environ.get("PINNED"):      # page-locked source buffers, as the file reader's (engine.pinned_array): one DMA transfer per piece
    from metamlst_amd.engine import pinned_array
    pinned = []
    for pc in pieces:
        b_ = pinned_array(pc.size)
        b_[:pc.size] = pc
        pinned.append(b_[:pc.size])
    pieces = pinned

out = {"reads": N, "text_bytes": len(raw), "bgzf_bytes": int(comp.size), "level": LEVEL, "pieces": len(pieces)}
ref = None
for mode in os.environ.get("MODES", "0,1").split(","):
    os.environ["MLST_BGZF_PIPE"] = mode
    eng = Engine(0)
    eng.load_reference(idx)
    ts = []
    for _ in range(4):
        eng.reset_sample()
        eng.synchronize()
        t0 = time.perf_counter()
        n = 0
        for k, pc in enumerate(pieces):
            n += eng.submit_fastq_bgzf(pc, final=(k == len(pieces) - 1))
        st = eng.stats()
        ts.append(time.perf_counter() - t0)
        assert n == N, (n, N)
    t = min(ts[1:])
    key = (st.sum_score.tobytes(), st.n_hits.tobytes(), st.locus_len_sum.tobytes(), st.locus_first.tobytes(), tuple(int(x) for x in st.counters[:4]))
    if ref is None:
        ref = key
    assert key == ref, "the two paths differ"
    out["pipe_" + mode] = {"s": round(t, 4), "Mreads_per_s": round(N / t / 1e6, 1), "ms_per_piece": round(t / len(pieces) * 1e3, 2), "records": int(st.counters[0]), "all_runs_s": [round(x, 4) for x in ts]}
    del eng
print(json.dumps(out))
