#!/usr/bin/env python3
"""Block-by-block check of the device inflate kernel against zlib (debugging aid for csrc/inflate_wave.h)."""
import os, struct, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build()
from metamlst_amd.engine import Engine, MlstError
import test_inflate as ti

def first_block_info(raw):
    b = int.from_bytes(raw[:8].ljust(8, b"\0"), "little")
    last, typ = b & 1, (b >> 1) & 3
    info = {"last": last, "type": typ}
    if typ == 2:
        info.update(nlen=((b >> 3) & 31) + 257, ndist=((b >> 8) & 31) + 1, ncode=((b >> 13) & 15) + 4)
    return info

rng = np.random.default_rng(5)
far = bytes(rng.integers(0, 256, 300, dtype=np.uint8))
extra = [far + bytes(rng.integers(65, 70, 32300, dtype=np.uint8)) + far, bytes(rng.integers(0, 4, 65000, dtype=np.uint8)),
         b"".join(bytes([k]) * (k + 1) for k in range(256)) * 2]
eng = Engine(0)
names = ["empty", "A", "abc", "random", "fastq", "zeros", "acgt+fq", "far", "lowent", "runs"]
for level, strategy in [(0, 0), (1, 0), (6, 0), (9, 0), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)]:
    for name, data in zip(names, ti.payloads() + extra):
        data = data[:65280]
        if not data:
            continue
        raw = ti.deflate(data, level, strategy)
        for pad in (0, 3):
            blk = ti._bgzf_raw(raw, data)
            buf = (ti._bgzf_raw(b"\x03\x00", b"") * 0) + blk
            try:
                got = eng.inflate_bgzf(blk if pad == 0 else blk)
                ok = got == data
                msg = "ok" if ok else "MISMATCH at %d of %d" % (next((i for i in range(min(len(got), len(data))) if got[i] != data[i]), -1), len(data))
            except MlstError as e:
                msg = "ERROR " + str(e)[-40:]
            if msg != "ok" or pad == 0:
                print(level, strategy, name, len(data), len(raw), first_block_info(raw), msg, flush=True)
            break
