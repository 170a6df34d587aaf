"""Host-side pieces of the multi-GPU paths that need no GPU: the rank launcher (`bench.py --gpus N` / `cli type --gpus N` start
their ranks with it before touching a device) and the FASTQ chunk dealing of `cli type --gpus N`."""
import os
import subprocess
import sys

import numpy as np

from metamlst_amd import multigpu
from metamlst_amd.fastq import text_chunks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spawn_ranks_exports_the_rendezvous_environment(tmp_path):
    out = tmp_path / "r"
    code = ("import os; open(r'%s' + os.environ['RANK'], 'w').write(' '.join(os.environ[k] for k in "
            "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')))" % str(out))
    assert multigpu.spawn_ranks(3, [sys.executable, "-c", code]) == 0
    seen = [open(str(out) + str(r)).read().split() for r in range(3)]
    assert [s[0] for s in seen] == ["0", "1", "2"] and all(s[1] == s[0] and s[2] == "3" and s[3] == "127.0.0.1" for s in seen)
    assert len({s[4] for s in seen}) == 1 and int(seen[0][4]) > 0


def test_spawn_ranks_returns_the_first_failure_and_ends_the_others():
    code = "import os, sys, time; r = int(os.environ['RANK']); sys.exit(7) if r == 1 else time.sleep(60)"
    rc = multigpu.spawn_ranks(3, [sys.executable, "-c", code])
    assert rc == 7


class FakeEngine:
    def __init__(self):
        self.calls = []

    def set_read_index_base(self, b):
        self.base = b

    def submit_fastq(self, chunk, paired=False):
        self.calls.append((self.base, bytes(chunk).count(b"\n") // 4))


def test_fastq_chunks_are_dealt_round_robin_with_global_read_indices(tmp_path):
    rng = np.random.default_rng(3)
    p = tmp_path / "s.fastq"
    n = 5000
    with open(p, "wb") as f:
        for k in range(n):
            L = int(rng.integers(30, 151))
            f.write(b"@r%d\n%s\n+\n%s\n" % (k, b"A" * L, b"I" * L))
    chunk = 64 << 10
    world = 3
    engines = [FakeEngine() for _ in range(world)]
    totals = [multigpu.submit_fastq_shard(engines[r], [str(p)], r, world, chunk) for r in range(world)]
    assert totals == [n] * world                                  # every rank walks the whole file and agrees on its size
    calls = sorted(c for e in engines for c in e.calls)
    assert sum(c[1] for c in calls) == n                          # every read submitted exactly once ...
    at = 0
    for base, cnt in calls:                                       # ... with the index it has in the file
        assert base == at
        at += cnt
    n_chunks = len(list(text_chunks(str(p), chunk)))
    assert [len(e.calls) for e in engines] == [len(range(r, n_chunks, world)) for r in range(world)]
