"""N>1 path on CPU: two processes, gloo, reads sharded by index range, the two all-reduces of
metamlst_amd/dist.py; the reduced result must equal the single-process result bit for bit."""
import os
import sys
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fixtures as fx
import oracle_lib
from metamlst_amd.dist import allreduce_pileup, allreduce_stats, allreduce_sum_with_min_slots, shard_range, split_counts
from metamlst_amd.engine import MLST_CNT_N
from metamlst_amd.typing import SampleStats, pick_alleles_fast

INT64_MAX = np.iinfo(np.int64).max


class OraclePort:
    """Host-tensor port with the layout of mlst_export_stats_device, backed by the oracle."""

    def __init__(self, orc, idx):
        self.orc, self.idx, self.st = orc, idx, None

    def flat_sizes(self):
        return 2 * self.idx.n_alleles + self.idx.n_loci + MLST_CNT_N, self.idx.n_loci

    def export_stats(self, t_sum, t_min):
        s = self.orc.stats()
        nA, nL = self.idx.n_alleles, self.idx.n_loci
        flat = np.concatenate([s.sum_score, s.n_hits.astype(np.int64), s.locus_len_sum.astype(np.int64), s.counters.astype(np.int64)])
        t_sum.copy_(torch.from_numpy(flat))
        first = np.where(s.locus_first > np.uint64(INT64_MAX), np.uint64(INT64_MAX), s.locus_first).astype(np.int64)
        t_min[:nL].copy_(torch.from_numpy(first))

    def import_stats(self, t_sum, t_min):
        nA, nL = self.idx.n_alleles, self.idx.n_loci
        f = t_sum.numpy()
        first = t_min.numpy()[:nL].astype(np.uint64)
        first[t_min.numpy()[:nL] == INT64_MAX] = np.uint64(0xFFFFFFFFFFFFFFFF)
        self.st = SampleStats(f[:nA].copy(), f[nA:2 * nA].astype(np.uint32), f[2 * nA:2 * nA + nL].astype(np.uint64), first,
                              f[2 * nA + nL:].astype(np.uint64))

    def pileup_into(self, chosen, t_counts):
        c = self.orc.pileup(chosen)
        flat = np.concatenate([c[a] for a in chosen]).astype(np.int32).reshape(-1)
        t_counts[:flat.size].copy_(torch.from_numpy(flat))
        return flat.size // 4


def worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    db, idx = fx.ecoli_small(40)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 3, n_reads=6001, genome=80_000)
    lo, hi = shard_range(len(off) - 1, rank, world)
    o = off[lo:hi + 1]
    orc = oracle_lib.Oracle(idx)
    orc.submit_reads(fb[int(o[0]):int(o[-1])], fq[int(o[0]):int(o[-1])], o - o[0], read_base=lo)
    p = OraclePort(orc, idx)
    # the single-collective form used by StreamedShard (SUM buffer with per-rank MIN slots) against two collectives
    n_sum, n_min = p.flat_sizes()
    t_all = torch.zeros(n_sum + world * n_min, dtype=torch.int64)
    p.export_stats(t_all[:n_sum], t_all[n_sum + rank * n_min:n_sum + (rank + 1) * n_min])
    t_min1 = torch.zeros(n_min, dtype=torch.int64)
    allreduce_sum_with_min_slots(t_all, n_sum, n_min, t_min1)
    t_sum2, t_min2 = torch.zeros(n_sum, dtype=torch.int64), torch.zeros(n_min, dtype=torch.int64)
    p.export_stats(t_sum2, t_min2)
    dist.all_reduce(t_sum2, op=dist.ReduceOp.SUM)
    dist.all_reduce(t_min2, op=dist.ReduceOp.MIN)
    assert torch.equal(t_all[:n_sum], t_sum2) and torch.equal(t_min1, t_min2)
    allreduce_stats(p, torch.device("cpu"))
    chosen = sorted(pick_alleles_fast(idx, p.st, 100).values())
    n_cols = sum(int(idx.off[a + 1] - idx.off[a]) for a in chosen)
    counts = allreduce_pileup(p, chosen, n_cols, torch.device("cpu"))
    np.savez(os.path.join(tmp, "r%d.npz" % rank), sum=p.st.sum_score, hits=p.st.n_hits, len=p.st.locus_len_sum, first=p.st.locus_first,
             cnt=p.st.counters, counts=counts, chosen=np.array(chosen))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_equals_single_process():
    tmp = tempfile.mkdtemp()
    port = 29500 + (os.getpid() % 500)
    mp.spawn(worker, args=(2, port, tmp), nprocs=2, join=True)
    db, idx = fx.ecoli_small(40)
    fb, fq, off, _, _ = fx.isolate_reads(db, "ecoli", 3, n_reads=6001, genome=80_000)
    orc = oracle_lib.Oracle(idx)
    orc.submit_reads(fb, fq, off)
    s = orc.stats()
    chosen = sorted(pick_alleles_fast(idx, s, 100).values())
    c = orc.pileup(chosen)
    whole = np.concatenate([c[a] for a in chosen])
    for r in range(2):
        z = np.load(os.path.join(tmp, "r%d.npz" % r))
        assert np.array_equal(z["sum"], s.sum_score) and np.array_equal(z["hits"], s.n_hits)
        assert np.array_equal(z["len"], s.locus_len_sum) and np.array_equal(z["first"], s.locus_first)
        assert all(int(z["cnt"][k]) == int(s.counters[k]) for k in (0, 1, 2, 4, 5, 6))
        assert list(z["chosen"]) == chosen and np.array_equal(z["counts"], whole)


def test_shard_range_keeps_mates_together_and_covers_everything():
    for n, w in ((10, 3), (7, 2), (1000001, 8), (4, 8)):
        got = [shard_range(n, r, w) for r in range(w)]
        assert got[0][0] == 0 and got[-1][1] == n and all(got[i][1] == got[i + 1][0] for i in range(w - 1))
    for n, w in ((10, 3), (1000, 8)):
        got = [shard_range(n, r, w, pair=True) for r in range(w)]
        assert all(lo % 2 == 0 and hi % 2 == 0 for lo, hi in got) and got[-1][1] == n
