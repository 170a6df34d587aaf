"""Multi-GPU typing: one process per GPU, reads sharded by contiguous index range, no
data-path collective.  Every cross-read quantity of the path is additive, so
two small all-reduces make the result independent of the GPU count:

  1. after pass 1:  SUM over ranks of {sum_score, n_hits, locus_read_len_sum, counters},
                    MIN over ranks of locus_first_read          -> every rank picks the same alleles
  2. after pass 2:  SUM of the pileup counts                    -> rank 0 runs the host tail
                    (columns of the chosen alleles only: the host-driven form lays them out on the host, the
                    streamed form on the device -- StreamedShard)

torch.distributed does the plumbing: backend "nccl" (= RCCL over xGMI) on device tensors that
the engine fills / reads through mlst_export_stats_device / mlst_import_stats_device, or
"gloo" on host tensors in the CPU tests.  Message sizes are KB..MB, i.e. latency bound.
The reference has no counterpart (single process); the integer sums make 1 vs N GPUs bit identical.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

INT64_MAX = np.iinfo(np.int64).max


def shard_range(n_units: int, rank: int, world: int, pair: bool = False) -> tuple[int, int]:
    """Contiguous read-index range of `rank`; mates (2k, 2k+1) stay together when pair=True."""
    q = n_units // 2 if pair else n_units
    lo = (q * rank) // world
    hi = (q * (rank + 1)) // world
    return (2 * lo, 2 * hi) if pair else (lo, hi)


class DeviceStatsPort:
    """Adapter between metamlst_amd.engine.Engine and torch tensors on its GPU.  Exchange tensors are allocated
    once and reused (a typing pass is ~1 ms; per-step allocations would show)."""

    def __init__(self, engine, device: torch.device):
        self.engine, self.device = engine, device
        self._cache: dict = {}

    def buffer(self, name: str, n: int, dtype) -> torch.Tensor:
        t = self._cache.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.zeros(max(1, n), dtype=dtype, device=self.device)
            self._cache[name] = t
        return t[:max(1, n)]

    def consensus_from_counts(self, t_counts: torch.Tensor, n_cols: int) -> bytes:
        torch.cuda.synchronize(self.device)
        return self.engine.consensus_from_counts_device(t_counts.data_ptr(), n_cols)

    def flat_sizes(self):
        return self.engine.flat_sizes()

    def export_stats(self, t_sum: torch.Tensor, t_min: torch.Tensor):
        self.engine.export_stats_device(t_sum.data_ptr(), t_min.data_ptr())

    def import_stats(self, t_sum: torch.Tensor, t_min: torch.Tensor):
        torch.cuda.synchronize(self.device)
        self.engine.import_stats_device(t_sum.data_ptr(), t_min.data_ptr())

    def pileup_into(self, chosen: list[int], t_counts: torch.Tensor) -> int:
        return self.engine.pileup_device(chosen, t_counts.data_ptr())


def allreduce_stats(port, device: torch.device, group=None) -> None:
    """Collective 1: make every rank's pass-1 statistics the whole-job statistics."""
    n_sum, n_min = port.flat_sizes()
    if hasattr(port, "buffer"):
        t_sum, t_min = port.buffer("sum", n_sum, torch.int64), port.buffer("min", max(1, n_min), torch.int64)
    else:
        t_sum = torch.empty(n_sum, dtype=torch.int64, device=device)
        t_min = torch.empty(max(1, n_min), dtype=torch.int64, device=device)
    port.export_stats(t_sum, t_min)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t_sum, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(t_min, op=dist.ReduceOp.MIN, group=group)
    port.import_stats(t_sum, t_min)


def allreduce_pileup(port, chosen: list[int], n_cols: int, device: torch.device, group=None) -> np.ndarray:
    """Collective 2: whole-job pileup counts uint32[n_cols, 4] (returned on the host)."""
    t = torch.zeros(max(1, n_cols) * 4, dtype=torch.int32, device=device)
    port.pileup_into(chosen, t)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy().view(np.uint32).reshape(-1, 4)[:n_cols]


def allreduce_consensus(port: DeviceStatsPort, index, chosen: list[int], n_cols: int, device: torch.device, group=None) -> dict[int, bytes]:
    """Collective 2, GPU form: all-reduce the pileup counts on the device and apply the majority rule there
    (mlst_consensus_from_counts_device); returns {allele idx: consensus bytes} on every rank."""
    t = port.buffer("counts", max(1, n_cols) * 4, torch.int32)
    port.pileup_into(chosen, t)                # zeroes the buffer, then counts
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    letters = port.consensus_from_counts(t, n_cols)
    out, at = {}, 0
    for a in chosen:
        L = int(index.off[a + 1] - index.off[a])
        out[int(a)] = letters[at:at + L]
        at += L
    return out


def split_counts(index, chosen: list[int], counts: np.ndarray) -> dict[int, np.ndarray]:
    out, at = {}, 0
    for a in chosen:
        L = int(index.off[a + 1] - index.off[a])
        out[int(a)] = counts[at:at + L]
        at += L
    return out


def allreduce_sum_with_min_slots(t_all: torch.Tensor, n_sum: int, n_min: int, t_min: torch.Tensor, group=None) -> None:
    """One all-reduce(SUM) that also delivers an element-wise MIN.  t_all = [n_sum additive values | world x n_min slots];
    the caller has zeroed the slots and written its own vector into slot `rank`.  After the sum every slot holds its
    rank's vector (the others added zeros), so the minimum over the slots is the MIN over ranks.  In place; t_min gets
    the minimum.  Works on host tensors (gloo) and device tensors (RCCL) alike."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if dist.is_initialized() and world >= 1:
        dist.all_reduce(t_all, op=dist.ReduceOp.SUM, group=group)
    torch.amin(t_all[n_sum:n_sum + world * n_min].view(world, n_min), dim=0, out=t_min)


class StreamedShard:
    """One engine of a rank with everything of a step queued on ONE torch stream: pass 1, export of the statistics,
    the two all-reduces (RCCL orders a collective against the current stream), import, allele choice + pileup, the
    all-reduce of the pileup counts, consensus and the copies to the host.  The host synchronises once per step, in
    fetch().  The engine library and torch share one HIP runtime in the process (the library binds to the
    libamdhip64.so.7 torch has already loaded), so the engine can run on a torch stream (mlst_set_stream).

    The pileup counts travel in the COMPACT layout of mlst_typing_choose_pileup_compact: slots only for the loci with a
    chosen allele (after the first exchange every rank chooses the same alleles, so every rank derives the same layout on
    its device; cfg3: 140 of 1,050 loci, 1.1 MB instead of 8.4 MB per step).  The size of a collective is a host decision
    taken before the device knows the need, so the capacity comes from the steps before (the first step takes the full
    layout; then 1.5 x the largest need of the last eight steps, the same number on every rank because the need is a
    function of the global choice).  A step that does not fit says so in fetch(): every rank then repeats the step's
    second half with the full layout -- results are those of the fixed layout either way."""

    HISTORY = 8

    def __init__(self, engine, device: torch.device, group=None, force_collectives: bool = False, compact: bool | None = None):
        self.engine, self.device, self.group = engine, device, group
        self.force = force_collectives           # tests: issue the collectives even in a group of one
        # the engine's own stream, known to torch: collectives are ordered against it, the engine keeps its CU partition
        # (mlst_set_cu_partition) and its hipGraphs
        self.stream = torch.cuda.ExternalStream(engine.own_stream(), device=device)
        n_sum, n_min = engine.flat_sizes()
        self.n_sum, self.n_min = n_sum, max(1, n_min)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if self.world > 1 and getattr(engine, "depth_cap", 0):
            raise ValueError("the depth-capped pile-up (mlst_set_depth_cap) orders records by read index over the whole sample; "
                             "a rank of a sharded sample holds its own reads' records only: type capped samples on one engine")
        # ONE all-reduce(SUM) per statistics exchange: [additive part | world x n_min slots].  Every rank writes its
        # first-read vector into its own slot (the other slots are zero), so after the sum every rank holds every rank's
        # vector and takes the minimum locally -- a second collective (MIN) costs more host time than two tiny kernels.
        self.t_all = torch.zeros(max(1, n_sum) + self.world * self.n_min, dtype=torch.int64, device=device)
        self.t_sum = self.t_all[:max(1, n_sum)]
        self.t_slots = self.t_all[max(1, n_sum):].view(self.world, self.n_min)
        self.t_min = torch.zeros(self.n_min, dtype=torch.int64, device=device)
        self.total_cols = max(1, engine.typing_total_cols())
        self.t_counts = torch.zeros(self.total_cols * 4, dtype=torch.int32, device=device)
        if compact is None:
            compact = os.environ.get("MLST_COMPACT_EXCHANGE", "1") != "0"
        self.compact = compact
        self.cap_cols = self.total_cols          # columns of the next step's counts exchange
        self.needs: list[int] = []               # columns the last steps needed
        self.repeats = 0                         # steps whose second half ran twice (capacity too small)
        # The statistics travel compact as well (round 4): before the exchange no rank knows where the others have hits, so
        # the layout is a host decision again -- the loci that had hits in the last HISTORY steps (whole-job figures, the same
        # on every rank).  One more value rides along: this rank's hits OUTSIDE those loci; if its sum over the ranks is not
        # zero the step's statistics exchange is repeated in the fixed layout (every rank sees the same sum).  cfg3: 140 of
        # 1,050 loci, 0.7 MB instead of 5.1 MB per step.
        self.hit_sets: list[frozenset] = []      # hit loci of the last steps
        self.listed: tuple = ()                  # loci of the current compact layout (sorted)
        self.sel = None                          # positions of their values in the fixed layout (device)
        self.stats_repeats = 0
        self.stats_bytes_last = int(self.t_all.numel()) * 8
        self.t_first = torch.zeros(self.n_min, dtype=torch.int64, device=device)      # this rank's first-read vector
        self.t_imp = torch.zeros_like(self.t_sum)
        self._last = None
        torch.cuda.synchronize(device)
        engine.set_stream(0)                     # (back on its own stream, should a caller have moved it)

    def rebind(self):
        """The engine's stream was created anew (mlst_set_cu_partition): take the new one."""
        torch.cuda.synchronize(self.device)
        self.stream = torch.cuda.ExternalStream(self.engine.own_stream(), device=self.device)

    def _set_listed(self, loci) -> None:
        """Compact statistics layout for the given loci: [sum_score of their alleles | n_hits of their alleles | their
        length sums | the counters | hits outside | world x first-read slots of these loci]."""
        loci = tuple(sorted(int(l) for l in loci))
        if loci == self.listed and self.sel is not None:
            return
        idx = self.engine.index
        nA, nL = int(idx.n_alleles), int(idx.n_loci)
        if not loci:
            self.listed, self.sel = (), None
            return
        la = np.asarray(loci, np.int64)
        alle = np.concatenate([np.arange(int(idx.locus_begin[l]), int(idx.locus_begin[l]) + int(idx.locus_count[l]), dtype=np.int64) for l in loci])
        n_cnt = self.n_sum - 2 * nA - nL
        sel = np.concatenate([alle, nA + alle, 2 * nA + la, 2 * nA + nL + np.arange(n_cnt, dtype=np.int64)])
        self.listed = loci
        self.n_alle_listed = int(len(alle))
        self.sel = torch.from_numpy(sel).to(self.device)
        self.sel_loci = torch.from_numpy(la).to(self.device)
        self.t_c = torch.zeros(len(sel) + 1 + self.world * len(loci), dtype=torch.int64, device=self.device)

    def _exchange_stats(self):
        """Statistics of this rank -> whole-job statistics in the engine (on self.stream)."""
        e = self.engine
        nA = int(e.index.n_alleles)
        use_compact = self.compact and self.sel is not None
        self._stats_compact = use_compact
        if not use_compact:
            self.t_slots.zero_()
            e.export_stats_device_async(self.t_sum.data_ptr(), self.t_slots[self.rank].data_ptr())
            allreduce_sum_with_min_slots(self.t_all, max(1, self.n_sum), self.n_min, self.t_min, self.group)
            e.import_stats_device_async(self.t_sum.data_ptr(), self.t_min.data_ptr())
            self.stats_bytes_last = int(self.t_all.numel()) * 8
            return
        k, nl, ns = int(self.sel.numel()), len(self.listed), self.n_alle_listed
        tc = self.t_c
        e.export_stats_device_async(self.t_sum.data_ptr(), self.t_first.data_ptr())
        torch.index_select(self.t_sum, 0, self.sel, out=tc[:k])
        tc[k] = self.t_sum[nA:2 * nA].sum() - tc[ns:2 * ns].sum()            # hits outside the listed loci
        tc[k + 1:].zero_()
        tc[k + 1 + self.rank * nl:k + 1 + (self.rank + 1) * nl] = self.t_first.index_select(0, self.sel_loci)
        dist.all_reduce(tc, op=dist.ReduceOp.SUM, group=self.group)
        self._flag_at = k
        self.t_imp.copy_(self.t_sum)                                          # (t_sum keeps this rank's own figures for a repeat)
        self.t_imp.index_copy_(0, self.sel, tc[:k])
        self.t_min.copy_(self.t_first)
        self.t_min.index_copy_(0, self.sel_loci, torch.amin(tc[k + 1:].view(self.world, nl), dim=0))
        e.import_stats_device_async(self.t_imp.data_ptr(), self.t_min.data_ptr())
        self.stats_bytes_last = int(tc.numel()) * 8

    def _multi(self) -> bool:
        return dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.force)

    def _second_half(self, penalty: int, mincov: int, multi: bool):
        e = self.engine
        if multi and self.compact:
            e.typing_choose_pileup_compact(penalty, self.t_counts.data_ptr(), self.cap_cols)
            dist.all_reduce(self.t_counts[:self.cap_cols * 4], op=dist.ReduceOp.SUM, group=self.group)
            e.typing_finish_compact(mincov, "N", self.t_counts.data_ptr())
        else:
            e.typing_choose_pileup(penalty, self.t_counts.data_ptr())
            if multi:
                dist.all_reduce(self.t_counts, op=dist.ReduceOp.SUM, group=self.group)
            e.typing_finish(mincov, "N", self.t_counts.data_ptr())

    def enqueue(self, submit_fn, penalty: int = 100, mincov: int = 1):
        """submit_fn() queues pass 1 on the engine (reset_sample + submit_*); everything else follows here."""
        e = self.engine
        multi = self._multi()
        with torch.cuda.stream(self.stream):
            submit_fn()
            if multi:
                self._exchange_stats()
            self._second_half(penalty, mincov, multi)
        self._last = (penalty, mincov, multi)

    def next_capacity(self, need: int) -> int:
        """Capacity of the next counts exchange, from the needs seen: 1.5 x the largest of the last HISTORY steps, in
        steps of 1,024 columns, at most the fixed layout."""
        self.needs = (self.needs + [int(need)])[-self.HISTORY:]
        want = (max(self.needs) * 3 // 2 + 2047) // 1024 * 1024
        return max(1024, min(self.total_cols, want))

    def fetch(self, per_allele: bool = True):
        """-> (SampleStats, {locus: chosen allele idx}, {allele idx: consensus bytes}); whole-job values on every rank."""
        res = self.engine.typing_fetch(per_allele)
        penalty, mincov, multi = self._last
        # (the flag is read with a plain copy behind the step's synchronisation: a pinned buffer filled by a copy queued on the
        # engine's stream outlives that stream when the engine moves to another CU share, and freeing it then took the process down)
        if multi and getattr(self, "_stats_compact", False) and int(self.t_c[self._flag_at].item()) != 0:
            # some rank had hits at a locus outside the compact layout (every rank reads the same sum): the step's exchanges
            # again, statistics in the fixed layout from this rank's own figures (still in t_sum)
            self.stats_repeats += 1
            with torch.cuda.stream(self.stream):
                self.t_slots.zero_()
                self.t_slots[self.rank].copy_(self.t_first)
                allreduce_sum_with_min_slots(self.t_all, max(1, self.n_sum), self.n_min, self.t_min, self.group)
                self.engine.import_stats_device_async(self.t_sum.data_ptr(), self.t_min.data_ptr())
                self.cap_cols = self.total_cols
                self._second_half(penalty, mincov, multi)
            res = self.engine.typing_fetch(per_allele)
        if multi and self.compact:
            # the next step's statistics layout: the loci with hits in the last steps (whole-job values: the same on every rank)
            hit = frozenset(np.nonzero(res[0].locus_len_sum)[0].tolist())
            self.hit_sets = (self.hit_sets + [hit])[-self.HISTORY:]
            self._set_listed(frozenset().union(*self.hit_sets))
        if multi and self.compact:
            need, over = self.engine.typing_compact_info()
            if over:                             # every rank sees the same flag: all of them repeat, in the same order
                self.repeats += 1
                self.cap_cols = self.total_cols
                with torch.cuda.stream(self.stream):
                    self._second_half(penalty, mincov, multi)
                res = self.engine.typing_fetch(per_allele)
                need, over = self.engine.typing_compact_info()
                assert not over
            self.cap_cols = self.next_capacity(need)
        return res

    def close(self):
        self.engine.set_stream(0)
