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
        self._last = None
        torch.cuda.synchronize(device)
        engine.set_stream(0)                     # (back on its own stream, should a caller have moved it)

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
                self.t_slots.zero_()
                e.export_stats_device_async(self.t_sum.data_ptr(), self.t_slots[self.rank].data_ptr())
                allreduce_sum_with_min_slots(self.t_all, max(1, self.n_sum), self.n_min, self.t_min, self.group)
                e.import_stats_device_async(self.t_sum.data_ptr(), self.t_min.data_ptr())
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
