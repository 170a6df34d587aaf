"""The pipelined typing loop: many samples through a few engines on one GPU.

The reference's real use is many samples into one folder, one `metamlst.py` run per sample
(/root/reference/metamlst-merge.py:93-107 reads that folder); a run there is strictly serial -- BAM scan, allele choice,
pileup, .nfo line (metamlst.py:96-289).  Here a sample's whole device side is queued without a host wait
(`mlst_typing_enqueue`: pass 1, allele choice, pile-up and consensus on the engine's stream), so while the host writes the
.nfo line of sample k the GPU works on the samples behind it:

    engine e finishes sample k          -> typing_wait (synchronise only)
    queue sample k + depth on engine e  -> feed(engine, job) + typing_enqueue        (the engine is free again)
    copy k's results out of their slot  -> typing_fetch(waited=True)                 (two pinned slots written in turn)
    host tail of sample k               -> tail(job, stats, chosen, letters)

Until round 3 this loop lived in bench.py only (missing 3); `cli type folder/`, `multigpu.type_many_samples`
and `bench.py` all run it from here now.  With `partitions` > 1 every engine's stream is restricted to its own share of the
CUs (`mlst_set_cu_partition`, whole XCDs): the launch sequences of the engines then run side by side instead of taking turns
(+ 4-6 % on a 2.6 ms step of cfg3; a step of well under a millisecond of kernels gains nothing -- DESIGN.md 4a).
"""
from __future__ import annotations

import time
from collections import deque
from typing import Callable, Iterable

_END = object()


class TypingPipeline:
    """`depth` engines on one device taking turns on a stream of samples.

    engines     -- Engine objects with the reference loaded (the host index is built once per process, every further
                   engine only uploads).
    shards      -- optional dist.StreamedShard per engine (N > 1 ranks: the step's two all-reduces ride on the engine's
                   stream; results then come from shard.fetch()).
    stagger_s   -- the first submissions go out this far apart when the engines have their own CU shares: started in the
                   same instant the launch sequences stay in step for a round or two (all in k_route, then all in
                   k_route_probe ...); 0.5-1 ms apart they mix from the start.
    feed_threads-- feed(engine, job) + typing_enqueue of every engine run on a thread of their own (one per engine).  For
                   feeds that BLOCK -- file reads, host-to-device copies that wait, the inflate of a bgzip'd file, which the
                   library waits for -- so that the engines' inputs overlap; resident batches (bench.py) are queued in
                   ~40 us and stay on the caller's thread.  Ignored with `shards` (collectives are issued in one order).
    """

    def __init__(self, engines: list, penalty: int = 100, mincov: int = 1, none_char: str = "N", shards: list | None = None,
                 stagger_s: float = 0.0, feed_threads: bool = False):
        if not engines:
            raise ValueError("a pipeline needs at least one engine")
        self.engines = list(engines)
        self.depth = len(self.engines)
        self.penalty, self.mincov, self.none_char = penalty, mincov, none_char
        self.shards = shards
        self.stagger_s = stagger_s
        self.feed_threads = bool(feed_threads) and shards is None
        self._pool = None
        self._pending: dict = {}
        self.partitions = 1
        self.host_ms = {"submit": 0.0, "wait_device": 0.0, "tail": 0.0}

    # ---- placement
    def place(self, partitions: int) -> None:
        """Engine k on share k mod `partitions` of the CUs (1 = every engine on the whole device)."""
        partitions = max(1, min(int(partitions), self.depth))
        for k, e in enumerate(self.engines):
            e.synchronize()
            e.set_cu_partition(k % partitions if partitions > 1 else 0, partitions)
        self.partitions = partitions

    @staticmethod
    def default_partitions(depth: int) -> int:
        """The largest of 1, 2, 4, 8 shares (whole XCDs) that leaves every share at least one engine."""
        return max(p for p in (1, 2, 4, 8) if p <= depth)

    # ---- one sample
    def _launch(self, k: int, job, feed: Callable) -> None:
        t0 = time.perf_counter()
        e = self.engines[k % self.depth]
        if self.shards is not None:
            self.shards[k % self.depth].enqueue(lambda: self._pass1(e, job, feed), penalty=self.penalty)
        elif self.feed_threads:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(self.depth, thread_name_prefix="feed")
            self._pending[k % self.depth] = self._pool.submit(self._feed_and_enqueue, e, job, feed)
        else:
            self._feed_and_enqueue(e, job, feed)
        self.host_ms["submit"] += (time.perf_counter() - t0) * 1e3

    def _feed_and_enqueue(self, e, job, feed: Callable) -> None:
        self._pass1(e, job, feed)
        e.typing_enqueue(penalty=self.penalty, mincov=self.mincov, none_char=self.none_char)

    def _fed(self, slot: int) -> None:
        """the engine of this slot has its sample queued (a feeder thread's exception surfaces here)"""
        fut = self._pending.pop(slot, None)
        if fut is not None:
            fut.result()

    @staticmethod
    def _pass1(e, job, feed: Callable) -> None:
        e.reset_sample()
        feed(e, job)

    def run(self, jobs: Iterable, feed: Callable, tail: Callable, per_allele: bool = True) -> list:
        """Every job through the pipeline, in order.  feed(engine, job) queues the job's reads into the engine (no wait needed);
        tail(job, stats, chosen, letters) is the host part (gap-fill, accuracy gate, .nfo line, ST call) and its return
        value is collected.  per_allele=False leaves the per-allele arrays out of the fetch (4 MB per step on the 315 k
        alleles of cfg3) when the tail does not read them."""
        it = iter(jobs)
        inflight: deque = deque()
        results = []
        k = 0
        for _ in range(self.depth):
            job = next(it, _END)
            if job is _END:
                break
            if k and self.stagger_s and self.partitions > 1:
                time.sleep(self.stagger_s)
            self._launch(k, job, feed)
            inflight.append((k, job))
            k += 1
        while inflight:
            kk, job = inflight.popleft()
            e = self.engines[kk % self.depth]
            t0 = time.perf_counter()
            if self.shards is not None:
                got = self.shards[kk % self.depth].fetch(per_allele)      # (a counts exchange that did not fit is repeated in there)
                self.host_ms["wait_device"] += (time.perf_counter() - t0) * 1e3
                nxt = next(it, _END)
                if nxt is not _END:
                    self._launch(k, nxt, feed)
                    inflight.append((k, nxt))
                    k += 1
            else:
                # wait, queue the engine's next sample, THEN copy the finished one's results out of their pinned slot: on its
                # own share of the CUs an engine idles from the end of a sample to the submission of the next
                self._fed(kk % self.depth)
                e.typing_wait()
                self.host_ms["wait_device"] += (time.perf_counter() - t0) * 1e3
                # A handle is not thread-safe (include/mlst.h): with feeder threads the finished sample's results leave their
                # pinned slot BEFORE the engine goes to its feeder for the next sample (the fetch on this thread ran
                # beside reset / submit / enqueue on the feeder's -- disjoint fields in the normal case, a race in the error and
                # profiling paths).  Without feeder threads the order stays: queue first, copy then.
                got = e.typing_fetch(per_allele, waited=True) if self.feed_threads else None
                nxt = next(it, _END)
                if nxt is not _END:
                    self._launch(k, nxt, feed)
                    inflight.append((k, nxt))
                    k += 1
                if got is None:
                    got = e.typing_fetch(per_allele, waited=True)
            t1 = time.perf_counter()
            results.append(tail(job, *got))
            self.host_ms["tail"] += (time.perf_counter() - t1) * 1e3
        return results

    def synchronize(self) -> None:
        for slot in list(self._pending):
            self._fed(slot)
        for e in self.engines:
            e.synchronize()

    def stop_feeders(self) -> None:
        """end the feeder threads (the engines stay as they are)"""
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None

    def close(self) -> None:
        self.stop_feeders()
        for e in self.engines:
            e.close()


def make_engines(idx, device: int = 0, depth: int = 4, params=None) -> list:
    """`depth` engines on one device with `idx` loaded."""
    from .engine import Engine
    engines = [Engine(device, params) for _ in range(depth)]
    for e in engines:
        e.load_reference(idx)
    return engines
