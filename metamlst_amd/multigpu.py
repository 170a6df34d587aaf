"""One sample typed on several GPUs of one node (north_star / BASELINE configs[3]: "FASTQ sharded per GPU, RCCL gather").

One process per GPU.  `cli type --gpus N` starts the N ranks itself (before the parent touches a GPU); rank r reads
the r-th byte range of the FASTQ file, resynchronised on a record boundary (whole BGZF blocks for bgzip input), with a
read-index base that keeps first-seen order (Q6) that of the file (submit_fastq_shard).  Everything that crosses reads is additive: one all-reduce of the pass-1
statistics, the same allele choice on every rank, one all-reduce of the pileup counts (metamlst_amd/dist.py), and
rank 0 writes the .nfo / --log files, byte for byte what one GPU writes.  The reference has no counterpart (one process,
metamlst.py:96-130 reads one BAM)."""
from __future__ import annotations

import os
import socket
import subprocess
import sys

import numpy as np


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n_gpus: int, cmd: list[str], attempts: int = 3) -> int:
    """Start `cmd` n_gpus times with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (what torch.distributed.run exports),
    wait for all of them, return the first non-zero exit code.  When one rank fails the others are ended (SIGTERM, then
    SIGKILL after a grace period); the same happens when this process is interrupted, so no rank outlives its parent.
    The rendezvous port is picked by binding port 0 and closing it again, which another job can win in between: a run
    whose ranks all fail within a few seconds is started again on a new port (at most `attempts` times).  The caller
    has not initialised the GPU: the ranks are ordinary children."""
    import time
    rc = 0
    for attempt in range(attempts):
        port = _free_port()
        procs, t0, rc = [], time.time(), 0
        try:
            for r in range(n_gpus):
                env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                           HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                           MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
                procs.append(subprocess.Popen(cmd, env=env))
            live = list(procs)
            while live:
                for p in list(live):
                    code = p.poll()
                    if code is None:
                        continue
                    live.remove(p)
                    if code != 0 and rc == 0:
                        rc = code
                        for q in live:
                            q.terminate()
                time.sleep(0.05)
        finally:
            _reap(procs)
        bind_race = rc != 0 and time.time() - t0 < 20 and os.environ.get("MLST_SPAWN_RETRY", "1") != "0" and _port_taken(port)
        if not bind_race:
            break
    return rc


def _port_taken(port: int) -> bool:
    s = socket.socket()
    try:
        s.bind(("127.0.0.1", port))
        return False
    except OSError:
        return True
    finally:
        s.close()


def _reap(procs, grace: float = 10.0) -> None:
    """terminate -> wait -> kill whatever is still running"""
    import time
    alive = [p for p in procs if p.poll() is None]
    for p in alive:
        p.terminate()
    t_end = time.time() + grace
    for p in alive:
        try:
            p.wait(timeout=max(0.1, t_end - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()


def launch_ranks(n_gpus: int, argv: list[str]) -> int:
    """Start `python -m metamlst_amd.cli <argv>` as n_gpus ranks."""
    return spawn_ranks(n_gpus, [sys.executable, "-m", "metamlst_amd.cli"] + argv)


def init_from_env():
    """-> (rank, world, torch device).  MLST_BACKEND=gloo and MLST_ONE_GPU=1 put every rank on device 0 with gloo
    collectives (the 2-rank test on a one-GPU box); the default is RCCL with one GPU per rank."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if os.environ.get("MLST_ONE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("MLST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, device


ORDER_SHIFT = 40      # read-index bases of the shards: (file number * ranks + rank) << 40 -- see submit_fastq_shard


def submit_fastq_shard(eng, paths: list[str], rank: int, world: int, chunk_bytes: int, paired: bool = False) -> None:
    """Submit this rank's share of the sample's FASTQ files.

    Plain FASTQ: rank r reads only the byte range [size r / N, size (r + 1) / N) of every file, resynchronised on a
    record boundary (fastq.text_chunks(start, end)): host reads and PCIe traffic are 1 / N of the file per rank.
    bgzip'd FASTQ: the same by whole BGZF blocks (fastq.bgzf_range_plan); the inflated text is resynchronised on the GPU side
    of the boundary (mlst_submit_fastq_bgzf's resync flags).
    Mate files (pairs are matched by record number): rank 0 walks the two files once and broadcasts the byte offsets of the
    chunk pairs (fastq.pair_cuts); every rank reads the chunks whose number is its rank modulo N by offset.
    gzip (one deflate stream, no random access): every rank walks the file and submits the chunks whose number is its rank
    modulo N -- the wall time of the decompression is that of one rank doing it alone either way, only host CPU is spent N
    times; bgzip the file to shard it by blocks.

    Read-index bases are ORDER KEYS, not indices: what the typing needs from a read index is the first-seen order of loci
    (Q6, metamlst.py:244 iterates a dict filled in BAM order); a shard's base only has to be larger than every index
    of the shards before it.  Range r of file f gets (f * N + r) << 40, chunk k of a walked file k << 36 -- no rank has
    to know how many records the others hold."""
    from .fastq import is_bgzf, pair_chunks, pair_cuts, prefetch, read_pair_cut, text_chunks
    if paired:
        if len(paths) != 2:
            raise ValueError("paired input is two files of mates")
        import torch.distributed as dist
        plain = not any(p.endswith(".gz") or is_bgzf(p) for p in paths)
        if plain and world > 1 and dist.is_initialized():
            # rank 0 walks the two files once (the cuts must fall after the same record number in both: lines are counted)
            # and hands the byte offsets out; every rank then reads only its own chunks (round 3: every rank walked both files)
            box = [pair_cuts(paths[0], paths[1], chunk_bytes // 2) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            mine = [(k, cut) for k, cut in enumerate(box[0]) if k % world == rank]
            for k, (c1, c2) in prefetch(((k, read_pair_cut(paths[0], paths[1], cut)) for k, cut in mine)):
                eng.set_read_index_base(k << 36)
                eng.submit_fastq_pair(c1, c2)
            return
        for k, (c1, c2) in enumerate(prefetch(pair_chunks(paths[0], paths[1], chunk_bytes // 2, reuse=True))):
            if k % world == rank:
                eng.set_read_index_base(k << 36)
                eng.submit_fastq_pair(c1, c2)
        return
    k = 0
    for f, path in enumerate(paths):
        if is_bgzf(path):
            size = os.path.getsize(path)
            eng.set_read_index_base((f * world + rank) << ORDER_SHIFT)
            eng.submit_fastq_bgzf_range(path, size * rank // world, size * (rank + 1) // world if rank + 1 < world else size, chunk_bytes)
        elif path.endswith(".gz"):
            for chunk in prefetch(text_chunks(path, chunk_bytes)):
                if k % world == rank:
                    eng.set_read_index_base(k << 36)
                    eng.submit_fastq(chunk, paired=False)
                k += 1
        else:
            size = os.path.getsize(path)
            lo, hi = size * rank // world, (size * (rank + 1) // world if rank + 1 < world else None)
            eng.set_read_index_base((f * world + rank) << ORDER_SHIFT)
            for chunk in prefetch(text_chunks(path, chunk_bytes, lo, hi, reuse=True)):      # (buffers go round inside the walk; none is handed on)
                eng.submit_fastq(chunk, paired=False)


def type_sharded(eng, idx, database, targs, rank: int, world: int, device, file_name: str, out_dir: str | None, log_path: str | None,
                 sample_path: str):
    """The exchanges + the host tail after every rank has submitted its shard.  -> results (rank 0) / None."""
    import torch.distributed as dist
    from .dist import DeviceStatsPort, allreduce_consensus, allreduce_stats
    from .typing import log_table, type_sample
    port = DeviceStatsPort(eng, device)
    allreduce_stats(port, device)                 # every rank now holds the whole sample's statistics
    st = eng.stats()

    def consensus_fn(chosen):
        n_cols = sum(int(idx.off[a + 1] - idx.off[a]) for a in chosen)
        return allreduce_consensus(port, idx, chosen, n_cols, device)

    if rank == 0 and log_path:
        with open(log_path, "w", newline="") as f:
            f.write(log_table(idx, st, targs, sample_path))
    # every rank walks the same plan (the choice is a pure function of the reduced statistics): the all-reduce of the
    # pileup counts inside consensus_fn is collective; only rank 0 writes
    res = type_sample(idx, st, None, database, file_name, targs, out_dir=out_dir if rank == 0 else None, consensus_fn=consensus_fn)
    dist.barrier()
    return res if rank == 0 else None


def deal_samples(sizes: list[int], world: int) -> list[int]:
    """Whole samples -> ranks: largest first, each to the rank with the least bytes so far (ties: lowest rank).  A pure
    function of the sizes, so every rank computes the same deal."""
    load = [0] * world
    owner = [0] * len(sizes)
    for i in sorted(range(len(sizes)), key=lambda k: (-sizes[k], k)):
        r = min(range(world), key=lambda q: (load[q], q))
        owner[i] = r
        load[r] += sizes[i]
    return owner


def type_many_samples(engines, idx, database, targs, samples: list[list[str]], rank: int, world: int, out_dir: str, log: bool,
                      chunk_bytes: int, printer=None, timing: dict | None = None) -> int:
    """Multi-sample mode (BASELINE configs[3]: "RCCL gather of per-species ST tables"; the reference's real use is many
    samples into one folder, one metamlst.py run each, metamlst-merge.py:93-107 reads the folder).  Whole samples are
    dealt to the ranks -- no collective on the data path --, every rank sends its samples through the pipelined typing
    loop on its GPU (metamlst_amd/pipeline.py: `engines` take turns, a sample's allele choice, pile-up and consensus run
    on the device behind its pass 1, the host writes sample k's line while the GPU works on the samples behind it), and
    rank 0 gathers the .nfo lines (and --log tables) and writes them, sample by sample in the order given: byte for
    byte what one run per sample writes."""
    import time
    from .cli import submit_sample_files
    from .pipeline import TypingPipeline
    from .typing import log_table, sample_name, type_sample
    t_begin = time.perf_counter()
    from . import fastq as _fq
    from .engine import pinned_array
    _fq.set_buffer_allocator(pinned_array)      # file chunks are read into page-locked buffers (one DMA transfer each)
    sizes = [sum(os.path.getsize(f) for f in files) for files in samples]
    owner = deal_samples(sizes, world)
    jobs = [(i, files) for i, files in enumerate(samples) if owner[i] == rank]
    pipe = TypingPipeline(engines if isinstance(engines, (list, tuple)) else [engines], penalty=targs.penalty, feed_threads=True)
    # CU shares: HALVES of the device here (the resident 50 M-read batches of bench.py's headline gain most from quarters, 2.6 ms of
    # kernels per step; a file sample of a few million reads is a chain of short kernels whose latency counts, and on a quarter of
    # the CUs each takes up to four times as long -- 16 bgzip'd samples of 2 M reads on six engines: 1 / 2 / 4 shares = 215 / 300 / 275
    # Mreads/s, profiles/round5/inflate.md 6)
    parts = int(os.environ.get("MLST_CU_PARTITIONS", "0")) or min(2, TypingPipeline.default_partitions(pipe.depth))
    if parts > 1:
        pipe.place(parts)
        pipe.stagger_s = 0.75e-3

    # every engine is fed by a thread of its own (pipeline.feed_threads): the file reads, the copies and -- for bgzip'd files --
    # the inflate the library waits for overlap across the engines instead of holding up the loop one after the other
    def feed(e, job):
        submit_sample_files(e, job[1], False, chunk_bytes)

    # The per-allele table (metamlst.py:133-151 over every allele with a hit: `cel`) is display -- the closest-allele listing of
    # metamlst.py:213-230 and the --log table; the .nfo line needs the device's choice and consensus only.  A quiet run
    # without --log skips it, and with it the 12 bytes per allele of every sample's fetch (cfg3: 0.3 s of Python and 4 MB per
    # sample against ~7 ms for a sample of a million reads otherwise).
    from . import db as mdb
    show = printer is not None
    cache = mdb.DbCache(database.conn, idx)

    def tail(job, st, chosen, letters):
        i, files = job
        name = sample_name(files[0])
        res = type_sample(idx, st, None, database, name, targs, out_dir=None, fast=not show, cache=cache, typed=(chosen, letters))
        return {"i": i, "name": name, "nfo": [r.nfo_line for r in res if r.written],
                "log": log_table(idx, st, targs, files[0]) if log else None, "results": res if show else None}

    t_run = time.perf_counter()
    try:
        mine = pipe.run(jobs, feed, tail, per_allele=show or log)
    finally:
        pipe.stop_feeders()      # (also when a feeder or the tail raised: the other feeders end with their current sample)
    if timing is not None:      # (what a run pays once -- database look-up tables, CU shares -- and what it pays per sample)
        timing["prologue_s"] = t_run - t_begin
        timing["samples_s"] = time.perf_counter() - t_run
        timing["host_ms"] = dict(pipe.host_ms)
        timing["cu_partitions"] = pipe.partitions
    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=0)
        if rank != 0:
            dist.barrier()
            dist.destroy_process_group()
            return 0
        mine = [x for part in gathered for x in part]
    if not os.path.isdir(out_dir):
        os.mkdir(out_dir)
    for x in sorted(mine, key=lambda x: x["i"]):
        if x["log"] is not None:
            with open(out_dir + "/" + x["name"] + "_" + str(int(time.time())) + ".out", "w", newline="") as f:
                f.write(x["log"])
        if x["nfo"]:
            with open(out_dir + "/" + x["name"] + ".nfo", "a", newline="") as f:      # append, as metamlst.py:284 does
                f.write("".join(x["nfo"]))
        if printer and x["results"] is not None:
            print("Sample " + x["name"])
            printer(x["results"])
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0
