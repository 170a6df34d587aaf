"""One sample typed on several GPUs of one node (north_star / BASELINE configs[3]: "FASTQ sharded per GPU, RCCL gather").

One process per GPU.  `cli type --gpus N` starts the N ranks itself (torch.distributed.run, before the parent touches
a GPU); every rank walks the FASTQ file, cuts it into the same record-aligned chunks and submits the chunks whose
number is its rank modulo N, with the read index of the chunk's first record as index base (so that first-seen order,
Q6, is that of the file).  Everything that crosses reads is additive: one all-reduce of the pass-1
statistics, the same allele choice on every rank, one all-reduce of the pileup counts (metamlst_amd/dist.py), and
rank 0 writes the .nfo / --log files, byte for byte what one GPU writes.  The reference has no counterpart (one process,
metamlst.py:96-130 reads one BAM)."""
from __future__ import annotations

import os
import socket
import subprocess
import sys

import numpy as np


def spawn_ranks(n_gpus: int, cmd: list[str]) -> int:
    """Start `cmd` n_gpus times with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (what torch.distributed.run exports),
    wait for all of them, return the first non-zero exit code (the others are ended when one rank fails).  The caller has
    not initialised the GPU: the ranks are ordinary children."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen(cmd, env=env))
    rc = 0
    import time
    while procs:
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in procs:
                    q.terminate()
        time.sleep(0.05)
    return rc


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


def submit_fastq_shard(eng, paths: list[str], rank: int, world: int, chunk_bytes: int) -> int:
    """Submit this rank's chunks of the sample's FASTQ files; returns the number of reads the whole sample holds."""
    from .fastq import text_chunks
    n_seen, k = 0, 0
    for path in paths:
        for chunk in text_chunks(path, chunk_bytes):
            n = int(np.count_nonzero(np.frombuffer(chunk, np.uint8) == 10))
            if len(chunk) and chunk[-1] != 10:
                n += 1                                    # last line of the file without a newline
            n //= 4
            if k % world == rank:
                eng.set_read_index_base(n_seen)
                eng.submit_fastq(chunk, paired=False)
            n_seen += n
            k += 1
    return n_seen


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
