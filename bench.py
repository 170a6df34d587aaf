#!/usr/bin/env python3
"""Headline benchmark: Mreads/s aligned + ST-typed (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W

Workload of the driver line = BASELINE.json configs[2] (cfg3), the configuration the metric is quoted on: the full
database (synthetic stand-in for metamlstDB_2022, which the reference downloads at run time: 150 species x 7 loci x
300 alleles) and a 50 M-read mixed metagenome of 20 genomes per GPU.  One "step" = one complete typing pass over one
resident batch: seed sieve -> exact seeds -> extension against every allele of the hit loci -> hit accumulation ->
allele choice -> pileup -> consensus -> .nfo lines -> allele match + ST call of every species in the sample.  cfg2
(10 M reads of one E. coli-like isolate, 7 loci x 1430 alleles) runs afterwards at N=1 and is reported as a secondary
block of the same JSON line, together with the end-to-end rates (FASTQ text / bgzip on the host -> ST) and the CPU
oracle on a bounded sample of the cfg3 workload.

--gpus N without a launcher starts the N ranks itself (torch.distributed.run, before this process touches the GPU);
under `python -m torch.distributed.run ... bench.py --gpus N` it reads RANK / LOCAL_RANK / WORLD_SIZE.  At N > 1 every
rank holds its own 50 M-read shard of the metagenome (weak scaling), the two all-reduces of metamlst_amd/dist.py run over
RCCL on the engine's stream, and rank 0 runs the host tail.

Several engines per GPU (--pipeline, default 4: four HIP streams, four sets of sample state) work in turn, each on
its OWN resident batch (distinct reads: no step re-reads what the step before it read), so that the host part of
step k overlaps the kernels of the following steps.  Every step is still one complete pass.  The timed region is a
series of blocks of exactly K steps, each bracketed by barrier + synchronize, repeated until at least --min-seconds have
been timed; the block with the median duration is the one reported (`ms_per_step`, `value`).

Reads are synthesised on the GPU before the timed region and are resident in HBM in the packed format of
SURVEY.md 8(d) (2-bit bases + Phred rows); the timed region contains no H2D copy of reads.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_BASES = 40      # 2-bit bases of a 150 bp read, rounded to the 10-word row the sieve streams
ALG_BYTES_SURVEY = 188    # 38 B 2-bit bases + 150 B Phred per 150 bp read (reported beside, never used for frac)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_GINSTR = 930.0  # measured simple-op issue rate of the whole chip, G wave-instructions/s (profiles/round1/valu_rate2.txt)
PROFILE_DIR = os.path.join(ROOT, "profiles", "round5")
_KEEP_ALIVE: list = []


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--host-profile", default=None, help="diagnostics: cProfile of the host loop over 200 extra (untimed) steps, written to this file")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="timed blocks of --steps steps are repeated until this much has been timed")
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg2", "skewed"], help="workload of the headline line")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU and batch (default: 50 M for cfg3, 10 M for cfg2)")
    ap.add_argument("--species", type=int, default=150)
    ap.add_argument("--alleles", type=int, default=0, help="alleles per locus (default: 300 for cfg3, 1430 for cfg2)")
    ap.add_argument("--genomes", type=int, default=20)
    ap.add_argument("--genome-size", type=int, default=0, help="isolate genome size (default: 2 Mb for cfg3, 4.6 Mb for cfg2)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline time (0 = skip)")
    ap.add_argument("--pipeline", type=int, default=4,
                    help="engines per GPU, each with its own resident batch: the GPU already works on the next steps while the host "
                         "types step k (every step is still a complete pass); 1 = strictly serial steps")
    ap.add_argument("--cu-partitions", type=int, default=0,
                    help="shares of the CUs the engines are spread over (engine k on share k mod n; mlst_set_cu_partition): "
                         "0 = one share per engine, 1 = every engine on the whole device (rounds 1-3)")
    ap.add_argument("--stagger-ms", type=float, default=None,
                    help="time between the first submissions of a block (default: 0.75 ms when the engines have their own CU shares, else 0)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the cfg2 block and the end-to-end rates")
    ap.add_argument("--e2e-reads", type=int, default=48_000_000, help="reads of the end-to-end (host FASTQ text -> ST) measurements (cfg2's legs: its 10 M); 24 M until round 5")
    ap.add_argument("--calibrate", action="store_true",
                    help="before the timed region copy the Phred rows 3x with torch (a known-size wide coalesced stream) "
                         "so a rocprofv3 --pmc FETCH_SIZE pass of this command can be calibrated")
    return ap.parse_args()


def self_launch(args) -> int:
    """--gpus N without a launcher: start the N ranks as children of a process that has not touched the GPU."""
    import __graft_entry__ as ge
    ge.build()                                       # once, here: the ranks find fresh libraries
    from metamlst_amd.multigpu import spawn_ranks
    return spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])


def clean_env_for_children():
    """Children started by build() (make, hipcc) must not inherit a profiler's preload: they would initialise the GPU."""
    env = dict(os.environ)
    for k in list(env):
        if k in ("LD_PRELOAD",) or k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS", "ROCTRACER")):
            env.pop(k)
    return env


def build_workload(name, args, eng_factory, torch, device, rank, n_batches, tmp):
    from metamlst_amd import db as mdb
    from metamlst_amd import synth
    from metamlst_amd.index import load_index
    w = types.SimpleNamespace()      # database + planted truth + engines + resident batches of one configuration on this rank
    w.name = name
    t0 = time.time()
    if name == "cfg3":
        alleles = args.alleles or 300
        w.reads = args.reads or 50_000_000
        w.genome_size = args.genome_size or 2_000_000
        w.sdb = synth.make_full_db(os.path.join(tmp, "full.db"), n_species=args.species, alleles_per_locus=alleles, n_profiles=200)
        w.plan = synth.metagenome_plan(w.sdb, args.genomes)
        w.planted = {sp: st_row + 1 for sp, _, st_row in w.plan}
        w.label = ("cfg3: %d x %d bp SE reads per GPU, mixed metagenome of %d genomes (%.1f Mb each, log-normal abundances), synthetic DB-full "
                   "%d species x 7 loci x %d alleles (stand-in for metamlstDB_2022, which is not available offline)"
                   % (w.reads, args.read_len, len(w.plan), w.genome_size / 1e6, args.species, alleles))
    elif name in ("skewed", "skewed_nodup"):
        # a PubMLST-shaped database (VERDICT r3 item 1 / missing 4; the real one, metaMLST_functions.py:39-57 with the schema of
        # metamlst-index.py:62-65, has loci with tens to thousands of alleles): alleles per locus log-uniform 10 ... 10,000
        w.reads = args.reads or 20_000_000
        w.genome_size = args.genome_size or 2_000_000
        w.sdb = synth.make_skewed_db(os.path.join(tmp, name + ".db"), n_species=6, hi=10_000, n_duplicates=3 if name == "skewed" else 0)
        w.plan = synth.metagenome_plan(w.sdb, 6)
        w.planted = {sp: st_row + 1 for sp, _, st_row in w.plan}
        counts = sorted(w.sdb.n_alleles.values())
        w.label = ("skewed: %d x %d bp SE reads, metagenome of %d genomes, PubMLST-shaped synthetic database: %d loci with %d ... %d alleles "
                   "(median %d; synth.make_skewed_db)" % (w.reads, args.read_len, len(w.plan), len(counts), counts[0], counts[-1], counts[len(counts) // 2]))
    else:
        alleles = args.alleles or 1430
        w.reads = args.reads or 10_000_000
        w.genome_size = args.genome_size or 4_600_000
        w.sdb = synth.make_ecoli_db(os.path.join(tmp, "ecoli.db"), alleles_per_locus=alleles, n_profiles=5000)
        w.plan = [("ecoli", 1.0, 11)]
        w.planted = {"ecoli": 12}
        w.label = ("cfg2: %d x %d bp SE reads per GPU, one E. coli-like isolate (%.1f Mb), synthetic DB 7 loci x %d alleles"
                   % (w.reads, args.read_len, w.genome_size / 1e6, alleles))
    w.t_db = time.time() - t0
    t0 = time.time()
    w.idx = load_index(w.sdb.path)
    w.database = mdb.metaMLST_db(w.sdb.path)
    w.t_index_host = time.time() - t0
    t0 = time.time()
    w.engines = [eng_factory() for _ in range(n_batches)]
    w.eng_factory = eng_factory
    ref_cache = w.sdb.path + ".mlstref"         # the built host index on disk, next to the database (mlst_set_reference_cache)
    for e in w.engines:
        e.load_reference(w.idx, cache_path=ref_cache)      # the host index is built once per process (cached inside the library) and stored
    w.t_index_dev = time.time() - t0
    w.t_cached = None
    if name == "cfg3" and rank == 0 and getattr(args, "measure_second_command", False):            # what a SECOND command on this database pays: index arrays and host index from their files
        from metamlst_amd.engine import load_library
        t1 = time.time(); idx2 = load_index(w.sdb.path); t_idx = time.time() - t1
        load_library().mlst_release_index_cache()
        e2 = eng_factory()
        t1 = time.time(); e2.load_reference(idx2, cache_path=ref_cache); t_ref = time.time() - t1
        t1 = time.time(); e2.load_reference(idx2, cache_path=ref_cache); t_up = time.time() - t1      # (in-process now: allocation + upload only)
        e2.close()
        w.t_cached = {"load_index_from_mlstidx_s": round(t_idx, 3), "mlst_load_reference_from_mlstref_s": round(t_ref, 3),
                      "of_which_allocation_and_upload_s": round(t_up, 3), "mlstref_bytes": os.path.getsize(ref_cache) if os.path.exists(ref_cache) else 0}
    from metamlst_amd.engine import load_library as _ll
    _ll().mlst_set_reference_cache(None)
    t0 = time.time()
    w.batches, w.genomes = [], {}
    for b in range(n_batches):
        if name in ("cfg3", "skewed", "skewed_nodup"):
            packed, qrows, lens, wpr, qstride, n_total = synth.make_metagenome_gpu(
                w.engines[0], torch, device, w.sdb, w.plan, w.reads, w.genome_size, seed=7 + 100 * rank + b, read_len=args.read_len, genomes=w.genomes)
        else:
            sp, _, st_row = w.plan[0]
            if sp not in w.genomes:
                w.genomes[sp], _ = synth.make_genome(w.sdb, sp, w.sdb.profiles[sp][st_row], size=w.genome_size)
            packed, qrows, lens, wpr, qstride = synth.synth_reads_gpu(w.engines[0], torch, device, w.genomes[sp], w.reads, args.read_len,
                                                                      seed=synth.SEED + 1000 * rank + b)
            n_total = w.reads
        w.batches.append((packed, qrows, lens, n_total))
        w.wpr, w.qstride = wpr, qstride
    w.n_reads = min(b[3] for b in w.batches)    # cfg3: the shares are floored, every batch holds the same count
    assert all(b[3] == w.n_reads for b in w.batches)
    w.t_reads = time.time() - t0
    return w


def run_workload(w, args, torch, dist, device, rank, world, backend):
    """Warm-up, isolated kernel times, serial latency, timed blocks.  -> dict (rank 0) / None."""
    from metamlst_amd import db as mdb
    from metamlst_amd.dist import DeviceStatsPort, StreamedShard, allreduce_consensus, allreduce_stats
    from metamlst_amd.merge import EngineMatcher, SpeciesSession, parse_nfo_line
    from metamlst_amd.typing import type_sample
    idx, database, engines = w.idx, w.database, w.engines
    depth = len(engines)
    eng = engines[0]
    # engine k runs on share k mod n_parts of the CUs; by default the largest of 1, 2, 4, 8 shares (whole XCDs) that leaves every
    # share at least one engine
    n_parts = max(1, min(args.cu_partitions, depth)) if args.cu_partitions else max(p2 for p2 in (1, 2, 4, 8) if p2 <= depth)

    placed = {"on": False}

    def place(on: bool):
        """Engines on their own shares of the CUs (the timed blocks) or all on the whole device (isolated launches, serial
        steps: the figures behind `roofline` are those of a kernel that has the GPU to itself)."""
        for k, e in enumerate(engines):
            e.synchronize()
            e.set_cu_partition(k % n_parts if on else 0, n_parts if on else 1)
        placed["on"] = bool(on)

    place(False)          # isolated launches and serial steps see the whole device at every N (the SCALE lines' `roofline` is BENCH's)
    # N > 1: one process group per engine slot.  torch's NCCL backend runs a group's collectives on that group's own
    # communicator and internal stream: with every engine on the default group, engine k + 1's statistics all-reduce would
    # queue behind engine k's counts all-reduce, which waits for k's allele choice and pile-up (head-of-line blocking).
    groups = [dist.new_group(ranks=list(range(world))) for _ in engines] if world > 1 else None
    _KEEP_ALIVE.append(groups)      # (until dist.destroy_process_group(): a group dropped while the backend lives has taken the process down)
    shards = [StreamedShard(e, device, group=g) for e, g in zip(engines, groups)] if world > 1 else None
    ports = [DeviceStatsPort(e, device) for e in engines] if world > 1 else None
    mode = {"streamed": world > 1}
    matcher = EngineMatcher(eng, idx)
    # merge-run prologue (metamlst-merge.py:119-142) happens once per run of many samples: untimed setup
    cache = mdb.DbCache(database.conn, idx)
    sessions = {sp: SpeciesSession(database, sp, 5, matcher, cache) for sp in w.planted} if rank == 0 else {}
    host_ms = {"submit": 0.0, "wait_device": 0.0, "typing": 0.0, "st_call": 0.0}

    def submit(k):
        """Everything of step k that runs on the GPU, queued without waiting, on engine k % depth and its batch."""
        t_a = time.perf_counter()
        e = engines[k % depth]
        packed, qrows, lens, n = w.batches[k % depth]

        def pass1():
            e.reset_sample()
            e.set_read_index_base(rank * n)
            e.submit_packed_device(packed.data_ptr(), qrows.data_ptr(), lens.data_ptr(), n, w.wpr, w.qstride)

        if world > 1 and mode["streamed"]:
            shards[k % depth].enqueue(pass1, penalty=100)
        elif world > 1:
            pass1()                                   # host-driven exchange: the collectives follow in finish()
        else:
            pass1()
            e.typing_enqueue(penalty=100)
        host_ms["submit"] += (time.perf_counter() - t_a) * 1e3

    def wait(k):
        """Step k's device results on the host (streamed / single GPU): from here on the step's engine is free again."""
        e = engines[k % depth]
        t_b = time.perf_counter()
        # (the streamed form fetches through its shard: a counts exchange that did not fit its buffer is repeated there)
        full = bool(mode.get("full_fetch"))      # the per-allele arrays (4 MB on cfg3) only where they are compared
        got = shards[k % depth].fetch(full) if (world > 1 and mode["streamed"]) else e.typing_fetch(full)
        host_ms["wait_device"] += (time.perf_counter() - t_b) * 1e3
        return got

    def tail(k, got=None):
        """The host part of step k: .nfo lines (gap-fill, accuracy gate) and ST calls.  Host-driven exchange (got is None):
        the collectives and the wait are in here too."""
        e = engines[k % depth]
        t_b = time.perf_counter()
        if got is None:
            port = ports[k % depth]
            allreduce_stats(port, device)
            st = e.stats()
            t_c = time.perf_counter()
            host_ms["wait_device"] += (t_c - t_b) * 1e3

            def consensus_fn(chosen):
                n_cols = sum(int(idx.off[a + 1] - idx.off[a]) for a in chosen)
                return allreduce_consensus(port, idx, chosen, n_cols, device)

            res = type_sample(idx, st, None, database, "sample", fast=True, cache=cache, consensus_fn=consensus_fn)
        else:
            st, chosen_dev, letters_dev = got
            t_c = t_b
            res = type_sample(idx, st, None, database, "sample", fast=True, cache=cache, typed=(chosen_dev, letters_dev))
        t_d = time.perf_counter()
        out = {}
        if rank == 0:
            for r in res:
                if r.written:       # .nfo line -> allele match + ST call (per-sample body of metamlst-merge.py:144-240)
                    organism, (bacteriumLine, sampleRecord) = parse_nfo_line(r.nfo_line)
                    if organism in sessions:
                        out[organism] = sessions[organism].add_sample(bacteriumLine, sampleRecord)
        t_e = time.perf_counter()
        host_ms["typing"] += (t_d - t_c) * 1e3
        host_ms["st_call"] += (t_e - t_d) * 1e3
        return out, st

    def finish(k):
        host_driven = world > 1 and not mode["streamed"]
        return tail(k, None if host_driven else wait(k))

    calls = {}
    # the loop itself is the product's: metamlst_amd/pipeline.py (cli type folder/, multigpu.type_many_samples run the same one)
    from metamlst_amd.pipeline import TypingPipeline
    pipe = TypingPipeline(engines, penalty=100)

    def run(n_steps):
        """K steps through the `depth` engines (TypingPipeline.run: an engine is free again as soon as its step's results are on
        the host, so the next step for it is queued BEFORE the host tail of the one just fetched; steps are fetched in the
        order they were queued -- with N > 1 the ranks must issue their collectives in one order anyway).  The host-driven
        exchange (N > 1 when the streamed form is switched off) keeps its own loop: there the engine is busy until its tail
        has run."""
        last = None
        calls.clear()
        host_driven = world > 1 and not mode["streamed"]
        if host_driven:
            ahead = depth - 1
            for k in range(min(ahead, n_steps)):
                submit(k)
            for k in range(n_steps):
                if k + ahead < n_steps:
                    submit(k + ahead)
                last = tail(k)
                calls[k % depth] = last[0]
            return last
        # the first submissions of a block 0.75 ms apart when every engine has its own share of the CUs (2.62-2.66 -> 2.56-2.57 ms
        # per step in blocks of 20, the waiting included; 1.5 ms apart: 2.62-2.64; 2.5 ms: 2.73)
        pipe.shards = shards if world > 1 else None
        pipe.stagger_s = (args.stagger_ms if args.stagger_ms is not None else (0.75 if n_parts > 1 else 0.0)) * 1e-3
        pipe.partitions = n_parts if placed["on"] else 1
        for k_ in pipe.host_ms:
            pipe.host_ms[k_] = 0.0

        def feed(e, k):
            packed, qrows, lens, n = w.batches[k % depth]
            e.set_read_index_base(rank * n)
            e.submit_packed_device(packed.data_ptr(), qrows.data_ptr(), lens.data_ptr(), n, w.wpr, w.qstride)

        def tail_k(k, *got):
            out = tail(k, got)
            calls[k % depth] = out[0]
            return out

        outs = pipe.run(range(n_steps), feed, tail_k, per_allele=bool(mode.get("full_fetch")))
        host_ms["submit"] += pipe.host_ms["submit"]
        host_ms["wait_device"] += pipe.host_ms["wait_device"]
        return outs[-1] if outs else None

    def fence():
        for e in engines:
            e.synchronize()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    # N > 1: one step both ways before anything is timed: the streamed step (kernels and RCCL collectives ordered on a
    # torch stream, one host synchronisation) must reproduce the host-driven exchange bit for bit.  Every rank runs both
    # sequences to the end (no exception handling around collectives: a failure ends the run loudly); the verdicts are
    # combined with one all-reduce and every rank takes the same form.
    if world > 1:
        mode["full_fetch"] = True
        mode["streamed"] = False
        submit(0)
        ref_out, ref_st = finish(0)
        mode["streamed"] = True
        submit(0)
        got_out, got_st = finish(0)
        ok = int(np.array_equal(ref_st.sum_score, got_st.sum_score) and np.array_equal(ref_st.n_hits, got_st.n_hits)
                 and np.array_equal(ref_st.locus_first, got_st.locus_first) and ref_out == got_out)
        if os.environ.get("MLST_BENCH_HOST_COLLECTIVES"):      # test switch: take the host-driven form
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int64, device=device)
        torch.cuda.synchronize(device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        mode["streamed"] = bool(int(flag.item()))
        mode["full_fetch"] = False
    # untimed priming: every engine sees its launch sequence often enough for the hipGraph of it to be built (that happens
    # on the second identical submission) and replayed once; then the W warm-up steps proper
    for e_i in range(depth):
        for _ in range(3):
            submit(e_i)
            finish(e_i)
    fence()
    run(args.warmup)
    fence()
    # strictly serial steps on one engine, untimed, with HIP events around the kernel groups (on the engine's stream):
    # every kernel has the GPU to itself, which is what `rocprofv3 --kernel-trace --stats -- python3 bench.py --pipeline 1`
    # shows as the kernel's average (profiles/round2/kernel_summary_*.md)
    from metamlst_amd.engine import KERNELS
    KNAMES = [k for k in KERNELS if k not in ("pack", "sieve_inkernel", "sieve_wg_longest")]
    engines[0].set_profiling(1)
    n_iso = 20
    per_launch = {k: [] for k in KNAMES}
    # (the next launch is queued before the host tail of the one just fetched, as in the timed loop: with the GPU idle for
    # the 1.3 ms of every host tail the same kernels ran 3-5 % slower -- clocks -- than rocprofv3 saw them in the timed blocks)
    engines[0].reset_kernel_time()
    submit(0)
    for k in range(n_iso):
        got = None if (world > 1 and not mode["streamed"]) else wait(0)
        if got is None:
            tail(0)
        for name in KNAMES:
            ms, n = engines[0].kernel_time(name)
            if n:
                per_launch[name].append(ms / n)
        engines[0].reset_kernel_time()
        if k + 1 < n_iso:
            submit(0)
        if got is not None:
            tail(0, got)
    fence()
    isolated = {k: (float(np.median(v)) if v else 0.0, 1) for k, v in per_launch.items()}      # (median launch, 1): what the rooflines use
    spread = {k: {"min": round(min(v), 4), "median": round(float(np.median(v)), 4), "max": round(max(v), 4), "n": len(v)} for k, v in per_launch.items() if v}
    engines[0].set_profiling(0)
    fence()
    t0 = time.perf_counter()
    n_serial = 5
    for k in range(n_serial):
        submit(0)
        finish(0)
    fence()
    serial_ms = (time.perf_counter() - t0) / n_serial * 1e3
    if world == 1 and n_parts > 1 and not args.cu_partitions and sum(v[0] for k_, v in isolated.items() if k_ in ("sieve", "seed", "extend", "banded_sw", "accumulate", "pileup")) < 1.0:
        n_parts = 1      # a step of well under a millisecond of kernels (cfg2: 0.35 ms) gains nothing from shares (0.342 -> 0.351 ms)
    if n_parts > 1:      # from here on every engine has its own share of the CUs; graphs are rebuilt, warm-up again
        place(True)
        if shards:
            for sh in shards:
                sh.rebind()                  # (an engine's stream is created anew with its CU mask)
        for e_i in range(depth):
            for _ in range(3):
                submit(e_i)
                finish(e_i)
        fence()
        run(args.warmup)
        fence()
    # ---- timed region: blocks of exactly K steps, repeated until min-seconds have been timed (at least 3 blocks)
    blocks, total = [], 0.0
    last = None
    while len(blocks) < 3 or total < args.min_seconds:
        for k in host_ms:
            host_ms[k] = 0.0
        fence()
        t0 = time.perf_counter()
        last = run(args.steps)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        blocks.append((dt, dict(host_ms)))
        total += dt
        if len(blocks) >= 200:
            break
    if args.host_profile and rank == 0 and world == 1:      # where the host thread spends a step (untimed, after the timed blocks)
        import cProfile
        import io
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        run(200)
        pr.disable()
        fence()
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(60)
        pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(40)
        with open(args.host_profile, "w") as fh:
            fh.write(buf.getvalue())
    if shards:
        for s_ in shards:
            s_.close()
    place(False)      # what follows (end-to-end legs, the oracle check) has the whole device again
    if rank != 0:
        return None
    order = sorted(range(len(blocks)), key=lambda i: blocks[i][0])
    dt, hm = blocks[order[len(order) // 2]]
    st_call, stats = last
    iso_launch = {k: (isolated[k][0] / max(1, isolated[k][1])) for k in isolated}
    # planted truth is checked on the last step of EVERY resident batch (each engine has its own), not only on the last one
    typed_ok = {sp: all(c.get(sp) == w.planted[sp] for c in calls.values()) for sp in w.planted}
    wrong = [c for c in calls.values() if any(c.get(sp) != w.planted[sp] for sp in w.planted)]
    if wrong:
        st_call = wrong[0]
    return {"ms_per_step": dt / args.steps * 1e3, "value": w.n_reads * world / (dt / args.steps) / 1e6,
            "blocks": len(blocks), "block_ms": [round(b[0] * 1e3, 3) for b in blocks], "timed_s": round(total, 3),
            "host_ms_per_step": {k: round(v / args.steps, 4) for k, v in hm.items()}, "serial_ms_per_step": serial_ms,
            "iso_launch_ms": iso_launch, "iso_launch_spread": spread, "stats": stats, "st_call": st_call, "typed_ok": typed_ok, "batches_checked": len(calls),
            "collectives": ("streamed on a torch stream" if mode["streamed"] else "host-driven") if world > 1 else None,
            "cu_partitions": n_parts,
            "iso_measured_on": "whole device",
            "exchange": ({"statistics_bytes": int(shards[0].stats_bytes_last), "statistics_bytes_fixed_layout": int(shards[0].t_all.numel()) * 8,
                          "statistics_loci_listed": len(shards[0].listed), "process_groups": len(groups),
                          "steps_repeated_for_statistics": sum(sh.stats_repeats for sh in shards),
                          "counts_layout": "compact" if shards[0].compact else "fixed",
                          "counts_bytes": int(shards[0].cap_cols) * 16, "counts_bytes_fixed_layout": int(shards[0].total_cols) * 16,
                          "counts_columns_needed": (shards[0].needs[-1] if shards[0].needs else None),
                          "steps_repeated_for_capacity": sum(sh.repeats for sh in shards)} if (world > 1 and mode["streamed"]) else None)}


def load_profile_json(name):
    try:
        return json.load(open(os.path.join(PROFILE_DIR, name)))
    except Exception:
        return None


def rooflines(w, res, eng):
    """`roofline` of the kernel that takes the most time (durations: HIP events on the engine's stream, kernels alone
    on the GPU), and the VALU view of k_extend."""
    iso = res["iso_launch_ms"]
    kind = eng.sieve_info()["kind"]
    stream_kernels = {"sieve_route": "k_route", "sieve_probe": "k_route_probe", "sieve_verify": "k_route_verify"} if kind == "routed" else {"sieve": "k_sieve_q"}
    cands = dict(stream_kernels)
    cands.update({"seed": "k_seed+k_retain", "extend": "k_extend", "banded_sw": "k_banded", "accumulate": "k_accumulate+k_locus", "pileup": "k_pileup"})
    dom = max(cands, key=lambda k: iso.get(k, 0.0))
    stats = res["stats"]
    n_items = int(stats.counters[5])
    pairs = n_items * (int(w.idx.n_alleles) // max(1, int(w.idx.n_loci)))
    pmc = load_profile_json("pmc_%s.json" % w.name) or {}

    n_entries = w.n_reads * (w.wpr - 1)            # seeds routed: one per 16 bases of a full-length read (4-byte entries)
    n_parked = int(stats.counters[7])              # entries that passed the LDS filter and are examined by k_route_verify
    # what each kernel NEEDS to read: the producer the 2-bit rows (40 B / 150-base read), the consumer its 4-byte entries, the
    # examination the rows of the parked entries only.  (Round 4 priced all three with reads x 40 B: 0.97 "of the roofline" for
    # k_route_verify, which moves 1.8 GB to use 0.1 GB.)  The sieve as a whole is priced once, with the rows: sieve_total.
    need = {"k_route": (w.n_reads * ALG_BYTES_BASES, "reads x %d B of 2-bit rows" % ALG_BYTES_BASES),
            "k_sieve_q": (w.n_reads * ALG_BYTES_BASES, "reads x %d B of 2-bit rows" % ALG_BYTES_BASES),
            "k_route_probe": (n_entries * 4, "%d routed entries x 4 B" % n_entries),
            "k_route_verify": (n_parked * ALG_BYTES_BASES, "%d parked entries x %d B of row" % (n_parked, ALG_BYTES_BASES))}

    def hbm_roof(key, kernel, ms=None, alg=None, what=None):
        ms = iso.get(key, 0.0) if ms is None else ms
        if alg is None:
            alg, what = need.get(kernel, (w.n_reads * ALG_BYTES_BASES, "reads x %d B of 2-bit rows" % ALG_BYTES_BASES))
        ach = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        tr = pmc.get(kernel, {}).get("hbm_bytes_per_launch")
        return {"kernel": kernel, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": int(tr * w.n_reads / pmc.get("reads_per_launch", w.n_reads)) if tr else None,
                "traffic_source": ("profiles/round5/pmc_%s.json: rocprofv3 --pmc passes of this command (profiles/pmc_round2.sh), committed -- not measured by this run" % w.name) if tr else None,
                "alg_bytes_per_launch": int(alg), "alg_bytes_are": what, "alg_bytes_per_read": ALG_BYTES_BASES, "reads_per_launch": w.n_reads, "avg_launch_ms": round(ms, 4),
                "launch_ms_spread": res.get("iso_launch_spread", {}).get(key),
                "duration_source": "HIP events on the engine's stream around each of 20 serial launches, median (= rocprofv3 --kernel-trace average of --pipeline 1)",
                "measured_on": res.get("iso_measured_on", "whole device")}

    if dom in stream_kernels:
        roof = hbm_roof(dom, stream_kernels[dom])
    else:      # a non-streaming kernel dominates: its own algorithmic bytes (arena window + result word per (item, allele) pair)
        ms = iso.get(dom, 0.0)
        alg = pairs * 48
        ach = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        roof = {"kernel": cands[dom], "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": None, "alg_bytes_per_pair": 48, "pairs_per_launch": pairs, "avg_launch_ms": round(ms, 4),
                "measured_on": res.get("iso_measured_on", "whole device"),
                "note": "VALU-bound kernel: see roofline_extend"}
    roof["dominant_by_time"] = dom
    roof["sieve_kernels"] = {v: hbm_roof(k, v) for k, v in stream_kernels.items()}
    sieve_ms = iso.get("sieve", 0.0) or sum(iso.get(k, 0.0) for k in stream_kernels)      # ("sieve" = every kernel of the sieve, k_flag_compact included)
    tot = hbm_roof(None, "sieve_total", ms=sieve_ms, alg=w.n_reads * ALG_BYTES_BASES, what="reads x %d B of 2-bit rows, once, over the sum of the sieve's kernels" % ALG_BYTES_BASES)
    tr = [v.get("traffic") for v in roof["sieve_kernels"].values()]
    tot["traffic"] = int(sum(tr)) if tr and all(t is not None for t in tr) else None
    tot["kernels"] = sorted(stream_kernels.values())
    roof["sieve_total"] = tot
    ext_ms = iso.get("extend", 0.0)
    valu = pmc.get("k_extend", {}).get("valu_wave_instr_per_launch")
    rext = {"kernel": "k_extend", "bound": "valu", "avg_launch_ms": round(ext_ms, 4), "pairs_per_launch": pairs,
            "Gcells_per_s": round(pairs * 150 / max(1e-9, ext_ms * 1e-3) / 1e9, 1), "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s"}
    if valu and ext_ms > 0:
        scale = pairs / max(1, pmc.get("k_extend", {}).get("pairs_per_launch", pairs))
        rext["achieved"] = round(valu * scale / (ext_ms * 1e-3) / 1e9, 1)
        rext["frac"] = round(rext["achieved"] / VALU_PEAK_GINSTR, 4)
        rext["valu_count_source"] = "profiles/round5/pmc_%s.json (SQ_INSTS_VALU of a rocprofv3 --pmc pass, committed); the duration is this run's" % w.name
    return roof, rext


def end_to_end(w, args, torch, device):
    """Host FASTQ text -> ST and bgzip'd FASTQ -> ST on a bounded slice of batch 0 (PCIe-inclusive; never `value`)."""
    import zlib
    import struct
    from concurrent.futures import ThreadPoolExecutor
    from metamlst_amd import synth
    from metamlst_amd import db as mdb
    from metamlst_amd.merge import EngineMatcher, SpeciesSession, parse_nfo_line
    from metamlst_amd.typing import type_sample
    eng = w.engines[0]
    packed, qrows, lens, n_total = w.batches[0]
    n = min(args.e2e_reads, n_total)
    L = args.read_len
    rec = 16 + 2 * L
    text_host = np.empty(n * rec, np.uint8)
    step = 1 << 20
    for at in range(0, n, step):
        c = min(step, n - at)
        text_host[at * rec:(at + c) * rec] = synth.resident_to_fastq_text(torch, packed, qrows, n_total, w.wpr, w.qstride, at, c, L).cpu().numpy()
    cache = mdb.DbCache(w.database.conn, w.idx)
    matcher = EngineMatcher(eng, w.idx)

    sessions = {sp: SpeciesSession(w.database, sp, 5, matcher, cache) for sp in w.planted}      # merge-run prologue: once per run, untimed

    def tail():
        st, chosen_dev, letters_dev = eng.typing_fetch()
        res = type_sample(w.idx, st, None, w.database, "sample", fast=True, cache=cache, typed=(chosen_dev, letters_dev))
        out = {}
        for r in res:
            if r.written:
                organism, (bl, sr) = parse_nfo_line(r.nfo_line)
                if organism in sessions:
                    out[organism] = sessions[organism].add_sample(bl, sr)
        return out

    chunk_reads = 1 << 20      # 1 M records = 331 MB of text per submission
    out = {"reads": n, "text_bytes": int(text_host.size)}

    def run_text():
        eng.reset_sample()
        for at in range(0, n, chunk_reads):
            c = min(chunk_reads, n - at)
            eng.submit_fastq(text_host[at * rec:(at + c) * rec])
        eng.typing_enqueue(penalty=100)
        return tail()

    run_text()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); calls = run_text(); ts.append(time.perf_counter() - t0)
    out["fastq_text_to_st"] = {"Mreads_per_s": round(n / min(ts) / 1e6, 1), "GB_per_s_of_text": round(text_host.size / min(ts) / 1e9, 2), "seconds": round(min(ts), 4),
                               "species_called": len(calls)}
    # ---- bgzip: BGZF blocks made here with zlib on the host cores (untimed), then compressed bytes -> ST
    nz = n                       # the same slice as the text leg: ~20 k BGZF blocks, three rounds of the inflate kernel's one-wave-per-block grid
    raw = text_host[:nz * rec].tobytes()

    def bgzf_block(data: bytes) -> bytes:
        c = zlib.compressobj(6, zlib.DEFLATED, -15)      # bgzip's default level (-l -1 = zlib's 6); level 1, used until round 3, inflates ~20 % slower here (more, shorter matches)
        comp = c.compress(data) + c.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp
                + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))

    with ThreadPoolExecutor(max(1, min(64, os.cpu_count() or 1))) as ex:      # zlib releases the GIL
        parts = list(ex.map(bgzf_block, [raw[at:at + 65280] for at in range(0, len(raw), 65280)]))
    comp = np.frombuffer(b"".join(parts) + bgzf_block(b""), np.uint8)
    # fed the way a file is (Engine.submit_fastq_bgzf_file: pieces of whole blocks from page-locked buffers): the copy of piece k + 1
    # and its inflate run beside the parse and pass 1 of piece k (three streams, mlst_submit_fastq_bgzf).  Pieces of 65,536 blocks
    # (one turn of k_inflate_tok2 with its four waves per CU) where the input has several of them, else of 16,384 (one turn of
    # k_inflate_tok): on 24 M reads 16,384 / 32,768 / 49,152 blocks gave 312 / 290 / 294-340 Mreads/s, on 48 M reads 356 / - / 380-390
    # with three waves per CU; 49,152 / 65,536 with four: 485 / 495 on a small database (profiles/round5/inflate.md)
    per_piece = int(os.environ.get("MLST_BENCH_BGZF_PIECE", "65536" if nz >= 40_000_000 else "16384"))
    cuts = np.concatenate([[0], np.cumsum([len(x) for x in parts])])
    pieces = [comp[int(cuts[a]):int(cuts[min(a + per_piece, len(parts))])] for a in range(0, len(parts), per_piece)]
    pieces[-1] = comp[int(cuts[(len(pieces) - 1) * per_piece]):]            # (with the end-of-file block)
    from metamlst_amd.engine import pinned_array
    pinned = []
    for pc in pieces:      # (mlst_alloc_host: a copy from pageable memory is staged by the runtime on the calling thread, 10 ms per 560 MB during which nothing else is queued)
        pb = pinned_array(int(pc.size))
        pb[:pc.size] = pc
        pinned.append(pb[:pc.size])
    pieces = pinned

    def run_bgzf():
        eng.reset_sample()
        for k, pc in enumerate(pieces):
            eng.submit_fastq_bgzf(pc, final=(k == len(pieces) - 1))
        eng.typing_enqueue(penalty=100)
        return tail()

    run_bgzf()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); calls = run_bgzf(); ts.append(time.perf_counter() - t0)
    # ---- host-packed transfer (mlst_submit_packed_host): bases + lengths over the link, the candidates' Phred rows behind them
    from metamlst_amd.engine import pack_fastq_host
    chunks_txt = [text_host[at * rec:(at + min(chunk_reads, n - at)) * rec] for at in range(0, n, chunk_reads)]

    def run_packed(prepacked=None):
        eng.reset_sample()
        for k, ct in enumerate(chunks_txt):
            pk = prepacked[k] if prepacked is not None else pack_fastq_host(ct, ((L + 31) // 32) * 32)
            eng.submit_packed_host(*pk)
        eng.typing_enqueue(penalty=100)
        return tail()

    t0 = time.perf_counter()
    pre = [pack_fastq_host(ct, ((L + 31) // 32) * 32) for ct in chunks_txt]
    t_pack = time.perf_counter() - t0
    run_packed(pre)
    tp = []
    for _ in range(3):
        t0 = time.perf_counter(); calls_p = run_packed(pre); tp.append(time.perf_counter() - t0)
    link_bytes = sum(int(p[0].nbytes) + 2 * p[3] for p in pre)
    out["packed_host_to_st"] = {"Mreads_per_s": round(n / min(tp) / 1e6, 1), "seconds": round(min(tp), 4), "bytes_over_the_link_before_the_sieve": link_bytes,
                                "species_called": len(calls_p),
                                "note": "arrays packed beforehand (mlst_pack_fastq_host, %d host threads: %.2f s = %.1f Mreads/s, %.1f GB/s of text); with the packing inside the timed loop: see fastq_text_packed_on_host_to_st"
                                        % (os.cpu_count() or 1, t_pack, n / t_pack / 1e6, text_host.size / t_pack / 1e9)}
    t0 = time.perf_counter(); calls_q = run_packed(None); t_q = time.perf_counter() - t0
    out["fastq_text_packed_on_host_to_st"] = {"Mreads_per_s": round(n / t_q / 1e6, 1), "seconds": round(t_q, 4), "species_called": len(calls_q)}
    out["bgzip_to_st"] = {"reads": nz, "Mreads_per_s": round(nz / min(ts) / 1e6, 1), "compressed_bytes": int(comp.size), "zlib_level": 6, "seconds": round(min(ts), 4),
                          "blocks_per_piece": per_piece, "pieces": len(pieces),
                          "species_called": len(calls)}
    del raw
    for key, kw in (("cli_folder_to_nfo", {}), ("cli_folder_bgzip_to_nfo", {"bgzf_parts": parts, "eof_block": bgzf_block(b"")})):
        try:
            out[key] = folder_leg(w, text_host, n, rec, **kw)
        except (OSError, MemoryError) as e:      # (no room for the sample files: the leg is skipped, the line stands)
            out[key] = {"skipped": "%s: %s" % (type(e).__name__, e)}
    del parts
    return out


def folder_leg(w, text_host, n, rec, n_files=16, reads_per_file=2 << 20, bgzf_parts=None, eof_block=b"", n_engines=6):
    """The product command on a folder of samples (`cli type folder/` = multigpu.type_many_samples: files -> reader thread -> GPU
    parser -> the pipelined loop of metamlst_amd/pipeline.py on the workload's engines -> one .nfo file per sample;
    /root/reference/metamlst-merge.py:93-107 reads that folder).  FASTQ text files in memory-backed storage, so what is
    timed is the path from the page cache on -- file reads, link, device and host tail -- not a disk."""
    import shutil
    from metamlst_amd.multigpu import type_many_samples
    from metamlst_amd.typing import TypingArgs
    reads_per_file = int(os.environ.get("MLST_FOLDER_READS", reads_per_file))      # (ad-hoc runs: larger samples)
    n_files = int(os.environ.get("MLST_FOLDER_SAMPLES", n_files))
    # the command's own number of engines (cli.py: MLST_PIPELINE_DEPTH, six) on halves of the device (multigpu.type_many_samples);
    # the headline's four stay as they are, the others are made here once
    n_engines = int(os.environ.get("MLST_PIPELINE_DEPTH", n_engines))
    if not hasattr(w, "folder_engines"):
        w.folder_engines = list(w.engines)
    while len(w.folder_engines) < n_engines:
        e = w.eng_factory()
        e.load_reference(w.idx)
        w.folder_engines.append(e)
    engines = w.folder_engines[:max(1, n_engines)]
    per = min(reads_per_file, n // n_files)
    if bgzf_parts is not None:      # whole BGZF blocks of 65,280 bytes of text: a sample = a run of them (its last record may be cut: the parser drops nothing, the reader completes it from the next block -- so samples are cut at block AND record boundaries: 65,280 x k bytes with k a multiple of rec / gcd)
        import math
        unit = rec // math.gcd(rec, 65280)                   # blocks per boundary that is also a record boundary
        blocks_per = (per * rec // 65280) // unit * unit
        per = blocks_per * 65280 // rec
    if per < 1000:
        return {"skipped": "slice too small"}
    need = n_files * per * rec * (1.3 if bgzf_parts is None else 0.4)
    shm_ok = False
    try:
        st = os.statvfs("/dev/shm")
        shm_ok = st.f_bavail * st.f_frsize > need
    except OSError:
        pass
    root = tempfile.mkdtemp(prefix="mlst_folder_", dir="/dev/shm" if shm_ok else None)
    try:
        files = []
        for k in range(n_files):
            if bgzf_parts is None:
                f = os.path.join(root, "s%02d.fastq" % k)
                text_host[k * per * rec:(k + 1) * per * rec].tofile(f)
            else:
                f = os.path.join(root, "s%02d.fastq.gz" % k)
                with open(f, "wb") as fh:
                    fh.write(b"".join(bgzf_parts[k * blocks_per:(k + 1) * blocks_per]) + eof_block)
            files.append([f])
        ts = []
        for r in range(3):
            od = os.path.join(root, "out%d" % r)
            prof = None
            if r == 2 and os.environ.get("MLST_PROFILE_FOLDER"):
                import cProfile
                prof = cProfile.Profile(); prof.enable()
            t0 = time.perf_counter()
            tm = {}
            rc = type_many_samples(engines, w.idx, w.database, TypingArgs(quiet=True), files, 0, 1, od, False, 256 << 20, timing=tm)
            ts.append((time.perf_counter() - t0, tm))
            if prof is not None:
                import pstats
                prof.disable()
                pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(28)
            assert rc == 0
        written = sorted(os.listdir(od))
        lines = sum(open(os.path.join(od, x)).read().count("\n") for x in written)
        # the same folder three times as long (links to the same files under further sample names): what is left of the first
        # samples' latency and the last samples' tail when the folder is longer
        longer = None
        if bgzf_parts is not None:
            more = list(files)
            for rep in (1, 2):
                for k in range(n_files):
                    f = os.path.join(root, "s%02d.fastq.gz" % (rep * n_files + k))
                    os.symlink(files[k][0], f)
                    more.append([f])
            best = None
            for r in range(2):
                tm2 = {}
                rc = type_many_samples(engines, w.idx, w.database, TypingArgs(quiet=True), more, 0, 1, os.path.join(root, "long%d" % r), False, 256 << 20, timing=tm2)
                assert rc == 0
                best = tm2["samples_s"] if best is None else min(best, tm2["samples_s"])
            longer = {"samples": len(more), "samples_s": round(best, 4), "ms_per_sample": round(best / len(more) * 1e3, 3), "Mreads_per_s": round(len(more) * per / best / 1e6, 1)}
    finally:
        shutil.rmtree(root, ignore_errors=True)
        for e in engines:      # (type_many_samples puts the engines on their CU shares)
            e.synchronize()
            e.set_cu_partition(0, 1)
    t, tm = min(ts, key=lambda x: x[1]["samples_s"])
    prologue = ts[0][1]["prologue_s"]      # the FIRST run's: it makes the engines' CU shares (later runs find them made), as a command does
    out = {"samples": n_files, "reads_per_sample": per, "input": "FASTQ text" if bgzf_parts is None else "bgzip (level 6), inflated on the GPU", "engines": len(engines), "cu_shares": tm.get("cu_partitions"),
           "seconds": round(tm["samples_s"] + prologue, 4),
           "prologue_s": round(prologue, 4), "samples_s": round(tm["samples_s"], 4), "ms_per_sample": round(tm["samples_s"] / n_files * 1e3, 3),
           "Mreads_per_s": round(n_files * per / tm["samples_s"] / 1e6, 1), "Mreads_per_s_with_prologue": round(n_files * per / (tm["samples_s"] + prologue) / 1e6, 1),
           "nfo_files": len(written), "species_lines": lines,
           "note": "prologue = what the command pays once (CU shares for the engines; the host tail's look-up tables are made when first asked for), taken from the "
                   "first of three runs; samples_s = the best of the three; per sample the path is %s "
                   "-- not the resident batches of the headline" % ("FASTQ text over the link (%d B/read) like fastq_text_to_st" % rec if bgzf_parts is None else "compressed bytes over the link, inflate + parse on the device like bgzip_to_st")}
    if longer is not None:
        out["folder_three_times_as_long"] = longer
    return out


def literal_leg(w, args, torch, tmp):
    """when bowtie2 + samtools are on PATH, the documented command (README.md:20) on cfg1's reads (the
    first 100 k reads of the isolate) -> BAM -> this build's --alignments path, compared with the FASTQ path.  Untimed."""
    from metamlst_amd import literal, synth
    if literal.tools() is None:
        return {"skipped": "bowtie2 / bowtie2-build / samtools not on PATH (the reference ships none of them)"}
    packed, qrows, lens, n_total = w.batches[0]
    n = min(100_000, n_total)
    fq = os.path.join(tmp, "cfg1.fastq")
    with open(fq, "wb") as f:
        f.write(synth.resident_to_fastq_text(torch, packed, qrows, n_total, w.wpr, w.qstride, 0, n, args.read_len).cpu().numpy().tobytes())
    out = literal.literal_parity(w.engines[0], w.idx, w.database, w.sdb.path, fq, threads=min(64, os.cpu_count() or 1))
    out["reads"] = n
    return out


def cpu_baseline(w, args, res):
    """The oracle (a plain CPU restatement of the path, oracle/) on a bounded slice of batch 0 of the same workload, all
    host cores; also the parity of the engine with it on that very slice."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from metamlst_amd import synth
    from metamlst_amd.typing import pick_alleles_fast
    eng = w.engines[0]
    packed, qrows, lens, n_total = w.batches[0]
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    orc = oracle_lib.Oracle(w.idx, threads=cores)
    t_build = time.perf_counter() - t0
    n_probe = min(200_000, n_total)
    b, q = synth.resident_to_host_reads(packed, qrows, n_total, w.wpr, w.qstride, 0, n_probe, args.read_len)
    fb, fq, off = synth.flatten_reads(b, q)
    orc.submit_reads(fb, fq, off)
    t1 = time.perf_counter(); orc.stats(); probe = time.perf_counter() - t1
    n_cpu = int(min(4_000_000, n_total, max(n_probe, n_probe * args.cpu_seconds / max(probe, 1e-3))))
    b, q = synth.resident_to_host_reads(packed, qrows, n_total, w.wpr, w.qstride, 0, n_cpu, args.read_len)
    fb, fq, off = synth.flatten_reads(b, q)
    orc.submit_reads(fb, fq, off)
    cpu_dt, reps = 0.0, 0
    while cpu_dt < args.cpu_seconds and reps < 64:      # repeat the bounded sample until ~cpu_seconds of CPU work
        t1 = time.perf_counter()
        so = orc.stats()
        ch = sorted(pick_alleles_fast(w.idx, so, 100).values())
        po = orc.pileup(ch)
        cpu_dt += time.perf_counter() - t1
        reps += 1
    cpu = {"value": round(n_cpu * reps / cpu_dt / 1e6, 4), "unit": "Mreads/s", "cores": cores, "kind": "port",
           "sample": "first %d reads of batch 0 of the %s workload (same database) x %d passes: oracle pass 1 + allele choice + pileup, OpenMP x %d threads, %.1f s "
                     "(index build %.1f s untimed)" % (n_cpu, w.name, reps, cores, cpu_dt, t_build)}
    eng.reset_sample()
    eng.submit_reads(fb, fq, off)
    sg = eng.stats()
    pg = eng.pileup(ch)
    same = bool(np.array_equal(sg.sum_score, so.sum_score) and np.array_equal(sg.n_hits, so.n_hits) and np.array_equal(sg.locus_first, so.locus_first)
                and all(np.array_equal(pg[a], po[a]) for a in ch))
    return cpu, same


def full_stats(w):
    """Statistics of resident batch 0 WITH the per-allele arrays (the timed loop leaves them on the device); untimed."""
    e = w.engines[0]
    packed, qrows, lens, n = w.batches[0]
    e.reset_sample()
    e.set_read_index_base(0)
    e.submit_packed_device(packed.data_ptr(), qrows.data_ptr(), lens.data_ptr(), n, w.wpr, w.qstride)
    return e.stats()


def mistyped_loci(w, stats):
    """Loci whose chosen allele (metamlst.py:133-151, 244 on the step's statistics) is not the planted one, with both alleles' hit
    counts and -- on the skewed database -- the locus it shares its seeds with (synth.make_skewed_db's near-duplicate loci)."""
    from metamlst_amd.typing import pick_alleles_fast
    idx = w.idx
    chosen = pick_alleles_fast(idx, stats, 100)
    dup = dict(w.sdb.duplicates)
    dup.update({v: k for k, v in w.sdb.duplicates.items()})
    out = []
    for sp, _, st_row in w.plan:
        for (gene, _len), al in zip(w.sdb.loci[sp], w.sdb.profiles[sp][st_row]):
            l = idx.locus_index(sp, gene)
            a = chosen.get(l)
            got = int(idx.allele_no[a]) if a is not None else None
            if got != int(al):
                lo = int(idx.locus_begin[l])
                planted_idx = lo + int(np.nonzero(idx.allele_no[lo:lo + int(idx.locus_count[l])] == int(al))[0][0])
                other = dup.get((sp, gene))
                sa, sb = idx.sequence(planted_idx), (idx.sequence(a) if a is not None else "")
                diff = [i for i in range(min(len(sa), len(sb))) if sa[i] != sb[i]]
                out.append({"locus": "%s_%s" % (sp, gene), "alleles_in_locus": int(idx.locus_count[l]), "allele_length": len(sa), "planted": int(al),
                            "hits_planted": int(stats.n_hits[planted_idx]), "sum_planted": int(stats.sum_score[planted_idx]),
                            "chosen": got, "hits_chosen": int(stats.n_hits[a]) if a is not None else 0, "sum_chosen": int(stats.sum_score[a]) if a is not None else 0,
                            "columns_differing": diff[:12], "n_columns_differing": len(diff),
                            "near_duplicate_of": ("%s_%s" % other) if other else None})
    return out


def summarize(w, res, eng, world, depth):
    stats = res["stats"]
    return {"workload": w.label, "reads_per_gpu": w.n_reads, "n_alleles": int(w.idx.n_alleles), "n_loci": int(w.idx.n_loci),
            "sieve": eng.sieve_info(), "parallelism": "reads sharded x%d" % world, "pipeline_depth": depth, "cu_partitions": res.get("cu_partitions"), "distinct_resident_batches": len(w.batches),
            "collectives": res["collectives"], "exchange_per_step": res.get("exchange"), "resident_format": "2-bit bases %d B/read + Phred rows %d B/read" % (w.wpr * 4, w.qstride)}


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (this pool's driver shares device memory between processes by dmabuf only: RCCL needs it)
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # native libraries first, before this process initialises the GPU (a stale build runs make / hipcc as children)
    import __graft_entry__ as ge
    if local_rank == 0:
        ge.build(env=clean_env_for_children())
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # MLST_BENCH_BACKEND=gloo + MLST_BENCH_ONE_GPU=1 let the whole N>1 path be exercised on a 1-GPU box
    # (every rank on device 0, collectives through gloo); the real runs use RCCL, one GPU per rank.
    backend = os.environ.get("MLST_BENCH_BACKEND", "nccl")
    if os.environ.get("MLST_BENCH_ONE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        dist.barrier()              # rank 0 has built the libraries
    from metamlst_amd.engine import Engine

    depth = max(1, min(8, args.pipeline))
    tmp = tempfile.mkdtemp(prefix="mlst_bench_%d_" % rank)
    t_start = time.time()
    args.measure_second_command = True
    w = build_workload(args.workload, args, lambda: Engine(local_rank), torch, device, rank, depth, tmp)
    if args.calibrate:
        for _ in range(3):
            _c = w.batches[0][1].clone()
        torch.cuda.synchronize(device)
        del _c
    res = run_workload(w, args, torch, dist, device, rank, world, backend)
    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    eng = w.engines[0]
    roof, roof_ext = rooflines(w, res, eng)
    stats = res["stats"]
    conc = {"species_planted": len(w.planted), "species_typed_correctly": int(sum(res["typed_ok"].values())),
            "st_match": all(res["typed_ok"].values()), "batches_checked": res["batches_checked"],
            "not_typed": {sp: {"called": res["st_call"].get(sp), "planted": w.planted[sp]} for sp, ok in res["typed_ok"].items() if not ok}}
    cpu = None
    e2e = None
    secondary = None
    if world == 1:
        if args.cpu_seconds > 0:
            cpu, same = cpu_baseline(w, args, res)
            conc["gpu_equals_cpu_oracle_on_sample"] = same
        if not args.no_secondary:
            e2e = end_to_end(w, args, torch, device)
    out = {"metric": "Mreads/s aligned+ST-typed, metamlstDB_2022; ST concordance vs CPU ref",
           "value": round(res["value"], 2), "unit": "Mreads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(res["ms_per_step"], 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "int32", "data": "synthetic",
           "config": summarize(w, res, eng, world, depth),
           "roofline": roof, "roofline_extend": roof_ext, "cpu_baseline": cpu, "concordance": conc, "end_to_end": e2e,
           "timed_region": {"blocks": res["blocks"], "steps_per_block": args.steps, "timed_s": res["timed_s"], "reported": "median block", "block_ms": res["block_ms"][:32]},
           "kernel_ms_per_launch_isolated": {k: round(v, 4) for k, v in res["iso_launch_ms"].items()},
           "host_ms_per_step": res["host_ms_per_step"], "serial_ms_per_step": round(res["serial_ms_per_step"], 4),
           "counters": {"records": int(stats.counters[0]), "ignored": int(stats.counters[1]), "candidates": int(stats.counters[3]),
                        "retained": int(stats.counters[4]), "items": int(stats.counters[5]), "banded_sw_pairs": int(stats.counters[6]),
                        "routed_filter_passes": int(stats.counters[7])},
           "index_bytes": dict(zip(("allele_arena_and_haplotype_tables", "sieve", "seed_table"), eng.index_bytes()[:3])), "extend": eng.extend_info(),
           "setup_s": {"database": round(w.t_db, 1), "index_host": round(w.t_index_host, 1), "index_device_x%d" % depth: round(w.t_index_dev, 1),
                       "resident_reads_x%d" % depth: round(w.t_reads, 1), "second_command": w.t_cached},
           "world_size_reported_by_backend": (dist.get_world_size() if world > 1 else 1), "literal_parity_bowtie2": None}
    # ---- secondary block: cfg2 (configs[1]) in the same run, N = 1 only
    if world == 1 and not args.no_secondary and args.workload == "cfg3":
        for e in list(w.engines) + list(getattr(w, "folder_engines", []))[len(w.engines):]:      # (the folder legs made two more)
            e.close()
        del w.batches
        torch.cuda.empty_cache()
        a2 = argparse.Namespace(**vars(args)); a2.reads = 0; a2.alleles = 0; a2.genome_size = 0
        w2 = build_workload("cfg2", a2, lambda: Engine(local_rank), torch, device, rank, depth, tmp)
        r2 = run_workload(w2, a2, torch, dist, device, rank, world, backend)
        roof2, rext2 = rooflines(w2, r2, w2.engines[0])
        secondary = {"config": summarize(w2, r2, w2.engines[0], world, depth), "value": round(r2["value"], 2), "unit": "Mreads/s",
                     "ms_per_step": round(r2["ms_per_step"], 4), "serial_ms_per_step": round(r2["serial_ms_per_step"], 4),
                     "roofline": roof2, "roofline_extend": rext2, "st_match": all(r2["typed_ok"].values()),
                     "kernel_ms_per_launch_isolated": {k: round(v, 4) for k, v in r2["iso_launch_ms"].items()},
                     "timed_region": {"blocks": r2["blocks"], "timed_s": r2["timed_s"]},
                     "end_to_end": end_to_end(w2, a2, torch, device)}
        out["literal_parity_bowtie2"] = literal_leg(w2, a2, torch, tmp)
    out["secondary_cfg2"] = secondary
    # ---- the same step on a PubMLST-shaped database (alleles per locus 10 ... 10,000): what the real database's skew does to it
    skewed = None
    if world == 1 and not args.no_secondary and args.workload == "cfg3":
        for e in list(w2.engines) + list(getattr(w2, "folder_engines", []))[len(w2.engines):]:
            e.close()
        del w2.batches
        torch.cuda.empty_cache()
        a3 = argparse.Namespace(**vars(args)); a3.reads = 0; a3.alleles = 0; a3.genome_size = 0
        w3 = build_workload("skewed", a3, lambda: Engine(local_rank), torch, device, rank, depth, tmp)
        r3 = run_workload(w3, a3, torch, dist, device, rank, world, backend)
        e3 = w3.engines[0]
        mis = mistyped_loci(w3, full_stats(w3))
        skewed = {"config": summarize(w3, r3, e3, world, depth), "value": round(r3["value"], 2), "unit": "Mreads/s",
                  "ms_per_step": round(r3["ms_per_step"], 4), "serial_ms_per_step": round(r3["serial_ms_per_step"], 4),
                  "kernel_ms_per_launch_isolated": {k: round(v, 4) for k, v in r3["iso_launch_ms"].items()},
                  "extend": e3.extend_info(), "items": int(r3["stats"].counters[5]), "records": int(r3["stats"].counters[0]),
                  "species_typed_as_planted": "%d of %d" % (int(sum(r3["typed_ok"].values())), len(w3.planted)),
                  "mistyped_loci": mis,
                  "why": "two causes, both properties of metamlst.py:142-151's penalised average, engine = oracle in each (profiles/round4/check_batch.json): "
                         "(1) synth.make_skewed_db plants %d loci that are ~1 %% copies of a locus of another species -- the reads of BOTH species are records of BOTH "
                         "loci, and (maxLen - nHits) * penalty favours the allele with the most records (mistyped_loci[].near_duplicate_of); (2) an allele one column "
                         "away from the planted one, the column within a read length of the allele's end: reads that overlap the allele by a few dozen bases are low-scoring "
                         "records of the planted allele and no records of the neighbour (the mismatch takes them under --minscore), and a missing record costs 100 where an "
                         "average record scores ~260, so the neighbour's average is HIGHER (tests/test_gpu_baseline_sizes.py: sk000_g4, 264.5 vs 264.3).  "
                         "Without the near-duplicate loci: secondary_skewed.no_duplicate_loci" % len(w3.sdb.duplicates),
                  "timed_region": {"blocks": r3["blocks"], "timed_s": r3["timed_s"]}}
        # the same database shape without the near-duplicate loci (statistics only: a few steps)
        for e in w3.engines:
            e.close()
        del w3.batches
        torch.cuda.empty_cache()
        a4 = argparse.Namespace(**vars(a3)); a4.min_seconds = 0.05; a4.steps = 5; a4.warmup = 1
        w4 = build_workload("skewed_nodup", a4, lambda: Engine(local_rank), torch, device, rank, depth, tmp)
        r4 = run_workload(w4, a4, torch, dist, device, rank, world, backend)
        skewed["no_duplicate_loci"] = {"species_typed_as_planted": "%d of %d" % (int(sum(r4["typed_ok"].values())), len(w4.planted)),
                                       "mistyped_loci": mistyped_loci(w4, full_stats(w4)), "ms_per_step": round(r4["ms_per_step"], 4),
                                       "steps_per_block": a4.steps, "blocks": r4["blocks"]}
    out["secondary_skewed"] = skewed
    out["wall_s"] = round(time.time() - t_start, 1)
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
