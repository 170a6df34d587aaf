#!/usr/bin/env python3
"""Headline benchmark: Mreads/s aligned + ST-typed (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full typing pass over one resident batch of reads: seed sieve -> exact seeds ->
extension against every allele -> hit accumulation -> allele choice -> pileup -> consensus ->
.nfo line -> allele match + ST call.  At N=1 the workload is BASELINE.json configs[1]
(10 M 150 bp SE reads, E. coli-like database: 7 loci, ~10 k alleles); at N>1 every rank holds
its own 10 M-read shard of the same isolate (weak scaling), the two all-reduces of
metamlst_amd/dist.py run over RCCL, and rank 0 runs the host tail.

By default four engines (four HIP streams, four sets of sample state) work on the same resident batch
in turn (--pipeline 4; measured: 2 / 3 / 4 / 5 / 6 engines -> 22.9 / 27.4 / 31.0 / 26.1 / 26.2 Greads/s): while the host
types step k (.nfo line, ST call) the GPU already runs the passes of the next steps, and the small latency-bound
kernels of one step overlap the streaming kernels of another.
At N=1 the allele choice, pileup and consensus are queued on the device right behind pass 1
(mlst_typing_enqueue), so a step has a single host round trip.  Every step is still one complete pass and
the timed region holds exactly K of them; `serial_ms_per_step` reports the strictly serial step
(--pipeline 1 times the whole run that way).

Reads are synthesised on the GPU before the timed region and are resident in HBM in the packed
format of SURVEY.md 8(d) (2-bit bases + Phred rows); the timed region contains no H2D copy of
reads.  The JSON line also carries the roofline of the dominant kernel (HIP events on the
engine's stream) and the CPU oracle timed on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_BASES = 40      # 2-bit bases of a 150 bp read, rounded to the 10-word row the sieve streams
ALG_BYTES_SURVEY = 188    # 38 B 2-bit bases + 150 B Phred per 150 bp read
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU")
    ap.add_argument("--alleles", type=int, default=1430, help="alleles per locus")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--genome", type=int, default=4_600_000)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline time (0 = skip)")
    ap.add_argument("--st-row", type=int, default=11)
    ap.add_argument("--pipeline", type=int, default=4,
                    help="engines per GPU: with more than one, the GPU already works on the next steps while the host types "
                         "step k (multiple buffering; every step is still a complete pass); 1 = strictly serial steps")
    ap.add_argument("--calibrate", action="store_true",
                    help="before the timed region copy the Phred rows 3x with torch (a known-size wide coalesced stream) "
                         "so a rocprofv3 --pmc FETCH_SIZE pass of this command can be calibrated")
    return ap.parse_args()


def synth_reads_gpu(eng, torch, device, genome: np.ndarray, n_reads: int, L: int, seed: int, chunk: int = 1 << 19):
    """Reads of SURVEY.md 8(d) cfg1/cfg2 made on the GPU: uniform starts, both strands, Phred 40 except
    0.1 % substitution errors at Phred 15; packed with the engine's own pack kernel (mlst_pack_reads_device)."""
    wpr = (L + 15) // 16
    wpr += wpr & 1
    qstride = (L + 7) & ~7
    g = torch.from_numpy(genome).to(device)
    comp = torch.full((256,), ord("N"), dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    code = torch.zeros(256, dtype=torch.int64, device=device)
    for k, a in enumerate(b"ACGT"):
        code[a] = k
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    packed = torch.zeros((n_reads + 63) // 64 * 64 * wpr + 4, dtype=torch.int32, device=device)   # whole groups of 64 rows (mlst.h)
    qrows = torch.zeros(n_reads * qstride, dtype=torch.uint8, device=device)
    lens = torch.zeros(n_reads + 2, dtype=torch.int16, device=device)
    ar = torch.arange(L, device=device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    for c0 in range(0, n_reads, chunk):
        n = min(chunk, n_reads - c0)
        start = torch.randint(0, len(genome) - L + 1, (n,), generator=gen, device=device)
        b = g[start[:, None] + ar[None, :]]
        rev = torch.rand(n, generator=gen, device=device) < 0.5
        b = torch.where(rev[:, None], comp[b.flip(1).long()], b)
        err = torch.rand((n, L), generator=gen, device=device) < 0.001
        sub = acgt[(code[b.long()] + torch.randint(1, 4, (n, L), generator=gen, device=device)) % 4]
        b = torch.where(err, sub, b).contiguous()
        q = torch.where(err, torch.tensor(15 + 33, dtype=torch.uint8, device=device),
                        torch.tensor(40 + 33, dtype=torch.uint8, device=device)).contiguous()
        off = (torch.arange(n + 1, device=device, dtype=torch.int64) * L).contiguous()
        torch.cuda.synchronize(device)
        eng.pack_reads_device(b.data_ptr(), q.data_ptr(), off.data_ptr(), n, packed.data_ptr() + c0 * wpr * 4,
                              qrows.data_ptr() + c0 * qstride, lens.data_ptr() + c0 * 2, wpr, qstride)
        eng.synchronize()
        del b, q, err, sub, start, rev, off
    return packed, qrows, lens, wpr, qstride


def tiled_to_rows(packed, n_reads: int, wpr: int):
    """Resident 2-bit rows (groups of 64 reads, transposed in 8-byte units, include/mlst.h) -> plain [n_reads, wpr] rows."""
    g = (n_reads + 63) // 64
    return packed[:g * 64 * wpr].view(g, wpr // 2, 64, 2).permute(0, 2, 1, 3).reshape(g * 64, wpr)[:n_reads]


def rows_to_tiled(rows, torch):
    """Plain [n, wpr] rows -> the resident group-transposed layout (+4 words of slack)."""
    n, wpr = rows.shape
    g = (n + 63) // 64
    pad = torch.zeros((g * 64, wpr), dtype=rows.dtype, device=rows.device)
    pad[:n] = rows
    t = pad.view(g, 64, wpr // 2, 2).permute(0, 2, 1, 3).contiguous().view(-1)
    return torch.cat([t, torch.zeros(4, dtype=rows.dtype, device=rows.device)])


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
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

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from metamlst_amd import db as mdb
    from metamlst_amd import synth
    from metamlst_amd.dist import DeviceStatsPort, StreamedShard, allreduce_consensus, allreduce_stats
    from metamlst_amd.engine import Engine
    from metamlst_amd.index import load_index
    from metamlst_amd.merge import EngineMatcher, SpeciesSession, parse_nfo_line
    from metamlst_amd.typing import SampleStats, type_sample

    # ---- database + isolate (same on every rank: seeded)
    tmp = tempfile.mkdtemp(prefix="mlst_bench_%d_" % rank)
    db_path = os.path.join(tmp, "ecoli.db")
    t0 = time.time()
    sdb = synth.make_ecoli_db(db_path, alleles_per_locus=args.alleles, n_profiles=5000)
    idx = load_index(db_path)
    database = mdb.metaMLST_db(db_path)
    st_tuple = sdb.profiles["ecoli"][args.st_row]
    genome, _ = synth.make_genome(sdb, "ecoli", st_tuple, size=args.genome)
    depth = max(1, min(8, args.pipeline))
    engines = [Engine(local_rank) for _ in range(depth)]       # one HIP stream and one set of sample state each
    for e in engines:
        e.load_reference(idx)
    eng = engines[0]
    t_setup = time.time() - t0
    packed, qrows, lens, wpr, qstride = synth_reads_gpu(eng, torch, device, genome, args.reads, args.read_len,
                                                        seed=synth.SEED + 1000 * rank)
    # N > 1: every engine runs on its own torch stream, so that its kernels and the RCCL all-reduces of its step are
    # ordered on the device and the host synchronises once per step (metamlst_amd.dist.StreamedShard)
    shards = [StreamedShard(e, device) for e in engines] if world > 1 else None
    ports = [DeviceStatsPort(e, device) for e in engines] if world > 1 else None
    mode = {"streamed": world > 1}
    matcher = EngineMatcher(eng, idx)
    true_st = args.st_row + 1
    # merge-run prologue (metamlst-merge.py:119-142) happens once per run of many samples: untimed setup
    cache = mdb.DbCache(database.conn)
    sessions = {sp: SpeciesSession(database, sp, 5, matcher, cache) for sp in idx.species} if rank == 0 else {}

    if args.calibrate:
        for _ in range(3):
            _c = qrows.clone()
        torch.cuda.synchronize(device)
        del _c
    host_ms = {"submit": 0.0, "stats": 0.0, "typing+pileup": 0.0, "st_call": 0.0}

    def submit(k):
        """Everything of step k that runs on the GPU, queued without waiting: pass 1 (sieve -> seeds -> extension ->
        accumulation), at N > 1 the all-reduce of the statistics, allele choice, pileup, at N > 1 the all-reduce of the
        pileup counts, consensus, copies to the host -- on engine k % depth."""
        t_a = time.perf_counter()
        e = engines[k % depth]

        def pass1():
            e.reset_sample()
            e.set_read_index_base(rank * args.reads)
            e.submit_packed_device(packed.data_ptr(), qrows.data_ptr(), lens.data_ptr(), args.reads, wpr, qstride)

        if world > 1 and mode["streamed"]:
            shards[k % depth].enqueue(pass1, penalty=100)
        elif world > 1:
            pass1()                                   # host-driven exchange: the collectives follow in finish()
        else:
            pass1()
            e.typing_enqueue(penalty=100)
        host_ms["submit"] += (time.perf_counter() - t_a) * 1e3

    def finish(k):
        """The host part of step k: wait for its device work, then .nfo line (gap-fill, accuracy gate) and ST call."""
        e = engines[k % depth]
        t_b = time.perf_counter()
        if world > 1 and not mode["streamed"]:
            # fallback: the same two exchanges driven from the host with a synchronisation around every collective
            port = ports[k % depth]
            allreduce_stats(port, device)
            st = e.stats()
            t_c = time.perf_counter()

            def consensus_fn(chosen):
                n_cols = sum(int(idx.off[a + 1] - idx.off[a]) for a in chosen)
                return allreduce_consensus(port, idx, chosen, n_cols, device)

            res = type_sample(idx, st, None, database, "sample", fast=True, cache=cache, consensus_fn=consensus_fn)
        else:
            st, chosen_dev, letters_dev = e.typing_fetch()
            t_c = time.perf_counter()
            res = type_sample(idx, st, None, database, "sample", fast=True, cache=cache, typed=(chosen_dev, letters_dev))
        t_d = time.perf_counter()
        out = {}
        if rank == 0:
            for r in res:
                if r.written:       # .nfo line -> allele match + ST call (per-sample body of metamlst-merge.py:144-240)
                    organism, (bacteriumLine, sampleRecord) = parse_nfo_line(r.nfo_line)
                    out[organism] = sessions[organism].add_sample(bacteriumLine, sampleRecord)
        t_e = time.perf_counter()
        host_ms["stats"] += (t_c - t_b) * 1e3        # waiting for the device work of the step
        host_ms["typing+pileup"] += (t_d - t_c) * 1e3
        host_ms["st_call"] += (t_e - t_d) * 1e3
        return out, st

    def run(n_steps):
        """n_steps complete steps; with depth > 1 the GPU works on step k+1 while the host finishes step k."""
        last = None
        for k in range(min(depth - 1, n_steps)):
            submit(k)
        for k in range(n_steps):
            if k + depth - 1 < n_steps:
                submit(k + depth - 1)
            last = finish(k)
        return last

    def fence():
        for e in engines:
            e.synchronize()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    # N > 1: one step both ways before anything is timed.  The streamed step (kernels and RCCL collectives ordered on a
    # torch stream, one host synchronisation) must reproduce the host-driven exchange bit for bit on every rank; if it
    # does not, or raises, every rank falls back to the host-driven form together (agreed through an all-reduce).
    if world > 1:
        ok = 1
        try:
            mode["streamed"] = False
            submit(0)
            ref_out, ref_st = finish(0)
            mode["streamed"] = True
            submit(0)
            got_out, got_st = finish(0)
            if not (np.array_equal(ref_st.sum_score, got_st.sum_score) and np.array_equal(ref_st.n_hits, got_st.n_hits)
                    and np.array_equal(ref_st.locus_first, got_st.locus_first) and ref_out == got_out):
                ok = 0
        except Exception as exc:      # noqa: BLE001 -- any failure of the streamed form means: use the other one
            print("rank %d: streamed step failed (%s); falling back to host-driven collectives" % (rank, exc), file=sys.stderr)
            ok = 0
        if os.environ.get("MLST_BENCH_HOST_COLLECTIVES"):      # test switch: take the fallback
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int64, device=device)
        torch.cuda.synchronize(device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        mode["streamed"] = bool(int(flag.item()))
    # untimed priming, whatever W is: every engine sees its launch sequence often enough for the hipGraph of it to be built
    # (that happens on the second identical submission) and replayed once; then the W warm-up steps proper
    for e_i in range(depth):
        for _ in range(3):
            submit(e_i)
            finish(e_i)
    fence()
    run(args.warmup)
    # strictly serial steps (one engine), a few of them: the latency of one step, reported beside the throughput
    # strictly serial steps on one engine, untimed, with HIP events around the kernel groups: every kernel has the GPU to
    # itself (event profiling launches kernel by kernel; the timed regions below replay the launch sequence as a hipGraph)
    KNAMES = ("sieve", "seed", "extend", "banded_sw", "accumulate", "pileup", "sieve_inkernel", "sieve_wg_longest")
    engines[0].set_profiling(1)
    engines[0].reset_kernel_time()
    for k in range(min(10, args.steps)):
        submit(0)
        finish(0)
    fence()
    isolated = {k: engines[0].kernel_time(k) for k in KNAMES}
    engines[0].set_profiling(0)
    # the latency of one strictly serial step, reported beside the throughput
    fence()
    t0 = time.perf_counter()
    n_serial = min(20, args.steps)
    for k in range(n_serial):
        submit(0)
        finish(0)
    fence()
    serial_ms = (time.perf_counter() - t0) / max(1, n_serial) * 1e3
    for e in engines:
        e.set_profiling(2)          # in-kernel sieve window only
        e.reset_kernel_time()
    for k in host_ms:
        host_ms[k] = 0.0
    fence()
    t0 = time.perf_counter()
    st_call, stats = run(args.steps)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    kernels = {}
    for k in KNAMES:
        parts = [e.kernel_time(k) for e in engines]
        kernels[k] = (sum(p_[0] for p_ in parts), sum(p_[1] for p_ in parts))
    for e in engines:
        e.set_profiling(0)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    total_reads = args.reads * world
    value = total_reads / (dt / args.steps) / 1e6
    # ---- roofline of the dominant kernel
    iso_launch = {k: (isolated[k][0] / max(1, isolated[k][1])) for k in isolated}
    dom = max((k for k in iso_launch if not k.startswith("sieve_")), key=lambda k: iso_launch[k])
    per_launch = {k: (kernels[k][0] / max(1, kernels[k][1])) for k in kernels}
    # The sieve's launch duration over the timed region is its execution window measured inside the kernel (wall clock
    # at the first workgroup's start / the last one's end); rocprofv3's kernel trace shows the same window.
    # With several engines on one GPU the workgroups of a sieve launch (one per CU, equal shares) start one by one as the
    # previous stream's k_extend leaves the CUs: the window from the first start to the last end stretches although no
    # workgroup is slower.  The launch duration used is the residency of the launch's longest-running workgroup (>= what
    # every other workgroup took; equal to the window when all start together); the window is reported beside it.
    sieve_ms = per_launch.get("sieve_wg_longest", 0) or per_launch.get("sieve_inkernel", 0) or iso_launch["sieve"]
    sieve_window_ms = per_launch.get("sieve_inkernel", 0)
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "sieve_pmc.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))      # measured on 10 M reads per launch; scales with the reads streamed
            traffic = int(pmc["hbm_bytes_per_launch"] * args.reads / 10_000_000)
        except Exception:
            traffic = None
    achieved = args.reads * ALG_BYTES_BASES / (sieve_ms * 1e-3) / 1e9 if sieve_ms > 0 else 0.0
    roofline = {"kernel": "k_sieve", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "alg_bytes_per_read": ALG_BYTES_BASES, "reads_per_launch": args.reads, "avg_launch_ms": round(sieve_ms, 4),
                "achieved_at_188B_per_read": round(args.reads * ALG_BYTES_SURVEY / (sieve_ms * 1e-3) / 1e9, 1) if sieve_ms > 0 else 0.0,
                "dominant_by_time": dom, "launch_window_ms": round(sieve_window_ms, 4)}
    # the same kernel with the GPU to itself (the strictly serial steps before the timed region): in the timed region
    # the kernels of up to `depth` steps share the GPU, which lengthens each launch
    iso_ms = isolated["sieve_inkernel"][0] / max(1, isolated["sieve_inkernel"][1]) or isolated["sieve"][0] / max(1, isolated["sieve"][1])
    if iso_ms > 0:
        roofline["avg_launch_ms_isolated"] = round(iso_ms, 4)
        roofline["achieved_isolated"] = round(args.reads * ALG_BYTES_BASES / (iso_ms * 1e-3) / 1e9, 1)
        roofline["frac_isolated"] = round(roofline["achieved_isolated"] / HBM_PEAK_GBS, 4)

    # ---- CPU baseline: the oracle on a bounded sample of the same workload, all host cores
    cpu = None
    conc = {"st_called": st_call.get("ecoli"), "st_planted": true_st, "st_match": st_call.get("ecoli") == true_st}
    if args.cpu_seconds > 0 and world == 1:      # the CPU baseline is timed at N=1 only
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        cores = os.cpu_count() or 1
        orc = oracle_lib.Oracle(idx, threads=cores)
        n_probe = 200_000
        b, q = synth.sample_reads(genome, n_probe, read_len=args.read_len, seed=99)
        fb, fq, off = synth.flatten_reads(b, q)
        orc.submit_reads(fb, fq, off)
        t1 = time.perf_counter()
        orc.stats()
        probe = time.perf_counter() - t1
        n_cpu = int(min(4_000_000, max(n_probe, n_probe * args.cpu_seconds / max(probe, 1e-3))))
        b, q = synth.sample_reads(genome, n_cpu, read_len=args.read_len, seed=101)
        fb, fq, off = synth.flatten_reads(b, q)
        orc.submit_reads(fb, fq, off)
        from metamlst_amd.typing import pick_alleles_fast
        cpu_dt, reps = 0.0, 0
        while cpu_dt < args.cpu_seconds and reps < 64:      # repeat the bounded sample until ~cpu_seconds of CPU work
            t1 = time.perf_counter()
            so = orc.stats()
            ch = sorted(pick_alleles_fast(idx, so, 100).values())
            po = orc.pileup(ch)
            cpu_dt += time.perf_counter() - t1
            reps += 1
        cpu = {"value": round(n_cpu * reps / cpu_dt / 1e6, 4), "unit": "Mreads/s", "cores": cores, "kind": "port",
               "sample": "%d reads of the same isolate/DB x %d passes, oracle pass 1 + allele choice + pileup, OpenMP x %d threads, %.1f s"
                         % (n_cpu, reps, cores, cpu_dt)}
        # parity of the GPU engine with the oracle on that very sample
        eng.reset_sample()
        eng.submit_reads(fb, fq, off)
        sg = eng.stats()
        pg = eng.pileup(ch)
        same = bool(np.array_equal(sg.sum_score, so.sum_score) and np.array_equal(sg.n_hits, so.n_hits)
                    and all(np.array_equal(pg[a], po[a]) for a in ch))
        conc["gpu_equals_cpu_oracle_on_sample"] = same
        conc["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1) if cpu["value"] > 0 else None

    out = {"metric": "Mreads/s aligned+ST-typed, metamlstDB_2022; ST concordance vs CPU ref",
           "value": round(value, 2), "unit": "Mreads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "int32", "data": "synthetic",
           "config": {"workload": "cfg2: %d x %d bp SE reads per GPU, one E. coli-like isolate (%.1f Mb), synthetic DB 7 loci x %d alleles "
                                  "(metamlstDB_2022 is not available offline)" % (args.reads, args.read_len, args.genome / 1e6, args.alleles),
                      "reads_per_gpu": args.reads, "n_alleles": int(idx.n_alleles), "parallelism": "reads sharded x%d" % world, "pipeline_depth": depth,
                      "collectives": ("streamed on a torch stream" if mode["streamed"] else "host-driven") if world > 1 else None,
                      "resident_format": "2-bit bases %d B/read + Phred rows %d B/read" % (wpr * 4, qstride)},
           "roofline": roofline, "cpu_baseline": cpu, "concordance": conc,
           "kernel_ms_per_launch_isolated": {k: round(v, 4) for k, v in iso_launch.items()},
           # SURVEY.md 8(d) secondary figure for the on-locus minority: ungapped (XOR / bit-plane) cells of k_extend
           "extend": {"pairs_per_launch": int(stats.counters[5]) * (int(idx.n_alleles) // max(1, int(idx.n_loci))),
                      "Gcells_per_s": round(int(stats.counters[5]) * (int(idx.n_alleles) // max(1, int(idx.n_loci))) * args.read_len
                                            / max(1e-9, iso_launch["extend"] * 1e-3) / 1e9, 1)},
           "host_ms_per_step": {k: round(v / args.steps, 4) for k, v in host_ms.items()},
           "serial_ms_per_step": round(serial_ms, 4),
           "counters": {"records": int(stats.counters[0]), "ignored": int(stats.counters[1]), "candidates": int(stats.counters[3]),
                        "retained": int(stats.counters[4]), "items": int(stats.counters[5]), "banded_sw_pairs": int(stats.counters[6])},
           "index_bytes": dict(zip(("allele_arena", "sieve", "seed_table"), eng.index_bytes()[:3])),
           "setup_s": round(t_setup, 1)}
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
