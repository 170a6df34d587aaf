#!/usr/bin/env python3
"""Scale check towards BASELINE.json configs[2] (cfg3): a many-species synthetic database and a mixed
metagenome on one MI355X.  NOT the contract benchmark (that is bench.py / cfg2): this script validates the
large-index paths (plain sieve kernel, multi-species typing) and reports where the time goes.

    python bench_scale.py --species 50 --genomes 20 --reads 20000000
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--species", type=int, default=50)
    ap.add_argument("--alleles", type=int, default=300)
    ap.add_argument("--genomes", type=int, default=20)
    ap.add_argument("--genome-size", type=int, default=2_000_000)
    ap.add_argument("--reads", type=int, default=20_000_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--oracle-reads", type=int, default=300_000)
    a = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    ge.build()
    from bench import rows_to_tiled, synth_reads_gpu, tiled_to_rows
    from metamlst_amd import db as mdb
    from metamlst_amd import synth
    from metamlst_amd.engine import Engine
    from metamlst_amd.index import load_index
    from metamlst_amd.merge import EngineMatcher, SpeciesSession, parse_nfo_line
    from metamlst_amd.typing import type_sample

    device = torch.device("cuda", 0)
    tmp = tempfile.mkdtemp(prefix="mlst_scale_")
    t0 = time.time()
    sdb = synth.make_full_db(os.path.join(tmp, "full.db"), n_species=a.species, alleles_per_locus=a.alleles, n_profiles=200)
    idx = load_index(sdb.path)
    t_db = time.time() - t0
    t0 = time.time()
    eng = Engine(0)
    eng.load_reference(idx)
    t_index = time.time() - t0
    # metagenome: `genomes` species with log-normal abundances, each an isolate of ST row k
    rng = np.random.default_rng(5)
    chosen_sp = list(rng.choice(a.species, size=min(a.genomes, a.species), replace=False))
    ab = rng.lognormal(0.0, 1.0, size=len(chosen_sp))
    ab /= ab.sum()
    parts, planted = [], {}
    packed_all, q_all, l_all = [], [], []
    wpr = qstride = None
    for k, (si, frac) in enumerate(zip(chosen_sp, ab)):
        sp = sdb.species[int(si)]
        st_row = k % len(sdb.profiles[sp])
        planted[sp] = st_row + 1
        g, _ = synth.make_genome(sdb, sp, sdb.profiles[sp][st_row], size=a.genome_size, seed=1000 + k)
        n = max(1000, int(a.reads * frac))
        p, q, l, wpr, qstride = synth_reads_gpu(eng, torch, device, g, n, 150, seed=2000 + k)
        packed_all.append(tiled_to_rows(p, n, wpr)); q_all.append(q[:n * qstride]); l_all.append(l[:n]); parts.append((sp, n))
    n_total = sum(n for _, n in parts)
    perm = torch.randperm(n_total, device=device)
    packed = rows_to_tiled(torch.cat(packed_all)[perm].contiguous(), torch)
    qrows = torch.cat(q_all).view(n_total, qstride)[perm].contiguous().view(-1)
    lens = torch.cat([torch.cat(l_all)[perm], torch.zeros(2, dtype=torch.int16, device=device)])
    del packed_all, q_all, l_all
    database = mdb.metaMLST_db(sdb.path)
    cache = mdb.DbCache(database.conn)
    matcher = EngineMatcher(eng, idx)
    sessions = {sp: SpeciesSession(database, sp, 5, matcher, cache) for sp in planted}     # merge-run prologue: once
    eng.set_profiling(True)
    times = []
    for s in range(a.steps + 1):
        if s == 1:
            eng.reset_kernel_time()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.reset_sample()
        eng.submit_packed_device(packed.data_ptr(), qrows.data_ptr(), lens.data_ptr(), n_total, wpr, qstride)
        eng.typing_enqueue(penalty=100)                 # allele choice + pileup + consensus queued behind pass 1
        st, chosen_dev, letters_dev = eng.typing_fetch()
        res = type_sample(idx, st, None, database, "meta", fast=True, cache=cache, typed=(chosen_dev, letters_dev))
        calls = {}
        for r in res:
            if r.written:
                org, (bl, sr) = parse_nfo_line(r.nfo_line)
                calls[org] = sessions[org].add_sample(bl, sr) if org in sessions else None
        times.append(time.perf_counter() - t0)
    kern = {k: eng.kernel_time(k) for k in ("sieve", "seed", "extend", "banded_sw", "accumulate", "pileup")}
    depth = {sp: n * 150 / a.genome_size for sp, n in parts}
    typed_ok = {sp: calls.get(sp) == planted[sp] for sp in planted}
    out = {"workload": "cfg3-style: %d species x 7 loci x %d alleles (%d alleles, synthetic stand-in for metamlstDB_2022), %d reads from %d genomes"
                       % (a.species, a.alleles, idx.n_alleles, n_total, len(parts)),
           "Mreads_per_s": round(n_total / np.median(times[1:]) / 1e6, 1), "ms_per_pass": round(float(np.median(times[1:])) * 1e3, 3),
           "kernel_ms_per_launch": {k: round(v[0] / max(1, v[1]), 4) for k, v in kern.items()},
           "index_bytes": dict(zip(("allele_arena", "sieve(+bitmap)", "seed_table"), eng.index_bytes()[:3])),
           "db_build_s": round(t_db, 1), "index_build_s": round(t_index, 1),
           "counters": {k: int(v) for k, v in zip(("records", "ignored", "reads", "candidates", "retained", "items", "banded_sw_pairs"), st.counters)},
           "species_typed_correctly": sum(typed_ok.values()), "species_planted": len(planted),
           "not_typed": {sp: {"depth_x": round(depth[sp], 2), "called": calls.get(sp), "planted": planted[sp]} for sp, ok in typed_ok.items() if not ok}}
    # parity of the engine with the oracle on a slice of the same reads
    if a.oracle_reads > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        n_o = min(a.oracle_reads, n_total)
        pk = tiled_to_rows(packed, n_total, wpr)[:n_o].cpu().numpy().view(np.uint32).reshape(n_o, wpr)
        qr = qrows[:n_o * qstride].cpu().numpy().reshape(n_o, qstride)
        codes = np.zeros((n_o, wpr * 16), np.uint8)
        for k in range(16):
            codes[:, k::16] = (pk >> (2 * k)) & 3
        bases = np.frombuffer(b"ACGT", np.uint8)[codes[:, :150]]
        quals = (qr[:, :150] & 0x7F) + 33
        fb, fq, off = synth.flatten_reads(bases, quals.astype(np.uint8))
        orc = oracle_lib.Oracle(idx, threads=os.cpu_count() or 1)
        orc.submit_reads(fb, fq, off)
        so = orc.stats()
        eng.reset_sample()
        eng.submit_reads(fb, fq, off)
        sg = eng.stats()
        out["engine_equals_oracle_on_slice"] = bool(np.array_equal(sg.sum_score, so.sum_score) and np.array_equal(sg.n_hits, so.n_hits)
                                                    and np.array_equal(sg.locus_first, so.locus_first))
        out["oracle_slice_reads"] = n_o
    print(json.dumps(out))


if __name__ == "__main__":
    main()
