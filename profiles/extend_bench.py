"""k_extend alone: HIP-event time per launch for several settings of the extension path, one process, one resident batch.

    python profiles/extend_bench.py --workload cfg3 --variants hap,pairs [--reads N] [--launches 12]

Variants are environment settings read by mlst_load_reference (one engine per variant, same database, same reads):
    hap            the default (block-haplotype summaries in LDS)
    pairs          MLST_EXT_LDS_KB=0: every (item, allele) pair aligned on its own (rounds 1-3)
    k=v[,k=v]      any MLST_* switches, e.g. MLST_EXT_THREADS=128+MLST_EXT_LDS_KB=64  ('+' separates switches)
Workloads: cfg3 / cfg2 (bench.py's), skewed (synth.make_skewed_db, alleles per locus 10 ... --hi).
Every variant's statistics are compared with the first one's (bit-exact) before its time is reported.
"""
import argparse
import json
import os
import sys
import tempfile
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg2", "skewed"])
    ap.add_argument("--variants", default="hap,pairs")
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--launches", type=int, default=12)
    ap.add_argument("--hi", type=int, default=10_000)
    ap.add_argument("--species", type=int, default=6)
    ap.add_argument("--out", default=None)
    ap.add_argument("--kernels", default="", help="further kernel groups to report (ms per SUBMISSION, all launches of the group): e.g. sieve,sieve_route,sieve_probe,sieve_verify,seed")
    a = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    import torch
    import bench
    from metamlst_amd import synth
    from metamlst_amd.engine import Engine
    from metamlst_amd.index import load_index
    device = torch.device("cuda:0")
    tmp = tempfile.mkdtemp()
    args = types.SimpleNamespace(alleles=0, reads=a.reads, genome_size=0, species=150, genomes=20, read_len=150)
    variants = a.variants.split(",")

    def env_of(v):
        if v == "hap":
            return {}
        if v == "pairs":
            return {"MLST_EXT_LDS_KB": "0"}
        return dict(kv.split("=") for kv in v.split("+"))

    engines = []

    def factory():
        v = variants[len(engines)]
        e = Engine(0)
        engines.append((v, e, None))
        return e

    if a.workload in ("cfg3", "cfg2"):
        # bench.build_workload loads the reference in the factory's environment: wrap load_reference
        t0 = time.time()
        orig_load = Engine.load_reference

        def load_with_env(self, idx, cache_path=None):
            v = [x for x in engines if x[1] is self][0][0]
            ev = env_of(v)
            old = {k: os.environ.get(k) for k in ev}
            os.environ.update(ev)
            try:
                return orig_load(self, idx)      # (no file cache: every variant builds or takes the in-process index)
            finally:
                for k, o in old.items():
                    if o is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = o
        Engine.load_reference = load_with_env
        w = bench.build_workload(a.workload, args, factory, torch, device, 0, len(variants), tmp)
        # one batch is enough: every engine works on batch 0
        batch = w.batches[0]
        idx, wpr, qstride = w.idx, w.wpr, w.qstride
        packed, qrows, lens, n = batch
        setup_s = time.time() - t0
    else:
        t0 = time.time()
        sdb = synth.make_skewed_db(os.path.join(tmp, "sk.db"), n_species=a.species, hi=a.hi)
        idx = load_index(sdb.path)
        orig_load = Engine.load_reference
        for v in variants:
            ev = env_of(v)
            old = {k: os.environ.get(k) for k in ev}
            os.environ.update(ev)
            e = Engine(0)
            e.load_reference(idx)
            engines.append((v, e, old))
            for k, o in old.items():
                if o is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = o
        plan = synth.metagenome_plan(sdb, min(a.species, len(sdb.species)))
        n_reads = a.reads or 20_000_000
        packed, qrows, lens, wpr, qstride, n = synth.make_metagenome_gpu(engines[0][1], torch, device, sdb, plan, n_reads, 2_000_000, seed=11, read_len=150)
        setup_s = time.time() - t0
    out = {"workload": a.workload, "reads": int(n), "setup_s": round(setup_s, 1), "variants": {}}
    ref = None
    for v, e, _ in engines:
        info = e.extend_info()
        ev_run = {k_: v_ for k_, v_ in env_of(v).items() if k_ in ("MLST_RT_DEBUG",)}      # switches read at submission time
        os.environ.update(ev_run)
        if os.environ.get("MLST_X_SKIP") and hasattr(e.lib, "mlst_debug_ext_skip"):
            e.lib.mlst_debug_ext_skip(int(os.environ["MLST_X_SKIP"]))
        e.reset_sample()
        e.submit_packed_device(packed.data_ptr(), qrows.data_ptr(), lens.data_ptr(), n, wpr, qstride)
        st = e.stats()
        key = (st.sum_score.copy(), st.n_hits.copy(), st.locus_len_sum.copy(), st.counters.copy())
        same = True
        if ref is None:
            ref = key
        else:
            same = all(np.array_equal(x, y) for x, y in zip(ref, key))
        e.set_profiling(1)
        times, prep = [], []
        more = {k_: [] for k_ in a.kernels.split(",") if k_}
        for _ in range(a.launches):
            e.reset_sample()
            e.reset_kernel_time()
            e.submit_packed_device(packed.data_ptr(), qrows.data_ptr(), lens.data_ptr(), n, wpr, qstride)
            e.synchronize()
            ms, k = e.kernel_time("extend")
            times.append(ms / max(k, 1))
            ms, k = e.kernel_time("extend_prep")
            prep.append(ms / max(k, 1))
            for name in more:
                ms, k = e.kernel_time(name)
                more[name].append(ms)
        e.set_profiling(0)
        for k_ in ev_run:
            os.environ.pop(k_, None)
        trace = None
        if hasattr(e.lib, "mlst_debug_ext_trace"):      # profiling build (-DMLST_EXT_TRACE): cycles per phase, summed over waves
            import ctypes as C
            buf = (C.c_uint64 * 8)()
            e.lib.mlst_debug_ext_trace(buf, 1)
            if hasattr(e.lib, "mlst_debug_ext_cnt"):
                e.lib.mlst_debug_ext_cnt((C.c_uint64 * 8)(), 1)
            e.reset_sample()
            e.submit_packed_device(packed.data_ptr(), qrows.data_ptr(), lens.data_ptr(), n, wpr, qstride)
            e.synchronize()
            e.lib.mlst_debug_ext_trace(buf, 1)
            t = [int(v) for v in buf]
            names = ("record", "requests", "summaries", "composition", "counts+fused", "handover")
            cnt = None
            if hasattr(e.lib, "mlst_debug_ext_cnt"):
                b2 = (C.c_uint64 * 8)()
                e.lib.mlst_debug_ext_cnt(b2, 1)
                cnt = {"fast_items": int(b2[0]), "fallback_items": int(b2[1]), "span_pairs": int(b2[2]), "turns_with_span_pair": int(b2[3])}
            trace = {"counts": cnt, "items": t[6], "waves": t[7], "cycles_per_item": {k: round(v / max(t[6], 1), 1) for k, v in zip(names, t[:6])}}
        out["variants"][v] = {"extend_ms_median": round(float(np.median(times)), 4), "prep_ms_median": round(float(np.median(prep)), 4), "min": round(min(times), 4), "max": round(max(times), 4),
                              "same_statistics_as_first": bool(same), "records": int(st.counters[0]), "candidates": int(st.counters[3]), "info": info, "trace": trace,
                              "ms_per_submission": {k_: round(float(np.median(v)), 4) for k_, v in more.items()}}
        print(v, out["variants"][v], flush=True)
    s = json.dumps(out)
    print(s)
    if a.out:
        with open(a.out, "w") as fh:
            fh.write(s + "\n")


if __name__ == "__main__":
    main()
