#!/usr/bin/env python3
"""Engine against the CPU oracle on WHOLE resident batches of the bench workloads (cfg3, or cfg2 with --workload) (GPU box; ~2 min of 256 host threads
per batch).  bench.py checks the planted STs on the batch of its last step only and the oracle on a 4 M-read slice; this
script takes any batch index, types it with the engine, runs the oracle over all of its reads in chunks (the statistics
are additive) and compares sums, hit counts, first-seen order, pile-up counts and the ST calls.

    python3 profiles/check_batch.py --batches 0,1 > gpurun_out/check_batch.json
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="0,1")
    ap.add_argument("--chunk", type=int, default=4_000_000)
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg2", "skewed"])
    ap.add_argument("--engine-only", action="store_true", help="skip the oracle (diagnostics of the engine-side sequence)")
    a = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    import torch
    import bench
    import oracle_lib
    from metamlst_amd import synth
    from metamlst_amd.engine import Engine
    from metamlst_amd.merge import EngineMatcher, SpeciesSession, parse_nfo_line
    from metamlst_amd import db as mdb
    from metamlst_amd.typing import pick_alleles_fast, type_sample
    sys.argv = [sys.argv[0]]
    args = bench.parse_args()
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    want = [int(x) for x in a.batches.split(",")]
    tmp = tempfile.mkdtemp(prefix="mlst_chk_")
    w = bench.build_workload(a.workload, args, lambda: Engine(0), torch, device, 0, max(want) + 1, tmp)
    eng = w.engines[0]
    orc = oracle_lib.Oracle(w.idx, threads=os.cpu_count() or 1)
    matcher = EngineMatcher(eng, w.idx)
    cache = mdb.DbCache(w.database.conn, w.idx)
    out = []
    for b in want:
        packed, qrows, lens, n_total = w.batches[b]
        t0 = time.time()
        def stage(name):
            torch.cuda.synchronize(device)
            eng.synchronize()
            sys.stderr.write("batch %d engine: %s done\n" % (b, name))
            sys.stderr.flush()

        eng.reset_sample()
        stage("reset")
        eng.submit_packed_device(packed.data_ptr(), qrows.data_ptr(), lens.data_ptr(), n_total, w.wpr, w.qstride)
        stage("pass 1")
        eng.typing_enqueue(penalty=100)
        sg, chosen_dev, letters_dev = eng.typing_fetch()
        stage("typing")
        ch = sorted(pick_alleles_fast(w.idx, sg, 100).values())
        pg = eng.pileup(ch)
        stage("explicit pile-up")
        sessions = {sp: SpeciesSession(w.database, sp, 5, matcher, cache) for sp in w.planted}
        res = type_sample(w.idx, sg, None, w.database, "sample", fast=True, cache=cache, typed=(chosen_dev, letters_dev))
        calls = {}
        for r in res:
            if r.written:
                organism, (bacteriumLine, sampleRecord) = parse_nfo_line(r.nfo_line)
                if organism in sessions:
                    calls[organism] = sessions[organism].add_sample(bacteriumLine, sampleRecord)
        stage("ST calls")
        t_eng = time.time() - t0
        if a.engine_only:
            print(json.dumps({"workload": a.workload, "batch": b, "st_calls_engine": calls, "engine_s": round(t_eng, 1)}), flush=True)
            continue
        # the oracle over every read of the batch, chunk by chunk
        t0 = time.time()
        so = None
        for first in range(0, n_total, a.chunk):
            cnt = min(a.chunk, n_total - first)
            bb, qq = synth.resident_to_host_reads(packed, qrows, n_total, w.wpr, w.qstride, first, cnt, args.read_len)
            fb, fq, off = synth.flatten_reads(bb, qq)
            orc.submit_reads(fb, fq, off, read_base=first)
            s = orc.stats()
            if so is None:
                so = s
            else:
                so.sum_score += s.sum_score; so.n_hits += s.n_hits; so.locus_len_sum += s.locus_len_sum
                so.locus_first = np.minimum(so.locus_first, s.locus_first); so.counters += s.counters
            sys.stderr.write("batch %d oracle pass 1: %d / %d reads, %.0f s\n" % (b, first + cnt, n_total, time.time() - t0))
        cho = sorted(pick_alleles_fast(w.idx, so, 100).values())
        po = None
        for first in range(0, n_total, a.chunk):
            cnt = min(a.chunk, n_total - first)
            bb, qq = synth.resident_to_host_reads(packed, qrows, n_total, w.wpr, w.qstride, first, cnt, args.read_len)
            fb, fq, off = synth.flatten_reads(bb, qq)
            orc.submit_reads(fb, fq, off, read_base=first)
            p = orc.pileup(cho)
            if po is None:
                po = p
            else:
                for k in p:
                    po[k] = po[k] + p[k]
            sys.stderr.write("batch %d oracle pile-up: %d / %d reads, %.0f s\n" % (b, first + cnt, n_total, time.time() - t0))
        rec = {"workload": a.workload, "batch": b, "reads": n_total, "engine_s": round(t_eng, 1), "oracle_s": round(time.time() - t0, 1),
               "sum_score_equal": bool(np.array_equal(sg.sum_score, so.sum_score)), "n_hits_equal": bool(np.array_equal(sg.n_hits, so.n_hits)),
               "locus_len_equal": bool(np.array_equal(sg.locus_len_sum, so.locus_len_sum)), "locus_first_equal": bool(np.array_equal(sg.locus_first, so.locus_first)),
               "chosen_equal": ch == cho, "pileup_equal": bool(ch == cho and all(np.array_equal(pg[k], po[k]) for k in ch)),
               "st_calls_engine": calls, "planted": w.planted,
               "not_as_planted": {sp: calls.get(sp) for sp in w.planted if calls.get(sp) != w.planted[sp]}}
        # for every species that did not come out as planted: the planted and the chosen allele of each locus, their scores
        # (sum, hits) and how the consensus of the locus differs from the planted allele
        detail = {}
        idx = w.idx
        for sp in rec["not_as_planted"]:
            st_row = [r for s_, _, r in w.plan if s_ == sp][0]
            planted_alleles = [int(x) for x in w.sdb.profiles[sp][st_row]]
            loci = [l for l in range(idx.n_loci) if idx.loci[l][0] == sp]
            d = []
            for l, pa in zip(loci, planted_alleles):
                a0, cnt = int(idx.locus_begin[l]), int(idx.locus_count[l])
                nos = idx.allele_no[a0:a0 + cnt]
                pidx = a0 + int(np.nonzero(nos == pa)[0][0])
                cidx = [a for a in ch if a0 <= a < a0 + cnt]
                e = {"gene": idx.loci[l][1], "planted_allele": pa, "planted_sum_hits": [int(sg.sum_score[pidx]), int(sg.n_hits[pidx])]}
                if cidx:
                    c = cidx[0]
                    e["chosen_allele"] = int(idx.allele_no[c]); e["chosen_sum_hits"] = [int(sg.sum_score[c]), int(sg.n_hits[c])]
                    ps, cs = idx.sequence(pidx), idx.sequence(c)
                    e["planted_vs_chosen_diff_columns"] = [i for i in range(min(len(ps), len(cs))) if ps[i] != cs[i]][:20]
                    e["len"] = [len(ps), len(cs)]
                    cnts = pg[c]
                    e["zero_coverage_columns"] = [int(i) for i in np.nonzero(cnts.sum(axis=1) == 0)[0][:20]]
                d.append(e)
            detail[sp] = d
        rec["detail"] = detail
        if not rec["sum_score_equal"]:
            d = np.nonzero(sg.sum_score != so.sum_score)[0]
            rec["sum_score_diff"] = {"n": int(d.size), "first": [[int(i), int(sg.sum_score[i]), int(so.sum_score[i])] for i in d[:10]]}
        out.append(rec)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
