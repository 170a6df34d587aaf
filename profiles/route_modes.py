#!/usr/bin/env python3
"""Why does the routed sieve's producer run in two speeds?  

One process = one observation: the cfg3 workload on one engine, N isolated launches of the sieve timed with HIP events,
per launch; the per-workgroup trace of the last launch (XCC, HW_ID, start and end of every workgroup); the addresses of
the buffers; the clocks the driver reports; then the same again after the routing arena has been freed and allocated
somewhere else (mlst_debug_route_realloc), to tell a property of the PROCESS from a property of the MEMORY it got.
Appends one JSON line to --out.  Run it several times in a row on one box (profiles/route_modes.sh).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def clocks():
    out = {}
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk") + glob.glob("/sys/class/drm/card*/device/pp_dpm_mclk") + glob.glob("/sys/class/drm/card*/device/pp_dpm_fclk"):
        try:
            cur = [ln.strip() for ln in open(f) if "*" in ln]
            out[f.split("/")[4] + "/" + os.path.basename(f)] = cur
        except Exception:
            pass
    return out


def hwmon_paths():
    """sclk / power / temperature files of every card (the busy one is told apart by its power afterwards)"""
    out = []
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        e = {"dir": d}
        for key, names in (("sclk", ("freq1_input",)), ("mclk", ("freq2_input",)), ("power", ("power1_average", "power1_input")),
                           ("temp", ("temp1_input",)), ("temp_mem", ("temp3_input",))):
            for nm in names:
                f = os.path.join(d, nm)
                if os.path.exists(f):
                    e[key] = f
                    break
        out.append(e)
    return out


def sample_hwmon(paths, stop, acc, period=0.01):
    while not stop.is_set():
        t = time.time()
        for k, e in enumerate(paths):
            row = []
            for key in ("sclk", "mclk", "power", "temp", "temp_mem"):
                try:
                    row.append(int(open(e[key]).read().strip()) if key in e else -1)
                except Exception:
                    row.append(-1)
            acc.append((t, k) + tuple(row))
        time.sleep(period)


def wg_summary(tr, which):
    """per-XCC residency of the producer (which = 0) or consumer (1) workgroups of the traced launch"""
    P = tr["P"]
    wg = tr["wg"][:P] if which == 0 else tr["wg"][P:P + 256]
    xcc = (wg[:, 0] & np.uint64(0xF)).astype(int)
    hw = (wg[:, 0] >> np.uint64(32)).astype(np.int64)
    dur = (wg[:, 2].astype(np.int64) - wg[:, 1].astype(np.int64)) / (tr["khz"] / 1e3)      # microseconds
    t0 = wg[:, 1].astype(np.int64)
    span = (wg[:, 2].astype(np.int64).max() - t0.min()) / (tr["khz"] / 1e3)
    per = {}
    for x in range(8):
        m = xcc == x
        if m.any():
            per[str(x)] = {"n": int(m.sum()), "mean_us": round(float(dur[m].mean()), 1), "max_us": round(float(dur[m].max()), 1),
                           "end_us": round(float((wg[m, 2].astype(np.int64).max() - t0.min()) / (tr["khz"] / 1e3)), 1)}
    cu = ((hw >> 8) & 0xF); se = ((hw >> 13) & 0x7)
    return {"span_us": round(float(span), 1), "xcc_of_wg0": int(xcc[0]), "xcc_of_first8": [int(v) for v in xcc[:8]], "per_xcc": per,
            "dur_us_min_med_max": [round(float(v), 1) for v in (dur.min(), np.median(dur), dur.max())],
            "distinct_cu_se_xcc": int(len(set(zip(cu.tolist(), se.tolist(), xcc.tolist()))))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "route_modes.jsonl"))
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--tag", default="")
    ap.add_argument("--pads", default="0,1048576,318767104", help="pad bytes kept between re-allocations of the arena (-1 = keep the old arena allocated)")
    ap.add_argument("--burst", type=float, default=0.0, help="seconds of back-to-back sieve launches during which sclk / power / temperature are sampled (hwmon)")
    ap.add_argument("--clone-reads", type=int, default=0, help="after the arena phases: that many phases on fresh copies of the resident reads (the old copies stay allocated)")
    args = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    import torch
    import bench
    from metamlst_amd.engine import Engine
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    bargs = argparse.Namespace(alleles=0, reads=args.reads, genome_size=0, species=150, genomes=20, read_len=150)
    tmp = tempfile.mkdtemp(prefix="mlst_modes_")
    w = bench.build_workload("cfg3", bargs, lambda: Engine(0), torch, device, 0, 1, tmp)
    eng = w.engines[0]
    packed, qrows, lens, n = w.batches[0]
    eng.route_trace()                       # switches the trace on
    eng.set_profiling(1)

    def launches(k):
        rt, pb = [], []
        for _ in range(k):
            eng.reset_sample()
            eng.reset_kernel_time()
            eng.submit_packed_device(packed.data_ptr(), qrows.data_ptr(), lens.data_ptr(), n, w.wpr, w.qstride)
            eng.synchronize()
            rt.append(eng.kernel_time("sieve_route")[0]); pb.append(eng.kernel_time("sieve_probe")[0])
        return rt, pb

    launches(3)
    # plain streaming rates of this process on this card (is the slow mode a property of the memory system as a whole?)
    def stream_rates():
        x = torch.empty(1 << 30, dtype=torch.int32, device=device)      # 4 GiB
        y = torch.empty(1 << 30, dtype=torch.int32, device=device)
        out = {}
        for name, fn, nbytes in (("fill_GBps", lambda: x.fill_(1), 4 << 30), ("copy_GBps", lambda: y.copy_(x), 8 << 30), ("sum_GBps", lambda: x.sum(), 4 << 30)):
            fn(); torch.cuda.synchronize(device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(5):
                e0.record(); fn(); e1.record(); torch.cuda.synchronize(device)
                ts.append(e0.elapsed_time(e1))
            out[name] = round(nbytes / (min(ts) * 1e-3) / 1e9, 1)
        # page-table / TLB reach: 32 M random 4-byte reads over the 4 GiB buffer, and 32 M random 4-byte writes
        idx = torch.randint(0, 1 << 30, (1 << 25,), device=device, dtype=torch.int64)
        val = torch.ones(1 << 25, dtype=torch.int32, device=device)
        for name, fn in (("gather_Gops", lambda: x[idx]), ("scatter_Gops", lambda: x.index_copy_(0, idx, val))):
            fn(); torch.cuda.synchronize(device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(5):
                e0.record(); fn(); e1.record(); torch.cuda.synchronize(device)
                ts.append(e0.elapsed_time(e1))
            out[name] = round((1 << 25) / (min(ts) * 1e-3) / 1e9, 2)
        del x, y, idx, val
        torch.cuda.empty_cache()
        return out
    streams = stream_rates()
    rec = {"tag": args.tag, "stream_rates": streams, "pid": os.getpid(), "time": time.time(), "clocks_before": clocks(), "reads": int(n), "phases": []}
    keep = []
    plan = [None] + [int(x) for x in args.pads.split(",") if x != ""] + ["clone"] * args.clone_reads
    for i, pad in enumerate(plan):
        if pad == "clone":
            keep.append((packed, lens))
            packed, lens = packed.clone(), lens.clone()
            torch.cuda.synchronize(device)
            launches(2)
        elif pad is not None:
            eng.debug_route_realloc(pad)
            launches(2)
        rt, pb = launches(args.launches)
        tr = eng.route_trace()
        cnt = eng.stats().counters
        ph = {"realloc_pad": pad, "arena": hex(tr["arena"]), "arena_mod_2MiB": tr["arena"] % (2 << 20), "packed": hex(tr["packed"]),
              "filter": hex(tr["filter"]), "flags": hex(tr["flags"]), "P": tr["P"], "cap": tr["cap"],
              "filter_passes": int(cnt[7]), "candidates": int(cnt[3]), "rt_debug": os.environ.get("MLST_RT_DEBUG"),
              "route_ms": {"min": round(min(rt), 4), "median": round(float(np.median(rt)), 4), "max": round(max(rt), 4), "all": [round(v, 4) for v in rt]},
              "probe_ms": {"min": round(min(pb), 4), "median": round(float(np.median(pb)), 4), "max": round(max(pb), 4)},
              "producer": wg_summary(tr, 0), "consumer": wg_summary(tr, 1)}
        rec["phases"].append(ph)
        if i == 0:
            np.save(os.path.join(os.path.dirname(args.out), "route_trace_%s.npy" % (args.tag or str(os.getpid()))), tr["wg"])
    if args.burst > 0:
        import threading
        paths = hwmon_paths()
        acc, stop = [], threading.Event()
        th = threading.Thread(target=sample_hwmon, args=(paths, stop, acc), daemon=True)
        eng.set_profiling(0)
        th.start()
        time.sleep(0.3)                      # idle baseline
        t0 = time.time()
        n_l = 0
        while time.time() - t0 < args.burst:
            for _ in range(20):
                eng.reset_sample()
                eng.submit_packed_device(packed.data_ptr(), qrows.data_ptr(), lens.data_ptr(), n, w.wpr, w.qstride)
                n_l += 1
            eng.synchronize()
        t1 = time.time()
        time.sleep(0.3)
        stop.set(); th.join()
        a = np.array(acc, dtype=np.float64)
        busy = {}
        for k in range(len(paths)):
            m = (a[:, 1] == k) & (a[:, 0] >= t0 + 0.3) & (a[:, 0] <= t1)
            idle = (a[:, 1] == k) & (a[:, 0] < t0)
            if m.any():
                busy[paths[k]["dir"].split("/")[4]] = {"n": int(m.sum()), "sclk_MHz_min_med_max": [float(np.min(a[m, 2])) / 1e6, float(np.median(a[m, 2])) / 1e6, float(np.max(a[m, 2])) / 1e6],
                                                       "mclk_MHz_med": float(np.median(a[m, 3])) / 1e6, "power_W_med_max": [float(np.median(a[m, 4])) / 1e6, float(np.max(a[m, 4])) / 1e6],
                                                       "temp_C_med": float(np.median(a[m, 5])) / 1e3, "temp_mem_C_med": float(np.median(a[m, 6])) / 1e3,
                                                       "idle_sclk_MHz": float(np.median(a[idle, 2])) / 1e6 if idle.any() else None, "idle_power_W": float(np.median(a[idle, 4])) / 1e6 if idle.any() else None}
        # this process's card: the one whose power rose most between the idle baseline and the burst (the sysfs of the box
        # shows all eight GPUs of the host, other tenants' included)
        hot = max(busy, key=lambda c: busy[c]["power_W_med_max"][0] - (busy[c]["idle_power_W"] or 0.0)) if busy else None
        rec["burst"] = {"seconds": round(t1 - t0, 2), "launches": n_l, "ms_per_sieve_launch_back_to_back": round((t1 - t0) / max(1, n_l) * 1e3, 4),
                        "busiest_card": hot, "cards": busy if hot is None else {hot: busy[hot]}, "n_cards": len(busy),
                        "other_cards_busy_power_W": sorted(round(busy[c]["power_W_med_max"][0]) for c in busy if c != hot)}
    rec["clocks_after"] = clocks()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "a") as f:
        f.write(json.dumps(rec) + "\n")
    print(json.dumps({"tag": args.tag, "route_ms_median": [p["route_ms"]["median"] for p in rec["phases"]], "probe_ms_median": [p["probe_ms"]["median"] for p in rec["phases"]],
                      "filter_passes": [p["filter_passes"] for p in rec["phases"]], "burst": rec.get("burst"), "stream_rates": rec.get("stream_rates"),
                      "arena": [p["arena"] for p in rec["phases"]], "xcc0": [p["producer"]["xcc_of_wg0"] for p in rec["phases"]]}))


if __name__ == "__main__":
    main()
