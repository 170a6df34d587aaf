#!/usr/bin/env python3
"""pmc_counters.json (profiles/pmc_extract.py) + the bench line of the same workload -> profiles/round4/pmc_<workload>.json,
the per-launch figures bench.py's `roofline.traffic` / `roofline_extend` quote.

    python3 profiles/pmc_to_bench.py gpurun_out/<tag>/pmc_counters.json gpurun_out/<tag>/bench.json cfg3 > profiles/round4/pmc_cfg3.json

HBM bytes per launch = 2 x 1024 x FETCH_SIZE (gfx950 tallies 128-byte read requests at 64 bytes: MI355X_MICROARCH.md) +
1024 x WRITE_SIZE.  Counter passes run `bench.py --pipeline 1 --no-secondary`: every launch of these kernels in them is a
launch of the headline workload (setup launches of k_pack excepted)."""
import json
import sys

pmc = json.load(open(sys.argv[1]))
bench = json.load(open(sys.argv[2]))
name = sys.argv[3]
out = {"workload": name, "reads_per_launch": bench["config"]["reads_per_gpu"], "source": "rocprofv3 --pmc, profiles/pmc_round2.sh"}
alias = {"k_route": "k_route<10>", "k_route_probe": "k_route_probe<10>", "k_route_verify": "k_route_verify<10>", "k_extend": "k_extend_160", "k_sieve_q": "k_sieve_q<10>"}
for short, full in alias.items():
    c = pmc.get(full)
    if not c:
        continue
    e = {}
    if "hbm_read_bytes" in c:
        e["hbm_read_bytes_per_launch"] = int(c["hbm_read_bytes"])
        e["hbm_write_bytes_per_launch"] = int(c.get("hbm_write_bytes", 0))
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
    for k_in, k_out in (("SQ_INSTS_VALU", "valu_wave_instr_per_launch"), ("SQ_INSTS_LDS", "lds_wave_instr_per_launch"),
                        ("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict_cycles"), ("SQ_LDS_IDX_ACTIVE", "lds_active_cycles"),
                        ("SQ_WAVE_CYCLES", "wave_cycles"), ("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst_any")):
        if k_in in c:
            e[k_out] = int(c[k_in]["mean"])
    out[short] = e
if "k_extend" in out:
    n_items = bench["counters"]["items"]
    out["k_extend"]["pairs_per_launch"] = n_items * (bench["config"]["n_alleles"] // max(1, bench["config"]["n_loci"]))
print(json.dumps(out, indent=1))
