#!/bin/bash
# usage (GPU box): bash profiles/pmc_sieve_ab.sh <tag> <variant> [<variant> ...]
# HBM bytes and instruction counts of the routed sieve's kernels for variants of the engine (environment switches as in
# profiles/extend_bench.py, e.g. hap  MLST_RT_SLICE=3145728): one rocprofv3 --pmc pass per counter group and variant,
# no trace flags; per kernel SUMS over the launches of one submission -> gpurun_out/<tag>/<variant>.json
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
python3 $ROOT/__graft_entry__.py > /dev/null
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  vn=$(echo "$v" | tr '=+' '__')
  i=0
  for grp in "FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "WRITE_SIZE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU"; do
    i=$((i+1))
    rocprofv3 --pmc $grp -d "$OUT/$vn/pass$i" --output-format csv -- python3 $ROOT/profiles/extend_bench.py --workload cfg3 --variants "$v" --launches 2 > "$OUT/$vn.pass$i.log" 2>&1 || { tail -5 "$OUT/$vn.pass$i.log"; exit 1; }
  done
  python3 - "$OUT/$vn" > "$OUT/$vn.json" <<'PY'
import csv, glob, json, re, sys
root = sys.argv[1]
tot, n = {}, {}
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        m = re.match(r"(?:void )?(k_route_probe|k_route_verify|k_route|k_flag_compact)", kn)
        if not m:
            continue
        k = m.group(1)
        tot.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] = n.get((k, r["Counter_Name"]), 0) + 1
subs = 3      # extend_bench: one statistics submission + --launches 2
out = {}
for k, cs in tot.items():
    out[k] = {c: v / subs for c, v in cs.items()}
    out[k]["launches_per_submission"] = max(n[(k, c)] for c in cs) / subs
    if "FETCH_SIZE" in cs:
        out[k]["hbm_read_bytes_per_submission"] = 2 * 1024 * cs["FETCH_SIZE"] / subs      # gfx950: 128-byte requests tallied at 64 bytes
    if "WRITE_SIZE" in cs:
        out[k]["hbm_write_bytes_per_submission"] = 1024 * cs["WRITE_SIZE"] / subs
print(json.dumps(out, indent=1))
PY
  find "$OUT/$vn" -name "*.csv" -size +2M -delete
done
