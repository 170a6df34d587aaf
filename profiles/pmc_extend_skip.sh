#!/bin/bash
# usage (GPU box): bash profiles/pmc_extend_skip.sh <tag> <workload>: VALU / SALU / LDS instructions of k_extend with phases switched
# off in the profiling build (build_variants/libmlst_trace.so): 0 = all, 1 = no summaries, 2 = no composition, 3 = neither
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; WL=$2
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MLST_LIB=$ROOT/build_variants/libmlst_trace.so MLST_LIB_ALLOW_MISSING=1
for sk in 0 1 2 3; do
  MLST_X_SKIP=$sk rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD -d "$OUT/skip$sk" --output-format csv -- python3 $ROOT/profiles/extend_bench.py --workload $WL --variants hap --launches 3 > "$OUT/skip$sk.log" 2>&1 || { tail -5 "$OUT/skip$sk.log"; exit 1; }
  python3 $ROOT/profiles/pmc_extract.py "$OUT/skip$sk" k_extend > "$OUT/skip$sk.json"
  find "$OUT/skip$sk" -name "*.csv" -size +2M -delete
done
python3 - <<PY
import json
for sk in range(4):
    d=json.load(open("$OUT/skip%d.json"%sk))
    for k,c in d.items():
        print("skip",sk,k,{n:round(x["mean"]/1e6,1) for n,x in c.items() if isinstance(x,dict)})
PY
