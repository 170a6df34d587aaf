#!/bin/bash
# usage (GPU box, repo root): bash profiles/pmc_extend.sh <tag> <workload> <variant> [<variant> ...]
# SQ counters of k_extend alone (profiles/extend_bench.py, one variant per process so that the kernel name is unambiguous):
# one rocprofv3 --pmc pass per counter group and variant, no trace flags -> gpurun_out/<tag>/<variant>.json
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; WL=$2; shift; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
python3 $ROOT/__graft_entry__.py > /dev/null
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  vn=$(echo "$v" | tr '=+' '__')
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"; do
    i=$((i+1))
    rocprofv3 --pmc $grp -d "$OUT/$vn/pass$i" --output-format csv -- python3 $ROOT/profiles/extend_bench.py --workload $WL --variants "$v" --launches 3 > "$OUT/$vn.pass$i.log" 2>&1 || { tail -5 "$OUT/$vn.pass$i.log"; exit 1; }
  done
  python3 $ROOT/profiles/pmc_extract.py "$OUT/$vn" k_extend > "$OUT/$vn.json"
  find "$OUT/$vn" -name "*.csv" -size +2M -delete
done
