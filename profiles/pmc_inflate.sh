#!/bin/bash
# usage (GPU box, repo root): bash profiles/pmc_inflate.sh <tag> [reads]
# SQ counters of k_inflate alone (profiles/inflate_rate.py): one rocprofv3 --pmc pass per counter group, no trace flags
# -> gpurun_out/<tag>/inflate.json (means over the four launches on the same level-6 data)
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; N=${2:-1000000}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
python3 $ROOT/__graft_entry__.py > /dev/null
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d "$OUT/inflate/pass$i" --output-format csv -- python3 $ROOT/profiles/inflate_rate.py $N 6 > "$OUT/inflate.pass$i.log" 2>&1 || { tail -5 "$OUT/inflate.pass$i.log"; exit 1; }
done
python3 $ROOT/profiles/pmc_extract.py "$OUT/inflate" k_inflate > "$OUT/inflate.json"
find "$OUT/inflate" -name "*.csv" -size +2M -delete
tail -3 "$OUT/inflate.pass1.log"
cat "$OUT/inflate.json"
