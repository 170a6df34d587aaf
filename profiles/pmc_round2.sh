#!/bin/bash
# usage (on the GPU box, from the repo root): bash profiles/pmc_round2.sh <tag> [extra bench.py flags]
# One rocprofv3 --pmc pass per counter group over a short serial bench.py run (cfg3 headline workload only), then the
# per-kernel means of every counter as JSON: gpurun_out/<tag>/pmc_counters.json.  Counter passes never carry trace flags.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
python3 $ROOT/__graft_entry__.py > /dev/null          # build outside the profiler
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
           "FETCH_SIZE TCC_HIT_sum" \
           "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d "$OUT/pass$i" --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --pipeline 1 --no-secondary --min-seconds 0.01 "$@" > "$OUT/pass$i.log" 2>&1 || { tail -5 "$OUT/pass$i.log"; exit 1; }
done
python3 $ROOT/profiles/pmc_extract.py "$OUT" > "$OUT/pmc_counters.json"
python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --pipeline 1 --no-secondary --min-seconds 0.01 "$@" > "$OUT/bench_for_pmc.json" 2> /dev/null      # the same command unprofiled: counters of the line
find "$OUT" -name "*.csv" -size +2M -delete            # the raw per-dispatch tables stay on the box
