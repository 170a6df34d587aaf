#!/bin/bash
# usage (GPU box, repo root): bash profiles/ab.sh <rounds> <libA.so> <libB.so> [bench flags]   -- alternating processes on one box
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; A=$2; B=$3; shift 3
mkdir -p $ROOT/gpurun_out
python3 $ROOT/__graft_entry__.py > /dev/null
for i in $(seq 1 $N); do
  for L in $A $B; do
    MLST_LIB=$L MLST_LIB_ALLOW_MISSING=1 timeout -k 10 400 python3 $ROOT/bench.py --no-secondary --cpu-seconds 0 --steps 10 --warmup 3 --min-seconds 0.5 "$@" 2>> $ROOT/gpurun_out/ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_ms_per_launch_isolated']
print('$L', 'value', d['value'], 'ms/step', d['ms_per_step'], 'serial', d['serial_ms_per_step'], {x: round(k[x],3) for x in k}, flush=True)
" | tee -a $ROOT/gpurun_out/ab.log
  done
done
