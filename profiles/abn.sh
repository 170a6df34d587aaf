#!/bin/bash
# usage (GPU box, repo root): bash profiles/abn.sh <rounds> <lib1.so> [<lib2.so> ...]   -- alternating processes on one box, any number of builds
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
mkdir -p $ROOT/gpurun_out
python3 $ROOT/__graft_entry__.py > /dev/null
for i in $(seq 1 $N); do
  for L in "$@"; do
    MLST_LIB=$L MLST_LIB_ALLOW_MISSING=1 timeout -k 10 400 python3 $ROOT/bench.py --no-secondary --cpu-seconds 0 --steps ${STEPS:-10} --warmup 3 --min-seconds 0.5 2>> $ROOT/gpurun_out/ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_ms_per_launch_isolated']
print('$L', 'value', d['value'], 'ms/step', d['ms_per_step'], 'serial', d['serial_ms_per_step'], {x: round(k[x],3) for x in k}, flush=True)
" | tee -a $ROOT/gpurun_out/ab.log
  done
done
