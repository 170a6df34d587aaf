#!/bin/bash
# usage (GPU box, repo root): bash profiles/depth_sweep.sh  -- pipeline depth of the headline (engines per GPU), one process each
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
python3 $ROOT/__graft_entry__.py > /dev/null
for d in 3 4 5 6 4; do
  timeout -k 10 400 python3 $ROOT/bench.py --no-secondary --cpu-seconds 0 --steps 10 --warmup 3 --min-seconds 0.5 --pipeline $d 2>> $ROOT/gpurun_out/depth.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('depth $d', 'value', d['value'], 'ms/step', d['ms_per_step'], 'route', d['kernel_ms_per_launch_isolated']['sieve_route'], flush=True)"
done
