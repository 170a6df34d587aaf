#!/bin/bash
# usage (GPU box, repo root): bash profiles/route_modes.sh <n_processes> [route_modes.py flags]  -- consecutive processes on one box
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-8}; shift || true
mkdir -p $ROOT/gpurun_out
python3 $ROOT/__graft_entry__.py > /dev/null
for i in $(seq 1 $N); do
  timeout -k 10 240 python3 $ROOT/profiles/route_modes.py --tag p$i "$@" 2>> $ROOT/gpurun_out/route_modes.err
done
