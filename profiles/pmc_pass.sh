#!/bin/bash
# usage: profiles/pmc_pass.sh <outdir under gpurun_out> "<counter list>"   (one rocprofv3 --pmc pass of a short bench.py run)
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $2 -d "$OUT" --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --pipeline 1 > "$OUT/bench.log" 2>&1 || { tail -5 "$OUT/bench.log"; exit 1; }
