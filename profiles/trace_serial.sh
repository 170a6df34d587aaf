#!/bin/bash
# usage (GPU box, repo root): bash profiles/trace_serial.sh <tag>   -- serial headline run under rocprofv3 --kernel-trace --stats
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1
mkdir -p "$OUT"
python3 $ROOT/__graft_entry__.py > /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/serial" --output-format csv -- python3 $ROOT/bench.py --pipeline 1 --no-secondary --cpu-seconds 0 --steps 10 --warmup 2 > "$OUT/bench_under_rocprof_serial.json" 2> "$OUT/serial.err"
f=$(find "$OUT/serial" -name "*kernel_stats.csv" | head -1)
python3 $ROOT/profiles/summarize.py "$f" > "$OUT/kernel_summary_serial.md"
find "$OUT" -name "*kernel_trace.csv" -delete
cat "$OUT/kernel_summary_serial.md"
