#!/bin/bash
# usage (GPU box, repo root): bash profiles/trace_round2.sh <tag>
# rocprofv3 --kernel-trace --stats of (a) the serial headline run (every kernel alone on the GPU: the durations bench.py's
# `roofline` uses) and (b) the default pipelined command; the kernel tables end up in gpurun_out/<tag>/.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
python3 $ROOT/__graft_entry__.py > /dev/null          # build outside the profiler
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/serial" --output-format csv -- python3 $ROOT/bench.py --pipeline 1 --no-secondary --cpu-seconds 0 --steps 10 --warmup 2 > "$OUT/bench_under_rocprof_serial.json" 2> "$OUT/serial.err"
rocprofv3 --kernel-trace --stats -d "$OUT/pipelined" --output-format csv -- python3 $ROOT/bench.py --cpu-seconds 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/pipelined.err"
for m in serial pipelined; do
  f=$(find "$OUT/$m" -name "*kernel_stats.csv" | head -1)
  cp "$f" "$OUT/kernel_stats_$m.csv"
  python3 $ROOT/profiles/summarize.py "$f" > "$OUT/kernel_summary_$m.md"
done
find "$OUT" -name "*kernel_trace.csv" -delete          # the per-dispatch traces stay on the box
