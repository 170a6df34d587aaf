#!/usr/bin/env python3
"""Kernel timeline of the last run in a rocprofv3 --kernel-trace results database: python profiles/timeline.py <results.db> [window_ms] [name filter,...]"""
import collections
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 85.0
flt = sys.argv[3].split(",") if len(sys.argv) > 3 else None
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
t_end = rows[-1][2]
sel = [r for r in rows if r[1] > t_end - win * 1e6]
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e, st in sel:
    k = n.split("(")[0][:44]; agg[k][0] += 1; agg[k][1] += (e - s) / 1e6
for k, (n, ms) in sorted(agg.items(), key=lambda x: -x[1][1])[:24]:
    print("%-46s %4d %8.3f ms  avg %.3f" % (k, n, ms, ms / n))
t0 = sel[0][1]
print("timeline (ms):")
for n, s, e, st in sel:
    k = n.split("(")[0]
    if (flt and any(f in k for f in flt)) or (not flt and (e - s) > 150e3):
        print("%-26s %8.3f -> %8.3f  (%.3f) stream %s" % (k[:26], (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, st))
