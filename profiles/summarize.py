#!/usr/bin/env python3
"""Filter a rocprofv3 --kernel-trace --stats kernel_stats.csv down to this engine's kernels (k_*)."""
import csv
import sys


def main(path):
    rows = [r for r in csv.DictReader(open(path)) if r["Name"].startswith("k_") or r["Name"].startswith("void k_")]
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    print("| kernel | calls | avg us | min us | max us | share of engine kernels |")
    print("|---|---|---|---|---|---|")
    for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"])):
        print("| %s | %s | %.1f | %.1f | %.1f | %.1f %% |" % (r["Name"].replace("void ", "").split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                           float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                           100.0 * int(r["TotalDurationNs"]) / tot))


if __name__ == "__main__":
    main(sys.argv[1])
