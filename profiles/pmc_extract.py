#!/usr/bin/env python3
"""Pull per-dispatch counter values of one kernel out of rocprofv3 --pmc counter_collection CSVs."""
import csv
import glob
import json
import sys


def main(root, kernel_prefix):
    out = {}
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_prefix in r.get("Kernel_Name", ""):
                out.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(json.dumps({k: {"n": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
