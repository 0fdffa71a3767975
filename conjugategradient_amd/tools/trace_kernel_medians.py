#!/usr/bin/env python3
"""Median duration per kernel name over the dispatches that really ran (longer than half the longest of that name), from a
`rocprofv3 --kernel-trace --output-format csv` directory: launches enqueued behind the stop flag return at once and would
dilute an average.  Usage: trace_kernel_medians.py DIR [substring]"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    t = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                t[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(t.items(), key=lambda kv: -sum(kv[1])):
        if want not in k:
            continue
        big = [x for x in v if x > 0.5 * max(v)]
        big.sort()
        print(f"{k[-64:]:64s} ran {len(big):4d} of {len(v):4d}  median {big[len(big) // 2]:9.1f} us  min {big[0]:9.1f}  max {big[-1]:9.1f}")


if __name__ == "__main__":
    main()
