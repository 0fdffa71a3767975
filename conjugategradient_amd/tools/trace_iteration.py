#!/usr/bin/env python3
"""One steady-state iteration out of a `rocprofv3 --kernel-trace --output-format csv` run: the kernels between two consecutive dispatches
of a marker kernel (default: the x/p update that ends every CG / MGCG iteration), taken from the MIDDLE of the run (the tail of a trace
holds launches enqueued behind the stop flag, which return at once), with durations, gaps, and the sums per kernel name.
Usage: trace_iteration.py DIR [--marker update_xp] [--which 0.5]"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    marker = sys.argv[sys.argv.index("--marker") + 1] if "--marker" in sys.argv else "update_xp"
    which = float(sys.argv[sys.argv.index("--which") + 1]) if "--which" in sys.argv else 0.5
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < 3:
        raise SystemExit("fewer than three dispatches of the marker kernel")
    # iterations that really ran: the marker's duration is above half of its longest
    longest = max(rows[i][1] - rows[i][0] for i in marks)
    live = [i for i in marks if rows[i][1] - rows[i][0] > 0.5 * longest]
    k = live[int(which * (len(live) - 1))]
    prev = max(i for i in marks if i < k)
    it = rows[prev + 1: k + 1]
    t0 = rows[prev][1]
    last = t0
    by = collections.OrderedDict()
    gaps = 0.0
    for s, e, name in it:
        short = name[-72:]
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - last) / 1e3:7.1f}  {short}")
        gaps += max(0.0, (s - last) / 1e3)
        a = by.setdefault(short, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
        last = max(last, e)
    print(f"-- iteration {(last - t0) / 1e3:.1f} us: {len(it)} kernels, {sum(v[1] for v in by.values()):.1f} us inside kernels, {gaps:.1f} us of gaps")
    for name, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"   {t:9.1f} us  x{n:<3d} {name}")


if __name__ == "__main__":
    main()
