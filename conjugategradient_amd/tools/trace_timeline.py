#!/usr/bin/env python3
"""Timeline of the last `--kernels` kernels of a `rocprofv3 --kernel-trace --output-format csv` run: start offset, duration, gap to the
previous kernel's end on the device (negative = overlapped), queue.  Usage: trace_timeline.py DIR [--kernels 40]"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    k = int(sys.argv[sys.argv.index("--kernels") + 1]) if "--kernels" in sys.argv else 40
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_trace.csv under " + d)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    rows = rows[-k:]
    t0 = rows[0][0]
    last_end = None
    for s, e, name, q in rows:
        gap = "" if last_end is None else f"{(s - last_end) / 1e3:8.1f}"
        short = name.split("(")[0][-70:]
        print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {gap:>8}  q{q}  {short}")
        last_end = max(e, last_end) if last_end is not None else e


if __name__ == "__main__":
    main()
