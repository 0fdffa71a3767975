#!/usr/bin/env python3
"""Collapse rocprofv3 --pmc CSV output (one row per dispatch and counter) into a per-kernel table."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(root):
    table = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                short = k.split("(")[0][-70:]
                table[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, cs in sorted(table.items()):
        out[k] = {c: {"n": len(v), "mean": sum(v) / len(v)} for c, v in sorted(cs.items())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
