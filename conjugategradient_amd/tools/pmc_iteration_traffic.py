#!/usr/bin/env python3
"""HBM-side bytes per solver iteration from a profile_bench.sh directory (two rocprofv3 --pmc passes of one bench command): every
dispatch's read / write bytes (128 / 64 / 32-byte request counters, profile_summarize.py's formula) summed per kernel over the whole
run and divided by the number of iterations that really ran -- dispatches of the loop's SpMV (spmv_rowtile_kernel<1, ...>) that moved
data; launches enqueued behind the stop flag return at once and count nothing.  Set-up kernels are listed apart.
Usage: pmc_iteration_traffic.py DIR [required_bytes_per_iteration]"""
import collections
import csv
import glob
import json
import os
import sys

LOOP = ("spmv_rowtile_kernel", "update_xp", "update_r", "prolong", "restrict", "jacobi_first", "finalize", "reduce", "dot_kernel", "spmv_")


def load(root):
    rows = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                d = rows[r["Dispatch_Id"]]
                d["name"] = r["Kernel_Name"].split("(")[0]
                d[r["Counter_Name"]] = float(r["Counter_Value"])
    return rows


def main():
    root = sys.argv[1]
    required = float(sys.argv[2]) if len(sys.argv) > 2 else None
    rd, wr = load(os.path.join(root, "pmc_rd")), load(os.path.join(root, "pmc_wr"))
    rbytes = lambda d: 128 * d.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * d.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * d.get("TCC_EA0_RDREQ_32B_sum", 0)   # noqa: E731
    wbytes = lambda d: 64 * d.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (d.get("TCC_EA0_WRREQ_sum", 0) - d.get("TCC_EA0_WRREQ_64B_sum", 0))          # noqa: E731
    tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
    # iterations that really ran = dispatches of the r update that moved data (one per iteration; the loop's SpMV is ALSO launched by the
    # library's placement draw in the warm-up -- 2 stages x 4 candidates x 7 launches -- so its dispatches no longer count iterations)
    biggest = max((rbytes(d) for d in rd.values() if "update_r_kernel" in d["name"]), default=0.0)
    its = sum(1 for d in rd.values() if "update_r_kernel" in d["name"] and rbytes(d) > 0.5 * biggest)
    big_spmv = max((rbytes(d) for d in rd.values() if "spmv_rowtile_kernel<1" in d["name"]), default=0.0)
    spmv_moving = sum(1 for d in rd.values() if "spmv_rowtile_kernel<1" in d["name"] and rbytes(d) > 0.5 * big_spmv)
    for d in rd.values():
        tot[d["name"]][0] += 1
        tot[d["name"]][1] += rbytes(d)
    for d in wr.values():
        tot[d["name"]][2] += wbytes(d)
    loop, setup = {}, {}
    for k, (n, r, w) in tot.items():
        rec = {"dispatches": n, "read_gb_total": r / 1e9, "write_gb_total": w / 1e9, "gb_per_iteration": (r + w) / max(its, 1) / 1e9}
        if "spmv_rowtile_kernel<1" in k:             # one launch per iteration: bytes per launch that moved data (the draw's launches move the same bytes)
            rec["gb_per_iteration"] = (r + w) / max(spmv_moving, 1) / 1e9
            rec["dispatches_that_moved_data"] = spmv_moving
        is_loop = any(m in k for m in LOOP) and "galerkin" not in k and not (k.endswith("<2, 7, false, 0>") or k.endswith("<0, 7, false, 0>"))   # (init residual / bare export)
        (loop if is_loop else setup)[k.replace("void mgcg::", "").replace("mgcg::", "")] = rec
    per_it = sum(v["gb_per_iteration"] for v in loop.values())
    out = {"iterations_that_ran": its, "loop_gb_per_iteration": per_it, "loop_kernels": dict(sorted(loop.items(), key=lambda kv: -kv[1]["gb_per_iteration"])),
           "other_kernels_gb_total": {k: v["read_gb_total"] + v["write_gb_total"] for k, v in sorted(setup.items(), key=lambda kv: -(kv[1]["read_gb_total"] + kv[1]["write_gb_total"])) if v["read_gb_total"] + v["write_gb_total"] > 0.5}}
    if required:
        out["required_gb_per_iteration"] = required / 1e9
        out["traffic_over_required"] = per_it / (required / 1e9)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
