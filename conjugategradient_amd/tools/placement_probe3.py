#!/usr/bin/env python3
"""Do the CG loop's vector passes depend on WHERE their vectors live, as the SpMV does (placement_probe2.py)?  Xpay (y = x + beta y: two reads,
one write, 24 N bytes -- the shape of the r update) on every pair of K x and K y allocations of 512^3 doubles, and Solve2 (x += a p, r -= a Ap,
r.r: four reads, two writes) with K allocations of its written x.  Prints one JSON object."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib  # noqa: E402
from conjugategradient_amd.solver import VectorDouble  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    N = n**3
    L = _lib.lib()
    _lib.require_gpu()
    L.SetDevice(0)
    blas = L.CreateBlas()
    ev0, ev1 = L.MgcgEventCreate(), L.MgcgEventCreate()
    xs, ys = [], []
    for _ in range(K):
        x, y = VectorDouble(N), VectorDouble(N)
        L.MgcgFill(x.Ptr, 1.0); L.MgcgFill(y.Ptr, 0.5)
        xs.append(x); ys.append(y)

    def timed(fn, reps=10):
        for _ in range(2):
            fn()
        L.MgcgEventRecord(ev0)
        for _ in range(reps):
            fn()
        L.MgcgEventRecord(ev1)
        return L.MgcgEventElapsedMs(ev0, ev1) / reps

    out = {"grid": n, "K": K}
    out["xpay_ms_rows_x_cols_y"] = [[round(timed(lambda: L.Xpay(blas, ys[j].ToRawPtr(), xs[i].ToRawPtr(), N, 0.5)), 4) for j in range(K)] for i in range(K)]
    out["xpay_gbps_best_worst"] = [24 * N / min(min(r) for r in out["xpay_ms_rows_x_cols_y"]) / 1e6, 24 * N / max(max(r) for r in out["xpay_ms_rows_x_cols_y"]) / 1e6]
    # Solve2 with xs[0] as p, xs[1] as Ap, ys[0] as r and each of the other y allocations as the written x
    out["solve2_ms_by_x_allocation"] = [round(timed(lambda: L.Solve2(blas, 1e-3, ys[j].Ptr, xs[1].Ptr, xs[0].Ptr, ys[0].Ptr, N, 0)), 4) for j in range(1, K)]
    out["solve2_ms_by_r_allocation"] = [round(timed(lambda: L.Solve2(blas, 1e-3, ys[0].Ptr, xs[1].Ptr, xs[0].Ptr, ys[j].Ptr, N, 0)), 4) for j in range(1, K)]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
