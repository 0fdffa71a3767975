#!/usr/bin/env python3
"""Where does the +-7 % of the CSR SpMV come from: x, y, or the PAIR?  7-point Poisson 512^3, CsrMV (row-tile kernel) on raw pointers.
One matrix; K allocations for x and K for y (each 4 MiB larger than the vector, so that sub-allocation offsets can be tried);
(1) every (x_i, y_j) pair at offset 0, (2) the best and the worst pair with y shifted by 0 .. 1.75 MiB, (3) the same with x shifted.
Prints one JSON object."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib  # noqa: E402
from conjugategradient_amd.solver import VectorDouble, VectorInt  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    N = n**3
    L = _lib.lib()
    _lib.require_gpu()
    L.SetDevice(0)
    blas, sparse, descr = L.CreateBlas(), L.CreateSparse(), L.CreateMatDescr()
    nnz = L.MgcgPoissonNnz(n, n, n, 0, n)
    ev0, ev1 = L.MgcgEventCreate(), L.MgcgEventCreate()
    e, c, r = VectorDouble(nnz), VectorInt(nnz), VectorInt(N + 1)
    assert L.MgcgGeneratePoisson(e.Ptr, r.Ptr, c.Ptr, n, n, n, 0, n) == 0
    pad = 4 << 20                      # bytes
    xs, ys = [], []
    for _ in range(K):                 # interleaved, as a solver allocates its vectors
        x, y = VectorDouble(N + pad // 8), VectorDouble(N + pad // 8)
        L.MgcgFill(x.Ptr, 1.0)
        xs.append(x); ys.append(y)

    def timed(xp, yp, reps=10):
        args = (sparse, descr, yp, e.ToRawPtr(), r.ToRawPtr(), c.ToRawPtr(), xp, nnz, N, N, 1.0, 0.0)
        for _ in range(2):
            L.CsrMV(*args)
        L.MgcgEventRecord(ev0)
        for _ in range(reps):
            L.CsrMV(*args)
        L.MgcgEventRecord(ev1)
        return L.MgcgEventElapsedMs(ev0, ev1) / reps

    out = {"grid": n, "K": K, "x_addresses": [hex(x.ToRawPtr()) for x in xs], "y_addresses": [hex(y.ToRawPtr()) for y in ys]}
    pairs = [[timed(xs[i].ToRawPtr(), ys[j].ToRawPtr()) for j in range(K)] for i in range(K)]
    out["pairs_ms_rows_x_cols_y"] = [[round(v, 4) for v in row] for row in pairs]
    flat = [(pairs[i][j], i, j) for i in range(K) for j in range(K)]
    best, worst = min(flat), max(flat)
    out["best"], out["worst"] = {"ms": best[0], "x": best[1], "y": best[2]}, {"ms": worst[0], "x": worst[1], "y": worst[2]}
    shifts = [k * (256 << 10) for k in range(8)]       # 0 .. 1.75 MiB in 256 KiB steps
    for name, (_, i, j) in (("best", best), ("worst", worst)):
        out[name]["y_shift_ms"] = [round(timed(xs[i].ToRawPtr(), ys[j].ToRawPtr() + s), 4) for s in shifts]
        out[name]["x_shift_ms"] = [round(timed(xs[i].ToRawPtr() + s, ys[j].ToRawPtr()), 4) for s in shifts]
    out["repeat_best_ms"] = round(timed(xs[best[1]].ToRawPtr(), ys[best[2]].ToRawPtr()), 4)
    out["repeat_worst_ms"] = round(timed(xs[worst[1]].ToRawPtr(), ys[worst[2]].ToRawPtr()), 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
