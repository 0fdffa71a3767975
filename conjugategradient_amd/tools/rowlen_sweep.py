#!/usr/bin/env python3
"""Which SpMV kernel family for which row length?  Banded |sin(i+j)| matrices (the MgcgMain generator) of width K and a
random SPD matrix, every kernel family timed on the same arrays.  Prints one line per (matrix, kernel)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib, problems  # noqa: E402
from conjugategradient_amd.solver import VectorDouble, VectorInt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nnz", type=int, default=60_000_000)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    L = _lib.lib()
    _lib.require_gpu()
    blas, sparse, descr = L.CreateBlas(), L.CreateSparse(), L.CreateMatDescr()
    ev0, ev1 = L.MgcgEventCreate(), L.MgcgEventCreate()
    systems = [(f"band {k}", lambda k=k: problems.mgcg_main(a.nnz // k, k)) for k in (8, 12, 16, 24, 32, 48, 64, 96, 160)]
    systems.append(("random ~31", lambda: problems.random_spd(a.nnz // 31, mean_upper=14.0, seed=1)))
    for name, build in systems:
        s = build()
        e, c, r = VectorDouble(s.nnz), VectorInt(s.nnz), VectorInt(s.Count + 1)
        e.CopyFrom(s.Elements, s.nnz); c.CopyFrom(s.ColumnIndeces, s.nnz); r.CopyFrom(s.RowOffsets, s.Count + 1)
        x, y = VectorDouble(s.Count), VectorDouble(s.Count)
        x.CopyFrom(np.cos(np.arange(s.Count) * 0.01), s.Count)
        algo = 12 * s.nnz + 4 * (s.Count + 1) + 16 * s.Count
        out = []
        for label, k in (("auto", 0), ("stream", 1), ("rows", 9), ("4/row", 4), ("8/row", 5), ("16/row", 6), ("32/row", 7), ("64/row", 8)):
            L.MgcgSetSpmvKernel(sparse, k)
            args = (sparse, descr, y.ToRawPtr(), e.ToRawPtr(), r.ToRawPtr(), c.ToRawPtr(), x.ToRawPtr(), s.nnz, s.Count, s.Count, 1.0, 0.0)
            L.CsrMV(*args)
            L.MgcgEventRecord(ev0)
            for _ in range(a.reps):
                L.CsrMV(*args)
            L.MgcgEventRecord(ev1)
            ms = L.MgcgEventElapsedMs(ev0, ev1) / a.reps
            out.append(f"{label} {ms:.3f} ms ({algo / ms / 1e6:.0f} GB/s)")
        _lib.check("sweep")
        print(f"{name:12s} rows {s.Count:9d} nnz/row {s.nnz / s.Count:6.1f} | " + " | ".join(out), flush=True)
        for v in (e, c, r, x, y):
            v.Dispose()


if __name__ == "__main__":
    main()
