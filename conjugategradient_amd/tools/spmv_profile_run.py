#!/usr/bin/env python3
"""Workload for rocprofv3 passes: a few launches of chosen SpMV variants on the 7-point Poisson n^3 matrix,
preceded by two Dot launches (exactly 16*N bytes read each) that calibrate the byte counters."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib  # noqa: E402
from conjugategradient_amd.solver import VectorDouble, VectorInt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--variants", default="1:128:0,1:128:4", help="kernel:rows:flags,...")
    ap.add_argument("--compression", type=int, default=0, help="MgcgSetMatrixCompression mode for every variant (0 off, 1 best, 2 per-nonzero codes)")
    a = ap.parse_args()
    L = _lib.lib()
    _lib.require_gpu()
    L.SetDevice(0)
    blas, sparse, descr = L.CreateBlas(), L.CreateSparse(), L.CreateMatDescr()
    n = a.grid
    N = n**3
    nnz = L.MgcgPoissonNnz(n, n, n, 0, n)
    e, c, r = VectorDouble(nnz), VectorInt(nnz), VectorInt(N + 1)
    assert L.MgcgGeneratePoisson(e.Ptr, r.Ptr, c.Ptr, n, n, n, 0, n) == 0
    x, y = VectorDouble(N), VectorDouble(N)
    L.MgcgFill(x.Ptr, 1.0)
    L.MgcgFill(y.Ptr, 2.0)
    L.MgcgDeviceSynchronize()
    for _ in range(2):
        L.Dot(blas, y.ToRawPtr(), x.ToRawPtr(), N)
    L.MgcgSetMatrixCompression(sparse, a.compression)
    for v in a.variants.split(","):
        k, rows, flags = (int(t) for t in v.split(":"))
        L.MgcgSetSpmvKernel(sparse, k)
        L.MgcgSetSpmvTuning(sparse, rows, flags, 0)
        L.MgcgSetSpmvPeriod(sparse, n * n if flags & 4 else 0)
        for _ in range(a.reps):
            L.CsrMV(sparse, descr, y.ToRawPtr(), e.ToRawPtr(), r.ToRawPtr(), c.ToRawPtr(), x.ToRawPtr(), nnz, N, N, 1.0, 0.0)
        L.MgcgDeviceSynchronize()
    _lib.check("profile run")
    print("algorithmic bytes per SpMV", 12 * nnz + 4 * (N + 1) + 16 * N, "dot bytes", 16 * N)


if __name__ == "__main__":
    main()
