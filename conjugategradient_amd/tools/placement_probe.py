#!/usr/bin/env python3
"""Does WHERE the arrays live change what the CSR SpMV achieves?  7-point Poisson 512^3, CsrMV (row-tile kernel).
(a) one matrix, eight different (x, y) vector pairs; (b) one (x, y) pair, the matrix generated again in newly allocated
arrays (the old ones kept alive in between so the addresses really differ).  Prints time and device addresses."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib  # noqa: E402
from conjugategradient_amd.solver import VectorDouble, VectorInt  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    N = n**3
    L = _lib.lib()
    _lib.require_gpu()
    L.SetDevice(0)
    blas, sparse, descr = L.CreateBlas(), L.CreateSparse(), L.CreateMatDescr()
    nnz = L.MgcgPoissonNnz(n, n, n, 0, n)
    algo = 12 * nnz + 4 * (N + 1) + 16 * N
    ev0, ev1 = L.MgcgEventCreate(), L.MgcgEventCreate()

    def matrix():
        e, c, r = VectorDouble(nnz), VectorInt(nnz), VectorInt(N + 1)
        assert L.MgcgGeneratePoisson(e.Ptr, r.Ptr, c.Ptr, n, n, n, 0, n) == 0
        return e, c, r

    def timed(m, x, y):
        e, c, r = m
        args = (sparse, descr, y.ToRawPtr(), e.ToRawPtr(), r.ToRawPtr(), c.ToRawPtr(), x.ToRawPtr(), nnz, N, N, 1.0, 0.0)
        for _ in range(3):
            L.CsrMV(*args)
        out = []
        for _ in range(5):
            L.MgcgEventRecord(ev0)
            for _ in range(8):
                L.CsrMV(*args)
            L.MgcgEventRecord(ev1)
            out.append(L.MgcgEventElapsedMs(ev0, ev1) / 8)
        return sorted(out)[2]

    def addr(v):
        return f"{v.ToRawPtr() or 0:#014x}"

    m = matrix()
    print("matrix at", addr(m[0]), addr(m[1]), addr(m[2]), flush=True)
    vecs = []
    for k in range(8):
        x, y = VectorDouble(N), VectorDouble(N)
        L.MgcgFill(x.Ptr, 1.0)
        vecs.append((x, y))
        ms = timed(m, x, y)
        print(f"(a) vectors {k}: x {addr(x)} y {addr(y)}  {ms:.3f} ms  {algo / ms / 1e6 / 8000:.3f}", flush=True)
    for x, y in vecs[1:]:
        x.Dispose(); y.Dispose()
    x, y = vecs[0]
    keep = [m]
    for k in range(4):
        m2 = matrix()
        keep.append(m2)
        ms = timed(m2, x, y)
        print(f"(b) matrix {k}: {addr(m2[0])} {addr(m2[1])} {addr(m2[2])}  {ms:.3f} ms  {algo / ms / 1e6 / 8000:.3f}", flush=True)
    ms = timed(keep[0], x, y)
    print(f"(b) first matrix again: {ms:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
