#!/usr/bin/env python3
"""For rocprofv3 --pmc: the same CsrMV (512^3, row-tile kernel) with x in the first-allocated 3 GiB buffer ("A") and in a
1 GiB + 4 MiB buffer allocated later ("B"); launches go A A A B B B A A A B B B, the event times are printed so that the
dispatch order can be matched with the counter rows."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib  # noqa: E402
from conjugategradient_amd.solver import VectorDouble, VectorInt  # noqa: E402


def main():
    n = 512
    N = n**3
    L = _lib.lib()
    _lib.require_gpu()
    L.SetDevice(0)
    blas, sparse, descr = L.CreateBlas(), L.CreateSparse(), L.CreateMatDescr()
    nnz = L.MgcgPoissonNnz(n, n, n, 0, n)
    ev0, ev1 = L.MgcgEventCreate(), L.MgcgEventCreate()
    e, c, r = VectorDouble(nnz), VectorInt(nnz), VectorInt(N + 1)
    assert L.MgcgGeneratePoisson(e.Ptr, r.Ptr, c.Ptr, n, n, n, 0, n) == 0
    XA, Y = VectorDouble(N + (1 << 28)), VectorDouble(N + (1 << 28))
    XB = VectorDouble(N + (1 << 19))
    L.MgcgFill(XA.Ptr, 1.0)
    L.MgcgFill(XB.Ptr, 1.0)
    for rep in range(2):
        for label, xv in (("A", XA), ("B", XB)):
            for k in range(3):
                L.MgcgEventRecord(ev0)
                L.CsrMV(sparse, descr, Y.ToRawPtr(), e.ToRawPtr(), r.ToRawPtr(), c.ToRawPtr(), xv.ToRawPtr(), nnz, N, N, 1.0, 0.0)
                L.MgcgEventRecord(ev1)
                print(f"x in {label}: {L.MgcgEventElapsedMs(ev0, ev1):.3f} ms", flush=True)
    _lib.check("placement_pmc")


if __name__ == "__main__":
    main()
