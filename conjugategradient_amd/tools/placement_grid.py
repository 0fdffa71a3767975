#!/usr/bin/env python3
"""Is the spread of CsrMV times over x allocations (placement_shift.py sizes) tied to the power-of-two plane of 512^3 (the far
neighbours of a row sit exactly +-2 MiB from it in x)?  Same test on 512x512x512 and on 504x520x512 (planes of 2 MiB - 512 B... no
power of two anywhere), six x buffers each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib  # noqa: E402
from conjugategradient_amd.solver import VectorDouble, VectorInt  # noqa: E402


def main():
    L = _lib.lib()
    _lib.require_gpu()
    L.SetDevice(0)
    blas, sparse, descr = L.CreateBlas(), L.CreateSparse(), L.CreateMatDescr()
    ev0, ev1 = L.MgcgEventCreate(), L.MgcgEventCreate()
    for (nx, ny, nz) in ((512, 512, 512), (504, 520, 512), (512, 512, 512), (496, 528, 512)):
        N = nx * ny * nz
        nnz = L.MgcgPoissonNnz(nx, ny, nz, 0, nz)
        algo = 12 * nnz + 4 * (N + 1) + 16 * N
        e, c, r = VectorDouble(nnz), VectorInt(nnz), VectorInt(N + 1)
        assert L.MgcgGeneratePoisson(e.Ptr, r.Ptr, c.Ptr, nx, ny, nz, 0, nz) == 0
        y = VectorDouble(N)
        out = []
        xs = []
        for k in range(6):
            x = VectorDouble(N)
            xs.append(x)
            L.MgcgFill(x.Ptr, 1.0)
            args = (sparse, descr, y.ToRawPtr(), e.ToRawPtr(), r.ToRawPtr(), c.ToRawPtr(), x.ToRawPtr(), nnz, N, N, 1.0, 0.0)
            for _ in range(2):
                L.CsrMV(*args)
            t = []
            for _ in range(3):
                L.MgcgEventRecord(ev0)
                for _ in range(6):
                    L.CsrMV(*args)
                L.MgcgEventRecord(ev1)
                t.append(L.MgcgEventElapsedMs(ev0, ev1) / 6)
            ms = sorted(t)[1]
            out.append(algo / ms / 1e6 / 8000)
        print(f"{nx}x{ny}x{nz}: fraction of 8 TB/s per x buffer: " + " ".join(f"{f:.3f}" for f in out), flush=True)
        for v in xs + [e, c, r, y]:
            v.Dispose()
    _lib.check("placement_grid")


if __name__ == "__main__":
    main()
