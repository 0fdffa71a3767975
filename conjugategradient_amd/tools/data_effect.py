#!/usr/bin/env python3
"""Does the DATA in x change what the CSR SpMV achieves?  7-point Poisson 512^3, the CsrMV export (row-tile kernel), x filled with:
ones (y = 0 away from the boundary), a checkerboard of +-1 (y = +-12: two values), uniform random numbers, the CG iterate after 50
iterations.  Same kernel, same addresses, same byte counts."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib  # noqa: E402
from conjugategradient_amd.parallel import ConjugateGradientRankGpu  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    N = n**3
    L = _lib.lib()
    _lib.require_gpu()
    cg = ConjugateGradientRankGpu(N, 7, 0, 10**9, 1e-8, rank=0, world=1, device=0)
    cg.InitializePoisson(n, n, n)
    nnz = cg.part.elementCount
    algo = 12 * nnz + 4 * (N + 1) + 16 * N
    ev0, ev1 = L.MgcgEventCreate(), L.MgcgEventCreate()
    ptrs = lambda xv: (cg.vectorAp.ToRawPtr(), cg.vectorElements.ToRawPtr(), cg.vectorRowOffsets.ToRawPtr(), cg.vectorColumnIndeces.ToRawPtr(), xv.ToRawPtr())

    def timed(label, xv):
        for _ in range(3):
            L.CsrMV(cg.cusparse, cg.matDescr, *ptrs(xv), nnz, N, N, 1.0, 0.0)
        out = []
        for _ in range(5):
            L.MgcgEventRecord(ev0)
            for _ in range(10):
                L.CsrMV(cg.cusparse, cg.matDescr, *ptrs(xv), nnz, N, N, 1.0, 0.0)
            L.MgcgEventRecord(ev1)
            out.append(L.MgcgEventElapsedMs(ev0, ev1) / 10)
        ms = sorted(out)[len(out) // 2]
        print(f"x = {label:34s} {ms:.3f} ms  {algo / ms / 1e6:7.1f} GB/s  {algo / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)

    x = cg.vectorP
    host = np.empty(N)
    for label, fill in (("ones", lambda: host.fill(1.0)),
                        ("checkerboard +-1", lambda: host.__setitem__(slice(None), 1.0 - 2.0 * ((np.arange(N) + np.arange(N) // n + np.arange(N) // (n * n)) & 1))),
                        ("uniform random in [0, 1)", lambda: host.__setitem__(slice(None), np.random.default_rng(0).random(N))),
                        ("random with random exponents", lambda: host.__setitem__(slice(None), np.ldexp(np.random.default_rng(1).random(N) - 0.5, np.random.default_rng(2).integers(-40, 40, N))))):
        fill()
        x.CopyFrom(host, N)
        timed(label, x)
    L.MgcgFill(cg.vectorX.Ptr, 0.0)
    cg.Steps(50, restart=True)
    timed("CG iterate p after 50 iterations", cg.vectorP)
    L.MgcgFill(x.Ptr, 1.0)
    timed("ones again", x)


if __name__ == "__main__":
    main()
