#!/usr/bin/env python3
"""Inside the CG loop: eight candidate allocations for the search direction p (the vector the SpMV gathers from), the loop's
iterations per second and the SpMV's average time with each.  Does the spread seen for the bare CsrMV (placement_probe.py) carry over,
and what would 'keep the best of k' buy?"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib  # noqa: E402
from conjugategradient_amd.parallel import ConjugateGradientRankGpu  # noqa: E402
from conjugategradient_amd.solver import VectorDouble  # noqa: E402


def main():
    n = 512
    N = n**3
    L = _lib.lib()
    _lib.require_gpu()
    cg = ConjugateGradientRankGpu(N, 7, 0, 10**9, 1e-8, rank=0, world=1, device=0)
    cg.InitializePoisson(n, n, n)
    keep = [cg.vectorP]
    for k in range(8):
        if k > 0:
            cg.vectorP = VectorDouble(N)
            keep.append(cg.vectorP)
        cg.Steps(5, restart=True)
        L.MgcgProfileSpmv(cg.cusparse, 1)
        L.MgcgDeviceSynchronize()
        t0 = time.perf_counter()
        cg.Steps(60, restart=False)
        L.MgcgDeviceSynchronize()
        dt = time.perf_counter() - t0
        ln = C.c_int(0)
        ms = L.MgcgProfileSpmvMs(cg.cusparse, C.byref(ln)) / max(ln.value, 1)
        L.MgcgProfileSpmv(cg.cusparse, 0)
        print(f"p candidate {k} at {cg.vectorP.ToRawPtr():#x}: {60 / dt:6.1f} it/s  SpMV {ms:.3f} ms = {13939769348 / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
    # and the first one again
    cg.vectorP = keep[0]
    cg.Steps(5, restart=True)
    L.MgcgDeviceSynchronize()
    t0 = time.perf_counter()
    cg.Steps(60, restart=False)
    L.MgcgDeviceSynchronize()
    print(f"p candidate 0 again: {60 / (time.perf_counter() - t0):6.1f} it/s", flush=True)


if __name__ == "__main__":
    main()
