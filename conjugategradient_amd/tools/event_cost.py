#!/usr/bin/env python3
"""Does recording a HIP event pair around every SpMV of the CG loop (MgcgProfileSpmv, what bench.py's roofline needs) slow the
loop down?  The same 100 iterations at 512^3 with the per-launch events off and on, alternating."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib  # noqa: E402
from conjugategradient_amd.parallel import ConjugateGradientRankGpu  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    L = _lib.lib()
    _lib.require_gpu()
    cg = ConjugateGradientRankGpu(n**3, 7, 0, 10**9, 1e-8, rank=0, world=1, device=0)
    cg.InitializePoisson(n, n, n)
    cg.Steps(10, restart=True)
    for rep in range(3):
        for on in (0, 1):
            L.MgcgProfileSpmv(cg.cusparse, on)
            L.MgcgDeviceSynchronize()
            t0 = time.perf_counter()
            cg.Steps(100, restart=False)
            L.MgcgDeviceSynchronize()
            dt = time.perf_counter() - t0
            ln = C.c_int(0)
            ms = L.MgcgProfileSpmvMs(cg.cusparse, C.byref(ln)) / max(ln.value, 1) if on else 0.0
            L.MgcgProfileSpmv(cg.cusparse, 0)
            print(f"events {'on ' if on else 'off'}: {dt / 100 * 1e3:.4f} ms per iteration" + (f", SpMV {ms:.4f} ms over {ln.value} launches" if on else ""), flush=True)


if __name__ == "__main__":
    main()
