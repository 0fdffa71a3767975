#!/usr/bin/env python3
"""Folded finalisation (the x/p update takes the stop decision itself: one launch fewer per iteration) against the separate
finalize kernel, alternating inside ONE process (same placement of every array): iterations per second at several sizes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib  # noqa: E402
from conjugategradient_amd.parallel import ConjugateGradientRankGpu  # noqa: E402


def main():
    L = _lib.lib()
    _lib.require_gpu()
    for dims, steps in (((256, 256, 1), 2000), ((64, 64, 64), 2000), ((128, 128, 128), 1000), ((256, 256, 256), 300), ((512, 512, 512), 100)):
        nx, ny, nz = dims
        N = nx * ny * nz
        cg = ConjugateGradientRankGpu(N, 7, 0, 10**9, 1e-8, rank=0, world=1, device=0)
        cg.InitializePoisson(nx, ny, nz)
        cg.Steps(10, restart=True)
        out = {0: [], 1: []}
        for rep in range(4):
            for nofold in (0, 1):
                L.MgcgSetTuning(b"no_folded_finalize", nofold)
                cg.Steps(5, restart=False)
                L.MgcgDeviceSynchronize()
                t0 = time.perf_counter()
                cg.Steps(steps, restart=False)
                L.MgcgDeviceSynchronize()
                out[nofold].append((time.perf_counter() - t0) / steps * 1e6)
        L.MgcgSetTuning(b"no_folded_finalize", 0)
        f, s = sorted(out[0])[1], sorted(out[1])[1]
        print(f"{nx}x{ny}x{nz}: folded {f:9.2f} us per iteration | separate finalize kernel {s:9.2f} us | {100 * (s - f) / s:+.1f} %", flush=True)
        cg.Dispose()


if __name__ == "__main__":
    main()
