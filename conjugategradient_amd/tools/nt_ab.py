#!/usr/bin/env python3
"""Matrix streams of the row-tile SpMV with and without the non-temporal hint (MGCG_ROWTILE_NT=1 / 0), alternating inside ONE
process (same placement of every array): CG iteration time at slab sizes of the 512^3 grid and at the full grid."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib  # noqa: E402
from conjugategradient_amd.parallel import ConjugateGradientRankGpu  # noqa: E402


def main():
    L = _lib.lib()
    _lib.require_gpu()
    for dims, steps in (((128, 128, 128), 1000), ((512, 512, 8), 1000), ((512, 512, 16), 800), ((512, 512, 64), 400), ((512, 512, 128), 200), ((512, 512, 160), 150), ((512, 512, 256), 100), ((512, 512, 512), 100)):
        nx, ny, nz = dims
        N = nx * ny * nz
        cg = ConjugateGradientRankGpu(N, 7, 0, 10**9, 1e-8, rank=0, world=1, device=0)
        cg.InitializePoisson(nx, ny, nz)
        cg.Steps(10, restart=True)
        out = {0: [], 1: []}
        for rep in range(4):
            for nofold in (0, 1):
                L.MgcgSetTuning(b"rowtile_nt", 1 if nofold else 0)
                cg.Steps(5, restart=False)
                L.MgcgDeviceSynchronize()
                t0 = time.perf_counter()
                cg.Steps(steps, restart=False)
                L.MgcgDeviceSynchronize()
                out[nofold].append((time.perf_counter() - t0) / steps * 1e6)
        L.MgcgSetTuning(b"rowtile_nt", -1)
        f, s = sorted(out[0])[1], sorted(out[1])[1]
        print(f"{nx}x{ny}x{nz}: plain matrix loads {f:9.2f} us per iteration | non-temporal matrix loads {s:9.2f} us | {100 * (s - f) / s:+.1f} %", flush=True)
        cg.Dispose()


if __name__ == "__main__":
    main()
