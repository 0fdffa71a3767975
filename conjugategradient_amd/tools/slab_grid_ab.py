#!/usr/bin/env python3
"""Launch shapes at the size one rank of an 8-GPU run works on (a 64-plane slab of 512^3; every earlier sweep was at the full grid):
CG iteration time against the workgroup count of the row-tile SpMV (MgcgSetSpmvTuning) and of the two vector passes
(MGCG_R_GRID / MGCG_XP_GRID, read per launch), alternating inside one process."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib  # noqa: E402
from conjugategradient_amd.parallel import ConjugateGradientRankGpu  # noqa: E402


def main():
    planes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    L = _lib.lib()
    _lib.require_gpu()
    nx = 512
    N = nx * nx * planes
    cg = ConjugateGradientRankGpu(N, 7, 0, 10**9, 1e-8, rank=0, world=1, device=0)
    cg.InitializePoisson(nx, nx, planes)
    cg.Steps(10, restart=True)
    steps = max(50, int(400 * 64 / planes))

    def run():
        cg.Steps(5, restart=False)
        L.MgcgDeviceSynchronize()
        t0 = time.perf_counter()
        cg.Steps(steps, restart=False)
        L.MgcgDeviceSynchronize()
        return (time.perf_counter() - t0) / steps * 1e6

    def sweep(label, values, apply):
        res = {v: [] for v in values}
        for rep in range(3):
            for v in values:
                apply(v)
                res[v].append(run())
        apply(None)
        print(f"{label}: " + "  ".join(f"{'default' if v is None else v}: {sorted(t)[1]:.1f}" for v, t in res.items()) + "  us per iteration", flush=True)

    def spmv_grid(v):
        L.MgcgSetSpmvTuning(cg.cusparse, 64, 0, 0 if v is None else 4 * v)

    def env(name):
        def f(v):
            L.MgcgSetTuning(name.encode(), 0 if v is None else int(v))
        return f

    sweep(f"512x512x{planes} tile order (1 = memory order instead of the z sweep)", [None, 1], env("MGCG_NO_ZSWEEP"))
    sweep(f"512x512x{planes} SpMV workgroups", [None, 256, 384, 512, 768, 1024, 2048], spmv_grid)
    sweep(f"512x512x{planes} r-update workgroups", [None, 256, 512, 1024, 2048], env("MGCG_R_GRID"))
    sweep(f"512x512x{planes} x/p-update workgroups", [None, 256, 512, 1024, 2048, 4096], env("MGCG_XP_GRID"))


if __name__ == "__main__":
    main()
