#!/usr/bin/env python3
"""The prolongation folded into the post-smoothing sweep of the V(1,1) cycle (x1 + P e formed per gather, MGCG_FOLD_UP=1) against the
prolongation kernel + stored iterate, alternating inside ONE process (same placement of every array): milliseconds per MGCG iteration
(whole solves of the 7-point Poisson problem, b = 1) and per application of the preconditioner alone."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib  # noqa: E402
from conjugategradient_amd.multigrid import ConjugateGradientMgGpu  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grids", default="256,512x256x256,512x512x256,512", help="n or nx x ny x nz, comma separated")
    ap.add_argument("--levels", type=int, default=3)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    L = _lib.lib()
    _lib.require_gpu()
    for g in a.grids.split(","):
        dims = tuple(int(v) for v in g.split("x")) if "x" in g else (int(g),) * 3
        n = "x".join(str(v) for v in dims)
        N = dims[0] * dims[1] * dims[2]
        tol = 1e-8 * np.sqrt(N)
        mg = ConjugateGradientMgGpu(N, 7, 0, 1000, tol, dims, levels=a.levels, rule=_lib.RULE_CSHARP)
        mg.InitializePoisson()
        L.MgcgSetMatrixCompression(mg.cusparse, 0)
        modes = (1, 0, -1)
        out = {m: [] for m in modes}
        its = {}
        for rep in range(a.reps + 1):
            for m in modes:
                L.MgcgSetTuning(b"fold_up", m)
                L.MgcgFill(mg.vectorX.Ptr, 0.0)
                L.MgcgDeviceSynchronize()
                t0 = time.perf_counter()
                mg.Solve()
                dt = time.perf_counter() - t0
                its[m] = (mg.Iteration + 1, mg.Residual, L.MgcgLastVcycleFolds())
                if rep > 0:
                    out[m].append(1e3 * dt / (mg.Iteration + 1))
        L.MgcgSetTuning(b"fold_up", -1)
        med = {m: sorted(out[m])[len(out[m]) // 2] for m in modes}
        print(f"{n}, {a.levels} levels ({N / 1e6:.1f} M rows): folded on every level {med[1]:8.3f} ms per iteration | prolongation kernels {med[0]:8.3f} ms | "
              f"by level size {med[-1]:8.3f} ms (folds {its[-1][2]})   iterations {its[1][0]} / {its[0][0]} / {its[-1][0]}, residual {its[1][1]:.6e} / {its[0][1]:.6e}", flush=True)
        mg.Dispose()


if __name__ == "__main__":
    main()
