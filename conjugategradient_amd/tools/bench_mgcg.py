#!/usr/bin/env python3
"""MGCG (BASELINE.json config 3: V-cycle, 3 levels, Jacobi, 7-pt 512^3, one GPU) next to plain CG on the same system:
iterations to the same absolute tolerance, wall time of Solve(), time per iteration.  Prints one JSON line."""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib  # noqa: E402
from conjugategradient_amd.multigrid import ConjugateGradientMgGpu  # noqa: E402
from conjugategradient_amd.parallel import ConjugateGradientRankGpu  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--levels", type=int, default=3)
    ap.add_argument("--nu", type=int, default=1)
    ap.add_argument("--nu-coarse", type=int, default=4)
    ap.add_argument("--rel-tol", type=float, default=1e-8, help="stop at ||r||_2 < rel_tol * ||b||_2 (b = 1)")
    ap.add_argument("--max-it", type=int, default=20000)
    ap.add_argument("--skip-cg", action="store_true")
    ap.add_argument("--interpolation", type=int, default=0, help="0: piecewise-constant transfer, 1: cell-centred linear (MgSetInterpolation)")
    ap.add_argument("--compression", action="store_true", help="MgcgSetMatrixCompression(1): lossless dictionary form of every level")
    a = ap.parse_args()
    L = _lib.lib()
    _lib.require_gpu()
    n = a.grid
    N = n**3
    tol = a.rel_tol * (N ** 0.5)
    out = {"grid": n, "rows": N, "abs_tol": tol, "levels": a.levels, "nu": a.nu, "nu_coarse": a.nu_coarse, "interpolation": a.interpolation}

    mg = ConjugateGradientMgGpu(N, 7, 0, a.max_it, tol, (n, n, n), levels=a.levels, nu=a.nu, nuCoarse=a.nu_coarse, rule=_lib.RULE_CSHARP, interpolation=a.interpolation)
    L.MgcgSetMatrixCompression(mg.cusparse, 1 if a.compression else 0)
    out["compression"] = bool(a.compression)
    t0 = time.perf_counter()
    mg.InitializePoisson()
    L.MgcgDeviceSynchronize()
    out["mg_setup_s"] = time.perf_counter() - t0
    out["mg_levels"] = mg.levels
    t0 = time.perf_counter()
    mg.Solve()
    dt = time.perf_counter() - t0
    out.update(mgcg_iterations=mg.Iteration + 1, mgcg_residual=mg.Residual, mgcg_solve_s=dt, mgcg_ms_per_iteration=1e3 * dt / (mg.Iteration + 1))
    mg.Dispose()

    if not a.skip_cg:
        cg = ConjugateGradientRankGpu(N, 7, 0, a.max_it, tol, rank=0, world=1, rule=_lib.RULE_CSHARP)
        L.MgcgSetMatrixCompression(cg.cusparse, 1 if a.compression else 0)
        cg.InitializePoisson(n, n, n)
        L.MgcgDeviceSynchronize()
        t0 = time.perf_counter()
        cg.Solve()
        dt = time.perf_counter() - t0
        out.update(cg_iterations=cg.Iteration + 1, cg_residual=cg.Residual, cg_solve_s=dt, cg_ms_per_iteration=1e3 * dt / (cg.Iteration + 1))
        out["speedup_time_to_solution"] = out["cg_solve_s"] / out["mgcg_solve_s"]
        cg.Dispose()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
