#!/usr/bin/env python3
"""The reference's own hand-over -- host CSR arrays, x and b uploaded by Initialize() (ConjugateGradientSingleGpu.cs:112-131), the whole
loop in one Solve(), x read back -- timed with the PCIe legs included, next to the loop alone.  bench.py's `value` is the loop with
its inputs resident in HBM (the driver contract); this is the other figure DESIGN.md section 7 quotes.  One JSON line per grid.
The host arrays come from the device generator (downloaded once, untimed): building a 512^3 CSR in numpy takes minutes."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib  # noqa: E402
from conjugategradient_amd.parallel import ConjugateGradientRankGpu  # noqa: E402
from conjugategradient_amd.solver import ConjugateGradientSingleGpu, VectorDouble, VectorInt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grids", default="256,512")
    a = ap.parse_args()
    L = _lib.lib()
    _lib.require_gpu()
    for n in [int(v) for v in a.grids.split(",")]:
        N = n ** 3
        gen = ConjugateGradientRankGpu(N, 7, 0, 10, 1e-8, rank=0, world=1, device=0)
        gen.InitializePoisson(n, n, n)
        nnz = gen.part.elementCount
        e, c, ro = np.empty(nnz), np.empty(nnz, dtype=np.int32), np.empty(N + 1, dtype=np.int32)
        gen.vectorElements.CopyTo(e, nnz)
        gen.vectorColumnIndeces.CopyTo(c, nnz)
        gen.vectorRowOffsets.CopyTo(ro, N + 1)
        gen.Dispose()
        tol = 1e-8 * np.sqrt(N)
        cg = ConjugateGradientSingleGpu(N, 7, 0, 100000, tol, rule=_lib.RULE_CSHARP)
        cg.vectorA.Dispose(); cg.vectorColumnIndeces.Dispose()
        cg.vectorA, cg.vectorColumnIndeces = VectorDouble(nnz), VectorInt(nnz)          # (exact size instead of count * maxNonZeroCount)
        cg.A = type("M", (), {})()
        cg.A.Elements, cg.A.ColumnIndeces, cg.A.RowOffsets = e, c, ro
        cg.b[:] = 1.0
        cg.x[:] = 0.0
        rec = {"grid": n, "rows": N, "nnz": int(nnz), "uploaded_bytes": int(12 * nnz + 4 * (N + 1) + 16 * N)}
        for rep in ("first", "second"):                  # (the first solve of a process also pays for the kernels' code objects and the row-shape sample)
            cg.x[:] = 0.0
            L.MgcgDeviceSynchronize()
            t0 = time.perf_counter()
            cg.Initialize()
            L.MgcgDeviceSynchronize()
            t1 = time.perf_counter()
            cg.Solve()
            t2 = time.perf_counter()
            cg.Read()
            t3 = time.perf_counter()
            its = cg.Iteration + 1
            rec[rep] = {"upload_s": t1 - t0, "upload_gbps": rec["uploaded_bytes"] / (t1 - t0) / 1e9, "solve_s": t2 - t1, "read_back_s": t3 - t2,
                        "iterations": its, "residual": cg.Residual, "iterations_per_s_loop_only": its / (t2 - t1), "iterations_per_s_with_pcie": its / (t3 - t0)}
        print(json.dumps(rec), flush=True)
        cg.Dispose()


if __name__ == "__main__":
    main()
