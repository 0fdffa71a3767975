#!/usr/bin/env python3
"""BASELINE.json config 5 at full size (random SPD, N = 10 M, ~30 nnz/row, irregular): SpMV of every kernel family against
the CPU oracle's product, timing, and the whole solve against a manufactured solution.  Prints one JSON line.
Host generation needs ~25 GB of RAM and a few minutes; --rows scales it down."""
# Lives under tests/ (not in the package) because it checks the GPU products against the CPU oracle: the oracle is test infrastructure.
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib, problems  # noqa: E402
from conjugategradient_amd.solver import ConjugateGradientSingleGpu, VectorDouble  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--no-solve", action="store_true")
    a = ap.parse_args()
    t0 = time.perf_counter()
    s = problems.random_spd(a.rows, mean_upper=14.0, seed=12345)
    out = {"rows": s.Count, "nnz": s.nnz, "nnz_per_row": s.nnz / s.Count, "max_row": int(np.diff(s.RowOffsets).max()), "generate_s": time.perf_counter() - t0}
    print(json.dumps(out), flush=True)
    from oracle import oracle as O

    x = np.cos(np.arange(s.Count) * 0.01)
    t0 = time.perf_counter()
    ref = O.spmv(s.Elements, s.ColumnIndeces, s.RowOffsets, x)
    out["oracle_spmv_s"] = time.perf_counter() - t0
    L = _lib.lib()
    _lib.require_gpu()
    cg = ConjugateGradientSingleGpu(s.Count, out["max_row"], 0, 1000, 1e-8, rule=_lib.RULE_CSHARP)
    cg.A = type("M", (), {})()
    cg.A.Elements, cg.A.ColumnIndeces, cg.A.RowOffsets = s.Elements, s.ColumnIndeces, s.RowOffsets
    cg.vectorA.Dispose(); cg.vectorColumnIndeces.Dispose()
    from conjugategradient_amd.solver import VectorInt
    cg.vectorA, cg.vectorColumnIndeces = VectorDouble(s.nnz), VectorInt(s.nnz)          # exact size instead of Count * maxNonZero
    # b = A.1 is the all-ones vector for this generator (row sums are 1), which CG solves in one step: use
    # b = A.(x + 2) = A.x + 2 instead, so the answer is x + 2 and the solve takes a few dozen iterations
    cg.x[:] = s.x
    cg.b[:] = ref + 2.0
    cg.Initialize()
    dx, dy = VectorDouble(s.Count), VectorDouble(s.Count)
    dx.CopyFrom(x, s.Count)
    algo = 12 * s.nnz + 4 * (s.Count + 1) + 16 * s.Count
    ev0, ev1 = L.MgcgEventCreate(), L.MgcgEventCreate()
    out["kernels"] = {}
    L.MgcgSetTuning(b"auto_tiles", 0)            # the CSR kernels themselves (per-op calls would switch to the library's column tiles after a few products)
    for name, k in (("auto", 0), ("row-block (stream form)", 1), ("rows (lane = row)", 9), ("8 lanes/row", 5), ("16 lanes/row", 6), ("32 lanes/row", 7)):
        L.MgcgSetSpmvKernel(cg.cusparse, k)
        args = (cg.cusparse, cg.matDescr, dy.ToRawPtr(), cg.vectorA.ToRawPtr(), cg.vectorRowOffsets.ToRawPtr(), cg.vectorColumnIndeces.ToRawPtr(), dx.ToRawPtr(), s.nnz, s.Count, s.Count, 1.0, 0.0)
        L.CsrMV(*args)
        _lib.check("CsrMV")
        got = dy.to_numpy()
        err = float(np.abs(got - ref).max() / np.abs(ref).max())
        L.MgcgEventRecord(ev0)
        for _ in range(a.reps):
            L.CsrMV(*args)
        L.MgcgEventRecord(ev1)
        ms = L.MgcgEventElapsedMs(ev0, ev1) / a.reps
        out["kernels"][name] = {"ms": ms, "algorithmic_gbps": algo / ms / 1e6, "bit_identical_to_oracle": bool(np.array_equal(got, ref)), "max_rel_err": err}
    L.MgcgSetSpmvKernel(cg.cusparse, 0)
    L.MgcgSetTuning(b"auto_tiles", 1)
    # per-op calls with everything at its default (compression off): CsrMV on library-owned vectors moves to the library's own column tiles after
    # 8 products of the same matrix (write registry instead of a checksum per call), Solve1 -- the product of the reference's phase driver -- with it
    L.MgcgAnalysisClear(cg.cusparse)
    for _ in range(12):
        L.CsrMV(*args)
    got = dy.to_numpy()
    L.MgcgEventRecord(ev0)
    for _ in range(a.reps):
        L.CsrMV(*args)
    L.MgcgEventRecord(ev1)
    ms = L.MgcgEventElapsedMs(ev0, ev1) / a.reps
    out["kernels"]["per-op CsrMV, defaults (library's own column tiles from the 8th product)"] = {
        "ms": ms, "algorithmic_gbps": algo / ms / 1e6, "bit_identical_to_oracle": bool(np.array_equal(got, ref)), "max_rel_err": float(np.abs(got - ref).max() / np.abs(ref).max())}
    L.Solve1(cg.cublas, cg.cusparse, cg.matDescr, cg.vectorA.Ptr, cg.vectorRowOffsets.Ptr, cg.vectorColumnIndeces.Ptr, dy.Ptr, dx.Ptr, s.Count, s.Count, 0, s.nnz)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        pap = L.Solve1(cg.cublas, cg.cusparse, cg.matDescr, cg.vectorA.Ptr, cg.vectorRowOffsets.Ptr, cg.vectorColumnIndeces.Ptr, dy.Ptr, dx.Ptr, s.Count, s.Count, 0, s.nnz)
    ms = (time.perf_counter() - t0) / a.reps * 1e3
    out["kernels"]["Solve1 (Ap = A p, p.Ap returned to the host), defaults"] = {
        "ms": ms, "bit_identical_to_oracle": bool(np.array_equal(dy.to_numpy(), ref)), "p_dot_Ap_rel_err": abs(pap - float(np.dot(x, ref))) / abs(float(np.dot(x, ref)))}
    L.MgcgAnalysisClear(cg.cusparse)
    # the column-tiled copy (class 4): opt-in for single products (MgcgSetMatrixCompression), the library's own choice inside solves;
    # 12-byte entries by default, the 16-byte form (round 2) for comparison
    def tiles_n():
        d = C.c_int(0)
        i = 0
        while True:
            c = L.MgcgAnalysisInfo(cg.cusparse, i, C.byref(d), None, None, None)
            if c == 4:
                return d.value
            if c < 0:
                return 0
            i += 1

    def tiled(label, pack):
        L.MgcgSetTuning(b"tile_pack", pack)
        L.MgcgAnalysisClear(cg.cusparse)
        L.MgcgSetMatrixCompression(cg.cusparse, 1)
        t0 = time.perf_counter()
        L.CsrMV(*args)
        L.MgcgDeviceSynchronize()
        out["analysis_s_" + label] = time.perf_counter() - t0
        out["analysis_class"] = L.MgcgAnalysisInfo(cg.cusparse, 0, None, None, None, None)
        got = dy.to_numpy()
        L.MgcgEventRecord(ev0)
        for _ in range(a.reps):
            L.CsrMV(*args)
        L.MgcgEventRecord(ev1)
        ms = L.MgcgEventElapsedMs(ev0, ev1) / a.reps
        out["kernels"]["column tiles, %s (%s)" % (label, ("2^%s columns per tile" % os.environ["MGCG_TILE_SHIFT"]) if os.environ.get("MGCG_TILE_SHIFT") else "%d tiles" % tiles_n())] = {
            "ms": ms, "algorithmic_gbps": algo / ms / 1e6, "bit_identical_to_oracle": bool(np.array_equal(got, ref)),
            "max_rel_err": float(np.abs(got - ref).max() / np.abs(ref).max())}
    tiled("16-byte entries", 0)
    tiled("12-byte entries", 1)
    print(json.dumps({k: v for k, v in out.items() if k != "kernels"} | {"tiled": out["kernels"][list(out["kernels"])[-1]]}), flush=True)
    if a.no_solve:
        print(json.dumps(out), flush=True)
        return
    # the whole solve with compression OFF: the library chooses the column tiles itself (Solve-family calls, matrices without locality)
    L.MgcgSetMatrixCompression(cg.cusparse, 0)
    L.MgcgAnalysisClear(cg.cusparse)
    for label, auto in (("csr_kernels", 0), ("automatic_column_tiles", 1)):
        L.MgcgSetTuning(b"auto_tiles", auto)
        cg.x[:] = s.x
        cg.vectorX.CopyFrom(cg.x, s.Count)
        t0 = time.perf_counter()
        cg.Solve()
        dt = time.perf_counter() - t0
        cg.Read()
        out["solve_" + label] = {"solve_s": dt, "iterations": cg.Iteration + 1, "residual": cg.Residual, "ms_per_iteration": 1e3 * dt / (cg.Iteration + 1),
                                 "max_abs_error_of_x": float(np.abs(cg.x - (x + 2.0)).max()),
                                 "analysis_class": L.MgcgAnalysisInfo(cg.cusparse, 0, None, None, None, None)}
    # again, the form is cached now (checksum-verified at entry): the steady-state time of a solve
    cg.x[:] = s.x
    cg.vectorX.CopyFrom(cg.x, s.Count)
    t0 = time.perf_counter()
    cg.Solve()
    dt = time.perf_counter() - t0
    out["solve_automatic_column_tiles_cached"] = {"solve_s": dt, "iterations": cg.Iteration + 1, "ms_per_iteration": 1e3 * dt / (cg.Iteration + 1)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
