#!/usr/bin/env python3
"""BASELINE.json config 5 row-partitioned over R loopback ranks on ONE GPU (each rank a host thread on a virtual device): which
SpMV form every rank's slab takes, iterations, and the wall time per iteration of the rehearsal (host-staged halo, ranks
serialised on one card -- NOT an 8-GPU figure).  Prints one JSON line.  --rows scales the matrix down."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib, problems  # noqa: E402
from conjugategradient_amd.parallel import ConjugateGradientRankGpu  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--no-solve", action="store_true")
    ap.add_argument("--balance", choices=["rows", "nnz"], default="rows", help="rows: the reference's partition; nnz: equal nonzero counts")
    a = ap.parse_args()
    os.environ["MGCG_VIRTUAL_DEVICES"] = str(a.ranks)
    from tests.test_gpu_parallel import _run_ranks_in_threads

    s = problems.random_spd(a.rows, mean_upper=14.0, seed=12345)
    N = s.Count
    xs = np.cos(np.arange(N) * 0.01)
    s.b[:] = 2.0 + np.sin(np.arange(N) * 0.003)
    s.x[:] = 0.0
    maxnz = int(np.diff(s.RowOffsets).max())
    L = _lib.lib()
    _lib.require_gpu()

    def make_rank(rank, comm):
        rk = ConjugateGradientRankGpu(N, maxnz, 0, 1000, 1e-8, rank=rank, world=a.ranks, comm=comm, device=rank, balance=a.balance).load(s)
        rk.Initialize()
        L.MgcgSetMatrixCompression(rk.cusparse, 0)
        t0 = time.perf_counter()
        rk.Solve()
        dt = time.perf_counter() - t0
        form = L.MgcgAnalysisInfo(rk.cusparse, 0, None, None, None, None)
        out = {"rank": rank, "rows": rk.part.count, "nnz": rk.part.elementCount, "form": form, "iterations": rk.Iteration + 1, "residual": rk.Residual, "solve_s": dt}
        rk.Dispose()
        return out

    out = {"rows": N, "ranks": a.ranks, "balance": a.balance}
    if not a.no_solve:
        out["per_rank"] = _run_ranks_in_threads(a.ranks, make_rank)
    # every rank's slab by itself (rows x N, x of full length resident): one product on the CSR kernels and on the column tiles
    ev0, ev1 = L.MgcgEventCreate(), L.MgcgEventCreate()
    slabs = []
    for rank in range(a.ranks):
        rk = ConjugateGradientRankGpu(N, maxnz, 0, 1000, 1e-8, rank=rank, world=a.ranks, comm=None, device=0, balance=a.balance).load(s)
        rk.Initialize()
        p = rk.part
        rk.vectorP.CopyFrom(xs, N)
        args = (rk.cusparse, rk.matDescr, rk.vectorAp.ToRawPtr(), rk.vectorElements.ToRawPtr(), rk.vectorRowOffsets.ToRawPtr(), rk.vectorColumnIndeces.ToRawPtr(),
                rk.vectorP.ToRawPtr(), p.elementCount, p.count, N, 1.0, 0.0)
        rec = {"rank": rank, "rows": p.count, "nnz": p.elementCount, "nnz_per_row": p.elementCount / p.count}
        ref = None
        for label, mode in (("csr_ms", 0), ("tiles_ms", 1)):
            L.MgcgAnalysisClear(rk.cusparse)
            L.MgcgSetMatrixCompression(rk.cusparse, mode)
            L.CsrMV(*args)
            _lib.check("CsrMV")
            got = rk.vectorAp.to_numpy()[: p.count]
            if ref is None:
                ref = got
            else:
                rec["tiles_max_rel_diff"] = float(np.abs(got - ref).max() / np.abs(ref).max())
                rec["tiles_class"] = L.MgcgAnalysisInfo(rk.cusparse, 0, None, None, None, None)
            L.MgcgEventRecord(ev0)
            for _ in range(a.reps):
                L.CsrMV(*args)
            L.MgcgEventRecord(ev1)
            rec[label] = L.MgcgEventElapsedMs(ev0, ev1) / a.reps
        slabs.append(rec)
        rk.Dispose()
    out["slab_products"] = slabs
    print(json.dumps(out))


if __name__ == "__main__":
    main()
