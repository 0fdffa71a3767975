#!/usr/bin/env python3
"""One rank's slab on the several-ranks code path (MGCG_FORCE_MULTIRANK, a real one-rank RCCL communicator) for a kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 conjugategradient_amd/tools/forced_path_run.py --overlap 2 --halo-stream 1
then tools/trace_iteration.py OUT prints the kernels of one steady-state iteration with the gaps between them.  Without the profiler it prints one
JSON line with the milliseconds per iteration (tools/slab_latency.py runs every schedule in a fresh process through this)."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=512)
    ap.add_argument("--planes", type=int, default=64)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--overlap", type=int, default=2)
    ap.add_argument("--halo-stream", type=int, default=0)
    ap.add_argument("--force", type=int, default=-1, help="entries of the artificial halo (default one plane; 0: the plain single-rank loop)")
    ap.add_argument("--solver", choices=["cg", "mgcg"], default="cg")
    ap.add_argument("--repeats", type=int, default=1)
    ap.add_argument("--fold-up", type=int, default=-1, help="MGCG_FOLD_UP: -1 by level size, 0 never, 1 everywhere")
    ap.add_argument("--ab-fold-up", action="store_true", help="mgcg: alternate fold_up = --fold-up and 0 inside this process (same placement), report both")
    a = ap.parse_args()
    from conjugategradient_amd import _lib
    from conjugategradient_amd.parallel import ConjugateGradientMgRankGpu, ConjugateGradientRankGpu

    L = _lib.lib()
    _lib.require_gpu()
    L.SetDevice(0)
    comm = None
    force = a.nx * a.nx if a.force < 0 else a.force
    if force > 0:
        buf = (C.c_char * 128)()
        assert L.MgcgCommGetUniqueId(buf) == 0, _lib.last_error()
        comm = L.MgcgCommInitRank(buf, 1, 0)
        _lib.check("MgcgCommInitRank")
    L.MgcgSetTuning(b"force_multirank", force)
    L.MgcgSetTuning(b"overlap", a.overlap)
    L.MgcgSetTuning(b"halo_stream", a.halo_stream)
    L.MgcgSetTuning(b"fold_up", a.fold_up)
    dims = (a.nx, a.nx, a.planes)
    n = dims[0] * dims[1] * dims[2]
    import json
    import time

    def timed(fn):
        L.MgcgDeviceSynchronize()
        t0 = time.perf_counter()
        fn()
        L.MgcgDeviceSynchronize()
        return (time.perf_counter() - t0) * 1e3 / a.steps

    if a.solver == "mgcg":
        cg = ConjugateGradientMgRankGpu(n, 7, 0, 10**9, 1e300, dims, rank=0, world=1, comm=comm, rule=_lib.RULE_NATIVE, levels=3, nu=1, nuCoarse=4)
        cg.InitializePoisson(*dims)
        cg.Setup()
        cg.MinIteration = a.steps - 1

        def run():
            L.MgcgFill(cg.vectorX.Ptr, 0.0)
            cg.Solve()
        run()
    else:
        cg = ConjugateGradientRankGpu(n, 7, 0, 10**9, 1e-8, rank=0, world=1, comm=comm)
        cg.InitializePoisson(*dims)
        cg.Steps(10, restart=True)

        def run():
            cg.Steps(a.steps, restart=False)
    ab = None
    if a.ab_fold_up and a.solver == "mgcg":
        ab = {"fold_up": [], "prolongation_kernel": []}
        folds = {}
        for _ in range(max(a.repeats, 3)):
            for key, v in (("fold_up", a.fold_up), ("prolongation_kernel", 0)):
                L.MgcgSetTuning(b"fold_up", v)
                run()
                ab[key].append(timed(run))
                folds[key] = L.MgcgLastVcycleFolds()
        L.MgcgSetTuning(b"fold_up", a.fold_up)
        ab = {k: sorted(v)[len(v) // 2] for k, v in ab.items()}
        ab["folds_reported"] = folds
    ms = [timed(run) for _ in range(a.repeats)]
    active = cg.LastOverlap()[0] if a.solver == "cg" else None
    us = (C.c_double * 2)(0.0, 0.0)
    measured = bool(L.MgcgLastOverlapTimes(us))       # overlap = 1: the library's measured rule (last plan of this thread: the CG loop's own halo)
    print(json.dumps({"overlap_decided_by_measurement": measured, "measured_exchange_in_line_us": us[0] if measured else None, "measured_fork_launch_join_us": us[1] if measured else None,
                      "solver": a.solver, "grid": list(dims), "force_multirank": force, "overlap": a.overlap, "halo_stream": a.halo_stream,
                      "steps": a.steps, "ms_per_iteration": min(ms), "ms_per_iteration_all": ms, "halo_overlap_active": active, "fold_up_ab_ms_per_iteration": ab}))
    cg.Dispose()


if __name__ == "__main__":
    main()
