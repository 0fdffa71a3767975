#!/usr/bin/env python3
"""bench.py -- CG iterations/sec and SpMV achieved HBM GB/s on the 7-point Poisson 512^3 matrix.

    python bench.py --gpus N --steps K --warmup W

N > 1: either under a launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...), or as the bare command above -- the process then starts its N ranks itself
(launch_ranks: a GPU-free parent, one child per GPU, rank 0's line forwarded, the worst exit code returned).

A "step" is one CG iteration of the hot path (halo exchange, SpMV fused with p.Ap, x/r update fused
with r.r, stop-test scalars, p update) on the synthetic 7-point 512^3 system (config of BASELINE.json's
metric; it fits one GPU: 11.8 GB CSR + 5.4 GB vectors).  With N GPUs the SAME matrix is row-partitioned
into N z-slabs (strong scaling, the north star's ">= 6x at 8 GPUs"), one process per GPU, RCCL inside
libMgcgGpu.so for the two scalar all-reduces and the halo exchange; torch.distributed only bootstraps
(unique id, barriers, max over ranks).  The matrix is generated directly in HBM by the library's device
generator; no reference data set exists or is needed ("data": "synthetic").

The timed loop multiplies by the PLAIN CSR arrays (12 B per nonzero: `value`, `ms_per_step` and `roofline` are all
plain-CSR figures).  The library's opt-in lossless compact forms are timed separately and reported under
`lossless_forms`; they never enter `value` or `roofline`.

Output: ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline       -- dominant kernel = CSR SpMV fused with p.Ap (spmv_rowtile_kernel<EPI_DOT>): algorithmic bytes per
                    launch (12*nnz + 4*(N+1) + 16*N) divided by its average duration measured
                    with HIP events on the library's stream inside the timed region; peak = 8.0 TB/s HBM3E;
                    traffic = PMC bytes per launch (profiles/spmv_traffic.json, separate rocprofv3 --pmc passes).
  cpu_baseline   -- the CPU oracle (single thread, the reference CPU path's behaviour) timed on a bounded
                    sample of the same workload (rank 0, N=1 only).
  mgcg           -- BASELINE config 3 (3-level V(1,1) Jacobi MGCG on the same matrix, plain CSR on every level):
                    iterations to 1e-8*||b||, ms per iteration, V-cycle algorithmic bytes and achieved fraction.
  lossless_forms -- the same CG loop on the row-pattern / per-nonzero-code forms (secondary, N=1 only).
  parity_vs_single_rank -- N > 1 only, untimed: rank 0 replays the same iterations as ONE rank on its own GPU; both residuals and
                    their relative difference (north star: 1e-10) travel in the line, so a scaling record checks itself; the
                    replay is timed too (single_rank_ms_per_step, speedup_vs_single_rank: the reference's own benchmark
                    times its 1-GPU and N-GPU legs in one run, Mgcg/cuBlas/Mgcg/MgcgMain.cs:143-167).
  N > 1, --solver cg (the driver's command), after the timed steps and outside `value`:
  mgcg           -- BASELINE config 4: the row-partitioned 3-level V(1,1) MGCG on the same communicator -- ms per iteration over
                    a fixed count, iterations and seconds to 1e-8*||b||, the same solve by ONE rank on rank 0's GPU
                    (speedup_vs_single_rank: the north star's ">= 6x solver-time speed-up at 8 GPUs") and the residuals of
                    both after the same fixed count (parity_vs_single_rank).
  comm_probe     -- MgcgCommProbe prices on the live communicator (max over ranks, microseconds): 8-byte and 16-byte
                    all-reduce, one grid plane to and from both z-neighbours, fork/join of the overlap schedule, a kernel boundary.
  schedules      -- ms per CG / MGCG iteration with the halo exchange in line, hidden behind the interior rows (interior rows on
                    the side stream), and on the side stream itself (halo_stream = 1), next to what the library's measured rule chose.
  A step of these extras that does not finish within MGCG_BENCH_EXTRAS_TIMEOUT seconds (default 150) ends the run with the
  line as far as it got ("extras_aborted" says where) and exit status 0: the timed result is never lost to an extra.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--grid", type=int, default=512, help="n for the n^3 7-point Poisson grid (BASELINE config: 512)")
    ap.add_argument("--solver", choices=["cg", "mgcg"], default="cg")
    ap.add_argument("--mg-levels", type=int, default=3, help="--solver mgcg: hierarchy depth (BASELINE config 3/4: 3)")
    ap.add_argument("--mg-nu", type=int, default=1, help="--solver mgcg: pre/post Jacobi sweeps")
    ap.add_argument("--mg-nu-coarse", type=int, default=4, help="--solver mgcg: sweeps on the coarsest level")
    ap.add_argument("--mg-interpolation", type=int, default=0, choices=[0, 1], help="--solver mgcg: 0 piecewise-constant transfer (default), 1 cell-centred linear")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=4, help="CG iterations of the CPU oracle sample")
    ap.add_argument("--no-compression", action="store_true", help="(default) plain CSR inside the loop")
    ap.add_argument("--compression", type=int, default=None, choices=[0, 1, 2],
                    help="MgcgSetMatrixCompression mode for the timed loop: 0 plain CSR (default, the BASELINE metric), "
                         "1 best lossless form, 2 per-nonzero codes only -- with 1 or 2 the line is NOT the BASELINE metric and says so")
    ap.add_argument("--no-extras", action="store_true", help="skip the mgcg and lossless_forms extra objects")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="N > 1: if RCCL does not form, run over the host-staged gloo transport instead of failing (never a scaling result)")
    ap.add_argument("--torch-first", action="store_true", help="import torch before the library even at N=1 (runtime-compat check)")
    ap.add_argument("--spmv-kernel", type=int, default=None)
    ap.add_argument("--spmv-rows", type=int, default=None)
    ap.add_argument("--spmv-flags", type=int, default=None)
    ap.add_argument("--spmv-grid", type=int, default=None)
    return ap.parse_args()


def _cpu_share() -> int:
    """Host threads this job may use: the affinity mask, the cgroup quota, and the GPU box's per-GPU share (16)."""
    share = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            share = min(share, max(1, int(int(quota) / int(period))))
    except Exception:       # noqa: BLE001 -- no cgroup v2 file: keep the affinity count
        pass
    return max(1, min(share, int(os.environ.get("MGCG_CPU_THREADS", "16"))))


def cpu_baseline(n: int, iters: int):
    """The CPU oracle (oracle/cg_oracle.c, single thread like the reference's serial C# loops) on the same
    7-point n^3 system; falls back to a smaller grid if host memory is short."""
    import numpy as np

    from oracle import oracle as O

    L = O.lib()
    avail = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    grid = n
    while grid > 64 and (12 * 7 + 8 * 6) * grid**3 * 1.15 > avail:
        grid //= 2
    N = grid**3
    nnz = L.oracle_poisson_nnz(grid, grid, grid)
    e = np.empty(nnz)
    c = np.empty(nnz, dtype=np.int32)
    r = np.empty(N + 1, dtype=np.int32)
    L.oracle_poisson_fill(grid, grid, grid, e, c, r)
    x, b, work = np.zeros(N), np.ones(N), np.empty(3 * N)
    res = C.c_double(0)
    # init (r = b - A x, p = r, rr) is outside the per-iteration figure, as on the GPU side
    L.oracle_cg_steps(e, c, r, N, x, b, 0, C.byref(res), work)      # first touch of the work arrays
    t0 = time.perf_counter()
    L.oracle_cg_steps(e, c, r, N, x, b, 0, C.byref(res), work)
    t_init = time.perf_counter() - t0
    x[:] = 0
    t0 = time.perf_counter()
    L.oracle_cg_steps(e, c, r, N, x, b, iters, C.byref(res), work)
    dt = max(time.perf_counter() - t0 - t_init, 1e-9)
    its = iters / dt
    scale = (grid / n) ** 3     # report in units of the full-size workload
    allcores = None
    try:                        # second figure: the same loop on all host cores (OpenMP) -- NOT the reference's behaviour
        omp_path = os.path.join(ROOT, "oracle", "liboracle_omp.so")
        if not os.path.exists(omp_path):
            import subprocess
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_omp.so"])
        M = C.CDLL(omp_path)
        dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
        ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
        M.oracle_cg_steps_omp.argtypes = [dp, ip, ip, C.c_int64, dp, dp, C.c_int, C.POINTER(C.c_double), dp]
        M.oracle_omp_threads.restype = C.c_int
        M.oracle_omp_set_threads(_cpu_share())
        res2 = C.c_double(0)
        x[:] = 0
        M.oracle_cg_steps_omp(e, c, r, N, x, b, 0, C.byref(res2), work)
        t0 = time.perf_counter()
        M.oracle_cg_steps_omp(e, c, r, N, x, b, 0, C.byref(res2), work)
        t_init2 = time.perf_counter() - t0
        x[:] = 0
        omp_iters = 4 * iters
        t0 = time.perf_counter()
        M.oracle_cg_steps_omp(e, c, r, N, x, b, omp_iters, C.byref(res2), work)
        dt2 = max(time.perf_counter() - t0 - t_init2, 1e-9)
        allcores = {"value": omp_iters / max(dt2, 1e-9) * scale, "unit": "iterations/s", "cores": int(M.oracle_omp_threads()),
                    "kind": "port, OpenMP all host cores -- not the reference's behaviour (its CPU loops are serial)",
                    "sample": f"{omp_iters} iterations on 7-pt Poisson {grid}^3", "seconds_per_iteration": dt2 / omp_iters}
    except Exception as ex:     # noqa: BLE001 -- the second figure is optional
        allcores = {"error": str(ex)}
    # the same iterations once more with the oracle's dot products summed with compensation (oracle/cg_oracle.c: oracle_set_dot_mode; untimed):
    # how far the reference's serial left-to-right sums are from the exact sums of the same products at this size -- the yardstick for
    # the HIP loop's distance from the reference-order figure above
    res_comp = None
    try:
        L.oracle_set_dot_mode(1)
        x[:] = 0
        rc = C.c_double(0)
        L.oracle_cg_steps(e, c, r, N, x, b, iters, C.byref(rc), work)
        res_comp = rc.value
    except Exception:       # noqa: BLE001
        res_comp = None
    finally:
        L.oracle_set_dot_mode(0)
    return {
        "residual_with_compensated_dots": res_comp,
        "all_cores_variant": allcores,
        "value": its * scale,
        "unit": "iterations/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{iters} CG iterations of the CPU oracle (C restatement of ConjugateGradientCpu.cs, 1 thread) on 7-pt Poisson {grid}^3"
                  + ("" if grid == n else f", rate scaled by ({grid}/{n})^3 to the {n}^3 workload"),
        "seconds_per_iteration": dt / iters,
        "residual": res.value,
        "grid": grid,
        "iterations": iters,
    }


def vcycle_bytes(n: int, levels: int, nu: int, nu_coarse: int):
    """Algorithmic bytes of one MGCG iteration on the 7-point n^3 hierarchy: per level
    sweep = 12 nnz + 36 N (SpMV-shaped pass + b, D^-1, x write; the first sweep from a zero guess reads b, D^-1 and writes x:
    24 N), residual = 12 nnz + 28 N, restriction / prolongation = 8 N_l + 8 N_{l+1} each; the CG shell adds one SpMV
    (12 nnz + 4 (N + 1) + 16 N) and 72 N of vector traffic."""
    def nnz(m):
        return 7 * m**3 - 6 * m * m
    total = 0
    m = n
    for lv in range(levels):
        N = m**3
        sweep, first = 12 * nnz(m) + 36 * N, 24 * N
        if lv == levels - 1:
            total += first + (nu_coarse - 1) * sweep
        else:
            Nc = (m // 2) ** 3
            total += first + (nu - 1) * sweep              # pre-smoothing from a zero guess
            total += 12 * nnz(m) + 28 * N                   # residual
            total += 2 * (8 * N + 8 * Nc)                   # restriction + prolongation
            total += nu * sweep                             # post-smoothing
        m //= 2
    N0 = n**3
    shell = 12 * nnz(n) + 4 * (N0 + 1) + 16 * N0 + 72 * N0
    return total, shell


FOLD_UP_MAX_ROWS = 100_000_000      # csrc/solver.hip kFoldUpMaxRows


def vcycle_required_bytes(n: int, levels: int, nu: int, nu_coarse: int, fold: bool = True):
    """Bytes that MUST move per MGCG iteration on the 7-point n^3 hierarchy as the library runs it (constant-coefficient operator:
    every level's diagonal is uniform, so D^-1 is a scalar and its array is not read): sweep = 12 nnz + 4 (N + 1) + 24 N (x gathered once,
    b, x written), first sweep from zero = 16 N, residual = 12 nnz + 4 (N + 1) + 24 N, restriction = 8 N + 8 N_c, prolongation = 16 N + 8 N_c
    (it reads and rewrites the fine iterate).  fold (V(1,*) on one rank): the first sweep is not stored -- 0 bytes -- and the residual pass
    gathers b itself: 12 nnz + 4 (N + 1) + 16 N.  V(1,1), power-of-two n, levels of up to FOLD_UP_MAX_ROWS rows: the prolongation is folded
    into the post-smoothing sweep as well -- no prolongation pass, and that sweep moves 12 nnz + 4 (N + 1) + 16 N + 8 N_c.
    Returns (V-cycle bytes, CG shell bytes)."""
    def nnz(m):
        return 7 * m**3 - 6 * m * m
    total = 0
    m = n
    for lv in range(levels):
        N = m**3
        sweep = 12 * nnz(m) + 4 * (N + 1) + 24 * N
        if lv == levels - 1:
            total += 16 * N + (nu_coarse - 1) * sweep
        else:
            Nc = (m // 2) ** 3
            folded = fold and nu == 1
            total += (0 if folded else 16 * N) + (nu - 1) * sweep
            total += 12 * nnz(m) + 4 * (N + 1) + (16 * N if folded else 24 * N)
            fold_up = folded and N <= FOLD_UP_MAX_ROWS and m >= 2 and (m & (m - 1)) == 0
            total += 8 * N + 8 * Nc
            if fold_up:
                total += 12 * nnz(m) + 4 * (N + 1) + 16 * N + 8 * Nc
            else:
                total += (16 * N + 8 * Nc) + nu * sweep
        m //= 2
    N0 = n**3
    shell = 12 * nnz(n) + 4 * (N0 + 1) + 16 * N0 + 72 * N0
    return total, shell


def mgcg_extra(L, n: int):
    """BASELINE config 3 on one GPU: 3-level V(1,1) weighted-Jacobi MGCG on the 7-point n^3 system, plain CSR on every level
    (and, second, on the opt-in row-pattern form), solved to 1e-8 * ||b||."""
    from conjugategradient_amd import _lib
    from conjugategradient_amd.multigrid import ConjugateGradientMgGpu

    N = n**3
    tol = 1e-8 * (N ** 0.5)
    levels, nu, nuc = 3, 1, 4
    vb, shell = vcycle_bytes(n, levels, nu, nuc)
    rb, rshell = vcycle_required_bytes(n, levels, nu, nuc)
    out = {"config": f"MGCG, V({nu},{nu}) cycle, {levels} levels, weighted Jacobi (omega = 6/7), {nuc} coarse sweeps, 7-pt Poisson {n}^3, b = 1, x0 = 0, stop at ||r|| <= 1e-8 ||b||",
           "algorithmic_bytes_per_iteration": vb + shell, "vcycle_algorithmic_bytes": vb,
           "required_bytes_per_iteration": rb + rshell, "vcycle_required_bytes": rb,
           "bytes_note": "algorithmic = SURVEY.md 8d per-pass formulas (D^-1 read as an array, prolongation 8 N + 8 N_c); required = what must move as the library "
                         "runs the cycle (uniform diagonal: no D^-1 array; prolongation rewrites the fine iterate: 16 N + 8 N_c; V(1,1) on one rank: the first "
                         "sweep is folded into the residual's gathers, and on levels of up to 100 M rows the prolongation into the last sweep's).  "
                         "frac_of_peak is quoted on the REQUIRED bytes"}
    # third row: what a caller who is free to choose the cycle would run -- the hierarchy taken down to 4^3 and the cell-centred
    # linear transfer (MgSetInterpolation; profiles/r2/mgcg_levels_sweep_512_csr*.log); its bytes are its own (the transfers
    # move the same HBM bytes, their extra operands come from cache)
    deep = (8, 1, 4) if n == 512 else None
    # fourth row: the same 3-level V(1,1) cycle with 64 Jacobi sweeps on the coarsest level instead of 4 (a 128^3 sweep costs 35 us at 512^3; the
    # coarse-level error is what limits a 3-level cycle): 53 instead of 157 iterations
    rows = ([("csr", 0, (levels, nu, nuc), 0), ("row_pattern", 1, (levels, nu, nuc), 0)] + ([("csr_8_levels_linear_transfer", 0, deep, 1)] if deep else []) +
            ([("csr_3_levels_64_coarse_sweeps", 0, (3, 1, 64), 0)] if n >= 256 else []))
    for key, mode, (lv_, nu_, nuc_), interp in rows:
        mg = ConjugateGradientMgGpu(N, 7, 0, 5000, tol, (n, n, n), levels=lv_, nu=nu_, nuCoarse=nuc_, rule=_lib.RULE_CSHARP, interpolation=interp)
        vb_, shell_ = vcycle_bytes(n, lv_, nu_, nuc_)
        try:
            L.MgcgSetMatrixCompression(mg.cusparse, mode)
            t0 = time.perf_counter()
            mg.InitializePoisson()
            L.MgcgDeviceSynchronize()
            setup = time.perf_counter() - t0
            mg.Solve()                                      # warm-up (builds analyses, clocks)
            L.MgcgFill(mg.vectorX.Ptr, 0.0)
            L.MgcgDeviceSynchronize()
            t0 = time.perf_counter()
            mg.Solve()
            dt = time.perf_counter() - t0
            its = mg.Iteration + 1
            ms = 1e3 * dt / its
            rb_, rshell_ = vcycle_required_bytes(n, lv_, nu_, nuc_, fold=(interp == 0))
            out[key] = {"iterations": its, "residual": mg.Residual, "solve_s": dt, "ms_per_iteration": ms, "setup_s": setup,
                        "achieved_gbps": ((rb_ + rshell_) / (ms * 1e-3) / 1e9) if mode == 0 else None,
                        "frac_of_peak": ((rb_ + rshell_) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if mode == 0 else None,
                        "frac_of_peak_on_survey_formula_bytes": ((vb_ + shell_) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if mode == 0 else None}
            if key == "csr":                                # PMC bytes of a whole iteration (profiles/spmv_traffic.json: tools/pmc_iteration_traffic.py on separate --pmc passes)
                try:
                    pj = json.load(open(os.path.join(ROOT, "profiles", "spmv_traffic.json"))).get("mgcg_csr_iteration", {})
                    if pj.get("grid") == n:
                        out[key]["traffic"] = {"recorded": True, "hbm_bytes_per_iteration": pj.get("hbm_bytes_per_iteration"), "source": pj.get("source"),
                                               "note": "recorded evidence (profiles/), not measured by this run"}
                except Exception:       # noqa: BLE001 -- the file is evidence, not an input
                    pass
            if key != "csr" and mode == 0:
                out[key]["config"] = f"V({nu_},{nu_}), {lv_} levels, {nuc_} coarse sweeps" + (", cell-centred linear transfer" if interp else "")
                out[key]["algorithmic_bytes_per_iteration"] = vb_ + shell_
                out[key]["required_bytes_per_iteration"] = rb_ + rshell_
        finally:
            mg.Dispose()
    return out


class _watchdog:
    """N > 1: a collective that never returns would hold the whole job until the launcher's limit.  If the guarded step takes longer than
    MGCG_BENCH_STEP_TIMEOUT seconds (default 240) the rank says where it is blocked and leaves with status 4; the launcher then ends the
    other ranks.  (os._exit: the blocked call holds the GPU runtime's locks, a normal exit would wait for it.)"""

    def __init__(self, what: str, enabled: bool = True, on_fire=None, limit_env: str = "MGCG_BENCH_STEP_TIMEOUT", limit_default: str = "240"):
        self.what, self.enabled, self.timer, self.on_fire = what, enabled, None, on_fire
        self.limit_env, self.limit_default = limit_env, limit_default

    def _fire(self):
        if self.on_fire is not None:
            print(f"bench.py: {self.what} did not finish within {self.limit:.0f} s", file=sys.stderr, flush=True)
            self.on_fire(self.what)
            os._exit(0)
        print(f"bench.py: {self.what} did not finish within {self.limit:.0f} s -- leaving with status 4", file=sys.stderr, flush=True)
        os._exit(4)

    def __enter__(self):
        if self.enabled:
            import threading

            self.limit = float(os.environ.get(self.limit_env, self.limit_default))
            self.timer = threading.Timer(self.limit, self._fire)
            self.timer.daemon = True
            self.timer.start()
        return self

    def __exit__(self, *exc):
        if self.timer is not None:
            self.timer.cancel()
        return False


class _ExtrasGuard:
    """The untimed extras of an N > 1 line (config 4, probes, schedule sweeps) run stage by stage under this guard: a stage that does not
    finish within MGCG_BENCH_EXTRAS_TIMEOUT seconds (default 150) -- a collective that never returns -- ends the run with status 0 and, on
    rank 0, the line as far as it got plus `extras_aborted` = the stage.  The timed result is in `out` before the first stage starts: an
    extra can never cost it.  (os._exit: the blocked call holds the GPU runtime's locks, a normal exit would wait for it.)"""

    def __init__(self, rank: int, out: dict):
        self.rank, self.out, self.name, self.timer = rank, out, "start", None
        self.limit = float(os.environ.get("MGCG_BENCH_EXTRAS_TIMEOUT", "150"))

    def _fire(self):
        print(f"bench.py: rank {self.rank}: extras stage '{self.name}' did not finish within {self.limit:.0f} s -- ending the run with the line as far as it got",
              file=sys.stderr, flush=True)
        if self.rank == 0:
            try:
                line = json.dumps(dict(self.out, extras_aborted=self.name))
            except Exception:       # noqa: BLE001 -- (the main thread was adding a key at this very moment)
                line = json.dumps({k: v for k, v in list(self.out.items())} | {"extras_aborted": self.name}, default=str)
            print(line, flush=True)
        os._exit(0)

    def stage(self, name: str):
        import threading

        self.done()
        self.name = name
        self.timer = threading.Timer(self.limit, self._fire)
        self.timer.daemon = True
        self.timer.start()

    def done(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


def _free_port() -> int:
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv: list[str], script: str | None = None, grace_s: float = 30.0) -> int:
    """`python bench.py --gpus N` without a launcher around it: start the N ranks (one process per GPU) as children of
    this GPU-free parent -- the N-GPU timed leg of the reference's own benchmark (Mgcg/cuBlas/Mgcg/MgcgMain.cs:143-167)
    is one command there too.  The parent imports neither torch nor the library and makes no GPU call; every child gets
    RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT, as torch.distributed.run would set
    them.  Rank 0's stdout (the ONE JSON line) is forwarded to the parent's stdout, everything else to stderr.  Returns
    the worst exit code of the children (3: RCCL could not form and --allow-fallback was not given).  When a rank fails,
    the others get `grace_s` seconds to leave their collectives on their own, then exactly those PIDs are terminated."""
    import subprocess

    # Under rocprofv3 (above all with --pmc) the profiler's preloaded library has already initialised the GPU in THIS process; starting
    # the rank processes from it is the exec-after-GPU-init hop that takes a machine of this pool down.  Profile one rank instead:
    # conjugategradient_amd/tools/forced_path_run.py (a one-rank RCCL communicator on the several-ranks path).
    preload = os.environ.get("LD_PRELOAD", "")
    if "rocprof" in preload.lower() or any(k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER_")) for k in os.environ):
        print("bench.py: --gpus N > 1 refuses to start its ranks under a profiler (LD_PRELOAD / ROCP* / ROCPROFILER_* set): the parent must be GPU-free. "
              "Profile a single rank with conjugategradient_amd/tools/forced_path_run.py instead.", file=sys.stderr, flush=True)
        return 5
    script = script or os.environ.get("MGCG_BENCH_RANK_SCRIPT") or os.path.abspath(__file__)     # (the variable: tests/test_bench_launcher.py)
    env0 = dict(os.environ)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL between processes needs it on this driver
    env0.setdefault("NCCL_DEBUG", "WARN")                    # RCCL says why when a communicator does not form (stderr; silent otherwise)
    env0["MASTER_ADDR"] = "127.0.0.1"
    env0["MASTER_PORT"] = str(env0.get("MGCG_BENCH_MASTER_PORT") or _free_port())
    env0["WORLD_SIZE"] = env0["LOCAL_WORLD_SIZE"] = str(n)
    children = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        children.append(subprocess.Popen([sys.executable, script, *argv], env=env,
                                         stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    import threading

    lines: list[str] = []
    reader = threading.Thread(target=lambda: lines.extend(l.decode(errors="replace") for l in children[0].stdout), daemon=True)
    reader.start()
    failed_at = None
    # a run that has not finished after MGCG_BENCH_TIMEOUT seconds (default 900; the bench takes about a minute, a cold import of torch on a fresh box two or three) is ended the same way:
    # ranks left in a collective that will never complete must not hold the node
    deadline_all = time.monotonic() + float(os.environ.get("MGCG_BENCH_TIMEOUT", "900"))
    while True:
        codes = [c.poll() for c in children]
        if all(c is not None for c in codes):
            break
        if failed_at is None and any(c not in (None, 0) for c in codes):
            failed_at = time.monotonic()
        if failed_at is None and time.monotonic() > deadline_all:
            print("bench.py: the ranks did not finish within MGCG_BENCH_TIMEOUT; ending them", file=sys.stderr, flush=True)
            failed_at = time.monotonic() - grace_s - 1.0
        if failed_at is not None and time.monotonic() - failed_at > grace_s:
            for c in children:
                if c.poll() is None:
                    c.terminate()
            deadline = time.monotonic() + 10.0
            for c in children:
                try:
                    c.wait(timeout=max(0.1, deadline - time.monotonic()))
                except subprocess.TimeoutExpired:
                    c.kill()
                    c.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10.0)
    codes = [c.returncode for c in children]
    worst = max((128 - c) if c < 0 else c for c in codes)      # killed by a signal: the shell's convention
    json_lines = [l for l in lines if l.lstrip().startswith("{")]
    for l in lines:
        if l in json_lines[-1:]:
            continue
        sys.stderr.write(l)
    if json_lines:
        sys.stdout.write(json_lines[-1] if json_lines[-1].endswith("\n") else json_lines[-1] + "\n")
        sys.stdout.flush()
    elif worst == 0:
        print("bench.py: the ranks exited cleanly but rank 0 printed no JSON line", file=sys.stderr)
        worst = 1
    return worst


def _max_over_ranks(dist, values):
    import torch

    t = torch.tensor([float(v) for v in values], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def _tuning_get(L, name: bytes) -> int:
    v = C.c_int(0)
    L.MgcgGetTuning(name, C.byref(v))
    return v.value


SCHEDULES = (("exchange_in_line", 0, 0), ("interior_rows_on_side_stream", 2, 0))
# RCCL on two streams of one communicator (the exchange on the side stream, the all-reduces on the main stream) has never run between real
# devices: this schedule is measured LAST, after everything else is in the line, so that a hang there costs nothing but itself
SCHEDULE_LAST = ("exchange_on_side_stream", 2, 1)


def comm_probe_extra(L, dist, comm, n: int):
    """MgcgCommProbe on the live communicator (collective; every rank makes the same calls): what one collective step of the
    several-ranks loop costs on its stream, max over ranks, microseconds.  The sums of ConjugateGradientParallelGpu.cs:463,499,525
    are the 8-byte all-reduces, the preconditioned loop's {r.r, r.z} the 16-byte one, SyncP (:384-419) the plane exchange."""
    plane = n * n
    rows = (("allreduce_8B_us", 0, 1, 200), ("allreduce_16B_us", 0, 2, 200), ("neighbour_exchange_8B_us", 4, 1, 100),
            ("neighbour_exchange_one_plane_us", 4, plane, 50), ("neighbour_exchange_quarter_plane_us", 4, plane // 4, 50),
            ("fork_join_us", 2, 0, 200), ("kernel_boundary_us", 3, 0, 200))
    vals = [L.MgcgCommProbe(comm, what, count, reps) for _, what, count, reps in rows]
    L.MgcgClearLastError()
    vals = _max_over_ranks(dist, [v if v == v else -1.0 for v in vals])
    out = {k: (v if v >= 0 else None) for (k, *_), v in zip(rows, vals)}
    out["plane_bytes"] = 8 * plane
    out["transport"] = L.MgcgCommTransport(comm).decode()
    out["note"] = "max over ranks; neighbour_exchange = one grouped send/recv with ranks rank-1 and rank+1 (the z-slab stencil's halo); null = nothing to time on the device (host-staged transports move their planes through the launcher)"
    return out


def cg_schedules_extra(L, dist, cg, schedules, iters: int = 20, with_default: bool = True):
    """ms per CG iteration under the given halo schedules (untimed extras; every rank sets the same knobs), and what the
    library's own measured rule (overlap = 1) chose for this plan."""
    saved = (_tuning_get(L, b"overlap"), _tuning_get(L, b"halo_stream"))
    out = {}
    try:
        for key, ov, hs in schedules:
            L.MgcgSetTuning(b"overlap", ov)
            L.MgcgSetTuning(b"halo_stream", hs)
            cg.Steps(3, restart=True)
            dist.barrier()
            L.MgcgDeviceSynchronize()
            t0 = time.perf_counter()
            cg.Steps(iters, restart=False)
            L.MgcgDeviceSynchronize()
            dt = time.perf_counter() - t0
            out[key] = {"ms_per_iteration": _max_over_ranks(dist, [dt / iters * 1e3])[0], "overlap_active_rank0": bool(L.MgcgLastOverlap(None))}
    finally:
        L.MgcgSetTuning(b"overlap", saved[0])
        L.MgcgSetTuning(b"halo_stream", saved[1])
    if with_default:
        cg.Steps(3, restart=True)
        dist.barrier()
        L.MgcgDeviceSynchronize()
        t0 = time.perf_counter()
        cg.Steps(iters, restart=False)
        L.MgcgDeviceSynchronize()
        dt = time.perf_counter() - t0
        us = (C.c_double * 2)(0.0, 0.0)
        measured = bool(L.MgcgLastOverlapTimes(us))
        out["library_default"] = {"ms_per_iteration": _max_over_ranks(dist, [dt / iters * 1e3])[0], "overlap_active_rank0": bool(L.MgcgLastOverlap(None)),
                                  "overlap_knob": saved[0], "halo_stream_knob": saved[1], "decided_by_measurement": measured,
                                  "measured_exchange_in_line_us": us[0] if measured else None, "measured_fork_launch_join_us": us[1] if measured else None}
        out["iterations_each"] = iters
    return out


def _mgcg_fixed(L, mg, k: int) -> tuple[float, float]:
    """k MGCG iterations from x = 0 (rule NATIVE with an infinite tolerance stops exactly at index k - 1); returns (seconds, residual)."""
    L.MgcgFill(mg.vectorX.Ptr, 0.0)
    mg.MinIteration = k - 1
    L.MgcgDeviceSynchronize()
    t0 = time.perf_counter()
    mg.Solve()
    L.MgcgDeviceSynchronize()
    return time.perf_counter() - t0, mg.Residual


def _mgcg_ms_per_iteration(L, m, k_short, k_long, sync):
    """ms per MGCG iteration from the difference of a long and a short fixed-count solve (what a solve costs before its first iteration --
    r = b - A x, the first V-cycle -- cancels), the smaller of two runs each; returns (ms, residual after k_long iterations).  On a problem so
    small that the difference drowns in jitter the long run's plain average is returned instead (never a non-positive figure)."""
    t_s, t_l, res = [], [], None
    for _ in range(2):
        sync()
        t_s.append(_mgcg_fixed(L, m, k_short)[0])
        sync()
        t, res = _mgcg_fixed(L, m, k_long)
        t_l.append(t)
    sync()
    ms = (min(t_l) - min(t_s)) / (k_long - k_short) * 1e3
    if ms <= 0.05 * min(t_l) / k_long * 1e3:
        ms = min(t_l) / k_long * 1e3
    return ms, res


def _mgcg_schedule(L, dist, build, rank, world, comm, sch, k_short, k_long, deep_halo=None):
    """ms per MGCG iteration of a freshly set-up hierarchy under one halo schedule (collective).  deep_halo = 0: round 4's cycle (one exchange
    per SpMV-shaped pass) instead of the deep-halo cycle (one per level)."""
    key, ov, hs = sch
    saved = (_tuning_get(L, b"overlap"), _tuning_get(L, b"halo_stream"), _tuning_get(L, b"deep_halo"))
    try:
        L.MgcgSetTuning(b"overlap", ov)
        L.MgcgSetTuning(b"halo_stream", hs)
        if deep_halo is not None:
            L.MgcgSetTuning(b"deep_halo", deep_halo)
        m2 = build(rank, world, comm)
        try:
            _mgcg_fixed(L, m2, k_short)
            ms2, _ = _mgcg_ms_per_iteration(L, m2, k_short, k_long, dist.barrier)
            return {"ms_per_iteration": _max_over_ranks(dist, [ms2])[0], "folds_rank0": int(L.MgcgLastVcycleFolds())}
        finally:
            m2.Dispose()
    finally:
        L.MgcgSetTuning(b"overlap", saved[0])
        L.MgcgSetTuning(b"halo_stream", saved[1])
        L.MgcgSetTuning(b"deep_halo", saved[2])


def mgcg_multirank_extra(L, _lib, dist, a, rank, world, local_rank, comm, n: int, stage):
    """BASELINE config 4 inside the N > 1 line: the row-partitioned 3-level V(1,1) Jacobi MGCG on the same communicator."""
    from conjugategradient_amd.parallel import ConjugateGradientMgRankGpu

    N = n**3
    levels, nu, nuc = 3, 1, 4
    tol = 1e-8 * (N ** 0.5)
    k_short = 5

    def build(rk, wd, cm):
        m = ConjugateGradientMgRankGpu(N, 7, 0, 10**9, 1e300, (n, n, n), rank=rk, world=wd, comm=cm, device=local_rank, rule=_lib.RULE_NATIVE,
                                       levels=levels, nu=nu, nuCoarse=nuc)
        L.MgcgSetMatrixCompression(m.cusparse, 0)
        m.InitializePoisson(n, n, n)
        m.Setup()
        return m

    def measure(m, sync, k_long=None):
        """(ms per iteration from the difference of a long and a short fixed run -- the set-up of a solve cancels --, residual after k_long
        iterations, iterations / seconds / residual of the solve to 1e-8 ||b||, k_long).  k_long is chosen from the iteration count of the
        solve -- at most 25 and at most half of it, so that the fixed-count residual compared between the partitioned and the one-rank run
        sits well above round-off (512^3: 157 iterations, k_long = 25)."""
        _mgcg_fixed(L, m, k_short)                                 # warm-up: halo plans, analyses, clocks
        m.rule, m.AllowableResidual, m.MinIteration, m.MaxIteration = _lib.RULE_CSHARP, tol, 0, 5000
        L.MgcgFill(m.vectorX.Ptr, 0.0)
        L.MgcgDeviceSynchronize()
        sync()
        t0 = time.perf_counter()
        m.Solve()
        L.MgcgDeviceSynchronize()
        dt = time.perf_counter() - t0
        its, res = m.Iteration + 1, m.Residual
        m.rule, m.AllowableResidual, m.MaxIteration = _lib.RULE_NATIVE, 1e300, 10**9
        if k_long is None:
            k_long = max(k_short + 4, min(25, its // 2))
        ms_it, res_fixed = _mgcg_ms_per_iteration(L, m, k_short, k_long, sync)
        return ms_it, res_fixed, its, dt, res, k_long

    out = {"config": f"row-partitioned MGCG, V({nu},{nu}) cycle, {levels} levels, weighted Jacobi (omega = 6/7), {nuc} coarse sweeps, 7-pt Poisson {n}^3 over {world} z-slabs, "
                     f"b = 1, x0 = 0, plain CSR on every level (BASELINE config 4)"}
    stage("mgcg: set-up of the partitioned hierarchy")
    t0 = time.perf_counter()
    mg = build(rank, world, comm)
    L.MgcgDeviceSynchronize()
    dist.barrier()
    out["setup_s"] = time.perf_counter() - t0
    try:
        stage("mgcg: partitioned iterations and solve")
        ms, res_fixed, its, dt, res, k_long = measure(mg, dist.barrier)
        ms, dt = _max_over_ranks(dist, [ms, dt])
        out.update({"ms_per_iteration": ms, "iterations_timed": k_long - k_short, "iterations_to_1e-8": its, "solve_s": dt, "residual": res,
                    "folds_rank0": int(L.MgcgLastVcycleFolds())})
    finally:
        mg.Dispose()
    # the same solve by ONE rank on rank 0's GPU (the other GPUs idle): the 1-GPU leg of the reference's benchmark in the same run
    stage("mgcg: single-rank replay on rank 0")
    if rank == 0:
        single = None
        try:
            free_b, total_b = C.c_longlong(0), C.c_longlong(0)
            L.MgcgMemGetInfo(C.byref(free_b), C.byref(total_b))
            need = (12 * 7 + 8 * 14) * N * 1.3
            if free_b.value < need:
                out["parity_vs_single_rank"] = {"skipped": f"{free_b.value / 1e9:.0f} GB free on rank 0's device, the single-rank replay needs about {need / 1e9:.0f} GB"}
            else:
                single = build(0, 1, None)
                ms1, res1, its1, dt1, resc1, _ = measure(single, lambda: None, k_long)
                rel = abs(res_fixed - res1) / abs(res1) if res1 else float("inf")
                out["single_rank"] = {"ms_per_iteration": ms1, "iterations_to_1e-8": its1, "solve_s": dt1, "residual": resc1}
                out["speedup_vs_single_rank"] = {"per_iteration": ms1 / ms if ms > 0 else None, "solver_time": dt1 / dt if dt > 0 else None,
                                                 "note": "the one-rank leg ran on rank 0's GPU in this same run while the other GPUs idled (the reference's benchmark times both legs in one run, MgcgMain.cs:143-167)"}
                out["parity_vs_single_rank"] = {"after_iterations": k_long, "single_rank_residual": res1, "partitioned_residual": res_fixed, "relative_difference": rel,
                                                "within_1e-10": bool(rel <= 1e-10), "same_iterations_to_1e-8": bool(its1 == its)}
        except Exception as ex:     # noqa: BLE001 -- the check must never cost the line
            out["parity_vs_single_rank"] = {"skipped": f"single-rank replay failed: {ex}"}
            L.MgcgClearLastError()
        finally:
            if single is not None:
                single.Dispose()
    dist.barrier()
    # the halo schedules (the hierarchy plans its overlap at set-up: one hierarchy per schedule); the exchange-on-side-stream one comes
    # last of all extras (main)
    stage("mgcg: halo schedules")
    sched = {}
    for sch in SCHEDULES:
        sched[sch[0]] = _mgcg_schedule(L, dist, build, rank, world, comm, sch, k_short, k_long)
    # the library's default schedule with round 4's cycle (an exchange before every SpMV-shaped pass: 8 per iteration) next to the deep-halo cycle (4)
    sched["per_pass_exchanges"] = _mgcg_schedule(L, dist, build, rank, world, comm, ("per_pass_exchanges", _tuning_get(L, b"overlap"), _tuning_get(L, b"halo_stream")), k_short, k_long, deep_halo=0)
    sched["library_default"] = {"ms_per_iteration": ms, "overlap_knob": _tuning_get(L, b"overlap"), "halo_stream_knob": _tuning_get(L, b"halo_stream"),
                                "deep_halo_knob": _tuning_get(L, b"deep_halo"), "folds_rank0": out.get("folds_rank0")}
    out["schedules"] = sched
    out["_k"] = (k_short, k_long)
    out["_build"] = build
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE=1")

    dist = None
    if a.torch_first:
        import torch  # noqa: F401
    if world > 1:
        # torch first: its bundled HIP runtime and RCCL then serve the whole process (one HIP, one RCCL).
        import torch  # noqa: F401
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("gloo")
        dist = dist_mod

    from conjugategradient_amd import _lib
    from conjugategradient_amd.parallel import ConjugateGradientRankGpu

    L = _lib.lib()
    _lib.require_gpu()
    n = a.grid
    N = n**3
    if world > 1 and n % world:
        raise SystemExit("grid extent must be divisible by the number of GPUs (z-slab partition)")

    # The RCCL communicator is formed BEFORE any other GPU work of this rank (handles, vectors, the matrix).
    transport = "single rank"
    comm = None
    if world > 1:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        if L.GetDeviceCount() < local_world and a.allow_fallback:
            # fewer devices than ranks (the one-GPU rehearsal of the N > 1 path): RCCL cannot form, the host-staged transport will be
            # used, and the ranks share the device(s) round-robin -- never a scaling result, the line says so
            L.MgcgSetTuning(b"virtual_devices", local_world)
        L.SetDevice(local_rank)
        L.MgcgClearLastError()          # (a rank without a device of its own is reported by the precondition below, not here)
        # RCCL communicator (the unique id travels over gloo).  Should it fail to form on this host, every rank falls back
        # to the library's host-staged callback transport over the same gloo group: slow, but the scaling line stays valid.
        import torch
        from conjugategradient_amd.parallel import create_callback_comm, create_comm

        ok, why = 1, ""
        if os.environ.get("MGCG_BENCH_TRANSPORT", "rccl") != "rccl":
            ok, why = 0, "MGCG_BENCH_TRANSPORT asked for the host-staged transport"
        else:
            # ncclCommInitRank is collective: agree over gloo that every rank CAN enter it before any rank does
            # (librccl resolves, one device per local rank), otherwise healthy ranks would block inside it.
            pre, pre_why = 1, ""
            try:
                if torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
                    pre, pre_why = 0, f"{torch.cuda.device_count()} device(s) for {os.environ.get('LOCAL_WORLD_SIZE', world)} local rank(s)"
                elif L.MgcgRcclAvailable() != 1:
                    pre, pre_why = 0, "librccl did not resolve: " + _lib.last_error()
            except Exception as ex:     # noqa: BLE001
                pre, pre_why = 0, str(ex)
            pflag = torch.tensor([pre], dtype=torch.int64)
            dist.all_reduce(pflag, op=dist.ReduceOp.MIN)
            if int(pflag[0]) != 1:
                ok, why = 0, pre_why or "another rank failed the RCCL precondition"
            else:
                try:
                    with _watchdog(f"rank {rank}: ncclCommInitRank / the first all-reduce over {world} ranks"):
                        comm = create_comm(rank, world)
                        probe = L.MgcgCommAllReduceSum(comm, 1.0)
                    if probe != float(world):
                        ok, why = 0, f"RCCL all-reduce probe returned {probe}"
                except Exception as ex:     # noqa: BLE001 -- any failure of the RCCL bootstrap
                    ok, why = 0, str(ex)
                    L.MgcgClearLastError()
        flag = torch.tensor([ok], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 1:
            transport = "rccl"
        else:
            if not a.allow_fallback and os.environ.get("MGCG_BENCH_TRANSPORT", "rccl") == "rccl":
                if rank == 0:
                    print(f"bench.py: RCCL transport unavailable ({why or 'another rank failed'}); refusing to report a scaling line over a fallback "
                          "(pass --allow-fallback to run over host-staged gloo)", file=sys.stderr, flush=True)
                if comm:
                    L.MgcgCommDestroy(comm)
                dist.barrier()
                dist.destroy_process_group()
                raise SystemExit(3)
            if rank == 0:
                print(f"bench.py: RCCL transport unavailable ({why or 'another rank failed'}); using the host-staged gloo transport (--allow-fallback)", file=sys.stderr, flush=True)
            if comm:
                L.MgcgCommDestroy(comm)
            comm = create_callback_comm(rank, world)
            transport = "host-staged callbacks over torch.distributed gloo (RCCL FALLBACK: not a scaling result)"
    if a.solver == "mgcg":
        from conjugategradient_amd.parallel import ConjugateGradientMgRankGpu

        # fixed-length runs: rule NATIVE with an infinite tolerance stops exactly at index minIteration
        cg = ConjugateGradientMgRankGpu(N, 7, 0, 10**9, 1e300, (n, n, n), rank=rank, world=world, device=local_rank,
                                        rule=_lib.RULE_NATIVE, levels=a.mg_levels, nu=a.mg_nu, nuCoarse=a.mg_nu_coarse, interpolation=a.mg_interpolation)
    else:
        cg = ConjugateGradientRankGpu(N, 7, 0, 10**9, 1e-8, rank=rank, world=world, device=local_rank)
    if a.compression is None:
        a.compression = 0
    a.no_compression = a.compression == 0
    L.MgcgSetMatrixCompression(cg.cusparse, a.compression)
    if comm is not None:
        cg.comm = comm
        cg._own_comm = True
    if a.spmv_kernel is not None:
        L.MgcgSetSpmvKernel(cg.cusparse, a.spmv_kernel)
    if a.spmv_rows is not None or a.spmv_flags is not None or a.spmv_grid is not None:
        L.MgcgSetSpmvTuning(cg.cusparse, a.spmv_rows or 64, a.spmv_flags or 0, a.spmv_grid or 0)
    cg.InitializePoisson(n, n, n)
    L.MgcgDeviceSynchronize()
    _lib.check("setup")
    nnz_local = cg.part.elementCount
    rows_local = cg.part.count

    def run_steps(k, restart, solver=None):
        s_ = cg if solver is None else solver
        if a.solver == "mgcg":          # init (r = b - A x, z = M^-1 r) is inside: about one extra V-cycle per call
            L.MgcgFill(s_.vectorX.Ptr, 0.0)
            s_.MinIteration = k - 1
            s_.Solve()
            return s_.Residual
        return s_.Steps(k, restart=restart)

    if a.solver == "mgcg":
        cg.Setup()
    # warm-up (also builds the RCCL communicator and the halo plans)
    with _watchdog(f"rank {rank}: the warm-up iterations (halo plan, first exchanges and all-reduces)", enabled=world > 1):
        run_steps(max(a.warmup, 1), True)

    # the library's placement draw (first solve on vectors beyond the Infinity Cache: the loop's SpMV timed on extra allocations of Ap, then of p; the fastest kept)
    placement = None
    stages = {}
    for which, name in ((0, "Ap (the SpMV's output)"), (1, "p (its gathered input)")):
        pl_ms = (C.c_double * 16)()
        pl_chosen = C.c_int(-1)
        pl_n = L.MgcgLastPlacement(which, pl_ms, 16, C.byref(pl_chosen))
        if pl_n > 0:
            stages[name] = {"candidates_spmv_ms": [pl_ms[i] for i in range(min(pl_n, 16))], "chosen": pl_chosen.value}
    if stages:
        placement = dict(stages, note="one-off, inside the warm-up: candidate 0 is the allocation the vector came with (MGCG_PLACEMENT=0 switches the draw off)")

    def barrier():
        if dist is not None:
            dist.barrier()

    L.MgcgProfileSpmv(cg.cusparse, 1)
    barrier()
    L.MgcgDeviceSynchronize()
    with _watchdog(f"rank {rank}: the timed steps", enabled=world > 1):        # (a sleeping timer thread: nothing inside the timed region)
        t0 = time.perf_counter()
        res = run_steps(a.steps, False)             # synchronises the stream before returning
        L.MgcgDeviceSynchronize()
        barrier()
        dt = time.perf_counter() - t0
    launches = C.c_int(0)
    spmv_ms_total = L.MgcgProfileSpmvMs(cg.cusparse, C.byref(launches))
    L.MgcgProfileSpmv(cg.cusparse, 0)


    # What this box streams ("% of attainable" next to "% of spec"): read-only = Dot over two vectors,
    # copy = the library's Copy export (device-to-device), on the same 1 GiB vectors; and the plain CsrMV export (beta = 0,
    # no fused dot) on the same matrix.
    csr_ms = 0.0
    read_gbps = copy_gbps = 0.0
    if world == 1 and a.solver == "cg":
        L.MgcgSetMatrixCompression(cg.cusparse, 0)
        ev0, ev1 = L.MgcgEventCreate(), L.MgcgEventCreate()
        ptrs = (cg.vectorAp.ToRawPtr(), cg.vectorElements.ToRawPtr(), cg.vectorRowOffsets.ToRawPtr(), cg.vectorColumnIndeces.ToRawPtr(), cg.vectorP.ToRawPtr())
        for _ in range(3):
            L.CsrMV(cg.cusparse, cg.matDescr, *ptrs, nnz_local, rows_local, N, 1.0, 0.0)
        L.MgcgEventRecord(ev0)
        for _ in range(20):
            L.CsrMV(cg.cusparse, cg.matDescr, *ptrs, nnz_local, rows_local, N, 1.0, 0.0)
        L.MgcgEventRecord(ev1)
        csr_ms = L.MgcgEventElapsedMs(ev0, ev1) / 20
        L.MgcgSetMatrixCompression(cg.cusparse, a.compression)
        pr, pa = cg.vectorR.ToRawPtr(), cg.vectorAp.ToRawPtr()
        L.Dot(cg.cublas, pa, pr, rows_local)
        L.MgcgEventRecord(ev0)
        for _ in range(10):
            L.Dot(cg.cublas, pa, pr, rows_local)
        L.MgcgEventRecord(ev1)
        read_gbps = 16 * rows_local / (L.MgcgEventElapsedMs(ev0, ev1) / 10 * 1e-3) / 1e9
        L.MgcgEventRecord(ev0)
        for _ in range(10):
            L.Copy(cg.cublas, pa, pr, rows_local, 0, 0)
        L.MgcgEventRecord(ev1)
        copy_gbps = 16 * rows_local / (L.MgcgEventElapsedMs(ev0, ev1) / 10 * 1e-3) / 1e9

    # ---- secondary figures (N = 1 only, never part of value / roofline)
    lossless = None
    if world == 1 and a.solver == "cg" and a.compression == 0 and not a.no_extras:
        lossless = {"note": "opt-in lossless re-encodings of the same matrix (MgcgSetMatrixCompression); results bit-identical; "
                            "NOT the BASELINE metric: the CSR arrays are not streamed, so no CSR roofline fraction is quoted"}
        pmc_all = {}
        try:
            pmc_all = json.load(open(os.path.join(ROOT, "profiles", "spmv_traffic.json")))
        except Exception:   # noqa: BLE001
            pass
        for mode, key in ((1, "row_pattern"), (2, "per_nonzero_codes")):
            try:
                L.MgcgSetMatrixCompression(cg.cusparse, mode)
                cg.Steps(5, restart=True)                   # builds the form, warms up
                cls = L.MgcgAnalysisInfo(cg.cusparse, 0, None, None, None, None)
                L.MgcgProfileSpmv(cg.cusparse, 1)
                L.MgcgDeviceSynchronize()
                t1 = time.perf_counter()
                cg.Steps(30, restart=False)
                L.MgcgDeviceSynchronize()
                d1 = time.perf_counter() - t1
                ln = C.c_int(0)
                ms = L.MgcgProfileSpmvMs(cg.cusparse, C.byref(ln)) / max(ln.value, 1)
                L.MgcgProfileSpmv(cg.cusparse, 0)
                own = {3: 17 * rows_local, 2: 2 * nnz_local + 4 * (rows_local + 1) + 16 * rows_local, 1: 9 * nnz_local + 4 * (rows_local + 1) + 16 * rows_local}.get(cls)
                fkey = {3: "pattern", 2: "dcsr", 1: "dcsr"}.get(cls)
                moved = (pmc_all.get(fkey, {}) or {}).get("hbm_bytes_per_launch") if (pmc_all.get(fkey, {}) or {}).get("grid") == n else None
                lossless[key] = {"analysis_class": cls, "iterations_per_s": 30 / d1, "ms_per_iteration": d1 / 30 * 1e3, "spmv_avg_launch_ms": ms,
                                 "own_algorithmic_bytes_per_launch": own,
                                 "own_bytes_frac_of_peak": (own / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if (own and ms > 0) else None,
                                 "pmc_bytes_per_launch": moved}
            except Exception as ex:   # noqa: BLE001 -- secondary figure
                lossless[key] = {"error": str(ex)}
                L.MgcgClearLastError()
        L.MgcgSetMatrixCompression(cg.cusparse, 0)
        L.MgcgAnalysisClear(cg.cusparse)

    if dist is not None:
        import torch

        mine = torch.tensor([dt, spmv_ms_total / max(launches.value, 1)], dtype=torch.float64)
        every = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(every, mine)                        # (per-rank figures travel in the line: a slow rank or link shows by itself)
        per_rank = {"seconds_for_the_timed_steps": [float(v[0]) for v in every], "spmv_avg_launch_ms": [float(v[1]) for v in every]}
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, spmv_ms = float(t[0]), float(t[1])
    else:
        per_rank = None
        spmv_ms = spmv_ms_total / max(launches.value, 1)

    # which form the analysis chose for the fine matrix (class 3 one byte per row, 2/1 per-nonzero codes, 0 plain CSR)
    cls = L.MgcgAnalysisInfo(cg.cusparse, 0, None, None, None, None) if a.compression else 0
    fmt = {3: "pattern", 2: "dcsr", 1: "dcsr"}.get(cls, "csr")
    fmt_text = {"pattern": "lossless row-pattern form (1 byte per row: 27 distinct rows-as-sequences; opt-in analysis, results bit-identical)",
                "dcsr": "lossless dictionary-compressed CSR (2 B/nnz; opt-in analysis, results bit-identical)",
                "csr": "plain CSR (12 B/nnz)"}[fmt]
    overlap = None
    if world > 1 and a.solver == "cg":
        ov = (C.c_longlong * 2)(0, 0)
        overlap = {"active": bool(L.MgcgLastOverlap(ov)), "interior_rows": [int(ov[0]), int(ov[1])], "local_rows": int(rows_local)}
    # N > 1 (untimed): rank 0 replays the same iterations as ONE rank on its own GPU (the whole grid fits one card) and the line
    # carries both residuals -- the partitioned run checks itself against the single-rank loop (north star: 1e-10 relative)
    parity = None
    if world > 1 and a.compression == 0:
        if rank == 0:
            single = None
            try:
                free_b, total_b = C.c_longlong(0), C.c_longlong(0)
                L.MgcgMemGetInfo(C.byref(free_b), C.byref(total_b))
                need = (12 * 7 + 8 * (14 if a.solver == "mgcg" else 6)) * N * 1.3
                if free_b.value < need:
                    parity = {"skipped": f"{free_b.value / 1e9:.0f} GB free on rank 0's device, the single-rank replay needs about {need / 1e9:.0f} GB"}
                else:
                    if a.solver == "mgcg":
                        single = ConjugateGradientMgRankGpu(N, 7, 0, 10**9, 1e300, (n, n, n), rank=0, world=1, device=local_rank, rule=_lib.RULE_NATIVE,
                                                            levels=a.mg_levels, nu=a.mg_nu, nuCoarse=a.mg_nu_coarse, interpolation=a.mg_interpolation)
                    else:
                        single = ConjugateGradientRankGpu(N, 7, 0, 10**9, 1e-8, rank=0, world=1, device=local_rank)
                    L.MgcgSetMatrixCompression(single.cusparse, 0)
                    single.InitializePoisson(n, n, n)
                    if a.solver == "mgcg":
                        single.Setup()
                    run_steps(max(a.warmup, 1), True, single)
                    L.MgcgDeviceSynchronize()
                    t1 = time.perf_counter()
                    r1 = run_steps(a.steps, False, single)
                    L.MgcgDeviceSynchronize()
                    dt1 = time.perf_counter() - t1
                    rel = abs(res - r1) / abs(r1) if r1 else float("inf")
                    parity = {"single_rank_residual_after_steps": r1, "partitioned_residual_after_steps": res, "relative_difference": rel, "within_1e-10": bool(rel <= 1e-10),
                              # the 1-GPU leg of the same run (rank 0's GPU, the others idle), as the reference's benchmark times it (MgcgMain.cs:143-167)
                              "single_rank_ms_per_step": dt1 / a.steps * 1e3, "speedup_vs_single_rank": dt1 / dt if dt > 0 else None}
            except Exception as ex:     # noqa: BLE001 -- the check must never cost the line
                parity = {"skipped": f"single-rank replay failed: {ex}"}
                L.MgcgClearLastError()
            finally:
                if single is not None:
                    single.Dispose()
        barrier()
    if rank == 0:
        spmv_bytes = 12 * nnz_local + 4 * (rows_local + 1) + 16 * rows_local     # per launch, per GPU
        achieved = spmv_bytes / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0
        nnz_total = 7 * N - 6 * n * n
        iter_bytes = 12 * nnz_total + 4 * (N + 1) + 16 * N + 72 * N
        if a.solver == "mgcg":                              # V-cycle + shell (SURVEY.md 8d per-pass formulas)
            iter_bytes = sum(vcycle_bytes(n, a.mg_levels, a.mg_nu, a.mg_nu_coarse))
        traffic = traffic_source = None
        pmc_file = os.path.join(ROOT, "profiles", "spmv_traffic.json")
        if os.path.exists(pmc_file) and world == 1:
            try:
                pj = json.load(open(pmc_file)).get(fmt, {})
                if pj.get("grid") == n:
                    traffic = pj.get("hbm_bytes_per_launch")
                    traffic_source = "RECORDED, not measured by this run: " + str(pj.get("source")) + " (separate rocprofv3 --pmc passes of this command on the builder's box; profiles/spmv_traffic.json)"
            except Exception:
                traffic = traffic_source = None
        out = {
            "metric": (f"CG iterations/sec (7-pt Poisson {n}^3); SpMV achieved HBM GB/s in roofline" if a.solver == "cg"
                       else f"MGCG iterations/sec ({a.mg_levels}-level V({a.mg_nu},{a.mg_nu}) Jacobi{', linear transfer' if a.mg_interpolation else ''}, 7-pt Poisson {n}^3)"),
            "value": a.steps / dt,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": ("unpreconditioned CG iteration" if a.solver == "cg" else "MGCG iteration (V-cycle + CG)") + f", 7-point Poisson {n}^3 CSR (fp64 values, int32 indices), "
                                   f"b=1, x0=0, {world} z-slab partition(s)",
                       "rows": N, "nnz": nnz_total, "parallelism": f"row-range dp{world}", "transport": transport},
            "iteration_algorithmic_gbps": iter_bytes / (dt / a.steps) / 1e9,
            # (several ranks fold the first sweep and the prolongation for their INTERIOR rows only and keep stored iterates in the zones next to the
            #  slab boundaries: no closed-form count is quoted for that schedule -- the single-rank figure would overstate the rate)
            "iteration_required_gbps": (sum(vcycle_required_bytes(n, a.mg_levels, a.mg_nu, a.mg_nu_coarse, fold=(a.mg_interpolation == 0))) / (dt / a.steps) / 1e9
                                        if (a.solver == "mgcg" and world == 1) else None),
            "gflops": (2 * nnz_total + 10 * N) * (a.steps / dt) / 1e9 if a.solver == "cg" else None,
            "residual_after_steps": res,
        }
        if placement is not None:
            out["placement_draw_rank0"] = placement
        if parity is not None:
            out["parity_vs_single_rank"] = parity
        if per_rank is not None:
            out["per_rank"] = per_rank
        try:    # PMC bytes of one whole iteration (evidence file written by tools/pmc_iteration_traffic.py from separate --pmc passes; N = 1, plain CSR)
            key = {"cg": "cg_csr_iteration", "mgcg": "mgcg_csr_iteration"}[a.solver]
            pj = json.load(open(os.path.join(ROOT, "profiles", "spmv_traffic.json"))).get(key, {})
            standard_cycle = a.solver == "cg" or (a.mg_levels, a.mg_nu, a.mg_nu_coarse, a.mg_interpolation) == (3, 1, 4, 0)
            if world == 1 and fmt == "csr" and pj.get("grid") == n and standard_cycle:
                out["iteration_traffic"] = {"recorded": True, "hbm_bytes_per_iteration": pj.get("hbm_bytes_per_iteration"), "source": pj.get("source"),
                                            "note": "evidence recorded under profiles/ (PMC passes on the builder's box), repeated here for reference -- NOT measured by this run"}
                out["iteration_algorithmic_bytes"] = iter_bytes
        except Exception:       # noqa: BLE001 -- the file is evidence, not an input
            pass
        if fmt == "csr":
            out["roofline"] = {"bound": "hbm",
                               "kernel": "spmv_rowtile_kernel<EPI_DOT> (CSR SpMV fused with p.Ap) on the plain CSR arrays (12 B/nnz), timed inside the CG loop",
                               "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                               "algorithmic_bytes_per_launch": spmv_bytes, "avg_launch_ms": spmv_ms, "launches_timed": launches.value}
        else:
            # a compact form was asked for explicitly: the bytes that really move are the PMC figure, never the CSR count
            moved = traffic
            out["roofline"] = {"bound": "hbm", "kernel": ("spmv_pattern_kernel" if fmt == "pattern" else "spmv_rows_kernel<DCSR>") + " on " + fmt_text + " -- NOT the BASELINE CSR metric",
                               "achieved": (moved / (spmv_ms * 1e-3) / 1e9) if (moved and spmv_ms > 0) else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": (moved / (spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if (moved and spmv_ms > 0) else None, "traffic": traffic, "traffic_source": traffic_source,
                               "algorithmic_bytes_per_launch": None, "avg_launch_ms": spmv_ms, "launches_timed": launches.value}
        if csr_ms > 0:
            out["roofline_csr_spmv"] = {"bound": "hbm", "kernel": "spmv_rowtile_kernel<EPI_AXPBY> on the plain CSR arrays (CsrMV export, beta = 0, 20 launches timed with HIP events)",
                                        "achieved": spmv_bytes / (csr_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                        "frac": spmv_bytes / (csr_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "avg_launch_ms": csr_ms,
                                        "measured_read_only_gbps": read_gbps, "measured_copy_gbps": copy_gbps,
                                        "frac_of_measured_read_only": spmv_bytes / (csr_ms * 1e-3) / 1e9 / read_gbps if read_gbps > 0 else None}
        if overlap is not None:
            out["config"]["halo_overlap_rank0"] = overlap
        out["config"]["matrix_format_in_loop"] = fmt_text + ("" if fmt == "csr" else f" -- built once by MgcgSetMatrixCompression({a.compression})")
        if lossless is not None:
            out["lossless_forms"] = lossless
        if world == 1 and not a.no_cpu_baseline:
            # the HIP loop on the SAME iterations of the SAME full-size system the oracle sample runs (x0 = 0, b = 1, plain CSR): a parity datum at
            # BASELINE size in every line (ConjugateGradientCpu.cs:45-98 is what the oracle restates)
            gpu_same = None
            if a.solver == "cg":
                try:
                    L.MgcgSetMatrixCompression(cg.cusparse, 0)
                    L.MgcgFill(cg.vectorX.Ptr, 0.0)
                    gpu_same = cg.Steps(a.cpu_iters, restart=True)
                except Exception:       # noqa: BLE001
                    L.MgcgClearLastError()
            # ... and once more in the library's validation mode (dot_order = 1: every sum in the reference's order, LongVector.cs:15-31) -- then the
            # residual must EQUAL the oracle's, bit for bit (untimed; ~0.5 s per 1.3e8-term sum)
            gpu_ref_order = None
            if gpu_same is not None:
                try:
                    L.MgcgSetTuning(b"dot_order", 1)
                    L.MgcgFill(cg.vectorX.Ptr, 0.0)
                    gpu_ref_order = cg.Steps(a.cpu_iters, restart=True)
                except Exception:       # noqa: BLE001
                    L.MgcgClearLastError()
                finally:
                    L.MgcgSetTuning(b"dot_order", 0)
            cb = cpu_baseline(n, a.cpu_iters)
            if gpu_ref_order is not None and cb.get("grid") == n and cb.get("residual"):
                cb["gpu_residual_in_reference_order_mode"] = gpu_ref_order
                cb["gpu_equals_oracle_in_reference_order_mode"] = bool(gpu_ref_order == cb["residual"])
            if gpu_same is not None and cb.get("grid") == n and cb.get("residual"):
                cb["gpu_residual_after_same_iterations"] = gpu_same
                cb["gpu_vs_oracle_relative_difference"] = abs(gpu_same - cb["residual"]) / abs(cb["residual"])
                cb["gpu_vs_oracle_within_1e-10"] = bool(cb["gpu_vs_oracle_relative_difference"] <= 1e-10)
                rc_ = cb.get("residual_with_compensated_dots")
                if rc_:
                    # the reference's serial sums of 1.3e8 terms carry a rounding error of their own (order 1e-9 at 512^3); the device's tree sums do not
                    cb["reference_order_rounding"] = abs(cb["residual"] - rc_) / abs(rc_)
                    cb["gpu_vs_compensated_oracle_relative_difference"] = abs(gpu_same - rc_) / abs(rc_)
                    cb["gpu_within_reference_rounding"] = bool(cb["gpu_vs_oracle_relative_difference"] <= max(1e-10, 2.0 * cb["reference_order_rounding"]))
            out["cpu_baseline"] = cb

    if rank != 0:
        out = {}
    # ---- N > 1, the driver's command (--solver cg): config 4, the communicator's prices and the halo schedules -- untimed extras behind a
    # guard that never costs the line
    if world > 1 and a.solver == "cg" and a.compression == 0 and not a.no_extras:
        guard = _ExtrasGuard(rank, out)
        stage = guard.stage

        broken = None
        try:
            stage("comm_probe")
            pr = comm_probe_extra(L, dist, cg.comm, n)
            if rank == 0:
                out["comm_probe"] = pr
            stage("cg schedules")
            sc = cg_schedules_extra(L, dist, cg, SCHEDULES)
            if rank == 0:
                out["schedules"] = sc
            mgx = mgcg_multirank_extra(L, _lib, dist, a, rank, world, local_rank, cg.comm, n, stage)
            k_short, k_long = mgx.pop("_k")
            build = mgx.pop("_build")
            if rank == 0:
                out["mgcg"] = mgx
            # last: the schedule with RCCL calls on two streams of the communicator
            stage("cg schedule: exchange on the side stream")
            last = cg_schedules_extra(L, dist, cg, (SCHEDULE_LAST,), with_default=False)
            if rank == 0:
                out["schedules"].update(last)
            stage("mgcg schedule: exchange on the side stream")
            lastm = _mgcg_schedule(L, dist, build, rank, world, cg.comm, SCHEDULE_LAST, k_short, k_long)
            if rank == 0:
                out["mgcg"]["schedules"][SCHEDULE_LAST[0]] = lastm
        except Exception as ex:     # noqa: BLE001 -- this rank is now out of step with its peers' collectives: no further extras, no final barrier
            broken = f"{guard.name}: {ex}"
            L.MgcgClearLastError()
        guard.done()
        if broken is not None:
            # An error the library REPORTED in an extra: the line is printed as far as it got -- the timed result is in it, and an untimed extra
            # must never cost the scaling record that result belongs to, so the status stays 0 -- but the line says so in two keys nobody can
            # miss: `extras_aborted` (the stage and the library's message) and `extras_failure` ("error in a BASELINE configuration" for
            # config 4, its probes and schedules; "optional schedule" for the two opt-in side-stream schedules at the very end).  MGCG_BENCH_STRICT=1
            # turns an error in a BASELINE configuration into status 5 (for a caller that would rather lose the line than miss the failure).
            optional = "exchange on the side stream" in broken
            print(f"bench.py: rank {rank}: EXTRAS FAILED at {broken}", file=sys.stderr, flush=True)
            if rank == 0:
                print(json.dumps(dict(out, extras_aborted=broken, extras_failure="optional schedule" if optional else "error in a BASELINE configuration")), flush=True)
            os._exit(5 if (not optional and os.environ.get("MGCG_BENCH_STRICT") == "1") else 0)
    cg.Dispose()
    if rank == 0 and world == 1 and a.solver == "cg" and a.compression == 0 and not a.no_extras:
        try:
            out["mgcg"] = mgcg_extra(L, n)
        except Exception as ex:     # noqa: BLE001 -- secondary figure
            out["mgcg"] = {"error": str(ex)}
            L.MgcgClearLastError()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
