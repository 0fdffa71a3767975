"""BASELINE.json sizes (7-point 256^3 and one 512^3 SpMV) through size-independent properties: the oracle
cannot finish these in seconds, so the checks are a closed-form product (A.1 counts the missing neighbours),
linearity, symmetry, and the recurrence residual against a recomputed b - A x."""
import ctypes as C
import os

import numpy as np
import pytest

from conjugategradient_amd import _lib
from conjugategradient_amd.parallel import ConjugateGradientRankGpu
from conjugategradient_amd.solver import VectorDouble

pytestmark = pytest.mark.gpu


def _missing_neighbours(n):
    i = np.arange(n)
    edge = ((i == 0) | (i == n - 1)).astype(np.float64)
    return (edge[None, None, :] + edge[None, :, None] + edge[:, None, None]).ravel()


def _spmv(cg, xv, yv, N, nnz):
    L = _lib.lib()
    L.CsrMV(cg.cusparse, cg.matDescr, yv.ToRawPtr(), cg.vectorElements.ToRawPtr(), cg.vectorRowOffsets.ToRawPtr(),
            cg.vectorColumnIndeces.ToRawPtr(), xv.ToRawPtr(), nnz, N, N, 1.0, 0.0)
    _lib.check("CsrMV")


@pytest.mark.parametrize("n", [256, 512])
def test_spmv_closed_form(n):
    """A.1 = 6 - (number of neighbours) = number of grid faces the cell touches: exact small integers."""
    N = n**3
    cg = ConjugateGradientRankGpu(N, 7, 0, 10, 1e-8, rank=0, world=1)
    cg.InitializePoisson(n, n, n)
    nnz = cg.part.elementCount
    assert nnz == 7 * N - 6 * n * n
    ones, y = VectorDouble(N), VectorDouble(N)
    _lib.lib().MgcgFill(ones.Ptr, 1.0)
    for period in (0, n * n):                       # plain and banded schedules give the same bits
        _lib.lib().MgcgSetSpmvPeriod(cg.cusparse, period)
        _spmv(cg, ones, y, N, nnz)
        got = y.to_numpy()
        assert np.array_equal(got, _missing_neighbours(n))
    cg.Dispose()
    ones.Dispose()
    y.Dispose()


def test_config2_linearity_symmetry_and_residual():
    n = 256
    N = n**3
    L = _lib.lib()
    cg = ConjugateGradientRankGpu(N, 7, 0, 100000, 1e-8, rank=0, world=1)
    cg.InitializePoisson(n, n, n)
    nnz = cg.part.elementCount
    rng = np.random.default_rng(0)
    u, v = rng.standard_normal(N), rng.standard_normal(N)
    du, dv, dw, y1, y2, y3 = (VectorDouble(N) for _ in range(6))
    du.CopyFrom(u, N)
    dv.CopyFrom(v, N)
    dw.CopyFrom(0.75 * u - 1.5 * v, N)
    _spmv(cg, du, y1, N, nnz)
    _spmv(cg, dv, y2, N, nnz)
    _spmv(cg, dw, y3, N, nnz)
    Au, Av, Aw = y1.to_numpy(), y2.to_numpy(), y3.to_numpy()
    assert np.abs(Aw - (0.75 * Au - 1.5 * Av)).max() <= 1e-13 * np.abs(Aw).max()           # linearity
    uAv = L.Dot(cg.cublas, du.ToRawPtr(), y2.ToRawPtr(), N)
    vAu = L.Dot(cg.cublas, dv.ToRawPtr(), y1.ToRawPtr(), N)
    assert abs(uAv - vAu) <= 1e-12 * abs(uAv)                                               # symmetry
    assert L.Dot(cg.cublas, du.ToRawPtr(), y1.ToRawPtr(), N) > 0                            # positive definite
    # 60 CG iterations: the recurrence residual the solver reports equals ||b - A x|| recomputed from x
    res = cg.Steps(60, restart=True)
    x = VectorDouble(N)
    L.Copy(cg.cublas, x.ToRawPtr(), cg.vectorX.ToRawPtr(), N, 0, 0)
    _spmv(cg, x, y1, N, nnz)
    r = 1.0 - y1.to_numpy()
    true_res = float(np.sqrt(np.dot(r, r)))
    assert abs(true_res - res) <= 1e-9 * res
    # (the 2-norm of the CG residual is not monotone -- with b = 1 it first grows -- so no ordering is asserted)
    for t in (du, dv, dw, y1, y2, y3, x):
        t.Dispose()
    cg.Dispose()


def test_config5_like_irregular_rows(oracle):
    """BASELINE.json config 5 in miniature: random SPD, ~30 nnz/row, irregular rows, unsorted columns -- every SpMV
    kernel against the oracle, then the whole solve."""
    import conjugategradient_amd.problems as problems
    from conjugategradient_amd.solver import ConjugateGradientSingleGpu
    from tests.gpu_util import DeviceCsr, Handles

    s = problems.random_spd(200000, mean_upper=14.0, seed=12345)
    assert 25 < s.nnz / s.Count < 35
    x = np.cos(np.arange(s.Count) * 0.01)
    ref = oracle.spmv(s.Elements, s.ColumnIndeces, s.RowOffsets, x)
    h = Handles()
    A = DeviceCsr(s)
    assert np.array_equal(A.spmv(h, x, kernel=1), ref)                      # row-block kernel: multi-pass rows, bit-exact
    for k in (0, 5, 6, 7):
        np.testing.assert_allclose(A.spmv(h, x, kernel=k), ref, rtol=1e-13, atol=1e-13 * np.abs(ref).max())
    h.close()
    r = oracle.cg(s, rule=oracle.RULE_CSHARP, max_iteration=1000, trace=True)
    cg = ConjugateGradientSingleGpu(s.Count, int(np.diff(s.RowOffsets).max()), 0, 1000, 1e-8, rule=_lib.RULE_CSHARP).load(s)
    cg.Initialize()
    cg.Solve()
    cg.Read()
    assert cg.Iteration == r["iteration"]
    assert np.abs(cg.x - r["x"]).max() <= 1e-10 * np.abs(r["x"]).max()
    np.testing.assert_allclose(cg.x, np.ones(s.Count), rtol=1e-7)             # b = A.1


def test_config3_first_iterations_against_the_oracle_at_full_size(oracle, mgcg_env):
    """BASELINE configs 3 and 4 AT their size against the ORACLE (oracle/mg_oracle.c: hierarchy, V(1,1) Jacobi cycle and PCG shell restated
    on the CPU): the first two MGCG iterations on the 7-point 512^3 system, plain CSR on every level.  They exercise every kernel of the
    cycle on every level at full size (Galerkin set-up, folded first sweep and residual, restriction, coarse sweeps, prolongation, last
    sweep fused with r.z, the PCG updates); the whole 157-iteration solve is beyond a CPU loop that takes ten seconds per V-cycle here.

    The V-cycle and every vector update are bit-identical to the oracle's, so the loops differ only in the ORDER of their dot-product sums --
    and at 1.3e8 terms the reference's serial left-to-right sums (LongVector.cs:15-31) carry a rounding error of their own of order 1e-9
    (a running sum 1e8 times the addend, rounded the same way for long stretches), where the device's tree sums are good to 1e-15.  So:
      (1) default mode against the oracle with the SAME products summed exactly (compensated_dots): trace and iterate to 1e-10;
      (2) validation mode dot_order = 1 (the sums in the reference's order) against the reference-order oracle: EQUAL, bit for bit --
          residual trace and all 134 217 728 entries of x;
      (3) config 4 as stated -- the same system row-partitioned over EIGHT ranks (loopback ranks on the one card), dot_order = 1 -- against
          the oracle with its sums cut at the ranks' rows and added in rank order (oracle_pcg_parts): equal again;
      and the evidence that (1) needed its yardstick: the reference order's own distance from the exact sums is above 1e-10 here.
    About three minutes of host time."""
    import conjugategradient_amd.problems as problems
    from conjugategradient_amd.multigrid import ConjugateGradientMgGpu
    from conjugategradient_amd.parallel import ConjugateGradientMgRankGpu
    from tests.test_gpu_dot_order import assert_equal_bits
    from tests.test_gpu_parallel import _run_ranks_in_threads

    n, its, world = 512, 2, 8
    N = n**3
    L = _lib.lib()
    e, c, r = oracle.poisson_csr(n, n, n)
    s = problems.LinearSystem(e, c, r, np.zeros(N), np.ones(N), f"poisson{n}", grid=(n, n, n))
    M = oracle.Multigrid(s, levels=3)
    kw = dict(rule=oracle.RULE_NATIVE, allowable_residual=1e300, min_iteration=its - 1, max_iteration=its + 2, trace=True)
    ref = M.pcg(**kw)
    ref_parts = M.pcg(offsets=oracle.partition(N, world), **kw)
    with oracle.compensated_dots():
        exact = M.pcg(**kw)
    assert ref["iteration"] == exact["iteration"] == ref_parts["iteration"] == its - 1 and len(ref["trace"]) == len(exact["trace"]) == its
    del M, s, e, c, r

    def solve_single():
        mg = ConjugateGradientMgGpu(N, 7, its - 1, 1000, 1e300, (n, n, n), levels=3, rule=_lib.RULE_NATIVE)
        L.MgcgSetMatrixCompression(mg.cusparse, 0)
        mg.InitializePoisson()
        mg.Solve(trace=True)
        assert mg.Iteration == its - 1
        folds = L.MgcgLastVcycleFolds()
        x = np.empty(N)
        mg.vectorX.CopyTo(x, N)
        mg.Dispose()
        return mg.trace, x, folds

    # (1) default mode: the same algorithm with exactly summed dot products
    mgcg_env.setenv("MGCG_DOT_ORDER", "0")             # (also when the whole suite runs under MGCG_DOT_ORDER=1)
    trace, x, folds = solve_single()
    if "MGCG_NO_FOLD" not in os.environ and os.environ.get("MGCG_FOLD_UP", "-1") != "0":
        assert folds == 3                                  # the schedule of record: first sweep and (on the 256^3 level) prolongation folded
    np.testing.assert_allclose(trace, exact["trace"], rtol=1e-10)
    scale = np.abs(exact["x"]).max()
    assert np.abs(x - exact["x"]).max() <= 1e-10 * scale
    own_trace = np.abs(ref["trace"] - exact["trace"]) / exact["trace"]
    assert own_trace.max() > 1e-10, own_trace               # the reference order's own rounding at this size: why (1) compares with exact sums
    del exact
    # (2) the reference's summation order: equality
    mgcg_env.setenv("MGCG_DOT_ORDER", "1")
    trace, x, _ = solve_single()
    assert_equal_bits(trace, ref["trace"], "config 3 trace")
    assert_equal_bits(x, ref["x"], "config 3 x")
    del ref
    # (3) config 4: eight z-slabs of 64 planes
    mgcg_env.setenv("MGCG_VIRTUAL_DEVICES", str(world))
    mgcg_env.setenv("MGCG_OVERLAP", "2")

    def make_rank(rank, comm):
        cg = ConjugateGradientMgRankGpu(N, 7, its - 1, 1000, 1e300, (n, n, n), rank=rank, world=world, comm=comm, device=rank, levels=3, rule=_lib.RULE_NATIVE)
        cg.InitializePoisson(n, n, n)
        cg.Setup()
        cg.Solve(trace=True)
        xs = np.empty(cg.part.count)
        cg.vectorX.CopyTo(xs, cg.part.count, 0)
        out = (cg.part.offset, cg.part.count, xs, cg.Iteration, cg.trace)
        cg.Dispose()
        return out

    res = _run_ranks_in_threads(world, make_rank)
    for off, cnt, xs, it, tr in res:
        assert it == its - 1 and cnt == N // world
        assert_equal_bits(tr, ref_parts["trace"], "config 4 trace")
        assert_equal_bits(xs, ref_parts["x"][off: off + cnt], "config 4 x")


@pytest.mark.parametrize("compression", [0, 1])
def test_config3_mgcg_at_full_size(compression):
    """BASELINE config 3 (7-point 512^3, 3-level V(1,1) Jacobi) through size-independent properties: the iteration count
    is the grid-independent one (157 here; 53 at 128^3, 157 +- a few at 512^3 because the coarsest grid is only smoothed) and
    the residual the solver reports equals ||b - A x|| recomputed from x.  compression 0: the form of record -- plain CSR on
    every level, what bench.py times; 1: the opt-in lossless analysis (row patterns, class 3)."""
    from conjugategradient_amd.multigrid import ConjugateGradientMgGpu

    n = 512
    N = n**3
    L = _lib.lib()
    tol = 1e-8 * np.sqrt(N)                      # 1e-8 * ||b||_2 for b = 1
    mg = ConjugateGradientMgGpu(N, 7, 0, 1000, tol, (n, n, n), levels=3, rule=_lib.RULE_CSHARP)
    L.MgcgSetMatrixCompression(mg.cusparse, compression)
    mg.InitializePoisson()
    mg.Solve()
    assert 150 <= mg.Iteration + 1 <= 165, mg.Iteration
    assert mg.Residual < tol
    assert L.MgcgAnalysisInfo(mg.cusparse, 0, None, None, None, None) == (3 if compression else -1)
    y = VectorDouble(N)
    nnz = 7 * N - 6 * n * n
    L.CsrMV(mg.cusparse, mg.matDescr, y.ToRawPtr(), mg.vectorA.ToRawPtr(), mg.vectorRowOffsets.ToRawPtr(), mg.vectorColumnIndeces.ToRawPtr(),
            mg.vectorX.ToRawPtr(), nnz, N, N, 1.0, 0.0)
    _lib.check("CsrMV")
    # r = 1 - A x on the device: y = -1 * y + 1  (Scal, then add the constant through Axpy with a vector of ones)
    ones = VectorDouble(N)
    L.MgcgFill(ones.Ptr, 1.0)
    L.Scal(mg.cublas, y.ToRawPtr(), -1.0, N)
    L.Axpy(mg.cublas, y.ToRawPtr(), ones.ToRawPtr(), N, 1.0)
    true_res = float(np.sqrt(L.Dot(mg.cublas, y.ToRawPtr(), y.ToRawPtr(), N)))
    # at a reduction of 1e-8 the recurrence residual and b - A x differ by the round-off of 157 updates of x
    # (about eps * ||A|| * ||x|| ~ 1e-6 * ||r|| here): they must agree to a few per cent, not to the last digits
    assert abs(true_res - mg.Residual) <= 0.05 * mg.Residual, (true_res, mg.Residual)
    for v in (y, ones):
        v.Dispose()
    mg.Dispose()


@pytest.mark.parametrize("world,nz", [(3, 96), (8, 512)])
def test_config4_rank_slabs_at_full_size(mgcg_env, world, nz):
    """BASELINE config 4 on ONE GPU.  (3, 96): its per-rank problem, 512 x 512 z-slabs of 32 planes on three loopback ranks (the
    middle one has a halo plane on BOTH sides, as six of the eight ranks of the 8-GPU run have).  (8, 512): the configuration AS
    STATED -- 7-point Poisson 512^3 row-partitioned over EIGHT ranks (slabs of 64 planes = 16.8 M rows each), every rank a host
    thread on a virtual device of the one card, joined by the loopback transport where the 8-GPU node has RCCL.  3-level V(1,1) MGCG.
    Properties (no oracle can run 25 M rows in seconds): the partitioned solve takes exactly as many iterations as the
    single-domain solve of the same grid and agrees with it, the preconditioner is independent of the partition bit for bit,
    and the reported recurrence residual equals ||b - A x|| recomputed from the gathered x."""
    from conjugategradient_amd.multigrid import ConjugateGradientMgGpu
    from conjugategradient_amd.parallel import ConjugateGradientMgRankGpu
    from tests.test_gpu_parallel import _run_ranks_in_threads

    nx = 512
    dims = (nx, nx, nz)
    N = nx * nx * nz
    L = _lib.lib()
    tol = 1e-8 * np.sqrt(N)
    rng = np.random.default_rng(4)
    rvec = rng.standard_normal(N)
    # single domain
    mg = ConjugateGradientMgGpu(N, 7, 0, 1000, tol, dims, levels=3, rule=_lib.RULE_CSHARP)
    mg.InitializePoisson()
    z1 = mg.Apply(rvec)
    mg.Solve()
    it1, res1 = mg.Iteration, mg.Residual
    x1 = np.empty(N)
    mg.vectorX.CopyTo(x1, N, 0)
    mg.Dispose()
    assert res1 < tol and 20 < it1 < 400

    mgcg_env.setenv("MGCG_VIRTUAL_DEVICES", str(world))
    mgcg_env.setenv("MGCG_OVERLAP", "2")

    def make_rank(rank, comm):
        cg = ConjugateGradientMgRankGpu(N, 7, 0, 1000, tol, dims, rank=rank, world=world, comm=comm, device=rank, levels=3, rule=_lib.RULE_CSHARP)
        cg.InitializePoisson(*dims)
        cg.Setup()
        off, cnt = cg.part.offset, cg.part.count
        assert cnt == nx * nx * (nz // world)
        z = cg.Apply(rvec[off: off + cnt])
        cg.Solve()
        xs = np.empty(cnt)
        cg.vectorX.CopyTo(xs, cnt, 0)
        out = (off, cnt, z, xs, cg.Iteration, cg.Residual)
        cg.Dispose()
        return out

    res = _run_ranks_in_threads(world, make_rank)
    x, z = np.empty(N), np.empty(N)
    for off, cnt, zs, xs, it, resid in res:
        z[off: off + cnt] = zs
        x[off: off + cnt] = xs
        assert it == it1
        assert resid == res[0][5]                             # every rank holds the same all-reduced bits
    assert np.array_equal(z, z1)                              # M^-1 does not depend on the partition
    assert np.abs(x - x1).max() <= 1e-10 * np.abs(x1).max()
    # ||b - A x|| recomputed with the closed-form stencil (b = 1)
    X = x.reshape(nz, nx, nx)
    AX = 6.0 * X
    AX[1:] -= X[:-1]; AX[:-1] -= X[1:]
    AX[:, 1:] -= X[:, :-1]; AX[:, :-1] -= X[:, 1:]
    AX[:, :, 1:] -= X[:, :, :-1]; AX[:, :, :-1] -= X[:, :, 1:]
    true_res = float(np.sqrt(np.sum((1.0 - AX) ** 2)))
    assert abs(true_res - res[0][5]) <= 0.05 * res[0][5], (true_res, res[0][5])


def _auto_tiles_on():
    """The library's own column tiles can be switched off from the environment (MGCG_AUTO_TILES=0; an explicit MGCG_TILE_SHIFT also leaves the
    choice to the caller): the assertions that the feature is IN USE follow the switch, the results are demanded either way."""
    return os.environ.get("MGCG_AUTO_TILES", "1") != "0" and "MGCG_TILE_SHIFT" not in os.environ


def test_config5_at_full_size(oracle):
    """BASELINE config 5 at 10 M rows (random SPD, ~31 nonzeros per row, rows of up to ~240): the automatic kernel against
    the oracle's product (the C oracle multiplies 310 M nonzeros in about a second), the row sums (A.1 = 1 by construction),
    and the whole solve against a manufactured solution with the residual recomputed."""
    import conjugategradient_amd.problems as problems
    from conjugategradient_amd.solver import ConjugateGradientSingleGpu, VectorInt

    s = problems.random_spd(10_000_000, mean_upper=14.0, seed=12345)
    N, nnz = s.Count, s.nnz
    assert 29 < nnz / N < 33
    L = _lib.lib()
    xs = np.cos(np.arange(N) * 0.01)
    ref = oracle.spmv(s.Elements, s.ColumnIndeces, s.RowOffsets, xs)
    cg = ConjugateGradientSingleGpu(N, int(np.diff(s.RowOffsets).max()), 0, 1000, 1e-8, rule=_lib.RULE_CSHARP)
    cg.A = type("M", (), {})()
    cg.A.Elements, cg.A.ColumnIndeces, cg.A.RowOffsets = s.Elements, s.ColumnIndeces, s.RowOffsets
    cg.vectorA.Dispose(); cg.vectorColumnIndeces.Dispose()
    cg.vectorA, cg.vectorColumnIndeces = VectorDouble(nnz), VectorInt(nnz)
    cg.x[:] = 0.0
    cg.b[:] = ref + 2.0                                      # A.(xs + 2) = A.xs + 2 A.1 = ref + 2
    cg.Initialize()
    dx, dy = VectorDouble(N), VectorDouble(N)
    args = lambda xin: (cg.cusparse, cg.matDescr, dy.ToRawPtr(), cg.vectorA.ToRawPtr(), cg.vectorRowOffsets.ToRawPtr(), cg.vectorColumnIndeces.ToRawPtr(), xin.ToRawPtr(), nnz, N, N, 1.0, 0.0)
    dx.CopyFrom(xs, N)
    L.CsrMV(*args(dx))
    _lib.check("CsrMV")
    np.testing.assert_allclose(dy.to_numpy(), ref, rtol=1e-13, atol=1e-13 * np.abs(ref).max())
    L.MgcgFill(dx.Ptr, 1.0)
    L.CsrMV(*args(dx))
    assert np.abs(dy.to_numpy() - 1.0).max() <= 1e-12       # row sums: 1 + sum|off| - sum|off|
    # the solve: with compression OFF the library builds the column tiles of this matrix itself (its sampled entries lie 3 M columns from
    # the diagonal on average, x is 80 MB) -- class 4 with 12-byte entries, re-verified by checksum at every solve
    L.MgcgSetMatrixCompression(cg.cusparse, 0)              # (also when MGCG_COMPRESSION pre-set a mode for every new handle)
    L.MgcgAnalysisClear(cg.cusparse)
    assert L.MgcgAnalysisInfo(cg.cusparse, 0, None, None, None, None) == -1
    cg.Solve()
    cg.Read()
    assert L.MgcgAnalysisInfo(cg.cusparse, 0, None, None, None, None) == (4 if _auto_tiles_on() else -1)
    assert 20 < cg.Iteration < 1000 and cg.Residual < 1e-8
    assert np.abs(cg.x - (xs + 2.0)).max() <= 1e-7
    it_tiles, res_tiles, x_tiles = cg.Iteration, cg.Residual, cg.x.copy()
    # the same solve on the CSR kernels (auto_tiles off; 16 lanes per row at this row length, whose sums differ from the stored-order
    # sums of the tiles in the last bits): same iteration count, same answer to round-off
    assert L.MgcgSetTuning(b"auto_tiles", 0) == 0
    cg.x[:] = 0.0
    cg.vectorX.CopyFrom(cg.x, N)
    cg.Solve()
    cg.Read()
    L.MgcgReloadEnvironment()                               # (back to what the environment says, MGCG_AUTO_TILES included)
    assert abs(cg.Iteration - it_tiles) <= 1 and np.abs(cg.x - x_tiles).max() <= 1e-8
    # a second solve finds the cached form (checksum verified); after the matrix values changed it is rebuilt, not reused:
    # A -> 2 A through Scal on the raw pointer of the values (the solution halves)
    L.Scal(cg.cublas, cg.vectorA.ToRawPtr(), 2.0, nnz)
    cg.x[:] = 0.0
    cg.vectorX.CopyFrom(cg.x, N)
    cg.Solve()
    cg.Read()
    assert np.abs(cg.x - 0.5 * (xs + 2.0)).max() <= 1e-7
    L.Scal(cg.cublas, cg.vectorA.ToRawPtr(), 0.5, nnz)
    cg.x[:] = x_tiles
    cg.vectorX.CopyFrom(cg.x, N)
    L.Copy(cg.cublas, dx.ToRawPtr(), cg.vectorX.ToRawPtr(), N, 0, 0)
    L.CsrMV(*args(dx))
    r = cg.b - dy.to_numpy()
    true_res = float(np.sqrt(np.dot(r, r)))
    assert abs(true_res - cg.Residual) <= 0.05 * cg.Residual + 1e-10, (true_res, cg.Residual)
    # The per-op API (what the reference's own phase driver multiplies through: CsrMV Mgcg.cu:10-19, Solve1 :145-163) gets the tiles too, under
    # the write registry instead of a checksum per call: compression off, library-owned vectors -- the first products of a matrix run on the
    # CSR kernels (sums in tree order: 1e-13), from the 8th on the library's column tiles serve them (stored order: bit-identical to the oracle);
    # a write through any export (Scal on the values) sends the next products back to the CSR kernels until the form is rebuilt.
    L.MgcgAnalysisClear(cg.cusparse)
    dx.CopyFrom(xs, N)
    if not _auto_tiles_on():
        for v in (dx, dy):
            v.Dispose()
        cg.Dispose()
        return

    def automatic_form():
        i = 0
        while True:
            c = L.MgcgAnalysisInfo(cg.cusparse, i, None, None, None, None)
            if c < 0:
                return None
            if c == 4:
                return i
            i += 1

    assert automatic_form() is None
    built_at = None
    for k in range(10):                                                    # (the handle has multiplied this matrix a few times already: the 8th product overall builds)
        L.CsrMV(*args(dx))
        got = dy.to_numpy()
        if automatic_form() is None:
            assert built_at is None and k < 8
            np.testing.assert_allclose(got, ref, rtol=1e-13, atol=1e-13 * np.abs(ref).max())
        else:
            built_at = k if built_at is None else built_at
            assert np.array_equal(got, ref), k
    assert built_at is not None and built_at >= 1
    pap = L.Solve1(cg.cublas, cg.cusparse, cg.matDescr, cg.vectorA.Ptr, cg.vectorRowOffsets.Ptr, cg.vectorColumnIndeces.Ptr, dy.Ptr, dx.Ptr, N, N, 0, nnz)
    _lib.check("Solve1")
    assert np.array_equal(dy.to_numpy(), ref) and abs(pap - float(np.dot(xs, ref))) <= 1e-12 * abs(pap)      # Ap = A p on the tiles ; p.Ap
    L.Scal(cg.cublas, cg.vectorA.ToRawPtr(), 2.0, nnz)                     # a write the library sees: the form is stale at once
    L.CsrMV(*args(dx))
    np.testing.assert_allclose(dy.to_numpy(), 2.0 * ref, rtol=1e-13, atol=1e-13 * np.abs(ref).max())
    L.Scal(cg.cublas, cg.vectorA.ToRawPtr(), 0.5, nnz)
    for k in range(16):                                                    # (the threshold doubles with every rebuild: 16 products this time)
        L.CsrMV(*args(dx))
    assert np.array_equal(dy.to_numpy(), ref)
    # a write BEHIND the library's back -- hipMemcpy on the raw pointer of the values, what a caller's own kernel would do through
    # ToRawPtr_Double: no export sees it, but every 16th product served from the form re-verifies the checksum of the CSR arrays
    # (runtime.hip: dcsr_lookup_op), so within 16 products the stale form is dropped and the products are those of the matrix in memory
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    e0 = C.c_double(3.0 * float(s.Elements[0]))
    assert hip.hipMemcpy(cg.vectorA.ToRawPtr(), C.byref(e0), 8, 1) == 0        # elements[0] *= 3 (host to device)
    want0 = ref[0] + 2.0 * float(s.Elements[0]) * xs[int(s.ColumnIndeces[0])]
    seen = None
    for k in range(17):
        L.CsrMV(*args(dx))
        if abs(dy.to_numpy(1)[0] - want0) <= 1e-12 * abs(want0):
            seen = k
            break
    assert seen is not None and seen <= 16, seen
    for v in (dx, dy):
        v.Dispose()
    cg.Dispose()


@pytest.mark.parametrize("balance", ["rows", "nnz"])
def test_config5_eight_rank_leg_at_full_size(oracle, mgcg_env, balance):
    """BASELINE config 5's 8-GPU leg at true size on ONE GPU: the 10 M-row random SPD matrix row-partitioned over eight loopback
    ranks (nearly every x entry of the other seven ranks is in a rank's halo, since the columns are uniform), plain CG against a
    manufactured solution.  balance="rows": the reference's partition, 1.25 M rows each -- 21 M to 78 M nonzeros per rank, because this
    generator's rows grow with the row number; "nnz": row ranges of 38.7 M nonzeros each (problems.partition_offsets).  Checked: the same iteration count (+-1) and answer as the single-domain solve, identical
    all-reduced residual bits on every rank, ||b - A x|| recomputed by the oracle's product from the gathered x, and that each
    rank's slab (1.25 M x 10 M) took the column tiles the library builds for matrices without locality."""
    import conjugategradient_amd.problems as problems
    from conjugategradient_amd.solver import ConjugateGradientSingleGpu, VectorInt
    from tests.test_gpu_parallel import _run_ranks_in_threads

    world = 8
    s = problems.random_spd(10_000_000, mean_upper=14.0, seed=12345)
    N, nnz = s.Count, s.nnz
    maxnz = int(np.diff(s.RowOffsets).max())
    L = _lib.lib()
    xs = np.cos(np.arange(N) * 0.01)
    ref = oracle.spmv(s.Elements, s.ColumnIndeces, s.RowOffsets, xs)
    s.b[:] = ref + 2.0                                       # A.(xs + 2) = ref + 2 (row sums are 1)
    s.x[:] = 0.0
    # single domain
    cg = ConjugateGradientSingleGpu(N, maxnz, 0, 1000, 1e-8, rule=_lib.RULE_CSHARP)
    cg.A = type("M", (), {})()
    cg.A.Elements, cg.A.ColumnIndeces, cg.A.RowOffsets = s.Elements, s.ColumnIndeces, s.RowOffsets
    cg.vectorA.Dispose(); cg.vectorColumnIndeces.Dispose()
    cg.vectorA, cg.vectorColumnIndeces = VectorDouble(nnz), VectorInt(nnz)
    cg.x[:] = 0.0
    cg.b[:] = s.b
    cg.Initialize()
    cg.Solve()
    cg.Read()
    it1, res1, x1 = cg.Iteration, cg.Residual, cg.x.copy()
    cg.Dispose()
    assert 20 < it1 < 1000 and res1 < 1e-8

    mgcg_env.setenv("MGCG_VIRTUAL_DEVICES", str(world))

    def make_rank(rank, comm):
        rk = ConjugateGradientRankGpu(N, maxnz, 0, 1000, 1e-8, rank=rank, world=world, comm=comm, device=rank, balance=balance).load(s)
        rk.Initialize()
        if balance == "rows":
            assert rk.part.count == N // world
        else:
            assert abs(rk.part.elementCount - nnz / world) <= maxnz
        L.MgcgSetMatrixCompression(rk.cusparse, 0)
        rk.Solve()
        form = L.MgcgAnalysisInfo(rk.cusparse, 0, None, None, None, None)
        rk.Read()
        off, cnt = rk.part.offset, rk.part.count
        out = (off, cnt, rk.x[off: off + cnt].copy(), rk.Iteration, rk.Residual, form, rk.part.elementCount / cnt)
        rk.Dispose()
        return out

    res = _run_ranks_in_threads(world, make_rank)
    x = np.empty(N)
    for off, cnt, xr, it, resid, form, per_row in res:
        x[off: off + cnt] = xr
        assert abs(it - it1) <= 1, (it, it1)
        assert it == res[0][3] and resid == res[0][4]        # one stop decision, the same all-reduced bits everywhere
        # the slab took the column tiles (12-byte entries) -- up to 64 nonzeros per row; beyond (the last slab of the equal-nonzero
        # partition has 76) 32 lanes per row are as fast and nothing is built
        assert form == (4 if (per_row <= 64 and _auto_tiles_on()) else -1), (form, per_row)
    assert sum(1 for r in res if r[5] == 4) >= (world - 1 if _auto_tiles_on() else 0)
    assert res[0][4] < 1e-8
    assert np.abs(x - (xs + 2.0)).max() <= 1e-7
    assert np.abs(x - x1).max() <= 1e-8
    r = s.b - oracle.spmv(s.Elements, s.ColumnIndeces, s.RowOffsets, x)
    true_res = float(np.sqrt(np.dot(r, r)))
    assert abs(true_res - res[0][4]) <= 0.05 * res[0][4] + 1e-10, (true_res, res[0][4])
