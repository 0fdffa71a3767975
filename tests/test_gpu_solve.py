"""The device-resident CG loop (Solve / SolveEx / phase functions) vs the CPU oracle and the golden fixtures.

Tolerances (north star: residual norm and iterate-wise within 1e-10 relative): x, r, p updates and the
SpMV are bit-identical to the oracle; only the dot products are summed in a different order, which
perturbs alpha/beta at the 1e-16 level.  While the residual is above round-off the per-iteration
residual trace must agree to rtol 1e-10 and x to 1e-10 * max|x|.
"""
import ctypes as C
import math

import numpy as np
import pytest

from conjugategradient_amd import _lib, problems
from conjugategradient_amd.solver import (ApplicationException, ConjugateGradientParallelGpu,
                                          ConjugateGradientSingleGpu)
from tests.conftest import golden
from tests.gpu_util import assert_iterate_close, assert_trace_close

pytestmark = pytest.mark.gpu

RTOL_TRACE = 1e-10
RTOL_X = 1e-10


def _solve_single(system, min_it, max_it, tol, rule=None, trace=False, max_nz=None):
    max_nz = int(np.diff(system.RowOffsets).max()) if max_nz is None else max_nz
    cg = ConjugateGradientSingleGpu(system.Count, max_nz, min_it, max_it, tol, rule=rule).load(system)
    cg.Initialize()
    cg.Solve(trace=trace)
    cg.Read()
    return cg


@pytest.mark.parametrize("name,builder,orule,grule,max_it", [
    ("ka1_tridiagonal10", lambda: problems.tridiagonal(10), "RULE_SIMPLE", _lib.RULE_SIMPLE, 10),
    ("ka2_rcg21", lambda: problems.mgcg_main(21, 6, 10.0), "RULE_NATIVE", _lib.RULE_NATIVE, 21),
    ("ka3_mgcgmain2000", lambda: problems.mgcg_main(2000, 160), "RULE_CSHARP", _lib.RULE_CSHARP, 2000),
    ("poisson5_32x32", lambda: problems.poisson(32, 32, 1), "RULE_NATIVE", _lib.RULE_NATIVE, 4096),
    ("poisson7_12x12x12", lambda: problems.poisson(12, 12, 12), "RULE_NATIVE", _lib.RULE_NATIVE, 4096),
])
def test_golden_known_answers(name, builder, orule, grule, max_it):
    g = golden(name)
    s = builder()
    cg = _solve_single(s, 0, max_it, 1e-8, rule=grule, trace=True)
    assert cg.Iteration == int(g["iteration"])
    assert_trace_close(cg.trace, g["trace"])
    assert abs(cg.Residual - float(g["residual"])) <= 1e-6 * float(g["residual"]) + 1e-12 * float(g["trace"][0])
    scale = np.abs(g["x_cg"]).max()
    assert np.abs(cg.x - g["x_cg"]).max() <= RTOL_X * scale
    assert np.abs(cg.x - g["x_direct"]).max() <= 1e-8 * max(scale, 1.0)


def test_native_solve_export_and_iteration_convention(oracle):
    """Solve (Mgcg.cu:201-270) returns the post-incremented counter; the C# wrapper subtracts 1."""
    s = problems.poisson(16, 16, 16)
    ref = oracle.cg(s, rule=oracle.RULE_NATIVE, max_iteration=5000, hard_cap=5000, trace=True)
    cg = _solve_single(s, 0, 5000, 1e-8)          # rule=None -> the reference's Solve export
    assert cg.Iteration == ref["iteration"] == 43
    assert abs(cg.Residual - ref["residual"]) <= 1e-7 * ref["residual"]
    assert np.abs(cg.x - ref["x"]).max() <= RTOL_X * np.abs(ref["x"]).max()
    # x really solves the system
    assert np.linalg.norm(s.b - s.to_scipy() @ cg.x) < 2e-8


@pytest.mark.parametrize("rule_o,rule_g", [("RULE_NATIVE", _lib.RULE_NATIVE), ("RULE_CSHARP", _lib.RULE_CSHARP),
                                           ("RULE_SIMPLE", _lib.RULE_SIMPLE), ("RULE_HANDMADECL", _lib.RULE_HANDMADECL),
                                           ("RULE_VIENNACL", _lib.RULE_VIENNACL)])
def test_rule_variants_match_oracle(oracle, rule_o, rule_g):
    s = problems.mgcg_main(1500, 160)
    s.x[:] = np.arange(s.Count) / 100.0
    tol = 1e-6
    ref = oracle.cg(s, rule=getattr(oracle, rule_o), allowable_residual=tol, min_iteration=3, max_iteration=1500, hard_cap=2000, trace=True)
    cg = _solve_single(s, 3, 1500, tol, rule=rule_g, trace=True)
    assert cg.Iteration == ref["iteration"]
    assert_trace_close(cg.trace, ref["trace"])
    assert np.abs(cg.x - ref["x"]).max() <= RTOL_X * np.abs(ref["x"]).max()


def test_min_iteration_runs_past_convergence(oracle):
    """MgcgMain's MIN_ITERATION=200 idiom (MgcgMain.cs:25): keep iterating at round-off level."""
    s = problems.mgcg_main(1200, 160)
    ref = oracle.cg(s, rule=oracle.RULE_CSHARP, min_iteration=60, max_iteration=1200, trace=True)
    cg = _solve_single(s, 60, 1200, 1e-8, rule=_lib.RULE_CSHARP, trace=True)
    assert cg.Iteration == ref["iteration"] == 60
    above = ref["trace"] > 1e-6 * ref["trace"][0]
    np.testing.assert_allclose(cg.trace[above], ref["trace"][above], rtol=RTOL_TRACE)
    assert cg.trace[-1] < 1e-8 and ref["trace"][-1] < 1e-8      # both sit at round-off level
    # 60 forced iterations end deep in the round-off tail: 1e-10, or within the oracle's own spread over device counts (asserted inside)
    assert_iterate_close(cg.x, ref["x"], spread_refs=[oracle.cg_parallel(s, w, min_iteration=60, max_iteration=1200)["x"] for w in (2, 3, 4)])


def test_max_iteration_raises_like_the_csharp_rule(oracle):
    s = problems.poisson(24, 24, 1)
    ref = oracle.cg(s, rule=oracle.RULE_CSHARP, max_iteration=5)
    assert ref["status"] == oracle.MAXIT_EXCEEDED
    cg = ConjugateGradientSingleGpu(s.Count, 5, 0, 5, 1e-8, rule=_lib.RULE_CSHARP).load(s)
    cg.Initialize()
    with pytest.raises(ApplicationException):
        cg.Solve()
    assert cg.Iteration == ref["iteration"] == 6
    # the native export does not hang either (the reference would: Mgcg.cu never reads maxIteration)
    cg2 = ConjugateGradientSingleGpu(s.Count, 5, 0, 5, 1e-8).load(s)
    cg2.Initialize()
    with pytest.raises(ApplicationException):
        cg2.Solve()


def test_nonfinite_residual_stops():
    s = problems.poisson(10, 10, 1)
    s.b[7] = np.nan
    cg = ConjugateGradientSingleGpu(s.Count, 5, 0, 100, 1e-8, rule=_lib.RULE_NATIVE).load(s)
    cg.Initialize()
    with pytest.raises(_lib.MgcgError):
        cg.Solve()
    assert cg.status == _lib.NONFINITE


def test_repeated_solve_restarts_from_x(oracle):
    """x is both initial guess and result (Mgcg.cu:225): a second Solve starts from the first's answer."""
    s = problems.poisson(14, 14, 14)
    cg = ConjugateGradientSingleGpu(s.Count, 7, 0, 5000, 1e-4, rule=_lib.RULE_NATIVE).load(s)
    cg.Initialize()
    cg.Solve()
    first = cg.Iteration
    cg.AllowableResidual = 1e-9
    cg.Solve()
    cg.Read()
    assert first > 5 and cg.Residual < 1e-9
    assert np.linalg.norm(s.b - s.to_scipy() @ cg.x) < 1e-8


@pytest.mark.parametrize("ndev", [1, 2, 3, 4])
def test_parallel_gpu_phases_with_virtual_devices(oracle, mgcg_env, ndev):
    """ConjugateGradientParallelGpu (Initialize / SyncP / Solve0-3 / Read) with the devices of this
    process; MGCG_VIRTUAL_DEVICES maps several device ids onto the one GPU of the test box."""
    mgcg_env.setenv("MGCG_VIRTUAL_DEVICES", str(ndev))
    s = problems.mgcg_main(2002, 160)
    ref = oracle.cg_parallel(s, ndev, allowable_residual=1e-8, min_iteration=0, max_iteration=2002, trace=True)
    cg = ConjugateGradientParallelGpu(s.Count, 160, 0, 2002, 1e-8).load(s)
    assert cg.deviceCount == ndev
    cg.Initialize()
    for d in range(ndev):
        lo, hi = oracle.minmax_column(s, cg.offsetsForDevice[d], cg.offsetsForDevice[d + 1])
        assert (cg.minJ[d], cg.maxJ[d]) == (lo, hi)
    cg.Solve()
    cg.Read()
    assert cg.Iteration == ref["iteration"]
    assert abs(cg.Residual - ref["residual"]) <= 1e-6 * ref["residual"]
    assert np.abs(cg.x - ref["x"]).max() <= RTOL_X * np.abs(ref["x"]).max()
    cg.Dispose()


def test_phase_functions_leave_reference_state(oracle):
    """Solve0 leaves Ap = A p, r = b - Ap, p_loc = r; Solve1 leaves Ap = A p (observable via the handles)."""
    from tests.gpu_util import Handles, dvec, ivec
    L = _lib.lib()
    h = Handles()
    s = problems.poisson(9, 9, 9)
    n = s.Count
    x0 = np.sin(np.arange(n))
    e, c, r_ = dvec(s.Elements), ivec(s.ColumnIndeces), ivec(s.RowOffsets)
    vx, vb, vAp, vp, vr = dvec(x0), dvec(s.b), dvec(np.zeros(n)), dvec(x0), dvec(np.zeros(n))
    rr = L.Solve0(h.blas, h.sparse, h.descr, e.Ptr, r_.Ptr, c.Ptr, vx.Ptr, vb.Ptr, vAp.Ptr, vp.Ptr, vr.Ptr, n, n, 0, s.nnz)
    Ax = oracle.spmv(s.Elements, s.ColumnIndeces, s.RowOffsets, x0)
    assert np.array_equal(vAp.to_numpy(), Ax)
    assert np.array_equal(vr.to_numpy(), s.b - Ax)
    assert np.array_equal(vp.to_numpy(), s.b - Ax)
    assert abs(rr - oracle.dot(s.b - Ax, s.b - Ax)) <= 1e-13 * rr
    pAp = L.Solve1(h.blas, h.sparse, h.descr, e.Ptr, r_.Ptr, c.Ptr, vAp.Ptr, vp.Ptr, n, n, 0, s.nnz)
    p = s.b - Ax
    Ap = oracle.spmv(s.Elements, s.ColumnIndeces, s.RowOffsets, p)
    assert np.array_equal(vAp.to_numpy(), Ap)
    assert abs(pAp - oracle.dot(p, Ap)) <= 1e-13 * abs(pAp)
    alpha = rr / pAp
    rr2 = L.Solve2(h.blas, alpha, vx.Ptr, vAp.Ptr, vp.Ptr, vr.Ptr, n, 0)
    assert np.array_equal(vx.to_numpy(), x0 + alpha * p)
    rnew = p + (-alpha) * Ap
    assert np.array_equal(vr.to_numpy(), rnew)
    assert abs(rr2 - oracle.dot(rnew, rnew)) <= 1e-13 * rr2
    L.Solve3(h.blas, rr2 / rr, vp.Ptr, vr.Ptr, n, 0)
    assert np.array_equal(vp.to_numpy(), rnew + (rr2 / rr) * p)
    _lib.check("phases")
    h.close()


def test_other_front_ends(oracle):
    """ViennaCL's ComputerGpu (relative rule) and HandmadeCL's max-norm solver on the same library (rows a13/a14)."""
    from conjugategradient_amd.frontends import ComputerGpu, ConjugateGradientCLGpu

    s = problems.mgcg_main(1800, 160)
    s.x[:] = 0.0
    ref = oracle.cg(s, rule=oracle.RULE_VIENNACL, allowable_residual=1e-4, min_iteration=0, max_iteration=1800, hard_cap=2000)
    gpu = ComputerGpu(s.Count)
    gpu.Write(s.Elements, s.RowOffsets.astype(np.uint32), s.ColumnIndeces.astype(np.uint32), s.x, s.b)
    gpu.Solve(1e-4, 0, 1800)                       # MgcgCL.cs: tolerance 1e-4 relative
    x = np.zeros(s.Count)
    gpu.Read(x)
    assert gpu.Iteration() == ref["iteration"] + 1
    assert np.abs(x - ref["x"]).max() <= 1e-10 * np.abs(ref["x"]).max()
    gpu.Dispose()

    ref = oracle.cg(s, rule=oracle.RULE_HANDMADECL, allowable_residual=1e-4, min_iteration=50, max_iteration=1800)   # MgcgCLMain.cs:25,35
    cl = ConjugateGradientCLGpu(s.Count, 160, 50, 1800, 1e-4).load(s)
    cl.Initialize()
    cl.Solve()
    cl.Read()
    assert cl.Iteration == ref["iteration"] == 50
    assert abs(cl.Residual - ref["residual"]) <= 1e-3 * ref["residual"] + 1e-12      # round-off level max-norm after 50 forced iterations
    # (the iterate after exactly 50 iterations does not depend on the norm of the stop rule: the 2-norm oracle over 2 .. 4 devices, stopped at the
    #  same index, gives the oracle's own spread there)
    assert_iterate_close(cl.x, ref["x"], spread_refs=[oracle.cg_parallel(s, w, allowable_residual=1e300, min_iteration=50, max_iteration=1800)["x"] for w in (2, 3, 4)])
    cl.Dispose()


def test_front_ends_driven_from_their_own_builders(oracle):
    """Row f1: the HandmadeCL ELL indexer and the ViennaCL dictionary-of-rows, filled like the reference drivers fill them,
    flattened in stored order and solved on the device; the oracle solves the flattened CSR."""
    from conjugategradient_amd.formats import CompressedMatrix
    from conjugategradient_amd.frontends import ComputerGpu, ConjugateGradientCLGpu

    n, K = 400, 160
    cl = ConjugateGradientCLGpu(n, K, 50, n, 1e-4)
    for i in range(n):                                                   # MgcgCLMain.cs:52-90
        cl.A[i, i] = 0
        for j in range(max(0, i - K // 2 + 1), min(n, i + K // 2)):
            if i != j:
                a = abs(np.sin(float(i + j)))
                cl.A[i, j] = a
                cl.A[i, i] = cl.A[i, i] + a
        cl.b[i] = np.cos(float(i)) * 10
        cl.x[i] = i / 100.0
    s = problems.mgcg_main(n, K)
    e, c, ro = cl.A.to_csr()
    assert np.array_equal(e, s.Elements) and np.array_equal(c, s.ColumnIndeces) and np.array_equal(ro, s.RowOffsets)
    ref = oracle.cg(s, rule=oracle.RULE_HANDMADECL, allowable_residual=1e-4, min_iteration=50, max_iteration=n)
    cl.Initialize()
    cl.Solve()
    cl.Read()
    assert cl.Iteration == ref["iteration"] == 50
    assert_iterate_close(cl.x, ref["x"], spread_refs=[oracle.cg_parallel(s, w, allowable_residual=1e300, min_iteration=50, max_iteration=n)["x"] for w in (2, 3, 4)])
    cl.Dispose()

    A = CompressedMatrix()
    b = np.zeros(n)
    for i in range(n):                                                   # MgcgCL.cs:31-45
        A[i, i] = i
        for j in range(max(0, i - K // 2), min(n - 1, i + K // 2) + 1):
            if i != j:
                a = abs(np.sin(float(i + j)))
                A[i, j] = a
                A[i, i] = A[i, i] + a
        b[i] = math.asin(i / n)            # (libm, as MgcgCL.cs's Math.Asin and the C++ twin; numpy's arcsin differs in the last bit for some arguments)
    v = problems.viennacl_main(n, K)
    e, c, ro = A.to_csr()
    assert np.array_equal(e, v.Elements) and np.array_equal(c, v.ColumnIndeces) and np.array_equal(b, v.b)
    ref = oracle.cg(v, rule=oracle.RULE_VIENNACL, allowable_residual=1e-4, min_iteration=0, max_iteration=n, hard_cap=n + 10)
    gpu = ComputerGpu(n)
    gpu.Write(e, ro, c, np.zeros(n), b)
    gpu.Solve(1e-4, 0, n)
    x = np.zeros(n)
    gpu.Read(x)
    assert gpu.Iteration() == ref["iteration"] + 1
    assert np.abs(x - ref["x"]).max() <= 1e-10 * np.abs(ref["x"]).max()
    gpu.Dispose()


def test_placement_draw_moves_p_once_and_changes_no_bit(mgcg_env):
    """The library's placement draw (solver.hip: placement_draw; knob `placement`): at the first solve on a p of >= 32 M entries the loop's
    own SpMV is timed on 3 more allocations of Ap, then of p, and the fastest is kept each time -- the same doubles at another address.  Checked on 328^3
    (35.3 M rows): four candidates timed, one chosen, the draw happens once per vector, results identical with the draw off, and a vector
    whose address ToRawPtr_Double has handed out is never moved."""
    from conjugategradient_amd.parallel import ConjugateGradientRankGpu

    L = _lib.lib()
    n = 328
    ms = (C.c_double * 16)()
    chosen = C.c_int(-1)

    def run(placement, export_first=False):
        mgcg_env.setenv("MGCG_PLACEMENT", str(placement))
        cg = ConjugateGradientRankGpu(n**3, 7, 0, 10**6, 1e-8, rank=0, world=1)
        cg.InitializePoisson(n, n, n)
        before = cg.vectorP.ToRawPtr() if export_first else None
        r1 = cg.Steps(6, restart=True)
        cand_ap = L.MgcgLastPlacement(0, None, 0, None)
        cand = L.MgcgLastPlacement(1, ms, 16, C.byref(chosen))
        times = [ms[i] for i in range(cand)]
        L.MgcgFill(cg.vectorX.Ptr, 0.0)
        r2 = cg.Steps(6, restart=True)                            # a second solve from the same x0 on the same vector: no second draw, same result
        again = L.MgcgLastPlacement(1, ms, 16, C.byref(chosen))
        after = cg.vectorP.ToRawPtr()
        cg.Dispose()
        assert cand_ap == (4 if placement else 0)      # (the written vector Ap is drawn first; only p's address was exported, if any)
        return r1, r2, cand, times, chosen.value, again, before, after

    r1, r2, cand, times, pick, again, _, _ = run(3)
    assert cand == 4 and 0 <= pick < 4 and all(t > 0 for t in times) and times[pick] == min(times), (times, pick)
    assert again == 4 and r1 == r2                                 # (the record of the one draw is still there; nothing was timed again)
    o1, o2, cand0, *_ = run(0)
    assert cand0 == 0 and (o1, o2) == (r1, r2)                      # the draw off: the same bits
    e1, e2, cande, _, _, _, before, after = run(3, export_first=True)
    assert cande == 0 and before == after and (e1, e2) == (r1, r2)  # an exported address stays where it is
    # Solve() draws only when the iteration cap leaves room to win the draw's price back (about a thousand iterations: solver.hip,
    # kPlacementMinIterations): a call capped at 200 iterations does not, the same call capped at 10^6 does -- same bits either way
    mgcg_env.setenv("MGCG_PLACEMENT", "3")
    out = []
    for cap in (200, 10**6):
        cg = ConjugateGradientRankGpu(n**3, 7, 3, cap, 1e300, rank=0, world=1, rule=_lib.RULE_NATIVE)
        cg.InitializePoisson(n, n, n)
        cg.Solve()
        out.append((L.MgcgLastPlacement(0, None, 0, None), cg.Iteration, cg.Residual))
        cg.Dispose()
    assert [o[0] for o in out] == [0, 4] and out[0][1:] == out[1][1:] and out[0][1] == 3
