"""Validation mode ``dot_order = 1`` (MGCG_DOT_ORDER): every dot product of the library adds its rounded products strictly left to
right (Mgcg/cuBlas/Mgcg/LongVector.cs:15-31) and the ranks' sums are added in rank order (resultsDot.Sum(),
Mgcg/cuBlas/Mgcg/ConjugateGradientParallelGpu.cs:463,499,525).  SpMV, the vector updates and the whole V-cycle are bit-identical to
the oracle already; with the sums in the reference's order too, the loops are no longer "within 1e-10" of the oracle -- they are
EQUAL to it: every entry of the residual trace, every entry of x, Iteration and Residual, for any size and any rank count.  These
tests therefore use ``np.array_equal`` / ``==`` only; a wrong halo plane, a missed update or a reordered operation shows as a bit.

The mode is a test instrument (one serial sum of 1.3e8 terms takes about half a second); nothing here is a timed path."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conjugategradient_amd import _lib, problems
from conjugategradient_amd.multigrid import ConjugateGradientMgGpu
from conjugategradient_amd.parallel import ConjugateGradientMgRankGpu, ConjugateGradientRankGpu
from conjugategradient_amd.solver import ConjugateGradientParallelGpu, ConjugateGradientSingleGpu
from tests.conftest import golden
from tests.gpu_util import Handles, dvec

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def reference_order(mgcg_env):
    mgcg_env.setenv("MGCG_DOT_ORDER", "1")
    v = C.c_int(0)
    assert _lib.lib().MgcgGetTuning(b"dot_order", C.byref(v)) == 0 and v.value == 1
    return mgcg_env


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_equal_bits(actual, desired, what=""):
    actual, desired = np.asarray(actual, dtype=np.float64), np.asarray(desired, dtype=np.float64)
    assert actual.shape == desired.shape, (what, actual.shape, desired.shape)
    same = _bits(actual) == _bits(desired)
    if not np.all(same):
        k = int(np.argmin(same))
        raise AssertionError(f"{what}: {int((~same).sum())} of {same.size} entries differ, first at {k}: {actual.ravel()[k]!r} != {desired.ravel()[k]!r}")


def test_dot_exports_add_in_the_reference_order(oracle, reference_order):
    """Dot and CsrMVDot (Mgcg.cu:30-38): the one sum, left to right -- at a length that is not a multiple of anything the kernel uses,
    on numbers whose sum depends on the order in the last bits (checked: the tree order of the default mode gives other bits)."""
    L = _lib.lib()
    h = Handles()
    rng = np.random.default_rng(7)
    for n in (1, 63, 2048, 2049, 100003, 1 << 20):
        x, y = rng.standard_normal(n) * 1e3, rng.standard_normal(n)
        vx, vy = dvec(x), dvec(y)
        got = L.Dot(h.blas, vy.ToRawPtr(), vx.ToRawPtr(), n)
        assert got == oracle.dot(x, y), (n, got, oracle.dot(x, y))
    s = problems.poisson(20, 12, 9)
    x = rng.standard_normal(s.Count)
    from tests.gpu_util import DeviceCsr
    A = DeviceCsr(s)
    vx, vy = dvec(x), dvec(np.zeros(s.Count))
    got = L.CsrMVDot(h.blas, h.sparse, vy.ToRawPtr(), A.e.ToRawPtr(), A.r.ToRawPtr(), A.c.ToRawPtr(), vx.ToRawPtr(), vx.ToRawPtr(), s.nnz, s.Count, s.Count)
    Ax = oracle.spmv(s.Elements, s.ColumnIndeces, s.RowOffsets, x)
    assert got == oracle.dot(x, Ax)
    assert_equal_bits(vy.to_numpy(s.Count), Ax, "CsrMVDot y")
    # the default mode is a different (better conditioned) order: same value to 1e-13, generally other bits
    reference_order.setenv("MGCG_DOT_ORDER", "0")
    n = 1 << 20
    x, y = rng.standard_normal(n) * 1e3, rng.standard_normal(n)
    vx, vy = dvec(x), dvec(y)                        # (named: a temporary would be freed before the kernel reads it)
    tree = L.Dot(h.blas, vy.ToRawPtr(), vx.ToRawPtr(), n)
    assert abs(tree - oracle.dot(x, y)) <= 1e-12 * np.abs(x * y).sum()
    h.close()


def _solve_single(system, min_it, max_it, tol, rule, max_nz=None):
    max_nz = int(np.diff(system.RowOffsets).max()) if max_nz is None else max_nz
    cg = ConjugateGradientSingleGpu(system.Count, max_nz, min_it, max_it, tol, rule=rule).load(system)
    cg.Initialize()
    cg.Solve(trace=True)
    cg.Read()
    return cg


@pytest.mark.parametrize("name,builder,grule,max_it", [
    ("ka1_tridiagonal10", lambda: problems.tridiagonal(10), _lib.RULE_SIMPLE, 10),
    ("ka2_rcg21", lambda: problems.mgcg_main(21, 6, 10.0), _lib.RULE_NATIVE, 21),
    ("ka3_mgcgmain2000", lambda: problems.mgcg_main(2000, 160), _lib.RULE_CSHARP, 2000),
    ("poisson5_32x32", lambda: problems.poisson(32, 32, 1), _lib.RULE_NATIVE, 4096),
    ("poisson7_12x12x12", lambda: problems.poisson(12, 12, 12), _lib.RULE_NATIVE, 4096),
])
def test_known_answer_systems_equal_the_golden_vectors(reference_order, name, builder, grule, max_it):
    """KA-1 / KA-2 / KA-3 (the systems the reference hard-codes) and the two Poisson fixtures: the committed oracle outputs
    (tests/golden/*.npz: trace, x, iteration, residual) reproduced bit for bit by the HIP loop."""
    g = golden(name)
    cg = _solve_single(builder(), 0, max_it, 1e-8, grule)
    assert cg.Iteration == int(g["iteration"])
    assert_equal_bits(cg.trace, g["trace"], "trace")
    assert_equal_bits(cg.x, g["x_cg"], "x")
    assert cg.Residual == float(g["residual"])


@pytest.mark.parametrize("rule_o,rule_g", [("RULE_NATIVE", _lib.RULE_NATIVE), ("RULE_CSHARP", _lib.RULE_CSHARP), ("RULE_SIMPLE", _lib.RULE_SIMPLE),
                                           ("RULE_HANDMADECL", _lib.RULE_HANDMADECL), ("RULE_VIENNACL", _lib.RULE_VIENNACL)])
def test_every_stop_rule_equals_the_oracle(oracle, reference_order, rule_o, rule_g):
    s = problems.mgcg_main(1500, 160)
    s.x[:] = np.arange(s.Count) / 100.0
    ref = oracle.cg(s, rule=getattr(oracle, rule_o), allowable_residual=1e-6, min_iteration=3, max_iteration=1500, hard_cap=2000, trace=True)
    cg = _solve_single(s, 3, 1500, 1e-6, rule_g)
    assert cg.Iteration == ref["iteration"]
    assert_equal_bits(cg.trace, ref["trace"], "trace")
    assert_equal_bits(cg.x, ref["x"], "x")


def test_forced_iterations_into_the_round_off_tail_stay_equal(oracle, reference_order):
    """MgcgMain's MIN_ITERATION idiom (MgcgMain.cs:25): 60 forced iterations end deep in CG's round-off tail, where the default mode
    needs the oracle's own spread over device counts as its tolerance (tests/test_gpu_solve.py).  In the reference's order there is
    no tail to argue about: equal is equal."""
    s = problems.mgcg_main(1200, 160)
    ref = oracle.cg(s, rule=oracle.RULE_CSHARP, min_iteration=60, max_iteration=1200, trace=True)
    cg = _solve_single(s, 60, 1200, 1e-8, _lib.RULE_CSHARP)
    assert cg.Iteration == ref["iteration"] == 60
    assert_equal_bits(cg.trace, ref["trace"], "trace")
    assert_equal_bits(cg.x, ref["x"], "x")


@pytest.mark.parametrize("devices", [1, 2, 3, 4])
def test_phase_functions_equal_the_multi_device_oracle(oracle, reference_order, devices):
    """The reference's own shape: Solve0..3 + P2Host / P2Device driven from the host, one thread per device, partial sums added in
    device order by the host (ConjugateGradientParallelGpu.cs:424-565)."""
    reference_order.setenv("MGCG_VIRTUAL_DEVICES", str(devices))
    s = problems.mgcg_main(2403, 160)
    ref = oracle.cg_parallel(s, devices, max_iteration=s.Count, trace=True)
    cg = ConjugateGradientParallelGpu(s.Count, 160, 0, s.Count, 1e-8, deviceCount=devices).load(s)
    cg.Initialize()
    cg.Solve()
    cg.Read()
    assert cg.Iteration == ref["iteration"]
    assert cg.Residual == ref["residual"]
    assert_equal_bits(cg.x, ref["x"], "x")
    cg.Dispose()


def _systems(which, world):
    if which == "banded":
        return problems.mgcg_main(2403, 160)
    if which == "poisson":
        return problems.poisson(12, 10, 9)
    if which == "poisson32":
        return problems.poisson(32, 32, 16 if world <= 4 else 4 * world)
    s = problems.random_spd(1500, mean_upper=6.0, seed=21)
    s.b[:] = np.cos(np.arange(s.Count) * 0.7) * (1.0 + np.arange(s.Count) % 5)
    return s


@pytest.mark.parametrize("overlap", ["0", "2"])
@pytest.mark.parametrize("world,which", [(2, "banded"), (3, "poisson"), (2, "poisson32"), (8, "poisson32"), (2, "unstructured"), (8, "unstructured"), (3, "banded")])
def test_native_multirank_loop_equals_the_multi_device_oracle(oracle, reference_order, world, which, overlap):
    """SolveParallel over 2, 3 and 8 ranks (loopback transport on one GPU): trace and x equal to oracle.cg_parallel -- also on the
    random system whose iterates are chaotic below 1e-6 r0 and need a spread argument in the default mode -- and the same under both
    halo schedules (the serial sum runs after all rows of Ap are written, whatever wrote them)."""
    from tests.test_gpu_parallel import _run_ranks_in_threads

    reference_order.setenv("MGCG_VIRTUAL_DEVICES", str(world))
    reference_order.setenv("MGCG_OVERLAP", overlap)
    s = _systems(which, world)
    ref = oracle.cg_parallel(s, world, max_iteration=s.Count, trace=True)
    maxnz = int(np.diff(s.RowOffsets).max())

    def make_rank(rank, comm):
        cg = ConjugateGradientRankGpu(s.Count, maxnz, 0, s.Count, 1e-8, rank=rank, world=world, comm=comm, device=rank).load(s)
        cg.Initialize()
        cg.Solve(trace=True)
        cg.Read()
        out = (cg.part.offset, cg.part.count, cg.x[cg.part.offset: cg.part.offset + cg.part.count].copy(), cg.Iteration, cg.Residual, cg.trace)
        cg.Dispose()
        return out

    res = _run_ranks_in_threads(world, make_rank)
    x = np.zeros(s.Count)
    for off, cnt, xs, it, resid, tr in res:
        x[off: off + cnt] = xs
        assert it == ref["iteration"] and resid == ref["residual"]
        assert_equal_bits(tr, ref["trace"], "trace")
    assert_equal_bits(x, ref["x"], "x")


@pytest.mark.parametrize("world", [3, 8])
def test_partition_of_equal_nonzero_counts_equals_the_oracle_cut_at_the_same_rows(oracle, reference_order, world):
    from tests.test_gpu_parallel import _run_ranks_in_threads

    reference_order.setenv("MGCG_VIRTUAL_DEVICES", str(world))
    s = _systems("unstructured", world)
    off = problems.partition_offsets(s.Count, world, s.RowOffsets, "nnz")
    ref = oracle.cg_parallel(s, world, max_iteration=s.Count, trace=True, offsets=off)
    maxnz = int(np.diff(s.RowOffsets).max())

    def make_rank(rank, comm):
        cg = ConjugateGradientRankGpu(s.Count, maxnz, 0, s.Count, 1e-8, rank=rank, world=world, comm=comm, device=rank, balance="nnz").load(s)
        cg.Initialize()
        cg.Solve(trace=True)
        cg.Read()
        out = (cg.part.offset, cg.part.count, cg.x[cg.part.offset: cg.part.offset + cg.part.count].copy(), cg.Iteration, cg.trace)
        cg.Dispose()
        return out

    res = _run_ranks_in_threads(world, make_rank)
    x = np.zeros(s.Count)
    for o, cnt, xs, it, tr in res:
        x[o: o + cnt] = xs
        assert it == ref["iteration"]
        assert_equal_bits(tr, ref["trace"], "trace")
    assert_equal_bits(x, ref["x"], "x")


@pytest.mark.parametrize("dims,levels,interpolation,compression", [((16, 16, 16), 3, 0, 0), ((24, 12, 8), 2, 0, 0), ((16, 16, 16), 3, 1, 0), ((32, 32, 32), 3, 0, 1)])
def test_mgcg_equals_the_oracle(oracle, reference_order, dims, levels, interpolation, compression):
    """The preconditioned loop on one rank: V-cycle bit-identical (tests/test_gpu_mg.py) + sums in the reference's order = the oracle's
    PCG, bit for bit (plain CSR with the folded sweeps, linear transfer, and the opt-in row-pattern form)."""
    s = problems.poisson(*dims)
    s.b[:] = np.random.default_rng(3).standard_normal(s.Count)
    ref = oracle.Multigrid(s, levels=levels, interpolation=interpolation).pcg(rule=oracle.RULE_CSHARP, max_iteration=400, trace=True)
    mg = ConjugateGradientMgGpu(s.Count, 7, 0, 400, 1e-8, s.grid, levels=levels, interpolation=interpolation).load(s)
    _lib.lib().MgcgSetMatrixCompression(mg.cusparse, compression)
    mg.Initialize()
    mg.Solve(trace=True)
    mg.Read()
    assert mg.Iteration == ref["iteration"]
    assert_equal_bits(mg.trace, ref["trace"], "trace")
    assert_equal_bits(mg.x, ref["x"], "x")
    mg.Dispose()


@pytest.mark.parametrize("world,dims,levels,interpolation", [(2, (8, 8, 16), 3, 0), (4, (16, 8, 16), 2, 0), (3, (8, 4, 24), 3, 1), (8, (16, 16, 64), 3, 0),
                                                            (2, (8, 8, 32), 3, 0), (4, (16, 16, 64), 3, 0)])      # (the last two: slabs thick enough for the deep-halo cycle)
def test_row_partitioned_mgcg_equals_the_oracle_with_partitioned_sums(oracle, reference_order, world, dims, levels, interpolation):
    """Config 4 in miniature: z-slabs over 2-8 ranks; the oracle's PCG with its dot products cut at the same rows and added in rank order
    (oracle_pcg_parts) -- trace and x equal, under the overlap schedule with the interior folds."""
    from tests.test_gpu_parallel import _run_ranks_in_threads

    reference_order.setenv("MGCG_VIRTUAL_DEVICES", str(world))
    reference_order.setenv("MGCG_OVERLAP", "2")
    s = problems.poisson(*dims)
    s.b[:] = np.random.default_rng(3).standard_normal(s.Count)
    off = oracle.partition(s.Count, world)
    ref = oracle.Multigrid(s, levels=levels, interpolation=interpolation).pcg(rule=oracle.RULE_CSHARP, max_iteration=400, trace=True, offsets=off)

    def make_rank(rank, comm):
        cg = ConjugateGradientMgRankGpu(s.Count, 7, 0, 400, 1e-8, s.grid, rank=rank, world=world, comm=comm, device=rank, levels=levels,
                                        interpolation=interpolation).load(s)
        cg.Initialize()
        cg.Setup()
        cg.Solve(trace=True)
        cg.Read()
        out = (cg.part.offset, cg.part.count, cg.x[cg.part.offset: cg.part.offset + cg.part.count].copy(), cg.Iteration, cg.trace)
        cg.Dispose()
        return out

    res = _run_ranks_in_threads(world, make_rank)
    x = np.zeros(s.Count)
    for rank, (o, cnt, xs, it, tr) in enumerate(res):
        assert (o, cnt) == (off[rank], off[rank + 1] - off[rank])
        x[o: o + cnt] = xs
        assert it == ref["iteration"]
        assert_equal_bits(tr, ref["trace"], "trace")
    assert_equal_bits(x, ref["x"], "x")


@pytest.mark.parametrize("which", ["banded", "mgcg"])
def test_callback_transport_adds_the_ranks_sums_in_rank_order(oracle, tmp_path, which):
    """MgcgCommInitCallbacks (collectives carried by the launcher, here torch.distributed gloo between two processes that share the
    GPU): in this mode the ranks' sums travel through the caller's all-gather as bit patterns and are added in rank order by the
    library, so the result no longer depends on the order the caller's all-reduce adds in."""
    import subprocess

    world = 2
    port = 33600 + (os.getpid() % 2000) + (11 if which == "mgcg" else 0)
    worker = os.path.join(ROOT, "tests", "_callback_worker.py")
    env = dict(os.environ, MGCG_DOT_ORDER="1")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(tmp_path), which], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    if which == "banded":
        system = problems.mgcg_main(2400, 160)
        ref = oracle.cg_parallel(system, world, max_iteration=system.Count)
    else:
        system = problems.poisson(16, 16, 16)
        system.b[:] = np.random.default_rng(3).standard_normal(system.Count)
        ref = oracle.Multigrid(system).pcg(rule=oracle.RULE_CSHARP, max_iteration=400, offsets=oracle.partition(system.Count, world))
    x = np.zeros(system.Count)
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        x[int(d["offset"]): int(d["offset"]) + int(d["count"])] = d["x"]
        assert int(d["iteration"]) == ref["iteration"] and float(d["residual"]) == ref["residual"]
    assert_equal_bits(x, ref["x"], "x")


def test_one_rank_rccl_communicator_on_the_several_ranks_path_equals_the_oracle(oracle, reference_order):
    """A REAL one-rank RCCL communicator forced onto the code path of N > 1 (reduction launches, ncclAllReduce on the stream, grouped
    ncclSend/ncclRecv, fork / join, full-length multigrid iterates): the sum over one rank is the rank's own value, so CG and MGCG
    must equal the single-domain oracle bit for bit."""
    L = _lib.lib()
    L.SetDevice(0)
    buf = (C.c_char * 128)()
    assert L.MgcgCommGetUniqueId(buf) == 0, _lib.last_error()
    comm = L.MgcgCommInitRank(buf, 1, 0)
    assert comm and L.MgcgCommTransport(comm) == b"rccl", _lib.last_error()
    n = 32
    reference_order.setenv("MGCG_FORCE_MULTIRANK", str(n * n))
    reference_order.setenv("MGCG_OVERLAP", "2")
    s = problems.poisson(n, n, n)
    ref = oracle.cg(s, rule=oracle.RULE_CSHARP, max_iteration=2000, trace=True)
    cg = ConjugateGradientRankGpu(s.Count, 7, 0, 2000, 1e-8, rank=0, world=1, comm=comm)
    cg.InitializePoisson(n, n, n)
    cg.Solve(trace=True)
    cg.Read()
    assert cg.Iteration == ref["iteration"]
    assert_equal_bits(cg.trace, ref["trace"], "trace")
    assert_equal_bits(cg.x, ref["x"], "x")
    cg.Dispose()
    mref = oracle.Multigrid(s, levels=3).pcg(rule=oracle.RULE_CSHARP, max_iteration=400, trace=True)
    mg = ConjugateGradientMgRankGpu(s.Count, 7, 0, 400, 1e-8, (n, n, n), rank=0, world=1, comm=comm, levels=3)
    mg.InitializePoisson(n, n, n)
    mg.Setup()
    mg.Solve(trace=True)
    mg.Read()
    assert mg.Iteration == mref["iteration"]
    assert_equal_bits(mg.trace, mref["trace"], "trace")
    assert_equal_bits(mg.x, mref["x"], "x")
    mg.Dispose()
    L.MgcgCommDestroy(comm)


def test_config2_sixty_cg_iterations_equal_the_oracle_at_full_size(oracle, reference_order):
    """BASELINE config 2 at its size (7-point 256^3, 16 777 216 rows, 117 047 296 nonzeros): 60 iterations of the unpreconditioned loop on
    one rank and 12 on eight loopback ranks -- residual trace and all of x equal to oracle.cg / oracle.cg_parallel, bit for bit."""
    from tests.test_gpu_parallel import _run_ranks_in_threads

    n, its = 256, 60
    N = n**3
    e, c, r = oracle.poisson_csr(n, n, n)
    s = problems.LinearSystem(e, c, r, np.zeros(N), np.ones(N), f"poisson{n}", grid=(n, n, n))
    ref = oracle.cg(s, rule=oracle.RULE_NATIVE, allowable_residual=1e300, min_iteration=its - 1, max_iteration=its + 2, hard_cap=its + 3, trace=True)
    assert ref["iteration"] == its - 1
    cg = ConjugateGradientRankGpu(N, 7, its - 1, 1000, 1e300, rank=0, world=1, rule=_lib.RULE_NATIVE)
    cg.InitializePoisson(n, n, n)
    cg.Solve(trace=True)
    assert cg.Iteration == its - 1
    x = np.empty(N)
    cg.vectorX.CopyTo(x, N, 0)
    cg.Dispose()
    assert_equal_bits(cg.trace, ref["trace"], "trace")
    assert_equal_bits(x, ref["x"], "x")
    # eight z-slabs of 32 planes (the reference's partition: floor(N / 8) rows each)
    world, its8 = 8, 12
    ref8 = oracle.cg_parallel(s, world, allowable_residual=1e300, min_iteration=its8 - 1, max_iteration=its8 + 2, trace=True)
    assert ref8["iteration"] == its8 - 1
    reference_order.setenv("MGCG_VIRTUAL_DEVICES", str(world))
    reference_order.setenv("MGCG_OVERLAP", "2")

    def make_rank(rank, comm):
        cg = ConjugateGradientRankGpu(N, 7, its8 - 1, 1000, 1e300, rank=rank, world=world, comm=comm, device=rank)
        cg.InitializePoisson(n, n, n)
        cg.Solve(trace=True)
        xs = np.empty(cg.part.count)
        cg.vectorX.CopyTo(xs, cg.part.count, 0)
        out = (cg.part.offset, cg.part.count, xs, cg.Iteration, cg.trace)
        cg.Dispose()
        return out

    for off, cnt, xs, it, tr in _run_ranks_in_threads(world, make_rank):
        assert it == its8 - 1
        assert_equal_bits(tr, ref8["trace"], "8 ranks: trace")
        assert_equal_bits(xs, ref8["x"][off: off + cnt], "8 ranks: x")


def test_the_other_two_families_front_ends_equal_the_oracle(oracle, reference_order):
    """the HandmadeCL family's class (max-norm residual, MIN_ITERATION idiom) and the ViennaCL family's
    ComputerGpu (relative rule) on the systems their drivers build (MgcgCLMain.cs:52-90, MgcgCL.cs:31-45): Iteration and every entry of x
    equal to the oracle's loop with the same rule."""
    from conjugategradient_amd.frontends import ComputerGpu, ConjugateGradientCLGpu

    n, K = 400, 160
    s = problems.mgcg_main(n, K)
    cl = ConjugateGradientCLGpu(n, K, 50, n, 1e-4)
    for i in range(n):
        lo, hi = int(s.RowOffsets[i]), int(s.RowOffsets[i + 1])
        for k in range(lo, hi):                                          # (diagonal first, then ascending columns: the builder's own order)
            cl.A[i, int(s.ColumnIndeces[k])] = float(s.Elements[k])
    cl.b[:] = s.b
    cl.x[:] = s.x
    ref = oracle.cg(s, rule=oracle.RULE_HANDMADECL, allowable_residual=1e-4, min_iteration=50, max_iteration=n)
    cl.Initialize()
    cl.Solve()
    cl.Read()
    assert cl.Iteration == ref["iteration"] == 50 and cl.Residual == ref["residual"]
    assert_equal_bits(cl.x, ref["x"], "HandmadeCL family x")
    cl.Dispose()
    v = problems.viennacl_main(n, K)
    ref = oracle.cg(v, rule=oracle.RULE_VIENNACL, allowable_residual=1e-4, min_iteration=0, max_iteration=n, hard_cap=n + 10)
    gpu = ComputerGpu(n)
    gpu.Write(v.Elements, v.RowOffsets, v.ColumnIndeces, np.zeros(n), v.b)
    gpu.Solve(1e-4, 0, n)
    x = np.zeros(n)
    gpu.Read(x)
    assert gpu.Iteration() == ref["iteration"] + 1
    assert_equal_bits(x, ref["x"], "ViennaCL family x")
    gpu.Dispose()
