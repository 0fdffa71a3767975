"""Opt-in dictionary-compressed CSR (MgcgSetMatrixCompression): a lossless re-encoding, so every result must be
bit-identical to the CSR kernels / the oracle -- SpMV, every fused epilogue through the V-cycle, whole solves,
one rank and several."""
import ctypes as C

import numpy as np
import pytest

from conjugategradient_amd import _lib, problems
from conjugategradient_amd.multigrid import ConjugateGradientMgGpu
from conjugategradient_amd.solver import ConjugateGradientSingleGpu
from tests.gpu_util import DeviceCsr, Handles

pytestmark = pytest.mark.gpu


def _info(sparse, idx=0):
    L = _lib.lib()
    d, v, r, n = C.c_int(), C.c_int(), C.c_longlong(), C.c_longlong()
    cls = L.MgcgAnalysisInfo(sparse, idx, C.byref(d), C.byref(v), C.byref(r), C.byref(n))
    return cls, d.value, v.value, r.value, n.value


def _scaled_poisson(dims, seed=4):
    import scipy.sparse as sp
    s = problems.poisson(*dims)
    rng = np.random.default_rng(seed)
    d = sp.diags(1.0 + rng.random(s.Count))
    A = (d @ s.to_scipy() @ d).tocsr()
    A.sort_indices()
    return problems.LinearSystem(A.data.copy(), A.indices.astype(np.int32), A.indptr.astype(np.int32), np.zeros(s.Count), np.ones(s.Count), "scaled", grid=s.grid)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("name,builder,expect_class,expect_offsets,expect_values", [
    ("poisson7", lambda: problems.poisson(20, 17, 13), 2, 7, 2),
    ("poisson5", lambda: problems.poisson(37, 29, 1), 2, 5, 2),
    ("tridiagonal", lambda: problems.tridiagonal(1003), 2, 3, 2),               # unsorted columns: diag, left, right
    ("scaled poisson", lambda: _scaled_poisson((12, 11, 10)), 1, 7, None),      # every value distinct: offsets coded only
    ("narrow band", lambda: problems.mgcg_main(3000, 8), 1, 7, None),           # diagonal first, |sin(i+j)| values
    ("random", lambda: problems.random_spd(4000, mean_upper=2.0, seed=9), 0, None, None),   # > 256 distinct offsets: stays CSR
    ("tiny", lambda: problems.poisson(2, 1, 1), 0, None, None),                 # fewer than 8 nonzeros: not encoded
])
def test_compressed_spmv_is_bit_exact(oracle, name, builder, expect_class, expect_offsets, expect_values, mode):
    """mode 2: per-nonzero codes (classes 2 / 1); mode 1: the best form, i.e. one byte per row (class 3) for the
    constant-coefficient stencils and the per-nonzero codes for the rest."""
    L = _lib.lib()
    s = builder()
    rng = np.random.default_rng(5)
    x = rng.standard_normal(s.Count)
    ref = oracle.spmv(s.Elements, s.ColumnIndeces, s.RowOffsets, x)
    h = Handles()
    A = DeviceCsr(s)
    plain = A.spmv(h, x, kernel=1)
    L.MgcgSetMatrixCompression(h.sparse, mode)
    got = A.spmv(h, x)
    assert np.array_equal(plain, ref) and np.array_equal(got, ref)
    cls, nd, nv, rows, nnz = _info(h.sparse)
    if mode == 1 and expect_class == 2:          # few distinct values and offsets here also means few distinct rows
        assert cls == 3, (cls, nd, nv)
        expected_rows = {"poisson7": 27, "poisson5": 9, "tridiagonal": 3}[name]
        assert (nd, nv) == (expected_rows, expect_offsets)       # distinct rows, longest row
        y0 = rng.standard_normal(s.Count)
        assert np.array_equal(A.spmv(h, x, alpha=-1.5, beta=0.25, y0=y0), -1.5 * ref + 0.25 * y0)
        L.MgcgAnalysisClear(h.sparse)
        h.close()
        return
    if expect_class == 0:
        assert cls in (0, -1), (cls, nd, nv)      # analysed and rejected, or not even a candidate (long average rows)
        h.close()
        return
    assert cls == expect_class, (cls, nd, nv)
    if expect_offsets is not None:
        assert nd == expect_offsets
    if expect_values is not None:
        assert nv == expect_values
    # alpha/beta epilogue and a second call (cache hit) give the same bits
    y0 = rng.standard_normal(s.Count)
    assert np.array_equal(A.spmv(h, x, alpha=-1.5, beta=0.25, y0=y0), -1.5 * ref + 0.25 * y0)
    assert _info(h.sparse, 1)[0] == -1          # still one cached analysis
    L.MgcgAnalysisClear(h.sparse)
    assert _info(h.sparse, 0)[0] == -1
    h.close()


def test_compressed_solves_match_plain_bits(oracle):
    """CG and MGCG with compression on.  Per-nonzero codes (mode 2): x, iteration count and the whole residual trace equal
    the uncompressed run bit for bit (same arithmetic, same reduction grids).  Best form (mode 1, one byte per row here):
    every SpMV is still bit-identical, but the fused p.Ap partial sums are grouped by 128-row blocks instead of 64, so
    the trace agrees to the dot-product tolerance of every other CG test (1e-10 while above round-off)."""
    from tests.gpu_util import assert_trace_close

    for s, mg in [(problems.poisson(24, 20, 16), False), (problems.poisson(16, 16, 16), True), (_scaled_poisson((16, 12, 8)), True)]:
        runs = []
        for comp in (0, 1, 2):
            if mg:
                cg = ConjugateGradientMgGpu(s.Count, 7, 0, 1000, 1e-9, s.grid, rule=_lib.RULE_CSHARP).load(s)
            else:
                cg = ConjugateGradientSingleGpu(s.Count, 7, 0, 1000, 1e-9, rule=_lib.RULE_CSHARP).load(s)
            _lib.lib().MgcgSetMatrixCompression(cg.cusparse, comp)
            cg.Initialize()
            cg.Solve(trace=True)
            cg.Read()
            if comp:
                assert _info(cg.cusparse, 0)[0] >= 1
                if mg:
                    assert _info(cg.cusparse, 2)[0] >= 1      # every level analysed
            runs.append((cg.x.copy(), cg.Iteration, cg.trace.copy()))
            cg.Dispose()
        plain, best, codes = runs
        assert plain[1] == codes[1] and np.array_equal(plain[2], codes[2]) and np.array_equal(plain[0], codes[0])
        assert plain[1] == best[1]
        assert_trace_close(best[2], plain[2])
        assert np.abs(best[0] - plain[0]).max() <= 1e-10 * np.abs(plain[0]).max()
    # and against the oracle's V-cycle: still bit-identical
    s = problems.poisson(16, 16, 16)
    cg = ConjugateGradientMgGpu(s.Count, 7, 0, 500, 1e-8, s.grid).load(s)
    _lib.lib().MgcgSetMatrixCompression(cg.cusparse, 1)
    cg.Initialize()
    r = np.random.default_rng(1).standard_normal(s.Count)
    assert np.array_equal(cg.Apply(r), oracle.Multigrid(s).apply(r))
    cg.Dispose()


@pytest.mark.parametrize("mode", [1, 2])
def test_compressed_multirank(oracle, mgcg_env, mode):
    """Row slices with a non-zero row base (col - row uses the GLOBAL row) through the loopback transport."""
    from conjugategradient_amd.parallel import ConjugateGradientMgRankGpu
    from tests.test_gpu_parallel import _run_ranks_in_threads

    world = 2
    mgcg_env.setenv("MGCG_VIRTUAL_DEVICES", str(world))
    mgcg_env.setenv("MGCG_OVERLAP", "2")              # compressed interior / boundary row ranges while the halo travels
    s = problems.poisson(8, 8, 16)
    s.b[:] = np.random.default_rng(3).standard_normal(s.Count)
    ref = oracle.Multigrid(s).pcg(rule=oracle.RULE_CSHARP, max_iteration=400)

    def make_rank(rank, comm):
        cg = ConjugateGradientMgRankGpu(s.Count, 7, 0, 400, 1e-8, s.grid, rank=rank, world=world, comm=comm, device=rank).load(s)
        _lib.lib().MgcgSetMatrixCompression(cg.cusparse, mode)
        cg.Initialize()
        cg.Setup()
        cg.Solve()
        cg.Read()
        assert _info(cg.cusparse, 0)[0] == (3 if mode == 1 else 2)
        out = (cg.part.offset, cg.part.count, cg.x[cg.part.offset: cg.part.offset + cg.part.count].copy(), cg.Iteration)
        cg.Dispose()
        return out

    x = np.zeros(s.Count)
    for off, cnt, xs, it in _run_ranks_in_threads(world, make_rank):
        x[off: off + cnt] = xs
        assert it == ref["iteration"]
    assert np.abs(x - ref["x"]).max() <= 1e-10 * np.abs(ref["x"]).max()


def _stencil27(n):
    """27-point stencil on an n^3 box (tensor product of 1-D [1 4 1] masses with a shifted diagonal): rows of up to 27
    entries, 27 distinct rows-as-sequences."""
    import scipy.sparse as sp
    t = sp.diags([1.0, 4.0, 1.0], [-1, 0, 1], shape=(n, n))
    A = (sp.kron(sp.kron(t, t), t) + 100.0 * sp.identity(n**3)).tocsr()
    A.sort_indices()
    return problems.LinearSystem(A.data.copy(), A.indices.astype(np.int32), A.indptr.astype(np.int32), np.zeros(n**3), np.ones(n**3), "stencil27", grid=(n, n, n))


def test_row_patterns_with_long_rows_and_many_passes(oracle):
    """Rows longer than one pass of the kernel (27 entries: four passes of 8 slots, the last one masked), mixed with
    shorter boundary rows in the same wavefront."""
    L = _lib.lib()
    s = _stencil27(9)
    rng = np.random.default_rng(11)
    x = rng.standard_normal(s.Count)
    ref = oracle.spmv(s.Elements, s.ColumnIndeces, s.RowOffsets, x)
    h = Handles()
    A = DeviceCsr(s)
    L.MgcgSetMatrixCompression(h.sparse, 1)
    assert np.array_equal(A.spmv(h, x), ref)
    cls, nrows_distinct, longest, rows, nnz = _info(h.sparse)
    assert (cls, nrows_distinct, longest) == (3, 27, 27)
    y0 = rng.standard_normal(s.Count)
    assert np.array_equal(A.spmv(h, x, alpha=0.5, beta=-2.0, y0=y0), 0.5 * ref + -2.0 * y0)
    h.close()
    # whole solve through the same form: the oracle's iteration count and x
    r = oracle.cg(s, rule=oracle.RULE_CSHARP, max_iteration=200)
    cg = ConjugateGradientSingleGpu(s.Count, 27, 0, 200, 1e-8, rule=_lib.RULE_CSHARP).load(s)
    L.MgcgSetMatrixCompression(cg.cusparse, 1)
    cg.Initialize()
    cg.Solve()
    cg.Read()
    assert _info(cg.cusparse)[0] == 3
    assert cg.Iteration == r["iteration"] and np.abs(cg.x - r["x"]).max() <= 1e-10 * np.abs(r["x"]).max()
    cg.Dispose()


def test_too_many_distinct_rows_fall_back_to_codes_or_csr(oracle):
    """257+ distinct rows: the row-pattern analysis declines, the per-nonzero codes take over (few offsets), and a matrix
    with neither property stays in CSR -- results identical in every case."""
    L = _lib.lib()
    s = problems.poisson(20, 17, 13)
    rng = np.random.default_rng(2)
    vals = s.Elements.copy()
    diag = s.ColumnIndeces == np.repeat(np.arange(s.Count), np.diff(s.RowOffsets))
    vals[diag] = 6.0 + (np.arange(s.Count) % 300)              # 300 distinct diagonals: 300+ distinct rows, 301 distinct values
    t = problems.LinearSystem(vals, s.ColumnIndeces, s.RowOffsets, s.x, s.b, "poisson-300-diagonals", grid=s.grid)
    x = rng.standard_normal(t.Count)
    ref = oracle.spmv(t.Elements, t.ColumnIndeces, t.RowOffsets, x)
    h = Handles()
    A = DeviceCsr(t)
    L.MgcgSetMatrixCompression(h.sparse, 1)
    assert np.array_equal(A.spmv(h, x), ref)
    cls, nd, nv, _, _ = _info(h.sparse)
    assert (cls, nd) == (1, 7) and nv == 0                     # offsets coded, values stay fp64
    h.close()


@pytest.mark.parametrize("pack", ["1", "0"])
def test_column_tiled_form_for_matrices_without_locality(oracle, mgcg_env, pack):
    """Class 4 (BASELINE config 5 in miniature; MGCG_TILE_SHIFT shrinks the tile so that a 40 000-column matrix needs 10
    tiles): sorted random rows are re-laid out by column tile, the running row sums travel through y from tile to tile
    in stored order -- bit-identical products; unsorted rows make the analysis decline.  pack: 12-byte entries (value + one
    packed word, the default) or the 16-byte form (MGCG_TILE_PACK=0)."""
    L = _lib.lib()
    mgcg_env.setenv("MGCG_TILE_SHIFT", "12")
    mgcg_env.setenv("MGCG_TILE_PACK", pack)
    s = problems.random_spd(40000, mean_upper=14.0, seed=3)
    rng = np.random.default_rng(8)
    x = rng.standard_normal(s.Count)
    ref = oracle.spmv(s.Elements, s.ColumnIndeces, s.RowOffsets, x)
    h = Handles()
    A = DeviceCsr(s)
    L.MgcgSetMatrixCompression(h.sparse, 1)
    assert np.array_equal(A.spmv(h, x), ref)
    cls, tiles, _, rows, nnz = _info(h.sparse)
    assert (cls, tiles, rows, nnz) == (4, 10, s.Count, s.nnz)
    y0 = rng.standard_normal(s.Count)
    assert np.array_equal(A.spmv(h, x, alpha=2.0, beta=0.0, y0=y0), 2.0 * ref)
    # beta != 0 reads y, which the tiled passes use as their accumulator: served by the CSR kernel for this row length
    # (16 lanes per row: tree-summed, hence a tolerance)
    np.testing.assert_allclose(A.spmv(h, x, alpha=-1.5, beta=0.25, y0=y0), -1.5 * ref + 0.25 * y0, rtol=1e-13, atol=1e-13 * np.abs(ref).max())
    h.close()
    # whole solve (fused p.Ap on the last tile, residual epilogue at the start)
    s.b[:] = oracle.spmv(s.Elements, s.ColumnIndeces, s.RowOffsets, np.cos(np.arange(s.Count) * 0.01) + 2.0)
    r = oracle.cg(s, rule=oracle.RULE_CSHARP, max_iteration=1000, trace=True)
    cg = ConjugateGradientSingleGpu(s.Count, int(np.diff(s.RowOffsets).max()), 0, 1000, 1e-8, rule=_lib.RULE_CSHARP).load(s)
    L.MgcgSetMatrixCompression(cg.cusparse, 1)
    cg.Initialize()
    cg.Solve(trace=True)
    cg.Read()
    assert _info(cg.cusparse)[0] == 4
    assert cg.Iteration == r["iteration"] and np.abs(cg.x - r["x"]).max() <= 1e-10 * np.abs(r["x"]).max()
    cg.Dispose()
    # rows stored in descending column order: no tiling (the summation order could not be kept), still exact
    u = problems.random_spd(40000, mean_upper=14.0, seed=3, sort_columns=False)
    refu = oracle.spmv(u.Elements, u.ColumnIndeces, u.RowOffsets, x)
    h = Handles()
    B = DeviceCsr(u)
    L.MgcgSetMatrixCompression(h.sparse, 1)
    assert np.array_equal(B.spmv(h, x, kernel=1), refu)
    assert _info(h.sparse)[0] in (0, -1)
    h.close()


def test_column_tiles_on_row_slices_over_loopback(oracle, mgcg_env):
    """Two ranks, unstructured matrix: every rank tiles its own row slice over the GLOBAL column range and the halo
    degenerates to an all-gather of p; same iteration count and solution as the multi-device oracle."""
    from conjugategradient_amd.parallel import ConjugateGradientRankGpu
    from tests.test_gpu_parallel import _run_ranks_in_threads

    world = 2
    mgcg_env.setenv("MGCG_VIRTUAL_DEVICES", str(world))
    mgcg_env.setenv("MGCG_TILE_SHIFT", "9")          # 512-column tiles: 12 of them, well below the mean |col - row| of both slices
    s = problems.random_spd(6000, mean_upper=8.0, seed=17)
    s.b[:] = np.cos(np.arange(s.Count) * 0.7) * (1.0 + np.arange(s.Count) % 5)
    ref = oracle.cg_parallel(s, world, max_iteration=s.Count)
    maxnz = int(np.diff(s.RowOffsets).max())

    def make_rank(rank, comm):
        cg = ConjugateGradientRankGpu(s.Count, maxnz, 0, s.Count, 1e-8, rank=rank, world=world, comm=comm, device=rank).load(s)
        _lib.lib().MgcgSetMatrixCompression(cg.cusparse, 1)
        cg.Initialize()
        cg.Solve()
        cg.Read()
        assert _info(cg.cusparse, 0)[0] == 4
        out = (cg.part.offset, cg.part.count, cg.x[cg.part.offset: cg.part.offset + cg.part.count].copy(), cg.Iteration)
        cg.Dispose()
        return out

    x = np.zeros(s.Count)
    for off, cnt, xs, it in _run_ranks_in_threads(world, make_rank):
        x[off: off + cnt] = xs
        assert it == ref["iteration"]
    assert np.abs(x - ref["x"]).max() <= 1e-10 * np.abs(ref["x"]).max()


@pytest.mark.parametrize("mode", [1, 2])
def test_reinitialize_with_other_values_reanalyses(oracle, mode):
    """The reference re-uploads A into the SAME device vectors on every Initialize()
    (Mgcg/cuBlas/Mgcg/ConjugateGradientSingleGpu.cs:134-147); an analysis made from the old values must not survive it."""
    L = _lib.lib()
    s1 = problems.poisson(14, 12, 10)
    cg = ConjugateGradientSingleGpu(s1.Count, 7, 0, 5000, 1e-8).load(s1)
    L.MgcgSetMatrixCompression(cg.cusparse, mode)
    cg.Initialize()
    cg.Solve()
    cg.Read()
    ref1 = oracle.cg(s1, rule=oracle.RULE_NATIVE, max_iteration=5000, hard_cap=5000)
    assert cg.Iteration == ref1["iteration"] and np.abs(cg.x - ref1["x"]).max() <= 1e-10 * np.abs(ref1["x"]).max()
    assert _info(cg.cusparse)[0] in (1, 2, 3)
    # same sparsity, same nnz, other values (diagonal 7, off-diagonals -1): same device vectors, same sizes
    e2 = s1.Elements.copy()
    e2[e2 > 0] = 7.0
    s2 = problems.LinearSystem(e2, s1.ColumnIndeces.copy(), s1.RowOffsets.copy(), np.zeros(s1.Count), np.ones(s1.Count), "poisson+I", grid=s1.grid)
    cg.A.Elements[: s2.nnz] = e2[: s2.nnz]
    cg.x[:] = 0.0
    cg.Initialize()
    cg.Solve()
    cg.Read()
    ref2 = oracle.cg(s2, rule=oracle.RULE_NATIVE, max_iteration=5000, hard_cap=5000)
    assert ref2["iteration"] != ref1["iteration"]
    assert cg.Iteration == ref2["iteration"], (cg.Iteration, ref2["iteration"], ref1["iteration"])
    assert np.abs(cg.x - ref2["x"]).max() <= 1e-10 * np.abs(ref2["x"]).max()
    cg.Dispose()


@pytest.mark.parametrize("mode", [1, 2])
def test_raw_pointer_writes_invalidate(oracle, mode):
    """CopyFromArray / Scal / Copy onto an analysed array, and a freed-and-reallocated array, all void the cached analysis."""
    L = _lib.lib()
    s = problems.poisson(16, 9, 7)
    rng = np.random.default_rng(31)
    x = rng.standard_normal(s.Count)
    h = Handles()
    A = DeviceCsr(s)
    L.MgcgSetMatrixCompression(h.sparse, mode)
    ref = oracle.spmv(s.Elements, s.ColumnIndeces, s.RowOffsets, x)
    assert np.array_equal(A.spmv(h, x), ref)
    L.Scal(h.blas, A.e.ToRawPtr(), 2.0, s.nnz)                               # every value doubled in place
    assert np.array_equal(A.spmv(h, x), oracle.spmv(2.0 * s.Elements, s.ColumnIndeces, s.RowOffsets, x))
    e3 = s.Elements[: s.nnz] * np.linspace(1.0, 2.0, s.nnz)
    A.e.CopyFrom(e3, s.nnz)                                                  # CopyFromArray_Double
    assert np.array_equal(A.spmv(h, x), oracle.spmv(e3, s.ColumnIndeces, s.RowOffsets, x))
    h.close()


def test_environment_selects_per_nonzero_codes(mgcg_env):
    """MGCG_COMPRESSION=2 means mode 2 (per-nonzero codes), as the setter does -- not 'anything non-zero is mode 1'."""
    L = _lib.lib()
    mgcg_env.setenv("MGCG_COMPRESSION", "2")
    s = problems.poisson(12, 10, 8)
    h = Handles()
    A = DeviceCsr(s)
    A.spmv(h, np.ones(s.Count))
    assert _info(h.sparse)[0] in (1, 2)
    h.close()
