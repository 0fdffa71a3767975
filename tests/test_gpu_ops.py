"""HIP kernels vs the CPU oracle, through the C ABI (raw-device-pointer ops of Mgcg.cu:10-54)."""
import ctypes as C
import os

import numpy as np
import pytest

from conjugategradient_amd import _lib, problems
from tests.gpu_util import DeviceCsr, Handles, dvec, ivec

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def h():
    hh = Handles()
    yield hh
    hh.close()


def _systems():
    yield problems.poisson(37, 29, 1)                 # 5-point, odd sizes
    yield problems.poisson(20, 17, 13)                # 7-point, ragged row-block tail
    yield problems.tridiagonal(1000)                  # diag,left,right order (unsorted)
    yield problems.mgcg_main(3000, 160)               # 159/row, diagonal first
    yield problems.mgcg_main(700, 24)
    yield problems.random_spd(5000, mean_upper=14.0, seed=7)
    yield problems.random_spd(3000, mean_upper=3.0, seed=8, sort_columns=False)


@pytest.mark.parametrize("system", list(_systems()), ids=lambda s: s.name)
def test_csrmv_stream_kernel_is_bit_exact(h, oracle, system):
    """Row-block LDS kernel: products rounded, added in stored order => identical bits to
    SparseMatrix.Multiply (SparseMatrix.cs:68-88), for every rows-per-block / mapping variant."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal(system.Count)
    ref = oracle.spmv(system.Elements, system.ColumnIndeces, system.RowOffsets, x)
    A = DeviceCsr(system)
    for tuning in [(256, 0, 0), (128, 0, 0), (64, 0, 0), (32, 0, 0), (32, 1, 24), (64, 2, 0), (256, 1, 0), (256, 2, 0), (256, 3, 0), (256, 0, 16)]:
        y = A.spmv(h, x, kernel=1, tuning=tuning)
        assert np.array_equal(y, ref), f"tuning {tuning}"
    for grid in (0, 7, 4096):                          # "stage raw, multiply by row" form of the row-block kernel
        assert np.array_equal(A.spmv(h, x, kernel=9, tuning=(64, 0, grid)), ref), f"rows kernel grid {grid}"
    for grid in (0, 4, 40, 4096):                      # row-tile kernel (the default for short rows); long rows take its slow path
        assert np.array_equal(A.spmv(h, x, kernel=10, tuning=(64, 0, grid)), ref), f"row-tile kernel grid {grid}"
    y = A.spmv(h, x)                                   # auto selection
    np.testing.assert_allclose(y, ref, rtol=1e-13, atol=1e-13 * np.abs(ref).max())


@pytest.mark.parametrize("kernel", [3, 4, 5, 6, 7, 8])
def test_csrmv_vector_kernels(h, oracle, kernel):
    """2..64 lanes per row with a shuffle tree: same value up to summation order (rtol 1e-13)."""
    for system in (problems.mgcg_main(2000, 160), problems.poisson(15, 14, 13), problems.random_spd(4000, seed=5)):
        x = np.sin(np.arange(system.Count) * 0.11) + 0.5
        ref = oracle.spmv(system.Elements, system.ColumnIndeces, system.RowOffsets, x)
        y = DeviceCsr(system).spmv(h, x, kernel=kernel)
        np.testing.assert_allclose(y, ref, rtol=2e-13, atol=2e-13 * np.abs(ref).max())


@pytest.mark.parametrize("dims,rows,grid,period,tile", [
    ((32, 16, 12), 64, 0, 512, (0, 0)),        # 1 row block per XCD per plane, many planes in flight
    ((64, 32, 6), 64, 16, 2048, (0, 0)),       # 4 row blocks per XCD per plane but only 2 workgroups per XCD
    ((64, 32, 9), 128, 0, 2048, (0, 0)),       # planes not divisible by the default plane count
    ((64, 64, 6), 256, 64, 4096, (0, 0)),      # R = 256 (two chunks per lane)
    ((64, 64, 8), 64, 0, 4096, (128, 4)),      # explicit tile: 2 row blocks x 4 planes
    ((64, 64, 8), 32, 0, 4096, (128, 2)),      # single-wavefront workgroups of 32 rows
    ((64, 64, 8), 64, 0, 4096, (512, 1)),      # explicit tile: the whole eighth, one plane
    ((64, 64, 8), 64, 0, 4096, (192, 3)),      # tile that does not divide -> default tile
    ((20, 17, 13), 128, 0, 340, (0, 0)),       # period the kernel cannot use -> silent fallback, same bits
])
def test_csrmv_banded_schedule_is_bit_exact(h, oracle, dims, rows, grid, period, tile):
    """XCD-aware banded schedule (flag bit2): a different traversal order of the row blocks, identical results."""
    s = problems.poisson(*dims)
    rng = np.random.default_rng(17)
    x = rng.standard_normal(s.Count)
    ref = oracle.spmv(s.Elements, s.ColumnIndeces, s.RowOffsets, x)
    y = DeviceCsr(s).spmv(h, x, kernel=1, tuning=(rows, 0, grid), period=period, tile=tile)
    assert np.array_equal(y, ref)
    y = DeviceCsr(s).spmv(h, x, kernel=1, tuning=(rows, 1, grid), period=period, tile=tile)      # + non-temporal loads
    assert np.array_equal(y, ref)


@pytest.mark.parametrize("dims,grid,period", [
    ((64, 32, 5), 0, 2048),          # one tile per XCD per plane, z sweep over 5 planes
    ((64, 64, 7), 64, 4096),         # 2 tiles per XCD per plane, 8 workgroups per XCD: every plane in one trip
    ((128, 64, 6), 32, 8192),        # 4 tiles per XCD per plane, 1 workgroup slot per XCD: 4 slices
    ((64, 96, 4), 64, 6144),         # 3 tiles per XCD per plane
    ((64, 32, 5), 0, 0),             # no hint: the far band is read off the matrix (2048)
    ((64, 32, 5), 0, 4096),          # a wrong hint that still tiles: another order, same bits
    ((64, 32, 5), 0, 640),           # a hint the kernel cannot use: plain order
    ((20, 17, 13), 0, 340),          # rows not a multiple of the tile: tail rows by workgroup 0
    ((64, 32, 2), 0, 2048),          # fewer than 3 planes: plain order
])
def test_csrmv_rowtile_orders_are_bit_exact(h, oracle, dims, grid, period):
    """Row-tile kernel: grid-stride order or the z sweep (XCD-contiguous plane slices); every epilogue variant of CsrMV."""
    s = problems.poisson(*dims)
    rng = np.random.default_rng(23)
    x, y0 = rng.standard_normal(s.Count), rng.standard_normal(s.Count)
    ref = oracle.spmv(s.Elements, s.ColumnIndeces, s.RowOffsets, x)
    A = DeviceCsr(s)
    assert np.array_equal(A.spmv(h, x, kernel=10, tuning=(64, 0, grid), period=period), ref)
    assert np.array_equal(A.spmv(h, x, alpha=-0.5, beta=1.75, y0=y0, kernel=10, tuning=(64, 0, grid), period=period), -0.5 * ref + 1.75 * y0)
    assert np.array_equal(A.spmv(h, x, alpha=1.0, beta=0.0, y0=np.full(s.Count, np.nan), kernel=10, tuning=(64, 0, grid), period=period), ref)


def test_csrmv_rowtile_slow_blocks_and_ragged_ends(h, oracle):
    """Blocks the fast path must decline: a row of more than 8 nonzeros, a 64-row span of more than 512 nonzeros, the last
    nonzeros of arrays whose length is not a multiple of four, matrices smaller than a tile, empty rows."""
    import scipy.sparse as sp
    rng = np.random.default_rng(29)
    cases = []
    base = problems.poisson(24, 20, 9).to_scipy().tolil()
    base[1000, 5:40] = 0.25                                   # one long row inside a short-row matrix
    base[3000:3064, 100:110] = -0.5                           # 64 consecutive rows of 15+: span > 512
    cases.append(base.tocsr())
    for n in (1, 7, 63, 64, 255, 256, 257, 511, 777):         # below / at / above one tile, nnz of any residue mod 4
        cases.append(sp.random(n, n, density=min(1.0, 5.0 / n), random_state=n, format="csr") + sp.eye(n, format="csr") * (n % 3 == 0))
    for M in cases:
        M = M.tocsr()
        M.sort_indices()
        sysm = problems.LinearSystem(M.data.astype(np.float64), M.indices.astype(np.int32), M.indptr.astype(np.int32), np.zeros(M.shape[0]), np.zeros(M.shape[0]), "case")
        xv = rng.standard_normal(M.shape[0])
        ref = oracle.spmv(sysm.Elements, sysm.ColumnIndeces, sysm.RowOffsets, xv)
        got = DeviceCsr(sysm).spmv(h, xv, kernel=10)
        assert np.array_equal(got, ref), (M.shape, M.nnz)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MGCG_FUZZ_SEEDS", "12"))))     # MGCG_FUZZ_SEEDS=400 for a long soak
def test_csrmv_rowtile_random_shapes(h, oracle, seed):
    """Randomised shapes for the row-tile kernel: sizes around the tile and block boundaries, densities that mix the fast path
    (rows <= 7 / 8, spans <= 512) with slow blocks, empty rows, unsorted columns, nnz of every residue mod 4; CsrMV and the
    fused CsrMVDot against the oracle (product bit for bit, dot to summation order)."""
    import scipy.sparse as sp
    L = _lib.lib()
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([200, 256, 300, 511, 512, 1000, 4096, 5000, 33333]))
    per_row = float(rng.choice([0.5, 2.0, 5.0, 7.0, 9.0]))
    k = rng.poisson(per_row, size=n)                            # row lengths; duplicates are merged below
    rows = np.repeat(np.arange(n), k)
    M = sp.coo_matrix((rng.standard_normal(rows.size), (rows, rng.integers(0, n, size=rows.size))), shape=(n, n)).tocsr()
    if seed % 3 == 0:
        M = M + sp.eye(n, format="csr")
    M = M.tocsr()
    M.sum_duplicates()
    M.sort_indices()
    data, indices, indptr = M.data.astype(np.float64), M.indices.astype(np.int32), M.indptr.astype(np.int32)
    if seed % 2 == 1:                                           # unsorted columns: reverse every row
        for i in range(n):
            data[indptr[i]:indptr[i + 1]] = data[indptr[i]:indptr[i + 1]][::-1]
            indices[indptr[i]:indptr[i + 1]] = indices[indptr[i]:indptr[i + 1]][::-1]
    sysm = problems.LinearSystem(data, indices, indptr, np.zeros(n), np.zeros(n), "rnd")
    x, y0 = rng.standard_normal(n), rng.standard_normal(n)
    ref = oracle.spmv(sysm.Elements, sysm.ColumnIndeces, sysm.RowOffsets, x)
    A = DeviceCsr(sysm)
    for grid in (0, 8, 64):
        assert np.array_equal(A.spmv(h, x, kernel=10, tuning=(64, 0, grid)), ref), (n, per_row, grid)
    assert np.array_equal(A.spmv(h, x, alpha=0.5, beta=-2.0, y0=y0, kernel=10), 0.5 * ref + (-2.0) * y0)
    if sysm.nnz >= 8:
        vx, vy = dvec(x), dvec(np.zeros(n))
        L.MgcgSetSpmvKernel(h.sparse, 10)
        got = L.CsrMVDot(h.blas, h.sparse, vy.ToRawPtr(), A.e.ToRawPtr(), A.r.ToRawPtr(), A.c.ToRawPtr(), vx.ToRawPtr(), vx.ToRawPtr(), sysm.nnz, n, n)
        L.MgcgSetSpmvKernel(h.sparse, 0)
        _lib.check("CsrMVDot")
        assert np.array_equal(vy.to_numpy(n), ref)
        want = float(np.dot(x, ref))
        assert abs(got - want) <= 1e-12 * max(1.0, float(np.abs(x * ref).sum()))


def test_csrmv_alpha_beta_and_edges(h, oracle):
    s = problems.poisson(9, 8, 7)
    rng = np.random.default_rng(11)
    x, y0 = rng.standard_normal(s.Count), rng.standard_normal(s.Count)
    Ax = oracle.spmv(s.Elements, s.ColumnIndeces, s.RowOffsets, x)
    A = DeviceCsr(s)
    assert np.array_equal(A.spmv(h, x, alpha=2.5, beta=0.0, kernel=1), 2.5 * Ax)
    assert np.array_equal(A.spmv(h, x, alpha=2.5, beta=-0.75, y0=y0, kernel=1), 2.5 * Ax + (-0.75) * y0)
    # beta == 0 must not read y (NaN in y does not propagate), as cusparse csrmv
    assert np.array_equal(A.spmv(h, x, alpha=1.0, beta=0.0, y0=np.full(s.Count, np.nan), kernel=1), Ax)
    # matrix with empty rows, an empty matrix, and a single row
    import scipy.sparse as sp
    M = sp.random(400, 400, density=0.004, random_state=3, format="csr")
    M.sort_indices()
    sysm = problems.LinearSystem(M.data.astype(np.float64), M.indices.astype(np.int32), M.indptr.astype(np.int32), np.zeros(400), np.zeros(400), "holes")
    xv = rng.standard_normal(400)
    for k in (1, 5, 8, 9):
        assert np.allclose(DeviceCsr(sysm).spmv(h, xv, kernel=k), oracle.spmv(sysm.Elements, sysm.ColumnIndeces, sysm.RowOffsets, xv), rtol=1e-13, atol=1e-15)
    assert np.array_equal(DeviceCsr(sysm).spmv(h, xv, kernel=1), oracle.spmv(sysm.Elements, sysm.ColumnIndeces, sysm.RowOffsets, xv))
    assert np.array_equal(DeviceCsr(sysm).spmv(h, xv, kernel=9), oracle.spmv(sysm.Elements, sysm.ColumnIndeces, sysm.RowOffsets, xv))
    assert np.array_equal(A.spmv(h, x, alpha=2.5, beta=-0.75, y0=y0, kernel=9), 2.5 * Ax + (-0.75) * y0)
    empty = problems.LinearSystem(np.zeros(0), np.zeros(0, np.int32), np.zeros(6, np.int32), np.zeros(5), np.zeros(5), "empty")
    assert np.array_equal(DeviceCsr(empty).spmv(h, np.ones(5), kernel=1, y0=np.full(5, 7.0)), np.zeros(5))


def test_csrmv_unaligned_subarrays(h, oracle):
    """Raw pointers offset by an odd number of elements (a caller slicing its own arrays)."""
    L = _lib.lib()
    s = problems.poisson(13, 11, 5)
    n, nnz = s.Count, s.nnz
    e = dvec(np.concatenate([[9.0], s.Elements[:nnz]]))
    c = ivec(np.concatenate([[0], s.ColumnIndeces[:nnz]]))
    r = ivec(s.RowOffsets)
    x = np.cos(np.arange(n) * 0.3)
    vx, vy = dvec(np.concatenate([[5.0], x])), dvec(np.zeros(n + 1))
    L.MgcgSetSpmvKernel(h.sparse, 1)
    L.CsrMV(h.sparse, h.descr, vy.ToRawPtr() + 8, e.ToRawPtr() + 8, r.ToRawPtr(), c.ToRawPtr() + 4, vx.ToRawPtr() + 8, nnz, n, n, 1.0, 0.0)
    _lib.check("CsrMV")
    L.MgcgSetSpmvKernel(h.sparse, 0)
    assert np.array_equal(vy.to_numpy(n + 1)[1:], oracle.spmv(s.Elements, s.ColumnIndeces, s.RowOffsets, x))


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 1001, 4097, 300001])
def test_blas1(h, oracle, n):
    L = _lib.lib()
    rng = np.random.default_rng(n)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    vx, vy = dvec(x), dvec(y)
    L.Axpy(h.blas, vy.ToRawPtr(), vx.ToRawPtr(), n, 0.37)                 # y += a x  (bit-exact: mul then add)
    assert np.array_equal(vy.to_numpy(n), oracle.set_added(y, x, 0.37))
    L.Scal(h.blas, vx.ToRawPtr(), -1.25, n)
    assert np.array_equal(vx.to_numpy(n), -1.25 * x)
    vx2, vy2 = dvec(x), dvec(y)
    L.Xpay(h.blas, vy2.ToRawPtr(), vx2.ToRawPtr(), n, 0.6)                # y = x + b y
    assert np.array_equal(vy2.to_numpy(n), oracle.set_added(x, y, 0.6))
    d = L.Dot(h.blas, vy2.ToRawPtr(), vx2.ToRawPtr(), n)
    ref = oracle.dot(vy2.to_numpy(n), x)
    assert abs(d - ref) <= 1e-13 * np.abs(vy2.to_numpy(n) * x).sum() + 1e-300
    assert d == L.Dot(h.blas, vy2.ToRawPtr(), vx2.ToRawPtr(), n)         # run-to-run reproducible
    assert L.NrmInf(h.blas, vx2.ToRawPtr(), n) == oracle.max_absolute(x)
    # Copy with element offsets (Mgcg.cu:49-54)
    if n >= 8:
        vz = dvec(np.zeros(n))
        L.Copy(h.blas, vz.ToRawPtr(), vx2.ToRawPtr(), n - 5, 3, 2)
        out = vz.to_numpy(n)
        assert np.array_equal(out[3:n - 2], x[2:n - 3]) and np.all(out[:3] == 0)
        # odd offsets => 8-byte aligned only: scalar path of axpy/dot
        L.Axpy(h.blas, vz.ToRawPtr() + 8, vx2.ToRawPtr() + 24, n - 4, 2.0)
        exp = out.copy()
        exp[1:n - 3] = exp[1:n - 3] + 2.0 * x[3:n - 1]
        assert np.array_equal(vz.to_numpy(n), exp)
    _lib.check("blas1")


def test_vectors_and_runtime(h):
    L = _lib.lib()
    assert L.GetDeviceCount() >= 1
    v = L.Create_Double(10)
    out = np.ones(10)
    L.CopyToArray_Double(v, out.ctypes.data_as(C.c_void_p), 10, 0, 0)
    assert np.all(out == 0)                                              # device_vector(size) is zero-initialised
    src = np.arange(10, dtype=np.float64)
    L.CopyFromArray_Double(v, src.ctypes.data_as(C.c_void_p), 4, 3, 5)   # src[3:7] -> v[5:9]
    L.CopyToArray_Double(v, out.ctypes.data_as(C.c_void_p), 10, 0, 0)
    assert list(out) == [0, 0, 0, 0, 0, 3, 4, 5, 6, 0]
    w = L.Create_Double(10)
    L.CopyFromDevice_Double(L.ToRawPtr_Double(v), L.ToRawPtr_Double(w), 3, 5, 1)  # count in ELEMENTS (reference bug fixed)
    L.CopyToArray_Double(w, out.ctypes.data_as(C.c_void_p), 10, 0, 0)
    assert list(out) == [0, 3, 4, 5, 0, 0, 0, 0, 0, 0]
    L.CopyToArray_Double(v, out.ctypes.data_as(C.c_void_p), 4, 8, 0)      # out of range -> error, no crash
    assert "outside vector" in _lib.last_error()
    L.MgcgClearLastError()
    L.Delete_Double(v)
    L.Delete_Double(w)
    iv = L.Create_Int(5)
    isrc = np.array([5, 4, 3, 2, 1], dtype=np.int32)
    L.CopyFromArray_Int(iv, isrc.ctypes.data_as(C.c_void_p), 5, 0, 0)
    iout = np.zeros(5, dtype=np.int32)
    L.CopyToArray_Int(iv, iout.ctypes.data_as(C.c_void_p), 5, 0, 0)
    assert list(iout) == [5, 4, 3, 2, 1]
    L.Delete_Int(iv)


def test_device_generator_matches_host(h):
    L = _lib.lib()
    for (nx, ny, nz) in [(8, 6, 4), (5, 7, 1), (16, 16, 16), (3, 2, 2)]:
        s = problems.poisson(nx, ny, nz)
        for (z0, z1) in [(0, nz), (0, max(1, nz // 2)), (nz // 2, nz)]:
            if z1 <= z0:
                continue
            nnz = L.MgcgPoissonNnz(nx, ny, nz, z0, z1)
            rows = (z1 - z0) * nx * ny
            lo, hi = s.RowOffsets[z0 * nx * ny], s.RowOffsets[z1 * nx * ny]
            assert nnz == hi - lo
            e, c, r = dvec(np.zeros(nnz)), ivec(np.zeros(nnz, np.int32)), ivec(np.zeros(rows + 1, np.int32))
            assert L.MgcgGeneratePoisson(e.Ptr, r.Ptr, c.Ptr, nx, ny, nz, z0, z1) == 0
            assert np.array_equal(e.to_numpy(nnz), s.Elements[lo:hi])
            assert np.array_equal(c.to_numpy(nnz), s.ColumnIndeces[lo:hi])
            assert np.array_equal(r.to_numpy(rows + 1), s.RowOffsets[z0 * nx * ny: z1 * nx * ny + 1] - lo)
            mn, mx = C.c_int(), C.c_int()
            assert L.MgcgMinMaxColumn(c.Ptr, nnz, C.byref(mn), C.byref(mx)) == 0
            assert (mn.value, mx.value) == (int(s.ColumnIndeces[lo:hi].min()), int(s.ColumnIndeces[lo:hi].max()))
