"""Linear systems the reference hard-codes, plus the BASELINE.json stencil configs.

Pure numpy, host side only: these build the (A, x0, b) triples that a caller of
the reference would hand to ``Initialize()``.  The CSR arrays use the
reference's field names (``Elements``, ``ColumnIndeces``, ``RowOffsets`` --
Mgcg/cuBlas/Mgcg/SparseMatrix.cs:13-23), fp64 values and 0-based int32 indices.

The large structured configs (7-point 512^3) are generated directly in HBM by
the device generator (``MgcgGeneratePoisson`` in include/MgcgGpu.h);
``poisson()`` here is its host twin for small grids and for tests.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class LinearSystem:
    """A, x0, b with the reference's member names (LinerEquations.cs:6-47)."""

    Elements: np.ndarray        # float64[nnz]
    ColumnIndeces: np.ndarray   # int32[nnz]
    RowOffsets: np.ndarray      # int32[Count+1]
    x: np.ndarray               # float64[Count]  initial guess
    b: np.ndarray               # float64[Count]
    name: str = ""
    grid: tuple | None = None   # (nx, ny, nz) for structured problems

    @property
    def Count(self) -> int:
        return int(self.x.shape[0])

    @property
    def nnz(self) -> int:
        return int(self.RowOffsets[-1])

    def to_scipy(self):
        import scipy.sparse as sp

        return sp.csr_matrix(
            (self.Elements[: self.nnz], self.ColumnIndeces[: self.nnz], self.RowOffsets),
            shape=(self.Count, self.Count),
        )


def tridiagonal(n: int) -> LinearSystem:
    """[1 2 1] central-difference matrix of
    SimpleConjugateGradient/SimpleConjugateGradient/SimpleConjugateGradient.cu:139-197
    (and SimpleConjugateGradientCpu.cpp:40-105): each row stores diagonal, left,
    right in that order; ``b[i] = i * i * 0.5`` with ``i * i`` evaluated in
    32-bit ``int`` exactly as the C++ does (it wraps for i > 46340 at N=65536);
    x0 = 0.
    """
    i = np.arange(n, dtype=np.int64)
    has_l = i > 0
    has_r = i < n - 1
    cnt = 1 + has_l.astype(np.int64) + has_r.astype(np.int64)
    ro = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(cnt, out=ro[1:])
    nnz = int(ro[-1])
    col = np.empty(nnz, dtype=np.int32)
    val = np.empty(nnz, dtype=np.float64)
    col[ro[:-1]] = i
    val[ro[:-1]] = 2.0
    col[ro[:-1][has_l] + 1] = i[has_l] - 1
    val[ro[:-1][has_l] + 1] = 1.0
    pos_r = ro[:-1] + 1 + has_l
    col[pos_r[has_r]] = i[has_r] + 1
    val[pos_r[has_r]] = 1.0
    ii = (i * i).astype(np.int32)  # int overflow as in the C++ expression
    b = ii.astype(np.float64) * 0.5
    return LinearSystem(val, col, ro.astype(np.int32), np.zeros(n), b, f"tridiagonal{n}")


def mgcg_main(count: int = 34567 * 6, max_nonzero: int = 160, x0_div: float = 100.0) -> LinearSystem:
    """The driver system of Mgcg/cuBlas/Mgcg/MgcgMain.cs:51-104.

    Row i stores the diagonal FIRST (columns are not sorted), then every
    ``j in [max(0, i-max_nonzero/2+1), min(count, i+max_nonzero/2))``, ``j != i``
    ascending, with ``a_ij = |sin(i+j)|``; the diagonal is the sum of the row's
    off-diagonals accumulated left to right.  ``b_i = 10 cos(i)``, ``x0_i = i/100``.
    ``mgcg_main(21, 6, 10.0)`` is the dense system of R/CG.R:1-24.
    """
    half = max_nonzero // 2
    i = np.arange(count, dtype=np.int64)
    val, col, ro = _sin_band(count, np.maximum(0, i - half + 1), np.minimum(count, i + half), np.zeros(count))
    b = np.cos(i.astype(np.float64)) * 10.0
    x0 = i.astype(np.float64) / x0_div
    return LinearSystem(val, col, ro, x0, b, f"mgcgmain{count}")


def _sin_band(count: int, jlo: np.ndarray, jhi: np.ndarray, diag0: np.ndarray):
    """Rows ``i`` with the diagonal FIRST, then every ``j in [jlo_i, jhi_i)``, ``j != i`` ascending, ``a_ij = |sin(i+j)|``;
    diagonal = ``diag0_i`` plus the row's off-diagonals added left to right (MgcgMain.cs:79, MgcgCL.cs:33-40)."""
    i = np.arange(count, dtype=np.int64)
    cnt = jhi - jlo  # includes the diagonal's own slot (moved to the front)
    ro = np.zeros(count + 1, dtype=np.int64)
    np.cumsum(cnt, out=ro[1:])
    nnz = int(ro[-1])
    row = np.repeat(i, cnt)
    k = np.arange(nnz, dtype=np.int64) - np.repeat(ro[:-1], cnt)  # position within row
    # position 0 -> diagonal; position q>=1 -> q-th off-diagonal in ascending j
    j = np.repeat(jlo, cnt) + (k - 1)
    j = np.where(j >= row, j + 1, j)  # skip the diagonal
    col = np.where(k == 0, row, j).astype(np.int32)
    val = np.abs(np.sin((row + col).astype(np.float64)))
    val[ro[:-1]] = 0.0
    diag = np.array(diag0, dtype=np.float64)
    for q in range(int(cnt.max()) - 1):
        sel = (cnt - 1) > q
        idx = ro[:-1][sel] + 1 + q
        diag[sel] = diag[sel] + val[idx]
    val[ro[:-1]] = diag
    return val, col, ro.astype(np.int32)


def viennacl_main(n: int = 34567 * 5, band_width: int = 160) -> LinearSystem:
    """The driver system of Mgcg/ViennaCL/MgcgCL/MgcgCL.cs:14-61: ``A[i,i] = i`` is assigned first (so the dictionary row
    starts with the diagonal), then ``j in [max(0, i-band/2), min(N-1, i+band/2)]`` INCLUSIVE, ``j != i`` ascending with
    ``a_ij = |sin(i+j)|`` added onto the diagonal; ``b_i = asin(i/N)``, ``x0 = 0``; solved to 1e-4 RELATIVE."""
    half = band_width // 2
    i = np.arange(n, dtype=np.int64)
    val, col, ro = _sin_band(n, np.maximum(0, i - half), np.minimum(n - 1, i + half) + 1, i.astype(np.float64))
    # b through libm's asin, the function the C++ twin (host/MgcgCLMain.cpp) and the reference's Math.Asin call: numpy's own vectorised
    # arcsin differs from it in the last bit for ~8 % of these arguments, which would show in every bit-for-bit comparison of x
    import math
    b = np.array([math.asin(v) for v in (i.astype(np.float64) / n)])
    return LinearSystem(val, col, ro, np.zeros(n), b, f"viennaclmain{n}")


def poisson(nx: int, ny: int, nz: int = 1) -> LinearSystem:
    """5-point (nz == 1) / 7-point Poisson, Dirichlet, lexicographic x-fastest:
    diagonal 2*dim, off-diagonals -1, columns ascending (SURVEY.md section 8,
    configs 1-4).  b = 1, x0 = 0.
    """
    n = nx * ny * nz
    assert n < 2**31
    idx = np.arange(n, dtype=np.int64)
    x = idx % nx
    y = (idx // nx) % ny
    z = idx // (nx * ny)
    diag = 6.0 if nz > 1 else 4.0
    # neighbour presence in ascending column order
    masks = [z > 0, y > 0, x > 0, np.ones(n, bool), x < nx - 1, y < ny - 1, z < nz - 1]
    offs = [-nx * ny, -nx, -1, 0, 1, nx, nx * ny]
    vals = [-1.0, -1.0, -1.0, diag, -1.0, -1.0, -1.0]
    cnt = np.zeros(n, dtype=np.int64)
    for m in masks:
        cnt += m
    ro = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(cnt, out=ro[1:])
    nnz = int(ro[-1])
    col = np.empty(nnz, dtype=np.int32)
    val = np.empty(nnz, dtype=np.float64)
    pos = ro[:-1].copy()
    for m, o, v in zip(masks, offs, vals):
        p = pos[m]
        col[p] = idx[m] + o
        val[p] = v
        pos += m
    return LinearSystem(val, col, ro.astype(np.int32), np.zeros(n), np.ones(n),
                        f"poisson{nx}x{ny}x{nz}", grid=(nx, ny, nz))


def poisson_nnz(nx: int, ny: int, nz: int = 1) -> int:
    n = nx * ny * nz
    nnz = n + 2 * (nx - 1) * ny * nz + 2 * nx * (ny - 1) * nz
    if nz > 1:
        nnz += 2 * nx * ny * (nz - 1)
    return nnz


def random_spd(n: int, mean_upper: float = 14.0, seed: int = 12345, sort_columns: bool = True) -> LinearSystem:
    """Config 5 of BASELINE.json: per row k ~ Poisson(mean_upper)+1
    strictly-upper random columns, values U(-1,0); A = U + U^T; diagonal =
    1 + sum|off-diag| (strictly diagonally dominant, hence SPD); b = A.1, x0 = 0.
    """
    import scipy.sparse as sp

    rng = np.random.default_rng(seed)
    k = rng.poisson(mean_upper, size=n) + 1
    rows = np.repeat(np.arange(n, dtype=np.int64), k)
    # strictly-upper columns, uniform over (row, n); the last row has none
    span = n - 1 - rows
    cols = rows + 1 + np.floor(rng.random(rows.shape[0]) * np.maximum(span, 1)).astype(np.int64)
    keep = span > 0
    rows, cols = rows[keep], np.minimum(cols[keep], n - 1)
    vals = -rng.random(rows.shape[0])
    U = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    U.sum_duplicates()
    A = (U + U.T).tocsr()
    d = 1.0 + np.asarray(abs(A).sum(axis=1)).ravel()
    A = (A + sp.diags(d)).tocsr()
    A.sort_indices()
    if not sort_columns:
        # deterministic in-row shuffle: reverse every row
        ro = A.indptr
        perm = np.concatenate([np.arange(ro[i + 1] - 1, ro[i] - 1, -1) for i in range(n)]) if n < 200000 else None
        if perm is not None:
            A = sp.csr_matrix((A.data[perm], A.indices[perm], ro), shape=(n, n))
    b = A @ np.ones(n)
    return LinearSystem(np.ascontiguousarray(A.data, dtype=np.float64),
                        np.ascontiguousarray(A.indices, dtype=np.int32),
                        np.ascontiguousarray(A.indptr, dtype=np.int32),
                        np.zeros(n), b, f"random_spd{n}")


def partition_offsets(count: int, device_count: int, row_offsets=None, balance: str = "rows") -> list[int]:
    """Row-range partition of Mgcg/cuBlas/Mgcg/ConjugateGradientParallelGpu.cs:271-277:
    floor(count/device_count) rows each, the last device takes the remainder.
    balance="nnz" (not in the reference; needs row_offsets): row ranges of equal NONZERO count instead -- rank r starts at the first
    row whose offset reaches r/devices of the nonzeros.  For a matrix with uneven rows (BASELINE config 5: the last of eight
    row-count slabs holds 3.7 times the nonzeros of the first) the slowest rank sets the iteration time."""
    import math

    if balance == "nnz":
        if row_offsets is None:
            raise ValueError("balance='nnz' needs the row offsets")
        ro = np.asarray(row_offsets)
        nnz = int(ro[count]) - int(ro[0])
        targets = int(ro[0]) + (np.arange(1, device_count, dtype=np.int64) * nnz) // device_count
        cuts = np.searchsorted(ro[: count + 1], targets, side="left")
        off = [0] + [int(min(max(c, 0), count)) for c in cuts] + [count]
        for i in range(1, device_count + 1):                  # monotone (empty ranks only when there are fewer rows than ranks)
            off[i] = max(off[i], off[i - 1])
        return off
    if balance != "rows":
        raise ValueError("balance must be 'rows' or 'nnz'")
    off = [0] * (device_count + 1)
    for i in range(1, device_count):
        off[i] = off[i - 1] + int(math.floor(float(count) / device_count))
    off[device_count] = count
    return off
