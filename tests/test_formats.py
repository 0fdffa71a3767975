"""Caller-side builders of the HandmadeCL and ViennaCL drivers and their flattening to CSR.
Host logic only: runs without a GPU."""
import numpy as np
import pytest

from conjugategradient_amd import problems
from conjugategradient_amd.formats import CompressedMatrix, EllSparseMatrix


def _fill_like_the_drivers(A, count, lo_of, hi_of, diag0_of):
    """The loops of MgcgCLMain.cs:52-83 / MgcgCL.cs:31-45 through the ``A[i, j]`` indexers."""
    for i in range(count):
        A[i, i] = diag0_of(i)
        for j in range(lo_of(i), hi_of(i)):
            if i != j:
                a = abs(np.sin(float(i + j)))
                A[i, j] = a
                A[i, i] = A[i, i] + a


def test_ell_indexer_semantics():
    A = EllSparseMatrix(4, 3)
    assert A.NonzeroCounts.tolist() == [1, 1, 1, 1] and A.ColumnIndeces[::3].tolist() == [0, 1, 2, 3]
    A[1, 1] = 5.0
    A[1, 3] = 2.0
    A[1, 0] = -1.0
    assert (A[1, 1], A[1, 3], A[1, 0], A[1, 2], A[1]) == (5.0, 2.0, -1.0, 0.0, 5.0)
    assert A.NonzeroCounts[1] == 3 and A.ColumnIndeces[3:6].tolist() == [1, 3, 0]      # diagonal first, then assignment order
    A[1, 3] = 7.0                                                                          # overwrite: no new slot
    assert A.NonzeroCounts[1] == 3 and A[1, 3] == 7.0
    with pytest.raises(IndexError):
        A[1, 2] = 1.0                                                                      # SparseMatrix.cs:121-124
    A.Clear()
    assert A.NonzeroCounts.tolist() == [1, 1, 1, 1] and A[1, 1] == 0.0 and A[1, 3] == 0.0


def test_ell_builder_reproduces_the_handmadecl_driver_matrix():
    n, K = 300, 160
    A = EllSparseMatrix(n, K)
    _fill_like_the_drivers(A, n, lambda i: max(0, i - K // 2 + 1), lambda i: min(n, i + K // 2), lambda i: 0.0)
    s = problems.mgcg_main(n, K)
    e, c, ro = A.to_csr()
    assert np.array_equal(ro, s.RowOffsets) and np.array_equal(c, s.ColumnIndeces) and np.array_equal(e, s.Elements)
    x = np.cos(np.arange(n) * 0.3)
    y = np.empty(n)
    A.Multiply(y, x)
    ref = np.zeros(n)                                     # SparseMatrix.cs:200-223 literally
    for i in range(n):
        for k in range(A.NonzeroCounts[i]):
            ref[i] += A.Elements[i * K + k] * x[A.ColumnIndeces[i * K + k]]
    assert np.array_equal(y, ref)


def test_ell_from_csr_round_trip_moves_the_diagonal_first():
    s = problems.poisson(5, 4, 3)                          # sorted columns: the diagonal sits in the middle of a row
    A = EllSparseMatrix.from_csr(s.Elements, s.ColumnIndeces, s.RowOffsets)
    assert A.MaxNonzeroCountPerRow == 7
    e, c, ro = A.to_csr()
    assert np.array_equal(ro, s.RowOffsets)
    assert np.array_equal(c[ro[:-1]], np.arange(s.Count))
    assert np.allclose(s.to_scipy().toarray(), problems.LinearSystem(e, c, ro, s.x, s.b).to_scipy().toarray(), rtol=0, atol=0)
    B = EllSparseMatrix.from_csr(e, c, ro, 9)              # already diagonal-first: order kept exactly
    e2, c2, ro2 = B.to_csr()
    assert np.array_equal(e2, e) and np.array_equal(c2, c) and np.array_equal(ro2, ro)
    with pytest.raises(IndexError):
        EllSparseMatrix.from_csr(e, c, ro, 6)
    # a row without a stored diagonal gets an explicit zero
    C = EllSparseMatrix.from_csr(np.array([2.0, 3.0]), np.array([1, 1], dtype=np.int32), np.array([0, 1, 2], dtype=np.int32))
    assert (C[0, 0], C[0, 1], C[1, 1], C.NonzeroCounts.tolist()) == (0.0, 2.0, 3.0, [2, 1])


def test_dictionary_builder_reproduces_the_viennacl_driver_matrix():
    n, band = 250, 160
    A = CompressedMatrix()
    _fill_like_the_drivers(A, n, lambda i: max(0, i - band // 2), lambda i: min(n - 1, i + band // 2) + 1, lambda i: float(i))
    e, c, ro = A.to_csr()
    assert c.dtype == np.uint32 and ro.dtype == np.uint32                       # MgcgCL.cs:86-87
    s = problems.viennacl_main(n, band)
    assert np.array_equal(ro, s.RowOffsets) and np.array_equal(c, s.ColumnIndeces) and np.array_equal(e, s.Elements)
    assert A[3, 200] == 0.0 and A[n + 5, 0] == 0.0 and A[3, 3] == s.Elements[s.RowOffsets[3]]
    t = A.to_system(np.zeros(n), s.b)
    assert t.ColumnIndeces.dtype == np.int32 and np.array_equal(t.Elements, s.Elements)
    # symmetric, strictly diagonally dominant from row 1 on
    M = s.to_scipy().toarray()
    assert np.array_equal(M, M.T)


def test_dictionary_rows_appear_on_first_assignment():
    A = CompressedMatrix()
    A[2, 1] = 4.0
    assert len(A.Elements) == 3 and A[2, 1] == 4.0 and A[0, 0] == 0.0
    A[2, 1] = 5.0
    e, c, ro = A.to_csr(4)
    assert ro.tolist() == [0, 0, 0, 1, 1] and e.tolist() == [5.0] and c.tolist() == [1]


def test_dense_jacobi_rotation_eigenvalues_match_lapack():
    """GetEigenValues (HandmadeCL SparseMatrix.cs:234-372) restated on the host, against numpy's symmetric solver."""
    from conjugategradient_amd.spectrum import GetEigenValues, jacobi_omega

    s = problems.mgcg_main(24, 10)
    A = EllSparseMatrix.from_csr(s.Elements, s.ColumnIndeces, s.RowOffsets)
    ev = np.sort(GetEigenValues(A, 10_000, 1e-12))
    ref = np.linalg.eigvalsh(s.to_scipy().toarray())
    assert np.abs(ev - ref).max() <= 1e-10 * np.abs(ref).max()
    assert np.array_equal(GetEigenValues(A, 0, 1e-12), s.Elements[s.RowOffsets[:-1]])     # no rotation: the diagonal
    assert jacobi_omega(2.0, 3) == pytest.approx(6 / 7) and jacobi_omega(2.0, 2) == pytest.approx(4 / 5) and jacobi_omega(2.0) == pytest.approx(2 / 3)
