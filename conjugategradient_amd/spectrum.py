"""Spectral diagnostics.

* ``GetEigenValues`` -- the reference's dense Jacobi-rotation routine (Mgcg/HandmadeCL/MgcgCL/SparseMatrix.cs:234-372)
  restated on the host for the slot-0-diagonal ELL builder: the lower triangle of the stored matrix is mirrored into a
  dense symmetric array, the largest off-diagonal element is rotated away until none exceeds ``residual`` (or the pivot
  stops moving, ``:300``).  The reference never reads its ``maxIteration`` argument; here it caps the rotations.
* ``EstimateSpectrum`` -- the scalable device counterpart (``MgcgEstimateSpectrum``: Lanczos on the SpMV / dot kernels).
* ``jacobi_omega`` -- damping of the weighted-Jacobi smoother from the largest eigenvalue of D^-1 A.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import MgcgError, lib


def GetEigenValues(A, maxIteration: int, residual: float) -> np.ndarray:
    """Eigenvalues (unsorted: the rotated diagonal) of the symmetric matrix held by an ``EllSparseMatrix``."""
    n, K = A.RowCount, A.MaxNonzeroCountPerRow
    M = np.zeros((n, n))
    for i in range(n):                                           # :238-262: diagonal + entries with i >= j, mirrored
        M[i, i] = A.Elements[i * K]
        for k in range(1, int(A.NonzeroCounts[i])):
            j = int(A.ColumnIndeces[i * K + k])
            if i >= j:
                M[i, j] = M[j, i] = A.Elements[i * K + k]
    p = q = 0
    for _ in range(max(int(maxIteration), 0)):
        oldP, oldQ = p, q
        off = np.abs(M)
        np.fill_diagonal(off, 0.0)
        flat = int(np.argmax(off))                               # first largest in row-major order, like the `>` scan (:277-297)
        maxValue = residual / 10
        if off.flat[flat] > maxValue:
            p, q = divmod(flat, n)
            maxValue = off.flat[flat]
        if (p == oldP and q == oldQ) or maxValue < residual:     # :300
            break
        alpha = (M[p, p] - M[q, q]) / 2
        beta = -M[p, q]
        gamma = abs(alpha) / np.sqrt(alpha * alpha + beta * beta)
        cos = np.sqrt((1 + gamma) / 2)
        sin = np.sqrt((1 - gamma) / 2) * np.sign(alpha * beta)
        a_pp, a_pq, a_qq = M[p, p], M[p, q], M[q, q]
        rp, rq = M[p, :].copy(), M[q, :].copy()
        M[p, :] = rp * cos - rq * sin
        M[q, :] = rp * sin + rq * cos
        cp, cq = M[:, p].copy(), M[:, q].copy()
        M[:, p] = cp * cos - cq * sin
        M[:, q] = cp * sin + cq * cos
        M[p, p] = cos * (a_pp * cos - a_pq * sin) - sin * (a_pq * cos - a_qq * sin)      # :346-349
        M[p, q] = sin * (a_pp * cos - a_pq * sin) + cos * (a_pq * cos - a_qq * sin)
        M[q, p] = M[p, q]
        M[q, q] = sin * (a_pp * sin + a_pq * cos) + cos * (a_pq * sin + a_qq * cos)
    return np.diag(M).copy()


def EstimateSpectrum(solver, jacobiScaled: bool = False, steps: int = 40, seed: int = 1, all_ritz: bool = False):
    """(lambdaMin, lambdaMax[, ritz]) of the matrix a single-GPU solver object has uploaded (after ``Initialize``)."""
    _lib.require_gpu()
    nnz = int(solver.A.RowOffsets[solver.Count]) if hasattr(solver, "A") and solver.A is not None and hasattr(solver.A, "RowOffsets") \
        else int(solver.nnz)
    lo, hi, done = C.c_double(0), C.c_double(0), C.c_int(0)
    ritz = np.zeros(max(steps, 1))
    vE = getattr(solver, "vectorA", None) or solver.vectorElements
    st = lib().MgcgEstimateSpectrum(solver.cublas, solver.cusparse, vE.Ptr, solver.vectorRowOffsets.Ptr, solver.vectorColumnIndeces.Ptr,
                                    nnz, solver.Count, 1 if jacobiScaled else 0, int(steps), int(seed),
                                    C.byref(lo), C.byref(hi), ritz.ctypes.data_as(C.c_void_p), C.byref(done))
    if st != _lib.OK:
        msg = _lib.last_error()
        lib().MgcgClearLastError()
        raise MgcgError(msg or "MgcgEstimateSpectrum failed")
    if all_ritz:
        return lo.value, hi.value, ritz[: done.value].copy()
    return lo.value, hi.value


def jacobi_omega(lambdaMaxDinvA: float, dim: int | None = None) -> float:
    """Damping of x += omega D^-1 (b - A x).  For the constant-coefficient Laplacian the smoothing optimum over the
    high-frequency band [lambdaMax/(2 dim), lambdaMax] of D^-1 A is 2 / (lambdaMax (1 + 1/(2 dim))): 6/7 in 3-D and 4/5
    in 2-D at lambdaMax = 2 -- the values the V-cycle uses.  Without a dimension: 4 / (3 lambdaMax)."""
    if dim is None:
        return 4.0 / (3.0 * lambdaMaxDinvA)
    return 2.0 / (lambdaMaxDinvA * (1.0 + 1.0 / (2.0 * dim)))
