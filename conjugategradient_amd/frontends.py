"""The two other solver front-ends of the reference, on the same HIP library:

* ``ComputerGpu`` -- Mgcg/ViennaCL/Mgcg/ComputerGpu.hpp:68-100 (``Write / Solve(residual, min, max) / Read / Iteration``):
  relative stop rule ``minIteration < it && rrNew/rr0 < residual^2`` (ComputerGpu.cpp:78), unsigned-int CSR indices.
* ``ConjugateGradientCLGpu`` -- Mgcg/HandmadeCL/MgcgCL/ConjugateGradientSingleGpu.cs: the max-norm residual
  (``ReductionMaxAbsolute``, Mgcg.cl:110-159; ConjugateGradientSingleGpu.cs:268) with the C# IsConverged rule.

Both are thin: the arithmetic is ``SolveEx`` with MGCG_RULE_VIENNACL / MGCG_RULE_HANDMADECL.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import MgcgError, check, lib
from .solver import ApplicationException, ConjugateGradientSingleGpu, VectorDouble, VectorInt, _ptr


class ComputerGpu:
    """ViennaCL front-end (ComputerGpu.hpp:68-100)."""

    def __init__(self, n: int):
        _lib.require_gpu()
        L = lib()
        self.count = int(n)
        self.cublas, self.cusparse, self.matDescr = L.CreateBlas(), L.CreateSparse(), L.CreateMatDescr()
        check("ComputerGpu")
        self.vx, self.vb, self.vAp, self.vp, self.vr = (VectorDouble(n) for _ in range(5))
        self.vE = self.vC = self.vRO = None
        self.nnz = 0
        self.iteration = 0

    def Write(self, elements, rowOffsets, columnIndeces, x, b):
        """SetMatrix + SetVector (ComputerGpu.cpp:18-45): CSR with unsigned-int indices, x is the initial guess."""
        e = np.ascontiguousarray(elements, dtype=np.float64)
        ro = np.ascontiguousarray(np.asarray(rowOffsets, dtype=np.uint32).astype(np.int32))
        ci = np.ascontiguousarray(np.asarray(columnIndeces, dtype=np.uint32).astype(np.int32))
        self.nnz = int(ro[self.count])
        self.vE, self.vC, self.vRO = VectorDouble(max(self.nnz, 1)), VectorInt(max(self.nnz, 1)), VectorInt(self.count + 1)
        self.vE.CopyFrom(e, self.nnz)
        self.vC.CopyFrom(ci, self.nnz)
        self.vRO.CopyFrom(ro, self.count + 1)
        self.vx.CopyFrom(np.ascontiguousarray(x, dtype=np.float64), self.count)
        self.vb.CopyFrom(np.ascontiguousarray(b, dtype=np.float64), self.count)

    def Solve(self, residual: float, minIteration: int, maxIteration: int):
        it, res = C.c_int(0), C.c_double(0.0)
        st = lib().SolveEx(self.cublas, self.cusparse, self.matDescr, self.vE.Ptr, self.vRO.Ptr, self.vC.Ptr,
                           self.vx.Ptr, self.vb.Ptr, self.vAp.Ptr, self.vp.Ptr, self.vr.Ptr, self.nnz, self.count,
                           float(residual), int(minIteration), int(maxIteration), _lib.RULE_VIENNACL,
                           C.byref(it), C.byref(res), None, 0)
        self.iteration = it.value + 1          # ComputerGpu.cpp:66: the post-incremented loop counter
        self.relative_residual = res.value
        if st != _lib.OK:
            msg = _lib.last_error()
            lib().MgcgClearLastError()
            raise MgcgError(msg or f"SolveEx failed with status {st}")

    def Read(self, x: np.ndarray):
        self.vx.CopyTo(x, self.count)

    def Iteration(self) -> int:
        return self.iteration

    def Dispose(self):
        if getattr(self, "cublas", None):
            for v in (self.vx, self.vb, self.vAp, self.vp, self.vr, self.vE, self.vC, self.vRO):
                if v is not None:
                    v.Dispose()
            lib().DestroyBlas(self.cublas)
            lib().DestroySparse(self.cusparse)
            lib().DestroyMatDescr(self.matDescr)
            self.cublas = None

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass


class ConjugateGradientCLGpu(ConjugateGradientSingleGpu):
    """HandmadeCL front-end: same class surface as ConjugateGradientSingleGpu, Residual = max|r_i|.
    ``A`` is the slot-0-diagonal ELL builder of that family (``cg.A[i, j] = v`` as in MgcgCLMain.cs:52-83) unless a CSR
    system was handed over with ``load``; ``Initialize`` packs it to CSR in stored order (the device sums a row in the
    order the reference's ``Matrix_x_Vector`` walks its slots, Mgcg.cl:171-216)."""

    def __init__(self, count, maxNonZeroCount, _minIteration, _maxIteration, allowableResidual):
        super().__init__(count, maxNonZeroCount, _minIteration, _maxIteration, allowableResidual, rule=_lib.RULE_HANDMADECL)
        from .formats import EllSparseMatrix

        self.A = EllSparseMatrix(count, maxNonZeroCount)

    def Initialize(self):
        from .formats import EllSparseMatrix
        from .solver import SparseMatrix

        if isinstance(self.A, EllSparseMatrix):
            ell = self.A
            csr = SparseMatrix.__new__(SparseMatrix)
            csr.Elements, csr.ColumnIndeces, csr.RowOffsets = ell.to_csr()
            self.A = csr
            try:
                super().Initialize()
            finally:
                self._csr, self.A = csr, ell
        else:
            self._csr = self.A
            super().Initialize()

    def Solve(self, trace: bool = False):
        held = self.A
        self.A = self._csr
        try:
            super().Solve(trace=trace)
        finally:
            self.A = held
