"""Multigrid-preconditioned CG ("MGCG") on one GPU through the C ABI (MgSetup / MgApply / SolveMg).

The reference names itself MGCG (Mgcg/cuBlas/Mgcg/MgcgMain.cs:8) but never implemented the
preconditioner; DESIGN.md section 6 defines the one built here (cell-centred geometric hierarchy,
piecewise-constant transfer, Galerkin coarse operators scaled by 1/2, weighted-Jacobi V(nu,nu)).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import MgcgError, check, lib
from .solver import ApplicationException, ConjugateGradientSingleGpu, VectorDouble, _ptr


class ConjugateGradientMgGpu(ConjugateGradientSingleGpu):
    """ConjugateGradientSingleGpu plus a V-cycle preconditioner built from ``grid = (nx, ny, nz)``."""

    def __init__(self, count, maxNonZeroCount, _minIteration, _maxIteration, allowableResidual, grid,
                 levels: int = 3, omega: float | None = None, nu: int = 1, nuCoarse: int = 4, sigma: float = 0.5,
                 rule=_lib.RULE_CSHARP, interpolation: int = 0):
        super().__init__(count, maxNonZeroCount, _minIteration, _maxIteration, allowableResidual, rule=rule)
        self.interpolation = int(interpolation)          # 0: piecewise constant, 1: cell-centred linear (MgSetInterpolation)
        self.grid = tuple(int(g) for g in grid)
        nx, ny, nz = self.grid
        if nx * ny * nz != count:
            raise MgcgError("grid does not match count")
        self.levels_requested = levels
        self.omega = (6.0 / 7.0 if nz > 1 else 4.0 / 5.0) if omega is None else float(omega)
        self.nu, self.nuCoarse, self.sigma = int(nu), int(nuCoarse), float(sigma)
        self.vectorZ = VectorDouble(count)
        self.mg = None

    def Dispose(self):
        if getattr(self, "mg", None):
            lib().MgDestroy(self.mg)
            self.mg = None
        if getattr(self, "vectorZ", None) is not None:
            self.vectorZ.Dispose()
        super().Dispose()

    def Initialize(self):
        super().Initialize()
        self.Setup()

    def InitializePoisson(self, b_value: float = 1.0, x_value: float = 0.0):
        """Generate the 5/7-point Poisson matrix of ``self.grid`` directly in HBM (no host arrays) and build
        the hierarchy; the 512^3 system is 11.8 GB and is never materialised on the host."""
        L = lib()
        nx, ny, nz = self.grid
        nnz = L.MgcgPoissonNnz(nx, ny, nz, 0, nz)
        if self.vectorA.size < nnz:
            raise MgcgError("construct with maxNonZeroCount >= 7 for the Poisson generator")
        if L.MgcgGeneratePoisson(self.vectorA.Ptr, self.vectorRowOffsets.Ptr, self.vectorColumnIndeces.Ptr, nx, ny, nz, 0, nz) != 0:
            check("MgcgGeneratePoisson")
        L.MgcgFill(self.vectorB.Ptr, b_value)
        L.MgcgFill(self.vectorX.Ptr, x_value)
        self.A = None
        self._nnz = int(nnz)
        self.Setup()

    def Setup(self):
        nx, ny, nz = self.grid
        nonzeroCount = int(self.A.RowOffsets[self.Count]) if self.A is not None else self._nnz
        if self.mg:
            lib().MgDestroy(self.mg)
        self.mg = lib().MgSetup(self.cublas, self.cusparse, self.vectorA.Ptr, self.vectorRowOffsets.Ptr, self.vectorColumnIndeces.Ptr,
                                nonzeroCount, nx, ny, nz, self.levels_requested, self.omega, self.nu, self.nuCoarse, self.sigma)
        check("MgSetup")
        if not self.mg:
            raise MgcgError("MgSetup returned NULL")
        if self.interpolation and lib().MgSetInterpolation(self.mg, self.interpolation) != 0:
            check("MgSetInterpolation")
        self.levels = lib().MgLevels(self.mg)

    def level_csr(self, l: int):
        n, nnz = lib().MgLevelRows(self.mg, l), lib().MgLevelNnz(self.mg, l)
        e, c, r = np.empty(nnz), np.empty(nnz, dtype=np.int32), np.empty(n + 1, dtype=np.int32)
        lib().MgLevelCopyCsr(self.mg, l, _ptr(e), _ptr(c), _ptr(r))
        check("MgLevelCopyCsr")
        return e, c, r

    def level_dinv(self, l: int):
        d = np.empty(lib().MgLevelRows(self.mg, l))
        lib().MgLevelCopyDinv(self.mg, l, _ptr(d))
        check("MgLevelCopyDinv")
        return d

    def Apply(self, r: np.ndarray) -> np.ndarray:
        """z = M^-1 r for a host vector (test helper)."""
        vr, vz = VectorDouble(self.Count), VectorDouble(self.Count)
        vr.CopyFrom(np.ascontiguousarray(r, dtype=np.float64), self.Count)
        lib().MgApply(self.mg, vr.ToRawPtr(), vz.ToRawPtr())
        check("MgApply")
        z = vz.to_numpy()
        vr.Dispose()
        vz.Dispose()
        return z

    def Solve(self, trace: bool = False):
        nonzeroCount = int(self.A.RowOffsets[self.Count]) if self.A is not None else self._nnz
        iteration, residual = C.c_int(0), C.c_double(0.0)
        cap = max(self.MaxIteration, self.MinIteration) + 8 if trace else 0
        tr = np.zeros(max(cap, 1)) if trace else None
        rule = _lib.RULE_CSHARP if self.rule is None else self.rule
        st = lib().SolveMg(self.cublas, self.cusparse, self.matDescr, self.mg,
                           self.vectorA.Ptr, self.vectorRowOffsets.Ptr, self.vectorColumnIndeces.Ptr,
                           self.vectorX.Ptr, self.vectorB.Ptr, self.vectorAp.Ptr, self.vectorP.Ptr, self.vectorR.Ptr, self.vectorZ.Ptr,
                           nonzeroCount, self.Count, self.AllowableResidual, self.MinIteration, self.MaxIteration, rule,
                           C.byref(iteration), C.byref(residual), _ptr(tr) if trace else None, cap)
        self.Iteration, self.Residual, self.status = iteration.value, residual.value, st
        if trace:
            self.trace = tr[: self.Iteration + 1].copy()
        if st == _lib.MAXIT_EXCEEDED:
            lib().MgcgClearLastError()
            raise ApplicationException(f"MGCG did not converge within MaxIteration={self.MaxIteration}")
        if st != _lib.OK:
            check("SolveMg")
            raise MgcgError(f"SolveMg failed with status {st}")
