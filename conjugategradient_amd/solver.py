"""Host-side mirror of the reference's solver classes (namespace ``LWisteria.Mgcg``), calling
libMgcgGpu.so through the C ABI exactly where the C# classes P/Invoke ``MgcgGpu.dll``.

Same class names, constructor arguments, members and error behaviour as
Mgcg/cuBlas/Mgcg/{LinerEquations,ConjugateGradient,ConjugateGradientGpu,
ConjugateGradientSingleGpu,ConjugateGradientParallelGpu,SparseMatrix,VectorDouble,VectorInt}.cs.
(The compiled-language twin of this file is host/Mgcg.hpp; C# itself cannot run in this image.)

This module holds NO arithmetic: every flop happens in the HIP library.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import MgcgError, check, lib
from .problems import partition_offsets


class ApplicationException(Exception):
    """What ConjugateGradient.IsConverged throws past MaxIteration (ConjugateGradient.cs:70-74)."""


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------- device vectors
class VectorDouble:
    """VectorDouble.cs:8-113 -- IDisposable wrapper of a device double[] handle."""

    def __init__(self, size: int):
        self.Ptr = lib().MgcgCreateDouble64(int(size))
        check("Create_Double")
        if not self.Ptr:
            raise MgcgError("Create_Double returned NULL")
        self.size = int(size)

    def Dispose(self):
        if self.Ptr:
            lib().Delete_Double(self.Ptr)
            self.Ptr = None

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass

    def CopyFrom(self, array: np.ndarray, count: int, arrayOffset: int = 0, vectorOffset: int = 0):
        assert array.dtype == np.float64 and array.flags.c_contiguous
        lib().CopyFromArray_Double(self.Ptr, _ptr(array), int(count), int(arrayOffset), int(vectorOffset))
        check("CopyFromArray_Double")

    def CopyTo(self, array: np.ndarray, count: int, arrayOffset: int = 0, vectorOffset: int = 0):
        assert array.dtype == np.float64 and array.flags.c_contiguous
        lib().CopyToArray_Double(self.Ptr, _ptr(array), int(count), int(vectorOffset), int(arrayOffset))
        check("CopyToArray_Double")

    def ToRawPtr(self) -> int:
        return lib().ToRawPtr_Double(self.Ptr)

    def to_numpy(self, count: int | None = None) -> np.ndarray:
        count = self.size if count is None else count
        out = np.empty(count, dtype=np.float64)
        self.CopyTo(out, count)
        return out


class VectorInt:
    """VectorInt.cs:8-106."""

    def __init__(self, size: int):
        self.Ptr = lib().MgcgCreateInt64(int(size))
        check("Create_Int")
        if not self.Ptr:
            raise MgcgError("Create_Int returned NULL")
        self.size = int(size)

    def Dispose(self):
        if self.Ptr:
            lib().Delete_Int(self.Ptr)
            self.Ptr = None

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass

    def CopyFrom(self, array: np.ndarray, count: int, arrayOffset: int = 0, vectorOffset: int = 0):
        assert array.dtype == np.int32 and array.flags.c_contiguous
        lib().CopyFromArray_Int(self.Ptr, _ptr(array), int(count), int(arrayOffset), int(vectorOffset))
        check("CopyFromArray_Int")

    def CopyTo(self, array: np.ndarray, count: int, arrayOffset: int = 0, vectorOffset: int = 0):
        assert array.dtype == np.int32 and array.flags.c_contiguous
        lib().CopyToArray_Int(self.Ptr, _ptr(array), int(count), int(vectorOffset), int(arrayOffset))
        check("CopyToArray_Int")

    def ToRawPtr(self) -> int:
        return lib().ToRawPtr_Int(self.Ptr)

    def to_numpy(self, count: int | None = None) -> np.ndarray:
        count = self.size if count is None else count
        out = np.empty(count, dtype=np.int32)
        self.CopyTo(out, count)
        return out


# --------------------------------------------------------------------------- host containers
class SparseMatrix:
    """SparseMatrix.cs:8-101 -- CSR with capacity rowCount*maxNonzeroCountPerRow; ``Clear`` zeroes the
    offsets and values and sets every column id to -1."""

    def __init__(self, rowCount: int, maxNonzeroCountPerRow: int):
        self.Elements = np.zeros(rowCount * maxNonzeroCountPerRow, dtype=np.float64)
        self.ColumnIndeces = np.full(rowCount * maxNonzeroCountPerRow, -1, dtype=np.int32)
        self.RowOffsets = np.zeros(rowCount + 1, dtype=np.int32)

    def Clear(self):
        self.RowOffsets[:] = 0
        self.Elements[:] = 0
        self.ColumnIndeces[:] = -1

    @property
    def RowCount(self) -> int:
        return int(self.RowOffsets.shape[0])  # the reference returns RowOffsets.Length (SparseMatrix.cs:93-100)

    @classmethod
    def from_system(cls, system) -> "SparseMatrix":
        m = cls.__new__(cls)
        m.Elements = np.ascontiguousarray(system.Elements, dtype=np.float64)
        m.ColumnIndeces = np.ascontiguousarray(system.ColumnIndeces, dtype=np.int32)
        m.RowOffsets = np.ascontiguousarray(system.RowOffsets, dtype=np.int32)
        return m


class LinerEquations:
    """LinerEquations.cs:6-47 (sic)."""

    def __init__(self, count: int, maxNonZeroCount: int):
        self.A: SparseMatrix | None = None
        self.x = np.zeros(count, dtype=np.float64)
        self.b = np.zeros(count, dtype=np.float64)

    @property
    def Count(self) -> int:
        return int(self.x.shape[0])


class ConjugateGradient(LinerEquations):
    """ConjugateGradient.cs:6-84."""

    def __init__(self, count, maxNonZeroCount, _minIteration, _maxIteration, _allowableResidual):
        super().__init__(count, maxNonZeroCount)
        self.MinIteration = int(_minIteration)
        self.MaxIteration = int(_maxIteration)
        self.AllowableResidual = float(_allowableResidual)
        self.Iteration = 0
        self.Residual = 0.0

    @property
    def IsConverged(self) -> bool:
        if self.Iteration < self.MinIteration:
            return False
        elif self.Iteration > self.MaxIteration:
            raise ApplicationException("the pressure equation did not converge")  # ConjugateGradient.cs:73
        return self.Residual < self.AllowableResidual

    def Solve(self):
        raise NotImplementedError

    def load(self, system):
        """Convenience: take A, x0, b from a problems.LinearSystem."""
        self.A = SparseMatrix.from_system(system)
        self.x[:] = system.x
        self.b[:] = system.b
        return self


class ConjugateGradientGpu(ConjugateGradient):
    """ConjugateGradientGpu.cs:10-90: handle P/Invokes + abstract Initialize/Read."""

    @staticmethod
    def SetDevice(deviceID: int):
        lib().SetDevice(int(deviceID))
        check("SetDevice")

    @staticmethod
    def CreateBlas():
        h = lib().CreateBlas()
        check("CreateBlas")
        return h

    @staticmethod
    def CreateSparse():
        h = lib().CreateSparse()
        check("CreateSparse")
        return h

    @staticmethod
    def CreateMatDescr():
        return lib().CreateMatDescr()

    def Initialize(self):
        raise NotImplementedError

    def Read(self):
        raise NotImplementedError


class ConjugateGradientSingleGpu(ConjugateGradientGpu):
    """ConjugateGradientSingleGpu.cs:9-179: whole solve in ONE native call."""

    DEVICE_ID = 0

    def __init__(self, count, maxNonZeroCount, _minIteration, _maxIteration, allowableResidual, rule=None):
        super().__init__(count, maxNonZeroCount, _minIteration, _maxIteration, allowableResidual)
        _lib.require_gpu()
        self.rule = rule            # None -> the reference's native Solve export
        self.cublas = self.CreateBlas()
        self.cusparse = self.CreateSparse()
        self.matDescr = self.CreateMatDescr()
        self.vectorA = VectorDouble(count * maxNonZeroCount)
        self.vectorColumnIndeces = VectorInt(count * maxNonZeroCount)
        self.vectorRowOffsets = VectorInt(count + 1)
        self.vectorX = VectorDouble(count)
        self.vectorB = VectorDouble(count)
        self.vectorAp = VectorDouble(count)
        self.vectorP = VectorDouble(count)
        self.vectorR = VectorDouble(count)
        self.trace = None
        self.status = 0

    def Dispose(self):
        for v in (self.vectorA, self.vectorColumnIndeces, self.vectorRowOffsets, self.vectorX, self.vectorB,
                  self.vectorAp, self.vectorP, self.vectorR):
            v.Dispose()
        if self.cublas:
            lib().DestroyBlas(self.cublas)
            lib().DestroySparse(self.cusparse)
            lib().DestroyMatDescr(self.matDescr)
            self.cublas = None

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass

    def Initialize(self):
        nonzeroCount = int(self.A.RowOffsets[self.Count])
        self.vectorA.CopyFrom(self.A.Elements, nonzeroCount)
        self.vectorColumnIndeces.CopyFrom(self.A.ColumnIndeces, nonzeroCount)
        self.vectorRowOffsets.CopyFrom(self.A.RowOffsets, self.Count + 1)
        self.vectorB.CopyFrom(self.b, self.Count)
        self.vectorX.CopyFrom(self.x, self.Count)

    def Solve(self, trace: bool = False):
        nonzeroCount = int(self.A.RowOffsets[self.Count])
        iteration = C.c_int(0)
        residual = C.c_double(0.0)
        L = lib()
        if self.rule is None and not trace:
            L.Solve(self.cublas, self.cusparse, self.matDescr,
                    self.vectorA.Ptr, self.vectorRowOffsets.Ptr, self.vectorColumnIndeces.Ptr,
                    self.vectorX.Ptr, self.vectorB.Ptr,
                    self.vectorAp.Ptr, self.vectorP.Ptr, self.vectorR.Ptr,
                    nonzeroCount, self.Count,
                    self.AllowableResidual, self.MinIteration, self.MaxIteration,
                    C.byref(iteration), C.byref(residual))
            self.Iteration = iteration.value - 1      # ConjugateGradientSingleGpu.cs:168
            self.Residual = residual.value
            msg = _lib.last_error()
            if msg:
                L.MgcgClearLastError()
                if "did not converge" in msg:
                    raise ApplicationException(msg)
                raise MgcgError(msg)
            return
        rule = _lib.RULE_NATIVE if self.rule is None else self.rule
        cap = max(self.MaxIteration, self.MinIteration) + 8 if trace else 0
        tr = np.zeros(max(cap, 1)) if trace else None
        st = L.SolveEx(self.cublas, self.cusparse, self.matDescr,
                       self.vectorA.Ptr, self.vectorRowOffsets.Ptr, self.vectorColumnIndeces.Ptr,
                       self.vectorX.Ptr, self.vectorB.Ptr,
                       self.vectorAp.Ptr, self.vectorP.Ptr, self.vectorR.Ptr,
                       nonzeroCount, self.Count,
                       self.AllowableResidual, self.MinIteration, self.MaxIteration, rule,
                       C.byref(iteration), C.byref(residual),
                       _ptr(tr) if trace else None, cap)
        self.Iteration = iteration.value
        self.Residual = residual.value
        self.status = st
        if trace:
            self.trace = tr[: self.Iteration + 1].copy()
        if st == _lib.MAXIT_EXCEEDED:
            L.MgcgClearLastError()
            raise ApplicationException(f"CG did not converge within MaxIteration={self.MaxIteration}")
        if st != _lib.OK:
            check("SolveEx")
            raise MgcgError(f"SolveEx failed with status {st}")

    def Read(self):
        self.vectorX.CopyTo(self.x, self.Count)


class ConjugateGradientParallelGpu(ConjugateGradientGpu):
    """ConjugateGradientParallelGpu.cs:11-595 -- every device of this process, host-driven phases
    (Solve0..3) with the host-staged halo (SyncP = P2Host + P2Device) and host sums of the per-device
    dot products, exactly as the reference orchestrates them.  Kept as the compatibility path; the
    fast path is ``ConjugateGradientRankGpu`` (one process per GPU, RCCL inside the library).
    The reference runs the per-device lambdas under Parallel.For; the phases here are issued in
    device order, which is what makes ``resultsDot`` sums reproducible."""

    def __init__(self, count, maxNonZeroCount, _minIteration, _maxIteration, allowableResidual, deviceCount=None):
        super().__init__(count, maxNonZeroCount, _minIteration, _maxIteration, allowableResidual)
        L = lib()
        _lib.require_gpu()
        self.deviceCount = L.GetDeviceCount() if deviceCount is None else int(deviceCount)
        self.offsetsForDevice = partition_offsets(self.Count, self.deviceCount)     # :271-277
        self.resultsDot = np.zeros(self.deviceCount)
        self.bufferHost = np.zeros(self.Count)
        self.minJ = [0] * self.deviceCount
        self.maxJ = [0] * self.deviceCount
        n = self.deviceCount
        self.cublas, self.cusparse, self.matDescr = [None] * n, [None] * n, [None] * n
        self.vectorElements, self.vectorColumnIndeces, self.vectorRowOffsets = [None] * n, [None] * n, [None] * n
        self.vectorX, self.vectorB, self.vectorAp, self.vectorP, self.vectorR = [None] * n, [None] * n, [None] * n, [None] * n, [None] * n
        for d in range(n):                                                           # :301-323
            self.SetDevice(d)
            c = self.CountForDevice(d)
            self.cublas[d] = self.CreateBlas()
            self.cusparse[d] = self.CreateSparse()
            self.matDescr[d] = self.CreateMatDescr()
            self.vectorElements[d] = VectorDouble(c * maxNonZeroCount)
            self.vectorColumnIndeces[d] = VectorInt(c * maxNonZeroCount)
            self.vectorRowOffsets[d] = VectorInt(c + 1)
            self.vectorX[d] = VectorDouble(c)
            self.vectorB[d] = VectorDouble(c)
            self.vectorAp[d] = VectorDouble(c)
            self.vectorP[d] = VectorDouble(count)
            self.vectorR[d] = VectorDouble(c)

    def Dispose(self):
        for d in range(self.deviceCount):
            if self.cublas[d] is None:
                continue
            self.SetDevice(d)
            for vs in (self.vectorElements, self.vectorColumnIndeces, self.vectorRowOffsets, self.vectorX, self.vectorB,
                       self.vectorAp, self.vectorP, self.vectorR):
                vs[d].Dispose()
            lib().DestroyBlas(self.cublas[d])
            lib().DestroySparse(self.cusparse[d])
            lib().DestroyMatDescr(self.matDescr[d])
            self.cublas[d] = None

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass

    def CountForDevice(self, deviceID: int) -> int:                                  # :590-594
        if 0 <= deviceID < self.deviceCount:
            return self.offsetsForDevice[deviceID + 1] - self.offsetsForDevice[deviceID]
        return 0

    def _elementRange(self, d):
        ro = self.A.RowOffsets
        off = self.offsetsForDevice
        return int(ro[off[d + 1]] - ro[off[d]]), int(ro[off[d]])

    def Initialize(self):                                                            # :358-379
        L = lib()
        for d in range(self.deviceCount):
            self.SetDevice(d)
            elementCount, elementOffset = self._elementRange(d)
            mn, mx = C.c_int(0), C.c_int(0)
            L.Initialize(_ptr(self.A.Elements), _ptr(self.A.RowOffsets), _ptr(self.A.ColumnIndeces),
                         _ptr(self.x), _ptr(self.b),
                         self.vectorElements[d].Ptr, self.vectorRowOffsets[d].Ptr, self.vectorColumnIndeces[d].Ptr,
                         self.vectorX[d].Ptr, self.vectorB[d].Ptr, self.vectorP[d].Ptr,
                         C.byref(mn), C.byref(mx), self.Count,
                         self.CountForDevice(d), self.offsetsForDevice[d], elementCount, elementOffset)
            check("Initialize")
            self.minJ[d], self.maxJ[d] = mn.value, mx.value

    def _halo(self, d):                                                              # :397-398
        offset = self.offsetsForDevice[d]
        count = self.CountForDevice(d)
        lastCount = offset - self.minJ[d] if d > 0 else 0
        nextCount = self.maxJ[d] - count - offset + 1 if d < self.deviceCount - 1 else 0
        return count, offset, lastCount, nextCount

    def SyncP(self):                                                                 # :384-419
        L = lib()
        for d in range(self.deviceCount):
            self.SetDevice(d)
            count, offset, lastCount, nextCount = self._halo(d)
            L.P2Host(self.vectorP[d].Ptr, _ptr(self.bufferHost), count, offset, lastCount, nextCount)
            check("P2Host")
        for d in range(self.deviceCount):
            self.SetDevice(d)
            count, offset, lastCount, nextCount = self._halo(d)
            L.P2Device(self.vectorP[d].Ptr, _ptr(self.bufferHost), count, offset, lastCount, nextCount)
            check("P2Device")

    def Solve(self):                                                                 # :424-565
        L = lib()
        self.SyncP()
        for d in range(self.deviceCount):
            self.SetDevice(d)
            elementCount, _ = self._elementRange(d)
            self.resultsDot[d] = L.Solve0(self.cublas[d], self.cusparse[d], self.matDescr[d],
                                          self.vectorElements[d].Ptr, self.vectorRowOffsets[d].Ptr, self.vectorColumnIndeces[d].Ptr,
                                          self.vectorX[d].Ptr, self.vectorB[d].Ptr,
                                          self.vectorAp[d].Ptr, self.vectorP[d].Ptr, self.vectorR[d].Ptr,
                                          self.Count, self.CountForDevice(d), self.offsetsForDevice[d], elementCount)
        check("Solve0")
        rr = _sum_in_order(self.resultsDot)
        self.Iteration = 0
        while True:
            self.SyncP()
            for d in range(self.deviceCount):
                self.SetDevice(d)
                elementCount, _ = self._elementRange(d)
                self.resultsDot[d] = L.Solve1(self.cublas[d], self.cusparse[d], self.matDescr[d],
                                              self.vectorElements[d].Ptr, self.vectorRowOffsets[d].Ptr, self.vectorColumnIndeces[d].Ptr,
                                              self.vectorAp[d].Ptr, self.vectorP[d].Ptr,
                                              self.Count, self.CountForDevice(d), self.offsetsForDevice[d], elementCount)
            alpha = rr / _sum_in_order(self.resultsDot)
            for d in range(self.deviceCount):
                self.SetDevice(d)
                self.resultsDot[d] = L.Solve2(self.cublas[d], alpha, self.vectorX[d].Ptr,
                                              self.vectorAp[d].Ptr, self.vectorP[d].Ptr, self.vectorR[d].Ptr,
                                              self.CountForDevice(d), self.offsetsForDevice[d])
            rrNew = _sum_in_order(self.resultsDot)
            self.Residual = float(np.sqrt(rrNew))
            check("Solve1/Solve2")
            if self.IsConverged:
                break
            beta = rrNew / rr
            for d in range(self.deviceCount):
                self.SetDevice(d)
                L.Solve3(self.cublas[d], beta, self.vectorP[d].Ptr, self.vectorR[d].Ptr,
                         self.CountForDevice(d), self.offsetsForDevice[d])
            rr = rrNew
            self.Iteration += 1

    def Read(self):                                                                  # :570-583
        for d in range(self.deviceCount):
            self.SetDevice(d)
            self.vectorX[d].CopyTo(self.x, self.CountForDevice(d), self.offsetsForDevice[d])


def _sum_in_order(values) -> float:
    """resultsDot.Sum(): device-id order (ConjugateGradientParallelGpu.cs:463,499,525)."""
    s = 0.0
    for v in values:
        s += float(v)
    return s
