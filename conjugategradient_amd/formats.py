"""Caller-side matrix builders of the reference's other two solver families and their flattening to the CSR arrays
the library takes.  Host side only, numpy.

* ``EllSparseMatrix`` -- Mgcg/HandmadeCL/MgcgCL/SparseMatrix.cs:8-230: fixed ``MaxNonzeroCountPerRow`` slots per row,
  slot 0 of every row is the diagonal (always present, column id = row), further entries are appended in the order of
  their first assignment; ``A[i, j]`` reads 0 for an entry that was never set and assigning beyond the capacity of a row
  raises ``IndexError`` (the reference throws ``IndexOutOfRangeException``, ``:121-124``).
* ``CompressedMatrix`` -- Mgcg/ViennaCL/MgcgCL/CompressedMatrix.cs:8-70: a list of per-row dictionaries keyed by the
  unsigned column id; rows appear when first assigned; flattened by the driver in dictionary (= insertion) order with
  ``uint`` offsets and column ids (MgcgCL.cs:85-97).

``to_csr()`` keeps the stored order of every row, so a CPU ``Multiply`` over the builder and the device SpMV over the
flattened arrays add the products of a row in the same order.
"""
from __future__ import annotations

import numpy as np

from .problems import LinearSystem


class EllSparseMatrix:
    """HandmadeCL's SparseMatrix (slot-0 diagonal ELL)."""

    def __init__(self, rowCount: int, maxNonzeroCountPerRow: int):
        if maxNonzeroCountPerRow < 1:
            raise ValueError("maxNonzeroCountPerRow must be >= 1 (slot 0 holds the diagonal)")
        self.MaxNonzeroCountPerRow = int(maxNonzeroCountPerRow)
        self.Elements = np.zeros(rowCount * maxNonzeroCountPerRow, dtype=np.float64)
        self.ColumnIndeces = np.zeros(rowCount * maxNonzeroCountPerRow, dtype=np.int32)
        self.NonzeroCounts = np.zeros(rowCount, dtype=np.int32)
        self.Clear()

    def Clear(self):
        """SparseMatrix.cs:52-63: one entry per row, the diagonal, set to zero (other slots keep stale data)."""
        K = self.MaxNonzeroCountPerRow
        self.NonzeroCounts[:] = 1
        self.Elements[::K] = 0.0
        self.ColumnIndeces[::K] = np.arange(self.NonzeroCounts.shape[0], dtype=np.int32)

    @property
    def RowCount(self) -> int:
        return int(self.NonzeroCounts.shape[0])

    def _local(self, i: int, j: int) -> int:
        """GetLocalIndex (:163-181): slots 1 .. count-1 are searched; the diagonal slot is never matched."""
        first = i * self.MaxNonzeroCountPerRow
        cols = self.ColumnIndeces[first + 1: first + int(self.NonzeroCounts[i])]
        hit = np.nonzero(cols == j)[0]
        return int(hit[0]) + 1 if hit.size else -1

    def __getitem__(self, key):
        K = self.MaxNonzeroCountPerRow
        if not isinstance(key, tuple):
            return float(self.Elements[int(key) * K])
        i, j = int(key[0]), int(key[1])
        if i == j:
            return float(self.Elements[i * K])
        k = self._local(i, j)
        return float(self.Elements[i * K + k]) if k >= 0 else 0.0

    def __setitem__(self, key, value):
        K = self.MaxNonzeroCountPerRow
        if not isinstance(key, tuple):
            key = (key, key)
        i, j = int(key[0]), int(key[1])
        if i == j:
            self.Elements[i * K] = value
            self.ColumnIndeces[i * K] = i
            return
        k = self._local(i, j)
        if k < 0:
            if self.NonzeroCounts[i] == K:
                raise IndexError(f"row {i} already holds MaxNonzeroCountPerRow = {K} entries")
            k = int(self.NonzeroCounts[i])
            self.ColumnIndeces[i * K + k] = j
            self.NonzeroCounts[i] += 1
        self.Elements[i * K + k] = value

    def Multiply(self, answer: np.ndarray, vector: np.ndarray):
        """SparseMatrix.cs:200-223, products added in slot order (vectorised over rows: slot k of every row at once,
        which adds in exactly that order)."""
        K = self.MaxNonzeroCountPerRow
        n = self.RowCount
        e = self.Elements.reshape(n, K)
        c = self.ColumnIndeces.reshape(n, K)
        answer[:] = 0.0
        for k in range(int(self.NonzeroCounts.max()) if n else 0):
            live = self.NonzeroCounts > k
            answer[live] += e[live, k] * vector[c[live, k]]

    def to_csr(self):
        """(Elements, ColumnIndeces, RowOffsets) with the slots of every row packed in stored order (diagonal first)."""
        K = self.MaxNonzeroCountPerRow
        n = self.RowCount
        live = np.arange(K, dtype=np.int32)[None, :] < self.NonzeroCounts[:, None]
        ro = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(self.NonzeroCounts, out=ro[1:])
        if ro[-1] > np.iinfo(np.int32).max:
            raise OverflowError("more than 2^31-1 nonzeros: int32 row offsets cannot address them")
        return (np.ascontiguousarray(self.Elements.reshape(n, K)[live]),
                np.ascontiguousarray(self.ColumnIndeces.reshape(n, K)[live]),
                ro.astype(np.int32))

    def to_system(self, x, b, name: str = "ell") -> LinearSystem:
        e, c, ro = self.to_csr()
        return LinearSystem(e, c, ro, np.array(x, dtype=np.float64), np.array(b, dtype=np.float64), name=name)

    @classmethod
    def from_csr(cls, Elements, ColumnIndeces, RowOffsets, maxNonzeroCountPerRow: int | None = None) -> "EllSparseMatrix":
        """CSR -> slot-0-diagonal ELL (the diagonal moves to the front of its row, a missing one becomes an explicit 0;
        the other entries keep their order)."""
        ro = np.asarray(RowOffsets, dtype=np.int64)
        n = ro.shape[0] - 1
        cnt = np.diff(ro)
        rows = np.repeat(np.arange(n, dtype=np.int64), cnt)
        cols = np.asarray(ColumnIndeces[: ro[-1]], dtype=np.int64)
        vals = np.asarray(Elements[: ro[-1]], dtype=np.float64)
        isdiag = cols == rows
        if np.any(np.bincount(rows[isdiag], minlength=n) > 1):
            raise ValueError("a row stores its diagonal more than once")
        has = np.zeros(n, dtype=bool)
        has[rows[isdiag]] = True
        need = cnt + (~has).astype(np.int64)
        K = int(maxNonzeroCountPerRow if maxNonzeroCountPerRow is not None else (need.max() if n else 1))
        if n and need.max() > K:
            raise IndexError(f"a row needs {int(need.max())} slots, capacity is {K}")
        m = cls(n, K)
        m.Elements.reshape(n, K)[rows[isdiag], 0] = vals[isdiag]
        # slot of every off-diagonal entry = 1 + its rank among the off-diagonals of its row
        off = ~isdiag
        before = np.cumsum(off) - off                       # off-diagonals before entry k, globally
        first = np.zeros(n, dtype=np.int64)
        rs = ro[:-1][cnt > 0]
        first[cnt > 0] = before[rs]                         # ... before the row starts
        slot = 1 + (before - first[rows])
        m.Elements.reshape(n, K)[rows[off], slot[off]] = vals[off]
        m.ColumnIndeces.reshape(n, K)[rows[off], slot[off]] = cols[off].astype(np.int32)
        m.NonzeroCounts[:] = (1 + np.bincount(rows[off], minlength=n)).astype(np.int32)
        return m


class CompressedMatrix:
    """The ViennaCL driver's dictionary-of-rows matrix (CompressedMatrix.cs:8-70)."""

    def __init__(self):
        self.Elements: list[dict[int, float]] = []

    def __getitem__(self, key):
        i, j = int(key[0]), int(key[1])
        if i >= len(self.Elements):                          # (the reference tests `i > Count` and would throw at i == Count)
            return 0.0
        return float(self.Elements[i].get(j & 0xFFFFFFFF, 0.0))

    def __setitem__(self, key, value):
        i, j = int(key[0]), int(key[1])
        while i >= len(self.Elements):
            self.Elements.append({})
        self.Elements[i][j & 0xFFFFFFFF] = float(value)      # (uint)j

    def to_csr(self, n: int | None = None):
        """MgcgCL.cs:85-97: keys and values of every row in dictionary order; uint32 offsets and column ids.
        ``n``: number of rows to emit (rows never assigned are empty)."""
        rows = len(self.Elements) if n is None else int(n)
        ro = np.zeros(rows + 1, dtype=np.uint32)
        cols: list[int] = []
        vals: list[float] = []
        for i in range(rows):
            d = self.Elements[i] if i < len(self.Elements) else {}
            cols.extend(d.keys())
            vals.extend(d.values())
            ro[i + 1] = ro[i] + len(d)
        return np.array(vals, dtype=np.float64), np.array(cols, dtype=np.uint32), ro

    def to_system(self, x, b, name: str = "dictionary") -> LinearSystem:
        e, c, ro = self.to_csr(len(x))
        if c.size and c.max() > np.iinfo(np.int32).max:
            raise OverflowError("column id does not fit the library's int32 indices")
        return LinearSystem(e, c.astype(np.int32), ro.astype(np.int32), np.array(x, dtype=np.float64), np.array(b, dtype=np.float64), name=name)
