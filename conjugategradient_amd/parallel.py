"""One process per GPU: the row-range data-parallel CG of
Mgcg/cuBlas/Mgcg/ConjugateGradientParallelGpu.cs re-hosted on ``torch.distributed`` ranks.

Two drivers share the partition / halo arithmetic in this file:

* ``ConjugateGradientRankGpu`` -- the fast path.  ``Solve()`` is ONE native call (``SolveParallel``):
  the loop, the two scalar all-reduces (RCCL ``ncclAllReduce`` over xGMI, replacing
  ``resultsDot.Sum()`` :463,499,525) and the halo exchange of p (grouped ``ncclSend/ncclRecv``,
  replacing ``SyncP`` :384-419) all run inside libMgcgGpu.so on the rank's stream.
  ``torch.distributed`` only carries the 128-byte RCCL unique id to the ranks.
* ``PhasedRankSolver`` -- the reference's host-driven phase structure (Solve0..3 around host-side
  sums) with ``torch.distributed`` collectives (gloo or nccl) between the phases.  The phase
  arithmetic is a ``backend`` object: ``HipPhases`` (the C ABI exports) in production; the CPU-only
  gloo tests plug in a backend built on the oracle to check this file's partition, halo plan and
  collective plumbing without a GPU.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import MgcgError, check, lib
from .problems import partition_offsets
from .solver import ApplicationException, ConjugateGradientGpu, VectorDouble, VectorInt, _ptr


# --------------------------------------------------------------------------- partition / halo arithmetic
@dataclass
class RankPartition:
    """Rows [offset, offset+count) of rank ``rank`` (ConjugateGradientParallelGpu.cs:271-277,364-367)."""

    rank: int
    world: int
    Count: int
    offset: int
    count: int
    elementOffset: int = 0
    elementCount: int = 0
    minJ: int = 0
    maxJ: int = -1

    @classmethod
    def of(cls, Count: int, world: int, rank: int, RowOffsets=None, balance: str = "rows"):
        off = partition_offsets(Count, world, RowOffsets, balance)
        p = cls(rank, world, Count, off[rank], off[rank + 1] - off[rank])
        if RowOffsets is not None:
            p.elementOffset = int(RowOffsets[off[rank]])
            p.elementCount = int(RowOffsets[off[rank + 1]] - RowOffsets[off[rank]])
        return p

    @property
    def lastCount(self) -> int:      # :397  entries needed before the slice
        return self.offset - self.minJ if (self.rank > 0 and self.count > 0 and self.maxJ >= self.minJ) else 0

    @property
    def nextCount(self) -> int:      # :398  entries needed after the slice
        if self.rank < self.world - 1 and self.count > 0 and self.maxJ >= self.minJ:
            return max(0, self.maxJ - self.count - self.offset + 1)
        return 0


def halo_plan(meta: list[tuple[int, int, int, int]], rank: int):
    """meta[q] = (offset, count, minJ, maxJ) of every rank.  Returns (sends, recvs): lists of
    (peer, begin, length) -- the contiguous range of p this rank sends to / receives from each peer.
    A rank needs exactly the columns [minJ, maxJ] of its slice (Mgcg.cu:83-84); what lies outside its
    own rows is owned by other ranks.  For a banded matrix this is the reference's adjacent-neighbour
    exchange; for an unstructured one it degenerates to an all-gather.  Same logic as
    halo_plan_create in csrc/comm.hip."""
    off, cnt, mn, mx = meta[rank]
    sends, recvs = [], []
    for q, (qo, qc, qmn, qmx) in enumerate(meta):
        if q == rank:
            continue
        if cnt > 0 and mx >= mn:
            s, e = max(mn, qo), min(mx + 1, qo + qc)
            if e > s:
                recvs.append((q, s, e - s))
        if qc > 0 and qmx >= qmn:
            s, e = max(qmn, off), min(qmx + 1, off + cnt)
            if e > s:
                sends.append((q, s, e - s))
    return sends, recvs


# --------------------------------------------------------------------------- RCCL bootstrap
def create_comm(rank: int, world: int):
    """MgcgComm for this rank; the RCCL unique id travels over the default torch.distributed group."""
    L = lib()
    if world == 1:
        c = L.MgcgCommInitRank(None, 1, 0)
        check("MgcgCommInitRank")
        return c
    import torch
    import torch.distributed as dist

    buf = (C.c_char * 128)()
    failed = 0
    if rank == 0 and L.MgcgCommGetUniqueId(buf) != 0:
        failed = 1                      # still take part in the broadcast: the other ranks are waiting in it
    t = torch.tensor(list(bytes(buf)) + [failed], dtype=torch.uint8)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.broadcast(t, src=0)
    raw = bytes(t.cpu().tolist())
    if raw[128]:
        if rank == 0:
            check("MgcgCommGetUniqueId")
        raise MgcgError("MgcgCommGetUniqueId failed on rank 0")
    idbuf = (C.c_char * 128).from_buffer_copy(raw[:128])
    c = L.MgcgCommInitRank(idbuf, world, rank)
    check("MgcgCommInitRank")
    if not c:
        raise MgcgError("MgcgCommInitRank returned NULL")
    return c


_ALLGATHER_FN = C.CFUNCTYPE(None, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.c_void_p)
_ALLREDUCE_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int, C.c_void_p)
_EXCHANGE_FN = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_longlong),
                           C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_longlong), C.c_void_p)
_CALLBACK_KEEPALIVE: dict = {}


def create_callback_comm(rank: int, world: int):
    """MgcgComm whose collectives are carried by the default torch.distributed group on HOST memory (gloo): the
    host-staged fallback for machines where RCCL cannot form a communicator (``MgcgCommInitCallbacks``)."""
    import torch
    import torch.distributed as dist

    def all_gather(mine, out, _user):
        t = torch.tensor([mine[i] for i in range(4)], dtype=torch.int64)
        parts = [torch.zeros(4, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(parts, t)
        for q in range(world):
            for i in range(4):
                out[4 * q + i] = int(parts[q][i])

    def all_reduce(values, count, _user):
        t = torch.from_numpy(np.ctypeslib.as_array(values, shape=(count,)))     # shares the library's buffer
        dist.all_reduce(t)

    def exchange(n, send_bufs, send_counts, recv_bufs, recv_counts, _user):
        reqs = []
        for q in range(n):
            if recv_counts[q] > 0:
                reqs.append(dist.irecv(torch.from_numpy(np.ctypeslib.as_array(recv_bufs[q], shape=(recv_counts[q],))), src=q))
        for q in range(n):
            if send_counts[q] > 0:
                reqs.append(dist.isend(torch.from_numpy(np.ctypeslib.as_array(send_bufs[q], shape=(send_counts[q],))), dst=q))
        for r in reqs:
            r.wait()

    fns = (_ALLGATHER_FN(all_gather), _ALLREDUCE_FN(all_reduce), _EXCHANGE_FN(exchange))
    c = lib().MgcgCommInitCallbacks(world, rank, C.cast(fns[0], C.c_void_p), C.cast(fns[1], C.c_void_p), C.cast(fns[2], C.c_void_p), None)
    check("MgcgCommInitCallbacks")
    if not c:
        raise MgcgError("MgcgCommInitCallbacks returned NULL")
    _CALLBACK_KEEPALIVE[c] = fns           # the library calls these for as long as the communicator lives
    return c


class ConjugateGradientRankGpu(ConjugateGradientGpu):
    """This rank's share of ConjugateGradientParallelGpu: same constructor arguments and members, plus
    (rank, world).  ``Initialize()`` uploads the rank's row slice with the reference's ``Initialize``
    export (or generates a Poisson slab in HBM); ``Solve()`` runs the native RCCL loop; ``Read()`` fills
    this rank's slice of ``x``."""

    def __init__(self, count, maxNonZeroCount, _minIteration, _maxIteration, allowableResidual,
                 rank: int = 0, world: int = 1, comm=None, rule=_lib.RULE_CSHARP, device: int | None = None, balance: str = "rows"):
        super().__init__(count, maxNonZeroCount, _minIteration, _maxIteration, allowableResidual)
        _lib.require_gpu()
        self.rank, self.world, self.rule = rank, world, rule
        self.balance = balance                  # "rows": the reference's partition; "nnz": equal nonzero counts (problems.partition_offsets)
        self.SetDevice(rank % _lib.device_count() if device is None else device)
        self.comm = comm
        self._own_comm = False
        self.part = RankPartition.of(count, world, rank)
        self.maxNonZeroCount = maxNonZeroCount
        self.cublas = self.CreateBlas()
        self.cusparse = self.CreateSparse()
        self.matDescr = self.CreateMatDescr()
        c = self.part.count
        self.vectorElements = None
        self.vectorColumnIndeces = None
        self.vectorRowOffsets = VectorInt(c + 1)
        self.vectorX = VectorDouble(c)
        self.vectorB = VectorDouble(c)
        self.vectorAp = VectorDouble(c)
        self.vectorP = VectorDouble(count)      # full length, as in the reference (:317)
        self.vectorR = VectorDouble(c)
        self.trace = None
        self.status = 0

    def Dispose(self):
        if getattr(self, "cublas", None):
            for v in (self.vectorElements, self.vectorColumnIndeces, self.vectorRowOffsets, self.vectorX, self.vectorB,
                      self.vectorAp, self.vectorP, self.vectorR):
                if v is not None:
                    v.Dispose()
            if self._own_comm and self.comm:
                lib().MgcgCommDestroy(self.comm)
            lib().DestroyBlas(self.cublas)
            lib().DestroySparse(self.cusparse)
            lib().DestroyMatDescr(self.matDescr)
            self.cublas = None

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass

    def _ensure_comm(self):
        if self.comm is None:
            self.comm = create_comm(self.rank, self.world)
            self._own_comm = True

    def Initialize(self):
        """Upload this rank's partition from the host arrays (A, x, b), as :358-379 does per device."""
        p = RankPartition.of(self.Count, self.world, self.rank, self.A.RowOffsets, self.balance)
        if p.count != self.part.count:          # (a partition that follows the matrix is only known now)
            for name in ("vectorX", "vectorB", "vectorAp", "vectorR"):
                getattr(self, name).Dispose()
                setattr(self, name, VectorDouble(p.count))
            self.vectorRowOffsets.Dispose()
            self.vectorRowOffsets = VectorInt(p.count + 1)
        self.vectorElements = VectorDouble(max(p.elementCount, 1))
        self.vectorColumnIndeces = VectorInt(max(p.elementCount, 1))
        mn, mx = C.c_int(0), C.c_int(0)
        lib().Initialize(_ptr(self.A.Elements), _ptr(self.A.RowOffsets), _ptr(self.A.ColumnIndeces),
                         _ptr(self.x), _ptr(self.b),
                         self.vectorElements.Ptr, self.vectorRowOffsets.Ptr, self.vectorColumnIndeces.Ptr,
                         self.vectorX.Ptr, self.vectorB.Ptr, self.vectorP.Ptr,
                         C.byref(mn), C.byref(mx), self.Count,
                         p.count, p.offset, p.elementCount, p.elementOffset)
        check("Initialize")
        p.minJ, p.maxJ = mn.value, mx.value
        self.part = p

    def InitializePoisson(self, nx: int, ny: int, nz: int, b_value: float = 1.0, x_value: float = 0.0):
        """Generate this rank's z-slab of the 5/7-point Poisson matrix directly in HBM (rows must split on
        plane boundaries, which floor(N/world) does whenever world divides nz)."""
        L = lib()
        p = self.part
        sxy = nx * ny
        if p.offset % sxy or p.count % sxy:
            raise MgcgError("the row partition does not fall on z-plane boundaries")
        z0, z1 = p.offset // sxy, (p.offset + p.count) // sxy
        nnz = L.MgcgPoissonNnz(nx, ny, nz, z0, z1)
        self.vectorElements = VectorDouble(nnz)
        self.vectorColumnIndeces = VectorInt(nnz)
        if L.MgcgGeneratePoisson(self.vectorElements.Ptr, self.vectorRowOffsets.Ptr, self.vectorColumnIndeces.Ptr, nx, ny, nz, z0, z1) != 0:
            check("MgcgGeneratePoisson")
        L.MgcgFill(self.vectorB.Ptr, b_value)
        L.MgcgFill(self.vectorX.Ptr, x_value)
        mn, mx = C.c_int(0), C.c_int(0)
        if L.MgcgMinMaxColumn(self.vectorColumnIndeces.Ptr, nnz, C.byref(mn), C.byref(mx)) != 0:
            check("MgcgMinMaxColumn")
        p.elementCount, p.elementOffset, p.minJ, p.maxJ = int(nnz), 0, mn.value, mx.value

    def Solve(self, trace: bool = False):
        self._ensure_comm()
        p = self.part
        iteration, residual = C.c_int(0), C.c_double(0.0)
        cap = max(self.MaxIteration, self.MinIteration) + 8 if trace else 0
        tr = np.zeros(max(cap, 1)) if trace else None
        st = lib().SolveParallel(self.comm, self.cublas, self.cusparse, self.matDescr,
                                 self.vectorElements.Ptr, self.vectorRowOffsets.Ptr, self.vectorColumnIndeces.Ptr,
                                 self.vectorX.Ptr, self.vectorB.Ptr, self.vectorAp.Ptr, self.vectorP.Ptr, self.vectorR.Ptr,
                                 self.Count, p.count, p.offset, p.elementCount, p.minJ, p.maxJ,
                                 self.AllowableResidual, self.MinIteration, self.MaxIteration, self.rule,
                                 C.byref(iteration), C.byref(residual), _ptr(tr) if trace else None, cap)
        self.Iteration, self.Residual, self.status = iteration.value, residual.value, st
        if trace:
            self.trace = tr[: self.Iteration + 1].copy()
        if st == _lib.MAXIT_EXCEEDED:
            lib().MgcgClearLastError()
            raise ApplicationException(f"CG did not converge within MaxIteration={self.MaxIteration}")
        if st != _lib.OK:
            check("SolveParallel")
            raise MgcgError(f"SolveParallel failed with status {st}")

    @staticmethod
    def LastOverlap():
        """(active, first interior row, end of interior rows) of the calling thread's last multi-rank solve."""
        rng = (C.c_longlong * 2)(0, 0)
        active = lib().MgcgLastOverlap(rng)
        return bool(active), int(rng[0]), int(rng[1])

    def Steps(self, steps: int, restart: bool = True) -> float:
        """``steps`` CG iterations with no stop test and no host sync inside (bench.py)."""
        self._ensure_comm()
        p = self.part
        res = lib().CgSteps(self.comm, self.cublas, self.cusparse,
                            self.vectorElements.Ptr, self.vectorRowOffsets.Ptr, self.vectorColumnIndeces.Ptr,
                            self.vectorX.Ptr, self.vectorB.Ptr, self.vectorAp.Ptr, self.vectorP.Ptr, self.vectorR.Ptr,
                            self.Count, p.count, p.offset, p.elementCount, p.minJ, p.maxJ, int(steps), 1 if restart else 0)
        check("CgSteps")
        return res

    def Read(self):
        self.vectorX.CopyTo(self.x, self.part.count, self.part.offset)


class ConjugateGradientMgRankGpu(ConjugateGradientRankGpu):
    """One rank of the row-partitioned MGCG (BASELINE.json config 4): the rank's z-slab of the grid, per-level halo
    planes over the communicator, V-cycle + PCG inside the library (MgSetupParallel / SolveMgParallel)."""

    def __init__(self, count, maxNonZeroCount, _minIteration, _maxIteration, allowableResidual, grid,
                 rank: int = 0, world: int = 1, comm=None, rule=_lib.RULE_CSHARP, device: int | None = None,
                 levels: int = 3, omega: float | None = None, nu: int = 1, nuCoarse: int = 4, sigma: float = 0.5,
                 interpolation: int = 0):
        super().__init__(count, maxNonZeroCount, _minIteration, _maxIteration, allowableResidual, rank=rank, world=world,
                         comm=comm, rule=rule, device=device)
        self.interpolation = int(interpolation)          # 0: piecewise constant, 1: cell-centred linear (MgSetInterpolation)
        self.grid = tuple(int(g) for g in grid)
        nx, ny, nz = self.grid
        if nx * ny * nz != count:
            raise MgcgError("grid does not match count")
        self.levels_requested = levels
        self.omega = (6.0 / 7.0 if nz > 1 else 4.0 / 5.0) if omega is None else float(omega)
        self.nu, self.nuCoarse, self.sigma = int(nu), int(nuCoarse), float(sigma)
        self.vectorZ = VectorDouble(self.part.count)
        self.mg = None

    def Dispose(self):
        if getattr(self, "mg", None):
            lib().MgDestroy(self.mg)
            self.mg = None
        if getattr(self, "vectorZ", None) is not None:
            self.vectorZ.Dispose()
        super().Dispose()

    def Setup(self):
        """Build the hierarchy on this rank's slab (call after Initialize / InitializePoisson; collective)."""
        self._ensure_comm()
        nx, ny, nz = self.grid
        p = self.part
        sxy = nx * ny
        if p.offset % sxy or p.count % sxy:
            raise MgcgError("the row partition does not fall on z-plane boundaries")
        self.mg = lib().MgSetupParallel(self.comm, self.cublas, self.cusparse,
                                        self.vectorElements.Ptr, self.vectorRowOffsets.Ptr, self.vectorColumnIndeces.Ptr,
                                        p.elementCount, nx, ny, nz, p.offset // sxy, (p.offset + p.count) // sxy,
                                        self.levels_requested, self.omega, self.nu, self.nuCoarse, self.sigma)
        check("MgSetupParallel")
        if not self.mg:
            raise MgcgError("MgSetupParallel returned NULL")
        if self.interpolation and lib().MgSetInterpolation(self.mg, self.interpolation) != 0:
            check("MgSetInterpolation")
        self.levels = lib().MgLevels(self.mg)

    def Apply(self, r_local: np.ndarray) -> np.ndarray:
        """z = M^-1 r on the local rows (test helper; collective)."""
        n = self.part.count
        vr, vz = VectorDouble(n), VectorDouble(n)
        vr.CopyFrom(np.ascontiguousarray(r_local, dtype=np.float64), n)
        lib().MgApply(self.mg, vr.ToRawPtr(), vz.ToRawPtr())
        check("MgApply")
        z = vz.to_numpy()
        vr.Dispose()
        vz.Dispose()
        return z

    def Solve(self, trace: bool = False):
        self._ensure_comm()
        p = self.part
        iteration, residual = C.c_int(0), C.c_double(0.0)
        cap = max(self.MaxIteration, self.MinIteration) + 8 if trace else 0
        tr = np.zeros(max(cap, 1)) if trace else None
        st = lib().SolveMgParallel(self.comm, self.cublas, self.cusparse, self.matDescr, self.mg,
                                   self.vectorElements.Ptr, self.vectorRowOffsets.Ptr, self.vectorColumnIndeces.Ptr,
                                   self.vectorX.Ptr, self.vectorB.Ptr, self.vectorAp.Ptr, self.vectorP.Ptr, self.vectorR.Ptr, self.vectorZ.Ptr,
                                   self.Count, p.count, p.offset, p.elementCount, p.minJ, p.maxJ,
                                   self.AllowableResidual, self.MinIteration, self.MaxIteration, self.rule,
                                   C.byref(iteration), C.byref(residual), _ptr(tr) if trace else None, cap)
        self.Iteration, self.Residual, self.status = iteration.value, residual.value, st
        if trace:
            self.trace = tr[: self.Iteration + 1].copy()
        if st == _lib.MAXIT_EXCEEDED:
            lib().MgcgClearLastError()
            raise ApplicationException(f"MGCG did not converge within MaxIteration={self.MaxIteration}")
        if st != _lib.OK:
            check("SolveMgParallel")
            raise MgcgError(f"SolveMgParallel failed with status {st}")


# --------------------------------------------------------------------------- host-driven phases over torch.distributed
class HipPhases:
    """The reference's per-device phase calls (Mgcg.cu:57-198) for ONE rank, through the C ABI."""

    def __init__(self, system_A, x, b, part: RankPartition, maxNonZeroCount: int):
        _lib.require_gpu()
        self.part = part
        L = lib()
        L.SetDevice(part.rank % _lib.device_count())
        self.cublas, self.cusparse, self.matDescr = L.CreateBlas(), L.CreateSparse(), L.CreateMatDescr()
        check("Create handles")
        c = part.count
        self.vE, self.vC, self.vRO = VectorDouble(max(part.elementCount, 1)), VectorInt(max(part.elementCount, 1)), VectorInt(c + 1)
        self.vX, self.vB, self.vAp, self.vP, self.vR = VectorDouble(c), VectorDouble(c), VectorDouble(c), VectorDouble(part.Count), VectorDouble(c)
        mn, mx = C.c_int(0), C.c_int(0)
        L.Initialize(_ptr(system_A.Elements), _ptr(system_A.RowOffsets), _ptr(system_A.ColumnIndeces), _ptr(x), _ptr(b),
                     self.vE.Ptr, self.vRO.Ptr, self.vC.Ptr, self.vX.Ptr, self.vB.Ptr, self.vP.Ptr,
                     C.byref(mn), C.byref(mx), part.Count, part.count, part.offset, part.elementCount, part.elementOffset)
        check("Initialize")
        part.minJ, part.maxJ = mn.value, mx.value

    def get_p(self, begin: int, length: int) -> np.ndarray:
        out = np.empty(length)
        self.vP.CopyTo(out, length, 0, begin)
        return out

    def set_p(self, begin: int, values: np.ndarray):
        self.vP.CopyFrom(np.ascontiguousarray(values, dtype=np.float64), values.shape[0], 0, begin)

    def solve0(self) -> float:
        p = self.part
        v = lib().Solve0(self.cublas, self.cusparse, self.matDescr, self.vE.Ptr, self.vRO.Ptr, self.vC.Ptr, self.vX.Ptr, self.vB.Ptr,
                         self.vAp.Ptr, self.vP.Ptr, self.vR.Ptr, p.Count, p.count, p.offset, p.elementCount)
        check("Solve0")
        return v

    def solve1(self) -> float:
        p = self.part
        v = lib().Solve1(self.cublas, self.cusparse, self.matDescr, self.vE.Ptr, self.vRO.Ptr, self.vC.Ptr,
                         self.vAp.Ptr, self.vP.Ptr, p.Count, p.count, p.offset, p.elementCount)
        check("Solve1")
        return v

    def solve2(self, alpha: float) -> float:
        p = self.part
        v = lib().Solve2(self.cublas, alpha, self.vX.Ptr, self.vAp.Ptr, self.vP.Ptr, self.vR.Ptr, p.count, p.offset)
        check("Solve2")
        return v

    def solve3(self, beta: float):
        p = self.part
        lib().Solve3(self.cublas, beta, self.vP.Ptr, self.vR.Ptr, p.count, p.offset)
        check("Solve3")

    def read_x(self) -> np.ndarray:
        return self.vX.to_numpy(self.part.count)


class PhasedRankSolver:
    """ConjugateGradientParallelGpu.Solve (:424-565) with one rank per process: the phases run on the
    ``backend``; SyncP and resultsDot.Sum() become torch.distributed point-to-point and all-reduce."""

    def __init__(self, backend, part: RankPartition, minIteration: int, maxIteration: int, allowableResidual: float, dist=None):
        self.backend, self.part = backend, part
        self.MinIteration, self.MaxIteration, self.AllowableResidual = minIteration, maxIteration, allowableResidual
        self.Iteration, self.Residual = 0, 0.0
        self.dist = dist
        self._plan = None

    # -- collectives -----------------------------------------------------------------
    def _allreduce(self, value: float) -> float:
        if self.part.world == 1:
            return value
        import torch

        t = torch.tensor([value], dtype=torch.float64)
        self.dist.all_reduce(t)       # SUM; every rank gets the identical bits
        return float(t[0])

    def _build_plan(self):
        import torch

        p = self.part
        mine = torch.tensor([p.offset, p.count, p.minJ, p.maxJ], dtype=torch.int64)
        if p.world == 1:
            allm = [mine]
        else:
            allm = [torch.zeros(4, dtype=torch.int64) for _ in range(p.world)]
            self.dist.all_gather(allm, mine)
        meta = [tuple(int(v) for v in m) for m in allm]
        self._plan = halo_plan(meta, p.rank)

    def SyncP(self):
        if self.part.world == 1:
            return
        import torch

        if self._plan is None:
            self._build_plan()
        sends, recvs = self._plan
        reqs, bufs = [], []
        for peer, begin, length in recvs:
            t = torch.empty(length, dtype=torch.float64)
            bufs.append((begin, t))
            reqs.append(self.dist.irecv(t, src=peer))
        for peer, begin, length in sends:
            t = torch.from_numpy(self.backend.get_p(begin, length))
            reqs.append(self.dist.isend(t, dst=peer))
        for r in reqs:
            r.wait()
        for begin, t in bufs:
            self.backend.set_p(begin, t.numpy())

    # -- the loop --------------------------------------------------------------------
    @property
    def IsConverged(self) -> bool:
        if self.Iteration < self.MinIteration:
            return False
        elif self.Iteration > self.MaxIteration:
            raise ApplicationException("the pressure equation did not converge")
        return self.Residual < self.AllowableResidual

    def Solve(self):
        self.SyncP()
        rr = self._allreduce(self.backend.solve0())
        self.Iteration = 0
        while True:
            self.SyncP()
            alpha = rr / self._allreduce(self.backend.solve1())
            rrNew = self._allreduce(self.backend.solve2(alpha))
            self.Residual = math.sqrt(rrNew)
            if self.IsConverged:
                break
            beta = rrNew / rr
            self.backend.solve3(beta)
            rr = rrNew
            self.Iteration += 1
