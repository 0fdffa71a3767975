"""ctypes binding of libMgcgGpu.so -- the same binding surface the reference's C#
classes declare with ``[DllImport(MgcgGpu.DLL_NAME, EntryPoint = ...)]``
(Mgcg/cuBlas/Mgcg/ConjugateGradientGpu.cs:17-66, ConjugateGradientSingleGpu.cs:70-98,
ConjugateGradientParallelGpu.cs:117-262, VectorDouble.cs:20-66, VectorInt.cs:20-60).

There is no CPU fallback: if the shared object is missing this module raises, and
on a host without a GPU every compute entry point fails with the library's own
"no HIP device" error.
"""
from __future__ import annotations

import ctypes as C
import os

DLL_NAME = "libMgcgGpu.so"  # MgcgGpu.cs:11 has "MgcgGpu.dll"
_PKG = os.path.dirname(os.path.abspath(__file__))
# MGCG_LIB_PATH: the lab build of the same sources (make -C csrc lab) for tools/spmv_sweep.py --ablate; never set by tests or bench.py
LIB_PATH = os.environ.get("MGCG_LIB_PATH") or os.path.join(_PKG, DLL_NAME)

_vp = C.c_void_p
_i = C.c_int
_d = C.c_double
_ll = C.c_longlong
_pi = C.POINTER(C.c_int)
_pd = C.POINTER(C.c_double)

# name -> (restype, argtypes); mirrors include/MgcgGpu.h one to one
SIGNATURES = {
    # Runtime.cu
    "GetDeviceCount": (_i, []),
    "SetDevice": (None, [_i]),
    "CreateBlas": (_vp, []),
    "DestroyBlas": (None, [_vp]),
    "CreateSparse": (_vp, []),
    "DestroySparse": (None, [_vp]),
    "CreateMatDescr": (_vp, []),
    "DestroyMatDescr": (None, [_vp]),
    # Vector_Double.cu / Vector_Int.cu
    "Create_Double": (_vp, [_i]),
    "CopyToArray_Double": (None, [_vp, _vp, _i, _i, _i]),
    "CopyFromArray_Double": (None, [_vp, _vp, _i, _i, _i]),
    "Delete_Double": (None, [_vp]),
    "ToRawPtr_Double": (_vp, [_vp]),
    "CopyFromDevice_Double": (None, [_vp, _vp, _i, _i, _i]),
    "Create_Int": (_vp, [_i]),
    "CopyToArray_Int": (None, [_vp, _vp, _i, _i, _i]),
    "CopyFromArray_Int": (None, [_vp, _vp, _i, _i, _i]),
    "Delete_Int": (None, [_vp]),
    "ToRawPtr_Int": (_vp, [_vp]),
    # Mgcg.cu ops
    "CsrMV": (None, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _d, _d]),
    "Axpy": (None, [_vp, _vp, _vp, _i, _d]),
    "Dot": (_d, [_vp, _vp, _vp, _i]),
    "Scal": (None, [_vp, _vp, _d, _i]),
    "Copy": (None, [_vp, _vp, _vp, _i, _i, _i]),
    "Initialize": (None, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _pi, _pi, _i, _i, _i, _i, _i]),
    "P2Host": (None, [_vp, _vp, _i, _i, _i, _i]),
    "P2Device": (None, [_vp, _vp, _i, _i, _i, _i]),
    "Solve0": (_d, [_vp] * 11 + [_i, _i, _i, _i]),
    "Solve1": (_d, [_vp] * 8 + [_i, _i, _i, _i]),
    "Solve2": (_d, [_vp, _d, _vp, _vp, _vp, _vp, _i, _i]),
    "Solve3": (None, [_vp, _d, _vp, _vp, _i, _i]),
    "Solve": (None, [_vp] * 11 + [_i, _i, _d, _i, _i, _pi, _pd]),
    # additive
    "MgcgGetLastError": (C.c_char_p, []),
    "MgcgClearLastError": (None, []),
    "MgcgAbiVersion": (_i, []),
    "MgcgCommInitAll": (_i, [_vp, _i]),
    "MgcgCommTransport": (C.c_char_p, [_vp]),
    "MgcgCommProbe": (_d, [_vp, _i, _i, _i]),
    "MgcgSetTuning": (_i, [C.c_char_p, _i]),
    "MgcgGetTuning": (_i, [C.c_char_p, _pi]),
    "MgcgReloadEnvironment": (None, []),
    "MgcgDeviceSynchronize": (_i, []),
    "MgcgEventCreate": (_vp, []),
    "MgcgEventRecord": (None, [_vp]),
    "MgcgEventElapsedMs": (C.c_float, [_vp, _vp]),
    "MgcgEventDestroy": (None, [_vp]),
    "MgcgMemGetInfo": (_i, [C.POINTER(_ll), C.POINTER(_ll)]),
    "MgcgCreateDouble64": (_vp, [_ll]),
    "MgcgCreateInt64": (_vp, [_ll]),
    "MgcgVectorSize": (_ll, [_vp]),
    "Xpay": (None, [_vp, _vp, _vp, _i, _d]),
    "NrmInf": (_d, [_vp, _vp, _i]),
    "CsrMVDot": (_d, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i]),
    "MgcgSetSpmvKernel": (None, [_vp, _i]),
    "MgcgSetSpmvTuning": (None, [_vp, _i, _i, _i]),
    "MgcgSetSpmvPeriod": (None, [_vp, _i]),
    "MgcgSetSpmvTile": (None, [_vp, _i, _i]),
    "MgcgSetMatrixCompression": (None, [_vp, _i]),
    "MgcgAnalysisClear": (None, [_vp]),
    "MgcgAnalysisInfo": (_i, [_vp, _i, _pi, _pi, C.POINTER(_ll), C.POINTER(_ll)]),
    "MgcgProfileSpmv": (None, [_vp, _i]),
    "MgcgProfileSpmvMs": (_d, [_vp, _pi]),
    "MgcgPoissonNnz": (_ll, [_i, _i, _i, _i, _i]),
    "MgcgGeneratePoisson": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "MgcgMinMaxColumn": (_i, [_vp, _i, _pi, _pi]),
    "MgcgFill": (None, [_vp, _d]),
    "SolveEx": (_i, [_vp] * 11 + [_i, _i, _d, _i, _i, _i, _pi, _pd, _vp, _i]),
    "MgSetup": (_vp, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _i, _i, _d]),
    "MgSetupParallel": (_vp, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _d, _i, _i, _d]),
    "MgDestroy": (None, [_vp]),
    "MgSetInterpolation": (_i, [_vp, _i]),
    "MgLevels": (_i, [_vp]),
    "MgLevelRows": (_ll, [_vp, _i]),
    "MgLevelNnz": (_ll, [_vp, _i]),
    "MgLevelCopyCsr": (None, [_vp, _i, _vp, _vp, _vp]),
    "MgLevelCopyDinv": (None, [_vp, _i, _vp]),
    "MgApply": (None, [_vp, _vp, _vp]),
    "SolveMg": (_i, [_vp] * 13 + [_i, _i, _d, _i, _i, _i, _pi, _pd, _vp, _i]),
    "SolveMgParallel": (_i, [_vp] * 14 + [_i, _i, _i, _i, _i, _i, _d, _i, _i, _i, _pi, _pd, _vp, _i]),
    "MgcgCommGetUniqueId": (_i, [_vp]),
    "MgcgRcclAvailable": (_i, []),
    "MgcgCommInitRank": (_vp, [_vp, _i, _i]),
    "MgcgLoopbackCreate": (_vp, [_i]),
    "MgcgLoopbackDestroy": (None, [_vp]),
    "MgcgCommInitLoopback": (_vp, [_vp, _i]),
    "MgcgCommDestroy": (None, [_vp]),
    "MgcgCommRank": (_i, [_vp]),
    "MgcgCommSize": (_i, [_vp]),
    "MgcgCommAllReduceSum": (_d, [_vp, _d]),
    "SolveParallel": (_i, [_vp] * 12 + [_i, _i, _i, _i, _i, _i, _d, _i, _i, _i, _pi, _pd, _vp, _i]),
    "CgSteps": (_d, [_vp] * 11 + [_i, _i, _i, _i, _i, _i, _i, _i]),
    "MgcgLastOverlap": (_i, [_vp]),
    "MgcgLastOverlapTimes": (_i, [_vp]),
    "MgcgLastPlacement": (_i, [_i, _vp, _i, _pi]),
    "MgcgLastVcycleFolds": (_i, []),
    "MgcgLastHalo": (_i, [_vp]),
    "MgcgDebugTileOrder": (_i, [_ll, _i, _i, _i, _vp, _i]),
    "MgcgCommInitCallbacks": (_vp, [_i, _i, _vp, _vp, _vp, _vp]),
    "MgcgEstimateSpectrum": (_i, [_vp] * 5 + [_i, _i, _i, _i, C.c_uint, _vp, _vp, _vp, _vp]),
}

RULE_NATIVE, RULE_CSHARP, RULE_SIMPLE, RULE_HANDMADECL, RULE_VIENNACL = range(5)
OK, MAXIT_EXCEEDED, NONFINITE, ERROR = 0, 1, 3, -1

_LIB = None


class MgcgError(RuntimeError):
    pass


def lib():
    """Load libMgcgGpu.so (built in-tree by ``__graft_entry__.build()`` / ``make -C csrc``)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise MgcgError(
                f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_PKG, 'csrc')}` "
                "(there is no CPU fallback for the HIP path)")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def last_error() -> str:
    return lib().MgcgGetLastError().decode("utf-8", "replace")


def check(where: str = ""):
    """Raise if the library recorded an error on this thread."""
    msg = last_error()
    if msg:
        lib().MgcgClearLastError()
        raise MgcgError(f"{where}: {msg}" if where else msg)


def device_count() -> int:
    return lib().GetDeviceCount()


def require_gpu():
    n = device_count()
    if n <= 0:
        raise MgcgError("no HIP device visible: the MI355X path cannot run (no CPU fallback exists)")
    return n
