"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

See oracle/cg_oracle.c and oracle/mg_oracle.c for what each entry restates
(reference file:line) and for the parity status ("parity unpinned": the
reference ships no recorded outputs).  Only tests/, ``__graft_entry__.smoke()``
and bench.py's ``cpu_baseline`` leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RULE_CSHARP, RULE_NATIVE, RULE_SIMPLE, RULE_HANDMADECL, RULE_VIENNACL = range(5)
OK, MAXIT_EXCEEDED, HARDCAP, NONFINITE = range(4)

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_lp = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build(asan: bool = False) -> str:
    target = "liboracle_asan.so" if asan else "liboracle.so"
    subprocess.check_call(["make", "-s", "-C", _HERE, target])
    return os.path.join(_HERE, target)


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("ORACLE_LIB_PATH") or os.path.join(_HERE, "liboracle.so")     # (the variable: the sanitizer build, make liboracle_asan.so)
        srcs = [os.path.join(_HERE, f) for f in ("cg_oracle.c", "mg_oracle.c")]
        if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs if os.path.exists(s)):
            build()
        L = C.CDLL(path)
        L.oracle_spmv.argtypes = [_dp, _ip, _ip, C.c_int64, _dp, _dp]
        L.oracle_dot.argtypes = [_dp, _dp, C.c_int64]
        L.oracle_dot.restype = C.c_double
        L.oracle_set_added.argtypes = [_dp, _dp, _dp, C.c_double, C.c_int64]
        L.oracle_max_absolute.argtypes = [_dp, C.c_int64]
        L.oracle_max_absolute.restype = C.c_double
        L.oracle_scal.argtypes = [_dp, C.c_double, C.c_int64]
        L.oracle_cg.argtypes = [_dp, _ip, _ip, C.c_int64, _dp, _dp, C.c_int, C.c_double, C.c_int, C.c_int,
                                C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p, C.c_int64, C.c_void_p]
        L.oracle_cg.restype = C.c_int
        L.oracle_cg_steps.argtypes = [_dp, _ip, _ip, C.c_int64, _dp, _dp, C.c_int, C.POINTER(C.c_double), _dp]
        L.oracle_partition.argtypes = [C.c_int64, C.c_int, _lp]
        L.oracle_minmax_column.argtypes = [_ip, _ip, C.c_int64, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_cg_parallel.argtypes = [_dp, _ip, _ip, C.c_int64, C.c_int, _dp, _dp, C.c_double, C.c_int, C.c_int,
                                         C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p, C.c_int64]
        L.oracle_cg_parallel.restype = C.c_int
        L.oracle_cg_parallel_offsets.argtypes = [_dp, _ip, _ip, C.c_int64, C.c_int, _lp, _dp, _dp, C.c_double, C.c_int, C.c_int,
                                                 C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p, C.c_int64]
        L.oracle_cg_parallel_offsets.restype = C.c_int
        L.oracle_poisson_nnz.argtypes = [C.c_int, C.c_int, C.c_int]
        L.oracle_poisson_nnz.restype = C.c_int64
        L.oracle_poisson_fill.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _ip, _ip]
        L.oracle_mgcgmain_fill.argtypes = [C.c_int, C.c_int, _dp, _ip, _ip]
        L.oracle_mgcgmain_fill.restype = C.c_int64
        # multigrid
        L.oracle_mg_galerkin.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _ip, _ip, C.c_double, C.c_void_p, C.c_void_p, _ip]
        L.oracle_mg_galerkin.restype = C.c_int
        L.oracle_mg_restrict.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp]
        L.oracle_mg_prolong_add.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp]
        L.oracle_mg_restrict_linear.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp]
        L.oracle_mg_prolong_add_linear.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp]
        L.oracle_mg_set_interpolation.argtypes = [C.c_void_p, C.c_int]
        L.oracle_mg_jacobi_first.argtypes = [C.c_int64, C.c_double, _dp, _dp, _dp]
        L.oracle_mg_jacobi.argtypes = [_dp, _ip, _ip, C.c_int64, C.c_double, _dp, _dp, _dp, _dp]
        L.oracle_mg_residual.argtypes = [_dp, _ip, _ip, C.c_int64, _dp, _dp, _dp]
        L.oracle_mg_setup.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _ip, _ip, C.c_double, C.c_int, C.c_int, C.c_double]
        L.oracle_mg_setup.restype = C.c_void_p
        L.oracle_mg_free.argtypes = [C.c_void_p]
        L.oracle_mg_levels.argtypes = [C.c_void_p]
        L.oracle_mg_levels.restype = C.c_int
        L.oracle_mg_level_rows.argtypes = [C.c_void_p, C.c_int]
        L.oracle_mg_level_rows.restype = C.c_int64
        L.oracle_mg_level_nnz.argtypes = [C.c_void_p, C.c_int]
        L.oracle_mg_level_nnz.restype = C.c_int64
        L.oracle_mg_level_dims.argtypes = [C.c_void_p, C.c_int, _ip]
        L.oracle_mg_level_csr.argtypes = [C.c_void_p, C.c_int, _dp, _ip, _ip]
        L.oracle_mg_level_dinv.argtypes = [C.c_void_p, C.c_int, _dp]
        L.oracle_mg_apply.argtypes = [C.c_void_p, _dp, _dp]
        L.oracle_pcg.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.c_double, C.c_int, C.c_int,
                                 C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p, C.c_int64]
        L.oracle_pcg.restype = C.c_int
        L.oracle_pcg_parts.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, _lp,
                                       C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p, C.c_int64]
        L.oracle_pcg_parts.restype = C.c_int
        _LIB = L
    return _LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class compensated_dots:
    """``with oracle.compensated_dots():`` -- every dot product of the oracle loops sums the same rounded products with Neumaier's
    compensation instead of the reference's serial left-to-right order (cg_oracle.c: oracle_set_dot_mode).  NOT the reference's
    arithmetic: a yardstick for the reference order's own rounding error at large sizes."""

    def __enter__(self):
        self._saved = lib().oracle_get_dot_mode()
        lib().oracle_set_dot_mode(1)
        return self

    def __exit__(self, *exc):
        lib().oracle_set_dot_mode(self._saved)
        return False


def spmv(elements, column_indeces, row_offsets, vector):
    ro = _i(row_offsets)
    n = ro.shape[0] - 1
    out = np.empty(n)
    lib().oracle_spmv(_f(elements), _i(column_indeces), ro, n, out, _f(vector))
    return out


def dot(left, right):
    left = _f(left)
    return lib().oracle_dot(left, _f(right), left.shape[0])


def set_added(left, right, a):
    left = _f(left)
    out = np.empty_like(left)
    lib().oracle_set_added(out, left, _f(right), a, left.shape[0])
    return out


def max_absolute(v):
    v = _f(v)
    return lib().oracle_max_absolute(v, v.shape[0])


def cg(system, rule=RULE_CSHARP, allowable_residual=1e-8, min_iteration=0, max_iteration=None,
       hard_cap=None, trace=False):
    """Solve with the reference CG.  Returns dict(x, iteration, residual, status[, trace]).
    ``iteration`` is the zero-based index of the last executed loop body (C# ``Iteration``)."""
    n = system.Count
    x = _f(system.x).copy()
    max_iteration = n if max_iteration is None else max_iteration
    hard_cap = (max_iteration + 2) if hard_cap is None else hard_cap
    it = C.c_int(0)
    res = C.c_double(0)
    tr = np.zeros(hard_cap + 1) if trace else None
    st = lib().oracle_cg(_f(system.Elements), _i(system.ColumnIndeces), _i(system.RowOffsets), n, x, _f(system.b),
                         rule, allowable_residual, min_iteration, max_iteration, hard_cap,
                         C.byref(it), C.byref(res),
                         tr.ctypes.data if trace else None, (hard_cap + 1) if trace else 0, None)
    out = dict(x=x, iteration=it.value, residual=res.value, status=st)
    if trace:
        out["trace"] = tr[: it.value + 1].copy()
    return out


def cg_parallel(system, device_count, allowable_residual=1e-8, min_iteration=0, max_iteration=None, trace=False, offsets=None):
    """offsets: device_count + 1 row offsets of another row-range partition than the reference's floor(count / devices)."""
    n = system.Count
    x = _f(system.x).copy()
    max_iteration = n if max_iteration is None else max_iteration
    it = C.c_int(0)
    res = C.c_double(0)
    cap = max_iteration + 3
    tr = np.zeros(cap) if trace else None
    if offsets is None:
        st = lib().oracle_cg_parallel(_f(system.Elements), _i(system.ColumnIndeces), _i(system.RowOffsets), n, device_count,
                                      x, _f(system.b), allowable_residual, min_iteration, max_iteration,
                                      C.byref(it), C.byref(res), tr.ctypes.data if trace else None, cap if trace else 0)
    else:
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        assert off.shape == (device_count + 1,) and off[0] == 0 and off[-1] == n and np.all(np.diff(off) >= 0)
        st = lib().oracle_cg_parallel_offsets(_f(system.Elements), _i(system.ColumnIndeces), _i(system.RowOffsets), n, device_count, off,
                                              x, _f(system.b), allowable_residual, min_iteration, max_iteration,
                                              C.byref(it), C.byref(res), tr.ctypes.data if trace else None, cap if trace else 0)
    out = dict(x=x, iteration=it.value, residual=res.value, status=st)
    if trace:
        out["trace"] = tr[: it.value + 1].copy()
    return out


def partition(count, device_count):
    off = np.zeros(device_count + 1, dtype=np.int64)
    lib().oracle_partition(count, device_count, off)
    return off


def minmax_column(system, row_begin, row_end):
    lo, hi = C.c_int(0), C.c_int(0)
    lib().oracle_minmax_column(_i(system.ColumnIndeces), _i(system.RowOffsets), row_begin, row_end, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def poisson_csr(nx, ny, nz):
    L = lib()
    n = nx * ny * nz
    nnz = L.oracle_poisson_nnz(nx, ny, nz)
    e = np.empty(nnz)
    c = np.empty(nnz, dtype=np.int32)
    r = np.empty(n + 1, dtype=np.int32)
    L.oracle_poisson_fill(nx, ny, nz, e, c, r)
    return e, c, r


def mgcgmain_csr(count, max_nonzero=160):
    L = lib()
    cap = count * max_nonzero
    e = np.zeros(cap)
    c = np.full(cap, -1, dtype=np.int32)
    r = np.zeros(count + 1, dtype=np.int32)
    nnz = L.oracle_mgcgmain_fill(count, max_nonzero, e, c, r)
    return e[:nnz].copy(), c[:nnz].copy(), r


class Multigrid:
    """V(nu,nu) weighted-Jacobi geometric multigrid preconditioner (mg_oracle.c)."""

    def __init__(self, system, levels=3, omega=None, nu=1, nu_coarse=4, sigma=0.5, interpolation=0):
        nx, ny, nz = system.grid
        if omega is None:
            omega = 6.0 / 7.0 if nz > 1 else 4.0 / 5.0
        self.system = system
        self._e, self._c, self._r = _f(system.Elements), _i(system.ColumnIndeces), _i(system.RowOffsets)
        self.h = lib().oracle_mg_setup(nx, ny, nz, levels, self._e, self._c, self._r, omega, nu, nu_coarse, sigma)
        self.levels = lib().oracle_mg_levels(self.h)
        self.omega, self.nu, self.nu_coarse, self.sigma = omega, nu, nu_coarse, sigma
        self.interpolation = int(interpolation)          # 0: piecewise constant, 1: cell-centred linear
        lib().oracle_mg_set_interpolation(self.h, self.interpolation)

    def __del__(self):
        try:
            lib().oracle_mg_free(self.h)
        except Exception:
            pass

    def level_dims(self, l):
        d = np.zeros(3, dtype=np.int32)
        lib().oracle_mg_level_dims(self.h, l, d)
        return tuple(int(v) for v in d)

    def level_csr(self, l):
        n = lib().oracle_mg_level_rows(self.h, l)
        nnz = lib().oracle_mg_level_nnz(self.h, l)
        e = np.empty(nnz)
        c = np.empty(nnz, dtype=np.int32)
        r = np.empty(n + 1, dtype=np.int32)
        lib().oracle_mg_level_csr(self.h, l, e, c, r)
        return e, c, r

    def level_dinv(self, l):
        d = np.empty(lib().oracle_mg_level_rows(self.h, l))
        lib().oracle_mg_level_dinv(self.h, l, d)
        return d

    def apply(self, r):
        z = np.empty(self.system.Count)
        lib().oracle_mg_apply(self.h, _f(r), z)
        return z

    def pcg(self, rule=RULE_CSHARP, allowable_residual=1e-8, min_iteration=0, max_iteration=None, trace=False, offsets=None):
        """offsets: row offsets of a row partition (len = devices + 1) -- the dot products are then per-device sums added in device
        order, as the reference's host does for the unpreconditioned loop (ConjugateGradientParallelGpu.cs:463,499,525)."""
        n = self.system.Count
        x = _f(self.system.x).copy()
        max_iteration = n if max_iteration is None else max_iteration
        it = C.c_int(0)
        res = C.c_double(0)
        cap = max_iteration + 3
        tr = np.zeros(cap) if trace else None
        if offsets is None:
            st = lib().oracle_pcg(self.h, x, _f(self.system.b), rule, allowable_residual, min_iteration, max_iteration,
                                  C.byref(it), C.byref(res), tr.ctypes.data if trace else None, cap if trace else 0)
        else:
            off = np.ascontiguousarray(offsets, dtype=np.int64)
            assert off[0] == 0 and off[-1] == n and np.all(np.diff(off) >= 0)
            st = lib().oracle_pcg_parts(self.h, x, _f(self.system.b), rule, allowable_residual, min_iteration, max_iteration,
                                        off.shape[0] - 1, off, C.byref(it), C.byref(res), tr.ctypes.data if trace else None, cap if trace else 0)
        out = dict(x=x, iteration=it.value, residual=res.value, status=st)
        if trace:
            out["trace"] = tr[: it.value + 1].copy()
        return out
