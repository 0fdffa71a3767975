"""The C-ABI library: loads on a host without a GPU, exports every symbol include/MgcgGpu.h declares,
and fails LOUDLY (no CPU fallback) when asked to compute without a device.  CPU only."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "MgcgGpu.h")

REFERENCE_EXPORTS = [  # the 32 exports of MgcgGpu.dll
    "GetDeviceCount", "SetDevice", "CreateBlas", "DestroyBlas", "CreateSparse", "DestroySparse", "CreateMatDescr", "DestroyMatDescr",
    "Create_Double", "CopyToArray_Double", "CopyFromArray_Double", "Delete_Double", "ToRawPtr_Double", "CopyFromDevice_Double",
    "Create_Int", "CopyToArray_Int", "CopyFromArray_Int", "Delete_Int", "ToRawPtr_Int",
    "CsrMV", "Axpy", "Dot", "Scal", "Copy", "Initialize", "P2Host", "P2Device", "Solve0", "Solve1", "Solve2", "Solve3", "Solve",
]


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#.*", "", text)
    text = re.sub(r"typedef[^;]*\(\s*\*[^;]*;", "", text)        # function-pointer typedefs (callback types) declare no export
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text)
    skip = {"defined", "sizeof"}
    return sorted({n for n in names if n not in skip})


def test_header_declares_the_reference_exports():
    names = declared_functions()
    assert len(REFERENCE_EXPORTS) == 32
    for n in REFERENCE_EXPORTS:
        assert n in names, n


def test_library_exports_every_declared_symbol(hiplib):
    from conjugategradient_amd import _lib

    names = declared_functions()
    assert len(names) >= 60
    for n in names:
        assert hasattr(hiplib, n), f"{n} declared in include/MgcgGpu.h but not exported by libMgcgGpu.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    # nothing but the C ABI leaks out of the shared object
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert exported == set(names), exported ^ set(names)


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "MgcgGpu.h"\nint main(void) { return MGCG_RULE_VIENNACL == 4 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "t.o")])


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="this check is for hosts without a GPU")
def test_no_device_fails_loudly(hiplib):
    """No HIP device: handles are NULL, value-returning ops return NaN, and the reason is reported --
    the product never computes on the CPU."""
    from conjugategradient_amd import _lib
    from conjugategradient_amd.solver import ConjugateGradientSingleGpu

    L = hiplib
    assert L.GetDeviceCount() == 0
    assert not L.CreateBlas()
    assert "no HIP device" in _lib.last_error()
    L.MgcgClearLastError()
    assert not L.Create_Double(8)
    L.MgcgClearLastError()
    x = np.ones(4)
    assert np.isnan(L.Dot(None, x.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), 4))
    L.MgcgClearLastError()
    with pytest.raises(_lib.MgcgError):
        ConjugateGradientSingleGpu(10, 3, 0, 10, 1e-8)
    with pytest.raises(_lib.MgcgError):
        _lib.require_gpu()
    assert L.MgcgPoissonNnz(512, 512, 512, 0, 512) == 937951232      # host-side arithmetic still answers
    assert L.MgcgPoissonNnz(256, 256, 1, 0, 1) == 326656


def test_missing_library_is_an_error(tmp_path, monkeypatch):
    code = ("import sys; sys.path.insert(0, %r); from conjugategradient_amd import _lib; "
            "_lib.LIB_PATH = %r; \ntry:\n _lib.lib()\nexcept _lib.MgcgError as e:\n print('raised', 'no CPU fallback' in str(e))") % (ROOT, str(tmp_path / "nope.so"))
    out = subprocess.check_output([sys.executable, "-c", code]).decode()
    assert "raised True" in out


def test_tuning_knobs_are_read_once_and_settable(hiplib, monkeypatch):
    """The library reads its MGCG_* variables once; knobs change through MgcgSetTuning / MgcgReloadEnvironment, never by a
    launch looking at the environment.  No device needed."""
    from conjugategradient_amd import _lib

    L = hiplib
    v = C.c_int(-99)
    L.MgcgReloadEnvironment()
    assert L.MgcgGetTuning(b"overlap", C.byref(v)) == 0 and v.value == 1          # default
    assert L.MgcgGetTuning(b"check_every", C.byref(v)) == 0 and v.value == 4
    monkeypatch.setenv("MGCG_OVERLAP", "2")
    assert L.MgcgGetTuning(b"overlap", C.byref(v)) == 0 and v.value == 1          # the environment is not re-read ...
    L.MgcgReloadEnvironment()
    assert L.MgcgGetTuning(b"MGCG_OVERLAP", C.byref(v)) == 0 and v.value == 2     # ... until asked; the variable's name works too
    assert L.MgcgSetTuning(b"overlap", 0) == 0
    assert L.MgcgGetTuning(b"overlap", C.byref(v)) == 0 and v.value == 0
    assert L.MgcgSetTuning(b"no_such_knob", 1) == -1 and "unknown knob" in _lib.last_error()
    L.MgcgClearLastError()
    monkeypatch.undo()
    L.MgcgReloadEnvironment()
    assert L.MgcgGetTuning(b"overlap", C.byref(v)) == 0 and v.value == 1
    # the timing ablations that produce wrong results are not in the product library
    assert b"MGCG_SPMV_ABLATE" not in open(_lib.LIB_PATH, "rb").read()


def test_comm_init_all_writes_exactly_the_entries_it_was_given(hiplib):
    """Regression test for the host SIGSEGVs of round 3 (gpurun_out/r3/pytest_gpu_2.log, pytest_gdb.log: a Python-internal crash inside,
    or some fifty tests after, test_comm_init_all_single_process -- heap corruption, not one bad call).  The call
    that was new in that working tree and writes into caller memory is MgcgCommInitAll(comms, ndev): it clears comms[0 .. ndev) before
    anything else, so an ndev larger than the ctypes array handed in overruns a Python-owned buffer.  The contract since: ndev outside
    [1, 64] is refused before any write, otherwise exactly ndev entries are written -- checked here with sentinels either side of the
    array (works with or without a GPU: every call below fails, or succeeds, after the same clearing step)."""
    L = hiplib
    sentinel = 0x5A5A5A5A5A5A5A5A
    guard = 16
    for ndev in (-1, 0, 1, 3, 64, 65, 1 << 20):
        room = max(0, min(ndev, 64))
        buf = (C.c_uint64 * (guard + room + guard))(*([sentinel] * (guard + room + guard)))
        base = C.addressof(buf) + 8 * guard
        rc = L.MgcgCommInitAll(C.c_void_p(base), ndev)
        made = [buf[guard + d] for d in range(room)]
        if rc == 0:                                          # (a GPU box with enough devices: real communicators -- give them back)
            for ptr in made:
                assert ptr not in (0, sentinel)
                L.MgcgCommDestroy(C.c_void_p(ptr))
        else:
            assert rc == -1 and hiplib.MgcgGetLastError()
            if 1 <= ndev <= 64:
                assert made == [0] * room                    # every entry the caller was promised is NULL after a failure
        L.MgcgClearLastError()
        assert list(buf[:guard]) == [sentinel] * guard and list(buf[guard + room:]) == [sentinel] * guard, ndev
    assert L.MgcgCommInitAll(None, 2) == -1
    L.MgcgClearLastError()


def test_the_knob_list_is_the_header_s(hiplib, monkeypatch):
    """Round 5 retired the knobs whose A/B was settled: the library answers to exactly the 15 names of include/MgcgGpu.h's tuning list, with
    the defaults written there, and no longer to the retired ones.  No device needed."""
    from conjugategradient_amd import _lib

    for name in list(os.environ):
        if name.startswith("MGCG_"):
            monkeypatch.delenv(name)
    L = hiplib
    L.MgcgReloadEnvironment()
    kept = {"overlap": 1, "halo_stream": 0, "deep_halo": 1, "no_fold": 0, "fold_up": -1, "check_every": 4, "auto_tiles": 1, "tile_shift": 0, "tile_pack": 1,
            "placement": 3, "dot_order": 0, "verbose": 0, "virtual_devices": 0, "force_multirank": 0, "fail_comm_init": 0}
    v = C.c_int(-99)
    for name, default in kept.items():
        assert L.MgcgGetTuning(name.encode(), C.byref(v)) == 0 and v.value == default, (name, v.value)
        assert L.MgcgGetTuning(("MGCG_" + name.upper()).encode(), C.byref(v)) == 0 and v.value == default
    header = open(os.path.join(ROOT, "include", "MgcgGpu.h")).read()
    assert "Tuning knobs (%d)" % len(kept) in header
    for name in kept:
        assert name in header, name
    for gone in ("no_folded_finalize", "no_uniform_diagonal", "no_zsweep", "rowtile_nt", "vec_nt", "vec_grid", "r_grid", "xp_grid", "pattern_group",
                 "pattern_waves", "no_indexed_halo", "tile_nt", "vector_vals_nt", "lazy_code_objects"):
        assert L.MgcgSetTuning(gone.encode(), 1) == -1, gone
    L.MgcgClearLastError()
    assert L.MgcgAbiVersion() == 3
