"""The C++ host twin of the reference classes (host/Mgcg.hpp) and its MgcgMain driver, run as a program.

The programs write their solutions as raw doubles and the tests compare them with the oracle ELEMENT BY ELEMENT, as the reference's driver
does with its CPU leg (Mgcg/cuBlas/Mgcg/MgcgMain.cs:129-162) -- at the north star's 1e-10 * max|x| in the default mode (or within the
oracle's own spread over device counts where forced iterations run into CG's round-off tail: computed and asserted), and bit for bit
with MGCG_DOT_ORDER=1 (every sum in the reference's order, tests/test_gpu_dot_order.py)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conjugategradient_amd import problems
from tests.gpu_util import assert_iterate_close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "conjugategradient_amd", "host", "MgcgMain")


def _equal_bits(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(np.all(a.view(np.uint64) == b.view(np.uint64)))


def _check_x(x, ref_x, dot_order, spread_refs=()):
    """element by element: equal under the reference's summation order, the north star's 1e-10 otherwise"""
    assert x.shape == ref_x.shape
    if dot_order == "1":
        assert _equal_bits(x, ref_x), (int((x != ref_x).sum()), float(np.abs(x - ref_x).max()))
    else:
        assert_iterate_close(x, ref_x, spread_refs=spread_refs)


@pytest.mark.parametrize("dot_order", ["0", "1"])
@pytest.mark.parametrize("devices,balance", [(1, False), (3, False), (3, True)])
def test_mgcg_main_driver(oracle, tmp_path, devices, balance, dot_order):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(EXE)])
    count, min_it = 20003, 40
    env = dict(os.environ, MGCG_VIRTUAL_DEVICES=str(devices), MGCG_DOT_ORDER=dot_order)
    prefix = str(tmp_path / "x")
    out = subprocess.run([EXE, str(count), str(min_it), "write=" + prefix] + (["balance"] if balance else []), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    s = problems.mgcg_main(count, 160)
    # BalanceNonzeros (not in the reference): the C++ class cuts where problems.partition_offsets(..., "nnz") does -- the short rows at both
    # ends of the band move the cuts off floor(count / devices)
    assert rec["offsets"] == problems.partition_offsets(count, devices, s.RowOffsets, "nnz" if balance else "rows")
    if balance:
        assert rec["offsets"] != problems.partition_offsets(count, devices)
    ref = oracle.cg(s, rule=oracle.RULE_NATIVE, min_iteration=min_it, max_iteration=count, hard_cap=count + 5)
    refp = oracle.cg_parallel(s, devices, min_iteration=min_it, max_iteration=count, offsets=rec["offsets"])
    assert rec["devices"] == devices
    assert rec["iteration_single"] == ref["iteration"] == refp["iteration"] == rec["iteration_parallel"] == rec["iteration_phases"] == min_it
    assert rec["mismatches"] == 0 and rec["max_rel_single_vs_parallel"] < 1e-8 and rec["max_rel_phases_vs_parallel"] < 1e-8
    # ConjugateGradientParallelGpu.Solve() took the native loop (SolveParallel per device thread), not the host-driven phases
    assert rec["parallel_path"] == "native loop (SolveParallel over %s)" % ("single" if devices == 1 else "loopback")
    # the three solutions, element by element (40 forced iterations end in the round-off tail: the default mode may use the oracle's spread)
    spread = [oracle.cg_parallel(s, w, min_iteration=min_it, max_iteration=count)["x"] for w in (1, 2, 3, 4)]
    _check_x(np.fromfile(prefix + ".single.f64"), ref["x"], dot_order, spread)
    _check_x(np.fromfile(prefix + ".phases.f64"), refp["x"], dot_order, spread)
    _check_x(np.fromfile(prefix + ".parallel.f64"), refp["x"], dot_order, spread)
    if dot_order == "1":
        assert rec["residual_single"] == ref["residual"] and rec["residual_parallel"] == refp["residual"]


def test_class_keeps_working_on_the_phases_when_no_communicator_forms(oracle, tmp_path):
    """A host whose RCCL cannot form a communicator (MGCG_FAIL_COMM_INIT: the test hook that makes MgcgCommInitAll report failure):
    ConjugateGradientParallelGpu must still construct and solve -- on the reference's own host-driven phases
    (ConjugateGradientParallelGpu.cs:424-565), which need no communicator -- and say so."""
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(EXE)])
    count, min_it = 6007, 30
    env = dict(os.environ, MGCG_VIRTUAL_DEVICES="3", MGCG_FAIL_COMM_INIT="1", MGCG_DOT_ORDER="1")
    prefix = str(tmp_path / "x")
    out = subprocess.run([EXE, str(count), str(min_it), "write=" + prefix], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["parallel_path"].startswith("host-driven phases (Solve0..3) -- no communicators:") and "MGCG_FAIL_COMM_INIT" in rec["parallel_path"]
    s = problems.mgcg_main(count, 160)
    ref = oracle.cg(s, rule=oracle.RULE_NATIVE, min_iteration=min_it, max_iteration=count, hard_cap=count + 5)
    assert rec["iteration_single"] == rec["iteration_parallel"] == ref["iteration"] == min_it and rec["mismatches"] == 0
    refp = oracle.cg_parallel(s, 3, min_iteration=min_it, max_iteration=count)
    _check_x(np.fromfile(prefix + ".single.f64"), ref["x"], "1")
    _check_x(np.fromfile(prefix + ".parallel.f64"), refp["x"], "1")          # (the kept answer: the phases, in the reference's summation order)


@pytest.mark.parametrize("dot_order", ["0", "1"])
def test_other_two_driver_families_in_cpp(oracle, tmp_path, dot_order):
    """host/MgcgFrontends.hpp + MgcgCLMain: the HandmadeCL ELL builder with the max-norm rule and the ViennaCL dictionary
    builder with the relative rule, filled by the reference drivers' loops in C++, against the oracle's same rules."""
    exe = os.path.join(ROOT, "conjugategradient_amd", "host", "MgcgCLMain")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(exe)])
    count = 2345
    prefix = str(tmp_path / "x")
    out = subprocess.run([exe, str(count), prefix], env=dict(os.environ, MGCG_DOT_ORDER=dot_order), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rec = {l.split()[0]: l.split()[1:] for l in out.stdout.splitlines() if l.split() and l.split()[0] in ("handmadecl", "viennacl")}
    s = problems.mgcg_main(count, 160)
    ref = oracle.cg(s, rule=oracle.RULE_HANDMADECL, allowable_residual=1e-4, min_iteration=50, max_iteration=count)
    assert int(rec["handmadecl"][0]) == ref["iteration"]
    # (50 forced iterations at a tolerance of 1e-4: the default mode may use the oracle's spread over device counts, computed inside)
    spread = [oracle.cg_parallel(s, w, allowable_residual=1e-4, min_iteration=50, max_iteration=count)["x"] for w in (2, 3)]
    _check_x(np.fromfile(prefix + ".handmadecl.f64"), ref["x"], dot_order, spread)
    v = problems.viennacl_main(count, 160)
    ref = oracle.cg(v, rule=oracle.RULE_VIENNACL, allowable_residual=1e-4, min_iteration=0, max_iteration=count, hard_cap=count + 10)
    assert int(rec["viennacl"][0]) == ref["iteration"] + 1
    _check_x(np.fromfile(prefix + ".viennacl.f64"), ref["x"], dot_order)


@pytest.mark.parametrize("dot_order", ["0", "1"])
def test_command_line_driver(oracle, tmp_path, dot_order):
    """host/mgcg_solve.cpp: flags instead of the reference's compile-time constants; CG and MGCG on a Poisson grid, on one device and on
    R devices of one process, each against the oracle's iteration count and its solution element by element (--write-x)."""
    exe = os.path.join(ROOT, "conjugategradient_amd", "host", "mgcg_solve")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(exe)])
    xfile = str(tmp_path / "x.f64")

    def run(*flags, ranks=1):
        env = dict(os.environ, MGCG_DOT_ORDER=dot_order)
        if ranks > 1:
            env["MGCG_VIRTUAL_DEVICES"] = str(ranks)
        out = subprocess.run([exe, *(("--ranks", str(ranks)) if ranks > 1 else ()), "--write-x", xfile, *flags], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        return json.loads(out.stdout.splitlines()[-1]), np.fromfile(xfile)

    s = problems.poisson(16, 16, 16)
    off = {r: oracle.partition(s.Count, r) for r in (2, 4)}
    ref = oracle.cg(s, rule=oracle.RULE_CSHARP, max_iteration=s.Count)
    rec, x = run("--n", "16", "--rule", "csharp", "--compression", "0")
    assert rec["iteration"] == ref["iteration"] == 43                       # SURVEY.md section 8c scratch count
    _check_x(x, ref["x"], dot_order)
    M3 = oracle.Multigrid(s, levels=3)
    mref = M3.pcg(rule=oracle.RULE_CSHARP, max_iteration=400)
    rec, x = run("--n", "16", "--mgcg", "--levels", "3", "--compression", "1")
    assert rec["levels"] == 3 and rec["iteration"] == mref["iteration"]
    _check_x(x, mref["x"], dot_order)
    ML = oracle.Multigrid(s, levels=3, interpolation=1)
    lref = ML.pcg(rule=oracle.RULE_CSHARP, max_iteration=400)
    rec, x = run("--n", "16", "--mgcg", "--levels", "3", "--linear-transfer", "--compression", "0")
    assert rec["iteration"] == lref["iteration"] < mref["iteration"]
    _check_x(x, lref["x"], dot_order)
    # --ranks R: R devices of ONE process, one host thread each, MgcgCommInitAll + SolveParallel / MgSetupParallel + SolveMgParallel
    # (the single-process shape of ConjugateGradientParallelGpu.cs:264-324), against the oracle with its sums cut at the ranks' rows
    refp = oracle.cg_parallel(s, 2, max_iteration=s.Count)
    rec, x = run("--n", "16", "--rule", "csharp", "--compression", "0", ranks=2)
    assert rec["ranks"] == 2 and rec["transport"] == "loopback" and rec["iteration"] == refp["iteration"] == ref["iteration"]
    _check_x(x, refp["x"], dot_order)
    for r in (2, 4):
        rec, x = run("--n", "16", "--mgcg", "--levels", "3" if r == 2 else "2", "--compression", "0", ranks=r)
        m = (M3 if r == 2 else oracle.Multigrid(s, levels=2)).pcg(rule=oracle.RULE_CSHARP, max_iteration=400, offsets=off[r])
        assert rec["ranks"] == r and rec["levels"] == (3 if r == 2 else 2) and rec["iteration"] == m["iteration"]
        _check_x(x, m["x"], dot_order)
    lrefp = ML.pcg(rule=oracle.RULE_CSHARP, max_iteration=400, offsets=off[2])
    rec, x = run("--n", "16", "--mgcg", "--levels", "3", "--linear-transfer", "--compression", "0", ranks=2)
    assert rec["iteration"] == lrefp["iteration"] == lref["iteration"]
    _check_x(x, lrefp["x"], dot_order)
    bad = subprocess.run([exe, "--ranks", "3", "--n", "16"], env=dict(os.environ, MGCG_VIRTUAL_DEVICES="3"), capture_output=True, text=True)
    assert bad.returncode == 1 and "must divide nz" in bad.stderr
    bad = subprocess.run([exe, "--rule", "nonsense"], capture_output=True, text=True)
    assert bad.returncode == 1 and "unknown --rule" in bad.stderr


@pytest.mark.parametrize("dot_order", ["0", "1"])
def test_single_process_ranks_over_rccl_when_the_box_has_two_devices(oracle, tmp_path, dot_order):
    """MgcgCommInitAll's RCCL branch (ncclGroupStart / N x ncclCommInitRank / ncclGroupEnd from ONE process, one host thread per device):
    runs wherever the box has two physical devices -- the shape of the reference's ConjugateGradientParallelGpu on a multi-GPU host.
    With MGCG_DOT_ORDER=1 (the ranks' sums gathered by ncclAllGather and added in rank order) the solution must EQUAL the two-device
    oracle's: a wrong halo plane on first contact with real hardware shows as a bit, not as a tolerance.  Skipped on a one-GPU box."""
    from conjugategradient_amd import _lib

    if _lib.lib().GetDeviceCount() < 2 or os.environ.get("MGCG_VIRTUAL_DEVICES"):
        pytest.skip("needs two physical devices")
    exe = os.path.join(ROOT, "conjugategradient_amd", "host", "mgcg_solve")
    s = problems.poisson(16, 16, 16)
    env = {k: v for k, v in os.environ.items() if k != "MGCG_VIRTUAL_DEVICES"}
    env["MGCG_DOT_ORDER"] = dot_order
    xfile = str(tmp_path / "x.f64")
    off = oracle.partition(s.Count, 2)
    for flags, ref in ((("--rule", "csharp"), oracle.cg_parallel(s, 2, max_iteration=2000)),
                       (("--mgcg", "--levels", "3"), oracle.Multigrid(s, levels=3).pcg(rule=oracle.RULE_CSHARP, max_iteration=400, offsets=off))):
        out = subprocess.run([exe, "--ranks", "2", "--n", "16", "--compression", "0", "--write-x", xfile, *flags], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        rec = json.loads(out.stdout.splitlines()[-1])
        assert rec["ranks"] == 2 and rec["transport"] == "rccl" and rec["iteration"] == ref["iteration"]
        _check_x(np.fromfile(xfile), ref["x"], dot_order)


@pytest.mark.parametrize("devices,dot_order", [(1, "0"), (3, "0"), (1, "1"), (3, "1")])
def test_the_reference_driver_in_full_three_ways(oracle, tmp_path, devices, dot_order):
    """MgcgMain.cs:41-178 in full -- CPU solver, single GPU, every GPU -- with its own element-by-element check (1 % relative against the CPU
    answer, :129-162).  The product's twin host/MgcgMain.cpp is two-way (the library has no CPU compute path); this harness lives under tests/
    because its CPU leg is the reference's ConjugateGradientCpu restated on the test oracle (tests/mgcg_main_three_way.cpp links liboracle).
    On top of the reference's 1 %: the three answers element by element at the north star's 1e-10 (or the oracle's own spread where the forced
    iterations run into the round-off tail), and with MGCG_DOT_ORDER=1 the single-GPU answer EQUALS the CPU answer bit for bit."""
    from conjugategradient_amd import _lib

    exe = str(tmp_path / "mgcg_main_three_way")
    src = os.path.join(ROOT, "tests", "mgcg_main_three_way.cpp")
    libdir = os.path.join(ROOT, "conjugategradient_amd")
    oracle.lib()                                                         # (builds oracle/liboracle.so if needed)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", src, "-o", exe, "-L" + libdir, "-lMgcgGpu", "-L" + os.path.join(ROOT, "oracle"), "-l:liboracle.so",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    count, min_it = 20003, 40
    prefix = str(tmp_path / "x")
    env = dict(os.environ, MGCG_VIRTUAL_DEVICES=str(devices), MGCG_DOT_ORDER=dot_order)
    out = subprocess.run([exe, str(count), str(min_it), "write=" + prefix], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["devices"] == devices and rec["mismatches_single"] == 0 and rec["mismatches_parallel"] == 0        # the reference's own check
    assert rec["iteration_cpu"] == rec["iteration_single"] == rec["iteration_parallel"] == min_it
    assert rec["us_per_iteration_cpu"] > rec["us_per_iteration_single"] > 0 and rec["us_per_iteration_parallel"] > 0   # the three figures of :165-167
    s = problems.mgcg_main(count, 160)
    ref = oracle.cg(s, rule=oracle.RULE_CSHARP, min_iteration=min_it, max_iteration=count)
    x_cpu, x_single, x_par = (np.fromfile(prefix + "." + k + ".f64") for k in ("cpu", "single", "parallel"))
    assert _equal_bits(x_cpu, ref["x"])                                  # the harness's CPU leg IS the oracle loop
    spread = [oracle.cg_parallel(s, w, min_iteration=min_it, max_iteration=count)["x"] for w in (1, 2, 3, 4)]
    _check_x(x_single, x_cpu, dot_order, spread)
    refp = oracle.cg_parallel(s, devices, min_iteration=min_it, max_iteration=count)
    _check_x(x_par, refp["x"], dot_order, spread)
    if dot_order == "1":
        assert rec["residual_single"] == rec["residual_cpu"] == ref["residual"]
