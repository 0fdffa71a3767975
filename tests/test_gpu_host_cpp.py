"""The C++ host twin of the reference classes (host/Mgcg.hpp) and its MgcgMain driver, run as a program."""
import json
import os
import subprocess

import numpy as np
import pytest

from conjugategradient_amd import problems

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "conjugategradient_amd", "host", "MgcgMain")


@pytest.mark.parametrize("devices,balance", [(1, False), (3, False), (3, True)])
def test_mgcg_main_driver(oracle, devices, balance):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(EXE)])
    count, min_it = 20003, 40
    env = dict(os.environ, MGCG_VIRTUAL_DEVICES=str(devices))
    out = subprocess.run([EXE, str(count), str(min_it)] + (["balance"] if balance else []), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    s = problems.mgcg_main(count, 160)
    # BalanceNonzeros (not in the reference): the C++ class cuts where problems.partition_offsets(..., "nnz") does -- the short rows at both
    # ends of the band move the cuts off floor(count / devices)
    assert rec["offsets"] == problems.partition_offsets(count, devices, s.RowOffsets, "nnz" if balance else "rows")
    if balance:
        assert rec["offsets"] != problems.partition_offsets(count, devices)
    ref = oracle.cg(s, rule=oracle.RULE_NATIVE, min_iteration=min_it, max_iteration=count, hard_cap=count + 5)
    assert rec["devices"] == devices
    assert rec["iteration_single"] == ref["iteration"] == rec["iteration_parallel"] == rec["iteration_phases"] == min_it
    assert rec["mismatches"] == 0 and rec["max_rel_single_vs_parallel"] < 1e-8 and rec["max_rel_phases_vs_parallel"] < 1e-8
    # ConjugateGradientParallelGpu.Solve() took the native loop (SolveParallel per device thread), not the host-driven phases
    assert rec["parallel_path"] == "native loop (SolveParallel over %s)" % ("single" if devices == 1 else "loopback")
    w = (np.arange(count) % 7) + 1.0
    assert abs(rec["checksum"] - float(np.dot(ref["x"], w))) <= 1e-9 * abs(rec["checksum"])
    assert abs(rec["x0"] - ref["x"][0]) <= 1e-10 * abs(ref["x"][0]) and abs(rec["xlast"] - ref["x"][-1]) <= 1e-10 * abs(ref["x"][-1])


def test_class_keeps_working_on_the_phases_when_no_communicator_forms(oracle):
    """A host whose RCCL cannot form a communicator (MGCG_FAIL_COMM_INIT: the test hook that makes MgcgCommInitAll report failure):
    ConjugateGradientParallelGpu must still construct and solve -- on the reference's own host-driven phases
    (ConjugateGradientParallelGpu.cs:424-565), which need no communicator -- and say so."""
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(EXE)])
    count, min_it = 6007, 30
    env = dict(os.environ, MGCG_VIRTUAL_DEVICES="3", MGCG_FAIL_COMM_INIT="1")
    out = subprocess.run([EXE, str(count), str(min_it)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["parallel_path"].startswith("host-driven phases (Solve0..3) -- no communicators:") and "MGCG_FAIL_COMM_INIT" in rec["parallel_path"]
    s = problems.mgcg_main(count, 160)
    ref = oracle.cg(s, rule=oracle.RULE_NATIVE, min_iteration=min_it, max_iteration=count, hard_cap=count + 5)
    assert rec["iteration_single"] == rec["iteration_parallel"] == ref["iteration"] == min_it and rec["mismatches"] == 0
    w = (np.arange(count) % 7) + 1.0
    assert abs(rec["checksum"] - float(np.dot(ref["x"], w))) <= 1e-9 * abs(rec["checksum"])


def test_other_two_driver_families_in_cpp(oracle):
    """host/MgcgFrontends.hpp + MgcgCLMain: the HandmadeCL ELL builder with the max-norm rule and the ViennaCL dictionary
    builder with the relative rule, filled by the reference drivers' loops in C++, against the oracle's same rules."""
    exe = os.path.join(ROOT, "conjugategradient_amd", "host", "MgcgCLMain")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(exe)])
    count = 2345
    out = subprocess.run([exe, str(count)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rec = {l.split()[0]: l.split()[1:] for l in out.stdout.splitlines() if l.split() and l.split()[0] in ("handmadecl", "viennacl")}
    s = problems.mgcg_main(count, 160)
    ref = oracle.cg(s, rule=oracle.RULE_HANDMADECL, allowable_residual=1e-4, min_iteration=50, max_iteration=count)
    assert int(rec["handmadecl"][0]) == ref["iteration"]
    assert abs(float(rec["handmadecl"][2]) - ref["x"].sum()) <= 1e-9 * np.abs(ref["x"]).sum()
    v = problems.viennacl_main(count, 160)
    ref = oracle.cg(v, rule=oracle.RULE_VIENNACL, allowable_residual=1e-4, min_iteration=0, max_iteration=count, hard_cap=count + 10)
    assert int(rec["viennacl"][0]) == ref["iteration"] + 1
    assert abs(float(rec["viennacl"][2]) - ref["x"].sum()) <= 1e-9 * np.abs(ref["x"]).sum()


def test_command_line_driver(oracle):
    """host/mgcg_solve.cpp: flags instead of the reference's compile-time constants; CG and MGCG on a Poisson grid, each
    against the oracle's iteration count and solution sum."""
    exe = os.path.join(ROOT, "conjugategradient_amd", "host", "mgcg_solve")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(exe)])

    def run(*flags):
        out = subprocess.run([exe, *flags], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        return json.loads(out.stdout.splitlines()[-1])

    s = problems.poisson(16, 16, 16)
    ref = oracle.cg(s, rule=oracle.RULE_CSHARP, max_iteration=s.Count)
    rec = run("--n", "16", "--rule", "csharp", "--compression", "0")
    assert rec["iteration"] == ref["iteration"] == 43                       # SURVEY.md section 8c scratch count
    assert abs(rec["sum_x"] - ref["x"].sum()) <= 1e-9 * np.abs(ref["x"]).sum()
    mref = oracle.Multigrid(s, levels=3).pcg(rule=oracle.RULE_CSHARP, max_iteration=400)
    rec = run("--n", "16", "--mgcg", "--levels", "3", "--compression", "1")
    assert rec["levels"] == 3 and rec["iteration"] == mref["iteration"]
    assert abs(rec["sum_x"] - mref["x"].sum()) <= 1e-9 * np.abs(mref["x"]).sum()
    lref = oracle.Multigrid(s, levels=3, interpolation=1).pcg(rule=oracle.RULE_CSHARP, max_iteration=400)
    rec = run("--n", "16", "--mgcg", "--levels", "3", "--linear-transfer", "--compression", "0")
    assert rec["iteration"] == lref["iteration"] < mref["iteration"]
    assert abs(rec["sum_x"] - lref["x"].sum()) <= 1e-9 * np.abs(lref["x"]).sum()
    # --ranks R: R devices of ONE process, one host thread each, MgcgCommInitAll + SolveParallel / MgSetupParallel + SolveMgParallel
    # (the single-process shape of ConjugateGradientParallelGpu.cs:264-324); same counts and sums as the single-rank runs
    def run_ranks(r, *flags):
        out = subprocess.run([exe, "--ranks", str(r), *flags], env=dict(os.environ, MGCG_VIRTUAL_DEVICES=str(r)), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        return json.loads(out.stdout.splitlines()[-1])

    rec = run_ranks(2, "--n", "16", "--rule", "csharp", "--compression", "0")
    assert rec["ranks"] == 2 and rec["transport"] == "loopback" and rec["iteration"] == ref["iteration"]
    assert abs(rec["sum_x"] - ref["x"].sum()) <= 1e-9 * np.abs(ref["x"]).sum()
    for r in (2, 4):
        rec = run_ranks(r, "--n", "16", "--mgcg", "--levels", "3" if r == 2 else "2", "--compression", "0")
        m = mref if r == 2 else oracle.Multigrid(s, levels=2).pcg(rule=oracle.RULE_CSHARP, max_iteration=400)
        assert rec["ranks"] == r and rec["levels"] == (3 if r == 2 else 2) and rec["iteration"] == m["iteration"]
        assert abs(rec["sum_x"] - m["x"].sum()) <= 1e-9 * np.abs(m["x"]).sum()
    rec = run_ranks(2, "--n", "16", "--mgcg", "--levels", "3", "--linear-transfer", "--compression", "0")
    assert rec["iteration"] == lref["iteration"] and abs(rec["sum_x"] - lref["x"].sum()) <= 1e-9 * np.abs(lref["x"]).sum()
    bad = subprocess.run([exe, "--ranks", "3", "--n", "16"], env=dict(os.environ, MGCG_VIRTUAL_DEVICES="3"), capture_output=True, text=True)
    assert bad.returncode == 1 and "must divide nz" in bad.stderr
    bad = subprocess.run([exe, "--rule", "nonsense"], capture_output=True, text=True)
    assert bad.returncode == 1 and "unknown --rule" in bad.stderr


def test_single_process_ranks_over_rccl_when_the_box_has_two_devices(oracle):
    """MgcgCommInitAll's RCCL branch (ncclGroupStart / N x ncclCommInitRank / ncclGroupEnd from ONE process, one host thread per device):
    runs wherever the box has two physical devices -- the shape of the reference's ConjugateGradientParallelGpu on a multi-GPU host.
    Skipped on a one-GPU box."""
    from conjugategradient_amd import _lib

    if _lib.lib().GetDeviceCount() < 2 or os.environ.get("MGCG_VIRTUAL_DEVICES"):
        pytest.skip("needs two physical devices")
    exe = os.path.join(ROOT, "conjugategradient_amd", "host", "mgcg_solve")
    s = problems.poisson(16, 16, 16)
    env = {k: v for k, v in os.environ.items() if k != "MGCG_VIRTUAL_DEVICES"}
    for flags, ref in ((("--rule", "csharp"), oracle.cg(s, rule=oracle.RULE_CSHARP, max_iteration=2000)),
                       (("--mgcg", "--levels", "3"), oracle.Multigrid(s, levels=3).pcg(rule=oracle.RULE_CSHARP, max_iteration=400))):
        out = subprocess.run([exe, "--ranks", "2", "--n", "16", "--compression", "0", *flags], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        rec = json.loads(out.stdout.splitlines()[-1])
        assert rec["ranks"] == 2 and rec["transport"] == "rccl" and rec["iteration"] == ref["iteration"]
        assert abs(rec["sum_x"] - ref["x"].sum()) <= 1e-9 * np.abs(ref["x"]).sum()
