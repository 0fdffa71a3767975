"""bench.py as a program on the GPU box: the N = 1 line's contract fields, and the self-launched N = 2 run over the host-staged
transport (two processes share the one GPU) with its self-check against a single-rank replay of the same iterations."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]                      # ONE JSON line
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    rec = _run("--grid", "128", "--steps", "20", "--warmup", "5", "--no-extras", "--cpu-iters", "1")
    assert rec["n_gpus"] == 1 and rec["steps"] == 20 and rec["warmup"] == 5 and rec["unit"] == "iterations/s" and rec["dtype"] == "f64"
    assert rec["higher_is_better"] is True and rec["data"] == "synthetic" and rec["vs_baseline"] is None and "workload" in rec["config"]
    assert abs(rec["value"] - 1e3 / rec["ms_per_step"]) <= 1e-6 * rec["value"]
    rf = rec["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["launches_timed"] == 20 and rf["algorithmic_bytes_per_launch"] == 12 * rec["config"]["nnz"] + 4 * (rec["config"]["rows"] + 1) + 16 * rec["config"]["rows"]
    cb = rec["cpu_baseline"]
    assert cb["cores"] == 1 and cb["kind"] == "port" and cb["value"] > 0 and "sample" in cb
    # the HIP loop against the oracle on the same iterations of the same full-size system, in the line itself (north star: 1e-10)
    # (against the oracle with exactly summed dot products: 1e-10; against the reference's serial order: within that order's own rounding)
    assert cb["gpu_vs_compensated_oracle_relative_difference"] <= 1e-10 and cb["gpu_within_reference_rounding"] is True, cb
    # ... and in the validation mode (dot_order = 1) the same iterations give the oracle's residual bit for bit
    assert cb["gpu_equals_oracle_in_reference_order_mode"] is True and cb["gpu_residual_in_reference_order_mode"] == cb["residual"], cb
    assert rf["traffic"] is None or "RECORDED" in rf["traffic_source"]


@pytest.mark.parametrize("solver", ["cg", "mgcg"])
def test_two_self_launched_ranks_check_themselves_against_one_rank(solver):
    rec = _run("--gpus", "2", "--grid", "64", "--steps", "8", "--warmup", "2", "--allow-fallback", "--solver", solver)
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong"
    if "rccl" not in rec["config"]["transport"].lower() or "fallback" in rec["config"]["transport"].lower():
        assert "not a scaling result" in rec["config"]["transport"]             # one GPU: the host-staged transport, and the line says so
    par = rec["parity_vs_single_rank"]
    assert par["within_1e-10"] is True and par["relative_difference"] <= 1e-10, par
    assert par["partitioned_residual_after_steps"] == rec["residual_after_steps"]
    pr = rec["per_rank"]
    assert len(pr["seconds_for_the_timed_steps"]) == 2 and len(pr["spmv_avg_launch_ms"]) == 2
    assert abs(max(pr["seconds_for_the_timed_steps"]) - rec["ms_per_step"] * rec["steps"] / 1e3) <= 1e-9       # the line reports the slowest rank
    assert par["single_rank_ms_per_step"] > 0 and par["speedup_vs_single_rank"] > 0
    if solver == "cg":
        _check_multirank_extras(rec, 2)


def _check_multirank_extras(rec, world):
    """The N > 1 line of the driver's command explains itself: config 4 (row-partitioned MGCG) with its own self-check and 1-GPU leg,
    the communicator's prices, the three halo schedules next to the library's choice (Mgcg/cuBlas/Mgcg/MgcgMain.cs:143-167 times the
    1-GPU and N-GPU legs in one run; ConjugateGradientParallelGpu.cs:384-419,463,499,525 are the steps priced)."""
    assert "extras_aborted" not in rec, rec.get("extras_aborted")
    cp = rec["comm_probe"]
    for k in ("allreduce_8B_us", "allreduce_16B_us", "fork_join_us", "kernel_boundary_us"):
        assert cp[k] is not None and cp[k] >= 0, (k, cp)
    for k in ("neighbour_exchange_one_plane_us", "neighbour_exchange_8B_us"):      # timed on RCCL only: a host-staged transport reports null, never a near-zero figure
        assert (cp[k] is not None and cp[k] > 0) if cp["transport"] == "rccl" else cp[k] is None, (k, cp)
    assert cp["fork_join_us"] > 0 and cp["kernel_boundary_us"] > 0 and cp["plane_bytes"] == 8 * round(rec["config"]["rows"] ** (1 / 3)) ** 2
    sc = rec["schedules"]
    for k in ("exchange_in_line", "interior_rows_on_side_stream", "exchange_on_side_stream", "library_default"):
        assert sc[k]["ms_per_iteration"] > 0, (k, sc)
    assert sc["exchange_in_line"]["overlap_active_rank0"] is False and sc["interior_rows_on_side_stream"]["overlap_active_rank0"] is True
    mg = rec["mgcg"]
    assert mg["ms_per_iteration"] > 0 and mg["iterations_to_1e-8"] > 0 and mg["solve_s"] > 0
    par = mg["parity_vs_single_rank"]
    assert par["within_1e-10"] is True and par["same_iterations_to_1e-8"] is True, par
    assert mg["single_rank"]["iterations_to_1e-8"] == mg["iterations_to_1e-8"]
    assert mg["speedup_vs_single_rank"]["solver_time"] > 0 and mg["speedup_vs_single_rank"]["per_iteration"] > 0
    for k in ("exchange_in_line", "interior_rows_on_side_stream", "exchange_on_side_stream", "library_default", "per_pass_exchanges"):
        assert mg["schedules"][k]["ms_per_iteration"] > 0, (k, mg["schedules"])
    # the deep-halo cycle (bit 2 of the folds) is the default wherever the slabs are thick enough; the per-pass schedule never has it
    assert not mg["schedules"]["per_pass_exchanges"]["folds_rank0"] & 4
    planes = round(rec["config"]["rows"] ** (1 / 3)) // world
    assert bool(mg["folds_rank0"] & 4) == (planes // 2 >= 2 and planes // 4 >= 4 and os.environ.get("MGCG_DEEP_HALO", "1") != "0"), (mg["folds_rank0"], planes)


@pytest.mark.parametrize("solver", ["cg", "mgcg"])
def test_two_ranks_over_rccl_when_the_box_has_two_devices(solver):
    """The same self-launched run WITHOUT --allow-fallback: two ranks on two devices form a real RCCL communicator (the one step no
    one-GPU box can take), and the line's self-check says whether the partitioned loop computed what one rank computes.  Skipped on a
    box with one device."""
    from conjugategradient_amd import _lib

    if _lib.lib().GetDeviceCount() < 2 or os.environ.get("MGCG_VIRTUAL_DEVICES"):
        pytest.skip("needs two physical devices")
    rec = _run("--gpus", "2", "--grid", "128", "--steps", "10", "--warmup", "3", "--solver", solver)
    assert rec["n_gpus"] == 2 and rec["config"]["transport"] == "rccl"
    par = rec["parity_vs_single_rank"]
    assert par["within_1e-10"] is True, par
    if solver == "cg":
        _check_multirank_extras(rec, 2)
        assert rec["comm_probe"]["transport"] == "rccl" and rec["comm_probe"]["allreduce_8B_us"] > 0 and rec["comm_probe"]["neighbour_exchange_one_plane_us"] > 0
