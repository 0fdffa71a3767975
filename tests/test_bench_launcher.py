"""bench.py's self-launcher (`python bench.py --gpus N` with no launcher around it): argument, environment, output and
exit-code plumbing, exercised with stand-in rank scripts -- no GPU, no torch in the parent.  The N-GPU timed leg it serves
is the reference's Mgcg/cuBlas/Mgcg/MgcgMain.cs:143-167."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _script(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_ranks_get_the_launcher_environment_and_rank0_line_is_forwarded(tmp_path, capfd):
    s = _script(tmp_path, """
        import json, os, sys
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(r) and os.environ["LOCAL_WORLD_SIZE"] == str(w)
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
        assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        print("chatter from rank", r)
        if r == 0:
            print(json.dumps({"metric": "m", "n_gpus": w, "argv": sys.argv[1:]}))
    """)
    rc = bench.launch_ranks(3, ["--gpus", "3", "--steps", "7"], script=s)
    out, err = capfd.readouterr()
    assert rc == 0
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1                                   # ONE JSON line on stdout, the chatter went to stderr
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 3 and rec["argv"] == ["--gpus", "3", "--steps", "7"]
    assert err.count("chatter from rank") == 3


def test_worst_exit_code_is_returned_and_stragglers_are_ended(tmp_path, capfd):
    s = _script(tmp_path, """
        import os, sys, time
        r = int(os.environ["RANK"])
        if r == 1:
            sys.exit(3)                                      # "RCCL could not form"
        if r == 2:
            time.sleep(600)                                  # a rank left waiting in a collective
        sys.exit(0)
    """)
    rc = bench.launch_ranks(3, [], script=s, grace_s=0.5)
    capfd.readouterr()
    assert rc == 128 + 15                                    # the straggler was terminated (SIGTERM), which outranks 3
    s = _script(tmp_path, "import os, sys; sys.exit(3)")
    assert bench.launch_ranks(2, [], script=s) == 3


def test_clean_exit_without_a_line_is_an_error(tmp_path, capfd):
    s = _script(tmp_path, "pass")
    assert bench.launch_ranks(2, [], script=s) == 1
    assert "no JSON line" in capfd.readouterr().err


def test_bench_parent_stays_gpu_free(tmp_path):
    """`python bench.py --gpus 2` as a program: the parent neither imports torch nor loads the library (checked through
    a sitecustomize hook that refuses both in the parent only) and hands its arguments to the ranks."""
    (tmp_path / "sitecustomize.py").write_text(textwrap.dedent("""
        import builtins, os, sys
        if "WORLD_SIZE" not in os.environ:
            real = builtins.__import__
            def guard(name, *a, **k):
                if name.split(".")[0] in ("torch", "conjugategradient_amd"):
                    raise ImportError("the launcher parent imported " + name)
                return real(name, *a, **k)
            builtins.__import__ = guard
    """))
    stub = _script(tmp_path, """
        import json, os, sys
        if os.environ["RANK"] == "0":
            print(json.dumps({"metric": "stub", "argv": sys.argv[1:]}))
    """)
    env = dict(os.environ, PYTHONPATH=str(tmp_path), MGCG_BENCH_RANK_SCRIPT=stub)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["argv"] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]


def test_a_run_that_never_finishes_is_ended(tmp_path, capfd, monkeypatch):
    monkeypatch.setenv("MGCG_BENCH_TIMEOUT", "1")
    s = _script(tmp_path, "import time; time.sleep(600)")
    assert bench.launch_ranks(2, [], script=s, grace_s=0.2) == 128 + 15
    assert "did not finish" in capfd.readouterr().err


def test_a_blocked_step_of_a_rank_ends_that_rank_with_a_message(tmp_path):
    """The per-step watchdog of the ranks (N > 1: RCCL bootstrap, warm-up, timed steps): a step that never returns ends the rank with
    status 4 and says where it was, instead of holding the job until the launcher's limit."""
    code = textwrap.dedent(f"""
        import os, sys, time
        sys.path.insert(0, {ROOT!r})
        os.environ["MGCG_BENCH_STEP_TIMEOUT"] = "0.3"
        import bench
        with bench._watchdog("rank 3: ncclCommInitRank"):
            pass                                            # a step that finishes: the timer is cancelled
        time.sleep(0.6)
        print("still here", flush=True)
        with bench._watchdog("rank 3: the warm-up iterations", enabled=False):
            time.sleep(0.6)                                 # N = 1: no watchdog
        with bench._watchdog("rank 3: the timed steps"):
            time.sleep(30)
        print("not reached")
    """)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 4
    assert "still here" in out.stdout and "not reached" not in out.stdout
    assert "rank 3: the timed steps did not finish" in out.stderr and "status 4" in out.stderr


def test_the_launcher_refuses_to_start_ranks_under_a_profiler(tmp_path, capfd, monkeypatch):
    """Under rocprofv3 the profiler's preloaded library has initialised the GPU in the parent: starting rank processes from it is the
    exec-after-GPU-init hop this pool forbids.  launch_ranks says so and starts nothing."""
    s = _script(tmp_path, "import pathlib, os; pathlib.Path(os.environ['MARK']).write_text('started')")
    mark = tmp_path / "mark"
    monkeypatch.setenv("MARK", str(mark))
    monkeypatch.setenv("ROCPROFILER_LIBRARY_CTOR", "1")
    assert bench.launch_ranks(2, [], script=s) == 5
    assert "refuses to start its ranks under a profiler" in capfd.readouterr().err and not mark.exists()
    monkeypatch.delenv("ROCPROFILER_LIBRARY_CTOR")
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert bench.launch_ranks(2, [], script=s) == 5 and not mark.exists()


def test_byte_counts_of_the_line():
    """The closed-form byte counts bench.py quotes: SURVEY.md section 8d's figures for 512^3, and the relation between the SURVEY-formula
    count and the count the library's schedule must move."""
    n, N = 512, 512**3
    nnz = 7 * N - 6 * n * n
    assert 12 * nnz + 4 * (N + 1) + 16 * N == 13_939_769_348                  # SpMV, SURVEY 8d
    v, shell = bench.vcycle_bytes(n, 3, 1, 4)
    assert shell == 13_939_769_348 + 72 * N                                    # 23.60 GB per CG iteration
    rv, rshell = bench.vcycle_required_bytes(n, 3, 1, 4)
    assert rshell == shell and rv < v                                          # no D^-1 array, folded first sweep and prolongation
    rv_nofold, _ = bench.vcycle_required_bytes(n, 3, 1, 4, fold=False)
    assert rv < rv_nofold
    assert abs((rv + rshell) - 60.43e9) < 0.01e9


def test_a_blocked_extra_ends_the_run_with_the_line_as_far_as_it_got():
    """The guard around the untimed extras of an N > 1 line (config 4, comm_probe, schedules): a stage that never returns -- a collective
    waiting for a rank that is gone -- must not cost the timed result.  Rank 0 prints the line it already has with `extras_aborted`,
    every rank leaves with status 0; a stage that finishes cancels its timer."""
    code = textwrap.dedent(f"""
        import json, os, sys, time
        sys.path.insert(0, {ROOT!r})
        os.environ["MGCG_BENCH_EXTRAS_TIMEOUT"] = "0.3"
        import bench
        rank = int(sys.argv[1])
        out = {{"metric": "m", "value": 123.0}} if rank == 0 else {{}}
        g = bench._ExtrasGuard(rank, out)
        g.stage("comm_probe")
        out["comm_probe"] = {{"allreduce_8B_us": 25.0}}
        g.stage("cg schedules")                         # the previous stage finished in time: its timer is gone
        time.sleep(0.1)
        g.stage("mgcg: partitioned iterations and solve")
        time.sleep(30)                                  # blocked
        print("not reached")
    """)
    for rank in (0, 1):
        out = subprocess.run([sys.executable, "-c", code, str(rank)], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0 and "not reached" not in out.stdout
        assert "extras stage 'mgcg: partitioned iterations and solve' did not finish" in out.stderr
        if rank == 0:
            rec = json.loads(out.stdout.strip().splitlines()[-1])
            assert rec["value"] == 123.0 and rec["comm_probe"]["allreduce_8B_us"] == 25.0
            assert rec["extras_aborted"] == "mgcg: partitioned iterations and solve"
        else:
            assert out.stdout.strip() == ""


def test_cpu_baseline_leg_reports_the_oracle_in_both_summation_orders():
    """bench.py's cpu_baseline leg on a small grid (the only place outside tests/ that may run the oracle): the reference-order figure, the
    exactly summed yardstick next to it (oracle_set_dot_mode, switched back afterwards), the bounded-sample bookkeeping."""
    from oracle import oracle as O

    cb = bench.cpu_baseline(24, 3)
    assert cb["cores"] == 1 and cb["kind"] == "port" and cb["grid"] == 24 and cb["iterations"] == 3 and cb["value"] > 0
    assert cb["residual"] > 0 and abs(cb["residual"] - cb["residual_with_compensated_dots"]) <= 1e-12 * cb["residual"]
    assert O.lib().oracle_get_dot_mode() == 0
    assert "24^3" in cb["sample"] and cb["all_cores_variant"].get("value", 1) > 0
