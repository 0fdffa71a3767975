#!/usr/bin/env python3
"""What one rank of the 8-GPU run will see, measured on ONE GPU.

The per-rank problem of BASELINE config 4 is a 64-plane slab of the 512^3 grid (16.8 M rows).  This tool times
  A. that slab as a stand-alone 512 x 512 x 64 system on one rank (no collectives: the pure compute of a rank),
  B. the same slab split over 2 loopback ranks of 32 planes (host threads on MGCG_VIRTUAL_DEVICES=2 sharing the GPU): the
     multi-rank code path -- halo plan, halo exchange, all-reduces, interior/boundary split -- with the loopback transport's
     host-staged collectives standing in for RCCL (each is a D2H copy + thread barrier + H2D copy: slower than an RCCL
     all-reduce over xGMI, so B - A is an upper bound of the per-iteration overhead),
  C. the slab on one rank through the several-ranks code path with a real one-rank RCCL communicator (MGCG_FORCE_MULTIRANK): every
     launch, collective call and event the path adds, on the device's own stream, without B's host synchronisations (and without
     the xGMI wire time, which a one-GPU box cannot show); plus MgcgCommProbe's per-step prices,
for plain CG and for the 3-level MGCG, and prints one JSON object with the implied 8-GPU ceiling
T(512^3, 1 GPU) / (T_slab + overhead).  Run under `rocprofv3 --kernel-trace --stats` to get launches per iteration."""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=512)
    ap.add_argument("--planes", type=int, default=64)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--full", type=int, default=512, help="n of the n^3 single-GPU reference run (0: skip)")
    ap.add_argument("--skip-mg", action="store_true")
    ap.add_argument("--skip-forced", action="store_true", help="skip section C (one-rank RCCL communicator on the several-ranks path)")
    a = ap.parse_args()
    os.environ["MGCG_VIRTUAL_DEVICES"] = "2"
    from conjugategradient_amd import _lib
    from conjugategradient_amd.parallel import ConjugateGradientMgRankGpu, ConjugateGradientRankGpu

    L = _lib.lib()
    _lib.require_gpu()
    nx, nz = a.nx, a.planes
    N = nx * nx * nz
    out = {"slab": f"{nx} x {nx} x {nz}", "rows": N, "steps": a.steps}

    def timed(fn):
        L.MgcgDeviceSynchronize()
        t0 = time.perf_counter()
        fn()
        L.MgcgDeviceSynchronize()
        return (time.perf_counter() - t0) * 1e3 / a.steps

    def make(kind, rank, world, comm, dims):
        n = dims[0] * dims[1] * dims[2]
        if kind == "mgcg":
            cg = ConjugateGradientMgRankGpu(n, 7, 0, 10**9, 1e300, dims, rank=rank, world=world, comm=comm, device=rank,
                                            rule=_lib.RULE_NATIVE, levels=3, nu=1, nuCoarse=4)
        else:
            cg = ConjugateGradientRankGpu(n, 7, 0, 10**9, 1e-8, rank=rank, world=world, comm=comm, device=rank)
        cg.InitializePoisson(*dims)
        if kind == "mgcg":
            cg.Setup()
        return cg

    def steps(kind, cg, k, restart):
        if kind == "mgcg":
            L.MgcgFill(cg.vectorX.Ptr, 0.0)
            cg.MinIteration = k - 1
            cg.Solve()
        else:
            cg.Steps(k, restart=restart)

    kinds = ["cg"] + ([] if a.skip_mg else ["mgcg"])
    rccl1 = None
    if not a.skip_forced:
        import ctypes as C

        L.SetDevice(0)
        buf = (C.c_char * 128)()
        if L.MgcgCommGetUniqueId(buf) != 0:
            raise SystemExit("no RCCL: " + _lib.last_error())
        rccl1 = L.MgcgCommInitRank(buf, 1, 0)
        _lib.check("MgcgCommInitRank")
        # device-side price of each step the several-ranks path adds, back to back on the communicator's stream (HIP events)
        probe = {}
        for name, what, count in (("allreduce_8B_us", 0, 1), ("allreduce_16B_us", 0, 2), ("self_send_recv_one_plane_us", 1, nx * nx), ("self_send_recv_8B_us", 1, 1),
                                  ("fork_join_us", 2, 0), ("kernel_boundary_us", 3, 0)):
            probe[name] = L.MgcgCommProbe(rccl1, what, count, 200)
        out["one_rank_rccl_probes"] = probe
    for kind in kinds:
        # A: one rank, the whole slab
        cg = make(kind, 0, 1, None, (nx, nx, nz))
        steps(kind, cg, 10, True)
        msA = timed(lambda: steps(kind, cg, a.steps, False))
        cg.Dispose()
        # B: two loopback ranks of nz / 2 planes
        group = L.MgcgLoopbackCreate(2)
        res, errs = [None, None], []
        bar = threading.Barrier(2)

        def worker(rank):
            try:
                L.SetDevice(rank)
                comm = L.MgcgCommInitLoopback(group, rank)
                c = make(kind, rank, 2, comm, (nx, nx, nz))
                steps(kind, c, 10, True)
                bar.wait()
                ms = timed(lambda: steps(kind, c, a.steps, False))
                active = c.LastOverlap()[0] if kind == "cg" else None
                bar.wait()
                res[rank] = (ms, active)
                c.Dispose()
                L.MgcgCommDestroy(comm)
            except Exception as ex:     # noqa: BLE001
                errs.append(ex)
                try:
                    bar.abort()
                except Exception:       # noqa: BLE001
                    pass

        ts = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        L.MgcgLoopbackDestroy(group)
        if errs:
            raise errs[0]
        msB = max(r[0] for r in res)
        out[kind] = {"one_rank_ms_per_iteration": msA, "two_loopback_ranks_ms_per_iteration": msB,
                     "overhead_upper_bound_ms": msB - msA, "halo_overlap_active": res[0][1]}
        # C: the whole slab on ONE rank again, but through the several-ranks code path with a REAL one-rank RCCL communicator
        #    (MGCG_FORCE_MULTIRANK = one grid plane: reduction launches + ncclAllReduce on the stream, the fold behind it, a grouped
        #    ncclSend/ncclRecv of one plane to itself, fork / join, interior + boundary row ranges) -- no host synchronisation inside
        #    an iteration, unlike B.  Every schedule runs in a FRESH process (tools/forced_path_run.py), the plain single-rank loop too:
        #    forced - plain = the device-side cost the path adds per iteration, without the xGMI wire time.
        if not a.skip_forced:
            import subprocess

            exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "forced_path_run.py")

            def fresh(*flags):
                o = subprocess.run([sys.executable, exe, "--nx", str(nx), "--planes", str(nz), "--steps", str(a.steps), "--repeats", "3", "--solver", kind, *flags],
                                   capture_output=True, text=True, timeout=600)
                if o.returncode != 0:
                    raise SystemExit(o.stdout[-2000:] + o.stderr[-2000:])
                return json.loads([l for l in o.stdout.splitlines() if l.startswith("{")][-1])

            plain = fresh("--force", "0")["ms_per_iteration"]
            rows = []
            for overlap, halo_stream in ((0, 0), (2, 0), (2, 1), (1, 0), (1, 1)):      # overlap 1 + halo_stream 0 = the library's defaults
                r = fresh("--overlap", str(overlap), "--halo-stream", str(halo_stream))
                rows.append({"overlap": overlap, "schedule": "exchange in line" if overlap == 0 else (("overlap where the measured exchange costs more than the hops that hide it (round 4's default rule, MGCG_OVERLAP=1): " if overlap == 1 else "overlap on every level: ") + (
                                 "exchange on the side stream, rows on the main stream (MGCG_HALO_STREAM=1)" if halo_stream else "interior rows on the side stream, every RCCL call on the main stream (default)")),
                             "ms_per_iteration": r["ms_per_iteration"], "added_us_vs_plain": 1e3 * (r["ms_per_iteration"] - plain), "halo_overlap_active": r["halo_overlap_active"],
                             "measured_exchange_in_line_us": r.get("measured_exchange_in_line_us"), "measured_fork_launch_join_us": r.get("measured_fork_launch_join_us")})
            out[kind]["fresh_process_plain_ms_per_iteration"] = plain
            out[kind]["one_rank_rccl_on_the_several_ranks_path"] = rows
    if a.full:
        n = a.full
        for kind in kinds:
            cg = make(kind, 0, 1, None, (n, n, n))
            steps(kind, cg, 10, True)
            ms = timed(lambda: steps(kind, cg, a.steps, False))
            cg.Dispose()
            out[kind]["full_grid_one_gpu_ms_per_iteration"] = ms
            slab, ov = out[kind]["one_rank_ms_per_iteration"], max(out[kind]["overhead_upper_bound_ms"], 0.0)
            out[kind]["implied_8gpu_speedup_no_overhead"] = ms / slab
            out[kind]["implied_8gpu_speedup_with_loopback_overhead"] = ms / (slab + ov)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
