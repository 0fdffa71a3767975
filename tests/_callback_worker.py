"""Worker of tests/test_gpu_parallel.py::test_native_loop_over_the_callback_transport (one rank per process): the native
multi-rank loop (SolveParallel / SolveMgParallel) with its collectives carried by torch.distributed gloo on host memory."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out_dir, which = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    import torch.distributed as dist

    from conjugategradient_amd import _lib, problems as P
    from conjugategradient_amd.parallel import ConjugateGradientMgRankGpu, ConjugateGradientRankGpu, create_callback_comm

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = create_callback_comm(rank, world)
    if which == "unstructured":
        # few couplings per row, scattered over the whole range: the halo plan turns into per-peer index lists
        import scipy.sparse as sp
        n = 6000
        rng = np.random.default_rng(5)
        i = rng.choice(n, size=n // 3, replace=False)
        j = rng.integers(0, n, size=i.size)
        keep = i != j
        U = sp.coo_matrix((-rng.random(int(keep.sum())), (i[keep], j[keep])), shape=(n, n)).tocsr()
        A = (U + U.T).tocsr()
        A = (A + sp.diags(1.0 + np.asarray(abs(A).sum(axis=1)).ravel())).tocsr()
        A.sort_indices()
        b = np.cos(np.arange(n) * 0.3) * (1.0 + np.arange(n) % 5)
        system = P.LinearSystem(A.data.astype(np.float64), A.indices.astype(np.int32), A.indptr.astype(np.int32), np.zeros(n), b, "sparse-unstructured")
        cg = ConjugateGradientRankGpu(system.Count, int(np.diff(system.RowOffsets).max()), 0, system.Count, 1e-8, rank=rank, world=world, comm=comm, device=0).load(system)
        cg.Initialize()
    elif which == "banded":
        system = P.mgcg_main(2400, 160)
        cg = ConjugateGradientRankGpu(system.Count, 160, 0, system.Count, 1e-8, rank=rank, world=world, comm=comm, device=0).load(system)
        cg.Initialize()
    else:
        system = P.poisson(16, 16, 16)
        system.b[:] = np.random.default_rng(3).standard_normal(system.Count)
        cg = ConjugateGradientMgRankGpu(system.Count, 7, 0, 400, 1e-8, system.grid, rank=rank, world=world, comm=comm, device=0).load(system)
        _lib.lib().MgcgSetMatrixCompression(cg.cusparse, 1)
        cg.Initialize()
        cg.Setup()
    cg.Solve()
    cg.Read()
    p = cg.part
    import ctypes as C
    vol = (C.c_longlong * 2)(0, 0)
    lists = _lib.lib().MgcgLastHalo(vol)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=cg.x[p.offset: p.offset + p.count], iteration=cg.Iteration, residual=cg.Residual,
             offset=p.offset, count=p.count, halo_lists=lists, halo_moved=int(vol[0]), halo_contiguous=int(vol[1]))
    cg.Dispose()                     # handles and vectors go before the process group the callbacks use
    _lib.lib().MgcgCommDestroy(comm)
    dist.barrier()
    dist.destroy_process_group()
    # normal interpreter exit: torch (imported first) and libMgcgGpu.so share ONE HIP runtime -- the library's DT_NEEDED
    # libamdhip64.so.7 is satisfied by the copy torch already mapped (conjugategradient_amd/tools/exit_probe.py)


if __name__ == "__main__":
    main()
