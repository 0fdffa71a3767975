"""Worker of tests/test_gpu_parallel.py::test_phased_ranks_share_the_gpu_over_gloo (one rank per process)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch.distributed as dist

    from conjugategradient_amd import problems as P
    from conjugategradient_amd.parallel import HipPhases, PhasedRankSolver, RankPartition
    from conjugategradient_amd.solver import SparseMatrix

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    system = P.mgcg_main(2400, 160)
    part = RankPartition.of(system.Count, world, rank, system.RowOffsets)
    backend = HipPhases(SparseMatrix.from_system(system), system.x, system.b, part, 160)
    solver = PhasedRankSolver(backend, part, 0, system.Count, 1e-8, dist=dist)
    solver.Solve()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=backend.read_x(), iteration=solver.Iteration, residual=solver.Residual,
             offset=part.offset, count=part.count)
    dist.barrier()
    dist.destroy_process_group()
    # normal interpreter exit (one HIP runtime per process: conjugategradient_amd/tools/exit_probe.py)


if __name__ == "__main__":
    main()
