#!/usr/bin/env python3
"""Does a process that holds torch's HIP runtime AND libMgcgGpu.so exit cleanly?  Prints the mapped HIP / RCCL libraries
and leaves through the normal interpreter teardown (the test workers used os._exit to skip it)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode = sys.argv[1] if len(sys.argv) > 1 else "torch-first"
if mode == "torch-first":
    import torch
    import torch.distributed  # noqa: F401
    if len(sys.argv) > 2 and sys.argv[2] == "cuda":
        torch.zeros(4, device="cuda").sum().item()
import numpy as np  # noqa: E402

from conjugategradient_amd import _lib, problems  # noqa: E402
from conjugategradient_amd.solver import ConjugateGradientSingleGpu  # noqa: E402

s = problems.poisson(12, 12, 12)
cg = ConjugateGradientSingleGpu(s.Count, 7, 0, 500, 1e-8).load(s)
cg.Initialize()
cg.Solve()
cg.Read()
print("iterations", cg.Iteration + 1, "residual", cg.Residual)
if len(sys.argv) > 3 and sys.argv[3] == "nodispose":
    pass
else:
    cg.Dispose()
libs = sorted({line.split()[-1] for line in open("/proc/self/maps") if any(k in line for k in ("amdhip", "rccl", "hsa-runtime", "MgcgGpu"))})
print("\n".join(libs))
sys.stdout.flush()
