"""MI355X-native CG / MGCG hot path of aokomoriuta/ConjugateGradient behind the reference's C ABI.

``_lib``      ctypes binding of libMgcgGpu.so (include/MgcgGpu.h)
``solver``    LinerEquations / ConjugateGradient / ...SingleGpu / ...ParallelGpu (reference class surface)
``parallel``  one-process-per-GPU driver (RCCL inside the library; torch.distributed bootstrap)
``multigrid`` the V-cycle preconditioner the reference named but never wrote
``problems``  the linear systems the reference hard-codes + the BASELINE.json stencils
"""
__all__ = ["_lib", "solver", "parallel", "multigrid", "problems"]
