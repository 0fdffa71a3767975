import time, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from conjugategradient_amd import problems, _lib
from conjugategradient_amd.solver import ConjugateGradientSingleGpu
from conjugategradient_amd.multigrid import ConjugateGradientMgGpu
for dims in [(256,256,1),(64,64,64),(128,128,128)]:
    s = problems.poisson(*dims)
    cg = ConjugateGradientSingleGpu(s.Count, 7, 0, s.Count, 1e-8, rule=_lib.RULE_CSHARP).load(s)
    cg.Initialize()
    for rep in range(3):
        cg.Initialize()
        t0=time.perf_counter(); cg.Solve(); dt=time.perf_counter()-t0
    print(dims, "CG its", cg.Iteration, "time %.2f ms"%(dt*1e3), "us/it %.1f"%(dt*1e6/(cg.Iteration+1)))
    cg.Dispose()
    if dims[2]>1:
        mg = ConjugateGradientMgGpu(s.Count, 7, 0, 500, 1e-8, dims, levels=3).load(s)
        mg.Initialize(); mg.Setup()
        for rep in range(3):
            mg.Initialize()
            t0=time.perf_counter(); mg.Solve(); dt=time.perf_counter()-t0
        print(dims, "MGCG its", mg.Iteration, "time %.2f ms"%(dt*1e3), "us/it %.1f"%(dt*1e6/(mg.Iteration+1)))
        mg.Dispose()
