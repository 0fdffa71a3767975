"""Regenerates the golden fixtures in this directory.  Run from the repo root:

    python tests/golden/make_golden.py

The reference (aokomoriuta/ConjugateGradient) ships no recorded outputs and cannot be built or
run in this image, so the fixtures are produced from the systems it HARD-CODES:
  KA-1  tridiagonal [1 2 1], b_i = i*i/2   (SimpleConjugateGradient.cu:139-197, SimpleConjugateGradientCpu.cpp:40-105)
  KA-2  dense N=21, band 6, |sin(i+j)|      (R/CG.R:1-24; the script prints solve(A,b))
  KA-3  MgcgMain banded system at COUNT=2000 (Mgcg/cuBlas/Mgcg/MgcgMain.cs:51-104)
Each fixture stores (a) the direct solution from numpy.linalg.solve / scipy spsolve -- independent
of any CG code -- and (b) the iteration index, residual and residual trace of the CPU oracle
(oracle/cg_oracle.c), which tests/test_oracle.py re-derives and compares, and which the GPU tests
use as expected outputs.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from conjugategradient_amd import problems  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def direct(system):
    import scipy.sparse.linalg as sla

    if system.Count <= 4000:
        return np.linalg.solve(system.to_scipy().toarray(), system.b)
    return sla.spsolve(system.to_scipy().tocsc(), system.b)


def save(name, system, rule, **kw):
    r = O.cg(system, rule=rule, trace=True, **kw)
    xd = direct(system)
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        x_direct=xd, x_cg=r["x"], iteration=r["iteration"], residual=r["residual"],
                        trace=r["trace"], rule=rule, b=system.b, x0=system.x,
                        rowOffsets=system.RowOffsets, nnz=system.nnz)
    print(name, "iteration", r["iteration"], "residual", r["residual"], "max|x_cg-x_direct|", np.abs(r["x"] - xd).max())


if __name__ == "__main__":
    save("ka1_tridiagonal10", problems.tridiagonal(10), O.RULE_SIMPLE, allowable_residual=1e-8, min_iteration=0, max_iteration=10, hard_cap=100)
    save("ka2_rcg21", problems.mgcg_main(21, 6, 10.0), O.RULE_NATIVE, allowable_residual=1e-8, min_iteration=0, max_iteration=21, hard_cap=100)
    save("ka3_mgcgmain2000", problems.mgcg_main(2000, 160), O.RULE_CSHARP, allowable_residual=1e-8, min_iteration=0, max_iteration=2000)
    save("poisson5_32x32", problems.poisson(32, 32, 1), O.RULE_NATIVE, allowable_residual=1e-8, min_iteration=0, max_iteration=4096, hard_cap=5000)
    save("poisson7_12x12x12", problems.poisson(12, 12, 12), O.RULE_NATIVE, allowable_residual=1e-8, min_iteration=0, max_iteration=4096, hard_cap=5000)
    # multigrid: per-level operators and one V-cycle on 16^3 (defined by this build; see oracle/mg_oracle.c)
    s = problems.poisson(16, 16, 16)
    M = O.Multigrid(s, levels=3, nu=1, nu_coarse=4, sigma=0.5)
    rng = np.random.default_rng(2024)
    r = rng.standard_normal(s.Count)
    z = M.apply(r)
    e1, c1, r1 = M.level_csr(1)
    e2, c2, r2 = M.level_csr(2)
    res = M.pcg(rule=O.RULE_CSHARP, allowable_residual=1e-8, max_iteration=500, trace=True)
    np.savez_compressed(os.path.join(OUT, "mg_poisson7_16.npz"), r=r, z=z, e1=e1, c1=c1, r1=r1, e2=e2, c2=c2, r2=r2,
                        dinv0=M.level_dinv(0), dinv2=M.level_dinv(2), pcg_iteration=res["iteration"], pcg_residual=res["residual"],
                        pcg_trace=res["trace"], pcg_x=res["x"], x_direct=direct(s))
    print("mg_poisson7_16 pcg iteration", res["iteration"], res["residual"], np.abs(res["x"] - direct(s)).max())
    # the same with the cell-centred linear transfer (MgSetInterpolation(mg, 1)); coarse operators are those above
    ML = O.Multigrid(s, levels=3, nu=1, nu_coarse=4, sigma=0.5, interpolation=1)
    zl = ML.apply(r)
    resl = ML.pcg(rule=O.RULE_CSHARP, allowable_residual=1e-8, max_iteration=500, trace=True)
    np.savez_compressed(os.path.join(OUT, "mg_poisson7_16_linear.npz"), r=r, z=zl, pcg_iteration=resl["iteration"], pcg_residual=resl["residual"],
                        pcg_trace=resl["trace"], pcg_x=resl["x"], x_direct=direct(s))
    print("mg_poisson7_16_linear pcg iteration", resl["iteration"], resl["residual"], np.abs(resl["x"] - direct(s)).max())
