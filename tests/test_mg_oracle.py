"""The multigrid statement in oracle/mg_oracle.c (defined by this build -- the reference never implemented
its "Mgcg") against explicit scipy.sparse operators and multigrid theory.  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp

from conjugategradient_amd import problems
from tests.conftest import golden


def prolongation(nx, ny, nz):
    cx, cy, cz = (2 if nx > 1 else 1), (2 if ny > 1 else 1), (2 if nz > 1 else 1)
    NX, NY = nx // cx, ny // cy
    i = np.arange(nx * ny * nz)
    x, y, z = i % nx, (i // nx) % ny, i // (nx * ny)
    parent = ((z // cz) * NY + (y // cy)) * NX + (x // cx)
    return sp.csr_matrix((np.ones(i.size), (i, parent)), shape=(i.size, int(parent.max()) + 1))


@pytest.mark.parametrize("dims", [(8, 8, 8), (12, 8, 4), (16, 12, 1)])
def test_transfer_and_galerkin_match_scipy(oracle, dims):
    s = problems.poisson(*dims)
    A = s.to_scipy()
    P = prolongation(*dims)
    M = oracle.Multigrid(s, levels=2, sigma=0.5)
    e, c, r = M.level_csr(1)
    Ac = sp.csr_matrix((e, c, r), shape=(P.shape[1], P.shape[1]))
    ref = (0.5 * (P.T @ A @ P)).tocsr()
    ref.sort_indices()
    assert abs(Ac - ref).max() == 0
    # for the 7-point Laplacian (6,-1) this IS the rediscretised operator 2*(6,-1)
    if dims[2] > 1:
        assert set(np.unique(e)) == {-2.0, 12.0}
    rng = np.random.default_rng(0)
    v = rng.standard_normal(s.Count)
    bc = np.empty(P.shape[1])
    oracle.lib().oracle_mg_restrict(*dims, v, bc)
    np.testing.assert_allclose(bc, P.T @ v, rtol=1e-13)
    ec = rng.standard_normal(P.shape[1])
    xx = v.copy()
    oracle.lib().oracle_mg_prolong_add(*dims, xx, ec)
    np.testing.assert_allclose(xx, v + P @ ec, rtol=1e-13)
    np.testing.assert_array_equal(M.level_dinv(0), 1.0 / A.diagonal())


def test_vcycle_equals_the_matrix_formula(oracle):
    """Two-level V(1,1): z = S2(S1 r + P Ac^~ P^T (r - A S1 r)) written with scipy matrices."""
    dims = (8, 8, 8)
    s = problems.poisson(*dims)
    A, P = s.to_scipy(), prolongation(*dims)
    omega, nuc = 6.0 / 7.0, 3
    M = oracle.Multigrid(s, levels=2, nu=1, nu_coarse=nuc, omega=omega, sigma=0.5)
    Ac = 0.5 * (P.T @ A @ P)
    Dinv = sp.diags(1.0 / A.diagonal())
    Dcinv = sp.diags(1.0 / Ac.diagonal())
    rng = np.random.default_rng(1)
    r = rng.standard_normal(s.Count)
    x = omega * (Dinv @ r)
    bc = P.T @ (r - A @ x)
    ec = omega * (Dcinv @ bc)
    for _ in range(nuc - 1):
        ec = ec + omega * (Dcinv @ (bc - Ac @ ec))
    x = x + P @ ec
    x = x + omega * (Dinv @ (r - A @ x))
    np.testing.assert_allclose(M.apply(r), x, rtol=1e-12, atol=1e-14)


def test_preconditioner_is_spd_and_contracts(oracle):
    s = problems.poisson(8, 8, 8)
    A = s.to_scipy().toarray()
    M = oracle.Multigrid(s, levels=3)
    Minv = np.column_stack([M.apply(np.eye(s.Count)[:, j]) for j in range(s.Count)])
    assert np.abs(Minv - Minv.T).max() < 1e-13
    assert np.linalg.eigvalsh(0.5 * (Minv + Minv.T)).min() > 0
    ev = np.linalg.eigvals(Minv @ A).real
    assert ev.min() > 0.3 and ev.max() < 2.5             # clustered spectrum: cond(M^-1 A) ~ 4 << cond(A)
    assert ev.max() / ev.min() < 0.5 * np.linalg.cond(A)


def test_pcg_iteration_counts_and_fixture(oracle):
    g = golden("mg_poisson7_16")
    s = problems.poisson(16, 16, 16)
    M = oracle.Multigrid(s)
    res = M.pcg(rule=oracle.RULE_CSHARP, max_iteration=500, trace=True)
    assert res["iteration"] == int(g["pcg_iteration"]) == 17
    assert np.array_equal(res["x"], g["pcg_x"]) and np.array_equal(M.apply(g["r"]), g["z"])
    assert np.abs(res["x"] - g["x_direct"]).max() < 1e-9
    # the point of the preconditioner: far fewer iterations than plain CG, growing slowly with the grid
    plain = {16: 43, 32: 91}
    for n, it_plain in plain.items():
        sn = problems.poisson(n, n, n)
        it = oracle.Multigrid(sn).pcg(rule=oracle.RULE_CSHARP, max_iteration=500)["iteration"]
        assert it < it_plain / 2.4


def test_levels_clip_to_the_grid(oracle):
    assert oracle.Multigrid(problems.poisson(12, 12, 12), levels=6).levels == 3     # 12 -> 6 -> 3 (odd: stop)
    assert oracle.Multigrid(problems.poisson(16, 16, 1), levels=3).level_dims(2) == (4, 4, 1)
