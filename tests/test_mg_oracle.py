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
    assert oracle.Multigrid(problems.poisson(4, 2, 2), levels=5).levels == 3        # (4,2,2) -> (2,1,1) -> (1,1,1): a single cell ends it


# ---------------------------------------------------------------- cell-centred linear transfer (interpolation = 1)
def linear_1d(n):
    """child i: 3/4 of parent i//2, 1/4 of the parent's neighbour on the child's side (nothing outside the grid)"""
    if n == 1:
        return sp.identity(1, format="csr")
    m = n // 2
    rows, cols, vals = [], [], []
    for i in range(n):
        I = i // 2
        J = I - 1 if i % 2 == 0 else I + 1
        rows.append(i); cols.append(I); vals.append(0.75)
        if 0 <= J < m:
            rows.append(i); cols.append(J); vals.append(0.25)
    return sp.csr_matrix((vals, (rows, cols)), shape=(n, m))


def linear_prolongation(nx, ny, nz):
    return sp.kron(sp.kron(linear_1d(nz), linear_1d(ny)), linear_1d(nx)).tocsr()


@pytest.mark.parametrize("dims", [(8, 8, 8), (12, 8, 4), (16, 12, 1), (2, 2, 2), (4, 1, 6), (1, 8, 12)])
def test_linear_transfer_matches_scipy(oracle, dims):
    P = linear_prolongation(*dims)
    rng = np.random.default_rng(3)
    v = rng.standard_normal(P.shape[0])
    bc = np.empty(P.shape[1])
    oracle.lib().oracle_mg_restrict_linear(*dims, v, bc)
    np.testing.assert_allclose(bc, P.T @ v, rtol=1e-13, atol=1e-14)
    ec = rng.standard_normal(P.shape[1])
    xx = v.copy()
    oracle.lib().oracle_mg_prolong_add_linear(*dims, xx, ec)
    np.testing.assert_allclose(xx, v + P @ ec, rtol=1e-13, atol=1e-14)
    # interior rows of P sum to one: constants are interpolated exactly away from the boundary
    if min(d for d in dims if d > 1) >= 8:
        assert np.asarray(P.sum(axis=1)).ravel().max() == 1.0


def test_linear_vcycle_equals_the_matrix_formula_and_is_spd(oracle):
    dims = (8, 8, 8)
    s = problems.poisson(*dims)
    A, P0, P = s.to_scipy(), prolongation(*dims), linear_prolongation(*dims)
    omega, nuc = 6.0 / 7.0, 3
    M = oracle.Multigrid(s, levels=2, nu=1, nu_coarse=nuc, omega=omega, sigma=0.5, interpolation=1)
    Ac = 0.5 * (P0.T @ A @ P0)                            # the coarse operator does not change with the transfer
    Dinv, Dcinv = sp.diags(1.0 / A.diagonal()), sp.diags(1.0 / Ac.diagonal())
    r = np.random.default_rng(1).standard_normal(s.Count)
    x = omega * (Dinv @ r)
    bc = P.T @ (r - A @ x)
    ec = omega * (Dcinv @ bc)
    for _ in range(nuc - 1):
        ec = ec + omega * (Dcinv @ (bc - Ac @ ec))
    x = x + P @ ec
    x = x + omega * (Dinv @ (r - A @ x))
    np.testing.assert_allclose(M.apply(r), x, rtol=1e-12, atol=1e-14)
    M3 = oracle.Multigrid(s, levels=3, interpolation=1)
    Minv = np.column_stack([M3.apply(np.eye(s.Count)[:, j]) for j in range(s.Count)])
    assert np.abs(Minv - Minv.T).max() < 1e-13
    assert np.linalg.eigvalsh(0.5 * (Minv + Minv.T)).min() > 0


def test_linear_transfer_needs_fewer_iterations(oracle):
    counts = {}
    for n in (16, 32):
        sn = problems.poisson(n, n, n)
        const = oracle.Multigrid(sn, levels=4).pcg(rule=oracle.RULE_CSHARP, max_iteration=500)
        lin = oracle.Multigrid(sn, levels=4, interpolation=1).pcg(rule=oracle.RULE_CSHARP, max_iteration=500)
        assert np.abs(lin["x"] - const["x"]).max() < 1e-7
        counts[n] = (const["iteration"], lin["iteration"])
    assert counts[32][1] < counts[32][0] and counts[32][1] <= counts[16][1] + 2     # nearly grid independent


def test_linear_transfer_fixture(oracle):
    g = golden("mg_poisson7_16_linear")
    s = problems.poisson(16, 16, 16)
    M = oracle.Multigrid(s, interpolation=1)
    res = M.pcg(rule=oracle.RULE_CSHARP, max_iteration=500, trace=True)
    assert res["iteration"] == int(g["pcg_iteration"]) == 12
    assert np.array_equal(res["x"], g["pcg_x"]) and np.array_equal(M.apply(g["r"]), g["z"])
    assert np.abs(res["x"] - g["x_direct"]).max() < 1e-9


def test_pcg_with_partitioned_sums(oracle):
    """oracle_pcg_parts: the PCG with its dot products cut at the rows of a row partition and the per-device sums added in device order from
    0 (resultsDot.Sum(), ConjugateGradientParallelGpu.cs:463,499,525) -- what the row-partitioned HIP loop must EQUAL under MGCG_DOT_ORDER=1
    (tests/test_gpu_dot_order.py).  One part = the plain loop, bit for bit; several parts: the same iteration count, values within the sums'
    rounding; the sum over parts really is rank-ordered serial sums of the slices."""
    s = problems.poisson(16, 16, 16)
    s.b[:] = np.random.default_rng(3).standard_normal(s.Count)
    M = oracle.Multigrid(s, levels=3)
    one = M.pcg(rule=oracle.RULE_CSHARP, max_iteration=400, trace=True)
    same = M.pcg(rule=oracle.RULE_CSHARP, max_iteration=400, trace=True, offsets=[0, s.Count])
    assert one["iteration"] == same["iteration"] and np.array_equal(one["trace"], same["trace"]) and np.array_equal(one["x"], same["x"])
    for parts in (2, 3, 8):
        off = oracle.partition(s.Count, parts)
        p = M.pcg(rule=oracle.RULE_CSHARP, max_iteration=400, trace=True, offsets=off)
        assert p["iteration"] == one["iteration"]
        np.testing.assert_allclose(p["trace"], one["trace"], rtol=1e-9)
        assert not np.array_equal(p["trace"], one["trace"])          # the association of the sums differs: other bits
    # the first residual of the partitioned loop, recomputed by hand: r0 = b - A x0 (x0 = 0: r0 = b), z0 = M^-1 r0, p0 = z0, rz = sum over parts
    off = oracle.partition(s.Count, 3)
    z0 = M.apply(s.b)
    rz = 0.0
    for d in range(3):
        rz += oracle.dot(s.b[off[d]: off[d + 1]], z0[off[d]: off[d + 1]])
    Ap = oracle.spmv(s.Elements, s.ColumnIndeces, s.RowOffsets, z0)
    pAp = 0.0
    for d in range(3):
        pAp += oracle.dot(z0[off[d]: off[d + 1]], Ap[off[d]: off[d + 1]])
    alpha = rz / pAp
    r1 = oracle.set_added(s.b, Ap, -alpha)
    rr = 0.0
    for d in range(3):
        rr += oracle.dot(r1[off[d]: off[d + 1]], r1[off[d]: off[d + 1]])
    p3 = M.pcg(rule=oracle.RULE_CSHARP, max_iteration=400, trace=True, offsets=off)
    assert p3["trace"][0] == np.sqrt(rr)
