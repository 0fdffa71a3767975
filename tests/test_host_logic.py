"""Host-side logic that needs no device: partition arithmetic, halo plan, the class surface.  CPU only."""
import numpy as np
import pytest

from conjugategradient_amd import problems
from conjugategradient_amd.parallel import RankPartition, halo_plan
from conjugategradient_amd.solver import ApplicationException, ConjugateGradient, LinerEquations, SparseMatrix


def test_partition_matches_reference_formula():
    # ConjugateGradientParallelGpu.cs:271-277: floor(N/ndev) each, remainder to the last device
    assert problems.partition_offsets(207402, 4) == [0, 51850, 103700, 155550, 207402]
    assert problems.partition_offsets(7, 3) == [0, 2, 4, 7]
    assert problems.partition_offsets(5, 8) == [0, 0, 0, 0, 0, 0, 0, 0, 5]
    for n, w in [(100, 1), (101, 7), (134217728, 8)]:
        off = problems.partition_offsets(n, w)
        assert off[0] == 0 and off[-1] == n and all(b >= a for a, b in zip(off, off[1:]))


def test_partition_of_equal_nonzero_counts():
    """balance="nnz" (not in the reference): contiguous row ranges, every rank within one row's length of nnz / ranks."""
    s = problems.random_spd(3000, mean_upper=6.0, seed=5)          # rows get longer towards the end of this generator's matrices
    ro = s.RowOffsets
    longest = int(np.diff(ro).max())
    for w in (1, 2, 3, 8):
        off = problems.partition_offsets(s.Count, w, ro, "nnz")
        assert off[0] == 0 and off[-1] == s.Count and len(off) == w + 1 and all(b >= a for a, b in zip(off, off[1:]))
        per = [int(ro[off[r + 1]] - ro[off[r]]) for r in range(w)]
        assert sum(per) == s.nnz and max(abs(v - s.nnz / w) for v in per) <= longest
        parts = [RankPartition.of(s.Count, w, r, ro, "nnz") for r in range(w)]
        assert [p.offset for p in parts] == off[:-1] and [p.elementCount for p in parts] == per
    rows = problems.partition_offsets(s.Count, 8)
    per_rows = [int(ro[rows[r + 1]] - ro[rows[r]]) for r in range(8)]
    assert max(per_rows) > 1.5 * s.nnz / 8                          # what the option is for
    assert problems.partition_offsets(5, 8, np.arange(6) * 3, "nnz")[-1] == 5      # fewer rows than ranks: empty ranks, still monotone
    with pytest.raises(ValueError):
        problems.partition_offsets(10, 2, None, "nnz")
    with pytest.raises(ValueError):
        problems.partition_offsets(10, 2, None, "columns")


def test_rank_partition_halo_widths():
    s = problems.mgcg_main(1000, 160)
    ro = s.RowOffsets
    parts = [RankPartition.of(s.Count, 4, r, ro) for r in range(4)]
    assert sum(p.count for p in parts) == s.Count and sum(p.elementCount for p in parts) == s.nnz
    for p in parts:
        cols = s.ColumnIndeces[p.elementOffset: p.elementOffset + p.elementCount]
        p.minJ, p.maxJ = int(cols.min()), int(cols.max())
    # :397-398  lastCount = offset - minJ, nextCount = maxJ - (offset+count) + 1; edge devices have one side only
    assert parts[0].lastCount == 0 and parts[0].nextCount == 79
    assert parts[1].lastCount == 79 and parts[1].nextCount == 79
    assert parts[3].lastCount == 79 and parts[3].nextCount == 0


def test_halo_plan_banded_and_unstructured():
    # banded: only adjacent ranks talk, widths = band half-width
    meta = [(0, 250, 0, 328), (250, 250, 171, 578), (500, 250, 421, 828), (750, 250, 671, 999)]
    sends, recvs = halo_plan(meta, 1)
    assert recvs == [(0, 171, 79), (2, 500, 79)]
    assert sends == [(0, 250, 79), (2, 421, 79)]
    # every send of rank a to rank b is a recv of rank b from rank a with the same range
    for a in range(4):
        for (b, beg, ln) in halo_plan(meta, a)[0]:
            assert (a, beg, ln) in halo_plan(meta, b)[1]
    # unstructured: every rank needs everything -> all-gather pattern
    meta = [(0, 30, 0, 99), (30, 30, 0, 99), (60, 40, 0, 99)]
    sends, recvs = halo_plan(meta, 0)
    assert recvs == [(1, 30, 30), (2, 60, 40)] and sends == [(1, 0, 30), (2, 0, 30)]
    # an empty rank neither sends nor receives
    meta = [(0, 10, 0, 9), (10, 0, 0, -1)]
    assert halo_plan(meta, 1) == ([], [])
    assert halo_plan(meta, 0) == ([], [])


def test_class_surface():
    le = LinerEquations(5, 3)
    assert le.Count == 5 and le.x.shape == (5,) and le.b.shape == (5,) and le.A is None
    m = SparseMatrix(4, 3)
    assert m.Elements.shape == (12,) and np.all(m.ColumnIndeces == -1) and m.RowCount == 5   # RowOffsets.Length (sic)
    m.Elements[:] = 1
    m.Clear()
    assert np.all(m.Elements == 0)

    class Dummy(ConjugateGradient):
        def Solve(self):
            pass

    cg = Dummy(5, 3, 2, 4, 1e-3)
    cg.Residual = 0.0
    cg.Iteration = 1
    assert cg.IsConverged is False           # below MinIteration
    cg.Iteration = 2
    assert cg.IsConverged is True
    cg.Residual = 1.0
    cg.Iteration = 4
    assert cg.IsConverged is False
    cg.Iteration = 5
    with pytest.raises(ApplicationException):
        cg.IsConverged


def test_problem_generators_shapes():
    s = problems.poisson(256, 256, 1)
    assert s.Count == 65536 and s.nnz == 326656                 # BASELINE config 1
    assert problems.poisson_nnz(256, 256, 256) == 117047296       # config 2
    assert problems.poisson_nnz(512, 512, 512) == 937951232       # config 3
    r = problems.random_spd(2000, seed=3)
    assert 25 < r.nnz / r.Count < 35                              # config 5: ~30 nnz/row on average, irregular
    assert np.diff(r.RowOffsets).min() >= 1 and np.diff(r.RowOffsets).max() > 40
    A = r.to_scipy()
    assert abs(A - A.T).max() == 0
    d = A.diagonal()
    off = np.asarray(abs(A).sum(axis=1)).ravel() - abs(d)
    assert np.all(d > off)                                        # strictly diagonally dominant => SPD
    np.testing.assert_allclose(A @ np.ones(2000), r.b)


@pytest.mark.parametrize("rows,period,wgs,tile_rows,expect_mode", [
    (512**3, 512 * 512, 512, 256, 1),        # BASELINE config 2/3: two half-plane slices, swept through 512 planes
    (512 * 512 * 64, 512 * 512, 512, 256, 1),    # one rank's slab of config 4
    (256**3, 256 * 256, 512, 256, 2),        # a plane is smaller than the grid: two planes per trip
    (128**3, 128 * 128, 512, 256, 2),        # eight planes per trip
    (128 * 128 * 13, 128 * 128, 512, 256, 2),    # planes not a multiple of the planes per trip: ragged last trip
    (512**3, 512 * 512, 4096, 128, 2),       # the row-pattern kernel's shape (tiles of 128 rows, 16 wavefronts per CU)
    (384**3, 384 * 384, 512, 256, 0),        # 576 tiles per plane do not split over 512 workgroups: memory order
    (512**3, 640, 512, 256, 0),              # a period the kernel cannot use
    (100000, 0, 512, 256, 0),                # unknown period, ragged tail (full tiles only)
    (256 * 256 * 2, 256 * 256, 512, 256, 0),     # fewer than 3 planes
])
def test_tile_order_visits_every_tile_once(rows, period, wgs, tile_rows, expect_mode):
    """The z sweep of the lane = row SpMV kernels is a permutation of the tiles: every full tile exactly once, whatever the
    mode (a skipped or doubled tile would be a wrong or racy product); in sweep modes every trip of the whole grid covers one
    contiguous run of tiles and XCD k = workgroup % 8 owns the k-th eighth of every such run (one run per plane in mode 2,
    one per slice of the plane in mode 1)."""
    import ctypes as C

    from conjugategradient_amd import _lib

    L = _lib.lib()
    n_tiles = rows // tile_rows
    max_trips = -(-n_tiles // wgs) + 8
    buf = np.full(wgs * max_trips, -2, dtype=np.int32)
    mode = L.MgcgDebugTileOrder(rows, period, wgs, tile_rows, buf.ctypes.data_as(C.c_void_p), max_trips)
    assert mode == expect_mode
    order = buf.reshape(wgs, max_trips)
    seen = order[order >= 0]
    assert seen.size == n_tiles and np.array_equal(np.sort(seen), np.arange(n_tiles))
    for wg in range(wgs):                                  # trips of a workgroup are a prefix: no holes
        valid = order[wg] >= 0
        assert not np.any(valid[1:] & ~valid[:-1])
    if mode != 0:
        per_plane = period // tile_rows
        for t in range(max_trips):
            col = order[:, t]
            col = col[col >= 0]
            if col.size == 0:
                continue
            assert col.max() - col.min() + 1 == col.size       # one contiguous run for the whole chip
        slices = per_plane // wgs if mode == 1 else 1
        for k in range(8):                                     # an XCD's tiles of a plane: an eighth, in `slices` contiguous runs
            mine = order[k::8]
            mine = mine[mine >= 0]
            in_plane = np.unique(mine % per_plane)
            assert in_plane.size == per_plane // 8
            assert int(np.count_nonzero(np.diff(in_plane) != 1)) + 1 == slices


def test_bench_vcycle_bytes_formula():
    """bench.py's algorithmic-byte count of one MGCG iteration (per-pass formulas) -- the denominator of the
    `mgcg.frac_of_peak` figure -- checked against a hand count on a tiny hierarchy and against the 512^3 figure DESIGN.md quotes."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    def nnz(m):
        return 7 * m**3 - 6 * m * m

    # two levels, V(1,1), 2 coarse sweeps on 8^3 -> 4^3
    n, nc = 8, 4
    N, Nc = n**3, nc**3
    fine = 24 * N + (12 * nnz(n) + 28 * N) + 2 * (8 * N + 8 * Nc) + (12 * nnz(n) + 36 * N)     # first sweep, residual, R + P, post sweep
    coarse = 24 * Nc + (12 * nnz(nc) + 36 * Nc)                                                  # first sweep + one more
    shell = 12 * nnz(n) + 4 * (N + 1) + 16 * N + 72 * N
    v, s = bench.vcycle_bytes(n, 2, 1, 2)
    assert (v, s) == (fine + coarse, shell)
    v, s = bench.vcycle_bytes(512, 3, 1, 4)
    assert v == 42127196160 and v + s == 65730641924


def test_device_workers_under_thread_sanitizer(tmp_path):
    """host/Mgcg.hpp DeviceWorkers (the long-lived per-device threads behind the C++ twin's Parallel.For): race-free hand-over
    of phases and results, exceptions surface once."""
    import os
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "workers"
    src = os.path.join(ROOT, "tests", "host_workers_check.cpp")
    inc = os.path.join(ROOT, "conjugategradient_amd", "host")
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-I" + inc, src, "-o", str(exe)], capture_output=True, text=True)
    if build.returncode != 0:                                   # no libtsan on this host: the plain build still checks the logic
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-I" + inc, src, "-o", str(exe)])
    for n, phases in ((8, 4000), (1, 100), (3, 2000)):
        out = subprocess.run([str(exe), str(n), str(phases)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and "ok 2" in out.stdout and "WARNING: ThreadSanitizer" not in out.stderr, out.stdout + out.stderr


def test_required_bytes_of_the_vcycle():
    """bench.py's second byte count: what must move as the library runs the cycle (no D^-1 array on constant-coefficient levels, the
    prolongation rewrites the fine iterate, the folded first sweep), next to the SURVEY-formula count."""
    import bench

    def nnz(m):
        return 7 * m**3 - 6 * m * m
    n, N, Nc = 8, 512, 64
    sweep = lambda m: 12 * nnz(m) + 4 * (m**3 + 1) + 24 * m**3     # noqa: E731
    shell = 12 * nnz(n) + 4 * (N + 1) + 16 * N + 72 * N
    v, s = bench.vcycle_required_bytes(n, 2, 1, 2, fold=False)
    fine = 16 * N + (12 * nnz(n) + 4 * (N + 1) + 24 * N) + (8 * N + 8 * Nc) + (16 * N + 8 * Nc) + sweep(n)
    coarse = 16 * Nc + sweep(4)
    assert (v, s) == (fine + coarse, shell)
    vf, _ = bench.vcycle_required_bytes(n, 2, 1, 2, fold=True)
    # the first sweep's store and the residual's separate read of x; on this small power-of-two level also the whole prolongation pass
    # (16 N + 8 Nc) and the last sweep's read of the stored iterate (8 N), for the parent's correction read per gather (8 Nc)
    assert v - vf == (16 * N + 8 * N) + (16 * N + 8 * Nc) + 8 * N - 8 * Nc
    v6, _ = bench.vcycle_required_bytes(6, 2, 1, 2, fold=True)        # n = 6: not a power of two, the prolongation kernel stays
    v6n, _ = bench.vcycle_required_bytes(6, 2, 1, 2, fold=False)
    assert v6n - v6 == 24 * 216
    big, bigshell = bench.vcycle_required_bytes(512, 3, 1, 4)         # 512^3: only the 256^3 level is under the row limit
    assert bench.FOLD_UP_MAX_ROWS < 512**3 and big + bigshell == 60831694880 - 24 * 256**3      # (60.83 GB: the count before this fold)
    v2, _ = bench.vcycle_required_bytes(n, 2, 2, 2, fold=True)
    assert v2 == fine + coarse + 2 * sweep(n)            # V(2,2): nothing to fold, one more sweep either side
    va, sa = bench.vcycle_bytes(512, 3, 1, 4)
    vr, sr = bench.vcycle_required_bytes(512, 3, 1, 4)
    assert sr == sa and vr < va
