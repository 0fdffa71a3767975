#!/usr/bin/env python3
"""Interleaved A/B sweep of the SpMV kernel variants on the 7-point Poisson matrix (one process,
rounds interleaved -- cdna_hip_programming.md rule 24).  Prints median/min ms and algorithmic GB/s."""
import argparse
import ctypes as C
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib  # noqa: E402
from conjugategradient_amd.solver import VectorDouble, VectorInt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--variants", default="")
    ap.add_argument("--extra-grids", default="", help="comma list of grid sizes to add as 'dcsr gN' and 'rows gN' variants")
    ap.add_argument("--only-ablations", default="", help="comma list of ablation codes to keep (default all)")
    ap.add_argument("--ablate", action="store_true", help="also time diagnostic ablations (no y store / gathers from L1)")
    a = ap.parse_args()
    L = _lib.lib()
    _lib.require_gpu()
    L.SetDevice(0)
    blas, sparse, descr = L.CreateBlas(), L.CreateSparse(), L.CreateMatDescr()
    n = a.grid
    N = n**3
    nnz = L.MgcgPoissonNnz(n, n, n, 0, n)
    e, c, r = VectorDouble(nnz), VectorInt(nnz), VectorInt(N + 1)
    assert L.MgcgGeneratePoisson(e.Ptr, r.Ptr, c.Ptr, n, n, n, 0, n) == 0
    x, y = VectorDouble(N), VectorDouble(N)
    L.MgcgFill(x.Ptr, 1.0)
    _lib.check("setup")
    algo = 12 * nnz + 4 * (N + 1) + 16 * N
    # (label, kernel, rows, flags, grid)
    # (label, kernel, rows, flags, grid, tileRows, tilePlanes)
    variants = [("wg256 R128", 1, 128, 0, 0, 0, 0), ("wg256 R256", 1, 256, 0, 0, 0, 0),
                ("wave R64", 1, 64, 0, 0, 0, 0), ("wave R64 nt", 1, 64, 1, 0, 0, 0), ("wave R32", 1, 32, 0, 0, 0, 0), ("wave R32 nt", 1, 32, 1, 0, 0, 0),
                ("wave R32 g4096", 1, 32, 0, 4096, 0, 0), ("wave R64 g4096", 1, 64, 0, 4096, 0, 0),
                ("wave R64 g2048", 1, 64, 0, 2048, 0, 0), ("wave R64 g3072", 1, 64, 0, 3072, 0, 0), ("wave R64 g5120", 1, 64, 0, 5120, 0, 0),
                ("wave R64 g4096 nt", 1, 64, 1, 4096, 0, 0), ("wave R32 g6144", 1, 32, 0, 6144, 0, 0),
                ("wave R64 band", 1, 64, 4, 0, 0, 0), ("wave R32 band", 1, 32, 4, 0, 0, 0), ("wave R32 band 8Lx8", 1, 32, 4, 0, 8 * n, 8),
                ("wg256 band 64Lx1", 1, 128, 4, 0, 64 * n, 1), ("vector 4 lanes", 4, 128, 0, 0, 0, 0),
                ("rowtile", 10, 64, 0, 0, 0, 0), ("DOT rowtile", 10, 64, 0, 0, 0, 0), ("rowtile no sweep", 10, 64, 0, 0, 0, 0), ("rowtile g4096", 10, 64, 0, 4096, 0, 0),
                ("rows", 9, 64, 0, 0, 0, 0), ("rows g2048", 9, 64, 0, 2048, 0, 0), ("rows g5120", 9, 64, 0, 5120, 0, 0), ("DOT rows", 9, 64, 0, 0, 0, 0),
                ("dcsr", 1, 64, 0, 0, 0, 0), ("dcsr g2048", 1, 64, 0, 2048, 0, 0), ("dcsr g8192", 1, 64, 0, 8192, 0, 0), ("dcsr g5120", 1, 64, 0, 5120, 0, 0), ("dcsr g6144", 1, 64, 0, 6144, 0, 0), ("DOT dcsr", 1, 64, 0, 0, 0, 0), ("pattern", 1, 64, 0, 0, 0, 0), ("DOT pattern", 1, 64, 0, 0, 0, 0),
                ("DOT wave R64 g4096", 1, 64, 0, 4096, 0, 0), ("DOT wave R64 band", 1, 64, 4, 0, 0, 0), ("DOT wg256 R128", 1, 128, 0, 0, 0, 0)]
    if a.variants:
        keep = set(a.variants.split(","))
        variants = [v for v in variants if v[0] in keep]
    for g in [int(t) for t in a.extra_grids.split(",") if t]:
        variants.append((f"dcsr g{g}", 1, 64, 0, g, 0, 0))
        variants.append((f"pattern g{g}", 1, 64, 0, g, 0, 0))
        variants.append((f"rows g{g}", 9, 64, 0, g, 0, 0))
    variants = list({v[0]: v for v in variants}.values())
    ev0, ev1 = L.MgcgEventCreate(), L.MgcgEventCreate()
    times = {v[0]: [] for v in variants}

    def run(v):
        os.environ["MGCG_SPMV_ABLATE"] = str(v[7]) if len(v) > 7 else "0"     # honoured by lab builds of the library only (make -C csrc lab)
        L.MgcgSetMatrixCompression(sparse, 2 if "dcsr" in v[0] else (1 if "pattern" in v[0] else 0))
        L.MgcgSetSpmvKernel(sparse, v[1])
        L.MgcgSetSpmvTuning(sparse, v[2], v[3], v[4])
        L.MgcgSetSpmvPeriod(sparse, n * n if (v[3] & 4) else (640 if v[0] == "rowtile no sweep" else 0))   # 640: a hint the row-tile kernel cannot use -> grid-stride order
        L.MgcgSetSpmvTile(sparse, v[5], v[6])
        L.MgcgEventRecord(ev0)
        for _ in range(a.reps):
            if v[0].startswith("DOT"):      # the fused variant of the CG loop: y = A x and sum x_i y_i (blocking scalar read included)
                L.CsrMVDot(blas, sparse, y.ToRawPtr(), e.ToRawPtr(), r.ToRawPtr(), c.ToRawPtr(), x.ToRawPtr(), x.ToRawPtr(), nnz, N, N)
            else:
                L.CsrMV(sparse, descr, y.ToRawPtr(), e.ToRawPtr(), r.ToRawPtr(), c.ToRawPtr(), x.ToRawPtr(), nnz, N, N, 1.0, 0.0)
        L.MgcgEventRecord(ev1)
        return L.MgcgEventElapsedMs(ev0, ev1) / a.reps

    if a.ablate:
        base = [v for v in variants if v[0] in ("wg256 R128", "wave R64 g4096", "wave R64 band", "wg256 band 64Lx1", "wave R64 g4096 nt", "dcsr", "rows", "pattern", "DOT pattern") or v[0].startswith(("rows g", "dcsr g", "pattern g"))]
        for ab, tag in ((1, "no y store"), (2, "gathers from L1"), (3, "no store + L1 gathers"), (8, "y store confined to 512 KB"), (128, "64-bit gather addresses")):
            if a.only_ablations and str(ab) not in a.only_ablations.split(","):
                continue
            for v in base:
                variants.append((v[0] + " | " + tag,) + tuple(v[1:]) + (ab,))
        times = {v[0]: [] for v in variants}
    for v in variants:      # warm-up
        run(v)
    for _ in range(a.rounds):
        for v in variants:
            times[v[0]].append(run(v))
    _lib.check("sweep")
    # attainable copy bandwidth for reference: Copy = hipMemcpyAsync D2D of N doubles (16N bytes moved)
    L.MgcgEventRecord(ev0)
    for _ in range(20):
        L.Copy(blas, y.ToRawPtr(), x.ToRawPtr(), N, 0, 0)
    L.MgcgEventRecord(ev1)
    copy_ms = L.MgcgEventElapsedMs(ev0, ev1) / 20
    L.MgcgEventRecord(ev0)
    for _ in range(20):
        L.Axpy(blas, y.ToRawPtr(), x.ToRawPtr(), N, 0.5)
    L.MgcgEventRecord(ev1)
    axpy_ms = L.MgcgEventElapsedMs(ev0, ev1) / 20
    print(f"grid {n}^3 rows {N} nnz {nnz} algorithmic bytes {algo}")
    print(f"  memcpy D2D {16 * N / copy_ms / 1e6:8.1f} GB/s   axpy {24 * N / axpy_ms / 1e6:8.1f} GB/s")
    rows = []
    for v in variants:
        t = times[v[0]]
        med, mn = statistics.median(t), min(t)
        rows.append(dict(variant=v[0], median_ms=med, min_ms=mn, gbps=algo / med / 1e6, frac=algo / med / 1e6 / 8000.0))
        print(f"  {v[0]:22s} median {med:8.3f} ms  min {mn:8.3f} ms  {algo / med / 1e6:8.1f} GB/s  {algo / med / 1e6 / 80.0:5.1f}% of 8 TB/s")
    print(json.dumps(dict(grid=n, copy_gbps=16 * N / copy_ms / 1e6, axpy_gbps=24 * N / axpy_ms / 1e6, variants=rows)))


if __name__ == "__main__":
    main()
