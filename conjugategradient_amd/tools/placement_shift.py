#!/usr/bin/env python3
"""placement_probe.py showed 2.24-2.49 ms for the same CsrMV with different (x, y) allocations.  Here x and y live inside
two oversized buffers and are SHIFTED by a byte offset, so that only the virtual (and physical) position of one stream
moves: which shifts matter, and at what granularity?"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib  # noqa: E402
from conjugategradient_amd.solver import VectorDouble, VectorInt  # noqa: E402


def main():
    n = 512
    N = n**3
    L = _lib.lib()
    _lib.require_gpu()
    L.SetDevice(0)
    blas, sparse, descr = L.CreateBlas(), L.CreateSparse(), L.CreateMatDescr()
    nnz = L.MgcgPoissonNnz(n, n, n, 0, n)
    algo = 12 * nnz + 4 * (N + 1) + 16 * N
    ev0, ev1 = L.MgcgEventCreate(), L.MgcgEventCreate()
    e, c, r = VectorDouble(nnz), VectorInt(nnz), VectorInt(N + 1)
    assert L.MgcgGeneratePoisson(e.Ptr, r.Ptr, c.Ptr, n, n, n, 0, n) == 0
    slack = 1 << 28                                   # doubles: 2 GiB of room to shift in
    X, Y = VectorDouble(N + slack), VectorDouble(N + slack)
    L.MgcgFill(X.Ptr, 1.0)
    xb, yb = X.ToRawPtr(), Y.ToRawPtr()
    print(f"e {e.ToRawPtr():#x} c {c.ToRawPtr():#x} r {r.ToRawPtr():#x} X {xb:#x} Y {yb:#x}", flush=True)

    def timed(xs, ys):
        args = (sparse, descr, C.c_void_p(yb + ys), e.ToRawPtr(), r.ToRawPtr(), c.ToRawPtr(), C.c_void_p(xb + xs), nnz, N, N, 1.0, 0.0)
        for _ in range(2):
            L.CsrMV(*args)
        out = []
        for _ in range(3):
            L.MgcgEventRecord(ev0)
            for _ in range(6):
                L.CsrMV(*args)
            L.MgcgEventRecord(ev1)
            out.append(L.MgcgEventElapsedMs(ev0, ev1) / 6)
        return sorted(out)[1]

    KB, MB = 1 << 10, 1 << 20
    if len(sys.argv) > 1 and sys.argv[1] == "sizes":
        # x buffers of different allocation sizes, allocated one after the other, x at the start of each (and 1 GiB in when it fits)
        GiB = 1 << 27                                  # doubles
        print(f"  original X (3 GiB): {timed(0, 0):.3f} ms", flush=True)
        keep = []
        for extra in (0, 1 << 19, GiB // 2, GiB, GiB, 2 * GiB, 2 * GiB, 0, 0, 3 * GiB, 1 << 19):
            xv = VectorDouble(N + extra)
            keep.append(xv)
            L.MgcgFill(xv.Ptr, 1.0)
            base = xv.ToRawPtr() - xb
            line = f"  x buffer of {8 * (N + extra) / 2**30:.3f} GiB at {xv.ToRawPtr():#x}: start {timed(base, 0):.3f}"
            if extra >= GiB:
                line += f" | +1 GiB {timed(base + (1 << 30), 0):.3f}"
            print(line + " ms", flush=True)
        print(f"  original X again: {timed(0, 0):.3f} ms", flush=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "lattice":
        # eight different y buffers (different physical pages, every one 2 MiB aligned like x): the product with y at its
        # start (x and y on the same 2 MiB lattice) and with y moved off the lattice by 0.5 / 1 / 1.5 MiB
        ys = [VectorDouble(N + (1 << 19)) for _ in range(8)]
        global_y = yb
        for k, yv in enumerate(ys):
            base = yv.ToRawPtr() - global_y
            print(f"  y buffer {k} at {yv.ToRawPtr():#x}: on the lattice {timed(0, base):.3f} | +0.5 MiB {timed(0, base + MB // 2):.3f} | +1 MiB {timed(0, base + MB):.3f} | +1.5 MiB {timed(0, base + 3 * MB // 2):.3f} | +2 MiB {timed(0, base + 2 * MB):.3f} ms", flush=True)
        xs = [VectorDouble(N + (1 << 19)) for _ in range(6)]
        for k, xv in enumerate(xs):
            L.MgcgFill(xv.Ptr, 1.0)
            base = xv.ToRawPtr() - xb
            print(f"  x buffer {k} at {xv.ToRawPtr():#x}: on the lattice {timed(base, 0):.3f} | +0.5 MiB {timed(base + MB // 2, 0):.3f} | +1 MiB {timed(base + MB, 0):.3f} | +1.5 MiB {timed(base + 3 * MB // 2, 0):.3f} | +2 MiB {timed(base + 2 * MB, 0):.3f} ms", flush=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fine":
        print("fine map of the relative offset: y shifted alone, 0 .. 40 MiB in 0.5 MiB steps (two passes)", flush=True)
        for rep in range(2):
            print("  pass", rep, " ".join(f"{timed(0, k * MB // 2):.3f}" for k in range(81)), flush=True)
        print("both shifted together, 1024 .. 2040 MiB in 64 MiB steps", flush=True)
        print("  ", " ".join(f"{1024 + 64 * k}:{timed((1024 + 64 * k) * MB, (1024 + 64 * k) * MB):.3f}" for k in range(16)), flush=True)
        print("x alone 1024 .. 2040", flush=True)
        print("  ", " ".join(f"{1024 + 64 * k}:{timed((1024 + 64 * k) * MB, 0):.3f}" for k in range(16)), flush=True)
        print("y alone 1024 .. 2040", flush=True)
        print("  ", " ".join(f"{1024 + 64 * k}:{timed(0, (1024 + 64 * k) * MB):.3f}" for k in range(16)), flush=True)
        return
    shifts = [0, 4 * KB, 64 * KB, 256 * KB, 1 * MB, 2 * MB, 4 * MB, 8 * MB, 16 * MB, 32 * MB, 64 * MB, 128 * MB, 256 * MB, 512 * MB, 768 * MB, 1024 * MB, 1536 * MB, 2040 * MB]
    print("shift of x alone (y at 0):", flush=True)
    for s in shifts:
        print(f"  x + {s / MB:9.3f} MiB  {timed(s, 0):.3f} ms", flush=True)
    print("shift of y alone (x at 0):", flush=True)
    for s in shifts:
        print(f"  y + {s / MB:9.3f} MiB  {timed(0, s):.3f} ms", flush=True)
    print("both shifted together:", flush=True)
    for s in shifts:
        print(f"  x, y + {s / MB:9.3f} MiB  {timed(s, s):.3f} ms", flush=True)
    _lib.check("placement_shift")


if __name__ == "__main__":
    main()
