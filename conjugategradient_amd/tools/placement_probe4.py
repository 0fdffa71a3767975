#!/usr/bin/env python3
"""Is the placement effect of the SpMV's written vector a periodic function of its address?  7-point Poisson 512^3, CsrMV on raw pointers:
y = one allocation with 96 MiB of slack, shifted in steps of 2 MiB (coarse scan) and, around the best and the worst coarse shift, in steps
of 256 KiB; the same for a second y allocation (does the pattern repeat?) and for x.  Prints one JSON object."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conjugategradient_amd import _lib  # noqa: E402
from conjugategradient_amd.solver import VectorDouble, VectorInt  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    N = n**3
    L = _lib.lib()
    _lib.require_gpu()
    L.SetDevice(0)
    blas, sparse, descr = L.CreateBlas(), L.CreateSparse(), L.CreateMatDescr()
    nnz = L.MgcgPoissonNnz(n, n, n, 0, n)
    ev0, ev1 = L.MgcgEventCreate(), L.MgcgEventCreate()
    e, c, r = VectorDouble(nnz), VectorInt(nnz), VectorInt(N + 1)
    assert L.MgcgGeneratePoisson(e.Ptr, r.Ptr, c.Ptr, n, n, n, 0, n) == 0
    slack = 96 << 20
    x = VectorDouble(N + slack // 8)
    ys = [VectorDouble(N + slack // 8) for _ in range(2)]
    L.MgcgFill(x.Ptr, 1.0)

    def timed(xp, yp, reps=8):
        args = (sparse, descr, yp, e.ToRawPtr(), r.ToRawPtr(), c.ToRawPtr(), xp, nnz, N, N, 1.0, 0.0)
        for _ in range(2):
            L.CsrMV(*args)
        L.MgcgEventRecord(ev0)
        for _ in range(reps):
            L.CsrMV(*args)
        L.MgcgEventRecord(ev1)
        return round(L.MgcgEventElapsedMs(ev0, ev1) / reps, 4)

    out = {"grid": n, "addresses": {"elements": hex(e.ToRawPtr()), "columns": hex(c.ToRawPtr()), "x": hex(x.ToRawPtr()), "y": [hex(y.ToRawPtr()) for y in ys]}}
    coarse = [k * (2 << 20) for k in range(48)]
    for j, y in enumerate(ys):
        out[f"y{j}_shift_2MiB_steps_ms"] = [timed(x.ToRawPtr(), y.ToRawPtr() + s) for s in coarse]
    t = out["y0_shift_2MiB_steps_ms"]
    kb, kw = t.index(min(t)), t.index(max(t))
    fine = [k * (256 << 10) for k in range(-8, 9)]
    out["y0_fine_around_best"] = {"coarse_shift_MiB": 2 * kb, "ms": [timed(x.ToRawPtr(), ys[0].ToRawPtr() + max(0, coarse[kb] + s)) for s in fine]}
    out["y0_fine_around_worst"] = {"coarse_shift_MiB": 2 * kw, "ms": [timed(x.ToRawPtr(), ys[0].ToRawPtr() + max(0, coarse[kw] + s)) for s in fine]}
    out["x_shift_2MiB_steps_ms_with_best_y"] = [timed(x.ToRawPtr() + s, ys[0].ToRawPtr() + coarse[kb]) for s in coarse]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
