#!/usr/bin/env python3
"""Diagnostic: per-Newton-iteration convergence scalars and linear iteration counts on the GPU for
both ILU0 orderings (run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd")); sys.path.insert(0, ROOT)
import numpy as np
from opmgpu import capi, decks
from opmgpu.model import GpuBlackoilModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dt_days = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
grid = decks.cartesian_grid(n, n, n, lognormal_sigma=0.5, seed=12345)
tab = decks.satfunc_standard_tables()
st = decks.initial_state(grid, tab, perturb=0.002, seed=12345)
for name, o in (("multicolor", capi.ORDER_MULTICOLOR), ("natural", capi.ORDER_NATURAL)):
    m = GpuBlackoilModel(grid, tab, capi.default_params(ilu_ordering=o))
    dt = dt_days * decks.DAY
    m.prepareStep(dt, st)
    it = 0
    for step in range(12):
        t = time.perf_counter()
        conv, lin = m.nonlinearIteration(it)
        a, s, u = m.timings()
        print("%-10s step %2d it %d conv=%d lin=%3d red=%.2e  asm %.2f ms solve %.2f ms upd %.2f ms wall %.2f ms  CNV=%s MB=%s" % (
            name, step, it, conv, lin, getattr(m, "linear_reduction", 0), a, s, u, 1e3 * (time.perf_counter() - t),
            np.array2string(m.CNV, precision=2), np.array2string(m.MB, precision=2)), flush=True)
        it += 1
        if conv or it > 10:
            m.prepareStep(dt); it = 0
    m.close()
