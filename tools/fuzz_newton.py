"""Randomised campaign for whole Newton iterations (assemble -> block-ILU0 / BiCGStab -> updateState) GPU vs the CPU oracle running freely
from the same start: random small decks (ACTNUM, NNC, threshold pressures, ENDSCALE, VAPPARS / ROCKTAB), both orderings, ILU0 or CPR on the
device, f64 solve with a tight reduction, three iterations: phase states identical, p within 1e-6 relative, s within 1e-6.   python tools/fuzz_newton.py [ncases] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from opmgpu import capi, decks
from opmgpu.model import GpuBlackoilModel
from oracle import oracle as orc

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 700
worst = {"p": 0.0, "sat": 0.0, "its_diff": 0.0}
compared = flips = 0
for case in range(ncases):
    rng = np.random.default_rng(seed0 + case)
    nx, ny, nz = (int(v) for v in rng.integers(3, 9, 3))
    kw = dict(lognormal_sigma=float(rng.uniform(0.0, 1.2)), seed=int(seed0 + case))
    if rng.random() < 0.3: kw["nnc_fraction"] = float(rng.uniform(0.02, 0.08))
    if rng.random() < 0.3: kw["actnum"] = rng.random(nx * ny * nz) > rng.uniform(0.1, 0.4)
    if rng.random() < 0.3: kw["thpres"] = float(rng.uniform(0.01, 0.1)) * decks.BAR
    grid = decks.cartesian_grid(nx, ny, nz, **kw)
    if grid.nc < 8: continue
    tkw = {}
    if rng.random() < 0.3: tkw["vappars"] = (float(rng.uniform(0.1, 2.0)), float(rng.uniform(0.1, 2.0)))
    if rng.random() < 0.3: tkw["rocktab"] = [(100.0, 0.97, 0.94), (200.0, 1.0, 1.0), (300.0, 1.02, 1.07), (500.0, 1.05, 1.1)]
    tab = decks.satfunc_standard_tables(**tkw)
    if rng.random() < 0.3: grid = decks.with_endpoints(grid, decks.random_endpoints(grid, seed=int(seed0 + case)))
    st = decks.initial_state(grid, tab, perturb=float(rng.uniform(0.001, 0.01)), seed=int(seed0 + case))
    ordering = int(rng.integers(0, 2))
    prm = capi.default_params(ilu_ordering=ordering, linear_solver_reduction=1e-12, linear_solver_maxiter=1000, cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=int(rng.integers(0, 2)))   # the oracle side is always ILU0
    scale = np.asarray(prm.matbalscale[:])
    dt = float(rng.uniform(0.5, 10.0)) * decks.DAY
    nc = grid.nc
    rowptr, col = orc.pattern(grid)
    m = GpuBlackoilModel(grid, tab, prm)
    try:
        m.prepareStep(dt, st)
        pos, so, acc0 = None, st.copy(), None
        for it in range(3):
            m.assemble(it == 0); m.getConvergence()
            m.solveJacobianSystem(single_precision=False)
            m.updateState()
            if pos is None: pos = m.ordering()[0]
            r, val, acc0, _ = orc.assemble(grid, tab, dt, so, rowptr, col, scale=tuple(scale), accum0=acc0)
            b = np.ascontiguousarray((r * np.repeat(scale, nc)).reshape(3, nc).T).ravel()
            sto, x, ito, _, _ = orc.bicgstab(rowptr, col, val, b, prm, position=pos, single=False)
            assert sto == 0, ("oracle solve", case, it)
            so = orc.update_state(grid, tab, prm, np.ascontiguousarray(x.reshape(nc, 3).T).ravel(), so)
            g = m.getState()
            if not np.array_equal(g.hc, so.hc):
                # a phase-state switch decided by a comparison within rounding distance of its threshold: count, and stop this case
                flips += 1; break
            ep, es = float(np.abs(g.p - so.p).max() / np.abs(so.p).max()), float(np.abs(g.sat - so.sat).max())
            worst["p"], worst["sat"] = max(worst["p"], ep), max(worst["sat"], es)
            if not prm.use_cpr: worst["its_diff"] = max(worst["its_diff"], abs(m.linear_iterations - ito))
            assert ep < 1e-6 and es < 1e-6, (case, it, ep, es)
        else:
            compared += 1
    finally:
        m.close()
print("cases", ncases, "compared", compared, "threshold flips", flips, "worst", {k: "%.1e" % v for k, v in worst.items()}, flush=True)
