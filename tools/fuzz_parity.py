"""Randomised parity campaign (GPU through the C ABI vs the CPU oracle): random small decks -- dimensions, heterogeneity, inactive
cells, non-neighbour connections, threshold pressures, ENDSCALE end points, VAPPARS / ROCKTAB, pc scaling, time step, ordering, Jacobian
precision -- each with random states covering all three phase states.  Checks residual, Jacobian, convergence scalars and updateState.
Prints the worst errors seen; exits non-zero on a violation.    python tools/fuzz_parity.py [ncases] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from opmgpu import capi, decks
from opmgpu.model import GpuBlackoilModel
from oracle import oracle as orc

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


worst = {"jac64": 0.0, "res64": 0.0, "jac32": 0.0, "res32": 0.0, "p": 0.0, "sat": 0.0}
for case in range(ncases):
    rng = np.random.default_rng(seed0 + case)
    nx, ny, nz = (int(v) for v in rng.integers(2, 9, 3))
    kw = dict(lognormal_sigma=float(rng.uniform(0.0, 1.5)), seed=int(seed0 + case))
    if rng.random() < 0.4:
        kw["nnc_fraction"] = float(rng.uniform(0.02, 0.1))
    if rng.random() < 0.4:
        kw["actnum"] = rng.random(nx * ny * nz) > rng.uniform(0.1, 0.5)
    if rng.random() < 0.4:
        kw["thpres"] = float(rng.uniform(0.01, 0.2)) * decks.BAR
    try:
        grid = decks.cartesian_grid(nx, ny, nz, **kw)
    except Exception as e:          # e.g. every cell inactive
        print("case", case, "skipped:", e); continue
    if grid.nc < 2:
        continue
    tkw = {}
    if rng.random() < 0.4:
        tkw["vappars"] = (float(rng.uniform(0.1, 2.0)), float(rng.uniform(0.1, 2.0)))
    if rng.random() < 0.4:
        tkw["rocktab"] = [(100.0, 0.97, 0.94), (200.0, 1.0, 1.0), (300.0, 1.02, 1.07), (500.0, 1.05, 1.1)]
    tab = decks.satfunc_standard_tables(pc_scale=float(rng.choice([0.0, 1.0, 3.0])), **tkw)
    if rng.random() < 0.4:
        grid = decks.with_endpoints(grid, decks.random_endpoints(grid, seed=int(seed0 + case)))
    ordering = int(rng.integers(0, 2))
    prm = capi.default_params(ilu_ordering=ordering)
    scale = tuple(prm.matbalscale)
    dt = float(rng.uniform(0.1, 30.0)) * decks.DAY
    rowptr, col = orc.pattern(grid)
    st = decks.random_state(grid, tab, seed=int(seed0 + case))
    st2 = decks.random_state(grid, tab, seed=int(seed0 + case + 7777)); st2.hc[:] = st.hc
    so_max = None
    if "vappars" in tkw:
        so_max = np.maximum(st.sat[:, 1], rng.uniform(0.2, 0.9, grid.nc))
    m = GpuBlackoilModel(grid, tab, prm)
    try:
        orc.set_sat_oil_max(so_max)
        m.prepareStep(dt, st)
        if so_max is not None:
            m.setSatOilMax(so_max)
        for single in (False, True):
            m.setSolvePrecision(single)
            m.setState(st); m.assemble(True)
            r0, v0, acc0, _ = orc.assemble(grid, tab, dt, st, rowptr, col, scale=scale)
            m.setState(st2); m.assemble(False)
            r1, v1, _, binv1 = orc.assemble(grid, tab, dt, st2, rowptr, col, scale=scale, accum0=acc0)
            gv, gr = m.jacobian()[2], m.residual()
            k = "32" if single else "64"
            worst["jac" + k] = max(worst["jac" + k], rel(gv, v1)); worst["res" + k] = max(worst["res" + k], rel(gr, r1))
            assert rel(gr, r1) < 1e-10, ("residual", case, single, rel(gr, r1))
            assert rel(gv, v1) < (2e-6 if single else 1e-10), ("jacobian", case, single, rel(gv, v1))
        nc = grid.nc
        dx = np.concatenate([rng.standard_normal(nc) * 30 * decks.BAR, rng.standard_normal(nc) * 0.25,
                             rng.standard_normal(nc) * np.where(st2.hc == capi.HC_OIL_ONLY, 30.0, np.where(st2.hc == capi.HC_GAS_ONLY, 1e-4, 0.25))])
        m.updateState(dx)
        g, o = m.getState(), orc.update_state(grid, tab, prm, dx, st2)
        assert np.array_equal(g.hc, o.hc), ("hc", case)
        worst["p"] = max(worst["p"], rel(g.p, o.p)); worst["sat"] = max(worst["sat"], float(np.abs(g.sat - o.sat).max()))
        assert rel(g.p, o.p) < 1e-13 and np.abs(g.sat - o.sat).max() < 1e-13, ("update", case)
    finally:
        orc.set_sat_oil_max(None)
        m.close()
print("cases", ncases, "worst", {k: "%.2e" % v for k, v in worst.items()}, flush=True)
