#!/usr/bin/env python3
"""Random irregular decks (inactive cells, NNCs, threshold pressures, heterogeneity up to sigma_lnK = 2, vertical wells on mixed controls) through
AdaptiveTimeStepping + NonlinearSolver with device wells for 200 days each: failed sub-steps by cause, per linear-solver configuration.
What the year on the Norne-like deck found (a pressure stage that is no contraction on an irregular graph) is the kind of defect this looks for.
    python tools/robust_sweep.py [ncases=12] [seed0=5000] [configs=cpr_bicgstab,cpr_gmres]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd"))
import numpy as np  # noqa: E402

from opmgpu import capi, decks, timestepping as ts, wells as W  # noqa: E402
from opmgpu.model import GpuBlackoilModel, NonlinearSolver  # noqa: E402

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
configs = (sys.argv[3] if len(sys.argv) > 3 else "cpr_bicgstab,cpr_gmres").split(",")
KW = {"cpr_bicgstab": dict(capi.CPR_AMG_VCYCLE), "cpr_gmres": dict(capi.CPR_AMG_VCYCLE, newton_use_gmres=1), "ilu0": dict(use_cpr=0),
      "cpr_ref_defaults": dict(use_cpr=1), "cpr_mixed": dict(capi.CPR_AMG_VCYCLE, preconditioner_single=1), "cpr_ilu1": dict(capi.CPR_AMG_VCYCLE, cpr_ilu_n=1),
      "ilu1": dict(use_cpr=0, ilu_fillin_level=1)}
tot = {c: {"substeps": 0, "failed": 0, "causes": {}, "wall": 0.0, "aborted": 0} for c in configs}
for case in range(ncases):
    rng = np.random.default_rng(seed0 + case)
    nx, ny, nz = int(rng.integers(20, 50)), int(rng.integers(20, 50)), int(rng.integers(5, 20))
    inactive = float(rng.uniform(0.0, 0.6))
    kw = dict(dx=float(rng.uniform(30, 120)), dy=float(rng.uniform(30, 120)), dz=float(rng.uniform(2, 8)), tops=2500.0, lognormal_sigma=float(rng.uniform(0.3, 2.0)), seed=seed0 + case)
    if inactive > 0.05:
        kw["actnum"] = rng.random(nx * ny * nz) > inactive
    if rng.random() < 0.6:
        kw["nnc_fraction"] = float(rng.uniform(0.01, 0.06))
    if rng.random() < 0.4:
        kw["thpres"] = float(rng.uniform(0.01, 0.05)) * decks.BAR
    grid = decks.cartesian_grid(nx, ny, nz, **kw)
    tkw, opts = {}, []
    if os.environ.get("OPMGPU_SWEEP_OPTIONS"):          # the saturation-function / rock options the parity fuzzers draw, through whole time steps
        if rng.random() < 0.4:
            tkw["vappars"] = (float(rng.uniform(0.1, 2.0)), float(rng.uniform(0.1, 2.0))); opts.append("vappars")
        if rng.random() < 0.4:
            tkw["rocktab"] = [(100.0, 0.97, 0.94), (200.0, 1.0, 1.0), (300.0, 1.02, 1.07), (500.0, 1.05, 1.1)]; opts.append("rocktab")
    tab = decks.satfunc_standard_tables(**tkw)
    if os.environ.get("OPMGPU_SWEEP_OPTIONS") and rng.random() < 0.5:
        grid = decks.with_endpoints(grid, decks.random_endpoints(grid, seed=seed0 + case)); opts.append("endscale")
    st = decks.initial_state(grid, tab, p_ref=270.0 * decks.BAR, z_ref=2500.0, perturb=0.005, seed=seed0 + case)
    nwells = int(rng.integers(3, 20))
    wl = W.column_wells(grid, nwells, n_injectors=max(1, nwells // 6), seed=seed0 + case, inj_rate_m3_per_day=float(rng.uniform(50, 400)),
                        prod_bhp_bar=float(rng.uniform(150, 230)), prod_oil_rate_m3_per_day=float(rng.uniform(10, 60)),
                        rate_wells_bhp_limits_bar=None if os.environ.get("OPMGPU_SWEEP_NO_LIMITS") else (450.0, 80.0))
    line = "case %d: %dx%dx%d, %d active, inactive %.2f, sigma %.2f, %d wells" % (seed0 + case, nx, ny, nz, grid.nc, inactive, kw["lognormal_sigma"], wl.nw) + (" " + "+".join(opts) if opts else "")
    for c in configs:
        gm = GpuBlackoilModel(grid, tab, capi.default_params(**KW[c]))
        model = W.DeviceWellModel(gm, wl, W.WellState(wl, st.p))
        gm.setState(st)
        ats = ts.AdaptiveTimeStepping(initial_timestep_days=1.0)
        solver = NonlinearSolver()

        trace = bool(os.environ.get("OPMGPU_SWEEP_TRACE"))

        class S:
            def step(self, m):
                if trace:
                    print("   %s substep dt %.3e d" % (c, m.m.dt / decks.DAY), flush=True)
                try:
                    out = solver.step(m, single_precision=False)
                except Exception as e:
                    if trace:
                        print("      failed: %r" % (e,), flush=True)
                    raise
                if trace:
                    print("      newton %d linear %d relative change %.3e" % (out[0], out[1], m.relativeChange()), flush=True)
                return out
        t, sub, failed, causes, status = 0.0, 0, 0, {}, "ok"
        t0 = time.time()
        try:
            while t < 200 * decks.DAY - 1e-6:
                rep = ats.step(t, 40 * decks.DAY, S(), model)
                sub += len(rep["substeps"]); failed += len(rep["failed"])
                for _, cause in rep["failed"]:
                    causes[cause] = causes.get(cause, 0) + 1
                t += 40 * decks.DAY
        except Exception as e:
            rep = getattr(e, "report", None)
            status = repr(e)[:80] + ("" if rep is None else " after " + "; ".join("%.3g d: %s" % (d / decks.DAY, cs.split(" - ")[-1]) for d, cs in rep["failed"][-11:]))
            tot[c]["aborted"] += 1
        wall = time.time() - t0
        tot[c]["substeps"] += sub; tot[c]["failed"] += failed; tot[c]["wall"] += wall
        for k, v in causes.items():
            tot[c]["causes"][k] = tot[c]["causes"].get(k, 0) + v
        line += " | %s: %d substeps, %d failed %s %.2f s %s" % (c, sub, failed, {k.split(" - ")[-1][:22]: v for k, v in causes.items()}, wall, "" if status == "ok" else status)
        gm.close()
    print(line, flush=True)
print("TOTAL", json.dumps(tot))
