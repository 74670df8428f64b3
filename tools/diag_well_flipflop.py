#!/usr/bin/env python3
"""Diagnostic for a deck of tools/robust_sweep.py whose Newton loop stops converging at ANY step length: replays the deck until the first
sub-step shorter than 1e-3 d fails, then repeats that sub-step printing, per Newton iteration, the reservoir's and the wells' convergence
flags, the wells' residuals and every well's current control.
    python tools/diag_well_flipflop.py <seed> [config=cpr_bicgstab]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd"))
import numpy as np  # noqa: E402

from opmgpu import baseline_decks, capi, decks, timestepping as ts, wells as W  # noqa: E402
from opmgpu.model import GpuBlackoilModel, NonlinearSolver  # noqa: E402

seed = int(sys.argv[1])
cfg = sys.argv[2] if len(sys.argv) > 2 else "cpr_bicgstab"
KW = {"cpr_bicgstab": dict(capi.CPR_AMG_VCYCLE), "cpr_gmres": dict(capi.CPR_AMG_VCYCLE, newton_use_gmres=1), "ilu0": dict(use_cpr=0)}
grid, tab, st, wl, desc = baseline_decks.random_irregular(seed, bhp_limits=not os.environ.get("OPMGPU_SWEEP_NO_LIMITS"))
nx, ny, nz = grid.dims
print("deck %d: %dx%dx%d, %d active, %d wells" % (seed, nx, ny, nz, grid.nc, wl.nw))
for w in range(wl.nw):
    print("  well %d %s type %d perforations %d controls %s" % (w, wl.name[w], wl.type[w], wl.connpos[w + 1] - wl.connpos[w], [(c[0], c[1]) for c in wl.controls[w]]))
gm = GpuBlackoilModel(grid, tab, capi.default_params(**KW[cfg]))
model = W.DeviceWellModel(gm, wl, W.WellState(wl, st.p))
gm.setState(st)
solver = NonlinearSolver()


class Found(Exception):
    pass


class S:
    def step(self, m):
        try:
            return solver.step(m, single_precision=False)
        except Exception:
            if m.m.dt < 1e-3 * decks.DAY:
                raise Found()
            raise


ats = ts.AdaptiveTimeStepping(initial_timestep_days=1.0)
t = 0.0
try:
    while t < 400 * decks.DAY:
        ats.step(t, 40 * decks.DAY, S(), model)
        t += 40 * decks.DAY
    print("no failing short sub-step found")
    sys.exit(0)
except Found:
    dt = gm.dt
print("first failing short sub-step: dt = %.3e d; repeating it from the saved state" % (dt / decks.DAY))
model.restoreState()
model.prepareStep(dt)
for it in range(12):
    gm.setSolvePrecision(False)
    gm.assemble(it == 0)
    conv_res = gm.getConvergence()
    conv_w = model.wellConvergence()
    ws = model.pull_well_state()
    print("it %2d reservoir %s (CNV %s MB %s) wells %s (flux %s ctrl %.2e) current %s" % (
        it, conv_res, np.array2string(np.asarray(gm.CNV), precision=1), np.array2string(np.asarray(gm.MB), precision=1), conv_w,
        np.array2string(model.well_flux_residual, precision=1), model.well_ctrl_residual, ws.current.tolist()))
    print("      bhp[bar] %s" % np.array2string(ws.bhp / decks.BAR, precision=1, max_line_width=200))
    print("      qs oil   %s" % np.array2string(ws.qs[:, 1] * 86400, precision=2, max_line_width=200))
    gm.solveJacobianSystem(single_precision=False)
    gm.updateState()
gm.close()
