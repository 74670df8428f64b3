#!/usr/bin/env python3
"""cpr_stage2_relax (damping of the stage-2 ILU0 alone under CPR: 1.0 = the reference's form, 0.9 what rounds 1-3 ran; cpr_relax itself only
scales the whole preconditioner, which a Krylov method does not notice) against the
iteration counts, per deck, Krylov method and reduction: first Newton iterations of one time step from the deck's initial state."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd")); sys.path.insert(0, ROOT)
from opmgpu import capi, decks, baseline_decks, wells as W
from opmgpu.model import GpuBlackoilModel, LinearSolverProblem

for name in sys.argv[1:] or ["nornelike", "spe9like", "cart60"]:
    grid, tab, st, wl = baseline_decks.make(name)
    dt = baseline_decks.DT_DAYS[name] * decks.DAY
    for gm_ in (0, 1):
        for red, maxit in ((1e-2, 50), (1e-10, 2000)):
            for relax in (1.0, 0.95, 0.9):
                m = GpuBlackoilModel(grid, tab, capi.default_params(newton_use_gmres=gm_, linear_solver_reduction=red, linear_solver_maxiter=maxit, cpr_stage2_relax=relax, **capi.CPR_AMG_VCYCLE))
                md = W.DeviceWellModel(m, wl, W.WellState(wl, st.p))
                md.prepareStep(dt, st)
                its = []
                try:
                    for it in range(3):
                        conv, lin = md.nonlinearIteration(it, single_precision=False)
                        its.append(lin)
                except LinearSolverProblem:
                    its.append("FAILED")
                print("%-10s %-8s red %-6g cpr_stage2_relax %.2f  linear its %s" % (name, "gmres" if gm_ else "bicgstab", red, relax, its), flush=True)
                m.close()
