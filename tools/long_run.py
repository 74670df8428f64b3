#!/usr/bin/env python3
"""A YEAR of the headline deck through the reference's own control loops -- AdaptiveTimeStepping (PID control, restart on failure) around
NonlinearSolver (update stabilisation) around the device Newton path with device wells -- for three linear-solver configurations:
wall time, sub-steps, failed sub-steps, Newton / linear iterations, and the material balance of the whole run (change of each component's
surface volume in place against the wells' surface rates integrated over the converged sub-steps).

    python tools/long_run.py [--days 365] [--report-days 30] [--deck cart100|spe10like] [--configs cpr_bicgstab,cpr_gmres,ilu0_default]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd"))
import numpy as np  # noqa: E402

from opmgpu import baseline_decks, capi, decks, timestepping as ts, wells as W  # noqa: E402
from opmgpu.model import GpuBlackoilModel, NonlinearSolver  # noqa: E402

CONFIGS = {
    # solver_approach=cpr with cpr_use_amg (one V-cycle per application), the CPR plug-in's default Krylov method, double
    "cpr_bicgstab": (dict(capi.CPR_AMG_VCYCLE), False),
    "cpr_bicgstab_damped": (dict(capi.CPR_AMG_VCYCLE, cpr_stage2_relax=0.9), False),
    # the bench headline: newton_use_gmres, dune's stopping rule
    "cpr_gmres": (dict(capi.CPR_AMG_VCYCLE, newton_use_gmres=1), False),
    # library extension: the double solve with its preconditioner in float
    "cpr_bicgstab_mixed": (dict(capi.CPR_AMG_VCYCLE, preconditioner_single=1), False),
    # block ILU(1) as the second stage (cpr_ilu_n) / as the interleaved solver's preconditioner (ilu_fillin_level)
    "cpr_bicgstab_ilu1": (dict(capi.CPR_AMG_VCYCLE, cpr_ilu_n=1), False),
    "ilu1_default": (dict(use_cpr=0, ilu_fillin_level=1), "reference"),
    # the CPR plug-in's documented defaults: ILU0-preconditioned inner BiCGStab on the pressure system
    "cpr_ref_defaults": (dict(use_cpr=1), False),
    "cpr_amg_inner": (dict(use_cpr=1, cpr_use_amg=1), False),
    # the reference's default: solver_approach=interleaved, ILU0 + BiCGStab, float below dt = 20 d
    "ilu0_default": (dict(use_cpr=0), "reference"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--days", type=float, default=365.0)
    ap.add_argument("--report-days", type=float, default=30.0)
    ap.add_argument("--deck", default="cart100", choices=["cart100", "spe10like", "cart60", "spe9like", "nornelike", "irregular_big"])
    ap.add_argument("--configs", default="cpr_bicgstab,cpr_gmres,ilu0_default")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if args.deck == "irregular_big":
        # the Norne-like recipe at 10^6 cells of box: 100 x 200 x 50 with 40 % of the cells inactive at random (~600 k active), 3 % extra NNCs, threshold
        # pressures, sigma_lnK = 1.2, 60 vertical wells on mixed controls with BHP limits
        rng = np.random.default_rng(77)
        act = rng.random(100 * 200 * 50) > 0.4
        grid = decks.cartesian_grid(100, 200, 50, dx=60.0, dy=60.0, dz=3.0, tops=2500.0, actnum=act, nnc_fraction=0.03, lognormal_sigma=1.2, thpres=0.02 * decks.BAR, seed=77)
        tab = decks.satfunc_standard_tables()
        st = decks.initial_state(grid, tab, p_ref=270.0 * decks.BAR, z_ref=2500.0, perturb=0.005, seed=77)
        wl = W.column_wells(grid, 60, n_injectors=10, seed=77, inj_rate_m3_per_day=400.0, prod_bhp_bar=200.0, prod_oil_rate_m3_per_day=60.0, rate_wells_bhp_limits_bar=(450.0, 80.0))
    else:
        grid, tab, st, wl = baseline_decks.make(args.deck)
    out = {"deck": args.deck, "cells": int(grid.nc), "days": args.days, "report_days": args.report_days, "runs": {}}
    for name in args.configs.split(","):
        kw, single = CONFIGS[name]
        gm = GpuBlackoilModel(grid, tab, capi.default_params(**kw))
        model = W.DeviceWellModel(gm, wl, W.WellState(wl, st.p))
        gm.setState(st)
        comp = lambda v: np.array([v[0, 0], v[0, 1] + v[0, 4], v[0, 2] + v[0, 3]])          # water, oil (+ vaporised), gas (+ dissolved)
        fip0 = comp(gm.computeFluidInPlace())
        produced = np.zeros(3)

        class Solver:
            """NonlinearSolver.step + the reference default solver's precision switch on the current step length; integrates the wells' rates"""
            def __init__(self):
                self.inner = NonlinearSolver()

            def step(self, m):
                sp = (m.m.dt < 20 * decks.DAY) if single == "reference" else single
                res = self.inner.step(m, single_precision=sp)
                ws = m.pull_well_state()
                produced[:] += ws.qs.sum(axis=0) * m.m.dt
                return res

        ats = ts.AdaptiveTimeStepping(initial_timestep_days=1.0)
        solver = Solver()
        t, newton, linear, substeps, failed = 0.0, 0, 0, 0, 0
        causes = {}
        t0 = time.time()
        status = "ok"
        try:
            while t < args.days * decks.DAY - 1e-6:
                step = min(args.report_days * decks.DAY, args.days * decks.DAY - t)
                rep = ats.step(t, step, solver, model)
                # (rates of failed sub-steps never reach `produced`: Solver.step adds only after a converged NonlinearSolver.step)
                newton += rep["newton_iterations"]; linear += rep["linear_iterations"]; substeps += len(rep["substeps"]); failed += len(rep["failed"])
                t += step
                for _, cause in rep["failed"]:
                    causes[cause] = causes.get(cause, 0) + 1
        except Exception as e:          # solver_restart_max consecutive failures
            status = repr(e)
        import torch
        torch.cuda.synchronize()
        wall = time.time() - t0
        fip1 = comp(gm.computeFluidInPlace())
        change = fip1 - fip0
        # the default tolerances (tolerance_mb 1e-5 of the pore volume per step) bound the imbalance; report it relative to what the wells moved
        moved = np.abs(produced).max()
        out["runs"][name] = {
            "status": status, "failure_causes": causes, "simulated_days": t / decks.DAY, "wall_s": round(wall, 2), "substeps": substeps, "failed_substeps": failed,
            "newton_iterations": newton, "linear_iterations": linear, "ms_per_simulated_day": round(1e3 * wall / max(t / decks.DAY, 1e-9), 3),
            "newton_per_substep": round(newton / max(substeps, 1), 2), "linear_per_newton": round(linear / max(newton, 1), 2),
            "in_place_change_wog_sm3": [float(x) for x in change], "wells_integrated_wog_sm3": [float(x) for x in produced],
            "imbalance_rel_to_moved": [float(abs(c - p) / moved) for c, p in zip(change, produced)],
        }
        print(name, json.dumps(out["runs"][name]), flush=True)
        gm.close()
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
