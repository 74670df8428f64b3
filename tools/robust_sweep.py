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

from opmgpu import baseline_decks, capi, decks, timestepping as ts, wells as W  # noqa: E402
from opmgpu.model import GpuBlackoilModel, NonlinearSolver  # noqa: E402

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
configs = (sys.argv[3] if len(sys.argv) > 3 else "cpr_bicgstab,cpr_gmres").split(",")
KW = {"cpr_bicgstab": dict(capi.CPR_AMG_VCYCLE), "cpr_gmres": dict(capi.CPR_AMG_VCYCLE, newton_use_gmres=1), "ilu0": dict(use_cpr=0),
      "cpr_ref_defaults": dict(use_cpr=1), "cpr_mixed": dict(capi.CPR_AMG_VCYCLE, preconditioner_single=1), "cpr_ilu1": dict(capi.CPR_AMG_VCYCLE, cpr_ilu_n=1),
      "ilu1": dict(use_cpr=0, ilu_fillin_level=1),
      # "_ref" suffix: the reference default solver's precision switch (float below dt = 20 d, BlackoilModelBase_impl.hpp:284); "_f32": float always
      "ilu0_ref": dict(use_cpr=0), "ilu0_f32": dict(use_cpr=0), "cpr_f32": dict(capi.CPR_AMG_VCYCLE)}
tot = {c: {"substeps": 0, "failed": 0, "causes": {}, "wall": 0.0, "aborted": 0} for c in configs}
for case in range(ncases):
    grid, tab, st, wl, desc = baseline_decks.random_irregular(seed0 + case, options=bool(os.environ.get("OPMGPU_SWEEP_OPTIONS")), bhp_limits=not os.environ.get("OPMGPU_SWEEP_NO_LIMITS"))
    line = "case %d: %s" % (seed0 + case, desc)
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
                    sp = True if c.endswith("_f32") else ((m.m.dt < 20 * decks.DAY) if c.endswith("_ref") else False)
                    out = solver.step(m, single_precision=sp)
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
