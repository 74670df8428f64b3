"""Randomised campaign: the device well model (csrc/wells.hip) against the INDEPENDENT restatement oracle/wells.py (complex-step Jacobian, one
coupled sparse system, direct solve) on the random cases of tests/test_oracle_wells.py (type, BHP / surface-rate / reservoir-rate control
with a BHP limit, perforation range, cross-flow flag; ILU0 or CPR by seed), three Newton iterations each.
    python tools/fuzz_wells_independent.py [ncases] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from opmgpu import capi, wells as W
from opmgpu.model import GpuBlackoilModel
from oracle.wells import CoupledOracleModel, NumericalIssue
from test_oracle_wells import _random_case, _arrays

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
worst = {"p": 0.0, "sat": 0.0, "bhp": 0.0, "qs": 0.0}
done = skipped = switches = knife = 0
for case in range(ncases):
    grid, tab, st, make, start, dt = _random_case(seed0 + case)
    prm = capi.default_params(linear_solver_reduction=1e-11, linear_solver_maxiter=800, cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=(seed0 + case) % 2)
    wl_d, wl_i = make(), make()
    gm = GpuBlackoilModel(grid, tab, prm)
    try:
        md = W.DeviceWellModel(gm, wl_d, start(wl_d))
        mi = CoupledOracleModel(grid, tab, prm, wl_i, _arrays(start(wl_i)))
        md.prepareStep(dt, st); mi.prepareStep(dt, st)
        for it in range(3):
            try:
                ci = mi.nonlinearIteration(it)
            except NumericalIssue:          # the checker side gives up on this random case: not a parity statement
                skipped += 1; break
            cd, _ = md.nonlinearIteration(it, single_precision=False)
            ws = md.pull_well_state()
            if (np.abs(mi.ws.qs).max(1) < 1e-12).any():
                # a well has stopped flowing (e.g. a producer switched to a BHP limit above the reservoir pressure): the reference's dead-well
                # test -- wellbore rate EXACTLY zero, StandardWells_impl.hpp:565-570 -- is then decided by the rounding of a 1e-18, in every
                # implementation alike; what follows is not a parity statement (the implementations re-converge one iteration later)
                knife += 1; break
            if not np.array_equal(ws.current, mi.ws.current):
                # the other knife edge of updateWellControls: right after a well Newton step under a rate control, distr . q_s equals the target
                # up to rounding, so `constraintBroken` (a strict inequality) is decided by the last bit when the loop revisits that control
                gaps = []
                for w_ in np.flatnonzero(ws.current != mi.ws.current):
                    for typ, target, distr in mi.controls[w_]:
                        val = mi.ws.bhp[w_] if typ == W.BHP else float(np.dot(distr, mi.ws.qs[w_]))
                        gaps.append(abs(val - target) / max(abs(target), 1e-300))
                assert min(gaps) < 1e-9, (case, it, ws.current, mi.ws.current, gaps)
                knife += 1; break
            assert cd == ci, (case, it)
            a, b = gm.getState(), mi.st
            assert np.array_equal(a.hc, b.hc), (case, it)
            e = {"p": np.abs(a.p - b.p).max() / np.abs(b.p).max(), "sat": np.abs(a.sat - b.sat).max(),
                 "bhp": np.abs(ws.bhp - mi.ws.bhp).max() / max(np.abs(mi.ws.bhp).max(), 1.0),
                 "qs": np.abs(ws.qs - mi.ws.qs).max() / max(np.abs(mi.ws.qs).max(), 1e-12)}
            for k, v in e.items():
                worst[k] = max(worst[k], float(v))
            assert e["p"] < 1e-6 and e["sat"] < 1e-6 and e["bhp"] < 1e-6 and e["qs"] < 1e-5, (case, it, e)
        else:
            done += 1
            switches += int((mi.ws.current != 0).any())
    finally:
        gm.close()
print("cases", ncases, "compared", done, "skipped", skipped, "stopped at a dead-well knife edge", knife, "with a control switch", switches, "worst", {k: "%.1e" % v for k, v in worst.items()}, flush=True)
