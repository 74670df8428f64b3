"""Randomised campaign for the device well model (csrc/wells.hip) against the CPU oracle driven by the host well model with the explicit
Schur complement: random small decks, 1-4 wells with random type / control (BHP, surface rate, or RESV: reservoir-volume rate with the
coefficients of RateConverter / computeRESV, device against oracle restatement) / perforations / crossflow flag, ILU0 or CPR, three Newton
iterations each: reservoir state, well state and well residuals must agree.      python tools/fuzz_wells.py [ncases] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from opmgpu import capi, decks, wells as W
from opmgpu.model import GpuBlackoilModel
from opmgpu.rateconverter import SurfaceToReservoirVoidage, computeRESV
from oracle import oracle as orc
from oracle.rateconverter import SurfaceToReservoirVoidage as OracleVoidage
from util import OracleBackend

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 300
worst = {"p": 0.0, "sat": 0.0, "bhp": 0.0, "qs": 0.0, "flux_res": 0.0}
done = skipped = 0
for case in range(ncases):
    rng = np.random.default_rng(seed0 + case)
    nx, ny, nz = int(rng.integers(4, 9)), int(rng.integers(4, 9)), int(rng.integers(2, 6))
    grid = decks.cartesian_grid(nx, ny, nz, dx=100.0, dy=100.0, dz=5.0, tops=2500.0, poro=0.25, permx_md=150.0, lognormal_sigma=float(rng.uniform(0, 0.8)), seed=seed0 + case)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, p_ref=250 * decks.BAR, z_ref=2500.0, gas_cap_fraction=0.0, gas_only_fraction=0.0, perturb=0.002, seed=seed0 + case)
    nw = int(rng.integers(1, 5))
    cols = rng.choice(nx * ny, size=nw, replace=False)
    wl = W.Wells()
    WI = float(rng.uniform(1.0, 8.0)) * float(np.median(grid.trans))
    pv_rate = float(grid.pv.sum()) / (8000.0 * decks.DAY)                     # a pore volume in ~20 years: deliverable by the random well indices
    have_bhp = False
    for w in range(nw):
        k0 = int(rng.integers(0, nz)); k1 = int(rng.integers(k0 + 1, nz + 1))
        cells = [int(cols[w]) + nx * ny * k for k in range(k0, k1)]
        inj = rng.random() < 0.5
        resv = rng.random() < 0.3
        bhp_ctrl = rng.random() < 0.5 or (w == nw - 1 and not have_bhp)          # at least one pressure control anchors the box
        have_bhp = have_bhp or bhp_ctrl
        if inj:
            ctrl = (W.BHP, float(rng.uniform(260, 300)) * decks.BAR) if bhp_ctrl else \
                   (W.RESERVOIR_RATE, float(rng.uniform(0.2, 1.0)) * pv_rate, (1.0, 1.0, 1.0)) if resv else (W.SURFACE_RATE, float(rng.uniform(0.2, 1.0)) * pv_rate, (1.0, 0.0, 0.0))
            wl.add_well("I%d" % w, W.INJECTOR, grid.z[cells[0]], cells, WI, (1.0, 0.0, 0.0), ctrl, allow_cf=bool(rng.random() < 0.7))
        else:
            ctrl = (W.BHP, float(rng.uniform(180, 240)) * decks.BAR) if bhp_ctrl else \
                   (W.RESERVOIR_RATE, -float(rng.uniform(0.2, 1.0)) * pv_rate, (1.0, 1.0, 1.0)) if resv else (W.SURFACE_RATE, -float(rng.uniform(0.2, 1.0)) * pv_rate, (0.0, 1.0, 0.0))
            wl.add_well("P%d" % w, W.PRODUCER, grid.z[cells[0]], cells, WI, (0.0, 1.0, 0.0), ctrl, allow_cf=bool(rng.random() < 0.7))
    cpr = int(rng.integers(0, 2))
    prm = capi.default_params(linear_solver_reduction=1e-11, linear_solver_maxiter=800, cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=cpr)
    prm_o = capi.default_params(linear_solver_reduction=1e-11, linear_solver_maxiter=800)
    dt = float(rng.uniform(0.5, 5.0)) * decks.DAY
    gm = GpuBlackoilModel(grid, tab, prm)
    ob = OracleBackend(orc, grid, tab, prm_o, wells=wl.arrays())
    try:
        # pressure-controlled wells start with a small rate in their flowing direction: from q_s = 0 the reference's dead-well test
        # (wellbore rate EXACTLY zero -> control equation sum q_s = 0, StandardWells_impl.hpp:486-506) sits on a knife edge -- after one
        # iteration q_s is +-1e-21 and its rounding decides whether the well is alive, on the device and on the host alike
        def start_state():
            w0 = W.WellState(wl, st.p)
            for w_ in range(wl.nw):
                if wl.ctrl_type[w_] in (W.BHP, W.RESERVOIR_RATE):          # neither control seeds the rates (updateWellStateWithTarget)
                    w0.qs[w_] = (1e-5 if wl.type[w_] == W.INJECTOR else -1e-5) * np.asarray(wl.comp_frac[w_])
            return w0
        md = W.DeviceWellModel(gm, wl, start_state())
        mo = W.WellCoupledModel(ob, W.StandardWellsHost(wl, grid.z, tab.surface_density[0]), start_state())
        md.prepareStep(dt, st); mo.prepareStep(dt, st)
        # computeRESV (SimulatorBase_impl.hpp:476-553): the device's coefficients go into the shared controls, the oracle's restatement must agree
        if computeRESV(SurfaceToReservoirVoidage(gm), wl, device_wells=md):
            want = OracleVoidage(tab, np.zeros(grid.nc, int)).defineState(st.p, st.rs, st.rv).calcCoeff(0, 0)
            for w_ in range(wl.nw):
                if wl.controls[w_][0][0] == W.RESERVOIR_RATE:
                    assert np.allclose(wl.controls[w_][0][2], want, rtol=1e-13, atol=0.0), (case, wl.controls[w_][0][2], want)
        for it in range(3):
            try:
                co, _ = mo.nonlinearIteration(it, single_precision=False)
            except Exception as e:      # the oracle side gives up on this random case (solver / numerical issue): not a parity statement
                skipped += 1; break
            cd, _ = md.nonlinearIteration(it, single_precision=False)
            assert cd == co, ("converged flag", case, it)
            a, b, ws = gm.getState(), ob.getState(), md.pull_well_state()
            if os.environ.get("FUZZ_VERBOSE") and case == int(os.environ.get("FUZZ_CASE", "-1")):
                np.set_printoptions(linewidth=250, precision=6)
                print(" it", it, "wellres gpu", md.well_flux_residual, md.well_ctrl_residual, "ref", mo.wh.well_flux_residual, mo.wh.well_ctrl_residual)
                print("   bhp gpu", ws.bhp, "ref", mo.ws.bhp)
                print("   qs gpu", ws.qs.ravel(), "\n   qs ref", mo.ws.qs.ravel())
                print("   perf rates gpu", ws.perf_rates.ravel(), "\n   perf rates ref", mo.ws.perf_rates.ravel(), flush=True)
            if os.environ.get("FUZZ_VERBOSE"):
                print("case", case, "it", it, "dp", np.abs(a.p - b.p).max() / np.abs(b.p).max(), "dsat", np.abs(a.sat - b.sat).max(), "drs", np.abs(a.rs - b.rs).max(),
                      "hc counts gpu", np.bincount(a.hc, minlength=3), "ref", np.bincount(b.hc, minlength=3), "lin", md.linear_iterations, mo.linear_iterations, flush=True)
            if not np.array_equal(a.hc, b.hc):
                d = np.flatnonzero(a.hc != b.hc)
                print("case", case, "it", it, "hc differs in", d.size, "cells", d[:5], "gpu hc", a.hc[d[:5]], "ref hc", b.hc[d[:5]], "gpu sat", a.sat[d[:5]].tolist(), "ref sat", b.sat[d[:5]].tolist(),
                      "gpu rs", a.rs[d[:5]], "ref rs", b.rs[d[:5]], "p", a.p[d[:5]], b.p[d[:5]], "max dp", np.abs(a.p - b.p).max(), flush=True)
            assert np.array_equal(a.hc, b.hc), ("hc", case, it)
            e = {"p": np.abs(a.p - b.p).max() / np.abs(b.p).max(), "sat": np.abs(a.sat - b.sat).max(),
                 "bhp": np.abs(ws.bhp - mo.ws.bhp).max() / max(np.abs(mo.ws.bhp).max(), 1.0),
                 "qs": np.abs(ws.qs - mo.ws.qs).max() / max(np.abs(mo.ws.qs).max(), 1e-12),
                 "flux_res": np.abs(md.well_flux_residual - mo.wh.well_flux_residual).max() / max(np.abs(mo.wh.well_flux_residual).max(), 1e-9)}
            if not all(np.isfinite(v) for v in e.values()):
                print("case", case, "it", it, "non-finite:", "gpu bhp", ws.bhp, "host bhp", mo.ws.bhp, "gpu qs", ws.qs.ravel(), "host qs", mo.ws.qs.ravel(),
                      "types", wl.type, "ctrl", wl.ctrl_type, wl.ctrl_target, flush=True)
            for k, v in e.items():
                worst[k] = max(worst[k], float(v))
            assert e["p"] < 1e-6 and e["sat"] < 1e-6 and e["bhp"] < 1e-6 and e["qs"] < 1e-5, (case, it, e)
        else:
            done += 1
    finally:
        gm.close()
print("cases", ncases, "compared", done, "skipped", skipped, "worst", {k: "%.1e" % v for k, v in worst.items()}, flush=True)
