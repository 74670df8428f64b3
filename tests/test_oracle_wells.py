"""oracle/wells.py -- the independent restatement of the standard well model (complex-step derivatives, one sparse system with the
well unknowns as extra rows and columns, direct solve) -- against the reference's known answer, against the product's host well model
(hand-written derivatives, Schur complement, ILU0 + BiCGStab on the oracle) and, on the GPU box, against the device well model."""
import json
import os

import numpy as np
import pytest

from opmgpu import capi, decks, wells as W
from util import OracleBackend

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _arrays(ws):
    from oracle.wells import WellStateArrays
    return WellStateArrays(ws.bhp, ws.qs, ws.perf_press, ws.perf_rates, ws.current)


def test_welldensitysegmented_known_answer_on_the_independent_restatement(oracle):
    """tests/test_welldensitysegmented.cpp:108-113 (the golden file of tests/test_wells_host.py) against oracle/wells.py"""
    from oracle import wells as OW
    g = json.load(open(os.path.join(GOLD, "welldensitysegmented.json")))
    n = 10
    connpos = np.array([0, 5, 10])
    comp = [g["comp_frac_inj"], g["comp_frac_prod"]]
    cd = OW.connection_densities(connpos, comp, np.asarray(g["perf_rates"]).reshape(n, 3), np.asarray(g["b_perf"]).reshape(n, 3),
                                 np.asarray(g["rsmax_perf"]), np.asarray(g["rvmax_perf"]), np.asarray(g["surf_dens"]).reshape(n, 3))
    dp = OW.connection_pressure_delta(connpos, [g["ref_depth"]] * 2, np.asarray(g["z_perf"]), cd, g["gravity"])
    assert np.allclose(dp, np.asarray(g["answer_over_gravity"]) * g["gravity"], rtol=1e-10)


def _deck(case):
    nx, ny, nz = 10, 10, 3
    grid = decks.cartesian_grid(nx, ny, nz, dx=300.0, dy=300.0, dz=10.0, tops=2500.0, poro=0.3, permx_md=200.0, lognormal_sigma=0.3)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, p_ref=250 * decks.BAR, z_ref=2500.0, gas_cap_fraction=0.0, gas_only_fraction=0.0)
    col = lambda i, j: [i + nx * j + nx * ny * k for k in range(nz)]
    WI = 5.0 * float(np.median(grid.trans))

    def make():
        wl = W.Wells()
        if case == "rate_inj_bhp_prod":
            wl.add_well("INJ", W.INJECTOR, grid.z[col(0, 0)[0]], col(0, 0), WI, (1.0, 0.0, 0.0), (W.SURFACE_RATE, 2000.0 / 86400.0, (1.0, 0.0, 0.0)),
                        limits=[(W.BHP, 600 * decks.BAR)])
            wl.add_well("PROD", W.PRODUCER, grid.z[col(nx - 1, ny - 1)[0]], col(nx - 1, ny - 1)[:2], WI, (0.0, 1.0, 0.0), (W.BHP, 200 * decks.BAR))
        elif case == "bhp_limit_switch":      # the injector runs into its BHP limit after the first Newton update
            wl.add_well("INJ", W.INJECTOR, grid.z[col(0, 0)[0]], col(0, 0), WI, (1.0, 0.0, 0.0), (W.SURFACE_RATE, 2000.0 / 86400.0, (1.0, 0.0, 0.0)),
                        limits=[(W.BHP, 262 * decks.BAR)])
            wl.add_well("PROD", W.PRODUCER, grid.z[col(nx - 1, ny - 1)[0]], col(nx - 1, ny - 1)[:2], WI, (0.0, 1.0, 0.0), (W.BHP, 200 * decks.BAR))
        elif case == "prod_rate_limit":       # a BHP producer with an oil-rate limit it breaks, no cross flow allowed in the injector
            wl.add_well("INJ", W.INJECTOR, grid.z[col(0, 0)[0]], col(0, 0), WI, (1.0, 0.0, 0.0), (W.SURFACE_RATE, 2000.0 / 86400.0, (1.0, 0.0, 0.0)),
                        allow_cf=False, limits=[(W.BHP, 600 * decks.BAR)])
            wl.add_well("PROD", W.PRODUCER, grid.z[col(nx - 1, ny - 1)[0]], col(nx - 1, ny - 1), WI, (0.0, 1.0, 0.0), (W.BHP, 200 * decks.BAR),
                        limits=[(W.SURFACE_RATE, -150.0 / 86400.0, (0.0, 1.0, 0.0))])
        else:                                  # liquid-rate producer (two phases under the control), gas injector on BHP
            wl.add_well("GINJ", W.INJECTOR, grid.z[col(0, 0)[0]], col(0, 0)[:2], WI, (0.0, 0.0, 1.0), (W.BHP, 290 * decks.BAR))
            wl.add_well("PROD", W.PRODUCER, grid.z[col(nx - 1, ny - 1)[0]], col(nx - 1, ny - 1), WI, (0.0, 1.0, 0.0),
                        (W.SURFACE_RATE, -300.0 / 86400.0, (1.0, 1.0, 0.0)), limits=[(W.BHP, 100 * decks.BAR)])
        return wl

    def start(wl):
        ws = W.WellState(wl, st.p)
        for w in range(wl.nw):               # pressure-controlled wells start with a small rate (the dead-well knife edge, tools/fuzz_wells.py)
            if wl.ctrl_type[w] == W.BHP or (wl.ctrl_type[w] == W.SURFACE_RATE and wl.type[w] == W.PRODUCER and (np.asarray(wl.ctrl_distr[w]) > 0).sum() > 1):
                ws.qs[w] = (1e-5 if wl.type[w] == W.INJECTOR else -1e-5) * np.asarray(wl.comp_frac[w])
        return ws

    return grid, tab, st, make, start


CASES = ["rate_inj_bhp_prod", "bhp_limit_switch", "prod_rate_limit", "lrat_prod_gas_inj"]


@pytest.mark.parametrize("case", CASES)
def test_independent_well_model_agrees_with_the_host_well_model(oracle, case):
    """Same deck, same initial well state: oracle/wells.py (complex-step Jacobian, one sparse system, SuperLU) and opmgpu/wells.py on the
    OracleBackend (hand-written derivatives, Schur complement into the reservoir matrix, ILU0 + BiCGStab at 1e-12) walk the same Newton path:
    control switches, pre-solve iteration counts, well residuals, well and reservoir states."""
    from oracle.wells import CoupledOracleModel
    grid, tab, st, make, start = _deck(case)
    prm = capi.default_params(linear_solver_reduction=1e-12, linear_solver_maxiter=2000)
    dt = 5 * decks.DAY
    wl_i, wl_h = make(), make()
    mi = CoupledOracleModel(grid, tab, prm, wl_i, _arrays(start(wl_i)))
    ob = OracleBackend(oracle, grid, tab, prm, wells=wl_h.arrays())
    mh = W.WellCoupledModel(ob, W.StandardWellsHost(wl_h, grid.z, tab.surface_density[0]), start(wl_h))
    mi.prepareStep(dt, st); mh.prepareStep(dt, st)
    switched = False
    for it in range(12):
        ci = mi.nonlinearIteration(it)
        ch, _ = mh.nonlinearIteration(it, single_precision=False)
        if it == 0:
            assert mi.presolve_converged and mi.well_iterations == mh.wh.well_iterations, (mi.well_iterations, mh.wh.well_iterations)
        assert ci == ch, it
        assert np.array_equal(mi.ws.current, mh.ws.current), (it, mi.ws.current, mh.ws.current)
        switched = switched or bool((mi.ws.current != 0).any())
        assert np.allclose(mi.well_flux_residual, mh.wh.well_flux_residual, rtol=1e-6, atol=1e-13), (it, mi.well_flux_residual, mh.wh.well_flux_residual)
        assert mi.well_ctrl_residual == pytest.approx(mh.wh.well_ctrl_residual, rel=1e-6, abs=1e-9), it
        a, b = mi.st, ob.getState()
        assert np.array_equal(a.hc, b.hc), it
        assert np.abs(a.p - b.p).max() <= 1e-8 * np.abs(b.p).max() and np.abs(a.sat - b.sat).max() <= 1e-8, (it, np.abs(a.p - b.p).max(), np.abs(a.sat - b.sat).max())
        assert np.allclose(mi.ws.bhp, mh.ws.bhp, rtol=1e-8), (it, mi.ws.bhp, mh.ws.bhp)
        assert np.allclose(mi.ws.qs, mh.ws.qs, rtol=1e-7, atol=1e-10 * np.abs(mh.ws.qs).max()), (it, mi.ws.qs, mh.ws.qs)
        if ci and it > 0:
            break
    assert ci and it < 12
    if case in ("bhp_limit_switch", "prod_rate_limit"):
        assert switched


@pytest.mark.parametrize("case", CASES)
def test_well_potentials_host_model_against_the_independent_one(oracle, case):
    """StandardWells::computeWellPotentials (StandardWells_impl.hpp:1003-1095): opmgpu/wells.py (per-well dense AD system) against
    oracle/wells.py (vectorised computeWellFlux) after two Newton iterations of the same deck -- each well at its most restrictive bhp limit,
    rates from the explicit cell state.  Signs: producers deliver negative surface rates, injectors positive ones; a well whose limit equals
    its current BHP control reproduces its current rates."""
    from oracle.wells import CoupledOracleModel
    grid, tab, st, make, start = _deck(case)
    prm = capi.default_params(linear_solver_reduction=1e-12, linear_solver_maxiter=2000)
    wl_i, wl_h = make(), make()
    mi = CoupledOracleModel(grid, tab, prm, wl_i, _arrays(start(wl_i)))
    ob = OracleBackend(oracle, grid, tab, prm, wells=wl_h.arrays())
    mh = W.WellCoupledModel(ob, W.StandardWellsHost(wl_h, grid.z, tab.surface_density[0]), start(wl_h))
    mi.prepareStep(5 * decks.DAY, st); mh.prepareStep(5 * decks.DAY, st)
    for it in range(2):
        mi.nonlinearIteration(it); mh.nonlinearIteration(it, single_precision=False)
    ph = mh.computeWellPotentials()
    vals, _ = mi.perf_props(mi.st)
    pi = mi.well_potentials(vals)
    assert ph.shape == (wl_h.nw, 3)
    assert np.allclose(ph, pi, rtol=1e-7, atol=1e-12 * np.abs(pi).max()), (ph, pi)
    for w in range(wl_h.nw):
        has_bhp = any(c[0] == W.BHP for c in wl_h.controls[w])
        if has_bhp and wl_h.type[w] == W.PRODUCER:
            assert (ph[w] <= 0).all() and ph[w].min() < 0, (w, ph[w])
        if has_bhp and wl_h.type[w] == W.INJECTOR:
            assert (ph[w] >= 0).all() and ph[w].max() > 0, (w, ph[w])


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_well_potentials_with_device_wells(gpu_lib, oracle, case):
    """The same quantity with the wells on the device: DeviceWellModel.computeWellPotentials evaluates the reference's once-per-report-step
    host formula from the device's well state, connection pressures and perforated-cell properties; against oracle/wells.py on the
    oracle's own Newton path."""
    from opmgpu.model import GpuBlackoilModel
    from oracle.wells import CoupledOracleModel
    grid, tab, st, make, start = _deck(case)
    prm = capi.default_params(linear_solver_reduction=1e-11, linear_solver_maxiter=500)
    wl_d, wl_i = make(), make()
    gm = GpuBlackoilModel(grid, tab, prm)
    md = W.DeviceWellModel(gm, wl_d, start(wl_d))
    mi = CoupledOracleModel(grid, tab, prm, wl_i, _arrays(start(wl_i)))
    md.prepareStep(5 * decks.DAY, st); mi.prepareStep(5 * decks.DAY, st)
    for it in range(2):
        md.nonlinearIteration(it, single_precision=False); mi.nonlinearIteration(it)
    pd = md.computeWellPotentials(grid.z, tab.surface_density[0])
    vals, _ = mi.perf_props(mi.st)
    pi = mi.well_potentials(vals)
    assert np.allclose(pd, pi, rtol=1e-5, atol=1e-9 * np.abs(pi).max()), (pd, pi)
    gm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cpr", [0, 1])
@pytest.mark.parametrize("case", CASES)
def test_device_wells_against_the_independent_well_model(gpu_lib, oracle, case, cpr):
    """The device well model (csrc/wells.hip: factored Schur complement as a rank-7 operator, ILU0 or CPR + BiCGStab on the GPU) against
    oracle/wells.py (explicit coupled matrix, direct solve): two implementations that share neither code nor method."""
    from opmgpu.model import GpuBlackoilModel
    from oracle.wells import CoupledOracleModel
    grid, tab, st, make, start = _deck(case)
    prm = capi.default_params(linear_solver_reduction=1e-11, linear_solver_maxiter=500, cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=cpr)
    dt = 5 * decks.DAY
    wl_d, wl_i = make(), make()
    gm = GpuBlackoilModel(grid, tab, prm)
    md = W.DeviceWellModel(gm, wl_d, start(wl_d))
    mi = CoupledOracleModel(grid, tab, prm, wl_i, _arrays(start(wl_i)))
    md.prepareStep(dt, st); mi.prepareStep(dt, st)
    for it in range(12):
        cd, _ = md.nonlinearIteration(it, single_precision=False)
        ci = mi.nonlinearIteration(it)
        ws = md.pull_well_state()
        if it == 0:
            assert md.presolve_converged and md.presolve_iterations == mi.well_iterations, (md.presolve_iterations, mi.well_iterations)
        assert cd == ci, it
        assert np.array_equal(ws.current, mi.ws.current), (it, ws.current, mi.ws.current)
        assert np.allclose(md.well_flux_residual, mi.well_flux_residual, rtol=1e-5, atol=1e-12), it
        assert md.well_ctrl_residual == pytest.approx(mi.well_ctrl_residual, rel=1e-5, abs=1e-9), it
        a, b = gm.getState(), mi.st
        assert np.array_equal(a.hc, b.hc), it
        assert np.abs(a.p - b.p).max() <= 1e-6 * np.abs(b.p).max() and np.abs(a.sat - b.sat).max() <= 1e-6, it
        assert np.allclose(ws.bhp, mi.ws.bhp, rtol=1e-7), (it, ws.bhp, mi.ws.bhp)
        assert np.allclose(ws.qs, mi.ws.qs, rtol=1e-6, atol=1e-9 * np.abs(mi.ws.qs).max()), it
        if cd and it > 0:
            break
    assert cd and it < 12
    gm.close()


def _random_case(seed):
    rng = np.random.default_rng(4200 + seed)
    nx, ny, nz = int(rng.integers(4, 8)), int(rng.integers(4, 8)), int(rng.integers(2, 5))
    grid = decks.cartesian_grid(nx, ny, nz, dx=100.0, dy=100.0, dz=5.0, tops=2500.0, poro=0.25, permx_md=150.0, lognormal_sigma=float(rng.uniform(0, 0.8)), seed=4200 + seed)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, p_ref=250 * decks.BAR, z_ref=2500.0, gas_cap_fraction=0.0, gas_only_fraction=0.0, perturb=0.002, seed=4200 + seed)
    nw = int(rng.integers(1, 4))
    cols = rng.choice(nx * ny, size=nw, replace=False)
    WI = float(rng.uniform(1.0, 8.0)) * float(np.median(grid.trans))
    pv_rate = float(grid.pv.sum()) / (8000.0 * decks.DAY)
    spec, have_bhp = [], False
    for w in range(nw):
        k0 = int(rng.integers(0, nz)); k1 = int(rng.integers(k0 + 1, nz + 1))
        cells = [int(cols[w]) + nx * ny * k for k in range(k0, k1)]
        inj = bool(rng.random() < 0.5)
        bhp_ctrl = bool(rng.random() < 0.4) or (w == nw - 1 and not have_bhp)
        have_bhp = have_bhp or bhp_ctrl
        resv = bool(rng.random() < 0.4)
        sgn = 1.0 if inj else -1.0
        rate = sgn * float(rng.uniform(0.2, 1.0)) * pv_rate
        if bhp_ctrl:
            ctrl, lim = (W.BHP, float(rng.uniform(260, 300) if inj else rng.uniform(180, 240)) * decks.BAR), []
        else:
            distr = tuple(rng.uniform(0.8, 1.3, 3)) if resv else ((1.0, 0.0, 0.0) if inj else (0.0, 1.0, 0.0))
            ctrl = (W.RESERVOIR_RATE if resv else W.SURFACE_RATE, rate, distr)
            lim = [(W.BHP, float(rng.uniform(258, 330) if inj else rng.uniform(120, 246)) * decks.BAR)]       # tight enough that some cases switch
        spec.append(("W%d" % w, W.INJECTOR if inj else W.PRODUCER, grid.z[cells[0]], cells, WI, (1.0, 0.0, 0.0) if inj else (0.0, 1.0, 0.0), ctrl,
                     bool(rng.random() < 0.7), lim))

    def make():
        wl = W.Wells()
        for name, typ, zref, cells, wi, comp, ctrl, cf, lim in spec:
            wl.add_well(name, typ, zref, cells, wi, comp, ctrl, allow_cf=cf, limits=lim)
        return wl

    def start(wl):
        ws = W.WellState(wl, st.p)
        for w in range(wl.nw):
            if wl.ctrl_type[w] in (W.BHP, W.RESERVOIR_RATE):
                ws.qs[w] = (1e-5 if wl.type[w] == W.INJECTOR else -1e-5) * np.asarray(wl.comp_frac[w])
        return ws

    dt = float(rng.uniform(0.5, 5.0)) * decks.DAY
    return grid, tab, st, make, start, dt


@pytest.mark.parametrize("seed", range(8))
def test_random_wells_independent_vs_host(oracle, seed):
    """random small decks and wells (type, BHP / surface-rate / reservoir-rate control with a BHP limit, perforation range, cross-flow flag):
    three Newton iterations of the independent restatement and of the host well model on the oracle side by side"""
    from oracle.wells import CoupledOracleModel
    grid, tab, st, make, start, dt = _random_case(seed)
    prm = capi.default_params(linear_solver_reduction=1e-12, linear_solver_maxiter=2000)
    wl_i, wl_h = make(), make()
    mi = CoupledOracleModel(grid, tab, prm, wl_i, _arrays(start(wl_i)))
    ob = OracleBackend(oracle, grid, tab, prm, wells=wl_h.arrays())
    mh = W.WellCoupledModel(ob, W.StandardWellsHost(wl_h, grid.z, tab.surface_density[0]), start(wl_h))
    mi.prepareStep(dt, st); mh.prepareStep(dt, st)
    for it in range(3):
        ci = mi.nonlinearIteration(it)
        ch, _ = mh.nonlinearIteration(it, single_precision=False)
        if (np.abs(mi.ws.qs).max(1) < 1e-12).any():
            break       # a well stopped flowing: the dead-well test (wellbore rate EXACTLY zero) is decided by rounding from here on (tools/fuzz_wells_independent.py)
        assert ci == ch and np.array_equal(mi.ws.current, mh.ws.current), (it, mi.ws.current, mh.ws.current)
        a, b = mi.st, ob.getState()
        assert np.array_equal(a.hc, b.hc), it
        assert np.abs(a.p - b.p).max() <= 1e-7 * np.abs(b.p).max() and np.abs(a.sat - b.sat).max() <= 1e-7, (it, np.abs(a.p - b.p).max(), np.abs(a.sat - b.sat).max())
        assert np.allclose(mi.ws.bhp, mh.ws.bhp, rtol=1e-7) and np.allclose(mi.ws.qs, mh.ws.qs, rtol=1e-6, atol=1e-10 * max(np.abs(mh.ws.qs).max(), 1e-12)), it


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_random_wells_device_vs_independent(gpu_lib, oracle, seed):
    """the same random cases: device well model (ILU0 or CPR, by seed) against the independent restatement"""
    from opmgpu.model import GpuBlackoilModel
    from oracle.wells import CoupledOracleModel
    grid, tab, st, make, start, dt = _random_case(seed)
    prm = capi.default_params(linear_solver_reduction=1e-11, linear_solver_maxiter=800, cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=seed % 2)
    wl_d, wl_i = make(), make()
    gm = GpuBlackoilModel(grid, tab, prm)
    md = W.DeviceWellModel(gm, wl_d, start(wl_d))
    mi = CoupledOracleModel(grid, tab, prm, wl_i, _arrays(start(wl_i)))
    md.prepareStep(dt, st); mi.prepareStep(dt, st)
    for it in range(3):
        cd, _ = md.nonlinearIteration(it, single_precision=False)
        ci = mi.nonlinearIteration(it)
        ws = md.pull_well_state()
        if (np.abs(mi.ws.qs).max(1) < 1e-12).any():
            break       # dead-well knife edge, as above
        assert cd == ci and np.array_equal(ws.current, mi.ws.current), (it, ws.current, mi.ws.current)
        a, b = gm.getState(), mi.st
        assert np.array_equal(a.hc, b.hc), it
        assert np.abs(a.p - b.p).max() <= 1e-6 * np.abs(b.p).max() and np.abs(a.sat - b.sat).max() <= 1e-6, it
        assert np.allclose(ws.bhp, mi.ws.bhp, rtol=1e-6) and np.allclose(ws.qs, mi.ws.qs, rtol=1e-5, atol=1e-9 * max(np.abs(mi.ws.qs).max(), 1e-12)), it
    gm.close()
