"""RateConverter::SurfaceToReservoirVoidage (RESV controls): the oracle's restatement against the reference's own test and against the
definition; the device path against the oracle.  Reference: RateConverterLegacy.hpp:407-770, tests/test_rateconverter.cpp."""
import numpy as np
import pytest

from opmgpu import decks


def _live_tables():
    return decks.spe9_like_tables() if hasattr(decks, "spe9_like_tables") else decks.satfunc_standard_tables()


def test_three_phase_incompressible_known_answer(oracle):
    """tests/test_rateconverter.cpp, ThreePhase: fluid.data (Bw = Bo = 1, Bg = 1 .. 0.99999999), a BlackoilState as constructed (pressure,
    rs, rv all zero), region {0}: every coefficient is 1 within BOOST_CHECK_CLOSE's 1e-6 per cent"""
    from oracle.rateconverter import SurfaceToReservoirVoidage
    t = decks.fluid_data_tables()
    cv = SurfaceToReservoirVoidage(t, [0])
    cv.defineState(np.zeros(1), np.zeros(1), np.zeros(1))
    c = cv.calcCoeff(0, 0)
    assert np.all(np.abs(c - 1.0) <= 1e-8)


def test_coefficients_reproduce_the_phase_by_phase_conversion(oracle):
    """the definition the coefficients come from (RateConverterLegacy.hpp:512-546): q_w/bw + (q_o - rv q_g)/(bo detR) + (q_g - rs q_o)/(bg detR)"""
    from oracle.rateconverter import SurfaceToReservoirVoidage
    t = _live_tables()
    rng = np.random.default_rng(3)
    n = 50
    p = rng.uniform(150e5, 300e5, n)
    rs_sat = oracle.pvt(t, "rsSat", p)[:, 0]
    rv_sat = oracle.pvt(t, "rvSat", p)[:, 0]
    rs, rv = rs_sat * rng.uniform(0.2, 1.0, n), rv_sat * rng.uniform(0.2, 1.0, n)
    region = rng.integers(0, 3, n)
    cv = SurfaceToReservoirVoidage(t, region).defineState(p, rs, rv)
    for r in range(3):
        a = cv.attr[r]
        sel = region == r
        assert a["pressure"] == pytest.approx(p[sel].mean(), rel=1e-14) and a["rs"] == pytest.approx(rs[sel].mean(), rel=1e-14)
        c = cv.calcCoeff(r)
        bw = oracle.pvt(t, "bWat", [a["pressure"]])[0, 0]
        bo = oracle.pvt(t, "bOil", [a["pressure"]], r=[a["rs"]], saturated=[0])[0, 0]
        bg = oracle.pvt(t, "bGas", [a["pressure"]], r=[a["rv"]], saturated=[0])[0, 0]
        det = 1.0 - a["rs"] * a["rv"]
        q = rng.uniform(1.0, 5.0, 3)
        direct = q[0] / bw + (q[1] - a["rv"] * q[2]) / (bo * det) + (q[2] - a["rs"] * q[1]) / (bg * det)
        assert float(c @ q) == pytest.approx(direct, rel=1e-13)
        assert c[0] > 0 and c[1] > 0 and c[2] > 0


def test_rs_and_rv_start_from_the_previous_averages(oracle):
    """calcAverages clears p and T but not rs / rv (RateConverterLegacy.hpp:733-737): the second call's numerator holds the first call's average"""
    from oracle.rateconverter import SurfaceToReservoirVoidage
    cv = SurfaceToReservoirVoidage(_live_tables(), np.zeros(4, int))
    p, rs, rv = np.full(4, 2e7), np.array([10.0, 20.0, 30.0, 40.0]), np.full(4, 1e-4)
    cv.defineState(p, rs, rv)
    assert cv.attr[0]["rs"] == 25.0
    cv.defineState(p, rs, rv)
    assert cv.attr[0]["rs"] == (25.0 + 100.0) / 4 and cv.attr[0]["pressure"] == 2e7


def test_resv_control_lookup():
    """SimFIBODetails::resv_control (SimulatorBase_impl.hpp:343-357): the first RESERVOIR_RATE control, -1 without one"""
    from opmgpu.rateconverter import resv_control
    from opmgpu.wells import BHP, SURFACE_RATE, RESERVOIR_RATE, _ctrl
    assert resv_control([_ctrl((BHP, 1e7))]) == -1
    assert resv_control([_ctrl((SURFACE_RATE, -1.0, (0, 1, 0))), _ctrl((RESERVOIR_RATE, -2.0, (1, 1, 1))), _ctrl((BHP, 1e7))]) == 1


# ---------------------------------------------------------------- device path
def _random_state(g, t, oracle, seed):
    st = decks.initial_state(g, t, perturb=0.01, seed=seed)
    return st


@pytest.mark.gpu
def test_device_rate_converter_matches_the_oracle(gpu_lib, oracle):
    from opmgpu import capi
    from opmgpu.model import GpuBlackoilModel
    from opmgpu.rateconverter import SurfaceToReservoirVoidage as Dev
    from oracle.rateconverter import SurfaceToReservoirVoidage as Ora
    t = _live_tables()
    g = decks.cartesian_grid(9, 8, 7, lognormal_sigma=0.5, seed=4)
    st = _random_state(g, t, oracle, 4)
    m = GpuBlackoilModel(g, t, capi.default_params())
    m.prepareStep(86400.0, st)
    rng = np.random.default_rng(5)
    region = rng.integers(1, 5, g.nc) * 3            # arbitrary ids, like FIPNUM
    dev, ora = Dev(m, region), Ora(t, region)
    for _ in range(2):                               # the second round exercises the rs / rv carry-over
        dev.defineState(); ora.defineState(st.p, st.rs, st.rv)
        for r in np.unique(region):
            for k in ("pressure", "rs", "rv"):
                assert dev.attr[int(r)][k] == pytest.approx(ora.attr[int(r)][k], rel=1e-14, abs=1e-300)
            cd, co = dev.calcCoeff(r), ora.calcCoeff(r)
            assert np.allclose(cd, co, rtol=1e-13, atol=0.0), (r, cd, co)
    one_d, one_o = Dev(m).defineState(), Ora(t, np.zeros(g.nc, int)).defineState(st.p, st.rs, st.rv)
    assert np.allclose(one_d.calcCoeff(0), one_o.calcCoeff(0), rtol=1e-13, atol=0.0)
    m.close()


@pytest.mark.gpu
def test_device_three_phase_incompressible_known_answer(gpu_lib):
    """the reference's own test case through the library: coefficients 1"""
    from opmgpu import capi
    from opmgpu.model import GpuBlackoilModel
    t = decks.fluid_data_tables()
    g = decks.cartesian_grid(2, 2, 2)
    m = GpuBlackoilModel(g, t, capi.default_params())
    out = np.zeros(3)
    z = capi.f64([0.0])
    m._chk(m.lib.opmgpu_voidage_coefficients(m.ctx, 1, capi.dptr(z), capi.dptr(z), capi.dptr(z), None, capi.dptr(out)))
    assert np.all(np.abs(out - 1.0) <= 1e-8)
    m.close()


@pytest.mark.gpu
def test_resv_controlled_producer(gpu_lib, oracle):
    """A producer on RESERVOIR_RATE control (computeRESV gives the control its coefficients) next to a BHP injector: the device well model
    on the device reservoir against the host well model on the oracle, Newton iteration by Newton iteration; at convergence the well's
    control equation sum_p distr[p] q_s[p] = target holds, i.e. its reservoir-volume rate is the target."""
    from opmgpu import capi, wells as W
    from opmgpu.model import GpuBlackoilModel
    from opmgpu.rateconverter import SurfaceToReservoirVoidage as Dev, computeRESV, resv_control
    from oracle.rateconverter import SurfaceToReservoirVoidage as Ora
    from util import OracleBackend
    nx, ny, nz = 10, 10, 3
    grid = decks.cartesian_grid(nx, ny, nz, dx=300.0, dy=300.0, dz=10.0, tops=2500.0, poro=0.3, permx_md=200.0, lognormal_sigma=0.3)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, p_ref=250 * decks.BAR, z_ref=2500.0, gas_cap_fraction=0.0, gas_only_fraction=0.0)
    col = lambda i, j: [i + nx * j + nx * ny * k for k in range(nz)]
    target = -400.0 / 86400.0                                    # reservoir m3/s; negative: production

    def make_wells():
        wl = W.Wells()
        WI = 5.0 * float(np.median(grid.trans))
        wl.add_well("INJ", W.INJECTOR, grid.z[col(0, 0)[0]], col(0, 0), WI, (1.0, 0.0, 0.0), (W.BHP, 300 * decks.BAR))
        wl.add_well("PROD", W.PRODUCER, grid.z[col(nx - 1, ny - 1)[0]], col(nx - 1, ny - 1)[:2], WI, (0.0, 1.0, 0.0),
                    (W.RESERVOIR_RATE, target, (1.0, 1.0, 1.0)), limits=[(W.BHP, 100 * decks.BAR)])
        return wl

    prm = capi.default_params(linear_solver_reduction=1e-11, linear_solver_maxiter=500)
    dt = 5 * decks.DAY
    wd, wo = make_wells(), make_wells()
    gm = GpuBlackoilModel(grid, tab, prm)
    ob = OracleBackend(oracle, grid, tab, prm, wells=wo.arrays())
    def start_state(wl):
        # both wells start with a small rate in their flowing direction (neither control seeds the rates): from q_s = 0 the reference's
        # dead-well test (wellbore rate EXACTLY zero, StandardWells_impl.hpp:486-506) sits on a knife edge, on the device and on the host alike
        w0 = W.WellState(wl, st.p)
        w0.qs[0] = 1e-5 * np.asarray(wl.comp_frac[0])
        w0.qs[1] = -1e-5 * np.asarray(wl.comp_frac[1])
        return w0

    md = W.DeviceWellModel(gm, wd, start_state(wd))
    mo = W.WellCoupledModel(ob, W.StandardWellsHost(wo, grid.z, tab.surface_density[0]), start_state(wo))
    md.prepareStep(dt, st); mo.prepareStep(dt, st)
    # computeRESV: device side through the library, oracle side through the restatement
    assert computeRESV(Dev(gm), wd, device_wells=md) == [1]
    co = Ora(tab, np.zeros(grid.nc, int)).defineState(st.p, st.rs, st.rv).calcCoeff(0, 0)
    rc = resv_control(wo.controls[1])
    c = wo.controls[1][rc]
    wo.controls[1][rc] = (c[0], c[1], co) + tuple(c[3:])
    distr = wd.controls[1][0][2]
    assert np.allclose(distr, co, rtol=1e-13) and np.all(distr != 1.0) and np.all(distr > 0)
    it = 0
    while True:
        cd, _ = md.nonlinearIteration(it, single_precision=False)
        c_o, _ = mo.nonlinearIteration(it, single_precision=False)
        ws = md.pull_well_state()
        assert cd == c_o, it
        assert np.array_equal(ws.current, mo.ws.current), (it, ws.current, mo.ws.current)
        assert np.allclose(ws.bhp, mo.ws.bhp, rtol=1e-7), (it, ws.bhp, mo.ws.bhp)
        assert np.allclose(ws.qs, mo.ws.qs, rtol=1e-6, atol=1e-9 * np.abs(mo.ws.qs).max()), it
        a, b = gm.getState(), ob.getState()
        assert np.abs(a.p - b.p).max() <= 1e-6 * np.abs(b.p).max() and np.abs(a.sat - b.sat).max() <= 1e-6, it
        it += 1
        if (cd and it > 1) or it > 12:
            break
    assert cd and it <= 12
    assert ws.current[1] == 0                                    # stays on the RESV control
    assert float(distr @ ws.qs[1]) == pytest.approx(target, rel=1e-7)
    gm.close()
