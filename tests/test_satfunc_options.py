"""Saturation-function options of SURVEY row a5 beyond two-point ENDSCALE, on the oracle (CPU): three-point scaling (SCALECRS),
vertical scaling (KRW / KRO / KRG / PCW / PCG) and Carlson relative-permeability hysteresis (EHYSTR default model, KR only).
The reference holds known answers only for the two-point form and for hysteresis with imbibition == drainage tables
(tests/test_satfunc.cpp GwsegEPS_D, reproduced in test_oracle_golden / test_deck_ingest); these options are restated from
opm-material's published code -- "parity unpinned" -- and checked here through the properties that define them."""
import numpy as np

from opmgpu import capi, decks


def _one_cell_grid(**kw):
    g = decks.cartesian_grid(2, 1, 1)
    base = {"SWL": 0.15, "SWCR": 0.25, "SWU": 0.95, "SOWCR": 0.3, "SGL": 0.0, "SGCR": 0.05, "SGU": 0.85, "SOGCR": 0.25}
    return decks.with_endpoints(g, kw.pop("eps", base), **kw), base


def _kr(oracle, tab, grid, sw, sg, cell=0):
    s = np.array([[sw, 1.0 - sw - sg, sg]])
    kr, dkr = oracle.relperm_eps(tab, grid, s, [cell])
    return kr[0], dkr[0].reshape(3, 3)


def test_three_point_scaling_maps_the_critical_saturations(oracle):
    """SCALECRS: besides the two end points, the displacing phase's critical saturation is a fixed point of every kr curve."""
    tab = decks.satfunc_standard_tables()
    g3, e = _one_cell_grid(scalecrs=True)
    g2, _ = _one_cell_grid()
    # table (unscaled) points of satfuncStandard: Swl .1 Swcr .2 Swu .9 Sowcr .2 | Sgl 0 Sgcr .1 Sgu .9 Sogcr .2
    krw_tab = lambda sw: np.interp(sw, tab.swof_sw, tab.swof_krw)            # noqa: E731
    # krw: scaled middle point 1 - SOWCR - SGL = 0.7 <-> unscaled 1 - Sowcr - Sgl = 0.8
    assert _kr(oracle, tab, g3, 0.7, 0.0)[0][0] == pytest_approx(krw_tab(0.8))
    assert _kr(oracle, tab, g3, e["SWCR"], 0.0)[0][0] == 0.0 and _kr(oracle, tab, g3, e["SWU"], 0.0)[0][0] == pytest_approx(krw_tab(0.9))
    # between the fixed points the two forms differ, at the end points they agree
    a3, a2 = _kr(oracle, tab, g3, 0.55, 0.0)[0][0], _kr(oracle, tab, g2, 0.55, 0.0)[0][0]
    assert abs(a3 - a2) > 1e-3
    assert _kr(oracle, tab, g3, e["SWU"], 0.0)[0][0] == pytest_approx(_kr(oracle, tab, g2, e["SWU"], 0.0)[0][0])
    # the first segment has slope (u1 - u0) / (s1 - s0): d krw / d sw = table slope times that factor
    sw = 0.45
    su = 0.2 + (sw - 0.25) * (0.8 - 0.2) / (0.7 - 0.25)
    kr, dkr = _kr(oracle, tab, g3, sw, 0.0)
    assert kr[0] == pytest_approx(krw_tab(su))
    h = 1e-7
    fd = (_kr(oracle, tab, g3, sw + h, 0.0)[0][0] - _kr(oracle, tab, g3, sw - h, 0.0)[0][0]) / (2 * h)
    assert dkr[0, 0] == pytest_approx(fd, rel=1e-6)
    # krg: middle point 1 - SOGCR - SWL = 0.6 <-> 1 - Sogcr - Swl = 0.7
    krg_tab = lambda sg: np.interp(sg, tab.sgof_sg, tab.sgof_krg)            # noqa: E731
    assert _kr(oracle, tab, g3, e["SWL"], 0.6)[0][2] == pytest_approx(krg_tab(0.7))


def test_vertical_scaling_sets_the_curve_maxima(oracle):
    tab = decks.satfunc_standard_tables()
    g, e = _one_cell_grid(eps_v={"KRW": 0.35, "KRO": 0.8, "KRG": 0.6, "PCW": 1.5e5, "PCG": 3.0e5})
    g0, _ = _one_cell_grid()
    assert _kr(oracle, tab, g, e["SWU"], 0.0)[0][0] == pytest_approx(0.35)            # krw at SWU = KRW
    assert _kr(oracle, tab, g, e["SWL"], e["SGU"])[0][2] == pytest_approx(0.6)        # krg at SGU = KRG
    assert _kr(oracle, tab, g, e["SWL"], 0.0)[0][1] == pytest_approx(0.8)             # kro at connate water, no gas = KRO
    # everywhere else: the unscaled-in-value curve times the constant factor
    for sw in (0.3, 0.5, 0.7):
        assert _kr(oracle, tab, g, sw, 0.0)[0][0] == pytest_approx(_kr(oracle, tab, g0, sw, 0.0)[0][0] * 0.35 / 0.7)
    # capillary pressure through the model's phase pressures: pcow at SWL = PCW
    st = decks.State([200e5, 200e5], [[e["SWL"], 1 - e["SWL"], 0.0]] * 2, [0.0, 0.0], [0.0, 0.0], [capi.HC_OIL_ONLY] * 2)
    props = oracle.cell_props(g, tab, st)
    names = oracle.PROP_NAMES
    assert props[0, names.index("p_o"), 0] - props[0, names.index("p_w"), 0] == pytest_approx(1.5e5)


def _hyst_tables():
    """region 0 = drainage (satfuncStandard), region 1 = imbibition: larger critical gas saturation and lower krg, lower krow"""
    t = decks.satfunc_standard_tables()
    swof = [list(zip(t.swof_sw, t.swof_krw, t.swof_krow, t.swof_pcow / decks.BAR))]
    sgof = [list(zip(t.sgof_sg, t.sgof_krg, t.sgof_krog, t.sgof_pcgo / decks.BAR))]
    swof.append([(0.1, 0.0, 1.0, 0.9), (0.2, 0.0, 0.7, 0.8), (0.3, 0.1, 0.45, 0.7), (0.4, 0.2, 0.25, 0.6), (0.6, 0.4, 0.0, 0.4), (0.9, 0.7, 0.0, 0.1)])
    sgof.append([(0.0, 0.0, 1.0, 0.2), (0.3, 0.0, 0.5, 0.8), (0.5, 0.2, 0.3, 1.2), (0.8, 0.6, 0.0, 2.0), (0.9, 1.0, 0.0, 2.1)])
    return decks.FluidTables(density_wog=[[1000.0, 700.0, 1.0]], pvtw=[[1.0, 1.0, 4.0e-5, 0.96, 0.0]],
                             pvto=[[(0, [(1., 1.0, 1.2)]), (200, [(400., 1.12, .94), (500., 1.1189, .94)])]],
                             pvtg=[[(100, [(0.0001, 0.010, 0.1), (0.0, 0.0104, 0.1)]), (200, [(0.0004, 0.005, 0.2), (0.0, 0.0054, 0.2)])]],
                             swof=swof, sgof=sgof, rock=(1.0, 5.0e-5))


def test_carlson_hysteresis_scanning_curve(oracle):
    """Gas invades to Sg = 0.6 (drainage curve), then the saturation falls: krg follows the imbibition curve shifted so that the two
    meet at the turning point; the gas is trapped at Sgcr_imb + shift; increasing Sg beyond the turning point is drainage again."""
    tab = _hyst_tables()
    grid = decks.GridData(2, [[0, 1]], [1e-12], [1.0, 1.0], [0.0, 0.0], satnum=[0, 0], imbnum=[1, 1])
    nohist = decks.GridData(2, [[0, 1]], [1e-12], [1.0, 1.0], [0.0, 0.0], satnum=[0, 0])
    krg = lambda g, sg: oracle.relperm_eps(tab, g, np.array([[0.1, 0.9 - sg, sg]]), [0])[0][0][2]       # noqa: E731
    h = oracle.Hysteresis(2)
    oracle.set_hysteresis(h)
    try:
        for sg in (0.1, 0.3, 0.6):                      # no history yet: the drainage curve
            assert krg(grid, sg) == krg(nohist, sg)
        h.update(grid, tab, np.array([[0.1, 0.3, 0.6], [0.1, 0.9, 0.0]]))
        assert h.mdc_go[0] == pytest_approx(0.4) and h.mdc_go[1] == 1.0 and h.mdc_ow[0] == pytest_approx(0.7)
        kd = krg(nohist, 0.6)                           # drainage value at the turning point: (0.6 - 0.2) / 0.6 * 0.6 + 0.1 = 0.5
        sg_imb = 0.5 + (kd - 0.2) / (0.6 - 0.2) * 0.3   # where the imbibition table has that value
        assert h.d_go[0] == pytest_approx(0.6 - sg_imb)
        assert krg(grid, 0.6) == pytest_approx(kd)      # continuous at the turning point
        assert krg(grid, 0.7) == krg(nohist, 0.7)       # beyond it: drainage
        for sg in (0.5, 0.4, 0.2):                      # scanning curve: the shifted imbibition table, below the drainage curve
            want = np.interp(sg - h.d_go[0], [0.0, 0.3, 0.5, 0.8, 0.9], [0.0, 0.0, 0.2, 0.6, 1.0])
            assert krg(grid, sg) == pytest_approx(want) and krg(grid, sg) <= krg(nohist, sg) + 1e-15
        assert krg(grid, 0.3 + h.d_go[0]) == 0.0 and krg(grid, 0.3 + h.d_go[0] + 0.02) > 0.0      # trapped gas saturation
        # a second, deeper invasion moves the turning point; a shallower one does not
        h.update(grid, tab, np.array([[0.1, 0.5, 0.4], [0.1, 0.9, 0.0]]))
        assert h.mdc_go[0] == pytest_approx(0.4)
        h.update(grid, tab, np.array([[0.1, 0.2, 0.7], [0.1, 0.9, 0.0]]))
        assert h.mdc_go[0] == pytest_approx(0.3) and krg(grid, 0.7) == pytest_approx(krg(nohist, 0.7))
        # imbibition tables == drainage tables (the reference's satfuncEPS_D.DATA): no shift, no effect
        same = decks.GridData(2, [[0, 1]], [1e-12], [1.0, 1.0], [0.0, 0.0], satnum=[0, 0], imbnum=[0, 0])
        h2 = oracle.Hysteresis(2)
        oracle.set_hysteresis(h2)
        h2.update(same, tab, np.array([[0.1, 0.3, 0.6], [0.1, 0.9, 0.0]]))
        assert np.allclose(h2.d_go, 0.0, atol=1e-15) and np.allclose(h2.d_ow, 0.0, atol=1e-15)
        for sg in (0.2, 0.4, 0.6):
            assert krg(same, sg) == pytest_approx(krg(nohist, sg))
    finally:
        oracle.set_hysteresis(None)


def pytest_approx(v, rel=1e-12):
    import pytest
    return pytest.approx(v, rel=rel, abs=1e-15)
