"""Pin the CPU oracle against the reference's own known answers."""
import json
import os

import numpy as np

from opmgpu import decks

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _close(v, e, reltol_percent):
    # the reference's CHECK macro (tests/test_satfunc.cpp:42-48)
    if abs(e) < 1e-14:
        return abs(v) < reltol_percent
    return abs(v - e) <= reltol_percent / 100.0 * min(abs(v), abs(e)) + 1e-300


def test_satfunc_gwseg_standard(oracle):
    g = json.load(open(os.path.join(GOLD, "satfunc_standard.json")))
    t = decks.satfunc_standard_tables()
    n = g["n"]
    s = np.zeros((n, 3))
    for i in range(n):
        s[i, 0] = i * g["sw_multiplier"]
        s[i, 1] = 1.0 - s[i, 0]
    kr, dkr = oracle.relperm(t, s)
    np_ = 3
    for i in range(n):
        tol = g["reltol_percent"]
        assert _close(kr[i, 0], g["krw"][i], tol), (i, kr[i, 0])
        assert _close(kr[i, 1], g["kro"][i], tol), (i, kr[i, 1])
        assert _close(dkr[i, 0], g["DkrwDsw"][i], tol), (i, dkr[i, 0])
        assert _close(dkr[i, 1], g["DkroDsw"][i], tol), (i, dkr[i, 1])
        assert _close(dkr[i, np_ * 2 + 1], g["DkroDsg"][i], tol), (i, dkr[i, np_ * 2 + 1])


def test_boprops_fluid_data(oracle):
    g = json.load(open(os.path.join(GOLD, "boprops_fluid_data.json")))
    t = decks.fluid_data_tables()
    assert t.surface_density[0, 0] == g["surface_density_water"]
    assert t.surface_density[0, 1] == g["surface_density_oil"]
    assert t.surface_density[0, 2] == g["surface_density_gas"]
    p = np.array(g["muwat_pressures_barsa"], dtype=float) * decks.BAR
    mu = oracle.pvt(t, "muWat", p)
    assert np.all(mu[:, 0] == mu[0, 0])          # zero pressure dependence, test_boprops_ad.cpp:132-160
    assert mu[0, 0] == 1000.0 * decks.CP
    # critical saturations: first Sg with krg > 0 is preceded by Sgcr = 0.02; Sogcr = 1 - 0.87 = 0.13
    sg, krg, krog = t.sgof_sg, t.sgof_krg, t.sgof_krog
    sgcr = sg[np.flatnonzero(krg > 0)[0] - 1]
    sogcr = 1.0 - sg[np.flatnonzero(krog == 0)[0]]
    assert abs(sgcr - g["sgcr"]) < 1e-15 and abs(sogcr - g["sogcr"]) < 1e-15


def _eps_grid(endpoints, n=10):
    """1x1xn column with per-cell scaled end points; everything not given defaults to the table's own points
    (SWOF/SGOF of satfuncStandard.DATA: Swl .1, Swcr .2, Swu .9, Sowcr .2, Sgl 0, Sgcr .1, Sgu .9, Sogcr .2)."""
    g = decks.cartesian_grid(1, 1, n)
    eps = {"SWL": 0.1, "SWCR": 0.2, "SWU": 0.9, "SOWCR": 0.2, "SGL": 0.0, "SGCR": 0.1, "SGU": 0.9, "SOGCR": 0.2}
    eps.update({k: np.asarray(v, float) for k, v in (endpoints or {}).items()})
    return decks.GridData(g.nc, g.conn_cells, g.trans, g.pv, g.z, eps=eps, dims=(1, 1, n))


def test_satfunc_endscale_known_answers(oracle):
    """GwsegEPSBase, GwsegEPS_A, GwsegEPS_C, GwsegEPS_D of the reference's tests/test_satfunc.cpp (two-point ENDSCALE;
    EPS_D = hysteresis enabled but no saturation history, i.e. the drainage curves).  GwsegEPS_B's deck is not in the
    reference tree, its arrays are transcribed but cannot be reproduced without the deck."""
    G = json.load(open(os.path.join(GOLD, "satfunc_eps.json")))["cases"]
    t = decks.satfunc_standard_tables()
    n = 11
    for name in ("GwsegEPSBase", "GwsegEPS_D", "GwsegEPS_A", "GwsegEPS_C"):
        g = G[name]
        grid = _eps_grid(g.get("endpoints"))
        ncell = 8 if isinstance(g["krw"][0], list) else 1
        tol = g["reltol_percent"]
        for icell in range(ncell):
            s = np.zeros((n, 3))
            s[:, 0] = np.arange(n) * 0.1
            s[:, 1] = 1.0 - s[:, 0]
            kr, dkr = oracle.relperm_eps(t, grid, s, np.full(n, icell))
            row = (lambda a: a[icell]) if ncell > 1 else (lambda a: a)
            for i in range(n):
                where = (name, icell, i)
                assert _close(kr[i, 0], row(g["krw"])[i], tol), where + ("krw", kr[i, 0])
                assert _close(kr[i, 1], row(g["kro"])[i], tol), where + ("kro", kr[i, 1])
                assert _close(dkr[i, 0], row(g["DkrwDsw"])[i], tol), where + ("DkrwDsw", dkr[i, 0])
                assert _close(dkr[i, 1], row(g["DkroDsw"])[i], tol), where + ("DkroDsw", dkr[i, 1])
                assert _close(dkr[i, 3 * 2 + 1], row(g["DkroDsg"])[i], tol), where + ("DkroDsg", dkr[i, 7])
