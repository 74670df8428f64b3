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
