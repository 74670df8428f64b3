"""VFP table lookup (THP well control) pinned by the reference's own known answers (tests/test_vfpproperties_legacy.cpp)."""
import json
import os

import numpy as np

from opmgpu import vfp

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vfp_known_answers.json")))


def _table(fill):
    ax = [np.asarray(G[k]) for k in ("thp_axis", "wfr_axis", "gfr_axis", "alq_axis", "flo_axis")]
    shape = tuple(a.size for a in ax)
    if fill == "random":        # fillDataRandom: 64-bit LCG, loops thp, wfr, gfr, alq, flo (flo fastest)
        n = int(np.prod(shape))
        data, x = np.zeros(n), G["lcg"]["seed"]
        for i in range(n):
            data[i] = x / float(2 ** 64 - 1)
            x = (x * G["lcg"]["mul"] + G["lcg"]["add"]) % 2 ** 64
        data = data.reshape(shape)
    else:                       # fillDataPlane: x + 2y + 3z + 4u + 5v on the unit cube
        grids = np.meshgrid(*[np.arange(s) / (s - 1.0) for s in shape], indexing="ij")
        data = sum(c * g for c, g in zip(G["plane"]["coefficients_thp_wfr_gfr_alq_flo"], grids))
    return vfp.VFPProdTable(1, 1000.0, vfp.FLO_OIL, vfp.WFR_WOR, vfp.GFR_GOR, ax[4], ax[0], ax[1], ax[2], ax[3], data)


def test_get_table_known_answer():
    """GetTable (:375-401): value and the five partial derivatives at a point that EXTRAPOLATES along gfr (GOR = 1.4)"""
    g = G["get_table"]
    t = _table("random")
    got = t.bhp(g["aqua"], g["liquid"], g["vapour"], g["thp"], g["alq"])
    want = [g[k] for k in ("value", "dthp", "dwfr", "dgfr", "dalq", "dflo")]
    assert np.allclose(got, want, rtol=g["tolerance_percent"] / 100.0, atol=0.0)


def test_conversions():
    """ConversionTests (:93-197)"""
    a, l, v = 300 + 75.0, 500 + 75.0, 700 + 75.0
    assert vfp.get_flo(a, l, v, vfp.FLO_OIL) == l and vfp.get_flo(a, l, v, vfp.FLO_LIQ) == a + l and vfp.get_flo(a, l, v, vfp.FLO_GAS) == v
    assert vfp.get_wfr(a, l, v, vfp.WFR_WOR) == a / l and vfp.get_wfr(a, l, v, vfp.WFR_WCT) == a / (a + l) and vfp.get_wfr(a, l, v, vfp.WFR_WGR) == a / v
    assert vfp.get_gfr(a, l, v, vfp.GFR_GOR) == v / l and vfp.get_gfr(a, l, v, vfp.GFR_GLR) == v / (l + a) and vfp.get_gfr(a, l, v, vfp.GFR_OGR) == l / v
    assert vfp.get_wfr(0.0, 0.0, 1.0, vfp.WFR_WOR) == 0.0          # zeroIfNanInf


def test_plane_interpolation_and_extrapolation():
    """ExtrapolatePlaneADB (:409-512) / InterpolateADBAndQs (:521-600): a linear table is reproduced exactly, inside and outside
    the axes; reference value thp + 2 wor + 3 gor + 4 alq - 5 flo (producer rates are negative)."""
    t = _table("plane")
    worst = 0.0
    for x in (0.0, 1.0, 3.0, 6.0):
        for aq in (-1.0, -4.0):
            for vap in (-2.0, -6.0):
                for u in (0.0, 2.0, 5.0):
                    for liq in (-1.0, -3.0, -6.0):
                        ref = x + 2 * (aq / liq) + 3 * (vap / liq) + 4 * u - 5 * liq
                        worst = max(worst, abs(t.bhp(aq, liq, vap, x, u)[0] - ref))
    assert worst < G["plane"]["max_d_tol"]
    nw = 5
    qs = -np.arange(3 * nw).reshape(3, nw) / (3 * nw - 1.0)
    for i in range(1, nw):
        thp = i / (nw - 1.0)
        ref = thp + 2 * qs[0, i] / qs[1, i] + 3 * qs[2, i] / qs[1, i] - 5 * qs[1, i]
        assert abs(t.bhp(qs[0, i], qs[1, i], qs[2, i], thp, 0.0)[0] - ref) < 1e-10
    # d bhp / d rates against central differences, and thp() inverts bhp()
    a, l, v, thp, alq = -0.3, -0.4, -0.2, 0.3, 0.4
    val, dq = t.bhp_dq(a, l, v, thp, alq)
    for k in range(3):
        h = 1e-6
        q1, q0 = [a, l, v], [a, l, v]
        q1[k] += h; q0[k] -= h
        fd = (t.bhp(*q1, thp, alq)[0] - t.bhp(*q0, thp, alq)[0]) / (2 * h)
        assert abs(fd - dq[k]) < 1e-7 * max(1.0, abs(fd))
    assert abs(t.thp_of(a, l, v, val, alq) - thp) < 1e-12


def test_find_thp_branches():
    thp = np.array([1.0, 2.0, 4.0])
    bhp = np.array([10.0, 20.0, 50.0])
    assert vfp.find_thp(bhp, thp, 15.0) == 1.5 and vfp.find_thp(bhp, thp, 5.0) == 0.5 and vfp.find_thp(bhp, thp, 65.0) == 5.0
    unsorted = np.array([10.0, 5.0, 50.0])
    assert vfp.find_thp(unsorted, thp, 27.5) == 3.0            # found inside the second interval
    assert vfp.find_thp(unsorted, thp, 4.0) == 1.0 + (1.0 / -5.0) * (4.0 - 10.0)
