"""EQUIL: hydrostatic initial state.  Host-side restatement of what `flow_legacy` does before the first report step when
the deck holds EQUIL instead of explicit PRESSURE / SWAT / SGAS arrays (opm/autodiff/FlowMain.hpp:638-659):

    EQUIL::DeckDependent::InitialStateComputer          opm/core/simulator/initStateEquil.hpp:233-443
      phasePressures  (one RK4 integration per phase, up and down from the datum / a contact)
                                                         opm/core/simulator/initStateEquil_impl.hpp:38-116 (RK4IVP), :118-236 (densities),
                                                         :239-469 (water / oil / gas and their order), :476-551 (vertical span)
      phaseSaturations (capillary-pressure inversion, gas-oil / oil-water overlap, pressure fix-up at the saturation limits)
                                                         initStateEquil_impl.hpp:566-741, opm/core/simulator/EquilibrationHelpers.hpp:596-795
      Rs / Rv of the cells (RSVD / RVVD / saturated value at the contact)
                                                         EquilibrationHelpers.hpp:100-390, initStateEquil_impl.hpp:744-770
    initHydroCarbonState                                 opm/core/utility/initHydroCarbonState.hpp:9-40

This runs once per simulation on the host (sequential ODE integration, a few thousand scalar PVT evaluations per region); its product
is the State the device model starts from.  The scalar root finder of the reference (opm-core RegulaFalsi, a dependency that is not in
the tree; tolerance 1e-6, EquilibrationHelpers.hpp:646-652) is replaced by an Illinois-type regula falsi run over all cells of a region
at once with the same stopping rule, so saturations agree with the reference's to that tolerance, not bit for bit.  SWATINIT is not
supported (it rescales PCW per cell, EclMaterialLawManager::applySwatinit, outside the tree).

Pins: tests/test_equil.py holds the known answers of the reference's tests/test_equil_legacy.cpp that do not need one of its (absent)
deck files: PhasePressure :188-218, CellSubset :220-304, RegMapping :309-394, CapillaryInversion :435-498 (tables rebuilt from the
vectors in the test itself).
"""
from bisect import bisect_right

import numpy as np

from . import capi
from .decks import BAR, State, _lin1d

STD_TEMP = 273.15 + 20.0          # "standard temperature for now" (initStateEquil_impl.hpp:314); the black-oil PVT ignores it


# ------------------------------------------------------------------------------------------
# scalar PVT on the host (same table semantics as the device: Tabulated1DFunction with linear extrapolation, the
# 2-D tables interpolated column-wise; see csrc/blackoil.hip `pvt_*` and the opm-material classes named there)
# ------------------------------------------------------------------------------------------
def _seg(x, xv):
    n = len(x)
    if n == 2 or xv <= x[1]:
        return 0
    if xv >= x[n - 2]:
        return n - 2
    return bisect_right(x, xv) - 1


def _lin(x, y, xv):
    i = _seg(x, xv)
    return y[i] + (y[i + 1] - y[i]) / (x[i + 1] - x[i]) * (xv - x[i])


class HostPvt:
    """b_w, b_o, b_g, Rs_sat, Rv_sat of one PVT region as plain Python floats"""

    def __init__(self, t, reg=0):
        self.disgas, self.vapoil = bool(t.has_disgas), bool(t.has_vapoil)
        self.rho_w, self.rho_o, self.rho_g = (float(v) for v in t.surface_density[reg])
        self.pvtw = [float(v) for v in t.pvtw[reg]]
        a, b = int(t.oil_node_ptr[reg]), int(t.oil_node_ptr[reg + 1])
        self.o_rs, self.o_psat, self.o_ib = list(t.oil_rs[a:b]), list(t.oil_psat[a:b]), list(t.oil_invb_sat[a:b])
        self.o_cols = [(list(t.oil_col_p[t.oil_col_ptr[k]:t.oil_col_ptr[k + 1]]), list(t.oil_col_invb[t.oil_col_ptr[k]:t.oil_col_ptr[k + 1]]))
                       for k in range(a, b)]
        a, b = int(t.gas_node_ptr[reg]), int(t.gas_node_ptr[reg + 1])
        self.g_pg, self.g_rv, self.g_ib = list(t.gas_pg[a:b]), list(t.gas_rvsat[a:b]), list(t.gas_invb_sat[a:b])
        self.g_cols = [(list(t.gas_col_rv[t.gas_col_ptr[k]:t.gas_col_ptr[k + 1]]), list(t.gas_col_invb[t.gas_col_ptr[k]:t.gas_col_ptr[k + 1]]))
                       for k in range(a, b)]

    def b_w(self, p):
        w = self.pvtw
        X = w[2] * (p - w[0])
        return (1.0 + X * (1.0 + X / 2.0)) / w[1]

    def rs_sat(self, p):
        if np.ndim(p):
            return _lin1d(np.array(self.o_psat), np.array(self.o_rs), p) if self.disgas else np.zeros(np.shape(p))
        return _lin(self.o_psat, self.o_rs, p) if self.disgas else 0.0

    def rv_sat(self, p):
        if np.ndim(p):
            return _lin1d(np.array(self.g_pg), np.array(self.g_rv), p) if self.vapoil else np.zeros(np.shape(p))
        return _lin(self.g_pg, self.g_rv, p) if self.vapoil else 0.0

    def b_o(self, p, rs, saturated):
        if saturated or not self.disgas:
            return _lin(self.o_psat, self.o_ib, p)
        i = _seg(self.o_rs, rs)
        al = (rs - self.o_rs[i]) / (self.o_rs[i + 1] - self.o_rs[i])
        return _lin(*self.o_cols[i], p) * (1.0 - al) + _lin(*self.o_cols[i + 1], p) * al

    def b_g(self, p, rv, saturated):
        if saturated or not self.vapoil:
            return _lin(self.g_pg, self.g_ib, p)
        i = _seg(self.g_pg, p)
        al = (p - self.g_pg[i]) / (self.g_pg[i + 1] - self.g_pg[i])
        return _lin(*self.g_cols[i], rv) * (1.0 - al) + _lin(*self.g_cols[i + 1], rv) * al


# ------------------------------------------------------------------------------------------
# Rs / Rv as functions of depth and pressure (EquilibrationHelpers.hpp: Miscibility::*)
# ------------------------------------------------------------------------------------------
class NoMixing:
    def __call__(self, depth, press, sat=0.0):
        return np.zeros_like(np.asarray(press, dtype=float)) if np.ndim(press) else 0.0


class _RVD:
    """RsVD / RvVD: the tabulated value at the depth (no extrapolation), capped by the saturated value; saturated where the other
    hydrocarbon phase is present (:147-157, :212-222)"""

    def __init__(self, sat_fn, depth, val, eps):
        self.sat_fn, self.depth, self.val, self.eps = sat_fn, np.asarray(depth, float), np.asarray(val, float), eps

    def __call__(self, depth, press, sat=0.0):
        if np.ndim(press) == 0:
            s = self.sat_fn(press)
            return s if abs(sat) > self.eps else min(s, float(np.interp(depth, self.depth, self.val)))
        s = self.sat_fn(np.asarray(press, float))
        return np.where(np.abs(sat) > self.eps, s, np.minimum(s, np.interp(depth, self.depth, self.val)))


class _RSatAtContact:
    """RsSatAtContact / RvSatAtContact: the saturated value at the contact pressure, capped by the local saturated value (:270-285, :330-345)"""

    def __init__(self, sat_fn, p_contact):
        self.sat_fn, self.at_contact = sat_fn, sat_fn(p_contact)

    def __call__(self, depth, press, sat=0.0):
        if np.ndim(press) == 0:
            s = self.sat_fn(press)
            return s if sat > 0.0 else min(s, self.at_contact)
        s = self.sat_fn(np.asarray(press, float))
        return np.where(np.asarray(sat) > 0.0, s, np.minimum(s, self.at_contact))


class EquilRecord:
    """one EQUIL record in SI units (opm-parser EquilRecord; items 7 / 8 <= 0: constant Rs / Rv from the contact)"""

    def __init__(self, datum, pressure, zwoc, pcow_woc, zgoc, pcgo_goc, live_oil_const_rs=True, wet_gas_const_rv=True):
        self.datum, self.pressure, self.zwoc, self.pcow_woc, self.zgoc, self.pcgo_goc = (float(v) for v in
                                                                                        (datum, pressure, zwoc, pcow_woc, zgoc, pcgo_goc))
        self.live_oil_const_rs, self.wet_gas_const_rv = bool(live_oil_const_rs), bool(wet_gas_const_rv)


class EquilReg:
    def __init__(self, rec, rs, rv, pvt):
        self.rec, self.rs, self.rv, self.pvt = rec, rs, rv, pvt


# ------------------------------------------------------------------------------------------
# phase pressures
# ------------------------------------------------------------------------------------------
class RK4IVP:
    """classical RK4 with N steps over span, Hermite dense output (initStateEquil_impl.hpp:38-116)"""

    def __init__(self, f, span, y0, N):
        self.N, self.span = N, (float(span[0]), float(span[1]))
        h = self.stepsize()
        h2, h6 = h / 2, h / 6
        y, fv = [y0], [f(self.span[0], y0)]
        for i in range(N):
            x, yi = self.span[0] + i * h, y[-1]
            k1 = fv[i]
            k2 = f(x + h2, yi + h2 * k1)
            k3 = f(x + h2, yi + h2 * k2)
            k4 = f(x + h, yi + h * k3)
            y.append(yi + h6 * (k1 + 2 * (k2 + k3) + k4))
            fv.append(f(x + h, y[-1]))
        self.y, self.f = np.array(y), np.array(fv)

    def stepsize(self):
        return (self.span[1] - self.span[0]) / self.N

    def __call__(self, x):
        x = np.asarray(x, dtype=float)
        h = self.stepsize()
        if h == 0.0:
            return np.full(x.shape, self.y[0]) if x.ndim else float(self.y[0])
        i = np.trunc((x - self.span[0]) / h).astype(np.int64)           # C++ double -> int conversion
        t = (x - (self.span[0] + i * h)) / h
        i = np.clip(i, 0, self.N - 1)                                   # "crude handling of evaluation point outside span"
        y0, y1, f0, f1 = self.y[i], self.y[i + 1], self.f[i], self.f[i + 1]
        u = (1 - 2 * t) * (y1 - y0)
        u = u + h * ((t - 1) * f0 + t * f1)
        u = u * (t * (t - 1))
        u = u + (1 - t) * y0 + t * y1
        return u if x.ndim else float(u)


def _assign(f, split, z):
    """PhasePressure::assign (:240-261): the upward integration above the split depth, the downward one from it on"""
    p = np.empty(z.shape)
    up = z < split
    if up.any():
        p[up] = f[0](z[up])
    if (~up).any():
        p[~up] = f[1](z[~up])
    return p


def phase_pressures(z, span, reg, grav, nsteps=2000):
    """EQUIL::phasePressures (:476-551) for the cells at depths `z` of one region whose cell corners span `span` = (top, bottom).
    Returns [pw, po, pg]."""
    rec, pvt = reg.rec, reg.pvt
    span = (min(span[0], rec.zgoc), max(span[1], rec.zwoc))             # contacts inside the span (:545-547)
    press = [None, None, None]
    st = {"po_woc": -1.0, "po_goc": -1.0}

    def rho_w(depth, p):
        return pvt.b_w(p) * pvt.rho_w * grav

    def rho_o(depth, p):                                                # PhasePressODE::Oil (:141-187)
        rs = reg.rs(depth, p)
        if not pvt.disgas or rs >= pvt.rs_sat(p):
            b = pvt.b_o(p, rs, True)
        else:
            b = pvt.b_o(p, rs, False)
        rho = b * pvt.rho_o
        if pvt.disgas:
            rho += rs * b * pvt.rho_g
        return rho * grav

    def rho_g(depth, p):                                                # PhasePressODE::Gas (:189-235)
        rv = reg.rv(depth, p)
        if not pvt.vapoil or rv >= pvt.rv_sat(p):
            b = pvt.b_g(p, rv, True)
        else:
            b = pvt.b_g(p, rv, False)
        rho = b * pvt.rho_g
        if pvt.vapoil:
            rho += rv * b * pvt.rho_o
        return rho * grav

    def both(f, z0, p0):
        return [RK4IVP(f, (z0, span[0]), p0, nsteps), RK4IVP(f, (z0, span[1]), p0, nsteps)]

    def water():                                                        # :263-305
        if rec.datum > rec.zwoc:
            z0, p0 = rec.datum, rec.pressure
        else:
            z0, p0 = rec.zwoc, st["po_woc"] - rec.pcow_woc
        w = both(rho_w, z0, p0)
        press[0] = _assign(w, z0, z)
        if rec.datum > rec.zwoc:
            st["po_woc"] = w[0](rec.zwoc) + rec.pcow_woc

    def oil():                                                          # :307-362
        if rec.datum > rec.zwoc:
            z0, p0 = rec.zwoc, st["po_woc"]
        elif rec.datum < rec.zgoc:
            z0, p0 = rec.zgoc, st["po_goc"]
        else:
            z0, p0 = rec.datum, rec.pressure
        o = both(rho_o, z0, p0)
        press[1] = _assign(o, z0, z)
        for key, c in (("po_woc", rec.zwoc), ("po_goc", rec.zgoc)):
            st[key] = o[0](c) if z0 > c else (o[1](c) if z0 < c else p0)

    def gas():                                                          # :364-407
        if rec.datum < rec.zgoc:
            z0, p0 = rec.datum, rec.pressure
        else:
            z0, p0 = rec.zgoc, st["po_goc"] + rec.pcgo_goc
        g = both(rho_g, z0, p0)
        press[2] = _assign(g, z0, z)
        if rec.datum < rec.zgoc:
            st["po_goc"] = g[1](rec.zgoc) - rec.pcgo_goc

    if rec.datum > rec.zwoc:                                            # equilibrateOWG (:410-469): the phase holding the datum first
        order = (water, oil, gas)
    elif rec.datum < rec.zgoc:
        order = (gas, oil, water)
    else:
        order = (oil, water, gas)
    for fn in order:
        fn()
    return press


# ------------------------------------------------------------------------------------------
# capillary pressure of the cells and its inversion
# ------------------------------------------------------------------------------------------
class CapPress:
    """pcow(Sw) and pcgo(Sg) of a set of cells, with the ENDSCALE horizontal (SWL..SWU, SGL..SGU) and vertical (PCW, PCG) scaling the
    device applies (csrc/blackoil.hip eps planes; opm-material EclEpsTwoPhaseLaw), and the scaled saturation limits
    oilWaterScaledEpsInfoDrainage(cell).{Swl,Swu,Sgl,Sgu} the equilibration reads (EquilibrationHelpers.hpp:596-640)."""

    def __init__(self, tables, grid, cells):
        t = tables
        self.n = len(cells)
        sat = np.zeros(self.n, np.int64) if grid.satnum is None else grid.satnum[cells].astype(np.int64)
        self.sat = sat
        self.tab_w, self.tab_g = {}, {}
        self.swl_t, self.swu_t, self.sgl_t, self.sgu_t = (np.zeros(self.n) for _ in range(4))
        self.vw, self.vg = np.ones(self.n), np.ones(self.n)
        for r in np.unique(sat):
            a, b = t.swof_ptr[r], t.swof_ptr[r + 1]
            c, d = t.sgof_ptr[r], t.sgof_ptr[r + 1]
            self.tab_w[r] = (t.swof_sw[a:b], t.swof_pcow[a:b])
            self.tab_g[r] = (t.sgof_sg[c:d], t.sgof_pcgo[c:d])
            m = sat == r
            self.swl_t[m], self.swu_t[m] = t.swof_sw[a], t.swof_sw[b - 1]
            self.sgl_t[m], self.sgu_t[m] = t.sgof_sg[c], t.sgof_sg[d - 1]
            ev = grid.eps_v or {}
            if "PCW" in ev and t.swof_pcow[a] != 0.0:
                self.vw[m] = ev["PCW"][cells][m] / t.swof_pcow[a]
            if "PCG" in ev and t.sgof_pcgo[d - 1] != 0.0:
                self.vg[m] = ev["PCG"][cells][m] / t.sgof_pcgo[d - 1]
        if grid.eps is not None:
            e = {k: grid.eps[i][cells] for i, k in enumerate(grid.EPS_NAMES)}
            self.swl, self.swu, self.sgl, self.sgu = e["SWL"], e["SWU"], e["SGL"], e["SGU"]
        else:
            self.swl, self.swu, self.sgl, self.sgu = self.swl_t, self.swu_t, self.sgl_t, self.sgu_t

    def _eval(self, tabs, s, s0, s1, u0, u1, vf, idx):
        idx = np.arange(self.n) if idx is None else idx
        su = u0[idx] + (s - s0[idx]) * ((u1[idx] - u0[idx]) / (s1[idx] - s0[idx]))
        out = np.empty(len(idx))
        for r, (x, y) in tabs.items():
            m = self.sat[idx] == r
            if m.any():
                out[m] = np.interp(su[m], x, y)
        return vf[idx] * out

    def pcow(self, sw, idx=None):
        return self._eval(self.tab_w, sw, self.swl, self.swu, self.swl_t, self.swu_t, self.vw, idx)

    def pcgo(self, sg, idx=None):
        return self._eval(self.tab_g, sg, self.sgl, self.sgu, self.sgl_t, self.sgu_t, self.vg, idx)


def _root(f, a, b, fa, fb, max_iter, tol):
    """roots of the element-wise function f(x, idx) bracketed by [a, b] (fa, fb of opposite sign), Illinois-type regula falsi with the
    stopping rule of the reference's solver (interval below 1e-9 + tol * max(|x0|, |x1|, 1))"""
    x0, x1, f0, f1 = a.copy(), b.copy(), fa.copy(), fb.copy()
    live = np.arange(len(a))
    for _ in range(max_iter):
        w = np.abs(x1[live] - x0[live])
        keep = (w >= 1e-9 + tol * np.maximum(np.maximum(np.abs(x0[live]), np.abs(x1[live])), 1.0)) & (f1[live] != 0.0)
        live = live[keep]
        if live.size == 0:
            break
        d = f1[live] - f0[live]
        with np.errstate(divide="ignore", invalid="ignore"):
            xn = np.where(d != 0.0, x1[live] - f1[live] * (x1[live] - x0[live]) / d, 0.5 * (x0[live] + x1[live]))
        fn = f(xn, live)
        flip = fn * f1[live] < 0.0
        x0[live] = np.where(flip, x1[live], x0[live])
        f0[live] = np.where(flip, f1[live], 0.5 * f0[live])
        x1[live], f1[live] = xn, fn
    return x1


def sat_from_pc(cp, phase, target, increasing=False):
    """EQUIL::satFromPc (EquilibrationHelpers.hpp:622-655): phase 0 water (pc = pcow, decreasing in Sw), phase 2 gas (pcgo, increasing)"""
    lo, hi = (cp.swl, cp.swu) if phase == 0 else (cp.sgl, cp.sgu)
    fn = cp.pcow if phase == 0 else cp.pcgo
    target = np.broadcast_to(np.asarray(target, float), (cp.n,))
    s0, s1 = (hi, lo) if increasing else (lo, hi)
    f0, f1 = fn(s0) - target, fn(s1) - target
    out = np.where(f0 <= 0.0, s0, s1)
    solve = np.flatnonzero((f0 > 0.0) & (f1 <= 0.0))
    if solve.size:
        a, b = np.minimum(s0, s1)[solve], np.maximum(s0, s1)[solve]
        fa = np.where(s0[solve] <= s1[solve], f0[solve], f1[solve])
        fb = np.where(s0[solve] <= s1[solve], f1[solve], f0[solve])
        out = out.copy()
        out[solve] = _root(lambda x, k: fn(x, solve[k]) - target[solve[k]], a, b, fa, fb, 60, 1e-6)
    return out


def sat_from_sum_of_pcs(cp, target, idx=None):
    """EQUIL::satFromSumOfPcs (:709-736) for water against gas: pcow(Sw) + pcgo(1 - Sw) = target"""
    idx = np.arange(cp.n) if idx is None else np.asarray(idx)
    target = np.broadcast_to(np.asarray(target, float), (len(idx),))
    f = lambda s, k: cp.pcow(s, idx[k]) + cp.pcgo(1.0 - s, idx[k]) - target[k]      # noqa: E731
    all_k = np.arange(len(idx))
    smin, smax = cp.swl[idx], cp.swu[idx]
    f0, f1 = f(smin, all_k), f(smax, all_k)
    out = np.where(f0 <= 0.0, smin, smax)
    solve = np.flatnonzero((f0 > 0.0) & (f1 <= 0.0))
    if solve.size:
        out = out.copy()
        out[solve] = _root(lambda x, k: f(x, solve[k]), smin[solve], smax[solve], f0[solve], f1[solve], 30, 1e-6)
    return out


def phase_saturations(z, reg, cp, press):
    """EQUIL::phaseSaturations (initStateEquil_impl.hpp:566-741); `press` = [pw, po, pg] is adjusted in place.  Returns [sw, so, sg]."""
    rec = reg.rec
    pw, po, pg = press
    eps = np.finfo(float).eps
    # water (:637-655)
    const_w = np.abs(cp.pcow(cp.swl) - cp.pcow(cp.swu)) < eps                       # isConstPc (:774-783)
    sw = np.where(const_w, np.where(z < rec.zwoc, cp.swl, cp.swu), sat_from_pc(cp, 0, po - pw))
    # gas (:657-671); pcog = pg - po increases with Sg
    const_g = np.abs(cp.pcgo(cp.sgl) - cp.pcgo(cp.sgu)) < eps
    sg = np.where(const_g, np.where(z < rec.zgoc, cp.sgu, cp.sgl), sat_from_pc(cp, 2, pg - po, increasing=True))
    # overlapping transition zones (:672-701): water against gas, oil pressure from the gas pressure
    over = np.flatnonzero(sg + sw > 1.0)
    if over.size:
        swo = sat_from_sum_of_pcs(cp, (pg - pw)[over], over)
        sw[over], sg[over] = swo, 1.0 - swo
        po[over] = pg[over] - cp.pcgo(sg[over], over)
    so = 1.0 - sw - sg
    # phase pressures where a saturation sits at one of its limits (:704-738)
    thr = 1.0e-6
    at_swu = sw > cp.swu - thr
    at_sgu = ~at_swu & (sg > cp.sgu - thr)
    po[at_swu] = (pw + cp.pcow(cp.swu))[at_swu]
    po[at_sgu] = (pg - cp.pcgo(cp.sgu))[at_sgu]
    at_sgl = sg < cp.sgl + thr
    pg[at_sgl] = (po + cp.pcgo(cp.sgl))[at_sgl]
    at_swl = sw < cp.swl + thr
    pw[at_swl] = (po - cp.pcow(cp.swl))[at_swl]
    return [sw, so, sg]


# ------------------------------------------------------------------------------------------
# the initial state
# ------------------------------------------------------------------------------------------
def init_hydrocarbon_state(sat, has_disgas, has_vapoil):
    """initHydroCarbonState.hpp:9-40"""
    hc = np.full(sat.shape[0], capi.HC_GAS_AND_OIL, dtype=np.int8)
    eps = np.sqrt(np.finfo(float).eps)
    free = ~(sat[:, 0] > 1.0 - eps)
    oil_only = free & (sat[:, 2] == 0.0) & bool(has_disgas)
    hc[oil_only] = capi.HC_OIL_ONLY
    hc[free & ~oil_only & (sat[:, 1] == 0.0) & bool(has_vapoil)] = capi.HC_GAS_ONLY
    return hc


def equilibrate(grid, tables, records, eqlnum=None, rsvd=None, rvvd=None, ztop=None, zbot=None, grav=None, nsteps=2000):
    """InitialStateComputer (initStateEquil.hpp:233-443) + the copy into the reservoir state (FlowMain.hpp:652-659).

    records: EquilRecord per EQLNUM region; eqlnum: 0-based region of every cell (None: one region); rsvd / rvvd: per region a
    (depth, value) pair of arrays or None; ztop / zbot: depth of the cells' top and bottom faces (their extremes over a region are the
    integration span; default: the cell-centre depths)."""
    n = grid.nc
    grav = grid.gravity if grav is None else grav
    eqlnum = np.zeros(n, np.int64) if eqlnum is None else np.asarray(eqlnum)
    ztop = grid.z if ztop is None else ztop
    zbot = grid.z if zbot is None else zbot
    p = np.zeros(n); sat = np.zeros((n, 3)); rs = np.zeros(n); rv = np.zeros(n)
    phase_p = np.zeros((n, 3))
    for r, rec in enumerate(records):
        cells = np.flatnonzero(eqlnum == r)
        if cells.size == 0:                                              # "Equilibration region has no active cells" (:403-408)
            continue
        pvt = HostPvt(tables, 0 if grid.pvtnum is None else int(grid.pvtnum[cells[0]]))     # setRegionPvtIdx (:376-385)
        rs_fn, rv_fn = NoMixing(), NoMixing()
        if tables.has_disgas:                                            # :261-297
            if not rec.live_oil_const_rs:
                if rsvd is None or rsvd[r] is None:
                    raise ValueError("Cannot initialise: RSVD table not available.")
                rs_fn = _RVD(pvt.rs_sat, rsvd[r][0], rsvd[r][1], 0.0)
            else:
                if rec.zgoc != rec.datum:
                    raise ValueError("Cannot initialise: when no explicit RSVD table is given, datum depth must be at the gas-oil-contact. "
                                     "In EQUIL region %d (counting from 1), this does not hold." % (r + 1))
                rs_fn = _RSatAtContact(pvt.rs_sat, rec.pressure)
        if tables.has_vapoil:                                            # :299-337
            if not rec.wet_gas_const_rv:
                if rvvd is None or rvvd[r] is None:
                    raise ValueError("Cannot initialise: RVVD table not available.")
                rv_fn = _RVD(pvt.rv_sat, rvvd[r][0], rvvd[r][1], 1e-16)
            else:
                if rec.zgoc != rec.datum:
                    raise ValueError("Cannot initialise: when no explicit RVVD table is given, datum depth must be at the gas-oil-contact. "
                                     "In EQUIL region %d (counting from 1), this does not hold." % (r + 1))
                rv_fn = _RSatAtContact(pvt.rv_sat, rec.pressure + rec.pcgo_goc)
        reg = EquilReg(rec, rs_fn, rv_fn, pvt)
        z = grid.z[cells]
        press = phase_pressures(z, (float(ztop[cells].min()), float(zbot[cells].max())), reg, grav, nsteps)
        cp = CapPress(tables, grid, cells)
        s = phase_saturations(z, reg, cp, press)
        p[cells] = press[1]
        phase_p[cells] = np.stack(press, 1)
        sat[cells] = np.stack(s, 1)
        rs[cells] = rs_fn(z, press[1], s[2])                             # computeRs (:754-770): oil pressure / gas saturation
        rv[cells] = rv_fn(z, press[2], s[1])                             #                     gas pressure / oil saturation
    st = State(p, sat, rs, rv, init_hydrocarbon_state(sat, tables.has_disgas, tables.has_vapoil))
    st.phase_pressure = phase_p
    return st


def from_deck(deck, grid, tables):
    """EQUIL / EQLNUM / RSVD / RVVD of a parsed deck (opmgpu/deck.py) -> State"""
    if deck.has("SWATINIT"):
        raise ValueError("SWATINIT is not supported")
    nx, ny, nz = deck.dims
    n = nx * ny * nz
    act = deck.active

    def item(r, i, default):
        return default if len(r) <= i or r[i] is None else r[i]
    recs = []
    for r in deck.records("EQUIL"):
        if not r:
            continue
        recs.append(EquilRecord(item(r, 0, 0.0), item(r, 1, 0.0) * BAR, item(r, 2, 0.0), item(r, 3, 0.0) * BAR, item(r, 4, 0.0),
                                item(r, 5, 0.0) * BAR, item(r, 6, 0) <= 0, item(r, 7, 0) <= 0))

    def vd(name):
        if not deck.has(name):
            return None
        out = []
        for r in deck.records(name):
            a = np.asarray(r, float).reshape(-1, 2)
            out.append((a[:, 0], a[:, 1]))
        return out + [None] * (len(recs) - len(out))
    eql = None if not deck.has("EQLNUM") else deck.array("EQLNUM", n)[act].astype(np.int64) - 1
    _, _, dz = deck._cell_sizes()
    half = 0.5 * dz.ravel()[act]
    return equilibrate(grid, tables, recs, eql, vd("RSVD"), vd("RVVD"), ztop=grid.z - half, zbot=grid.z + half)
