"""Synthetic decks for the Newton-step hot path.

The reference gets grid topology, transmissibilities, pore volumes, fluid tables and the initial
state from external packages (opm-grid, opm-common's parser, opm-material).
None of them exists offline, so the harness synthesises the same *data*: a Cartesian generator
with opm-grid's face ordering (all x-normal faces, then y, then z; cells i-fastest), two-point
harmonic transmissibilities (GeoProps.hpp:121-159), and the PROPS section of the reference's own
test deck tests/satfuncStandard.DATA converted METRIC -> SI and pre-processed the way
opm-material stores it (LiveOilPvt master-table extension, 1/B and 1/(B mu) tables).
"""
import ctypes as C

import numpy as np

from . import capi

BAR = 1.0e5
CP = 1.0e-3
MD = 9.869233e-16
DAY = 86400.0
GRAVITY = 9.80665


# ------------------------------------------------------------------------------------------
# fluid tables
# ------------------------------------------------------------------------------------------
class FluidTables:
    """Flat table arrays matching `opmgpu_tables` (include/opmgpu.h)."""

    def __init__(self, density_wog, pvtw, pvto, pvtg, swof, sgof, rock, disgas=True, vapoil=True, vappars=(0.0, 0.0), rocktab=None):
        """All inputs in deck units (METRIC): lists per region.

        pvto: per region, list of (rs, [(p, Bo, muo), ...]) saturated rows with undersaturated
        branches; pvtg: per region, list of (pg, [(rv, Bg, mug), ...]) (first = saturated).
        pvdo / pvdg style dead tables: pass rows with a single column entry and disgas/vapoil False.
        vappars = (vap1, vap2) of VAPPARS; rocktab = rows (p [bar], pv_mult, trans_mult) of ROCKTAB (replaces `rock`).
        """
        self.n_pvt = len(pvtw)
        self.n_sat = len(swof)
        self.has_disgas, self.has_vapoil = int(disgas), int(vapoil)
        self.surface_density = capi.f64(density_wog).reshape(self.n_pvt, 3)
        w = capi.f64(pvtw).reshape(self.n_pvt, 5).copy()
        w[:, 0] *= BAR; w[:, 2] /= BAR; w[:, 3] *= CP; w[:, 4] /= BAR
        self.pvtw = w
        self._build_oil(pvto)
        self._build_gas(pvtg)
        self._build_sat(swof, sgof)
        self.rock_pref, self.rock_comp = rock[0] * BAR, rock[1] / BAR
        self.vap1, self.vap2 = float(vappars[0]), float(vappars[1])
        self.rocktab_n = 0
        if rocktab is not None:
            rt = capi.f64(rocktab).reshape(-1, 3)
            self.rocktab_n = rt.shape[0]
            self.rocktab_p, self.rocktab_pvmult, self.rocktab_transmult = capi.f64(rt[:, 0] * BAR), capi.f64(rt[:, 1]), capi.f64(rt[:, 2])
        self._struct = None

    # opm-material LiveOilPvt::initFromDeck + extendPvtoTable_ (restated): rows without
    # undersaturated data are extended with the compressibility / "viscosibility" of the next
    # row that has it (the master table).
    def _build_oil(self, pvto):
        node_ptr, col_ptr = [0], [0]
        rs_l, psat_l, ib_l, ibm_l, cp_l, cib_l, cibm_l = [], [], [], [], [], [], []
        for rows in pvto:
            rows = [(rs, list(col)) for rs, col in rows]
            if self.has_disgas:
                for i, (rs, col) in enumerate(rows):
                    if len(col) > 1:
                        continue
                    master = next((c for _, c in rows[i + 1:] if len(c) > 1), None)
                    if master is None:
                        master = next((c for _, c in reversed(rows[:i]) if len(c) > 1), None)
                    if master is None:
                        raise ValueError("PVTO needs at least one undersaturated branch")
                    p, B, mu = [col[0][0]], [col[0][1]], [col[0][2]]
                    for k in range(1, len(master)):
                        dpo = master[k][0] - master[k - 1][0]
                        B1, B2 = master[k][1], master[k - 1][1]
                        x = (B1 - B2) / ((B1 + B2) / 2.0)
                        m1, m2 = master[k][2], master[k - 1][2]
                        xm = (m1 - m2) / ((m1 + m2) / 2.0)
                        p.append(p[-1] + dpo)
                        B.append(B[-1] * (1.0 + x / 2.0) / (1.0 - x / 2.0))
                        mu.append(mu[-1] * (1.0 + xm / 2.0) / (1.0 - xm / 2.0))
                    rows[i] = (rs, list(zip(p, B, mu)))
            for rs, col in rows:
                p0, B0, mu0 = col[0]
                rs_l.append(rs); psat_l.append(p0 * BAR); ib_l.append(1.0 / B0); ibm_l.append(1.0 / (B0 * mu0 * CP))
                for p, B, mu in col:
                    cp_l.append(p * BAR); cib_l.append(1.0 / B); cibm_l.append(1.0 / (B * mu * CP))
                col_ptr.append(len(cp_l))
            node_ptr.append(len(rs_l))
        self.oil_node_ptr = capi.i32(node_ptr); self.oil_col_ptr = capi.i32(col_ptr)
        self.oil_rs, self.oil_psat = capi.f64(rs_l), capi.f64(psat_l)
        self.oil_invb_sat, self.oil_invbmu_sat = capi.f64(ib_l), capi.f64(ibm_l)
        self.oil_col_p, self.oil_col_invb, self.oil_col_invbmu = capi.f64(cp_l), capi.f64(cib_l), capi.f64(cibm_l)

    # opm-material WetGasPvt::initFromDeck (restated): nodes keyed by pg, columns over Rv ascending.
    def _build_gas(self, pvtg):
        node_ptr, col_ptr = [0], [0]
        pg_l, rv_l, ib_l, ibm_l, crv_l, cib_l, cibm_l = [], [], [], [], [], [], []
        for rows in pvtg:
            for pg, col in rows:
                rv0, B0, mu0 = col[0]
                pg_l.append(pg * BAR); rv_l.append(rv0); ib_l.append(1.0 / B0); ibm_l.append(1.0 / (B0 * mu0 * CP))
                col = sorted(col, key=lambda e: e[0])
                if len(col) == 1:        # dry gas: a flat column so the 2-D evaluator stays defined
                    col = [col[0], (col[0][0] + 1.0, col[0][1], col[0][2])]
                for rv, B, mu in col:
                    crv_l.append(rv); cib_l.append(1.0 / B); cibm_l.append(1.0 / (B * mu * CP))
                col_ptr.append(len(crv_l))
            node_ptr.append(len(pg_l))
        self.gas_node_ptr = capi.i32(node_ptr); self.gas_col_ptr = capi.i32(col_ptr)
        self.gas_pg, self.gas_rvsat = capi.f64(pg_l), capi.f64(rv_l)
        self.gas_invb_sat, self.gas_invbmu_sat = capi.f64(ib_l), capi.f64(ibm_l)
        self.gas_col_rv, self.gas_col_invb, self.gas_col_invbmu = capi.f64(crv_l), capi.f64(cib_l), capi.f64(cibm_l)

    def _build_sat(self, swof, sgof):
        wp, gp = [0], [0]
        w, g = [], []
        for tab in swof:
            w.extend(tab); wp.append(len(w))
        for tab in sgof:
            g.extend(tab); gp.append(len(g))
        w = capi.f64(w).reshape(-1, 4); g = capi.f64(g).reshape(-1, 4)
        self.swof_ptr, self.sgof_ptr = capi.i32(wp), capi.i32(gp)
        self.swof_sw, self.swof_krw, self.swof_krow = capi.f64(w[:, 0]), capi.f64(w[:, 1]), capi.f64(w[:, 2])
        self.swof_pcow = capi.f64(w[:, 3] * BAR)
        self.sgof_sg, self.sgof_krg, self.sgof_krog = capi.f64(g[:, 0]), capi.f64(g[:, 1]), capi.f64(g[:, 2])
        self.sgof_pcgo = capi.f64(g[:, 3] * BAR)

    def struct(self):
        if self._struct is None:
            t = capi.Tables()
            t.n_pvt_regions, t.n_sat_regions = self.n_pvt, self.n_sat
            t.has_disgas, t.has_vapoil = self.has_disgas, self.has_vapoil
            for name, _ in capi.Tables._fields_:
                if hasattr(self, name) and isinstance(getattr(self, name), np.ndarray):
                    a = getattr(self, name)
                    setattr(t, name, capi.iptr(a) if a.dtype == np.int32 else capi.dptr(a))
            t.rock_pref, t.rock_comp = self.rock_pref, self.rock_comp
            t.vap1, t.vap2, t.rocktab_n = self.vap1, self.vap2, self.rocktab_n
            self._struct = t
        return self._struct


def satfunc_standard_tables(pc_scale=1.0, **extra):
    """PROPS of the reference's tests/satfuncStandard.DATA (METRIC); extra = vappars= / rocktab= passed on to FluidTables."""
    pvto = [[(0, [(1., 1.0000, 1.20)]), (20, [(40., 1.0120, 1.17)]), (40, [(80., 1.0255, 1.14)]),
             (60, [(120., 1.0380, 1.11)]), (80, [(160., 1.0510, 1.08)]), (100, [(200., 1.0630, 1.06)]),
             (120, [(240., 1.0750, 1.03)]), (140, [(280., 1.0870, 1.00)]), (160, [(320., 1.0985, .98)]),
             (180, [(360., 1.1100, .95)]), (200, [(400., 1.1200, .94), (500., 1.1189, .94)])]]
    pvtg = [[(100, [(0.0001, 0.010, 0.1), (0.0, 0.0104, 0.1)]),
             (200, [(0.0004, 0.005, 0.2), (0.0, 0.0054, 0.2)])]]
    swof = [[(0.1, 0.0, 1.0, 0.9), (0.2, 0.0, 0.8, 0.8), (0.3, 0.1, 0.6, 0.7), (0.4, 0.2, 0.4, 0.6),
             (0.7, 0.5, 0.1, 0.3), (0.8, 0.6, 0.0, 0.2), (0.9, 0.7, 0.0, 0.1)]]
    sgof = [[(0.0, 0.0, 1.0, 0.2), (0.1, 0.0, 0.7, 0.4), (0.2, 0.1, 0.6, 0.6), (0.8, 0.7, 0.0, 2.0),
             (0.9, 1.0, 0.0, 2.1)]]
    swof = [[(a, b, c, d * pc_scale) for a, b, c, d in swof[0]]]
    sgof = [[(a, b, c, d * pc_scale) for a, b, c, d in sgof[0]]]
    # DENSITY 700 1000 1 is (oil, water, gas); ours is (water, oil, gas)
    return FluidTables(density_wog=[[1000.0, 700.0, 1.0]], pvtw=[[1.0, 1.0, 4.0e-5, 0.96, 0.0]],
                       pvto=pvto, pvtg=pvtg, swof=swof, sgof=sgof, rock=(1.0, 5.0e-5), **extra)


def fluid_data_tables():
    """PROPS of the reference's tests/fluid.data (PVCDO dead oil, PVDG dry gas), METRIC."""
    pvto = [[(0, [(1., 1.0, 1000.0)]), (0, [(801., 1.0, 1000.0)])]]       # PVCDO with Co = 0: flat
    pvtg = [[(1, [(0.0, 1.0, 1.0)]), (800, [(0.0, 0.99999999, 1.0)])]]
    swof = [[(0.12, 0, 1, 0), (0.15, 0, 1, 0), (0.17, 0.01, 1, 0), (0.2, 0.5, 0.5, 0), (0.96, 0.9, 0.1, 0),
             (0.98, 0.95, 0, 0), (1, 1, 0, 0)]]
    sgof = [[(0, 0, 1.0, 0), (0.02, 0, 1.0, 0), (0.05, 0.1, 0.9, 0), (0.86, 0.9, 0.1, 0), (0.87, 0.95, 0, 0),
             (0.88, 1.0, 0, 0)]]
    return FluidTables(density_wog=[[1000.0, 800.0, 1.0]], pvtw=[[1.0, 1.0, 0.0, 1000.0, 0.0]],
                       pvto=pvto, pvtg=pvtg, swof=swof, sgof=sgof, rock=(1.0, 0.0), disgas=False, vapoil=False)


# ------------------------------------------------------------------------------------------
# grids
# ------------------------------------------------------------------------------------------
class GridData:
    """Static grid data matching `opmgpu_grid`."""

    EPS_NAMES = ("SWL", "SWCR", "SWU", "SOWCR", "SGL", "SGCR", "SGU", "SOGCR")
    EPSV_NAMES = ("KRW", "KRO", "KRG", "PCW", "PCG")

    def __init__(self, nc, conn_cells, trans, pv, z, gravity=GRAVITY, thpres=None, pvtnum=None, satnum=None, dims=None, eps=None,
                 scalecrs=False, eps_v=None, imbnum=None, ieps=None):
        self.nc = int(nc)
        self.conn_cells = capi.i32(conn_cells).reshape(-1, 2)
        self.nconn = self.conn_cells.shape[0]
        self.trans, self.pv, self.z = capi.f64(trans), capi.f64(pv), capi.f64(z)
        self.gravity = float(gravity)
        self.thpres = None if thpres is None else capi.f64(thpres)
        self.pvtnum = None if pvtnum is None else capi.i32(pvtnum)
        self.satnum = None if satnum is None else capi.i32(satnum)
        self.dims = dims
        # ENDSCALE: dict name -> per-cell array for the eight scaled end points (all or none)
        self.eps = None if eps is None else [capi.f64(np.broadcast_to(eps[k], (self.nc,))) for k in self.EPS_NAMES]
        # SCALECRS (three-point kr scaling), vertical scaling maxima (dict KRW / KRO / KRG / PCW / PCG -> per-cell array, any subset;
        # PCW / PCG in Pa), hysteresis: IMBNUM regions (0-based) and optionally the imbibition curves' scaled end points (ISWL ... as SWL ...)
        self.scalecrs = bool(scalecrs)
        self.eps_v = None if eps_v is None else {k: capi.f64(np.broadcast_to(v, (self.nc,))) for k, v in eps_v.items()}
        self.imbnum = None if imbnum is None else capi.i32(np.broadcast_to(imbnum, (self.nc,)))
        self.ieps = None if ieps is None else [capi.f64(np.broadcast_to(ieps[k], (self.nc,))) for k in self.EPS_NAMES]
        self._struct = None

    def struct(self):
        if self._struct is None:
            g = capi.Grid()
            g.nc, g.nconn = self.nc, self.nconn
            g.conn_cells, g.trans, g.pv, g.z = capi.iptr(self.conn_cells), capi.dptr(self.trans), capi.dptr(self.pv), capi.dptr(self.z)
            g.gravity = self.gravity
            g.thpres, g.pvtnum, g.satnum = capi.dptr(self.thpres), capi.iptr(self.pvtnum), capi.iptr(self.satnum)
            for k in range(8):
                g.eps[k] = capi.dptr(None if self.eps is None else self.eps[k])
                g.ieps[k] = capi.dptr(None if self.ieps is None else self.ieps[k])
            g.scalecrs = int(self.scalecrs)
            for k, name in enumerate(self.EPSV_NAMES):
                g.eps_v[k] = capi.dptr(None if self.eps_v is None else self.eps_v.get(name))
            g.imbnum = capi.iptr(self.imbnum)
            self._struct = g
        return self._struct


def with_endpoints(grid, eps, **more):
    """Copy of `grid` carrying ENDSCALE end points (dict name -> scalar / per-cell array, all eight names); more = scalecrs=True,
    eps_v={KRW..}, imbnum=..., ieps={...} (see GridData)."""
    g = GridData(grid.nc, grid.conn_cells, grid.trans, grid.pv, grid.z, gravity=grid.gravity, thpres=grid.thpres,
                 pvtnum=grid.pvtnum, satnum=grid.satnum, dims=grid.dims, eps=eps, **more)
    if hasattr(grid, "active_index"):
        g.active_index = grid.active_index
    return g


def random_endpoints(grid, seed=0, base=None, jitter=0.06):
    """Synthetic per-cell scaled end points scattered around `base` (default: the satfuncStandard tables' own points)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    base = base or {"SWL": 0.1, "SWCR": 0.2, "SWU": 0.9, "SOWCR": 0.2, "SGL": 0.0, "SGCR": 0.1, "SGU": 0.9, "SOGCR": 0.2}
    eps = {k: np.clip(v + jitter * (rng.random(grid.nc) - 0.5), 0.0, 1.0) for k, v in base.items()}
    eps["SGL"] = np.zeros(grid.nc)
    eps["SWCR"] = np.maximum(eps["SWCR"], eps["SWL"])
    return eps


def cartesian_grid(nx, ny, nz, dx=10.0, dy=10.0, dz=2.0, tops=2000.0, poro=0.2, permx_md=100.0, permz_ratio=0.1,
                   lognormal_sigma=0.0, seed=12345, actnum=None, nnc_fraction=0.0, thpres=None):
    """Cartesian corner-point-free generator; cells i-fastest, faces x- then y- then z-normal."""
    n = nx * ny * nz
    rng = np.random.Generator(np.random.PCG64(seed))
    kx = np.full(n, permx_md * MD)
    if lognormal_sigma > 0:
        kx = kx * np.exp(lognormal_sigma * rng.standard_normal(n))
    ky, kz = kx, permz_ratio * kx
    idx = np.arange(n).reshape(nz, ny, nx)
    vol = dx * dy * dz
    k_of = np.arange(n) // (nx * ny)
    zc = tops + (k_of + 0.5) * dz

    def faces(a, b, kperm, area, d):
        a, b = a.ravel(), b.ravel()
        h1 = kperm[a] * area / (d / 2.0); h2 = kperm[b] * area / (d / 2.0)
        return np.stack([a, b], 1), 1.0 / (1.0 / h1 + 1.0 / h2)

    cx, tx = faces(idx[:, :, :-1], idx[:, :, 1:], kx, dy * dz, dx)
    cy, ty = faces(idx[:, :-1, :], idx[:, 1:, :], ky, dx * dz, dy)
    cz, tz = faces(idx[:-1, :, :], idx[1:, :, :], kz, dx * dy, dz)
    conn = np.concatenate([cx, cy, cz]); trans = np.concatenate([tx, ty, tz])
    pv = np.full(n, poro * vol)
    if actnum is not None:
        act = np.asarray(actnum, dtype=bool).ravel()
        newid = -np.ones(n, dtype=np.int64); newid[act] = np.arange(act.sum())
        keep = act[conn[:, 0]] & act[conn[:, 1]]
        conn = newid[conn[keep]]; trans = trans[keep]
        pv, zc = pv[act], zc[act]
        n = int(act.sum())
    if nnc_fraction > 0:       # fault-style non-neighbour connections appended after the faces
        m = int(nnc_fraction * conn.shape[0])
        a = rng.integers(0, n, m); b = rng.integers(0, n, m)
        ok = a != b
        a, b = a[ok], b[ok]
        pairs = {(min(i, j), max(i, j)) for i, j in conn.tolist()}
        nn = [(i, j) for i, j in zip(a.tolist(), b.tolist()) if (min(i, j), max(i, j)) not in pairs]
        nn = list(dict.fromkeys((min(i, j), max(i, j)) for i, j in nn))
        if nn:
            conn = np.concatenate([conn, np.asarray(nn)]); trans = np.concatenate([trans, np.full(len(nn), np.median(trans) * 0.1)])
    th = None
    if thpres is not None:
        th = np.full(conn.shape[0], float(thpres))
    g = GridData(n, conn, trans, pv, zc, thpres=th, dims=(nx, ny, nz))
    # Cartesian (global) index -> active cell index, -1 for inactive cells (opm-grid's global_cell, inverted)
    g.active_index = np.arange(nx * ny * nz) if actnum is None else newid
    return g


# ------------------------------------------------------------------------------------------
# state
# ------------------------------------------------------------------------------------------
class State:
    """BlackoilState layout (opm/core/simulator/BlackoilState.hpp:40-90)."""

    def __init__(self, p, sat, rs, rv, hc):
        self.p, self.sat = capi.f64(p).copy(), capi.f64(sat).reshape(-1, 3).copy()
        self.rs, self.rv = capi.f64(rs).copy(), capi.f64(rv).copy()
        self.hc = np.ascontiguousarray(hc, dtype=np.int8).copy()

    def copy(self):
        return State(self.p, self.sat, self.rs, self.rv, self.hc)


def _lin1d(x, y, xv):
    """Tabulated1DFunction semantics (linear extrapolation)."""
    i = np.clip(np.searchsorted(x, xv, side="right") - 1, 0, len(x) - 2)
    return y[i] + (y[i + 1] - y[i]) / (x[i + 1] - x[i]) * (xv - x[i])


def initial_state(grid, tables, p_ref=200.0 * BAR, z_ref=2000.0, sw=0.25, gas_cap_fraction=0.1, gas_only_fraction=0.01,
                  perturb=0.0, seed=12345):
    """Hydrostatic-ish initial state in which all three HydroCarbonStates occur
    (initHydroCarbonState.hpp:9-40 decides the enum from the saturations)."""
    n = grid.nc
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    rho_o = 700.0
    p = p_ref + rho_o * grid.gravity * (grid.z - z_ref)
    if perturb > 0:
        p = p * (1.0 + perturb * rng.standard_normal(n))
    zmin, zmax = grid.z.min(), grid.z.max()
    goc = zmin + gas_cap_fraction * (zmax - zmin + 1e-12)
    sat = np.zeros((n, 3)); sat[:, 0] = sw
    gas_cap = grid.z < goc
    sat[gas_cap, 2] = 0.3
    sat[:, 1] = 1.0 - sat[:, 0] - sat[:, 2]
    t = tables
    a, b = t.oil_node_ptr[0], t.oil_node_ptr[1]
    rs_sat = _lin1d(t.oil_psat[a:b], t.oil_rs[a:b], p) if t.has_disgas else np.zeros(n)
    a, b = t.gas_node_ptr[0], t.gas_node_ptr[1]
    rv_sat = _lin1d(t.gas_pg[a:b], t.gas_rvsat[a:b], p) if t.has_vapoil else np.zeros(n)
    rs = np.where(gas_cap, rs_sat, 0.9 * rs_sat)
    rv = rv_sat.copy()
    hc = np.where(gas_cap, capi.HC_GAS_AND_OIL, capi.HC_OIL_ONLY).astype(np.int8)
    if not t.has_disgas:
        hc[:] = capi.HC_GAS_AND_OIL
    if t.has_vapoil and gas_only_fraction > 0:
        cand = np.flatnonzero(gas_cap)
        pick = cand[rng.random(cand.size) < gas_only_fraction / max(gas_cap_fraction, 1e-9)]
        sat[pick, 2] = 1.0 - sat[pick, 0]; sat[pick, 1] = 0.0
        rv[pick] = 0.5 * rv_sat[pick]
        hc[pick] = capi.HC_GAS_ONLY
    return State(p, sat, rs, rv, hc)


def random_state(grid, tables, seed=7, breakpoints=True):
    """Adversarial state for parity tests: random pressures/saturations/ratios, all three states,
    some cells exactly on table breakpoints and saturation bounds."""
    n = grid.nc
    rng = np.random.Generator(np.random.PCG64(seed))
    p = (60.0 + 380.0 * rng.random(n)) * BAR
    sw = 0.05 + 0.9 * rng.random(n)
    sg = (1.0 - sw) * rng.random(n)
    hc = rng.integers(0, 3, n).astype(np.int8)
    if not tables.has_disgas:
        hc[hc == capi.HC_OIL_ONLY] = capi.HC_GAS_AND_OIL
    if not tables.has_vapoil:
        hc[hc == capi.HC_GAS_ONLY] = capi.HC_GAS_AND_OIL
    bp = np.array([0.1, 0.2, 0.3, 0.4, 0.7, 0.8, 0.9])
    k = (rng.random(n) < 0.1) & bool(breakpoints)
    sw[k] = rng.choice(bp, k.sum())
    sg = np.minimum(sg, 1.0 - sw)
    sg[hc == capi.HC_OIL_ONLY] = 0.0
    so = 1.0 - sw - sg
    go = hc == capi.HC_GAS_ONLY
    sg[go] = 1.0 - sw[go]; so[go] = 0.0
    sat = np.stack([sw, so, sg], 1)
    rs = 250.0 * rng.random(n)
    rv = 5e-4 * rng.random(n)
    return State(p, sat, rs, rv, hc)
