"""Independent CPU restatement of the standard well model coupled to the oracle's reservoir equations (TEST INFRASTRUCTURE -- see
oracle/oracle.h; only tests/ may import this).

Why a second restatement: the wells tests used the product package's own host well model (opmgpu/wells.py) as the checker of the device
well model (csrc/wells.hip).  This one shares no code and no technique with either:
  * no hand-written derivatives: the well equations are plain functions of (properties of the perforated cells, well unknowns), written
    once for real or complex arguments, and every Jacobian entry comes from COMPLEX-STEP differentiation (Im f(x + ih) / h, h = 1e-30:
    exact to rounding, no subtraction); the properties' own derivatives with respect to the cells' primary variables come from
    oracle_cell_props;
  * no Schur complement and no Krylov solver: reservoir and well equations are assembled as ONE sparse matrix, the well unknowns (q_s of
    three phases, bhp) as extra rows and columns with all the perforation-to-perforation coupling in the explicit blocks, and solved by
    a sparse direct solver (SuperLU through scipy).
What it follows (file:line of the reference):
  computeWellFlux                      StandardWells_impl.hpp:397-570
  addWellFluxEq / addWellControlEq     :806-835 / :837-998 (BHP, SURFACE_RATE, RESERVOIR_RATE, dead wells; no THP here)
  updatePerfPhaseRatesAndPressures     :578-606
  updateWellState                      :612-700 (rates, limited bhp)
  updateWellControls                   :709-800, updateWellStateWithTarget :1450-1550, wellhelpers::constraintBroken (external: injectors
                                       break a limit from above, producers from below)
  computeWellConnectionPressures       :224-366 + WellDensitySegmented.cpp:30-196
  assemble order, well source terms    BlackoilModelBase_impl.hpp:757-840, :956-975
  solveWellEq / getWellConvergence     :1018-1133 / :1859-1905
  nonlinearIteration / updateState     :239-326 / :1147-1389 (the reservoir part is oracle_update_state)
Parity unpinned beyond the reference's test_welldensitysegmented known answer (tests/golden/welldensitysegmented.json, checked against
this file too); its value is independence: device == host model == this, by three different routes."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import oracle as orc

BHP, SURFACE_RATE, THP, RESERVOIR_RATE = 0, 1, 2, 3          # the numbering of opmgpu/wells.py (data schema)
INJECTOR, PRODUCER = 0, 1
GRAVITY = 9.80665
_H = 1e-30


class NumericalIssue(RuntimeError):
    pass


def connection_densities(connpos, comp_frac, perf_rates, b_perf, rsmax_perf, rvmax_perf, surf_dens_perf):
    """WellDensitySegmented::computeConnectionDensities (WellDensitySegmented.cpp:30-146); phases water, oil, gas"""
    nw = len(connpos) - 1
    nperf = connpos[nw]
    q_out = np.zeros((nperf, 3))
    for w in range(nw):
        for perf in range(connpos[w + 1] - 1, connpos[w] - 1, -1):       # bottom to top
            below = q_out[perf + 1] if perf != connpos[w + 1] - 1 else 0.0
            q_out[perf] = below - perf_rates[perf]
    dens = np.zeros(nperf)
    for w in range(nw):
        for perf in range(connpos[w], connpos[w + 1]):
            tot = q_out[perf].sum()
            mix = np.abs(q_out[perf] / tot) if tot != 0.0 else np.asarray(comp_frac[w], float).copy()
            x = mix.copy()
            rs = rv = 0.0
            if mix[1] > 0.0:
                rs = min(mix[2] / mix[1], rsmax_perf[perf])
            if mix[2] > 0.0:
                rv = min(mix[1] / mix[2], rvmax_perf[perf])
            if rs != 0.0:
                x[2] = (mix[2] - mix[1] * rs) / (1.0 - rs * rv)
            if rv != 0.0:
                x[1] = (mix[1] - mix[2] * rv) / (1.0 - rs * rv)
            volrat = (x / b_perf[perf]).sum()
            dens[perf] = float(np.dot(surf_dens_perf[perf], mix)) / volrat
    return dens


def connection_pressure_delta(connpos, depth_ref, z_perf, dens_perf, gravity=GRAVITY):
    """WellDensitySegmented::computeConnectionPressureDelta (:150-196)"""
    nw = len(connpos) - 1
    dp = np.zeros(connpos[nw])
    for w in range(nw):
        for perf in range(connpos[w], connpos[w + 1]):
            z_above = depth_ref[w] if perf == connpos[w] else z_perf[perf - 1]
            dp[perf] = (z_perf[perf] - z_above) * dens_perf[perf] * gravity
        dp[connpos[w]:connpos[w + 1]] = np.cumsum(dp[connpos[w]:connpos[w + 1]])
    return dp


class WellStateArrays:
    """bhp, wellRates, perfPress, perfPhaseRates, currentControls of WellStateFullyImplicitBlackoil (its init is external: given)"""

    def __init__(self, bhp, qs, perf_press, perf_rates, current):
        self.bhp, self.qs = np.array(bhp, float), np.array(qs, float).reshape(-1, 3)
        self.perf_press, self.perf_rates = np.array(perf_press, float), np.array(perf_rates, float).reshape(-1, 3)
        self.current = np.array(current, np.int64)

    def copy(self):
        return WellStateArrays(self.bhp, self.qs, self.perf_press, self.perf_rates, self.current)


class CoupledOracleModel:
    """BlackoilModelBase with StandardWells on the oracle's reservoir equations; `wells` is read as data (connpos, cells, WI, type,
    comp_frac, allow_cf, depth_ref, controls = per well a list of (type, target, distr, ...))."""

    def __init__(self, grid, tables, params, wells, well_state, dbhp_max_rel=1.0):
        self.grid, self.tab, self.prm = grid, tables, params
        self.nc = grid.nc
        self.connpos = np.asarray(wells.connpos, np.int64)
        self.cells = np.asarray(wells.cells, np.int64)
        self.nw, self.nperf = len(self.connpos) - 1, int(self.connpos[-1])
        self.WI = np.asarray(wells.WI, float)
        self.type = np.asarray(wells.type, np.int64)
        self.compi = np.asarray(wells.comp_frac, float).reshape(self.nw, 3)
        self.allow_cf = np.asarray([bool(a) for a in wells.allow_cf])
        self.depth_ref = np.asarray(wells.depth_ref, float)
        self.controls = [[(int(c[0]), float(c[1]), np.asarray(c[2], float)) for c in cl] for cl in wells.controls]
        assert all(c[0] != THP for cl in self.controls for c in cl), "THP controls are outside this restatement"
        self.perf_well = np.repeat(np.arange(self.nw), np.diff(self.connpos))
        self.perf_pos = np.arange(self.nperf) - self.connpos[self.perf_well]          # position of a perforation inside its well
        self.z_perf = np.asarray(grid.z, float)[self.cells]
        pvtnum = getattr(grid, "pvtnum", None)
        self.pvt_perf = None if pvtnum is None else np.asarray(pvtnum, np.int32)[self.cells]
        sd = np.asarray(tables.surface_density, float).reshape(-1, 3)
        self.surf_dens_perf = sd[0 if self.pvt_perf is None else self.pvt_perf] * np.ones((self.nperf, 3))
        self.ws = well_state
        self.dbhp_max_rel = float(dbhp_max_rel)
        self.rowptr, self.col = orc.pattern(grid)
        self.cdp = np.zeros(self.nperf)
        self.perf_dens = np.zeros(self.nperf)
        self.well_iterations = 0

    # ------------------------------------------------------------------ the well equations, real or complex
    def well_flux(self, p, mob, b, rs, rv, bhp, qs):
        """computeWellFlux: cq_s[nperf, 3], alive[nw].  Every selection looks at values only (real parts)."""
        pw = self.perf_well
        Tw = self.WI
        drawdown = p - (bhp[pw] + self.cdp)
        sel_inj = (drawdown.real < 0).astype(float)
        sel_prod = 1.0 - sel_inj
        n_inj = np.bincount(pw, sel_inj, self.nw)
        n_prod = np.bincount(pw, sel_prod, self.nw)
        for w in range(self.nw):
            if not self.allow_cf[w]:
                s = slice(self.connpos[w], self.connpos[w + 1])
                if self.type[w] == INJECTOR and n_inj[w] > 0:
                    sel_prod[s] = 0.0
                elif self.type[w] == PRODUCER and n_prod[w] > 0:
                    sel_inj[s] = 0.0
        # flow into the wellbore: phase volumetric rates at standard conditions
        cq_p = -(sel_prod * Tw)[:, None] * (mob * drawdown[:, None])
        cq_ps = b * cq_p
        cq_ps_oil, cq_ps_gas = cq_ps[:, 1].copy(), cq_ps[:, 2].copy()
        cq_ps[:, 2] = cq_ps[:, 2] + rs * cq_ps_oil
        cq_ps[:, 1] = cq_ps[:, 1] + rv * cq_ps_gas
        # flow out of the wellbore, total mobility
        cqt_i = -(sel_inj * Tw) * (mob.sum(1) * drawdown)
        # wellbore mixture at standard conditions
        q_ps = np.zeros((self.nw, 3), dtype=cq_ps.dtype)
        np.add.at(q_ps, pw, cq_ps)
        wbq = self.compi * np.where(qs.real > 0, qs, 0.0) - q_ps
        wbqt = wbq.sum(1)
        dead = wbqt.real == 0
        safe = np.where(dead, 1.0, wbqt)
        cmix_w = np.where(dead[:, None], self.compi, wbq / safe[:, None])
        cmix = cmix_w[pw]
        d = 1.0 - rv * rs
        volume_ratio = cmix[:, 0] / b[:, 0] + (cmix[:, 1] - rv * cmix[:, 2]) / d / b[:, 1] + (cmix[:, 2] - rs * cmix[:, 1]) / d / b[:, 2]
        cqt_is = cqt_i / volume_ratio
        cq_s = cq_ps + cmix * cqt_is[:, None]
        return cq_s, ~dead

    def well_potentials(self, props):
        """StandardWells::computeWellPotentials (StandardWells_impl.hpp:1003-1095) for BHP limits (THP is outside this restatement): every
        well at the last BHP target of its control list (0 without one), rates from the explicit cell properties: [nw, 3]"""
        bhp = np.zeros(self.nw)
        for w in range(self.nw):
            for typ, target, _ in self.controls[w]:
                if typ == BHP:
                    bhp[w] = target
        cq_s, _ = self.well_flux(props["p"], props["mob"], props["b"], props["rs"], props["rv"], bhp, self.ws.qs)
        pot = np.zeros((self.nw, 3))
        np.add.at(pot, self.perf_well, cq_s)
        return pot

    def equations(self, props, bhp, qs):
        """flux equations [nw, 3], control equations [nw], cq_s [nperf, 3], alive"""
        cq_s, alive = self.well_flux(props["p"], props["mob"], props["b"], props["rs"], props["rv"], bhp, qs)
        q = np.zeros((self.nw, 3), dtype=cq_s.dtype)
        np.add.at(q, self.perf_well, cq_s)
        flux_eq = qs - q
        ctrl = np.zeros(self.nw, dtype=np.result_type(bhp.dtype, qs.dtype))
        for w in range(self.nw):
            typ, target, distr = self.controls[w][int(self.ws.current[w])]
            if not alive[w]:
                ctrl[w] = qs[w].sum()
            elif typ == BHP:
                ctrl[w] = bhp[w] - target
            else:                                   # SURFACE_RATE and RESERVOIR_RATE look the same (:938-955)
                ctrl[w] = (distr * qs[w]).sum() - target
        return flux_eq, ctrl, cq_s, alive

    # ------------------------------------------------------------------ properties of the perforated cells
    def perf_props(self, st):
        """values and derivatives (value, d/dP, d/dSw, d/dXvar) of p_o, mob_a, b_a, rs, rv in the perforated cells"""
        allp = orc.cell_props(self.grid, self.tab, st)[self.cells]
        names = orc.PROP_NAMES
        g = lambda n: allp[:, names.index(n), :]
        vals = {"p": g("p_o")[:, 0].copy(), "mob": np.stack([g("mob_" + a)[:, 0] for a in "wog"], 1), "b": np.stack([g("b_" + a)[:, 0] for a in "wog"], 1),
                "rs": g("rs")[:, 0].copy(), "rv": g("rv")[:, 0].copy()}
        # order of the 9 scalar properties: p, mob_w, mob_o, mob_g, b_w, b_o, b_g, rs, rv
        ders = np.stack([g("p_o")[:, 1:]] + [g("mob_" + a)[:, 1:] for a in "wog"] + [g("b_" + a)[:, 1:] for a in "wog"] + [g("rs")[:, 1:], g("rv")[:, 1:]], 1)
        return vals, ders                          # ders[nperf, 9, 3]

    @staticmethod
    def _perturbed(vals, k, mask):
        """props with i*h added to scalar property k of the perforations in mask"""
        out = {n: v.astype(complex) for n, v in vals.items()}
        step = 1j * _H * mask
        if k == 0:
            out["p"] = out["p"] + step
        elif k <= 3:
            out["mob"][:, k - 1] += step
        elif k <= 6:
            out["b"][:, k - 4] += step
        elif k == 7:
            out["rs"] = out["rs"] + step
        else:
            out["rv"] = out["rv"] + step
        return out

    def linearise(self, vals, ders, bhp, qs, with_cells=True):
        """Residuals and, by complex steps, every derivative of the well part:
        E[nw, 4] (three flux equations, control equation), cq_s;  dE/dy[nw, 4, 4] (y = q_s[3], bhp);  dcq/dy[nperf, 3, 4];
        dE/dx[nperf, 4, 3] (equations of the perforation's well, primary variables of its cell);  dcq/dx[(i, j)] -> [3, 3] for every
        pair of perforations of one well (source of cell i, variables of cell j)."""
        nw, nperf, pw = self.nw, self.nperf, self.perf_well
        flux_eq, ctrl, cq_s, alive = self.equations(vals, bhp, qs)
        E = np.concatenate([flux_eq, ctrl[:, None]], 1)
        cvals = {n: v.astype(complex) for n, v in vals.items()}
        dE_dy, dcq_dy = np.zeros((nw, 4, 4)), np.zeros((nperf, 3, 4))
        for v in range(4):                                               # all wells at once: wells do not couple
            bh, q = bhp.astype(complex), qs.astype(complex)
            if v < 3:
                q[:, v] += 1j * _H
            else:
                bh = bh + 1j * _H
            f, c, cq, _ = self.equations(cvals, bh, q)
            dE_dy[:, :3, v], dE_dy[:, 3, v], dcq_dy[:, :, v] = f.imag / _H, c.imag / _H, cq.imag / _H
        if not with_cells:
            return E, cq_s, alive, dE_dy, dcq_dy, None, None
        dE_dx = np.zeros((nperf, 4, 3))
        dcq_dx = {}
        maxlen = int(np.diff(self.connpos).max())
        for pos in range(maxlen):                                        # the pos-th perforation of every well at once
            mask = (self.perf_pos == pos).astype(float)
            dE_du, dcq_du = np.zeros((nw, 4, 9)), np.zeros((nperf, 3, 9))
            for k in range(9):
                f, c, cq, _ = self.equations(self._perturbed(vals, k, mask), bhp.astype(complex), qs.astype(complex))
                dE_du[:, :3, k], dE_du[:, 3, k], dcq_du[:, :, k] = f.imag / _H, c.imag / _H, cq.imag / _H
            for j in np.flatnonzero(mask):
                w = pw[j]
                dE_dx[j] = dE_du[w] @ ders[j]                          # chain rule through the 9 properties of cell j
                for i in range(self.connpos[w], self.connpos[w + 1]):
                    dcq_dx[(i, int(j))] = dcq_du[i] @ ders[j]
        return E, cq_s, alive, dE_dy, dcq_dy, dE_dx, dcq_dx

    # ------------------------------------------------------------------ control logic
    def _apply_target(self, w, current):
        typ, target, distr = self.controls[w][current]
        ws = self.ws
        if typ == BHP:
            ws.bhp[w] = target
        elif typ == SURFACE_RATE:
            if self.type[w] == INJECTOR:
                for a in range(3):
                    if self.compi[w, a] > 0.0:
                        ws.qs[w, a] = target * self.compi[w, a]
            else:
                n = int((distr > 0.0).sum())
                for a in range(3):
                    if distr[a] > 0.0 and n < 2:
                        ws.qs[w, a] = target * distr[a]
        # RESERVOIR_RATE: the existing rates stay (:1514-1519)

    def _broken(self, w, k):
        typ, target, distr = self.controls[w][k]
        val = self.ws.bhp[w] if typ == BHP else float(np.dot(distr, self.ws.qs[w]))
        return val > target if self.type[w] == INJECTOR else val < target

    def update_well_controls(self):
        for w in range(self.nw):
            nwc = len(self.controls[w])
            current, its = int(self.ws.current[w]), 0
            while True:
                self._apply_target(w, current)
                k = 0
                while k < nwc and (k == current or not self._broken(w, k)):
                    k += 1
                violated = k != nwc
                if violated:
                    self.ws.current[w] = current = k
                its += 1
                if its > 2 * nwc:
                    raise NumericalIssue("Could not find proper control within %d iterations!" % its)
                if not violated:
                    break

    def update_well_state(self, dy):
        """dy[nw, 4] = increments of (q_s, bhp)"""
        ws = self.ws
        ws.qs -= dy[:, :3]
        d = dy[:, 3]
        ws.bhp = ws.bhp - np.sign(d) * np.minimum(np.abs(d), np.abs(ws.bhp) * self.dbhp_max_rel)

    def _store_perf(self, cq_s):
        self.ws.perf_rates = np.array(cq_s.real, float)
        self.ws.perf_press = self.ws.bhp[self.perf_well] + self.cdp

    def compute_connection_pressures(self):
        """computePropertiesForWellConnectionPressures + computeWellConnectionDensitesPressures (:224-340)"""
        ws, st, t = self.ws, self.st, self.tab
        avg = np.zeros(self.nperf)
        for w in range(self.nw):
            for perf in range(self.connpos[w], self.connpos[w + 1]):
                above = ws.bhp[w] if perf == self.connpos[w] else ws.perf_press[perf - 1]
                avg[perf] = (ws.perf_press[perf] + above) / 2
        hc = st.hc[self.cells]
        free_gas = (hc != 2).astype(np.int8) if t.has_disgas else np.ones(self.nperf, np.int8)        # hydroCarbonState: 0 gas only, 1 gas and oil, 2 oil only
        free_oil = (hc != 0).astype(np.int8) if t.has_vapoil else np.ones(self.nperf, np.int8)
        b = np.stack([orc.pvt(t, "bWat", avg, pvtnum=self.pvt_perf)[:, 0],
                      orc.pvt(t, "bOil", avg, st.rs[self.cells], free_gas, self.pvt_perf)[:, 0],
                      orc.pvt(t, "bGas", avg, st.rv[self.cells], free_oil, self.pvt_perf)[:, 0]], 1)
        rsmax = orc.pvt(t, "rsSat", avg, pvtnum=self.pvt_perf)[:, 0] if t.has_disgas else np.zeros(self.nperf)
        rvmax = orc.pvt(t, "rvSat", avg, pvtnum=self.pvt_perf)[:, 0] if t.has_vapoil else np.zeros(self.nperf)
        assert t.vap1 == 0.0 and t.vap2 == 0.0, "VAPPARS is outside this restatement"
        self.perf_dens = connection_densities(self.connpos, self.compi, ws.perf_rates, b, rsmax, rvmax, self.surf_dens_perf)
        self.cdp = connection_pressure_delta(self.connpos, self.depth_ref, self.z_perf, self.perf_dens)

    def well_convergence(self, E, B_avg):
        self.well_flux_residual = B_avg * np.abs(E[:, :3]).max(0)
        self.well_ctrl_residual = float(np.abs(E[:, 3]).max())
        if np.isnan(self.well_flux_residual).any():
            raise NumericalIssue("NaN residual for a well flux equation")
        if (self.well_flux_residual > self.prm.max_residual_allowed).any():
            raise NumericalIssue("Too large residual for a well flux equation")
        return bool((self.well_flux_residual < self.prm.tolerance_wells).all() and self.well_ctrl_residual < self.prm.tolerance_well_control)

    def solve_well_eq(self, vals, B_avg):
        """solveWellEq (BlackoilModelBase_impl.hpp:1018-1133): Newton on the well unknowns with the reservoir frozen"""
        ws0 = self.ws.copy()
        it, converged = 0, False
        while True:
            E, cq_s, _, dE_dy, _, _, _ = self.linearise(vals, None, self.ws.bhp, self.ws.qs, with_cells=False)
            self._store_perf(cq_s)
            converged = self.well_convergence(E, B_avg)
            if converged:
                break
            it += 1
            dy = np.stack([np.linalg.solve(dE_dy[w], E[w]) for w in range(self.nw)])
            self.update_well_state(dy)
            self.update_well_controls()
            if it >= 15:
                break
        self.well_iterations = it
        if converged:
            self.compute_connection_pressures()
        else:
            self.ws = ws0
        return converged

    # ------------------------------------------------------------------ the model
    def prepareStep(self, dt, state):
        self.dt = float(dt)
        self.st = state.copy()
        self.acc0 = None

    def assemble(self, initial):
        """assemble (:757-840): control switching, reservoir equations, [connection pressures + well pre-solve], well equations"""
        self.update_well_controls()
        if initial:
            self.acc0 = None
        self.r, self.val, self.acc0, self.binv = orc.assemble(self.grid, self.tab, self.dt, self.st, self.rowptr, self.col, scale=(1.0, 1.0, 1.0), accum0=self.acc0)
        if initial:
            self.compute_connection_pressures()
        vals, ders = self.perf_props(self.st)
        self.B_avg = self.binv.reshape(3, self.nc).mean(1)
        if initial and self.prm.solve_welleq_initially:
            self.presolve_converged = self.solve_well_eq(vals, self.B_avg)
        self.E, cq_s, self.alive, self.dE_dy, self.dcq_dy, self.dE_dx, self.dcq_dx = self.linearise(vals, ders, self.ws.bhp, self.ws.qs)
        self._store_perf(cq_s)
        for a in range(3):                                    # material_balance_eq[phase] -= cq_s[phase] at the well cells (:956-975)
            np.add.at(self.r, a * self.nc + self.cells, -cq_s[:, a].real)

    def getConvergence(self):
        st, self.B_avg, self.CNV, self.MB, self.linf, conv = orc.convergence(self.grid, self.prm, self.dt, self.r, self.binv)
        if st != 0:
            raise NumericalIssue("NaN or too large reservoir residual")
        return self.well_convergence(self.E, self.B_avg) and conv

    def solveJacobianSystem(self):
        """ONE sparse system: unknowns [cell 0: P Sw Xvar, cell 1: ..., well 0: q_w q_o q_g bhp, ...], solved directly"""
        nc, nw = self.nc, self.nw
        n = 3 * nc + 4 * nw
        A = sp.bsr_matrix((self.val.reshape(-1, 3, 3), self.col, self.rowptr), shape=(3 * nc, 3 * nc)).tocoo()
        rows, cols, data = [A.row], [A.col], [A.data]
        rr, cc, dd = [], [], []

        def block(r0, c0, m):
            m = np.asarray(m)
            for a in range(m.shape[0]):
                for v in range(m.shape[1]):
                    rr.append(r0 + a); cc.append(c0 + v); dd.append(m[a, v])

        for (i, j), m in self.dcq_dx.items():                 # well source of cell i against the variables of cell j: -d cq_s / dx
            block(3 * self.cells[i], 3 * self.cells[j], -m)
        for j in range(self.nperf):
            w = self.perf_well[j]
            block(3 * self.cells[j], 3 * nc + 4 * w, -self.dcq_dy[j])          # reservoir rows, well columns
            block(3 * nc + 4 * w, 3 * self.cells[j], self.dE_dx[j])           # well rows, reservoir columns
        for w in range(nw):
            block(3 * nc + 4 * w, 3 * nc + 4 * w, self.dE_dy[w])
        rows.append(np.asarray(rr)); cols.append(np.asarray(cc)); data.append(np.asarray(dd, float))
        J = sp.coo_matrix((np.concatenate(data), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsc()
        rhs = np.concatenate([np.ascontiguousarray(self.r.reshape(3, nc).T).ravel(), self.E.ravel()])
        d = spla.spsolve(J, rhs)
        if not np.isfinite(d).all():
            raise NumericalIssue("the direct solve of the coupled system failed")
        self.dx = np.ascontiguousarray(d[:3 * nc].reshape(nc, 3).T).ravel()
        self.dy = d[3 * nc:].reshape(nw, 4)
        return self.dx

    def updateState(self):
        self.st = orc.update_state(self.grid, self.tab, self.prm, self.dx, self.st)
        self.update_well_state(self.dy)

    def nonlinearIteration(self, iteration, min_iter=1):
        """nonlinearIteration (:239-326) without update stabilisation"""
        self.assemble(iteration == 0)
        converged = self.getConvergence()
        if iteration < min_iter or not converged:
            self.solveJacobianSystem()
            self.updateState()
        return converged
