"""Host-side standard well model: stays on the CPU like the reference's.

Restates, on small dense per-well systems, what `Opm::StandardWells` does with AutoDiffBlocks:
  computeWellFlux            opm/autodiff/StandardWells_impl.hpp:396-571
  addWellFluxEq              :806-829          addWellControlEq (BHP / SURFACE_RATE / dead wells)  :836-998
  updateWellState            :611-650          computeWellConnectionPressures  :223-298
  WellDensitySegmented::{computeConnectionDensities, computeConnectionPressureDelta}   WellDensitySegmented.cpp:66-181
  addWellContributionToMassBalanceEq            opm/autodiff/BlackoilModelBase_impl.hpp:953-975
  eliminateVariable / recoverVariable (Schur)   opm/autodiff/NewtonIterationUtilities.cpp:45-184
The reservoir side is any backend with the GpuBlackoilModel interface (the device model, or the oracle
in the parity tests): per-perforation cell properties come from `perfProps`, the Schur-reduced well
terms go back through `addWellTerms` / `addWellRhs`, the perforated cells' increments through `perfDx`.

Not restated (documented simplifications): control switching (updateWellControls), the explicit well
pre-solve (solveWellEq), THP/VFP and group controls, RESERVOIR_RATE conversion, efficiency factors;
connection densities use the perforated cells' own b / rs / rv instead of re-evaluating the PVT at the
average well-block pressure.
"""
import numpy as np

INJECTOR, PRODUCER = 0, 1
BHP, SURFACE_RATE = 0, 1
GRAVITY = 9.80665


class AD:
    """Tiny dense forward AD over a vector of n entries and nv independent variables."""

    __slots__ = ("v", "j")
    __array_ufunc__ = None          # numpy arrays on the left defer to __rmul__ / __radd__ ...

    def __init__(self, v, j):
        self.v, self.j = np.asarray(v, dtype=float), np.asarray(j, dtype=float)

    @staticmethod
    def const(v, nv):
        v = np.atleast_1d(np.asarray(v, dtype=float))
        return AD(v, np.zeros((v.size, nv)))

    def _co(self, o):
        return o if isinstance(o, AD) else AD.const(np.broadcast_to(np.asarray(o, dtype=float), self.v.shape), self.j.shape[1])

    def __add__(self, o):
        o = self._co(o); return AD(self.v + o.v, self.j + o.j)
    __radd__ = __add__

    def __sub__(self, o):
        o = self._co(o); return AD(self.v - o.v, self.j - o.j)

    def __rsub__(self, o):
        o = self._co(o); return AD(o.v - self.v, o.j - self.j)

    def __neg__(self):
        return AD(-self.v, -self.j)

    def __mul__(self, o):
        o = self._co(o); return AD(self.v * o.v, self.j * o.v[:, None] + o.j * self.v[:, None])
    __rmul__ = __mul__

    def __truediv__(self, o):
        o = self._co(o); q = self.v / o.v
        return AD(q, (self.j - o.j * q[:, None]) / o.v[:, None])

    def __rtruediv__(self, o):
        return self._co(o) / self

    def sum(self):
        return AD(np.array([self.v.sum()]), self.j.sum(0, keepdims=True))

    def bcast(self, n):        # (1,) -> (n,)
        return AD(np.repeat(self.v, n), np.repeat(self.j, n, axis=0))


class Wells:
    """The parts of opm-core's `Wells` struct the model reads."""

    def __init__(self):
        self.type, self.depth_ref, self.comp_frac, self.allow_cf = [], [], [], []
        self.connpos, self.cells, self.WI = [0], [], []
        self.ctrl_type, self.ctrl_target, self.ctrl_distr = [], [], []
        self.name = []

    def add_well(self, name, wtype, depth_ref, cells, WI, comp_frac, control, allow_cf=True):
        self.name.append(name); self.type.append(wtype); self.depth_ref.append(float(depth_ref))
        self.comp_frac.append(np.asarray(comp_frac, float)); self.allow_cf.append(bool(allow_cf))
        self.cells += [int(c) for c in cells]; self.WI += [float(w) for w in np.broadcast_to(WI, (len(cells),))]
        self.connpos.append(len(self.cells))
        self.ctrl_type.append(control[0]); self.ctrl_target.append(float(control[1]))
        self.ctrl_distr.append(np.asarray(control[2] if len(control) > 2 else (0, 0, 0), float))
        return self

    @property
    def nw(self):
        return len(self.type)

    @property
    def nperf(self):
        return len(self.cells)

    def arrays(self):
        return np.asarray(self.connpos, np.int32), np.asarray(self.cells, np.int32)


class WellState:
    """WellStateFullyImplicitBlackoil fields used here: bhp, wellRates (well-major), perfPress, perfPhaseRates."""

    def __init__(self, wells, cell_pressure):
        nw = wells.nw
        self.bhp, self.qs = np.zeros(nw), np.zeros((nw, 3))
        self.perf_press = np.zeros(wells.nperf); self.perf_rates = np.zeros((wells.nperf, 3))
        for w in range(nw):
            p0 = cell_pressure[wells.cells[wells.connpos[w]]]
            if wells.ctrl_type[w] == BHP:
                self.bhp[w] = wells.ctrl_target[w]
            else:       # rate control: start near the first perforated cell's pressure, rate at target
                self.bhp[w] = p0 * (1.01 if wells.type[w] == INJECTOR else 0.99)
                d = wells.ctrl_distr[w]
                self.qs[w] = wells.ctrl_target[w] * d / max(d.sum(), 1e-300) if wells.type[w] == PRODUCER else wells.ctrl_target[w] * wells.comp_frac[w]
            self.perf_press[wells.connpos[w]:wells.connpos[w + 1]] = self.bhp[w]

    def copy(self):
        import copy
        return copy.deepcopy(self)

    def assign(self, other):
        """state = last_state (AdaptiveTimeStepping_impl.hpp:346-347), in place so that holders of this object see it"""
        self.bhp[:], self.qs[:], self.perf_press[:], self.perf_rates[:] = other.bhp, other.qs, other.perf_press, other.perf_rates


def connection_densities(wells, perf_rates, b_perf, rsmax_perf, rvmax_perf, surf_dens_perf):
    """WellDensitySegmented::computeConnectionDensities (WellDensitySegmented.cpp:66-135); components w, o, g."""
    nperf = wells.nperf
    q_out = np.zeros((nperf, 3))
    for w in range(wells.nw):
        lo, hi = wells.connpos[w], wells.connpos[w + 1]
        for perf in range(hi - 1, lo - 1, -1):          # bottom to top
            below = q_out[perf + 1] if perf < hi - 1 else 0.0
            q_out[perf] = below - perf_rates[perf]
    dens = np.zeros(nperf)
    for w in range(wells.nw):
        for perf in range(wells.connpos[w], wells.connpos[w + 1]):
            tot = q_out[perf].sum()
            mix = np.abs(q_out[perf] / tot) if tot != 0.0 else np.asarray(wells.comp_frac[w], float).copy()
            x = mix.copy()
            rs = rv = 0.0
            if rsmax_perf is not None and mix[1] > 0.0:
                rs = min(mix[2] / mix[1], rsmax_perf[perf])
            if rvmax_perf is not None and mix[2] > 0.0:
                rv = min(mix[1] / mix[2], rvmax_perf[perf])
            if rs != 0.0:
                x[2] = (mix[2] - mix[1] * rs) / (1.0 - rs * rv)
            if rv != 0.0:
                x[1] = (mix[1] - mix[2] * rv) / (1.0 - rs * rv)
            volrat = (x / b_perf[perf]).sum()
            dens[perf] = float(np.dot(surf_dens_perf[perf], mix)) / volrat
    return dens


def connection_pressure_delta(wells, z_perf, dens_perf, gravity=GRAVITY):
    """WellDensitySegmented::computeConnectionPressureDelta (WellDensitySegmented.cpp:140-181)."""
    dp = np.zeros(wells.nperf)
    for w in range(wells.nw):
        lo, hi = wells.connpos[w], wells.connpos[w + 1]
        for perf in range(lo, hi):
            z_above = wells.depth_ref[w] if perf == lo else z_perf[perf - 1]
            dp[perf] = (z_perf[perf] - z_above) * dens_perf[perf] * gravity
        dp[lo:hi] = np.cumsum(dp[lo:hi])
    return dp


class StandardWellsHost:
    def __init__(self, wells, z_cells, surface_density_wog, gravity=GRAVITY, dbhp_max_rel=1.0,
                 tolerance_wells=1e-4, tolerance_well_control=1e-7):
        self.w, self.gravity, self.dbhp_max_rel = wells, gravity, dbhp_max_rel
        self.z_perf = np.asarray(z_cells, float)[np.asarray(wells.cells, int)]
        self.surf_dens = np.asarray(surface_density_wog, float).reshape(1, 3)
        self.tol_wells, self.tol_ctrl = tolerance_wells, tolerance_well_control
        self.cdp = np.zeros(wells.nperf)
        self._sys = None

    # computeWellConnectionPressures: once per time step, from the explicit state (BlackoilModelBase_impl.hpp:797-805)
    def compute_connection_pressures(self, pp, ws):
        b = pp[:, 3:6, 0]
        dens = connection_densities(self.w, ws.perf_rates, b, pp[:, 1, 0], pp[:, 2, 0], np.repeat(self.surf_dens, self.w.nperf, 0))
        self.cdp = connection_pressure_delta(self.w, self.z_perf, dens, self.gravity)

    # computeWellFlux + addWellFluxEq + addWellControlEq for all wells; Schur-reduce every well onto its cells
    def assemble(self, pp, ws):
        W = self.w
        nperf = W.nperf
        resid_delta = np.zeros((nperf, 3)); rhs_delta = np.zeros((nperf, 3))
        rc, blocks = [], []
        self._sys = []
        flux_eq = np.zeros((W.nw, 3)); ctrl_eq = np.zeros(W.nw)
        cq_all = np.zeros((nperf, 3))
        for w in range(W.nw):
            lo, hi = W.connpos[w], W.connpos[w + 1]
            n = hi - lo
            nv = 3 * n + 4
            I = np.arange(lo, hi)

            def perf_q(k):      # perf quantity k of OPMGPU_PERF_K as AD w.r.t. its own cell's (P, Sw, Xvar)
                j = np.zeros((n, nv))
                for d in range(3):
                    j[np.arange(n), 3 * np.arange(n) + d] = pp[I, k, 1 + d]
                return AD(pp[I, k, 0], j)
            p_cell, rs, rv = perf_q(0), perf_q(1), perf_q(2)
            b = [perf_q(3), perf_q(4), perf_q(5)]
            mob = [perf_q(6), perf_q(7), perf_q(8)]
            jb = np.zeros((1, nv)); jb[0, 3 * n + 3] = 1.0
            bhp = AD(np.array([ws.bhp[w]]), jb)
            qs = []
            for a in range(3):
                jq = np.zeros((1, nv)); jq[0, 3 * n + a] = 1.0
                qs.append(AD(np.array([ws.qs[w, a]]), jq))
            Tw = np.asarray(W.WI[lo:hi])
            drawdown = p_cell - (bhp.bcast(n) + self.cdp[lo:hi])
            sel_inj = (drawdown.v < 0).astype(float); sel_prod = 1.0 - sel_inj
            if not W.allow_cf[w]:
                if W.type[w] == INJECTOR and sel_inj.sum() > 0:
                    sel_prod[:] = 0.0
                elif W.type[w] == PRODUCER and sel_prod.sum() > 0:
                    sel_inj[:] = 0.0
            cq_ps = [b[a] * (-(sel_prod * Tw) * (mob[a] * drawdown)) for a in range(3)]     # flow INTO the wellbore
            cq_ps_oil, cq_ps_gas = cq_ps[1], cq_ps[2]
            cq_ps[2] = cq_ps[2] + rs * cq_ps_oil
            cq_ps[1] = cq_ps[1] + rv * cq_ps_gas
            total_mob = mob[0] + mob[1] + mob[2]
            cqt_i = -(sel_inj * Tw) * (total_mob * drawdown)                                # flow OUT of the wellbore
            compi = W.comp_frac[w]
            wbq = []
            for a in range(3):
                inj = qs[a] if qs[a].v[0] > 0 else AD.const([0.0], nv)
                wbq.append(compi[a] * inj - cq_ps[a].sum())
            wbqt = wbq[0] + wbq[1] + wbq[2]
            alive = wbqt.v[0] != 0.0
            cmix = [(wbq[a] / wbqt if alive else AD.const([compi[a]], nv)).bcast(n) for a in range(3)]
            d = 1.0 - rv * rs
            vol = cmix[0] / b[0] + ((cmix[1] - rv * cmix[2]) / d) / b[1] + ((cmix[2] - rs * cmix[1]) / d) / b[2]
            cqt_is = cqt_i / vol
            cq_s = [cq_ps[a] + cmix[a] * cqt_is for a in range(3)]
            # well equations: E = [q_s - sum cq_s (3), control (1)]
            E = [qs[a] - cq_s[a].sum() for a in range(3)]
            if not alive:
                ctrl = qs[0] + qs[1] + qs[2]
            elif W.ctrl_type[w] == BHP:
                ctrl = bhp - W.ctrl_target[w]
            else:
                ctrl = W.ctrl_distr[w][0] * qs[0] + W.ctrl_distr[w][1] * qs[1] + W.ctrl_distr[w][2] * qs[2] - W.ctrl_target[w]
            E.append(ctrl)
            Ev = np.array([e.v[0] for e in E]); Ej = np.vstack([e.j for e in E])
            C, D = Ej[:, :3 * n], Ej[:, 3 * n:]
            # cell rows: R_a[cell_i] -= cq_s[a][i]   (addWellContributionToMassBalanceEq)
            Jc = np.zeros((3 * n, nv))
            for a in range(3):
                Jc[a::3] = -cq_s[a].j
                resid_delta[I, a] = -cq_s[a].v
                cq_all[I, a] = cq_s[a].v
            Jcc, B = Jc[:, :3 * n], Jc[:, 3 * n:]
            Dinv = np.linalg.inv(D)
            S = Jcc - B @ Dinv @ C                          # Schur complement (eliminateVariable x2)
            rhs_delta[I] = (-(B @ Dinv @ Ev)).reshape(n, 3)
            cells = np.asarray(W.cells[lo:hi])
            rc.append(np.stack([np.repeat(cells, n), np.tile(cells, n)], 1))
            blocks.append(S.reshape(n, 3, n, 3).transpose(0, 2, 1, 3).reshape(n * n, 9))
            self._sys.append((Dinv, C, Ev))
            flux_eq[w] = Ev[:3]; ctrl_eq[w] = Ev[3]
        ws.perf_rates = cq_all                                # updatePerfPhaseRatesAndPressures
        for w in range(W.nw):
            ws.perf_press[W.connpos[w]:W.connpos[w + 1]] = ws.bhp[w] + self.cdp[W.connpos[w]:W.connpos[w + 1]]
        self.flux_eq, self.ctrl_eq = flux_eq, ctrl_eq
        return resid_delta, np.concatenate(rc).astype(np.int32), np.concatenate(blocks), rhs_delta

    def converged(self, B_avg):
        """well part of getConvergence (BlackoilModelBase_impl.hpp:1769-1779)."""
        wf = np.asarray(B_avg) * np.abs(self.flux_eq).max(0)
        self.well_flux_residual, self.well_ctrl_residual = wf, np.abs(self.ctrl_eq).max()
        return bool(np.all(wf < self.tol_wells) and self.well_ctrl_residual < self.tol_ctrl)

    def recover_and_update(self, dx_perf, ws):
        """recoverVariable (NewtonIterationUtilities.cpp:134-184) + updateWellState (StandardWells_impl.hpp:611-650)."""
        W = self.w
        for w in range(W.nw):
            lo, hi = W.connpos[w], W.connpos[w + 1]
            Dinv, C, Ev = self._sys[w]
            dy = Dinv @ (Ev - C @ np.asarray(dx_perf[lo:hi], float).ravel())
            ws.qs[w] -= dy[:3]
            d = dy[3]
            ws.bhp[w] -= np.sign(d) * min(abs(d), abs(ws.bhp[w]) * self.dbhp_max_rel)


class WellCoupledModel:
    """BlackoilModelBase::nonlinearIteration with wells: reservoir backend (device) + StandardWellsHost."""

    def __init__(self, backend, wells_host, well_state):
        self.m, self.wh, self.ws = backend, wells_host, well_state
        self.nperf = wells_host.w.nperf
        self.linear_iterations = 0

    def prepareStep(self, dt, state=None):
        self.m.prepareStep(dt, state)

    def saveState(self):
        self.m.saveState()

    def restoreState(self):
        self.m.restoreState()

    def relativeChange(self):
        return self.m.relativeChange()

    def nonlinearIteration(self, iteration, single_precision=None, nonlinear_solver=None):
        m, wh, ws = self.m, self.wh, self.ws
        m.assemble(iteration == 0)
        pp = m.perfProps(self.nperf).reshape(self.nperf, 9, 4)
        if iteration == 0:
            wh.compute_connection_pressures(pp, ws)
        resid_delta, rc, blocks, rhs_delta = wh.assemble(pp, ws)
        m.addWellTerms(resid_delta, rc, blocks)
        m.addWellRhs(rhs_delta)
        converged = m.getConvergence()
        converged = wh.converged(m.B_avg) and converged
        lin = 0
        if not converged or iteration < 1:
            m.solveJacobianSystem(single_precision=single_precision)
            lin = self.linear_iterations = m.linear_iterations
            wh.recover_and_update(m.perfDx(self.nperf), ws)
            m.updateState()
        return converged, lin


class DeviceWellModel:
    """The same nonlinear iteration with the well model ON THE DEVICE (csrc/wells.hip): no per-iteration
    read-back of perforation properties, no clique fill -- opmgpu_set_device_wells / well_state_set / well_convergence.
    Interface of WellCoupledModel, so NonlinearSolver / AdaptiveTimeStepping drive either."""

    def __init__(self, backend, wells, well_state, tolerance_wells=1e-4, tolerance_well_control=1e-7):
        import ctypes as C
        from . import capi
        self.m, self.w, self.ws = backend, wells, well_state
        self.tol_wells, self.tol_ctrl = tolerance_wells, tolerance_well_control
        self.linear_iterations = 0
        nw = wells.nw
        spec = capi.WellsSpec()
        self._keep = [np.asarray(wells.connpos, np.int32), np.asarray(wells.cells, np.int32), capi.f64(wells.WI),
                      np.asarray(wells.type, np.int32), np.asarray([int(a) for a in wells.allow_cf], np.int32), capi.f64(wells.depth_ref),
                      capi.f64(np.asarray(wells.comp_frac, float).reshape(nw, 3)), np.asarray(wells.ctrl_type, np.int32),
                      capi.f64(wells.ctrl_target), capi.f64(np.asarray(wells.ctrl_distr, float).reshape(nw, 3))]
        k = self._keep
        spec.nw = nw
        spec.well_connpos, spec.well_cells, spec.WI, spec.type, spec.allow_cf = capi.iptr(k[0]), capi.iptr(k[1]), capi.dptr(k[2]), capi.iptr(k[3]), capi.iptr(k[4])
        spec.depth_ref, spec.comp_frac, spec.ctrl_type, spec.ctrl_target, spec.ctrl_distr = capi.dptr(k[5]), capi.dptr(k[6]), capi.iptr(k[7]), capi.dptr(k[8]), capi.dptr(k[9])
        backend._chk(backend.lib.opmgpu_set_device_wells(backend.ctx, C.byref(spec)))
        self.push_well_state()

    def push_well_state(self):
        from . import capi
        m, ws = self.m, self.ws
        if self.w.nw == 0:          # a rank without wells (multi-GPU): nothing to upload, the convergence call stays collective
            return
        m._chk(m.lib.opmgpu_well_state_set(m.ctx, capi.dptr(capi.f64(ws.bhp)), capi.dptr(capi.f64(ws.qs)), capi.dptr(capi.f64(ws.perf_rates))))

    def pull_well_state(self):
        """device well state -> the WellState object (bhp, wellRates, perfPress, perfPhaseRates)"""
        from . import capi
        m, ws = self.m, self.ws
        if self.w.nw == 0:
            return ws
        bhp, qs = np.zeros(self.w.nw), np.zeros((self.w.nw, 3))
        pp, pr = np.zeros(self.w.nperf), np.zeros((self.w.nperf, 3))
        m._chk(m.lib.opmgpu_well_state_get(m.ctx, capi.dptr(bhp), capi.dptr(qs), capi.dptr(pp), capi.dptr(pr)))
        ws.bhp[:], ws.qs[:], ws.perf_press[:], ws.perf_rates[:] = bhp, qs, pp, pr
        return ws

    def prepareStep(self, dt, state=None):
        self.m.prepareStep(dt, state)

    def saveState(self):
        self.m.saveState()

    def restoreState(self):
        self.m.restoreState()

    def relativeChange(self):
        return self.m.relativeChange()

    def wellConvergence(self):
        from . import capi
        m = self.m
        flux, ctrl = np.zeros(3), np.zeros(1)
        m._chk(m.lib.opmgpu_well_convergence(m.ctx, capi.dptr(flux), capi.dptr(ctrl)))
        self.well_flux_residual, self.well_ctrl_residual = np.asarray(m.B_avg) * flux, float(ctrl[0])
        return bool(np.all(self.well_flux_residual < self.tol_wells) and self.well_ctrl_residual < self.tol_ctrl)

    def nonlinearIteration(self, iteration, single_precision=None, nonlinear_solver=None):
        m = self.m
        m.setSolvePrecision(single_precision)
        m.assemble(iteration == 0)                      # reservoir + wells, connection pressures at iteration 0
        converged = m.getConvergence()
        converged = self.wellConvergence() and converged
        lin = 0
        if not converged or iteration < 1:
            m.solveJacobianSystem(single_precision=single_precision)
            lin = self.linear_iterations = m.linear_iterations
            m.updateState()                              # also recovers and updates (q_s, bhp) on the device
        return converged, lin


def five_spot(grid, rate_m3_per_day=500.0, bhp_prod_bar=150.0, wi=None):
    """SURVEY 8d synthetic wells: one water injector (rate controlled, full column) in the centre and four
    BHP-controlled producers in the corners of a Cartesian grid.  Peaceman-like WI from the cell transmissibility scale."""
    nx, ny, nz = grid.dims
    wells = Wells()
    col = lambda i, j: [i + nx * j + nx * ny * k for k in range(nz)]
    WI = wi if wi is not None else 10.0 * float(np.median(grid.trans))
    z = grid.z
    wells.add_well("INJ", INJECTOR, z[col(nx // 2, ny // 2)[0]], col(nx // 2, ny // 2), WI, (1.0, 0.0, 0.0),
                   (SURFACE_RATE, rate_m3_per_day / 86400.0, (1.0, 0.0, 0.0)))
    for k, (i, j) in enumerate([(0, 0), (nx - 1, 0), (0, ny - 1), (nx - 1, ny - 1)]):
        wells.add_well("PROD%d" % k, PRODUCER, z[col(i, j)[0]], col(i, j), WI, (0.0, 1.0, 0.0), (BHP, bhp_prod_bar * 1e5))
    return wells


def column_wells(grid, n_wells, n_injectors=1, seed=0, inj_layers=None, prod_layers=None, inj_rate_m3_per_day=800.0,
                 prod_bhp_bar=150.0, prod_oil_rate_m3_per_day=None, wi=None):
    """Vertical wells in distinct random (i, j) columns of a Cartesian deck with inactive cells (SPE9-like: 1 rate-controlled water
    injector + 25 producers; Norne-like: 36 wells through whatever is active in their column).  A well perforates the ACTIVE cells of
    its column inside the layer range; columns without an active cell there are skipped.  Producers alternate between BHP control and
    (if `prod_oil_rate_m3_per_day` is given) oil SURFACE_RATE control, so both control equations occur."""
    nx, ny, nz = grid.dims
    act = np.asarray(grid.active_index)
    rng = np.random.Generator(np.random.PCG64(seed))
    WI = wi if wi is not None else 10.0 * float(np.median(grid.trans))
    wells = Wells()
    for colidx in rng.permutation(nx * ny):
        if wells.nw == n_wells:
            break
        w = wells.nw
        inj = w < n_injectors
        layers = (inj_layers if inj else prod_layers) or range(nz)
        cells = [int(act[colidx + nx * ny * k]) for k in layers if act[colidx + nx * ny * k] >= 0]
        if not cells:
            continue
        zref = grid.z[cells[0]]
        if inj:
            wells.add_well("INJ%d" % w, INJECTOR, zref, cells, WI, (1.0, 0.0, 0.0), (SURFACE_RATE, inj_rate_m3_per_day / 86400.0, (1.0, 0.0, 0.0)))
        elif prod_oil_rate_m3_per_day is not None and w % 2 == 0:
            wells.add_well("PROD%d" % w, PRODUCER, zref, cells, WI, (0.0, 1.0, 0.0), (SURFACE_RATE, -prod_oil_rate_m3_per_day / 86400.0, (0.0, 1.0, 0.0)))
        else:
            wells.add_well("PROD%d" % w, PRODUCER, zref, cells, WI, (0.0, 1.0, 0.0), (BHP, prod_bhp_bar * 1e5))
    if wells.nw < n_wells:
        raise ValueError("not enough columns with active cells for %d wells" % n_wells)
    return wells
