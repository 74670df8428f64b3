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

  updateWellControls / updateWellStateWithTarget   StandardWells_impl.hpp:709-800 / :1452-1550 (+ wellhelpers::constraintBroken)
  solveWellEq (explicit well pre-solve at the initial assembly, default on)   BlackoilModelBase_impl.hpp:1018-1133
  THP control through VFP tables (opmgpu/vfp.py)   StandardWells_impl.hpp:655-700, :895-960
  computePropertiesForWellConnectionPressures (PVT at the average well-block pressure)   :218-296

  RESERVOIR_RATE controls: the control equation is the weighted rate sum with the control's distribution (:938-955); the distribution
  comes from opmgpu/rateconverter.py (RateConverter::SurfaceToReservoirVoidage + SimulatorBase::computeRESV)

Not restated: group controls / guide rates (WellCollection), efficiency factors.
"""
import numpy as np

INJECTOR, PRODUCER = 0, 1
BHP, SURFACE_RATE, THP, RESERVOIR_RATE = 0, 1, 2, 3          # WellControlType (opm-core well_controls.h), our numbering
GRAVITY = 9.80665


def _ctrl(c):
    """(type, target[, distr[, vfp_table_id[, alq]]]) -> normalised tuple"""
    distr = np.asarray(c[2] if len(c) > 2 and c[2] is not None else (0, 0, 0), float)
    return (int(c[0]), float(c[1]), distr, int(c[3]) if len(c) > 3 else 0, float(c[4]) if len(c) > 4 else 0.0)


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
        self.ctrl_type, self.ctrl_target, self.ctrl_distr = [], [], []        # the initial CURRENT control of each well (controls[w][current0[w]])
        self.current0 = []
        self.controls = []                                                     # all controls of each well (WellControls), control 0 first
        self.name = []

    def add_well(self, name, wtype, depth_ref, cells, WI, comp_frac, control, allow_cf=True, limits=(), current=0):
        """The well's controls are [control] + limits IN THAT ORDER; `current` is the index of its initial current control in that list
        (default: `control`), the others are inequality constraints that updateWellControls switches to when broken (e.g. a BHP limit
        on a rate-controlled well).  The order matters: updateWellControls switches to the FIRST broken constraint
        (StandardWells_impl.hpp:709-780), and WellsManager keeps a fixed order (ORAT, WRAT, GRAT, LRAT, RESV, BHP, THP) with the current
        control as an index into it -- opmgpu/schedule.py builds its wells that way."""
        self.name.append(name); self.type.append(wtype); self.depth_ref.append(float(depth_ref))
        self.comp_frac.append(np.asarray(comp_frac, float)); self.allow_cf.append(bool(allow_cf))
        self.cells += [int(c) for c in cells]; self.WI += [float(w) for w in np.broadcast_to(WI, (len(cells),))]
        self.connpos.append(len(self.cells))
        ctrls = [_ctrl(control)] + [_ctrl(c) for c in limits]
        if not 0 <= current < len(ctrls):
            raise ValueError("well %s: current control %d out of range" % (name, current))
        cur = ctrls[current]
        self.ctrl_type.append(cur[0]); self.ctrl_target.append(float(cur[1]))
        self.ctrl_distr.append(np.asarray(cur[2], float))
        self.controls.append(ctrls)
        self.current0.append(int(current))
        return self

    def control_arrays(self):
        """flat WellControls arrays for the device: ctrl_ptr[nw+1], type, target, distr[n*3], vfp id, alq"""
        ptr, typ, tgt, dis, vfp, alq = [0], [], [], [], [], []
        for cl in self.controls:
            for c in cl:
                typ.append(c[0]); tgt.append(c[1]); dis.append(c[2]); vfp.append(c[3]); alq.append(c[4])
            ptr.append(len(typ))
        return (np.asarray(ptr, np.int32), np.asarray(typ, np.int32), np.asarray(tgt, float), np.asarray(dis, float).reshape(-1, 3),
                np.asarray(vfp, np.int32), np.asarray(alq, float))

    @property
    def nw(self):
        return len(self.type)

    @property
    def nperf(self):
        return len(self.cells)

    def arrays(self):
        return np.asarray(self.connpos, np.int32), np.asarray(self.cells, np.int32)


class WellState:
    """WellStateFullyImplicitBlackoil fields used here: bhp, wellRates (well-major), perfPress, perfPhaseRates, currentControls, thp.
    Initial values as WellState::init / WellStateFullyImplicitBlackoil::init set them (opm-core / opm-simulators, external: restated
    from the published code): a BHP-controlled well starts at its target with zero rates, a rate-controlled one at its rate target
    with bhp = 1.01 / 0.99 x the first perforated cell's pressure; perfPress = the perforated cells' pressures, perfPhaseRates =
    the well rates divided evenly over the perforations."""

    def __init__(self, wells, cell_pressure):
        nw = wells.nw
        self.bhp, self.qs = np.zeros(nw), np.zeros((nw, 3))
        self.perf_press = np.zeros(wells.nperf); self.perf_rates = np.zeros((wells.nperf, 3))
        self.current = np.asarray(getattr(wells, "current0", None) or np.zeros(nw), np.int32).copy()          # currentControls(): index into the well's controls
        self.thp = np.zeros(nw)
        for w in range(nw):
            p0 = cell_pressure[wells.cells[wells.connpos[w]]]
            if wells.ctrl_type[w] == BHP:
                self.bhp[w] = wells.ctrl_target[w]
            else:       # rate control: start near the first perforated cell's pressure, rate at target
                self.bhp[w] = p0 * (1.01 if wells.type[w] == INJECTOR else 0.99)
                d = wells.ctrl_distr[w]
                self.qs[w] = wells.ctrl_target[w] * d / max(d.sum(), 1e-300) if wells.type[w] == PRODUCER else wells.ctrl_target[w] * wells.comp_frac[w]
            lo, hi = wells.connpos[w], wells.connpos[w + 1]
            self.perf_press[lo:hi] = np.asarray(cell_pressure)[np.asarray(wells.cells[lo:hi], int)]
            self.perf_rates[lo:hi] = self.qs[w] / max(hi - lo, 1)

    def copy(self):
        import copy
        return copy.deepcopy(self)

    def assign(self, other):
        """state = last_state (AdaptiveTimeStepping_impl.hpp:346-347), in place so that holders of this object see it"""
        self.bhp[:], self.qs[:], self.perf_press[:], self.perf_rates[:] = other.bhp, other.qs, other.perf_press, other.perf_rates
        self.current[:], self.thp[:] = other.current, other.thp


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
                 tolerance_wells=1e-4, tolerance_well_control=1e-7, vfp_tables=(), solve_welleq_initially=True):
        self.w, self.gravity, self.dbhp_max_rel = wells, gravity, dbhp_max_rel
        self.z_perf = np.asarray(z_cells, float)[np.asarray(wells.cells, int)]
        self.surf_dens = np.asarray(surface_density_wog, float).reshape(1, 3)
        self.tol_wells, self.tol_ctrl = tolerance_wells, tolerance_well_control
        self.solve_welleq_initially = solve_welleq_initially          # BlackoilModelParameters.cpp:96
        self.cdp = np.zeros(wells.nperf)
        self.perf_dens = np.zeros(wells.nperf)                        # well_perforation_densities_
        self.vfp_prod = {t.id: t for t in vfp_tables if not t.is_injector}
        self.vfp_inj = {t.id: t for t in vfp_tables if t.is_injector}
        self.vfp_active = any(c[0] == THP for cl in wells.controls for c in cl)      # isVFPActive (BlackoilModelBase_impl.hpp:982-1008)
        self.well_iterations = 0
        self._sys = None

    # computeWellConnectionPressures (StandardWells_impl.hpp:336-358): once per time step from the explicit state
    # (BlackoilModelBase_impl.hpp:797-805), again after a converged solveWellEq (:1122), and before updateWellControls when VFP
    # tables are active (:771-776).  `pvt_at(avg_press)` -> (b[nperf,3], rsmax, rvmax) evaluates the perforated cells' PVT at the
    # AVERAGE WELL-BLOCK PRESSURE with the cells' rs / rv / phase condition (computePropertiesForWellConnectionPressures, :218-296);
    # without it the cells' own values from `pp` are used.
    def compute_connection_pressures(self, pp, ws, pvt_at=None):
        W = self.w
        if pvt_at is not None:
            avg = np.zeros(W.nperf)
            for w in range(W.nw):
                for perf in range(W.connpos[w], W.connpos[w + 1]):
                    p_above = ws.bhp[w] if perf == W.connpos[w] else ws.perf_press[perf - 1]
                    avg[perf] = (ws.perf_press[perf] + p_above) / 2
            b, rsmax, rvmax = pvt_at(avg)
        else:
            b, rsmax, rvmax = pp[:, 3:6, 0], pp[:, 1, 0], pp[:, 2, 0]
        self.perf_dens = connection_densities(W, ws.perf_rates, b, rsmax, rvmax, np.repeat(self.surf_dens, W.nperf, 0))
        self.cdp = connection_pressure_delta(W, self.z_perf, self.perf_dens, self.gravity)

    # ---- THP / VFP helpers ------------------------------------------------------------------------------------------------
    def _vfp(self, w, table_id):
        tabs = self.vfp_inj if self.w.type[w] == INJECTOR else self.vfp_prod
        if table_id not in tabs:
            raise ValueError("well %s: VFP table %d does not exist" % (self.w.name[w], table_id))
        return tabs[table_id]

    def _vfp_dp(self, w, table):
        """wellhelpers::computeHydrostaticCorrection with the density of the well's FIRST perforation"""
        W = self.w
        from .vfp import hydrostatic_correction
        return hydrostatic_correction(W.depth_ref[w], table.datum_depth, self.perf_dens[W.connpos[w]], self.gravity, W.connpos[w + 1] > W.connpos[w])

    def _bhp_from_thp(self, w, ctrl, qs):
        """bhp the THP control asks for at the rates qs = (aqua, liquid, vapour): value (hydrostatic correction applied) and d/dqs"""
        table = self._vfp(w, ctrl[3])
        v, dq = table.bhp_dq(qs[0], qs[1], qs[2], ctrl[1], ctrl[4])
        return v - self._vfp_dp(w, table), dq

    # ---- updateWellControls (StandardWells_impl.hpp:709-800) ---------------------------------------------------------------
    def _update_well_state_with_target(self, w, current, ws):
        """updateWellStateWithTarget (:1452-1550): targets become the initial guesses of the well unknowns"""
        W = self.w
        typ, target, distr = W.controls[w][current][:3]
        if typ == BHP:
            ws.bhp[w] = target
        elif typ == THP:
            ws.bhp[w] = self._bhp_from_thp(w, W.controls[w][current], ws.qs[w])[0]
        elif typ == SURFACE_RATE:
            if W.type[w] == INJECTOR:
                for a in range(3):
                    if W.comp_frac[w][a] > 0.0:
                        ws.qs[w, a] = target * W.comp_frac[w][a]
            else:       # only single-phase rate targets (orat / wrat / grat, not lrat) seed the rates
                if int((distr > 0.0).sum()) < 2:
                    for a in range(3):
                        if distr[a] > 0.0:
                            ws.qs[w, a] = target * distr[a]
        # RESERVOIR_RATE: nothing (existing rates stay)

    def _constraint_broken(self, w, ctrl, ws):
        """wellhelpers::constraintBroken (opm-simulators WellHelpers.hpp, external): injectors break a limit from above,
        producers from below (their rates are negative)"""
        typ, target, distr = ctrl[:3]
        val = ws.bhp[w] if typ == BHP else (ws.thp[w] if typ == THP else float(np.dot(ws.qs[w], distr)))
        return val > target if self.w.type[w] == INJECTOR else val < target

    def update_well_controls(self, ws):
        """for every well: apply the current control's target to the well state, then switch to the FIRST other control whose
        constraint is broken; repeat until none is (at most 2 * ncontrols rounds, then NumericalIssue like the reference)."""
        from .model import NumericalIssue
        W = self.w
        switched = []
        for w in range(W.nw):
            ctrls = W.controls[w]
            current, rounds = int(ws.current[w]), 0
            while True:
                self._update_well_state_with_target(w, current, ws)
                broken = next((k for k in range(len(ctrls)) if k != current and self._constraint_broken(w, ctrls[k], ws)), None)
                if broken is not None:
                    switched.append((w, current, broken))
                    ws.current[w] = current = broken
                rounds += 1
                if rounds > 2 * len(ctrls):
                    raise NumericalIssue("Could not find proper control within %d iterations!" % rounds)
                if broken is None:
                    break
        return switched

    # ---- well equations ----------------------------------------------------------------------------------------------------
    def _well_system(self, w, pp, ws, const_cells=False):
        """computeWellFlux + addWellFluxEq + addWellControlEq of one well as dense AD: returns (cq_s[3] AD over [3n cell vars | qs(3) | bhp], E values (4), E Jacobian (4 x nv))"""
        W = self.w
        lo, hi = W.connpos[w], W.connpos[w + 1]
        n = hi - lo
        nv = 3 * n + 4
        I = np.arange(lo, hi)

        def perf_q(k):      # perf quantity k of OPMGPU_PERF_K as AD w.r.t. its own cell's (P, Sw, Xvar)
            j = np.zeros((n, nv))
            if not const_cells:
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
        ctrl = W.controls[w][int(ws.current[w])]
        if not alive:
            ce = qs[0] + qs[1] + qs[2]
        elif ctrl[0] == BHP:
            ce = bhp - ctrl[1]
        elif ctrl[0] == THP:        # bhp - bhp_from_thp(qs) + dp (:944-949)
            v, dq = self._bhp_from_thp(w, ctrl, ws.qs[w])
            jt = np.zeros((1, nv)); jt[0, 3 * n:3 * n + 3] = dq
            ce = bhp - AD(np.array([v]), jt)
        else:                       # SURFACE_RATE / RESERVOIR_RATE
            ce = ctrl[2][0] * qs[0] + ctrl[2][1] * qs[1] + ctrl[2][2] * qs[2] - ctrl[1]
        E.append(ce)
        Ev = np.array([e.v[0] for e in E]); Ej = np.vstack([e.j for e in E])
        return cq_s, Ev, Ej

    # computeWellPotentials (StandardWells_impl.hpp:1003-1095; called at the start of a report step when the deck has group controls,
    # BlackoilModelBase_impl.hpp:2576-2619): the surface rates each well would deliver at its MOST RESTRICTIVE bhp limit -- the target of a
    # BHP control, or the bhp a THP control implies at the well's current rates through its VFP table (hydrostatic correction applied):
    # the smallest such bhp for an injector, the largest for a producer -- with the cells' explicit state.  The search starts from 0 like
    # the reference's `Vector::Zero(nw)`, so a well without any BHP / THP control is evaluated at bhp = 0, and an injector's THP limit
    # only counts when it lies below a BHP target seen earlier in the list.  Returns [nw, 3] (sum of computeWellFlux's cq_s per well).
    def compute_well_potentials(self, pp, ws):
        W = self.w
        ws0 = ws.copy()
        for w in range(W.nw):
            bhp = 0.0
            for ctrl in W.controls[w]:
                if ctrl[0] == BHP:
                    bhp = ctrl[1]
                if ctrl[0] == THP:
                    v, _ = self._bhp_from_thp(w, ctrl, ws.qs[w])
                    if W.type[w] == INJECTOR:
                        if v < bhp:
                            bhp = v
                    elif v > bhp:
                        bhp = v
            ws0.bhp[w] = bhp
        pot = np.zeros((W.nw, 3))
        for w in range(W.nw):
            cq_s, _, _ = self._well_system(w, pp, ws0, const_cells=True)
            pot[w] = [cq_s[a].v.sum() for a in range(3)]
        return pot

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
            cq_s, Ev, Ej = self._well_system(w, pp, ws)
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

    # solveWellEq (BlackoilModelBase_impl.hpp:1018-1133): Newton on the well equations alone with the reservoir frozen
    def solve_well_eq(self, pp, ws, B_avg, pvt_at=None, max_it=15):
        """Returns (converged, iterations).  On convergence the well state keeps the solution and the connection pressures are
        recomputed from it; otherwise the well state is restored (:1124-1126)."""
        W = self.w
        ws0 = ws.copy()
        it, converged = 0, False
        while True:
            flux_eq = np.zeros((W.nw, 3)); ctrl_eq = np.zeros(W.nw)
            systems, cq_all = [], np.zeros((W.nperf, 3))
            for w in range(W.nw):
                n = W.connpos[w + 1] - W.connpos[w]
                cq_s, Ev, Ej = self._well_system(w, pp, ws, const_cells=True)
                systems.append((Ev, Ej[:, 3 * n:]))
                flux_eq[w] = Ev[:3]; ctrl_eq[w] = Ev[3]
                for a in range(3):
                    cq_all[W.connpos[w]:W.connpos[w + 1], a] = cq_s[a].v
            ws.perf_rates = cq_all                            # updatePerfPhaseRatesAndPressures (:1059)
            for w in range(W.nw):
                ws.perf_press[W.connpos[w]:W.connpos[w + 1]] = ws.bhp[w] + self.cdp[W.connpos[w]:W.connpos[w + 1]]
            self.flux_eq, self.ctrl_eq = flux_eq, ctrl_eq
            converged = self.converged(B_avg)                 # getWellConvergence (:1866-1950)
            if converged:
                break
            it += 1
            for w in range(W.nw):
                Ev, D = systems[w]
                self._apply_well_increment(w, np.linalg.solve(D, Ev), ws)
            self.update_well_controls(ws)
            if it >= max_it:
                break
        self.well_iterations = it
        if converged:
            self.compute_connection_pressures(pp, ws, pvt_at)
        else:
            ws.assign(ws0)
        return converged, it

    def converged(self, B_avg):
        """well part of getConvergence (BlackoilModelBase_impl.hpp:1769-1779)."""
        wf = np.asarray(B_avg) * np.abs(self.flux_eq).max(0)
        self.well_flux_residual, self.well_ctrl_residual = wf, np.abs(self.ctrl_eq).max()
        return bool(np.all(wf < self.tol_wells) and self.well_ctrl_residual < self.tol_ctrl)

    def _apply_well_increment(self, w, dy, ws):
        """updateWellState (StandardWells_impl.hpp:611-700) for one well: rates, limited bhp, thp of wells that have a THP control"""
        W = self.w
        ws.qs[w] -= dy[:3]
        d = dy[3]
        ws.bhp[w] -= np.sign(d) * min(abs(d), abs(ws.bhp[w]) * self.dbhp_max_rel)
        thp_ctrl = next((c for c in W.controls[w] if c[0] == THP), None)
        if thp_ctrl is not None:
            table = self._vfp(w, thp_ctrl[3])
            ws.thp[w] = table.thp_of(ws.qs[w, 0], ws.qs[w, 1], ws.qs[w, 2], ws.bhp[w] + self._vfp_dp(w, table), thp_ctrl[4])

    def recover(self, dx_perf):
        """recoverVariable (NewtonIterationUtilities.cpp:134-184): the wells' part of the Newton increment, [nw][4] = d qs (3), d bhp"""
        W = self.w
        dy = np.zeros((W.nw, 4))
        for w in range(W.nw):
            lo, hi = W.connpos[w], W.connpos[w + 1]
            Dinv, C, Ev = self._sys[w]
            dy[w] = Dinv @ (Ev - C @ np.asarray(dx_perf[lo:hi], float).ravel())
        return dy

    def recover_and_update(self, dx_perf, ws, dy=None):
        """recoverVariable + updateWellState (StandardWells_impl.hpp:611-700); `dy` = an already recovered (and relaxed) increment"""
        dy = self.recover(dx_perf) if dy is None else dy
        for w in range(self.w.nw):
            self._apply_well_increment(w, dy[w], ws)


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

    def _pvt_at(self):
        """PVT of the perforated cells at given pressures, if the backend offers it (opmgpu_perf_pvt)"""
        return getattr(self.m, "perfPvtAt", None)

    def assemble(self, initial):
        """BlackoilModelBase::assemble in the reference's order (BlackoilModelBase_impl.hpp:757-840): control switching, reservoir
        equations, [initial: connection pressures, explicit well pre-solve], well equations and their Schur-reduced terms"""
        m, wh, ws = self.m, self.wh, self.ws
        if wh.vfp_active:       # VFP tables need the connection densities for the hydrostatic correction (:771-776)
            wh.compute_connection_pressures(m.perfProps(self.nperf).reshape(self.nperf, 9, 4), ws, self._pvt_at())
        wh.update_well_controls(ws)                                   # :785
        m.assemble(initial)
        pp = m.perfProps(self.nperf).reshape(self.nperf, 9, 4)
        if initial:
            wh.compute_connection_pressures(pp, ws, self._pvt_at())    # :797-805
            if wh.solve_welleq_initially:                              # :827-829
                wh.solve_well_eq(pp, ws, m.averageB(), self._pvt_at())
        resid_delta, rc, blocks, rhs_delta = wh.assemble(pp, ws)
        m.addWellTerms(resid_delta, rc, blocks)
        m.addWellRhs(rhs_delta)

    def computeWellPotentials(self):
        """well potentials from the resident state (StandardWells::computeWellPotentials): [nw, 3]"""
        pp = self.m.perfProps(self.nperf).reshape(self.nperf, 9, 4)
        return self.wh.compute_well_potentials(pp, self.ws)

    def nonlinearIteration(self, iteration, single_precision=None, nonlinear_solver=None):
        """BlackoilModelBase::nonlinearIteration (BlackoilModelBase_impl.hpp:239-326)"""
        m, wh, ws = self.m, self.wh, self.ws
        ns = nonlinear_solver
        initial = iteration == 0
        if initial:
            self.residual_norms_history, self.current_relaxation, self.dy_old = [], 1.0, np.zeros((wh.w.nw, 4))
        self.assemble(initial)
        converged = m.getConvergence()
        converged = wh.converged(m.B_avg) and converged
        self.residual_norms_history.append(list(m.linf) + list(np.abs(wh.flux_eq).max(0)) + [np.abs(wh.ctrl_eq).max()])
        lin = 0
        if not converged or iteration < (ns.min_iter if ns else 1):
            m.solveJacobianSystem(single_precision=single_precision)
            lin = self.linear_iterations = m.linear_iterations
            dy = wh.recover(m.perfDx(self.nperf))
            if ns is not None and getattr(m, "use_update_stabilization", True):
                # detectOscillations + stabilizeNonlinearUpdate on the WHOLE increment, wells included (NonlinearSolver_impl.hpp:221-301)
                oscillate, _ = ns.detectOscillations(self.residual_norms_history, iteration)
                if oscillate:
                    self.current_relaxation = max(self.current_relaxation - ns.relax_increment, ns.relax_max)
                om = self.current_relaxation
                m.stabilizeUpdate(ns.relax_type, om)
                dy_new = dy.copy()
                if om != 1.0:
                    dy = om * dy + (1.0 - om) * self.dy_old if ns.relax_type == 1 else om * dy
                self.dy_old = dy_new
            wh.recover_and_update(None, ws, dy=dy)
            m.updateState()
        return converged, lin


class DeviceWellModel:
    """The same nonlinear iteration with the well model ON THE DEVICE (csrc/wells.hip): no per-iteration
    read-back of perforation properties, no clique fill -- opmgpu_set_device_wells / well_state_set / well_convergence.
    Interface of WellCoupledModel, so NonlinearSolver / AdaptiveTimeStepping drive either."""

    def __init__(self, backend, wells, well_state, tolerance_wells=None, tolerance_well_control=None, vfp_tables=()):
        import ctypes as C
        from . import capi
        self.m, self.w, self.ws = backend, wells, well_state
        # the tolerances of the context (BlackoilModelParameters: tolerance_wells 1e-4, tolerance_well_control 1e-7) unless given
        if tolerance_wells is None:
            tolerance_wells = backend.params.tolerance_wells
        if tolerance_well_control is None:
            tolerance_well_control = backend.params.tolerance_well_control
        self.tol_wells, self.tol_ctrl = tolerance_wells, tolerance_well_control
        self.linear_iterations = 0
        nw = wells.nw
        if vfp_tables:
            arr = (capi.VfpTable * len(vfp_tables))()
            self._keep_vfp = []
            for k, t in enumerate(vfp_tables):
                v = arr[k]
                v.id, v.is_injector, v.flo_type, v.datum_depth = t.id, int(t.is_injector), t.flo_type, t.datum_depth
                axes = [capi.f64(t.flo), capi.f64(t.thp)] + ([] if t.is_injector else [capi.f64(t.wfr), capi.f64(t.gfr), capi.f64(t.alq)])
                data = capi.f64(t.data).ravel()
                self._keep_vfp += axes + [data]
                v.nflo, v.nthp, v.flo, v.thp, v.data = axes[0].size, axes[1].size, capi.dptr(axes[0]), capi.dptr(axes[1]), capi.dptr(data)
                if not t.is_injector:
                    v.wfr_type, v.gfr_type = t.wfr_type, t.gfr_type
                    v.nwfr, v.ngfr, v.nalq = axes[2].size, axes[3].size, axes[4].size
                    v.wfr, v.gfr, v.alq = capi.dptr(axes[2]), capi.dptr(axes[3]), capi.dptr(axes[4])
            backend._chk(backend.lib.opmgpu_set_vfp_tables(backend.ctx, len(vfp_tables), arr))
        spec = capi.WellsSpec()
        cptr, ctyp, ctgt, cdis, cvfp, calq = wells.control_arrays()
        self._keep = [np.asarray(wells.connpos, np.int32), np.asarray(wells.cells, np.int32), capi.f64(wells.WI),
                      np.asarray(wells.type, np.int32), np.asarray([int(a) for a in wells.allow_cf], np.int32), capi.f64(wells.depth_ref),
                      capi.f64(np.asarray(wells.comp_frac, float).reshape(nw, 3)), capi.i32(ctyp), capi.f64(ctgt), capi.f64(cdis),
                      capi.i32(cptr), capi.i32(cvfp), capi.f64(calq)]
        k = self._keep
        spec.nw = nw
        spec.well_connpos, spec.well_cells, spec.WI, spec.type, spec.allow_cf = capi.iptr(k[0]), capi.iptr(k[1]), capi.dptr(k[2]), capi.iptr(k[3]), capi.iptr(k[4])
        spec.depth_ref, spec.comp_frac, spec.ctrl_type, spec.ctrl_target, spec.ctrl_distr = capi.dptr(k[5]), capi.dptr(k[6]), capi.iptr(k[7]), capi.dptr(k[8]), capi.dptr(k[9])
        spec.ctrl_ptr, spec.ctrl_vfp, spec.ctrl_alq = capi.iptr(k[10]), capi.iptr(k[11]), capi.dptr(k[12])
        backend._chk(backend.lib.opmgpu_set_device_wells(backend.ctx, C.byref(spec)))
        if tolerance_wells != backend.params.tolerance_wells or tolerance_well_control != backend.params.tolerance_well_control:
            pass        # the pre-solve on the device uses opmgpu_params.tolerance_wells / tolerance_well_control of the context
        self.push_well_state()

    def push_well_state(self):
        from . import capi
        m, ws = self.m, self.ws
        if self.w.nw == 0:          # a rank without wells (multi-GPU): nothing to upload, the convergence call stays collective
            return
        m._chk(m.lib.opmgpu_well_state_set(m.ctx, capi.dptr(capi.f64(ws.bhp)), capi.dptr(capi.f64(ws.qs)), capi.dptr(capi.f64(ws.perf_press)), capi.dptr(capi.f64(ws.perf_rates))))
        m._chk(m.lib.opmgpu_well_controls_set(m.ctx, capi.iptr(capi.i32(ws.current)), capi.dptr(capi.f64(ws.thp))))

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
        import ctypes as C
        cur, thp, its, conv = np.zeros(self.w.nw, np.int32), np.zeros(self.w.nw), C.c_int32(0), C.c_int32(0)
        m._chk(m.lib.opmgpu_well_controls_get(m.ctx, capi.iptr(cur), capi.dptr(thp), C.byref(its), C.byref(conv)))
        ws.current[:], ws.thp[:] = cur, thp
        self.presolve_iterations, self.presolve_converged = its.value, bool(conv.value)
        return ws

    def prepareStep(self, dt, state=None):
        self.m.prepareStep(dt, state)

    def saveState(self):
        self.m.saveState()

    def restoreState(self):
        self.m.restoreState()

    def relativeChange(self):
        return self.m.relativeChange()

    def computeWellPotentials(self, z_cells, surface_density_wog, vfp_tables=()):
        """StandardWells::computeWellPotentials with the wells on the device: a once-per-report-step host evaluation (as in the reference) from
        the device's well state and the perforated cells' properties; the connection pressure differences are the device's own
        (perfPress - bhp, StandardWells_impl.hpp:664-673)."""
        ws = self.pull_well_state()
        host = StandardWellsHost(self.w, z_cells, surface_density_wog, vfp_tables=vfp_tables)
        pp = self.m.perfProps(self.w.nperf).reshape(self.w.nperf, 9, 4)
        if host.vfp_active:
            host.compute_connection_pressures(pp, ws, getattr(self.m, "perfPvtAt", None))          # densities for the hydrostatic correction
        perf_well = np.repeat(np.arange(self.w.nw), np.diff(np.asarray(self.w.connpos)))
        host.cdp = np.asarray(ws.perf_press) - np.asarray(ws.bhp)[perf_well]
        return host.compute_well_potentials(pp, ws)

    def wellConvergence(self):
        from . import capi
        m = self.m
        flux, ctrl = np.zeros(3), np.zeros(1)
        m._chk(m.lib.opmgpu_well_convergence(m.ctx, capi.dptr(flux), capi.dptr(ctrl)))
        self.well_flux_residual, self.well_ctrl_residual = np.asarray(m.B_avg) * flux, float(ctrl[0])
        return bool(np.all(self.well_flux_residual < self.tol_wells) and self.well_ctrl_residual < self.tol_ctrl)

    def nonlinearIteration(self, iteration, single_precision=None, nonlinear_solver=None):
        m, ns = self.m, nonlinear_solver
        if getattr(m, "fused_iteration", False) and ns is not None and (self.tol_wells, self.tol_ctrl) == (m.params.tolerance_wells, m.params.tolerance_well_control):
            converged, lin = m._fused_iteration(iteration, single_precision, ns)          # the library checks the wells' convergence itself
            self.linear_iterations, self.current_relaxation = lin, m.current_relaxation
            return converged, lin
        if iteration == 0:
            self.residual_norms_history, self.current_relaxation = [], 1.0
        m.setSolvePrecision(single_precision)
        m.assemble(iteration == 0)                      # reservoir + wells: control switching, connection pressures + pre-solve at iteration 0
        converged = m.getConvergence()
        converged = self.wellConvergence() and converged
        self.residual_norms_history.append(list(m.linf))
        lin = 0
        if not converged or iteration < (ns.min_iter if ns else 1):
            m.solveJacobianSystem(single_precision=single_precision)
            lin = self.linear_iterations = m.linear_iterations
            if ns is not None and getattr(m, "use_update_stabilization", True):
                oscillate, _ = ns.detectOscillations(self.residual_norms_history, iteration)
                if oscillate:
                    self.current_relaxation = max(self.current_relaxation - ns.relax_increment, ns.relax_max)
                m.stabilizeUpdate(ns.relax_type, self.current_relaxation)       # relaxes the well part of the increment too
            m.updateState()                              # also updates (q_s, bhp, thp) on the device
        return converged, lin


def five_spot(grid, rate_m3_per_day=500.0, bhp_prod_bar=150.0, wi=None, slabs=1, slab_axis=2):
    """SURVEY 8d synthetic wells: one water injector (rate controlled, full column) in the centre and four
    BHP-controlled producers in the corners of a Cartesian grid.  Peaceman-like WI from the cell transmissibility scale.
    slabs > 1 (weak-scaling decks stacked along k): one such 5-spot per slab of nz / slabs layers, every well inside its slab (a well lives
    on one rank); the producers' BHP follows the hydrostatic pressure of the slab's top (700 kg/m3, the gradient decks.initial_state
    uses) so that every slab sees the drawdown of the first."""
    nx, ny, nz = grid.dims
    wells = Wells()
    WI = wi if wi is not None else 10.0 * float(np.median(grid.trans))
    z = grid.z
    if slabs > 1 and slab_axis == 1:
        # copies of the deck side by side along j: one 5-spot of full columns per slab of ny / slabs rows (all at the same depth)
        per = ny // slabs
        for s in range(slabs):
            j0, j1 = s * per, (ny if s == slabs - 1 else (s + 1) * per)
            col = lambda i, j: [i + nx * j + nx * ny * k for k in range(nz)]      # noqa: E731
            tag = "_S%d" % s
            wells.add_well("INJ" + tag, INJECTOR, z[col(nx // 2, (j0 + j1) // 2)[0]], col(nx // 2, (j0 + j1) // 2), WI, (1.0, 0.0, 0.0),
                           (SURFACE_RATE, rate_m3_per_day / 86400.0, (1.0, 0.0, 0.0)))
            for k, (i, j) in enumerate([(0, j0), (nx - 1, j0), (0, j1 - 1), (nx - 1, j1 - 1)]):
                wells.add_well("PROD%d%s" % (k, tag), PRODUCER, z[col(i, j)[0]], col(i, j), WI, (0.0, 1.0, 0.0), (BHP, bhp_prod_bar * 1e5))
        return wells
    per = nz // slabs
    for s in range(slabs):
        k0, k1 = s * per, (nz if s == slabs - 1 else (s + 1) * per)
        col = lambda i, j: [i + nx * j + nx * ny * k for k in range(k0, k1)]      # noqa: E731
        tag = "" if slabs == 1 else "_S%d" % s
        dbhp = 700.0 * grid.gravity * (z[col(0, 0)[0]] - z[0])
        wells.add_well("INJ" + tag, INJECTOR, z[col(nx // 2, ny // 2)[0]], col(nx // 2, ny // 2), WI, (1.0, 0.0, 0.0),
                       (SURFACE_RATE, rate_m3_per_day / 86400.0, (1.0, 0.0, 0.0)))
        for k, (i, j) in enumerate([(0, 0), (nx - 1, 0), (0, ny - 1), (nx - 1, ny - 1)]):
            wells.add_well("PROD%d%s" % (k, tag), PRODUCER, z[col(i, j)[0]], col(i, j), WI, (0.0, 1.0, 0.0), (BHP, bhp_prod_bar * 1e5 + dbhp))
    return wells


def column_wells(grid, n_wells, n_injectors=1, seed=0, inj_layers=None, prod_layers=None, inj_rate_m3_per_day=800.0,
                 prod_bhp_bar=150.0, prod_oil_rate_m3_per_day=None, wi=None, rate_wells_bhp_limits_bar=None):
    """Vertical wells in distinct random (i, j) columns of a Cartesian deck with inactive cells (SPE9-like: 1 rate-controlled water
    injector + 25 producers; Norne-like: 36 wells through whatever is active in their column).  A well perforates the ACTIVE cells of
    its column inside the layer range; columns without an active cell there are skipped.  Producers alternate between BHP control and
    (if `prod_oil_rate_m3_per_day` is given) oil SURFACE_RATE control, so both control equations occur.  rate_wells_bhp_limits_bar =
    (injectors' upper, producers' lower BHP limit): what every real deck gives its rate-controlled wells; without it (the BASELINE-like
    decks of baseline_decks.py, kept as they were measured) a rate target a pocket cannot deliver drives the well's bhp to zero and the well
    equations stop converging at any step length (tools/robust_sweep.py, deck 6057)."""
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
            wells.add_well("INJ%d" % w, INJECTOR, zref, cells, WI, (1.0, 0.0, 0.0), (SURFACE_RATE, inj_rate_m3_per_day / 86400.0, (1.0, 0.0, 0.0)),
                           limits=[] if rate_wells_bhp_limits_bar is None else [(BHP, rate_wells_bhp_limits_bar[0] * 1e5)])
        elif prod_oil_rate_m3_per_day is not None and w % 2 == 0:
            wells.add_well("PROD%d" % w, PRODUCER, zref, cells, WI, (0.0, 1.0, 0.0), (SURFACE_RATE, -prod_oil_rate_m3_per_day / 86400.0, (0.0, 1.0, 0.0)),
                           limits=[] if rate_wells_bhp_limits_bar is None else [(BHP, rate_wells_bhp_limits_bar[1] * 1e5)])
        else:
            wells.add_well("PROD%d" % w, PRODUCER, zref, cells, WI, (0.0, 1.0, 0.0), (BHP, prod_bhp_bar * 1e5))
    if wells.nw < n_wells:
        raise ValueError("not enough columns with active cells for %d wells" % n_wells)
    return wells
