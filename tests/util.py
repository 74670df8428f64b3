"""Shared helpers for the parity tests."""
import numpy as np
import scipy.sparse as sp


def bsr_to_scipy(rowptr, col, val9):
    nb = rowptr.size - 1
    return sp.bsr_matrix((np.asarray(val9).reshape(-1, 3, 3), col, rowptr), shape=(3 * nb, 3 * nb)).tocsr()


def eqmajor_to_interleaved(v, nc):
    return np.ascontiguousarray(np.asarray(v).reshape(3, nc).T).ravel()


def interleaved_to_eqmajor(v, nc):
    return np.ascontiguousarray(np.asarray(v).reshape(nc, 3).T).ravel()


def random_block_matrix(rowptr, col, seed=1, dominance=6.0):
    """SURVEY 8d 'SpMV micro': blocks = I*(dominance+u) on the diagonal, -0.3*u off it."""
    rng = np.random.default_rng(seed)
    nb = rowptr.size - 1
    val = -0.3 * rng.random((col.size, 9))
    rows = np.repeat(np.arange(nb), np.diff(rowptr))
    d = np.flatnonzero(rows == col)
    val[d] = 0.3 * rng.random((d.size, 9))
    val[d, 0] += dominance + rng.random(d.size); val[d, 4] += dominance + rng.random(d.size); val[d, 8] += dominance + rng.random(d.size)
    return val


def rel_err(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-300)


class OracleBackend:
    """The GpuBlackoilModel interface on top of the CPU oracle (checker side of the well-coupled parity tests)."""

    def __init__(self, orc, grid, tables, params, wells=None):
        from opmgpu import capi
        self.orc, self.grid, self.tab, self.prm = orc, grid, tables, params
        self.scale = np.asarray(params.matbalscale[:])
        self.wells = wells
        self.rowptr, self.col = orc.pattern(grid, *(wells or (None, None)))
        self.nc = grid.nc
        self.position = None
        self.linear_iterations = 0
        self.capi = capi

    def prepareStep(self, dt, state=None):
        self.dt = float(dt)
        if state is not None:
            self.st = state.copy()
        self.acc0 = None

    def assemble(self, initial):
        if initial:
            self.acc0 = None
            self.dx_old = None
        self.r, self.val, self.acc0, self.binv = self.orc.assemble(self.grid, self.tab, self.dt, self.st, self.rowptr, self.col,
                                                                   scale=tuple(self.scale), accum0=self.acc0)
        self.rhs_extra = np.zeros(3 * self.nc)

    def computeFluidInPlace(self, fipnum=None, cells=False):
        """numpy restatement of BlackoilModelBase::computeFluidInPlace (BlackoilModelBase_impl.hpp:2263-2366) on the oracle's cell properties"""
        import numpy as np
        st, nc = self.st, self.grid.nc
        props = self.orc.cell_props(self.grid, self.tab, st)
        nm = self.orc.PROP_NAMES
        assert self.tab.rocktab_n == 0
        cp = self.tab.rock_comp * (st.p - self.tab.rock_pref)
        pvm = 1.0 + cp + 0.5 * cp * cp
        pv = np.asarray(self.grid.pv)
        fip = np.zeros((7, nc))
        for a, ph in enumerate("wog"):
            fip[a] = ((pvm * props[:, nm.index("b_" + ph), 0]) * st.sat[:, a]) * pv
        fip[3], fip[4] = st.rs * fip[1], st.rv * fip[2]
        fn = np.ones(nc, np.int32) if fipnum is None else np.asarray(fipnum, np.int32)
        dims = max(1, int(fn.max()))
        values, hcpv, pres = np.zeros((dims, 7)), np.zeros(dims), np.zeros(dims)
        hyd = st.sat[:, 1] + st.sat[:, 2]
        for c in range(nc):
            r = fn[c] - 1
            if r != -1:
                values[r, :5] += fip[:5, c]; hcpv[r] += pv[c] * hyd[c]; pres[r] += pv[c] * st.p[c]
        for c in range(nc):
            r = fn[c] - 1
            if r != -1:
                fip[5, c] = pv[c]
                fip[6, c] = pv[c] * st.p[c] * hyd[c] / hcpv[r] if hcpv[r] != 0 else pres[r] / pv[c]
                values[r, 5] += fip[5, c]; values[r, 6] += fip[6, c]
        return (values, fip) if cells else values

    def perfProps(self, nperf):
        props = self.orc.cell_props(self.grid, self.tab, self.st)[self.wells[1]]
        names = self.orc.PROP_NAMES
        idx = [names.index(n) for n in ["p_o", "rs", "rv", "b_w", "b_o", "b_g", "mob_w", "mob_o", "mob_g"]]
        return props[:, idx, :].reshape(nperf, 36)

    def averageB(self):
        """B_avg of getWellConvergence (BlackoilModelBase_impl.hpp:1876-1891): mean of 1/b per phase over the cells"""
        return self.binv.reshape(3, self.nc).mean(1)

    def perfPvtAt(self, press):
        """computePropertiesForWellConnectionPressures (StandardWells_impl.hpp:218-296): b_w, b_o, b_g, rsSat, rvSat of the
        perforated cells at the given pressures, with the cells' rs / rv / phase condition / oil saturation"""
        from opmgpu import capi
        cells = np.asarray(self.wells[1])
        st, t = self.st, self.tab
        pvtnum = None if self.grid.pvtnum is None else self.grid.pvtnum[cells]
        hc = st.hc[cells]
        free_gas = (hc != capi.HC_OIL_ONLY).astype(np.int8); free_oil = (hc != capi.HC_GAS_ONLY).astype(np.int8)
        if not t.has_disgas:
            free_gas[:] = 1
        if not t.has_vapoil:
            free_oil[:] = 1
        b = np.stack([self.orc.pvt(t, "bWat", press, pvtnum=pvtnum)[:, 0],
                      self.orc.pvt(t, "bOil", press, st.rs[cells], free_gas, pvtnum)[:, 0],
                      self.orc.pvt(t, "bGas", press, st.rv[cells], free_oil, pvtnum)[:, 0]], 1)
        rsmax = self.orc.pvt(t, "rsSat", press, pvtnum=pvtnum)[:, 0] if t.has_disgas else np.zeros(cells.size)
        rvmax = self.orc.pvt(t, "rvSat", press, pvtnum=pvtnum)[:, 0] if t.has_vapoil else np.zeros(cells.size)
        so, somax = st.sat[cells, 1], getattr(self, "so_max", np.zeros(self.nc))[cells]
        for vap, arr in ((t.vap2, rsmax), (t.vap1, rvmax)):          # applyVap (BlackoilPropsAdFromDeck.cpp:1052-1078)
            if vap > 0.0:
                k = (somax > 0.01) & (so < somax)
                arr[k] *= (np.maximum(so[k], 1.4901161193847656e-08) / somax[k]) ** vap
        return b, rsmax, rvmax

    def stabilizeUpdate(self, relax_type, omega):
        """NonlinearSolver::stabilizeNonlinearUpdate (NonlinearSolver_impl.hpp:260-301) on the reservoir part of dx"""
        old = getattr(self, "dx_old", None)
        if old is None or old.size != self.dx.size:
            old = np.zeros_like(self.dx)
        new = self.dx.copy()
        if omega != 1.0:
            self.dx = omega * self.dx + (1.0 - omega) * old if relax_type == 1 else omega * self.dx
        self.dx_old = new

    def addWellTerms(self, resid_delta, rc, blocks):
        nc = self.nc
        cells = np.asarray(self.wells[1])
        for a in range(3):
            np.add.at(self.r, a * nc + cells, np.asarray(resid_delta)[:, a])
        sc = np.repeat(self.scale, 3)
        rc = np.asarray(rc).reshape(-1, 2)
        # position of every (row, col) block: binary search in the row-major key list of the pattern (rows and columns ascending)
        if getattr(self, "_keys", None) is None:
            rows = np.repeat(np.arange(nc, dtype=np.int64), np.diff(self.rowptr))
            self._keys = rows * nc + self.col
        pos = np.searchsorted(self._keys, rc[:, 0].astype(np.int64) * nc + rc[:, 1])
        assert np.array_equal(self._keys[pos], rc[:, 0].astype(np.int64) * nc + rc[:, 1])
        np.add.at(self.val, pos, np.asarray(blocks) * sc)

    def addWellRhs(self, rhs_delta):
        nc = self.nc
        cells = np.asarray(self.wells[1])
        for a in range(3):
            np.add.at(self.rhs_extra, a * nc + cells, np.asarray(rhs_delta)[:, a])

    def getConvergence(self):
        st, self.B_avg, self.CNV, self.MB, self.linf, conv = self.orc.convergence(self.grid, self.prm, self.dt, self.r, self.binv)
        assert st == 0
        return conv

    def solveJacobianSystem(self, want_dx=False, single_precision=False):
        nc = self.nc
        b = np.ascontiguousarray(((self.r + self.rhs_extra) * np.repeat(self.scale, nc)).reshape(3, nc).T).ravel()
        sto, x, it, red, _ = self.orc.bicgstab(self.rowptr, self.col, self.val, b, self.prm, position=self.position, single=bool(single_precision))
        assert sto == 0, sto
        self.linear_iterations = it
        self.dx = np.ascontiguousarray(x.reshape(nc, 3).T).ravel()
        return self.dx

    def perfDx(self, nperf):
        nc = self.nc
        return np.stack([self.dx[a * nc + self.wells[1]] for a in range(3)], 1)

    def updateState(self):
        self.st = self.orc.update_state(self.grid, self.tab, self.prm, self.dx, self.st)

    def getState(self):
        return self.st

    # last_state of AdaptiveTimeStepping and BlackoilModelBase::relativeChange (BlackoilModelBase_impl.hpp:1595-1631)
    def saveState(self):
        self.saved = self.st.copy()

    def restoreState(self):
        self.st = self.saved.copy()

    def relativeChange(self):
        a, b = self.saved, self.st
        num = ((a.p - b.p) ** 2).sum() + ((a.sat - b.sat) ** 2).sum()
        den = (b.p ** 2).sum() + (b.sat ** 2).sum()
        return num / den if den > 0 else 0.0
