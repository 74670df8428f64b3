"""BASELINE.json configurations at FULL size on the GPU against the oracle (which assembles a 1 M-cell system in about a second
and solves it in seconds, so it IS the checker at these sizes): residual, Jacobian / coupled operator, convergence scalars and the
post-update reservoir + well state of the first Newton iterations, then the Newton iteration count of a whole time step at the
reference's default tolerances.  Decks: 100x100x100 with the 5-spot (device wells), 60x220x85 with sigma_lnK = 2.5 and a 5-spot,
46x112x22 with 60 % inactive cells + NNCs + threshold pressures + 36 wells, 24x25x15 with 26 wells.  The size-independent
property checks and the small -like cases of round 1 stay below them."""
import numpy as np
import pytest

from opmgpu import capi, decks
from opmgpu.model import GpuBlackoilModel, GpuNewtonIteration
from util import rel_err

pytestmark = pytest.mark.gpu


def test_cart100_full_size_properties(gpu_lib):
    """configs[2]: synthetic 100x100x100, 1 M cells, 6.94 M blocks."""
    grid = decks.cartesian_grid(100, 100, 100, lognormal_sigma=0.5, seed=12345)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, perturb=0.002, seed=12345)
    prm = capi.default_params()
    nc = grid.nc
    m = GpuBlackoilModel(grid, tab, prm)
    pos, lev, nlev = m.ordering()
    assert nlev == 2 and np.bincount(lev).tolist() == [500000, 500000]          # red-black on the 7-point grid
    dt = 5 * decks.DAY
    m.prepareStep(dt, st)
    m.assemble(True)
    r = m.residual()
    # mass conservation of the flux part: with accum1 == accum0 every interior flux enters two rows with opposite signs
    for a in range(3):
        ra = r[a * nc:(a + 1) * nc]
        assert abs(ra.sum()) <= 1e-9 * np.abs(ra).sum()
    conv = m.getConvergence()
    assert not conv and np.all(np.isfinite(m.CNV)) and np.all(m.MB < 1e-12)
    # solve, then verify the TRUE residual of the returned increment with an independent SpMV on the same matrix
    rowptr, col, val = m.jacobian()
    assert col.size == 6940000
    scale = np.asarray(prm.matbalscale[:])
    for single in (True, False):
        dx = m.solveJacobianSystem(want_dx=True, single_precision=single)
        assert 1 <= m.linear_iterations <= 150 and m.linear_reduction < 1e-2
        s = GpuNewtonIteration(prm)
        s.load(rowptr, col, val, False)
        x3 = np.ascontiguousarray(dx.reshape(3, nc).T).ravel()
        b3 = np.ascontiguousarray((r * np.repeat(scale, nc)).reshape(3, nc).T).ravel()
        ax = s.spmv(x3)
        assert np.linalg.norm(ax - b3) <= 1.5e-2 * np.linalg.norm(b3)              # linear_solver_reduction = 1e-2
        if not single:
            # linearity of the SpMV at full size and determinism of the whole solve (bitwise repeatable)
            rng = np.random.default_rng(0)
            z3 = rng.standard_normal(3 * nc)
            assert rel_err(s.spmv(2.0 * x3 - 0.5 * z3), 2.0 * ax - 0.5 * s.spmv(z3)) < 1e-12
            dx2 = m.solveJacobianSystem(want_dx=True, single_precision=False)
            assert np.array_equal(dx, dx2)
        s.close()
    # update with a zero increment is the identity on pressures/saturations; a real update keeps the invariants
    before = m.getState()
    m.updateState(np.zeros(3 * nc))
    same = m.getState()
    assert np.array_equal(before.p, same.p) and np.array_equal(before.sat, same.sat)
    m.updateState(dx)
    after = m.getState()
    assert np.all(after.p > 0) and np.all(after.sat >= 0) and np.all(after.sat <= 1 + 1e-12)
    assert np.abs(after.sat.sum(1) - 1).max() < 1e-12 or np.all(after.sat.sum(1) <= 1 + 1e-9)
    assert np.all(np.abs(after.p - before.p) <= 0.3 * np.abs(before.p) * (1 + 1e-12))
    m.close()


def test_spe9_like_with_well_cliques(gpu_lib, oracle):
    """configs[1]-like: 24x25x15 = 9000 cells, 26 wells (pattern = stencil U per-well cliques), vs the oracle."""
    grid = decks.cartesian_grid(24, 25, 15, dx=91.44, dy=91.44, dz=6.0, tops=2743.0, lognormal_sigma=1.0, seed=9)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, p_ref=248.0 * decks.BAR, z_ref=2743.0, perturb=0.005, seed=9)
    rng = np.random.default_rng(9)
    cols = rng.choice(24 * 25, 26, replace=False)
    connpos, cells = [0], []
    for w, c in enumerate(cols):
        layers = range(10, 15) if w == 0 else range(1, 4)            # injector completed low, producers in layers 2-4
        cells += [int(c + 600 * k) for k in layers]
        connpos.append(len(cells))
    wells = (np.asarray(connpos, np.int32), np.asarray(cells, np.int32))
    prm = capi.default_params(linear_solver_reduction=1e-10, linear_solver_maxiter=500)
    m = GpuBlackoilModel(grid, tab, prm, wells=wells)
    dt = 10 * decks.DAY
    m.prepareStep(dt, st)
    m.assemble(True)
    rowptr, col = oracle.pattern(grid, *wells)
    scale = tuple(prm.matbalscale)
    r0, v0, _, _ = oracle.assemble(grid, tab, dt, st, rowptr, col, scale=scale)
    gr, gc, gv = m.jacobian()
    assert np.array_equal(gr, rowptr) and np.array_equal(gc, col)
    assert col.size > 9000 + 2 * grid.nconn                         # clique fill is part of the pattern
    assert rel_err(gv, v0) < 1e-11 and rel_err(m.residual(), r0) < 1e-11
    # a synthetic Schur complement on the cliques, then the solve against the oracle on the modified system
    rc, blocks = [], []
    for w in range(26):
        cw = cells[connpos[w]:connpos[w + 1]]
        for a in cw:
            for b in cw:
                rc.append((a, b)); blocks.append(np.eye(3).ravel() * (1e-9 if a == b else -1e-10) * np.array([1, 1e5, 1e5, 1, 1e5, 1e5, 1, 1e5, 1e5]))
    rc, blocks = np.asarray(rc, np.int32), np.asarray(blocks)
    delta = rng.standard_normal((len(cells), 3)) * 1e-6
    m.addWellTerms(delta, rc, blocks)
    nc = grid.nc
    r1 = r0.copy()
    for i, c in enumerate(cells):
        for a in range(3):
            r1[a * nc + c] += delta[i, a]
    v1 = v0.copy()
    sc = np.repeat(np.asarray(scale), 3)
    for k, (a, b) in enumerate(rc):
        s = rowptr[a] + np.searchsorted(col[rowptr[a]:rowptr[a + 1]], b)
        v1[s] += blocks[k] * sc
    assert rel_err(m.jacobian()[2], v1) < 1e-11 and rel_err(m.residual(), r1) < 1e-11
    dx = m.solveJacobianSystem(want_dx=True, single_precision=False)
    pos = m.ordering()[0]
    b3 = np.ascontiguousarray((r1 * np.repeat(np.asarray(scale), nc)).reshape(3, nc).T).ravel()
    sto, x, ito, redo, _ = oracle.bicgstab(rowptr, col, v1, b3, prm, position=pos)
    assert sto == 0 and abs(ito - m.linear_iterations) <= 2
    dxo = np.ascontiguousarray(x.reshape(nc, 3).T).ravel()
    for a in range(3):
        blk = slice(a * nc, (a + 1) * nc)
        assert np.abs(dx[blk] - dxo[blk]).max() <= 1e-6 * np.abs(dxo[blk]).max()
    m.close()


def _newton_parity(gpu_lib, oracle, grid, tab, st, dt, prm, wells=None, niter=3):
    """free-running Newton iterations GPU vs oracle (f64 solve, tight linear tolerance)"""
    from util import bsr_to_scipy  # noqa: F401
    m = GpuBlackoilModel(grid, tab, prm)
    m.prepareStep(dt, st)
    rowptr, col = oracle.pattern(grid)
    scale = np.asarray(prm.matbalscale[:])
    nc = grid.nc
    so, acc0, pos = st.copy(), None, None
    for it in range(niter):
        m.assemble(it == 0)
        m.getConvergence()
        m.solveJacobianSystem(single_precision=False)
        m.updateState()
        if pos is None:
            pos = m.ordering()[0]
        r, val, acc0, binv = oracle.assemble(grid, tab, dt, so, rowptr, col, scale=tuple(scale), accum0=acc0)
        b = np.ascontiguousarray((r * np.repeat(scale, nc)).reshape(3, nc).T).ravel()
        sto, x, _, _, _ = oracle.bicgstab(rowptr, col, val, b, prm, position=pos, single=False)
        assert sto == 0
        so = oracle.update_state(grid, tab, prm, np.ascontiguousarray(x.reshape(nc, 3).T).ravel(), so)
        g = m.getState()
        assert np.array_equal(g.hc, so.hc), it
        assert np.abs(g.p - so.p).max() / np.abs(so.p).max() < 1e-6, it
        assert np.abs(g.sat - so.sat).max() < 1e-6, it
    m.close()


def test_spe10_like_heterogeneity(gpu_lib, oracle):
    """configs[3]-like at an oracle-checkable size: channelised lognormal permeability with sigma_lnK = 2.5 (four orders of
    magnitude of transmissibility contrast), thin cells, CPR and ILU0."""
    grid = decks.cartesian_grid(12, 22, 17, dx=6.096, dy=3.048, dz=0.6096, tops=3657.6, lognormal_sigma=2.5, seed=10)
    assert np.log10(grid.trans.max() / grid.trans.min()) > 4
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, p_ref=413.0 * decks.BAR, z_ref=3657.6, perturb=0.005, seed=10)
    for cpr in (0, 1):
        # four orders of magnitude of contrast: the error of a solve is cond(A) x its residual reduction, and which side of the 1e-6
        # state tolerance a 1e-11 reduction lands on depends on the Krylov path (measured 0.9e-6 .. 1.1e-6) -- hence 1e-12
        prm = capi.default_params(linear_solver_reduction=1e-12, linear_solver_maxiter=1500, cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=cpr)
        _newton_parity(gpu_lib, oracle, grid, tab, st, 2 * decks.DAY, prm)


def test_norne_like_unstructured(gpu_lib, oracle):
    """configs[4]-like: 60 % of the cells inactive, fault-style non-neighbour connections (5 % extra), threshold pressures:
    arbitrary BSR rows (1-25 blocks), several ILU colours, isolated cells."""
    rng = np.random.default_rng(44)
    act = rng.random(23 * 28 * 11) > 0.6
    grid = decks.cartesian_grid(23, 28, 11, actnum=act, nnc_fraction=0.05, lognormal_sigma=1.0, thpres=0.02 * decks.BAR, seed=44)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, perturb=0.01, seed=44)
    rowptr, _ = oracle.pattern(grid)
    assert np.diff(rowptr).min() >= 1 and np.diff(rowptr).max() >= 8
    for cpr in (0, 1):
        prm = capi.default_params(linear_solver_reduction=1e-11, linear_solver_maxiter=1500, cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=cpr)
        _newton_parity(gpu_lib, oracle, grid, tab, st, 3 * decks.DAY, prm)


# ------------------------------------------------------------------------------------------------------------------------
# full-size oracle parity
# ------------------------------------------------------------------------------------------------------------------------
def _perforated_diag_mask(rowptr, col, cells):
    """blocks (c, c) of the perforated cells: the only Jacobian entries the device well model changes in the matrix itself"""
    nb = rowptr.size - 1
    rows = np.repeat(np.arange(nb, dtype=np.int64), np.diff(rowptr))
    perf = np.zeros(nb, bool); perf[np.asarray(cells, int)] = True
    return (rows == col) & perf[rows]


def _lockstep_parity(gpu_lib, oracle, grid, tab, st, dt, wl, niter=2, cpr=1, reduction=1e-10, maxiter=2000, tol_p=1e-6, tol_s=1e-6,
                     single=False, gmres=0, verify=0, oracle_reduction=None, tol_jac=1e-11, tol_op=1e-9, stage2_relax=1.0):
    """Newton iterations 0..niter-1 of one time step, GPU (device wells, CPR or ILU0) and oracle (+ host well model with
    the explicit Schur complement) side by side.  Every assembly is compared at rounding level; after every update the two states are
    compared at the linear tolerance and the oracle then CONTINUES FROM THE GPU's state, so the next assembly is again a rounding-level
    comparison (a free-running comparison is test_*_newton_count below).

    single / gmres / verify select the configurations bench.py TIMES: restarted GMRES(40) (newton_use_gmres) under CPR in double -- the
    headline --, and the float variant (the Jacobian written as float, the float CPR solve, GMRES with the true-residual check).  The
    oracle side stays the f64 reference solve (ILU0 + BiCGStab) at `oracle_reduction`, the residual (always f64) stays a rounding-level
    comparison, a float Jacobian / operator are compared at float rounding level (tol_jac / tol_op)."""
    from opmgpu import wells as W
    from util import OracleBackend, rel_err
    oracle.set_threads(16)
    prm_g = capi.default_params(linear_solver_reduction=reduction, linear_solver_maxiter=maxiter, cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=cpr, newton_use_gmres=gmres, gmres_verify_residual=verify,
                                cpr_stage2_relax=stage2_relax)
    prm_o = capi.default_params(linear_solver_reduction=oracle_reduction or reduction, linear_solver_maxiter=4 * maxiter)
    nc = grid.nc
    gm = GpuBlackoilModel(grid, tab, prm_g)
    rowptr0, col0 = oracle.pattern(grid)
    scale = np.asarray(prm_g.matbalscale[:])
    if wl is not None:
        md = W.DeviceWellModel(gm, wl, W.WellState(wl, st.p))
        ob = OracleBackend(oracle, grid, tab, prm_o, wells=wl.arrays())
        mo = W.WellCoupledModel(ob, W.StandardWellsHost(wl, grid.z, tab.surface_density[0]), W.WellState(wl, st.p))
        diag_perf = _perforated_diag_mask(rowptr0, col0, wl.cells)
    else:
        md, ob = gm, OracleBackend(oracle, grid, tab, prm_o)
        mo = ob
    md.prepareStep(dt, st); mo.prepareStep(dt, st)
    rng = np.random.default_rng(5)
    for it in range(niter):
        # ---- assembly ----
        gm.setSolvePrecision(single)
        gm.assemble(it == 0)
        mo.assemble(it == 0)            # with wells: control switching, reservoir, [connection pressures + well pre-solve], well terms
        val_res = None
        if wl is not None:
            # reservoir-only oracle Jacobian on the stencil pattern: everything but the perforated cells' diagonal blocks must match it
            _, val_res, _, _ = oracle.assemble(grid, tab, dt, ob.st, rowptr0, col0, scale=tuple(scale), accum0=ob.acc0)
            if it == 0:
                md.pull_well_state()
                assert md.presolve_converged and md.presolve_iterations == mo.wh.well_iterations
        gr, gc, gv = gm.jacobian()
        assert np.array_equal(gr, rowptr0) and np.array_equal(gc, col0)
        assert rel_err(gm.residual(), ob.r) < 1e-11, (it, rel_err(gm.residual(), ob.r))
        if wl is None:
            assert rel_err(gv, ob.val) < tol_jac, it
        else:
            keep = ~diag_perf
            assert rel_err(gv[keep], val_res[keep]) < tol_jac, (it, rel_err(gv[keep], val_res[keep]))
            # the coupled operator (matrix + factored rank-7 Schur complement per well) against the oracle's explicit clique matrix
            for _ in range(2):
                x3 = rng.standard_normal(3 * nc) * np.tile([1e5, 1e-2, 1e-2], nc)
                yo = oracle.spmv(ob.rowptr, ob.col, ob.val, x3)
                yg = gm.spmv(x3)
                assert rel_err(yg, yo) < tol_op, (it, rel_err(yg, yo))
        del gv, val_res
        # ---- convergence scalars ----
        cg = gm.getConvergence(); co = ob.getConvergence()
        assert np.allclose(gm.CNV, ob.CNV, rtol=1e-9) and np.allclose(gm.MB, ob.MB, rtol=1e-7, atol=1e-18) and np.allclose(gm.B_avg, ob.B_avg, rtol=1e-12)
        if wl is not None:
            cg = md.wellConvergence() and cg; co = mo.wh.converged(ob.B_avg) and co
            assert np.allclose(md.well_flux_residual, mo.wh.well_flux_residual, rtol=1e-7, atol=1e-14)
            assert md.well_ctrl_residual == pytest.approx(mo.wh.well_ctrl_residual, rel=1e-7, abs=1e-14)
        assert cg == co
        # ---- solve + update ----
        gm.solveJacobianSystem(single_precision=single)
        assert gm.linear_reduction <= reduction, (it, gm.linear_reduction)        # with gmres_verify_residual: the TRUE residual's reduction
        gm.updateState()
        ob.solveJacobianSystem(single_precision=False)
        if wl is not None:
            mo.wh.recover_and_update(ob.perfDx(wl.nperf), mo.ws)
        ob.updateState()
        a, b = gm.getState(), ob.getState()
        assert np.array_equal(a.hc, b.hc), it
        assert np.abs(a.p - b.p).max() <= tol_p * np.abs(b.p).max(), (it, np.abs(a.p - b.p).max() / np.abs(b.p).max())
        assert np.abs(a.sat - b.sat).max() <= tol_s, (it, np.abs(a.sat - b.sat).max())
        # rs / rv at a fixed 1e-5 (float: rs / rv of the undersaturated cells are unknowns of the float solve themselves).  Round 3 had widened
        # this to 10 * tol_p after the SPE10-like leg measured 0.9e-5 .. 1.9e-5 from run to run: the checker's 16-thread dot products were not
        # bit-reproducible (an OpenMP reduction clause).  They are summed in fixed blocks now (oracle.cpp dot_t), so the old gate is back.
        tol_r = 1e-5 if not single else 2.5 * tol_s
        assert np.abs(a.rs - b.rs).max() <= tol_r * max(np.abs(b.rs).max(), 1.0) and np.abs(a.rv - b.rv).max() <= tol_r * max(np.abs(b.rv).max(), 1e-3)
        ob.st = a.copy()                                 # lockstep: the oracle continues from the device state
        if wl is not None:
            ws = md.pull_well_state()
            assert np.allclose(ws.bhp, mo.ws.bhp, rtol=max(1e-6, tol_p)), (it, ws.bhp, mo.ws.bhp)
            assert np.allclose(ws.qs, mo.ws.qs, rtol=max(1e-5, 10 * tol_p), atol=max(1e-8, tol_p) * np.abs(mo.ws.qs).max()), it
            mo.ws.assign(ws)
    gm.close()


def _run_time_step(model, dt, st, single, ns, max_iter=15):
    """NonlinearSolver::step (NonlinearSolver_impl.hpp:119-174) with the reference's default update stabilisation: returns the number
    of nonlinear iterations (max_iter + 1 = not converged)"""
    model.prepareStep(dt, st)
    it = 0
    while True:
        conv, _ = model.nonlinearIteration(it, single_precision=single, nonlinear_solver=ns)
        it += 1
        if (conv and it > 1) or it > max_iter:
            return it


class _OracleModel:
    """BlackoilModelBase::nonlinearIteration on the bare oracle backend (decks without wells)"""

    def __init__(self, ob):
        self.m = ob

    def prepareStep(self, dt, st):
        self.m.prepareStep(dt, st)

    def nonlinearIteration(self, it, single_precision=False, nonlinear_solver=None):
        ob, ns = self.m, nonlinear_solver
        if it == 0:
            self.hist, self.relax = [], 1.0
        ob.assemble(it == 0)
        conv = ob.getConvergence()
        self.hist.append(list(ob.linf))
        if not conv or it < 1:
            ob.solveJacobianSystem(single_precision=single_precision)
            if ns is not None:
                if ns.detectOscillations(self.hist, it)[0]:
                    self.relax = max(self.relax - ns.relax_increment, ns.relax_max)
                ob.stabilizeUpdate(ns.relax_type, self.relax)
            ob.updateState()
        return conv, ob.linear_iterations


def _spin_up(grid, tab, st, wl, dt, nsteps=2):
    """The synthetic initial states are far from capillary-gravity equilibrium and the wells start at full rate (first-step CNV of
    several hundred): like a simulator run, let `nsteps` time steps pass (on the device, default tolerances) and compare Newton
    behaviour on the step after them.  Returns the state and well state at the end."""
    from opmgpu import wells as W
    from opmgpu.model import NonlinearSolver
    gm = GpuBlackoilModel(grid, tab, capi.default_params(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1))
    md = gm if wl is None else W.DeviceWellModel(gm, wl, W.WellState(wl, st.p))
    ns = NonlinearSolver()
    cur = st
    for k in range(nsteps):
        n = _run_time_step(md, dt, cur if k == 0 else None, dt < 20 * decks.DAY, ns, max_iter=25)
        assert n <= 25, "spin-up step %d did not converge" % k
    out = gm.getState()
    ws = None if wl is None else md.pull_well_state().copy()
    gm.close()
    return out, ws


def _check_newton_count(gpu_lib, oracle, grid, tab, st, dt, wl, solvers=(0, 1), reduction=1e-6, spin_up=2, oracle_gmres=True, gmres_reduction_factor=1e-1, gmres_tol=1e-4, stage2_relax=1.0):
    """One whole time step with the reference's NonlinearSolver (update stabilisation on) and the reference's Newton tolerances
    (MB 1e-5, CNV 1e-2, wells 1e-4 / 1e-7), free-running on both sides from the same spun-up state.  The linear solves are double
    precision to a 1e-6 reduction on BOTH sides: at the default 1e-2 an inexact-Newton path depends on the preconditioner (measured:
    8-16 iterations for one deck across ILU0 orderings / CPR), at equal tight tolerance the paths coincide and the device must need
    exactly the oracle's number of Newton iterations."""
    from opmgpu import wells as W
    from opmgpu.model import NonlinearSolver
    from util import OracleBackend
    oracle.set_threads(16)
    st1, ws1 = _spin_up(grid, tab, st, wl, dt, spin_up) if spin_up else (st, None)
    def lin_of(gmres):
        # GMRES against GMRES: both sides stop on their PRECONDITIONED residual (dune's rule), the device's behind CPR and the oracle's behind
        # ILU0 -- two different norms; at 1e-6 the converged states sat 2e-4 apart on the Norne-like deck for that reason alone.  Both sides
        # therefore solve to 1e-7 there, which puts the linear error below the 1e-4 state tolerance whatever the norm (restarted GMRES(40) does
        # not reach 1e-8 on the Norne-like grid with its isolated cells).
        return dict(linear_solver_reduction=reduction * (gmres_reduction_factor if (gmres and oracle_gmres) else 1.0), linear_solver_maxiter=3000)

    def oracle_step(gmres):
        """the oracle's time step with ITS restatement of the same Krylov method (BiCGStab, or Dune::RestartedGMResSolver: oracle.cpp gmres_t)"""
        ob = OracleBackend(oracle, grid, tab, capi.default_params(newton_use_gmres=gmres, **lin_of(gmres)), wells=None if wl is None else wl.arrays())
        if wl is None:
            mo = _OracleModel(ob)
        else:
            wso = W.WellState(wl, st1.p)
            if ws1 is not None:
                wso.assign(ws1)
            mo = W.WellCoupledModel(ob, W.StandardWellsHost(wl, grid.z, tab.surface_density[0]), wso)
        n = _run_time_step(mo, dt, st1, False, NonlinearSolver())
        assert n <= 15, n
        return n, ob.getState()

    oracle_runs = {}
    for code in solvers:          # bit 0: CPR instead of ILU0; bit 1: restarted GMRES instead of BiCGStab (newton_use_gmres)
        cpr = code & 1
        gmres = code >> 1
        okey = gmres if oracle_gmres else 0
        if okey not in oracle_runs:
            oracle_runs[okey] = oracle_step(okey)
        n_oracle, b = oracle_runs[okey]
        # device GMRES against the oracle's GMRES: both stop on their preconditioned residual like dune's.  Where the oracle's
        # ILU0-preconditioned GMRES(40) is not affordable (1 M cells), the device checks the TRUE residual (gmres_verify_residual),
        # which is the statement the oracle's BiCGStab makes -- the same 1e-4 state tolerance in both cases.
        gm = GpuBlackoilModel(grid, tab, capi.default_params(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=cpr, newton_use_gmres=gmres, gmres_verify_residual=int(gmres == 1 and not oracle_gmres),
                                                             cpr_stage2_relax=stage2_relax, **lin_of(gmres)))
        if wl is None:
            md = gm
        else:
            wsd = W.WellState(wl, st1.p)
            if ws1 is not None:
                wsd.assign(ws1)
            md = W.DeviceWellModel(gm, wl, wsd)
        n_gpu = _run_time_step(md, dt, st1, False, NonlinearSolver())
        a = gm.getState()
        gm.close()
        assert n_gpu == n_oracle, (code, n_gpu, n_oracle)
        assert np.array_equal(a.hc, b.hc)
        tol = gmres_tol if (gmres and oracle_gmres) else 1e-4
        assert np.abs(a.p - b.p).max() <= tol * np.abs(b.p).max() and np.abs(a.sat - b.sat).max() <= tol, (code, np.abs(a.p - b.p).max() / np.abs(b.p).max(), np.abs(a.sat - b.sat).max())


from opmgpu import baseline_decks as _bd          # the deck recipes live with the package: bench.py times the same ones

_cart100, _spe10_like, _norne_like, _spe9_like, _cart60 = _bd.cart100, _bd.spe10_like, _bd.norne_like, _bd.spe9_like, _bd.cart60


DECKS = {"cart100": (_cart100, 5.0), "spe10like": (_spe10_like, 2.0), "nornelike": (_norne_like, 3.0), "spe9like": (_spe9_like, 3.0)}
COUNT_DECKS = dict(DECKS, cart60=(_cart60, 5.0))
# (the two 1 M-cell decks walk their SECOND iteration in the timed configuration below -- same assembly and well comparisons, GMRES instead of
# BiCGStab --, so the BiCGStab leg stops after the first there: the whole GPU suite has to stay well inside the box's time limit)
# cpr_stage2_relax = 0.9 (library extension, include/opmgpu.h; what rounds 1-3 ran under the name ilu_relaxation): the Norne-like deck with its
# isolated cells does not reach the tight reductions of these tests with the undamped second stage of the reference's cpr_relax = 1 (measured:
# Convergence failure at 1e-10 / 1e-6), and the float solve of cart100_f32 then leaves 1.5e-2 in the saturations instead of 1.4e-3
LOCKSTEP_KW = {"cart100": dict(niter=1), "nornelike": dict(stage2_relax=0.9),
               "spe10like": dict(niter=1, reduction=1e-8, tol_p=1e-5, tol_s=1e-5)}        # sigma_lnK = 2.5: a 1e-10 reduction is below what BiCGStab attains in f64
# the configurations bench.py times, device wells everywhere:
#   *_f64: CPR in double + GMRES(40) with dune's stopping rule -- the headline (the reference's CPR plug-in is double-only) -- at the f64 tolerances
#          of the BiCGStab legs above;
#   cart100_f32: the float variant exactly as bench.py runs it (float Jacobian, float CPR solve, GMRES with dune's rule), asked for 1e-5.  What a
#          float solve attains on this system: with the true-residual check, 1e-5 on the true residual was reached by an early form of the check
#          (pressures then agreed to 2e-5 relative, the first iteration's saturations -- |ds| up to the 0.2 chop -- to 8e-5) and is NOT reached
#          within 1000 iterations by the present one; dune's rule stops on the preconditioned residual, a few times further from the true one.
#          The UPDATED states are therefore compared at 1e-3 / 5e-3 here (measured: 1.4e-3 in the first iteration's saturations, whose
#          increments reach the 0.2 chop, 4e-4 in the second iteration's pressures) -- the linear solve's accuracy times cond(A), not a kernel error: the
#          float Jacobian (5e-7), the float operator (2e-5), the residual (1e-11), the convergence scalars and the well residuals of the same
#          assemblies are compared at rounding level above it
TIMED_KW = {"cart100_f64": ("cart100", dict(gmres=1, reduction=1e-10, maxiter=400)),
            # (sigma_lnK = 2.5: dune's rule stops on the PRECONDITIONED residual, which at 1e-8 left 7e-4 in the saturations here: 1e-11)
            "spe10like_f64": ("spe10like", dict(gmres=1, reduction=1e-11, oracle_reduction=1e-8, maxiter=800, tol_p=1e-5, tol_s=1e-5)),
            "cart100_f32": ("cart100", dict(single=True, gmres=1, reduction=1e-5, oracle_reduction=1e-10, maxiter=300, tol_p=1e-3, tol_s=5e-3, tol_jac=5e-7, tol_op=2e-5, stage2_relax=0.9))}
# solvers: bit 0 = CPR, bit 1 = GMRES.  Multicolour ILU0 alone needs ~1000 iterations for 1e-6 at 1 M cells.  GMRES legs run against the
# oracle's own GMRES restatement, both at 1e-7 -- except at 1 M cells (cart100: too slow on the host, see _cart60; the device verifies the true
# residual there) and on the Norne-like grid: with its isolated cells restarted GMRES(40) stalls near 1e-7 on either side (and does not reach a
# verified 1e-6 either), so both sides stop at 1e-6 on their OWN preconditioned residuals -- CPR's and ILU0's, two different norms -- which
# leaves the converged states 2e-4 apart (measured); the Newton counts must still be equal
COUNT_KW = {"cart100": dict(solvers=(1, 3), oracle_gmres=False), "cart60": dict(solvers=(1, 3)), "spe10like": dict(solvers=(1,)),
            "nornelike": dict(spin_up=0, solvers=(0, 1, 3), gmres_reduction_factor=1.0, gmres_tol=3e-4, stage2_relax=0.9), "spe9like": dict(spin_up=0, solvers=(0, 1, 2, 3))}


@pytest.mark.parametrize("name", list(DECKS))
def test_fullsize_lockstep_parity(gpu_lib, oracle, name):
    make, dt_days = DECKS[name]
    grid, tab, st, wl = make()
    _lockstep_parity(gpu_lib, oracle, grid, tab, st, dt_days * decks.DAY, wl, **LOCKSTEP_KW.get(name, {}))


@pytest.mark.parametrize("name", list(TIMED_KW))
def test_fullsize_lockstep_parity_of_the_timed_configuration(gpu_lib, oracle, name):
    """CPR + GMRES(40) + device wells, in double (bench.py's headline) and in float (its cpr_f32 variant), in lockstep with the oracle"""
    deck, kw = TIMED_KW[name]
    make, dt_days = DECKS[deck]
    grid, tab, st, wl = make()
    _lockstep_parity(gpu_lib, oracle, grid, tab, st, dt_days * decks.DAY, wl, **kw)


@pytest.mark.parametrize("name", list(COUNT_DECKS))
def test_fullsize_newton_count(gpu_lib, oracle, name):
    make, dt_days = COUNT_DECKS[name]
    grid, tab, st, wl = make()
    _check_newton_count(gpu_lib, oracle, grid, tab, st, dt_days * decks.DAY, wl, **COUNT_KW.get(name, {}))
