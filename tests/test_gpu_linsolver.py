"""GPU parity: SELL-64 SpMV, level-scheduled block-ILU0 and BiCGStab vs the CPU oracle, through the C ABI."""
import os

import numpy as np
import pytest
import scipy.sparse.linalg as spla

from opmgpu import capi, decks
from opmgpu.model import GpuNewtonIteration, ISTLError, LinearSolverProblem
from util import bsr_to_scipy, random_block_matrix, rel_err

pytestmark = pytest.mark.gpu

WELLS = (np.array([0, 5, 9], np.int32), np.array([3, 40, 77, 110, 200, 5, 6, 90, 301], np.int32))


def _patterns(oracle):
    g1 = decks.cartesian_grid(9, 7, 6)                                    # 378 rows: not a multiple of 64
    g2 = decks.cartesian_grid(8, 8, 5, nnc_fraction=0.05)                 # NNC: unstructured rows
    act = np.random.default_rng(3).random(10 * 9 * 6) > 0.4
    g3 = decks.cartesian_grid(10, 9, 6, actnum=act)                       # holes: ragged rows
    return [("cart", oracle.pattern(g1)), ("nnc+wells", oracle.pattern(g2, *WELLS)), ("actnum", oracle.pattern(g3))]


@pytest.mark.parametrize("ordering", [capi.ORDER_NATURAL, capi.ORDER_MULTICOLOR])
@pytest.mark.parametrize("single", [False, True])
def test_spmv_ilu_parity(gpu_lib, oracle, ordering, single):
    for name, (rowptr, col) in _patterns(oracle):
        nb = rowptr.size - 1
        val = random_block_matrix(rowptr, col, seed=nb)
        rng = np.random.default_rng(nb + 1)
        x = rng.uniform(-1, 1, 3 * nb)
        s = GpuNewtonIteration(capi.default_params(ilu_ordering=ordering))
        s.load(rowptr, col, val, single)
        # SpMV
        y, yo = s.spmv(x), oracle.spmv(rowptr, col, val, x, single)
        assert rel_err(y, yo) < (2e-6 if single else 1e-14), name
        # ordering reported by the library is a valid elimination order; feed it to the oracle
        pos, lev, nl = s.ordering()
        assert sorted(pos.tolist()) == list(range(nb))
        # ILU0 factors
        s.ilu0_factor()
        lu = s.ilu0_get(col.size)
        st, luo = oracle.ilu0(rowptr, col, val, position=pos, single=single)
        assert st == 0
        assert rel_err(lu, luo) < (5e-6 if single else 1e-13), name
        # ILU0 apply (relaxation 0.9 folded into the forward sweep)
        v, vo = s.ilu0_apply(x), oracle.ilu0_apply(rowptr, col, luo, x, position=pos, relax=0.9, single=single)
        assert rel_err(v, vo) < (5e-6 if single else 1e-13), name
        s.close()


def test_spmv_linearity_and_edge_sizes(gpu_lib, oracle):
    for nb in (1, 2, 63, 64, 65):
        rowptr = np.arange(nb + 1, dtype=np.int32) if nb < 3 else None
        if rowptr is None:
            rows = [[j for j in (i - 1, i, i + 1) if 0 <= j < nb] for i in range(nb)]
            rowptr = np.cumsum([0] + [len(r) for r in rows]).astype(np.int32)
            col = np.concatenate(rows).astype(np.int32)
        else:
            col = np.arange(nb, dtype=np.int32)
        val = random_block_matrix(rowptr, col, seed=nb)
        s = GpuNewtonIteration()
        s.load(rowptr, col, val)
        rng = np.random.default_rng(nb)
        x, z = rng.standard_normal(3 * nb), rng.standard_normal(3 * nb)
        assert rel_err(s.spmv(2.0 * x - 3.0 * z), 2.0 * s.spmv(x) - 3.0 * s.spmv(z)) < 1e-13
        assert rel_err(s.spmv(x), bsr_to_scipy(rowptr, col, val) @ x) < 1e-14
        s.close()


@pytest.mark.parametrize("ordering", [capi.ORDER_NATURAL, capi.ORDER_MULTICOLOR])
@pytest.mark.parametrize("single", [False, True])
def test_bicgstab_parity(gpu_lib, oracle, ordering, single):
    """computeNewtonIncrement on an assembled black-oil Jacobian: same stopping rule as the oracle,
    solution within the solver tolerance of the exact solve, iteration count within +-2."""
    grid = decks.cartesian_grid(12, 10, 8, lognormal_sigma=1.0)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, perturb=0.01)
    prm = capi.default_params(ilu_ordering=ordering)
    rowptr, col = oracle.pattern(grid)
    r, val, _, _ = oracle.assemble(grid, tab, 10 * decks.DAY, st, rowptr, col, scale=tuple(prm.matbalscale))
    nc = grid.nc
    b = np.ascontiguousarray((r * np.repeat(np.asarray(prm.matbalscale[:]), nc)).reshape(3, nc).T).ravel()
    A = bsr_to_scipy(rowptr, col, val)
    xe = spla.spsolve(A.tocsc(), b)
    s = GpuNewtonIteration(prm)
    x = s.computeNewtonIncrement(rowptr, col, val, b, single)
    pos, _, _ = s.ordering()
    sto, xo, ito, redo, _ = oracle.bicgstab(rowptr, col, val, b, prm, position=pos, single=single)
    assert sto == 0
    assert abs(s.iterations() - ito) <= 2, (s.iterations(), ito)
    res = np.linalg.norm(A @ x - b) / np.linalg.norm(b)
    assert res < 1.5e-2 and s.reduction < 1e-2, (res, s.reduction)          # linear_solver_reduction = 1e-2
    # tight tolerance: both converge to the direct solution
    prm2 = capi.default_params(ilu_ordering=ordering, linear_solver_reduction=1e-5 if single else 1e-10, linear_solver_maxiter=300)
    s2 = GpuNewtonIteration(prm2)
    x2 = s2.computeNewtonIncrement(rowptr, col, val, b, single)
    if single:      # f32: as accurate as the oracle's f32 solve of the same system (cond(A) * eps_f32 limits both)
        _, xo2, _, _, _ = oracle.bicgstab(rowptr, col, val, b, prm2, position=pos, single=True)
        err_o = np.linalg.norm(xo2 - xe) / np.linalg.norm(xe)
        assert np.linalg.norm(x2 - xe) / np.linalg.norm(xe) < 3.0 * err_o + 1e-6
    else:
        assert np.linalg.norm(x2 - xe) / np.linalg.norm(xe) < 1e-7
    s.close(); s2.close()


def test_solver_error_contract(gpu_lib, oracle):
    grid = decks.cartesian_grid(6, 5, 4)
    rowptr, col = oracle.pattern(grid)
    val = random_block_matrix(rowptr, col, seed=2, dominance=0.31)       # weakly dominant: needs iterations
    b = np.random.default_rng(0).standard_normal(3 * grid.nc)
    s = GpuNewtonIteration(capi.default_params(linear_solver_maxiter=1, linear_solver_reduction=1e-12))
    with pytest.raises(LinearSolverProblem):                             # ISTLSolver.hpp:358-368
        s.computeNewtonIncrement(rowptr, col, val, b, False)
    s.close()
    s = GpuNewtonIteration(capi.default_params(linear_solver_maxiter=1, linear_solver_reduction=1e-12, ignore_convergence_failure=1))
    s.computeNewtonIncrement(rowptr, col, val, b, False)                 # ignoreConvergenceFailure_
    s.close()
    val2 = val.copy()
    val2[np.flatnonzero(col == 0)[0]] = 0.0                              # singular pivot block in row 0
    s = GpuNewtonIteration()
    with pytest.raises(ISTLError):
        s.computeNewtonIncrement(rowptr, col, val2, b, False)
    with pytest.raises(ValueError):                                      # pattern without a diagonal
        s.computeNewtonIncrement(np.array([0, 1, 2], np.int32), np.array([1, 0], np.int32), np.ones((2, 9)), np.ones(6), False)
    s.close()


def test_pattern_cache_and_replan(gpu_lib, oracle):
    s = GpuNewtonIteration()
    for dims in ((5, 4, 3), (5, 4, 3), (6, 4, 3)):
        grid = decks.cartesian_grid(*dims)
        rowptr, col = oracle.pattern(grid)
        val = random_block_matrix(rowptr, col, seed=sum(dims))
        b = np.random.default_rng(1).standard_normal(3 * grid.nc)
        x = s.computeNewtonIncrement(rowptr, col, val, b, False)
        A = bsr_to_scipy(rowptr, col, val)
        assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-2
    s.close()


@pytest.mark.parametrize("single", [False, True])
def test_cpr_amg_solve(gpu_lib, oracle, single):
    """solver_approach=cpr: AMG V-cycle on the pressure system + ILU0.  Contract: same stopping rule,
    dx solves the system to the linear tolerance; and it must need far fewer iterations than ILU0 alone on a
    Newton iteration with a hard pressure part."""
    grid = decks.cartesian_grid(20, 18, 16, lognormal_sigma=0.8)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, perturb=0.004)
    prm0 = capi.default_params()
    scale = np.asarray(prm0.matbalscale[:])
    rowptr, col = oracle.pattern(grid)
    nc = grid.nc
    dt = 5 * decks.DAY
    # second Newton iteration of the step: the first one is easy for any preconditioner
    r, val, acc0, _ = oracle.assemble(grid, tab, dt, st, rowptr, col, scale=tuple(scale))
    b = np.ascontiguousarray((r * np.repeat(scale, nc)).reshape(3, nc).T).ravel()
    _, x, _, _, _ = oracle.bicgstab(rowptr, col, val, b, prm0)
    st1 = oracle.update_state(grid, tab, prm0, np.ascontiguousarray(x.reshape(nc, 3).T).ravel(), st)
    r, val, _, _ = oracle.assemble(grid, tab, dt, st1, rowptr, col, scale=tuple(scale), accum0=acc0)
    b = np.ascontiguousarray((r * np.repeat(scale, nc)).reshape(3, nc).T).ravel()
    A = bsr_to_scipy(rowptr, col, val)
    xe = spla.spsolve(A.tocsc(), b)
    its = {}
    for cpr in (0, 1):
        s = GpuNewtonIteration(capi.default_params(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=cpr))
        xs = s.computeNewtonIncrement(rowptr, col, val, b, single)
        its[cpr] = s.iterations()
        assert np.linalg.norm(A @ xs - b) <= 1.5e-2 * np.linalg.norm(b) and s.reduction < 1e-2
        s.close()
    assert its[1] * 3 <= its[0], its                      # CPR: at least 3x fewer iterations here
    red = 1e-5 if single else 1e-10
    s = GpuNewtonIteration(capi.default_params(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1, linear_solver_reduction=red, linear_solver_maxiter=200))
    xs = s.computeNewtonIncrement(rowptr, col, val, b, single)
    if single:
        assert np.linalg.norm(A @ xs - b) <= 30 * red * np.linalg.norm(b)
    else:
        assert np.linalg.norm(xs - xe) <= 1e-6 * np.linalg.norm(xe)
    # the same matrix again (hierarchy reused, numeric Galerkin repeated): bitwise identical result
    xs2 = s.computeNewtonIncrement(rowptr, col, val, b, single)
    assert np.array_equal(xs, xs2)
    s.close()
    # opt-in smoother of level 0: Gauss-Seidel by colour (OPMGPU_AMG_GS=1, read when the hierarchy is built): same contract
    os.environ["OPMGPU_AMG_GS"] = "1"; os.environ["OPMGPU_AMG_NPOST0"] = "1"
    try:
        s = GpuNewtonIteration(capi.default_params(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1, linear_solver_reduction=red, linear_solver_maxiter=200))
        xg = s.computeNewtonIncrement(rowptr, col, val, b, single)
        assert s.iterations() <= its[0]
        if single:
            assert np.linalg.norm(A @ xg - b) <= 30 * red * np.linalg.norm(b)
        else:
            assert np.linalg.norm(xg - xe) <= 1e-6 * np.linalg.norm(xe)
        s.close()
    finally:
        del os.environ["OPMGPU_AMG_GS"], os.environ["OPMGPU_AMG_NPOST0"]


def test_cpr_pressure_equation_weights(gpu_lib, oracle):
    """formEllipticSystem's dominance test (NewtonIterationUtilities.cpp:212-252) restated here in numpy from the reference
    text: per cell and phase, strong <=> |J_ii| / (column sum of |J_ji|, j != i) > 0.01 on the pressure derivative of the
    scaled equations; a weak oil equation with nothing else strong falls back to the oil equation alone."""
    import ctypes as C
    grid = decks.cartesian_grid(9, 8, 7, nnc_fraction=0.03, lognormal_sigma=0.8)
    rowptr, col = oracle.pattern(grid)
    nb = rowptr.size - 1
    val = random_block_matrix(rowptr, col, seed=5).reshape(-1, 3, 3)
    rng = np.random.default_rng(6)
    rows = np.repeat(np.arange(nb), np.diff(rowptr))
    diag = rows == col
    # manufacture weak pressure diagonals: gas equation in a third of the cells, all three in a few, oil only in some
    weak_g, weak_all, weak_o = rng.random(nb) < 0.3, rng.random(nb) < 0.05, rng.random(nb) < 0.1
    for c in range(nb):
        d = np.flatnonzero(diag & (rows == c))[0]
        if weak_g[c] or weak_all[c]:
            val[d, 2, 0] *= 1e-4
        if weak_all[c] or weak_o[c]:
            val[d, 1, 0] *= 1e-4
        if weak_all[c]:
            val[d, 0, 0] *= 1e-4
    val = val.reshape(-1, 9)
    # reference rule in numpy
    expect = np.zeros((3, nb))
    v3 = val.reshape(-1, 3, 3)
    for eq in range(3):
        dj = np.abs(v3[diag, eq, 0])[np.argsort(rows[diag])]
        colsum = np.zeros(nb)
        np.add.at(colsum, col, np.abs(v3[:, eq, 0]))
        with np.errstate(divide="ignore", invalid="ignore"):
            expect[eq] = (dj / (colsum - dj) > 0.01)
    none = (expect.sum(0) == 0)
    expect[1, none] = 1.0
    assert none.any() and (expect[2] == 0).any() and ((expect[1] == 0) & (expect[0] == 1)).any()
    b = rng.standard_normal(3 * nb)
    s = GpuNewtonIteration(capi.default_params(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1, linear_solver_maxiter=400))
    try:
        s.computeNewtonIncrement(rowptr, col, val, b, False)
    except LinearSolverProblem:
        pass                                             # the random matrix is not the point here
    w = np.zeros(3 * nb)
    assert s.lib.opmgpu_get_cpr_weights(s.ctx, capi.dptr(w)) == capi.OK
    assert np.array_equal(w.reshape(3, nb), expect)
    s.close()


@pytest.mark.parametrize("single", [False, True])
def test_gmres_option(gpu_lib, oracle, single):
    """newton_use_gmres (ISTLSolver.hpp:257-264): restarted, left-preconditioned GMRES as Dune::RestartedGMResSolver does it, vs
    the oracle's restatement with the ILU0 in the same elimination order -- same iteration count (+-1: rounding at the
    threshold), same solution; restart shorter than the iteration count; CPR as the preconditioner; failure contract."""
    grid = decks.cartesian_grid(12, 10, 8, lognormal_sigma=0.8)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, perturb=0.01)
    red = 1e-4 if single else 1e-6
    # natural-order ILU0 (the stronger one): GMRES(m) stagnates more easily than BiCGStab behind the 2-colour ILU0
    prm = capi.default_params(newton_use_gmres=1, linear_solver_reduction=red, linear_solver_maxiter=400, ilu_ordering=capi.ORDER_NATURAL)
    scale = np.asarray(prm.matbalscale[:])
    rowptr, col = oracle.pattern(grid)
    nc = grid.nc
    r, val, _, _ = oracle.assemble(grid, tab, 5 * decks.DAY, st, rowptr, col, scale=tuple(scale))
    b = np.ascontiguousarray((r * np.repeat(scale, nc)).reshape(3, nc).T).ravel()
    A = bsr_to_scipy(rowptr, col, val)
    xe = spla.spsolve(A.tocsc(), b)
    for restart in (40, 12):
        prm.linear_solver_restart = restart
        sto, xo, ito, redo, _ = oracle.bicgstab(rowptr, col, val, b, prm, position=None, single=single)
        assert sto == 0, (restart, ito, redo)
        s = GpuNewtonIteration(prm)
        xg = s.computeNewtonIncrement(rowptr, col, val, b, single)
        assert abs(s.iterations() - ito) <= (3 if single else 1), (restart, s.iterations(), ito)
        assert s.reduction < red
        # left-preconditioned: the PRECONDITIONED residual is controlled; the two implementations must agree with each other
        assert np.linalg.norm(xg - xo) <= (50 * red if single else 1e-6) * np.linalg.norm(xo), restart
        if restart == 12:
            assert s.iterations() > 12                        # the restart path ran
        s.close()
    # CPR as the preconditioner of GMRES (NewtonIterationBlackoilCPR.hpp:116-119)
    red_c = 1e-5 if single else 1e-10
    prm_c = capi.default_params(newton_use_gmres=1, cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1, linear_solver_reduction=red_c, linear_solver_maxiter=200)
    s = GpuNewtonIteration(prm_c)
    xg = s.computeNewtonIncrement(rowptr, col, val, b, single)
    assert np.linalg.norm(xg - xe) <= (3e-3 if single else 1e-6) * np.linalg.norm(xe) and s.iterations() < 40
    s.close()
    # iteration limit: LinearSolverProblem like ISTLSolver.hpp:358-368
    s = GpuNewtonIteration(capi.default_params(newton_use_gmres=1, linear_solver_reduction=1e-12, linear_solver_maxiter=3))
    with pytest.raises(LinearSolverProblem):
        s.computeNewtonIncrement(rowptr, col, val, b, single)
    s.close()


@pytest.mark.parametrize("ordering", [capi.ORDER_NATURAL, capi.ORDER_MULTICOLOR])
def test_reference_point_ilu0_second_stage(gpu_lib, ordering):
    """cpr_reference_transform = 2: the reference's own second stage under CPR -- a POINT ILU0 of the row-transformed system
    taken as a scalar equation-major matrix (NewtonIterationBlackoilCPR.cpp:129-133) -- instead of the 3x3-block ILU0.  Preconditioner-only:
    the solution is the untransformed system's under BiCGStab and GMRES; and the point ILU0 ITSELF is checked against a numpy ILU0
    (IKJ, dune's bilu0 for 1x1 blocks) of the transformed matrix read back from the device, in the natural (dune's) elimination order."""
    import ctypes as C
    from opmgpu.model import GpuBlackoilModel
    grid = decks.cartesian_grid(6, 5, 4, lognormal_sigma=0.8, seed=2)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, perturb=0.01, seed=2)
    nc = grid.nc
    out = {}
    for tr in (0, 1, 2):
        for gmres in (0, 1):
            m = GpuBlackoilModel(grid, tab, capi.default_params(cpr_reference_transform=tr, newton_use_gmres=gmres, ilu_ordering=ordering,
                                                               linear_solver_reduction=1e-10, linear_solver_maxiter=400, **capi.CPR_AMG_VCYCLE))
            m.prepareStep(5 * decks.DAY, st)
            m.assemble(True); m.getConvergence()
            dx = m.solveJacobianSystem(want_dx=True, single_precision=False)
            out[(tr, gmres)] = (dx, m.linear_iterations)
            if tr == 2 and gmres == 0 and ordering == capi.ORDER_NATURAL:
                rowptr, col, val = m.jacobian()                    # L A: the transform works in place
                n = 3 * nc
                A = np.zeros((n, n)); P = np.zeros((n, n), bool)
                for i in range(nc):
                    for s_ in range(rowptr[i], rowptr[i + 1]):
                        j = col[s_]
                        for e in range(3):
                            for v in range(3):
                                A[e * nc + i, v * nc + j] = val[s_][3 * e + v]; P[e * nc + i, v * nc + j] = True
                LU = A.copy()
                idx = np.arange(n)
                for i in range(n):
                    for k in np.where(P[i, :i])[0]:
                        LU[i, k] /= LU[k, k]
                        cols = np.where(P[i] & P[k] & (idx > k))[0]
                        LU[i, cols] -= LU[i, k] * LU[k, cols]
                rng = np.random.default_rng(0)
                d = rng.standard_normal(n)
                y = d.copy()
                for i in range(n):
                    ks = np.where(P[i, :i])[0]
                    y[i] -= LU[i, ks] @ y[ks]
                x = y.copy()
                for i in range(n - 1, -1, -1):
                    ks = np.where(P[i, i + 1:])[0] + i + 1
                    x[i] = (x[i] - LU[i, ks] @ x[ks]) / LU[i, i]
                d3 = np.ascontiguousarray(d.reshape(3, nc).T).ravel()
                v3 = np.zeros(n)
                m._chk(m.lib.opmgpu_point_ilu_apply(m.ctx, capi.dptr(d3), capi.dptr(v3), C.c_double(1.0)))
                vg = np.ascontiguousarray(v3.reshape(nc, 3).T).ravel()
                assert np.abs(vg - x).max() <= 1e-10 * np.abs(x).max(), np.abs(vg - x).max() / np.abs(x).max()
            m.close()
    for gmres in (0, 1):
        ref = out[(0, gmres)][0]
        for tr in (1, 2):
            dx = out[(tr, gmres)][0]
            for a in range(3):
                blk = slice(a * nc, (a + 1) * nc)
                assert np.abs(dx[blk] - ref[blk]).max() <= 2e-6 * np.abs(ref[blk]).max(), (tr, gmres, a)
        # a point ILU0 is a weaker second stage than the block ILU0 of the same system, never a broken one
        assert out[(2, gmres)][1] <= 3 * out[(1, gmres)][1] + 5, {k: v[1] for k, v in out.items()}
    # float solves are refused in this mode (the reference's CPR plug-in is double only)
    m = GpuBlackoilModel(grid, tab, capi.default_params(cpr_reference_transform=2, **capi.CPR_AMG_VCYCLE))
    m.prepareStep(5 * decks.DAY, st); m.assemble(True); m.getConvergence()
    with pytest.raises(Exception, match="double only"):
        m.solveJacobianSystem(single_precision=True)
    m.close()


def _ilun_numpy(rowptr, col, val, n, d, relax):
    """Block ILU(n) with level-of-fill as csrc/fillilu.inl states it: lev(i,j) = min over eliminations of lev(i,k) + lev(k,j) + 1, kept when <= n,
    rows in the caller's order, IKJ numeric phase on that pattern, pivots inverted; v = relax * (LU)^-1 d."""
    nb = rowptr.size - 1
    gen = []
    for i in range(nb):
        pat = {int(col[s]): 0 for s in range(rowptr[i], rowptr[i + 1])}
        done = set()
        while True:
            ks = sorted(k for k in pat if k < i and k not in done)
            if not ks:
                break
            k = ks[0]; done.add(k)
            for j, g in gen[k].items():
                if j > k and pat[k] + g + 1 <= n:
                    pat[j] = min(pat.get(j, 99), pat[k] + g + 1)
        gen.append(pat)
    B = [{j: np.zeros((3, 3)) for j in gen[i]} for i in range(nb)]
    for i in range(nb):
        for s in range(rowptr[i], rowptr[i + 1]):
            B[i][int(col[s])] = np.array(val[s], float).reshape(3, 3)
    for i in range(nb):
        for k in sorted(j for j in B[i] if j < i):
            B[i][k] = B[i][k] @ np.linalg.inv(B[k][k])
            for j in B[k]:
                if j > k and j in B[i]:
                    B[i][j] = B[i][j] - B[i][k] @ B[k][j]
    y = relax * d.reshape(nb, 3).copy()
    for i in range(nb):
        for k in B[i]:
            if k < i:
                y[i] -= B[i][k] @ y[k]
    for i in range(nb - 1, -1, -1):
        for j in B[i]:
            if j > i:
                y[i] -= B[i][j] @ y[j]
        y[i] = np.linalg.solve(B[i][i], y[i])
    return y.ravel(), sum(len(g) for g in gen)


@pytest.mark.parametrize("single", [False, True])
def test_block_ilu_n_against_numpy(gpu_lib, oracle, single):
    """ilu_fillin_level = n (ISTLSolver.hpp:205) / cpr_ilu_n (NewtonIterationBlackoilCPR.hpp:61): block ILU(n) with level-of-fill, csrc/fillilu.inl.
    In the natural (dune's) elimination order the application equals a numpy restatement of dune's ILU(n) on every pattern of the ILU0 parity
    test; n = 0 of the restatement is the ILU0 the library already pins against the oracle.  Multicolour ordering eliminates the same
    pattern in another order: checked through the solver (same solution, no more iterations than the ILU0)."""
    tol = 5e-5 if single else 1e-11
    for name, (rowptr, col) in _patterns(oracle):
        nb = rowptr.size - 1
        val = random_block_matrix(rowptr, col, seed=nb)
        d = np.random.default_rng(nb + 7).uniform(-1, 1, 3 * nb)
        for n in (0, 1, 2):
            s = GpuNewtonIteration(capi.default_params(ilu_ordering=capi.ORDER_NATURAL, ilu_fillin_level=n))
            s.load(rowptr, col, val, single)
            s.ilu0_factor()
            v = s.ilu0_apply(d)
            vo, nnz = _ilun_numpy(rowptr, col, val, n, d, 0.9)
            assert rel_err(v, vo) < tol, (name, n, rel_err(v, vo))
            if n == 0:
                assert nnz == col.size
            else:
                assert nnz > col.size
                with pytest.raises(Exception, match="pattern"):
                    s.ilu0_get(col.size)
            s.close()
    # the solver with it: same solution, fewer iterations with every level (natural order); multicolour: same solution, not more than the ILU0
    name, (rowptr, col) = _patterns(oracle)[0]
    nb = rowptr.size - 1
    val = random_block_matrix(rowptr, col, seed=5, dominance=2.2)
    xt = np.random.default_rng(1).uniform(-1, 1, 3 * nb)
    b = bsr_to_scipy(rowptr, col, val) @ xt
    for ordering in (capi.ORDER_NATURAL, capi.ORDER_MULTICOLOR):
        its = []
        for n in (0, 1, 2):
            for gmres in (0, 1):
                s = GpuNewtonIteration(capi.default_params(ilu_ordering=ordering, ilu_fillin_level=n, newton_use_gmres=gmres,
                                                           linear_solver_reduction=1e-6 if single else 1e-11, linear_solver_maxiter=300))
                try:
                    x = s.computeNewtonIncrement(rowptr, col, val, b, single)
                except Exception as e:
                    raise AssertionError((ordering, n, gmres, s.iterations(), s.reduction, str(e)))
                assert rel_err(x, xt) < (2e-3 if single else 1e-8), (ordering, n, gmres, rel_err(x, xt))
                if gmres == 0:
                    its.append(s.iterations())
                s.close()
        print("ILU(n) iterations, ordering %d: %s" % (ordering, its))
        assert its[1] <= its[0] and its[2] <= its[1] + 1, its
    with pytest.raises(Exception):
        s = GpuNewtonIteration(capi.default_params(ilu_fillin_level=9))
        s.computeNewtonIncrement(rowptr, col, val, b, single)


@pytest.mark.parametrize("cpr", [0, 1])
def test_float_preconditioner_inside_a_double_solve(gpu_lib, oracle, cpr):
    """opmgpu_params.preconditioner_single (library extension): the double Krylov method with its preconditioner built and applied in float.
    The solve is still a solve of the DOUBLE system: BiCGStab reaches 1e-8 on the true residual (checked with an independent product;
    the documented range of the option -- a float M^-1 is not one fixed linear operator, and the recurrences stall around 1e-9 .. 1e-10
    on harder matrices: tools/fuzz_newton.py case 4080) and the direct solution; GMRES at the reductions Newton solves ask for."""
    grid = decks.cartesian_grid(12, 10, 8, lognormal_sigma=0.8)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, perturb=0.01)
    prm0 = capi.default_params()
    scale = np.asarray(prm0.matbalscale[:])
    rowptr, col = oracle.pattern(grid)
    nc = grid.nc
    r, val, _, _ = oracle.assemble(grid, tab, 5 * decks.DAY, st, rowptr, col, scale=tuple(scale))
    b = np.ascontiguousarray((r * np.repeat(scale, nc)).reshape(3, nc).T).ravel()
    A = bsr_to_scipy(rowptr, col, val)
    xe = spla.spsolve(A.tocsc(), b)
    kw = dict(capi.CPR_AMG_VCYCLE) if cpr else dict()
    its = {}
    for mixed in (0, 1):
        s = GpuNewtonIteration(capi.default_params(preconditioner_single=mixed, linear_solver_reduction=1e-8, linear_solver_maxiter=400, **kw))
        x = s.computeNewtonIncrement(rowptr, col, val, b, False)
        its[mixed] = s.iterations()
        assert np.linalg.norm(b - A @ x) <= 1.01e-8 * np.linalg.norm(b), (cpr, mixed)
        assert np.linalg.norm(x - xe) <= 1e-5 * np.linalg.norm(xe), (cpr, mixed)
        s.close()
    assert its[1] <= its[0] + max(2, its[0] // 5), its          # a float preconditioner is as good a preconditioner
    for red in (1e-2, 1e-5):
        s = GpuNewtonIteration(capi.default_params(preconditioner_single=1, newton_use_gmres=1, linear_solver_reduction=red, linear_solver_maxiter=200, **kw))
        x = s.computeNewtonIncrement(rowptr, col, val, b, False)
        s0 = GpuNewtonIteration(capi.default_params(newton_use_gmres=1, linear_solver_reduction=red, linear_solver_maxiter=200, **kw))
        x0 = s0.computeNewtonIncrement(rowptr, col, val, b, False)
        assert abs(s.iterations() - s0.iterations()) <= 1, (cpr, red, s.iterations(), s0.iterations())
        assert np.linalg.norm(x - x0) <= 20 * red * np.linalg.norm(x0), (cpr, red)
        s.close(); s0.close()
    # a FLOAT solve ignores the switch (its preconditioner is float anyway)
    s = GpuNewtonIteration(capi.default_params(preconditioner_single=1, linear_solver_reduction=1e-4, linear_solver_maxiter=200, **kw))
    x = s.computeNewtonIncrement(rowptr, col, val, b, True)
    assert np.linalg.norm(b - A @ x) <= 2e-4 * np.linalg.norm(b)
    s.close()


def test_gmres_verify_with_float_vectors(gpu_lib, oracle, monkeypatch):
    """gmres_verify_residual with a FLOAT solve: the documented range -- reductions down to 1e-3 -- is met on the TRUE residual
    (checked with an independent double product), also with the speculative product of the next column enqueued ahead of the verdict
    (OPMGPU_GMRES_SPECULATE=1: a verify round that takes `done` back must not reuse a product that returned at once); beyond what float
    rounding of b - A x allows the solve reports LinearSolverProblem instead of a wrong answer."""
    grid = decks.cartesian_grid(12, 10, 8, lognormal_sigma=0.8)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, perturb=0.01)
    prm0 = capi.default_params()
    scale = np.asarray(prm0.matbalscale[:])
    rowptr, col = oracle.pattern(grid)
    nc = grid.nc
    r, val, _, _ = oracle.assemble(grid, tab, 5 * decks.DAY, st, rowptr, col, scale=tuple(scale))
    b = np.ascontiguousarray((r * np.repeat(scale, nc)).reshape(3, nc).T).ravel()
    A = bsr_to_scipy(rowptr, col, val)
    for speculate in ("0", "1"):
        monkeypatch.setenv("OPMGPU_GMRES_SPECULATE", speculate)
        for kw in (dict(), dict(capi.CPR_AMG_VCYCLE)):
            for red in (1e-2, 1e-3):
                s = GpuNewtonIteration(capi.default_params(newton_use_gmres=1, gmres_verify_residual=1, linear_solver_reduction=red, linear_solver_maxiter=300, **kw))
                x = s.computeNewtonIncrement(rowptr, col, val, b, True)
                true_red = np.linalg.norm(b - A @ x) / np.linalg.norm(b)
                assert true_red <= 1.05 * red, (speculate, kw, red, true_red)          # (5 %: the check itself is a float product)
                assert s.reduction <= red
                s.close()
    monkeypatch.delenv("OPMGPU_GMRES_SPECULATE")
    # far below float rounding: an error, never a silently unverified answer
    s = GpuNewtonIteration(capi.default_params(newton_use_gmres=1, gmres_verify_residual=1, linear_solver_reduction=1e-12, linear_solver_maxiter=60))
    with pytest.raises(LinearSolverProblem):
        s.computeNewtonIncrement(rowptr, col, val, b, True)
    s.close()


@pytest.mark.parametrize("single", [False, True])
def test_flexible_gmres_stops_on_the_true_residual(gpu_lib, oracle, single):
    """newton_use_gmres = 2 (not a reference solver): right-preconditioned GMRES with the preconditioned basis kept.
    Its stopping quantity is the TRUE residual (the criterion of the reference's default BiCGStab), so ||b - A x|| <= reduction ||b||
    must hold for what comes back; ILU0 and CPR as the preconditioner, a restart shorter than the iteration count."""
    grid = decks.cartesian_grid(12, 10, 8, lognormal_sigma=0.8)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, perturb=0.01)
    prm0 = capi.default_params()
    scale = np.asarray(prm0.matbalscale[:])
    rowptr, col = oracle.pattern(grid)
    nc = grid.nc
    r, val, _, _ = oracle.assemble(grid, tab, 5 * decks.DAY, st, rowptr, col, scale=tuple(scale))
    b = np.ascontiguousarray((r * np.repeat(scale, nc)).reshape(3, nc).T).ravel()
    A = bsr_to_scipy(rowptr, col, val)
    red = 1e-4 if single else 1e-6
    slack = 20.0 if single else 1.0 + 1e-6           # float: the recurrence residual and the true one drift apart by rounding
    for kw in (dict(ilu_ordering=capi.ORDER_NATURAL), dict(ilu_ordering=capi.ORDER_NATURAL, linear_solver_restart=12), dict(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1)):
        s = GpuNewtonIteration(capi.default_params(newton_use_gmres=2, linear_solver_reduction=red, linear_solver_maxiter=400, **kw))
        try:
            x = s.computeNewtonIncrement(rowptr, col, val, b, single)
        except LinearSolverProblem:
            pytest.fail("no convergence with %r after %d iterations, reduction %.2e" % (kw, s.iterations(), s.reduction))
        assert s.reduction < red
        assert np.linalg.norm(b - A @ x) <= slack * red * np.linalg.norm(b), kw
        if "linear_solver_restart" in kw:
            assert s.iterations() > 12                       # the restart path ran
        s.close()
    s = GpuNewtonIteration(capi.default_params(newton_use_gmres=2, linear_solver_reduction=1e-12, linear_solver_maxiter=3))
    with pytest.raises(LinearSolverProblem):
        s.computeNewtonIncrement(rowptr, col, val, b, single)
    s.close()


@pytest.mark.parametrize("single", [False, True])
def test_gmres_true_residual_check(gpu_lib, oracle, single):
    """opmgpu_params.gmres_verify_residual: Dune's left-preconditioned GMRES stops on ||M^-1 (b - A x)||; with the check the solve is only
    reported converged when ||b - A x|| <= reduction ||b|| holds as well (the statement the reference's default BiCGStab makes), the
    iteration continuing from the true defect otherwise, and the reported reduction is the true one.  Without the check the same solver
    leaves a larger true residual on the heterogeneous deck (which is what chopped a time step of the 200^3 bench deck in round 2)."""
    grid = decks.cartesian_grid(12, 22, 17, dx=6.096, dy=3.048, dz=0.6096, tops=3657.6, lognormal_sigma=2.5, seed=10)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, p_ref=413.0 * decks.BAR, z_ref=3657.6, perturb=0.005, seed=10)
    prm0 = capi.default_params()
    scale = np.asarray(prm0.matbalscale[:])
    rowptr, col = oracle.pattern(grid)
    nc = grid.nc
    r, val, _, _ = oracle.assemble(grid, tab, 2 * decks.DAY, st, rowptr, col, scale=tuple(scale))
    b = np.ascontiguousarray((r * np.repeat(scale, nc)).reshape(3, nc).T).ravel()
    A = bsr_to_scipy(rowptr, col, val)
    red = 1e-3 if single else 1e-5
    slack = 3.0 if single else 1.0 + 1e-6            # float: x is rounded to 24 bits after the check measured it in the solve's precision
    true_res = {}
    for kw in (dict(ilu_ordering=capi.ORDER_NATURAL), dict(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1), dict(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1, linear_solver_restart=5)):
        for verify in (0, 1):
            s = GpuNewtonIteration(capi.default_params(newton_use_gmres=1, gmres_verify_residual=verify, linear_solver_reduction=red, linear_solver_maxiter=1500, **kw))
            try:
                x = s.computeNewtonIncrement(rowptr, col, val, b, single)
            except LinearSolverProblem:
                pytest.fail("no convergence with %r, verify %d, after %d iterations, reduction %.2e" % (kw, verify, s.iterations(), s.reduction))
            true_res[verify] = np.linalg.norm(b - A @ x) / np.linalg.norm(b)
            assert s.reduction < red
            if verify:
                assert true_res[1] <= slack * red, (kw, true_res)
                assert s.reduction == pytest.approx(true_res[1], rel=0.5 if single else 1e-3), kw      # the reported number IS the true reduction
            its = s.iterations()
            s.close()
        assert true_res[1] <= true_res[0] * (1 + 1e-6), (kw, true_res)
    # a right-hand side of zero is converged at once with or without the check
    s = GpuNewtonIteration(capi.default_params(newton_use_gmres=1, gmres_verify_residual=1))
    x = s.computeNewtonIncrement(rowptr, col, val, np.zeros_like(b), single)
    assert np.all(x == 0) and s.iterations() == 0
    s.close()


@pytest.mark.parametrize("gmres", [0, 1])
@pytest.mark.parametrize("single", [False, True])
def test_reference_cpr_formulation_gives_the_same_solution(gpu_lib, oracle, gmres, single):
    """opmgpu_params.cpr_reference_transform: the reference's CPR formulation -- the whole system row-transformed by formEllipticSystem's
    per-cell L (NewtonIterationUtilities.cpp:253-287), the pressure row scaled by 200 bar (NewtonIterationBlackoilCPR.cpp:117-121), the
    Krylov method iterating on L A x = L b -- must return the solution of the untransformed system: both against a sparse direct solve.
    (A maintainer with an OPM install compares iteration counts one to one with this option; the default measures ||r|| instead of ||L r||.)"""
    grid = decks.cartesian_grid(12, 10, 8, lognormal_sigma=1.2, seed=4)
    tab = decks.satfunc_standard_tables()
    st = decks.random_state(grid, tab, seed=4)            # every phase state occurs: cells with a weak oil equation among them
    prm0 = capi.default_params()
    scale = np.asarray(prm0.matbalscale[:])
    rowptr, col = oracle.pattern(grid)
    nc = grid.nc
    r, val, _, _ = oracle.assemble(grid, tab, 5 * decks.DAY, st, rowptr, col, scale=tuple(scale))
    b = np.ascontiguousarray((r * np.repeat(scale, nc)).reshape(3, nc).T).ravel()
    A = bsr_to_scipy(rowptr, col, val)
    xe = spla.spsolve(A.tocsc(), b)
    red = 1e-4 if single else 1e-11
    sol, its = {}, {}
    for tr in (0, 1):
        s = GpuNewtonIteration(capi.default_params(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1, newton_use_gmres=gmres, cpr_reference_transform=tr, linear_solver_reduction=red, linear_solver_maxiter=300))
        sol[tr] = s.computeNewtonIncrement(rowptr, col, val, b, single)
        its[tr] = s.iterations()
        assert s.reduction < red and 1 <= its[tr] < 300, (tr, its)
        w = np.zeros(3 * nc)
        assert s.lib.opmgpu_get_cpr_weights(s.ctx, capi.dptr(w)) == capi.OK
        if tr:          # the pressure stage reads the transformed system's first row
            assert np.array_equal(w.reshape(3, nc), np.stack([np.ones(nc), np.zeros(nc), np.zeros(nc)]))
        s.close()
    tol = 6e-2 if single else 1e-6                        # (float: the attainable error is cond(A) eps of either system; measured 2.5e-2 untransformed)
    for tr in (0, 1):
        assert np.linalg.norm(sol[tr] - xe) <= tol * np.linalg.norm(xe), (tr, its)
    assert np.linalg.norm(sol[1] - sol[0]) <= tol * np.linalg.norm(xe)


def test_global_coarse_space_restores_convergence_of_decomposed_preconditioner(gpu_lib, monkeypatch):
    """The CPR pressure stage's global coarse space (one unknown per subdomain).  OPMGPU_EMULATE_RANKS builds the preconditioner
    as a 4-rank run would (no coupling across the cuts in the ILU0's and the AMG's matrix): without the coarse space the
    BiCGStab count grows, with it it is back at the single-domain count; the solution is the same in all cases."""
    from opmgpu.model import GpuBlackoilModel
    grid = decks.cartesian_grid(24, 24, 96, lognormal_sigma=0.5)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, perturb=0.002)

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        # (cpr_stage2_relax = 0.9: the damped second stage the thresholds below were measured with in rounds 1-3; with the reference's undamped
        # form the same comparison reads 31 / 59 / 39 instead of 31 / 59 / <= 38 -- the coarse space repairs the cut either way)
        m = GpuBlackoilModel(grid, tab, capi.default_params(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1, cpr_stage2_relax=0.9, linear_solver_reduction=1e-6, linear_solver_maxiter=200))
        m.prepareStep(5 * decks.DAY, st)
        its = 0
        for it in range(3):
            m.assemble(it == 0); m.getConvergence()
            dx = m.solveJacobianSystem(want_dx=True, single_precision=False)
            its += m.linear_iterations
            m.updateState()
        out = m.getState()
        m.close()
        for k in env:
            monkeypatch.delenv(k)
        return its, out

    base_its, base = run({"OPMGPU_COARSE": "0"})
    cut_its, cut = run({"OPMGPU_COARSE": "0", "OPMGPU_EMULATE_RANKS": "4"})
    fix_its, fix = run({"OPMGPU_COARSE": "1", "OPMGPU_EMULATE_RANKS": "4"})
    one_its, one = run({"OPMGPU_COARSE": "1"})
    assert cut_its > 1.3 * base_its, (base_its, cut_its)                 # the decomposition hurts ...
    assert fix_its <= 1.25 * base_its and fix_its <= 0.7 * cut_its, (base_its, cut_its, fix_its)     # ... the coarse space repairs it
    assert one_its < base_its, (base_its, one_its)                        # and the global constant alone already helps
    for s in (cut, fix, one):
        # same Newton path up to the linear tolerance (1e-6 on the residual, three iterations)
        assert np.abs(s.p - base.p).max() <= 1e-5 * np.abs(base.p).max() and np.abs(s.sat - base.sat).max() <= 3e-4


@pytest.mark.gpu
def test_default_precision_follows_the_reference_plugins(gpu_lib):
    """residual_.singlePrecision = dt < 20 d (BlackoilModelBase_impl.hpp:284) reaches the interleaved solver only
    (NewtonIterationBlackoilInterleaved.cpp:478-480); the CPR plug-in computes in double whatever it says (NewtonIterationBlackoilCPR.cpp:117-140).
    The host mirror's default (single_precision=None) follows that."""
    from opmgpu.model import GpuBlackoilModel
    g = decks.cartesian_grid(4, 4, 3)
    t = decks.satfunc_standard_tables()
    st = decks.initial_state(g, t)
    for use_cpr, dt_days, want in ((0, 1.0, True), (0, 19.9, True), (0, 20.0, False), (1, 1.0, False), (1, 30.0, False)):
        m = GpuBlackoilModel(g, t, capi.default_params(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=use_cpr))
        m.prepareStep(dt_days * decks.DAY, st)
        assert m.referencePrecision() is want, (use_cpr, dt_days)
        m.close()
