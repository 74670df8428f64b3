"""Independent checks of the CPU oracle itself (finite differences, exact solves)."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from opmgpu import capi, decks


def _bsr_to_scipy(rowptr, col, val9):
    nb = rowptr.size - 1
    return sp.bsr_matrix((val9.reshape(-1, 3, 3), col, rowptr), shape=(3 * nb, 3 * nb)).tocsr()


def _eqmajor_to_interleaved(v, nc):
    return np.ascontiguousarray(v.reshape(3, nc).T).ravel()


ROCKTAB = [(100.0, 0.98, 0.95), (200.0, 1.0, 1.0), (300.0, 1.03, 1.08), (500.0, 1.05, 1.12)]


@pytest.mark.parametrize("variant", ["plain", "endscale", "vappars+rocktab"])
def test_fd_jacobian(oracle, variant):
    """Analytic (forward-AD) Jacobian vs central differences of the residual."""
    grid = decks.cartesian_grid(4, 3, 3, lognormal_sigma=0.5)
    tab = decks.satfunc_standard_tables()
    nc = grid.nc
    if variant == "endscale":
        grid = decks.with_endpoints(grid, decks.random_endpoints(grid, seed=5))
    if variant == "vappars+rocktab":
        tab = decks.satfunc_standard_tables(vappars=(0.7, 1.3), rocktab=ROCKTAB)
        oracle.set_sat_oil_max(np.random.default_rng(2).uniform(0.3, 0.9, nc))
    try:
        _fd_jacobian(oracle, grid, tab)
    finally:
        oracle.set_sat_oil_max(None)


def _fd_jacobian(oracle, grid, tab):
    st = decks.random_state(grid, tab, seed=3, breakpoints=False)
    nc = grid.nc
    dt = 5 * decks.DAY
    rowptr, col = oracle.pattern(grid)
    r0, val, acc0, _ = oracle.assemble(grid, tab, dt, st, rowptr, col)
    J = _bsr_to_scipy(rowptr, col, val).toarray()          # interleaved rows/cols: 3*cell + eq/var

    def resid(s):
        r, _, _, _ = oracle.assemble(grid, tab, dt, s, rowptr, col, accum0=acc0)
        return _eqmajor_to_interleaved(r, nc)

    worst = 0.0
    for c in range(nc):
        for k in range(3):
            hs = {0: 1e-3 * decks.BAR, 1: 1e-7, 2: 1e-7}
            h = hs[k]
            sp_, sm = st.copy(), st.copy()
            if k == 0:
                sp_.p[c] += h; sm.p[c] -= h
            elif k == 1:
                sp_.sat[c, 0] += h; sm.sat[c, 0] -= h
            else:
                hc = st.hc[c]
                if hc == capi.HC_GAS_AND_OIL:
                    sp_.sat[c, 2] += h; sm.sat[c, 2] -= h
                elif hc == capi.HC_OIL_ONLY:
                    h = 1e-5; sp_.rs[c] += h; sm.rs[c] -= h
                else:
                    h = 1e-9; sp_.rv[c] += h; sm.rv[c] -= h
            fd = (resid(sp_) - resid(sm)) / (2 * h)
            an = J[:, 3 * c + k]
            scale = np.abs(an).max() + 1e-30
            worst = max(worst, np.abs(fd - an).max() / scale)
    assert worst < 2e-5, worst


def test_ilu0_exact_on_chain(oracle):
    """On a 1-D chain ILU(0) has no dropped fill: apply(relax=1) must solve exactly."""
    nb = 40
    rng = np.random.default_rng(0)
    rows, cols = [], []
    for i in range(nb):
        for j in (i - 1, i, i + 1):
            if 0 <= j < nb:
                rows.append(i); cols.append(j)
    rowptr = np.zeros(nb + 1, np.int32); np.add.at(rowptr, np.asarray(rows) + 1, 1); rowptr = np.cumsum(rowptr).astype(np.int32)
    col = np.asarray(cols, np.int32)
    val = rng.uniform(-0.3, 0.3, (col.size, 9))
    for s in range(col.size):
        if col[s] == rows[s]:
            val[s] += np.eye(3).ravel() * 4
    A = _bsr_to_scipy(rowptr, col, val)
    b = rng.standard_normal(3 * nb)
    for pos in (None, rng.permutation(nb).astype(np.int32)):
        st, lu = oracle.ilu0(rowptr, col, val, position=pos)
        assert st == 0
        if pos is None:       # natural order on a chain: exact
            x = oracle.ilu0_apply(rowptr, col, lu, b, position=pos, relax=1.0)
            assert np.allclose(A @ x, b, rtol=1e-11, atol=1e-11)
    # float path agrees with double to float accuracy
    st, luf = oracle.ilu0(rowptr, col, val, single=True)
    st, lud = oracle.ilu0(rowptr, col, val)
    assert np.allclose(luf, lud, rtol=2e-5, atol=2e-6)


def test_bicgstab_against_direct(oracle):
    grid = decks.cartesian_grid(6, 5, 4, lognormal_sigma=1.0)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, perturb=0.01)
    prm = capi.default_params()
    rowptr, col = oracle.pattern(grid)
    r, val, _, _ = oracle.assemble(grid, tab, 10 * decks.DAY, st, rowptr, col, scale=tuple(prm.matbalscale))
    nc = grid.nc
    b = _eqmajor_to_interleaved(r * np.repeat(np.asarray(prm.matbalscale[:]), nc), nc)
    A = _bsr_to_scipy(rowptr, col, val)
    xe = spla.spsolve(A.tocsc(), b)
    for single in (False, True):
        prm2 = capi.default_params(linear_solver_reduction=1e-6 if single else 1e-10, linear_solver_maxiter=200)
        status, x, it, red, hist = oracle.bicgstab(rowptr, col, val, b, prm2, single=single, nhist=400)
        assert status == 0 and it > 0
        res = np.linalg.norm(A @ x - b) / np.linalg.norm(b)
        assert res < (5e-5 if single else 1e-9), res
        assert np.linalg.norm(x - xe) / np.linalg.norm(xe) < (1e-3 if single else 1e-7)
        assert np.all(np.diff(hist) < 1e300)


def test_update_state_invariants(oracle):
    grid = decks.cartesian_grid(5, 4, 3)
    tab = decks.satfunc_standard_tables()
    st = decks.random_state(grid, tab, seed=11)
    prm = capi.default_params()
    rng = np.random.default_rng(5)
    nc = grid.nc
    dx = np.concatenate([rng.standard_normal(nc) * 40 * decks.BAR, rng.standard_normal(nc) * 0.3, rng.standard_normal(nc) * 0.3])
    new = oracle.update_state(grid, tab, prm, dx, st)
    assert np.all(new.p >= 0) and np.all(np.abs(new.p - st.p) <= 0.3 * np.abs(st.p) * (1 + 1e-12))
    assert np.all(new.sat >= 0) and np.all(new.sat <= 1 + 1e-12)
    assert set(np.unique(new.hc)) <= {0, 1, 2}
    assert np.all(new.rs >= 0)      # rv may follow a (linearly extrapolated) negative rvSat, as in the reference
    # zero increment keeps pressures and saturations
    same = oracle.update_state(grid, tab, prm, np.zeros(3 * nc), st)
    assert np.array_equal(same.p, st.p)


def test_threads_do_not_change_a_bit(oracle):
    """The checker's OpenMP paths are bit-reproducible and independent of the thread count: dot products are summed in
    fixed blocks, the ILU0 factorisation and sweeps run by levels whose rows do exactly the sequential row's operations.  One thread keeps
    the plain sequential sum (dune's SeqScalarProduct), so 1 vs n threads may differ in the dot products' rounding -- n vs m threads and
    repeated runs must not, and the factors must be identical for every count."""
    grid = decks.cartesian_grid(14, 12, 9, lognormal_sigma=1.5, seed=3)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, perturb=0.01, seed=3)
    prm = capi.default_params(linear_solver_reduction=1e-9, linear_solver_maxiter=300)
    rowptr, col = oracle.pattern(grid)
    r, val, _, _ = oracle.assemble(grid, tab, 10 * decks.DAY, st, rowptr, col, scale=tuple(prm.matbalscale))
    nc = grid.nc
    b = _eqmajor_to_interleaved(r * np.repeat(np.asarray(prm.matbalscale[:]), nc), nc)
    rng = np.random.default_rng(0)
    perm = rng.permutation(nc).astype(np.int32)            # an arbitrary elimination order: many short levels
    out = {}
    try:
        for nt in (1, 2, 5, 8, 8):
            oracle.set_threads(nt)
            for tag, pos in (("nat", None), ("perm", perm)):
                stf, lu = oracle.ilu0(rowptr, col, val, position=pos)
                sts, x, it, red, hist = oracle.bicgstab(rowptr, col, val, b, prm, position=pos, nhist=600)
                assert stf == 0 and sts == 0
                v = oracle.ilu0_apply(rowptr, col, lu, b, position=pos)
                out.setdefault(tag, []).append((nt, lu, x, it, hist, v))
    finally:
        oracle.set_threads(1)
    for tag, runs in out.items():
        base = runs[0]
        for nt, lu, x, it, hist, v in runs[1:]:
            assert np.array_equal(lu, base[1]), (tag, nt)                       # factors: identical for every thread count
            assert np.array_equal(v, base[5]), (tag, nt)
        multi = [r_ for r_ in runs if r_[0] > 1]
        for nt, lu, x, it, hist, v in multi[1:]:
            assert it == multi[0][3] and np.array_equal(x, multi[0][2]) and np.array_equal(hist, multi[0][4]), (tag, nt)
        # one thread against many: the same solve up to the dot products' rounding
        assert abs(base[3] - multi[0][3]) <= 1 and np.abs(base[2] - multi[0][2]).max() <= 1e-6 * np.abs(base[2]).max(), tag
