"""Multi-GPU code path exercised on ONE GPU: a z-periodic deck solved (A) as a plain grid whose
wrap-around faces are ordinary connections and (B) as a one-rank RCCL "decomposition" whose wrap-around
neighbours are ghost copies refreshed by a halo exchange with itself (grouped ncclSend/ncclRecv to
self).  B runs every multi-GPU mechanism -- owner mask, identity ghost rows, halo pack/exchange/unpack
before each SpMV, partial-sum bridge + all-reduce, deterministic stop rule, block-Jacobi ILU0 -- and
must reproduce A."""
import numpy as np
import pytest

from opmgpu import capi, decks, partition
from opmgpu.model import GpuBlackoilModel
from util import rel_err

pytestmark = pytest.mark.gpu


def _periodic_pair(nx=6, ny=5, nz=4):
    base = decks.cartesian_grid(nx, ny, nz, lognormal_sigma=0.4)
    n, L = base.nc, nx * ny
    bottom, top = np.arange(L), np.arange(n - L, n)           # k = 0 layer and k = nz-1 layer
    twrap = np.full(L, np.median(base.trans))
    # A: wrap faces as extra connections (top cell is c1)
    connA = np.concatenate([base.conn_cells, np.stack([top, bottom], 1)])
    gridA = decks.GridData(n, connA, np.concatenate([base.trans, twrap]), base.pv, base.z)
    # B: owned cells + ghost copies [bottom copies | top copies]
    gb, gt = n + np.arange(L), n + L + np.arange(L)
    connB = np.concatenate([base.conn_cells, np.stack([top, gb], 1), np.stack([gt, bottom], 1)])
    pv = np.concatenate([base.pv, base.pv[bottom], base.pv[top]])
    z = np.concatenate([base.z, base.z[bottom], base.z[top]])
    gridB = decks.GridData(n + 2 * L, connB, np.concatenate([base.trans, twrap, twrap]), pv, z)
    src = np.concatenate([np.arange(n), bottom, top])          # original cell of every local cell
    halo = dict(n_owned=n, neigh_rank=capi.i32([0]), send_ptr=capi.i32([0, 2 * L]), send_cells=capi.i32(np.concatenate([bottom, top])),
                recv_ptr=capi.i32([0, 2 * L]), recv_cells=capi.i32(n + np.arange(2 * L)))
    return gridA, gridB, src, halo


class _Dom:
    pass


@pytest.mark.parametrize("cpr", [0, 1, 2])
@pytest.mark.parametrize("single", [False, True])
def test_self_halo_reproduces_plain_grid(gpu_lib, oracle, single, cpr, monkeypatch):
    # cpr == 2: CPR with the global coarse space of the pressure stage forced on although there is only one subdomain
    # (OPMGPU_COARSE=2): its multi-GPU kernels (deterministic partial sums, all-reduces, per-subdomain correction) all run
    if cpr == 2:
        monkeypatch.setenv("OPMGPU_COARSE", "2")
        cpr = 1
    gridA, gridB, src, halo = _periodic_pair()
    tab = decks.satfunc_standard_tables()
    stA = decks.initial_state(gridA, tab, perturb=0.01)
    stB = decks.State(stA.p[src], stA.sat[src], stA.rs[src], stA.rv[src], stA.hc[src])
    red = 1e-4 if single else 1e-11
    prm = capi.default_params(linear_solver_reduction=red, linear_solver_maxiter=400, cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=cpr)
    dt = 5 * decks.DAY
    n = gridA.nc
    A = GpuBlackoilModel(gridA, tab, prm)
    B = GpuBlackoilModel(gridB, tab, prm)
    dom = _Dom()
    for k, v in halo.items():
        setattr(dom, k, v)
    partition.attach_comm(B, dom, 0, 1, partition.make_unique_id())
    A.prepareStep(dt, stA)
    B.prepareStep(dt, stB)
    for it in range(3):
        A.assemble(it == 0); B.assemble(it == 0)
        # owned rows of B == rows of A (ghost column g <-> its original cell), ghost rows = identity, zero residual
        rA, rB = A.residual(), B.residual()
        nB = gridB.nc
        # identical states only at it == 0; afterwards A and B differ by the linear tolerance (different ILU0)
        tol = 1e-12 if it == 0 else (1e-2 if single else 1e-6)
        for a in range(3):
            assert rel_err(rB[a * nB:a * nB + n], rA[a * n:(a + 1) * n]) < tol
            assert np.all(rB[a * nB + n:(a + 1) * nB] == 0.0)
        rpA, clA, vA = A.jacobian(); rpB, clB, vB = B.jacobian()
        for i in range(n if it == 0 else 0):
            ca = {int(c): vA[k] for k, c in zip(range(rpA[i], rpA[i + 1]), clA[rpA[i]:rpA[i + 1]])}
            cb = {int(src[c]): vB[k] for k, c in zip(range(rpB[i], rpB[i + 1]), clB[rpB[i]:rpB[i + 1]])}
            assert ca.keys() == cb.keys()
            for c in ca:
                assert np.allclose(ca[c], cb[c], rtol=1e-12, atol=1e-14 * np.abs(vA).max())
        for i in range(n, nB):
            blk = {int(c): vB[k] for k, c in zip(range(rpB[i], rpB[i + 1]), clB[rpB[i]:rpB[i + 1]])}
            assert np.array_equal(blk.pop(i), np.eye(3).ravel()) and all(np.all(b == 0) for b in blk.values())
        cA, cB = A.getConvergence(), B.getConvergence()
        if it == 0:
            assert cA == cB and np.allclose(A.CNV, B.CNV, rtol=1e-10) and np.allclose(A.MB, B.MB, rtol=1e-6, atol=1e-12) and np.allclose(A.B_avg, B.B_avg, rtol=1e-12)
        dxA = A.solveJacobianSystem(want_dx=True, single_precision=single)
        dxB = B.solveJacobianSystem(want_dx=True, single_precision=single)
        assert B.linear_iterations >= 1 and B.linear_reduction < red
        print("self-halo cpr=%d single=%d it %d: linear iterations plain %d, decomposed %d" % (cpr, single, it, A.linear_iterations, B.linear_iterations))
        if not single:
            for a in range(3):
                blkA, blkB = dxA[a * n:(a + 1) * n], dxB[a * nB:a * nB + n]
                assert np.abs(blkA - blkB).max() <= 1e-6 * np.abs(blkA).max() + 1e-300, (it, a)
                # ghosts carry their owner's increment bit for bit (halo-exchanged search directions)
                assert np.array_equal(dxB[a * nB + n:(a + 1) * nB], dxB[a * nB:a * nB + n][src[n:]])
        A.updateState(); B.updateState()
        sa, sb = A.getState(), B.getState()
        if not single:
            assert np.array_equal(sa.hc, sb.hc[:n])
            assert np.abs(sa.p - sb.p[:n]).max() <= 1e-6 * np.abs(sa.p).max() and np.abs(sa.sat - sb.sat[:n]).max() <= 1e-6
            # ghost state stays a copy of the owner's state without any state exchange
            assert np.array_equal(sb.p[n:], sb.p[src[n:]]) and np.array_equal(sb.sat[n:], sb.sat[src[n:]]) and np.array_equal(sb.hc[n:], sb.hc[src[n:]])
    A.close(); B.close()


@pytest.mark.parametrize("cpr", [0, 1])
def test_self_halo_with_device_wells(gpu_lib, cpr):
    """Wells in multi-GPU mode live on one rank (all perforated cells owned): the one-rank decomposition with ghost copies and
    device wells must walk the same Newton path as the plain periodic grid with the same wells."""
    from opmgpu import wells as W
    gridA, gridB, src, halo = _periodic_pair(nx=6, ny=5, nz=5)
    tab = decks.satfunc_standard_tables()
    stA = decks.initial_state(gridA, tab, perturb=0.005)
    stB = decks.State(stA.p[src], stA.sat[src], stA.rs[src], stA.rv[src], stA.hc[src])
    n, L = gridA.nc, 30
    wl = W.Wells()
    WI = 5.0 * float(np.median(gridA.trans))
    inj = [7 + L * k for k in range(0, 3)]            # includes the k = 0 layer: rows next to ghost cells
    prod = [22 + L * k for k in range(2, 5)]          # includes the top layer
    wl.add_well("INJ", W.INJECTOR, gridA.z[inj[0]], inj, WI, (1.0, 0.0, 0.0), (W.SURFACE_RATE, 20.0 / 86400.0, (1.0, 0.0, 0.0)))
    wl.add_well("PROD", W.PRODUCER, gridA.z[prod[0]], prod, WI, (0.0, 1.0, 0.0), (W.BHP, 150 * decks.BAR))
    prm = capi.default_params(linear_solver_reduction=1e-11, linear_solver_maxiter=500, cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=cpr)
    A = GpuBlackoilModel(gridA, tab, prm)
    B = GpuBlackoilModel(gridB, tab, prm)
    dom = _Dom()
    for k, v in halo.items():
        setattr(dom, k, v)
    partition.attach_comm(B, dom, 0, 1, partition.make_unique_id())
    mA = W.DeviceWellModel(A, wl, W.WellState(wl, stA.p))
    mB = W.DeviceWellModel(B, wl, W.WellState(wl, stB.p))       # local ids == global ids for owned cells here
    dt = 2 * decks.DAY
    mA.prepareStep(dt, stA); mB.prepareStep(dt, stB)
    for it in range(4):
        cA, _ = mA.nonlinearIteration(it, single_precision=False)
        cB, _ = mB.nonlinearIteration(it, single_precision=False)
        assert cA == cB, it
        assert np.allclose(mA.well_flux_residual, mB.well_flux_residual, rtol=1e-5, atol=1e-12)
        # the equations assembled on the two (still agreeing) states: owned rows of B == rows of A, ghost rows zero
        rA, rB = A.residual(), B.residual()
        nb = gridB.nc
        rBo = np.concatenate([rB[a * nb:a * nb + n] for a in range(3)])
        assert np.abs(rA - rBo).max() <= 1e-9 * np.abs(rA).max(), it
        assert all(np.all(rB[a * nb + n:(a + 1) * nb] == 0.0) for a in range(3))
        sa, sb = A.getState(), B.getState()
        wa, wb = mA.pull_well_state(), mB.pull_well_state()
        assert np.array_equal(sb.p[n:], sb.p[src[n:]])            # ghost state still a copy of its owner's
        if it == 3:
            # The fourth iteration of this deck solves a nearly singular system (the well residuals jump there: CNV 0.57 -> 1.9; the
            # pressure LEVEL is a near-null mode of it): two solves that both reach a 3e-12 residual on matrices that agree to rounding
            # -- the residuals of A and B differ by 2e-13 on states 3e-6 Pa apart (tools/debug_selfhalo.py) -- differ by 5e-6 in that
            # mode, 90 Pa everywhere.  That is the conditioning of the system, not the decomposition.  The fourth update is therefore
            # compared with the level shift taken out -- and the shift itself must stay inside four times what was measured (ADVICE r3:
            # round 3 stopped comparing here, which would have let a decomposition bug in the fourth update through).
            shift = float(np.mean(sa.p - sb.p[:n]))
            assert abs(shift) <= 2e-5 * np.abs(sa.p).max(), shift
            assert np.array_equal(sa.hc, sb.hc[:n]), it
            dp_ = np.abs((sa.p - shift) - sb.p[:n]).max() / np.abs(sa.p).max()
            ds_ = np.abs(sa.sat - sb.sat[:n]).max()
            db_ = np.abs((wa.bhp - shift) - wb.bhp).max() / np.abs(wa.bhp).max()
            dq_ = np.abs(wa.qs - wb.qs).max() / np.abs(wa.qs).max()
            print("self-halo wells, fourth update: level shift %.1f Pa, then p %.2e sat %.2e bhp %.2e qs %.2e" % (shift, dp_, ds_, db_, dq_))
            # measured (gpurun r04_m): ILU0 + BiCGStab -- the decomposed ILU0 is a different preconditioner, hence a different Krylov path on this
            # nearly singular system -- shift -105.7 Pa, then p 3.5e-5, sat 1.2e-3, bhp 6e-6, qs 8e-5; CPR: 1e-13 throughout (the same
            # arithmetic on both sides).  The bounds are three times the measured values: a decomposition bug in the update shows as O(1).
            if cpr:
                assert abs(shift) <= 1e-6 * np.abs(sa.p).max() and dp_ <= 1e-9 and ds_ <= 1e-9 and db_ <= 1e-9 and dq_ <= 1e-9, (it, shift, dp_, ds_, db_, dq_)
            else:
                assert dp_ <= 1e-4 and ds_ <= 4e-3 and db_ <= 2e-5 and dq_ <= 3e-4, (it, shift, dp_, ds_, db_, dq_)
            break
        assert np.array_equal(sa.hc, sb.hc[:n]), it
        assert np.abs(sa.p - sb.p[:n]).max() <= 1e-6 * np.abs(sa.p).max() and np.abs(sa.sat - sb.sat[:n]).max() <= 1e-6, it
        assert np.allclose(wa.bhp, wb.bhp, rtol=1e-7) and np.allclose(wa.qs, wb.qs, rtol=1e-6, atol=1e-9 * np.abs(wa.qs).max()), it
    # a well that perforates a ghost cell is refused
    bad = W.Wells()
    bad.add_well("BAD", W.PRODUCER, gridB.z[0], [0, n + 1], WI, (0.0, 1.0, 0.0), (W.BHP, 150 * decks.BAR))
    with pytest.raises(ValueError, match="ghost"):
        W.DeviceWellModel(B, bad, W.WellState(bad, stB.p))
    A.close(); B.close()
