#!/usr/bin/env python3
"""Diagnostic: the self-halo deck with device wells, plain periodic grid (A) against the one-rank decomposition (B), per Newton iteration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from opmgpu import capi, decks, partition, wells as W
from opmgpu.model import GpuBlackoilModel
from test_gpu_dist import _periodic_pair, _Dom
cpr = int(sys.argv[1]) if len(sys.argv) > 1 else 0
gridA, gridB, src, halo = _periodic_pair(nx=6, ny=5, nz=5)
tab = decks.satfunc_standard_tables()
stA = decks.initial_state(gridA, tab, perturb=0.005)
stB = decks.State(stA.p[src], stA.sat[src], stA.rs[src], stA.rv[src], stA.hc[src])
n, L = gridA.nc, 30
wl = W.Wells()
WI = 5.0 * float(np.median(gridA.trans))
inj = [7 + L * k for k in range(0, 3)]; prod = [22 + L * k for k in range(2, 5)]
wl.add_well("INJ", W.INJECTOR, gridA.z[inj[0]], inj, WI, (1.0, 0.0, 0.0), (W.SURFACE_RATE, 20.0 / 86400.0, (1.0, 0.0, 0.0)))
wl.add_well("PROD", W.PRODUCER, gridA.z[prod[0]], prod, WI, (0.0, 1.0, 0.0), (W.BHP, 150 * decks.BAR))
prm = capi.default_params(linear_solver_reduction=1e-11, linear_solver_maxiter=500, cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=cpr)
A = GpuBlackoilModel(gridA, tab, prm); B = GpuBlackoilModel(gridB, tab, prm)
dom = _Dom()
for k, v in halo.items():
    setattr(dom, k, v)
partition.attach_comm(B, dom, 0, 1, partition.make_unique_id())
mA = W.DeviceWellModel(A, wl, W.WellState(wl, stA.p)); mB = W.DeviceWellModel(B, wl, W.WellState(wl, stB.p))
dt = 2 * decks.DAY
mA.prepareStep(dt, stA); mB.prepareStep(dt, stB)
for it in range(5):
    for tag, m, core in (("A", mA, A), ("B", mB, B)):
        core.setSolvePrecision(False); core.assemble(it == 0)
        r = core.residual()
        conv = core.getConvergence(); wc = m.wellConvergence()
        print(it, tag, "conv", conv, wc, "CNV", core.CNV, "MB", core.MB, "|R|", np.abs(r[:3 * n] if tag == "A" else r).max(), "wellres", m.well_flux_residual, m.well_ctrl_residual, flush=True)
        if tag == "A":
            rA = r.copy()
        else:
            nb = gridB.nc
            rB = np.concatenate([r[a * nb:a * nb + n] for a in range(3)])
            print("   residual diff A-B", np.abs(rA - rB).max(), "ghost rows R", np.abs(np.concatenate([r[a * nb + n:(a + 1) * nb] for a in range(3)])).max())
        core.solveJacobianSystem(single_precision=False)
        print("   lin its", core.linear_iterations, core.linear_reduction)
        core.updateState()
    sa, sb = A.getState(), B.getState()
    print(it, "state diff p", np.abs(sa.p - sb.p[:n]).max(), "sat", np.abs(sa.sat - sb.sat[:n]).max(), "ghost copy ok", np.array_equal(sb.p[n:], sb.p[src[n:]]), flush=True)
