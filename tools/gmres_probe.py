"""BiCGStab vs restarted GMRES (the reference's newton_use_gmres) under CPR on the bench deck: time per Newton iteration and iterations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd")); sys.path.insert(0, ROOT)
import torch
from opmgpu import capi, decks
from opmgpu.model import GpuBlackoilModel
n = 100
grid = decks.cartesian_grid(n, n, n, lognormal_sigma=0.5, seed=12345)
tab = decks.satfunc_standard_tables()
st = decks.initial_state(grid, tab, perturb=0.002, seed=12345)
for gm in (0, 1):
    prm = capi.default_params(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1, newton_use_gmres=gm)
    m = GpuBlackoilModel(grid, tab, prm)
    m.setState(st)
    tot_it, tot_lin, t_acc, steps = 0, 0, 0.0, 0
    for step in range(4):
        m.prepareStep(5 * decks.DAY)
        for it in range(15):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            conv, lin = m.nonlinearIteration(it, single_precision=True)
            m.getState if False else None
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if step > 0: t_acc += dt; tot_it += 1; tot_lin += lin
            if conv and it >= 1: break
    print("gmres" if gm else "bicgstab", "ms/newton %.3f" % (1e3 * t_acc / tot_it), "lin/newton %.2f" % (tot_lin / tot_it), flush=True)
    m.close()
