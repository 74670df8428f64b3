import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd")); sys.path.insert(0, ROOT)
import torch
from opmgpu import capi, decks
from opmgpu.model import GpuBlackoilModel
n = 100
for sigma in (0.5, 2.0):
    grid = decks.cartesian_grid(n, n, n, lognormal_sigma=sigma, seed=12345)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, perturb=0.002, seed=12345)
    for k in (1, 2, 3):
        prm = capi.default_params(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1, linear_solver_maxiter=k, linear_solver_reduction=1e-30, ignore_convergence_failure=1)
        m = GpuBlackoilModel(grid, tab, prm)
        m.prepareStep(5 * decks.DAY, st)
        reds = []
        for it in range(3):
            m.setSolvePrecision(True)
            m.assemble(it == 0); m.getConvergence()
            m.solveJacobianSystem(single_precision=True)
            reds.append(m.linear_reduction if hasattr(m, "linear_reduction") else None)
            m.updateState()
        print("sigma", sigma, "iters", k, "reductions", reds, flush=True)
        m.close()
