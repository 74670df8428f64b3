import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from opmgpu import capi, decks
from opmgpu.model import GpuBlackoilModel
red = float(sys.argv[1]); out = sys.argv[2]
grid = decks.cartesian_grid(30, 30, 20, lognormal_sigma=1.5, seed=3)
tab = decks.satfunc_standard_tables()
st = decks.initial_state(grid, tab, perturb=0.01, seed=3)
prm = capi.default_params(linear_solver_reduction=red, linear_solver_maxiter=500, cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1, ignore_convergence_failure=1)
m = GpuBlackoilModel(grid, tab, prm)
m.prepareStep(10 * decks.DAY, st)
res = []
for it in range(6):
    c, lin = m.nonlinearIteration(it, single_precision=False)
    s = m.getState()
    res.append(np.concatenate([s.p, s.sat.ravel()]))
    print("it", it, "conv", c, "lin", lin, "red %.2e" % m.linear_reduction, flush=True)
np.save(out, np.array(res))
