#!/usr/bin/env python3
"""HIP-event timing (no profiler) of the pieces of one CPR-BiCGStab iteration on the bench deck, after one real solve."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd")); sys.path.insert(0, ROOT)
import torch  # noqa: F401
from opmgpu import capi, decks
from opmgpu.model import GpuBlackoilModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
grid = decks.cartesian_grid(n, n, n, lognormal_sigma=0.5, seed=12345)
tab = decks.satfunc_standard_tables()
st = decks.initial_state(grid, tab, perturb=0.002, seed=12345)
m = GpuBlackoilModel(grid, tab, capi.default_params(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1))
m.prepareStep(5 * decks.DAY, st)
m.nonlinearIteration(0)
m.nonlinearIteration(1)
out = {}
for name, k in (("spmv", capi.K_SPMV), ("ilu_apply", capi.K_ILU_APPLY), ("ilu_factor", capi.K_ILU_FACTOR), ("cpr_apply", capi.K_CPR_APPLY),
                ("vcycle", capi.K_VCYCLE), ("cpr_setup", capi.K_CPR_SETUP), ("dot", capi.K_DOT), ("axpy", capi.K_AXPY), ("assemble", capi.K_ASSEMBLE),
                ("props", capi.K_PROPS)):
    out[name] = round(1e3 * m.time_kernel(k, reps=30), 1)
print(json.dumps(out), "(us per launch group)")
