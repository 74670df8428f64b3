#!/usr/bin/env python3
"""Tiny driver for PMC passes over the AMG set-up kernels (Galerkin sums): a few Newton iterations of the bench deck with the CPR solver.
   rocprofv3 --pmc <counters> --kernel-trace -d out -- python3 tools/pmc_amg_setup.py [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd")); sys.path.insert(0, ROOT)
import torch  # noqa: F401
from opmgpu import capi, decks
from opmgpu.model import GpuBlackoilModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
grid = decks.cartesian_grid(n, n, n, lognormal_sigma=0.5, seed=12345)
tab = decks.satfunc_standard_tables()
st = decks.initial_state(grid, tab, perturb=0.002, seed=12345)
m = GpuBlackoilModel(grid, tab, capi.default_params(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1, newton_use_gmres=1))
m.prepareStep(5 * decks.DAY, st)
for it in range(4):
    try:
        m.nonlinearIteration(it)
    except Exception as e:
        print(e)
m.close()
