#!/usr/bin/env python3
"""Tiny driver for PMC passes over the assembly kernels and the SpMV: a few assemblies of the bench deck, 8 cold f32 SpMVs, nothing else.
   rocprofv3 --pmc <counters> --kernel-trace -d out -- python3 tools/pmc_assembly.py [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd")); sys.path.insert(0, ROOT)
import torch  # noqa: F401
from opmgpu import capi, decks
from opmgpu.model import GpuBlackoilModel, GpuNewtonIteration
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
grid = decks.cartesian_grid(n, n, n, lognormal_sigma=0.5, seed=12345)
tab = decks.satfunc_standard_tables()
st = decks.initial_state(grid, tab, perturb=0.002, seed=12345)
m = GpuBlackoilModel(grid, tab, capi.default_params())
m.prepareStep(5 * decks.DAY, st)
m.setSolvePrecision(True)               # float Jacobian (k_assemble_rows<float, ...>)
for i in range(4):
    m.assemble(i == 0)
m.getConvergence()
m.setSolvePrecision(False)              # double Jacobian (k_assemble_rows<double, ...>): what bench.py's headline assembles
for i in range(4):
    m.assemble(False)
# the roofline kernel of bench.py: the SpMV over rotating copies of the matrix (out of cache), a few launches in both precisions
rowptr, col, val = m.jacobian()
for single in (True, False):
    s = GpuNewtonIteration(capi.default_params()); s.load(rowptr, col, val, single); s.ilu0_factor()
    s.time_kernel(capi.K_SPMV_COLD, 8)
    s.close()
m.close()
