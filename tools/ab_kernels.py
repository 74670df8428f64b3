#!/usr/bin/env python3
"""A/B timing of the main kernels at 100^3 (run on the GPU box); env OPMGPU_CLUSTER selects row clustering."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd")); sys.path.insert(0, ROOT)
import numpy as np
from opmgpu import capi, decks
from opmgpu.model import GpuBlackoilModel, GpuNewtonIteration
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
grid = decks.cartesian_grid(n, n, n, lognormal_sigma=0.5, seed=12345)
tab = decks.satfunc_standard_tables()
st = decks.initial_state(grid, tab, perturb=0.002, seed=12345)
prm = capi.default_params()
m = GpuBlackoilModel(grid, tab, prm)
m.prepareStep(5 * decks.DAY, st)
m.assemble(True)
t_asm = m.time_kernel(capi.K_ASSEMBLE, 10); t_props = m.time_kernel(capi.K_PROPS, 10)
m.setSolvePrecision(True)          # float Jacobian (what a dt < 20 d step assembles)
m.assemble(False)
t_asm_f = m.time_kernel(capi.K_ASSEMBLE, 10)
m.setSolvePrecision(False)
m.assemble(False)
rowptr, col, val = m.jacobian()
out = {"knobs": {k: v for k, v in os.environ.items() if k.startswith("OPMGPU_")}, "assemble_f64_ms": round(t_asm, 4), "assemble_f32_ms": round(t_asm_f, 4),
       "props_f64_ms": round(t_props, 4), "flux_f64_ms": round(t_asm - t_props, 4)}
for name, sp in (("f32", True), ("f64", False)):
    s = GpuNewtonIteration(prm); s.load(rowptr, col, val, sp); s.ilu0_factor()
    out["spmv_" + name] = round(s.time_kernel(capi.K_SPMV, 50), 4)
    out["ilu_" + name] = round(s.time_kernel(capi.K_ILU_APPLY, 30), 4)
    out["factor_" + name] = round(s.time_kernel(capi.K_ILU_FACTOR, 5), 4)
    s.close()
print(out, flush=True)
