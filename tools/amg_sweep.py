#!/usr/bin/env python3
"""A/B of the CPR pressure-stage knobs (env OPMGPU_AMG_*) on the bench deck: every configuration runs in its own process
(the knobs are read when the hierarchy is built) and reports ms per Newton iteration and linear iterations per Newton.
Run on the GPU box:  python tools/amg_sweep.py [n] > gpurun_out/amg_sweep.log"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(n, steps, warmup, sigma=0.5):
    sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd")); sys.path.insert(0, ROOT)
    import torch
    from opmgpu import capi, decks
    from opmgpu.model import GpuBlackoilModel
    grid = decks.cartesian_grid(n, n, n, lognormal_sigma=sigma, seed=12345)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, perturb=0.002, seed=12345)
    dt = 5 * decks.DAY
    m = GpuBlackoilModel(grid, tab, capi.default_params(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1))
    m.prepareStep(dt, st)
    it, lin_total, t_sol, t_asm, t0, nsteps = 0, 0, 0.0, 0.0, None, 0
    for step in range(warmup + steps):
        if step == warmup:
            torch.cuda.synchronize(); t0 = time.perf_counter(); lin_total = 0; t_sol = 0.0; t_asm = 0.0
        conv, lin = m.nonlinearIteration(it)
        lin_total += lin; t_sol += m.timings()[1]; t_asm += m.timings()[0]
        it += 1
        if conv or it > 10:
            m.prepareStep(dt); it = 0; nsteps += 1
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"ms_per_newton": 1e3 * el / steps, "solve_ms": t_sol / steps, "assemble_ms": t_asm / steps, "lin_per_newton": lin_total / steps, "time_steps": nsteps}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]))
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    sigma = sys.argv[2] if len(sys.argv) > 2 else "0.5"
    if len(sys.argv) > 3 and sys.argv[3] == "grid2":
        import itertools
        configs = [{}] + [{"OPMGPU_AMG_PDAMP": a, "OPMGPU_AMG_OMEGA": b, "OPMGPU_AMG_NPOST": c}
                          for a, b, c in itertools.product(("1.0", "1.3", "1.6", "1.9", "2.2"), ("0.8", "0.9", "1.0"), ("1", "2"))]
    elif len(sys.argv) > 3 and sys.argv[3] == "grid":
        import itertools
        configs = [{}] + [{"OPMGPU_AMG_PDAMP": a, "OPMGPU_AMG_OMEGA": b, "OPMGPU_AMG_NPOST": c, "OPMGPU_AMG_NPRE": d}
                          for a, b, c, d in itertools.product(("1.6", "1.9", "2.2", "2.5"), ("0.67", "0.8", "0.9", "1.0"), ("1", "2", "3"), ("1", "2"))]
    else:
        configs = None
    if configs is None:
        configs = [{}]
        for pd in ("1.3", "1.6", "1.9"):
            configs.append({"OPMGPU_AMG_PDAMP": pd})
        for om in ("0.8", "0.9"):
            configs.append({"OPMGPU_AMG_OMEGA": om})
        configs.append({"OPMGPU_AMG_NPOST": "2"})
        configs.append({"OPMGPU_AMG_NPRE": "2", "OPMGPU_AMG_NPOST": "2"})
        configs.append({"OPMGPU_AMG_PDAMP": "1.6", "OPMGPU_AMG_OMEGA": "0.8"})
        configs.append({"OPMGPU_AMG_PDAMP": "1.6", "OPMGPU_AMG_NPOST": "2"})
        configs.append({"OPMGPU_AMG_PDAMP": "1.6", "OPMGPU_AMG_NPOST": "2", "OPMGPU_AMG_OMEGA": "0.8"})
    for cfg in configs:
        env = dict(os.environ); env.update(cfg)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n), "20", "3", sigma], env=env, capture_output=True, text=True, timeout=600)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else ("FAILED: " + out.stderr[-400:])
        print(json.dumps(cfg), line, flush=True)
