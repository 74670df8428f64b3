"""Real ranks (shm test transport) with wells: linear iterations per Newton iteration for OPMGPU_COARSE / OPMGPU_COARSE_BLOCKS settings."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd"))
import numpy as np
import test_gpu_dist_shm as T
cfg = dict(nx=40, ny=40, nz=48, sigma=0.5, seed=21, perturb=0.002, dt_days=5.0, newton=6, rate=2000.0 / 86400.0,
           params=dict(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1), wells=True, single=True)
for world in (1, 2, 4):
    for coarse, blocks in ((0, 1), (1, 1), (1, 4)):
        if world == 1 and (coarse, blocks) != (1, 1):
            continue
        os.environ["OPMGPU_COARSE"] = str(coarse); os.environ["OPMGPU_COARSE_BLOCKS"] = str(blocks)
        with tempfile.TemporaryDirectory() as tmp:
            p, sat, hc, hist = T._launch(cfg, world, tmp)
        print("world", world, "coarse", coarse, "blocks", blocks, "lin", hist[:, 1].tolist(), "sum", int(hist[:, 1].sum()), flush=True)
