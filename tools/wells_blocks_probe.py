"""5-spot deck on ONE rank with a communicator (no neighbours): does a multi-block coarse space (OPMGPU_COARSE_BLOCKS) pay with wells,
where the single global constant does not?  Prints time per Newton iteration and linear iterations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from opmgpu import capi, decks, partition, wells as W
from opmgpu.model import GpuBlackoilModel
n = 100
grid = decks.cartesian_grid(n, n, n, lognormal_sigma=0.5, seed=12345)
tab = decks.satfunc_standard_tables()
st = decks.initial_state(grid, tab, perturb=0.002, seed=12345)
prm = capi.default_params(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1)
m = GpuBlackoilModel(grid, tab, prm)
if os.environ.get("PROBE_COMM", "1") == "1":
    part = np.zeros(grid.nc, dtype=np.int64)
    dom = partition.LocalDomain(grid, part, 0)
    partition.attach_comm(m, dom, 0, 1, partition.make_unique_id())
wl = W.five_spot(grid, rate_m3_per_day=5000.0, bhp_prod_bar=150.0)
d = W.DeviceWellModel(m, wl, W.WellState(wl, st.p))
d.prepareStep(5 * decks.DAY, st)
it = tot = lin_tot = 0; t_acc = 0.0
for k in range(24):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    conv, lin = d.nonlinearIteration(it, single_precision=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if k >= 4: t_acc += dt; tot += 1; lin_tot += lin
    it += 1
    if (conv and it >= 1) or it > 10:
        d.prepareStep(5 * decks.DAY); it = 0
print("ms/newton %.3f lin/newton %.2f" % (1e3 * t_acc / tot, lin_tot / tot), flush=True)
