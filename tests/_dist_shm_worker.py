"""One rank of a multi-process run over the shared-memory TEST transport (tests/support/shm_transport.cpp, OPMGPU_COMM_TRANSPORT=shm): builds the global
synthetic deck, keeps its slab, joins the communicator, runs Newton iterations and writes its OWNED cells' state.  Started by
tests/test_gpu_dist_shm.py; every rank uses cuda:0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd"))

import numpy as np  # noqa: E402

from opmgpu import capi, decks, partition  # noqa: E402
from opmgpu import wells as W  # noqa: E402
from opmgpu.model import GpuBlackoilModel  # noqa: E402


def deck(cfg):
    kw = {}
    if cfg.get("unstructured"):          # inactive cells + non-neighbour connections: index-range partition, ranks with several neighbours
        kw["actnum"] = np.random.default_rng(cfg["seed"]).random(cfg["nx"] * cfg["ny"] * cfg["nz"]) > 0.3
        kw["nnc_fraction"] = 0.05
    grid = decks.cartesian_grid(cfg["nx"], cfg["ny"], cfg["nz"], lognormal_sigma=cfg["sigma"], seed=cfg["seed"], **kw)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, perturb=cfg["perturb"], seed=cfg["seed"])
    return grid, tab, st


def wells_of(grid, cfg):
    """One rate-controlled injector and one BHP producer, each a vertical well inside ONE z-slab (a well lives on one rank)."""
    nx, ny, nz = cfg["nx"], cfg["ny"], cfg["nz"]
    wl = W.Wells()
    WI = 5.0 * float(np.median(grid.trans))
    col = lambda i, j, ks: [i + nx * (j + ny * k) for k in ks]      # noqa: E731
    inj = col(1, 1, range(0, 2))
    prod = col(nx - 2, ny - 2, range(nz - 2, nz))
    wl.add_well("INJ", W.INJECTOR, grid.z[inj[0]], inj, WI, (1.0, 0.0, 0.0), (W.SURFACE_RATE, cfg["rate"], (1.0, 0.0, 0.0)))
    wl.add_well("PROD", W.PRODUCER, grid.z[prod[0]], prod, WI, (0.0, 1.0, 0.0), (W.BHP, 150 * decks.BAR))
    return wl


def run(cfg, rank, world, uid, out):
    grid, tab, st = deck(cfg)
    prm = capi.default_params(**cfg["params"])
    if world == 1:
        model, lst = GpuBlackoilModel(grid, tab, prm), st
        owned_global = np.arange(grid.nc)
    else:
        part = partition.slab_partition(grid, world, axis=cfg.get("axis", 2))
        dom = partition.LocalDomain(grid, part, rank)
        model = GpuBlackoilModel(dom.grid, tab, prm)
        partition.attach_comm(model, dom, rank, world, uid)
        if cfg.get("subslabs"):
            # caller-supplied coarse blocks of the pressure stage (opmgpu_comm_set_coarse_blocks): sub-slabs of this rank's slab along the cut
            # direction j, which keep the vertical wells of the deck inside one block each
            m = int(cfg["subslabs"])
            gid = dom.global_of_local[:dom.n_owned]
            coord = (gid // cfg["nx"]) % cfg["ny"]
            lo, hi = int(coord.min()), int(coord.max()) + 1
            blk = capi.i32(np.minimum(m - 1, (coord - lo) * m // max(1, hi - lo)))
            assert model.lib.opmgpu_comm_set_coarse_blocks(model.ctx, m, capi.iptr(blk)) == capi.OK
            bad = blk.copy(); bad[0] = m
            assert model.lib.opmgpu_comm_set_coarse_blocks(model.ctx, m, capi.iptr(bad)) == capi.EINVAL      # out of range: refused, the valid map stays
        lst = dom.local_state(st)
        owned_global = dom.global_of_local[:dom.n_owned]
    driver = model
    if cfg["wells"]:
        wl = wells_of(grid, cfg)
        if world > 1:
            wl = dom.local_wells(wl, part)
        driver = W.DeviceWellModel(model, wl, W.WellState(wl, lst.p))
    hist = []
    if cfg.get("ats"):
        # one report step through the adaptive sub-stepping loop (restart on failure): every decision in it -- Newton convergence,
        # the relative change that feeds the PID controller -- is a collective result, so all ranks chop and grow the same way
        from opmgpu import timestepping as ts
        from opmgpu.model import NonlinearSolver
        model.setState(lst)
        if os.environ.get("OPMGPU_TEST_TRACE"):         # diagnostic: every Newton iteration's outcome on stderr
            inner = model.nonlinearIteration

            def traced(it, **kw):
                try:
                    r = inner(it, **kw)
                except Exception as e:
                    print("[trace] it %d raised %r" % (it, e), file=sys.stderr, flush=True)
                    raise
                print("[trace] it %d -> %s" % (it, r), file=sys.stderr, flush=True)
                return r
            model.nonlinearIteration = traced
        ats = ts.AdaptiveTimeStepping(initial_timestep_days=cfg["ats"]["first_days"])
        rep = ats.step(0.0, cfg["ats"]["report_days"] * decks.DAY, NonlinearSolver(max_iter=cfg["ats"]["max_iter"]), model)
        assert rep["converged"]
        hist = [[int(round(dt)), 1] for dt in rep["substeps"]] + [[int(round(f[0])), 0] for f in rep["failed"]]
    else:
        driver.prepareStep(cfg["dt_days"] * decks.DAY, lst)
        for it in range(cfg["newton"]):
            conv, lin = driver.nonlinearIteration(it, single_precision=cfg["single"])
            hist.append([bool(conv), int(lin)])
    s = model.getState()
    n = owned_global.size
    # computeFluidInPlace on the decomposed state: three regions by global cell index (+ region 0 = "in no region" for every 7th cell),
    # the fipnum of this rank's cells incl. its ghosts (which must not be counted twice)
    gids = np.arange(grid.nc) if world == 1 else dom.global_of_local
    fipnum = (1 + (gids * 3) // grid.nc).astype(np.int32)
    fipnum[gids % 7 == 3] = 0
    fip = model.computeFluidInPlace(fipnum, nregions=3)
    np.savez(out, ids=owned_global, p=s.p[:n], sat=s.sat[:n], hc=s.hc[:n], hist=np.array(hist, dtype=np.int64), fip=fip)
    model.close()


if __name__ == "__main__":
    cfg = json.loads(sys.argv[1])
    rank, world = int(sys.argv[2]), int(sys.argv[3])
    uid = bytes.fromhex(sys.argv[4]) if sys.argv[4] != "-" else None
    run(cfg, rank, world, uid, sys.argv[5])
