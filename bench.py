#!/usr/bin/env python3
"""bench.py -- Mcell-updates/s per Newton step (assembly + solve) on MI355X, with the SpMV roofline
and the CPU baseline timed beside it.

A "step" is one Newton iteration of the fully-implicit black-oil model on the synthetic
100x100x100 three-phase deck (BASELINE.json configs[2], the configuration the metric is quoted
on): assemble -> getConvergence -> solveJacobianSystem (block-ILU0 + BiCGStab, float because
dt < 20 d exactly as the reference switches) -> updateState, state resident in HBM.  Time steps
follow each other like in the simulator: when a step converges the next one starts from the
updated state with a fresh accum0.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "opm-simulators-legacy_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable


def spmv_bytes(nb, nnzb, scalar_bytes):
    """nnzb*(9*S + 4) + (nb+1)*4 + nb*3*S (x read once) + nb*3*S (y write)."""
    return nnzb * (9 * scalar_bytes + 4) + (nb + 1) * 4 + 2 * nb * 3 * scalar_bytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nx", type=int, default=100)
    ap.add_argument("--ny", type=int, default=100)
    ap.add_argument("--nz", type=int, default=100)
    ap.add_argument("--dt-days", type=float, default=5.0)
    ap.add_argument("--ordering", choices=["multicolor", "natural"], default="multicolor")
    ap.add_argument("--solver", choices=["cpr", "ilu0"], default="cpr",
                    help="cpr: AMG pressure stage + ILU0 (reference solver_approach=cpr); ilu0: reference default solver_approach=interleaved")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = every GPU keeps an nx x ny x nz slab (global deck nx x ny x nz*N, sized for 288 GB/GPU); strong = the fixed nx x ny x nz deck is cut into N slabs")
    ap.add_argument("--wells", choices=["none", "fivespot"], default="none",
                    help="fivespot: SURVEY 8d synthetic wells (1 rate-controlled water injector + 4 BHP producers, full columns) with the device well model; single GPU only")
    ap.add_argument("--only-main", action="store_true", help="skip the same-run ILU0 / 5-spot variants and the roofline micro-runs (profiling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="take the domain-decomposition code path (torch.distributed + RCCL communicator) even with one rank")
    ap.add_argument("--cpu-threads", type=int, default=1)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from opmgpu import capi, decks
    from opmgpu.model import GpuBlackoilModel, GpuNewtonIteration

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # rehearsal on ONE GPU (OPMGPU_COMM_TRANSPORT=shm): every rank on cuda:0, the library's shared-memory test transport instead of
    # RCCL and a gloo group for the barrier -- runs the N > 1 code path of this script where no multi-GPU node is available
    rehearsal = os.environ.get("OPMGPU_COMM_TRANSPORT") == "shm"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    ordering = capi.ORDER_MULTICOLOR if args.ordering == "multicolor" else capi.ORDER_NATURAL
    prm = capi.default_params(ilu_ordering=ordering, use_cpr=int(args.solver == "cpr"))
    tab = decks.satfunc_standard_tables()
    dt = args.dt_days * decks.DAY
    single = dt < 20 * decks.DAY            # BlackoilModelBase_impl.hpp:284

    if use_dist:
        from opmgpu import partition
        nz_global = args.nz * world if args.scaling == "weak" else args.nz
        model, grid, st, info = partition.build_distributed_model(args.nx, args.ny, nz_global, tab, prm, rank, world, local_rank)
    else:
        grid = decks.cartesian_grid(args.nx, args.ny, args.nz, lognormal_sigma=0.5, seed=12345)
        st = decks.initial_state(grid, tab, perturb=0.002, seed=12345)
        model = GpuBlackoilModel(grid, tab, prm, device=local_rank)
        info = {"n_owned": grid.nc, "n_global": grid.nc}
    nc_global = info["n_global"]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def with_wells(model, which):
        if which == "none" or use_dist:
            return model
        from opmgpu import wells as W
        wl = W.five_spot(grid, rate_m3_per_day=5000.0, bhp_prod_bar=150.0)
        return W.DeviceWellModel(model, wl, W.WellState(wl, st.p))

    sample_phases = os.environ.get("OPMGPU_BENCH_NO_PHASES") is None      # (A/B: what reading the phase events every iteration costs)

    def timed_run(model, wells=None):
        """exactly K timed Newton iterations after W warm-up ones; time steps follow each other like in the simulator"""
        core = model
        model = with_wells(model, wells if wells is not None else args.wells)
        model.prepareStep(dt, st)
        it, lin_total, steps_done = 0, 0, 0
        t_asm = t_sol = t_upd = 0.0
        t0 = None
        for step in range(args.warmup + args.steps):
            if step == args.warmup:
                barrier()
                t0 = time.perf_counter()
                lin_total = 0
                t_asm = t_sol = t_upd = 0.0
            converged, lin = model.nonlinearIteration(it)
            if sample_phases:
                a, s, u = core.timings()
                t_asm += a; t_sol += s; t_upd += u
            lin_total += lin
            it += 1
            if (converged and it >= 1) or it > 10:
                model.prepareStep(dt)           # next time step from the resident state
                it = 0
                steps_done += 1
        barrier()
        elapsed = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return {"elapsed": elapsed, "lin": lin_total / args.steps, "steps_done": steps_done,
                "breakdown": {"assemble": t_asm / args.steps, "linear_solve": t_sol / args.steps, "update": t_upd / args.steps}}

    # ---- timed region: exactly K Newton iterations ----
    res = timed_run(model)
    elapsed, steps_done = res["elapsed"], res["steps_done"]
    ms_per_step = 1e3 * elapsed / args.steps
    value = nc_global / (elapsed / args.steps) / 1e6
    # the same K iterations with the reference's DEFAULT linear solver (solver_approach=interleaved: ILU0 + BiCGStab),
    # which is also what the CPU baseline runs
    ilu0 = None
    if not use_dist and prm.use_cpr and not args.only_main:
        prm0 = capi.default_params(ilu_ordering=ordering, use_cpr=0)
        m0 = GpuBlackoilModel(grid, tab, prm0, device=local_rank)
        r0 = timed_run(m0)
        m0.close()
        ilu0 = {"value": nc_global / (r0["elapsed"] / args.steps) / 1e6, "ms_per_step": 1e3 * r0["elapsed"] / args.steps,
                "linear_iterations_per_newton": r0["lin"], "breakdown_ms_per_step": r0["breakdown"]}

    # ... and with the SURVEY 8d 5-spot (device well model), unless that already is the main run
    fivespot = None
    if not use_dist and args.wells == "none" and not args.only_main:
        m1 = GpuBlackoilModel(grid, tab, prm, device=local_rank)
        r1 = timed_run(m1, wells="fivespot")
        m1.close()
        fivespot = {"value": nc_global / (r1["elapsed"] / args.steps) / 1e6, "ms_per_step": 1e3 * r1["elapsed"] / args.steps,
                    "linear_iterations_per_newton": r1["lin"], "breakdown_ms_per_step": r1["breakdown"],
                    "wells": "1 rate-controlled water injector + 4 BHP producers, %d perforations each, device well model" % args.nz}

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel: the 3x3-block SpMV, HIP events on the launch stream ----
        nb = grid.nc
        rowptr, col, _ = model.jacobian()
        nnzb = int(col.size)
        roof = {}
        for name, sp in (("f32", True), ("f64", False)):
            # micro-run on the same matrix in a solver-only context (its own stream), values = assembled Jacobian
            _, _, val = model.jacobian()
            s = GpuNewtonIteration(prm, device=local_rank)
            s.load(rowptr, col, val, sp)
            ms = s.time_kernel(capi.K_SPMV, reps=50)
            s.ilu0_factor()
            ms_ilu = s.time_kernel(capi.K_ILU_APPLY, reps=20)
            ms_copy = s.time_kernel(capi.K_STREAM_COPY, reps=20)
            sb = 4 if sp else 8
            nbytes = spmv_bytes(nb, nnzb, sb)
            roof[name] = {"bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "k_spmv<%s,0>" % ("float" if sp else "double"),
                          "ms_per_launch": ms, "algorithmic_bytes": nbytes,
                          "ilu0_apply_ms": ms_ilu, "stream_copy_GBs": 2 * nnzb * 9 * sb / (ms_copy * 1e-3) / 1e9}
            s.close()
            del val
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_spmv.json")
        if os.path.exists(pmc):
            try:
                d = json.load(open(pmc))
                for name in roof:
                    if name in d:      # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE), profiles/r01_pmc_spmv.json
                        roof[name]["traffic"] = d[name]["hbm_bytes_per_launch"]
                        roof[name]["traffic_over_algorithmic"] = d[name]["traffic_over_algorithmic"]
            except Exception:
                pass
        main_roof = roof["f32" if single else "f64"]

        cpu = cpu_all = None
        if not args.no_cpu_baseline and not use_dist:
            cpu = cpu_baseline(args, grid, tab, st, prm, dt, single, threads=args.cpu_threads)
            # the same port with its OpenMP-able loops (assembly, SpMV, vector updates; the ILU sweeps stay sequential like the
            # reference's) on all host cores of this box -- the reference itself caps OpenMP at 4 threads (FlowMain.hpp:269-271)
            try:
                ncores = len(os.sched_getaffinity(0))
            except AttributeError:
                ncores = os.cpu_count() or 1
            ncores = min(ncores, 16)             # a one-GPU box shares its host: 16 cores is this job's share (more threads only oversubscribe)
            if ncores > args.cpu_threads:
                cpu_all = cpu_baseline(args, grid, tab, st, prm, dt, single, threads=ncores, budget_s=8.0, max_newton=2)

        out = {
            "metric": "Mcell-updates/sec per Newton step (assembly+solve)", "value": value, "unit": "Mcell-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
            "dtype": "f64 assembly + %s linear solve" % ("f32" if single else "f64"), "data": "synthetic",
            "config": {"workload": "cart%dx%dx%d_3phase_blackoil" % (args.nx, args.ny, args.nz * (world if (world > 1 and args.scaling == "weak") else 1)),
                       "cells": nc_global, "cells_per_gpu": info["n_owned"], "nnzb_rank0": nnzb,
                       "dt_days": args.dt_days, "linear_solver": ("cpr(amg V-cycle + ilu0)" if prm.use_cpr else "ilu0") + " + bicgstab", "ilu0_ordering": args.ordering, "linear_iterations_per_newton": res["lin"],
                       "time_steps_completed": steps_done, "tables": "tests/satfuncStandard.DATA PROPS (reference's own test deck)",
                       "wells": "none" if (args.wells == "none" or use_dist) else "5-spot, 5 wells x %d perforations, device well model (rank-7 operator per well)" % args.nz,
                       "parallelism": "1 GPU" if world == 1 else "domain decomposition x%d, RCCL halo" % world},
            "breakdown_ms_per_step": res["breakdown"],
            "same_run_with_reference_default_solver_ilu0": ilu0,
            "same_run_with_fivespot_wells": fivespot,
            "roofline": main_roof, "roofline_f64_spmv": roof["f64"], "roofline_f32_spmv": roof["f32"],
            "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all,
        }
    model.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def cpu_baseline(args, grid, tab, st, prm, dt, single, threads=1, budget_s=15.0, max_newton=4):
    """The oracle (CPU restatement of the reference's algorithm: AD assembly into BSR, natural-order
    block-ILU0, BiCGStab, same precision switch, same update) timed on the host cores for a bounded
    sample: the first Newton iterations of the SAME deck from the SAME initial state (at least one, until
    the time budget is used; later iterations have the harder linear systems, like on the GPU)."""
    import numpy as np
    from oracle import oracle as orc
    from opmgpu import capi
    orc.set_threads(threads)
    nc = grid.nc
    scale = np.asarray(prm.matbalscale[:])
    prm_nat = capi.default_params(ilu_ordering=capi.ORDER_NATURAL)
    rowptr, col = orc.pattern(grid)
    cur, acc0 = st.copy(), None
    t_asm = t_sol = t_upd = 0.0
    its, lin = 0, []
    while its < max_newton and (its == 0 or t_asm + t_sol + t_upd < budget_s):
        t1 = time.perf_counter()
        r, val, acc0, binv = orc.assemble(grid, tab, dt, cur, rowptr, col, scale=tuple(scale), accum0=acc0)
        orc.convergence(grid, prm, dt, r, binv)
        t2 = time.perf_counter()
        b = np.ascontiguousarray((r * np.repeat(scale, nc)).reshape(3, nc).T).ravel()
        sto, x, it, red, _ = orc.bicgstab(rowptr, col, val, b, prm_nat, position=None, single=single)
        t3 = time.perf_counter()
        dx = np.ascontiguousarray(x.reshape(nc, 3).T).ravel()
        cur = orc.update_state(grid, tab, prm, dx, cur)
        t4 = time.perf_counter()
        t_asm += t2 - t1; t_sol += t3 - t2; t_upd += t4 - t3
        its += 1; lin.append(it)
        if sto != 0:
            break
    tot = t_asm + t_sol + t_upd
    return {"value": its * nc / tot / 1e6, "unit": "Mcell-updates/s", "cores": threads, "kind": "port",
            "sample": "first %d Newton iterations of the same deck and initial state: assembly %.2fs + natural-order ILU0/BiCGStab %s (linear its %s) %.2fs + update %.2fs"
                      % (its, t_asm, "f32" if single else "f64", lin, t_sol, t_upd),
            "newton_iterations": its, "linear_iterations": lin}


if __name__ == "__main__":
    main()
