#!/usr/bin/env python3
"""bench.py -- Mcell-updates/s per Newton step (assembly+solve) on MI355X, with the SpMV roofline and the CPU baseline timed beside it.

OUTPUT: stdout carries exactly ONE JSON line of < 4 KB (the driver keeps 8 KB of stdout): the contract keys, a short `config`, `roofline`,
`cpu_baseline`, `per_time_step` and compact name -> number maps of the same-run variants and of the other BASELINE decks.  Everything
else -- per-call durations, the full variant records, the kernel table, the prose notes -- goes to `bench_detail.json` beside this script
(and to stderr when OPMGPU_BENCH_VERBOSE=1).

Workload (BASELINE.json configs[2]): the synthetic 100x100x100 three-phase deck WITH its 5-spot (one rate-controlled water
injector + four BHP producers, full columns, device well model incl. control switching and the explicit well pre-solve).  A "step" is
one Newton iteration of the fully-implicit black-oil model: assemble (reservoir + wells) -> getConvergence -> solveJacobianSystem ->
updateState, state resident in HBM.  Time steps follow each other like in the simulator (NonlinearSolver with the reference's update
stabilisation; a step the reference's time stepper would cut -- not converged within max_iter, or a caught solver / numerical
condition -- restarts with 0.33 dt as in AdaptiveTimeStepping_impl.hpp:244-359).  The run starts like a simulation does, with a 1-day
step that grows by at most 3 per step towards --dt-days (timestep.initial_timestep_in_days, :110); timing begins once two steps have
passed and the step length has reached dt (deck set-up, untimed).

Which solver the headline runs, and in which arithmetic:
  * the reference's DEFAULT is solver_approach=interleaved: block-ILU0 + BiCGStab, in float when dt < 20 d and in double otherwise
    (BlackoilModelBase_impl.hpp:284, NewtonIterationBlackoilInterleaved.cpp:478-480) -> variants.ref_default_ilu0*;
  * the headline is solver_approach=cpr with cpr_use_amg=true and newton_use_gmres=true (NewtonIterationBlackoilCPR.hpp:59-63,
    .cpp:61-64), in DOUBLE like that plug-in (.cpp:117-140 never reads singlePrecision), stage 2 relaxed by cpr_relax = 1.0 -- with ONE
    difference, stated in config.reference_equivalence: the pressure stage is one AMG V-cycle per application (cpr_max_ell_iter = 0, a
    library extension) where the reference's external CPRPreconditioner wraps its AMG in an inner BiCGStab;
  * variants.cpr_ref_defaults is the CPR plug-in with ITS documented defaults (no AMG: ILU0-preconditioned inner BiCGStab on the
    pressure system, BiCGStab outside), variants.cpr_amg_inner_bicgstab the AMG behind the inner BiCGStab;
  * the float CPR solve is a combination the reference cannot run: kept as variants.cpr_f32_gmres, no like-for-like claim.
`value` is the cell count over the MEDIAN duration of the K timed Newton iterations that include a linear solve -- a
converged call skips solveJacobianSystem + updateState (BlackoilModelBase_impl.hpp:277-281) and is not a Newton iteration of the
reference's count; `value_mean_solving` is the cell count over their MEAN, `ms_per_step` the wall time of the K calls / K.

Per time step: a looser linear solve is cheaper per iteration and costs Newton iterations.  After the K timed calls
the same run goes on to --stat-calls calls in all; over the whole converged time steps inside them `per_time_step` reports the Newton
iterations per time step, the device ms per converged time step (chopped attempts included) and the ms per simulated day.

    python bench.py --gpus N --steps K --warmup W
(N > 1: one rank per GPU under torch.distributed.run; launched without WORLD_SIZE the script starts that launcher itself as a child process)
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "opm-simulators-legacy_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable


def spmv_bytes(nb, nnzb, S):
    """nnzb*(9*S + 4) + (nb+1)*4 + nb*3*S (x read once) + nb*3*S (y write)."""
    return nnzb * (9 * S + 4) + (nb + 1) * 4 + 2 * nb * 3 * S


def assembly_bytes(nb, nnzb, nconn, S):
    """state 41 + statics 24 + accum0 24 + connection data (nconn / nc) * 16 + residual 24 per cell, + the Jacobian write"""
    return nb * (41 + 24 + 24 + (nconn / nb) * 16 + 24) + nnzb * 9 * S


def algorithmic_bytes(nb, nnzb, nconn, S, n_scalar_nnz):
    """algorithmic HBM bytes per launch (group) of every kernel class the in-situ timers bracket.
    S = scalar size of the matrix / solver vectors (4: float solve, 8: double)."""
    return {
        "assembly(cell_props+flux)": assembly_bytes(nb, nnzb, nconn, S),
        # the two passes of the assembly by themselves: the value pass reads the state and writes the ten value planes
        # the neighbours need; the flux pass reads state + statics + values of the row and of its neighbours and writes the Jacobian
        "cell_props": nb * (49 + 16 + 24 + 80 + 24),
        "flux": nb * (49 + 16 + 24 + 24 + 8 + 24 + 12) + nnzb * (80 + 4 + 9 * S) + (nnzb - nb) * 16,
        "spmv_fused_dot1": spmv_bytes(nb, nnzb, S) + nb * 3 * S, "spmv_fused_dot2": spmv_bytes(nb, nnzb, S) + nb * 3 * S,
        "ilu0_apply": nnzb * (9 * S + 4) + nb * 3 * S * 3,
        "ilu0_factor": 2 * nnzb * 9 * S + nnzb * 4,
        "amg_vcycle": int(1.14 * 4 * n_scalar_nnz * (S + 4)),       # level 0: residual + 1 pre + 2 post sweeps over the scalar matrix; coarser levels add ~14 %
        "vector_updates": None, "cpr_other": None, "cpr_setup": None, "wells": None, "convergence": nb * 7 * 8, "update_state": nb * (3 * 8 + 2 * 49),
    }


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start `python -m torch.distributed.run --nproc-per-node N
    bench.py ...` as a CHILD process -- before this process has imported torch or touched the GPU --, relay its JSON line and exit with its
    code.  (Never an exec: a process that has initialised the GPU must not replace itself, and a child keeps this one free of HIP.)"""
    port = int(os.environ.get("MASTER_PORT", "0")) or (29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OPMGPU_BENCH_CHILD"] = "1"
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    for l in proc.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if proc.returncode != 0 or len(lines) != 1:
        print("bench.py: the %d-rank child run failed (exit code %d, %d JSON lines)" % (args.gpus, proc.returncode, len(lines)), file=sys.stderr)
        raise SystemExit(proc.returncode or 1)
    print(lines[0])
    raise SystemExit(0)


def parse_args(argv):
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
                    help="cpr: AMG pressure stage + ILU0 (reference solver_approach=cpr cpr_use_amg=true); ilu0: reference default solver_approach=interleaved")
    ap.add_argument("--precision", choices=["reference", "f32", "f64"], default="reference",
                    help="arithmetic of the linear solve and of the Jacobian.  reference: what the reference's plug-in of --solver computes in -- cpr: double "
                         "always (NewtonIterationBlackoilCPR.cpp:117-140); ilu0: float when dt < 20 d, else double (BlackoilModelBase_impl.hpp:284)")
    ap.add_argument("--krylov", choices=["auto", "bicgstab", "gmres", "fgmres"], default="auto",
                    help="gmres: the reference's newton_use_gmres option (restarted GMRES(40), left-preconditioned).  auto: gmres under CPR on the deck with "
                         "wells, bicgstab otherwise -- the other method runs as a same-run variant; fgmres: the flexible (right-preconditioned) form, not a reference solver")
    ap.add_argument("--verify", action="store_true",
                    help="GMRES with the true-residual check (opmgpu_params.gmres_verify_residual = 1; not a reference option).  Default: exactly dune's stopping rule, the "
                         "PRECONDITIONED residual -- what the reference's newton_use_gmres does; the check runs as the variant cpr_f64_gmres_verified")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = every GPU keeps an nx x ny x nz copy of the workload with its own 5-spot (global deck nx x ny*N x nz, see --weak-axis); strong = the fixed nx x ny x nz deck "
                         "is cut into N slabs along j, which keeps its vertical wells whole")
    ap.add_argument("--seed", type=int, default=7000, help="--deck random: the deck of tools/robust_sweep.py with this seed")
    ap.add_argument("--deck", choices=["cart", "spe10like", "nornelike", "random"], default="cart",
                    help="spe10like: 60 x 220 x 85 cells, sigma_lnK = 2.5 (BASELINE configs[3]); nornelike: Norne's 46 x 112 x 22 box with 60 %% of the cells inactive, NNCs, "
                         "threshold pressures and 36 wells (BASELINE configs[4]; N > 1: slabs of j-rows); both imply their own dimensions and wells")
    ap.add_argument("--wells", choices=["none", "fivespot"], default="fivespot",
                    help="fivespot (default, SURVEY 8d): 1 rate-controlled water injector + 4 BHP producers, full columns, device well model")
    ap.add_argument("--rate", type=float, default=1000.0, help="injection rate of the 5-spot, m3/day")
    ap.add_argument("--spin-up", type=int, default=2, help="time steps that pass before the measurement (deck set-up, untimed)")
    ap.add_argument("--cut-axis", type=int, default=1, choices=[0, 1], help="N > 1, --deck spe10like: cut into slabs along i (0) or j (1, default); both keep the vertical wells whole")
    ap.add_argument("--weak-axis", type=int, default=1, choices=[1, 2],
                    help="weak scaling: the N copies of the workload are put side by side along j (1, default: the cut crosses the weaker lateral coupling -- 10.2 against "
                         "11.3 GMRES columns per solve at N = 2 -- and no vertical well can straddle it) or stacked along k (2)")
    ap.add_argument("--stack", type=int, default=1, help="N = 1 only: the weak-scaling deck of N ranks (N copies of the workload stacked along k, one 5-spot per copy) on ONE GPU -- "
                    "the single-domain iteration counts the decomposed run is compared with (diagnostic)")
    ap.add_argument("--reduction", type=float, default=None, help="linear_solver_reduction (default: the reference's 1e-2); a sweep shows what dune's GMRES rule costs per time step")
    ap.add_argument("--ilu-fill", type=int, default=0, help="block ILU(n) with level-of-fill instead of the ILU0: cpr_ilu_n under --solver cpr, ilu_fillin_level otherwise (A/B; default 0)")
    ap.add_argument("--stage2-relax", type=float, default=None,
                    help="opmgpu_params.cpr_stage2_relax (library extension: damping of the stage-2 ILU0 alone; 1.0 = the reference's form).  Default: 1.0 -- except on the "
                         "DECOMPOSED SPE10-like deck, 0.9: with the undamped second stage the 4-rank run chops 9 of 12 steps (profiles/r04_z_dist_spe10_relax.log)")
    ap.add_argument("--stat-calls", type=int, default=60, help="calls (the K timed ones included) the per-time-step statistics are taken over")
    ap.add_argument("--only-main", action="store_true", help="skip the same-run variants, the other decks, the roofline micro-runs and the per-kernel pass (profiling)")
    ap.add_argument("--no-other-decks", action="store_true", help="skip the SPE9-like / SPE10-like / Norne-like legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="take the domain-decomposition code path (torch.distributed + RCCL communicator) even with one rank")
    ap.add_argument("--cpu-threads", type=int, default=1)
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"), help="where the full record goes (the stdout line stays < 4 KB)")
    ap.add_argument("--call-by-call", action="store_true", help="drive every Newton iteration through the seven single C calls instead of opmgpu_nonlinear_iteration "
                                                                 "(no per-call marks then: value falls back to the mean over all calls)")
    return ap.parse_args(argv)


MAX_LINE_BYTES = 4096          # the driver keeps 8 KB of stdout; a longer line reaches it without its head


def per_time_step_stats(log, ms, solved, lin):
    """Whole converged time steps inside a run of calls.  log[i] = (event, dt_seconds) of call i, event in {None, "step", "chop"}: "step"
    = this call found the time step converged (the step is complete), "chop" = the attempt was given up (exception or max_iter: the state is
    rolled back and the step restarts with 0.33 dt).  The window runs from the call after the first step boundary to the last completed
    step, so that only whole steps -- their chopped attempts included -- are counted."""
    n = min(len(log), len(ms))
    bounds = [i for i in range(n) if log[i][0] in ("step", "chop")]
    done = [i for i in range(n) if log[i][0] == "step"]
    if not bounds or not done or done[-1] <= bounds[0]:
        return None
    lo, hi = bounds[0] + 1, done[-1]              # calls lo..hi inclusive
    steps = [i for i in done if lo <= i <= hi]
    if not steps:
        return None
    tot = float(sum(ms[lo:hi + 1]))
    days = sum(log[i][1] for i in steps) / 86400.0
    nsolve = int(sum(1 for i in range(lo, hi + 1) if solved[i]))
    return {"time_steps": len(steps), "calls": hi - lo + 1, "chopped_attempts": sum(1 for i in range(lo, hi + 1) if log[i][0] == "chop"),
            "newton_iterations_per_time_step": nsolve / len(steps), "linear_iterations_per_time_step": float(sum(lin[lo:hi + 1])) / len(steps),
            "ms_per_converged_time_step": tot / len(steps), "simulated_days": days, "ms_per_simulated_day": tot / days if days > 0 else None}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, argv)                 # does not return

    # everything a library prints on the way (RCCL's version banner at communicator creation, gloo's connection notes, ...) goes to stderr:
    # file descriptor 1 carries exactly ONE line, the JSON line at the end
    sys.stdout.flush()
    saved_stdout_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    from opmgpu import capi, decks, baseline_decks, wells as W
    from opmgpu.model import GpuBlackoilModel, GpuNewtonIteration, NonlinearSolver

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d (or without a launcher)" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # rehearsal on ONE GPU (OPMGPU_COMM_TRANSPORT=shm): every rank on cuda:0, the test library's shared-memory transport instead of
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

    GpuBlackoilModel.fused_iteration = not args.call_by_call       # one library call per Newton iteration (the same sequence, fewer host round trips)
    if args.deck == "spe10like":
        args.nx, args.ny, args.nz = 60, 220, 85
    if args.deck == "nornelike":
        args.nx, args.ny, args.nz = 46, 112, 22
    irregular = None          # (grid, tables, state, wells) of the decks that bring their own wells
    if args.deck == "nornelike":
        irregular = baseline_decks.norne_like()
    if args.deck == "random":
        irregular = baseline_decks.random_irregular(args.seed)[:4]
        args.nx, args.ny, args.nz = irregular[0].dims
    ordering = capi.ORDER_MULTICOLOR if args.ordering == "multicolor" else capi.ORDER_NATURAL
    tab = decks.satfunc_standard_tables()
    dt_main = args.dt_days * decks.DAY

    def reference_single(solver, dt):
        """the arithmetic the reference's plug-in computes in: CPR is double throughout (NewtonIterationBlackoilCPR.cpp:117-140), the
        interleaved solver follows residual_.singlePrecision = dt < 20 d (BlackoilModelBase_impl.hpp:284)"""
        return False if solver.startswith("cpr") else dt < 20 * decks.DAY

    single_main = {"reference": reference_single(args.solver, dt_main), "f32": True, "f64": False}[args.precision]
    world_hint = int(os.environ.get("WORLD_SIZE", "1"))
    use_wells = args.wells == "fivespot"
    decomposed_spe10 = args.deck == "spe10like" and world_hint > 1
    if args.stage2_relax is None:
        args.stage2_relax = 0.9 if decomposed_spe10 else 1.0
    if args.krylov == "auto":
        # (decomposed SPE10-like: BiCGStab -- 13.0 iterations per solve, no chopped step on 4 ranks, where dune's GMRES rule needs 23.3 and chops)
        args.krylov = "bicgstab" if decomposed_spe10 else ("gmres" if (use_wells and args.solver == "cpr") else "bicgstab")
    verify = 1 if args.verify else 0

    # solver names: "ilu0" = solver_approach=interleaved; "cpr" = the headline's pressure stage (cpr_use_amg=true, ONE V-cycle per application:
    # cpr_max_ell_iter = 0); "cpr_ref" = the CPR plug-in's documented defaults (cpr_use_amg=false: ILU0-preconditioned inner BiCGStab on the
    # pressure system, cpr_solver_tol / cpr_max_ell_iter at their defaults); "cpr_amg_inner" = cpr_use_amg=true behind that inner BiCGStab
    CPR_KW = {"ilu0": dict(use_cpr=0), "cpr": dict(capi.CPR_AMG_VCYCLE), "cpr_ref": dict(use_cpr=1, cpr_use_amg=0), "cpr_amg_inner": dict(use_cpr=1, cpr_use_amg=1),
              # library extension (opmgpu_params.preconditioner_single): the headline's solver with its preconditioner in float, Krylov method in double
              "cpr_mixed": dict(capi.CPR_AMG_VCYCLE, preconditioner_single=1),
              # library extension (opmgpu_params.cpr_stage2_relax = 0.9): the damped second stage of rounds 1-3 -- the recommended, most robust setting
              "cpr_damped": dict(capi.CPR_AMG_VCYCLE, cpr_stage2_relax=0.9),
              # block ILU(1) with level-of-fill (csrc/fillilu.inl): the interleaved solver's ilu_fillin_level, the CPR plug-in's cpr_ilu_n
              "ilu1": dict(use_cpr=0, ilu_fillin_level=1), "cpr_ilu1": dict(capi.CPR_AMG_VCYCLE, cpr_ilu_n=1)}

    def make_params(solver=args.solver, krylov=args.krylov, verify=verify):
        kw = dict(CPR_KW[solver])
        if kw.get("use_cpr"):
            kw.setdefault("cpr_stage2_relax", args.stage2_relax)
        if args.ilu_fill:
            kw.setdefault("cpr_ilu_n" if kw.get("use_cpr") else "ilu_fillin_level", args.ilu_fill)
        if args.reduction:
            kw.setdefault("linear_solver_reduction", args.reduction)
        return capi.default_params(ilu_ordering=ordering, newton_use_gmres={"gmres": 1, "fgmres": 2}.get(krylov, 0),
                                   gmres_verify_residual=verify if krylov == "gmres" else 0, **kw)

    prm = make_params()
    spe10_spec = (200.0, 380.0)

    def make_deck():
        if args.deck == "spe10like":
            g, _, s, _ = baseline_decks.spe10_like()
            return g, s, spe10_spec
        if irregular is not None:
            return irregular[0], irregular[2], None
        g = decks.cartesian_grid(args.nx, args.ny, args.nz, lognormal_sigma=0.5, seed=12345)
        return g, decks.initial_state(g, tab, perturb=0.002, seed=12345), (args.rate, 150.0)

    # multi-GPU: every well lives on one rank.  Weak scaling: N copies of the one-GPU workload side by side along j (--weak-axis 1, default) or
    # stacked along k (2), one 5-spot per copy; strong scaling and the SPE10-like deck: slabs of whole j-rows, which keeps the deck's own vertical wells whole
    if use_dist:
        from opmgpu import partition
        if irregular is not None:
            g0, _, s0, wl0 = irregular
            model, grid, st, info = partition.build_distributed_model(args.nx, args.ny, args.nz, tab, prm, rank, world, local_rank, deck=(g0, s0), wells_fn=(lambda g: wl0) if use_wells else None, axis=1)
        elif args.deck == "spe10like":
            wells_fn = (lambda g: W.five_spot(g, rate_m3_per_day=spe10_spec[0], bhp_prod_bar=spe10_spec[1])) if use_wells else None
            model, grid, st, info = partition.build_distributed_model(60, 220, 85, tab, prm, rank, world, local_rank, deck="spe10like", wells_fn=wells_fn, axis=args.cut_axis)
        elif args.scaling == "strong":
            wells_fn = (lambda g: W.five_spot(g, rate_m3_per_day=args.rate, bhp_prod_bar=150.0)) if use_wells else None
            model, grid, st, info = partition.build_distributed_model(args.nx, args.ny, args.nz, tab, prm, rank, world, local_rank, wells_fn=wells_fn, axis=1)
        else:
            wells_fn = (lambda g: W.five_spot(g, rate_m3_per_day=args.rate, bhp_prod_bar=150.0, slabs=world, slab_axis=args.weak_axis)) if use_wells else None
            dims = (args.nx, args.ny * world, args.nz) if args.weak_axis == 1 else (args.nx, args.ny, args.nz * world)
            model, grid, st, info = partition.build_distributed_model(dims[0], dims[1], dims[2], tab, prm, rank, world, local_rank, wells_fn=wells_fn, axis=args.weak_axis)
        well_spec = None
        main_wells = info["wells"] if use_wells else None
    else:
        if args.stack > 1:
            if args.weak_axis == 1:
                args.ny *= args.stack
            else:
                args.nz *= args.stack
        grid, st, well_spec = make_deck()
        model = GpuBlackoilModel(grid, tab, prm, device=local_rank)
        info = {"n_owned": grid.nc, "n_global": grid.nc}
        if irregular is not None:
            main_wells = irregular[3] if use_wells else None
        else:
            main_wells = W.five_spot(grid, rate_m3_per_day=well_spec[0], bhp_prod_bar=well_spec[1], slabs=args.stack, slab_axis=args.weak_axis) if use_wells else None
    nc_global = info["n_global"]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    ns = NonlinearSolver()                  # reference defaults: max_iter 10, min_iter 1, dampening on detected oscillation

    def newton_iterations(model, n, state, dt, single):
        """n Newton iterations; time steps follow each other like in the simulator.  A Newton iteration that ends in one of the conditions the
        reference's time stepper catches (NumericalIssue, LinearSolverProblem, ISTLError: AdaptiveTimeStepping_impl.hpp:251-281), and a step
        that has not converged after the reference's max_iter (NonlinearSolver_impl.hpp:165 throws TooManyIterations, caught at :244), are
        handled the way it handles them: the state of the step's start is restored and the step restarts with 0.33 dt (:359); the step after a
        restart grows by at most 2 (:304), later ones by at most 3 (:300) back towards dt.  Such an iteration still counts as one of the n (its
        work was done).  single = "reference": the default solver's switch on the CURRENT step length (float below 20 d,
        BlackoilModelBase_impl.hpp:284), which a chopped step crosses.  state["log"] gets one (event, dt) entry per call."""
        from opmgpu.model import NumericalIssue, LinearSolverProblem, ISTLError
        it, lin, steps_done, failed = state["it"], 0, 0, 0
        log = state.setdefault("log", [])

        def chop():
            model.restoreState()
            state["dt"] *= 0.33
            state["chopped"] = state.get("chopped", 0) + 1
            state["restarted"] = True
            model.prepareStep(state["dt"])

        for _ in range(n):
            sp = (state["dt"] < 20 * decks.DAY) if single == "reference" else single
            state["calls_f32"] = state.get("calls_f32", 0) + int(bool(sp))
            dt_now = state["dt"]
            try:
                converged, l = model.nonlinearIteration(it, single_precision=sp, nonlinear_solver=ns)
            except (NumericalIssue, LinearSolverProblem, ISTLError):
                log.append(("chop", dt_now))
                chop()
                it = 0
                continue
            lin += l
            it += 1
            if converged and it > ns.min_iter:
                log.append(("step", dt_now))
                model.saveState()               # last_state of the time stepper
                state["dt"] = min(dt, (2.0 if state.pop("restarted", False) else 3.0) * state["dt"])
                model.prepareStep(state["dt"])  # next time step from the resident state
                it = 0
                steps_done += 1
            elif it > ns.max_iter:
                log.append(("chop", dt_now))
                failed += 1
                chop()
                it = 0
            else:
                log.append((None, dt_now))
        state["it"] = it
        return lin, steps_done, failed

    def timed_run(core, wells, st0, dt, single, nc, kernel_table=False, stat_calls=None):
        """deck set-up (spin-up time steps) -> W warm-up Newton iterations -> barrier -> exactly K timed Newton iterations -> barrier ->
        (statistics only) further calls up to stat_calls"""
        model = core if wells is None else W.DeviceWellModel(core, wells, W.WellState(wells, st0.p))
        stat_calls = max(args.steps, args.stat_calls if stat_calls is None else stat_calls)
        # set-up: the run starts the way the reference's time stepper starts one, with a first step of at most 1 d
        # (timestep.initial_timestep_in_days = 1, AdaptiveTimeStepping_impl.hpp:110) that grows by at most 3 per step towards dt; the measurement begins
        # once `spin_up` steps have passed AND the step length has reached dt
        dt0 = min(dt, decks.DAY)
        model.prepareStep(dt0, st0)
        model.saveState()
        state = {"it": 0, "dt": dt0, "chopped": 0}
        done = 0
        guard = 0
        while (done < args.spin_up or state["dt"] < dt * (1 - 1e-12) or state["it"] != 0) and guard < 40 * max(2, args.spin_up):
            _, d, _ = newton_iterations(model, 1, state, dt, single)
            done += d; guard += 1
        newton_iterations(model, args.warmup, state, dt, single)
        marks = GpuBlackoilModel.fused_iteration
        state["log"] = []
        barrier()
        if marks:
            core._chk(core.lib.opmgpu_iteration_marks(core.ctx, 1))
        t0 = time.perf_counter()
        lin_total = steps_done = failed = 0
        f32_before, chopped_before = state.get("calls_f32", 0), state["chopped"]
        for _ in range(args.steps):
            l, d, f = newton_iterations(model, 1, state, dt, single)
            lin_total += l; steps_done += d; failed += f
        barrier()
        elapsed = time.perf_counter() - t0
        out = {"lin": lin_total, "steps_done": steps_done, "steps_not_converged": failed, "chopped": state["chopped"], "chopped_timed": state["chopped"] - chopped_before,
               "dt_end": state["dt"], "calls_f32": state.get("calls_f32", 0) - f32_before, "nc": nc}
        solving_ms = solving_mean = None
        if marks:
            # beyond the timed region: the same run goes on, for the per-time-step statistics only
            newton_iterations(model, stat_calls - args.steps, state, dt, single)
            K, T = args.steps, stat_calls
            ms, sol, lit, ph = np.zeros(T), np.zeros(T, np.int32), np.zeros(T, np.int32), np.zeros((T, 3))
            n = C.c_int(0)
            core._chk(core.lib.opmgpu_iteration_marks_get(core.ctx, T, capi.dptr(ms), capi.iptr(sol), capi.iptr(lit), capi.dptr(ph), C.byref(n)))
            core._chk(core.lib.opmgpu_iteration_marks(core.ctx, 0))
            n = min(n.value, T)              # (a call that ended in an exception left its mark too)
            ms, sol, lit, ph = ms[:n], sol[:n].astype(bool), lit[:n], ph[:n]
            kk = min(K, n)
            out["calls"] = {"ms": [round(float(x), 4) for x in ms], "solved": [int(x) for x in sol], "linear_iterations": [int(x) for x in lit],
                            "event": [e for e, _ in state["log"][:n]], "timed": kk}
            if n == len(state["log"]):
                out["per_time_step"] = per_time_step_stats(state["log"], ms, sol, lit)
            if sol[:kk].any():
                s = sol[:kk]
                solving_ms, solving_mean = float(np.median(ms[:kk][s])), float(np.mean(ms[:kk][s]))
                out["n_solving"] = int(s.sum())
                out["lin_per_solving"] = float(lit[:kk][s].mean())
                out["breakdown"] = {"assemble": float(np.median(ph[:kk][s, 0])), "linear_solve": float(np.median(ph[:kk][s, 1])), "update": float(np.median(ph[:kk][s, 2])),
                                    "basis": "medians over the %d timed calls that include a solve; the phases are device time between events on the library's stream, "
                                             "the call's remainder is the host's decisions between them" % int(s.sum())}
                if (~s).any():
                    out["non_solving_ms_median"] = float(np.median(ms[:kk][~s]))
        if use_dist:
            tt = torch.tensor([elapsed, solving_ms if solving_ms is not None else -1.0, solving_mean if solving_mean is not None else -1.0], dtype=torch.float64,
                              device="cpu" if rehearsal else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt[0].item())
            if solving_ms is not None:
                solving_ms, solving_mean = float(tt[1].item()), float(tt[2].item())
        out["elapsed"], out["solving_ms"], out["solving_mean_ms"] = elapsed, solving_ms, solving_mean
        if kernel_table:
            # per-kernel pass: the NEXT K Newton iterations of the same run with the in-situ event brackets on (not part of `value`: the
            # brackets cost time); its own iteration counts are reported with it
            core._chk(core.lib.opmgpu_kernel_timing(core.ctx, 1))
            core._chk(core.lib.opmgpu_iteration_marks(core.ctx, 1))
            newton_iterations(model, args.steps, state, dt, single)
            tot, cnt = np.zeros(len(capi.KT_NAMES)), np.zeros(len(capi.KT_NAMES), np.int64)
            core._chk(core.lib.opmgpu_kernel_timing_get(core.ctx, capi.dptr(tot), cnt.ctypes.data_as(C.POINTER(C.c_int64))))
            core._chk(core.lib.opmgpu_kernel_timing(core.ctx, 0))
            K = args.steps
            sol, lit, n = np.zeros(K, np.int32), np.zeros(K, np.int32), C.c_int(0)
            core._chk(core.lib.opmgpu_iteration_marks_get(core.ctx, K, None, capi.iptr(sol), capi.iptr(lit), None, C.byref(n)))
            core._chk(core.lib.opmgpu_iteration_marks(core.ctx, 0))
            n = min(n.value, K)
            out["kt"] = (tot, cnt, n, int(sol[:n].sum()), int(lit[:n].sum()))
        return out

    def summary(r, dt, single, note=None):
        nc = r["nc"]
        ms_mean = 1e3 * r["elapsed"] / args.steps
        ms = r["solving_ms"] if r["solving_ms"] is not None else ms_mean
        s = {"value": nc / (ms * 1e-3) / 1e6, "ms_per_solving_iteration_median": r["solving_ms"], "ms_per_solving_iteration_mean": r["solving_mean_ms"],
             "value_mean_solving": nc / (r["solving_mean_ms"] * 1e-3) / 1e6 if r["solving_mean_ms"] else None,
             "value_all_calls_mean": nc / (ms_mean * 1e-3) / 1e6, "ms_per_step": ms_mean, "dt_days": dt / decks.DAY, "cells": nc,
             "arithmetic": ("f64 assembly; Jacobian and linear solve follow the reference's switch on the CURRENT step length (float below 20 d, which a chopped "
                            "step crosses): %d of the %d timed calls in float" % (r["calls_f32"], args.steps)) if single == "reference" else
                           "f64 assembly + %s Jacobian and linear solve" % ("f32" if single else "f64"),
             "solving_iterations": r.get("n_solving"), "linear_iterations_per_solving_iteration": r.get("lin_per_solving"),
             "time_steps_completed": r["steps_done"], "time_steps_not_converged": r["steps_not_converged"], "time_steps_chopped": r["chopped"],
             "time_steps_chopped_in_timed_region": r["chopped_timed"], "dt_days_at_end": r["dt_end"] / decks.DAY,
             "breakdown_ms_per_solving_iteration": r.get("breakdown"), "per_time_step": r.get("per_time_step"), "calls": r.get("calls")}
        if r["chopped_timed"]:
            s["caution"] = ("%d time step(s) of the timed region were cut (0.33 dt, the reference's rule): the timed iterations ran at a MIX of step lengths, "
                            "so this figure is not a dt = %g d figure" % (r["chopped_timed"], dt / decks.DAY))
        if note:
            s["note"] = note
        return s

    # ---- timed region: exactly K Newton iterations of the headline workload ----
    extras = rank == 0 and not use_dist and not args.only_main
    res = timed_run(model, main_wells, st, dt_main, single_main, nc_global, kernel_table=extras)
    main_sum = summary(res, dt_main, single_main)

    # same-run variants: the reference's default solver, the CPR plug-in's documented defaults, the other arithmetic / Krylov method, the
    # well-free deck, and SURVEY 8d's dt sweep (20 d is where the default solver's precision switches, BlackoilModelBase_impl.hpp:284)
    variants, other_decks = {}, {}
    if extras:
        def variant(name, solver, krylov, dt, single, wells_on=use_wells, note=None, verify=verify, env=None, into=variants, deck=None):
            saved = {k: os.environ.get(k) for k in (env or {})}
            os.environ.update(env or {})          # library knobs are read when the solver context is created
            g, t, s0, wl = deck if deck is not None else (grid, tab, st, main_wells)
            try:
                m = GpuBlackoilModel(g, t, make_params(solver, krylov, verify), device=local_rank)
            finally:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            try:
                into[name] = summary(timed_run(m, wl if wells_on else None, s0, dt, single, g.nc), dt, single, note)
            except Exception as e:          # e.g. a Krylov method running out of iterations: say so instead of dying
                into[name] = {"failed": repr(e)}
            m.close()

        variant("ref_default_ilu0", "ilu0", "bicgstab", dt_main, reference_single("ilu0", dt_main),
                note="solver_approach=interleaved, the reference's DEFAULT: block-ILU0 + BiCGStab, float because dt < 20 d")
        variant("ilu1_bicgstab", "ilu1", "bicgstab", dt_main, reference_single("ilu0", dt_main),
                note="solver_approach=interleaved with ilu_fillin_level = 1 (ISTLSolver.hpp:205): block ILU(1) + BiCGStab, float because dt < 20 d")
        if args.solver == "cpr":
            variant("cpr_ref_defaults", "cpr_ref", "bicgstab", dt_main, False,
                    note="solver_approach=cpr with the plug-in's documented defaults (NewtonIterationBlackoilCPR.hpp:59-63): cpr_use_amg=false -- the pressure system "
                         "by an ILU0-preconditioned inner BiCGStab (cpr_solver_tol 1e-2, at most 25 iterations: recollection of the external CPRPreconditioner) --, "
                         "cpr_relax 1.0, BiCGStab outside, double")
            variant("cpr_amg_inner_bicgstab", "cpr_amg_inner", args.krylov, dt_main, False,
                    note="the headline's solver with the AMG behind the inner BiCGStab of the reference's CPRPreconditioner (cpr_max_ell_iter 25) instead of one V-cycle")
            other = "bicgstab" if args.krylov != "bicgstab" else "gmres"
            variant("cpr_f64_%s" % other, "cpr", other, dt_main, False,
                    note="solver_approach=cpr cpr_use_amg=true (one V-cycle) in double with the reference's %s Krylov method" % ("default" if other == "bicgstab" else "newton_use_gmres"))
            if args.krylov == "gmres":
                variant("cpr_f64_gmres_verified", "cpr", "gmres", dt_main, False, verify=1 - verify,
                        note="the same solver %s the true-residual check (gmres_verify_residual): the solve is converged only when || b - A x || <= reduction || b || "
                             "too -- BiCGStab's statement; left-preconditioned GMRES by itself stops on || M^-1 (b - A x) ||" % ("WITH" if not verify else "WITHOUT"))
            variant("cpr_f64_%s_fixed_factor" % args.krylov, "cpr", args.krylov, dt_main, False, env={"OPMGPU_AMG_ADAPT": "0"},
                    note="the headline's solver with the pressure stage's coarse-grid corrections scaled by the fixed 1.9 instead of the "
                         "per-time-step choice between 1.9 and 2.3 (DESIGN.md section 4b; OPMGPU_AMG_ADAPT=0)")
            for kry in sorted({args.krylov, "bicgstab"}):
                variant("cpr_f64_%s_f32_precond" % kry, "cpr_mixed", kry, dt_main, False,
                        note="library extension, not a reference option: the double solve (Krylov method, operator, residual, solution in double) with its whole "
                             "preconditioner -- ILU0, pressure stage, stage-2 residual -- built and applied in float (opmgpu_params.preconditioner_single)")
            variant("cpr_f64_bicgstab_damped", "cpr_damped", "bicgstab", dt_main, False,
                    note="library extension, the RECOMMENDED configuration: CPR + BiCGStab in double with the second stage damped by 0.9 (cpr_stage2_relax) -- "
                         "the lowest ms per converged time step of the reference-arithmetic configurations and the robust one at long time steps and on the other decks")
            variant("cpr_f64_%s_ilu1" % args.krylov, "cpr_ilu1", args.krylov, dt_main, False,
                    note="the headline's solver with cpr_ilu_n = 1 (NewtonIterationBlackoilCPR.hpp:61): block ILU(1) with level-of-fill as the second stage, "
                         "multicolour elimination order of the filled pattern")
            variant("cpr_f32_%s" % args.krylov, "cpr", args.krylov, dt_main, True,
                    note="NOT a configuration the reference can run (its CPR plug-in is double-only, NewtonIterationBlackoilCPR.cpp:117-140): kept for continuity")
        if use_wells:
            variant("without_wells", args.solver, "bicgstab" if args.solver == "cpr" else args.krylov, dt_main, single_main, wells_on=False)
        for days in (1.0, 20.0, 30.0):
            if abs(args.dt_days - days) < 1e-9:
                continue
            dts = days * decks.DAY
            sp = reference_single("ilu0", dts)
            variant("dt%d_%s_ilu0" % (days, "f32" if sp else "f64"), "ilu0", "bicgstab", dts, sp if sp else "reference",
                    note="the reference's default solver at dt = %g d: %s (BlackoilModelBase_impl.hpp:284)" % (days, "float, dt < 20 d" if sp else "double, dt >= 20 d"))
            variant("dt%d_f64_cpr_%s" % (days, args.krylov), "cpr", args.krylov, dts, False)
            if days >= 20.0 and args.krylov != "bicgstab":
                variant("dt%d_f64_cpr_bicgstab" % days, "cpr", "bicgstab", dts, False)
        # the other BASELINE decks on one GPU (north_star: "synthetic Cartesian and SPE decks"): headline solver + reference default each
        if not args.no_other_decks and args.deck == "cart" and (args.nx, args.ny, args.nz) == (100, 100, 100):
            for name in ("spe9like", "spe10like", "nornelike"):
                dk = baseline_decks.make(name)
                dts = baseline_decks.DT_DAYS[name] * decks.DAY
                variant(name + "_cpr_%s" % args.krylov, "cpr", args.krylov, dts, False, deck=dk, into=other_decks)
                variant(name + "_ref_default_ilu0", "ilu0", "bicgstab", dts, reference_single("ilu0", dts), deck=dk, into=other_decks)
                del dk

    out = detail = None
    if rank == 0:
        nb = grid.nc
        rowptr, col, _ = model.jacobian()
        nnzb = int(col.size)
        roof = {}
        if not args.only_main:
            # ---- roofline of the dominant kernel: the 3x3-block SpMV (k_spmv<S,0>), HIP events on the launch stream.  `achieved` is
            # measured with the matrix rotating over 4 copies (HBM-resident operands, like any deck larger than the 256 MiB Infinity
            # Cache); the back-to-back replay of ONE copy (partly cache-resident at 100^3 in float) is reported beside it.
            for name, sp in (("f32", True), ("f64", False)):
                _, _, val = model.jacobian()
                s = GpuNewtonIteration(prm, device=local_rank)
                s.load(rowptr, col, val, sp)
                ms_cold = s.time_kernel(capi.K_SPMV_COLD, reps=40)
                ms_warm = s.time_kernel(capi.K_SPMV, reps=50)
                s.ilu0_factor()
                ms_ilu = s.time_kernel(capi.K_ILU_APPLY, reps=20)
                ms_copy = s.time_kernel(capi.K_STREAM_COPY, reps=20)
                sb = 4 if sp else 8
                nbytes = spmv_bytes(nb, nnzb, sb)
                roof[name] = {"bound": "hbm", "achieved": nbytes / (ms_cold * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": nbytes / (ms_cold * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "traffic": None,        # PMC counters need rocprofv3 (separate passes): profiles/r04_*_pmc_summary.json, never pasted here
                              "kernel": "k_spmv<%s,0>" % ("float" if sp else "double"), "ms_per_launch": ms_cold, "algorithmic_bytes": nbytes,
                              "operands": "matrix rotating over 4 copies (%.0f MB each): HBM-resident" % ((nnzb * (9 * sb + 4)) / 1e6),
                              "cache_resident_replay": {"ms_per_launch": ms_warm, "achieved": nbytes / (ms_warm * 1e-3) / 1e9, "frac": nbytes / (ms_warm * 1e-3) / 1e9 / HBM_PEAK_GBS},
                              "ilu0_apply_ms": ms_ilu, "stream_copy_GBs": 2 * nnzb * 9 * sb / (ms_copy * 1e-3) / 1e9}
                s.close()
                del val
        main_roof = roof.get("f32" if single_main else "f64")

        kernel_table = None
        if "kt" in res:
            tot, cnt, n_calls, n_solving, lin_sum = res["kt"]
            S = 4 if single_main else 8
            ab = algorithmic_bytes(nb, nnzb, grid.nconn, S, nnzb)
            nsol = max(1, n_solving)
            kernel_table = {"note": "HIP-event brackets on the launch stream around every launch (group) of a class during the %d Newton iterations right after the timed "
                                    "ones (the brackets cost time: a separate pass).  THIS pass: %d calls, %d of them with a solve, %.2f linear iterations per solving "
                                    "iteration.  ms_per_newton = class total / calls; algorithmic bytes per launch: DESIGN.md section 4; frac = achieved / %.0f GB/s"
                                    % (args.steps, n_calls, n_solving, lin_sum / nsol, HBM_PEAK_GBS),
                            "calls": n_calls, "solving_iterations": n_solving, "linear_iterations_per_solving_iteration": lin_sum / nsol, "classes": {}}
            for i, name in enumerate(capi.KT_NAMES):
                if cnt[i] == 0:
                    continue
                ms = tot[i] / cnt[i]
                e = {"ms_per_newton": tot[i] / max(1, n_calls), "launches_per_newton": cnt[i] / max(1, n_calls), "ms_per_launch": ms}
                if ab.get(name):
                    e.update({"algorithmic_bytes": int(ab[name]), "achieved_GBs": ab[name] / (ms * 1e-3) / 1e9, "frac": ab[name] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
                kernel_table["classes"][name] = e
            if cnt[0] and cnt[1]:
                ms = tot[0] / cnt[0] + tot[1] / cnt[1]
                b = ab["assembly(cell_props+flux)"]
                kernel_table["classes"]["assembly(cell_props+flux)"] = {"ms_per_launch": ms, "algorithmic_bytes": int(b), "achieved_GBs": b / (ms * 1e-3) / 1e9,
                                                                      "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                                      "note": "SURVEY 8d counts the assembly as ONE pass (state in, Jacobian out): both kernels' time against those bytes"}
            kernel_table["sum_ms_per_newton"] = float(tot.sum() / max(1, n_calls))

        cpu = cpu_all = None
        if not args.no_cpu_baseline and not use_dist:
            cpu_single = reference_single("ilu0", dt_main)          # the port runs the reference's DEFAULT solver in the reference's arithmetic for it
            cpu = cpu_baseline(grid, tab, st, main_wells, prm, dt_main, cpu_single, threads=args.cpu_threads)
            # the same port with its OpenMP loops (assembly, SpMV, vector updates, and since round 4 the ILU0 factorisation and sweeps by
            # levels -- bit-identical to the sequential ones; the reference's own sweeps are sequential) on this job's host cores -- the
            # reference itself caps OpenMP at 4 threads (FlowMain.hpp:269-271)
            try:
                ncores = len(os.sched_getaffinity(0))
            except AttributeError:
                ncores = os.cpu_count() or 1
            ncores = min(ncores, 16)             # a one-GPU box shares its host: 16 cores is this job's share
            if ncores > args.cpu_threads:
                cpu_all = cpu_baseline(grid, tab, st, main_wells, prm, dt_main, cpu_single, threads=ncores, budget_s=8.0, max_newton=2)

        if use_wells and irregular is not None:
            wells_txt = "%d vertical wells on the device (water injectors on rate control, producers on BHP or oil-rate control)" % irregular[3].nw + ("" if world == 1 else "; cut along j, every well on one rank")
        elif use_wells:
            spec = well_spec if well_spec else (spe10_spec if args.deck == "spe10like" else (args.rate, 150.0))
            wells_txt = ("5-spot on the device: 1 water injector (%.0f m3/d) + 4 BHP producers (%.0f bar), %d perforations each%s" %
                         (spec[0], spec[1], args.nz, "" if world == 1 else ("; one 5-spot per rank's copy of the workload" if (args.scaling == "weak" and args.deck == "cart")
                                                                            else "; cut along j, every well on one rank")))
        else:
            wells_txt = "none"
        weak = world > 1 and args.scaling == "weak" and args.deck == "cart"
        kry = {1: "gmres(40)", 2: "flexible gmres(40)"}.get(prm.newton_use_gmres, "bicgstab")
        if prm.use_cpr:
            stage1 = ("amg" if prm.cpr_use_amg else "ilu0(A_p)") + (" V-cycle" if (prm.cpr_use_amg and prm.cpr_max_ell_iter == 0) else
                                                                    " in %s(tol %g, <= %d)" % ("bicgstab" if prm.cpr_use_bicgstab else "cg", prm.cpr_solver_tol, prm.cpr_max_ell_iter))
            lin_name = "cpr(%s + ilu%d, relax %g%s) + %s" % (stage1, prm.cpr_ilu_n, prm.cpr_relax, "" if prm.cpr_stage2_relax == 1.0 else ", stage-2 damping %g" % prm.cpr_stage2_relax, kry)
        else:
            lin_name = "ilu%d(relax %g) + %s" % (prm.ilu_fillin_level, prm.ilu_relaxation, kry)
        if prm.use_cpr and not single_main and prm.newton_use_gmres != 2:
            equiv = "solver_approach=cpr cpr_use_amg=%s%s cpr_relax=%g in double (NewtonIterationBlackoilCPR.hpp:59-63, .cpp:117-140)" % (
                "true" if prm.cpr_use_amg else "false", " newton_use_gmres=true" if prm.newton_use_gmres else "", prm.cpr_relax)
            if prm.cpr_use_amg and prm.cpr_max_ell_iter == 0:
                equiv += ("; DIFFERS in the pressure stage: ONE V-cycle per application (cpr_max_ell_iter=0, library extension), the reference's "
                          "CPRPreconditioner wraps the AMG in an inner BiCGStab -> variants.cpr_amg_inner_bicgstab")
        elif not prm.use_cpr and single_main == reference_single("ilu0", dt_main) and not prm.newton_use_gmres:
            equiv = "reference default (solver_approach=interleaved)"
        else:
            equiv = "NOT a configuration the reference can run as is (see the docstring of bench.py)"
        pts = main_sum.get("per_time_step") or {}

        def r3(x):
            return None if x is None else round(float(x), 3)

        def compact_variants(vs):
            """name -> [Mcell-updates/s (median over solving calls), Newton iterations per time step, ms per converged time step, ms per simulated day]"""
            o = {}
            for k, v in vs.items():
                if "failed" in v:
                    o[k] = "failed"
                    continue
                p = v.get("per_time_step") or {}
                o[k] = [round(v["value"], 1), r3(p.get("newton_iterations_per_time_step")), r3(p.get("ms_per_converged_time_step")), r3(p.get("ms_per_simulated_day"))]
            return o

        out = {
            "metric": "Mcell-updates/sec per Newton step (assembly+solve)", "value": round(main_sum["value"], 3), "unit": "Mcell-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(main_sum["ms_per_step"], 4),
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
            "dtype": "f32" if single_main else "f64", "data": "synthetic",
            "value_basis": "cells / median ms of the timed calls with a linear solve (SURVEY 8d M1)" if res["solving_ms"] is not None else "cells / mean ms of the timed calls",
            "ms_per_solving_iteration_median": r3(main_sum["ms_per_solving_iteration_median"]),
            "ms_per_solving_iteration_mean": r3(main_sum["ms_per_solving_iteration_mean"]),
            "value_mean_solving": r3(main_sum["value_mean_solving"]), "value_all_calls_mean": r3(main_sum["value_all_calls_mean"]),
            "config": {"workload": "%s%dx%dx%d_3phase_blackoil%s" % ({"spe10like": "spe10like_", "nornelike": "nornelike_", "random": "random%d_" % args.seed}.get(args.deck, "cart"), args.nx, args.ny * (world if (weak and args.weak_axis == 1 and args.deck == "cart") else 1), args.nz * (world if (weak and args.weak_axis == 2) else 1),
                                                                     (("_%dwells" % irregular[3].nw) if irregular is not None else "_fivespot") if use_wells else ""),
                       "cells": nc_global, "cells_per_gpu": info["n_owned"], "dt_days": args.dt_days, "linear_solver": lin_name,
                       "arithmetic": main_sum["arithmetic"], "reference_equivalence": equiv,
                       "solving_iterations": res.get("n_solving"), "linear_its_per_solve": r3(res.get("lin_per_solving")),
                       "time_steps_completed": res["steps_done"], "time_steps_chopped": res["chopped_timed"], "time_steps_not_converged": res["steps_not_converged"],
                       "wells": wells_txt, "parallelism": "1 GPU" if world == 1 else "domain decomposition x%d, RCCL halo" % world},
            "per_time_step": {k: (r3(v) if isinstance(v, float) else v) for k, v in pts.items()} if pts else None,
            "breakdown_ms": {k: r3(v) for k, v in (res.get("breakdown") or {}).items() if k != "basis"} or None,
            "roofline": None if main_roof is None else {k: (round(v, 4) if isinstance(v, float) else v) for k, v in main_roof.items()
                                                        if k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "ms_per_launch", "algorithmic_bytes")},
            "cpu_baseline": None if cpu is None else {"value": round(cpu["value"], 4), "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"], "sample": cpu["sample_short"]},
            "cpu_all_cores": None if cpu_all is None else {"value": round(cpu_all["value"], 4), "cores": cpu_all["cores"]},
            "variants_columns": "[Mcell-updates/s, newton its / time step, ms / converged time step, ms / simulated day]",
            "variants": compact_variants(variants),
            "decks": compact_variants(other_decks),
            "detail": os.path.basename(args.detail),
        }
        # the line must reach the driver whole: shed the optional maps first, never the contract keys
        for k in ("variants_columns", "cpu_all_cores", "breakdown_ms", "decks", "variants", "per_time_step"):
            if len(json.dumps(out)) < MAX_LINE_BYTES:
                break
            out[k] = "see " + os.path.basename(args.detail)
        assert len(json.dumps(out)) < MAX_LINE_BYTES, "bench line too long for the driver"
        detail = {"line": out, "main": main_sum, "same_run_variants": variants, "other_decks": other_decks,
                  "roofline_f64_spmv": roof.get("f64"), "roofline_f32_spmv": roof.get("f32"), "kernel_table": kernel_table,
                  "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all, "nnzb_rank0": nnzb, "ilu0_ordering": args.ordering,
                  "spin_up_time_steps": args.spin_up, "nonlinear_solver": "reference NonlinearSolver (max_iter 10, update stabilisation on)",
                  "tables": "tests/satfuncStandard.DATA PROPS (reference's own test deck)",
                  "pressure_stage_correction_factor": ("fixed 1.9 (OPMGPU_AMG_ADAPT=0)" if os.environ.get("OPMGPU_AMG_ADAPT") == "0" else
                                                       "chosen per time step between 1.9 and 2.3 by the step's linear iterations per solve (library default)") if prm.use_cpr else None,
                  "gmres_true_residual_check": bool(prm.gmres_verify_residual), "argv": argv}
    model.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(saved_stdout_fd, 1)
    os.close(saved_stdout_fd)
    if rank == 0:
        try:
            with open(args.detail, "w") as f:
                json.dump(detail, f, indent=1)
        except OSError as e:
            print("bench.py: could not write %s: %r" % (args.detail, e), file=sys.stderr)
        if os.environ.get("OPMGPU_BENCH_VERBOSE"):
            print(json.dumps(detail), file=sys.stderr)
        print(json.dumps(out), flush=True)


def cpu_baseline(grid, tab, st, wl, prm, dt, single, threads=1, budget_s=15.0, max_newton=4):
    """The oracle (CPU restatement of the reference's algorithm: AD assembly into BSR, host standard-well model with the explicit
    Schur complement, natural-order block-ILU0, BiCGStab, same precision switch, same update) timed on the host cores for a bounded
    sample: the first Newton iterations of the SAME deck (wells included) from the SAME initial state (at least one, until the time
    budget is used)."""
    from oracle import oracle as orc
    from opmgpu import capi, wells as W
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import OracleBackend            # test infrastructure: the model interface on top of the oracle
    orc.set_threads(threads)
    ob = OracleBackend(orc, grid, tab, capi.default_params(ilu_ordering=capi.ORDER_NATURAL), wells=None if wl is None else wl.arrays())
    mo = None if wl is None else W.WellCoupledModel(ob, W.StandardWellsHost(wl, grid.z, tab.surface_density[0]), W.WellState(wl, st.p))
    (mo or ob).prepareStep(dt, st)
    its, lin, tot = 0, [], 0.0
    while its < max_newton and (its == 0 or tot < budget_s):
        t1 = time.perf_counter()
        if mo is not None:
            _, l = mo.nonlinearIteration(its, single_precision=single)
        else:
            ob.assemble(its == 0)
            ob.getConvergence()
            ob.solveJacobianSystem(single_precision=single); ob.updateState()
            l = ob.linear_iterations
        tot += time.perf_counter() - t1
        its += 1; lin.append(l)
    short = "first %d Newton iterations of the same deck%s from the same state: oracle assembly + natural-order ILU0/BiCGStab %s (reference default) + update, %.1f s" % (
        its, " + its wells (host well model)" if wl is not None else "", "f32" if single else "f64", tot)
    return {"value": its * grid.nc / tot / 1e6, "unit": "Mcell-updates/s", "cores": threads, "kind": "port", "sample_short": short,
            "sample": "first %d Newton iterations of the same deck%s and initial state: assembly + natural-order ILU0/BiCGStab %s (the reference's default "
                      "solver_approach=interleaved; linear its %s) + update, %.2f s"
                      % (its, " with its 5-spot (host well model, explicit Schur complement)" if wl is not None else "", "f32" if single else "f64", lin, tot),
            "newton_iterations": its, "linear_iterations": lin}


if __name__ == "__main__":
    main()
