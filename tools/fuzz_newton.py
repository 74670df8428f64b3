"""Randomised campaign for whole Newton iterations (assemble -> block-ILU0 / BiCGStab -> updateState) GPU vs the CPU oracle running freely
from the same start: random small decks (ACTNUM, NNC, threshold pressures, ENDSCALE, VAPPARS / ROCKTAB), both orderings, ILU0 or CPR on the
device, f64 solve with a tight reduction, three iterations: phase states identical, p within 1e-6 relative, s within 1e-6.   python tools/fuzz_newton.py [ncases] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from opmgpu import capi, decks
from opmgpu.model import GpuBlackoilModel
from oracle import oracle as orc

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 700
worst = {"p": 0.0, "sat": 0.0, "its_diff": 0.0}
compared = flips = 0
for case in range(ncases):
    rng = np.random.default_rng(seed0 + case)
    nx, ny, nz = (int(v) for v in rng.integers(3, 9, 3))
    kw = dict(lognormal_sigma=float(rng.uniform(0.0, 1.2)), seed=int(seed0 + case))
    if rng.random() < 0.3: kw["nnc_fraction"] = float(rng.uniform(0.02, 0.08))
    if rng.random() < 0.3: kw["actnum"] = rng.random(nx * ny * nz) > rng.uniform(0.1, 0.4)
    if rng.random() < 0.3: kw["thpres"] = float(rng.uniform(0.01, 0.1)) * decks.BAR
    grid = decks.cartesian_grid(nx, ny, nz, **kw)
    if grid.nc < 8: continue
    tkw = {}
    if rng.random() < 0.3: tkw["vappars"] = (float(rng.uniform(0.1, 2.0)), float(rng.uniform(0.1, 2.0)))
    if rng.random() < 0.3: tkw["rocktab"] = [(100.0, 0.97, 0.94), (200.0, 1.0, 1.0), (300.0, 1.02, 1.07), (500.0, 1.05, 1.1)]
    tab = decks.satfunc_standard_tables(**tkw)
    if rng.random() < 0.3: grid = decks.with_endpoints(grid, decks.random_endpoints(grid, seed=int(seed0 + case)))
    st = decks.initial_state(grid, tab, perturb=float(rng.uniform(0.001, 0.01)), seed=int(seed0 + case))
    ordering = int(rng.integers(0, 2))
    # the oracle side is always ILU0; the device side draws its preconditioner: ILU0, CPR with one V-cycle, the CPR plug-in's documented
    # defaults (ILU0-preconditioned inner BiCGStab / CG on the pressure system), the AMG behind the inner method, each optionally with the
    # float preconditioner inside the double solve and the damped second stage -- all preconditioner-only choices: BiCGStab outside, 1e-12
    stage = int(rng.integers(0, 5))
    pk = [dict(use_cpr=0), dict(capi.CPR_AMG_VCYCLE), dict(use_cpr=1), dict(use_cpr=1, cpr_use_bicgstab=0), dict(use_cpr=1, cpr_use_amg=1)][stage]
    if rng.random() < 0.4: pk["preconditioner_single"] = 1
    if stage and rng.random() < 0.4: pk["cpr_stage2_relax"] = 0.9
    # block ILU(n) with level-of-fill instead of the ILU0 (csrc/fillilu.inl): cpr_ilu_n under CPR, ilu_fillin_level otherwise
    if rng.random() < 0.3: pk["cpr_ilu_n" if stage else "ilu_fillin_level"] = int(rng.integers(1, 4))
    # (a float preconditioner is not one fixed linear operator: BiCGStab's recurrences stall around 1e-9 .. 1e-10 with it -- case 4080 of this
    # campaign: 163 iterations to 9.9e-10, then a breakdown, where the double preconditioner needs 12 iterations for 7.5e-13 -- so the mixed
    # cases ask for 1e-8 and are compared at 1e-5; the Newton solves the option is meant for ask for 1e-2)
    mixed = bool(pk.get("preconditioner_single"))
    tol_state = 1e-5 if mixed else 1e-6
    prm = capi.default_params(ilu_ordering=ordering, linear_solver_reduction=1e-8 if mixed else 1e-12, linear_solver_maxiter=1000, **pk)
    scale = np.asarray(prm.matbalscale[:])
    dt = float(rng.uniform(0.5, 10.0)) * decks.DAY
    nc = grid.nc
    rowptr, col = orc.pattern(grid)
    m = GpuBlackoilModel(grid, tab, prm)
    try:
        m.prepareStep(dt, st)
        pos, so, acc0 = None, st.copy(), None
        for it in range(3):
            m.assemble(it == 0); m.getConvergence()
            try:
                m.solveJacobianSystem(single_precision=False)
            except Exception as e:
                print("case %d it %d options %s ordering %d: %r" % (seed0 + case, it, pk, ordering, e), flush=True)
                if os.environ.get("OPMGPU_FUZZ_DIAG"):
                    from opmgpu.model import GpuNewtonIteration
                    print("   failed solve: %d iterations, reduction %.3e" % (m.linear_iterations, m.linear_reduction), flush=True)
                    rp_, cl_, vv_ = m.jacobian(); rr_ = m.residual()
                    bb_ = np.ascontiguousarray((rr_ * np.repeat(scale, nc)).reshape(3, nc).T).ravel()
                    print("   |b| range: %.3e .. %.3e (nonzero min %.3e)" % (np.abs(bb_).min(), np.abs(bb_).max(), np.abs(bb_[bb_ != 0]).min()), flush=True)
                    for name, extra in (("B1 double", {}), ("B1 mixed", dict(preconditioner_single=1))):
                        q = dict(pk); q.pop("preconditioner_single", None); q.update(extra)
                        s_ = GpuNewtonIteration(capi.default_params(ilu_ordering=ordering, linear_solver_reduction=1e-12, linear_solver_maxiter=1000, **q))
                        try:
                            s_.computeNewtonIncrement(rp_, cl_, vv_, bb_, False)
                            print("   %-12s ok: %d iterations, reduction %.2e" % (name, s_.iterations(), s_.reduction), flush=True)
                        except Exception as e2:
                            print("   %-12s %r (%d iterations, reduction %.2e)" % (name, e2, s_.iterations(), s_.reduction), flush=True)
                        s_.close()
                    # the same matrix through the other arithmetic paths: which of them fail?
                    for name, sp, extra in (("double", False, {}), ("float solve", True, dict(linear_solver_reduction=1e-4)), ("mixed", False, dict(preconditioner_single=1))):
                        q = dict(pk); q.pop("preconditioner_single", None); q.update(extra)
                        p2 = capi.default_params(ilu_ordering=ordering, **dict(dict(linear_solver_reduction=1e-12, linear_solver_maxiter=1000), **q))
                        m2 = GpuBlackoilModel(grid, tab, p2)
                        m2.prepareStep(dt, m.getState()); m2.assemble(True); m2.getConvergence()
                        try:
                            m2.solveJacobianSystem(single_precision=sp)
                            print("   %-12s ok: %d iterations, reduction %.2e" % (name, m2.linear_iterations, m2.linear_reduction), flush=True)
                        except Exception as e2:
                            print("   %-12s %r" % (name, e2), flush=True)
                        m2.close()
                raise
            m.updateState()
            if pos is None: pos = m.ordering()[0]
            r, val, acc0, _ = orc.assemble(grid, tab, dt, so, rowptr, col, scale=tuple(scale), accum0=acc0)
            b = np.ascontiguousarray((r * np.repeat(scale, nc)).reshape(3, nc).T).ravel()
            prm_o = capi.default_params(ilu_ordering=ordering, linear_solver_reduction=1e-12, linear_solver_maxiter=1000)
            sto, x, ito, _, _ = orc.bicgstab(rowptr, col, val, b, prm_o, position=pos, single=False)
            assert sto == 0, ("oracle solve", case, it)
            so = orc.update_state(grid, tab, prm, np.ascontiguousarray(x.reshape(nc, 3).T).ravel(), so)
            g = m.getState()
            if not np.array_equal(g.hc, so.hc):
                # a phase-state switch decided by a comparison within rounding distance of its threshold: count, and stop this case
                flips += 1; break
            ep, es = float(np.abs(g.p - so.p).max() / np.abs(so.p).max()), float(np.abs(g.sat - so.sat).max())
            if not mixed: worst["p"], worst["sat"] = max(worst["p"], ep), max(worst["sat"], es)
            else: worst["p_mixed"], worst["sat_mixed"] = max(worst.get("p_mixed", 0.0), ep), max(worst.get("sat_mixed", 0.0), es)
            if not prm.use_cpr and not mixed: worst["its_diff"] = max(worst["its_diff"], abs(m.linear_iterations - ito))
            assert ep < tol_state and es < tol_state, (case, it, pk, ep, es)
        else:
            compared += 1
    finally:
        m.close()
print("cases", ncases, "compared", compared, "threshold flips", flips, "worst", {k: "%.1e" % v for k, v in worst.items()}, flush=True)
