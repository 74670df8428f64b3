#!/usr/bin/env python3
"""Per-iteration trace of one time step, device and/or oracle, on the full-size decks of tests/test_gpu_fullsize.py.
    python tools/newton_trace.py --deck spe9like --cpr 0 --side both [--double] [--reduction 1e-2] [--maxiter 150] [--dt-days D] [--rate-scale s]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "opm-simulators-legacy_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--deck", default="spe9like")
    ap.add_argument("--cpr", type=int, default=0)
    ap.add_argument("--side", default="both")
    ap.add_argument("--double", action="store_true")
    ap.add_argument("--reduction", type=float, default=1e-2)
    ap.add_argument("--maxiter", type=int, default=150)
    ap.add_argument("--dt-days", type=float, default=None)
    ap.add_argument("--max-newton", type=int, default=15)
    ap.add_argument("--no-wells", action="store_true")
    ap.add_argument("--rate", type=float, default=None)
    ap.add_argument("--perturb", type=float, default=None)
    ap.add_argument("--bhp", type=float, default=None)
    ap.add_argument("--gascap", type=float, default=None)
    ap.add_argument("--plain", action="store_true", help="plain Newton instead of the reference's stabilised NonlinearSolver")
    args = ap.parse_args()
    from opmgpu import capi, decks, wells as W
    from opmgpu.model import GpuBlackoilModel
    import test_gpu_fullsize as T
    make, dtd = T.DECKS[args.deck]
    kw = {}
    if args.rate is not None:
        kw["rate"] = args.rate
    if args.perturb is not None:
        kw["perturb"] = args.perturb
    if args.gascap is not None:
        kw["gascap"] = args.gascap
    if args.bhp is not None:
        kw["bhp"] = args.bhp
    grid, tab, st, wl = make(**kw)
    if args.no_wells:
        wl = None
    dt = (args.dt_days or dtd) * decks.DAY
    single = (dt < 20 * decks.DAY) and not args.double
    print("deck %s: %d cells, %d conns, wells %s, dt %.2f d, single %s" % (args.deck, grid.nc, grid.nconn, None if wl is None else (wl.nw, wl.nperf), dt / decks.DAY, single), flush=True)

    from opmgpu.model import NonlinearSolver
    ns = None if args.plain else NonlinearSolver()

    def run(model, core, label):
        model.prepareStep(dt, st)
        it = 0
        while True:
            t = time.time()
            try:
                if hasattr(model, "nonlinearIteration"):
                    conv, lin = model.nonlinearIteration(it, single_precision=single, nonlinear_solver=ns)
                else:
                    core.assemble(it == 0); conv = core.getConvergence(); lin = 0
                    if not conv or it < 1:
                        core.solveJacobianSystem(single_precision=single); core.updateState(); lin = core.linear_iterations
            except Exception as e:
                print("%s it %d: EXCEPTION %r" % (label, it, e), flush=True)
                return
            wf = getattr(model, "well_flux_residual", None) if wl is not None and label == "gpu" else (getattr(getattr(model, "wh", None), "well_flux_residual", None))
            wc = getattr(model, "well_ctrl_residual", None) if wl is not None and label == "gpu" else (getattr(getattr(model, "wh", None), "well_ctrl_residual", None))
            print("%s it %2d conv %d lin %4d CNV %s MB %s wf %s wc %s  %.2fs" % (label, it, conv, lin, np.array2string(core.CNV, precision=3), np.array2string(core.MB, precision=3),
                                                                                None if wf is None else np.array2string(np.asarray(wf), precision=3), wc, time.time() - t), flush=True)
            it += 1
            if (conv and it > 1) or it > args.max_newton:
                break
        print("%s: %d Newton iterations" % (label, it), flush=True)

    if args.side in ("gpu", "both"):
        gm = GpuBlackoilModel(grid, tab, capi.default_params(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=args.cpr, linear_solver_reduction=args.reduction, linear_solver_maxiter=args.maxiter))
        md = gm if wl is None else W.DeviceWellModel(gm, wl, W.WellState(wl, st.p))
        run(md, gm, "gpu")
        gm.close()
    if args.side in ("oracle", "both"):
        from oracle import oracle as orc
        from util import OracleBackend
        orc.set_threads(16)
        ob = OracleBackend(orc, grid, tab, capi.default_params(linear_solver_reduction=args.reduction, linear_solver_maxiter=args.maxiter), wells=None if wl is None else wl.arrays())
        mo = ob if wl is None else W.WellCoupledModel(ob, W.StandardWellsHost(wl, grid.z, tab.surface_density[0]), W.WellState(wl, st.p))
        run(mo, ob, "oracle")


if __name__ == "__main__":
    main()
