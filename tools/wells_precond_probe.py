#!/usr/bin/env python3
"""How many BiCGStab iterations does the rank-7 well operator cost because neither preconditioner stage sees it?
Same deck and Newton path (f64, tight tolerance so the paths coincide) with (a) the device well model (factored operator, ILU0 and
AMG built from A alone) and (b) the host well model (explicit Schur-complement cliques inside A: both stages see the wells).
    python tools/wells_precond_probe.py [--n 60] [--rate 1000] [--reduction 1e-2]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "opm-simulators-legacy_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=60)
    ap.add_argument("--rate", type=float, default=1000.0)
    ap.add_argument("--reduction", type=float, default=1e-2)
    ap.add_argument("--newton", type=int, default=8)
    ap.add_argument("--dt-days", type=float, default=5.0)
    ap.add_argument("--gmres", type=int, default=0, help="newton_use_gmres")
    args = ap.parse_args()
    from opmgpu import capi, decks, wells as W
    from opmgpu.model import GpuBlackoilModel
    n = args.n
    grid = decks.cartesian_grid(n, n, n, lognormal_sigma=0.5, seed=12345)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, perturb=0.002, seed=12345)
    wl = W.five_spot(grid, rate_m3_per_day=args.rate * (n / 100.0) ** 2, bhp_prod_bar=150.0)
    dt = args.dt_days * decks.DAY
    for cpr in ((1,) if args.gmres else (1, 0)):
        prm = capi.default_params(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=cpr, linear_solver_reduction=args.reduction, linear_solver_maxiter=400, newton_use_gmres=args.gmres)
        out = {}
        for kind in ("device", "host", "none"):
            if kind == "host":
                gm = GpuBlackoilModel(grid, tab, prm, wells=wl.arrays())
                m = W.WellCoupledModel(gm, W.StandardWellsHost(wl, grid.z, tab.surface_density[0]), W.WellState(wl, st.p))
            elif kind == "device":
                gm = GpuBlackoilModel(grid, tab, prm)
                m = W.DeviceWellModel(gm, wl, W.WellState(wl, st.p))
            else:
                gm = GpuBlackoilModel(grid, tab, prm)
                m = gm
            m.prepareStep(dt, st)
            its = []
            for it in range(args.newton):
                conv, lin = m.nonlinearIteration(it, single_precision=False)
                its.append(lin)
                if conv and it >= 1:
                    break
            out[kind] = its
            gm.close()
        print("cpr=%d reduction %g: linear iterations per Newton iteration" % (cpr, args.reduction))
        for k, v in out.items():
            print("   %-7s %s  mean %.2f" % (k, v, np.mean([x for x in v if x > 0])), flush=True)


if __name__ == "__main__":
    main()
