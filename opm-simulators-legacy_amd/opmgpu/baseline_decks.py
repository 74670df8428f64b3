"""The synthetic stand-ins of BASELINE.json's configurations (the original decks are not available offline, so every one is
"-like", authored from the public problem statements), in ONE place: the full-size parity tests and bench.py run the same recipes.

    make(name) -> (grid, tables, initial_state, wells), DT_DAYS[name]

cart100   configs[2]: 100 x 100 x 100, lognormal permeability (sigma_lnK 0.5), 5-spot            1 000 000 cells
spe10like configs[3]: 60 x 220 x 85, 20 x 10 x 2 ft cells, sigma_lnK 2.5, 5-spot                 1 122 000 cells
nornelike configs[4]: 46 x 112 x 22 box, 60 % inactive, 5 % NNCs, threshold pressures, 36 wells    ~45 000 cells
spe9like  configs[1]: 24 x 25 x 15, 300 ft cells, 1 injector + 25 producers                          9 000 cells
cart60    the bench deck's recipe at 60^3 (the size the CPU checker's GMRES still affords)
"""
import numpy as np

from . import decks
from . import wells as W


def cart100(rate=1000.0, perturb=0.002, n=100):
    grid = decks.cartesian_grid(n, n, n, lognormal_sigma=0.5, seed=12345)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, perturb=perturb, seed=12345)
    return grid, tab, st, W.five_spot(grid, rate_m3_per_day=rate, bhp_prod_bar=150.0)


def spe10_like(rate=200.0, perturb=1e-4, bhp=380.0, gascap=0.0):
    """SPE10 Model 2 dimensions and cell sizes (20 x 10 x 2 ft), channel-free lognormal permeability with sigma_lnK = 2.5 (the SPE10
    permeability file is not available offline), 5-spot like the original: central water injector, four corner producers.  Like
    SPE10 there is no gas cap (undersaturated oil everywhere); pore volumes are ~2 m3 per cell, so rates and drawdown are moderate."""
    grid = decks.cartesian_grid(60, 220, 85, dx=6.096, dy=3.048, dz=0.6096, tops=3657.6, lognormal_sigma=2.5, seed=10)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, p_ref=413.0 * decks.BAR, z_ref=3657.6, perturb=perturb, seed=10, gas_cap_fraction=gascap,
                             gas_only_fraction=0.01 if gascap > 0 else 0.0)
    return grid, tab, st, W.five_spot(grid, rate_m3_per_day=rate, bhp_prod_bar=bhp)


def norne_like():
    """Norne's Cartesian box 46 x 112 x 22 with 60 % of the cells inactive (~45 k active), fault-style NNCs (5 % extra connections),
    threshold pressures, 36 wells (4 water injectors, producers on BHP or oil-rate control)."""
    rng = np.random.default_rng(44)
    act = rng.random(46 * 112 * 22) > 0.6
    grid = decks.cartesian_grid(46, 112, 22, dx=80.0, dy=80.0, dz=4.0, tops=2500.0, actnum=act, nnc_fraction=0.05, lognormal_sigma=1.0,
                                thpres=0.02 * decks.BAR, seed=44)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, p_ref=270.0 * decks.BAR, z_ref=2500.0, perturb=0.005, seed=44)
    return grid, tab, st, W.column_wells(grid, 36, n_injectors=4, seed=44, inj_rate_m3_per_day=300.0, prod_bhp_bar=200.0, prod_oil_rate_m3_per_day=40.0)


def spe9_like():
    """SPE9 dimensions (24 x 25 x 15, 300 ft cells), 26 wells: one water injector completed in layers 11-15 and 25 producers in
    layers 2-4 (half on BHP, half on oil-rate control)."""
    grid = decks.cartesian_grid(24, 25, 15, dx=91.44, dy=91.44, dz=6.0, tops=2743.0, lognormal_sigma=1.0, seed=9)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, p_ref=248.0 * decks.BAR, z_ref=2743.0, perturb=0.005, seed=9)
    return grid, tab, st, W.column_wells(grid, 26, n_injectors=1, seed=9, inj_layers=range(10, 15), prod_layers=range(1, 4),
                                         inj_rate_m3_per_day=800.0, prod_bhp_bar=150.0, prod_oil_rate_m3_per_day=60.0)


def random_irregular(seed, options=False, bhp_limits=True):
    """One deck of the robustness sweep (tools/robust_sweep.py): a 20-50 x 20-50 x 5-20 box with up to 60 % of its cells inactive at random, NNCs and
    threshold pressures in some, sigma_lnK 0.3 .. 2, 3-19 vertical wells on mixed controls (rate-controlled wells with BHP limits, as real decks
    have them); options: VAPPARS / ROCKTAB / end-point scaling drawn at random as well.  Returns (grid, tables, state, wells, description)."""
    rng = np.random.default_rng(seed)
    nx, ny, nz = int(rng.integers(20, 50)), int(rng.integers(20, 50)), int(rng.integers(5, 20))
    inactive = float(rng.uniform(0.0, 0.6))
    kw = dict(dx=float(rng.uniform(30, 120)), dy=float(rng.uniform(30, 120)), dz=float(rng.uniform(2, 8)), tops=2500.0, lognormal_sigma=float(rng.uniform(0.3, 2.0)), seed=seed)
    if inactive > 0.05:
        kw["actnum"] = rng.random(nx * ny * nz) > inactive
    if rng.random() < 0.6:
        kw["nnc_fraction"] = float(rng.uniform(0.01, 0.06))
    if rng.random() < 0.4:
        kw["thpres"] = float(rng.uniform(0.01, 0.05)) * decks.BAR
    grid = decks.cartesian_grid(nx, ny, nz, **kw)
    tkw, opts = {}, []
    if options:
        if rng.random() < 0.4:
            tkw["vappars"] = (float(rng.uniform(0.1, 2.0)), float(rng.uniform(0.1, 2.0))); opts.append("vappars")
        if rng.random() < 0.4:
            tkw["rocktab"] = [(100.0, 0.97, 0.94), (200.0, 1.0, 1.0), (300.0, 1.02, 1.07), (500.0, 1.05, 1.1)]; opts.append("rocktab")
    tab = decks.satfunc_standard_tables(**tkw)
    if options and rng.random() < 0.5:
        grid = decks.with_endpoints(grid, decks.random_endpoints(grid, seed=seed)); opts.append("endscale")
    st = decks.initial_state(grid, tab, p_ref=270.0 * decks.BAR, z_ref=2500.0, perturb=0.005, seed=seed)
    nwells = int(rng.integers(3, 20))
    wl = W.column_wells(grid, nwells, n_injectors=max(1, nwells // 6), seed=seed, inj_rate_m3_per_day=float(rng.uniform(50, 400)),
                        prod_bhp_bar=float(rng.uniform(150, 230)), prod_oil_rate_m3_per_day=float(rng.uniform(10, 60)),
                        rate_wells_bhp_limits_bar=(450.0, 80.0) if bhp_limits else None)
    desc = "%dx%dx%d, %d active, inactive %.2f, sigma %.2f, %d wells%s" % (nx, ny, nz, grid.nc, inactive, kw["lognormal_sigma"], wl.nw, (" " + "+".join(opts)) if opts else "")
    return grid, tab, st, wl, desc


def cart60():
    """the bench deck's recipe at 60^3 = 216 k cells: the size at which the CPU checker's ILU0-preconditioned GMRES(40) is still
    affordable, so that device GMRES and checker GMRES can run whole time steps side by side"""
    grid = decks.cartesian_grid(60, 60, 60, lognormal_sigma=0.5, seed=12345)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, perturb=0.002, seed=12345)
    return grid, tab, st, W.five_spot(grid, rate_m3_per_day=300.0, bhp_prod_bar=150.0)


MAKERS = {"cart100": cart100, "spe10like": spe10_like, "nornelike": norne_like, "spe9like": spe9_like, "cart60": cart60}
# the step length each deck's parity tests and bench legs run at (days)
DT_DAYS = {"cart100": 5.0, "spe10like": 2.0, "nornelike": 3.0, "spe9like": 3.0, "cart60": 5.0}


def make(name):
    return MAKERS[name]()
