"""Host-side standard well model: the reference's known answer for the segmented wellbore density, a
finite-difference check of the well Jacobians, and a well-driven Newton loop on the SPE1-like deck
(BASELINE configs[0]: 10x10x3 = 300 cells, 2 wells) entirely on the CPU through the oracle backend."""
import json
import os

import numpy as np
import pytest

from opmgpu import capi, decks, wells as W
from util import OracleBackend

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_welldensitysegmented_known_answer():
    g = json.load(open(os.path.join(GOLD, "welldensitysegmented.json")))
    wells = W.Wells()
    wells.add_well("INJ", W.INJECTOR, g["ref_depth"], g["cells"], 1.0, g["comp_frac_inj"], (W.BHP, 0.0))
    wells.add_well("PROD", W.PRODUCER, g["ref_depth"], g["cells"], 1.0, g["comp_frac_prod"], (W.BHP, 0.0))
    n = 10
    rates = np.asarray(g["perf_rates"]).reshape(n, 3)
    b = np.asarray(g["b_perf"]).reshape(n, 3)
    cd = W.connection_densities(wells, rates, b, np.asarray(g["rsmax_perf"]), np.asarray(g["rvmax_perf"]), np.asarray(g["surf_dens"]).reshape(n, 3))
    dp = W.connection_pressure_delta(wells, np.asarray(g["z_perf"]), cd, g["gravity"])
    ans = np.asarray(g["answer_over_gravity"]) * g["gravity"]
    assert np.allclose(dp, ans, rtol=1e-10)            # BOOST_CHECK_CLOSE(..., 1e-8) percent


def _setup(nx=10, ny=10, nz=3):
    grid = decks.cartesian_grid(nx, ny, nz, dx=300.0, dy=300.0, dz=10.0, tops=2500.0, poro=0.3, permx_md=200.0, lognormal_sigma=0.3)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, p_ref=250 * decks.BAR, z_ref=2500.0, gas_cap_fraction=0.0, gas_only_fraction=0.0)
    col = lambda i, j: [i + nx * j + nx * ny * k for k in range(nz)]
    wl = W.Wells()
    WI = 5.0 * float(np.median(grid.trans))
    wl.add_well("INJ", W.INJECTOR, grid.z[col(0, 0)[0]], col(0, 0), WI, (1.0, 0.0, 0.0), (W.SURFACE_RATE, 2000.0 / 86400.0, (1.0, 0.0, 0.0)))
    wl.add_well("PROD", W.PRODUCER, grid.z[col(nx - 1, ny - 1)[0]], col(nx - 1, ny - 1)[:2], WI, (0.0, 1.0, 0.0), (W.BHP, 200 * decks.BAR))
    return grid, tab, st, wl


def test_well_jacobian_finite_differences(oracle):
    grid, tab, st, wl = _setup()
    prm = capi.default_params()
    be = OracleBackend(oracle, grid, tab, prm, wells=wl.arrays())
    be.prepareStep(decks.DAY, st); be.assemble(True)
    pp = be.perfProps(wl.nperf).reshape(wl.nperf, 9, 4)
    wh = W.StandardWellsHost(wl, grid.z, tab.surface_density[0])
    ws = W.WellState(wl, st.p)
    ws.qs[0] = [2000.0 / 86400.0, 0.0, 0.0]; ws.bhp[0] = 300 * decks.BAR          # injecting
    wh.compute_connection_pressures(pp, ws)
    rd, rc, blocks, rhs = wh.assemble(pp, ws.copy())
    sys0 = [tuple(np.array(a) for a in s) for s in wh._sys]
    E0 = np.concatenate([wh.flux_eq, wh.ctrl_eq[:, None]], 1)
    # d E / d (q_s, bhp): perturb the well unknowns
    for w in range(wl.nw):
        D = np.linalg.inv(sys0[w][0])
        for k in range(4):
            h = 1e-7 * (abs(ws.qs[w, k]) + 1e-3) if k < 3 else 1.0
            wp = ws.copy()
            if k < 3:
                wp.qs[w, k] += h
            else:
                wp.bhp[w] += h
            wh.assemble(pp, wp)
            E1 = np.concatenate([wh.flux_eq, wh.ctrl_eq[:, None]], 1)
            fd = (E1[w] - E0[w]) / h
            assert np.allclose(fd, D[:, k], rtol=2e-5, atol=1e-9 * np.abs(D).max()), (w, k, fd, D[:, k])
    # d E / d cell variables: move perforation i's properties along their own d/dP, d/dSw, d/dXvar directions
    for w in range(wl.nw):
        lo = wl.connpos[w]
        C = sys0[w][1]
        for i in range(wl.connpos[w + 1] - lo):
            for d in range(3):
                t = [1e2, 1e-6, 1e-6][d]
                pp1 = pp.copy()
                pp1[lo + i, :, 0] += t * pp[lo + i, :, 1 + d]
                wh.assemble(pp1, ws.copy())
                E1 = np.concatenate([wh.flux_eq, wh.ctrl_eq[:, None]], 1)
                fd = (E1[w] - E0[w]) / t
                assert np.allclose(fd, C[:, 3 * i + d], rtol=5e-4, atol=1e-7 * np.abs(C[:, 3 * i + d]).max() + 1e-30), (w, i, d)


def test_spe1_like_well_driven_newton_on_cpu(oracle):
    """configs[0] plumbing: the 300-cell, 2-well case through the reference-shaped Newton loop on the CPU."""
    grid, tab, st, wl = _setup()
    prm = capi.default_params(ilu_ordering=capi.ORDER_NATURAL)
    be = OracleBackend(oracle, grid, tab, prm, wells=wl.arrays())
    wh = W.StandardWellsHost(wl, grid.z, tab.surface_density[0])
    ws = W.WellState(wl, st.p)
    model = W.WellCoupledModel(be, wh, ws)
    dt = 2 * decks.DAY
    water0 = None
    model.prepareStep(dt, st)
    for step in range(2):
        it = 0
        while True:
            conv, lin = model.nonlinearIteration(it, single_precision=False)
            it += 1
            if conv and it >= 1:
                break
            assert it <= 12, "Newton did not converge"
        # surface water volume in place, from the accumulation terms of the converged state
        props = oracle.cell_props(grid, tab, be.st)
        water = float((props[:, oracle.PROP_NAMES.index("accum_w"), 0] * grid.pv).sum())
        if water0 is not None:
            # implicit Euler mass balance of the water component: d(water in place) = dt * sum of the wells' water rates
            inj = ws.qs[0, 0] * dt
            net = ws.qs[:, 0].sum() * dt
            assert abs((water - water0) - net) <= 1e-4 * inj, (water - water0, net)
        water0 = water
        assert abs(ws.qs[0, 0] - 2000.0 / 86400.0) <= 1e-9 and ws.qs[1, 1] < 0          # injects water at target, produces oil
        assert abs(ws.bhp[1] - 200 * decks.BAR) < 1e-3
        model.prepareStep(dt)


def test_partition_keeps_wells_on_one_rank():
    """slab_partition(axis=1) + LocalDomain.local_wells: vertical wells of a 5-spot stay intact with slabs of j-rows, each well
    belongs to exactly one rank in that rank's local numbering; slabs of k-layers would cut them (refused)."""
    import pytest
    from opmgpu import partition
    grid = decks.cartesian_grid(10, 12, 4)
    wl = W.five_spot(grid)
    part = partition.slab_partition(grid, 3, axis=1)
    assert np.array_equal(np.unique(part), [0, 1, 2])
    seen = []
    for r in range(3):
        dom = partition.LocalDomain(grid, part, r)
        lw = dom.local_wells(wl, part)
        for k, w in enumerate(dom.well_index):
            loc = np.asarray(lw.cells[lw.connpos[k]:lw.connpos[k + 1]])
            assert np.all(loc < dom.n_owned)
            assert np.array_equal(dom.global_of_local[loc], wl.cells[wl.connpos[w]:wl.connpos[w + 1]])
            assert lw.ctrl_type[k] == wl.ctrl_type[w] and lw.ctrl_target[k] == wl.ctrl_target[w]
        seen += dom.well_index
    assert sorted(seen) == list(range(wl.nw))
    part_k = partition.slab_partition(grid, 2, axis=2)
    with pytest.raises(ValueError, match="straddles"):
        partition.LocalDomain(grid, part_k, 0).local_wells(wl, part_k)


def test_row_slabs_of_an_actnum_deck_keep_vertical_wells_whole():
    """slab_partition(axis = 1) on a deck with inactive cells (the Norne-like deck of BASELINE configs[4]): slabs of whole j-rows of the Cartesian
    box, balanced by ACTIVE cells, so that every vertical well stays on one rank (bench.py --deck nornelike --gpus N)."""
    from opmgpu import baseline_decks, partition
    g, _, _, wl = baseline_decks.norne_like()
    for n in (2, 4, 8):
        part = partition.slab_partition(g, n, axis=1)
        counts = np.bincount(part, minlength=n)
        assert counts.min() > 0.8 * g.nc / n and counts.max() < 1.2 * g.nc / n, counts
        seen = 0
        for r in range(n):
            seen += len(partition.LocalDomain(g, part, r).local_wells(wl, part).name)
        assert seen == wl.nw


def test_random_irregular_decks_are_reproducible_and_decompose_with_whole_wells():
    """opmgpu/baseline_decks.py::random_irregular (tools/robust_sweep.py, bench.py --deck random --seed N): the same seed gives the same deck, and
    slabs of j-rows keep every (vertical) well of every deck on one rank at 2, 3 and 4 ranks -- with inactive cells (balanced by active
    cells) and without."""
    from opmgpu import baseline_decks, partition
    for seed in range(8100, 8112):
        g, _, st, wl, desc = baseline_decks.random_irregular(seed)
        g2, _, st2, wl2, desc2 = baseline_decks.random_irregular(seed)
        assert desc == desc2 and g.nc == g2.nc and np.array_equal(g.trans, g2.trans) and np.array_equal(st.p, st2.p) and list(wl.cells) == list(wl2.cells)
        assert all(any(c[0] == W.BHP for c in wl.controls[w]) for w in range(wl.nw))          # every well has a BHP control or limit
        for n in (2, 3, 4):
            part = partition.slab_partition(g, n, axis=1)
            assert np.unique(part).size == n, (seed, n)
            seen = 0
            for r in range(n):
                seen += len(partition.LocalDomain(g, part, r).local_wells(wl, part).name)
            assert seen == wl.nw, (seed, n)


def test_weak_scaling_five_spots_live_inside_their_copy():
    """bench.py's weak-scaling decks: N copies of the workload with one 5-spot each, stacked along k (slab_axis 2) or side by side along j
    (slab_axis 1, --weak-axis 1).  Every well lies inside its copy, so the matching slab partition leaves every well on one rank."""
    from opmgpu import partition
    nx, ny, nz, n = 6, 5, 4, 3
    for axis, dims in ((1, (nx, ny * n, nz)), (2, (nx, ny, nz * n))):
        grid = decks.cartesian_grid(*dims)
        wl = W.five_spot(grid, slabs=n, slab_axis=axis)
        assert wl.nw == 5 * n
        part = partition.slab_partition(grid, n, axis=axis)
        owners = []
        for w in range(wl.nw):
            cells = np.asarray(wl.cells[wl.connpos[w]:wl.connpos[w + 1]])
            assert len(cells) == (nz if axis == 1 else nz)            # a full column of its copy
            r = np.unique(part[cells])
            assert r.size == 1
            owners.append(int(r[0]))
        assert sorted(owners) == sorted(list(range(n)) * 5)            # five wells per copy
        for r in range(n):
            partition.LocalDomain(grid, part, r).local_wells(wl, part)  # no well straddles


# ---- control logic: updateWellControls, solveWellEq, THP through VFP tables ------------------------------
def _limits_setup(inj_bhp_limit_bar=400.0, prod_rate_limit=None, thp=False):
    """SPE1-like deck; the injector is rate controlled with a BHP limit, the producer BHP controlled (optionally with an oil-rate
    limit, or THP controlled through a synthetic VFP table)."""
    from opmgpu import vfp
    nx, ny, nz = 10, 10, 3
    grid = decks.cartesian_grid(nx, ny, nz, dx=300.0, dy=300.0, dz=10.0, tops=2500.0, poro=0.3, permx_md=200.0, lognormal_sigma=0.3)
    tab = decks.satfunc_standard_tables()
    st = decks.initial_state(grid, tab, p_ref=250 * decks.BAR, z_ref=2500.0, gas_cap_fraction=0.0, gas_only_fraction=0.0)
    col = lambda i, j: [i + nx * j + nx * ny * k for k in range(nz)]
    wl = W.Wells()
    WI = 5.0 * float(np.median(grid.trans))
    wl.add_well("INJ", W.INJECTOR, grid.z[col(0, 0)[0]], col(0, 0), WI, (1.0, 0.0, 0.0), (W.SURFACE_RATE, 2000.0 / 86400.0, (1.0, 0.0, 0.0)),
                limits=[(W.BHP, inj_bhp_limit_bar * decks.BAR)])
    tables = []
    if thp:
        # bhp = 150 bar + 1.0 * thp + 2e9 Pa s/m3 * |oil rate|: linear in (thp, flo), so the multilinear table is exact
        thp_ax, flo_ax = np.array([10.0, 50.0, 100.0]) * decks.BAR, np.array([0.0, 0.005, 0.02, 0.05])
        data = np.zeros((3, 1, 1, 1, 4))
        for i, t in enumerate(thp_ax):
            data[i, 0, 0, 0, :] = 150 * decks.BAR + t + 2e9 * flo_ax
        tables.append(vfp.VFPProdTable(3, grid.z[col(nx - 1, ny - 1)[0]] - 5.0, vfp.FLO_OIL, vfp.WFR_WOR, vfp.GFR_GOR, flo_ax, thp_ax, [0.0], [0.0], [0.0], data))
        ctrl, lim = (W.THP, 30 * decks.BAR, None, 3, 0.0), [(W.BHP, 100 * decks.BAR)]
    else:
        ctrl = (W.BHP, 200 * decks.BAR)
        lim = [] if prod_rate_limit is None else [(W.SURFACE_RATE, -prod_rate_limit / 86400.0, (0.0, 1.0, 0.0))]
    wl.add_well("PROD", W.PRODUCER, grid.z[col(nx - 1, ny - 1)[0]], col(nx - 1, ny - 1)[:2], WI, (0.0, 1.0, 0.0), ctrl, limits=lim)
    return grid, tab, st, wl, tables


def test_update_well_controls_switches_to_the_broken_limit(oracle):
    grid, tab, st, wl, _ = _limits_setup(inj_bhp_limit_bar=300.0)
    wh = W.StandardWellsHost(wl, grid.z, tab.surface_density[0])
    ws = W.WellState(wl, st.p)
    ws.bhp[0] = 350 * decks.BAR                         # above the injector's BHP limit
    sw = wh.update_well_controls(ws)
    assert sw == [(0, 0, 1)] and ws.current[0] == 1 and ws.bhp[0] == 300 * decks.BAR
    # updateWellStateWithTarget re-applies the rate target while the rate control is current
    ws2 = W.WellState(wl, st.p); ws2.qs[0, 0] = 0.5
    wh.update_well_controls(ws2)
    assert ws2.current[0] == 0 and ws2.qs[0, 0] == 2000.0 / 86400.0
    # a producer breaks a rate limit from below (its rates are negative)
    grid, tab, st, wl, _ = _limits_setup(prod_rate_limit=100.0)
    wh = W.StandardWellsHost(wl, grid.z, tab.surface_density[0])
    ws = W.WellState(wl, st.p); ws.qs[1] = [-1e-4, -200.0 / 86400.0, -1e-2]
    wh.update_well_controls(ws)
    assert ws.current[1] == 1 and ws.qs[1, 1] == -100.0 / 86400.0 and ws.qs[1, 0] == -1e-4


def _time_step(model, dt, st, max_iter=15):
    model.prepareStep(dt, st)
    it = 0
    while True:
        conv, _ = model.nonlinearIteration(it, single_precision=False)
        it += 1
        if (conv and it > 1) or it > max_iter:
            return it, conv


def test_presolve_and_control_switch_in_a_time_step(oracle):
    """A rate-controlled injector whose BHP limit is too low for its rate target: the pre-solve (solveWellEq) drives the well
    equations to their tolerance before the first linearisation, updateWellControls switches the well to BHP control, and the
    converged step has bhp == limit with a rate below the target."""
    grid, tab, st, wl, _ = _limits_setup(inj_bhp_limit_bar=262.0)
    prm = capi.default_params(linear_solver_reduction=1e-10, linear_solver_maxiter=500)
    be = OracleBackend(oracle, grid, tab, prm, wells=wl.arrays())
    wh = W.StandardWellsHost(wl, grid.z, tab.surface_density[0])
    ws = W.WellState(wl, st.p)
    mo = W.WellCoupledModel(be, wh, ws)
    # the pre-solve alone: well residuals below their tolerances, reservoir untouched
    be.prepareStep(decks.DAY, st); be.assemble(True)
    pp = be.perfProps(wl.nperf).reshape(wl.nperf, 9, 4)
    ws0 = W.WellState(wl, st.p)
    wh.update_well_controls(ws0); wh.compute_connection_pressures(pp, ws0, be.perfPvtAt)
    ok, its = wh.solve_well_eq(pp, ws0, be.averageB(), be.perfPvtAt)
    assert ok and 1 <= its <= 15 and np.all(wh.well_flux_residual < 1e-4) and wh.well_ctrl_residual < 1e-7
    assert ws0.current[0] == 0 and 255 * decks.BAR < ws0.bhp[0] < 262 * decks.BAR     # with the reservoir frozen the rate target fits under the limit
    mo.prepareStep(5 * decks.DAY, st)
    conv, _ = mo.nonlinearIteration(0, single_precision=False)
    assert ws.current[0] == 0 and ws.bhp[0] > 262 * decks.BAR       # the first Newton update pushes bhp over the limit ...
    conv, _ = mo.nonlinearIteration(1, single_precision=False)
    assert ws.current[0] == 1                                       # ... and updateWellControls switches at the next assembly (mid-step)
    its, conv = _time_step(mo, 5 * decks.DAY, st)
    assert conv and its <= 10
    assert ws.current[0] == 1 and ws.bhp[0] == pytest.approx(262 * decks.BAR, rel=1e-12)
    assert 0 < ws.qs[0, 0] < 2000.0 / 86400.0 and ws.qs[1, 1] < 0
    # a limit below what the rate needs even with the reservoir frozen: the pre-solve itself switches
    grid, tab, st, wl, _ = _limits_setup(inj_bhp_limit_bar=255.0)
    be = OracleBackend(oracle, grid, tab, prm, wells=wl.arrays())
    wh = W.StandardWellsHost(wl, grid.z, tab.surface_density[0])
    be.prepareStep(decks.DAY, st); be.assemble(True)
    ws0 = W.WellState(wl, st.p)
    wh.update_well_controls(ws0); wh.compute_connection_pressures(pp, ws0, be.perfPvtAt)
    ok, its = wh.solve_well_eq(pp, ws0, be.averageB(), be.perfPvtAt)
    assert ok and ws0.current[0] == 1 and ws0.bhp[0] == 255 * decks.BAR and 0 < ws0.qs[0, 0] < 2000.0 / 86400.0
    # with a generous limit nothing switches and the rate target is met
    grid, tab, st, wl, _ = _limits_setup(inj_bhp_limit_bar=600.0)
    be = OracleBackend(oracle, grid, tab, prm, wells=wl.arrays())
    ws = W.WellState(wl, st.p)
    mo = W.WellCoupledModel(be, W.StandardWellsHost(wl, grid.z, tab.surface_density[0]), ws)
    its, conv = _time_step(mo, decks.DAY, st)
    assert conv and ws.current[0] == 0 and ws.qs[0, 0] == pytest.approx(2000.0 / 86400.0, rel=1e-9)


def test_thp_controlled_producer(oracle):
    """THP control: at convergence bhp = VFP table bhp(rates, thp target) - hydrostatic correction, and the well state's thp
    (updateWellState inverts the table) equals the target."""
    grid, tab, st, wl, tables = _limits_setup(thp=True)
    prm = capi.default_params(linear_solver_reduction=1e-10, linear_solver_maxiter=500)
    be = OracleBackend(oracle, grid, tab, prm, wells=wl.arrays())
    wh = W.StandardWellsHost(wl, grid.z, tab.surface_density[0], vfp_tables=tables)
    assert wh.vfp_active
    ws = W.WellState(wl, st.p)
    mo = W.WellCoupledModel(be, wh, ws)
    its, conv = _time_step(mo, decks.DAY, st)
    assert conv and ws.current[1] == 0
    t = tables[0]
    dp = wh.perf_dens[wl.connpos[1]] * wh.gravity * (t.datum_depth - wl.depth_ref[1])
    want = t.bhp(ws.qs[1, 0], ws.qs[1, 1], ws.qs[1, 2], 30 * decks.BAR, 0.0)[0] - dp
    assert ws.bhp[1] == pytest.approx(want, rel=1e-7)
    assert ws.thp[1] == pytest.approx(30 * decks.BAR, rel=1e-5)
    assert 150 * decks.BAR < ws.bhp[1] < 250 * decks.BAR and ws.qs[1, 1] < 0


def test_well_potentials_take_the_most_restrictive_bhp_limit(oracle):
    """computeWellPotentials (StandardWells_impl.hpp:1003-1095) walks a well's control list: a BHP control ASSIGNS its target, a THP control
    replaces it only when the bhp it implies (VFP table at the current rates, hydrostatic correction applied) is more restrictive -- larger
    for a producer.  The THP-controlled producer of this deck (THP 30 bar -> bhp > 150 bar; BHP limit 100 bar) is evaluated at the THP's bhp
    when the BHP limit comes first in the list and at 100 bar when it comes last (the assignment wins): the reference's order dependence,
    restated as it is.  A lower bhp means a larger drawdown, so the second potential is the larger one."""
    grid, tab, st, wl, tables = _limits_setup(thp=True)
    prm = capi.default_params(linear_solver_reduction=1e-10, linear_solver_maxiter=500)
    be = OracleBackend(oracle, grid, tab, prm, wells=wl.arrays())
    wh = W.StandardWellsHost(wl, grid.z, tab.surface_density[0], vfp_tables=tables)
    ws = W.WellState(wl, st.p)
    mo = W.WellCoupledModel(be, wh, ws)
    its, conv = _time_step(mo, decks.DAY, st)
    assert conv
    assert [c[0] for c in wl.controls[1]] == [W.THP, W.BHP]
    pot_thp_first = mo.computeWellPotentials()
    # at the converged state the well sits on its THP control: bhp(THP) is its current bhp, the BHP limit (100 bar) is far below
    v, _ = wh._bhp_from_thp(1, wl.controls[1][0], ws.qs[1])
    assert v == pytest.approx(ws.bhp[1], rel=1e-6) and v > 100 * decks.BAR
    wl.controls[1] = [wl.controls[1][1], wl.controls[1][0]]            # BHP limit first, THP control last
    pot_bhp_first = mo.computeWellPotentials()
    wl.controls[1] = [wl.controls[1][1], wl.controls[1][0]]
    # THP last: evaluated at the well's own bhp -> its current rates; BHP last: at 100 bar -> a larger production
    assert np.allclose(pot_bhp_first[1], ws.qs[1], rtol=1e-5, atol=1e-9 * np.abs(ws.qs[1]).max())
    assert pot_thp_first[1, 1] < pot_bhp_first[1, 1] < 0
    # the rate-controlled injector with a BHP limit: evaluated at the limit (400 bar), above its operating bhp -> more injection than its target
    assert pot_thp_first[0, 0] > ws.qs[0, 0] > 0 and np.allclose(pot_thp_first[0], pot_bhp_first[0])


def test_first_broken_constraint_wins_in_wellsmanagers_order(tmp_path):
    """updateWellControls switches to the FIRST broken constraint (StandardWells_impl.hpp:709-780), so the order of a well's controls
    decides which one wins when two are broken.  WellsManager keeps a fixed order (ORAT, WRAT, GRAT, LRAT, RESV, BHP, THP) and stores the
    current control as an index into it; opmgpu/schedule.py builds its wells the same way: a producer on BHP control with a
    WRAT and an ORAT limit both broken goes to ORAT (index 0), not to whichever limit the deck happens to list last."""
    from opmgpu import deck as deckmod, schedule
    src = open(os.path.join(GOLD, "decks", "SCHEDULE_SMALL.DATA")).read()
    # PROD1: current control BHP (index 2 of ORAT, WRAT, BHP), limits ORAT 150 and WRAT 20
    src = src.replace(" 'PROD1' 'OPEN' 'ORAT' 150 4* 180 /", " 'PROD1' 'OPEN' 'BHP' 150 20 3* 180 /", 1)
    path = tmp_path / "TWO_LIMITS.DATA"
    path.write_text(src)
    d = deckmod.read_deck(str(path))
    g = d.grid()
    n = g.nc
    dx, dy, dz = d._cell_sizes()
    s = schedule.Schedule(d, g, perm_md=(d.array("PERMX", n), d.array("PERMY", n)), dz=dz.ravel(), dxdy=(dx.ravel(), dy.ravel()), ntg=np.ones(n))
    wl = s.wells(0)
    w = wl.name.index("PROD1")
    types = [(c[0], tuple(c[2])) for c in wl.controls[w]]
    assert types == [(W.SURFACE_RATE, (0.0, 1.0, 0.0)), (W.SURFACE_RATE, (1.0, 0.0, 0.0)), (W.BHP, (0.0, 0.0, 0.0))]      # ORAT, WRAT, BHP
    assert wl.current0[w] == 2 and wl.ctrl_type[w] == W.BHP and wl.ctrl_target[w] == pytest.approx(180e5)
    ws = W.WellState(wl, np.full(g.nc, 250e5))
    assert ws.current[w] == 2 and ws.bhp[w] == pytest.approx(180e5)
    # the well produces far more oil AND water than its limits allow: both rate constraints are broken
    ws.qs[w] = [-100.0 / 86400.0, -400.0 / 86400.0, 0.0]
    wh = W.StandardWellsHost(wl, g.z, d.tables().surface_density[0])
    switched = wh.update_well_controls(ws)
    mine = [x for x in switched if x[0] == w]
    assert mine[0] == (w, 2, 0)                  # ORAT (first in the fixed order) wins the first switch, not WRAT
    # ... its target becomes the oil rate; the water limit is still broken then, so the loop goes on to WRAT and rests there
    assert mine[1:] == [(w, 0, 1)] and ws.current[w] == 1
    assert ws.qs[w, 1] == pytest.approx(-150.0 / 86400.0) and ws.qs[w, 0] == pytest.approx(-20.0 / 86400.0)


def test_resv_mode_becomes_a_reservoir_rate_control(tmp_path):
    """WCONPROD ... 'RESV': a RESERVOIR_RATE control in WellsManager's slot (after LRAT, before BHP) with distr {1, 1, 1} and a negative
    target; SimulatorBase::computeRESV (opmgpu/rateconverter.py) overwrites the distr once per report step"""
    from opmgpu import deck as deckmod, schedule
    from opmgpu.rateconverter import resv_control
    src = open(os.path.join(GOLD, "decks", "SCHEDULE_SMALL.DATA")).read()
    src = src.replace(" 'PROD1' 'OPEN' 'ORAT' 150 4* 180 /", " 'PROD1' 'OPEN' 'RESV' 150 3* 400 180 /", 1)
    path = tmp_path / "RESV.DATA"
    path.write_text(src)
    d = deckmod.read_deck(str(path))
    g = d.grid()
    n = g.nc
    dx, dy, dz = d._cell_sizes()
    s = schedule.Schedule(d, g, perm_md=(d.array("PERMX", n), d.array("PERMY", n)), dz=dz.ravel(), dxdy=(dx.ravel(), dy.ravel()), ntg=np.ones(n))
    wl = s.wells(0)
    w = wl.name.index("PROD1")
    types = [c[0] for c in wl.controls[w]]
    assert types == [W.SURFACE_RATE, W.RESERVOIR_RATE, W.BHP]                     # ORAT, RESV, BHP
    assert resv_control(wl.controls[w]) == 1 and wl.current0[w] == 1
    c = wl.controls[w][1]
    assert c[1] == pytest.approx(-400.0 / 86400.0) and tuple(c[2]) == (1.0, 1.0, 1.0)
