"""SCHEDULE ingest (opmgpu/schedule.py) and ECLIPSE binary output (opmgpu/eclio.py), host-side parts of SURVEY 8f-4."""
import datetime
import os

import numpy as np
import pytest

from opmgpu import deck as deckmod, decks, eclio, schedule, wells as W

DECK = os.path.join(os.path.dirname(__file__), "golden", "decks", "SCHEDULE_SMALL.DATA")


def _sched():
    d = deckmod.read_deck(DECK)
    g = d.grid()
    n = 90
    dx, dy, dz = d._cell_sizes()
    return d, g, schedule.Schedule(d, g, perm_md=(d.array("PERMX", n), d.array("PERMY", n)), dz=dz.ravel(), dxdy=(dx.ravel(), dy.ravel()), ntg=np.ones(n))


def test_schedule_report_steps_and_wells():
    d, g, s = _sched()
    assert s.start == datetime.date(2020, 1, 1)
    assert [x[0] / decks.DAY for x in s.steps] == [10.0, 10.0, 20.0]
    w0, w1 = s.wells(0), s.wells(1)
    assert w0.name == ["INJ", "PROD1", "PROD2"] and w0.type == [W.INJECTOR, W.PRODUCER, W.PRODUCER]
    assert w0.cells[w0.connpos[0]:w0.connpos[1]] == [0, 30, 60]                       # column (1,1), layers 1-3
    assert w0.cells[w0.connpos[1]:w0.connpos[2]] == [29, 59] and w0.cells[w0.connpos[2]:w0.connpos[3]] == [35]
    # controls: the deck's mode is the current control, its other limits are the constraints; producer rates negative
    inj = w0.controls[0]
    assert inj[0][0] == W.SURFACE_RATE and inj[0][1] == pytest.approx(400.0 / decks.DAY) and list(inj[0][2]) == [1.0, 0.0, 0.0]
    assert inj[1][0] == W.BHP and inj[1][1] == pytest.approx(320e5)
    p1 = w0.controls[1]
    assert p1[0][0] == W.SURFACE_RATE and p1[0][1] == pytest.approx(-150.0 / decks.DAY) and list(p1[0][2]) == [0.0, 1.0, 0.0] and p1[1][:2] == (W.BHP, pytest.approx(180e5))
    assert w0.controls[2][0][:2] == (W.BHP, pytest.approx(200e5))
    # the second WCONPROD changes PROD1 to BHP control at 190 bar from report step 1 on
    assert w1.controls[1][0][:2] == (W.BHP, pytest.approx(190e5)) and len(w1.controls[1]) == 1
    # connection factors: the given one (30 cP rm3/day/bar) and Peaceman's for the defaulted ones
    assert w0.WI[3] == pytest.approx(30.0 * schedule.CP_RM3_PER_DAY_BAR)
    kx, ky, dxy, dz, rw = 200 * decks.MD, 150 * decks.MD, 100.0, 10.0, 0.1
    r0 = 0.28 * np.sqrt(np.sqrt(ky / kx) * dxy ** 2 + np.sqrt(kx / ky) * dxy ** 2) / ((ky / kx) ** 0.25 + (kx / ky) ** 0.25)
    assert w0.WI[0] == pytest.approx(2 * np.pi * np.sqrt(kx * ky) * dz / np.log(r0 / rw), rel=1e-12)
    assert w0.WI[5] == pytest.approx(2 * np.pi * np.sqrt(kx * ky) * dz / (np.log(r0 / rw) + 1.5), rel=1e-12)      # skin


def test_eclipse_files_round_trip(tmp_path):
    d, g, s = _sched()
    st = d.initial_state(d.tables())
    base = str(tmp_path / "CASE")
    n = 90
    dx, dy, dz = d._cell_sizes()
    porv = np.zeros(n); porv[d.active] = g.pv
    out = eclio.EclOutput(base, d.dims, g.active_index, s.start, cell_sizes=(dx, dy, dz), tops=d.array("TOPS"), porv=porv)
    out.write_restart(0.0, st)
    wl = s.wells(0)
    ws = W.WellState(wl, st.p)
    ws.qs[1] = [-1e-4, -150.0 / decks.DAY, -0.2]
    st2 = st.copy(); st2.p += 1e5; st2.sat[:, 0] += 0.01; st2.sat[:, 1] -= 0.01
    out.write_restart(10.0, st2)
    out.write_summary(10.0, wl, ws, new_report_step=True)
    # restart: two report steps, headers and solution arrays in the documented order, big-endian, blocked by 1000
    rst = eclio.read_arrays(base + ".UNRST")
    names = [a[0] for a in rst]
    assert names == ["SEQNUM", "INTEHEAD", "LOGIHEAD", "DOUBHEAD", "STARTSOL", "PRESSURE", "SWAT", "SGAS", "RS", "RV", "ENDSOL"] * 2
    ih = rst[1][2]
    assert ih.size == 411 and ih[2] == 1 and list(ih[8:12]) == [6, 5, 3, 90] and ih[14] == 7 and list(ih[64:67]) == [1, 1, 2020]
    assert list(rst[12][2][64:67]) == [11, 1, 2020] and rst[14][2][0] == 10.0
    assert rst[5][1] == "REAL" and np.allclose(rst[5][2], st.p / 1e5, rtol=1e-7) and np.allclose(rst[16][2], st2.p / 1e5, rtol=1e-7)
    assert np.allclose(rst[17][2], st2.sat[:, 0], rtol=1e-7)
    raw = open(base + ".UNRST", "rb").read()
    assert raw[:4] == (16).to_bytes(4, "big") and raw[4:12] == b"SEQNUM  " and raw[16:20] == b"INTE"
    # grid: pillars and corner depths of the block-centred cells
    eg = {a[0]: a[2] for a in eclio.read_arrays(base + ".EGRID")}
    assert list(eg["GRIDHEAD"][:4]) == [1, 6, 5, 3] and eg["COORD"].size == 6 * 7 * 6 and eg["ZCORN"].size == 8 * 90 and eg["ACTNUM"].sum() == 90
    z = eg["ZCORN"].reshape(3, 2, 5, 2, 6, 2)
    assert np.all(z[0, 0] == 2500.0) and np.all(z[2, 1] == 2530.0)
    # summary: vector list and one ministep
    sp = {a[0]: a[2] for a in eclio.read_arrays(base + ".SMSPEC")}
    kws, wgn = list(sp["KEYWORDS"]), list(sp["WGNAMES"])
    assert kws[:2] == ["TIME", "YEARS"] and sp["DIMENS"][0] == len(kws) and list(sp["STARTDAT"][:3]) == [1, 1, 2020]
    sm = eclio.read_arrays(base + ".UNSMRY")
    assert [a[0] for a in sm] == ["SEQHDR", "MINISTEP", "PARAMS"]
    prm = sm[2][2]
    idx = lambda k, g: next(i for i, (a, b) in enumerate(zip(kws, wgn)) if a == k and b == g)      # noqa: E731
    assert prm[idx("TIME", ":+:+:+:+")] == 10.0 and prm[idx("WOPR", "PROD1")] == pytest.approx(150.0, rel=1e-6)
    assert prm[idx("FOPR", ":+:+:+:+")] == pytest.approx(150.0, rel=1e-6) and prm[idx("WBHP", "PROD2")] == pytest.approx(200.0)


def test_restarted_run_reproduces_the_full_run(tmp_path):
    """tests/run-restart-regressionTest.sh: a run restarted from a report step of the full run's UNRST must reproduce the rest of the full
    run within abs 2e-1 / rel 4e-5 (compareECLFiles.cmake:121-135).  Here with the CPU oracle + host well model behind the report-step
    driver (the driver, the restart reader and the file comparison are host code; the device path runs the same driver in
    tests/test_gpu_simulator.py)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle
    from opmgpu import capi, eclio, wells as W
    from opmgpu.simulator import Simulator
    from util import OracleBackend
    oracle.build()

    def oracle_model(grid, tables, params):
        return OracleBackend(oracle, grid, tables, params)

    def host_wells(model, wl, ws):
        if model.wells is None or list(model.wells[1]) != list(wl.arrays()[1]):
            model.wells = wl.arrays()
            model.rowptr, model.col = oracle.pattern(model.grid, *model.wells)
        return W.WellCoupledModel(model, W.StandardWellsHost(wl, model.grid.z, model.tab.surface_density[0]), ws)

    prm = capi.default_params(linear_solver_reduction=1e-8, linear_solver_maxiter=400)
    full, part = str(tmp_path / "FULL"), str(tmp_path / "RESTARTED")
    s1 = Simulator(DECK, params=prm, output_base=full, model_factory=oracle_model, well_model_factory=host_wells)
    r1 = s1.run()
    s2 = Simulator(DECK, params=prm, output_base=part, model_factory=oracle_model, well_model_factory=host_wells, restart=(full, 2))
    r2 = s2.run()
    assert [r["days"] for r in r2] == [r["days"] for r in r1[1:]]
    seq = [int(a[2][0]) for a in eclio.read_arrays(part + ".UNRST") if a[0] == "SEQNUM"]
    assert seq == [2, 3, 4]
    assert not eclio.compare(full, part, abs_tol=2e-1, rel_tol=4e-5, by_seqnum=True, summary=False)
    # the restart file's well state was used: the first restarted report step starts from the producers' rates of the full run
    x = eclio.read_restart(full, 2)
    assert "OPMGXWEL" in x and len(x["OPMGWNAM"]) == 3
    # and the comparison has teeth: a restart from the WRONG report step does not reproduce the full run
    wrong = str(tmp_path / "WRONG")
    s3 = Simulator(DECK, params=prm, output_base=wrong, model_factory=oracle_model, well_model_factory=host_wells, restart=(full, 2))
    s3.state0.p[:] *= 1.01
    s3.model.prepareStep(1.0, s3.state0)
    s3.run()
    assert eclio.compare(full, wrong, abs_tol=2e-1, rel_tol=4e-5, by_seqnum=True, summary=False)


def test_welopen_and_weltarg(tmp_path):
    """WELOPEN shuts / reopens a well or single completions, WELTARG changes one limit of the current control record"""
    from opmgpu import deck as deckmod, schedule as schedmod, wells as W
    src = open(DECK).read()
    a = src.index("DATES")
    extra = ("WELOPEN\n 'PROD2' 'SHUT' /\n 'INJ' 'SHUT' 0 0 3 /\n/\nWELTARG\n 'PROD1' 'ORAT' 90 /\n 'INJ' 'BHP' 300 /\n/\n")
    path = os.path.join(str(tmp_path), "WELOPEN.DATA")
    open(path, "w").write(src[:a] + extra + src[a:])
    d = deckmod.read_deck(path)
    grid = d.grid()
    nx, ny, nz = d.dims
    n = nx * ny * nz
    dx, dy, dz = d._cell_sizes()
    kx = d.array("PERMX", n)
    sch = schedmod.Schedule(d, grid, perm_md=(kx, d.array("PERMY", n, kx)), dz=dz.ravel(), dxdy=(dx.ravel(), dy.ravel()), ntg=np.ones(n))
    w0 = sch.wells(0)
    assert list(w0.name) == ["INJ", "PROD1"]                                  # PROD2 is shut from the first report step on
    inj = list(w0.name).index("INJ")
    assert w0.connpos[inj + 1] - w0.connpos[inj] == 2                         # its completion in layer 3 is shut
    p1 = list(w0.name).index("PROD1")
    tgt = [c for c in w0.controls[p1] if c[0] == W.SURFACE_RATE][0]
    assert tgt[1] == pytest.approx(-90.0 / 86400.0)
    bhp = [c for c in w0.controls[inj] if c[0] == W.BHP][0]
    assert bhp[1] == pytest.approx(300e5)


def test_fluid_in_place_restatement_and_summary_vectors(tmp_path):
    """The numpy restatement of computeFluidInPlace that checks the device (tests/util.py::OracleBackend.computeFluidInPlace) against
    independent facts: its per-cell water volume equals the oracle's accumulation term x pore volume (the term the assembly parity tests
    pin), region sums add up to the one-region sums, the weighted pressure is a pressure of the region; and on the summary of a run through
    the report-step driver: the change of FOIP / FWIP between two report steps equals what the wells produced / injected in between
    (implicit Euler: rates at the END of each ministep x its length, to the Newton tolerance)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle
    from opmgpu import capi, decks, eclio, wells as W
    from opmgpu.simulator import Simulator
    from util import OracleBackend
    oracle.build()
    tab = decks.satfunc_standard_tables()
    grid = decks.cartesian_grid(6, 5, 4, lognormal_sigma=0.4, seed=2)
    st = decks.random_state(grid, tab, seed=9)
    ob = OracleBackend(oracle, grid, tab, capi.default_params())
    ob.prepareStep(1 * decks.DAY, st)
    nc = grid.nc
    fipnum = np.random.default_rng(1).integers(0, 4, nc).astype(np.int32)
    values, cells = ob.computeFluidInPlace(fipnum, cells=True)
    props = oracle.cell_props(grid, tab, st)
    nm = oracle.PROP_NAMES
    assert np.allclose(cells[0], props[:, nm.index("accum_w"), 0] * grid.pv, rtol=1e-13)
    # oil + vaporised oil and gas + dissolved gas are the other two accumulation terms -- with the MODEL's rs / rv (saturated values where
    # the phase condition says so); computeFluidInPlace itself multiplies by the STATE's arrays (x.gasoilratio(), x.rv(), :2275-2276, :2294-2295)
    mrs, mrv = props[:, nm.index("rs"), 0], props[:, nm.index("rv"), 0]
    assert np.allclose(cells[1] + mrv * cells[2], props[:, nm.index("accum_o"), 0] * grid.pv, rtol=1e-12)
    assert np.allclose(cells[2] + mrs * cells[1], props[:, nm.index("accum_g"), 0] * grid.pv, rtol=1e-12)
    assert np.array_equal(cells[3], st.rs * cells[1]) and np.array_equal(cells[4], st.rv * cells[2])
    inside = fipnum > 0
    one = ob.computeFluidInPlace(np.where(inside, 1, 0).astype(np.int32))
    assert np.allclose(values[:, :6].sum(0), one[0, :6], rtol=1e-12)
    for r in range(values.shape[0]):
        pr = st.p[fipnum == r + 1]
        assert pr.min() <= values[r, 6] <= pr.max()

    def oracle_model(g, t, params):
        return OracleBackend(oracle, g, t, params)

    def host_wells(model, wl, ws):
        if model.wells is None or list(model.wells[1]) != list(wl.arrays()[1]):
            model.wells = wl.arrays()
            model.rowptr, model.col = oracle.pattern(model.grid, *model.wells)
        return W.WellCoupledModel(model, W.StandardWellsHost(wl, model.grid.z, model.tab.surface_density[0], tolerance_wells=1e-9), ws)

    base = str(tmp_path / "FIP")
    prm = capi.default_params(linear_solver_reduction=1e-10, linear_solver_maxiter=600, tolerance_mb=1e-10, tolerance_cnv=1e-7, tolerance_wells=1e-9)
    sim = Simulator(DECK, params=prm, output_base=base, model_factory=oracle_model, well_model_factory=host_wells)
    reps = sim.run()
    sp = {a[0]: a[2] for a in eclio.read_arrays(base + ".SMSPEC")}
    kws = list(sp["KEYWORDS"])
    rows = [a[2] for a in eclio.read_arrays(base + ".UNSMRY") if a[0] == "PARAMS"]
    col = lambda k: np.array([r[kws.index(k)] for r in rows], float)          # noqa: E731  (field vectors come first: unique keywords)
    foip, fwip, fpr, t = col("FOIP"), col("FWIP"), col("FPR"), col("TIME")
    assert (np.diff(foip) < 0).all() and (np.diff(fwip) > 0).all() and (fpr > 50).all() and (fpr < 600).all()
    # material balance between report steps with ONE ministep in between (rates of the end of the step, implicit Euler)
    checked = 0
    for i in range(1, len(rows)):
        if reps[i]["substeps"] != 1:
            continue
        checked += 1
        dt = t[i] - t[i - 1]
        assert fwip[i] - fwip[i - 1] == pytest.approx((col("FWIR")[i] - col("FWPR")[i]) * dt, rel=2e-4)
        assert foip[i - 1] - foip[i] == pytest.approx(col("FOPR")[i] * dt, rel=2e-4)
    assert checked >= 1          # (the summary stores 4-byte reals: the differences carry ~2e-5 relative noise)


def test_fipnum_regions_reach_the_report_steps(tmp_path):
    """REGIONS FIPNUM -> Deck.fipnum() -> the report-step driver's computeFluidInPlace(fipnum) (SimulatorBase_impl.hpp:278): one row of seven
    numbers per region at the end of every report step; the regions' volumes add up to the field's, cells with FIPNUM 0 belong to none."""
    import re
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle
    from opmgpu import capi, deck as deckmod, wells as W
    from opmgpu.simulator import Simulator
    from util import OracleBackend
    oracle.build()
    text = open(DECK).read()
    nx, ny, nz = (int(v) for v in re.search(r"DIMENS\s+(\d+)\s+(\d+)\s+(\d+)", text).groups())
    n = nx * ny * nz
    n1 = n // 3
    path = str(tmp_path / "FIPNUM.DATA")
    open(path, "w").write(text.replace("SOLUTION\n", "REGIONS\nFIPNUM\n %d*1 %d*2 4*0 /\n\nSOLUTION\n" % (n1, n - n1 - 4), 1))
    d = deckmod.read_deck(path)
    fn = d.fipnum()
    assert fn.shape == (n,) and (fn[:n1] == 1).all() and (fn[-4:] == 0).all() and fn.max() == 2
    assert deckmod.read_deck(DECK).fipnum() is None

    def oracle_model(g, t, params):
        return OracleBackend(oracle, g, t, params)

    def host_wells(model, wl, ws):
        if model.wells is None or list(model.wells[1]) != list(wl.arrays()[1]):
            model.wells = wl.arrays()
            model.rowptr, model.col = oracle.pattern(model.grid, *model.wells)
        return W.WellCoupledModel(model, W.StandardWellsHost(wl, model.grid.z, model.tab.surface_density[0]), ws)

    sim = Simulator(path, params=capi.default_params(linear_solver_reduction=1e-8, linear_solver_maxiter=400), model_factory=oracle_model,
                    well_model_factory=host_wells)
    reps = sim.run(max_steps=2)
    for r in reps:
        assert r["fip"].shape == (2, 7) and (r["fip"][:, :2] > 0).all() and (r["fip"][:, 2] >= 0).all() and (r["fip"][:, 3] > 0).all()      # (no free gas in this deck)
    field, cells = sim.model.computeFluidInPlace(None, cells=True)
    last = reps[-1]["fip"]
    assert np.allclose(last[:, :6].sum(0), field[0, :6] - cells[:6, -4:].sum(1), rtol=1e-12)      # the four cells outside every region
    assert last[:, 6].min() > 50e5 and last[:, 6].max() < 600e5


def test_record_layout_known_answer(tmp_path):
    """Byte-level known answer of the file format, independent of eclio.read_arrays: Fortran sequential records with 4-byte
    big-endian length markers, a 16-byte header record (8-char keyword, int32 count, 4-char type), data in blocks of at most 1000
    elements (105 for CHAR), everything big-endian -- parsed here with struct only."""
    import struct
    path = tmp_path / "X.BIN"
    vals = np.arange(2500, dtype=np.float32) * 0.5
    names = ["W%05d" % i for i in range(106)]
    with open(path, "wb") as f:
        eclio.write_array(f, "PRESSURE", "REAL", vals)
        eclio.write_array(f, "WGNAMES", "CHAR", names)
        eclio.write_array(f, "INTEHEAD", "INTE", [7, -3])
    raw = open(path, "rb").read()
    pos = 0

    def record():
        nonlocal pos
        n, = struct.unpack(">i", raw[pos:pos + 4]); body = raw[pos + 4:pos + 4 + n]; tail, = struct.unpack(">i", raw[pos + 4 + n:pos + 8 + n])
        assert tail == n
        pos += 8 + n
        return body

    h = record()
    assert len(h) == 16 and h[:8] == b"PRESSURE" and struct.unpack(">i", h[8:12])[0] == 2500 and h[12:] == b"REAL"
    blocks = [record() for _ in range(3)]
    assert [len(b) for b in blocks] == [4000, 4000, 2000]
    got = np.frombuffer(b"".join(blocks), dtype=">f4")
    assert np.array_equal(got, vals) and struct.unpack(">f", blocks[1][:4])[0] == 500.0
    h = record()
    assert h[:8] == b"WGNAMES " and struct.unpack(">i", h[8:12])[0] == 106 and h[12:] == b"CHAR"
    blocks = [record() for _ in range(2)]
    assert [len(b) for b in blocks] == [105 * 8, 8] and blocks[0][:8] == b"W00000  " and blocks[1] == b"W00105  "
    h = record()
    assert h[:8] == b"INTEHEAD" and struct.unpack(">i", h[8:12])[0] == 2 and h[12:] == b"INTE"
    assert struct.unpack(">ii", record()) == (7, -3) and pos == len(raw)
