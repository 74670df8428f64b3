"""Deck ingest (opmgpu/deck.py) on the reference's own test decks (fixtures under tests/golden/decks, copied from
the reference's tests/ directory as data): the parsed tables equal the hand-transcribed ones used everywhere else, and the
known answers of tests/test_satfunc.cpp are reproduced end to end from the deck files."""
import json
import os

import numpy as np
import pytest

from opmgpu import deck, decks

HERE = os.path.dirname(__file__)
DECKS = os.path.join(HERE, "golden", "decks")
TABLE_ARRAYS = ["surface_density", "pvtw", "oil_node_ptr", "oil_rs", "oil_psat", "oil_invb_sat", "oil_invbmu_sat", "oil_col_ptr", "oil_col_p",
                "oil_col_invb", "oil_col_invbmu", "gas_node_ptr", "gas_pg", "gas_rvsat", "gas_invb_sat", "gas_invbmu_sat", "gas_col_ptr",
                "gas_col_rv", "gas_col_invb", "gas_col_invbmu", "swof_ptr", "swof_sw", "swof_krw", "swof_krow", "swof_pcow", "sgof_ptr",
                "sgof_sg", "sgof_krg", "sgof_krog", "sgof_pcgo"]


def _same_tables(a, b):
    for name in TABLE_ARRAYS:
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert (a.has_disgas, a.has_vapoil, a.rock_pref, a.rock_comp) == (b.has_disgas, b.has_vapoil, b.rock_pref, b.rock_comp)


@pytest.mark.parametrize("name", ["satfuncStandard.DATA", "satfuncEPSBase.DATA", "satfuncEPS_A.DATA", "satfuncEPS_C.DATA", "satfuncEPS_D.DATA"])
def test_satfunc_decks_give_the_transcribed_tables(name):
    _same_tables(deck.read_deck(os.path.join(DECKS, name)).tables(), decks.satfunc_standard_tables())


def test_fluid_data_deck():
    d = deck.read_deck(os.path.join(DECKS, "fluid.data"))
    _same_tables(d.tables(), decks.fluid_data_tables())
    assert d.dims == (1, 1, 1) and not d.has("PORO")        # a PROPS-only deck: its single cell has no pore volume


def test_parser_syntax(tmp_path):
    p = tmp_path / "T.DATA"
    p.write_text("""-- comment line
RUNSPEC
DIMENS
 2 2 1 /    -- trailing comment
OIL
GRID
DXV
 2*10.0 /
DYV
 10 10
/
DZV
 2 /
TOPS
 4*1000 /
PORO
 0.1 2*0.2 1*0.3 /
PERMX
 4*100 /
TITLE
 a title with WORDS
SCALECRS
 NO/
RPTSOL
 'PRES' 'SOIL -- not a comment' /
MULTZ
 1* 3*0.5 /
END
""")
    d = deck.read_deck(str(p))
    assert d.dims == (2, 2, 1)
    assert np.array_equal(d.array("PORO"), [0.1, 0.2, 0.2, 0.3]) and np.array_equal(d.array("DXV"), [10.0, 10.0])
    assert d.records("RPTSOL")[0] == ["PRES", "SOIL -- not a comment"] and d.records("SCALECRS")[0] == ["NO"]
    m = d.array("MULTZ")
    assert np.isnan(m[0]) and np.array_equal(m[1:], [0.5, 0.5, 0.5])
    assert d.has("OIL") and d.has("END") and not d.has("WATER")


def test_cartesian_geometry_matches_the_generator(tmp_path):
    nx, ny, nz = 5, 4, 3
    ref = decks.cartesian_grid(nx, ny, nz, dx=10.0, dy=20.0, dz=2.0, tops=2000.0, poro=0.25, permx_md=150.0, permz_ratio=0.1)
    n = nx * ny * nz
    p = tmp_path / "G.DATA"
    p.write_text("DIMENS\n %d %d %d /\nDXV\n %d*10 /\nDYV\n %d*20 /\nDZV\n %d*2 /\nTOPS\n %d*2000 /\nPORO\n %d*0.25 /\nPERMX\n %d*150 /\nPERMZ\n %d*15 /\n"
                 % (nx, ny, nz, nx, ny, nz, nx * ny, n, n, n))
    g = deck.read_deck(str(p)).grid()
    assert g.nc == ref.nc and np.array_equal(g.conn_cells, ref.conn_cells)
    assert np.allclose(g.trans, ref.trans, rtol=1e-14) and np.allclose(g.pv, ref.pv, rtol=1e-14) and np.allclose(g.z, ref.z, rtol=1e-14)


def test_ntg_mult_actnum(tmp_path):
    p = tmp_path / "N.DATA"
    p.write_text("DIMENS\n 3 1 2 /\nDXV\n 3*10 /\nDYV\n 10 /\nDZV\n 2*2 /\nTOPS\n 3*0 /\nPORO\n 6*0.2 /\nPERMX\n 6*100 /\nNTG\n 0.5 5*1 /\n"
                 "MULTX\n 1 0.25 4*1 /\nACTNUM\n 1 1 1 1 0 1 /\n")
    g = deck.read_deck(str(p)).grid()
    assert g.nc == 5
    assert g.pv[0] == pytest.approx(0.5 * g.pv[1])                   # NTG scales the pore volume ...
    h = 100 * decks.MD * 20.0 / 5.0
    t = {(int(a), int(b)): v for (a, b), v in zip(g.conn_cells, g.trans)}
    assert t[(0, 1)] == pytest.approx(1.0 / (1.0 / (0.5 * h) + 1.0 / h))          # ... and the horizontal half-transmissibility
    assert t[(1, 2)] == pytest.approx(0.25 * h / 2)                                # MULTX of cell 2 on its +x face
    assert (0, 3) in t and (2, 4) in t and all(4 not in k or k == (2, 4) for k in t)   # cell (2,1,2) inactive: (1,4) gone, ids compacted
    assert t[(0, 3)] == pytest.approx((100 * decks.MD * 100.0 / 1.0) / 2)           # vertical: no NTG


@pytest.mark.parametrize("case,fname", [("GwsegEPSBase", "satfuncEPSBase.DATA"), ("GwsegEPS_A", "satfuncEPS_A.DATA"), ("GwsegEPS_D", "satfuncEPS_D.DATA")])
def test_satfunc_known_answers_from_the_deck_files(oracle, case, fname):
    """tests/test_satfunc.cpp end to end: deck file -> tables + per-cell end points -> relperm, against the reference's numbers."""
    G = json.load(open(os.path.join(HERE, "golden", "satfunc_eps.json")))["cases"][case]
    d = deck.read_deck(os.path.join(DECKS, fname))
    tab, grid = d.tables(), d.grid()
    assert grid.nc in (10, 20) and grid.eps is not None
    if "endpoints" in G:
        for k, v in G["endpoints"].items():
            assert np.array_equal(grid.eps[decks.GridData.EPS_NAMES.index(k)], v), k
    n, tol = 11, G["reltol_percent"] / 100.0
    rows = G["krw"] if isinstance(G["krw"][0], list) else [G["krw"]]
    for icell in range(len(rows)):
        s = np.zeros((n, 3)); s[:, 0] = np.arange(n) * 0.1; s[:, 1] = 1.0 - s[:, 0]
        kr, dkr = oracle.relperm_eps(tab, grid, s, np.full(n, icell))
        pick = (lambda a: a[icell]) if isinstance(G["krw"][0], list) else (lambda a: a)
        for key, got in (("krw", kr[:, 0]), ("kro", kr[:, 1]), ("DkrwDsw", dkr[:, 0]), ("DkroDsw", dkr[:, 1]), ("DkroDsg", dkr[:, 7])):
            exp = np.asarray(pick(G[key]))
            assert np.all((np.abs(got - exp) <= tol * np.maximum(np.abs(got), np.abs(exp))) | ((np.abs(got) < 1e-14) & (np.abs(exp) < 1e-14))), (case, icell, key)


def test_transmissibility_multipliers_like_the_reference_test(tmp_path):
    """tests/test_transmissibilitymultipliers.cpp: a 2x2x2 deck with MULT? = 1..8 / MULT?- = 1..8 / NTG = 0.5 against the plain deck
    -- trans(face) * (inside cell + 1) with MULT?, * (outside cell + 1) with MULT?-, * 0.5 on horizontal faces with NTG."""
    pre = "RUNSPEC\nTABDIMS\n/\nOIL\nGAS\nWATER\nMETRIC\nDIMENS\n2 2 2/\nGRID\nDXV\n1.0 2.0 /\nDYV\n3.0 4.0 /\nDZV\n5.0 6.0/\nTOPS\n4*100 /\n"
    post = "PROPS\nPORO\n8*0.3 /\nPERMX\n8*1 /\nSCHEDULE\nTSTEP\n1.0 2.0 3.0 4.0 /\n"
    one8 = "1 2 3 4 5 6 7 8 /\n"
    texts = {"orig": pre + post,
             "mult": pre + "MULTX\n" + one8 + "MULTY\n" + one8 + "MULTZ\n" + one8 + post,
             "minus": pre + "MULTX-\n" + one8 + "MULTY-\n" + one8 + "MULTZ-\n" + one8 + post,
             "ntg": pre + "NTG\n8*0.5 /\n" + post}
    g = {}
    for k, t in texts.items():
        p = tmp_path / (k + ".DATA")
        p.write_text(t)
        g[k] = deck.read_deck(str(p)).grid()
    o = g["orig"]
    assert o.nc == 8 and o.nconn == 12
    for k in ("mult", "minus", "ntg"):
        assert np.array_equal(g[k].conn_cells, o.conn_cells)
    inside, outside = o.conn_cells[:, 0], o.conn_cells[:, 1]
    assert np.all(inside < outside)
    assert np.allclose(g["mult"].trans, o.trans * (inside + 1), rtol=1e-8)
    assert np.allclose(g["minus"].trans, o.trans * (outside + 1), rtol=1e-8)
    same_layer = (inside // 4) == (outside // 4)
    assert np.allclose(g["ntg"].trans[same_layer], 0.5 * o.trans[same_layer], rtol=1e-8)
    assert np.allclose(g["ntg"].trans[~same_layer], o.trans[~same_layer], rtol=1e-8)


def _corner_point_deck(tmp_path, mutate=None, name="CP.DATA"):
    """tests/golden/decks/SCHEDULE_SMALL.DATA with its DXV / DYV / DZV / TOPS replaced by the equivalent COORD / ZCORN"""
    from opmgpu import deck as D
    src = open(os.path.join(os.path.dirname(__file__), "golden", "decks", "SCHEDULE_SMALL.DATA")).read()
    d0 = D.read_deck(os.path.join(os.path.dirname(__file__), "golden", "decks", "SCHEDULE_SMALL.DATA"))
    nx, ny, nz = d0.dims
    dx, dy, dz = d0._cell_sizes()
    xs = np.concatenate([[0.0], np.cumsum(dx[0, 0, :])]); ys = np.concatenate([[0.0], np.cumsum(dy[0, :, 0])])
    zt = 2500.0 + np.concatenate([np.zeros((1, ny, nx)), np.cumsum(dz, 0)[:-1]], 0)
    zb = zt + dz
    coord = np.zeros((ny + 1, nx + 1, 6))
    coord[..., 0] = xs[None, :]; coord[..., 1] = ys[:, None]; coord[..., 2] = zt.min()
    coord[..., 3] = xs[None, :]; coord[..., 4] = ys[:, None]; coord[..., 5] = zb.max()
    zcorn = np.zeros((nz, 2, ny, 2, nx, 2))
    zcorn[:, 0] = zt[:, :, None, :, None]; zcorn[:, 1] = zb[:, :, None, :, None]
    if mutate:
        coord, zcorn = mutate(coord, zcorn)
    a, b = src.index("DXV"), src.index("PORO")
    txt = (src[:a] + "COORD\n" + " ".join("%.12g" % v for v in coord.ravel()) + " /\nZCORN\n" + " ".join("%.12g" % v for v in zcorn.ravel()) + " /\n" + src[b:])
    path = os.path.join(str(tmp_path), name)
    open(path, "w").write(txt)
    return d0, D.read_deck(path)


def test_corner_point_grid(tmp_path):
    """COORD / ZCORN ingest (opmgpu/deck.py::_corner_point; opm-grid's geometry restated, no reference vectors): the corner-point form of a
    block-centred grid gives the block-centred transmissibilities, pore volumes and depths; a grid dipping along x keeps its volume and
    lowers the transmissibility of the tilted faces' neighbours consistently; across a fault the cells are connected through the
    overlaps of their faces (areas and centroids checked analytically); gaps between layers are refused."""
    d0, d1 = _corner_point_deck(tmp_path)
    g0, g1 = d0.grid(), d1.grid()
    assert np.array_equal(g0.conn_cells, g1.conn_cells)
    assert np.allclose(g1.trans, g0.trans, rtol=1e-11) and np.allclose(g1.pv, g0.pv, rtol=1e-12) and np.allclose(g1.z, g0.z, atol=1e-9)
    for a, b in zip(d0._cell_sizes(), d1._cell_sizes()):
        assert np.allclose(a, b, rtol=1e-12)

    def dip(coord, zcorn):          # every corner 0.05 m deeper per metre of x: a sheared (parallelepiped) grid of the same volume
        nz, _, ny, _, nx, _ = zcorn.shape
        xcorner = np.zeros((ny, 2, nx, 2))
        for xs in (0, 1):
            for ys in (0, 1):
                xcorner[:, ys, :, xs] = coord[ys:ys + ny, xs:xs + nx, 0]
        z2 = zcorn + 0.05 * xcorner[None, None]
        c2 = coord.copy(); c2[..., 2] = z2.min() - 1.0; c2[..., 5] = z2.max() + 1.0
        return c2, z2
    _, d2 = _corner_point_deck(tmp_path, dip, "DIP.DATA")
    g2 = d2.grid()
    assert np.allclose(g2.pv, g0.pv, rtol=1e-12)                                   # shearing along z keeps the volumes
    assert np.allclose(g2.z, g0.z + 0.05 * (np.arange(g0.nc) % 6 * 100.0 + 50.0), atol=1e-9)
    nxf = 5 * 5 * 3                                                                # x-faces come first in the connection list
    assert np.all(g2.trans[:nxf] < g0.trans[:nxf]) and np.allclose(g2.trans[:nxf], g0.trans[:nxf] / (1 + 0.05 ** 2), rtol=1e-12)
    assert np.allclose(g2.trans[nxf:], g0.trans[nxf:], rtol=1e-12)

    def fault(coord, zcorn):        # the columns i >= 3 thrown down by 4 m (layers are 10 m thick)
        z2 = zcorn.copy(); z2[:, :, :, :, 3:, :] += 4.0
        c2 = coord.copy(); c2[..., 5] += 4.0
        return c2, z2
    _, d3 = _corner_point_deck(tmp_path, fault, "FAULT.DATA")
    g3 = d3.grid()
    # across the fault a cell meets its own layer over 6 m of its 10 m and the layer above it (on the downthrown side) over 4 m; the
    # half-transmissibilities see the overlap's area and its centroid: K A_ov (dx/2) / ((dx/2)^2 + dz_c^2) with dz_c = 2 m resp. 3 m
    t0 = {tuple(c): t for c, t in zip(g0.conn_cells.tolist(), g0.trans)}
    t3 = {tuple(c): t for c, t in zip(g3.conn_cells.tolist(), g3.trans)}
    cell = lambda i, j, k: i + 6 * (j + 5 * k)          # noqa: E731
    for j in range(5):
        for k in range(3):
            same = t3[(cell(2, j, k), cell(3, j, k))]
            assert same == pytest.approx(t0[(cell(2, j, k), cell(3, j, k))] * 0.6 * 2500.0 / 2504.0, rel=1e-12)
            if k >= 1:
                up = t3[(cell(2, j, k), cell(3, j, k - 1))]
                assert up == pytest.approx(t0[(cell(2, j, k), cell(3, j, k))] * 0.4 * 2500.0 / 2509.0, rel=1e-12)
            assert (cell(2, j, k), cell(3, j, k + 1)) not in t3
    # everything away from the fault is untouched
    for (a, b), t in t0.items():
        if not (a % 6 == 2 and b % 6 == 3):
            assert t3[(a, b)] == pytest.approx(t, rel=1e-12)
    assert len(t3) == len(t0) + 5 * 2
    assert np.allclose(g3.pv, g0.pv, rtol=1e-12)

    def gap(coord, zcorn):          # layer 2 detached from layer 1
        z2 = zcorn.copy(); z2[2:] += 1.0
        c2 = coord.copy(); c2[..., 5] += 1.0
        return c2, z2
    _, d4 = _corner_point_deck(tmp_path, gap, "GAP.DATA")
    with pytest.raises(ValueError, match="gaps between layers"):
        d4.grid()


def test_scissor_fault_overlaps_tile_the_face(tmp_path):
    """a fault whose throw changes sign along the fault plane (the top / bottom edges of the two sides CROSS): the overlaps of one cell's
    face with the cells of the other side are polygons with crossing points as corners; together they tile the face exactly"""
    from opmgpu import deck as D

    def scissor(coord, zcorn):
        nz, _, ny, _, nx, _ = zcorn.shape
        ycorner = np.zeros((ny, 2))
        for ys in (0, 1):
            ycorner[:, ys] = coord[ys:ys + ny, 0, 1]
        shift = 6.0 * (ycorner / coord[-1, 0, 1] - 0.5)                  # -3 m at y = 0 ... +3 m at y = Ly
        z2 = zcorn.copy(); z2[:, :, :, :, 3:, :] += shift[None, None, :, :, None, None]
        c2 = coord.copy(); c2[..., 2] -= 4.0; c2[..., 5] += 4.0
        return c2, z2
    _, d = _corner_point_deck(tmp_path, scissor, "SCISSOR.DATA")
    g = d.grid()
    cp = d._corner_point()
    P = cp["P"]
    assert cp["fault_x"][:, 2].all() and not cp["fault_x"][:, [0, 1, 3, 4]].any()
    for j in range(5):
        FA = np.array([[P[1, j, 2, tb, ys, 1] for ys in (0, 1)] for tb in (0, 1)])            # the middle layer: covered by the other side's stack
        area = 0.0
        pieces = 0
        for kb in range(3):
            FB = np.array([[P[kb, j, 3, tb, ys, 0] for ys in (0, 1)] for tb in (0, 1)])
            ov = D.Deck._face_overlap(FA, FB)
            if ov is not None:
                area += np.linalg.norm(ov[0]); pieces += 1
        assert area == pytest.approx(100.0 * 10.0, rel=1e-12), (j, area)
        assert pieces >= 2
    # in the middle row (j = 2) the throw changes sign inside the cell: the middle layer touches all three layers of the other side
    conn = {tuple(c) for c in g.conn_cells.tolist()}
    cell = lambda i, j, k: i + 6 * (j + 5 * k)          # noqa: E731
    assert all((cell(2, 2, 1), cell(3, 2, kb)) in conn for kb in range(3))


def test_fault_multipliers_and_minpv(tmp_path):
    """FAULTS + MULTFLT scale every connection through the named faces -- the regular ones of a block-centred grid and the overlaps across a
    corner-point fault alike; MINPV deactivates small cells"""
    from opmgpu import deck as D
    src = open(os.path.join(os.path.dirname(__file__), "golden", "decks", "SCHEDULE_SMALL.DATA")).read()
    a = src.index("PORO")
    flt = "FAULTS\n 'F1' 3 3 1 5 1 3 'X' /\n 'F2' 1 6 2 2 1 1 'Y' /\n/\nMULTFLT\n 'F1' 0.1 /\n 'F2' 0.5 /\n/\n"
    path = os.path.join(str(tmp_path), "FLT.DATA")
    open(path, "w").write(src[:a] + flt + src[a:])
    g0 = D.read_deck(os.path.join(os.path.dirname(__file__), "golden", "decks", "SCHEDULE_SMALL.DATA")).grid()
    g1 = D.read_deck(path).grid()
    t0 = {tuple(c): t for c, t in zip(g0.conn_cells.tolist(), g0.trans)}
    t1 = {tuple(c): t for c, t in zip(g1.conn_cells.tolist(), g1.trans)}
    cell = lambda i, j, k: i + 6 * (j + 5 * k)          # noqa: E731
    for (x, y), t in t0.items():
        f = 1.0
        if x % 6 == 2 and y == x + 1:
            f = 0.1                                     # the X+ faces of the cells with I = 3
        if (x // 6) % 5 == 1 and y == x + 6 and x // 30 == 0:
            f = 0.5                                     # the Y+ faces of row J = 2 in the top layer
        assert t1[(x, y)] == pytest.approx(f * t, rel=1e-13)

    def fault(coord, zcorn):
        z2 = zcorn.copy(); z2[:, :, :, :, 3:, :] += 4.0
        c2 = coord.copy(); c2[..., 5] += 4.0
        return c2, z2
    _, d2 = _corner_point_deck(tmp_path, fault, "F.DATA")
    txt = open(os.path.join(str(tmp_path), "F.DATA")).read()
    b = txt.index("PORO")
    open(os.path.join(str(tmp_path), "F2.DATA"), "w").write(txt[:b] + "FAULTS\n 'F1' 3 3 1 5 1 3 'X' /\n/\nMULTFLT\n 'F1' 0.1 /\n/\nMINPV\n 1 /\n" + txt[b:])
    g2, g3 = d2.grid(), D.read_deck(os.path.join(str(tmp_path), "F2.DATA")).grid()
    t2 = {tuple(c): t for c, t in zip(g2.conn_cells.tolist(), g2.trans)}
    t3 = {tuple(c): t for c, t in zip(g3.conn_cells.tolist(), g3.trans)}
    assert set(t2) == set(t3)
    for (x, y), t in t2.items():
        assert t3[(x, y)] == pytest.approx((0.1 if (x % 6 == 2 and y % 6 == 3) else 1.0) * t, rel=1e-13)
    # MINPV above every pore volume: nothing is left
    open(os.path.join(str(tmp_path), "F3.DATA"), "w").write(txt[:b] + "MINPV\n 1e9 /\n" + txt[b:])
    assert D.read_deck(os.path.join(str(tmp_path), "F3.DATA")).grid().nc == 0
