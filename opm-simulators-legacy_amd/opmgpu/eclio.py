"""ECLIPSE binary output: unified restart (.UNRST), grid (.EGRID), init (.INIT) and summary (.SMSPEC / .UNSMRY)
files, so that a run of this path can be diffed against a real `flow_legacy` run with opm-common's `compareECL` the way the
reference's regression tests do (tests/run-regressionTest.sh; tolerances compareECLFiles.cmake:83-118: abs 2e-2, rel 1e-5 on the
restart, and the summary compared keyword by keyword).

What `flow_legacy` writes through opm-common's EclipseIO (SimulatorFullyImplicitBlackoilOutput.cpp -> external), restated from the
published file format: Fortran sequential records (4-byte big-endian length before and after), every array = a 16-byte header record
(8-char keyword, int32 count, 4-char type INTE / REAL / DOUB / LOGI / CHAR / MESS) followed by its data in blocks of at most 1000
elements (105 for CHAR), all big-endian.  Units METRIC (pressure in bar, rates per day, volumes in m3).  No reference vectors for
the FILE FORMAT exist in /root/reference (EclipseIO is external): checked here by a reader written against the same description
(tests/test_schedule_eclio.py; one byte-level known answer of the record layout there is independent of that reader) -- parity unpinned.
"""
import datetime
import struct

import numpy as np

from .decks import BAR, DAY

_TYPES = {"INTE": (">i4", 4, 1000), "REAL": (">f4", 4, 1000), "DOUB": (">f8", 8, 1000), "LOGI": (">i4", 4, 1000), "CHAR": ("S8", 8, 105)}


def _record(f, payload):
    f.write(struct.pack(">i", len(payload))); f.write(payload); f.write(struct.pack(">i", len(payload)))


def write_array(f, name, typ, data):
    """one keyword: header record + blocked data records"""
    if typ == "MESS":
        _record(f, name.ljust(8).encode()[:8] + struct.pack(">i", 0) + b"MESS")
        return
    dt, size, block = _TYPES[typ]
    if typ == "CHAR":
        arr = np.asarray([str(s).ljust(8)[:8].encode() for s in data], dtype="S8")
    elif typ == "LOGI":
        arr = np.where(np.asarray(data, bool), -1, 0).astype(dt)          # .TRUE. is all bits set
    else:
        arr = np.ascontiguousarray(data).astype(dt)
    _record(f, name.ljust(8).encode()[:8] + struct.pack(">i", arr.size) + typ.encode())
    for i in range(0, arr.size, block):
        _record(f, arr[i:i + block].tobytes())


def read_arrays(path):
    """[(keyword, type, numpy array)] of a unified file (the checker of tests/test_schedule_eclio.py; also handy for diffs)"""
    out = []
    with open(path, "rb") as f:
        blob = f.read()
    pos = 0

    def rec():
        nonlocal pos
        n = struct.unpack(">i", blob[pos:pos + 4])[0]
        data = blob[pos + 4:pos + 4 + n]
        assert struct.unpack(">i", blob[pos + 4 + n:pos + 8 + n])[0] == n, "record length mismatch"
        pos += 8 + n
        return data
    while pos < len(blob):
        h = rec()
        name, count, typ = h[:8].decode().strip(), struct.unpack(">i", h[8:12])[0], h[12:16].decode()
        if typ == "MESS":
            out.append((name, typ, np.zeros(0)))
            continue
        dt, size, block = _TYPES[typ]
        parts, got = [], 0
        while got < count:
            d = rec()
            parts.append(np.frombuffer(d, dtype=dt))
            got += parts[-1].size
        a = np.concatenate(parts) if parts else np.zeros(0, dtype=dt)
        if typ == "CHAR":
            a = np.array([s.decode().strip() for s in a])
        elif typ == "LOGI":
            a = a != 0
        else:
            a = a.astype(a.dtype.newbyteorder("="))
        out.append((name, typ, a))
    return out


def _intehead(dims, nactive, date, nwells=0, ncwmax=0):
    """INTEHEAD (411 entries; the ones readers look at): [2] units (1 METRIC), [8..10] NX NY NZ, [11] NACTIV, [14] phase indicator
    (7 = oil + water + gas), [16] NWELLS, [17] NCWMAX (most completions of a well), [24..27] NIWELZ NSWELZ NXWELZ NZWELZ,
    [32..34] NICONZ NSCONZ NXCONZ, [64..66] day month year, [94] simulator (100 = ECLIPSE 100 conventions).  (The report step number is
    the SEQNUM keyword's, not an INTEHEAD entry.)"""
    ih = np.zeros(411, np.int32)
    ih[2] = 1
    ih[8], ih[9], ih[10], ih[11] = dims[0], dims[1], dims[2], nactive
    ih[14] = 7
    ih[16], ih[17] = nwells, ncwmax
    ih[24], ih[25], ih[26], ih[27] = 155, 122, 130, 3
    ih[32], ih[33], ih[34] = 25, 41, 58
    ih[64], ih[65], ih[66] = date.day, date.month, date.year
    ih[94] = 100
    ih[410] = 0
    return ih


class EclOutput:
    """BASE.EGRID / .INIT once, then BASE.UNRST / .SMSPEC / .UNSMRY per report step."""

    def __init__(self, base, dims, active_index, start_date, cell_sizes=None, tops=None, porv=None, extra_init=None, coord_zcorn=None):
        """dims (nx, ny, nz); active_index[cartesian cell] = active cell or -1; cell_sizes = (dx, dy, dz) per Cartesian cell and
        tops [ny*nx] for the EGRID of a block-centred grid; porv per Cartesian cell [m3]; extra_init = {keyword: per-active-cell array}"""
        self.base, self.dims = base, tuple(int(d) for d in dims)
        self.act = np.asarray(active_index)
        self.nactive = int((self.act >= 0).sum())
        self.start = start_date
        self.report = 0
        self.elapsed = 0.0
        self.ministep = 0
        self._smspec_written = False
        self._vectors = None
        open(base + ".UNRST", "wb").close(); open(base + ".UNSMRY", "wb").close()
        if coord_zcorn is not None:                       # a corner-point deck: its own COORD / ZCORN
            self._write_egrid_arrays(*coord_zcorn)
        elif cell_sizes is not None:
            self._write_egrid(cell_sizes, tops)
        self._write_init(porv, extra_init or {})

    # ---------------------------------------------------------------- EGRID (corner-point form of the block-centred grid)
    def _write_egrid(self, cell_sizes, tops):
        nx, ny, nz = self.dims
        dx, dy, dz = (np.asarray(a, float).reshape(nz, ny, nx) for a in cell_sizes)
        top = np.zeros((ny, nx)) if tops is None else np.asarray(tops, float).reshape(ny, nx)
        xs = np.concatenate([[0.0], np.cumsum(dx[0, 0, :])]); ys = np.concatenate([[0.0], np.cumsum(dy[0, :, 0])])
        zt = top[None] + np.concatenate([np.zeros((1, ny, nx)), np.cumsum(dz, 0)[:-1]], 0)
        zb = zt + dz
        coord = np.zeros((ny + 1, nx + 1, 6))             # vertical pillars
        coord[..., 0] = xs[None, :]; coord[..., 1] = ys[:, None]; coord[..., 2] = zt.min()
        coord[..., 3] = xs[None, :]; coord[..., 4] = ys[:, None]; coord[..., 5] = zb.max()
        zcorn = np.zeros((nz, 2, ny, 2, nx, 2))           # (k, top/bottom, j, y-side, i, x-side)
        zcorn[:, 0] = zt[:, :, None, :, None]; zcorn[:, 1] = zb[:, :, None, :, None]
        self._write_egrid_arrays(coord.ravel(), zcorn.ravel())

    def _write_egrid_arrays(self, coord, zcorn):
        nx, ny, nz = self.dims
        with open(self.base + ".EGRID", "wb") as f:
            fh = np.zeros(100, np.int32); fh[0] = 3; fh[1] = 2007; fh[4] = 0; fh[5] = 0
            write_array(f, "FILEHEAD", "INTE", fh)
            write_array(f, "GRIDUNIT", "CHAR", ["METRES", ""])
            gh = np.zeros(100, np.int32); gh[0] = 1; gh[1], gh[2], gh[3] = nx, ny, nz; gh[24] = 1; gh[25] = 1
            write_array(f, "GRIDHEAD", "INTE", gh)
            write_array(f, "COORD", "REAL", np.asarray(coord, float).ravel())
            write_array(f, "ZCORN", "REAL", np.asarray(zcorn, float).ravel())
            write_array(f, "ACTNUM", "INTE", (self.act >= 0).astype(np.int32))
            write_array(f, "ENDGRID", "INTE", np.zeros(0, np.int32))

    def _write_init(self, porv, extra):
        with open(self.base + ".INIT", "wb") as f:
            write_array(f, "INTEHEAD", "INTE", _intehead(self.dims, self.nactive, self.start))
            write_array(f, "LOGIHEAD", "LOGI", np.zeros(121, bool))
            write_array(f, "DOUBHEAD", "DOUB", np.zeros(229))
            if porv is not None:
                write_array(f, "PORV", "REAL", np.asarray(porv, float))          # PORV is per Cartesian cell
            for k, v in extra.items():
                write_array(f, k, "REAL", np.asarray(v, float))

    # ---------------------------------------------------------------- restart
    def write_restart(self, elapsed_days, state, extra=None, wells=None, well_state=None, next_step_days=None):
        """one report step: PRESSURE [bar], SWAT, SGAS, RS, RV per ACTIVE cell (the solution section `compareECL` diffs).  With wells the
        header section also carries what a restarted run needs of the well state, like flow_legacy's OPM_XWEL / OPM_IWEL do -- under this
        library's own keywords, its own layout: OPMGWNAM (names), OPMGXWEL (per well bhp [Pa], thp [Pa], q_s[3] [m3/s], DOUB), OPMGIWEL
        (current control); OPMGDTNX = the time stepper's suggestion for the next sub-step [days] (flow_legacy's restarts begin with the
        stepper's initial logic instead; carrying the suggestion makes the restarted run take the full run's sub-steps)."""
        self.report += 1
        self.elapsed = float(elapsed_days)
        date = self.start + datetime.timedelta(days=self.elapsed)
        with open(self.base + ".UNRST", "ab") as f:
            write_array(f, "SEQNUM", "INTE", [self.report])
            nw = wells.nw if wells is not None else 0
            ncw = max((wells.connpos[w + 1] - wells.connpos[w] for w in range(nw)), default=0)
            write_array(f, "INTEHEAD", "INTE", _intehead(self.dims, self.nactive, date, nwells=nw, ncwmax=ncw))
            write_array(f, "LOGIHEAD", "LOGI", np.zeros(121, bool))
            dh = np.zeros(229); dh[0] = self.elapsed
            write_array(f, "DOUBHEAD", "DOUB", dh)
            if wells is not None and well_state is not None and wells.nw > 0:
                ws = well_state
                write_array(f, "OPMGWNAM", "CHAR", [str(n)[:8] for n in wells.name])
                write_array(f, "OPMGXWEL", "DOUB", np.concatenate([[ws.bhp[w], ws.thp[w], *ws.qs[w]] for w in range(wells.nw)]))
                write_array(f, "OPMGIWEL", "INTE", np.asarray(ws.current, np.int32))
            if next_step_days is not None:
                write_array(f, "OPMGDTNX", "DOUB", [float(next_step_days)])
            write_array(f, "STARTSOL", "MESS", None)
            write_array(f, "PRESSURE", "REAL", state.p / BAR)
            write_array(f, "SWAT", "REAL", state.sat[:, 0])
            write_array(f, "SGAS", "REAL", state.sat[:, 2])
            write_array(f, "RS", "REAL", state.rs)
            write_array(f, "RV", "REAL", state.rv)
            for k, v in (extra or {}).items():
                write_array(f, k, "REAL", v)
            write_array(f, "ENDSOL", "MESS", None)

    # ---------------------------------------------------------------- summary
    def _write_smspec(self, well_names):
        """vectors: TIME, YEARS, field rates, then per well WBHP WOPR WWPR WGPR WWIR WGIR (positive rates, per day)"""
        kws, wgn, nums, units = ["TIME", "YEARS"], [":+:+:+:+", ":+:+:+:+"], [0, 0], ["DAYS", "YEARS"]
        for k, u in (("FOPR", "SM3/DAY"), ("FWPR", "SM3/DAY"), ("FGPR", "SM3/DAY"), ("FWIR", "SM3/DAY"), ("FGIR", "SM3/DAY"),
                     ("FOIP", "SM3"), ("FWIP", "SM3"), ("FGIP", "SM3"), ("FPR", "BARSA")):          # in place + average pressure: computeFluidInPlace
            kws.append(k); wgn.append(":+:+:+:+"); nums.append(0); units.append(u)
        for w in well_names:
            for k, u in (("WBHP", "BARSA"), ("WOPR", "SM3/DAY"), ("WWPR", "SM3/DAY"), ("WGPR", "SM3/DAY"), ("WWIR", "SM3/DAY"), ("WGIR", "SM3/DAY")):
                kws.append(k); wgn.append(w); nums.append(0); units.append(u)
        self._vectors = (kws, wgn)
        with open(self.base + ".SMSPEC", "wb") as f:
            write_array(f, "INTEHEAD", "INTE", [1, 100])
            write_array(f, "RESTART", "CHAR", [""] * 9)
            write_array(f, "DIMENS", "INTE", [len(kws), self.dims[0], self.dims[1], self.dims[2], 0, -1])
            write_array(f, "KEYWORDS", "CHAR", kws)
            write_array(f, "WGNAMES", "CHAR", wgn)
            write_array(f, "NUMS", "INTE", nums)
            write_array(f, "UNITS", "CHAR", units)
            write_array(f, "STARTDAT", "INTE", [self.start.day, self.start.month, self.start.year, 0, 0, 0])
        self._smspec_written = True

    def write_summary(self, elapsed_days, wells, well_state, new_report_step=False, fip=None):
        """one ministep: wells = opmgpu.wells.Wells (names, types), well_state = its WellState (bhp [Pa], qs [m3/s], w-o-g, production < 0);
        fip = computeFluidInPlace's row of the whole field (water, oil, gas, dissolved gas, vaporised oil, pore volume, hydrocarbon-pv
        weighted pressure [Pa]) or None"""
        names = list(wells.name)
        if not self._smspec_written:
            self._write_smspec(names)
        kws, wgn = self._vectors
        qs = np.asarray(well_state.qs) * DAY
        prod = np.maximum(-qs, 0.0); inj = np.maximum(qs, 0.0)
        row = {("TIME", ":+:+:+:+"): elapsed_days, ("YEARS", ":+:+:+:+"): elapsed_days / 365.25,
               ("FOPR", ":+:+:+:+"): prod[:, 1].sum(), ("FWPR", ":+:+:+:+"): prod[:, 0].sum(), ("FGPR", ":+:+:+:+"): prod[:, 2].sum(),
               ("FWIR", ":+:+:+:+"): inj[:, 0].sum(), ("FGIR", ":+:+:+:+"): inj[:, 2].sum()}
        if fip is not None:
            f7 = np.asarray(fip, float).reshape(-1)
            row[("FWIP", ":+:+:+:+")], row[("FOIP", ":+:+:+:+")], row[("FGIP", ":+:+:+:+")] = f7[0], f7[1] + f7[4], f7[2] + f7[3]
            row[("FPR", ":+:+:+:+")] = f7[6] / BAR
        for w, n in enumerate(names):
            row[("WBHP", n)] = well_state.bhp[w] / BAR
            row[("WOPR", n)], row[("WWPR", n)], row[("WGPR", n)] = prod[w, 1], prod[w, 0], prod[w, 2]
            row[("WWIR", n)], row[("WGIR", n)] = inj[w, 0], inj[w, 2]
        params = [row.get((k, g), 0.0) for k, g in zip(kws, wgn)]
        with open(self.base + ".UNSMRY", "ab") as f:
            if new_report_step or self.ministep == 0:
                write_array(f, "SEQHDR", "INTE", [self.report + 1])
            write_array(f, "MINISTEP", "INTE", [self.ministep])
            write_array(f, "PARAMS", "REAL", params)
        self.ministep += 1


def compare(base_a, base_b, abs_tol=2e-2, rel_tol=1e-5, restart_keywords=("PRESSURE", "SWAT", "SGAS", "RS", "RV"), by_seqnum=False, summary=True):
    """What the reference's regression tests ask of two runs (tests/run-regressionTest.sh -> compareECL, tolerances of
    compareECLFiles.cmake:83-85: abs 2e-2, rel 1e-5 or 1e-2): every value of every solution array of every report step of the UNRST
    files, and every summary vector of every ministep of the UNSMRY files, may deviate by at most abs_tol OR by at most rel_tol of the
    larger magnitude (opm-common ECLFilesComparator: a deviation counts only when both are exceeded).  Returns the list of violations
    [(file kind, keyword, occurrence, worst abs deviation, worst rel deviation)]; empty = the runs agree."""
    import numpy as np
    bad = []

    def check(kind, name, k, a, b):
        a, b = np.asarray(a, float), np.asarray(b, float)
        if a.shape != b.shape:
            bad.append((kind, name, k, float("inf"), float("inf")))
            return
        d = np.abs(a - b)
        with np.errstate(divide="ignore", invalid="ignore"):
            rel = np.where(d > 0, d / np.maximum(np.abs(a), np.abs(b)), 0.0)
        viol = (d > abs_tol) & (rel > rel_tol)
        if viol.any():
            bad.append((kind, name, k, float(d[viol].max()), float(rel[viol].max())))

    ra, rb = read_arrays(base_a + ".UNRST"), read_arrays(base_b + ".UNRST")
    if by_seqnum:
        # a restarted run against the full one (tests/run-restart-regressionTest.sh): the report steps both files hold, matched by SEQNUM
        def by_report(arrs):
            out, cur = {}, None
            for name, _, data in arrs:
                if name == "SEQNUM":
                    cur = int(data[0])
                elif cur is not None:
                    out.setdefault(cur, {})[name] = data
            return out
        ga, gb = by_report(ra), by_report(rb)
        common = sorted(set(ga) & set(gb))
        if not common:
            bad.append(("UNRST", "SEQNUM", -1, float("inf"), float("inf")))
        for r in common:
            for name in restart_keywords:
                if name in ga[r] and name in gb[r]:
                    check("UNRST", name, r, ga[r][name], gb[r][name])
    else:
        for name in restart_keywords:
            xa, xb = [x[2] for x in ra if x[0] == name], [x[2] for x in rb if x[0] == name]
            if len(xa) != len(xb):
                bad.append(("UNRST", name, -1, float("inf"), float("inf")))
                continue
            for k, (a, b) in enumerate(zip(xa, xb)):
                check("UNRST", name, k, a, b)
    if not summary:
        return bad
    sa, sb = {x[0]: x[2] for x in read_arrays(base_a + ".SMSPEC")}, {x[0]: x[2] for x in read_arrays(base_b + ".SMSPEC")}
    if list(sa["KEYWORDS"]) != list(sb["KEYWORDS"]) or list(sa["WGNAMES"]) != list(sb["WGNAMES"]):
        bad.append(("SMSPEC", "KEYWORDS", -1, float("inf"), float("inf")))
        return bad
    pa, pb = [x[2] for x in read_arrays(base_a + ".UNSMRY") if x[0] == "PARAMS"], [x[2] for x in read_arrays(base_b + ".UNSMRY") if x[0] == "PARAMS"]
    if len(pa) != len(pb):
        bad.append(("UNSMRY", "PARAMS", -1, float("inf"), float("inf")))
        return bad
    for k, (a, b) in enumerate(zip(pa, pb)):
        for i, (kw, wg) in enumerate(zip(sa["KEYWORDS"], sa["WGNAMES"])):
            check("UNSMRY", "%s:%s" % (str(kw).strip(), str(wg).strip()), k, [a[i]], [b[i]])
    return bad


def read_restart(base, report):
    """the arrays of report step `report` (its SEQNUM) of BASE.UNRST as {keyword: array}; "DAYS" = the elapsed time of DOUBHEAD"""
    out, cur = {}, None
    for name, typ, data in read_arrays(base + ".UNRST"):
        if name == "SEQNUM":
            cur = int(data[0])
            continue
        if cur == report:
            out[name] = data
    if not out:
        raise ValueError("report step %d is not in %s.UNRST" % (report, base))
    out["DAYS"] = float(out["DOUBHEAD"][0])
    return out
