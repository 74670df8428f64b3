"""Minimal ECLIPSE deck ingest for the hot path's static inputs.

What `flow_legacy` gets from opm-parser + `DerivedGeology` (opm/autodiff/GeoProps.hpp:84-195) + `BlackoilPropsAdFromDeck`
(opm/autodiff/BlackoilPropsAdFromDeck.cpp:60-240), restricted to what the device path consumes:

  RUNSPEC   DIMENS TABDIMS OIL WATER GAS DISGAS VAPOIL METRIC ENDSCALE
  GRID      DX DY DZ / DXV DYV DZV, TOPS (+BOX for the top layer) or DEPTHZ (flat) -- or COORD / ZCORN (corner-point, no faults) --, PORO PERMX PERMY PERMZ NTG ACTNUM MINPV, FAULTS + MULTFLT
            MULTX MULTY MULTZ MULTX- MULTY- MULTZ- MULTPV NNC
  PROPS     SWOF SGOF PVTO PVDO PVCDO PVTG PVDG PVTW DENSITY ROCK ROCKTAB VAPPARS SCALECRS (NO and YES) EHYSTR
            SWL SWCR SWU SOWCR SGL SGCR SGU SOGCR  KRW KRO KRG PCW PCG  ISWL ISWCR ISWU ISOWCR ISGL ISGCR ISGU ISOGCR
            (SATOPTS HYSTER in RUNSPEC switches the hysteresis on; EHYSTR item 2 = 0 and item 5 = KR -- Carlson, relative permeabilities
            only -- is the model the device implements)
  REGIONS   PVTNUM SATNUM IMBNUM FIPNUM (fluid-in-place regions of computeFluidInPlace; 1-based in the deck and in `fipnum()`)
  SOLUTION  PRESSURE SWAT SGAS RS RV (explicit initial state; EQUIL is outside the hot path)

Block-centred Cartesian geometry only (corner-point COORD/ZCORN needs opm-grid's processing, out of scope).  TPFA
transmissibilities as `tpfa_htrans_compute` / `tpfa_trans_compute` do for such cells: half-transmissibility
K A / (d/2) per side (horizontal ones times NTG, DerivedGeology :135-160), harmonic sum, times the MULT? of the face.
Everything else (SCHEDULE, SUMMARY, report keywords) is skipped.  METRIC units only.
"""
import re

import numpy as np

from . import decks
from .decks import BAR, DAY, FluidTables, GridData, MD, State

FLAG_KEYWORDS = {"RUNSPEC", "GRID", "EDIT", "PROPS", "REGIONS", "SOLUTION", "SUMMARY", "SCHEDULE", "END", "NOECHO", "ECHO", "OIL", "WATER",
                 "GAS", "DISGAS", "VAPOIL", "METRIC", "FIELD", "LAB", "FMTOUT", "FMTIN", "UNIFOUT", "UNIFIN", "RUNSUM", "SEPARATE", "ALL",
                 "INIT", "NOSIM", "ENDBOX", "EXCEL", "NOGGF", "NEWTRAN", "OLDTRAN"}
_KW = re.compile(r"^[A-Z][A-Z0-9_+\-]{0,7}$")
EPS_NAMES = GridData.EPS_NAMES


def _tokens(text):
    """comment-stripped tokens; quoted strings stay single tokens; '/' is its own token"""
    out = []
    for line in text.splitlines():
        # cut '--' comments outside quotes
        q, cut = False, len(line)
        for i, ch in enumerate(line):
            if ch == "'":
                q = not q
            elif not q and line.startswith("--", i):
                cut = i
                break
        line = line[:cut]
        toks = re.findall(r"'[^']*'|/|[^\s/']+", line)
        out.append(toks)
    return out


def _expand(tok):
    """'3*0.2' -> [0.2]*3 ; '2*' -> [None]*2 ; number / string otherwise"""
    m = re.match(r"^(\d+)\*(.*)$", tok)
    if m:
        n, v = int(m.group(1)), m.group(2)
        return [(_value(v) if v != "" else None)] * n
    return [_value(tok)]


def _value(tok):
    if tok.startswith("'"):
        return tok.strip("'")
    try:
        return float(tok.replace("D", "E").replace("d", "e"))
    except ValueError:
        return tok


class Deck:
    def __init__(self, keywords, schedule=None):
        self.kw = keywords              # name -> list of records (each a list of values), last occurrence wins; BOX-scoped ones keep their box
        self.schedule = schedule or []  # SCHEDULE section in deck order: [(keyword, records)], every occurrence (opmgpu/schedule.py)

    def has(self, name):
        return name in self.kw

    def records(self, name):
        return self.kw[name]["records"]

    def array(self, name, n=None, default=None):
        if name not in self.kw:
            return default
        v = [x for x in self.kw[name]["records"][0]]
        a = np.asarray([np.nan if x is None else x for x in v], dtype=float)
        if n is not None and a.size != n:
            raise ValueError("%s holds %d values, expected %d" % (name, a.size, n))
        return a

    # ---------------------------------------------------------------- RUNSPEC
    @property
    def dims(self):
        r = self.records("DIMENS")[0]
        return int(r[0]), int(r[1]), int(r[2])

    def tabdims(self):
        r = self.records("TABDIMS")[0] if self.has("TABDIMS") else []
        get = lambda i, d: int(r[i]) if len(r) > i and r[i] is not None else d
        return get(0, 1), get(1, 1)          # NTSFUN, NTPVT

    # ---------------------------------------------------------------- PROPS
    def tables(self):
        if self.has("FIELD") or self.has("LAB"):
            raise ValueError("only METRIC decks are supported")
        ntsfun, ntpvt = self.tabdims()
        dens = [[r[1], r[0], r[2]] for r in self.records("DENSITY")[:ntpvt]]            # deck order oil water gas -> (w, o, g)
        pvtw = [[r[0], r[1], r[2], r[3], r[4] if len(r) > 4 and r[4] is not None else 0.0] for r in self.records("PVTW")[:ntpvt]]
        disgas, vapoil = self.has("DISGAS"), self.has("VAPOIL")
        # oil
        if self.has("PVTO"):
            pvto = []
            for tab in _split_tables(self.records("PVTO")):
                rows = []
                for rec in tab:
                    rs, rest = rec[0], rec[1:]
                    rows.append((rs, [tuple(rest[i:i + 3]) for i in range(0, len(rest), 3)]))
                pvto.append(rows)
        elif self.has("PVDO"):
            pvto = [[(0.0, [tuple(rec[i:i + 3])]) for i in range(0, len(rec), 3)] for rec in self.records("PVDO")[:ntpvt]]
            disgas = False
        elif self.has("PVCDO"):
            # constant compressibility dead oil: two nodes far apart reproduce it when Co = Cv = 0 (the reference's tests/fluid.data)
            pvto = []
            for r in self.records("PVCDO")[:ntpvt]:
                pref, bo, co, mu = r[0], r[1], r[2], r[3]
                cv = r[4] if len(r) > 4 and r[4] is not None else 0.0
                if co != 0.0 or cv != 0.0:
                    raise ValueError("PVCDO with non-zero compressibility / viscosibility is not supported")
                pvto.append([(0.0, [(pref, bo, mu)]), (0.0, [(pref + 800.0, bo, mu)])])
            disgas = False
        else:
            raise ValueError("no oil PVT keyword (PVTO / PVDO / PVCDO)")
        # gas
        if self.has("PVTG"):
            pvtg = []
            for tab in _split_tables(self.records("PVTG")):
                rows = []
                for rec in tab:
                    pg, rest = rec[0], rec[1:]
                    rows.append((pg, [tuple(rest[i:i + 3]) for i in range(0, len(rest), 3)]))
                pvtg.append(rows)
        elif self.has("PVDG"):
            pvtg = [[(rec[i], [(0.0, rec[i + 1], rec[i + 2])]) for i in range(0, len(rec), 3)] for rec in self.records("PVDG")[:ntpvt]]
            vapoil = False
        else:
            raise ValueError("no gas PVT keyword (PVTG / PVDG)")
        swof = [[tuple(rec[i:i + 4]) for i in range(0, len(rec), 4)] for rec in self.records("SWOF")[:ntsfun]]
        sgof = [[tuple(rec[i:i + 4]) for i in range(0, len(rec), 4)] for rec in self.records("SGOF")[:ntsfun]]
        rock = (1.0, 0.0)
        if self.has("ROCK"):
            r = self.records("ROCK")[0]
            rock = (r[0], r[1])
        rocktab = None
        if self.has("ROCKTAB"):
            rec = self.records("ROCKTAB")[0]
            ncol = 5 if len(rec) % 5 == 0 and len(rec) % 3 != 0 else 3
            rocktab = [(rec[i], rec[i + 1], rec[i + 2]) for i in range(0, len(rec), ncol)]
        vappars = (0.0, 0.0)
        if self.has("VAPPARS"):
            r = self.records("VAPPARS")[0]
            vappars = (r[0], r[1])
        return FluidTables(density_wog=dens, pvtw=pvtw, pvto=pvto, pvtg=pvtg, swof=swof, sgof=sgof, rock=rock, disgas=disgas, vapoil=vapoil,
                           vappars=vappars, rocktab=rocktab)

    # ---------------------------------------------------------------- GRID
    # ---------------------------------------------------------------- corner-point geometry (COORD / ZCORN)
    def _corner_point(self):
        """Geometry of a corner-point grid WITHOUT faults: what the reference takes from opm-grid (processEclipseFormat + compute_geometry; a
        dependency outside its tree) for `DerivedGeology` (GeoProps.hpp:84-195), restated from the published algorithm: a cell is the
        hexahedron of its eight corners (the ZCORN depths on the four COORD pillars around it); every face is triangulated about the mean
        of its four nodes (area vector = sum of the triangle normals, centroid = area-weighted mean of the triangle centroids), the cell
        about the mean of its face centroids (volume = sum of the tetrahedra, centroid = volume-weighted mean).  Neighbouring columns
        whose shared pillar depths differ (faults) are connected cell by cell through the OVERLAP of their faces on the common pillar pair
        (`_fault_connections`); gaps between layers of one column are refused.
        Pinned only by consistency: the corner-point description of a block-centred grid reproduces the DX / DY / DZ / TOPS result
        (tests/test_deck_ingest.py)."""
        if getattr(self, "_cp", None) is not None:
            return self._cp
        nx, ny, nz = self.dims
        coord = self.array("COORD", 6 * (nx + 1) * (ny + 1)).reshape(ny + 1, nx + 1, 6)
        zc = self.array("ZCORN", 8 * nx * ny * nz).reshape(nz, 2, ny, 2, nx, 2)            # (k, top/bottom, j, y-side, i, x-side)
        # corner coordinates P[k, j, i, tb, ys, xs, xyz]
        P = np.zeros((nz, ny, nx, 2, 2, 2, 3))
        for ys in (0, 1):
            for xs in (0, 1):
                pil = coord[ys:ys + ny, xs:xs + nx]                                        # pillar of that corner, [ny, nx, 6]
                for tb in (0, 1):
                    z = zc[:, tb, :, ys, :, xs]                                            # [nz, ny, nx]
                    dzp = pil[..., 5] - pil[..., 2]
                    with np.errstate(divide="ignore", invalid="ignore"):
                        t = np.where(dzp != 0.0, (z - pil[..., 2]) / dzp, 0.0)
                    P[:, :, :, tb, ys, xs, 0] = pil[..., 0] + t * (pil[..., 3] - pil[..., 0])
                    P[:, :, :, tb, ys, xs, 1] = pil[..., 1] + t * (pil[..., 4] - pil[..., 1])
                    P[:, :, :, tb, ys, xs, 2] = z
        # faults: neighbouring columns whose shared pillar depths differ are connected through the overlaps of their faces (_fault_connections)
        tol = 1e-6 * max(1.0, float(np.abs(P[..., 2]).max()))
        fault_x = (np.abs(P[:, :, :-1, :, :, 1] - P[:, :, 1:, :, :, 0]).max(axis=(0, 3, 4, 5)) > tol) if nx > 1 else np.zeros((ny, 0), bool)      # [ny, nx-1]
        fault_y = (np.abs(P[:, :-1, :, :, 1] - P[:, 1:, :, :, 0]).max(axis=(0, 3, 4, 5)) > tol) if ny > 1 else np.zeros((0, nx), bool)            # [ny-1, nx]
        if nz > 1 and np.abs(P[:-1, :, :, 1] - P[1:, :, :, 0]).max() > tol:
            raise ValueError("corner-point grid with gaps between layers: not supported")

        def face(a, b, c, d):                        # nodes in order around the face, each [..., 3]
            m = 0.25 * (a + b + c + d)
            nodes = (a, b, c, d)
            N = np.zeros_like(m); cw = np.zeros_like(m); aw = np.zeros(m.shape[:-1])
            for q in range(4):
                u, v = nodes[q] - m, nodes[(q + 1) % 4] - m
                tn = 0.5 * np.cross(u, v)
                ta = np.linalg.norm(tn, axis=-1)
                N += tn; aw += ta
                cw += ta[..., None] * (m + nodes[q] + nodes[(q + 1) % 4]) / 3.0
            return cw / np.maximum(aw, 1e-300)[..., None], N, (m, nodes)
        c = lambda tb, ys, xs: P[:, :, :, tb, ys, xs]                                      # noqa: E731
        faces = {"x-": face(c(0, 0, 0), c(0, 1, 0), c(1, 1, 0), c(1, 0, 0)), "x+": face(c(0, 0, 1), c(0, 1, 1), c(1, 1, 1), c(1, 0, 1)),
                 "y-": face(c(0, 0, 0), c(0, 0, 1), c(1, 0, 1), c(1, 0, 0)), "y+": face(c(0, 1, 0), c(0, 1, 1), c(1, 1, 1), c(1, 1, 0)),
                 "z-": face(c(0, 0, 0), c(0, 0, 1), c(0, 1, 1), c(0, 1, 0)), "z+": face(c(1, 0, 0), c(1, 0, 1), c(1, 1, 1), c(1, 1, 0))}
        inner = sum(f[0] for f in faces.values()) / 6.0
        vol = np.zeros(inner.shape[:-1]); cen = np.zeros_like(inner)
        for fc, _, (m, nodes) in faces.values():
            for q in range(4):
                a, b = nodes[q], nodes[(q + 1) % 4]
                tv = np.abs(np.einsum("...i,...i->...", np.cross(a - inner, b - inner), m - inner)) / 6.0
                vol += tv
                cen += tv[..., None] * (inner + a + b + m) / 4.0
        cen /= np.maximum(vol, 1e-300)[..., None]
        ext = lambda lo, hi: np.linalg.norm(faces[hi][0] - faces[lo][0], axis=-1)           # noqa: E731  (cube dimensions between face centroids)
        self._cp = dict(vol=vol, cen=cen, faces=faces, dx=ext("x-", "x+"), dy=ext("y-", "y+"), dz=ext("z-", "z+"), ztop=faces["z-"][0][..., 2], zbot=faces["z+"][0][..., 2],
                        P=P, fault_x=fault_x, fault_y=fault_y)
        return self._cp

    @staticmethod
    def _face_overlap(A, B):
        """Overlap of two faces that lie between the same two pillars.  A, B: [2 (top, bottom), 2 (pillar), 3] corner points.  In the
        pillar-pair plane a face is the region between its top edge and its bottom edge, both straight from pillar 0 (u = 0) to pillar 1
        (u = 1); the overlap is the region between the deeper of the two top edges and the shallower of the two bottom edges where that
        gap is positive -- one interval of u, because the gap is concave in u.  Returns (area vector, centroid) or None."""
        lines = {"tA": A[0], "bA": A[1], "tB": B[0], "bB": B[1]}                  # name -> [2 (u = 0, 1), 3]
        zat = lambda name, u: lines[name][0][2] + u * (lines[name][1][2] - lines[name][0][2])       # noqa: E731
        us = {0.0, 1.0}
        names = list(lines)
        for a in range(4):
            for b in range(a + 1, 4):
                la, lb = lines[names[a]], lines[names[b]]
                d0, d1 = la[0][2] - lb[0][2], la[1][2] - lb[1][2]
                if d0 * d1 < 0.0:
                    us.add(d0 / (d0 - d1))
        us = sorted(us)
        gap = lambda u: min(zat("bA", u), zat("bB", u)) - max(zat("tA", u), zat("tB", u))            # noqa: E731
        eps = 1e-12 * max(1.0, abs(A[0][0][2]))
        pos = [u for u in us if gap(u) > eps]
        if not pos:
            return None
        lo, hi = us.index(pos[0]), us.index(pos[-1])
        u0 = us[lo - 1] if lo > 0 and gap(us[lo - 1]) > -eps else pos[0]           # the zero of the gap next to the positive stretch is a breakpoint
        u1 = us[hi + 1] if hi + 1 < len(us) and gap(us[hi + 1]) > -eps else pos[-1]
        span = [u for u in us if u0 <= u <= u1]
        if len(span) < 2 and gap(span[0]) <= eps:
            return None

        def pt(name, u):
            return lines[name][0] + u * (lines[name][1] - lines[name][0])
        upper = [pt("tA" if zat("tA", u) >= zat("tB", u) else "tB", u) for u in span]
        lower = [pt("bA" if zat("bA", u) <= zat("bB", u) else "bB", u) for u in reversed(span)]
        poly = []
        for q in upper + lower:
            if not poly or np.abs(q - poly[-1]).max() > eps:
                poly.append(q)
        if len(poly) > 1 and np.abs(poly[0] - poly[-1]).max() <= eps:
            poly.pop()
        if len(poly) < 3:
            return None
        poly = np.array(poly)
        m = poly.mean(0)
        N = np.zeros(3); cw = np.zeros(3); aw = 0.0
        for q in range(len(poly)):
            a, b = poly[q], poly[(q + 1) % len(poly)]
            tn = 0.5 * np.cross(a - m, b - m)
            ta = np.linalg.norm(tn)
            N += tn; aw += ta; cw += ta * (m + a + b) / 3.0
        if aw <= 0.0:
            return None
        return N, cw / aw

    def _fault_connections(self, cp, perm, ntg, mult, mult_minus, direction):
        """connections across the faulted column pairs of one direction: [(cell a, cell b, transmissibility)] with a on the - side"""
        nx, ny, nz = self.dims
        P, cen = cp["P"], cp["cen"]
        out = []
        mask = cp["fault_x"] if direction == "x" else cp["fault_y"]
        for j, i in zip(*np.nonzero(mask)):
            ja, ia, jb, ib = (j, i, j, i + 1) if direction == "x" else (j, i, j + 1, i)
            for ka in range(nz):
                # face of A towards B / of B towards A as [top / bottom][pillar 0 / 1]
                if direction == "x":
                    FA = np.array([[P[ka, ja, ia, tb, ys, 1] for ys in (0, 1)] for tb in (0, 1)])
                else:
                    FA = np.array([[P[ka, ja, ia, tb, 1, xs] for xs in (0, 1)] for tb in (0, 1)])
                for kb in range(nz):
                    if direction == "x":
                        FB = np.array([[P[kb, jb, ib, tb, ys, 0] for ys in (0, 1)] for tb in (0, 1)])
                    else:
                        FB = np.array([[P[kb, jb, ib, tb, 0, xs] for xs in (0, 1)] for tb in (0, 1)])
                    if FB[0, :, 2].min() >= FA[1, :, 2].max() or FB[1, :, 2].max() <= FA[0, :, 2].min():
                        continue                                                      # depth ranges do not meet
                    ov = self._face_overlap(FA, FB)
                    if ov is None:
                        continue
                    N, cf = ov
                    a, b = ia + nx * (ja + ny * ka), ib + nx * (jb + ny * kb)
                    h = []
                    for (kk, jj, ii), cell in (((ka, ja, ia), a), ((kb, jb, ib), b)):
                        cvec = cf - cen[kk, jj, ii]
                        h.append(abs(float(np.dot(cvec * perm[kk, jj, ii], N))) / float(np.dot(cvec, cvec)) * ntg[cell])
                    if h[0] <= 0.0 or h[1] <= 0.0:
                        continue
                    t = 1.0 / (1.0 / h[0] + 1.0 / h[1])
                    if mult is not None and not np.isnan(mult[a]):
                        t *= mult[a]
                    if mult_minus is not None and not np.isnan(mult_minus[b]):
                        t *= mult_minus[b]
                    out.append((a, b, t))
        return out

    def _cell_sizes(self):
        nx, ny, nz = self.dims
        n = nx * ny * nz
        if self.has("ZCORN"):
            cp = self._corner_point()
            return cp["dx"], cp["dy"], cp["dz"]

        def axis(full, vec, count, shape):
            if self.has(full):
                return self.array(full, n).reshape(nz, ny, nx)
            v = self.array(vec, count)
            return np.broadcast_to(v.reshape(shape), (nz, ny, nx)).copy()
        dx = axis("DX", "DXV", nx, (1, 1, nx)); dy = axis("DY", "DYV", ny, (1, ny, 1)); dz = axis("DZ", "DZV", nz, (nz, 1, 1))
        return dx, dy, dz

    def _face_multipliers(self):
        """MULTX / MULTX- / ... arrays combined with the fault multipliers: FAULTS (name I1 I2 J1 J2 K1 K2 face) names sets of cell faces,
        MULTFLT (name factor) multiplies the transmissibility of every connection through them (regular faces and the overlaps of a
        faulted corner-point pillar pair alike).  {"X": per-cell factor of its X+ face, "X-": of its X- face, ...}"""
        nx, ny, nz = self.dims
        n = nx * ny * nz
        out = {}
        for key in ("X", "X-", "Y", "Y-", "Z", "Z-"):
            a = self.array("MULT" + key, n)
            out[key] = np.ones(n) if a is None else np.where(np.isnan(a), 1.0, a)
        if self.has("FAULTS") and self.has("MULTFLT"):
            fac = {}
            for r in self.records("MULTFLT"):
                if r:
                    fac[str(r[0]).upper()] = fac.get(str(r[0]).upper(), 1.0) * float(r[1])
            idx = np.arange(n).reshape(nz, ny, nx)
            for r in self.records("FAULTS"):
                if not r or str(r[0]).upper() not in fac:
                    continue
                i1, i2, j1, j2, k1, k2 = (int(v) for v in r[1:7])
                face = str(r[7]).upper().replace("I", "X").replace("J", "Y").replace("K", "Z")
                cells = idx[k1 - 1:k2, j1 - 1:j2, i1 - 1:i2].ravel()
                out[face][cells] *= fac[str(r[0]).upper()]
        return out

    def grid(self, gravity=decks.GRAVITY):
        """GridData of the active cells: connections x- then y- then z-normal faces (grid face order), then the NNCs."""
        nx, ny, nz = self.dims
        n = nx * ny * nz
        dx, dy, dz = self._cell_sizes()
        cp = self._corner_point() if self.has("ZCORN") else None
        if cp is not None:
            top = None
        elif self.has("TOPS"):
            t = self.array("TOPS")
            top = t[:nx * ny].reshape(ny, nx) if t.size >= nx * ny else np.full((ny, nx), t[0])
        elif self.has("DEPTHZ"):
            top = np.full((ny, nx), self.array("DEPTHZ")[0])
        else:
            top = np.zeros((ny, nx))
        if cp is not None:
            zc = cp["cen"][..., 2].ravel()
        else:
            ztop = top[None, :, :] + np.concatenate([np.zeros((1, ny, nx)), np.cumsum(dz, axis=0)[:-1]], axis=0)
            zc = (ztop + 0.5 * dz).ravel()
        poro = self.array("PORO", n, np.full(n, 0.0))
        ntg = self.array("NTG", n, np.ones(n))
        kx = self.array("PERMX", n, np.zeros(n)) * MD
        ky = self.array("PERMY", n, kx / MD) * MD
        kz = self.array("PERMZ", n, kx / MD) * MD
        multpv = self.array("MULTPV", n, np.ones(n))
        act = self.array("ACTNUM", n, np.ones(n)) > 0
        vol = cp["vol"].ravel() if cp is not None else (dx * dy * dz).ravel()
        pv = poro * ntg * multpv * vol
        minpv = float(self.records("MINPV")[0][0]) if self.has("MINPV") and self.records("MINPV")[0] else 0.0
        act &= (pv > 0.0) & (pv >= minpv)                          # MINPV: cells below it are inactive (no PINCH connection across them)
        idx = np.arange(n).reshape(nz, ny, nx)
        dxr, dyr, dzr = dx.ravel(), dy.ravel(), dz.ravel()

        def faces(a, b, kperm, area_a, area_b, da, db, horiz, mult, mult_minus):
            a, b = a.ravel(), b.ravel()
            h1 = kperm[a] * area_a[a] / (da[a] / 2.0); h2 = kperm[b] * area_b[b] / (db[b] / 2.0)
            if horiz:
                h1, h2 = h1 * ntg[a], h2 * ntg[b]
            with np.errstate(divide="ignore", invalid="ignore"):
                t = 1.0 / (1.0 / h1 + 1.0 / h2)
            t = np.where(np.isfinite(t), t, 0.0)
            if mult is not None:
                t = t * np.where(np.isnan(mult[a]), 1.0, mult[a])              # MULTX: the face towards +x of the cell it is given for
            if mult_minus is not None:
                t = t * np.where(np.isnan(mult_minus[b]), 1.0, mult_minus[b])  # MULTX-: the face towards -x of that cell
            return np.stack([a, b], 1), t
        ayz, axz, axy = dyr * dzr, dxr * dzr, dxr * dyr
        fm = self._face_multipliers()
        if cp is not None:
            # tpfa_htrans_compute (opm-core, restated): hT = |c . K n| / (c . c) with c = face centroid - cell centroid, n = the face's area
            # vector, K the diagonal permeability tensor; NTG on the horizontal faces (DerivedGeology, GeoProps.hpp:121-159)
            perm = np.stack([kx, ky, kz], -1).reshape(nz, ny, nx, 3)

            def cp_faces(a, b, lo, hi, horiz, mult, mult_minus):
                def half(sel, which):
                    fc, N, _ = cp["faces"][which]
                    cvec = fc[sel] - cp["cen"][sel]
                    h = np.abs(np.einsum("...i,...i->...", cvec * perm[sel], N[sel])) / np.einsum("...i,...i->...", cvec, cvec)
                    return h.ravel()
                sa = (slice(None, -1) if lo == "z+" else slice(None), slice(None, -1) if lo == "y+" else slice(None), slice(None, -1) if lo == "x+" else slice(None))
                sb = (slice(1, None) if lo == "z+" else slice(None), slice(1, None) if lo == "y+" else slice(None), slice(1, None) if lo == "x+" else slice(None))
                a, b = a.ravel(), b.ravel()
                h1, h2 = half(sa, lo), half(sb, hi)
                if horiz:
                    h1, h2 = h1 * ntg[a], h2 * ntg[b]
                with np.errstate(divide="ignore", invalid="ignore"):
                    t = 1.0 / (1.0 / h1 + 1.0 / h2)
                t = np.where(np.isfinite(t), t, 0.0)
                if mult is not None:
                    t = t * np.where(np.isnan(mult[a]), 1.0, mult[a])
                if mult_minus is not None:
                    t = t * np.where(np.isnan(mult_minus[b]), 1.0, mult_minus[b])
                return np.stack([a, b], 1), t
            cx, tx = cp_faces(idx[:, :, :-1], idx[:, :, 1:], "x+", "x-", True, fm["X"], fm["X-"])
            cy, ty = cp_faces(idx[:, :-1, :], idx[:, 1:, :], "y+", "y-", True, fm["Y"], fm["Y-"])
            cz, tz = cp_faces(idx[:-1, :, :], idx[1:, :, :], "z+", "z-", False, fm["Z"], fm["Z-"])
            # faulted column pairs: their layer-to-layer faces do not coincide; the connections come from the face overlaps instead
            tx = np.where(np.broadcast_to(cp["fault_x"][None], (nz,) + cp["fault_x"].shape).ravel(), 0.0, tx)
            ty = np.where(np.broadcast_to(cp["fault_y"][None], (nz,) + cp["fault_y"].shape).ravel(), 0.0, ty)
            fc = (self._fault_connections(cp, perm, ntg, fm["X"], fm["X-"], "x") + self._fault_connections(cp, perm, ntg, fm["Y"], fm["Y-"], "y"))
            if fc:
                cz = np.concatenate([cz, np.array([[a, b] for a, b, _ in fc], dtype=cz.dtype)])
                tz = np.concatenate([tz, np.array([t for _, _, t in fc])])
        else:
            cx, tx = faces(idx[:, :, :-1], idx[:, :, 1:], kx, ayz, ayz, dxr, dxr, True, fm["X"], fm["X-"])
            cy, ty = faces(idx[:, :-1, :], idx[:, 1:, :], ky, axz, axz, dyr, dyr, True, fm["Y"], fm["Y-"])
            cz, tz = faces(idx[:-1, :, :], idx[1:, :, :], kz, axy, axy, dzr, dzr, False, fm["Z"], fm["Z-"])
        conn = np.concatenate([cx, cy, cz]); trans = np.concatenate([tx, ty, tz])
        if self.has("NNC"):
            for r in self.records("NNC"):
                if not r:
                    continue
                c1 = int(r[0]) - 1 + nx * (int(r[1]) - 1) + nx * ny * (int(r[2]) - 1)
                c2 = int(r[3]) - 1 + nx * (int(r[4]) - 1) + nx * ny * (int(r[5]) - 1)
                conn = np.concatenate([conn, [[c1, c2]]]); trans = np.concatenate([trans, [r[6] * decks.CP * 1.0 / (DAY * BAR)]])
        newid = -np.ones(n, dtype=np.int64); newid[act] = np.arange(int(act.sum()))
        keep = act[conn[:, 0]] & act[conn[:, 1]] & (trans > 0.0)
        conn = newid[conn[keep]]; trans = trans[keep]
        self.active = np.flatnonzero(act)
        reg = lambda name: None if not self.has(name) else (self.array(name, n)[act].astype(np.int32) - 1)
        eps = self.endpoints()
        if eps is not None:
            eps = {k: v[act] for k, v in eps.items()}
        more = {}
        if eps is not None and self.has("SCALECRS") and str(self.records("SCALECRS")[0][0]).upper().startswith("Y"):
            more["scalecrs"] = True
        ev = {k: self.array(k, n)[act] * (BAR if k.startswith("PC") else 1.0) for k in GridData.EPSV_NAMES if self.has(k)}
        if ev:
            if any(np.isnan(v).any() for v in ev.values()):
                raise ValueError("vertical scaling arrays (KRW KRO KRG PCW PCG) must be given for every cell")
            more["eps_v"] = ev
        if self.hysteresis():
            imb = reg("IMBNUM")
            more["imbnum"] = imb if imb is not None else (reg("SATNUM") if reg("SATNUM") is not None else np.zeros(int(act.sum()), np.int32))
            ieps = self.endpoints(prefix="I", regions="IMBNUM")
            if ieps is not None and any(self.has("I" + k) for k in EPS_NAMES):
                more["ieps"] = {k: v[act] for k, v in ieps.items()}
        g = GridData(int(act.sum()), conn, trans, pv[act], zc[act], gravity=gravity, pvtnum=reg("PVTNUM"), satnum=reg("SATNUM"),
                     dims=(nx, ny, nz), eps=eps, **more)
        g.active_index = newid
        self._grid = g
        return g

    def fipnum(self):
        """FIPNUM of the active cells, 1-based as computeFluidInPlace takes it (0 = in no region); None without the keyword -- the
        reference then reports the whole field as one region (SimulatorBase_impl.hpp:140-150)."""
        if not self.has("FIPNUM"):
            return None
        nx, ny, nz = self.dims
        self.grid()                                   # (fixes the set of active cells)
        full = self.array("FIPNUM", nx * ny * nz)[self.active]
        return np.where(np.isnan(full), 0, full).astype(np.int32)

    def hysteresis(self):
        """SATOPTS HYSTER with the EHYSTR model the device implements (item 2 = 0: Carlson / drainage for the wetting phase, item 5 = KR)"""
        if not self.has("SATOPTS") or not any(str(x).upper().startswith("HYST") for r in self.records("SATOPTS") for x in r):
            return False
        if self.has("EHYSTR"):
            r = self.records("EHYSTR")[0]
            model = int(r[1]) if len(r) > 1 and r[1] is not None else 0
            what = str(r[4]).upper() if len(r) > 4 and r[4] is not None else "BOTH"
            if model != 0 or what != "KR":
                raise ValueError("EHYSTR: only item 2 = 0 (Carlson) with item 5 = KR is supported")
        return True

    def endpoints(self, prefix="", regions="SATNUM"):
        """ENDSCALE: per-cell scaled end points; the ones the deck does not give default to the cell's table values
        (EclEpsScalingPointsInfo::extractScaled falls back to the unscaled points).  prefix "I" / regions "IMBNUM": the
        imbibition curves' points (ISWL ...)."""
        if not self.has("ENDSCALE"):
            return None
        nx, ny, nz = self.dims
        n = nx * ny * nz
        t = self.tables()
        regkw = regions if self.has(regions) else "SATNUM"
        sat = (self.array(regkw, n).astype(int) - 1) if self.has(regkw) else np.zeros(n, int)
        un = np.zeros((t.n_sat, 8))
        for r in range(t.n_sat):
            a, b = t.swof_ptr[r], t.swof_ptr[r + 1]
            sw, krw, krow = t.swof_sw[a:b], t.swof_krw[a:b], t.swof_krow[a:b]
            a, b = t.sgof_ptr[r], t.sgof_ptr[r + 1]
            sg, krg, krog = t.sgof_sg[a:b], t.sgof_krg[a:b], t.sgof_krog[a:b]
            un[r] = [sw[0], last_zero(sw, krw), sw[-1], 1.0 - first_zero(sw, krow), sg[0], last_zero(sg, krg), sg[-1], 1.0 - first_zero(sg, krog)]
        out = {}
        for k, name in enumerate(EPS_NAMES):
            a = self.array(prefix + name, n, None)
            dflt = un[sat, k]
            out[name] = dflt if a is None else np.where(np.isnan(a), dflt, a)
        return out

    # ---------------------------------------------------------------- SOLUTION
    def initial_state(self, tables):
        """EQUIL (opmgpu/equil.py), else explicit PRESSURE / SWAT / SGAS / RS / RV arrays (active cells) with the hydrocarbon state from the
        saturations (FlowMain.hpp:626-673 takes the same two routes)"""
        if self.has("EQUIL") and not self.has("PRESSURE"):
            from . import equil
            return equil.from_deck(self, self._grid, tables)
        nx, ny, nz = self.dims
        n = nx * ny * nz
        act = self.active
        p = self.array("PRESSURE", n)[act] * BAR
        sw = self.array("SWAT", n, np.zeros(n))[act]; sg = self.array("SGAS", n, np.zeros(n))[act]
        rs = self.array("RS", n, np.zeros(n))[act]; rv = self.array("RV", n, np.zeros(n))[act]
        sat = np.stack([sw, 1.0 - sw - sg, sg], 1)
        from . import capi
        hc = np.where(sg > 0, np.where(sat[:, 1] > 0, capi.HC_GAS_AND_OIL, capi.HC_GAS_ONLY), capi.HC_OIL_ONLY).astype(np.int8)
        if not tables.has_disgas:
            hc[hc == capi.HC_OIL_ONLY] = capi.HC_GAS_AND_OIL
        return State(p, sat, rs, rv, hc)


def last_zero(x, y):
    """abscissa of the last leading zero of y (critical saturation: the curve leaves zero after it)"""
    i = 0
    while i + 1 < len(y) and y[i + 1] == 0.0:
        i += 1
    return x[i]


def first_zero(x, y):
    i = 0
    while i < len(y) - 1 and y[i] != 0.0:
        i += 1
    return x[i]


def _split_tables(records):
    """PVTO / PVTG: records of one table are closed by an empty record"""
    tabs, cur = [], []
    for r in records:
        if len(r) == 0:
            if cur:
                tabs.append(cur)
            cur = []
        else:
            cur.append(r)
    if cur:
        tabs.append(cur)
    return tabs


def read_deck(path):
    text = open(path).read()
    lines = _tokens(text)
    kws, cur, open_record = {}, None, False
    box = None
    skip_line = False
    schedule, in_schedule = [], False
    for toks in lines:
        if not toks:
            continue
        if skip_line:                 # TITLE: the next line is free text without a terminating slash
            skip_line = False
            continue
        first = toks[0]
        expects_data = cur is not None and not cur["closed"] and not cur["records"] and not cur["pending"]   # e.g. SCALECRS \n NO /
        if not open_record and not expects_data and _KW.match(first) and not _is_data(first):
            name = first
            cur = {"records": [], "pending": [], "closed": False, "box": box}
            if name == "ENDBOX":
                box = None
            kws[name] = cur
            if name == "SCHEDULE":
                in_schedule = True
            elif in_schedule:
                schedule.append((name, cur["records"]))
            toks = toks[1:]
            if name == "TITLE":
                cur["closed"] = True
                skip_line = True
                continue
            if name in FLAG_KEYWORDS:
                cur["closed"] = True
                continue
        if cur is None:
            continue
        for t in toks:
            if t == "/":
                cur["records"].append(cur["pending"]); cur["pending"] = []
                open_record = False
            else:
                cur["pending"].extend(_expand(t))
                open_record = True
    return Deck(kws, schedule)


def _is_data(tok):
    try:
        float(tok.replace("D", "E"))
        return True
    except ValueError:
        return False
