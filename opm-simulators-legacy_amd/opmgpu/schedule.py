"""SCHEDULE ingest: what `flow_legacy` gets from opm-parser's Schedule + opm-core's WellsManager, restricted to the
keywords a standard-well black-oil run needs:

  WELSPECS  name group I J ref-depth preferred-phase            COMPDAT  name I J K1 K2 OPEN|SHUT satnum CF diameter Kh skin D dir
  WCONPROD  name OPEN|SHUT mode ORAT WRAT GRAT LRAT RESV BHP THP VFP ALQ     WCONINJE  name phase OPEN|SHUT mode RATE RESV BHP THP VFP
  WELOPEN   name OPEN|SHUT|STOP [I J K]                            WELTARG  name ORAT|WRAT|GRAT|LRAT|BHP|THP|RATE value
  DATES / TSTEP (report steps), START (RUNSPEC)

Every report step gets a `Wells` object (opmgpu/wells.py) the way WellsManager builds opm-core's `Wells` struct: one control per limit
the deck gives, the deck's control mode is the initial current control, the others are the inequality constraints
updateWellControls switches to (StandardWells_impl.hpp:709-800).  Producer rate targets are negative (flow into the wellbore).
A defaulted connection factor is Peaceman's for a vertical well in a block-centred cell (WellsManager::createWellsFromSpecs ->
computeWellIndices, opm-core, external: restated from the published formula).  METRIC units.
Not read: groups (GCONPROD ...), WCONHIST, multi-segment wells, horizontal completions (dir X / Y).  A RESV target becomes a
RESERVOIR_RATE control with distr {1, 1, 1}; opmgpu/rateconverter.py's computeRESV gives it the conversion coefficients once per report
step (SimulatorBase_impl.hpp:196, :476-553).
"""
import datetime

import numpy as np

from . import wells as W
from .decks import BAR, DAY

CP_RM3_PER_DAY_BAR = 1e-3 / (DAY * BAR)         # connection factor: cP rm3 / (day bar) -> SI
MONTHS = {m: i + 1 for i, m in enumerate(["JAN", "FEB", "MAR", "APR", "MAY", "JUN", "JUL", "AUG", "SEP", "OCT", "NOV", "DEC"])}
MONTHS["JLY"] = 7


def _get(rec, i, default=None):
    return rec[i] if len(rec) > i and rec[i] is not None else default


def _date(rec):
    return datetime.date(int(rec[2]), MONTHS[str(rec[1]).upper()[:3]], int(rec[0]))


class WellSpec:
    def __init__(self, name, i, j, ref_depth, phase):
        self.name, self.i, self.j, self.ref_depth, self.phase = name, i, j, ref_depth, phase
        self.completions = []           # (i, j, k, open, CF or None, diameter, Kh or None, skin)
        self.control = None             # ("PROD", open, mode, limits dict) / ("INJ", phase, open, mode, limits dict)


class Schedule:
    def __init__(self, deck, grid, perm_md=None, dz=None, dxdy=None, ntg=None):
        """deck: opmgpu.deck.Deck (already read); grid: the GridData made from it (active_index maps Cartesian -> active cell);
        perm_md = (kx, ky) per Cartesian cell [mD], dz / dxdy = (dx, dy) / ntg per Cartesian cell for the Peaceman factor."""
        self.deck, self.grid = deck, grid
        self.nx, self.ny, self.nz = deck.dims
        self.perm, self.dz, self.dxdy, self.ntg = perm_md, dz, dxdy, ntg
        self.start = _date(deck.records("START")[0]) if deck.has("START") else datetime.date(1983, 1, 1)
        self.steps = []                 # [(length in s, {name: WellSpec snapshot})]
        self._build()

    def _build(self):
        import copy
        specs = {}
        order = []
        now = self.start
        for name, recs in self.deck.schedule:
            if name == "WELSPECS":
                for r in recs:
                    if not r:
                        continue
                    wn = str(r[0])
                    if wn not in specs:
                        order.append(wn)
                    specs[wn] = WellSpec(wn, int(r[2]) - 1, int(r[3]) - 1, _get(r, 4), str(_get(r, 5, "OIL")).upper())
            elif name == "COMPDAT":
                for r in recs:
                    if not r:
                        continue
                    ws = specs[str(r[0])]
                    i = int(_get(r, 1, ws.i + 1) or ws.i + 1) - 1 if _get(r, 1, 0) not in (0, None) else ws.i
                    j = int(_get(r, 2, ws.j + 1) or ws.j + 1) - 1 if _get(r, 2, 0) not in (0, None) else ws.j
                    direction = str(_get(r, 12, "Z")).upper()
                    if direction != "Z":
                        raise ValueError("COMPDAT %s: only vertical completions (direction Z) are supported" % ws.name)
                    for k in range(int(r[3]) - 1, int(r[4])):
                        ws.completions = [c for c in ws.completions if (c[0], c[1], c[2]) != (i, j, k)]
                        ws.completions.append((i, j, k, str(_get(r, 5, "OPEN")).upper() == "OPEN", _get(r, 7), _get(r, 8, 0.3048), _get(r, 9), _get(r, 10, 0.0)))
                    ws.completions.sort(key=lambda c: c[2])
            elif name == "WCONPROD":
                for r in recs:
                    if not r:
                        continue
                    lim = {"ORAT": _get(r, 3), "WRAT": _get(r, 4), "GRAT": _get(r, 5), "LRAT": _get(r, 6), "RESV": _get(r, 7), "BHP": _get(r, 8, 1.01325),
                           "THP": _get(r, 9), "VFP": int(_get(r, 10, 0) or 0), "ALQ": _get(r, 11, 0.0)}
                    for wn in self._match(specs, str(r[0])):
                        specs[wn].control = ("PROD", str(_get(r, 1, "OPEN")).upper() == "OPEN", str(_get(r, 2, "")).upper(), lim)
            elif name == "WCONINJE":
                for r in recs:
                    if not r:
                        continue
                    lim = {"RATE": _get(r, 4), "RESV": _get(r, 5), "BHP": _get(r, 6, 6895.0), "THP": _get(r, 7), "VFP": int(_get(r, 8, 0) or 0)}
                    for wn in self._match(specs, str(r[0])):
                        specs[wn].control = ("INJ", str(r[1]).upper(), str(_get(r, 2, "OPEN")).upper() == "OPEN", str(_get(r, 3, "")).upper(), lim)
            elif name == "WELOPEN":          # name OPEN|SHUT|STOP [I J K C1 C2]: the whole well, or the completions that match (defaults / 0 = any)
                for r in recs:
                    if not r:
                        continue
                    status = str(_get(r, 1, "OPEN")).upper()
                    sel = [int(_get(r, q, 0) or 0) for q in (2, 3, 4)]
                    for wn in self._match(specs, str(r[0])):
                        ws = specs[wn]
                        if any(v > 0 for v in sel):
                            ws.completions = [(ci, cj, ck, status == "OPEN" if all(v <= 0 or v - 1 == c for v, c in zip(sel, (ci, cj, ck))) else op, cf, dia, kh, sk)
                                              for (ci, cj, ck, op, cf, dia, kh, sk) in ws.completions]
                        elif ws.control is not None:
                            c = list(ws.control)
                            c[2 if c[0] == "INJ" else 1] = status == "OPEN"          # STOP (shut above the formation) is treated as SHUT: no crossflow model
                            ws.control = tuple(c)
            elif name == "WELTARG":          # name control value: one limit of the current WCONPROD / WCONINJE record changed
                for r in recs:
                    if not r:
                        continue
                    key, val = str(r[1]).upper(), float(r[2])
                    for wn in self._match(specs, str(r[0])):
                        ws = specs[wn]
                        if ws.control is None:
                            raise ValueError("WELTARG %s before WCONPROD / WCONINJE" % wn)
                        lim = dict(ws.control[-1])
                        if ws.control[0] == "INJ" and key in ("ORAT", "WRAT", "GRAT"):
                            key = "RATE"
                        if key not in lim:
                            raise ValueError("WELTARG %s: control %s is not supported" % (wn, key))
                        lim[key] = val
                        ws.control = ws.control[:-1] + (lim,)
            elif name == "DATES":
                for r in recs:
                    if not r:
                        continue
                    d = _date(r)
                    self.steps.append(((d - now).days * DAY, copy.deepcopy({n: specs[n] for n in order})))
                    now = d
            elif name == "TSTEP":
                for r in recs:
                    for dt in r:
                        self.steps.append((float(dt) * DAY, copy.deepcopy({n: specs[n] for n in order})))
                        now = now + datetime.timedelta(days=float(dt))

    @staticmethod
    def _match(specs, pattern):
        if pattern.endswith("*"):
            return [n for n in specs if n.startswith(pattern[:-1])]
        return [pattern]

    def _peaceman(self, cart, diameter, skin):
        """connection factor of a vertical well in a block-centred cell [SI]: 2 pi sqrt(kx ky) dz ntg / (ln(r0 / rw) + skin),
        r0 = 0.28 sqrt(sqrt(ky/kx) dx^2 + sqrt(kx/ky) dy^2) / ((ky/kx)^(1/4) + (kx/ky)^(1/4))"""
        from .decks import MD
        kx, ky = self.perm[0][cart] * MD, self.perm[1][cart] * MD
        dx, dy, dz = self.dxdy[0][cart], self.dxdy[1][cart], self.dz[cart]
        r0 = 0.28 * np.sqrt(np.sqrt(ky / kx) * dx ** 2 + np.sqrt(kx / ky) * dy ** 2) / ((ky / kx) ** 0.25 + (kx / ky) ** 0.25)
        return 2.0 * np.pi * np.sqrt(kx * ky) * dz * (self.ntg[cart] if self.ntg is not None else 1.0) / (np.log(r0 / (0.5 * diameter)) + skin)

    def wells(self, step):
        """The `Wells` of report step `step` (open wells with open completions in active cells only; order of WELSPECS)."""
        out = W.Wells()
        act = np.asarray(self.grid.active_index)
        for name, ws in self.steps[step][1].items():
            if ws.control is None:
                continue
            is_inj = ws.control[0] == "INJ"
            is_open = ws.control[2] if is_inj else ws.control[1]
            if not is_open:
                continue
            cells, wi = [], []
            for (i, j, k, copen, cf, diam, kh, skin) in ws.completions:
                cart = i + self.nx * (j + self.ny * k)
                if not copen or act[cart] < 0:
                    continue
                cells.append(int(act[cart]))
                wi.append(cf * CP_RM3_PER_DAY_BAR if cf is not None else self._peaceman(cart, diam, skin))
            if not cells:
                continue
            ref = ws.ref_depth if ws.ref_depth is not None else float(self.grid.z[cells[0]])
            if is_inj:
                _, phase, _, mode, lim = ws.control
                comp = {"WATER": (1.0, 0.0, 0.0), "WAT": (1.0, 0.0, 0.0), "OIL": (0.0, 1.0, 0.0), "GAS": (0.0, 0.0, 1.0)}[phase]
                ctrls = {}
                if lim["RATE"] is not None:
                    ctrls["RATE"] = (W.SURFACE_RATE, lim["RATE"] / DAY, comp)
                if lim["RESV"] is not None:          # distr {1, 1, 1} until SimulatorBase::computeRESV fills in the conversion coefficients
                    ctrls["RESV"] = (W.RESERVOIR_RATE, lim["RESV"] / DAY, (1.0, 1.0, 1.0))
                if lim["BHP"] is not None:
                    ctrls["BHP"] = (W.BHP, lim["BHP"] * BAR)
                if lim["THP"] is not None and lim["VFP"] > 0:
                    ctrls["THP"] = (W.THP, lim["THP"] * BAR, None, lim["VFP"], 0.0)
                if mode == "GRUP":
                    raise ValueError("WCONINJE %s: control mode %s is not supported" % (name, mode))
                wtype = W.INJECTOR
            else:
                _, _, mode, lim = ws.control
                comp = (0.0, 1.0, 0.0)
                ctrls = {}
                for key, distr in (("ORAT", (0.0, 1.0, 0.0)), ("WRAT", (1.0, 0.0, 0.0)), ("GRAT", (0.0, 0.0, 1.0)), ("LRAT", (1.0, 1.0, 0.0))):
                    if lim[key] is not None:
                        ctrls[key] = (W.SURFACE_RATE, -lim[key] / DAY, distr)
                if lim["RESV"] is not None:
                    ctrls["RESV"] = (W.RESERVOIR_RATE, -lim["RESV"] / DAY, (1.0, 1.0, 1.0))
                if lim["BHP"] is not None:
                    ctrls["BHP"] = (W.BHP, lim["BHP"] * BAR)
                if lim["THP"] is not None and lim["VFP"] > 0:
                    ctrls["THP"] = (W.THP, lim["THP"] * BAR, None, lim["VFP"], lim["ALQ"] or 0.0)
                if mode in ("GRUP", "CRAT"):
                    raise ValueError("WCONPROD %s: control mode %s is not supported" % (name, mode))
                wtype = W.PRODUCER
            if mode not in ctrls:
                raise ValueError("well %s: control mode %r has no target in the deck" % (name, mode))
            # WellsManager's fixed order (ORAT, WRAT, GRAT, LRAT, [RESV], BHP, THP; injectors RATE, [RESV], BHP, THP) with the deck's
            # control mode as the index of the current control: updateWellControls switches to the FIRST broken constraint, so the
            # order decides which one wins when two are broken (and the index goes into the restart file)
            order = list(ctrls.values())
            out.add_well(name, wtype, ref, cells, wi, comp, order[0], limits=order[1:], current=list(ctrls).index(mode))
        return out
