"""VFP (vertical flow performance) tables for THP well control: host-side restatement.

What the reference reaches from StandardWells_impl.hpp:836-998 (addWellControlEq), :655-700 (thp update in updateWellState),
:1452-1550 (updateWellStateWithTarget) through
  VFPProdPropertiesLegacy::bhp / thp    opm/autodiff/VFPProdPropertiesLegacy.cpp:36-154
  VFPInjPropertiesLegacy::bhp / thp     opm/autodiff/VFPInjPropertiesLegacy.cpp:36-130
and, below those, opm-simulators' VFPHelpers.hpp (detail::findInterpData / interpolate / getFlo / getWFR / getGFR / findTHP --
external to the reference tree; restated from the published algorithm and PINNED by the reference's own known answer
tests/test_vfpproperties_legacy.cpp:375-401, see tests/test_vfp.py).

Conventions: producer rates are negative in OPM, the FLO axis of a table is positive (the lookup negates flo for producers);
the interpolant is multilinear on the table cell found per axis, with LINEAR EXTRAPOLATION outside the axes.
The device evaluates the same tables in csrc/wells.hip (`opmgpu_set_vfp_tables`).
"""
import numpy as np

FLO_OIL, FLO_LIQ, FLO_GAS = 0, 1, 2
WFR_WOR, WFR_WCT, WFR_WGR = 0, 1, 2
GFR_GOR, GFR_GLR, GFR_OGR = 0, 1, 2


def _safe_div(a, b):
    """ADB getWFR / getGFR pass through detail::zeroIfNanInf (VFPHelpersLegacy.hpp:40-48): 0/0 and x/0 give 0"""
    if b == 0.0:
        return 0.0
    r = a / b
    return r if np.isfinite(r) else 0.0


def get_flo(aqua, liquid, vapour, flo_type):
    return liquid if flo_type == FLO_OIL else (aqua + liquid if flo_type == FLO_LIQ else vapour)


def get_wfr(aqua, liquid, vapour, wfr_type):
    if wfr_type == WFR_WOR:
        return _safe_div(aqua, liquid)
    if wfr_type == WFR_WCT:
        return _safe_div(aqua, aqua + liquid)
    return _safe_div(aqua, vapour)


def get_gfr(aqua, liquid, vapour, gfr_type):
    if gfr_type == GFR_GOR:
        return _safe_div(vapour, liquid)
    if gfr_type == GFR_GLR:
        return _safe_div(vapour, liquid + aqua)
    return _safe_div(liquid, vapour)


def _d_ratio(num, den, dnum, dden):
    """derivative of zeroIfNanInf(num / den) w.r.t. (aqua, liquid, vapour); dnum / dden are 3-vectors"""
    if den == 0.0 or not np.isfinite(num / den):
        return np.zeros(3)
    return (np.asarray(dnum, float) - (num / den) * np.asarray(dden, float)) / den


def d_flo(flo_type):
    return np.array([[0.0, 1.0, 0.0], [1.0, 1.0, 0.0], [0.0, 0.0, 1.0]][flo_type])


def d_wfr(aqua, liquid, vapour, wfr_type):
    if wfr_type == WFR_WOR:
        return _d_ratio(aqua, liquid, [1, 0, 0], [0, 1, 0])
    if wfr_type == WFR_WCT:
        return _d_ratio(aqua, aqua + liquid, [1, 0, 0], [1, 1, 0])
    return _d_ratio(aqua, vapour, [1, 0, 0], [0, 0, 1])


def d_gfr(aqua, liquid, vapour, gfr_type):
    if gfr_type == GFR_GOR:
        return _d_ratio(vapour, liquid, [0, 0, 1], [0, 1, 0])
    if gfr_type == GFR_GLR:
        return _d_ratio(vapour, liquid + aqua, [0, 0, 1], [1, 1, 0])
    return _d_ratio(liquid, vapour, [0, 1, 0], [0, 0, 1])


def find_interp_data(value, axis):
    """detail::findInterpData: (i0, i1, 1/(x1 - x0), factor); a one-point axis gives (0, 0, 0, 0).  The interval is the first
    one whose upper end is >= value; below the axis the first, at or above the last point the last interval."""
    n = len(axis)
    if n == 1:
        return 0, 0, 0.0, 0.0
    if value < axis[0]:
        i0, i1 = 0, 1
    elif value >= axis[-1]:
        i0, i1 = n - 2, n - 1
    else:
        i1 = 1
        while not (axis[i1] >= value):
            i1 += 1
        i0 = i1 - 1
    start, end = axis[i0], axis[i1]
    if end > start:
        inv = 1.0 / (end - start)
        return i0, i1, inv, (value - start) * inv
    return i0, i1, 0.0, 0.0


def _interpolate(cube, interp):
    """detail::interpolate: multilinear value and its partial derivatives on the selected cell.
    cube: array of shape (2,)*k picked at the (i0, i1) pairs; interp: list of k (i0, i1, inv_dist, factor).
    Returns value, [d/d axis_0 .. d/d axis_k-1]."""
    k = len(interp)
    derivs = []
    for ax in range(k):
        d = (np.take(cube, 1, axis=ax) - np.take(cube, 0, axis=ax)) * interp[ax][2]      # same on both end points of the axis
        # reduce the remaining axes (all but `ax`) with their interpolation factors, last axis first like the reference
        rest = [interp[j] for j in range(k) if j != ax]
        for j in range(len(rest) - 1, -1, -1):
            t2 = rest[j][3]
            d = (1.0 - t2) * np.take(d, 0, axis=j) + t2 * np.take(d, 1, axis=j)
        derivs.append(float(d))
    v = cube
    for j in range(k - 1, -1, -1):
        t2 = interp[j][3]
        v = (1.0 - t2) * np.take(v, 0, axis=j) + t2 * np.take(v, 1, axis=j)
    return float(v), derivs


class VFPProdTable:
    """opm-common's VFPProdTable fields the legacy properties read; data[thp][wfr][gfr][alq][flo] (SI)."""

    is_injector = False

    def __init__(self, table_id, datum_depth, flo_type, wfr_type, gfr_type, flo_axis, thp_axis, wfr_axis, gfr_axis, alq_axis, data):
        self.id, self.datum_depth = int(table_id), float(datum_depth)
        self.flo_type, self.wfr_type, self.gfr_type = flo_type, wfr_type, gfr_type
        self.flo, self.thp, self.wfr, self.gfr, self.alq = (np.asarray(a, float) for a in (flo_axis, thp_axis, wfr_axis, gfr_axis, alq_axis))
        self.data = np.ascontiguousarray(data, float).reshape(self.thp.size, self.wfr.size, self.gfr.size, self.alq.size, self.flo.size)

    def _lookup(self, aqua, liquid, vapour, thp, alq):
        flo = get_flo(aqua, liquid, vapour, self.flo_type)
        wfr = get_wfr(aqua, liquid, vapour, self.wfr_type)
        gfr = get_gfr(aqua, liquid, vapour, self.gfr_type)
        it = [find_interp_data(thp, self.thp), find_interp_data(wfr, self.wfr), find_interp_data(gfr, self.gfr),
              find_interp_data(alq, self.alq), find_interp_data(-flo, self.flo)]            # flo is negative for producers
        cube = self.data[np.ix_(*[(i[0], i[1]) for i in it])]
        return _interpolate(cube, it)

    def bhp(self, aqua, liquid, vapour, thp, alq):
        """detail::bhp -> VFPEvaluation: value, dthp, dwfr, dgfr, dalq, dflo (derivatives w.r.t. the TABLE variables)"""
        v, (dthp, dwfr, dgfr, dalq, dflo) = self._lookup(aqua, liquid, vapour, thp, alq)
        return v, dthp, dwfr, dgfr, dalq, dflo

    def bhp_dq(self, aqua, liquid, vapour, thp, alq):
        """value and d bhp / d (aqua, liquid, vapour): VFPProdPropertiesLegacy::bhp's Jacobian (.cpp:125-148; note the MINUS on dflo)"""
        v, dthp, dwfr, dgfr, dalq, dflo = self.bhp(aqua, liquid, vapour, thp, alq)
        dq = dwfr * d_wfr(aqua, liquid, vapour, self.wfr_type) + dgfr * d_gfr(aqua, liquid, vapour, self.gfr_type) - dflo * d_flo(self.flo_type)
        return v, dq

    def thp_of(self, aqua, liquid, vapour, bhp, alq):
        """VFPProdPropertiesLegacy::thp -> detail::findTHP on the 1-D view bhp(thp_axis)"""
        arr = np.array([self._lookup(aqua, liquid, vapour, t, alq)[0] for t in self.thp])
        return find_thp(arr, self.thp, bhp)


class VFPInjTable:
    """VFPInjTable: data[thp][flo] (SI); flo is positive for injectors"""

    is_injector = True

    def __init__(self, table_id, datum_depth, flo_type, flo_axis, thp_axis, data):
        self.id, self.datum_depth, self.flo_type = int(table_id), float(datum_depth), flo_type
        self.flo, self.thp = np.asarray(flo_axis, float), np.asarray(thp_axis, float)
        self.data = np.ascontiguousarray(data, float).reshape(self.thp.size, self.flo.size)

    def _lookup(self, aqua, liquid, vapour, thp):
        flo = get_flo(aqua, liquid, vapour, self.flo_type)
        it = [find_interp_data(thp, self.thp), find_interp_data(flo, self.flo)]
        cube = self.data[np.ix_(*[(i[0], i[1]) for i in it])]
        return _interpolate(cube, it)

    def bhp(self, aqua, liquid, vapour, thp):
        v, (dthp, dflo) = self._lookup(aqua, liquid, vapour, thp)
        return v, dthp, dflo

    def bhp_dq(self, aqua, liquid, vapour, thp, alq=0.0):
        v, dthp, dflo = self.bhp(aqua, liquid, vapour, thp)
        return v, dflo * d_flo(self.flo_type)                       # PLUS for injectors (VFPInjPropertiesLegacy.cpp:118-120)

    def thp_of(self, aqua, liquid, vapour, bhp, alq=0.0):
        arr = np.array([self._lookup(aqua, liquid, vapour, t)[0] for t in self.thp])
        return find_thp(arr, self.thp, bhp)


def _find_x(x0, x1, y0, y1, y):
    return x0 + ((x1 - x0) / (y1 - y0)) * (y - y0)


def find_thp(bhp_array, thp_array, bhp):
    """detail::findTHP: invert the piecewise-linear bhp(thp); extrapolates from the end intervals; an unsorted bhp array (possible
    after extrapolation along the other axes) is searched interval by interval first."""
    n = len(thp_array)
    b, t = bhp_array, thp_array
    is_sorted = all(b[i] <= b[i + 1] for i in range(n - 1))

    def search():
        for i in range(n - 1):
            if b[i] < bhp <= b[i + 1]:
                return i
        return -1

    if is_sorted:
        if bhp <= b[0]:
            return _find_x(t[0], t[1], b[0], b[1], bhp)
        if bhp > b[n - 1]:
            return _find_x(t[n - 2], t[n - 1], b[n - 2], b[n - 1], bhp)
        i = search()
        return _find_x(t[i], t[i + 1], b[i], b[i + 1], bhp)
    i = search()
    if i >= 0:
        return _find_x(t[i], t[i + 1], b[i], b[i + 1], bhp)
    if bhp <= b[0]:
        return _find_x(t[0], t[1], b[0], b[1], bhp)
    if bhp > b[n - 1]:
        return _find_x(t[n - 2], t[n - 1], b[n - 2], b[n - 1], bhp)
    raise RuntimeError("Programmer error: Unable to find THP in THP array")


def hydrostatic_correction(well_depth_ref, vfp_ref_depth, rho, gravity, has_perforations=True):
    """wellhelpers::computeHydrostaticCorrection (opm-simulators WellHelpers.hpp, external): rho g (vfp datum - well reference depth)"""
    if not has_perforations:
        return 0.0
    return rho * gravity * (vfp_ref_depth - well_depth_ref)
