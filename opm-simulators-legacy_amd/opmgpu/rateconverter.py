"""Surface-rate -> reservoir-voidage conversion for RESV well controls: the host-side mirror of
`RateConverter::SurfaceToReservoirVoidage` (RateConverterLegacy.hpp:407-770) and of `SimulatorBase::computeRESV`
(SimulatorBase_impl.hpp:476-553, the prediction-mode half: WCONHIST is not read by opmgpu/schedule.py).

The state stays on the device: `defineState` asks the library for the regions' sums (opmgpu_region_state_sums, collective in decomposed
runs), `calcCoeff` evaluates the PVT tables on the device at the region's average state (opmgpu_voidage_coefficients)."""
import numpy as np

from . import capi
from .wells import RESERVOIR_RATE


class SurfaceToReservoirVoidage:
    """region: one id per cell (any integers, e.g. FIPNUM) or None = one region 0 holding every cell -- what SimulatorBase constructs
    (`std::vector<int>(numCells, 0)`, SimulatorBase_impl.hpp:66) and then asks for (`fipreg = 0; // Hack.  Ignore FIP regions.`, :548)."""

    def __init__(self, model, region=None, n_ranks=1):
        self.m = model
        self.n_ranks = int(n_ranks)
        if region is None:
            self.ids, self.index = np.zeros(1, np.int64), None
        else:
            region = np.asarray(region)
            self.ids, idx = np.unique(region, return_inverse=True)
            self.index = capi.i32(idx)
        # Attributes(): pressure, temperature, rs, rv, pv all start at zero (:684-692)
        self.attr = {int(r): {"pressure": 0.0, "rs": 0.0, "rv": 0.0} for r in self.ids}

    def defineState(self):
        """calcAverages (:718-768): plain means over the region's cells.  As the reference is written, `p` and `T` are cleared before the
        loop but `rs` and `rv` are not (:733-737) -- they start from the PREVIOUS call's averages (in a parallel run on every rank, so the
        summed numerator holds the old value once per rank); restated as found."""
        m, n = self.m, len(self.ids)
        sums = np.zeros((n, 4))
        m._chk(m.lib.opmgpu_region_state_sums(m.ctx, capi.iptr(self.index) if self.index is not None else None, n, capi.dptr(sums)))
        for k, r in enumerate(self.ids):
            a = self.attr[int(r)]
            cnt = sums[k, 3]
            a["pressure"] = sums[k, 0] / cnt
            a["rs"] = (self.n_ranks * a["rs"] + sums[k, 1]) / cnt
            a["rv"] = (self.n_ranks * a["rv"] + sums[k, 2]) / cnt
        return self

    def calcCoeff(self, r, pvtRegionIdx=0):
        """coeff[water, oil, gas] with q_rT = sum_p coeff[p] q_s[p] (:495-548)"""
        a = self.attr[int(r)]
        m = self.m
        out = np.zeros(3)
        m._chk(m.lib.opmgpu_voidage_coefficients(m.ctx, 1, capi.dptr(capi.f64([a["pressure"]])), capi.dptr(capi.f64([a["rs"]])),
                                                 capi.dptr(capi.f64([a["rv"]])), capi.iptr(capi.i32([pvtRegionIdx])), capi.dptr(out)))
        return out


def resv_control(ctrls):
    """SimFIBODetails::resv_control (SimulatorBase_impl.hpp:343-357): index of the well's first RESERVOIR_RATE control, -1 if none"""
    for i, c in enumerate(ctrls):
        if c[0] == RESERVOIR_RATE:
            return i
    return -1


def computeRESV(rate_converter, wells, pvtnum=None, device_wells=None, global_number_resv_wells=None):
    """SimulatorBase::computeRESV, prediction mode (SimulatorBase_impl.hpp:476-553): when any well has a RESERVOIR_RATE control, take the
    field's average state and give that control the conversion coefficients of the PVT region of the well's top perforation as its
    `distr` (well_controls_iset_distr).  device_wells: a DeviceWellModel whose control arrays are refreshed on the device.
    Decomposed runs: pass the number of RESV wells summed over the ranks (:489-505) -- the averages are collective, so a rank without RESV
    wells of its own still takes part when another rank has one.  Returns the indices of this rank's RESV wells."""
    resv_wells = [w for w in range(wells.nw) if resv_control(wells.controls[w]) >= 0]
    if (len(resv_wells) if global_number_resv_wells is None else global_number_resv_wells):
        rate_converter.defineState()
    if not resv_wells:
        return resv_wells
    for w in resv_wells:
        rctrl = resv_control(wells.controls[w])
        top = wells.cells[wells.connpos[w]]
        pvtreg = 0 if pvtnum is None else int(pvtnum[top])
        distr = rate_converter.calcCoeff(0, pvtreg)              # "fipreg = 0; // Hack.  Ignore FIP regions."
        c = wells.controls[w][rctrl]
        wells.controls[w][rctrl] = (c[0], c[1], np.asarray(distr, float)) + tuple(c[3:])
    if device_wells is not None:
        _, _, ctgt, cdis, _, _ = wells.control_arrays()
        m = device_wells.m
        m._chk(m.lib.opmgpu_well_controls_set_targets(m.ctx, capi.dptr(capi.f64(ctgt)), capi.dptr(capi.f64(cdis))))
    return resv_wells
