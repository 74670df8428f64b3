"""CPU restatement of RateConverter::SurfaceToReservoirVoidage (TEST INFRASTRUCTURE -- see oracle/oracle.h; only tests/ may import this).

Follows RateConverterLegacy.hpp: Attributes() :684-692, calcAverages :718-768, calcCoeff :495-548.  Pinned by the reference's own
tests/test_rateconverter.cpp (ThreePhase: fluid.data, a zero-initialised BlackoilState, all three coefficients 1 within 1e-6 %)."""
import numpy as np

from . import oracle as orc


class SurfaceToReservoirVoidage:
    def __init__(self, tables, region):
        """region: one id per cell (RegionMapping); SimulatorBase passes all zeros (SimulatorBase_impl.hpp:66)"""
        self.t = tables
        self.region = np.asarray(region)
        self.attr = {int(r): {"pressure": 0.0, "rs": 0.0, "rv": 0.0} for r in np.unique(self.region)}

    def defineState(self, pressure, rs, rv):
        # calcAverages<false>: p (and T) are cleared, rs and rv are NOT (:733-737): they start from the previous call's averages
        for r, a in self.attr.items():
            p_acc, n = 0.0, 0
            rs_acc, rv_acc = a["rs"], a["rv"]
            for c in np.flatnonzero(self.region == r):          # rmap_.cells(reg), ascending cell order
                p_acc += pressure[c]
                rs_acc += rs[c]
                rv_acc += rv[c]
                n += 1
            a["pressure"], a["rs"], a["rv"] = p_acc / n, rs_acc / n, rv_acc / n
        return self

    def calcCoeff(self, r, pvtRegionIdx=0):
        a = self.attr[int(r)]
        p, Rs, Rv = a["pressure"], a["rs"], a["rv"]
        reg = np.asarray([pvtRegionIdx], np.int32)
        coeff = np.zeros(3)
        bw = orc.pvt(self.t, "bWat", [p], pvtnum=reg)[0, 0]
        coeff[0] = 1.0 / bw                                    # q[w]_r = q[w]_s / bw
        detR = 1.0 - (Rs * Rv)
        # inverseFormationVolumeFactor(region, T, p, Rs): the table at the GIVEN ratio, whatever the saturated curve says
        bo = orc.pvt(self.t, "bOil", [p], r=[Rs], saturated=[0], pvtnum=reg)[0, 0]
        den = bo * detR
        coeff[1] += 1.0 / den                                  # q[o]_r = 1/(bo (1 - rs rv)) (q[o]_s - rv q[g]_s)
        coeff[2] -= Rv / den
        bg = orc.pvt(self.t, "bGas", [p], r=[Rv], saturated=[0], pvtnum=reg)[0, 0]
        den = bg * detR
        coeff[2] += 1.0 / den                                  # q[g]_r = 1/(bg (1 - rs rv)) (q[g]_s - rs q[o]_s)
        coeff[1] -= Rs / den
        return coeff
