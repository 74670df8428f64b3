"""Well-coupled Newton iterations: the device model and the CPU oracle driven by the SAME host well model
(opmgpu/wells.py) must walk the same Newton path on the SPE1-like deck (BASELINE configs[0])."""
import numpy as np
import pytest

from opmgpu import capi, decks, wells as W
from opmgpu.model import GpuBlackoilModel
from test_wells_host import _setup
from util import OracleBackend

pytestmark = pytest.mark.gpu


def test_well_coupled_newton_parity(gpu_lib, oracle):
    grid, tab, st, wl = _setup()
    prm = capi.default_params(linear_solver_reduction=1e-11, linear_solver_maxiter=500)
    dt = 2 * decks.DAY
    gm = GpuBlackoilModel(grid, tab, prm, wells=wl.arrays())
    ob = OracleBackend(oracle, grid, tab, prm, wells=wl.arrays())
    ob.position = gm.ordering()[0]
    mg = W.WellCoupledModel(gm, W.StandardWellsHost(wl, grid.z, tab.surface_density[0]), W.WellState(wl, st.p))
    mo = W.WellCoupledModel(ob, W.StandardWellsHost(wl, grid.z, tab.surface_density[0]), W.WellState(wl, st.p))
    mg.prepareStep(dt, st); mo.prepareStep(dt, st)
    for step in range(2):
        it = 0
        while True:
            cg, lg = mg.nonlinearIteration(it, single_precision=False)
            co, lo = mo.nonlinearIteration(it, single_precision=False)
            assert cg == co, (step, it)
            it += 1
            a, b = gm.getState(), ob.getState()
            assert np.array_equal(a.hc, b.hc), (step, it)
            assert np.abs(a.p - b.p).max() <= 1e-6 * np.abs(b.p).max(), (step, it)
            assert np.abs(a.sat - b.sat).max() <= 1e-6, (step, it)
            assert np.allclose(mg.ws.bhp, mo.ws.bhp, rtol=1e-7) and np.allclose(mg.ws.qs, mo.ws.qs, rtol=1e-6, atol=1e-9 * np.abs(mo.ws.qs).max())
            assert np.allclose(gm.CNV, ob.CNV, rtol=1e-5, atol=1e-9) and np.allclose(mg.wh.flux_eq, mo.wh.flux_eq, rtol=1e-5, atol=1e-12)
            if cg and it >= 1:
                break
            assert it <= 12
        mg.prepareStep(dt); mo.prepareStep(dt)
    assert mg.ws.qs[0, 0] > 0 and mg.ws.qs[1, 1] < 0
    gm.close()
