"""Report-step driver on top of the device Newton path: what SimulatorBase::run does around `flow_legacy`'s hot path
(opm/autodiff/SimulatorBase_impl.hpp:90-324), reduced to what an external diff needs:

    deck -> grid / tables / initial state -> for every report step of the SCHEDULE: wells of the step (WellsManager), well state
    carried over by name, updateSatOilMax + updateSatHyst (:190-192), AdaptiveTimeStepping.step with the NonlinearSolver (:236-253),
    restart + summary output (:300-306).

The physics runs in libopmgpu.so (state resident on the device across sub-steps); this module is host plumbing.  Output goes to
ECLIPSE binary files (opmgpu/eclio.py) that `compareECL` can diff against a `flow_legacy` run of the same deck.
"""
import numpy as np

from . import capi, deck as deckmod, eclio, schedule as schedmod, timestepping as ts, wells as W
from .decks import DAY
from .model import GpuBlackoilModel, NonlinearSolver


class Simulator:
    def __init__(self, deck_path, params=None, output_base=None, device=0, ats=None, vfp_tables=(), model_factory=None, well_model_factory=None):
        """model_factory(grid, tables, params) / well_model_factory(model, wells, well_state): other implementations of the model interface
        behind the same driver (the tests run the CPU oracle + host well model through it and diff the two runs' output files like the
        reference's regression tests diff flow_legacy's); default: the device model / the device well model."""
        self.deck = deckmod.read_deck(deck_path)
        self.tables = self.deck.tables()
        self.grid = self.deck.grid()
        self.state0 = self.deck.initial_state(self.tables)
        nx, ny, nz = self.deck.dims
        n = nx * ny * nz
        dx, dy, dz = self.deck._cell_sizes()
        kx = self.deck.array("PERMX", n, np.zeros(n))
        ky = self.deck.array("PERMY", n, kx)
        self.schedule = schedmod.Schedule(self.deck, self.grid, perm_md=(kx, ky), dz=dz.ravel(), dxdy=(dx.ravel(), dy.ravel()),
                                          ntg=self.deck.array("NTG", n, np.ones(n)))
        self.params = params or capi.default_params(use_cpr=1)
        self.well_model_factory = well_model_factory
        if model_factory is None:
            self.model = GpuBlackoilModel(self.grid, self.tables, self.params, device=device)
            self.model.setState(self.state0)
        else:
            self.model = model_factory(self.grid, self.tables, self.params)
            self.model.prepareStep(1.0, self.state0)
        self.ats = ats or ts.AdaptiveTimeStepping(initial_timestep_days=1.0)
        self.vfp_tables = vfp_tables
        self.out = None
        if output_base:
            porv = np.zeros(n); porv[self.deck.active] = self.grid.pv
            tops = self.deck.array("TOPS")[:nx * ny] if self.deck.has("TOPS") and self.deck.array("TOPS").size >= nx * ny else None
            self.out = eclio.EclOutput(output_base, (nx, ny, nz), self.grid.active_index, self.schedule.start, cell_sizes=(dx, dy, dz), tops=tops, porv=porv)
        self.reports = []

    def run(self, max_steps=None):
        gm = self.model
        t, prev_ws, prev_names = 0.0, None, None
        if self.out:
            self.out.write_restart(0.0, gm.getState())
        nsteps = len(self.schedule.steps) if max_steps is None else min(max_steps, len(self.schedule.steps))
        for step in range(nsteps):
            dt = self.schedule.steps[step][0]
            wl = self.schedule.wells(step)
            st = gm.getState()
            ws = W.WellState(wl, st.p)
            if prev_ws is not None:                       # WellStateFullyImplicitBlackoil::init(..., prevState): same-name wells keep their state
                for w, name in enumerate(wl.name):
                    if name in prev_names:
                        k = prev_names.index(name)
                        ws.bhp[w], ws.qs[w], ws.thp[w] = prev_ws.bhp[k], prev_ws.qs[k], prev_ws.thp[k]
                        if prev_ws.current[k] < len(wl.controls[w]):
                            ws.current[w] = prev_ws.current[k]
            event = prev_names != list(wl.name)
            if hasattr(gm, "updateSatOilMax"):
                gm.updateSatOilMax()                      # SimulatorBase_impl.hpp:190-192
                gm.updateHysteresis()
            if wl.nw == 0:
                model = gm
            elif self.well_model_factory is not None:
                model = self.well_model_factory(gm, wl, ws)
            else:
                model = W.DeviceWellModel(gm, wl, ws, vfp_tables=self.vfp_tables)
            host_ws = ws if (wl.nw > 0 and not hasattr(model, "pull_well_state")) else None        # a host well model keeps the well state here
            rep = self.ats.step(t, dt, _Solver(), model, event=event, well_state=host_ws)
            t += dt
            if wl.nw > 0 and hasattr(model, "pull_well_state"):
                ws = model.pull_well_state()
            self.reports.append({"step": step, "days": t / DAY, "substeps": len(rep["substeps"]), "newton": rep["newton_iterations"],
                                 "linear": rep["linear_iterations"], "failed": len(rep["failed"])})
            if self.out:
                self.out.write_restart(t / DAY, gm.getState())
                if wl.nw > 0:
                    self.out.write_summary(t / DAY, wl, ws, new_report_step=True)
            prev_ws, prev_names = (ws.copy(), list(wl.name)) if wl.nw > 0 else (None, None)
        return self.reports

    def close(self):
        if hasattr(self.model, "close"):
            self.model.close()


class _Solver:
    """NonlinearSolver::step with the reference defaults; AdaptiveTimeStepping expects solver.step(model) -> (newton, linear)"""

    def __init__(self):
        self.ns = NonlinearSolver()

    def step(self, model):
        it, lin_total = 0, 0
        while True:
            converged, lin = model.nonlinearIteration(it, nonlinear_solver=self.ns)
            lin_total += lin
            it += 1
            if not ((not converged and it <= self.ns.max_iter) or it <= self.ns.min_iter):
                break
        if not converged:
            from .model import TooManyIterations
            raise TooManyIterations("Solver convergence failure - Failed to complete a time step within %d iterations." % self.ns.max_iter)
        return it, lin_total
