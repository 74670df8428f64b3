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
    def __init__(self, deck_path, params=None, output_base=None, device=0, ats=None, vfp_tables=(), model_factory=None, well_model_factory=None,
                 restart=None):
        """model_factory(grid, tables, params) / well_model_factory(model, wells, well_state): other implementations of the model interface
        behind the same driver (the tests run the CPU oracle + host well model through it and diff the two runs' output files like the
        reference's regression tests diff flow_legacy's); default: the device model / the device well model.
        restart = (base, report): start from report step `report` of BASE.UNRST written by an earlier run of the same deck instead of the
        deck's initial state (what tests/run-restart-regressionTest.sh exercises: the restarted run must reproduce the full one within
        abs 2e-1 / rel 4e-5): reservoir state, VAPPARS / hysteresis history, well state by name; the run continues with the SCHEDULE's
        next report step."""
        self.deck = deckmod.read_deck(deck_path)
        self.tables = self.deck.tables()
        self.grid = self.deck.grid()
        self.state0 = self.deck.initial_state(self.tables)
        self.start_step, self.t0, self._restart_ws = 0, 0.0, None
        if restart is not None:
            self._load_restart(*restart)
        nx, ny, nz = self.deck.dims
        n = nx * ny * nz
        dx, dy, dz = self.deck._cell_sizes()
        kx = self.deck.array("PERMX", n, np.zeros(n))
        ky = self.deck.array("PERMY", n, kx)
        self.schedule = schedmod.Schedule(self.deck, self.grid, perm_md=(kx, ky), dz=dz.ravel(), dxdy=(dx.ravel(), dy.ravel()),
                                          ntg=self.deck.array("NTG", n, np.ones(n)))
        self.params = params or capi.default_params(cpr_use_amg=1, cpr_max_ell_iter=0, use_cpr=1)
        self.well_model_factory = well_model_factory
        if model_factory is None:
            self.model = GpuBlackoilModel(self.grid, self.tables, self.params, device=device)
            self.model.setState(self.state0)
        else:
            self.model = model_factory(self.grid, self.tables, self.params)
            self.model.prepareStep(1.0, self.state0)
        self.ats = ats or ts.AdaptiveTimeStepping(initial_timestep_days=1.0)
        if getattr(self, "_restart_dtnx", None) is not None:
            self.ats.suggested_next_timestep = self._restart_dtnx
        self.vfp_tables = vfp_tables
        self.fipnum = self.deck.fipnum()             # REGIONS FIPNUM (None: the field is one region)
        self.out = None
        if output_base:
            # EclOutput truncates BASE.UNRST / .UNSMRY: a restarted run writing over the files it was just started from would destroy the full run's
            import os
            if restart is not None and os.path.abspath(output_base) == os.path.abspath(restart[0]):
                raise ValueError("output_base %r is the restart source: choose another base name for the restarted run's files" % output_base)
            porv = np.zeros(n); porv[self.deck.active] = self.grid.pv
            tops = self.deck.array("TOPS")[:nx * ny] if self.deck.has("TOPS") and self.deck.array("TOPS").size >= nx * ny else None
            cz = (self.deck.array("COORD"), self.deck.array("ZCORN")) if self.deck.has("ZCORN") else None
            self.out = eclio.EclOutput(output_base, (nx, ny, nz), self.grid.active_index, self.schedule.start, cell_sizes=(dx, dy, dz), tops=tops, porv=porv,
                                       coord_zcorn=cz)
        self.reports = []

    def _load_restart(self, base, report):
        from .decks import BAR, State
        r = eclio.read_restart(base, report)
        sw, sg = np.asarray(r["SWAT"], float), np.asarray(r["SGAS"], float)
        sat = np.stack([sw, 1.0 - sw - sg, sg], 1)
        hc = np.where(sg > 0, np.where(sat[:, 1] > 0, capi.HC_GAS_AND_OIL, capi.HC_GAS_ONLY), capi.HC_OIL_ONLY).astype(np.int8)     # initHydroCarbonState
        if not self.tables.has_disgas:
            hc[hc == capi.HC_OIL_ONLY] = capi.HC_GAS_AND_OIL
        self.state0 = State(np.asarray(r["PRESSURE"], float) * BAR, sat, np.asarray(r["RS"], float), np.asarray(r["RV"], float), hc)
        self.start_step, self.t0 = report - 1, r["DAYS"] * DAY        # report 1 is the initial state, report k + 1 follows schedule step k - 1
        self._restart_extra = {k: np.asarray(r[k], float) for k in ("SOMAX", "HMDC_OW", "HMDC_GO") if k in r}
        self._restart_dtnx = float(r["OPMGDTNX"][0]) * DAY if "OPMGDTNX" in r else None
        if "OPMGXWEL" in r:
            x = np.asarray(r["OPMGXWEL"], float).reshape(-1, 5)
            self._restart_ws = ([str(n).strip() for n in r["OPMGWNAM"]], x, np.asarray(r["OPMGIWEL"], int))

    def run(self, max_steps=None):
        gm = self.model
        t, prev_ws, prev_names = self.t0, None, None
        if self.start_step > 0:
            ex = getattr(self, "_restart_extra", {})
            if "SOMAX" in ex and hasattr(gm, "setSatOilMax"):
                gm.setSatOilMax(ex["SOMAX"])
            if "HMDC_OW" in ex and hasattr(gm, "setHysteresis"):
                gm.setHysteresis(ex["HMDC_OW"], ex["HMDC_GO"])
        if self.out:
            self.out.report = self.start_step             # the restarted run numbers its report steps like the full one
            self.out.write_restart(t / DAY, gm.getState())
        nsteps = len(self.schedule.steps) if max_steps is None else min(max_steps, len(self.schedule.steps))
        for step in range(self.start_step, nsteps):
            dt = self.schedule.steps[step][0]
            wl = self.schedule.wells(step)
            st = gm.getState()
            ws = W.WellState(wl, st.p)
            if step == self.start_step and self._restart_ws is not None:          # the restart file's well state, by name
                names, x, cur = self._restart_ws
                for w, name in enumerate(wl.name):
                    if str(name)[:8] in names:
                        k = names.index(str(name)[:8])
                        ws.bhp[w], ws.thp[w], ws.qs[w] = x[k, 0], x[k, 1], x[k, 2:5]
                        if cur[k] < len(wl.controls[w]):
                            ws.current[w] = cur[k]
            if prev_ws is not None:                       # WellStateFullyImplicitBlackoil::init(..., prevState): same-name wells keep their state
                for w, name in enumerate(wl.name):
                    if name in prev_names:
                        k = prev_names.index(name)
                        ws.bhp[w], ws.qs[w], ws.thp[w] = prev_ws.bhp[k], prev_ws.qs[k], prev_ws.thp[k]
                        if prev_ws.current[k] < len(wl.controls[w]):
                            ws.current[w] = prev_ws.current[k]
            event = prev_names != list(wl.name)
            if step == self.start_step and self._restart_ws is not None and [str(n)[:8] for n in wl.name] == self._restart_ws[0]:
                event = False                             # the same wells as at the end of the run the restart file comes from
            if hasattr(gm, "updateSatOilMax"):
                gm.updateSatOilMax()                      # SimulatorBase_impl.hpp:190-192
                gm.updateHysteresis()
            if wl.nw > 0 and hasattr(gm, "lib"):         # SimulatorBase_impl.hpp:196: computeRESV before the step's solver is built
                from .rateconverter import SurfaceToReservoirVoidage, computeRESV
                if getattr(self, "_rate_converter", None) is None:
                    self._rate_converter = SurfaceToReservoirVoidage(gm)      # one region of all cells (:66), a member like rateConverter_
                computeRESV(self._rate_converter, wl, pvtnum=getattr(self.grid, "pvtnum", None))
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
            if hasattr(gm, "computeFluidInPlace"):      # COIP of SimulatorBase_impl.hpp:278: values[region][7] at the end of every report step
                self.reports[-1]["fip"] = gm.computeFluidInPlace(self.fipnum)
            if self.out:
                extra = {}
                if hasattr(gm, "satOilMax") and self.tables.vap1 + self.tables.vap2 > 0:
                    extra["SOMAX"] = gm.satOilMax()
                if hasattr(gm, "getHysteresis") and self.grid.imbnum is not None:
                    h = gm.getHysteresis()
                    extra["HMDC_OW"], extra["HMDC_GO"] = h[0], h[1]
                self.out.write_restart(t / DAY, gm.getState(), extra=extra, wells=wl if wl.nw > 0 else None, well_state=ws if wl.nw > 0 else None,
                                       next_step_days=self.ats.suggested_next_timestep / DAY)
                if wl.nw > 0:
                    # field totals in place and the hydrocarbon-pv weighted pressure (SimulatorBase_impl.hpp:278: COIP at every report step)
                    fip = gm.computeFluidInPlace()[0] if hasattr(gm, "computeFluidInPlace") else None
                    self.out.write_summary(t / DAY, wl, ws, new_report_step=True, fip=fip)
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
