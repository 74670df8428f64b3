"""Adaptive sub-stepping of a report step around the device-resident Newton loop.

Host mirror of `Opm::AdaptiveTimeStepping::stepImpl` (opm/simulators/timestepping/AdaptiveTimeStepping_impl.hpp:183-372):
sub-step loop, restart-on-failure contract (TooManyIterations / LinearSolverProblem / NumericalIssue / ISTLError ->
chop by `restart_factor`, restore the last converged state, at most `solver_restart_max` times in a row), growth
limits, suggested next step.  The state never leaves the device: `last_state` is `opmgpu_save_state` /
`opmgpu_restore_state`, the time error is `opmgpu_relative_change`.

The step-size controllers and the sub-step timer are opm-core code that is not in the reference tree
(`TimeStepControl.cpp`, `AdaptiveSimulatorTimer.cpp`): restated from the published algorithm, parity unpinned.
"""
import math

from .model import ISTLError, LinearSolverProblem, NumericalIssue, TooManyIterations

DAY = 86400.0


class PIDTimeStepControl:
    """PID controller on the relative change of the solution (Turek's constants kP .075, kI .175, kD .01)."""

    def __init__(self, tol=1e-1):
        self.tol = tol
        self.errors = [tol, tol, tol]

    def computeTimeStepSize(self, dt, iterations, relative_change, simulation_time_elapsed=0.0):
        e = self.errors
        e[0], e[1] = e[1], e[2]
        e[2] = relative_change()
        if not all(math.isfinite(x) for x in e):
            raise NumericalIssue("non-finite relative change in the time step control")
        if e[2] > self.tol:                       # error too large: shrink proportionally
            return dt * self.tol / e[2]
        if e[2] == 0.0 or e[1] == 0.0:            # unchanged state: the reference's floating-point division gives +inf, the caller's growth limits clip it
            return float("inf")
        kP, kI, kD = 0.075, 0.175, 0.01
        return dt * (e[1] / e[2]) ** kP * (self.tol / e[2]) ** kI * (e[0] * e[0] / e[1] / e[2]) ** kD


class PIDAndIterationCountTimeStepControl(PIDTimeStepControl):
    """PID estimate, cut further when more than `target_iterations` (linear or Newton) iterations were needed."""

    def __init__(self, target_iterations=30, tol=1e-1):
        super().__init__(tol)
        self.target_iterations = target_iterations

    def computeTimeStepSize(self, dt, iterations, relative_change, simulation_time_elapsed=0.0):
        est = super().computeTimeStepSize(dt, iterations, relative_change, simulation_time_elapsed)
        if iterations > self.target_iterations:
            est *= float(self.target_iterations) / float(iterations)
        return est


class SimpleIterationCountTimeStepControl:
    def __init__(self, target_iterations=30, decayrate=0.75, growthrate=1.25):
        self.target_iterations, self.decayrate, self.growthrate = target_iterations, decayrate, growthrate

    def computeTimeStepSize(self, dt, iterations, relative_change, simulation_time_elapsed=0.0):
        est = dt
        if iterations > self.target_iterations:
            est *= self.decayrate
        elif iterations < self.target_iterations - 1:
            est *= self.growthrate
        return est


class HardcodedTimeStepControl:
    """Sub-step lengths looked up by elapsed simulation time from a given list of (time, dt) [s]."""

    def __init__(self, times_and_steps):
        self.table = sorted(times_and_steps)

    def computeTimeStepSize(self, dt, iterations, relative_change, simulation_time_elapsed=0.0):
        for t, step in self.table:
            if t > simulation_time_elapsed:
                return step
        return self.table[-1][1] if self.table else dt


class AdaptiveSimulatorTimer:
    """Sub-step timer of one report step: clips the estimate to max_time_step and to the remaining time, and avoids a
    tiny last sub-step by halving the remainder."""

    def __init__(self, start_time, total_step, last_step_taken, max_time_step=float("inf")):
        self.start_time, self.total_time = start_time, start_time + total_step
        self.current_time = start_time
        self.max_time_step = max_time_step
        self.dt = 0.0
        self.current_step = 0
        self.steps = []
        self.last_step_failed = False
        self.provideTimeStepEstimate(last_step_taken)

    def provideTimeStepEstimate(self, dt_estimate):
        remaining = self.total_time - self.current_time
        self.dt = min(dt_estimate, self.max_time_step)
        if remaining > 0:
            if 1.05 * self.dt > remaining:
                self.dt = remaining
                if self.dt > self.max_time_step:
                    self.dt = 0.5 * remaining
                return
            if 1.5 * self.dt > remaining:
                self.dt = 0.5 * remaining

    def advance(self):
        self.current_time += self.dt
        self.current_step += 1
        self.steps.append(self.dt)

    def currentStepLength(self):
        return self.dt

    def simulationTimeElapsed(self):
        return self.current_time

    def done(self):
        return self.current_time >= self.total_time or abs(self.total_time - self.current_time) <= 1e-9 * max(1.0, abs(self.total_time))


class AdaptiveTimeStepping:
    """AdaptiveTimeStepping (AdaptiveTimeStepping_impl.hpp:75-372) with the parameter-file defaults (:101-121)."""

    def __init__(self, control="pid", restart_factor=0.33, growth_factor=2.0, max_growth=3.0, max_time_step_days=365.0,
                 solver_restart_max=10, initial_timestep_days=1.0, full_timestep_initially=False, timestep_after_event_days=-1.0,
                 tol=1e-1, target_iterations=None, decayrate=0.75, growthrate=1.25, hardcoded=None, min_time_step_fraction=1e-12):
        self.restart_factor, self.growth_factor, self.max_growth = restart_factor, growth_factor, max_growth
        assert growth_factor >= 1.0
        self.max_time_step = max_time_step_days * DAY
        self.solver_restart_max = solver_restart_max
        self.suggested_next_timestep = initial_timestep_days * DAY
        self.full_timestep_initially = full_timestep_initially
        self.timestep_after_event = timestep_after_event_days * DAY
        self.use_newton_iteration = False
        if control == "pid":
            self.control = PIDTimeStepControl(tol)
        elif control == "pid+iteration":
            self.control = PIDAndIterationCountTimeStepControl(target_iterations or 30, tol)
        elif control == "pid+newtoniteration":
            self.control = PIDAndIterationCountTimeStepControl(target_iterations or 8, tol)
            self.use_newton_iteration = True
        elif control == "iterationcount":
            self.control = SimpleIterationCountTimeStepControl(target_iterations or 30, decayrate, growthrate)
        elif control == "hardcoded":
            self.control = HardcodedTimeStepControl(hardcoded or [])
        else:
            raise RuntimeError("Unsupported time step control selected " + str(control))
        self.failed_substeps = 0
        # NOT in the reference: a failed sub-step shorter than this fraction of the report step ends the report step with NumericalIssue.
        # The reference's loop resets its restart counter after every converged sub-step (:285-289), so a state in which steps of 1e-17 d
        # converge and steps of 2e-17 d do not (seen on a generated deck whose well controls flip-flop: profiles/r04_ba_*) never reaches
        # solver_restart_max and never ends.  0 switches the guard off.
        self.min_time_step_fraction = min_time_step_fraction

    def step(self, start_time, timestep, solver, model, event=False, well_state=None):
        """One report step of length `timestep` [s] from the model's resident state.  `solver.step(model)` runs one
        sub-step (NonlinearSolver.step); `model` offers prepareStep(dt), saveState(), restoreState(), relativeChange().
        Returns a report dict; raises NumericalIssue after solver_restart_max consecutive failed sub-steps."""
        if self.suggested_next_timestep < 0:
            self.suggested_next_timestep = self.restart_factor * timestep
        if self.full_timestep_initially:
            self.suggested_next_timestep = timestep
        if event and self.timestep_after_event > 0:
            self.suggested_next_timestep = self.timestep_after_event
        timer = AdaptiveSimulatorTimer(start_time, timestep, self.suggested_next_timestep, self.max_time_step)
        model.saveState()                                   # last_state (:211)
        last_well_state = None if well_state is None else well_state.copy()
        report = {"newton_iterations": 0, "linear_iterations": 0, "substeps": [], "failed": []}
        restarts = 0
        while not timer.done():
            dt = timer.currentStepLength()
            converged, cause = False, ""
            newton = linear = 0
            try:
                model.prepareStep(dt)
                newton, linear = solver.step(model)
                converged = True
            except TooManyIterations:
                cause = "Solver convergence failure - Iteration limit reached"
            except LinearSolverProblem:
                cause = "Linear solver convergence failure"
            except NumericalIssue:
                cause = "Solver convergence failure - Numerical problem encountered"
            except ISTLError:
                cause = "ISTL error"
            if converged:
                report["newton_iterations"] += newton; report["linear_iterations"] += linear
                timer.advance()
                iterations = newton if self.use_newton_iteration else linear
                est = self.control.computeTimeStepSize(dt, iterations, model.relativeChange, timer.simulationTimeElapsed())
                est = min(est, self.max_growth * dt)
                if restarts > 0:                            # be careful after convergence problems (:285-289)
                    est = min(self.growth_factor * dt, est)
                    restarts = 0
                report["substeps"].append(dt)
                timer.provideTimeStepEstimate(est)
                model.saveState()
                if well_state is not None:
                    last_well_state = well_state.copy()
                timer.last_step_failed = False
            else:
                timer.last_step_failed = True
                self.failed_substeps += 1
                report["failed"].append((dt, cause))
                if self.min_time_step_fraction > 0 and dt < self.min_time_step_fraction * timestep:
                    err = NumericalIssue("Solver failed to converge with a time step of %.3g s (%.1e of the report step): giving up." % (dt, dt / timestep))
                    err.report = report
                    raise err
                if restarts >= self.solver_restart_max:
                    err = NumericalIssue("Solver failed to converge after cutting timestep %d times." % restarts)
                    err.report = report          # what was tried: sub-steps taken, (dt, cause) of every failure
                    raise err
                timer.provideTimeStepEstimate(self.restart_factor * dt)
                model.restoreState()                        # state = last_state (:346)
                if well_state is not None:
                    well_state.assign(last_well_state)
                restarts += 1
        self.suggested_next_timestep = timer.currentStepLength()
        if not math.isfinite(self.suggested_next_timestep):
            self.suggested_next_timestep = timestep
        report["converged"] = True
        return report
