"""Host logic of the adaptive sub-stepping (opmgpu/timestepping.py) on a fake model: controllers, sub-step timer,
restart-on-failure contract of AdaptiveTimeStepping::stepImpl (AdaptiveTimeStepping_impl.hpp:183-372).  No GPU."""
import math

import pytest

from opmgpu import timestepping as ts
from opmgpu.model import NumericalIssue, TooManyIterations

DAY = ts.DAY


class FakeModel:
    """Scalar 'state' x that relaxes towards 1; a sub-step longer than dt_fail does not converge."""

    def __init__(self, dt_fail=float("inf")):
        self.x, self.saved, self.dt, self.dt_fail = 0.0, None, None, dt_fail
        self.log = []

    def prepareStep(self, dt):
        self.dt = dt

    def saveState(self):
        self.saved = self.x

    def restoreState(self):
        self.x = self.saved
        self.log.append("restore")

    def relativeChange(self):
        return (self.x - self.saved) ** 2 / max(self.x ** 2, 1e-300)


class FakeSolver:
    def step(self, model):
        if model.dt > model.dt_fail:
            model.x = float("nan")          # a failed solve leaves garbage behind
            raise TooManyIterations("too many")
        model.x += (1.0 - model.x) * (1.0 - math.exp(-model.dt / (30 * DAY)))
        return 3, 12


def test_pid_controller_arithmetic():
    c = ts.PIDTimeStepControl(tol=0.1)
    assert c.computeTimeStepSize(10.0, 5, lambda: 0.4) == pytest.approx(10.0 * 0.1 / 0.4)           # too large an error: proportional cut
    e0, e1, e2 = 0.1, 0.4, 0.05
    exp = 10.0 * (e1 / e2) ** 0.075 * (0.1 / e2) ** 0.175 * (e0 * e0 / e1 / e2) ** 0.01
    assert c.computeTimeStepSize(10.0, 5, lambda: 0.05) == pytest.approx(exp)
    c2 = ts.PIDAndIterationCountTimeStepControl(target_iterations=8, tol=0.1)
    base = ts.PIDTimeStepControl(tol=0.1).computeTimeStepSize(10.0, 16, lambda: 0.05)
    assert c2.computeTimeStepSize(10.0, 16, lambda: 0.05) == pytest.approx(base * 8 / 16)
    c3 = ts.SimpleIterationCountTimeStepControl(target_iterations=10, decayrate=0.75, growthrate=1.25)
    assert c3.computeTimeStepSize(4.0, 11, None) == 3.0 and c3.computeTimeStepSize(4.0, 8, None) == 5.0 and c3.computeTimeStepSize(4.0, 9, None) == 4.0


def test_substep_timer_clipping():
    t = ts.AdaptiveSimulatorTimer(0.0, 10.0, 4.0, max_time_step=100.0)
    assert t.currentStepLength() == 4.0
    t.advance(); t.provideTimeStepEstimate(4.1)             # remaining 6: 1.5*4.1 > 6 -> two halves instead of 4.1 + 1.9
    assert t.currentStepLength() == 3.0
    t.advance(); t.provideTimeStepEstimate(2.9)             # remaining 3: 1.05*2.9 > 3 -> take it all
    assert t.currentStepLength() == 3.0
    t.advance()
    assert t.done() and t.steps == [4.0, 3.0, 3.0]
    t2 = ts.AdaptiveSimulatorTimer(0.0, 10.0, 50.0, max_time_step=6.0)   # estimate clipped to the max step
    assert t2.currentStepLength() == 6.0
    t3 = ts.AdaptiveSimulatorTimer(0.0, 6.2, 50.0, max_time_step=6.0)    # remainder just above the max step -> two halves
    assert t3.currentStepLength() == pytest.approx(3.1)


def test_report_step_growth_and_suggestion():
    m, a = FakeModel(), ts.AdaptiveTimeStepping(initial_timestep_days=1.0)
    rep = a.step(0.0, 30 * DAY, FakeSolver(), m)
    assert rep["converged"] and sum(rep["substeps"]) == pytest.approx(30 * DAY)
    assert rep["substeps"][0] == DAY
    for prev, cur in zip(rep["substeps"], rep["substeps"][1:]):
        assert cur <= 3.0 * prev * (1 + 1e-12)               # max_growth
    assert rep["newton_iterations"] == 3 * len(rep["substeps"]) and not rep["failed"]
    assert a.suggested_next_timestep > DAY                    # carried to the next report step
    rep2 = a.step(30 * DAY, 30 * DAY, FakeSolver(), m)
    assert rep2["substeps"][0] == pytest.approx(min(a.max_time_step, rep2["substeps"][0]))
    assert len(rep2["substeps"]) <= len(rep["substeps"])


def test_restart_on_failure_restores_state_and_limits_growth():
    m = FakeModel(dt_fail=2.5 * DAY)
    a = ts.AdaptiveTimeStepping(initial_timestep_days=6.0)
    rep = a.step(0.0, 12 * DAY, FakeSolver(), m)
    assert rep["converged"] and math.isfinite(m.x)
    assert [round(d / DAY, 6) for d, _ in rep["failed"][:1]] == [6.0]
    assert rep["failed"][0][1].startswith("Solver convergence failure - Iteration limit")
    assert "restore" in m.log
    first_ok = rep["substeps"][0]
    assert first_ok == pytest.approx(6.0 * DAY * 0.33)       # chopped by restart_factor
    assert all(d <= 2.5 * DAY * (1 + 1e-12) for d in rep["substeps"])
    assert sum(rep["substeps"]) == pytest.approx(12 * DAY)


def test_gives_up_after_solver_restart_max():
    m = FakeModel(dt_fail=0.0)                                # nothing ever converges
    a = ts.AdaptiveTimeStepping(initial_timestep_days=1.0, solver_restart_max=4)
    with pytest.raises(NumericalIssue, match="cutting timestep 4 times"):
        a.step(0.0, 10 * DAY, FakeSolver(), m)
    assert a.failed_substeps == 5


def test_livelock_guard_ends_a_report_step_whose_sub_steps_shrink_without_end():
    """Not in the reference (its loop resets the restart counter after every converged sub-step, AdaptiveTimeStepping_impl.hpp:285-289): a
    model whose sub-steps converge only every other time -- found on a generated deck with flip-flopping well controls, where sub-steps
    of 1e-17 d alternated between converging and failing for ever -- ends in NumericalIssue once a failed sub-step is shorter than
    min_time_step_fraction of the report step; the exception carries the report.  With the guard off the reference's behaviour is kept
    (checked with a bounded number of calls)."""
    class Alternating:
        def __init__(self, limit):
            self.calls, self.limit = 0, limit

        def step(self, model):
            self.calls += 1
            if self.calls > self.limit:
                raise KeyboardInterrupt          # the test's own way out of the livelock
            if self.calls % 2 == 0 or model.dt > 1e-3 * DAY:
                raise TooManyIterations("too many")
            return 2, 1                                # converges without changing anything
    m = FakeModel()
    with pytest.raises(NumericalIssue, match="giving up") as e:
        ts.AdaptiveTimeStepping(initial_timestep_days=1.0).step(0.0, 30 * DAY, Alternating(10 ** 6), m)
    rep = e.value.report
    assert len(rep["failed"]) > 10 and min(d for d, _ in rep["failed"]) < 1e-12 * 30 * DAY * 3
    m = FakeModel()
    with pytest.raises(KeyboardInterrupt):
        ts.AdaptiveTimeStepping(initial_timestep_days=1.0, min_time_step_fraction=0.0).step(0.0, 30 * DAY, Alternating(2000), m)


def test_defaults_are_the_references():
    """The host mirrors' defaults against the lines of the reference that set them: NonlinearSolver::SolverParameters::reset
    (NonlinearSolver_impl.hpp:183-188) and AdaptiveTimeStepping's parameter defaults (AdaptiveTimeStepping_impl.hpp:101-112, :123-147)."""
    from opmgpu import capi
    from opmgpu.model import NonlinearSolver
    ns = NonlinearSolver()
    assert (ns.relax_type, ns.relax_max, ns.relax_increment, ns.relax_rel_tol, ns.max_iter, ns.min_iter) == (capi.RELAX_DAMPEN, 0.5, 0.1, 0.2, 10, 1)
    a = ts.AdaptiveTimeStepping()
    assert (a.restart_factor, a.growth_factor, a.max_growth, a.solver_restart_max) == (0.33, 2.0, 3.0, 10)
    assert a.max_time_step == 365.0 * ts.DAY and a.suggested_next_timestep == 1.0 * ts.DAY
    assert a.full_timestep_initially is False and a.timestep_after_event == -1.0 * ts.DAY
    assert isinstance(a.control, ts.PIDTimeStepControl) and a.control.tol == 1e-1 and not a.use_newton_iteration
    assert ts.AdaptiveTimeStepping(control="pid+iteration").control.target_iterations == 30
    b = ts.AdaptiveTimeStepping(control="pid+newtoniteration")
    assert b.control.target_iterations == 8 and b.use_newton_iteration
    c = ts.AdaptiveTimeStepping(control="iterationcount").control
    assert (c.target_iterations, c.decayrate, c.growthrate) == (30, 0.75, 1.25)
