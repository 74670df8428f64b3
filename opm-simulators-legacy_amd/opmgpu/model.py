"""Host-side mirror of the reference's model / solver interfaces over the C ABI.

`GpuBlackoilModel` carries the method names and call order of Opm::BlackoilModelBase
(opm/autodiff/BlackoilModelBase_impl.hpp:222-326: prepareStep, nonlinearIteration = assemble ->
getConvergence -> solveJacobianSystem -> updateState) and `GpuNewtonIteration` those of
NewtonIterationBlackoilInterface (NewtonIterationBlackoilInterface.hpp:31-52).  Error behaviour
follows the reference: status codes of the C ABI become the exceptions flow_legacy's time stepper
catches (AdaptiveTimeStepping_impl.hpp:244-281).  All compute happens in libopmgpu.so.
"""
import ctypes as C

import numpy as np

from . import capi


class NumericalIssue(RuntimeError):
    """Opm::NumericalIssue"""


class LinearSolverProblem(RuntimeError):
    """Opm::LinearSolverProblem"""


class ISTLError(RuntimeError):
    """Dune::ISTLError / Dune::MatrixBlockError"""


class TooManyIterations(RuntimeError):
    """Opm::TooManyIterations"""


def _raise(lib, ctx, st):
    if st == capi.OK:
        return
    msg = lib.opmgpu_last_error(ctx)
    msg = msg.decode() if msg else ""
    if st == capi.ENUMERICAL:
        raise NumericalIssue(msg)
    if st == capi.ELINSOLVE:
        raise LinearSolverProblem(msg)
    if st in (capi.EBREAKDOWN, capi.ESINGULAR):
        raise ISTLError(msg)
    if st == capi.EINVAL:
        raise ValueError("opmgpu: invalid argument: " + msg)
    raise RuntimeError("opmgpu status %d: %s" % (st, msg))


class GpuNewtonIteration:
    """B1: computeNewtonIncrement on a BCRS<3x3> system handed over as BSR."""

    def __init__(self, params=None, device=0):
        self.lib = capi.load()
        self.params = params or capi.default_params()
        self.ctx = C.c_void_p()
        st = self.lib.opmgpu_create_solver(C.byref(self.ctx), device, C.byref(self.params))
        if st != capi.OK:
            raise RuntimeError("opmgpu_create_solver failed with status %d (no GPU? there is no CPU fallback)" % st)
        self._iterations = 0
        self.reduction = 0.0

    def close(self):
        if self.ctx:
            self.lib.opmgpu_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def iterations(self):
        return self._iterations

    def parallelInformation(self):
        return None         # empty boost::any <=> serial

    def computeNewtonIncrement(self, rowptr, col, val9, rhs3, single_precision):
        nb = self.nb = rowptr.size - 1
        x = np.zeros(3 * nb)
        it, red = C.c_int(0), C.c_double(0)
        st = self.lib.opmgpu_solve_bsr(self.ctx, nb, capi.iptr(rowptr), capi.iptr(col), capi.dptr(capi.f64(val9)),
                                       capi.dptr(capi.f64(rhs3)), int(single_precision), capi.dptr(x), C.byref(it), C.byref(red))
        self._iterations, self.reduction = it.value, red.value
        _raise(self.lib, self.ctx, st)
        return x

    # kernel-level helpers (parity tests / bench)
    def load(self, rowptr, col, val9, single_precision=False):
        nb = rowptr.size - 1
        _raise(self.lib, self.ctx, self.lib.opmgpu_load_bsr(self.ctx, nb, capi.iptr(rowptr), capi.iptr(col), capi.dptr(capi.f64(val9)), int(single_precision)))
        self.nb = nb

    def spmv(self, x3):
        y = np.zeros(3 * self.nb)
        _raise(self.lib, self.ctx, self.lib.opmgpu_spmv(self.ctx, capi.dptr(capi.f64(x3)), capi.dptr(y)))
        return y

    def ilu0_factor(self):
        _raise(self.lib, self.ctx, self.lib.opmgpu_ilu0_factor(self.ctx))

    def ilu0_apply(self, d3):
        v = np.zeros(3 * self.nb)
        _raise(self.lib, self.ctx, self.lib.opmgpu_ilu0_apply(self.ctx, capi.dptr(capi.f64(d3)), capi.dptr(v)))
        return v

    def ilu0_get(self, nnzb):
        lu = np.zeros((nnzb, 9))
        _raise(self.lib, self.ctx, self.lib.opmgpu_ilu0_get(self.ctx, capi.dptr(lu)))
        return lu

    def ordering(self):
        pos, lev = np.zeros(self.nb, np.int32), np.zeros(self.nb, np.int32)
        nl = C.c_int32(0)
        _raise(self.lib, self.ctx, self.lib.opmgpu_get_ordering(self.ctx, capi.iptr(pos), capi.iptr(lev), C.byref(nl)))
        return pos, lev, nl.value

    def time_kernel(self, kernel, reps=20):
        ms = C.c_double(0)
        _raise(self.lib, self.ctx, self.lib.opmgpu_time_kernel(self.ctx, kernel, reps, C.byref(ms)))
        return ms.value


class GpuBlackoilModel:
    """B2: the BlackoilModel hooks, state resident on the device."""

    def __init__(self, grid, tables, params=None, device=0, wells=None):
        self.lib = capi.load()
        self.grid, self.tables = grid, tables
        self.params = params or capi.default_params()
        self.ctx = C.c_void_p()
        st = self.lib.opmgpu_create(C.byref(self.ctx), device, C.byref(grid.struct()), C.byref(tables.struct()), C.byref(self.params))
        if st != capi.OK:
            raise RuntimeError("opmgpu_create failed with status %d (no GPU? there is no CPU fallback)" % st)
        self.nc = grid.nc
        self.max_single_precision_days = 20.0       # BlackoilModelParameters.cpp:95
        self.linear_iterations = 0
        self.dt = None
        self.use_update_stabilization = True        # BlackoilModelParameters.cpp:98
        self.residual_norms_history, self.current_relaxation = [], 1.0
        if wells is not None:
            self.setWells(*wells)

    def close(self):
        if self.ctx:
            self.lib.opmgpu_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st):
        _raise(self.lib, self.ctx, st)

    def setWells(self, well_connpos, well_cells):
        cp, wc = capi.i32(well_connpos), capi.i32(well_cells)
        self._chk(self.lib.opmgpu_set_wells(self.ctx, cp.size - 1, capi.iptr(cp), capi.iptr(wc)))

    def setState(self, st):
        self._chk(self.lib.opmgpu_set_state(self.ctx, capi.dptr(st.p), capi.dptr(st.sat), capi.dptr(st.rs), capi.dptr(st.rv), capi.bptr(st.hc)))

    def getState(self):
        from .decks import State
        n = self.nc
        st = State(np.zeros(n), np.zeros((n, 3)), np.zeros(n), np.zeros(n), np.zeros(n, np.int8))
        self._chk(self.lib.opmgpu_get_state(self.ctx, capi.dptr(st.p), capi.dptr(st.sat), capi.dptr(st.rs), capi.dptr(st.rv), capi.bptr(st.hc)))
        return st

    # --- BlackoilModelBase hooks -------------------------------------------------------------
    def prepareStep(self, dt, state=None):
        """prepareStep (:222-232): pvdt = pv/dt is folded into assemble(); state upload if given."""
        self.dt = float(dt)
        if state is not None:
            self.setState(state)

    def assemble(self, initial_assembly):
        self._chk(self.lib.opmgpu_assemble(self.ctx, self.dt, int(initial_assembly), None, None, None, None, None))

    def getConvergence(self):
        B, CNV, MB, linf = np.zeros(3), np.zeros(3), np.zeros(3), np.zeros(3)
        conv = C.c_int(0)
        st = self.lib.opmgpu_convergence(self.ctx, self.dt, capi.dptr(B), capi.dptr(CNV), capi.dptr(MB), capi.dptr(linf), C.byref(conv))
        self.B_avg, self.CNV, self.MB, self.linf = B, CNV, MB, linf
        self._chk(st)
        return bool(conv.value)

    def referencePrecision(self):
        """The arithmetic the reference's plug-in solves in: residual_.singlePrecision = dt < maxSinglePrecisionTimeStep_
        (BlackoilModelBase_impl.hpp:284) is honoured by the interleaved solver only (NewtonIterationBlackoilInterleaved.cpp:478-480);
        the CPR plug-in never reads it and computes in double (NewtonIterationBlackoilCPR.cpp:117-140)."""
        if self.params.use_cpr:
            return False
        return self.dt < self.max_single_precision_days * 86400.0

    def solveJacobianSystem(self, want_dx=False, single_precision=None):
        if single_precision is None:
            single_precision = self.referencePrecision()
        dx = np.zeros(3 * self.nc) if want_dx else None
        it, red = C.c_int(0), C.c_double(0)
        st = self.lib.opmgpu_solve(self.ctx, int(single_precision), capi.dptr(dx), C.byref(it), C.byref(red))
        self.linear_iterations, self.linear_reduction = it.value, red.value
        self._chk(st)
        return dx

    def updateState(self, dx=None, relax=1.0):
        self._chk(self.lib.opmgpu_update_state(self.ctx, capi.dptr(None if dx is None else capi.f64(dx)), float(relax)))

    # last_state of AdaptiveTimeStepping, kept on the device
    def saveState(self):
        self._chk(self.lib.opmgpu_save_state(self.ctx))

    def restoreState(self):
        self._chk(self.lib.opmgpu_restore_state(self.ctx))

    def computeFluidInPlace(self, fipnum=None, cells=False, nregions=None):
        """BlackoilModelBase::computeFluidInPlace (BlackoilModelBase_impl.hpp:2263-2445) for the resident state: values[region][7]
        (water, oil, gas, dissolved gas, vaporised oil, pore volume, hydrocarbon-pv weighted pressure); cells=True also returns the
        per-cell arrays [7][nc] (SimulatorData::fip).  Decomposed runs: collective; every rank passes the fipnum of its local cells
        (ghosts included, they are not counted) and the GLOBAL number of regions, and receives the global sums."""
        fn = None if fipnum is None else capi.i32(fipnum)
        dims = nregions if nregions is not None else (1 if fn is None else max(1, int(fn.max())))
        values = np.zeros((dims, 7))
        fc = np.zeros((7, self.nc)) if cells else None
        self._chk(self.lib.opmgpu_compute_fluid_in_place(self.ctx, capi.iptr(fn), dims, capi.dptr(fc), capi.dptr(values)))
        return (values, fc) if cells else values

    def relativeChange(self):
        """BlackoilModelBase::relativeChange(saved, resident) (BlackoilModelBase_impl.hpp:1595-1631)."""
        v = C.c_double(0.0)
        self._chk(self.lib.opmgpu_relative_change(self.ctx, C.byref(v)))
        return v.value

    # satOilMax_ of BlackoilPropsAdFromDeck (VAPPARS); updateSatOilMax is called once per report step (SimulatorBase_impl.hpp:192)
    def setSatOilMax(self, so_max):
        self._chk(self.lib.opmgpu_set_sat_oil_max(self.ctx, capi.dptr(capi.f64(so_max))))

    def updateSatOilMax(self):
        self._chk(self.lib.opmgpu_update_sat_oil_max(self.ctx))

    def satOilMax(self):
        out = np.zeros(self.nc)
        self._chk(self.lib.opmgpu_get_sat_oil_max(self.ctx, capi.dptr(out)))
        return out

    # hysteresis history (SaturationPropsFromDeck::updateSatHyst, called once per report step: SimulatorBase_impl.hpp:190-191)
    def updateHysteresis(self):
        self._chk(self.lib.opmgpu_update_hysteresis(self.ctx))

    def getHysteresis(self):
        """(krnSwMdc_ow, krnSwMdc_go, delta_ow, delta_go), caller cell order"""
        out = [np.zeros(self.nc) for _ in range(4)]
        self._chk(self.lib.opmgpu_get_hysteresis(self.ctx, *[capi.dptr(a) for a in out]))
        return out

    def setHysteresis(self, mdc_ow, mdc_go):
        self._chk(self.lib.opmgpu_set_hysteresis(self.ctx, capi.dptr(capi.f64(mdc_ow)), capi.dptr(capi.f64(mdc_go))))

    def setSolvePrecision(self, single_precision=None):
        """residual_.singlePrecision = dt < maxSinglePrecisionTimeStep (:284), told to the device BEFORE the assembly so that
        the Jacobian is written in the solve's precision."""
        if single_precision is None:
            single_precision = self.referencePrecision()
        self._chk(self.lib.opmgpu_set_solve_precision(self.ctx, int(bool(single_precision))))

    def stabilizeUpdate(self, relax_type, omega):
        self._chk(self.lib.opmgpu_stabilize_update(self.ctx, int(relax_type), float(omega)))

    def nonlinearIteration(self, iteration, single_precision=None, nonlinear_solver=None):
        """nonlinearIteration (BlackoilModelBase_impl.hpp:239-326). Returns (converged, linear_iterations).
        With a `NonlinearSolver` the update is stabilised exactly like the reference's use_update_stabilization path."""
        ns = nonlinear_solver
        if self.fused_iteration and ns is not None:
            return self._fused_iteration(iteration, single_precision, ns)
        if iteration == 0:
            self.residual_norms_history, self.current_relaxation = [], 1.0
        self.setSolvePrecision(single_precision)
        self.assemble(iteration == 0)
        converged = self.getConvergence()
        self.residual_norms_history.append(list(self.linf))          # computeResidualNorms (:1551-1589)
        lin = 0
        if not converged or iteration < (ns.min_iter if ns else 1):
            self.solveJacobianSystem(single_precision=single_precision)
            lin = self.linear_iterations
            if ns is not None and self.use_update_stabilization:
                oscillate, _ = ns.detectOscillations(self.residual_norms_history, iteration)
                if oscillate:
                    self.current_relaxation = max(self.current_relaxation - ns.relax_increment, ns.relax_max)
                self.stabilizeUpdate(ns.relax_type, self.current_relaxation)
            self.updateState()
        return converged, lin

    # the same iteration through ONE library call (opmgpu_nonlinear_iteration): the host round trips between the phases stay inside the
    # library.  Off by default here (the call-by-call sequence above is what the parity tests walk); bench.py switches it on.
    fused_iteration = False

    def _fused_iteration(self, iteration, single_precision, ns):
        if single_precision is None:
            single_precision = self.referencePrecision()
        ctl = capi.NewtonCtl(int(ns.min_iter), int(self.use_update_stabilization), int(ns.relax_type), float(ns.relax_max), float(ns.relax_increment),
                             float(ns.relax_rel_tol))
        conv, lin, relax = C.c_int(0), C.c_int(0), C.c_double(1.0)
        linf = np.zeros(3)
        st = self.lib.opmgpu_nonlinear_iteration(self.ctx, self.dt, int(iteration), int(bool(single_precision)), C.byref(ctl), C.byref(conv), C.byref(lin),
                                                 capi.dptr(linf), C.byref(relax))
        self.linear_iterations, self.linf, self.current_relaxation = lin.value, linf, relax.value
        self._chk(st)
        return bool(conv.value), lin.value

    # --- parity / bench helpers --------------------------------------------------------------
    def residual(self):
        r = np.zeros(3 * self.nc)
        self._chk(self.lib.opmgpu_get_residual(self.ctx, capi.dptr(r)))
        return r

    def jacobian(self):
        n = C.c_int32(0)
        self._chk(self.lib.opmgpu_get_jacobian_nnzb(self.ctx, C.byref(n)))
        rowptr, col, val = np.zeros(self.nc + 1, np.int32), np.zeros(n.value, np.int32), np.zeros((n.value, 9))
        self._chk(self.lib.opmgpu_get_jacobian_bsr(self.ctx, capi.iptr(rowptr), capi.iptr(col), capi.dptr(val)))
        return rowptr, col, val

    def spmv(self, x3):
        """y = J x with the assembled system (block-interleaved [nc][3]); with device wells J includes their factored
        Schur complement (the rank-7 operator per well), i.e. the operator the linear solver sees."""
        y = np.zeros(3 * self.nc)
        self._chk(self.lib.opmgpu_spmv(self.ctx, capi.dptr(capi.f64(x3)), capi.dptr(y)))
        return y

    def perfProps(self, nperf):
        out = np.zeros((nperf, capi.PERF_K))
        self._chk(self.lib.opmgpu_perf_props(self.ctx, capi.dptr(out)))
        return out

    def perfPvtAt(self, press):
        """b (nperf x 3), rsSat, rvSat of the perforated cells at the given pressures (opmgpu_perf_pvt) -- what the host well
        model's computeWellConnectionPressures needs (StandardWells_impl.hpp:218-296)"""
        press = capi.f64(press)
        out = np.zeros((press.size, 5))
        self._chk(self.lib.opmgpu_perf_pvt(self.ctx, capi.dptr(press), capi.dptr(out)))
        return out[:, :3], out[:, 3].copy(), out[:, 4].copy()

    def averageB(self):
        """B_avg of getWellConvergence for the host well model's pre-solve (opmgpu_average_b)"""
        B = np.zeros(3)
        self._chk(self.lib.opmgpu_average_b(self.ctx, capi.dptr(B)))
        return B

    def addWellTerms(self, resid_delta, rc, blocks):
        rc = capi.i32(rc)
        nblk = rc.size // 2
        self._chk(self.lib.opmgpu_add_well_terms(self.ctx, capi.dptr(capi.f64(resid_delta)), nblk, capi.iptr(rc), capi.dptr(capi.f64(blocks))))

    def addWellRhs(self, rhs_delta):
        self._chk(self.lib.opmgpu_add_well_rhs(self.ctx, capi.dptr(capi.f64(rhs_delta))))

    def perfDx(self, nperf):
        out = np.zeros((nperf, 3))
        self._chk(self.lib.opmgpu_perf_dx(self.ctx, capi.dptr(out)))
        return out

    def timings(self):
        a, s, u = C.c_double(0), C.c_double(0), C.c_double(0)
        self.lib.opmgpu_last_timings(self.ctx, C.byref(a), C.byref(s), C.byref(u))
        return a.value, s.value, u.value

    def time_kernel(self, kernel, reps=20):
        ms = C.c_double(0)
        self._chk(self.lib.opmgpu_time_kernel(self.ctx, kernel, reps, C.byref(ms)))
        return ms.value

    def ordering(self):
        pos, lev = np.zeros(self.nc, np.int32), np.zeros(self.nc, np.int32)
        nl = C.c_int32(0)
        self._chk(self.lib.opmgpu_get_ordering(self.ctx, capi.iptr(pos), capi.iptr(lev), C.byref(nl)))
        return pos, lev, nl.value


class NonlinearSolver:
    """NonlinearSolver (NonlinearSolver_impl.hpp:119-301): step loop, oscillation detection, relaxation parameters."""

    def __init__(self, max_iter=10, min_iter=1, relax_type=capi.RELAX_DAMPEN, relax_max=0.5, relax_increment=0.1, relax_rel_tol=0.2):
        self.max_iter, self.min_iter = max_iter, min_iter               # SolverParameters::reset (:183-192)
        self.relax_type, self.relax_max, self.relax_increment, self.relax_rel_tol = relax_type, relax_max, relax_increment, relax_rel_tol

    def detectOscillations(self, norms, it):
        """The rule of NonlinearSolver::detectOscillations (:221-257) -> (oscillate, stagnate); only the three mass-balance norms take part.
        A phase "swings" when its norm is back within relax_rel_tol (relative to the newest value) of its value two iterations ago but not
        of last iteration's; two swinging phases = oscillation.  Stagnation = no phase moved by more than 0.1 % between the two previous
        iterations."""
        if it < 2:
            return False, False
        now, last, before = norms[it], norms[it - 1], norms[it - 2]
        swinging, any_moved = 0, False
        with np.errstate(divide="ignore", invalid="ignore"):
            for ph in range(3):
                to_before = abs(np.float64(now[ph] - before[ph]) / now[ph])
                to_last = abs(np.float64(now[ph] - last[ph]) / now[ph])
                swinging += int(to_before < self.relax_rel_tol < to_last)
                any_moved = any_moved or bool(abs(np.float64(last[ph] - before[ph]) / before[ph]) > 1.0e-3)
        return swinging >= 2, not any_moved

    def step(self, model, single_precision=None):
        """(:119-174). Returns (newton_iterations, linear_iterations); raises TooManyIterations."""
        it, lin_total = 0, 0
        while True:
            converged, lin = model.nonlinearIteration(it, single_precision=single_precision, nonlinear_solver=self)
            lin_total += lin
            it += 1
            if not ((not converged and it <= self.max_iter) or it <= self.min_iter):
                break
        if not converged:
            raise TooManyIterations("Solver convergence failure - Failed to complete a time step within %d iterations." % self.max_iter)
        return it, lin_total


def newton_step(model, max_iter=10, min_iter=1):
    """NonlinearSolver::step (NonlinearSolver_impl.hpp:119-174) with use_update_stabilization=false (plain Newton).
    Returns (newton_iterations, linear_iterations)."""
    it, lin_total = 0, 0
    while True:
        converged, lin = model.nonlinearIteration(it)
        lin_total += lin
        it += 1
        if not ((not converged and it <= max_iter) or it <= min_iter):
            break
    if not converged:
        raise TooManyIterations("Failed to complete a time step within %d iterations." % max_iter)
    return it, lin_total
