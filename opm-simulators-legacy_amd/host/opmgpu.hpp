// opmgpu.hpp -- C++ host-side mirror of the reference's plug-in interfaces over the C ABI (include/opmgpu.h).
//
// Header-only, no OPM / Eigen / Dune dependency: these are the classes a flow_legacy maintainer derives
// the real adaptors from (INTEGRATION.md shows the OPM-side glue).  Names, call order and error
// behaviour follow the reference:
//   opmgpu::NewtonIterationBlackoilGpu   <->  Opm::NewtonIterationBlackoilInterface
//                                             (opm/autodiff/NewtonIterationBlackoilInterface.hpp:31-52)
//   opmgpu::BlackoilModelGpu             <->  Opm::BlackoilModelBase hooks
//                                             (opm/autodiff/BlackoilModelBase_impl.hpp:222-326)
//   opmgpu::NonlinearSolverGpu::step     <->  Opm::NonlinearSolver::step (NonlinearSolver_impl.hpp:119-174)
// Status codes of the C ABI become the exception types the reference's AdaptiveTimeStepping catches
// (AdaptiveTimeStepping_impl.hpp:244-281).
#ifndef OPMGPU_HOST_HPP
#define OPMGPU_HOST_HPP

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/opmgpu.h"

namespace opmgpu {

struct NumericalIssue : std::runtime_error { using std::runtime_error::runtime_error; };        // Opm::NumericalIssue
struct LinearSolverProblem : std::runtime_error { using std::runtime_error::runtime_error; };   // Opm::LinearSolverProblem
struct ISTLError : std::runtime_error { using std::runtime_error::runtime_error; };             // Dune::ISTLError / MatrixBlockError
struct TooManyIterations : std::runtime_error { using std::runtime_error::runtime_error; };     // Opm::TooManyIterations

inline void throw_on_status(const opmgpu_ctx* ctx, int st)
{
    if (st == OPMGPU_OK) return;
    const std::string msg = ctx ? opmgpu_last_error(ctx) : "";
    switch (st) {
    case OPMGPU_ENUMERICAL: throw NumericalIssue(msg);
    case OPMGPU_ELINSOLVE:  throw LinearSolverProblem(msg);
    case OPMGPU_EBREAKDOWN:
    case OPMGPU_ESINGULAR:  throw ISTLError(msg);
    case OPMGPU_EINVAL:     throw std::logic_error("opmgpu: invalid argument: " + msg);
    default:                throw std::runtime_error("opmgpu status " + std::to_string(st) + ": " + msg);
    }
}

/// BCRSMatrix<MatrixBlock<Scalar,3,3>> + BlockVector handed over as plain arrays
/// (what formInterleavedSystem produces, NewtonIterationBlackoilInterleaved.cpp:110-194, 263-269).
struct BsrSystem {
    int nb = 0;
    std::vector<int32_t> rowptr, col;      // columns ascending per row, diagonal present
    std::vector<double> val9;              // nnzb * 9, row-major blocks  A[row][col][eq][var]
    std::vector<double> rhs3;              // nb * 3, block-interleaved
    bool singlePrecision = false;          // LinearisedBlackoilResidual::singlePrecision
};

/// B1: drop-in for NewtonIterationBlackoilInterleaved's solve stage.
class NewtonIterationBlackoilGpu {
public:
    explicit NewtonIterationBlackoilGpu(const opmgpu_params* prm = nullptr, int device = 0)
    {
        const int st = opmgpu_create_solver(&ctx_, device, prm);
        if (st != OPMGPU_OK) throw std::runtime_error("opmgpu_create_solver failed (no GPU? there is no CPU fallback), status " + std::to_string(st));
    }
    ~NewtonIterationBlackoilGpu() { opmgpu_destroy(ctx_); }
    NewtonIterationBlackoilGpu(const NewtonIterationBlackoilGpu&) = delete;
    NewtonIterationBlackoilGpu& operator=(const NewtonIterationBlackoilGpu&) = delete;

    /// x (nb*3, block-interleaved) with J x = rhs; const like the reference's (mutable internals, :75-79)
    std::vector<double> computeNewtonIncrement(const BsrSystem& sys) const
    {
        std::vector<double> x(size_t(3) * sys.nb, 0.0);
        const int st = opmgpu_solve_bsr(ctx_, sys.nb, sys.rowptr.data(), sys.col.data(), sys.val9.data(), sys.rhs3.data(),
                                        sys.singlePrecision ? 1 : 0, x.data(), &iterations_, &reduction_);
        throw_on_status(ctx_, st);
        return x;
    }
    int iterations() const { return iterations_; }                   // NewtonIterationBlackoilInterface::iterations
    double reduction() const { return reduction_; }
    opmgpu_ctx* handle() const { return ctx_; }

private:
    mutable opmgpu_ctx* ctx_ = nullptr;
    mutable int iterations_ = 0;
    mutable double reduction_ = 0.0;
};

/// ReservoirState view (opm/core/simulator/BlackoilState.hpp:40-90)
struct ReservoirStateView {
    double* pressure; double* saturation; double* gasoilratio; double* rv; int8_t* hydroCarbonState;
};

struct ConvergenceReport { double B_avg[3], CNV[3], MB[3], linf[3]; bool converged; };

/// B2: the BlackoilModel hooks on the device.  The context outlives the model object like the reference's
/// linear solver outlives BlackoilModel (FlowMain.hpp:237 vs SimulatorBase_impl.hpp:201-203).
class BlackoilModelGpu {
public:
    BlackoilModelGpu(const opmgpu_grid& grid, const opmgpu_tables& tables, const opmgpu_params* prm = nullptr, int device = 0)
    {
        nc_ = grid.nc;
        use_cpr_ = prm && prm->use_cpr;
        const int st = opmgpu_create(&ctx_, device, &grid, &tables, prm);
        if (st != OPMGPU_OK) throw std::runtime_error("opmgpu_create failed (no GPU? there is no CPU fallback), status " + std::to_string(st));
    }
    ~BlackoilModelGpu() { opmgpu_destroy(ctx_); }
    BlackoilModelGpu(const BlackoilModelGpu&) = delete;
    BlackoilModelGpu& operator=(const BlackoilModelGpu&) = delete;

    int numPhases() const { return 3; }
    void setWells(int nw, const int32_t* well_connpos, const int32_t* well_cells) { throw_on_status(ctx_, opmgpu_set_wells(ctx_, nw, well_connpos, well_cells)); }

    /// wells on the device (StandardWells restated in csrc/wells.hip): topology + controls, then the WellState fields
    void setDeviceWells(const opmgpu_wells& wells) { throw_on_status(ctx_, opmgpu_set_device_wells(ctx_, &wells)); device_wells_ = wells.nw > 0; }
    void setWellState(const double* bhp, const double* well_rates, const double* perf_press = nullptr, const double* perf_rates = nullptr) { throw_on_status(ctx_, opmgpu_well_state_set(ctx_, bhp, well_rates, perf_press, perf_rates)); }
    void getWellState(double* bhp, double* well_rates, double* perf_press = nullptr, double* perf_rates = nullptr) { throw_on_status(ctx_, opmgpu_well_state_get(ctx_, bhp, well_rates, perf_press, perf_rates)); }
    /// WellStateFullyImplicitBlackoil::currentControls() / thp(); after an assembly also what solveWellEq did (iterations, converged) per well
    void setWellControls(const int32_t* current, const double* thp = nullptr) { throw_on_status(ctx_, opmgpu_well_controls_set(ctx_, current, thp)); }
    void getWellControls(int32_t* current, double* thp = nullptr, int32_t* presolve_iterations = nullptr, int32_t* presolve_converged = nullptr)
    {
        throw_on_status(ctx_, opmgpu_well_controls_get(ctx_, current, thp, presolve_iterations, presolve_converged));
    }
    /// VFPPROD / VFPINJ tables for THP controls (VFPProperties), before setDeviceWells
    void setVfpTables(int n, const opmgpu_vfp_table* tables) { throw_on_status(ctx_, opmgpu_set_vfp_tables(ctx_, n, tables)); }
    /// once per report step, where SimulatorBase::run calls props.updateSatOilMax / updateSatHyst (SimulatorBase_impl.hpp:190-192)
    void updateSatOilMax() { throw_on_status(ctx_, opmgpu_update_sat_oil_max(ctx_)); }
    void updateSatHyst() { throw_on_status(ctx_, opmgpu_update_hysteresis(ctx_)); }
    /// BlackoilModelBase::nonlinearIteration + the NonlinearSolver's update stabilisation in ONE library call (opmgpu_nonlinear_iteration):
    /// the same sequence NonlinearSolverGpu::step issues call by call, without the host round trips between the phases.
    /// Returns true when the iteration found the step converged; throws like the single calls do.
    bool nonlinearIterationOneCall(int iteration, bool single_precision, const opmgpu_newton_ctl& ctl, int* linear_iterations = nullptr)
    {
        int conv = 0, lin = 0;
        throw_on_status(ctx_, opmgpu_nonlinear_iteration(ctx_, dt_, iteration, single_precision ? 1 : 0, &ctl, &conv, &lin, nullptr, nullptr));
        if (linear_iterations) *linear_iterations = lin;
        return conv != 0;
    }
    /// well part of getConvergence (:1769-1779)
    bool wellsConverged(const ConvergenceReport& r)
    {
        if (!device_wells_) return true;
        double flux[3], ctrl = 0.0;
        throw_on_status(ctx_, opmgpu_well_convergence(ctx_, flux, &ctrl));
        bool ok = ctrl < tolerance_well_control_;
        for (int a = 0; a < 3; ++a) ok = ok && (r.B_avg[a] * flux[a] < tolerance_wells_);
        return ok;
    }

    /// prepareStep (:222-232): remembers dt (pvdt = pv/dt is applied in assemble) and uploads the state
    void prepareStep(double dt, const ReservoirStateView& s)
    {
        dt_ = dt;
        throw_on_status(ctx_, opmgpu_set_state(ctx_, s.pressure, s.saturation, s.gasoilratio, s.rv, s.hydroCarbonState));
    }
    /// the next time step from the state that is resident on the device (AdaptiveTimeStepping's sub-steps)
    void prepareStep(double dt) { dt_ = dt; }
    void assemble(bool initial_assembly) { throw_on_status(ctx_, opmgpu_assemble(ctx_, dt_, initial_assembly ? 1 : 0, nullptr, nullptr, nullptr, nullptr, nullptr)); }
    ConvergenceReport getConvergence()
    {
        ConvergenceReport r; int conv = 0;
        const int st = opmgpu_convergence(ctx_, dt_, r.B_avg, r.CNV, r.MB, r.linf, &conv);
        r.converged = conv != 0;
        throw_on_status(ctx_, st);         // NumericalIssue on NaN / too large residual
        return r;
    }
    /// residual_.singlePrecision = dt < maxSinglePrecisionTimeStep (:284) is honoured by the interleaved solver only
    /// (NewtonIterationBlackoilInterleaved.cpp:478-480); the CPR plug-in computes in double whatever it says (NewtonIterationBlackoilCPR.cpp:117-140)
    bool referencePrecisionIsSingle() const { return !use_cpr_ && dt_ < max_single_precision_days_ * 86400.0; }
    /// solveJacobianSystem (:1139-1145)
    void solveJacobianSystem()
    {
        const int single = referencePrecisionIsSingle() ? 1 : 0;
        throw_on_status(ctx_, opmgpu_solve(ctx_, single, nullptr, &linear_iterations_, &linear_reduction_));
    }
    void updateState(double relax = 1.0) { throw_on_status(ctx_, opmgpu_update_state(ctx_, nullptr, relax)); }
    void downloadState(const ReservoirStateView& s) { throw_on_status(ctx_, opmgpu_get_state(ctx_, s.pressure, s.saturation, s.gasoilratio, s.rv, s.hydroCarbonState)); }

    /// nonlinearIteration (:239-326): assemble -> getConvergence -> [solve -> stabilise -> update].
    /// NonlinearSolverType supplies minIter / detectOscillations / relaxIncrement / relaxMax / relaxType like the reference's.
    template <class NonlinearSolverType>
    bool nonlinearIteration(int iteration, const NonlinearSolverType& nonlinear_solver)
    {
        if (iteration == 0) { residual_norms_history_.clear(); current_relaxation_ = 1.0; }    // dx_old is zeroed by the initial assembly
        throw_on_status(ctx_, opmgpu_set_solve_precision(ctx_, referencePrecisionIsSingle() ? 1 : 0));   // :284, before the assembly
        assemble(iteration == 0);
        ConvergenceReport r = getConvergence();
        r.converged = wellsConverged(r) && r.converged;
        residual_norms_history_.push_back({ r.linf[0], r.linf[1], r.linf[2] });               // computeResidualNorms (:1551-1589)
        const bool must_solve = (iteration < nonlinear_solver.minIter()) || !r.converged;
        if (must_solve) {
            solveJacobianSystem();
            if (use_update_stabilization_) {
                bool isOscillate = false, isStagnate = false;
                nonlinear_solver.detectOscillations(residual_norms_history_, iteration, isOscillate, isStagnate);
                if (isOscillate) current_relaxation_ = std::max(current_relaxation_ - nonlinear_solver.relaxIncrement(), nonlinear_solver.relaxMax());
                throw_on_status(ctx_, opmgpu_stabilize_update(ctx_, nonlinear_solver.relaxType(), current_relaxation_));
            }
            updateState();
        }
        return r.converged;
    }
    /// last_state of AdaptiveTimeStepping kept on the device, and BlackoilModelBase::relativeChange (:1595-1631) against it
    void saveState() { throw_on_status(ctx_, opmgpu_save_state(ctx_)); }
    void restoreState() { throw_on_status(ctx_, opmgpu_restore_state(ctx_)); }
    double relativeChange() { double v = 0.0; throw_on_status(ctx_, opmgpu_relative_change(ctx_, &v)); return v; }
    /// BlackoilModelBase::computeFluidInPlace(state, fipnum) (:2263-2445) for the resident state: values[region][7]
    std::vector<std::vector<double>> computeFluidInPlace(const std::vector<int>& fipnum)
    {
        int dims = 1;
        for (int f : fipnum) dims = std::max(dims, f);
        std::vector<int32_t> fn(fipnum.begin(), fipnum.end());
        std::vector<double> flat(std::size_t(dims) * 7, 0.0);
        throw_on_status(ctx_, opmgpu_compute_fluid_in_place(ctx_, fn.empty() ? nullptr : fn.data(), dims, nullptr, flat.data()));
        std::vector<std::vector<double>> values(dims, std::vector<double>(7));
        for (int r = 0; r < dims; ++r) for (int k = 0; k < 7; ++k) values[r][k] = flat[std::size_t(r) * 7 + k];
        return values;
    }
    void setStepLength(double dt) { dt_ = dt; }
    double relaxation() const { return current_relaxation_; }
    void setUseUpdateStabilization(bool on) { use_update_stabilization_ = on; }
    int linearIterationsLastSolve() const { return linear_iterations_; }
    opmgpu_ctx* handle() const { return ctx_; }

private:
    opmgpu_ctx* ctx_ = nullptr;
    int nc_ = 0;
    bool use_cpr_ = false;
    double dt_ = 0.0, max_single_precision_days_ = 20.0, linear_reduction_ = 0.0;
    int linear_iterations_ = 0;
    bool use_update_stabilization_ = true;              // BlackoilModelParameters.cpp:98
    bool device_wells_ = false;
    double tolerance_wells_ = 1e-4, tolerance_well_control_ = 1e-7;    // BlackoilModelParameters.cpp:88-89
    double current_relaxation_ = 1.0;
    std::vector<std::array<double, 3>> residual_norms_history_;
};

/// NonlinearSolver (NonlinearSolver_impl.hpp:119-301): step loop, oscillation detection, relaxation parameters
/// RateConverter::SurfaceToReservoirVoidage (RateConverterLegacy.hpp:407-770) over the resident state: defineState takes the regions'
/// sums from the library (collective in decomposed runs), calcCoeff evaluates the PVT tables on the device at the region's average state.
/// `region` empty = one region 0 of all cells, what SimulatorBase builds (SimulatorBase_impl.hpp:66) and asks for (:548).
class SurfaceToReservoirVoidageGpu {
public:
    SurfaceToReservoirVoidageGpu(opmgpu_ctx* ctx, const std::vector<int>& region = {}, int n_ranks = 1)
        : ctx_(ctx), n_ranks_(n_ranks)
    {
        ids_ = region;
        std::sort(ids_.begin(), ids_.end());
        ids_.erase(std::unique(ids_.begin(), ids_.end()), ids_.end());
        if (ids_.empty()) ids_.push_back(0);
        index_.reserve(region.size());
        for (int r : region) index_.push_back(int32_t(std::lower_bound(ids_.begin(), ids_.end(), r) - ids_.begin()));
        attr_.assign(ids_.size(), Attributes{ 0.0, 0.0, 0.0 });          // Attributes(): all zero (:684-692)
    }
    /// calcAverages (:718-768).  p is cleared before the loop, rs and rv are NOT (:733-737): they start from the previous call's averages
    /// (on every rank of a parallel run) -- restated as found.
    void defineState()
    {
        std::vector<double> sums(4 * ids_.size(), 0.0);
        throw_on_status(ctx_, opmgpu_region_state_sums(ctx_, index_.empty() ? nullptr : index_.data(), int(ids_.size()), sums.data()));
        for (std::size_t k = 0; k < ids_.size(); ++k) {
            const double n = sums[4 * k + 3];
            attr_[k].pressure = sums[4 * k + 0] / n;
            attr_[k].rs = (n_ranks_ * attr_[k].rs + sums[4 * k + 1]) / n;
            attr_[k].rv = (n_ranks_ * attr_[k].rv + sums[4 * k + 2]) / n;
        }
    }
    /// calcCoeff (:495-548): q_rT = sum_p coeff[p] q_s[p], phases water, oil, gas
    template <class Coeff> void calcCoeff(int r, int pvtRegionIdx, Coeff& coeff) const
    {
        const std::size_t k = std::size_t(std::lower_bound(ids_.begin(), ids_.end(), r) - ids_.begin());
        if (k >= ids_.size() || ids_[k] != r) throw std::invalid_argument("SurfaceToReservoirVoidageGpu::calcCoeff: unknown region");
        const int32_t reg = pvtRegionIdx;
        double c[3];
        throw_on_status(ctx_, opmgpu_voidage_coefficients(ctx_, 1, &attr_[k].pressure, &attr_[k].rs, &attr_[k].rv, &reg, c));
        for (int p = 0; p < 3; ++p) coeff[p] = c[p];
    }
    struct Attributes { double pressure, rs, rv; };
    const Attributes& attributes(int r) const { return attr_[std::size_t(std::lower_bound(ids_.begin(), ids_.end(), r) - ids_.begin())]; }

private:
    opmgpu_ctx* ctx_;
    int n_ranks_;
    std::vector<int> ids_;
    std::vector<int32_t> index_;
    std::vector<Attributes> attr_;
};

struct NonlinearSolverGpu {
    int max_iter = 10, min_iter = 1;                                          // SolverParameters::reset (:183-192)
    int relax_type = OPMGPU_RELAX_DAMPEN;
    double relax_max = 0.5, relax_increment = 0.1, relax_rel_tol = 0.2;
    int minIter() const { return min_iter; }
    int maxIter() const { return max_iter; }
    int relaxType() const { return relax_type; }
    double relaxMax() const { return relax_max; }
    double relaxIncrement() const { return relax_increment; }
    double relaxRelTol() const { return relax_rel_tol; }

    /// The rule of NonlinearSolver::detectOscillations (NonlinearSolver_impl.hpp:221-257) in this library's words.  A phase's residual norm
    /// "swings" when this iteration's value is back within relax_rel_tol of where it stood TWO iterations ago while it differs by more
    /// than that from LAST iteration's; the update oscillates when at least two of the three phases swing.  It stagnates when no phase's
    /// norm moved by more than 0.1 % between the two previous iterations.  All differences are relative (to the newest value for the
    /// swing test, to the oldest for the stagnation test).
    void detectOscillations(const std::vector<std::array<double, 3>>& norms, int it, bool& oscillate, bool& stagnate) const
    {
        oscillate = false; stagnate = false;
        if (it < 2) return;
        const std::array<double, 3>&now = norms[it], &last = norms[it - 1], &before = norms[it - 2];
        int swinging = 0;
        bool any_moved = false;
        for (int ph = 0; ph < 3; ++ph) {
            const double to_before = std::abs((now[ph] - before[ph]) / now[ph]);
            const double to_last = std::abs((now[ph] - last[ph]) / now[ph]);
            if (to_before < relax_rel_tol && relax_rel_tol < to_last) ++swinging;
            if (std::abs((last[ph] - before[ph]) / before[ph]) > 1.0e-3) any_moved = true;
        }
        oscillate = swinging >= 2;
        stagnate = !any_moved;
    }

    int step(BlackoilModelGpu& model) const
    {
        int iteration = 0; bool converged = false;
        do {
            converged = model.nonlinearIteration(iteration, *this);
            ++iteration;
        } while ((!converged && iteration <= max_iter) || iteration <= min_iter);
        if (!converged) throw TooManyIterations("Failed to complete a time step within " + std::to_string(max_iter) + " iterations.");
        return iteration;
    }
};

/// PIDTimeStepControl / AdaptiveSimulatorTimer (opm-core, not in the reference tree: restated) and
/// AdaptiveTimeStepping::stepImpl (AdaptiveTimeStepping_impl.hpp:183-372) around the device-resident Newton loop.
struct PIDTimeStepControl {
    double tol = 1e-1, errors[3] = { 1e-1, 1e-1, 1e-1 };
    int target_iterations = 0;                         // > 0: the "pid+iteration" variant
    double computeTimeStepSize(double dt, int iterations, double relative_change)
    {
        errors[0] = errors[1]; errors[1] = errors[2]; errors[2] = relative_change;
        for (double e : errors) if (!std::isfinite(e)) throw NumericalIssue("non-finite relative change in the time step control");
        double est;
        if (errors[2] > tol) est = dt * tol / errors[2];
        else est = dt * std::pow(errors[1] / errors[2], 0.075) * std::pow(tol / errors[2], 0.175) * std::pow(errors[0] * errors[0] / errors[1] / errors[2], 0.01);
        if (target_iterations > 0 && iterations > target_iterations) est *= double(target_iterations) / double(iterations);
        return est;
    }
};

struct AdaptiveTimeSteppingGpu {
    double restart_factor = 0.33, growth_factor = 2.0, max_growth = 3.0, max_time_step = 365.0 * 86400.0;     // :101-112
    int solver_restart_max = 10;
    double min_time_step_fraction = 1e-12;             // NOT in the reference: a failed sub-step shorter than this fraction of the report step ends it (the
                                                       // reference's loop resets its restart counter after every converged sub-step and can shrink for ever); 0 = off
    double suggested_next_timestep = 86400.0;
    PIDTimeStepControl control;
    std::vector<double> substeps;                      // of the last report step
    int failed_substeps = 0;

    static double clip(double est, double remaining, double max_step)        // AdaptiveSimulatorTimer::provideTimeStepEstimate
    {
        double dt = std::min(est, max_step);
        if (remaining > 0) {
            if (1.05 * dt > remaining) { dt = remaining; if (dt > max_step) dt = 0.5 * remaining; return dt; }
            if (1.5 * dt > remaining) dt = 0.5 * remaining;
        }
        return dt;
    }
    /// one report step of length `timestep` [s] from the model's resident state
    void step(double timestep, const NonlinearSolverGpu& solver, BlackoilModelGpu& model)
    {
        if (suggested_next_timestep < 0) suggested_next_timestep = restart_factor * timestep;
        double done = 0.0, dt = clip(suggested_next_timestep, timestep, max_time_step);
        model.saveState();
        substeps.clear();
        int restarts = 0;
        while (timestep - done > 1e-9 * std::max(1.0, timestep)) {
            bool converged = false; int linear = 0;
            try {
                model.setStepLength(dt);
                solver.step(model);
                linear = model.linearIterationsLastSolve();
                converged = true;
            }
            catch (const TooManyIterations&) {}
            catch (const LinearSolverProblem&) {}
            catch (const NumericalIssue&) {}
            catch (const std::runtime_error&) {}
            if (converged) {
                done += dt; substeps.push_back(dt);
                double est = std::min(control.computeTimeStepSize(dt, linear, model.relativeChange()), max_growth * dt);
                if (restarts > 0) { est = std::min(growth_factor * dt, est); restarts = 0; }
                dt = clip(est, timestep - done, max_time_step);
                model.saveState();
            } else {
                ++failed_substeps;
                if (min_time_step_fraction > 0 && dt < min_time_step_fraction * timestep) throw NumericalIssue("Solver failed to converge with a time step of " + std::to_string(dt) + " s: giving up.");
                if (restarts >= solver_restart_max) throw NumericalIssue("Solver failed to converge after cutting timestep " + std::to_string(restarts) + " times.");
                dt = clip(restart_factor * dt, timestep - done, max_time_step);
                model.restoreState();
                ++restarts;
            }
        }
        suggested_next_timestep = std::isfinite(dt) ? dt : timestep;
    }
};

} // namespace opmgpu
#endif
