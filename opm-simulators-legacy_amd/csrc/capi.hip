// capi.hip -- the extern "C" boundary of libopmgpu.so (include/opmgpu.h).
//
// Every entry point catches C++ exceptions and converts them to the status codes the reference-side
// shim maps back to the exception flow_legacy's time stepper expects.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <array>
#include <memory>
#include <string>

#include "blackoil.hpp"
#include "dist.hpp"
#include "linsolver.hpp"

using namespace opmgpu;

struct opmgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    opmgpu_params prm;
    std::unique_ptr<LinSolver> ls;
    std::unique_ptr<BlackoilDevice> model;
    std::string err;
    int cur_single = 0;          // precision of the loaded / prepared matrix
    bool matrix_loaded = false;
    bool factored = false;
    // phase timings (assemble, solve, update): events recorded around the phase, resolved only when opmgpu_last_timings asks --
    // a synchronisation at the end of every call would leave the GPU idle until the host has enqueued the next phase
    double t_phase[3] = { 0, 0, 0 };
    hipEvent_t evp[3][2] = { { nullptr, nullptr }, { nullptr, nullptr }, { nullptr, nullptr } };
    bool ev_pending[3] = { false, false, false };
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::unique_ptr<RcclComm> comm;   // multi-GPU (dist.hip); empty = single GPU
    // opmgpu_nonlinear_iteration: residual_norms_history_ / current_relaxation_ of BlackoilModelBase (:244-247, :300-315)
    std::vector<std::array<double, 3>> norm_history;
    double relaxation = 1.0;
    // opmgpu_iteration_marks: one event at the end of every opmgpu_nonlinear_iteration call and one pair per phase, kept per call (nothing
    // is synchronised while the marks are on; opmgpu_iteration_marks_get resolves them afterwards)
    struct IterMark { hipEvent_t end = nullptr; hipEvent_t ph[3][2] = { { nullptr, nullptr }, { nullptr, nullptr }, { nullptr, nullptr } }; int solved = 0, lin = 0, status = 0; };
    bool marks_on = false;
    hipEvent_t mark_start = nullptr;
    std::vector<IterMark> marks;
    IterMark* cur_mark = nullptr;
    // events of the marks are recycled through a pool (a caller that leaves the marks on for a long run does not create seven new events per
    // call for ever), and the list is capped: beyond kMaxMarks calls the marks stop being recorded (n_calls of marks_get keeps counting)
    static constexpr size_t kMaxMarks = 16384;
    long marks_dropped = 0;
    std::vector<hipEvent_t> mark_pool;
    hipEvent_t mark_event() { if (!mark_pool.empty()) { hipEvent_t e = mark_pool.back(); mark_pool.pop_back(); return e; } hipEvent_t e = nullptr; return hipEventCreate(&e) == hipSuccess ? e : nullptr; }
    void marks_clear(bool destroy = false) {
        for (IterMark& m : marks) { if (m.end) mark_pool.push_back(m.end); for (auto& pr : m.ph) for (auto& e : pr) if (e) mark_pool.push_back(e); }
        marks.clear(); cur_mark = nullptr; marks_dropped = 0;
        if (mark_start) { mark_pool.push_back(mark_start); mark_start = nullptr; }
        if (destroy) { for (hipEvent_t e : mark_pool) (void)hipEventDestroy(e); mark_pool.clear(); }
    }
};

namespace {

int fail(opmgpu_ctx* c, int code, const std::string& msg) { if (c) c->err = msg; return code; }

template <class F> int guarded(opmgpu_ctx* c, F&& f)
{
    try {
        if (c) { if (hipSetDevice(c->device) != hipSuccess) return fail(c, OPMGPU_ENODEVICE, "hipSetDevice failed"); }
        return f();
    } catch (const HipError& e) { return fail(c, e.code, e.what()); }
    catch (const std::bad_alloc&) { return fail(c, OPMGPU_ENOMEM, "host allocation failed"); }
    catch (const std::exception& e) { return fail(c, OPMGPU_EINVAL, e.what()); }
}

int make_ctx(opmgpu_ctx** out, int device, const opmgpu_params* params)
{
    if (!out) return OPMGPU_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1 || device < 0 || device >= n) return OPMGPU_ENODEVICE;   // no CPU fallback
    std::unique_ptr<opmgpu_ctx> c(new opmgpu_ctx());
    c->device = device;
    if (params) c->prm = *params; else opmgpu_default_params(&c->prm);
    if (hipSetDevice(device) != hipSuccess) return OPMGPU_ENODEVICE;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return OPMGPU_ENODEVICE;
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) return OPMGPU_ENODEVICE;
    for (auto& pr : c->evp) for (auto& e : pr) if (hipEventCreate(&e) != hipSuccess) return OPMGPU_ENODEVICE;
    try { c->ls.reset(new LinSolver(c->stream)); } catch (const HipError& e) { return e.code; }
    *out = c.release();
    return OPMGPU_OK;
}

enum { PH_ASSEMBLE = 0, PH_SOLVE = 1, PH_UPDATE = 2 };
struct Timed {
    opmgpu_ctx* c; int ph;
    hipEvent_t e1 = nullptr;
    Timed(opmgpu_ctx* c_, int phase) : c(c_), ph(phase) {
        if (c->cur_mark) {                  // iteration marks: this call's own event pair (resolved by opmgpu_iteration_marks_get)
            hipEvent_t e0 = c->mark_event();
            e1 = c->mark_event();
            if (e0 && e1) {
                // a phase that runs twice in one call (never today) keeps its last pair
                for (auto& e : c->cur_mark->ph[ph]) if (e) { c->mark_pool.push_back(e); e = nullptr; }
                c->cur_mark->ph[ph][0] = e0; c->cur_mark->ph[ph][1] = e1;
                (void)hipEventRecord(e0, c->stream);
                return;
            }
            e1 = nullptr;
        }
        c->ev_pending[ph] = false; (void)hipEventRecord(c->evp[ph][0], c->stream);
    }
    ~Timed() { if (e1) { (void)hipEventRecord(e1, c->stream); return; } (void)hipEventRecord(c->evp[ph][1], c->stream); c->ev_pending[ph] = true; }
};

static inline int fill_level_of(const opmgpu_params& prm) { return prm.use_cpr ? prm.cpr_ilu_n : prm.ilu_fillin_level; }

template <class S> int solve_loaded(opmgpu_ctx* c, bool matrix_changed, SolveResult& res)
{
    LinSolver& ls = *c->ls;
    // under CPR the second stage's ILU0 takes the CPR preconditioner's own relaxation (cpr_relax, NewtonIterationBlackoilCPR.hpp:59), not
    // the interleaved solver's ilu_relaxation: the solvers below read ONE relaxation field
    opmgpu_params prm = c->prm;
    if (prm.use_cpr) {
        if (prm.cpr_ilu_n > 0 && prm.cpr_reference_transform == 2) return fail(c, OPMGPU_EINVAL, "cpr_ilu_n > 0 with cpr_reference_transform = 2: the point ILU of the scalar system is an ILU0");
        if (!(prm.cpr_relax > 0.0) || !(prm.cpr_solver_tol > 0.0) || prm.cpr_max_ell_iter < 0) return fail(c, OPMGPU_EINVAL, "cpr_relax / cpr_solver_tol must be positive, cpr_max_ell_iter >= 0");
        if (!(prm.cpr_stage2_relax > 0.0)) return fail(c, OPMGPU_EINVAL, "cpr_stage2_relax must be positive");
        prm.ilu_relaxation = prm.cpr_relax * prm.cpr_stage2_relax;
        // the elliptic part (elliptic.inl): cpr_max_ell_iter = 0 is this library's one-V-cycle stage, which needs the AMG
        if (prm.cpr_max_ell_iter == 0 && !prm.cpr_use_amg) return fail(c, OPMGPU_EINVAL, "cpr_max_ell_iter = 0 (one application, no inner Krylov method) needs cpr_use_amg = 1");
        ls.ell.inner = prm.cpr_max_ell_iter > 0; ls.ell.use_amg = prm.cpr_use_amg != 0; ls.ell.bicgstab = prm.cpr_use_bicgstab != 0;
        ls.ell.tol = prm.cpr_solver_tol; ls.ell.maxit = prm.cpr_max_ell_iter; ls.ell.relax = prm.cpr_relax;
    }
    ls.wb_relax = prm.ilu_relaxation;
    // block ILU(n) instead of the ILU0 (fillilu.inl): cpr_ilu_n under CPR, the interleaved solver's ilu_fillin_level otherwise
    const int fill = fill_level_of(prm);
    if (fill < 0 || fill > 8) return fail(c, OPMGPU_EINVAL, "cpr_ilu_n / ilu_fillin_level must be in 0..8");
    if (fill != ls.fill_level) { ls.join_factor(); ls.factor_early = 0; ls.fill_level = fill; }        // (an early factorisation of the other kind is void)
    if (prm.use_cpr) ls.correction_policy_choose();
    // mixed precision (preconditioner_single): a double solve with its preconditioner in the float work set; not with the in-place transform
    const bool mixed = prm.preconditioner_single && sizeof(S) == 8 && !(prm.use_cpr && prm.cpr_reference_transform) && ls.emulate_ranks <= 1;
    ls.mixed = mixed;
    ls.prepare<S>(matrix_changed);
    if (mixed) ls.mixed_prepare(matrix_changed);
    // the reference's CPR formulation (whole-system L transform, 200-bar pressure row, ||L r|| stopping) as an option; once per matrix
    if (prm.use_cpr && prm.cpr_reference_transform) ls.cpr_reference_transform<S>();
    else { ls.border_weights = nullptr; ls.border_colscale = 1.0; }
    // cpr_reference_transform = 2: the reference's own second stage, a point ILU0 of the transformed system as a scalar equation-major matrix
    ls.point_stage2 = prm.use_cpr && prm.cpr_reference_transform == 2;
    if (ls.point_stage2) {
        if (sizeof(S) != 8) return fail(c, OPMGPU_EINVAL, "cpr_reference_transform = 2 (point ILU0 of the scalar system) is double only, like the reference's CPR plug-in");
        if (ls.comm || ls.emulate_ranks > 1) return fail(c, OPMGPU_EINVAL, "cpr_reference_transform = 2 is a single-GPU comparison mode");
        if (ls.pilu.stale) { ls.point_ilu_factor(); ls.pilu.stale = false; }
    }
    // next to the pressure stage's set-up (cpr_prepare, inside the solver); not in the emulated-decomposition diagnostics, whose cut copy of
    // the matrix is built lazily by whichever of the two asks first
    static const bool after_rows = !(std::getenv("OPMGPU_FACTOR_AFTER_ROWS") && std::atoi(std::getenv("OPMGPU_FACTOR_AFTER_ROWS")) == 0);      // measured +0.5 %
    const bool early = ls.factor_early == (mixed ? 4 : int(sizeof(S))) && matrix_changed && !prm.cpr_reference_transform;      // the model started it behind the assembly (LinSolver::factor_early)
    ls.factor_early = 0;
    if (ls.point_stage2) { /* the point ILU0 above is the second stage: no block factorisation */ }
    else if (early) { /* running on the factor stream already; the first ILU0 sweep joins it */ }
    else if (ls.factor_overlap && prm.use_cpr && ls.emulate_ranks <= 1) { if (after_rows) ls.factor_deferred = true; else if (mixed) ls.factor_async<float>(); else ls.factor_async<S>(); }
    else if (mixed) (void)ls.factor<float>(false);
    else (void)ls.factor<S>(false);      // status read below: the solver's own final synchronisation covers it
    res = prm.newton_use_gmres ? ls.gmres<S>(prm) : ls.bicgstab<S>(prm);
    if (prm.use_cpr) ls.correction_policy_report(res.iterations, res.status == OPMGPU_OK);
    if (res.status != OPMGPU_OK && prm.use_cpr && ls.corr_policy.active && ls.corr_policy.cur > 0 && ls.factor_status() == OPMGPU_OK) {
        // the solve ran with a scaled coarse-grid correction: once more with the plain Galerkin correction (factor 1.0, the safe end of the policy's
        // ladder) before anything is reported -- a failed linear solve costs the caller a chopped time step.  The rest of the step stays on it;
        // the failure counts against the factor at once and bans it and every larger one for a while (LinSolver::CorrectionPolicy)
        ls.corr_policy.fail_at_current(res.iterations);
        if (mixed) ls.work<float>().amg->pdamp0 = ls.work<float>().amg->pdamp = ls.corr_policy.arm[0];
        else ls.work<S>().amg->pdamp0 = ls.work<S>().amg->pdamp = ls.corr_policy.arm[0];
        res = prm.newton_use_gmres ? ls.gmres<S>(prm) : ls.bicgstab<S>(prm);
        ls.correction_policy_report(res.iterations, res.status == OPMGPU_OK);
    }
    if (res.status != OPMGPU_OK && prm.use_cpr && !ls.refreshed && ls.factor_status() == OPMGPU_OK) {
        // the solve ran on lagged coarse operators of the pressure hierarchy (LinSolver::cpr_prepare): once more on fresh ones
        ls.force_refresh = true; ls.lag_block = 8;
        res = prm.newton_use_gmres ? ls.gmres<S>(prm) : ls.bicgstab<S>(prm);
    }
    if (!ls.point_stage2 && ls.factor_status() != OPMGPU_OK) { c->factored = false; return fail(c, OPMGPU_ESINGULAR, "singular diagonal block in ILU0"); }
    c->factored = true;
    if (res.status == OPMGPU_ELINSOLVE) c->err = "Convergence failure for linear solver.";
    if (res.status == OPMGPU_EBREAKDOWN) c->err = ls.breakdown_note.empty() ? std::string("breakdown in the Krylov method") : ls.breakdown_note;
    return res.status;
}

} // namespace

extern "C" {

const char* opmgpu_version(void) { return "opmgpu 0.1 (gfx950, HIP, SELL-64 block-ILU0/BiCGStab + black-oil assembly)"; }

int opmgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void opmgpu_default_params(opmgpu_params* p)
{
    if (!p) return;
    p->dp_max_rel = 0.3; p->ds_max = 0.2; p->dr_max_rel = 1e9;                 // BlackoilModelParameters.cpp:76-102
    p->max_residual_allowed = 1e7; p->tolerance_mb = 1e-5; p->tolerance_cnv = 1e-2;
    p->matbalscale[0] = 1.1169; p->matbalscale[1] = 1.0031; p->matbalscale[2] = 0.0031;   // BlackoilModelBase_impl.hpp:139
    p->linear_solver_reduction = 1e-2; p->linear_solver_maxiter = 150;                    // FlowLinearSolverParameters
    p->ilu_relaxation = 0.9; p->ilu_ordering = OPMGPU_ORDER_MULTICOLOR; p->ignore_convergence_failure = 0; p->use_cpr = 0;
    p->newton_use_gmres = 0; p->linear_solver_restart = 40;                                // NewtonIterationBlackoilCPR.cpp:61-64
    p->solve_welleq_initially = 1; p->tolerance_wells = 1e-4; p->tolerance_well_control = 1e-7; p->dbhp_max_rel = 1.0; p->update_equations_scaling = 0; p->gmres_verify_residual = 0; p->cpr_reference_transform = 0;   // BlackoilModelParameters.cpp:80-96
    p->cpr_relax = 1.0; p->cpr_ilu_n = 0; p->cpr_use_amg = 0; p->cpr_use_bicgstab = 1;       // NewtonIterationBlackoilCPR.hpp:59-63
    p->ilu_fillin_level = 0;                                                                   // ISTLSolver.hpp:205
    p->cpr_solver_tol = 1e-2; p->cpr_stage2_relax = 1.0; p->preconditioner_single = 0; p->cpr_max_ell_iter = 25;                                       // external CPRPreconditioner (recollection, see opmgpu.h)
}

int opmgpu_create_solver(opmgpu_ctx** ctx, int device, const opmgpu_params* params) { return make_ctx(ctx, device, params); }

int opmgpu_create(opmgpu_ctx** ctx, int device, const opmgpu_grid* grid, const opmgpu_tables* tables, const opmgpu_params* params)
{
    if (!grid || !tables || grid->nc <= 0 || grid->nconn < 0) return OPMGPU_EINVAL;
    const int st = make_ctx(ctx, device, params);
    if (st != OPMGPU_OK) return st;
    opmgpu_ctx* c = *ctx;
    const int st2 = guarded(c, [&]() { c->model.reset(new BlackoilDevice(c->stream, *c->ls, grid, tables, &c->prm)); return OPMGPU_OK; });
    if (st2 != OPMGPU_OK) { opmgpu_destroy(c); *ctx = nullptr; }
    return st2;
}

void opmgpu_destroy(opmgpu_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->model.reset(); c->ls.reset(); c->comm.reset();
    c->marks_clear(true);
    for (auto& pr : c->evp) for (auto& e : pr) if (e) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* opmgpu_last_error(const opmgpu_ctx* c) { return c ? c->err.c_str() : "null context"; }

int opmgpu_set_wells(opmgpu_ctx* c, int nw, const int32_t* well_connpos, const int32_t* well_cells)
{
    if (!c || !c->model) return OPMGPU_EINVAL;
    return guarded(c, [&]() {
        c->matrix_loaded = false;
        if (c->comm && nw > 0) return fail(c, OPMGPU_EINVAL, "wells are not supported in multi-GPU mode yet");
        return c->model->set_wells(nw, well_connpos, well_cells);
    });
}

int opmgpu_set_state(opmgpu_ctx* c, const double* p, const double* sat, const double* rs, const double* rv, const int8_t* hc)
{
    if (!c || !c->model || !p || !sat || !rs || !rv || !hc) return OPMGPU_EINVAL;
    return guarded(c, [&]() { c->model->set_state(p, sat, rs, rv, hc); return OPMGPU_OK; });
}
int opmgpu_get_state(opmgpu_ctx* c, double* p, double* sat, double* rs, double* rv, int8_t* hc)
{
    if (!c || !c->model || !c->model->has_state) return OPMGPU_EINVAL;
    return guarded(c, [&]() { c->model->get_state(p, sat, rs, rv, hc); return OPMGPU_OK; });
}

int opmgpu_assemble(opmgpu_ctx* c, double dt, int initial, const double* p, const double* sat, const double* rs, const double* rv, const int8_t* hc)
{
    if (!c || !c->model || !(dt > 0.0)) return OPMGPU_EINVAL;
    return guarded(c, [&]() {
        if (p || sat || rs || rv || hc) {
            if (!(p && sat && rs && rv && hc)) return fail(c, OPMGPU_EINVAL, "state pointers must be all set or all NULL");
            c->model->set_state(p, sat, rs, rv, hc);
        }
        if (!c->model->has_state) return fail(c, OPMGPU_EINVAL, "no reservoir state on the device");
        c->t_phase[PH_SOLVE] = 0.0; c->t_phase[PH_UPDATE] = 0.0;       // a Newton iteration that converges here runs no solve / update
        c->ev_pending[PH_SOLVE] = false; c->ev_pending[PH_UPDATE] = false;
        {
            Timed t(c, PH_ASSEMBLE);
            c->model->assemble(dt, initial != 0);
        }
        c->matrix_loaded = true; c->factored = false; c->cur_single = -1; c->ls->ref_transformed = false;
        return int(OPMGPU_OK);
    });
}

int opmgpu_cpr_elliptic_stats(opmgpu_ctx* c, int64_t* solves, int64_t* iterations)
{
    if (!c || !c->ls) return OPMGPU_EINVAL;
    if (solves) *solves = c->ls->ell_solves;
    if (iterations) *iterations = c->ls->ell_iterations;
    return OPMGPU_OK;
}

int opmgpu_cpr_correction_factors(opmgpu_ctx* c, double* into_level0, double* below)
{
    if (!c || !c->ls) return OPMGPU_EINVAL;
    LinSolver& ls = *c->ls;
    const bool f = c->cur_single == 1 || ls.mixed;
    if (f ? !(ls.work<float>().amg && ls.work<float>().amg->ready()) : !(ls.work<double>().amg && ls.work<double>().amg->ready())) return OPMGPU_EINVAL;
    if (into_level0) *into_level0 = f ? ls.work<float>().amg->pdamp0 : ls.work<double>().amg->pdamp0;
    if (below) *below = f ? ls.work<float>().amg->pdamp : ls.work<double>().amg->pdamp;
    return OPMGPU_OK;
}

int opmgpu_get_matbalscale(opmgpu_ctx* c, double* scale3)
{
    if (!c || !c->model || !scale3) return OPMGPU_EINVAL;
    for (int a = 0; a < 3; ++a) scale3[a] = c->model->prm.matbalscale[a];
    return OPMGPU_OK;
}

int opmgpu_set_solve_precision(opmgpu_ctx* c, int single_precision)
{
    if (!c || !c->model) return OPMGPU_EINVAL;
    c->model->assemble_single = single_precision != 0;
    return OPMGPU_OK;
}

int opmgpu_perf_props(opmgpu_ctx* c, double* out)
{
    if (!c || !c->model || !out) return OPMGPU_EINVAL;
    return guarded(c, [&]() { c->model->perf_props(out); return OPMGPU_OK; });
}

int opmgpu_add_well_terms(opmgpu_ctx* c, const double* resid_delta, int nblk, const int32_t* schur_rc, const double* schur_blocks)
{
    if (!c || !c->model || nblk < 0 || (nblk > 0 && (!schur_rc || !schur_blocks))) return OPMGPU_EINVAL;
    return guarded(c, [&]() { c->factored = false; c->cur_single = -1; return c->model->add_well_terms(resid_delta, nblk, schur_rc, schur_blocks); });
}

int opmgpu_add_well_rhs(opmgpu_ctx* c, const double* rhs_delta)
{
    if (!c || !c->model || !rhs_delta) return OPMGPU_EINVAL;
    return guarded(c, [&]() { c->model->add_well_rhs(rhs_delta); return OPMGPU_OK; });
}
int opmgpu_perf_dx(opmgpu_ctx* c, double* out)
{
    if (!c || !c->model || !out || !c->model->has_dx) return OPMGPU_EINVAL;
    return guarded(c, [&]() { c->model->perf_dx(out); return OPMGPU_OK; });
}

int opmgpu_convergence(opmgpu_ctx* c, double dt, double* B_avg3, double* CNV3, double* MB3, double* linf3, int* converged)
{
    if (!c || !c->model || !c->matrix_loaded) return OPMGPU_EINVAL;
    return guarded(c, [&]() {
        const int st = c->model->convergence(dt, B_avg3, CNV3, MB3, linf3, converged);
        if (st == OPMGPU_ENUMERICAL) c->err = "NaN or too large residual";
        return st;
    });
}

int opmgpu_solve(opmgpu_ctx* c, int single_precision, double* dx, int* iters, double* reduction)
{
    if (!c || !c->model || !c->matrix_loaded) return OPMGPU_EINVAL;
    return guarded(c, [&]() {
        SolveResult res;
        int st;
        {
            Timed t(c, PH_SOLVE);
            c->model->early_factor_pending = false;          // (no convergence check between the assembly and this solve: the solve starts the factorisation itself)
            if (single_precision) { c->ls->ensure_work<float>(); c->model->build_rhs<float>(); st = solve_loaded<float>(c, true, res); if (st == OPMGPU_OK || st == OPMGPU_ELINSOLVE) c->model->store_dx<float>(); }
            else { c->ls->ensure_work<double>(); c->model->build_rhs<double>(); st = solve_loaded<double>(c, true, res); if (st == OPMGPU_OK || st == OPMGPU_ELINSOLVE) c->model->store_dx<double>(); }
        }
        c->cur_single = single_precision ? 1 : 0;
        if (iters) *iters = res.iterations;
        if (reduction) *reduction = res.reduction;
        if (dx && (st == OPMGPU_OK || st == OPMGPU_ELINSOLVE)) c->model->dx_to_host(dx);
        return st;
    });
}

int opmgpu_update_state(opmgpu_ctx* c, const double* dx, double relax)
{
    if (!c || !c->model || !c->model->has_state) return OPMGPU_EINVAL;
    if (!dx && !c->model->has_dx) return fail(c, OPMGPU_EINVAL, "no resident Newton increment");
    return guarded(c, [&]() {
        Timed t(c, PH_UPDATE);
        c->model->update_state(dx, relax);
        return OPMGPU_OK;
    });
}

int opmgpu_set_device_wells(opmgpu_ctx* c, const opmgpu_wells* wells)
{
    if (!c || !c->model || !wells) return OPMGPU_EINVAL;
    return guarded(c, [&]() {
        c->matrix_loaded = false;
        // multi-GPU: a well lives on ONE rank (the reference hands the wells to loadBalance for the same reason,
        // RedistributeDataHandles.hpp:559-560): every perforated cell must be an owned cell of this rank
        if (c->comm && wells->nw > 0 && wells->well_connpos && wells->well_cells)
            for (int j = 0; j < wells->well_connpos[wells->nw]; ++j)
                if (wells->well_cells[j] >= c->model->n_owned_cells) return fail(c, OPMGPU_EINVAL, "multi-GPU: a well perforates a ghost cell; keep every well on one rank");
        c->ls->run_has_wells = true; c->ls->cs_for = nullptr;       // the coarse space of the pressure stage keeps one unknown per rank then
        const int st = c->model->set_device_wells(wells);
        if (st != OPMGPU_OK) return fail(c, st, "invalid well specification (missing array, cell out of range or perforated twice)");
        return st;
    });
}
int opmgpu_well_state_set(opmgpu_ctx* c, const double* bhp, const double* qs, const double* perf_press, const double* perf_rates)
{
    if (!c || !c->model) return OPMGPU_EINVAL;
    return guarded(c, [&]() { return c->model->well_state_set(bhp, qs, perf_press, perf_rates); });
}
int opmgpu_well_state_get(opmgpu_ctx* c, double* bhp, double* qs, double* perf_press, double* perf_rates)
{
    if (!c || !c->model) return OPMGPU_EINVAL;
    return guarded(c, [&]() { return c->model->well_state_get(bhp, qs, perf_press, perf_rates); });
}
int opmgpu_set_vfp_tables(opmgpu_ctx* c, int n, const opmgpu_vfp_table* tables)
{
    if (!c || !c->model) return OPMGPU_EINVAL;
    return guarded(c, [&]() {
        const int st = c->model->set_vfp_tables(n, tables);
        return st == OPMGPU_OK ? st : fail(c, st, "invalid VFP table (missing axis / data, empty axis, or more than 64 THP values)");
    });
}
int opmgpu_well_controls_set(opmgpu_ctx* c, const int32_t* current, const double* thp)
{
    if (!c || !c->model) return OPMGPU_EINVAL;
    return guarded(c, [&]() { return c->model->well_controls_set(current, thp); });
}
int opmgpu_well_controls_set_targets(opmgpu_ctx* c, const double* target, const double* distr)
{
    if (!c || !c->model || (!target && !distr)) return OPMGPU_EINVAL;
    return guarded(c, [&]() { return c->model->well_controls_set_targets(target, distr); });
}
int opmgpu_well_controls_get(opmgpu_ctx* c, int32_t* current, double* thp, int32_t* pre_its, int32_t* pre_conv)
{
    if (!c || !c->model) return OPMGPU_EINVAL;
    return guarded(c, [&]() { return c->model->well_controls_get(current, thp, pre_its, pre_conv); });
}
int opmgpu_perf_pvt(opmgpu_ctx* c, const double* pressure, double* out)
{
    if (!c || !c->model || !pressure || !out || !c->model->has_state) return OPMGPU_EINVAL;
    return guarded(c, [&]() { c->model->perf_pvt(pressure, out); return OPMGPU_OK; });
}
int opmgpu_average_b(opmgpu_ctx* c, double* B3)
{
    if (!c || !c->model || !B3 || !c->matrix_loaded) return OPMGPU_EINVAL;
    return guarded(c, [&]() { c->model->average_b(B3); return OPMGPU_OK; });
}
int opmgpu_well_convergence(opmgpu_ctx* c, double* flux3, double* ctrl)
{
    if (!c || !c->model) return OPMGPU_EINVAL;
    return guarded(c, [&]() {
        const int st = c->model->well_convergence(flux3, ctrl);
        // the statuses the reference's time stepper catches, with a text: a singular 4 x 4 well block is Dune::MatrixBlockError's case (the inverse in
        // eliminateVariable / solveWellEq), the other marks are NumericalIssue's (StandardWells_impl.hpp:742-748, NonlinearSolver_impl.hpp:165)
        if (st == OPMGPU_ESINGULAR) return fail(c, st, "singular well equation block (no open perforation can carry the well's flow)");
        if (st == OPMGPU_ENUMERICAL) return fail(c, st, "well model: non-finite or too large well residual, no consistent control, or the pre-solve gave up");
        return st;
    });
}

static int nonlinear_iteration_body(opmgpu_ctx* c, double dt, int iteration, int single_precision, const opmgpu_newton_ctl* ctl, int* converged,
                                    int* linear_iterations, double* linf3, double* relaxation, int* solved);

int opmgpu_nonlinear_iteration(opmgpu_ctx* c, double dt, int iteration, int single_precision, const opmgpu_newton_ctl* ctl, int* converged,
                               int* linear_iterations, double* linf3, double* relaxation)
{
    if (!c || !c->model || !ctl || !converged || iteration < 0) return OPMGPU_EINVAL;
    int solved = 0, lin = 0;
    if (!c->marks_on) return nonlinear_iteration_body(c, dt, iteration, single_precision, ctl, converged, linear_iterations ? linear_iterations : &lin, linf3, relaxation, &solved);
    if (hipSetDevice(c->device) != hipSuccess) return fail(c, OPMGPU_ENODEVICE, "hipSetDevice failed");
    if (c->marks.size() >= opmgpu_ctx::kMaxMarks) {      // the list is full: the call runs unmarked
        ++c->marks_dropped;
        return nonlinear_iteration_body(c, dt, iteration, single_precision, ctl, converged, linear_iterations ? linear_iterations : &lin, linf3, relaxation, &solved);
    }
    if (!c->mark_start) { c->mark_start = c->mark_event(); if (!c->mark_start) return OPMGPU_ENODEVICE; (void)hipEventRecord(c->mark_start, c->stream); }
    c->marks.emplace_back();
    c->cur_mark = &c->marks.back();
    if (!linear_iterations) linear_iterations = &lin;
    const int st = nonlinear_iteration_body(c, dt, iteration, single_precision, ctl, converged, linear_iterations, linf3, relaxation, &solved);
    opmgpu_ctx::IterMark& m = c->marks.back();
    m.solved = solved; m.lin = *linear_iterations; m.status = st;
    m.end = c->mark_event();
    if (m.end) (void)hipEventRecord(m.end, c->stream);
    c->cur_mark = nullptr;
    return st;
}

int opmgpu_iteration_marks(opmgpu_ctx* c, int enable)
{
    if (!c) return OPMGPU_EINVAL;
    if (hipSetDevice(c->device) != hipSuccess) return fail(c, OPMGPU_ENODEVICE, "hipSetDevice failed");
    c->marks_clear();
    c->marks_on = enable != 0;
    return OPMGPU_OK;
}

int opmgpu_iteration_marks_get(opmgpu_ctx* c, int max_calls, double* call_ms, int32_t* solved, int32_t* linear_iterations, double* phase_ms, int* n_calls)
{
    if (!c || !n_calls || max_calls < 0) return OPMGPU_EINVAL;
    if (hipSetDevice(c->device) != hipSuccess) return fail(c, OPMGPU_ENODEVICE, "hipSetDevice failed");
    const int n = std::min<int>(max_calls, int(c->marks.size()));
    *n_calls = int(c->marks.size() + c->marks_dropped);
    if (n > 0 && c->marks[n - 1].end && hipEventSynchronize(c->marks[n - 1].end) != hipSuccess) return fail(c, OPMGPU_ENODEVICE, "hipEventSynchronize failed");
    hipEvent_t prev = c->mark_start;
    for (int i = 0; i < n; ++i) {
        const opmgpu_ctx::IterMark& m = c->marks[i];
        float ms = 0.f;
        if (call_ms) { call_ms[i] = (prev && m.end && hipEventElapsedTime(&ms, prev, m.end) == hipSuccess) ? double(ms) : -1.0; }
        if (solved) solved[i] = m.solved;
        if (linear_iterations) linear_iterations[i] = m.lin;
        if (phase_ms) for (int ph = 0; ph < 3; ++ph) {
            float pm = 0.f;
            phase_ms[3 * i + ph] = (m.ph[ph][0] && m.ph[ph][1] && hipEventSynchronize(m.ph[ph][1]) == hipSuccess && hipEventElapsedTime(&pm, m.ph[ph][0], m.ph[ph][1]) == hipSuccess) ? double(pm) : 0.0;
        }
        if (m.end) prev = m.end;
    }
    return OPMGPU_OK;
}

static int nonlinear_iteration_body(opmgpu_ctx* c, double dt, int iteration, int single_precision, const opmgpu_newton_ctl* ctl, int* converged,
                                    int* linear_iterations, double* linf3, double* relaxation, int* solved)
{
    if (iteration == 0) { c->norm_history.clear(); c->relaxation = 1.0; }
    if (int(c->norm_history.size()) != iteration) return fail(c, OPMGPU_EINVAL, "opmgpu_nonlinear_iteration: iterations of a time step must be consecutive from 0");
    *converged = 0;
    if (linear_iterations) *linear_iterations = 0;
    int st = opmgpu_set_solve_precision(c, single_precision);
    if (st == OPMGPU_OK) st = opmgpu_assemble(c, dt, iteration == 0 ? 1 : 0, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (st != OPMGPU_OK) return st;
    double B[3], CNV[3], MB[3], linf[3];
    int conv = 0;
    st = opmgpu_convergence(c, dt, B, CNV, MB, linf, &conv);
    if (st != OPMGPU_OK) return st;
    if (c->model->has_device_wells()) {                    // getWellConvergence (:1769-1779)
        double flux[3], ctrl = 0.0;
        st = opmgpu_well_convergence(c, flux, &ctrl);
        if (st != OPMGPU_OK) return st;
        bool ok = ctrl < c->prm.tolerance_well_control;
        for (int a = 0; a < 3; ++a) ok = ok && (B[a] * flux[a] < c->prm.tolerance_wells);
        conv = conv && ok;
    }
    c->norm_history.push_back({ linf[0], linf[1], linf[2] });
    if (linf3) for (int a = 0; a < 3; ++a) linf3[a] = linf[a];
    *converged = conv;
    if (!conv || iteration < ctl->min_iter) {
        int its = 0;
        *solved = 1;
        st = opmgpu_solve(c, single_precision, nullptr, &its, nullptr);
        if (linear_iterations) *linear_iterations = its;
        if (st != OPMGPU_OK) return st;
        if (ctl->use_update_stabilization) {
            // the oscillation rule of NonlinearSolver_impl.hpp:221-257 (only the three mass-balance norms take part): a phase swings when its
            // norm is back within relax_rel_tol of its value two iterations ago but not of last iteration's; two swinging phases oscillate
            bool oscillate = false;
            if (iteration >= 2) {
                const auto &now = c->norm_history[iteration], &last = c->norm_history[iteration - 1], &before = c->norm_history[iteration - 2];
                int swinging = 0;
                for (int ph = 0; ph < 3; ++ph) {
                    const double to_before = std::fabs((now[ph] - before[ph]) / now[ph]), to_last = std::fabs((now[ph] - last[ph]) / now[ph]);
                    if (to_before < ctl->relax_rel_tol && ctl->relax_rel_tol < to_last) ++swinging;
                }
                oscillate = swinging >= 2;
            }
            if (oscillate) c->relaxation = std::max(c->relaxation - ctl->relax_increment, ctl->relax_max);
            st = opmgpu_stabilize_update(c, ctl->relax_type, c->relaxation);
            if (st != OPMGPU_OK) return st;
        }
        st = opmgpu_update_state(c, nullptr, 1.0);
    }
    if (relaxation) *relaxation = c->relaxation;
    return st;
}

int opmgpu_save_state(opmgpu_ctx* c)
{
    if (!c || !c->model || !c->model->has_state) return OPMGPU_EINVAL;
    return guarded(c, [&]() { c->model->save_state(); return OPMGPU_OK; });
}
int opmgpu_restore_state(opmgpu_ctx* c)
{
    if (!c || !c->model) return OPMGPU_EINVAL;
    if (!c->model->has_saved) return fail(c, OPMGPU_EINVAL, "no saved state (opmgpu_save_state; a well re-plan discards it)");
    return guarded(c, [&]() { c->model->restore_state(); return OPMGPU_OK; });
}
int opmgpu_compute_fluid_in_place(opmgpu_ctx* c, const int32_t* fipnum, int nregions, double* fip_cells, double* values)
{
    if (!c || !c->model || !values || nregions < 1) return OPMGPU_EINVAL;
    if (!c->model->has_state) return fail(c, OPMGPU_EINVAL, "no reservoir state on the device");
    if (fipnum) for (int i = 0; i < c->model->nc; ++i) if (fipnum[i] < 0 || fipnum[i] > nregions) return fail(c, OPMGPU_EINVAL, "fipnum outside [0, nregions]");
    return guarded(c, [&]() { c->model->fluid_in_place(fipnum, nregions, fip_cells, values); return int(OPMGPU_OK); });
}

int opmgpu_region_state_sums(opmgpu_ctx* c, const int32_t* region, int nregions, double* sums)
{
    if (!c || !c->model || !sums || nregions < 1) return OPMGPU_EINVAL;
    if (!c->model->has_state) return fail(c, OPMGPU_EINVAL, "no reservoir state on the device");
    return guarded(c, [&]() { c->model->region_state_sums(region, nregions, sums); return int(OPMGPU_OK); });
}
int opmgpu_voidage_coefficients(opmgpu_ctx* c, int n, const double* p, const double* rs, const double* rv, const int32_t* pvt_region, double* coeff)
{
    if (!c || !c->model || n < 0 || (n > 0 && (!p || !rs || !rv || !coeff))) return OPMGPU_EINVAL;
    return guarded(c, [&]() { c->model->voidage_coefficients(n, p, rs, rv, pvt_region, coeff); return int(OPMGPU_OK); });
}

int opmgpu_relative_change(opmgpu_ctx* c, double* value)
{
    if (!c || !c->model || !value) return OPMGPU_EINVAL;
    if (!c->model->has_saved) return fail(c, OPMGPU_EINVAL, "no saved state (opmgpu_save_state; a well re-plan discards it)");
    return guarded(c, [&]() { *value = c->model->relative_change(); return OPMGPU_OK; });
}

int opmgpu_set_sat_oil_max(opmgpu_ctx* c, const double* so_max)
{
    if (!c || !c->model || !so_max) return OPMGPU_EINVAL;
    return guarded(c, [&]() { c->model->set_sat_oil_max(so_max); return OPMGPU_OK; });
}
int opmgpu_update_sat_oil_max(opmgpu_ctx* c)
{
    if (!c || !c->model || !c->model->has_state) return OPMGPU_EINVAL;
    return guarded(c, [&]() { c->model->update_sat_oil_max(); return OPMGPU_OK; });
}
int opmgpu_get_sat_oil_max(opmgpu_ctx* c, double* so_max)
{
    if (!c || !c->model || !so_max) return OPMGPU_EINVAL;
    return guarded(c, [&]() { c->model->get_sat_oil_max(so_max); return OPMGPU_OK; });
}

int opmgpu_update_hysteresis(opmgpu_ctx* c)
{
    if (!c || !c->model) return OPMGPU_EINVAL;
    return guarded(c, [&]() { return c->model->update_hysteresis(); });
}
int opmgpu_set_hysteresis(opmgpu_ctx* c, const double* mdc_ow, const double* mdc_go)
{
    if (!c || !c->model) return OPMGPU_EINVAL;
    return guarded(c, [&]() { return c->model->set_hysteresis(mdc_ow, mdc_go); });
}
int opmgpu_get_hysteresis(opmgpu_ctx* c, double* mdc_ow, double* mdc_go, double* d_ow, double* d_go)
{
    if (!c || !c->model) return OPMGPU_EINVAL;
    return guarded(c, [&]() { return c->model->get_hysteresis(mdc_ow, mdc_go, d_ow, d_go); });
}

int opmgpu_stabilize_update(opmgpu_ctx* c, int relax_type, double omega)
{
    if (!c || !c->model || (relax_type != OPMGPU_RELAX_DAMPEN && relax_type != OPMGPU_RELAX_SOR)) return OPMGPU_EINVAL;
    if (!c->model->has_dx) return fail(c, OPMGPU_EINVAL, "no resident Newton increment");
    return guarded(c, [&]() {
        c->model->stabilize_update(relax_type, omega);
        return OPMGPU_OK;
    });
}

int opmgpu_load_bsr(opmgpu_ctx* c, int nb, const int32_t* rowptr, const int32_t* col, const double* val9, int single_precision)
{
    if (!c || nb <= 0 || !rowptr || !col || !val9) return OPMGPU_EINVAL;
    if (c->model) return fail(c, OPMGPU_EINVAL, "opmgpu_load_bsr needs a solver-only context (opmgpu_create_solver)");
    return guarded(c, [&]() {
        const int st = c->ls->set_pattern(nb, rowptr, col, c->prm.ilu_ordering);
        if (st != OPMGPU_OK) return fail(c, st, "invalid BSR pattern");
        c->ls->load_host_bsr(val9);
        if (single_precision) c->ls->prepare<float>(true); else c->ls->prepare<double>(true);
        c->cur_single = single_precision ? 1 : 0;
        c->matrix_loaded = true; c->factored = false; c->ls->ref_transformed = false;
        return int(OPMGPU_OK);
    });
}

int opmgpu_solve_bsr(opmgpu_ctx* c, int nb, const int32_t* rowptr, const int32_t* col, const double* val9, const double* rhs3,
                     int single_precision, double* x3, int* iters, double* reduction)
{
    if (!c || !rhs3 || !x3) return OPMGPU_EINVAL;
    const int st0 = opmgpu_load_bsr(c, nb, rowptr, col, val9, single_precision);
    if (st0 != OPMGPU_OK) return st0;
    return guarded(c, [&]() {
        SolveResult res; int st;
        {
            Timed t(c, PH_SOLVE);
            if (single_precision) { c->ls->vec_from_host<float>(rhs3, VEC_BLOCK_INTERLEAVED, c->ls->work<float>().b.p); st = solve_loaded<float>(c, false, res); }
            else { c->ls->vec_from_host<double>(rhs3, VEC_BLOCK_INTERLEAVED, c->ls->work<double>().b.p); st = solve_loaded<double>(c, false, res); }
        }
        if (iters) *iters = res.iterations;
        if (reduction) *reduction = res.reduction;
        if (st == OPMGPU_OK || st == OPMGPU_ELINSOLVE) {
            if (single_precision) c->ls->vec_to_host<float>(c->ls->work<float>().x.p, VEC_BLOCK_INTERLEAVED, x3);
            else c->ls->vec_to_host<double>(c->ls->work<double>().x.p, VEC_BLOCK_INTERLEAVED, x3);
        }
        return st;
    });
}

// make sure the matrix exists in the precision the kernel-level entry points will use
static int ensure_prepared(opmgpu_ctx* c)
{
    if (!c->matrix_loaded) return fail(c, OPMGPU_EINVAL, "no matrix loaded");
    if (c->cur_single < 0) { c->ls->prepare<double>(true); c->cur_single = 0; }
    return OPMGPU_OK;
}

int opmgpu_spmv(opmgpu_ctx* c, const double* x3, double* y3)
{
    if (!c || !x3 || !y3) return OPMGPU_EINVAL;
    return guarded(c, [&]() {
        const int st = ensure_prepared(c); if (st) return st;
        LinSolver& ls = *c->ls;
        if (c->cur_single) { auto& w = ls.work<float>(); ls.vec_from_host<float>(x3, VEC_BLOCK_INTERLEAVED, w.p.p); ls.spmv<float>(w.p.p, w.v.p); ls.vec_to_host<float>(w.v.p, VEC_BLOCK_INTERLEAVED, y3); }
        else { auto& w = ls.work<double>(); ls.vec_from_host<double>(x3, VEC_BLOCK_INTERLEAVED, w.p.p); ls.spmv<double>(w.p.p, w.v.p); ls.vec_to_host<double>(w.v.p, VEC_BLOCK_INTERLEAVED, y3); }
        return int(OPMGPU_OK);
    });
}

int opmgpu_ilu0_factor(opmgpu_ctx* c)
{
    if (!c) return OPMGPU_EINVAL;
    return guarded(c, [&]() {
        const int st = ensure_prepared(c); if (st) return st;
        const int fill = fill_level_of(c->prm);
        if (fill < 0 || fill > 8) return fail(c, OPMGPU_EINVAL, "cpr_ilu_n / ilu_fillin_level must be in 0..8");
        c->ls->fill_level = fill;
        const int s2 = c->cur_single ? c->ls->factor<float>() : c->ls->factor<double>();
        c->factored = (s2 == OPMGPU_OK);
        return s2 == OPMGPU_OK ? int(OPMGPU_OK) : fail(c, s2, "singular diagonal block in ILU0");
    });
}

int opmgpu_ilu0_apply(opmgpu_ctx* c, const double* d3, double* v3)
{
    if (!c || !d3 || !v3) return OPMGPU_EINVAL;
    if (!c->factored) return fail(c, OPMGPU_EINVAL, "call opmgpu_ilu0_factor first");
    return guarded(c, [&]() {
        LinSolver& ls = *c->ls;
        if (c->cur_single) { auto& w = ls.work<float>(); ls.vec_from_host<float>(d3, VEC_BLOCK_INTERLEAVED, w.p.p); ls.ilu_apply<float>(w.p.p, w.y.p, c->prm.ilu_relaxation, nullptr); ls.vec_to_host<float>(w.y.p, VEC_BLOCK_INTERLEAVED, v3); }
        else { auto& w = ls.work<double>(); ls.vec_from_host<double>(d3, VEC_BLOCK_INTERLEAVED, w.p.p); ls.ilu_apply<double>(w.p.p, w.y.p, c->prm.ilu_relaxation, nullptr); ls.vec_to_host<double>(w.y.p, VEC_BLOCK_INTERLEAVED, v3); }
        return int(OPMGPU_OK);
    });
}

int opmgpu_point_ilu_apply(opmgpu_ctx* c, const double* d3, double* v3, double relax)
{
    if (!c || !d3 || !v3) return OPMGPU_EINVAL;
    if (!c->ls || !c->ls->pilu.built || c->ls->pilu.stale) return fail(c, OPMGPU_EINVAL, "solve once with cpr_reference_transform = 2 first");
    return guarded(c, [&]() {
        LinSolver& ls = *c->ls;
        auto& w = ls.work<double>();
        ls.vec_from_host<double>(d3, VEC_BLOCK_INTERLEAVED, w.p.p);
        ls.point_ilu_apply(w.p.p, w.y.p, relax);
        ls.vec_to_host<double>(w.y.p, VEC_BLOCK_INTERLEAVED, v3);
        return int(OPMGPU_OK);
    });
}

int opmgpu_ilu0_get(opmgpu_ctx* c, double* val9)
{
    if (!c || !val9) return OPMGPU_EINVAL;
    if (!c->factored) return fail(c, OPMGPU_EINVAL, "call opmgpu_ilu0_factor first");
    if (c->ls->fill_level > 0) return fail(c, OPMGPU_EINVAL, "opmgpu_ilu0_get returns the ILU0 factors on the matrix's pattern: not with cpr_ilu_n / ilu_fillin_level > 0");
    return guarded(c, [&]() { if (c->cur_single) c->ls->get_lu_bsr<float>(val9); else c->ls->get_lu_bsr<double>(val9); return OPMGPU_OK; });
}

int opmgpu_get_ordering(opmgpu_ctx* c, int32_t* position, int32_t* level, int32_t* nlevels)
{
    if (!c || !c->ls->has_pattern()) return OPMGPU_EINVAL;
    const Plan& P = c->ls->plan;
    for (int i = 0; i < P.nb; ++i) { if (position) position[i] = P.pos[i]; if (level) level[i] = P.level[P.pos[i]]; }
    if (nlevels) *nlevels = P.nlevels;
    return OPMGPU_OK;
}

int opmgpu_get_cpr_weights(opmgpu_ctx* c, double* w)
{
    if (!c || !w || !c->ls->has_pattern()) return OPMGPU_EINVAL;
    return guarded(c, [&]() {
        LinSolver& ls = *c->ls;
        if (c->cur_single == 1) { if (!ls.work<float>().cprw.p) return fail(c, OPMGPU_EINVAL, "no CPR solve yet"); ls.vec_to_host<float>(ls.work<float>().cprw.p, VEC_EQUATION_MAJOR, w); }
        else { if (!ls.work<double>().cprw.p) return fail(c, OPMGPU_EINVAL, "no CPR solve yet"); ls.vec_to_host<double>(ls.work<double>().cprw.p, VEC_EQUATION_MAJOR, w); }
        return int(OPMGPU_OK);
    });
}

int opmgpu_get_residual(opmgpu_ctx* c, double* r)
{
    if (!c || !c->model || !r) return OPMGPU_EINVAL;
    return guarded(c, [&]() { c->model->get_residual(r); return OPMGPU_OK; });
}
int opmgpu_get_jacobian_nnzb(opmgpu_ctx* c, int32_t* nnzb)
{
    if (!c || !nnzb || !c->ls->has_pattern()) return OPMGPU_EINVAL;
    *nnzb = c->ls->plan.nnzb;
    return OPMGPU_OK;
}
int opmgpu_get_jacobian_bsr(opmgpu_ctx* c, int32_t* rowptr, int32_t* col, double* val9)
{
    if (!c || !c->ls->has_pattern()) return OPMGPU_EINVAL;
    return guarded(c, [&]() {
        const Plan& P = c->ls->plan;
        if (rowptr) std::memcpy(rowptr, P.rowptr.data(), sizeof(int32_t) * (P.nb + 1));
        if (col) std::memcpy(col, P.col.data(), sizeof(int32_t) * P.nnzb);
        if (val9) { c->ls->widen_matrix(); c->ls->get_matrix_bsr(c->ls->matrix_d(), val9); }
        return OPMGPU_OK;
    });
}

int opmgpu_time_kernel(opmgpu_ctx* c, int kernel, int reps, double* ms_per_launch)
{
    if (!c || reps < 1 || !ms_per_launch) return OPMGPU_EINVAL;
    return guarded(c, [&]() {
        if (kernel == OPMGPU_K_ASSEMBLE || kernel == OPMGPU_K_PROPS) {
            if (!c->model || !c->model->has_state) return fail(c, OPMGPU_EINVAL, "no model / state");
            *ms_per_launch = c->model->time_assemble(reps, kernel == OPMGPU_K_PROPS);
            c->factored = false; c->cur_single = -1; c->matrix_loaded = true; c->ls->ref_transformed = false;
            return int(OPMGPU_OK);
        }
        const int st = ensure_prepared(c); if (st) return st;
        if (c->cur_single) c->ls->ensure_work<float>(); else c->ls->ensure_work<double>();
        if ((kernel == OPMGPU_K_ILU_APPLY) && !c->factored) return fail(c, OPMGPU_EINVAL, "call opmgpu_ilu0_factor first");
        *ms_per_launch = c->ls->time_kernel(kernel, reps, c->cur_single);
        return int(OPMGPU_OK);
    });
}

int opmgpu_kernel_timing(opmgpu_ctx* c, int enable)
{
    if (!c) return OPMGPU_EINVAL;
    return guarded(c, [&]() { c->ls->kt.reset(); c->ls->kt.on = enable != 0; return OPMGPU_OK; });
}
int opmgpu_kernel_timing_get(opmgpu_ctx* c, double* total_ms, int64_t* launches)
{
    if (!c || !total_ms) return OPMGPU_EINVAL;
    return guarded(c, [&]() {
        KernelTimers& kt = c->ls->kt;
        kt.collect();
        for (int i = 0; i < KT_COUNT; ++i) { total_ms[i] = kt.total_ms[i]; if (launches) launches[i] = kt.count[i]; }
        return OPMGPU_OK;
    });
}

int opmgpu_last_timings(opmgpu_ctx* c, double* assemble_ms, double* solve_ms, double* update_ms)
{
    if (!c) return OPMGPU_EINVAL;
    for (int ph = 0; ph < 3; ++ph) {
        if (!c->ev_pending[ph]) continue;
        float ms = 0.f;
        if (hipEventSynchronize(c->evp[ph][1]) == hipSuccess && hipEventElapsedTime(&ms, c->evp[ph][0], c->evp[ph][1]) == hipSuccess) c->t_phase[ph] = ms;
        c->ev_pending[ph] = false;
    }
    if (assemble_ms) *assemble_ms = c->t_phase[PH_ASSEMBLE];
    if (solve_ms) *solve_ms = c->t_phase[PH_SOLVE];
    if (update_ms) *update_ms = c->t_phase[PH_UPDATE];
    return OPMGPU_OK;
}

int opmgpu_comm_unique_id(uint8_t* id)
{
    if (!id) return OPMGPU_EINVAL;
    try { return RcclComm::unique_id(id); } catch (...) { return OPMGPU_ECOMM; }
}

static int comm_init_common(opmgpu_ctx* c, int rank, int nranks, const uint8_t* id, const opmgpu_transport* transport, int32_t n_owned, int n_neigh,
                            const int32_t* neigh_rank, const int32_t* send_ptr, const int32_t* send_cells, const int32_t* recv_ptr, const int32_t* recv_cells)
{
    if (!c || !c->model || (!id && !transport) || nranks < 1 || rank < 0 || rank >= nranks || n_owned <= 0 || n_owned > c->model->nc || n_neigh < 0) return OPMGPU_EINVAL;
    if (n_neigh > 0 && (!neigh_rank || !send_ptr || !send_cells || !recv_ptr || !recv_cells)) return OPMGPU_EINVAL;
    return guarded(c, [&]() {
        std::unique_ptr<RcclComm> cm(new RcclComm());
        static const int32_t zero2[2] = { 0, 0 };
        const int st = cm->init(rank, nranks, id, transport, n_owned, c->model->nc, n_neigh, neigh_rank, n_neigh ? send_ptr : zero2, send_cells,
                                n_neigh ? recv_ptr : zero2, recv_cells);
        if (st != OPMGPU_OK) return fail(c, st, transport ? "invalid transport / neighbour lists" : "RCCL communicator initialisation failed");
        cm->rebuild(c->model->plan(), c->stream);
        c->comm = std::move(cm);
        c->model->attach_comm(c->comm.get(), n_owned);
        c->matrix_loaded = false;
        return int(OPMGPU_OK);
    });
}

int opmgpu_comm_init(opmgpu_ctx* c, int rank, int nranks, const uint8_t* id, int32_t n_owned, int n_neigh, const int32_t* neigh_rank,
                     const int32_t* send_ptr, const int32_t* send_cells, const int32_t* recv_ptr, const int32_t* recv_cells)
{
    if (!id) return OPMGPU_EINVAL;
    return comm_init_common(c, rank, nranks, id, nullptr, n_owned, n_neigh, neigh_rank, send_ptr, send_cells, recv_ptr, recv_cells);
}

int opmgpu_comm_init_transport(opmgpu_ctx* c, int rank, int nranks, const opmgpu_transport* transport, int32_t n_owned, int n_neigh,
                               const int32_t* neigh_rank, const int32_t* send_ptr, const int32_t* send_cells, const int32_t* recv_ptr, const int32_t* recv_cells)
{
    if (!transport) return OPMGPU_EINVAL;
    return comm_init_common(c, rank, nranks, nullptr, transport, n_owned, n_neigh, neigh_rank, send_ptr, send_cells, recv_ptr, recv_cells);
}

int opmgpu_comm_set_coarse_blocks(opmgpu_ctx* c, int m, const int32_t* block_of_owned_cell)
{
    if (!c || !c->comm) return OPMGPU_EINVAL;
    const int st = c->comm->set_coarse_blocks(m, block_of_owned_cell);
    if (st != OPMGPU_OK) return fail(c, st, "opmgpu_comm_set_coarse_blocks: 0 <= block < m <= 8 for every owned cell");
    c->ls->cs_for = nullptr;           // the subdomain map is rebuilt (collectively) at the next CPR solve
    return OPMGPU_OK;
}

int opmgpu_plan_ordering(int nb, const int32_t* rowptr, const int32_t* col, int ordering, int32_t* position, int32_t* level, int32_t* nlevels)
{
    Plan P;
    const int st = build_plan(nb, rowptr, col, ordering, P);
    if (st != OPMGPU_OK) return st;
    for (int i = 0; i < nb; ++i) { if (position) position[i] = P.pos[i]; if (level) level[i] = P.level[P.pos[i]]; }
    if (nlevels) *nlevels = P.nlevels;
    return OPMGPU_OK;
}

} // extern "C"
