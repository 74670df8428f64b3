/*
 * opmgpu.h -- C ABI of libopmgpu.so: the MI355X (gfx950) fully-implicit black-oil
 * Newton step that drops in behind flow_legacy's BlackoilModel /
 * NewtonIterationBlackoilInterface.
 *
 * Every entry point names the reference interface it replaces (paths relative to
 * the OPM/opm-simulators-legacy tree).  Conventions:
 *   - plain C linkage, plain pointers + sizes, no C++/torch types cross the boundary;
 *   - every call returns an int status: OPMGPU_OK (0) or an OPMGPU_E* code that the
 *     C++ shim maps to the exception the reference would throw;
 *   - the caller owns all host buffers, the library owns all device memory behind the
 *     opaque handle; a handle is bound to ONE GPU and is not thread-safe;
 *   - there is NO CPU fallback: without a usable HIP device every compute entry
 *     point returns OPMGPU_ENODEVICE;
 *   - cell-indexed host arrays are in the CALLER's (natural) cell order, vectors of
 *     unknowns are equation-major [all p | all sw | all xvar] exactly like the
 *     reference's dx (BlackoilModelBase_impl.hpp:1162-1175);
 *   - sign convention of the Newton update: x_new = x_old - dx (ibid. :1177-1183).
 */
#ifndef OPMGPU_H
#define OPMGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes -> reference exceptions (AdaptiveTimeStepping_impl.hpp:244-281) ---- */
enum {
    OPMGPU_OK = 0,
    OPMGPU_EINVAL = 1,            /* bad argument                      -> std::logic_error      */
    OPMGPU_ENODEVICE = 2,         /* no HIP device / HIP runtime error -> std::runtime_error    */
    OPMGPU_ENUMERICAL = 3,        /* NaN / too large residual          -> Opm::NumericalIssue   (BlackoilModelBase_impl.hpp:1828-1854) */
    OPMGPU_ELINSOLVE = 4,         /* linear solver did not converge    -> Opm::LinearSolverProblem (ISTLSolver.hpp:358-368) */
    OPMGPU_EBREAKDOWN = 5,        /* BiCGStab breakdown (rho/omega/h)  -> Dune::ISTLError       */
    OPMGPU_ESINGULAR = 6,         /* singular diagonal block in ILU0   -> Dune::MatrixBlockError */
    OPMGPU_ENOMEM = 7,
    OPMGPU_ECOMM = 8              /* RCCL failure                                               */
};

/* HydroCarbonState, opm/core/simulator/BlackoilState.hpp:33-37 */
enum { OPMGPU_HC_GAS_ONLY = 0, OPMGPU_HC_GAS_AND_OIL = 1, OPMGPU_HC_OIL_ONLY = 2 };

/* ILU0 elimination order.  NATURAL reproduces serial dune-istl bilu0 in the caller's
 * row order (level-scheduled on the device); MULTICOLOR is the reference's own
 * ilu_redblack idea (ISTLSolver.hpp:204-209) generalised to greedy colouring. */
enum { OPMGPU_ORDER_NATURAL = 0, OPMGPU_ORDER_MULTICOLOR = 1 };

typedef struct opmgpu_ctx opmgpu_ctx;

/* Static grid data = what the reference pulls from UnstructuredGrid + DerivedGeology
 * (GeoProps.hpp:84-195) + HelperOps (AutoDiffHelpers.hpp:44-174).  Connections are the
 * interior faces in grid face order followed by the NNCs; conn_cells[2*f+0] is the cell
 * with ngrad coefficient +1 (c1), conn_cells[2*f+1] the one with -1 (c2). */
typedef struct opmgpu_grid {
    int32_t        nc;          /* number of (active) cells                                     */
    int32_t        nconn;       /* interior faces + NNCs                                        */
    const int32_t* conn_cells;  /* [nconn*2]                                                    */
    const double*  trans;       /* [nconn]  transmissibility, SI                                */
    const double*  pv;          /* [nc]     pore volume (geo_.poreVolume())                     */
    const double*  z;           /* [nc]     cell centroid depth (geo_.z())                      */
    double         gravity;     /* geo_.gravity()[2]                                            */
    const double*  thpres;      /* [nconn] threshold pressures by connection, or NULL           */
    const int32_t* pvtnum;      /* [nc] 0-based PVT region (cellPvtRegionIdx_), or NULL (=0)    */
    const int32_t* satnum;      /* [nc] 0-based saturation region, or NULL (=0)                 */
    /* ENDSCALE (two-point saturation scaling, SCALECRS NO): per-cell scaled end points as opm-material's
     * EclEpsScalingPointsInfo holds them (materialLawParams(cell), SaturationPropsFromDeck.cpp:91-92).  Either all
     * eight are given or all are NULL (no end-point scaling).  Order: SWL SWCR SWU SOWCR SGL SGCR SGU SOGCR. */
    const double*  eps[8];
    /* SCALECRS YES: three-point horizontal scaling of the relative-permeability curves (EclEpsScalingPoints with
     * enableThreePointKrSatScaling): the critical saturation of the displacing phase is a third fixed point -- krw [SWCR,
     * 1-SOWCR-SGL, SWU], krow (in Sw) [SWL+SGL, SWCR+SGL, 1-SOWCR-SGL], krg [SGCR, 1-SOGCR-SWL, SGU], krog (in So) [SOGCR,
     * 1-SGCR-SWL, 1-SWL-SGL]; capillary pressures stay two-point.  Needs eps[]. */
    int32_t        scalecrs;
    /* vertical scaling (ENDSCALE with KRW / KRO / KRG / PCW / PCG arrays): per-cell maximum of the curve; the table value is
     * multiplied by (cell maximum / table maximum) (EclEpsTwoPhaseLaw::unscaledToScaledKrw_ etc.).  Any pointer may be NULL. */
    const double*  eps_v[5];            /* KRW, KRO, KRG, PCW, PCG */
    /* Relative-permeability hysteresis (SATOPTS HYSTER, EHYSTR item 2 = 0: Carlson's model for the non-wetting phases -- oil in the
     * oil-water system, gas in the gas-oil system -- drainage curves for the wetting phases, EHYSTR item 5 = KR: no capillary
     * pressure hysteresis; EclHysteresisTwoPhaseLaw, reached from SaturationPropsFromDeck.cpp:74-204 / updateSatHyst :206-222).
     * imbnum: 0-based saturation region of every cell's IMBIBITION curves (IMBNUM), NULL = no hysteresis.  ieps: scaled end points
     * of the imbibition curves (ISWL ISWCR ISWU ISOWCR ISGL ISGCR ISGU ISOGCR), all eight or all NULL (= the drainage ones, eps[]).
     * The history (minimum wetting saturation seen by each two-phase system) lives on the device: opmgpu_update_hysteresis. */
    const int32_t* imbnum;
    const double*  ieps[8];
} opmgpu_grid;

/* Fluid tables, already converted to SI and pre-processed the way opm-material stores
 * them (call sites BlackoilPropsAdFromDeck.cpp:264-738, SaturationPropsFromDeck.cpp:74-204).
 * All "ptr" arrays are CSR-style offsets with one extra trailing entry. */
typedef struct opmgpu_tables {
    int32_t n_pvt_regions, n_sat_regions;
    int32_t has_disgas, has_vapoil;            /* DISGAS / VAPOIL                               */
    const double*  surface_density;            /* [n_pvt][3]  water, oil, gas                   */
    const double*  pvtw;                       /* [n_pvt][5]  pref, Bw_ref, Cw, mu_ref, Cv      */
    /* oil: saturated curve nodes (PVTO rows; PVDO = nodes with rs == 0, has_disgas == 0)       */
    const int32_t* oil_node_ptr;               /* [n_pvt+1]                                     */
    const double  *oil_rs, *oil_psat, *oil_invb_sat, *oil_invbmu_sat;   /* per node             */
    const int32_t* oil_col_ptr;                /* [n_nodes+1] undersaturated column of node     */
    const double  *oil_col_p, *oil_col_invb, *oil_col_invbmu;           /* per column sample    */
    /* gas: nodes keyed by gas pressure (PVTG; PVDG = nodes with rv == 0, has_vapoil == 0)      */
    const int32_t* gas_node_ptr;               /* [n_pvt+1]                                     */
    const double  *gas_pg, *gas_rvsat, *gas_invb_sat, *gas_invbmu_sat;  /* per node             */
    const int32_t* gas_col_ptr;                /* [n_nodes+1] column over Rv, ascending         */
    const double  *gas_col_rv, *gas_col_invb, *gas_col_invbmu;
    /* SWOF / SGOF per saturation region                                                        */
    const int32_t* swof_ptr;                   /* [n_sat+1]                                     */
    const double  *swof_sw, *swof_krw, *swof_krow, *swof_pcow;
    const int32_t* sgof_ptr;                   /* [n_sat+1]                                     */
    const double  *sgof_sg, *sgof_krg, *sgof_krog, *sgof_pcgo;
    /* ROCK (RockCompressibility.cpp:86-125, quadratic form)                                    */
    double rock_pref, rock_comp;
    /* VAPPARS (BlackoilPropsAdFromDeck.cpp:168-172, applyVap :1052-1078): rvSat *= (so/soMax)^vap1,
     * rsSat *= (so/soMax)^vap2 where soMax > 0.01 and so < soMax; 0 = off.  soMax per cell: see
     * opmgpu_set_sat_oil_max / opmgpu_update_sat_oil_max.                                       */
    double vap1, vap2;
    /* ROCKTAB (RockCompressibility.cpp:50-62, :86-125), single region: pore-volume and
     * transmissibility multipliers tabulated against pressure, piecewise linear with linear
     * extrapolation; rocktab_n == 0 = use the quadratic ROCK form above (trans multiplier 1).    */
    int32_t rocktab_n;
    const double *rocktab_p, *rocktab_pvmult, *rocktab_transmult;
} opmgpu_tables;

/* Newton + linear-solver knobs: BlackoilModelParameters.cpp:76-102, BlackoilModelBase_impl.hpp:139,
 * FlowLinearSolverParameters fields read at ISTLSolver.hpp:142-270. */
typedef struct opmgpu_params {
    double dp_max_rel;              /* 0.3    */
    double ds_max;                  /* 0.2    */
    double dr_max_rel;              /* 1e9    */
    double max_residual_allowed;    /* 1e7    */
    double tolerance_mb;            /* 1e-5   */
    double tolerance_cnv;           /* 1e-2   */
    double matbalscale[3];          /* {1.1169, 1.0031, 0.0031}                                 */
    double linear_solver_reduction; /* 1e-2   */
    int32_t linear_solver_maxiter;  /* 150 (solver_approach=interleaved); the reference's CPR plug-in defaults to 50 (NewtonIterationBlackoilCPR.cpp:63) */
    double ilu_relaxation;          /* 0.9    */
    int32_t ilu_ordering;           /* OPMGPU_ORDER_*                                           */
    int32_t ignore_convergence_failure; /* 0  */
    int32_t use_cpr;                /* 0 = block-ILU0 (solver_approach=interleaved, the default, FlowMain.hpp:806-830);
                                       1 = CPR: AMG V-cycle on the pressure system + block-ILU0
                                           (solver_approach=cpr, NewtonIterationBlackoilCPR.cpp:79-185)     */
    int32_t newton_use_gmres;       /* 0 = BiCGStab; 1 = Dune::RestartedGMResSolver (ISTLSolver.hpp:257-264): left-preconditioned
                                       restarted GMRES, modified Gram-Schmidt (also decomposed);
                                       2 = flexible (right-preconditioned) GMRES, NOT a reference solver: stops on the true residual like
                                           the default BiCGStab and saves the application M^-1 b, but needs more columns for the same
                                           reduction (measured 4.95 against 3.65 on the bench deck: slower) */
    int32_t linear_solver_restart;  /* 40     */
    /* well model (device wells): BlackoilModelParameters.cpp:78-79, :96, :85 */
    int32_t solve_welleq_initially; /* 1: explicit well pre-solve at the initial assembly (BlackoilModelBase_impl.hpp:827-829) */
    double tolerance_wells;         /* 1e-4   */
    double tolerance_well_control;  /* 1e-7   */
    double dbhp_max_rel;            /* 1.0    */
    int32_t update_equations_scaling; /* 0; 1: matbalscale[a] = mean over the cells of 1 / b_a of the state being assembled
                                         (BlackoilModelBase::updateEquationsScaling, BlackoilModelBase_impl.hpp:909, :919-947); the values in
                                         use are read back with opmgpu_get_matbalscale */
    int32_t gmres_verify_residual;  /* 0 = Dune::RestartedGMResSolver as it is (ISTLSolver.hpp:257-264): the iteration stops when the
                                         PRECONDITIONED residual || M^-1 (b - A x) || has fallen by linear_solver_reduction;
                                       1 = before such a solve is reported converged the TRUE residual b - A x is formed (one product) and the
                                           iteration continues from it, with a lowered threshold, until || b - A x || <= reduction || b || -- the
                                           statement the reference's default BiCGStab makes; the reported reduction is then the true one.
                                           Not a reference option; costs one SpMV per solve; left-preconditioned GMRES only.
                                           LIMIT with a float solve (single_precision = 1): b - A x is then formed in float, so the verified
                                           reduction cannot go below the rounding of that product (about 1e-6 x the conditioning of the
                                           scaled system; on the 1 M-cell bench deck a verified 1e-5 is NOT reached within 1000 iterations
                                           and the solve ends in OPMGPU_ELINSOLVE) -- ask a float solve for reductions >= 1e-3 with the
                                           check, or solve in double (tests/test_gpu_linsolver.py::test_gmres_verify_with_float_vectors) */
    int32_t cpr_reference_transform; /* 0 = CPR as DESIGN.md 4b describes it (only the pressure stage sees the combined equation);
                                        1 = the reference's formulation for comparability (NewtonIterationUtilities.cpp:253-287,
                                            NewtonIterationBlackoilCPR.cpp:117-131): the WHOLE system is row-transformed by L (per cell: its
                                            first equation is replaced by the sum of the dominant equations), the pressure row scaled by
                                            200 bar, and the Krylov method iterates on (and measures) L A x = L b;
                                        2 = 1, and the second stage is the reference's own: a POINT ILU0 of L A taken as a scalar matrix in
                                            equation-major order (`DuneMatrix istlA(A)`, NewtonIterationBlackoilCPR.cpp:129-133), where 0 and
                                            1 keep the 3x3-block ILU0.  With ilu_ordering = OPMGPU_ORDER_NATURAL this is dune's elimination
                                            order: linear iteration counts become comparable one to one with a solver_approach=cpr run of
                                            flow_legacy.  Double, single GPU; a comparison mode (the scalar plan's index lists take ~3.6 GB
                                            at 10^6 cells and a minute of host time per pattern) */
    /* --- the reference's CPR parameters (NewtonIterationBlackoilCPR.hpp:59-63 documents the first four with these defaults; they are read
     *     by the external opm-simulators CPRPreconditioner, whose source is NOT under /root/reference) ------------------------------------ */
    double cpr_relax;               /* 1.0: relaxation of the CPR preconditioner -- the ILU0 of stage 2 is built with it INSTEAD of
                                       ilu_relaxation, and a value != 1 also scales the pressure correction of stage 1 */
    int32_t cpr_ilu_n;              /* 0: fill-in level of stage 2's ILU(n) (NewtonIterationBlackoilCPR.hpp:61): n > 0 = block ILU(n) with level-of-fill, see
                                       ilu_fillin_level below.  0..8; not with cpr_reference_transform = 2 */
    int32_t cpr_use_amg;            /* 0 = the elliptic (pressure) part is preconditioned by a POINT ILU0 of A_p (the documented default),
                                       1 = by one AMG V-cycle on A_p (amg.hip) */
    int32_t cpr_use_bicgstab;       /* 1 = BiCGStab for the elliptic part, 0 = CG */
    /* the elliptic part's inner Krylov solve.  Its stopping rule lives in the external CPRPreconditioner: the two values below are this
     * library's RECOLLECTION of that file (cpr_solver_tol 1e-2, cpr_max_elliptic_iter 25), unpinned by anything in the container */
    double cpr_solver_tol;          /* 1e-2: reduction of || b_p - A_p x_p || the inner solve stops at */
    double cpr_stage2_relax;        /* 1.0.  LIBRARY EXTENSION: an extra relaxation of the second stage ALONE,
                                           M^-1 d = cpr_relax (x_p + cpr_stage2_relax ILU0^-1 (d - A x_p)).
                                       cpr_relax by itself multiplies the whole preconditioner by a scalar (that is what the reference's
                                       CPRPreconditioner::apply does with it, as recalled), which no Krylov method notices; damping stage 2
                                       against the pressure correction is a different preconditioner -- rounds 1-3 of this library ran 0.9,
                                       which is what the Norne-like deck (isolated cells) needs to reach 1e-10: tests/test_gpu_fullsize.py */
    int32_t preconditioner_single;  /* 0.  LIBRARY EXTENSION, 1 = mixed precision: a DOUBLE solve (single_precision = 0) builds and applies its
                                       preconditioner -- the ILU0 factors and sweeps, under CPR also the pressure stage and the stage-2 residual --
                                       in FLOAT; the Krylov method, its operator A, its residual and the solution stay double, so the solve meets
                                       the same reduction on the same double system.  The preconditioner's bytes are half (the path is HBM-bound).
                                       The reference's plug-ins have no such option: bench.py runs it as a variant, not as the headline.
                                       LIMIT: a float M^-1 is not one fixed linear operator (its rounding depends on the vector), which the
                                       recurrences of both Krylov methods assume -- BiCGStab stalls around 1e-9 .. 1e-10 (measured: 163
                                       iterations to 9.9e-10, then a breakdown, where the double preconditioner takes 12 iterations to
                                       7.5e-13), left-preconditioned GMRES's residual estimate drifts below ~1e-6.  Meant for the reductions
                                       Newton solves ask for (1e-2 by default); ask for no less than 1e-8 / 1e-6 with it */
    int32_t cpr_max_ell_iter;       /* 25: iteration limit of the inner solve (reaching it is NOT an error here: the outer method goes on with
                                       what the inner one attained).  0 = library extension, no inner Krylov method at all: ONE application
                                       of the elliptic preconditioner (with cpr_use_amg = 1: one V-cycle, its coarse-grid corrections scaled
                                       as DESIGN.md section 4b describes) -- what bench.py's headline runs, and says so */
    int32_t ilu_fillin_level;       /* 0: `ilu_fillin_level` of the interleaved solver (ISTLSolver.hpp:205, handed to ParallelOverlappingILU0):
                                       n > 0 = block ILU(n) with level-of-fill instead of the ILU0.  Under use_cpr the second stage takes
                                       cpr_ilu_n (above) instead.  Both run csrc/fillilu.inl: the level-of-fill rule for the pattern
                                       (lev(i,j) = min lev(i,k) + lev(k,j) + 1 <= n, rows in the caller's order; dune-istl is not in the
                                       reference tree: parity unpinned), this library's block ILU kernels on that pattern's own plan.  With
                                       ilu_ordering = OPMGPU_ORDER_NATURAL the elimination order is the caller's (dune's); MULTICOLOR colours
                                       the filled graph.  n <= 8; a filled pattern beyond 12 x the matrix's blocks is refused.  Measured
                                       (profiles/r04_aj_ilu_n.log): fewer iterations, never less time -- an option for parity of settings */
} opmgpu_params;

void opmgpu_default_params(opmgpu_params* p);
/* the equation scaling the last assembly used (== params.matbalscale unless update_equations_scaling) */
int opmgpu_get_matbalscale(opmgpu_ctx* ctx, double* scale3);
/* CPR with an inner Krylov method on the elliptic part (cpr_max_ell_iter > 0; the reference's external CPRPreconditioner::solveElliptic,
 * reached from NewtonIterationBlackoilCPR.cpp:148-165): inner solves and inner iterations since the context was created */
int opmgpu_cpr_elliptic_stats(opmgpu_ctx* ctx, int64_t* solves, int64_t* iterations);
/* diagnostic of cpr_reference_transform = 2: v = relax * (L U)^-1 d with the point ILU0 of the last such solve's (transformed) matrix;
 * d3 / v3 block-interleaved like opmgpu_ilu0_apply.  tests/test_gpu_linsolver.py checks it against a numpy ILU0 of the scalar matrix. */
int opmgpu_point_ilu_apply(opmgpu_ctx* ctx, const double* d3, double* v3, double relax);
/* diagnostic: the scaling of the pressure cycle's coarse-grid corrections the LAST CPR solve ran with (into level 0 / below it; DESIGN.md
 * section 5: 1.9, or on matrices of the model's own assembly the per-time-step choice between 1.9 and 2.3 -- and, after a failed solve, which is
 * repeated with 1.0, from the ladder 1.0 / 1.45 / 1.9 / 2.3; external matrices of the B1 path keep 1.9).  OPMGPU_EINVAL before the first CPR solve. */
int opmgpu_cpr_correction_factors(opmgpu_ctx* ctx, double* into_level0, double* below);

/* ------------------------------------------------------------------------------------------
 * B2 boundary: BlackoilModel hooks (BlackoilModelBase_impl.hpp:239-326: assemble ->
 * getConvergence -> solveJacobianSystem -> updateState).
 * ---------------------------------------------------------------------------------------- */

/* Replaces the BlackoilModel constructor's static inputs (BlackoilModel.hpp:63-79): grid,
 * BlackoilPropsAdFromDeck, DerivedGeology, RockCompressibility, model parameters.
 * Persists across report steps (SimulatorBase_impl.hpp:201-203 rebuilds the model object,
 * the device context lives with the linear solver, FlowMain.hpp:237). */
int opmgpu_create(opmgpu_ctx** ctx, int device, const opmgpu_grid* grid,
                  const opmgpu_tables* tables, const opmgpu_params* params);
void opmgpu_destroy(opmgpu_ctx* ctx);
const char* opmgpu_last_error(const opmgpu_ctx* ctx);

/* Wells topology (Wells::well_connpos / well_cells).  The BSR pattern becomes
 * stencil U per-well cliques, the fill eliminateVariable() creates
 * (NewtonIterationUtilities.cpp:98-115).  Call before the first assemble of a report step. */
int opmgpu_set_wells(opmgpu_ctx* ctx, int nw, const int32_t* well_connpos /*nw+1*/,
                     const int32_t* well_cells /*nperf*/);

/* ReservoirState in/out (BlackoilState.hpp:40-90): pressure[nc], saturation[nc*3]
 * cell-major interleaved (w,o,g), gasoilratio[nc], rv[nc], hydroCarbonState[nc]. */
int opmgpu_set_state(opmgpu_ctx* ctx, const double* p, const double* sat, const double* rs,
                     const double* rv, const int8_t* hcstate);
int opmgpu_get_state(opmgpu_ctx* ctx, double* p, double* sat, double* rs, double* rv,
                     int8_t* hcstate);

/* BlackoilModelBase::assemble (BlackoilModelBase_impl.hpp:757-840) minus the host well model:
 * variableState -> [initial: computeAccum(state0,0)] -> assembleMassBalanceEq.
 * State pointers may all be NULL to use the device-resident state.  `initial` is the
 * reference's initial_assembly flag (iteration == 0): accum0 is (re)computed, which is what
 * makes re-entry with a rolled-back state and a chopped dt safe. */
int opmgpu_assemble(opmgpu_ctx* ctx, double dt, int initial, const double* p, const double* sat,
                    const double* rs, const double* rv, const int8_t* hcstate);

/* Per-perforation cell quantities the host StandardWells model needs
 * (extractWellPerfProperties, StandardWells_impl.hpp:396-571).  out is [nperf][OPMGPU_PERF_K]:
 * for q in {p_o, rs, rv, b_w, b_o, b_g, mob_w, mob_o, mob_g}: value, d/dP, d/dSw, d/dXvar. */
#define OPMGPU_PERF_K 36
int opmgpu_perf_props(opmgpu_ctx* ctx, double* out);

/* addWellContributionToMassBalanceEq (BlackoilModelBase_impl.hpp:953-975) + the Schur
 * complement of eliminateVariable (NewtonIterationUtilities.cpp:45-128), computed on the host:
 * resid_delta[nperf*3] is ADDED to the (unscaled) residual of the perforated cells (phase
 * fastest), schur blocks (UNSCALED, row-major 3x3, d eq / d var) are ADDED to J at
 * (schur_rc[2k], schur_rc[2k+1]) in caller cell numbering; the library applies matbalscale. */
int opmgpu_add_well_terms(opmgpu_ctx* ctx, const double* resid_delta, int nblk,
                          const int32_t* schur_rc, const double* schur_blocks);

/* Right-hand-side part of the Schur elimination: -B D^-1 r_well per perforation (UNSCALED, phase fastest).  Unlike
 * resid_delta above it must NOT enter the convergence check (the reference eliminates on a copy inside the linear
 * solver, NewtonIterationBlackoilInterleaved.cpp:221-231), so it is kept apart and only added when the RHS is built.
 * Cleared by every opmgpu_assemble. */
int opmgpu_add_well_rhs(opmgpu_ctx* ctx, const double* rhs_delta /*nperf*3*/);
/* Newton increment of the perforated cells after opmgpu_solve (for recoverVariable, NewtonIterationUtilities.cpp:134-184):
 * out[nperf*3] = dP, dSw, dXvar per perforation. */
int opmgpu_perf_dx(opmgpu_ctx* ctx, double* out);

/* Precision of the coming opmgpu_solve, known before the assembly like the reference's residual_.singlePrecision = dt <
 * maxSinglePrecisionTimeStep (BlackoilModelBase_impl.hpp:284).  With single_precision != 0 the next opmgpu_assemble writes
 * the Jacobian as float directly (no f64 copy, no conversion pass; opmgpu_get_jacobian_bsr then returns the float values
 * widened, and a double solve after such an assembly works on the widened values).  Default 0 = double. */
int opmgpu_set_solve_precision(opmgpu_ctx* ctx, int single_precision);

/* getConvergence / convergenceReduction (BlackoilModelBase_impl.hpp:1633-1857) for the reservoir
 * equations: B_avg, CNV, MB per phase plus the L-inf residual norms of computeResidualNorms
 * (:1551-1589).  *converged = all MB < tol_mb && all CNV < tol_cnv.  Returns OPMGPU_ENUMERICAL
 * on NaN or > max_residual_allowed exactly where the reference throws NumericalIssue. */
int opmgpu_convergence(opmgpu_ctx* ctx, double dt, double* B_avg3, double* CNV3, double* MB3,
                       double* linf3, int* converged);

/* solveJacobianSystem -> NewtonIterationBlackoilInterleaved::computeNewtonIncrement
 * (NewtonIterationBlackoilInterleaved.cpp:202-292, :467-487) on the assembled, matbal-scaled
 * device system: block-ILU0 + BiCGStab in float if single_precision (the reference's
 * dt < 20 d switch, :478-480), else double.  dx (3*nc, equation-major, caller cell order) may be
 * NULL to keep the increment resident for opmgpu_update_state. */
int opmgpu_solve(opmgpu_ctx* ctx, int single_precision, double* dx, int* iters, double* reduction);

/* updateState (BlackoilModelBase_impl.hpp:1147-1389): dp/ds chopping, saturation
 * renormalisation, rs/rv limits, phase-state switching.  dx NULL = use the resident increment
 * of the last opmgpu_solve.  relax multiplies dx first (NonlinearSolver_impl.hpp:283-301, dampen). */
int opmgpu_update_state(opmgpu_ctx* ctx, const double* dx, double relax);

/* ------------------------------------------------------------------------------------------
 * Wells on the device: the standard well model of StandardWells_impl.hpp:396-998 evaluated per well on
 * the GPU, the well unknowns (q_s[3], bhp) Schur-eliminated like NewtonIterationUtilities.cpp:45-128 -- but kept in
 * factored form (diagonal-block updates + a rank-7 operator per well applied inside the SpMV) instead of filling the
 * matrix with a clique per well.  Alternative to the host-well path (opmgpu_set_wells / perf_props / add_well_terms):
 * after opmgpu_set_device_wells, opmgpu_assemble adds the well terms itself, opmgpu_solve uses the coupled operator,
 * opmgpu_update_state also recovers and updates the well unknowns (updateWellState, :611-650), and
 * opmgpu_save_state / restore_state carry the well state along.  Fields mirror opm-core's `Wells` struct.
 * ---------------------------------------------------------------------------------------- */
typedef struct opmgpu_wells {
    int32_t        nw;
    const int32_t* well_connpos;   /* [nw+1]                                                   */
    const int32_t* well_cells;     /* [nperf] (a cell may be perforated by one well only)      */
    const double*  WI;             /* [nperf] well index (connection transmissibility factor)  */
    const int32_t* type;           /* [nw] 0 = INJECTOR, 1 = PRODUCER                          */
    const int32_t* allow_cf;       /* [nw] allow cross flow; NULL = all 1                      */
    const double*  depth_ref;      /* [nw] bhp reference depth                                 */
    const double*  comp_frac;      /* [nw*3] injection stream composition (w, o, g)            */
    const int32_t* ctrl_type;      /* [nctrl] OPMGPU_CTRL_*                                    */
    const double*  ctrl_target;    /* [nctrl]                                                  */
    const double*  ctrl_distr;     /* [nctrl*3] rate-control phase weights; NULL = 0           */
    /* WellControls with several controls per well (opm-core well_controls.h): control ctrl_ptr[w] is the well's initial
     * CURRENT control (an equation), the others are inequality constraints that updateWellControls switches to when broken
     * (StandardWells_impl.hpp:709-800).  ctrl_ptr == NULL: one control per well (nctrl == nw).                           */
    const int32_t* ctrl_ptr;       /* [nw+1] or NULL                                           */
    const int32_t* ctrl_vfp;       /* [nctrl] VFP table id of the THP controls, or NULL        */
    const double*  ctrl_alq;       /* [nctrl] artificial lift quantity of THP controls, or NULL */
} opmgpu_wells;
enum { OPMGPU_CTRL_BHP = 0, OPMGPU_CTRL_SURFACE_RATE = 1, OPMGPU_CTRL_THP = 2, OPMGPU_CTRL_RESERVOIR_RATE = 3 };

/* VFP tables for THP control (VFPProdPropertiesLegacy.cpp:36-154, VFPInjPropertiesLegacy.cpp:36-130; opm-common's VFPProdTable /
 * VFPInjTable).  SI units.  Producer data is [nthp][nwfr][ngfr][nalq][nflo], injector data [nthp][nflo] (the other axes have
 * length 1 and may be NULL).  Multilinear interpolation, linear extrapolation outside the axes. */
enum { OPMGPU_VFP_FLO_OIL = 0, OPMGPU_VFP_FLO_LIQ = 1, OPMGPU_VFP_FLO_GAS = 2 };
enum { OPMGPU_VFP_WFR_WOR = 0, OPMGPU_VFP_WFR_WCT = 1, OPMGPU_VFP_WFR_WGR = 2 };
enum { OPMGPU_VFP_GFR_GOR = 0, OPMGPU_VFP_GFR_GLR = 1, OPMGPU_VFP_GFR_OGR = 2 };
typedef struct opmgpu_vfp_table {
    int32_t id, is_injector;
    int32_t flo_type, wfr_type, gfr_type;
    double  datum_depth;
    int32_t nflo, nthp, nwfr, ngfr, nalq;
    const double *flo, *thp, *wfr, *gfr, *alq;
    const double* data;
} opmgpu_vfp_table;

/* Call BEFORE opmgpu_set_device_wells when a well has a THP control; n == 0 removes the tables. */
int opmgpu_set_vfp_tables(opmgpu_ctx* ctx, int n, const opmgpu_vfp_table* tables);
/* nw == 0 removes them.  Multi-GPU: a well lives on ONE rank, and EVERY rank of a run with wells makes this call (nw = 0 where it owns
 * none) -- opmgpu_well_convergence is collective, and the call tells the pressure stage's coarse space that the run has wells. */
int opmgpu_set_device_wells(opmgpu_ctx* ctx, const opmgpu_wells* wells);
/* WellStateFullyImplicitBlackoil fields: bhp[nw], wellRates qs[nw*3]; perfPress perf_press[nperf] and perfPhaseRates
 * perf_rates[nperf*3] may be NULL (keep).  perf_press feeds the average well-block pressures of computeWellConnectionPressures. */
int opmgpu_well_state_set(opmgpu_ctx* ctx, const double* bhp, const double* qs, const double* perf_press, const double* perf_rates);
int opmgpu_well_state_get(opmgpu_ctx* ctx, double* bhp, double* qs, double* perf_press, double* perf_rates);
/* currentControls() (index into each well's controls, relative to ctrl_ptr[w]) and thp() of the well state; any pointer may be NULL.
 * get also reports the iteration count and outcome of the last explicit well pre-solve (solveWellEq). */
int opmgpu_well_controls_set(opmgpu_ctx* ctx, const int32_t* current, const double* thp);
int opmgpu_well_controls_get(opmgpu_ctx* ctx, int32_t* current, double* thp, int32_t* presolve_iterations, int32_t* presolve_converged);
/* well_controls_iset_target / well_controls_iset_distr for ALL controls (order of opmgpu_set_device_wells; either pointer may be NULL = keep):
 * what SimulatorBase::computeRESV does to the RESERVOIR_RATE controls once per report step (SimulatorBase_impl.hpp:551-553). */
int opmgpu_well_controls_set_targets(opmgpu_ctx* ctx, const double* ctrl_target, const double* ctrl_distr);
/* computePropertiesForWellConnectionPressures (StandardWells_impl.hpp:218-296) for the HOST well model: b_w, b_o, b_g, rsSat, rvSat
 * of the perforated cells (opmgpu_set_wells order) evaluated at the given pressures with the cells' own rs / rv / phase condition /
 * oil saturation.  out[nperf*5]. */
int opmgpu_perf_pvt(opmgpu_ctx* ctx, const double* pressure, double* out);
/* B_avg of getWellConvergence (BlackoilModelBase_impl.hpp:1876-1891) for the host well model's pre-solve: mean 1/b per phase of the
 * last assembly. */
int opmgpu_average_b(opmgpu_ctx* ctx, double* B_avg3);
/* well part of getConvergence (BlackoilModelBase_impl.hpp:1769-1779) after opmgpu_assemble: max |flux equation| per
 * phase (to be multiplied by B_avg and compared with tolerance_wells) and max |control equation|. */
int opmgpu_well_convergence(opmgpu_ctx* ctx, double* flux_residual3, double* control_residual);

/* Device-side last_state of AdaptiveTimeStepping (AdaptiveTimeStepping_impl.hpp:211-212, :318-319, :346-347): save = copy
 * the resident reservoir state aside, restore = copy it back after a failed sub-step (the state never leaves the device),
 * relative_change = BlackoilModelBase::relativeChange(previous = saved, current = resident)
 * (BlackoilModelBase_impl.hpp:1595-1631): (|p0-p|^2 + |s0-s|^2) / (|p|^2 + |s|^2), what the PID controllers consume. */
int opmgpu_save_state(opmgpu_ctx* ctx);
int opmgpu_restore_state(opmgpu_ctx* ctx);
int opmgpu_relative_change(opmgpu_ctx* ctx, double* value);

/* RateConverter::SurfaceToReservoirVoidage (RateConverterLegacy.hpp:407-770), the two halves SimulatorBase::computeRESV
 * (SimulatorBase_impl.hpp:476-553) and computeWellVoidageRates (BlackoilModelBase_impl.hpp:2490-2534) call:
 * defineState -> calcAverages (:718-768): per region the SUMS of p, rs, rv of the RESIDENT state over the cells and their number,
 *   sums[nregions][4]; region[nc] in the caller's cell order with values 0 .. nregions-1, NULL = one region of all cells (what
 *   SimulatorBase builds, SimulatorBase_impl.hpp:66).  Collective in decomposed runs (owned cells, summed over the ranks: the is_parallel
 *   branch).  The division -- and the reference's habit of NOT clearing rs / rv between calls (:733-737) -- stay with the caller's mirror.
 * calcCoeff (:495-548): coeff[n][3] (water, oil, gas) with q_rT = sum_p coeff[p] q_s[p] at n given (p, rs, rv) points; pvt_region NULL = 0. */
int opmgpu_region_state_sums(opmgpu_ctx* ctx, const int32_t* region, int nregions, double* sums);
int opmgpu_voidage_coefficients(opmgpu_ctx* ctx, int n, const double* p, const double* rs, const double* rv, const int32_t* pvt_region, double* coeff);

/* BlackoilModelBase::computeFluidInPlace (BlackoilModelBase_impl.hpp:2263-2445; called by SimulatorBase_impl.hpp:207, :278 and
 * AdaptiveTimeStepping_impl.hpp:322) for the RESIDENT state: per cell fip[phase] = pv_mult * b_phase * s_phase * pv (b at the phase
 * pressures and the cell's phase condition), dissolved gas rs * fip[oil], vaporised oil rv * fip[gas], summed per region together with
 * the pore volume and the hydrocarbon-pore-volume weighted average pressure.
 *   fipnum    [nc] region of every cell in the caller's cell order, 1-based, 0 = in no region (the reference's fipnum[c] - 1 == -1);
 *             NULL = one region of all cells
 *   values    [nregions][7]: water, oil, gas, dissolved gas, vaporised oil, pore volume, weighted pressure (SimulatorData::FipId,
 *             BlackoilModelEnums.hpp:53-61)
 *   fip_cells [7][nc] or NULL: the per-cell arrays of SimulatorData::fip (getFIPData()), caller's cell order
 * The per-cell evaluation runs on the device, the region loops on the host in the reference's cell order.  Decomposed runs (the
 * reference's parallel branch, :2369-2446): a COLLECTIVE call; every rank passes the fipnum of its local cells (ghost cells included --
 * only owned cells are counted) and the GLOBAL number of regions, the sums are taken over the owned cells and all-reduced, every rank
 * receives the global values (fip_cells: this rank's cells, zeros in the pv / weighted-pressure rows of its ghosts). */
int opmgpu_compute_fluid_in_place(opmgpu_ctx* ctx, const int32_t* fipnum, int nregions, double* fip_cells, double* values);

/* Maximum historical oil saturation per cell (BlackoilPropsAdFromDeck::satOilMax_, used by VAPPARS).
 * set: explicit values (nc, caller order; restart).  update: soMax = max(soMax, so of the resident state) --
 * what SimulatorBase_impl.hpp:192 does at the start of every report step (updateSatOilMax, :933-945).
 * Starts at zero like the reference's (BlackoilPropsAdFromDeck.cpp:175). */
int opmgpu_set_sat_oil_max(opmgpu_ctx* ctx, const double* so_max);
int opmgpu_update_sat_oil_max(opmgpu_ctx* ctx);
int opmgpu_get_sat_oil_max(opmgpu_ctx* ctx, double* so_max);

/* Hysteresis history (EclHysteresisTwoPhaseLawParams::update, called once per report step by SimulatorBase_impl.hpp:190-191 ->
 * BlackoilPropsAdFromDeck::updateSatHyst): update = take the resident state's saturations into the history (krnSwMdc of both two-phase
 * systems = running minimum of 1 - So resp. 1 - Sg, the reference's "inconsistent" update) and recompute the Carlson shifts.
 * set / get: the two history planes [nc] each (restart); they start at the no-history value 2.0 (EclHysteresisTwoPhaseLawParams). */
int opmgpu_update_hysteresis(opmgpu_ctx* ctx);
int opmgpu_set_hysteresis(opmgpu_ctx* ctx, const double* krn_sw_mdc_ow, const double* krn_sw_mdc_go);
int opmgpu_get_hysteresis(opmgpu_ctx* ctx, double* krn_sw_mdc_ow, double* krn_sw_mdc_go, double* delta_ow, double* delta_go);

/* NonlinearSolver::stabilizeNonlinearUpdate (NonlinearSolver_impl.hpp:260-301) on the resident
 * increment: dx_old <- dx, then DAMPEN: dx *= omega, SOR: dx = omega*dx + (1-omega)*dx_old(previous).
 * dx_old is zeroed by opmgpu_assemble(initial_assembly = 1) (BlackoilModelBase_impl.hpp:254-260).
 * The oscillation detection itself (detectOscillations, :221-257) works on the three L-inf norms
 * that opmgpu_convergence returns and stays with the caller (host/opmgpu.hpp, opmgpu/model.py). */
#define OPMGPU_RELAX_DAMPEN 0
#define OPMGPU_RELAX_SOR 1
int opmgpu_stabilize_update(opmgpu_ctx* ctx, int relax_type, double omega);

/* ------------------------------------------------------------------------------------------
 * B1 boundary: NewtonIterationBlackoilInterface::computeNewtonIncrement
 * (NewtonIterationBlackoilInterface.hpp:31-52) with the matrix supplied by the reference's own
 * AD assembly after formInterleavedSystem (NewtonIterationBlackoilInterleaved.cpp:110-194):
 * BCRSMatrix<3x3> as BSR (rowptr[nb+1], col[nnzb], val[nnzb*9] row-major blocks), rhs/x
 * block-interleaved [nb][3] like Dune BlockVector.  The context may come from
 * opmgpu_create_solver (no grid/tables).  The sparsity plan is cached while the pattern
 * is unchanged. */
int opmgpu_create_solver(opmgpu_ctx** ctx, int device, const opmgpu_params* params);
int opmgpu_solve_bsr(opmgpu_ctx* ctx, int nb, const int32_t* rowptr, const int32_t* col,
                     const double* val9, const double* rhs3, int single_precision, double* x3,
                     int* iters, double* reduction);

/* ------------------------------------------------------------------------------------------
 * Kernel-level entry points (parity tests, roofline bench).  They operate on the matrix held
 * by the context: the last opmgpu_solve_bsr / opmgpu_load_bsr / opmgpu_assemble system.
 * ---------------------------------------------------------------------------------------- */
int opmgpu_load_bsr(opmgpu_ctx* ctx, int nb, const int32_t* rowptr, const int32_t* col,
                    const double* val9, int single_precision);
/* y = A x, x/y block-interleaved [nb][3] (MatrixAdapter::apply, used at ISTLSolver.hpp:267). */
int opmgpu_spmv(opmgpu_ctx* ctx, const double* x3, double* y3);
/* ILU0 of the loaded matrix (ParallelOverlappingILU0 ctor) and one application
 * v = w * U^-1 L^-1 d (its apply()). */
int opmgpu_ilu0_factor(opmgpu_ctx* ctx);
int opmgpu_ilu0_apply(opmgpu_ctx* ctx, const double* d3, double* v3);
/* factors back in the caller's BSR layout: L strictly lower / U strictly upper blocks in the
 * slots of the input pattern w.r.t. the ELIMINATION order, diagonal slot = inverted pivot. */
int opmgpu_ilu0_get(opmgpu_ctx* ctx, double* val9);
/* elimination position of every caller row (perm[row] = position) and its level. */
int opmgpu_get_ordering(opmgpu_ctx* ctx, int32_t* position, int32_t* level, int32_t* nlevels);
/* per-cell 0/1 weights of the CPR pressure equation after a solve with use_cpr (formEllipticSystem's dominance test,
 * NewtonIterationUtilities.cpp:212-252): w[3*nb], equation-major (water, oil, gas), caller row order. */
int opmgpu_get_cpr_weights(opmgpu_ctx* ctx, double* w);
/* assembled reservoir system back to the host in caller numbering: residual (unscaled,
 * equation-major 3*nc), and the matbal-scaled Jacobian in BSR. */
int opmgpu_get_residual(opmgpu_ctx* ctx, double* r);
int opmgpu_get_jacobian_nnzb(opmgpu_ctx* ctx, int32_t* nnzb);
int opmgpu_get_jacobian_bsr(opmgpu_ctx* ctx, int32_t* rowptr, int32_t* col, double* val9);
/* timing of device work: runs `reps` launches of one kernel on the context's stream between two
 * hipEvents and returns the average milliseconds per launch (HIP events on the launch stream). */
enum { OPMGPU_K_SPMV = 0, OPMGPU_K_ILU_APPLY = 1, OPMGPU_K_ILU_FACTOR = 2, OPMGPU_K_ASSEMBLE = 3,
       OPMGPU_K_DOT = 4, OPMGPU_K_AXPY = 5, OPMGPU_K_PROPS = 6, OPMGPU_K_STREAM_COPY = 7,
       OPMGPU_K_CPR_APPLY = 8 /* whole two-stage preconditioner application */, OPMGPU_K_VCYCLE = 9 /* its AMG V-cycle alone */,
       OPMGPU_K_CPR_SETUP = 10 /* pressure extraction + Galerkin + coarsest inverse */,
       OPMGPU_K_SPMV_COLD = 11 /* the SpMV rotating over copies of the matrix that together exceed the 256 MiB Infinity Cache: HBM-resident operands */ };
int opmgpu_time_kernel(opmgpu_ctx* ctx, int kernel, int reps, double* ms_per_launch);
/* In-situ timing of kernel CLASSES during real Newton iterations: after opmgpu_kernel_timing(ctx, 1) every launch (group) of a class
 * is bracketed by a HIP event pair on the launch stream; opmgpu_kernel_timing_get sums the elapsed times per class since the switch-on
 * (total_ms[OPMGPU_KT_COUNT], launches[OPMGPU_KT_COUNT] bracket counts).  Off by default -- the brackets cost a few microseconds each,
 * so bench.py uses a separate profiled pass for its per-kernel roofline table. */
enum { OPMGPU_KT_CELL_PROPS = 0, OPMGPU_KT_FLUX, OPMGPU_KT_WELLS, OPMGPU_KT_CONV, OPMGPU_KT_ILU_FACTOR, OPMGPU_KT_CPR_SETUP, OPMGPU_KT_SPMV1,
       OPMGPU_KT_SPMV2, OPMGPU_KT_ILU_APPLY, OPMGPU_KT_VCYCLE, OPMGPU_KT_CPR_OTHER, OPMGPU_KT_VECTOR, OPMGPU_KT_UPDATE_STATE, OPMGPU_KT_COUNT };
int opmgpu_kernel_timing(opmgpu_ctx* ctx, int enable);
int opmgpu_kernel_timing_get(opmgpu_ctx* ctx, double* total_ms, int64_t* launches);
/* elapsed device milliseconds of the last assemble / solve / update_state call. */
int opmgpu_last_timings(opmgpu_ctx* ctx, double* assemble_ms, double* solve_ms, double* update_ms);
/* Per-call timing of opmgpu_nonlinear_iteration WITHOUT a synchronisation inside the timed calls (the median over the Newton
 * iterations that include a linear solve).  After opmgpu_iteration_marks(ctx, 1) every call records one event on the library's stream when
 * it has enqueued its last kernel, and one event pair around each of its phases; opmgpu_iteration_marks_get waits for the last recorded
 * event and returns for the first max_calls calls since the switch-on: call_ms[i] = device time from the end of call i - 1 (the switch-on
 * for i = 0) to the end of call i -- the host round trips of an iteration serialise consecutive calls, so this is the iteration's wall
 * time --, solved[i] = 1 if the call ran solveJacobianSystem + updateState (a converged call does not: BlackoilModelBase_impl.hpp:277-281),
 * linear_iterations[i], phase_ms[3 i + {0, 1, 2}] = assemble / solve / update (0 where the phase did not run).  Any output may be NULL;
 * *n_calls = calls made since the switch-on.  At most 16384 calls are recorded (later ones run unmarked and only count); the events are
 * recycled through a pool that lives as long as the context.  opmgpu_iteration_marks(ctx, 0) switches off and returns the events to it. */
int opmgpu_iteration_marks(opmgpu_ctx* ctx, int enable);
int opmgpu_iteration_marks_get(opmgpu_ctx* ctx, int max_calls, double* call_ms, int32_t* solved, int32_t* linear_iterations, double* phase_ms, int* n_calls);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU (domain decomposition with a one-cell halo; mirrors owner/overlap of
 * ParallelISTLInformation, ISTLSolver.hpp:286-298).  One process per GPU; the RCCL unique id
 * is created on rank 0 and handed to the other ranks by the caller (torch.distributed / MPI).
 * Cells [0, n_owned) of the rank-local grid are owned, [n_owned, nc) are ghosts.
 * send lists name owned local cells, recv lists name ghost local cells, grouped per
 * neighbour rank.
 * ---------------------------------------------------------------------------------------- */
#define OPMGPU_UNIQUE_ID_BYTES 128
int opmgpu_comm_unique_id(uint8_t* id /*OPMGPU_UNIQUE_ID_BYTES*/);
int opmgpu_comm_init(opmgpu_ctx* ctx, int rank, int nranks, const uint8_t* id, int32_t n_owned,
                     int n_neigh, const int32_t* neigh_rank, const int32_t* send_ptr,
                     const int32_t* send_cells, const int32_t* recv_ptr, const int32_t* recv_cells);

/* The same communicator over a caller-supplied TRANSPORT instead of RCCL -- e.g. the MPI communicator flow_legacy already owns
 * (ParallelISTLInformation), or the shared-memory test transport of tests/support.  The library calls back for its two
 * primitives, both on device buffers and ordered on the given HIP stream (the callback enqueues on it or synchronises it):
 *   allreduce: in-place sum (is_max = 0) or max of n doubles;
 *   exchange:  for every neighbour q send scount[q] bytes from sbuf + soff[q] to rank neigh_rank[q] and receive rcount[q] bytes
 *              from it into rbuf + roff[q] (byte offsets / counts).
 * The stream is not always the same one: exchanges of the solver arrive on the library's second ("halo") stream while the rows that
 * need no ghost value are multiplied on the main stream; calls are issued in the same order on every rank and never concurrently.
 * A non-zero return value becomes OPMGPU_ECOMM.  The struct is copied; `destroy(self)` is called when the context goes away. */
typedef struct opmgpu_transport {
    void* self;
    int  (*allreduce)(void* self, double* dbuf, int n, int is_max, void* hip_stream);
    int  (*exchange)(void* self, int n_neigh, const int32_t* neigh_rank, const void* sbuf, const int64_t* soff, const int64_t* scount,
                     void* rbuf, const int64_t* roff, const int64_t* rcount, void* hip_stream);
    void (*destroy)(void* self);
    /* OPTIONAL (may be NULL: the library then calls allreduce and exchange one after the other).  One all-reduce (sum) and one neighbour
     * exchange that do not depend on each other, as ONE operation -- the decomposed GMRES sends the new basis vector's halo together with
     * its projections, and the pressure correction's halo together with the coarse space's restricted residual: two latencies per column
     * instead of four.  The built-in RCCL transport issues them back to back by default (OPMGPU_RCCL_FUSED=1: one ncclGroup). */
    int  (*allreduce_exchange)(void* self, double* dbuf, int n, int n_neigh, const int32_t* neigh_rank, const void* sbuf, const int64_t* soff,
                               const int64_t* scount, void* rbuf, const int64_t* roff, const int64_t* rcount, void* hip_stream);
} opmgpu_transport;
/* Coarse space of the decomposed pressure stage: by default one unknown per rank in runs with wells (index-range
 * blocks of the owned cells would cut the wells, which was measured to hurt), m = 4 index-range blocks without.  A caller that knows the
 * geometry supplies m <= 8 blocks per rank that keep every well inside ONE block -- e.g. sub-slabs along the cut direction for vertical
 * wells: block_of_owned_cell[c] in [0, m) for the owned cells in local numbering.  Same m on every rank; m = 0 restores the default.
 * Call after opmgpu_comm_init*, before the first CPR solve (the map is agreed collectively there). */
int opmgpu_comm_set_coarse_blocks(opmgpu_ctx* ctx, int m, const int32_t* block_of_owned_cell);
int opmgpu_comm_init_transport(opmgpu_ctx* ctx, int rank, int nranks, const opmgpu_transport* transport, int32_t n_owned,
                               int n_neigh, const int32_t* neigh_rank, const int32_t* send_ptr,
                               const int32_t* send_cells, const int32_t* recv_ptr, const int32_t* recv_cells);

/* ----------------------------------------------------------------------------------------
 * One whole Newton iteration in one call: BlackoilModelBase::nonlinearIteration (BlackoilModelBase_impl.hpp:239-326) with the
 * update stabilisation of NonlinearSolver (NonlinearSolver_impl.hpp:221-301) --
 *   iteration == 0: the residual-norm history and the relaxation factor are reset;
 *   assemble(initial = iteration == 0) -> getConvergence (reservoir, and the wells of the device well model) ->
 *   if not converged, or iteration < min_iter:  solveJacobianSystem -> detectOscillations -> stabilizeNonlinearUpdate -> updateState.
 * It is the sequence opmgpu_set_solve_precision / assemble / convergence / well_convergence / solve / stabilize_update / update_state
 * that the host mirrors issue call by call (opmgpu/model.py, host/opmgpu.hpp), with the host round trips between them inside the
 * library: the GPU idles ~0.1 ms per iteration less.  The history lives in the context.  Status codes as for the single calls.
 * ---------------------------------------------------------------------------------------- */
typedef struct opmgpu_newton_ctl {
    int32_t min_iter;                  /* 1   (NonlinearSolver: min_iter_) */
    int32_t use_update_stabilization;  /* 1   (BlackoilModelParameters::use_update_stabilization_) */
    int32_t relax_type;                /* OPMGPU_RELAX_DAMPEN */
    double  relax_max;                 /* 0.5 */
    double  relax_increment;           /* 0.1 */
    double  relax_rel_tol;             /* 0.2 */
} opmgpu_newton_ctl;
int opmgpu_nonlinear_iteration(opmgpu_ctx* ctx, double dt, int iteration, int single_precision, const opmgpu_newton_ctl* ctl,
                               int* converged, int* linear_iterations, double* linf3 /* may be NULL */, double* relaxation /* may be NULL */);

/* Host-only planning entry (no device needed): elimination position and level of every row for
 * the given ordering -- the same plan opmgpu_get_ordering reports for a loaded matrix. */
int opmgpu_plan_ordering(int nb, const int32_t* rowptr, const int32_t* col, int ordering,
                         int32_t* position, int32_t* level, int32_t* nlevels);

/* library / build information */
const char* opmgpu_version(void);
int opmgpu_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* OPMGPU_H */
