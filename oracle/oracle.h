/*
 * oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the flow_legacy Newton-step maths (OPM/opm-simulators-legacy,
 * opm/autodiff) used ONLY as the checker by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg.  Nothing under opm-simulators-legacy_amd/ may include,
 * link or call this.  Each function cites the reference file:line it follows.
 *
 * Parity pinning (see DESIGN.md "Oracle"): the relperm path is pinned by the
 * reference's own known answers (tests/test_satfunc.cpp:93-108 with
 * tests/satfuncStandard.DATA; :140-379, :480-660 with the satfuncEPS{Base,_A,_C,_D}.DATA
 * end-point-scaling decks; tests/test_boprops_ad.cpp:109-208 with tests/fluid.data),
 * transcribed under tests/golden/.  PVT table interpolation (opm-material), ILU0
 * (opm-simulators ParallelOverlappingILU0) and BiCGStab (dune-istl) are third-party
 * code absent from the reference tree with no numeric pins in it: for those the
 * oracle restates the published algorithm -- "parity unpinned".
 *
 * Data schema (grid / tables / params structs) is shared with include/opmgpu.h.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include "../include/opmgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* SaturationPropsFromDeck::relperm / capPress (SaturationPropsFromDeck.cpp:74-204):
 * s[n*3] (w,o,g), kr[n*3], dkrds[n*9] Fortran order dkrds[9*i + 3*sat + kr]. */
void oracle_relperm(const opmgpu_tables* t, int n, const double* s, const int32_t* satnum,
                    double* kr, double* dkrds);
/* same with ENDSCALE: cells[i] indexes the per-cell end points / satnum of the grid (tests/test_satfunc.cpp:227-379) */
void oracle_relperm_eps(const opmgpu_tables* t, const opmgpu_grid* g, int n, const double* s, const int32_t* cells,
                        double* kr, double* dkrds);
void oracle_cappress(const opmgpu_tables* t, int n, const double* s, const int32_t* satnum,
                     double* pc, double* dpcds);

/* BlackoilPropsAdFromDeck::{bWat,bOil,bGas,muWat,muOil,muGas,rsSat,rvSat}
 * (BlackoilPropsAdFromDeck.cpp:264-738).  which: 0 bWat 1 bOil 2 bGas 3 muWat 4 muOil 5 muGas
 * 6 rsSat 7 rvSat.  r = rs (oil) or rv (gas); saturated = cond.hasFreeGas()/hasFreeOil().
 * out[n*3] = value, d/dp, d/dr. */
void oracle_pvt(const opmgpu_tables* t, int which, int n, const double* p, const double* r,
                const int8_t* saturated, const int32_t* pvtnum, double* out);

/* Per-cell quantities of the SolutionState + ReservoirResidualQuant
 * (BlackoilModelBase_impl.hpp:614-703, :709-751, :1484-1512, :2009-2027).
 * out[nc][ORACLE_NPROP][4] = value, d/dP, d/dSw, d/dXvar for, in order:
 * p_w p_o p_g  b_w b_o b_g  mu_w mu_o mu_g  kr_w kr_o kr_g  rho_w rho_o rho_g
 * mob_w mob_o mob_g  rs rv  accum_w accum_o accum_g  (accum = pv_mult-weighted, :728-750). */
#define ORACLE_NPROP 23
void oracle_cell_props(const opmgpu_grid* g, const opmgpu_tables* t, const double* p,
                       const double* sat, const double* rs, const double* rv,
                       const int8_t* hc, double* out);

/* BSR pattern of the reservoir Jacobian: {c} U neighbours U well cliques, columns ascending
 * (formInterleavedSystem, NewtonIterationBlackoilInterleaved.cpp:110-155).
 * Call with rowptr==NULL to get nnzb only. */
int oracle_pattern(const opmgpu_grid* g, int nw, const int32_t* well_connpos,
                   const int32_t* well_cells, int32_t* rowptr, int32_t* col);

/* assemble() for the reservoir equations (BlackoilModelBase_impl.hpp:757-913):
 * residual r[3*nc] equation-major UNSCALED; Jacobian val9 on the given pattern, rows of
 * equation a scaled by scale[a] (NewtonIterationBlackoilInterleaved.cpp:234-236).
 * accum0[3*nc] (equation-major) is written when initial != 0, read otherwise. */
void oracle_assemble(const opmgpu_grid* g, const opmgpu_tables* t, double dt, int initial,
                     const double* p, const double* sat, const double* rs, const double* rv,
                     const int8_t* hc, const double* scale3, double* accum0,
                     const int32_t* rowptr, const int32_t* col, double* r, double* val9,
                     double* binv /* [3*nc] 1/b per phase, equation-major; may be NULL */);

/* getConvergence / convergenceReduction (BlackoilModelBase_impl.hpp:1633-1857). Returns the
 * OPMGPU_* status the reference's throws map to. */
int oracle_convergence(const opmgpu_grid* g, const opmgpu_params* prm, double dt, const double* r,
                       const double* binv, double* B_avg3, double* CNV3, double* MB3,
                       double* linf3, int* converged);

/* updateState (BlackoilModelBase_impl.hpp:1147-1389), in place on the state arrays. */
void oracle_update_state(const opmgpu_grid* g, const opmgpu_tables* t, const opmgpu_params* prm,
                         const double* dx, double* p, double* sat, double* rs, double* rv,
                         int8_t* hc);

/* VAPPARS: satOilMax_ per cell used by every later call (BlackoilPropsAdFromDeck.cpp:933-955); the pointer is kept, NULL = zeros */
void oracle_set_sat_oil_max(const double* so_max);

/* MatrixAdapter::apply: y = A x on BSR, x/y block-interleaved. */
void oracle_spmv(int nb, const int32_t* rowptr, const int32_t* col, const double* val9,
                 const double* x3, double* y3, int single_precision);

/* Block ILU(0), IKJ variant with explicitly inverted pivots (dune-istl bilu0_decomposition as
 * used by opm-simulators ParallelOverlappingILU0; 3x3 cofactor inverse of MatrixBlock.hpp).
 * position[row] = elimination position (NULL = natural).  lu9 uses the slots of the input
 * pattern; diagonal slot = inverted pivot.  Returns OPMGPU_ESINGULAR on a zero pivot. */
int oracle_ilu0(int nb, const int32_t* rowptr, const int32_t* col, const double* val9,
                const int32_t* position, int single_precision, double* lu9);
/* v = w * U^-1 L^-1 d (ParallelOverlappingILU0::apply), block-interleaved vectors. */
void oracle_ilu0_apply(int nb, const int32_t* rowptr, const int32_t* col, const double* lu9,
                       const int32_t* position, double relax, int single_precision,
                       const double* d3, double* v3);

/* Dune::BiCGSTABSolver::apply with the ILU0 above as right... (sic: dune applies the
 * preconditioner to the search directions) preconditioner, x0 = 0
 * (ISTLSolver.hpp:250-274; NewtonIterationBlackoilInterleaved.cpp:272-276).
 * Returns OPMGPU_OK / OPMGPU_ELINSOLVE / OPMGPU_EBREAKDOWN / OPMGPU_ESINGULAR.
 * hist (may be NULL) receives ||r||_2 after every half step, at most nhist entries. */
int oracle_bicgstab_ilu0(int nb, const int32_t* rowptr, const int32_t* col, const double* val9,
                         const double* rhs3, const int32_t* position, const opmgpu_params* prm,
                         int single_precision, double* x3, int* iters, double* reduction,
                         double* hist, int nhist, int* nhist_out);

/* Relative-permeability hysteresis (EclHysteresisTwoPhaseLaw, Carlson / KR only; opmgpu_grid.imbnum): history planes used by every
 * later call (NULL = none), and the once-per-report-step update (EclDefaultMaterial::updateHysteresis, "inconsistent" form:
 * krnSw = 1 - So resp. 1 - Sg; EclHysteresisTwoPhaseLawParams::update + updateDynamicParams_) in place on four [nc] arrays.
 * Restated from opm-material's published code; the reference holds no vectors for it: parity unpinned. */
void oracle_set_hysteresis(const double* mdc_ow, const double* mdc_go, const double* d_ow, const double* d_go);
void oracle_update_hysteresis(const opmgpu_grid* g, const opmgpu_tables* t, const double* sat, double* mdc_ow, double* mdc_go,
                              double* d_ow, double* d_go);

/* timing helper for bench.py's cpu_baseline: threads used by the OpenMP-able loops
 * (assembly, SpMV, vector updates; the ILU sweeps stay sequential like the reference's). */
void oracle_set_threads(int n);
int oracle_get_threads(void);

#ifdef __cplusplus
}
#endif
#endif
