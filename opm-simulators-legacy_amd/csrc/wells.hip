// wells.hip -- standard well model on the device.
//
// Restates, per well and without leaving the GPU, what Opm::StandardWells does with AutoDiffBlocks:
//   computeWellFlux                    opm/autodiff/StandardWells_impl.hpp:396-571
//   addWellFluxEq / addWellControlEq   :806-829 / :836-998   (BHP and SURFACE_RATE controls, dead wells)
//   updateWellState                    :611-650              computeWellConnectionPressures :223-298
//   WellDensitySegmented               WellDensitySegmented.cpp:66-181
//   addWellContributionToMassBalanceEq opm/autodiff/BlackoilModelBase_impl.hpp:953-975
//   eliminateVariable / recoverVariable opm/autodiff/NewtonIterationUtilities.cpp:45-184
// Difference by design: the reference forms the Schur complement S = A - B D^-1 C explicitly, which fills the reservoir
// matrix with a dense clique per well.  Here the same reduced system is kept in factored form: the perforated cells' own
// derivatives go into the diagonal blocks, everything that couples different perforations of a well is a rank-7 operator
// P_w Q_w (3 mixture fractions + 4 well unknowns) applied matrix-free by the solver (LowRankOp, linsolver.hpp).
// One workgroup per well; derivatives by a small forward-AD type (the work is O(nperf), AD costs nothing here).
// Round 2: the control logic of the reference runs on the device too --
//   updateWellControls / updateWellStateWithTarget   StandardWells_impl.hpp:709-800 / :1452-1550   (k_well_controls)
//   solveWellEq (explicit well pre-solve, default on) BlackoilModelBase_impl.hpp:1018-1133          (k_well_assemble<PRE> + k_well_presolve_step)
//   THP control through VFP tables                   :655-700, :895-960; VFPProd/InjPropertiesLegacy.cpp   (vfp_* below)
//   PVT at the average well-block pressure           :218-296                                       (k_well_avg_press + k_perf_pvt)
//   RESERVOIR_RATE conversion coefficients          RateConverterLegacy.hpp:495-548, :718-768        (k_voidage_coeff, region_state_sums in blackoil.hip)
// Not restated: group controls / guide rates, efficiency factors.
#include "blackoil.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace opmgpu {

namespace {

template <int N> struct Du { double v; double d[N]; };

template <int N> __device__ __forceinline__ Du<N> du_const(double v) { Du<N> r; r.v = v; for (int i = 0; i < N; ++i) r.d[i] = 0.0; return r; }
template <int N> __device__ __forceinline__ Du<N> du_var(double v, int k) { Du<N> r = du_const<N>(v); r.d[k] = 1.0; return r; }
template <int N> __device__ __forceinline__ Du<N> operator+(const Du<N>& a, const Du<N>& b) { Du<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> __device__ __forceinline__ Du<N> operator-(const Du<N>& a, const Du<N>& b) { Du<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> __device__ __forceinline__ Du<N> operator*(const Du<N>& a, const Du<N>& b) { Du<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + b.d[i] * a.v; return r; }
template <int N> __device__ __forceinline__ Du<N> operator/(const Du<N>& a, const Du<N>& b) { Du<N> r; r.v = a.v / b.v; for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - b.d[i] * r.v) / b.v; return r; }
template <int N> __device__ __forceinline__ Du<N> operator*(double s, const Du<N>& a) { Du<N> r; r.v = s * a.v; for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i]; return r; }
template <int N> __device__ __forceinline__ Du<N> operator-(double s, const Du<N>& a) { Du<N> r; r.v = s - a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }

// perforation property k of OPMGPU_PERF_K (value, d/dP, d/dSw, d/dXvar) as a dual over the first three variables
template <int N> __device__ __forceinline__ Du<N> perf_q(const double* __restrict__ pp, int k)
{
    Du<N> r = du_const<N>(pp[4 * k]);
    r.d[0] = pp[4 * k + 1]; r.d[1] = pp[4 * k + 2]; r.d[2] = pp[4 * k + 3];
    return r;
}

// variables: 0..2 = (P, Sw, Xvar) of the perforated cell, 3 = bhp, 4..6 = mixture fractions (only with N == 7)
template <int N> struct PerfFlux { Du<N> cq_ps[3]; Du<N> cqt_i; };

template <int N>
__device__ void perf_flux(const double* __restrict__ pp, double bhp_v, double cdp, double Tw, bool sel_inj, bool sel_prod, PerfFlux<N>& f)
{
    const Du<N> p = perf_q<N>(pp, 0), rs = perf_q<N>(pp, 1), rv = perf_q<N>(pp, 2);
    const Du<N> b[3] = { perf_q<N>(pp, 3), perf_q<N>(pp, 4), perf_q<N>(pp, 5) };
    const Du<N> mob[3] = { perf_q<N>(pp, 6), perf_q<N>(pp, 7), perf_q<N>(pp, 8) };
    Du<N> bhp = du_var<N>(bhp_v, 3);
    const Du<N> drawdown = p - (bhp + du_const<N>(cdp));
    const double sp = sel_prod ? Tw : 0.0, si = sel_inj ? Tw : 0.0;
    for (int a = 0; a < 3; ++a) f.cq_ps[a] = b[a] * ((-sp) * (mob[a] * drawdown));           // :457-463, flow INTO the wellbore
    const Du<N> oil = f.cq_ps[1], gas = f.cq_ps[2];
    f.cq_ps[2] = f.cq_ps[2] + rs * oil;                                                         // :466-474
    f.cq_ps[1] = f.cq_ps[1] + rv * gas;
    f.cqt_i = (-si) * ((mob[0] + mob[1] + mob[2]) * drawdown);                                  // :481-486, flow OUT of the wellbore
}


// ---- VFP tables on the device (THP control) ------------------------------------------------------------------------------
// meta[t][12]: id, is_injector, flo_type, wfr_type, gfr_type, nflo, nthp, nwfr, ngfr, nalq, axis offset, data offset (into blob);
// axes are stored flo | thp | wfr | gfr | alq, data [thp][wfr][gfr][alq][flo] (VFPProdTable) -- an injector table has one-point
// wfr / gfr / alq axes, so the same five-dimensional code serves both kinds.
struct VfpArgs { int ntab; const int32_t* meta; const double* datum; const double* blob; };
enum { VM_ID = 0, VM_INJ, VM_FLO_T, VM_WFR_T, VM_GFR_T, VM_NFLO, VM_NTHP, VM_NWFR, VM_NGFR, VM_NALQ, VM_AXIS, VM_DATA, VM_COUNT };
constexpr int kVfpMaxThp = 64;

struct VfpInterp { int i0, i1; double inv, f; };
// detail::findInterpData (opm-simulators VFPHelpers.hpp, pinned by tests/test_vfp.py on the host restatement)
__device__ VfpInterp vfp_find(double v, const double* __restrict__ ax, int n)
{
    VfpInterp r; r.i0 = 0; r.i1 = 0; r.inv = 0.0; r.f = 0.0;
    if (n == 1) return r;
    if (v < ax[0]) { r.i0 = 0; r.i1 = 1; }
    else if (v >= ax[n - 1]) { r.i0 = n - 2; r.i1 = n - 1; }
    else { int i = 1; while (!(ax[i] >= v)) ++i; r.i0 = i - 1; r.i1 = i; }
    const double a = ax[r.i0], b = ax[r.i1];
    if (b > a) { r.inv = 1.0 / (b - a); r.f = (v - a) * r.inv; }
    return r;
}
// detail::interpolate: multilinear value on the cell it[] = (thp, wfr, gfr, alq, flo) and the partial derivative along each axis
__device__ double vfp_interp5(const double* __restrict__ data, const int32_t* __restrict__ m, const VfpInterp it[5], double d[5])
{
    const int n[5] = { m[VM_NTHP], m[VM_NWFR], m[VM_NGFR], m[VM_NALQ], m[VM_NFLO] };
    double val = 0.0;
    for (int k = 0; k < 5; ++k) d[k] = 0.0;
    for (int c = 0; c < 32; ++c) {
        long idx = 0;
        double w[5];
        for (int k = 0; k < 5; ++k) {
            const int bit = (c >> (4 - k)) & 1;
            idx = idx * n[k] + (bit ? it[k].i1 : it[k].i0);
            w[k] = bit ? it[k].f : 1.0 - it[k].f;
        }
        const double v = data[idx];
        val += w[0] * w[1] * w[2] * w[3] * w[4] * v;
        for (int k = 0; k < 5; ++k) {
            double pw = ((c >> (4 - k)) & 1) ? it[k].inv : -it[k].inv;
            for (int j = 0; j < 5; ++j) if (j != k) pw *= w[j];
            d[k] += pw * v;
        }
    }
    return val;
}
__device__ __forceinline__ double vfp_ratio(double num, double den) { if (den == 0.0) return 0.0; const double r = num / den; return isfinite(r) ? r : 0.0; }
// the table variables flo / wfr / gfr of the rates q = (aqua, liquid, vapour) and their gradients (getFlo / getWFR / getGFR + zeroIfNanInf)
__device__ void vfp_vars(const int32_t* __restrict__ m, const double q[3], double& flo, double& wfr, double& gfr, double dflo[3], double dwfr[3], double dgfr[3])
{
    for (int k = 0; k < 3; ++k) { dflo[k] = 0.0; dwfr[k] = 0.0; dgfr[k] = 0.0; }
    const double a = q[0], l = q[1], v = q[2];
    switch (m[VM_FLO_T]) { case 0: flo = l; dflo[1] = 1.0; break; case 1: flo = a + l; dflo[0] = 1.0; dflo[1] = 1.0; break; default: flo = v; dflo[2] = 1.0; }
    double num, den, dn[3] = { 0, 0, 0 }, dd[3] = { 0, 0, 0 };
    switch (m[VM_WFR_T]) { case 0: num = a; den = l; dn[0] = 1; dd[1] = 1; break; case 1: num = a; den = a + l; dn[0] = 1; dd[0] = 1; dd[1] = 1; break; default: num = a; den = v; dn[0] = 1; dd[2] = 1; }
    wfr = vfp_ratio(num, den);
    if (den != 0.0 && isfinite(num / den)) for (int k = 0; k < 3; ++k) dwfr[k] = (dn[k] - wfr * dd[k]) / den;
    for (int k = 0; k < 3; ++k) { dn[k] = 0.0; dd[k] = 0.0; }
    switch (m[VM_GFR_T]) { case 0: num = v; den = l; dn[2] = 1; dd[1] = 1; break; case 1: num = v; den = l + a; dn[2] = 1; dd[0] = 1; dd[1] = 1; break; default: num = l; den = v; dn[1] = 1; dd[2] = 1; }
    gfr = vfp_ratio(num, den);
    if (den != 0.0 && isfinite(num / den)) for (int k = 0; k < 3; ++k) dgfr[k] = (dn[k] - gfr * dd[k]) / den;
}
// VFPProd/InjPropertiesLegacy::bhp: value and d bhp / d (aqua, liquid, vapour).  Producer rates are negative: the lookup negates flo
// and the flo term of the gradient carries a minus (VFPProdPropertiesLegacy.cpp:100, :146); injectors plus (VFPInjPropertiesLegacy.cpp:92, :119).
__device__ double vfp_bhp(const VfpArgs& V, int t, const double q[3], double thp, double alq, double dq[3])
{
    const int32_t* m = V.meta + VM_COUNT * t;
    const double* ax = V.blob + m[VM_AXIS];
    const double* a_flo = ax; const double* a_thp = a_flo + m[VM_NFLO]; const double* a_wfr = a_thp + m[VM_NTHP];
    const double* a_gfr = a_wfr + m[VM_NWFR]; const double* a_alq = a_gfr + m[VM_NGFR];
    double flo, wfr, gfr, dflo[3], dwfr[3], dgfr[3];
    vfp_vars(m, q, flo, wfr, gfr, dflo, dwfr, dgfr);
    const double sgn = m[VM_INJ] ? 1.0 : -1.0;
    VfpInterp it[5] = { vfp_find(thp, a_thp, m[VM_NTHP]), vfp_find(wfr, a_wfr, m[VM_NWFR]), vfp_find(gfr, a_gfr, m[VM_NGFR]),
                        vfp_find(alq, a_alq, m[VM_NALQ]), vfp_find(sgn * flo, a_flo, m[VM_NFLO]) };
    double d[5];
    const double v = vfp_interp5(V.blob + m[VM_DATA], m, it, d);
    if (dq) for (int k = 0; k < 3; ++k) dq[k] = d[1] * dwfr[k] + d[2] * dgfr[k] + sgn * d[4] * dflo[k];
    return v;
}
__device__ __forceinline__ double vfp_find_x(double x0, double x1, double y0, double y1, double y) { return x0 + ((x1 - x0) / (y1 - y0)) * (y - y0); }
// VFPProd/InjPropertiesLegacy::thp -> detail::findTHP: invert the piecewise-linear bhp(thp axis) at the given rates
__device__ double vfp_thp(const VfpArgs& V, int t, const double q[3], double bhp, double alq)
{
    const int32_t* m = V.meta + VM_COUNT * t;
    const int n = m[VM_NTHP];
    const double* ta = V.blob + m[VM_AXIS] + m[VM_NFLO];
    double b[kVfpMaxThp];
    bool sorted = true;
    for (int i = 0; i < n; ++i) { b[i] = vfp_bhp(V, t, q, ta[i], alq, nullptr); if (i > 0 && b[i - 1] > b[i]) sorted = false; }
    if (n < 2) return ta[0];
    int found = -1;
    for (int i = 0; i < n - 1; ++i) if (b[i] < bhp && bhp <= b[i + 1]) { found = i; break; }
    if (sorted) {
        if (bhp <= b[0]) return vfp_find_x(ta[0], ta[1], b[0], b[1], bhp);
        if (bhp > b[n - 1]) return vfp_find_x(ta[n - 2], ta[n - 1], b[n - 2], b[n - 1], bhp);
        return vfp_find_x(ta[found], ta[found + 1], b[found], b[found + 1], bhp);
    }
    if (found >= 0) return vfp_find_x(ta[found], ta[found + 1], b[found], b[found + 1], bhp);
    if (bhp <= b[0]) return vfp_find_x(ta[0], ta[1], b[0], b[1], bhp);
    return vfp_find_x(ta[n - 2], ta[n - 1], b[n - 2], b[n - 1], bhp);
}

struct WellArgs {
    int nbp;
    const int32_t *connpos, *perf_row, *type, *allow_cf, *ctrl_type;
    const double *WI, *comp_frac, *ctrl_target, *ctrl_distr, *depth_ref, *z_perf, *surf_dens_perf;
    const int32_t *ctrl_ptr, *ctrl_vfp, *thp_ctrl;       // controls of well w: [ctrl_ptr[w], ctrl_ptr[w+1]); VFP table INDEX per control (-1); first THP control of the well (-1)
    const double* ctrl_alq;
    int32_t* current;            // [nw] currentControls(): index relative to ctrl_ptr[w]
    double* thp;                 // [nw]
    double* perf_dens;           // [nperf] well_perforation_densities_
    const double* perf_pvt;      // [nperf][5] b_w b_o b_g rsSat rvSat at the average well-block pressure (k_perf_pvt)
    double* wdy;                 // [nw][4] recovered Newton increment of the well unknowns
    double* ctrl_row;            // [nw][4] d control equation / d (qs, bhp)
    VfpArgs V;
    double gravity;
    const double* perf;          // [nperf][OPMGPU_PERF_K]
    double* wstate;              // [nw][4] qs(3), bhp
    double* cdp;                 // [nperf]
    double* perf_rates;          // [nperf][3]
    double* perf_press;          // [nperf]
    double* P; double* Q;        // [nperf][21]
    double* Fsave;               // [nperf][9+9+3]: F_i, M_i, fb_i
    double* wellE;               // [nw][4]
    double* Dinv;                // [nw][16]
};

// average well-block pressure per perforation (computePropertiesForWellConnectionPressures, StandardWells_impl.hpp:228-237):
// mean of the perforation's pressure and the one above it (bhp for the first), both from the WELL STATE
__global__ void k_well_avg_press(int nw, WellArgs A, const int32_t* __restrict__ gate, double* __restrict__ avg)
{
    if (gate && !*gate) return;
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    for (int perf = A.connpos[w]; perf < A.connpos[w + 1]; ++perf) {
        const double p_above = perf == A.connpos[w] ? A.wstate[4 * w + 3] : A.perf_press[perf - 1];
        avg[perf] = (A.perf_press[perf] + p_above) / 2;
    }
}

// computeWellConnectionPressures: WellDensitySegmented::computeConnectionDensities + computeConnectionPressureDelta.  One wavefront per
// well: the lanes stage a tile of perforation data in LDS with parallel loads, lane 0 then walks the tile in the reference's order (the
// running sums are sequential by definition; a thread fetching every perforation's operands itself spends 1.6 us per perforation waiting
// for them: 164 us for a 100-perforation well, 8x this version).
constexpr int kCdpTile = 128;
__global__ __launch_bounds__(64) void k_well_cdp(int nw, WellArgs A, double gravity, const int32_t* __restrict__ gate)
{
    __shared__ double s_rate[3 * kCdpTile], s_pvt[5 * kCdpTile], s_sd[3 * kCdpTile], s_z[kCdpTile], s_dens[kCdpTile], s_cdp[kCdpTile];
    if (gate && !*gate) return;
    const int w = blockIdx.x, lane = threadIdx.x;
    if (w >= nw) return;
    const int lo = A.connpos[w], hi = A.connpos[w + 1];
    // q_out[perf] = flow out of the segment above perforation perf = sum_{k >= perf} (-rate_k) (the reference fills it bottom to
    // top, WellDensitySegmented.cpp:83-95); here: total at the top first, then q_out[perf + 1] = q_out[perf] + rate[perf] going down.
    double q[3] = { 0.0, 0.0, 0.0 };
    for (int t1 = hi; t1 > lo; t1 -= kCdpTile) {
        const int t0 = t1 - kCdpTile > lo ? t1 - kCdpTile : lo, nt = t1 - t0;
        for (int i = lane; i < 3 * nt; i += 64) s_rate[i] = A.perf_rates[3 * long(t0) + i];
        __syncthreads();
        if (lane == 0) for (int i = nt - 1; i >= 0; --i) for (int a = 0; a < 3; ++a) q[a] -= s_rate[3 * i + a];
        __syncthreads();
    }
    double run = 0.0, z_above = A.depth_ref[w];
    for (int t0 = lo; t0 < hi; t0 += kCdpTile) {
        const int nt = hi - t0 < kCdpTile ? hi - t0 : kCdpTile;
        for (int i = lane; i < 3 * nt; i += 64) { s_rate[i] = A.perf_rates[3 * long(t0) + i]; s_sd[i] = A.surf_dens_perf[3 * long(t0) + i]; }
        for (int i = lane; i < 5 * nt; i += 64) s_pvt[i] = A.perf_pvt[5 * long(t0) + i];
        for (int i = lane; i < nt; i += 64) s_z[i] = A.z_perf[t0 + i];
        __syncthreads();
        if (lane == 0) {
            for (int i = 0; i < nt; ++i) {
                const double* pv = s_pvt + 5 * i;
                const double tot = q[0] + q[1] + q[2];
                double mix[3], x[3];
                for (int a = 0; a < 3; ++a) { mix[a] = tot != 0.0 ? fabs(q[a] / tot) : A.comp_frac[3 * w + a]; x[a] = mix[a]; }
                const double rsmax = pv[3], rvmax = pv[4];
                double rs = 0.0, rv = 0.0;
                if (mix[1] > 0.0) rs = fmin(mix[2] / mix[1], rsmax);
                if (mix[2] > 0.0) rv = fmin(mix[1] / mix[2], rvmax);
                if (rs != 0.0) x[2] = (mix[2] - mix[1] * rs) / (1.0 - rs * rv);
                if (rv != 0.0) x[1] = (mix[1] - mix[2] * rv) / (1.0 - rs * rv);
                const double volrat = x[0] / pv[0] + x[1] / pv[1] + x[2] / pv[2];
                const double* sd = s_sd + 3 * i;
                const double dens = (sd[0] * mix[0] + sd[1] * mix[1] + sd[2] * mix[2]) / volrat;
                s_dens[i] = dens;
                run += (s_z[i] - z_above) * dens * gravity;
                z_above = s_z[i];
                s_cdp[i] = run;
                for (int a = 0; a < 3; ++a) q[a] += s_rate[3 * i + a];
            }
        }
        __syncthreads();
        for (int i = lane; i < nt; i += 64) { A.perf_dens[t0 + i] = s_dens[i]; A.cdp[t0 + i] = s_cdp[i]; }
        __syncthreads();
    }
}

// ---- control logic shared by k_well_controls and the pre-solve ---------------------------------------------------------------
// wellhelpers::computeHydrostaticCorrection with the density of the well's first perforation
__device__ __forceinline__ double vfp_dp(const WellArgs& A, int w, int t)
{
    if (A.connpos[w + 1] == A.connpos[w]) return 0.0;
    return A.perf_dens[A.connpos[w]] * A.gravity * (A.V.datum[t] - A.depth_ref[w]);
}
// updateWellStateWithTarget (StandardWells_impl.hpp:1452-1550)
__device__ void well_apply_target(const WellArgs& A, int w, int c)
{
    const int typ = A.ctrl_type[c];
    const double target = A.ctrl_target[c];
    double* ws = A.wstate + 4 * w;
    if (typ == OPMGPU_CTRL_BHP) ws[3] = target;
    else if (typ == OPMGPU_CTRL_THP) {
        const int t = A.ctrl_vfp[c];
        const double q[3] = { ws[0], ws[1], ws[2] };
        ws[3] = vfp_bhp(A.V, t, q, target, A.ctrl_alq[c], nullptr) - vfp_dp(A, w, t);
    } else if (typ == OPMGPU_CTRL_SURFACE_RATE) {
        const double* distr = A.ctrl_distr + 3 * c;
        if (A.type[w] == 0) { for (int a = 0; a < 3; ++a) if (A.comp_frac[3 * w + a] > 0.0) ws[a] = target * A.comp_frac[3 * w + a]; }
        else {
            int n = 0;
            for (int a = 0; a < 3; ++a) n += distr[a] > 0.0 ? 1 : 0;
            if (n < 2) for (int a = 0; a < 3; ++a) if (distr[a] > 0.0) ws[a] = target * distr[a];
        }
    }
}
// wellhelpers::constraintBroken (opm-simulators WellHelpers.hpp): injectors break a limit from above, producers from below
__device__ bool well_constraint_broken(const WellArgs& A, int w, int c)
{
    const int typ = A.ctrl_type[c];
    const double target = A.ctrl_target[c];
    const double* ws = A.wstate + 4 * w;
    double val;
    if (typ == OPMGPU_CTRL_BHP) val = ws[3];
    else if (typ == OPMGPU_CTRL_THP) val = A.thp[w];
    else { const double* d = A.ctrl_distr + 3 * c; val = ws[0] * d[0] + ws[1] * d[1] + ws[2] * d[2]; }
    return A.type[w] == 0 ? val > target : val < target;
}
// updateWellControls for one well (:709-800); returns false when no consistent control is found within 2 * ncontrols rounds
__device__ bool well_update_controls(const WellArgs& A, int w)
{
    const int c0 = A.ctrl_ptr[w], nwc = A.ctrl_ptr[w + 1] - c0;
    int current = A.current[w], rounds = 0;
    for (;;) {
        well_apply_target(A, w, c0 + current);
        int broken = -1;
        for (int k = 0; k < nwc; ++k) if (k != current && well_constraint_broken(A, w, c0 + k)) { broken = k; break; }
        if (broken >= 0) { current = broken; A.current[w] = current; }
        ++rounds;
        if (rounds > 2 * nwc) return false;
        if (broken < 0) return true;
    }
}
// updateWellState for one well (:611-700): rates, limited bhp, thp of a well that has a THP control
__device__ void well_apply_increment(const WellArgs& A, int w, const double dy[4], double dbhp_max_rel)
{
    double* ws = A.wstate + 4 * w;
    for (int a = 0; a < 3; ++a) ws[a] -= dy[a];
    const double d = dy[3], bhp = ws[3];
    const double sg = d > 0.0 ? 1.0 : (d < 0.0 ? -1.0 : 0.0);
    ws[3] = bhp - sg * fmin(fabs(d), fabs(bhp) * dbhp_max_rel);
    const int tc = A.thp_ctrl[w];
    if (tc >= 0) {
        const int t = A.ctrl_vfp[tc];
        const double q[3] = { ws[0], ws[1], ws[2] };
        A.thp[w] = vfp_thp(A.V, t, q, ws[3] + vfp_dp(A, w, t), A.ctrl_alq[tc]);
    }
}

// flags[]: 0 = error bits (2: singular D, 4: no consistent control, 8: NaN / too large well residual), 1 = pre-solve done,
// 2 = pre-solve converged, 3 = pre-solve iterations
enum { WF_ERR = 0, WF_DONE, WF_CONV, WF_ITS, WF_COUNT };

__global__ void k_well_controls(int nw, WellArgs A, int32_t* __restrict__ flags)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    if (!well_update_controls(A, w)) atomicOr(&flags[WF_ERR], 4);
}

// 4x4 inverse with partial pivoting (D of one well); returns false when singular
__device__ bool inv4(const double* m, double* out)
{
    double a[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { a[i][j] = m[4 * i + j]; a[i][4 + j] = i == j ? 1.0 : 0.0; }
    for (int p = 0; p < 4; ++p) {
        int piv = p;
        for (int i = p + 1; i < 4; ++i) if (fabs(a[i][p]) > fabs(a[piv][p])) piv = i;
        if (a[piv][p] == 0.0) return false;
        if (piv != p) for (int j = 0; j < 8; ++j) { const double t = a[p][j]; a[p][j] = a[piv][j]; a[piv][j] = t; }
        const double d = 1.0 / a[p][p];
        for (int j = 0; j < 8; ++j) a[p][j] *= d;
        for (int i = 0; i < 4; ++i) if (i != p) { const double f = a[i][p]; for (int j = 0; j < 8; ++j) a[i][j] -= f * a[p][j]; }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = a[i][4 + j];
    return true;
}

// well equations + reduced (Schur) contributions of one well by one workgroup.  PRE = the explicit well pre-solve (solveWellEq,
// BlackoilModelBase_impl.hpp:1043-1062): the reservoir is frozen, only E and D^-1 of the well are formed (nothing is written to the
// reservoir system).  Eout: where the four well-equation residuals go (PRE + published: agent-scope stores, read by other workgroups).
template <class MS, bool PRE>
__device__ void well_assemble_dev(const WellArgs& A, int w, const int32_t* __restrict__ slice_ptr, const int16_t* __restrict__ nlower,
                                  double s0, double s1, double s2, double* __restrict__ R, MS* __restrict__ Amat,
                                  double* __restrict__ rhs_extra, int32_t* __restrict__ flags, double* Eout, bool publish)
{
    __shared__ double sm[64];
    __shared__ int any_flag[2];
    __shared__ double wl[64];         // well-level values shared by all threads
    const int tid = threadIdx.x;
    const int lo = A.connpos[w], hi = A.connpos[w + 1];
    const double qs[3] = { A.wstate[4 * w], A.wstate[4 * w + 1], A.wstate[4 * w + 2] };
    const double bhp = A.wstate[4 * w + 3];
    const double compi[3] = { A.comp_frac[3 * w], A.comp_frac[3 * w + 1], A.comp_frac[3 * w + 2] };
    const double scale[3] = { s0, s1, s2 };
    // ---- phase 0: which perforations inject (drawdown < 0), cross-flow rule (:418-452) ----
    if (tid < 2) any_flag[tid] = 0;
    __syncthreads();
    for (int j = lo + tid; j < hi; j += kBlock) {
        const double dd = A.perf[long(j) * OPMGPU_PERF_K] - (bhp + A.cdp[j]);
        atomicOr(&any_flag[dd < 0.0 ? 0 : 1], 1);
    }
    __syncthreads();
    const bool any_inj = any_flag[0] != 0, any_prod = any_flag[1] != 0;
    const bool kill_prod = !A.allow_cf[w] && A.type[w] == 0 && any_inj;          // injector: no producing perforations
    const bool kill_inj = !A.allow_cf[w] && A.type[w] == 1 && any_prod && !kill_prod;
    // ---- phase A: sums of cq_ps and of d cq_ps / d bhp ----
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
    for (int j = lo + tid; j < hi; j += kBlock) {
        const double* pp = A.perf + long(j) * OPMGPU_PERF_K;
        const double dd = pp[0] - (bhp + A.cdp[j]);
        const bool si = dd < 0.0 && !kill_inj, sp = !(dd < 0.0) && !kill_prod;
        PerfFlux<4> f;
        perf_flux<4>(pp, bhp, A.cdp[j], A.WI[j], si, sp, f);
        for (int a = 0; a < 3; ++a) { acc[a] += f.cq_ps[a].v; acc[3 + a] += f.cq_ps[a].d[3]; }
    }
    block_sum<6>(acc, sm);
    // ---- phase B: mixture in the wellbore (:489-519) ----
    if (tid == 0) {
        double wbq[3], wbqt = 0.0, inj[3], dinj[3];
        for (int a = 0; a < 3; ++a) { inj[a] = qs[a] > 0.0 ? qs[a] : 0.0; dinj[a] = qs[a] > 0.0 ? 1.0 : 0.0; wbq[a] = compi[a] * inj[a] - acc[a]; wbqt += wbq[a]; }
        const bool alive = wbqt != 0.0;
        double Sb = acc[3] + acc[4] + acc[5];
        for (int a = 0; a < 3; ++a) {
            const double cm = alive ? wbq[a] / wbqt : compi[a];
            wl[a] = cm;
            for (int b = 0; b < 3; ++b)       // g_q[a][b] = d cmix_a / d qs_b
                wl[3 + 3 * a + b] = alive ? ((a == b ? compi[b] * dinj[b] : 0.0) - cm * compi[b] * dinj[b]) / wbqt : 0.0;
            wl[12 + a] = alive ? (-acc[3 + a] + cm * Sb) / wbqt : 0.0;      // g_b[a] = d cmix_a / d bhp
        }
        wl[15] = alive ? 1.0 : 0.0;
        wl[16] = wbqt;
    }
    __syncthreads();
    const double cmix[3] = { wl[0], wl[1], wl[2] };
    const bool alive = wl[15] != 0.0;
    const double wbqt = wl[16];
    // ---- phase C: component rates per perforation with all partials; residual, diagonal blocks, sums ----
    double acc2[15];
    for (int k = 0; k < 15; ++k) acc2[k] = 0.0;          // 0..2 sum cq_s, 3..11 sum M, 12..14 sum fb
    for (int j = lo + tid; j < hi; j += kBlock) {
        const double* pp = A.perf + long(j) * OPMGPU_PERF_K;
        const double dd = pp[0] - (bhp + A.cdp[j]);
        const bool si = dd < 0.0 && !kill_inj, sp = !(dd < 0.0) && !kill_prod;
        PerfFlux<7> f;
        perf_flux<7>(pp, bhp, A.cdp[j], A.WI[j], si, sp, f);
        const Du<7> rs = perf_q<7>(pp, 1), rv = perf_q<7>(pp, 2);
        const Du<7> b[3] = { perf_q<7>(pp, 3), perf_q<7>(pp, 4), perf_q<7>(pp, 5) };
        const Du<7> cm[3] = { du_var<7>(cmix[0], 4), du_var<7>(cmix[1], 5), du_var<7>(cmix[2], 6) };
        const Du<7> d = 1.0 - rv * rs;
        const Du<7> vol = cm[0] / b[0] + ((cm[1] - rv * cm[2]) / d) / b[1] + ((cm[2] - rs * cm[1]) / d) / b[2];     // :521-551
        const Du<7> cqt_is = f.cqt_i / vol;
        Du<7> cq_s[3];
        for (int a = 0; a < 3; ++a) cq_s[a] = f.cq_ps[a] + cm[a] * cqt_is;                                          // :553-560
        const int row = A.perf_row[j];
        if (PRE) {
            for (int a = 0; a < 3; ++a) {
                A.perf_rates[3 * j + a] = cq_s[a].v;                                 // updatePerfPhaseRatesAndPressures (:1059)
                acc2[a] += cq_s[a].v;
                acc2[12 + a] += cq_s[a].d[3];
                for (int v = 0; v < 3; ++v) acc2[3 + 3 * a + v] += cq_s[a].d[4 + v];
            }
            A.perf_press[j] = bhp + A.cdp[j];
            continue;
        }
        MS* dptr = Amat + long(slice_ptr[row >> 6] + nlower[row]) * 576 + (row & 63);
        double* Qj = A.Q + 21 * long(j); double* Fs = A.Fsave + 21 * long(j);
        // H = d cq_ps / d cell, G = d cmix / d cell = (-H + cmix (1^T H)) / wbqt
        for (int v = 0; v < 3; ++v) {
            const double hs = f.cq_ps[0].d[v] + f.cq_ps[1].d[v] + f.cq_ps[2].d[v];
            for (int a = 0; a < 3; ++a) Qj[3 * a + v] = alive ? (-f.cq_ps[a].d[v] + cmix[a] * hs) / wbqt : 0.0;
        }
        for (int a = 0; a < 3; ++a) {
            R[long(a) * A.nbp + row] -= cq_s[a].v;                                   // addWellContributionToMassBalanceEq
            A.perf_rates[3 * j + a] = cq_s[a].v;
            acc2[a] += cq_s[a].v;
            acc2[12 + a] += cq_s[a].d[3];
            Fs[18 + a] = cq_s[a].d[3];
            for (int v = 0; v < 3; ++v) {
                dptr[(3 * a + v) * 64] = MS(double(dptr[(3 * a + v) * 64]) - scale[a] * cq_s[a].d[v]);    // own-cell part of -d cq_s / d cell
                Fs[3 * a + v] = cq_s[a].d[v];
                Fs[9 + 3 * a + v] = cq_s[a].d[4 + v];
                acc2[3 + 3 * a + v] += cq_s[a].d[4 + v];
            }
        }
        A.perf_press[j] = bhp + A.cdp[j];
    }
    __syncthreads();
    block_sum<15>(acc2, sm);
    // ---- phase D: well equations E, D = dE/d(qs, bhp), D^-1 ----
    if (tid == 0) {
        const double* gq = wl + 3; const double* gb = wl + 12;
        const double* Ms = acc2 + 3; const double* fbs = acc2 + 12;
        double D[16], E[4];
        for (int a = 0; a < 3; ++a) {
            E[a] = qs[a] - acc2[a];                                                  // addWellFluxEq
            for (int b = 0; b < 3; ++b) {
                double mg = 0.0;
                for (int c = 0; c < 3; ++c) mg += Ms[3 * a + c] * gq[3 * c + b];
                D[4 * a + b] = (a == b ? 1.0 : 0.0) - mg;
            }
            double mgb = 0.0;
            for (int c = 0; c < 3; ++c) mgb += Ms[3 * a + c] * gb[c];
            D[4 * a + 3] = -(fbs[a] + mgb);
        }
        // addWellControlEq (:836-998) for the well's CURRENT control: dead well -> sum of rates; BHP; THP through the VFP table; rate
        const int cc = A.ctrl_ptr[w] + A.current[w];
        const int ctyp = A.ctrl_type[cc];
        if (!alive) { E[3] = qs[0] + qs[1] + qs[2]; D[12] = 1.0; D[13] = 1.0; D[14] = 1.0; D[15] = 0.0; }
        else if (ctyp == OPMGPU_CTRL_BHP) { E[3] = bhp - A.ctrl_target[cc]; D[12] = 0.0; D[13] = 0.0; D[14] = 0.0; D[15] = 1.0; }
        else if (ctyp == OPMGPU_CTRL_THP) {         // bhp - bhp_from_thp(qs) + dp (:944-949)
            const int t = A.ctrl_vfp[cc];
            double dq[3];
            const double bt = vfp_bhp(A.V, t, qs, A.ctrl_target[cc], A.ctrl_alq[cc], dq) - vfp_dp(A, w, t);
            E[3] = bhp - bt; D[12] = -dq[0]; D[13] = -dq[1]; D[14] = -dq[2]; D[15] = 1.0;
        } else {
            const double* ds = A.ctrl_distr + 3 * cc;
            E[3] = ds[0] * qs[0] + ds[1] * qs[1] + ds[2] * qs[2] - A.ctrl_target[cc];
            D[12] = ds[0]; D[13] = ds[1]; D[14] = ds[2]; D[15] = 0.0;
        }
        double Di[16];
        if (!inv4(D, Di)) { atomicOr(&flags[WF_ERR], 2); for (int k = 0; k < 16; ++k) Di[k] = 0.0; }
        for (int k = 0; k < 16; ++k) { wl[20 + k] = Di[k]; A.Dinv[16 * w + k] = Di[k]; }
        if (!PRE) for (int k = 0; k < 4; ++k) A.ctrl_row[4 * w + k] = D[12 + k];      // gradient of the control equation (bordered pressure system, linsolver.hip)
        for (int k = 0; k < 4; ++k) {
            wl[36 + k] = E[k];
            if (publish) __hip_atomic_store(&Eout[k], E[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else Eout[k] = E[k];
        }
        for (int k = 0; k < 9; ++k) wl[40 + k] = Ms[k];
    }
    __syncthreads();
    if (PRE) return;
    // ---- phase E: P_i = -[M_i | B_i D^-1], Q_i = [G_i ; C_i], rhs extra = -(B_i D^-1 E) ----
    {
        const double* gq = wl + 3; const double* gb = wl + 12; const double* Di = wl + 20; const double* E = wl + 36; const double* Ms = wl + 40;
        for (int j = lo + tid; j < hi; j += kBlock) {
            const double* Fs = A.Fsave + 21 * long(j);
            const double* F = Fs; const double* M = Fs + 9; const double* fb = Fs + 18;
            double* Pj = A.P + 21 * long(j); double* Qj = A.Q + 21 * long(j);
            double B[12];                                       // B_i = -[M_i g_q | fb_i + M_i g_b]  (3 x 4)
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) { double s = 0.0; for (int c = 0; c < 3; ++c) s += M[3 * a + c] * gq[3 * c + b]; B[4 * a + b] = -s; }
                double s = 0.0; for (int c = 0; c < 3; ++c) s += M[3 * a + c] * gb[c];
                B[4 * a + 3] = -(fb[a] + s);
            }
            const int row = A.perf_row[j];
            for (int a = 0; a < 3; ++a) {
                double bde = 0.0;
                for (int k = 0; k < 4; ++k) {
                    double bd = 0.0;
                    for (int c = 0; c < 4; ++c) bd += B[4 * a + c] * Di[4 * c + k];
                    Pj[7 * a + 3 + k] = -scale[a] * bd;
                    bde += bd * E[k];
                }
                for (int c = 0; c < 3; ++c) Pj[7 * a + c] = -scale[a] * M[3 * a + c];
                rhs_extra[long(a) * A.nbp + row] = -bde;
            }
            // C_i rows 0..2 = -(F_i + Ms G_i); row 3 (control equation) = 0.  G_i is already in Q rows 0..2.
            double G[9];
            for (int k = 0; k < 9; ++k) G[k] = Qj[k];
            for (int a = 0; a < 3; ++a)
                for (int v = 0; v < 3; ++v) {
                    double s = 0.0;
                    for (int c = 0; c < 3; ++c) s += Ms[3 * a + c] * G[3 * c + v];
                    Qj[9 + 3 * a + v] = -(F[3 * a + v] + s);
                }
            Qj[18] = 0.0; Qj[19] = 0.0; Qj[20] = 0.0;
        }
    }
}

template <class MS, bool PRE>
__global__ __launch_bounds__(kBlock) void k_well_assemble(WellArgs A, const int32_t* __restrict__ slice_ptr, const int16_t* __restrict__ nlower,
                                                          double s0, double s1, double s2, double* __restrict__ R, MS* __restrict__ Amat,
                                                          double* __restrict__ rhs_extra, int32_t* __restrict__ flags)
{
    if (PRE && flags[WF_DONE]) return;
    well_assemble_dev<MS, PRE>(A, blockIdx.x, slice_ptr, nlower, s0, s1, s2, R, Amat, rhs_extra, flags, A.wellE + 4 * blockIdx.x, false);
}

// The whole pre-solve loop in ONE launch when the wells fit one resident grid (one workgroup per well, nw <= kFusedWells): per
// iteration every workgroup assembles its well, publishes its four residuals, all meet at a counter barrier, every workgroup takes
// the SAME convergence decision from all wells' residuals and updates its own well.  The residual buffer alternates with the iteration
// parity, so one barrier per iteration suffices.  Replaces 32 dependent launches (~150 us) by one (~5 us + ~4 us per iteration).
// Spins are bounded: a workgroup that gives up raises error bit 16 and every workgroup leaves at its next poll.
constexpr int kFusedWells = 256;
__device__ __forceinline__ bool presolve_barrier(int32_t* counter, int target, int32_t* flags)
{
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int good = 1;
        for (long spins = 0; __hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target; ++spins) {
            __builtin_amdgcn_s_sleep(4);
            if (spins > (1L << 22) || (__hip_atomic_load(&flags[WF_ERR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 16)) {
                __hip_atomic_fetch_or(&flags[WF_ERR], 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); good = 0; break;
            }
        }
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}
__global__ __launch_bounds__(kBlock) void k_well_presolve_fused(int nw, WellArgs A, const int32_t* __restrict__ slice_ptr, const int16_t* __restrict__ nlower,
                                                                const double* __restrict__ bsums, double ncells, double tol_wells, double tol_ctrl,
                                                                double max_resid, double dbhp_max_rel, int max_it, int32_t* __restrict__ flags,
                                                                int32_t* __restrict__ counter, double* __restrict__ Ebuf /* [2][nw][4] */)
{
    const int w = blockIdx.x, tid = threadIdx.x;
    __shared__ int decision;         // 0 continue, 1 converged, 2 numerical failure
    int it = 0;
    for (;;) {
        double* Eme = Ebuf + (size_t(it & 1) * nw + w) * 4;
        well_assemble_dev<double, true>(A, w, slice_ptr, nlower, 1.0, 1.0, 1.0, nullptr, (double*)nullptr, nullptr, flags, Eme, true);
        if (!presolve_barrier(counter, nw * (it + 1), flags)) return;
        // getWellConvergence over all wells (every workgroup computes the same numbers)
        if (tid < 64) {
            double mx[4] = { 0, 0, 0, 0 }; int bad = 0;
            const double* Eall = Ebuf + size_t(it & 1) * nw * 4;
            for (int v = tid; v < nw; v += 64) for (int k = 0; k < 4; ++k) {
                const double e = fabs(__hip_atomic_load(&Eall[4 * v + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (!(e == e)) bad = 1; mx[k] = fmax(mx[k], e);
            }
            for (int k = 0; k < 4; ++k) mx[k] = wave_max(mx[k]);
            bad = __any(bad);
            if (tid == 0) {
                bool conv = true, toolarge = false;
                for (int a = 0; a < 3; ++a) { const double wf = (bsums[a] / ncells) * mx[a]; conv = conv && wf < tol_wells; toolarge = toolarge || wf > max_resid || !(wf == wf); }
                conv = conv && mx[3] < tol_ctrl;
                decision = (bad || toolarge) ? 2 : (conv ? 1 : 0);
            }
        }
        __syncthreads();
        const int dec = decision;
        if (dec != 0) {
            if (w == 0 && tid == 0) { if (dec == 2) atomicOr(&flags[WF_ERR], 8); flags[WF_CONV] = dec == 1 ? 1 : 0; flags[WF_ITS] = it; flags[WF_DONE] = 1; }
            return;
        }
        ++it;
        if (tid == 0) {
            const double* Di = A.Dinv + 16 * w;
            double E[4], dy[4];
            for (int k = 0; k < 4; ++k) E[k] = Eme[k];
            for (int k = 0; k < 4; ++k) { dy[k] = 0.0; for (int c = 0; c < 4; ++c) dy[k] += Di[4 * k + c] * E[c]; }
            well_apply_increment(A, w, dy, dbhp_max_rel);
            if (!well_update_controls(A, w)) atomicOr(&flags[WF_ERR], 4);
            __threadfence();
        }
        __syncthreads();
        if (it >= max_it) { if (w == 0 && tid == 0) { flags[WF_CONV] = 0; flags[WF_ITS] = it; flags[WF_DONE] = 1; } return; }
    }
}

__global__ void k_well_flag_or(int32_t* f, int bits) { atomicOr(f, bits); }
// Fallback of the fused pre-solve: its counter barrier needs all nw workgroups resident together, which the launch guarantees only against
// the kernel's own occupancy (fused_presolve_capacity) -- other streams' and other processes' kernels give their slots back, but slowly enough
// under co-tenancy (several ranks rehearsing on one GPU) for a workgroup's bounded spin to give up (error bit 16).  That is a SCHEDULING
// condition, not a numerical one: this kernel, launched right behind the fused one, does nothing unless bit 16 is up; then it clears the
// bit, restores the well state the pre-solve started from and runs the whole loop (the same arithmetic: assemble every well, one decision
// from all residuals, update every well) inside ONE workgroup, wells one after the other.  Slow, correct, and never a NumericalIssue.
__global__ __launch_bounds__(kBlock) void k_well_presolve_serial(int nw, int np, WellArgs A, const int32_t* __restrict__ slice_ptr, const int16_t* __restrict__ nlower,
                                                                 const double* __restrict__ bsums, double ncells, double tol_wells, double tol_ctrl,
                                                                 double max_resid, double dbhp_max_rel, int max_it, int32_t* __restrict__ flags,
                                                                 const double* __restrict__ snap, const int32_t* __restrict__ isnap, double* __restrict__ Ebuf /* [nw][4] */)
{
    const int tid = threadIdx.x;
    __shared__ int go, decision;
    if (tid == 0) go = (flags[WF_ERR] & 16) ? 1 : 0;
    __syncthreads();
    if (!go) return;
    // back to the state the pre-solve started from (layout of snap: k_well_snapshot)
    {
        double* parts[4] = { A.wstate, A.thp, A.perf_rates, A.perf_press };
        const int len[4] = { 4 * nw, nw, 3 * np, np };
        int off = 0;
        for (int k = 0; k < 4; ++k) { for (int i = tid; i < len[k]; i += kBlock) parts[k][i] = snap[off + i]; off += len[k]; }
        for (int i = tid; i < nw; i += kBlock) A.current[i] = isnap[i];
        if (tid == 0) { flags[WF_ERR] = isnap[nw] & ~16; flags[WF_DONE] = 0; flags[WF_CONV] = 0; flags[WF_ITS] = 0; }     // the whole word as snapshot before the fused attempt
    }
    __threadfence();
    __syncthreads();
    for (int it = 0;; ) {
        for (int w = 0; w < nw; ++w) {
            well_assemble_dev<double, true>(A, w, slice_ptr, nlower, 1.0, 1.0, 1.0, nullptr, (double*)nullptr, nullptr, flags, Ebuf + 4 * size_t(w), false);
            __threadfence();
            __syncthreads();
        }
        if (tid < 64) {
            double mx[4] = { 0, 0, 0, 0 }; int bad = 0;
            for (int v = tid; v < nw; v += 64) for (int k = 0; k < 4; ++k) { const double e = fabs(Ebuf[4 * v + k]); if (!(e == e)) bad = 1; mx[k] = fmax(mx[k], e); }
            for (int k = 0; k < 4; ++k) mx[k] = wave_max(mx[k]);
            bad = __any(bad);
            if (tid == 0) {
                bool conv = true, toolarge = false;
                for (int a = 0; a < 3; ++a) { const double wf = (bsums[a] / ncells) * mx[a]; conv = conv && wf < tol_wells; toolarge = toolarge || wf > max_resid || !(wf == wf); }
                conv = conv && mx[3] < tol_ctrl;
                decision = (bad || toolarge) ? 2 : (conv ? 1 : 0);
            }
        }
        __syncthreads();
        const int dec = decision;
        if (dec != 0) {
            if (tid == 0) { if (dec == 2) atomicOr(&flags[WF_ERR], 8); flags[WF_CONV] = dec == 1 ? 1 : 0; flags[WF_ITS] = it; flags[WF_DONE] = 1; }
            return;
        }
        ++it;
        for (int w = tid; w < nw; w += kBlock) {
            const double* Di = A.Dinv + 16 * w;
            double dy[4];
            for (int k = 0; k < 4; ++k) { dy[k] = 0.0; for (int c = 0; c < 4; ++c) dy[k] += Di[4 * k + c] * Ebuf[4 * w + c]; }
            well_apply_increment(A, w, dy, dbhp_max_rel);
            if (!well_update_controls(A, w)) atomicOr(&flags[WF_ERR], 4);
        }
        __threadfence();
        __syncthreads();
        if (it >= max_it) { if (tid == 0) { flags[WF_CONV] = 0; flags[WF_ITS] = it; flags[WF_DONE] = 1; } return; }
    }
}
// how many workgroups of the fused pre-solve the device holds at once (occupancy query x compute units, halved as a margin for the side
// streams that run next to it): the fused path is taken only for nw up to this
static int fused_presolve_capacity()
{
    static int cap = -1;
    if (cap >= 0) return cap;
    int dev = 0, ncu = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_well_presolve_fused), kBlock, 0) != hipSuccess) return cap = 0;
    return cap = std::max(0, (ncu * per_cu) / 2);
}

// one step of the pre-solve loop (BlackoilModelBase_impl.hpp:1063-1097), ONE workgroup for all wells: getWellConvergence over all
// wells (B_avg * max |flux eq| < tolerance_wells, max |control eq| < tolerance_well_control); if not converged every well takes the
// Newton update dy = D^-1 E of its own 4x4 system (the well equations of different wells are independent while the reservoir is
// frozen), updateWellState, updateWellControls.  bsums = sums of 1/b per phase, ncells = their cell count.
__global__ __launch_bounds__(kBlock) void k_well_presolve_step(int nw, WellArgs A, const double* __restrict__ bsums, double ncells, double tol_wells,
                                                               double tol_ctrl, double max_resid, double dbhp_max_rel, int max_it, int32_t* __restrict__ flags)
{
    if (flags[WF_DONE]) return;
    __shared__ double sm[16];
    __shared__ int bad;
    const int tid = threadIdx.x;
    if (tid == 0) bad = 0;
    __syncthreads();
    double mx[4] = { 0, 0, 0, 0 };
    for (int w = tid; w < nw; w += kBlock) for (int k = 0; k < 4; ++k) { const double e = fabs(A.wellE[4 * w + k]); if (!(e == e)) bad = 1; mx[k] = fmax(mx[k], e); }
    for (int k = 0; k < 4; ++k) {
        const double m = wave_max(mx[k]);
        if ((tid & 63) == 0) sm[4 * k + (tid >> 6)] = m;
    }
    __syncthreads();
    bool conv = true, toolarge = false;
    for (int a = 0; a < 3; ++a) {
        const double wf = (bsums[a] / ncells) * fmax(fmax(sm[4 * a], sm[4 * a + 1]), fmax(sm[4 * a + 2], sm[4 * a + 3]));
        conv = conv && wf < tol_wells;
        toolarge = toolarge || wf > max_resid || !(wf == wf);
    }
    conv = conv && fmax(fmax(sm[12], sm[13]), fmax(sm[14], sm[15])) < tol_ctrl;
    if (bad || toolarge) {          // NaN / too large well residual: NumericalIssue (:1909-1923), reported by opmgpu_well_convergence
        if (tid == 0) { atomicOr(&flags[WF_ERR], 8); flags[WF_DONE] = 1; flags[WF_CONV] = 0; }
        return;
    }
    if (conv) { if (tid == 0) { flags[WF_DONE] = 1; flags[WF_CONV] = 1; } return; }
    for (int w = tid; w < nw; w += kBlock) {
        const double* Di = A.Dinv + 16 * w; const double* E = A.wellE + 4 * w;
        double dy[4];
        for (int k = 0; k < 4; ++k) { dy[k] = 0.0; for (int c = 0; c < 4; ++c) dy[k] += Di[4 * k + c] * E[c]; }
        well_apply_increment(A, w, dy, dbhp_max_rel);
        if (!well_update_controls(A, w)) atomicOr(&flags[WF_ERR], 4);
    }
    if (tid == 0) { const int it = flags[WF_ITS] + 1; flags[WF_ITS] = it; if (it >= max_it) flags[WF_DONE] = 1; }
}

// well state snapshot / restore around the pre-solve ("if (!converged) well_state = well_state0", :1124-1126); layout of snap:
// wstate[4 nw] | thp[nw] | perf_rates[3 np] | perf_press[np], current in isnap
__global__ __launch_bounds__(kBlock) void k_well_snapshot(int nw, int np, WellArgs A, double* __restrict__ snap, int32_t* __restrict__ isnap, int restore,
                                                          const int32_t* __restrict__ flags)
{
    if (restore && flags[WF_CONV]) return;          // a converged pre-solve keeps its solution
    const int i = blockIdx.x * kBlock + threadIdx.x;
    double* parts[4] = { A.wstate, A.thp, A.perf_rates, A.perf_press };
    const int len[4] = { 4 * nw, nw, 3 * np, np };
    int off = 0;
    for (int k = 0; k < 4; ++k) {
        if (i < len[k]) { if (restore) parts[k][i] = snap[off + i]; else snap[off + i] = parts[k][i]; }
        off += len[k];
    }
    if (i < nw) { if (restore) A.current[i] = isnap[i]; else isnap[i] = A.current[i]; }
    // the error word as it stood before the pre-solve (isnap[nw]): the serial fallback puts it back WHOLE -- a half-run fused attempt may have
    // left control-switch / assembly bits behind that are not the pre-solve's verdict
    if (!restore && i == 0) isnap[nw] = flags[WF_ERR];
}

// recoverVariable (NewtonIterationUtilities.cpp:134-184): dy = D^-1 (E - sum_j C_j dx_j) -- the well part of the Newton increment,
// kept resident so that stabilizeNonlinearUpdate relaxes the WHOLE increment like the reference does (NonlinearSolver_impl.hpp:260-301)
__global__ __launch_bounds__(kBlock) void k_well_recover(WellArgs A, const double* __restrict__ dx)
{
    __shared__ double sm[16];
    const int w = blockIdx.x, tid = threadIdx.x;
    double acc[4] = { 0, 0, 0, 0 };
    for (int j = A.connpos[w] + tid; j < A.connpos[w + 1]; j += kBlock) {
        const int row = A.perf_row[j];
        const double x0 = dx[row], x1 = dx[A.nbp + row], x2 = dx[2 * long(A.nbp) + row];
        const double* C = A.Q + 21 * long(j) + 9;
        for (int k = 0; k < 4; ++k) acc[k] += C[3 * k] * x0 + C[3 * k + 1] * x1 + C[3 * k + 2] * x2;
    }
    block_sum<4>(acc, sm);
    if (tid == 0) {
        const double* Di = A.Dinv + 16 * w; const double* E = A.wellE + 4 * w;
        for (int k = 0; k < 4; ++k) { double d = 0.0; for (int c = 0; c < 4; ++c) d += Di[4 * k + c] * (E[c] - acc[c]); A.wdy[4 * w + k] = d; }
    }
}
// stabilizeNonlinearUpdate on the well part: dy_old <- dy; dy <- omega dy (+ (1 - omega) previous dy_old for SOR)
__global__ void k_well_stabilize(int n, int sor, double omega, double* __restrict__ dy, double* __restrict__ dy_old)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = dy[i], o = dy_old[i];
    dy_old[i] = d;
    if (omega == 1.0) return;
    dy[i] = sor ? omega * d + (1.0 - omega) * o : omega * d;
}
// updateWellState (:611-700) from the resident increment
__global__ void k_well_update(int nw, WellArgs A, double relax, double dbhp_max_rel)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    double dy[4];
    for (int k = 0; k < 4; ++k) dy[k] = relax * A.wdy[4 * w + k];
    well_apply_increment(A, w, dy, dbhp_max_rel);
}

} // namespace

// ------------------------------------------------------------------------------------------ host side
struct BlackoilDevice::VfpDev {
    int ntab = 0;
    std::vector<int32_t> h_meta;        // [ntab][VM_COUNT]
    std::vector<int32_t> h_ids;
    DevArray<int32_t> meta;
    DevArray<double> datum, blob;
};

struct BlackoilDevice::WellsDev {
    int nw = 0, nctrl = 0;
    bool vfp_active = false;            // isVFPActive (BlackoilModelBase_impl.hpp:982-1008): some well has a THP control
    DevArray<int32_t> connpos, perf_row, perf_well, perf_of_row, type, allow_cf, ctrl_type, ctrl_ptr, ctrl_vfp, thp_ctrl, current, isnap, isaved;
    DevArray<double> WI, comp_frac, ctrl_target, ctrl_distr, ctrl_alq, depth_ref, z_perf, surf_dens_perf;
    DevArray<double> wstate, thp, cdp, perf_dens, perf_pvt, avgp, perf_rates, perf_press, P, Q, Fsave, wellE, Dinv, t, wdy, wdy_old;
    DevArray<double> bsums, bscratch, snap, presolve_sync, ctrl_row;
    DevArray<double> saved;         // snapshot for AdaptiveTimeStepping: wstate | thp | cdp | perf_rates | perf_press | perf_dens
    DevArray<int32_t> flags;
    std::vector<int32_t> h_connpos, h_cells, h_ctrl_ptr;
    double* h_pinned = nullptr;
    bool dy_valid = false;
};

void BlackoilDevice::vfp_free() { delete vfp; vfp = nullptr; }
void BlackoilDevice::wells_free() { if (wd) { if (wd->h_pinned) (void)hipHostFree(wd->h_pinned); delete wd; wd = nullptr; } }

static WellArgs args_of(BlackoilDevice::WellsDev& W, BlackoilDevice::VfpDev* V, int nbp, const double* perf, double gravity)
{
    WellArgs A;
    A.nbp = nbp;
    A.connpos = W.connpos.p; A.perf_row = W.perf_row.p; A.type = W.type.p; A.allow_cf = W.allow_cf.p; A.ctrl_type = W.ctrl_type.p;
    A.WI = W.WI.p; A.comp_frac = W.comp_frac.p; A.ctrl_target = W.ctrl_target.p; A.ctrl_distr = W.ctrl_distr.p; A.depth_ref = W.depth_ref.p;
    A.ctrl_ptr = W.ctrl_ptr.p; A.ctrl_vfp = W.ctrl_vfp.p; A.thp_ctrl = W.thp_ctrl.p; A.ctrl_alq = W.ctrl_alq.p; A.current = W.current.p; A.thp = W.thp.p;
    A.perf_dens = W.perf_dens.p; A.perf_pvt = W.perf_pvt.p; A.wdy = W.wdy.p; A.gravity = gravity; A.ctrl_row = W.ctrl_row.p;
    A.V.ntab = V ? V->ntab : 0; A.V.meta = V ? V->meta.p : nullptr; A.V.datum = V ? V->datum.p : nullptr; A.V.blob = V ? V->blob.p : nullptr;
    A.z_perf = W.z_perf.p; A.surf_dens_perf = W.surf_dens_perf.p; A.perf = perf;
    A.wstate = W.wstate.p; A.cdp = W.cdp.p; A.perf_rates = W.perf_rates.p; A.perf_press = W.perf_press.p; A.P = W.P.p; A.Q = W.Q.p;
    A.Fsave = W.Fsave.p; A.wellE = W.wellE.p; A.Dinv = W.Dinv.p;
    return A;
}

int BlackoilDevice::set_vfp_tables(int n, const opmgpu_vfp_table* tabs)
{
    if (n < 0 || (n > 0 && !tabs)) return OPMGPU_EINVAL;
    delete vfp; vfp = nullptr;
    if (n == 0) return OPMGPU_OK;
    VfpDev* V = new VfpDev();
    V->ntab = n;
    std::vector<double> blob, datum;
    const double one_point[1] = { 0.0 };
    for (int t = 0; t < n; ++t) {
        const opmgpu_vfp_table& T = tabs[t];
        const int nwfr = T.is_injector ? 1 : T.nwfr, ngfr = T.is_injector ? 1 : T.ngfr, nalq = T.is_injector ? 1 : T.nalq;
        if (T.nflo < 1 || T.nthp < 1 || nwfr < 1 || ngfr < 1 || nalq < 1 || T.nthp > kVfpMaxThp || !T.flo || !T.thp || !T.data ||
            (!T.is_injector && (!T.wfr || !T.gfr || !T.alq))) { delete V; return OPMGPU_EINVAL; }
        int32_t m[VM_COUNT] = { T.id, T.is_injector ? 1 : 0, T.flo_type, T.wfr_type, T.gfr_type, T.nflo, T.nthp, nwfr, ngfr, nalq, int32_t(blob.size()), 0 };
        blob.insert(blob.end(), T.flo, T.flo + T.nflo); blob.insert(blob.end(), T.thp, T.thp + T.nthp);
        const double* aw = T.is_injector ? one_point : T.wfr; const double* ag = T.is_injector ? one_point : T.gfr; const double* aa = T.is_injector ? one_point : T.alq;
        blob.insert(blob.end(), aw, aw + nwfr); blob.insert(blob.end(), ag, ag + ngfr); blob.insert(blob.end(), aa, aa + nalq);
        m[VM_DATA] = int32_t(blob.size());
        const size_t nd = size_t(T.nthp) * nwfr * ngfr * nalq * T.nflo;
        blob.insert(blob.end(), T.data, T.data + nd);
        V->h_meta.insert(V->h_meta.end(), m, m + VM_COUNT);
        V->h_ids.push_back(T.id);
        datum.push_back(T.datum_depth);
    }
    V->meta.upload(V->h_meta, stream); V->datum.upload(datum, stream); V->blob.upload(blob, stream);
    OPMGPU_HIP(hipStreamSynchronize(stream));
    vfp = V;
    return OPMGPU_OK;
}

int BlackoilDevice::set_device_wells(const opmgpu_wells* s)
{
    if (!s || s->nw < 0) return OPMGPU_EINVAL;
    if (s->nw == 0) { wells_free(); device_wells = false; ls.lowrank = LowRankOp(); ls.drop_hierarchies(); return set_wells(0, nullptr, nullptr); }
    if (!s->well_connpos || !s->well_cells || !s->WI || !s->type || !s->depth_ref || !s->comp_frac || !s->ctrl_type || !s->ctrl_target) return OPMGPU_EINVAL;
    const int nw = s->nw, np = s->well_connpos[nw];
    if (s->ctrl_ptr && s->ctrl_ptr[0] != 0) return OPMGPU_EINVAL;
    std::vector<int8_t> seen(nc, 0);
    for (int j = 0; j < np; ++j) {
        const int c = s->well_cells[j];
        if (c < 0 || c >= nc || seen[c]) return OPMGPU_EINVAL;         // a cell perforated twice is not supported
        seen[c] = 1;
    }
    wells_free();
    wd = new WellsDev();
    WellsDev& W = *wd;
    W.nw = nw;
    W.h_connpos.assign(s->well_connpos, s->well_connpos + nw + 1);
    W.h_cells.assign(s->well_cells, s->well_cells + np);
    device_wells = true;
    // pattern: plain stencil (no cliques); the perforated cells are still registered for k_perf_props
    h_well_connpos = W.h_connpos; h_well_cells = W.h_cells;
    rebuild_structure();
    auto upi = [&](DevArray<int32_t>& d, const int32_t* src, size_t n, int32_t dflt) {
        std::vector<int32_t> h(n, dflt); if (src) h.assign(src, src + n); d.upload(h, stream);
    };
    auto upd = [&](DevArray<double>& d, const double* src, size_t n) {
        std::vector<double> h(n, 0.0); if (src) h.assign(src, src + n); d.upload(h, stream);
    };
    // WellControls: several controls per well, or (ctrl_ptr == NULL) exactly one
    W.h_ctrl_ptr.resize(nw + 1);
    for (int w = 0; w <= nw; ++w) W.h_ctrl_ptr[w] = s->ctrl_ptr ? s->ctrl_ptr[w] : w;
    const int nct = W.nctrl = W.h_ctrl_ptr[nw];
    std::vector<int32_t> vfp_idx(nct, -1), thp_ctrl(nw, -1);
    W.vfp_active = false;
    for (int w = 0; w < nw; ++w) {
        if (W.h_ctrl_ptr[w + 1] <= W.h_ctrl_ptr[w]) { wells_free(); device_wells = false; return OPMGPU_EINVAL; }        // "There should be at least one control"
        for (int c = W.h_ctrl_ptr[w]; c < W.h_ctrl_ptr[w + 1]; ++c) {
            if (s->ctrl_type[c] < 0 || s->ctrl_type[c] > OPMGPU_CTRL_RESERVOIR_RATE) { wells_free(); device_wells = false; return OPMGPU_EINVAL; }
            if (s->ctrl_type[c] != OPMGPU_CTRL_THP) continue;
            const int id = s->ctrl_vfp ? s->ctrl_vfp[c] : -1;
            int found = -1;
            if (vfp) for (int t = 0; t < vfp->ntab; ++t) if (vfp->h_ids[t] == id && (vfp->h_meta[VM_COUNT * t + VM_INJ] != 0) == (s->type[w] == 0)) found = t;
            if (found < 0) { wells_free(); device_wells = false; return OPMGPU_EINVAL; }      // the reference throws std::invalid_argument for a missing table
            vfp_idx[c] = found;
            if (thp_ctrl[w] < 0) thp_ctrl[w] = c;
            W.vfp_active = true;
        }
    }
    upi(W.connpos, s->well_connpos, nw + 1, 0); upi(W.type, s->type, nw, 1); upi(W.allow_cf, s->allow_cf, nw, 1); upi(W.ctrl_type, s->ctrl_type, nct, 0);
    W.ctrl_ptr.upload(W.h_ctrl_ptr, stream); W.ctrl_vfp.upload(vfp_idx, stream); W.thp_ctrl.upload(thp_ctrl, stream);
    upd(W.WI, s->WI, np); upd(W.comp_frac, s->comp_frac, 3 * size_t(nw)); upd(W.ctrl_target, s->ctrl_target, nct);
    upd(W.ctrl_distr, s->ctrl_distr, 3 * size_t(nct)); upd(W.ctrl_alq, s->ctrl_alq, nct); upd(W.depth_ref, s->depth_ref, nw);
    W.current.alloc(nw); W.current.zero(stream); W.isnap.alloc(nw + 1); W.isaved.alloc(nw);
    W.thp.alloc(nw); W.thp.zero(stream);
    W.perf_dens.alloc(np); W.perf_dens.zero(stream); W.perf_pvt.alloc(5 * size_t(np)); W.avgp.alloc(np);
    W.ctrl_row.alloc(4 * size_t(nw)); W.ctrl_row.zero(stream);
    W.wdy.alloc(4 * size_t(nw)); W.wdy.zero(stream); W.wdy_old.alloc(4 * size_t(nw)); W.wdy_old.zero(stream);
    W.bsums.alloc(16); W.bscratch.alloc(13 * size_t(kMaxRedBlocks)); W.snap.alloc(5 * size_t(nw) + 4 * size_t(np));
    std::vector<double> zp(np), sd(3 * size_t(np));
    std::vector<int32_t> pw(np);
    for (int w = 0; w < nw; ++w) for (int j = s->well_connpos[w]; j < s->well_connpos[w + 1]; ++j) pw[j] = w;
    for (int j = 0; j < np; ++j) {
        const int c = s->well_cells[j];
        zp[j] = h_z[c];
        for (int a = 0; a < 3; ++a) sd[3 * size_t(j) + a] = h_surface_density[3 * size_t(h_pvtnum[c]) + a];
    }
    W.z_perf.upload(zp, stream); W.surf_dens_perf.upload(sd, stream); W.perf_well.upload(pw, stream);
    W.wstate.alloc(4 * size_t(nw)); W.wstate.zero(stream);
    W.cdp.alloc(np); W.cdp.zero(stream); W.perf_rates.alloc(3 * size_t(np)); W.perf_rates.zero(stream); W.perf_press.alloc(np); W.perf_press.zero(stream);
    W.P.alloc(21 * size_t(np)); W.Q.alloc(21 * size_t(np)); W.Fsave.alloc(21 * size_t(np)); W.wellE.alloc(4 * size_t(nw)); W.Dinv.alloc(16 * size_t(nw));
    W.P.zero(stream); W.Q.zero(stream); W.wellE.zero(stream); W.Dinv.zero(stream);
    W.t.alloc(7 * size_t(nw)); W.t.zero(stream);
    W.flags.alloc(WF_COUNT); W.flags.zero(stream);
    OPMGPU_HIP(hipHostMalloc(reinterpret_cast<void**>(&W.h_pinned), (4 * size_t(nw) + 8) * sizeof(double)));
    wells_rebind();
    OPMGPU_HIP(hipStreamSynchronize(stream));
    return OPMGPU_OK;
}

// (re)derive the row-indexed maps after a (re)plan and hand the operator to the solver
void BlackoilDevice::wells_rebind()
{
    if (!wd) return;
    WellsDev& W = *wd;
    const Plan& P = ls.plan;
    const int np = int(W.h_cells.size());
    std::vector<int32_t> pr(np), por(P.nbp, -1);
    for (int j = 0; j < np; ++j) { pr[j] = P.pos[W.h_cells[j]]; por[pr[j]] = j; }
    W.perf_row.upload(pr, stream); W.perf_of_row.upload(por, stream);
    d_rhs_extra.alloc(3 * size_t(P.nbp));
    LowRankOp& L = ls.lowrank;
    L.nw = W.nw; L.nperf = np; L.connpos = W.connpos.p; L.perf_row = W.perf_row.p; L.perf_well = W.perf_well.p; L.perf_of_row = W.perf_of_row.p;
    L.P = W.P.p; L.Q = W.Q.p; L.t = W.t.p;
    L.Fsave = W.Fsave.p; L.ctrl_row = W.ctrl_row.p;
    for (int a = 0; a < 3; ++a) L.scale[a] = prm.matbalscale[a];
    ls.drop_hierarchies();
}

int BlackoilDevice::well_state_set(const double* bhp, const double* qs, const double* perf_press, const double* perf_rates)
{
    if (!wd || !bhp || !qs) return OPMGPU_EINVAL;
    WellsDev& W = *wd;
    std::vector<double> h(4 * size_t(W.nw));
    for (int w = 0; w < W.nw; ++w) { for (int a = 0; a < 3; ++a) h[4 * w + a] = qs[3 * w + a]; h[4 * w + 3] = bhp[w]; }
    W.wstate.upload(h, stream);
    if (perf_rates) W.perf_rates.upload(perf_rates, 3 * W.h_cells.size(), stream);
    if (perf_press) W.perf_press.upload(perf_press, W.h_cells.size(), stream);
    OPMGPU_HIP(hipStreamSynchronize(stream));
    return OPMGPU_OK;
}

int BlackoilDevice::well_state_get(double* bhp, double* qs, double* perf_press, double* perf_rates)
{
    if (!wd) return OPMGPU_EINVAL;
    WellsDev& W = *wd;
    std::vector<double> h(4 * size_t(W.nw));
    W.wstate.download(h.data(), h.size(), stream);
    if (perf_press) W.perf_press.download(perf_press, W.h_cells.size(), stream);
    if (perf_rates) W.perf_rates.download(perf_rates, 3 * W.h_cells.size(), stream);
    OPMGPU_HIP(hipStreamSynchronize(stream));
    for (int w = 0; w < W.nw; ++w) { if (qs) for (int a = 0; a < 3; ++a) qs[3 * w + a] = h[4 * w + a]; if (bhp) bhp[w] = h[4 * w + 3]; }
    return OPMGPU_OK;
}

// computeWellConnectionPressures (StandardWells_impl.hpp:336-358) from the resident reservoir + well state; gate: run only if *gate != 0
void BlackoilDevice::wells_connection_pressures(const int32_t* gate)
{
    WellsDev& W = *wd;
    WellArgs A = args_of(W, vfp, ls.plan.nbp, d_perf.p, gravity);
    const int g = (W.nw + 63) / 64;
    hipLaunchKernelGGL(k_well_avg_press, dim3(g), dim3(64), 0, stream, W.nw, A, gate, W.avgp.p);
    perf_pvt_device(W.avgp.p, W.perf_pvt.p, gate);
    hipLaunchKernelGGL(k_well_cdp, dim3(W.nw), dim3(64), 0, stream, W.nw, A, gravity, gate);
}

// called by assemble() after the reservoir kernels; the well part of BlackoilModelBase::assemble in the reference's order (:757-840)
void BlackoilDevice::wells_prologue()
{
    WellsDev& W = *wd;
    perf_props_device();
    WellArgs A = args_of(W, vfp, ls.plan.nbp, d_perf.p, gravity);
    if (W.vfp_active) wells_connection_pressures(nullptr);        // VFP: densities for the hydrostatic correction, every assembly (:771-776)
    hipLaunchKernelGGL(k_well_controls, dim3((W.nw + 63) / 64), dim3(64), 0, stream, W.nw, A, W.flags.p);        // updateWellControls (:785)
}
// Newton iterations after the first of a time step: the prologue needs the state only, the reservoir assembly (0.36 ms) does not touch what
// it writes (d_perf, the wells' control state) -- so it runs beside it on its own stream instead of 35 us behind it.  Not on the first
// iteration (the pre-solve that follows needs the assembly's 1/b sums anyway) and not with THP controls (connection pressures in between).
bool BlackoilDevice::wells_prologue_async(bool initial)
{
    static const bool on = !(std::getenv("OPMGPU_WELL_PROLOGUE_ASYNC") && std::atoi(std::getenv("OPMGPU_WELL_PROLOGUE_ASYNC")) == 0);
    well_prologue_done = false;
    if (!on || !wd || initial || wd->vfp_active) return false;
    if (!well_stream) {
        OPMGPU_HIP(hipStreamCreateWithFlags(&well_stream, hipStreamNonBlocking));
        for (auto& e : ev_well) OPMGPU_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    OPMGPU_HIP(hipEventRecord(ev_well[0], stream));               // the state the previous update left
    OPMGPU_HIP(hipStreamWaitEvent(well_stream, ev_well[0], 0));
    {
        StreamSwapGuard g(stream, well_stream, ls.kt.on);          // (event brackets belong to the main stream); restored also if the prologue throws
        wells_prologue();
    }
    OPMGPU_HIP(hipEventRecord(ev_well[1], well_stream));
    well_prologue_done = true;
    return true;
}

void BlackoilDevice::wells_assemble(bool initial)
{
    if (!wd) {
        // multi-GPU run with wells on OTHER ranks: the pre-solve's B_avg is a global mean -- take part in its all-reduce
        if (initial && prm.solve_welleq_initially && ls.comm && ls.run_has_wells) binv_sums_device(d_red.p, d_red.p + kRedPart);
        return;
    }
    WellsDev& W = *wd;
    const Plan& P = ls.plan;
    if (!well_prologue_done) wells_prologue();
    well_prologue_done = false;
    WellArgs A = args_of(W, vfp, P.nbp, d_perf.p, gravity);
    const double* sc = prm.matbalscale;
    if (initial) {
        wells_connection_pressures(nullptr);      // once per time step from the explicit state (:797-805)
        W.wdy_old.zero(stream);
        if (prm.solve_welleq_initially) {
            // solveWellEq (:1018-1133): <= 15 Newton iterations on the well equations with the reservoir frozen.  Every launch is a
            // no-op once the device-side `done` flag is up, so the loop needs no host synchronisation.
            const int np = int(W.h_cells.size());
            const int gs = grid_for(std::max(4 * W.nw, 3 * np));
            hipLaunchKernelGGL(k_well_snapshot, dim3(gs), dim3(kBlock), 0, stream, W.nw, np, A, W.snap.p, W.isnap.p, 0, (const int32_t*)W.flags.p);
            OPMGPU_HIP(hipMemsetAsync(W.flags.p + WF_DONE, 0, 3 * sizeof(int32_t), stream));
            binv_sums_device(W.bsums.p, W.bscratch.p);
            const double ncg = ls.comm ? double(ls.comm->n_owned_global) : double(nc);
            const int max_it = 15;
            // OPMGPU_WELL_PRESOLVE_FUSED: 1 (default) one launch with a counter barrier; 0 two launches per iteration; 2 (tests) the serial
            // fallback that takes over when the fused kernel's barrier gives up.  Read per call: once per time step.
            const char* fenv = std::getenv("OPMGPU_WELL_PRESOLVE_FUSED");
            const int fmode = fenv ? std::atoi(fenv) : 1;
            if (fmode != 0 && W.nw <= std::min(kFusedWells, fused_presolve_capacity())) {
                // one launch: one workgroup per well, counter barrier per iteration -- all nw workgroups fit the device together
                // (fused_presolve_capacity: the kernel's occupancy x compute units, with a margin)
                W.presolve_sync.alloc(4 + 8 * size_t(W.nw)); W.presolve_sync.zero(stream);
                if (fmode == 2) hipLaunchKernelGGL(k_well_flag_or, dim3(1), dim3(1), 0, stream, W.flags.p + WF_ERR, 16 | 4);      // (tests) the give-up bit AND a stray control bit of a half-run attempt: the fallback must wipe both
                else
                hipLaunchKernelGGL(k_well_presolve_fused, dim3(W.nw), dim3(kBlock), 0, stream, W.nw, A, ls.dp.slice_ptr.p, ls.dp.nlower.p, (const double*)W.bsums.p, ncg,
                                   prm.tolerance_wells, prm.tolerance_well_control, prm.max_residual_allowed, prm.dbhp_max_rel, max_it, W.flags.p,
                                   reinterpret_cast<int32_t*>(W.presolve_sync.p), W.presolve_sync.p + 4);
                // does nothing unless the barrier gave up (error bit 16): then the loop once more from the snapshot, in one workgroup
                hipLaunchKernelGGL(k_well_presolve_serial, dim3(1), dim3(kBlock), 0, stream, W.nw, np, A, ls.dp.slice_ptr.p, ls.dp.nlower.p, (const double*)W.bsums.p, ncg,
                                   prm.tolerance_wells, prm.tolerance_well_control, prm.max_residual_allowed, prm.dbhp_max_rel, max_it, W.flags.p,
                                   (const double*)W.snap.p, (const int32_t*)W.isnap.p, W.presolve_sync.p + 4);
            } else
            for (int it = 0; it <= max_it; ++it) {
                hipLaunchKernelGGL((k_well_assemble<double, true>), dim3(W.nw), dim3(kBlock), 0, stream, A, ls.dp.slice_ptr.p, ls.dp.nlower.p, sc[0], sc[1], sc[2],
                                   (double*)nullptr, (double*)nullptr, (double*)nullptr, W.flags.p);
                hipLaunchKernelGGL(k_well_presolve_step, dim3(1), dim3(kBlock), 0, stream, W.nw, A, (const double*)W.bsums.p, ncg, prm.tolerance_wells,
                                   prm.tolerance_well_control, prm.max_residual_allowed, prm.dbhp_max_rel, max_it, W.flags.p);
            }
            // converged: connection pressures from the new well state (:1122); otherwise the well state is restored (:1124-1126)
            wells_connection_pressures(W.flags.p + WF_CONV);
            hipLaunchKernelGGL(k_well_snapshot, dim3(gs), dim3(kBlock), 0, stream, W.nw, np, A, W.snap.p, W.isnap.p, 1, (const int32_t*)W.flags.p);
        }
    }
    d_rhs_extra.zero(stream);
    if (ls.matrix_is_float)
        hipLaunchKernelGGL((k_well_assemble<float, false>), dim3(W.nw), dim3(kBlock), 0, stream, A, ls.dp.slice_ptr.p, ls.dp.nlower.p, sc[0], sc[1], sc[2], d_R.p,
                           ls.matrix_f(), d_rhs_extra.p, W.flags.p);
    else
        hipLaunchKernelGGL((k_well_assemble<double, false>), dim3(W.nw), dim3(kBlock), 0, stream, A, ls.dp.slice_ptr.p, ls.dp.nlower.p, sc[0], sc[1], sc[2], d_R.p,
                           ls.matrix_d(), d_rhs_extra.p, W.flags.p);
    if (ls.matrix_is_float) ls.cpr_reweigh_rows<float>(W.perf_row.p, int(W.h_cells.size())); else ls.cpr_reweigh_rows<double>(W.perf_row.p, int(W.h_cells.size()));
    has_rhs_extra = true;
    W.dy_valid = false;
}

// well part of getConvergence (BlackoilModelBase_impl.hpp:1769-1779): max |flux equation| per phase, max |control equation|
bool BlackoilDevice::has_device_wells() const { return wd != nullptr || ls.run_has_wells; }

// max |flux equation| per phase, max |control equation|, NaN / failure mark, singular mark of this rank's wells (zeros without wells); the
// error word is consumed (reset) here, like well_convergence() does on the host path
__global__ __launch_bounds__(kBlock) void k_well_conv_pack(int nw, const double* __restrict__ wellE, int32_t* __restrict__ flags, double* __restrict__ out)
{
    __shared__ double sm[4][6];
    double m[6] = { 0, 0, 0, 0, 0, 0 };
    for (int w = threadIdx.x; w < nw; w += kBlock)
        for (int k = 0; k < 4; ++k) { const double e = fabs(wellE[4 * w + k]); if (!(e == e)) m[4] = 1.0; else m[k] = fmax(m[k], e); }
    for (int k = 0; k < 6; ++k) {
        double v = m[k];
        for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < 6; ++k) out[k] = fmax(fmax(sm[0][k], sm[1][k]), fmax(sm[2][k], sm[3][k]));
        const int32_t fl = flags ? flags[0] : 0;
        if (fl & (4 | 8 | 16)) out[4] = 1.0;
        if (fl & 2) out[5] = 1.0;
        if (fl) flags[0] = 0;
    }
}
void BlackoilDevice::well_conv_pack(double* d_out6)
{
    if (wd) hipLaunchKernelGGL(k_well_conv_pack, dim3(1), dim3(kBlock), 0, stream, wd->nw, (const double*)wd->wellE.p, wd->flags.p + WF_ERR, d_out6);
    else OPMGPU_HIP(hipMemsetAsync(d_out6, 0, 6 * sizeof(double), stream));
}

bool BlackoilDevice::well_words_sources(const void*& e, int& ne, const void*& f) const
{
    if (!wd) return false;
    e = wd->wellE.p; ne = 8 * wd->nw; f = wd->flags.p + WF_ERR;
    return true;
}

int BlackoilDevice::well_convergence(double* flux3, double* ctrl)
{
    if (!wd && !ls.comm) return OPMGPU_EINVAL;
    double f[3] = { 0, 0, 0 }, c = 0.0;
    bool bad = false, singular = false;
    if (ls.comm && well_red_valid) {            // all-reduced with the cells' maxima by convergence() after this assembly
        well_red_valid = false;
        for (int a = 0; a < 3; ++a) f[a] = h_red[13 + a];
        c = h_red[16]; bad = h_red[17] != 0.0; singular = h_red[18] != 0.0;
        if (flux3) for (int a = 0; a < 3; ++a) flux3[a] = f[a];
        if (ctrl) *ctrl = c;
        if (singular) return OPMGPU_ESINGULAR;
        return bad ? OPMGPU_ENUMERICAL : OPMGPU_OK;
    }
    if (wd) {
        WellsDev& W = *wd;
        int32_t fl = 0;
        if (well_words_valid && int(well_words.size()) == 8 * W.nw + 1) {            // read back by convergence() after this assembly
            std::memcpy(W.h_pinned, well_words.data(), 4 * size_t(W.nw) * sizeof(double));
            std::memcpy(&fl, well_words.data() + 8 * W.nw, sizeof(int32_t));
            well_words_valid = false;
        } else if (8 * W.nw + 1 <= LinSolver::kPubWords) {
            const uint32_t* h = ls.fetch_words(W.wellE.p, 8 * W.nw, W.flags.p + WF_ERR, 1);
            std::memcpy(W.h_pinned, h, 4 * size_t(W.nw) * sizeof(double));
            std::memcpy(&fl, h + 8 * W.nw, sizeof(int32_t));
        } else {
            OPMGPU_HIP(hipMemcpyAsync(W.h_pinned, W.wellE.p, 4 * size_t(W.nw) * sizeof(double), hipMemcpyDeviceToHost, stream));
            OPMGPU_HIP(hipMemcpyAsync(&fl, W.flags.p + WF_ERR, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
            OPMGPU_HIP(hipStreamSynchronize(stream));
        }
        for (int w = 0; w < W.nw; ++w) {
            for (int a = 0; a < 3; ++a) { const double e = std::fabs(W.h_pinned[4 * w + a]); if (!(e == e)) bad = true; f[a] = std::max(f[a], e); }
            const double e = std::fabs(W.h_pinned[4 * w + 3]); if (!(e == e)) bad = true; c = std::max(c, e);
        }
        if (fl) OPMGPU_HIP(hipMemsetAsync(W.flags.p + WF_ERR, 0, sizeof(int32_t), stream));
        if (fl & 2) singular = true;
        if (fl & (4 | 8 | 16)) bad = true;  // no consistent control / NaN or too large residual in the pre-solve / its barrier gave up: NumericalIssue
    }
    if (ls.comm) {      // collective: every rank calls it, also the ones without wells
        double loc[6] = { f[0], f[1], f[2], c, bad ? 1.0 : 0.0, singular ? 1.0 : 0.0 };
        OPMGPU_HIP(hipMemcpyAsync(d_red.p, loc, sizeof(loc), hipMemcpyHostToDevice, stream));
        ls.comm->allreduce_max(d_red.p, 6, stream);
        OPMGPU_HIP(hipMemcpyAsync(h_red, d_red.p, sizeof(loc), hipMemcpyDeviceToHost, stream));
        OPMGPU_HIP(hipStreamSynchronize(stream));
        for (int a = 0; a < 3; ++a) f[a] = h_red[a];
        c = h_red[3]; bad = h_red[4] != 0.0; singular = h_red[5] != 0.0;
    }
    if (flux3) for (int a = 0; a < 3; ++a) flux3[a] = f[a];
    if (ctrl) *ctrl = c;
    if (singular) return OPMGPU_ESINGULAR;
    return bad ? OPMGPU_ENUMERICAL : OPMGPU_OK;
}

void BlackoilDevice::wells_recover()
{
    if (!wd) return;
    WellsDev& W = *wd;
    WellArgs A = args_of(W, vfp, ls.plan.nbp, d_perf.p, gravity);
    hipLaunchKernelGGL(k_well_recover, dim3(W.nw), dim3(kBlock), 0, stream, A, (const double*)d_dx.p);
    W.dy_valid = true;
}

void BlackoilDevice::wells_stabilize(int sor, double omega)
{
    if (!wd) return;
    WellsDev& W = *wd;
    if (!W.dy_valid) wells_recover();
    const int n = 4 * W.nw;
    hipLaunchKernelGGL(k_well_stabilize, dim3((n + 63) / 64), dim3(64), 0, stream, n, sor, omega, W.wdy.p, W.wdy_old.p);
}

void BlackoilDevice::wells_update(double relax, bool dx_from_host)
{
    if (!wd) return;
    WellsDev& W = *wd;
    if (!W.dy_valid || dx_from_host) wells_recover();
    WellArgs A = args_of(W, vfp, ls.plan.nbp, d_perf.p, gravity);
    hipLaunchKernelGGL(k_well_update, dim3((W.nw + 63) / 64), dim3(64), 0, stream, W.nw, A, relax, prm.dbhp_max_rel);
    W.dy_valid = false;
}

// device-side last_state of AdaptiveTimeStepping: wstate | thp | cdp | perf_rates | perf_press | perf_dens, current
void BlackoilDevice::wells_save()
{
    if (!wd) return;
    WellsDev& W = *wd;
    const size_t nw = size_t(W.nw), np = W.h_cells.size();
    W.saved.alloc(5 * nw + 6 * np);
    double* parts[6] = { W.wstate.p, W.thp.p, W.cdp.p, W.perf_rates.p, W.perf_press.p, W.perf_dens.p };
    const size_t len[6] = { 4 * nw, nw, np, 3 * np, np, np };
    size_t off = 0;
    for (int k = 0; k < 6; ++k) { OPMGPU_HIP(hipMemcpyAsync(W.saved.p + off, parts[k], len[k] * sizeof(double), hipMemcpyDeviceToDevice, stream)); off += len[k]; }
    OPMGPU_HIP(hipMemcpyAsync(W.isaved.p, W.current.p, nw * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
}
void BlackoilDevice::wells_restore()
{
    if (!wd || !wd->saved.p) return;
    WellsDev& W = *wd;
    const size_t nw = size_t(W.nw), np = W.h_cells.size();
    double* parts[6] = { W.wstate.p, W.thp.p, W.cdp.p, W.perf_rates.p, W.perf_press.p, W.perf_dens.p };
    const size_t len[6] = { 4 * nw, nw, np, 3 * np, np, np };
    size_t off = 0;
    for (int k = 0; k < 6; ++k) { OPMGPU_HIP(hipMemcpyAsync(parts[k], W.saved.p + off, len[k] * sizeof(double), hipMemcpyDeviceToDevice, stream)); off += len[k]; }
    OPMGPU_HIP(hipMemcpyAsync(W.current.p, W.isaved.p, nw * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
}

int BlackoilDevice::well_controls_set(const int32_t* current, const double* thp)
{
    if (!wd) return OPMGPU_EINVAL;
    WellsDev& W = *wd;
    if (current) {
        for (int w = 0; w < W.nw; ++w) if (current[w] < 0 || current[w] >= W.h_ctrl_ptr[w + 1] - W.h_ctrl_ptr[w]) return OPMGPU_EINVAL;
        W.current.upload(current, size_t(W.nw), stream);
    }
    if (thp) W.thp.upload(thp, size_t(W.nw), stream);
    OPMGPU_HIP(hipStreamSynchronize(stream));
    return OPMGPU_OK;
}
// well_controls_iset_target / well_controls_iset_distr of every control at once (what SimulatorBase::computeRESV does to the RESERVOIR_RATE
// controls once per report step, SimulatorBase_impl.hpp:551-553): target[nctrl] and / or distr[nctrl][3] in the order of opmgpu_set_device_wells
int BlackoilDevice::well_controls_set_targets(const double* target, const double* distr)
{
    if (!wd) return OPMGPU_EINVAL;
    WellsDev& W = *wd;
    if (target) W.ctrl_target.upload(target, size_t(W.nctrl), stream);
    if (distr) W.ctrl_distr.upload(distr, 3 * size_t(W.nctrl), stream);
    OPMGPU_HIP(hipStreamSynchronize(stream));
    return OPMGPU_OK;
}
int BlackoilDevice::well_controls_get(int32_t* current, double* thp, int32_t* pre_its, int32_t* pre_conv)
{
    if (!wd) return OPMGPU_EINVAL;
    WellsDev& W = *wd;
    int32_t fl[WF_COUNT];
    if (current) W.current.download(current, size_t(W.nw), stream);
    if (thp) W.thp.download(thp, size_t(W.nw), stream);
    W.flags.download(fl, WF_COUNT, stream);
    OPMGPU_HIP(hipStreamSynchronize(stream));
    if (pre_its) *pre_its = fl[WF_ITS];
    if (pre_conv) *pre_conv = fl[WF_CONV];
    return OPMGPU_OK;
}

} // namespace opmgpu
