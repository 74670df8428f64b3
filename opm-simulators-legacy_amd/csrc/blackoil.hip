// blackoil.hip -- gfx950 kernels for the black-oil assembly / convergence / update (see blackoil.hpp).
//
// Two kernels per assembly, both one-thread-per-cell over the solver's internal (level-major)
// numbering so that every per-cell plane access of a wavefront is one contiguous segment:
//   k_cell_values   : state -> PVT / relperm / pc VALUES; writes the ten value planes a cell's
//                     neighbours need of it (phase pressures, densities, b * mobility, rs, rv).
//   k_assemble_rows : per row: eval_cell again for the row's OWN derivatives, accumulation term,
//                     then the row's SELL slots: every connection's TPFA flux from the neighbour's
//                     values, its derivative with respect to the row's own variables into the
//                     row's diagonal block and, negated, into the TRANSPOSED entry (neighbour, row)
//                     (each face is evaluated from both sides: 2x the flops, zero atomics, each
//                     Jacobian block written exactly once).
// Algorithmic HBM bytes per cell are given in DESIGN.md; the Jacobian write dominates.
#include "blackoil.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace opmgpu {

// ------------------------------------------------------------------------------------------
// device-side fluid property evaluation.  Written independently of oracle/ (explicit chain rule,
// branch-free linear scans over the small tables instead of bisection).
// ------------------------------------------------------------------------------------------
struct V4 { double v, p, w, x; };      // value, d/dP, d/dSw, d/dXvar

__device__ __forceinline__ V4 mk(double v, double p, double w, double x) { V4 r; r.v = v; r.p = p; r.w = w; r.x = x; return r; }
__device__ __forceinline__ V4 vmul(const V4& a, const V4& b) { return mk(a.v * b.v, a.p * b.v + a.v * b.p, a.w * b.v + a.v * b.w, a.x * b.v + a.v * b.x); }
__device__ __forceinline__ V4 vadd(const V4& a, const V4& b) { return mk(a.v + b.v, a.p + b.p, a.w + b.w, a.x + b.x); }
__device__ __forceinline__ V4 vscale(double s, const V4& a) { return mk(s * a.v, s * a.p, s * a.w, s * a.x); }
__device__ __forceinline__ V4 vdiv(const V4& a, const V4& b)
{
    const double q = a.v / b.v, ib = 1.0 / b.v;
    return mk(q, (a.p - q * b.p) * ib, (a.w - q * b.w) * ib, (a.x - q * b.x) * ib);
}
// f(g): f value, df = f'(g.v)
__device__ __forceinline__ V4 vchain(double f, double df, const V4& g) { return mk(f, df * g.p, df * g.w, df * g.x); }
__device__ __forceinline__ V4 vchain2(double f, double dfa, const V4& a, double dfb, const V4& b)
{
    return mk(f, dfa * a.p + dfb * b.p, dfa * a.w + dfb * b.w, dfa * a.x + dfb * b.x);
}

// saturation tables: constant extrapolation; LEFT = segment x[i] < xv <= x[i+1] (SWOF by Sw),
// RIGHT = x[i] <= xv < x[i+1] (SGOF by Sg: opm-material tabulates those against So).
template <bool RIGHT>
__device__ __forceinline__ void sat_eval(const double* __restrict__ x, const double* __restrict__ y, int n, double xv, double& f, double& df)
{
    if (xv <= x[0]) { f = y[0]; df = 0.0; return; }
    if (xv >= x[n - 1]) { f = y[n - 1]; df = 0.0; return; }
    int i = 0;
    for (int k = 1; k < n - 1; ++k) i += (RIGHT ? (x[k] <= xv) : (x[k] < xv)) ? 1 : 0;
    df = (y[i + 1] - y[i]) / (x[i + 1] - x[i]);
    f = y[i] + df * (xv - x[i]);
}
// PVT tables: linear extrapolation, opm-material Tabulated1DFunction segment rule
__device__ __forceinline__ int pvt_seg(const double* __restrict__ x, int n, double xv)
{
    int i = 0;
    if (n > 2) i = (x[1] < xv) ? 1 : 0;
    for (int k = 2; k <= n - 2; ++k) i += (x[k] <= xv) ? 1 : 0;
    return i;
}
__device__ __forceinline__ void pvt1(const double* __restrict__ x, const double* __restrict__ y, int n, double xv, double& f, double& df)
{
    const int i = pvt_seg(x, n, xv);
    df = (y[i + 1] - y[i]) / (x[i + 1] - x[i]);
    f = y[i] + df * (xv - x[i]);
}
// UniformXTabulated2DFunction::eval: f(xnode, ycol)
__device__ __forceinline__ void pvt2(const double* __restrict__ xs, int nn, const int32_t* __restrict__ cp, const double* __restrict__ cy,
                                     const double* __restrict__ cv, double xv, double yv, double& f, double& dfx, double& dfy)
{
    const int i = pvt_seg(xs, nn, xv);
    const double h = xs[i + 1] - xs[i];
    const double alpha = (xv - xs[i]) / h;
    double s1, d1, s2, d2;
    pvt1(cy + cp[i], cv + cp[i], cp[i + 1] - cp[i], yv, s1, d1);
    pvt1(cy + cp[i + 1], cv + cp[i + 1], cp[i + 2] - cp[i + 1], yv, s2, d2);
    f = s1 * (1.0 - alpha) + s2 * alpha;
    dfx = (s2 - s1) / h;
    dfy = d1 * (1.0 - alpha) + d2 * alpha;
}

__device__ __forceinline__ void rs_sat_d(const opmgpu_tables& T, int reg, double p, double& f, double& df)
{
    if (!T.has_disgas) { f = 0.0; df = 0.0; return; }
    const int a = T.oil_node_ptr[reg];
    pvt1(T.oil_psat + a, T.oil_rs + a, T.oil_node_ptr[reg + 1] - a, p, f, df);
}
__device__ __forceinline__ void rv_sat_d(const opmgpu_tables& T, int reg, double p, double& f, double& df)
{
    if (!T.has_vapoil) { f = 0.0; df = 0.0; return; }
    const int a = T.gas_node_ptr[reg];
    pvt1(T.gas_pg + a, T.gas_rvsat + a, T.gas_node_ptr[reg + 1] - a, p, f, df);
}

// ENDSCALE: per curve S_unscaled = u0 + (S - s0) * k (two-point, k = (u2 - u0) / (s2 - s0)); with SCALECRS the relative-permeability
// curves go through a middle point: S >= s1 -> u1 + (S - s1) * k1 (scaledToUnscaledSatThreePoint_).  Vertical scaling (KRW / KRO / KRG /
// PCW / PCG): the table value times v = cell maximum / table maximum.  All precomputed on the host (BlackoilDevice::rebuild_structure).
// Curve order: krw, krow, pcow, krg, krog (in oil saturation), pcgo.
enum { EC_KRW = 0, EC_KROW, EC_PCOW, EC_KRG, EC_KROG, EC_PCGO, EC_COUNT };
constexpr int kEpsPlanes = 5 * EC_COUNT;       // [s0 | k | s1 | k1 | v] x 6 curves, stride nbp
constexpr int kEpsRegion = 2 * EC_COUNT;       // [u0 | u1] x 6 curves per saturation region
struct EpsD {
    bool on;
    double u0[EC_COUNT], s0[EC_COUNT], k[EC_COUNT], u1[EC_COUNT], s1[EC_COUNT], k1[EC_COUNT], v[EC_COUNT];
};
__device__ __forceinline__ void eps_load(const double* __restrict__ eps, const double* __restrict__ ureg, long nbp, int row, int sreg, EpsD& e)
{
    e.on = eps != nullptr;
    if (!e.on) return;
#pragma unroll
    for (int c = 0; c < EC_COUNT; ++c) {
        e.u0[c] = ureg[kEpsRegion * sreg + c]; e.u1[c] = ureg[kEpsRegion * sreg + EC_COUNT + c];
        e.s0[c] = eps[long(c) * nbp + row];
        e.k[c] = eps[long(EC_COUNT + c) * nbp + row];
        e.s1[c] = eps[long(2 * EC_COUNT + c) * nbp + row];
        e.k1[c] = eps[long(3 * EC_COUNT + c) * nbp + row];
        e.v[c] = eps[long(4 * EC_COUNT + c) * nbp + row];
    }
}
// scaled -> unscaled saturation of curve c and the slope of the map there
__device__ __forceinline__ double eps_map(const EpsD& e, int c, double sv, double& slope)
{
    if (sv >= e.s1[c]) { slope = e.k1[c]; return e.u1[c] + (sv - e.s1[c]) * e.k1[c]; }
    slope = e.k[c];
    return e.u0[c] + (sv - e.s0[c]) * e.k[c];
}
__device__ __forceinline__ double eps_unmap(const EpsD& e, int c, double su)       // unscaled -> scaled (inverse map)
{
    return su >= e.u1[c] && e.s1[c] < 1e30 ? e.s1[c] + (su - e.u1[c]) / e.k1[c] : e.s0[c] + (su - e.u0[c]) / e.k[c];
}
template <bool RIGHT>
__device__ __forceinline__ void sat_curve(const double* __restrict__ x, const double* __restrict__ y, int n, double sv, const EpsD& e, int c, double& f, double& df)
{
    if (!e.on) { sat_eval<RIGHT>(x, y, n, sv, f, df); return; }
    double slope;
    sat_eval<RIGHT>(x, y, n, eps_map(e, c, sv, slope), f, df);
    f *= e.v[c]; df *= slope * e.v[c];
}
// Relative-permeability hysteresis of a cell (EclHysteresisTwoPhaseLaw, Carlson for the non-wetting phases, KR only): the history
// planes hold the smallest wetting saturation each two-phase system has seen (2.0 = none) and the shift of the imbibition curve.
struct HystD { bool on; int ireg; double mdc_ow, mdc_go, d_ow, d_go; };
struct HystArgs { const int32_t* imbnum; const double* hist; const double* ieps; const double* iureg; };      // kernel argument: imbnum == nullptr = no hysteresis
__device__ __forceinline__ void hyst_load(const int32_t* __restrict__ imbnum, const double* __restrict__ hist, long nbp, int row, HystD& h)
{
    h.on = imbnum != nullptr;
    if (!h.on) return;
    h.ireg = imbnum[row];
    h.mdc_ow = hist[row]; h.mdc_go = hist[nbp + row]; h.d_ow = hist[2 * nbp + row]; h.d_go = hist[3 * nbp + row];
}

// Tables in LDS: every table function is two dependent memory round trips (find the segment, read its end points) and a cell
// evaluates ~14 of them; from L1/L2 that chain is ~10 us of pure latency per wave, from LDS a tenth of it.  The whole blob is
// copied by the workgroup (words > 0; the host passes 0 when it does not fit) and the struct's pointers are rebased onto it.
__device__ __forceinline__ void stage_tables(opmgpu_tables& T, const double* __restrict__ blob, int words, double* lds)
{
    if (words <= 0) return;
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = blob[i];
    __syncthreads();
    auto rd = [&](const double*& p) { if (p) p = lds + (p - blob); };
    auto ri = [&](const int32_t*& p) { if (p) p = reinterpret_cast<const int32_t*>(lds + (reinterpret_cast<const double*>(p) - blob)); };
    rd(T.surface_density); rd(T.pvtw);
    ri(T.oil_node_ptr); rd(T.oil_rs); rd(T.oil_psat); rd(T.oil_invb_sat); rd(T.oil_invbmu_sat);
    ri(T.oil_col_ptr); rd(T.oil_col_p); rd(T.oil_col_invb); rd(T.oil_col_invbmu);
    ri(T.gas_node_ptr); rd(T.gas_pg); rd(T.gas_rvsat); rd(T.gas_invb_sat); rd(T.gas_invbmu_sat);
    ri(T.gas_col_ptr); rd(T.gas_col_rv); rd(T.gas_col_invb); rd(T.gas_col_invbmu);
    ri(T.swof_ptr); rd(T.swof_sw); rd(T.swof_krw); rd(T.swof_krow); rd(T.swof_pcow);
    ri(T.sgof_ptr); rd(T.sgof_sg); rd(T.sgof_krg); rd(T.sgof_krog); rd(T.sgof_pcgo);
    rd(T.rocktab_p); rd(T.rocktab_pvmult); rd(T.rocktab_transmult);
}

// VAPPARS (applyVap, BlackoilPropsAdFromDeck.cpp:1052-1078): factor (so/soMax)^vap and its so-derivative
__device__ __forceinline__ void vap_factor(double vap, double so, double so_max, double& f, double& df)
{
    f = 1.0; df = 0.0;
    if (vap > 0.0 && so_max > 0.01 && so < so_max) {
        const double so_i = fmax(so, 1.4901161193847656e-08);
        f = pow(so_i / so_max, vap);
        df = vap * pow(so_i / so_max, vap - 1.0) / so_max;
    }
}
// ROCKTAB (RockCompressibility.cpp:86-125 -> Opm::linearInterpolation): left segment at a breakpoint, linear extrapolation
__device__ __forceinline__ void rocktab_eval(const double* __restrict__ x, const double* __restrict__ y, int n, double xv, double& f, double& df)
{
    int i = 0;
    for (int k = 1; k <= n - 2; ++k) i += (x[k] < xv) ? 1 : 0;
    df = (y[i + 1] - y[i]) / (x[i + 1] - x[i]);
    f = y[i] + df * (xv - x[i]);
}

// Device-internal companions of the tables: the slope (y[i+1] - y[i]) / (x[i+1] - x[i]) of every segment of every 1-D table, formed once
// on the host by the same IEEE division the kernels used to repeat per cell and per table (an f64 division is ~12 quarter-rate-class
// instructions on gfx950 and eval_cell made ~50 of them; it runs twice per cell and assembly now, and it is VALU-bound: 3 800 static
// instructions in its value-only form).  Arrays are as long as their table (the last entry of a table is unused).
// (struct TabX / DevTables: blackoil.hpp)

// The hot kernels take the tables with every pointer field holding a WORD OFFSET into the blob and resolve them here against the blob's
// copy in LDS (LDS = true: the workgroup stages it first) or against the blob itself.  With the base a compile-time choice every table
// access of the LDS instantiation is a ds_read at a 32-bit offset (the round-2 form rebased generic pointers at run time: flat loads
// with 64-bit address arithmetic, 284 v_lshl_add_u64 in the value-only kernel).
template <bool LDS>
__device__ __forceinline__ void resolve_tables(DevTables& D, const double* __restrict__ blob, int words, double* lds)
{
    const double* base = blob;
    if constexpr (LDS) {
        for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = blob[i];
        __syncthreads();
        base = lds;
    }
    auto rd = [&](const double*& p) { p = base + reinterpret_cast<size_t>(p); };
    auto ri = [&](const int32_t*& p) { p = reinterpret_cast<const int32_t*>(base + reinterpret_cast<size_t>(p)); };
    opmgpu_tables& T = D.t; TabX& X = D.x;
    rd(T.surface_density); rd(T.pvtw);
    ri(T.oil_node_ptr); rd(T.oil_rs); rd(T.oil_psat); rd(T.oil_invb_sat); rd(T.oil_invbmu_sat);
    ri(T.oil_col_ptr); rd(T.oil_col_p); rd(T.oil_col_invb); rd(T.oil_col_invbmu);
    ri(T.gas_node_ptr); rd(T.gas_pg); rd(T.gas_rvsat); rd(T.gas_invb_sat); rd(T.gas_invbmu_sat);
    ri(T.gas_col_ptr); rd(T.gas_col_rv); rd(T.gas_col_invb); rd(T.gas_col_invbmu);
    ri(T.swof_ptr); rd(T.swof_sw); rd(T.swof_krw); rd(T.swof_krow); rd(T.swof_pcow);
    ri(T.sgof_ptr); rd(T.sgof_sg); rd(T.sgof_krg); rd(T.sgof_krog); rd(T.sgof_pcgo);
    rd(T.rocktab_p); rd(T.rocktab_pvmult); rd(T.rocktab_transmult);
    rd(X.swof_dkrw); rd(X.swof_dkrow); rd(X.swof_dpcow); rd(X.sgof_dkrg); rd(X.sgof_dkrog); rd(X.sgof_dpcgo);
    rd(X.oil_drs); rd(X.oil_dinvb_sat); rd(X.oil_dinvbmu_sat); rd(X.oil_col_dinvb); rd(X.oil_col_dinvbmu);
    rd(X.gas_drvsat); rd(X.gas_dinvb_sat); rd(X.gas_dinvbmu_sat); rd(X.gas_col_dinvb); rd(X.gas_col_dinvbmu);
}

// saturation table with the segment's slope tabulated: the segment search (sat_eval's rule) and the evaluation are separate so that
// curves over the same abscissa share one search.  clamp: -1 / +1 = constant extrapolation below / above the table.
template <bool RIGHT>
__device__ __forceinline__ int sat_seg(const double* __restrict__ x, int n, double xv, int& clamp)
{
    clamp = (xv <= x[0]) ? -1 : ((xv >= x[n - 1]) ? 1 : 0);
    int i = 0;
    for (int k = 1; k < n - 1; ++k) i += (RIGHT ? (x[k] <= xv) : (x[k] < xv)) ? 1 : 0;
    return i;
}
__device__ __forceinline__ void sat_at(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ dy, int n, int i, int clamp, double xv,
                                       double& f, double& df)
{
    if (clamp < 0) { f = y[0]; df = 0.0; return; }
    if (clamp > 0) { f = y[n - 1]; df = 0.0; return; }
    df = dy[i];
    f = y[i] + df * (xv - x[i]);
}
template <bool RIGHT>
__device__ __forceinline__ void sat_curve_s(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ dy, int n, double sv, const EpsD& e, int c,
                                            double& f, double& df)
{
    int cl;
    if (!e.on) { const int i = sat_seg<RIGHT>(x, n, sv, cl); sat_at(x, y, dy, n, i, cl, sv, f, df); return; }
    double slope;
    const double su = eps_map(e, c, sv, slope);
    const int i = sat_seg<RIGHT>(x, n, su, cl);
    sat_at(x, y, dy, n, i, cl, su, f, df);
    f *= e.v[c]; df *= slope * e.v[c];
}
__device__ __forceinline__ void lin_at(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ dy, int i, double xv, double& f, double& df)
{
    df = dy[i];
    f = y[i] + df * (xv - x[i]);
}
// UniformXTabulated2DFunction::eval for TWO functions over the same columns (1/B and 1/(B mu)): node segment i given, column searches shared
__device__ __forceinline__ void pvt2_pair(const double* __restrict__ xs, int i, const int32_t* __restrict__ cp, const double* __restrict__ cy,
                                          const double* __restrict__ cv1, const double* __restrict__ dcv1, const double* __restrict__ cv2, const double* __restrict__ dcv2,
                                          double xv, double yv, double& f1, double& dfx1, double& dfy1, double& f2, double& dfx2, double& dfy2)
{
    const double ih = 1.0 / (xs[i + 1] - xs[i]);
    const double alpha = (xv - xs[i]) * ih, oma = 1.0 - alpha;
    const int c0 = cp[i], c1 = cp[i + 1];
    const int j0 = pvt_seg(cy + c0, c1 - c0, yv), j1 = pvt_seg(cy + c1, cp[i + 2] - c1, yv);
    double s1, d1, s2, d2;
    lin_at(cy + c0, cv1 + c0, dcv1 + c0, j0, yv, s1, d1); lin_at(cy + c1, cv1 + c1, dcv1 + c1, j1, yv, s2, d2);
    f1 = s1 * oma + s2 * alpha; dfx1 = (s2 - s1) * ih; dfy1 = d1 * oma + d2 * alpha;
    lin_at(cy + c0, cv2 + c0, dcv2 + c0, j0, yv, s1, d1); lin_at(cy + c1, cv2 + c1, dcv2 + c1, j1, yv, s2, d2);
    f2 = s1 * oma + s2 * alpha; dfx2 = (s2 - s1) * ih; dfy2 = d1 * oma + d2 * alpha;
}
// a / b with ONE division (the reciprocal), value and derivatives
__device__ __forceinline__ V4 vdiv1(const V4& a, const V4& b)
{
    const double ib = 1.0 / b.v, q = a.v * ib;
    return mk(q, (a.p - q * b.p) * ib, (a.w - q * b.w) * ib, (a.x - q * b.x) * ib);
}

struct CellEval {
    V4 pw, pg, rs, rv, sw, so, sg;
    V4 b[3], mob[3], rho[3], accum[3];
    V4 pvm;              // pore-volume multiplier (poroMult)
};

// SolutionState + ReservoirResidualQuant of one cell (BlackoilModelBase_impl.hpp:614-751, 1484-1497, 2009-2027)
__device__ void eval_cell(const DevTables& DT, const EpsD& E, double so_max, int preg, int sreg, double p, double sw_, double sg_, double rs_, double rv_, int hc,
                          CellEval& q, const HystD& H = HystD{ false, 0, 2.0, 2.0, 0.0, 0.0 }, const EpsD* EI = nullptr)
{
    const opmgpu_tables& T = DT.t; const TabX& X = DT.x;
    const bool isSg = hc == OPMGPU_HC_GAS_AND_OIL, isRs = hc == OPMGPU_HC_OIL_ONLY, isRv = hc == OPMGPU_HC_GAS_ONLY;
    const bool freeOil = isSg || isRs, freeGas = isSg || isRv;
    const V4 P = mk(p, 1, 0, 0), W = mk(sw_, 0, 1, 0);
    const V4 Xv = mk(isRs ? rs_ : (isRv ? rv_ : sg_), 0, 0, 1);
    // sg = isSg*X + isRv*(1 - W);  so = 1 - W - sg
    V4 sg = mk(0, 0, 0, 0);
    if (isSg) sg = Xv;
    else if (isRv) sg = mk(1.0 - sw_, 0, -1, 0);
    const V4 so = mk((1.0 - sw_) - sg.v, 0, -1.0 - sg.w, -sg.x);
    q.sw = W; q.so = so; q.sg = sg;
    // saturation functions
    const int wa = T.swof_ptr[sreg], nw = T.swof_ptr[sreg + 1] - wa;
    const int ga = T.sgof_ptr[sreg], ng = T.sgof_ptr[sreg + 1] - ga;
    const double* xsw = T.swof_sw + wa; const double* xsg = T.sgof_sg + ga;
    double f, df;
    V4 krw, krg;
    const bool gas_imb = H.on && (1.0 - sg.v) > H.mdc_go;        // gas on its (shifted) imbibition curve: krn_imb(Sw + delta) = krg_imb(Sg - delta)
    if (!E.on) {
        // no end-point scaling: pcow and krw share the segment of Sw, pcgo and krg the segment of Sg
        int cw, cg;
        const int iw = sat_seg<false>(xsw, nw, sw_, cw);
        sat_at(xsw, T.swof_pcow + wa, X.swof_dpcow + wa, nw, iw, cw, sw_, f, df);
        q.pw = mk(p - f, 1, -df, 0);
        sat_at(xsw, T.swof_krw + wa, X.swof_dkrw + wa, nw, iw, cw, sw_, f, df);
        krw = mk(f, 0, df, 0);
        const int ig = sat_seg<true>(xsg, ng, sg.v, cg);
        sat_at(xsg, T.sgof_pcgo + ga, X.sgof_dpcgo + ga, ng, ig, cg, sg.v, f, df);
        q.pg = mk(p + f, 1, df * sg.w, df * sg.x);
        if (!gas_imb) { sat_at(xsg, T.sgof_krg + ga, X.sgof_dkrg + ga, ng, ig, cg, sg.v, f, df); krg = vchain(f, df, sg); }
    } else {
        sat_curve_s<false>(xsw, T.swof_pcow + wa, X.swof_dpcow + wa, nw, sw_, E, EC_PCOW, f, df);
        q.pw = mk(p - f, 1, -df, 0);
        sat_curve_s<true>(xsg, T.sgof_pcgo + ga, X.sgof_dpcgo + ga, ng, sg.v, E, EC_PCGO, f, df);
        q.pg = mk(p + f, 1, df * sg.w, df * sg.x);
        sat_curve_s<false>(xsw, T.swof_krw + wa, X.swof_dkrw + wa, nw, sw_, E, EC_KRW, f, df);
        krw = mk(f, 0, df, 0);
        if (!gas_imb) { sat_curve_s<true>(xsg, T.sgof_krg + ga, X.sgof_dkrg + ga, ng, sg.v, E, EC_KRG, f, df); krg = vchain(f, df, sg); }
    }
    if (gas_imb) {
        const int gi = T.sgof_ptr[H.ireg];
        sat_curve_s<true>(T.sgof_sg + gi, T.sgof_krg + gi, X.sgof_dkrg + gi, T.sgof_ptr[H.ireg + 1] - gi, sg.v - H.d_go, *EI, EC_KRG, f, df);
        krg = vchain(f, df, sg);
    }
    V4 kro;
    {   // EclDefaultMaterial::krn
        // connate water of the three-phase law: the cell's scaled SWL; a set without horizontal scaling (s0 = u0 = 0, k = 1) has the table's
        const bool hscaled = E.on && (E.k[EC_PCOW] != 1.0 || E.s0[EC_PCOW] != 0.0 || E.u0[EC_PCOW] != 0.0);
        const double swco = hscaled ? E.s0[EC_PCOW] : xsw[0];
        const V4 swp = (sw_ > swco) ? W : mk(swco, 0, 0, 0);
        const V4 swow = vadd(sg, swp);
        if (H.on && swow.v > H.mdc_ow) {         // oil against water on its (shifted) imbibition curve
            const int wi = T.swof_ptr[H.ireg];
            sat_curve_s<false>(T.swof_sw + wi, T.swof_krow + wi, X.swof_dkrow + wi, T.swof_ptr[H.ireg + 1] - wi, swow.v + H.d_ow, *EI, EC_KROW, f, df);
        } else sat_curve_s<false>(xsw, T.swof_krow + wa, X.swof_dkrow + wa, nw, swow.v, E, EC_KROW, f, df);
        const V4 kow = vchain(f, df, swow);
        const V4 sgeq = mk(swow.v - swco, swow.p, swow.w, swow.x);
        int cl;
        if (E.on) {     // krog is tabulated against the oil saturation 1 - Swco_table - Sg; the scaling acts on that axis
            double slope;
            const double so_u = eps_map(E, EC_KROG, 1.0 - swow.v, slope);
            const double sgu = 1.0 - xsw[0] - so_u;
            const int i = sat_seg<true>(xsg, ng, sgu, cl);
            sat_at(xsg, T.sgof_krog + ga, X.sgof_dkrog + ga, ng, i, cl, sgu, f, df);
            f *= E.v[EC_KROG]; df *= slope * E.v[EC_KROG];
        } else { const int i = sat_seg<true>(xsg, ng, sgeq.v, cl); sat_at(xsg, T.sgof_krog + ga, X.sgof_dkrog + ga, ng, i, cl, sgeq.v, f, df); }
        const V4 kgo = vchain(f, df, sgeq);
        const double eps = 1e-5;
        const V4 den = sgeq;                                    // Sw_ow - Swco
        const V4 num = vadd(vmul(sg, kgo), vmul(mk(swp.v - swco, swp.p, swp.w, swp.x), kow));
        if (swow.v - swco < eps) {
            const V4 k2 = vscale(0.5, vadd(kow, kgo));
            if (swow.v - swco > eps / 2) {
                const V4 k1 = vdiv(num, den);
                const V4 al = mk((eps - den.v) / (eps / 2), -den.p / (eps / 2), -den.w / (eps / 2), -den.x / (eps / 2));
                const V4 oma = mk(1.0 - al.v, -al.p, -al.w, -al.x);
                kro = vadd(vmul(k2, al), vmul(k1, oma));
            } else kro = k2;
        } else kro = vdiv1(num, den);
    }
    // rs / rv (saturated curves over the nodes' pressures: the node segment of p is shared with the saturated oil PVT below, that of
    // p_g with the saturated gas PVT)
    const int oa = T.oil_node_ptr[preg], on = T.oil_node_ptr[preg + 1] - oa;
    const int gna = T.gas_node_ptr[preg], gn = T.gas_node_ptr[preg + 1] - gna;
    const int ip = pvt_seg(T.oil_psat + oa, on, p);
    const int ipg = pvt_seg(T.gas_pg + gna, gn, q.pg.v);
    V4 rsSat = mk(0, 0, 0, 0), rvSat = mk(0, 0, 0, 0);
    if (T.has_disgas) { lin_at(T.oil_psat + oa, T.oil_rs + oa, X.oil_drs + oa, ip, p, f, df); rsSat = mk(f, df, 0, 0); }
    if (T.vap2 > 0.0) { vap_factor(T.vap2, so.v, so_max, f, df); rsSat = vmul(vchain(f, df, so), rsSat); }
    q.rs = (T.has_disgas && isRs) ? Xv : rsSat;
    if (T.has_vapoil) { lin_at(T.gas_pg + gna, T.gas_rvsat + gna, X.gas_drvsat + gna, ipg, q.pg.v, f, df); rvSat = vchain(f, df, q.pg); }
    if (T.vap1 > 0.0) { vap_factor(T.vap1, so.v, so_max, f, df); rvSat = vmul(vchain(f, df, so), rvSat); }
    q.rv = (T.has_vapoil && isRv) ? Xv : rvSat;
    // water PVT (ConstantCompressibilityWaterPvt)
    V4 mu[3];
    {
        const double* w = T.pvtw + 5 * preg;
        const double iw1 = 1.0 / w[1];
        const double Xc = w[2] * (q.pw.v - w[0]);
        const double bw = (1.0 + Xc * (1.0 + Xc / 2.0)) * iw1;
        const double dbw = w[2] * (1.0 + Xc) * iw1;
        q.b[0] = vchain(bw, dbw, q.pw);
        const double c = w[2] - w[4];
        const double Y = c * (q.pw.v - w[0]);
        const double den = 1.0 + Y * (1.0 + Y / 2.0), iden = 1.0 / den;
        const double BM = w[3] * w[1];
        mu[0] = vchain(BM * bw * iden, BM * (dbw * den - bw * c * (1.0 + Y)) * (iden * iden), q.pw);
    }
    // oil PVT (LiveOilPvt; saturated branch when free gas is present or no DISGAS)
    {
        double ib, dibp, dibr = 0.0, ibm, dibmp, dibmr = 0.0;
        if (freeGas || !T.has_disgas) {
            lin_at(T.oil_psat + oa, T.oil_invb_sat + oa, X.oil_dinvb_sat + oa, ip, p, ib, dibp);
            lin_at(T.oil_psat + oa, T.oil_invbmu_sat + oa, X.oil_dinvbmu_sat + oa, ip, p, ibm, dibmp);
        } else {
            const int ir = pvt_seg(T.oil_rs + oa, on, q.rs.v);
            pvt2_pair(T.oil_rs + oa, ir, T.oil_col_ptr + oa, T.oil_col_p, T.oil_col_invb, X.oil_col_dinvb, T.oil_col_invbmu, X.oil_col_dinvbmu,
                      q.rs.v, p, ib, dibr, dibp, ibm, dibmr, dibmp);
        }
        q.b[1] = vchain2(ib, dibp, P, dibr, q.rs);
        const double r = 1.0 / ibm, m = ib * r;
        mu[1] = vchain2(m, (dibp - m * dibmp) * r, P, (dibr - m * dibmr) * r, q.rs);
    }
    // gas PVT (WetGasPvt; saturated branch when free oil is present or no VAPOIL)
    {
        double ib, dibp, dibr = 0.0, ibm, dibmp, dibmr = 0.0;
        if (freeOil || !T.has_vapoil) {
            lin_at(T.gas_pg + gna, T.gas_invb_sat + gna, X.gas_dinvb_sat + gna, ipg, q.pg.v, ib, dibp);
            lin_at(T.gas_pg + gna, T.gas_invbmu_sat + gna, X.gas_dinvbmu_sat + gna, ipg, q.pg.v, ibm, dibmp);
        } else {
            pvt2_pair(T.gas_pg + gna, ipg, T.gas_col_ptr + gna, T.gas_col_rv, T.gas_col_invb, X.gas_col_dinvb, T.gas_col_invbmu, X.gas_col_dinvbmu,
                      q.pg.v, q.rv.v, ib, dibp, dibr, ibm, dibmp, dibmr);
        }
        q.b[2] = vchain2(ib, dibp, q.pg, dibr, q.rv);
        const double r = 1.0 / ibm, m = ib * r;
        mu[2] = vchain2(m, (dibp - m * dibmp) * r, q.pg, (dibr - m * dibmr) * r, q.rv);
    }
    // densities, mobilities (tr_mult == 1: no ROCKTAB)
    const double* rhos = T.surface_density + 3 * preg;
    q.rho[0] = vscale(rhos[0], q.b[0]);
    q.rho[1] = vadd(vscale(rhos[1], q.b[1]), vscale(rhos[2], vmul(q.rs, q.b[1])));
    q.rho[2] = vadd(vscale(rhos[2], q.b[2]), vscale(rhos[1], vmul(q.rv, q.b[2])));
    // accumulation with rock compressibility, transmissibility multiplier (RockCompressibility.cpp:86-125)
    V4 pvm = mk(1, 0, 0, 0);
    if (T.rocktab_n > 0) {
        rocktab_eval(T.rocktab_p, T.rocktab_pvmult, T.rocktab_n, p, f, df); pvm = mk(f, df, 0, 0);
        rocktab_eval(T.rocktab_p, T.rocktab_transmult, T.rocktab_n, p, f, df);
        const V4 trm = mk(f, df, 0, 0);
        q.mob[0] = vdiv1(vmul(trm, krw), mu[0]); q.mob[1] = vdiv1(vmul(trm, kro), mu[1]); q.mob[2] = vdiv1(vmul(trm, krg), mu[2]);
    } else {
        q.mob[0] = vdiv1(krw, mu[0]); q.mob[1] = vdiv1(kro, mu[1]); q.mob[2] = vdiv1(krg, mu[2]);
    }
    if (T.rocktab_n == 0 && T.rock_comp != 0.0) {
        const double cp = T.rock_comp * (p - T.rock_pref);
        pvm = mk(1.0 + cp + 0.5 * cp * cp, T.rock_comp + cp * T.rock_comp, 0, 0);
    }
    q.pvm = pvm;
    const V4 aw = vmul(vmul(pvm, q.b[0]), W);
    const V4 ao = vmul(vmul(pvm, q.b[1]), so);
    const V4 ag = vmul(vmul(pvm, q.b[2]), sg);
    q.accum[0] = aw;
    q.accum[1] = vadd(ao, vmul(q.rv, ag));
    q.accum[2] = vadd(ag, vmul(q.rs, ao));
}

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
// What a row needs of its NEIGHBOURS are ten VALUES per cell -- the phase pressures p_w and p_g (p_o is the state's pressure), the three
// densities, the three b * mobility products, rs and rv -- and nothing else: a row computes the derivatives of each of its connections'
// fluxes with respect to its OWN variables only, from the derivatives eval_cell gives it for its own cell, and writes them twice -- into
// its own diagonal block, and (negated) into the block (neighbour, row) of the NEIGHBOUR's matrix row, which is exactly that entry:
// dR_j / d x_i = -(dR_i / d x_i)|this connection.  Every block is still written by exactly one thread and every residual entry by its own
// row (no atomics), but the 27 derivative planes round 2 exchanged between its two kernels (written once, read ~2.8 times: 1.37 GB of
// HBM traffic for 0.41 GB of algorithmic bytes at 100^3, PMC in profiles/r02_w_pmc_summary.json) no longer exist.
//   k_cell_values : state -> the ten value planes (+ 1 / b for getConvergence)                                  [pass 1, all cells]
//   k_assemble_rows: state -> own derivatives (eval_cell again: arithmetic is free next to the bytes), accumulation term, TPFA fluxes
//                    from the neighbours' value planes, residual, diagonal block, transposed off-diagonal blocks, CPR weights  [pass 2]
template <bool LDS>
__global__ __launch_bounds__(kBlock) void k_cell_values(int nb, int nbp, DevTables D, const int32_t* __restrict__ pvtnum, const int32_t* __restrict__ satnum,
                                                        const double* __restrict__ p, const double* __restrict__ sw, const double* __restrict__ sg,
                                                        const double* __restrict__ rs, const double* __restrict__ rv, const int8_t* __restrict__ hc,
                                                        const double* __restrict__ eps, const double* __restrict__ eps_u0, const double* __restrict__ somax,
                                                        double* __restrict__ vals, double* __restrict__ bpart, const int8_t* __restrict__ mask,
                                                        const double* __restrict__ tab_blob, int tab_words, HystArgs hy)
{
    extern __shared__ double tab_lds[];
    __shared__ double bsm[12];
    resolve_tables<LDS>(D, tab_blob, tab_words, tab_lds);
    const int row = blockIdx.x * kBlock + threadIdx.x;
    double ib[3] = { 0.0, 0.0, 0.0 };
    if (row < nb) {
        CellEval q;
        EpsD E, EI;
        HystD H;
        eps_load(eps, eps_u0, nbp, row, satnum[row], E);
        hyst_load(hy.imbnum, hy.hist, nbp, row, H);
        if (H.on) eps_load(hy.ieps, hy.iureg, nbp, row, H.ireg, EI);
        eval_cell(D, E, (D.t.vap1 > 0.0 || D.t.vap2 > 0.0) ? somax[row] : 0.0, pvtnum[row], satnum[row], p[row], sw[row], sg[row], rs[row], rv[row], hc[row], q, H, &EI);
        vals[long(VP_PW) * nbp + row] = q.pw.v; vals[long(VP_PG) * nbp + row] = q.pg.v;
        const bool owned = !mask || mask[row];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            vals[long(VP_RHO + a) * nbp + row] = q.rho[a].v;
            vals[long(VP_U + a) * nbp + row] = q.b[a].v * q.mob[a].v;
            if (owned) ib[a] = 1.0 / q.b[a].v;
        }
        vals[long(VP_RS) * nbp + row] = q.rs.v; vals[long(VP_RV) * nbp + row] = q.rv.v;
    }
    // getConvergence needs only the SUMS of 1 / b_a over the owned cells (B_avg, BlackoilModelBase_impl.hpp:1650-1660): one partial per
    // workgroup here instead of three planes written now and read back by k_conv_partial (48 MB of traffic at 100^3)
    block_sum<3>(ib, bsm);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) bpart[long(a) * gridDim.x + blockIdx.x] = ib[a];
    }
}

// computeAccum / assembleMassBalanceEq / computeMassFlux / applyThresholdPressures / UpwindSelector (BlackoilModelBase_impl.hpp:709-751, 845-913,
// 1484-1545, AutoDiffHelpers.hpp:204-221; rs / rv cross terms :889-906, div = ngrad^T), one thread per row.
// Per SELL entry of a row: the neighbour's row (col), the connection's transmissibility with the row's SIDE of the connection in its sign
// bit (+: the row is c1, ngrad coefficient +1; NaN: not a connection -- the fill of an explicit well clique), g (z_c1 - z_c2), the
// threshold pressure, and the index of the transposed entry (tpos).  All four are read coalesced at known addresses: the only dependent
// loads of the loop are the neighbour's values.
template <class MS, int WAVES, bool LDS, bool BATCH, bool DUAL>
__global__ __launch_bounds__(kBlock, WAVES) void k_assemble_rows(int xm, int nb, int nbp, DevTables DT, const int32_t* __restrict__ pvtnum, const int32_t* __restrict__ satnum,
                                                          const double* __restrict__ pv, const double* __restrict__ p, const double* __restrict__ sw,
                                                          const double* __restrict__ sg, const double* __restrict__ rs, const double* __restrict__ rv,
                                                          const int8_t* __restrict__ hc, double inv_dt, int initial, double s0, double s1, double s2,
                                                          const int32_t* __restrict__ slice_ptr, const int32_t* __restrict__ col, const int16_t* __restrict__ rowlen,
                                                          const int16_t* __restrict__ nlower, const int32_t* __restrict__ tpos,
                                                          const double* __restrict__ tr_e, const double* __restrict__ zc, double grav, const double* __restrict__ thp_e,
                                                          const double* __restrict__ eps, const double* __restrict__ eps_u0, const double* __restrict__ somax,
                                                          const double* __restrict__ vals, double* __restrict__ accum0, const int8_t* __restrict__ mask,
                                                          double* __restrict__ R, MS* __restrict__ A, MS* __restrict__ wout, const int32_t* __restrict__ chunk_perm,
                                                          const double* __restrict__ tab_blob, int tab_words, HystArgs hy, float* __restrict__ A32)
{
    // A32 != nullptr (mixed precision, opmgpu_params.preconditioner_single): every block is ALSO written as float into the preconditioner's
    // copy of the matrix -- 250 MB more written here instead of a conversion pass that reads 500 MB and writes 250 MB before the solve
    extern __shared__ double tab_lds[];
    const int nchunks = (nb + kBlock - 1) / kBlock;
    const int lch = xcd_first(nchunks, xm);
    if (lch >= xcd_end(nchunks, xm)) return;               // (uniform over the workgroup)
    resolve_tables<LDS>(DT, tab_blob, tab_words, tab_lds);
    const int ch = chunk_perm[lch];
    const int row = ch * kBlock + threadIdx.x;
    if (row >= nb) return;
    const int base = slice_ptr[row >> 6], lane = row & 63, nl = nlower[row], len = rowlen[row];
    const bool ghost = mask && !mask[row];                 // multi-GPU: a ghost row is an identity row with zero residual -- the owner rank
                                                           // assembles the real equation; its thread still writes the blocks (owned row, ghost column)
    // ---- the row's own cell: values from the planes (bit for bit what the neighbours read), derivatives from eval_cell ----
    CellEval q;
    {
        EpsD E, EI;
        HystD H;
        eps_load(eps, eps_u0, nbp, row, satnum[row], E);
        hyst_load(hy.imbnum, hy.hist, nbp, row, H);
        if (H.on) eps_load(hy.ieps, hy.iureg, nbp, row, H.ireg, EI);
        eval_cell(DT, E, (DT.t.vap1 > 0.0 || DT.t.vap2 > 0.0) ? somax[row] : 0.0, pvtnum[row], satnum[row], p[row], sw[row], sg[row], rs[row], rv[row], hc[row], q, H, &EI);
    }
    const double scale[3] = { s0, s1, s2 };
    double op[3], orho[3], oU[3];                          // own values
    double dP[3][3], dRho[3][3], dU[3][3];                 // own derivatives d/d(P, Sw, Xvar) of phase pressure, density, b * mobility
    // (the row's own VALUES are read from the planes pass 1 wrote -- bit for bit what the neighbours see of this cell.  Taking them from this
    // kernel's own eval_cell instead saves 80 B per cell and was measured SLOWER, 0.239 against 0.202 ms: the values then stay live in
    // registers across the whole evaluation and the loop spills more; profiles/r03_asm_ownvals_ab.log)
    op[0] = vals[long(VP_PW) * nbp + row]; op[1] = p[row]; op[2] = vals[long(VP_PG) * nbp + row];
    dP[0][0] = 1.0; dP[0][1] = q.pw.w; dP[0][2] = 0.0;
    dP[1][0] = 1.0; dP[1][1] = 0.0; dP[1][2] = 0.0;
    dP[2][0] = 1.0; dP[2][1] = q.pg.w; dP[2][2] = q.pg.x;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        orho[a] = vals[long(VP_RHO + a) * nbp + row]; oU[a] = vals[long(VP_U + a) * nbp + row];
        const V4 u = vmul(q.b[a], q.mob[a]);
        dRho[a][0] = q.rho[a].p; dRho[a][1] = q.rho[a].w; dRho[a][2] = q.rho[a].x;
        dU[a][0] = u.p; dU[a][1] = u.w; dU[a][2] = u.x;
    }
    const double oRs = vals[long(VP_RS) * nbp + row], oRv = vals[long(VP_RV) * nbp + row];
    const double dRs[3] = { q.rs.p, q.rs.w, q.rs.x }, dRv[3] = { q.rv.p, q.rv.w, q.rv.x };
    // ---- accumulation term pvdt * (accum1 - accum0) and its part of the diagonal block ----
    const double pvdt = pv[row] * inv_dt;
    double Rl[3], D[9];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (initial) accum0[long(a) * nbp + row] = q.accum[a].v;
        const double a0 = initial ? q.accum[a].v : accum0[long(a) * nbp + row];
        Rl[a] = pvdt * (q.accum[a].v - a0);
        D[3 * a] = scale[a] * pvdt * q.accum[a].p; D[3 * a + 1] = scale[a] * pvdt * q.accum[a].w; D[3 * a + 2] = scale[a] * pvdt * q.accum[a].x;
    }
    // CPR weights (formEllipticSystem, see k_cpr_weights): column sums of |dR_j[eq]/dp_i| over the rows j != i -- the blocks (j, i) this
    // row writes itself, so the sums come for free (systems without explicit well cliques; k_cpr_weights stays for everything else)
    double sod[3] = { 0.0, 0.0, 0.0 };
    int k0 = (nl == 0) ? 1 : 0;
    // per entry: the neighbour (4 B), the transmissibility with the side in its sign bit (8 B), the index of the transposed entry (4 B; as a
    // one-byte slot of the neighbour's row it was measured slower: one more dependent load per connection, 0.240 against 0.210 ms), the
    // threshold pressure where the deck has one.  g (z_c1 - z_c2) is formed from the depth plane: the neighbour's depth rides on the gather
    // of its values (round 3 kept a per-entry word for it: 55 MB at 100^3 against ~16 MB now; 0.210 against 0.245 ms, profiles/r04_f_ab.log)
    const double z_own = zc[row];
    int nbr_n = 0, tp_n = 0; double T_n = 0.0, th_n = 0.0;
    if (k0 < len) {
        const long e0 = long(base + k0) * 64 + lane;
        nbr_n = __builtin_nontemporal_load(&col[e0]); tp_n = __builtin_nontemporal_load(&tpos[e0]);
        T_n = __builtin_nontemporal_load(&tr_e[e0]);
        if (thp_e) th_n = __builtin_nontemporal_load(&thp_e[e0]);
    }
    for (int k = k0; k < len; ) {
        const int nbr = nbr_n, tp = tp_n; const double Te = T_n, thp = th_n;
        const int kn = (k + 1 == nl) ? k + 2 : k + 1;
        if (kn < len) {           // the next entry's words are in flight while this one is computed
            const long en = long(base + kn) * 64 + lane;
            nbr_n = __builtin_nontemporal_load(&col[en]); tp_n = __builtin_nontemporal_load(&tpos[en]);
            T_n = __builtin_nontemporal_load(&tr_e[en]);
            if (thp_e) th_n = __builtin_nontemporal_load(&thp_e[en]);
        }
        MS* bptr = A + long(base + k) * 576 + lane;
        float* bptr32 = DUAL ? A32 + long(base + k) * 576 + lane : nullptr;
        k = kn;
        if (Te != Te) {          // pure well fill: the host adds the Schur block later
#pragma unroll
            for (int c = 0; c < 9; ++c) { bptr[c * 64] = MS(0); if (DUAL) bptr32[c * 64] = 0.f; }
            continue;
        }
        const int side = __builtin_signbit(Te) ? 1 : 0;     // side 0: this row is c1 (ngrad +1), side 1: it is c2
        const double Tf = fabs(Te);
        const double z_n = zc[nbr];
        const double g = grav * (side ? z_n - z_own : z_own - z_n);          // g (z_c1 - z_c2): the expression the host evaluated per entry in round 3
        // neighbour values that every connection needs: phase pressures and densities
        double np_[3], nrho[3];
        np_[0] = vals[long(VP_PW) * nbp + nbr]; np_[1] = p[nbr]; np_[2] = vals[long(VP_PG) * nbp + nbr];
#pragma unroll
        for (int a = 0; a < 3; ++a) nrho[a] = vals[long(VP_RHO + a) * nbp + nbr];
        double nU[3] = { 0.0, 0.0, 0.0 }, nRs = 0.0, nRv = 0.0;
        if (BATCH) {
#pragma unroll
            for (int a = 0; a < 3; ++a) nU[a] = vals[long(VP_U + a) * nbp + nbr];
            nRs = vals[long(VP_RS) * nbp + nbr]; nRv = vals[long(VP_RV) * nbp + nbr];
        }
        double Tdh[3], ddh[3][3];
        bool own_up[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double p1 = side ? np_[a] : op[a], p2 = side ? op[a] : np_[a];
            const double r1 = side ? nrho[a] : orho[a], r2 = side ? orho[a] : nrho[a];
            double dh = (p1 - p2) - g * (0.5 * r1 + 0.5 * r2);
            double keep = 1.0;
            if (thp_e) {
                keep = (fabs(dh) >= thp) ? 1.0 : 0.0;
                const double sg_ = (dh > 0.0) ? 1.0 : ((dh < 0.0) ? -1.0 : 0.0);
                dh = keep * (dh - sg_ * thp);
            }
            const double hg = 0.5 * g, sp = side ? -1.0 : 1.0;          // d dh / d(own variables): own is c1 (+) or c2 (-) in the pressure difference
#pragma unroll
            for (int v = 0; v < 3; ++v) ddh[a][v] = keep * (sp * dP[a][v] - hg * dRho[a][v]);
            own_up[a] = ((dh >= 0.0) ? 0 : 1) == side;                   // upwind cell = c1 if dh >= 0 else c2
            Tdh[a] = Tf * dh;
        }
        // upwind-dependent values of the neighbour.  BATCH: all five loaded with the pressures and densities above (one dependent round trip
        // per connection instead of two; the lines are fetched by some lane of the wave anyway); otherwise only the ones the upwinding asks for
        double Uv[3] = { oU[0], oU[1], oU[2] }, rsu = oRs, rvu = oRv;
        if (BATCH) {
#pragma unroll
            for (int a = 0; a < 3; ++a) Uv[a] = own_up[a] ? oU[a] : nU[a];
            rsu = own_up[1] ? oRs : nRs; rvu = own_up[2] ? oRv : nRv;
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) if (!own_up[a]) Uv[a] = vals[long(VP_U + a) * nbp + nbr];
            if (!own_up[1]) rsu = vals[long(VP_RS) * nbp + nbr];
            if (!own_up[2]) rvu = vals[long(VP_RV) * nbp + nbr];
        }
        double F[3], dF[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            F[a] = Uv[a] * Tdh[a];
#pragma unroll
            for (int v = 0; v < 3; ++v) dF[a][v] = Uv[a] * (Tf * ddh[a][v]) + (own_up[a] ? dU[a][v] * Tdh[a] : 0.0);
        }
        // G_o = F_o + rv_up(g) F_g ; G_g = F_g + rs_up(o) F_o
        const double G[3] = { F[0], F[1] + rvu * F[2], F[2] + rsu * F[1] };
        double dG[3][3];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            dG[0][v] = dF[0][v];
            dG[1][v] = dF[1][v] + rvu * dF[2][v] + (own_up[2] ? dRv[v] * F[2] : 0.0);
            dG[2][v] = dF[2][v] + rsu * dF[1][v] + (own_up[1] ? dRs[v] * F[1] : 0.0);
        }
        const double s = side ? -1.0 : 1.0;
        // the neighbour's row sees this flux with the opposite sign: block (nbr, row) = -s scale dG / d(own); zero where that row is a ghost's
        const bool nbr_ghost = mask && !mask[nbr];
        MS* tptr = A + long(tp >> 6) * 576 + (tp & 63);
        float* tptr32 = DUAL ? A32 + long(tp >> 6) * 576 + (tp & 63) : nullptr;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            Rl[a] += s * G[a];
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const double own = s * scale[a] * dG[a][v];
                if (v == 0) sod[a] += fabs(own);
                D[3 * a + v] += own;
                __builtin_nontemporal_store(nbr_ghost ? MS(0) : MS(-own), &tptr[(3 * a + v) * 64]);
                if (DUAL) __builtin_nontemporal_store(nbr_ghost ? 0.f : float(-own), &tptr32[(3 * a + v) * 64]);
            }
        }
    }
    MS* dptr = A + long(base + nl) * 576 + lane;
    float* dptr32 = DUAL ? A32 + long(base + nl) * 576 + lane : nullptr;
    if (ghost) {
#pragma unroll
        for (int c = 0; c < 9; ++c) { dptr[c * 64] = (c == 0 || c == 4 || c == 8) ? MS(1) : MS(0); if (DUAL) dptr32[c * 64] = (c == 0 || c == 4 || c == 8) ? 1.f : 0.f; }
        R[row] = 0.0; R[nbp + row] = 0.0; R[2 * long(nbp) + row] = 0.0;
        if (wout) { wout[row] = MS(1); wout[nbp + row] = MS(0); wout[2 * long(nbp) + row] = MS(0); }      // identity row: its pressure entry
        return;
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) { __builtin_nontemporal_store(MS(D[c]), &dptr[c * 64]); if (DUAL) __builtin_nontemporal_store(float(D[c]), &dptr32[c * 64]); }
    if (wout) {
        const bool w_ = fabs(D[0]) / sod[0] > 0.01, g_ = fabs(D[6]) / sod[2] > 0.01;       // NaN (0/0) compares false like the reference's Eigen cast
        bool o_ = fabs(D[3]) / sod[1] > 0.01;
        if (!o_ && !w_ && !g_) o_ = true;
        wout[row] = w_ ? MS(1) : MS(0); wout[nbp + row] = o_ ? MS(1) : MS(0); wout[2 * long(nbp) + row] = g_ ? MS(1) : MS(0);
    }
    R[row] = Rl[0]; R[nbp + row] = Rl[1]; R[2 * long(nbp) + row] = Rl[2];
}

// mixed precision: the device well model adds its own-cell derivatives to the DOUBLE diagonal blocks of the perforated cells after the
// reservoir assembly; their float copies follow
__global__ __launch_bounds__(kBlock) void k_refresh_f32_diag(int nperf, const int32_t* __restrict__ rows, const int32_t* __restrict__ slice_ptr,
                                                            const int16_t* __restrict__ nlower, const double* __restrict__ A, float* __restrict__ A32)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= nperf) return;
    const int row = rows[j];
    const long e = long(slice_ptr[row >> 6] + nlower[row]) * 576 + (row & 63);
#pragma unroll
    for (int c = 0; c < 9; ++c) A32[e + c * 64] = float(A[e + c * 64]);
}

// convergenceReduction (BlackoilModelBase_impl.hpp:1633-1714): per phase sum(1/b), sum R, non-finite flag (sums: slots 0..6),
// max|R|/pv, max|R| (maxima: slots 7..12); owned rows only (multi-GPU: followed by an all-reduce of each group)
__device__ __forceinline__ bool conv_is_max(int q) { return q >= 7; }
__global__ __launch_bounds__(kBlock) void k_conv_partial(int nb, int nbp, const double* __restrict__ R, const double* __restrict__ bpart, int nbpart,
                                                         const double* __restrict__ pv, const int8_t* __restrict__ mask, double* __restrict__ part)
{
    __shared__ double sm[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double vals[13];
#pragma unroll
    for (int q = 0; q < 13; ++q) vals[q] = 0.0;
    // sums of 1 / b: the per-workgroup partials of k_cell_values (owned cells only), spread over this launch's threads in a fixed order
    for (long i = blockIdx.x * long(kBlock) + threadIdx.x; i < nbpart; i += long(gridDim.x) * kBlock) {
#pragma unroll
        for (int a = 0; a < 3; ++a) vals[a] += bpart[long(a) * nbpart + i];
    }
    for (long i = blockIdx.x * long(kBlock) + threadIdx.x; i < nb; i += long(gridDim.x) * kBlock) {
        if (mask && !mask[i]) continue;
        const double ipv = 1.0 / pv[i];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double r = R[long(a) * nbp + i];
            vals[3 + a] += r;
            vals[7 + a] = fmax(vals[7 + a], fabs(r) * ipv);
            vals[10 + a] = fmax(vals[10 + a], fabs(r));
            if (!(fabs(r) <= 1.79e308)) vals[6] = 1.0;
        }
    }
#pragma unroll
    for (int q = 0; q < 13; ++q) {
        const bool is_max = conv_is_max(q);
        const double s = is_max ? wave_max(vals[q]) : wave_sum(vals[q]);
        __syncthreads();
        if (lane == 0) sm[w] = s;
        __syncthreads();
        if (threadIdx.x == 0) part[long(q) * gridDim.x + blockIdx.x] = is_max ? fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3])) : (sm[0] + sm[1]) + (sm[2] + sm[3]);
    }
}
__global__ __launch_bounds__(kBlock) void k_conv_final(int nblocks, const double* __restrict__ part, double* __restrict__ out)
{
    // one workgroup per scalar (13 of them), fixed order within each
    __shared__ double sm[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x;
    const bool is_max = conv_is_max(q);
    double v = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += kBlock) { const double x = part[long(q) * nblocks + i]; v = is_max ? fmax(v, x) : v + x; }
    const double s_ = is_max ? wave_max(v) : wave_sum(v);
    if (lane == 0) sm[w] = s_;
    __syncthreads();
    if (threadIdx.x == 0) out[q] = is_max ? fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3])) : (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// decomposed runs: the 7 sums and 6 (+ 6) maxima of getConvergence through ONE sum-all-reduce (see BlackoilDevice::convergence).
// phase 0: table[rank][:] = red[:], every other row zero; phase 1: red[q] = sum (q < 7) or max over the ranks' rows, in rank order
__global__ __launch_bounds__(kBlock) void k_conv_gather(int phase, int nv, int nranks, int rank, double* __restrict__ red, double* __restrict__ table)
{
    if (phase == 0) {
        for (int i = threadIdx.x; i < nranks * nv; i += kBlock) table[i] = (i / nv == rank) ? red[i % nv] : 0.0;
        return;
    }
    for (int q = threadIdx.x; q < nv; q += kBlock) {
        double v = table[q];
        for (int r = 1; r < nranks; ++r) { const double x = table[r * nv + q]; v = conv_is_max(q) ? fmax(v, x) : v + x; }
        red[q] = v;
    }
}

// updateState (BlackoilModelBase_impl.hpp:1147-1389), one thread per cell
__global__ __launch_bounds__(kBlock) void k_update_state(int nb, int nbp, opmgpu_tables T, const int32_t* __restrict__ pvtnum,
                                                         const int32_t* __restrict__ satnum, const double* __restrict__ dx, double relax,
                                                         double dp_max_rel, double ds_max, double dr_max_rel,
                                                         double* __restrict__ p, double* __restrict__ sw, double* __restrict__ so,
                                                         double* __restrict__ sg, double* __restrict__ rs, double* __restrict__ rv, int8_t* __restrict__ hc,
                                                         const double* __restrict__ eps_planes, const double* __restrict__ eps_u0,
                                                         const double* __restrict__ somax, const double* __restrict__ tab_blob, int tab_words)
{
    extern __shared__ double tab_lds[];
    stage_tables(T, tab_blob, tab_words, tab_lds);
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= nb) return;
    const double eps = 1.4901161193847656e-08;      // sqrt(DBL_EPSILON)
    const int preg = pvtnum[c], sreg = satnum[c];
    const int h = hc[c];
    const bool isSg = h == OPMGPU_HC_GAS_AND_OIL, isRs = h == OPMGPU_HC_OIL_ONLY, isRv = h == OPMGPU_HC_GAS_ONLY;
    const double dp = relax * dx[c], dsw = relax * dx[nbp + c], dxv = relax * dx[2 * long(nbp) + c];
    auto sgn = [](double x) { return (x > 0.0) ? 1.0 : ((x < 0.0) ? -1.0 : 0.0); };
    const double p_old = p[c];
    const double pn = fmax(p_old - sgn(dp) * fmin(fabs(dp), dp_max_rel * fabs(p_old)), 0.0);
    const double sw_old = sw[c], so_old = so[c], sg_old = sg[c];
    const double dsg = (isSg ? dxv : 0.0) - (isRv ? dsw : 0.0);
    const double dso = -dsw - dsg;
    const double maxVal = fmax(fabs(dso), fmax(fabs(dsg), fabs(dsw)));
    const double step = fmin(ds_max / maxVal, 1.0);
    double w_ = sw_old - step * dsw, g_ = sg_old - step * dsg, o_ = so_old - step * dso;
    if (g_ < 0) { w_ = w_ / (1 - g_); o_ = o_ / (1 - g_); g_ = 0; }
    if (o_ < 0) { w_ = w_ / (1 - o_); g_ = g_ / (1 - o_); o_ = 0; }
    if (w_ < 0) { o_ = o_ / (1 - w_); g_ = g_ / (1 - w_); w_ = 0; }
    const double rs_old = rs[c], rv_old = rv[c];
    double rsn = rs_old, rvn = rv_old;
    if (T.has_disgas) {
        const double d = isRs ? dxv : 0.0;
        rsn = fmax(rs_old - sgn(d) * fmin(fabs(d), fmax(fabs(rs_old) * dr_max_rel, 1.0)), 0.0);
    }
    if (T.has_vapoil) {
        const double d = isRv ? dxv : 0.0;
        rvn = fmax(rv_old - sgn(d) * fmin(fabs(d), fmax(fabs(rv_old) * dr_max_rel, 1e-3)), 0.0);
    }
    const bool watOnly = w_ > (1 - eps);
    int hn = OPMGPU_HC_GAS_AND_OIL;
    double f, df;
    if (T.has_disgas) {
        double v0, v1;
        vap_factor(T.vap2, so_old, somax[c], v0, df);
        vap_factor(T.vap2, o_, somax[c], v1, df);
        rs_sat_d(T, preg, p_old, f, df); const double rsSat0 = v0 * f;
        rs_sat_d(T, preg, pn, f, df); const double rsSat = v1 * f;
        const bool hasGas = (g_ > 0 && !isRs);
        const bool gasVaporized = ((rsn > rsSat * (1 + eps) && isRs) && (rs_old > rsSat0 * (1 - eps)));
        if (watOnly || hasGas || gasVaporized) { rsn = rsSat; if (watOnly) { o_ = 0; g_ = 0; rsn = 0; } }
        else hn = OPMGPU_HC_OIL_ONLY;
    }
    if (T.has_vapoil) {
        const int ga = T.sgof_ptr[sreg], ng = T.sgof_ptr[sreg + 1] - ga;
        EpsD E;
        eps_load(eps_planes, eps_u0, nbp, c, sreg, E);
        sat_curve<true>(T.sgof_sg + ga, T.sgof_pcgo + ga, ng, sg_old, E, EC_PCGO, f, df); const double pg_old = p_old + f;
        sat_curve<true>(T.sgof_sg + ga, T.sgof_pcgo + ga, ng, g_, E, EC_PCGO, f, df); const double pg_new = pn + f;
        double v0, v1;
        vap_factor(T.vap1, so_old, somax[c], v0, df);
        vap_factor(T.vap1, o_, somax[c], v1, df);
        rv_sat_d(T, preg, pg_old, f, df); const double rvSat0 = v0 * f;
        rv_sat_d(T, preg, pg_new, f, df); const double rvSat = v1 * f;
        const bool hasOil = (o_ > 0 && !isRv);
        const bool oilCondensed = ((rvn > rvSat * (1 + eps) && isRv) && (rv_old > rvSat0 * (1 - eps)));
        if (watOnly || hasOil || oilCondensed) { rvn = rvSat; if (watOnly) { o_ = 0; g_ = 0; rvn = 0; } }
        else hn = OPMGPU_HC_GAS_ONLY;
    }
    p[c] = pn; sw[c] = w_; so[c] = o_; sg[c] = g_;
    if (T.has_disgas) rs[c] = rsn;
    if (T.has_vapoil) rv[c] = rvn;
    hc[c] = int8_t(hn);
}

// b = matbalscale * R in the solver's precision (NewtonIterationBlackoilInterleaved.cpp:234-236, 263-269)
template <class S>
__global__ __launch_bounds__(kBlock) void k_build_rhs(int nb, int nbp, double s0, double s1, double s2, const double* __restrict__ R,
                                                      const double* __restrict__ extra, S* __restrict__ b)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nb) return;
    const long i1 = nbp + i, i2 = 2 * long(nbp) + i;
    double r0 = R[i], r1 = R[i1], r2 = R[i2];
    if (extra) { r0 += extra[i]; r1 += extra[i1]; r2 += extra[i2]; }
    b[i] = S(s0 * r0); b[i1] = S(s1 * r1); b[i2] = S(s2 * r2);
}
__global__ __launch_bounds__(kBlock) void k_gather_perf3(int nperf, int nbp, const int32_t* __restrict__ cells, const double* __restrict__ v, double* __restrict__ out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nperf) return;
    const int c = cells[i];
    out[3 * long(i)] = v[c]; out[3 * long(i) + 1] = v[nbp + c]; out[3 * long(i) + 2] = v[2 * long(nbp) + c];
}

// per-perforation properties for the host well model (extractWellPerfProperties)
__global__ __launch_bounds__(kBlock) void k_perf_props(int nperf, DevTables T, const int32_t* __restrict__ cells, const int32_t* __restrict__ pvtnum,
                                                       const int32_t* __restrict__ satnum, const double* __restrict__ p, const double* __restrict__ sw,
                                                       const double* __restrict__ sg, const double* __restrict__ rs, const double* __restrict__ rv,
                                                       const int8_t* __restrict__ hc, const double* __restrict__ eps,
                                                       const double* __restrict__ eps_u0, const double* __restrict__ somax, long nbp,
                                                       double* __restrict__ out, HystArgs hy)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nperf) return;
    const int c = cells[i];
    CellEval q;
    EpsD E, EI;
    HystD H;
    eps_load(eps, eps_u0, nbp, c, satnum[c], E);
    hyst_load(hy.imbnum, hy.hist, nbp, c, H);
    if (H.on) eps_load(hy.ieps, hy.iureg, nbp, c, H.ireg, EI);
    eval_cell(T, E, somax[c], pvtnum[c], satnum[c], p[c], sw[c], sg[c], rs[c], rv[c], hc[c], q, H, &EI);
    const V4 list[9] = { mk(p[c], 1, 0, 0), q.rs, q.rv, q.b[0], q.b[1], q.b[2], q.mob[0], q.mob[1], q.mob[2] };
    double* o = out + long(i) * OPMGPU_PERF_K;
#pragma unroll
    for (int k = 0; k < 9; ++k) { o[4 * k] = list[k].v; o[4 * k + 1] = list[k].p; o[4 * k + 2] = list[k].w; o[4 * k + 3] = list[k].x; }
}

// computeFluidInPlace, the per-cell part (BlackoilModelBase_impl.hpp:2263-2296): fip[phase] = ((pv_mult * b_phase) * s_phase) * pv with b at the
// phase pressures and the cell's phase condition, dissolved gas = rs * fip[oil], vaporised oil = rv * fip[gas]; plus what the region
// loops need of the state (pore volume, pressure, so + sg).  Output in the CALLER's cell order: out[q * nc + nat[row]], q = 0..7.
__global__ __launch_bounds__(kBlock) void k_fip_cells(int nc, DevTables T, const int32_t* __restrict__ nat, const int32_t* __restrict__ pvtnum,
                                                      const int32_t* __restrict__ satnum, const double* __restrict__ pv, const double* __restrict__ p,
                                                      const double* __restrict__ sw, const double* __restrict__ so_, const double* __restrict__ sg, const double* __restrict__ rs,
                                                      const double* __restrict__ rv, const int8_t* __restrict__ hc, const double* __restrict__ eps,
                                                      const double* __restrict__ eps_u0, const double* __restrict__ somax, long nbp,
                                                      double* __restrict__ out, HystArgs hy)
{
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= nc) return;
    CellEval q;
    EpsD E, EI;
    HystD H;
    eps_load(eps, eps_u0, nbp, c, satnum[c], E);
    hyst_load(hy.imbnum, hy.hist, nbp, c, H);
    if (H.on) eps_load(hy.ieps, hy.iureg, nbp, c, H.ireg, EI);
    eval_cell(T, E, somax[c], pvtnum[c], satnum[c], p[c], sw[c], sg[c], rs[c], rv[c], hc[c], q, H, &EI);
    const long n = nat[c];
    const double so = so_[c], sgv = sg[c], swv = sw[c];          // the state's saturations, as the reference takes them
    const double fw = ((q.pvm.v * q.b[0].v) * swv) * pv[c];
    const double fo = ((q.pvm.v * q.b[1].v) * so) * pv[c];
    const double fg = ((q.pvm.v * q.b[2].v) * sgv) * pv[c];
    out[0 * long(nc) + n] = fw; out[1 * long(nc) + n] = fo; out[2 * long(nc) + n] = fg;
    out[3 * long(nc) + n] = rs[c] * fo; out[4 * long(nc) + n] = rv[c] * fg;
    out[5 * long(nc) + n] = pv[c]; out[6 * long(nc) + n] = p[c]; out[7 * long(nc) + n] = so + sgv;
}

// computePropertiesForWellConnectionPressures (StandardWells_impl.hpp:218-296): b_w, b_o, b_g, rsSat, rvSat of the perforated cells at
// GIVEN pressures (the average well-block pressures) with the cells' own rs / rv / phase condition / oil saturation
__global__ __launch_bounds__(kBlock) void k_perf_pvt(int nperf, opmgpu_tables T, const int32_t* __restrict__ cells, const int32_t* __restrict__ pvtnum,
                                                     const double* __restrict__ so, const double* __restrict__ rs, const double* __restrict__ rv,
                                                     const int8_t* __restrict__ hc, const double* __restrict__ somax, const double* __restrict__ press,
                                                     const int32_t* __restrict__ gate, double* __restrict__ out)
{
    if (gate && !*gate) return;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nperf) return;
    const int c = cells[i], preg = pvtnum[c], h = hc[c];
    const bool freeGas = h != OPMGPU_HC_OIL_ONLY, freeOil = h != OPMGPU_HC_GAS_ONLY;
    const double p = press[i];
    double f, df, d2;
    double* o = out + 5 * long(i);
    {
        const double* w = T.pvtw + 5 * preg;
        const double Xc = w[2] * (p - w[0]);
        o[0] = (1.0 + Xc * (1.0 + Xc / 2.0)) / w[1];
    }
    {
        const int a = T.oil_node_ptr[preg], nn = T.oil_node_ptr[preg + 1] - a;
        if (freeGas || !T.has_disgas) pvt1(T.oil_psat + a, T.oil_invb_sat + a, nn, p, f, df);
        else pvt2(T.oil_rs + a, nn, T.oil_col_ptr + a, T.oil_col_p, T.oil_col_invb, rs[c], p, f, d2, df);
        o[1] = f;
    }
    {
        const int a = T.gas_node_ptr[preg], nn = T.gas_node_ptr[preg + 1] - a;
        if (freeOil || !T.has_vapoil) pvt1(T.gas_pg + a, T.gas_invb_sat + a, nn, p, f, df);
        else pvt2(T.gas_pg + a, nn, T.gas_col_ptr + a, T.gas_col_rv, T.gas_col_invb, p, rv[c], f, df, d2);
        o[2] = f;
    }
    double v;
    rs_sat_d(T, preg, p, f, df);
    if (T.vap2 > 0.0) { vap_factor(T.vap2, so[c], somax[c], v, df); f *= v; }
    o[3] = f;
    rv_sat_d(T, preg, p, f, df);
    if (T.vap1 > 0.0) { vap_factor(T.vap1, so[c], somax[c], v, df); f *= v; }
    o[4] = f;
}

// RateConverter::SurfaceToReservoirVoidage::calcCoeff (RateConverterLegacy.hpp:495-548): coefficients c with q_rT = sum_p c[p] q_s[p] at a
// region's average state (p, rs, rv) -- b_w(p), the UNDERSATURATED b_o(p, rs) and b_g(p, rv) (FluidSystem::oilPvt().inverseFormationVolumeFactor(
// region, T, p, Rs): the tables are evaluated at the given ratios whatever the saturated curve says), detR = 1 - rs rv.
__global__ __launch_bounds__(kBlock) void k_voidage_coeff(int n, opmgpu_tables T, const double* __restrict__ press, const double* __restrict__ rs_,
                                                          const double* __restrict__ rv_, const int32_t* __restrict__ pvtreg, double* __restrict__ coeff)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int preg = pvtreg ? pvtreg[i] : 0;
    const double p = press[i], rs = rs_[i], rv = rv_[i];
    double f, df, d2, bw, bo, bg;
    {
        const double* w = T.pvtw + 5 * preg;
        const double Xc = w[2] * (p - w[0]);
        bw = (1.0 + Xc * (1.0 + Xc / 2.0)) / w[1];
    }
    {
        const int a = T.oil_node_ptr[preg], nn = T.oil_node_ptr[preg + 1] - a;
        if (!T.has_disgas) pvt1(T.oil_psat + a, T.oil_invb_sat + a, nn, p, f, df);
        else pvt2(T.oil_rs + a, nn, T.oil_col_ptr + a, T.oil_col_p, T.oil_col_invb, rs, p, f, d2, df);
        bo = f;
    }
    {
        const int a = T.gas_node_ptr[preg], nn = T.gas_node_ptr[preg + 1] - a;
        if (!T.has_vapoil) pvt1(T.gas_pg + a, T.gas_invb_sat + a, nn, p, f, df);
        else pvt2(T.gas_pg + a, nn, T.gas_col_ptr + a, T.gas_col_rv, T.gas_col_invb, p, rv, f, df, d2);
        bg = f;
    }
    const double detR = 1.0 - (rs * rv);
    double cw = 0.0, co = 0.0, cg = 0.0;
    cw = 1.0 / bw;                                  // q[w]_r = q[w]_s / bw
    {
        const double den = bo * detR;               // q[o]_r = 1/(bo (1 - rs rv)) (q[o]_s - rv q[g]_s)
        co += 1.0 / den;
        cg -= rv / den;
    }
    {
        const double den = bg * detR;               // q[g]_r = 1/(bg (1 - rs rv)) (q[g]_s - rs q[o]_s)
        cg += 1.0 / den;
        co -= rs / den;
    }
    coeff[3 * long(i) + 0] = cw; coeff[3 * long(i) + 1] = co; coeff[3 * long(i) + 2] = cg;
}

__global__ __launch_bounds__(kBlock) void k_add_well_resid(int nperf, int nbp, const int32_t* __restrict__ cells, const double* __restrict__ delta, double* __restrict__ R)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nperf) return;
    for (int a = 0; a < 3; ++a) atomicAdd(&R[long(a) * nbp + cells[i]], delta[3 * i + a]);
}
__global__ __launch_bounds__(kBlock) void k_add_well_blocks(int nblk, const int32_t* __restrict__ entries, const double* __restrict__ blocks,
                                                            double s0, double s1, double s2, double* __restrict__ A)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nblk) return;
    const int e = entries[i];
    const double scale[3] = { s0, s1, s2 };
    double* o = A + long(e >> 6) * 576 + (e & 63);
    for (int a = 0; a < 3; ++a) for (int v = 0; v < 3; ++v) atomicAdd(&o[(3 * a + v) * 64], scale[a] * blocks[9 * long(i) + 3 * a + v]);
}

// stabilizeNonlinearUpdate: dx_old <- dx; dx <- omega*dx (+ (1-omega)*previous dx_old for SOR)
__global__ __launch_bounds__(kBlock) void k_stabilize(long n, int sor, double omega, double* __restrict__ dx, double* __restrict__ dx_old)
{
    const long i = long(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const double d = dx[i], o = dx_old[i];
    dx_old[i] = d;
    if (omega == 1.0) return;
    dx[i] = sor ? omega * d + (1.0 - omega) * o : omega * d;
}

template <class T_>
__global__ __launch_bounds__(kBlock) void k_permute(int nb, const int32_t* __restrict__ nat, const T_* __restrict__ in, T_* __restrict__ out, int to_internal)
{
    const int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= nb) return;
    if (to_internal) out[r] = in[nat[r]]; else out[nat[r]] = in[r];
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
BlackoilDevice::BlackoilDevice(hipStream_t s, LinSolver& ls_, const opmgpu_grid* g, const opmgpu_tables* t, const opmgpu_params* prm_)
    : prm(*prm_), stream(s), ls(ls_)
{
    nc = g->nc; nconn = g->nconn; gravity = g->gravity;
    n_owned_cells = nc;
    h_conn.assign(g->conn_cells, g->conn_cells + 2 * size_t(nconn));
    h_trans.assign(g->trans, g->trans + nconn);
    h_pv.assign(g->pv, g->pv + nc);
    h_z.assign(g->z, g->z + nc);
    use_thpres = g->thpres != nullptr;
    if (use_thpres) h_thpres.assign(g->thpres, g->thpres + nconn);
    h_pvtnum.assign(nc, 0); h_satnum.assign(nc, 0);
    if (g->pvtnum) h_pvtnum.assign(g->pvtnum, g->pvtnum + nc);
    if (g->satnum) h_satnum.assign(g->satnum, g->satnum + nc);
    has_endpoints = g->eps[0] != nullptr;
    if (has_endpoints)
        for (int k = 0; k < 8; ++k) {
            if (!g->eps[k]) throw HipError(OPMGPU_EINVAL, "ENDSCALE needs all eight end-point arrays (SWL SWCR SWU SOWCR SGL SGCR SGU SOGCR)");
            h_eps[k].assign(g->eps[k], g->eps[k] + nc);
        }
    scalecrs = g->scalecrs != 0;
    if (scalecrs && !has_endpoints) throw HipError(OPMGPU_EINVAL, "SCALECRS needs the end-point arrays");
    bool vert = false;
    for (int k = 0; k < 5; ++k) if (g->eps_v[k]) { h_eps_v[k].assign(g->eps_v[k], g->eps_v[k] + nc); vert = true; }
    use_hyst = g->imbnum != nullptr;
    if (use_hyst) {
        h_imbnum.assign(g->imbnum, g->imbnum + nc);
        for (int c = 0; c < nc; ++c) if (h_imbnum[c] < 0 || h_imbnum[c] >= t->n_sat_regions) throw HipError(OPMGPU_EINVAL, "IMBNUM region out of range");
        has_iendpoints = g->ieps[0] != nullptr;
        if (has_iendpoints)
            for (int k = 0; k < 8; ++k) {
                if (!g->ieps[k]) throw HipError(OPMGPU_EINVAL, "imbibition end points: all eight arrays or none");
                h_ieps[k].assign(g->ieps[k], g->ieps[k] + nc);
            }
    }
    use_eps = has_endpoints || vert || use_hyst;
    pvsum = 0.0;
    for (int c = 0; c < nc; ++c) pvsum += h_pv[c];
    upload_tables(t);
    OPMGPU_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_red), 32 * sizeof(double)));
    h_well_connpos.assign(1, 0);
    rebuild_structure();
}

BlackoilDevice::~BlackoilDevice()
{
    wells_free();
    vfp_free();
    if (h_red) (void)hipHostFree(h_red);
    for (auto e : ev_well) if (e) (void)hipEventDestroy(e);
    if (well_stream) (void)hipStreamDestroy(well_stream);
}

void BlackoilDevice::upload_tables(const opmgpu_tables* t)
{
    dt_ = *t;
    // all table arrays live in ONE device blob (8-byte words) so that a kernel can stage them in LDS with one cooperative copy
    std::vector<double> blob;
    auto upd = [&](const double* src, size_t n) -> const double* {
        const size_t off = blob.size();
        blob.insert(blob.end(), src, src + n);
        if (n == 0) blob.push_back(0.0);
        return reinterpret_cast<const double*>(off + 1);         // encoded offset (+1: never null), resolved below
    };
    auto upi = [&](const int32_t* src, size_t n) -> const int32_t* {
        const size_t off = blob.size();
        blob.resize(off + (n + 1) / 2 + 1, 0.0);
        std::memcpy(&blob[off], src, n * sizeof(int32_t));
        return reinterpret_cast<const int32_t*>(off + 1);
    };
    const int np = t->n_pvt_regions, ns = t->n_sat_regions;
    const int non = t->oil_node_ptr[np], ngn = t->gas_node_ptr[np];
    const int noc = t->oil_col_ptr[non], ngc = t->gas_col_ptr[ngn];
    const int nsw = t->swof_ptr[ns], nsg = t->sgof_ptr[ns];
    h_surface_density.assign(t->surface_density, t->surface_density + 3 * size_t(np));
    dt_.surface_density = upd(t->surface_density, 3 * np); dt_.pvtw = upd(t->pvtw, 5 * np);
    dt_.oil_node_ptr = upi(t->oil_node_ptr, np + 1);
    dt_.oil_rs = upd(t->oil_rs, non); dt_.oil_psat = upd(t->oil_psat, non);
    dt_.oil_invb_sat = upd(t->oil_invb_sat, non); dt_.oil_invbmu_sat = upd(t->oil_invbmu_sat, non);
    dt_.oil_col_ptr = upi(t->oil_col_ptr, non + 1);
    dt_.oil_col_p = upd(t->oil_col_p, noc); dt_.oil_col_invb = upd(t->oil_col_invb, noc); dt_.oil_col_invbmu = upd(t->oil_col_invbmu, noc);
    dt_.gas_node_ptr = upi(t->gas_node_ptr, np + 1);
    dt_.gas_pg = upd(t->gas_pg, ngn); dt_.gas_rvsat = upd(t->gas_rvsat, ngn);
    dt_.gas_invb_sat = upd(t->gas_invb_sat, ngn); dt_.gas_invbmu_sat = upd(t->gas_invbmu_sat, ngn);
    dt_.gas_col_ptr = upi(t->gas_col_ptr, ngn + 1);
    dt_.gas_col_rv = upd(t->gas_col_rv, ngc); dt_.gas_col_invb = upd(t->gas_col_invb, ngc); dt_.gas_col_invbmu = upd(t->gas_col_invbmu, ngc);
    dt_.swof_ptr = upi(t->swof_ptr, ns + 1);
    dt_.swof_sw = upd(t->swof_sw, nsw); dt_.swof_krw = upd(t->swof_krw, nsw); dt_.swof_krow = upd(t->swof_krow, nsw); dt_.swof_pcow = upd(t->swof_pcow, nsw);
    dt_.sgof_ptr = upi(t->sgof_ptr, ns + 1);
    dt_.sgof_sg = upd(t->sgof_sg, nsg); dt_.sgof_krg = upd(t->sgof_krg, nsg); dt_.sgof_krog = upd(t->sgof_krog, nsg); dt_.sgof_pcgo = upd(t->sgof_pcgo, nsg);
    if (t->rocktab_n > 0) {
        if (t->rocktab_n < 2) throw HipError(OPMGPU_EINVAL, "ROCKTAB needs at least two rows");
        dt_.rocktab_p = upd(t->rocktab_p, t->rocktab_n); dt_.rocktab_pvmult = upd(t->rocktab_pvmult, t->rocktab_n);
        dt_.rocktab_transmult = upd(t->rocktab_transmult, t->rocktab_n);
    }
    // slopes of every 1-D table (TabX): (y[i+1] - y[i]) / (x[i+1] - x[i]) per segment, segments never cross the tables of a CSR-style array
    auto slopes = [&](const double* x, const double* y, const int32_t* ptr, int ntab) -> const double* {
        const int n = ptr[ntab];
        std::vector<double> d(size_t(std::max(n, 1)), 0.0);
        for (int t_ = 0; t_ < ntab; ++t_)
            for (int i = ptr[t_]; i + 1 < ptr[t_ + 1]; ++i) d[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]);
        return upd(d.data(), size_t(n));
    };
    dx_.swof_dkrw = slopes(t->swof_sw, t->swof_krw, t->swof_ptr, ns); dx_.swof_dkrow = slopes(t->swof_sw, t->swof_krow, t->swof_ptr, ns);
    dx_.swof_dpcow = slopes(t->swof_sw, t->swof_pcow, t->swof_ptr, ns);
    dx_.sgof_dkrg = slopes(t->sgof_sg, t->sgof_krg, t->sgof_ptr, ns); dx_.sgof_dkrog = slopes(t->sgof_sg, t->sgof_krog, t->sgof_ptr, ns);
    dx_.sgof_dpcgo = slopes(t->sgof_sg, t->sgof_pcgo, t->sgof_ptr, ns);
    dx_.oil_drs = slopes(t->oil_psat, t->oil_rs, t->oil_node_ptr, np); dx_.oil_dinvb_sat = slopes(t->oil_psat, t->oil_invb_sat, t->oil_node_ptr, np);
    dx_.oil_dinvbmu_sat = slopes(t->oil_psat, t->oil_invbmu_sat, t->oil_node_ptr, np);
    dx_.oil_col_dinvb = slopes(t->oil_col_p, t->oil_col_invb, t->oil_col_ptr, non); dx_.oil_col_dinvbmu = slopes(t->oil_col_p, t->oil_col_invbmu, t->oil_col_ptr, non);
    dx_.gas_drvsat = slopes(t->gas_pg, t->gas_rvsat, t->gas_node_ptr, np); dx_.gas_dinvb_sat = slopes(t->gas_pg, t->gas_invb_sat, t->gas_node_ptr, np);
    dx_.gas_dinvbmu_sat = slopes(t->gas_pg, t->gas_invbmu_sat, t->gas_node_ptr, np);
    dx_.gas_col_dinvb = slopes(t->gas_col_rv, t->gas_col_invb, t->gas_col_ptr, ngn); dx_.gas_col_dinvbmu = slopes(t->gas_col_rv, t->gas_col_invbmu, t->gas_col_ptr, ngn);
    d_tab.upload(blob, stream);
    tab_words = int(blob.size());
    {
        // offset form for the kernels that resolve the tables against the blob's LDS copy (resolve_tables): pointer fields = word offsets
        dto_.t = dt_; dto_.x = dx_;
        auto offd = [&](const double*& p) { p = reinterpret_cast<const double*>(p ? reinterpret_cast<size_t>(p) - 1 : size_t(0)); };
        auto offi = [&](const int32_t*& p) { p = reinterpret_cast<const int32_t*>(p ? reinterpret_cast<size_t>(p) - 1 : size_t(0)); };
        opmgpu_tables& o = dto_.t;
        offd(o.surface_density); offd(o.pvtw);
        offi(o.oil_node_ptr); offd(o.oil_rs); offd(o.oil_psat); offd(o.oil_invb_sat); offd(o.oil_invbmu_sat);
        offi(o.oil_col_ptr); offd(o.oil_col_p); offd(o.oil_col_invb); offd(o.oil_col_invbmu);
        offi(o.gas_node_ptr); offd(o.gas_pg); offd(o.gas_rvsat); offd(o.gas_invb_sat); offd(o.gas_invbmu_sat);
        offi(o.gas_col_ptr); offd(o.gas_col_rv); offd(o.gas_col_invb); offd(o.gas_col_invbmu);
        offi(o.swof_ptr); offd(o.swof_sw); offd(o.swof_krw); offd(o.swof_krow); offd(o.swof_pcow);
        offi(o.sgof_ptr); offd(o.sgof_sg); offd(o.sgof_krg); offd(o.sgof_krog); offd(o.sgof_pcgo);
        if (t->rocktab_n > 0) { offd(o.rocktab_p); offd(o.rocktab_pvmult); offd(o.rocktab_transmult); }
        else { o.rocktab_p = nullptr; o.rocktab_pvmult = nullptr; o.rocktab_transmult = nullptr; }
        const double** xs[] = { &dto_.x.swof_dkrw, &dto_.x.swof_dkrow, &dto_.x.swof_dpcow, &dto_.x.sgof_dkrg, &dto_.x.sgof_dkrog, &dto_.x.sgof_dpcgo, &dto_.x.oil_drs,
                                &dto_.x.oil_dinvb_sat, &dto_.x.oil_dinvbmu_sat, &dto_.x.oil_col_dinvb, &dto_.x.oil_col_dinvbmu, &dto_.x.gas_drvsat, &dto_.x.gas_dinvb_sat,
                                &dto_.x.gas_dinvbmu_sat, &dto_.x.gas_col_dinvb, &dto_.x.gas_col_dinvbmu };
        const double** xg[] = { &dx_.swof_dkrw, &dx_.swof_dkrow, &dx_.swof_dpcow, &dx_.sgof_dkrg, &dx_.sgof_dkrog, &dx_.sgof_dpcgo, &dx_.oil_drs,
                                &dx_.oil_dinvb_sat, &dx_.oil_dinvbmu_sat, &dx_.oil_col_dinvb, &dx_.oil_col_dinvbmu, &dx_.gas_drvsat, &dx_.gas_dinvb_sat,
                                &dx_.gas_dinvbmu_sat, &dx_.gas_col_dinvb, &dx_.gas_col_dinvbmu };
        for (size_t k = 0; k < sizeof(xs) / sizeof(xs[0]); ++k) { offd(*xs[k]); *xg[k] = d_tab.p + (reinterpret_cast<size_t>(*xg[k]) - 1); }
    }
    {
        auto fixd = [&](const double*& p) { if (p) p = d_tab.p + (reinterpret_cast<size_t>(p) - 1); };
        auto fixi = [&](const int32_t*& p) { if (p) p = reinterpret_cast<const int32_t*>(d_tab.p + (reinterpret_cast<size_t>(p) - 1)); };
        fixd(dt_.surface_density); fixd(dt_.pvtw);
        fixi(dt_.oil_node_ptr); fixd(dt_.oil_rs); fixd(dt_.oil_psat); fixd(dt_.oil_invb_sat); fixd(dt_.oil_invbmu_sat);
        fixi(dt_.oil_col_ptr); fixd(dt_.oil_col_p); fixd(dt_.oil_col_invb); fixd(dt_.oil_col_invbmu);
        fixi(dt_.gas_node_ptr); fixd(dt_.gas_pg); fixd(dt_.gas_rvsat); fixd(dt_.gas_invb_sat); fixd(dt_.gas_invbmu_sat);
        fixi(dt_.gas_col_ptr); fixd(dt_.gas_col_rv); fixd(dt_.gas_col_invb); fixd(dt_.gas_col_invbmu);
        fixi(dt_.swof_ptr); fixd(dt_.swof_sw); fixd(dt_.swof_krw); fixd(dt_.swof_krow); fixd(dt_.swof_pcow);
        fixi(dt_.sgof_ptr); fixd(dt_.sgof_sg); fixd(dt_.sgof_krg); fixd(dt_.sgof_krog); fixd(dt_.sgof_pcgo);
        if (t->rocktab_n > 0) { fixd(dt_.rocktab_p); fixd(dt_.rocktab_pvmult); fixd(dt_.rocktab_transmult); }
        else { dt_.rocktab_p = nullptr; dt_.rocktab_pvmult = nullptr; dt_.rocktab_transmult = nullptr; }
    }
    // unscaled end points of every saturation region (what opm-material's EclEpsScalingPointsInfo::extractUnscaled reads off the
    // tables): Swl Swcr Swu Sowcr Sgl Sgcr Sgu Sogcr
    h_unscaled.assign(8 * size_t(ns), 0.0);
    for (int r = 0; r < ns; ++r) {
        const int a = t->swof_ptr[r], nw = t->swof_ptr[r + 1] - a, b = t->sgof_ptr[r], ng = t->sgof_ptr[r + 1] - b;
        auto last_zero = [](const double* x, const double* y, int n) { int i = 0; while (i + 1 < n && y[i + 1] == 0.0) ++i; return x[i]; };
        auto first_zero = [](const double* x, const double* y, int n) { int i = 0; while (i < n - 1 && y[i] != 0.0) ++i; return x[i]; };
        double* u = &h_unscaled[8 * size_t(r)];
        u[0] = t->swof_sw[a]; u[1] = last_zero(t->swof_sw + a, t->swof_krw + a, nw); u[2] = t->swof_sw[a + nw - 1];
        u[3] = 1.0 - first_zero(t->swof_sw + a, t->swof_krow + a, nw);
        u[4] = t->sgof_sg[b]; u[5] = last_zero(t->sgof_sg + b, t->sgof_krg + b, ng); u[6] = t->sgof_sg[b + ng - 1];
        u[7] = 1.0 - first_zero(t->sgof_sg + b, t->sgof_krog + b, ng);
    }
    // table maxima the vertical scaling refers to, per region in curve order: krw(Swu), krow(Swl), pcow(Swl), krg(Sgu), krog(Sgl), pcgo(Sgu)
    h_tabmax.assign(6 * size_t(ns), 0.0);
    for (int r = 0; r < ns; ++r) {
        const int a = t->swof_ptr[r], nw = t->swof_ptr[r + 1] - a, b = t->sgof_ptr[r], ng = t->sgof_ptr[r + 1] - b;
        double* m = &h_tabmax[6 * size_t(r)];
        m[0] = t->swof_krw[a + nw - 1]; m[1] = t->swof_krow[a]; m[2] = t->swof_pcow[a];
        m[3] = t->sgof_krg[b + ng - 1]; m[4] = t->sgof_krog[b]; m[5] = t->sgof_pcgo[b + ng - 1];
    }
    OPMGPU_HIP(hipStreamSynchronize(stream));
}

// pattern = stencil U well cliques -> solver plan -> internal numbering of every per-cell array
void BlackoilDevice::rebuild_structure()
{
    std::vector<int32_t> rowptr, col, code;
    const int nw = device_wells ? 0 : int(h_well_connpos.size()) - 1;     // device wells: factored operator, no clique fill
    const int st = build_reservoir_pattern(nc, nconn, h_conn.data(), nw, h_well_connpos.data(), h_well_cells.data(), rowptr, col, code);
    if (st != OPMGPU_OK) throw HipError(st, "invalid grid connections / wells (out of range or duplicate cell pair)");
    // keep the state across a re-plan (wells change between report steps)
    std::vector<double> sp, ssat, srs, srv; std::vector<int8_t> shc;
    std::vector<double> smax;
    if (d_somax.p) { smax.resize(nc); get_sat_oil_max(smax.data()); }
    if (has_state) { sp.resize(nc); ssat.resize(3 * size_t(nc)); srs.resize(nc); srv.resize(nc); shc.resize(nc); get_state(sp.data(), ssat.data(), srs.data(), srv.data(), shc.data()); }
    const int st2 = ls.set_pattern(nc, rowptr.data(), col.data(), prm.ilu_ordering);
    if (st2 != OPMGPU_OK) throw HipError(st2, "sparsity plan failed");
    const Plan& P = ls.plan;
    const int nbp = P.nbp;
    // per SELL entry: transmissibility with the row's side of the connection in the sign bit (NaN: well fill, 0: diagonal / padding),
    // g (z_c1 - z_c2) of the connection, its threshold pressure (k_assemble_rows reads them coalesced next to the column index)
    {
        const double nan = std::numeric_limits<double>::quiet_NaN();
        std::vector<double> te(P.nentries, 0.0), he;
        if (use_thpres) he.assign(P.nentries, 0.0);
        for (int b = 0; b < P.nnzb; ++b) {
            const int e = P.entry_of_block[b], c = code[b];
            if (c >= 0) {
                const int f = c >> 1;
                te[e] = std::copysign(h_trans[f], (c & 1) ? -1.0 : 1.0);
                if (use_thpres) he[e] = h_thpres[f];
            } else if (c == -2) te[e] = nan;
        }
        d_tr_e.upload(te, stream);
        if (use_thpres) d_thp_e.upload(he, stream);
        // cell depths in internal numbering: g (z_c1 - z_c2) of a connection is formed in the kernel (gravity * (z[c1] - z[c2]), the same
        // expression the reference's geometry evaluates once per face)
        std::vector<double> zi(P.nbp, 0.0);
        for (int r = 0; r < nc; ++r) zi[r] = h_z[P.nat[r]];
        d_zc.upload(zi, stream);
    }
    std::vector<double> pvi(nbp, 1.0); std::vector<int32_t> pn(nbp, 0), sn(nbp, 0);
    for (int r = 0; r < nc; ++r) { pvi[r] = h_pv[P.nat[r]]; pn[r] = h_pvtnum[P.nat[r]]; sn[r] = h_satnum[P.nat[r]]; }
    d_pv.upload(pvi, stream); d_pvtnum.upload(pn, stream); d_satnum.upload(sn, stream);
    if (use_eps) {
        // per region: unscaled fixed points of every curve (u0 | u1); per cell: the maps and the vertical factors (build_eps_planes)
        // A set of curves WITHOUT scaled end points keeps the saturation exactly (S_u = 0 + (S - 0) * 1: states that sit on a table
        // breakpoint must not move by an ulp), so its region table is zero and build_eps_planes writes s0 = 0, k = 1.
        const int ns = dt_.n_sat_regions;
        auto region_table = [&](bool have_points) {
            std::vector<double> ureg(size_t(kEpsRegion) * ns, 0.0);
            for (int r = 0; r < ns && have_points; ++r) {
                const double* u = &h_unscaled[8 * size_t(r)];
                double* o = &ureg[size_t(kEpsRegion) * r];
                o[EC_KRW] = u[1]; o[EC_KROW] = u[0] + u[4]; o[EC_PCOW] = u[0]; o[EC_KRG] = u[5]; o[EC_KROG] = u[7]; o[EC_PCGO] = u[4];
                double* m = o + EC_COUNT;   // middle points (SCALECRS): krw 1-Sowcr-Sgl, krow Swcr+Sgl, krg 1-Sogcr-Swl, krog 1-Sgcr-Swl
                m[EC_KRW] = 1.0 - u[3] - u[4]; m[EC_KROW] = u[1] + u[4]; m[EC_PCOW] = 0.0; m[EC_KRG] = 1.0 - u[7] - u[0]; m[EC_KROG] = 1.0 - u[5] - u[0]; m[EC_PCGO] = 0.0;
            }
            return ureg;
        };
        d_eps_u0.upload(region_table(has_endpoints), stream);
        std::vector<double> planes;
        build_eps_planes(h_eps, has_endpoints, false, planes);
        d_eps.upload(planes, stream);
        if (use_hyst) {
            d_ieps_u0.upload(region_table(has_iendpoints || has_endpoints), stream);
            build_eps_planes(has_iendpoints ? h_ieps : h_eps, has_iendpoints || has_endpoints, true, planes);
            d_ieps.upload(planes, stream);
            std::vector<int32_t> im(nbp, 0);
            for (int r = 0; r < nc; ++r) im[r] = h_imbnum[P.nat[r]];
            d_imbnum.upload(im, stream);
            std::vector<double> hist;
            if (d_hist.p && has_state) { hist.resize(4 * size_t(nc)); get_hysteresis(hist.data(), hist.data() + nc, hist.data() + 2 * size_t(nc), hist.data() + 3 * size_t(nc)); }
            std::vector<double> hp(4 * size_t(nbp), 0.0);
            for (int r = 0; r < nbp; ++r) { hp[r] = 2.0; hp[size_t(nbp) + r] = 2.0; }       // no history
            if (!hist.empty())
                for (int r = 0; r < nc; ++r) for (int k = 0; k < 4; ++k) hp[size_t(k) * nbp + r] = hist[size_t(k) * nc + P.nat[r]];
            d_hist.upload(hp, stream);
        }
    }
    std::vector<int32_t> pc(std::max<size_t>(h_well_cells.size(), 1), 0);
    for (size_t i = 0; i < h_well_cells.size(); ++i) pc[i] = P.pos[h_well_cells[i]];
    d_perf_cells.upload(pc, stream);
    nperf = int(h_well_cells.size());
    d_perf.alloc(std::max(nperf, 1) * size_t(OPMGPU_PERF_K));
    DevArray<double>* planes[] = { &d_p, &d_sw, &d_so, &d_sg, &d_rs, &d_rv };
    for (DevArray<double>* a : planes) { a->alloc(nbp); a->zero(stream); }
    d_hc.alloc(nbp); d_hc.zero(stream);
    d_vals.alloc(size_t(VP_COUNT) * nbp); d_vals.zero(stream);
    d_accum0.alloc(3 * size_t(nbp)); d_accum0.zero(stream);
    d_R.alloc(3 * size_t(nbp)); d_R.zero(stream);
    d_bpart.alloc(3 * size_t(grid_for(nc))); d_bpart.zero(stream);
    d_dx.alloc(3 * size_t(nbp)); d_dx.zero(stream);
    d_dx_old.alloc(3 * size_t(nbp)); d_dx_old.zero(stream);
    d_red.alloc(13 * size_t(kMaxRedBlocks) + kRedPart);
    OPMGPU_HIP(hipStreamSynchronize(stream));
    has_dx = false; has_saved = false;
    d_somax.alloc(nbp); d_somax.zero(stream);
    OPMGPU_HIP(hipStreamSynchronize(stream));
    if (!smax.empty()) set_sat_oil_max(smax.data());
    wells_rebind();
    if (has_state) set_state(sp.data(), ssat.data(), srs.data(), srv.data(), shc.data());
}

HystArgs BlackoilDevice::hyst_args() const
{
    HystArgs h;
    h.imbnum = use_hyst ? d_imbnum.p : nullptr; h.hist = d_hist.p; h.ieps = d_ieps.p; h.iureg = d_ieps_u0.p;
    return h;
}

// per-cell planes [s0 | k | s1 | k1 | v] x 6 curves (internal numbering) for the table of the cell's region `reg_of` against the end
// points ep8 (caller numbering; have_points = false: identity maps); imbibition planes use the IMBNUM region's table
void BlackoilDevice::build_eps_planes(const std::vector<double>* ep8, bool have_points, bool imbibition, std::vector<double>& ep) const
{
    const Plan& P = ls.plan;
    const int nbp = P.nbp;
    ep.assign(size_t(kEpsPlanes) * nbp, 0.0);
    for (int r = 0; r < nbp; ++r) {
        const int c = r < nc ? P.nat[r] : -1;
        // which table: the drainage region, or (planes built for the imbibition curves) the IMBNUM region
        const int reg = c < 0 ? 0 : (imbibition ? h_imbnum[c] : h_satnum[c]);
        const double* u = &h_unscaled[8 * size_t(reg)];
        double e[8];
        for (int k = 0; k < 8; ++k) e[k] = (c < 0 || !have_points) ? u[k] : ep8[k][c];
        const double SWL = e[0], SWCR = e[1], SWU = e[2], SOWCR = e[3], SGL = e[4], SGCR = e[5], SGU = e[6], SOGCR = e[7];
        const double s0[EC_COUNT] = { SWCR, SWL + SGL, SWL, SGCR, SOGCR, SGL };
        const double s1[EC_COUNT] = { 1.0 - SOWCR - SGL, SWCR + SGL, 0.0, 1.0 - SOGCR - SWL, 1.0 - SGCR - SWL, 0.0 };
        const double s2[EC_COUNT] = { SWU, 1.0 - SOWCR - SGL, SWU, SGU, 1.0 - SWL - SGL, SGU };
        const double u0c[EC_COUNT] = { u[1], u[0] + u[4], u[0], u[5], u[7], u[4] };
        const double u1c[EC_COUNT] = { 1.0 - u[3] - u[4], u[1] + u[4], 0.0, 1.0 - u[7] - u[0], 1.0 - u[5] - u[0], 0.0 };
        const double u2c[EC_COUNT] = { u[2], 1.0 - u[3] - u[4], u[2], u[6], 1.0 - u[0] - u[4], u[6] };
        for (int k = 0; k < EC_COUNT && !have_points; ++k) {          // identity map, exact
            ep[size_t(k) * nbp + r] = 0.0; ep[size_t(EC_COUNT + k) * nbp + r] = 1.0; ep[size_t(2 * EC_COUNT + k) * nbp + r] = 1e300; ep[size_t(3 * EC_COUNT + k) * nbp + r] = 1.0;
        }
        for (int k = 0; k < EC_COUNT && have_points; ++k) {
            if (c >= 0 && !(s2[k] > s0[k])) throw HipError(OPMGPU_EINVAL, "ENDSCALE end points of a cell are not increasing");
            const bool three = scalecrs && have_points && k != EC_PCOW && k != EC_PCGO;
            if (three && c >= 0 && !(s1[k] > s0[k] && s2[k] > s1[k])) throw HipError(OPMGPU_EINVAL, "SCALECRS: a cell's critical saturation is not between its end points");
            ep[size_t(k) * nbp + r] = s0[k];
            ep[size_t(EC_COUNT + k) * nbp + r] = three ? (u1c[k] - u0c[k]) / (s1[k] - s0[k]) : (u2c[k] - u0c[k]) / (s2[k] - s0[k]);
            ep[size_t(2 * EC_COUNT + k) * nbp + r] = three ? s1[k] : 1e300;
            ep[size_t(3 * EC_COUNT + k) * nbp + r] = three ? (u2c[k] - u1c[k]) / (s2[k] - s1[k]) : 1.0;
        }
        // vertical factors: cell maximum / table maximum (KRW at Swu, KRO at Swl resp. Sgl, KRG at Sgu, PCW at Swl, PCG at Sgu)
        double* v[EC_COUNT];
        for (int k = 0; k < EC_COUNT; ++k) { v[k] = &ep[size_t(4 * EC_COUNT + k) * nbp + r]; *v[k] = 1.0; }
        if (c >= 0) {
            const double* tm = &h_tabmax[6 * size_t(reg)];
            const int src[EC_COUNT] = { 0, 1, 3, 2, 1, 4 };     // KRW, KRO, PCW, KRG, KRO, PCG
            for (int k = 0; k < EC_COUNT; ++k) if (!h_eps_v[src[k]].empty() && tm[k] != 0.0) *v[k] = h_eps_v[src[k]][c] / tm[k];
        }
    }
}

// ---- hysteresis history -------------------------------------------------------------------------------------------------------
// smallest abscissa at which the monotone piecewise-linear table takes the value yv (flat pieces: the left end)
__device__ double table_inverse(const double* __restrict__ x, const double* __restrict__ y, int n, double yv, bool ascending)
{
    if (ascending) {
        if (yv <= y[0]) return x[0];
        for (int i = 0; i + 1 < n; ++i) if (y[i] < yv && yv <= y[i + 1]) return x[i] + (yv - y[i]) / (y[i + 1] - y[i]) * (x[i + 1] - x[i]);
        return x[n - 1];
    }
    if (yv >= y[0]) return x[0];
    for (int i = 0; i + 1 < n; ++i) if (y[i] > yv && yv >= y[i + 1]) return x[i] + (yv - y[i]) / (y[i + 1] - y[i]) * (x[i + 1] - x[i]);
    return x[n - 1];
}
// EclDefaultMaterial::updateHysteresis ("inconsistent" form: krnSw = 1 - So / 1 - Sg) + EclHysteresisTwoPhaseLawParams::update +
// updateDynamicParams_ (Carlson shift: delta = Sw_imb(krn_drain(mdc)) - mdc), one thread per cell
__global__ __launch_bounds__(kBlock) void k_hyst_update(int nb, int nbp, opmgpu_tables T, const int32_t* __restrict__ satnum, const int32_t* __restrict__ imbnum,
                                                        const double* __restrict__ so, const double* __restrict__ sg, const double* __restrict__ eps,
                                                        const double* __restrict__ ieps, const double* __restrict__ ureg, const double* __restrict__ iureg,
                                                        double* __restrict__ hist, int force)
{
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= nb) return;
    double mow = hist[c], mgo = hist[nbp + c];
    bool upd = force != 0;
    const double sgc = fmin(1.0, fmax(0.0, sg[c]));
    if (!force) {
        if (1.0 - so[c] < mow) { mow = 1.0 - so[c]; upd = true; }
        if (1.0 - sgc < mgo) { mgo = 1.0 - sgc; upd = true; }
    }
    if (!upd) return;
    const int sreg = satnum[c], ireg = imbnum[c];
    EpsD E, EI;
    eps_load(eps, ureg, nbp, c, sreg, E);
    eps_load(ieps, iureg, nbp, c, ireg, EI);
    double dow = 0.0, dgo = 0.0, f, df;
    if (mow < 2.0) {
        const int wa = T.swof_ptr[sreg], wi = T.swof_ptr[ireg];
        sat_curve<false>(T.swof_sw + wa, T.swof_krow + wa, T.swof_ptr[sreg + 1] - wa, mow, E, EC_KROW, f, df);
        const double ku = f / EI.v[EC_KROW];
        if (ku > 0.0) dow = eps_unmap(EI, EC_KROW, table_inverse(T.swof_sw + wi, T.swof_krow + wi, T.swof_ptr[ireg + 1] - wi, ku, false)) - mow;
    }
    if (mgo < 2.0) {
        const int ga = T.sgof_ptr[sreg], gi = T.sgof_ptr[ireg];
        const double sgm = 1.0 - mgo;
        sat_curve<true>(T.sgof_sg + ga, T.sgof_krg + ga, T.sgof_ptr[sreg + 1] - ga, sgm, E, EC_KRG, f, df);
        const double ku = f / EI.v[EC_KRG];
        if (ku > 0.0) dgo = sgm - eps_unmap(EI, EC_KRG, table_inverse(T.sgof_sg + gi, T.sgof_krg + gi, T.sgof_ptr[ireg + 1] - gi, ku, true));
    }
    hist[c] = mow; hist[nbp + c] = mgo; hist[2 * long(nbp) + c] = dow; hist[3 * long(nbp) + c] = dgo;
}

int BlackoilDevice::update_hysteresis()
{
    if (!use_hyst) return OPMGPU_OK;
    if (!has_state) return OPMGPU_EINVAL;
    hipLaunchKernelGGL(k_hyst_update, dim3(grid_for(nc)), dim3(kBlock), 0, stream, nc, ls.plan.nbp, dt_, d_satnum.p, d_imbnum.p, d_so.p, d_sg.p,
                       (const double*)d_eps.p, (const double*)d_ieps.p, (const double*)d_eps_u0.p, (const double*)d_ieps_u0.p, d_hist.p, 0);
    return OPMGPU_OK;
}
int BlackoilDevice::set_hysteresis(const double* mdc_ow, const double* mdc_go)
{
    if (!use_hyst || !mdc_ow || !mdc_go) return OPMGPU_EINVAL;
    const Plan& P = ls.plan;
    std::vector<double> hp(4 * size_t(P.nbp), 0.0);
    for (int r = 0; r < P.nbp; ++r) { hp[r] = r < nc ? mdc_ow[P.nat[r]] : 2.0; hp[size_t(P.nbp) + r] = r < nc ? mdc_go[P.nat[r]] : 2.0; }
    d_hist.upload(hp, stream);
    // the shifts follow from the history
    hipLaunchKernelGGL(k_hyst_update, dim3(grid_for(nc)), dim3(kBlock), 0, stream, nc, P.nbp, dt_, d_satnum.p, d_imbnum.p, d_so.p, d_sg.p,
                       (const double*)d_eps.p, (const double*)d_ieps.p, (const double*)d_eps_u0.p, (const double*)d_ieps_u0.p, d_hist.p, 1);
    OPMGPU_HIP(hipStreamSynchronize(stream));
    return OPMGPU_OK;
}
int BlackoilDevice::get_hysteresis(double* mdc_ow, double* mdc_go, double* d_ow, double* d_go)
{
    if (!use_hyst) return OPMGPU_EINVAL;
    const Plan& P = ls.plan;
    std::vector<double> hp(4 * size_t(P.nbp));
    d_hist.download(hp.data(), hp.size(), stream);
    OPMGPU_HIP(hipStreamSynchronize(stream));
    double* out[4] = { mdc_ow, mdc_go, d_ow, d_go };
    for (int k = 0; k < 4; ++k) if (out[k]) for (int r = 0; r < nc; ++r) out[k][P.nat[r]] = hp[size_t(k) * P.nbp + r];
    return OPMGPU_OK;
}

int BlackoilDevice::set_wells(int nw, const int32_t* connpos, const int32_t* cells)
{
    if (nw < 0 || (nw > 0 && (!connpos || !cells))) return OPMGPU_EINVAL;
    std::vector<int32_t> cp(1, 0), wc;
    if (nw > 0) { cp.assign(connpos, connpos + nw + 1); wc.assign(cells, cells + connpos[nw]); }
    if (cp == h_well_connpos && wc == h_well_cells) return OPMGPU_OK;
    h_well_connpos = cp; h_well_cells = wc;
    rebuild_structure();
    return OPMGPU_OK;
}

void BlackoilDevice::set_state(const double* p, const double* sat, const double* rs, const double* rv, const int8_t* hc)
{
    const Plan& P = ls.plan;
    const int nbp = P.nbp;
    hbuf.assign(6 * size_t(nbp), 0.0); hbuf8.assign(nbp, int8_t(OPMGPU_HC_GAS_AND_OIL));
    for (int r = 0; r < nc; ++r) {
        const int c = P.nat[r];
        hbuf[r] = p[c]; hbuf[nbp + r] = sat[3 * size_t(c)]; hbuf[2 * size_t(nbp) + r] = sat[3 * size_t(c) + 1];
        hbuf[3 * size_t(nbp) + r] = sat[3 * size_t(c) + 2]; hbuf[4 * size_t(nbp) + r] = rs[c]; hbuf[5 * size_t(nbp) + r] = rv[c];
        hbuf8[r] = hc[c];
    }
    DevArray<double>* planes[] = { &d_p, &d_sw, &d_so, &d_sg, &d_rs, &d_rv };
    for (int k = 0; k < 6; ++k) OPMGPU_HIP(hipMemcpyAsync(planes[k]->p, hbuf.data() + size_t(k) * nbp, nbp * sizeof(double), hipMemcpyHostToDevice, stream));
    OPMGPU_HIP(hipMemcpyAsync(d_hc.p, hbuf8.data(), nbp, hipMemcpyHostToDevice, stream));
    OPMGPU_HIP(hipStreamSynchronize(stream));
    has_state = true;
}

void BlackoilDevice::get_state(double* p, double* sat, double* rs, double* rv, int8_t* hc)
{
    const Plan& P = ls.plan;
    const int nbp = P.nbp;
    hbuf.resize(6 * size_t(nbp)); hbuf8.resize(nbp);
    DevArray<double>* planes[] = { &d_p, &d_sw, &d_so, &d_sg, &d_rs, &d_rv };
    for (int k = 0; k < 6; ++k) OPMGPU_HIP(hipMemcpyAsync(hbuf.data() + size_t(k) * nbp, planes[k]->p, nbp * sizeof(double), hipMemcpyDeviceToHost, stream));
    OPMGPU_HIP(hipMemcpyAsync(hbuf8.data(), d_hc.p, nbp, hipMemcpyDeviceToHost, stream));
    OPMGPU_HIP(hipStreamSynchronize(stream));
    for (int r = 0; r < nc; ++r) {
        const int c = P.nat[r];
        if (p) p[c] = hbuf[r];
        if (sat) { sat[3 * size_t(c)] = hbuf[nbp + r]; sat[3 * size_t(c) + 1] = hbuf[2 * size_t(nbp) + r]; sat[3 * size_t(c) + 2] = hbuf[3 * size_t(nbp) + r]; }
        if (rs) rs[c] = hbuf[4 * size_t(nbp) + r];
        if (rv) rv[c] = hbuf[5 * size_t(nbp) + r];
        if (hc) hc[c] = hbuf8[r];
    }
}

void BlackoilDevice::launch_cell_values()
{
    const Plan& P = ls.plan;
    auto kern = tab_lds_words() > 0 ? k_cell_values<true> : k_cell_values<false>;
    hipLaunchKernelGGL(kern, dim3(grid_for(nc)), dim3(kBlock), tab_lds_bytes(), stream, nc, P.nbp, dto_, d_pvtnum.p, d_satnum.p,
                       d_p.p, d_sw.p, d_sg.p, d_rs.p, d_rv.p, d_hc.p, eps_planes(), d_eps_u0.p, d_somax.p, d_vals.p, d_bpart.p,
                       ls.comm ? ls.comm->owner_mask() : (const int8_t*)nullptr, (const double*)d_tab.p, tab_lds_words(), hyst_args());
}
template <class MS> void BlackoilDevice::assemble_kernels(double dt, bool initial, MS* A, bool props_only)
{
    const Plan& P = ls.plan;
    const double* sc = prm.matbalscale;
    // CPR: k_assemble_rows also writes the weights of the pressure equation (the device well model redoes those of its perforated cells after
    // its diagonal contributions, wells_assemble; explicit host well cliques change off-diagonal blocks: the solver's own pass then)
    MS* wout = nullptr;
    if (prm.use_cpr && (nperf == 0 || device_wells) && ls.cpr_weight_mode == 0 && ls.emulate_ranks <= 1) { ls.ensure_work<MS>(); ls.work<MS>().cprw.alloc(3 * size_t(P.nbp)); wout = ls.work<MS>().cprw.p; }
    hipEvent_t kt_a = ls.kt.begin();
    launch_cell_values();
    ls.kt.end(KT_CELL_PROPS, kt_a);
    if (props_only) return;
    KtScope kts(ls.kt, KT_FLUX);
    // compiled for 3 waves per SIMD (168 VGPRs, 14 spilled; default) or for 2 (173 VGPRs, none; OPMGPU_ASM_WAVES=2).  Measured at 100^3
    // (profiles/r03_asm_waves_ab.log): 0.232 against 0.240 ms with a double Jacobian, 0.181 against 0.206 ms with a float one
    static const int waves = std::getenv("OPMGPU_ASM_WAVES") ? std::atoi(std::getenv("OPMGPU_ASM_WAVES")) : 3;
    const bool lds = tab_lds_words() > 0;
    // all ten neighbour values in one batch (default; measured 0.202 against 0.211 ms with a double Jacobian, profiles/r03_asm_batch_ab.log) or
    // the upwind-dependent five only when needed (OPMGPU_ASM_BATCH=0)
    static const bool batch = !(std::getenv("OPMGPU_ASM_BATCH") && std::atoi(std::getenv("OPMGPU_ASM_BATCH")) == 0);
    // mixed precision: the float copy of a DOUBLE Jacobian is written in the same pass (A/B: OPMGPU_MIXED_DUALWRITE=0 converts before the solve)
    static const bool dual = !(std::getenv("OPMGPU_MIXED_DUALWRITE") && std::atoi(std::getenv("OPMGPU_MIXED_DUALWRITE")) == 0);
    float* a32 = nullptr;
    if (dual && sizeof(MS) == 8 && prm.preconditioner_single && !props_only && ls.emulate_ranks <= 1 && !(prm.use_cpr && prm.cpr_reference_transform)) { a32 = ls.matrix_f(); dual_written = true; }
    auto kern = !lds ? k_assemble_rows<MS, 2, false, false, false>
                     : (waves == 3 ? (batch ? k_assemble_rows<MS, 3, true, true, false> : k_assemble_rows<MS, 3, true, false, false>)
                                   : (batch ? k_assemble_rows<MS, 2, true, true, false> : k_assemble_rows<MS, 2, true, false, false>));
    if (a32) kern = lds ? k_assemble_rows<MS, 2, true, true, true> : k_assemble_rows<MS, 2, false, false, true>;      // (the dual-write variant: 2 waves per SIMD, no spills)
    hipLaunchKernelGGL(kern, dim3(grid8_for(nc)), dim3(kBlock), tab_lds_bytes(), stream, xcd_mode(), nc, P.nbp, dto_, d_pvtnum.p, d_satnum.p, d_pv.p,
                       d_p.p, d_sw.p, d_sg.p, d_rs.p, d_rv.p, d_hc.p, 1.0 / dt, int(initial), sc[0], sc[1], sc[2],
                       ls.dp.slice_ptr.p, ls.dp.col.p, ls.dp.rowlen.p, ls.dp.nlower.p, ls.dp.tpos.p, d_tr_e.p, (const double*)d_zc.p, gravity, use_thpres ? d_thp_e.p : (const double*)nullptr,
                       eps_planes(), d_eps_u0.p, d_somax.p, (const double*)d_vals.p, d_accum0.p, ls.comm ? ls.comm->owner_mask() : (const int8_t*)nullptr, d_R.p, A, wout,
                       (const int32_t*)ls.dp.flux_perm.p, (const double*)d_tab.p, tab_lds_words(), hyst_args(), a32);
    ls.weights_from_assembly = wout != nullptr;
}

// The Jacobian is written in the precision of the coming solve (opmgpu_set_solve_precision): float saves the f64 -> f32 copy
// of the whole matrix and half of the assembly's write traffic.  The host-well path adds f64 blocks, so it keeps the double matrix.
void BlackoilDevice::assemble(double dt, bool initial)
{
    well_words_valid = false; well_red_valid = false;
    last_dt = dt;
    has_rhs_extra = false;
    if (initial) { d_dx_old.zero(stream); ls.new_step_hint = true; }       // first matrix of a time step: coarse AMG operators are rebuilt
    const bool host_wells = nperf > 0 && !device_wells;
    ls.corr_policy.external = false;           // a matrix of the model's own assembly: the correction-factor policy may score its time steps
    ls.float_copy_valid = false;               // (mixed precision: the float copy of the previous matrix is stale)
    dual_written = false;
    ls.coarse_single_ok = nperf == 0;          // no wells of either kind: the global constant is the near-null-space vector
    ls.matrix_is_float = assemble_single && !host_wells;
    if (prm.update_equations_scaling) {
        // updateEquationsScaling (BlackoilModelBase_impl.hpp:919-947; default off): every equation scaled by the mean 1/b of its phase in the
        // state being assembled.  The reference sets it at the end of assemble() and its linear solver scales this assembly's system with
        // it; here the scaling is applied while the Jacobian is written, so the mean is taken first: one extra property pass (it fills
        // d_bpart), the three sums, then the assembly proper with the new factors (wells and right-hand side read prm.matbalscale too).
        if (ls.matrix_is_float) assemble_kernels<float>(dt, initial, ls.matrix_f(), true);
        else assemble_kernels<double>(dt, initial, ls.matrix_d(), true);
        double B[3];
        average_b(B);
        for (int a = 0; a < 3; ++a) { prm.matbalscale[a] = B[a]; ls.lowrank.scale[a] = B[a]; }      // the wells' bordered pressure column is built from the same factors (k_cpr_border)
    }
    const bool forked = wells_prologue_async(initial);
    if (ls.matrix_is_float) assemble_kernels<float>(dt, initial, ls.matrix_f());
    else assemble_kernels<double>(dt, initial, ls.matrix_d());
    if (forked) OPMGPU_HIP(hipStreamWaitEvent(stream, ev_well[1], 0));
    {
        KtScope kts(ls.kt, KT_WELLS);
        wells_assemble(initial);
    }
    // the matrix is final (the host well model, if any, adds its blocks later: not then): start its ILU0 factorisation now, see LinSolver::factor_early
    // -- or, factor_early_mode 2, behind the kernels of the convergence check that follows (convergence()): the factorisation then has the device
    // to itself while the host reads the check's result back, instead of slowing the check's small kernels down
    ls.factor_early = 0;
    early_factor_pending = ls.factor_early_on && ls.factor_overlap && prm.use_cpr && !prm.cpr_reference_transform && ls.emulate_ranks <= 1 && !host_wells && prm.cpr_ilu_n == 0 && ls.fill_level == 0;
    if (early_factor_pending && ls.factor_early_mode == 1) start_early_factor();
}

void BlackoilDevice::start_early_factor()
{
    if (!early_factor_pending) return;
    early_factor_pending = false;
    ls.wb_relax = prm.cpr_relax * prm.cpr_stage2_relax;
    if (dual_written && !ls.matrix_is_float) {       // the float copy came with the assembly; the wells' diagonal contributions follow it
        if (device_wells && nperf > 0)
            hipLaunchKernelGGL(k_refresh_f32_diag, dim3(grid_for(nperf)), dim3(kBlock), 0, stream, nperf, (const int32_t*)d_perf_cells.p, ls.dp.slice_ptr.p, ls.dp.nlower.p,
                               (const double*)ls.matrix_d(), ls.matrix_f());
        ls.float_copy_valid = true;
    }
    if (ls.matrix_is_float) { ls.ensure_work<float>(); ls.factor_async<float>(); ls.factor_early = 4; }
    else if (prm.preconditioner_single) { ls.mixed_prepare(true); ls.factor_async<float>(); ls.factor_early = 4; }      // mixed precision: the float copy, then its factors
    else { ls.ensure_work<double>(); ls.factor_async<double>(); ls.factor_early = 8; }
}

double BlackoilDevice::time_assemble(int reps, int props_only)
{
    const double dt = last_dt > 0 ? last_dt : 86400.0;
    hipEvent_t e0, e1;
    OPMGPU_HIP(hipEventCreate(&e0)); OPMGPU_HIP(hipEventCreate(&e1));
    auto launch = [&]() {
        if (props_only) launch_cell_values();
        else assemble(dt, false);
    };
    launch();
    OPMGPU_HIP(hipEventRecord(e0, stream));
    for (int i = 0; i < reps; ++i) launch();
    OPMGPU_HIP(hipEventRecord(e1, stream));
    OPMGPU_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    OPMGPU_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return double(ms) / reps;
}

int BlackoilDevice::convergence(double dt, double* B3, double* CNV3, double* MB3, double* linf3, int* converged)
{
    const Plan& P = ls.plan;
    const int g = std::min(grid_for(nc), kMaxRedBlocks);
    const int8_t* mask = ls.comm ? ls.comm->owner_mask() : nullptr;
    hipEvent_t kt_a = ls.kt.begin();
    hipLaunchKernelGGL(k_conv_partial, dim3(g), dim3(kBlock), 0, stream, nc, P.nbp, d_R.p, (const double*)d_bpart.p, grid_for(nc), d_pv.p, mask, d_red.p + kRedPart);
    hipLaunchKernelGGL(k_conv_final, dim3(13), dim3(kBlock), 0, stream, g, d_red.p + kRedPart, d_red.p);
    ls.kt.end(KT_CONV, kt_a);
    // decomposed run with device wells somewhere: their convergence maxima (flux equations, control equation, error marks) ride on the same
    // max-all-reduce in the slots 13..18 -- well_convergence() then needs neither its own all-reduce nor a stream synchronisation
    const bool wells_ride = ls.comm && has_device_wells();
    if (wells_ride) well_conv_pack(d_red.p + 13);
    if (ls.comm) {
        // sums (7) and maxima (6 + the wells' 6) in ONE all-reduce: every rank places its numbers in its own slot of a zeroed
        // [ranks][nv] table, the table is sum-all-reduced (= gathered), and every rank reduces the columns itself in rank order --
        // the same bits everywhere, and one small-message latency (~13 us over RCCL) instead of two per Newton iteration
        const int nv = wells_ride ? 19 : 13, nr = ls.comm->num_ranks();
        d_gather.ensure(size_t(nr) * 19);
        hipLaunchKernelGGL(k_conv_gather, dim3(1), dim3(kBlock), 0, stream, 0, nv, nr, ls.comm->my_rank(), d_red.p, d_gather.p);
        ls.comm->allreduce_sum(d_gather.p, nr * nv, stream);
        hipLaunchKernelGGL(k_conv_gather, dim3(1), dim3(kBlock), 0, stream, 1, nv, nr, ls.comm->my_rank(), d_red.p, d_gather.p);
    }
    start_early_factor();          // (factor_early_mode 2: behind the check's kernels, beside the read-back below)
    // (polled host-mapped copy: no stream synchronisation); the device wells' residuals and error flags come along for well_convergence()
    const void *we = nullptr, *wf = nullptr; int nwe = 0;
    const bool with_wells = !ls.comm && well_words_sources(we, nwe, wf) && 26 + nwe + 1 <= LinSolver::kPubWords;
    const uint32_t* h = with_wells ? ls.fetch_words(d_red.p, 26, we, nwe, wf, 1) : ls.fetch_words(d_red.p, wells_ride ? 38 : 26);
    std::memcpy(h_red, h, (wells_ride ? 19 : 13) * sizeof(double));
    well_red_valid = wells_ride;
    well_words_valid = with_wells;
    if (with_wells) well_words.assign(h + 26, h + 26 + nwe + 1);
    bool conv = true; int status = OPMGPU_OK;
    const double ncg = ls.comm ? double(ls.comm->n_owned_global) : double(nc);
    const double pvs = ls.comm ? pvsum_global : pvsum;
    if (h_red[6] != 0.0) status = OPMGPU_ENUMERICAL;                                    // non-finite residual, :1562-1566
    for (int a = 0; a < 3; ++a) {
        const double B = h_red[a] / ncg;
        const double cnv = B * dt * h_red[7 + a], mb = std::fabs(B * h_red[3 + a]) * dt / pvs;
        if (B3) B3[a] = B; if (CNV3) CNV3[a] = cnv; if (MB3) MB3[a] = mb; if (linf3) linf3[a] = h_red[10 + a];
        conv = conv && (mb < prm.tolerance_mb) && (cnv < prm.tolerance_cnv);
        if (std::isnan(mb) || std::isnan(cnv)) status = OPMGPU_ENUMERICAL;              // :1828-1836
        if (mb > prm.max_residual_allowed || cnv > prm.max_residual_allowed) status = OPMGPU_ENUMERICAL;   // :1837-1845
    }
    if (converged) *converged = conv ? 1 : 0;
    return status;
}

// relativeChange (BlackoilModelBase_impl.hpp:1595-1631): partial sums of |p0-p|^2 + |s0-s|^2 and |p|^2 + |s|^2, owned rows
__global__ __launch_bounds__(kBlock) void k_relchange_partial(int nb, const double* __restrict__ p, const double* __restrict__ sw,
                                                              const double* __restrict__ so, const double* __restrict__ sg,
                                                              const double* __restrict__ p0, const double* __restrict__ sw0,
                                                              const double* __restrict__ so0, const double* __restrict__ sg0,
                                                              const int8_t* __restrict__ mask, double* __restrict__ part)
{
    __shared__ double sm[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double vals[2] = { 0.0, 0.0 };
    for (long i = blockIdx.x * long(kBlock) + threadIdx.x; i < nb; i += long(gridDim.x) * kBlock) {
        if (mask && !mask[i]) continue;
        const double a = p[i], b = sw[i], c = so[i], d = sg[i];
        const double da = p0[i] - a, db = sw0[i] - b, dc = so0[i] - c, dd = sg0[i] - d;
        vals[0] += da * da + (db * db + dc * dc + dd * dd);
        vals[1] += a * a + (b * b + c * c + d * d);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const double s = wave_sum(vals[q]);
        __syncthreads();
        if (lane == 0) sm[w] = s;
        __syncthreads();
        if (threadIdx.x == 0) part[long(q) * gridDim.x + blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    }
}
__global__ __launch_bounds__(kBlock) void k_sum2_final(int nblocks, const double* __restrict__ part, double* __restrict__ out)
{
    __shared__ double sm[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int q = 0; q < 2; ++q) {
        double v = 0.0;
        for (int i = threadIdx.x; i < nblocks; i += kBlock) v += part[long(q) * nblocks + i];
        const double s = wave_sum(v);
        __syncthreads();
        if (lane == 0) sm[w] = s;
        __syncthreads();
        if (threadIdx.x == 0) out[q] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    }
}

// device-side copy of the reservoir state: what AdaptiveTimeStepping keeps as last_state (AdaptiveTimeStepping_impl.hpp:211-212)
void BlackoilDevice::save_state()
{
    const size_t nbp = size_t(ls.plan.nbp);
    d_saved.alloc(6 * nbp); d_saved_hc.alloc(nbp);
    DevArray<double>* planes[] = { &d_p, &d_sw, &d_so, &d_sg, &d_rs, &d_rv };
    for (int k = 0; k < 6; ++k) OPMGPU_HIP(hipMemcpyAsync(d_saved.p + size_t(k) * nbp, planes[k]->p, nbp * sizeof(double), hipMemcpyDeviceToDevice, stream));
    OPMGPU_HIP(hipMemcpyAsync(d_saved_hc.p, d_hc.p, nbp, hipMemcpyDeviceToDevice, stream));
    wells_save();
    has_saved = true;
}
void BlackoilDevice::restore_state()
{
    const size_t nbp = size_t(ls.plan.nbp);
    DevArray<double>* planes[] = { &d_p, &d_sw, &d_so, &d_sg, &d_rs, &d_rv };
    for (int k = 0; k < 6; ++k) OPMGPU_HIP(hipMemcpyAsync(planes[k]->p, d_saved.p + size_t(k) * nbp, nbp * sizeof(double), hipMemcpyDeviceToDevice, stream));
    OPMGPU_HIP(hipMemcpyAsync(d_hc.p, d_saved_hc.p, nbp, hipMemcpyDeviceToDevice, stream));
    wells_restore();
}
double BlackoilDevice::relative_change()
{
    const size_t nbp = size_t(ls.plan.nbp);
    const int g = std::min(grid_for(nc), kMaxRedBlocks);
    const int8_t* mask = ls.comm ? ls.comm->owner_mask() : nullptr;
    hipLaunchKernelGGL(k_relchange_partial, dim3(g), dim3(kBlock), 0, stream, nc, d_p.p, d_sw.p, d_so.p, d_sg.p, d_saved.p, d_saved.p + nbp,
                       d_saved.p + 2 * nbp, d_saved.p + 3 * nbp, mask, d_red.p + kRedPart);
    hipLaunchKernelGGL(k_sum2_final, dim3(1), dim3(kBlock), 0, stream, g, d_red.p + kRedPart, d_red.p);
    if (ls.comm) ls.comm->allreduce_sum(d_red.p, 2, stream);
    OPMGPU_HIP(hipMemcpyAsync(h_red, d_red.p, 2 * sizeof(double), hipMemcpyDeviceToHost, stream));
    OPMGPU_HIP(hipStreamSynchronize(stream));
    return h_red[1] > 0.0 ? h_red[0] / h_red[1] : 0.0;
}

// multi-GPU: the communicator knows rank-local caller numbering; (re)derive internal rows + global sums
void BlackoilDevice::attach_comm(CommBase* c, int n_owned)
{
    ls.comm = c;
    n_owned_cells = n_owned;
    for (int32_t pc : h_well_cells) if (device_wells && pc >= n_owned) throw HipError(OPMGPU_EINVAL, "multi-GPU: a well perforates a ghost cell; keep every well on one rank");
    double loc[2] = { 0.0, double(n_owned) };
    for (int i = 0; i < n_owned; ++i) loc[0] += h_pv[i];
    OPMGPU_HIP(hipMemcpyAsync(d_red.p, loc, 2 * sizeof(double), hipMemcpyHostToDevice, stream));
    c->allreduce_sum(d_red.p, 2, stream);
    OPMGPU_HIP(hipMemcpyAsync(h_red, d_red.p, 2 * sizeof(double), hipMemcpyDeviceToHost, stream));
    OPMGPU_HIP(hipStreamSynchronize(stream));
    pvsum_global = h_red[0];
    c->n_owned_global = int(h_red[1] + 0.5);
}

void BlackoilDevice::perf_props_device()
{
    if (nperf == 0) return;
    hipLaunchKernelGGL(k_perf_props, dim3(grid_for(nperf)), dim3(kBlock), 0, stream, nperf, DevTables{ dt_, dx_ }, d_perf_cells.p, d_pvtnum.p, d_satnum.p,
                       d_p.p, d_sw.p, d_sg.p, d_rs.p, d_rv.p, d_hc.p, eps_planes(), d_eps_u0.p, d_somax.p, long(ls.plan.nbp), d_perf.p, hyst_args());
}

void BlackoilDevice::perf_pvt_device(const double* press_dev, double* out_dev, const int32_t* gate)
{
    if (nperf == 0) return;
    hipLaunchKernelGGL(k_perf_pvt, dim3(grid_for(nperf)), dim3(kBlock), 0, stream, nperf, dt_, d_perf_cells.p, d_pvtnum.p, d_so.p, d_rs.p, d_rv.p, d_hc.p,
                       d_somax.p, press_dev, gate, out_dev);
}

void BlackoilDevice::perf_pvt(const double* press, double* out)
{
    if (nperf == 0) return;
    DevArray<double> dp_, dout; dp_.upload(press, size_t(nperf), stream); dout.alloc(5 * size_t(nperf));
    perf_pvt_device(dp_.p, dout.p, nullptr);
    OPMGPU_HIP(hipMemcpyAsync(out, dout.p, 5 * size_t(nperf) * sizeof(double), hipMemcpyDeviceToHost, stream));
    OPMGPU_HIP(hipStreamSynchronize(stream));
}

// sum of 1/b per phase over the owned cells (-> B_avg of getWellConvergence) into out_dev[0..2] (sums; the caller divides by the cell count)
void BlackoilDevice::binv_sums_device(double* out13_dev, double* scratch_dev)
{
    const Plan& P = ls.plan;
    const int g = std::min(grid_for(nc), kMaxRedBlocks);
    const int8_t* mask = ls.comm ? ls.comm->owner_mask() : nullptr;
    hipLaunchKernelGGL(k_conv_partial, dim3(g), dim3(kBlock), 0, stream, nc, P.nbp, d_R.p, (const double*)d_bpart.p, grid_for(nc), d_pv.p, mask, scratch_dev);
    hipLaunchKernelGGL(k_conv_final, dim3(13), dim3(kBlock), 0, stream, g, scratch_dev, out13_dev);
    if (ls.comm) ls.comm->allreduce_sum(out13_dev, 3, stream);
}

void BlackoilDevice::average_b(double* B3)
{
    DevArray<double> out, scratch; out.alloc(16); scratch.alloc(13 * size_t(kMaxRedBlocks));
    binv_sums_device(out.p, scratch.p);
    double h[3];
    OPMGPU_HIP(hipMemcpyAsync(h, out.p, 3 * sizeof(double), hipMemcpyDeviceToHost, stream));
    OPMGPU_HIP(hipStreamSynchronize(stream));
    const double ncg = ls.comm ? double(ls.comm->n_owned_global) : double(nc);
    for (int a = 0; a < 3; ++a) B3[a] = h[a] / ncg;
}

void BlackoilDevice::perf_props(double* out)
{
    if (nperf == 0) return;
    hipLaunchKernelGGL(k_perf_props, dim3(grid_for(nperf)), dim3(kBlock), 0, stream, nperf, DevTables{ dt_, dx_ }, d_perf_cells.p, d_pvtnum.p, d_satnum.p,
                       d_p.p, d_sw.p, d_sg.p, d_rs.p, d_rv.p, d_hc.p, eps_planes(), d_eps_u0.p, d_somax.p, long(ls.plan.nbp), d_perf.p, hyst_args());
    OPMGPU_HIP(hipMemcpyAsync(out, d_perf.p, size_t(nperf) * OPMGPU_PERF_K * sizeof(double), hipMemcpyDeviceToHost, stream));
    OPMGPU_HIP(hipStreamSynchronize(stream));
}

int BlackoilDevice::add_well_terms(const double* resid_delta, int nblk, const int32_t* rc, const double* blocks)
{
    const Plan& P = ls.plan;
    const double* sc = prm.matbalscale;
    if (nperf > 0 && resid_delta) {
        DevArray<double> dd; dd.upload(resid_delta, 3 * size_t(nperf), stream);
        hipLaunchKernelGGL(k_add_well_resid, dim3(grid_for(nperf)), dim3(kBlock), 0, stream, nperf, P.nbp, d_perf_cells.p, dd.p, d_R.p);
        OPMGPU_HIP(hipStreamSynchronize(stream));
    }
    if (nblk > 0) {
        std::vector<int32_t> ent(nblk);
        for (int k = 0; k < nblk; ++k) {
            const int r = rc[2 * k], c = rc[2 * k + 1];
            if (r < 0 || r >= nc) return OPMGPU_EINVAL;
            const int32_t* b = P.col.data() + P.rowptr[r]; const int32_t* e = P.col.data() + P.rowptr[r + 1];
            const int32_t* it = std::lower_bound(b, e, c);
            if (it == e || *it != c) return OPMGPU_EINVAL;
            ent[k] = P.entry_of_block[it - P.col.data()];
        }
        DevArray<int32_t> de; de.upload(ent, stream);
        DevArray<double> db; db.upload(blocks, 9 * size_t(nblk), stream);
        hipLaunchKernelGGL(k_add_well_blocks, dim3(grid_for(nblk)), dim3(kBlock), 0, stream, nblk, de.p, db.p, sc[0], sc[1], sc[2], ls.matrix_d());
        OPMGPU_HIP(hipStreamSynchronize(stream));
    }
    return OPMGPU_OK;
}

__global__ __launch_bounds__(kBlock) void k_f2d(long n, const float* __restrict__ a, double* __restrict__ b)
{
    for (long i = blockIdx.x * long(kBlock) + threadIdx.x; i < n; i += long(gridDim.x) * kBlock) b[i] = double(a[i]);
}
void launch_convert_f2d(long n, const float* a, double* b, hipStream_t s)
{
    hipLaunchKernelGGL(k_f2d, dim3(std::min(grid_for(n), kMaxRedBlocks)), dim3(kBlock), 0, s, n, a, b);
}

void BlackoilDevice::add_well_rhs(const double* rhs_delta)
{
    if (nperf == 0) return;
    const Plan& P = ls.plan;
    d_rhs_extra.alloc(3 * size_t(P.nbp));
    d_rhs_extra.zero(stream);
    DevArray<double> dd; dd.upload(rhs_delta, 3 * size_t(nperf), stream);
    hipLaunchKernelGGL(k_add_well_resid, dim3(grid_for(nperf)), dim3(kBlock), 0, stream, nperf, P.nbp, d_perf_cells.p, dd.p, d_rhs_extra.p);
    OPMGPU_HIP(hipStreamSynchronize(stream));
    has_rhs_extra = true;
}

void BlackoilDevice::perf_dx(double* out)
{
    if (nperf == 0) return;
    hipLaunchKernelGGL(k_gather_perf3, dim3(grid_for(nperf)), dim3(kBlock), 0, stream, nperf, ls.plan.nbp, d_perf_cells.p, d_dx.p, d_perf.p);
    OPMGPU_HIP(hipMemcpyAsync(out, d_perf.p, size_t(nperf) * 3 * sizeof(double), hipMemcpyDeviceToHost, stream));
    OPMGPU_HIP(hipStreamSynchronize(stream));
}

template <class S> void BlackoilDevice::build_rhs()
{
    const Plan& P = ls.plan;
    const double* sc = prm.matbalscale;
    hipLaunchKernelGGL((k_build_rhs<S>), dim3(grid_for(nc)), dim3(kBlock), 0, stream, nc, P.nbp, sc[0], sc[1], sc[2], d_R.p,
                       has_rhs_extra ? d_rhs_extra.p : (const double*)nullptr, ls.work<S>().b.p);
}
template <class S> void BlackoilDevice::store_dx()
{
    const Plan& P = ls.plan;
    const long n = 3 * long(P.nbp);
    if (sizeof(S) == 8) OPMGPU_HIP(hipMemcpyAsync(d_dx.p, ls.work<S>().x.p, n * sizeof(double), hipMemcpyDeviceToDevice, stream));
    else {
        launch_convert_f2d(n, reinterpret_cast<const float*>(ls.work<S>().x.p), d_dx.p, stream);
    }
    has_dx = true;
}
template void BlackoilDevice::build_rhs<float>();
template void BlackoilDevice::build_rhs<double>();
template void BlackoilDevice::store_dx<float>();
template void BlackoilDevice::store_dx<double>();

void BlackoilDevice::dx_to_host(double* dx)
{
    ls.vec_to_host<double>(d_dx.p, VEC_EQUATION_MAJOR, dx);
}

void BlackoilDevice::update_state(const double* dx_host, double relax)
{
    const Plan& P = ls.plan;
    if (dx_host) { ls.vec_from_host<double>(dx_host, VEC_EQUATION_MAJOR, d_dx.p); OPMGPU_HIP(hipStreamSynchronize(stream)); has_dx = true; }   // caller's buffer: done with it on return
    KtScope kts(ls.kt, KT_UPDATE_STATE);
    if (device_wells) wells_update(relax, dx_host != nullptr);          // updateWellState from the recovered (and possibly relaxed) well increment
    hipLaunchKernelGGL(k_update_state, dim3(grid_for(nc)), dim3(kBlock), tab_lds_bytes(), stream, nc, P.nbp, dt_, d_pvtnum.p, d_satnum.p, d_dx.p, relax,
                       prm.dp_max_rel, prm.ds_max, prm.dr_max_rel, d_p.p, d_sw.p, d_so.p, d_sg.p, d_rs.p, d_rv.p, d_hc.p, eps_planes(), d_eps_u0.p, d_somax.p,
                       (const double*)d_tab.p, tab_lds_words());
}

__global__ __launch_bounds__(kBlock) void k_somax_update(int nb, const double* __restrict__ so, double* __restrict__ somax)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < nb) somax[i] = fmax(somax[i], so[i]);
}

void BlackoilDevice::update_sat_oil_max()
{
    hipLaunchKernelGGL(k_somax_update, dim3(grid_for(nc)), dim3(kBlock), 0, stream, nc, d_so.p, d_somax.p);
}
void BlackoilDevice::set_sat_oil_max(const double* v)
{
    const Plan& P = ls.plan;
    std::vector<double> h(P.nbp, 0.0);
    for (int r = 0; r < nc; ++r) h[r] = v[P.nat[r]];
    d_somax.upload(h, stream);
    OPMGPU_HIP(hipStreamSynchronize(stream));
}
void BlackoilDevice::get_sat_oil_max(double* v)
{
    const Plan& P = ls.plan;
    std::vector<double> h(P.nbp);
    OPMGPU_HIP(hipMemcpyAsync(h.data(), d_somax.p, size_t(P.nbp) * sizeof(double), hipMemcpyDeviceToHost, stream));
    OPMGPU_HIP(hipStreamSynchronize(stream));
    for (int r = 0; r < nc; ++r) v[P.nat[r]] = h[r];
}

void BlackoilDevice::stabilize_update(int relax_type, double omega)
{
    // dampening by 1 changes nothing, and the previous increment (dx_old) is only ever read by the SOR form: no launch then.  (A run
    // uses one relaxation type throughout -- NonlinearSolver's parameter --, so SOR never meets a dx_old that was skipped here.)
    if (relax_type != OPMGPU_RELAX_SOR && omega == 1.0) return;
    const long n = 3 * long(ls.plan.nbp);
    if (device_wells) wells_stabilize(relax_type == OPMGPU_RELAX_SOR ? 1 : 0, omega);      // the well part first: it is recovered from the UNRELAXED dx
    hipLaunchKernelGGL(k_stabilize, dim3(unsigned((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, n, relax_type == OPMGPU_RELAX_SOR ? 1 : 0, omega,
                       d_dx.p, d_dx_old.p);
}

// computeFluidInPlace (BlackoilModelBase_impl.hpp:2263-2445, serial and parallel branch): per-cell volumes on the device, the region sums on the host
// in cell order like the reference's loops (an output path: once per sub-step / report step).  fipnum: region per cell in the caller's
// order, 0 = in no region, nullptr = one region of all cells; values: [dims][7]; fip_cells (optional): [7][nc] like SimulatorData::fip.
void BlackoilDevice::fluid_in_place(const int32_t* fipnum, int dims, double* fip_cells, double* values)
{
    const Plan& P = ls.plan;
    DevArray<double> dout; dout.alloc(size_t(8) * nc);
    hipLaunchKernelGGL(k_fip_cells, dim3(grid_for(nc)), dim3(kBlock), 0, stream, nc, DevTables{ dt_, dx_ }, ls.dp.nat.p, d_pvtnum.p, d_satnum.p, d_pv.p,
                       d_p.p, d_sw.p, d_so.p, d_sg.p, d_rs.p, d_rv.p, d_hc.p, eps_planes(), d_eps_u0.p, d_somax.p, long(P.nbp), dout.p, hyst_args());
    std::vector<double> h(size_t(8) * nc);
    dout.download(h.data(), h.size(), stream);
    OPMGPU_HIP(hipStreamSynchronize(stream));
    const double *pvol = h.data() + size_t(5) * nc, *pres_c = h.data() + size_t(6) * nc, *hyd = h.data() + size_t(7) * nc;
    // decomposed runs (the reference's parallel branch, :2369-2446): every rank sums over the cells it OWNS (caller cells [0, n_owned)),
    // the regions' hydrocarbon pore volumes / pressure sums and finally the region values are summed over the ranks; `dims` must be the
    // GLOBAL number of regions on every rank (the reference takes comm.max of the local maxima; here the buffer is the caller's)
    const int n_own = ls.comm ? n_owned_cells : nc;
    DevArray<double> dsum;
    auto allsum = [&](double* v, int n) {
        if (!ls.comm || n <= 0) return;
        dsum.upload(v, size_t(n), stream);
        ls.comm->allreduce_sum(dsum.p, n, stream);
        dsum.download(v, size_t(n), stream);
        OPMGPU_HIP(hipStreamSynchronize(stream));
    };
    if (ls.comm) {
        double dmax = double(dims);
        dsum.upload(&dmax, 1, stream);
        ls.comm->allreduce_max(dsum.p, 1, stream);
        dsum.download(&dmax, 1, stream);
        OPMGPU_HIP(hipStreamSynchronize(stream));
        if (int(dmax + 0.5) != dims) throw HipError(OPMGPU_EINVAL, "computeFluidInPlace: pass the GLOBAL number of regions on every rank of a decomposed run");
    }
    for (int i = 0; i < dims * 7; ++i) values[i] = 0.0;
    auto region = [&](int c) { return fipnum ? fipnum[c] - 1 : 0; };
    for (int ph = 0; ph < 5; ++ph)                       // phases, then the rs / rv volumes (:2312-2332)
        for (int c = 0; c < n_own; ++c) { const int r = region(c); if (r != -1) values[r * 7 + ph] += h[size_t(ph) * nc + c]; }
    std::vector<double> hp(2 * size_t(dims), 0.0);       // hydrocarbon pore volume | pv-weighted pressure sum, per region
    double* hcpv = hp.data(); double* pres = hp.data() + dims;
    for (int c = 0; c < n_own; ++c) { const int r = region(c); if (r != -1) { hcpv[r] += pvol[c] * hyd[c]; pres[r] += pvol[c] * pres_c[c]; } }
    allsum(hp.data(), 2 * dims);                         // comm.sum(hcpv), comm.sum(pres)
    std::vector<double> fpv(nc, 0.0), fwp(nc, 0.0);
    for (int c = 0; c < n_own; ++c) {
        const int r = region(c);
        if (r == -1) continue;
        fpv[c] = pvol[c];
        // hydrocarbon-pore-volume weighted average pressure; a region without hydrocarbons: as the reference writes it (:2356-2360)
        if (hcpv[r] != 0) fwp[c] = pvol[c] * pres_c[c] * hyd[c] / hcpv[r];
        else fwp[c] = pres[r] / pvol[c];
        values[r * 7 + 5] += fpv[c];
        values[r * 7 + 6] += fwp[c];
    }
    allsum(values, dims * 7);                            // one sum for all regions (the reference loops comm.sum over the regions)
    if (fip_cells) {
        std::copy(h.begin(), h.begin() + size_t(5) * nc, fip_cells);
        std::copy(fpv.begin(), fpv.end(), fip_cells + size_t(5) * nc);
        std::copy(fwp.begin(), fwp.end(), fip_cells + size_t(6) * nc);
    }
}

// RateConverter::SurfaceToReservoirVoidage::calcAverages (RateConverterLegacy.hpp:718-768), the sums only: per region sum of p, rs, rv over the
// cells and their number, in cell order like the reference's loop; decomposed runs: over the OWNED cells, then summed over the ranks (its
// is_parallel branch).  region: per cell in the caller's order, values 0 .. nregions-1 (nullptr = one region: what SimulatorBase builds,
// SimulatorBase_impl.hpp:66).  sums: [nregions][4] = sum p, sum rs, sum rv, n.  An output-cadence path (once per report step).
void BlackoilDevice::region_state_sums(const int32_t* region, int nregions, double* sums)
{
    std::vector<double> p(nc), sat(3 * size_t(nc)), rs(nc), rv(nc);
    std::vector<int8_t> hc(nc);
    get_state(p.data(), sat.data(), rs.data(), rv.data(), hc.data());
    const int n_own = ls.comm ? n_owned_cells : nc;
    for (int i = 0; i < 4 * nregions; ++i) sums[i] = 0.0;
    for (int c = 0; c < n_own; ++c) {
        const int r = region ? region[c] : 0;
        if (r < 0 || r >= nregions) throw HipError(OPMGPU_EINVAL, "region_state_sums: region index out of range");
        sums[4 * r + 0] += p[c]; sums[4 * r + 1] += rs[c]; sums[4 * r + 2] += rv[c]; sums[4 * r + 3] += 1.0;
    }
    if (ls.comm) {
        DevArray<double> d;
        d.upload(sums, size_t(4) * nregions, stream);
        ls.comm->allreduce_sum(d.p, 4 * nregions, stream);
        d.download(sums, size_t(4) * nregions, stream);
        OPMGPU_HIP(hipStreamSynchronize(stream));
    }
}

void BlackoilDevice::voidage_coefficients(int n, const double* p, const double* rs, const double* rv, const int32_t* pvtreg, double* coeff)
{
    if (n <= 0) return;
    if (pvtreg) for (int i = 0; i < n; ++i) if (pvtreg[i] < 0 || pvtreg[i] >= dt_.n_pvt_regions) throw HipError(OPMGPU_EINVAL, "voidage_coefficients: PVT region out of range");
    DevArray<double> din, dout; DevArray<int32_t> dreg;
    std::vector<double> h(3 * size_t(n));
    std::copy(p, p + n, h.begin()); std::copy(rs, rs + n, h.begin() + n); std::copy(rv, rv + n, h.begin() + 2 * size_t(n));
    din.upload(h.data(), h.size(), stream);
    if (pvtreg) dreg.upload(pvtreg, size_t(n), stream);
    dout.alloc(3 * size_t(n));
    hipLaunchKernelGGL(k_voidage_coeff, dim3(grid_for(n)), dim3(kBlock), 0, stream, n, dt_, din.p, din.p + n, din.p + 2 * size_t(n),
                       pvtreg ? (const int32_t*)dreg.p : (const int32_t*)nullptr, dout.p);
    dout.download(coeff, 3 * size_t(n), stream);
    OPMGPU_HIP(hipStreamSynchronize(stream));
}

void BlackoilDevice::get_residual(double* r)
{
    ls.vec_to_host<double>(d_R.p, VEC_EQUATION_MAJOR, r);
}

} // namespace opmgpu
