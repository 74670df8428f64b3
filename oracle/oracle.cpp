/*
 * oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * Plain C++ restatement of the flow_legacy Newton step.  Derivatives are carried by a tiny
 * dense forward-AD type (Dual<N>) so that the Jacobian falls out of the same formulas the
 * reference writes with AutoDiffBlock (AutoDiffBlock.hpp:327-404): this is deliberately a
 * DIFFERENT derivation from the hand-written chain rule in the HIP kernels.
 *
 * Third-party arithmetic restated from the published algorithms (not in the reference tree):
 *   opm-material  PiecewiseLinearTwoPhaseMaterial, EclDefaultMaterial, Tabulated1DFunction,
 *                 UniformXTabulated2DFunction, LiveOilPvt, WetGasPvt,
 *                 ConstantCompressibilityWaterPvt          (call sites cited per function)
 *   dune-istl     bilu0_decomposition, BiCGSTABSolver::apply
 *   opm-simulators ParallelOverlappingILU0::apply, MatrixBlock 3x3 inverse
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

int g_threads = 1;

// ----------------------------------------------------------------------------------------
// forward AD
// ----------------------------------------------------------------------------------------
template <int N>
struct Dual {
    double v;
    double d[N];
    Dual() : v(0.0) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
    Dual(double x) : v(x) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
    static Dual var(double x, int k) { Dual r(x); r.d[k] = 1.0; return r; }
};
template <int N> Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> Dual<N> operator-(const Dual<N>& a) { Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v / b.v; for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v; return r; }
template <int N> Dual<N> operator*(double a, const Dual<N>& b) { Dual<N> r; r.v = a * b.v; for (int i = 0; i < N; ++i) r.d[i] = a * b.d[i]; return r; }
template <int N> Dual<N> operator*(const Dual<N>& b, double a) { return a * b; }
template <int N> Dual<N> operator+(const Dual<N>& a, double b) { Dual<N> r = a; r.v += b; return r; }
template <int N> Dual<N> operator+(double b, const Dual<N>& a) { return a + b; }
template <int N> Dual<N> operator-(const Dual<N>& a, double b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> Dual<N> operator-(double b, const Dual<N>& a) { return (-a) + b; }
template <int N> Dual<N> operator/(const Dual<N>& a, double b) { return a * (1.0 / b); }
template <int N> Dual<N> operator/(double a, const Dual<N>& b) { return Dual<N>(a) / b; }
// y = f(x) with known f(x.v) and f'(x.v)
template <int N> Dual<N> chain(double f, double df, const Dual<N>& x) { Dual<N> r; r.v = f; for (int i = 0; i < N; ++i) r.d[i] = df * x.d[i]; return r; }
template <int N> Dual<N> chain2(double f, double dfx, const Dual<N>& x, double dfy, const Dual<N>& y) { Dual<N> r; r.v = f; for (int i = 0; i < N; ++i) r.d[i] = dfx * x.d[i] + dfy * y.d[i]; return r; }

typedef Dual<3> D3;
typedef Dual<6> D6;

// ----------------------------------------------------------------------------------------
// saturation-function tables: opm-material PiecewiseLinearTwoPhaseMaterial::eval_
// (constant extrapolation; bisection with strict '<' picks the LEFT segment at a breakpoint of
// the law's own abscissa).  SWOF columns are tabulated against Sw (ascending), so "left" is left
// in Sw.  SGOF columns are stored by opm-material against So = 1 - Swco - Sg (reversed to
// ascending), so left-in-So is RIGHT-in-Sg: sat_right() below.  Pinned by
// tests/test_satfunc.cpp:93-108 (sw = 0.4 breakpoint, end points).
// ----------------------------------------------------------------------------------------
void sat_left(const double* x, const double* y, int n, double xv, double& f, double& df)
{
    if (xv <= x[0]) { f = y[0]; df = 0.0; return; }
    if (xv >= x[n - 1]) { f = y[n - 1]; df = 0.0; return; }
    int lo = 0, hi = n - 1;
    while (lo + 1 < hi) { const int mid = (lo + hi) / 2; if (x[mid] < xv) lo = mid; else hi = mid; }
    df = (y[lo + 1] - y[lo]) / (x[lo + 1] - x[lo]);
    f = y[lo] + df * (xv - x[lo]);
}
void sat_right(const double* x, const double* y, int n, double xv, double& f, double& df)
{
    if (xv <= x[0]) { f = y[0]; df = 0.0; return; }
    if (xv >= x[n - 1]) { f = y[n - 1]; df = 0.0; return; }
    int lo = 0, hi = n - 1;
    while (lo + 1 < hi) { const int mid = (lo + hi) / 2; if (x[mid] <= xv) lo = mid; else hi = mid; }
    df = (y[lo + 1] - y[lo]) / (x[lo + 1] - x[lo]);
    f = y[lo] + df * (xv - x[lo]);
}

struct SatTab {
    int nw, ng;
    const double *sw, *krw, *krow, *pcow, *sg, *krg, *krog, *pcgo;
    double swco;
};
SatTab sat_tab(const opmgpu_tables* t, int reg)
{
    SatTab s;
    const int a = t->swof_ptr[reg], b = t->sgof_ptr[reg];
    s.nw = t->swof_ptr[reg + 1] - a; s.ng = t->sgof_ptr[reg + 1] - b;
    s.sw = t->swof_sw + a; s.krw = t->swof_krw + a; s.krow = t->swof_krow + a; s.pcow = t->swof_pcow + a;
    s.sg = t->sgof_sg + b; s.krg = t->sgof_krg + b; s.krog = t->sgof_krog + b; s.pcgo = t->sgof_pcgo + b;
    s.swco = s.sw[0];
    return s;
}

// ENDSCALE (opm-material EclEpsTwoPhaseLaw / EclEpsScalingPoints).  Horizontal scaling maps the scaled saturation S to the table's
// unscaled one: two-point (SCALECRS NO)  S_u = u0 + (S - s0) * ((u2 - u0) / (s2 - s0));  three-point (SCALECRS YES, relative
// permeabilities only) the same through a middle point (s1, u1): first segment below s1, second above (scaledToUnscaledSatThreePoint_).
// Unscaled points of a table (EclEpsScalingPointsInfo::extractUnscaled): Swl = first Sw, Swu = last Sw, Swcr = last Sw with
// krw == 0, Sowcr = 1 - (first Sw with krow == 0); Sgl, Sgu, Sgcr alike, Sogcr = 1 - (first Sg with krog == 0)  [no Swl
// subtracted: pinned by tests/test_boprops_ad.cpp:197-208 (0.13) and the EPS_A derivatives of tests/test_satfunc.cpp:299-307].
// Points per curve (scaled <-> unscaled): krw [Swcr, (1-Sowcr-Sgl), Swu]; krow in Sw [Swl+Sgl, (Swcr+Sgl), 1-Sowcr-Sgl]; pcow [Swl, Swu];
// krg [Sgcr, (1-Sogcr-Swl), Sgu]; krog in oil saturation [Sogcr, (1-Sgcr-Swl), 1-Swl-Sgl]; pcgo [Sgl, Sgu]; the bracketed middle points
// only with SCALECRS.  The two-point forms are pinned by tests/test_satfunc.cpp:140-379; the three-point middle points, the vertical
// scaling and the hysteresis below are restated from opm-material's published code with no reference vectors: parity unpinned.
// Vertical scaling (KRW / KRO / KRG / PCW / PCG): value * (cell maximum / table maximum), unscaledToScaledKrw_ etc.
struct Lin { double u0, s0, k, s1, u1, k1; bool three; };
inline double lin_map(const Lin& m, double s) { return (m.three && s >= m.s1) ? m.u1 + (s - m.s1) * m.k1 : m.u0 + (s - m.s0) * m.k; }
inline double lin_slope(const Lin& m, double s) { return (m.three && s >= m.s1) ? m.k1 : m.k; }
// scaled saturation of an unscaled one (inverse map; unscaledToScaledSat*)
inline double lin_inv(const Lin& m, double u) { return (m.three && u >= m.u1) ? m.s1 + (u - m.u1) / m.k1 : m.s0 + (u - m.u0) / m.k; }
struct Eps {
    bool on;
    Lin krw, krow, pcow, krg, krog, pcgo;
    double v[6];       // vertical factors of krw, krow, pcow, krg, krog, pcgo (1 = none)
    double swl;        // scaled connate water used by the three-phase law (params.Swl())
    double swl_t;      // unscaled Swco of the table (the SGOF oil-saturation abscissa is 1 - swl_t - Sg)
};
enum { V_KRW = 0, V_KROW, V_PCOW, V_KRG, V_KROG, V_PCGO };
// end points `ep8` (SWL SWCR SWU SOWCR SGL SGCR SGU SOGCR of cell c) against the table `s`; vertical maxima from g->eps_v
Eps eps_build(const opmgpu_grid* g, const SatTab& s, int c, const double* const* ep8)
{
    Eps e; e.on = ep8 && ep8[0] != nullptr; e.swl = s.swco; e.swl_t = s.swco;
    for (int k = 0; k < 6; ++k) e.v[k] = 1.0;
    if (g) {
        const double* const* ev = g->eps_v;
        auto last = [](const double* y, int n) { return y[n - 1]; };
        if (ev[0] && last(s.krw, s.nw) != 0.0) e.v[V_KRW] = ev[0][c] / last(s.krw, s.nw);              // maxKrw: at Swu
        if (ev[1] && s.krow[0] != 0.0) e.v[V_KROW] = ev[1][c] / s.krow[0];                             // maxKrn (oil): at Swl
        if (ev[1] && s.krog[0] != 0.0) e.v[V_KROG] = ev[1][c] / s.krog[0];                             // gas-oil system: oil at Sgl
        if (ev[2] && last(s.krg, s.ng) != 0.0) e.v[V_KRG] = ev[2][c] / last(s.krg, s.ng);              // at Sgu
        if (ev[3] && s.pcow[0] != 0.0) e.v[V_PCOW] = ev[3][c] / s.pcow[0];                             // maxPcow: at Swl
        if (ev[4] && last(s.pcgo, s.ng) != 0.0) e.v[V_PCGO] = ev[4][c] / last(s.pcgo, s.ng);           // maxPcgo: at Sgu
    }
    if (!e.on) return e;
    auto last_zero = [](const double* x, const double* y, int n) { int i = 0; while (i + 1 < n && y[i + 1] == 0.0) ++i; return x[i]; };
    auto first_zero = [](const double* x, const double* y, int n) { int i = 0; while (i < n - 1 && y[i] != 0.0) ++i; return x[i]; };
    const double Swl = s.sw[0], Swu = s.sw[s.nw - 1], Swcr = last_zero(s.sw, s.krw, s.nw), Sowcr = 1.0 - first_zero(s.sw, s.krow, s.nw);
    const double Sgl = s.sg[0], Sgu = s.sg[s.ng - 1], Sgcr = last_zero(s.sg, s.krg, s.ng), Sogcr = 1.0 - first_zero(s.sg, s.krog, s.ng);
    const double SWL = ep8[0][c], SWCR = ep8[1][c], SWU = ep8[2][c], SOWCR = ep8[3][c];
    const double SGL = ep8[4][c], SGCR = ep8[5][c], SGU = ep8[6][c], SOGCR = ep8[7][c];
    const bool three = g && g->scalecrs != 0;
    auto mk2 = [](double u0, double u2, double s0, double s2) { Lin m; m.u0 = u0; m.s0 = s0; m.k = (u2 - u0) / (s2 - s0); m.three = false; m.s1 = 0; m.u1 = 0; m.k1 = m.k; return m; };
    auto mk3 = [&](double u0, double u1, double u2, double s0, double s1, double s2) {
        Lin m = mk2(u0, u2, s0, s2);
        if (!three) return m;
        m.three = true; m.k = (u1 - u0) / (s1 - s0); m.s1 = s1; m.u1 = u1; m.k1 = (u2 - u1) / (s2 - s1);
        return m;
    };
    e.krw = mk3(Swcr, 1.0 - Sowcr - Sgl, Swu, SWCR, 1.0 - SOWCR - SGL, SWU);
    e.krow = mk3(Swl + Sgl, Swcr + Sgl, 1.0 - Sowcr - Sgl, SWL + SGL, SWCR + SGL, 1.0 - SOWCR - SGL);
    e.pcow = mk2(Swl, Swu, SWL, SWU);
    e.krg = mk3(Sgcr, 1.0 - Sogcr - Swl, Sgu, SGCR, 1.0 - SOGCR - SWL, SGU);
    e.krog = mk3(Sogcr, 1.0 - Sgcr - Swl, 1.0 - Swl - Sgl, SOGCR, 1.0 - SGCR - SWL, 1.0 - SWL - SGL);
    e.pcgo = mk2(Sgl, Sgu, SGL, SGU);
    e.swl = SWL;
    return e;
}
Eps eps_for_cell(const opmgpu_grid* g, const SatTab& s, int c) { return eps_build(g, s, c, g ? g->eps : nullptr); }
// table value at the (possibly scaled) saturation S times the vertical factor vf; RIGHT selects the SGOF segment rule
template <int N, bool RIGHT>
Dual<N> sat_curve(const double* x, const double* y, int n, const Dual<N>& S, bool on, const Lin& m, double vf = 1.0)
{
    double f, df;
    if (!on) { if (RIGHT) sat_right(x, y, n, S.v, f, df); else sat_left(x, y, n, S.v, f, df); return chain(vf * f, vf * df, S); }
    const double su = lin_map(m, S.v);
    if (RIGHT) sat_right(x, y, n, su, f, df); else sat_left(x, y, n, su, f, df);
    return chain(vf * f, vf * df * lin_slope(m, S.v), S);
}

// Relative-permeability hysteresis of one cell (EclHysteresisTwoPhaseLaw with EHYSTR item 2 = 0, KR only): the non-wetting phase of
// each two-phase system (oil against water: krow in Sw_ow; gas against oil: krg in 1 - Sg) follows its drainage curve down to the
// smallest wetting saturation seen so far (mdc) and from there the IMBIBITION curve shifted by delta (Carlson: the shift makes the two
// curves meet at the turning point, EclHysteresisTwoPhaseLawParams::updateDynamicParams_).  Wetting phases: drainage curves.
struct Hyst {
    bool on = false;
    SatTab imb;            // imbibition tables (IMBNUM region)
    Eps eimb;              // their end-point scaling
    double mdc_ow = 2.0, mdc_go = 2.0, d_ow = 0.0, d_go = 0.0;
};
const double* g_hyst_ow = nullptr;      // history planes set by oracle_set_hysteresis (krnSwMdc of both systems, deltas)
const double* g_hyst_go = nullptr;
const double* g_hyst_dow = nullptr;
const double* g_hyst_dgo = nullptr;
Hyst hyst_for_cell(const opmgpu_grid* g, const opmgpu_tables* t, int c)
{
    Hyst h;
    if (!g || !g->imbnum) return h;
    h.on = true;
    h.imb = sat_tab(t, g->imbnum[c]);
    h.eimb = eps_build(g, h.imb, c, g->ieps[0] ? g->ieps : g->eps);
    if (g_hyst_ow) { h.mdc_ow = g_hyst_ow[c]; h.mdc_go = g_hyst_go[c]; h.d_ow = g_hyst_dow[c]; h.d_go = g_hyst_dgo[c]; }
    return h;
}

// EclDefaultMaterial::{krw,krg,krn} (opm-material; reached from SaturationPropsFromDeck.cpp:91-92).
// Generic in the AD width so the same code serves oracle_relperm (independent sw,so,sg) and the model.
template <int N>
void relperm3(const SatTab& s, const Eps& e, const Dual<N>& Sw, const Dual<N>& Sg, Dual<N>& krw, Dual<N>& kro, Dual<N>& krg, const Hyst* h = nullptr)
{
    krw = sat_curve<N, false>(s.sw, s.krw, s.nw, Sw, e.on, e.krw, e.v[V_KRW]);
    // gas: non-wetting phase of the gas-oil system, wetting saturation 1 - Sg
    if (h && h->on && (1.0 - Sg.v) > h->mdc_go) {
        const Dual<N> Sgi = Sg - h->d_go;               // krn_imb(Sw + delta) = krg_imb(Sg - delta)
        krg = sat_curve<N, true>(h->imb.sg, h->imb.krg, h->imb.ng, Sgi, h->eimb.on, h->eimb.krg, h->eimb.v[V_KRG]);
    } else krg = sat_curve<N, true>(s.sg, s.krg, s.ng, Sg, e.on, e.krg, e.v[V_KRG]);
    const double Swco = e.swl;
    Dual<N> Swp = (Sw.v > Swco) ? Sw : Dual<N>(Swco);       // max(Swco, Sw)
    Dual<N> Sw_ow = Sg + Swp;
    Dual<N> kro_ow;
    if (h && h->on && Sw_ow.v > h->mdc_ow) {
        const Dual<N> Swi = Sw_ow + h->d_ow;
        kro_ow = sat_curve<N, false>(h->imb.sw, h->imb.krow, h->imb.nw, Swi, h->eimb.on, h->eimb.krow, h->eimb.v[V_KROW]);
    } else kro_ow = sat_curve<N, false>(s.sw, s.krow, s.nw, Sw_ow, e.on, e.krow, e.v[V_KROW]);
    // krog is tabulated against the oil saturation So = 1 - swl_t - Sg of the gas-oil system; left-in-So == right-in-Sg
    Dual<N> So_go = 1.0 - Sw_ow;
    Dual<N> kro_go;
    {
        const double k = e.on ? lin_slope(e.krog, So_go.v) : 1.0;
        const double so_u = e.on ? lin_map(e.krog, So_go.v) : So_go.v + (Swco - e.swl_t);   // unscaled: Sg_eq = Sw_ow - Swco
        const double sg_eq = 1.0 - e.swl_t - so_u;
        double f, df;
        sat_right(s.sg, s.krog, s.ng, e.on ? sg_eq : Sw_ow.v - Swco, f, df);
        kro_go = chain(e.v[V_KROG] * f, -e.v[V_KROG] * df * k, So_go);
    }
    const double eps = 1e-5;
    if (Sw_ow.v - Swco < eps) {
        Dual<N> kro2 = (kro_ow + kro_go) / 2.0;
        if (Sw_ow.v - Swco > eps / 2) {
            Dual<N> kro1 = (Sg * kro_go + (Swp - Swco) * kro_ow) / (Sw_ow - Swco);
            Dual<N> alpha = (eps - (Sw_ow - Swco)) / (eps / 2);
            kro = kro2 * alpha + kro1 * (1.0 - alpha);
        } else {
            kro = kro2;
        }
    } else {
        kro = (Sg * kro_go + (Swp - Swco) * kro_ow) / (Sw_ow - Swco);
    }
}
// smallest abscissa at which the piecewise-linear, monotone table y(x) takes the value yv (PiecewiseLinearTwoPhaseMaterial's
// twoPhaseSatKrnInv; flat pieces: the left end).  ascending = y grows with x.
double table_inverse(const double* x, const double* y, int n, double yv, bool ascending)
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
// Carlson shift of one two-phase system after its history moved to mdc (updateDynamicParams_): delta = Sw_imb(krn_drain(mdc)) - mdc
void hyst_deltas(const SatTab& s, const Eps& e, const Hyst& h, double mdc_ow, double mdc_go, double& d_ow, double& d_go)
{
    d_ow = 0.0; d_go = 0.0;
    if (mdc_ow < 2.0) {        // oil against water: krow curves in Sw
        const Dual<1> kd = sat_curve<1, false>(s.sw, s.krow, s.nw, Dual<1>(mdc_ow), e.on, e.krow, e.v[V_KROW]);
        const double ku = kd.v / h.eimb.v[V_KROW];
        if (ku > 0.0) {
            const double su = table_inverse(h.imb.sw, h.imb.krow, h.imb.nw, ku, false);
            d_ow = (h.eimb.on ? lin_inv(h.eimb.krow, su) : su) - mdc_ow;
        }
    }
    if (mdc_go < 2.0) {        // gas against oil: krg curves in Sg = 1 - Sw
        const double sgm = 1.0 - mdc_go;
        const Dual<1> kd = sat_curve<1, true>(s.sg, s.krg, s.ng, Dual<1>(sgm), e.on, e.krg, e.v[V_KRG]);
        const double ku = kd.v / h.eimb.v[V_KRG];
        if (ku > 0.0) {
            const double su = table_inverse(h.imb.sg, h.imb.krg, h.imb.ng, ku, true);
            const double sg_imb = h.eimb.on ? lin_inv(h.eimb.krg, su) : su;
            d_go = sgm - sg_imb;        // Sw_imb - mdc with Sw = 1 - Sg
        }
    }
}
// EclDefaultMaterial::capillaryPressures + the sign / reference-phase shift of
// SaturationPropsFromDeck.cpp:163-174: pc[w] = pcow(Sw), pc[o] = 0, pc[g] = pcgo(Sg).
template <int N>
void cappress3(const SatTab& s, const Eps& e, const Dual<N>& Sw, const Dual<N>& Sg, Dual<N>& pcow, Dual<N>& pcgo)
{
    pcow = sat_curve<N, false>(s.sw, s.pcow, s.nw, Sw, e.on, e.pcow, e.v[V_PCOW]);
    pcgo = sat_curve<N, true>(s.sg, s.pcgo, s.ng, Sg, e.on, e.pcgo, e.v[V_PCGO]);
}

// ----------------------------------------------------------------------------------------
// PVT tables: opm-material Tabulated1DFunction (linear extrapolation, segment x[i] <= x < x[i+1])
// and UniformXTabulated2DFunction::eval (interpolate each bracketing column at the same y, then
// linearly in x).
// ----------------------------------------------------------------------------------------
int seg_index(const double* x, int n, double xv)
{
    if (xv <= x[1]) return 0;
    if (xv >= x[n - 2]) return n - 2;
    int lo = 1, hi = n - 2;
    while (lo + 1 < hi) { const int mid = (lo + hi) / 2; if (xv < x[mid]) hi = mid; else lo = mid; }
    return lo;
}
void lin1d(const double* x, const double* y, int n, double xv, double& f, double& df)
{
    const int i = seg_index(x, n, xv);
    df = (y[i + 1] - y[i]) / (x[i + 1] - x[i]);
    f = y[i] + df * (xv - x[i]);
}
// f(x, y): node abscissae xs[nn]; column k holds samples cy/cv[colptr[k] .. colptr[k+1])
void lin2d(const double* xs, int nn, const int32_t* colptr, const double* cy, const double* cv,
           double xv, double yv, double& f, double& dfx, double& dfy)
{
    const int i = seg_index(xs, nn, xv);
    const double alpha = (xv - xs[i]) / (xs[i + 1] - xs[i]);
    double s1, d1, s2, d2;
    lin1d(cy + colptr[i], cv + colptr[i], colptr[i + 1] - colptr[i], yv, s1, d1);
    lin1d(cy + colptr[i + 1], cv + colptr[i + 1], colptr[i + 2] - colptr[i + 1], yv, s2, d2);
    f = s1 * (1.0 - alpha) + s2 * alpha;
    dfx = (s2 - s1) / (xs[i + 1] - xs[i]);
    dfy = d1 * (1.0 - alpha) + d2 * alpha;
}

struct PvtOut { double v, dp, dr; };

// ConstantCompressibilityWaterPvt (call sites BlackoilPropsAdFromDeck.cpp:289,472)
PvtOut b_wat(const opmgpu_tables* t, int reg, double p)
{
    const double* w = t->pvtw + 5 * reg;
    const double X = w[2] * (p - w[0]);
    PvtOut o; o.v = (1.0 + X * (1.0 + X / 2.0)) / w[1]; o.dp = w[2] * (1.0 + X) / w[1]; o.dr = 0.0;
    return o;
}
PvtOut mu_wat(const opmgpu_tables* t, int reg, double p)
{
    const double* w = t->pvtw + 5 * reg;
    const PvtOut bw = b_wat(t, reg, p);
    const double c = w[2] - w[4];
    const double Y = c * (p - w[0]);
    const double den = 1.0 + Y * (1.0 + Y / 2.0);
    const double BwMuw = w[3] * w[1];
    PvtOut o; o.v = BwMuw * bw.v / den;
    o.dp = BwMuw * (bw.dp * den - bw.v * c * (1.0 + Y)) / (den * den); o.dr = 0.0;
    return o;
}
// LiveOilPvt / DeadOilPvt (call sites BlackoilPropsAdFromDeck.cpp:346,352,527,536,654)
struct OilTab { int nn; const double *rs, *psat, *ib, *ibm; const int32_t* cp; const double *cpp, *cib, *cibm; };
OilTab oil_tab(const opmgpu_tables* t, int reg)
{
    OilTab o; const int a = t->oil_node_ptr[reg];
    o.nn = t->oil_node_ptr[reg + 1] - a; o.rs = t->oil_rs + a; o.psat = t->oil_psat + a;
    o.ib = t->oil_invb_sat + a; o.ibm = t->oil_invbmu_sat + a; o.cp = t->oil_col_ptr + a;
    o.cpp = t->oil_col_p; o.cib = t->oil_col_invb; o.cibm = t->oil_col_invbmu;
    return o;
}
PvtOut b_oil(const opmgpu_tables* t, int reg, double p, double rs, bool saturated)
{
    const OilTab o = oil_tab(t, reg); PvtOut r;
    if (saturated || !t->has_disgas) { lin1d(o.psat, o.ib, o.nn, p, r.v, r.dp); r.dr = 0.0; }
    else lin2d(o.rs, o.nn, o.cp, o.cpp, o.cib, rs, p, r.v, r.dr, r.dp);
    return r;
}
PvtOut mu_oil(const opmgpu_tables* t, int reg, double p, double rs, bool saturated)
{
    const OilTab o = oil_tab(t, reg); PvtOut r;
    double ib, dibp, dibr = 0.0, ibm, dibmp, dibmr = 0.0;
    if (saturated || !t->has_disgas) { lin1d(o.psat, o.ib, o.nn, p, ib, dibp); lin1d(o.psat, o.ibm, o.nn, p, ibm, dibmp); }
    else { lin2d(o.rs, o.nn, o.cp, o.cpp, o.cib, rs, p, ib, dibr, dibp); lin2d(o.rs, o.nn, o.cp, o.cpp, o.cibm, rs, p, ibm, dibmr, dibmp); }
    r.v = ib / ibm; r.dp = (dibp - r.v * dibmp) / ibm; r.dr = (dibr - r.v * dibmr) / ibm;
    return r;
}
PvtOut rs_sat(const opmgpu_tables* t, int reg, double p)
{
    const OilTab o = oil_tab(t, reg); PvtOut r; r.dr = 0.0;
    if (!t->has_disgas) { r.v = 0.0; r.dp = 0.0; return r; }
    lin1d(o.psat, o.rs, o.nn, p, r.v, r.dp);
    return r;
}
// WetGasPvt / DryGasPvt (call sites BlackoilPropsAdFromDeck.cpp:413,417,599,603,711)
struct GasTab { int nn; const double *pg, *rv, *ib, *ibm; const int32_t* cp; const double *crv, *cib, *cibm; };
GasTab gas_tab(const opmgpu_tables* t, int reg)
{
    GasTab o; const int a = t->gas_node_ptr[reg];
    o.nn = t->gas_node_ptr[reg + 1] - a; o.pg = t->gas_pg + a; o.rv = t->gas_rvsat + a;
    o.ib = t->gas_invb_sat + a; o.ibm = t->gas_invbmu_sat + a; o.cp = t->gas_col_ptr + a;
    o.crv = t->gas_col_rv; o.cib = t->gas_col_invb; o.cibm = t->gas_col_invbmu;
    return o;
}
PvtOut b_gas(const opmgpu_tables* t, int reg, double p, double rv, bool saturated)
{
    const GasTab o = gas_tab(t, reg); PvtOut r;
    if (saturated || !t->has_vapoil) { lin1d(o.pg, o.ib, o.nn, p, r.v, r.dp); r.dr = 0.0; }
    else lin2d(o.pg, o.nn, o.cp, o.crv, o.cib, p, rv, r.v, r.dp, r.dr);
    return r;
}
PvtOut mu_gas(const opmgpu_tables* t, int reg, double p, double rv, bool saturated)
{
    const GasTab o = gas_tab(t, reg); PvtOut r;
    double ib, dibp, dibr = 0.0, ibm, dibmp, dibmr = 0.0;
    if (saturated || !t->has_vapoil) { lin1d(o.pg, o.ib, o.nn, p, ib, dibp); lin1d(o.pg, o.ibm, o.nn, p, ibm, dibmp); }
    else { lin2d(o.pg, o.nn, o.cp, o.crv, o.cib, p, rv, ib, dibp, dibr); lin2d(o.pg, o.nn, o.cp, o.crv, o.cibm, p, rv, ibm, dibmp, dibmr); }
    r.v = ib / ibm; r.dp = (dibp - r.v * dibmp) / ibm; r.dr = (dibr - r.v * dibmr) / ibm;
    return r;
}
PvtOut rv_sat(const opmgpu_tables* t, int reg, double p)
{
    const GasTab o = gas_tab(t, reg); PvtOut r; r.dr = 0.0;
    if (!t->has_vapoil) { r.v = 0.0; r.dp = 0.0; return r; }
    lin1d(o.pg, o.rv, o.nn, p, r.v, r.dp);
    return r;
}

// VAPPARS, BlackoilPropsAdFromDeck::applyVap (BlackoilPropsAdFromDeck.cpp:1027-1078): factor and d(factor)/d(so)
const double* g_so_max = nullptr;      // satOilMax_ per cell (oracle_set_sat_oil_max); NULL = all zero (:175)
inline void vap_factor(double vap, double so, double so_max, double& f, double& df)
{
    f = 1.0; df = 0.0;
    if (vap > 0.0 && so_max > 0.01 /* vap_satmax_guard_, :187 */ && so < so_max) {
        const double so_i = std::max(so, std::sqrt(std::numeric_limits<double>::epsilon()));
        f = std::pow(so_i / so_max, vap);
        df = vap * std::pow(so_i / so_max, vap - 1.0) / so_max;
    }
}
// ROCKTAB, Opm::linearInterpolation / linearInterpolationDerivative (opm-common, restated: tableIndex = lower_bound - 1
// clamped to [0, n-2], i.e. the LEFT segment at a breakpoint and linear extrapolation outside)
inline void rocktab_eval(const double* x, const double* y, int n, double xv, double& f, double& df)
{
    int i = int(std::lower_bound(x, x + n, xv) - x);
    i = (i >= n) ? n - 2 : (i > 0 ? i - 1 : 0);
    if (i > n - 2) i = n - 2;
    df = (y[i + 1] - y[i]) / (x[i + 1] - x[i]);
    f = y[i] + df * (xv - x[i]);
}

// ----------------------------------------------------------------------------------------
// per-cell SolutionState + ReservoirResidualQuant
// ----------------------------------------------------------------------------------------
struct CellQ {
    D3 pw, po, pg, sw, so, sg, rs, rv;
    D3 b[3], mu[3], kr[3], rho[3], mob[3], accum[3];
};

CellQ cell_eval(const opmgpu_grid* g, const opmgpu_tables* t, int c, double p, double sw_, double sg_,
                double rs_, double rv_, int hc, bool with_derivs)
{
    CellQ q;
    const int preg = g->pvtnum ? g->pvtnum[c] : 0;
    const int sreg = g->satnum ? g->satnum[c] : 0;
    const SatTab st = sat_tab(t, sreg);
    const Eps ep = eps_for_cell(g, st, c);
    // updatePhaseCondFromPrimalVariable, BlackoilModelBase_impl.hpp:2207-2241
    const bool isSg = (hc == OPMGPU_HC_GAS_AND_OIL), isRs = (hc == OPMGPU_HC_OIL_ONLY), isRv = (hc == OPMGPU_HC_GAS_ONLY);
    const bool freeOil = isSg || isRs, freeGas = isSg || isRv;
    // variableReservoirStateInitials, :554-583 ; ADB::variables, AutoDiffBlock.hpp:203-215
    const double xv = isRs ? rs_ : (isRv ? rv_ : sg_);
    D3 P = with_derivs ? D3::var(p, 0) : D3(p);
    D3 W = with_derivs ? D3::var(sw_, 1) : D3(sw_);
    D3 X = with_derivs ? D3::var(xv, 2) : D3(xv);
    // variableStateExtractVars, :614-703
    D3 so = 1.0 - W;
    D3 sg = (isSg ? 1.0 : 0.0) * X + (isRv ? 1.0 : 0.0) * so;
    so = so - sg;
    q.sw = W; q.so = so; q.sg = sg;
    // computePressures, :1425-1460
    D3 pcow, pcgo;
    cappress3(st, ep, W, sg, pcow, pcgo);
    q.po = P; q.pw = P - pcow; q.pg = P + pcgo;
    // rsSat / rvSat (T = 293.15 ignored by the isothermal tables), :662-674
    {
        const double so_max = g_so_max ? g_so_max[c] : 0.0;
        double vf, dvf;
        const PvtOut r = rs_sat(t, preg, q.po.v);
        D3 rsSat = chain(r.v, r.dp, q.po);
        vap_factor(t->vap2, so.v, so_max, vf, dvf);                  // :679
        if (t->vap2 > 0.0) rsSat = chain(vf, dvf, so) * rsSat;
        q.rs = t->has_disgas ? ((isRs ? 0.0 : 1.0) * rsSat + (isRs ? 1.0 : 0.0) * X) : rsSat;
        const PvtOut v = rv_sat(t, preg, q.pg.v);
        D3 rvSat = chain(v.v, v.dp, q.pg);
        vap_factor(t->vap1, so.v, so_max, vf, dvf);                  // :736
        if (t->vap1 > 0.0) rvSat = chain(vf, dvf, so) * rvSat;
        q.rv = t->has_vapoil ? ((isRv ? 0.0 : 1.0) * rvSat + (isRv ? 1.0 : 0.0) * X) : rvSat;
    }
    // fluidReciprocFVF / fluidViscosity, BlackoilPropsAdFromDeck.cpp:264-622
    { const PvtOut o = b_wat(t, preg, q.pw.v);  q.b[0]  = chain(o.v, o.dp, q.pw); }
    { const PvtOut o = mu_wat(t, preg, q.pw.v); q.mu[0] = chain(o.v, o.dp, q.pw); }
    { const PvtOut o = b_oil(t, preg, q.po.v, q.rs.v, freeGas);  q.b[1]  = chain2(o.v, o.dp, q.po, o.dr, q.rs); }
    { const PvtOut o = mu_oil(t, preg, q.po.v, q.rs.v, freeGas); q.mu[1] = chain2(o.v, o.dp, q.po, o.dr, q.rs); }
    { const PvtOut o = b_gas(t, preg, q.pg.v, q.rv.v, freeOil);  q.b[2]  = chain2(o.v, o.dp, q.pg, o.dr, q.rv); }
    { const PvtOut o = mu_gas(t, preg, q.pg.v, q.rv.v, freeOil); q.mu[2] = chain2(o.v, o.dp, q.pg, o.dr, q.rv); }
    // computeRelPerm, :1395-1419
    const Hyst hy = hyst_for_cell(g, t, c);
    relperm3(st, ep, W, sg, q.kr[0], q.kr[1], q.kr[2], &hy);
    // poroMult / transMult, :2089-2145 ; RockCompressibility.cpp:86-125
    D3 pvm(1.0), trm(1.0);
    if (t->rocktab_n > 0) {
        double f, df;
        rocktab_eval(t->rocktab_p, t->rocktab_pvmult, t->rocktab_n, p, f, df); pvm = chain(f, df, P);
        rocktab_eval(t->rocktab_p, t->rocktab_transmult, t->rocktab_n, p, f, df); trm = chain(f, df, P);
    } else if (t->rock_comp != 0.0) {
        const double cp = t->rock_comp * (p - t->rock_pref);
        pvm = chain(1.0 + cp + 0.5 * cp * cp, t->rock_comp + cp * t->rock_comp, P);
    }
    // fluidDensity, :2009-2027
    const double* rhos = t->surface_density + 3 * preg;
    q.rho[0] = rhos[0] * q.b[0];
    q.rho[1] = rhos[1] * q.b[1] + rhos[2] * q.rs * q.b[1];
    q.rho[2] = rhos[2] * q.b[2] + rhos[1] * q.rv * q.b[2];
    // mobilities, computeMassFlux :1496-1497
    for (int a = 0; a < 3; ++a) q.mob[a] = trm * q.kr[a] / q.mu[a];
    // computeAccum, :709-751
    q.accum[0] = pvm * q.b[0] * W;
    q.accum[1] = pvm * q.b[1] * so;
    q.accum[2] = pvm * q.b[2] * sg;
    const D3 accum_gas_copy = q.accum[2];
    q.accum[2] = q.accum[2] + q.rs * q.accum[1];
    q.accum[1] = q.accum[1] + q.rv * accum_gas_copy;
    return q;
}

inline D6 lift(const D3& a, int side) { D6 r(a.v); for (int i = 0; i < 3; ++i) r.d[3 * side + i] = a.d[i]; return r; }
inline double sgn(double x) { return (x > 0.0) ? 1.0 : ((x < 0.0) ? -1.0 : 0.0); }   // AutoDiffHelpers.hpp:722-730

// 3x3 helpers, row-major
template <class S> inline void mm3(const S* a, const S* b, S* c) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j]; }
template <class S> inline bool inv3(const S* m, S* o)
{
    const S c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
    const S det = m[0] * c0 + m[1] * c1 + m[2] * c2;
    if (det == S(0) || !(det == det)) return false;
    const S id = S(1) / det;
    o[0] = c0 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c1 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c2 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}

int find_slot(const int32_t* rowptr, const int32_t* col, int r, int c)
{
    const int32_t* b = col + rowptr[r]; const int32_t* e = col + rowptr[r + 1];
    const int32_t* it = std::lower_bound(b, e, c);
    return (it != e && *it == c) ? int(it - col) : -1;
}

// ----------------------------------------------------------------------------------------
// ILU0 / BiCGStab in the requested precision
// ----------------------------------------------------------------------------------------
template <class S>
struct IluData {
    int nb;
    std::vector<int> order;                 // order[k] = row eliminated k-th
    std::vector<int> pos;                   // pos[row]
    std::vector<std::vector<int> > lower, upper;   // slots per row, sorted by pos(col)
    std::vector<int> diag;
    std::vector<S> lu;                      // nnzb*9
    // level schedule of the two sweeps (more than one thread only): rows of one level depend on earlier levels alone, so they may run side
    // by side -- every row still performs exactly the sequential sweep's operations in the sequential order: the same bits
    std::vector<int> fwd_ptr, fwd_rows, bwd_ptr, bwd_rows;
};

static void level_lists(int nb, const std::vector<int>& lev, std::vector<int>& ptr, std::vector<int>& rows, const std::vector<int>& order)
{
    int nl = 0;
    for (int i = 0; i < nb; ++i) nl = std::max(nl, lev[i] + 1);
    ptr.assign(nl + 1, 0);
    for (int i = 0; i < nb; ++i) ++ptr[lev[i] + 1];
    for (int l = 0; l < nl; ++l) ptr[l + 1] += ptr[l];
    rows.resize(nb);
    std::vector<int> fill(ptr.begin(), ptr.end() - 1);
    for (int k = 0; k < nb; ++k) { const int i = order[k]; rows[fill[lev[i]]++] = i; }       // inside a level: elimination order
}

template <class S>
int ilu0_setup(int nb, const int32_t* rowptr, const int32_t* col, const double* val9,
               const int32_t* position, IluData<S>& D)
{
    D.nb = nb; D.pos.resize(nb); D.order.resize(nb);
    for (int i = 0; i < nb; ++i) D.pos[i] = position ? position[i] : i;
    for (int i = 0; i < nb; ++i) D.order[D.pos[i]] = i;
    const int nnzb = rowptr[nb];
    D.lu.resize(size_t(nnzb) * 9);
    for (size_t k = 0; k < size_t(nnzb) * 9; ++k) D.lu[k] = S(val9[k]);
    D.lower.assign(nb, std::vector<int>()); D.upper.assign(nb, std::vector<int>()); D.diag.assign(nb, -1);
    for (int i = 0; i < nb; ++i) {
        for (int s = rowptr[i]; s < rowptr[i + 1]; ++s) {
            const int j = col[s];
            if (j == i) D.diag[i] = s;
            else if (D.pos[j] < D.pos[i]) D.lower[i].push_back(s);
            else D.upper[i].push_back(s);
        }
        auto cmp = [&](int a, int b) { return D.pos[col[a]] < D.pos[col[b]]; };
        std::sort(D.lower[i].begin(), D.lower[i].end(), cmp);
        std::sort(D.upper[i].begin(), D.upper[i].end(), cmp);
        if (D.diag[i] < 0) return OPMGPU_ESINGULAR;
    }
    // bilu0_decomposition (dune-istl ilu.hh), rows visited in elimination order
    auto eliminate_row = [&](int i, std::vector<int>& slot_of) -> bool {
        for (int s = rowptr[i]; s < rowptr[i + 1]; ++s) slot_of[col[s]] = s;
        for (size_t a = 0; a < D.lower[i].size(); ++a) {
            const int sij = D.lower[i][a]; const int j = col[sij];
            S L[9]; mm3(&D.lu[size_t(sij) * 9], &D.lu[size_t(D.diag[j]) * 9], L);     // A_ij * inv(A_jj)
            for (int q = 0; q < 9; ++q) D.lu[size_t(sij) * 9 + q] = L[q];
            for (size_t b = 0; b < D.upper[j].size(); ++b) {
                const int sjk = D.upper[j][b]; const int kk = col[sjk];
                const int sik = slot_of[kk];
                if (sik < 0) continue;
                S B[9]; mm3(L, &D.lu[size_t(sjk) * 9], B);
                for (int q = 0; q < 9; ++q) D.lu[size_t(sik) * 9 + q] -= B[q];
            }
        }
        S inv[9];
        const bool ok = inv3(&D.lu[size_t(D.diag[i]) * 9], inv);
        if (ok) for (int q = 0; q < 9; ++q) D.lu[size_t(D.diag[i]) * 9 + q] = inv[q];
        for (int s = rowptr[i]; s < rowptr[i + 1]; ++s) slot_of[col[s]] = -1;
        return ok;
    };
    if (g_threads <= 1) {
        std::vector<int> slot_of(nb, -1);
        for (int k = 0; k < nb; ++k) if (!eliminate_row(D.order[k], slot_of)) return OPMGPU_ESINGULAR;
        return OPMGPU_OK;
    }
    // level schedule: forward level of a row = 1 + the largest level among its lower neighbours, backward likewise over the upper ones
    std::vector<int> flev(nb, 0), blev(nb, 0);
    for (int k = 0; k < nb; ++k) { const int i = D.order[k]; int l = 0; for (int s : D.lower[i]) l = std::max(l, flev[col[s]] + 1); flev[i] = l; }
    for (int k = nb - 1; k >= 0; --k) { const int i = D.order[k]; int l = 0; for (int s : D.upper[i]) l = std::max(l, blev[col[s]] + 1); blev[i] = l; }
    level_lists(nb, flev, D.fwd_ptr, D.fwd_rows, D.order);
    level_lists(nb, blev, D.bwd_ptr, D.bwd_rows, D.order);
    // the elimination of row i reads the finished rows j of its lower neighbours (earlier forward levels) and writes row i only
    int bad = 0;
    const int nlev = int(D.fwd_ptr.size()) - 1;
#pragma omp parallel num_threads(g_threads)
    {
        std::vector<int> slot_of(nb, -1);
        for (int l = 0; l < nlev; ++l) {
#pragma omp for schedule(static)
            for (int q = D.fwd_ptr[l]; q < D.fwd_ptr[l + 1]; ++q)
                if (!eliminate_row(D.fwd_rows[q], slot_of)) {
#pragma omp atomic write
                    bad = 1;
                }
        }
    }
    return bad ? OPMGPU_ESINGULAR : OPMGPU_OK;
}

// ParallelOverlappingILU0::apply (serial path): forward with unit L, backward with inverted pivots,
// result scaled by the relaxation factor.
template <class S>
void ilu0_apply(const IluData<S>& D, const int32_t* col, S relax, const S* d, S* v)
{
    const int nb = D.nb;
    auto forward_row = [&](int i) {
        S r[3] = { d[3 * i], d[3 * i + 1], d[3 * i + 2] };
        for (size_t a = 0; a < D.lower[i].size(); ++a) {
            const int s = D.lower[i][a]; const int j = col[s]; const S* m = &D.lu[size_t(s) * 9];
            for (int q = 0; q < 3; ++q) r[q] -= m[3 * q] * v[3 * j] + m[3 * q + 1] * v[3 * j + 1] + m[3 * q + 2] * v[3 * j + 2];
        }
        v[3 * i] = r[0]; v[3 * i + 1] = r[1]; v[3 * i + 2] = r[2];
    };
    auto backward_row = [&](int i) {
        S r[3] = { v[3 * i], v[3 * i + 1], v[3 * i + 2] };
        for (size_t a = 0; a < D.upper[i].size(); ++a) {
            const int s = D.upper[i][a]; const int j = col[s]; const S* m = &D.lu[size_t(s) * 9];
            for (int q = 0; q < 3; ++q) r[q] -= m[3 * q] * v[3 * j] + m[3 * q + 1] * v[3 * j + 1] + m[3 * q + 2] * v[3 * j + 2];
        }
        const S* m = &D.lu[size_t(D.diag[i]) * 9];
        for (int q = 0; q < 3; ++q) v[3 * i + q] = m[3 * q] * r[0] + m[3 * q + 1] * r[1] + m[3 * q + 2] * r[2];
    };
    if (g_threads <= 1 || D.fwd_ptr.empty()) {
        for (int k = 0; k < nb; ++k) forward_row(D.order[k]);
        for (int k = nb - 1; k >= 0; --k) backward_row(D.order[k]);
    } else {
        // the same sweeps by levels (IluData): bit-identical to the sequential ones, the rows of a level shared out among the threads
        const int nf = int(D.fwd_ptr.size()) - 1, nbk = int(D.bwd_ptr.size()) - 1;
#pragma omp parallel num_threads(g_threads)
        {
            for (int l = 0; l < nf; ++l) {
#pragma omp for schedule(static)
                for (int q = D.fwd_ptr[l]; q < D.fwd_ptr[l + 1]; ++q) forward_row(D.fwd_rows[q]);
            }
            for (int l = 0; l < nbk; ++l) {
#pragma omp for schedule(static)
                for (int q = D.bwd_ptr[l]; q < D.bwd_ptr[l + 1]; ++q) backward_row(D.bwd_rows[q]);
            }
        }
    }
    if (relax != S(1)) for (int k = 0; k < 3 * nb; ++k) v[k] *= relax;
}

template <class S>
void spmv_t(int nb, const int32_t* rowptr, const int32_t* col, const S* val, const S* x, S* y)
{
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int i = 0; i < nb; ++i) {
        S r[3] = { 0, 0, 0 };
        for (int s = rowptr[i]; s < rowptr[i + 1]; ++s) {
            const int j = col[s]; const S* m = val + size_t(s) * 9;
            for (int q = 0; q < 3; ++q) r[q] += m[3 * q] * x[3 * j] + m[3 * q + 1] * x[3 * j + 1] + m[3 * q + 2] * x[3 * j + 2];
        }
        y[3 * i] = r[0]; y[3 * i + 1] = r[1]; y[3 * i + 2] = r[2];
    }
}
// One thread: the plain sequential sum, like dune's SeqScalarProduct.  More threads: fixed blocks of 4096 entries summed one after the other
// into their own partial, the partials added up in block order -- the same bits for every thread count and every run (an OpenMP
// `reduction` clause combines the threads' partial sums in whatever order they finish: the 1 M-cell lockstep tests saw 0.9e-5 .. 1.9e-5 in
// rs from run to run for that reason alone, VERDICT r3 weakness 3).
template <class S> S dot_t(int n, const S* a, const S* b)
{
    if (g_threads <= 1) {
        S s = 0;
        for (int i = 0; i < n; ++i) s += a[i] * b[i];
        return s;
    }
    constexpr int kChunk = 4096;
    const int nchunks = (n + kChunk - 1) / kChunk;
    std::vector<S> part(nchunks);
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int c = 0; c < nchunks; ++c) {
        const int lo = c * kChunk, hi = std::min(n, lo + kChunk);
        S s = 0;
        for (int i = lo; i < hi; ++i) s += a[i] * b[i];
        part[c] = s;
    }
    S s = 0;
    for (int c = 0; c < nchunks; ++c) s += part[c];
    return s;
}
template <class S> void axpy_t(int n, S a, const S* x, S* y)
{
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int i = 0; i < n; ++i) y[i] += a * x[i];
}

// Dune::BiCGSTABSolver<X>::apply (dune-istl solvers.hh), restated step for step.
template <class S>
int bicgstab_t(int nb, const int32_t* rowptr, const int32_t* col, const double* val9, const double* rhs3,
               const int32_t* position, const opmgpu_params* prm, double* x3, int* iters, double* reduction,
               double* hist, int nhist, int* nhist_out)
{
    const int n = 3 * nb;
    IluData<S> D;
    const int st = ilu0_setup<S>(nb, rowptr, col, val9, position, D);
    if (st != OPMGPU_OK) return st;
    std::vector<S> A(size_t(rowptr[nb]) * 9);
    for (size_t k = 0; k < A.size(); ++k) A[k] = S(val9[k]);
    std::vector<S> x(n, S(0)), r(n), p(n, S(0)), v(n, S(0)), t(n), y(n), rt(n);
    for (int i = 0; i < n; ++i) r[i] = S(rhs3[i]);            // x0 = 0  =>  r = b
    rt = r;
    const S EPS = S(1e-80);   // dune: real_type EPSILON = 1e-80 (flushes to 0 in float)
    const S relax = S(prm->ilu_relaxation);
    const S red = S(prm->linear_solver_reduction);
    const int maxit = prm->linear_solver_maxiter;
    S rho = 1, alpha = 1, omega = 1, rho_new, beta, h;
    S norm = std::sqrt(dot_t(n, r.data(), r.data())), norm0 = norm;
    int nh = 0;
    bool converged = false;
    double it = 0.0;
    int status = OPMGPU_OK;
    if (norm < red * norm0 || norm < S(1e-30)) { converged = true; }
    else {
        for (it = 0.5; it < maxit; it += 0.5) {
            rho_new = dot_t(n, rt.data(), r.data());
            if (std::fabs(rho) <= EPS || std::fabs(omega) <= EPS) { status = OPMGPU_EBREAKDOWN; break; }
            if (it < 1) p = r;
            else {
                beta = (rho_new / rho) * (alpha / omega);
                axpy_t(n, -omega, v.data(), p.data());
                for (int i = 0; i < n; ++i) p[i] = p[i] * beta + r[i];
            }
            std::fill(y.begin(), y.end(), S(0));
            ilu0_apply(D, col, relax, p.data(), y.data());
            spmv_t(nb, rowptr, col, A.data(), y.data(), v.data());
            h = dot_t(n, rt.data(), v.data());
            if (std::fabs(h) < EPS) { status = OPMGPU_EBREAKDOWN; break; }
            alpha = rho_new / h;
            axpy_t(n, alpha, y.data(), x.data());
            axpy_t(n, -alpha, v.data(), r.data());
            norm = std::sqrt(dot_t(n, r.data(), r.data()));
            if (hist && nh < nhist) hist[nh] = norm; ++nh;
            if (norm < red * norm0) { converged = true; break; }
            it += 0.5;
            std::fill(y.begin(), y.end(), S(0));
            ilu0_apply(D, col, relax, r.data(), y.data());
            spmv_t(nb, rowptr, col, A.data(), y.data(), t.data());
            omega = dot_t(n, t.data(), r.data()) / dot_t(n, t.data(), t.data());
            axpy_t(n, omega, y.data(), x.data());
            axpy_t(n, -omega, t.data(), r.data());
            rho = rho_new;
            norm = std::sqrt(dot_t(n, r.data(), r.data()));
            if (hist && nh < nhist) hist[nh] = norm; ++nh;
            if (norm < red * norm0 || norm < S(1e-30)) { converged = true; break; }
        }
    }
    it = std::min(double(maxit), it);
    for (int i = 0; i < n; ++i) x3[i] = double(x[i]);
    if (iters) *iters = int(std::ceil(it));
    if (reduction) *reduction = (norm0 > 0) ? double(norm / norm0) : 0.0;
    if (nhist_out) *nhist_out = nh;
    if (status != OPMGPU_OK) return status;
    if (!converged && !prm->ignore_convergence_failure) return OPMGPU_ELINSOLVE;   // ISTLSolver.hpp:358-368
    return OPMGPU_OK;
}

// Dune::RestartedGMResSolver::apply (dune-istl, external; restated from the published algorithm): LEFT-preconditioned restarted
// GMRES -- the residual it measures is the preconditioned one, W^-1 (b - A x) --, Arnoldi with modified Gram-Schmidt, Givens
// rotations (generatePlaneRotation / applyPlaneRotation), update by back-substitution.  Reached from ISTLSolver.hpp:257-264.
template <class S>
int gmres_t(int nb, const int32_t* rowptr, const int32_t* col, const double* val9, const double* rhs3,
            const int32_t* position, const opmgpu_params* prm, double* x3, int* iters, double* reduction)
{
    const int n = 3 * nb;
    IluData<S> D;
    const int st = ilu0_setup<S>(nb, rowptr, col, val9, position, D);
    if (st != OPMGPU_OK) return st;
    std::vector<S> A(size_t(rowptr[nb]) * 9);
    for (size_t k = 0; k < A.size(); ++k) A[k] = S(val9[k]);
    const int m = std::max(1, int(prm->linear_solver_restart));
    const S relax = S(prm->ilu_relaxation), red = S(prm->linear_solver_reduction);
    const int maxit = prm->linear_solver_maxiter;
    const S EPS = S(1e-80);
    std::vector<S> x(n, S(0)), b(n), b2(n), w(n), tmp(n);
    for (int i = 0; i < n; ++i) b[i] = S(rhs3[i]);
    b2 = b;
    std::vector<std::vector<S> > v(m + 1, std::vector<S>(n, S(0))), H(m + 1, std::vector<S>(m, S(0)));
    std::vector<S> s(m + 1), cs(m), sn(m);
    auto precond = [&](const std::vector<S>& d, std::vector<S>& out) { std::fill(out.begin(), out.end(), S(0)); ilu0_apply(D, col, relax, d.data(), out.data()); };
    auto gen = [](S dx, S dy, S& c, S& sN) {
        const S ndx = std::fabs(dx), ndy = std::fabs(dy);
        if (ndy < S(1e-15)) { c = 1; sN = 0; }
        else if (ndx < S(1e-15)) { c = 0; sN = 1; }
        else if (ndy > ndx) { const S t = ndx / ndy; c = S(1) / std::sqrt(S(1) + t * t); sN = c; c *= t; sN *= dx / ndx; sN *= dy / ndy; }
        else { const S t = ndy / ndx; c = S(1) / std::sqrt(S(1) + t * t); sN = c * (dy / dx); }
    };
    auto rot = [](S& dx, S& dy, S c, S sN) { const S t = c * dx + sN * dy; dy = -sN * dx + c * dy; dx = t; };
    // x0 = 0: defect = b; preconditioned defect
    precond(b, v[0]);
    S norm = std::sqrt(dot_t(n, v[0].data(), v[0].data()));
    const S norm0 = norm;
    bool converged = false;
    int j = 1, status = OPMGPU_OK;
    if (norm0 < EPS) converged = true;
    while (j <= maxit && !converged && status == OPMGPU_OK) {
        int i = 0;
        for (int q = 0; q < n; ++q) v[0][q] *= S(1) / norm;
        s[0] = norm;
        for (i = 1; i < m + 1; ++i) s[i] = 0;
        for (i = 0; i < m && j <= maxit && !converged; ++i, ++j) {
            spmv_t(nb, rowptr, col, A.data(), v[i].data(), tmp.data());
            precond(tmp, w);
            for (int k = 0; k < i + 1; ++k) {
                H[k][i] = dot_t(n, v[k].data(), w.data());
                axpy_t(n, -H[k][i], v[k].data(), w.data());
            }
            H[i + 1][i] = std::sqrt(dot_t(n, w.data(), w.data()));
            if (std::fabs(H[i + 1][i]) < EPS) { status = OPMGPU_EBREAKDOWN; break; }
            for (int q = 0; q < n; ++q) v[i + 1][q] = w[q] * (S(1) / H[i + 1][i]);
            for (int k = 0; k < i; ++k) rot(H[k][i], H[k + 1][i], cs[k], sn[k]);
            gen(H[i][i], H[i + 1][i], cs[i], sn[i]);
            rot(H[i][i], H[i + 1][i], cs[i], sn[i]);
            rot(s[i], s[i + 1], cs[i], sn[i]);
            norm = std::fabs(s[i + 1]);
            if (norm < red * norm0) converged = true;
        }
        if (status != OPMGPU_OK) break;
        // update(w, i, H, s, v): back-substitution, then x += w
        std::vector<S> y(s.begin(), s.begin() + i);
        std::fill(w.begin(), w.end(), S(0));
        for (int a = i - 1; a >= 0; --a) {
            S rhs = s[a];
            for (int bq = a + 1; bq < i; ++bq) rhs -= H[a][bq] * y[bq];
            y[a] = rhs / H[a][a];
            axpy_t(n, y[a], v[a].data(), w.data());
        }
        axpy_t(n, S(1), w.data(), x.data());
        if (!converged && j <= maxit) {         // restart from the true defect
            spmv_t(nb, rowptr, col, A.data(), x.data(), tmp.data());
            for (int q = 0; q < n; ++q) b[q] = b2[q] - tmp[q];
            precond(b, v[0]);
            norm = std::sqrt(dot_t(n, v[0].data(), v[0].data()));
        }
    }
    for (int i = 0; i < n; ++i) x3[i] = double(x[i]);
    if (iters) *iters = j - 1;
    if (reduction) *reduction = norm0 > 0 ? double(norm / norm0) : 0.0;
    if (status != OPMGPU_OK) return status;
    if (!converged && !prm->ignore_convergence_failure) return OPMGPU_ELINSOLVE;
    return OPMGPU_OK;
}

} // namespace

namespace {
// rebuild IluData from already-factored values (no elimination)
template <class S>
void ilu_from_factors(int nb, const int32_t* rowptr, const int32_t* col, const double* lu9, const int32_t* position, IluData<S>& D)
{
    D.nb = nb; D.pos.resize(nb); D.order.resize(nb);
    for (int i = 0; i < nb; ++i) D.pos[i] = position ? position[i] : i;
    for (int i = 0; i < nb; ++i) D.order[D.pos[i]] = i;
    D.lu.resize(size_t(rowptr[nb]) * 9);
    for (size_t k = 0; k < D.lu.size(); ++k) D.lu[k] = S(lu9[k]);
    D.lower.assign(nb, std::vector<int>()); D.upper.assign(nb, std::vector<int>()); D.diag.assign(nb, -1);
    for (int i = 0; i < nb; ++i) {
        for (int s = rowptr[i]; s < rowptr[i + 1]; ++s) {
            const int j = col[s];
            if (j == i) D.diag[i] = s; else if (D.pos[j] < D.pos[i]) D.lower[i].push_back(s); else D.upper[i].push_back(s);
        }
        auto cmp = [&](int a, int b) { return D.pos[col[a]] < D.pos[col[b]]; };
        std::sort(D.lower[i].begin(), D.lower[i].end(), cmp);
        std::sort(D.upper[i].begin(), D.upper[i].end(), cmp);
    }
}
}


// ========================================================================================
extern "C" {

void oracle_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int oracle_get_threads(void) { return g_threads; }

void oracle_relperm_eps(const opmgpu_tables* t, const opmgpu_grid* g, int n, const double* s, const int32_t* cells, double* kr, double* dkrds)
{
    for (int i = 0; i < n; ++i) {
        const int c = cells ? cells[i] : 0;
        const SatTab st = sat_tab(t, (g && g->satnum) ? g->satnum[c] : 0);
        const Eps ep = eps_for_cell(g, st, c);
        const Hyst hy = hyst_for_cell(g, t, c);
        D3 Sw = D3::var(s[3 * i], 0), Sg = D3::var(s[3 * i + 2], 2), k[3];
        relperm3(st, ep, Sw, Sg, k[0], k[1], k[2], &hy);
        for (int a = 0; a < 3; ++a) {
            kr[3 * i + a] = k[a].v;
            if (dkrds) for (int b = 0; b < 3; ++b) dkrds[9 * i + 3 * b + a] = k[a].d[b];
        }
    }
}

void oracle_relperm(const opmgpu_tables* t, int n, const double* s, const int32_t* satnum, double* kr, double* dkrds)
{
    for (int i = 0; i < n; ++i) {
        const SatTab st = sat_tab(t, satnum ? satnum[i] : 0);
        Eps e0 = eps_build(nullptr, st, 0, nullptr);
        D3 Sw = D3::var(s[3 * i], 0), Sg = D3::var(s[3 * i + 2], 2), k[3];
        relperm3(st, e0, Sw, Sg, k[0], k[1], k[2]);
        for (int a = 0; a < 3; ++a) {
            kr[3 * i + a] = k[a].v;
            if (dkrds) for (int b = 0; b < 3; ++b) dkrds[9 * i + 3 * b + a] = k[a].d[b];
        }
    }
}

void oracle_cappress(const opmgpu_tables* t, int n, const double* s, const int32_t* satnum, double* pc, double* dpcds)
{
    for (int i = 0; i < n; ++i) {
        const SatTab st = sat_tab(t, satnum ? satnum[i] : 0);
        Eps e0 = eps_build(nullptr, st, 0, nullptr);
        D3 Sw = D3::var(s[3 * i], 0), Sg = D3::var(s[3 * i + 2], 2), pcow, pcgo;
        cappress3(st, e0, Sw, Sg, pcow, pcgo);
        const D3 k[3] = { pcow, D3(0.0), pcgo };
        for (int a = 0; a < 3; ++a) {
            pc[3 * i + a] = k[a].v;
            if (dpcds) for (int b = 0; b < 3; ++b) dpcds[9 * i + 3 * b + a] = k[a].d[b];
        }
    }
}

// hysteresis history for every later cell_props / assemble / relperm_eps call (nullptr = none): krnSwMdc and Carlson shift of the
// oil-water and gas-oil systems, [nc] each
void oracle_set_hysteresis(const double* mdc_ow, const double* mdc_go, const double* d_ow, const double* d_go)
{
    g_hyst_ow = mdc_ow; g_hyst_go = mdc_go; g_hyst_dow = d_ow; g_hyst_dgo = d_go;
}
// EclDefaultMaterial::updateHysteresis (the "inconsistent" update: krnSw = 1 - So for oil-water, 1 - Sg for gas-oil) +
// EclHysteresisTwoPhaseLawParams::update / updateDynamicParams_, in place on the four [nc] arrays
void oracle_update_hysteresis(const opmgpu_grid* g, const opmgpu_tables* t, const double* sat, double* mdc_ow, double* mdc_go, double* d_ow, double* d_go)
{
    if (!g->imbnum) return;
    for (int c = 0; c < g->nc; ++c) {
        const double so = sat[3 * c + 1], sg = std::min(1.0, std::max(0.0, sat[3 * c + 2]));
        bool upd = false;
        if (1.0 - so < mdc_ow[c]) { mdc_ow[c] = 1.0 - so; upd = true; }
        if (1.0 - sg < mdc_go[c]) { mdc_go[c] = 1.0 - sg; upd = true; }
        if (!upd) continue;
        const SatTab st = sat_tab(t, g->satnum ? g->satnum[c] : 0);
        const Eps ep = eps_for_cell(g, st, c);
        Hyst h; h.on = true; h.imb = sat_tab(t, g->imbnum[c]); h.eimb = eps_build(g, h.imb, c, g->ieps[0] ? g->ieps : g->eps);
        hyst_deltas(st, ep, h, mdc_ow[c], mdc_go[c], d_ow[c], d_go[c]);
    }
}

void oracle_pvt(const opmgpu_tables* t, int which, int n, const double* p, const double* r,
                const int8_t* saturated, const int32_t* pvtnum, double* out)
{
    for (int i = 0; i < n; ++i) {
        const int reg = pvtnum ? pvtnum[i] : 0;
        const bool sat = saturated ? saturated[i] != 0 : true;
        const double rr = r ? r[i] : 0.0;
        PvtOut o;
        switch (which) {
        case 0: o = b_wat(t, reg, p[i]); break;
        case 1: o = b_oil(t, reg, p[i], rr, sat); break;
        case 2: o = b_gas(t, reg, p[i], rr, sat); break;
        case 3: o = mu_wat(t, reg, p[i]); break;
        case 4: o = mu_oil(t, reg, p[i], rr, sat); break;
        case 5: o = mu_gas(t, reg, p[i], rr, sat); break;
        case 6: o = rs_sat(t, reg, p[i]); break;
        default: o = rv_sat(t, reg, p[i]); break;
        }
        out[3 * i] = o.v; out[3 * i + 1] = o.dp; out[3 * i + 2] = o.dr;
    }
}

void oracle_cell_props(const opmgpu_grid* g, const opmgpu_tables* t, const double* p, const double* sat,
                       const double* rs, const double* rv, const int8_t* hc, double* out)
{
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int c = 0; c < g->nc; ++c) {
        const CellQ q = cell_eval(g, t, c, p[c], sat[3 * c], sat[3 * c + 2], rs[c], rv[c], hc[c], true);
        const D3* list[ORACLE_NPROP] = { &q.pw, &q.po, &q.pg, &q.b[0], &q.b[1], &q.b[2], &q.mu[0], &q.mu[1], &q.mu[2],
                                         &q.kr[0], &q.kr[1], &q.kr[2], &q.rho[0], &q.rho[1], &q.rho[2],
                                         &q.mob[0], &q.mob[1], &q.mob[2], &q.rs, &q.rv, &q.accum[0], &q.accum[1], &q.accum[2] };
        double* o = out + size_t(c) * ORACLE_NPROP * 4;
        for (int k = 0; k < ORACLE_NPROP; ++k) { o[4 * k] = list[k]->v; for (int d = 0; d < 3; ++d) o[4 * k + 1 + d] = list[k]->d[d]; }
    }
}

int oracle_pattern(const opmgpu_grid* g, int nw, const int32_t* well_connpos, const int32_t* well_cells,
                   int32_t* rowptr, int32_t* col)
{
    const int nc = g->nc;
    std::vector<std::vector<int> > adj(nc);
    for (int c = 0; c < nc; ++c) adj[c].push_back(c);
    for (int f = 0; f < g->nconn; ++f) {
        const int a = g->conn_cells[2 * f], b = g->conn_cells[2 * f + 1];
        adj[a].push_back(b); adj[b].push_back(a);
    }
    for (int w = 0; w < nw; ++w)
        for (int i = well_connpos[w]; i < well_connpos[w + 1]; ++i)
            for (int j = well_connpos[w]; j < well_connpos[w + 1]; ++j)
                adj[well_cells[i]].push_back(well_cells[j]);
    int nnz = 0;
    for (int c = 0; c < nc; ++c) {
        std::sort(adj[c].begin(), adj[c].end());
        adj[c].erase(std::unique(adj[c].begin(), adj[c].end()), adj[c].end());
        if (rowptr) { rowptr[c] = nnz; for (size_t k = 0; k < adj[c].size(); ++k) col[nnz + k] = adj[c][k]; }
        nnz += int(adj[c].size());
    }
    if (rowptr) rowptr[nc] = nnz;
    return nnz;
}

void oracle_assemble(const opmgpu_grid* g, const opmgpu_tables* t, double dt, int initial,
                     const double* p, const double* sat, const double* rs, const double* rv, const int8_t* hc,
                     const double* scale3, double* accum0, const int32_t* rowptr, const int32_t* col,
                     double* r, double* val9, double* binv)
{
    const int nc = g->nc;
    std::vector<CellQ> Q(nc);
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int c = 0; c < nc; ++c)
        Q[c] = cell_eval(g, t, c, p[c], sat[3 * c], sat[3 * c + 2], rs[c], rv[c], hc[c], true);
    if (initial) {
        // makeConstantState + computeAccum(state0, 0), BlackoilModelBase_impl.hpp:797-805:
        // state0 is the SAME state with derivatives dropped => the accumulation values.
        for (int c = 0; c < nc; ++c) for (int a = 0; a < 3; ++a) accum0[size_t(a) * nc + c] = Q[c].accum[a].v;
    }
    std::fill(val9, val9 + size_t(rowptr[nc]) * 9, 0.0);
    // accumulation term, :879-881 (pvdt_ = pv / dt, :222-232)
    for (int c = 0; c < nc; ++c) {
        const double pvdt = g->pv[c] / dt;
        const int sd = find_slot(rowptr, col, c, c);
        for (int a = 0; a < 3; ++a) {
            r[size_t(a) * nc + c] = pvdt * (Q[c].accum[a].v - accum0[size_t(a) * nc + c]);
            for (int k = 0; k < 3; ++k) val9[size_t(sd) * 9 + 3 * a + k] = pvdt * Q[c].accum[a].d[k];
            if (binv) binv[size_t(a) * nc + c] = 1.0 / Q[c].b[a].v;
        }
    }
    // flux terms: computeMassFlux :1484-1512, rs/rv cross terms :889-906, div = ngrad^T
    for (int f = 0; f < g->nconn; ++f) {
        const int c1 = g->conn_cells[2 * f], c2 = g->conn_cells[2 * f + 1];
        const double T = g->trans[f];
        const double dz = g->z[c1] - g->z[c2];
        const CellQ& q1 = Q[c1]; const CellQ& q2 = Q[c2];
        const D3* ph1[3] = { &q1.pw, &q1.po, &q1.pg }; const D3* ph2[3] = { &q2.pw, &q2.po, &q2.pg };
        D6 F[3]; int up[3];
        for (int a = 0; a < 3; ++a) {
            const D6 rhoavg = 0.5 * lift(q1.rho[a], 0) + 0.5 * lift(q2.rho[a], 1);
            D6 dh = (lift(*ph1[a], 0) - lift(*ph2[a], 1)) - g->gravity * (rhoavg * dz);
            if (g->thpres) {          // applyThresholdPressures, :1518-1545
                const double thp = g->thpres[f];
                const double high = (std::fabs(dh.v) >= thp) ? 1.0 : 0.0;
                dh = high * (dh - sgn(dh.v) * thp);
            }
            up[a] = (dh.v >= 0.0) ? 0 : 1;      // UpwindSelector, AutoDiffHelpers.hpp:212
            const CellQ& qu = up[a] == 0 ? q1 : q2;
            const D6 bmob = lift(qu.b[a] * qu.mob[a], up[a]);
            F[a] = bmob * (T * dh);
        }
        const D6 rs_face = up[1] == 0 ? lift(q1.rs, 0) : lift(q2.rs, 1);
        const D6 rv_face = up[2] == 0 ? lift(q1.rv, 0) : lift(q2.rv, 1);
        D6 G[3];
        G[0] = F[0];
        G[1] = F[1] + rv_face * F[2];
        G[2] = F[2] + rs_face * F[1];
        const int s11 = find_slot(rowptr, col, c1, c1), s12 = find_slot(rowptr, col, c1, c2);
        const int s21 = find_slot(rowptr, col, c2, c1), s22 = find_slot(rowptr, col, c2, c2);
        for (int a = 0; a < 3; ++a) {
            r[size_t(a) * nc + c1] += G[a].v;
            r[size_t(a) * nc + c2] -= G[a].v;
            for (int k = 0; k < 3; ++k) {
                val9[size_t(s11) * 9 + 3 * a + k] += G[a].d[k];
                val9[size_t(s12) * 9 + 3 * a + k] += G[a].d[3 + k];
                val9[size_t(s21) * 9 + 3 * a + k] -= G[a].d[k];
                val9[size_t(s22) * 9 + 3 * a + k] -= G[a].d[3 + k];
            }
        }
    }
    // matbalscale, NewtonIterationBlackoilInterleaved.cpp:234-236 (Jacobian rows only; r stays unscaled)
    if (scale3)
        for (size_t s = 0; s < size_t(rowptr[nc]); ++s)
            for (int a = 0; a < 3; ++a) for (int k = 0; k < 3; ++k) val9[s * 9 + 3 * a + k] *= scale3[a];
}

int oracle_convergence(const opmgpu_grid* g, const opmgpu_params* prm, double dt, const double* r, const double* binv,
                       double* B_avg3, double* CNV3, double* MB3, double* linf3, int* converged)
{
    const int nc = g->nc;
    double pvsum = 0.0;
    for (int c = 0; c < nc; ++c) pvsum += g->pv[c];
    bool conv = true; int status = OPMGPU_OK;
    for (int a = 0; a < 3; ++a) {
        double bsum = 0.0, maxc = 0.0, rsum = 0.0, linf = 0.0; bool first = true;
        for (int c = 0; c < nc; ++c) {
            const double R = r[size_t(a) * nc + c];
            bsum += binv[size_t(a) * nc + c];
            const double tv = std::fabs(R) / g->pv[c];
            if (first || tv > maxc || tv != tv) { maxc = tv; first = false; }     // maxCoeff keeps NaN visible
            rsum += R;
            linf = std::max(linf, std::fabs(R));
            if (!std::isfinite(R)) status = OPMGPU_ENUMERICAL;                    // :1562-1566
        }
        const double B = bsum / nc;
        B_avg3[a] = B; CNV3[a] = B * dt * maxc; MB3[a] = std::fabs(B * rsum) * dt / pvsum; linf3[a] = linf;
        conv = conv && (MB3[a] < prm->tolerance_mb) && (CNV3[a] < prm->tolerance_cnv);
        if (std::isnan(MB3[a]) || std::isnan(CNV3[a])) status = OPMGPU_ENUMERICAL;                        // :1828-1836
        if (MB3[a] > prm->max_residual_allowed || CNV3[a] > prm->max_residual_allowed) status = OPMGPU_ENUMERICAL; // :1837-1845
    }
    if (converged) *converged = conv ? 1 : 0;
    return status;
}

void oracle_update_state(const opmgpu_grid* g, const opmgpu_tables* t, const opmgpu_params* prm, const double* dx,
                         double* p, double* sat, double* rs, double* rv, int8_t* hc)
{
    const int nc = g->nc;
    const double eps = std::sqrt(std::numeric_limits<double>::epsilon());
    for (int c = 0; c < nc; ++c) {
        const int preg = g->pvtnum ? g->pvtnum[c] : 0;
        const SatTab st = sat_tab(t, g->satnum ? g->satnum[c] : 0);
        const bool isSg = hc[c] == OPMGPU_HC_GAS_AND_OIL, isRs = hc[c] == OPMGPU_HC_OIL_ONLY, isRv = hc[c] == OPMGPU_HC_GAS_ONLY;
        const double dp = dx[c], dsw = dx[nc + c], dxv = dx[2 * size_t(nc) + c];
        // pressure, :1177-1183
        const double p_old = p[c];
        const double dp_lim = sgn(dp) * std::min(std::fabs(dp), prm->dp_max_rel * std::fabs(p_old));
        const double pn = std::max(p_old - dp_lim, 0.0);
        // saturations, :1185-1230
        const double sw_old = sat[3 * c], so_old = sat[3 * c + 1], sg_old = sat[3 * c + 2];
        const double dsg = (isSg ? dxv : 0.0) - (isRv ? dsw : 0.0);
        const double dso = -dsw - dsg;
        const double maxVal = std::max(std::fabs(dso), std::max(std::fabs(dsg), std::fabs(dsw)));
        const double step = std::min(prm->ds_max / maxVal, 1.0);
        double sw = sw_old - step * dsw, sg = sg_old - step * dsg, so = so_old - step * dso;
        // negative saturation fixes in the order g, o, w, :1232-1271
        if (sg < 0) { sw = sw / (1 - sg); so = so / (1 - sg); sg = 0; }
        if (so < 0) { sw = sw / (1 - so); sg = sg / (1 - so); so = 0; }
        if (sw < 0) { so = so / (1 - sw); sg = sg / (1 - sw); sw = 0; }
        // rs / rv, :1273-1290
        const double rs_old = rs[c], rv_old = rv[c];
        double rsn = rs_old, rvn = rv_old;
        if (t->has_disgas) {
            const double drs = isRs ? dxv : 0.0;
            const double lim = sgn(drs) * std::min(std::fabs(drs), std::max(std::fabs(rs_old) * prm->dr_max_rel, 1.0));
            rsn = std::max(rs_old - lim, 0.0);
        }
        if (t->has_vapoil) {
            const double drv = isRv ? dxv : 0.0;
            const double lim = sgn(drv) * std::min(std::fabs(drv), std::max(std::fabs(rv_old) * prm->dr_max_rel, 1e-3));
            rvn = std::max(rv_old - lim, 0.0);
        }
        // phase-state switching, :1292-1356
        const double so_max = g_so_max ? g_so_max[c] : 0.0;
        const bool watOnly = sw > (1 - eps);
        int hcn = OPMGPU_HC_GAS_AND_OIL;
        if (t->has_disgas) {
            double f0, f1, df;
            vap_factor(t->vap2, so_old, so_max, f0, df);
            vap_factor(t->vap2, so, so_max, f1, df);
            const double rsSat0 = f0 * rs_sat(t, preg, p_old).v;          // fluidRsSat(p_old, so_old), :1301
            const double rsSat = f1 * rs_sat(t, preg, pn).v;              // fluidRsSat(p, so), :1302
            const bool hasGas = (sg > 0 && !isRs);
            const bool gasVaporized = ((rsn > rsSat * (1 + eps) && isRs) && (rs_old > rsSat0 * (1 - eps)));
            const bool useSg = watOnly || hasGas || gasVaporized;
            if (useSg) { rsn = rsSat; if (watOnly) { so = 0; sg = 0; rsn = 0; } }
            else hcn = OPMGPU_HC_OIL_ONLY;
        }
        if (t->has_vapoil) {
            // computeGasPressure, :1466-1480 (old saturations for the old pressure, new for the new)
            D3 a, b;
            const Eps ep = eps_for_cell(g, st, c);
            cappress3(st, ep, D3(sw_old), D3(sg_old), a, b);
            const double pg_old = p_old + b.v;
            cappress3(st, ep, D3(sw), D3(sg), a, b);
            const double pg_new = pn + b.v;
            double f0, f1, df;
            vap_factor(t->vap1, so_old, so_max, f0, df);
            vap_factor(t->vap1, so, so_max, f1, df);
            const double rvSat0 = f0 * rv_sat(t, preg, pg_old).v;         // :1332
            const double rvSat = f1 * rv_sat(t, preg, pg_new).v;          // :1333 (so possibly zeroed by the rs block above)
            const bool hasOil = (so > 0 && !isRv);
            const bool oilCondensed = ((rvn > rvSat * (1 + eps) && isRv) && (rv_old > rvSat0 * (1 - eps)));
            const bool useSg = watOnly || hasOil || oilCondensed;
            if (useSg) { rvn = rvSat; if (watOnly) { so = 0; sg = 0; rvn = 0; } }
            else hcn = OPMGPU_HC_GAS_ONLY;
        }
        p[c] = pn; sat[3 * c] = sw; sat[3 * c + 1] = so; sat[3 * c + 2] = sg;
        if (t->has_disgas) rs[c] = rsn;
        if (t->has_vapoil) rv[c] = rvn;
        hc[c] = int8_t(hcn);
    }
}

void oracle_set_sat_oil_max(const double* so_max) { g_so_max = so_max; }

void oracle_spmv(int nb, const int32_t* rowptr, const int32_t* col, const double* val9, const double* x3, double* y3, int sp)
{
    if (!sp) { spmv_t<double>(nb, rowptr, col, val9, x3, y3); return; }
    std::vector<float> A(size_t(rowptr[nb]) * 9), x(3 * size_t(nb)), y(3 * size_t(nb));
    for (size_t k = 0; k < A.size(); ++k) A[k] = float(val9[k]);
    for (size_t k = 0; k < x.size(); ++k) x[k] = float(x3[k]);
    spmv_t<float>(nb, rowptr, col, A.data(), x.data(), y.data());
    for (size_t k = 0; k < y.size(); ++k) y3[k] = y[k];
}

int oracle_ilu0(int nb, const int32_t* rowptr, const int32_t* col, const double* val9, const int32_t* position, int sp, double* lu9)
{
    const size_t n9 = size_t(rowptr[nb]) * 9;
    if (sp) { IluData<float> D; const int st = ilu0_setup<float>(nb, rowptr, col, val9, position, D); if (st) return st; for (size_t k = 0; k < n9; ++k) lu9[k] = D.lu[k]; }
    else { IluData<double> D; const int st = ilu0_setup<double>(nb, rowptr, col, val9, position, D); if (st) return st; for (size_t k = 0; k < n9; ++k) lu9[k] = D.lu[k]; }
    return OPMGPU_OK;
}

void oracle_ilu0_apply(int nb, const int32_t* rowptr, const int32_t* col, const double* lu9, const int32_t* position,
                       double relax, int sp, const double* d3, double* v3)
{
    const int n = 3 * nb;
    if (sp) {
        IluData<float> D; ilu_from_factors<float>(nb, rowptr, col, lu9, position, D);
        std::vector<float> d(n), v(n, 0.f);
        for (int i = 0; i < n; ++i) d[i] = float(d3[i]);
        ilu0_apply<float>(D, col, float(relax), d.data(), v.data());
        for (int i = 0; i < n; ++i) v3[i] = v[i];
    } else {
        IluData<double> D; ilu_from_factors<double>(nb, rowptr, col, lu9, position, D);
        std::vector<double> v(n, 0.0);
        ilu0_apply<double>(D, col, relax, d3, v.data());
        for (int i = 0; i < n; ++i) v3[i] = v[i];
    }
}

int oracle_bicgstab_ilu0(int nb, const int32_t* rowptr, const int32_t* col, const double* val9, const double* rhs3,
                         const int32_t* position, const opmgpu_params* prm, int sp, double* x3, int* iters,
                         double* reduction, double* hist, int nhist, int* nhist_out)
{
    if (prm->newton_use_gmres) {        // ISTLSolver.hpp:257-264
        if (nhist_out) *nhist_out = 0;
        return sp ? gmres_t<float>(nb, rowptr, col, val9, rhs3, position, prm, x3, iters, reduction)
                  : gmres_t<double>(nb, rowptr, col, val9, rhs3, position, prm, x3, iters, reduction);
    }
    if (sp) return bicgstab_t<float>(nb, rowptr, col, val9, rhs3, position, prm, x3, iters, reduction, hist, nhist, nhist_out);
    return bicgstab_t<double>(nb, rowptr, col, val9, rhs3, position, prm, x3, iters, reduction, hist, nhist, nhist_out);
}

} // extern "C"
