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
// Same documented simplifications as the host model (opmgpu/wells.py): no control switching, no explicit well pre-solve,
// no THP/VFP/groups, connection densities from the perforated cells' own b / rs / rv.
#include "blackoil.hpp"

#include <algorithm>
#include <cmath>

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

struct WellArgs {
    int nbp;
    const int32_t *connpos, *perf_row, *type, *allow_cf, *ctrl_type;
    const double *WI, *comp_frac, *ctrl_target, *ctrl_distr, *depth_ref, *z_perf, *surf_dens_perf;
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

// computeWellConnectionPressures: WellDensitySegmented::computeConnectionDensities + computeConnectionPressureDelta, one thread per well
__global__ void k_well_cdp(int nw, WellArgs A, double gravity)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    const int lo = A.connpos[w], hi = A.connpos[w + 1];
    // q_out[perf] = flow out of the segment above perforation perf = sum_{k >= perf} (-rate_k) (the reference fills it bottom to
    // top, WellDensitySegmented.cpp:83-95); here: total at the top first, then q_out[perf + 1] = q_out[perf] + rate[perf] going down.
    double q[3] = { 0.0, 0.0, 0.0 };
    for (int perf = hi - 1; perf >= lo; --perf) for (int a = 0; a < 3; ++a) q[a] -= A.perf_rates[3 * perf + a];
    double run = 0.0;
    for (int perf = lo; perf < hi; ++perf) {
        const double* pp = A.perf + long(perf) * OPMGPU_PERF_K;
        const double tot = q[0] + q[1] + q[2];
        double mix[3], x[3];
        for (int a = 0; a < 3; ++a) { mix[a] = tot != 0.0 ? fabs(q[a] / tot) : A.comp_frac[3 * w + a]; x[a] = mix[a]; }
        const double rsmax = pp[4 * 1], rvmax = pp[4 * 2];
        double rs = 0.0, rv = 0.0;
        if (mix[1] > 0.0) rs = fmin(mix[2] / mix[1], rsmax);
        if (mix[2] > 0.0) rv = fmin(mix[1] / mix[2], rvmax);
        if (rs != 0.0) x[2] = (mix[2] - mix[1] * rs) / (1.0 - rs * rv);
        if (rv != 0.0) x[1] = (mix[1] - mix[2] * rv) / (1.0 - rs * rv);
        const double volrat = x[0] / pp[4 * 3] + x[1] / pp[4 * 4] + x[2] / pp[4 * 5];
        const double* sd = A.surf_dens_perf + 3 * long(perf);
        const double dens = (sd[0] * mix[0] + sd[1] * mix[1] + sd[2] * mix[2]) / volrat;
        const double z_above = perf == lo ? A.depth_ref[w] : A.z_perf[perf - 1];
        run += (A.z_perf[perf] - z_above) * dens * gravity;
        A.cdp[perf] = run;
        for (int a = 0; a < 3; ++a) q[a] += A.perf_rates[3 * perf + a];
    }
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

// well equations + reduced (Schur) contributions of one well per workgroup
template <class MS>
__global__ __launch_bounds__(kBlock) void k_well_assemble(WellArgs A, const int32_t* __restrict__ slice_ptr, const int16_t* __restrict__ nlower,
                                                          double s0, double s1, double s2, double* __restrict__ R, MS* __restrict__ Amat,
                                                          double* __restrict__ rhs_extra, int32_t* __restrict__ flags)
{
    __shared__ double sm[64];
    __shared__ int any_flag[2];
    __shared__ double wl[64];         // well-level values shared by all threads
    const int w = blockIdx.x, tid = threadIdx.x;
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
        // addWellControlEq: dead well -> sum of rates; BHP; SURFACE_RATE
        if (!alive) { E[3] = qs[0] + qs[1] + qs[2]; D[12] = 1.0; D[13] = 1.0; D[14] = 1.0; D[15] = 0.0; }
        else if (A.ctrl_type[w] == 0) { E[3] = bhp - A.ctrl_target[w]; D[12] = 0.0; D[13] = 0.0; D[14] = 0.0; D[15] = 1.0; }
        else {
            const double* ds = A.ctrl_distr + 3 * w;
            E[3] = ds[0] * qs[0] + ds[1] * qs[1] + ds[2] * qs[2] - A.ctrl_target[w];
            D[12] = ds[0]; D[13] = ds[1]; D[14] = ds[2]; D[15] = 0.0;
        }
        double Di[16];
        if (!inv4(D, Di)) { atomicOr(flags, 2); for (int k = 0; k < 16; ++k) Di[k] = 0.0; }
        for (int k = 0; k < 16; ++k) { wl[20 + k] = Di[k]; A.Dinv[16 * w + k] = Di[k]; }
        for (int k = 0; k < 4; ++k) { wl[36 + k] = E[k]; A.wellE[4 * w + k] = E[k]; }
        for (int k = 0; k < 9; ++k) wl[40 + k] = Ms[k];
    }
    __syncthreads();
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

// recoverVariable + updateWellState: dy = D^-1 (E - sum_j C_j dx_j); qs -= dy[0..2]; bhp -= sign(d) min(|d|, |bhp| dbhp_max_rel)
__global__ __launch_bounds__(kBlock) void k_well_update(WellArgs A, const double* __restrict__ dx, double relax, double dbhp_max_rel)
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
        double dy[4];
        for (int k = 0; k < 4; ++k) { dy[k] = 0.0; for (int c = 0; c < 4; ++c) dy[k] += Di[4 * k + c] * (E[c] - acc[c]); dy[k] *= relax; }
        for (int a = 0; a < 3; ++a) A.wstate[4 * w + a] -= dy[a];
        const double d = dy[3], bhp = A.wstate[4 * w + 3];
        const double sg = d > 0.0 ? 1.0 : (d < 0.0 ? -1.0 : 0.0);
        A.wstate[4 * w + 3] = bhp - sg * fmin(fabs(d), fabs(bhp) * dbhp_max_rel);
    }
}

} // namespace

// ------------------------------------------------------------------------------------------ host side
struct BlackoilDevice::WellsDev {
    int nw = 0;
    DevArray<int32_t> connpos, perf_row, perf_well, perf_of_row, type, allow_cf, ctrl_type;
    DevArray<double> WI, comp_frac, ctrl_target, ctrl_distr, depth_ref, z_perf, surf_dens_perf;
    DevArray<double> wstate, cdp, perf_rates, perf_press, P, Q, Fsave, wellE, Dinv, t;
    DevArray<double> saved;         // snapshot for AdaptiveTimeStepping: wstate | cdp | perf_rates
    DevArray<int32_t> flags;
    std::vector<int32_t> h_connpos, h_cells;
    double dbhp_max_rel = 1.0;
    double* h_pinned = nullptr;
};

void BlackoilDevice::wells_free() { if (wd) { if (wd->h_pinned) (void)hipHostFree(wd->h_pinned); delete wd; wd = nullptr; } }

static WellArgs args_of(BlackoilDevice::WellsDev& W, int nbp, const double* perf)
{
    WellArgs A;
    A.nbp = nbp;
    A.connpos = W.connpos.p; A.perf_row = W.perf_row.p; A.type = W.type.p; A.allow_cf = W.allow_cf.p; A.ctrl_type = W.ctrl_type.p;
    A.WI = W.WI.p; A.comp_frac = W.comp_frac.p; A.ctrl_target = W.ctrl_target.p; A.ctrl_distr = W.ctrl_distr.p; A.depth_ref = W.depth_ref.p;
    A.z_perf = W.z_perf.p; A.surf_dens_perf = W.surf_dens_perf.p; A.perf = perf;
    A.wstate = W.wstate.p; A.cdp = W.cdp.p; A.perf_rates = W.perf_rates.p; A.perf_press = W.perf_press.p; A.P = W.P.p; A.Q = W.Q.p;
    A.Fsave = W.Fsave.p; A.wellE = W.wellE.p; A.Dinv = W.Dinv.p;
    return A;
}

int BlackoilDevice::set_device_wells(const opmgpu_wells* s)
{
    if (!s || s->nw < 0) return OPMGPU_EINVAL;
    if (s->nw == 0) { wells_free(); device_wells = false; ls.lowrank = LowRankOp(); return set_wells(0, nullptr, nullptr); }
    if (!s->well_connpos || !s->well_cells || !s->WI || !s->type || !s->depth_ref || !s->comp_frac || !s->ctrl_type || !s->ctrl_target) return OPMGPU_EINVAL;
    const int nw = s->nw, np = s->well_connpos[nw];
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
    upi(W.connpos, s->well_connpos, nw + 1, 0); upi(W.type, s->type, nw, 1); upi(W.allow_cf, s->allow_cf, nw, 1); upi(W.ctrl_type, s->ctrl_type, nw, 0);
    upd(W.WI, s->WI, np); upd(W.comp_frac, s->comp_frac, 3 * size_t(nw)); upd(W.ctrl_target, s->ctrl_target, nw);
    upd(W.ctrl_distr, s->ctrl_distr, 3 * size_t(nw)); upd(W.depth_ref, s->depth_ref, nw);
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
    W.flags.alloc(1); W.flags.zero(stream);
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
}

int BlackoilDevice::well_state_set(const double* bhp, const double* qs, const double* perf_rates)
{
    if (!wd || !bhp || !qs) return OPMGPU_EINVAL;
    WellsDev& W = *wd;
    std::vector<double> h(4 * size_t(W.nw));
    for (int w = 0; w < W.nw; ++w) { for (int a = 0; a < 3; ++a) h[4 * w + a] = qs[3 * w + a]; h[4 * w + 3] = bhp[w]; }
    W.wstate.upload(h, stream);
    if (perf_rates) W.perf_rates.upload(perf_rates, 3 * W.h_cells.size(), stream);
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

// called by assemble() after the reservoir kernels
void BlackoilDevice::wells_assemble(bool initial)
{
    if (!wd) return;
    WellsDev& W = *wd;
    const Plan& P = ls.plan;
    perf_props_device();
    WellArgs A = args_of(W, P.nbp, d_perf.p);
    if (initial)      // computeWellConnectionPressures: once per time step from the explicit state (BlackoilModelBase_impl.hpp:797-805)
        hipLaunchKernelGGL(k_well_cdp, dim3((W.nw + 63) / 64), dim3(64), 0, stream, W.nw, A, gravity);
    d_rhs_extra.zero(stream);
    const double* sc = prm.matbalscale;
    if (ls.matrix_is_float)
        hipLaunchKernelGGL((k_well_assemble<float>), dim3(W.nw), dim3(kBlock), 0, stream, A, ls.dp.slice_ptr.p, ls.dp.nlower.p, sc[0], sc[1], sc[2], d_R.p,
                           ls.matrix_f(), d_rhs_extra.p, W.flags.p);
    else
        hipLaunchKernelGGL((k_well_assemble<double>), dim3(W.nw), dim3(kBlock), 0, stream, A, ls.dp.slice_ptr.p, ls.dp.nlower.p, sc[0], sc[1], sc[2], d_R.p,
                           ls.matrix_d(), d_rhs_extra.p, W.flags.p);
    has_rhs_extra = true;
}

// well part of getConvergence (BlackoilModelBase_impl.hpp:1769-1779): max |flux equation| per phase, max |control equation|
int BlackoilDevice::well_convergence(double* flux3, double* ctrl)
{
    if (!wd && !ls.comm) return OPMGPU_EINVAL;
    double f[3] = { 0, 0, 0 }, c = 0.0;
    bool bad = false, singular = false;
    if (wd) {
        WellsDev& W = *wd;
        OPMGPU_HIP(hipMemcpyAsync(W.h_pinned, W.wellE.p, 4 * size_t(W.nw) * sizeof(double), hipMemcpyDeviceToHost, stream));
        int32_t fl = 0;
        OPMGPU_HIP(hipMemcpyAsync(&fl, W.flags.p, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        OPMGPU_HIP(hipStreamSynchronize(stream));
        for (int w = 0; w < W.nw; ++w) {
            for (int a = 0; a < 3; ++a) { const double e = std::fabs(W.h_pinned[4 * w + a]); if (!(e == e)) bad = true; f[a] = std::max(f[a], e); }
            const double e = std::fabs(W.h_pinned[4 * w + 3]); if (!(e == e)) bad = true; c = std::max(c, e);
        }
        if (fl & 2) { W.flags.zero(stream); singular = true; }
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

void BlackoilDevice::wells_update(double relax)
{
    if (!wd) return;
    WellsDev& W = *wd;
    WellArgs A = args_of(W, ls.plan.nbp, d_perf.p);
    hipLaunchKernelGGL(k_well_update, dim3(W.nw), dim3(kBlock), 0, stream, A, (const double*)d_dx.p, relax, W.dbhp_max_rel);
}

void BlackoilDevice::wells_save()
{
    if (!wd) return;
    WellsDev& W = *wd;
    const size_t nw4 = 4 * size_t(W.nw), np = W.h_cells.size();
    W.saved.alloc(nw4 + 4 * np);
    OPMGPU_HIP(hipMemcpyAsync(W.saved.p, W.wstate.p, nw4 * sizeof(double), hipMemcpyDeviceToDevice, stream));
    OPMGPU_HIP(hipMemcpyAsync(W.saved.p + nw4, W.cdp.p, np * sizeof(double), hipMemcpyDeviceToDevice, stream));
    OPMGPU_HIP(hipMemcpyAsync(W.saved.p + nw4 + np, W.perf_rates.p, 3 * np * sizeof(double), hipMemcpyDeviceToDevice, stream));
}
void BlackoilDevice::wells_restore()
{
    if (!wd || !wd->saved.p) return;
    WellsDev& W = *wd;
    const size_t nw4 = 4 * size_t(W.nw), np = W.h_cells.size();
    OPMGPU_HIP(hipMemcpyAsync(W.wstate.p, W.saved.p, nw4 * sizeof(double), hipMemcpyDeviceToDevice, stream));
    OPMGPU_HIP(hipMemcpyAsync(W.cdp.p, W.saved.p + nw4, np * sizeof(double), hipMemcpyDeviceToDevice, stream));
    OPMGPU_HIP(hipMemcpyAsync(W.perf_rates.p, W.saved.p + nw4 + np, 3 * np * sizeof(double), hipMemcpyDeviceToDevice, stream));
}

void BlackoilDevice::set_dbhp_max_rel(double v) { if (wd) wd->dbhp_max_rel = v; }

} // namespace opmgpu
