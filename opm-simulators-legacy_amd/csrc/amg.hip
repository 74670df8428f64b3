// amg.hip -- plain-aggregation AMG for the CPR pressure stage (see amg.hpp).
#include "amg.hpp"
#include <functional>

#include <string>
#include <utility>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include "linsolver.hpp"

namespace opmgpu {

namespace {

// ---------------------------------------------------------------- host: hierarchy setup
struct HostCsr {
    int n = 0;
    std::vector<int32_t> rowptr, col, dev;   // dev = device entry id of every csr entry
    std::vector<double> val;
};

// greedy strength-based aggregation (Vanek-style, as in dune-istl's aggregation AMG): a node whose strong
// neighbours are all free seeds an aggregate with them; leftovers join their strongest aggregated neighbour.
// npin: the last npin rows (well unknowns of the bordered pressure system) stay singletons on every level
int aggregate(const HostCsr& A, double theta, std::vector<int32_t>& agg, int npin = 0)
{
    const int n = A.n - npin;
    agg.assign(A.n, -1);
    std::vector<double> maxoff(A.n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int s = A.rowptr[i]; s < A.rowptr[i + 1]; ++s)
            if (A.col[s] != i && A.col[s] < n) maxoff[i] = std::max(maxoff[i], std::fabs(A.val[s]));
    auto strong = [&](int i, int s) { return A.col[s] != i && A.col[s] < n && std::fabs(A.val[s]) >= theta * maxoff[i] && maxoff[i] > 0.0; };
    int na = 0;
    for (int i = 0; i < n; ++i) {
        if (agg[i] >= 0) continue;
        bool free_nb = true;
        for (int s = A.rowptr[i]; s < A.rowptr[i + 1] && free_nb; ++s) if (strong(i, s) && agg[A.col[s]] >= 0) free_nb = false;
        if (!free_nb) continue;
        agg[i] = na;
        for (int s = A.rowptr[i]; s < A.rowptr[i + 1]; ++s) if (strong(i, s)) agg[A.col[s]] = na;
        ++na;
    }
    for (int i = 0; i < n; ++i) {
        if (agg[i] >= 0) continue;
        double best = -1.0; int bj = -1;
        for (int s = A.rowptr[i]; s < A.rowptr[i + 1]; ++s)
            if (strong(i, s) && agg[A.col[s]] >= 0 && std::fabs(A.val[s]) > best) { best = std::fabs(A.val[s]); bj = A.col[s]; }
        agg[i] = bj >= 0 ? agg[bj] : na++;
    }
    for (int i = n; i < A.n; ++i) agg[i] = na++;
    return na;
}

// Galerkin product with piecewise-constant prolongation: Ac(I,J) = sum_{i in I, j in J} A(i,j).
// Also returns, for every fine csr entry, the coarse csr entry it is added to.
void coarsen(const HostCsr& A, const std::vector<int32_t>& agg, int na, HostCsr& C, std::vector<int32_t>& coarse_of_fine,
             std::vector<int32_t>& agg_ptr, std::vector<int32_t>& agg_rows)
{
    agg_ptr.assign(na + 1, 0);
    for (int i = 0; i < A.n; ++i) agg_ptr[agg[i] + 1]++;
    for (int I = 0; I < na; ++I) agg_ptr[I + 1] += agg_ptr[I];
    agg_rows.resize(A.n);
    { std::vector<int32_t> fill(agg_ptr.begin(), agg_ptr.end() - 1); for (int i = 0; i < A.n; ++i) agg_rows[fill[agg[i]]++] = i; }
    C.n = na; C.rowptr.assign(na + 1, 0); C.col.clear(); C.val.clear();
    coarse_of_fine.assign(A.col.size(), -1);
    std::vector<int32_t> marker(na, -1);
    for (int I = 0; I < na; ++I) {
        const int start = int(C.col.size());
        // diagonal first
        marker[I] = start; C.col.push_back(I); C.val.push_back(0.0);
        for (int q = agg_ptr[I]; q < agg_ptr[I + 1]; ++q) {
            const int i = agg_rows[q];
            for (int s = A.rowptr[i]; s < A.rowptr[i + 1]; ++s) {
                const int J = agg[A.col[s]];
                if (marker[J] < start) { marker[J] = int(C.col.size()); C.col.push_back(J); C.val.push_back(0.0); }
                C.val[marker[J]] += A.val[s];
                coarse_of_fine[s] = marker[J];
            }
        }
        C.rowptr[I + 1] = int(C.col.size());
    }
}

// scalar SELL-64 layout of a csr matrix; fills C.dev
void to_sell(HostCsr& C, std::vector<int32_t>& slice_ptr, std::vector<int32_t>& sell_col, std::vector<int32_t>& diag_entry, int& nentries)
{
    const int n = C.n, ns = (n + 63) / 64;
    slice_ptr.assign(ns + 1, 0);
    for (int s = 0; s < ns; ++s) {
        int w = 0;
        for (int r = s * 64; r < std::min(n, s * 64 + 64); ++r) w = std::max(w, C.rowptr[r + 1] - C.rowptr[r]);
        slice_ptr[s + 1] = slice_ptr[s] + w;
    }
    nentries = slice_ptr[ns] * 64;
    sell_col.assign(nentries, 0);
    diag_entry.assign(n, 0);
    C.dev.assign(C.col.size(), -1);
    for (int r = 0; r < ns * 64; ++r) {
        const int base = slice_ptr[r >> 6], w = slice_ptr[(r >> 6) + 1] - base;
        int len = 0;
        if (r < n) {
            len = C.rowptr[r + 1] - C.rowptr[r];
            for (int k = 0; k < len; ++k) {
                const int e = (base + k) * 64 + (r & 63);
                sell_col[e] = C.col[C.rowptr[r] + k]; C.dev[C.rowptr[r] + k] = e;
                if (C.col[C.rowptr[r] + k] == r) diag_entry[r] = e;
            }
        }
        for (int k = len; k < w; ++k) sell_col[(base + k) * 64 + (r & 63)] = std::min(r, n - 1);
    }
}

} // namespace

// ---------------------------------------------------------------- device kernels
constexpr int galerkin_unroll(int lpe) { return lpe == 1 ? 8 : (lpe <= 8 ? 2 : 1); }
// Galerkin sums: coarse device slot e = sum of the fine entries cidx[cptr[e] .. cptr[e+1]) in a fixed order (deterministic).  The
// work list is in the coarse level's SELL order (setup()): a wavefront's lanes are 64 neighbouring coarse rows at the same position of
// the row -- all diagonal entries (~33 contributions at 100^3) or all off-diagonal ones (~4) --, so they loop equally often, and
// the stores are contiguous.  (In csr order every ~13th lane was a diagonal and each wave ran at its pace, with a scattered
// 4-byte store per thread: 79 us for level 0 -> 1, whose lists are 56 MB.)  LPE lanes share one entry (strided, butterfly-summed);
// a diagonal entry also writes the coarse level's inverse diagonal (cdiag[e] = its row, else -1): one launch less per level.
// What remains (PMC, level 0 -> 1 at 100^3: 459 MB fetched for 60 MB of lists and values): the gathers.  Level 0 is SELL-64 in the ILU's
// two-colour row order, so the ~9 rows of an aggregate lie in ~7 row ranges and each of a row's entries in another 128-byte line; a
// gather instruction touches 23 lines on average (measured on the dumped lists) and the fine values cost 35 of the kernel's 60 us.
template <class S, int LPE>
__global__ __launch_bounds__(kBlock) void k_amg_galerkin(int xm, int nce, const int32_t* __restrict__ cptr, const int32_t* __restrict__ cidx,
                                                         const int32_t* __restrict__ cdiag,
                                                         const S* __restrict__ fine, S* __restrict__ coarse, S* __restrict__ dinv_coarse)
{
    constexpr int U = galerkin_unroll(LPE);
    // XCD-contiguous chunks (common.hpp): the ~13 positions of one coarse slice gather from the SAME fine rows and are 13 consecutive
    // wavefronts = 3-4 workgroups; dealt round-robin they land in 3-4 different L2s and every fine line is fetched that often
    // (PMC at 100^3, level 0 -> 1: 459 MB fetched for 60 MB of lists and values, L2 hit rate 0.42)
    const int nchunks = (int)(((long)nce * LPE + kBlock - 1) / kBlock);
    const int ch = xcd_first(nchunks, xm);
    if (ch >= xcd_end(nchunks, xm)) return;
    const int t = ch * kBlock + threadIdx.x, e = t / LPE, l = t % LPE;
    const bool live = e < nce;
    double s = 0.0;
    const int q0 = live ? cptr[e] : 0, q1 = live ? cptr[e + 1] : 0;
    for (int q = q0 + l; q < q1; q += U * LPE) {          // index loads, then gathers, of a batch in flight together
        int id[U]; S v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) id[u] = q + u * LPE < q1 ? cidx[q + u * LPE] : -1;
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = id[u] >= 0 ? fine[id[u]] : S(0);
#pragma unroll
        for (int u = 0; u < U; ++u) s += double(v[u]);
    }
#pragma unroll
    for (int off = LPE / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (live && l == 0) {
        const S c = S(s);
        coarse[e] = c;
        const int r = cdiag[e];
        if (r >= 0) dinv_coarse[r] = c != S(0) ? S(1) / c : S(0);
    }
}
template <class S>
__global__ __launch_bounds__(kBlock) void k_amg_dinv(int n, const int32_t* __restrict__ diag_entry, const S* __restrict__ val, S* __restrict__ dinv)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const S d = val[diag_entry[i]];
    dinv[i] = d != S(0) ? S(1) / d : S(0);
}
// first pre-smoothing sweep from a zero guess: x = omega D^-1 b
template <class S>
__global__ __launch_bounds__(kBlock) void k_amg_smooth0(int n, S omega, const S* __restrict__ dinv, const S* __restrict__ b, S* __restrict__ x,
                                                        const SolveCtl* __restrict__ ctl)
{
    if (ctl && ctl->done) return;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    x[i] = omega * dinv[i] * b[i];
}
// acc -= sum_k val[k] * xfun(col[k]) over one SELL row.  The slots are walked in predicated batches of 8 so that all
// index loads, then all gathers of a batch are in flight together: with 4-byte scalars a row is only ~7 x 12 bytes and a
// plain dependent loop (col -> x -> fma) is latency bound at a fraction of the HBM rate.
template <class S, class XF>
__device__ __forceinline__ S sell_row_dot(const S* __restrict__ v, const int32_t* __restrict__ c, int width, XF xfun)
{
    S acc = 0;
    for (int k0 = 0; k0 < width; k0 += 8) {
        int cc[8]; S vv[8], xx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const bool ok = k0 + u < width; cc[u] = ok ? c[(k0 + u) * 64] : -1; vv[u] = ok ? v[(k0 + u) * 64] : S(0); }
#pragma unroll
        for (int u = 0; u < 8; ++u) xx[u] = cc[u] >= 0 ? xfun(cc[u]) : S(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += vv[u] * xx[u];
    }
    return acc;
}

// Border of a level-0 operator (amg.hpp): the row kernels get `gcells` workgroups for the cell rows plus one workgroup per well.
template <class S>
struct Border {
    int nw = 0, n = 0, gcells = 0;
    const int32_t *connpos = nullptr, *perf_row = nullptr, *perf_of_row = nullptr, *perf_well = nullptr;
    const S *bcol = nullptr, *crow = nullptr, *dw = nullptr;
};
// cell row: the column entry towards its well's unknown
template <class S, class XF>
__device__ __forceinline__ S border_cell(const Border<S>& B, int row, XF xfun)
{
    if (!B.nw) return S(0);
    const int j = B.perf_of_row[row];
    return j >= 0 ? B.bcol[j] * xfun(B.n + B.perf_well[j]) : S(0);
}
// well row k by one workgroup: sum_j crow[j] xfun(perf_row[j]) + dw[k] xfun(n + k); result valid in thread 0 (fixed order: deterministic)
template <class S, class XF>
__device__ double border_well(const Border<S>& B, int k, XF xfun)
{
    __shared__ double sm[4];
    double acc[1] = { 0.0 };
    for (int j = B.connpos[k] + threadIdx.x; j < B.connpos[k + 1]; j += kBlock) acc[0] += double(B.crow[j]) * double(xfun(B.perf_row[j]));
    block_sum<1>(acc, sm);
    return acc[0] + double(B.dw[k]) * double(xfun(B.n + k));
}

// pre-smoothing from a zero guess fused with the residual (small, launch-bound levels): x = omega D^-1 b ; r = b - A x
template <class S>
__global__ __launch_bounds__(kBlock) void k_amg_smooth0_residual(int n, const int32_t* __restrict__ slice_ptr, const int32_t* __restrict__ col,
                                                                 const S* __restrict__ val, const S* __restrict__ b, S omega, const S* __restrict__ dinv,
                                                                 S* __restrict__ x, S* __restrict__ r, const SolveCtl* __restrict__ ctl, Border<S> B = Border<S>())
{
    if (ctl && ctl->done) return;
    auto xf = [&](int j) { return omega * dinv[j] * b[j]; };
    if (B.nw && int(blockIdx.x) >= B.gcells) {
        const int k = blockIdx.x - B.gcells;
        const double ax = border_well(B, k, xf);
        if (threadIdx.x == 0) { x[n + k] = xf(n + k); r[n + k] = b[n + k] - S(ax); }
        return;
    }
    const int row = blockIdx.x * kBlock + threadIdx.x;
    if (row >= n) return;
    const int base = slice_ptr[row >> 6], width = slice_ptr[(row >> 6) + 1] - base, lane = row & 63;
    const S ax = sell_row_dot<S>(val + long(base) * 64 + lane, col + long(base) * 64 + lane, width, xf) + border_cell(B, row, xf);
    x[row] = omega * dinv[row] * b[row];
    r[row] = b[row] - ax;
}
// MODE 0: r = b - A x ;  MODE 1: xout = x + omega D^-1 (b - A x)   (damped Jacobi sweep)
// MODE 3: the same sweep on the prolongated iterate x + pdamp P xc, gathered on the fly (saves the prolongation launch)
template <class S, int MODE>
__global__ __launch_bounds__(kBlock) void k_amg_residual(int n, const int32_t* __restrict__ slice_ptr, const int32_t* __restrict__ col,
                                                         const S* __restrict__ val, const S* __restrict__ b, const S* __restrict__ x,
                                                         S omega, const S* __restrict__ dinv, S* __restrict__ out, const SolveCtl* __restrict__ ctl,
                                                         const int32_t* __restrict__ agg = nullptr, const S* __restrict__ xc = nullptr, S pdamp = S(0),
                                                         Border<S> B = Border<S>())
{
    if (ctl && ctl->done) return;
    if (B.nw && int(blockIdx.x) >= B.gcells) {          // a well row of the bordered level 0
        const int k = blockIdx.x - B.gcells, i = n + k;
        double ax;
        if (MODE == 3) ax = border_well(B, k, [&](int j) { return x[j] + pdamp * xc[agg[j]]; });
        else ax = border_well(B, k, [&](int j) { return x[j]; });
        if (threadIdx.x == 0) {
            const S acc = b[i] - S(ax);
            if (MODE == 3) out[i] = (x[i] + pdamp * xc[agg[i]]) + omega * dinv[i] * acc;
            else out[i] = MODE == 0 ? acc : x[i] + omega * dinv[i] * acc;
        }
        return;
    }
    const int row = blockIdx.x * kBlock + threadIdx.x;
    if (row >= n) return;
    const int base = slice_ptr[row >> 6], width = slice_ptr[(row >> 6) + 1] - base, lane = row & 63;
    if (MODE == 3) {
        auto xf = [&](int j) { return x[j] + pdamp * xc[agg[j]]; };
        const S acc = b[row] - sell_row_dot<S>(val + long(base) * 64 + lane, col + long(base) * 64 + lane, width, xf) - border_cell(B, row, xf);
        out[row] = (x[row] + pdamp * xc[agg[row]]) + omega * dinv[row] * acc;
        return;
    }
    auto xf = [&](int j) { return x[j]; };
    const S acc = b[row] - sell_row_dot<S>(val + long(base) * 64 + lane, col + long(base) * 64 + lane, width, xf) - border_cell(B, row, xf);
    out[row] = MODE == 0 ? acc : x[row] + omega * dinv[row] * acc;
}
// level 0 with a two-colour row order (the block plan's multicolour ordering of a grid stencil): Gauss-Seidel by colour.  Rows of one
// colour do not couple, so the rows [lo, hi) of a colour are updated IN PLACE from the other colour's current values -- the traffic of
// half a Jacobi sweep per colour, the smoothing of Gauss-Seidel.  MODE 1: x += D^-1 (b - A x) ; MODE 0: out = b - A x (residual rows)
template <class S, int MODE>
__global__ __launch_bounds__(kBlock) void k_amg_gs(int lo, int hi, const int32_t* __restrict__ slice_ptr, const int32_t* __restrict__ col,
                                                   const S* __restrict__ val, const S* __restrict__ b, const S* __restrict__ dinv,
                                                   S* __restrict__ x, S* __restrict__ out, const SolveCtl* __restrict__ ctl)
{
    if (ctl && ctl->done) return;
    const int row = lo + blockIdx.x * kBlock + threadIdx.x;
    if (row >= hi) return;
    const int base = slice_ptr[row >> 6], width = slice_ptr[(row >> 6) + 1] - base, lane = row & 63;
    const S* xr = x;
    const S acc = b[row] - sell_row_dot<S>(val + long(base) * 64 + lane, col + long(base) * 64 + lane, width, [&](int j) { return xr[j]; });
    if (MODE == 0) out[row] = acc; else x[row] = x[row] + dinv[row] * acc;
}
// small levels: one wavefront per row (aggregated stencils are 30-100 entries wide there; a single thread walking them
// is latency bound).  MODE 0: r = b - A x ; MODE 1: out = x + omega D^-1 (b - A x) ; MODE 2: x = omega D^-1 b, r = b - A x fused
template <class S, int MODE>
__global__ __launch_bounds__(kBlock) void k_amg_row_wave(int n, const int32_t* __restrict__ slice_ptr, const int32_t* __restrict__ col,
                                                         const S* __restrict__ val, const S* __restrict__ b, const S* __restrict__ x,
                                                         S omega, const S* __restrict__ dinv, S* __restrict__ out, S* __restrict__ xout,
                                                         const SolveCtl* __restrict__ ctl,
                                                         const int32_t* __restrict__ agg = nullptr, const S* __restrict__ xc = nullptr, S pdamp = S(0),
                                                         Border<S> B = Border<S>())
{
    if (ctl && ctl->done) return;
    auto xf = [&](int j) { return MODE == 2 ? omega * dinv[j] * b[j] : (MODE == 3 ? x[j] + pdamp * xc[agg[j]] : x[j]); };
    if (B.nw && int(blockIdx.x) >= B.gcells) {          // a well row of the bordered level 0
        const int k = blockIdx.x - B.gcells, i = n + k;
        const double ax = border_well(B, k, xf);
        if (threadIdx.x == 0) {
            const S res = b[i] - S(ax);
            if (MODE == 0) out[i] = res;
            else if (MODE == 1) out[i] = x[i] + omega * dinv[i] * res;
            else if (MODE == 3) out[i] = (x[i] + pdamp * xc[agg[i]]) + omega * dinv[i] * res;
            else { out[i] = res; xout[i] = omega * dinv[i] * b[i]; }
        }
        return;
    }
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (row >= n) return;
    const int base = slice_ptr[row >> 6], width = slice_ptr[(row >> 6) + 1] - base, lane = row & 63;
    double acc = 0.0;
    for (int k = l; k < width; k += 64) {
        const long e = long(base + k) * 64 + lane;
        const int j = col[e];
        acc += double(val[e]) * double(xf(j));
    }
    if (l == 0) acc += double(border_cell(B, row, xf));
    acc = wave_sum(acc);
    if (l == 0) {
        const S res = b[row] - S(acc);
        if (MODE == 0) out[row] = res;
        else if (MODE == 1) out[row] = x[row] + omega * dinv[row] * res;
        else if (MODE == 3) out[row] = (x[row] + pdamp * xc[agg[row]]) + omega * dinv[row] * res;
        else { out[row] = res; xout[row] = omega * dinv[row] * b[row]; }
    }
}
// middle levels (10^4 .. 10^5.5 rows of 15-30 entries): LPR lanes per row.  A thread per row walks its entries one after the other at
// two workgroups per CU (latency bound: 12 us for 113 k rows, 3x the time its bytes take), a wavefront per row leaves most lanes idle.
// Same MODEs as k_amg_row_wave; fixed reduction tree (deterministic); levels without a border only.
template <class S, int MODE, int LPR>
__global__ __launch_bounds__(kBlock) void k_amg_row_sub(int n, const int32_t* __restrict__ slice_ptr, const int32_t* __restrict__ col,
                                                        const S* __restrict__ val, const S* __restrict__ b, const S* __restrict__ x,
                                                        S omega, const S* __restrict__ dinv, S* __restrict__ out, S* __restrict__ xout,
                                                        const SolveCtl* __restrict__ ctl,
                                                        const int32_t* __restrict__ agg = nullptr, const S* __restrict__ xc = nullptr, S pdamp = S(0))
{
    if (ctl && ctl->done) return;
    auto xf = [&](int j) { return MODE == 2 ? omega * dinv[j] * b[j] : (MODE == 3 ? x[j] + pdamp * xc[agg[j]] : x[j]); };
    const int row = blockIdx.x * (kBlock / LPR) + int(threadIdx.x) / LPR, l = int(threadIdx.x) % LPR;
    const bool live = row < n;
    double acc = 0.0;
    if (live) {
        const int base = slice_ptr[row >> 6], width = slice_ptr[(row >> 6) + 1] - base, lane = row & 63;
        // predicated batches of 4 slots per lane: all (column, value) loads of a batch are issued before the first gather -- the plain
        // loop is a chain of dependent round trips (column -> x) per slot, and these levels are bound by exactly that latency
        for (int k0 = l; k0 < width; k0 += 4 * LPR) {
            int jj[4]; S vv[4], xx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u * LPR;
                const bool ok = k < width;
                const long e = long(base + (ok ? k : 0)) * 64 + lane;
                jj[u] = ok ? col[e] : -1;
                vv[u] = ok ? val[e] : S(0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) xx[u] = jj[u] >= 0 ? S(xf(jj[u])) : S(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += double(vv[u]) * double(xx[u]);
        }
    }
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (live && l == 0) {
        const S res = b[row] - S(acc);
        if (MODE == 0) out[row] = res;
        else if (MODE == 1) out[row] = x[row] + omega * dinv[row] * res;
        else if (MODE == 3) out[row] = (x[row] + pdamp * xc[agg[row]]) + omega * dinv[row] * res;
        else { out[row] = res; xout[row] = omega * dinv[row] * b[row]; }
    }
}
template <class S>
__global__ __launch_bounds__(kBlock) void k_amg_restrict(int nc, const int32_t* __restrict__ aptr, const int32_t* __restrict__ arows,
                                                         const S* __restrict__ r, S* __restrict__ bc, S omega, const S* __restrict__ dinv_c,
                                                         S* __restrict__ xc, const SolveCtl* __restrict__ ctl)
{
    if (ctl && ctl->done) return;
    const int I = blockIdx.x * kBlock + threadIdx.x;
    if (I >= nc) return;
    const int q0 = aptr[I], q1 = aptr[I + 1];
    S s = 0;
    for (int q = q0; q < q1; q += 8) {          // fixed order, batched gathers
        int rr[8]; S vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) rr[u] = q + u < q1 ? arows[q + u] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) vv[u] = rr[u] >= 0 ? r[rr[u]] : S(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += vv[u];
    }
    bc[I] = s;
    if (xc) xc[I] = omega * dinv_c[I] * s;      // the coarse level's first pre-smoothing sweep from a zero guess, fused
}
template <class S>
__global__ __launch_bounds__(kBlock) void k_amg_prolong(int n, const int32_t* __restrict__ agg, const S* __restrict__ xc, S* __restrict__ x, S pdamp,
                                                        const SolveCtl* __restrict__ ctl)
{
    if (ctl && ctl->done) return;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    x[i] += pdamp * xc[agg[i]];
}
// coarsest level: the sparse operator is scattered into LDS and inverted by Gauss-Jordan between two LDS copies (ping-pong: every
// element of step p is a function of the previous copy only, so ONE barrier per pivot instead of three; n <= kDenseMax = 96:
// 2 x 72 KiB of the 160 KiB per CU); one workgroup of 1024 threads, no pivoting (the pressure operators are diagonally
// dominant M-matrix-like).  Thread (i0, j) owns column j of rows i0, i0 + rows_per_pass, ...: no integer division in the loop.
constexpr int kDenseMax = 96;
template <class S, int kRows>
__global__ __launch_bounds__(1024) void k_dense_invert(int n, int log2_np, const int32_t* __restrict__ slice_ptr, const int32_t* __restrict__ col,
                                                       const S* __restrict__ val, double* __restrict__ inv)
{
    extern __shared__ double a[];    // 2 x [n][n]
    for (int t = threadIdx.x; t < n * n; t += blockDim.x) a[t] = 0.0;
    __syncthreads();
    for (int row = threadIdx.x; row < n; row += blockDim.x) {
        const int base = slice_ptr[row >> 6], width = slice_ptr[(row >> 6) + 1] - base, lane = row & 63;
        for (int k = 0; k < width; ++k) { const long e = long(base + k) * 64 + lane; a[row * n + col[e]] += double(val[e]); }   // padding entries carry 0
    }
    __syncthreads();
    const int np = 1 << log2_np, j = threadIdx.x & (np - 1), i0 = threadIdx.x >> log2_np, istep = blockDim.x >> log2_np;
    double* cur = a;
    double* nxt = a + n * n;
    for (int p = 0; p < n; ++p) {
        if (j < n) {
            // all LDS reads of the step first, then the arithmetic and the stores (the compiler cannot prove cur and nxt disjoint).
            // ~0.85 us per pivot at n = 59 either way: the step is the f64 reciprocal + two LDS round trips + the barrier of 16 waves;
            // fewer threads are slower (512: +14 us, 256: +53 us per inversion)
            // kRows >= ceil(n / istep): 4 for n <= 64 (istep 16), 12 up to kDenseMax (istep 8), 1024 threads
            double aip[kRows], aij[kRows];
            const double piv = cur[p * n + p], prow = cur[p * n + j];
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                const int i = i0 + u * istep;
                aip[u] = i < n ? cur[i * n + p] : 0.0;
                aij[u] = i < n ? cur[i * n + j] : 0.0;
            }
            const double d = 1.0 / piv;
            const double apj = (j == p) ? d : prow * d;          // row p of the next copy
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                const int i = i0 + u * istep;
                if (i < n) nxt[i * n + j] = (i == p) ? apj : ((j == p) ? 0.0 : aij[u]) - aip[u] * apj;
            }
        }
        __syncthreads();
        double* t = cur; cur = nxt; nxt = t;
    }
    // stored TRANSPOSED (inv[j*n + i] = Ainv(i,j)) so that k_dense_apply's loads are contiguous across lanes
    if (j < n) for (int i = i0; i < n; i += istep) inv[j * n + i] = cur[i * n + j];
}
// x = Ainv b on the coarsest level: one wavefront per row (inv is stored transposed, so row i is read with stride n by
// its wave -- n <= 256 keeps that inside a few cache lines per step; what matters is 64-way parallelism per row)
template <class S>
__global__ __launch_bounds__(kBlock) void k_dense_apply(int n, const double* __restrict__ inv, const S* __restrict__ b, S* __restrict__ x,
                                                        const SolveCtl* __restrict__ ctl)
{
    if (ctl && ctl->done) return;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    double s = 0.0;
    for (int j = lane; j < n; j += 64) s += inv[long(j) * n + i] * double(b[j]);
    s = wave_sum(s);
    if (lane == 0) x[i] = S(s);
}

// ---------------------------------------------------------------- host driver
// lanes per coarse entry of the Galerkin kernel: one thread per entry on the big levels (millions of entries with ~5 contributions), 8
// on the small ones (10^5 entries with tens to hundreds: too few threads otherwise).  A/B: OPMGPU_AMG_GALERKIN_LPE=big,small[,threshold]
static int galerkin_lanes(int nce)
{
    static int big = 1, small = 8, below = 400000, parsed = 0;
    if (!parsed) { parsed = 1; if (const char* e = std::getenv("OPMGPU_AMG_GALERKIN_LPE")) std::sscanf(e, "%d,%d,%d", &big, &small, &below); }
    const int v = nce > below ? big : small;
    return (v == 1 || v == 8 || v == 16) ? v : 64;
}
template <class S>
void AmgHierarchy<S>::setup(const Plan& P, const int32_t* d_slice_ptr, const int32_t* d_col, const std::vector<double>& ap_host, const AmgBorderSpec* border)
{
    if (inv_stream) { OPMGPU_HIP(hipStreamSynchronize(inv_stream)); inv_pending = false; }       // dense_inv is re-allocated below
    levels.clear(); level_sizes.clear();
    if (const char* e = std::getenv("OPMGPU_AMG_INV_OVERLAP")) inv_overlap = std::atoi(e) != 0;
    if (const char* e = std::getenv("OPMGPU_AMG_OMEGA")) omega = std::atof(e);
    if (const char* e = std::getenv("OPMGPU_AMG_PDAMP")) { pdamp = std::atof(e); pdamp0 = pdamp; pdamp_user = true; }
    if (const char* e = std::getenv("OPMGPU_AMG_PDAMP0")) { pdamp0 = std::atof(e); pdamp_user = true; }
    if (const char* e = std::getenv("OPMGPU_AMG_NPRE")) npre = std::atoi(e);
    if (const char* e = std::getenv("OPMGPU_AMG_NPOST")) npost = std::atoi(e);
    if (const char* e = std::getenv("OPMGPU_AMG_FUSE")) fuse = std::atoi(e) != 0;
    if (const char* e = std::getenv("OPMGPU_AMG_GRAPH")) use_graph = std::atoi(e) != 0;
    if (const char* e = std::getenv("OPMGPU_AMG_GS")) use_gs = std::atoi(e) != 0;
    gs_n0 = (P.nlevels == 2) ? P.level_ptr[1] : 0;        // two colours: rows [0, n0) and [n0, nb)
    if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
    npost0 = npost;
    if (const char* e = std::getenv("OPMGPU_AMG_NPOST0")) { npost0 = std::atoi(e); npost0_user = true; }
    // bordered level 0 (amg.hpp): one extra unknown per well; too many wells for the dense coarsest level -> no border
    const int nwb = (border && border->nw > 0 && border->nw <= kDenseMax / 2) ? border->nw : 0;
    const int npb = nwb ? border->nperf : 0;
    std::vector<int32_t> perf_of(P.nb, -1);
    for (int j = 0; j < npb; ++j) perf_of[border->perf_row[j]] = j;
    HostCsr A;
    A.n = P.nb + nwb; A.rowptr.assign(A.n + 1, 0);
    for (int r = 0; r < P.nb; ++r) A.rowptr[r + 1] = A.rowptr[r] + P.rowlen[r] + (perf_of[r] >= 0 ? 1 : 0);
    for (int k = 0; k < nwb; ++k) A.rowptr[P.nb + k + 1] = A.rowptr[P.nb + k] + (border->connpos[k + 1] - border->connpos[k]) + 1;
    A.col.resize(A.rowptr[A.n]); A.val.resize(A.col.size()); A.dev.resize(A.col.size());
    std::vector<int32_t> diag0(A.n, 0), well_of(npb, 0);
    for (int k = 0; k < nwb; ++k) for (int j = border->connpos[k]; j < border->connpos[k + 1]; ++j) well_of[j] = k;
    for (int r = 0; r < P.nb; ++r) {
        for (int k = 0; k < P.rowlen[r]; ++k) {
            const int e = P.entry(r, k), s = A.rowptr[r] + k;
            A.col[s] = P.sell_col[e]; A.val[s] = ap_host[e]; A.dev[s] = e;
            if (A.col[s] == r) diag0[r] = e;
        }
        if (perf_of[r] >= 0) {          // column entry towards the well's unknown: device id behind the SELL values
            const int j = perf_of[r], s = A.rowptr[r] + P.rowlen[r];
            A.col[s] = P.nb + well_of[j]; A.val[s] = border->bcol[j]; A.dev[s] = P.nentries + j;
        }
    }
    for (int k = 0; k < nwb; ++k) {
        int s = A.rowptr[P.nb + k];
        for (int j = border->connpos[k]; j < border->connpos[k + 1]; ++j, ++s) { A.col[s] = border->perf_row[j]; A.val[s] = border->crow[j]; A.dev[s] = P.nentries + npb + j; }
        A.col[s] = P.nb + k; A.val[s] = border->dw[k]; A.dev[s] = P.nentries + 2 * npb + k;
        diag0[P.nb + k] = A.dev[s];
    }
    // level 0 borrows the block plan's SELL structure
    std::unique_ptr<AmgLevel<S>> L(new AmgLevel<S>());
    L->n = P.nb; L->nslices = P.nslices; L->nentries = P.nentries; L->slice_ptr = d_slice_ptr; L->col = d_col;
    L->nw = nwb; L->nperf = npb;
    if (nwb) { L->b_connpos = border->d_connpos; L->b_perf_row = border->d_perf_row; L->b_perf_of_row = border->d_perf_of_row; L->b_perf_well = border->d_perf_well; }
    L->diag_entry.upload(diag0, stream);
    int npin = nwb;
    const int kMaxDense = kDenseMax;
    int kMaxLevels = 12;
    if (const char* e = std::getenv("OPMGPU_AMG_MAXLEVELS")) kMaxLevels = std::max(2, std::atoi(e));
    if (const char* e = std::getenv("OPMGPU_AMG_COARSE_SWEEPS")) coarse_sweeps = std::max(0, std::atoi(e));
    while (true) {
        const int n = A.n;              // unknowns of this level (a bordered level 0: cells + wells)
        L->val.alloc(L->nentries + 2 * size_t(L->nperf) + L->nw); L->dinv.alloc(n); L->x.alloc(n); L->b.alloc(n); L->r.alloc(n); L->x2.alloc(n);
        L->x.zero(stream); L->b.zero(stream); L->r.zero(stream); L->x2.zero(stream);
        if (!levels.empty()) L->val.zero(stream);
        level_sizes.push_back(n);
        std::vector<int32_t> agg;
        int na = 0;
        const bool stop = n <= kMaxDense || int(levels.size()) + 1 >= kMaxLevels;
        if (!stop) na = aggregate(A, 0.25, agg, npin);
        // aggressive coarsening (OPMGPU_AMG_AGGR=l: from level l on, aggregate twice and compose): the levels below ~100 k rows are
        // launch-latency bound (~5 us per dependent kernel), so fewer of them shortens the cycle; costs convergence per cycle
        static const int aggr_from = std::getenv("OPMGPU_AMG_AGGR") ? std::atoi(std::getenv("OPMGPU_AMG_AGGR")) : -1;
        if (!stop && aggr_from >= 0 && int(levels.size()) >= aggr_from && na > kMaxDense && na * 10 <= n * 8) {
            HostCsr C1; std::vector<int32_t> cof1, aptr1, arows1, agg2;
            coarsen(A, agg, na, C1, cof1, aptr1, arows1);
            const int na2 = aggregate(C1, 0.25, agg2, npin);
            if (na2 >= 1 && na2 * 10 <= na * 8) { for (int i = 0; i < n; ++i) agg[i] = agg2[agg[i]]; na = na2; }
        }
        if (stop || na * 10 > n * 8 || na < 1) {           // coarsest level (or coarsening stalled)
            levels.push_back(std::move(L));
            break;
        }
        HostCsr C; std::vector<int32_t> cof, aptr, arows;
        coarsen(A, agg, na, C, cof, aptr, arows);
        std::vector<int32_t> sp, scol, dent; int nent = 0;
        to_sell(C, sp, scol, dent, nent);
        // contribution lists per coarse csr entry, in fine DEVICE entry ids
        const int nce = int(C.col.size());
        std::vector<int32_t> cptr(nce + 1, 0), cidx(A.col.size());
        for (size_t s = 0; s < A.col.size(); ++s) cptr[cof[s] + 1]++;
        for (int e = 0; e < nce; ++e) cptr[e + 1] += cptr[e];
        { std::vector<int32_t> fill(cptr.begin(), cptr.end() - 1); for (size_t s = 0; s < A.col.size(); ++s) cidx[fill[cof[s]]++] = A.dev[s]; }
        L->agg.upload(agg, stream); L->agg_ptr.upload(aptr, stream); L->agg_rows.upload(arows, stream);
        // work list of k_amg_galerkin, indexed by the coarse level's DEVICE slot (SELL-64: slice, position in the row, lane): a wavefront
        // covers 64 neighbouring coarse rows at the same position, and coarsen() puts the diagonal first, so its lanes loop equally often
        // and its stores are one contiguous line; padding slots have empty lists (they get 0).  The order of the sum inside an entry is
        // the csr order of the fine entries, as before.
        const int lpe = galerkin_lanes(nce);
        std::vector<int32_t> cptr2(nent + 1, 0), cidx2(cidx.size()), cdiag2(nent, -1);
        for (int e = 0; e < nce; ++e) cptr2[C.dev[e] + 1] = cptr[e + 1] - cptr[e];
        for (int d = 0; d < nent; ++d) cptr2[d + 1] += cptr2[d];
        for (int r = 0; r < C.n; ++r)
            for (int e = C.rowptr[r]; e < C.rowptr[r + 1]; ++e) {
                std::copy(cidx.begin() + cptr[e], cidx.begin() + cptr[e + 1], cidx2.begin() + cptr2[C.dev[e]]);
                if (C.col[e] == r) cdiag2[C.dev[e]] = r;
            }
        L->contrib_ptr.upload(cptr2, stream); L->contrib_idx.upload(cidx2, stream); L->contrib_diag.upload(cdiag2, stream);
        L->n_coarse = na; L->nentries_coarse = nent; L->galerkin_lpe = lpe;
        std::unique_ptr<AmgLevel<S>> Lc(new AmgLevel<S>());
        Lc->n = na; Lc->nslices = (na + 63) / 64; Lc->nentries = nent;
        Lc->own_slice_ptr.upload(sp, stream); Lc->own_col.upload(scol, stream);
        Lc->slice_ptr = Lc->own_slice_ptr.p; Lc->col = Lc->own_col.p;
        Lc->diag_entry.upload(dent, stream);
        levels.push_back(std::move(L));
        L = std::move(Lc);
        A = std::move(C);
    }
    n_coarsest = levels.back()->n;
    if (n_coarsest <= kDenseMax)        // > 64 KiB of dynamic LDS must be requested explicitly
        OPMGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense_invert<S, 12>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kDenseMax * kDenseMax * int(sizeof(double))));
    if (std::getenv("OPMGPU_VERBOSE")) { std::fprintf(stderr, "[opmgpu] AMG levels:"); for (int n : level_sizes) std::fprintf(stderr, " %d", n); std::fprintf(stderr, "\n"); }
    if (n_coarsest <= kDenseMax) {
        dense_inv.alloc(size_t(n_coarsest) * n_coarsest);
    }
    OPMGPU_HIP(hipStreamSynchronize(stream));
}

template <class S>
void AmgHierarchy<S>::galerkin(bool coarse_levels, const std::function<void()>& after_level0)
{
    // the inverse diagonal of level 0 here; those of the coarse levels are written by the Galerkin kernel that produces their operator
    AmgLevel<S>& F0 = *levels[0];
    hipLaunchKernelGGL((k_amg_dinv<S>), dim3(grid_for(F0.ntot())), dim3(kBlock), 0, stream, F0.ntot(), F0.diag_entry.p, F0.val.p, F0.dinv.p);
    if (!coarse_levels) { if (after_level0) after_level0(); return; }           // level 0 follows the matrix, the coarse operators lag
    if (levels.size() < 2 && after_level0) after_level0();
    for (size_t l = 0; l + 1 < levels.size(); ++l) {
        AmgLevel<S>& F = *levels[l]; AmgLevel<S>& C = *levels[l + 1];
        const int nce = F.nentries_coarse;
#define OPMGPU_GALERKIN(LPE) hipLaunchKernelGGL((k_amg_galerkin<S, LPE>), dim3(grid8_for(long(nce) * LPE)), dim3(kBlock), 0, stream, xcd_mode(), nce, \
                               F.contrib_ptr.p, F.contrib_idx.p, F.contrib_diag.p, F.val.p, C.val.p, C.dinv.p)
        switch (F.galerkin_lpe) {
            case 1: OPMGPU_GALERKIN(1); break;
            case 8: OPMGPU_GALERKIN(8); break;
            case 16: OPMGPU_GALERKIN(16); break;
            default: OPMGPU_GALERKIN(64); break;
        }
#undef OPMGPU_GALERKIN
        if (l == 0 && after_level0) after_level0();          // the bandwidth-heavy part of the chain is enqueued: see LinSolver::cpr_prepare
    }
    AmgLevel<S>& B = *levels.back();
    if (n_coarsest <= kDenseMax) {
        // the explicit inverse of the coarsest operator (one workgroup, ~50 us) on a side stream: the first cycle needs it only at the
        // bottom of its down leg, so the inversion runs behind the right-hand side set-up and the down leg (join_inverse())
        if (!inv_stream) {
            OPMGPU_HIP(hipStreamCreateWithFlags(&inv_stream, hipStreamNonBlocking));
            for (auto& e : ev_inv) OPMGPU_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        hipStream_t q = stream;
        if (inv_overlap) {
            OPMGPU_HIP(hipEventRecord(ev_inv[0], stream));
            OPMGPU_HIP(hipStreamWaitEvent(inv_stream, ev_inv[0], 0));
            q = inv_stream;
        }
        int lg = 0; while ((1 << lg) < B.n) ++lg;
        if (B.n <= 64) hipLaunchKernelGGL((k_dense_invert<S, 4>), dim3(1), dim3(1024), size_t(2) * B.n * B.n * sizeof(double), q, B.n, lg, B.slice_ptr, B.col, B.val.p, dense_inv.p);
        else hipLaunchKernelGGL((k_dense_invert<S, 12>), dim3(1), dim3(1024), size_t(2) * B.n * B.n * sizeof(double), q, B.n, lg, B.slice_ptr, B.col, B.val.p, dense_inv.p);
        OPMGPU_HIP(hipGetLastError());
        if (inv_overlap) { OPMGPU_HIP(hipEventRecord(ev_inv[1], inv_stream)); inv_pending = true; }
    }
}
template <class S>
void AmgHierarchy<S>::join_inverse()
{
    if (!inv_pending) return;
    OPMGPU_HIP(hipStreamWaitEvent(stream, ev_inv[1], 0));
    inv_pending = false;
}

constexpr int kSubLanes = 8;
static inline int sub_grid(int n) { return (n + kBlock / kSubLanes - 1) / (kBlock / kSubLanes); }
// levels whose row kernels run kSubLanes lanes per row (k_amg_row_sub); A/B: OPMGPU_AMG_SUB=lo,hi (0,0 = off)
template <class S> static bool sub_rows(const AmgLevel<S>& F)
{
    static int lo = -1, hi = -1;
    if (lo < 0) { lo = 2000; hi = 400000; if (const char* e = std::getenv("OPMGPU_AMG_SUB")) std::sscanf(e, "%d,%d", &lo, &hi); }
    return F.nw == 0 && F.n > lo && F.n <= hi;
}

template <class S>
void AmgHierarchy<S>::vcycle(const SolveCtl* ctl, bool level0_presmoothed)
{
    const S om = S(omega);
    const int nl = int(levels.size());
    // bordered level 0: `gcells` workgroups for the cell rows + one per well (see Border)
    auto bord = [&](const AmgLevel<S>& F, int gcells) {
        Border<S> B;
        if (F.nw == 0) return B;
        B.nw = F.nw; B.n = F.n; B.gcells = gcells; B.connpos = F.b_connpos; B.perf_row = F.b_perf_row; B.perf_of_row = F.b_perf_of_row; B.perf_well = F.b_perf_well;
        B.bcol = F.val.p + F.nentries; B.crow = B.bcol + F.nperf; B.dw = B.crow + F.nperf;
        return B;
    };
    // OPMGPU_AMG_TIME=1: HIP events between the launches of ONE cycle, printed to stderr (diagnostic; no profiler distortion)
    static const bool timing = std::getenv("OPMGPU_AMG_TIME") != nullptr;
    std::vector<std::pair<std::string, hipEvent_t>> marks;
    auto mark = [&](const std::string& name) {
        if (!timing) return;
        hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, stream); marks.emplace_back(name, e);
    };
    mark("start");
    bool presmoothed = level0_presmoothed && fuse;  // F.x already holds omega D^-1 b (fused into the kernel that produced b)
    for (int l = 0; l < nl - 1; ++l) {
        AmgLevel<S>& F = *levels[l]; AmgLevel<S>& C = *levels[l + 1];
        const int g = grid_for(F.n);
        if (l == 0 && level0_halo) {
            // decomposed run: x0 = omega D^-1 b, ghost entries from their owners, then the residual on the global level-0 matrix
            if (!presmoothed) hipLaunchKernelGGL((k_amg_smooth0<S>), dim3(grid_for(F.ntot())), dim3(kBlock), 0, stream, F.ntot(), om, F.dinv.p, F.b.p, F.x.p, ctl);
            if (level0_halo_down) level0_halo(F.x.p, F.b.p);
            hipLaunchKernelGGL((k_amg_residual<S, 0>), dim3(g + F.nw), dim3(kBlock), 0, stream, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, F.x.p, om, F.dinv.p, F.r.p, ctl,
                               (const int32_t*)nullptr, (const S*)nullptr, S(0), bord(F, g));
        } else if (l == 0 && gs_level0() && F.nw == 0) {
            // x = D^-1 b (the first colour's sweep from zero; the caller's fused kernel did it with omega0() = 1), second colour in place,
            // then the residual: zero on the rows just solved, b - A x on the first colour
            const int n0 = gs_n0;
            if (!presmoothed) hipLaunchKernelGGL((k_amg_smooth0<S>), dim3(g), dim3(kBlock), 0, stream, F.n, S(1), F.dinv.p, F.b.p, F.x.p, ctl);
            hipLaunchKernelGGL((k_amg_gs<S, 1>), dim3(grid_for(F.n - n0)), dim3(kBlock), 0, stream, n0, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, F.dinv.p, F.x.p, (S*)nullptr, ctl);
            hipLaunchKernelGGL((k_amg_gs<S, 0>), dim3(grid_for(n0)), dim3(kBlock), 0, stream, 0, n0, F.slice_ptr, F.col, F.val.p, F.b.p, F.dinv.p, F.x.p, F.r.p, ctl);
            OPMGPU_HIP(hipMemsetAsync(F.r.p + n0, 0, size_t(F.n - n0) * sizeof(S), stream));
        } else if (sub_rows(F)) {
            // x = omega D^-1 b came with the restriction (presmoothed) or is formed on the fly (MODE 2)
            if (presmoothed) hipLaunchKernelGGL((k_amg_row_sub<S, 0, kSubLanes>), dim3(sub_grid(F.n)), dim3(kBlock), 0, stream, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, (const S*)F.x.p, om, F.dinv.p, F.r.p, (S*)nullptr, ctl);
            else hipLaunchKernelGGL((k_amg_row_sub<S, 2, kSubLanes>), dim3(sub_grid(F.n)), dim3(kBlock), 0, stream, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, (const S*)F.x.p, om, F.dinv.p, F.r.p, F.x.p, ctl);
        } else if (F.n > 50000) {
            if (!presmoothed) hipLaunchKernelGGL((k_amg_smooth0<S>), dim3(grid_for(F.ntot())), dim3(kBlock), 0, stream, F.ntot(), om, F.dinv.p, F.b.p, F.x.p, ctl);
            hipLaunchKernelGGL((k_amg_residual<S, 0>), dim3(g + F.nw), dim3(kBlock), 0, stream, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, F.x.p, om, F.dinv.p, F.r.p, ctl,
                               (const int32_t*)nullptr, (const S*)nullptr, S(0), bord(F, g));
        } else if (F.n > 20000) {
            hipLaunchKernelGGL((k_amg_smooth0_residual<S>), dim3(g + F.nw), dim3(kBlock), 0, stream, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, om, F.dinv.p, F.x.p, F.r.p, ctl, bord(F, g));
        } else {
            hipLaunchKernelGGL((k_amg_row_wave<S, 2>), dim3((F.n + 3) / 4 + F.nw), dim3(kBlock), 0, stream, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, (const S*)F.x.p, om, F.dinv.p, F.r.p, F.x.p, ctl,
                               (const int32_t*)nullptr, (const S*)nullptr, S(0), bord(F, (F.n + 3) / 4));
        }
        for (int sw = 1; sw < npre; ++sw) {      // further pre-smoothing sweeps, then the residual again
            sweep(F, ctl);
            if (sub_rows(F))
                hipLaunchKernelGGL((k_amg_row_sub<S, 0, kSubLanes>), dim3(sub_grid(F.n)), dim3(kBlock), 0, stream, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, (const S*)F.x.p, om, F.dinv.p, F.r.p, (S*)nullptr, ctl);
            else if (F.n > 20000)
                hipLaunchKernelGGL((k_amg_residual<S, 0>), dim3(g + F.nw), dim3(kBlock), 0, stream, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, F.x.p, om, F.dinv.p, F.r.p, ctl,
                                   (const int32_t*)nullptr, (const S*)nullptr, S(0), bord(F, g));
            else
                hipLaunchKernelGGL((k_amg_row_wave<S, 0>), dim3((F.n + 3) / 4 + F.nw), dim3(kBlock), 0, stream, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, (const S*)F.x.p, om, F.dinv.p, F.r.p, (S*)nullptr, ctl,
                                   (const int32_t*)nullptr, (const S*)nullptr, S(0), bord(F, (F.n + 3) / 4));
        }
        // the restriction also performs the coarse level's first sweep when that level would launch a separate kernel for it
        presmoothed = fuse && (l + 1 < nl - 1) && C.n > 50000;
        hipLaunchKernelGGL((k_amg_restrict<S>), dim3(grid_for(C.n)), dim3(kBlock), 0, stream, C.n, F.agg_ptr.p, F.agg_rows.p, F.r.p, C.b.p, om,
                           presmoothed ? (const S*)C.dinv.p : (const S*)nullptr, presmoothed ? C.x.p : (S*)nullptr, ctl);
        mark("down L" + std::to_string(l));
    }
    AmgLevel<S>& B = *levels.back();
    if (n_coarsest <= kDenseMax) {
        join_inverse();
        hipLaunchKernelGGL((k_dense_apply<S>), dim3((B.n + 3) / 4), dim3(kBlock), 0, stream, B.n, dense_inv.p, B.b.p, B.x.p, ctl);
    } else {        // coarsening stalled above the dense limit: a few Jacobi sweeps stand in for the coarse solve
        hipLaunchKernelGGL((k_amg_smooth0<S>), dim3(grid_for(B.n)), dim3(kBlock), 0, stream, B.n, om, B.dinv.p, B.b.p, B.x.p, ctl);
        for (int s = 0; s < coarse_sweeps; ++s) {
            hipLaunchKernelGGL((k_amg_residual<S, 1>), dim3(grid_for(B.n)), dim3(kBlock), 0, stream, B.n, B.slice_ptr, B.col, B.val.p, B.b.p, B.x.p, om, B.dinv.p, B.x2.p, ctl);
            hipLaunchKernelGGL((k_amg_residual<S, 1>), dim3(grid_for(B.n)), dim3(kBlock), 0, stream, B.n, B.slice_ptr, B.col, B.val.p, B.b.p, B.x2.p, om, B.dinv.p, B.x.p, ctl);
        }
    }
    mark("coarsest");
    for (int l = nl - 2; l >= 0; --l) {
        AmgLevel<S>& F = *levels[l]; AmgLevel<S>& C = *levels[l + 1];
        const int g = grid_for(F.n);
        int done_sweeps = 0;
        const int npost = l == 0 ? this->npost0 : this->npost;
        const double pdamp = l == 0 ? this->pdamp0 : this->pdamp;          // (shadows the member: the correction INTO level l)
        if (l == 0 && level0_halo) {
            // decomposed run: prolongation, then every post-smoothing sweep on an iterate whose ghost entries are the owners'
            hipLaunchKernelGGL((k_amg_prolong<S>), dim3(grid_for(F.ntot())), dim3(kBlock), 0, stream, F.ntot(), F.agg.p, C.x.p, F.x.p, S(pdamp), ctl);
            for (int sw = 0; sw < npost; ++sw) { level0_halo(F.x.p, F.b.p); sweep(F, ctl); }
            mark("up L0");
            continue;
        }
        if (l == 0 && gs_level0() && F.nw == 0) {
            const int n0 = gs_n0;
            hipLaunchKernelGGL((k_amg_prolong<S>), dim3(g), dim3(kBlock), 0, stream, F.n, F.agg.p, C.x.p, F.x.p, S(pdamp), ctl);
            for (int sw = 0; sw < npost; ++sw) {          // colours in reverse order
                hipLaunchKernelGGL((k_amg_gs<S, 1>), dim3(grid_for(F.n - n0)), dim3(kBlock), 0, stream, n0, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, F.dinv.p, F.x.p, (S*)nullptr, ctl);
                hipLaunchKernelGGL((k_amg_gs<S, 1>), dim3(grid_for(n0)), dim3(kBlock), 0, stream, 0, n0, F.slice_ptr, F.col, F.val.p, F.b.p, F.dinv.p, F.x.p, (S*)nullptr, ctl);
            }
            mark("up L0");
            continue;
        }
        if (fuse && npost >= 1 && F.n <= 200000) {
            // small and medium levels: the prolongation is gathered inside the first post-smoothing sweep (one launch less)
            if (sub_rows(F))
                hipLaunchKernelGGL((k_amg_row_sub<S, 3, kSubLanes>), dim3(sub_grid(F.n)), dim3(kBlock), 0, stream, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, (const S*)F.x.p, om, F.dinv.p, F.x2.p,
                                   (S*)nullptr, ctl, (const int32_t*)F.agg.p, (const S*)C.x.p, S(pdamp));
            else if (F.n > 20000)
                hipLaunchKernelGGL((k_amg_residual<S, 3>), dim3(g + F.nw), dim3(kBlock), 0, stream, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, (const S*)F.x.p, om, F.dinv.p, F.x2.p, ctl,
                                   (const int32_t*)F.agg.p, (const S*)C.x.p, S(pdamp), bord(F, g));
            else
                hipLaunchKernelGGL((k_amg_row_wave<S, 3>), dim3((F.n + 3) / 4 + F.nw), dim3(kBlock), 0, stream, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, (const S*)F.x.p, om, F.dinv.p, F.x2.p,
                                   (S*)nullptr, ctl, (const int32_t*)F.agg.p, (const S*)C.x.p, S(pdamp), bord(F, (F.n + 3) / 4));
            std::swap(F.x.p, F.x2.p);
            done_sweeps = 1;
        } else {
            hipLaunchKernelGGL((k_amg_prolong<S>), dim3(grid_for(F.ntot())), dim3(kBlock), 0, stream, F.ntot(), F.agg.p, C.x.p, F.x.p, S(pdamp), ctl);
        }
        for (int sw = done_sweeps; sw < npost; ++sw) sweep(F, ctl);
        mark("up L" + std::to_string(l));
    }
    if (timing && !marks.empty()) {
        (void)hipStreamSynchronize(stream);
        static int printed = 0;
        if (printed++ == 40) {           // one cycle deep inside a solve
            for (size_t i = 1; i < marks.size(); ++i) { float ms = 0; (void)hipEventElapsedTime(&ms, marks[i - 1].second, marks[i].second); std::fprintf(stderr, "[amg] %-10s %7.1f us\n", marks[i].first.c_str(), 1e3 * ms); }
        }
        for (auto& m : marks) (void)hipEventDestroy(m.second);
    }
}

// V-cycle through a captured graph: the launches of one cycle are recorded once per (ctl, presmoothed) and replayed.  Valid
// because a cycle performs an even number of x/x2 swaps per level (1 + 2 sweeps), so the buffer roles are the same at every entry.
template <class S>
void AmgHierarchy<S>::vcycle_graph(const SolveCtl* ctl, bool level0_presmoothed)
{
    const bool even = (npost % 2 == 0) && (npost0 % 2 == 0) && npre == 1;
    if (!use_graph || !even || level0_halo) { vcycle(ctl, level0_presmoothed); return; }          // (the exchange is not capturable)
    join_inverse();          // an event of another stream cannot be waited for inside a capture
    // the captured launches carry the smoother / correction constants as baked-in kernel arguments: they are part of the cache key
    const double key[6] = { pdamp0, pdamp, omega0(), double(npost), double(npost0), double(npre) };
    bool same = graph_exec && graph_ctl == ctl && graph_pre == level0_presmoothed;
    for (int k = 0; k < 6 && same; ++k) same = graph_key[k] == key[k];
    if (!same) {
        for (int k = 0; k < 6; ++k) graph_key[k] = key[k];
        if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
        hipGraph_t g = nullptr;
        OPMGPU_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        vcycle(ctl, level0_presmoothed);
        OPMGPU_HIP(hipStreamEndCapture(stream, &g));
        OPMGPU_HIP(hipGraphInstantiate(&graph_exec, g, nullptr, nullptr, 0));
        (void)hipGraphDestroy(g);
        graph_ctl = ctl; graph_pre = level0_presmoothed;
    }
    OPMGPU_HIP(hipGraphLaunch(graph_exec, stream));
}

template <class S>
__global__ __launch_bounds__(kBlock) void k_amg_norm2_parts(int n, const S* __restrict__ r, double* __restrict__ parts)
{
    __shared__ double sm[4];
    double acc[1] = { 0.0 };
    for (long i = blockIdx.x * long(kBlock) + threadIdx.x; i < n; i += long(gridDim.x) * kBlock) acc[0] += double(r[i]) * double(r[i]);
    block_sum<1>(acc, sm);
    if (threadIdx.x == 0) parts[blockIdx.x] = acc[0];
}
__global__ __launch_bounds__(kBlock) void k_amg_sum_parts(int np, const double* __restrict__ parts, double* __restrict__ out)
{
    __shared__ double sm[4];
    double acc[1] = { 0.0 };
    for (int i = threadIdx.x; i < np; i += kBlock) acc[0] += parts[i];
    block_sum<1>(acc, sm);
    if (threadIdx.x == 0) out[0] = acc[0];
}
template <class S>
void AmgHierarchy<S>::residual0(const SolveCtl* ctl)
{
    AmgLevel<S>& F = *levels[0];
    const int g = grid_for(F.n);
    Border<S> B;
    if (F.nw) {
        B.nw = F.nw; B.n = F.n; B.gcells = g; B.connpos = F.b_connpos; B.perf_row = F.b_perf_row; B.perf_of_row = F.b_perf_of_row; B.perf_well = F.b_perf_well;
        B.bcol = F.val.p + F.nentries; B.crow = B.bcol + F.nperf; B.dw = B.crow + F.nperf;
    }
    hipLaunchKernelGGL((k_amg_residual<S, 0>), dim3(g + F.nw), dim3(kBlock), 0, stream, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, F.x.p, S(omega), F.dinv.p, F.r.p, ctl,
                       (const int32_t*)nullptr, (const S*)nullptr, S(0), B);
}
template <class S>
void AmgHierarchy<S>::residual_norm2(double* d_out)
{
    AmgLevel<S>& F = *levels[0];
    const int g = grid_for(F.n);
    Border<S> B;
    if (F.nw) {
        B.nw = F.nw; B.n = F.n; B.gcells = g; B.connpos = F.b_connpos; B.perf_row = F.b_perf_row; B.perf_of_row = F.b_perf_of_row; B.perf_well = F.b_perf_well;
        B.bcol = F.val.p + F.nentries; B.crow = B.bcol + F.nperf; B.dw = B.crow + F.nperf;
    }
    hipLaunchKernelGGL((k_amg_residual<S, 0>), dim3(g + F.nw), dim3(kBlock), 0, stream, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, F.x.p, S(omega), F.dinv.p, F.r.p, (const SolveCtl*)nullptr,
                       (const int32_t*)nullptr, (const S*)nullptr, S(0), B);
    const int np = 512;
    if (tune_parts.n < size_t(np)) tune_parts.alloc(np);
    hipLaunchKernelGGL((k_amg_norm2_parts<S>), dim3(np), dim3(kBlock), 0, stream, F.ntot(), (const S*)F.r.p, tune_parts.p);
    hipLaunchKernelGGL(k_amg_sum_parts, dim3(1), dim3(kBlock), 0, stream, np, (const double*)tune_parts.p, d_out);
}

template <class S>
__global__ __launch_bounds__(kBlock) void k_amg_hash_fill(int n, S* __restrict__ x)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint32_t h = uint32_t(i) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    x[i] = S(double(h) / 2147483648.0 - 1.0);
}
template <class S>
void AmgHierarchy<S>::smooth_test_rhs(int sweeps)
{
    AmgLevel<S>& F = *levels[0];
    const size_t nt = size_t(F.ntot());
    if (tune_b.n < nt) tune_b.alloc(nt);
    OPMGPU_HIP(hipMemcpyAsync(tune_b.p, F.b.p, nt * sizeof(S), hipMemcpyDeviceToDevice, stream));
    OPMGPU_HIP(hipMemsetAsync(F.b.p, 0, nt * sizeof(S), stream));
    hipLaunchKernelGGL((k_amg_hash_fill<S>), dim3(grid_for(F.n)), dim3(kBlock), 0, stream, F.n, F.x.p);
    if (F.nw) OPMGPU_HIP(hipMemsetAsync(F.x.p + F.n, 0, F.nw * sizeof(S), stream));
    for (int k = 0; k < 2 * ((sweeps + 1) / 2); ++k) sweep(F, nullptr);       // an even count: x / x2 keep their roles
    if (tune_parts.n < 512) tune_parts.alloc(512);
    residual_norm2(tune_parts.p);                                             // r = 0 - A s
    OPMGPU_HIP(hipMemcpyAsync(F.b.p, F.r.p, nt * sizeof(S), hipMemcpyDeviceToDevice, stream));
}
template <class S>
void AmgHierarchy<S>::restore_rhs()
{
    AmgLevel<S>& F = *levels[0];
    OPMGPU_HIP(hipMemcpyAsync(F.b.p, tune_b.p, size_t(F.ntot()) * sizeof(S), hipMemcpyDeviceToDevice, stream));
}

// one damped-Jacobi sweep x <- x + omega D^-1 (b - A x) (ping-pong between x and x2)
template <class S>
void AmgHierarchy<S>::sweep(AmgLevel<S>& F, const SolveCtl* ctl)
{
    const S om = S(omega);
    Border<S> B;
    if (F.nw) {
        B.nw = F.nw; B.n = F.n; B.connpos = F.b_connpos; B.perf_row = F.b_perf_row; B.perf_of_row = F.b_perf_of_row; B.perf_well = F.b_perf_well;
        B.bcol = F.val.p + F.nentries; B.crow = B.bcol + F.nperf; B.dw = B.crow + F.nperf;
    }
    if (sub_rows(F)) {
        hipLaunchKernelGGL((k_amg_row_sub<S, 1, kSubLanes>), dim3(sub_grid(F.n)), dim3(kBlock), 0, stream, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, (const S*)F.x.p, om, F.dinv.p, F.x2.p, (S*)nullptr, ctl);
    } else if (F.n > 20000) {
        B.gcells = grid_for(F.n);
        hipLaunchKernelGGL((k_amg_residual<S, 1>), dim3(grid_for(F.n) + F.nw), dim3(kBlock), 0, stream, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, F.x.p, om, F.dinv.p, F.x2.p, ctl,
                           (const int32_t*)nullptr, (const S*)nullptr, S(0), B);
    } else {
        B.gcells = (F.n + 3) / 4;
        hipLaunchKernelGGL((k_amg_row_wave<S, 1>), dim3((F.n + 3) / 4 + F.nw), dim3(kBlock), 0, stream, F.n, F.slice_ptr, F.col, F.val.p, F.b.p, (const S*)F.x.p, om, F.dinv.p, F.x2.p, (S*)nullptr, ctl,
                           (const int32_t*)nullptr, (const S*)nullptr, S(0), B);
    }
    std::swap(F.x.p, F.x2.p);       // the swept iterate becomes x (buffers are the same size)
}

template class AmgHierarchy<float>;
template class AmgHierarchy<double>;

} // namespace opmgpu
