// linsolver.hip -- gfx950 kernels for the block-ILU0 / BiCGStab solve (see linsolver.hpp).
//
// All kernels are HBM-bandwidth bound (<= 0.25 flop/byte): no MFMA.  One thread owns one block row;
// the SELL-64 layout makes every matrix / index load of a wavefront one contiguous segment, the
// vectors are component-major planes so x[col] gathers of neighbouring rows coalesce as well.
// Reductions are two-stage and order-deterministic (per-workgroup partials, single-workgroup
// finalise that also updates the BiCGStab scalars on the device).
#include "linsolver.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace opmgpu {

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ long vidx(int base_slot, int lane) { return long(base_slot) * 576 + lane; }

// y = A x  (+ fused dot products: NDOT==1: <w1,y>; NDOT==2: <y,w1>, <y,y>)
// MatrixAdapter::apply + the scalar products of BiCGSTABSolver::apply.
template <class S, int NDOT>
__global__ __launch_bounds__(kBlock) void k_spmv(int nb, int nbp, const int32_t* __restrict__ slice_ptr,
                                                 const int32_t* __restrict__ col, const S* __restrict__ val,
                                                 const S* __restrict__ x, S* __restrict__ y,
                                                 const S* __restrict__ w1, double* __restrict__ partials)
{
    __shared__ double sm[8];
    const int row = blockIdx.x * kBlock + threadIdx.x;
    double acc[2] = { 0.0, 0.0 };
    if (row < nb) {
        const int sl = row >> 6, lane = row & 63;
        const int base = slice_ptr[sl], width = slice_ptr[sl + 1] - base;
        const S* __restrict__ v = val + vidx(base, lane);
        const int32_t* __restrict__ c = col + long(base) * 64 + lane;
        S y0 = 0, y1 = 0, y2 = 0;
        for (int k = 0; k < width; ++k) {
            const int cc = c[k * 64];
            const S x0 = x[cc], x1 = x[nbp + cc], x2 = x[2 * nbp + cc];
            const S* __restrict__ b = v + k * 576;
            y0 += b[0] * x0 + b[64] * x1 + b[128] * x2;
            y1 += b[192] * x0 + b[256] * x1 + b[320] * x2;
            y2 += b[384] * x0 + b[448] * x1 + b[512] * x2;
        }
        y[row] = y0; y[nbp + row] = y1; y[2 * nbp + row] = y2;
        if (NDOT >= 1) acc[0] = double(w1[row]) * double(y0) + double(w1[nbp + row]) * double(y1) + double(w1[2 * nbp + row]) * double(y2);
        if (NDOT == 2) acc[1] = double(y0) * double(y0) + double(y1) * double(y1) + double(y2) * double(y2);
    }
    if (NDOT >= 1) {
        block_sum<2>(acc, sm);
        if (threadIdx.x == 0) {
            partials[blockIdx.x] = acc[0];
            if (NDOT == 2) partials[gridDim.x + blockIdx.x] = acc[1];
        }
    }
}

// forward sweep of one level: v_i = w d_i - sum_{j lower} L_ij v_j ; for the top level the pivot
// inverse is applied at once (no upper entries there).  ParallelOverlappingILU0::apply, lower part;
// the relaxation factor is folded in here (the sweeps are linear in d).
template <class S>
__global__ __launch_bounds__(kBlock) void k_ilu_lower(int lo, int hi, int nbp, int top, S w, const int32_t* __restrict__ slice_ptr,
                                                      const int32_t* __restrict__ col, const int16_t* __restrict__ nlower,
                                                      const S* __restrict__ lu, const S* __restrict__ d, S* __restrict__ v)
{
    const int row = lo + blockIdx.x * kBlock + threadIdx.x;
    if (row >= hi) return;
    const int base = slice_ptr[row >> 6], lane = row & 63, nl = nlower[row];
    const S* __restrict__ m = lu + vidx(base, lane);
    const int32_t* __restrict__ c = col + long(base) * 64 + lane;
    S r0 = w * d[row], r1 = w * d[nbp + row], r2 = w * d[2 * nbp + row];
    for (int k = 0; k < nl; ++k) {
        const int cc = c[k * 64];
        const S x0 = v[cc], x1 = v[nbp + cc], x2 = v[2 * nbp + cc];
        const S* __restrict__ b = m + k * 576;
        r0 -= b[0] * x0 + b[64] * x1 + b[128] * x2;
        r1 -= b[192] * x0 + b[256] * x1 + b[320] * x2;
        r2 -= b[384] * x0 + b[448] * x1 + b[512] * x2;
    }
    if (top) {
        const S* __restrict__ b = m + nl * 576;
        const S t0 = b[0] * r0 + b[64] * r1 + b[128] * r2;
        const S t1 = b[192] * r0 + b[256] * r1 + b[320] * r2;
        const S t2 = b[384] * r0 + b[448] * r1 + b[512] * r2;
        r0 = t0; r1 = t1; r2 = t2;
    }
    v[row] = r0; v[nbp + row] = r1; v[2 * nbp + row] = r2;
}

// backward sweep of one level: v_i = Dinv_i (v_i - sum_{j upper} U_ij v_j)
template <class S>
__global__ __launch_bounds__(kBlock) void k_ilu_upper(int lo, int hi, int nbp, const int32_t* __restrict__ slice_ptr,
                                                      const int32_t* __restrict__ col, const int16_t* __restrict__ nlower,
                                                      const int16_t* __restrict__ rowlen, const S* __restrict__ lu, S* __restrict__ v)
{
    const int row = lo + blockIdx.x * kBlock + threadIdx.x;
    if (row >= hi) return;
    const int base = slice_ptr[row >> 6], lane = row & 63, nl = nlower[row], len = rowlen[row];
    const S* __restrict__ m = lu + vidx(base, lane);
    const int32_t* __restrict__ c = col + long(base) * 64 + lane;
    S r0 = v[row], r1 = v[nbp + row], r2 = v[2 * nbp + row];
    for (int k = nl + 1; k < len; ++k) {
        const int cc = c[k * 64];
        const S x0 = v[cc], x1 = v[nbp + cc], x2 = v[2 * nbp + cc];
        const S* __restrict__ b = m + k * 576;
        r0 -= b[0] * x0 + b[64] * x1 + b[128] * x2;
        r1 -= b[192] * x0 + b[256] * x1 + b[320] * x2;
        r2 -= b[384] * x0 + b[448] * x1 + b[512] * x2;
    }
    const S* __restrict__ b = m + nl * 576;
    v[row] = b[0] * r0 + b[64] * r1 + b[128] * r2;
    v[nbp + row] = b[192] * r0 + b[256] * r1 + b[320] * r2;
    v[2 * nbp + row] = b[384] * r0 + b[448] * r1 + b[512] * r2;
}

template <class S> __device__ __forceinline__ void ld9(const S* __restrict__ a, int32_t e, S (&m)[9])
{
    const S* p = a + long(e >> 6) * 576 + (e & 63);
#pragma unroll
    for (int q = 0; q < 9; ++q) m[q] = p[q * 64];
}
template <class S> __device__ __forceinline__ void st9(S* __restrict__ a, int32_t e, const S (&m)[9])
{
    S* p = a + long(e >> 6) * 576 + (e & 63);
#pragma unroll
    for (int q = 0; q < 9; ++q) p[q * 64] = m[q];
}
template <class S> __device__ __forceinline__ void mm9(const S (&a)[9], const S (&b)[9], S (&c)[9])
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

// numeric block-ILU(0) of the rows of one level (IKJ; dune-istl bilu0_decomposition order inside a
// row; pivots inverted explicitly with the cofactor formula like opm-simulators' MatrixBlock).
template <class S>
__global__ __launch_bounds__(kBlock) void k_ilu_factor(int lo, int hi, const int32_t* __restrict__ slice_ptr, const int32_t* __restrict__ col,
                                                       const int16_t* __restrict__ nlower, const int32_t* __restrict__ trip_ptr,
                                                       const int32_t* __restrict__ trip_l, const int32_t* __restrict__ trip_u,
                                                       const int32_t* __restrict__ trip_t, S* __restrict__ lu, int32_t* __restrict__ flags)
{
    const int row = lo + blockIdx.x * kBlock + threadIdx.x;
    if (row >= hi) return;
    const int base = slice_ptr[row >> 6], lane = row & 63, nl = nlower[row];
    int tp = trip_ptr[row];
    const int te = trip_ptr[row + 1];
    for (int k = 0; k < nl; ++k) {
        const int32_t e = (base + k) * 64 + lane;
        const int j = col[e];
        const int32_t ej = (slice_ptr[j >> 6] + nlower[j]) * 64 + (j & 63);
        S a[9], dj[9], L[9];
        ld9(lu, e, a); ld9(lu, ej, dj);
        mm9(a, dj, L);
        st9(lu, e, L);
        while (tp < te && trip_l[tp] == e) {
            S u[9], t[9], bb[9];
            ld9(lu, trip_u[tp], u); ld9(lu, trip_t[tp], t);
            mm9(L, u, bb);
#pragma unroll
            for (int q = 0; q < 9; ++q) t[q] -= bb[q];
            st9(lu, trip_t[tp], t);
            ++tp;
        }
    }
    const int32_t ed = (base + nl) * 64 + lane;
    S m[9], o[9];
    ld9(lu, ed, m);
    const S c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
    const S det = m[0] * c0 + m[1] * c1 + m[2] * c2;
    if (det == S(0) || !(det == det)) { atomicOr(flags, 1); return; }
    const S id = S(1) / det;
    o[0] = c0 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c1 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c2 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    st9(lu, ed, o);
}

// ---- BiCGStab vector updates (grid-stride over the 3*nbp plane elements) ----
// p = r + beta (p - omega v)   (first half-step: p = r)
template <class S>
__global__ __launch_bounds__(kBlock) void k_update_p(long n, int first, const double* __restrict__ sc, const S* __restrict__ r,
                                                     const S* __restrict__ v, S* __restrict__ p)
{
    const S beta = S(sc[SC_BETA]), omega = S(sc[SC_OMEGA]);
    for (long i = blockIdx.x * long(kBlock) + threadIdx.x; i < n; i += long(gridDim.x) * kBlock)
        p[i] = first ? r[i] : (p[i] - omega * v[i]) * beta + r[i];
}
// x += a y ; r -= a q ; partials: <r,r> [, <rt,r>]      (a = alpha or omega, read from the device scalars)
template <class S, int NDOT>
__global__ __launch_bounds__(kBlock) void k_update_xr(long n, int which, const double* __restrict__ sc, const S* __restrict__ y,
                                                      const S* __restrict__ q, const S* __restrict__ rt, S* __restrict__ x,
                                                      S* __restrict__ r, double* __restrict__ partials)
{
    __shared__ double sm[8];
    const S a = S(sc[which]);
    double acc[2] = { 0.0, 0.0 };
    for (long i = blockIdx.x * long(kBlock) + threadIdx.x; i < n; i += long(gridDim.x) * kBlock) {
        x[i] += a * y[i];
        const S rn = r[i] - a * q[i];
        r[i] = rn;
        acc[0] += double(rn) * double(rn);
        if (NDOT == 2) acc[1] += double(rt[i]) * double(rn);
    }
    block_sum<2>(acc, sm);
    if (threadIdx.x == 0) { partials[blockIdx.x] = acc[0]; if (NDOT == 2) partials[gridDim.x + blockIdx.x] = acc[1]; }
}
template <class S>
__global__ __launch_bounds__(kBlock) void k_dot(long n, const S* __restrict__ a, const S* __restrict__ b, double* __restrict__ partials)
{
    __shared__ double sm[8];
    double acc[1] = { 0.0 };
    for (long i = blockIdx.x * long(kBlock) + threadIdx.x; i < n; i += long(gridDim.x) * kBlock) acc[0] += double(a[i]) * double(b[i]);
    block_sum<1>(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}
template <class S>
__global__ __launch_bounds__(kBlock) void k_axpy(long n, S a, const S* __restrict__ x, S* __restrict__ y)
{
    for (long i = blockIdx.x * long(kBlock) + threadIdx.x; i < n; i += long(gridDim.x) * kBlock) y[i] += a * x[i];
}
template <class A, class B>
__global__ __launch_bounds__(kBlock) void k_convert(long n, const A* __restrict__ a, B* __restrict__ b)
{
    for (long i = blockIdx.x * long(kBlock) + threadIdx.x; i < n; i += long(gridDim.x) * kBlock) b[i] = B(a[i]);
}
__global__ __launch_bounds__(kBlock) void k_copy16(long n16, const double2* __restrict__ a, double2* __restrict__ b)
{
    for (long i = blockIdx.x * long(kBlock) + threadIdx.x; i < n16; i += long(gridDim.x) * kBlock) b[i] = a[i];
}

// single-workgroup finalise: fixed-order sum of the per-workgroup partials + scalar recurrences of
// Dune::BiCGSTABSolver::apply.
enum { FIN_INIT = 0, FIN_H, FIN_NORM1, FIN_OMEGA, FIN_NORM2, FIN_PLAIN };
__global__ __launch_bounds__(kBlock) void k_finalize(int op, int nblocks, double eps, const double* __restrict__ partials, double* __restrict__ sc)
{
    __shared__ double sm[8];
    double acc[2] = { 0.0, 0.0 };
    for (int i = threadIdx.x; i < nblocks; i += kBlock) { acc[0] += partials[i]; acc[1] += partials[nblocks + i]; }
    block_sum<2>(acc, sm);
    if (threadIdx.x != 0) return;
    switch (op) {
    case FIN_INIT:      // r = rt = b
        sc[SC_NORM0_2] = acc[0]; sc[SC_NORM2] = acc[0]; sc[SC_RHONEW] = acc[0];
        sc[SC_RHO] = 1.0; sc[SC_ALPHA] = 1.0; sc[SC_OMEGA] = 1.0; sc[SC_BETA] = 0.0; sc[SC_FLAG] = 0.0;
        break;
    case FIN_H:         // h = <rt,v>; alpha = rho_new / h
        sc[SC_H] = acc[0];
        if (fabs(acc[0]) < eps || !(acc[0] == acc[0])) sc[SC_FLAG] = 1.0;
        sc[SC_ALPHA] = sc[SC_RHONEW] / acc[0];
        break;
    case FIN_NORM1:
        sc[SC_NORM2] = acc[0];
        break;
    case FIN_OMEGA:     // omega = <t,r> / <t,t>
        sc[SC_TR] = acc[0]; sc[SC_TT] = acc[1];
        sc[SC_OMEGA] = acc[0] / acc[1];
        break;
    case FIN_NORM2: {   // rho = rho_new; rho_new = <rt,r>; beta for the next half step
        sc[SC_NORM2] = acc[0];
        const double rho = sc[SC_RHONEW], omega = sc[SC_OMEGA];
        sc[SC_RHO] = rho; sc[SC_RHONEW] = acc[1];
        if (fabs(rho) <= eps || fabs(omega) <= eps || !(rho == rho) || !(omega == omega)) sc[SC_FLAG] = 2.0;
        sc[SC_BETA] = (acc[1] / rho) * (sc[SC_ALPHA] / omega);
        break;
    }
    default:
        sc[SC_NORM2] = acc[0];
        break;
    }
}

// ---- layout conversion kernels ----
__global__ __launch_bounds__(kBlock) void k_bsr_to_sell(int nentries, const int32_t* __restrict__ src, const double* __restrict__ bsr, double* __restrict__ sell)
{
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= nentries) return;
    const int s = src[e];
    double* o = sell + long(e >> 6) * 576 + (e & 63);
#pragma unroll
    for (int q = 0; q < 9; ++q) o[q * 64] = s >= 0 ? bsr[long(s) * 9 + q] : 0.0;
}
template <class S>
__global__ __launch_bounds__(kBlock) void k_sell_to_bsr(int nnzb, const int32_t* __restrict__ entry_of_block, const S* __restrict__ sell, double* __restrict__ bsr)
{
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= nnzb) return;
    const int e = entry_of_block[b];
    const S* o = sell + long(e >> 6) * 576 + (e & 63);
#pragma unroll
    for (int q = 0; q < 9; ++q) bsr[long(b) * 9 + q] = double(o[q * 64]);
}
template <class S>
__global__ __launch_bounds__(kBlock) void k_vec_in(int nb, int nbp, int layout, const int32_t* __restrict__ nat, const double* __restrict__ h, S* __restrict__ d)
{
    const int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= nb) return;
    const int c = nat[r];
#pragma unroll
    for (int k = 0; k < 3; ++k) d[long(k) * nbp + r] = S(layout == VEC_BLOCK_INTERLEAVED ? h[3 * long(c) + k] : h[long(k) * nb + c]);
}
template <class S>
__global__ __launch_bounds__(kBlock) void k_vec_out(int nb, int nbp, int layout, const int32_t* __restrict__ nat, const S* __restrict__ d, double* __restrict__ h)
{
    const int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= nb) return;
    const int c = nat[r];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double val = double(d[long(k) * nbp + r]);
        if (layout == VEC_BLOCK_INTERLEAVED) h[3 * long(c) + k] = val; else h[long(k) * nb + c] = val;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
void DevPlan::upload(const Plan& P, hipStream_t s)
{
    nb = P.nb; nbp = P.nbp; nslices = P.nslices; nentries = P.nentries; nlevels = P.nlevels; nnzb = P.nnzb;
    slice_ptr.upload(P.slice_ptr, s); col.upload(P.sell_col, s); src.upload(P.sell_src, s);
    entry_of_block.upload(P.entry_of_block, s); nat.upload(P.nat, s); pos.upload(P.pos, s);
    trip_ptr.upload(P.trip_ptr, s);
    // hipMalloc(0) is avoided: keep at least one element
    std::vector<int32_t> one(1, 0);
    trip_l.upload(P.trip_l.empty() ? one : P.trip_l, s); trip_u.upload(P.trip_u.empty() ? one : P.trip_u, s);
    trip_t.upload(P.trip_t.empty() ? one : P.trip_t, s);
    rowlen.upload(P.rowlen, s); nlower.upload(P.nlower, s);
    level_ptr = P.level_ptr;
    OPMGPU_HIP(hipStreamSynchronize(s));      // the host vectors may go away
}

LinSolver::LinSolver(hipStream_t s) : stream(s)
{
    partials.alloc(2 * kMaxRedBlocks);
    scalars.alloc(SC_COUNT);
    flags.alloc(4);
    partials.zero(stream); scalars.zero(stream); flags.zero(stream);
    OPMGPU_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_scalars), SC_COUNT * sizeof(double)));
    OPMGPU_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_flags), 4 * sizeof(int32_t)));
}
LinSolver::~LinSolver()
{
    if (h_scalars) (void)hipHostFree(h_scalars);
    if (h_flags) (void)hipHostFree(h_flags);
}

template <> SolverWork<double>& LinSolver::work<double>() { return wd; }
template <> SolverWork<float>& LinSolver::work<float>() { return wf; }

int LinSolver::set_pattern(int nb, const int32_t* rowptr, const int32_t* col, int ordering)
{
    if (plan.nb == nb && cur_ordering == ordering && plan.nnzb == rowptr[nb] &&
        std::memcmp(plan.rowptr.data(), rowptr, sizeof(int32_t) * (nb + 1)) == 0 &&
        std::memcmp(plan.col.data(), col, sizeof(int32_t) * plan.nnzb) == 0)
        return OPMGPU_OK;
    Plan P;
    const int st = build_plan(nb, rowptr, col, ordering, P);
    if (st != OPMGPU_OK) return st;
    plan.rowptr.clear();
    plan = std::move(P);
    cur_ordering = ordering;
    dp.upload(plan, stream);
    Ad.alloc(size_t(plan.nentries) * 9);
    Ad.zero(stream);
    partials.alloc(size_t(2) * std::max(grid_for(plan.nb), kMaxRedBlocks));
    partials.zero(stream);
    wd.allocated = false; wf.allocated = false;
    return OPMGPU_OK;
}

template <class S> void LinSolver::ensure_work()
{
    SolverWork<S>& w = work<S>();
    if (w.allocated) return;
    const size_t nv = size_t(3) * plan.nbp, nm = size_t(plan.nentries) * 9;
    if (sizeof(S) == 4) { w.A.alloc(nm); w.A.zero(stream); }
    w.LU.alloc(nm); w.LU.zero(stream);
    DevArray<S>* vs[] = { &w.r, &w.rt, &w.p, &w.v, &w.t, &w.y, &w.x, &w.b };
    for (DevArray<S>* a : vs) { a->alloc(nv); a->zero(stream); }
    w.allocated = true;
}

void LinSolver::load_host_bsr(const double* val9)
{
    stage.ensure(std::max(size_t(plan.nnzb) * 9, size_t(3) * plan.nbp));
    OPMGPU_HIP(hipMemcpyAsync(stage.p, val9, size_t(plan.nnzb) * 9 * sizeof(double), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_bsr_to_sell, dim3(grid_for(plan.nentries)), dim3(kBlock), 0, stream, plan.nentries, dp.src.p, stage.p, Ad.p);
}

template <> void LinSolver::prepare<double>(bool) { ensure_work<double>(); }
template <> void LinSolver::prepare<float>(bool matrix_changed)
{
    ensure_work<float>();
    if (!matrix_changed) return;
    const long n = long(plan.nentries) * 9;
    hipLaunchKernelGGL((k_convert<double, float>), dim3(std::min(grid_for(n), kMaxRedBlocks)), dim3(kBlock), 0, stream, n, Ad.p, wf.A.p);
}
template <> const double* LinSolver::matrix<double>() { return Ad.p; }
template <> const float* LinSolver::matrix<float>() { return wf.A.p; }

template <class S> int LinSolver::factor()
{
    SolverWork<S>& w = work<S>();
    const size_t bytes = size_t(plan.nentries) * 9 * sizeof(S);
    OPMGPU_HIP(hipMemcpyAsync(w.LU.p, matrix<S>(), bytes, hipMemcpyDeviceToDevice, stream));
    flags.zero(stream);
    for (int l = 0; l < plan.nlevels; ++l) {
        const int lo = plan.level_ptr[l], hi = plan.level_ptr[l + 1];
        if (hi == lo) continue;
        hipLaunchKernelGGL((k_ilu_factor<S>), dim3(grid_for(hi - lo)), dim3(kBlock), 0, stream, lo, hi, dp.slice_ptr.p, dp.col.p,
                           dp.nlower.p, dp.trip_ptr.p, dp.trip_l.p, dp.trip_u.p, dp.trip_t.p, w.LU.p, flags.p);
    }
    OPMGPU_HIP(hipMemcpyAsync(h_flags, flags.p, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    OPMGPU_HIP(hipStreamSynchronize(stream));
    return (h_flags[0] & 1) ? OPMGPU_ESINGULAR : OPMGPU_OK;
}

template <class S> void LinSolver::ilu_apply(const S* d, S* v, double relax)
{
    SolverWork<S>& w = work<S>();
    const int L = plan.nlevels;
    for (int l = 0; l < L; ++l) {
        const int lo = plan.level_ptr[l], hi = plan.level_ptr[l + 1];
        if (hi == lo) continue;
        hipLaunchKernelGGL((k_ilu_lower<S>), dim3(grid_for(hi - lo)), dim3(kBlock), 0, stream, lo, hi, plan.nbp, int(l == L - 1), S(relax),
                           dp.slice_ptr.p, dp.col.p, dp.nlower.p, w.LU.p, d, v);
    }
    for (int l = L - 2; l >= 0; --l) {
        const int lo = plan.level_ptr[l], hi = plan.level_ptr[l + 1];
        if (hi == lo) continue;
        hipLaunchKernelGGL((k_ilu_upper<S>), dim3(grid_for(hi - lo)), dim3(kBlock), 0, stream, lo, hi, plan.nbp, dp.slice_ptr.p, dp.col.p,
                           dp.nlower.p, dp.rowlen.p, w.LU.p, v);
    }
}

template <class S> void LinSolver::spmv(const S* x, S* y)
{
    hipLaunchKernelGGL((k_spmv<S, 0>), dim3(grid_for(plan.nb)), dim3(kBlock), 0, stream, plan.nb, plan.nbp, dp.slice_ptr.p, dp.col.p,
                       matrix<S>(), x, y, (const S*)nullptr, partials.p);
}

template <class S> SolveResult LinSolver::bicgstab(const opmgpu_params& prm)
{
    SolverWork<S>& w = work<S>();
    SolveResult res;
    const long n = long(3) * plan.nbp;
    const int gv = std::min(grid_for(n), kMaxRedBlocks);
    const int gs = grid_for(plan.nb);
    const double eps = sizeof(S) == 8 ? 1e-80 : 0.0;      // dune: real_type EPSILON = 1e-80 (0 in float)
    const double red = prm.linear_solver_reduction;
    const int maxit = prm.linear_solver_maxiter;
    auto finalize = [&](int op, int nblocks) {
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(kBlock), 0, stream, op, nblocks, eps, partials.p, scalars.p);
    };
    auto fetch = [&]() {
        OPMGPU_HIP(hipMemcpyAsync(h_scalars, scalars.p, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, stream));
        OPMGPU_HIP(hipStreamSynchronize(stream));
    };
    // x = 0, r = rt = b, p = v = 0
    w.x.zero(stream); w.p.zero(stream); w.v.zero(stream);
    OPMGPU_HIP(hipMemcpyAsync(w.r.p, w.b.p, n * sizeof(S), hipMemcpyDeviceToDevice, stream));
    OPMGPU_HIP(hipMemcpyAsync(w.rt.p, w.b.p, n * sizeof(S), hipMemcpyDeviceToDevice, stream));
    partials.zero(stream);
    hipLaunchKernelGGL((k_dot<S>), dim3(gv), dim3(kBlock), 0, stream, n, w.r.p, w.r.p, partials.p);
    finalize(FIN_INIT, gv);
    fetch();
    const double norm0 = std::sqrt(h_scalars[SC_NORM0_2]);
    double norm = norm0;
    if (!(norm0 == norm0)) { res.status = OPMGPU_EBREAKDOWN; return res; }
    double it = 0.0;
    if (norm < red * norm0 || norm < 1e-30) { res.converged = true; }
    else {
        for (it = 0.5; it < maxit; it += 0.5) {
            hipLaunchKernelGGL((k_update_p<S>), dim3(gv), dim3(kBlock), 0, stream, n, int(it < 1), scalars.p, w.r.p, w.v.p, w.p.p);
            ilu_apply<S>(w.p.p, w.y.p, prm.ilu_relaxation);
            hipLaunchKernelGGL((k_spmv<S, 1>), dim3(gs), dim3(kBlock), 0, stream, plan.nb, plan.nbp, dp.slice_ptr.p, dp.col.p, matrix<S>(),
                               w.y.p, w.v.p, w.rt.p, partials.p);
            finalize(FIN_H, gs);
            hipLaunchKernelGGL((k_update_xr<S, 1>), dim3(gv), dim3(kBlock), 0, stream, n, int(SC_ALPHA), scalars.p, w.y.p, w.v.p, w.rt.p,
                               w.x.p, w.r.p, partials.p);
            finalize(FIN_NORM1, gv);
            fetch();
            if (h_scalars[SC_FLAG] != 0.0) { res.status = OPMGPU_EBREAKDOWN; break; }
            norm = std::sqrt(h_scalars[SC_NORM2]);
            if (norm < red * norm0) { res.converged = true; break; }
            it += 0.5;
            ilu_apply<S>(w.r.p, w.y.p, prm.ilu_relaxation);
            hipLaunchKernelGGL((k_spmv<S, 2>), dim3(gs), dim3(kBlock), 0, stream, plan.nb, plan.nbp, dp.slice_ptr.p, dp.col.p, matrix<S>(),
                               w.y.p, w.t.p, w.r.p, partials.p);
            finalize(FIN_OMEGA, gs);
            hipLaunchKernelGGL((k_update_xr<S, 2>), dim3(gv), dim3(kBlock), 0, stream, n, int(SC_OMEGA), scalars.p, w.y.p, w.t.p, w.rt.p,
                               w.x.p, w.r.p, partials.p);
            finalize(FIN_NORM2, gv);
            fetch();
            norm = std::sqrt(h_scalars[SC_NORM2]);
            if (norm < red * norm0 || norm < 1e-30) { res.converged = true; break; }
            if (h_scalars[SC_FLAG] != 0.0 || !(norm == norm)) { res.status = OPMGPU_EBREAKDOWN; break; }
        }
    }
    it = std::min(double(maxit), it);
    res.iterations = int(std::ceil(it));
    res.reduction = norm0 > 0 ? norm / norm0 : 0.0;
    if (res.status == OPMGPU_OK && !res.converged && !prm.ignore_convergence_failure) res.status = OPMGPU_ELINSOLVE;   // ISTLSolver.hpp:358-368
    return res;
}

template <class S> void LinSolver::vec_in(const double* dsrc, int layout, S* d)
{
    hipLaunchKernelGGL((k_vec_in<S>), dim3(grid_for(plan.nb)), dim3(kBlock), 0, stream, plan.nb, plan.nbp, layout, dp.nat.p, dsrc, d);
}
template <class S> void LinSolver::vec_out(const S* d, int layout, double* ddst)
{
    hipLaunchKernelGGL((k_vec_out<S>), dim3(grid_for(plan.nb)), dim3(kBlock), 0, stream, plan.nb, plan.nbp, layout, dp.nat.p, d, ddst);
}
template <class S> void LinSolver::vec_from_host(const double* h, int layout, S* d)
{
    stage.ensure(std::max(size_t(plan.nnzb) * 9, size_t(3) * plan.nbp));
    OPMGPU_HIP(hipMemcpyAsync(stage.p, h, size_t(3) * plan.nb * sizeof(double), hipMemcpyHostToDevice, stream));
    vec_in<S>(stage.p, layout, d);
}
template <class S> void LinSolver::vec_to_host(const S* d, int layout, double* h)
{
    stage.ensure(std::max(size_t(plan.nnzb) * 9, size_t(3) * plan.nbp));
    vec_out<S>(d, layout, stage.p);
    OPMGPU_HIP(hipMemcpyAsync(h, stage.p, size_t(3) * plan.nb * sizeof(double), hipMemcpyDeviceToHost, stream));
    OPMGPU_HIP(hipStreamSynchronize(stream));
}

void LinSolver::get_matrix_bsr(const double* sell, double* val9)
{
    stage.ensure(std::max(size_t(plan.nnzb) * 9, size_t(3) * plan.nbp));
    hipLaunchKernelGGL((k_sell_to_bsr<double>), dim3(grid_for(plan.nnzb)), dim3(kBlock), 0, stream, plan.nnzb, dp.entry_of_block.p, sell, stage.p);
    OPMGPU_HIP(hipMemcpyAsync(val9, stage.p, size_t(plan.nnzb) * 9 * sizeof(double), hipMemcpyDeviceToHost, stream));
    OPMGPU_HIP(hipStreamSynchronize(stream));
}
template <class S> void LinSolver::get_lu_bsr(double* val9)
{
    stage.ensure(std::max(size_t(plan.nnzb) * 9, size_t(3) * plan.nbp));
    hipLaunchKernelGGL((k_sell_to_bsr<S>), dim3(grid_for(plan.nnzb)), dim3(kBlock), 0, stream, plan.nnzb, dp.entry_of_block.p, work<S>().LU.p, stage.p);
    OPMGPU_HIP(hipMemcpyAsync(val9, stage.p, size_t(plan.nnzb) * 9 * sizeof(double), hipMemcpyDeviceToHost, stream));
    OPMGPU_HIP(hipStreamSynchronize(stream));
}

template <class S> static double time_kernel_t(LinSolver& ls, int kernel, int reps, double relax)
{
    SolverWork<S>& w = ls.work<S>();
    const Plan& P = ls.plan;
    const long n = long(3) * P.nbp;
    const int gv = std::min(grid_for(n), kMaxRedBlocks);
    hipEvent_t e0, e1;
    OPMGPU_HIP(hipEventCreate(&e0)); OPMGPU_HIP(hipEventCreate(&e1));
    auto launch = [&]() {
        switch (kernel) {
        case OPMGPU_K_SPMV: ls.spmv<S>(w.p.p, w.v.p); break;
        case OPMGPU_K_ILU_APPLY: ls.ilu_apply<S>(w.p.p, w.y.p, relax); break;
        case OPMGPU_K_DOT: hipLaunchKernelGGL((k_dot<S>), dim3(gv), dim3(kBlock), 0, ls.stream, n, w.p.p, w.v.p, ls.partials.p); break;
        case OPMGPU_K_AXPY: hipLaunchKernelGGL((k_axpy<S>), dim3(gv), dim3(kBlock), 0, ls.stream, n, S(1e-3), w.p.p, w.t.p); break;
        case OPMGPU_K_STREAM_COPY: {
            const long n16 = long(P.nentries) * 9 * sizeof(S) / 16;
            hipLaunchKernelGGL(k_copy16, dim3(kMaxRedBlocks), dim3(kBlock), 0, ls.stream, n16, reinterpret_cast<const double2*>(ls.matrix<S>()),
                               reinterpret_cast<double2*>(w.LU.p));
            break;
        }
        default: break;
        }
    };
    launch();                                   // warm-up
    OPMGPU_HIP(hipEventRecord(e0, ls.stream));
    for (int i = 0; i < reps; ++i) launch();
    OPMGPU_HIP(hipEventRecord(e1, ls.stream));
    OPMGPU_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    OPMGPU_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return double(ms) / reps;
}

double LinSolver::time_kernel(int kernel, int reps, int single_precision)
{
    if (kernel == OPMGPU_K_ILU_FACTOR) {
        hipEvent_t e0, e1;
        OPMGPU_HIP(hipEventCreate(&e0)); OPMGPU_HIP(hipEventCreate(&e1));
        if (single_precision) factor<float>(); else factor<double>();
        OPMGPU_HIP(hipEventRecord(e0, stream));
        for (int i = 0; i < reps; ++i) { if (single_precision) factor<float>(); else factor<double>(); }
        OPMGPU_HIP(hipEventRecord(e1, stream));
        OPMGPU_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        OPMGPU_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        return double(ms) / reps;
    }
    return single_precision ? time_kernel_t<float>(*this, kernel, reps, 0.9) : time_kernel_t<double>(*this, kernel, reps, 0.9);
}

// explicit instantiations
#define OPMGPU_INST(S)                                                        \
    template void LinSolver::ensure_work<S>();                                 \
    template int LinSolver::factor<S>();                                       \
    template void LinSolver::ilu_apply<S>(const S*, S*, double);               \
    template void LinSolver::spmv<S>(const S*, S*);                            \
    template SolveResult LinSolver::bicgstab<S>(const opmgpu_params&);         \
    template void LinSolver::vec_from_host<S>(const double*, int, S*);         \
    template void LinSolver::vec_to_host<S>(const S*, int, double*);           \
    template void LinSolver::vec_in<S>(const double*, int, S*);                \
    template void LinSolver::vec_out<S>(const S*, int, double*);               \
    template void LinSolver::get_lu_bsr<S>(double*);
OPMGPU_INST(float)
OPMGPU_INST(double)

} // namespace opmgpu
