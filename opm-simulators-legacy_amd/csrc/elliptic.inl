// elliptic.inl -- the elliptic (pressure) part of the CPR preconditioner solved by an inner Krylov method, the way the reference's
// CPR plug-in is documented to do it (included by linsolver.hip).
//
// NewtonIterationBlackoilCPR.hpp:59-63 names the knobs: cpr_use_amg (default FALSE: the elliptic system A_p is preconditioned by an
// ILU0), cpr_use_bicgstab (default true: BiCGStab, else CG, "for elliptic part"), cpr_relax, cpr_ilu_n.  The solve itself lives in
// opm-simulators' CPRPreconditioner (external, not under /root/reference): x_p starts from zero, the Krylov method runs to a relative
// reduction cpr_solver_tol of || b_p - A_p x_p || or cpr_max_ell_iter iterations -- the two numbers are this library's recollection of
// that file and are parameters here (opmgpu_params.cpr_solver_tol / cpr_max_ell_iter).
//
// A_p is level 0 of the pressure hierarchy (amg.hpp): scalar SELL-64 with the block plan's structure, so the plan's ILU0 data (levels of
// the multicolour ordering, lower / upper split, update triplets) serve a POINT ILU0 of A_p unchanged.  With device wells the level is
// bordered by one bhp unknown per well; the ILU0 covers the cell rows, a border row is preconditioned by its own diagonal.
// Decomposed runs: the inner solve is rank-local like the AMG cycle (additive Schwarz: ghost rows are identity rows, no collective).
//
// The method is driven from the host: every scalar product is a partial-sum launch + a one-workgroup fixed-order reduction + one read of
// the host-mapped word buffer (fetch_words).  Four reads per BiCGStab iteration, ~10 us each, against ~100 us of kernels at 1 M cells.

template <class S>
__global__ __launch_bounds__(kBlock) void k_pilu_factor(int lo, int hi, const int32_t* __restrict__ slice_ptr, const int32_t* __restrict__ col,
                                                        const int16_t* __restrict__ nlower, const int16_t* __restrict__ rowlen, const int32_t* __restrict__ trip_ptr,
                                                        const int32_t* __restrict__ trip_l, const int32_t* __restrict__ trip_u, const int32_t* __restrict__ trip_t,
                                                        const S* __restrict__ A, S* __restrict__ lu)
{
    const int row = lo + blockIdx.x * kBlock + threadIdx.x;
    if (row >= hi) return;
    const int base = slice_ptr[row >> 6], lane = row & 63, nl = nlower[row], len = rowlen[row];
    for (int k = 0; k < len; ++k) { const int32_t e = (base + k) * 64 + lane; lu[e] = A[e]; }
    int tp = trip_ptr[row];
    const int te = trip_ptr[row + 1];
    for (int k = 0; k < nl; ++k) {                  // IKJ in dune's bilu0 order: L_ij = a_ij / u_jj, then row i -= L_ij * (row j's upper part)
        const int32_t e = (base + k) * 64 + lane;
        const int j = col[e];
        const int32_t ej = (slice_ptr[j >> 6] + nlower[j]) * 64 + (j & 63);
        const S L = lu[e] * lu[ej];                 // lu[ej] already holds 1 / u_jj (row j belongs to an earlier level)
        lu[e] = L;
        while (tp < te && trip_l[tp] == e) { lu[trip_t[tp]] -= L * lu[trip_u[tp]]; ++tp; }
    }
    const int32_t ed = (base + nl) * 64 + lane;
    const S d = lu[ed];
    lu[ed] = (d != S(0) && d == d) ? S(1) / d : S(0);          // a zero pivot leaves the row out of the sweeps; the Krylov method then reports what it attains
}

// forward sweep of one level l >= 1 (level-0 rows have no lower entries: y_j = w d_j there), cf. k_ilu_lower
template <class S>
__global__ __launch_bounds__(kBlock) void k_pilu_lower(int lo, int hi, int n0, int top, S w, const int32_t* __restrict__ slice_ptr, const int32_t* __restrict__ col,
                                                       const int16_t* __restrict__ nlower, const S* __restrict__ lu, const S* __restrict__ d, S* __restrict__ v)
{
    const int row = lo + blockIdx.x * kBlock + threadIdx.x;
    if (row >= hi) return;
    const int base = slice_ptr[row >> 6], lane = row & 63, nl = nlower[row];
    S r = w * d[row];
    for (int k = 0; k < nl; ++k) {
        const int32_t e = (base + k) * 64 + lane;
        const int cc = col[e];
        r -= lu[e] * (cc < n0 ? w * d[cc] : v[cc]);
    }
    if (top) r *= lu[(base + nl) * 64 + lane];
    v[row] = r;
}
// backward sweep of one level, cf. k_ilu_upper
template <class S>
__global__ __launch_bounds__(kBlock) void k_pilu_upper(int lo, int hi, int n0, S w, const int32_t* __restrict__ slice_ptr, const int32_t* __restrict__ col,
                                                       const int16_t* __restrict__ nlower, const int16_t* __restrict__ rowlen, const S* __restrict__ lu,
                                                       const S* __restrict__ d, S* __restrict__ v)
{
    const int row = lo + blockIdx.x * kBlock + threadIdx.x;
    if (row >= hi) return;
    const int base = slice_ptr[row >> 6], lane = row & 63, nl = nlower[row], len = rowlen[row];
    S r = row < n0 ? w * d[row] : v[row];
    for (int k = nl + 1; k < len; ++k) { const int32_t e = (base + k) * 64 + lane; r -= lu[e] * v[col[e]]; }
    v[row] = lu[(base + nl) * 64 + lane] * r;
}
// border rows (one bhp unknown per well): their own diagonal
template <class S>
__global__ __launch_bounds__(kBlock) void k_ell_border_diag(int n, int nw, S w, const S* __restrict__ dinv, const S* __restrict__ d, S* __restrict__ v)
{
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k < nw) v[n + k] = w * dinv[n + k] * d[n + k];
}

struct EllBorder { int nw, gcells; const int32_t *connpos, *perf_row, *perf_of_row, *perf_well; };

// y = A_p x on the (bordered) level 0: `gcells` workgroups for the cell rows + one workgroup per well row
template <class S>
__global__ __launch_bounds__(kBlock) void k_ell_spmv(int n, const int32_t* __restrict__ slice_ptr, const int32_t* __restrict__ col, const S* __restrict__ val,
                                                     const S* __restrict__ x, S* __restrict__ y, EllBorder B, const S* __restrict__ bcol, const S* __restrict__ crow,
                                                     const S* __restrict__ dw)
{
    __shared__ double sm[4];
    if (B.nw && int(blockIdx.x) >= B.gcells) {
        const int k = blockIdx.x - B.gcells;
        double acc[1] = { 0.0 };
        for (int j = B.connpos[k] + threadIdx.x; j < B.connpos[k + 1]; j += kBlock) acc[0] += double(crow[j]) * double(x[B.perf_row[j]]);
        block_sum<1>(acc, sm);
        if (threadIdx.x == 0) y[n + k] = S(acc[0] + double(dw[k]) * double(x[n + k]));
        return;
    }
    const int row = blockIdx.x * kBlock + threadIdx.x;
    if (row >= n) return;
    const int base = slice_ptr[row >> 6], width = slice_ptr[(row >> 6) + 1] - base, lane = row & 63;
    S acc = 0;
    for (int k = 0; k < width; ++k) { const long e = long(base + k) * 64 + lane; acc += val[e] * x[col[e]]; }
    if (B.nw) { const int j = B.perf_of_row[row]; if (j >= 0) acc += bcol[j] * x[n + B.perf_well[j]]; }
    y[row] = acc;
}

// partial sums of <a, b> and (c != nullptr) <c, d>: parts[workgroup], parts[G + workgroup]
template <class S>
__global__ __launch_bounds__(kBlock) void k_ell_dot2(long n, const S* __restrict__ a, const S* __restrict__ b, const S* __restrict__ c, const S* __restrict__ d,
                                                     double* __restrict__ parts)
{
    __shared__ double sm[8];
    double acc[2] = { 0.0, 0.0 };
    for (long i = blockIdx.x * long(kBlock) + threadIdx.x; i < n; i += long(gridDim.x) * kBlock) {
        acc[0] += double(a[i]) * double(b[i]);
        if (c) acc[1] += double(c[i]) * double(d[i]);
    }
    block_sum<2>(acc, sm);
    if (threadIdx.x == 0) { parts[blockIdx.x] = acc[0]; parts[gridDim.x + blockIdx.x] = acc[1]; }
}
// p = r + beta (p - omega v)      (first iteration: beta = 0 gives p = r)
template <class S>
__global__ __launch_bounds__(kBlock) void k_ell_p_update(long n, S beta, S omega, const S* __restrict__ r, const S* __restrict__ v, S* __restrict__ p)
{
    for (long i = blockIdx.x * long(kBlock) + threadIdx.x; i < n; i += long(gridDim.x) * kBlock) p[i] = r[i] + beta * (p[i] - omega * v[i]);
}
// x += a y ; r -= a q ; partial sums of <r, r> and (rt != nullptr) <rt, r>
template <class S>
__global__ __launch_bounds__(kBlock) void k_ell_xr(long n, S a, const S* __restrict__ y, const S* __restrict__ q, S* __restrict__ x, S* __restrict__ r,
                                                   const S* __restrict__ rt, double* __restrict__ parts)
{
    __shared__ double sm[8];
    double acc[2] = { 0.0, 0.0 };
    for (long i = blockIdx.x * long(kBlock) + threadIdx.x; i < n; i += long(gridDim.x) * kBlock) {
        x[i] += a * y[i];
        const S ri = r[i] - a * q[i];
        r[i] = ri;
        acc[0] += double(ri) * double(ri);
        if (rt) acc[1] += double(rt[i]) * double(ri);
    }
    block_sum<2>(acc, sm);
    if (threadIdx.x == 0) { parts[blockIdx.x] = acc[0]; parts[gridDim.x + blockIdx.x] = acc[1]; }
}
// p = y + beta p  (CG)
template <class S>
__global__ __launch_bounds__(kBlock) void k_ell_cg_p(long n, S beta, const S* __restrict__ y, S* __restrict__ p)
{
    for (long i = blockIdx.x * long(kBlock) + threadIdx.x; i < n; i += long(gridDim.x) * kBlock) p[i] = y[i] + beta * p[i];
}
// out[k] = sum of parts[k * G .. k * G + G) in a fixed order; one workgroup per k
__global__ __launch_bounds__(kBlock) void k_ell_reduce(const double* __restrict__ parts, int G, double* __restrict__ out)
{
    __shared__ double sm[4];
    double acc[1] = { 0.0 };
    for (int i = threadIdx.x; i < G; i += kBlock) acc[0] += parts[long(blockIdx.x) * G + i];
    block_sum<1>(acc, sm);
    if (threadIdx.x == 0) out[blockIdx.x] = acc[0];
}

// point ILU0 of the current A_p (levels[0].val): once per matrix, behind cpr_prepare's pass that wrote the values
template <class S> void LinSolver::elliptic_factor()
{
    SolverWork<S>& w = work<S>();
    AmgLevel<S>& L0 = *w.amg->levels[0];
    w.plu.ensure(size_t(plan.nentries));
    for (int l = 0; l < plan.nlevels; ++l) {
        const int lo = plan.level_ptr[l], hi = plan.level_ptr[l + 1];
        if (hi == lo) continue;
        hipLaunchKernelGGL((k_pilu_factor<S>), dim3(grid_for(hi - lo)), dim3(kBlock), 0, stream, lo, hi, dp.slice_ptr.p, dp.col.p, dp.nlower.p, dp.rowlen.p,
                           dp.trip_ptr.p, dp.trip_l.p, dp.trip_u.p, dp.trip_t.p, (const S*)L0.val.p, w.plu.p);
    }
}

// levels[0].x = approximate solution of A_p x = levels[0].b from x = 0 (see the head of this file)
template <class S> void LinSolver::elliptic_solve()
{
    SolverWork<S>& w = work<S>();
    AmgHierarchy<S>& H = *w.amg;
    AmgLevel<S>& L0 = *H.levels[0];
    const int n = L0.n, nw = L0.nw;
    const long N = long(n) + nw;
    const int G = std::min(grid_for(N), kMaxPart);
    const int gcells = grid_for(n);
    // work vectors: b (the right-hand side, kept: the AMG cycle uses levels[0].b as its own input), x, r, rt, p, v, t, y
    const size_t stride = (size_t(N) + 63) / 64 * 64;
    w.ellv.ensure(8 * stride);
    S* const bb = w.ellv.p; S* const x = bb + stride; S* const r = x + stride; S* const rt = r + stride; S* const p = rt + stride;
    S* const v = p + stride; S* const t = v + stride; S* const y = t + stride;
    ell_parts.ensure(size_t(2) * kMaxPart + 8);
    double* const parts = ell_parts.p; double* const red = parts + size_t(2) * kMaxPart;
    EllBorder B = { nw, gcells, L0.b_connpos, L0.b_perf_row, L0.b_perf_of_row, L0.b_perf_well };
    const S* const bcol = L0.val.p + L0.nentries; const S* const crow = bcol + L0.nperf; const S* const dw = crow + L0.nperf;
    const S wrel = S(ell.relax);
    auto spmv = [&](const S* in, S* out) {
        hipLaunchKernelGGL((k_ell_spmv<S>), dim3(gcells + nw), dim3(kBlock), 0, stream, n, L0.slice_ptr, L0.col, (const S*)L0.val.p, in, out, B, bcol, crow, dw);
    };
    auto precond = [&](const S* d, S* out) {
        if (ell.use_amg) {
            // one V-cycle with levels[0].b = d; the cycle may swap its x buffers, so the result is read from levels[0].x afterwards
            OPMGPU_HIP(hipMemcpyAsync(L0.b.p, d, size_t(N) * sizeof(S), hipMemcpyDeviceToDevice, stream));
            H.vcycle(nullptr, false);
            OPMGPU_HIP(hipMemcpyAsync(out, H.levels[0]->x.p, size_t(N) * sizeof(S), hipMemcpyDeviceToDevice, stream));
            return;
        }
        const int L = plan.nlevels, n0 = plan.level_ptr[1];
        if (L == 1) {
            hipLaunchKernelGGL((k_pilu_lower<S>), dim3(grid_for(n0)), dim3(kBlock), 0, stream, 0, n0, 0, 1, wrel, dp.slice_ptr.p, dp.col.p, dp.nlower.p, (const S*)w.plu.p, d, out);
        } else {
            for (int l = 1; l < L; ++l) {
                const int lo = plan.level_ptr[l], hi = plan.level_ptr[l + 1];
                if (hi > lo) hipLaunchKernelGGL((k_pilu_lower<S>), dim3(grid_for(hi - lo)), dim3(kBlock), 0, stream, lo, hi, n0, int(l == L - 1), wrel, dp.slice_ptr.p, dp.col.p,
                                                dp.nlower.p, (const S*)w.plu.p, d, out);
            }
            for (int l = L - 2; l >= 0; --l) {
                const int lo = plan.level_ptr[l], hi = plan.level_ptr[l + 1];
                if (hi > lo) hipLaunchKernelGGL((k_pilu_upper<S>), dim3(grid_for(hi - lo)), dim3(kBlock), 0, stream, lo, hi, n0, wrel, dp.slice_ptr.p, dp.col.p, dp.nlower.p,
                                                dp.rowlen.p, (const S*)w.plu.p, d, out);
            }
        }
        if (nw) hipLaunchKernelGGL((k_ell_border_diag<S>), dim3(grid_for(nw)), dim3(kBlock), 0, stream, n, nw, wrel, (const S*)L0.dinv.p, d, out);
    };
    // the two sums a step needs, on the host
    auto fetch2 = [&](double& s0, double& s1) {
        hipLaunchKernelGGL(k_ell_reduce, dim3(2), dim3(kBlock), 0, stream, (const double*)parts, G, red);
        const uint32_t* h = fetch_words(red, 4);
        std::memcpy(&s0, h, 8); std::memcpy(&s1, h + 2, 8);
    };
    ++ell_solves;
    OPMGPU_HIP(hipMemcpyAsync(bb, L0.b.p, size_t(N) * sizeof(S), hipMemcpyDeviceToDevice, stream));
    OPMGPU_HIP(hipMemsetAsync(x, 0, size_t(N) * sizeof(S), stream));
    OPMGPU_HIP(hipMemcpyAsync(r, bb, size_t(N) * sizeof(S), hipMemcpyDeviceToDevice, stream));
    auto finish = [&]() { OPMGPU_HIP(hipMemcpyAsync(H.levels[0]->x.p, x, size_t(N) * sizeof(S), hipMemcpyDeviceToDevice, stream)); };
    double norm0_2 = 0.0, dummy = 0.0;
    hipLaunchKernelGGL((k_ell_dot2<S>), dim3(G), dim3(kBlock), 0, stream, N, (const S*)r, (const S*)r, (const S*)nullptr, (const S*)nullptr, parts);
    fetch2(norm0_2, dummy);
    if (!(norm0_2 > 0.0)) { finish(); return; }             // zero (or NaN) right-hand side: x = 0
    const double thresh2 = ell.tol * ell.tol * norm0_2;
    if (ell.bicgstab) {
        // Dune::BiCGSTABSolver::apply: rt = r0; per iteration two preconditioner applications, convergence tested after each half step
        OPMGPU_HIP(hipMemcpyAsync(rt, r, size_t(N) * sizeof(S), hipMemcpyDeviceToDevice, stream));
        OPMGPU_HIP(hipMemsetAsync(p, 0, size_t(N) * sizeof(S), stream));
        OPMGPU_HIP(hipMemsetAsync(v, 0, size_t(N) * sizeof(S), stream));
        double rho = 1.0, alpha = 1.0, omega = 1.0, rho_new = norm0_2;      // <rt, r0> = ||r0||^2
        for (int it = 1; it <= ell.maxit; ++it) {
            ++ell_iterations;
            if (rho_new == 0.0 || omega == 0.0 || !(rho_new == rho_new)) break;          // breakdown: keep what has been attained
            const double beta = it == 1 ? 0.0 : (rho_new / rho) * (alpha / omega);
            hipLaunchKernelGGL((k_ell_p_update<S>), dim3(G), dim3(kBlock), 0, stream, N, S(beta), S(omega), (const S*)r, (const S*)v, p);
            precond(p, y);
            spmv(y, v);
            double h = 0.0;
            hipLaunchKernelGGL((k_ell_dot2<S>), dim3(G), dim3(kBlock), 0, stream, N, (const S*)rt, (const S*)v, (const S*)nullptr, (const S*)nullptr, parts);
            fetch2(h, dummy);
            if (h == 0.0 || !(h == h)) break;
            alpha = rho_new / h;
            double n2 = 0.0;
            hipLaunchKernelGGL((k_ell_xr<S>), dim3(G), dim3(kBlock), 0, stream, N, S(alpha), (const S*)y, (const S*)v, x, r, (const S*)nullptr, parts);
            fetch2(n2, dummy);
            if (n2 <= thresh2) break;
            precond(r, y);
            spmv(y, t);
            double tr = 0.0, tt = 0.0;
            hipLaunchKernelGGL((k_ell_dot2<S>), dim3(G), dim3(kBlock), 0, stream, N, (const S*)t, (const S*)r, (const S*)t, (const S*)t, parts);
            fetch2(tr, tt);
            if (tt == 0.0 || !(tt == tt)) break;
            omega = tr / tt;
            rho = rho_new;
            hipLaunchKernelGGL((k_ell_xr<S>), dim3(G), dim3(kBlock), 0, stream, N, S(omega), (const S*)y, (const S*)t, x, r, (const S*)rt, parts);
            fetch2(n2, rho_new);
            if (n2 <= thresh2) break;
        }
    } else {
        // Dune::CGSolver::apply with the same preconditioner (the reference offers it for an elliptic part it takes to be symmetric)
        precond(r, y);
        OPMGPU_HIP(hipMemcpyAsync(p, y, size_t(N) * sizeof(S), hipMemcpyDeviceToDevice, stream));
        double rholast = 0.0;
        hipLaunchKernelGGL((k_ell_dot2<S>), dim3(G), dim3(kBlock), 0, stream, N, (const S*)p, (const S*)r, (const S*)nullptr, (const S*)nullptr, parts);
        fetch2(rholast, dummy);
        for (int it = 1; it <= ell.maxit; ++it) {
            ++ell_iterations;
            spmv(p, v);
            double pq = 0.0;
            hipLaunchKernelGGL((k_ell_dot2<S>), dim3(G), dim3(kBlock), 0, stream, N, (const S*)p, (const S*)v, (const S*)nullptr, (const S*)nullptr, parts);
            fetch2(pq, dummy);
            if (pq == 0.0 || !(pq == pq)) break;
            const double lambda = rholast / pq;
            double n2 = 0.0;
            hipLaunchKernelGGL((k_ell_xr<S>), dim3(G), dim3(kBlock), 0, stream, N, S(lambda), (const S*)p, (const S*)v, x, r, (const S*)nullptr, parts);
            fetch2(n2, dummy);
            if (n2 <= thresh2) break;
            precond(r, y);
            double rho = 0.0;
            hipLaunchKernelGGL((k_ell_dot2<S>), dim3(G), dim3(kBlock), 0, stream, N, (const S*)y, (const S*)r, (const S*)nullptr, (const S*)nullptr, parts);
            fetch2(rho, dummy);
            if (rholast == 0.0 || !(rho == rho)) break;
            hipLaunchKernelGGL((k_ell_cg_p<S>), dim3(G), dim3(kBlock), 0, stream, N, S(rho / rholast), (const S*)y, p);
            rholast = rho;
        }
    }
    if (ell.use_amg) OPMGPU_HIP(hipMemcpyAsync(H.levels[0]->b.p, bb, size_t(N) * sizeof(S), hipMemcpyDeviceToDevice, stream));      // callers may read the right-hand side again
    finish();
}
