// pointilu.inl -- the reference's OWN second stage under CPR (included by linsolver.hip; opmgpu_params.cpr_reference_transform = 2).
//
// NewtonIterationBlackoilCPR.cpp:117-133 hands the row-transformed system L A -- as a SCALAR matrix in equation-major order,
// `DuneMatrix istlA(A)` with Dune::FieldMatrix<double, 1, 1> blocks -- to the CPR preconditioner, whose second stage is therefore a POINT
// ILU0 of the 3 nc x 3 nc matrix in the order [all pressure-equation rows | all second-equation rows | all third-equation rows], cells
// in the caller's order inside each group.  cpr_reference_transform = 1 keeps this library's 3x3-block ILU0 on the transformed system;
// value 2 builds exactly that point ILU0, so that a maintainer with an OPM install can compare linear iteration counts one to one
// (with ilu_ordering = OPMGPU_ORDER_NATURAL: dune's elimination order; MULTICOLOR colours the scalar graph instead: >= 6 levels).
//
// Nothing new on the device: the scalar system gets its own sparsity plan (plan.cpp is agnostic of what a "block" is), the point-ILU0
// kernels of elliptic.inl run on it, and three index maps connect it to the block layout -- `gather` (scalar entry -> component of a
// block in the SELL-64 block matrix), `vmap` (scalar row -> plane and row of the block vectors).  Double only, like the reference's
// CPR plug-in.  Cost: the update triplets of a scalar ILU0 are ~100 per row (3.6 GB of index lists at 10^6 cells, built once per
// pattern on the host): a comparison mode, not a production path.

__global__ __launch_bounds__(kBlock) void k_pt_gather_values(long n, const int64_t* __restrict__ src, const double* __restrict__ A, double* __restrict__ out)
{
    for (long e = blockIdx.x * long(kBlock) + threadIdx.x; e < n; e += long(gridDim.x) * kBlock) { const int64_t s = src[e]; out[e] = s >= 0 ? A[s] : 0.0; }
}
__global__ __launch_bounds__(kBlock) void k_pt_gather_vec(int n, const int32_t* __restrict__ vmap, const double* __restrict__ v, double* __restrict__ out)
{
    const int r = blockIdx.x * kBlock + threadIdx.x;
    if (r < n) { const int s = vmap[r]; out[r] = s >= 0 ? v[s] : 0.0; }
}
__global__ __launch_bounds__(kBlock) void k_pt_scatter_vec(int n, const int32_t* __restrict__ vmap, const double* __restrict__ in, double* __restrict__ v)
{
    const int r = blockIdx.x * kBlock + threadIdx.x;
    if (r < n) { const int s = vmap[r]; if (s >= 0) v[s] = in[r]; }
}

// scalar pattern, plan and index maps for the current block plan (once per pattern / ordering)
void LinSolver::point_ilu_setup()
{
    PointIlu& Q = pilu;
    if (Q.built && Q.for_nb == plan.nb && Q.for_nnzb == plan.nnzb && Q.for_ordering == cur_ordering && Q.for_plan_id == plan_id) return;
    const int nb = plan.nb;
    const long nnz2 = long(plan.nnzb) * 9;
    if (long(nb) * 3 > 0x7fffffffL || nnz2 > 0x7fffffffL) throw HipError(OPMGPU_EINVAL, "cpr_reference_transform = 2: the scalar system is too large for 32-bit indices");
    std::vector<int32_t> rowptr2(size_t(3) * nb + 1), col2(static_cast<size_t>(nnz2));
    std::vector<int32_t> blk_of(static_cast<size_t>(nnz2));          // caller-pattern entry of the scalar system -> block * 9 + component
    long k = 0;
    for (int eq = 0; eq < 3; ++eq)
        for (int i = 0; i < nb; ++i) {
            rowptr2[size_t(eq) * nb + i] = int32_t(k);
            for (int var = 0; var < 3; ++var)
                for (int s = plan.rowptr[i]; s < plan.rowptr[i + 1]; ++s) { col2[k] = var * nb + plan.col[s]; blk_of[k] = s * 9 + 3 * eq + var; ++k; }
        }
    rowptr2[size_t(3) * nb] = int32_t(k);
    const int st = build_plan(3 * nb, rowptr2.data(), col2.data(), cur_ordering, Q.plan);
    if (st != OPMGPU_OK) throw HipError(st, "cpr_reference_transform = 2: sparsity plan of the scalar system failed");
    Q.dp.upload(Q.plan, stream);
    std::vector<int64_t> gather(Q.plan.nentries, -1);
    for (int e2 = 0; e2 < Q.plan.nentries; ++e2) {
        const int32_t src = Q.plan.sell_src[e2];
        if (src < 0) continue;
        const int b = blk_of[src] / 9, comp = blk_of[src] % 9;
        const int32_t e1 = plan.entry_of_block[b];
        gather[e2] = Plan::val_index(e1, comp);
    }
    std::vector<int32_t> vmap(Q.plan.nbp, -1);
    for (int r2 = 0; r2 < 3 * nb; ++r2) { const int u = Q.plan.nat[r2]; vmap[r2] = (u / nb) * plan.nbp + plan.pos[u % nb]; }
    Q.gather.upload(gather, stream); Q.vmap.upload(vmap, stream);
    Q.val.alloc(Q.plan.nentries); Q.lu.alloc(Q.plan.nentries); Q.d.alloc(Q.plan.nbp); Q.v.alloc(Q.plan.nbp);
    Q.d.zero(stream); Q.v.zero(stream);
    OPMGPU_HIP(hipStreamSynchronize(stream));
    Q.built = true; Q.for_nb = plan.nb; Q.for_nnzb = plan.nnzb; Q.for_ordering = cur_ordering; Q.for_plan_id = plan_id;
}

// point ILU0 of the (transformed) double matrix: gather the scalar values, factorise level by level
void LinSolver::point_ilu_factor()
{
    point_ilu_setup();
    PointIlu& Q = pilu;
    const long ne = Q.plan.nentries;
    hipLaunchKernelGGL(k_pt_gather_values, dim3(std::min(grid_for(ne), kMaxRedBlocks)), dim3(kBlock), 0, stream, ne, (const int64_t*)Q.gather.p, (const double*)Ad.p, Q.val.p);
    for (int l = 0; l < Q.plan.nlevels; ++l) {
        const int lo = Q.plan.level_ptr[l], hi = Q.plan.level_ptr[l + 1];
        if (hi == lo) continue;
        hipLaunchKernelGGL((k_pilu_factor<double>), dim3(grid_for(hi - lo)), dim3(kBlock), 0, stream, lo, hi, Q.dp.slice_ptr.p, Q.dp.col.p, Q.dp.nlower.p, Q.dp.rowlen.p,
                           Q.dp.trip_ptr.p, Q.dp.trip_l.p, Q.dp.trip_u.p, Q.dp.trip_t.p, (const double*)Q.val.p, Q.lu.p);
    }
}

// v = relax * (L U)^-1 d on block-layout vectors (three planes of nbp)
void LinSolver::point_ilu_apply(const double* d, double* v, double relax)
{
    PointIlu& Q = pilu;
    const int n2 = Q.plan.nb, L = Q.plan.nlevels, n0 = Q.plan.level_ptr[1];
    hipLaunchKernelGGL(k_pt_gather_vec, dim3(grid_for(n2)), dim3(kBlock), 0, stream, n2, (const int32_t*)Q.vmap.p, d, Q.d.p);
    if (L == 1) {
        hipLaunchKernelGGL((k_pilu_lower<double>), dim3(grid_for(n0)), dim3(kBlock), 0, stream, 0, n0, 0, 1, relax, Q.dp.slice_ptr.p, Q.dp.col.p, Q.dp.nlower.p, (const double*)Q.lu.p,
                           (const double*)Q.d.p, Q.v.p);
    } else {
        for (int l = 1; l < L; ++l) {
            const int lo = Q.plan.level_ptr[l], hi = Q.plan.level_ptr[l + 1];
            if (hi > lo) hipLaunchKernelGGL((k_pilu_lower<double>), dim3(grid_for(hi - lo)), dim3(kBlock), 0, stream, lo, hi, n0, int(l == L - 1), relax, Q.dp.slice_ptr.p, Q.dp.col.p,
                                            Q.dp.nlower.p, (const double*)Q.lu.p, (const double*)Q.d.p, Q.v.p);
        }
        for (int l = L - 2; l >= 0; --l) {
            const int lo = Q.plan.level_ptr[l], hi = Q.plan.level_ptr[l + 1];
            if (hi > lo) hipLaunchKernelGGL((k_pilu_upper<double>), dim3(grid_for(hi - lo)), dim3(kBlock), 0, stream, lo, hi, n0, relax, Q.dp.slice_ptr.p, Q.dp.col.p, Q.dp.nlower.p,
                                            Q.dp.rowlen.p, (const double*)Q.lu.p, (const double*)Q.d.p, Q.v.p);
        }
    }
    hipLaunchKernelGGL(k_pt_scatter_vec, dim3(grid_for(n2)), dim3(kBlock), 0, stream, n2, (const int32_t*)Q.vmap.p, (const double*)Q.v.p, v);
}
