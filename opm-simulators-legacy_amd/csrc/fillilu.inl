// fillilu.inl -- block ILU(n) with level-of-fill (included by linsolver.hip): opmgpu_params.cpr_ilu_n (NewtonIterationBlackoilCPR.hpp:61, "use
// ILU(n) for preconditioning of the linear system": Dune::SeqILUn as the CPR plug-in's second stage) and opmgpu_params.ilu_fillin_level
// (the interleaved solver's `ilu_fillin_level`, ISTLSolver.hpp:205).
//
// dune-istl is not under /root/reference ("parity unpinned", like bilu0).  The symbolic phase (plan.cpp, build_fill_pattern) is the textbook
// level-of-fill rule: rows in the caller's order, entries of A at level 0, lev(i,j) = min over the eliminations k < min(i,j) of
// lev(i,k) + lev(k,j) + 1, kept when <= n.  For n = 1 that is also what dune-istl's bilu_decomposition(A, n, ILU) builds (fill from two
// entries of A); its generation bookkeeping for n >= 2, as far as recalled, keeps more entries than the sum rule -- not restated here.
// The numeric phase is an ILU0 on that pattern, so nothing new runs on the device: the extended pattern gets its own sparsity plan
// (levels or colours of the EXTENDED graph), A's blocks are gathered into it (fill entries start at zero), and the block kernels of the
// ILU0 factorise and sweep it.  Vectors are permuted between the two plans' row orders around each application.
// With ilu_ordering = OPMGPU_ORDER_NATURAL the elimination order is dune's; MULTICOLOR colours the extended graph (same pattern,
// another -- more parallel -- elimination order: ~8 colours on a 7-point grid at n = 1 instead of ~300 wavefront levels).

template <class S>
__global__ __launch_bounds__(kBlock) void k_fill_gather_blocks(int ne, const int32_t* __restrict__ src, const S* __restrict__ A, S* __restrict__ out)
{
    for (int e = blockIdx.x * kBlock + threadIdx.x; e < ne; e += gridDim.x * kBlock) {
        const int32_t s = src[e];
        S* o = out + long(e >> 6) * 576 + (e & 63);
        if (s >= 0) {
            const S* a = A + long(s >> 6) * 576 + (s & 63);
#pragma unroll
            for (int q = 0; q < 9; ++q) o[q * 64] = a[q * 64];
        } else {
#pragma unroll
            for (int q = 0; q < 9; ++q) o[q * 64] = S(0);
        }
    }
}
// out[q][r] = in[q][map[r]] (gather) / out[q][map[r]] = in[q][r] (scatter) on three planes of nbp
template <class S, bool SCATTER>
__global__ __launch_bounds__(kBlock) void k_fill_permute(int nb, int nbp, const int32_t* __restrict__ map, const S* __restrict__ in, S* __restrict__ out,
                                                         const SolveCtl* __restrict__ ctl)
{
    if (ctl && ctl->done) return;
    const int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= nb) return;
    const int m = map[r];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        if (SCATTER) out[q * long(nbp) + m] = in[q * long(nbp) + r];
        else out[q * long(nbp) + r] = in[q * long(nbp) + m];
    }
}

void LinSolver::fill_setup()
{
    FillIlu& F = fill;
    if (F.built && F.for_level == fill_level && F.for_ordering == cur_ordering && F.for_plan_id == plan_id) return;
    std::vector<int32_t> rowptr2, col2, src2;
    build_fill_pattern(plan, fill_level, rowptr2, col2, src2);
    if (long(rowptr2[plan.nb]) > 12L * plan.nnzb) throw HipError(OPMGPU_EINVAL, "ILU(n): the filled pattern has more than 12 x the blocks of the matrix: lower cpr_ilu_n / ilu_fillin_level");
    if (rowptr2[plan.nb] / 64 > (0x7fffffff / 576)) throw HipError(OPMGPU_EINVAL, "ILU(n): the filled pattern is too large for 32-bit entry ids");
    const int st = build_plan(plan.nb, rowptr2.data(), col2.data(), cur_ordering, F.plan);
    if (st != OPMGPU_OK) throw HipError(st, "ILU(n): sparsity plan of the filled pattern failed");
    F.dp.upload(F.plan, stream);
    std::vector<int32_t> src(F.plan.nentries, -1), vmap(plan.nb);
    for (int e2 = 0; e2 < F.plan.nentries; ++e2) {
        const int32_t b2 = F.plan.sell_src[e2];
        if (b2 >= 0 && src2[b2] >= 0) src[e2] = plan.entry_of_block[src2[b2]];
    }
    for (int r2 = 0; r2 < plan.nb; ++r2) vmap[r2] = plan.pos[F.plan.nat[r2]];
    F.src.upload(src, stream); F.vmap.upload(vmap, stream);
    OPMGPU_HIP(hipStreamSynchronize(stream));
    F.nnzb_filled = rowptr2[plan.nb];
    F.built = true; F.for_level = fill_level; F.for_ordering = cur_ordering; F.for_plan_id = plan_id;
    F.wf.val.release(); F.wf.lu.release(); F.wd.val.release(); F.wd.lu.release();
}

template <class S> int LinSolver::fill_factor(bool wait)
{
    fill_setup();
    FillIlu& F = fill;
    FillWork<S>& W = F.work<S>();
    const size_t nval = size_t(F.plan.nentries / 64) * 576;
    if (W.val.n != nval) { W.val.alloc(nval); W.lu.alloc(nval); W.d.alloc(size_t(3) * plan.nbp); W.v.alloc(size_t(3) * plan.nbp); W.d.zero(stream); W.v.zero(stream); }
    KtScope kts(kt, KT_ILU_FACTOR);
    flags.zero(stream);
    hipLaunchKernelGGL((k_fill_gather_blocks<S>), dim3(std::min(grid_for(F.plan.nentries), 8 * kMaxPart)), dim3(kBlock), 0, stream, F.plan.nentries, (const int32_t*)F.src.p,
                       matrix<S>(), W.val.p);
    for (int l = 0; l < F.plan.nlevels; ++l) {
        const int lo = F.plan.level_ptr[l], hi = F.plan.level_ptr[l + 1];
        if (hi == lo) continue;
        hipLaunchKernelGGL((k_ilu_factor<S>), dim3(grid_for(hi - lo)), dim3(kBlock), 0, stream, lo, hi, F.dp.slice_ptr.p, F.dp.col.p, F.dp.nlower.p, F.dp.trip_ptr.p,
                           F.dp.trip_l.p, F.dp.trip_u.p, F.dp.trip_t.p, (const S*)W.val.p, F.dp.rowlen.p, W.lu.p, flags.p, (const int8_t*)F.dp.simple.p, 0);
    }
    OPMGPU_HIP(hipMemcpyAsync(h_flags, flags.p, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    if (!wait) return OPMGPU_OK;
    OPMGPU_HIP(hipStreamSynchronize(stream));
    return factor_status();
}

template <class S> void LinSolver::fill_apply(const S* d, S* v, double relax, const SolveCtl* ctl)
{
    FillIlu& F = fill;
    FillWork<S>& W = F.work<S>();
    const int nb = plan.nb, nbp = plan.nbp, L = F.plan.nlevels, n0 = F.plan.level_ptr[1];
    hipLaunchKernelGGL((k_fill_permute<S, false>), dim3(grid_for(nb)), dim3(kBlock), 0, stream, nb, nbp, (const int32_t*)F.vmap.p, d, W.d.p, ctl);
    if (L == 1) {
        hipLaunchKernelGGL((k_ilu_lower<S>), dim3(grid8_for(n0)), dim3(kBlock), 0, stream, xcd_mode(), 0, n0, 0, nbp, 1, S(relax), F.dp.slice_ptr.p, F.dp.col.p, F.dp.nlower.p,
                           (const S*)W.lu.p, (const S*)W.d.p, W.v.p, ctl);
    } else {
        for (int l = 1; l < L; ++l) {
            const int lo = F.plan.level_ptr[l], hi = F.plan.level_ptr[l + 1];
            if (hi > lo) hipLaunchKernelGGL((k_ilu_lower<S>), dim3(grid8_for(hi - lo)), dim3(kBlock), 0, stream, xcd_mode(), lo, hi, n0, nbp, int(l == L - 1), S(relax), F.dp.slice_ptr.p,
                                            F.dp.col.p, F.dp.nlower.p, (const S*)W.lu.p, (const S*)W.d.p, W.v.p, ctl);
        }
        for (int l = L - 2; l >= 0; --l) {
            const int lo = F.plan.level_ptr[l], hi = F.plan.level_ptr[l + 1];
            if (hi > lo) hipLaunchKernelGGL((k_ilu_upper<S>), dim3(grid8_for(hi - lo)), dim3(kBlock), 0, stream, xcd_mode(), lo, hi, n0, nbp, S(relax), F.dp.slice_ptr.p, F.dp.col.p,
                                            F.dp.nlower.p, F.dp.rowlen.p, (const S*)W.lu.p, (const S*)W.d.p, W.v.p, ctl, (const int8_t*)F.dp.simple.p, (const S*)W.val.p);
        }
    }
    hipLaunchKernelGGL((k_fill_permute<S, true>), dim3(grid_for(nb)), dim3(kBlock), 0, stream, nb, nbp, (const int32_t*)F.vmap.p, (const S*)W.v.p, v, ctl);
}
