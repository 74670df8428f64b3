// plan.hpp -- host-side sparsity plan for the device block solver.
//
// Replaces, for the GPU, what the reference does per Newton iteration on the CPU:
//   * formInterleavedSystem's pattern build (NewtonIterationBlackoilInterleaved.cpp:110-155),
//   * ParallelOverlappingILU0's reordering / L-U split (constructed at ISTLSolver.hpp:202-211).
// It is computed ONCE per sparsity pattern and cached.
//
// Device layout decided here (see DESIGN.md "Data layout in HBM"):
//   * rows are renumbered to the ILU elimination order: sorted by (level, caller index), so the
//     rows of one level are contiguous and a level is one coalesced kernel launch;
//   * matrix blocks live in SELL-64: rows are grouped in slices of 64 (= one wavefront), each slice
//     stores `width` slots; slot k of the 64 rows of a slice is stored as 9 component planes of 64
//     scalars, so lane l of a wave reads val[((slice_base + k) * 9 + comp) * 64 + l]: every load of
//     a wave is one contiguous 256/512-byte segment;
//   * inside a row the slots are sorted by internal column: [lower... | diag | upper... | padding].
#ifndef OPMGPU_PLAN_HPP
#define OPMGPU_PLAN_HPP

#include <cstdint>
#include <vector>

namespace opmgpu {

struct Plan {
    int nb = 0;        // block rows
    int nbp = 0;       // rows padded to a multiple of 64 (= vector plane stride)
    int nnzb = 0;      // blocks of the caller's pattern
    std::vector<int32_t> rowptr, col;      // caller's BSR pattern (columns ascending per row)

    std::vector<int32_t> pos;              // pos[caller row]    = internal row (elimination position)
    std::vector<int32_t> nat;              // nat[internal row]  = caller row
    std::vector<int32_t> level;            // level[internal row]
    int nlevels = 0;
    std::vector<int32_t> level_ptr;        // [nlevels+1] internal row ranges

    int nslices = 0;
    int nentries = 0;                      // slice_ptr[nslices] * 64
    std::vector<int32_t> slice_ptr;        // [nslices+1] cumulative slot widths
    std::vector<int32_t> sell_col;         // [nentries] internal column; padding -> own row (value 0)
    std::vector<int32_t> sell_src;         // [nentries] index of the caller's block, -1 for padding
    std::vector<int32_t> entry_of_block;   // [nnzb] entry id of each caller block
    std::vector<int16_t> rowlen;           // [nbp] real slots of the row
    std::vector<int16_t> nlower;           // [nbp] slots with internal col < row  (diag slot == nlower)

    // ILU0 updates of row i, grouped per row and ordered like dune's bilu0 (by j, then k):
    //   target(i,k) -= L(i,j) * U(j,k)
    std::vector<int32_t> trip_ptr;         // [nb+1]
    std::vector<int32_t> trip_l, trip_u, trip_t;   // entry ids
    std::vector<int8_t> simple;            // [nbp] 1 = every ILU0 update of the row hits its diagonal block: its U entries are those of A
    std::vector<int32_t> tpos;             // [nentries] entry id of the transposed block (j,i) of entry (i,j); -1: none / padding
    std::vector<int32_t> flux_perm;        // [ceil(nb / 256)] launch order of the assembly kernel's 256-row chunks (see build_plan)

    static inline int64_t val_index(int32_t entry, int comp) { return int64_t(entry >> 6) * 576 + comp * 64 + (entry & 63); }
    int32_t entry(int row, int slot) const { return (slice_ptr[row >> 6] + slot) * 64 + (row & 63); }
};

// ordering: OPMGPU_ORDER_NATURAL / OPMGPU_ORDER_MULTICOLOR.  Returns 0, or OPMGPU_EINVAL for a
// pattern without full diagonal / unsorted or duplicate columns / rows longer than 32767 blocks.
int build_plan(int nb, const int32_t* rowptr, const int32_t* col, int ordering, Plan& plan);

// symbolic block ILU(n) of P's caller pattern (level-of-fill, dune-istl's generation rule; csrc/fillilu.inl): the filled pattern in the
// caller's numbering, src2[b2] = block of the caller's pattern or -1 for a fill entry
void build_fill_pattern(const Plan& P, int n, std::vector<int32_t>& rowptr2, std::vector<int32_t>& col2, std::vector<int32_t>& src2);

// structure of the reservoir Jacobian: {c} U face/NNC neighbours U well cliques (formInterleavedSystem
// + the Schur fill of NewtonIterationUtilities.cpp:98-115).  conn_of_block[b] = (conn << 1 | side)
// for the block in row c coupling to the other cell of connection `conn` (side 1 = c is c2),
// -1 for diagonal, -2 for pure well fill.  Returns 0 or OPMGPU_EINVAL (duplicate cell pair).
int build_reservoir_pattern(int nc, int nconn, const int32_t* conn_cells, int nw, const int32_t* well_connpos,
                            const int32_t* well_cells, std::vector<int32_t>& rowptr, std::vector<int32_t>& col,
                            std::vector<int32_t>& conn_of_block);

} // namespace opmgpu
#endif
