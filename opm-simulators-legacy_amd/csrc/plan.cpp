// plan.cpp -- see plan.hpp.  Pure host code (no HIP), unit-tested on the CPU.
#include "plan.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <numeric>

#include "../../include/opmgpu.h"

namespace opmgpu {

constexpr int kDefaultZBlock = 1;      // plane-interleaving block of the intra-level row order (see build_plan)

int build_reservoir_pattern(int nc, int nconn, const int32_t* conn_cells, int nw, const int32_t* well_connpos,
                            const int32_t* well_cells, std::vector<int32_t>& rowptr, std::vector<int32_t>& col,
                            std::vector<int32_t>& conn_of_block)
{
    // (col, code) pairs per row; code as documented in plan.hpp
    std::vector<std::vector<std::pair<int32_t, int32_t> > > adj(nc);
    for (int c = 0; c < nc; ++c) adj[c].push_back(std::make_pair(c, -1));
    for (int f = 0; f < nconn; ++f) {
        const int a = conn_cells[2 * f], b = conn_cells[2 * f + 1];
        if (a < 0 || b < 0 || a >= nc || b >= nc || a == b) return OPMGPU_EINVAL;
        adj[a].push_back(std::make_pair(b, (f << 1) | 0));
        adj[b].push_back(std::make_pair(a, (f << 1) | 1));
    }
    for (int w = 0; w < nw; ++w)
        for (int i = well_connpos[w]; i < well_connpos[w + 1]; ++i)
            for (int j = well_connpos[w]; j < well_connpos[w + 1]; ++j) {
                const int a = well_cells[i], b = well_cells[j];
                if (a < 0 || b < 0 || a >= nc || b >= nc) return OPMGPU_EINVAL;
                if (a != b) adj[a].push_back(std::make_pair(b, -2));
            }
    rowptr.assign(nc + 1, 0); col.clear(); conn_of_block.clear();
    for (int c = 0; c < nc; ++c) {
        auto& v = adj[c];
        // sort by column; a real connection (code >= 0) wins over well fill (-2) for the same column
        std::sort(v.begin(), v.end(), [](const std::pair<int32_t, int32_t>& x, const std::pair<int32_t, int32_t>& y) {
            return x.first != y.first ? x.first < y.first : x.second > y.second; });
        int32_t last = -1;
        for (size_t k = 0; k < v.size(); ++k) {
            if (v[k].first == last) {
                if (v[k].second >= 0) return OPMGPU_EINVAL;        // two connections between one cell pair
                continue;
            }
            last = v[k].first;
            col.push_back(v[k].first); conn_of_block.push_back(v[k].second);
        }
        rowptr[c + 1] = int32_t(col.size());
    }
    return OPMGPU_OK;
}

// Is the pattern the 7-point stencil of a full nx x ny x nz Cartesian box in natural (i fastest) order?  (all offsets in
// {+-1, +-nx, +-nx*ny} with exactly the face counts of the box)
static bool infer_cartesian_dims(int nb, const int32_t* rowptr, const int32_t* col, int& nx, int& ny, int& nz)
{
    std::map<int32_t, int64_t> cnt;
    for (int i = 0; i < nb; ++i)
        for (int s = rowptr[i]; s < rowptr[i + 1]; ++s) { const int d = col[s] - i; if (d > 0) { if (cnt.size() > 3 && !cnt.count(d)) return false; cnt[d]++; } }
    if (cnt.size() != 3 || cnt.begin()->first != 1) return false;
    auto it = cnt.begin(); ++it; const int a = it->first; ++it; const int b = it->first;
    if (a <= 1 || b <= a || b % a != 0 || nb % b != 0) return false;
    nx = a; ny = b / a; nz = nb / b;
    return cnt[1] == int64_t(nx - 1) * ny * nz && cnt[a] == int64_t(nx) * (ny - 1) * nz && cnt[b] == int64_t(nx) * ny * (nz - 1);
}

int build_plan(int nb, const int32_t* rowptr, const int32_t* col, int ordering, Plan& P)
{
    if (nb <= 0 || !rowptr || !col) return OPMGPU_EINVAL;
    P = Plan();
    P.nb = nb; P.nbp = (nb + 63) / 64 * 64; P.nnzb = rowptr[nb];
    P.rowptr.assign(rowptr, rowptr + nb + 1); P.col.assign(col, col + P.nnzb);
    // validate: ascending unique columns, diagonal present
    for (int i = 0; i < nb; ++i) {
        bool diag = false;
        if (rowptr[i + 1] - rowptr[i] > 32767) return OPMGPU_EINVAL;
        for (int s = rowptr[i]; s < rowptr[i + 1]; ++s) {
            if (col[s] < 0 || col[s] >= nb) return OPMGPU_EINVAL;
            if (s > rowptr[i] && col[s] <= col[s - 1]) return OPMGPU_EINVAL;
            diag = diag || col[s] == i;
        }
        if (!diag) return OPMGPU_EINVAL;
    }
    // --- structurally non-symmetric patterns (the ILU(n) fill of fillilu.inl from n = 2 on): the two sweeps of the ILU need every coupled
    // pair of rows in DIFFERENT levels, ordered like the elimination order, whichever of (i,j), (j,i) exists -- the dependency graph is the
    // undirected one.  xadj lists, per row, the neighbours that appear only in the OTHER row's pattern (empty for the Jacobian's pattern).
    std::vector<int32_t> xptr(size_t(nb) + 1, 0), xadj;
    {
        std::vector<std::pair<int32_t, int32_t> > miss;          // (j, i): entry (i,j) without (j,i)
        for (int i = 0; i < nb; ++i)
            for (int s = rowptr[i]; s < rowptr[i + 1]; ++s) {
                const int j = col[s];
                if (j != i && !std::binary_search(col + rowptr[j], col + rowptr[j + 1], i)) miss.push_back(std::make_pair(j, i));
            }
        if (!miss.empty()) {
            std::sort(miss.begin(), miss.end());
            xadj.resize(miss.size());
            for (size_t q = 0; q < miss.size(); ++q) { xptr[size_t(miss[q].first) + 1]++; xadj[q] = miss[q].second; }
            for (int i = 0; i < nb; ++i) xptr[size_t(i) + 1] += xptr[i];
        }
    }
    // --- orientation key: NATURAL = caller index; MULTICOLOR = (greedy first-fit colour, caller index)
    std::vector<int32_t> colour(nb, 0);
    if (ordering == OPMGPU_ORDER_MULTICOLOR) {
        std::vector<int32_t> mark(nb + 1, -1);          // mark[c] == i  <=> colour c used by a neighbour of i
        for (int i = 0; i < nb; ++i) {
            for (int s = rowptr[i]; s < rowptr[i + 1]; ++s) { const int j = col[s]; if (j < i) mark[colour[j]] = i; }
            for (int s = xptr[i]; s < xptr[i + 1]; ++s) { const int j = xadj[s]; if (j < i) mark[colour[j]] = i; }
            int c = 0; while (mark[c] == i) ++c;
            colour[i] = c;
        }
    } else if (ordering != OPMGPU_ORDER_NATURAL) return OPMGPU_EINVAL;
    auto before = [&](int a, int b) { return colour[a] != colour[b] ? colour[a] < colour[b] : a < b; };
    // --- levels = longest path in the DAG oriented by `before`; visit rows in a topological order
    std::vector<int32_t> topo(nb); std::iota(topo.begin(), topo.end(), 0);
    if (ordering == OPMGPU_ORDER_MULTICOLOR) std::stable_sort(topo.begin(), topo.end(), [&](int a, int b) { return colour[a] < colour[b]; });
    std::vector<int32_t> lev(nb, 0);
    int nlev = 0;
    for (int t = 0; t < nb; ++t) {
        const int i = topo[t]; int l = 0;
        for (int s = rowptr[i]; s < rowptr[i + 1]; ++s) { const int j = col[s]; if (j != i && before(j, i)) l = std::max(l, lev[j] + 1); }
        for (int s = xptr[i]; s < xptr[i + 1]; ++s) { const int j = xadj[s]; if (before(j, i)) l = std::max(l, lev[j] + 1); }
        lev[i] = l; nlev = std::max(nlev, l + 1);
    }
    // --- internal numbering: sort by (level, caller index).  (A/B on MI355X, 100^3: re-grouping the rows of a level
    // into compact BFS clusters of 512 / 4096 rows made SpMV 5-12 % and the assembly 30-45 % SLOWER than keeping the
    // caller's order inside a level -- the banded Cartesian order already keeps gathers within a few cache lines.)
    // Rows of one level are mutually independent, so their order inside the level is free (speed only).  For a full Cartesian
    // box the k-planes are interleaved line by line in blocks of B planes -- key ((k / B), j, (k % B), i) -- so that both the y- and
    // (B - 1 of B) z-neighbours of a row sit a few lines away instead of a whole plane away: every per-wave gather stays a
    // contiguous run, but its reuse distance shrinks from nx*ny to B*nx rows and fits the L2.  OPMGPU_ZBLOCK overrides B (1 = off).
    P.nat.resize(nb); std::iota(P.nat.begin(), P.nat.end(), 0);
    std::vector<int64_t> lkey;
    {
        int B = kDefaultZBlock, nx = 0, ny = 0, nz = 0;
        if (const char* e = std::getenv("OPMGPU_ZBLOCK")) B = std::atoi(e);
        // OPMGPU_BRICK=bx,by,bz: rows of a level ordered brick by brick (bricks x-fastest, cells inside a brick x-fastest).  With the
        // level-interleaved chunk order of the assembly kernel (flux_perm below) the rows in flight on one XCD are then BOTH colours of a
        // compact set of bricks: a cell record is fetched once and found in the L2 by the six rows that need it, instead of once per
        // colour pass plus every time its k-plane neighbours come round.
        int bx = 0, by = 0, bz = 0;
        if (const char* e = std::getenv("OPMGPU_BRICK")) std::sscanf(e, "%d,%d,%d", &bx, &by, &bz);
        if (bx > 0 && by > 0 && bz > 0 && ordering == OPMGPU_ORDER_MULTICOLOR && infer_cartesian_dims(nb, rowptr, col, nx, ny, nz)) {
            lkey.resize(nb);
            const int64_t nbi = (nx + bx - 1) / bx, nbj = (ny + by - 1) / by;
            for (int c = 0; c < nb; ++c) {
                const int64_t i = c % nx, j = (c / nx) % ny, k = c / (int64_t(nx) * ny);
                const int64_t brick = ((k / bz) * nbj + (j / by)) * nbi + (i / bx);
                lkey[c] = ((brick * bz + (k % bz)) * by + (j % by)) * bx + (i % bx);
            }
        } else
        if (B > 1 && ordering == OPMGPU_ORDER_MULTICOLOR && infer_cartesian_dims(nb, rowptr, col, nx, ny, nz)) {
            lkey.resize(nb);
            for (int c = 0; c < nb; ++c) {
                const int64_t i = c % nx, j = (c / nx) % ny, k = c / (int64_t(nx) * ny);
                lkey[c] = (((k / B) * ny + j) * B + (k % B)) * nx + i;
            }
        }
    }
    if (lkey.empty()) std::stable_sort(P.nat.begin(), P.nat.end(), [&](int a, int b) { return lev[a] < lev[b]; });
    else std::sort(P.nat.begin(), P.nat.end(), [&](int a, int b) { return lev[a] != lev[b] ? lev[a] < lev[b] : lkey[a] < lkey[b]; });
    P.pos.resize(nb);
    for (int r = 0; r < nb; ++r) P.pos[P.nat[r]] = r;
    P.level.resize(nb); P.nlevels = nlev; P.level_ptr.assign(nlev + 1, 0);
    for (int r = 0; r < nb; ++r) { P.level[r] = lev[P.nat[r]]; P.level_ptr[P.level[r] + 1]++; }
    for (int l = 0; l < nlev; ++l) P.level_ptr[l + 1] += P.level_ptr[l];
    // --- launch order of the assembly's 256-row chunks: the chunks of all levels interleaved in proportion to their position inside
    // their level (chunk q of level l has key q / chunks(l)), so that rows of different colours that sit in the same place are in
    // flight together.  Identity unless OPMGPU_FLUX_INTERLEAVE=1 (needs an intra-level order that means the same place in every
    // level: the brick order above).
    {
        const int nch = (nb + 255) / 256;
        P.flux_perm.resize(nch); std::iota(P.flux_perm.begin(), P.flux_perm.end(), 0);
        const char* e = std::getenv("OPMGPU_FLUX_INTERLEAVE");
        if (e && std::atoi(e) != 0 && nlev > 1 && nlev <= 16) {
            std::vector<double> key(nch);
            for (int ch = 0; ch < nch; ++ch) {
                const int r = ch * 256, l = P.level[r];
                const double lo = P.level_ptr[l], len = std::max(1, P.level_ptr[l + 1] - P.level_ptr[l]);
                key[ch] = (r - lo) / len + 1e-9 * l;
            }
            std::stable_sort(P.flux_perm.begin(), P.flux_perm.end(), [&](int a, int b) { return key[a] < key[b]; });
        }
    }
    // --- SELL-64
    P.nslices = P.nbp / 64; P.slice_ptr.assign(P.nslices + 1, 0);
    P.rowlen.assign(P.nbp, 0); P.nlower.assign(P.nbp, 0);
    for (int s = 0; s < P.nslices; ++s) {
        int w = 0;
        for (int r = s * 64; r < std::min(nb, s * 64 + 64); ++r) { const int c = P.nat[r]; w = std::max(w, rowptr[c + 1] - rowptr[c]); }
        P.slice_ptr[s + 1] = P.slice_ptr[s] + w;
    }
    P.nentries = P.slice_ptr[P.nslices] * 64;
    P.sell_col.assign(P.nentries, 0); P.sell_src.assign(P.nentries, -1); P.entry_of_block.assign(P.nnzb, -1);
    std::vector<std::pair<int32_t, int32_t> > tmp;
    for (int r = 0; r < P.nbp; ++r) {
        const int base = P.slice_ptr[r >> 6], w = P.slice_ptr[(r >> 6) + 1] - base;
        int len = 0;
        if (r < nb) {
            const int c = P.nat[r];
            tmp.clear();
            for (int s = rowptr[c]; s < rowptr[c + 1]; ++s) tmp.push_back(std::make_pair(P.pos[col[s]], s));
            std::sort(tmp.begin(), tmp.end());
            len = int(tmp.size());
            int nl = 0;
            for (int k = 0; k < len; ++k) {
                const int e = (base + k) * 64 + (r & 63);
                P.sell_col[e] = tmp[k].first; P.sell_src[e] = tmp[k].second; P.entry_of_block[tmp[k].second] = e;
                if (tmp[k].first < r) ++nl;
            }
            P.rowlen[r] = int16_t(len); P.nlower[r] = int16_t(nl);
        }
        for (int k = len; k < w; ++k) P.sell_col[(base + k) * 64 + (r & 63)] = std::min(r, nb - 1);   // padding: value 0, safe gather
    }
    // --- transposed entries (column sums of formEllipticSystem's dominance test are taken over rows)
    P.tpos.assign(P.nentries, -1);
    for (int r = 0; r < nb; ++r)
        for (int k = 0; k < P.rowlen[r]; ++k) {
            const int e = P.entry(r, k), j = P.sell_col[e];
            int lo = 0, hi = P.rowlen[j];                   // columns of row j ascend over its real slots
            while (lo < hi) { const int mid = (lo + hi) / 2; if (P.sell_col[P.entry(j, mid)] < r) lo = mid + 1; else hi = mid; }
            if (lo < P.rowlen[j] && P.sell_col[P.entry(j, lo)] == r) P.tpos[e] = P.entry(j, lo);
        }
    // --- ILU0 update triples
    P.trip_ptr.assign(nb + 1, 0);
    std::vector<int32_t> slot_of(nb, -1);
    for (int i = 0; i < nb; ++i) {
        const int len = P.rowlen[i], nl = P.nlower[i];
        for (int k = 0; k < len; ++k) slot_of[P.sell_col[P.entry(i, k)]] = k;
        for (int a = 0; a < nl; ++a) {
            const int eij = P.entry(i, a); const int j = P.sell_col[eij];
            for (int b = P.nlower[j] + 1; b < P.rowlen[j]; ++b) {
                const int ejk = P.entry(j, b); const int k = P.sell_col[ejk];
                if (slot_of[k] < 0) continue;
                if (P.trip_l.size() >= size_t(0x7fff0000)) return OPMGPU_EINVAL;          // (int32 triple ids: a pattern this dense is not an ILU pattern)
                P.trip_l.push_back(eij); P.trip_u.push_back(ejk); P.trip_t.push_back(P.entry(i, slot_of[k]));
            }
        }
        for (int k = 0; k < len; ++k) slot_of[P.sell_col[P.entry(i, k)]] = -1;
        P.trip_ptr[i + 1] = int32_t(P.trip_l.size());
    }
    P.simple.assign(P.nbp, 0);
    for (int i = 0; i < nb; ++i) {
        const int32_t ed = P.entry(i, P.nlower[i]);
        bool sm = true;
        for (int q = P.trip_ptr[i]; q < P.trip_ptr[i + 1]; ++q) sm = sm && (P.trip_t[q] == ed);
        P.simple[i] = sm ? 1 : 0;
    }
    return OPMGPU_OK;
}

// symbolic ILU(n) of the caller's pattern, level-of-fill (csrc/fillilu.inl): rows in the caller's order; entries of A have level 0, the
// elimination of (i,k) by row k offers (i,j) the level lev(i,k) + lev(k,j) + 1 for every kept (k,j), j > k; an entry is kept at its
// smallest offer when that is <= n.  src2[b2] = block of the caller's pattern, -1 for fill
void build_fill_pattern(const Plan& P, int n, std::vector<int32_t>& rowptr2, std::vector<int32_t>& col2, std::vector<int32_t>& src2)
{
    const int nb = P.nb;
    std::vector<std::vector<std::pair<int32_t, int8_t>>> rows(nb);       // (column, level), ascending columns
    std::vector<int32_t> diag_at(nb, 0);
    std::map<int32_t, int> pat;
    rowptr2.assign(size_t(nb) + 1, 0);
    for (int i = 0; i < nb; ++i) {
        pat.clear();
        for (int s = P.rowptr[i]; s < P.rowptr[i + 1]; ++s) pat[P.col[s]] = 0;
        for (auto ik = pat.begin(); ik != pat.end() && ik->first < i; ++ik) {          // (entries inserted below sit behind ik: columns > k)
            const auto& rk = rows[ik->first];
            for (size_t q = size_t(diag_at[ik->first]) + 1; q < rk.size(); ++q) {
                const int lev = ik->second + rk[q].second + 1;
                if (lev > n) continue;
                auto it = pat.find(rk[q].first);
                if (it == pat.end()) pat[rk[q].first] = lev; else if (lev < it->second) it->second = lev;
            }
        }
        auto& ri = rows[i];
        ri.reserve(pat.size());
        for (const auto& e : pat) { if (e.first == i) diag_at[i] = int32_t(ri.size()); ri.emplace_back(e.first, int8_t(e.second)); }
        rowptr2[size_t(i) + 1] = rowptr2[i] + int32_t(ri.size());
    }
    col2.resize(size_t(rowptr2[nb])); src2.assign(size_t(rowptr2[nb]), -1);
    for (int i = 0; i < nb; ++i) {
        int s = P.rowptr[i];
        int32_t o = rowptr2[i];
        for (const auto& e : rows[i]) {
            col2[o] = e.first;
            while (s < P.rowptr[i + 1] && P.col[s] < e.first) ++s;
            if (s < P.rowptr[i + 1] && P.col[s] == e.first) src2[o] = s;
            ++o;
        }
    }
}

} // namespace opmgpu
