"""Domain decomposition for multi-GPU runs: owned cells + one ghost layer per rank.

Role of the reference's `Dune::CpGrid::loadBalance` + `ParallelISTLInformation` index sets
(RedistributeDataHandles.hpp:540-628, ISTLSolver.hpp:286-298).  Rank-local numbering is
[owned (ascending global id) | ghosts (by owner rank, then global id)]; the send list towards a
neighbour and that neighbour's receive list name the same cells in the same (global id) order, so a
halo exchange is a plain pack / send / recv / unpack.
"""
import ctypes as C
import os

import numpy as np

from . import capi
from .decks import GridData, State


def slab_partition(grid, nranks, axis=2):
    """Contiguous slabs of whole layers along `axis` (default: the slowest, k) when the Cartesian dims are known, else
    contiguous index ranges.  axis = 1 (slabs of whole j-rows) keeps vertical wells on one rank -- the reference hands the
    wells to loadBalance for the same reason (RedistributeDataHandles.hpp:559-560).  Returns part[cell] = owner rank."""
    n = grid.nc
    if grid.dims is not None and grid.dims[0] * grid.dims[1] * grid.dims[2] == n:
        nx, ny, nz = grid.dims
        i, j, k = np.arange(n) % nx, (np.arange(n) // nx) % ny, np.arange(n) // (nx * ny)
        coord, extent = ((i, nx), (j, ny), (k, nz))[axis]
        return ((coord.astype(np.int64) * nranks) // extent).astype(np.int32)
    act = getattr(grid, "active_index", None)
    if grid.dims is not None and act is not None and axis in (0, 1) and len(act) == grid.dims[0] * grid.dims[1] * grid.dims[2]:
        # inactive cells (ACTNUM): slabs of whole i- or j-rows of the Cartesian BOX, the cut positions chosen so that every rank gets about the
        # same number of ACTIVE cells -- whole columns stay together, so vertical wells stay on one rank here too
        nx, ny, nz = grid.dims
        box = np.flatnonzero(np.asarray(act) >= 0)                     # Cartesian index of every active cell, in active order
        coord, extent = ((box % nx, nx), ((box // nx) % ny, ny))[axis]
        per_row = np.bincount(coord, minlength=extent)
        cum = np.cumsum(per_row)
        owner_of_row = np.minimum(nranks - 1, (np.maximum(cum - 1, 0).astype(np.int64) * nranks) // max(1, n)).astype(np.int32)
        return owner_of_row[coord]
    return ((np.arange(n, dtype=np.int64) * nranks) // n).astype(np.int32)


class LocalDomain:
    """One rank's share of a partitioned grid."""

    def __init__(self, grid, part, rank):
        part = np.asarray(part)
        conn = grid.conn_cells
        p1, p2 = part[conn[:, 0]], part[conn[:, 1]]
        keep = (p1 == rank) | (p2 == rank)
        lc = conn[keep]
        owned = np.flatnonzero(part == rank)
        cells = np.unique(lc)
        ghosts = cells[part[cells] != rank]
        ghosts = ghosts[np.lexsort((ghosts, part[ghosts]))]           # by owner, then global id
        self.rank, self.n_owned, self.n_ghost = rank, owned.size, ghosts.size
        self.global_of_local = np.concatenate([owned, ghosts]).astype(np.int64)
        g2l = -np.ones(grid.nc, dtype=np.int64)
        g2l[self.global_of_local] = np.arange(self.global_of_local.size)
        self.conn_index = np.flatnonzero(keep)
        self.grid = GridData(self.global_of_local.size, g2l[lc], grid.trans[keep], grid.pv[self.global_of_local],
                             grid.z[self.global_of_local], gravity=grid.gravity,
                             thpres=None if grid.thpres is None else grid.thpres[keep],
                             pvtnum=None if grid.pvtnum is None else grid.pvtnum[self.global_of_local],
                             satnum=None if grid.satnum is None else grid.satnum[self.global_of_local],
                             eps=None if grid.eps is None else {k: grid.eps[i][self.global_of_local] for i, k in enumerate(GridData.EPS_NAMES)})
        # halo lists
        gown = part[ghosts]
        self.neigh_rank = np.unique(gown).astype(np.int32)
        recv_ptr, send_ptr, send_cells = [0], [0], []
        cut = p1 != p2
        mine_a = cut & (p1 == rank)
        mine_b = cut & (p2 == rank)
        for q in self.neigh_rank:
            recv_ptr.append(recv_ptr[-1] + int((gown == q).sum()))
            a = conn[mine_a & (p2 == q), 0]
            b = conn[mine_b & (p1 == q), 1]
            s = np.unique(np.concatenate([a, b]))
            send_cells.append(g2l[s])
            send_ptr.append(send_ptr[-1] + s.size)
        self.recv_ptr, self.send_ptr = capi.i32(recv_ptr), capi.i32(send_ptr)
        self.recv_cells = capi.i32(self.n_owned + np.arange(ghosts.size))          # ghosts are already grouped by owner
        self.send_cells = capi.i32(np.concatenate(send_cells) if send_cells else np.zeros(0, np.int64))

    def local_wells(self, wells, part):
        """The wells of this rank in rank-local cell numbering: a well belongs to the rank that owns ALL of its perforated
        cells (ValueError if a well straddles ranks: choose a partition that keeps wells intact)."""
        from .wells import Wells
        g2l = -np.ones(part.size, dtype=np.int64)
        g2l[self.global_of_local] = np.arange(self.global_of_local.size)
        out = Wells()
        self.well_index = []
        for w in range(wells.nw):
            cells = np.asarray(wells.cells[wells.connpos[w]:wells.connpos[w + 1]])
            owners = np.unique(part[cells])
            if owners.size != 1:
                raise ValueError("well %s straddles ranks %s" % (wells.name[w], owners.tolist()))
            if owners[0] != self.rank:
                continue
            ctrls = wells.controls[w]
            out.add_well(wells.name[w], wells.type[w], wells.depth_ref[w], g2l[cells], wells.WI[wells.connpos[w]:wells.connpos[w + 1]],
                         wells.comp_frac[w], ctrls[0], allow_cf=wells.allow_cf[w], limits=ctrls[1:], current=wells.current0[w])
            self.well_index.append(w)
        return out

    def local_state(self, st):
        g = self.global_of_local
        return State(st.p[g], st.sat[g], st.rs[g], st.rv[g], st.hc[g])


SHM_TEST_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "support", "_build", "libshmtransport.so")


def shm_test_transport_wanted():
    """OPMGPU_COMM_TRANSPORT=shm: the multi-rank TESTS / the one-GPU rehearsal of bench.py couple their ranks (processes of one host,
    usually sharing one GPU) through the shared-memory test transport of tests/support instead of RCCL.  The product library knows
    nothing about it: it comes in through the public transport hook (opmgpu_comm_init_transport)."""
    return os.environ.get("OPMGPU_COMM_TRANSPORT") == "shm"


def attach_comm(model, dom, rank, world, unique_id):
    """opmgpu_comm_init (RCCL) -- or opmgpu_comm_init_transport with the shared-memory test transport -- for a model created on dom.grid."""
    lib = capi.load()
    lists = (dom.n_owned, int(dom.neigh_rank.size), capi.iptr(dom.neigh_rank), capi.iptr(dom.send_ptr), capi.iptr(dom.send_cells),
             capi.iptr(dom.recv_ptr), capi.iptr(dom.recv_cells))
    if shm_test_transport_wanted():
        if not os.path.exists(SHM_TEST_LIB):
            raise RuntimeError("test transport not built: make -C tests/support")
        tl = C.CDLL(SHM_TEST_LIB)
        tl.shm_transport_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(capi.Transport)]
        tr = capi.Transport()
        name = b"/opmgpu_" + bytes(unique_id)[:8].hex().encode()
        if tl.shm_transport_create(name, rank, world, C.byref(tr)) != 0:
            raise RuntimeError("shm test transport: could not open the segment")
        model._keep_transport = (tl, tr)
        st = lib.opmgpu_comm_init_transport(model.ctx, rank, world, C.byref(tr), *lists)
    else:
        idb = (C.c_uint8 * capi.UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
        st = lib.opmgpu_comm_init(model.ctx, rank, world, idb, *lists)
    if st != capi.OK:
        raise RuntimeError("opmgpu_comm_init failed with status %d: %s" % (st, lib.opmgpu_last_error(model.ctx)))


def make_unique_id():
    if shm_test_transport_wanted():           # a random segment name instead of an RCCL id
        return os.urandom(capi.UNIQUE_ID_BYTES)
    lib = capi.load()
    buf = (C.c_uint8 * capi.UNIQUE_ID_BYTES)()
    st = lib.opmgpu_comm_unique_id(buf)
    if st != capi.OK:
        raise RuntimeError("opmgpu_comm_unique_id failed with status %d" % st)
    return bytes(buf)


def build_distributed_model(nx, ny, nz, tables, params, rank, world, local_rank, lognormal_sigma=0.5, seed=12345, perturb=0.002, deck="cart", wells_fn=None, axis=2):
    """Every rank builds the same global synthetic deck, keeps its slab (+ghosts) and joins the RCCL communicator.
    The unique id travels through torch.distributed (backend nccl = RCCL).  deck = "spe10like": BASELINE configs[3], the 60 x 220 x 85
    deck with sigma_lnK = 2.5 cut along j (27-28 rows of 60 x 85 cells per GPU at N = 8: strong scaling by construction).  axis: the
    direction the Cartesian deck is cut along (2 = slabs of k-layers; 1 = slabs of whole j-rows, which keeps vertical wells on one rank --
    the strong-scaling leg and the SPE10-like deck carry their wells that way)."""
    import torch
    import torch.distributed as dist
    from . import decks
    from .model import GpuBlackoilModel
    if isinstance(deck, tuple):
        # a deck the caller built (grid, state): every rank passes the same one; cut into slabs of whole j-rows (vertical wells stay whole)
        grid, st = deck
        part = slab_partition(grid, world, axis=axis if axis in (0, 1) else 1)
    elif deck == "spe10like":
        grid = decks.cartesian_grid(60, 220, 85, dx=6.096, dy=3.048, dz=0.6096, tops=3657.6, lognormal_sigma=2.5, seed=10)
        st = decks.initial_state(grid, tables, p_ref=413.0 * decks.BAR, z_ref=3657.6, perturb=1e-4, seed=10, gas_cap_fraction=0.0, gas_only_fraction=0.0)
        part = slab_partition(grid, world, axis=axis if axis in (0, 1) else 1)      # vertical wells: along i or j only
    else:
        grid = decks.cartesian_grid(nx, ny, nz, lognormal_sigma=lognormal_sigma, seed=seed)
        st = decks.initial_state(grid, tables, perturb=perturb, seed=seed)
        part = slab_partition(grid, world, axis=axis)
    dom = LocalDomain(grid, part, rank)
    model = GpuBlackoilModel(dom.grid, tables, params, device=local_rank)
    dev = torch.device("cuda", local_rank) if dist.get_backend() == "nccl" else torch.device("cpu")      # gloo: one-GPU rehearsal
    idt = torch.zeros(capi.UNIQUE_ID_BYTES, dtype=torch.uint8, device=dev)
    if rank == 0:
        idt.copy_(torch.frombuffer(bytearray(make_unique_id()), dtype=torch.uint8))
    dist.broadcast(idt, src=0)
    attach_comm(model, dom, rank, world, bytes(idt.cpu().numpy().tobytes()))
    # coarse blocks of the pressure stage: sub-slabs of this rank's slab along the CUT direction (they keep vertical wells whole)
    mblk = int(os.environ.get("OPMGPU_COARSE_SUBSLABS", "0"))
    cut_axis = (axis if axis in (0, 1) else 1) if (deck == "spe10like" or isinstance(deck, tuple)) else axis
    if mblk > 1 and grid.dims is not None and cut_axis in (0, 1):
        nxg, nyg, _ = grid.dims
        gid = dom.global_of_local[:dom.n_owned]
        coord = (gid % nxg) if cut_axis == 0 else ((gid // nxg) % nyg)
        lo, hi = int(coord.min()), int(coord.max()) + 1
        blk = capi.i32(np.minimum(mblk - 1, (coord - lo) * mblk // max(1, hi - lo)))
        st_ = capi.load().opmgpu_comm_set_coarse_blocks(model.ctx, mblk, capi.iptr(blk))
        if st_ != capi.OK:
            raise RuntimeError("opmgpu_comm_set_coarse_blocks failed with status %d" % st_)
    info = {"n_owned": dom.n_owned, "n_global": grid.nc, "n_ghost": dom.n_ghost, "neighbours": dom.neigh_rank.tolist()}
    if wells_fn is not None:          # wells of the GLOBAL deck (every well inside one rank's cells) -> this rank's, in local numbering
        info["wells"] = dom.local_wells(wells_fn(grid), part)
    return model, dom.grid, dom.local_state(st), info
