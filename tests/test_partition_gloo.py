"""N > 1 path on the CPU: world_size-2 gloo processes check the domain decomposition the multi-GPU
solver is built on -- ownership, matching send/recv lists, that rank-local assembly of owned rows
reproduces the global Jacobian/residual, and that halo exchange + local SpMV reproduces the global
SpMV (the owner/overlap scheme of ISTLSolver.hpp:286-298)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kind, out_q):
    try:
        for p in (os.path.join(ROOT, "opm-simulators-legacy_amd"), ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from opmgpu import capi, decks, partition
        from oracle import oracle as orc
        from util import bsr_to_scipy

        if kind == "cart":
            grid = decks.cartesian_grid(6, 5, 8, lognormal_sigma=0.6)
            part = partition.slab_partition(grid, world)
        elif kind == "actnum_rows":   # holes + NNCs with the box known: slabs of whole j-rows balanced by active cells (bench.py --deck nornelike / random, N > 1)
            act = np.random.default_rng(4).random(8 * 9 * 5) > 0.4
            grid = decks.cartesian_grid(8, 9, 5, actnum=act, nnc_fraction=0.05)
            part = partition.slab_partition(grid, world, axis=1)
            assert np.unique(part).size == world
        else:   # unstructured: holes + NNCs, index-range partition
            act = np.random.default_rng(2).random(7 * 6 * 6) > 0.25
            grid = decks.cartesian_grid(7, 6, 6, actnum=act, nnc_fraction=0.05)
            grid.dims = None
            part = partition.slab_partition(grid, world)
        tab = decks.satfunc_standard_tables()
        st = decks.random_state(grid, tab, seed=5)
        dom = partition.LocalDomain(grid, part, rank)

        # 1. ownership: every cell owned exactly once, ghosts are owned elsewhere
        owned = dom.global_of_local[:dom.n_owned]
        gathered = [None] * world
        dist.all_gather_object(gathered, owned.tolist())
        allc = np.sort(np.concatenate([np.asarray(g, dtype=np.int64) for g in gathered]))
        assert np.array_equal(allc, np.arange(grid.nc))
        assert np.all(part[dom.global_of_local[dom.n_owned:]] != rank)

        # 2. send list towards q == q's receive list from me (same cells, same order)
        lists = {"send": {int(q): dom.global_of_local[dom.send_cells[dom.send_ptr[i]:dom.send_ptr[i + 1]]].tolist() for i, q in enumerate(dom.neigh_rank)},
                 "recv": {int(q): dom.global_of_local[dom.recv_cells[dom.recv_ptr[i]:dom.recv_ptr[i + 1]]].tolist() for i, q in enumerate(dom.neigh_rank)}}
        alll = [None] * world
        dist.all_gather_object(alll, lists)
        for q in range(world):
            if q == rank:
                continue
            assert lists["send"].get(q, []) == alll[q]["recv"].get(rank, [])
            assert lists["recv"].get(q, []) == alll[q]["send"].get(rank, [])

        # 3. rank-local assembly of the owned rows == global assembly (residual and every block)
        prm = capi.default_params()
        scale = tuple(prm.matbalscale)
        dt = 4 * decks.DAY
        rp, cl = orc.pattern(grid)
        rg, vg, acc0, _ = orc.assemble(grid, tab, dt, st, rp, cl, scale=scale)
        stl = dom.local_state(st)
        rpl, cll = orc.pattern(dom.grid)
        rl, vl, _, _ = orc.assemble(dom.grid, tab, dt, stl, rpl, cll, scale=scale)
        ncl, g = dom.grid.nc, dom.global_of_local
        for a in range(3):
            assert np.allclose(rl[a * ncl:a * ncl + dom.n_owned], rg[a * grid.nc + g[:dom.n_owned]], rtol=1e-12, atol=1e-9 * np.abs(rg).max())
        Ag = bsr_to_scipy(rp, cl, vg).tolil() if grid.nc < 400 else None
        Al = bsr_to_scipy(rpl, cll, vl).tocsr()
        Agc = bsr_to_scipy(rp, cl, vg).tocsr()
        for i in range(dom.n_owned):
            for k in range(rpl[i], rpl[i + 1]):
                j = cll[k]
                blk_l = Al[3 * i:3 * i + 3, 3 * j:3 * j + 3].toarray()
                blk_g = Agc[3 * g[i]:3 * g[i] + 3, 3 * g[j]:3 * g[j] + 3].toarray()
                assert np.allclose(blk_l, blk_g, rtol=1e-11, atol=1e-12 * np.abs(vg).max())
            assert rpl[i + 1] - rpl[i] == rp[g[i] + 1] - rp[g[i]]          # owned rows are complete

        # 4. halo exchange + local SpMV on owned rows == global SpMV
        x_glob = np.random.default_rng(0).standard_normal(3 * grid.nc)
        x_loc = np.zeros((ncl, 3))
        x_loc[:dom.n_owned] = x_glob.reshape(-1, 3)[g[:dom.n_owned]]
        reqs, rbufs = [], []
        for i, q in enumerate(dom.neigh_rank):
            sb = torch.from_numpy(np.ascontiguousarray(x_loc[dom.send_cells[dom.send_ptr[i]:dom.send_ptr[i + 1]]]))
            rb = torch.zeros((int(dom.recv_ptr[i + 1] - dom.recv_ptr[i]), 3), dtype=torch.float64)
            reqs.append(dist.isend(sb, int(q))); reqs.append(dist.irecv(rb, int(q))); rbufs.append((i, rb))
        for r in reqs:
            r.wait()
        for i, rb in rbufs:
            x_loc[dom.recv_cells[dom.recv_ptr[i]:dom.recv_ptr[i + 1]]] = rb.numpy()
        assert np.array_equal(x_loc, x_glob.reshape(-1, 3)[g])               # ghosts now hold the owners' values
        y_loc = (Al @ x_loc.ravel()).reshape(-1, 3)[:dom.n_owned]
        y_glob = (Agc @ x_glob).reshape(-1, 3)[g[:dom.n_owned]]
        assert np.allclose(y_loc, y_glob, rtol=1e-12, atol=1e-12 * np.abs(y_glob).max())

        # 5. owner-masked dot + all-reduce == global dot
        loc = torch.tensor([float(np.dot(x_loc[:dom.n_owned].ravel(), x_loc[:dom.n_owned].ravel()))], dtype=torch.float64)
        dist.all_reduce(loc)
        assert abs(loc.item() - float(x_glob @ x_glob)) < 1e-9 * float(x_glob @ x_glob)
        dist.barrier()
        dist.destroy_process_group()
        out_q.put((rank, "ok"))
    except Exception as e:      # noqa: BLE001
        import traceback
        out_q.put((rank, "FAIL: " + traceback.format_exc()))


@pytest.mark.parametrize("kind", ["cart", "unstructured", "actnum_rows"])
def test_partition_world2_gloo(kind):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r, msg in res:
        assert msg == "ok", "rank %d: %s" % (r, msg)


def test_slab_partition_properties():
    sys.path.insert(0, os.path.join(ROOT, "opm-simulators-legacy_amd"))
    from opmgpu import decks, partition
    grid = decks.cartesian_grid(5, 4, 16)
    for n in (1, 2, 3, 4, 8):
        part = partition.slab_partition(grid, n)
        assert part.min() == 0 and part.max() == n - 1
        counts = np.bincount(part)
        assert counts.max() - counts.min() <= 5 * 4           # at most one layer of imbalance
        doms = [partition.LocalDomain(grid, part, r) for r in range(n)]
        assert sum(d.n_owned for d in doms) == grid.nc
        for d in doms:
            assert len(d.neigh_rank) <= 2                     # slabs: at most two neighbours
            assert d.send_ptr[-1] == d.send_cells.size and d.recv_ptr[-1] == d.n_ghost
