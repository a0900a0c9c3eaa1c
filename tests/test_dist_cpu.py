"""CPU tests of the decomposition layer (die_amd/dist.py) over gloo, world sizes 2 and 4: tile
geometry, two-phase periodic halo exchange incl. corners, agent-record routing, hole filling.
No kernel runs here; the GPU equality test is tests/test_gpu_dist.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from die_amd.dist import Comm, TileGeometry, fill_holes, halo_exchange, halo_merge_max, peer_message_layout, route_records


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, size, port, fn, args):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    try:
        fn(rank, size, *args)
    finally:
        dist.destroy_process_group()


def _run(size, fn, *args):
    mp.spawn(_worker, args=(size, _free_port(), fn, args), nprocs=size, join=True)


def test_tile_geometry():
    g = TileGeometry((64, 48), (2, 2), 3, 6)
    assert (g.px, g.py, g.Wi, g.Hi, g.x0, g.y0) == (1, 1, 32, 24, 32, 24)
    assert (g.W, g.H, g.ox, g.oy) == (44, 36, 26, 18)
    assert g.neighbour(1, 1) == 0 and g.neighbour(-1, 0) == 1 and g.neighbour(0, -1) == 2
    assert list(g.tile_of_cell([0, 31, 32, 63], [0, 23, 24, 47])) == [0, 0, 3, 3]
    with pytest.raises(ValueError):
        TileGeometry((64, 48), (3, 2), 0, 2)
    with pytest.raises(ValueError):
        TileGeometry((64, 48), (2, 2), 0, 40)
    g1 = TileGeometry((16, 16), (1, 1), 0, 4)
    assert g1.neighbour(1, 0) == 0 and g1.neighbour(0, -1) == 0
    # per-axis halos (ghost-agent mode): one rank along x → no halo there, only the y sides exchange
    g2 = TileGeometry((64, 48), (1, 2), 1, (0, 8))
    assert (g2.W, g2.H, g2.ox, g2.oy, g2.hx, g2.hy) == (64, 24 + 16, 0, 24 - 8, 0, 8)
    assert g2.DIRS == [(0, -1), (0, 1)] and g2.interior() == (slice(0, 64), slice(8, 32))
    assert g2._band(0, -1) == (slice(0, 64), slice(8, 16)) and g2._halo(0, 1) == (slice(0, 64), slice(32, 40))
    sends, recvs = g2.plan8()
    assert [p for p, _ in sends] == [0, 0] and [v for _, v in recvs] == [g2._halo(0, 1), g2._halo(0, -1)]
    with pytest.raises(ValueError):
        TileGeometry((64, 48), (2, 2), 0, (0, 8))          # an axis without halo needs exactly one rank


def _halo_case(rank, size, world, grid, h):
    g = TileGeometry(world, grid, rank, h)
    comm = Comm()
    gid = np.arange(world[0] * world[1], dtype=np.float32).reshape(world)
    plane = torch.full((g.W, g.H), -1.0)
    ri, ci = g.interior()
    plane[ri, ci] = torch.from_numpy(gid[g.x0:g.x0 + g.Wi, g.y0:g.y0 + g.Hi])
    halo_exchange(plane, g, comm)
    ix = (np.arange(g.W) + g.ox) % world[0]
    iy = (np.arange(g.H) + g.oy) % world[1]
    want = gid[ix][:, iy]
    assert np.array_equal(plane.numpy(), want), f'rank {rank}: halo mismatch'


@pytest.mark.parametrize('size,world,grid,h', [(2, (12, 16), (1, 2), 3), (2, (16, 12), (2, 1), 4), (4, (16, 24), (2, 2), 5),
                                                (1, (8, 8), (1, 1), 3), (4, (32, 8), (4, 1), 2),
                                                (2, (12, 16), (1, 2), (0, 4)), (4, (32, 8), (4, 1), (3, 0))])
def test_halo_exchange_is_periodic_including_corners(size, world, grid, h):
    _run(size, _halo_case, world, grid, h)


def _peer_message_case(rank, size, world, grid, h):
    """ONE message per peer (die_amd/dist.py peer_message_layout, the ghost refresh by tiles): every side's block — here the
    band cells of a plane of world cell ids plus a (sender, side) tag — must arrive as the halo block of the opposite side."""
    g = TileGeometry(world, grid, rank, h)
    comm = Comm()
    gid = np.arange(world[0] * world[1], dtype=np.float32).reshape(world)
    plane = torch.full((g.W, g.H), -1.0)
    ri, ci = g.interior()
    plane[ri, ci] = torch.from_numpy(gid[g.x0:g.x0 + g.Wi, g.y0:g.y0 + g.Hi])
    nd = len(g.DIRS)
    shapes = [tuple(sl.stop - sl.start for sl in g._band(dx, dy)) for dx, dy in g.DIRS]
    sizes = [(2 + a * b) * 4 for a, b in shapes]                          # tag (sender, side) + the cells, float32
    peers, soff, roff, sspan, rspan = peer_message_layout(g.DIRS, g.neighbour, sizes)
    assert len(peers) == len({g.neighbour(dx, dy) for dx, dy in g.DIRS}) and sum(b - a for a, b in sspan.values()) == sum(sizes)
    sbuf, rbuf = torch.zeros(sum(sizes) // 4), torch.full((sum(sizes) // 4,), -7.0)
    for k, (dx, dy) in enumerate(g.DIRS):
        blk = plane[g._band(dx, dy)].reshape(-1)
        sbuf[soff[k] // 4:soff[k] // 4 + 2] = torch.tensor([float(rank), float(k)])
        sbuf[soff[k] // 4 + 2:(soff[k] + sizes[k]) // 4] = blk
    comm.exchange([(p, sbuf[sspan[p][0] // 4:sspan[p][1] // 4]) for p in peers], [(p, rbuf[rspan[p][0] // 4:rspan[p][1] // 4]) for p in peers])
    for k, (dx, dy) in enumerate(g.DIRS):
        tag = rbuf[roff[k] // 4:roff[k] // 4 + 2].tolist()
        assert tag == [float(g.neighbour(dx, dy)), float(nd - 1 - k)], f'rank {rank} side {k}: block of {tag}'
        plane[g._halo(dx, dy)] = rbuf[roff[k] // 4 + 2:(roff[k] + sizes[k]) // 4].reshape(shapes[k])
    ix = (np.arange(g.W) + g.ox) % world[0]
    iy = (np.arange(g.H) + g.oy) % world[1]
    assert np.array_equal(plane.numpy(), gid[ix][:, iy]), f'rank {rank}: halo mismatch'


@pytest.mark.parametrize('size,world,grid,h', [(2, (12, 16), (1, 2), (0, 4)), (2, (16, 12), (2, 1), 4), (4, (16, 24), (2, 2), 5),
                                                (4, (32, 8), (4, 1), (3, 0)), (1, (8, 8), (1, 1), 3)])
def test_one_message_per_peer_carries_every_side_to_its_opposite(size, world, grid, h):
    _run(size, _peer_message_case, world, grid, h)


def _value(q, gx, gy):
    """Deterministic pseudo-random claim word of rank q for world cell (gx, gy)."""
    x = (np.uint64(q + 1) * np.uint64(0x9E3779B97F4A7C15)) ^ (gx.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)) \
        ^ (gy.astype(np.uint64) * np.uint64(0x165667B19E3779F9))
    return (x ^ (x >> np.uint64(29))) * np.uint64(0xBF58476D1CE4E5B9)


def _merge_case(rank, size, world, grid, h):
    comm = Comm()
    geos = [TileGeometry(world, grid, q, h) for q in range(size)]

    def plane_of(q):
        g = geos[q]
        gx = ((np.arange(g.W) + g.ox) % world[0])[:, None] + np.zeros((1, g.H), dtype=np.int64)
        gy = ((np.arange(g.H) + g.oy) % world[1])[None, :] + np.zeros((g.W, 1), dtype=np.int64)
        return _value(q, gx, gy), gx, gy
    mine, _, _ = plane_of(rank)
    plane = torch.from_numpy(mine.view(np.int64).copy())
    halo_merge_max(plane, geos[rank], comm)
    g = geos[rank]
    want = mine.copy()
    for q in range(size):                      # every rank's halo cells that are images of my interior cells
        vals, gx, gy = plane_of(q)
        gq = geos[q]
        halo = np.ones((gq.W, gq.H), dtype=bool)
        halo[gq.h:gq.h + gq.Wi, gq.h:gq.h + gq.Hi] = False
        inside = halo & (gx >= g.x0) & (gx < g.x0 + g.Wi) & (gy >= g.y0) & (gy < g.y0 + g.Hi)
        li, lj = gx[inside] - g.ox, gy[inside] - g.oy
        np.maximum.at(want, (li, lj), vals[inside])
    got = plane.numpy().view(np.uint64)
    ri, ci = g.interior()
    assert np.array_equal(got[ri, ci], want[ri, ci]), f'rank {rank}: merged interior differs'
    mask = np.ones_like(got, dtype=bool)
    mask[ri, ci] = False
    assert np.array_equal(got[mask], mine[mask])      # the halo itself is left alone


@pytest.mark.parametrize('size,world,grid,h', [(2, (12, 16), (1, 2), 3), (2, (16, 12), (2, 1), 4), (4, (16, 24), (2, 2), 5),
                                                (4, (32, 8), (4, 1), 2), (1, (8, 8), (1, 1), 3)])
def test_halo_claim_merge_takes_the_maximum_over_all_images(size, world, grid, h):
    _run(size, _merge_case, world, grid, h)


def _route_case(rank, size, n):
    comm = Comm()
    rs = np.random.RandomState(100 + rank)
    dest = torch.from_numpy(rs.randint(0, size, n))
    payload = torch.from_numpy(np.stack([np.full(n, rank), np.arange(n), dest.numpy()]).astype(np.int32))
    kept, arrivals = route_records(payload, dest, comm)
    assert kept.sum().item() == int((dest == rank).sum())
    a = arrivals.numpy()
    assert (a[2] == rank).all() and (a[0] != rank).all()
    # every rank can recompute what the others drew: nothing lost, nothing duplicated
    want = 0
    for r in range(size):
        if r != rank:
            want += int((np.random.RandomState(100 + r).randint(0, size, n) == rank).sum())
    assert a.shape[1] == want
    assert len({(int(s), int(i)) for s, i in zip(a[0], a[1])}) == want


@pytest.mark.parametrize('size', [2, 4])
def test_route_records_delivers_every_record_once(size):
    _run(size, _route_case, 500)


def test_fill_holes_plans():
    rs = np.random.RandomState(0)
    for n, n_leave, n_arr in [(100, 10, 10), (100, 10, 25), (100, 30, 5), (50, 50, 0), (50, 0, 7), (40, 15, 0), (1, 1, 0)]:
        leaving = torch.zeros(n, dtype=torch.bool)
        leaving[torch.from_numpy(rs.permutation(n)[:n_leave])] = True
        vals = torch.arange(n + n_arr + 5)                       # capacity beyond n
        arrivals = torch.arange(1000, 1000 + n_arr)
        n_new, arr_dst, mv_src, mv_dst = fill_holes(n, leaving, n_arr)
        assert n_new == n - n_leave + n_arr
        if mv_src.numel():
            vals[mv_dst] = vals[mv_src]
        if n_arr:
            vals[arr_dst] = arrivals
        got = sorted(vals[:n_new].tolist())
        want = sorted(torch.arange(n)[~leaving].tolist() + arrivals.tolist())
        assert got == want


@pytest.mark.parametrize('grid', [(1, 2), (2, 1), (2, 2), (2, 4), (4, 2), (1, 8), (3, 3), (2, 3), (4, 4)])
def test_peer_message_layouts_of_all_ranks_fit_together(grid):
    """peer_message_layout for every rank of a grid at once (no processes: the layout is a pure function) — incl. the 2 x 4 grid
    `bench.py --gpus 8` launches, which no test can start as 8 processes here.  What rank r packs as its side k must lie, within the
    ONE message r sends to its neighbour p on that side, exactly where p expects the block of ITS opposite side from r; every
    message's length agrees on both ends; a rank posts one send and one receive per distinct peer."""
    Px, Py = grid
    world = (64 * Px, 64 * Py)
    h = (8 if Px > 1 else 0, 8 if Py > 1 else 0)
    geos = [TileGeometry(world, grid, q, h) for q in range(Px * Py)]
    lay = []
    for g in geos:
        nd = len(g.DIRS)
        sizes = [16 + 8 * k for k in range(nd)]
        sizes = [sizes[min(k, nd - 1 - k)] + 64 * (abs(g.DIRS[k][0]) + 2 * abs(g.DIRS[k][1])) for k in range(nd)]   # side k and its opposite: same size
        lay.append((sizes,) + peer_message_layout(g.DIRS, g.neighbour, sizes))
    for r, g in enumerate(geos):
        sizes, peers, soff, roff, sspan, rspan = lay[r]
        nd = len(g.DIRS)
        assert len(peers) == len(set(g.neighbour(dx, dy) for dx, dy in g.DIRS))
        for k, (dx, dy) in enumerate(g.DIRS):
            p_ = g.neighbour(dx, dy)
            psizes, ppeers, psoff, proff, psspan, prspan = lay[p_]
            ko = nd - 1 - k                                            # the peer's side that faces me: DIRS negated = reversed
            assert geos[p_].DIRS[ko] == (-dx, -dy) and geos[p_].neighbour(-dx, -dy) == r
            assert psizes[ko] == sizes[k]
            # position inside the message r -> p_ == position inside what p_ receives from r
            assert soff[k] - sspan[p_][0] == proff[ko] - prspan[r][0], (grid, r, k)
            assert sspan[p_][1] - sspan[p_][0] == prspan[r][1] - prspan[r][0]
        # blocks of one message do not overlap and fill it
        for p_ in peers:
            mine = sorted((soff[k], sizes[k]) for k in range(nd) if g.neighbour(*g.DIRS[k]) == p_)
            at = sspan[p_][0]
            for o, n in mine:
                assert o == at
                at += n
            assert at == sspan[p_][1]


def test_room_test_of_the_refresh_in_place_is_a_true_bound(monkeypatch):
    """DistEnv._inplace_room (host side of die_pic_ghost_inplace): the halo tiles' new segments hold what arrives (≤ the messages'
    capacities) plus today's halo population = the ghosts of the previous refresh + the owned agents that drifted into the halo
    since (≤ what that refresh sent; the capacities while that is unknown).  ADVICE r5: the owned count of the previous refresh
    alone was no bound.  Pure host arithmetic: no GPU, no process group."""
    from types import SimpleNamespace
    from die_amd.dist import DistEnv
    monkeypatch.delenv('DIE_REFRESH_IN_PLACE_FORCE', raising=False)
    P = SimpleNamespace(caps=[1000, 1000, 500])
    def env(capacity, owned, sent_prev=None):
        e = SimpleNamespace(capacity=capacity, _owned=owned)
        if sent_prev is not None:
            e._sent_prev = sent_prev
        return e
    n = 10000
    # ghosts = n - owned = 2000; arrivals <= 2500; drift <= sent_prev = 1800  ->  needs 10000 + 2500 + 2000 + 1800 = 16300
    assert DistEnv._inplace_room(env(16300, 8000, 1800), n, P)
    assert not DistEnv._inplace_room(env(16299, 8000, 1800), n, P)
    # what the previous refresh sent is not known yet: the capacities stand in (2500)  ->  17000
    assert DistEnv._inplace_room(env(17000, 8000), n, P) and not DistEnv._inplace_room(env(16999, 8000), n, P)
    # no refresh has happened at all: every agent may be a halo agent
    assert DistEnv._inplace_room(env(25000, None), n, P) and not DistEnv._inplace_room(env(24999, None), n, P)
    # round 5's test (n + caps + ghosts = 14500) would have said yes here although 1800 drifters need room too
    assert not DistEnv._inplace_room(env(14500, 8000, 1800), n, P)
    monkeypatch.setenv('DIE_REFRESH_IN_PLACE_FORCE', '1')          # (tests only: the kernels' own guards are what is exercised then)
    assert DistEnv._inplace_room(env(1, 8000, 1800), n, P)
