"""Multi-GPU Physarum step: 2-D block decomposition of the torus, one process per GPU
(DESIGN.md §7).  The reference is single-process; this layer is new design.

Each rank owns a Wi×Hi tile of the gW×gH world and keeps its planes padded by a halo of
`h = probe reach + gaussian radius` cells.  One step:

    die_agent_move            positions + destination tile of every local agent
    migrate                   agents whose new cell left the tile travel (with their action and
                              any attached Agent state) to the owning rank — point-to-point
    die_agent_claim_feed      claims / feeding on the owner rank
    die_agent_resolve         winners deposit, cells are fed
    halo exchange             post-deposit chem, width h, two phases (y then x → corners)
    die_diffuse_decay_tile    every rank diffuses interior + halo redundantly, so the diffused halo the
                              next forward() probes is already local: ONE exchange per step
    reward / num_agents       scalar all-reduce, only when the caller reads them

Transport is torch.distributed point-to-point (RCCL on GPUs: `backend='nccl'`; `gloo` for the
CPU tests and for several ranks sharing one GPU, staged through host memory).  Results do not
depend on the decomposition: Philox counters and ownership are keyed by global slot ids.
"""
import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


class TileGeometry:
    """Pure index arithmetic of the decomposition (no device, no communication)."""

    def __init__(self, world: Tuple[int, int], grid: Tuple[int, int], rank: int, halo):
        """`halo`: cells of padding per side, one number or (hx, hy).  An axis with halo 0 must have one rank
        along it: the tile spans the world there and the kernels wrap on it natively."""
        gW, gH = int(world[0]), int(world[1])
        Px, Py = int(grid[0]), int(grid[1])
        if gW % Px or gH % Py:
            raise ValueError(f'world {world} is not divisible by the rank grid {grid}')
        hx, hy = (int(halo), int(halo)) if np.isscalar(halo) else (int(halo[0]), int(halo[1]))
        if (hx == 0 and Px != 1) or (hy == 0 and Py != 1):
            raise ValueError('an axis without halo needs exactly one rank along it')
        self.gW, self.gH, self.Px, self.Py, self.rank = gW, gH, Px, Py, int(rank)
        self.hx, self.hy, self.h = hx, hy, max(hx, hy)
        self.px, self.py = divmod(self.rank, Py)
        self.Wi, self.Hi = gW // Px, gH // Py
        if hx > self.Wi or hy > self.Hi:
            raise ValueError(f'halo {halo} wider than the tile {(self.Wi, self.Hi)}')
        self.x0, self.y0 = self.px * self.Wi, self.py * self.Hi
        self.W, self.H = self.Wi + 2 * hx, self.Hi + 2 * hy
        self.ox, self.oy = self.x0 - hx, self.y0 - hy
        # the sides that have a halo, sorted: negation reverses the list
        self.DIRS = [(dx, dy) for dx, dy in self.ALL_DIRS if (dx == 0 or hx > 0) and (dy == 0 or hy > 0)]

    @property
    def size(self) -> int:
        return self.Px * self.Py

    def rank_at(self, px: int, py: int) -> int:
        return (px % self.Px) * self.Py + (py % self.Py)

    def neighbour(self, dx: int, dy: int) -> int:
        return self.rank_at(self.px + dx, self.py + dy)

    def tile_of_cell(self, ix, iy):
        return (np.asarray(ix) // self.Wi) * self.Py + np.asarray(iy) // self.Hi

    def interior(self) -> Tuple[slice, slice]:
        return slice(self.hx, self.hx + self.Wi), slice(self.hy, self.hy + self.Hi)

    ALL_DIRS = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]

    def _band(self, dx, dy):
        """Interior cells adjacent to side (dx, dy) — what the neighbour on that side needs as its halo."""
        hx, hy, Wi, Hi = self.hx, self.hy, self.Wi, self.Hi
        rows = {-1: slice(hx, 2 * hx), 0: slice(hx, hx + Wi), 1: slice(Wi, Wi + hx)}[dx]
        cols = {-1: slice(hy, 2 * hy), 0: slice(hy, hy + Hi), 1: slice(Hi, Hi + hy)}[dy]
        return rows, cols

    def _halo(self, dx, dy):
        """Halo cells on side (dx, dy) (edge strips exclude the corners, which are their own blocks)."""
        hx, hy, Wi, Hi, W, H = self.hx, self.hy, self.Wi, self.Hi, self.W, self.H
        rows = {-1: slice(0, hx), 0: slice(hx, hx + Wi), 1: slice(hx + Wi, W)}[dx]
        cols = {-1: slice(0, hy), 0: slice(hy, hy + Hi), 1: slice(hy + Hi, H)}[dy]
        return rows, cols

    def plan8(self):
        """One-phase exchange with all 8 neighbours: [(peer, send_view)] in DIRS order and [(peer, recv_view)] in
        reversed DIRS order.  The message sent towards s is the one the peer receives for its side −s, and
        reversing the receive order makes the k-th send to a peer meet the k-th receive posted for it, also when
        one rank is the neighbour on several sides (2-rank axes, self-neighbours)."""
        sends = [(self.neighbour(dx, dy), self._band(dx, dy)) for dx, dy in self.DIRS]
        recvs = [(self.neighbour(dx, dy), self._halo(dx, dy)) for dx, dy in reversed(self.DIRS)]
        return sends, recvs

    def reverse_plan8(self):
        """Claim merge in one phase: every halo block goes to the rank that owns those cells and is max-merged
        into the matching interior band there."""
        sends = [(self.neighbour(dx, dy), self._halo(dx, dy)) for dx, dy in self.DIRS]
        merges = [(self.neighbour(dx, dy), self._band(dx, dy)) for dx, dy in reversed(self.DIRS)]
        return sends, merges


def peer_message_layout(dirs, neighbour, sizes):
    """ONE message per peer for blocks that are defined per side (a side = one of the up to 8 directions with a halo).
    `dirs`: TileGeometry.DIRS (sorted so that negation reverses the list), `neighbour(dx, dy)` -> rank, `sizes[k]`: bytes of
    side k's block (side k of a rank and side nd - 1 - k of its neighbour there have the same shape).  Returns
    (peers, send_off, recv_off, send_span, recv_span): byte offset of side k's block in the send / receive buffer and, per peer,
    the (start, end) of its one message in each buffer.  The message for peer p holds the sender's sides towards p in ascending
    side order; the block the peer packed as ITS side j is what arrives for my side nd - 1 - j, so my sides from p lie in the
    message in DESCENDING order of my side index — also when one rank is the neighbour on several sides (2-rank axes: left and
    right neighbour are one rank; a 2 x 2 torus has 3 peers for 8 sides) or a rank is its own neighbour."""
    nd = len(dirs)
    peer = [neighbour(dx, dy) for dx, dy in dirs]
    peers = sorted(set(peer))
    send_off, recv_off, send_span, recv_span = [0] * nd, [0] * nd, {}, {}
    so = ro = 0
    for p_ in peers:
        mine = [k for k in range(nd) if peer[k] == p_]
        s0, r0 = so, ro
        for k in mine:
            send_off[k] = so
            so += sizes[k]
        for k in reversed(mine):
            recv_off[k] = ro
            ro += sizes[k]
        send_span[p_], recv_span[p_] = (s0, so), (r0, ro)
    return peers, send_off, recv_off, send_span, recv_span


class Comm:
    """Point-to-point transport over torch.distributed; `stage_cpu` routes device tensors through
    host memory (gloo).  Messages between one pair of ranks are matched in issue order."""

    def __init__(self, group=None, stage_cpu: Optional[bool] = None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        backend = dist.get_backend(group)
        self.stage_cpu = (backend == 'gloo') if stage_cpu is None else stage_cpu
        # DIE_DIST_SELF_VIA_BACKEND=1: messages a rank sends to itself (periodic self-neighbours) also go through
        # the backend's send/recv instead of a local copy — lets a single GPU rehearse the RCCL call pattern
        # (grouping, several messages per peer, matching order) that multi-GPU runs use.
        self.self_local = os.environ.get('DIE_DIST_SELF_VIA_BACKEND', '0') != '1'

    def exchange(self, sends: Sequence[Tuple[int, torch.Tensor]], recvs: Sequence[Tuple[int, torch.Tensor]]):
        """Send every (peer, tensor) and fill every (peer, buffer).  Self-messages are copied."""
        me = self.rank if self.self_local else -1
        self_send = [t for p, t in sends if p == me]
        self_recv = [t for p, t in recvs if p == me]
        assert len(self_send) == len(self_recv)
        for s, r in zip(self_send, self_recv):
            r.copy_(s.reshape(r.shape))
        ops, staged = [], []
        for p, t in sends:
            if p == me:
                continue
            buf = t.contiguous()
            if self.stage_cpu and buf.is_cuda:
                buf = buf.cpu()
            ops.append(dist.P2POp(dist.isend, buf, p, self.group))
        for p, t in recvs:
            if p == me:
                continue
            if (self.stage_cpu and t.is_cuda) or not t.is_contiguous():
                buf = torch.empty(t.shape, dtype=t.dtype, device='cpu' if self.stage_cpu else t.device)
                staged.append((buf, t))
            else:
                buf = t
            ops.append(dist.P2POp(dist.irecv, buf, p, self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for buf, t in staged:
            t.copy_(buf)

    def all_gather_counts(self, counts: torch.Tensor) -> torch.Tensor:
        """(size,) int64 per rank → (size, size) matrix, row r = what rank r sends to each rank."""
        c = counts.to('cpu' if self.stage_cpu else counts.device, torch.int64)
        out = [torch.empty_like(c) for _ in range(self.size)]
        dist.all_gather(out, c, group=self.group)
        return torch.stack(out).cpu()

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        c = t.cpu() if (self.stage_cpu and t.is_cuda) else t.clone()
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
        return c


_HALO_BUFFERS: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}


def _slice_rect(_lib, plane: torch.Tensor, view, offset: int):
    rs, cs = view
    return _lib.Rect(plane.data_ptr(), plane.shape[1], rs.start, rs.stop, cs.start, cs.stop, plane.element_size(), offset)


def _exchange_blocks(planes, sends, recvs, comm: Comm, merge_max: bool, tag: str, cache: Optional[dict] = None):
    """Move blocks of padded planes between ranks: `sends` / `recvs` are [(peer, view)] lists (views = (row slice,
    column slice)), every plane travels in the same message.  GPU: one pack launch, one message per
    neighbour, one unpack (or max-merge) launch.  CPU (tests): plain slicing."""
    on_gpu = all(p.is_cuda for p in planes)
    if not on_gpu:
        msg_s = [(peer, torch.cat([p[v].reshape(-1).view(torch.uint8) for p in planes])) for peer, v in sends]
        msg_r = [(peer, torch.empty(sum(p[v].numel() * p.element_size() for p in planes), dtype=torch.uint8)) for peer, v in recvs]
        comm.exchange(msg_s, msg_r)
        for (peer, v), (_, buf) in zip(recvs, msg_r):
            off = 0
            for p in planes:
                n = p[v].numel() * p.element_size()
                blk = buf[off:off + n].view(p.dtype).reshape(p[v].shape)
                if merge_max:
                    cur = p[v].contiguous().numpy().view(np.uint64)
                    blk = torch.from_numpy(np.maximum(cur, blk.numpy().view(np.uint64)).view(np.int64))
                p[v] = blk
                off += n
        return
    from . import _lib
    from .device_array import stream_ptr
    dev = planes[0].device
    key = (tag, merge_max, comm.rank, tuple(p.data_ptr() for p in planes))
    cache = _EXCHANGE_PLANS if cache is None else cache
    plan = cache.get(key)
    if plan is None:
        def layout(items):
            rects, offs, off = [], [], 0
            for peer, v in items:
                offs.append(off)
                for p in planes:
                    rects.append(_slice_rect(_lib, p, v, off))
                    off += ((v[0].stop - v[0].start) * (v[1].stop - v[1].start) * p.element_size() + 7) & ~7
            offs.append(off)
            chunks = [rects[i:i + 16] for i in range(0, len(rects), 16)]
            return [((_lib.Rect * len(c))(*c), len(c)) for c in chunks], offs
        schunks, soffs = layout(sends)
        rchunks, roffs = layout(recvs)
        sbuf = torch.empty(soffs[-1], dtype=torch.uint8, device=dev)
        rbuf = torch.empty(roffs[-1], dtype=torch.uint8, device=dev)
        smsg = [(peer, sbuf[soffs[k]:soffs[k + 1]]) for k, (peer, v) in enumerate(sends)]
        rmsg = [(peer, rbuf[roffs[k]:roffs[k + 1]]) for k, (peer, v) in enumerate(recvs)]
        ops = None
        if not comm.stage_cpu:                      # RCCL: the P2P ops over the persistent buffers are built once
            me = comm.rank if comm.self_local else -1
            selfs = [t for p_, t in smsg if p_ == me]
            selfr = [t for p_, t in rmsg if p_ == me]
            ops = ([dist.P2POp(dist.isend, t, p_, comm.group) for p_, t in smsg if p_ != me] +
                   [dist.P2POp(dist.irecv, t, p_, comm.group) for p_, t in rmsg if p_ != me], list(zip(selfr, selfs)))
        plan = (schunks, rchunks, sbuf, rbuf, smsg, rmsg, ops, planes)     # `planes` keeps the pointers alive
        cache[key] = plan
    schunks, rchunks, sbuf, rbuf, smsg, rmsg, ops, _ = plan
    sp = stream_ptr(dev)
    for arr, n in schunks:
        _lib.check(_lib.lib.die_rects_pack(arr, n, C.c_void_p(sbuf.data_ptr()), sp), 'die_rects_pack')
    if ops is None:
        comm.exchange(smsg, rmsg)
    else:
        p2p, self_pairs = ops
        for r, t in self_pairs:
            r.copy_(t)
        if p2p:
            for req in dist.batch_isend_irecv(p2p):
                req.wait()
    fn = _lib.lib.die_rects_unpack_max if merge_max else _lib.lib.die_rects_unpack
    for arr, n in rchunks:
        _lib.check(fn(arr, n, C.c_void_p(rbuf.data_ptr()), sp), 'die_rects_unpack')


_EXCHANGE_PLANS: Dict[tuple, tuple] = {}


def halo_exchange(planes, geo: TileGeometry, comm: Comm, cache: Optional[dict] = None):
    """Fill the halo ring of one or several padded (W, H) planes from the 8 periodic neighbours (edges and
    corners in ONE phase; all planes share the messages)."""
    if isinstance(planes, torch.Tensor):
        planes = [planes]
    sends, recvs = geo.plan8()
    _exchange_blocks(planes, sends, recvs, comm, False, 'halo%d' % len(planes), cache)


def halo_merge_max(plane: torch.Tensor, geo: TileGeometry, comm: Comm, cache: Optional[dict] = None):
    """Send the claim words of the halo ring to the ranks that own those cells; each rank raises its
    interior bands to the maximum of what it holds and what arrives (unsigned 64-bit compare)."""
    sends, merges = geo.reverse_plan8()
    _exchange_blocks([plane], sends, merges, comm, True, 'merge', cache)


def route_records(records: torch.Tensor, dest: torch.Tensor, comm: Comm) -> Tuple[torch.Tensor, torch.Tensor]:
    """Deliver the columns of `records` (F, n) to the ranks in `dest` (n,).  Returns
    (kept_mask, arrivals) where arrivals is (F, m) gathered from all other ranks in rank order.
    One host synchronisation (the count matrix); the outgoing records are grouped by one stable sort."""
    size, me = comm.size, comm.rank
    leaving = dest != me
    d = dest[leaving].to(torch.int64)
    counts = torch.bincount(d, minlength=size)
    matrix = comm.all_gather_counts(counts)                     # host: matrix[src, dst]
    incoming = matrix[:, me].tolist()
    outgoing = matrix[me].tolist()
    sends, recvs, parts = [], [], []
    if sum(outgoing):
        order = torch.sort(d, stable=True).indices
        grouped = records[:, leaving][:, order]
        start = 0
        for peer in range(size):
            if outgoing[peer]:
                sends.append((peer, grouped[:, start:start + outgoing[peer]].contiguous()))
                start += outgoing[peer]
    for peer in range(size):
        if peer != me and incoming[peer]:
            buf = torch.empty((records.shape[0], incoming[peer]), dtype=records.dtype, device=records.device)
            recvs.append((peer, buf))
            parts.append(buf)
    comm.exchange(sends, recvs)
    arrivals = torch.cat(parts, dim=1) if parts else records[:, :0]
    return ~leaving, arrivals


def fill_holes(n: int, leaving: torch.Tensor, n_arrive: int):
    """Index plan that removes `leaving` entries from an array prefix of length n and adds
    n_arrive new ones without moving more than the touched entries.
    Returns (n_new, arrive_dst, move_src, move_dst)."""
    holes = torch.nonzero(leaving[:n], as_tuple=False).squeeze(1)
    L = int(holes.numel())
    dev = leaving.device
    if n_arrive >= L:
        extra = torch.arange(n, n + n_arrive - L, device=dev)
        empty = holes[:0]
        return n + n_arrive - L, torch.cat([holes, extra]), empty, empty
    n_new = n - (L - n_arrive)
    rest = holes[n_arrive:]
    low = rest[rest < n_new]                                   # holes that stay inside the new prefix
    tail = torch.arange(n_new, n, device=dev)
    tail_keep = tail[~leaving[n_new:n]]                        # survivors beyond the new end
    return n_new, holes[:n_arrive], tail_keep, low


class DistEnv:
    """Env over a decomposed world.  Mirrors `die_amd.Env.step/_get_current_obs` per rank; `medium` /
    `agents` hold the local tile (halo-padded planes, local agents with global slot ids)."""

    def __init__(self, world: Tuple[int, int], grid: Tuple[int, int], dynamics=None, *, probe_reach: int,
                 capacity: Optional[int] = None, device=None, group=None, field_dtype=torch.float32,
                 seed: int = 0, init: bool = True, sort_every: int = 8, overlap: bool = True,
                 migrate_every: int = 1, max_step_cells: float = 2.0, ghosts: bool = False, ghost_headroom: float = 2.0,
                 pic: bool = True):
        from . import _lib
        from .data_init import DataInitializer
        from .device_array import DeviceAgents, DeviceMedium
        from .env import Dynamics
        self._lib = _lib
        self.dynamics = dynamics or Dynamics()
        if self.dynamics.apply_sense_mask:
            raise NotImplementedError('apply_sense_mask is implemented for single-tile worlds only')
        if self.dynamics.diffuse_mode != 'wrap':
            raise NotImplementedError("decomposed worlds diffuse on the torus (diffuse_mode='wrap') only")
        self.comm = Comm(group)
        R = int(4.0 * float(self.dynamics.diffuse_sigma) + 0.5)
        # guard band: agents may stay on a rank for `migrate_every` steps after leaving its interior (their
        # claims are merged across ranks every step instead); a lifecycle that teleports slots needs M = 1
        self.migrate_every = 1 if self.dynamics.agents_die else max(1, int(migrate_every))
        self.ghosts = bool(ghosts) and not self.dynamics.agents_die
        self.band = 0 if self.migrate_every == 1 else int(np.ceil(self.migrate_every * float(max_step_cells))) + 1
        if self.ghosts:
            # ghost agents: nothing crosses ranks for M steps.  Per step the exactly-known depth of the halo shrinks
            # by (probe reach + 1 tap) + (longest step) + (diffusion radius): a ghost on the frontier senses that far
            # beyond itself, its deposit diffuses R further (see _step_ghost).  One rank along an axis: no halo there.
            self.band = 0
            self._loss = int(probe_reach) + 1 + int(np.ceil(float(max_step_cells))) + R
            self._loss += int(os.environ.get('DIE_GHOST_LOSS_DELTA', '0'))      # testing: shows the bound is tight
            deep = self.migrate_every * self._loss
            hx = 0 if int(grid[0]) == 1 else deep
            hy = 0 if int(grid[1]) == 1 else deep
            while hy and (world[1] // grid[1] + 2 * hy) % 4:
                hy += 1
            if (hx and 2 * hx > world[0] // grid[0]) or (hy and 2 * hy > world[1] // grid[1]):
                raise ValueError(f'ghost halo {(hx, hy)} needs tiles of at least twice that size (tile '
                                 f'{(world[0] // grid[0], world[1] // grid[1])}): lower migrate_every')
            # tile-binned step on the padded tile (die_amd/pic.py; the N = 1 step on a padded tile): the planes must split
            # into whole tiles — a deeper halo is always valid, so it is rounded up to the next fit of the largest shape
            # — preferably to WHOLE TILES of halo (first pass): the refresh then goes by tiles (_refresh_ghosts_tiles)
            self._pic_tile = None
            if pic:
                from .pic import TILE_SHAPES
                for aligned, (xs, ys) in [(True, t) for t in TILE_SHAPES] + [(False, t) for t in TILE_SHAPES]:
                    TX, TY = 1 << xs, 1 << ys
                    fx, fy = hx, hy
                    if aligned:
                        fx, fy = -(-hx // TX) * TX, -(-hy // TY) * TY
                        if (world[0] // grid[0]) % TX or (world[1] // grid[1]) % TY:
                            continue
                    while fx and (world[0] // grid[0] + 2 * fx) % TX:
                        fx += 1
                    while fy and ((world[1] // grid[1] + 2 * fy) % TY or (world[1] // grid[1] + 2 * fy) % 4):
                        fy += 1
                    Wp, Hp = world[0] // grid[0] + 2 * fx, world[1] // grid[1] + 2 * fy
                    if Wp % TX or Hp % TY or Wp // TX < 3 or Hp // TY < 3 or 2 * fx > world[0] // grid[0] or 2 * fy > world[1] // grid[1]:
                        continue
                    if fx - hx > max(hx // 4, TX // 2) or fy - hy > max(hy // 4, TY // 2):      # (not at the price of much more halo)
                        continue
                    hx, hy, self._pic_tile = fx, fy, (xs, ys)
                    break
            halo = (hx, hy)
        else:
            self._pic_tile = None
            halo = self.band + int(probe_reach) + 1 + R
            while (world[1] // grid[1] + 2 * halo) % 4:                    # die_diffuse_decay_tile needs H % 4 == 0
                halo += 1
        self._probe_reach = int(probe_reach)
        self.geo = TileGeometry(world, grid, self.comm.rank, halo)
        self.R = R
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        g = self.geo
        self.medium = DeviceMedium((g.W, g.H), self.device, field_dtype)
        self.medium.world = (g.gW, g.gH, g.ox, g.oy)
        if self.ghosts:
            self.medium.own = (g.hx, g.hy, g.hx + g.Wi, g.hy + g.Hi)
        self._ghosts_fresh = False
        self._refresh_due, self._refresh_action = False, None     # a ghost refresh left for the next step (overlap)
        self._refresh_in_place = os.environ.get('DIE_REFRESH_IN_PLACE', '1') != '0'      # … keeps the interior tiles' segments where they are (die_pic_ghost_inplace)
        self._pic = None                 # PicState (die_amd/pic.py), built at the first step that qualifies
        self._pic_off = False            # the library refused the binned step for this configuration once
        self.pic_steps = 0               # steps taken by the tile-binned path (tests, bench)
        self._ghost_headroom = float(ghost_headroom)
        self._owned = None
        self._profile, self._prof = os.environ.get('DIE_DIST_PROFILE', '0') == '1', {}
        self._seed = int(seed)
        self._sort_every = int(sort_every)
        self._overlap = bool(overlap)
        self._steps = 0
        self.capacity = int(capacity) if capacity else None
        self.agents = None
        self.last_result = None
        # second HIP stream: the chem halo of the NEXT sweep is exchanged while forward / move /
        # migration / claims of the next step run on the compute stream
        self._comm_stream = torch.cuda.Stream(self.device) if self.device.type == 'cuda' else None
        self._chem_halo_in_flight = False
        self._plans = {}             # cached exchange plans (message buffers, rect lists, P2P ops)
        if init:
            self._init_tile()

    # ------------------------------------------------------------------ construction
    def _alloc_agents(self, n_local: int):
        from .device_array import DeviceAgents
        from .data_init import DataInitializer
        g = self.geo
        grow = (1 + 2 * g.hx / g.Wi) * (1 + 2 * g.hy / g.Hi) if self.ghosts else 1.0
        cap = self.capacity or max(int(n_local * 1.5 * grow) + 1024, 4096)
        self.capacity = cap
        A = DeviceAgents(cap, self.device)
        A.N = n_local
        A.global_slots = True
        A.slot = torch.zeros(cap, dtype=torch.int32, device=self.device)
        self.agents = A
        self._tile_of = torch.zeros(cap, dtype=torch.int32, device=self.device)
        self._all_alive = False
        self._workspace = DataInitializer.workspace((self.geo.W, self.geo.H), cap, self.device)
        self._shadow = None
        return A

    def _init_tile(self):
        """Per-rank device init: the world's synthetic medium restricted to this tile (Philox is
        keyed by world cell), agents seeded on interior cells, globally unique slot ids."""
        from .data_init import DataInitializer, food_spec_from_seed
        lib, g = self._lib, self.geo
        DataInitializer.init_medium(self.medium, self.dynamics.init_agent_ratio, self._seed)
        ri, ci = g.interior()
        mask = torch.zeros_like(self.medium.owner, dtype=torch.bool)
        mask[ri, ci] = True
        self.medium.owner.mul_(mask)                                   # seed agents on interior cells only
        k_local = int((self.medium.owner != 0).sum().item())
        counts = self.comm.all_gather_counts(torch.tensor([k_local], dtype=torch.int64, device=self.device))
        base = int(counts[:self.comm.rank].sum().item())
        self.world_agents = int(counts.sum().item())
        A = self._alloc_agents(k_local)
        tmp_n = A.N
        A.N = A.x.numel()                                              # die_init_agents zero-fills the tail
        count = torch.zeros(2, dtype=torch.int64, device=self.device)
        from .device_array import _ptr, stream_ptr
        m, a = self.medium.c_struct(), self._struct(A, with_slot=False)
        lib.check(lib.lib.die_init_agents(C.byref(m), C.byref(a), self._seed & 0xFFFFFFFFFFFFFFFF, _ptr(count),
                                          _ptr(self._workspace), self._workspace.numel(), stream_ptr(self.device)),
                  'die_init_agents')
        A.N = tmp_n
        A.slot[:k_local] = torch.arange(base, base + k_local, dtype=torch.int32, device=self.device)
        self._all_alive = not self.dynamics.agents_die

    def _struct(self, A, with_slot=True):
        from .device_array import _ptr
        return self._lib.Agents(A.N, _ptr(A.x), _ptr(A.y), _ptr(A.alive), _ptr(A.agent_food),
                                _ptr(A.slot) if with_slot else None)

    @classmethod
    def from_global_numpy(cls, medium: np.ndarray, agents: np.ndarray, grid, dynamics=None, *, probe_reach: int,
                          **kw) -> 'DistEnv':
        """Every rank cuts its tile and its agents out of the same global arrays (tests)."""
        from .device_array import to_q32
        medium = np.asarray(medium, dtype=np.float64)
        agents = np.asarray(agents, dtype=np.float64)
        env = cls((medium.shape[1], medium.shape[2]), grid, dynamics, probe_reach=probe_reach, init=False, **kw)
        g = env.geo
        ix = (np.arange(g.W) + g.ox) % g.gW
        iy = (np.arange(g.H) + g.oy) % g.gH
        local = medium[:, ix][:, :, iy]
        occ = local[0].copy()
        occ_int = np.zeros_like(occ)
        ri, ci = g.interior()
        occ_int[ri, ci] = occ[ri, ci]
        env.medium.upload(np.stack([occ_int, local[1], local[2]]))
        qx, qy = to_q32(agents[0]).astype(np.uint64), to_q32(agents[1]).astype(np.uint64)
        cx = ((qx * np.uint64(g.gW - 1) + np.uint64(2 ** 31)) >> np.uint64(32)).astype(np.int64)
        cy = ((qy * np.uint64(g.gH - 1) + np.uint64(2 ** 31)) >> np.uint64(32)).astype(np.int64)
        mine = np.nonzero(g.tile_of_cell(cx, cy) == g.rank)[0]
        A = env._alloc_agents(len(mine))
        n = len(mine)
        A.x[:n] = torch.from_numpy(qx[mine].astype(np.uint32).view(np.int32)).to(env.device)
        A.y[:n] = torch.from_numpy(qy[mine].astype(np.uint32).view(np.int32)).to(env.device)
        A.alive[:n] = torch.from_numpy((agents[2, mine] > 0).astype(np.uint8)).to(env.device)
        A.agent_food[:n] = torch.from_numpy(agents[3, mine].astype(np.float32)).to(env.device)
        A.slot[:n] = torch.from_numpy(mine.astype(np.int32)).to(env.device)
        env.world_agents = agents.shape[1]
        env._all_alive = bool((agents[2] > 0).all()) and not env.dynamics.agents_die
        return env

    def local_slots(self) -> torch.Tensor:
        return self.agents.slot[:self.agents.N].to(torch.int64)

    # ------------------------------------------------------------------ step
    @property
    def _get_current_obs(self):
        return self.agents, self.medium

    def _c_dynamics(self):
        from .env import BoundaryCondition, linear_action_cost
        d, lib = self.dynamics, self._lib
        boundary = lib.DIE_BOUNDARY_WRAP if d.boundary == BoundaryCondition.wrap else lib.DIE_BOUNDARY_LIMIT
        cost = lib.DIE_COST_LINEAR if d.op_action_cost is linear_action_cost else lib.DIE_COST_ZERO
        return lib.Dynamics(d.rate_feed, d.rate_decay_chem, d.diffuse_sigma, boundary, cost, 0.02, 0.01,
                            int(d.food_infinite), int(d.agents_die), int(not self._all_alive), 0)

    def _per_agent_tensors(self, action, with_action: bool = True) -> List[torch.Tensor]:
        """Every per-agent array that must travel with a migrating agent (4-byte views).  The action travels only
        while it still means something (materialised and not yet stepped)."""
        A = self.agents
        ts = [A.x, A.y, A.slot, A.agent_food]
        if with_action:
            ts += [action.data[0], action.data[1], action.data[2]]
        for obj in A.attached():
            ts += obj._die_state_tensors(A)
        return ts

    def _record_arrays(self, tensors):
        arrs = tensors + [self.agents.alive]
        ptrs = (C.c_void_p * len(arrs))(*[t.data_ptr() for t in arrs])
        esz = (C.c_int32 * len(arrs))(*[t.element_size() for t in arrs])
        return arrs, ptrs, esz

    def _migrate(self, action):
        """Send agents whose cell left the tile to the owning rank and take in the arrivals: one
        gather launch for the leavers' records, one scatter launch for the arrivals (and one pair for the
        tail entries that move into remaining holes)."""
        from .device_array import stream_ptr
        A, comm, lib = self.agents, self.comm, self._lib
        if comm.size == 1:
            return
        n = A.N
        dest = self._tile_of[:n]
        tensors = self._per_agent_tensors(action)
        arrs, ptrs, esz = self._record_arrays(tensors)
        sp = stream_ptr(self.device)
        lv_idx = torch.nonzero(dest != comm.rank, as_tuple=False).squeeze(1)
        L = int(lv_idx.numel())
        rec = torch.empty((len(arrs), L), dtype=torch.int32, device=self.device)
        lib.check(lib.lib.die_records_gather(ptrs, esz, len(arrs), C.c_void_p(lv_idx.data_ptr()), L,
                                             C.c_void_p(rec.data_ptr()), sp), 'die_records_gather')
        _, arrivals = route_records(rec, dest[lv_idx], comm)
        n_arr = int(arrivals.shape[1])
        mask = torch.zeros(max(n, 1), dtype=torch.bool, device=self.device)
        mask[lv_idx] = True
        n_new, arr_dst, mv_src, mv_dst = fill_holes(n, mask, n_arr)
        if n_new > self.capacity:
            raise RuntimeError(f'rank {comm.rank}: {n_new} agents exceed the local capacity {self.capacity}')
        if mv_src.numel():
            tmp = torch.empty((len(arrs), int(mv_src.numel())), dtype=torch.int32, device=self.device)
            lib.check(lib.lib.die_records_gather(ptrs, esz, len(arrs), C.c_void_p(mv_src.data_ptr()), mv_src.numel(),
                                                 C.c_void_p(tmp.data_ptr()), sp), 'die_records_gather')
            lib.check(lib.lib.die_records_scatter(ptrs, esz, len(arrs), C.c_void_p(mv_dst.data_ptr()), mv_dst.numel(),
                                                  C.c_void_p(tmp.data_ptr()), sp), 'die_records_scatter')
        if n_arr:
            arrivals = arrivals.contiguous()
            lib.check(lib.lib.die_records_scatter(ptrs, esz, len(arrs), C.c_void_p(arr_dst.data_ptr()), n_arr,
                                                  C.c_void_p(arrivals.data_ptr()), sp), 'die_records_scatter')
        A.N = n_new
        action.N = n_new

    def _start_chem_halo(self):
        """chem is final until the next sweep: start its halo exchange on the comm stream now.  The
        forward pass that runs meanwhile only reads halo cells whose redundantly diffused values equal
        the incoming ones bit for bit (same kernel, same inputs), so the overlap is race-free in value."""
        if self._comm_stream is None:
            halo_exchange([self.medium.chem], self.geo, self.comm, self._plans)
        else:
            self._comm_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._comm_stream):
                halo_exchange([self.medium.chem], self.geo, self.comm, self._plans)
        self._chem_halo_in_flight = True

    def step(self, action):
        """One decomposed env step.  Returns (obs, result_tensor): result is the LOCAL
        die_step_result; `read_result` all-reduces it."""
        if self.ghosts:
            return self._step_ghost(action)
        if self.migrate_every > 1:
            return self._step_guard_band(action)
        return self._step_migrate_each(action)

    # ------------------------------------------------------------------ ghost-agent mode
    def _step_ghost(self, action):
        """Communication-avoiding mode: besides its own agents a rank steps GHOSTS, copies of the neighbours' agents
        that stand in its halo.  Every per-agent computation is keyed by the world slot id, so a ghost and its
        original do the same thing bit for bit as long as what they sense is the same; claims, feeding, deposits and
        diffusion of the halo are recomputed locally instead of exchanged.  The depth to which the halo is exact
        shrinks by probe reach + 1 + step + radius cells per step, so with a halo of M times that
            M steps run with NO communication and no host synchronisation (the single-GPU kernels on the padded tile),
        then `_refresh_ghosts` re-seats everything: agents are re-assigned by the cell they stand on (a rank already
        holds a valid copy of whoever walked into its interior: nothing is handed over), ghosts are dropped and
        re-sent from the owners' border bands, chem and food halos are exchanged — one message per neighbour.
        reward / num_agents count the agents standing on interior cells (die_medium.own_*)."""
        from .device_array import PendingAction, _ptr, stream_ptr
        lib, A, M = self._lib, self.agents, self.medium
        sp = stream_ptr(self.device)
        d = self._c_dynamics()
        result = torch.empty(2, dtype=torch.float64, device=self.device)
        if self._refresh_due:
            # the refresh that the previous step left for this one: the messages travel under this step's interior tiles
            self._refresh_due, self._refresh_action = False, None
            self.medium.before_sense = None
            self._send_action = False
            if self._pic_applies(action) and self._tile_refresh_applies():
                self._check_reach(action)
                self._check_seed(int(action.g_struct.seed))
                self._refresh_ghosts_tiles(action, step=(d, result))
                action.agent._forward_consumed(action)
                M.owner_stale = self._mark_owner
                self.pic_steps += 1
                self.overlapped_refreshes = getattr(self, 'overlapped_refreshes', 0) + 1
                return self._after_pic_step(action, result)
            self._refresh_ghosts(action)
        if not self._ghosts_fresh:
            self._refresh_ghosts(action)
        if self._pic_applies(action) and self._pic_step(action, d, result):
            return self._after_pic_step(action, result)
        if self._pic is not None:
            self._pic_void()                # a classic step moves the agents in place: the tile order is gone
        M.next_epoch()
        ws, wsn = _ptr(self._workspace), self._workspace.numel()
        second = not self._all_alive
        if A.N > 0:
            m, a = M.c_struct(), self._struct(A)
            if isinstance(action, PendingAction) and action.pending and action.agents is A and action.medium is M:
                self._check_reach(action)
                self._check_seed(int(action.g_struct.seed))
                u = action.raw_struct()
                lib.check(lib.lib.die_forward_move_claim_tile(C.byref(m), C.byref(a), C.byref(action.g_struct), C.byref(u),
                                                              C.byref(d), ws, wsn, sp), 'die_forward_move_claim_tile')
                action.agent._forward_consumed(action)
            else:
                action.N = A.N
                u = action.c_struct()
                lib.check(lib.lib.die_agent_move_claim(C.byref(m), C.byref(a), C.byref(u), C.byref(d), ws, wsn, sp),
                          'die_agent_move_claim')
            u = action.c_struct()
            if second:
                lib.check(lib.lib.die_agent_dead_slots(C.byref(m), C.byref(a), C.byref(u), C.byref(d), ws, wsn, sp),
                          'die_agent_dead_slots')
                lib.check(lib.lib.die_step_reduce_ex(C.byref(a), _ptr(result), ws, wsn, 1, -1, sp), 'die_step_reduce_ex')
        else:
            result.zero_()
        m = M.c_struct()
        if A.N > 0 and not second:                  # reduction folded into the sweep launch
            lib.check(lib.lib.die_tile_sweep_reduce(C.byref(m), C.byref(self._struct(A)), C.byref(d), 0, _ptr(result), ws, wsn, sp),
                      'die_tile_sweep_reduce')
        else:
            lib.check(lib.lib.die_medium_deposit_feed_diffuse_tile(C.byref(m), C.byref(d), 0, sp),
                      'die_medium_deposit_feed_diffuse_tile')
        M.swap_chem()
        self._food_flow()
        self._steps += 1
        if self._steps % self.migrate_every == 0:
            self._refresh_ghosts(action, after_step=True)
        if self._sort_every > 0 and self._steps % self._sort_every == 0 and A.N > 1:
            self.sort_agents()
        self.last_result = result
        return self._get_current_obs, result

    def _after_pic_step(self, action, result):
        M = self.medium
        M.swap_chem()
        self._food_flow()
        self._steps += 1
        if self._steps % self.migrate_every == 0:
            if self._overlap and self.geo.DIRS and self._tile_refresh_could_apply():
                # … with the NEXT step (_step_ghost), as long as that step's forward runs fused: anything that senses the planes
                # before — a stand-alone forward (agent.lazy = False, an action read before the step), NCA sensing, a host read —
                # finds `before_sense` and the refresh happens first (collective: the ranks run the same program).  ADVICE r4.
                self._refresh_due = True            # (the consumed action is not kept: a reference would make flush_lazy() rebuild it)
                self.medium.before_sense = self.flush_refresh
            else:
                self._refresh_ghosts(action, after_step=True)
        self.last_result = result
        return self._get_current_obs, result

    def _tile_refresh_could_apply(self) -> bool:
        self._send_action = False
        return self._tile_refresh_applies()

    def flush_refresh(self):
        """A ghost refresh that was left for the next step (to travel under its interior tiles) is done NOW.  COLLECTIVE: every rank
        must call it (gather_world does; the rank-local observers — owned_mask, check, read_result — need no refresh)."""
        self.medium.before_sense = None
        if self._refresh_due:
            self._refresh_due = False
            self._refresh_ghosts(None, after_step=True)

    # -- tile-binned step on the padded tile (die_amd/pic.py, csrc/die_pic.hip TILED): a rank's step is the N = 1 step ------
    def _pic_applies(self, action) -> bool:
        from .device_array import PendingAction
        from .env import BoundaryCondition
        if self._pic_tile is None or self._pic_off or not self.ghosts or not self._all_alive or self.agents.N <= 0:
            return False
        if not (isinstance(action, PendingAction) and action.pending and action.agents is self.agents and action.medium is self.medium):
            return False
        ag, dyn, g = action.agent, self.dynamics, self.geo
        if dyn.agents_die or not isinstance(dyn.boundary, BoundaryCondition) or not 1 <= self.R <= 4:
            return False
        if not (ag._normalized and ag._inertia == 0 and ag._noise_scale == 0 and ag._step_base is None and ag._prev_grad is None
                and ag._turn_sign is None):
            return False
        TX, TY = 1 << self._pic_tile[0], 1 << self._pic_tile[1]
        wmax = max(g.gW, g.gH) - 1
        reach = float(np.float32(abs(ag._scale)) * np.float32(wmax))
        probe = int(np.floor(float(np.float32(abs(ag._sense_offset_scale)) * np.float32(wmax)))) + 2
        vec = 4 if self.medium.dtype == torch.float32 else 8
        return int(reach) + 2 + self.R <= min(TX, TY) and (probe + vec - 1) // vec * vec <= 24

    def _pic_void(self):
        if self._pic is not None:
            self._pic.flush_lazy()
            self._pic.held = None

    def _pic_step(self, action, d, result) -> bool:
        from .pic import PicState
        lib, ag = self._lib, action.agent
        self._check_reach(action)
        self._check_seed(int(action.g_struct.seed))
        if self._pic is None:
            self._pic = PicState(self, self._pic_tile)
        if not self._pic.is_current(self, ag):
            self._pic.bin(self, ag)
            action.rebind(self.agents)
        # the step after which a refresh falls due (and will be left for the next step): the band tiles' agents are final once the
        # agent kernel has run, so they are packed on the second stream UNDER this step's field kernel, not in front of the refresh
        plan = None
        if self._overlap and self.geo.DIRS and self._comm_stream is not None and (self._steps + 1) % self.migrate_every == 0 \
                and os.environ.get('DIE_REFRESH_EARLY_PACK', '1') != '0' and self._pic.two_launch(self, ag) and self._early_pack_ok():
            plan = [(1, None), self._pack_bands_early, (2, None)]
        self._packed_early = None
        rc = self._pic.step(self, ag, action, d, result, plan=plan)
        if rc == -3:                         # DIE_ERR_UNSUPPORTED: nothing was launched; the classic step takes over for good
            self._pic_off = True
            self._pic_void()
            return False
        lib.check(rc, 'die_pic_forward_env_step')
        ag._forward_consumed(action)
        self.medium.owner_stale = self._mark_owner
        self.pic_steps += 1
        return True

    def _early_pack_ok(self) -> bool:
        """Will the refresh that falls due after this step go by tiles?  (What _tile_refresh_applies asks, for the layout the step
        is about to write: the geometry, not the arrays.)"""
        g, pic = self.geo, self._pic
        TX, TY = 1 << pic.xs, 1 << pic.ys
        if os.environ.get('DIE_GHOST_REFRESH', 'native') in ('torch', 'agents') or self.device.type != 'cuda' or not self.ghosts:
            return False
        return not (g.hx % TX or g.hy % TY or g.Wi % TX or g.Hi % TY or (g.hx and g.Wi < 2 * g.hx) or (g.hy and g.Hi < 2 * g.hy))

    def _pack_bands_early(self):
        """Between the agent kernel and the field kernel of a step (PicState.step(plan=…)): die_pic_ghost_pack of the layout the agent
        kernel has just written, on the second stream.  The field kernel reads those arrays too and writes none of them."""
        from .device_array import _ptr
        lib, pic = self._lib, self._pic
        P = self.__dict__.get('_tplan')
        if P is None:
            P = self._tplan = self._build_tile_plan()
        main = torch.cuda.current_stream(self.device)
        m = self.medium.c_struct(need_owner=False)
        p = pic._struct(pic.held, pic.step_out)
        self._comm_stream.wait_stream(main)
        with torch.cuda.stream(self._comm_stream):
            lib.check(lib.lib.die_pic_ghost_pack(C.byref(m), C.byref(p), 1 - pic.cur, P.nd, P.sides, _ptr(P.summary), self._comm_stream.cuda_stream),
                      'die_pic_ghost_pack')
        self._packed_early = pic.step_out[0]             # (the x array of the packed layout: the refresh checks that it is the one it refreshes)

    def _mark_owner(self):
        """Rebuild the claim plane ('agents' channel) of the padded tile from the local agents after tile-binned steps."""
        from .device_array import stream_ptr
        M = self.medium
        M.next_epoch()
        m, a = M.c_struct(need_owner=False), self._struct(self.agents)
        self._lib.check(self._lib.lib.die_agents_mark_owner(C.byref(m), C.byref(a), stream_ptr(self.device)), 'die_agents_mark_owner')

    def check(self):
        """Synchronise; raise if the tile-binned step reported a bookkeeping error since the last check."""
        torch.cuda.synchronize(self.device)
        if self._pic is not None and self._pic.steps_since_check:
            self._pic.check()

    def _check_seed(self, seed: int):
        """A ghost must draw what its original draws: every rank's Agent needs the same seed (once per seed value)."""
        if getattr(self, '_seed_checked', None) == seed:
            return
        seeds = self.comm.all_gather_counts(torch.tensor([seed & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=self.device))
        if int(seeds.min()) != int(seeds.max()):
            raise ValueError('ghost-agent mode: the Agent objects of all ranks must be built with the same seed '
                             '(random streams are keyed by world slot id, not by rank)')
        self._seed_checked = seed

    def _food_flow(self):
        """Env._medium_resource_dynamics (core/env.py:147-150) on the padded tile: the WaveSequence operator evaluates
        its field at world cells, so halo cells get what their owners compute."""
        from .env import _identity_food_flow
        op = self.dynamics.op_food_flow
        if op is _identity_food_flow:
            return
        from .data_init import WaveFoodFlow
        if not isinstance(op, WaveFoodFlow):
            raise NotImplementedError('decomposed worlds run device-side food-flow operators only (WaveSequence.get_flow_operator)')
        op.apply(self.medium)

    def _cells(self):
        """World cell of every local agent, as offsets from this rank's interior origin (periodic)."""
        A, g = self.agents, self.geo
        n = A.N
        cx = ((A.x[:n].to(torch.int64) & 0xFFFFFFFF) * (g.gW - 1) + (1 << 31)) >> 32
        cy = ((A.y[:n].to(torch.int64) & 0xFFFFFFFF) * (g.gH - 1) + (1 << 31)) >> 32
        return (cx - g.x0) % g.gW, (cy - g.y0) % g.gH

    def owned_mask(self) -> torch.Tensor:
        """Which local agents this rank accounts for (ghost mode: those standing on interior cells — also true while a refresh is
        waiting for the next step: every agent on an interior cell is a valid copy)."""
        if not self.ghosts:
            return torch.ones(self.agents.N, dtype=torch.bool, device=self.device)
        lx, ly = self._cells()
        return (lx < self.geo.Wi) & (ly < self.geo.Hi)

    def _tick(self, name=None, stagger=False):
        """DIE_DIST_PROFILE=1: synchronising phase timer for _refresh_ghosts (scratch/ghost_phases.py reads self._prof).
        `stagger` (DIE_DIST_PROFILE_STAGGER=1, ranks sharing one GPU): the ranks take the next phases one after the other, so
        that a phase's time is its own and not its share of a GPU that the other ranks use at the same moment.
        DIE_DIST_PROFILE_EVENTS=1: the phases are bracketed by HIP events instead of host synchronisations — device time of what
        was enqueued, as in a production run where the host does not wait between phases (a stagger point still synchronises)."""
        if not self._profile:
            return
        import time
        events = os.environ.get('DIE_DIST_PROFILE_EVENTS', '0') == '1'
        staggering = stagger and os.environ.get('DIE_DIST_PROFILE_STAGGER', '0') == '1'
        if events and not staggering:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            pend = self.__dict__.setdefault('_prof_events', [])
            pend.append((name, ev))
            return
        torch.cuda.synchronize(self.device)
        if events:                                             # settle the pending event intervals
            pend = self.__dict__.get('_prof_events', [])
            for (_, a), (nm, b) in zip(pend, pend[1:]):
                if nm is not None:
                    self._prof[nm + ' [device]'] = self._prof.get(nm + ' [device]', 0.0) + a.elapsed_time(b) * 1e-3
            self._prof_events = []
        if staggering:
            dist.barrier(self.comm.group)
            time.sleep(0.004 * self.comm.rank)
        now = time.perf_counter()
        if name is not None and not events:
            self._prof[name] = self._prof.get(name, 0.0) + now - self._t_last
        self._t_last = now
        if events:
            if staggering:
                torch.cuda._sleep(400000)        # ≈ 0.2 ms of spinning first: what follows is queued behind it, so the intervals hold no host latency
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._prof_events = [(None, ev)]

    def _refresh_ghosts(self, action, after_step: bool = False):
        """Re-seat owners, ghosts and the chem / food halos (see _step_ghost)."""
        from .device_array import PendingAction
        if not self.geo.DIRS:                      # one rank: the tile is the world, nobody to exchange with (and the tile order stays)
            self._owned = self.agents.N
            self._ghosts_fresh = True
            return
        # a consumed action, or one whose forward() has not run yet, holds nothing worth sending
        self._send_action = not after_step and not (isinstance(action, PendingAction) and action.pending)
        if self._send_action and action.data.shape[1] < self.capacity:
            raise ValueError(f'the action arrays hold {action.data.shape[1]} slots, the local agent arrays {self.capacity}: arriving '
                             f'ghosts bring their action with them (build the action for `env.capacity` slots)')
        if self._tile_refresh_applies():
            return self._refresh_ghosts_tiles(action)
        self._pic_void()                    # arrays are edited in place below: an un-read lazy action first, then no tile order
        if self.device.type == 'cuda' and os.environ.get('DIE_GHOST_REFRESH', 'native') != 'torch':
            return self._refresh_ghosts_native(action)
        return self._refresh_ghosts_torch(action)

    # -- refresh by tiles (csrc/die_pic_refresh.hip): the agents are in tile order and the halo is whole tiles deep, so "owned"
    #    and "in the band of side d" are properties of a TILE — nothing classifies agents, nothing re-bins afterwards ----------
    def _tile_refresh_applies(self) -> bool:
        pic = self._pic
        if pic is None or pic.held is None or pic.agent is None or self.device.type != 'cuda' or not self.ghosts:
            return False
        if os.environ.get('DIE_GHOST_REFRESH', 'native') in ('torch', 'agents') or self._send_action:
            return False
        g = self.geo
        TX, TY = 1 << pic.xs, 1 << pic.ys
        if g.hx % TX or g.hy % TY or g.Wi % TX or g.Hi % TY or (g.hx and g.Wi < 2 * g.hx) or (g.hy and g.Hi < 2 * g.hy):
            return False
        return pic.is_current(self, pic.agent)

    def _inplace_room(self, n: int, P) -> bool:
        """Is there surely room behind the arrays' old end for the halo tiles' new segments of a refresh in place?  The new
        segments hold what arrives (at most the messages' capacities) plus the agents the halo tiles hold today: the ghosts of
        the previous refresh (n − owned; n does not change between two refreshes) plus the owned agents that have drifted into
        the halo since — those stood in the bands then (a halo is as wide as `migrate_every` steps reach, the ghost mode's own
        precondition), so they are at most what the previous refresh SENT (the message capacities while that is unknown).  The
        owned count of the previous refresh alone was no bound.  `DIE_REFRESH_IN_PLACE_FORCE=1` skips the test
        (tests only: it sends a refresh that does not fit through the kernels' own capacity guards, RF_FLAG_CAPACITY)."""
        if os.environ.get('DIE_REFRESH_IN_PLACE_FORCE', '0') == '1':
            return True
        owned = self._owned
        ghosts = n if owned is None else max(n - int(owned), 0)
        drift = self.__dict__.get('_sent_prev')
        if drift is None:
            drift = sum(P.caps)
        return n + sum(P.caps) + ghosts + int(drift) <= self.capacity

    def _build_tile_plan(self):
        from types import SimpleNamespace
        lib, g, dev, pic = self._lib, self.geo, self.device, self._pic
        TX, TY = 1 << pic.xs, 1 << pic.ys
        nd = len(g.DIRS)
        P = SimpleNamespace(nd=nd, rects={})
        dens = self.world_agents / float(g.gW * g.gH)
        esz = self.medium.chem.element_size()
        # per side, relative to the start of its block: (records, chem block, food block), block size; the per-tile counts come first
        P.caps, rel, sizes, geo = [], [], [], []
        for dx, dy in g.DIRS:                         # identical on every rank: depends on the side's shape only
            (rs, cs), (hr, hc) = g._band(dx, dy), g._halo(dx, dy)
            cells = (rs.stop - rs.start) * (cs.stop - cs.start)
            cap = int(min(self.capacity, np.ceil(cells * dens * self._ghost_headroom) + 1024))
            ntx, nty = (rs.stop - rs.start) // TX, (cs.stop - cs.start) // TY
            # (every block on a 16-byte boundary, cap a multiple of 4: the vector pack / unpack of csrc/die_pack.hip k_rects_vec takes
            # 16-byte offsets only and would otherwise fall back to the scalar kernel depending on the parity of cap — ADVICE r4)
            cap = (cap + 3) & ~3
            blk = (cells * esz + 15) & ~15
            cnt = 16
            rec = (cnt + ntx * nty * 4 + 15) & ~15
            chem = (rec + 6 * cap * 4 + 15) & ~15
            food = chem + blk
            P.caps.append(cap); rel.append((cnt, rec, chem, food)); sizes.append(food + blk)
            geo.append((rs.start // TX, cs.start // TY, ntx, nty, hr.start // TX, hc.start // TY))
        # ONE message per peer (a 2 x 2 torus: 3 peers, not 8 messages — a grouped exchange costs per message, whatever its size)
        peers, soff, roff, sspan, rspan = peer_message_layout(g.DIRS, g.neighbour, sizes)
        P.soff = [tuple(soff[k] + r for r in rel[k]) for k in range(nd)]      # (counts, records, chem, food) of side k in sbuf
        P.roff = [tuple(roff[k] + r for r in rel[k]) for k in range(nd)]
        total = sum(sizes)
        P.sbuf = torch.zeros(max(total, 8), dtype=torch.uint8, device=dev)
        P.rbuf = torch.zeros(max(total, 8), dtype=torch.uint8, device=dev)
        sb, rb = P.sbuf.data_ptr(), P.rbuf.data_ptr()
        P.sides = (lib.PicSide * max(nd, 1))(*[lib.PicSide(*geo[k], P.caps[k], sb + P.soff[k][0], sb + P.soff[k][1], rb + P.roff[k][0], rb + P.roff[k][1])
                                               for k in range(nd)])
        P.smsg = [(p_, P.sbuf[sspan[p_][0]:sspan[p_][1]]) for p_ in peers]
        P.rmsg = [(p_, P.rbuf[rspan[p_][0]:rspan[p_][1]]) for p_ in peers]
        P.summary = torch.zeros(lib.PIC_GHOST_SUMMARY_WORDS, dtype=torch.int64, device=dev)
        P.ops = None
        if nd and not self.comm.stage_cpu:
            P.ops = ([dist.P2POp(dist.isend, t, p_, self.comm.group) for p_, t in P.smsg] +
                     [dist.P2POp(dist.irecv, t, p_, self.comm.group) for p_, t in P.rmsg])
        # the tiles of a step that need nothing from a neighbour (_refresh_step_tiles): the agent kernel of a tile reads its
        # chem window +- the probe margin and the segments of the 8 tiles around it, the field kernel the agent kernel's lists of
        # the 9 tiles around it — one ring of interior tiles for the first, two for the second stay behind (an axis without halo:
        # nothing to stay clear of)
        ntx_, nty_ = g.W // TX, g.H // TY
        hx_t, hy_t = g.hx // TX, g.hy // TY
        def inner(ring):
            x0, x1 = (hx_t + ring, ntx_ - hx_t - ring) if g.hx else (0, ntx_)
            y0, y1 = (hy_t + ring, nty_ - hy_t - ring) if g.hy else (0, nty_)
            return (x0, y0, max(x1 - x0, 0), max(y1 - y0, 0))
        P.inner_agents, P.inner_field = inner(1), inner(2)
        return P

    def _tile_field_rects(self, P):
        lib, g, M = self._lib, self.geo, self.medium
        key = (M.chem.data_ptr(), M.food.data_ptr())
        r = P.rects.get(key)
        if r is None:
            send, recv = [], []
            for k, (dx, dy) in enumerate(g.DIRS):
                send += [_slice_rect(lib, M.chem, g._band(dx, dy), P.soff[k][2]), _slice_rect(lib, M.food, g._band(dx, dy), P.soff[k][3])]
                recv += [_slice_rect(lib, M.chem, g._halo(dx, dy), P.roff[k][2]), _slice_rect(lib, M.food, g._halo(dx, dy), P.roff[k][3])]
            chunk = lambda rects: [((lib.Rect * len(c))(*c), len(c)) for c in (rects[i:i + 16] for i in range(0, len(rects), 16))]
            r = (chunk(send), chunk(recv), (M.chem, M.food))
            P.rects[key] = r
        return r

    def _refresh_ghosts_tiles(self, action, step=None):
        """The refresh by tiles.  `step` = (d, result): the step that follows runs inside the refresh — on the tiles that need
        nothing from a neighbour while the messages are in flight (second stream), on the others once they have arrived
        (BASELINE north_star: "halo exchange over xGMI overlapped with interior-tile compute on a second HIP stream")."""
        from .device_array import _ptr, stream_ptr
        A, g, comm, lib, dev, pic = self.agents, self.geo, self.comm, self._lib, self.device, self._pic
        n = A.N
        self._tick(stagger=True)
        pic.flush_lazy()                                   # (an un-read lazy action refers to the arrays that are replaced below)
        P = self.__dict__.get('_tplan')
        if P is None:
            P = self._tplan = self._build_tile_plan()
        nd, sp = P.nd, stream_ptr(dev)
        m = self.medium.c_struct(need_owner=False)
        pic._n_agents = int(n)
        out = pic._out_tensors(self)
        p = pic._struct(pic.held, out)
        cur = pic.cur
        # everything below is enqueued without looking at a count; the host reads the summary once, at the end
        early, self._packed_early = getattr(self, '_packed_early', None), None
        if early is not None and early is pic.held[0] and self._comm_stream is not None:
            torch.cuda.current_stream(dev).wait_stream(self._comm_stream)      # packed under the previous step's field kernel (_pack_bands_early)
            self.early_packs = getattr(self, 'early_packs', 0) + 1
        else:
            lib.check(lib.lib.die_pic_ghost_pack(C.byref(m), C.byref(p), cur, nd, P.sides, _ptr(P.summary), sp), 'die_pic_ghost_pack')
        self._tick('band tiles packed')
        send_r, recv_r, _ = self._tile_field_rects(P)
        sb, rb = C.c_void_p(P.sbuf.data_ptr()), C.c_void_p(P.rbuf.data_ptr())
        for arr, cnt in send_r:
            lib.check(lib.lib.die_rects_pack(arr, cnt, sb, sp), 'die_rects_pack')
        self._tick('field pack')
        works = None
        main = torch.cuda.current_stream(dev)
        if step is not None and P.ops is not None and self._comm_stream is not None:
            # the messages go out behind the pack kernels, on the second stream; this stream carries on with the interior
            self._comm_stream.wait_stream(main)
            with torch.cuda.stream(self._comm_stream):
                works = dist.batch_isend_irecv(P.ops)

        def exchange_and_unpack():
            if works is not None:
                for req in works:
                    req.wait()                             # (stream-ordered: this stream waits for the messages, the host does not)
            elif P.ops is None:
                comm.exchange(P.smsg, P.rmsg)
            else:
                for req in dist.batch_isend_irecv(P.ops):
                    req.wait()
            self._tick('exchange (records + fields, one message per peer)')
            self._tick(stagger=True)
            for arr, cnt in recv_r:
                lib.check(lib.lib.die_rects_unpack(arr, cnt, rb, sp), 'die_rects_unpack')
            self._tick('field unpack')

        copied = []

        def summary_on_its_way():
            """The summary is final once the halo tiles are laid: copied to pinned host memory on the second stream, next to the
            step's remaining kernels — the host then waits for THIS copy, not for the step's end, and queues the next step while the
            last tiles are still being stepped (read behind the step, the one host read of a refresh was 25–50 µs of idle GPU)."""
            if self._comm_stream is None:
                return
            if getattr(P, 'summary_host', None) is None:
                P.summary_host = torch.zeros(lib.PIC_GHOST_SUMMARY_WORDS, dtype=torch.int64).pin_memory()
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ev)
                P.summary_host.copy_(P.summary, non_blocking=True)
                done = torch.cuda.Event()
                done.record(self._comm_stream)
            copied.append(done)

        def adopt_new_layout():
            pic.cur = 1 - cur
            pic._adopt(self, pic.agent, out)
            if step is not None and hasattr(action, 'rebind'):
                action.rebind(A)                           # (forward() ran before the arrays were replaced)

        if step is None:
            exchange_and_unpack()
            lib.check(lib.lib.die_pic_ghost_merge(C.byref(m), C.byref(p), cur, nd, P.sides, self.capacity, _ptr(P.summary), sp),
                      'die_pic_ghost_merge')
            self._tick('new layout (scan, merge, words)')
            adopt_new_layout()
        elif self._refresh_in_place and self._inplace_room(n, P):
            # (… when the halo tiles' new segments surely fit behind the arrays' old end: what can arrive + the old halo agents)
            # IN PLACE (round 5): the step that follows reads the layout the step before left (interior tiles: stayers + leavers where
            # they are — the agent kernel gathers a tile's agents from there anyway; nothing is copied, where the merge rewrote every
            # tile's segment: 120 MB, ≈ 80 µs per refresh); only the halo tiles get new segments, behind the arrays' old end, once
            # the messages are here.  The layout never outlives this call: the step writes the other one, compact, as always.
            d, result = step
            if getattr(P, 'tail', None) is None:
                P.tail = torch.zeros(pic.NT, dtype=torch.int32, device=dev)
            lib.check(lib.lib.die_pic_ghost_inplace(C.byref(m), C.byref(p), cur, nd, P.sides, self.capacity, _ptr(P.summary), 1, _ptr(P.tail), sp),
                      'die_pic_ghost_inplace')
            self._tick('places of the interior segments (scan)')

            def second_half():
                exchange_and_unpack()
                lib.check(lib.lib.die_pic_ghost_inplace(C.byref(m), C.byref(p), cur, nd, P.sides, self.capacity, _ptr(P.summary), 2, _ptr(P.tail), sp),
                          'die_pic_ghost_inplace')
                self._tick('halo tiles: scan + new segments in place')
                summary_on_its_way()
            ia, if_ = P.inner_agents, P.inner_field
            rc = pic.step(self, pic.agent, action, d, result,
                          plan=[(1, (1,) + ia), (2, (1,) + if_), second_half, (1, (2,) + ia, 1), (2, (2,) + if_)])
            lib.check(rc, 'die_pic_forward_env_step')
            self._tick('agent + field kernel on the remaining tiles')
            self.inplace_refreshes = getattr(self, 'inplace_refreshes', 0) + 1
        else:
            d, result = step
            # the interior tiles' segments of the new layout need nothing that arrives …
            lib.check(lib.lib.die_pic_ghost_merge_phase(C.byref(m), C.byref(p), cur, nd, P.sides, self.capacity, _ptr(P.summary), 1, sp),
                      'die_pic_ghost_merge_phase')
            adopt_new_layout()

            def second_half():
                # … the halo tiles' segments, and the halo cells of the planes, do
                exchange_and_unpack()
                lib.check(lib.lib.die_pic_ghost_merge_phase(C.byref(m), C.byref(p), cur, nd, P.sides, self.capacity, _ptr(P.summary), 2, sp),
                          'die_pic_ghost_merge_phase')
                summary_on_its_way()
            ia, if_ = P.inner_agents, P.inner_field
            rc = pic.step(self, pic.agent, action, d, result,
                          plan=[(1, (1,) + ia), (2, (1,) + if_), second_half, (1, (2,) + ia), (2, (2,) + if_)])
            lib.check(rc, 'die_pic_forward_env_step')
            self._tick('agent + field kernel on the remaining tiles')
        if copied:
            copied[0].synchronize()
            t = P.summary_host.tolist()                                # the one host read (already on its way: summary_on_its_way)
        else:
            t = P.summary.cpu().tolist()                               # the one host read
        self._tick('counts to host')
        n_new, kept, sent, arrived, flags = t[0], t[1], t[2:2 + nd], t[10:10 + nd], int(t[18])
        for k in range(nd):
            if sent[k] > P.caps[k] or arrived[k] > P.caps[k]:
                raise RuntimeError(f'ghost refresh: {max(sent[k], arrived[k])} agents in the band of side {g.DIRS[k]}, the messages hold '
                                   f'{P.caps[k]}: build DistEnv with a larger ghost_headroom (agents cluster at this border)')
        if n_new > self.capacity:
            raise RuntimeError(f'rank {comm.rank}: {n_new} agents (ghosts included) exceed the local capacity {self.capacity}')
        if flags:
            raise RuntimeError('ghost refresh by tiles: ' + '; '.join(v for b, v in lib.PIC_GHOST_FLAGS.items() if flags & b))
        if n_new != kept + sum(arrived):
            raise RuntimeError(f'ghost refresh: device count {n_new} != {kept} owned + {sum(arrived)} arrived')
        if n_new > n:                                      # every slot is alive on this path; readers (gather_world) look at the bytes
            A.alive[n:n_new] = 1
        self._owned = kept
        self._sent_prev = int(sum(sent))                   # (bounds the drift into the halo until the next refresh: _inplace_room)
        if nd:
            self.ghost_fill = max(getattr(self, 'ghost_fill', 0.0), max(max(sent[k], arrived[k]) / P.caps[k] for k in range(nd)))
        A.N = n_new
        if action is not None:
            action.N = n_new
        self.tile_refreshes = getattr(self, 'tile_refreshes', 0) + 1
        self._ghosts_fresh = True

    # -- native refresh: one classification pass, records packed straight into fixed-size messages that also carry the
    #    field bands, ONE grouped send/recv per refresh, one host read (counts) -------------------------------------
    def _build_ghost_plan(self, F: int):
        from types import SimpleNamespace
        lib, g, dev = self._lib, self.geo, self.device
        nd = len(g.DIRS)
        P = SimpleNamespace(F=F, nd=nd, rects={})
        dens = self.world_agents / float(g.gW * g.gH)
        P.caps = []
        for dx, dy in g.DIRS:                         # identical on every rank: depends on the side's shape only
            rs, cs = g._band(dx, dy)
            cells = (rs.stop - rs.start) * (cs.stop - cs.start)
            P.caps.append(int(min(self.capacity, np.ceil(cells * dens * self._ghost_headroom) + 1024)))
        P.lists = [torch.empty(c, dtype=torch.int32, device=dev) for c in P.caps] + \
                  [torch.empty(self.capacity, dtype=torch.int32, device=dev)]
        P.list_ptrs = (C.c_void_p * (nd + 1))(*[t.data_ptr() for t in P.lists])
        P.list_caps = (C.c_int64 * (nd + 1))(*(P.caps + [self.capacity]))
        P.dirs = (C.c_int8 * max(2 * nd, 1))(*[v for d in g.DIRS for v in d])
        P.totals = torch.zeros(nd + 2, dtype=torch.int64, device=dev)
        P.ws = torch.empty(lib.lib.die_ghost_workspace_bytes(self.capacity), dtype=torch.uint8, device=dev)
        esz = self.medium.chem.element_size()
        P.off, off = [], 0                             # per side: (header, records, chem block, food block, end)
        for k, (dx, dy) in enumerate(g.DIRS):
            rs, cs = g._band(dx, dy)
            blk = ((rs.stop - rs.start) * (cs.stop - cs.start) * esz + 7) & ~7
            hdr, rec = off, off + 16
            chem = rec + F * P.caps[k] * 4
            chem = (chem + 7) & ~7
            food = chem + blk
            off = food + blk
            P.off.append((hdr, rec, chem, food, off))
        P.sbuf = torch.zeros(max(off, 8), dtype=torch.uint8, device=dev)
        P.rbuf = torch.zeros(max(off, 8), dtype=torch.uint8, device=dev)
        # the message arriving for my side d was packed by the neighbour as its side −d: same shape, same size
        P.recv_order = list(reversed(range(nd)))       # receives are posted in reversed side order (see plan8)
        P.smsg = [(g.neighbour(*g.DIRS[k]), P.sbuf[P.off[k][0]:P.off[k][4]]) for k in range(nd)]
        P.rmsg = [(g.neighbour(*g.DIRS[k]), P.rbuf[P.off[k][0]:P.off[k][4]]) for k in P.recv_order]
        P.hdr_idx = torch.tensor([P.off[k][0] // 8 for k in range(nd)], dtype=torch.int64, device=dev)
        P.c_caps = (C.c_int64 * max(nd, 1))(*P.caps)
        P.c_hdr = (C.c_int64 * max(nd, 1))(*[o[0] for o in P.off])
        P.c_rec = (C.c_int64 * max(nd, 1))(*[o[1] for o in P.off])
        P.n_new = torch.zeros(3 + 2 * nd, dtype=torch.int64, device=dev)      # die_ghost_apply's summary for the host
        P.ops = None
        if nd and not self.comm.stage_cpu:
            P.ops = ([dist.P2POp(dist.isend, t, p_, self.comm.group) for p_, t in P.smsg] +
                     [dist.P2POp(dist.irecv, t, p_, self.comm.group) for p_, t in P.rmsg])
        return P

    def _ghost_field_rects(self, P):
        lib, g, M = self._lib, self.geo, self.medium
        key = (M.chem.data_ptr(), M.food.data_ptr())
        r = P.rects.get(key)
        if r is None:
            send, recv = [], []
            for k, (dx, dy) in enumerate(g.DIRS):
                send += [_slice_rect(lib, M.chem, g._band(dx, dy), P.off[k][2]), _slice_rect(lib, M.food, g._band(dx, dy), P.off[k][3])]
                recv += [_slice_rect(lib, M.chem, g._halo(dx, dy), P.off[k][2]), _slice_rect(lib, M.food, g._halo(dx, dy), P.off[k][3])]
            chunk = lambda rects: [((lib.Rect * len(c))(*c), len(c)) for c in (rects[i:i + 16] for i in range(0, len(rects), 16))]
            r = (chunk(send), chunk(recv), (M.chem, M.food))
            P.rects[key] = r
        return r

    def _refresh_ghosts_native(self, action):
        from .device_array import _ptr, stream_ptr
        A, g, comm, lib, dev = self.agents, self.geo, self.comm, self._lib, self.device
        n = A.N
        self._tick(stagger=True)
        tensors = self._per_agent_tensors(action, self._send_action)
        arrs, ptrs, esz = self._record_arrays(tensors)
        F = len(arrs)
        plans = self.__dict__.setdefault('_gplans', {})
        P = plans.get(F)
        if P is None:
            P = plans[F] = self._build_ghost_plan(F)
        nd, sp = P.nd, stream_ptr(dev)
        m, a = self.medium.c_struct(), self._struct(A)
        # everything below is enqueued without looking at a count; the host reads them once, at the end
        lib.check(lib.lib.die_ghost_plan(C.byref(m), C.byref(a), nd, P.dirs, P.list_ptrs, P.list_caps, _ptr(P.totals),
                                         _ptr(P.ws), P.ws.numel(), sp), 'die_ghost_plan')
        lib.check(lib.lib.die_ghost_pack(ptrs, esz, F, nd, P.list_ptrs, _ptr(P.totals), P.c_caps, P.c_hdr, P.c_rec,
                                         _ptr(P.sbuf), sp), 'die_ghost_pack')
        self._tick('plan + record pack')
        if nd:
            send_r, recv_r, _ = self._ghost_field_rects(P)
            sb, rb = C.c_void_p(P.sbuf.data_ptr()), C.c_void_p(P.rbuf.data_ptr())
            for arr, cnt in send_r:
                lib.check(lib.lib.die_rects_pack(arr, cnt, sb, sp), 'die_rects_pack')
            self._tick('field pack')
            if P.ops is None:
                comm.exchange(P.smsg, P.rmsg)
            else:
                for req in dist.batch_isend_irecv(P.ops):
                    req.wait()
            self._tick('exchange (records + fields, one message per side)')
            self._tick(stagger=True)
            for arr, cnt in recv_r:
                lib.check(lib.lib.die_rects_unpack(arr, cnt, rb, sp), 'die_rects_unpack')
        lib.check(lib.lib.die_ghost_apply(ptrs, esz, F, nd, _ptr(P.totals), P.c_caps, P.c_hdr, P.c_rec, _ptr(P.rbuf),
                                          _ptr(P.lists[nd]), _ptr(P.ws), n, self.capacity, _ptr(P.n_new), sp), 'die_ghost_apply')
        self._tick('field unpack + arrivals + compaction')
        t = P.n_new.cpu().tolist()                                   # the one host read
        n_new, H, kept, sent, arrived = t[0], t[1], t[2], t[3:3 + nd], t[3 + nd:]
        self._tick('counts to host')
        if kept + H != n:
            raise RuntimeError(f'ghost refresh: {kept} owned + {H} dropped != {n} local agents')
        for k in range(nd):
            if sent[k] > P.caps[k] or arrived[k] > P.caps[k]:
                raise RuntimeError(f'ghost refresh: {max(sent[k], arrived[k])} agents in the band of side {g.DIRS[k]}, the messages hold '
                                   f'{P.caps[k]}: build DistEnv with a larger ghost_headroom (agents cluster at this border)')
        if n_new != n - H + sum(arrived):
            raise RuntimeError(f'ghost refresh: device count {n_new} != {n} - {H} + {sum(arrived)}')
        if n_new > self.capacity:
            raise RuntimeError(f'rank {comm.rank}: {n_new} agents (ghosts included) exceed the local capacity {self.capacity}')
        self._owned = kept
        self._sent_prev = int(sum(sent))
        if nd:
            self.ghost_fill = max(getattr(self, 'ghost_fill', 0.0), max(max(sent[k], arrived[k]) / P.caps[k] for k in range(nd)))
        A.N = n_new
        if action is not None:                 # (None: a refresh that had been left for the next step, done by flush_refresh)
            action.N = n_new
        self._ghosts_fresh = True

    # -- the same refresh written with torch index operations (cross-check of the native path: DIE_GHOST_REFRESH=torch) --
    def _refresh_ghosts_torch(self, action):
        from .device_array import stream_ptr
        A, g, comm, lib = self.agents, self.geo, self.comm, self._lib
        n = A.N
        self._tick()
        lx, ly = self._cells()
        keep = (lx < g.Wi) & (ly < g.Hi)
        near = {(-1, 0): lx < g.hx, (1, 0): lx >= g.Wi - g.hx, (0, -1): ly < g.hy, (0, 1): ly >= g.Hi - g.hy}
        idx = []
        for dx, dy in g.DIRS:                                  # copies for the neighbour on side (dx, dy)
            m = keep
            if dx:
                m = m & near[(dx, 0)]
            if dy:
                m = m & near[(0, dy)]
            idx.append(torch.nonzero(m, as_tuple=False).squeeze(1))
        nd = len(g.DIRS)
        counts = torch.tensor([int(i.numel()) for i in idx] + [int(keep.sum())], dtype=torch.int64, device=self.device)
        self._tick('plan (cells, masks, index lists)')
        matrix = comm.all_gather_counts(counts)                # host: matrix[rank] = (per-side counts…, owned)
        self._tick('count all_gather')
        self._owned = int(matrix[comm.rank, -1])
        self._sent_prev = int(matrix[comm.rank, :-1].sum())
        owned_world = int(matrix[:, -1].sum())
        if owned_world != self.world_agents:
            raise RuntimeError(f'ghost refresh: {owned_world} agents are owned, the world has {self.world_agents}: an agent moved '
                               f'further than max_step_cells per step or sensed further than probe_reach')
        tensors = self._per_agent_tensors(action, self._send_action)
        arrs, ptrs, esz = self._record_arrays(tensors)
        sp = stream_ptr(self.device)
        sends, recvs, parts = [], [], []
        if nd:
            out_idx = torch.cat(idx)
            L = int(out_idx.numel())
            rec = torch.empty((len(arrs), max(L, 1)), dtype=torch.int32, device=self.device)
            if L:
                lib.check(lib.lib.die_records_gather(ptrs, esz, len(arrs), C.c_void_p(out_idx.data_ptr()), L,
                                                     C.c_void_p(rec.data_ptr()), sp), 'die_records_gather')
            start = 0
            for k, (dx, dy) in enumerate(g.DIRS):              # k-th send to a peer meets the k-th receive posted for it
                c = int(matrix[comm.rank, k])
                if c:
                    sends.append((g.neighbour(dx, dy), rec[:, start:start + c].contiguous()))
                start += c
            for dx, dy in reversed(g.DIRS):                    # what arrives for my side (dx, dy) left as side (−dx, −dy)
                peer = g.neighbour(dx, dy)
                c = int(matrix[peer, g.DIRS.index((-dx, -dy))])
                if c:
                    buf = torch.empty((len(arrs), c), dtype=torch.int32, device=self.device)
                    recvs.append((peer, buf))
                    parts.append(buf)
            self._tick('records gather')
            comm.exchange(sends, recvs)
            self._tick('records exchange')
        arrivals = torch.cat(parts, dim=1).contiguous() if parts else None
        n_arr = int(arrivals.shape[1]) if arrivals is not None else 0
        n_new, arr_dst, mv_src, mv_dst = fill_holes(n, ~keep if n else torch.zeros(1, dtype=torch.bool, device=self.device), n_arr)
        if n_new > self.capacity:
            raise RuntimeError(f'rank {comm.rank}: {n_new} agents (ghosts included) exceed the local capacity {self.capacity}')
        if mv_src.numel():
            tmp = torch.empty((len(arrs), int(mv_src.numel())), dtype=torch.int32, device=self.device)
            lib.check(lib.lib.die_records_gather(ptrs, esz, len(arrs), C.c_void_p(mv_src.data_ptr()), mv_src.numel(),
                                                 C.c_void_p(tmp.data_ptr()), sp), 'die_records_gather')
            lib.check(lib.lib.die_records_scatter(ptrs, esz, len(arrs), C.c_void_p(mv_dst.data_ptr()), mv_dst.numel(),
                                                  C.c_void_p(tmp.data_ptr()), sp), 'die_records_scatter')
        if n_arr:
            lib.check(lib.lib.die_records_scatter(ptrs, esz, len(arrs), C.c_void_p(arr_dst.data_ptr()), n_arr,
                                                  C.c_void_p(arrivals.data_ptr()), sp), 'die_records_scatter')
        A.N = n_new
        if action is not None:
            action.N = n_new
        self._tick('compaction + scatter')
        if nd:
            halo_exchange([self.medium.chem, self.medium.food], g, comm, self._plans)
        self._tick('chem + food halo exchange')
        self._ghosts_fresh = True

    # ------------------------------------------------------------------ guard-band mode
    def _step_guard_band(self, action):
        """Agents stay where they are for `migrate_every` steps; what crosses ranks every step is fields only:
            die_forward_move_claim_tile     the single-GPU front kernel; a stray agent claims a HALO cell
            halo_merge_max(claims)          halo claims travel to the owners, max-merged into their interior
            halo_exchange(chem, claims)     authoritative interior bands → everybody's halo
            die_medium_deposit_feed_diffuse_tile(halo=0)   deposits, feeding (halo included: its food stays
                                            consistent with the owner's) and diffusion over the padded tile
        No host synchronisation in the loop; every `migrate_every` steps the strays are handed over."""
        from .device_array import PendingAction, _ptr, stream_ptr
        lib, A, g, M = self._lib, self.agents, self.geo, self.medium
        sp = stream_ptr(self.device)
        d = self._c_dynamics()
        M.next_epoch()
        result = torch.empty(2, dtype=torch.float64, device=self.device)
        ws, wsn = _ptr(self._workspace), self._workspace.numel()
        second = not self._all_alive
        if A.N > 0:
            m, a = M.c_struct(), self._struct(A)
            if isinstance(action, PendingAction) and action.pending and action.agents is A and action.medium is M:
                self._check_reach(action)
                u = action.raw_struct()
                lib.check(lib.lib.die_forward_move_claim_tile(C.byref(m), C.byref(a), C.byref(action.g_struct), C.byref(u),
                                                              C.byref(d), ws, wsn, sp), 'die_forward_move_claim_tile')
                action.agent._forward_consumed(action)
            else:
                action.N = A.N
                u = action.c_struct()
                lib.check(lib.lib.die_agent_move_claim(C.byref(m), C.byref(a), C.byref(u), C.byref(d), ws, wsn, sp),
                          'die_agent_move_claim')
        halo_merge_max(M.owner, g, self.comm, self._plans)
        if self._chem_halo_in_flight:
            if self._comm_stream is not None:
                torch.cuda.current_stream(self.device).wait_stream(self._comm_stream)
            self._chem_halo_in_flight = False
            halo_exchange([M.owner], g, self.comm, self._plans)
        else:
            halo_exchange([M.chem, M.owner], g, self.comm, self._plans)
        if A.N > 0:
            m, a, u = M.c_struct(), self._struct(A), action.c_struct()
            if second:
                lib.check(lib.lib.die_agent_dead_slots(C.byref(m), C.byref(a), C.byref(u), C.byref(d), ws, wsn, sp),
                          'die_agent_dead_slots')
            lib.check(lib.lib.die_step_reduce_ex(C.byref(a), _ptr(result), ws, wsn, int(second), A.N, sp), 'die_step_reduce_ex')
        else:
            result.zero_()
        m = M.c_struct()
        lib.check(lib.lib.die_medium_deposit_feed_diffuse_tile(C.byref(m), C.byref(d), 0, sp),
                  'die_medium_deposit_feed_diffuse_tile')
        M.swap_chem()
        self._food_flow()
        self._steps += 1
        if self._steps % self.migrate_every == 0:
            self._hand_over_strays(action)
        if self._sort_every > 0 and self._steps % self._sort_every == 0 and A.N > 1:
            self.sort_agents()
        if self._overlap:
            self._start_chem_halo()
        self.last_result = result
        return self._get_current_obs, result

    def _hand_over_strays(self, action):
        """Migration round of the guard-band mode: agents standing outside the interior move to the owner."""
        from .device_array import _ptr, stream_ptr
        lib, A, g = self._lib, self.agents, self.geo
        n = A.N
        if n:
            zero = torch.zeros((3, max(n, 1)), dtype=torch.float32, device=self.device)
            u = lib.Action(n, _ptr(zero[0]), _ptr(zero[1]), _ptr(zero[2]))
            m, a, d = self.medium.c_struct(), self._struct(A), self._c_dynamics()
            lib.check(lib.lib.die_agent_move(C.byref(m), C.byref(a), C.byref(u), C.byref(d), g.Wi, g.Hi, g.Py,
                                             _ptr(self._tile_of), stream_ptr(self.device)), 'die_agent_move')
            # every stray must still be inside the guard band, or its claims went to a clamped (wrong) cell
            lv = torch.nonzero(self._tile_of[:n] != self.comm.rank, as_tuple=False).squeeze(1)
            if lv.numel():
                far = self._stray_distance(lv)
                if far > self.band:
                    raise RuntimeError(f'rank {self.comm.rank}: an agent strayed {far} cells outside its tile, the guard band '
                                       f'is {self.band}: lower migrate_every or raise max_step_cells')
        self._migrate(action)
        halo_exchange([self.medium.food], self.geo, self.comm, self._plans)          # belt and braces: re-seat the food halo

    def _stray_distance(self, idx: torch.Tensor) -> int:
        g, A = self.geo, self.agents
        def cells(q, n_world):
            return (((q[idx].to(torch.int64) & 0xFFFFFFFF) * (n_world - 1) + (1 << 31)) >> 32)
        def outside(c, lo, size, world):
            l = (c - lo) % world                                       # offset from the interior's origin, periodic
            d = torch.minimum(l - (size - 1), world - l)               # … to the far edge going on, or back round the seam
            return torch.where(l < size, torch.zeros_like(l), d)
        dx = outside(cells(A.x, g.gW), g.x0, g.Wi, g.gW)
        dy = outside(cells(A.y, g.gH), g.y0, g.Hi, g.gH)
        return int(torch.maximum(dx, dy).max().item())

    def _check_reach(self, action):
        g = self.geo
        if self.ghosts:
            reach = int(np.ceil(abs(float(action.g_struct.sense_offset)) * (max(g.gW, g.gH) - 1)))
            if reach > self._probe_reach:
                raise ValueError(f'the agent senses {reach} cells ahead, DistEnv was built for probe_reach={self._probe_reach}')
            return
        reach = int(np.ceil(abs(float(action.g_struct.sense_offset)) * (max(g.gW, g.gH) - 1))) + 1 + self.R + self.band
        if reach > g.h:
            raise ValueError(f'probe reach + guard band = {reach} cells exceeds the halo {g.h}: build DistEnv with a larger '
                             f'probe_reach')

    # ------------------------------------------------------------------ migrate-every-step mode
    def _step_migrate_each(self, action):
        from .device_array import _ptr, stream_ptr
        lib, A, g = self._lib, self.agents, self.geo
        from .device_array import PendingAction
        sp = stream_ptr(self.device)
        d = self._c_dynamics()
        m, a = self.medium.c_struct(), self._struct(A)
        if isinstance(action, PendingAction) and action.pending and action.agents is A and action.medium is self.medium:
            self._check_reach(action)
            u = action.raw_struct()                     # forward runs fused with the move half
            lib.check(lib.lib.die_forward_move(C.byref(m), C.byref(a), C.byref(action.g_struct), C.byref(u), C.byref(d),
                                               g.Wi, g.Hi, g.Py, _ptr(self._tile_of), sp), 'die_forward_move')
            action.agent._forward_consumed(action)
        else:
            action.N = A.N
            u = action.c_struct()
            lib.check(lib.lib.die_agent_move(C.byref(m), C.byref(a), C.byref(u), C.byref(d), g.Wi, g.Hi, g.Py,
                                             _ptr(self._tile_of), sp), 'die_agent_move')
        self._migrate(action)
        self.medium.next_epoch()
        result = torch.empty(2, dtype=torch.float64, device=self.device)
        if A.N > 0:
            m, a, u = self.medium.c_struct(), self._struct(A), action.c_struct()
            ws, wsn = _ptr(self._workspace), self._workspace.numel()
            lib.check(lib.lib.die_agent_claim_feed(C.byref(m), C.byref(a), C.byref(u), C.byref(d), ws, wsn, sp),
                      'die_agent_claim_feed')
            second = not self._all_alive
            if second:                      # dead slots / lifecycle: needs the claims, not the field
                lib.check(lib.lib.die_agent_dead_slots(C.byref(m), C.byref(a), C.byref(u), C.byref(d), ws, wsn, sp),
                          'die_agent_dead_slots')
            lib.check(lib.lib.die_step_reduce_ex(C.byref(a), _ptr(result), ws, wsn, int(second), A.N, sp),
                      'die_step_reduce_ex')
        else:
            result.zero_()
        if self.dynamics.agents_die and self.comm.size > 1:
            # the lifecycle zeroes every channel of a starved slot: it now stands on world cell (0, 0), which
            # belongs to rank 0 — send it there before the next forward() senses from that position
            n = A.N
            self._tile_of[:n] = self.comm.rank
            if n:
                gone = (A.x[:n] == 0) & (A.y[:n] == 0)
                self._tile_of[:n][gone] = 0
            self._migrate(action)
        # the field sweep applies deposits on load, so it needs chem AND claims of the halo
        M = self.medium
        if self._chem_halo_in_flight:
            if self._comm_stream is not None:
                torch.cuda.current_stream(self.device).wait_stream(self._comm_stream)
            self._chem_halo_in_flight = False
            halo_exchange([M.owner], g, self.comm, self._plans)
        else:
            halo_exchange([M.chem, M.owner], g, self.comm, self._plans)
        m = M.c_struct()
        lib.check(lib.lib.die_medium_deposit_feed_diffuse_tile(C.byref(m), C.byref(d), g.h, sp),
                  'die_medium_deposit_feed_diffuse_tile')
        M.swap_chem()
        self._food_flow()
        self._steps += 1
        if self._sort_every > 0 and self._steps % self._sort_every == 0 and A.N > 1:
            self.sort_agents()
        if self._overlap:
            self._start_chem_halo()
        self.last_result = result
        return self._get_current_obs, result

    def sort_agents(self):
        """Bucket-sort the local agent arrays (die_agents_sort), attached Agent state included."""
        from .device_array import _ptr, stream_ptr
        self._pic_void()                   # (a refresh that waits for the next step then goes agent by agent: the tile order is gone)
        lib, A = self._lib, self.agents
        cap = self.capacity
        if self._shadow is None:
            self._shadow = [torch.zeros_like(t) for t in (A.x, A.y, A.alive, A.agent_food, A.slot)]
            n = lib.lib.die_sort_workspace_bytes(self.medium.W, self.medium.H, cap)
            self._sort_ws = torch.empty(n, dtype=torch.uint8, device=self.device)
        owners, tensors = [], []
        for obj in A.attached():
            ts = obj._die_state_tensors(A)
            if ts and len(tensors) + len(ts) <= 4:
                owners.append((obj, len(ts)))
                tensors += ts
        outs = [torch.zeros_like(t) for t in tensors]
        ein = (C.c_void_p * max(len(tensors), 1))(*[t.data_ptr() for t in tensors])
        eout = (C.c_void_p * max(len(outs), 1))(*[t.data_ptr() for t in outs])
        ox, oy, oalive, ofood, oslot = self._shadow
        m, a_in = self.medium.c_struct(), self._struct(A)
        a_out = lib.Agents(A.N, _ptr(ox), _ptr(oy), _ptr(oalive), _ptr(ofood), _ptr(oslot))
        lib.check(lib.lib.die_agents_sort(C.byref(m), C.byref(a_in), C.byref(a_out), len(tensors), ein, eout,
                                          _ptr(self._sort_ws), self._sort_ws.numel(), stream_ptr(self.device)),
                  'die_agents_sort')
        self._shadow = [A.x, A.y, A.alive, A.agent_food, A.slot]
        A.x, A.y, A.alive, A.agent_food, A.slot = ox, oy, oalive, ofood, oslot
        k = 0
        for obj, n in owners:
            obj._die_state_permuted(outs[k:k + n], A.slot)
            k += n

    def _zero_second_partials(self):
        """die_step_reduce sums both partial arrays; the dead-slot pass did not run, clear its slots."""
        n = 8192 * 8
        self._workspace[n:3 * n].zero_()

    def read_result(self, result: torch.Tensor) -> Tuple[float, int]:
        """World-wide (reward, num_agents): scalar all-reduce of the local results.  Ghost mode also sums the owned
        agents counted at the last refresh: every world agent must have exactly one owner."""
        host = result.cpu()
        if self._pic is not None and self._pic.steps_since_check:      # the tile-binned step's sticky error word: the stream is idle now,
            self._pic.check()                                          # one more 4-byte read (a loop that only reads results never saw it)
        owned = float(self._owned) if (self.ghosts and self._owned is not None) else -1.0
        trip = torch.tensor([float(host[0]), float(int(host.view(torch.int64)[1])), owned], dtype=torch.float64)
        if not self.comm.stage_cpu:
            trip = trip.to(self.device)
        tot = self.comm.all_reduce_sum(trip).cpu()
        if owned >= 0 and int(round(float(tot[2]))) != self.world_agents:
            raise RuntimeError(f'ghost-agent mode: {int(round(float(tot[2])))} agents were owned at the last refresh, the world has '
                               f'{self.world_agents}: an agent moved further than max_step_cells per step or sensed further than '
                               f'probe_reach (ghosts diverged from their originals)')
        return float(tot[0]), int(round(float(tot[1])))

    # ------------------------------------------------------------------ gathering (tests, checkpoints)
    def gather_world(self):
        """Rank 0 gets (medium (3, gW, gH), agents (4, N_world)) in slot order; others get None."""
        self.flush_refresh()
        g, A = self.geo, self.agents
        ri, ci = g.interior()
        tile = np.stack([self.medium.sel(c)[ri, ci].to(torch.float64).cpu().numpy() for c in self.medium.channels])
        n = A.N
        from .device_array import Q32
        own = self.owned_mask().cpu().numpy()
        rows = np.stack([(A.x[:n].to(torch.int64) & 0xFFFFFFFF).to(torch.float64).cpu().numpy() / Q32,
                         (A.y[:n].to(torch.int64) & 0xFFFFFFFF).to(torch.float64).cpu().numpy() / Q32,
                         A.alive[:n].to(torch.float64).cpu().numpy(), A.agent_food[:n].to(torch.float64).cpu().numpy()])[:, own]
        slots = A.slot[:n].cpu().numpy().astype(np.int64)[own]
        payload = (g.rank, tile, rows, slots)
        out = [None] * self.comm.size
        dist.all_gather_object(out, payload, group=self.comm.group)
        if self.comm.rank != 0:
            return None
        medium = np.zeros((3, g.gW, g.gH))
        agents = np.zeros((4, self.world_agents))
        for r, t, rw, sl in out:
            tg = TileGeometry((g.gW, g.gH), (g.Px, g.Py), r, (g.hx, g.hy))
            medium[:, tg.x0:tg.x0 + tg.Wi, tg.y0:tg.y0 + tg.Hi] = t
            agents[:, sl] = rw
        return medium, agents
