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
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


class TileGeometry:
    """Pure index arithmetic of the decomposition (no device, no communication)."""

    def __init__(self, world: Tuple[int, int], grid: Tuple[int, int], rank: int, halo: int):
        gW, gH = int(world[0]), int(world[1])
        Px, Py = int(grid[0]), int(grid[1])
        if gW % Px or gH % Py:
            raise ValueError(f'world {world} is not divisible by the rank grid {grid}')
        self.gW, self.gH, self.Px, self.Py, self.rank, self.h = gW, gH, Px, Py, int(rank), int(halo)
        self.px, self.py = divmod(self.rank, Py)
        self.Wi, self.Hi = gW // Px, gH // Py
        if self.h > self.Wi or self.h > self.Hi:
            raise ValueError(f'halo {halo} wider than the tile {(self.Wi, self.Hi)}')
        self.x0, self.y0 = self.px * self.Wi, self.py * self.Hi
        self.W, self.H = self.Wi + 2 * self.h, self.Hi + 2 * self.h
        self.ox, self.oy = self.x0 - self.h, self.y0 - self.h

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
        return slice(self.h, self.h + self.Wi), slice(self.h, self.h + self.Hi)

    def halo_plan(self):
        """[(phase, send_to, send_view, recv_from, recv_view)] as index tuples into a padded plane.
        Phase 0 moves columns between y-neighbours (interior rows only), phase 1 moves full-width
        rows between x-neighbours, so corner cells arrive in two hops."""
        h, Wi, Hi, W, H = self.h, self.Wi, self.Hi, self.W, self.H
        rows = slice(h, h + Wi)
        allc = slice(0, H)
        return [
            (0, self.neighbour(0, -1), (rows, slice(h, 2 * h)), self.neighbour(0, +1), (rows, slice(h + Hi, H))),
            (0, self.neighbour(0, +1), (rows, slice(Hi, Hi + h)), self.neighbour(0, -1), (rows, slice(0, h))),
            (1, self.neighbour(-1, 0), (slice(h, 2 * h), allc), self.neighbour(+1, 0), (slice(h + Wi, W), allc)),
            (1, self.neighbour(+1, 0), (slice(Wi, Wi + h), allc), self.neighbour(-1, 0), (slice(0, h), allc)),
        ]


class Comm:
    """Point-to-point transport over torch.distributed; `stage_cpu` routes device tensors through
    host memory (gloo).  Messages between one pair of ranks are matched in issue order."""

    def __init__(self, group=None, stage_cpu: Optional[bool] = None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        backend = dist.get_backend(group)
        self.stage_cpu = (backend == 'gloo') if stage_cpu is None else stage_cpu

    def exchange(self, sends: Sequence[Tuple[int, torch.Tensor]], recvs: Sequence[Tuple[int, torch.Tensor]]):
        """Send every (peer, tensor) and fill every (peer, buffer).  Self-messages are copied."""
        self_send = [t for p, t in sends if p == self.rank]
        self_recv = [t for p, t in recvs if p == self.rank]
        assert len(self_send) == len(self_recv)
        for s, r in zip(self_send, self_recv):
            r.copy_(s.reshape(r.shape))
        ops, staged = [], []
        for p, t in sends:
            if p == self.rank:
                continue
            buf = t.contiguous()
            if self.stage_cpu and buf.is_cuda:
                buf = buf.cpu()
            ops.append(dist.P2POp(dist.isend, buf, p, self.group))
        for p, t in recvs:
            if p == self.rank:
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


def halo_exchange(planes, geo: TileGeometry, comm: Comm):
    """Fill the halo ring of one or several padded (W, H) planes from the periodic neighbours; all
    planes travel together in two batches (phase 0: columns, phase 1: full-width rows).  On a GPU the
    strips of all planes are packed into ONE message per neighbour by die_rects_pack / _unpack."""
    if isinstance(planes, torch.Tensor):
        planes = [planes]
    plan = geo.halo_plan()
    on_gpu = all(p.is_cuda for p in planes)
    if on_gpu:
        from . import _lib
        from .device_array import stream_ptr
    for phase in (0, 1):
        entries = [e for e in plan if e[0] == phase]          # [to −side, to +side]
        if not on_gpu:
            sends, recvs = [], []
            for plane in planes:
                for ph, to, sview, frm, rview in entries:
                    sends.append((to, plane[sview]))
                    recvs.append((frm, plane[rview]))
            comm.exchange(sends, recvs)
            continue
        # message layout: for each direction, the strips of every plane back to back (8-byte aligned)
        srects, rrects, offs, off = [], [], [], 0
        for ph, to, sview, frm, rview in entries:
            offs.append(off)
            for plane in planes:
                rows = sview[0].stop - sview[0].start
                cols = sview[1].stop - sview[1].start
                srects.append(_slice_rect(_lib, plane, sview, off))
                rrects.append(_slice_rect(_lib, plane, rview, off))
                off += (rows * cols * plane.element_size() + 7) & ~7
        offs.append(off)
        key = (planes[0].device, phase, off)
        if key not in _HALO_BUFFERS:
            _HALO_BUFFERS[key] = (torch.empty(off, dtype=torch.uint8, device=planes[0].device),
                                  torch.empty(off, dtype=torch.uint8, device=planes[0].device))
        sbuf, rbuf = _HALO_BUFFERS[key]
        sp = stream_ptr(planes[0].device)
        sarr = (_lib.Rect * len(srects))(*srects)
        _lib.check(_lib.lib.die_rects_pack(sarr, len(srects), C.c_void_p(sbuf.data_ptr()), sp), 'die_rects_pack')
        # "send to the −side, send to the +side" pairs with "recv from the +side, recv from the −side": the
        # message a rank sends to its −side neighbour is what that neighbour receives from its +side
        sends = [(entries[0][1], sbuf[offs[0]:offs[1]]), (entries[1][1], sbuf[offs[1]:offs[2]])]
        recvs = [(entries[0][3], rbuf[offs[0]:offs[1]]), (entries[1][3], rbuf[offs[1]:offs[2]])]
        comm.exchange(sends, recvs)
        rarr = (_lib.Rect * len(rrects))(*rrects)
        _lib.check(_lib.lib.die_rects_unpack(rarr, len(rrects), C.c_void_p(rbuf.data_ptr()), sp), 'die_rects_unpack')


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
                 seed: int = 0, init: bool = True, sort_every: int = 8, overlap: bool = True):
        from . import _lib
        from .data_init import DataInitializer
        from .device_array import DeviceAgents, DeviceMedium
        from .env import Dynamics
        self._lib = _lib
        self.dynamics = dynamics or Dynamics()
        self.comm = Comm(group)
        R = int(4.0 * float(self.dynamics.diffuse_sigma) + 0.5)
        halo = int(probe_reach) + 1 + R
        while (world[1] // grid[1] + 2 * halo) % 4:                    # die_diffuse_decay_tile needs H % 4 == 0
            halo += 1
        self.geo = TileGeometry(world, grid, self.comm.rank, halo)
        self.R = R
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        g = self.geo
        self.medium = DeviceMedium((g.W, g.H), self.device, field_dtype)
        self.medium.world = (g.gW, g.gH, g.ox, g.oy)
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
        if init:
            self._init_tile()

    # ------------------------------------------------------------------ construction
    def _alloc_agents(self, n_local: int):
        from .device_array import DeviceAgents
        from .data_init import DataInitializer
        cap = self.capacity or max(int(n_local * 1.5) + 1024, 4096)
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
        occ_int[g.h:g.h + g.Wi, g.h:g.h + g.Hi] = occ[g.h:g.h + g.Wi, g.h:g.h + g.Hi]
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
                            int(d.food_infinite), int(d.agents_die), int(not self._all_alive))

    def _per_agent_tensors(self, action) -> List[torch.Tensor]:
        """Every per-agent array that must travel with a migrating agent (4-byte views)."""
        A = self.agents
        ts = [A.x, A.y, A.slot, A.agent_food, action.data[0], action.data[1], action.data[2]]
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
            halo_exchange([self.medium.chem], self.geo, self.comm)
        else:
            self._comm_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._comm_stream):
                halo_exchange([self.medium.chem], self.geo, self.comm)
        self._chem_halo_in_flight = True

    def step(self, action):
        """One decomposed env step.  Returns (obs, result_tensor): result is the LOCAL
        die_step_result; `read_result` all-reduces it."""
        from .device_array import _ptr, stream_ptr
        lib, A, g = self._lib, self.agents, self.geo
        from .device_array import PendingAction
        sp = stream_ptr(self.device)
        d = self._c_dynamics()
        m, a = self.medium.c_struct(), self._struct(A)
        if isinstance(action, PendingAction) and action.pending and action.agents is A and action.medium is self.medium:
            # the halo must cover the probe: sense_offset is in units of the world, i.e. up to this many cells
            reach = int(np.ceil(abs(float(action.g_struct.sense_offset)) * (max(g.gW, g.gH) - 1))) + 1 + self.R
            if reach > g.h:
                raise ValueError(f'probe reach {reach} cells exceeds the halo {g.h}: build DistEnv with probe_reach >= '
                                 f'{reach - 1 - self.R}')
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
            halo_exchange([M.owner], g, self.comm)
        else:
            halo_exchange([M.chem, M.owner], g, self.comm)
        m = M.c_struct()
        lib.check(lib.lib.die_medium_deposit_feed_diffuse_tile(C.byref(m), C.byref(d), g.h, sp),
                  'die_medium_deposit_feed_diffuse_tile')
        M.swap_chem()
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
        """World-wide (reward, num_agents): scalar all-reduce of the local results."""
        host = result.cpu()
        pair = torch.tensor([float(host[0]), float(int(host.view(torch.int64)[1]))], dtype=torch.float64)
        if not self.comm.stage_cpu:
            pair = pair.to(self.device)
        tot = self.comm.all_reduce_sum(pair).cpu()
        return float(tot[0]), int(round(float(tot[1])))

    # ------------------------------------------------------------------ gathering (tests, checkpoints)
    def gather_world(self):
        """Rank 0 gets (medium (3, gW, gH), agents (4, N_world)) in slot order; others get None."""
        g, A = self.geo, self.agents
        ri, ci = g.interior()
        tile = np.stack([self.medium.sel(c)[ri, ci].to(torch.float64).cpu().numpy() for c in self.medium.channels])
        n = A.N
        from .device_array import Q32
        rows = np.stack([(A.x[:n].to(torch.int64) & 0xFFFFFFFF).to(torch.float64).cpu().numpy() / Q32,
                         (A.y[:n].to(torch.int64) & 0xFFFFFFFF).to(torch.float64).cpu().numpy() / Q32,
                         A.alive[:n].to(torch.float64).cpu().numpy(), A.agent_food[:n].to(torch.float64).cpu().numpy()])
        slots = A.slot[:n].cpu().numpy().astype(np.int64)
        payload = (g.rank, tile, rows, slots)
        out = [None] * self.comm.size
        dist.all_gather_object(out, payload, group=self.comm.group)
        if self.comm.rank != 0:
            return None
        medium = np.zeros((3, g.gW, g.gH))
        agents = np.zeros((4, self.world_agents))
        for r, t, rw, sl in out:
            tg = TileGeometry((g.gW, g.gH), (g.Px, g.Py), r, g.h)
            medium[:, tg.x0:tg.x0 + tg.Wi, tg.y0:tg.y0 + tg.Hi] = t
            agents[:, sl] = rw
        return medium, agents
