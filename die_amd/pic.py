"""Host side of the tile-binned step (csrc/die_pic.hip, include/die_hip.h `die_pic`): buffers of the two agent layouts,
(re)binning, and the per-step swap of the array handles.  No reference counterpart: the reference keeps agents in slot
order (core/data_init.py:133-150); here the order of the arrays is free (results are keyed by slot id) and this path
keeps it exactly tile-sorted so that `Env.step` needs no claim plane."""
import ctypes as C
import os
import weakref
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from .device_array import _ptr, stream_ptr

TILE_SHAPES = ((6, 6), (5, 6), (4, 5))        # log2 (rows, columns), in order of preference; (5, 7) by request only


def pick_tile(W: int, H: int, reach_cells: float, shapes=TILE_SHAPES) -> Optional[Tuple[int, int]]:
    """Largest compiled tile shape (of `shapes`) that cuts the world into at least 3×3 whole tiles and is wider than a step."""
    for xs, ys in shapes:
        TX, TY = 1 << xs, 1 << ys
        if W % TX == 0 and H % TY == 0 and W // TX >= 3 and H // TY >= 3 and reach_cells <= min(TX, TY) - 1:
            return xs, ys
    return None


def step_scale(agent) -> float:
    """`scale` times the per-axis bound of the vector the action is `scale` times (die_pic_step_bound: 1 without momentum), as the
    library computes it in float32: what the tile rules take for `scale`.  (Kept per agent: it is asked for at every step.)"""
    key = (agent._scale, agent._inertia, agent._noise_scale)
    hit = agent.__dict__.get('_pic_step_scale')
    if hit is not None and hit[0] == key:
        return hit[1]
    import numpy as np
    bound = _lib.lib.die_pic_step_bound(float(agent._inertia), float(agent._noise_scale))
    value = float(np.float32(agent._scale) * np.float32(bound))
    agent.__dict__['_pic_step_scale'] = (key, value)
    return value


def lazy_ok(agent) -> bool:
    """Does the field kernel leave the next step's turn bits behind for this agent?  (A PhysarumAgent drawing from Philox.)"""
    return agent._kind == _lib.DIE_AGENT_PHYSARUM and agent._turn_sign is None


class PicState:
    """`env`: a die_amd.Env, or a die_amd.dist.DistEnv in ghost-agent mode (its planes are a padded tile of the world, its agent
    arrays hold `capacity` entries of which the first `env.agents.N` are agents: the buffers here are sized by the arrays,
    the count is read at every call)."""

    def __init__(self, env, tile: Tuple[int, int]):
        self.xs, self.ys = tile
        W, H = env.medium.W, env.medium.H
        self.NT = int(_lib.lib.die_pic_tiles(W, H, self.xs, self.ys))
        if self.NT <= 0:
            raise ValueError(f'tile shape 2^{tile} is not available')
        dev, N = env.device, int(env.agents.x.numel())
        self.cap = N
        self.meta = [torch.zeros((4, self.NT), dtype=torch.int32, device=dev) for _ in range(2)]     # off, n, s, inc
        self.dep = torch.empty(N, dtype=torch.float32, device=dev)
        # two-launch form (include/die_hip.h `die_pic.rim`): per tile a short list from the agent kernel to the field kernels
        self.fused = bool(getattr(env, '_pic_fused', True))
        cap = int(_lib.lib.die_pic_rim_cap(self.xs, self.ys))
        self.rim = torch.zeros((self.NT * cap, 4), dtype=torch.int32, device=dev) if self.fused else None
        self.rim_code = torch.zeros(self.NT * cap, dtype=torch.uint8, device=dev) if self.fused else None
        self.rim_cnt = torch.zeros(self.NT, dtype=torch.int32, device=dev) if self.fused else None
        self._plane_shape = (W, H)
        self._world_shape = (env.medium.world[0], env.medium.world[1]) if env.medium.world is not None else (W, H)
        self._n_agents = int(env.agents.N)
        # three-launch form only (else allocated when a step turns out to need it: a long step on small tiles)
        self._dep_plane = None if self.fused else torch.empty((W, H), dtype=torch.float32, device=dev)
        # PhysarumAgent's random turn bits, one per slot id (include/die_hip.h `die_pic.turn_bits`); `_turn_for`: the (seed, step)
        # whose bits the table holds — a step's field kernel leaves the next step's behind
        self.turn_slots = max(N, int(getattr(env, 'world_agents', 0) or 0))
        self.turn_bits = torch.zeros(4 * ((self.turn_slots + 127) // 128), dtype=torch.int32, device=dev)
        self._turn_for = None
        # the order table of the two-launch form's workgroups (include/die_hip.h `die_pic.order`): crowded tiles first inside every XCD
        # band, rebuilt by the library every 32nd step; an undivided world whose tiles per row divide by 8 (else the band mapping)
        nty = H >> self.ys
        self.order = torch.zeros(self.NT, dtype=torch.int16, device=dev) \
            if (self.fused and env.medium.world is None and nty % 8 == 0 and self.NT <= 65536 and os.environ.get('DIE_PIC_ORDER', '1') != '0') else None
        self._order_ready = False
        # the reference's default slot layout (max_agents = W·H: most slots never lived): the alive agents in the tiles' segments,
        # the dead slots behind them (include/die_hip.h `die_pic.n_alive`); `occ`: this step's occupancy map (a byte per cell) for their feeding
        self.n_alive = int(getattr(env, '_pic_n_alive', 0) or 0)
        self.occ = torch.zeros(W * H, dtype=torch.uint8, device=dev) if 0 < self.n_alive < N else None
        self.part = torch.zeros(2 * self.NT, dtype=torch.int64, device=dev)       # reward partials | owned agents (decomposed tiles)
        self.error = torch.zeros(2 + 40 * self.NT, dtype=torch.int32, device=dev)      # [0]: error word; the rest: diagnostic builds (-DPIC_STAMPS)
        i32 = lambda: torch.empty(N, dtype=torch.int32, device=dev)
        self.spare = [i32(), i32(), torch.empty(N, dtype=torch.float32, device=dev), i32(), i32()]     # x, y, agent_food, heading hi / lo
        self.spare_pg = None         # GradientAgent with inertia: the (2, N) _prev_grad array of the layout that is not current (die_pic.prev_grad)
        self.k1_threads = int(os.environ.get('DIE_PIC_THREADS', '0'))   # die_pic.k1_threads: 0 = library default, > 0: workgroup size of the agent kernel
        self.cur = 0                 # layout index that holds the agents
        self.held = None             # (x, y, agent_food, slot, heading hi, lo) tensors of layout[cur] — identity = validity
        self._structs = {}           # die_pic structs by the device addresses of the two array sets (_struct)
        self._two_key = self._two = None
        self.agent = None
        self.agent_for_out = None    # the agent whose state the next output tensors carry (bin / step set it)
        self.steps_since_check = 0   # binned steps whose error word nobody has read yet
        self.lazy_actions = True     # PhysarumAgent: the step keeps the action in registers, PendingAction re-derives it on demand
        self._lazy_ref = None        # weak reference to the last such action (it must be filled in before its inputs change)

    # ------------------------------------------------------------------
    def _layout(self, tensors, meta) -> _lib.PicLayout:
        x, y, af, slot, hh, hl = tensors[:6]
        return _lib.PicLayout(_ptr(x), _ptr(y), _ptr(af), _ptr(slot), _ptr(hh), _ptr(hl), _ptr(meta[0]), _ptr(meta[1]), _ptr(meta[2]), _ptr(meta[3]))

    def _struct(self, cur_tensors, other_tensors, stages: int = 0, status_out=None) -> _lib.Pic:
        """The die_pic of a call: layout[cur] = `cur_tensors`, layout[1 - cur] = `other_tensors`.  The arrays ping-pong between
        two fixed sets, so the struct of a (set, set) pair is built once and kept — keyed by the device addresses, not by the
        tensor objects — and only what changes from step to step is written into it: the two slot arrays (never reused:
        actions may still refer to them), the agent count, stages, status_out.  (Host time per step matters to
        Env(sync=True), where nothing hides it: 6.3 → 1.9 µs.)"""
        ct, ot, cur = cur_tensors, other_tensors, self.cur
        key = (cur, ct[0].data_ptr(), ct[1].data_ptr(), ct[2].data_ptr(), ct[4].data_ptr(), ct[5].data_ptr(),
               ot[0].data_ptr(), ot[1].data_ptr(), ot[2].data_ptr(), ot[4].data_ptr(), ot[5].data_ptr(),
               0 if self._dep_plane is None else self._dep_plane.data_ptr(),
               0 if ct[6] is None else ct[6].data_ptr(), 0 if ot[6] is None else ot[6].data_ptr())
        p = self._structs.get(key)
        if p is None:
            if len(self._structs) >= 8:
                self._structs.clear()
            lay = [None, None]
            lay[cur] = self._layout(ct, self.meta[cur])
            lay[1 - cur] = self._layout(ot, self.meta[1 - cur])
            p = self._structs[key] = _lib.Pic(self.xs, self.ys, self._n_agents, (_lib.PicLayout * 2)(*lay), _ptr(self.dep), _ptr(self._dep_plane),
                                              _ptr(self.part), _ptr(self.error), self.k1_threads, stages, _ptr(self.rim), _ptr(self.rim_code),
                                              _ptr(self.rim_cnt), status_out, _ptr(self.turn_bits), self.turn_slots, 0, 0, _ptr(self.order), 0, 0, 0, 0, 0, 0,
                                              self.n_alive if self.occ is not None else 0, _ptr(self.occ))
            for lay_i, t in ((cur, ct[6]), (1 - cur, ot[6])):
                for axis in (0, 1):
                    p.prev_grad[lay_i][axis] = None if t is None else t[axis].data_ptr()
        L = p.layout
        L[cur].slot, L[1 - cur].slot = ct[3].data_ptr(), ot[3].data_ptr()
        p.N, p.k1_threads, p.stages, p.status_out = self._n_agents, self.k1_threads, stages, status_out
        return p

    def two_launch(self, env, agent) -> bool:
        """Does die_pic_forward_env_step take the two-launch form for this agent?  (The library decides by the same rule;
        here it only settles whether the deposit plane of the three-launch form has to exist.)"""
        if not self.fused:
            return False
        key = (agent._scale, env.dynamics.diffuse_sigma, agent._inertia, agent._noise_scale)
        if key == self._two_key:
            return self._two
        self._two_key, self._two = key, self._two_launch(env, agent)
        return self._two

    def _two_launch(self, env, agent) -> bool:
        """The library's rule (die_pic_two_launch): nothing is restated here."""
        W, H = self._world_shape
        mode = _lib.DIFFUSE_MODES[env.dynamics.diffuse_mode]
        return _lib.lib.die_pic_two_launch(max(W, H), self.xs, self.ys, step_scale(agent), float(env.dynamics.diffuse_sigma), mode) == 1

    def is_current(self, env, agent) -> bool:
        A, h = env.agents, self.held
        return h is not None and self.agent is agent and A.x is h[0] and A.y is h[1] and A.agent_food is h[2] and A.slot is h[3] and \
            agent._hd_hi is h[4] and agent._hd_lo is h[5] and agent._order is A.slot and getattr(agent, '_prev_grad', None) is h[6]

    def _adopt(self, env, agent, new):
        """The agents now live in `new` = (x, y, agent_food, slot, heading hi, lo): hand the arrays to their owners and keep the
        old ones as the next step's output buffers (the slot array is never reused: actions may still refer to it)."""
        A = env.agents
        self.spare = [A.x, A.y, A.agent_food, agent._hd_hi, agent._hd_lo]
        A.x, A.y, A.agent_food, A.slot = new[0], new[1], new[2], new[3]
        agent._hd_hi, agent._hd_lo = new[4], new[5]
        if new[6] is not None:
            self.spare_pg, agent._prev_grad = agent._prev_grad, new[6]
        agent._order = A.slot
        self.held, self.agent = tuple(new), agent

    def _out_tensors(self, env):
        slot = torch.empty(self.cap, dtype=torch.int32, device=env.device)
        pg = None
        if getattr(self.agent_for_out, '_prev_grad', None) is not None:
            if self.spare_pg is None or self.spare_pg.shape != (2, self.cap):
                self.spare_pg = torch.empty((2, self.cap), dtype=torch.float32, device=env.device)
            pg = self.spare_pg
        return (self.spare[0], self.spare[1], self.spare[2], slot, self.spare[3], self.spare[4], pg)

    def bin(self, env, agent):
        """Agents in any order → layout[1 - cur]; both layouts' per-tile words are reset.  (The sticky error word is read first
        — a re-bin would otherwise paper over a broken layout; the library never clears it.)"""
        self.flush_lazy()
        if self.steps_since_check:
            self.check()
        A = env.agents
        self._n_agents = int(A.N)
        self.agent_for_out = agent
        pg = getattr(agent, '_prev_grad', None)
        if pg is not None and (pg.dtype != torch.float32 or not pg.is_contiguous() or pg.shape != (2, self.cap)):
            pg = agent._prev_grad = pg.to(torch.float32).contiguous()
        if pg is not None and self.spare_pg is pg:
            self.spare_pg = None                 # (never bin an array into itself)
        out = self._out_tensors(env)
        cur_t = (A.x, A.y, A.agent_food, A.slot if A.slot is not None else out[3], agent._hd_hi, agent._hd_lo, pg)
        p = self._struct(cur_t, out)
        m, a = env.medium.c_struct(need_owner=False), A.c_struct()
        if pg is None:
            _lib.check(_lib.lib.die_pic_bin(C.byref(m), C.byref(a), _ptr(agent._hd_hi), _ptr(agent._hd_lo), C.byref(p), 1 - self.cur,
                                            stream_ptr(env.device)), 'die_pic_bin')
        else:
            _lib.check(_lib.lib.die_pic_bin_momentum(C.byref(m), C.byref(a), _ptr(agent._hd_hi), _ptr(agent._hd_lo), _ptr(pg[0]), _ptr(pg[1]),
                                                     C.byref(p), 1 - self.cur, stream_ptr(env.device)), 'die_pic_bin_momentum')
        self.cur = 1 - self.cur
        self._adopt(env, agent, out)
        if self.occ is not None:                 # the array order is now: alive agents (tile by tile), then the dead slots
            A.alive = torch.cat([torch.ones(self.n_alive, dtype=A.alive.dtype, device=env.device),
                                 torch.zeros(self._n_agents - self.n_alive, dtype=A.alive.dtype, device=env.device)])

    def flush_lazy(self):
        """The action of the previous step, if somebody still holds it without having read it: fill it in now (the next
        step overwrites the deposit array it is derived from)."""
        ref, self._lazy_ref = self._lazy_ref, None
        act = ref() if ref is not None else None
        if act is not None:
            act.ensure()

    def _rebuilder(self, env, agent, out):
        """die_pic_action_physarum on what the step left in `out` (the layout it wrote) — see include/die_hip.h."""
        lay, dep, dev = self.cur, self.dep, env.device
        slot, hh, hl = out[3], out[4], out[5]
        agents = env.agents

        def rebuild(act):
            # the number of entries of the layout the step wrote: the agents' count when the action is read — a decomposed rank's
            # step that ran inside a ghost refresh (die_amd/dist.py) changed it, and the action was built for the count before
            # (found by scratch/fuzz_dist.py with a refresh at every step: "bad arrays")
            N = act.N = int(agents.N)
            L = [_lib.PicLayout(), _lib.PicLayout()]
            L[lay] = _lib.PicLayout(None, None, None, _ptr(slot), _ptr(hh), _ptr(hl), None, None, None, None)
            p = _lib.Pic(self.xs, self.ys, N, (_lib.PicLayout * 2)(*L), _ptr(dep), None, None, None, 0, 0, None, None, None, None, None, 0, 0, 0, None, 0, 0, 0, 0, 0, 0, 0, None)
            act.slot = slot                                    # the values come out in the order of the layout the step wrote
            u = act.raw_struct()
            _lib.check(_lib.lib.die_pic_action_physarum(C.byref(p), lay, C.byref(act.g_struct), C.byref(u), stream_ptr(dev)),
                       'die_pic_action_physarum')
        return rebuild

    def step(self, env, agent, action, dyn, result, events=None, status_out=None, plan=None):
        """One step.  `events`: torch.cuda.Event objects (one more than launches: 3 in the two-launch form, 4 in the
        three-launch form) — the launches are then issued one call each (die_pic.stages) with an event between them, so
        that bench.py times each kernel inside real steps.

        `plan` (two-launch form; a decomposed rank's step behind a ghost refresh, die_amd/dist.py): the step as a list of
        launches over subsets of the tiles, `(stages, (sub_mode, tx0, ty0, ntx, nty)[, halo_fresh])`, with callables in between (they run on
        the host between two launches: wait for the messages, unpack, second half of the merge).

        A normalised PhysarumAgent's action is a function of what the step leaves behind (heading', deposit array), so the
        step does not store it (30 MB and 7 % of the agent kernel at 4096²): the PendingAction gets a `_rebuild` hook and
        is filled in when somebody reads it — or, if it is still referenced then, before the next step."""
        self.flush_lazy()
        self._n_agents = int(env.agents.N)
        self.agent_for_out = agent
        if self._dep_plane is None and not self.two_launch(env, agent):
            self._dep_plane = torch.empty(self._plane_shape, dtype=torch.float32, device=env.device)
        out = self._out_tensors(env)
        lazy = self.lazy_actions and agent._kind == _lib.DIE_AGENT_PHYSARUM and hasattr(action, '_rebuild')
        m = env.medium.c_struct(need_owner=False)
        u = None if lazy else C.byref(action.raw_struct())
        two = self.two_launch(env, agent)
        g = action.g_struct
        turn_key = (int(g.seed), int(g.step))
        if plan is not None:
            if not two:
                raise ValueError('a step over subsets of the tiles exists in the two-launch form only')
            items = list(plan)
        elif events is None:
            items = [(0, None)]
        else:
            items = [(st, None) for st in ((1, 2) if two else (1, 2, 4))]
        i = 0
        self.step_out = out                                    # (for callables of `plan`: the layout this step writes)
        for item in items:
            if callable(item):
                item()
                m = env.medium.c_struct(need_owner=False)
                continue
            stages, sub = item[:2]
            p = self._struct(self.held, out, stages, status_out if two else None)
            p.halo_fresh = int(item[2]) if len(item) > 2 else 0      # (the agent kernel behind a refresh in place: die_pic_ghost_inplace)
            p.turn_ready = int(self._turn_for == turn_key)
            p.order_ready = int(self._order_ready)
            p.sub_mode, p.sub_tx0, p.sub_ty0, p.sub_ntx, p.sub_nty = sub if sub is not None else (0, 0, 0, 0, 0)
            if events is not None:
                events[i].record()
            i += 1
            rc = _lib.lib.die_pic_forward_env_step(C.byref(m), C.byref(p), self.cur, C.byref(action.g_struct), u, C.byref(dyn),
                                                   _ptr(result), stream_ptr(env.device))
            if rc != 0:
                return rc
            if (stages & 1 or stages == 0) and lazy_ok(agent):
                self._turn_for = turn_key                      # (the agent kernel's launch has filled the table if it was not ready)
            if (stages & 1 or stages == 0) and two and sub is None and self.order is not None:
                self._order_ready = True                       # (that launch has built the order table if it was not there)
        if events is not None:
            events[2 if two else 3].record()
        # (the field kernel of the two-launch form has filled the table for the next step of this seed)
        self._turn_for = (turn_key[0], (turn_key[1] + 1) & 0xFFFFFFFF) if two and lazy_ok(agent) else None
        self.cur = 1 - self.cur
        self.steps_since_check += 1
        self._adopt(env, agent, out)
        if lazy:
            action._rebuild = self._rebuilder(env, agent, out)
            self._lazy_ref = agent._lazy_action = weakref.ref(action)
        return 0

    def run(self, env, agent, action, dyn, results, n: int) -> int:
        """`n` whole steps in ONE library call (die_pic_run): the loop of examples/minimal_run.py:23-25 without the host between two
        steps.  `action`: the pending action of the first step (its g_struct carries seed and step counter); `results`: (n, 2) float64
        on the device, one die_step_result per step.  No action is handed out, so the two layouts' slot arrays simply ping-pong —
        in fresh arrays: actions handed out earlier keep referring to theirs."""
        self.flush_lazy()
        self._n_agents = int(env.agents.N)
        self.agent_for_out = agent
        out = self._out_tensors(env)
        held = self.held[:3] + (self.held[3].clone(),) + self.held[4:]
        g = action.g_struct
        turn_key = (int(g.seed), int(g.step))
        two = self.two_launch(env, agent)
        p = self._struct(held, out, 0, None)
        p.turn_ready = int(self._turn_for == turn_key)
        p.order_ready = int(self._order_ready)
        p.sub_mode, p.sub_tx0, p.sub_ty0, p.sub_ntx, p.sub_nty, p.halo_fresh = 0, 0, 0, 0, 0, 0
        m = env.medium.c_struct(need_owner=False)
        rc = _lib.lib.die_pic_run(C.byref(m), C.byref(p), self.cur, C.byref(g), C.byref(dyn), int(n), _ptr(results), stream_ptr(env.device))
        # (an error behind the first step — a failed launch — leaves `done` whole steps in the stream: the state THEY reach is adopted,
        # so that a caller who catches the exception holds a consistent Env)
        self.run_done = n = int(n) if rc == 0 else int(_lib.lib.die_pic_run_completed())
        if n == 0:
            return rc
        self._turn_for = (turn_key[0], (turn_key[1] + n) & 0xFFFFFFFF) if two and lazy_ok(agent) else None
        if two and self.order is not None:
            self._order_ready = True
        self.steps_since_check += n
        if n & 1:
            self.cur = 1 - self.cur
            self._adopt(env, agent, out)
        else:                                                  # the agents are back in the arrays they started from (new slot array)
            env.agents.slot = agent._order = held[3]
            self.held = held
        return rc

    def check(self):
        """After a synchronisation: did every agent stay within its tile's neighbourhood?"""
        e = int(self.error[0].item())
        self.raise_for(e)

    def raise_for(self, e: int):
        """`e`: the error word as read from the device (here, or with the step result: die_pic.status_out)."""
        self.steps_since_check = 0
        if e:
            self.error[:1].zero_()           # the library never clears it: the host does, once it has reported it
            raise RuntimeError(f'tile-binned step: bookkeeping error {e} (an agent moved further than one tile in a step)')
