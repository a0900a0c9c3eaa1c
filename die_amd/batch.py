"""Batched env replicas (BASELINE configs[4]): R worlds of one shape stepped by ONE launch pair per step
(`die_forward_env_step_batch`).  No reference counterpart — the reference steps one `Env` per Python call; a small grid
costs more in launches and host calls than in kernel time, and R replicas share those.

Replica r is exactly the stand-alone `Env(field_size, dynamics, seed=seed + r, max_agents='alive')` driven by
`PhysarumAgent(max_agents=K_r, seed=agent_seed + r, ...)`: same initial state, same Philox streams, same kernels'
arithmetic — bit for bit (tests/test_gpu_parity.py::test_batched_replicas_equal_stand_alone_runs)."""
import ctypes as C
import math
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .agent.gradient import join64, split64
from .device_array import Q32, _ptr, stream_ptr
from .env import BoundaryCondition, Dynamics, Env, linear_action_cost


class BatchedEnv:
    """R replicas.  Two regimes, chosen by world size (`per_replica`):
      * small worlds (below Env.PIC_MIN_CELLS cells): ONE launch pair for the whole batch (die_forward_env_step_batch) — a
        small grid is bound by launches and host calls, which the replicas then share;
      * large worlds (BASELINE configs[4]: 16384² fp16): a replica fills the GPU by itself, and what counts is the step
        each replica takes — the tile-binned step (no claim plane, no re-sort), which the one-launch-pair form does not
        have.  Each replica is then a stand-alone `Env` stepped on its OWN HIP stream (the latency-bound agent kernel of one
        replica overlaps the bandwidth-bound field kernel of another); `step` fans out and joins the streams.
    Either way replica r is the stand-alone run of seed + r, bit for bit."""

    def __init__(self, field_size: Tuple[int, int], dynamics: Optional[Dynamics] = None, *, replicas: int, seed: int = 0,
                 field_dtype: torch.dtype = torch.float32, device=None, per_replica: Optional[bool] = None):
        if not 1 <= replicas <= 64:
            raise ValueError('1..64 replicas')
        self.dynamics = dynamics or Dynamics()
        d = self.dynamics
        if d.agents_die or d.apply_sense_mask or d.diffuse_mode != 'wrap' or not isinstance(d.boundary, BoundaryCondition):
            raise NotImplementedError('batched replicas: wrap diffusion, no agents_die, no sense mask')
        self.R, self.seed = int(replicas), int(seed)
        self.W, self.H = int(field_size[0]), int(field_size[1])
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        self.dtype = field_dtype
        self.per_replica = (self.W * self.H >= Env.PIC_MIN_CELLS) if per_replica is None else bool(per_replica)
        if self.per_replica:
            self.envs = [Env(field_size, d, seed=self.seed + r, max_agents='alive', field_dtype=field_dtype, device=self.device,
                             sync=False) for r in range(self.R)]
            self.n = [e.agents.N for e in self.envs]
            self.Nmax = max(self.n)
            self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.R)]
            self._obs = [e._get_current_obs for e in self.envs]
            self._steps = 0
            return
        # every replica starts as the stand-alone Env with seed + r would; its state is copied into slice r
        envs = [Env(field_size, d, seed=self.seed + r, max_agents='alive', field_dtype=field_dtype, device=self.device, sort_every=0,
                    pic=False) for r in range(self.R)]
        self.n = [e.agents.N for e in envs]
        for r, e in enumerate(envs):
            if not e._all_alive:            # (only a world seeded with no agent at all: 'alive' keeps one dead placeholder slot)
                raise NotImplementedError(f'batched replicas: replica {r} (seed {self.seed + r}) has no alive agent; dead slots are '
                                          'not modelled by the batched step')
        self.Nmax = max(self.n)
        R, W, H, Nm, dev = self.R, self.W, self.H, self.Nmax, self.device
        self.owner = torch.zeros((R, W, H), dtype=torch.int64, device=dev)
        self.food = torch.zeros((R, W, H), dtype=field_dtype, device=dev)
        self.chem = torch.zeros((R, W, H), dtype=field_dtype, device=dev)
        self.chem_next = torch.empty((R, W, H), dtype=field_dtype, device=dev)
        self.x = torch.zeros((R, Nm), dtype=torch.int32, device=dev)
        self.y = torch.zeros((R, Nm), dtype=torch.int32, device=dev)
        self.alive = torch.zeros((R, Nm), dtype=torch.uint8, device=dev)
        self.agent_food = torch.zeros((R, Nm), dtype=torch.float32, device=dev)
        for r, e in enumerate(envs):
            self.owner[r].copy_(e.medium.owner); self.food[r].copy_(e.medium.food); self.chem[r].copy_(e.medium.chem)
            k = self.n[r]
            self.x[r, :k].copy_(e.agents.x); self.y[r, :k].copy_(e.agents.y)
            self.alive[r, :k].copy_(e.agents.alive); self.agent_food[r, :k].copy_(e.agents.agent_food)
        self.epoch = 1
        self._ws = torch.zeros(int(_lib.lib.die_batch_workspace_bytes(R)), dtype=torch.uint8, device=dev)
        self._steps = 0

    # ------------------------------------------------------------------
    def _structs(self):
        fdt = _lib.DIE_F32 if self.dtype == torch.float32 else _lib.DIE_F16
        m = _lib.Medium(self.W, self.H, fdt, self.epoch, _ptr(self.owner), _ptr(self.food), _ptr(self.chem), _ptr(self.chem_next),
                        0, 0, 0, 0, 0, 0, 0, 0, None)
        a = _lib.Agents(self.Nmax, _ptr(self.x), _ptr(self.y), _ptr(self.alive), _ptr(self.agent_food), None)
        d = self.dynamics
        boundary = _lib.DIE_BOUNDARY_WRAP if d.boundary == BoundaryCondition.wrap else _lib.DIE_BOUNDARY_LIMIT
        cost = _lib.DIE_COST_LINEAR if d.op_action_cost is linear_action_cost else _lib.DIE_COST_ZERO
        dyn = _lib.Dynamics(d.rate_feed, d.rate_decay_chem, d.diffuse_sigma, boundary, cost, 0.02, 0.01, int(d.food_infinite), 0, 0, 0, 0)
        b = _lib.Batch(self.R, 0, self.W * self.H, self.Nmax, 1, (C.c_int64 * 64)(*self.n))
        return m, a, dyn, b

    def check(self):
        """Synchronise; raise if a replica's tile-binned step reported a bookkeeping error since the last check (per-replica
        regime: every replica is an `Env` of its own; the one-launch-pair regime runs the classic kernels, which have no such word)."""
        torch.cuda.synchronize(self.device)
        if self.per_replica:
            for e in self.envs:
                e.check()

    def step(self, agent: 'BatchedPhysarumAgent', results: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One step of every replica: `agent.forward` + `Env.step` fused, two launches for the whole batch.  Returns the
        (R, 2) float64 tensor of die_step_result words (device; `read_results` decodes)."""
        if results is None:
            results = torch.empty((self.R, 2), dtype=torch.float64, device=self.device)
        if self.per_replica:
            cur = torch.cuda.current_stream(self.device)
            start = cur.record_event()
            for r, (e, st) in enumerate(zip(self.envs, self.streams)):
                st.wait_event(start)
                with torch.cuda.stream(st):
                    self._obs[r], res, *_ = e.step(agent.agents[r].forward(self._obs[r]))
                    results[r].copy_(res)
                cur.wait_stream(st)
            agent._calls += 1
            self._steps += 1
            return results
        self.epoch += 1
        if self.epoch > _lib.OWNER_EPOCH_MAX:
            self.owner.zero_()
            self.epoch = 1
        m, a, dyn, b = self._structs()
        g = agent._struct()
        _lib.check(_lib.lib.die_forward_env_step_batch(C.byref(m), C.byref(a), C.byref(g), None, C.byref(dyn), C.byref(b), _ptr(results),
                                                       _ptr(self._ws), self._ws.numel(), stream_ptr(self.device)),
                   'die_forward_env_step_batch')
        agent._calls += 1
        self.chem, self.chem_next = self.chem_next, self.chem
        self._steps += 1
        return results

    def run(self, agent: 'BatchedPhysarumAgent', n_steps: int) -> torch.Tensor:
        out = torch.empty((n_steps, self.R, 2), dtype=torch.float64, device=self.device)
        for i in range(n_steps):
            self.step(agent, out[i])
        return out

    @staticmethod
    def read_results(results: torch.Tensor):
        host = results.cpu()
        return host[..., 0].numpy().copy(), host[..., 1].contiguous().view(torch.int64).numpy().copy()

    def replica_numpy(self, r: int):
        """(medium (3, W, H), agents (4, K_r)) of replica r, float64 like `Env.medium.to_numpy()` / `Env.agents.to_numpy()`."""
        if self.per_replica:
            torch.cuda.synchronize(self.device)
            return self.envs[r].medium.to_numpy(), self.envs[r].agents.to_numpy()
        occ = (((self.owner[r] >> (32 + _lib.OWNER_EPOCH_SHIFT)) & _lib.OWNER_EPOCH_MAX) == self.epoch).to(torch.float64)
        medium = np.stack([occ.cpu().numpy(), self.food[r].to(torch.float64).cpu().numpy(), self.chem[r].to(torch.float64).cpu().numpy()])
        k = self.n[r]
        q = lambda t: ((t[r, :k].to(torch.int64) & 0xFFFFFFFF).to(torch.float64) / Q32).cpu().numpy()
        agents = np.stack([q(self.x), q(self.y), self.alive[r, :k].to(torch.float64).cpu().numpy(),
                           self.agent_food[r, :k].to(torch.float64).cpu().numpy()])
        return medium, agents


class BatchedPhysarumAgent:
    """R PhysarumAgent objects as one: replica r's headings and Philox streams are those of
    `PhysarumAgent(max_agents=K_r, seed=seed + r, ...)` (core/agent/gradient.py:139-166)."""

    def __init__(self, env: BatchedEnv, scale: float = 0.005, deposit: float = 4.0, sense_offset: float = 0.03,
                 normalized_grad: bool = True, grad_clip: Optional[float] = 1e-5, turn_angle: int = 30, sense_angle: int = 90,
                 turn_tolerance: float = 0.1, seed: int = 0):
        self.env, self.seed = env, int(seed)
        self._calls = 0
        if env.per_replica:
            from .agent.gradient import PhysarumAgent
            self.agents = [PhysarumAgent(max_agents=env.n[r], scale=scale, deposit=deposit, sense_offset=sense_offset,
                                         normalized_grad=normalized_grad, grad_clip=grad_clip, turn_angle=turn_angle,
                                         sense_angle=sense_angle, turn_tolerance=turn_tolerance, seed=self.seed + r) for r in range(env.R)]
            return
        self._p = dict(scale=scale, deposit=deposit, sense_offset=sense_offset, normalized=normalized_grad,
                       grad_clip=-1.0 if grad_clip is None else grad_clip, turn=math.radians(turn_angle),
                       sense=math.radians(sense_angle), rtol=turn_tolerance)
        dev = env.device
        self._hd_hi = torch.zeros((env.R, env.Nmax), dtype=torch.int32, device=dev)
        self._hd_lo = torch.zeros((env.R, env.Nmax), dtype=torch.int32, device=dev)
        for r in range(env.R):                      # the headings a stand-alone agent of that seed starts with
            k = env.n[r]
            _lib.check(_lib.lib.die_init_heading(_ptr(self._hd_hi[r]), _ptr(self._hd_lo[r]), None, None, k, self._p['turn'],
                                                 (self.seed + r) & 0xFFFFFFFFFFFFFFFF, stream_ptr(dev)), 'die_init_heading')
        self._calls = 0

    def _struct(self) -> _lib.GradientAgent:
        p = self._p
        return _lib.GradientAgent(_lib.DIE_AGENT_PHYSARUM, int(bool(p['normalized'])), p['scale'], p['deposit'], 0.0, p['sense_offset'], 0.0,
                                  p['grad_clip'], p['turn'], p['sense'], p['rtol'], _ptr(self._hd_hi), _ptr(self._hd_lo), None, None, None,
                                  self.seed & 0xFFFFFFFFFFFFFFFF, self._calls & 0xFFFFFFFF, 0, None)

    def direction_rads_numpy(self, r: int) -> np.ndarray:
        if self.env.per_replica:
            torch.cuda.synchronize(self.env.device)
            return self.agents[r].direction_rads_numpy()
        k = self.env.n[r]
        return join64(self._hd_hi[r, :k].contiguous(), self._hd_lo[r, :k].contiguous()).cpu().numpy()
