"""Env / Dynamics with the reference's Gymnasium-style API (core/env.py:24-311), state in
HBM and every substep of `step` executed by libdie_hip.so.

Differences a caller can observe, all deliberate:
  * `medium`, `agents`, actions and observations are device-array handles
    (die_amd/device_array.py), not xarray objects; `.to_numpy()` gives the reference layout.
  * the observation is a live view of the state, not a per-step copy (core/env.py:292-293).
  * `Env(..., max_agents=...)`: the reference always allocates W·H agent slots
    (core/data_init.py:143-144) and moves / burns / rewards the dead ones too; that is the
    default here as well.  `max_agents='alive'` allocates only the seeded agents.
  * `reset(seed=...)` honours the seed (the reference ignores it, core/env.py:94-99).
"""
import ctypes as C
import logging
import os
from dataclasses import dataclass
from enum import Enum
from typing import Callable, Optional, Tuple, Union

import numpy as np
import torch

from . import _lib
from .data_init import DataInitializer
from .device_array import DeviceAction, DeviceAgents, DeviceMedium, PendingAction, _ptr, stream_ptr

def _np_round(x: float, decimals: int):
    """np.round(x, decimals) of a scalar (core/env.py:127-128) by numpy's own recipe — multiply, round half to even, divide —
    without the 4 µs of its scalar dispatch: these two calls were 8 µs of a synchronous step."""
    try:
        p = 10.0 ** decimals
        return np.float64(round(x * p) / p)
    except (ValueError, OverflowError):              # nan / inf
        return np.round(x, decimals)


_HOST_SENTINEL_NAN = 0x7FF8DEADBEEF0001      # "not written yet" in the reward word of the pinned result buffer (a NaN no sum produces)

try:                                    # the reference subclasses gym.Env but defines no spaces
    import gymnasium as _gym
    _EnvBase = _gym.Env
except ImportError:                     # gymnasium is optional
    _EnvBase = object


class BoundaryCondition(Enum):
    """core/env.py:24-26."""
    wrap = 'wrap'
    limit = 'limit'


class ActionView(np.ndarray):
    """The (3, N) host action as a numpy array that answers the label-based calls of the reference's xarray action
    (core/base_types.py: channels dx, dy, deposit1): `.sel(channel='dx')`, `.sel(channel=['dx', 'dy'])`, `.coords['channel']`,
    `.values`, `.to_numpy()` — enough for a CostOperator written against the reference (core/env.py:29-39)."""
    CHANNELS = ('dx', 'dy', 'deposit1')

    def __new__(cls, data):
        return np.asarray(data, dtype=np.float64).view(cls)

    def sel(self, channel=None, **kw):
        if kw or channel is None:
            raise NotImplementedError('ActionView.sel selects by channel only')
        if isinstance(channel, str):
            return np.asarray(self)[self.CHANNELS.index(channel)].view(np.ndarray)
        return np.asarray(self)[[self.CHANNELS.index(c) for c in channel]].view(np.ndarray)

    @property
    def coords(self):
        return {'channel': np.array(self.CHANNELS), 'index': np.arange(self.shape[1])}

    @property
    def values(self):
        return np.asarray(self)

    def to_numpy(self):
        return np.asarray(self)


def linear_action_cost(action, weights=(0.02, 0.01)):
    """core/env.py:29-35 — marker for the device cost operator (evaluated in k_move_claim);
    callable on a (3, N) numpy array for host-side use."""
    a = action.to_numpy() if hasattr(action, 'to_numpy') else np.asarray(action)
    return weights[0] * np.abs(a[2]) + weights[1] * np.linalg.norm(a[:2], axis=0)


def zero_cost(action):
    """core/env.py:38-39."""
    return np.zeros(action.shape[1:])


def _identity_food_flow(x):
    return x


@dataclass
class Dynamics:
    """core/env.py:42-61, same field names and defaults."""
    op_action_cost: Callable = linear_action_cost
    op_food_flow: Callable = _identity_food_flow
    rate_feed: float = 0.1
    rate_decay_chem: float = 0.1
    boundary: BoundaryCondition = BoundaryCondition.wrap
    diffuse_mode: str = 'wrap'
    diffuse_sigma: float = .5
    apply_sense_mask: bool = False
    strict_cost: bool = True
    food_infinite: bool = False
    agents_die: bool = False
    agents_born: bool = False
    init_agent_ratio: float = 0.1
    # not a reference field — which `agents_die` the env runs: 'intended' zeroes starved slots and everything follows
    # them; 'reference' reproduces what the reference actually does: `_agent_lifecycle` rebinds `self.agents`
    # (core/env.py:249) while the AgentIndexer keeps the array it was built with (core/utils.py:22), so from the second
    # step on deposits, the agents channel, feeding and `num_agents` use the positions / alive flags frozen at the end
    # of the first step's move, while the agents the policy sees keep moving, starving and being zeroed
    compat: str = 'intended'


class Env(_EnvBase):
    PIC_MIN_CELLS = 1 << 23          # worlds at least this large take the tile-binned step when it applies …
    PIC_MIN_CELLS_64 = 1 << 22       # … with 64×64 tiles already from here (scratch/pic_threshold.py)
    SORT_MIN_CELLS = 1 << 19         # smaller worlds live in the L2: re-sorting the agent arrays only costs launches there

    @classmethod
    def _auto_sort_every(cls, field_size, sort_every) -> int:
        """`sort_every=None`: every 8 steps (measured best at 4096² with the classic step), never on small worlds
        (256²: 20.7 vs 23.7 µs per step, 512²: 21.4 vs 23.7; 1024²: 28.3 vs 24.2 — there the sort pays)."""
        if sort_every is not None:
            return int(sort_every)
        return 8 if int(field_size[0]) * int(field_size[1]) >= cls.SORT_MIN_CELLS else 0

    def __init__(self, field_size: Tuple[int, int], dynamics: Optional[Dynamics] = None, *,
                 max_agents: Union[None, int, str] = None, seed: Optional[int] = None,
                 field_dtype: torch.dtype = torch.float32, device: Union[str, torch.device, None] = None,
                 sync: bool = True, sort_every: Optional[int] = None, staged: bool = False, pic: bool = True):
        if not torch.cuda.is_available():
            raise RuntimeError('die_amd.Env needs a ROCm GPU (MI355X); there is no CPU path')
        self._field_size = (int(field_size[0]), int(field_size[1]))
        self.dynamics = dynamics or Dynamics()
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        self._max_agents = max_agents
        self._field_dtype = field_dtype
        self._sync = sync
        self._sort_every = self._auto_sort_every(field_size, sort_every)
        self._staged = bool(staged)       # cross-checks: one kernel per stage of the step instead of the fused sweep
        self._pic_enabled = bool(pic)     # tile-binned step (die_amd/pic.py) whenever it applies
        self._seed = int.from_bytes(os.urandom(8), 'little') if seed is None else int(seed)
        self._renderer = None
        self.last_result = None
        self._check_dynamics()
        self._init_data(self._field_size)

    # ------------------------------------------------------------------ construction
    def _check_dynamics(self):
        d = self.dynamics
        if d.diffuse_mode not in _lib.DIFFUSE_MODES:
            raise ValueError(f"diffuse_mode={d.diffuse_mode!r}: one of {sorted(_lib.DIFFUSE_MODES)}")
        if d.compat not in ('intended', 'reference'):
            raise ValueError(f"compat={d.compat!r}: 'intended' or 'reference'")
        if d.op_action_cost not in (linear_action_cost, zero_cost) and d.agents_die:
            raise NotImplementedError('a custom op_action_cost runs through a host round trip AFTER the step; with agents_die the '
                                      'lifecycle inside the step would see agent_food before the cost: use linear_action_cost or zero_cost')
        if self._field_size[0] < 2 or self._field_size[1] < 2:
            raise ValueError('field must be at least 2x2')

    def _init_data(self, field_size):
        """core/env.py:74-86."""
        self.medium = DataInitializer.init_field_array(field_size, self.device, self._field_dtype)
        DataInitializer.init_medium(self.medium, self.dynamics.init_agent_ratio, self._seed)
        self.agents, self._num_seeded = DataInitializer.agents_from_medium(self.medium, self._max_agents, self._seed)
        self._after_state_change()

    def _after_state_change(self):
        self._workspace = DataInitializer.workspace(self._field_size, self.agents.N, self.device)
        self._all_alive = bool(self.agents.alive.all().item())
        self._steps = 0
        self._shadow = None
        self._sort_ws = None
        if self._sort_every > 0:            # the re-sort's shadow arrays and workspace exist before the first step, so that
            self._alloc_sort_buffers()      # no step of a timed loop pays for allocations
        self._fuse_forward = True           # False once die_forward_env_step reports the shape unsupported
        self._pic = None                    # PicState, built at the first eligible step
        self._host_read_pending = False     # Env(sync=True): a step's result words were requested and not read yet
        self._pic_status_written = False    # the last step copied the binned step's error word behind its result
        self._status_word = None
        self._frozen = None                 # compat='reference' with agents_die: (x, y, alive, K) the stale indexer sees
        self._pic_tile = None
        self.medium.sense_mask = None
        if self.dynamics.apply_sense_mask:
            W, H = self._field_size
            self.medium.sense_mask = torch.ones((W, H), dtype=torch.uint8, device=self.device)
            self._sense_tmp = torch.empty((W, H), dtype=torch.float64, device=self.device)
            self._update_sense_mask()

    def _update_sense_mask(self):
        """Env._get_sense_mask (core/env.py:276-290): the neighbourhood of the agents channel the next forward() may
        see — sigma 2.0, 3 decimals, as the reference hard-codes them."""
        m = self.medium.c_struct()
        _lib.check(_lib.lib.die_sense_mask(C.byref(m), 2.0, 3, _ptr(self.medium.sense_mask), _ptr(self._sense_tmp),
                                           stream_ptr(self.device)), 'die_sense_mask')


    @classmethod
    def from_numpy(cls, medium: np.ndarray, agents: np.ndarray, dynamics: Optional[Dynamics] = None, **kw) -> 'Env':
        """Build an Env around given (3, W, H) / (4, N) arrays (tests, checkpoints)."""
        env = cls.__new__(cls)
        if not torch.cuda.is_available():
            raise RuntimeError('die_amd.Env needs a ROCm GPU (MI355X); there is no CPU path')
        medium = np.asarray(medium)
        env._field_size = (medium.shape[1], medium.shape[2])
        env.dynamics = dynamics or Dynamics()
        env.device = torch.device(kw.get('device') or f'cuda:{torch.cuda.current_device()}')
        env._max_agents = agents.shape[1]
        env._field_dtype = kw.get('field_dtype', torch.float32)
        env._sync = kw.get('sync', True)
        env._sort_every = cls._auto_sort_every(env._field_size, kw.get('sort_every'))
        env._staged = bool(kw.get('staged', False))
        env._pic_enabled = bool(kw.get('pic', True))
        env._seed = int(kw.get('seed', 0))
        env._renderer = None
        env.last_result = None
        env._check_dynamics()
        env.medium = DeviceMedium(env._field_size, env.device, env._field_dtype)
        env.medium.upload(medium)
        env.agents = DeviceAgents(agents.shape[1], env.device)
        env.agents.upload(agents)
        env._num_seeded = int((np.asarray(agents)[2] > 0).sum())
        env._after_state_change()
        return env

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        """core/env.py:94-99."""
        if seed is not None:
            self._seed = int(seed)
        else:
            self._seed = (self._seed * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        self._init_data(self._field_size)
        return self._get_current_obs, {}

    # ------------------------------------------------------------------ step
    def _c_dynamics(self) -> _lib.Dynamics:
        d = self.dynamics
        if isinstance(d.boundary, BoundaryCondition):
            boundary = _lib.DIE_BOUNDARY_WRAP if d.boundary == BoundaryCondition.wrap else _lib.DIE_BOUNDARY_LIMIT
        else:
            logging.warning(f'Unfamiliar boundary condition: {d.boundary}!')      # core/env.py:158-161
            boundary = _lib.DIE_BOUNDARY_NONE
        # (a custom cost operator: the step runs with zero cost, `_apply_custom_cost` subtracts the operator's values afterwards)
        cost = _lib.DIE_COST_LINEAR if d.op_action_cost is linear_action_cost else _lib.DIE_COST_ZERO
        return _lib.Dynamics(d.rate_feed, d.rate_decay_chem, d.diffuse_sigma, boundary, cost, 0.02, 0.01,
                             int(d.food_infinite), int(d.agents_die), int(not self._all_alive), _lib.DIFFUSE_MODES[d.diffuse_mode],
                             int(self._staged))

    def _as_action(self, action) -> DeviceAction:
        if isinstance(action, DeviceAction):
            act = action
        else:
            act = DeviceAction.from_numpy(action.to_numpy() if hasattr(action, 'to_numpy') else action, self.device)
        if act.N != self.agents.N:
            raise ValueError(f'action has {act.N} slots, env has {self.agents.N}')
        return act.in_order_of(self.agents.slot)

    def _step_compat(self, action, result):
        """One step of `Dynamics(agents_die=True, compat='reference')`: the stages of die_env_step one call each, the
        claim / deposit / feeding stages looking agents up in the frozen copy (see Dynamics.compat)."""
        act = self._as_action(action)
        A, M = self.agents, self.medium
        M.next_epoch()
        sp = stream_ptr(self.device)
        ws, wsn = _ptr(self._workspace), self._workspace.numel()
        m, a, u, d = M.c_struct(), A.c_struct(), act.c_struct(), self._c_dynamics()
        d.agents_die = 0
        tile_of = torch.empty(A.N, dtype=torch.int32, device=self.device)
        _lib.check(_lib.lib.die_agent_move(C.byref(m), C.byref(a), C.byref(u), C.byref(d), M.W, M.H, 1, _ptr(tile_of), sp), 'die_agent_move')
        if self._frozen is None:            # AgentIndexer's array: what `self.agents` was when the lifecycle first rebound it
            self._frozen = (A.x.clone(), A.y.clone(), A.alive.clone(), int(A.alive.sum().item()))
        fx, fy, falive, K0 = self._frozen
        d.has_dead_slots = int(K0 < A.N)
        af = _lib.Agents(A.N, _ptr(fx), _ptr(fy), _ptr(falive), _ptr(A.agent_food), _ptr(A.slot))
        _lib.check(_lib.lib.die_agent_claim_feed(C.byref(m), C.byref(af), C.byref(u), C.byref(d), ws, wsn, sp), 'die_agent_claim_feed')
        if d.has_dead_slots:
            _lib.check(_lib.lib.die_agent_dead_slots(C.byref(m), C.byref(af), C.byref(u), C.byref(d), ws, wsn, sp), 'die_agent_dead_slots')
        _lib.check(_lib.lib.die_step_reduce_ex(C.byref(a), _ptr(result), ws, wsn, 3 if d.has_dead_slots else 0, K0, sp), 'die_step_reduce_ex')
        _lib.check(_lib.lib.die_medium_deposit_feed_diffuse(C.byref(m), C.byref(d), sp), 'die_medium_deposit_feed_diffuse')
        _lib.check(_lib.lib.die_agents_lifecycle(C.byref(a), sp), 'die_agents_lifecycle')

    def _custom_cost(self, action):
        """`Dynamics.op_action_cost` as an arbitrary callable (core/env.py:43,209: any CostOperator): evaluated on the host on
        the (3, N) action in slot order, like `op_food_flow` — a round trip per step.  The operator receives a numpy array that
        also answers the xarray calls the reference's own operators make (`action.sel(channel=[...])`, core/env.py:29-35), so a
        CostOperator written against the reference runs unchanged.  Returns the (N,) device tensor of costs."""
        a = action.to_numpy() if hasattr(action, 'to_numpy') else np.asarray(action, dtype=np.float64)
        burned = np.asarray(self.dynamics.op_action_cost(ActionView(a)), dtype=np.float64).reshape(-1)
        if burned.shape[0] != self.agents.N:
            raise ValueError(f'op_action_cost returned {burned.shape[0]} values for {self.agents.N} slots')
        return torch.from_numpy(burned.astype(np.float32)).to(self.device)

    def _apply_custom_cost(self, burned, result):
        """agent_food −= burned, reward −= Σ burned (core/env.py:229-243: agent_food += consumed − burned over ALL slots); the
        step itself ran with zero cost."""
        A = self.agents
        A.agent_food -= burned if A.slot is None else burned[A.slot.long()]
        result[0] -= burned.double().sum()

    def step(self, action):
        """core/env.py:101-131 → (obs, reward, terminated, truncated, info)."""
        fused = binned = False
        burned = None
        if self.dynamics.op_action_cost not in (linear_action_cost, zero_cost):
            burned = self._custom_cost(action)
        # sync=True: reward, num_agents and the binned step's error word (die_pic.status_out) are three words that ONE thread of
        # the step's last kernel writes (every path: k_reduce, the sweep's reduction workgroup, the field kernel's) — straight
        # into pinned host memory, where the host waits for exactly those words instead of a 24-byte copy behind a stream
        # synchronisation (≈ 20 µs of a 187-µs step at 4096², most of a step at 256²).  A custom cost corrects the result on
        # the device afterwards: that case keeps the copy.
        host = self._sync and burned is None and self._host_result_buffer() is not None
        if host:
            res3 = self._host_res
            if self._host_read_pending:
                # the previous step's words were never waited for (it raised between its launch and its read): its last kernel
                # may still be on its way and would satisfy THIS step's wait with the old values — let it finish first
                torch.cuda.synchronize(self.device)
            self._host_i64[0], self._host_i64[1], self._host_i64[2] = _HOST_SENTINEL_NAN, -1, -1
            self._host_read_pending = True
        else:
            res3 = torch.empty(3, dtype=torch.float64, device=self.device) if self._sync else None
        result = res3[:2] if res3 is not None else torch.empty(2, dtype=torch.float64, device=self.device)
        self._status_word = res3
        self._pic_status_written = False
        if self._pic is not None:
            self._pic.flush_lazy()          # an un-read action of the previous binned step: its inputs are about to change
        if self.dynamics.agents_die and self.dynamics.compat == 'reference':
            self._step_compat(action, result)
            fused = binned = True           # (no re-sorting either: the frozen copy is in this array order)
        if not fused and isinstance(action, PendingAction) and action.pending and action.agents is self.agents \
                and action.medium is self.medium and action.slot is self.agents.slot and self._fuse_forward:
            fused = binned = self._pic_step(action, result)
        if not fused and isinstance(action, PendingAction) and action.pending and action.agents is self.agents \
                and action.medium is self.medium and action.slot is self.agents.slot and self._fuse_forward:
            # `env.step(agent.forward(obs))`: forward runs fused with the move / claim pass
            self.medium.next_epoch()
            m, a, u, d = self.medium.c_struct(), self.agents.c_struct(), action.raw_struct(), self._c_dynamics()
            ws, wsn, sp = _ptr(self._workspace), self._workspace.numel(), stream_ptr(self.device)
            # (running the periodic re-sort on a second stream next to the sweep was measured: 223 µs/step
            # against 220 µs with the sort after the step — the sweep is bandwidth-bound, nothing to hide behind)
            rc = _lib.lib.die_forward_env_step(C.byref(m), C.byref(a), C.byref(action.g_struct), C.byref(u), C.byref(d),
                                               _ptr(result), ws, wsn, sp)
            if rc == -3:                            # DIE_ERR_UNSUPPORTED for this shape: two calls instead
                self._fuse_forward = False
                self.medium.epoch -= 1
            else:
                _lib.check(rc, 'die_forward_env_step')
                action.agent._forward_consumed(action)
                fused = True
        if not fused:
            act = self._as_action(action)
            self.medium.next_epoch()
            m, a, u, d = self.medium.c_struct(), self.agents.c_struct(), act.c_struct(), self._c_dynamics()
            _lib.check(_lib.lib.die_env_step(C.byref(m), C.byref(a), C.byref(u), C.byref(d), _ptr(result),
                                             _ptr(self._workspace), self._workspace.numel(), stream_ptr(self.device)),
                       'die_env_step')
        if not binned:
            self._agents_changed()          # a classic step moved the agents in place: the tile order is gone, bin again
        if burned is not None:
            self._apply_custom_cost(burned, result)
        self.medium.swap_chem()
        if self.dynamics.op_food_flow is not _identity_food_flow:
            self._food_flow()
        if self.dynamics.apply_sense_mask:
            self._update_sense_mask()
        self._steps += 1
        if self._sort_every > 0 and self._steps % self._sort_every == 0 and not binned:
            self.sort_agents()              # (the tile-binned step keeps the arrays exactly sorted by itself)
        self.last_result = result
        if not self._sync:
            return self._get_current_obs, result, False, False, {}
        if host:
            reward, num_agents = self._read_host_result()
            self._host_read_pending = False
            self.last_result = res3[:2].clone()
        else:
            reward, num_agents = self.read_result(res3)
        if self.dynamics.agents_die:
            self._all_alive = False
        mean_gain = reward / num_agents if num_agents > 0 else 0.
        info = {'num_agents': num_agents, 'reward': _np_round(reward, 3), 'mean_reward': _np_round(mean_gain, 5)}
        return self._get_current_obs, reward, num_agents == 0, False, info

    def read_result(self, result: torch.Tensor) -> Tuple[float, int]:
        """(reward, num_agents) of a die_step_result buffer (synchronises).  A 3-word buffer (`Env(sync=True)` builds one per
        step) carries the tile-binned step's error word behind the result: one host copy instead of two."""
        host = result.cpu()
        if self._pic is not None:
            if host.numel() >= 3 and self._pic_status_written:
                self._pic.raise_for(int(host.view(torch.int64)[2]))
            elif self._pic.steps_since_check:
                self._pic.check()
        return float(host[0]), int(host.view(torch.int64)[1])

    def _host_result_buffer(self):
        """A pinned host buffer of 3 words that the device can write (same address on both sides), or None.  Checked once."""
        if not hasattr(self, '_host_res'):
            self._host_res = self._host_i64 = None
            try:
                if os.environ.get('DIE_HOST_RESULT', '1') != '0':
                    buf = torch.zeros(3, dtype=torch.float64).pin_memory()
                    # (through libdie_hip.so: the HIP runtime instance that launches the kernels; queried on THIS env's device)
                    dev = C.c_void_p()
                    rc = _lib.lib.die_host_device_pointer(C.c_void_p(buf.data_ptr()), self.device.index if self.device.index is not None else -1, C.byref(dev))
                    if rc == 0 and dev.value == buf.data_ptr():
                        self._host_res, self._host_i64 = buf, buf.numpy().view(np.int64)
                    else:
                        logging.getLogger('die_amd').info('Env(sync=True): no device-visible host buffer (%s): the step result is read by copy',
                                                          _lib.lib.die_last_error().decode() if rc else 'address differs on the device')
            except Exception as e:
                logging.getLogger('die_amd').info('Env(sync=True): the step result is read by copy (%s: %s)', type(e).__name__, e)
                self._host_res = self._host_i64 = None
        return self._host_res

    def _read_host_result(self) -> Tuple[float, int]:
        """Wait for the three words the field kernel writes into the pinned buffer (die_pic.status_out), then what
        read_result does."""
        import time
        v, t0, spins, err = self._host_i64, None, 0, self._pic_status_written
        while v[0] == _HOST_SENTINEL_NAN or v[1] == -1 or (err and v[2] == -1):
            spins += 1
            if spins & 0xFFFF == 0:                       # (not expected: a step takes microseconds)
                t0 = t0 or time.perf_counter()
                if time.perf_counter() - t0 > 20.0:
                    torch.cuda.synchronize(self.device)
                    if v[0] == _HOST_SENTINEL_NAN or v[1] == -1 or (err and v[2] == -1):
                        raise RuntimeError('the step result never arrived in host memory (set DIE_HOST_RESULT=0 to read it by copy)')
        if self._pic is not None:
            if err:
                self._pic.raise_for(int(v[2]))
            elif self._pic.steps_since_check:
                self._pic.check()
        return float(self._host_res[0]), int(v[1])

    def check(self):
        """Synchronise and raise if the tile-binned step has reported a bookkeeping error since the last check (an agent
        that moved further than a tile).  `sync=True` steps and `read_result` do this by themselves; loops that never read
        anything back (`sync=False`, `run`) call it when they want to know."""
        torch.cuda.synchronize(self.device)
        if self._pic is not None and self._pic.steps_since_check:
            self._pic.check()

    # ------------------------------------------------------------------ tile-binned step (die_amd/pic.py)
    def _pic_applies(self, action) -> bool:
        d, ag = self.dynamics, action.agent
        if self._pic_tile is False:                              # (settled at the first step: this world is too small / does not divide into tiles)
            return False
        # (dead slots — the reference's default max_agents = W·H — ride behind the tiles' segments: without agents_die they stay dead)
        if not (self._pic_enabled and not d.agents_die and not d.apply_sense_mask and not self._staged
                and self.medium.world is None and isinstance(d.boundary, BoundaryCondition) and d.diffuse_mode == 'wrap'
                and 1 <= int(4.0 * float(d.diffuse_sigma) + 0.5) <= 4):
            return False
        if not (ag._normalized and ag._step_base is None):
            return False
        momentum = ag._inertia != 0 or ag._noise_scale != 0
        if momentum and (ag._kind != _lib.DIE_AGENT_GRADIENT or not self._all_alive):
            return False                                         # (a PhysarumAgent with momentum, momentum with dead slots: the classic step)
        if ag._prev_grad is not None and ag._inertia == 0:
            return False                                         # (_prev_grad is kept up to date by the classic step only when nothing reads it: inertia 0 — ADVICE r4)
        W, H = self._field_size
        from .pic import step_scale
        eff_scale = step_scale(ag)                               # |scale| · the bound of the vector it multiplies (momentum: die_pic_step_bound)
        reach = abs(eff_scale) * (max(W, H) - 1)                 # cells per step, at most
        if self._pic_tile is None:
            from .pic import pick_tile
            # small worlds are bound by launches and their gaps, not by what the binned step saves.  Measured (r3, µs per step,
            # classic / binned with 64×64 / 32×64 tiles, scratch/pic_threshold.py): 1024² 23.8 / 27.7 / 39.5, 1536² 41.2 / 39.5 /
            # 53.4, 2048² 58.5 / 53.3 / 70.7 (bench.py, with its per-step events: 58.4 / 57.1), 3072×2048 78.0 / 64.5 / 80.5,
            # 3072² 107 / 90 / 112, 4096×2048 95.6 / 77.6 / 95.2: 64×64 tiles from 2^22 cells upwards, the smaller shapes (worlds
            # that 64 does not divide) from 2^23
            # fp16 field channels (BASELINE configs[4]): 32×128 tiles first.  A row of a 64-cell-wide tile is ONE 128-byte line of an
            # fp16 plane; with 128 cells per row the field kernel's loads and stores run 256 bytes like the fp32 planes' — measured
            # (r6, profiles/r06_f16_tile_shapes.txt): 4096² 9 132 → 9 540 steps/s, 16384² 541 → 587 (field kernel 878 → 751 µs)
            from .pic import TILE_SHAPES
            f16 = self.medium.dtype == torch.float16
            if W * H >= self.PIC_MIN_CELLS:
                self._pic_tile = pick_tile(W, H, reach, shapes=(((5, 7),) if f16 else ()) + TILE_SHAPES) or False
            elif W * H >= self.PIC_MIN_CELLS_64:
                self._pic_tile = pick_tile(W, H, reach, shapes=((5, 7), (6, 6)) if f16 else ((6, 6),)) or False
            else:
                self._pic_tile = False
        if not (bool(self._pic_tile) and reach <= min(1 << self._pic_tile[0], 1 << self._pic_tile[1]) - 1):
            return False
        if not self._all_alive:                  # dead slots ride along in the two-launch form only
            return _lib.lib.die_pic_two_launch(max(W, H), self._pic_tile[0], self._pic_tile[1], eff_scale, float(d.diffuse_sigma), 0) == 1 \
                and getattr(self, '_pic_fused', True)
        return True

    def _pic_step(self, action, result) -> bool:
        """`env.step(agent.forward(obs))` on tile-binned agents: die_pic_forward_env_step.  False: not applicable, the
        caller takes the classic path (nothing has been touched)."""
        if not self._pic_applies(action):
            return False
        from .pic import PicState
        ag = action.agent
        if self._pic is None:
            self._pic_n_alive = 0 if self._all_alive else int(self.agents.alive.sum().item())
            if not self._all_alive and self._pic_n_alive == 0:
                return False                        # (nobody alive: nothing to bin)
            self._pic = PicState(self, self._pic_tile)
            if getattr(self, '_pic_k1_threads', 0):
                self._pic.k1_threads = int(self._pic_k1_threads)
            self._pic.lazy_actions = bool(getattr(self, '_pic_lazy_actions', True))
        if not self._pic.is_current(self, ag):
            self._pic.bin(self, ag)
            action.rebind(self.agents)
        status = getattr(self, '_status_word', None)
        two = self._pic.two_launch(self, ag)
        _lib.check(self._pic.step(self, ag, action, self._c_dynamics(), result, getattr(self, '_pic_events', None),
                                  status_out=status.data_ptr() + 16 if status is not None and two else None,
                                  plan=getattr(self, '_pic_plan', None)),       # (scratch/split_step_cost.py: the step as launches over subsets of the tiles)
                   'die_pic_forward_env_step')
        self._pic_status_written = status is not None and two
        ag._forward_consumed(action)
        self.medium.owner_stale = self._mark_owner
        return True

    def _mark_owner(self):
        """Rebuild the 'agents' channel (claim plane) from the agent arrays after tile-binned steps."""
        M = self.medium
        M.next_epoch()
        m, a = M.c_struct(need_owner=False), self.agents.c_struct()
        _lib.check(_lib.lib.die_agents_mark_owner(C.byref(m), C.byref(a), stream_ptr(self.device)), 'die_agents_mark_owner')

    # ------------------------------------------------------------------ many steps without the host in the loop
    def run(self, agent, n_steps: int, graph: Optional[bool] = None) -> torch.Tensor:
        """`n_steps` × `obs, … = env.step(agent.forward(obs))` (the loop of examples/minimal_run.py:23-25) without reading
        anything back in between.  Returns the (n_steps, 2) float64 device tensor of die_step_result words
        (`read_results` decodes it).  `graph=True` (Gradient / Physarum agents): the loop is captured ONCE as a hipGraph
        of K = lcm(epochs 31, chem planes 2, two re-sorts) steps and replayed; the launch arguments of a graph are
        frozen, so the Philox step counter comes from a device word (`step_base`).  Same bits as the step-by-step loop
        (tests/test_gpu_parity.py::test_graph_run_equals_step_loop).  Measured on MI355X / ROCm 7 the replay is NOT
        faster than the plain launch loop (256²: 33.8 vs 30.4 µs/step, 1024²: 49.0 vs 42.3, 4096²: 243 vs 224 —
        ≈ 5 µs per graph node), so the default is the loop."""
        from .agent.gradient import GradientAgent
        n_steps = int(n_steps)
        out = torch.empty((max(n_steps, 0), 2), dtype=torch.float64, device=self.device)
        sync, self._sync = self._sync, False
        try:
            ok = isinstance(agent, GradientAgent) and agent.lazy and agent._turn_sign is None and \
                self.dynamics.op_food_flow is _identity_food_flow and not self.dynamics.agents_die and self._fuse_forward
            if graph is None:
                graph = False
            elif graph and not ok:
                raise ValueError('Env.run(graph=True) needs a lazy Gradient/Physarum agent, identity food flow, no agents_die')
            done = 0
            obs = self._get_current_obs
            K = self._graph_period()
            if graph and n_steps >= K:
                # plain steps first until the lazily created state exists (agent arrays; the slot array of the first
                # re-sort): the graph must start and end on the same storages
                while done < n_steps - K and (self._steps == 0 or (self._sort_every > 0 and self.agents.slot is None)):
                    obs, res, *_ = self.step(agent.forward(obs))
                    out[done].copy_(res)
                    done += 1
                while n_steps - done >= K and self._steps > 0 and (self._sort_every <= 0 or self.agents.slot is not None):
                    G = self._graph_for(agent, K)
                    G['base'].fill_((agent._calls - G['calls0']) & 0x7FFFFFFF)
                    G['graph'].replay()
                    out[done:done + K].copy_(G['results'])
                    self._steps += K
                    agent._calls += K
                    done += K
                self.last_result = out[done - 1] if done else self.last_result
            if ok and not graph:
                done += self._run_binned(agent, n_steps - done, out[done:])
            obs = self._get_current_obs
            for i in range(done, n_steps):
                obs, res, *_ = self.step(agent.forward(obs))
                out[i].copy_(res)
        finally:
            self._sync = sync
        if sync:                        # (an env built with sync=True reads results back anyway: report a broken layout now)
            self.check()
        return out

    def _run_binned(self, agent, n: int, results: torch.Tensor) -> int:
        """The tile-binned two-launch steps of `run` as ONE library call (die_pic_run: a C loop over die_pic_forward_env_step — no
        Python, no ctypes marshalling between two steps; SURVEY §8b's `die_step_fused(handle, n_steps)`).  Returns the number of steps
        it took: 0 when this world / agent does not take the binned two-launch step (the caller's step loop does everything)."""
        from types import SimpleNamespace
        from .pic import PicState
        if n <= 0 or self.dynamics.op_action_cost not in (linear_action_cost, zero_cost) or getattr(self, '_pic_events', None) is not None \
                or getattr(self, '_pic_plan', None) is not None or (self.dynamics.agents_die and self.dynamics.compat == 'reference'):
            return 0
        if not self._pic_applies(SimpleNamespace(agent=agent)):
            return 0
        from .pic import step_scale
        if _lib.lib.die_pic_two_launch(max(self._field_size), self._pic_tile[0], self._pic_tile[1], step_scale(agent),
                                       float(self.dynamics.diffuse_sigma), 0) != 1 or not getattr(self, '_pic_fused', True):
            return 0
        if not (getattr(agent, 'lazy', False) and agent._turn_sign is None):
            return 0                                                    # (forward() must hand out a PENDING action: nothing may run before the library call)
        if self._pic is None:
            self._pic_n_alive = 0 if self._all_alive else int(self.agents.alive.sum().item())
            if not self._all_alive and self._pic_n_alive == 0:
                return 0                                                # (nobody alive: nothing to bin)
            self._pic = PicState(self, self._pic_tile)
            if getattr(self, '_pic_k1_threads', 0):
                self._pic.k1_threads = int(self._pic_k1_threads)
            self._pic.lazy_actions = bool(getattr(self, '_pic_lazy_actions', True))
        self._pic.flush_lazy()
        action = agent.forward(self._get_current_obs)                   # the first step's pending action: seed, step counter, state pointers
        assert isinstance(action, PendingAction) and action.pending
        if not self._pic.is_current(self, agent):
            self._pic.bin(self, agent)
            action.rebind(self.agents)
        rc = self._pic.run(self, agent, action, self._c_dynamics(), results, n)
        n = self._pic.run_done                                          # (= n unless the call failed part-way: the state reached is adopted first)
        if n > 0:
            agent._forward_consumed(action)
            agent._calls += n - 1
            if n & 1:
                self.medium.swap_chem()
            self.medium.owner_stale = self._mark_owner
            self._steps += n
            self.last_result = results[n - 1]
        _lib.check(rc, 'die_pic_run')
        self.library_runs = getattr(self, 'library_runs', 0) + 1
        return n

    @staticmethod
    def read_results(results: torch.Tensor) -> Tuple[np.ndarray, np.ndarray]:
        """(rewards (n,), num_agents (n,)) of the tensor `run` returns (synchronises)."""
        host = results.cpu()
        return host[:, 0].numpy().copy(), host[:, 1].contiguous().view(torch.int64).numpy().copy()

    def _graph_period(self) -> int:
        import math
        k = math.lcm(_lib.OWNER_EPOCH_MAX, 2)
        if self._sort_every > 0:
            k = math.lcm(k, 2 * self._sort_every)
        return k

    def _graph_for(self, agent, K: int):
        """The captured K-step graph for the current alignment (epoch, chem plane, sort phase, array storages)."""
        A, M = self.agents, self.medium
        key = (id(agent), K, M.epoch, M.chem.data_ptr(), self._steps % max(self._sort_every, 1), A.x.data_ptr(),
               agent._hd_hi.data_ptr() if agent._hd_hi is not None else 0, A.N)
        cache = self.__dict__.setdefault('_graphs', {})
        G = cache.get(key)
        if G is not None:
            return G
        cache.clear()                                  # one alignment at a time: the pools hold K steps of temporaries
        # everything the K steps may re-seat, and where it has to be again afterwards
        holders = [(A, n) for n in ('x', 'y', 'alive', 'agent_food', 'slot')] + [(M, 'chem'), (M, 'chem_next')] + \
                  [(agent, '_hd_hi'), (agent, '_hd_lo'), (agent, '_prev_grad'), (agent, '_order'), (self, '_shadow')]
        init = [(o, n, getattr(o, n)) for o, n in holders]
        steps0, calls0, epoch0 = self._steps, agent._calls, M.epoch
        results = torch.empty((K, 2), dtype=torch.float64, device=self.device)
        base = torch.zeros(1, dtype=torch.int32, device=self.device)
        agent._step_base = base
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(self.device)
        try:
            with torch.cuda.graph(g):
                obs = self._get_current_obs
                for i in range(K):
                    obs, res, *_ = self.step(agent.forward(obs))
                    results[i].copy_(res)
                for o, n, t0 in init:                  # state back into the storages the next replay starts from
                    cur = getattr(o, n)
                    if isinstance(t0, torch.Tensor) and cur is not t0 and n not in ('_order', 'chem', 'chem_next'):
                        t0.copy_(cur)
                base.add_(K)
        finally:
            agent._step_base = None
            for o, n, t0 in init:                      # capture executed nothing: the host-side bookkeeping goes back too
                setattr(o, n, t0)
            self._steps, agent._calls, M.epoch = steps0, calls0, epoch0
            agent._pending = None
        if agent._order is not None:
            agent._order = A.slot
        G = dict(graph=g, results=results, base=base, calls0=calls0)
        cache[key] = G
        return G

    def _food_flow(self):
        """core/env.py:147-150.  The WaveSequence operator runs on the device (die_food_flow_wave); an arbitrary Python
        operator needs a host round trip.  (It touches env_food only, so running it after the chem sweep instead of
        before it, as the reference does, gives the same state.)"""
        from .data_init import WaveFoodFlow
        if isinstance(self.dynamics.op_food_flow, WaveFoodFlow):
            self.dynamics.op_food_flow.apply(self.medium)
            return
        food = self.medium.food.to(torch.float64).cpu().numpy()
        self.medium.upload_channel('env_food', np.asarray(self.dynamics.op_food_flow(food)))

    def _alloc_sort_buffers(self):
        A = self.agents
        self._shadow = [torch.empty_like(t) for t in (A.x, A.y, A.alive, A.agent_food)]
        n = _lib.lib.die_sort_workspace_bytes(self.medium.W, self.medium.H, A.N)
        self._sort_ws = torch.empty(n, dtype=torch.uint8, device=self.device)

    def sort_agents(self):
        """Re-order the agent arrays so that array neighbours are grid neighbours (die_agents_sort).
        Invisible to callers: slot ids travel with the agents, attached Agent objects have their
        per-slot state permuted alongside."""
        A = self.agents
        if self._shadow is None:
            self._alloc_sort_buffers()
        ox, oy, oalive, ofood = self._shadow[:4]
        # the slot array is NEVER recycled: every action computed so far holds the one that was current then, to un-permute
        # itself when it is read (found by tests/fuzz_cases.py: an action read two sorts later saw a recycled array)
        oslot = torch.empty(A.N, dtype=torch.int32, device=self.device)
        owners, tensors = [], []
        for obj in A.attached():
            ts = obj._die_state_tensors(A)
            if ts and len(tensors) + len(ts) <= 4:
                owners.append((obj, len(ts)))
                tensors += ts
        outs = [torch.empty_like(t) for t in tensors]
        ein = (C.c_void_p * max(len(tensors), 1))(*[t.data_ptr() for t in tensors])
        eout = (C.c_void_p * max(len(outs), 1))(*[t.data_ptr() for t in outs])
        m, a_in = self.medium.c_struct(need_owner=False), A.c_struct()       # (the sort reads positions only)
        a_out = _lib.Agents(A.N, _ptr(ox), _ptr(oy), _ptr(oalive), _ptr(ofood), _ptr(oslot))
        _lib.check(_lib.lib.die_agents_sort(C.byref(m), C.byref(a_in), C.byref(a_out), len(tensors), ein, eout,
                                            _ptr(self._sort_ws), self._sort_ws.numel(), stream_ptr(self.device)),
                   'die_agents_sort')
        self._shadow = [A.x, A.y, A.alive, A.agent_food]
        A.x, A.y, A.alive, A.agent_food, A.slot = ox, oy, oalive, ofood, oslot
        k = 0
        for obj, n in owners:
            obj._die_state_permuted(outs[k:k + n], A.slot)
            k += n

    # substeps, for custom update cycles (examples/simple_agents.py:16-30)
    def _agents_changed(self):
        """The agent arrays were modified in place by something other than the tile-binned step: its tile order is void."""
        if self._pic is not None:
            self._pic.flush_lazy()
            self._pic.held = None

    def _stage(self, fn_name, action):
        self._agents_changed()
        act = self._as_action(action)
        m, a, u, d = self.medium.c_struct(), self.agents.c_struct(), act.c_struct(), self._c_dynamics()
        _lib.check(getattr(_lib.lib, fn_name)(C.byref(m), C.byref(a), C.byref(u), C.byref(d), _ptr(self._workspace),
                                              self._workspace.numel(), stream_ptr(self.device)), fn_name)

    def _medium_deposit_feed_diffuse(self):
        """Deposit + feeding + diffusion in one field sweep (after `_stage('die_agent_move_claim')`)."""
        m, d = self.medium.c_struct(), self._c_dynamics()
        _lib.check(_lib.lib.die_medium_deposit_feed_diffuse(C.byref(m), C.byref(d), stream_ptr(self.device)),
                   'die_medium_deposit_feed_diffuse')
        self.medium.swap_chem()

    def _medium_diffuse_decay(self):
        d = self.dynamics
        m = self.medium
        _lib.check(_lib.lib.die_diffuse_decay_mode(_ptr(m.chem), _ptr(m.chem_next), m.W, m.H, m.c_struct().dtype,
                                                   d.diffuse_sigma, d.rate_decay_chem, _lib.DIFFUSE_MODES[d.diffuse_mode],
                                                   stream_ptr(self.device)), 'die_diffuse_decay_mode')
        m.swap_chem()

    # ------------------------------------------------------------------ observation / render
    @property
    def _num_alive_agents(self) -> int:
        return int(self.agents.alive.sum().item())

    @property
    def _get_agent_mask(self) -> torch.Tensor:
        return self.medium.occupied()

    @property
    def _get_current_obs(self):
        """core/env.py:296-298 (live handles, not copies)."""
        return self.agents, self.medium

    @property
    def coordgrid(self) -> np.ndarray:
        """core/utils.py:113-118."""
        xcs = [np.linspace(0., 1., num=size) for size in reversed(self._field_size)]
        return np.stack(np.meshgrid(*xcs))

    def render(self):
        """core/env.py:133-134: [medium RGB (W, H, 3), agent-trace RGBA, agents RGBA], built on the device
        (die_render_frames) and downloaded as float32 images."""
        from .render import DeviceRenderer
        if self._renderer is None:
            self._renderer = DeviceRenderer(self._field_size, self.device)
        return self._renderer.render(self.medium, self.agents)

    def render_rgb8(self) -> np.ndarray:
        """The medium frame as one (W, H, 3) uint8 image — what a per-step plot loop needs to download."""
        from .render import DeviceRenderer
        if self._renderer is None:
            self._renderer = DeviceRenderer(self._field_size, self.device)
        return self._renderer.rgb8(self.medium)
