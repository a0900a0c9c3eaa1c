"""Device-resident stand-ins for the reference's xarray objects (core/base_types.py:6-10):
`medium` (3, W, H), `agents` (4, N) and `action` (3, N).  Each holds one torch tensor per
channel in HBM and hands raw device pointers to libdie_hip.so; `.to_numpy()` gives the
float64 array the reference would hold, `.sel(channel=...)` one channel.

torch is plumbing here (allocation, streams, host copies); no torch op is on the step path.
"""
import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from .base_types import DataChannels

Q32 = 4294967296.0


def _ptr(t: Optional[torch.Tensor]):
    """Device address for a `c_void_p` argument or struct field (ctypes takes a plain int there, None = NULL; every
    entry point has its argtypes declared in _lib.py)."""
    return t.data_ptr() if t is not None else None


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr(device):
    """The current HIP stream of `device` as an integer handle.  (torch.cuda.current_stream() builds a Stream object:
    5 µs of a 28-µs host step on a launch-bound world; the raw getter is 0.2 µs.)"""
    if _raw_stream is not None:
        idx = device.index if isinstance(device, torch.device) else torch.device(device).index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


def to_q32(v: np.ndarray) -> np.ndarray:
    """[0, 1] float coordinates → Q0.32 words (1.0 saturates to 2^32 − 1)."""
    q = np.rint(np.asarray(v, dtype=np.float64) * Q32)
    return np.clip(q, 0, Q32 - 1).astype(np.uint64).astype(np.uint32)


def from_q32(q: np.ndarray) -> np.ndarray:
    return q.astype(np.float64) / Q32


class DeviceMedium:
    """(3, W, H) field: channels ('agents', 'env_food', 'chem1')."""
    channels = DataChannels.medium

    def __init__(self, field_size, device, dtype=torch.float32):
        W, H = int(field_size[0]), int(field_size[1])
        if dtype not in (torch.float32, torch.float16):
            raise ValueError('field dtype must be float32 or float16')
        self.W, self.H, self.device, self.dtype = W, H, torch.device(device), dtype
        self.owner = torch.zeros((W, H), dtype=torch.int64, device=device)   # uint64 claim words
        self.food = torch.zeros((W, H), dtype=dtype, device=device)
        self.chem = torch.zeros((W, H), dtype=dtype, device=device)
        self.chem_next = torch.empty((W, H), dtype=dtype, device=device)
        self.epoch = 1
        self.world = None        # (gW, gH, ox, oy) when these planes are one tile of a decomposed world
        self.owner_stale = None  # callable that rebuilds `owner` (the tile-binned step does not maintain the claim plane)
        # callable run before anything SENSES these planes outside a fused step — a stand-alone forward, a host read: a decomposed
        # rank whose ghost refresh was left to travel under the next step (die_amd/dist.py) does it now instead.  The fused step
        # itself takes the planes through c_struct() and never triggers it
        self.before_sense = None

    @property
    def shape(self):
        return (3, self.W, self.H)

    def _ensure_owner(self):
        if self.owner_stale is not None:
            rebuild, self.owner_stale = self.owner_stale, None
            rebuild()

    def c_struct(self, need_owner: bool = True) -> _lib.Medium:
        """`need_owner=False`: the callee does not read the claim plane (or overwrites it with a new epoch)."""
        if need_owner:
            self._ensure_owner()
        return _lib.Medium(self.W, self.H, _lib.DIE_F32 if self.dtype == torch.float32 else _lib.DIE_F16, self.epoch,
                           _ptr(self.owner), _ptr(self.food), _ptr(self.chem), _ptr(self.chem_next),
                           *(self.world or (0, 0, 0, 0)), *(getattr(self, 'own', None) or (0, 0, 0, 0)),
                           _ptr(self.sense_mask) if getattr(self, 'sense_mask', None) is not None else None)

    def next_epoch(self):
        """Advance the ownership epoch; zero the plane when the 5-bit tag (1..31) wraps."""
        self.owner_stale = None          # whoever advances the epoch re-claims every occupied cell
        self.epoch += 1
        if self.epoch > _lib.OWNER_EPOCH_MAX:
            self.owner.zero_()
            self.epoch = 1

    def swap_chem(self):
        self.chem, self.chem_next = self.chem_next, self.chem

    def observed_numpy(self) -> np.ndarray:
        """What the agents see (core/env.py:292-295): the medium with the cells outside the sense mask zeroed.
        Without Dynamics.apply_sense_mask this is to_numpy()."""
        m = self.to_numpy()
        mask = getattr(self, 'sense_mask', None)
        return m if mask is None else np.where(mask.cpu().numpy().astype(bool), m, 0.)

    def occupied(self) -> torch.Tensor:
        """Boolean (W, H): the 'agents' channel > 0."""
        self._ensure_owner()
        return ((self.owner >> (32 + _lib.OWNER_EPOCH_SHIFT)) & _lib.OWNER_EPOCH_MAX) == self.epoch

    def owner_slots(self) -> torch.Tensor:
        """int64 (W, H): owning slot id, −1 for empty cells."""
        self._ensure_owner()
        w = (self.owner >> 32) & 0xFFFFFFFF
        return torch.where((w >> _lib.OWNER_EPOCH_SHIFT) == self.epoch, (w & _lib.OWNER_SLOT_MASK) - 1,
                           torch.full_like(w, -1))

    def sensed(self):
        """Somebody is about to read these planes as an observation (see `before_sense`)."""
        hook = self.before_sense
        if hook is not None:
            hook()
            return True                  # (the caller's view of the agents may be stale now: arrays re-seated, another count)
        return False

    def sel(self, channel: str) -> torch.Tensor:
        # (No `sensed()` here: on a decomposed rank the hook is a COLLECTIVE ghost refresh, and sel() / to_numpy() are what a
        # rank-local observer calls — `if rank == 0: env.medium.to_numpy()` must not start an exchange its peers do not join.
        # The cells a rank owns are right without a refresh; whoever wants fresh halos calls DistEnv.flush_refresh() on every
        # rank, as gather_world(), the stand-alone forward and the NCA sensing do.  ADVICE r5.)
        if channel == 'agents':
            return self.occupied().to(torch.float32)
        if channel == 'env_food':
            return self.food
        if channel == 'chem1':
            return self.chem
        raise KeyError(channel)

    def to_numpy(self) -> np.ndarray:
        return np.stack([self.sel(c).to(torch.float64).cpu().numpy() for c in self.channels])

    def upload(self, medium: np.ndarray):
        """Load a (3, W, H) array; occupied cells get an ownership word of the current epoch."""
        medium = np.asarray(medium)
        assert medium.shape == self.shape, (medium.shape, self.shape)
        self.owner_stale = None
        occ = medium[0] > 0
        words = np.where(occ, np.uint64(((self.epoch << _lib.OWNER_EPOCH_SHIFT) | 1) << 32), np.uint64(0)).astype(np.uint64)
        self.owner.copy_(torch.from_numpy(words.view(np.int64)))
        self.food.copy_(torch.from_numpy(np.ascontiguousarray(medium[1], dtype=np.float32)))
        self.chem.copy_(torch.from_numpy(np.ascontiguousarray(medium[2], dtype=np.float32)))

    def upload_channel(self, channel: str, data: np.ndarray):
        t = {'env_food': self.food, 'chem1': self.chem}[channel]
        t.copy_(torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)))


def unpermute(values: torch.Tensor, slot: Optional[torch.Tensor]) -> torch.Tensor:
    """Array order → reference slot order along the last axis."""
    if slot is None:
        return values
    out = torch.empty_like(values)
    out[..., slot.to(torch.int64)] = values
    return out


class DeviceAgents:
    """(4, N) agent array: channels ('x', 'y', 'alive', 'agent_food'); x, y are Q0.32.

    The arrays may be held in a spatially sorted order (`slot[j]` = reference slot id of array
    entry j, None = identity); everything a caller sees (`sel`, `to_numpy`) is in slot order."""
    channels = DataChannels.agents

    def __init__(self, num_slots: int, device):
        N = int(num_slots)
        self.N, self.device = N, torch.device(device)
        self.x = torch.zeros(N, dtype=torch.int32, device=device)
        self.y = torch.zeros(N, dtype=torch.int32, device=device)
        self.alive = torch.zeros(N, dtype=torch.uint8, device=device)
        self.agent_food = torch.zeros(N, dtype=torch.float32, device=device)
        self.slot: Optional[torch.Tensor] = None
        self.global_slots = False    # decomposed world: `slot` holds world slot ids, N <= capacity of the arrays
        self._attached = []          # weak references to objects holding per-slot state (Agent objects)

    @property
    def shape(self):
        return (4, self.N)

    def c_struct(self) -> _lib.Agents:
        return _lib.Agents(self.N, _ptr(self.x), _ptr(self.y), _ptr(self.alive), _ptr(self.agent_food), _ptr(self.slot))

    def attach(self, obj):
        """Register an object whose per-slot arrays must follow re-orderings (see Env.sort_agents)."""
        import weakref
        if not any(r() is obj for r in self._attached):
            self._attached.append(weakref.ref(obj))

    def attached(self):
        live = [r() for r in self._attached]
        self._attached = [r for r, o in zip(self._attached, live) if o is not None]
        return [o for o in live if o is not None]

    def sel(self, channel: str) -> torch.Tensor:
        if channel in ('x', 'y'):
            q = getattr(self, channel).to(torch.int64) & 0xFFFFFFFF
            v = q.to(torch.float64) / Q32
        elif channel == 'alive':
            v = self.alive.to(torch.float32)
        elif channel == 'agent_food':
            v = self.agent_food
        else:
            raise KeyError(channel)
        if self.global_slots:
            return v[:self.N]                       # local agents in array order (see DistEnv.gather_world)
        return unpermute(v, self.slot)

    def to_numpy(self) -> np.ndarray:
        return np.stack([self.sel(c).to(torch.float64).cpu().numpy() for c in self.channels])

    @property
    def capacity(self) -> int:
        return int(self.x.numel())

    def q32_numpy(self):
        """Raw coordinate words as uint32 arrays, slot order."""
        return (unpermute(self.x, self.slot).cpu().numpy().view(np.uint32),
                unpermute(self.y, self.slot).cpu().numpy().view(np.uint32))

    def upload(self, agents: np.ndarray):
        agents = np.asarray(agents, dtype=np.float64)
        assert agents.shape == self.shape, (agents.shape, self.shape)
        self.slot = None
        self.x.copy_(torch.from_numpy(to_q32(agents[0]).view(np.int32)))
        self.y.copy_(torch.from_numpy(to_q32(agents[1]).view(np.int32)))
        self.alive.copy_(torch.from_numpy((agents[2] > 0).astype(np.uint8)))
        self.agent_food.copy_(torch.from_numpy(agents[3].astype(np.float32)))


class DeviceAction:
    """(3, N) action array: channels ('dx', 'dy', 'deposit1'), in the array order of the agents it
    was computed for (`slot` as in DeviceAgents); `to_numpy` / `sel` give slot order."""
    channels = DataChannels.actions

    def __init__(self, num_slots: int, device, slot: Optional[torch.Tensor] = None, capacity: Optional[int] = None):
        N = int(num_slots)
        self.N, self.device = N, torch.device(device)
        self.data = torch.empty((3, int(capacity or N)), dtype=torch.float32, device=device)
        self.slot = slot
        self.global_slots = False

    @property
    def shape(self):
        return (3, self.N)

    def c_struct(self) -> _lib.Action:
        return _lib.Action(self.N, _ptr(self.data[0]), _ptr(self.data[1]), _ptr(self.data[2]))

    def sel(self, channel: str) -> torch.Tensor:
        row = self.data[self.channels.index(channel)]
        return row[:self.N] if self.global_slots else unpermute(row, self.slot)

    def to_numpy(self) -> np.ndarray:
        if self.global_slots:
            return self.data[:, :self.N].to(torch.float64).cpu().numpy()
        return unpermute(self.data, self.slot).to(torch.float64).cpu().numpy()

    def in_order_of(self, slot: Optional[torch.Tensor]) -> 'DeviceAction':
        """This action re-ordered for agents held in order `slot` (no copy when already so)."""
        if self.slot is slot:
            return self
        out = DeviceAction(self.N, self.device, slot)
        v = unpermute(self.data, self.slot)
        out.data.copy_(v if slot is None else v[:, slot.to(torch.int64)])
        return out

    @staticmethod
    def from_numpy(action: np.ndarray, device, slot: Optional[torch.Tensor] = None) -> 'DeviceAction':
        """Upload a (3, N) array given in slot order."""
        action = np.asarray(action)
        assert action.ndim == 2 and action.shape[0] == 3, action.shape
        a = DeviceAction(action.shape[1], device, None)
        a.data.copy_(torch.from_numpy(np.ascontiguousarray(action, dtype=np.float32)))
        return a.in_order_of(slot)


class PendingAction(DeviceAction):
    """The action of a GradientAgent/PhysarumAgent.forward() call whose kernel has not run yet.

    `Env.step` recognises it and runs forward fused with the first half of the step
    (die_forward_env_step); any other use — `to_numpy()`, `sel()`, handing it to another Env, a second
    `forward()` on the same agent — first runs the stand-alone forward kernel.  Either way the arrays
    hold the same values afterwards."""

    def __init__(self, agent, agents, medium, g_struct, keepalive):
        N = agents.N
        self.N, self.device = N, agents.device
        self._capacity = agents.capacity
        self._buf = None                 # (3, capacity) float32, allocated when somebody needs the values
        self.slot = agents.slot
        self.global_slots = agents.global_slots
        self.agent, self.agents, self.medium = agent, agents, medium
        self.g_struct, self._keepalive = g_struct, keepalive
        self.pending = True
        self._rebuild = None             # set by the tile-binned step when it kept the action in registers (die_amd/pic.py)

    @property
    def _data(self):
        if self._buf is None:
            self._buf = torch.empty((3, self._capacity), dtype=torch.float32, device=self.device)
        return self._buf

    @property
    def data(self):
        self.ensure()
        return self._data

    def raw_struct(self) -> _lib.Action:
        """Pointers of the (possibly still unfilled) arrays, without forcing the forward kernel."""
        base, row = self._data.data_ptr(), self._data.stride(0) * 4          # (three tensor slices cost 4 µs)
        return _lib.Action(self.N, base, base + row, base + 2 * row)

    def rebind(self, agents):
        """The agents (and the agent object's state) were re-ordered between forward() and the fused step."""
        self.slot = agents.slot
        hi, lo = self.agent._hd_hi, self.agent._hd_lo
        self.g_struct.heading_hi, self.g_struct.heading_lo = hi.data_ptr(), lo.data_ptr()
        self._keepalive = (hi, lo) + tuple(self._keepalive[2:])

    def in_order_of(self, slot):
        self.ensure()                    # (a lazily rebuilt action also learns its array order there)
        return super().in_order_of(slot)

    def ensure(self):
        if self.pending:
            self.pending = False
            self.agent._run_forward(self)
        elif self._rebuild is not None:          # consumed by a step that did not store it: re-derive it now
            rebuild, self._rebuild = self._rebuild, None
            rebuild(self)

    def done(self):
        self.pending = False
