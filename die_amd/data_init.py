"""Field / agent allocation on device — the DataInitializer of the reference
(core/data_init.py:92-253) as Env._init_data uses it (core/env.py:74-86).

The reference fills env_food with Perlin noise from the un-vendored `perlin_noise` package
in an O(W·H) Python loop (:190-196); here the same construction (2-D gradient noise, `octaves`
lattice cells per unit length, `.round(3)`, `_mask`) runs on the device on a Philox-seeded
lattice (`die_perlin2` in csrc/die_rng.h).  Agent seeding (`with_agents`, :222-226), `agents_from_medium`
(:133-150) and `init_action_for` (:152-157) keep the reference semantics.
"""
import ctypes as C
import math
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .device_array import DeviceAction, DeviceAgents, DeviceMedium, _ptr, stream_ptr

_STREAM_INIT_FOOD = 4


def _philox4x32_10(c, k):
    """Scalar Philox4x32-10 (host side only needs a handful of draws for the food spec)."""
    c = list(c)
    k0, k1 = k
    for _ in range(10):
        p0 = 0xD2511F53 * c[0]
        p1 = 0xCD9E8D57 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k0, p1 & 0xFFFFFFFF, (p0 >> 32) ^ c[3] ^ k1, p0 & 0xFFFFFFFF]
        k0 = (k0 + 0x9E3779B9) & 0xFFFFFFFF
        k1 = (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c


def food_spec_from_seed(seed: int, n_waves: int = 6, max_freq: int = 5, scale: float = 0.5, perlin_octaves: int = 0,
                        threshold: float = 1.0) -> _lib.FoodSpec:
    """`perlin_octaves > 0`: the Perlin food of Env._init_data (core/env.py:75-79: threshold 1.0, octaves 8); else the
    sinusoid mix of round 1 (kept for comparisons)."""
    spec = _lib.FoodSpec()
    spec.n_waves = n_waves
    spec.scale = scale
    spec.perlin_octaves = int(perlin_octaves)
    spec.threshold = float(threshold)
    amps = []
    for i in range(n_waves):
        r = _philox4x32_10((i, 0, 0, _STREAM_INIT_FOOD), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
        spec.fx[i] = float(r[0] % (2 * max_freq + 1) - max_freq)
        spec.fy[i] = float(r[1] % max_freq + 1)
        spec.phase[i] = r[2] * (2 * math.pi / 4294967296.0)
        amps.append(0.5 + r[3] / 4294967296.0)
    tot = sum(amps)
    for i in range(n_waves):
        spec.amp[i] = amps[i] / tot
    return spec


class DeviceFoodFlow:
    """The operator returned by `FieldSequence.get_flow_operator` (core/data_init.py:29-38) for the sequences whose field is
    evaluated on the device (WaveSequence, PerlinNoiseSequence): called by `Env.step` as `op_food_flow`; recognised there
    and applied in HBM without a host round trip."""

    def __init__(self, seq: 'FieldSequence', scale: float, decay: float):
        self.seq, self.scale, self.decay = seq, float(scale), float(decay)
        self._k = 0

    def next_t(self) -> float:
        t = float(self.seq._ts[self._k % len(self.seq._ts)])          # `cycle(self._ts)` of the reference
        self._k += 1
        return t

    def apply(self, medium: DeviceMedium):
        self.seq._flow(medium, self.next_t(), self.scale, self.decay)

    def __call__(self, current):
        """Host arrays (the reference's calling convention): same arithmetic, evaluated on the device."""
        cur = np.asarray(current, dtype=np.float64)
        dev = torch.device('cuda', torch.cuda.current_device())
        m = DeviceMedium(cur.shape, dev, torch.float32)
        m.upload_channel('env_food', cur)
        self.apply(m)
        return m.food.to(torch.float64).cpu().numpy()


WaveFoodFlow = DeviceFoodFlow          # (round-1 name)


class FieldSequence:
    """core/data_init.py:16-52: a time sequence of fields over arange(*t_bounds, dt), cycled by `__iter__`;
    `get_flow_operator(scale, decay)` gives food ← scale·next(field) + (1 − decay)·food.  Subclasses that define `_flow`
    (the same update as one kernel on a DeviceMedium) run inside `Env.step` on the device; a subclass that only defines
    `__getitem__` (numpy) works too, through the host round trip `Env` gives any Python operator."""

    def __init__(self, field_size: Sequence[int], dt: float = 0.01, t_bounds: Tuple[float, float] = (0, 10)):
        self._size = tuple(int(v) for v in field_size)
        self._tbounds = t_bounds
        self._ts = np.arange(*t_bounds, dt)

    _flow = None

    def get_flow_operator(self, scale: float = 1.0, decay: float = 0.0):
        if self._flow is not None:
            return DeviceFoodFlow(self, scale, decay)
        it = iter(self)

        def food_flow(current):
            return scale * next(it) + (1 - decay) * current
        return food_flow

    def __iter__(self):
        from itertools import cycle
        for t in cycle(self._ts):
            yield self[t]

    def __len__(self):
        return len(self._ts)

    def __contains__(self, t: float):
        t0, t_end = self._tbounds
        return t0 <= t < t_end

    def __getitem__(self, t: float) -> np.ndarray:
        if self._flow is None:
            raise NotImplementedError
        dev = torch.device('cuda', torch.cuda.current_device())         # the field itself: the update applied to zeros
        m = DeviceMedium(self._size, dev, torch.float32)
        m.food.zero_()
        self._flow(m, float(t), 1.0, 0.0)
        return m.food.to(torch.float64).cpu().numpy()


class WaveSequence(FieldSequence):
    """core/data_init.py:71-89: running waves + moving islands (`die_food_flow_wave`)."""

    def _flow(self, medium: DeviceMedium, t: float, scale: float, decay: float):
        _lib.check(_lib.lib.die_food_flow_wave(C.byref(medium.c_struct(need_owner=False)), t, scale, decay, stream_ptr(medium.device)), 'die_food_flow_wave')


class PerlinNoiseSequence(FieldSequence):
    """core/data_init.py:55-69: `PerlinNoise(octaves)((x, y, t))` on the cell labels, `.round(3)` (`die_food_flow_perlin`).
    The un-vendored `perlin_noise` package is seeded from the global RNG; here the lattice gradients come from `seed`."""

    def __init__(self, field_size: Sequence[int], dt: float = 0.01, t_bounds: Tuple[float, float] = (0, 1), octaves: int = 8,
                 seed: int = 0):
        super().__init__(field_size, dt, t_bounds)
        self._octaves, self._seed = int(octaves), int(seed)

    def _flow(self, medium: DeviceMedium, t: float, scale: float, decay: float):
        _lib.check(_lib.lib.die_food_flow_perlin(C.byref(medium.c_struct(need_owner=False)), t, self._octaves, scale, decay,
                                                 self._seed & 0xFFFFFFFFFFFFFFFF, stream_ptr(medium.device)), 'die_food_flow_perlin')


class DataInitializer:
    """core/data_init.py:92-253 with the arrays in HBM: the static allocators of the reference under their names, and its
    builder — `DataInitializer(field_size, channels).with_const(...).with_noise(...).with_agents(...)
    .with_food_perlin(...).with_chem(...).build()`, `DataInitializer.action_for(agents).with_noise(...).build_agents()` —
    every step one launch of `die_field_fill` on a float32 device array.  Random steps draw from Philox(seed, step) where the
    reference uses the unseeded global numpy generator; the Perlin field is the same construction as the un-vendored
    `perlin_noise` package (gradient noise on `octaves` lattice cells per unit length) on a Philox-seeded lattice."""

    def __init__(self, field_size, channels: Optional[Sequence[str]] = None, name: Optional[str] = None, mask=1., *,
                 seed: int = 0, device=None):
        self._size = (int(field_size),) if np.isscalar(field_size) else tuple(int(v) for v in field_size)
        self._n = int(np.prod(self._size))
        self._device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        self._channels = {chan: torch.zeros(self._n, dtype=torch.float32, device=self._device) for chan in channels or ()}
        self._name = name
        self._static_mask = mask
        self._seed = int(seed)
        self._draws = {}                         # random calls made so far, per kind (the Philox step word)

    def _fill(self, channel: str, op: int, a: float = 0., b: float = 0., kind: str = ''):
        dst = self._channels.get(channel)
        if dst is None:
            dst = self._channels[channel] = torch.empty(self._n, dtype=torch.float32, device=self._device)
        step = self._draws.get(kind, 0)
        if kind:
            self._draws[kind] = step + 1
        W, H = (self._size + (1,))[:2]
        _lib.check(_lib.lib.die_field_fill(_ptr(dst), self._n, op, W, H, float(a), float(b), self._seed & 0xFFFFFFFFFFFFFFFF,
                                           step // 4 if kind == 'noise' else step, step % 4 if kind == 'noise' else 0,
                                           stream_ptr(self._device)), 'die_field_fill')
        return self

    def with_const(self, channel: str, value=0.):
        """:214-216."""
        return self._fill(channel, _lib.DIE_FIELD_CONST, value)

    def with_noise(self, channel: str, a=0, b=1):
        """:218-220 / get_random (:168-169): (b − a)·random_sample().round(3) + a."""
        return self._fill(channel, _lib.DIE_FIELD_NOISE, a, b, 'noise')

    def with_agents(self, ratio: float):
        """:222-226: ceil(_mask(random_sample().round(3), mask_above=ratio))."""
        return self._fill('agents', _lib.DIE_FIELD_AGENTS, 0., ratio, 'agents')

    def with_food_perlin(self, threshold: float = 0.25, octaves: int = 8):
        """:228-231."""
        return self._fill('env_food', _lib.DIE_FIELD_PERLIN, octaves, threshold, 'perlin')

    def with_chem(self, threshold: float = 0.1):
        """:233-236 (24 octaves)."""
        return self._fill('chem1', _lib.DIE_FIELD_PERLIN, 24, threshold, 'perlin')

    def _mask_tensor(self):
        m = self._static_mask
        if isinstance(m, torch.Tensor):
            return m.to(device=self._device, dtype=torch.float32).reshape(-1).contiguous(), 1.0
        if np.isscalar(m):
            return None, float(m)
        return torch.from_numpy(np.ascontiguousarray(m, dtype=np.float32)).to(self._device).reshape(-1), 1.0

    def build_numpy(self) -> np.ndarray:
        """:238-239 (host copy of the channels, unmasked)."""
        return np.stack([t.reshape(self._size).to(torch.float64).cpu().numpy() for t in self._channels.values()])

    def build(self, name: Optional[str] = None, dtype=torch.float32) -> DeviceMedium:
        """:241-246 for the medium channels → a DeviceMedium (occupied cells flagged; `agents_from_medium` seats them)."""
        from .base_types import DataChannels
        extra = set(self._channels) - set(DataChannels.medium)
        if extra or len(self._size) != 2:
            raise ValueError(f'build() makes the (agents, env_food, chem1) medium; got channels {sorted(self._channels)} of shape {self._size}')
        medium = DeviceMedium(self._size, self._device, dtype)
        mask, scalar = self._mask_tensor()
        ch = [self._channels.get(c) for c in DataChannels.medium]
        _lib.check(_lib.lib.die_medium_from_fields(C.byref(medium.c_struct(need_owner=False)), _ptr(ch[0]), _ptr(ch[1]), _ptr(ch[2]),
                                                   _ptr(mask), scalar, stream_ptr(self._device)), 'die_medium_from_fields')
        return medium

    def build_agents(self, name: Optional[str] = None) -> DeviceAction:
        """:248-253 for the action channels (what BrownianAgent.forward builds, core/agent/static.py:40-50)."""
        from .base_types import DataChannels
        if tuple(self._channels) != tuple(DataChannels.actions):
            raise ValueError(f'build_agents() makes a (dx, dy, deposit1) action; got channels {list(self._channels)}')
        act = DeviceAction(self._n, self._device)
        mask, scalar = self._mask_tensor()
        for i, c in enumerate(DataChannels.actions):
            torch.mul(self._channels[c], mask if mask is not None else scalar, out=act.data[i])
        return act

    @staticmethod
    def action_for(agents: DeviceAgents, seed: int = 0) -> 'DataInitializer':
        """:159-165: an action builder masked by `alive` (slot order)."""
        from .base_types import DataChannels
        return DataInitializer(field_size=agents.N, channels=DataChannels.actions, name='actions', mask=agents.sel('alive'),
                               seed=seed, device=agents.device)

    @staticmethod
    def get_random(size, a=0., b=1., seed: int = 0) -> np.ndarray:
        """:167-169 (host array)."""
        d = DataInitializer(field_size=size, channels=('v',), seed=seed)
        return d.with_noise('v', a, b).build_numpy()[0]

    @staticmethod
    def init_field_array(field_size: Tuple[int, int], device='cuda:0', dtype=torch.float32) -> DeviceMedium:
        """core/data_init.py:94-112 (zero-filled 3-channel field)."""
        return DeviceMedium(field_size, device, dtype)

    @staticmethod
    def init_agential_array(num_agents: int, device='cuda:0') -> DeviceAgents:
        """core/data_init.py:114-130."""
        return DeviceAgents(num_agents, device)

    @staticmethod
    def init_action_for(agents: DeviceAgents, init_data: float = 0.) -> DeviceAction:
        """core/data_init.py:152-157."""
        act = DeviceAction(agents.N, agents.device)
        act.data.fill_(init_data)
        return act

    @staticmethod
    def workspace(field_size, num_slots: int, device) -> torch.Tensor:
        nbytes = _lib.lib.die_workspace_bytes(int(field_size[0]), int(field_size[1]), int(num_slots))
        if nbytes < 0:
            raise ValueError(f'bad problem size {field_size}, {num_slots}')
        return torch.empty(nbytes, dtype=torch.uint8, device=device)

    @staticmethod
    def init_medium(medium: DeviceMedium, agent_ratio: float, seed: int, food_scale: float = 0.5, food: str = 'perlin'):
        """Env._init_data (core/env.py:75-79) in one launch: with_const('env_food', .5) → overwritten by
        with_food_perlin(threshold=1.0, octaves=8) → with_agents(ratio) → build.  `food='waves'`: the sinusoid mix of
        round 1 instead of the Perlin field."""
        spec = food_spec_from_seed(seed, scale=food_scale, perlin_octaves=8 if food == 'perlin' else 0, threshold=1.0)
        medium.epoch = 1
        m = medium.c_struct()
        _lib.check(_lib.lib.die_init_medium(C.byref(m), float(agent_ratio), seed & 0xFFFFFFFFFFFFFFFF, C.byref(spec),
                                            stream_ptr(medium.device)), 'die_init_medium')

    @staticmethod
    def agents_from_medium(medium: DeviceMedium, max_agents: Optional[int], seed: int,
                           workspace: Optional[torch.Tensor] = None) -> Tuple[DeviceAgents, int]:
        """core/data_init.py:133-150.  `max_agents=None` → W·H slots like the reference;
        `max_agents='alive'` → exactly as many slots as seeded agents (no dead slots).
        Returns (agents, K)."""
        dev = medium.device
        if max_agents == 'alive':
            n_slots = int(medium.owner.ne(0).sum().item())
            n_slots = max(n_slots, 1)
        else:
            n_slots = int(max_agents) if max_agents else medium.W * medium.H
        agents = DeviceAgents(n_slots, dev)
        ws = workspace if workspace is not None and workspace.numel() >= _lib.lib.die_workspace_bytes(
            medium.W, medium.H, n_slots) else DataInitializer.workspace((medium.W, medium.H), n_slots, dev)
        count = torch.zeros(2, dtype=torch.int64, device=dev)
        m, a = medium.c_struct(), agents.c_struct()
        _lib.check(_lib.lib.die_init_agents(C.byref(m), C.byref(a), seed & 0xFFFFFFFFFFFFFFFF, _ptr(count), _ptr(ws),
                                            ws.numel(), stream_ptr(dev)), 'die_init_agents')
        k, overflow = (int(v) for v in count.cpu())
        if overflow:
            raise ValueError(f'max_agents={n_slots} is smaller than the number of seeded agents')
        return agents, k
