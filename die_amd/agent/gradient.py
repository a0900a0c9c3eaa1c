"""GradientAgent / PhysarumAgent (reference core/agent/gradient.py:13-219): constructor
signatures and `forward(obs) -> action` kept; the whole of `forward` is one launch of
`die_gradient_forward` and the per-agent state (`_direction_rads`, `_prev_grad`) lives in HBM.
"""
import ctypes as C
import math
import os
from typing import Any, Dict, Optional, Sequence

import numpy as np
import torch

from .. import _lib
from ..device_array import DeviceAction, PendingAction, _ptr, stream_ptr
from .base import Agent, save_args


def split64(t64: torch.Tensor):
    """float64 (N,) → (hi, lo) int32 (N,) tensors: the two halves of every value (die_gradient_agent.heading_hi / _lo)."""
    v = t64.to(torch.float64).contiguous().view(torch.int32).reshape(-1, 2)       # little endian: [lo, hi]
    return v[:, 1].contiguous(), v[:, 0].contiguous()


def join64(hi: torch.Tensor, lo: torch.Tensor) -> torch.Tensor:
    return torch.stack([lo, hi], dim=1).contiguous().view(torch.float64).reshape(-1)


class GradientAgent(Agent):
    _kind = _lib.DIE_AGENT_GRADIENT

    def __init__(self,
                 max_agents: int = 10**6,
                 scale: float = 0.01,
                 deposit: float = 4.0,
                 inertia: float = 0.9,
                 sense_offset: float = 0.,
                 noise_scale: float = 0.025,
                 normalized_grad: bool = True,
                 grad_clip: Optional[float] = 1e-5,
                 seed: Optional[int] = None,
                 ):
        self._init_params = save_args(self.__init__, locals())
        self._size = int(max_agents)
        self._seed = int.from_bytes(os.urandom(8), 'little') if seed is None else int(seed)
        self._noise_scale = noise_scale
        self._scale = scale
        self._deposit = deposit
        self._inertia = inertia
        self._sense_offset_scale = sense_offset
        self._normalized = normalized_grad
        self._grad_clip = grad_clip
        self._turn_radians = 0.0
        self._sense_radians = 0.0
        self._rtol = 0.0
        self._calls = 0
        # `_direction_rads` is float64 like the reference's; on the device it is held as two int32 arrays (high / low
        # halves), which lets it travel through the 4-byte sort / migration machinery.  Allocated on first forward.
        self._hd_hi: Optional[torch.Tensor] = None
        self._hd_lo: Optional[torch.Tensor] = None
        self._prev_grad: Optional[torch.Tensor] = None
        self._turn_sign: Optional[torch.Tensor] = None            # test hook: per-slot ±1 instead of Philox
        self._order: Optional[torch.Tensor] = None                # slot tensor the state arrays are aligned with
        self._pending = None                                      # forward() whose kernel has not run yet
        self._step_base: Optional[torch.Tensor] = None            # device word added to the step counter (Env.run graphs)
        self.lazy = True                                          # let Env.step fuse forward with the step

    @property
    def init_params(self) -> Dict[str, Any]:
        return self._init_params

    @property
    def _direction_rads(self) -> Optional[torch.Tensor]:
        """The headings as one float64 tensor in the current array order (read-only view of the two halves)."""
        return None if self._hd_hi is None else join64(self._hd_hi, self._hd_lo)

    def _alloc_state(self, device):
        """__init__ state of the reference (:42-43,163): heading from N(0, .4) noise."""
        self._hd_hi = torch.empty(self._size, dtype=torch.int32, device=device)
        self._hd_lo = torch.empty(self._size, dtype=torch.int32, device=device)
        if self._inertia != 0:
            self._prev_grad = torch.empty((2, self._size), dtype=torch.float32, device=device)
        pg = self._prev_grad
        _lib.check(_lib.lib.die_init_heading(_ptr(self._hd_hi), _ptr(self._hd_lo), _ptr(pg[0]) if pg is not None else None,
                                             _ptr(pg[1]) if pg is not None else None, self._size,
                                             float(self._turn_radians), self._seed & 0xFFFFFFFFFFFFFFFF,
                                             stream_ptr(device)), 'die_init_heading')

    # -- per-slot state follows the agents' array order (Env.sort_agents) ---------------------
    def _die_state_tensors(self, agents):
        if self._hd_hi is None or self._order is not agents.slot:
            return []
        ts = [self._hd_hi, self._hd_lo]
        if self._prev_grad is not None:
            ts += [self._prev_grad[0], self._prev_grad[1]]
        return ts

    def _die_state_permuted(self, tensors, slot):
        self._hd_hi, self._hd_lo = tensors[0], tensors[1]
        if self._prev_grad is not None:
            self._prev_grad = torch.stack([tensors[2], tensors[3]])
        self._order = slot

    def _align_to(self, slot):
        """Bring the state arrays from order `self._order` to order `slot` (rare path)."""
        from ..device_array import unpermute
        def move(t):
            v = unpermute(t, self._order)
            return v if slot is None else v[..., slot.to(torch.int64)].contiguous()
        self._hd_hi, self._hd_lo = move(self._hd_hi), move(self._hd_lo)
        if self._prev_grad is not None:
            self._prev_grad = move(self._prev_grad)
        self._order = slot

    def direction_rads_numpy(self) -> np.ndarray:
        """`_direction_rads` in slot order (float64)."""
        from ..device_array import unpermute
        return unpermute(self._direction_rads, self._order).cpu().numpy()

    def prev_grad_numpy(self) -> np.ndarray:
        from ..device_array import unpermute
        return unpermute(self._prev_grad, self._order).to(torch.float64).cpu().numpy()

    def set_state_local(self, agents, direction_rads: torch.Tensor, prev_grad: Optional[torch.Tensor] = None):
        """Decomposed world: state already in the local array order of `agents` (capacity-sized; any float dtype)."""
        self._hd_hi, self._hd_lo = split64(direction_rads)
        self._prev_grad = prev_grad
        self._order = agents.slot

    def set_state(self, direction_rads: np.ndarray, prev_grad: Optional[np.ndarray] = None, device='cuda:0'):
        """Load `_direction_rads` (and `_prev_grad`) from host arrays given in slot order."""
        self._order = None
        self._hd_hi, self._hd_lo = split64(torch.from_numpy(np.ascontiguousarray(direction_rads, dtype=np.float64)).to(device))
        assert self._hd_hi.numel() == self._size
        if prev_grad is not None:
            self._prev_grad = torch.from_numpy(np.ascontiguousarray(prev_grad, dtype=np.float32)).to(device)
        elif self._inertia != 0:
            self._prev_grad = torch.zeros((2, self._size), dtype=torch.float32, device=device)

    def set_turn_signs(self, signs: Optional[np.ndarray], device='cuda:0'):
        """Replace the random ±1 turn choice (core/agent/gradient.py:183) by given values."""
        self._turn_sign = None if signs is None else torch.from_numpy(np.asarray(signs).astype(np.int8)).to(device)

    def forward(self, obs) -> DeviceAction:
        """:96-124."""
        agents, medium = obs
        if agents.capacity != self._size:
            raise ValueError(f'agent built for max_agents={self._size}, observation has {agents.capacity} slots')
        dev = agents.device
        if self._hd_hi is None:
            self._alloc_state(dev)
        if self._order is not agents.slot:
            if agents.global_slots:                # decomposed world: state is kept in local array order
                self._order = agents.slot
            else:
                self._align_to(agents.slot)
        agents.attach(self)
        self._render_src = (medium, medium.chem)    # what render() draws the gradient field of
        if self._pending is not None:               # keep heading updates in call order
            self._pending.ensure()
        pg = self._prev_grad
        g = _lib.GradientAgent(
            self._kind, int(bool(self._normalized)), self._scale, self._deposit, self._inertia,
            self._sense_offset_scale, self._noise_scale, -1.0 if self._grad_clip is None else self._grad_clip,
            self._turn_radians, self._sense_radians, self._rtol, _ptr(self._hd_hi), _ptr(self._hd_lo),
            _ptr(pg[0]) if pg is not None else None, _ptr(pg[1]) if pg is not None else None,
            _ptr(self._turn_sign), self._seed & 0xFFFFFFFFFFFFFFFF, self._calls & 0xFFFFFFFF, 0,
            _ptr(self._step_base))
        self._calls += 1
        action = PendingAction(self, agents, medium, g, (self._hd_hi, self._hd_lo, pg, self._turn_sign))
        self._pending = action
        if not self.lazy:
            action.ensure()
        return action

    def _run_forward(self, action):
        """Launch the stand-alone forward kernel for a pending action."""
        self._flush_lazy()                          # (it rewrites the headings an un-read lazy action is derived from)
        if self._pending is action:
            self._pending = None
        agents, medium = action.agents, action.medium
        if medium.sensed():                         # (a decomposed rank: a ghost refresh that was left for the next step happens now …
            action.rebind(agents)                   #  … and re-seats the agents: the action follows — array order, headings, count)
            action.N = agents.N
        m, a, u = medium.c_struct(), agents.c_struct(), action.raw_struct()
        _lib.check(_lib.lib.die_gradient_forward(C.byref(m), C.byref(a), C.byref(action.g_struct), C.byref(u),
                                                 stream_ptr(agents.device)), 'die_gradient_forward')

    def _flush_lazy(self):
        """Fill in the action of the last tile-binned step if somebody still holds it un-read (die_amd/pic.py)."""
        ref = getattr(self, '_lazy_action', None)
        act = ref() if ref is not None else None
        self._lazy_action = None
        if act is not None:
            act.ensure()

    def _forward_consumed(self, action):
        """Env.step ran this action's forward inside die_forward_env_step."""
        if self._pending is action:
            self._pending = None
        action.done()

    def render(self) -> Sequence[np.ndarray]:
        """:126-135: before the first forward one white pixel; afterwards the gradient field the agents sense,
        0.5·(stack(gx, gy, 0) + 1) as a (W, H, 3) image — computed when asked for (die_gradient_render) from the chem
        plane the last forward() saw (the fused step leaves that plane untouched as the medium's spare one; two steps
        later it has been overwritten and the image is that of a newer field)."""
        src = getattr(self, '_render_src', None)
        if src is None:
            return [np.ones((1, 1, 3))]
        medium, chem = src
        rgb = torch.empty((medium.W, medium.H, 3), dtype=torch.float32, device=chem.device)
        _lib.check(_lib.lib.die_gradient_render(_ptr(chem), medium.W, medium.H, _lib.DIE_F32 if chem.dtype == torch.float32 else _lib.DIE_F16,
                                                int(bool(self._normalized)), -1.0 if self._grad_clip is None else self._grad_clip,
                                                _ptr(rgb), stream_ptr(chem.device)), 'die_gradient_render')
        return [rgb.cpu().numpy()]


class PhysarumAgent(GradientAgent):
    _kind = _lib.DIE_AGENT_PHYSARUM

    def __init__(self,
                 max_agents: int = 10**6,
                 scale: float = 0.005,
                 deposit: float = 4.0,
                 inertia: float = 0.0,
                 sense_offset: float = 0.03,
                 noise_scale: float = 0.0,
                 normalized_grad: bool = True,
                 grad_clip: Optional[float] = 1e-5,
                 turn_angle: int = 30,
                 sense_angle: int = 90,
                 turn_tolerance: float = 0.1,
                 seed: Optional[int] = None,
                 ):
        super().__init__(max_agents, scale, deposit, inertia, sense_offset, noise_scale, normalized_grad, grad_clip,
                         seed)
        self._init_params = save_args(self.__init__, locals())
        self._turn_radians = math.radians(turn_angle)
        self._sense_radians = math.radians(sense_angle)
        self._rtol = turn_tolerance
