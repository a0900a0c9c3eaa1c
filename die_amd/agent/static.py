"""ConstAgent / BrownianAgent (reference core/agent/static.py:9-51)."""
import ctypes as C
import os
from typing import Any, Dict, Optional, Tuple

from .. import _lib
from ..device_array import DeviceAction, stream_ptr
from .base import Agent, save_args


class ConstAgent(Agent):
    def __init__(self, delta_xy: Tuple[float, float], deposit: float = 0.):
        self._init_params = save_args(self.__init__, locals())
        self._data = (float(delta_xy[0]), float(delta_xy[1]), float(deposit))

    @property
    def init_params(self) -> Dict[str, Any]:
        return self._init_params

    def forward(self, obs) -> DeviceAction:
        agents, _ = obs
        action = DeviceAction(agents.N, agents.device, agents.slot, capacity=agents.capacity)
        action.global_slots = agents.global_slots
        u = action.c_struct()
        _lib.check(_lib.lib.die_const_forward(agents.N, *self._data, C.byref(u), stream_ptr(agents.device)),
                   'die_const_forward')
        return action


class BrownianAgent(Agent):
    def __init__(self, move_scale: float = 0.01, deposit_scale: float = 0.5, seed: Optional[int] = None):
        self._init_params = save_args(self.__init__, locals())
        self._scale = move_scale
        self._dep_scale = deposit_scale
        self._seed = int.from_bytes(os.urandom(8), 'little') if seed is None else int(seed)
        self._calls = 0

    @property
    def init_params(self) -> Dict[str, Any]:
        return self._init_params

    def forward(self, obs) -> DeviceAction:
        agents, _ = obs
        action = DeviceAction(agents.N, agents.device, agents.slot, capacity=agents.capacity)
        action.global_slots = agents.global_slots
        a, u = agents.c_struct(), action.c_struct()
        _lib.check(_lib.lib.die_brownian_forward(C.byref(a), self._scale, self._dep_scale,
                                                 self._seed & 0xFFFFFFFFFFFFFFFF, self._calls & 0xFFFFFFFF,
                                                 C.byref(u), stream_ptr(agents.device)), 'die_brownian_forward')
        self._calls += 1
        return action
