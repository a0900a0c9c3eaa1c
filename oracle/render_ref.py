"""Test infrastructure, not product: the reference's renderer (core/render.py:9-30,76-132) restated on plain numpy
arrays — the checker of the device frames (`die_render_frames`) in tests/test_gpu_parity.py.  `FieldTrace.update` is pinned
by the reference's own function body in tests/golden/ref_helpers.npz (tests/test_oracle_pins.py)."""
from typing import Tuple

import numpy as np


class FieldTrace:
    """core/render.py:9-30: exponentially fading footprint of the agents channel."""

    def __init__(self, field_size: Tuple[int, int], trace_steps: int = 8):
        self._decay = 1 - 1 / trace_steps
        self._trace_field = np.zeros(field_size)

    @property
    def trace(self) -> np.ndarray:
        return self._trace_field

    def update(self, field):
        self._trace_field = self._trace_field * self._decay + field


class EnvRenderer:
    def __init__(self, field_size: Tuple[int, int], is_trace_colored: bool = True):
        self.field_size = field_size
        self._is_trace_colored = is_trace_colored
        self._agent_trace = FieldTrace(field_size)

    def render(self, medium: np.ndarray, agents: np.ndarray):
        return [self._img_medium(medium), self._img_trace(medium), self._img_agents(agents)]

    def _img_medium(self, medium):
        """(W, H, 3): agents, env_food, chem1 as R, G, B (core/render.py:92-101, 'rgb' colours)."""
        return np.stack([medium[0], medium[1], medium[2]], axis=-1)

    def _img_trace(self, medium):
        """core/render.py:103-110."""
        self._agent_trace.update(medium[0])
        t = self._agent_trace.trace
        try:
            import matplotlib
            return matplotlib.colormaps['magma' if self._is_trace_colored else 'gray'](t)
        except ImportError:
            g = np.clip(t, 0, 1)
            return np.stack([g, g, g, np.ones_like(g)], axis=-1)

    def _img_agents(self, agents):
        """core/render.py:112-132: (alive, agent_food) laid out as an image of `height` rows."""
        width, height = self.field_size
        n = agents.shape[1]
        cols = -(-n // height)
        data = np.zeros((2, height * cols))
        data[:, :n] = agents[2:4]
        data = data.reshape((2, height, -1)).transpose((1, 2, 0))
        alive_mask = data[:, :, 0].astype(bool)
        zero = np.zeros(alive_mask.shape)
        return np.stack([zero, data[:, :, 1], zero, alive_mask], axis=-1)
