"""Frames for `Env.render()` (reference core/render.py:76-132): the plotting loop of examples/minimal_run.py
needs three images per call.  `DeviceRenderer` builds them on the GPU (`die_render_frames`: one sweep over the
medium, the agent trace kept in HBM) and downloads finished float32 images — or one uint8 image (`rgb8`);
`EnvRenderer` is the same thing from downloaded float64 arrays (the reference's own code path, kept as the
checker of the device frames in tests/test_gpu_parity.py)."""
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


def _colormap_lut(colored: bool) -> np.ndarray:
    """matplotlib's lookup table of the trace colormap: N colours + the under / over / bad rows."""
    try:
        import matplotlib
        cmap = matplotlib.colormaps['magma' if colored else 'gray']
        cmap._init()
        return np.ascontiguousarray(cmap._lut, dtype=np.float32)
    except ImportError:
        g = np.linspace(0., 1., 256)
        lut = np.stack([g, g, g, np.ones_like(g)], axis=-1)
        return np.concatenate([lut, lut[:1], lut[-1:], np.zeros((1, 4))]).astype(np.float32)


class DeviceRenderer:
    """core/render.py:76-132 on the device.  Frames come back as float32 numpy arrays of the reference's shapes."""

    def __init__(self, field_size: Tuple[int, int], device, is_trace_colored: bool = True, trace_steps: int = 8):
        import torch
        self.field_size = field_size
        self.device = device
        self._decay = 1 - 1 / trace_steps
        W, H = field_size
        self._trace = torch.zeros((W, H), dtype=torch.float32, device=device)
        lut = _colormap_lut(is_trace_colored)
        self._lut_n = lut.shape[0] - 3
        self._lut = torch.from_numpy(lut).to(device)
        self._rgb = torch.empty((W, H, 3), dtype=torch.float32, device=device)
        self._rgba = torch.empty((W, H, 4), dtype=torch.float32, device=device)
        self._rgb8 = None
        # downloads land in pinned host buffers (two alternating sets: the frames of a call stay valid until the
        # call after the next one, which is what a draw-every-step loop needs) and are returned as numpy views
        self._host, self._flip = {}, 0

    def _download(self, name, t):
        import torch
        key = (name, self._flip)
        buf = self._host.get(key)
        if buf is None or buf.shape != t.shape:
            buf = self._host[key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        buf.copy_(t, non_blocking=True)
        return buf

    def _run(self, medium, rgb, rgba, rgb8, update_trace=True):
        import ctypes as C
        from . import _lib
        from .device_array import _ptr, stream_ptr
        m = medium.c_struct()
        _lib.check(_lib.lib.die_render_frames(C.byref(m), _ptr(self._trace) if update_trace else None, self._decay, _ptr(self._lut),
                                              self._lut_n, _ptr(rgb) if rgb is not None else None,
                                              _ptr(rgba) if rgba is not None else None,
                                              _ptr(rgb8) if rgb8 is not None else None, stream_ptr(self.device)), 'die_render_frames')

    def render(self, medium, agents):
        """[medium (W, H, 3), agent trace (W, H, 4), agents (height, cols, 4)] — one sweep, three downloads."""
        import torch
        self._run(medium, self._rgb, self._rgba, None)
        self._flip ^= 1
        out = [self._download('rgb', self._rgb), self._download('rgba', self._rgba), self._download('agents', self._img_agents(agents))]
        torch.cuda.synchronize(self.device)
        return [b.numpy() for b in out]

    def rgb8(self, medium) -> np.ndarray:
        """The medium image as (W, H, 3) uint8 (values clipped to [0, 1]): 3 bytes per cell cross the bus.  Does not
        advance the agent trace."""
        import torch
        if self._rgb8 is None:
            self._rgb8 = torch.empty(self.field_size + (3,), dtype=torch.uint8, device=self.device)
        self._run(medium, None, None, self._rgb8, update_trace=False)
        self._flip ^= 1
        buf = self._download('rgb8', self._rgb8)
        torch.cuda.synchronize(self.device)
        return buf.numpy()

    def _img_agents(self, agents):
        """core/render.py:112-132: (alive, agent_food) in slot order laid out as an image of `height` rows."""
        import torch
        width, height = self.field_size
        n = agents.N
        cols = -(-n // height)
        data = torch.zeros((2, height * cols), dtype=torch.float32, device=self.device)
        data[0, :n] = agents.sel('alive').to(torch.float32)
        data[1, :n] = agents.sel('agent_food')
        data = data.reshape(2, height, cols)
        zero = torch.zeros_like(data[0])
        return torch.stack([zero, data[1], zero, (data[0] != 0).to(torch.float32)], dim=-1)
