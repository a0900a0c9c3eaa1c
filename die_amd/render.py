"""Frames for `Env.render()` (reference core/render.py:76-132): the plotting loop of examples/minimal_run.py
needs three images per call.  `DeviceRenderer` builds them on the GPU (`die_render_frames`: one sweep over the
medium, the agent trace kept in HBM) and downloads finished float32 images — or one uint8 image (`rgb8`);
the host restatement of the reference's renderer that checks these frames lives in oracle/render_ref.py."""
from typing import Tuple

import numpy as np


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
