"""NeuralAutomataAgent / ConvolutionModel (reference core/agent/evo.py:18-209): same constructor arguments,
`forward(obs) -> action`, `render()`, `save` / `load`.  The model stays a torch `nn.Module` (its parameters are what the
evolution loop of examples/learning_agents.py edits); `NeuralAutomataAgent.forward` — medium → conv stack → tanh →
per-agent read-out → scaled action — runs on the device through libdie_hip.so (`die_conv2d_circular`,
`die_gather_scale`) on the medium's own planes: no host copy of the field, no torch op on the path."""
import ctypes as C
import io
import os
from abc import abstractmethod
from typing import Any, Dict, Optional, Sequence, Union

import numpy as np
import torch as th
from torch import nn

from .. import _lib
from ..base_types import DataChannels
from ..device_array import DeviceAction, _ptr, stream_ptr
from .base import Agent, save_args


_CHECKPOINT_KEYS = ('params_dict', 'model_state')      # the layout of a saved agent (core/agent/evo.py:24-42): kept, so that files interchange


class TorchAgent(Agent, nn.Module):
    """An agent whose behaviour is a torch model (core/agent/evo.py:18-42): `save` writes the constructor arguments and the
    model's state_dict under the reference's two keys, `load` rebuilds the agent from them."""

    @property
    @abstractmethod
    def model(self) -> nn.Module:
        """The module whose parameters the evolution loop edits."""

    def save(self, file: Union[str, os.PathLike, io.IOBase]):
        th.save(dict(zip(_CHECKPOINT_KEYS, (self.init_params, self.model.state_dict()))), file)

    @classmethod
    def load(cls, file: Union[str, os.PathLike, io.IOBase]):
        loaded = th.load(file)                                         # ONE read: `file` may be a stream
        ctor_args, weights = (loaded[k] for k in _CHECKPOINT_KEYS)
        ctor_args = dict(ctor_args)
        ctor_args.update(ctor_args.pop('model_kwargs', {}))            # **model_kwargs of NeuralAutomataAgent are stored nested
        restored = cls(**ctor_args)
        restored.model.load_state_dict(weights)
        return restored


class ConvolutionModel(nn.Module):
    """The perception model of core/agent/evo.py:45-118: a stack of bias-free 'same'-padded Conv2d layers (one per entry of
    `kernel_sizes`; every layer keeps the observation channels except the last, which maps them to the action channels), a Tanh
    that brings the outputs into [-1, 1], and an (inverted-)dropout mask over cells.  The submodule names `kernels.<i>` and
    `agent_dropout` are the reference's, so state_dicts interchange; the device path (NeuralAutomataAgent.sense) reads the
    layers' weights and never calls this module's forward."""

    def __init__(self,
                 num_obs_channels: int = 3,
                 num_act_channels: int = 3,
                 kernel_sizes: Sequence[int] = (3,),
                 boundary: str = 'circular',
                 p_agent_dropout: float = 0.,
                 requires_grad: bool = True,
                 ):
        self._init_params = save_args(self.__init__, locals())
        super().__init__()
        depth = len(kernel_sizes)
        stack = nn.Sequential()
        for level, size in enumerate(kernel_sizes):
            widens_to_actions = level == depth - 1
            stack.append(nn.Conv2d(num_obs_channels, num_act_channels if widens_to_actions else num_obs_channels, size,
                                   padding='same', padding_mode=boundary, bias=False))
        stack.append(nn.Tanh())
        self.agent_dropout = nn.Dropout(p_agent_dropout)
        self.kernels = stack
        self.requires_grad_(requires_grad)

    def conv_layers(self):
        return [layer for layer in self.kernels if isinstance(layer, nn.Conv2d)]

    def init_weights(self):
        for layer in self.conv_layers():
            nn.init.xavier_uniform_(layer.weight)

    def forward(self, input: th.Tensor) -> th.Tensor:
        """The torch evaluation (what the reference runs; here the fp32 reference of the device path in tests, and for callers
        that hold plain tensors): the cells' dropout mask is drawn on the field's own device."""
        keep = self.agent_dropout(th.ones(input.shape[-2:], device=input.device, dtype=input.dtype))
        return self.kernels(input) * keep


class NeuralAutomataAgent(TorchAgent):
    """core/agent/evo.py:121-209."""

    def __init__(self,
                 scale: float = 0.1,
                 deposit: float = 1.0,
                 with_agent_channel: bool = True,
                 initial_obs=None,
                 **model_kwargs,
                 ):
        self._init_params = save_args(self.__init__, locals())
        self._init_params.pop('initial_obs', None)            # an observation is not a constructor parameter worth saving
        super().__init__()
        self.obs_channels = list(DataChannels.medium) if with_agent_channel else list(DataChannels.medium[1:])
        self._model = ConvolutionModel(num_obs_channels=len(self.obs_channels), num_act_channels=len(DataChannels.actions),
                                       **model_kwargs)
        self.action_coefs = (float(scale), float(scale), float(deposit))
        self._sense_output: Optional[th.Tensor] = None         # (3, W, H) on the device after the first forward
        self._planes = {}                                      # device scratch by (W, H, device): ping-pong plane sets

    @property
    def model(self) -> ConvolutionModel:
        return self._model

    @property
    def init_params(self) -> Dict[str, Any]:
        return self._init_params

    # ------------------------------------------------------------------
    def _device_weights(self, device):
        layers = self._model.conv_layers()
        for k in layers:
            if k.padding_mode not in _lib.PAD_MODES:
                raise NotImplementedError(f"boundary={k.padding_mode!r}: one of {sorted(_lib.PAD_MODES)}")
        # uploaded on every call (≈ 100 floats): in-place writes through `param.data` — how evotorch's fill_parameters
        # loads each candidate — change neither `_version` nor `data_ptr()`, so no cheap key tells a stale copy apart
        return [k.weight.detach().to(device=device, dtype=th.float32).contiguous() for k in layers]

    def _scratch(self, W, H, device, n_sets):
        key = (W, H, str(device))
        sets = self._planes.get(key)
        if sets is None or len(sets) < n_sets:
            sets = [th.empty((4, W, H), dtype=th.float32, device=device) for _ in range(n_sets)]
            self._planes[key] = sets
        return sets

    def sense(self, medium) -> th.Tensor:
        """ConvolutionModel.forward over the medium on the device → (3, W, H) float32 tensor."""
        medium.sensed()
        dev, W, H = medium.device, medium.W, medium.H
        weights = self._device_weights(dev)
        sp = stream_ptr(dev)
        medium._ensure_owner()
        fkind = _lib.DIE_PLANE_F32 if medium.dtype == th.float32 else _lib.DIE_PLANE_F16
        src = {'agents': (medium.owner, _lib.DIE_PLANE_AGENTS), 'env_food': (medium.food, fkind), 'chem1': (medium.chem, fkind)}
        planes = [src[c] for c in self.obs_channels]
        sets = self._scratch(W, H, dev, 2)
        for li, w in enumerate(weights):
            cout, cin, k, k2 = w.shape
            if k != k2 or cin != len(planes):
                raise ValueError(f'layer {li}: weight {tuple(w.shape)} does not fit {len(planes)} input planes')
            dst = sets[li % 2]
            cin_arr = (_lib.ConvPlane * cin)(*[_lib.ConvPlane(t.data_ptr(), kind, 0) for t, kind in planes])
            out_arr = (C.c_void_p * cout)(*[dst[o].data_ptr() for o in range(cout)])
            _lib.check(_lib.lib.die_conv2d(W, H, cin, cin_arr, medium.epoch, cout, out_arr, k, _ptr(w), int(li == len(weights) - 1),
                                           _lib.PAD_MODES[self._model.conv_layers()[li].padding_mode], sp), 'die_conv2d')
            planes = [(dst[o], _lib.DIE_PLANE_F32) for o in range(cout)]
        out = sets[(len(weights) - 1) % 2][:len(planes)]         # the last layer's planes (scratch: valid until the next call)
        p = self._model.agent_dropout.p
        if p > 0 and self._model.training:                       # rare path: the mask of core/agent/evo.py:114-116
            out = out * self._model.agent_dropout(th.ones((W, H), device=dev))
        return out

    def forward(self, obs) -> DeviceAction:
        """:150-174."""
        agents, medium = obs
        sense = self.sense(medium)
        self._sense_output = sense
        action = DeviceAction(agents.N, agents.device, agents.slot, capacity=agents.capacity)
        action.global_slots = agents.global_slots
        planes = (C.c_void_p * 3)(*[sense[c].data_ptr() for c in range(3)])
        coefs = (C.c_float * 3)(*self.action_coefs)
        m, a, u = medium.c_struct(need_owner=False), agents.c_struct(), action.c_struct()
        _lib.check(_lib.lib.die_gather_scale(C.byref(m), C.byref(a), planes, coefs, C.byref(u), stream_ptr(agents.device)),
                   'die_gather_scale')
        action._keepalive = sense
        return action

    def render(self) -> Sequence[np.ndarray]:
        """:176-181: the transformed medium with the channel axis last."""
        if self._sense_output is None:
            return [np.ones((2, 2, 3))]
        return [th.moveaxis(self._sense_output, 0, -1).cpu().numpy()]
