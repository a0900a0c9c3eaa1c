"""die_amd — MI355X-native implementation of the grid-update hot path of gkirgizov/die:
`Env.step` field dynamics, `Agent.forward` (Physarum / Gradient / Brownian / Const) and the
`DataInitializer` allocation, behind the reference's Env / Dynamics / Agent Python API.

All compute runs in libdie_hip.so (hand-written HIP for gfx950, include/die_hip.h);
importing this package fails if that library is missing — there is no CPU fallback.
"""
from . import _lib
from .agent import Agent, BrownianAgent, ConstAgent, ConvolutionModel, GradientAgent, NeuralAutomataAgent, PhysarumAgent
from .base_types import DataChannels
from .data_init import DataInitializer, FieldSequence, PerlinNoiseSequence, WaveSequence
from .device_array import DeviceAction, DeviceAgents, DeviceMedium
from .env import BoundaryCondition, Dynamics, Env, linear_action_cost, zero_cost

__all__ = ['WaveSequence', 'PerlinNoiseSequence', 'FieldSequence', 'Env', 'Dynamics', 'BoundaryCondition', 'linear_action_cost', 'zero_cost', 'Agent', 'PhysarumAgent',
           'GradientAgent', 'BrownianAgent', 'ConstAgent', 'NeuralAutomataAgent', 'ConvolutionModel', 'DataInitializer', 'DataChannels', 'DeviceMedium',
           'DeviceAgents', 'DeviceAction']
