from .base import Agent
from .evo import ConvolutionModel, NeuralAutomataAgent, TorchAgent
from .gradient import GradientAgent, PhysarumAgent
from .static import BrownianAgent, ConstAgent

__all__ = ['Agent', 'GradientAgent', 'PhysarumAgent', 'BrownianAgent', 'ConstAgent', 'NeuralAutomataAgent', 'ConvolutionModel',
           'TorchAgent']
