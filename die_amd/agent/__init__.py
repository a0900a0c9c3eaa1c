from .base import Agent
from .gradient import GradientAgent, PhysarumAgent
from .static import BrownianAgent, ConstAgent

__all__ = ['Agent', 'GradientAgent', 'PhysarumAgent', 'BrownianAgent', 'ConstAgent']
