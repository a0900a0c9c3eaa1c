"""Agent base class (reference core/agent/base.py:12-62)."""
import inspect
import io
import json
import os
from abc import ABC, abstractmethod
from copy import deepcopy
from typing import Any, Callable, Dict, Optional, Sequence, Union

import numpy as np


def save_args(fun: Callable, locs: Dict[str, Any]) -> Dict[str, Any]:
    """core/utils.py:214-220: constructor arguments (with defaults) as a dict."""
    sig = inspect.signature(fun)
    return {name: deepcopy(locs.get(name, param.default)) for name, param in sig.parameters.items()}


class Agent(ABC):
    @abstractmethod
    def forward(self, obs):
        """Act in the environment given observations."""

    def render(self) -> Sequence[Optional[np.ndarray]]:
        return [None]

    @property
    @abstractmethod
    def init_params(self) -> Dict[str, Any]:
        """Parameters from which the agent can be reconstructed.  (The reference declares
        this a property in the base class and overrides it with plain methods, which breaks
        its own `save`; here it is a property throughout.)"""

    def save(self, file: Union[str, os.PathLike, io.IOBase]):
        data = json.dumps(self.init_params)
        if isinstance(file, (str, os.PathLike)):
            with open(file, 'w') as f:
                f.write(data)
        else:
            file.write(data)

    @classmethod
    def load(cls, file: Union[str, os.PathLike, io.IOBase]) -> 'Agent':
        if isinstance(file, (str, os.PathLike)):
            with open(file, 'r') as f:
                params = json.load(f)
        else:
            params = json.load(file)
        return cls(**params)
