"""Material properties: constant or MLP(x[,y],load_factor).

Mirror of the reference interface FEM/python/fem/properties.py (Property :17, ScalarProperty :43,
NNProperty :57, to_property :192): same constructor arguments, same `.value(inputs)` conventions.
`.value()` is the reference's per-point evaluation API used for post-processing
(examples/json/generic.py:498-799); the solver never calls it — inside `solve*` the nets are
evaluated for all elements at once by the HIP kernels (pf_net_forward).
"""
from __future__ import annotations

from typing import Any, Optional

import numpy as np


class Property:
    """properties.py:17-40."""

    def value(self, inputs: Optional[Any] = None):
        raise NotImplementedError

    def is_trainable(self) -> bool:
        return False

    def get_torch_params(self) -> list:
        return []


class ScalarProperty(Property):
    """properties.py:43-54."""

    def __init__(self, value: float):
        self._value = float(value)

    def value(self, inputs: Optional[Any] = None) -> float:
        return self._value

    def __repr__(self) -> str:
        return f"ScalarProperty({self._value:.3e})"


class NNProperty(Property):
    """properties.py:57-189."""

    def __init__(self, net: Any, input_dim: int = 1, enforce_positive: bool = True,
                 scale: float = 1.0):
        self.net = net
        self.input_dim = input_dim
        self.enforce_positive = enforce_positive
        self.scale = scale

    def _input_tensor(self, inputs):
        import torch
        if inputs is None:                                       # :113-115
            x = np.zeros((1, self.input_dim))
        elif isinstance(inputs, dict):                           # :116-125 sorted keys
            cols = []
            for key in sorted(inputs.keys()):
                val = inputs[key]
                cols.append([val] if isinstance(val, (int, float)) else np.atleast_1d(val))
            x = np.column_stack(cols)
        else:                                                    # :126-143
            x = np.atleast_1d(inputs).astype(float)
            if x.ndim == 1:
                x = x.reshape(1, -1) if len(x) == self.input_dim else x.reshape(-1, 1)
            if x.shape[1] < self.input_dim:
                x = np.column_stack([x, np.zeros((x.shape[0], self.input_dim - x.shape[1]))])
        dev = next(self.net.parameters()).device
        return torch.tensor(np.asarray(x), dtype=torch.float32, device=dev)

    def value(self, inputs: Optional[Any] = None):
        import torch
        x = self._input_tensor(inputs)
        scalar_in = isinstance(inputs, (int, float)) or inputs is None
        if torch.is_grad_enabled():                              # :148-161
            out = self.net(x)
            if self.enforce_positive:
                out = torch.nn.functional.softplus(out)
            out = out * self.scale
            return out.squeeze() if scalar_in else out
        with torch.no_grad():                                    # :162-179
            out = self.net(x)
            if self.enforce_positive:
                out = torch.nn.functional.softplus(out)
            out = out * self.scale
            res = out.squeeze().cpu().numpy()
            if scalar_in:
                return float(res) if res.size == 1 else res
            return res

    def is_trainable(self) -> bool:
        return True

    def get_torch_params(self) -> list:
        return list(self.net.parameters())

    def __repr__(self) -> str:
        n_params = sum(p.numel() for p in self.net.parameters())
        return f"NNProperty(dim={self.input_dim}, params={n_params}, scale={self.scale:.3e})"


def to_property(value: Any) -> Property:
    """properties.py:192-205."""
    if isinstance(value, Property):
        return value
    if isinstance(value, (int, float)):
        return ScalarProperty(float(value))
    raise TypeError(f"Cannot convert {type(value)} to Property")
