"""MLP description + flat parameter storage for the HIP kernels.

The reference lets NNProperty wrap any torch.nn.Module (FEM/python/fem/properties.py:66-84) but
its entry point only ever builds SimpleNN = Linear(in,h)-Tanh-[Linear(h,h)-Tanh]*(L-1)-Linear(h,1)
(FEM/python/examples/json/generic.py:118-142).  The HIP kernels implement exactly that family
(in 2|3, h 1..32, L 1..3); anything else raises NotImplementedError — there is no CPU fallback.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.nn as nn


class SimpleNN(nn.Module):
    """Same architecture and initialisation as the reference's SimpleNN
    (examples/json/generic.py:118-142): default torch init, last bias = 1.0, last weight = 0.1."""

    def __init__(self, hidden_layers: int = 2, neurons_per_layer: int = 20, input_dim: int = 1):
        super().__init__()
        layers: List[nn.Module] = [nn.Linear(input_dim, neurons_per_layer), nn.Tanh()]
        for _ in range(hidden_layers - 1):
            layers.append(nn.Linear(neurons_per_layer, neurons_per_layer))
            layers.append(nn.Tanh())
        layers.append(nn.Linear(neurons_per_layer, 1))
        self.net = nn.Sequential(*layers)
        with torch.no_grad():
            self.net[-1].bias.fill_(1.0)
            self.net[-1].weight.fill_(0.1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.net(x)


@dataclass
class NetSpec:
    enabled: bool = False
    in_dim: int = 0
    width: int = 0
    n_hidden: int = 0
    positive: bool = True
    scale: float = 1.0
    n_params: int = 0
    shapes: List[tuple] = field(default_factory=list)  # parameters() shapes, in order


def describe_module(module: nn.Module) -> NetSpec:
    """Check that `module` is a Linear/Tanh MLP of the supported family and describe it."""
    leaves = [m for m in module.modules() if len(list(m.children())) == 0]
    linears = [m for m in leaves if isinstance(m, nn.Linear)]
    if not linears or len(leaves) != 2 * len(linears) - 1:
        raise NotImplementedError(
            "HIP path supports Linear-Tanh-...-Linear MLPs only (reference SimpleNN, generic.py:118-142)")
    for i, m in enumerate(leaves):
        want = nn.Linear if i % 2 == 0 else nn.Tanh
        if not isinstance(m, want):
            raise NotImplementedError(f"unsupported layer {type(m).__name__} at position {i}")
    n_hidden = len(linears) - 1
    width = linears[0].out_features
    in_dim = linears[0].in_features
    if n_hidden < 1 or n_hidden > 3 or width < 1 or width > 32 or in_dim not in (2, 3):
        raise NotImplementedError(
            f"MLP shape in={in_dim} width={width} hidden_layers={n_hidden} is outside the HIP kernel menu "
            "(in 2|3, width 1..32, hidden layers 1..3)")
    params = list(module.parameters())
    expect = []
    for k, lin in enumerate(linears):
        if lin.bias is None:
            raise NotImplementedError("Linear layers without bias are not supported")
        fin = in_dim if k == 0 else width
        fout = 1 if k == n_hidden else width
        if lin.in_features != fin or lin.out_features != fout:
            raise NotImplementedError("hidden layers must all have the same width and the output 1 unit")
        expect += [lin.weight, lin.bias]
    if len(params) != len(expect) or any(a is not b for a, b in zip(params, expect)):
        raise NotImplementedError("parameters() order differs from W1,b1,...,Wout,bout")
    return NetSpec(enabled=True, in_dim=in_dim, width=width, n_hidden=n_hidden,
                   n_params=sum(p.numel() for p in params), shapes=[tuple(p.shape) for p in params])


class FlatTheta:
    """All trainable parameters of a Material in ONE flat float32 device vector, in the reference's
    enumeration order young -> area -> density (FEM/python/fem/model.py:36-42).  Each nn.Parameter
    of the user's modules is re-pointed to a view of this vector, so the modules stay live views of
    what the kernels update in place."""

    def __init__(self, param_lists: List[List[nn.Parameter]], device: torch.device):
        self.params: List[nn.Parameter] = [p for lst in param_lists for p in lst]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(max(n, 1), dtype=torch.float32, device=device)
        self.n = n
        self.tensor_off = [0]
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat[off:off + k].copy_(p.detach().reshape(-1).to(torch.float32))
                p.data = self.flat[off:off + k].view(p.shape)
                off += k
                self.tensor_off.append(off)

    def still_bound(self) -> bool:
        """True while every parameter is still a view of the flat vector."""
        base = self.flat.data_ptr()
        off = 0
        for p in self.params:
            if p.data_ptr() != base + 4 * off or p.device != self.flat.device:
                return False
            off += p.numel()
        return True
