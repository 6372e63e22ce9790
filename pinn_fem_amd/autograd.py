"""torch.autograd.Function over the HIP kernels: f_int = K(theta) u with gradients to u and to the
network parameters.  This is the differentiable seam the reference exposes as
assemble_system_torch (FEM/python/fem/nn_assembly.py:105-231): forward = per-element MLP + element
stiffness + assembly, backward = the vector-Jacobian products autograd derives from it.
"""
from __future__ import annotations

import torch

from .engine import HipEngine


class InternalForceFn(torch.autograd.Function):
    """f_int(u, theta) -> (n_dofs,).  Inputs: engine, load_factor, u, *active parameters (young net
    then area net, parameters() order).  Parameters must be the engine's views of its flat theta."""

    @staticmethod
    def forward(ctx, engine: HipEngine, lam: float, u: torch.Tensor, *params: torch.Tensor):
        ctx.engine, ctx.lam = engine, float(lam)
        ctx.n_params = len(params)
        ctx.shapes = [tuple(p.shape) for p in params]
        u_dev = u.detach().to(device=engine.device, dtype=torch.float32).contiguous()
        ctx.save_for_backward(u_dev)
        ctx.u_device = u.device
        f = engine.internal_force(u_dev, lam)
        return f.to(u.device)

    @staticmethod
    def backward(ctx, g_f: torch.Tensor):
        (u_dev,) = ctx.saved_tensors
        eng = ctx.engine
        gu, gt = eng.vjp(u_dev, g_f.detach(), ctx.lam)
        grads = []
        off = 0
        for shp in ctx.shapes:
            n = 1
            for s in shp:
                n *= s
            grads.append(gt[off:off + n].clone().view(shp))
            off += n
        return (None, None, gu.clone().to(ctx.u_device), *grads)


def internal_force(engine: HipEngine, u: torch.Tensor, load_factor: float) -> torch.Tensor:
    """Differentiable f_int for `engine`'s model at displacement u."""
    mat = engine.model.material
    params = []
    for prop in (mat.young, mat.area):
        if prop.is_trainable():
            params.extend(prop.get_torch_params())
    return InternalForceFn.apply(engine, float(load_factor), u, *params)
