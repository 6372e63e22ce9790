"""`pinn_inverse_problem_gd` — the callee the reference's api_pinn_gradient_descent.py imports but never
defines (FEM/python/api_pinn_gradient_descent.py:19 raises ImportError).

Contract taken from the wrapper's own call site and result handling (api_pinn_gradient_descent.py:
102-121, 154-176): scalar Young's modulus and area are identified together with the displacement field
from measured displacements.  Arithmetic: **parity unpinned** (there is no reference behaviour to
match); it follows the nearest existing code, the legacy solver FEM/python/fem/nn_solver_gd.py:105-125:
    loss = alpha * mean(r_free^2) + beta * mean((u_meas - u[md])^2),  r = K(E,A) u - F,  Adam.
K is linear in E*A, so f_int = (E*A) * K_1 u with K_1 u evaluated (and differentiated w.r.t. u) by the
HIP kernels through InternalForceFn; E = young_init*exp(p_E), A = area_init*exp(p_A) keep both positive
and make one learning rate meaningful for quantities of very different magnitude.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .boundary import free_and_fixed_dofs
from .model import FEMModel, Material
from ..autograd import internal_force
from ..engine import HipEngine


def pinn_inverse_problem_gd(nodes, elements, f_ext, fixed_dofs, young_init: float, area_init: float,
                            u_measured, measured_dofs, n_iterations: int = 500,
                            learning_rate: float = 1e-3, alpha: float = 1.0, beta: float = 100.0,
                            young_bounds: Optional[Sequence[float]] = None,
                            area_bounds: Optional[Sequence[float]] = None) -> Dict:
    nodes = np.asarray(nodes, dtype=float)
    elements = np.asarray(elements, dtype=int)
    f_ext = np.asarray(f_ext, dtype=float).reshape(-1)
    model = FEMModel(nodes=nodes, elements=elements, material=Material(young=1.0, area=1.0),
                     loads=f_ext, fixed_dofs=np.asarray(fixed_dofs, dtype=int), dimension=2)
    eng = HipEngine(model)                       # unit stiffness operator K_1 on the device
    dev = eng.device
    free, fixed = free_and_fixed_dofs(model.ndof, model.fixed_dofs)
    free_t = torch.as_tensor(free, device=dev)
    fixed_t = torch.as_tensor(fixed, device=dev)
    md = torch.as_tensor(np.asarray(measured_dofs, dtype=np.int64), device=dev)
    um = torch.as_tensor(np.asarray(u_measured, dtype=np.float32), device=dev)
    fx = torch.as_tensor(f_ext.astype(np.float32), device=dev)

    # the residual is scaled by 1/(E0*A0) so that its size does not depend on the units of E
    ea0 = float(young_init) * float(area_init)
    u = torch.zeros(model.ndof, dtype=torch.float32, device=dev, requires_grad=True)
    p = torch.zeros(2, dtype=torch.float32, device=dev, requires_grad=True)   # log-multipliers of E, A
    u_scale = float(um.abs().max().item()) if um.numel() else 1.0
    opt = torch.optim.Adam([{"params": [u], "lr": learning_rate * max(u_scale, 1e-30)},
                            {"params": [p], "lr": learning_rate}])
    lo = hi = None
    if young_bounds is not None and area_bounds is not None:
        lo = torch.tensor([np.log(young_bounds[0] / young_init), np.log(area_bounds[0] / area_init)],
                          dtype=torch.float32, device=dev)
        hi = torch.tensor([np.log(young_bounds[1] / young_init), np.log(area_bounds[1] / area_init)],
                          dtype=torch.float32, device=dev)
    # No host synchronisation inside the loop: the monitors of every iteration go to a device table that is read
    # once at the end (the loop is then bound by launch rate, not by a .item() round trip per quantity).
    n_it = int(n_iterations)
    table = torch.zeros((max(n_it, 1), 5), dtype=torch.float32, device=dev)   # loss, loss_p, loss_d, p_E, p_A
    for it in range(n_it):
        opt.zero_grad(set_to_none=True)
        k1u = internal_force(eng, u, 1.0)                         # HIP: K_1 u, differentiable in u
        r = torch.exp(p[0] + p[1]) * k1u[free_t] - fx[free_t] / ea0
        loss_p = torch.mean(r ** 2)                               # nn_solver_gd.py:113
        loss_d = torch.mean((um - u[md]) ** 2)                    # nn_solver_gd.py:117-118
        loss = alpha * loss_p + beta * loss_d
        loss.backward()
        opt.step()
        with torch.no_grad():
            u[fixed_t] = 0.0
            if lo is not None:
                p.copy_(torch.minimum(torch.maximum(p, lo), hi))
            table[it, 0], table[it, 1], table[it, 2] = loss.detach(), loss_p.detach(), loss_d.detach()
            table[it, 3:5] = p.detach()
    rows = table.cpu().numpy().astype(np.float64)
    history: List[Dict[str, float]] = [
        {"iteration": it + 1, "loss_total": float(rows[it, 0]), "loss_physics": float(rows[it, 1]),
         "loss_data": float(rows[it, 2]), "young": float(young_init * np.exp(rows[it, 3])),
         "area": float(area_init * np.exp(rows[it, 4]))} for it in range(n_it)]
    pe, pa = (float(x) for x in p.detach().cpu())
    return {"u_final": u.detach().cpu().numpy().astype(np.float64),
            "young_final": float(young_init * np.exp(pe)), "area_final": float(area_init * np.exp(pa)),
            "history": history}
