"""`pinn_inverse_problem_gd` — the callee the reference's api_pinn_gradient_descent.py imports but never
defines (FEM/python/api_pinn_gradient_descent.py:19 raises ImportError).

Contract taken from the wrapper's own call site and result handling (api_pinn_gradient_descent.py:
102-121, 154-176): scalar Young's modulus and area are identified together with the displacement field
from measured displacements.  Arithmetic: **parity unpinned** (there is no reference behaviour to
match); it follows the nearest existing code, the legacy solver FEM/python/fem/nn_solver_gd.py:105-125:
    loss = alpha * mean(r_free^2) + beta * mean((u_meas - u[md])^2),  r = K(E,A) u - F,  Adam.
K is linear in E*A, so f_int = (E*A) * K_1 u with K_1 u and its transpose applied by the GD path's node
kernels; E = young_init*exp(p_E), A = area_init*exp(p_A) keep both positive and make one learning rate
meaningful for quantities of very different magnitude.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .boundary import free_and_fixed_dofs
from .model import FEMModel, Material
from ..engine import HipEngine


def pinn_inverse_problem_gd(nodes, elements, f_ext, fixed_dofs, young_init: float, area_init: float,
                            u_measured, measured_dofs, n_iterations: int = 500,
                            learning_rate: float = 1e-3, alpha: float = 1.0, beta: float = 100.0,
                            young_bounds: Optional[Sequence[float]] = None,
                            area_bounds: Optional[Sequence[float]] = None) -> Dict:
    """The whole loop runs on the device (pf_scalar_gd_iterations, include/pinnfem_hip.h): per iteration three launches
    — residual of the unit-stiffness operator scaled by c = exp(p_E + p_A) with its block sums, the displacement update
    (K_1^T g + data term, Adam, u[fixed] = 0: the GD path's own kernel), and a one-block kernel with the losses,
    d loss / d(p_E + p_A), Adam on the two log-multipliers, the bounds and the history row — enqueued by ONE call, read
    back once at the end."""
    import ctypes as C
    from .. import _capi
    from ..fem.solver import SolverConfig
    nodes = np.asarray(nodes, dtype=float)
    elements = np.asarray(elements, dtype=int)
    f_ext = np.asarray(f_ext, dtype=float).reshape(-1)
    model = FEMModel(nodes=nodes, elements=elements, material=Material(young=1.0, area=1.0),
                     loads=f_ext, fixed_dofs=np.asarray(fixed_dofs, dtype=int), dimension=2)
    md_np = np.asarray(measured_dofs, dtype=np.int64)
    um_np = np.asarray(u_measured, dtype=np.float64)
    eng = HipEngine(model, um_np if um_np.size else None, md_np if md_np.size else None)   # unit stiffness K_1
    dev = eng.device
    free, _fixed = free_and_fixed_dofs(model.ndof, model.fixed_dofs)
    ea0 = float(young_init) * float(area_init)
    u_scale = float(np.max(np.abs(um_np))) if um_np.size else 1.0
    n_it = int(n_iterations)
    cfg = SolverConfig(max_iterations=max(n_it, 1), tolerance=0.0, learning_rate_u=learning_rate * max(u_scale, 1e-30),
                       learning_rate_theta=learning_rate, alpha_physics=float(alpha), alpha_data=float(beta))
    eng.begin(None, 1.0, cfg, want_history=False)          # u = 0, fresh Adam state, Adam scalars of step 1
    eng.P.use_data = int(um_np.size > 0)                   # (beta = 0 still evaluates the data loss for the history)
    f32 = dict(dtype=torch.float32, device=dev)
    pst = torch.zeros(6, **f32)                            # p_E, p_A | m | v
    table = torch.zeros((max(n_it, 1), 5), **f32)          # loss, loss_p, loss_d, p_E, p_A
    sp = _capi.PfScalarId()
    base = pst.data_ptr()
    sp.p, sp.m_p, sp.v_p, sp.table, sp.n_rows = base, base + 8, base + 16, table.data_ptr(), max(n_it, 1)
    sp.has_bounds = 0
    if young_bounds is not None and area_bounds is not None:
        sp.has_bounds = 1
        sp.lo[0], sp.lo[1] = np.log(young_bounds[0] / young_init), np.log(area_bounds[0] / area_init)
        sp.hi[0], sp.hi[1] = np.log(young_bounds[1] / young_init), np.log(area_bounds[1] / area_init)
    sp.inv_ea0, sp.lr_p, sp.n_free_f = 1.0 / ea0, float(learning_rate), float(max(len(free), 1))
    with eng.on_stream():
        _capi.check(eng.lib.pf_scalar_gd_iterations(eng._ref(), C.byref(sp), n_it, eng._stream()),
                    "pf_scalar_gd_iterations")
    rows = table.cpu().numpy().astype(np.float64)           # (synchronises)
    history: List[Dict[str, float]] = [
        {"iteration": it + 1, "loss_total": float(rows[it, 0]), "loss_physics": float(rows[it, 1]),
         "loss_data": float(rows[it, 2]), "young": float(young_init * np.exp(rows[it, 3])),
         "area": float(area_init * np.exp(rows[it, 4]))} for it in range(n_it)]
    pe, pa = (float(x) for x in pst[:2].cpu())
    return {"u_final": eng.u.detach().cpu().numpy().astype(np.float64),
            "young_final": float(young_init * np.exp(pe)), "area_final": float(area_init * np.exp(pa)),
            "history": history}
