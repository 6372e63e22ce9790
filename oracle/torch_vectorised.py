"""torch_vectorised.py — the reference's PINN+GD iteration restated in BATCHED CPU PyTorch with autograd.

TEST / BENCH INFRASTRUCTURE ONLY (like pinn_oracle.py): imported by tests/ and by bench.py's `cpu_baseline` leg,
never by the product.  It is the "(V) vectorised CPU PyTorch restatement, all host cores" row SURVEY.md §8(d) and
BASELINE.md §3 define: the only CPU form of the reference's algorithm that can run at 10^5..10^6 elements, in the
reference's own cost class (torch tensors, autograd's `loss.backward()`, `torch.optim.Adam`).

What it restates, per iteration (FEM/python/...):
  fem/properties.py:116-156    NN input columns [load_factor, x, y], net(x), softplus, * scale — here ONE batch of
                               n_elems rows per property instead of n_elems batch-1 calls
  fem/nn_assembly.py:64-100    l0, cx, cy from the initial coordinates (float64 -> float32), s = E*A/l0,
                               fe = (s*pattern) @ u_e as the 4-term dot per row, b ascending
  fem/nn_assembly.py:226-227   f_int[g] += fe[a]  — here `index_add_` over the flattened element dofs
  fem/solver.py:267-283        r = f_int[free] - lam*f_ext[free]; 0.5*sum r^2; mean((u_meas - u[md])^2); total loss
  fem/solver.py:289-298        loss.backward(); optimizer_u.step(); optimizer_theta.step(); u[fixed] = 0
  fem/solver.py:304-320        monitors
The dense k_global the reference also fills (16 indexed += per element, never read by the GD loss) is NOT built:
at 10^6 elements it would be a 16 TB matrix.  Checked against pinn_oracle.py in tests/test_oracle_golden.py.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch


class _MLP(torch.nn.Module):
    """SimpleNN (examples/json/generic.py:118-142) built from a given parameter list (torch parameters() order)."""

    def __init__(self, tensors: List[np.ndarray]):
        super().__init__()
        layers = []
        n_lin = len(tensors) // 2
        for l in range(n_lin):
            w, b = tensors[2 * l], tensors[2 * l + 1]
            lin = torch.nn.Linear(w.shape[1], w.shape[0])
            with torch.no_grad():
                lin.weight.copy_(torch.from_numpy(np.asarray(w, dtype=np.float32)))
                lin.bias.copy_(torch.from_numpy(np.asarray(b, dtype=np.float32)))
            layers.append(lin)
            if l < n_lin - 1:
                layers.append(torch.nn.Tanh())
        self.net = torch.nn.Sequential(*layers)

    def forward(self, x):
        return self.net(x)


class TorchVectorisedGD:
    """State of one solve_gd call (fresh Adam on u and theta, solver.py:234-236) on a pinn_oracle.Problem."""

    def __init__(self, pb, lam: float, lr_u: float, lr_theta: float, alpha_physics: float = 1.0,
                 alpha_data: float = 100.0, u_initial: Optional[np.ndarray] = None):
        from . import pinn_oracle as orc
        geo = orc.element_geometry(pb)
        self.pb, self.lam = pb, float(lam)
        self.alpha_p, self.alpha_d = float(alpha_physics), float(alpha_data)
        t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
        self.dofs = t(geo.dofs.reshape(-1), torch.int64)                # (nelm*nd,)
        self.nd = geo.dofs.shape[1]
        self.pattern = t(geo.pattern)                                   # (nelm, nd, nd) float32 of the float64 cosines
        self.l0 = t(geo.l0)
        self.x_in = t(np.concatenate([np.full((geo.nn_input.shape[0], 1), np.float32(lam), dtype=np.float32),
                                      geo.nn_input], axis=1))           # columns load_factor, x[, y]
        ndof = pb.ndof
        free, fixed = orc.free_and_fixed_dofs(ndof, pb.fixed_dofs)
        self.free, self.fixed = t(free, torch.int64), t(fixed, torch.int64)
        self.f_ext = t(pb.loads)
        self.has_meas = pb.measured_vals is not None and pb.measured_dofs is not None and len(pb.measured_vals) > 0
        if self.has_meas:
            self.mv, self.md = t(pb.measured_vals), t(pb.measured_dofs, torch.int64)
        self.nets, self.scales = [], []
        for prop in (pb.young, pb.area):
            if isinstance(prop, orc.NetParams):
                self.nets.append(_MLP(prop.tensors))
                self.scales.append((float(prop.scale), bool(prop.enforce_positive)))
            else:
                self.nets.append(None)
                self.scales.append((float(prop), False))
        u0 = np.zeros(ndof, dtype=np.float32) if u_initial is None else np.asarray(u_initial, dtype=np.float32)
        self.u = torch.tensor(u0, dtype=torch.float32, requires_grad=True)
        self.opt_u = torch.optim.Adam([self.u], lr=lr_u)
        theta = [p for n in self.nets if n is not None for p in n.parameters()]
        self.theta = theta
        self.opt_t = torch.optim.Adam(theta, lr=lr_theta) if theta else None

    def _prop(self, k):
        net, (scale, positive) = self.nets[k], self.scales[k]
        if net is None:
            return torch.full((self.x_in.shape[0],), scale, dtype=torch.float32)
        z = net(self.x_in)[:, 0]
        return (torch.nn.functional.softplus(z) if positive else z) * scale      # properties.py:154-156

    def loss(self):
        s = (self._prop(0) * self._prop(1)) / self.l0                               # nn_assembly.py:74
        ke = s[:, None, None] * self.pattern
        ue = self.u[self.dofs].reshape(-1, self.nd)
        fe = torch.zeros_like(ue)
        for b in range(self.nd):                                                    # 4-term dot, b ascending (:96-100)
            fe = fe + ke[:, :, b] * ue[:, b:b + 1]
        f_int = torch.zeros_like(self.u).index_add_(0, self.dofs, fe.reshape(-1))   # :226-227
        r = f_int[self.free] - self.lam * self.f_ext[self.free]                     # solver.py:267-269
        loss_p = 0.5 * torch.sum(r ** 2)
        if self.has_meas and self.alpha_d > 0:
            loss_d = torch.mean((self.mv - self.u[self.md]) ** 2)
            loss = self.alpha_p * loss_p + self.alpha_d * loss_d
        else:
            loss_d = torch.zeros(())
            loss = self.alpha_p * loss_p
        return loss, loss_p, loss_d, r

    def step(self):
        """One GD iteration (solver.py:254-320); returns the history entry's numbers."""
        self.opt_u.zero_grad()
        if self.opt_t is not None:
            self.opt_t.zero_grad()
        loss, loss_p, loss_d, r = self.loss()
        loss.backward()                                                             # :289
        self.opt_u.step()                                                           # :292
        if self.opt_t is not None:
            self.opt_t.step()                                                       # :294
        with torch.no_grad():
            self.u[self.fixed] = 0.0                                                # :297-298
            return {"loss_total": float(loss), "loss_physics": float(loss_p), "loss_data": float(loss_d),
                    "residual_norm": float(torch.norm(r)), "u_norm": float(torch.norm(self.u[self.free]))}

    def run(self, n_iter: int):
        return [self.step() for _ in range(n_iter)]
