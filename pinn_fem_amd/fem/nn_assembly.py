"""Differentiable assembly — compatibility view of the reference's
FEM/python/fem/nn_assembly.py:105-231 `assemble_system_torch(model, disp, load_factor)`.

f_int keeps the autograd graph to `disp` and to the NN parameters (through
pinn_fem_amd.autograd.InternalForceFn, i.e. the HIP kernels).  k_global is the dense matrix the
reference returns, materialised on request for small problems only: the GD solver never reads it
(fem/solver.py:262 discards it), and at the sizes this build targets it cannot exist
(N = 10^6 elements would need a 16 TB float32 matrix).
"""
from __future__ import annotations

from typing import Tuple

import torch

from .model import FEMModel
from ..autograd import internal_force
from ..engine import HipEngine

DENSE_K_MAX_DOFS = 4096


class LazyStiffness:
    """Stand-in for k_global when n_dofs is too large to materialise; matrix-free products."""

    def __init__(self, engine: HipEngine, load_factor: float):
        self.engine, self.load_factor = engine, load_factor
        n = engine.plan.n_dofs
        self.shape = (n, n)

    def diagonal(self) -> torch.Tensor:
        return self.engine.diag_k(self.load_factor)

    def matvec(self, v: torch.Tensor) -> torch.Tensor:
        return self.engine.internal_force(v, self.load_factor)

    def __matmul__(self, v):
        return self.matvec(v)

    def to_dense(self) -> torch.Tensor:
        return self.engine.dense_k(self.load_factor)

    def to_sparse(self) -> torch.Tensor:
        """k_global as a coalesced sparse COO tensor: the reference's dense matrix (nn_assembly.py:228-231) in the
        only form that exists at scale."""
        return self.engine.sparse_k(self.load_factor)


def _engine(model: FEMModel) -> HipEngine:
    eng = getattr(model, "_pf_assembly_engine", None)
    if eng is None or (eng.n_theta and not eng.theta.still_bound()):
        eng = HipEngine(model)
        model._pf_assembly_engine = eng
    return eng


def assemble_system_torch(model: FEMModel, disp: torch.Tensor, load_factor: float = 1.0
                          ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (k_global, f_int) like nn_assembly.py:105-231.  k_global is dense (no autograd) for
    n_dofs <= 4096, else a LazyStiffness."""
    eng = _engine(model)
    f_int = internal_force(eng, disp, load_factor)
    if eng.plan.n_dofs <= DENSE_K_MAX_DOFS:
        k_global = eng.dense_k(load_factor).to(disp.device)
    else:
        k_global = LazyStiffness(eng, load_factor)
    return k_global, f_int
