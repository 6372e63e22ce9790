"""free/fixed DOF partition — same function as the reference's FEM/python/fem/boundary.py:8-13."""
from __future__ import annotations

from ..plan import free_and_fixed_dofs

__all__ = ["free_and_fixed_dofs"]
