"""Mirror of the reference's `fem` package for the accelerated PINN+GD path."""
from .model import FEMModel, Material
from .properties import Property, ScalarProperty, NNProperty, to_property
from .boundary import free_and_fixed_dofs

__all__ = ["FEMModel", "Material", "Property", "ScalarProperty", "NNProperty", "to_property",
           "free_and_fixed_dofs"]
