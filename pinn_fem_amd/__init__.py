"""pinn_fem_amd — MI355X-native PINN+GD inverse-identification hot path of PINN-FEM.

Importing the package does not need a GPU; constructing an engine or calling a solver does, and
fails loudly without one (no CPU fallback).  Build the HIP library with `python -m pinn_fem_amd.build`.
"""
__version__ = "0.1.0"
