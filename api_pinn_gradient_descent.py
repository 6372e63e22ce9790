#!/usr/bin/env python3
"""`python api_pinn_gradient_descent.py input.json output.json` — same command line as the reference's
FEM/python/api_pinn_gradient_descent.py, running on the MI355X HIP kernels."""
from pinn_fem_amd.cli.api_pinn_gradient_descent import main

if __name__ == "__main__":
    main()
