#!/usr/bin/env python3
"""`python generic.py exampleN.json [output.json]` — same command line as the reference's
FEM/python/examples/json/generic.py, running the PINN+GD path on the MI355X HIP kernels."""
from pinn_fem_amd.cli.generic import main

if __name__ == "__main__":
    main()
