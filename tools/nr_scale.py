#!/usr/bin/env python3
"""solve_nr (matrix-free float64 K v + Jacobi-PCG on the device) on Warren girders of growing size; for the
smaller ones also the oracle's dense float64 restatement of the reference (np.linalg.solve) on the host."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PINNFEM_QUIET", "1")
from pinn_fem_amd.fem.model import FEMModel, Material
from pinn_fem_amd.fem.solver import SolverConfig, solve_nr
from pinn_fem_amd.plan import warren_mesh
import torch
rows = []
for panels in [int(a) for a in sys.argv[1:]] or [100, 1000, 5000]:
    nodes, el, loads, fixed, mv, md = warren_mesh(panels)
    model = FEMModel(nodes=nodes, elements=el, material=Material(2.0, 0.5, 1.0), loads=loads, fixed_dofs=fixed, dimension=2)
    cfg = SolverConfig(max_iterations=10, tolerance=1e-10)
    solve_nr(model, cfg, 1.0) if panels <= 100 else None     # warm-up (library load, allocator)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = solve_nr(model, cfg, 1.0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    row = {"panels": panels, "elements": len(el), "dofs": 2 * len(nodes), "hip_seconds": dt, "converged": bool(res.converged),
           "nr_iterations": res.history[-1]["iterations"]}
    if 2 * len(nodes) <= 4100:
        from oracle import pinn_oracle as orc
        pb = orc.Problem(nodes=nodes, elements=el, loads=loads, fixed_dofs=fixed, dimension=2, young=2.0, area=0.5, density=1.0)
        t0 = time.perf_counter(); ref = orc.solve_nr(pb, orc.SolverConfig(max_iterations=10, tolerance=1e-10), 1.0)
        row["dense_numpy_seconds"] = time.perf_counter() - t0
        row["rel_err_u"] = float(np.max(np.abs(res.displacements - ref.displacements)) / np.max(np.abs(ref.displacements)))
    rows.append(row)
    print(json.dumps(row), flush=True)
