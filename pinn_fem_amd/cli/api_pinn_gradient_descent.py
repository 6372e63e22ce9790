#!/usr/bin/env python3
"""JSON-in / JSON-out wrapper for scalar (E, A) identification by gradient descent — same surface as
the reference's FEM/python/api_pinn_gradient_descent.py (`python api_pinn_gradient_descent.py in.json
out.json`): input keys `parse_input` :22-89, output keys :166-176, on any exception an
{"error","type"} JSON is written to the output file and the exit code is 1 (:206-219).

The reference file cannot run (ImportError at :19); the solver it calls is provided by
pinn_fem_amd.fem.nn_solver_gd.pinn_inverse_problem_gd (HIP kernels; arithmetic parity unpinned).
"""
from __future__ import annotations

import json
import sys

import numpy as np


def parse_input(input_data):
    """api_pinn_gradient_descent.py:22-89 (incl. its `elif` chain: a node with both fixed_x and fixed_y
    but not `fixed` only gets its x dof fixed, :45-50)."""
    nodes = np.array([[n["x"], n["y"]] for n in input_data["nodes"]])
    n_dofs = len(nodes) * 2
    elements = np.array([[e["nodes"][0], e["nodes"][1]] for e in input_data["elements"]])
    material = input_data.get("material", {})
    fixed_dofs = []
    for i, node in enumerate(input_data["nodes"]):
        if node.get("fixed", False):
            fixed_dofs.extend([2 * i, 2 * i + 1])
        elif node.get("fixed_x", False):
            fixed_dofs.append(2 * i)
        elif node.get("fixed_y", False):
            fixed_dofs.append(2 * i + 1)
    measured_disp = input_data.get("measured_disp", [])
    measured_dofs = input_data.get("measured_dofs", [])
    if not measured_disp or not measured_dofs:
        raise ValueError("PINN requires measured_disp and measured_dofs for inverse problem")
    sc = input_data.get("solver_config", {})
    return {
        "nodes": nodes, "elements": elements,
        "f_ext": np.array(input_data.get("loads", [0.0] * n_dofs)),
        "fixed_dofs": fixed_dofs,
        "young_init": material.get("young", 210e9), "area_init": material.get("area", 0.01),
        "u_measured": np.array(measured_disp), "measured_dofs": np.array(measured_dofs, dtype=int),
        "n_iterations": sc.get("max_iterations", 500), "learning_rate": sc.get("learning_rate", 0.001),
        "alpha": sc.get("alpha", 1.0), "beta": sc.get("beta", 100.0),
        "young_bounds": sc.get("young_bounds", [1e9, 500e9]),
        "area_bounds": sc.get("area_bounds", [0.001, 0.1]),
        "n_dofs": n_dofs,
    }


def solve_pinn_gd(problem):
    """api_pinn_gradient_descent.py:92-176."""
    from ..fem.nn_solver_gd import pinn_inverse_problem_gd
    print("Starting PINN Gradient Descent solver...")
    print(f"  Measured DOFs: {len(problem['measured_dofs'])}")
    print(f"  Initial Young's modulus: {problem['young_init']:.3e} Pa")
    print(f"  Initial Area: {problem['area_init']:.6f} m²")
    print(f"  Iterations: {problem['n_iterations']}")
    print(f"  Learning rate: {problem['learning_rate']}")
    result = pinn_inverse_problem_gd(
        nodes=problem["nodes"], elements=problem["elements"], f_ext=problem["f_ext"],
        fixed_dofs=problem["fixed_dofs"], young_init=problem["young_init"],
        area_init=problem["area_init"], u_measured=problem["u_measured"],
        measured_dofs=problem["measured_dofs"], n_iterations=problem["n_iterations"],
        learning_rate=problem["learning_rate"], alpha=problem["alpha"], beta=problem["beta"],
        young_bounds=problem["young_bounds"], area_bounds=problem["area_bounds"])
    u_final, young_final = result["u_final"], result["young_final"]
    history = result["history"]
    stresses, strains = [], []
    nodes = problem["nodes"]
    for i, j in problem["elements"]:                       # :134-151 (engineering strain of the deformed bar)
        xi, yi = nodes[i]
        xj, yj = nodes[j]
        ui, uj = u_final[2 * i:2 * i + 2], u_final[2 * j:2 * j + 2]
        l0 = np.sqrt((xj - xi) ** 2 + (yj - yi) ** 2)
        l = np.sqrt((xj + uj[0] - xi - ui[0]) ** 2 + (yj + uj[1] - yi - ui[1]) ** 2)
        eps = (l - l0) / l0
        strains.append(float(eps))
        stresses.append(float(young_final * eps))
    return {
        "displacements": u_final.tolist(), "stresses": stresses, "strains": strains,
        "identified_params": {"young": float(young_final), "area": float(result["area_final"])},
        "convergence_history": [
            {k: h[k] for k in ("iteration", "loss_total", "loss_physics", "loss_data", "young", "area")}
            for h in history[::10]],
        "final_loss": float(history[-1]["loss_total"]) if history else None,
    }


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) != 3:
        print("Usage: python api_pinn_gradient_descent.py input.json output.json")
        sys.exit(1)
    input_file, output_file = argv[1], argv[2]
    print(f"Reading input from {input_file}")
    try:
        with open(input_file, "r") as f:
            input_data = json.load(f)
        result = solve_pinn_gd(parse_input(input_data))
        with open(output_file, "w") as f:
            json.dump(result, f, indent=2)
        print(f"[OK] Results written to {output_file}")
        print(f"  Identified Young's modulus: {result['identified_params']['young']:.3e} Pa")
        print(f"  Identified Area: {result['identified_params']['area']:.6f} m^2")
    except Exception as e:
        with open(output_file, "w") as f:
            json.dump({"error": str(e), "type": type(e).__name__}, f, indent=2)
        print(f"[ERROR] {e}")
        import traceback
        traceback.print_exc()
        sys.exit(1)


if __name__ == "__main__":
    main()
