#!/usr/bin/env python3
"""Generic JSON entry point — drop-in for the reference's
FEM/python/examples/json/generic.py (`python generic.py problem.json [output.json]`).

Same input keys, same precedence rules, same output schema, same file placement
(<stem>.log and <stem>.res.json next to the input, exit code 1 on any exception):
  setup_logging :67   parse_problem :145   solve_problem :447   extract_nn_properties :498   main :802
The solve itself runs through pinn_fem_amd.fem.solver (HIP kernels).
"""
from __future__ import annotations

import json
import logging
import sys
from datetime import datetime
from pathlib import Path

import numpy as np
import torch

from ..fem.model import FEMModel, Material
from ..fem.properties import NNProperty
from ..fem.solver import SolverConfig, solve
from ..nets import SimpleNN

logger = None


def setup_logging(problem_file):
    """generic.py:67-99: <stem>.log next to the input (mode 'w'), DEBUG level, file + stdout."""
    global logger
    log_file = Path(problem_file).parent / f"{Path(problem_file).stem}.log"
    logging.basicConfig(
        level=logging.DEBUG, format="%(asctime)s [%(levelname)s] %(message)s",
        handlers=[logging.FileHandler(log_file, mode="w", encoding="utf-8"),
                  logging.StreamHandler(sys.stdout)], force=True)
    logger = logging.getLogger(__name__)
    logger.info("=" * 60)
    logger.info("PINN-FEM Generic Solver Log")
    logger.info(f"Timestamp: {datetime.now().strftime('%Y-%m-%d %H:%M:%S')}")
    logger.info(f"Problem file: {problem_file}")
    logger.info(f"Log file: {log_file}")
    logger.info("=" * 60)
    return log_file


def log_print(msg="", level="info"):
    if logger:
        getattr(logger, level if level in ("debug", "warning", "error") else "info")(msg)
    else:
        print(msg)


def _make_property(name, nn_config, base_value):
    """generic.py:233-312: NNProperty(scale=base, enforce_positive=True) when enabled."""
    arch = nn_config.get(name, {})
    if not arch.get("enabled", False):
        log_print(f"[DEBUG] {name.capitalize()}: Scalar ({base_value})", level="debug")
        return base_value
    input_dim = arch.get("input_dim", 1)
    net = SimpleNN(
        hidden_layers=arch.get("hidden_layers", arch.get("hiddenLayers", 2)),
        neurons_per_layer=arch.get("neurons_per_layer", arch.get("neuronsPerLayer", 20)),
        input_dim=input_dim)
    log_print(f"[DEBUG] {name.capitalize()}: NNProperty (scale={base_value}, input_dim={input_dim})",
              level="debug")
    return NNProperty(net=net, input_dim=input_dim, enforce_positive=True, scale=base_value)


def parse_problem(problem_file):
    """JSON -> FEMModel + SolverConfig + measurements (generic.py:145-444)."""
    log_print("\n[DEBUG] Starting parse_problem...", level="debug")
    with open(problem_file, "r") as f:
        data = json.load(f)

    # ---- build-only extension, namespaced so that reference JSONs stay valid -----------
    # "accel": {"synthetic_chain": {"n_elements": N, "h": 1.0, "tip_load": 1.0},   mesh generator
    #           "compact_output": true|false|"auto",  big arrays -> <stem>.res.npz instead of JSON lists
    #           "fe_mode": "reference"|"delta",       element-force formulation
    #           "mlp_dtype": "f32"|"bf16"}            precision of the MLP matrix products (bf16 study)
    accel = data.get("accel", {})
    chain = accel.get("synthetic_chain")
    if chain and not data.get("nodes"):
        return _parse_synthetic_chain(data, accel, chain)

    nodes_list = data.get("nodes", [])
    if nodes_list and isinstance(nodes_list[0], list):           # :155-164
        nodes_array = np.array(nodes_list, dtype=float)
        problem_dim = nodes_array.shape[1]
        nodes = nodes_array.flatten() if problem_dim == 1 else nodes_array
    else:                                                        # :166-168
        nodes = np.array([[n["x"], n["y"]] for n in nodes_list])
        problem_dim = 2
    n_nodes = len(nodes_list) if nodes_list else 0
    n_dofs = n_nodes * problem_dim
    log_print(f"[DEBUG] Nodes: {n_nodes}, DOFs: {n_dofs}", level="debug")

    elements_data = data.get("elements", [])                     # :177-183
    if elements_data and isinstance(elements_data[0], list):
        elements = np.array(elements_data)
    else:
        elements = np.array([[e["nodes"][0], e["nodes"][1]] for e in elements_data])
    log_print(f"[DEBUG] Elements: {len(elements)}", level="debug")

    fixed_dofs_list = data.get("fixed_dofs", [])                 # :189-205
    if fixed_dofs_list:
        fixed_dofs = np.array(fixed_dofs_list, dtype=int)
    else:
        fixed = []
        if nodes_list and isinstance(nodes_list[0], dict):
            for i, node in enumerate(nodes_list):
                if node.get("fixed", False):
                    fixed.extend([2 * i, 2 * i + 1])
                else:
                    if node.get("fixed_x", False):
                        fixed.append(2 * i)
                    if node.get("fixed_y", False):
                        fixed.append(2 * i + 1)
        fixed_dofs = np.array(fixed, dtype=int)
    log_print(f"[DEBUG] Fixed DOFs: {fixed_dofs}", level="debug")

    f_ext = np.array(data.get("loads", [0.0] * n_dofs), dtype=float)   # :210-211

    material_data = data.get("material", {})                     # :216-219
    base = {"young": material_data.get("young", 210e9), "area": material_data.get("area", 0.01),
            "density": material_data.get("density", 7850)}
    nn_config = data.get("nn_config", {})
    solver_type = data.get("solver_type", "fem")
    # construction order young -> area -> density fixes the RNG consumption order
    material = Material(young=_make_property("young", nn_config, base["young"]),
                        area=_make_property("area", nn_config, base["area"]),
                        density=_make_property("density", nn_config, base["density"]))

    measured_data = {}
    if solver_type.startswith("pinn"):                           # :320-362
        measured_dofs, measured_values = [], []
        md = data.get("measured_displacements", None)
        if md:
            if "global_dof" in md and "measured_u" in md:
                measured_dofs, measured_values = md["global_dof"], md["measured_u"]
            else:
                ux, uy = md.get("ux", []), md.get("uy", [])
                for idx, node_id in enumerate(md.get("nodes", [])):
                    if idx < len(ux):
                        measured_dofs.append(2 * node_id)
                        measured_values.append(ux[idx])
                    if idx < len(uy):
                        measured_dofs.append(2 * node_id + 1)
                        measured_values.append(uy[idx])
        else:
            for i, node in enumerate(nodes_list):
                if not isinstance(node, dict):
                    raise AttributeError("'list' object has no attribute 'get'")  # as the reference
                ux_m, uy_m = node.get("measured_ux", 0), node.get("measured_uy", 0)
                if ux_m != 0:
                    measured_dofs.append(2 * i)
                    measured_values.append(ux_m)
                if uy_m != 0:
                    measured_dofs.append(2 * i + 1)
                    measured_values.append(uy_m)
        measured_data = {"dofs": np.array(measured_dofs, dtype=int),
                         "values": np.array(measured_values)}

    model = FEMModel(nodes=nodes, elements=elements, material=material, loads=f_ext,
                     fixed_dofs=fixed_dofs, dimension=problem_dim)

    solver_config = _solver_config_from(data)                                 # :377-428
    log_print(f"[DEBUG] Solver config: method={solver_config.method}, tol={solver_config.tolerance}, "
              f"max_iter={solver_config.max_iterations}", level="debug")
    log_print("[DEBUG] parse_problem completed successfully", level="debug")
    if accel.get("fe_mode") == "delta":
        model._pf_fe_mode = 1
    if accel.get("mlp_dtype"):
        model._pf_mlp_dtype = str(accel["mlp_dtype"])
    return {"model": model, "solver_config": solver_config, "measured_data": measured_data,
            "accel": accel}


def _solver_config_from(data):
    """SolverConfig with the reference's precedence rules (generic.py:377-428)."""
    sc, pc = data.get("solver_config", {}), data.get("pinn_config", {})
    solver_type = data.get("solver_type", "auto")
    explicit = sc.get("method", None)
    if explicit:
        method = explicit
    elif solver_type == "fem":
        method = "nr"
    elif solver_type in ["pinn-gd", "pinn"]:
        method = "gd"
    elif solver_type == "pinn-hybrid":
        method = "hybrid"
    else:
        method = "auto"
    return SolverConfig(
        max_iterations=pc.get("max_iterations", sc.get("max_iterations", 1000)),
        tolerance=pc.get("tolerance", sc.get("tolerance", 1e-6)),
        print_every=pc.get("print_every", 10),
        n_increments=sc.get("n_increments", 10),
        min_denominator=sc.get("min_denominator", 1e-10),
        learning_rate_u=sc.get("learning_rate_u", pc.get("learning_rate_u", 1e-7)),
        learning_rate_theta=sc.get("learning_rate_theta", pc.get("learning_rate_theta", 1e-4)),
        alpha_physics=pc.get("alpha_physics", 1.0),
        alpha_data=pc.get("alpha_data", 100.0),
        preconditioning=pc.get("preconditioning", sc.get("preconditioning", False)),
        method=method)


def _parse_synthetic_chain(data, accel, chain):
    """Mesh generator for large runs (SURVEY.md §8(d) inputs, §8(f) rank 2): collinear 2-D truss with
    nodes (i*h, 0), elements (e, e+1), node 0 ux fixed and every uy fixed, tip load, measurements
    ux_i = x_i / uy_i = 0 at every node >= 1 — the shape of the reference's example3/4 at any size,
    without a 100 MB node list in the JSON."""
    from ..plan import chain_mesh
    n = int(chain["n_elements"])
    h = float(chain.get("h", 1.0))
    nodes, elements, loads, fixed, mv, md = chain_mesh(n, h)
    loads[2 * n] = float(chain.get("tip_load", 1.0))
    material_data = data.get("material", {})
    base = {"young": material_data.get("young", 210e9), "area": material_data.get("area", 0.01),
            "density": material_data.get("density", 7850)}
    nn_config = data.get("nn_config", {})
    material = Material(young=_make_property("young", nn_config, base["young"]),
                        area=_make_property("area", nn_config, base["area"]),
                        density=_make_property("density", nn_config, base["density"]))
    model = FEMModel(nodes=nodes, elements=elements, material=material, loads=loads, fixed_dofs=fixed,
                     dimension=2)
    if accel.get("fe_mode") == "delta":
        model._pf_fe_mode = 1
    if accel.get("mlp_dtype"):
        model._pf_mlp_dtype = str(accel["mlp_dtype"])
    measured = {}
    if data.get("solver_type", "fem").startswith("pinn") and chain.get("measure_every_node", True):
        measured = {"dofs": md, "values": mv}
    log_print(f"[DEBUG] synthetic chain: {n} elements, h={h}", level="debug")
    return {"model": model, "solver_config": _solver_config_from(data), "measured_data": measured,
            "accel": accel}


def _eval_points(prop: NNProperty, pts: np.ndarray, lf, dimension: int) -> list:
    """Batched NNProperty evaluation at points (rows of x[, y]) — replaces the reference's
    per-point batch-1 loop (generic.py:545-581) with one device batch per point set."""
    pts = np.asarray(pts, dtype=float).reshape(len(pts), -1)
    cols = []
    if lf is not None:
        cols.append(np.full((len(pts), 1), lf))
    cols.append(pts[:, :1])
    if dimension >= 2:
        cols.append(pts[:, 1:2])
    x = np.column_stack(cols)
    dev = next(prop.net.parameters()).device
    with torch.no_grad():
        out = prop.net(torch.tensor(x, dtype=torch.float32, device=dev))
        if prop.enforce_positive:
            out = torch.nn.functional.softplus(out)
        out = out * prop.scale
    return [float(v) for v in out.reshape(-1).cpu().numpy()]


def extract_nn_properties(model, load_factors=None):
    """Identified properties at nodes and element centroids (generic.py:498-799)."""
    if load_factors is None:
        load_factors = [0.2, 0.5, 1.0]
    node_coords = model.nodes
    centroids = (node_coords[model.elements[:, 0]] + node_coords[model.elements[:, 1]]) / 2.0
    properties = {}
    for name in ("young", "area", "density"):
        prop = getattr(model.material, name)
        if not hasattr(prop, "net"):
            properties[name] = {"value": float(prop.value()), "type": "scalar"}
            continue
        cent_list = [c.tolist() for c in centroids]
        if prop.input_dim > model.dimension:
            variations = {}
            for lf in load_factors:
                variations[f"load_factor_{lf:.1f}"] = {
                    "at_nodes": {"coords": node_coords.tolist(),
                                 "values": _eval_points(prop, node_coords, lf, model.dimension)},
                    "at_elements": {"centroids": cent_list,
                                    "values": _eval_points(prop, centroids, lf, model.dimension)},
                }
            properties[name] = {"load_factor_variations": variations, "type": "nn_load_dependent",
                                "input_dim": prop.input_dim}
        else:
            properties[name] = {
                "at_nodes": {"coords": node_coords.tolist(),
                             "values": _eval_points(prop, node_coords, None, model.dimension)},
                "at_elements": {"centroids": cent_list,
                                "values": _eval_points(prop, centroids, None, model.dimension)},
                "type": "nn", "input_dim": prop.input_dim}
    return properties


def solve_problem(parsed_data):
    """generic.py:447-495."""
    model = parsed_data["model"]
    solver_config = parsed_data["solver_config"]
    measured_data = parsed_data.get("measured_data", {})
    log_print(f"\n{'='*60}")
    log_print("UNIFIED SOLVER")
    log_print(f"{'='*60}")
    log_print(f"Nodes: {len(model.nodes)}")
    log_print(f"Elements: {len(model.elements)}")
    log_print(f"Fixed DOFs: {len(model.fixed_dofs)}")
    log_print(f"Has NN: {model.material.has_trainable_params()}")
    log_print(f"Has measurements: {len(measured_data.get('dofs', [])) > 0}")
    log_print(f"Solver method: {solver_config.method}")
    result = solve(model=model, config=solver_config,
                   measured_disp=measured_data.get("values", None),
                   measured_dofs=measured_data.get("dofs", None))
    compact = parsed_data.get("accel", {}).get("compact_output", "auto")
    if compact == "auto":
        compact = model.nnode > 50_000
    output = {
        "success": result.converged,
        "converged": result.converged,
        "iterations": len(result.history),
        "history": result.history,
    }
    if compact:
        # §8(f) rank 1: at 10^6 nodes the reference's JSON lists (every coordinate three times per
        # property) would be hundreds of MB; arrays go to a side file, the JSON keeps the small fields
        arrays = {"displacements": result.displacements.flatten(),
                  "reactions": result.reactions.flatten()}
        if result.nn_parameters:
            output["nn_parameters"] = {k: v.tolist() for k, v in result.nn_parameters.items()}
            cent = (model.nodes[model.elements[:, 0]] + model.nodes[model.elements[:, 1]]) / 2.0
            for name in ("young", "area", "density"):
                prop = getattr(model.material, name)
                if hasattr(prop, "net"):
                    for lf in (0.2, 0.5, 1.0):
                        arrays[f"{name}_at_elements_lf{lf:.1f}"] = np.asarray(
                            _eval_points(prop, cent, lf if prop.input_dim > model.dimension else None,
                                         model.dimension), dtype=np.float32)
        output["arrays_npz"] = "__SIDE_FILE__"
        output["_arrays"] = arrays
        return output
    output["displacements"] = result.displacements.flatten().tolist()
    output["reactions"] = result.reactions.flatten().tolist() if result.reactions is not None else []
    if result.nn_parameters:
        output["nn_parameters"] = {k: v.tolist() for k, v in result.nn_parameters.items()}
        output["identified_properties"] = extract_nn_properties(model)
    return output


def _init_distributed():
    """`python -m torch.distributed.run --nproc-per-node N generic.py problem.json`: one process per GPU,
    elements sharded over the ranks (pinn_fem_amd/dist.py); every rank computes the same result, rank 0
    writes the files.  PINNFEM_DIST_BACKEND=gloo + PINNFEM_ONE_GPU=1 rehearses this on a single GPU."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    import torch.distributed as dist
    one_gpu = os.environ.get("PINNFEM_ONE_GPU", "0") == "1"
    local = 0 if one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        backend = os.environ.get("PINNFEM_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return dist.get_rank(), world


def _shutdown_distributed(failed: bool):
    """Tear the sharded run down.  Clean exit: device idle, the C driver's RCCL communicators, then torch's process
    group.  After a rank-local failure the other ranks may sit in — or this rank may already have enqueued — a
    collective that will never be matched: NO device synchronisation then (it would never return); abort the C
    driver's communicators and torch's backends first, skip the collective destroy, and let the process exit
    non-zero so that the launcher ends the peers."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return
    from ..dist import destroy_rccl_comms
    if failed:
        try:
            destroy_rccl_comms(failed=True)
        except Exception:
            pass
        try:
            pg = dist.distributed_c10d._get_default_group()
            for be_name in ("cuda", "cpu"):
                try:
                    be = pg._get_backend(torch.device(be_name))
                    if hasattr(be, "abort"):
                        be.abort()
                except Exception:
                    pass
        except Exception:
            pass
        return        # no destroy_process_group(): it is collective for some backends
    try:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        destroy_rccl_comms()
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) < 2:
        print("Usage: python generic.py problem.json [output.json]")
        sys.exit(1)
    rank, world = _init_distributed()
    failed = True
    try:
        _main_body(argv, rank, world)
        failed = False
    finally:
        if world > 1:
            _shutdown_distributed(failed)


def _main_body(argv, rank, world):
    problem_file = argv[1]
    if rank != 0:
        # non-zero ranks take part in the solve only: no log file, no result files
        try:
            solve_problem(parse_problem(problem_file))
        except Exception:
            import traceback
            traceback.print_exc()
            sys.exit(1)
        return
    log_file = setup_logging(problem_file)
    if len(argv) > 2:
        output_file = argv[2]
    else:
        p = Path(problem_file)
        output_file = str(p.parent / f"{p.stem}.res.json")
    log_print(f"Output file will be: {output_file}")
    log_print("=" * 60)
    try:
        log_print("\n[STEP 1] Parsing problem file...")
        parsed = parse_problem(problem_file)
        log_print("[OK] Problem parsed successfully")
        log_print("\n[STEP 2] Solving problem...")
        result = solve_problem(parsed)
        log_print("[OK] Problem solved")
        log_print("\n[STEP 3] Writing results...")
        arrays = result.pop("_arrays", None)
        if arrays is not None:
            side = str(Path(output_file).with_suffix("")) + ".npz"
            np.savez(side, **arrays)
            result["arrays_npz"] = Path(side).name
            result["displacement_abs_max"] = float(np.max(np.abs(arrays["displacements"])))
        with open(output_file, "w") as f:
            json.dump(result, f, indent=2)
        log_print(f"[OK] Results written to {output_file}")
        log_print(f"\n{'='*60}")
        log_print("SOLUTION SUMMARY:")
        if result.get("success"):
            log_print("  Status: SUCCESS")
            log_print(f"  Iterations: {result['iterations']}")
            max_u = (result["displacement_abs_max"] if "displacement_abs_max" in result
                     else max(abs(d) for d in result["displacements"]))
            log_print(f"  Max displacement: {max_u:.6e}")
        else:
            log_print("  Status: FAILED")
        log_print(f"{'='*60}")
        log_print("[SUCCESS] Solve completed successfully")
        log_print(f"{'='*60}\n")
        log_print(f"Log file saved: {log_file}")
    except Exception as e:  # generic.py:861-867: traceback to the log, exit code 1, no error JSON
        log_print(f"\n[ERROR] {e}", level="error")
        import traceback
        log_print(traceback.format_exc(), level="error")
        sys.exit(1)


if __name__ == "__main__":
    main()
