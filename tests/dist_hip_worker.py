"""Rehearsal of the multi-GPU HIP path on ONE GPU: N ranks (gloo collectives, device tensors staged
through the host) all using cuda:0.  Runs the product solver end to end and writes rank 0's result."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
os.environ.setdefault("PINNFEM_QUIET", "1")


def random_case(seed):
    """(model, config, measured values, measured dofs) of the random-truss case; solve_gd(*random_case(seed))."""
    from dist_oracle_worker import random_truss
    from helpers import load_npz, product_model, theta_from
    from pinn_fem_amd.fem.solver import SolverConfig
    rec = load_npz("step_warren_EA.npz")
    nodes, el, loads, fixed, md, mv = random_truss(seed)
    model = product_model(nodes, el, loads, fixed, 2, (20, 15, None), (2.0, 0.5, 1.0), theta_from(rec))
    cfg = SolverConfig(max_iterations=20, learning_rate_u=1e-3, learning_rate_theta=5e-4, tolerance=1e-12)
    return model, cfg, mv, md, 0.7


def main():
    kind, out = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    torch.cuda.set_device(0)
    from helpers import load_npz, load_run, product_example, product_model, theta_from
    from pinn_fem_amd.fem.solver import SolverConfig, solve, solve_gd
    if kind.startswith("example"):
        run = load_run(kind)
        theta0 = [np.array(t, dtype=np.float32) for t in run["theta0"]]
        parsed = product_example(kind, theta0 if theta0 else None)
        md = parsed["measured_data"]
        res = solve(parsed["model"], parsed["solver_config"], md.get("values"), md.get("dofs"))
    elif kind == "warren":  # the 19-element Warren truss fixture (E and A nets), 20 iterations: a shared node with
        # several incident elements on both sides, i.e. more than one interface element per shared node
        rec = load_npz("step_warren_EA.npz")
        model = product_model(rec["nodes"], rec["elements"], rec["loads"], rec["fixed"], 2, (20, 15, None),
                              (2.0, 0.5, 1.0), theta_from(rec))
        cfg = SolverConfig(max_iterations=20, learning_rate_u=1e-3, learning_rate_theta=5e-4, tolerance=1e-12)
        res = solve_gd(model, cfg, rec["meas_vals"], rec["meas_dofs"], target_load_factor=0.7,
                       u_initial=torch.from_numpy(rec["u"]))
    elif kind.startswith("rand"):  # the Warren fixture's nets on a random truss with shuffled element order
        res = solve_gd(*random_case(int(kind[4:] or 0)))
    else:  # a 300-element chain, 25 iterations from the fixture state
        rec = load_npz("step_chain300_ex4shape.npz")
        model = product_model(rec["nodes"], rec["elements"], rec["loads"], rec["fixed"], 2, (20, 15, 10),
                              (1.0, 1.0, 1.0), theta_from(rec))
        cfg = SolverConfig(max_iterations=25, learning_rate_u=0.01, learning_rate_theta=5e-4, tolerance=1e-12)
        res = solve_gd(model, cfg, rec["meas_vals"], rec["meas_dofs"], target_load_factor=0.7,
                       u_initial=torch.from_numpy(rec["u"]))
    if rank == 0:
        payload = {"u": res.displacements.flatten().tolist(), "reactions": res.reactions.flatten().tolist(),
                   "converged": bool(res.converged), "n_history": len(res.history),
                   "loss": [h["loss_total"] for h in res.history],
                   "theta": ({k: v.reshape(-1).tolist() for k, v in res.nn_parameters.items()}
                             if res.nn_parameters else {})}
        with open(out, "w") as f:
            json.dump(payload, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
