#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE implementation (CPU PyTorch).

This script is the only place that imports /root/reference. It runs in the build
container only (the reference never travels to the GPU box); what it writes under
tests/golden/ is plain data (npz / json): inputs and the outputs the reference produced
for them.  Nothing here is imported by the product or by the tests.

    python tests/golden/make_golden.py [--only NAME ...]

Reference entry points exercised (paths relative to /root/reference):
  FEM/python/examples/json/generic.py:145  parse_problem
  FEM/python/examples/json/generic.py:447  solve_problem
  FEM/python/fem/solver.py:83              solve_gd
  FEM/python/fem/solver.py:1045            solve
  FEM/python/fem/nn_assembly.py:105        assemble_system_torch
The reference never seeds its RNG; every capture sets torch.manual_seed(S) right before
parse_problem() (which constructs the nets young -> area -> density) and stores the
initial parameters explicitly.
"""
import argparse
import contextlib
import importlib.util
import io
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

REF = "/root/reference/FEM/python"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

import fem.solver as ref_solver  # noqa: E402
from fem.model import FEMModel, Material  # noqa: E402
from fem.nn_assembly import assemble_system_torch  # noqa: E402
from fem.properties import NNProperty  # noqa: E402
from fem.boundary import free_and_fixed_dofs  # noqa: E402

_spec = importlib.util.spec_from_file_location(
    "ref_generic", os.path.join(REF, "examples/json/generic.py"))
ref_generic = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ref_generic)

EXAMPLES = ["example2", "example2-P", "example3", "example3-P", "example4", "example4-P",
            "example6", "example6-P", "example7", "example7-P",
            # classical Newton-Raphson (solver_type fem) and hybrid with scalar materials (GD -> NR switch)
            "example1", "example1-1", "example5", "example5-P"]


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def theta_arrays(model):
    return [p.detach().numpy().copy() for p in model.material.get_all_torch_params()]


def parse_seeded(json_path, seed):
    torch.manual_seed(seed)
    with quiet():
        return ref_generic.parse_problem(json_path)


def save_npz(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name} ({os.path.getsize(path)} B)")


# ----------------------------------------------------------------------------------
# single-step record: everything one GD iteration computes before the optimiser step
# (fem/solver.py:262-289) plus dense K / diag(K) from assemble_system_torch
# ----------------------------------------------------------------------------------
def single_step(model, u_np, lam, meas_vals, meas_dofs, alpha_p, alpha_d):
    u = torch.tensor(u_np, dtype=torch.float32, requires_grad=True)
    theta = model.material.get_all_torch_params()
    for p in theta:
        p.grad = None
    k_global, f_int = assemble_system_torch(model, u, load_factor=lam)
    f_ext = torch.tensor(model.loads, dtype=torch.float32)
    free, fixed = free_and_fixed_dofs(model.ndof, model.fixed_dofs)
    free_t = torch.tensor(free, dtype=torch.long)
    r = f_int[free_t] - lam * f_ext[free_t]
    loss_p = 0.5 * torch.sum(r ** 2)
    has_meas = meas_vals is not None and len(meas_vals) > 0 and alpha_d > 0
    if has_meas:
        mv = torch.tensor(meas_vals, dtype=torch.float32)
        md = torch.tensor(meas_dofs, dtype=torch.long)
        d = mv - u[md]
        loss_d = torch.mean(d ** 2)
        loss = alpha_p * loss_p + alpha_d * loss_d
    else:
        loss_d = torch.tensor(0.0)
        loss = alpha_p * loss_p
    loss.backward()
    out = {
        "u": np.asarray(u_np, dtype=np.float32),
        "lam": np.float64(lam),
        "f_int": f_int.detach().numpy().copy(),
        "r": r.detach().numpy().copy(),
        "free": free.astype(np.int64),
        "loss_physics": np.float32(loss_p.item()),
        "loss_data": np.float32(loss_d.item()),
        "loss_total": np.float32(loss.item()),
        "residual_norm": np.float32(torch.norm(r).item()),
        "grad_u": u.grad.numpy().copy(),
        "K_diag": torch.diagonal(k_global).detach().numpy().copy(),
    }
    if model.ndof <= 64:
        out["K"] = k_global.detach().numpy().copy()
    for i, p in enumerate(theta):
        # density tensors never receive a gradient (nn_assembly.py:207-208)
        out[f"grad_theta_{i}"] = (p.grad.numpy().copy() if p.grad is not None
                                  else np.zeros(0, dtype=np.float32))
        out[f"grad_theta_{i}_is_none"] = np.bool_(p.grad is None)
    return out


def gen_init_and_steps(tmp):
    """(1) initial theta for seeds 0/1; (2) single-step records; (3) first Adam steps."""
    for ex in ("example3", "example4"):
        src = os.path.join(tmp, ex + ".json")
        for seed in (0, 1):
            parsed = parse_seeded(src, seed)
            th = theta_arrays(parsed["model"])
            save_npz(f"init_{ex}_seed{seed}.npz", **{f"param_{i}": a for i, a in enumerate(th)})

        parsed = parse_seeded(src, 0)
        model, cfg, md = parsed["model"], parsed["solver_config"], parsed["measured_data"]
        th0 = theta_arrays(model)
        states = {
            "cold": (np.zeros(8), 0.1),
            "analytic": (np.array([0, 0, 1, 0, 2, 0, 3, 0], dtype=float), 1.0),
            "mid": (np.array([0, 0, 0.31, 0, 0.58, 0, 0.93, 0], dtype=float), 0.55),
        }
        for name, (u_np, lam) in states.items():
            rec = single_step(model, u_np, lam, md["values"], md["dofs"],
                              cfg.alpha_physics, cfg.alpha_data)
            rec.update({f"theta_{i}": a for i, a in enumerate(th0)})
            rec["meas_vals"] = np.asarray(md["values"], dtype=np.float64)
            rec["meas_dofs"] = np.asarray(md["dofs"], dtype=np.int64)
            save_npz(f"step_{ex}_{name}.npz", **rec)

        # (3) first iterations through the reference's own solve_gd, capturing the
        # torch.optim.Adam instances it creates (solver.py:234-236)
        for n_it in (1, 3, 12):
            parsed = parse_seeded(src, 0)
            model, cfg, md = parsed["model"], parsed["solver_config"], parsed["measured_data"]
            cfg.max_iterations = n_it
            created = []
            orig_adam = torch.optim.Adam

            def wrap(*a, **k):
                o = orig_adam(*a, **k)
                created.append(o)
                return o

            ref_solver.torch.optim.Adam = wrap
            try:
                with quiet():
                    res = ref_solver.solve_gd(model, cfg, md["values"], md["dofs"],
                                              target_load_factor=0.1)
            finally:
                ref_solver.torch.optim.Adam = orig_adam
            rec = {f"theta0_{i}": a for i, a in enumerate(th0)}
            rec.update({f"theta_{i}": a for i, a in enumerate(theta_arrays(model))})
            rec["u"] = res.displacements.flatten().copy()
            rec["reactions"] = res.reactions.flatten().copy()
            opt_u, opt_t = created[0], created[1]
            st = opt_u.state[opt_u.param_groups[0]["params"][0]]
            rec["u_exp_avg"] = st["exp_avg"].numpy().copy()
            rec["u_exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()
            for i, p in enumerate(opt_t.param_groups[0]["params"]):
                s = opt_t.state.get(p, None)
                if s:
                    rec[f"theta_exp_avg_{i}"] = s["exp_avg"].numpy().copy()
                    rec[f"theta_exp_avg_sq_{i}"] = s["exp_avg_sq"].numpy().copy()
            for key in ("loss_total", "loss_physics", "loss_data", "u_norm",
                        "residual_norm", "theta_norm"):
                rec["hist_" + key] = np.array([h[key] for h in res.history])
            save_npz(f"adam_{ex}_it{n_it}.npz", **rec)


# ----------------------------------------------------------------------------------
# (4) whole runs through generic.parse_problem/solve_problem
# ----------------------------------------------------------------------------------
def gen_whole_runs(tmp, names):
    for ex in names:
        src = os.path.join(tmp, ex + ".json")
        parsed = parse_seeded(src, 0)
        th0 = theta_arrays(parsed["model"])
        calls = []
        orig = ref_solver.solve_gd

        def wrapper(model, config=None, measured_disp=None, measured_dofs=None,
                    target_load_factor=1.0, u_initial=None, skip_preconditioning=False):
            res = orig(model, config, measured_disp, measured_dofs, target_load_factor,
                       u_initial, skip_preconditioning)
            entry = {
                "load_factor": float(target_load_factor),
                "n_history": len(res.history),
                "converged": bool(res.converged),
                "skip_preconditioning": bool(skip_preconditioning),
                "max_iterations": int(config.max_iterations),
                "tolerance": float(config.tolerance),
                "preconditioning": bool(config.preconditioning),
                "u": [float(x) for x in res.displacements.flatten()],
                "last": res.history[-1] if res.history else None,
            }
            # keep the loss trajectory of leaf calls (those that actually iterate)
            if not (config.preconditioning and not skip_preconditioning):
                entry["loss_total"] = [h["loss_total"] for h in res.history]
                entry["residual_norm"] = [h["residual_norm"] for h in res.history]
            calls.append(entry)
            return res

        ref_solver.solve_gd = wrapper
        t0 = time.time()
        try:
            with quiet():
                out = ref_generic.solve_problem(parsed)
        finally:
            ref_solver.solve_gd = orig
        wall = time.time() - t0
        golden = {
            "example": ex, "seed": 0, "wall_s_reference_cpu": wall,
            "torch": torch.__version__, "threads": torch.get_num_threads(),
            "theta0": [a.tolist() for a in th0],
            "calls": calls,
            "result": out,
        }
        path = os.path.join(HERE, f"run_{ex}.json")
        with open(path, "w") as f:
            json.dump(golden, f)
        leaf = [c["n_history"] for c in calls
                if not (c["preconditioning"] and not c["skip_preconditioning"])]
        print(f"  wrote run_{ex}.json  leaf iteration counts {leaf} total {sum(leaf)} "
              f"wall {wall:.1f}s")


# ----------------------------------------------------------------------------------
# (5) medium / irregular / 1-D meshes: single-step records only
# ----------------------------------------------------------------------------------
def make_nets(in_dim, widths, scales):
    props = {}
    for name, h, sc in zip(("young", "area", "density"), widths, scales):
        if h is None:
            props[name] = sc
        else:
            net = ref_generic.SimpleNN(hidden_layers=2, neurons_per_layer=h, input_dim=in_dim)
            props[name] = NNProperty(net=net, input_dim=in_dim, enforce_positive=True, scale=sc)
    return Material(**props)


def gen_meshes():
    # collinear 2-D chain, example4 shape, N in {300, 1000}  (SURVEY 8(d) synthetic inputs)
    for n in (300, 1000):
        torch.manual_seed(0)
        mat = make_nets(3, (20, 15, 10), (1.0, 1.0, 1.0))
        nodes = np.stack([np.arange(n + 1, dtype=float), np.zeros(n + 1)], axis=1)
        elements = np.stack([np.arange(n), np.arange(1, n + 1)], axis=1)
        loads = np.zeros(2 * (n + 1))
        loads[2 * n] = 1.0
        fixed = np.array([0] + [2 * i + 1 for i in range(n + 1)])
        model = FEMModel(nodes=nodes, elements=elements, material=mat, loads=loads,
                         fixed_dofs=fixed, dimension=2)
        i = np.arange(n + 1, dtype=float)
        u = np.zeros(2 * (n + 1))
        lam = 0.7
        u[0::2] = lam * i * (1.0 + 0.01 * np.sin(i))
        u[0] = 0.0
        meas_dofs = np.array([d for k in range(1, n + 1) for d in (2 * k, 2 * k + 1)])
        meas_vals = np.array([v for k in range(1, n + 1) for v in (float(k), 0.0)])
        t0 = time.time()
        rec = single_step(model, u, lam, meas_vals, meas_dofs, 1.0, 100.0)
        rec.update({f"theta_{k}": a for k, a in enumerate(theta_arrays(model))})
        rec.update(nodes=nodes, elements=elements.astype(np.int64), loads=loads,
                   fixed=fixed.astype(np.int64), meas_vals=meas_vals,
                   meas_dofs=meas_dofs.astype(np.int64))
        save_npz(f"step_chain{n}_ex4shape.npz", **rec)
        print(f"    (reference fwd+bwd at N={n}: {time.time()-t0:.2f}s)")

    # irregular 2-D truss (Warren-type), node degree up to 5, E and A nets, non-zero y
    torch.manual_seed(3)
    mat = make_nets(3, (20, 15, None), (2.0, 0.5, 1.0))
    nb = 6
    bottom = [(float(k), 0.0) for k in range(nb)]
    top = [(k + 0.5, 0.8) for k in range(nb - 1)]
    nodes = np.array(bottom + top)
    el = []
    for k in range(nb - 1):
        el.append((k, k + 1))            # bottom chord
        el.append((k, nb + k))           # diagonal up
        el.append((nb + k, k + 1))       # diagonal down
    for k in range(nb - 2):
        el.append((nb + k, nb + k + 1))  # top chord
    elements = np.array(el)
    nn_ = len(nodes)
    loads = np.zeros(2 * nn_)
    loads[2 * (nb - 1) + 1] = -1.0
    loads[2 * (nb + 2)] = 0.5
    fixed = np.array([0, 1, 2 * (nb - 1) + 1])
    model = FEMModel(nodes=nodes, elements=elements, material=mat, loads=loads,
                     fixed_dofs=fixed, dimension=2)
    rng = np.random.default_rng(7)
    u = rng.normal(scale=0.05, size=2 * nn_)
    u[fixed] = 0.0
    meas_dofs = np.array([4, 5, 9, 14, 15, 20])
    meas_vals = rng.normal(scale=0.05, size=len(meas_dofs))
    rec = single_step(model, u, 0.6, meas_vals, meas_dofs, 1.0, 100.0)
    rec.update({f"theta_{k}": a for k, a in enumerate(theta_arrays(model))})
    rec.update(nodes=nodes, elements=elements.astype(np.int64), loads=loads,
               fixed=fixed.astype(np.int64), meas_vals=meas_vals,
               meas_dofs=meas_dofs.astype(np.int64),
               scales=np.array([2.0, 0.5, 1.0]))
    save_npz("step_warren_EA.npz", **rec)

    # 1-D list-format bar (dimension=1, NN input = [load_factor, x])  nn_assembly.py:129-179
    torch.manual_seed(5)
    mat = make_nets(2, (20, None, None), (3.0, 2.0, 1.0))
    nodes = np.array([0.0, 0.5, 1.25, 2.0, 3.0])
    elements = np.array([[0, 1], [1, 2], [2, 3], [3, 4]])
    loads = np.array([0, 0, 0.2, 0, 1.0])
    fixed = np.array([0])
    model = FEMModel(nodes=nodes, elements=elements, material=mat, loads=loads,
                     fixed_dofs=fixed, dimension=1)
    u = np.array([0.0, 0.1, 0.22, 0.31, 0.5])
    meas_dofs = np.array([2, 4])
    meas_vals = np.array([0.2, 0.55])
    rec = single_step(model, u, 0.8, meas_vals, meas_dofs, 1.0, 100.0)
    rec.update({f"theta_{k}": a for k, a in enumerate(theta_arrays(model))})
    rec.update(nodes=nodes, elements=elements.astype(np.int64), loads=loads,
               fixed=fixed.astype(np.int64), meas_vals=meas_vals,
               meas_dofs=meas_dofs.astype(np.int64), scales=np.array([3.0, 2.0, 1.0]))
    save_npz("step_bar1d_E.npz", **rec)

    # scalar-material chain (example2 shape) single step: no theta at all
    mat = Material(young=1.0, area=1.0, density=1.0)
    nodes = np.stack([np.arange(4, dtype=float), np.zeros(4)], axis=1)
    elements = np.array([[0, 1], [1, 2], [2, 3]])
    loads = np.zeros(8)
    loads[6] = 1.0
    fixed = np.array([0, 1, 3, 5, 7])
    model = FEMModel(nodes=nodes, elements=elements, material=mat, loads=loads,
                     fixed_dofs=fixed, dimension=2)
    rec = single_step(model, np.array([0, 0, 0.2, 0, 0.5, 0, 0.6, 0.0]), 0.3,
                      np.array([]), np.array([], dtype=int), 1.0, 0.0)
    save_npz("step_example2_scalar.npz", **rec)


# ----------------------------------------------------------------------------------
# (6) classical Newton-Raphson on the fixture meshes with scalar materials (fem/solver.py:408-512)
# ----------------------------------------------------------------------------------
def gen_nr_meshes():
    from fem.model import FEMModel, Material
    from fem.solver import SolverConfig, solve_nr
    for src, name, young, area, lam in (("step_warren_EA.npz", "nr_warren_scalar.npz", 2.0, 0.5, 0.7),
                                        ("step_chain300_ex4shape.npz", "nr_chain300_scalar.npz", 1.0, 1.0, 1.0)):
        with np.load(os.path.join(HERE, src)) as z:
            rec = {k: z[k] for k in z.files}
        nodes = np.asarray(rec["nodes"], dtype=float)
        elements = [tuple(int(v) for v in e) for e in rec["elements"]]
        loads = np.asarray(rec["loads"], dtype=float)
        fixed = [int(v) for v in rec["fixed"]]
        model = FEMModel(nodes=nodes, elements=elements, material=Material(young, area, 1.0), loads=loads,
                         fixed_dofs=fixed, dimension=2)
        cfg = SolverConfig(max_iterations=20, tolerance=1e-10)
        with quiet():
            res = solve_nr(model, cfg, target_load_factor=lam)
        save_npz(name, nodes=nodes, elements=np.asarray(elements), loads=loads, fixed=np.asarray(fixed),
                 young=young, area=area, lam=lam, tolerance=cfg.tolerance,
                 u=res.displacements.reshape(-1), reactions=res.reactions.reshape(-1),
                 iterations=res.history[-1]["iterations"], converged=res.converged)
        print(f"  wrote {name}: {len(elements)} elements, NR iterations {res.history[-1]['iterations']}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None,
                    help="subset of: steps runs meshes, or example names")
    args = ap.parse_args()
    todo = set(args.only) if args.only else {"steps", "runs", "meshes"}

    tmp = tempfile.mkdtemp(prefix="golden_")
    inputs_dir = os.path.join(HERE, "inputs")
    os.makedirs(inputs_dir, exist_ok=True)
    for ex in EXAMPLES:
        # the reference writes <stem>.log/.res.json next to its input: work on copies
        shutil.copy(os.path.join(REF, "examples/json", ex + ".json"), tmp)
        # the problem definitions are the inputs of the vectors: keep them with the outputs
        shutil.copy(os.path.join(REF, "examples/json", ex + ".json"), inputs_dir)

    if "steps" in todo:
        print("[steps]")
        gen_init_and_steps(tmp)
    if "meshes" in todo:
        print("[meshes]")
        gen_meshes()
    if "meshes" in todo or "nr" in todo:
        print("[nr meshes]")
        gen_nr_meshes()
    run_names = [e for e in EXAMPLES if "runs" in todo or e in todo]
    if run_names:
        print("[runs]")
        gen_whole_runs(tmp, run_names)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
