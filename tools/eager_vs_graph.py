#!/usr/bin/env python3
"""Consistency check at scale: hipGraph replay vs eager launches vs the numpy oracle on one chain."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_model
from pinn_fem_amd.engine import HipEngine
from pinn_fem_amd.fem.solver import SolverConfig

def run(n, iters, use_graph):
    model, mv, md, _ = build_model(n, "ex4")
    cfg = SolverConfig(max_iterations=iters + 5, tolerance=0.0, learning_rate_u=0.01, learning_rate_theta=5e-4)
    eng = HipEngine(model, mv, md, device=torch.device("cuda", 0))
    eng.begin(None, 0.1, cfg, want_history=True)
    eng.iterate(iters, use_graph=use_graph)
    torch.cuda.synchronize()
    return eng.u.cpu().numpy(), eng.theta.flat.cpu().numpy(), eng.history(iters)

def oracle(n, iters):
    from oracle import pinn_oracle as orc
    from pinn_fem_amd.nets import SimpleNN
    from pinn_fem_amd.plan import chain_mesh
    nodes, elements, loads, fixed, mv, md = chain_mesh(n, 1.0)
    torch.manual_seed(0)
    props = [orc.NetParams([p.detach().numpy().copy() for p in SimpleNN(2, w, 3).parameters()]) for w in (20, 15, 10)]
    pb = orc.Problem(nodes=nodes, elements=elements, loads=loads, fixed_dofs=fixed, dimension=2,
                     young=props[0], area=props[1], density=props[2], measured_vals=mv, measured_dofs=md)
    cfg = orc.SolverConfig(max_iterations=iters, learning_rate_u=0.01, learning_rate_theta=5e-4, tolerance=0.0)
    res = orc.solve_gd(pb, cfg, 0.1)
    return res.displacements.reshape(-1), np.concatenate([t.reshape(-1) for t in pb.theta_list()]), res.history

n, iters = int(sys.argv[1]), int(sys.argv[2])
ug, tg, hg = run(n, iters, True)
ue, te, he = run(n, iters, False)
rel = lambda a, b: float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))) / max(np.max(np.abs(b)), 1e-30))
out = {"graph_vs_eager_u": rel(ug, ue), "graph_vs_eager_theta": rel(tg, te), "graph_vs_eager_loss": rel(hg[:, 0], he[:, 0])}
if len(sys.argv) > 3:
    uo, to, ho = oracle(n, iters)
    lo = [h["loss_total"] for h in ho]
    out.update({"graph_vs_oracle_u": rel(ug, uo), "eager_vs_oracle_u": rel(ue, uo), "graph_vs_oracle_theta": rel(tg, to),
                "eager_vs_oracle_theta": rel(te, to), "graph_vs_oracle_loss": rel(hg[:, 0], lo), "eager_vs_oracle_loss": rel(he[:, 0], lo)})
print(json.dumps(out))
if os.environ.get("REPEAT"):
    ug2, tg2, _ = run(n, iters, True)
    ue2, te2, _ = run(n, iters, False)
    print(json.dumps({"graph_vs_graph_u": rel(ug2, ug), "eager_vs_eager_u": rel(ue2, ue),
                      "graph_vs_graph_theta": rel(tg2, tg), "eager_vs_eager_theta": rel(te2, te)}))
    for k in (1, 2, 3, 5, 10, 11, 20):
        a = run(n, k, True); b = run(n, k, False)
        print(k, "iters: graph vs eager u", rel(a[0], b[0]), "theta", rel(a[1], b[1]))
