import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.argv = [sys.argv[0], "1000", "1"]
import importlib
n = int(os.environ.get("N", "1000000")); iters = int(os.environ.get("IT", "10"))
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
    return eng.u.cpu().numpy(), eng.theta.flat.cpu().numpy(), eng.m_u.cpu().numpy(), eng.v_u.cpu().numpy(), eng.history(iters)
for nn in (1000, 100000, n):
    a = run(nn, iters, True); b = run(nn, iters, False)
    d = np.abs(a[0] - b[0]); idx = np.flatnonzero(d > 0)
    print(nn, "u differing dofs", idx.size, "of", d.size, "first", idx[:8], "last", idx[-8:], "max", d.max() if d.size else 0)
    dm = np.abs(a[2] - b[2]); print("   m_u differing", np.count_nonzero(dm), "v_u differing", np.count_nonzero(np.abs(a[3]-b[3])),
          "theta differing", np.count_nonzero(a[1] != b[1]))
    print("   hist diff cols", [float(np.max(np.abs(a[4][:, c] - b[4][:, c]))) for c in range(6)])
