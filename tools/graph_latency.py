"""Fixed cost of a chunk of graph-replayed iterations: host enqueue time and wall time of 10*r iterations
(r replays of the 10-iteration graph) after a device sync, for several r and graph lengths.
usage: graph_latency.py [n_elems]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model
from pinn_fem_amd.engine import HipEngine
from pinn_fem_amd.fem.solver import SolverConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
model, mv, md, widths = build_model(n, "ex4")
for gk in (10, 2, 1):
    eng = HipEngine(model, mv, md)
    eng.GRAPH_ITERS = gk
    cfg = SolverConfig(max_iterations=100000, tolerance=0.0, learning_rate_u=0.01, learning_rate_theta=5e-4)
    eng.begin(None, 0.1, cfg, want_history=False)
    eng.prepare_graph()
    eng.iterate(2 * gk)
    torch.cuda.synchronize()
    for iters in (20, 40, 100, 200):
        best = (1e9, 0)
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); eng.iterate(iters); t1 = time.perf_counter(); torch.cuda.synchronize()
            t2 = time.perf_counter()
            best = min(best, (t2 - t0, t1 - t0))
        print(f"graph of {gk:2d}: {iters:4d} iterations: wall {best[0]*1e3:.3f} ms = {best[0]/iters*1e3:.4f} ms/iter, "
              f"host enqueue {best[1]*1e3:.3f} ms")
    del eng
