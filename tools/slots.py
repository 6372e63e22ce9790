"""Per-kernel slot times (HIP events, eager launches) and graph-replay time per iteration of one configuration.
usage: slots.py [n_elems] [workload] [wg_mode] [mesh]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model
from pinn_fem_amd import _capi
from pinn_fem_amd.engine import HipEngine
from pinn_fem_amd.fem.solver import SolverConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
workload = sys.argv[2] if len(sys.argv) > 2 else "ex4"
wg = int(sys.argv[3]) if len(sys.argv) > 3 else None
mesh = sys.argv[4] if len(sys.argv) > 4 else "chain"
model, mv, md, widths = build_model(n, workload, mesh=mesh)
eng = HipEngine(model, mv, md, wg_mode=wg)
cfg = SolverConfig(max_iterations=400, tolerance=0.0, learning_rate_u=0.01, learning_rate_theta=5e-4)
eng.begin(None, 0.1, cfg, want_history=False)
eng.prepare_graph()
eng.iterate(20)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter(); eng.iterate(50); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 50)
ms = eng.iterate_timed(20)
print(f"wg={eng.wg_mode} n={n} {workload} {mesh}: graph {best*1e3:.4f} ms/iter; slots(us): " +
      ", ".join(f"{k}={v*1e3:.1f}" for k, v in zip(_capi.KERNEL_SLOT_NAMES, ms)) + f"; sum {ms.sum()*1e3:.1f}")
