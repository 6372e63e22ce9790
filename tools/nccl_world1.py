#!/usr/bin/env python3
"""Sharded (multi-GPU) iteration path exercised with the real RCCL backend on ONE rank: the same
HipShardBackend / run_iterations / torch.distributed all_reduce sequence bench.py uses for --gpus N,
world_size 1.  Checks the result against the single-engine path and prints ms per iteration of both,
i.e. the per-iteration cost of the sharded driver itself (extra kernels, host calls, collectives)."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29701")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    from bench import build_model
    from pinn_fem_amd.dist import ShardedChainEngine
    from pinn_fem_amd.engine import HipEngine
    from pinn_fem_amd.fem.solver import SolverConfig
    cfg = SolverConfig(max_iterations=10 + steps, tolerance=0.0, learning_rate_u=0.01,
                       learning_rate_theta=5e-4, alpha_physics=1.0, alpha_data=100.0)
    out = {}
    sh = ShardedChainEngine(n, "ex4", 0, 1, dev)
    sh.begin(None, 0.1, cfg)
    sh.iterate(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sh.iterate(steps)
    torch.cuda.synchronize()
    out["sharded_ms_per_iter"] = (time.perf_counter() - t0) / steps * 1e3
    u_sh = sh.backend.eng.u.cpu().numpy()
    th_sh = sh.backend.eng.theta.flat.cpu().numpy()
    st = sh.state()
    model, mv, md, _ = build_model(n, "ex4")
    eng = HipEngine(model, mv, md, device=dev)
    eng.begin(None, 0.1, cfg, want_history=False)
    eng.iterate(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.iterate(steps)
    torch.cuda.synchronize()
    out["single_ms_per_iter"] = (time.perf_counter() - t0) / steps * 1e3
    u_1 = eng.u.cpu().numpy()
    th_1 = eng.theta.flat.cpu().numpy()
    out["iters"] = [int(st.iter), int(eng.state().iter)]
    out["rel_err_u"] = float(np.max(np.abs(u_sh - u_1)) / np.max(np.abs(u_1)))
    out["rel_err_theta"] = float(np.max(np.abs(th_sh - th_1)) / np.max(np.abs(th_1)))
    print(json.dumps(out))
    dist.destroy_process_group()
    assert out["rel_err_u"] < 1e-5 and out["rel_err_theta"] < 1e-5, out


if __name__ == "__main__":
    main()
