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
    out, res = {}, {}
    for drv in ("python", "c", "c+graph"):
        os.environ["PINNFEM_SHARD_DRIVER"] = drv.split("+")[0]
        os.environ["PINNFEM_SHARD_GRAPH"] = "1" if drv.endswith("graph") else "0"
        sh = ShardedChainEngine(n, "ex4", 0, 1, dev)
        sh.begin(None, 0.1, cfg)
        sh.prepare()
        sh.iterate(10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sh.iterate(steps)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        out["sharded_ms_per_iter_%s_driver" % drv] = (time.perf_counter() - t0) / steps * 1e3
        out["host_enqueue_ms_per_iter_%s_driver" % drv] = (t1 - t0) / steps * 1e3
        out["driver_used_%s" % drv] = sh.backend.driver_used
        res[drv] = (sh.backend.eng.u.cpu().numpy(), sh.backend.eng.theta.flat.cpu().numpy(), int(sh.state().iter))
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
    out["iters"] = [res["c"][2], int(eng.state().iter)]
    out["rel_err_u"] = float(max(np.max(np.abs(r[0] - u_1)) for r in res.values()) / np.max(np.abs(u_1)))
    out["rel_err_theta"] = float(max(np.max(np.abs(r[1] - th_1)) for r in res.values()) / np.max(np.abs(th_1)))
    out["python_driver_iters"] = res["python"][2]
    print(json.dumps(out))
    from pinn_fem_amd.dist import destroy_rccl_comms
    destroy_rccl_comms()
    dist.destroy_process_group()
    assert out["rel_err_u"] < 1e-5 and out["rel_err_theta"] < 1e-5, out


if __name__ == "__main__":
    main()
