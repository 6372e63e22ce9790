"""World-size-N CPU worker (gloo): pinn_fem_amd.dist.broadcast_theta makes deliberately different per-rank
parameter vectors equal to rank 0's.  Launched by torch.distributed.run from tests/test_dist_gloo.py."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    out = sys.argv[1]
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    from pinn_fem_amd.dist import broadcast_theta
    torch.manual_seed(100 + rank)                 # what an unseeded process does: a different stream per rank
    flat = torch.rand(998)
    before = flat.clone()
    broadcast_theta(flat)
    gathered = [torch.zeros_like(flat) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, flat)
    firsts = [torch.zeros_like(before) for _ in range(dist.get_world_size())]
    dist.all_gather(firsts, before)
    if rank == 0:
        np.savez(out, after=torch.stack(gathered).numpy(), before=torch.stack(firsts).numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
