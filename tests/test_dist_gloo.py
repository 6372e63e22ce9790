"""N>1 path on CPU: world_size 2 and 3 over gloo.  The product's partition + interface exchange +
collective schedule (pinn_fem_amd.dist) must reproduce the single-process result: sum of shards ==
whole, to reduction-order round-off."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT, orc, rel_err

sys.path.insert(0, os.path.join(ROOT, "tests"))


def _run(kind, world, n_iter, tmp_path, port):
    out = str(tmp_path / f"{kind}_{world}.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_oracle_worker.py"), kind, str(n_iter), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    with np.load(out) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("kind,world,port", [("chain", 2, 29611), ("chain", 3, 29612), ("warren", 2, 29613)])
def test_sharded_matches_single_process(kind, world, port, tmp_path):
    from dist_oracle_worker import build_problem
    n_iter = 15
    got = _run(kind, world, n_iter, tmp_path, port)
    pb = build_problem(kind)
    cfg = orc.SolverConfig(max_iterations=n_iter, learning_rate_u=0.01, learning_rate_theta=5e-4,
                           tolerance=1e-12)
    ref = orc.solve_gd(pb, cfg, 0.6)
    assert int(got["n_iface"]) > 0
    assert rel_err(got["u"], ref.displacements.flatten()) < 2e-5
    th = np.concatenate([t.reshape(-1) for t in pb.theta_list()])
    assert rel_err(got["theta"], th) < 2e-5
    assert rel_err(got["loss"], [h["loss_total"] for h in ref.history]) < 2e-5
    assert rel_err(got["rn"], [h["residual_norm"] for h in ref.history]) < 2e-5
    assert rel_err(got["un"], [h["u_norm"] for h in ref.history]) < 2e-5


def test_partition_properties():
    from pinn_fem_amd.dist import element_ranges, partition_mesh
    assert element_ranges(10, 3) == [(0, 4), (4, 7), (7, 10)]
    el = np.stack([np.arange(10), np.arange(1, 11)], axis=1)
    shards = [partition_mesh(el, 11, 2, r, 3) for r in range(3)]
    assert all(s.n_iface == 4 for s in shards)            # nodes 4 and 7 are shared: 2 dofs each
    assert list(shards[0].shared_slot) == [0, 1] and list(shards[1].shared_slot) == [0, 1, 2, 3]
    assert list(shards[2].shared_slot) == [2, 3]
    assert not shards[0].ghost_mask.any()                 # lowest sharing rank owns
    assert shards[1].ghost_mask[:2].all() and not shards[1].ghost_mask[2:].any()
    # every global node owned exactly once
    owned = np.concatenate([s.nodes_global[~s.ghost_mask[0::2]] for s in shards])
    assert sorted(owned) == list(range(11))


def test_broadcast_theta_makes_replicas_identical(tmp_path):
    """Networks built from an unseeded RNG differ per process; the sharded path replicates theta, so every
    rank must start from rank 0's values (the CLI under torchrun never reconciled them)."""
    out = str(tmp_path / "bcast.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
           "--master-addr", "127.0.0.1", "--master-port", "29614",
           os.path.join(ROOT, "tests", "dist_bcast_worker.py"), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    with np.load(out) as z:
        before, after = z["before"], z["after"]
    assert not np.array_equal(before[0], before[1]) and not np.array_equal(before[0], before[2])
    for k in range(3):
        assert np.array_equal(after[k], before[0])     # bit for bit rank 0's vector
