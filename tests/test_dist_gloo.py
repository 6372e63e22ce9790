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


# world 8 on the 37-element chain: 4-5 own elements per rank, interface rings of neighbouring cuts two elements apart
# (the rank count of the driver's full-node run); world 4 on the Warren girder: shared nodes of degree 4; random
# trusses with shuffled element order (dist_oracle_worker.random_truss): cuts through nodes of any degree
@pytest.mark.parametrize("kind,world,port", [("chain", 2, 29611), ("chain", 3, 29612), ("warren", 2, 29613),
                                             ("warren", 4, 29615), ("chain", 8, 29616),
                                             ("rand1", 3, 29617), ("rand2", 5, 29618)])
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
    from pinn_fem_amd.dist import chain_shard, element_ranges, partition_mesh
    assert element_ranges(10, 3) == [(0, 4), (4, 7), (7, 10)]
    el = np.stack([np.arange(10), np.arange(1, 11)], axis=1)
    shards = [partition_mesh(el, 11, 2, r, 3) for r in range(3)]
    # nodes 4 and 7 are shared; interface = their element rings: nodes 3,4,5 and 6,7,8 -> 6 nodes x 2 dofs
    assert all(s.n_iface == 12 for s in shards)
    # ghost elements: rank 0 holds element 4, rank 1 elements 3 and 7, rank 2 element 6
    assert list(shards[0].elems_global) == [0, 1, 2, 3, 4] and (shards[0].own_lo, shards[0].own_hi) == (0, 4)
    assert list(shards[1].elems_global) == [3, 4, 5, 6, 7] and (shards[1].own_lo, shards[1].own_hi) == (1, 4)
    assert list(shards[2].elems_global) == [6, 7, 8, 9] and (shards[2].own_lo, shards[2].own_hi) == (1, 4)
    assert list(shards[0].nodes_global) == [0, 1, 2, 3, 4, 5]
    assert list(shards[0].shared_slot) == [0, 1, 2, 3, 4, 5]                 # nodes 3, 4, 5
    assert list(shards[1].shared_slot) == list(range(12))                    # nodes 3, 4, 5, 6, 7, 8
    assert list(shards[2].shared_slot) == [6, 7, 8, 9, 10, 11]
    # owner = lowest rank owning an incident element; everything else is flagged (incl. pure ghost nodes)
    assert not shards[0].ghost_mask[:10].any() and shards[0].ghost_mask[10:].all()      # node 5: ghost on rank 0
    assert shards[1].ghost_mask[:4].all() and not shards[1].ghost_mask[4:10].any() and shards[1].ghost_mask[10:].all()
    # every global node owned exactly once
    owned = np.concatenate([s.nodes_global[~s.ghost_mask[0::2]] for s in shards])
    assert sorted(owned) == list(range(11))
    # the analytic chain shards of the bench are the same records
    for world, n in ((3, 5), (2, 4), (4, 3)):
        el = np.stack([np.arange(world * n), np.arange(1, world * n + 1)], axis=1)
        for r in range(world):
            a, b = partition_mesh(el, world * n + 1, 2, r, world), chain_shard(n, r, world)
            for f in ("elem_lo", "elem_hi", "own_lo", "own_hi", "n_iface"):
                assert getattr(a, f) == getattr(b, f), (f, world, n, r)
            for f in ("elems_global", "nodes_global", "elements_local", "shared_dofs", "shared_slot", "ghost_mask"):
                assert np.array_equal(getattr(a, f), getattr(b, f)), (f, world, n, r)
    # an irregular mesh: every element incident to a node of an own element is local
    rng = np.random.default_rng(0)
    n_nodes, el = 60, []
    for i in range(1, 60):
        for j in rng.choice(i, size=min(i, 2), replace=False):
            el.append((int(j), i))
    el = np.array(el)
    for r in range(4):
        s = partition_mesh(el, n_nodes, 2, r, 4)
        own_nodes = np.unique(el[s.elem_lo:s.elem_hi])
        touching = np.flatnonzero(np.isin(el, own_nodes).any(axis=1))
        assert set(touching) <= set(s.elems_global)


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


def test_partition_invariants_on_random_meshes():
    """Structural invariants of the ghost-element partition on random connected meshes and rank counts up to 8: what
    the one-collective schedule relies on (every element owned once, ghost ring complete, one global slot per
    interface dof shared by all ranks that hold it, exactly one owner per node, interface closed under 'other node of
    an element around a shared node')."""
    from pinn_fem_amd.dist import partition_mesh
    rng = np.random.default_rng(7)
    for trial in range(12):
        n_nodes = int(rng.integers(12, 90))
        el = [(int(rng.integers(0, i)), i) for i in range(1, n_nodes)]            # a random tree: connected
        for _ in range(int(rng.integers(0, n_nodes))):                           # + random chords
            a, b = rng.choice(n_nodes, size=2, replace=False)
            el.append((int(min(a, b)), int(max(a, b))))
        el = np.array(el)
        el = el[rng.permutation(len(el))]                                        # element order = ownership order
        dim = int(rng.integers(1, 3))
        world = int(rng.integers(2, 9))
        if len(el) < world:
            continue
        shards = [partition_mesh(el, n_nodes, dim, r, world) for r in range(world)]
        # every element owned exactly once, own range = contiguous slice of the local list
        owned = np.concatenate([s.elems_global[s.own_lo:s.own_hi] for s in shards])
        assert sorted(owned) == list(range(len(el)))
        for s in shards:
            assert list(s.elems_global[s.own_lo:s.own_hi]) == list(range(s.elem_lo, s.elem_hi))
            assert np.all(np.diff(s.elems_global) > 0)
        # shared nodes: touched by own elements of more than one rank
        touch = [set(np.unique(el[s.elem_lo:s.elem_hi])) for s in shards]
        count = np.zeros(n_nodes, dtype=int)
        for t in touch:
            count[list(t)] += 1
        shared = set(np.flatnonzero(count > 1))
        for r, s in enumerate(shards):
            my_shared = touch[r] & shared
            ring = set(np.flatnonzero(np.isin(el, list(my_shared)).any(axis=1))) if my_shared else set()
            assert set(s.elems_global) == set(range(s.elem_lo, s.elem_hi)) | ring          # own + complete ghost ring
        # the global interface: shared nodes and the other nodes of the elements around them
        iface_nodes = set(shared)
        for e in np.flatnonzero(np.isin(el, list(shared)).any(axis=1)) if shared else []:
            iface_nodes |= set(el[e])
        n_iface = dim * len(iface_nodes)
        slot_of = {}
        for s in shards:
            assert s.n_iface == n_iface
            gdofs = (np.asarray(s.nodes_global)[s.shared_dofs // dim] * dim + s.shared_dofs % dim)
            assert set(gdofs // dim) == iface_nodes & set(s.nodes_global)               # every local copy takes part
            for g, k in zip(gdofs, s.shared_slot):
                assert slot_of.setdefault(int(g), int(k)) == int(k)                       # same slot on every rank
        assert sorted(slot_of.values()) == list(range(n_iface))
        # exactly one owner per node among the ranks whose own elements touch it (isolated nodes do not occur)
        owners = np.zeros(n_nodes, dtype=int)
        for r, s in enumerate(shards):
            mine = np.asarray(s.nodes_global)[~s.ghost_mask[0::dim]]
            assert set(mine) <= touch[r]
            owners[mine] += 1
        assert np.all(owners[sorted(set().union(*touch))] == 1)


def test_rank_failure_ends_the_job_quickly(tmp_path):
    """when one rank raises mid-solve while its peer already waits in a collective, the failing
    rank's shutdown (pinn_fem_amd.cli.generic._shutdown_distributed(failed=True): no device synchronisation, abort of
    the communicators, no collective destroy) must let the process exit so that the launcher ends the job — the run
    must END (non-zero) well within the timeout instead of hanging in a teardown that waits for the lost collective."""
    import time
    runner = tmp_path / "fail_rank.py"
    runner.write_text(
        "import os, sys, time, torch\n"
        "import torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from pinn_fem_amd.cli import generic as g\n"
        "dist.init_process_group('gloo')\n"
        "rank = dist.get_rank()\n"
        "t = torch.ones(4)\n"
        "dist.all_reduce(t)                      # the group works\n"
        "failed = True\n"
        "try:\n"
        "    if rank == 1:\n"
        "        time.sleep(1.0)                 # the peer is inside the next collective by now\n"
        "        raise RuntimeError('rank-local failure mid-solve')\n"
        "    dist.all_reduce(t)                  # never matched by rank 1\n"
        "    failed = False\n"
        "finally:\n"
        "    g._shutdown_distributed(failed)\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29619", str(runner)]
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0
    assert "rank-local failure mid-solve" in r.stderr
    assert time.time() - t0 < 120
