"""World-size-N CPU worker (gloo) for tests/test_dist_gloo.py.

Drives the PRODUCT's sharding logic (pinn_fem_amd.dist: partition_mesh, shard_host_plan,
run_iterations and its collectives) with an oracle-backed ShardBackend standing in for the HIP
kernels, and writes rank 0's view of the result.  Launched by torch.distributed.run.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from oracle import pinn_oracle as orc  # noqa: E402
from pinn_fem_amd import _capi  # noqa: E402
from pinn_fem_amd.dist import (ShardBackend, make_buffers, partition_mesh, run_iterations,  # noqa: E402
                               shard_host_plan)

f32 = np.float32


class OracleShardBackend(ShardBackend):
    """The rank-local arithmetic by the numpy oracle on the local mesh (own + ghost elements) the PRODUCT's partition
    produced; the phases, buffers and the collective are the product's (pinn_fem_amd.dist.run_iterations)."""

    def __init__(self, pb_global: orc.Problem, cfg: orc.SolverConfig, lam: float, rank: int, world: int):
        self.device = torch.device("cpu")
        dim = pb_global.dimension
        self.shard = sh = partition_mesh(pb_global.elements, pb_global.nnode, dim, rank, world)
        hp, _ = shard_host_plan(sh, pb_global.nodes, pb_global.loads, pb_global.fixed_dofs,
                                pb_global.measured_vals, pb_global.measured_dofs, pb_global.ndof)
        self.hp = hp
        comp = np.arange(dim)
        self.dofs_global = (sh.nodes_global[:, None] * dim + comp[None, :]).reshape(-1)
        copy_net = lambda p: p.copy() if isinstance(p, orc.NetParams) else p
        mk = lambda elements: orc.Problem(nodes=pb_global.nodes[sh.nodes_global], elements=elements,
                                          loads=pb_global.loads[self.dofs_global], fixed_dofs=hp.fixed_dofs,
                                          dimension=dim, young=self.young, area=self.area, density=self.density)
        self.young, self.area, self.density = (copy_net(pb_global.young), copy_net(pb_global.area),
                                               copy_net(pb_global.density))
        self.pb = mk(sh.elements_local)                                   # own + ghost: forces and residual
        self.pb_own = mk(sh.elements_local[sh.own_lo:sh.own_hi])         # own only: gradients
        self.geo, self.geo_own = orc.element_geometry(self.pb), orc.element_geometry(self.pb_own)
        self.cfg, self.lam = cfg, lam
        self.n_iface = sh.n_iface
        self.theta = self.pb.theta_list()
        n_active = sum(t.size for p in (self.pb.young, self.pb.area) if isinstance(p, orc.NetParams)
                       for t in p.tensors)
        self.n_theta_active = n_active
        self.n_active_tensors = sum(len(p.tensors) for p in (self.pb.young, self.pb.area)
                                    if isinstance(p, orc.NetParams))
        self.flags = hp.dof_flags
        self.free = (self.flags & _capi.PF_DOF_FIXED) == 0
        self.owned = (self.flags & _capi.PF_DOF_GHOST) == 0
        self.meas = (self.flags & _capi.PF_DOF_MEASURED) != 0
        self.u = np.zeros(hp.n_dofs, dtype=f32)
        self.opt_u = orc.AdamState(lr=cfg.learning_rate_u)
        self.opt_t = orc.AdamState(lr=cfg.learning_rate_theta)
        self.use_data = pb_global.measured_vals is not None and cfg.alpha_data > 0
        self.m = float(hp.n_meas)
        self.history, self.done, self.converged, self.it = [], False, False, 0

    def forward(self):
        s, *_ = orc.element_stiffness(self.pb, self.geo, self.lam)        # all local elements
        self.f_int = orc.internal_force(self.geo, s, self.u, self.hp.n_dofs)

    def update_interior(self):  # folded into update_shared (same arithmetic, one Adam call)
        pass

    def backward(self, buf, u2):
        r = (self.f_int - f32(self.lam) * self.hp.f_ext).astype(f32)       # complete on every node of an own element
        r[~self.free] = 0
        self.g_f = (f32(self.cfg.alpha_physics) * r).astype(f32)
        self.r2 = float(np.sum((r * r)[self.owned], dtype=f32))
        d = (self.hp.meas_val - self.u).astype(f32)
        self.d2 = float(np.sum((d * d)[self.meas & self.owned], dtype=f32)) if self.use_data else 0.0
        gu, gt = orc.vjp_internal_force(self.pb_own, self.geo_own, self.u, self.lam, self.g_f)   # OWN elements only
        if self.use_data:
            gd = (f32(self.cfg.alpha_data) / f32(self.m)) * (f32(2.0) * d)
            gu[self.meas] += (-gd[self.meas]).astype(f32)               # the owner carries the MEASURED flag
        self.grad_u = gu
        b = buf.numpy()
        b[:] = 0
        b[self.shard.shared_slot] = gu[self.shard.shared_dofs]
        off = self.n_iface
        for g in gt:
            b[off:off + g.size] = g.reshape(-1)
            off += g.size
        tail = self.n_iface + self.n_theta_active
        b[tail:tail + 3] = (self.r2, self.d2, float(u2[0]))

    def update_shared(self, buf, u2):
        b = buf.numpy()
        if not self.done:
            self.grad_u[self.shard.shared_dofs] = b[self.shard.shared_slot]
            self.opt_u.update([self.u], [self.grad_u])
            grads, off = [], self.n_iface
            for i, t in enumerate(self.theta):
                if i < self.n_active_tensors:
                    grads.append(b[off:off + t.size].reshape(t.shape).copy())
                    off += t.size
                else:
                    grads.append(None)
            if self.theta:
                self.opt_t.update(self.theta, grads)
            self.u[~self.free] = 0
            u2.numpy()[0] = np.sum((self.u * self.u)[self.free & self.owned], dtype=f32)
        tail = self.n_iface + self.n_theta_active
        self._finalize(b[tail], b[tail + 1], b[tail + 2])

    def _finalize(self, r2, d2, u2_prev):
        if self.done:
            return
        lp = f32(0.5) * f32(r2)
        ld = f32(d2) / f32(self.m) if self.use_data else f32(0)
        loss = f32(self.cfg.alpha_physics) * lp + (f32(self.cfg.alpha_data) * ld if self.use_data else f32(0))
        rn = float(np.sqrt(f32(r2)))
        if self.history:
            self.history[-1]["u_norm"] = float(np.sqrt(f32(u2_prev)))     # the u-norm travels one iteration late
        self.history.append(dict(loss_total=float(loss), loss_physics=float(lp), loss_data=float(ld),
                                 residual_norm=rn, u_norm=0.0))
        if self.it > 10 and (rn < self.cfg.tolerance or float(loss) < self.cfg.tolerance):
            self.done = self.converged = True
        self.it += 1

    def flush(self, u2_reduced):
        if self.history:
            self.history[-1]["u_norm"] = float(np.sqrt(f32(float(u2_reduced[0]))))


def random_truss(seed, n_nodes=40):
    """A random connected planar truss (spanning tree + chords, element order shuffled, so rank boundaries cut through
    nodes of any degree): nodes, elements, loads, fixed dofs, measured dofs/values."""
    rng = np.random.default_rng(seed)
    nodes = np.stack([rng.uniform(0.0, 8.0, n_nodes), rng.uniform(0.0, 3.0, n_nodes)], axis=1)
    el = {(int(rng.integers(0, i)), i) for i in range(1, n_nodes)}
    while len(el) < 2 * n_nodes:
        a, b = rng.choice(n_nodes, size=2, replace=False)
        el.add((int(min(a, b)), int(max(a, b))))
    el = np.array(sorted(el))[rng.permutation(len(el))]
    fixed = np.unique(np.concatenate([[0, 1, 2, 3], rng.choice(np.arange(4, 2 * n_nodes), size=5, replace=False)]))
    loads = np.zeros(2 * n_nodes)
    loads[rng.choice(2 * n_nodes, size=6, replace=False)] = rng.uniform(-0.5, 0.5, 6)
    meas_dofs = np.setdiff1d(np.arange(2 * n_nodes), fixed)
    meas_vals = rng.normal(size=meas_dofs.size) * 0.05
    return nodes, el, loads, fixed, meas_dofs, meas_vals


def build_problem(kind):
    from helpers import load_npz, mesh_problem
    if kind.startswith("rand"):
        # the Warren fixture's networks (E net 20 wide, A net 15 wide) on a random truss
        base = mesh_problem(load_npz("step_warren_EA.npz"), (20, 15, None), (2.0, 0.5, 1.0))
        nodes, el, loads, fixed, md, mv = random_truss(int(kind[4:] or 0))
        return orc.Problem(nodes=nodes, elements=el, loads=loads, fixed_dofs=fixed, dimension=2, young=base.young,
                           area=base.area, density=base.density, measured_vals=mv, measured_dofs=md)
    if kind == "warren":
        return mesh_problem(load_npz("step_warren_EA.npz"), (20, 15, None), (2.0, 0.5, 1.0))
    rec = load_npz("step_chain300_ex4shape.npz")
    pb = mesh_problem(rec, (20, 15, 10))
    n = 37                                     # a 37-element prefix of the chain (odd split sizes)
    fixed = pb.fixed_dofs[pb.fixed_dofs < 2 * (n + 1)]
    keep = pb.measured_dofs < 2 * (n + 1)
    loads = np.zeros(2 * (n + 1))
    loads[2 * n] = 1.0
    return orc.Problem(nodes=pb.nodes[: n + 1], elements=pb.elements[:n], loads=loads, fixed_dofs=fixed,
                       dimension=2, young=pb.young, area=pb.area, density=pb.density,
                       measured_vals=pb.measured_vals[keep], measured_dofs=pb.measured_dofs[keep])


def main():
    kind, n_iter, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pb = build_problem(kind)
    cfg = orc.SolverConfig(max_iterations=n_iter, learning_rate_u=0.01, learning_rate_theta=5e-4,
                           tolerance=1e-12)
    be = OracleShardBackend(pb, cfg, 0.6, rank, world)
    run_iterations(be, n_iter, bufs=make_buffers(be))
    # gather owned displacements on every rank
    objs = [None] * world
    dist.all_gather_object(objs, (be.dofs_global[be.owned], be.u[be.owned]))
    if rank == 0:
        u = np.zeros(pb.ndof, dtype=f32)
        for idx, val in objs:
            u[idx] = val
        np.savez(out, u=u, theta=np.concatenate([t.reshape(-1) for t in be.theta]) if be.theta else np.zeros(0),
                 loss=np.array([h["loss_total"] for h in be.history]),
                 rn=np.array([h["residual_norm"] for h in be.history]),
                 un=np.array([h["u_norm"] for h in be.history]),
                 n_iface=be.n_iface)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
