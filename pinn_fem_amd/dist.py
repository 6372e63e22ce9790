"""Multi-GPU: elements sharded across ranks, one process per GPU, torch.distributed (RCCL over xGMI).

The reference is single-process (no collectives, no devices); this module is the
build's answer to BASELINE.json's "shard elements across the 8 GPUs of one node".

Partition: contiguous element ranges.  A node whose elements live on several ranks is SHARED.  Besides its own
elements a rank holds one ring of GHOST elements: every element of another rank that touches one of its shared
nodes (and, with them, the ghost nodes at their far ends).  Consequences:
  * the internal force, the residual and dL/df_int of every node of an own element are complete on the rank itself —
    nothing is exchanged before the residual;
  * gradients are taken over the own elements only (MLP backward, grad_u), so every element contributes exactly once;
  * the INTERFACE nodes = shared nodes plus the other nodes of the elements around them.  Their grad_u is the sum of
    the shares of the ranks that own an incident element; every rank holding a copy (as a node of an own or of a ghost
    element) then applies the same Adam step, so the copies — unknowns and Adam moments — stay bit-identical.
Per GD iteration there is ONE small all-reduce, in stream order after the rank-local kernels:
      [share of grad_u on the interface dofs | grad_theta | sum r^2, sum d^2 | sum u_free^2 of the PREVIOUS iteration]
followed by Adam(theta), Adam(u) on the interface dofs and the iteration's bookkeeping (history row, stop test, next
Adam scalars) from the reduced sums.  The u-norm is a monitor only (not part of the stop test, solver.py:341-355), so
it travels one iteration late and completes the previous history row; a chunk ends with a one-float flush.
Everything else (u, f_int, Adam-u state, residual) stays sharded; theta and its Adam state are replicated and stay
bit-identical because every rank applies the same reduced gradient (the initial theta is broadcast from rank 0).
For a 1-D chain cut into G shards the interface has 3(G-1) nodes: ~6(G-1) floats + grad_theta (<= 4 kB) per
iteration regardless of N — latency-bound, as SURVEY.md §5/§8(e) predicts.

The iteration driver (`run_iterations`) only talks to a ShardBackend; the product backend is
HipShardBackend (HIP kernels).  Tests drive the same driver with a CPU reference backend over gloo.
"""
from __future__ import annotations

import contextlib
import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _capi
from .plan import HostPlan, build_host_plan


# ---------------------------------------------------------------------------------------------------
# partition (host logic, numpy)
# ---------------------------------------------------------------------------------------------------
@dataclass
class Shard:
    rank: int
    world: int
    elem_lo: int                 # own elements: global ids [elem_lo, elem_hi)
    elem_hi: int
    elems_global: np.ndarray     # global id of each LOCAL element (own + ghost), ascending
    own_lo: int                  # the own elements are the local range [own_lo, own_hi)
    own_hi: int
    nodes_global: np.ndarray     # global node id of each local node (ascending)
    elements_local: np.ndarray   # (n_local_elems, 2) in local node numbering
    shared_dofs: np.ndarray      # local dof indices of the interface nodes present here (ascending global dof)
    shared_slot: np.ndarray      # their positions in the global interface vector
    ghost_mask: np.ndarray       # bool per local dof: owned by another rank (lower sharer, or a pure ghost node)
    n_iface: int
    dim: int


def element_ranges(n_elems: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced element ranges [lo, hi) per rank."""
    base, rem = divmod(n_elems, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def partition_mesh(elements: np.ndarray, n_nodes: int, dim: int, rank: int, world: int) -> Shard:
    """Deterministic on every rank (same inputs -> same interface numbering).

    LOCALITY REQUIREMENT: the cut is by contiguous ELEMENT-ID ranges, so the element numbering must be spatially local
    (consecutive ids = neighbouring elements, as mesh generators and the synthetic chain / girder produce).  On a mesh
    numbered at random the interface grows to most of the mesh: the per-iteration all-reduce then carries O(N) floats
    and the single-block interface kernels (k_shard_pack / k_shard_update) become the critical path.  A warning is
    issued when the interface exceeds 5 % of the dofs (and 4096 floats); renumber the elements (e.g. along a
    space-filling curve or by reverse Cuthill-McKee) before sharding such a mesh."""
    elements = np.asarray(elements, dtype=np.int64)
    n_elems = elements.shape[0]
    ranges = element_ranges(n_elems, world)
    erank = np.empty(n_elems, dtype=np.int64)
    for r, (lo, hi) in enumerate(ranges):
        erank[lo:hi] = r
    rmin = np.full(n_nodes, world, dtype=np.int64)         # lowest / highest rank owning an incident element
    rmax = np.full(n_nodes, -1, dtype=np.int64)
    for col in range(2):
        np.minimum.at(rmin, elements[:, col], erank)
        np.maximum.at(rmax, elements[:, col], erank)
    shared = (rmax > rmin) & (rmax >= 0)
    # interface nodes: shared nodes and every node of an element that touches one (ascending global id)
    ring = shared[elements].any(axis=1) if n_elems else np.zeros(0, dtype=bool)
    iface = shared.copy()
    if ring.any():
        iface[elements[ring].reshape(-1)] = True
    iface_nodes = np.flatnonzero(iface)
    if rank == 0 and iface_nodes.size * dim > max(4096, 0.05 * n_nodes * dim):
        import warnings
        warnings.warn(f"sharding: the interface holds {iface_nodes.size * dim} of {n_nodes * dim} dofs — the element "
                      "numbering is not spatially local; every iteration all-reduces that many floats and updates them "
                      "in one block.  Renumber the elements before sharding (see partition_mesh).", RuntimeWarning)
    slot_of = -np.ones(n_nodes, dtype=np.int64)
    slot_of[iface_nodes] = np.arange(iface_nodes.size)
    lo, hi = ranges[rank]
    own_nodes = np.zeros(n_nodes, dtype=bool)
    if hi > lo:
        own_nodes[elements[lo:hi].reshape(-1)] = True
    # ghost elements: other ranks' elements touching a shared node of mine
    mine_shared = shared & own_nodes
    ghost = (mine_shared[elements].any(axis=1) & (erank != rank)) if n_elems else np.zeros(0, dtype=bool)
    local_mask = ghost.copy()
    local_mask[lo:hi] = True
    elems_global = np.flatnonzero(local_mask)               # ascending: ghosts below | own | ghosts above
    own_lo = int(np.searchsorted(elems_global, lo))
    own_hi = own_lo + (hi - lo)
    assert own_hi <= elems_global.size and (hi == lo or (elems_global[own_lo] == lo and elems_global[own_hi - 1] == hi - 1))
    loc = elements[elems_global]
    nodes_global = np.unique(loc.reshape(-1)) if loc.size else np.zeros(0, dtype=np.int64)
    elements_local = np.searchsorted(nodes_global, loc) if loc.size else np.zeros((0, 2), dtype=np.int64)
    comp = np.arange(dim)
    my_iface = iface[nodes_global]
    local_idx = np.flatnonzero(my_iface)
    shared_dofs = (local_idx[:, None] * dim + comp[None, :]).reshape(-1)
    shared_slot = (slot_of[nodes_global[local_idx]][:, None] * dim + comp[None, :]).reshape(-1)
    # owner of a node: the lowest rank that owns an incident element; everybody else holds a copy
    not_owner = rmin[nodes_global] != rank
    ghost_dofs = np.repeat(not_owner, dim)
    return Shard(rank=rank, world=world, elem_lo=lo, elem_hi=hi, elems_global=elems_global, own_lo=own_lo,
                 own_hi=own_hi, nodes_global=nodes_global, elements_local=elements_local,
                 shared_dofs=shared_dofs.astype(np.int32), shared_slot=shared_slot.astype(np.int32),
                 ghost_mask=ghost_dofs, n_iface=int(iface_nodes.size * dim), dim=dim)


def shard_host_plan(shard: Shard, nodes, loads, fixed_dofs, measured_disp, measured_dofs,
                    n_dofs_global: int) -> Tuple[HostPlan, int]:
    """Local HostPlan (local numbering) + the global measurement count."""
    dim = shard.dim
    ng = shard.nodes_global
    nodes = np.asarray(nodes, dtype=float)
    nodes_l = nodes[ng]
    comp = np.arange(dim)
    dofs_g = (ng[:, None] * dim + comp[None, :]).reshape(-1)            # global dof of each local dof
    loads_l = np.asarray(loads, dtype=float).reshape(-1)[dofs_g]
    g2l = -np.ones(n_dofs_global, dtype=np.int64)
    g2l[dofs_g] = np.arange(dofs_g.size)
    fixed_g = np.asarray(fixed_dofs, dtype=np.int64).reshape(-1)
    fixed_l = g2l[fixed_g]
    fixed_l = fixed_l[fixed_l >= 0]
    mv_l = md_l = None
    n_meas_global = 0
    if measured_disp is not None and measured_dofs is not None:
        md_g = np.asarray(measured_dofs, dtype=np.int64).reshape(-1)
        mv_g = np.asarray(measured_disp, dtype=float).reshape(-1)
        n_meas_global = int(md_g.size)
        md_loc = g2l[md_g]
        keep = md_loc >= 0
        keep[keep] &= ~shard.ghost_mask[md_loc[keep]]                  # the owner adds the data term
        md_l, mv_l = md_loc[keep], mv_g[keep]
    hp = build_host_plan(nodes_l, shard.elements_local, loads_l, fixed_l, dim, mv_l, md_l)
    hp.dof_flags[shard.shared_dofs] |= _capi.PF_DOF_SHARED
    hp.dof_flags[shard.ghost_mask] |= _capi.PF_DOF_GHOST
    hp.n_meas = n_meas_global
    return hp, n_meas_global


# ---------------------------------------------------------------------------------------------------
# iteration driver (backend-agnostic)
# ---------------------------------------------------------------------------------------------------
class ShardBackend:
    """What the driver needs from one rank's local problem.  All tensors live on `device`.
    Buffers: buf = [share of grad_u on the interface dofs (n_iface) | grad_theta (n_theta_active) |
                    sum r^2, sum d^2, sum u_free^2 of the previous iteration],
             u2  = [this rank's sum u_free^2 of the iteration just finished].
    Phases of one iteration (stream-ordered; see the module docstring):"""
    device: torch.device
    n_iface: int
    n_theta_active: int

    def forward(self): ...                                # nets on all local elements (own + ghost)
    def backward(self, buf: torch.Tensor, u2: torch.Tensor): ...   # residual, backward over own elements, pack buf
    def update_interior(self): ...                        # grad_u + Adam(u) of the dofs that are not interface dofs
    def update_shared(self, buf: torch.Tensor, u2: torch.Tensor): ...  # Adam(theta), interface dofs, bookkeeping; u2[0]
    def flush(self, u2_reduced: torch.Tensor): ...        # last iteration's u-norm into the history


def _all_reduce(t: torch.Tensor, group=None):
    """Sum over ranks, in place, ordered on the current stream.  RCCL reduces device tensors directly;
    with a gloo group (CPU tests, or several ranks sharing one GPU in the single-GPU rehearsal) device
    tensors are staged through the host."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)


# own RCCL communicator for the C iteration driver (pf_comm.hip): one per process and group
_RCCL_COMMS = {}


def _librccl_path() -> str:
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return cand if os.path.exists(cand) else "librccl.so"


def rccl_comm(backend, group=None):
    """The C driver's communicator, or None when the Python driver has to be used: not the HIP
    backend, not an RCCL ("nccl") process group, PINNFEM_SHARD_DRIVER=python, or creation failed on any
    rank (decided collectively, so every rank takes the same path)."""
    if not isinstance(backend, HipShardBackend) or not dist.is_initialized():
        return None
    if dist.get_backend(group) != "nccl" or os.environ.get("PINNFEM_SHARD_DRIVER", "c") == "python":
        return None
    key = id(group)
    if key in _RCCL_COMMS:
        return _RCCL_COMMS[key]
    import ctypes as C
    lib, dev = backend.eng.lib, backend.device
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    path = _librccl_path().encode()
    idbuf = (C.c_char * _capi.PF_COMM_ID_BYTES)()
    ok = 1
    if rank == 0 and lib.pf_comm_unique_id(path, idbuf) != 0:
        ok = 0
    t = torch.zeros(_capi.PF_COMM_ID_BYTES + 1, dtype=torch.uint8, device=dev)
    if rank == 0:
        t[:-1] = torch.frombuffer(bytearray(idbuf.raw), dtype=torch.uint8).to(dev)
        t[-1] = ok
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    host = t.cpu().numpy()
    comm = C.c_void_p()
    if int(host[-1]) == 1:
        with torch.cuda.device(dev):
            rc = lib.pf_comm_create(path, host[:-1].tobytes(), rank, world, C.byref(comm))
        ok = 1 if rc == 0 else 0
    else:
        ok = 0
    def all_ok(v):      # collective: 1 only if every rank says 1
        flag = torch.tensor([v], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return int(flag.item())

    ok = all_ok(ok)
    if ok:
        # trust, but verify: the communicator's all-reduce must give what torch.distributed's gives
        # (small integers: the sum is exact whatever the reduction order)
        probe = (torch.arange(16, dtype=torch.float32, device=dev) + 1.0) * float(rank + 1)
        want = probe.clone()
        dist.all_reduce(want, group=group)
        s = torch.cuda.current_stream(dev)
        rc = lib.pf_comm_all_reduce(comm, probe.data_ptr(), probe.numel(), s.cuda_stream)
        s.synchronize()
        ok = all_ok(1 if (rc == 0 and bool(torch.equal(probe, want))) else 0)
    if ok != 1:
        if comm.value:
            lib.pf_comm_destroy(comm)
        if rank == 0 and os.environ.get("PINNFEM_QUIET", "0") != "1":
            print("pinn_fem_amd: own RCCL communicator unavailable (%s); using the torch.distributed driver"
                  % lib.pf_last_error().decode(errors="replace"), flush=True)
        comm = None
    _RCCL_COMMS[key] = comm
    return comm


def shard_driver_info(backend, group=None) -> dict:
    """Which driver runs this backend's iterations and what the communicator reports: for the bench line."""
    out = {"shard_driver": getattr(backend, "driver_used", None), "comm_ranks": None, "comm_rank": None}
    comm = _RCCL_COMMS.get(id(group))
    if comm is not None:
        import ctypes as C
        r, n = C.c_int(-1), C.c_int(-1)
        if backend.eng.lib.pf_comm_info(comm, C.byref(r), C.byref(n)) == 0:
            out["comm_rank"], out["comm_ranks"] = int(r.value), int(n.value)
    elif dist.is_initialized():
        out["comm_ranks"] = dist.get_world_size(group)
        out["comm_rank"] = dist.get_rank(group)
    return out


def broadcast_theta(flat: torch.Tensor, group=None, check: bool = True):
    """Every rank takes rank 0's parameters (in place).  The sharded path REPLICATES theta and relies on the
    replicas being bit-identical (every rank applies the same reduced gradient); networks built from an
    unseeded torch RNG differ per process, so the replicas are made equal here, before anything is packed.
    check: assert with a MIN/MAX all-reduce that they now are."""
    if not dist.is_initialized() or dist.get_world_size(group) < 2 or flat.numel() == 0:
        return
    src = dist.get_global_rank(group, 0) if group is not None else 0
    staged = flat.is_cuda and dist.get_backend(group) == "gloo"
    h = flat.detach().cpu() if staged else flat.detach()
    dist.broadcast(h, src=src, group=group)
    if staged:
        with torch.no_grad():
            flat.copy_(h)
    if check:
        lo, hi = h.clone(), h.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        if not bool(torch.equal(lo, hi)):
            raise RuntimeError("theta differs between ranks after the broadcast")


def destroy_rccl_comms(failed: bool = False):
    """Release the C driver's communicators (call before torch.distributed.destroy_process_group).
    failed=True (this rank raised): abort them instead — no device synchronisation and no ncclCommDestroy, both of
    which would wait for a collective the peers will never match."""
    lib = _capi.load()
    for comm in list(_RCCL_COMMS.values()):
        if comm is not None:
            if failed:
                lib.pf_comm_abort(comm)
            else:
                torch.cuda.synchronize()
                lib.pf_comm_destroy(comm)
    _RCCL_COMMS.clear()


def run_iterations(backend: ShardBackend, n_iter: int, group=None,
                   bufs: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """n_iter GD iterations with ONE collective per iteration (+ a one-float flush at the end, so the device state is
    final when this returns).  On an RCCL process group the HIP backend's iterations are issued by the C driver
    (pf_shard_iterations: kernels and ncclAllReduce from one C loop, no Python between them); everything else goes
    through the Python schedule below."""
    if bufs is None:
        bufs = getattr(backend, "bufs", None) or make_buffers(backend)
    eng = getattr(backend, "eng", None)
    comm = rccl_comm(backend, group) if n_iter > 0 else None
    if n_iter > 0:
        backend.driver_used = "c-rccl" if comm is not None else "python"
    if comm is not None:
        graph = backend.shard_graph(comm, bufs, group) if hasattr(backend, "shard_graph") else None
        with eng.on_stream():
            if graph is not None:
                _capi.check(eng.lib.pf_shard_iterations_graph(eng._ref(), comm, graph, backend.SHARD_GRAPH_ITERS,
                                                              int(n_iter), bufs[0].data_ptr(), bufs[1].data_ptr(),
                                                              eng._stream()), "pf_shard_iterations_graph")
            else:
                _capi.check(eng.lib.pf_shard_iterations(eng._ref(), comm, int(n_iter), bufs[0].data_ptr(),
                                                        bufs[1].data_ptr(), eng._stream()), "pf_shard_iterations")
        backend.driver_used = "c-rccl" + ("+graph" if graph is not None else "")
        return bufs
    ctx = eng.on_stream() if eng is not None else contextlib.nullcontext()
    with ctx:   # kernels and collectives on one stream (the engine's)
        _run_iterations(backend, n_iter, group, bufs[0], bufs[1])
    return bufs


def _run_iterations(backend, n_iter, group, buf, u2):
    for _ in range(n_iter):
        backend.forward()
        backend.backward(buf, u2)
        backend.update_interior()
        _all_reduce(buf, group)                  # THE collective of the iteration
        backend.update_shared(buf, u2)
    if n_iter > 0:
        last = buf[-1:]                          # free slot: u2 keeps the LOCAL sum for the next chunk's first collective
        last.copy_(u2)
        _all_reduce(last, group)                 # flush: the last iteration's sum u^2
        backend.flush(last)


def make_buffers(backend: ShardBackend):
    dev = backend.device
    f32 = dict(dtype=torch.float32, device=dev)
    return (torch.zeros(backend.n_iface + backend.n_theta_active + 3, **f32), torch.zeros(1, **f32))


# ---------------------------------------------------------------------------------------------------
# HIP backend
# ---------------------------------------------------------------------------------------------------
class HipShardBackend(ShardBackend):
    """One rank's shard on its GPU: a HipEngine over the local sub-mesh (own + ghost elements) plus the interface maps."""

    def __init__(self, local_model, host_plan: HostPlan, shard: Shard, has_measurements: bool,
                 device=None, wg_mode=None, fe_mode=None):
        from .engine import HipEngine
        self.shard = shard
        self.eng = HipEngine(local_model, device=device, host_plan=host_plan, wg_mode=wg_mode, fe_mode=fe_mode,
                             iface=(shard.shared_dofs, shard.shared_slot, shard.n_iface, (shard.own_lo, shard.own_hi)))
        self.eng.has_measurements = has_measurements
        self.device = self.eng.device
        self.n_iface = shard.n_iface
        self.n_theta_active = self.eng.n_theta_active
        self.bufs = make_buffers(self)           # (buf, u2): fixed addresses

    # -- solve_gd-level control ---------------------------------------------------------------------
    def begin(self, u_initial_local, lam, config, want_history=True, group=None):
        if self.eng.n_theta:
            # replicas of theta must be bit-identical on every rank (see broadcast_theta)
            with self.eng.on_stream():
                broadcast_theta(self.eng.theta.flat, group)
        self.eng.begin(u_initial_local, lam, config, want_history=want_history)
        # grad_theta is reduced straight into buf, where the collective reads it
        self.eng.P.grad_theta = self.bufs[0].data_ptr() + 4 * self.n_iface
        self.bufs[0].zero_()
        self.bufs[1].zero_()
        self._drop_shard_graph()                 # it holds the previous pf_problem record by value

    # -- whole sharded iterations (kernels + the collective) as one hipGraph, C driver only -------------
    SHARD_GRAPH_ITERS = int(os.environ.get("PINNFEM_SHARD_GRAPH_ITERS", 10))
    _sgraph = None          # None: not tried yet for this begin(); False: unavailable; else the handle
    graph_creates = 0

    def _drop_shard_graph(self):
        if self._sgraph:
            self.eng.lib.pf_graph_destroy(self._sgraph)
        self._sgraph = None

    def shard_graph(self, comm, bufs=None, group=None):
        """Handle of pf_shard_graph_create for the current begin(), or None (not switched on, foreign buffers, or the
        capture failed on ANY rank — decided collectively, the collective is inside the graph).
        OPT-IN (PINNFEM_SHARD_GRAPH=1): on this pool the path can only be exercised on an RCCL group of ONE rank,
        where ncclAllReduce short-cuts and RCCL's graph-capture machinery is not entered at all; there it is
        bit-identical to the eager driver and 7 % faster (0.1835 vs 0.197 ms per iteration at 10^6 elements).  Until it
        has run on a multi-GPU node the default stays the eager C driver."""
        import ctypes as C
        if comm is None or os.environ.get("PINNFEM_SHARD_GRAPH", "0") != "1":
            return None
        if bufs is not None and (bufs[0].data_ptr() != self.bufs[0].data_ptr() or
                                 bufs[1].data_ptr() != self.bufs[1].data_ptr()):
            return None
        if self._sgraph is None:
            e = self.eng
            g = C.c_void_p()
            with e.on_stream():
                rc = e.lib.pf_shard_graph_create(e._ref(), comm, self.SHARD_GRAPH_ITERS, self.bufs[0].data_ptr(),
                                                 self.bufs[1].data_ptr(), e._stream(), C.byref(g))
            ok = torch.tensor([1 if rc == 0 and g.value else 0], dtype=torch.int32, device=self.device)
            if dist.is_initialized():
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 1:
                self._sgraph = g
                self.graph_creates += 1
            else:
                if rc == 0 and g.value:
                    e.lib.pf_graph_destroy(g)
                self._sgraph = False
        return self._sgraph or None

    def __del__(self):
        try:
            self._drop_shard_graph()
        except Exception:
            pass

    def state(self):
        return self.eng.state()

    def history(self, n):
        return self.eng.history(n)

    # -- ShardBackend (Python driver: gloo tests, single-GPU rehearsal, fallback): one C call per phase ----
    def _check_bufs(self, *tensors):
        buf, u2 = self.bufs
        spans = [(t.data_ptr(), t.data_ptr() + 4 * t.numel()) for t in (buf, u2)]
        for t in tensors:
            p = t.data_ptr()
            if t.numel() and not any(lo <= p < hi for lo, hi in spans):
                raise ValueError("HipShardBackend works on its own collective buffers (backend.bufs)")

    def forward(self):
        e = self.eng
        _capi.check(e.lib.pf_shard_forward(e._ref(), e._stream()), "pf_shard_forward")

    def backward(self, buf, u2):
        e = self.eng
        self._check_bufs(buf, u2)
        _capi.check(e.lib.pf_shard_backward(e._ref(), buf.data_ptr(), u2.data_ptr(), e._stream()), "pf_shard_backward")

    def update_interior(self):
        e = self.eng
        _capi.check(e.lib.pf_shard_update_interior(e._ref(), e._stream()), "pf_shard_update_interior")

    def update_shared(self, buf, u2):
        e = self.eng
        self._check_bufs(buf, u2)
        _capi.check(e.lib.pf_shard_update_shared(e._ref(), buf.data_ptr(), u2.data_ptr(), e._stream()),
                    "pf_shard_update_shared")

    def flush(self, u2_reduced):
        e = self.eng
        _capi.check(e.lib.pf_shard_flush(e._ref(), u2_reduced.data_ptr(), e._stream()), "pf_shard_flush")


# ---------------------------------------------------------------------------------------------------
# sharded solve_gd core (called by fem.solver.solve_gd when torch.distributed has world_size > 1)
# ---------------------------------------------------------------------------------------------------
def build_shard_backend(model, measured_disp, measured_dofs, rank: int, world: int, device=None,
                        wg_mode=None, fe_mode=None) -> HipShardBackend:
    from .fem.model import FEMModel
    shard = partition_mesh(model.elements, model.nnode, model.dimension, rank, world)
    has_meas = measured_disp is not None and measured_dofs is not None
    hp, _ = shard_host_plan(shard, model.nodes, model.loads, model.fixed_dofs, measured_disp,
                            measured_dofs, model.ndof)
    dim = model.dimension
    comp = np.arange(dim)
    dofs_g = (shard.nodes_global[:, None] * dim + comp[None, :]).reshape(-1)
    local_model = FEMModel(nodes=model.nodes[shard.nodes_global], elements=shard.elements_local,
                           material=model.material, loads=model.loads[dofs_g],
                           fixed_dofs=hp.fixed_dofs, dimension=dim)
    be = HipShardBackend(local_model, hp, shard, has_meas, device=device, wg_mode=wg_mode, fe_mode=fe_mode)
    be.dofs_global = dofs_g
    return be


def assembled_f_int(backend: "HipShardBackend", lam: float, group=None) -> torch.Tensor:
    """Local view of the assembled f_int (re-evaluating the nets).  With the ghost elements of every shared node in the
    local mesh it is complete on every node this rank owns (the only entries gather_global_vector takes); no exchange."""
    e = backend.eng
    with e.on_stream():
        return e.internal_force(lam=lam)


def gather_global_vector(local: torch.Tensor, backend, n_dofs_global: int, group=None) -> np.ndarray:
    """All ranks receive the full displacement vector (owned entries of every rank)."""
    world = dist.get_world_size(group)
    own = ~torch.from_numpy(backend.shard.ghost_mask)
    idx = torch.from_numpy(backend.dofs_global)[own]
    vals = local.detach().cpu()[own]
    objs = [None] * world
    dist.all_gather_object(objs, (idx.numpy(), vals.numpy()), group=group)
    out = np.zeros(n_dofs_global, dtype=np.float32)
    for i, v in objs:
        out[i] = v
    return out


# ---------------------------------------------------------------------------------------------------
# bench helper: weak-scaling chain, every rank builds only its own shard analytically
# ---------------------------------------------------------------------------------------------------
def chain_shard(n: int, rank: int, world: int) -> Shard:
    """The shard of rank `rank` of a world*n-element chain (element e joins nodes e, e+1), built analytically: the same
    record partition_mesh gives for that mesh (tests/test_dist_gloo.py checks it), without the global arrays."""
    if world > 1 and n < 3:
        raise ValueError("chain shards need at least 3 elements per rank")
    e0 = rank * n
    g_lo = e0 - 1 if rank > 0 else e0                     # first / last local ELEMENT (global id)
    g_hi = e0 + n if rank < world - 1 else e0 + n - 1
    elems_global = np.arange(g_lo, g_hi + 1)
    nodes_global = np.arange(g_lo, g_hi + 2)
    el = np.stack([elems_global - g_lo, elems_global - g_lo + 1], axis=1)
    own_lo = e0 - g_lo
    # interface nodes of cut k (between ranks k-1 and k, k = 1..world-1): k*n-1, k*n, k*n+1 -> slots 3(k-1)+0..2
    sd, ss, ghost_nodes = [], [], []
    for k in ([rank] if rank > 0 else []) + ([rank + 1] if rank < world - 1 else []):
        for j, node in enumerate((k * n - 1, k * n, k * n + 1)):
            loc = node - g_lo
            sd += [2 * loc, 2 * loc + 1]
            ss += [2 * (3 * (k - 1) + j), 2 * (3 * (k - 1) + j) + 1]
            owner = k - 1 if j < 2 else k               # lowest rank owning an incident element
            if owner != rank:
                ghost_nodes.append(loc)
    ghost = np.zeros(2 * nodes_global.size, dtype=bool)
    for loc in ghost_nodes:
        ghost[2 * loc:2 * loc + 2] = True
    order = np.argsort(sd)
    return Shard(rank=rank, world=world, elem_lo=e0, elem_hi=e0 + n, elems_global=elems_global, own_lo=own_lo,
                 own_hi=own_lo + n, nodes_global=nodes_global, elements_local=el,
                 shared_dofs=np.array(sd, dtype=np.int32)[order], shared_slot=np.array(ss, dtype=np.int32)[order],
                 ghost_mask=ghost, n_iface=6 * (world - 1), dim=2)


class ShardedChainEngine:
    """Rank r owns elements [r*n, (r+1)*n) of a world*n-element collinear truss (SURVEY.md §8d
    inputs).  Same kernels and collective as the general path; no global arrays are built."""

    def __init__(self, n_local: int, workload: str, rank: int, world: int, device):
        from .fem.model import FEMModel, Material
        from .fem.properties import NNProperty
        from .nets import SimpleNN
        n = int(n_local)
        n_total = n * world
        shard = chain_shard(n, rank, world)
        ng = shard.nodes_global
        x = ng.astype(np.float64)
        nodes_l = np.stack([x, np.zeros(ng.size)], axis=1)
        el = shard.elements_local
        loads = np.zeros(2 * ng.size)
        tip = n_total - ng[0]
        if 0 <= tip < ng.size:
            loads[2 * tip] = 1.0
        fixed = list(2 * np.arange(ng.size) + 1)
        if ng[0] == 0:
            fixed = [0] + fixed
        fixed = np.array(sorted(fixed))
        # measurements ux_i = x_i, uy_i = 0 at every global node >= 1: the owner adds the data term
        k = np.arange(ng.size)
        keep = (ng >= 1) & ~shard.ghost_mask[0::2]
        kk = k[keep]
        md = np.stack([2 * kk, 2 * kk + 1], axis=1).reshape(-1)
        mv = np.stack([x[kk], np.zeros(kk.size)], axis=1).reshape(-1)
        hp = build_host_plan(nodes_l, el, loads, fixed, 2, mv, md)
        hp.dof_flags[shard.shared_dofs] |= _capi.PF_DOF_SHARED
        hp.dof_flags[shard.ghost_mask] |= _capi.PF_DOF_GHOST
        hp.n_meas = 2 * n_total
        torch.manual_seed(0)                      # identical theta on every rank (begin() broadcasts rank 0's anyway)
        widths = {"ex4": (20, 15, 10), "ex3": (20, None, None)}[workload]
        props = [1.0 if w is None else NNProperty(SimpleNN(2, w, 3), input_dim=3, scale=1.0)
                 for w in widths]
        model = FEMModel(nodes=nodes_l, elements=el, material=Material(*props), loads=loads,
                         fixed_dofs=fixed, dimension=2)
        self.backend = HipShardBackend(model, hp, shard, True, device=device)
        self.bufs = self.backend.bufs

    def begin(self, u0, lam, config):
        self.backend.begin(u0, lam, config, want_history=False)

    def prepare(self):
        """Create the C driver's RCCL communicator (collective) ahead of the first iteration."""
        comm = rccl_comm(self.backend)
        graph = self.backend.shard_graph(comm) if comm is not None else None     # captured now, never inside a timed region
        self.backend.driver_used = ("c-rccl" + ("+graph" if graph is not None else "")) if comm is not None else "python"

    def iterate(self, n):
        run_iterations(self.backend, n, bufs=self.bufs)

    def iterate_timed(self, n):
        """Per-kernel HIP-event times of the rank-local kernels (collectives excluded): runs the
        single-GPU iteration sequence on this shard's engine; shared dofs see partial sums, which is
        irrelevant for timing.  bench.py uses it only for the roofline entry."""
        return self.backend.eng.iterate_timed(n)

    def state(self):
        return self.backend.state()
