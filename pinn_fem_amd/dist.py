"""Multi-GPU: elements sharded across ranks, one process per GPU, torch.distributed (RCCL over xGMI).

The reference is single-process (no collectives, no devices); this module is the
build's answer to BASELINE.json's "shard elements across the 8 GPUs of one node".

Partition: contiguous element ranges.  A node whose elements live on several ranks is SHARED; every
sharing rank keeps a copy of its u and Adam state, and the lowest sharing rank OWNS it (counts it in
global sums, adds its data-loss term).  Per GD iteration there are TWO small all-reduces, the floor
for this partition (one exchange before the residual, one after the backward), in stream order:
  (1) [sum u_free^2 of the PREVIOUS iteration | partial f_int of the shared dofs]
      The few elements touching a shared node are evaluated first (phase A, one tiny kernel with the
      arithmetic of the full net kernels, bit for bit), so the exchange does not wait for the forward
      pass over all elements (phase B).
  (2) [partial grad_u of the shared dofs | grad_theta | sum r^2, sum d^2]
      after the backward (phase C) and grad_u + Adam(u) of every dof that is NOT shared (phase D);
      Adam(theta) and the shared dofs follow (phase E).
Measured on MI355X: hiding the two collectives on a second stream costs more than it saves (cross-stream
event pairs between eager launches ~9 us each on this runtime; the vectors are <= 4 kB, i.e. pure
latency), and so does one hipGraph replay per iteration; everything runs in stream order on one stream.
The u-norm is a monitor only, so it rides on the next iteration's first collective and the
bookkeeping kernel of iteration t (history row, stop test, next Adam scalars) runs right after that
collective, before anything of iteration t+1 that could change state; a chunk ends with one flush
collective.
Everything else (u, f_int, Adam-u state, residual) stays sharded; theta and its Adam state are
replicated and stay bit-identical because every rank applies the same reduced gradient.
For a 1-D chain cut into G shards the interface has 2(G-1) dofs, so the traffic is ~4 kB per
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
    elem_lo: int
    elem_hi: int
    nodes_global: np.ndarray     # global node id of each local node (ascending)
    elements_local: np.ndarray   # (n_local_elems, 2) in local node numbering
    shared_dofs: np.ndarray      # local dof indices shared with other ranks (ascending global dof)
    shared_slot: np.ndarray      # their positions in the global interface vector
    ghost_mask: np.ndarray       # bool per local dof: shared and owned by a lower rank
    n_iface: int
    dim: int
    iface_elems: Optional[np.ndarray] = None   # local element ids touching a shared node (ascending)


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
    """Deterministic on every rank (same inputs -> same interface numbering)."""
    elements = np.asarray(elements, dtype=np.int64)
    n_elems = elements.shape[0]
    ranges = element_ranges(n_elems, world)
    erank = np.empty(n_elems, dtype=np.int64)
    for r, (lo, hi) in enumerate(ranges):
        erank[lo:hi] = r
    rmin = np.full(n_nodes, world, dtype=np.int64)
    rmax = np.full(n_nodes, -1, dtype=np.int64)
    for col in range(2):
        np.minimum.at(rmin, elements[:, col], erank)
        np.maximum.at(rmax, elements[:, col], erank)
    shared_nodes = np.flatnonzero((rmax > rmin) & (rmax >= 0))       # ascending global id
    lo, hi = ranges[rank]
    loc = elements[lo:hi]
    nodes_global = np.unique(loc.reshape(-1)) if loc.size else np.zeros(0, dtype=np.int64)
    elements_local = np.searchsorted(nodes_global, loc) if loc.size else np.zeros((0, 2), dtype=np.int64)
    # interface numbering: shared node k -> slots k*dim .. k*dim+dim-1
    mine = np.isin(shared_nodes, nodes_global)
    my_shared_nodes = shared_nodes[mine]
    k_idx = np.flatnonzero(mine)
    local_idx = np.searchsorted(nodes_global, my_shared_nodes)
    comp = np.arange(dim)
    shared_dofs = (local_idx[:, None] * dim + comp[None, :]).reshape(-1)
    shared_slot = (k_idx[:, None] * dim + comp[None, :]).reshape(-1)
    ghost = np.zeros(len(nodes_global) * dim, dtype=bool)
    not_owner = rmin[my_shared_nodes] != rank
    ghost_nodes = local_idx[not_owner]
    if ghost_nodes.size:
        ghost[(ghost_nodes[:, None] * dim + comp[None, :]).reshape(-1)] = True
    touches = np.isin(loc, my_shared_nodes).any(axis=1) if loc.size else np.zeros(0, dtype=bool)
    return Shard(rank=rank, world=world, elem_lo=lo, elem_hi=hi, nodes_global=nodes_global,
                 elements_local=elements_local, shared_dofs=shared_dofs.astype(np.int32),
                 shared_slot=shared_slot.astype(np.int32), ghost_mask=ghost,
                 n_iface=int(len(shared_nodes) * dim), dim=dim,
                 iface_elems=np.flatnonzero(touches).astype(np.int32))


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
    Buffers: buf1 = [3 floats: 0, 0, sum u_free^2 | partial f_int of the shared dofs (n_iface)],
             buf2 = [partial grad_u of the shared dofs (n_iface) | grad_theta (n_theta_active) | sum r^2, sum d^2, 0].
    Phases of one iteration (stream-ordered; see the module docstring):"""
    device: torch.device
    n_iface: int
    n_theta_active: int

    def iface_forward(self, iface: torch.Tensor): ...     # A: interface elements -> iface[slot] (else 0)
    def forward(self): ...                                # B: nets on all elements
    def backward(self, iface: torch.Tensor, buf2: torch.Tensor): ...   # C: residual (reduced f_int), backward, pack buf2
    def update_interior(self): ...                        # D: grad_u + Adam(u) of the dofs that are not shared
    def update_shared(self, buf2: torch.Tensor, sums3: torch.Tensor): ...  # E: Adam(theta), shared dofs; sums3[2]
    def finalize(self, r2d2: torch.Tensor, u2: torch.Tensor): ...


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


def destroy_rccl_comms():
    """Release the C driver's communicators (call before torch.distributed.destroy_process_group)."""
    lib = _capi.load()
    for comm in list(_RCCL_COMMS.values()):
        if comm is not None:
            torch.cuda.synchronize()
            lib.pf_comm_destroy(comm)
    _RCCL_COMMS.clear()


def run_iterations(backend: ShardBackend, n_iter: int, group=None,
                   bufs: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """n_iter GD iterations with the two collectives per iteration of the module docstring (+ one
    flush at the end, so the device state is final when this returns).  On an RCCL process group the
    HIP backend's iterations are issued by the C driver (pf_shard_iterations: kernels and ncclAllReduce
    from one C loop, no Python between them); everything else goes through the Python schedule below."""
    if bufs is None:
        bufs = getattr(backend, "bufs", None) or make_buffers(backend)
    eng = getattr(backend, "eng", None)
    comm = rccl_comm(backend, group) if n_iter > 0 else None
    if n_iter > 0:
        backend.driver_used = "c-rccl" if (comm is not None and backend.early_iface) else "python"
    if comm is not None and backend.early_iface:
        g0, g1 = backend.graphs()
        with eng.on_stream():
            _capi.check(eng.lib.pf_shard_iterations(eng._ref(), comm, int(n_iter), bufs[0].data_ptr(),
                                                    bufs[1].data_ptr(), g0, g1, eng._stream()),
                        "pf_shard_iterations")
        return bufs
    ctx = eng.on_stream() if eng is not None else contextlib.nullcontext()
    with ctx:   # kernels and collectives on one stream (the engine's)
        _run_iterations(backend, n_iter, group, bufs[0], bufs[1])
    return bufs


def _run_iterations(backend, n_iter, group, buf1, buf2):
    ni, nt = backend.n_iface, backend.n_theta_active
    r2d2 = buf2[ni + nt:ni + nt + 2]
    pending = False
    for _ in range(n_iter):
        backend.iface_forward(buf1[3:])
        _all_reduce(buf1, group)                 # (1) + the previous iteration's sum u^2 in buf1[2]
        backend.forward()
        if pending:
            backend.finalize(r2d2, buf1[2:3])    # bookkeeping of the previous iteration
        backend.backward(buf1[3:], buf2)
        backend.update_interior()
        _all_reduce(buf2, group)                 # (2)
        backend.update_shared(buf2, buf1[:3])
        pending = True
    if pending:
        _all_reduce(buf1[:3], group)             # flush: the last iteration's sum u^2
        backend.finalize(r2d2, buf1[2:3])


def make_buffers(backend: ShardBackend):
    dev = backend.device
    f32 = dict(dtype=torch.float32, device=dev)
    return (torch.zeros(3 + backend.n_iface, **f32),
            torch.zeros(backend.n_iface + backend.n_theta_active + 3, **f32))


# ---------------------------------------------------------------------------------------------------
# HIP backend
# ---------------------------------------------------------------------------------------------------
class HipShardBackend(ShardBackend):
    """One rank's shard on its GPU: a HipEngine over the local sub-mesh plus the interface maps."""

    def __init__(self, local_model, host_plan: HostPlan, shard: Shard, has_measurements: bool,
                 device=None, wg_mode=None, fe_mode=None):
        from .engine import HipEngine
        self.shard = shard
        ie = shard.iface_elems if shard.iface_elems is not None else np.zeros(0, dtype=np.int32)
        # phase A needs the interface elements in one block's LDS; a very long interface (a mesh cut
        # across thousands of elements) falls back to "full forward, then gather": correct, no overlap
        self.early_iface = len(ie) <= _capi.PF_MAX_IFACE_ELEMS
        self.eng = HipEngine(local_model, device=device, host_plan=host_plan, wg_mode=wg_mode,
                             fe_mode=fe_mode,
                             iface=(shard.shared_dofs, shard.shared_slot, shard.n_iface,
                                    ie if self.early_iface else None))
        self.eng.has_measurements = has_measurements
        self.device = self.eng.device
        self.n_iface = shard.n_iface
        self.n_theta_active = self.eng.n_theta_active
        self.fbuf = None if self.early_iface else torch.zeros(host_plan.n_dofs, dtype=torch.float32,
                                                               device=self.device)
        self.bufs = make_buffers(self)           # (buf1, buf2): fixed addresses, baked into the graphs
        self._graphs = None                      # C driver only

    # -- solve_gd-level control ---------------------------------------------------------------------
    def begin(self, u_initial_local, lam, config, want_history=True, group=None):
        self._drop_graphs()
        if self.eng.n_theta:
            # replicas of theta must be bit-identical on every rank (see broadcast_theta)
            with self.eng.on_stream():
                broadcast_theta(self.eng.theta.flat, group)
        self.eng.begin(u_initial_local, lam, config, want_history=want_history)
        # grad_theta is reduced straight into buf2, where the second collective reads it
        self.eng.P.grad_theta = self.bufs[1].data_ptr() + 4 * self.n_iface
        self.bufs[0].zero_()

    def _drop_graphs(self):
        if self._graphs:
            for g in self._graphs:
                if g:
                    self.eng.lib.pf_graph_destroy(g)
        self._graphs = None

    def graphs(self):
        """The two hipGraphs of pf_shard_graph_create (phases B..D without / with the previous
        iteration's bookkeeping) for the C driver.  OFF by default: one graph replay per iteration
        measured slower than the same kernels launched one by one (0.271 vs 0.226 ms per iteration at 10^6
        elements, world_size 1, MI355X) — the replay's fixed cost is not amortised over one iteration the
        way the single-GPU path amortises it over ten.  PINNFEM_SHARD_GRAPH=1 turns it on."""
        if os.environ.get("PINNFEM_SHARD_GRAPH", "0") != "1":
            return None, None
        if self._graphs is None:
            import ctypes as C
            e = self.eng
            buf1, buf2 = self.bufs
            out = []
            with e.on_stream():
                for with_fin in (0, 1):
                    g = C.c_void_p()
                    _capi.check(e.lib.pf_shard_graph_create(e._ref(), buf1.data_ptr(), buf2.data_ptr(), with_fin,
                                                            e._stream(), C.byref(g)), "pf_shard_graph_create")
                    out.append(g)
            self._graphs = out
        return self._graphs[0], self._graphs[1]

    def _phase(self, k, eager):
        return eager()

    def state(self):
        return self.eng.state()

    def history(self, n):
        return self.eng.history(n)

    # -- ShardBackend (Python driver: gloo tests, single-GPU rehearsal, fallback): one C call per phase ----
    def _check_bufs(self, *tensors):
        buf1, buf2 = self.bufs
        lo1, hi1 = buf1.data_ptr(), buf1.data_ptr() + 4 * buf1.numel()
        lo2, hi2 = buf2.data_ptr(), buf2.data_ptr() + 4 * buf2.numel()
        for t in tensors:
            p = t.data_ptr()
            if t.numel() and not (lo1 <= p < hi1 or lo2 <= p < hi2):
                raise ValueError("HipShardBackend works on its own collective buffers (backend.bufs)")

    def _iface_ptr(self, iface):
        # an empty view (world_size 1: no interface) has no address of its own
        return iface.data_ptr() if iface.numel() else self.bufs[0].data_ptr() + 12

    def iface_forward(self, iface):
        e = self.eng
        self._check_bufs(iface)
        if not self.early_iface:
            # fallback: full forward, partial f_int of every dof, pack the shared ones
            s = e._stream()
            _capi.check(e.lib.pf_shard_forward(e._ref(), s), "pf_shard_forward")
            _capi.check(e.lib.pf_internal_force(e._ref(), e.u.data_ptr(), self.fbuf.data_ptr(), s), "pf_internal_force")
            _capi.check(e.lib.pf_iface_pack(e._ref(), self.fbuf.data_ptr(), iface.data_ptr(), s), "pf_iface_pack")
            return
        self._phase(0, lambda: _capi.check(
            e.lib.pf_shard_iface_forward(e._ref(), self._iface_ptr(iface), e._stream()), "pf_shard_iface_forward"))

    def forward(self):
        e = self.eng
        if not self.early_iface:
            return
        self._phase(1, lambda: _capi.check(e.lib.pf_shard_forward(e._ref(), e._stream()), "pf_shard_forward"))

    def backward(self, iface, buf2):
        e = self.eng
        self._check_bufs(iface, buf2)
        self._phase(2, lambda: _capi.check(
            e.lib.pf_shard_backward(e._ref(), self._iface_ptr(iface), buf2.data_ptr(), e._stream()), "pf_shard_backward"))

    def update_interior(self):
        e = self.eng
        self._phase(3, lambda: _capi.check(
            e.lib.pf_shard_update_interior(e._ref(), e._stream()), "pf_shard_update_interior"))

    def update_shared(self, buf2, sums3):
        e = self.eng
        self._check_bufs(buf2, sums3)
        self._phase(4, lambda: _capi.check(
            e.lib.pf_shard_update_shared(e._ref(), buf2.data_ptr(), sums3.data_ptr(), e._stream()),
            "pf_shard_update_shared"))

    def finalize(self, r2d2, u2):
        e = self.eng
        _capi.check(e.lib.pf_finalize_from(e._ref(), r2d2.data_ptr(), u2.data_ptr(), e._stream()),
                    "pf_finalize_from")


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
    """Local view of the fully assembled f_int (re-evaluating the nets): partial gather + interface sum."""
    e = backend.eng
    with e.on_stream():
        f = e.internal_force(lam=lam)
        iface = torch.zeros(max(backend.n_iface, 1), dtype=torch.float32, device=backend.device)
        _capi.check(e.lib.pf_iface_pack(e._ref(), f.data_ptr(), iface.data_ptr(), e._stream()), "pf_iface_pack")
        _all_reduce(iface, group)
        _capi.check(e.lib.pf_iface_unpack(e._ref(), iface.data_ptr(), f.data_ptr(), e._stream()), "pf_iface_unpack")
    return f


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
class ShardedChainEngine:
    """Rank r owns elements [r*n, (r+1)*n) of a world*n-element collinear truss (SURVEY.md §8d
    inputs).  Same kernels and collectives as the general path; no global arrays are built."""

    def __init__(self, n_local: int, workload: str, rank: int, world: int, device):
        from .fem.model import FEMModel, Material
        from .fem.properties import NNProperty
        from .nets import SimpleNN
        n = int(n_local)
        n_total = n * world
        e0 = rank * n
        x = (np.arange(n + 1, dtype=np.float64) + e0)
        nodes_l = np.stack([x, np.zeros(n + 1)], axis=1)
        el = np.stack([np.arange(n), np.arange(1, n + 1)], axis=1)
        loads = np.zeros(2 * (n + 1))
        if rank == world - 1:
            loads[2 * n] = 1.0
        fixed = list(2 * np.arange(n + 1) + 1)
        if rank == 0:
            fixed = [0] + fixed
        fixed = np.array(sorted(fixed))
        # shared nodes: global node r*n for r = 1..world-1  (local 0 of rank r, local n of rank r-1)
        sd, ss = [], []
        ghost = np.zeros(2 * (n + 1), dtype=bool)
        if rank > 0:
            sd += [0, 1]
            ss += [2 * (rank - 1), 2 * (rank - 1) + 1]
            ghost[0:2] = True                      # owned by rank-1
        if rank < world - 1:
            sd += [2 * n, 2 * n + 1]
            ss += [2 * rank, 2 * rank + 1]
        ie = ([0] if rank > 0 else []) + ([n - 1] if rank < world - 1 else [])   # elements at the shared nodes
        shard = Shard(rank=rank, world=world, elem_lo=e0, elem_hi=e0 + n,
                      nodes_global=np.arange(e0, e0 + n + 1), elements_local=el,
                      shared_dofs=np.array(sd, dtype=np.int32), shared_slot=np.array(ss, dtype=np.int32),
                      ghost_mask=ghost, n_iface=2 * (world - 1), dim=2,
                      iface_elems=np.array(sorted(set(ie)), dtype=np.int32))
        # measurements ux_i = x_i, uy_i = 0 at every global node >= 1 (owner adds the shared ones)
        k = np.arange(n + 1)
        keep = (k + e0 >= 1) & ~ghost[0::2]
        kk = k[keep]
        md = np.stack([2 * kk, 2 * kk + 1], axis=1).reshape(-1)
        mv = np.stack([x[kk], np.zeros(kk.size)], axis=1).reshape(-1)
        hp = build_host_plan(nodes_l, el, loads, fixed, 2, mv, md)
        hp.dof_flags[shard.shared_dofs] |= _capi.PF_DOF_SHARED
        hp.dof_flags[ghost] |= _capi.PF_DOF_GHOST
        hp.n_meas = 2 * n_total
        torch.manual_seed(0)                      # identical theta on every rank
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
        self.backend.driver_used = "c-rccl" if (comm is not None and self.backend.early_iface) else "python"

    def iterate(self, n):
        run_iterations(self.backend, n, bufs=self.bufs)

    def iterate_timed(self, n):
        """Per-kernel HIP-event times of the rank-local kernels (collectives excluded): runs the
        single-GPU iteration sequence on this shard's engine; shared dofs see partial sums, which is
        irrelevant for timing.  bench.py uses it only for the roofline entry."""
        return self.backend.eng.iterate_timed(n)

    def state(self):
        return self.backend.state()
