"""HipEngine — host-side owner of the device buffers of one FEM model and driver of the HIP kernels.

PyTorch is used for device memory and streams only; all arithmetic of the path runs in
libpinnfem_hip.so through the C ABI (include/pinnfem_hip.h).  There is no CPU fallback: without a
GPU or without the built library construction raises.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import functools
import os
from typing import List, Optional

import numpy as np
import torch

from . import _capi
from ._capi import PfProblem, PfState, PinnFemHipError
from .nets import FlatTheta, NetSpec, describe_module
from .plan import HostPlan, build_host_plan


def _require_gpu(device=None) -> torch.device:
    if not torch.cuda.is_available():
        raise PinnFemHipError(
            "no ROCm GPU visible: pinn_fem_amd runs its hot path only through the HIP kernels "
            "(gfx950) and has no CPU fallback")
    return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())


def _on_engine_stream(fn):
    """Run a HipEngine method on the engine's own HIP stream (hipGraph capture is not allowed on the
    default stream), ordered after the caller's current stream on entry and before it on exit."""
    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        with self.on_stream():
            return fn(self, *a, **k)
    return wrapper


class HipEngine:
    """Device-resident problem: mesh plan, flat theta, Adam state, workspaces."""

    def __init__(self, model, measured_disp=None, measured_dofs=None, device=None,
                 wg_mode: Optional[int] = None, n_part_blocks: Optional[int] = None,
                 host_plan: Optional[HostPlan] = None, fe_mode: Optional[int] = None,
                 iface=None, mlp_dtype: Optional[str] = None):
        self.lib = _capi.load()
        self.device = _require_gpu(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.model = model
        hp = host_plan or build_host_plan(model.nodes, model.elements, model.loads, model.fixed_dofs,
                                          model.dimension, measured_disp, measured_dofs)
        self.plan = hp
        dev = self.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.conn, self.egeo, self.ecent = t(hp.conn), t(hp.egeo), t(hp.ecent)
        self.adj_ptr, self.adj, self.adj_other = t(hp.adj_ptr), t(hp.adj), t(hp.adj_other)
        self.f_ext, self.dof_flags, self.meas_val = t(hp.f_ext), t(hp.dof_flags), t(hp.meas_val)
        self.has_measurements = measured_disp is not None and measured_dofs is not None
        # multi-GPU shard interface: (interface dofs int32[], their slots int32[], n_iface, (own_lo, own_hi))
        self.own_range = (0, 0)
        if iface is not None:
            self.shared_dofs, self.shared_slot = t(np.asarray(iface[0], dtype=np.int32)), t(np.asarray(iface[1], dtype=np.int32))
            self.n_shared, self.n_iface = int(len(iface[0])), int(iface[2])
            if len(iface) > 3 and iface[3] is not None:
                self.own_range = (int(iface[3][0]), int(iface[3][1]))
        else:
            self.shared_dofs = self.shared_slot = None
            self.n_shared = self.n_iface = 0

        # ---- nets: young, area evaluated; density's parameters only ride along in theta ----------
        mat = model.material
        self.specs: List[NetSpec] = []
        param_lists = []
        for prop in (mat.young, mat.area):
            if prop.is_trainable():
                spec = describe_module(prop.net)
                spec.positive = bool(prop.enforce_positive)
                spec.scale = float(prop.scale)
                if spec.in_dim != model.dimension + 1:
                    # the reference feeds [load_factor, x(, y)] whatever input_dim says
                    # (properties.py:116-125) and torch then raises a shape error
                    raise RuntimeError(
                        f"mat1 and mat2 shapes cannot be multiplied: NN input has {model.dimension + 1} "
                        f"columns [load_factor, x(, y)] but the net expects {spec.in_dim}")
                param_lists.append(prop.get_torch_params())
            else:
                spec = NetSpec(enabled=False, scale=float(prop.value()))
            self.specs.append(spec)
        n_active = sum(p.numel() for lst in param_lists for p in lst)
        if mat.density.is_trainable():
            param_lists.append(mat.density.get_torch_params())
        self.theta = FlatTheta(param_lists, dev)
        self.n_theta = self.theta.n
        self.n_theta_active = n_active
        self.tensor_off = torch.tensor(self.theta.tensor_off, dtype=torch.int32, device=dev)

        lib = self.lib
        pad_off, theta_off, pad_index = 0, 0, []
        self._net_offsets = []
        for spec in self.specs:
            if not spec.enabled:
                self._net_offsets.append((0, 0))
                continue
            cnt = lib.pf_net_pad_count(spec.in_dim, spec.width, spec.n_hidden)
            _capi.check(min(cnt, 0), "pf_net_pad_count")
            self._net_offsets.append((theta_off, pad_off))
            for q in range(spec.n_params):
                pad_index.append(pad_off + lib.pf_net_pad_index(spec.in_dim, spec.width, spec.n_hidden, q))
            pad_off += cnt
            theta_off += spec.n_params
        self.pad_total = pad_off
        self.pad_index = torch.tensor(pad_index if pad_index else [0], dtype=torch.int32, device=dev)

        if wg_mode is None:
            wg_mode = int(os.environ.get("PINNFEM_WG_MODE", _capi.PF_WG_MFMA32))
        if wg_mode == _capi.PF_WG_MFMA32 and any(sp.enabled and sp.width > _capi.PF_N32_WIDTH_MAX for sp in self.specs):
            wg_mode = _capi.PF_WG_MFMA44          # widths 31, 32: the exact-f32 4x4x1 engine (another HIP engine)
        self.wg_mode = wg_mode
        # precision of the MLP matrix products: "f32" (split-f16 operands, float32-grade) or "bf16" (plain bf16 operands);
        # the reduced-precision variant exists only in the MFMA32 engine
        if mlp_dtype is None:
            mlp_dtype = getattr(model, "_pf_mlp_dtype", None) or os.environ.get("PINNFEM_MLP_DTYPE", "f32")
        if mlp_dtype not in ("f32", "bf16"):
            raise ValueError(f"mlp_dtype must be 'f32' or 'bf16', got {mlp_dtype!r}")
        if mlp_dtype == "bf16" and wg_mode != _capi.PF_WG_MFMA32:
            raise NotImplementedError("mlp_dtype='bf16' needs the MFMA32 engine (net widths <= 30)")
        self.mlp_dtype = mlp_dtype
        # MFMA32 engine: operand images of the enabled nets; scale of the coordinates in the f16 gradient products
        op_off, self._op_off = 0, [0, 0]
        if wg_mode == _capi.PF_WG_MFMA32:
            for k, spec in enumerate(self.specs):
                if spec.enabled:
                    cnt = lib.pf_net_op_count(spec.in_dim, spec.width, spec.n_hidden)
                    _capi.check(min(cnt, 0), "pf_net_op_count")
                    self._op_off[k] = op_off
                    op_off += (cnt + 63) // 64 * 64
        self.net_op = torch.zeros(max(op_off, 1), dtype=torch.float32, device=dev)
        cmax = float(np.max(np.abs(hp.ecent))) if hp.n_elems else 1.0
        self.coord_exp = 14 - int(np.ceil(np.log2(max(cmax, 1e-30)))) if cmax > 0 else 0
        self.coord_exp = int(min(max(self.coord_exp, -100), 100))
        if fe_mode is None:
            fe_mode = int(os.environ.get("PINNFEM_FE_MODE", _capi.PF_FE_REFERENCE))
        self.fe_mode = fe_mode
        if n_part_blocks is None:
            n_part_blocks = int(os.environ.get("PINNFEM_PART_BLOCKS", 1024))
        self.n_part_blocks = max(1, min(int(n_part_blocks), _capi.PF_MAX_BLOCKS))

        f32 = dict(dtype=torch.float32, device=dev)
        nd, ne = hp.n_dofs, max(hp.n_elems, 1)
        self.u = torch.zeros(nd, **f32)
        self.u_alt = torch.zeros(nd, **f32)      # the iteration graph's second displacement vector (pf_problem.u_alt)
        self.m_u = torch.zeros(nd, **f32)
        self.v_u = torch.zeros(nd, **f32)
        self.m_t = torch.zeros(max(self.n_theta, 1), **f32)
        self.v_t = torch.zeros(max(self.n_theta, 1), **f32)
        self.theta_pad = torch.zeros(max(self.pad_total, 1), **f32)
        # second half of (theta, m_t, v_t): lets the iteration graph fold the parameter update into the next forward launch
        self.theta_alt = torch.zeros(3 * max(self.n_theta, 1), **f32)
        # two halves: the iteration graph ping-pongs between them (pf_problem.prop_double); everything else
        # uses the first
        self.prop_e = torch.zeros(2 * ne, **f32)
        self.prop_a = torch.zeros(2 * ne, **f32)
        # entries of ke = s*pattern per element (s*c2, s*cs, s*s2; 1-D: s), written by the MFMA32 forward pass for the node
        # kernels (two halves, like the properties)
        self.elem_k = (torch.zeros(2 * ne * (3 if hp.dim == 2 else 1), **f32)
                       if self.wg_mode == _capi.PF_WG_MFMA32 and any(sp.enabled for sp in self.specs) else None)
        self.g_f = torch.zeros(nd, **f32)
        self.g_ea = torch.zeros(ne, **f32)
        self.grad_u = torch.zeros(nd, **f32)
        self.grad_theta = torch.zeros(max(self.n_theta, 1), **f32)
        size_probe = PfProblem()                      # the library owns the workspace layout: ask it for the size
        size_probe.n_part_blocks, size_probe.pad_total = self.n_part_blocks, max(self.pad_total, 1)
        self.partials = torch.zeros(int(self.lib.pf_partials_count(C.byref(size_probe))), **f32)
        self.state_t = torch.zeros(C.sizeof(PfState) // 4, dtype=torch.int32, device=dev)
        self.hist = torch.zeros(1, **f32)
        self.hist_rows = 0
        self.P = PfProblem()
        self._graph = None
        self._configured = False
        env_k = os.environ.get("PINNFEM_GRAPH_ITERS")
        self.GRAPH_ITERS = int(env_k) if env_k else (self.GRAPH_ITERS_LARGE if hp.n_elems >= 200_000 else self.GRAPH_ITERS)
        self.configure(lam=1.0)

    # ------------------------------------------------------------------------------------------
    def _stream(self) -> int:
        return self.stream.cuda_stream

    @contextlib.contextmanager
    def on_stream(self):
        outer = torch.cuda.current_stream(self.device)
        if outer == self.stream:
            yield
            return
        self.stream.wait_stream(outer)
        with torch.cuda.stream(self.stream):
            yield
        outer.wait_stream(self.stream)

    def configure(self, lam: float, alpha_physics: float = 1.0, alpha_data: float = 100.0,
                  lr_u: float = 1e-7, lr_t: float = 1e-4, tol: float = 1e-6, max_iter: int = 0,
                  want_history: bool = True, want_grad_u: bool = False):
        """Fill the pf_problem record (scalars of SolverConfig + pointers)."""
        self._drop_graph()          # a captured graph holds the old record by value
        hp, P = self.plan, self.P
        M = P.mesh
        M.dim, M.n_nodes, M.n_elems, M.n_dofs = hp.dim, hp.n_nodes, hp.n_elems, hp.n_dofs
        M.conn, M.egeo, M.ecent = self.conn.data_ptr(), self.egeo.data_ptr(), self.ecent.data_ptr()
        M.adj_ptr, M.adj = self.adj_ptr.data_ptr(), self.adj.data_ptr()
        M.f_ext, M.dof_flags, M.meas_val = (self.f_ext.data_ptr(), self.dof_flags.data_ptr(),
                                            self.meas_val.data_ptr())
        M.n_meas = hp.n_meas
        for k, spec in enumerate(self.specs):
            n = P.net[k]
            n.enabled = int(spec.enabled)
            n.in_dim, n.width, n.n_hidden = spec.in_dim, spec.width, spec.n_hidden
            n.positive, n.scale = int(spec.positive), float(spec.scale)
            n.theta_off, n.pad_off = self._net_offsets[k]
        P.u, P.m_u, P.v_u = self.u.data_ptr(), self.m_u.data_ptr(), self.v_u.data_ptr()
        P.theta, P.m_t, P.v_t = self.theta.flat.data_ptr(), self.m_t.data_ptr(), self.v_t.data_ptr()
        P.n_theta, P.n_theta_active = self.n_theta, self.n_theta_active
        P.tensor_off, P.n_tensors = self.tensor_off.data_ptr(), len(self.theta.tensor_off) - 1
        P.wg_mode = self.wg_mode
        P.lam, P.alpha_physics, P.alpha_data = float(lam), float(alpha_physics), float(alpha_data)
        P.lr_u, P.lr_t, P.tol = float(lr_u), float(lr_t), float(tol)
        P.beta1, P.beta2, P.eps = 0.9, 0.999, 1e-8          # torch.optim.Adam defaults (solver.py:234)
        P.use_data = int(self.has_measurements and alpha_data > 0)   # solver.py:273
        P.max_iter = int(max_iter)
        P.theta_pad, P.prop_e, P.prop_a = (self.theta_pad.data_ptr(), self.prop_e.data_ptr(),
                                           self.prop_a.data_ptr())
        P.g_f, P.g_ea = self.g_f.data_ptr(), self.g_ea.data_ptr()
        P.grad_u = self.grad_u.data_ptr() if want_grad_u else None
        P.grad_theta, P.partials = self.grad_theta.data_ptr(), self.partials.data_ptr()
        if want_history and max_iter > 0:
            if self.hist_rows < max_iter:
                self.hist = torch.zeros(max_iter * _capi.PF_HIST_COLS, dtype=torch.float32,
                                        device=self.device)
                self.hist_rows = max_iter
            P.hist = self.hist.data_ptr()
        else:
            P.hist = None
        P.state = self.state_t.data_ptr()
        P.n_part_blocks, P.pad_total = self.n_part_blocks, self.pad_total
        P.pad_index = self.pad_index.data_ptr()
        P.n_meas_f = float(hp.n_meas)
        P.fe_mode = int(self.fe_mode)
        P.shared_dofs = self.shared_dofs.data_ptr() if self.n_shared else None
        P.shared_slot = self.shared_slot.data_ptr() if self.n_shared else None
        P.n_shared, P.n_iface = self.n_shared, self.n_iface
        P.own_lo, P.own_hi = self.own_range
        P.prop_double = 1
        P.net_op = self.net_op.data_ptr() if self.wg_mode == _capi.PF_WG_MFMA32 else None
        P.op_off[0], P.op_off[1] = self._op_off
        P.coord_exp = self.coord_exp
        P.mlp_dtype = _capi.PF_MLP_BF16 if self.mlp_dtype == "bf16" else _capi.PF_MLP_F32
        P.elem_k = self.elem_k.data_ptr() if self.elem_k is not None else None
        P.theta_alt = self.theta_alt.data_ptr() if self.n_theta_active > 0 else None
        P.u_alt = self.u_alt.data_ptr()
        P.adj_other = self.adj_other.data_ptr()
        self._configured = True

    def _ref(self):
        return C.byref(self.P)

    def fusion_info(self) -> int:
        """Bit mask of the fused launches of this problem (_capi.PF_FUSED_*)."""
        return int(self.lib.pf_fusion_info(self._ref()))

    # ---- solve_gd support ------------------------------------------------------------------------
    @_on_engine_stream
    def begin(self, u_initial, lam, config, max_iter: Optional[int] = None, want_history=True):
        """Start one solve_gd call: fresh Adam state (solver.py:234-238), u = warm start or 0."""
        if self.n_theta and not self.theta.still_bound():
            raise PinnFemHipError("a network parameter was re-assigned outside the engine; "
                                  "rebuild the engine for this model")
        self.configure(lam=lam, alpha_physics=config.alpha_physics, alpha_data=config.alpha_data,
                       lr_u=config.learning_rate_u, lr_t=config.learning_rate_theta,
                       tol=config.tolerance,
                       max_iter=config.max_iterations if max_iter is None else max_iter,
                       want_history=want_history)
        if u_initial is None:
            self.u.zero_()
        else:
            src = u_initial.detach() if isinstance(u_initial, torch.Tensor) else torch.as_tensor(
                np.asarray(u_initial))
            self.u.copy_(src.reshape(-1).to(device=self.device, dtype=torch.float32))
        s = self._stream()
        self._pending_tail = False          # (a new solve: nothing of the previous one is pending)
        _capi.check(self.lib.pf_reset(self._ref(), s), "pf_reset")
        _capi.check(self.lib.pf_pack_theta(self._ref(), s), "pf_pack_theta")

    # iterations per captured hipGraph.  A PLAIN replay ends with ~45 us of stand-alone kernels (parameter update, displacement
    # update, finalize) that the iterations inside it do not pay, so large meshes replay 20 at a time (the host polls the stop
    # flag every 50 anyway); small meshes keep 10 (a graph is captured per solve_gd call: capture time counts there).
    # Chained replays (iterate(defer_tail=True)) pay that tail once per solve.
    # PINNFEM_GRAPH_ITERS overrides both (even numbers: the graph ping-pongs state between two halves).
    GRAPH_ITERS = 10
    GRAPH_ITERS_LARGE = 20

    def _drop_graph(self):
        g = getattr(self, "_graph", None)
        if g:
            self.lib.pf_graph_destroy(g)
        self._graph = None
        for h in getattr(self, "_chain_graphs", {}).values():
            if h:
                self.lib.pf_graph_destroy(h)
        self._chain_graphs = {}

    graph_creates = 0        # hipGraph captures + instantiations so far (bench.py asserts none is timed)
    _pending_tail = False    # a chained replay has left its last iteration's updates / bookkeeping pending (see iterate)

    @_on_engine_stream
    def prepare_graph(self, chained: bool = False):
        """Capture and instantiate the iteration hipGraph of the current pf_problem record now (it is
        otherwise created by the first iterate() call that replays it).  Enqueues no iteration.
        chained: also the two graphs of chained replays (iterate(defer_tail=True)); returns whether the problem has them."""
        if getattr(self, "_graph", None) is None:
            g = C.c_void_p()
            _capi.check(self.lib.pf_graph_create(self._ref(), self.GRAPH_ITERS, self._stream(), C.byref(g)),
                        "pf_graph_create")
            self._graph = g
            self.graph_creates += 1
        if not chained:
            return True
        if not hasattr(self, "_chain_graphs"):
            self._chain_graphs = {}
        for flags in (_capi.PF_GRAPH_NO_TAIL, _capi.PF_GRAPH_NO_TAIL | _capi.PF_GRAPH_CONT_HEAD):
            if flags not in self._chain_graphs:
                g = C.c_void_p()
                rc = self.lib.pf_graph_create_ex(self._ref(), self.GRAPH_ITERS, flags, self._stream(), C.byref(g))
                if rc == _capi.PF_ERR_UNSUPPORTED:
                    self._chain_graphs[flags] = None        # (not the one-chain form: plain replays)
                    continue
                _capi.check(rc, "pf_graph_create_ex")
                self._chain_graphs[flags] = g
                self.graph_creates += 1
        return all(self._chain_graphs.get(f) for f in (2, 3))

    @_on_engine_stream
    def flush(self):
        """Run the pending tail of chained replays (parameter update, displacement update and bookkeeping of the last
        iteration).  iterate() calls it before anything but another chained replay; call it before reading the state."""
        if self._pending_tail:
            _capi.check(self.lib.pf_graph_tail(self._ref(), self.GRAPH_ITERS, self._stream()), "pf_graph_tail")
            self._pending_tail = False

    @_on_engine_stream
    def iterate(self, n_iter: int, use_graph: Optional[bool] = None, defer_tail: bool = False):
        """Enqueue n_iter GD iterations.  Whole multiples of GRAPH_ITERS replay a captured hipGraph
        (created lazily per begin(), or ahead of time by prepare_graph(); the graph bakes in the current
        pf_problem record), the remainder is launched eagerly.  Launches after the device-side stop are
        no-ops either way.
        defer_tail: the replays are CHAINED where the problem allows it — a replay ends behind its last gradient-row
        reduction and the next replay's first iteration carries that iteration's updates and bookkeeping, like every other
        iteration of a replay carries its predecessor's; the ~45 us of stand-alone launches a plain replay ends with are
        paid once, by flush().  state() / history() / u / theta lag by that one iteration until flush()."""
        n_iter = int(n_iter)
        if use_graph is None:
            use_graph = os.environ.get("PINNFEM_GRAPH", "1") != "0"
        s = self._stream()
        k = self.GRAPH_ITERS
        if use_graph and n_iter >= k:
            chained = defer_tail and self.prepare_graph(chained=True)
            if not chained:
                self.flush()
                self.prepare_graph()
            while n_iter >= k:
                if chained:
                    g = self._chain_graphs[3 if self._pending_tail else 2]
                    _capi.check(self.lib.pf_graph_launch(g, s), "pf_graph_launch")
                    self._pending_tail = True
                else:
                    _capi.check(self.lib.pf_graph_launch(self._graph, s), "pf_graph_launch")
                n_iter -= k
        if n_iter > 0:
            self.flush()
            _capi.check(self.lib.pf_gd_iterations(self._ref(), n_iter, s), "pf_gd_iterations")

    def __del__(self):
        try:
            self._drop_graph()
        except Exception:
            pass

    @_on_engine_stream
    def iterate_timed(self, n_iter: int) -> np.ndarray:
        """Like iterate() but with HIP events around every kernel; synchronises.  Returns the average
        milliseconds of each kernel slot (_capi.KERNEL_SLOT_NAMES)."""
        out = (C.c_float * _capi.PF_KERNEL_SLOTS)()
        _capi.check(self.lib.pf_gd_iterations_timed(self._ref(), int(n_iter), self._stream(), out),
                    "pf_gd_iterations_timed")
        return np.array(list(out), dtype=np.float64)

    @_on_engine_stream
    def time_step_launch(self, what: str, reps: int = 20) -> float:
        """Average device time (ms) of `reps` back-to-back launches of one step of the current iteration state — "backward":
        the backward pass of every enabled net incl. the element adjoint, "forward": the forward pass of every enabled net —
        between two HIP events on the engine's stream (no launch gaps inside: the launches queue up behind each other).
        Idempotent steps: they read the state and rewrite workspaces only."""
        fn = {"backward": self.lib.pf_net_backward_all, "forward": self.lib.pf_net_forward_all}[what]
        s = self._stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            _capi.check(fn(self._ref(), s), what)
        e0.record(self.stream)
        for _ in range(int(reps)):
            _capi.check(fn(self._ref(), s), what)
        e1.record(self.stream)
        e1.synchronize()
        return e0.elapsed_time(e1) / float(reps)

    @_on_engine_stream
    def state(self) -> PfState:
        raw = self.state_t.cpu().numpy().tobytes()
        return PfState.from_buffer_copy(raw)

    @_on_engine_stream
    def history(self, n_rows: int) -> np.ndarray:
        if n_rows <= 0 or self.P.hist is None:
            return np.zeros((0, _capi.PF_HIST_COLS), dtype=np.float32)
        return self.hist[: n_rows * _capi.PF_HIST_COLS].cpu().numpy().reshape(n_rows, _capi.PF_HIST_COLS)

    # ---- building blocks -------------------------------------------------------------------------
    @_on_engine_stream
    def eval_properties(self, lam: Optional[float] = None):
        """young/area per element with the current theta (pf_pack_theta + pf_net_forward)."""
        if lam is not None:
            self.P.lam = float(lam)
        s = self._stream()
        self._clear_done()
        _capi.check(self.lib.pf_pack_theta(self._ref(), s), "pf_pack_theta")
        for k, spec in enumerate(self.specs):
            if spec.enabled:
                _capi.check(self.lib.pf_net_forward(self._ref(), k, s), "pf_net_forward")

    def _clear_done(self):
        # building-block calls outside a solve_gd run must not be masked by a finished run
        self.state_t[1] = 0

    @_on_engine_stream
    def internal_force(self, u: Optional[torch.Tensor] = None, lam: Optional[float] = None) -> torch.Tensor:
        """f_int = K(theta) u, re-evaluating the nets (the reference's no-grad re-assembly,
        solver.py:374-377)."""
        self.eval_properties(lam)
        uu = self.u if u is None else u.to(device=self.device, dtype=torch.float32).contiguous()
        out = torch.empty(self.plan.n_dofs, dtype=torch.float32, device=self.device)
        _capi.check(self.lib.pf_internal_force(self._ref(), uu.data_ptr(), out.data_ptr(), self._stream()),
                    "pf_internal_force")
        return out

    @_on_engine_stream
    def loss_and_grads(self, u: torch.Tensor, lam: float, alpha_physics=1.0, alpha_data=100.0):
        """One forward+backward without optimiser step.  Returns (dict of loss terms, grad_u,
        grad_theta) as device tensors (views of engine workspaces)."""
        self.configure(lam=lam, alpha_physics=alpha_physics, alpha_data=alpha_data, want_history=False,
                       want_grad_u=True)
        self.u.copy_(u.reshape(-1).to(device=self.device, dtype=torch.float32))
        s = self._stream()
        self._clear_done()
        _capi.check(self.lib.pf_pack_theta(self._ref(), s), "pf_pack_theta")
        _capi.check(self.lib.pf_loss_and_grads(self._ref(), s), "pf_loss_and_grads")
        st = self.state()
        losses = dict(loss_total=st.loss_total, loss_physics=st.loss_physics, loss_data=st.loss_data,
                      residual_norm=st.residual_norm)
        return losses, self.grad_u, self.grad_theta[: max(self.n_theta_active, 0)]

    @_on_engine_stream
    def vjp(self, u: torch.Tensor, g_f: torch.Tensor, lam: float):
        """(K^T g_f, d(g_f . f_int)/dtheta): the backward of f_int = K(theta) u for an arbitrary
        upstream gradient (autograd.Function backward)."""
        self.configure(lam=lam, alpha_data=0.0, want_history=False, want_grad_u=True)
        self.P.use_data = 0
        self.u.copy_(u.reshape(-1).to(device=self.device, dtype=torch.float32))
        self.g_f.copy_(g_f.reshape(-1).to(device=self.device, dtype=torch.float32))
        s, lib, ref = self._stream(), self.lib, self._ref()
        self._clear_done()
        _capi.check(lib.pf_pack_theta(ref, s), "pf_pack_theta")
        any_net = False
        for k, spec in enumerate(self.specs):
            if spec.enabled:
                any_net = True
                _capi.check(lib.pf_net_forward(ref, k, s), "pf_net_forward")
        if any_net:
            _capi.check(lib.pf_elem_adjoint(ref, s), "pf_elem_adjoint")
            for k, spec in enumerate(self.specs):
                if spec.enabled:
                    _capi.check(lib.pf_net_backward(ref, k, s), "pf_net_backward")
        _capi.check(lib.pf_node_gradu(ref, 0, s), "pf_node_gradu")
        if any_net:
            _capi.check(lib.pf_theta_reduce(ref, 0, s), "pf_theta_reduce")
        return self.grad_u, self.grad_theta[: max(self.n_theta_active, 0)]

    # ---- classical Newton-Raphson support: float64 matrix-free K v and Jacobi-PCG (pf_pcg.hip) ---------
    @_on_engine_stream
    def kv_f64(self, v: torch.Tensor, zero_fixed: bool = False) -> torch.Tensor:
        """K(E, A) v in float64 (fem/assembly.py:16-75 without forming K).  NN properties, if any, must
        have been evaluated (eval_properties)."""
        vv = v.to(device=self.device, dtype=torch.float64).contiguous()
        out = torch.empty(self.plan.n_dofs, dtype=torch.float64, device=self.device)
        _capi.check(self.lib.pf_kv_f64(self._ref(), vv.data_ptr(), out.data_ptr(), int(zero_fixed), self._stream()),
                    "pf_kv_f64")
        return out

    @_on_engine_stream
    def pcg_solve(self, b: torch.Tensor, rtol: float = 1e-13, max_iter: Optional[int] = None, poll: int = 64):
        """K_ff x = b by conjugate gradients with the diag(K_ff) preconditioner, float64, on the device.
        Returns (x with zeros on fixed dofs, iterations, converged)."""
        n = self.plan.n_dofs
        bb = b.to(device=self.device, dtype=torch.float64).contiguous()
        x = torch.zeros(n, dtype=torch.float64, device=self.device)
        ws = torch.zeros(int(self.lib.pf_pcg_workspace_count(self._ref())), dtype=torch.float64, device=self.device)
        s = self._stream()
        _capi.check(self.lib.pf_pcg_begin(self._ref(), bb.data_ptr(), x.data_ptr(), ws.data_ptr(), float(rtol), s),
                    "pf_pcg_begin")
        if max_iter is None:
            max_iter = 40 * n + 2000            # slender trusses are beam-like: CG needs far more than n steps
        st = (C.c_double * 4)()
        done_it = 0
        graph = C.c_void_p()
        use_graph = os.environ.get("PINNFEM_GRAPH", "1") != "0" and max_iter >= poll
        if use_graph:
            _capi.check(self.lib.pf_pcg_graph_create(self._ref(), x.data_ptr(), ws.data_ptr(), int(poll), s,
                                                     C.byref(graph)), "pf_pcg_graph_create")
        try:
            while True:
                k = min(poll, max_iter - done_it)
                if use_graph and k == poll:
                    _capi.check(self.lib.pf_graph_launch(graph, s), "pf_graph_launch")
                    _capi.check(self.lib.pf_pcg_state(self._ref(), ws.data_ptr(), st, s), "pf_pcg_state")
                else:
                    _capi.check(self.lib.pf_pcg_iterations(self._ref(), x.data_ptr(), ws.data_ptr(), int(max(k, 0)), st, s),
                                "pf_pcg_iterations")
                done_it += max(k, 0)
                if st[1] != 0.0 or done_it >= max_iter:
                    break
        finally:
            if graph:
                self.lib.pf_graph_destroy(graph)
        converged = st[2] <= (rtol * rtol) * st[3] * 4.0 or st[3] == 0.0     # |r| <= 2 rtol |b|
        return x, int(st[0]), bool(converged), float(st[2]), float(st[3])

    @_on_engine_stream
    def diag_k(self, lam: Optional[float] = None) -> torch.Tensor:
        self.eval_properties(lam)
        out = torch.empty(self.plan.n_dofs, dtype=torch.float32, device=self.device)
        _capi.check(self.lib.pf_diag_k(self._ref(), out.data_ptr(), self._stream()), "pf_diag_k")
        return out

    @_on_engine_stream
    def sparse_k(self, lam: Optional[float] = None) -> torch.Tensor:
        """k_global as a coalesced torch sparse COO tensor (any mesh size; pf_coo_k)."""
        self.eval_properties(lam)
        n, nd = self.plan.n_dofs, 2 * self.plan.dim
        nnz = max(self.plan.n_elems, 0) * nd * nd
        idx = torch.empty((2, max(nnz, 1)), dtype=torch.int64, device=self.device)
        vals = torch.empty(max(nnz, 1), dtype=torch.float32, device=self.device)
        if nnz:
            _capi.check(self.lib.pf_coo_k(self._ref(), idx[0].data_ptr(), idx[1].data_ptr(), vals.data_ptr(), self._stream()),
                        "pf_coo_k")
        return torch.sparse_coo_tensor(idx[:, :nnz], vals[:nnz], (n, n)).coalesce()

    @_on_engine_stream
    def dense_k(self, lam: Optional[float] = None) -> torch.Tensor:
        self.eval_properties(lam)
        n = self.plan.n_dofs
        out = torch.zeros((n, n), dtype=torch.float32, device=self.device)
        _capi.check(self.lib.pf_dense_k(self._ref(), out.data_ptr(), self._stream()), "pf_dense_k")
        return out
