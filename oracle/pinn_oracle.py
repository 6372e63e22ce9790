"""CPU oracle: a numpy (float32) restatement of the reference's PINN+GD hot path.

TEST INFRASTRUCTURE ONLY.  This module is the checker the HIP path is compared with.
Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it;
the product package (pinn_fem_amd/) never does and fails loudly without its HIP library.

Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py
against vectors produced by running the reference itself in the build container
(tests/golden/make_golden.py, torch 2.10.0 CPU, torch.manual_seed set before
parse_problem).  The reference has no test-suite of its own for this path; its three
script-style known answers (FEM/python/test_torch_element.py:14-244) are included.

Each function cites the reference lines it restates (paths relative to the reference
root).  Arithmetic is float32 in the same operation order as the reference wherever numpy
lets us state the order; third-party arithmetic (torch's tanh/softplus/Adam, BLAS
summation order) is restated from its published definition and agrees to float32
round-off, not bit for bit.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

f32 = np.float32


# ----------------------------------------------------------------------------------
# data model  (FEM/python/fem/model.py:11-91, fem/properties.py:43-189)
# ----------------------------------------------------------------------------------
@dataclass
class NetParams:
    """An MLP Linear(in,h)-Tanh-[Linear(h,h)-Tanh]*(L-1)-Linear(h,1)
    (FEM/python/examples/json/generic.py:118-142).  `tensors` is torch's
    `parameters()` order: W1 (h,in), b1 (h), W2 (h,h), b2 (h), ..., Wout (1,h), bout (1)."""
    tensors: List[np.ndarray]
    scale: float = 1.0
    enforce_positive: bool = True

    @property
    def n_linear(self) -> int:
        return len(self.tensors) // 2

    def copy(self) -> "NetParams":
        return NetParams([t.copy() for t in self.tensors], self.scale, self.enforce_positive)


@dataclass
class Problem:
    """FEMModel + measurements (fem/model.py:45-91; generic.py:318-362)."""
    nodes: np.ndarray            # (nnode,2) float64, or (nnode,) for dimension 1
    elements: np.ndarray         # (nelm,2) int
    loads: np.ndarray            # (ndof,) float64
    fixed_dofs: np.ndarray       # int
    dimension: int = 2
    young: object = 1.0          # float | NetParams
    area: object = 1.0
    density: object = 0.0
    measured_vals: Optional[np.ndarray] = None
    measured_dofs: Optional[np.ndarray] = None

    def __post_init__(self):
        self.nodes = np.asarray(self.nodes, dtype=float)
        self.elements = np.asarray(self.elements, dtype=int)
        self.loads = np.asarray(self.loads, dtype=float).reshape(-1)
        self.fixed_dofs = np.asarray(self.fixed_dofs, dtype=int).reshape(-1)

    @property
    def nnode(self) -> int:
        return self.nodes.shape[0]

    @property
    def ndof(self) -> int:
        return self.nnode * self.dimension

    def theta_list(self) -> List[np.ndarray]:
        """young -> area -> density  (fem/model.py:36-42)."""
        out: List[np.ndarray] = []
        for p in (self.young, self.area, self.density):
            if isinstance(p, NetParams):
                out.extend(p.tensors)
        return out

    def has_nn(self) -> bool:
        return any(isinstance(p, NetParams) for p in (self.young, self.area, self.density))


def free_and_fixed_dofs(ndof: int, fixed) -> Tuple[np.ndarray, np.ndarray]:
    """fem/boundary.py:8-13."""
    fixed = np.unique(np.asarray(fixed, dtype=int).reshape(-1))
    mask = np.ones(ndof, dtype=bool)
    mask[fixed] = False
    return np.flatnonzero(mask), fixed


# ----------------------------------------------------------------------------------
# geometry  (fem/nn_assembly.py:64-94 for 2-D, :32-41 for 1-D)
# ----------------------------------------------------------------------------------
@dataclass
class Geometry:
    l0: np.ndarray        # (nelm,) float32 (the reference divides by a Python float = f64 -> f32)
    pattern: np.ndarray   # (nelm, nd, nd) float32 : ke = stiffness * pattern
    nn_input: np.ndarray  # (nelm, dimension) float32 : centroid columns (x[,y]) of the NN input
    dofs: np.ndarray      # (nelm, nd) int : global dof of each local dof


def element_geometry(pb: Problem) -> Geometry:
    ni, nj = pb.elements[:, 0], pb.elements[:, 1]
    if pb.dimension == 1:
        xi, xj = pb.nodes[ni], pb.nodes[nj]
        l0 = np.abs(xj - xi)                                   # nn_assembly.py:32
        if np.any(l0 <= 0.0):
            raise ValueError("Element with zero initial length")
        pat = np.broadcast_to(np.array([[1.0, -1.0], [-1.0, 1.0]], dtype=f32),
                              (len(ni), 2, 2)).copy()           # :40
        xc = ((xi + xj) / 2.0).reshape(-1, 1)                   # :141
        dofs = np.stack([ni, nj], axis=1)
    else:
        xi, xj = pb.nodes[ni], pb.nodes[nj]
        dx = xj - xi
        l0 = np.sqrt(dx[:, 0] * dx[:, 0] + dx[:, 1] * dx[:, 1])  # np.linalg.norm, :65
        if np.any(l0 <= 0.0):
            raise ValueError("Element with zero initial length")
        cx, cy = dx[:, 0] / l0, dx[:, 1] / l0                   # :70-71 (float64)
        c2, s2, cs = cx * cx, cy * cy, cx * cy                  # :80-82 (float64)
        pat = np.empty((len(ni), 4, 4), dtype=np.float64)
        rows = ((c2, cs, -c2, -cs), (cs, s2, -cs, -s2), (-c2, -cs, c2, cs), (-cs, -s2, cs, s2))
        for a in range(4):
            for b in range(4):
                pat[:, a, b] = rows[a][b]
        pat = pat.astype(f32)                                   # torch.tensor(..., float32) :85-94
        xc = (xi + xj) / 2.0                                    # :198
        dofs = np.stack([2 * ni, 2 * ni + 1, 2 * nj, 2 * nj + 1], axis=1)   # :185-189
    return Geometry(l0=l0.astype(f32), pattern=pat, nn_input=xc.astype(f32), dofs=dofs)


# ----------------------------------------------------------------------------------
# MLP + softplus  (generic.py:118-142, fem/properties.py:116-161)
# ----------------------------------------------------------------------------------
def softplus(z: np.ndarray) -> np.ndarray:
    """torch.nn.functional.softplus, beta=1, threshold=20 (properties.py:154)."""
    with np.errstate(over="ignore"):
        return np.where(z > f32(20.0), z, np.log1p(np.exp(z, dtype=f32), dtype=f32)).astype(f32)


def softplus_grad(z: np.ndarray) -> np.ndarray:
    """torch softplus_backward: z>threshold ? 1 : e^z/(e^z+1)."""
    with np.errstate(over="ignore"):
        ez = np.exp(z, dtype=f32)
        return np.where(z > f32(20.0), f32(1.0), ez / (ez + f32(1.0))).astype(f32)


def nn_inputs(geo: Geometry, lam: float) -> np.ndarray:
    """Columns in sorted-key order: load_factor, x[, y]  (properties.py:119-125)."""
    n = geo.nn_input.shape[0]
    return np.concatenate([np.full((n, 1), f32(lam), dtype=f32), geo.nn_input], axis=1)


def mlp_forward(net: NetParams, x: np.ndarray):
    """Returns (z_out (n,), activations list [x, h1, h2, ...])."""
    acts = [x.astype(f32)]
    h = acts[0]
    nl = net.n_linear
    for l in range(nl):
        w, b = net.tensors[2 * l], net.tensors[2 * l + 1]
        z = (h @ w.T.astype(f32) + b.astype(f32)).astype(f32)
        if l < nl - 1:
            h = np.tanh(z, dtype=f32)
            acts.append(h)
        else:
            return z[:, 0], acts
    raise AssertionError


def mlp_backward(net: NetParams, acts, g_z: np.ndarray, acc64: bool = False,
                 abs_out: Optional[list] = None) -> List[np.ndarray]:
    """Gradients of sum_e g_z[e]*z_out[e] w.r.t. net.tensors (autograd of generic.py:141).
    acc64: the sums over ELEMENTS are accumulated in float64 (every per-element term is still the float32
    value): a checker for large meshes whose own summation error is negligible, so that a comparison
    measures the device kernel's error and not the order of a float32 sum over 10^6 terms."""
    nl = net.n_linear
    grads: List[Optional[np.ndarray]] = [None] * (2 * nl)
    g = g_z.reshape(-1, 1).astype(f32)                     # grad wrt pre-activation of layer l
    for l in range(nl - 1, -1, -1):
        w = net.tensors[2 * l].astype(f32)
        h_in = acts[l]
        if abs_out is not None:
            # sum over elements of |term|: the scale a float32 summation error is measured against when the terms cancel
            ga, ha = np.abs(g).astype(np.float64), np.abs(h_in).astype(np.float64)
            abs_out.extend([ga.T @ ha, ga.sum(axis=0)])
        if acc64:
            grads[2 * l] = (g.T.astype(np.float64) @ h_in.astype(np.float64)).astype(f32)
            grads[2 * l + 1] = g.sum(axis=0, dtype=np.float64).astype(f32)
        else:
            grads[2 * l] = (g.T @ h_in).astype(f32)
            grads[2 * l + 1] = g.sum(axis=0, dtype=f32)
        if l > 0:
            gh = (g @ w).astype(f32)
            g = (gh * (f32(1.0) - h_in * h_in)).astype(f32)   # tanh backward: g*(1-y*y)
    if abs_out is not None:                                # appended last layer first: put this net's into tensor order
        abs_out[-2 * nl:] = [x for l in range(nl) for x in abs_out[len(abs_out) - 2 * (l + 1):len(abs_out) - 2 * l]]
    return grads  # type: ignore[return-value]


def property_forward(prop, x: np.ndarray):
    """NNProperty.value / ScalarProperty.value evaluated for every element
    (fem/properties.py:97-161; nn_assembly.py:207-214)."""
    n = x.shape[0]
    if isinstance(prop, NetParams):
        z, acts = mlp_forward(prop, x)
        out = softplus(z) if prop.enforce_positive else z
        return (out * f32(prop.scale)).astype(f32), (z, acts)
    return np.full(n, f32(prop), dtype=f32), None


# ----------------------------------------------------------------------------------
# assembly + loss + gradients  (fem/nn_assembly.py:105-231, fem/solver.py:262-289)
# ----------------------------------------------------------------------------------
@dataclass
class StepOut:
    f_int: np.ndarray
    r: np.ndarray
    loss_physics: float
    loss_data: float
    loss_total: float
    residual_norm: float
    grad_u: np.ndarray
    grad_theta: List[Optional[np.ndarray]]
    stiffness: np.ndarray
    young: np.ndarray
    area: np.ndarray
    grad_theta_abs: Optional[List[np.ndarray]] = None      # acc64 only: sum over elements of |term| per entry


def element_stiffness(pb: Problem, geo: Geometry, lam: float):
    x = nn_inputs(geo, lam)
    e_val, e_ctx = property_forward(pb.young, x)
    a_val, a_ctx = property_forward(pb.area, x)
    s = ((e_val * a_val) / geo.l0).astype(f32)             # nn_assembly.py:74 / :37
    return s, e_val, a_val, e_ctx, a_ctx


def ke_times(geo: Geometry, s: np.ndarray, ve: np.ndarray, fe_mode: str = "reference") -> np.ndarray:
    """(s*pattern) @ v_elem for every element, v_elem = ve[e] (nelm, nd).
    "reference": ke = s*pattern, then a 4-term dot per row, b ascending (nn_assembly.py:96-100).
    "delta": the same product written on d = v_j - v_i (mathematically identical; avoids the
    float32 cancellation of the reference order on long chains; opt-in extension)."""
    nd = geo.dofs.shape[1]
    if fe_mode == "delta":
        # rows of ke = s*pattern (the reference's float32 product, nn_assembly.py:84-94) applied to d = v_j - v_i
        h = nd // 2
        ke = (s[:, None, None] * geo.pattern).astype(f32)
        d = (ve[:, h:] - ve[:, :h]).astype(f32)
        q = np.zeros((len(s), h), dtype=f32)
        for a in range(h):
            acc = (ke[:, a, 0] * d[:, 0]).astype(f32)
            for b in range(1, h):
                acc = (acc + ke[:, a, b] * d[:, b]).astype(f32)
            q[:, a] = acc
        return np.concatenate([-q, q], axis=1).astype(f32)
    ke = (s[:, None, None] * geo.pattern).astype(f32)
    out = np.zeros((len(s), nd), dtype=f32)
    for b in range(nd):                                    # 4-term dot, b ascending
        out = (out + ke[:, :, b] * ve[:, b:b + 1]).astype(f32)
    return out


def internal_force(geo: Geometry, s: np.ndarray, u: np.ndarray, ndof: int,
                   fe_mode: str = "reference") -> np.ndarray:
    """fe = (s*pattern) @ u_elem, then f_int[g] += fe[a] in element order
    (nn_assembly.py:96-100, :226-227)."""
    fe = ke_times(geo, s, u.astype(f32)[geo.dofs], fe_mode)
    f_int = np.zeros(ndof, dtype=f32)
    np.add.at(f_int, geo.dofs.reshape(-1), fe.reshape(-1))  # sequential, element order
    return f_int


def loss_and_grads(pb: Problem, geo: Geometry, u: np.ndarray, lam: float,
                   alpha_physics: float = 1.0, alpha_data: float = 100.0,
                   want_grads: bool = True, fe_mode: str = "reference", acc64: bool = False) -> StepOut:
    """acc64: accumulate the sums over elements / dofs (loss terms, parameter gradients) in float64 — see
    mlp_backward; per-element arithmetic stays float32."""
    sdt = np.float64 if acc64 else f32
    u = u.astype(f32)
    ndof = pb.ndof
    free, fixed = free_and_fixed_dofs(ndof, pb.fixed_dofs)
    s, e_val, a_val, e_ctx, a_ctx = element_stiffness(pb, geo, lam)
    f_int = internal_force(geo, s, u, ndof, fe_mode)
    f_ext = pb.loads.astype(f32)
    r = (f_int[free] - f32(lam) * f_ext[free]).astype(f32)         # solver.py:267-269
    loss_p = f32(0.5) * f32(np.sum(r * r, dtype=sdt))               # :270
    has_meas = pb.measured_vals is not None and pb.measured_dofs is not None
    use_data = has_meas and alpha_data > 0 and len(pb.measured_vals) > 0
    if use_data:
        mv = pb.measured_vals.astype(f32)
        md = pb.measured_dofs.astype(int)
        d = (mv - u[md]).astype(f32)                                # :274
        loss_d = f32(np.mean(d * d, dtype=sdt))                     # :275
        loss = f32(alpha_physics) * loss_p + f32(alpha_data) * loss_d   # :277-279
    else:
        loss_d = f32(0.0)
        loss = f32(alpha_physics) * loss_p                          # :283
    rn = f32(np.sqrt(np.sum(r * r, dtype=sdt)))                     # torch.norm :306
    grad_u = np.zeros(ndof, dtype=f32)
    grad_theta: List[Optional[np.ndarray]] = []
    abs_out: Optional[list] = [] if acc64 else None
    if want_grads:
        g_f = np.zeros(ndof, dtype=f32)
        g_f[free] = (f32(alpha_physics) * r).astype(f32)            # dL/df_int (free rows only)
        g_fe = g_f[geo.dofs]                                        # (nelm, nd)
        # grad_u[dof_b] += sum_a g_fe[a]*ke[a][b]   (ke symmetric: the same operator applied to g_fe)
        gue = ke_times(geo, s, g_fe, fe_mode)
        np.add.at(grad_u, geo.dofs.reshape(-1), gue.reshape(-1))
        if use_data:
            g_d = (f32(alpha_data) / f32(len(mv))) * (f32(2.0) * d)
            np.add.at(grad_u, md, (-g_d).astype(f32))
        # g_s = sum_ab g_fe[a]*pattern[a][b]*u[b]
        pu = ke_times(geo, np.ones_like(s), u[geo.dofs], fe_mode)
        g_s = np.sum(g_fe * pu, axis=1, dtype=f32)
        g_ea = (g_s / geo.l0).astype(f32)
        for prop, ctx, other in ((pb.young, e_ctx, a_val), (pb.area, a_ctx, e_val)):
            if isinstance(prop, NetParams):
                z, acts = ctx
                g_out = (g_ea * other * f32(prop.scale)).astype(f32)
                g_z = (g_out * softplus_grad(z)).astype(f32) if prop.enforce_positive else g_out
                grad_theta.extend(mlp_backward(prop, acts, g_z, acc64, abs_out))
        if isinstance(pb.density, NetParams):
            # never evaluated by the assembly: grad stays None (nn_assembly.py:207-208)
            grad_theta.extend([None] * len(pb.density.tensors))
    return StepOut(f_int=f_int, r=r, loss_physics=float(loss_p), loss_data=float(loss_d),
                   loss_total=float(loss), residual_norm=float(rn), grad_u=grad_u,
                   grad_theta=grad_theta, stiffness=s, young=e_val, area=a_val, grad_theta_abs=abs_out)


def vjp_internal_force(pb: Problem, geo: Geometry, u: np.ndarray, lam: float, g_f: np.ndarray,
                       fe_mode: str = "reference"):
    """(K^T g_f, d<g_f, f_int>/dtheta) for an arbitrary upstream gradient g_f (n_dofs,): the part of
    loss.backward() (solver.py:289) that goes through assemble_system_torch.  Used by the sharded
    driver tests, where g_f on shared dofs comes from another rank."""
    u = u.astype(f32)
    s, e_val, a_val, e_ctx, a_ctx = element_stiffness(pb, geo, lam)
    g_fe = g_f.astype(f32)[geo.dofs]
    grad_u = np.zeros(pb.ndof, dtype=f32)
    np.add.at(grad_u, geo.dofs.reshape(-1), ke_times(geo, s, g_fe, fe_mode).reshape(-1))
    pu = ke_times(geo, np.ones_like(s), u[geo.dofs], fe_mode)
    g_ea = (np.sum(g_fe * pu, axis=1, dtype=f32) / geo.l0).astype(f32)
    grad_theta: List[Optional[np.ndarray]] = []
    for prop, ctx, other in ((pb.young, e_ctx, a_val), (pb.area, a_ctx, e_val)):
        if isinstance(prop, NetParams):
            z, acts = ctx
            g_out = (g_ea * other * f32(prop.scale)).astype(f32)
            g_z = (g_out * softplus_grad(z)).astype(f32) if prop.enforce_positive else g_out
            grad_theta.extend(mlp_backward(prop, acts, g_z))
    return grad_u, grad_theta


def dense_stiffness(pb: Problem, geo: Geometry, lam: float) -> np.ndarray:
    """k_global of assemble_system_torch (nn_assembly.py:228-229); small ndof only."""
    s, *_ = element_stiffness(pb, geo, lam)
    ke = (s[:, None, None] * geo.pattern).astype(f32)
    k = np.zeros((pb.ndof, pb.ndof), dtype=f32)
    nd = geo.dofs.shape[1]
    for e in range(len(s)):
        for a in range(nd):
            for b in range(nd):
                k[geo.dofs[e, a], geo.dofs[e, b]] += ke[e, a, b]
    return k


def diag_stiffness(pb: Problem, geo: Geometry, lam: float) -> np.ndarray:
    s, *_ = element_stiffness(pb, geo, lam)
    nd = geo.dofs.shape[1]
    d = np.zeros(pb.ndof, dtype=f32)
    ke_d = (s[:, None] * np.stack([geo.pattern[:, a, a] for a in range(nd)], axis=1)).astype(f32)
    np.add.at(d, geo.dofs.reshape(-1), ke_d.reshape(-1))
    return d


# ----------------------------------------------------------------------------------
# torch.optim.Adam, single-tensor path, defaults  (call sites fem/solver.py:234-236,
# 292-294; arithmetic: torch/optim/adam.py::_single_tensor_adam, torch 2.10)
# ----------------------------------------------------------------------------------
@dataclass
class AdamState:
    lr: float
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    step: Dict[int, int] = field(default_factory=dict)
    m: Dict[int, np.ndarray] = field(default_factory=dict)
    v: Dict[int, np.ndarray] = field(default_factory=dict)

    def update(self, params: Sequence[np.ndarray], grads: Sequence[Optional[np.ndarray]]):
        for i, (p, g) in enumerate(zip(params, grads)):
            if g is None:                      # params without grad are skipped
                continue
            if i not in self.m:
                self.m[i] = np.zeros_like(p, dtype=f32)
                self.v[i] = np.zeros_like(p, dtype=f32)
                self.step[i] = 0
            self.step[i] += 1
            t = self.step[i]
            g = g.astype(f32).reshape(p.shape)
            m, v = self.m[i], self.v[i]
            m += f32(1 - self.beta1) * (g - m)                        # exp_avg.lerp_(grad, 1-b1)
            v *= f32(self.beta2)                                      # mul_(beta2)
            v += (f32(1 - self.beta2) * g) * g                        # addcmul_(g, g, 1-b2)
            bc1 = 1 - self.beta1 ** t
            bc2 = 1 - self.beta2 ** t
            step_size = self.lr / bc1
            bc2_sqrt = bc2 ** 0.5
            denom = np.sqrt(v, dtype=f32) / f32(bc2_sqrt) + f32(self.eps)
            p += f32(-step_size) * (m / denom)                        # addcdiv_


# ----------------------------------------------------------------------------------
# solver drivers  (fem/solver.py:35-75, 83-400, 520-651, 1045-1167)
# ----------------------------------------------------------------------------------
@dataclass
class SolverConfig:
    """fem/solver.py:35-62 (same fields, same defaults)."""
    max_iterations: int = 1000
    tolerance: float = 1e-6
    print_every: int = 10
    n_increments: int = 10
    load_factor_initial: float = 0.0
    load_factor_final: float = 1.0
    min_denominator: float = 1e-10
    learning_rate_u: float = 1e-7
    learning_rate_theta: float = 1e-4
    alpha_physics: float = 1.0
    alpha_data: float = 100.0
    method: str = "auto"
    preconditioning: bool = False


@dataclass
class SolverResult:
    """fem/solver.py:65-75."""
    displacements: np.ndarray
    reactions: np.ndarray
    converged: bool
    history: List[Dict[str, float]] = field(default_factory=list)
    nn_parameters: Optional[Dict[str, np.ndarray]] = None


def solve_gd(pb: Problem, config: Optional[SolverConfig] = None,
             target_load_factor: float = 1.0, u_initial: Optional[np.ndarray] = None,
             skip_preconditioning: bool = False, geo: Optional[Geometry] = None,
             call_log: Optional[list] = None, fe_mode: str = "reference") -> SolverResult:
    """fem/solver.py:83-400.  Measurements travel inside `pb`."""
    config = config or SolverConfig()
    geo = geo or element_geometry(pb)

    if config.preconditioning and not skip_preconditioning:            # :114-198
        pre = copy.deepcopy(config)
        pre.max_iterations = min(300, config.max_iterations // 3)
        pre.tolerance = max(1e-4, config.tolerance * 10)
        pre.preconditioning = False
        pre_res = solve_gd(pb, pre, target_load_factor, u_initial, True, geo, call_log, fe_mode)
        if pre_res.converged and pre_res.history[-1].get("residual_norm", 1.0) < config.tolerance:
            return pre_res
        main = copy.deepcopy(config)
        main.max_iterations = config.max_iterations - pre.max_iterations
        main.preconditioning = False
        main_res = solve_gd(pb, main, target_load_factor,
                            pre_res.displacements.flatten().astype(f32), True, geo, call_log, fe_mode)
        off = pre_res.history[-1].get("iteration", 0) if pre_res.history else 0
        merged = list(pre_res.history)
        for h in main_res.history:
            h2 = dict(h)
            h2["iteration"] = h.get("iteration", 0) + off
            merged.append(h2)
        main_res.history = merged
        return main_res

    theta = pb.theta_list()
    ndof = pb.ndof
    u = (np.zeros(ndof, dtype=f32) if u_initial is None
         else np.asarray(u_initial, dtype=f32).copy())                 # :205-216
    free, fixed = free_and_fixed_dofs(ndof, pb.fixed_dofs)
    has_meas = pb.measured_vals is not None and pb.measured_dofs is not None
    opt_u = AdamState(lr=config.learning_rate_u)                       # fresh every call :234
    opt_t = AdamState(lr=config.learning_rate_theta) if theta else None
    history: List[Dict[str, float]] = []
    converged = False
    lam = target_load_factor
    for it in range(config.max_iterations):                            # :252
        st = loss_and_grads(pb, geo, u, lam, config.alpha_physics, config.alpha_data,
                            fe_mode=fe_mode)
        opt_u.update([u], [st.grad_u])                                 # :292
        if opt_t is not None:
            opt_t.update(theta, st.grad_theta)                         # :294
        u[fixed] = f32(0.0)                                            # :297-298
        u_norm = float(np.sqrt(np.sum(u[free] * u[free], dtype=f32)))  # :304
        entry = {"iteration": float(it + 1), "loss_total": st.loss_total,
                 "loss_physics": st.loss_physics,
                 "loss_data": st.loss_data if has_meas else 0.0,
                 "u_norm": u_norm, "residual_norm": st.residual_norm}
        if theta:
            entry["theta_norm"] = float(sum(
                float(np.sqrt(np.sum(p.astype(f32) ** 2, dtype=f32))) for p in theta))   # :319
        history.append(entry)
        if it > 10:                                                    # :341-355
            if st.residual_norm < config.tolerance:
                converged = True
                break
            if not np.isnan(st.loss_total) and st.loss_total < config.tolerance:
                converged = True
                break
    # reactions from a re-assembly  (:374-385)
    s, *_ = element_stiffness(pb, geo, lam)
    f_int = internal_force(geo, s, u, ndof, fe_mode)
    reac = (f_int - f32(lam) * pb.loads.astype(f32)).astype(f32)
    reac[free] = f32(0.0)
    shape = (-1, 1) if pb.dimension == 1 else (pb.nnode, pb.dimension)
    res = SolverResult(displacements=u.reshape(shape), reactions=reac.reshape(shape),
                       converged=converged, history=history,
                       nn_parameters=({f"param_{i}": p for i, p in enumerate(theta)}
                                      if theta else None))
    if call_log is not None:
        call_log.append({"load_factor": float(lam), "n_history": len(history),
                         "converged": converged, "max_iterations": config.max_iterations,
                         "tolerance": config.tolerance})
    return res


def assemble_system_f64(pb: Problem, u: np.ndarray):
    """fem/assembly.py:16-75 with fem/element.py:15-42 (1-D) and :45-102 (2-D linear truss): dense float64
    K, f_int = sum of ke @ u_e, max |strain|.  Scalar materials only (the NumPy twin has no nets)."""
    if pb.has_nn():
        raise ValueError("the float64 NumPy assembly works on scalar materials")
    ndof = pb.ndof
    k = np.zeros((ndof, ndof), dtype=float)
    f = np.zeros(ndof, dtype=float)
    max_eps = 0.0
    young, area = float(pb.young), float(pb.area)
    for ni, nj in pb.elements:
        if pb.dimension == 1:
            x = pb.nodes.reshape(-1)
            l0 = abs(float(x[nj] - x[ni]))                                   # element.py:24-26
            stiff = (young * area) / l0
            ke = stiff * np.array([[1.0, -1.0], [-1.0, 1.0]])
            dofs = np.array([ni, nj])
            fe = stiff * np.array([u[ni] - u[nj], u[nj] - u[ni]])            # element.py:40
            eps = (u[nj] - u[ni]) / l0
        else:
            dx0 = pb.nodes[nj] - pb.nodes[ni]
            l0 = float(np.linalg.norm(dx0))                                  # element.py:60-61
            cx, cy = dx0[0] / l0, dx0[1] / l0
            dofs = np.array([2 * ni, 2 * ni + 1, 2 * nj, 2 * nj + 1])
            ue = u[dofs]
            eps = (cx * (ue[2] - ue[0]) + cy * (ue[3] - ue[1])) / l0        # :70-74
            stiff = (young * area) / l0
            c2, s2, cs = cx * cx, cy * cy, cx * cy
            ke = stiff * np.array([[c2, cs, -c2, -cs], [cs, s2, -cs, -s2],
                                   [-c2, -cs, c2, cs], [-cs, -s2, cs, s2]])  # :85-95
            fe = ke @ ue                                                     # :99-100
        k[np.ix_(dofs, dofs)] += ke
        f[dofs] += fe
        max_eps = max(max_eps, abs(float(eps)))
    return k, f, max_eps


def solve_nr(pb: Problem, config: Optional[SolverConfig] = None, target_load_factor: float = 1.0,
             u_initial=None) -> SolverResult:
    """fem/solver.py:408-512: Newton-Raphson with a dense float64 solve on K_ff; u starts from zero
    whatever u_initial says (:443)."""
    config = config or SolverConfig()
    if pb.has_nn():
        raise ValueError("Newton-Raphson solver with NN materials not fully supported yet. "
                         "Use solve_gd() for problems with NN parameters.")
    u = np.zeros(pb.ndof, dtype=float)
    free, fixed = free_and_fixed_dofs(pb.ndof, pb.fixed_dofs)
    lam = target_load_factor
    f_ext = lam * pb.loads
    converged, res_norm, max_e, ite = False, np.inf, 0.0, -1
    for ite in range(config.max_iterations):
        k, f_int, max_e = assemble_system_f64(pb, u)
        rhs = f_ext - f_int
        try:
            du_f = np.linalg.solve(k[np.ix_(free, free)], rhs[free])
        except np.linalg.LinAlgError as exc:
            raise RuntimeError("Tangent stiffness became singular during solve") from exc
        du = np.zeros_like(u)
        du[free] = du_f
        u += du
        res_norm = np.linalg.norm(du) / max(np.linalg.norm(u), getattr(config, "min_denominator", 1e-10))
        if res_norm <= config.tolerance:
            converged = True
            break
    hist = [{"load_factor": float(lam), "iterations": float(ite + 1), "residual": float(res_norm),
             "max_strain": float(max_e), "converged": float(1.0 if converged else 0.0)}]
    k, _, _ = assemble_system_f64(pb, u)
    reactions = k @ u - lam * pb.loads
    reactions[free] = 0.0
    shape = (-1, 1) if pb.dimension == 1 else (pb.nnode, pb.dimension)
    return SolverResult(displacements=u.reshape(shape), reactions=reactions.reshape(shape),
                        converged=converged, history=hist)


def solve_hybrid(pb: Problem, config: Optional[SolverConfig] = None,
                 target_load_factor: float = 1.0, u_initial: Optional[np.ndarray] = None,
                 geo: Optional[Geometry] = None, call_log: Optional[list] = None,
                 fe_mode: str = "reference") -> SolverResult:
    """fem/solver.py:520-651 — NN branch only (phase 2 is GD again when NNs are present)."""
    config = config or SolverConfig()
    geo = geo or element_geometry(pb)
    gd_res = None
    gd_cfg = None
    if config.preconditioning:                                         # :554-582
        gd_cfg = copy.deepcopy(config)
        gd_cfg.max_iterations = min(300, config.max_iterations // 3)
        gd_cfg.tolerance = max(1e-4, config.tolerance * 10)
        gd_res = solve_gd(pb, gd_cfg, target_load_factor, u_initial, True, geo, call_log, fe_mode)
        if gd_res.converged and gd_res.history[-1].get("residual_norm", 1.0) < config.tolerance:
            return gd_res
    if not pb.has_nn():                                                # :653-692: the GD -> NR switch
        nr = solve_nr(pb, config, target_load_factor)
        if gd_res:
            off = gd_res.history[-1].get("iteration", 0) if gd_res.history else 0
            merged = list(gd_res.history)
            last = dict(nr.history[-1])
            last["iteration"] = off + nr.history[-1].get("iterations", 1)
            merged.append(last)
            nr.history = merged
        return nr
    fin = copy.deepcopy(config)                                        # :600-604
    fin.max_iterations = config.max_iterations - (gd_cfg.max_iterations if gd_res else 0)
    u_warm = gd_res.displacements.flatten().astype(f32) if gd_res else u_initial
    final = solve_gd(pb, fin, target_load_factor, u_warm, True, geo, call_log, fe_mode)
    if gd_res:                                                         # :623-645
        off = gd_res.history[-1].get("iteration", 0) if gd_res.history else 0
        merged = list(gd_res.history)
        for h in final.history:
            h2 = dict(h)
            h2["iteration"] = h.get("iteration", 0) + off
            merged.append(h2)
        final.history = merged
    return final


def solve(pb: Problem, config: Optional[SolverConfig] = None,
          call_log: Optional[list] = None, fe_mode: str = "reference") -> SolverResult:
    """fem/solver.py:1045-1167 (methods gd / hybrid; 'auto' resolves to gd here because the
    NR branch is out of scope)."""
    config = config or SolverConfig()
    if config.method != "auto":
        method = config.method.lower()
    else:                                                               # :1075-1090
        has_meas = pb.measured_vals is not None and pb.measured_dofs is not None
        method = "gd" if (pb.has_nn() or has_meas) else "nr"
    geo = element_geometry(pb)
    result = None
    u_cur = None
    for iinc in range(1, config.n_increments + 1):
        lam = config.load_factor_initial + (iinc / config.n_increments) * (
            config.load_factor_final - config.load_factor_initial)      # :1096-1098
        u0 = None if u_cur is None else np.asarray(u_cur, dtype=f32)
        if method == "gd":
            result = solve_gd(pb, config, lam, u0, False, geo, call_log, fe_mode)
        elif method == "hybrid":
            result = solve_hybrid(pb, config, lam, u0, geo, call_log, fe_mode)
        elif method == "nr":
            result = solve_nr(pb, config, lam, u0)
        else:
            raise ValueError(f"Unknown solver method: {method}")
        u_cur = result.displacements.flatten()
        if not result.converged:                                       # :1161-1165
            break
    return result


# ----------------------------------------------------------------------------------
# (R) reference-ORDER mode: one element at a time, like assemble_system_torch's Python loop
# (fem/nn_assembly.py:181-229): batch-1 MLP call per property, 4x4 (2x2) element matrix, 16 (4) indexed
# `+=` into the dense K and 4 (2) into f_int, then the per-element chain rule that autograd applies
# in reverse element order.  Same numbers as the vectorised functions above (tests/test_oracle_golden.py),
# at the cost class of the reference (10^2..10^3 element-evals/s): bench.py's cpu_baseline rows "(R)".
# ----------------------------------------------------------------------------------
def loss_and_grads_loop(pb: Problem, geo: Geometry, u: np.ndarray, lam: float,
                        alpha_physics: float = 1.0, alpha_data: float = 100.0) -> StepOut:
    u = u.astype(f32)
    ndof, nelm, nd = pb.ndof, geo.dofs.shape[0], geo.dofs.shape[1]
    free, fixed = free_and_fixed_dofs(ndof, pb.fixed_dofs)
    k_global = np.zeros((ndof, ndof), dtype=f32)            # nn_assembly.py:128 (never read by the loss)
    f_int = np.zeros(ndof, dtype=f32)
    x_all = nn_inputs(geo, lam)
    ctx = []
    for e in range(nelm):                                   # :181
        x = x_all[e:e + 1]
        e_val, e_ctx = property_forward(pb.young, x)        # :207 (batch of one)
        a_val, a_ctx = property_forward(pb.area, x)
        s = (e_val[0] * a_val[0]) / geo.l0[e]               # :74
        ke = (f32(s) * geo.pattern[e]).astype(f32)
        dofs = geo.dofs[e]
        ue = u[dofs]
        fe = np.zeros(nd, dtype=f32)
        for b in range(nd):                                 # ke @ u_e, b ascending (:96-100)
            fe = (fe + ke[:, b] * ue[b]).astype(f32)
        for a in range(nd):                                 # :226-229
            f_int[dofs[a]] = f32(f_int[dofs[a]] + fe[a])
            for b in range(nd):
                k_global[dofs[a], dofs[b]] = f32(k_global[dofs[a], dofs[b]] + ke[a, b])
        ctx.append((e_val[0], a_val[0], e_ctx, a_ctx, f32(s)))
    f_ext = pb.loads.astype(f32)
    r = (f_int[free] - f32(lam) * f_ext[free]).astype(f32)
    loss_p = f32(0.5) * np.sum(r * r, dtype=f32)
    has_meas = pb.measured_vals is not None and pb.measured_dofs is not None
    use_data = has_meas and alpha_data > 0 and len(pb.measured_vals) > 0
    if use_data:
        mv, md = pb.measured_vals.astype(f32), pb.measured_dofs.astype(int)
        d = (mv - u[md]).astype(f32)
        loss_d = np.mean(d * d, dtype=f32)
        loss = f32(alpha_physics) * loss_p + f32(alpha_data) * loss_d
    else:
        loss_d = f32(0.0)
        loss = f32(alpha_physics) * loss_p
    rn = f32(np.sqrt(np.sum(r * r, dtype=f32)))
    g_f = np.zeros(ndof, dtype=f32)
    g_f[free] = (f32(alpha_physics) * r).astype(f32)
    grad_u = np.zeros(ndof, dtype=f32)
    if use_data:
        np.add.at(grad_u, md, (-(f32(alpha_data) / f32(len(mv))) * (f32(2.0) * d)).astype(f32))
    nets = [p for p in (pb.young, pb.area) if isinstance(p, NetParams)]
    acc: List[List[np.ndarray]] = [[np.zeros_like(t, dtype=f32) for t in p.tensors] for p in nets]
    for e in range(nelm):
        e_val, a_val, e_ctx, a_ctx, s = ctx[e]
        dofs = geo.dofs[e]
        g_fe, ue = g_f[dofs], u[dofs]
        ke = (s * geo.pattern[e]).astype(f32)
        for b in range(nd):                                 # grad wrt u_e: ke^T g_fe
            grad_u[dofs[b]] = f32(grad_u[dofs[b]] + np.sum(ke[:, b] * g_fe, dtype=f32))
        pu = np.zeros(nd, dtype=f32)
        for b in range(nd):
            pu = (pu + geo.pattern[e][:, b] * ue[b]).astype(f32)
        g_ea = f32(np.sum(g_fe * pu, dtype=f32) / geo.l0[e])
        k = 0
        for prop, c, other in ((pb.young, e_ctx, a_val), (pb.area, a_ctx, e_val)):
            if isinstance(prop, NetParams):
                z, acts = c
                g_out = np.array([g_ea * other * f32(prop.scale)], dtype=f32)
                g_z = (g_out * softplus_grad(z)).astype(f32) if prop.enforce_positive else g_out
                for t, g in enumerate(mlp_backward(prop, acts, g_z)):
                    acc[k][t] = (acc[k][t] + g.reshape(acc[k][t].shape)).astype(f32)
                k += 1
    grad_theta: List[Optional[np.ndarray]] = [g for lst in acc for g in lst]
    if isinstance(pb.density, NetParams):
        grad_theta.extend([None] * len(pb.density.tensors))
    s_all = np.array([c[4] for c in ctx], dtype=f32)
    return StepOut(f_int=f_int, r=r, loss_physics=float(loss_p), loss_data=float(loss_d), loss_total=float(loss),
                   residual_norm=float(rn), grad_u=grad_u, grad_theta=grad_theta, stiffness=s_all,
                   young=np.array([c[0] for c in ctx], dtype=f32), area=np.array([c[1] for c in ctx], dtype=f32))


def gd_iterations_loop(pb: Problem, config: SolverConfig, lam: float, n_iter: int) -> np.ndarray:
    """n_iter iterations of solve_gd's loop body (solver.py:254-298) in reference-order mode; returns u."""
    geo = element_geometry(pb)
    theta = pb.theta_list()
    u = np.zeros(pb.ndof, dtype=f32)
    free, fixed = free_and_fixed_dofs(pb.ndof, pb.fixed_dofs)
    opt_u = AdamState(lr=config.learning_rate_u)
    opt_t = AdamState(lr=config.learning_rate_theta) if theta else None
    for _ in range(n_iter):
        st = loss_and_grads_loop(pb, geo, u, lam, config.alpha_physics, config.alpha_data)
        opt_u.update([u], [st.grad_u])
        if opt_t is not None:
            opt_t.update(theta, st.grad_theta)
        u[fixed] = f32(0.0)
    return u


# ----------------------------------------------------------------------------------
# pinn_inverse_problem_gd: the callee FEM/python/api_pinn_gradient_descent.py:19 imports but the reference
# never defines (ImportError at import).  PARITY UNPINNED against the reference: there is no
# reference behaviour.  This restates the PRODUCT's definition (pinn_fem_amd/fem/nn_solver_gd.py), which takes the
# contract from the wrapper's call site (api_pinn_gradient_descent.py:102-121, 154-176) and the arithmetic from the
# nearest existing code, the legacy solver fem/nn_solver_gd.py:105-125:
#     loss = alpha * mean(r_free^2) + beta * mean((u_meas - u[md])^2),  r = (E A / E0 A0) K_1 u - F / (E0 A0),
#     E = E0 exp(p_E), A = A0 exp(p_A);  torch.optim.Adam, two parameter groups (u: lr * max|u_meas|, p: lr).
# ----------------------------------------------------------------------------------
def pinn_inverse_problem_gd(nodes, elements, f_ext, fixed_dofs, young_init, area_init, u_measured, measured_dofs,
                            n_iterations=500, learning_rate=1e-3, alpha=1.0, beta=100.0,
                            young_bounds=None, area_bounds=None) -> Dict:
    nodes = np.asarray(nodes, dtype=float)
    elements = np.asarray(elements, dtype=int)
    f_ext = np.asarray(f_ext, dtype=float).reshape(-1)
    pb = Problem(nodes=nodes, elements=elements, loads=f_ext, fixed_dofs=np.asarray(fixed_dofs, dtype=int),
                 dimension=2, young=1.0, area=1.0, density=1.0)
    geo = element_geometry(pb)
    ndof = pb.ndof
    free, fixed = free_and_fixed_dofs(ndof, pb.fixed_dofs)
    md = np.asarray(measured_dofs, dtype=int)
    um = np.asarray(u_measured, dtype=f32)
    fx = f_ext.astype(f32)
    ea0 = float(young_init) * float(area_init)
    s1 = (f32(1.0) / geo.l0).astype(f32)                    # unit stiffness: E = A = 1
    u = np.zeros(ndof, dtype=f32)
    p = np.zeros(2, dtype=f32)
    u_scale = float(np.max(np.abs(um))) if um.size else 1.0
    opt_u = AdamState(lr=learning_rate * max(u_scale, 1e-30))
    opt_p = AdamState(lr=learning_rate)
    lo = hi = None
    if young_bounds is not None and area_bounds is not None:
        lo = np.array([np.log(young_bounds[0] / young_init), np.log(area_bounds[0] / area_init)], dtype=f32)
        hi = np.array([np.log(young_bounds[1] / young_init), np.log(area_bounds[1] / area_init)], dtype=f32)
    history = []
    nf, nm = f32(len(free)), f32(max(len(md), 1))
    for it in range(int(n_iterations)):
        k1u = internal_force(geo, s1, u, ndof)
        c = f32(np.exp(f32(p[0] + p[1]), dtype=f32))
        r = (c * k1u[free] - fx[free] / f32(ea0)).astype(f32)
        loss_p = np.mean(r * r, dtype=f32)
        d = (um - u[md]).astype(f32)
        loss_d = np.mean(d * d, dtype=f32) if len(md) else f32(0.0)
        loss = f32(alpha) * loss_p + f32(beta) * loss_d
        g_r = (f32(alpha) * f32(2.0) * r / nf).astype(f32)
        g_f = np.zeros(ndof, dtype=f32)
        g_f[free] = (c * g_r).astype(f32)
        grad_u = np.zeros(ndof, dtype=f32)
        np.add.at(grad_u, geo.dofs.reshape(-1), ke_times(geo, s1, g_f[geo.dofs]).reshape(-1))
        if len(md):
            np.add.at(grad_u, md, (-(f32(beta) * f32(2.0) / nm) * d).astype(f32))
        g_c = np.sum(g_r * k1u[free], dtype=f32) * c        # d loss / d (p_E + p_A)
        opt_u.update([u], [grad_u])
        opt_p.update([p], [np.array([g_c, g_c], dtype=f32)])
        u[fixed] = f32(0.0)
        if lo is not None:
            np.clip(p, lo, hi, out=p)
        history.append({"iteration": it + 1, "loss_total": float(loss), "loss_physics": float(loss_p),
                        "loss_data": float(loss_d), "young": float(young_init * np.exp(float(p[0]))),
                        "area": float(area_init * np.exp(float(p[1])))})
    return {"u_final": u.astype(np.float64), "young_final": float(young_init * np.exp(float(p[0]))),
            "area_final": float(area_init * np.exp(float(p[1]))), "history": history}
