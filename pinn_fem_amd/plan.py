"""Mesh plan: immutable, mesh-derived arrays the HIP kernels read (pf_mesh in the C ABI).

Host logic only (numpy, vectorised so 10^6..10^7 elements build in seconds).  It replaces the
per-iteration Python work of the reference's assembly loop:
  dof numbering, centroid, direction cosines   FEM/python/fem/nn_assembly.py:181-205, 64-82
  free/fixed partition                         FEM/python/fem/boundary.py:8-13
and adds the node -> incident-element CSR that makes the assembly a deterministic gather.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from ._capi import PF_DOF_FIXED, PF_DOF_MEASURED


@dataclass
class HostPlan:
    """numpy arrays in exactly the layout pf_mesh expects."""
    dim: int
    n_nodes: int
    n_elems: int
    n_dofs: int
    conn: np.ndarray       # int32 [n_elems,2]
    egeo: np.ndarray       # float32 [n_elems,4] c2, cs, s2, l0
    ecent: np.ndarray      # float32 [n_elems,dim]
    adj_ptr: np.ndarray    # int32 [n_nodes+1]
    adj: np.ndarray        # int32 [2*n_elems]  (elem<<1)|end, ascending element id per node
    adj_other: np.ndarray  # int32 [2*n_elems]  node at the other end of that element
    f_ext: np.ndarray      # float32 [n_dofs]
    dof_flags: np.ndarray  # uint8 [n_dofs]
    meas_val: np.ndarray   # float32 [n_dofs]
    n_meas: int
    free_dofs: np.ndarray  # int64 (host-side convenience)
    fixed_dofs: np.ndarray


def free_and_fixed_dofs(ndof: int, fixed_dofs):
    """FEM/python/fem/boundary.py:8-13."""
    fixed = np.unique(np.asarray(fixed_dofs, dtype=int).reshape(-1))
    mask = np.ones(ndof, dtype=bool)
    mask[fixed] = False
    return np.flatnonzero(mask), fixed


def build_host_plan(nodes, elements, loads, fixed_dofs, dimension: int,
                    measured_disp: Optional[np.ndarray] = None,
                    measured_dofs: Optional[np.ndarray] = None) -> HostPlan:
    nodes = np.asarray(nodes, dtype=np.float64)
    elements = np.asarray(elements, dtype=np.int64)
    if elements.ndim != 2 or elements.shape[1] != 2:
        raise ValueError("elements must have shape (nelm, 2)")
    n_elems = elements.shape[0]
    n_nodes = nodes.shape[0]
    if n_elems and (elements.min() < 0 or elements.max() >= n_nodes):
        raise ValueError("element connectivity references a node out of range")
    if n_elems >= 2 ** 30:
        raise ValueError("too many elements for the int32 adjacency encoding")
    ni, nj = elements[:, 0], elements[:, 1]
    if dimension == 1:
        if nodes.ndim != 1:
            raise ValueError("For 1D, nodes must be 1D array of positions")
        xi, xj = nodes[ni], nodes[nj]
        l0 = np.abs(xj - xi)                                   # nn_assembly.py:32
        egeo = np.stack([np.ones_like(l0), np.zeros_like(l0), np.zeros_like(l0), l0], axis=1)
        ecent = ((xi + xj) / 2.0).reshape(-1, 1)               # :141
    elif dimension == 2:
        if nodes.ndim != 2 or nodes.shape[1] != 2:
            raise ValueError("For 2D, nodes must have shape (nnode, 2)")
        dx = nodes[nj] - nodes[ni]
        l0 = np.sqrt(dx[:, 0] * dx[:, 0] + dx[:, 1] * dx[:, 1])   # :65
        with np.errstate(divide="ignore", invalid="ignore"):
            cx, cy = dx[:, 0] / l0, dx[:, 1] / l0              # :70-71 (float64)
        egeo = np.stack([cx * cx, cx * cy, cy * cy, l0], axis=1)  # :80-82, cast to f32 like :85-94
        ecent = (nodes[ni] + nodes[nj]) / 2.0                  # :198
    else:
        raise ValueError("dimension must be 1 or 2")
    if n_elems and np.any(l0 <= 0.0):
        raise ValueError("Element with zero initial length")   # :33-34 / :66-67
    n_dofs = n_nodes * dimension

    flat = elements.reshape(-1)                                # position k = (elem<<1)|end
    order = np.argsort(flat, kind="stable")                    # ascending element id inside a node
    counts = np.bincount(flat, minlength=n_nodes)
    adj_ptr = np.zeros(n_nodes + 1, dtype=np.int64)
    np.cumsum(counts, out=adj_ptr[1:])

    loads = np.asarray(loads, dtype=np.float64).reshape(-1)
    if loads.size != n_dofs:
        raise ValueError(f"loads size must be {n_dofs}, got {loads.size}")
    fixed_dofs = np.asarray(fixed_dofs, dtype=np.int64).reshape(-1)
    if np.any(fixed_dofs < 0) or np.any(fixed_dofs >= n_dofs):
        raise ValueError("fixed_dofs contain out-of-range indices")
    free, fixed = free_and_fixed_dofs(n_dofs, fixed_dofs)
    flags = np.zeros(n_dofs, dtype=np.uint8)
    flags[fixed] |= PF_DOF_FIXED
    meas_val = np.zeros(n_dofs, dtype=np.float32)
    n_meas = 0
    if measured_disp is not None and measured_dofs is not None:
        md = np.asarray(measured_dofs, dtype=np.int64).reshape(-1)
        mv = np.asarray(measured_disp, dtype=np.float64).reshape(-1)
        if md.size != mv.size:
            raise ValueError("measured_disp and measured_dofs must have the same length")
        if md.size and (md.min() < 0 or md.max() >= n_dofs):
            raise IndexError("measured_dofs contain out-of-range indices")
        if np.unique(md).size != md.size:
            raise NotImplementedError(
                "duplicate entries in measured_dofs are not supported by the HIP path")
        flags[md] |= PF_DOF_MEASURED
        meas_val[md] = mv.astype(np.float32)
        n_meas = int(md.size)
    return HostPlan(
        dim=dimension, n_nodes=n_nodes, n_elems=n_elems, n_dofs=n_dofs,
        conn=np.ascontiguousarray(elements, dtype=np.int32),
        egeo=np.ascontiguousarray(egeo, dtype=np.float32),
        ecent=np.ascontiguousarray(ecent, dtype=np.float32),
        adj_ptr=adj_ptr.astype(np.int32), adj=order.astype(np.int32),
        adj_other=np.ascontiguousarray(flat[order ^ 1], dtype=np.int32),     # position k^1 = the element's other end
        f_ext=loads.astype(np.float32), dof_flags=flags, meas_val=meas_val, n_meas=n_meas,
        free_dofs=free, fixed_dofs=fixed)


def chain_mesh(n_elems: int, h: float = 1.0):
    """Synthetic collinear 2-D truss of SURVEY.md §8(d): nodes (i*h, 0), elements (e, e+1),
    fixed = {0} U {all uy}, tip load 1, measurements ux_i = x_i, uy_i = 0 at every node >= 1."""
    n = int(n_elems)
    x = np.arange(n + 1, dtype=np.float64) * h
    nodes = np.stack([x, np.zeros(n + 1)], axis=1)
    e = np.arange(n, dtype=np.int64)
    elements = np.stack([e, e + 1], axis=1)
    loads = np.zeros(2 * (n + 1))
    loads[2 * n] = 1.0
    fixed = np.concatenate([[0], 2 * np.arange(n + 1) + 1])
    k = np.arange(1, n + 1)
    meas_dofs = np.stack([2 * k, 2 * k + 1], axis=1).reshape(-1)
    meas_vals = np.stack([x[1:], np.zeros(n)], axis=1).reshape(-1)
    return nodes, elements, loads, fixed, meas_vals, meas_dofs


def warren_mesh(n_panels: int, h: float = 1.0, height: float = 1.0):
    """Synthetic Warren girder (a genuinely 2-D truss, node degree 4): bottom chord nodes (i*h, 0),
    i = 0..n, top chord nodes ((i+0.5)*h, height), i = 0..n-1; elements per panel, in mesh order: bottom
    chord, rising diagonal, falling diagonal, top chord -> 4n-1 elements.  Pin at the first bottom node,
    roller (uy) at the last, unit downward load on every interior bottom node, measurements at every free
    node dof (a smooth synthetic sag plus a horizontal drift, non-zero at every dof so that Adam never
    normalises pure round-off; throughput runs only need the shape)."""
    n = int(n_panels)
    xb = np.arange(n + 1, dtype=np.float64) * h
    xt = (np.arange(n, dtype=np.float64) + 0.5) * h
    # interleave bottom and top nodes so that neighbouring nodes stay close in memory: b0 t0 b1 t1 ... bn
    nodes = np.zeros((2 * n + 1, 2))
    nodes[0::2, 0] = xb
    nodes[1::2, 0] = xt
    nodes[1::2, 1] = height
    b = lambda i: 2 * i          # node id of bottom node i
    t = lambda i: 2 * i + 1      # node id of top node i
    i = np.arange(n)
    per_panel = [np.stack([b(i), b(i + 1)], 1), np.stack([b(i), t(i)], 1), np.stack([t(i), b(i + 1)], 1)]
    top = np.stack([t(i[:-1]), t(i[:-1] + 1)], 1)
    el = np.empty((4 * n - 1, 2), dtype=np.int64)
    el[0:4 * (n - 1):4] = per_panel[0][:-1]
    el[1:4 * (n - 1):4] = per_panel[1][:-1]
    el[2:4 * (n - 1):4] = per_panel[2][:-1]
    el[3:4 * (n - 1):4] = top
    el[4 * (n - 1):] = np.stack([per_panel[0][-1], per_panel[1][-1], per_panel[2][-1]])
    loads = np.zeros(2 * (2 * n + 1))
    loads[2 * b(np.arange(1, n)) + 1] = -1.0
    fixed = np.array([0, 1, 2 * b(n) + 1])
    free = np.ones(2 * (2 * n + 1), dtype=bool)
    free[fixed] = False
    meas_dofs = np.flatnonzero(free)
    span = max(xb[-1], 1.0)
    sag = -1e-3 * np.sin(np.pi * nodes[:, 0] / span)
    vals = np.zeros(2 * (2 * n + 1))
    vals[0::2] = 2e-4 * (0.1 + nodes[:, 0] / span)     # non-zero everywhere: no dof with a round-off-sized gradient
    vals[1::2] = sag - 1e-4
    return nodes, el, loads, fixed, vals[meas_dofs], meas_dofs
