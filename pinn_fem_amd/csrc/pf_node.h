// pf_node.h — dL/du + Adam(u) of a node, shared by k_node_gradu (pf_mesh.hip) and by the MFMA32 forward launch that runs the
// displacement update of the PREVIOUS iteration beside its own element tasks (pf_net32.hip: k_net32_forward2, gu_nb).
//
// Reference: autograd of fem/nn_assembly.py:96-100,194-195 w.r.t. u + the data term (fem/solver.py:274-283),
// optimizer_u.step() (solver.py:292, torch/optim/adam.py single-tensor arithmetic), u[fixed] = 0 (solver.py:297-298).
// The arithmetic of one dof lives in ONE place (dof_grad_u, dof_adam_u: contraction off, every operation an IEEE single
// operation) so that the two callers agree bit for bit whatever surrounds them.
#pragma once
#include "pf_common.h"

struct GraduConsts { float step_size, bc2s, b1w, b2, b2w, eps, dcoef; };

// PP: pointer to a pf_problem in any address space (the fused forward launch reads its own kernel argument through the
// kernarg segment pointer, so that these values are loaded where a node task starts instead of living in scalar registers
// over the whole launch)
template <class PP>
__device__ __forceinline__ GraduConsts gradu_consts_from(PP p) {
  GraduConsts K;
  const pf_state* st = p->state;
  K.step_size = st->step_size_u;
  K.bc2s = st->bc2_sqrt;
  K.b1w = (float)(1.0 - p->beta1);
  K.b2 = (float)p->beta2;
  K.b2w = (float)(1.0 - p->beta2);
  K.eps = (float)p->eps;
  const float nm = p->n_meas_f;
  K.dcoef = nm > 0.f ? p->alpha_data / nm : 0.f;  // mean backward
  return K;
}
__device__ __forceinline__ GraduConsts gradu_consts(const pf_problem& P) { return gradu_consts_from(&P); }

// what a node task reads of the problem
struct NodeView {
  const int32_t *adj_ptr, *adj, *adj_other;
  const uint8_t* dof_flags;
  const float *meas_val, *g_f;
  float *u, *m_u, *v_u, *grad_u;
  int n_nodes, use_data, fe_mode;
};
template <class PP>
__device__ __forceinline__ NodeView node_view_from(PP p) {
  NodeView V;
  V.adj_ptr = p->mesh.adj_ptr;
  V.adj = p->mesh.adj;
  V.adj_other = p->adj_other;
  V.dof_flags = p->mesh.dof_flags;
  V.meas_val = p->mesh.meas_val;
  V.g_f = p->g_f;
  V.u = p->u;
  V.m_u = p->m_u;
  V.v_u = p->v_u;
  V.grad_u = p->grad_u;
  V.n_nodes = p->mesh.n_nodes;
  V.use_data = p->use_data;
  V.fe_mode = p->fe_mode;
  return V;
}

// dL/du of one dof: the gathered K^T g_f entry + d(alpha_d * mean d^2)/du where the dof is measured
__device__ __forceinline__ float dof_grad_u(const GraduConsts& K, bool measured, float g, float meas, float uo) {
  PF_NO_CONTRACT
  float gu = g;
  if (measured) {
    const float d = meas - uo;
    gu += -(K.dcoef * (2.f * d));
  }
  return gu;
}

// torch.optim.Adam single-tensor arithmetic (torch/optim/adam.py) on one free dof
__device__ __forceinline__ void dof_adam_u(const GraduConsts& K, float gu, float& uo, float& m, float& v) {
  PF_NO_CONTRACT
  m = m + K.b1w * (gu - m);                       // lerp_
  v = v * K.b2;                                   // mul_
  v = v + (K.b2w * gu) * gu;                      // addcmul_
  const float denom = sqrtf(v) / K.bc2s + K.eps;
  uo = uo + (-K.step_size) * (m / denom);         // addcdiv_
}

// ---- M x 64 nodes per wave, every level of the gather in flight for all of them together -----------------------------
// The stand-alone kernel hides the three dependent round trips of a node (adj_ptr -> adjacency -> records and neighbour
// values) behind 32 waves per CU; inside the forward launch there are 16, most of them busy with element tasks, so a node
// task carries M nodes per lane and issues each level's loads for all of them before it waits.  Straight-line code with
// selects: a branch per node would let the compiler sink the loads into it and serialise the round trips.
// elem_k: the stiffness records to read (the iteration graph: the half the PREVIOUS forward launch wrote).
// Needs pf_problem.adj_other; single-GPU meshes only (no ghost elements, no shared dofs).  Accumulation in ascending
// element id like gather_kv (pf_mesh.hip), same ke_rows_times: same bits.
template <int DIM, int M>
__device__ __forceinline__ float node_gradu_task(const NodeView& P, const float* __restrict__ elem_k, const GraduConsts& K,
                                                 int node0, int lane) {
  const int nn = P.n_nodes;
  int node[M], b[M], e_[M];
  bool ok[M];
  float vs[M][2], acc[M][2], uo[M][2], meas[M][2];
  unsigned fl[M][2];
  const float* __restrict__ mvals = P.use_data ? P.meas_val : P.u;     // (a select, not a branch: never read without use_data)
  // level 1: CSR row, the node's own values
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int n_ = node0 + m * 64 + lane;
    ok[m] = n_ < nn;
    node[m] = ok[m] ? n_ : nn - 1;
    b[m] = P.adj_ptr[node[m]];
    e_[m] = P.adj_ptr[node[m] + 1];
    load_vec<DIM>(P.g_f, node[m], vs[m]);
    load_vec<DIM>(P.u, node[m], uo[m]);
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      fl[m][c] = P.dof_flags[node[m] * DIM + c];
      meas[m][c] = mvals[node[m] * DIM + c];
      acc[m][c] = 0.f;
    }
  }
  int rounds = 0;
#pragma unroll
  for (int m = 0; m < M; ++m) rounds = max(rounds, (e_[m] - b[m] + 1) >> 1);
  // the Adam moments travel with level 1 too (a fixed dof's are dead state: loaded, never stored)
  float mo[M][2], vo[M][2];
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      mo[m][c] = P.m_u[node[m] * DIM + c];
      vo[m][c] = P.v_u[node[m] * DIM + c];
    }
  for (int r = 0; r < rounds; ++r) {
    int code0[M], code1[M], oth0[M], oth1[M];
    bool has[M], two[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int idx = b[m] + 2 * r;
      has[m] = idx < e_[m];
      two[m] = idx + 1 < e_[m];
      const int i0 = has[m] ? idx : 0, i1 = two[m] ? idx + 1 : i0;
      code0[m] = P.adj[i0];
      oth0[m] = P.adj_other[i0];
      code1[m] = P.adj[i1];
      oth1[m] = P.adj_other[i1];
    }
    ElemK k0[M], k1[M];
    float w0[M][2], w1[M][2];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      k0[m] = load_k_record<DIM>(elem_k, code0[m] >> 1);
      k1[m] = load_k_record<DIM>(elem_k, code1[m] >> 1);
      load_vec<DIM>(P.g_f, oth0[m], w0[m]);
      load_vec<DIM>(P.g_f, oth1[m], w1[m]);
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float fe[2];
      const int end0 = code0[m] & 1, end1 = code1[m] & 1;
      ke_rows_times<DIM>(k0[m], end0, end0 ? w0[m] : vs[m], end0 ? vs[m] : w0[m], fe, P.fe_mode);
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        const float t = acc[m][c] + fe[c];
        acc[m][c] = has[m] ? t : acc[m][c];
      }
      ke_rows_times<DIM>(k1[m], end1, end1 ? w1[m] : vs[m], end1 ? vs[m] : w1[m], fe, P.fe_mode);
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        const float t = acc[m][c] + fe[c];
        acc[m][c] = two[m] ? t : acc[m][c];
      }
    }
    // the own values are only USED behind this loop: an (empty) use here keeps their loads in front of it — the compiler
    // may otherwise sink them to their first use, one more round trip per task
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int c = 0; c < DIM; ++c)
        asm volatile("" : "+v"(uo[m][c]), "+v"(meas[m][c]), "+v"(mo[m][c]), "+v"(vo[m][c]));
  }
  float sum_u2 = 0.f;
#pragma unroll
  for (int m = 0; m < M; ++m) {
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      const int dof = node[m] * DIM + c;
      const unsigned f = fl[m][c];
      const float gu = dof_grad_u(K, P.use_data && (f & PF_DOF_MEASURED), acc[m][c], meas[m][c], uo[m][c]);
      float un = uo[m][c], mn = mo[m][c], vn = vo[m][c];
      dof_adam_u(K, gu, un, mn, vn);
      const bool fixed = (f & PF_DOF_FIXED) != 0;
      if (ok[m]) {
        if (P.grad_u) P.grad_u[dof] = gu;
        if (fixed) {
          if (uo[m][c] != 0.f) P.u[dof] = 0.f;      // solver.py:297-298
        } else {
          P.m_u[dof] = mn;
          P.v_u[dof] = vn;
          P.u[dof] = un;
        }
      }
      { PF_NO_CONTRACT
        if (ok[m] && !fixed) sum_u2 += un * un; }
    }
  }
  return sum_u2;
}

// ---- the same gather for M nodes per THREAD (node kernels): (K(theta) v)[node] over the incident elements of each node,
// every level's loads of all M nodes issued before the first wait.  The stand-alone node kernels walk two nodes per
// thread one after the other (grid-stride): six dependent round trips per thread; taken together they are three.
// Same accumulation order per node as gather_kv (pf_mesh.hip), same ke_rows_times: same bits.  Needs adj_other and the
// stiffness records.
template <int DIM, int M>
__device__ __forceinline__ void gather_kv_multi(const pf_problem& P, const float* __restrict__ elem_k,
                                                const float* __restrict__ v, const int (&node)[M], float (&acc)[M][2]) {
  const pf_mesh& Ms = P.mesh;
  int b[M], e_[M];
  float vs[M][2];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    b[m] = Ms.adj_ptr[node[m]];
    e_[m] = Ms.adj_ptr[node[m] + 1];
    load_vec<DIM>(v, node[m], vs[m]);
    acc[m][0] = 0.f;
    acc[m][1] = 0.f;
  }
  int rounds = 0;
#pragma unroll
  for (int m = 0; m < M; ++m) rounds = max(rounds, (e_[m] - b[m] + 1) >> 1);
  for (int r = 0; r < rounds; ++r) {
    int code0[M], code1[M], oth0[M], oth1[M];
    bool has[M], two[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int idx = b[m] + 2 * r;
      has[m] = idx < e_[m];
      two[m] = idx + 1 < e_[m];
      const int i0 = has[m] ? idx : 0, i1 = two[m] ? idx + 1 : i0;
      code0[m] = Ms.adj[i0];
      oth0[m] = P.adj_other[i0];
      code1[m] = Ms.adj[i1];
      oth1[m] = P.adj_other[i1];
    }
    ElemK k0[M], k1[M];
    float w0[M][2], w1[M][2];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      k0[m] = load_k_record<DIM>(elem_k, code0[m] >> 1);
      k1[m] = load_k_record<DIM>(elem_k, code1[m] >> 1);
      load_vec<DIM>(v, oth0[m], w0[m]);
      load_vec<DIM>(v, oth1[m], w1[m]);
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float fe[2];
      const int end0 = code0[m] & 1, end1 = code1[m] & 1;
      ke_rows_times<DIM>(k0[m], end0, end0 ? w0[m] : vs[m], end0 ? vs[m] : w0[m], fe, P.fe_mode);
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        const float t = acc[m][c] + fe[c];
        acc[m][c] = has[m] ? t : acc[m][c];
      }
      ke_rows_times<DIM>(k1[m], end1, end1 ? w1[m] : vs[m], end1 ? vs[m] : w1[m], fe, P.fe_mode);
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        const float t = acc[m][c] + fe[c];
        acc[m][c] = two[m] ? t : acc[m][c];
      }
    }
  }
}
