// pf_mesh.hip — node-parallel and element-parallel FEM kernels, reductions, optimiser, monitors.
//
// Reference lines replaced (FEM/python/...):
//   k_node_residual   fem/nn_assembly.py:64-100,226-227 (fe = (s*pattern)@u_e, f_int[g] += fe[a])
//                     fem/solver.py:267-283 (residual, 0.5*sum r^2, mean d^2)
//   k_elem_adjoint    autograd of nn_assembly.py:74,96-100 w.r.t. the element stiffness
//   k_node_gradu      autograd of nn_assembly.py:96-100,194-195 w.r.t. u + data term
//                     fem/solver.py:292 (Adam on u) + :297-298 (u[fixed]=0) + :304 (||u_free||)
//   k_theta_reduce    sum over elements of parameter gradients; fem/solver.py:293-294 (Adam on theta)
//   k_finalize        fem/solver.py:304-355 (monitors, history entry, stop test)
//
// The scatter-add of the reference (20 indexed `+=` per element into a dense K it never reads) is
// replaced by an owner-computes GATHER: one thread per node walks the node's incident elements in
// ascending element id, which reproduces the reference's accumulation order exactly, needs no
// atomics and is bitwise reproducible.  K is never formed: K u and K^T g are applied matrix-free.
#include <stdlib.h>
#include "pf_common.h"
#include "pf_node.h"
#include "pf_net32.h"

// (PF_NO_CONTRACT, the parameter update pf_theta_update and the theta-norm helpers live in pf_common.h: the MFMA32
// forward kernels run the same update in their prologue)

namespace {

// (K(theta) v)[node] by gather over incident elements, ascending element id
// own_only (multi-GPU): skip the ghost elements (local ids outside [own_lo, own_hi)): their contributions to a
// gradient belong to the rank that owns them
template <int DIM>
__device__ __forceinline__ void gather_kv(const pf_problem& P, const float* __restrict__ v, int node,
                                          float* acc, bool own_only = false) {
  const pf_mesh& M = P.mesh;
#pragma unroll
  for (int c = 0; c < DIM; ++c) acc[c] = 0.f;
  const int b = M.adj_ptr[node], e_ = M.adj_ptr[node + 1];
  if (P.adj_other) {
    // Three dependent round trips per node whatever its degree (<= 2 per round): adj_ptr -> (adj, adj_other) of two
    // incidences -> (geometry, stiffness, neighbour values) of both -> arithmetic in ascending element order (the
    // reference's accumulation order, as below).  The node's own entries need no indirection at all.
    float vs[2];
    load_vec<DIM>(v, node, vs);
    for (int idx = b; idx < e_; idx += 2) {
      const bool two = idx + 1 < e_;
      const int code0 = M.adj[idx], oth0 = P.adj_other[idx];
      const int code1 = two ? M.adj[idx + 1] : code0, oth1 = two ? P.adj_other[idx + 1] : oth0;
      const int e0 = code0 >> 1, e1 = code1 >> 1;
      const ElemK k0 = load_k<DIM>(P, e0), k1 = load_k<DIM>(P, e1);
      float vo0[2], vo1[2], fe[2];
      load_vec<DIM>(v, oth0, vo0);
      load_vec<DIM>(v, oth1, vo1);
      if (!(own_only && (e0 < P.own_lo || e0 >= P.own_hi))) {
        const int end = code0 & 1;
        ke_rows_times<DIM>(k0, end, end ? vo0 : vs, end ? vs : vo0, fe, P.fe_mode);
#pragma unroll
        for (int c = 0; c < DIM; ++c) acc[c] += fe[c];
      }
      if (two && !(own_only && (e1 < P.own_lo || e1 >= P.own_hi))) {
        const int end = code1 & 1;
        ke_rows_times<DIM>(k1, end, end ? vo1 : vs, end ? vs : vo1, fe, P.fe_mode);
#pragma unroll
        for (int c = 0; c < DIM; ++c) acc[c] += fe[c];
      }
    }
    return;
  }
  for (int idx = b; idx < e_; ++idx) {
    const int code = M.adj[idx];
    const int e = code >> 1, end = code & 1;
    if (own_only && (e < P.own_lo || e >= P.own_hi)) continue;
    const int2 nn = reinterpret_cast<const int2*>(M.conn)[e];
    const ElemK k = load_k<DIM>(P, e);
    float vi[2], vj[2], fe[2];
    load_vec<DIM>(v, nn.x, vi);
    load_vec<DIM>(v, nn.y, vj);
    ke_rows_times<DIM>(k, end, vi, vj, fe, P.fe_mode);
#pragma unroll
    for (int c = 0; c < DIM; ++c) acc[c] += fe[c];
  }
}

// ---- residual, losses, dL/df_int --------------------------------------------------------------
// Multi-GPU: the local mesh carries the ghost elements of every shared node, so f_int is complete on every node of
// an own element; a dof flagged PF_DOF_GHOST (owned by another rank, or a pure ghost node whose f_int is partial and
// never used) is left out of the sums.
__device__ void finalize_body(const pf_problem& P, int nb_node, int mode, int with_theta, const float* __restrict__ ext_rd,
                              const float* __restrict__ ext_u2, int u2_lag, int half, float* new_theta, int tn_ready);

// fin_prev (1, or 2 = with tn_ready): block 0 does the bookkeeping of the PREVIOUS iteration (finalize_body: monitors, history row, stop test,
// next Adam scalars) from the other half of the residual sums while the remaining blocks work on this iteration's
// nodes — the single-block, latency-bound finalize then costs nothing and needs no branch of its own in the graph.
#ifndef PF_RESIDUAL_NODES
#define PF_RESIDUAL_NODES 2
#endif
template <int DIM>
__global__ __launch_bounds__(PF_NODE_THREADS) __attribute__((amdgpu_num_sgpr(72))) void k_node_residual(pf_problem P, float* f_int_out,
                                                                    int compute_loss, int fin_prev) {
  if (fin_prev && blockIdx.x == 0) {
    finalize_body(P, (int)gridDim.x - 1, 0, 0, nullptr, nullptr, 0, P.part_half ^ 1, nullptr, fin_prev == 2);
    return;
  }
  __shared__ float red[16];
  // the stop flag, ONE read per block: block 0 of this very launch (the previous iteration's bookkeeping) may be raising
  // it while the other blocks start, and waves of a block that disagreed would part ways before the barriers below
  __shared__ int s_done;
  if (threadIdx.x == 0) s_done = P.state->done;
  __syncthreads();
  if (s_done) return;
  const pf_mesh& M = P.mesh;
  const int fp = fin_prev ? 1 : 0;     // (fin_prev is 1 or 2: ONE extra block either way)
  const int bid = (int)blockIdx.x - fp, nblk = (int)gridDim.x - fp;
  float sum_r2 = 0.f, sum_d2 = 0.f;
  // one node: residual, dL/df_int, the two loss sums (f = this node's internal force)
  auto finish = [&](int node, const float* f, const unsigned* fl, const float* fx, const float* un, const float* mv) {
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      const int dof = node * DIM + c;
      const bool mine = !(fl[c] & PF_DOF_GHOST);
      if (f_int_out) f_int_out[dof] = f[c];
      if (!compute_loss) continue;
      float gf = 0.f;
      if (!(fl[c] & PF_DOF_FIXED)) {
        const float r = f[c] - P.lam * fx[c];           // solver.py:267-269
        if (mine) sum_r2 += r * r;
        gf = P.alpha_physics * r;                       // d(alpha_p * 0.5*sum r^2)/dr
      }
      P.g_f[dof] = gf;
      if (mine && P.use_data && (fl[c] & PF_DOF_MEASURED)) {
        const float d = mv[c] - un[c];                  // solver.py:274
        sum_d2 += d * d;
      }
    }
  };
  const int stride = nblk * (int)blockDim.x;
  int node = bid * (int)blockDim.x + (int)threadIdx.x;
  if (P.elem_k && P.adj_other && !f_int_out) {
    // PF_RESIDUAL_NODES nodes of the thread's grid-stride walk at a time, every level of their gathers in flight together
    // (gather_kv_multi, pf_node.h): three dependent round trips per group instead of three per node.  Same per-node
    // arithmetic and the same order of the thread's loss sums (the nodes in walk order) as the one-by-one walk below.
    const float* __restrict__ mvals = P.use_data ? M.meas_val : P.u;
    constexpr int NW = PF_RESIDUAL_NODES;       // nodes of the walk taken together
    for (; node < M.n_nodes; node += NW * stride) {
      int nd[NW];
      bool ok[NW];
      unsigned fl[NW][2];
      float fx[NW][2], un[NW][2], mv[NW][2], f[NW][2];
#pragma unroll
      for (int m = 0; m < NW; ++m) {
        ok[m] = node + m * stride < M.n_nodes;
        nd[m] = ok[m] ? node + m * stride : node;
        load_vec<DIM>(M.f_ext, nd[m], fx[m]);
        load_vec<DIM>(P.u, nd[m], un[m]);
        load_vec<DIM>(mvals, nd[m], mv[m]);
#pragma unroll
        for (int c = 0; c < DIM; ++c) fl[m][c] = M.dof_flags[nd[m] * DIM + c];
      }
      gather_kv_multi<DIM, NW>(P, P.elem_k, P.u, nd, f);
#pragma unroll
      for (int m = 0; m < NW; ++m)
        if (ok[m]) finish(nd[m], f[m], fl[m], fx[m], un[m], mv[m]);
    }
  } else {
    for (; node < M.n_nodes; node += stride) {
      float f[2], fx[2] = {0.f, 0.f}, un[2] = {0.f, 0.f}, mv[2] = {0.f, 0.f};
      unsigned fl[2] = {0u, 0u};
      gather_kv<DIM>(P, P.u, node, f);
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        const int dof = node * DIM + c;
        fl[c] = M.dof_flags[dof];
        fx[c] = M.f_ext[dof];
        if (P.use_data && (fl[c] & PF_DOF_MEASURED)) { mv[c] = M.meas_val[dof]; un[c] = P.u[dof]; }
      }
      finish(node, f, fl, fx, un, mv);
    }
  }
  if (!compute_loss) return;
  const float t0 = pf_block_sum(sum_r2, red);
  const float t1 = pf_block_sum(sum_d2, red);
  if (threadIdx.x == 0) {
    P.partials[PF_PART_R2H(P.part_half) + bid] = t0;
    P.partials[PF_PART_D2H(P.part_half) + bid] = t1;
  }
}

// ---- dL/d(E*A) per element ---------------------------------------------------------------------
template <int DIM>
__global__ __launch_bounds__(256) void k_elem_adjoint(pf_problem P) {
  if (P.state->done) return;
  const pf_mesh& M = P.mesh;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < M.n_elems; e += gridDim.x * blockDim.x) {
    P.g_ea[e] = pf_elem_gea<DIM>(P, e);
  }
}

// ---- dL/du (+ Adam on u, BC clamp, ||u_free||^2) ------------------------------------------------
// skip_shared (multi-GPU): shared dofs are left alone; their gradient is completed by the second
// all-reduce and pf_shard_update_shared steps them.
// u_out (iteration graph, FUSE_ADAM only): the updated displacements go to this vector instead of P.u, EVERY dof of it
// (the other kernels of the iteration still read P.u); out_alt = 1 when u_out is pf_problem.u_alt.
template <int DIM, bool FUSE_ADAM>
__global__ __launch_bounds__(PF_NODE_THREADS) __attribute__((amdgpu_num_sgpr(72))) void k_node_gradu(pf_problem P, int skip_shared,
                                                                                                   float* __restrict__ u_out, int out_alt) {
  PF_NO_CONTRACT
  if (P.state->done) return;
  if (FUSE_ADAM && u_out && blockIdx.x == 0 && threadIdx.x == 0) P.state->u_half = out_alt;
  __shared__ float red[16];
  const pf_mesh& M = P.mesh;
  const GraduConsts K = gradu_consts(P);
  float sum_u2 = 0.f;
  // one node: data term, gradient store, Adam step / clamp, norm of u_free (g = this node's K^T g_f entry)
  auto finish = [&](int node, const float* g) {
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      const int dof = node * DIM + c;
      const unsigned fl = M.dof_flags[dof];
      if (skip_shared && (fl & PF_DOF_SHARED)) continue;
      float uo = P.u[dof];
      const bool measured = P.use_data && (fl & PF_DOF_MEASURED);
      const float gu = dof_grad_u(K, measured, g[c], measured ? M.meas_val[dof] : 0.f, uo);   // (pf_node.h)
      if (P.grad_u) P.grad_u[dof] = gu;
      if (FUSE_ADAM) {
        if (fl & PF_DOF_FIXED) {
          // solver.py:297-298: u[fixed] = 0 after the step.  The Adam moments of a fixed dof only ever
          // feed that dof's own (discarded) update, so they are dead state: not loaded, not stored.
          if (u_out) u_out[dof] = 0.f;
          else if (uo != 0.f) P.u[dof] = 0.f;
          continue;
        }
        float m = P.m_u[dof], v = P.v_u[dof];
        dof_adam_u(K, gu, uo, m, v);                    // torch.optim.Adam single-tensor arithmetic
        if (!(fl & PF_DOF_GHOST)) sum_u2 += uo * uo;
        P.m_u[dof] = m;
        P.v_u[dof] = v;
        (u_out ? u_out : P.u)[dof] = uo;
      }
    }
  };
  const int stride = (int)(gridDim.x * blockDim.x);
  int node = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (P.elem_k && P.adj_other) {
    // two nodes of the thread's walk at a time, both gathers level by level together (gather_kv_multi, pf_node.h; K
    // symmetric: K^T g_f by the same gather as the residual's); same arithmetic and order as the walk below
    for (; node < M.n_nodes; node += 2 * stride) {
      const bool two = node + stride < M.n_nodes;
      const int nd[2] = {node, two ? node + stride : node};
      float g2[2][2];
      gather_kv_multi<DIM, 2>(P, P.elem_k, P.g_f, nd, g2);
      finish(nd[0], g2[0]);
      if (two) finish(nd[1], g2[1]);
    }
  } else {
    for (; node < M.n_nodes; node += stride) {
      float g[2];
      gather_kv<DIM>(P, P.g_f, node, g);  // K symmetric: K^T g_f by the same gather
      finish(node, g);
    }
  }
  if (FUSE_ADAM) {
    const float t = pf_block_sum(sum_u2, red);
    if (threadIdx.x == 0) P.partials[PF_PART_U2 + blockIdx.x] = t;
  }
}

// ---- parameter gradient: sum block partials (fixed order) -> torch layout (+ Adam on theta) --------
// stage 1: [nb_rows][pad_total] -> [PF_RG][pad_total]; grid (ceil(pad_total/64), PF_RG), 256 threads =
// 64 columns x 4 row lanes, every thread a short strided row sum, then a fixed-order LDS combine.
__global__ __launch_bounds__(256) void k_theta_stage1(pf_problem P, int nb_rows) {
  const int done = P.state->done;   // checked before the store: the row loads are issued beside this load, not behind it
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  const int rpg = (nb_rows + PF_RG - 1) / PF_RG;
  const int r0 = blockIdx.y * rpg, r1 = min(r0 + rpg, nb_rows);
  float a = 0.f;
  if (col < P.pad_total) {
    const float* __restrict__ rows = P.partials + PF_PART_WG + col;
    for (int r = r0 + rl; r < r1; r += 4) a += rows[(size_t)r * P.pad_total];
  }
  red[rl][cl] = a;
  __syncthreads();
  if (done) return;
  if (rl == 0 && col < P.pad_total) {
    const float t = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    P.partials[PF_PART_WG + (size_t)P.n_part_blocks * P.pad_total + (size_t)blockIdx.y * P.pad_total + col] = t;
  }
}

// MFMA32 engine: rebuild the split-f16 operand images of the enabled nets from `th` (the flat active
// parameters: LDS copy of the values just written, or global theta).  Every thread of the block; no barriers.
__device__ __forceinline__ void pack_net_ops(const pf_problem& P, const float* th) {
  if (P.wg_mode != PF_WG_MFMA32 || !P.net_op) return;
  for (int k = 0; k < 2; ++k) {
    if (!P.net[k].enabled) continue;
    pf_n32_pack(P.net[k], th + P.net[k].theta_off, reinterpret_cast<unsigned char*>(P.net_op + P.op_off[k]),
                P.mlp_dtype);
  }
}

// MFMA32 engine (iteration graph): besides the update and the operand images, the block's otherwise idle waves compute
// the iteration's theta-norm monitor from the LDS copy of the new parameters (state->theta_norm), so that the
// bookkeeping (finalize_body with tn_ready) has no dependent chain of global loads left.
// The stand-alone update reads the state from the half the iteration graph left it in (state->theta_half; 0 outside
// the graph) and always writes half 0; after a stop it only brings a state left in half 1 home.
__device__ __forceinline__ void theta_update_standalone(const pf_problem& P, int fuse_adam, float* new_theta, int done) {
  const int half = (fuse_adam && P.theta_alt) ? P.state->theta_half : 0;
  if (done) {
    if (half != 0)
      for (int w = 0; w < 3; ++w) {
        const float* src = pf_theta_half_ptr(P, 1, w);
        float* dst = pf_theta_half_ptr(P, 0, w);
        for (int q = threadIdx.x; q < P.n_theta_active; q += blockDim.x) dst[q] = src[q];
      }
  } else {
    pf_theta_update(P, fuse_adam, new_theta, 0, half, 0);
  }
  if (half != 0) {
    __syncthreads();
    if (threadIdx.x == 0) P.state->theta_half = 0;
  }
}

__global__ __launch_bounds__(1024) void k_theta_stage2(pf_problem P, int fuse_adam) {
  const int done = P.state->done;        // acted on after the loads of the update have been issued
  extern __shared__ float new_theta[];   // n_theta_active floats (MFMA32 engine only)
  __shared__ float tnorm[PF_MAX_TENSORS];
  const bool ops = P.wg_mode == PF_WG_MFMA32 && fuse_adam;
  theta_update_standalone(P, fuse_adam, ops ? new_theta : nullptr, done);
  if (done) return;
  if (ops) {
    __syncthreads();
    pack_net_ops(P, new_theta);           // operand entries: waves 0..3, scaling bound: the last wave
    const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (nw >= 8) {
      if (wv >= 4 && wv < nw - 1) tensor_norms(P, new_theta, wv - 4, nw - 5, tnorm);
    } else {
      tensor_norms(P, new_theta, wv, nw, tnorm);
    }
    __syncthreads();
    if (threadIdx.x == 0) P.state->theta_norm = (float)tensor_norm_total(P, tnorm);
  }
}

__global__ __launch_bounds__(256) void k_pack_theta(pf_problem P) {
  for (int q = threadIdx.x; q < P.n_theta_active; q += blockDim.x) P.theta_pad[P.pad_index[q]] = P.theta[q];
  pack_net_ops(P, P.theta);
}

// ---- parameter update + monitors / history / stop test / next Adam scalars ------------------------
// mode 0: full iteration bookkeeping incl. theta stage 2 + Adam (solver.py:293-294, 304-355);
// mode 1: losses + gradient reduction only (autograd binding).
// ext_rd / ext_u2 (multi-GPU): globally reduced [sum r^2, sum d^2] and [sum u_free^2] to use instead of
// this rank's block partials (which already belong to the next iteration when this runs).
// u2_lag (multi-GPU): ext_u2 is the sum of the PREVIOUS iteration (it travelled on this iteration's all-reduce):
// it completes the previous history row; this iteration's u_norm is filled in by the next call (or by k_shard_flush).
// `half`: which half of the residual's partial sums to read.  One block of PF_FIN_THREADS threads or more.
#define PF_FIN_THREADS PF_NODE_THREADS
// tn_ready: state->theta_norm already holds this iteration's value (k_theta_stage2).
__device__ void finalize_body(const pf_problem& P, int nb_node, int mode, int with_theta, const float* __restrict__ ext_rd,
                              const float* __restrict__ ext_u2, int u2_lag, int half, float* new_theta, int tn_ready) {
  // Latency-bound (one block, a handful of dependent global round trips — and inside k_node_residual every one of them
  // competes with 2047 memory-bound blocks): all loads are issued up front, the stop flag is only acted on at the end,
  // and the two double-precision pow() of the next Adam scalars run on a second wave beside the sums.
  pf_state* S = P.state;
  const int done0 = S->done;
  __shared__ double dred[16];
  __shared__ double adam_bc[2];
  __shared__ float tnorm[PF_MAX_TENSORS];
  if (with_theta) {
    if (done0) return;
    pf_theta_update(P, mode == 0, new_theta);
    if (mode == 0) {
      __syncthreads();
      pack_net_ops(P, new_theta);
    }
  }
  const int it = S->iter;  // 0-based index of the iteration just completed
  const float tn_state = S->theta_norm;
  if (mode == 0 && threadIdx.x == 64) {
    // Adam scalars of step t = it+2, in double like torch's Python floats
    const double t = (double)(it + 2);
    adam_bc[0] = 1.0 - pow(P.beta1, t);
    adam_bc[1] = 1.0 - pow(P.beta2, t);
  }
  // fixed summation order whatever the block size (256 inside k_node_residual, 1024 stand-alone with theta stage 2):
  // PF_FIN_THREADS strided partial sums, then the waves in order (further waves add exact zeros)
  double a = 0.0, b = 0.0, c = 0.0;
  if (threadIdx.x < PF_FIN_THREADS) {
#pragma unroll 4
    for (int i = threadIdx.x; i < nb_node; i += PF_FIN_THREADS) {
      a += (double)P.partials[PF_PART_R2H(half) + i];
      b += (double)P.partials[PF_PART_D2H(half) + i];
      if (mode == 0) c += (double)P.partials[PF_PART_U2 + i];
    }
  }
  float sum_r2 = (float)pf_block_sum_d(a, dred);
  float sum_d2 = (float)pf_block_sum_d(b, dred);
  float sum_u2 = (float)pf_block_sum_d(c, dred);   // (also orders new_theta writes before reads)
  if (ext_rd) {
    sum_r2 = ext_rd[0];
    sum_d2 = ext_rd[1];
    sum_u2 = ext_u2[0];
  }
  const float loss_p = 0.5f * sum_r2;                                   // solver.py:270
  float loss_d = 0.f, loss;
  if (P.use_data) {
    loss_d = sum_d2 / P.n_meas_f;                                       // torch.mean :275
    loss = P.alpha_physics * loss_p + P.alpha_data * loss_d;            // :277-279
  } else {
    loss = P.alpha_physics * loss_p;                                    // :283
  }
  const float rn = sqrtf(sum_r2);                                       // torch.norm :306
  double tn = 0.0;
  if (mode == 0 && P.n_tensors > 0) {
    if (tn_ready) {
      tn = (double)tn_state;
    } else {
      tensor_norms(P, new_theta, threadIdx.x >> 6, blockDim.x >> 6, tnorm);   // new_theta: LDS copy of the active parameters, or null
      __syncthreads();
      if (threadIdx.x == 0) tn = tensor_norm_total(P, tnorm);
    }
  }
  if (threadIdx.x != 0 || done0) return;   // (after the last barrier)
  S->loss_total = loss;
  S->loss_physics = loss_p;
  S->loss_data = loss_d;
  S->residual_norm = rn;
  if (mode != 0) return;
  const float un = sqrtf(sum_u2);                                       // :304
  S->u_norm = un;
  S->theta_norm = (float)tn;
  if (P.hist && it < P.max_iter) {
    float* h = P.hist + (size_t)it * PF_HIST_COLS;
    h[0] = loss; h[1] = loss_p; h[2] = loss_d; h[3] = u2_lag ? 0.f : un; h[4] = rn; h[5] = (float)tn;
    if (u2_lag && it > 0) h[3 - PF_HIST_COLS] = un;                     // the previous row's u_norm
  }
  S->iter = it + 1;
  if (it > 10) {                                                         // :341-355
    if ((double)rn < P.tol || (double)loss < P.tol) {  // NaN compares false, like the reference
      S->converged = 1;
      S->done = 1;
    }
  }
  if (it + 1 >= P.max_iter) S->done = 1;
  const double bc1 = adam_bc[0], bc2 = adam_bc[1];
  S->step_size_u = (float)((double)P.lr_u / bc1);
  S->step_size_t = (float)((double)P.lr_t / bc1);
  S->bc2_sqrt = (float)sqrt(bc2);
}

__global__ __launch_bounds__(1024) void k_finalize(pf_problem P, int nb_node, int mode, int with_theta,
                                                             const float* __restrict__ ext_rd,
                                                             const float* __restrict__ ext_u2, int u2_lag, int tn_ready) {
  extern __shared__ float new_theta[];  // n_theta_active floats
  finalize_body(P, nb_node, mode, with_theta, ext_rd, ext_u2, u2_lag, P.part_half, with_theta ? new_theta : nullptr, tn_ready);
}

// ---- multi-GPU shard interface -------------------------------------------------------------------
// ---- tail of the rank-local part of a sharded iteration: theta stage 2 (no Adam) into grad_theta (inside buf),
// this rank's share of grad_u on the interface dofs (OWN elements only; the data term by the owner), local loss sums:
//   buf = [iface grad_u (n_iface) | grad_theta (n_theta_active) | sum r^2, sum d^2, sum u_free^2 of the PREVIOUS
//          iteration (u2_local, written by k_shard_update)]                              => the iteration's ONE all-reduce
template <int DIM>
__global__ __launch_bounds__(1024) void k_shard_pack(pf_problem P, int nb_node, float* __restrict__ buf2,
                                                     const float* __restrict__ u2_local) {
  if (P.state->done) {
    // after the stop the remaining iterations of a chunk still all-reduce buf: zero it, or the stale sums would be
    // multiplied by the world size every iteration and run to inf (harmless only while every consumer returns early)
    for (int k = threadIdx.x; k < P.n_iface + P.n_theta_active + 3; k += blockDim.x) buf2[k] = 0.f;
    return;
  }
  __shared__ double dred[16];
  const pf_mesh& M = P.mesh;
  if (P.n_theta_active > 0) pf_theta_update(P, 0, nullptr);
  for (int k = threadIdx.x; k < P.n_iface; k += blockDim.x) buf2[k] = 0.f;
  __syncthreads();
  const float dcoef = P.n_meas_f > 0.f ? P.alpha_data / P.n_meas_f : 0.f;
  for (int k = threadIdx.x; k < P.n_shared; k += blockDim.x) {
    const int dof = P.shared_dofs[k], node = dof / DIM, c = dof % DIM;
    float g[2];
    gather_kv<DIM>(P, P.g_f, node, g, true);
    float gu = g[c];
    if (P.use_data && (M.dof_flags[dof] & PF_DOF_MEASURED)) {   // only the owner carries the flag
      const float d = M.meas_val[dof] - P.u[dof];
      gu += -(dcoef * (2.f * d));
    }
    buf2[P.shared_slot[k]] = gu;
  }
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nb_node; i += blockDim.x) {
    a += (double)P.partials[PF_PART_R2H(P.part_half) + i];
    b += (double)P.partials[PF_PART_D2H(P.part_half) + i];
  }
  const double ta = pf_block_sum_d(a, dred), tb = pf_block_sum_d(b, dred);
  if (threadIdx.x == 0) {
    float* tail = buf2 + P.n_iface + P.n_theta_active;
    tail[0] = (float)ta; tail[1] = (float)tb; tail[2] = u2_local[0];
  }
}

// ---- after the all-reduce: Adam(theta) from the reduced gradient, Adam(u) of the interface dofs (shared and ghost
// nodes: every rank that holds a copy applies the same reduced gradient, so the copies stay bit-identical), and this
// rank's sum u_free^2 over the dofs it owns -> u2_local[0] (rides on the NEXT iteration's all-reduce) -----------
__global__ __launch_bounds__(1024) void k_shard_update(pf_problem P, int nb_node, const float* __restrict__ buf2,
                                                       float* __restrict__ sums3) {
  PF_NO_CONTRACT
  if (P.state->done) return;   // sums3 keeps the previous (final) value: finalize ignores it once done
  __shared__ double dred[16];
  extern __shared__ float new_theta_dyn[];     // n_theta_active floats with the MFMA32 engine, else none
  float* new_theta = P.wg_mode == PF_WG_MFMA32 ? new_theta_dyn : nullptr;
  const pf_mesh& M = P.mesh;
  const float bc2s = P.state->bc2_sqrt;
  const float b1w = (float)(1.0 - P.beta1), b2 = (float)P.beta2, b2w = (float)(1.0 - P.beta2);
  const float eps = (float)P.eps;
  // Three independent pieces of latency-bound work; they start from opposite ends of the block (theta from thread 0 up,
  // the interface dofs from the last thread down, the partial sums' loads first of all), so that for the usual sizes
  // different waves carry them and their global round trips overlap instead of queueing behind each other.
  double c = 0.0;
  for (int i = threadIdx.x; i < nb_node; i += blockDim.x) c += (double)P.partials[PF_PART_U2 + i];
  {
    const float step_size = P.state->step_size_u;
    for (int k = (int)blockDim.x - 1 - (int)threadIdx.x; k < P.n_shared; k += blockDim.x) {
      const int dof = P.shared_dofs[k];
      const unsigned fl = M.dof_flags[dof];
      float uo = P.u[dof];
      if (fl & PF_DOF_FIXED) {
        if (uo != 0.f) P.u[dof] = 0.f;
        continue;
      }
      const float gu = buf2[P.shared_slot[k]];
      float m = P.m_u[dof], v = P.v_u[dof];
      m = m + b1w * (gu - m);
      v = v * b2;
      v = v + (b2w * gu) * gu;
      const float denom = sqrtf(v) / bc2s + eps;
      uo = uo + (-step_size) * (m / denom);
      if (!(fl & PF_DOF_GHOST)) c += (double)(uo * uo);
      P.m_u[dof] = m;
      P.v_u[dof] = v;
      P.u[dof] = uo;
    }
  }
  {
    const float step_size = P.state->step_size_t;
    for (int q = threadIdx.x; q < P.n_theta_active; q += blockDim.x) {
      const float g = P.grad_theta[q];
      float m = P.m_t[q], v = P.v_t[q], th = P.theta[q];
      m = m + b1w * (g - m);
      v = v * b2;
      v = v + (b2w * g) * g;
      const float denom = sqrtf(v) / bc2s + eps;
      th = th + (-step_size) * (m / denom);
      P.m_t[q] = m;
      P.v_t[q] = v;
      P.theta[q] = th;
      P.theta_pad[P.pad_index[q]] = th;
      if (new_theta) new_theta[q] = th;
    }
  }
  if (new_theta) {
    __syncthreads();
    pack_net_ops(P, new_theta);
  }
  const double tc = pf_block_sum_d(c, dred);
  if (threadIdx.x == 0) sums3[0] = (float)tc;
  // the iteration's bookkeeping from the reduced sums, in the same block (one launch less per sharded iteration):
  // [sum r^2, sum d^2] of this iteration and sum u_free^2 of the previous one (u2_lag) sit behind grad_theta in buf2
  if (!new_theta) __threadfence_block();       // theta written above is read back from global memory
  __syncthreads();
  const float* tail = buf2 + P.n_iface + P.n_theta_active;
  finalize_body(P, 0, 0, 0, tail, tail + 2, 1, 0, new_theta, 0);
}

// end of a chunk of sharded iterations: the last iteration's reduced sum u_free^2 completes the last history row
__global__ void k_shard_flush(pf_problem P, const float* __restrict__ u2) {
  pf_state* S = P.state;
  const int it = S->iter;            // iterations completed
  const float un = sqrtf(u2[0]);
  S->u_norm = un;
  if (P.hist && it >= 1 && it <= P.max_iter) P.hist[(size_t)(it - 1) * PF_HIST_COLS + 3] = un;
}

__global__ void k_reset(pf_problem P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < P.mesh.n_dofs) { P.m_u[i] = 0.f; P.v_u[i] = 0.f; }
  if (i < P.n_theta) { P.m_t[i] = 0.f; P.v_t[i] = 0.f; }
  if (i == 0) {
    pf_state* S = P.state;
    S->iter = 0; S->done = 0; S->converged = 0; S->theta_half = 0; S->u_half = 0;
    const double bc1 = 1.0 - P.beta1, bc2 = 1.0 - P.beta2;
    S->step_size_u = (float)((double)P.lr_u / bc1);
    S->step_size_t = (float)((double)P.lr_t / bc1);
    S->bc2_sqrt = (float)sqrt(bc2);
    S->loss_total = S->loss_physics = S->loss_data = 0.f;
    S->u_norm = S->residual_norm = S->theta_norm = 0.f;
  }
}

// ---- scalar (E, A) identification (pf_scalar_gd_iterations; include/pinnfem_hip.h) ---------------------------------------
// residual of the unit-stiffness operator scaled by c = exp(p_E + p_A), its adjoint seed g_f = c g_r, and the block sums
// the scalar update needs: sum r^2 | sum d^2 | sum g_r (K_1 u)   (slots R2H(0), D2H(0), R2H(1) of the partial sums)
template <int DIM>
__global__ __launch_bounds__(PF_NODE_THREADS) void k_scalar_residual(pf_problem P, pf_scalar_id SP) {
  __shared__ float red[16];
  const pf_mesh& M = P.mesh;
  const float c = expf(SP.p[0] + SP.p[1]);
  const float gcoef = P.alpha_physics * 2.f / SP.n_free_f;          // d(alpha * mean r^2) / dr = gcoef * r
  float sum_r2 = 0.f, sum_d2 = 0.f, sum_gk = 0.f;
  for (int node = blockIdx.x * blockDim.x + threadIdx.x; node < M.n_nodes; node += gridDim.x * blockDim.x) {
    float f[2];
    gather_kv<DIM>(P, P.u, node, f);
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      const int dof = node * DIM + k;
      const unsigned fl = M.dof_flags[dof];
      float gf = 0.f;
      if (!(fl & PF_DOF_FIXED)) {
        const float r = c * f[k] - M.f_ext[dof] * SP.inv_ea0;
        const float gr = gcoef * r;
        sum_r2 += r * r;
        sum_gk += gr * f[k];
        gf = c * gr;
      }
      P.g_f[dof] = gf;
      if (P.use_data && (fl & PF_DOF_MEASURED)) {
        const float d = M.meas_val[dof] - P.u[dof];
        sum_d2 += d * d;
      }
    }
  }
  const float t0 = pf_block_sum(sum_r2, red);
  const float t1 = pf_block_sum(sum_d2, red);
  const float t2 = pf_block_sum(sum_gk, red);
  if (threadIdx.x == 0) {
    P.partials[PF_PART_R2H(0) + blockIdx.x] = t0;
    P.partials[PF_PART_D2H(0) + blockIdx.x] = t1;
    P.partials[PF_PART_R2H(1) + blockIdx.x] = t2;
  }
}

// one block: losses, d loss / d(p_E + p_A), Adam on (p_E, p_A) (torch single-tensor arithmetic), bounds, table row, the
// next step's Adam scalars of the displacement update
__global__ __launch_bounds__(256) void k_scalar_update(pf_problem P, pf_scalar_id SP, int nb_node) {
  PF_NO_CONTRACT
  __shared__ double dred[16];
  pf_state* S = P.state;
  const int it = S->iter;
  const float c = expf(SP.p[0] + SP.p[1]);
  double a = 0.0, b = 0.0, g = 0.0;
  for (int i = threadIdx.x; i < nb_node; i += blockDim.x) {
    a += (double)P.partials[PF_PART_R2H(0) + i];
    b += (double)P.partials[PF_PART_D2H(0) + i];
    g += (double)P.partials[PF_PART_R2H(1) + i];
  }
  const float sum_r2 = (float)pf_block_sum_d(a, dred);
  const float sum_d2 = (float)pf_block_sum_d(b, dred);
  const float sum_gk = (float)pf_block_sum_d(g, dred);
  if (threadIdx.x != 0) return;
  const float loss_p = sum_r2 / SP.n_free_f;
  const float loss_d = P.use_data ? sum_d2 / P.n_meas_f : 0.f;
  const float loss = P.alpha_physics * loss_p + P.alpha_data * loss_d;
  const float g_c = sum_gk * c;                                     // the same gradient for p_E and p_A
  const double t = (double)(it + 1);
  const double bc1 = 1.0 - pow(P.beta1, t), bc2 = 1.0 - pow(P.beta2, t);
  const float step_size = (float)((double)SP.lr_p / bc1), bc2s = (float)sqrt(bc2);
  const float b1w = (float)(1.0 - P.beta1), b2 = (float)P.beta2, b2w = (float)(1.0 - P.beta2), eps = (float)P.eps;
  float pn[2];
  for (int k = 0; k < 2; ++k) {
    float m = SP.m_p[k], v = SP.v_p[k], x = SP.p[k];
    m = m + b1w * (g_c - m);
    v = v * b2;
    v = v + (b2w * g_c) * g_c;
    const float denom = sqrtf(v) / bc2s + eps;
    x = x + (-step_size) * (m / denom);
    if (SP.has_bounds) x = fminf(fmaxf(x, SP.lo[k]), SP.hi[k]);
    SP.m_p[k] = m; SP.v_p[k] = v; SP.p[k] = x;
    pn[k] = x;
  }
  if (it < SP.n_rows) {
    float* row = SP.table + (size_t)it * 5;
    row[0] = loss; row[1] = loss_p; row[2] = loss_d; row[3] = pn[0]; row[4] = pn[1];
  }
  S->loss_total = loss; S->loss_physics = loss_p; S->loss_data = loss_d; S->residual_norm = sqrtf(sum_r2);
  S->iter = it + 1;
  const double t2 = (double)(it + 2);
  S->step_size_u = (float)((double)P.lr_u / (1.0 - pow(P.beta1, t2)));
  S->bc2_sqrt = (float)sqrt(1.0 - pow(P.beta2, t2));
}

// ---- generic Adam ------------------------------------------------------------------------------
__global__ void k_adam(float* p, const float* g, float* m, float* v, int n, float step_size,
                       float bc2s, float b1w, float b2, float b2w, float eps) {
  PF_NO_CONTRACT
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  float mi = m[i], vi = v[i];
  mi = mi + b1w * (gi - mi);
  vi = vi * b2;
  vi = vi + (b2w * gi) * gi;
  const float denom = sqrtf(vi) / bc2s + eps;
  p[i] = p[i] + (-step_size) * (mi / denom);
  m[i] = mi;
  v[i] = vi;
}

// ---- extensions: diag(K), dense K -----------------------------------------------------------------
template <int DIM>
__global__ void k_diag_k(pf_problem P, float* diag) {
  const pf_mesh& M = P.mesh;
  const int node = blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= M.n_nodes) return;
  float d[2] = {0.f, 0.f};
  for (int idx = M.adj_ptr[node]; idx < M.adj_ptr[node + 1]; ++idx) {
    const int e = M.adj[idx] >> 1;
    const ElemK k = load_k<DIM>(P, e);
    if (DIM == 2) { d[0] += k.c2; d[1] += k.s2; }
    else d[0] += k.c2;
  }
#pragma unroll
  for (int c = 0; c < DIM; ++c) diag[node * DIM + c] = d[c];
}

// one thread per element, sequential over elements would be needed for the reference's exact
// accumulation order; dense K is a small-problem compatibility view, so a single block walks
// the elements in order (n_dofs <= 4096).
template <int DIM>
__global__ void k_dense_k(pf_problem P, float* K) {
  const pf_mesh& M = P.mesh;
  const int nd = 2 * DIM;
  for (int e = 0; e < M.n_elems; ++e) {
    const int2 nn = reinterpret_cast<const int2*>(M.conn)[e];
    const ElemK k = load_k<DIM>(P, e);
    const int t = threadIdx.x;
    if (t < nd * nd) {
      const int a = t / nd, b = t % nd;
      int da, db;
      float pat;
      if (DIM == 2) {
        da = (a < 2 ? nn.x : nn.y) * 2 + (a & 1);
        db = (b < 2 ? nn.x : nn.y) * 2 + (b & 1);
        const float base = ((a & 1) == 0 && (b & 1) == 0) ? k.c2 : (((a & 1) && (b & 1)) ? k.s2 : k.cs);
        pat = ((a < 2) == (b < 2)) ? base : -base;     // entry of ke = s*pattern (the product is already in k)
      } else {
        da = a == 0 ? nn.x : nn.y;
        db = b == 0 ? nn.x : nn.y;
        pat = a == b ? k.c2 : -k.c2;
      }
      K[(size_t)da * M.n_dofs + db] += pat;
    }
    __syncthreads();
  }
}

// k_global in coordinate format (the reference's 16 `k_global[g, h] += ke[a, b]` per element, nn_assembly.py:228-229, kept
// as triplets instead of summed into a dense matrix): entry t of element e at e*(2*DIM)^2 + t, rows / cols int64 like
// torch.sparse_coo_tensor wants them; duplicates (shared nodes) are summed by the consumer (coalesce).
template <int DIM>
__global__ __launch_bounds__(256) void k_coo_k(pf_problem P, long long* __restrict__ rows, long long* __restrict__ cols,
                                               float* __restrict__ vals) {
  const pf_mesh& M = P.mesh;
  constexpr int nd = 2 * DIM;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < M.n_elems; e += gridDim.x * blockDim.x) {
    const int2 nn = reinterpret_cast<const int2*>(M.conn)[e];
    const ElemK k = load_k<DIM>(P, e);
    const size_t base = (size_t)e * nd * nd;
#pragma unroll
    for (int a = 0; a < nd; ++a)
#pragma unroll
      for (int b = 0; b < nd; ++b) {
        int da, db;
        float pat;
        if (DIM == 2) {
          da = (a < 2 ? nn.x : nn.y) * 2 + (a & 1);
          db = (b < 2 ? nn.x : nn.y) * 2 + (b & 1);
          const float v = ((a & 1) == 0 && (b & 1) == 0) ? k.c2 : (((a & 1) && (b & 1)) ? k.s2 : k.cs);
          pat = ((a < 2) == (b < 2)) ? v : -v;
        } else {
          da = a == 0 ? nn.x : nn.y;
          db = b == 0 ? nn.x : nn.y;
          pat = a == b ? k.c2 : -k.c2;
        }
        rows[base + a * nd + b] = da;
        cols[base + a * nd + b] = db;
        vals[base + a * nd + b] = pat;
      }
  }
}

}  // namespace

// ---- host launchers (called from pf_api.hip) ---------------------------------------------------
#define PF_CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_HIP)

int pf_node_blocks(int n_nodes) {
  // measured at 10^6 nodes (MI355X) with the pairwise walk of the node kernels (two nodes of a thread at a time):
  // 1024 blocks (4 waves per SIMD, two pairs per thread) 0.1398 ms per iteration against 0.1412 (2048), 0.1417 (1280),
  // 0.1419 (768), 0.1432 (512) (profiles/r03_ab.txt; the one-node-at-a-time kernels of round 2 were best at 2048).
  // PF_NODE_BLOCKS: experiment knob.
  static const int cap = getenv("PF_NODE_BLOCKS") ? atoi(getenv("PF_NODE_BLOCKS")) : 1024;
  int nb = (n_nodes + PF_NODE_THREADS - 1) / PF_NODE_THREADS;
  if (nb > cap) nb = cap;
  if (nb > PF_MAX_NODE_BLOCKS) nb = PF_MAX_NODE_BLOCKS;
  if (nb < 1) nb = 1;
  return nb;
}

int pf_launch_node_residual(const pf_problem* p, float* f_int_out, int compute_loss, hipStream_t s, int fin_prev) {
  const int nb = pf_node_blocks(p->mesh.n_nodes) + (fin_prev ? 1 : 0);
  if (p->mesh.dim == 2)
    hipLaunchKernelGGL(k_node_residual<2>, dim3(nb), dim3(PF_NODE_THREADS), 0, s, *p, f_int_out, compute_loss, fin_prev);
  else
    hipLaunchKernelGGL(k_node_residual<1>, dim3(nb), dim3(PF_NODE_THREADS), 0, s, *p, f_int_out, compute_loss, fin_prev);
  return PF_CHECK_LAUNCH();
}

int pf_launch_elem_adjoint(const pf_problem* p, hipStream_t s) {
  int nb = (p->mesh.n_elems + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  if (p->mesh.dim == 2) hipLaunchKernelGGL(k_elem_adjoint<2>, dim3(nb), dim3(256), 0, s, *p);
  else hipLaunchKernelGGL(k_elem_adjoint<1>, dim3(nb), dim3(256), 0, s, *p);
  return PF_CHECK_LAUNCH();
}

int pf_launch_node_gradu(const pf_problem* p, int fuse_adam, hipStream_t s, int skip_shared, float* u_out) {
  const int nb = pf_node_blocks(p->mesh.n_nodes);
  const dim3 g(nb), b(PF_NODE_THREADS);
  const int out_alt = u_out != nullptr && u_out == p->u_alt ? 1 : 0;
  // PF_GRADU_LDS (experiment knob): dynamic LDS bytes the side-branch launch of the iteration graph (u_out != null) asks
  // for without using them — caps how many of its blocks a CU holds, so that a forward block always finds room beside them
  static const int lds_knob = getenv("PF_GRADU_LDS") ? atoi(getenv("PF_GRADU_LDS")) : 0;
  const size_t lds = u_out ? (size_t)lds_knob : 0;
  if (p->mesh.dim == 2) {
    if (fuse_adam) hipLaunchKernelGGL((k_node_gradu<2, true>), g, b, lds, s, *p, skip_shared, u_out, out_alt);
    else hipLaunchKernelGGL((k_node_gradu<2, false>), g, b, 0, s, *p, skip_shared, (float*)nullptr, 0);
  } else {
    if (fuse_adam) hipLaunchKernelGGL((k_node_gradu<1, true>), g, b, lds, s, *p, skip_shared, u_out, out_alt);
    else hipLaunchKernelGGL((k_node_gradu<1, false>), g, b, 0, s, *p, skip_shared, (float*)nullptr, 0);
  }
  return PF_CHECK_LAUNCH();
}

// end of a replay of the iteration graph: a stop raised in mid-replay can leave the final displacements in u_alt
// (state->u_half): bring them home.  Nothing to do otherwise (one uniform load per block).
__global__ __launch_bounds__(256) void k_u_home(pf_problem P) {
  if (!P.state->u_half || !P.u_alt) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.mesh.n_dofs; i += gridDim.x * blockDim.x) P.u[i] = P.u_alt[i];
}
int pf_launch_u_home(const pf_problem* p, hipStream_t s) {
  int nb = (p->mesh.n_dofs + 255) / 256;
  if (nb > 64) nb = 64;       // it copies only after a stop in mid-replay; in every other replay it should cost nothing
  hipLaunchKernelGGL(k_u_home, dim3(nb), dim3(256), 0, s, *p);
  return PF_CHECK_LAUNCH();
}

int pf_launch_theta_stage1(const pf_problem* p, hipStream_t s) {
  if (p->n_theta_active <= 0) return PF_OK;
  const int nb_rows = pf_net_blocks(p);
  hipLaunchKernelGGL(k_theta_stage1, dim3((p->pad_total + 63) / 64, PF_RG), dim3(256), 0, s, *p, nb_rows);
  return PF_CHECK_LAUNCH();
}

int pf_launch_theta_stage2(const pf_problem* p, int fuse_adam, hipStream_t s) {
  if (p->n_theta_active <= 0) return PF_OK;
  const size_t lds = p->wg_mode == PF_WG_MFMA32 ? (size_t)p->n_theta_active * sizeof(float) : 0;
  if (lds > 60000) { pf_set_error("too many trainable parameters for the fused theta update"); return PF_ERR_UNSUPPORTED; }
  hipLaunchKernelGGL(k_theta_stage2, dim3(1), dim3(1024), lds, s, *p, fuse_adam);
  return PF_CHECK_LAUNCH();
}

int pf_launch_theta_reduce(const pf_problem* p, int fuse_adam, hipStream_t s) {
  if (p->n_theta_active <= 0) return PF_OK;
  int rc = pf_launch_theta_stage1(p, s);
  if (rc != PF_OK) return rc;
  return pf_launch_theta_stage2(p, fuse_adam, s);
}

int pf_launch_pack_theta(const pf_problem* p, hipStream_t s) {
  if (p->n_theta_active <= 0) return PF_OK;
  hipLaunchKernelGGL(k_pack_theta, dim3(1), dim3(256), 0, s, *p);
  return PF_CHECK_LAUNCH();
}

// with_theta: also run theta stage 2 (+Adam in mode 0) inside the finalize block
int pf_launch_finalize(const pf_problem* p, int mode, int with_theta, hipStream_t s, int tn_ready) {
  const int nb_node = pf_node_blocks(p->mesh.n_nodes);
  const int wt = with_theta && p->n_theta_active > 0;
  const size_t lds = wt ? (size_t)p->n_theta_active * sizeof(float) : 0;
  if (lds > 60000) {
    pf_set_error("too many trainable parameters for the fused finalize kernel");
    return PF_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(wt ? 1024 : PF_FIN_THREADS), lds, s, *p, nb_node, mode, wt,
                     (const float*)nullptr, (const float*)nullptr, 0, tn_ready);
  return PF_CHECK_LAUNCH();
}

int pf_launch_reset(const pf_problem* p, hipStream_t s) {
  int n = p->mesh.n_dofs > p->n_theta ? p->mesh.n_dofs : p->n_theta;
  if (n < 1) n = 1;
  hipLaunchKernelGGL(k_reset, dim3((n + 255) / 256), dim3(256), 0, s, *p);
  return PF_CHECK_LAUNCH();
}

int pf_launch_adam(float* param, const float* grad, float* m, float* v, int n, int step, double lr,
                   double beta1, double beta2, double eps, hipStream_t s) {
  if (n <= 0) return PF_OK;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(k_adam, dim3((n + 255) / 256), dim3(256), 0, s, param, grad, m, v, n,
                     (float)(lr / bc1), (float)sqrt(bc2), (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), (float)eps);
  return PF_CHECK_LAUNCH();
}

int pf_launch_diag_k(const pf_problem* p, float* diag, hipStream_t s) {
  const int nb = (p->mesh.n_nodes + 255) / 256;
  if (p->mesh.dim == 2) hipLaunchKernelGGL(k_diag_k<2>, dim3(nb), dim3(256), 0, s, *p, diag);
  else hipLaunchKernelGGL(k_diag_k<1>, dim3(nb), dim3(256), 0, s, *p, diag);
  return PF_CHECK_LAUNCH();
}

int pf_launch_dense_k(const pf_problem* p, float* K, hipStream_t s) {
  if (p->mesh.dim == 2) hipLaunchKernelGGL(k_dense_k<2>, dim3(1), dim3(64), 0, s, *p, K);
  else hipLaunchKernelGGL(k_dense_k<1>, dim3(1), dim3(64), 0, s, *p, K);
  return PF_CHECK_LAUNCH();
}

int pf_launch_coo_k(const pf_problem* p, long long* rows, long long* cols, float* vals, hipStream_t s) {
  int nb = (p->mesh.n_elems + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  if (p->mesh.dim == 2) hipLaunchKernelGGL(k_coo_k<2>, dim3(nb), dim3(256), 0, s, *p, rows, cols, vals);
  else hipLaunchKernelGGL(k_coo_k<1>, dim3(nb), dim3(256), 0, s, *p, rows, cols, vals);
  return PF_CHECK_LAUNCH();
}

int pf_launch_scalar_residual(const pf_problem* p, const pf_scalar_id* sp, hipStream_t s) {
  const int nb = pf_node_blocks(p->mesh.n_nodes);
  if (p->mesh.dim == 2) hipLaunchKernelGGL(k_scalar_residual<2>, dim3(nb), dim3(PF_NODE_THREADS), 0, s, *p, *sp);
  else hipLaunchKernelGGL(k_scalar_residual<1>, dim3(nb), dim3(PF_NODE_THREADS), 0, s, *p, *sp);
  return PF_CHECK_LAUNCH();
}
int pf_launch_scalar_update(const pf_problem* p, const pf_scalar_id* sp, hipStream_t s) {
  hipLaunchKernelGGL(k_scalar_update, dim3(1), dim3(256), 0, s, *p, *sp, pf_node_blocks(p->mesh.n_nodes));
  return PF_CHECK_LAUNCH();
}

// ---- multi-GPU launchers ---------------------------------------------------------------------------
int pf_launch_shard_pack(const pf_problem* p, float* buf2, const float* u2_local, hipStream_t s) {
  const int nb = pf_node_blocks(p->mesh.n_nodes);
  if (p->mesh.dim == 2) hipLaunchKernelGGL(k_shard_pack<2>, dim3(1), dim3(1024), 0, s, *p, nb, buf2, u2_local);
  else hipLaunchKernelGGL(k_shard_pack<1>, dim3(1), dim3(1024), 0, s, *p, nb, buf2, u2_local);
  return PF_CHECK_LAUNCH();
}
int pf_launch_shard_update(const pf_problem* p, const float* buf2, float* u2_local, hipStream_t s) {
  const int nb = pf_node_blocks(p->mesh.n_nodes);
  const size_t lds = p->wg_mode == PF_WG_MFMA32 ? (size_t)p->n_theta_active * sizeof(float) : 0;
  hipLaunchKernelGGL(k_shard_update, dim3(1), dim3(1024), lds, s, *p, nb, buf2, u2_local);
  return PF_CHECK_LAUNCH();
}
int pf_launch_shard_flush(const pf_problem* p, const float* u2, hipStream_t s) {
  hipLaunchKernelGGL(k_shard_flush, dim3(1), dim3(1), 0, s, *p, u2);
  return PF_CHECK_LAUNCH();
}
