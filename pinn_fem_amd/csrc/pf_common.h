// pf_common.h — shared device helpers and layout constants (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "pinnfem_hip.h"

#define PF_WAVE 64

// ---- partial-sum workspace layout (floats) -------------------------------------------
// [0, 5*PF_NODE_SLOTS)            : per-node-block scalar partials: sum r^2 | sum d^2 (half 0) | sum u_free^2 |
//                                   sum r^2 | sum d^2 (half 1).  The residual's sums exist twice (pf_problem.part_half)
//                                   because the bookkeeping of iteration t runs INSIDE the residual launch of t+1
//                                   (iteration graph, pf_api.hip), which is already writing the other half.
// [5*PF_NODE_SLOTS, ...)          : per-block padded weight-gradient rows [n_part_blocks][pad_total]
#define PF_PART_R2H(h) ((h) ? 3 * PF_NODE_SLOTS : 0)
#define PF_PART_D2H(h) ((h) ? 4 * PF_NODE_SLOTS : PF_NODE_SLOTS)
#define PF_PART_U2 (2 * PF_NODE_SLOTS)
#define PF_PART_WG (5 * PF_NODE_SLOTS)
// after the [n_part_blocks][pad_total] rows: [PF_RG][pad_total] second-level partial rows, then PF_RG + 16 int32 row-group
// tickets (zero between launches) of the fused backward launch, which does the first reduction level itself
#define PF_RG 16
#define PF_TICKETS (PF_RG + 16)

// elements per block-iteration of the net kernels (2 waves)
#define PF_NET_THREADS 128
// nodes per block of the node kernels
#define PF_NODE_THREADS 256

// ---- padded parameter image of one MLP ------------------------------------------------
// layer 1      : W1e [HP][4]        cols 0..IN-1 weights, col IN bias, rest 0
// layers 2..L  : Wle [HP][HP+4]     cols 0..H-1 weights, col HP bias, rest 0
// output       : Woe [HP+4]         cols 0..H-1 weights, [HP] bias
// rows/cols >= H are zero, so the padded net computes exactly the unpadded one.
__host__ __device__ constexpr int pf_pad_w1(int) { return 0; }
__host__ __device__ constexpr int pf_pad_wh(int hp, int l /*2..L*/) {
  return hp * 4 + (l - 2) * hp * (hp + 4);
}
__host__ __device__ constexpr int pf_pad_wo(int hp, int nh) {
  return hp * 4 + (nh - 1) * hp * (hp + 4);
}
__host__ __device__ constexpr int pf_pad_count(int hp, int nh) {
  return pf_pad_wo(hp, nh) + hp + 4;
}

// ---- wave / block reductions (fixed order => bitwise reproducible) ---------------------
__device__ __forceinline__ float pf_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block sum for blockDim.x <= 1024; result valid in every thread. smem: >= 16 floats.
__device__ __forceinline__ float pf_block_sum(float v, float* smem) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = pf_wave_sum(v);
  __syncthreads();
  if (lane == 0) smem[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += smem[i];
  return t;
}

__device__ __forceinline__ double pf_block_sum_d(double v, double* smem) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if (lane == 0) smem[w] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < nw; ++i) t += smem[i];
  return t;
}

// tanh(x) = 1 - 2/(exp(2x)+1): v_mul, v_exp, v_add, v_rcp, v_fma; |abs err| <~ 2e-7 (parity tests
// run with it).  -DPF_ACCURATE_TANH switches back to the libm-grade tanhf (~25 instructions).
__device__ __forceinline__ float pf_tanh(float x) {
#ifdef PF_ACCURATE_TANH
  return tanhf(x);
#else
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  const float r = __builtin_amdgcn_rcpf(e + 1.0f);
  return fmaf(-2.0f, r, 1.0f);
#endif
}

// torch.nn.functional.softplus (beta=1, threshold=20) and its backward factor.
// softplus(z) = max(z,0) + log1p(exp(-|z|)); log1p(e) by Kahan's correction log(u)*e/(u-1), u = 1+e,
// which keeps full relative accuracy for tiny e without libm's ~100-instruction log1pf.
__device__ __forceinline__ float pf_softplus(float z) {
  if (z > 20.f) return z;
  const float e = expf(-fabsf(z));
  const float u = 1.f + e;
  const float d = u - 1.f;                       // exact
  const float l = d == 0.f ? e : __logf(u) * (e * __builtin_amdgcn_rcpf(d));
  return fmaxf(z, 0.f) + l;
}
// torch softplus_backward: z > 20 ? 1 : e^z/(e^z+1)
__device__ __forceinline__ float pf_softplus_grad(float z) {
  if (z > 20.f) return 1.f;
  const float ez = expf(z);
  return ez * __builtin_amdgcn_rcpf(ez + 1.f);
}

// torch.optim.Adam is a sequence of separately rounded tensor ops (lerp_, mul_, addcmul_, sqrt, div, add,
// addcdiv_); with the default -ffp-contract=fast hipcc fuses some of them into fma, and not the same ones in
// every inlined copy (measured: the hipGraph path, which runs the theta update as its own kernel, and the eager
// path, which runs it inside k_finalize, drifted apart by 1 ulp in theta).  Contraction is switched off in
// every function that carries optimiser arithmetic.
#define PF_NO_CONTRACT _Pragma("clang fp contract(off)")

// the two halves of the parameter state (pf_problem.theta_alt): which = 0 theta, 1 m_t, 2 v_t
__device__ __forceinline__ float* pf_theta_half_ptr(const pf_problem& P, int half, int which) {
  if (half == 0) return which == 0 ? P.theta : (which == 1 ? P.m_t : P.v_t);
  return P.theta_alt + (size_t)which * P.n_theta;
}

// Padded-image index of active parameter q (what pf_problem.pad_index[q] holds: pf_net_pad_index of the parameter's net +
// the net's pad_off), by arithmetic: the forward launch's update prologue is a chain of dependent global round trips with
// an idle vector ALU, and this replaces the first of them (index table -> row addresses) in every block of the launch.
// Active parameters are laid out net by net, young before area.
__host__ __device__ __forceinline__ int pf_pad_index_of(const pf_problem& P, int q) {
  const int k = (P.net[1].enabled && (!P.net[0].enabled || q >= P.net[1].theta_off)) ? 1 : 0;
  const int W = P.net[k].width, IN = P.net[k].in_dim, L = P.net[k].n_hidden;
  const int hp = ((W + 3) / 4) * 4;
  int r = q - P.net[k].theta_off;
  const int off = P.net[k].pad_off;
  if (r < W * IN) return off + (r / IN) * 4 + (r % IN);        // W1 (width, in_dim)
  r -= W * IN;
  if (r < W) return off + r * 4 + IN;                          // b1
  r -= W;
  for (int l = 2; l <= L; ++l) {
    if (r < W * W) return off + pf_pad_wh(hp, l) + (r / W) * (hp + 4) + (r % W);
    r -= W * W;
    if (r < W) return off + pf_pad_wh(hp, l) + r * (hp + 4) + hp;
    r -= W;
  }
  if (r < W) return off + pf_pad_wo(hp, L) + r;                // output weights, then the output bias
  return off + pf_pad_wo(hp, L) + hp;
}

// Parameter update (one block, thread q -> parameter q): PF_RG second-level partial rows -> grad_theta[q], then
// optimizer_theta.step() (solver.py:293-294).  new_theta (LDS, n_theta_active floats) receives the updated parameters
// when non-null.  The state is read from half `half_in` and stored to half `half_out` (0, 0: in place).
// skip_stores: compute only (the block-uniform stop flag as loaded by the caller — the loads below are then issued WITH
// that load instead of behind a branch on it; or a block of the forward launch that only needs its own copy).
// calc_index: the padded-image index by arithmetic (pf_pad_index_of) instead of from the table.
__device__ __forceinline__ void pf_theta_update(const pf_problem& P, int fuse_adam, float* new_theta, int skip_stores = 0,
                                                int half_in = 0, int half_out = 0, bool calc_index = false) {
  PF_NO_CONTRACT
  const float* __restrict__ p2 = P.partials + PF_PART_WG + (size_t)P.n_part_blocks * P.pad_total;
  const float step_size = P.state->step_size_t, bc2s = P.state->bc2_sqrt;
  const float b1w = (float)(1.0 - P.beta1), b2 = (float)P.beta2, b2w = (float)(1.0 - P.beta2);
  const float eps = (float)P.eps;
  const float* th_i = pf_theta_half_ptr(P, half_in, 0);
  const float* m_i = pf_theta_half_ptr(P, half_in, 1);
  const float* v_i = pf_theta_half_ptr(P, half_in, 2);
  float* th_o = pf_theta_half_ptr(P, half_out, 0);
  float* m_o = pf_theta_half_ptr(P, half_out, 1);
  float* v_o = pf_theta_half_ptr(P, half_out, 2);
  for (int q = threadIdx.x; q < P.n_theta_active; q += blockDim.x) {
    const int pi = calc_index ? pf_pad_index_of(P, q) : P.pad_index[q];
    float th = th_i[q];
    float m = 0.f, v = 0.f;
    if (fuse_adam) { m = m_i[q]; v = v_i[q]; }
    float g = 0.f;
#pragma unroll
    for (int r = 0; r < PF_RG; ++r) g += p2[(size_t)r * P.pad_total + pi];
    if (fuse_adam) {
      m = m + b1w * (g - m);
      v = v * b2;
      v = v + (b2w * g) * g;
      const float denom = sqrtf(v) / bc2s + eps;
      th = th + (-step_size) * (m / denom);
    }
    if (!skip_stores) {
      P.grad_theta[q] = g;
      if (fuse_adam) {
        m_o[q] = m;
        v_o[q] = v;
        th_o[q] = th;
        P.theta_pad[pi] = th;
      }
    }
    if (new_theta) new_theta[q] = th;
  }
}

// theta_norm = sum_k ||theta_k||_2 over ALL parameter tensors (density included), solver.py:319.  The calling wave
// writes the norms of tensors first, first + step, ... to tnorm[]; each is one wave's lane-strided sum, and the caller
// adds them in tensor order, so the value does not depend on how many waves share the work.  new_theta: LDS copy of the
// ACTIVE parameters (or null: read p->theta); parameters that receive no gradient are read from p->theta either way.
#define PF_MAX_TENSORS 64
__device__ __forceinline__ void tensor_norms(const pf_problem& P, const float* new_theta, int first, int step,
                                             float* tnorm) {
  const int lane = threadIdx.x & 63;
  for (int t = first; t < P.n_tensors; t += step) {
    const int lo = P.tensor_off[t], hi = P.tensor_off[t + 1];
    float s = 0.f;
    for (int i = lo + lane; i < hi; i += 64) {
      const float x = (new_theta && i < P.n_theta_active) ? new_theta[i] : P.theta[i];
      s += x * x;
    }
    const float v = sqrtf(pf_wave_sum(s));
    if (lane == 0) tnorm[t] = v;
  }
}
__device__ __forceinline__ double tensor_norm_total(const pf_problem& P, const float* tnorm) {
  double tn = 0.0;
  for (int t = 0; t < P.n_tensors; ++t) tn += (double)tnorm[t];
  return tn;
}

// ---- element algebra shared by the node kernels and the fused backward ---------------------------
struct ElemGeo { float c2, cs, s2, l0; };

__device__ __forceinline__ ElemGeo load_geo(const float* __restrict__ egeo, int e) {
  const float4 g = reinterpret_cast<const float4*>(egeo)[e];
  return ElemGeo{g.x, g.y, g.z, g.w};
}

__device__ __forceinline__ float elem_stiffness(const pf_problem& P, int e, float l0) {
  const float E = P.net[0].enabled ? P.prop_e[e] : P.net[0].scale;
  const float A = P.net[1].enabled ? P.prop_a[e] : P.net[1].scale;
  return (E * A) / l0;  // nn_assembly.py:74 (2-D), :37 (1-D)
}

// The three distinct entries of an element's stiffness matrix ke = s * pattern (nn_assembly.py:74, 84-94): s*c2, s*cs,
// s*s2 (1-D: s in c2).  The MFMA32 forward pass writes them per element (pf_problem.elem_k: 12 B, or 4 B in 1-D), so the
// node kernels read ONE record per incidence instead of the geometry (16 B) and the stiffness (4 B); without the record
// they are formed here by the same float operations, so both routes give the same bits.
struct ElemK { float c2, cs, s2; };

template <int DIM>
__device__ __forceinline__ ElemK elem_k_from(const ElemGeo& g, float s) {
  if (DIM == 2) return ElemK{s * g.c2, s * g.cs, s * g.s2};
  return ElemK{s, 0.f, 0.f};
}
// unit stiffness: the pattern itself (1 * c2 == c2 exactly)
template <int DIM>
__device__ __forceinline__ ElemK elem_k_unit(const ElemGeo& g) {
  if (DIM == 2) return ElemK{g.c2, g.cs, g.s2};
  return ElemK{1.f, 0.f, 0.f};
}
template <int DIM>
__device__ __forceinline__ ElemK load_k_record(const float* __restrict__ elem_k, int e) {
  if (DIM == 2) {
    const float* __restrict__ r = elem_k + 3 * (size_t)e;
    return ElemK{r[0], r[1], r[2]};
  }
  return ElemK{elem_k[e], 0.f, 0.f};
}
template <int DIM>
__device__ __forceinline__ ElemK load_k(const pf_problem& P, int e) {
  if (P.elem_k) return load_k_record<DIM>(P.elem_k, e);
  const ElemGeo g = load_geo(P.mesh.egeo, e);
  return elem_k_from<DIM>(g, elem_stiffness(P, e, g.l0));
}

// rows `2*end`, `2*end+1` of ke @ [v_i; v_j], ke = s*pattern given by its entries k.
// mode PF_FE_REFERENCE: the reference's operation order, a 4-term dot per row with b ascending
//   (nn_assembly.py:96-100): (s*c2)*v_ix + (s*cs)*v_iy - (s*c2)*v_jx - (s*cs)*v_jy.  With |v| >> |v_j - v_i|
//   (long chains) this cancels in float32 exactly like the reference does.
// mode PF_FE_DELTA: the same rows applied to d = v_j - v_i (mathematically identical, no cancellation);
//   not the reference's round-off, so it is opt-in.
template <int DIM>
__device__ __forceinline__ void ke_rows_times(const ElemK& k, int end, const float* vi, const float* vj, float* out,
                                              int mode) {
  const float sg = end ? -1.f : 1.f;  // rows 2,3 are the exactly negated rows 0,1
  if (DIM == 2) {
    if (mode == PF_FE_DELTA) {
      const float d0 = vj[0] - vi[0], d1 = vj[1] - vi[1];
      const float q0 = fmaf(k.cs, d1, k.c2 * d0), q1 = fmaf(k.s2, d1, k.cs * d0);
      out[0] = -sg * q0;
      out[1] = -sg * q1;
      return;
    }
    // row 0: [ c2, cs, -c2, -cs ]   row 1: [ cs, s2, -cs, -s2 ]
    float r0 = (sg * k.c2) * vi[0];
    r0 = fmaf(sg * k.cs, vi[1], r0);
    r0 = fmaf(-(sg * k.c2), vj[0], r0);
    r0 = fmaf(-(sg * k.cs), vj[1], r0);
    float r1 = (sg * k.cs) * vi[0];
    r1 = fmaf(sg * k.s2, vi[1], r1);
    r1 = fmaf(-(sg * k.cs), vj[0], r1);
    r1 = fmaf(-(sg * k.s2), vj[1], r1);
    out[0] = r0;
    out[1] = r1;
  } else {
    if (mode == PF_FE_DELTA) {
      out[0] = (-(sg * k.c2)) * (vj[0] - vi[0]);
      return;
    }
    float r0 = (sg * k.c2) * vi[0];  // [[1,-1],[-1,1]]
    r0 = fmaf(-(sg * k.c2), vj[0], r0);
    out[0] = r0;
  }
}

template <int DIM>
__device__ __forceinline__ void load_vec(const float* __restrict__ v, int node, float* out) {
  if (DIM == 2) {
    const float2 t = reinterpret_cast<const float2*>(v)[node];
    out[0] = t.x;
    out[1] = t.y;
  } else {
    out[0] = v[node];
  }
}

// dL/d(E*A) of element e: g_s / l0 with g_s = sum_ab g_fe[a] pattern[a][b] u[b]
// (autograd through ke = s*pattern, fe = ke@u_e, s = (E*A)/l0: nn_assembly.py:74, 96-100)
template <int DIM>
__device__ __forceinline__ float pf_elem_gea(const pf_problem& P, int e) {
  const pf_mesh& M = P.mesh;
  const int2 nn = reinterpret_cast<const int2*>(M.conn)[e];
  const ElemGeo g = load_geo(M.egeo, e);
  float ui[2], uj[2], gi[2], gj[2], pu0[2], pu1[2];
  load_vec<DIM>(P.u, nn.x, ui);
  load_vec<DIM>(P.u, nn.y, uj);
  load_vec<DIM>(P.g_f, nn.x, gi);
  load_vec<DIM>(P.g_f, nn.y, gj);
  // (pattern @ u_e): rows 0,1 (end 0) and rows 2,3 (end 1) with unit stiffness
  const ElemK k1 = elem_k_unit<DIM>(g);
  ke_rows_times<DIM>(k1, 0, ui, uj, pu0, P.fe_mode);
  ke_rows_times<DIM>(k1, 1, ui, uj, pu1, P.fe_mode);
  float gs = 0.f;
#pragma unroll
  for (int c = 0; c < DIM; ++c) gs = fmaf(gi[c], pu0[c], gs);
#pragma unroll
  for (int c = 0; c < DIM; ++c) gs = fmaf(gj[c], pu1[c], gs);
  return gs / g.l0;  // div backward of (young*area)/l0
}

// blocks the element-parallel net kernels launch for n elements = rows of per-block partial weight
// gradients the theta reduction sums.  The MFMA44 engine runs up to 8 waves per block (one block per CU at
// two waves per SIMD: the same residency as 2-wave blocks, a quarter of the partial rows); the wave count is
// the largest of 8/4/2 whose gradient-tile LDS image (pf_net44.hip: Row<L>) fits 160 KiB for EVERY enabled
// net, so both backward kernels of a problem write the same number of rows.
#define PF_NET44_MAX_THREADS 512
#define PF_NET44_CS 36   /* floats per LDS column (Row<L>::CS) */
__host__ __device__ constexpr int pf_net44_row_len(int hp, int nh) { return nh * hp + 8 + nh * (hp + 4); }
inline int pf_net44_threads(const pf_problem* p) {
  // PF_NET44_THREADS (128 | 256 | 512): experiment knob
  static const int knob = getenv("PF_NET44_THREADS") ? atoi(getenv("PF_NET44_THREADS")) : PF_NET44_MAX_THREADS;
  int waves = (knob >= 128 && knob <= PF_NET44_MAX_THREADS ? knob : PF_NET44_MAX_THREADS) / 64;
  for (int k = 0; k < 2; ++k) {
    if (!p->net[k].enabled) continue;
    const int hp = ((p->net[k].width + 3) / 4) * 4;
    while (waves > 2 && (size_t)waves * pf_net44_row_len(hp, p->net[k].n_hidden) * PF_NET44_CS * 4 > 160u * 1024u) waves >>= 1;
  }
  return waves * 64;
}
// MFMA32 engine, backward: one block of 8 or 12 waves per CU (pf_net32.hip decides per kernel), every wave walks
// 64-element tasks.  Both backward launches of a problem use the same number of blocks (= partial gradient rows).
inline int pf_net32_blocks(const pf_problem* p) {
  static const int cap = getenv("PF_NET32_BLOCKS") ? atoi(getenv("PF_NET32_BLOCKS")) : 256;
  int nb = (p->mesh.n_elems + 511) / 512;
  if (nb > cap) nb = cap;
  if (nb > p->n_part_blocks) nb = p->n_part_blocks;
  if (nb < 1) nb = 1;
  return nb;
}
inline int pf_net_blocks(const pf_problem* p) {
  if (p->wg_mode == PF_WG_MFMA32) return pf_net32_blocks(p);
  const bool m44 = p->wg_mode == PF_WG_MFMA44;
  const int threads = m44 ? pf_net44_threads(p) : PF_NET_THREADS;
  int cap = m44 ? p->n_part_blocks * PF_NET_THREADS / threads : p->n_part_blocks;   // same number of waves
  if (cap < 1) cap = 1;
  int nb = (p->mesh.n_elems + threads - 1) / threads;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  return nb;
}
// blocks the node-parallel kernels launch (grid-stride over nodes, at most PF_MAX_NODE_BLOCKS: one
// partial-sum slot per block; pf_mesh.hip)
int pf_node_blocks(int n_nodes);

// launchers implemented once per padded width in pf_net.hip (compiled with -DPF_HP=<hp>)
#define PF_DECL_NET_LAUNCHERS(HP)                                                         \
  int pf_launch_net_forward_##HP(const pf_problem* p, int which, hipStream_t s);          \
  int pf_launch_net_backward_##HP(const pf_problem* p, int which, hipStream_t s);          \
  int pf_launch_net44_forward_##HP(const pf_problem* p, int which, hipStream_t s);        \
  int pf_launch_net44_backward_##HP(const pf_problem* p, int which, hipStream_t s);       \
  int pf_launch_net44_backward_gea_##HP(const pf_problem* p, int which, hipStream_t s);
PF_DECL_NET_LAUNCHERS(4)
PF_DECL_NET_LAUNCHERS(8)
PF_DECL_NET_LAUNCHERS(12)
PF_DECL_NET_LAUNCHERS(16)
PF_DECL_NET_LAUNCHERS(20)
PF_DECL_NET_LAUNCHERS(24)
PF_DECL_NET_LAUNCHERS(28)
PF_DECL_NET_LAUNCHERS(32)

// What a fused forward launch (pf_net32.hip: k_net32_forward2) does besides the two forward passes (the iteration graph):
//   s2_half >= 0  the parameter update of the previous iteration from that state half (fwd_theta_prologue)
//   calc_index    ... with the padded-image index of a parameter by arithmetic (pf_pad_index_of; the caller has checked
//                 that pf_problem.pad_index holds exactly that) instead of from the table
//   gu_nb > 0     the displacement update of the previous iteration (pf_node.h) reading the stiffness records gu_k; gu_nb =
//                 entries of the u-norm partial sums the bookkeeping reads
struct pf_fwd2_opts {
  int s2_half = -1;
  bool calc_index = false;
  int gu_nb = 0;
  const float* gu_k = nullptr;
};
// launchers of pf_net32.hip, one translation unit per register bucket (-DPF_NR=<nr>): nets of width <= 2*nr
#define PF_DECL_NET32_LAUNCHERS(NRB)                                                       \
  int pf_launch_net32_forward_##NRB(const pf_problem* p, int which, hipStream_t s, int s2_half); \
  int pf_launch_net32_backward_##NRB(const pf_problem* p, int which, hipStream_t s);      \
  int pf_launch_net32_backward_gea_##NRB(const pf_problem* p, int which, hipStream_t s);  \
  int pf_launch_net32b_forward_##NRB(const pf_problem* p, int which, hipStream_t s, int s2_half); \
  int pf_launch_net32b_backward_##NRB(const pf_problem* p, int which, hipStream_t s);     \
  int pf_launch_net32b_backward_gea_##NRB(const pf_problem* p, int which, hipStream_t s); \
  int pf_launch_net32_forward2_##NRB(const pf_problem* p, hipStream_t s, const pf_fwd2_opts& o);    \
  int pf_launch_net32b_forward2_##NRB(const pf_problem* p, hipStream_t s, const pf_fwd2_opts& o);   \
  int pf_launch_net32_backward2_##NRB(const pf_problem* p, hipStream_t s, int reduce_rows); \
  int pf_launch_net32b_backward2_##NRB(const pf_problem* p, hipStream_t s, int reduce_rows);
PF_DECL_NET32_LAUNCHERS(2)
PF_DECL_NET32_LAUNCHERS(4)
PF_DECL_NET32_LAUNCHERS(6)
PF_DECL_NET32_LAUNCHERS(8)
PF_DECL_NET32_LAUNCHERS(10)
PF_DECL_NET32_LAUNCHERS(12)
PF_DECL_NET32_LAUNCHERS(15)
// register bucket of a net width (1..30), or -1
inline int pf_net32_bucket(int width) {
  const int nr = (width + 1) / 2;
  if (width < 1 || width > PF_N32_WIDTH_MAX) return -1;
  return nr <= 2 ? 2 : nr <= 4 ? 4 : nr <= 6 ? 6 : nr <= 8 ? 8 : nr <= 10 ? 10 : nr <= 12 ? 12 : 15;
}

void pf_set_error(const char* msg);

// pf_api.hip: `iters` sharded iterations as one hipGraph with the caller's collective captured inside (pf_comm.hip)
int pf_shard_graph_capture(const pf_problem* p, int iters, float* buf, float* u2_local, hipStream_t stream,
                           int (*all_reduce)(void* ctx, float* buf, size_t n, hipStream_t s), void* ctx, void** graph_out);
