// pf_common.h — shared device helpers and layout constants (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pinnfem_hip.h"

#define PF_WAVE 64

// ---- partial-sum workspace layout (floats) -------------------------------------------
// [0, 3*PF_MAX_BLOCKS)            : per-block scalar partials  (sum r^2 | sum d^2 | sum u_free^2)
// [3*PF_MAX_BLOCKS, ...)          : per-block padded weight-gradient rows [n_blocks][pad_total]
#define PF_PART_R2 0
#define PF_PART_D2 (PF_MAX_BLOCKS)
#define PF_PART_U2 (2 * PF_MAX_BLOCKS)
#define PF_PART_WG (3 * PF_MAX_BLOCKS)
// after the [n_part_blocks][pad_total] rows: [PF_RG][pad_total] second-level partial rows
#define PF_RG 16

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

// blocks the element-parallel net kernels launch for n elements
__host__ __device__ inline int pf_net_blocks(int n_elems, int n_part_blocks) {
  int nb = (n_elems + PF_NET_THREADS - 1) / PF_NET_THREADS;
  if (nb > n_part_blocks) nb = n_part_blocks;
  if (nb < 1) nb = 1;
  return nb;
}
// blocks the node-parallel kernels launch
__host__ __device__ inline int pf_node_blocks(int n_nodes, int n_part_blocks) {
  int nb = (n_nodes + PF_NODE_THREADS - 1) / PF_NODE_THREADS;
  if (nb > n_part_blocks) nb = n_part_blocks;
  if (nb < 1) nb = 1;
  return nb;
}

// launchers implemented once per padded width in pf_net.hip (compiled with -DPF_HP=<hp>)
#define PF_DECL_NET_LAUNCHERS(HP)                                                         \
  int pf_launch_net_forward_##HP(const pf_problem* p, int which, hipStream_t s);          \
  int pf_launch_net_backward_##HP(const pf_problem* p, int which, hipStream_t s);          \
  int pf_launch_net44_forward_##HP(const pf_problem* p, int which, hipStream_t s);        \
  int pf_launch_net44_backward_##HP(const pf_problem* p, int which, hipStream_t s);
PF_DECL_NET_LAUNCHERS(4)
PF_DECL_NET_LAUNCHERS(8)
PF_DECL_NET_LAUNCHERS(12)
PF_DECL_NET_LAUNCHERS(16)
PF_DECL_NET_LAUNCHERS(20)
PF_DECL_NET_LAUNCHERS(24)
PF_DECL_NET_LAUNCHERS(28)
PF_DECL_NET_LAUNCHERS(32)

void pf_set_error(const char* msg);
