// pf_net32.h — layout of the split-f16 operand image of one MLP (MFMA32 engine, pf_net32.hip) and the
// device routine that builds it from the torch-order parameters.  Shared by pf_net32.hip (consumer) and
// pf_mesh.hip (producer: pf_pack_theta and the theta update of every iteration).
//
// Unit u of a layer lives in accumulator register r = u>>1 of the lanes of half-wave h = u&1 of a
// v_mfma_f32_32x32x16_f16 result tile, i.e. in tile row rho(u) = (r&3) + 8*(r>>2) + 4*h, the ELEMENT on the
// tile column (= lane&31).  A lane's result registers 8s..8s+7 are then, converted to f16, exactly its eight
// k-slots of k-step s of the next product (slot j of half h carries unit 2*(8s+j)+h): activations never leave
// their lane between layers, and only the weight operand has to agree with that order — which is what the
// image below encodes.  Units >= width are structural zeros.
#pragma once
#include "pf_common.h"

#define PF_N32_MAX_WIDTH 30        /* units 30, 31 (register 15) are reserved: bias column of the gradient tiles */
#define PF_N32_KA 2048.0f          /* activations are carried as KA * tanh: lo = a' - f16(a') stays a normal f16 */
#define PF_N32_KW 16.0f            /* forward weights as KW * w, backward (transposed) as KB * w */
#define PF_N32_KB 16.0f
// load factor in the gradient tile: kl = 2^(14 - e) with |lam| = m 2^e, 0.5 <= m < 1, so that kl |lam| is in
// [2^13, 2^14) whatever the load factor (NaN / inf / 0: 1)
__host__ __device__ inline float pf_n32_lam_scale(float lam) {
  const float a = lam < 0.f ? -lam : lam;
  if (!(a > 0.f) || !(a < 3.0e38f)) return 1.0f;
  int ex = 0;
  (void)frexpf(a, &ex);
  ex = ex < -100 ? -100 : ex;
  return ldexpf(1.0f, 14 - ex);
}

// byte offsets inside one net's image (all multiples of 16)
__host__ __device__ constexpr int pf_n32_off_w1() { return 16; }                       // [16 r][2 h] float4: w1[u][0..in-1], b1[u]
__host__ __device__ constexpr int pf_n32_off_wo() { return 16 + 512; }                 // [2 h][16 r] float: wo[2r+h]
__host__ __device__ constexpr int pf_n32_off_bo() { return 16 + 512 + 128; }           // float bo (+pad)
__host__ __device__ constexpr int pf_n32_off_layers() { return 16 + 512 + 128 + 16; }
// per hidden layer l = 2..L: bias C-vector [2 h][16 r] float (KA*KW*b), forward operand, backward operand
#define PF_N32_LAYER_BYTES (128 + 4096 + 4096)
__host__ __device__ constexpr int pf_n32_off_bias(int l) { return pf_n32_off_layers() + (l - 2) * PF_N32_LAYER_BYTES; }
__host__ __device__ constexpr int pf_n32_off_af(int l) { return pf_n32_off_bias(l) + 128; }   // [2 split][2 ks][64 lane][8] f16
__host__ __device__ constexpr int pf_n32_off_ab(int l) { return pf_n32_off_af(l) + 4096; }
__host__ __device__ constexpr int pf_n32_bytes(int n_hidden) { return pf_n32_off_layers() + (n_hidden - 1) * PF_N32_LAYER_BYTES; }
// header: float[0] = bound on max_l 4^(L-l) |d_l| / |g_z| (the backward's power-of-two scaling), float[1..3] spare

// Build the image of net `which` from its parameters `th` (torch parameters() order: W1 [w][in], b1, (Wl [w][w],
// bl)*, Wo [1][w], bo).  Called by every thread of one block (a multiple of 64 threads); no barriers, reads only th.
// prec 0: split f16 (hi, lo); prec 1: plain bf16 in the hi slots (round to nearest), lo slots zero.
// with_bound: also compute the header's backward-scaling bound (only the backward kernels read it: a forward launch that
// builds the image in its own LDS skips it).
// t_first, t_count: the (multiple of 64) thread range of the block that does the work (default: the whole block), so that
// two nets can be packed side by side by the two halves of a block; threads outside the range return at once.
__device__ inline void pf_n32_pack(const pf_net& net, const float* th, unsigned char* img, int prec, bool with_bound = true,
                                   int t_first = 0, int t_count = 0) {
  const int W = net.width, L = net.n_hidden, IN = net.in_dim;
  const int o_b1 = W * IN, o_h = W * IN + W, per = W * W + W, o_wo = o_h + (L - 1) * per, o_bo = o_wo + W;
  const int nt = t_count > 0 ? t_count : (int)blockDim.x;
  const int tid = (int)threadIdx.x - t_first;
  if (tid < 0 || tid >= nt) return;
  // layer 1 rows and the output row (float)
  float4* w1 = reinterpret_cast<float4*>(img + pf_n32_off_w1());
  for (int i = tid; i < 32; i += nt) {
    const int r = i >> 1, h = i & 1, u = 2 * r + h;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (u < W) {
      v.x = th[u * IN + 0];
      v.y = th[u * IN + 1];
      if (IN == 3) { v.z = th[u * IN + 2]; v.w = th[o_b1 + u]; }
      else { v.z = th[o_b1 + u]; }
    }
    w1[i] = v;
    const int hh = i >> 4, rr = i & 15, uu = 2 * rr + hh;
    reinterpret_cast<float*>(img + pf_n32_off_wo())[i] = uu < W ? th[o_wo + uu] : 0.f;
  }
  if (tid == 0) reinterpret_cast<float*>(img + pf_n32_off_bo())[0] = th[o_bo];
  for (int l = 2; l <= L; ++l) {
    const float* Wl = th + o_h + (l - 2) * per;
    const float* bl = Wl + W * W;
    float* bias = reinterpret_cast<float*>(img + pf_n32_off_bias(l));
    for (int i = tid; i < 32; i += nt) {
      const int hh = i >> 4, rr = i & 15, uu = 2 * rr + hh;
      bias[i] = uu < W ? (PF_N32_KA * PF_N32_KW) * bl[uu] : 0.f;
    }
    // operand entries: one thread builds the eight k-slots of one (split, k-step, lane) and stores them as 16 bytes
    uint4* af = reinterpret_cast<uint4*>(img + pf_n32_off_af(l));
    uint4* ab = reinterpret_cast<uint4*>(img + pf_n32_off_ab(l));
    for (int i = tid; i < 2 * 2 * 64; i += nt) {
      const int lane = i & 63, ks = (i >> 6) & 1, sp = i >> 7;
      const int rho = lane & 31, hh = lane >> 5;
      const int u_row = 2 * ((rho & 3) + 4 * (rho >> 3)) + ((rho >> 2) & 1);   // unit on tile row rho
      unsigned short hf[8], hb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int u_k = 2 * (8 * ks + j) + hh;                                   // unit in k-slot j of half hh
        float vf = 0.f, vb = 0.f;
        if (u_row < W && u_k < W) {
          vf = PF_N32_KW * Wl[u_row * W + u_k];     // z_l[u_row]   += W_l[u_row][u_k] a_{l-1}[u_k]
          vb = PF_N32_KB * Wl[u_k * W + u_row];     // dh_{l-1}[u_row] += W_l[u_k][u_row] d_l[u_k]
        }
        if (prec == 1) {
          hf[j] = sp ? (unsigned short)0 : __builtin_bit_cast(unsigned short, (__bf16)vf);
          hb[j] = sp ? (unsigned short)0 : __builtin_bit_cast(unsigned short, (__bf16)vb);
        } else {
          const _Float16 fh = (_Float16)vf, bh = (_Float16)vb;
          hf[j] = __builtin_bit_cast(unsigned short, sp ? (_Float16)(vf - (float)fh) : fh);
          hb[j] = __builtin_bit_cast(unsigned short, sp ? (_Float16)(vb - (float)bh) : bh);
        }
      }
      af[i] = make_uint4(hf[0] | (unsigned)hf[1] << 16, hf[2] | (unsigned)hf[3] << 16, hf[4] | (unsigned)hf[5] << 16,
                         hf[6] | (unsigned)hf[7] << 16);
      ab[i] = make_uint4(hb[0] | (unsigned)hb[1] << 16, hb[2] | (unsigned)hb[3] << 16, hb[4] | (unsigned)hb[5] << 16,
                         hb[6] | (unsigned)hb[7] << 16);
    }
  }
  // bound for the backward scaling: v_L[j] = |wo[j]|, v_{l-1}[k] = sum_j |W_l[j][k]| v_l[j];
  // bound = max_l 4^(L-l) max_k v_l[k]  (|d_l| <= |g_z| v_l since 4 t = 1 - a^2 <= 1).
  // One wave (the block's last, which has the least operand work), v in registers: lane k holds v[k].
  if (with_bound && (tid >> 6) == ((nt - 1) >> 6)) {
    const int lane = tid & 63;
    float v = lane < W ? fabsf(th[o_wo + lane]) : 0.f;
    float bound = 0.f, grow = 1.f;
    for (int l = L; l >= 1; --l) {
      float m = v;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
      bound = fmaxf(bound, grow * m);
      grow *= 4.f;
      if (l >= 2) {
        const float* Wl = th + o_h + (l - 2) * per;
        float a = 0.f;
#pragma unroll 4
        for (int j = 0; j < W; ++j) {
          const float w = lane < W ? fabsf(Wl[j * W + lane]) : 0.f;
          a += w * __shfl(v, j);
        }
        v = a;
      }
    }
    if (lane == 0) {
      float* hdr = reinterpret_cast<float*>(img);
      hdr[0] = bound > 0.f ? bound : 1.f;
      hdr[1] = hdr[2] = hdr[3] = 0.f;
    }
  }
}

// ---- host side of the fused forward launch (pf_net32.hip: k_net32_forward2) ------------------------------------------
// blocks of the fused forward launch: one 1024-thread block per CU (PF_FWD32_BLOCKS: experiment knob)
inline int pf_n32_fwd2_blocks(int n_elems) {
  static const int cap = getenv("PF_FWD32_BLOCKS") ? atoi(getenv("PF_FWD32_BLOCKS")) : 256;
  int nb = (n_elems + 1023) / 1024;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  return nb;
}
// nodes per lane of a node task inside the fused forward launch (k_net32_forward2, gu_nb)
#ifndef PF_GU_M
#define PF_GU_M 2
#endif
// Can the fused forward launch also run the displacement update of the previous iteration (gu_nb = entries of the
// u-norm partial sums the bookkeeping reads)?  It needs the other-end adjacency, a single-GPU mesh (no ghost elements, no
// shared dofs), every block's partial inside what the bookkeeping sums, and its node tasks inside the block's LDS table.
inline bool pf_n32_fwd2_can_update_u(const pf_problem* p, int gu_nb) {
  if (!p->adj_other || !p->m_u || !p->v_u || p->mesh.n_elems <= 0) return false;
  if (p->n_shared != 0 || p->own_lo != 0 || p->own_hi != 0) return false;
  const int nb = pf_n32_fwd2_blocks(p->mesh.n_elems);
  if (nb > gu_nb) return false;
  const int ntasks = (p->mesh.n_nodes + 64 * PF_GU_M - 1) / (64 * PF_GU_M);
  return (ntasks + nb - 1) / nb <= 1024 - 16;
}
