// pf_net.hip — MLP property kernels for one padded width (compile with -DPF_HP=<4..32 step 4>).
//
// Replaces, for ALL elements in one launch, the reference's per-element batch-1 calls
//   NNProperty.value            FEM/python/fem/properties.py:97-161
//   SimpleNN.forward            FEM/python/examples/json/generic.py:118-142
// and the autograd backward of both (loss.backward(), fem/solver.py:289), including the
// sum over elements of the parameter gradients.
//
// Layout: one element per lane ("natural layout"): inputs, hidden activations and their
// adjoints live in VGPRs; the padded weight image is read through wave-uniform (scalar)
// loads.  The parameter-gradient outer products  dW_l = sum_e dz_l[e] (x) a_{l-1}[e]  are the
// genuinely dense GEMMs of this path (M = K = width, reduction over elements): they run on
// the matrix cores as v_mfma_f32_16x16x4_f32 over LDS-transposed tiles, accumulating in
// registers across the whole grid-stride loop, so no atomics and a fixed summation order.
#include "pf_common.h"

#ifndef PF_HP
#error "compile with -DPF_HP=<padded width>"
#endif

#define PF_CAT2(a, b) a##b
#define PF_CAT(a, b) PF_CAT2(a, b)

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int HP = PF_HP;
constexpr int MT = (HP + 15) / 16;  // 16-row tiles over output units
constexpr int NT = HP / 16 + 1;     // 16-col tiles over input units + bias column
constexpr int TILE = 64 * 16;       // floats per [64 elements][16] tile
constexpr int WAVES = PF_NET_THREADS / 64;
constexpr int TILES_PER_WAVE = MT + NT + 1;

// ---- natural-layout MLP (registers) -----------------------------------------------------
template <int L, int IN>
struct Mlp {
  float x[4];      // ext input: IN columns, then 1, then 0-padding
  float h[L][HP];  // tanh activations
  float z;         // pre-softplus output

  __device__ __forceinline__ void forward(const float* __restrict__ w) {
#pragma unroll
    for (int j = 0; j < HP; ++j) {
      float acc = w[j * 4 + IN];
#pragma unroll
      for (int c = 0; c < IN; ++c) acc = fmaf(w[j * 4 + c], x[c], acc);
      h[0][j] = pf_tanh(acc);
    }
#pragma unroll
    for (int l = 2; l <= L; ++l) {
      const float* __restrict__ wl = w + pf_pad_wh(HP, l);
#pragma unroll
      for (int j = 0; j < HP; ++j) {
        float acc = wl[j * (HP + 4) + HP];
#pragma unroll
        for (int k = 0; k < HP; ++k) acc = fmaf(wl[j * (HP + 4) + k], h[l - 2][k], acc);
        h[l - 1][j] = pf_tanh(acc);
      }
    }
    const float* __restrict__ wo = w + pf_pad_wo(HP, L);
    float acc = wo[HP];
#pragma unroll
    for (int k = 0; k < HP; ++k) acc = fmaf(wo[k], h[L - 1][k], acc);
    z = acc;
  }
};

template <int IN>
__device__ __forceinline__ void load_input(float (&x)[4], const float* __restrict__ ecent, int e,
                                           float lam) {
  x[0] = lam;
  if (IN == 3) {
    const float2 c = reinterpret_cast<const float2*>(ecent)[e];
    x[1] = c.x;
    x[2] = c.y;
    x[3] = 1.f;
  } else {
    x[1] = ecent[e];
    x[2] = 1.f;
    x[3] = 0.f;
  }
}

// ---- forward kernel: property value per element -------------------------------------------
template <int L, int IN>
__global__ __launch_bounds__(256) void k_net_forward(pf_problem P, int which) {
  if (P.state->done) return;
  const pf_net net = P.net[which];
  const float* __restrict__ w = P.theta_pad + net.pad_off;
  float* __restrict__ out = which == 0 ? P.prop_e : P.prop_a;
  // one element per thread, no loop: every weight load precedes the only store, so the
  // compiler can keep the wave-uniform weight reads on the scalar unit (s_load_dwordx16)
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= P.mesh.n_elems) return;
  Mlp<L, IN> m;
  load_input<IN>(m.x, P.mesh.ecent, e, P.lam);
  m.forward(w);
  const float o = net.positive ? pf_softplus(m.z) : m.z;
  out[e] = o * net.scale;
}

// ---- LDS tile helpers (weight-gradient GEMM operands) ----------------------------------------
// tile[e][16] floats, 16-B chunk q of row e stored at chunk q ^ ((e>>1)&3): b128 writes of 8
// consecutive lanes and the b32 operand reads of each 32-lane half are then bank-conflict-free.
__device__ __forceinline__ void tile_write4(float* T, int e, int q, float a, float b, float c,
                                            float d) {
  *reinterpret_cast<float4*>(T + e * 16 + ((q ^ ((e >> 1) & 3)) << 2)) = make_float4(a, b, c, d);
}
__device__ __forceinline__ float tile_read(const float* T, int row, int col) {
  return T[row * 16 + ((((col >> 2) ^ ((row >> 1) & 3)) << 2) | (col & 3))];
}

// write HP values v[0..HP) (+ optional 1.0 at column HP) of lane e into consecutive tiles
template <bool WITH_ONE>
__device__ __forceinline__ void tiles_write_vec(float* T0, int e, const float (&v)[HP]) {
  constexpr int NCH = (HP + (WITH_ONE ? 1 : 0) + 3) / 4;  // 16-B chunks to write
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = ch * 4 + i;
      t[i] = idx < HP ? v[idx < HP ? idx : 0] : ((WITH_ONE && idx == HP) ? 1.f : 0.f);
    }
    tile_write4(T0 + (ch / 4) * TILE, e, ch & 3, t[0], t[1], t[2], t[3]);
  }
}

// ---- backward kernel --------------------------------------------------------------------------
// g_z[e] = g_ea[e] * other[e] * scale * softplus'(z[e]);  back-propagate through the MLP;
// accumulate the padded parameter gradient; one partial row per block.
template <int L, int IN, bool USE_MFMA>
__global__ __launch_bounds__(PF_NET_THREADS) void k_net_backward(pf_problem P, int which) {
  if (P.state->done) return;
  extern __shared__ __align__(16) float lds[];
  constexpr int PADC = pf_pad_count(HP, L);
  const pf_net net = P.net[which];
  const pf_net onet = P.net[1 - which];
  const float* __restrict__ w = P.theta_pad + net.pad_off;
  const float* __restrict__ other = which == 0 ? P.prop_a : P.prop_e;
  const float* __restrict__ g_ea = P.g_ea;
  const int n = P.mesh.n_elems;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;

  float* tiles = lds + wv * TILES_PER_WAVE * TILE;  // this wave's tiles
  float* dzT = tiles;                               // MT tiles
  float* aT = tiles + MT * TILE;                    // NT tiles
  float* sT = tiles + (MT + NT) * TILE;             // 1 tile: cols 0..3 = x ext, col 4 = g_z
  float* wacc = lds + wv * PADC;                    // shuffle mode: per-wave accumulator (aliases)

  if (USE_MFMA) {
    for (int i = threadIdx.x; i < WAVES * TILES_PER_WAVE * TILE; i += blockDim.x) lds[i] = 0.f;
  } else {
    for (int i = threadIdx.x; i < WAVES * PADC; i += blockDim.x) lds[i] = 0.f;
  }
  __syncthreads();

  f32x4 acc1[MT];
  f32x4 acch[L > 1 ? L - 1 : 1][MT][NT];
  f32x4 acco[NT];
#pragma unroll
  for (int a = 0; a < MT; ++a) acc1[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int l = 0; l < (L > 1 ? L - 1 : 1); ++l)
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) acch[l][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < NT; ++b) acco[b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // operand coordinates of this lane inside a k-step of 4 elements
  const int orow = lane >> 4, ocol = lane & 15;

  for (int base = blockIdx.x * PF_NET_THREADS; base < n; base += gridDim.x * PF_NET_THREADS) {
    const int e = base + threadIdx.x;
    const bool live = e < n;
    Mlp<L, IN> m;
    float gz = 0.f;
    if (live) {
      load_input<IN>(m.x, P.mesh.ecent, e, P.lam);
    } else {
      m.x[0] = m.x[1] = m.x[2] = m.x[3] = 0.f;
    }
    m.forward(w);
    if (live) {
      const float oth = onet.enabled ? other[e] : onet.scale;
      float g = g_ea[e] * oth;      // mul backward of young*area   (nn_assembly.py:74)
      g = g * net.scale;            // output*scale backward         (properties.py:156)
      gz = net.positive ? g * pf_softplus_grad(m.z) : g;
    }

    const float* __restrict__ wo = w + pf_pad_wo(HP, L);
    float dz[HP];
#pragma unroll
    for (int k = 0; k < HP; ++k) {
      const float hk = m.h[L - 1][k];
      dz[k] = (gz * wo[k]) * (1.f - hk * hk);
    }

    if (USE_MFMA) {
      // ---- output layer: dWo[kk] = sum_e gz[e] * hLext[e][kk] --------------------------------
      tile_write4(sT, lane, 0, m.x[0], m.x[1], m.x[2], m.x[3]);
      tile_write4(sT, lane, 1, gz, 0.f, 0.f, 0.f);
      tiles_write_vec<true>(aT, lane, m.h[L - 1]);
      __syncthreads();
#pragma unroll 4
      for (int ks = 0; ks < 16; ++ks) {
        const int row = ks * 4 + orow;
        const float a = tile_read(sT, row, ocol);
#pragma unroll
        for (int b = 0; b < NT; ++b)
          acco[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, tile_read(aT + b * TILE, row, ocol),
                                                         acco[b], 0, 0, 0);
      }
      __syncthreads();
    } else {
      float* go = wacc + pf_pad_wo(HP, L);
#pragma unroll
      for (int k = 0; k < HP; ++k) {
        const float s = pf_wave_sum(gz * m.h[L - 1][k]);
        if (lane == 0) go[k] += s;
      }
      const float s = pf_wave_sum(gz);
      if (lane == 0) go[HP] += s;
    }

    // ---- hidden layers L..2 -------------------------------------------------------------------
#pragma unroll
    for (int l = L; l >= 2; --l) {
      const float* __restrict__ wl = w + pf_pad_wh(HP, l);
      if (USE_MFMA) {
        tiles_write_vec<false>(dzT, lane, dz);
        tiles_write_vec<true>(aT, lane, m.h[l - 2]);
        __syncthreads();
#pragma unroll 2
        for (int ks = 0; ks < 16; ++ks) {
          const int row = ks * 4 + orow;
          float av[MT], bv[NT];
#pragma unroll
          for (int a = 0; a < MT; ++a) av[a] = tile_read(dzT + a * TILE, row, ocol);
#pragma unroll
          for (int b = 0; b < NT; ++b) bv[b] = tile_read(aT + b * TILE, row, ocol);
#pragma unroll
          for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
              acch[l - 2][a][b] =
                  __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acch[l - 2][a][b], 0, 0, 0);
        }
        __syncthreads();
      } else {
        float* gl = wacc + pf_pad_wh(HP, l);
#pragma unroll
        for (int j = 0; j < HP; ++j) {
#pragma unroll
          for (int k = 0; k < HP; ++k) {
            const float s = pf_wave_sum(dz[j] * m.h[l - 2][k]);
            if (lane == 0) gl[j * (HP + 4) + k] += s;
          }
          const float s = pf_wave_sum(dz[j]);
          if (lane == 0) gl[j * (HP + 4) + HP] += s;
        }
      }
      // dh_{l-1} = W_l^T dz_l ; dz_{l-1} = dh * (1 - h^2)
      float dprev[HP];
#pragma unroll
      for (int k = 0; k < HP; ++k) {
        float accv = 0.f;
#pragma unroll
        for (int j = 0; j < HP; ++j) accv = fmaf(dz[j], wl[j * (HP + 4) + k], accv);
        const float hk = m.h[l - 2][k];
        dprev[k] = accv * (1.f - hk * hk);
      }
#pragma unroll
      for (int k = 0; k < HP; ++k) dz[k] = dprev[k];
    }

    // ---- layer 1: dW1e[j][c] = sum_e dz1[e][j] * xext[e][c] ----------------------------------
    if (USE_MFMA) {
      tiles_write_vec<false>(dzT, lane, dz);
      __syncthreads();
#pragma unroll 4
      for (int ks = 0; ks < 16; ++ks) {
        const int row = ks * 4 + orow;
        const float b = tile_read(sT, row, ocol);
#pragma unroll
        for (int a = 0; a < MT; ++a)
          acc1[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(tile_read(dzT + a * TILE, row, ocol), b,
                                                         acc1[a], 0, 0, 0);
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int j = 0; j < HP; ++j) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float s = pf_wave_sum(dz[j] * m.x[c]);
          if (lane == 0) wacc[j * 4 + c] += s;
        }
      }
    }
  }

  // ---- write-out: registers -> per-wave padded image in LDS -> fixed-order sum over waves -------
  __syncthreads();
  if (USE_MFMA) {
    for (int i = threadIdx.x; i < WAVES * PADC; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    // D layout of v_mfma_f32_16x16x4_f32: lane l, reg r -> row (l>>4)*4 + r, col l&15
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = a * 16 + orow * 4 + r;
        if (j < HP && ocol < 4) wacc[j * 4 + ocol] = acc1[a][r];
      }
#pragma unroll
    for (int l = 2; l <= L; ++l)
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = a * 16 + orow * 4 + r, kk = b * 16 + ocol;
            if (j < HP && kk <= HP) wacc[pf_pad_wh(HP, l) + j * (HP + 4) + kk] = acch[l - 2][a][b][r];
          }
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int kk = b * 16 + ocol;
      if (orow == 1 && kk <= HP) wacc[pf_pad_wo(HP, L) + kk] = acco[b][0];  // row 4 = g_z column of sT
    }
    __syncthreads();
  }
  float* __restrict__ prow = P.partials + PF_PART_WG + (size_t)blockIdx.x * P.pad_total + net.pad_off;
  for (int i = threadIdx.x; i < PADC; i += blockDim.x) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < WAVES; ++q) t += lds[q * PADC + i];
    prow[i] = t;
  }
}

template <int L, int IN>
int launch_fwd(const pf_problem* p, int which, hipStream_t s) {
  const int n = p->mesh.n_elems;
  int nb = (n + 255) / 256;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL((k_net_forward<L, IN>), dim3(nb), dim3(256), 0, s, *p, which);
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_HIP;
}

template <int L, int IN>
int launch_bwd(const pf_problem* p, int which, hipStream_t s) {
  const int nb = pf_net_blocks(p);
  constexpr int PADC = pf_pad_count(HP, L);
  if (p->wg_mode == PF_WG_MFMA) {
    constexpr int tile_floats = WAVES * TILES_PER_WAVE * TILE;
    constexpr int lds_floats = tile_floats > WAVES * PADC ? tile_floats : WAVES * PADC;
    hipLaunchKernelGGL((k_net_backward<L, IN, true>), dim3(nb), dim3(PF_NET_THREADS),
                       lds_floats * sizeof(float), s, *p, which);
  } else {
    hipLaunchKernelGGL((k_net_backward<L, IN, false>), dim3(nb), dim3(PF_NET_THREADS),
                       WAVES * PADC * sizeof(float), s, *p, which);
  }
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_HIP;
}

}  // namespace

#define PF_DISPATCH(FN)                                             \
  const pf_net& net = p->net[which];                                \
  const int L = net.n_hidden, IN = net.in_dim;                      \
  if (IN == 3) {                                                    \
    if (L == 1) return FN<1, 3>(p, which, s);                       \
    if (L == 2) return FN<2, 3>(p, which, s);                       \
    if (L == 3) return FN<3, 3>(p, which, s);                       \
  } else if (IN == 2) {                                             \
    if (L == 1) return FN<1, 2>(p, which, s);                       \
    if (L == 2) return FN<2, 2>(p, which, s);                       \
    if (L == 3) return FN<3, 2>(p, which, s);                       \
  }                                                                 \
  pf_set_error("net shape outside the compiled menu (in_dim 2|3, hidden layers 1..3)"); \
  return PF_ERR_UNSUPPORTED;

int PF_CAT(pf_launch_net_forward_, PF_HP)(const pf_problem* p, int which, hipStream_t s) {
  PF_DISPATCH(launch_fwd)
}
int PF_CAT(pf_launch_net_backward_, PF_HP)(const pf_problem* p, int which, hipStream_t s) {
  PF_DISPATCH(launch_bwd)
}
