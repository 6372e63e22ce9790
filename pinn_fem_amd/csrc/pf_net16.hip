// pf_net16.hip — MLP property FORWARD pass with the hidden layers on the f16 matrix pipe, 2-way split operands
// (compile with -DPF_HP=<4..32 step 4>).  Same contract as k_net44_forward (pf_net44.hip): per-element
// NNProperty.value (FEM/python/fem/properties.py:97-161, examples/json/generic.py:118-142).
//
// Why.  The f32 matrix cores run at the f32 vector rate and share its pipe; the f16 cores run 16x
// faster on their own pipe.  A float32 product a*w is recovered from f16 operands by splitting both into
// hi = f16(v) and lo = f16((v - hi) * 2^11): a*w = hi_a*hi_w + (lo_a*hi_w + hi_a*lo_w) * 2^-11 to 2^-22 relative,
// every f16 x f16 product being exact in the f32 accumulator.  Measured on a 20x20 layer (tools/f16split_rate.hip):
// max abs error 2.9e-7 against 3.4e-7 for the plain f32 fma chain, 1.6x the layer rate of the 4x4x1 f32 form.
//
// Layout.  v_mfma_f32_32x32x16_f16: D[m][n] = sum_k A[m][k] B[k][n]; lane l holds result column n = l%32, rows
// m = 8*(r/4) + 4*(l/32) + r%4 in registers r = 0..15; operand lanes hold row m (A) / column n (B) = l%32 and
// k = 8*(l/32)+j, j = 0..7.  ELEMENTS sit on n (32 per wave pass), hidden units on m.  Unit u = hw*UH + r lives in
// register r of half-wave hw = l/32 (UH = HP/2 units per half), and a lane fills its eight k-slots of every MFMA
// from its OWN registers — the sum over k does not care which slot carries which unit, as long as the A operand
// (the weights, rebuilt from theta_pad by every wave at kernel start) agrees — so activations never leave their lane
// from one layer to the next.  Layer 1 (inputs are coordinates up to 1e6: not f16 material) and the 1-row output unit
// stay on the f32 vector ALU with the reference's fma order; the hidden-layer bias is added after the products.
#include <type_traits>
#include <stdlib.h>
#include "pf_common.h"

#ifndef PF_HP
#error "compile with -DPF_HP=<padded width>"
#endif
#define PF_CAT2(a, b) a##b
#define PF_CAT(a, b) PF_CAT2(a, b)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int HP = PF_HP;
constexpr int UH = HP / 2;              // units per half-wave (HP is a multiple of 4: UH even, <= 16)
constexpr int NQH = (UH + 7) / 8;       // MFMAs of hi_a * hi_w
constexpr int NQL = (2 * UH + 7) / 8;   // MFMAs of lo_a * hi_w + hi_a * lo_w (one accumulator, scale 2^-11)
constexpr int NQ = NQH + NQL;

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

// k-slot t of MFMA q -> own register s of the lane that fills it (or -1) and whether the matching weight is the lo part
__host__ __device__ constexpr int slot_reg(int q, int j) {
  if (q < NQH) { const int t = 8 * q + j; return t < UH ? t : -1; }
  const int t = 8 * (q - NQH) + j;
  return t < UH ? t : (t < 2 * UH ? t - UH : -1);
}
__host__ __device__ constexpr bool slot_act_lo(int q, int j) { return q >= NQH && 8 * (q - NQH) + j < UH; }   // activation lo x weight hi
__host__ __device__ constexpr bool slot_w_lo(int q, int j) { return q >= NQH && 8 * (q - NQH) + j >= UH; }   // activation hi x weight lo

template <int L, int IN>
__global__ __launch_bounds__(256) void k_net16_forward(pf_problem P, int which) {
  const pf_net net = P.net[which];
  const float* __restrict__ w = P.theta_pad + net.pad_off;
  float* __restrict__ out = which == 0 ? P.prop_e : P.prop_a;
  const int lane = threadIdx.x & 63, hw = lane >> 5, nn = lane & 31, wvid = threadIdx.x >> 6;
  const int n = P.mesh.n_elems;

  // ---- per-lane weights ---------------------------------------------------------------------------------
  float w1[UH][IN + 1];                         // layer-1 rows of this lane's units, bias in column IN
  sfor<0, UH>([&](auto r) {
    constexpr int R = r;
    sfor<0, IN + 1>([&](auto c) { constexpr int C = c; w1[R][C] = w[(hw * UH + R) * 4 + C]; });
  });
  h8 A[L > 1 ? L - 1 : 1][NQ];                  // hidden-layer weights in MFMA A layout, split
  float bh[L > 1 ? L - 1 : 1][UH];
  {
    const int m = lane & 31;
    const int ro = 4 * (m >> 3) + (m & 3), ho = (m >> 2) & 1;
    const int uo = ro < UH ? ho * UH + ro : -1;   // output unit of tile row m
    sfor<2, L + 1>([&](auto l) {
      constexpr int LL = l;
      const float* __restrict__ wl = w + pf_pad_wh(HP, LL);
      sfor<0, NQ>([&](auto q) {
        constexpr int Q = q;
        sfor<0, 8>([&](auto j) {
          constexpr int J = j;
          constexpr int S = slot_reg(Q, J);
          float v = 0.f;
          if constexpr (S >= 0) {
            if (uo >= 0) {
              const float wv = wl[uo * (HP + 4) + hw * UH + S];
              const _Float16 whi = (_Float16)wv;
              v = slot_w_lo(Q, J) ? (float)(_Float16)((wv - (float)whi) * 2048.0f) : (float)whi;
            }
          }
          A[LL - 2][Q][J] = (_Float16)v;
        });
      });
      sfor<0, UH>([&](auto r) { constexpr int R = r; bh[LL - 2][R] = wl[(hw * UH + R) * (HP + 4) + HP]; });
    });
  }
  float wo[UH];
  sfor<0, UH>([&](auto r) { constexpr int R = r; wo[R] = w[pf_pad_wo(HP, L) + hw * UH + R]; });
  const float bo = w[pf_pad_wo(HP, L) + HP];

  const int waves = blockDim.x >> 6;
  const int stride = gridDim.x * waves * 32;
  int base = (blockIdx.x * waves + wvid) * 32;                // wave-uniform
  // Lanes past the end work on the LAST element again (same inputs, same value, a redundant store of identical
  // bits), and both half-waves store the result: no exec-masked branch is left in the loop, so the compiler's
  // vmcnt accounting is exact and the wait for the next pass's inputs does not also wait for this pass's store.
  auto fetch = [&](float (&xx)[4], int ee) {
    const int ec = ee < n ? ee : n - 1;
    xx[0] = P.lam; xx[1] = xx[2] = xx[3] = 0.f;
    if (IN == 3) {
      const float2 c = reinterpret_cast<const float2*>(P.mesh.ecent)[ec];
      xx[1] = c.x; xx[2] = c.y;
    } else {
      xx[1] = P.mesh.ecent[ec];
    }
  };
  float xn[4];
  fetch(xn, base + nn);                                       // first pass's inputs leave with the weights
  if (P.state->done || n <= 0) return;
  for (; base < n; base += stride) {
    const int e = base + nn < n ? base + nn : n - 1;
    float x[4] = {xn[0], xn[1], xn[2], xn[3]};
    fetch(xn, base + stride + nn);                            // next pass's inputs one pass ahead (clamped)
    __builtin_amdgcn_sched_barrier(0);
    // ---- layer 1 on the vector ALU, the reference's order: bias, then the inputs ascending --------------------
    float h[UH];
    sfor<0, UH>([&](auto r) {
      constexpr int R = r;
      float acc = fmaf(w1[R][IN], 1.0f, 0.f);
      sfor<0, IN>([&](auto c) { constexpr int C = c; acc = fmaf(w1[R][C], x[C], acc); });
      h[R] = pf_tanh(acc);
    });
    // ---- hidden layers on the f16 matrix pipe ----------------------------------------------------------------
    sfor<2, L + 1>([&](auto l) {
      constexpr int LL = l;
      _Float16 ah[UH], al[UH];
      sfor<0, UH>([&](auto r) {
        constexpr int R = r;
        ah[R] = (_Float16)h[R];
        al[R] = (_Float16)((h[R] - (float)ah[R]) * 2048.0f);
      });
      f32x16 dhi = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dlo = dhi;
      sfor<0, NQ>([&](auto q) {
        constexpr int Q = q;
        h8 B;
        sfor<0, 8>([&](auto j) {
          constexpr int J = j;
          constexpr int S = slot_reg(Q, J);
          if constexpr (S < 0) B[J] = (_Float16)0.0f;
          else B[J] = slot_act_lo(Q, J) ? al[S] : ah[S];
        });
        if constexpr (Q < NQH) dhi = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[LL - 2][Q], B, dhi, 0, 0, 0);
        else dlo = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[LL - 2][Q], B, dlo, 0, 0, 0);
      });
      sfor<0, UH>([&](auto r) {
        constexpr int R = r;
        h[R] = pf_tanh(fmaf(dlo[R], 1.0f / 2048.0f, dhi[R]) + bh[LL - 2][R]);
      });
    });
    // ---- output unit: own units, then the other half-wave's ----------------------------------------------------
    float part = 0.f;
    sfor<0, UH>([&](auto r) { constexpr int R = r; part = fmaf(wo[R], h[R], part); });
    const float z = (part + __shfl_xor(part, 32, 64)) + bo;
    out[e] = (net.positive ? pf_softplus(z) : z) * net.scale;
  }
}

template <int L, int IN>
int launch_fwd16(const pf_problem* p, int which, hipStream_t s) {
  const int n = p->mesh.n_elems;
  int nb = (n + 127) / 128;                    // 4 waves x 32 elements per block pass
  static const int cap = getenv("PF_FWD16_BLOCKS") ? atoi(getenv("PF_FWD16_BLOCKS")) : 512;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL((k_net16_forward<L, IN>), dim3(nb), dim3(256), 0, s, *p, which);
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_HIP;
}

}  // namespace

int PF_CAT(pf_launch_net16_forward_, PF_HP)(const pf_problem* p, int which, hipStream_t s) {
  const pf_net& net = p->net[which];
  const int L = net.n_hidden, IN = net.in_dim;
  if (IN == 3) {
    if (L == 1) return launch_fwd16<1, 3>(p, which, s);
    if (L == 2) return launch_fwd16<2, 3>(p, which, s);
    if (L == 3) return launch_fwd16<3, 3>(p, which, s);
  } else if (IN == 2) {
    if (L == 1) return launch_fwd16<1, 2>(p, which, s);
    if (L == 2) return launch_fwd16<2, 2>(p, which, s);
    if (L == 3) return launch_fwd16<3, 2>(p, which, s);
  }
  pf_set_error("net shape outside the compiled menu (in_dim 2|3, hidden layers 1..3)");
  return PF_ERR_UNSUPPORTED;
}
