// pf_net44.hip — MLP property kernels on the f32 matrix cores, 4x4x1 multi-block form
// (compile with -DPF_HP=<4..32 step 4>).  Same contract as pf_net.hip (which stays as the VALU
// cross-check engine): per-element NNProperty.value (FEM/python/fem/properties.py:97-161,
// examples/json/generic.py:118-142) and its autograd backward incl. the sum over elements of the
// parameter gradients (loss.backward(), fem/solver.py:289).
//
// Why this shape.  v_mfma_f32_4x4x1_16B_f32 computes, for each of 16 blocks of 4 lanes,
// D[i][j] += A[i]*B[j] with A taken from lane 4*blk+i, B from lane 4*blk+j and D[i][j] in
// register i of lane 4*blk+j (measured: tools/mfma_probe.hip).  With cbsz=4 every block takes A
// from block `abid`.  So with ONE ELEMENT PER LANE:
//   z[4jb+i] += W[4jb+i][k] * h[k]   is   acc_jb = mfma(A = 4 weights (broadcast), B = h[k] (own
//   register), acc_jb)
// i.e. a mat-vec per element with the activations never leaving their lane, no LDS, no padding of
// the 20-wide layers to 32 (the 16x16/32x32 forms would waste 37-61 %), and 16 different 4-weight
// vectors packed per VGPR (selected by abid), so a whole net's weights sit in <= 16 VGPRs.
// The parameter gradients  G_l = sum_e dz_l[e] (x) a_{l-1}[e]  use the same instruction without
// broadcast: each block takes one 4x4 tile of (dz (x) a) of ONE element per instruction, operands
// fetched from that element's row in LDS; the 41 tiles of a 20-20 net are covered by 3 MFMAs per
// element and accumulate in 12 registers over the whole grid-stride loop (fixed order, no atomics).
#include <type_traits>
#include <stdlib.h>
#include "pf_common.h"

#ifndef PF_HP
#error "compile with -DPF_HP=<padded width>"
#endif

#define PF_CAT2(a, b) a##b
#define PF_CAT(a, b) PF_CAT2(a, b)

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int HP = PF_HP;
constexpr int NB = HP / 4;  // blocks of 4 hidden units

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

// ---- enumeration of the 4-weight A vectors -------------------------------------------------------
template <int L>
struct WIdx {
  static constexpr int F1 = 0;                                   // (jb, c)      c in [0,4)
  static constexpr int FH = NB * 4;                              // (l, jb, k)   k in [0,HP], k==HP bias
  static constexpr int FO = FH + (L - 1) * NB * (HP + 1);        // (k)          row 0 only
  static constexpr int FWD_END = FO + HP + 1;
  static constexpr int BO = FWD_END;                             // (kb)
  static constexpr int BH = BO + NB;                             // (l, kb, j)   transposed blocks
  static constexpr int END = BH + (L - 1) * NB * HP;
  static constexpr int NV_FWD = (FWD_END + 15) / 16;
  static constexpr int NV_ALL = (END + 15) / 16;
  static constexpr int f1(int jb, int c) { return F1 + jb * 4 + c; }
  static constexpr int fh(int l, int jb, int k) { return FH + ((l - 2) * NB + jb) * (HP + 1) + k; }
  static constexpr int fo(int k) { return FO + k; }
  static constexpr int bo(int kb) { return BO + kb; }
  static constexpr int bh(int l, int kb, int j) { return BH + ((l - 2) * NB + kb) * HP + j; }

  // offset in the padded parameter image of component r of vector t, or -1 for a structural zero
  __device__ static int src(int t, int r) {
    if (t < FH) return (4 * (t / 4) + r) * 4 + (t % 4);
    if (t < FO) {
      const int q = t - FH, l2 = q / (NB * (HP + 1)), rem = q % (NB * (HP + 1));
      const int jb = rem / (HP + 1), k = rem % (HP + 1);
      return pf_pad_wh(HP, l2 + 2) + (4 * jb + r) * (HP + 4) + k;
    }
    if (t < FWD_END) return r == 0 ? pf_pad_wo(HP, L) + (t - FO) : -1;
    if (t < BH) return pf_pad_wo(HP, L) + 4 * (t - BO) + r;
    if (t < END) {
      const int q = t - BH, l2 = q / (NB * HP), rem = q % (NB * HP);
      const int kb = rem / HP, j = rem % HP;
      return pf_pad_wh(HP, l2 + 2) + j * (HP + 4) + 4 * kb + r;
    }
    return -1;
  }
};

template <int NV, int L>
__device__ __forceinline__ void load_weights(float (&wv)[NV], const float* __restrict__ w, int lane) {
  sfor<0, NV>([&](auto v) {
    constexpr int V = v;
    const int off = WIdx<L>::src(16 * V + (lane >> 2), lane & 3);
    wv[V] = off >= 0 ? w[off] : 0.f;
  });
}

#define MFMA44(T, b, acc) __builtin_amdgcn_mfma_f32_4x4x1f32(wv[(T) / 16], (b), (acc), 4, (T) % 16, 0)

template <int IN>
__device__ __forceinline__ void load_input44(float (&x)[4], const float* __restrict__ ecent, int e,
                                             float lam, bool live) {
  x[0] = x[1] = x[2] = x[3] = 0.f;
  if (!live) return;
  x[0] = lam;
  if (IN == 3) {
    const float2 c = reinterpret_cast<const float2*>(ecent)[e];
    x[1] = c.x;
    x[2] = c.y;
    x[3] = 1.f;
  } else {
    x[1] = ecent[e];
    x[2] = 1.f;
  }
}

// forward through the net; h[l][k] tanh activations, returns pre-softplus output z
// output-unit weights [HP] + bias, wave-uniform: held in scalar registers for the whole kernel
__device__ __forceinline__ void load_out_weights(float (&wo)[HP + 1], const float* __restrict__ w, int L) {
  sfor<0, HP + 1>([&](auto k) {
    constexpr int K = k;
    wo[K] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w[pf_pad_wo(HP, L) + K])));
  });
}

template <int L, int IN, int NV>
__device__ __forceinline__ float mlp_forward44(const float (&wv)[NV], const float (&wo)[HP + 1],
                                               const float (&x)[4], float (&h)[L][HP]) {
  using W = WIdx<L>;
  // PHASES.  On gfx950 an f32 MFMA and f32 VALU share one pipe and every MFMA<->VALU switch in the
  // issue stream costs tens of cycles (tools/mfma_rate.hip); left alone, the scheduler interleaves
  // the tanh of layer l one-by-one with the MFMAs of layer l+1 (222 switches per 64 elements).
  // sched_barrier(0) pins each layer to [all MFMAs][all tanh].
#define PF_PHASE() __builtin_amdgcn_sched_barrier(0)
  {
    f32x4 acc[NB];
    sfor<0, NB>([&](auto jb) {
      constexpr int JB = jb;
      acc[JB] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[JB] = MFMA44(W::f1(JB, IN), 1.0f, acc[JB]);  // bias first, like addmm(bias, x, W^T)
    });
    sfor<0, IN>([&](auto c) {
      constexpr int C = c;
      sfor<0, NB>([&](auto jb) { constexpr int JB = jb; acc[JB] = MFMA44(W::f1(JB, C), x[C], acc[JB]); });
    });
    PF_PHASE();
    sfor<0, NB>([&](auto jb) {
      constexpr int JB = jb;
      sfor<0, 4>([&](auto r) { constexpr int R = r; h[0][4 * JB + R] = pf_tanh(acc[JB][R]); });
    });
    PF_PHASE();
  }
  sfor<2, L + 1>([&](auto l) {
    constexpr int LL = l;
    // k outer / jb inner: NB independent accumulators between two updates of the same one
    f32x4 acc[NB];
    sfor<0, NB>([&](auto jb) {
      constexpr int JB = jb;
      acc[JB] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[JB] = MFMA44(W::fh(LL, JB, HP), 1.0f, acc[JB]);
    });
    sfor<0, HP>([&](auto k) {
      constexpr int K = k;
      sfor<0, NB>([&](auto jb) { constexpr int JB = jb; acc[JB] = MFMA44(W::fh(LL, JB, K), h[LL - 2][K], acc[JB]); });
    });
    PF_PHASE();
    sfor<0, NB>([&](auto jb) {
      constexpr int JB = jb;
      sfor<0, 4>([&](auto r) { constexpr int R = r; h[LL - 1][4 * JB + R] = pf_tanh(acc[JB][R]); });
    });
    PF_PHASE();
  });
  // output unit on the vector ALU: one row of weights (scalar loads, uniform address) against the own
  // activations.  Same four fma chains and the same final sum as the former MFMA form (an f32 MFMA is
  // a chain of fmaf, bitwise), at a quarter of its issue cycles: a 4x4x1 spends 4 rows on this 1-row product.
  float ao[4];
  ao[0] = wo[HP];                                      // 0 + bias*1
  ao[1] = ao[2] = ao[3] = 0.f;
  sfor<0, HP>([&](auto k) { constexpr int K = k; ao[(K + 1) % 4] = fmaf(wo[K], h[L - 1][K], ao[(K + 1) % 4]); });
  PF_PHASE();
  return (ao[0] + ao[1]) + (ao[2] + ao[3]);
}

// ---- forward kernel --------------------------------------------------------------------------------
template <int L, int IN>
__global__ __launch_bounds__(256) void k_net44_forward(pf_problem P, int which) {
  const pf_net net = P.net[which];
  const float* __restrict__ w = P.theta_pad + net.pad_off;
  float* __restrict__ out = which == 0 ? P.prop_e : P.prop_a;
  constexpr int NV = WIdx<L>::NV_FWD;
  float wv[NV];
  load_weights<NV, L>(wv, w, threadIdx.x & 63);   // issued before the stop flag is waited for
  float wo[HP + 1];
  load_out_weights(wo, w, L);
  const int n = P.mesh.n_elems;
  const int stride = gridDim.x * blockDim.x;
  int base = blockIdx.x * blockDim.x;
  // the first task's inputs leave with the weights; every later task's inputs one task ahead
  float xn[4];
  load_input44<IN>(xn, P.mesh.ecent, base + (int)threadIdx.x, P.lam, base + (int)threadIdx.x < n);
  if (P.state->done) return;
  // block-uniform trip count: every lane of a wave executes the same MFMAs (EXEC all ones)
  for (; base < n; base += stride) {
    const int e = base + threadIdx.x;
    const bool live = e < n;
    float x[4], h[L][HP];
    sfor<0, 4>([&](auto c) { constexpr int C = c; x[C] = xn[C]; });
    if (base + stride < n) load_input44<IN>(xn, P.mesh.ecent, e + stride, P.lam, e + stride < n);
    PF_PHASE();
    const float z = mlp_forward44<L, IN, NV>(wv, wo, x, h);
    if (live) out[e] = (net.positive ? pf_softplus(z) : z) * net.scale;
  }
}

// ---- LDS row of one element for the parameter-gradient tiles ------------------------------------------
template <int L>
struct Row {
  static constexpr int DZ0 = 0;                       // dz_l at (l-1)*HP, l = 1..L
  static constexpr int DZO = L * HP;                  // [g_z, 0, 0, 0]
  static constexpr int XE = L * HP + 4;               // x ext (4)
  static constexpr int HE = L * HP + 8;               // h_l ext at HE + (l-1)*(HP+4): h, 1, 0, 0, 0
  static constexpr int LEN = HE + L * (HP + 4);
  // LDS image of one pass (32 elements) is COLUMN-major: column c of element q at c*CS + q, so the 4x4
  // tile operands of 4 consecutive elements come back from ONE ds_read_b128 with an immediate offset
  // (no address arithmetic between the MFMAs).  CS = 36 = 4 (mod 32) spreads the columns over banks.
  static constexpr int CS = PF_NET44_CS;
  static constexpr int NT_L1 = NB;
  static constexpr int NT_H = (L - 1) * NB * (NB + 1);
  static constexpr int NT_O = NB + 1;
  static constexpr int NTILE = NT_L1 + NT_H + NT_O;
  static constexpr int M = (NTILE + 15) / 16;         // MFMAs per element

  // tile tau -> LDS columns of its A (dz) and B (activation) 4-vectors and its place in the padded
  // gradient image: entry (r, jq) of the tile goes to goff + r*gstride + jq  (r < rmax)
  __device__ static void decode(int tau, int& colA, int& colB, int& goff, int& gstride, int& rmax) {
    if (tau < NT_L1) {
      colA = 4 * tau; colB = XE; goff = (4 * tau) * 4; gstride = 4; rmax = 4;
    } else if (tau < NT_L1 + NT_H) {
      const int q = tau - NT_L1, l2 = q / (NB * (NB + 1)), rem = q % (NB * (NB + 1));
      const int a = rem / (NB + 1), c = rem % (NB + 1);
      colA = (l2 + 1) * HP + 4 * a;
      colB = HE + l2 * (HP + 4) + 4 * c;
      goff = pf_pad_wh(HP, l2 + 2) + (4 * a) * (HP + 4) + 4 * c; gstride = HP + 4; rmax = 4;
    } else {
      const int c = tau - NT_L1 - NT_H;
      colA = DZO; colB = HE + (L - 1) * (HP + 4) + 4 * c;
      goff = pf_pad_wo(HP, L) + 4 * c; gstride = 0; rmax = 1;
    }
  }
};

// ---- backward kernel -------------------------------------------------------------------------------------
constexpr int BW_MAX_THREADS = PF_NET44_MAX_THREADS;   // blocks run pf_net44_threads(p) <= this (runtime blockDim.x)

// Inputs of one task (64 elements per wave), fetched ONE TASK AHEAD so that no global-memory latency
// sits between two tasks of a wave (two waves per SIMD cannot hide it: measured 28 % of the wave
// lifetime in s_waitcnt before this).
template <int DIM>
struct TaskIn {
  float x[4];
  float oth, gea;
  int2 nn;
  ElemGeo g;
  float ui[2], uj[2], gi[2], gj[2];
};

// stage A of the fetch: everything addressed by the element id
template <int IN, bool GEA>
__device__ __forceinline__ void task_fetch_a(TaskIn<IN - 1>& t, const pf_problem& P, const pf_net& onet,
                                             const float* __restrict__ other, int e, bool live) {
  load_input44<IN>(t.x, P.mesh.ecent, e, P.lam, live);
  t.oth = onet.scale;
  t.gea = 0.f;
  t.nn = int2{0, 0};
  t.g = ElemGeo{0.f, 0.f, 0.f, 1.f};
  if (!live) return;
  if (onet.enabled) t.oth = other[e];
  if (GEA) {
    t.nn = reinterpret_cast<const int2*>(P.mesh.conn)[e];
    t.g = load_geo(P.mesh.egeo, e);
  } else {
    t.gea = P.g_ea[e];
  }
}

// stage B: the nodal gathers behind the connectivity (GEA only)
template <int IN, bool GEA>
__device__ __forceinline__ void task_fetch_b(TaskIn<IN - 1>& t, const pf_problem& P) {
  constexpr int DIM = IN - 1;
  if (!GEA) return;
  load_vec<DIM>(P.u, t.nn.x, t.ui);
  load_vec<DIM>(P.u, t.nn.y, t.uj);
  load_vec<DIM>(P.g_f, t.nn.x, t.gi);
  load_vec<DIM>(P.g_f, t.nn.y, t.gj);
}

// dL/d(E*A) from the fetched operands: the arithmetic of pf_elem_gea (pf_common.h), op for op
template <int DIM>
__device__ __forceinline__ float task_gea(const TaskIn<DIM>& t, int fe_mode) {
  float pu0[2], pu1[2];
  const ElemK k1 = elem_k_unit<DIM>(t.g);
  ke_rows_times<DIM>(k1, 0, t.ui, t.uj, pu0, fe_mode);
  ke_rows_times<DIM>(k1, 1, t.ui, t.uj, pu1, fe_mode);
  float gs = 0.f;
#pragma unroll
  for (int c = 0; c < DIM; ++c) gs = fmaf(t.gi[c], pu0[c], gs);
#pragma unroll
  for (int c = 0; c < DIM; ++c) gs = fmaf(t.gj[c], pu1[c], gs);
  return gs / t.g.l0;
}

// GEA: this launch also computes dL/d(E*A) per element (the element adjoint) and stores it for the
// other net's backward: saves the separate k_elem_adjoint pass in the fused iteration.
// Two waves per SIMD are resident (grid <= 1023 blocks of 2 waves on 1024 SIMDs; LDS would allow 2.5), so
// the register budget is the full 256: no scratch spills, room for the prefetched task and for
// double-buffered gradient-tile operands.
template <int L, int IN, bool GEA>
__global__ __launch_bounds__(BW_MAX_THREADS, 2) void k_net44_backward(pf_problem P, int which) {
  extern __shared__ __align__(16) float lds[];
  using W = WIdx<L>;
  using R = Row<L>;
  constexpr int DIM = IN - 1;
  constexpr int NV = W::NV_ALL;
  constexpr int PADC = pf_pad_count(HP, L);
  constexpr int M = R::M;
  const pf_net net = P.net[which];
  const pf_net onet = P.net[1 - which];
  const float* __restrict__ w = P.theta_pad + net.pad_off;
  const float* __restrict__ other = which == 0 ? P.prop_a : P.prop_e;
  const int n = P.mesh.n_elems;
  const int lane = threadIdx.x & 63, wvid = threadIdx.x >> 6;

  float wv[NV];
  load_weights<NV, L>(wv, w, lane);
  float wo[HP + 1];
  load_out_weights(wo, w, L);
  // first task's operands go out together with the weights, ahead of the stop-flag test
  const int bw_threads = blockDim.x, bw_waves = blockDim.x >> 6;
  const int stride = gridDim.x * bw_threads;
  int base = blockIdx.x * bw_threads;
  TaskIn<DIM> nxt;
  task_fetch_a<IN, GEA>(nxt, P, onet, other, base + (int)threadIdx.x, base + (int)threadIdx.x < n);
  if (P.state->done) return;
  task_fetch_b<IN, GEA>(nxt, P);

  // LDS: one pass = 32 elements, column-major (R::CS floats per column); two passes per 64-element
  // batch keep the footprint at 13.8 kB per wave for a 20-20 net
  constexpr int ROWS = 32;
  float* cols = lds + wvid * R::LEN * R::CS;
  float* mycol = cols + (lane & (ROWS - 1));       // + c*CS addresses column c of this lane's element
  // per-lane operand columns of the M gradient tiles this lane's block serves
  const float* pa[M];
  const float* pb[M];
  sfor<0, M>([&](auto m) {
    constexpr int MM = m;
    int tau = 16 * MM + (lane >> 2);
    if (tau >= R::NTILE) tau = R::NTILE - 1;   // duplicate of the last tile, never written out
    int colA, colB, goff, gstride, rmax;
    R::decode(tau, colA, colB, goff, gstride, rmax);
    pa[MM] = cols + (colA + (lane & 3)) * R::CS;
    pb[MM] = cols + (colB + (lane & 3)) * R::CS;
  });
  f32x4 accw[M];
  sfor<0, M>([&](auto m) { constexpr int MM = m; accw[MM] = f32x4{0.f, 0.f, 0.f, 0.f}; });

  // constant columns: zero paddings and the ones of the ext vectors
  for (int i = lane; i < R::LEN * R::CS; i += 64) cols[i] = 0.f;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < ROWS) {
    sfor<1, L + 1>([&](auto l) { constexpr int LL = l; mycol[(R::HE + (LL - 1) * (HP + 4) + HP) * R::CS] = 1.f; });
  }

  for (; base < n; base += stride) {
    const int e = base + threadIdx.x;
    const bool live = e < n;
    const TaskIn<DIM> cur = nxt;
    // stage A of the NEXT task's fetch goes out now; its stage B after this task's forward recompute
    const int e_nxt = e + stride;
    const bool more = base + stride < n;           // wave-uniform
    if (more) task_fetch_a<IN, GEA>(nxt, P, onet, other, e_nxt, e_nxt < n);
    PF_PHASE();
    float h[L][HP];
    const float z = mlp_forward44<L, IN, NV>(wv, wo, cur.x, h);
    if (more) task_fetch_b<IN, GEA>(nxt, P);
    PF_PHASE();
    float gz = 0.f;
    if (live) {
      float gea;
      if (GEA) {
        gea = task_gea<DIM>(cur, P.fe_mode);
        P.g_ea[e] = gea;
      } else {
        gea = cur.gea;
      }
      float g = gea * cur.oth;   // mul backward of young*area        (nn_assembly.py:74)
      g = g * net.scale;         // output*scale backward              (properties.py:156)
      gz = net.positive ? g * pf_softplus_grad(z) : g;
    }

    // ---- back-propagation, natural layout: dzs[l-1] = dL/d(pre-activation of layer l) -------------------
    float dzs[L][HP];
    PF_PHASE();
    {
      f32x4 acc[NB];
      sfor<0, NB>([&](auto kb) {
        constexpr int KB = kb;
        acc[KB] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[KB] = MFMA44(W::bo(KB), gz, acc[KB]);              // dh_L = Wo^T g_z
      });
      PF_PHASE();
      sfor<0, NB>([&](auto kb) {
        constexpr int KB = kb;
        sfor<0, 4>([&](auto r) {
          constexpr int RR = r;
          const float hk = h[L - 1][4 * KB + RR];
          dzs[L - 1][4 * KB + RR] = acc[KB][RR] * fmaf(-hk, hk, 1.f);   // tanh backward
        });
      });
      PF_PHASE();
    }
    sfor<0, L - 1>([&](auto s) {
      constexpr int LL = L - s;                               // L .. 2
      f32x4 acc[NB];
      sfor<0, NB>([&](auto kb) { constexpr int KB = kb; acc[KB] = f32x4{0.f, 0.f, 0.f, 0.f}; });
      sfor<0, HP>([&](auto j) {
        constexpr int J = j;
        sfor<0, NB>([&](auto kb) { constexpr int KB = kb; acc[KB] = MFMA44(W::bh(LL, KB, J), dzs[LL - 1][J], acc[KB]); });
      });
      PF_PHASE();
      sfor<0, NB>([&](auto kb) {
        constexpr int KB = kb;
        sfor<0, 4>([&](auto r) {
          constexpr int RR = r;
          const float hk = h[LL - 2][4 * KB + RR];
          dzs[LL - 2][4 * KB + RR] = acc[KB][RR] * fmaf(-hk, hk, 1.f);
        });
      });
      PF_PHASE();
    });

    // ---- parameter-gradient tiles: two passes of 32 elements; in each pass every step takes ONE
    // element's row and all 64 lanes cooperate on its 4x4 tiles ------------------------------------------
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      if ((lane >> 5) == hf) {
        sfor<0, 4>([&](auto c) { constexpr int C = c; mycol[(R::XE + C) * R::CS] = cur.x[C]; });
        mycol[R::DZO * R::CS] = gz;
        sfor<1, L + 1>([&](auto l) {
          constexpr int LL = l;
          sfor<0, HP>([&](auto k) {
            constexpr int K = k;
            mycol[((LL - 1) * HP + K) * R::CS] = dzs[LL - 1][K];
            mycol[(R::HE + (LL - 1) * (HP + 4) + K) * R::CS] = h[LL - 1][K];
          });
        });
      }
      // columns are private to the wave and LDS executes a wave's accesses in order: only the
      // compiler has to be told not to move the reads above the writes
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      PF_PHASE();
      // operands of step q4+1 are read while the 4*M MFMAs of step q4 issue (two register sets)
      float4 av[2][M], bv[2][M];
      sfor<0, M>([&](auto m) {
        constexpr int MM = m;
        av[0][MM] = *reinterpret_cast<const float4*>(pa[MM]);
        bv[0][MM] = *reinterpret_cast<const float4*>(pb[MM]);
      });
      sfor<0, ROWS / 4>([&](auto q) {
        constexpr int Q4 = q;
        constexpr int CB = Q4 & 1, NBUF = 1 - CB;
        if constexpr (Q4 + 1 < ROWS / 4) {
          sfor<0, M>([&](auto m) {
            constexpr int MM = m;
            av[NBUF][MM] = *reinterpret_cast<const float4*>(pa[MM] + 4 * (Q4 + 1));
            bv[NBUF][MM] = *reinterpret_cast<const float4*>(pb[MM] + 4 * (Q4 + 1));
          });
        }
        PF_PHASE();
        sfor<0, M>([&](auto m) {
          constexpr int MM = m;
          accw[MM] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[CB][MM].x, bv[CB][MM].x, accw[MM], 0, 0, 0);
        });
        sfor<0, M>([&](auto m) {
          constexpr int MM = m;
          accw[MM] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[CB][MM].y, bv[CB][MM].y, accw[MM], 0, 0, 0);
        });
        sfor<0, M>([&](auto m) {
          constexpr int MM = m;
          accw[MM] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[CB][MM].z, bv[CB][MM].z, accw[MM], 0, 0, 0);
        });
        sfor<0, M>([&](auto m) {
          constexpr int MM = m;
          accw[MM] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[CB][MM].w, bv[CB][MM].w, accw[MM], 0, 0, 0);
        });
        PF_PHASE();
      });
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }

  // ---- write-out: tiles -> per-wave padded image in LDS -> fixed-order sum over waves -> partial row -----------
  __syncthreads();
  for (int i = threadIdx.x; i < bw_waves * PADC; i += bw_threads) lds[i] = 0.f;
  __syncthreads();
  float* wimg = lds + wvid * PADC;
  sfor<0, M>([&](auto m) {
    constexpr int MM = m;
    const int tau = 16 * MM + (lane >> 2);
    if (tau < R::NTILE) {
      int colA, colB, goff, gstride, rmax;
      R::decode(tau, colA, colB, goff, gstride, rmax);
      sfor<0, 4>([&](auto r) {
        constexpr int RR = r;
        if (RR < rmax) wimg[goff + RR * gstride + (lane & 3)] = accw[MM][RR];
      });
    }
  });
  __syncthreads();
  float* __restrict__ prow = P.partials + PF_PART_WG + (size_t)blockIdx.x * P.pad_total + net.pad_off;
  for (int i = threadIdx.x; i < PADC; i += bw_threads) {
    float t = 0.f;
    for (int q = 0; q < bw_waves; ++q) t += lds[q * PADC + i];
    prow[i] = t;
  }
}

template <int L, int IN>
int launch_fwd(const pf_problem* p, int which, hipStream_t s) {
  const int n = p->mesh.n_elems;
  int nb = (n + 255) / 256;
  // EXPERIMENT knobs: PF_FWD_BLOCKS caps the grid, PF_FWD_LDS (bytes of dummy dynamic LDS) limits
  // the blocks resident per CU
  static const int cap = getenv("PF_FWD_BLOCKS") ? atoi(getenv("PF_FWD_BLOCKS")) : 2048;
  static const int ldsb = getenv("PF_FWD_LDS") ? atoi(getenv("PF_FWD_LDS")) : 0;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL((k_net44_forward<L, IN>), dim3(nb), dim3(256), ldsb, s, *p, which);
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_HIP;
}

template <int L, int IN, bool GEA>
int launch_bwd_t(const pf_problem* p, int which, hipStream_t s) {
  const int nb = pf_net_blocks(p);
  const int threads = pf_net44_threads(p), waves = threads / 64;
  constexpr int PADC = pf_pad_count(HP, L);
  static_assert(pf_net44_row_len(HP, L) == Row<L>::LEN, "host LDS sizing out of step with Row<L>");
  const int row_floats = waves * Row<L>::LEN * Row<L>::CS;
  const int lds_floats = row_floats > waves * PADC ? row_floats : waves * PADC;
  if ((size_t)lds_floats * 4 > 160u * 1024u) {
    pf_set_error("net too large for the gradient-tile LDS image");
    return PF_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL((k_net44_backward<L, IN, GEA>), dim3(nb), dim3(threads), lds_floats * sizeof(float), s,
                     *p, which);
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_HIP;
}
template <int L, int IN>
int launch_bwd(const pf_problem* p, int which, hipStream_t s) { return launch_bwd_t<L, IN, false>(p, which, s); }
template <int L, int IN>
int launch_bwd_gea(const pf_problem* p, int which, hipStream_t s) { return launch_bwd_t<L, IN, true>(p, which, s); }

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

int PF_CAT(pf_launch_net44_forward_, PF_HP)(const pf_problem* p, int which, hipStream_t s) {
  PF_DISPATCH(launch_fwd)
}
int PF_CAT(pf_launch_net44_backward_, PF_HP)(const pf_problem* p, int which, hipStream_t s) {
  PF_DISPATCH(launch_bwd)
}
int PF_CAT(pf_launch_net44_backward_gea_, PF_HP)(const pf_problem* p, int which, hipStream_t s) {
  PF_DISPATCH(launch_bwd_gea)
}
