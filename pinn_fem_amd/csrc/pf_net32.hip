// pf_net32.hip — MLP property kernels on the f16 matrix cores with 2-way split operands (MFMA32 engine;
// compile with -DPF_NR=<registers per lane>: nets of width <= 2*PF_NR, PF_NR <= 15).
// Same contract as pf_net44.hip: per-element NNProperty.value (FEM/python/fem/properties.py:97-161,
// examples/json/generic.py:118-142) and its autograd backward incl. the sum over elements of the parameter
// gradients (loss.backward(), fem/solver.py:289).
//
// Why.  On gfx950 the f32 matrix cores run at the f32 vector rate and share its issue pipe; the
// f16 cores are 16x faster and run beside the vector ALU.  A float32 product is recovered from f16 operands by
// splitting both factors into hi = f16(v) and lo = f16(v - hi): v w = hi_v hi_w + hi_v lo_w + lo_v hi_w to 2^-22
// relative, every f16 x f16 product being exact in the f32 accumulator (three MFMAs per product).  Operands are
// pre-scaled by powers of two (pf_net32.h) so that the lo parts stay normal f16 numbers.
//
// Layout.  v_mfma_f32_32x32x16_f16, ELEMENTS on the tile columns (lane & 31), hidden units on the rows: a wave
// works on 64 elements = two tiles; lane (c, h = lane>>5) holds, for element c of each tile, the units 2r+h in
// accumulator register r (pf_net32.h).  Consequences:
//   * tanh runs on PF_NR registers per lane and tile — no lane computes a padding unit (the 32x32 tile has 32 rows,
//     a 20-wide layer fills 10 registers of both half-waves);
//   * the result registers of one layer are, converted to f16, the B operand of the next (forward) and of the
//     transposed product (back-propagation): activations never move between lanes;
//   * per-element scalars (output unit, softplus, element adjoint, loads, stores) sit one element per lane:
//     element = task base + lane, the two half-waves exchange what the other needs with v_permlane32_swap;
//   * the parameter-gradient products sum over ELEMENTS, i.e. over the lane index: their operands go once through
//     LDS ([element][unit] images written 16 B per lane, conflict free) and come back transposed by
//     ds_read_b64_tr_b16.  Gradient operands are scaled per task by an exact power of two from the task's largest
//     |g_z| and the weight bound of the image header, so nothing overflows f16 and small gradients keep their bits.
#include <type_traits>
#include <stdlib.h>
#include "pf_net32.h"
#include "pf_node.h"

#ifndef PF_NR
#error "compile with -DPF_NR=<registers per lane>"
#endif
// PF_PREC 0 (default): 2-way split f16 operands, three products: float32-grade results.
// PF_PREC 1: plain bf16 operands (round to nearest), ONE product, f32 accumulate; everything else (layer 1, tanh,
//            output unit, element algebra, sums) stays float32.  The reduced-precision variant of BASELINE.json
//            configs[4] ("fp32 vs bf16 residual tolerance study"), selected by pf_problem.mlp_dtype.
#ifndef PF_PREC
#define PF_PREC 0
#endif
// Timing-experiment knobs (PF_N32_DBG bits: skip gradient tiles / back-propagation / transcendentals / MFMAs / barriers,
// time stamps) exist only in builds with -DPF_N32_DBG_ENABLE=1 (PINNFEM_N32_DBG=1 python -m pinn_fem_amd.build): as
// run-time branches in the task loop they split it into basic blocks, and the s_waitcnt the compiler must then place at
// every join made the product kernels wait for LDS reads they did not need yet.
#ifndef PF_N32_DBG_ENABLE
#define PF_N32_DBG_ENABLE 0
#endif
// Experiment knobs of the backward's block shape (build-time; pinn_fem_amd/build.py: PINNFEM_BW2_NOPAIR=1 sets them for the
// fused two-net translation units): PF_BW_PAIR 0 = recompute tile by tile (fewer registers, see bw_pair),
// PF_BW_MAX_THREADS = cap on the block size.
// PF_N32_PIPE 1: in the two-tile hidden layer, tile 1's matrix products carry tile 0's first tanh stage in their gaps
// (MI355X_MICROARCH.md / tools/pipe_probe.hip: an MFMA gap hides 24 cycles of vector issue): same values, other order.
#ifndef PF_N32_PIPE
#define PF_N32_PIPE 1
#endif
#ifndef PF_BW_PAIR
#define PF_BW_PAIR 1
#endif
#ifndef PF_BW_MAX_THREADS
#define PF_BW_MAX_THREADS 1024
#endif
#define PF_CAT2(a, b) a##b
#define PF_CAT(a, b) PF_CAT2(a, b)

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

namespace {

constexpr bool BF = PF_PREC == 1;

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

__device__ __forceinline__ f32x16 zero16() {
  return f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ h8 as_h8(unsigned a, unsigned b, unsigned c, unsigned d) {
  return __builtin_bit_cast(h8, u32x4{a, b, c, d});
}

// two values -> packed hi and lo f16 pairs: hi = rtz(a), lo = f16(a - hi) by one mixed-precision fma each
// (v_fma_mixlo/hi_f16 take the f32 value and the f16 half of `hi` directly: 3 instructions per pair against 8
// for convert-back, subtract, convert; checked by tools/mix_probe.hip)
__device__ __forceinline__ void split_pair(float a0, float a1, unsigned& hi, unsigned& lo) {
  if constexpr (BF) {   // plain bf16, round to nearest even (v_cvt_pk_bf16_f32); no lo part
    hi = __builtin_bit_cast(unsigned, bf2{(__bf16)a0, (__bf16)a1});
    lo = 0u;
    return;
  }
  hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a0, a1));
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a0), "v"(hi));
  asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(a1), "v"(hi));
  lo = l;
}

// acc += (Ahi + Alo)(Bhi + Blo) without the lo*lo term; small terms first
__device__ __forceinline__ f32x16 mfma3(f32x16 acc, h8 ahi, h8 alo, h8 bhi, h8 blo) {
  if constexpr (BF)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, ahi), __builtin_bit_cast(bf8, bhi), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc, 0, 0, 0);
  return acc;
}

// the K-th of the (up to) three products of mfma3, on its own (the pipelined hidden layer places vector work between them)
template <int K>
__device__ __forceinline__ f32x16 mfma3_part(f32x16 acc, h8 ahi, h8 alo, h8 bhi, h8 blo) {
  if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, ahi), __builtin_bit_cast(bf8, bhi), acc, 0, 0, 0);
  if constexpr (K == 0) return __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc, 0, 0, 0);
  if constexpr (K == 1) return __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc, 0, 0, 0);
}

// v_permlane32_swap_b32 vdst, src: lanes 32-63 of vdst swap with lanes 0-31 of src.  Inline asm: the clang
// builtin of this toolchain returns element 0 for BOTH results (r[1] == r[0]; checked in the emitted IR).  The
// s_nop covers the VALU-write -> permlane-read hazard the compiler does not pad inside asm.
__device__ __forceinline__ void permlane32_swap(unsigned& vdst, unsigned& src) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(vdst), "+v"(src));
}

// both half-waves get the value lane 32t+c holds, for t = 0 and 1 (c = own lane & 31)
__device__ __forceinline__ void both_tiles(float own, float& t0, float& t1) {
  unsigned a = __builtin_bit_cast(unsigned, own), b = a;
  permlane32_swap(a, b);      // a: lower own | lower's ; b: upper's | upper own
  t0 = __builtin_bit_cast(float, a);
  t1 = __builtin_bit_cast(float, b);
}

// p0 / p1: this lane's partial sums for tile 0 / tile 1; returns the complete sum of the lane's OWN element
// (lower half-wave: tile 0, upper: tile 1)
__device__ __forceinline__ float own_total(float p0, float p1) {
  unsigned a = __builtin_bit_cast(unsigned, p0), b = __builtin_bit_cast(unsigned, p1);
  permlane32_swap(a, b);      // lower: own p0, upper's p0 ; upper: lower's p1, own p1
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

// wave maximum of a non-negative value, uniform result; DPP row shifts and broadcasts (no LDS round trips):
// after the four row_shr steps lane 15 of every row holds the row maximum, row_bcast15 / row_bcast31 carry it on,
// lane 63 ends with the maximum of the wave (invalid source lanes read 0 = the identity for non-negative values)
__device__ __forceinline__ float wave_max(float v) {
  int x = __builtin_bit_cast(int, v);
#define PF_DPP_MAX(CTRL, ROWS)                                                                       \
  x = __builtin_bit_cast(int, fmaxf(__builtin_bit_cast(float, x),                                    \
                                    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, CTRL, ROWS, 0xf, false))))
  PF_DPP_MAX(0x111, 0xf);   // row_shr:1
  PF_DPP_MAX(0x112, 0xf);   // row_shr:2
  PF_DPP_MAX(0x114, 0xf);   // row_shr:4
  PF_DPP_MAX(0x118, 0xf);   // row_shr:8
  PF_DPP_MAX(0x142, 0xa);   // row_bcast15 -> rows 1, 3
  PF_DPP_MAX(0x143, 0xc);   // row_bcast31 -> rows 2, 3
#undef PF_DPP_MAX
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 63));
}

template <int IN>
__device__ __forceinline__ float layer1_row(const float4& w, const float (&x)[3]) {
  // layer 1 on the vector ALU in float, the reference's order: bias, then the inputs ascending
  float c;                 // (the bias itself: fma(b, 1, 0) would only turn a -0.0 bias into +0.0)
  if constexpr (IN == 3) {
    c = fmaf(w.x, x[0], w.w);
    c = fmaf(w.y, x[1], c);
    c = fmaf(w.z, x[2], c);
  } else {
    c = fmaf(w.x, x[0], w.z);
    c = fmaf(w.y, x[1], c);
  }
  return c;
}

template <int IN>
__device__ __forceinline__ void load_x(float (&x)[3], const pf_problem& P, int e) {
  x[0] = P.lam;
  if (IN == 3) {
    const float2 c = reinterpret_cast<const float2*>(P.mesh.ecent)[e];
    x[1] = c.x;
    x[2] = c.y;
  } else {
    x[1] = P.mesh.ecent[e];
    x[2] = 0.f;
  }
}

__device__ __forceinline__ void copy_image(unsigned char* dst, const unsigned char* __restrict__ src, int bytes) {
  const uint4* __restrict__ s = reinterpret_cast<const uint4*>(src);
  uint4* d = reinterpret_cast<uint4*>(dst);
  for (int i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
}

// two images at once, every load of both in flight before the first LDS store (copy_image twice is up to four dependent
// round trips for a 512-thread block: each call loops load -> wait -> store, and an image is 562 x 16 bytes)
__device__ __forceinline__ void copy_images2(unsigned char* dst0, const unsigned char* __restrict__ src0, unsigned char* dst1,
                                             const unsigned char* __restrict__ src1, int bytes) {
  const uint4* __restrict__ s0 = reinterpret_cast<const uint4*>(src0);
  const uint4* __restrict__ s1 = reinterpret_cast<const uint4*>(src1);
  uint4* d0 = reinterpret_cast<uint4*>(dst0);
  uint4* d1 = reinterpret_cast<uint4*>(dst1);
  const int n16 = bytes / 16;
  for (int i0 = threadIdx.x; i0 < n16; i0 += 2 * blockDim.x) {
    const int i1 = i0 + blockDim.x;
    const bool two = i1 < n16;
    const uint4 a0 = s0[i0], b0 = s1[i0];
    const uint4 a1 = s0[two ? i1 : i0], b1 = s1[two ? i1 : i0];
    d0[i0] = a0;
    d1[i0] = b0;
    if (two) { d0[i1] = a1; d1[i1] = b1; }
  }
}

template <int DIM>
struct TaskIn {
  float x[3];
  float oth, own, gea;
  int2 nn;
  ElemGeo g;
  float ui[2], uj[2], gi[2], gj[2];
};

// 1 - e^(-s) for s >= 0 (softplus'(z) = sigmoid(z) from s = softplus(z)) to ~3 ulp in 11 instructions: the hardware exp2
// where 1 - t does not cancel (s > 0.25: t < 0.78), the Taylor polynomial below it (next term s^8 / 8! < 4e-10 relative).
// (libm's expm1f: ~25 instructions with a division, once per element and net in the backward's task head.)
__device__ __forceinline__ float one_minus_exp_neg(float s) {
  const float big = 1.0f - __builtin_amdgcn_exp2f(s * -1.4426950408889634f);
  float p = 1.0f / 5040.0f;
  p = fmaf(p, s, -1.0f / 720.0f);
  p = fmaf(p, s, 1.0f / 120.0f);
  p = fmaf(p, s, -1.0f / 24.0f);
  p = fmaf(p, s, 1.0f / 6.0f);
  p = fmaf(p, s, -0.5f);
  p = fmaf(p, s, 1.0f);
  return s > 0.25f ? big : p * s;
}

// with_nn: also load the element's node ids (the main loop has them already: they are fetched TWO tasks ahead, so that
// the nodal gathers of task_fetch_b, which need them as addresses, never wait for a load issued in the same iteration)
template <int IN, bool GEA>
__device__ __forceinline__ void task_fetch_a(TaskIn<IN - 1>& t, const pf_problem& P, const pf_net& onet,
                                             const float* __restrict__ other, const float* __restrict__ mine, int e,
                                             bool with_nn = true) {
  load_x<IN>(t.x, P, e);
  t.oth = onet.enabled ? other[e] : onet.scale;
  t.own = mine[e];
  t.gea = 0.f;
  if (GEA) {
    if (with_nn) t.nn = reinterpret_cast<const int2*>(P.mesh.conn)[e];
    t.g = load_geo(P.mesh.egeo, e);
  } else {
    t.nn = int2{0, 0};
    t.g = ElemGeo{0.f, 0.f, 0.f, 1.f};
    t.gea = P.g_ea[e];
  }
}

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

// entries of ke = s*pattern of one element -> the record the node kernels read (pf_common.h: ElemK, load_k)
template <int DIM>
__device__ __forceinline__ void store_k(float* __restrict__ elem_k, int e, const ElemGeo& g, float s) {
  if (DIM == 2) {
    float* __restrict__ r = elem_k + 3 * (size_t)e;
    r[0] = s * g.c2;
    r[1] = s * g.cs;
    r[2] = s * g.s2;
  } else {
    elem_k[e] = s;
  }
}

// ---- everything that depends on the register bucket NR (registers per lane that carry real units: nets of width <= 2 NR)
// lives in one class template, so that one kernel can run nets of two different buckets (the fused E + A launches)
template <int NR_>
struct Eng {
  static constexpr int NR = NR_;            // accumulator registers per lane that carry real units
  static constexpr int KS = NR > 8 ? 2 : 1;  // k-steps of 16 units
  static constexpr int NPK = 4 * KS;         // packed f16 pairs per operand set (pairs >= (NR+1)/2 are zero)
  static constexpr int NPR = (NR + 1) / 2;   // pairs that carry data
  static_assert(NR >= 1 && NR <= 15, "register bucket out of range");

  // ---- activations of one tile kept for the backward pass -------------------------------------------------
  template <int L, bool BWD>
  struct TileAct {
    unsigned hi[L][NPK], lo[L][NPK];   // a'_l = KA tanh(z_l) as packed f16 pairs (pair q = registers 2q, 2q+1)
    float t[BWD ? L : 1][NR];          // r (1 - r) with tanh = 1 - 2 r: (1 - tanh^2) / 4
    float aL[NR];                      // a'_L in float (output unit and its gradient)
  };

  // Scheduling fence.  Left alone, hipcc serialises each unit's exp -> add -> rcp -> fma chain (the transcendental
  // results are needed a few cycles after issue, so the wave stalls on every step); the stages below issue one kind
  // of instruction for all units of both tiles back to back, which hides those latencies inside the wave.
#define PF_STAGE() __builtin_amdgcn_sched_barrier(0)

  // tanh of the NR pre-activations z[r] * cz of BOTH tiles; writes the packed pairs (and t, aL) of layer LL
  template <int L, bool BWD, int LL>
  // go (backward pass, last hidden layer): the output unit's weight-gradient row takes g0 a'_L (tile 0), then g1 a'_L (tile
  // 1) right here, where a'_L is formed — kept for the back-propagation's start it had to be rebuilt from r at the end of
  // every task (twenty fma) and held twenty registers over both tiles' back-propagation.
  static __device__ __forceinline__ void activate2(TileAct<L, BWD>& A0, TileAct<L, BWD>& A1, const float (&z0)[NR],
                                            const float (&z1)[NR], float cz, int dbg = 0, float* go = nullptr,
                                            float g0 = 0.f, float g1 = 0.f) {
    float e0[NR], e1[NR];
    PF_STAGE();
    if (dbg & 4) {   // timing experiment: no transcendentals
      sfor<0, NR>([&](auto r) { constexpr int R = r; e0[R] = z0[R] * cz; e1[R] = z1[R] * cz; });
    } else {
    sfor<0, NR>([&](auto r) {
      constexpr int R = r;
      e0[R] = __builtin_amdgcn_exp2f(z0[R] * cz);
      e1[R] = __builtin_amdgcn_exp2f(z1[R] * cz);
    });
    PF_STAGE();
    sfor<0, NR>([&](auto r) {
      constexpr int R = r;
      e0[R] = __builtin_amdgcn_rcpf(e0[R] + 1.0f);
      e1[R] = __builtin_amdgcn_rcpf(e1[R] + 1.0f);
    });
    }
    PF_STAGE();
    float a0[2 * NPR], a1[2 * NPR];
    sfor<0, NR>([&](auto r) {
      constexpr int R = r;
      a0[R] = fmaf(-2.0f * PF_N32_KA, e0[R], PF_N32_KA);
      a1[R] = fmaf(-2.0f * PF_N32_KA, e1[R], PF_N32_KA);
      if constexpr (BWD) {
        A0.t[LL - 1][R] = fmaf(-e0[R], e0[R], e0[R]);
        A1.t[LL - 1][R] = fmaf(-e1[R], e1[R], e1[R]);
      }
      if constexpr (LL == L) {
        if (BWD && go) { go[R] = fmaf(g0, a0[R], go[R]); go[R] = fmaf(g1, a1[R], go[R]); }
        else { A0.aL[R] = a0[R]; A1.aL[R] = a1[R]; }
      }
    });
    if constexpr (NR & 1) { a0[NR] = 0.f; a1[NR] = 0.f; }
    sfor<0, NPK>([&](auto q) {
      constexpr int Q = q;
      if constexpr (Q < NPR) {
        split_pair(a0[2 * Q], a0[2 * Q + 1], A0.hi[LL - 1][Q], A0.lo[LL - 1][Q]);
        split_pair(a1[2 * Q], a1[2 * Q + 1], A1.hi[LL - 1][Q], A1.lo[LL - 1][Q]);
      } else {
        A0.hi[LL - 1][Q] = 0u; A0.lo[LL - 1][Q] = 0u;
        A1.hi[LL - 1][Q] = 0u; A1.lo[LL - 1][Q] = 0u;
      }
    });
    PF_STAGE();
  }

  // The same activation when tile 0's first stage (e0 = exp2(z0 cz)) has already been issued between tile 1's matrix
  // products: the stages of the two tiles are staggered by one, every stage still holds ~2 NR independent instructions.
  template <int L, bool BWD, int LL>
  static __device__ __forceinline__ void activate2_staggered(TileAct<L, BWD>& A0, TileAct<L, BWD>& A1, float (&e0)[NR],
                                                             const float (&z1)[NR], float cz, float* go = nullptr,
                                                             float g0 = 0.f, float g1 = 0.f) {
    float e1[NR];
    sfor<0, NR>([&](auto r) {
      constexpr int R = r;
      e1[R] = __builtin_amdgcn_exp2f(z1[R] * cz);
      e0[R] = __builtin_amdgcn_rcpf(e0[R] + 1.0f);
    });
    PF_STAGE();
    float a0[2 * NPR], a1[2 * NPR];
    sfor<0, NR>([&](auto r) {
      constexpr int R = r;
      e1[R] = __builtin_amdgcn_rcpf(e1[R] + 1.0f);
      a0[R] = fmaf(-2.0f * PF_N32_KA, e0[R], PF_N32_KA);
      if constexpr (BWD) A0.t[LL - 1][R] = fmaf(-e0[R], e0[R], e0[R]);
      if constexpr (LL == L) {
        if (BWD && go) go[R] = fmaf(g0, a0[R], go[R]);
        else A0.aL[R] = a0[R];
      }
    });
    if constexpr (NR & 1) { a0[NR] = 0.f; a1[NR] = 0.f; }
    PF_STAGE();
    sfor<0, NR>([&](auto r) {
      constexpr int R = r;
      a1[R] = fmaf(-2.0f * PF_N32_KA, e1[R], PF_N32_KA);
      if constexpr (BWD) A1.t[LL - 1][R] = fmaf(-e1[R], e1[R], e1[R]);
      if constexpr (LL == L) {
        if (BWD && go) go[R] = fmaf(g1, a1[R], go[R]);
        else A1.aL[R] = a1[R];
      }
    });
    sfor<0, NPK>([&](auto q) {
      constexpr int Q = q;
      if constexpr (Q < NPR) split_pair(a0[2 * Q], a0[2 * Q + 1], A0.hi[LL - 1][Q], A0.lo[LL - 1][Q]);
      else { A0.hi[LL - 1][Q] = 0u; A0.lo[LL - 1][Q] = 0u; }
    });
    PF_STAGE();
    sfor<0, NPK>([&](auto q) {
      constexpr int Q = q;
      if constexpr (Q < NPR) split_pair(a1[2 * Q], a1[2 * Q + 1], A1.hi[LL - 1][Q], A1.lo[LL - 1][Q]);
      else { A1.hi[LL - 1][Q] = 0u; A1.lo[LL - 1][Q] = 0u; }
    });
    PF_STAGE();
  }

  // weights of one hidden layer as a wave reads them from the LDS image: bias vector (initial accumulator) and the
  // split A operands of both k-steps.  Loaded one phase AHEAD of their use (before the tanh stages of the previous
  // layer), so that the LDS latency hides behind the transcendentals instead of stalling the matrix products.
  struct LayerW {
    float4 b[4];
    h8 ahi[KS], alo[KS];
  };
  template <int LL, bool BWDOP>
  static __device__ __forceinline__ void load_layer(const unsigned char* __restrict__ img, int lane, LayerW& w) {
    const int h = lane >> 5;
    if constexpr (!BWDOP) {
      const float4* __restrict__ bias = reinterpret_cast<const float4*>(img + pf_n32_off_bias(LL)) + h * 4;
      w.b[0] = bias[0]; w.b[1] = bias[1]; w.b[2] = bias[2]; w.b[3] = bias[3];
    }
    const h8* __restrict__ af = reinterpret_cast<const h8*>(img + (BWDOP ? pf_n32_off_ab(LL) : pf_n32_off_af(LL))) + lane;
    sfor<0, KS>([&](auto s) { constexpr int S = s; w.ahi[S] = af[(0 * 2 + S) * 64]; w.alo[S] = af[(1 * 2 + S) * 64]; });
  }
  static __device__ __forceinline__ f32x16 bias_acc(const LayerW& w) {
    return f32x16{w.b[0].x, w.b[0].y, w.b[0].z, w.b[0].w, w.b[1].x, w.b[1].y, w.b[1].z, w.b[1].w,
                  w.b[2].x, w.b[2].y, w.b[2].z, w.b[2].w, w.b[3].x, w.b[3].y, w.b[3].z, w.b[3].w};
  }
  // the output unit's weights of this lane's units: wo[2r+h], r < NR
  static __device__ __forceinline__ void load_wo(const unsigned char* __restrict__ img, int lane, float (&wv)[16]) {
    const float4* __restrict__ wo = reinterpret_cast<const float4*>(img + pf_n32_off_wo()) + (lane >> 5) * 4;
    sfor<0, (NR + 3) / 4>([&](auto q) {
      constexpr int Q = q;
      const float4 w = wo[Q];
      wv[4 * Q] = w.x; wv[4 * Q + 1] = w.y; wv[4 * Q + 2] = w.z; wv[4 * Q + 3] = w.w;
    });
  }
  // forward through the hidden layers for BOTH tiles of a task; x0 / x1 = (load factor, coordinates) of the tile's
  // element on this lane's column.  p0 / p1: this lane's partial sums of the output unit, KA * sum_r wo[2r+h] a_L[2r+h].
  template <int L, int IN, bool BWD>
  static __device__ __forceinline__ void forward_tiles(const unsigned char* __restrict__ img, int lane, const float (&x0)[3],
                                                const float (&x1)[3], TileAct<L, BWD>& A0, TileAct<L, BWD>& A1,
                                                float& p0, float& p1, int dbg = 0, int grp = -1, float* go = nullptr,
                                                float g0 = 0.f, float g1 = 0.f) {
    const int h = lane >> 5;
    constexpr float C2 = 2.8853900817779268f;   // 2 log2(e)
    LayerW wl[L > 1 ? L - 1 : 1];
    float wv[16];
    {
      const float4* __restrict__ w1 = reinterpret_cast<const float4*>(img + pf_n32_off_w1());
      float4 w1v[NR];
      sfor<0, NR>([&](auto r) { constexpr int R = r; w1v[R] = w1[R * 2 + h]; });
      if constexpr (L >= 2) load_layer<2, false>(img, lane, wl[0]);
      else load_wo(img, lane, wv);
      float z0[NR], z1[NR];
      sfor<0, NR>([&](auto r) {
        constexpr int R = r;
        z0[R] = layer1_row<IN>(w1v[R], x0);
        z1[R] = layer1_row<IN>(w1v[R], x1);
      });
      activate2<L, BWD, 1>(A0, A1, z0, z1, C2, dbg, go, g0, g1);
    }
    if (grp == 1) __builtin_amdgcn_s_barrier();      // lockstep point of wave group 1 (see k_net32_forward)
    // hidden layers 2..L on the matrix cores: z' = KA KW z, bias as the initial accumulator
    sfor<2, L + 1>([&](auto l) {
      constexpr int LL = l;
      const LayerW& w = wl[LL - 2];
      f32x16 acc0 = bias_acc(w), acc1 = acc0;
      constexpr float CZ = C2 / (PF_N32_KA * PF_N32_KW);
      if constexpr (PF_N32_PIPE && !PF_N32_DBG_ENABLE) {
        // tile 0's products, then tile 1's with tile 0's mul + exp2 stage spread over their gaps (each gap: <= PER
        // registers = PER x (4 + 8) cycles of issue against the 24 a gap hides); sched_barriers pin the order
        sfor<0, KS>([&](auto s) {
          constexpr int S = s;
          acc0 = mfma3(acc0, w.ahi[S], w.alo[S],
                       as_h8(A0.hi[LL - 2][4 * S], A0.hi[LL - 2][4 * S + 1], A0.hi[LL - 2][4 * S + 2], A0.hi[LL - 2][4 * S + 3]),
                       as_h8(A0.lo[LL - 2][4 * S], A0.lo[LL - 2][4 * S + 1], A0.lo[LL - 2][4 * S + 2], A0.lo[LL - 2][4 * S + 3]));
        });
        if constexpr (LL < L) load_layer<LL + 1, false>(img, lane, wl[LL - 1]);
        else load_wo(img, lane, wv);
        constexpr int NP = BF ? 1 : 3;                 // products per k-step
        constexpr int NM = NP * KS;                    // matrix instructions of tile 1
        constexpr int PER = (NR + NM - 1) / NM;        // tile-0 registers per gap
        float e0[NR];
        PF_STAGE();
        sfor<0, NM>([&](auto m) {
          constexpr int M = m, S = M / NP, K = BF ? 2 : M % NP;
          acc1 = mfma3_part<K>(acc1, w.ahi[S], w.alo[S],
                               as_h8(A1.hi[LL - 2][4 * S], A1.hi[LL - 2][4 * S + 1], A1.hi[LL - 2][4 * S + 2], A1.hi[LL - 2][4 * S + 3]),
                               as_h8(A1.lo[LL - 2][4 * S], A1.lo[LL - 2][4 * S + 1], A1.lo[LL - 2][4 * S + 2], A1.lo[LL - 2][4 * S + 3]));
          sfor<M * PER, (M + 1) * PER < NR ? (M + 1) * PER : NR>([&](auto r) {
            constexpr int R = r;
            e0[R] = __builtin_amdgcn_exp2f(acc0[R] * CZ);
            // (an empty volatile use: without it the compiler sinks the stage below a later branch — the lockstep barrier of
            //  the forward kernels — and the gap stays empty)
            asm volatile("" : "+v"(e0[R]));
          });
          PF_STAGE();
        });
        float z1[NR];
        sfor<0, NR>([&](auto r) { constexpr int R = r; z1[R] = acc1[R]; });
        activate2_staggered<L, BWD, LL>(A0, A1, e0, z1, CZ, go, g0, g1);
      } else {
      if (!(dbg & 8)) {
        sfor<0, KS>([&](auto s) {
          constexpr int S = s;
          acc0 = mfma3(acc0, w.ahi[S], w.alo[S],
                       as_h8(A0.hi[LL - 2][4 * S], A0.hi[LL - 2][4 * S + 1], A0.hi[LL - 2][4 * S + 2], A0.hi[LL - 2][4 * S + 3]),
                       as_h8(A0.lo[LL - 2][4 * S], A0.lo[LL - 2][4 * S + 1], A0.lo[LL - 2][4 * S + 2], A0.lo[LL - 2][4 * S + 3]));
        });
        sfor<0, KS>([&](auto s) {
          constexpr int S = s;
          acc1 = mfma3(acc1, w.ahi[S], w.alo[S],
                       as_h8(A1.hi[LL - 2][4 * S], A1.hi[LL - 2][4 * S + 1], A1.hi[LL - 2][4 * S + 2], A1.hi[LL - 2][4 * S + 3]),
                       as_h8(A1.lo[LL - 2][4 * S], A1.lo[LL - 2][4 * S + 1], A1.lo[LL - 2][4 * S + 2], A1.lo[LL - 2][4 * S + 3]));
        });
      }
      // next phase's weights leave now, behind the matrix products and ahead of the tanh stages
      if constexpr (LL < L) load_layer<LL + 1, false>(img, lane, wl[LL - 1]);
      else load_wo(img, lane, wv);
      float z0[NR], z1[NR];
      sfor<0, NR>([&](auto r) { constexpr int R = r; z0[R] = acc0[R]; z1[R] = acc1[R]; });
      activate2<L, BWD, LL>(A0, A1, z0, z1, CZ, dbg, go, g0, g1);
      }
    });
    if (grp == 2) __builtin_amdgcn_s_barrier();
    p0 = 0.f;
    p1 = 0.f;
    if constexpr (!BWD) {      // (the backward pass takes the output from the forward launch's stored value)
      sfor<0, NR>([&](auto r) {
        constexpr int R = r;
        p0 = fmaf(wv[R], A0.aL[R], p0);
        p1 = fmaf(wv[R], A1.aL[R], p1);
      });
    }
  }

  // tanh of the NR pre-activations of ONE tile (backward recompute): same stages as activate2
  template <int L, int LL>
  static __device__ __forceinline__ void activate1(TileAct<L, true>& A, const float (&z)[NR], float cz) {
    float e[NR];
    PF_STAGE();
    sfor<0, NR>([&](auto r) { constexpr int R = r; e[R] = __builtin_amdgcn_exp2f(z[R] * cz); });
    PF_STAGE();
    sfor<0, NR>([&](auto r) { constexpr int R = r; e[R] = __builtin_amdgcn_rcpf(e[R] + 1.0f); });
    PF_STAGE();
    float a[2 * NPR];
    sfor<0, NR>([&](auto r) {
      constexpr int R = r;
      a[R] = fmaf(-2.0f * PF_N32_KA, e[R], PF_N32_KA);
      A.t[LL - 1][R] = fmaf(-e[R], e[R], e[R]);
      if constexpr (LL == L) A.aL[R] = a[R];
    });
    if constexpr (NR & 1) a[NR] = 0.f;
    sfor<0, NPK>([&](auto q) {
      constexpr int Q = q;
      if constexpr (Q < NPR) split_pair(a[2 * Q], a[2 * Q + 1], A.hi[LL - 1][Q], A.lo[LL - 1][Q]);
      else { A.hi[LL - 1][Q] = 0u; A.lo[LL - 1][Q] = 0u; }
    });
    PF_STAGE();
  }

  // hidden layers of ONE tile, no output unit: what the backward pass needs (activations and their derivatives).
  // Weights are read from the LDS image where they are used: this kernel lives on its register budget (three waves
  // per SIMD), and with three waves a SIMD hides the LDS latency by itself.
  template <int L, int IN>
  static __device__ __forceinline__ void recompute_tile(const unsigned char* __restrict__ img, int lane, const float (&x)[3],
                                                 TileAct<L, true>& A) {
    const int h = lane >> 5;
    constexpr float C2 = 2.8853900817779268f;
    {
      const float4* __restrict__ w1 = reinterpret_cast<const float4*>(img + pf_n32_off_w1());
      float z[NR];
      sfor<0, NR>([&](auto r) { constexpr int R = r; z[R] = layer1_row<IN>(w1[R * 2 + h], x); });
      activate1<L, 1>(A, z, C2);
    }
    sfor<2, L + 1>([&](auto l) {
      constexpr int LL = l;
      LayerW w;
      load_layer<LL, false>(img, lane, w);
      f32x16 acc = bias_acc(w);
      sfor<0, KS>([&](auto s) {
        constexpr int S = s;
        acc = mfma3(acc, w.ahi[S], w.alo[S],
                    as_h8(A.hi[LL - 2][4 * S], A.hi[LL - 2][4 * S + 1], A.hi[LL - 2][4 * S + 2], A.hi[LL - 2][4 * S + 3]),
                    as_h8(A.lo[LL - 2][4 * S], A.lo[LL - 2][4 * S + 1], A.lo[LL - 2][4 * S + 2], A.lo[LL - 2][4 * S + 3]));
      });
      float z[NR];
      sfor<0, NR>([&](auto r) { constexpr int R = r; z[R] = acc[R]; });
      activate1<L, LL>(A, z, C2 / (PF_N32_KA * PF_N32_KW));
    });
  }

  // ---- backward kernel ---------------------------------------------------------------------------------------
  // LDS per wave: two regions, A side (rows d_l of a gradient tile) and B side (columns a_{l-1}; the inputs image of the
  // first layer's tile aliases it: its other columns then hold stale activations, which only reach tile entries that are
  // never written out).  A region holds, per split (hi, lo), the registers 0..7 of every lane ("chunk 0", 16 B per lane
  // at lane*16 + (lane>>5)*64) and the registers 8..15 ("chunk 1").  COMPACT layout (PF_NR <= 12): chunk 1 keeps only
  // registers 8..11 (8 B per lane, [element/4][half-wave][element%4]), column quad 12..15 is read from a constant block
  // (zeros; for the hi image of the B side the block that carries KA in column 15 = the bias column of the tile).
  // All strides are chosen so that the 16-B / 8-B writes and the transposed 8-B reads touch every bank once:
  // a 32-lane half reads 16 pieces of chunk 0 in [c, c+128), 8 of chunk 1 in [c+128, c+192) and 8 of the constant block
  // in [c+192, c+256) (mod 256 B), c = 128*(lane>>5) + 64*(read half).
  static constexpr bool COMPACT = NR <= 12;
  static constexpr int SP_STRIDE = COMPACT ? 1792 : 2304;      // hi image -> lo image
  static constexpr int C1_OFF = 1152;                          // chunk 0 -> chunk 1
  static constexpr int REGION = COMPACT ? 3584 : 4608;
  static constexpr int WAVE_SCRATCH = 2 * REGION;
  // Constant blocks (512 B each, all = 192 mod 256) sit at the SAME distances from each other as the images they
  // complete: zero block Z at +192 (A hi), Z + SP_STRIDE (A lo), Z + REGION = bias block (B hi), Z + REGION + SP_STRIDE
  // (B lo).  Every lane therefore reads all four operands at ONE per-lane base plus compile-time offsets.
  static constexpr int CONST_BYTES = COMPACT ? 6144 : 256;
  // Waves per block (one block per CU) and the shape of the recompute.  Without pairing: as many waves as the kernel's
  // registers allow without spilling (ScratchSize 0 in the compiler's asm for every bucket) — 16 (four per SIMD, 128
  // registers) with one hidden layer, 12 with two, else 8.  A spill reload inside the task loop is followed by
  // s_waitcnt vmcnt(0), which also waits for the NEXT task's prefetched gathers: a spilling 12-wave variant ran slower
  // than the 8-wave one — hence the single per-lane LDS base (operand_base / write_base) and the SGPR-held block-uniform
  // scalars.  Going from 8 to 12 waves did NOT change the kernel time, though: the kernel is not short of waves.
  // PAIRED recompute: both tiles of a task go through the hidden layers together (forward_tiles, the forward kernel's
  // routine: twice the independent work in every MFMA / tanh / split phase) and only the back-propagation runs tile by
  // tile.  That needs ~200 registers, i.e. 8 waves per CU instead of 12 — and is FASTER (same box, 10^6 elements: E net
  // 51.6 -> 50.0 us, A net 40.0 -> 38.4 us): instruction-level parallelism inside a wave hides the MFMA -> tanh -> split
  // -> MFMA dependency chain better than a third wave does.  Used where it compiles without spills (ScratchSize 0 in
  // every bucket): two hidden layers, or three with PF_NR <= 8; one hidden layer keeps the per-tile form at 16 waves.
  template <int L>
  static constexpr bool bw_pair() { return PF_BW_PAIR && (L == 2 || (L == 3 && NR <= 8)); }
  template <int L, bool GEA>
  static constexpr int bw_threads() {
    if (bw_pair<L>()) return 512;
    constexpr int t = !COMPACT ? 512 : (L == 1 ? 1024 : (L == 2 && !(GEA && NR > 10) ? 768 : 512));
    return t < PF_BW_MAX_THREADS ? t : PF_BW_MAX_THREADS;
  }

  // lane's pairs (hi or lo) of registers 0..15 -> split `sp` of the region at byte offset `reg` of the wave scratch.
  // wr0 / wr1: the lane's write positions of chunk 0 / chunk 1 in the A region's hi image (write_base).  bias_col: wide
  // layout only, B side hi image: the lane's register 15 carries the bias column's KA (lower half-wave)
  struct WriteBase { unsigned char* c0; unsigned char* c1; };
  static __device__ __forceinline__ WriteBase write_base(unsigned char* scratch, int lane) {
    const int hs = lane >> 5, e = lane & 31;
    WriteBase w;
    w.c0 = scratch + hs * 576 + e * 16;
    w.c1 = COMPACT ? scratch + C1_OFF + 64 * (e >> 2) + 32 * hs + 8 * (e & 3) : scratch + C1_OFF + hs * 576 + e * 16;
    return w;
  }
  template <int REG, int SP>
  static __device__ __forceinline__ void write_rows(const WriteBase& wb, int lane, const unsigned (&pk)[NPK], bool bias_col = false) {
    constexpr int OFF = REG + SP * SP_STRIDE;
    *reinterpret_cast<u32x4*>(wb.c0 + OFF) = u32x4{pk[0], pk[1], pk[2], pk[3]};
    if constexpr (!COMPACT) {
      const unsigned k7 = pk[7] | ((bias_col && (lane >> 5) == 0) ? (BF ? 0x45000000u : 0x68000000u) : 0u);   // 2048.0 (bf16 / f16) in the high half
      *reinterpret_cast<u32x4*>(wb.c1 + OFF) = u32x4{pk[4], pk[5], pk[6], k7};
    } else if constexpr (NR > 8) {
      *reinterpret_cast<u32x2*>(wb.c1 + OFF) = u32x2{pk[4], pk[5]};
    }
  }

  // Per-lane base address of the transposed operand reads of one image (cdna_hip_programming.md T10): group g = lane>>4
  // reads the 4x16 block rows (elements) e0..e0+3, columns 16*(g&1)..+15; lane 4q+p of the group supplies the address of
  // row q, columns 4p..4p+3.  The element part 16*e0 = 256*ks + 64*half (+128*(lane>>5), folded in here) is the same for
  // every lane, so the reads of a tile use immediate offsets.
  static __device__ __forceinline__ const unsigned char* operand_base(const unsigned char* scratch, int lane,
                                                               const unsigned char* zblock) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int hsrc = g & 1, hl = g >> 1;
    const unsigned char* a;
    if constexpr (COMPACT) {
      a = p < 2 ? scratch + hsrc * 576 + q * 16 + 8 * (p & 1)
                : (p == 2 ? scratch + C1_OFF + 32 * hsrc + 8 * q : zblock + 32 * hsrc + 8 * q);
    } else {
      a = scratch + (p >> 1) * C1_OFF + hsrc * 576 + q * 16 + 8 * (p & 1);
    }
    return a + 128 * hl;
  }
  template <int OFF>
  static __device__ __forceinline__ h8 read_operand(const unsigned char* base, int ks) {
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(base + OFF + 256 * ks));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(base + OFF + 256 * ks + 64));
    typedef short s8v __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(h8, s8v{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
  }

  // backward of one tile: d_L from g_z S, back-propagation, and the tile's contribution to the gradient products
  // T[0] = combined tile (rows d_1, columns inputs), T[l-1] = rows d_l, columns a_{l-1} (l = 2..L).
  // wv: wo of this lane's units; wb: transposed operands of layer L (both loaded by recompute_tile).
  template <int L, int IN>
  static __device__ __forceinline__ void backward_tile(const unsigned char* __restrict__ img, unsigned char* scratch, int lane,
                                                const TileAct<L, true>& A, float gs /* g_z S of the column's element */,
                                                const unsigned (&xhi)[NPK], const unsigned (&xlo)[NPK],
                                                f32x16 (&T)[L], const WriteBase& wb,
                                                const unsigned char* rd /* operand_base: A hi; A lo, B hi, B lo at fixed offsets */,
                                                int dbg = 0) {
    const int h = lane >> 5;
    // d_L[r] = (4 wo[2r+h] g_z S) t_L[r]
    float d[2 * NPR];
    {
      float wv[16];
      load_wo(img, lane, wv);
      const float g4 = 4.0f * gs;
      sfor<0, NR>([&](auto r) { constexpr int R = r; d[R] = (wv[R] * g4) * A.t[L - 1][R]; });
      if constexpr (NR & 1) d[NR] = 0.f;
    }
    sfor<0, L>([&](auto s) {
      constexpr int LL = L - s;          // L .. 1: d holds d_LL (scaled by S 4^(L-LL))
      unsigned dhi[NPK], dlo[NPK];
      sfor<0, NPK>([&](auto q) {
        constexpr int Q = q;
        if constexpr (Q < NPR) split_pair(d[2 * Q], d[2 * Q + 1], dhi[Q], dlo[Q]);
        else { dhi[Q] = 0u; dlo[Q] = 0u; }
      });
      // Order of issue (a wave issues in order, so this order IS the overlap): the LDS writes of the gradient tile's
      // operands, the weights and ALL transposed reads, then the back-propagation products — whose 6 x 32 matrix cycles
      // cover the LDS write -> transposed read latency — and only then the tile products that consume the reads.
      // (Back-propagation first and the LDS round trip after it left ~200-300 cycles exposed per layer and tile.)
      f32x16 acc = zero16();
      LayerW w;
      if constexpr (LL >= 2) load_layer<LL, true>(img, lane, w);
      h8 ahi[2], alo[2], bhi[2], blo[2];
      if (!(dbg & 1)) {
        // gradient tile of layer LL: rows d_LL through LDS; columns a_{LL-1} (LL >= 2) or the inputs (LL == 1)
        write_rows<0, 0>(wb, lane, dhi);
        if constexpr (!BF) write_rows<0, 1>(wb, lane, dlo);
        if constexpr (LL >= 2) {
          write_rows<REGION, 0>(wb, lane, A.hi[LL - 2], true);
          if constexpr (!BF) write_rows<REGION, 1>(wb, lane, A.lo[LL - 2]);
        } else {
          // inputs of the element: registers 0..3 of the lower half-wave's slot (columns 0..3) of the B region
          // (for h == 0 the chunk-0 position IS lane * 16)
          if (h == 0) {
            *reinterpret_cast<u32x2*>(wb.c0 + REGION) = u32x2{xhi[0], xhi[1]};
            if constexpr (!BF) *reinterpret_cast<u32x2*>(wb.c0 + REGION + SP_STRIDE) = u32x2{xlo[0], xlo[1]};
          }
        }
        // the wave's own LDS writes are visible to its own later reads (in-order); tell the compiler only
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // inputs tile (LL == 1): its B hi image has no bias column (the element's inputs carry their own 1.0), so the
        // lanes that read column quad 12..15 take the ZERO block (their base, as it is) instead of the bias block
        const unsigned char* rdb = rd;
        if constexpr (LL == 1 && COMPACT) rdb = ((lane & 3) == 3) ? rd - REGION : rd;
        auto read_all = [&]() {
          sfor<0, 2>([&](auto ks) {
            constexpr int S = ks;
            ahi[S] = read_operand<0>(rd, S);
            bhi[S] = read_operand<REGION>(rdb, S);
            alo[S] = ahi[S]; blo[S] = bhi[S];     // (unused with plain bf16 operands)
            if constexpr (!BF) { alo[S] = read_operand<SP_STRIDE>(rd, S); blo[S] = read_operand<REGION + SP_STRIDE>(rd, S); }
          });
        };
        // (the wide layout with three hidden layers has no registers left for 16 reads in flight across the products:
        //  there the reads follow the back-propagation, as they used to)
        constexpr bool EARLY = COMPACT || L < 3;
        if constexpr (EARLY) {
          read_all();
          PF_STAGE();
        }
        // back-propagation to layer LL-1
        if constexpr (LL >= 2) {
          sfor<0, KS>([&](auto ks) {
            constexpr int S = ks;
            acc = mfma3(acc, w.ahi[S], w.alo[S], as_h8(dhi[4 * S], dhi[4 * S + 1], dhi[4 * S + 2], dhi[4 * S + 3]),
                        as_h8(dlo[4 * S], dlo[4 * S + 1], dlo[4 * S + 2], dlo[4 * S + 3]));
          });
        }
        if constexpr (!EARLY) read_all();
        PF_STAGE();
        sfor<0, 2>([&](auto ks) { constexpr int S = ks; T[LL - 1] = mfma3(T[LL - 1], ahi[S], alo[S], bhi[S], blo[S]); });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      } else if constexpr (LL >= 2) {     // (timing build with the gradient tiles switched off)
        sfor<0, KS>([&](auto ks) {
          constexpr int S = ks;
          acc = mfma3(acc, w.ahi[S], w.alo[S], as_h8(dhi[4 * S], dhi[4 * S + 1], dhi[4 * S + 2], dhi[4 * S + 3]),
                      as_h8(dlo[4 * S], dlo[4 * S + 1], dlo[4 * S + 2], dlo[4 * S + 3]));
        });
      }
      if constexpr (LL >= 2) {
        sfor<0, NR>([&](auto r) { constexpr int R = r; d[R] = acc[R] * A.t[LL - 2][R]; });
      }
    });
  }

};

// Stores of a block's partial gradient row; wt: agent-scope write-through (global_store ... sc1).  With PF_FUSE_S1=1 the
// fused backward launch hands the rows to the LAST block of each row group inside the launch (rows_reduce_last below), and a hand-off without a
// release fence needs every handed-off byte stored this way (MI355X_MICROARCH.md, inter-workgroup visibility, valid forms).
__device__ __forceinline__ void row_store(float* p, float v, bool wt) {
  if (wt) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;                                       // (the usual case: the rows are read by the NEXT launch)
}
__device__ __forceinline__ float row_load(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The arithmetic of k_theta_stage1 for row group g, by one whole block: four interleaved row sums per column, combined
// (0 + 1) + (2 + 3).  wt_loads: the rows were stored write-through inside this launch; wt_store: the group's row is handed to
// other blocks of this launch (agent-scope store).
__device__ __forceinline__ void stage1_group(const pf_problem& P, int nb_rows, int g, bool wt_loads, bool wt_store) {
  const int rpg = (nb_rows + PF_RG - 1) / PF_RG;
  const int r0 = g * rpg, r1 = min(r0 + rpg, nb_rows);
  float* __restrict__ out = P.partials + PF_PART_WG + (size_t)P.n_part_blocks * P.pad_total + (size_t)g * P.pad_total;
  for (int col = threadIdx.x; col < P.pad_total; col += blockDim.x) {
    const float* rows = P.partials + PF_PART_WG + col;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < r1; r += 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (r + k < r1) {
          const float* src = rows + (size_t)(r + k) * P.pad_total;
          a[k] += wt_loads ? row_load(src) : *src;
        }
    }
    row_store(out + col, (a[0] + a[1]) + (a[2] + a[3]), wt_store);
  }
}

// ---- the parameter update of the PREVIOUS iteration, run by a forward launch for itself (pf_problem.theta_alt) --------
// Every block: second-level partial rows -> gradient -> Adam step from state half `half_in` (pf_theta_update: the
// arithmetic of the stand-alone update kernel, bit for bit) -> its own LDS copy of the new parameters -> the operand
// images of the enabled nets straight into LDS (img_lds[k]; null: that net's image is not needed by this launch).  Block 0
// also stores the new state into the other half, the images (with the backward's scaling bound) and the theta-norm
// monitor to global memory and flips state->theta_half: nothing any block of this launch reads.  Two block barriers.
__device__ __forceinline__ void fwd_theta_prologue(const pf_problem& P, int half_in, unsigned char* img_lds0,
                                                   unsigned char* img_lds1, float* new_theta, int img_bytes, int* s_done,
                                                   bool calc_index = false, unsigned long long* stamps = nullptr) {
  const bool lead = blockIdx.x == 0;
  // The stop flag is stable while a forward launch runs (the bookkeeping that raises it is ordered behind it), so the lead
  // block may read it for itself, and every block's update loads leave together with the block's one flag read instead
  // of behind a barrier on it.
  const int done0 = lead ? P.state->done : 1;
  if (threadIdx.x == 0) *s_done = P.state->done;
  pf_theta_update(P, 1, new_theta, done0, half_in, half_in ^ 1, calc_index);
  if (PF_N32_DBG_ENABLE && stamps) stamps[1] = __builtin_amdgcn_s_memrealtime();
  __syncthreads();
  if (PF_N32_DBG_ENABLE && stamps) stamps[2] = __builtin_amdgcn_s_memrealtime();
  if (*s_done) return;
  // (two nets: the two halves of the block pack them side by side)
  const bool two = P.net[0].enabled && P.net[1].enabled && blockDim.x >= 256;
  const int half = two ? ((int)blockDim.x / 2) & ~63 : 0;
  auto pack_net = [&](int k, unsigned char* lds_img) {
    if (!P.net[k].enabled) return;
    unsigned char* gimg = reinterpret_cast<unsigned char*>(P.net_op + P.op_off[k]);
    const int t_first = two ? k * half : 0, t_count = two ? (k == 0 ? half : (int)blockDim.x - half) : 0;
    if (lds_img) pf_n32_pack(P.net[k], new_theta + P.net[k].theta_off, lds_img, P.mlp_dtype, lead, t_first, t_count);
    else if (lead) pf_n32_pack(P.net[k], new_theta + P.net[k].theta_off, gimg, P.mlp_dtype, true, t_first, t_count);
  };
  pack_net(0, img_lds0);
  pack_net(1, img_lds1);
  if (PF_N32_DBG_ENABLE && stamps) stamps[3] = __builtin_amdgcn_s_memrealtime();
  __shared__ float tnorm[PF_MAX_TENSORS];
  if (lead) {
    // (the theta norm from the LDS copy, one wave per tensor in turn; inactive parameters come from p->theta)
    tensor_norms(P, new_theta, threadIdx.x >> 6, blockDim.x >> 6, tnorm);
  }
  __syncthreads();
  if (lead) {
    if (P.net[0].enabled && img_lds0) copy_image(reinterpret_cast<unsigned char*>(P.net_op + P.op_off[0]), img_lds0, img_bytes);
    if (P.net[1].enabled && img_lds1) copy_image(reinterpret_cast<unsigned char*>(P.net_op + P.op_off[1]), img_lds1, img_bytes);
    if (threadIdx.x == 0) {
      P.state->theta_norm = (float)tensor_norm_total(P, tnorm);
      P.state->theta_half = half_in ^ 1;
    }
  }
}

// ---- forward kernel ----------------------------------------------------------------------------------------
// One block of 16 waves per CU (four per SIMD); every wave walks 64-element tasks, the block in rounds of 16 tasks.
// The SIMD arbitrates its waves by age: left alone the oldest runs nearly unimpeded, the waves of a SIMD finish far
// apart (measured: 26 ... 71 us for equal work) and the tail runs on one wave per SIMD.  ONE s_barrier per task
// keeps the block's waves within a task of each other; the four wave groups (wave >> 2: one wave per SIMD each)
// pass it at four different places of the task body, so the waves of a SIMD stay a quarter task out of phase
// instead of reaching their transcendental, matrix and LDS phases together (MI355X_MICROARCH.md, two waves per
// SIMD, item 9).  Lanes past the end work on the last element again (same inputs, same value) and do not store.
constexpr int FW_THREADS = 1024;

// s2_half >= 0: this launch first runs the parameter update of the previous iteration from that state half
// (fwd_theta_prologue) and takes its operand image from there instead of from global memory.
template <int NR, int L, int IN>
__global__ __launch_bounds__(FW_THREADS) void k_net32_forward(pf_problem P, int which, int dbg_arg, int ws, int s2_half) {
  using E = Eng<NR>;
  const bool calc_index = s2_half >= 2;       // (s2_half + 2: padded-image indices by arithmetic, pf_fwd2_opts.calc_index)
  if (calc_index) s2_half -= 2;
  const int dbg = PF_N32_DBG_ENABLE ? dbg_arg : 0;     // compile-time 0 in the product build (see PF_N32_DBG_ENABLE)
  extern __shared__ __align__(16) unsigned char smem[];
  const pf_net net = P.net[which];
  if (s2_half < 0) copy_image(smem, reinterpret_cast<const unsigned char*>(P.net_op + P.op_off[which]), pf_n32_bytes(L));
  float* __restrict__ out = which == 0 ? P.prop_e : P.prop_a;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const int grp = (dbg & 32) ? -1 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);   // provably wave-uniform
  const int n = P.mesh.n_elems;
  const int ntasks = (n + 63) >> 6;
  const int per_round = gridDim.x * waves;
  const int rounds = (ntasks + per_round - 1) / per_round;        // block-uniform trip count: equal barrier counts
  int task = blockIdx.x * waves + wv;
  float xn[3];
  // ws: this launch also writes the stiffness records from both properties: the other property and the element geometry
  // travel with the coordinates, one round ahead (loaded where they are used they cost a global round trip per round)
  const pf_net onet = P.net[1 - which];
  const float* __restrict__ oprop = which == 0 ? P.prop_a : P.prop_e;
  float on = onet.scale;
  ElemGeo gn = ElemGeo{0.f, 0.f, 0.f, 1.f};
  if (n > 0) {
    const int e0 = min(task * 64 + lane, n - 1);
    load_x<IN>(xn, P, e0);
    if (ws) {
      if (onet.enabled) on = oprop[e0];
      gn = load_geo(P.mesh.egeo, e0);
    }
  }
  // the stop flag, one read per block (the bookkeeping that raises it may run beside this launch: a block whose waves
  // disagreed would part ways before the lockstep barriers below)
  __shared__ int s_done;
  if (s2_half >= 0) {
    fwd_theta_prologue(P, s2_half, which == 0 ? smem : nullptr, which == 1 ? smem : nullptr,
                       reinterpret_cast<float*>(smem + ((pf_n32_bytes(L) + 255) & ~255)), pf_n32_bytes(L), &s_done, calc_index);
  } else {
    if (threadIdx.x == 0) s_done = P.state->done;
    __syncthreads();
  }
  if (s_done || n <= 0) return;
  const float bo = reinterpret_cast<const float*>(smem + pf_n32_off_bo())[0];
  unsigned long long st0 = 0, sr0 = 0;
  if (dbg & 16) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); }
  for (int r = 0; r < rounds; ++r, task += per_round) {
    if (grp == 0) __builtin_amdgcn_s_barrier();
    const int e = task * 64 + lane;
    float x0[3], x1[3];
    sfor<0, 3>([&](auto c) { constexpr int C = c; both_tiles(xn[C], x0[C], x1[C]); });
    const float o = on;
    const ElemGeo g = gn;
    if (r + 1 < rounds) {
      const int en = min(e + per_round * 64, n - 1);
      load_x<IN>(xn, P, en);
      if (ws) {
        if (onet.enabled) on = oprop[en];
        gn = load_geo(P.mesh.egeo, en);
      }
    }
    typename E::template TileAct<L, false> A0, A1;
    float p0, p1;
    E::template forward_tiles<L, IN, false>(smem, lane, x0, x1, A0, A1, p0, p1, dbg, grp);
    const float z = own_total(p0, p1) * (1.0f / PF_N32_KA) + bo;
    if (e < n) {
      const float v = (net.positive ? pf_softplus(z) : z) * net.scale;
      out[e] = v;
      // stiffness record for the node kernels: s = (young * area) / l0, nn_assembly.py:74 (2-D), :37 (1-D), times the
      // pattern entries (:84-94)
      if (ws) store_k<IN - 1>(P.elem_k, e, g, (which == 0 ? v * o : o * v) / g.l0);
    }
    if (grp == 3) __builtin_amdgcn_s_barrier();
  }
  if ((dbg & 16) && lane == 0) {   // diagnostic build only: per-wave stamps into the (unused here) partial-sum workspace
    unsigned long long* d = reinterpret_cast<unsigned long long*>(P.partials) + (size_t)(blockIdx.x * waves + wv) * 4 + (size_t)which * 65536;
    d[0] = sr0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = __builtin_amdgcn_s_memtime() - st0;
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    d[3] = ((unsigned long long)rounds << 48) | ((unsigned long long)(xcc & 0xf) << 32) | hwid;
  }
}

// ---- fused forward of BOTH nets (young and area) --------------------------------------------------------------------
// One launch instead of two: the wave's 64 elements go through the E net and then the A net (both operand images sit in
// LDS), the coordinates are loaded and exchanged between the half-waves once, and the stiffness record is formed from
// both values in registers — no second launch floor, no second prologue, no round trip of the first property through
// memory.  Same block shape and lockstep scheme as k_net32_forward; the four wave groups pass their ONE barrier per task
// at: task start | after the E net's first activation | after the A net's hidden layers | task end.
// s2: this launch is also the THETA UPDATE of the previous iteration (the iteration graph): every block sums the
// second-level partial rows, applies Adam to its own copy of theta and builds both operand images straight into its LDS
// (all blocks compute the same bits); block 0 stores theta, the moments, the images and the theta-norm monitor — see
// fwd2_theta_prologue.
// gu_nb > 0 (the iteration graph, queue form only): this launch is also the DISPLACEMENT UPDATE of the previous iteration
// (dL/du + Adam(u) + clamp: what k_node_gradu does, pf_node.h).  The block owns a contiguous run of node tasks (PF_GU_M x
// 64 nodes each) beside its element tasks and its waves draw one of them after every `gu_every` element tasks: the node
// tasks are three dependent round trips through memory with next to no arithmetic, the element tasks vector-issue bound —
// the waves of a SIMD hide one behind the other, and the graph needs no side branch (no fork, no join: ~12 us per
// iteration on MI355X) and no second displacement vector.  Each task's sum of u_free^2 lands in LDS at the task's index
// and the block adds them in index order: the monitor does not depend on which wave drew what.  The block's sum goes to
// partials[PF_PART_U2 + block]; entries up to gu_nb (what the bookkeeping sums: pf_node_blocks) are zeroed.  gu_k: the
// stiffness records of the previous iteration (the half this launch does NOT write).
#ifndef PF_GU_M
#define PF_GU_M 2
#endif
#define PF_GU_MAX_TASKS 1024
template <int NRE, int NRA, int L, int IN>
__global__ __launch_bounds__(FW_THREADS) void k_net32_forward2(pf_problem P, int dbg_arg, int s2_half, int queue, int gu_nb,
                                                               const float* __restrict__ gu_k) {
  using EE = Eng<NRE>;
  using EA = Eng<NRA>;
  const bool calc_index = s2_half >= 2;       // (s2_half + 2: padded-image indices by arithmetic, pf_fwd2_opts.calc_index)
  if (calc_index) s2_half -= 2;
  const int dbg = PF_N32_DBG_ENABLE ? dbg_arg : 0;
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int IMG = pf_n32_bytes(L), IMGP = (IMG + 255) & ~255;
  const unsigned char* __restrict__ img_e = smem;
  const unsigned char* __restrict__ img_a = smem + IMGP;
  const pf_net net_e = P.net[0], net_a = P.net[1];
  if (s2_half < 0) {
    copy_images2(smem, reinterpret_cast<const unsigned char*>(P.net_op + P.op_off[0]), smem + IMGP,
                 reinterpret_cast<const unsigned char*>(P.net_op + P.op_off[1]), IMG);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const int n = P.mesh.n_elems;
  const int ntasks = (n + 63) >> 6;
  // Two ways to hand the 64-element tasks to the waves.  queue (default): the block owns a contiguous run of tasks and
  // its waves draw them from a counter in LDS, so no wave ever waits for another one — the age-ordered arbitration of a
  // SIMD lets its oldest wave run ahead, and with a queue that wave simply takes more tasks (the waves drift out of
  // phase by themselves).  Lockstep (queue == 0, round 2): block-uniform rounds of one task per wave with ONE s_barrier
  // per task, passed at a different place of the task body by each of the four wave groups.
  const int grp = (queue || (dbg & 32)) ? -1 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
  const int per_round = gridDim.x * waves;
  const int rounds = (ntasks + per_round - 1) / per_round;
  const int per_block = (ntasks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t0 = (int)blockIdx.x * per_block, t1 = min(t0 + per_block, ntasks);
  __shared__ int s_next, s_next_n;
  __shared__ float s_u2[PF_GU_MAX_TASKS];
  if (threadIdx.x == 0) {
    s_next = waves;           // tasks t0 .. t0+waves-1 go to the waves in order, the rest through the queue
    s_next_n = 0;
  }
  int task = queue ? t0 + wv : (int)blockIdx.x * waves + wv;
  float xn[3];
  ElemGeo gn = ElemGeo{0.f, 0.f, 0.f, 1.f};
  if (n > 0) {
    const int e0 = min(task * 64 + lane, n - 1);
    load_x<IN>(xn, P, e0);
    gn = load_geo(P.mesh.egeo, e0);
  }
  // node tasks of this block (gu_nb > 0)
  constexpr int GUN = 64 * PF_GU_M;
  const int n_ntasks = gu_nb > 0 ? (P.mesh.n_nodes + GUN - 1) / GUN : 0;
  const int npb = (n_ntasks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int nt0 = (int)blockIdx.x * npb, nn_b = max(min(nt0 + npb, n_ntasks) - nt0, 0);
  const int gu_every = nn_b > 0 ? max((t1 - t0) / nn_b + (int)(signed char)(queue >> 8), 0) : 0;     // (queue >> 8: experiment, PF_GU_EVERY_ADD)
  bool gu = nn_b > 0;
  int since = nn_b > 0 ? wv % (gu_every + 1) : 0;     // (the waves start out of phase)
  __shared__ int s_done;
  unsigned long long* dstamps = nullptr;       // (diagnostic build, PF_N32_DBG=16: per-wave time stamps into pf_problem.u_alt)
  // one node task of the block's queue by the calling wave; false when the queue is empty
  auto node_task = [&]() {
    int k = 0;
    if (lane == 0) k = atomicAdd(&s_next_n, 1);
    k = __builtin_amdgcn_readfirstlane(k);
    if (k >= nn_b) return false;
    // what the task needs of the problem is read from the kernel argument HERE (P is the first argument: offset 0 of the
    // kernarg segment), behind an opaque move: left to the compiler, every pointer is loaded once at the top of the launch
    // and the element loop spills scalar registers (75 against 7 without the node tasks)
    const __attribute__((address_space(4))) pf_problem* pk =
        (const __attribute__((address_space(4))) pf_problem*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(pk));
    const NodeView NV = node_view_from(pk);
    const GraduConsts GK = gradu_consts_from(pk);
    float su = node_gradu_task<IN - 1, PF_GU_M>(NV, gu_k, GK, (nt0 + k) * GUN, lane);
    su = pf_wave_sum(su);
    if (lane == 0) s_u2[k] = su;
    return true;
  };
  if (s2_half >= 0) {
    if (PF_N32_DBG_ENABLE && (dbg & 16)) {
      dstamps = reinterpret_cast<unsigned long long*>(P.u_alt) + (size_t)(blockIdx.x * waves + wv) * 8;
      if (lane == 0) dstamps[0] = __builtin_amdgcn_s_memrealtime();
    }
    fwd_theta_prologue(P, s2_half, smem, smem + IMGP, reinterpret_cast<float*>(smem + 2 * IMGP), IMG, &s_done, calc_index,
                       (dstamps && lane == 0) ? dstamps : nullptr);
    if (dstamps && lane == 0) dstamps[4] = __builtin_amdgcn_s_memrealtime();
  } else {
    if (threadIdx.x == 0) s_done = P.state->done;
    __syncthreads();
  }
  if (s_done || n <= 0) return;
  const float bo_e = reinterpret_cast<const float*>(img_e + pf_n32_off_bo())[0];
  const float bo_a = reinterpret_cast<const float*>(img_a + pf_n32_off_bo())[0];
  for (int r = 0; queue ? (task < t1 || gu) : r < rounds; ++r) {
    if (gu && (since >= gu_every || task >= t1)) {
      // a node task: the next element task's inputs (already on their way) are not touched
      since = 0;
      gu = node_task();
      continue;
    }
    if (task >= t1) break;
    ++since;
    if (grp == 0) __builtin_amdgcn_s_barrier();
    // the task after this one: drawn now (queue) so that its inputs travel while this one computes
    int nxt;
    bool more;
    if (queue) {
      int k = 0;
      if (lane == 0) k = atomicAdd(&s_next, 1);
      nxt = t0 + __builtin_amdgcn_readfirstlane(k);
      more = nxt < t1;
    } else {
      nxt = task + per_round;
      more = r + 1 < rounds;
    }
    const int e = task * 64 + lane;
    float x0[3], x1[3];
    sfor<0, 3>([&](auto c) { constexpr int C = c; both_tiles(xn[C], x0[C], x1[C]); });
    const ElemGeo g = gn;
    if (more) {
      const int en = min(nxt * 64 + lane, n - 1);
      load_x<IN>(xn, P, en);
      gn = load_geo(P.mesh.egeo, en);
    }
    float ze, za;
    {
      typename EE::template TileAct<L, false> A0, A1;
      float p0, p1;
      EE::template forward_tiles<L, IN, false>(img_e, lane, x0, x1, A0, A1, p0, p1, dbg, grp == 1 ? 1 : -1);
      ze = own_total(p0, p1) * (1.0f / PF_N32_KA) + bo_e;
    }
    {
      typename EA::template TileAct<L, false> A0, A1;
      float p0, p1;
      EA::template forward_tiles<L, IN, false>(img_a, lane, x0, x1, A0, A1, p0, p1, dbg, grp == 2 ? 2 : -1);
      za = own_total(p0, p1) * (1.0f / PF_N32_KA) + bo_a;
    }
    if (e < n) {
      const float ve = (net_e.positive ? pf_softplus(ze) : ze) * net_e.scale;
      const float va = (net_a.positive ? pf_softplus(za) : za) * net_a.scale;
      P.prop_e[e] = ve;
      P.prop_a[e] = va;
      if (P.elem_k) store_k<IN - 1>(P.elem_k, e, g, (ve * va) / g.l0);    // (young * area) / l0, nn_assembly.py:74, :37
    }
    if (grp == 3) __builtin_amdgcn_s_barrier();
    task = nxt;
  }
  if (dstamps && lane == 0) dstamps[5] = __builtin_amdgcn_s_memrealtime();
  if (gu_nb > 0) {
    // the block's share of sum u_free^2, node tasks in index order (every wave's loop has ended: no return above this point
    // once the stop flag was read)
    __syncthreads();
    float su = 0.f;
    for (int i = threadIdx.x; i < nn_b; i += blockDim.x) su += s_u2[i];
    float* red = reinterpret_cast<float*>(s_u2 + PF_GU_MAX_TASKS - 16);
    __syncthreads();
    su = pf_block_sum(su, red);
    if (threadIdx.x == 0) P.partials[PF_PART_U2 + blockIdx.x] = su;
    for (int j = (int)blockIdx.x + (int)gridDim.x + (int)threadIdx.x * (int)gridDim.x; j < gu_nb; j += (int)blockDim.x * (int)gridDim.x)
      P.partials[PF_PART_U2 + j] = 0.f;
  }
}

// First level of the parameter-gradient reduction INSIDE the backward launch (what the k_theta_stage1 launch does
// otherwise, to the bit: same row groups, same summation order): the block's row is complete and stored; every wave drains
// its stores, the block takes a ticket of its row group, and the block that draws the group's last ticket sums the
// group's rows into the group's second-level row.  No block ever waits for another one.  tickets: PF_RG counters, zero
// between launches (the last block of a group resets its counter).
__device__ __forceinline__ void rows_reduce_last(const pf_problem& P, int nb_rows) {
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's row stores have left
  __syncthreads();
  const int rpg = (nb_rows + PF_RG - 1) / PF_RG;
  const int g = (int)blockIdx.x / rpg;
  const int r0 = g * rpg, r1 = min(r0 + rpg, nb_rows);
  int* tickets = reinterpret_cast<int*>(P.partials + PF_PART_WG + ((size_t)P.n_part_blocks + PF_RG) * P.pad_total);
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(tickets + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = old == (r1 - r0) - 1;
    if (s_last) __hip_atomic_store(tickets + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last) return;
  stage1_group(P, nb_rows, g, true, false);
}

// ---- a wave's finished gradient tiles parked in LDS, and the block's row from the parked tiles ---------------------------
// (the fused two-phase backward launch: a wave that has finished the young net's walk parks its tiles and starts the area
// net's walk at once instead of waiting at a block barrier for the block's slowest wave; both rows are formed at the end of
// the launch.  Same values in the same order as the write-out at the end of backward_phase: sums over the waves ascending.)
// A wave's parking place: [L][1024] tile entries (row m = tile row, 32 columns), then [2][16] output-unit row, then g_bo.
template <int L>
constexpr int bw_park_floats() { return L * 1024 + 64; }
template <int NR, int L>
__device__ __forceinline__ void bw_park(float* __restrict__ mine, const f32x16 (&T)[L], const float (&go)[NR], float gbo, int lane) {
  const int h = lane >> 5;
  sfor<0, L>([&](auto l) {
    constexpr int LL = l;
    sfor<0, 16>([&](auto i) {
      constexpr int I = i;
      const int m = (I & 3) + 8 * (I >> 2) + 4 * h;      // tile row
      mine[LL * 1024 + m * 32 + (lane & 31)] = T[LL][I];
    });
  });
  sfor<0, NR>([&](auto r) {
    constexpr int R = r;
    float v = go[R];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((lane & 31) == 0) mine[L * 1024 + h * 16 + R] = v;
  });
  const float v = pf_wave_sum(gbo);
  if (lane == 0) mine[L * 1024 + 32] = v;
}
// every thread of the block; the caller has placed a block barrier behind the last bw_park.  No barrier inside except the
// one between the row's zero fill and its entries.
template <int NR, int L, int IN>
__device__ __forceinline__ void bw_row_from_parked(const pf_problem& P, int which, int hp, const float* __restrict__ park,
                                                   float kl, float kx, bool wt) {
  const pf_net net = P.net[which];
  const int waves = blockDim.x >> 6, W = net.width;
  constexpr int PS = bw_park_floats<L>();
  float* __restrict__ prow = P.partials + PF_PART_WG + (size_t)blockIdx.x * P.pad_total + net.pad_off;
  const int padc = pf_pad_count(hp, L);
  for (int i = threadIdx.x; i < padc; i += blockDim.x) row_store(prow + i, 0.f, wt);
  __syncthreads();
  sfor<0, L>([&](auto l) {
    constexpr int LL = l + 1;                      // layer whose weight gradient tile T[LL-1] holds
    float sc = 1.f;
    for (int k = 0; k < L - LL; ++k) sc *= 4.f;
    for (int idx = threadIdx.x; idx < 1024; idx += blockDim.x) {
      const int m = idx >> 5, c = idx & 31;
      const int j = 2 * (m & 15) + (m >> 4);
      if ((m & 15) >= NR || j >= W) continue;
      float t = 0.f;
      for (int q = 0; q < waves; ++q) t += park[q * PS + (LL - 1) * 1024 + idx];
      int dst = -1;
      float k = sc;
      if (LL >= 2) {
        const int kk = 2 * (c & 15) + (c >> 4);
        k *= PF_N32_KA;
        if (c == 15) dst = pf_pad_wh(hp, LL) + j * (hp + 4) + hp;                      // bias column
        else if ((c & 15) < NR && kk < W) dst = pf_pad_wh(hp, LL) + j * (hp + 4) + kk;
      } else {
        if (c < IN) { dst = j * 4 + c; k *= c == 0 ? kl : kx; }
        else if (c == IN) dst = j * 4 + IN;                                           // bias (input 1.0)
      }
      if (dst >= 0) row_store(prow + dst, t / k, wt);
    }
  });
  if (threadIdx.x < 32) {
    const int hh = threadIdx.x >> 4, r = threadIdx.x & 15, u = 2 * r + hh;
    if (r < NR && u < W) {
      float t = 0.f;
      for (int q = 0; q < waves; ++q) t += park[q * PS + L * 1024 + hh * 16 + r];
      row_store(prow + pf_pad_wo(hp, L) + u, t * (1.0f / PF_N32_KA), wt);
    }
  }
  if (threadIdx.x == 32) {
    float t = 0.f;
    for (int q = 0; q < waves; ++q) t += park[q * PS + L * 1024 + 32];
    row_store(prow + pf_pad_wo(hp, L) + hp, t, wt);
  }
}

// One net's backward over the block's tasks, as a phase of a launch: `smem` = this net's operand image in LDS (copied by
// the caller, no barrier needed before the call), `cst` = the constant blocks (CONST_BYTES of this bucket), `wscr` = the
// block's wave scratches (the write-out staging reuses them).  Ends with the block's partial gradient row written; the
// caller places a block barrier between two phases that share cst / wscr.
// MODE 0: a phase of its own (entry barriers, write-out at its end).  MODE 1 / 2 (fused launch, both buckets in the compact
// layout: same constant blocks, same scratch size): 1 = first phase, the wave PARKS its tiles in `park` (bw_park) instead
// of the write-out; 2 = second phase entered WITHOUT a block barrier (constant blocks and images are in place since phase
// 1; the wave's scratch is its own), at its end: block barrier, `first_row()` (the first phase's row from the parked tiles),
// block barrier, park, block barrier, own row.
template <int NR, int L, int IN, bool GEA, int MODE = 0, class FirstRow = int>
__device__ __forceinline__ void backward_phase(const pf_problem& P, int which, int hp, int dbg, const unsigned char* smem,
                                               unsigned char* cst, unsigned char* wscr, bool wt = false,
                                               unsigned long long* stamps = nullptr, float* park = nullptr,
                                               FirstRow first_row = FirstRow(), int shift8 = 0) {
  using E = Eng<NR>;
  constexpr int CONST_BYTES = E::CONST_BYTES, WAVE_SCRATCH = E::WAVE_SCRATCH, REGION = E::REGION, NPK = E::NPK;
  constexpr bool COMPACT = E::COMPACT;
  typedef typename E::WriteBase WriteBase;
  constexpr int DIM = IN - 1;
  const pf_net net = P.net[which];
  const pf_net onet = P.net[1 - which];
  const float* __restrict__ other = which == 0 ? P.prop_a : P.prop_e;
  const float* __restrict__ mine = which == 0 ? P.prop_e : P.prop_a;   // this net's forward values (pf_net_forward)
  const int n = P.mesh.n_elems;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const int h = lane >> 5;
  unsigned char* scratch = wscr + wv * WAVE_SCRATCH;
  if constexpr (MODE != 2)
    for (int i = threadIdx.x; i < CONST_BYTES / 4; i += blockDim.x) reinterpret_cast<unsigned*>(cst)[i] = 0u;
  for (int i = lane; i < WAVE_SCRATCH / 16; i += 64) reinterpret_cast<uint4*>(scratch)[i] = make_uint4(0, 0, 0, 0);
  if constexpr (MODE != 2) __syncthreads();
  // bias block: [element/4][half-wave][element%4] x (registers 12..15 as f16): KA in register 15 of the lower half-wave
  if (MODE != 2 && threadIdx.x < 32)
    if constexpr (COMPACT)
      *reinterpret_cast<unsigned short*>(cst + 192 + REGION + 64 * (threadIdx.x >> 2) + 8 * (threadIdx.x & 3) + 6) = BF ? 0x4500 : 0x6800;   // 2048.0
  const unsigned char* rd = E::operand_base(scratch, lane, cst + 192);
  const WriteBase wb = E::write_base(scratch, lane);

  // Which 64-element tasks the wave walks: its own column of the round-robin (task = round * waves of the launch + the
  // wave's slot), and — shift8 > 0 — a handicap between the two waves of a SIMD (w and w + waves/2): the SIMD arbitrates by
  // age, the older wave (lower id) runs nearly unimpeded and left alone finishes its walk long before the younger one, which
  // then does its last tasks alone with nothing to hide its latencies behind (time stamps: the waves of one block leave a
  // loop up to 14 us apart).  So the older wave also takes the LAST shift8/8 of the younger wave's column.  A fixed
  // assignment: the sums stay reproducible, and the same in both phases of the fused launch (a wave reads back the element
  // adjoints it wrote itself).
  const int per_round = (int)gridDim.x * waves;
  const int ntasks_all = (n + 63) >> 6;
  const int slot = (int)blockIdx.x * waves + wv;
  const int hw = waves >> 1;
  const bool elder = wv < hw;
  const int partner = elder ? slot + hw : slot - hw;
  const int c_own = slot < ntasks_all ? (ntasks_all - 1 - slot) / per_round + 1 : 0;
  const int c_par = (hw > 0 && partner < ntasks_all) ? (ntasks_all - 1 - partner) / per_round + 1 : 0;
  // (shift8 & 15: the handicap; a handicap can never exceed the younger wave's column)
  const int sh = min(((shift8 & 15) > 0 && hw > 0 && (waves & 1) == 0) ? (((elder ? c_par : c_own) * (shift8 & 15) + 4) >> 3) : 0,
                     elder ? c_par : c_own);
  const int n_mine = elder ? c_own + sh : c_own - sh;
  auto task_base = [&](int k) {                  // element base of the wave's k-th task; past its last one: the last element
    if (k < (elder ? c_own : n_mine)) return (k * per_round + slot) * 64;
    if (elder && k < n_mine) return ((c_par - sh + (k - c_own)) * per_round + partner) * 64;
    return n - 1;
  };
  TaskIn<DIM> nxt;
  int2 nn_ahead = int2{0, 0};                    // node ids of the task after next (GEA only)
  if (n > 0) {
    task_fetch_a<IN, GEA>(nxt, P, onet, other, mine, min(task_base(0) + lane, n - 1));
    if (GEA) nn_ahead = reinterpret_cast<const int2*>(P.mesh.conn)[min(task_base(1) + lane, n - 1)];
  }
  if constexpr (MODE != 2) __syncthreads();      // (image, constant blocks and scratch are in place)
  task_fetch_b<IN, GEA>(nxt, P);

  // block-uniform scalars: keep them in SGPRs (the kernel runs at its register budget)
  const float bound = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, reinterpret_cast<const float*>(smem)[0])));
  const float kx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, __builtin_ldexpf(1.0f, P.coord_exp))));
  const float inv_scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.0f / net.scale)));
  // scale of the load factor inside the f16 gradient products: the power of two with 2^13 <= kl |lam| < 2^14 (like the
  // coordinates' kx; a fixed factor would overflow f16 for large |lam| and lose the lo part for small ones)
  const float kl = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pf_n32_lam_scale(P.lam))));
  f32x16 T[L];          // sum over this wave's tasks of S * (tile products); S = Srun, a power of two that only falls
  sfor<0, L>([&](auto l) { constexpr int LL = l; T[LL] = zero16(); });
  float Srun = 0.f;     // 0: not chosen yet
  float go[NR];         // sum over own-column elements of g_z a'_L[r]
  sfor<0, NR>([&](auto r) { constexpr int R = r; go[R] = 0.f; });
  float gbo = 0.f;      // sum over own elements of g_z

  if (PF_N32_DBG_ENABLE && stamps) stamps[0] = __builtin_amdgcn_s_memrealtime();
  for (int k = 0; k < n_mine; ++k) {
    const int base = task_base(k);
    const int e = base + lane;
    const bool live = e < n;
    const TaskIn<DIM> cur = nxt;
    const bool more = k + 1 < n_mine;
    // ---- per-element scalars: one element per lane ----------------------------------------------------------
    // softplus'(z) = sigmoid(z) = 1 - exp(-softplus(z)) from the forward's stored value (= softplus(z) * scale), so the
    // output unit need not be recomputed (torch: z > 20 ? 1 : e^z / (e^z + 1), the same number to float round-off)
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
      gz = net.positive ? g * one_minus_exp_neg(cur.own * inv_scale) : g;
    }
    // power-of-two scale: max |d| S <= 2^14 with the weight bound of the image header; S only ever falls, T follows
    // The wave-wide maximum (seven DPP steps) is only formed when it can change something: S = 2^(14 - ex) with
    // gmax = m 2^ex, 0.5 <= m < 1, is smaller than the running S exactly when gmax * Srun >= 2^14, and a maximum passes
    // that test iff one lane does (Srun is a power of two: the products are exact) — one multiply, one compare and a vote
    // per task instead.
    const float gabs = fabsf(gz) * bound;
    if (Srun == 0.f || __builtin_amdgcn_ballot_w64(gabs * Srun >= 16384.0f) != 0ull) {
      const float gmax = wave_max(gabs);
      if (gmax > 0.f && gmax < 3.0e38f) {
        int ex = 0;
        (void)frexpf(gmax, &ex);
        ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
        const float Sneed = __builtin_ldexpf(1.0f, 14 - ex);
        if (Srun == 0.f) Srun = Sneed;
        else if (Sneed < Srun) {
          const float f = Sneed / Srun;      // exact: powers of two
          sfor<0, L>([&](auto l) {
            constexpr int LL = l;
            sfor<0, 16>([&](auto i) { constexpr int I = i; T[LL][I] *= f; });
          });
          Srun = Sneed;
        }
      }
    }
    const float S = Srun == 0.f ? 1.0f : Srun;
    float g0, g1;
    both_tiles(gz, g0, g1);
    float xa0, xa1, xb0, xb1;
    both_tiles(cur.x[1], xa0, xa1);
    both_tiles(cur.x[2], xb0, xb1);
    const float lam_in = cur.x[0];
    if (more) {
      // The next task's inputs leave HERE, behind the last use of this task's record (everything above consumed it: the
      // tiles below work on derived values only), so the new values can land in the registers the old ones held — issued
      // at the top of the task they needed a second set and a 19-register copy per task.  Its node ids are already here
      // (fetched two tasks ahead), so its nodal gathers leave in the same breath; the ids of the task after it start now.
      task_fetch_a<IN, GEA>(nxt, P, onet, other, mine, min(task_base(k + 1) + lane, n - 1), false);
      if (GEA) {
        nxt.nn = nn_ahead;
        task_fetch_b<IN, GEA>(nxt, P);
        nn_ahead = reinterpret_cast<const int2*>(P.mesh.conn)[min(task_base(k + 2) + lane, n - 1)];
      }
    }
    // ---- the two tiles ------------------------------------------------------------------------------------------
    auto tile_backward = [&](const typename E::template TileAct<L, true>& A, const float (&xt)[3], float gt, bool go_done) {
      // inputs of the combined tile: (kl lam, kx x, kx y, 1) resp. (kl lam, kx x, 1, 0)
      unsigned xhi[NPK], xlo[NPK];
      sfor<2, NPK>([&](auto q) { constexpr int Q = q; xhi[Q] = 0u; xlo[Q] = 0u; });
      split_pair(kl * xt[0], kx * xt[1], xhi[0], xlo[0]);
      split_pair(IN == 3 ? kx * xt[2] : 1.0f, IN == 3 ? 1.0f : 0.f, xhi[1], xlo[1]);
      // (before backward_tile: a'_L is dead there, ten registers less at the kernel's pressure peak)
      if (!go_done) sfor<0, NR>([&](auto r) { constexpr int R = r; go[R] = fmaf(gt, A.aL[R], go[R]); });
      if (!(dbg & 2)) E::template backward_tile<L, IN>(smem, scratch, lane, A, gt * S, xhi, xlo, T, wb, rd, dbg);
    };
    if constexpr (E::template bw_pair<L>()) {   // recompute both tiles together, then the two back-propagations (see bw_pair)
      const float x0[3] = {lam_in, xa0, xb0}, x1[3] = {lam_in, xa1, xb1};
      typename E::template TileAct<L, true> A0, A1;
      float p0, p1;
      E::template forward_tiles<L, IN, true>(smem, lane, x0, x1, A0, A1, p0, p1, 0, -1, go, g0, g1);
      tile_backward(A0, x0, g0, true);
      tile_backward(A1, x1, g1, true);
    } else {
      sfor<0, 2>([&](auto tt) {
        constexpr int TT = tt;
        const float xt[3] = {lam_in, TT ? xa1 : xa0, TT ? xb1 : xb0};
        typename E::template TileAct<L, true> A;
        E::template recompute_tile<L, IN>(smem, lane, xt, A);
        tile_backward(A, xt, TT ? g1 : g0, false);
      });
    }
    gbo += gz;
  }
  if (PF_N32_DBG_ENABLE && stamps) stamps[1] = __builtin_amdgcn_s_memrealtime();
  const float invS = Srun == 0.f ? 1.0f : 1.0f / Srun;
  sfor<0, L>([&](auto l) {
    constexpr int LL = l;
    sfor<0, 16>([&](auto i) { constexpr int I = i; T[LL][I] *= invS; });
  });
  if constexpr (MODE == 1) {
    bw_park<NR, L>(park + (size_t)wv * bw_park_floats<L>(), T, go, gbo, lane);
    if (PF_N32_DBG_ENABLE && stamps) stamps[2] = __builtin_amdgcn_s_memrealtime();
    return;
  } else if constexpr (MODE == 2) {
    __syncthreads();                               // every wave of the block has parked its first-phase tiles and left both loops
    first_row();
    __syncthreads();                               // the parked tiles are read
    bw_park<NR, L>(park + (size_t)wv * bw_park_floats<L>(), T, go, gbo, lane);
    __syncthreads();
    bw_row_from_parked<NR, L, IN>(P, which, hp, park, kl, kx, wt);
    if (PF_N32_DBG_ENABLE && stamps) stamps[2] = __builtin_amdgcn_s_memrealtime();
    return;
  }

  // ---- write-out: fixed-order sums over the block's waves -> this block's partial gradient row ---------------------------
  __syncthreads();
  float* stage = reinterpret_cast<float*>(wscr);   // reuses the scratch: [wave][1024]
  const int W = net.width;
  float* __restrict__ prow = P.partials + PF_PART_WG + (size_t)blockIdx.x * P.pad_total + net.pad_off;
  const int padc = pf_pad_count(hp, L);
  for (int i = threadIdx.x; i < padc; i += blockDim.x) row_store(prow + i, 0.f, wt);
  // scale of d_l relative to the true gradient after the 1/S: 4^(L-l)
  sfor<0, L>([&](auto l) {
    constexpr int LL = l + 1;                      // layer whose weight gradient tile T[LL-1] holds
    __syncthreads();
    sfor<0, 16>([&](auto i) {
      constexpr int I = i;
      const int m = (I & 3) + 8 * (I >> 2) + 4 * h;      // tile row
      stage[wv * 1024 + m * 32 + (lane & 31)] = T[LL - 1][I];
    });
    __syncthreads();
    float sc = 1.f;
    for (int k = 0; k < L - LL; ++k) sc *= 4.f;
    for (int idx = threadIdx.x; idx < 1024; idx += blockDim.x) {
      const int m = idx >> 5, c = idx & 31;
      // row m <-> image column m of the A region: unit j = 2*(m & 15) + (m >> 4)
      const int j = 2 * (m & 15) + (m >> 4);
      if ((m & 15) >= NR || j >= W) continue;
      float t = 0.f;
      for (int q = 0; q < waves; ++q) t += stage[q * 1024 + idx];
      int dst = -1;
      float k = sc;
      if (LL >= 2) {
        const int kk = 2 * (c & 15) + (c >> 4);
        k *= PF_N32_KA;
        if (c == 15) dst = pf_pad_wh(hp, LL) + j * (hp + 4) + hp;                      // bias column
        else if ((c & 15) < NR && kk < W) dst = pf_pad_wh(hp, LL) + j * (hp + 4) + kk;
      } else {
        if (c < IN) { dst = j * 4 + c; k *= c == 0 ? kl : kx; }
        else if (c == IN) dst = j * 4 + IN;                                           // bias (input 1.0)
      }
      if (dst >= 0) row_store(prow + dst, t / k, wt);
    }
  });
  // output unit row: sum over the 32 columns of each half-wave, then over the waves
  __syncthreads();
  sfor<0, NR>([&](auto r) {
    constexpr int R = r;
    float v = go[R];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((lane & 31) == 0) stage[wv * 64 + h * 16 + R] = v;
  });
  {
    const float v = pf_wave_sum(gbo);
    if (lane == 0) stage[waves * 64 + wv] = v;
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int hh = threadIdx.x >> 4, r = threadIdx.x & 15, u = 2 * r + hh;
    if (r < NR && u < W) {
      float t = 0.f;
      for (int q = 0; q < waves; ++q) t += stage[q * 64 + hh * 16 + r];
      row_store(prow + pf_pad_wo(hp, L) + u, t * (1.0f / PF_N32_KA), wt);
    }
  }
  if (threadIdx.x == 32) {
    float t = 0.f;
    for (int q = 0; q < waves; ++q) t += stage[waves * 64 + q];
    row_store(prow + pf_pad_wo(hp, L) + hp, t, wt);
  }
  if (PF_N32_DBG_ENABLE && stamps) stamps[2] = __builtin_amdgcn_s_memrealtime();
}


// GEA: this launch also computes dL/d(E*A) per element (the element adjoint) and stores it for the other net's
// backward.  Partial gradient row of the block: the padded image of pf_common.h (what theta_stage1 sums).
template <int NR, int L, int IN, bool GEA>
__global__ __launch_bounds__((Eng<NR>::template bw_threads<L, GEA>())) void k_net32_backward(pf_problem P, int which, int hp, int dbg_arg) {
  using E = Eng<NR>;
  const int dbg = PF_N32_DBG_ENABLE ? dbg_arg : 0;     // compile-time 0 in the product build (see PF_N32_DBG_ENABLE)
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int IMG = pf_n32_bytes(L), IMGPAD = (IMG + 255) & ~255;
  __shared__ int s_done;                         // the stop flag, one read per block
  if (threadIdx.x == 0) s_done = P.state->done;
  copy_image(smem, reinterpret_cast<const unsigned char*>(P.net_op + P.op_off[which]), IMG);
  __syncthreads();
  if (s_done != 0 || P.mesh.n_elems <= 0) return;
  backward_phase<NR, L, IN, GEA>(P, which, hp, dbg, smem, smem + IMGPAD, smem + IMGPAD + E::CONST_BYTES);
}

// ---- backward of BOTH nets in one launch: two phases --------------------------------------------------------------------
// Phase 1 = the young net's backward with the fused element adjoint, phase 2 = the area net's, each exactly the
// single-net kernel's work (a wave walks the same tasks in both, so the adjoint it stored in phase 1 is its own when
// it reads it back in phase 2).  The nets are NOT interleaved per task: the gradient tiles of one net plus the paired
// recompute already fill the 256 registers a wave may hold at two waves per SIMD (234 for the 20-wide net).  What the
// one launch saves is a launch floor, a prologue and the drain of the first kernel: blocks that finish phase 1 early start
// phase 2 at once.
template <int NRE, int NRA, int L, int IN>
constexpr int bw2_threads() {
  return Eng<NRE>::template bw_threads<L, true>() < Eng<NRA>::template bw_threads<L, false>()
             ? Eng<NRE>::template bw_threads<L, true>() : Eng<NRA>::template bw_threads<L, false>();
}
template <int NRE, int NRA>
constexpr int bw2_const_bytes() { return Eng<NRE>::CONST_BYTES > Eng<NRA>::CONST_BYTES ? Eng<NRE>::CONST_BYTES : Eng<NRA>::CONST_BYTES; }
template <int NRE, int NRA>
constexpr int bw2_wave_scratch() { return Eng<NRE>::WAVE_SCRATCH > Eng<NRA>::WAVE_SCRATCH ? Eng<NRE>::WAVE_SCRATCH : Eng<NRA>::WAVE_SCRATCH; }

// the phases run without a barrier between them (parked tiles) where both buckets use the compact LDS layout (same constant
// blocks, same scratch) and the block's LDS has room for the parking places; PF_BW_PARK=0: experiment build (barrier form)
#ifndef PF_BW_PARK
#define PF_BW_PARK 1
#endif
template <int NRE, int NRA>
constexpr bool bw2_parked() { return PF_BW_PARK && Eng<NRE>::COMPACT && Eng<NRA>::COMPACT; }
template <int NRE, int NRA, int L, int IN>
constexpr size_t bw2_lds_bytes() {
  return 2 * (size_t)((pf_n32_bytes(L) + 255) & ~255) + bw2_const_bytes<NRE, NRA>() +
         (size_t)(bw2_threads<NRE, NRA, L, IN>() / 64) * bw2_wave_scratch<NRE, NRA>() +
         (bw2_parked<NRE, NRA>() ? (size_t)(bw2_threads<NRE, NRA, L, IN>() / 64) * bw_park_floats<L>() * sizeof(float) : 0);
}
template <int NRE, int NRA, int L, int IN>
__global__ __launch_bounds__((bw2_threads<NRE, NRA, L, IN>())) void k_net32_backward2(pf_problem P, int hp_e, int hp_a, int dbg_arg,
                                                                                     int reduce_rows) {
  const int dbg = PF_N32_DBG_ENABLE ? dbg_arg : 0;
  const unsigned long long t_entry = PF_N32_DBG_ENABLE ? __builtin_amdgcn_s_memrealtime() : 0ull;
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int IMG = pf_n32_bytes(L), IMGPAD = (IMG + 255) & ~255;
  __shared__ int s_done;
  if (threadIdx.x == 0) s_done = P.state->done;
  copy_images2(smem, reinterpret_cast<const unsigned char*>(P.net_op + P.op_off[0]), smem + IMGPAD,
               reinterpret_cast<const unsigned char*>(P.net_op + P.op_off[1]), IMG);
  __syncthreads();
  if (s_done != 0 || P.mesh.n_elems <= 0) return;
  unsigned char* cst = smem + 2 * IMGPAD;
  unsigned char* wscr = cst + bw2_const_bytes<NRE, NRA>();
  // (diagnostic build, PF_N32_DBG=16: per-wave time stamps into pf_problem.u_alt behind the forward launch's: entry |
  //  E loop entered | E loop left | E row written | A loop entered | A loop left | A row written; tools/fwd2_stamps.py)
  unsigned long long* st = nullptr;
  if (PF_N32_DBG_ENABLE && (dbg & 16) && (threadIdx.x & 63) == 0) {
    st = reinterpret_cast<unsigned long long*>(P.u_alt) + 65536 + (size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8;
    st[0] = t_entry;
  }
  if constexpr (bw2_parked<NRE, NRA>()) {
    // no barrier between the phases: a wave parks its young-net tiles and walks on (see backward_phase, MODE)
    float* park = reinterpret_cast<float*>(wscr + (size_t)(blockDim.x >> 6) * bw2_wave_scratch<NRE, NRA>());
    const float kl_e = pf_n32_lam_scale(P.lam), kx_e = __builtin_ldexpf(1.0f, P.coord_exp);
    const bool wt = (reduce_rows & 1) != 0;
    auto first_row = [&]() { bw_row_from_parked<NRE, L, IN>(P, 0, hp_e, park, kl_e, kx_e, wt); };
    backward_phase<NRE, L, IN, true, 1>(P, 0, hp_e, dbg, smem, cst, wscr, wt, st ? st + 1 : nullptr, park, 0, reduce_rows >> 1);
    backward_phase<NRA, L, IN, false, 2>(P, 1, hp_a, dbg, smem + IMGPAD, cst, wscr, wt, st ? st + 4 : nullptr, park, first_row,
                                         reduce_rows >> 1);
  } else {
    backward_phase<NRE, L, IN, true>(P, 0, hp_e, dbg, smem, cst, wscr, (reduce_rows & 1) != 0, st ? st + 1 : nullptr, nullptr, 0,
                                     reduce_rows >> 1);
    __syncthreads();                             // the write-out staging of phase 1 is read; scratch and constants are re-initialised
    backward_phase<NRA, L, IN, false>(P, 1, hp_a, dbg, smem + IMGPAD, cst, wscr, (reduce_rows & 1) != 0, st ? st + 4 : nullptr,
                                      nullptr, 0, reduce_rows >> 1);
  }
  // reduce_rows: the launch is also theta stage 1 (the last block of every row group sums the group's rows)
  if (reduce_rows & 1) rows_reduce_last(P, (int)gridDim.x);
}

template <int L, int IN>
int launch_fwd(const pf_problem* p, int which, hipStream_t s, int s2_half) {
  constexpr int NR = PF_NR;
  const int n = p->mesh.n_elems;
  int nb = (n + FW_THREADS - 1) / FW_THREADS;
  static const int cap = getenv("PF_FWD32_BLOCKS") ? atoi(getenv("PF_FWD32_BLOCKS")) : 256;   // one block per CU
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  static const int dbg = getenv("PF_N32_DBG") ? atoi(getenv("PF_N32_DBG")) : 0;   // timing experiments only
  const int ws = p->elem_k != nullptr ? 1 : 0;
  const size_t lds = s2_half < 0 ? (size_t)pf_n32_bytes(L)
                                 : (size_t)((pf_n32_bytes(L) + 255) & ~255) + (size_t)p->n_theta_active * sizeof(float);
  if (lds > 64000) { pf_set_error("too many trainable parameters for the fused theta update"); return PF_ERR_UNSUPPORTED; }
  hipLaunchKernelGGL((k_net32_forward<NR, L, IN>), dim3(nb), dim3(FW_THREADS), lds, s, *p, which, dbg, ws, s2_half);
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_HIP;
}

template <int L, int IN, bool GEA>
int launch_bwd_t(const pf_problem* p, int which, hipStream_t s) {
  const int nb = pf_net_blocks(p);
  const int hp = ((p->net[which].width + 3) / 4) * 4;
  constexpr int NR = PF_NR;
  using E = Eng<NR>;
  constexpr int BW_THREADS = E::template bw_threads<L, GEA>();
  const size_t lds = ((pf_n32_bytes(L) + 255) & ~255) + E::CONST_BYTES + (size_t)(BW_THREADS / 64) * E::WAVE_SCRATCH;
  static_assert(E::WAVE_SCRATCH >= 4096 + 512, "write-out staging must fit the wave scratch");
  static const int dbg = getenv("PF_N32_DBG") ? atoi(getenv("PF_N32_DBG")) : 0;   // timing experiments only
  hipLaunchKernelGGL((k_net32_backward<NR, L, IN, GEA>), dim3(nb), dim3(BW_THREADS), lds, s, *p, which, hp, dbg);
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_HIP;
}
template <int L, int IN>
int launch_bwd(const pf_problem* p, int which, hipStream_t s) { return launch_bwd_t<L, IN, false>(p, which, s); }
template <int L, int IN>
int launch_bwd_gea(const pf_problem* p, int which, hipStream_t s) { return launch_bwd_t<L, IN, true>(p, which, s); }

// fused forward of both nets: the E net's bucket is this translation unit's PF_NR, the A net's bucket is dispatched here
template <int NRA, int L, int IN>
int launch_fwd2_t(const pf_problem* p, hipStream_t s, const pf_fwd2_opts& o) {
  const int s2_half = o.s2_half, gu_nb = o.gu_nb;
  constexpr int NRE = PF_NR;
  const int n = p->mesh.n_elems;
  const int nb = pf_n32_fwd2_blocks(n);
  static const int dbg = getenv("PF_N32_DBG") ? atoi(getenv("PF_N32_DBG")) : 0;
  const size_t lds = 2 * (size_t)((pf_n32_bytes(L) + 255) & ~255) + (s2_half < 0 ? 0 : (size_t)p->n_theta_active * sizeof(float));
  if (lds > 64000) { pf_set_error("too many trainable parameters for the fused theta update"); return PF_ERR_UNSUPPORTED; }
  // PF_FWD_QUEUE=0: experiment knob (round 2's lockstep rounds instead of the per-block task queue)
  static const int queue_knob = getenv("PF_FWD_QUEUE") ? atoi(getenv("PF_FWD_QUEUE")) : 1;
  static const int every_add = getenv("PF_GU_EVERY_ADD") ? atoi(getenv("PF_GU_EVERY_ADD")) : 0;
  const int queue = (gu_nb > 0 || queue_knob) ? (1 | ((every_add & 0xff) << 8)) : 0;       // (the displacement update rides on the queue form only)
  if (gu_nb > 0 && !pf_n32_fwd2_can_update_u(p, gu_nb)) { pf_set_error("fused forward: displacement update not possible on this problem"); return PF_ERR_UNSUPPORTED; }
  hipLaunchKernelGGL((k_net32_forward2<NRE, NRA, L, IN>), dim3(nb), dim3(FW_THREADS), lds, s, *p, dbg, s2_half >= 0 && o.calc_index ? s2_half + 2 : s2_half, queue, gu_nb, o.gu_k);
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_HIP;
}
template <int L, int IN>
int launch_fwd2(const pf_problem* p, hipStream_t s, const pf_fwd2_opts& o) {
  switch (pf_net32_bucket(p->net[1].width)) {
    case 2: return launch_fwd2_t<2, L, IN>(p, s, o);
    case 4: return launch_fwd2_t<4, L, IN>(p, s, o);
    case 6: return launch_fwd2_t<6, L, IN>(p, s, o);
    case 8: return launch_fwd2_t<8, L, IN>(p, s, o);
    case 10: return launch_fwd2_t<10, L, IN>(p, s, o);
    case 12: return launch_fwd2_t<12, L, IN>(p, s, o);
    case 15: return launch_fwd2_t<15, L, IN>(p, s, o);
  }
  pf_set_error("MFMA32 engine: area net width outside 1..30");
  return PF_ERR_UNSUPPORTED;
}

// fused backward of both nets (two phases): E net of this translation unit's bucket, A net's bucket dispatched here
template <int NRA, int L, int IN>
int launch_bwd2_t(const pf_problem* p, hipStream_t s, int reduce_rows) {
  constexpr int NRE = PF_NR;
  const int nb = pf_net_blocks(p);
  const int hp_e = ((p->net[0].width + 3) / 4) * 4, hp_a = ((p->net[1].width + 3) / 4) * 4;
  constexpr int THREADS = bw2_threads<NRE, NRA, L, IN>();
  const size_t lds = bw2_lds_bytes<NRE, NRA, L, IN>();
  static_assert(bw2_lds_bytes<NRE, NRA, L, IN>() <= 160 * 1024, "fused backward: LDS budget");
  static const int dbg = getenv("PF_N32_DBG") ? atoi(getenv("PF_N32_DBG")) : 0;
  // PF_BW_SHIFT: eighths of the younger wave's tasks the older wave of its SIMD takes over (backward_phase, shift8; bits 1..
  // of the kernel's reduce_rows argument)
  static const int shift8 = getenv("PF_BW_SHIFT") ? atoi(getenv("PF_BW_SHIFT")) : 2;     // measured best of 0..3 (r03_ab.txt)
  hipLaunchKernelGGL((k_net32_backward2<NRE, NRA, L, IN>), dim3(nb), dim3(THREADS), lds, s, *p, hp_e, hp_a, dbg, (reduce_rows ? 1 : 0) | (shift8 << 1));
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_HIP;
}
template <int L, int IN>
int launch_bwd2(const pf_problem* p, hipStream_t s, int reduce_rows) {
  switch (pf_net32_bucket(p->net[1].width)) {
    case 2: return launch_bwd2_t<2, L, IN>(p, s, reduce_rows);
    case 4: return launch_bwd2_t<4, L, IN>(p, s, reduce_rows);
    case 6: return launch_bwd2_t<6, L, IN>(p, s, reduce_rows);
    case 8: return launch_bwd2_t<8, L, IN>(p, s, reduce_rows);
    case 10: return launch_bwd2_t<10, L, IN>(p, s, reduce_rows);
    case 12: return launch_bwd2_t<12, L, IN>(p, s, reduce_rows);
    case 15: return launch_bwd2_t<15, L, IN>(p, s, reduce_rows);
  }
  pf_set_error("MFMA32 engine: area net width outside 1..30");
  return PF_ERR_UNSUPPORTED;
}

}  // namespace

#if PF_PREC == 1
#define PF_N32_SYM(kind) PF_CAT(PF_CAT(pf_launch_net32b_, kind), PF_NR)
#else
#define PF_N32_SYM(kind) PF_CAT(PF_CAT(pf_launch_net32_, kind), PF_NR)
#endif

// PF_N32_PART 0: the single-net kernels of this bucket; 1: the fused two-net kernels whose E net is of this bucket
// (separate translation units: the build compiles them side by side)
#ifndef PF_N32_PART
#define PF_N32_PART 0
#endif

#if PF_N32_PART == 0
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

int PF_N32_SYM(forward_)(const pf_problem* p, int which, hipStream_t s, int s2_half) {
  const pf_net& net = p->net[which];
  const int L = net.n_hidden, IN = net.in_dim;
  if (IN == 3) {
    if (L == 1) return launch_fwd<1, 3>(p, which, s, s2_half);
    if (L == 2) return launch_fwd<2, 3>(p, which, s, s2_half);
    if (L == 3) return launch_fwd<3, 3>(p, which, s, s2_half);
  } else if (IN == 2) {
    if (L == 1) return launch_fwd<1, 2>(p, which, s, s2_half);
    if (L == 2) return launch_fwd<2, 2>(p, which, s, s2_half);
    if (L == 3) return launch_fwd<3, 2>(p, which, s, s2_half);
  }
  pf_set_error("net shape outside the compiled menu (in_dim 2|3, hidden layers 1..3)");
  return PF_ERR_UNSUPPORTED;
}
int PF_N32_SYM(backward_)(const pf_problem* p, int which, hipStream_t s) {
  PF_DISPATCH(launch_bwd)
}
int PF_N32_SYM(backward_gea_)(const pf_problem* p, int which, hipStream_t s) {
  PF_DISPATCH(launch_bwd_gea)
}
#else
// both nets enabled, same number of hidden layers and the same inputs (pf_api.hip checks): young net of THIS bucket
int PF_N32_SYM(forward2_)(const pf_problem* p, hipStream_t s, const pf_fwd2_opts& o) {
  const int L = p->net[0].n_hidden, IN = p->net[0].in_dim;
  if (IN == 3) {
    if (L == 1) return launch_fwd2<1, 3>(p, s, o);
    if (L == 2) return launch_fwd2<2, 3>(p, s, o);
    if (L == 3) return launch_fwd2<3, 3>(p, s, o);
  } else if (IN == 2) {
    if (L == 1) return launch_fwd2<1, 2>(p, s, o);
    if (L == 2) return launch_fwd2<2, 2>(p, s, o);
    if (L == 3) return launch_fwd2<3, 2>(p, s, o);
  }
  pf_set_error("net shape outside the compiled menu (in_dim 2|3, hidden layers 1..3)");
  return PF_ERR_UNSUPPORTED;
}
// Two hidden layers only (the reference's SimpleNN default and every example): with one or three the two phases in one
// kernel no longer fit the register budget of their block shapes without spilling (checked in the compiler's asm), and
// a spill reload in the task loop costs more than a launch boundary — those shapes keep the two launches.
int PF_N32_SYM(backward2_)(const pf_problem* p, hipStream_t s, int reduce_rows) {
  const int L = p->net[0].n_hidden, IN = p->net[0].in_dim;
  if (L == 2 && IN == 3) return launch_bwd2<2, 3>(p, s, reduce_rows);
  if (L == 2 && IN == 2) return launch_bwd2<2, 2>(p, s, reduce_rows);
  pf_set_error("fused backward: two hidden layers only");
  return PF_ERR_UNSUPPORTED;
}
#endif
