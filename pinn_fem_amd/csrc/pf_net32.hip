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

#ifndef PF_NR
#error "compile with -DPF_NR=<registers per lane>"
#endif
#define PF_CAT2(a, b) a##b
#define PF_CAT(a, b) PF_CAT2(a, b)

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int NR = PF_NR;                 // accumulator registers per lane that carry real units
constexpr int KS = NR > 8 ? 2 : 1;        // k-steps of 16 units
constexpr int NPK = 4 * KS;               // packed f16 pairs per operand set (pairs >= (NR+1)/2 are zero)
constexpr int NPR = (NR + 1) / 2;         // pairs that carry data
static_assert(NR >= 1 && NR <= 15, "PF_NR out of range");

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

// two values -> packed hi and lo f16 pairs
__device__ __forceinline__ void split_pair(float a0, float a1, unsigned& hi, unsigned& lo) {
  const h2 H = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a0, a1));
  const h2 Lo = h2{(_Float16)(a0 - (float)H[0]), (_Float16)(a1 - (float)H[1])};
  hi = __builtin_bit_cast(unsigned, H);
  lo = __builtin_bit_cast(unsigned, Lo);
}

// acc += (Ahi + Alo)(Bhi + Blo) without the lo*lo term; small terms first
__device__ __forceinline__ f32x16 mfma3(f32x16 acc, h8 ahi, h8 alo, h8 bhi, h8 blo) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc, 0, 0, 0);
  return acc;
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

// wave maximum of a non-negative value, uniform result
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// ---- activations of one tile kept for the backward pass -------------------------------------------------
template <int L, bool BWD>
struct TileAct {
  unsigned hi[L][NPK], lo[L][NPK];   // a'_l = KA tanh(z_l) as packed f16 pairs (pair q = registers 2q, 2q+1)
  float t[BWD ? L : 1][NR];          // r (1 - r) with tanh = 1 - 2 r: (1 - tanh^2) / 4
  float aL[NR];                      // a'_L in float (output unit and its gradient)
};

// tanh of the NR pre-activations z[r] * cz; writes the packed pairs (and t, aL) of layer LL
template <int L, bool BWD, int LL>
__device__ __forceinline__ void activate(TileAct<L, BWD>& A, const float (&z)[NR], float cz) {
  float a[2 * NPR];
  sfor<0, NR>([&](auto r) {
    constexpr int R = r;
    const float e = __builtin_amdgcn_exp2f(z[R] * cz);
    const float q = __builtin_amdgcn_rcpf(e + 1.0f);
    a[R] = fmaf(-2.0f * PF_N32_KA, q, PF_N32_KA);
    if constexpr (BWD) A.t[LL - 1][R] = fmaf(-q, q, q);
    if constexpr (LL == L) A.aL[R] = a[R];
  });
  if constexpr (NR & 1) a[NR] = 0.f;
  sfor<0, NPK>([&](auto q) {
    constexpr int Q = q;
    if constexpr (Q < NPR) split_pair(a[2 * Q], a[2 * Q + 1], A.hi[LL - 1][Q], A.lo[LL - 1][Q]);
    else { A.hi[LL - 1][Q] = 0u; A.lo[LL - 1][Q] = 0u; }
  });
}

// forward through the hidden layers for ONE tile; x = (load factor, coordinates) of the tile's element on this
// lane's column.  Returns this lane's partial sum of the output unit, KA * sum_r wo[2r+h] a_L[2r+h].
template <int L, int IN, bool BWD>
__device__ __forceinline__ float forward_tile(const unsigned char* __restrict__ img, int lane, const float (&x)[3],
                                              TileAct<L, BWD>& A) {
  const int h = lane >> 5;
  constexpr float C2 = 2.8853900817779268f;   // 2 log2(e)
  // layer 1 on the vector ALU in float, the reference's order: bias, then the inputs ascending
  {
    const float4* __restrict__ w1 = reinterpret_cast<const float4*>(img + pf_n32_off_w1());
    float z[NR];
    sfor<0, NR>([&](auto r) {
      constexpr int R = r;
      const float4 w = w1[R * 2 + h];
      float acc;
      if constexpr (IN == 3) {
        acc = fmaf(w.w, 1.0f, 0.f);
        acc = fmaf(w.x, x[0], acc);
        acc = fmaf(w.y, x[1], acc);
        acc = fmaf(w.z, x[2], acc);
      } else {
        acc = fmaf(w.z, 1.0f, 0.f);
        acc = fmaf(w.x, x[0], acc);
        acc = fmaf(w.y, x[1], acc);
      }
      z[R] = acc;
    });
    activate<L, BWD, 1>(A, z, C2);
  }
  // hidden layers 2..L on the matrix cores: z' = KA KW z, bias as the initial accumulator
  sfor<2, L + 1>([&](auto l) {
    constexpr int LL = l;
    const float4* __restrict__ bias = reinterpret_cast<const float4*>(img + pf_n32_off_bias(LL)) + h * 4;
    const h8* __restrict__ af = reinterpret_cast<const h8*>(img + pf_n32_off_af(LL)) + lane;
    f32x16 acc;
    {
      const float4 b0 = bias[0], b1 = bias[1], b2 = bias[2], b3 = bias[3];
      acc = f32x16{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
    }
    sfor<0, KS>([&](auto s) {
      constexpr int S = s;
      const h8 ahi = af[(0 * 2 + S) * 64], alo = af[(1 * 2 + S) * 64];
      const h8 bhi = as_h8(A.hi[LL - 2][4 * S], A.hi[LL - 2][4 * S + 1], A.hi[LL - 2][4 * S + 2], A.hi[LL - 2][4 * S + 3]);
      const h8 blo = as_h8(A.lo[LL - 2][4 * S], A.lo[LL - 2][4 * S + 1], A.lo[LL - 2][4 * S + 2], A.lo[LL - 2][4 * S + 3]);
      acc = mfma3(acc, ahi, alo, bhi, blo);
    });
    float z[NR];
    sfor<0, NR>([&](auto r) { constexpr int R = r; z[R] = acc[R]; });
    activate<L, BWD, LL>(A, z, C2 / (PF_N32_KA * PF_N32_KW));
  });
  // output unit: this lane's share
  const float4* __restrict__ wo = reinterpret_cast<const float4*>(img + pf_n32_off_wo()) + h * 4;
  float wv[16];
  sfor<0, (NR + 3) / 4>([&](auto q) {
    constexpr int Q = q;
    const float4 w = wo[Q];
    wv[4 * Q] = w.x; wv[4 * Q + 1] = w.y; wv[4 * Q + 2] = w.z; wv[4 * Q + 3] = w.w;
  });
  float p = 0.f;
  sfor<0, NR>([&](auto r) { constexpr int R = r; p = fmaf(wv[R], A.aL[R], p); });
  return p;
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

// ---- forward kernel ----------------------------------------------------------------------------------------
// Each wave walks 64-element tasks.  Lanes past the end work on the last element again (same inputs, same value)
// and do not store.
template <int L, int IN>
__global__ __launch_bounds__(256) void k_net32_forward(pf_problem P, int which) {
  extern __shared__ __align__(16) unsigned char smem[];
  const pf_net net = P.net[which];
  copy_image(smem, reinterpret_cast<const unsigned char*>(P.net_op + P.op_off[which]), pf_n32_bytes(L));
  float* __restrict__ out = which == 0 ? P.prop_e : P.prop_a;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const int n = P.mesh.n_elems;
  const int stride = gridDim.x * waves * 64;
  int base = (blockIdx.x * waves + wv) * 64;
  float xn[3];
  if (n > 0) load_x<IN>(xn, P, min(base + lane, n - 1));
  __syncthreads();
  if (P.state->done || n <= 0) return;
  const float bo = reinterpret_cast<const float*>(smem + pf_n32_off_bo())[0];
  for (; base < n; base += stride) {
    const int e = base + lane;
    float x0[3], x1[3];
    sfor<0, 3>([&](auto c) { constexpr int C = c; both_tiles(xn[C], x0[C], x1[C]); });
    if (base + stride < n) load_x<IN>(xn, P, min(e + stride, n - 1));
    TileAct<L, false> A0, A1;
    const float p0 = forward_tile<L, IN, false>(smem, lane, x0, A0);
    const float p1 = forward_tile<L, IN, false>(smem, lane, x1, A1);
    const float z = own_total(p0, p1) * (1.0f / PF_N32_KA) + bo;
    if (e < n) out[e] = (net.positive ? pf_softplus(z) : z) * net.scale;
  }
}

// ---- backward kernel ---------------------------------------------------------------------------------------
// LDS per wave: three regions of [2 split][2 chunk] x 1152 B: A side (d rows), B side (activation columns), X (inputs).
// A lane writes its 16 f16 of a chunk at lane*16 + (lane>>5)*64; chunk 1 (registers 8..15) sits 1152 B further:
// with these strides both the 16-B writes and the transposed 8-B reads are bank-conflict free.
constexpr int CHUNK = 1152;
constexpr int REGION = 4 * CHUNK;           // [split][chunk]
constexpr int WAVE_SCRATCH = 3 * REGION;

__device__ __forceinline__ int lane_slot(int lane) { return lane * 16 + (lane >> 5) * 64; }

// lane's pairs (hi or lo) of registers 0..15 -> its slot of the region's split `sp`
__device__ __forceinline__ void write_rows(unsigned char* region, int sp, int lane, const unsigned (&pk)[NPK]) {
  unsigned char* p = region + sp * 2 * CHUNK + lane_slot(lane);
  *reinterpret_cast<u32x4*>(p) = u32x4{pk[0], pk[1], pk[2], pk[3]};
  if constexpr (NR > 12) *reinterpret_cast<u32x4*>(p + CHUNK) = u32x4{pk[4], pk[5], pk[6], pk[7]};
  else if constexpr (NR > 8) *reinterpret_cast<u32x2*>(p + CHUNK) = u32x2{pk[4], pk[5]};
}

// MFMA operand (A: rows = image columns, B: columns = image columns; k = 16 elements of step ks) read transposed.
// Group g = lane>>4 reads the 4x16 block rows (elements) e0..e0+3, columns 16*(g&1)..+15; lane 4q+p of the
// group supplies the address of row q, columns 4p..4p+3 (cdna_hip_programming.md T10).
__device__ __forceinline__ h8 read_operand(const unsigned char* region, int sp, int ks, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int hsrc = g & 1, hl = g >> 1;
  const unsigned char* base = region + sp * 2 * CHUNK + (p >> 1) * CHUNK + hsrc * (32 * 16 + 64) + 8 * (p & 1);
  const int e0 = 16 * ks + 8 * hl + q;
  const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s4v*)(base + e0 * 16));
  const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s4v*)(base + (e0 + 4) * 16));
  typedef short s8v __attribute__((ext_vector_type(8)));
  return __builtin_bit_cast(h8, s8v{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}

template <int DIM>
struct TaskIn {
  float x[3];
  float oth, gea;
  int2 nn;
  ElemGeo g;
  float ui[2], uj[2], gi[2], gj[2];
};

template <int IN, bool GEA>
__device__ __forceinline__ void task_fetch_a(TaskIn<IN - 1>& t, const pf_problem& P, const pf_net& onet,
                                             const float* __restrict__ other, int e) {
  load_x<IN>(t.x, P, e);
  t.oth = onet.enabled ? other[e] : onet.scale;
  t.gea = 0.f;
  if (GEA) {
    t.nn = reinterpret_cast<const int2*>(P.mesh.conn)[e];
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
  ke_rows_times<DIM>(t.g, 1.f, 0, t.ui, t.uj, pu0, fe_mode);
  ke_rows_times<DIM>(t.g, 1.f, 1, t.ui, t.uj, pu1, fe_mode);
  float gs = 0.f;
#pragma unroll
  for (int c = 0; c < DIM; ++c) gs = fmaf(t.gi[c], pu0[c], gs);
#pragma unroll
  for (int c = 0; c < DIM; ++c) gs = fmaf(t.gj[c], pu1[c], gs);
  return gs / t.g.l0;
}

// backward of one tile: d_L from g_z S, back-propagation, and the tile's contribution to the gradient products
// T[0] = combined tile (rows d_1, columns inputs), T[l-1] = rows d_l, columns a_{l-1} (l = 2..L).
template <int L, int IN>
__device__ __forceinline__ void backward_tile(const unsigned char* __restrict__ img, unsigned char* scratch, int lane,
                                              const TileAct<L, true>& A, float gs /* g_z S of the column's element */,
                                              const unsigned (&xhi)[NPK], const unsigned (&xlo)[NPK],
                                              f32x16 (&T)[L]) {
  const int h = lane >> 5;
  unsigned char* regA = scratch;
  unsigned char* regB = scratch + REGION;
  unsigned char* regX = scratch + 2 * REGION;
  // d_L[r] = (4 wo[2r+h] g_z S) t_L[r]
  float d[2 * NPR];
  {
    const float4* __restrict__ wo = reinterpret_cast<const float4*>(img + pf_n32_off_wo()) + h * 4;
    float wv[16];
    sfor<0, (NR + 3) / 4>([&](auto q) {
      constexpr int Q = q;
      const float4 w = wo[Q];
      wv[4 * Q] = w.x; wv[4 * Q + 1] = w.y; wv[4 * Q + 2] = w.z; wv[4 * Q + 3] = w.w;
    });
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
    // gradient tile of layer LL: rows d_LL through LDS; columns a_{LL-1} (LL >= 2) or the inputs (LL == 1)
    write_rows(regA, 0, lane, dhi);
    write_rows(regA, 1, lane, dlo);
    unsigned char* regC = regX;
    if constexpr (LL >= 2) {
      write_rows(regB, 0, lane, A.hi[LL - 2]);
      write_rows(regB, 1, lane, A.lo[LL - 2]);
      regC = regB;
    } else {
      // inputs of the element: registers 0..3 of the lower half-wave's slot (columns 0..3)
      if (h == 0) {
        *reinterpret_cast<u32x2*>(regX + lane_slot(lane)) = u32x2{xhi[0], xhi[1]};
        *reinterpret_cast<u32x2*>(regX + 2 * CHUNK + lane_slot(lane)) = u32x2{xlo[0], xlo[1]};
      }
    }
    // back-propagation to layer LL-1 while the LDS round trip is in flight
    f32x16 acc = zero16();
    if constexpr (LL >= 2) {
      const h8* __restrict__ ab = reinterpret_cast<const h8*>(img + pf_n32_off_ab(LL)) + lane;
      sfor<0, KS>([&](auto ks) {
        constexpr int S = ks;
        const h8 ahi = ab[(0 * 2 + S) * 64], alo = ab[(1 * 2 + S) * 64];
        const h8 bhi = as_h8(dhi[4 * S], dhi[4 * S + 1], dhi[4 * S + 2], dhi[4 * S + 3]);
        const h8 blo = as_h8(dlo[4 * S], dlo[4 * S + 1], dlo[4 * S + 2], dlo[4 * S + 3]);
        acc = mfma3(acc, ahi, alo, bhi, blo);
      });
    }
    // the wave's own LDS writes are visible to its own later reads (in-order); tell the compiler only
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    sfor<0, 2>([&](auto ks) {
      constexpr int S = ks;
      const h8 ahi = read_operand(regA, 0, S, lane), alo = read_operand(regA, 1, S, lane);
      const h8 bhi = read_operand(regC, 0, S, lane), blo = read_operand(regC, 1, S, lane);
      T[LL - 1] = mfma3(T[LL - 1], ahi, alo, bhi, blo);
    });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if constexpr (LL >= 2) {
      sfor<0, NR>([&](auto r) { constexpr int R = r; d[R] = acc[R] * A.t[LL - 2][R]; });
    }
  });
}

constexpr int BW_THREADS = 512;

// GEA: this launch also computes dL/d(E*A) per element (the element adjoint) and stores it for the other net's
// backward.  Partial gradient row of the block: the padded image of pf_common.h (what theta_stage1 sums).
template <int L, int IN, bool GEA>
__global__ __launch_bounds__(BW_THREADS, 2) void k_net32_backward(pf_problem P, int which, int hp) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int DIM = IN - 1;
  constexpr int IMG = pf_n32_bytes(L);
  const pf_net net = P.net[which];
  const pf_net onet = P.net[1 - which];
  const float* __restrict__ other = which == 0 ? P.prop_a : P.prop_e;
  const int n = P.mesh.n_elems;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const int h = lane >> 5;
  copy_image(smem, reinterpret_cast<const unsigned char*>(P.net_op + P.op_off[which]), IMG);
  unsigned char* scratch = smem + ((IMG + 127) & ~127) + wv * WAVE_SCRATCH;
  // zero the scratch (padding columns stay zero for the whole kernel), then the bias column of the B region:
  // KA in column 15 of the lower half-wave's slots (hi image, chunk 1, bytes 14..15)
  for (int i = lane; i < WAVE_SCRATCH / 16; i += 64) reinterpret_cast<uint4*>(scratch)[i] = make_uint4(0, 0, 0, 0);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (h == 0) *reinterpret_cast<_Float16*>(scratch + REGION + CHUNK + lane_slot(lane) + 14) = (_Float16)PF_N32_KA;

  const int stride = gridDim.x * waves * 64;
  int base = (blockIdx.x * waves + wv) * 64;
  TaskIn<DIM> nxt;
  if (n > 0) task_fetch_a<IN, GEA>(nxt, P, onet, other, min(base + lane, n - 1));
  __syncthreads();
  if (P.state->done != 0 || n <= 0) return;      // block-uniform
  task_fetch_b<IN, GEA>(nxt, P);

  const float bound = reinterpret_cast<const float*>(smem)[0];
  const float bo = reinterpret_cast<const float*>(smem + pf_n32_off_bo())[0];
  const float kx = __builtin_ldexpf(1.0f, P.coord_exp);
  f32x16 T[L];          // sum over tasks of (tile products) / S
  sfor<0, L>([&](auto l) { constexpr int LL = l; T[LL] = zero16(); });
  float go[NR];         // sum over own-column elements of g_z a'_L[r]
  sfor<0, NR>([&](auto r) { constexpr int R = r; go[R] = 0.f; });
  float gbo = 0.f;      // sum over own elements of g_z

  {
    for (; base < n; base += stride) {
      const int e = base + lane;
      const bool live = e < n;
      const TaskIn<DIM> cur = nxt;
      const bool more = base + stride < n;
      if (more) task_fetch_a<IN, GEA>(nxt, P, onet, other, min(e + stride, n - 1));
      // ---- forward recompute of both tiles ---------------------------------------------------------------
      float x0[3], x1[3];
      sfor<0, 3>([&](auto c) { constexpr int C = c; both_tiles(cur.x[C], x0[C], x1[C]); });
      TileAct<L, true> A0, A1;
      const float p0 = forward_tile<L, IN, true>(smem, lane, x0, A0);
      const float p1 = forward_tile<L, IN, true>(smem, lane, x1, A1);
      const float z = own_total(p0, p1) * (1.0f / PF_N32_KA) + bo;
      if (more) task_fetch_b<IN, GEA>(nxt, P);
      // ---- per-element scalars: one element per lane ----------------------------------------------------------
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
      // power-of-two scale of this task: max |d| S <= 2^14
      const float gmax = wave_max(fabsf(gz)) * bound;
      int ex = 0;
      if (gmax > 0.f && gmax < 3.0e38f) (void)frexpf(gmax, &ex);
      ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
      const float S = __builtin_ldexpf(1.0f, 14 - ex), invS = __builtin_ldexpf(1.0f, ex - 14);
      const float gzS = gz * S;
      float gs0, gs1;
      both_tiles(gzS, gs0, gs1);
      // inputs of the combined tile: (KL lam, kx x, kx y, 1) resp. (KL lam, kx x, 1, 0), own element
      unsigned xhi[NPK], xlo[NPK];
      {
        const float i0 = PF_N32_KL * cur.x[0], i1 = kx * cur.x[1];
        const float i2 = IN == 3 ? kx * cur.x[2] : 1.0f, i3 = IN == 3 ? 1.0f : 0.f;
        split_pair(i0, i1, xhi[0], xlo[0]);
        split_pair(i2, i3, xhi[1], xlo[1]);
        sfor<2, NPK>([&](auto q) { constexpr int Q = q; xhi[Q] = 0u; xlo[Q] = 0u; });
      }
      // the inputs image wants, in the LOWER half-wave's slot c, the inputs of the tile's element c
      unsigned xh0[NPK], xl0[NPK], xh1[NPK], xl1[NPK];
      sfor<0, NPK>([&](auto q) { constexpr int Q = q; xh0[Q] = xl0[Q] = xh1[Q] = xl1[Q] = 0u; });
      sfor<0, 2>([&](auto q) {
        constexpr int Q = q;
        float a0, a1;
        both_tiles(__builtin_bit_cast(float, xhi[Q]), a0, a1);
        xh0[Q] = __builtin_bit_cast(unsigned, a0); xh1[Q] = __builtin_bit_cast(unsigned, a1);
        both_tiles(__builtin_bit_cast(float, xlo[Q]), a0, a1);
        xl0[Q] = __builtin_bit_cast(unsigned, a0); xl1[Q] = __builtin_bit_cast(unsigned, a1);
      });
      // ---- backward of both tiles into fresh product tiles ---------------------------------------------------------
      f32x16 F[L];
      sfor<0, L>([&](auto l) { constexpr int LL = l; F[LL] = zero16(); });
      backward_tile<L, IN>(smem, scratch, lane, A0, gs0, xh0, xl0, F);
      backward_tile<L, IN>(smem, scratch, lane, A1, gs1, xh1, xl1, F);
      sfor<0, L>([&](auto l) {
        constexpr int LL = l;
        sfor<0, 16>([&](auto i) { constexpr int I = i; T[LL][I] = fmaf(F[LL][I], invS, T[LL][I]); });
      });
      // output unit: g_z a'_L, the lane's share of both tiles
      const float g0 = gs0 * invS, g1 = gs1 * invS;
      sfor<0, NR>([&](auto r) {
        constexpr int R = r;
        go[R] = fmaf(g0, A0.aL[R], go[R]);
        go[R] = fmaf(g1, A1.aL[R], go[R]);
      });
      gbo += gz;
    }
  }

  // ---- write-out: fixed-order sums over the block's waves -> this block's partial gradient row ---------------------------
  __syncthreads();
  float* stage = reinterpret_cast<float*>(smem + ((IMG + 127) & ~127));   // reuses the scratch: [wave][1024]
  const int W = net.width;
  float* __restrict__ prow = P.partials + PF_PART_WG + (size_t)blockIdx.x * P.pad_total + net.pad_off;
  const int padc = pf_pad_count(hp, L);
  for (int i = threadIdx.x; i < padc; i += blockDim.x) prow[i] = 0.f;
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
        if (c < IN) { dst = j * 4 + c; k *= c == 0 ? PF_N32_KL : kx; }
        else if (c == IN) dst = j * 4 + IN;                                           // bias (input 1.0)
      }
      if (dst >= 0) prow[dst] = t / k;
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
      prow[pf_pad_wo(hp, L) + u] = t * (1.0f / PF_N32_KA);
    }
  }
  if (threadIdx.x == 32) {
    float t = 0.f;
    for (int q = 0; q < waves; ++q) t += stage[waves * 64 + q];
    prow[pf_pad_wo(hp, L) + hp] = t;
  }
}

template <int L, int IN>
int launch_fwd(const pf_problem* p, int which, hipStream_t s) {
  const int n = p->mesh.n_elems;
  int nb = (n + 255) / 256;
  static const int cap = getenv("PF_FWD32_BLOCKS") ? atoi(getenv("PF_FWD32_BLOCKS")) : 1024;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL((k_net32_forward<L, IN>), dim3(nb), dim3(256), pf_n32_bytes(L), s, *p, which);
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_HIP;
}

template <int L, int IN, bool GEA>
int launch_bwd_t(const pf_problem* p, int which, hipStream_t s) {
  const int nb = pf_net_blocks(p);
  const int hp = ((p->net[which].width + 3) / 4) * 4;
  const size_t lds = ((pf_n32_bytes(L) + 127) & ~127) + (size_t)(BW_THREADS / 64) * WAVE_SCRATCH;
  static_assert(WAVE_SCRATCH >= 4096 + 512, "write-out staging must fit the wave scratch");
  hipLaunchKernelGGL((k_net32_backward<L, IN, GEA>), dim3(nb), dim3(BW_THREADS), lds, s, *p, which, hp);
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

int PF_CAT(pf_launch_net32_forward_, PF_NR)(const pf_problem* p, int which, hipStream_t s) {
  PF_DISPATCH(launch_fwd)
}
int PF_CAT(pf_launch_net32_backward_, PF_NR)(const pf_problem* p, int which, hipStream_t s) {
  PF_DISPATCH(launch_bwd)
}
int PF_CAT(pf_launch_net32_backward_gea_, PF_NR)(const pf_problem* p, int which, hipStream_t s) {
  PF_DISPATCH(launch_bwd_gea)
}
