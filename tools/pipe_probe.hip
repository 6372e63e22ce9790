// pipe_probe.hip — what one SIMD of gfx950 issues beside a v_mfma_f32_32x32x16_f16 (replaces round 2's pipe_overlap.hip,
// whose run-time `if` inside the loop put 32 accumulator moves into every "pure" stream).
//
// Every stream is COMPILE-TIME (template parameters), every instruction is inline asm on VGPR operands (build with
// -mllvm -amdgpu-mfma-vgpr-form=1 like the product), the loop body is unrolled U times and timed per wave with
// s_memtime (shader cycles; s_memrealtime beside it gives the clock).  Check the ISA before trusting a number:
//     hipcc -S --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/pipe_probe.hip -o - | less
// (the loop bodies must hold nothing but the instructions named here plus the loop counter: no v_accvgpr_*, no v_mov).
//
// Rows (cycles per loop "slot" = one MFMA gap, or per filler where there is no MFMA), for 1, 2 and 4 waves per SIMD:
//   M            MFMA only, 4 independent accumulators           (guide: 32)
//   F<k>         k independent v_fma_f32 only                    (guide: 4 per instruction for one wave, 2 at >= 2 waves)
//   E<k>         k independent v_exp_f32 only                    (guide: 8 / ?)
//   M+F<k>       one MFMA then k v_fma_f32 in its gap, same wave (guide: max(32, 8 + 4k))
//   M+E1+F<k>    one MFMA, one v_exp_f32, k v_fma_f32
//   M+E<k>       one MFMA, k v_exp_f32
//   M|F          2 (4) waves per SIMD: half of them MFMA only, the other half v_fma_f32 only (roles by wave >= half)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(x, y, z) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))

template <int N, class F>
__device__ __forceinline__ void rep(F&& f) {
  if constexpr (N > 0) { f(); rep<N - 1>(f); }
}

template <int NM, int NE, int NF, int ROLE /* 0: all waves alike; 1: waves >= half do fillers only, the rest MFMA only */>
__global__ __launch_bounds__(1024) void k_probe(unsigned long long* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.5f + 0.01f * j); b[j] = (_Float16)(0.25f + threadIdx.x * 1e-3f); }
  float r[8];
  for (int i = 0; i < 8; ++i) r[i] = 0.5f + threadIdx.x * 1e-4f + i;
  const float y = 0.999f, z = 1e-3f;
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const bool filler_wave = ROLE == 1 && wave >= nw / 2;
  const bool mfma_wave = ROLE == 1 && wave < nw / 2;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
  if (ROLE == 0) {
    for (int it = 0; it < iters; ++it) {
      // 4 slots per trip, accumulators 0..3 in turn; fillers walk the 8 registers
#define SLOT(A, R0)                                                                     \
      if constexpr (NM > 0) MFMA(acc[A], a, b);                                         \
      if constexpr (NE > 0) EXP(r[(R0) & 7]);                                           \
      if constexpr (NE > 1) EXP(r[(R0 + 1) & 7]);                                       \
      if constexpr (NE > 2) EXP(r[(R0 + 2) & 7]);                                       \
      if constexpr (NE > 3) EXP(r[(R0 + 3) & 7]);                                       \
      if constexpr (NF > 0) FMA(r[(R0 + 4) & 7], y, z);                                 \
      if constexpr (NF > 1) FMA(r[(R0 + 5) & 7], y, z);                                 \
      if constexpr (NF > 2) FMA(r[(R0 + 6) & 7], y, z);                                 \
      if constexpr (NF > 3) FMA(r[(R0 + 7) & 7], y, z);                                 \
      if constexpr (NF > 4) FMA(r[(R0 + 0) & 7], y, z);                                 \
      if constexpr (NF > 5) FMA(r[(R0 + 1) & 7], y, z);                                 \
      if constexpr (NF > 6) FMA(r[(R0 + 2) & 7], y, z);                                 \
      if constexpr (NF > 7) FMA(r[(R0 + 3) & 7], y, z);
      SLOT(0, 0) SLOT(1, 1) SLOT(2, 2) SLOT(3, 3)
    }
  } else if (mfma_wave) {
    for (int it = 0; it < iters; ++it) { MFMA(acc[0], a, b); MFMA(acc[1], a, b); MFMA(acc[2], a, b); MFMA(acc[3], a, b); }
  } else if (filler_wave) {
    for (int it = 0; it < iters; ++it) {
#define FSLOT(R0)                                                                       \
      if constexpr (NE > 0) EXP(r[(R0) & 7]);                                           \
      if constexpr (NE > 1) EXP(r[(R0 + 1) & 7]);                                       \
      if constexpr (NE > 2) EXP(r[(R0 + 2) & 7]);                                       \
      if constexpr (NE > 3) EXP(r[(R0 + 3) & 7]);                                       \
      if constexpr (NF > 0) FMA(r[(R0 + 4) & 7], y, z);                                 \
      if constexpr (NF > 1) FMA(r[(R0 + 5) & 7], y, z);                                 \
      if constexpr (NF > 2) FMA(r[(R0 + 6) & 7], y, z);                                 \
      if constexpr (NF > 3) FMA(r[(R0 + 7) & 7], y, z);                                 \
      if constexpr (NF > 4) FMA(r[(R0 + 0) & 7], y, z);                                 \
      if constexpr (NF > 5) FMA(r[(R0 + 1) & 7], y, z);                                 \
      if constexpr (NF > 6) FMA(r[(R0 + 2) & 7], y, z);                                 \
      if constexpr (NF > 7) FMA(r[(R0 + 3) & 7], y, z);
      FSLOT(0) FSLOT(1) FSLOT(2) FSLOT(3)
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  for (int i = 0; i < 8; ++i) s += r[i];
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* d = out + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 4;
    d[0] = t1 - t0;
    d[1] = w1 - w0;
    d[2] = (unsigned long long)__float_as_uint(s);
    d[3] = filler_wave ? 1 : 0;
  }
}

struct Res { double cyc_all, cyc_mfma, cyc_fill, ghz; };

template <int NM, int NE, int NF, int ROLE>
static Res run(unsigned long long* dout, int waves_per_simd) {
  const int iters = 4000, threads = 256 * waves_per_simd, blocks = 256;
  hipLaunchKernelGGL((k_probe<NM, NE, NF, ROLE>), dim3(blocks), dim3(threads), 0, 0, dout, 200);      // warm-up
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL((k_probe<NM, NE, NF, ROLE>), dim3(blocks), dim3(threads), 0, 0, dout, iters);
  (void)hipDeviceSynchronize();
  const int nw = blocks * threads / 64;
  std::vector<unsigned long long> h((size_t)nw * 4);
  (void)hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> all, m, f, clk;
  for (int w = 0; w < nw; ++w) {
    const double c = (double)h[w * 4] / (iters * 4.0);
    all.push_back(c);
    (h[w * 4 + 3] ? f : m).push_back(c);
    if (h[w * 4 + 1]) clk.push_back((double)h[w * 4] / (double)h[w * 4 + 1] * 0.1);    // s_memrealtime: 100 MHz
  }
  auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  return Res{med(all), med(m), med(f), med(clk)};
}

#define ROW(NAME, NM, NE, NF)                                                                            \
  {                                                                                                      \
    printf("%-12s", NAME);                                                                               \
    for (int w : {1, 2, 4}) {                                                                            \
      Res r = run<NM, NE, NF, 0>(dout, w);                                                               \
      printf("  %dw/SIMD %7.2f cyc/slot/wave (= %6.2f per SIMD-slot, %.2f GHz)", w, r.cyc_all, r.cyc_all / w, r.ghz); \
    }                                                                                                    \
    printf("\n");                                                                                        \
  }
#define ROWX(NAME, NE, NF)                                                                               \
  {                                                                                                      \
    printf("%-12s", NAME);                                                                               \
    for (int w : {2, 4}) {                                                                               \
      Res r = run<1, NE, NF, 1>(dout, w);                                                                \
      printf("  %dw/SIMD mfma-waves %7.2f cyc/MFMA, filler-waves %7.2f cyc/slot (%.2f GHz)", w, r.cyc_mfma, r.cyc_fill, r.ghz); \
    }                                                                                                    \
    printf("\n");                                                                                        \
  }

int main() {
  unsigned long long* dout;
  (void)hipMalloc(&dout, (size_t)256 * 16 * 4 * 8);
  printf("# slot = one MFMA gap (or one group of fillers where the stream has no MFMA); cycles from s_memtime, median over waves\n");
  ROW("M", 1, 0, 0)
  ROW("F1", 0, 0, 1) ROW("F4", 0, 0, 4) ROW("F8", 0, 0, 8)
  ROW("E1", 0, 1, 0) ROW("E4", 0, 4, 0)
  ROW("E1+F4", 0, 1, 4)
  ROW("M+F1", 1, 0, 1) ROW("M+F2", 1, 0, 2) ROW("M+F3", 1, 0, 3) ROW("M+F4", 1, 0, 4)
  ROW("M+F5", 1, 0, 5) ROW("M+F6", 1, 0, 6) ROW("M+F7", 1, 0, 7) ROW("M+F8", 1, 0, 8)
  ROW("M+E1", 1, 1, 0) ROW("M+E2", 1, 2, 0) ROW("M+E3", 1, 3, 0) ROW("M+E4", 1, 4, 0)
  ROW("M+E1+F1", 1, 1, 1) ROW("M+E1+F2", 1, 1, 2) ROW("M+E1+F3", 1, 1, 3) ROW("M+E1+F4", 1, 1, 4)
  ROW("M+E2+F2", 1, 2, 2) ROW("M+E2+F4", 1, 2, 4)
  printf("# roles split between the waves of a SIMD: waves < half issue MFMAs only, waves >= half fillers only\n");
  ROWX("M|F4", 0, 4) ROWX("M|F8", 0, 8) ROWX("M|E4", 4, 0) ROWX("M|E2+F4", 2, 4)
  return 0;
}
