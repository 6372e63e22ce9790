// mfma_rate.hip — issue-rate microbenchmarks on gfx950: f32 MFMA shapes vs VALU, alone and mixed.
// Prints cycles per instruction per SIMD (s_memtime ticks = shader cycles) for W waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ITERS 2000

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc) {
  const int lane = threadIdx.x;
  float a = 1.0f + lane * 1e-3f, b = 0.5f + lane * 1e-4f;
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  f32x16 d0 = {0}, d1 = {0};
  float v0 = a, v1 = b, v2 = a + b, v3 = a - b, v4 = a * b, v5 = a + 1, v6 = b + 1, v7 = a + 2;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < ITERS; ++i) {
    if (MODE == 0) {  // 8 independent 4x4x1 broadcast MFMAs
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 4, 1, 0);
      c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 4, 2, 0);
      c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 4, 3, 0);
      c4 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c4, 4, 4, 0);
      c5 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c5, 4, 5, 0);
      c6 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c6, 4, 6, 0);
      c7 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c7, 4, 7, 0);
    } else if (MODE == 1) {  // 8 independent 4x4x1 plain
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c4, 0, 0, 0);
      c5 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c5, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c6, 0, 0, 0);
      c7 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c7, 0, 0, 0);
    } else if (MODE == 2) {  // 8 x 16x16x4 (4 independent accumulators)
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    } else if (MODE == 3) {  // 8 x 32x32x2
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
    } else if (MODE == 4) {  // 8 independent v_fma
      v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
      v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
    } else if (MODE == 5) {  // 4 x (4x4x1 + 2 v_fma) mixed
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 0, 0); v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b);
      c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 4, 1, 0); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
      c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 4, 2, 0); v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b);
      c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 4, 3, 0); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
    } else if (MODE == 6) {  // 2 x (32x32x2 + 12 v_fma) mixed
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
      v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
      v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
      v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
      d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
      v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
      v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
      v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
    } else if (MODE == 7) {  // 4 x (16x16x4 + 6 v_fma) mixed
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b); v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
      v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b); v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
      v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b); v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b);
      c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
      v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b); v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
    } else if (MODE == 9) {   // 8 x 4x4x1 on ONE accumulator (dependent chain)
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 0, 0); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 1, 0);
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 2, 0); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 3, 0);
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 4, 0); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 5, 0);
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 6, 0); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 7, 0);
    } else if (MODE == 10) {  // 8 x 4x4x1 on TWO accumulators
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 0, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 4, 1, 0);
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 2, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 4, 3, 0);
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 4, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 4, 5, 0);
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 6, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 4, 7, 0);
    } else if (MODE == 11) {  // B operand produced by a VALU op right before each MFMA (4 accs)
      v0 = fmaf(v0, a, b); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, v0, c0, 4, 0, 0);
      v1 = fmaf(v1, a, b); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, v1, c1, 4, 1, 0);
      v2 = fmaf(v2, a, b); c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, v2, c2, 4, 2, 0);
      v3 = fmaf(v3, a, b); c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, v3, c3, 4, 3, 0);
    } else if (MODE == 12) {  // 8 MFMAs then 8 VALU consuming their results (grouped phases)
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, v0, c0, 4, 0, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, v1, c1, 4, 1, 0);
      c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, v2, c2, 4, 2, 0); c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, v3, c3, 4, 3, 0);
      c4 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, v4, c4, 4, 4, 0); c5 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, v5, c5, 4, 5, 0);
      c6 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, v6, c6, 4, 6, 0); c7 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, v7, c7, 4, 7, 0);
      v0 = fmaf(c0[0], a, b); v1 = fmaf(c1[0], a, b); v2 = fmaf(c2[0], a, b); v3 = fmaf(c3[0], a, b);
      v4 = fmaf(c4[0], a, b); v5 = fmaf(c5[0], a, b); v6 = fmaf(c6[0], a, b); v7 = fmaf(c7[0], a, b);
    } else if (MODE == 8) {  // transcendental pair exp+rcp x4
      v0 = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v0)); v1 = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v1));
      v2 = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v2)); v3 = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v3));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  for (int r = 0; r < 4; ++r) s += c0[r] + c1[r] + c2[r] + c3[r] + c4[r] + c5[r] + c6[r] + c7[r];
  for (int r = 0; r < 16; ++r) s += d0[r] + d1[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * sizeof(float));
  hipMallocManaged(&cyc, 256 * sizeof(unsigned long long));
  for (int waves : {4, 16}) {  // waves per block = per CU (1 block per CU): 1, 2, 4 per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(waves * 64), 0, 0, out, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(waves * 64), 0, 0, out, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double c = (double)cyc[0];
    int wps = waves / 4;
    printf("%-28s waves/SIMD %d: %.2f cycles per instr per SIMD (wave view %.2f), kernel %.3f ms\n", name, wps,
           c / ((double)ITERS * per_iter * wps), c / ((double)ITERS * per_iter), ms);
  }
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("mfma 4x4x1 cbsz4", 8);
  run<1>("mfma 4x4x1 plain", 8);
  run<2>("mfma 16x16x4", 8);
  run<3>("mfma 32x32x2", 8);
  run<4>("v_fma_f32", 8);
  run<5>("4x4x1 + 2 v_fma (x4)", 12);
  run<6>("32x32x2 + 12 v_fma (x2)", 26);
  run<7>("16x16x4 + 6 v_fma (x4)", 28);
  run<8>("exp2+rcp (x4 pairs)", 8);
  run<9>("4x4x1 chain on 1 acc", 8);
  run<10>("4x4x1 chain on 2 acc", 8);
  run<11>("(v_fma -> 4x4x1 B) x4", 8);
  run<12>("8 mfma then 8 dependent fma", 16);
  return 0;
}
