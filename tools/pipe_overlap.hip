// pipe_overlap.hip — do the transcendental unit, the plain vector ALU and the matrix cores of one SIMD work for DIFFERENT
// waves at the same time?  Blocks of 256 threads (one wave per SIMD each); kind(block) selects what the block's waves
// issue: 0 v_exp_f32 chain, 1 v_fma_f32 chain, 2 v_mfma_f32_32x32x16_f16 chain.  Time of mixes against the parts.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(float* out, int iters, int kindA, int kindB, int split) {
  const int kind = (int)(blockIdx.x % 4) < split ? kindA : kindB;   // which blocks of a CU's four do what
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = 0.5f + threadIdx.x * 1e-4f + i;
  f16v acc = {0};
  h8 x = {1, 1, 1, 1, 1, 1, 1, 1};
  for (int it = 0; it < iters; ++it) {
    if (kind == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
    } else if (kind == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, acc, 0, 0, 0);
    }
  }
  float s = acc[0];
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float run(float* out, int kindA, int kindB, int split, int blocks_per_cu) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL(k, dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, iters, kindA, kindB, split);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k, dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, iters, kindA, kindB, split);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 8 * 256 * 4);
  const char* nm[3] = {"exp", "fma", "mfma"};
  for (int k0 = 0; k0 < 3; ++k0) printf("%-5s alone, 2 waves/SIMD: %7.3f ms   4 waves/SIMD: %7.3f ms\n", nm[k0], run(out, k0, k0, 4, 2), run(out, k0, k0, 4, 4));
  for (int a = 0; a < 3; ++a)
    for (int b = a + 1; b < 3; ++b)
      printf("%-5s + %-5s, 2 + 2 waves/SIMD: %7.3f ms\n", nm[a], nm[b], run(out, a, b, 2, 4));
  return 0;
}
