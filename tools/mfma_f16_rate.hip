// mfma_f16_rate.hip — does the f16 matrix pipe overlap with f32 VALU on gfx950?  (it does not for f32 MFMA)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ITERS 2000
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out) {
  const int lane = threadIdx.x;
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
  f32x16 d0 = {0}, d1 = {0};
  float x = 1.0f + lane * 1e-3f, y = 0.5f;
  float v0 = x, v1 = y, v2 = x + y, v3 = x - y, v4 = x * y, v5 = x + 1, v6 = y + 1, v7 = x + 2;
  for (int i = 0; i < ITERS; ++i) {
    if (MODE == 0) {        // 4 MFMA f16 32x32x16
      d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d1, 0, 0, 0);
    } else if (MODE == 1) { // 48 v_fma
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        v0 = fmaf(v0, x, y); v1 = fmaf(v1, x, y); v2 = fmaf(v2, x, y); v3 = fmaf(v3, x, y);
        v4 = fmaf(v4, x, y); v5 = fmaf(v5, x, y); v6 = fmaf(v6, x, y); v7 = fmaf(v7, x, y);
      }
    } else {                // 4 x (MFMA f16 + 12 v_fma)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r & 1) d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d1, 0, 0, 0);
        else d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d0, 0, 0, 0);
        v0 = fmaf(v0, x, y); v1 = fmaf(v1, x, y); v2 = fmaf(v2, x, y); v3 = fmaf(v3, x, y);
        v4 = fmaf(v4, x, y); v5 = fmaf(v5, x, y); v6 = fmaf(v6, x, y); v7 = fmaf(v7, x, y);
        v0 = fmaf(v0, x, y); v1 = fmaf(v1, x, y); v2 = fmaf(v2, x, y); v3 = fmaf(v3, x, y);
      }
    }
  }
  float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  for (int r = 0; r < 16; ++r) s += d0[r] + d1[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name) {
  float* out; hipMalloc(&out, 1024 * 256 * sizeof(float));
  for (int waves : {4, 8, 16}) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(waves * 64), 0, 0, out); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(waves * 64), 0, 0, out); hipEventRecord(e1);
    hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves/SIMD %d: %.3f ms (per loop iteration per wave-slot %.1f ns)\n", name, waves / 4, ms,
           ms * 1e6 / ITERS / (waves / 4));
  }
  hipFree(out);
}
int main() {
  run<0>("4 x mfma_f32_32x32x16_f16");
  run<1>("48 x v_fma_f32");
  run<2>("4 x (mfma f16 + 12 v_fma)");
  return 0;
}
