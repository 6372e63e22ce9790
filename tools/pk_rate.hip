// pk_rate.hip — v_pk_fma_f32 vs v_fma_f32 issue rate on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 4000
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out) {
  const int lane = threadIdx.x;
  f2 a = {1.0f + lane * 1e-3f, 0.5f}, b = {0.25f, 0.125f};
  f2 v0 = a, v1 = b, v2 = a + b, v3 = a - b, v4 = a * b, v5 = a + 1.f, v6 = b + 1.f, v7 = a + 2.f;
  float s0 = a.x, s1 = a.y, s2 = b.x, s3 = b.y, s4 = 1.f, s5 = 2.f, s6 = 3.f, s7 = 4.f;
  for (int i = 0; i < ITERS; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v0 = __builtin_elementwise_fma(v0, a, b); v1 = __builtin_elementwise_fma(v1, a, b);
        v2 = __builtin_elementwise_fma(v2, a, b); v3 = __builtin_elementwise_fma(v3, a, b);
        v4 = __builtin_elementwise_fma(v4, a, b); v5 = __builtin_elementwise_fma(v5, a, b);
        v6 = __builtin_elementwise_fma(v6, a, b); v7 = __builtin_elementwise_fma(v7, a, b);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0 = fmaf(s0, a.x, b.x); s1 = fmaf(s1, a.x, b.x); s2 = fmaf(s2, a.x, b.x); s3 = fmaf(s3, a.x, b.x);
        s4 = fmaf(s4, a.x, b.x); s5 = fmaf(s5, a.x, b.x); s6 = fmaf(s6, a.x, b.x); s7 = fmaf(s7, a.x, b.x);
      }
    }
  }
  f2 t = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = t.x + t.y + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
}
template <int MODE> void run(const char* name) {
  float* out; hipMalloc(&out, 1024 * 256 * sizeof(float));
  for (int waves : {4, 16}) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(waves * 64), 0, 0, out); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(waves * 64), 0, 0, out); hipEventRecord(e1);
    hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-16s waves/SIMD %d: %.3f ms  -> %.2f ns per instruction per SIMD\n", name, waves / 4, ms,
           ms * 1e6 / (ITERS * 32.0 * (waves / 4)));
  }
  hipFree(out);
}
int main() { run<0>("v_pk_fma_f32"); run<1>("v_fma_f32"); return 0; }
