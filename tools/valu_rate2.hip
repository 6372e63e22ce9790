// valu_rate2.hip — issue cost of the individual instructions the MFMA32 kernels use beside tanh: conversions,
// mixed-precision fma, packed f32, permlane swap, LDS broadcast reads; 4 blocks of 256 threads per CU (4 waves/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* st, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = i * 0.001f;
  __syncthreads();
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
  unsigned u[8];
  for (int i = 0; i < 8; ++i) u[i] = threadIdx.x + i;
  const int h = (threadIdx.x >> 5) & 1;
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
        if (KIND == 1) asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "+v"(u[i]) : "v"(a[i]), "v"(u[(i + 1) & 7]));
        if (KIND == 2) { f2 x = {a[i], a[(i + 1) & 7]}; f2 y; asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(y) : "v"(x)); a[i] = y[0]; }
        if (KIND == 3) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 1) & 7]));
        if (KIND == 4) { float4 w = reinterpret_cast<const float4*>(lds)[(i * 2 + h + it) & 63]; a[i] += w.x + w.w; }
        if (KIND == 5) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(u[i]));
        if (KIND == 6) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
        if (KIND == 7) asm volatile("v_fmamk_f32 %0, %1, 0xc5800000, %2" : "=v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_sched_barrier(0);
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i] + (float)u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    st[2 * w] = t1 - t0;
    st[2 * w + 1] = r1 - r0;
  }
}

int main() {
  const char* names[8] = {"v_cvt_pkrtz_f16_f32", "v_fma_mixlo_f16", "v_pk_fma_f32", "s_nop 1 + v_permlane32_swap", "ds_read_b128 (2-address broadcast) + 2 v_add",
                          "v_cvt_f32_f16", "v_fmac_f32", "v_fmamk_f32"};
  float* out; unsigned long long* st;
  (void)hipMalloc(&out, 256 * 8 * 256 * 4); (void)hipMalloc(&st, 256 * 8 * 4 * 16);
  for (int kind = 0; kind < 8; ++kind)
    for (int bpc : {1, 4}) {
      const int nb = 256 * bpc, iters = 5000;
      auto launch = [&]() {
        switch (kind) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, out, st, iters, 0.5f); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, out, st, iters, 0.5f); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(nb), dim3(256), 0, 0, out, st, iters, 0.5f); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(nb), dim3(256), 0, 0, out, st, iters, 0.5f); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(nb), dim3(256), 0, 0, out, st, iters, 0.5f); break;
          case 5: hipLaunchKernelGGL(k<5>, dim3(nb), dim3(256), 0, 0, out, st, iters, 0.5f); break;
          case 6: hipLaunchKernelGGL(k<6>, dim3(nb), dim3(256), 0, 0, out, st, iters, 0.5f); break;
          case 7: hipLaunchKernelGGL(k<7>, dim3(nb), dim3(256), 0, 0, out, st, iters, 0.5f); break;
        }
      };
      launch(); (void)hipDeviceSynchronize();
      launch(); (void)hipDeviceSynchronize();
      std::vector<unsigned long long> h(nb * 4 * 2);
      (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
      std::vector<double> cyc;
      for (int w = 0; w < nb * 4; ++w) cyc.push_back((double)h[2 * w]);
      std::sort(cyc.begin(), cyc.end());
      printf("%-46s %d wave(s)/SIMD: %6.2f cycles per instruction per SIMD\n", names[kind], bpc, cyc[cyc.size() / 2] / (32.0 * iters) / bpc);
    }
  return 0;
}
