// valu_rate.hip — what one SIMD sustains on the vector ALU, and the clock the chip holds meanwhile.
// Kernels of independent instruction chains (8 per wave) of one kind; s_memtime (shader cycles) and s_memrealtime
// (100 MHz) stamped around the loop.  Blocks of 256 threads; B blocks per CU -> B waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* st, int iters, float seed) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
  const float c = 0.999f, d = 1e-3f;
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) a[i] = fmaf(a[i], c, d);                                  // v_fma / v_fmac
        if (KIND == 1) a[i] = __builtin_amdgcn_exp2f(a[i] * 0.5f);               // v_mul + v_exp
        if (KIND == 2) a[i] = __builtin_amdgcn_rcpf(a[i] + 1.0f);                // v_add + v_rcp
        if (KIND == 3) {                                                          // the tanh of the kernels: 5 ops
          const float e = __builtin_amdgcn_exp2f(a[i] * 2.885f);
          const float q = __builtin_amdgcn_rcpf(e + 1.0f);
          a[i] = fmaf(-2.0f, q, 1.0f);
        }
        if (KIND == 4) a[i] = a[i] * c;                                          // v_mul
        if (KIND == 5) {                                                          // split: pkrtz, 2 cvt back, 2 sub, cvt_pk
          typedef _Float16 h2 __attribute__((ext_vector_type(2)));
          const h2 H = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(a[i], a[(i + 1) & 7]));
          const h2 L = h2{(_Float16)(a[i] - (float)H[0]), (_Float16)(a[(i + 1) & 7] - (float)H[1])};
          a[i] = (float)L[0] + (float)H[1] + (float)L[1];
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_sched_barrier(0);
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    st[2 * w] = t1 - t0;
    st[2 * w + 1] = r1 - r0;
  }
}

int main() {
  const char* names[6] = {"v_fma_f32", "v_mul+v_exp", "v_add+v_rcp", "tanh (mul,exp,add,rcp,fma)", "v_mul_f32", "split pair (8 ops)"};
  const int per_it[6] = {32, 64, 64, 160, 32, 32 * 10};
  float* out; unsigned long long* st;
  hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&st, 256 * 8 * 4 * 16);
  for (int kind = 0; kind < 6; ++kind)
    for (int bpc : {1, 2, 4}) {
      const int nb = 256 * bpc, iters = 20000;
      auto launch = [&]() {
        switch (kind) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, out, st, iters, 0.5f); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, out, st, iters, 0.5f); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(nb), dim3(256), 0, 0, out, st, iters, 0.5f); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(nb), dim3(256), 0, 0, out, st, iters, 0.5f); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(nb), dim3(256), 0, 0, out, st, iters, 0.5f); break;
          case 5: hipLaunchKernelGGL(k<5>, dim3(nb), dim3(256), 0, 0, out, st, iters, 0.5f); break;
        }
      };
      launch(); hipDeviceSynchronize();
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(nb * 4 * 2);
      hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
      std::vector<double> cyc, clk;
      for (int w = 0; w < nb * 4; ++w) { cyc.push_back((double)h[2 * w]); clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1); }
      std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
      const double c_med = cyc[cyc.size() / 2], f_med = clk[clk.size() / 2];
      // per SIMD: bpc waves, each `per_it*iters` instructions in c_med cycles
      printf("%-28s %d wave(s)/SIMD: %6.2f cycles per wave-instruction per SIMD (wave view %6.2f), clock %.2f GHz, %.3f ms\n",
             names[kind], bpc, c_med / ((double)per_it[kind] * iters) / bpc, c_med / ((double)per_it[kind] * iters), f_med, ms);
    }
  return 0;
}
