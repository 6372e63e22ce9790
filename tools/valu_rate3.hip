// valu_rate3.hip — does a LONG straight-line body of vector instructions over MANY registers issue as fast as a short
// loop?  N independent fma chains (N registers), the body unrolled to N*REPS instructions; 4 blocks of 256 per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

template <int N, int REPS>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* st, int iters, float seed) {
  float a[N];
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < REPS; ++rep) {
#pragma unroll
      for (int i = 0; i < N; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) % N]), "v"(a[(i + 2) % N]));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int N, int REPS>
void run(const char* name, float* out, unsigned long long* st) {
  for (int bpc : {1, 2, 4}) {
    const int nb = 256 * bpc, iters = 200000 / (N * REPS) + 1;
    hipLaunchKernelGGL((k<N, REPS>), dim3(nb), dim3(256), 0, 0, out, st, iters, 0.5f);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<N, REPS>), dim3(nb), dim3(256), 0, 0, out, st, iters, 0.5f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb * 4);
    (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-40s %d wave(s)/SIMD: %5.2f cycles per instruction per SIMD\n", name, bpc, (double)h[h.size() / 2] / ((double)N * REPS * iters) / bpc);
  }
}

int main() {
  float* out; unsigned long long* st;
  (void)hipMalloc(&out, 256 * 8 * 256 * 4); (void)hipMalloc(&st, 256 * 8 * 4 * 8);
  run<8, 4>("8 regs, body 32 instr", out, st);
  run<32, 4>("32 regs, body 128 instr", out, st);
  run<64, 8>("64 regs, body 512 instr", out, st);
  run<96, 8>("96 regs, body 768 instr", out, st);
  run<96, 32>("96 regs, body 3072 instr (24 KB)", out, st);
  return 0;
}
