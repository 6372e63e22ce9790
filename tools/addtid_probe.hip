// addtid_probe.hip — where does ds_write_addtid_b32 put a lane's dword?  (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int OFF>
__device__ __forceinline__ void put(float v, unsigned m0v) {
  asm volatile("s_mov_b32 m0, %2\n\tds_write_addtid_b32 %0 offset:%1" ::"v"(v), "n"(OFF), "s"(m0v) : "memory", "m0");
}
__global__ void k(float* out, int use_hi) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = -1.f;
  __syncthreads();
  float* cols = lds + w * 1024 + 32;
  const unsigned base = (unsigned)(size_t)cols;
  if (threadIdx.x == 0) out[4096] = (float)base;
  if (threadIdx.x == 64) out[4097] = (float)base;
  const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(base - 128u * (unsigned)use_hi));
  if ((lane >> 5) == use_hi) {
    put<0>(1000.f * w + lane, m0v);
    put<36 * 4>(2000.f * w + lane + 0.5f, m0v);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = lds[i];
}
int main() {
  float* d; hipMalloc(&d, 5000 * sizeof(float));
  float h[5000];
  for (int hi = 0; hi < 2; ++hi) {
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 2048 * sizeof(float), 0, d, hi);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("use_hi=%d base(w0)=%g base(w1)=%g\n", hi, h[4096], h[4097]);
    for (int i = 0; i < 2048; ++i) if (h[i] != -1.f) printf("  lds[%d]=%g", i, h[i]);
    printf("\n");
  }
  return 0;
}
