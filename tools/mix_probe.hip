#include <hip/hip_runtime.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float a0, float a1, unsigned& hi, unsigned& lo) {
  hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a0, a1));
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a0), "v"(hi));
  asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(a1), "v"(hi));
  lo = l;
}
__global__ void k(unsigned* out, const float* in) {
  int l = threadIdx.x;
  unsigned hi, lo;
  split_pair(in[l], in[l + 64], hi, lo);
  out[l] = hi; out[l + 64] = lo;
}
int main() {
  float h[128]; unsigned o[128];
  for (int i = 0; i < 128; ++i) h[i] = (i - 60) * 0.0123456789f * (1 + i * 3.7f);
  float* d; unsigned* od;
  hipMalloc(&d, 512); hipMalloc(&od, 512);
  hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, od, d);
  hipMemcpy(o, od, 512, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int i = 0; i < 64; ++i) {
    h2 H = __builtin_bit_cast(h2, o[i]), L = __builtin_bit_cast(h2, o[i + 64]);
    double e0 = fabs(((double)(float)H[0] + (double)(float)L[0]) - h[i]) / fmax(fabs(h[i]), 1e-30);
    double e1 = fabs(((double)(float)H[1] + (double)(float)L[1]) - h[i + 64]) / fmax(fabs(h[i + 64]), 1e-30);
    worst = fmax(worst, fmax(e0, e1));
  }
  printf("split_pair worst relative error %.3e (want <= 2^-21 = 4.8e-7)\n", worst);
  return 0;
}
