// mfma_probe.hip — empirically map operand/result lanes of the f32 MFMA forms on gfx950.
// For every source lane s: A = one-hot(lane == s), B = lane + 1  =>  each non-zero D[lane][reg]
// names its A source lane (s) and its B source lane (value - 1).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID>
__global__ void probe_4x4x1(int* a_src, int* b_src) {
  const int lane = threadIdx.x;
  for (int s = 0; s < 64; ++s) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    const float a = lane == s ? 1.f : 0.f, b = (float)(lane + 1);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, CBSZ, ABID, 0);
    for (int r = 0; r < 4; ++r)
      if (c[r] != 0.f) { a_src[lane * 4 + r] = s; b_src[lane * 4 + r] = (int)c[r] - 1; }
  }
}

__global__ void probe_16x16x4(int* a_src, int* b_src) {
  const int lane = threadIdx.x;
  for (int s = 0; s < 64; ++s) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    const float a = lane == s ? 1.f : 0.f, b = (float)(lane + 1);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    // K=4: each D sums 4 products, only one has a != 0 per s
    for (int r = 0; r < 4; ++r)
      if (c[r] != 0.f) { a_src[(lane * 4 + r) * 4 + (s >> 4)] = s; b_src[(lane * 4 + r) * 4 + (s >> 4)] = (int)c[r] - 1; }
  }
}

static void dump(const char* name, int* a, int* b, int per) {
  printf("== %s  (lane reg : A-lane B-lane ...)\n", name);
  for (int lane = 0; lane < 64; ++lane) {
    if (!(lane < 8 || lane == 15 || lane == 16 || lane == 17 || lane == 32 || lane == 48 || lane == 63)) continue;
    printf("lane %2d:", lane);
    for (int r = 0; r < 4; ++r) {
      printf("  r%d", r);
      for (int k = 0; k < per; ++k) printf(" (%d,%d)", a[(lane * 4 + r) * per + k], b[(lane * 4 + r) * per + k]);
    }
    printf("\n");
  }
}

int main() {
  int *a, *b;
  hipMallocManaged(&a, 64 * 16 * sizeof(int));
  hipMallocManaged(&b, 64 * 16 * sizeof(int));
  auto reset = [&]() { for (int i = 0; i < 64 * 16; ++i) a[i] = b[i] = -1; };
  reset(); probe_4x4x1<0, 0><<<1, 64>>>(a, b); hipDeviceSynchronize(); dump("4x4x1 cbsz=0", a, b, 1);
  reset(); probe_4x4x1<4, 0><<<1, 64>>>(a, b); hipDeviceSynchronize(); dump("4x4x1 cbsz=4 abid=0", a, b, 1);
  reset(); probe_4x4x1<4, 5><<<1, 64>>>(a, b); hipDeviceSynchronize(); dump("4x4x1 cbsz=4 abid=5", a, b, 1);
  reset(); probe_4x4x1<2, 1><<<1, 64>>>(a, b); hipDeviceSynchronize(); dump("4x4x1 cbsz=2 abid=1", a, b, 1);
  reset(); probe_16x16x4<<<1, 64>>>(a, b); hipDeviceSynchronize(); dump("16x16x4", a, b, 4);
  return 0;
}
