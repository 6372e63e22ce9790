// f16split_rate.hip — one hidden layer (20 -> 20, tanh) of the property MLP on the f16 matrix pipe with 2-way split
// operands (hi + lo*2^-11, three products), elements on the N side of v_mfma_f32_32x32x16_f16: cycles per 64
// elements and error against the exact f32 layer.  Question for DESIGN.md §7: is the split-f16 route worth building?
//
// Layout: a 32x32 result tile D[m][n] = sum_k A[m][k] B[k][n]; lane l holds column n = l%32 and rows
// m = 8*(r/4) + 4*(l/32) + r%4 in registers r = 0..15; A: lane l holds row m = l%32, k = 8*(l/32)+j, j = 0..7;
// B: lane l holds column n = l%32, k = 8*(l/32)+j.  Hidden unit u lives in row m(u); rows are chosen so that each
// half-wave owns 10 of the 20 units (registers r = 0..9 of every lane), and k-slots are filled from the lane's OWN
// registers, so activations never move between lanes from one layer to the next.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define H 20
#define ITERS 400

// row of the tile that register r of half-wave hw holds
__host__ __device__ inline int row_of(int r, int hw) { return 8 * (r / 4) + 4 * hw + (r % 4); }
// hidden unit <-> (half-wave, register): unit u = 10*hw + r, r = 0..9
__host__ __device__ inline int unit_of(int r, int hw) { return r < 10 ? 10 * hw + r : -1; }

__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// MODE 0: split-f16 layer; MODE 1: only the conversions + tanh (no MFMA); MODE 2: only the MFMAs
template <int MODE>
__global__ __launch_bounds__(256) void k(const _Float16* __restrict__ wA, const float* __restrict__ a0, float* __restrict__ out,
                                         unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63, hw = lane >> 5;
  // A operands of the 5 MFMAs: [mfma][lane][8] f16
  h8 A[5];
  for (int q = 0; q < 5; ++q) A[q] = *reinterpret_cast<const h8*>(wA + ((size_t)q * 64 + lane) * 8);
  float a[10];
  for (int r = 0; r < 10; ++r) a[r] = a0[(size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 10 + r];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    // split own activations: hi = f16(a), lo' = f16((a - hi) * 2^11)
    _Float16 ah[10], al[10];
    if (MODE != 2) {
#pragma unroll
      for (int r = 0; r < 10; ++r) {
        ah[r] = (_Float16)a[r];
        al[r] = (_Float16)((a[r] - (float)ah[r]) * 2048.0f);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 10; ++r) { ah[r] = (_Float16)0.25f; al[r] = (_Float16)0.125f; }
    }
    const _Float16 z0 = (_Float16)0.0f;
    const h8 Bh1 = {ah[0], ah[1], ah[2], ah[3], ah[4], ah[5], ah[6], ah[7]};
    const h8 Bh2 = {ah[8], ah[9], z0, z0, z0, z0, z0, z0};
    const h8 Bl1 = {al[0], al[1], al[2], al[3], al[4], al[5], al[6], al[7]};
    const h8 Bl2 = {al[8], al[9], ah[0], ah[1], ah[2], ah[3], ah[4], ah[5]};
    const h8 Bl3 = {ah[6], ah[7], ah[8], ah[9], z0, z0, z0, z0};
    f16v dhi = {0}, dlo = {0};
    if (MODE != 1) {
      dhi = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], Bh1, dhi, 0, 0, 0);
      dhi = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1], Bh2, dhi, 0, 0, 0);
      dlo = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[2], Bl1, dlo, 0, 0, 0);
      dlo = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[3], Bl2, dlo, 0, 0, 0);
      dlo = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[4], Bl3, dlo, 0, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 10; ++r) { dhi[r] = (float)ah[r]; dlo[r] = (float)al[r]; }
    }
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const float z = fmaf(dlo[r], 1.0f / 2048.0f, dhi[r]);
      a[r] = (it + 1 == iters) ? z : tanh_fast(z);      // last pass leaves the pre-activation for the accuracy check
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < 10; ++r) out[(size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 10 + r] = a[r];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  // weights W[u_out][u_in], |W| ~ U(-0.22, 0.22) like torch's default init for fan-in 20
  std::vector<float> W(H * H);
  srand(3);
  for (auto& w : W) w = (rand() / (float)RAND_MAX * 2.f - 1.f) * 0.2236f;
  // A operands: lane (m = lane%32, hw = lane/32), slot j: weight of OUTPUT unit at row m against the INPUT unit that
  // half-wave hw puts in slot j of that MFMA (see the B vectors in the kernel)
  auto out_unit_of_row = [](int m) { for (int hw = 0; hw < 2; ++hw) for (int r = 0; r < 10; ++r) if (row_of(r, hw) == m) return unit_of(r, hw); return -1; };
  std::vector<_Float16> wA(5 * 64 * 8);
  for (int q = 0; q < 5; ++q)
    for (int lane = 0; lane < 64; ++lane)
      for (int j = 0; j < 8; ++j) {
        const int m = lane % 32, hw = lane / 32, uo = out_unit_of_row(m);
        int r_in = -1; bool want_lo_w = false;    // which input register of half-wave hw sits in slot j, and hi or lo' weight
        if (q == 0) r_in = j;                                   // ah[0..7]  x Wh
        if (q == 1) r_in = j < 2 ? 8 + j : -1;                  // ah[8..9]  x Wh
        if (q == 2) r_in = j;                                   // al[0..7]  x Wh
        if (q == 3) { if (j < 2) r_in = 8 + j; else { r_in = j - 2; want_lo_w = true; } }   // al[8..9] x Wh | ah[0..5] x Wl'
        if (q == 4) { if (j < 4) { r_in = 6 + j; want_lo_w = true; } }                      // ah[6..9] x Wl'
        float v = 0.f;
        if (uo >= 0 && r_in >= 0) {
          const float w = W[uo * H + unit_of(r_in, hw)];
          const _Float16 whi = (_Float16)w;
          v = want_lo_w ? (float)(_Float16)((w - (float)whi) * 2048.0f) : (float)whi;
        }
        wA[((size_t)q * 64 + lane) * 8 + j] = (_Float16)v;
      }
  const int blocks = 256 * 8, threads = 256, nthr = blocks * threads;
  std::vector<float> a0((size_t)nthr * 10);
  for (auto& v : a0) v = tanhf((rand() / (float)RAND_MAX * 2.f - 1.f) * 2.0f);
  _Float16* dW; float *dA, *dO; unsigned long long* dC;
  hipMalloc(&dW, wA.size() * 2); hipMalloc(&dA, a0.size() * 4); hipMalloc(&dO, a0.size() * 4); hipMalloc(&dC, blocks * 8);
  hipMemcpy(dW, wA.data(), wA.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dA, a0.data(), a0.size() * 4, hipMemcpyHostToDevice);
  // accuracy: ONE layer (iters = 1 leaves the pre-activation z) against the exact float64 product of the f32 operands
  hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, dW, dA, dO, dC, 1);
  std::vector<float> z((size_t)nthr * 10);
  hipMemcpy(z.data(), dO, z.size() * 4, hipMemcpyDeviceToHost);
  double emax = 0, e32max = 0, zmax = 0;
  for (int t = 0; t < 4096; ++t) {                 // first 64 waves
    const int lane = t % 64, wave = t / 64, n = lane % 32, hw = lane / 32;
    for (int r = 0; r < 10; ++r) {
      const int uo = unit_of(r, hw);
      double ref = 0; float f32 = 0.f;
      for (int hw2 = 0; hw2 < 2; ++hw2)
        for (int r2 = 0; r2 < 10; ++r2) {
          const float ain = a0[((size_t)wave * 64 + hw2 * 32 + n) * 10 + r2];
          const float w = W[uo * H + unit_of(r2, hw2)];
          ref += (double)w * (double)ain;
          f32 = fmaf(w, ain, f32);
        }
      emax = fmax(emax, fabs((double)z[(size_t)t * 10 + r] - ref));
      e32max = fmax(e32max, fabs((double)f32 - ref));
      zmax = fmax(zmax, fabs(ref));
    }
  }
  printf("accuracy of one 20x20 layer, max |z| %.3f: split-f16 max abs err %.3e, plain f32 fma chain %.3e\n", zmax, emax, e32max);
  // rate
  const char* names[3] = {"split-f16 layer (convert + 5 MFMA + combine + tanh)", "VALU part only (convert + combine + tanh)", "MFMA part only"};
  for (int mode = 0; mode < 3; ++mode) {
    for (int bpc : {1, 2}) {                        // blocks per CU -> 1 or 2 waves per SIMD
      const int nb = 256 * bpc;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(threads), 0, 0, dW, dA, dO, dC, ITERS);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(threads), 0, 0, dW, dA, dO, dC, ITERS);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nb), dim3(threads), 0, 0, dW, dA, dO, dC, ITERS);
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c; hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
      printf("%-52s %d wave(s)/SIMD: %7.1f cycles per wave-pass of 32 elements x 20 units (wave view), %.3f ms\n", names[mode], bpc,
             (double)c / ITERS, ms);
    }
  }
  printf("for comparison, the f32 path per 64 elements: 105 v_mfma_f32_4x4x1 (8.3 cycles) + 20 tanh (22 cycles) = %d cycles\n", (int)(105 * 8.3 + 20 * 22));
  return 0;
}
