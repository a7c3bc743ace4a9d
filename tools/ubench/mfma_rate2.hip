// MFMA issue rate with the real operand pattern of the conv kernels: 4x4 accumulator tile, 4 A and 4 B fragments
// (16 MFMAs reuse each fragment 4 times), fragments optionally refreshed by a cheap VALU op every step.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int REFRESH>
__global__ void k(float *out, unsigned long long *cyc, int iters) {
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) { a[i][j] = 0x3c003c00u + threadIdx.x * 7 + i * 13 + j; b[i][j] = 0x3a003b00u + threadIdx.x * 3 + i * 5 + j; }
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (REFRESH) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i][0] ^= it; b[i][1] ^= it; }   // operands change every step (like new ds_reads)
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[ni]), __builtin_bit_cast(bf16x8, b[mi]), acc[mi][ni], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  float *out; unsigned long long *cyc, h;
  hipMalloc(&out, 256 * 1024 * 4 * 4); hipMalloc(&cyc, 8);
  const int iters = 4000;
  for (int refresh = 0; refresh < 2; ++refresh)
    for (int threads = 256; threads <= 512; threads += 256) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (refresh) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        else hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      const double n_mfma = (double)iters * 16;
      const double flop = n_mfma * 16384.0 * (threads / 64) * 256;
      printf("4x4 tile refresh=%d waves/SIMD=%d: %.1f memtime ticks per MFMA (wave 0), %.1f TFLOP/s, %.3f ms -> %.1f ns per MFMA per SIMD\n",
             refresh, threads / 256, (double)h / n_mfma, flop / ms / 1e9, ms, ms * 1e6 / (n_mfma * (threads / 256)));
    }
  return 0;
}
