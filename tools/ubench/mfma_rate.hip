// Micro-benchmark: issue rate of v_mfma_f32_16x16x32_bf16 / 32x32x16 on gfx950, 1 or 2 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void k(float *out, unsigned long long *cyc, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (threadIdx.x + j)); b[j] = (__bf16)(0.02f * (threadIdx.x * 3 + j)); }
  f32x4 acc[16];
  f32x16 big[4];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) big[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big[i], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  for (int i = 0; i < 4; ++i) s += big[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  float *out; unsigned long long *cyc, h;
  hipMalloc(&out, 256 * 1024 * 4 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int mode = 0; mode < 2; ++mode)
    for (int threads = 256; threads <= 512; threads += 256) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        else hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      const double n_mfma = (double)iters * (mode == 0 ? 16 : 4);
      const double flop = n_mfma * (mode == 0 ? 16384.0 : 32768.0) * (threads / 64) * 256;
      printf("%s waves/SIMD=%d: %.1f cycles per MFMA per wave, %.1f TFLOP/s, %.3f ms, clock %.2f GHz\n",
             mode == 0 ? "16x16x32" : "32x32x16", threads / 256, (double)h / n_mfma, flop / ms / 1e9, ms,
             (double)h / (ms * 1e6));
    }
  return 0;
}
