// Micro-benchmark: cost of global_load_lds_dwordx4 issue and of ds_read_b128, per wave, on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o glds_cost glds_cost.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// MODE 0: N LDS-DMA loads then vmcnt(0).  MODE 1: N ds_read_b128 then lgkmcnt(0).  MODE 2: N global_load_dwordx4 to VGPR.
template <int MODE, int N>
__global__ void k(const char *src, unsigned long long *cyc, float *sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const char *mine = src + ((size_t)blockIdx.x * 65536) + tid * 16;
  u32x4 acc = {0, 0, 0, 0};
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < N; ++i)
        __builtin_amdgcn_global_load_lds((gbl_void *)(mine + i * 8192), (lds_void *)(smem + wave * 1024 + i * 8192), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        u32x4 v = *reinterpret_cast<const u32x4 *>(smem + ((lane * 16 + wave * 1024 + i * 8192) & 65535));
        acc += v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
      u32x4 v[N];
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = *reinterpret_cast<const u32x4 *>(mine + i * 8192);
#pragma unroll
      for (int i = 0; i < N; ++i) acc += v[i];
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0 && wave == 0) cyc[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * blockDim.x + tid] = (float)(acc[0] + acc[1] + acc[2] + acc[3]) + ((float *)smem)[tid];
}

template <int MODE, int N>
void run(const char *name, int threads, const char *src, unsigned long long *cyc, float *sink) {
  const int iters = 200, blocks = 256;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, N>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, N>), dim3(blocks), dim3(threads), 65536, 0, src, cyc, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
  const double bytes = (double)blocks * (threads / 64) * iters * N * 1024.0;
  printf("%-28s waves/CU=%d N=%d: %.0f cycles per instruction per wave (batch of N then wait), %.2f TB/s aggregate, %.3f ms\n",
         name, threads / 64, N, avg / (iters * N), bytes / ms / 1e9, ms);
}

int main() {
  char *src; unsigned long long *cyc; float *sink;
  hipMalloc(&src, 256 * 65536 + 65536); hipMemset(src, 1, 256 * 65536 + 65536);
  hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 256 * 1024 * 4);
  run<0, 8>("global_load_lds_dwordx4", 256, src, cyc, sink);
  run<0, 8>("global_load_lds_dwordx4", 512, src, cyc, sink);
  run<0, 3>("global_load_lds_dwordx4", 512, src, cyc, sink);
  run<2, 8>("global_load_dwordx4 (vgpr)", 256, src, cyc, sink);
  run<2, 8>("global_load_dwordx4 (vgpr)", 512, src, cyc, sink);
  run<1, 8>("ds_read_b128", 256, src, cyc, sink);
  run<1, 8>("ds_read_b128", 512, src, cyc, sink);
  run<1, 16>("ds_read_b128", 512, src, cyc, sink);
  return 0;
}
