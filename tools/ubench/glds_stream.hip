// Micro-benchmark: what rate does global_load_lds_dwordx4 (LDS-DMA) sustain per CU when the bytes come from the XCD's L2,
// from the Infinity Cache / HBM, and how many instructions have to be in flight for it?  (glds_cost.hip re-reads 32 KiB per
// CU: that is the L1 rate.)  All 256 CUs stream at once, 4 loader waves per CU like the conv kernels; every instruction
// covers 8 rows x 128 B (one LDS-DMA piece of the strip kernel) with a row pitch of ROWB bytes.
// Build: hipcc --offload-arch=gfx950 -O3 -o glds_stream glds_stream.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// Each block streams `pieces` 4-KiB pieces (4 waves x 1 KiB) starting at its own offset inside `region` bytes (wrapping).
// DEPTH = LDS-DMA instructions a wave keeps in flight (counted vmcnt wait).
template <int DEPTH>
__global__ __launch_bounds__(256) void k(const char *src, size_t region, size_t xcd_stride, size_t blk_stride, int rowb,
                                          int pieces, unsigned long long *cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const char *base = src + (size_t)xcd * xcd_stride;
  // a piece = 32 rows of rowb bytes (the 4 waves take 8 rows each); lane: row = tid >> 3, 16-byte slot = tid & 7
  const size_t lane_off = (size_t)(tid >> 3) * rowb + (tid & 7) * 16;
  const size_t piece_b = (size_t)32 * rowb;
  size_t off = ((size_t)idx * blk_stride) % region;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < pieces; ++i) {
    __builtin_amdgcn_global_load_lds((gbl_void *)(base + off + lane_off), (lds_void *)(smem + wave * 1024 + (i & 15) * 4096), 16, 0, 0);
    off += piece_b;
    if (off + piece_b > region) off = 0;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

static int g_blocks = 256;
static int g_vgpr = 0;      // argv[2] = 1: the register path (kv) instead of LDS-DMA (k); 2: both at once (kmix)   // argv[1]: fewer blocks = only some CUs stream (8 per XCD with 64)

// The same stream through the ordinary vector-memory path: global_load_dwordx4 into registers, DEPTH loads in flight per wave,
// each followed by a ds_write_b128 (what a loader without LDS-DMA would do).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int DEPTH>
__global__ __launch_bounds__(256) void kv(const char *src, size_t region, size_t xcd_stride, size_t blk_stride, int rowb,
                                           int pieces, unsigned long long *cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const char *base = src + (size_t)xcd * xcd_stride;
  const size_t lane_off = (size_t)(tid >> 3) * rowb + (tid & 7) * 16;
  const size_t piece_b = (size_t)32 * rowb;
  size_t off = ((size_t)idx * blk_stride) % region;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  u32x4 v[DEPTH];
  for (int i = 0; i < pieces; i += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      v[d] = *reinterpret_cast<const u32x4 *>(base + off + lane_off);
      off += piece_b;
      if (off + piece_b > region) off = 0;
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) *reinterpret_cast<u32x4 *>(smem + tid * 16 + ((i + d) & 15) * 4096) = v[d];
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// Both paths at once: waves 0-1 stream by LDS-DMA, waves 2-3 through registers (each pair moves half of every 4-KiB piece).
template <int DEPTH>
__global__ __launch_bounds__(256) void kmix(const char *src, size_t region, size_t xcd_stride, size_t blk_stride, int rowb,
                                             int pieces, unsigned long long *cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const char *base = src + (size_t)xcd * xcd_stride;
  const size_t lane_off = (size_t)(tid >> 3) * rowb + (tid & 7) * 16;
  const size_t piece_b = (size_t)32 * rowb;
  size_t off = ((size_t)idx * blk_stride) % region;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 2) {
    for (int i = 0; i < pieces; ++i) {
      __builtin_amdgcn_global_load_lds((gbl_void *)(base + off + lane_off), (lds_void *)(smem + wave * 1024 + (i & 15) * 4096), 16, 0, 0);
      off += piece_b;
      if (off + piece_b > region) off = 0;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    u32x4 v[DEPTH];
    for (int i = 0; i < pieces; i += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        v[d] = *reinterpret_cast<const u32x4 *>(base + off + lane_off);
        off += piece_b;
        if (off + piece_b > region) off = 0;
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) *reinterpret_cast<u32x4 *>(smem + tid * 16 + ((i + d) & 15) * 4096) = v[d];
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int DEPTH>
void run(const char *what, const char *src, size_t region, size_t xcd_stride, size_t blk_stride, int rowb, int pieces,
         unsigned long long *cyc, int reps) {
  const int blocks = g_blocks;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute(reinterpret_cast<const void *>(kv<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute(reinterpret_cast<const void *>(kmix<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < reps; ++rep) {
    hipEventRecord(e0);
    if (g_vgpr == 2) hipLaunchKernelGGL((kmix<DEPTH>), dim3(blocks), dim3(256), 65536, 0, src, region, xcd_stride, blk_stride, rowb, pieces, cyc);
    else if (g_vgpr) hipLaunchKernelGGL((kv<DEPTH>), dim3(blocks), dim3(256), 65536, 0, src, region, xcd_stride, blk_stride, rowb, pieces, cyc);
    else hipLaunchKernelGGL((k<DEPTH>), dim3(blocks), dim3(256), 65536, 0, src, region, xcd_stride, blk_stride, rowb, pieces, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long h[256]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
  const double bytes_blk = (double)pieces * 4096.0;
  printf("%-34s depth %2d rowb %5d: %6.1f B/clk/CU (%5.0f cycles per 4-KiB piece), %6.2f TB/s aggregate (last of %d launches)\n",
         what, DEPTH, rowb, bytes_blk / avg, avg / pieces, bytes_blk * blocks / ms / 1e9, reps);
}

template <int DEPTH>
void suite(const char *src, unsigned long long *cyc) {
  const size_t MiB = 1 << 20;
  // L2: every XCD re-reads its own 2 MiB (32 blocks, each starting 64 KiB further): hits after the first launch
  run<DEPTH>("L2-resident (2 MiB per XCD)", src, 2 * MiB, 2 * MiB, 64 << 10, 256, 256, cyc, 3);
  run<DEPTH>("L2-resident (2 MiB per XCD)", src, 2 * MiB, 2 * MiB, 64 << 10, 2304, 256, cyc, 3);
  // Infinity Cache: 128 MiB in all, every block its own 512 KiB: the second launch finds it in the 256-MiB MALL, not in L2 (32 MiB)
  run<DEPTH>("MALL (128 MiB span, own 512 KiB)", src, 16 * MiB, 16 * MiB, 512 << 10, 256, 64, cyc, 3);
  // HBM: 2 GiB in all, every block its own 8 MiB
  run<DEPTH>("HBM (2 GiB span, own 8 MiB)", src, 256 * MiB, 256 * MiB, 8 * MiB, 256, 1024, cyc, 2);
}

int main(int argc, char **argv) {
  if (argc > 1) g_blocks = atoi(argv[1]);
  if (argc > 2) g_vgpr = atoi(argv[2]);
  printf("%d blocks, %s\n", g_blocks, g_vgpr == 2 ? "waves 0-1 LDS-DMA, waves 2-3 through registers" : g_vgpr ? "global_load_dwordx4 -> VGPR -> ds_write_b128" : "global_load_lds_dwordx4");
  char *src; unsigned long long *cyc;
  const size_t total = (size_t)2 << 30;
  if (hipMalloc(&src, total + (1 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(src, 1, total + (1 << 20));
  hipMalloc(&cyc, 256 * 8);
  suite<1>(src, cyc);
  suite<2>(src, cyc);
  suite<4>(src, cyc);
  suite<8>(src, cyc);
  suite<16>(src, cyc);
  return 0;
}
