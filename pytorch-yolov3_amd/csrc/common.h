// Shared device/host helpers for libyolov3_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>

#include <mutex>

#include "yolov3_hip.h"

typedef __bf16 bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define Y3_WAVE 64
// cache policy of the LDS-DMA loads (the builtin's last argument: gfx940+ CPol bits, 1 = sc0, 2 = nt, 16 = sc1): 0 in the
// product; experiment builds set them per operand (`make variant FLAGS="-DY3_AUX_W=2"`: weight tiles non-temporal, i.e. past L1)
#ifndef Y3_AUX_W
#define Y3_AUX_W 0
#endif
#ifndef Y3_AUX_A
#define Y3_AUX_A 0
#endif
#ifndef Y3_AUX_H
#define Y3_AUX_H 0        // strip kernel: the halo (pixel) slices
#endif
#define Y3_LEAKY_SLOPE 0.1f

// diagnostic builds only (y3_set_tuning("debug", v)): 0 in the product
int y3_debug_flags();
// CU count of the current device (api.hip); the launchers' grid-size heuristics scale with it
int y3_device_cus();

// thread-local error string shared by all translation units
void y3_set_error(const char *fmt, ...);

#define Y3_HIP_CHECK(expr)                                                               \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      y3_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return Y3_ERR_HIP;                                                                 \
    }                                                                                    \
  } while (0)

#define Y3_REQUIRE(cond, ...)     \
  do {                            \
    if (!(cond)) {                \
      y3_set_error(__VA_ARGS__);  \
      return Y3_ERR_INVALID;      \
    }                             \
  } while (0)

// Every kernel of the library is launched through Y3_LAUNCH.  Normally that is hipLaunchKernelGGL.  Inside
// y3_plan_run_profiled (api.hip) the calling thread has a Y3KernelTimer set and the launch goes through
// hipExtLaunchKernelGGL with a start / stop event pair BOUND TO THE DISPATCH: their elapsed time is the kernel's own
// begin -> end on the device (the figure rocprofv3's kernel trace reports), without the ~5 us of dispatch and barrier-packet
// handling that an event recorded on the stream before and after a launch includes (y3_plan_run_timed).
struct Y3KernelTimer {
  hipEvent_t *start, *stop;
  int n, cap;
};
Y3KernelTimer *y3_kernel_timer();
#define Y3_LAUNCH(kernel, grid, block, lds, stream, ...)                                                             \
  do {                                                                                                               \
    Y3KernelTimer *_kt = y3_kernel_timer();                                                                          \
    if (_kt && _kt->n < _kt->cap) {                                                                                  \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, _kt->start[_kt->n], _kt->stop[_kt->n], 0, __VA_ARGS__); \
      ++_kt->n;                                                                                                      \
    } else {                                                                                                         \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                             \
    }                                                                                                                \
  } while (0)

// One-time set-up per DEVICE (dynamic-LDS function attributes are per device; so is the CU count a persistent grid is
// sized by), safe to call from several host threads.
struct Y3DeviceOnce {
  std::mutex mu;
  bool done[32] = {};
  int n_cu[32] = {};
  template <typename F>
  int run(F &&setup, int *cu_out = nullptr) {
    int dev = 0;
    Y3_HIP_CHECK(hipGetDevice(&dev));
    Y3_REQUIRE(dev >= 0 && dev < 32, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(mu);
    if (!done[dev]) {
      const int rc = setup();
      if (rc != Y3_OK) return rc;
      Y3_HIP_CHECK(hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev));
      done[dev] = true;
    }
    if (cu_out) *cu_out = n_cu[dev];
    return Y3_OK;
  }
};

// 16-bit storage modes (bf16, IEEE half): the same kernels, instantiated per element type
static inline bool y3_is16(int dtype) { return dtype == Y3_BF16 || dtype == Y3_F16; }
static inline int y3_elem_size(int dtype) { return y3_is16(dtype) ? 2 : 4; }
// kernel names carry the element type: Y3_KNAME(dtype, "conv_halo_ws_", "_256x128") -> "conv_halo_ws_bf16_256x128"
#define Y3_KNAME(dt, pre, post) ((dt) == Y3_BF16 ? pre "bf16" post : ((dt) == Y3_F16 ? pre "f16" post : pre "f32" post))
// host-side dispatch on the element type: f(T{}) with T = bf16_t / f16_t / float (a generic lambda: `using T = decltype(tag)`)
template <typename F>
static inline int y3_by_dtype(int dtype, F &&f) {
  if (dtype == Y3_BF16) return f(bf16_t{});
  if (dtype == Y3_F16) return f(f16_t{});
  return f(float{});
}
template <typename F>
static inline int y3_by_dtype16(int dtype, F &&f) {   // kernels that exist for the 16-bit modes only
  return dtype == Y3_F16 ? f(f16_t{}) : f(bf16_t{});
}
static inline int y3_ceil_div(int a, int b) { return (a + b - 1) / b; }
// n / d == (umulhi(n, mul) + n) >> sh for 0 <= n < 2^31 (round-up method, d >= 1)
static inline void y3_fast_div(uint32_t d, uint32_t &mul, uint32_t &sh) {
  if (d <= 1) { mul = 0; sh = 0; return; }
  sh = 0;
  while ((1u << sh) < d) ++sh;
  mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << sh) - d)) / d + 1);
}

// Workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2); give each
// XCD one contiguous run of logical tile ids so that neighbouring tiles (which share an input
// halo / weight panel) hit the same L2.  Bijective for any grid size.  Placement only affects
// speed, never results.
__device__ __forceinline__ int y3_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Output-channel placement that makes a lane's accumulators of an MFMA fragment PAIR eight consecutive channels.
// v_mfma_f32_16x16x32 leaves lane (fr, fq) with rows 4 fq .. 4 fq + 3 of each 16-row fragment for column (pixel) fr.  If,
// inside every aligned block of 32 weight rows (fragments h = 0, 1), row h*16 + 4q + i carries channel 8q + 4h + i, the
// lane holds channels 8 fq .. 8 fq + 7 of its pixel after both fragments: one 16-byte LDS write (8-lane groups, no
// conflicts with the usual XOR swizzle or an odd multiple of 16 bytes as pixel pitch) instead of two 8-byte ones whose
// 16-lane groups hit every 128-byte bank window twice or four times (tools/lds_conflicts.py).  Which row of a
// fragment a channel occupies does not enter any sum: results are bit-identical.
__host__ __device__ __forceinline__ constexpr int y3_pair_perm(int j) {
  return (j & ~31) | (((j >> 2) & 3) << 3) | (((j >> 4) & 1) << 2) | (j & 3);
}

template <typename T>
__device__ __forceinline__ float y3_to_float(T v);
template <>
__device__ __forceinline__ float y3_to_float<float>(float v) { return v; }
template <>
__device__ __forceinline__ float y3_to_float<bf16_t>(bf16_t v) { return (float)v; }
template <>
__device__ __forceinline__ float y3_to_float<f16_t>(f16_t v) { return (float)v; }

template <typename T>
__device__ __forceinline__ T y3_from_float(float v);
template <>
__device__ __forceinline__ float y3_from_float<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_t y3_from_float<bf16_t>(float v) { return (bf16_t)v; }
template <>
__device__ __forceinline__ f16_t y3_from_float<f16_t>(float v) { return (f16_t)v; }   // v_cvt_f16_f32: nearest even, like numpy / torch

// The two 16-bit element types behind one interface: 8- / 4-wide vectors, the MFMA that consumes them (both issue at the
// same rate on gfx950: 16 passes for 16x16x32), conversions.  float32 accumulation in both.
template <typename T>
struct H16;
template <>
struct H16<bf16_t> {
  typedef bf16x8 v8;
  typedef bf16x4 v4;
  static __device__ __forceinline__ f32x4 mfma(const v8 &w, const v8 &x, const f32x4 &acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc, 0, 0, 0);
  }
};
template <>
struct H16<f16_t> {
  typedef f16x8 v8;
  typedef f16x4 v4;
  static __device__ __forceinline__ f32x4 mfma(const v8 &w, const v8 &x, const f32x4 &acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, acc, 0, 0, 0);
  }
};
// acc += W (16 rows x 32 k) * X (32 k x 16 columns); operands as the 16 raw bytes a lane holds
template <typename T>
__device__ __forceinline__ f32x4 y3_mfma16(const u32x4 &w, const u32x4 &x, const f32x4 &acc) {
  typedef typename H16<T>::v8 v8;
  return H16<T>::mfma(__builtin_bit_cast(v8, w), __builtin_bit_cast(v8, x), acc);
}
// eight consecutive stored elements (16 raw bytes) added to v[0..7]: the shortcut operand of a conv epilogue
template <typename T>
__device__ __forceinline__ void y3_add8(float (&v)[8], const u32x4 &raw) {
  const typename H16<T>::v8 r = __builtin_bit_cast(typename H16<T>::v8, raw);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] += (float)r[i];
}
template <typename T>
__device__ __forceinline__ void y3_unpack8(float (&v)[8], const u32x4 &raw) {
  const typename H16<T>::v8 r = __builtin_bit_cast(typename H16<T>::v8, raw);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
}
// v[0..7] rounded to the storage type (nearest even), as the 16 bytes of one store
template <typename T>
__device__ __forceinline__ u32x4 y3_pack8(const float (&v)[8]) {
  typename H16<T>::v8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (T)v[i];
  return __builtin_bit_cast(u32x4, o);
}
template <typename T>
__device__ __forceinline__ u32x2 y3_pack4(float a, float b, float c, float d) {
  const typename H16<T>::v4 o = {(T)a, (T)b, (T)c, (T)d};
  return __builtin_bit_cast(u32x2, o);
}

// Conv epilogue arithmetic for eight consecutive output channels: acc * scale + bias, then LeakyReLU(0.1).  Written
// on 2-wide vectors so that hipcc emits v_pk_fma_f32 / v_pk_mul_f32, and with v_max_f32 spelled out: max(v, 0.1 v) is
// v > 0 ? v : 0.1 v bit for bit (both operands carry v's sign), one instruction instead of compare + select, and fmaxf
// would add a canonicalising v_max per value.  The epilogues are VALU-bound (MI355X: 4 cycles per instruction and
// wave, 64 values per lane and tile).
__device__ __forceinline__ float y3_vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// `leaky` is a run-time flag of the layer: a slope of 1 makes max(v, slope * v) the identity without a branch (phi
// copies of all eight values otherwise)
__device__ __forceinline__ void y3_bn_leaky8(float (&v)[8], const f32x4 &lo, const f32x4 &hi, const f32x4 &sc_lo,
                                             const f32x4 &sc_hi, const f32x4 &bi_lo, const f32x4 &bi_hi, bool leaky) {
  const float slope = leaky ? Y3_LEAKY_SLOPE : 1.0f;
  f32x2 t[4];
  t[0] = f32x2{lo[0], lo[1]} * f32x2{sc_lo[0], sc_lo[1]} + f32x2{bi_lo[0], bi_lo[1]};
  t[1] = f32x2{lo[2], lo[3]} * f32x2{sc_lo[2], sc_lo[3]} + f32x2{bi_lo[2], bi_lo[3]};
  t[2] = f32x2{hi[0], hi[1]} * f32x2{sc_hi[0], sc_hi[1]} + f32x2{bi_hi[0], bi_hi[1]};
  t[3] = f32x2{hi[2], hi[3]} * f32x2{sc_hi[2], sc_hi[3]} + f32x2{bi_hi[2], bi_hi[3]};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 s = t[i] * slope;
    v[2 * i] = y3_vmax(t[i][0], s[0]);
    v[2 * i + 1] = y3_vmax(t[i][1], s[1]);
  }
}

// Diagnostic build only (-DY3_STAMPS, `make stamps`): per-workgroup phase timing with s_memtime.
// Lane 0 of wave 0 adds the cycle count of each phase into g_y3_stamps[phase]; slot 7 counts
// workgroups.  Never compiled into the shipped library.
#ifdef Y3_STAMPS
static __device__ unsigned long long g_y3_stamps[8];  // one copy per translation unit
__device__ __forceinline__ unsigned long long y3_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define Y3_STAMP_DECL                                \
  unsigned long long _st_acc[7] = {0, 0, 0, 0, 0, 0, 0}; \
  unsigned long long _st_prev = y3_now();
// accumulate in registers; nothing touches memory until the kernel's last instruction
#define Y3_STAMP(slot)                            \
  do {                                            \
    const unsigned long long _now = y3_now();     \
    _st_acc[slot] += _now - _st_prev;             \
    _st_prev = _now;                              \
  } while (0)
#define Y3_STAMP_COUNT()                                                         \
  do {                                                                           \
    if (threadIdx.x == 0) {                                                      \
      for (int _i = 0; _i < 7; ++_i) atomicAdd(&g_y3_stamps[_i], _st_acc[_i]);   \
      atomicAdd(&g_y3_stamps[7], 1ull);                                          \
    }                                                                            \
  } while (0)
// defines  extern "C" int NAME(unsigned long long out[8])  that reads + clears this unit's counters
#define Y3_STAMP_READER(NAME)                                                                        \
  extern "C" int NAME(unsigned long long *out8) {                                                    \
    if (hipDeviceSynchronize() != hipSuccess) return Y3_ERR_HIP;                                     \
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_y3_stamps), 64) != hipSuccess) return Y3_ERR_HIP;     \
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                           \
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_y3_stamps), zero, 64) != hipSuccess) return Y3_ERR_HIP;       \
    return Y3_OK;                                                                                    \
  }
#else
#define Y3_STAMP_READER(NAME)
#define Y3_STAMP_DECL
#define Y3_STAMP(slot) do {} while (0)
#define Y3_STAMP_COUNT() do {} while (0)
#endif

// Y3_STAMPS_CLOCK (diagnostic, `make stamps STAMP_FLAGS=-DY3_STAMPS_CLOCK`): no stamp inside the K loop; thread 0
// stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around it: slot 0 = loop cycles, slot 1 = loop time in
// 10 ns ticks (in-kernel clock = slot0 / slot1 * 100 MHz: MI355X_MICROARCH.md, DVFS give-back item 6), slot 2 =
// cycles before the loop, slot 3 = cycles after it
#if defined(Y3_STAMPS) && defined(Y3_STAMPS_CLOCK)
#undef Y3_STAMP
#define Y3_STAMP(slot) do {} while (0)
__device__ __forceinline__ unsigned long long y3_now_real() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define Y3_CLK_BEGIN() unsigned long long _clk_r0 = 0; do { const unsigned long long _n = y3_now(); _st_acc[2] = _n - _st_prev; _st_prev = _n; _clk_r0 = y3_now_real(); } while (0)
#define Y3_CLK_END() do { const unsigned long long _n = y3_now(); _st_acc[0] = _n - _st_prev; _st_prev = _n; _st_acc[1] = y3_now_real() - _clk_r0; } while (0)
#ifdef Y3_STAMPS_EPI
// ... and the pieces of the epilogue in slots 4-6 (thread 0: K loop end -> first barrier -> tile parked -> written out)
#define Y3_EPI(slot) do { const unsigned long long _n = y3_now(); _st_acc[slot] = _n - _st_prev; _st_prev = _n; } while (0)
#define Y3_CLK_TAIL() do { if (threadIdx.x == 0) { _st_acc[3] = y3_now() - _st_prev; for (int _i = 0; _i < 7; ++_i) atomicAdd(&g_y3_stamps[_i], _st_acc[_i]); atomicAdd(&g_y3_stamps[7], 1ull); } } while (0)
#else
// slots 4-6: the first round of workgroups (blockIdx < 256) on its own: cycles before the loop, count x 1000, cycles after it
#define Y3_EPI(slot) do {} while (0)
#define Y3_CLK_TAIL() do { if (threadIdx.x == 0) { _st_acc[3] = y3_now() - _st_prev; for (int _i = 0; _i < 4; ++_i) atomicAdd(&g_y3_stamps[_i], _st_acc[_i]); if (blockIdx.x < 256) { atomicAdd(&g_y3_stamps[4], _st_acc[2]); atomicAdd(&g_y3_stamps[5], 1000ull); atomicAdd(&g_y3_stamps[6], _st_acc[3]); } atomicAdd(&g_y3_stamps[7], 1ull); } } while (0)
#endif
#else
#define Y3_CLK_BEGIN() do {} while (0)
#define Y3_CLK_END() do {} while (0)
#define Y3_CLK_TAIL() do {} while (0)
#define Y3_EPI(slot) do {} while (0)
#endif

// Y3_STAMPS_FINE (diagnostic): the ping-pong kernel stamps the pieces of its K-step instead of its phases
#if defined(Y3_STAMPS) && defined(Y3_STAMPS_CLOCK)
#define Y3_FINE(slot) do {} while (0)
#define Y3_COARSE(slot) do {} while (0)
#elif defined(Y3_STAMPS) && defined(Y3_STAMPS_FINE)
#define Y3_FINE(slot) Y3_STAMP(slot)
#define Y3_COARSE(slot) do { _st_prev = y3_now(); } while (0)
#else
#define Y3_FINE(slot) do {} while (0)
#define Y3_COARSE(slot) Y3_STAMP(slot)
#endif

// launchers implemented in the .hip files; each fills *kernel_name with a static string
// force_version / force_ns / force_bm: 0 = the "igemm_version" / "igemm_ns" / "igemm_bm" knobs
int y3_launch_conv_igemm(const y3_op &op, const void *d_in, const void *d_zero, hipStream_t s,
                         const char **kernel_name, bool dry_run, int force_version = 0, int force_ns = 0,
                         int force_bm = 0);
int y3_launch_conv_small(const y3_op &op, const void *d_in, hipStream_t s, const char **kernel_name,
                         bool dry_run);
int y3_launch_conv_direct(const y3_op &op, const void *d_in, hipStream_t s, const char **kernel_name,
                          bool dry_run);
bool y3_conv_stem_mfma_supported(const y3_op &op);
int y3_launch_conv_stem_mfma(const y3_op &op, const void *d_in, hipStream_t s, const char **kernel_name,
                             bool dry_run);
int y3_launch_maxpool(const y3_op &op, const void *d_in, hipStream_t s, const char **kernel_name,
                      bool dry_run);
// SPP pyramid: three stride-1 max-pools (5 / 9 / 13) of one tensor in one launch (layers.hip)
bool y3_maxpool_spp_supported(const y3_op &a, const y3_op &b, const y3_op &c);
int y3_launch_maxpool_spp(const y3_op &a, const y3_op &b, const y3_op &c, hipStream_t s, const char **kernel_name,
                          bool dry_run);
int y3_launch_upsample(const y3_op &op, const void *d_in, hipStream_t s, const char **kernel_name,
                       bool dry_run);
int y3_launch_add(const y3_op &op, const void *d_in, hipStream_t s, const char **kernel_name,
                  bool dry_run);
int y3_launch_copy(const y3_op &op, const void *d_in, hipStream_t s, const char **kernel_name,
                   bool dry_run);
int y3_launch_yolo(const y3_op &op, const void *d_in, hipStream_t s, const char **kernel_name,
                   bool dry_run);
// first two layers of Darknet-53 as one kernel (conv_fused.hip); op1 must read only op0's output
bool y3_conv_fused_stem_s2_supported(const y3_op &op0, const y3_op &op1);
int y3_launch_conv_fused_stem_s2(const y3_op &op0, const y3_op &op1, const void *d_in, hipStream_t s,
                                 const char **kernel_name, bool dry_run);
bool y3_conv_fused_resblock_supported(const y3_op &op0, const y3_op &op1);
int y3_launch_conv_fused_resblock(const y3_op &op0, const y3_op &op1, hipStream_t s, const char **kernel_name,
                                  bool dry_run);
// 1x1 (-> 128 channels) + 3x3 (+ shortcut) in one kernel, bottleneck tensor in LDS (conv_block.hip)
bool y3_conv_block_fused_supported(const y3_op &op0, const y3_op &op1);
int y3_launch_conv_block_fused(const y3_op &op0, const y3_op &op1, hipStream_t s, const char **kernel_name, bool dry_run);
// detection head: 1x1 conv + YOLO decode in one launch (conv_igemm.hip: the tiled form and the choice between the two; conv_1x1.hip:
// the direct-weights form, which reads the fragment-order copy of the head conv's weights: `frag_w`, nullptr = made per launch)
bool y3_conv_head_decode_supported(const y3_op &op0, const y3_op &op1);
int y3_launch_conv_head_decode(const y3_op &op0, const y3_op &op1, const void *d_zero, hipStream_t s,
                               const char **kernel_name, bool dry_run, const void *frag_w = nullptr);
bool y3_conv_head_dw_fits(const y3_op &op0);
int y3_launch_conv_head_decode_dw(const y3_op &op0, const y3_op &op1, const void *d_zero, hipStream_t s, const char **kernel_name,
                                  bool dry_run, const void *frag_w);
// halo-reuse 3x3 kernel (conv_halo.hip): whether it can take this conv, and its launcher
bool y3_conv_halo_ws_fits(const y3_op &op);
bool y3_conv_halo_dw_fits(const y3_op &op);
bool y3_conv_halo_dw_pays(const y3_op &op);
size_t y3_conv_halo_dw_weight_bytes(const y3_op &op);
int y3_conv_halo_dw_make_weights(const y3_op &op, void *dst, hipStream_t s);
int y3_launch_conv_halo_dw(const y3_op &op, const void *d_in, const void *d_zero, hipStream_t s,
                           const char **kernel_name, bool dry_run, const void *frag_w);
int y3_launch_conv_halo(const y3_op &op, const void *d_in, const void *d_zero, hipStream_t s,
                        const char **kernel_name, bool dry_run);
// 1x1 conv with LDS-resident weights, persistent workgroups (conv_1x1.hip)
bool y3_conv1x1_wres_supported(const y3_op &op);
bool y3_conv1x1_wres_pays(const y3_op &op);
int y3_launch_conv1x1_wres(const y3_op &op, const void *d_in, const void *d_zero, hipStream_t s, const char **kernel_name,
                           bool dry_run);
// direct-weights 1x1 kernel for the short-K bottleneck layers on small maps (conv_1x1.hip; fragment-order weights)
bool y3_conv1x1_dw_pays(const y3_op &op);
int y3_launch_conv1x1_dw(const y3_op &op, const void *d_in, const void *d_zero, hipStream_t s, const char **kernel_name,
                         bool dry_run, const void *frag_w);
// small-grid direct-weights kernel, 1x1 and 3x3 (conv_dw48.hip; fragment-order weights): 48-pixel x 32..256-channel tiles
bool y3_conv_dw48_fits(const y3_op &op);
bool y3_conv_dw48_fits_wide(const y3_op &op);
int y3_launch_conv_dw48(const y3_op &op, const void *d_in, const void *d_zero, hipStream_t s, const char **kernel_name,
                        bool dry_run, const void *frag_w);
// 2-D patch form of the halo kernel for rows wider than 128 pixels (conv_halo.hip)
bool y3_conv_patch_fits(const y3_op &op);
int y3_launch_conv_patch(const y3_op &op, const void *d_in, const void *d_zero, hipStream_t s,
                         const char **kernel_name, bool dry_run);
// options of the plan being created / run on this thread (api.hip), else the process defaults (y3_set_tuning)
const y3_options &y3_opt();
// true when the MFMA implicit-GEMM kernel can take this conv
bool y3_conv_igemm_supported(const y3_op &op);
