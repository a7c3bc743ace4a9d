// Memory-bound NHWC layer kernels: max-pool, nearest upsample, add, channel-slice copy.
// One thread moves one 16-byte channel chunk of one output pixel when strides allow it
// (VEC = 16 / sizeof(T)), else one element (VEC = 1).  All of them take pixel strides
// (in_ld / out_ld) so producers write straight into route-concat buffers
// (/root/reference/yolov3/darknet.py:369-375) and no concat copy is needed on the usual path.
#include "common.h"
#include <initializer_list>

namespace {

struct LayerArgs {
  const void *in;
  const void *in2;
  void *out;
  int B, H, W, C, in_ld, in2_ld, Ho, Wo, out_ld, k, stride, zero_pad;
  long long total;  // B*Ho*Wo*(C/VEC)
};

template <typename T, int VEC>
struct Vec {
  T v[VEC];
};

template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const T *p, float out[VEC]) {
  if constexpr (VEC == 1) {
    out[0] = y3_to_float<T>(*p);
  } else {
    const u32x4 raw = *reinterpret_cast<const u32x4 *>(p);
    if constexpr (sizeof(T) == 4) {
      const f32x4 f = __builtin_bit_cast(f32x4, raw);
#pragma unroll
      for (int j = 0; j < 4; ++j) out[j] = f[j];
    } else {
      float v8[8];
      y3_unpack8<T>(v8, raw);
#pragma unroll
      for (int j = 0; j < 8; ++j) out[j] = v8[j];
    }
  }
}

template <typename T, int VEC>
__device__ __forceinline__ void store_vec(T *p, const float in[VEC]) {
  if constexpr (VEC == 1) {
    *p = y3_from_float<T>(in[0]);
  } else if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<f32x4 *>(p) = f32x4{in[0], in[1], in[2], in[3]};
  } else {
    float v8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v8[j] = in[j];
    *reinterpret_cast<u32x4 *>(p) = y3_pack8<T>(v8);
  }
}

__device__ __forceinline__ void decode_idx(long long idx, int cgroups, int Wo, int Ho, int &b, int &oy,
                                           int &ox, int &cg) {
  cg = (int)(idx % cgroups);
  long long pix = idx / cgroups;
  ox = (int)(pix % Wo);
  pix /= Wo;
  oy = (int)(pix % Ho);
  b = (int)(pix / Ho);
}

// Reference MaxPool2d.forward (darknet.py:21-29): stride 1 & k > 1 -> window [y, y+k) x [x, x+k)
// with out-of-range taps contributing 0.0 (zero padding right/bottom); otherwise floor-mode
// pooling without padding (all taps in range).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool_kernel(LayerArgs p) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.total) return;
  int b, oy, ox, cg;
  decode_idx(idx, p.C / VEC, p.Wo, p.Ho, b, oy, ox, cg);
  float best[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) best[j] = -INFINITY;
  bool padded = false;
  for (int ky = 0; ky < p.k; ++ky) {
    const int iy = oy * p.stride + ky;
    for (int kx = 0; kx < p.k; ++kx) {
      const int ix = ox * p.stride + kx;
      if (iy >= p.H || ix >= p.W) {
        padded = true;
        continue;
      }
      float v[VEC];
      load_vec<T, VEC>(static_cast<const T *>(p.in) + (((long long)b * p.H + iy) * p.W + ix) * p.in_ld + cg * VEC, v);
#pragma unroll
      for (int j = 0; j < VEC; ++j) best[j] = fmaxf(best[j], v[j]);
    }
  }
  if (padded && p.zero_pad) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) best[j] = fmaxf(best[j], 0.f);
  }
  store_vec<T, VEC>(static_cast<T *>(p.out) + (((long long)b * p.Ho + oy) * p.Wo + ox) * p.out_ld + cg * VEC, best);
}

// nn.Upsample(scale_factor, mode="nearest") (darknet.py:302-305)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void upsample_kernel(LayerArgs p) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.total) return;
  int b, oy, ox, cg;
  decode_idx(idx, p.C / VEC, p.Wo, p.Ho, b, oy, ox, cg);
  const int iy = oy / p.stride, ix = ox / p.stride;
  float v[VEC];
  load_vec<T, VEC>(static_cast<const T *>(p.in) + (((long long)b * p.H + iy) * p.W + ix) * p.in_ld + cg * VEC, v);
  store_vec<T, VEC>(static_cast<T *>(p.out) + (((long long)b * p.Ho + oy) * p.Wo + ox) * p.out_ld + cg * VEC, v);
}

// shortcut that could not be fused into a conv epilogue (darknet.py:379)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void add_kernel(LayerArgs p) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.total) return;
  int b, oy, ox, cg;
  decode_idx(idx, p.C / VEC, p.Wo, p.Ho, b, oy, ox, cg);
  const long long pix = ((long long)b * p.Ho + oy) * p.Wo + ox;
  float x[VEC], y[VEC];
  load_vec<T, VEC>(static_cast<const T *>(p.in) + pix * p.in_ld + cg * VEC, x);
  load_vec<T, VEC>(static_cast<const T *>(p.in2) + pix * p.in2_ld + cg * VEC, y);
#pragma unroll
  for (int j = 0; j < VEC; ++j) x[j] += y[j];
  store_vec<T, VEC>(static_cast<T *>(p.out) + pix * p.out_ld + cg * VEC, x);
}

// route member that could not be produced in place (darknet.py:372-375)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void copy_kernel(LayerArgs p) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.total) return;
  int b, oy, ox, cg;
  decode_idx(idx, p.C / VEC, p.Wo, p.Ho, b, oy, ox, cg);
  const long long pix = ((long long)b * p.Ho + oy) * p.Wo + ox;
  float x[VEC];
  load_vec<T, VEC>(static_cast<const T *>(p.in) + pix * p.in_ld + cg * VEC, x);
  store_vec<T, VEC>(static_cast<T *>(p.out) + pix * p.out_ld + cg * VEC, x);
}

enum Which { MAXPOOL, UPSAMPLE, ADD, COPY };

template <typename T, int VEC>
void launch_one(Which w, const LayerArgs &a, hipStream_t s) {
  const dim3 grid((unsigned)((a.total + 255) / 256)), block(256);
  switch (w) {
    case MAXPOOL: Y3_LAUNCH((maxpool_kernel<T, VEC>), grid, block, 0, s, a); break;
    case UPSAMPLE: Y3_LAUNCH((upsample_kernel<T, VEC>), grid, block, 0, s, a); break;
    case ADD: Y3_LAUNCH((add_kernel<T, VEC>), grid, block, 0, s, a); break;
    case COPY: Y3_LAUNCH((copy_kernel<T, VEC>), grid, block, 0, s, a); break;
  }
}

int launch_layer(Which w, const y3_op &op, const void *d_in, hipStream_t s, bool dry_run) {
  LayerArgs a;
  a.in = d_in;
  a.in2 = op.d_res;
  a.out = op.d_out;
  a.B = op.batch; a.H = op.in_h; a.W = op.in_w; a.C = op.in_c; a.in_ld = op.in_ld; a.in2_ld = op.res_ld;
  a.Ho = op.out_h; a.Wo = op.out_w; a.out_ld = op.out_ld;
  a.k = op.ksize; a.stride = op.stride;
  a.zero_pad = (op.ksize > 1 && op.stride == 1) ? 1 : 0;
  Y3_REQUIRE(op.in_c == op.out_c, "block %d: channel count changes in a pool/upsample/add/copy op", op.block_idx);
  const int es = y3_elem_size(op.dtype);
  const int vec = 16 / es;
  bool wide = op.in_c % vec == 0 && op.in_ld % vec == 0 && op.out_ld % vec == 0 &&
              ((uintptr_t)d_in % 16 == 0) && ((uintptr_t)op.d_out % 16 == 0);
  if (w == ADD) wide = wide && op.res_ld % vec == 0 && ((uintptr_t)op.d_res % 16 == 0);
  a.total = (long long)op.batch * op.out_h * op.out_w * (wide ? op.in_c / vec : op.in_c);
  if (dry_run) return Y3_OK;
  return y3_by_dtype(op.dtype, [&](auto tag) {
    typedef decltype(tag) T;
    if (wide) launch_one<T, 16 / sizeof(T)>(w, a, s); else launch_one<T, 1>(w, a, s);
    Y3_HIP_CHECK(hipGetLastError());
    return Y3_OK;
  });
}

}  // namespace

int y3_launch_maxpool(const y3_op &op, const void *d_in, hipStream_t s, const char **kernel_name,
                      bool dry_run) {
  Y3_REQUIRE(op.ksize >= 1 && op.stride >= 1, "maxpool block %d: bad size/stride", op.block_idx);
  if (op.stride == 1) {
    Y3_REQUIRE(op.out_h == op.in_h && op.out_w == op.in_w, "maxpool block %d: stride-1 keeps H,W", op.block_idx);
  } else {
    Y3_REQUIRE(op.out_h == (op.in_h - op.ksize) / op.stride + 1 && op.out_w == (op.in_w - op.ksize) / op.stride + 1,
               "maxpool block %d: output size mismatch", op.block_idx);
  }
  *kernel_name = Y3_KNAME(op.dtype, "maxpool_", "");
  return launch_layer(MAXPOOL, op, d_in, s, dry_run);
}

int y3_launch_upsample(const y3_op &op, const void *d_in, hipStream_t s, const char **kernel_name,
                       bool dry_run) {
  Y3_REQUIRE(op.stride >= 1 && op.out_h == op.in_h * op.stride && op.out_w == op.in_w * op.stride,
             "upsample block %d: output size mismatch", op.block_idx);
  *kernel_name = Y3_KNAME(op.dtype, "upsample_", "");
  return launch_layer(UPSAMPLE, op, d_in, s, dry_run);
}

int y3_launch_add(const y3_op &op, const void *d_in, hipStream_t s, const char **kernel_name, bool dry_run) {
  Y3_REQUIRE(op.out_h == op.in_h && op.out_w == op.in_w, "add block %d: size mismatch", op.block_idx);
  *kernel_name = Y3_KNAME(op.dtype, "add_", "");
  return launch_layer(ADD, op, d_in, s, dry_run);
}

int y3_launch_copy(const y3_op &op, const void *d_in, hipStream_t s, const char **kernel_name, bool dry_run) {
  Y3_REQUIRE(op.out_h == op.in_h && op.out_w == op.in_w, "copy block %d: size mismatch", op.block_idx);
  *kernel_name = Y3_KNAME(op.dtype, "copy_", "");
  return launch_layer(COPY, op, d_in, s, dry_run);
}

// ------------------------------------------------------------------------------------------------
// SPP pyramid (models/yolov3-spp.cfg:576-595): three stride-1 "same" max-pools of sizes 5 / 9 / 13 that read ONE
// tensor and feed one route concat, as ONE launch.  Reference semantics (darknet.py:16-29): window
// [y, y+k) x [x, x+k), out-of-range taps contribute 0.0 (ZeroPad2d right/bottom, then an unpadded pool).
//
// One workgroup = one frame x one 32-byte channel group (16 bf16 / 8 float32 channels).  The (H+12) x (W+12)
// zero-padded image of that group is staged in LDS once (16-byte vectors, two per position), then the pyramid is
// a cascade, exact because max is associative and the windows nest:
//   R5[y][x]  = max(A[y][x .. x+4])                               (row pass)
//   P5[y][x]  = max(R5[y .. y+4][x])                              on (H+8) x (W+8)
//   P9[y][x]  = max(P5[y][x], P5[y+4][x], P5[y][x+4], P5[y+4][x+4])   on (H+4) x (W+4)   ([y,y+9) = [y,y+5) u [y+4,y+9))
//   P13[y][x] = max of the same four taps of P9                   on H x W
// 4 + 4 + 3 + 3 = 14 LDS reads per output vector instead of 25 + 81 + 169 global loads, and each result goes
// straight into its channel slice of the concat buffer (out pointers / pixel stride of the three ops).
namespace {

struct SppArgs {
  const char *in;
  char *out5, *out9, *out13;
  int H, W, in_ld, ld5, ld9, ld13, cgroups;
};

template <typename T>
__device__ __forceinline__ u32x4 vmax16(const u32x4 &a, const u32x4 &b) {
  if constexpr (sizeof(T) == 4) {
    const f32x4 x = __builtin_bit_cast(f32x4, a), y = __builtin_bit_cast(f32x4, b);
    return __builtin_bit_cast(u32x4, f32x4{fmaxf(x[0], y[0]), fmaxf(x[1], y[1]), fmaxf(x[2], y[2]), fmaxf(x[3], y[3])});
  } else {
    typedef typename H16<T>::v8 V8;
    const V8 x = __builtin_bit_cast(V8, a), y = __builtin_bit_cast(V8, b);
    V8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (float)x[j] >= (float)y[j] ? x[j] : y[j];
    return __builtin_bit_cast(u32x4, r);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_spp_kernel(SppArgs p) {
  extern __shared__ __attribute__((aligned(16))) u32x4 spp_lds[];
  constexpr int ES = sizeof(T), V = 2;                 // 16-byte vectors per position
  const int PH = p.H + 12, PW = p.W + 12;
  const int b = blockIdx.x / p.cgroups, cg = blockIdx.x - b * p.cgroups;
  u32x4 *A = spp_lds, *Bf = spp_lds + PH * PW * V;
  const int tid = threadIdx.x;
  const long long frame = (long long)b * p.H * p.W;
  const u32x4 zero = u32x4{0u, 0u, 0u, 0u};
  // 1. stage the zero-padded image
  for (int idx = tid; idx < PH * PW * V; idx += 256) {
    const int pos = idx >> 1, v = idx & 1, y = pos / PW, x = pos - y * PW;
    A[idx] = (y < p.H && x < p.W)
                 ? *reinterpret_cast<const u32x4 *>(p.in + ((frame + y * p.W + x) * p.in_ld) * ES + cg * 32 + v * 16)
                 : zero;
  }
  __syncthreads();
  // 2. row pass of the 5 x 5 pool
  const int W5 = PW - 4, H5 = PH - 4;
  for (int idx = tid; idx < PH * W5 * V; idx += 256) {
    const int pos = idx >> 1, v = idx & 1, y = pos / W5, x = pos - y * W5;
    const u32x4 *a = A + (y * PW + x) * V + v;
    u32x4 m = vmax16<T>(a[0], a[V]);
    m = vmax16<T>(m, a[2 * V]);
    m = vmax16<T>(m, a[3 * V]);
    Bf[(y * PW + x) * V + v] = vmax16<T>(m, a[4 * V]);
  }
  __syncthreads();
  // 3. column pass -> P5 on (H+8) x (W+8), kept in A; the H x W part is the 5 x 5 pool's output
  for (int idx = tid; idx < H5 * W5 * V; idx += 256) {
    const int pos = idx >> 1, v = idx & 1, y = pos / W5, x = pos - y * W5;
    const u32x4 *r = Bf + (y * PW + x) * V + v;
    u32x4 m = vmax16<T>(r[0], r[PW * V]);
    m = vmax16<T>(m, r[2 * PW * V]);
    m = vmax16<T>(m, r[3 * PW * V]);
    m = vmax16<T>(m, r[4 * PW * V]);
    A[(y * PW + x) * V + v] = m;
    if (y < p.H && x < p.W)
      *reinterpret_cast<u32x4 *>(p.out5 + ((frame + y * p.W + x) * p.ld5) * ES + cg * 32 + v * 16) = m;
  }
  __syncthreads();
  // 4. P9 on (H+4) x (W+4) from four taps of P5, kept in Bf
  const int W9 = PW - 8, H9 = PH - 8;
  for (int idx = tid; idx < H9 * W9 * V; idx += 256) {
    const int pos = idx >> 1, v = idx & 1, y = pos / W9, x = pos - y * W9;
    const u32x4 *a = A + (y * PW + x) * V + v;
    const u32x4 m = vmax16<T>(vmax16<T>(a[0], a[4 * V]), vmax16<T>(a[4 * PW * V], a[(4 * PW + 4) * V]));
    Bf[(y * PW + x) * V + v] = m;
    if (y < p.H && x < p.W)
      *reinterpret_cast<u32x4 *>(p.out9 + ((frame + y * p.W + x) * p.ld9) * ES + cg * 32 + v * 16) = m;
  }
  __syncthreads();
  // 5. P13 on H x W from four taps of P9
  for (int idx = tid; idx < p.H * p.W * V; idx += 256) {
    const int pos = idx >> 1, v = idx & 1, y = pos / p.W, x = pos - y * p.W;
    const u32x4 *a = Bf + (y * PW + x) * V + v;
    const u32x4 m = vmax16<T>(vmax16<T>(a[0], a[4 * V]), vmax16<T>(a[4 * PW * V], a[(4 * PW + 4) * V]));
    *reinterpret_cast<u32x4 *>(p.out13 + ((frame + y * p.W + x) * p.ld13) * ES + cg * 32 + v * 16) = m;
  }
}

size_t spp_lds_bytes(const y3_op &op) { return (size_t)(op.in_h + 12) * (op.in_w + 12) * 2 * 16 * 2; }

}  // namespace

// ops[0..2]: three consecutive max-pool ops of a plan.  True when they form the SPP pyramid this kernel computes:
// sizes {5, 9, 13} in any order, stride 1, the same input view, 32-byte channel groups, and the image fits LDS.
bool y3_maxpool_spp_supported(const y3_op &a, const y3_op &b, const y3_op &c) {
  const y3_op *o[3] = {&a, &b, &c};
  int seen = 0;
  for (const y3_op *q : o) {
    if (q->kind != Y3_OP_MAXPOOL || q->stride != 1) return false;
    if (q->ksize == 5) seen |= 1; else if (q->ksize == 9) seen |= 2; else if (q->ksize == 13) seen |= 4; else return false;
    if (q->d_in != a.d_in || q->in_ld != a.in_ld || q->in_h != a.in_h || q->in_w != a.in_w || q->in_c != a.in_c ||
        q->batch != a.batch || q->dtype != a.dtype || (q->flags & Y3_F_PLAN_INPUT))
      return false;
    const int es = y3_elem_size(q->dtype), vec = 16 / es;
    if (q->in_c % (2 * vec) || q->in_ld % vec || q->out_ld % vec || ((uintptr_t)q->d_in % 16) || ((uintptr_t)q->d_out % 16))
      return false;
    if (q->out_h != q->in_h || q->out_w != q->in_w || q->out_c != q->in_c) return false;
  }
  return seen == 7 && spp_lds_bytes(a) <= 64 * 1024;
}

int y3_launch_maxpool_spp(const y3_op &a, const y3_op &b, const y3_op &c, hipStream_t s, const char **kernel_name,
                          bool dry_run) {
  *kernel_name = Y3_KNAME(a.dtype, "maxpool_spp_pyramid_", "");
  if (dry_run) return Y3_OK;
  const y3_op *o[3] = {&a, &b, &c};
  SppArgs p;
  p.in = static_cast<const char *>(a.d_in);
  p.H = a.in_h; p.W = a.in_w; p.in_ld = a.in_ld;
  p.cgroups = a.in_c * y3_elem_size(a.dtype) / 32;
  for (const y3_op *q : o) {
    if (q->ksize == 5) { p.out5 = static_cast<char *>(q->d_out); p.ld5 = q->out_ld; }
    else if (q->ksize == 9) { p.out9 = static_cast<char *>(q->d_out); p.ld9 = q->out_ld; }
    else { p.out13 = static_cast<char *>(q->d_out); p.ld13 = q->out_ld; }
  }
  const size_t lds = spp_lds_bytes(a);
  const dim3 grid((unsigned)(a.batch * p.cgroups)), block(256);
  return y3_by_dtype(a.dtype, [&](auto tag) {
    Y3_LAUNCH((maxpool_spp_kernel<decltype(tag)>), grid, block, lds, s, p);
    Y3_HIP_CHECK(hipGetLastError());
    return Y3_OK;
  });
}

// ------------------------------------------------------------------------------------------------
// Frame resize on device: uint8 HxWx3 -> net_h x net_w x 3, OpenCV's 8-bit INTER_LINEAR
// arithmetic.  Replaces the host `cv2.resize(image, (net_h, net_w))` of /root/reference/yolov3/inference.py:320-326
// for frames that are not net-sized.  Integer arithmetic only, identical to yolov3/preprocess.py:
// resize_bilinear_u8 (the tap tables -- lo index, hi index, 11-bit weights per output row / column, OpenCV's
// xofs / ialpha / yofs / ibeta -- are computed once on the host and passed in: host and device are bit-identical).
namespace {
__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t *src, int sh, int sw, uint8_t *dst, int dh,
                                                        int dw, const int *ytab, const int *xtab) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= dh * dw) return;
  const int y = idx / dw, x = idx - y * dw;
  const int ylo = ytab[y * 4 + 0], yhi = ytab[y * 4 + 1], wy0 = ytab[y * 4 + 2], wy1 = ytab[y * 4 + 3];
  const int xlo = xtab[x * 4 + 0], xhi = xtab[x * 4 + 1], wx0 = xtab[x * 4 + 2], wx1 = xtab[x * 4 + 3];
  const uint8_t *r0 = src + (long long)ylo * sw * 3, *r1 = src + (long long)yhi * sw * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    // OpenCV's two truncating stages (resize.cpp: HResizeLinear to int, then VResizeLinear<uchar, int, short, ...>)
    const int top = r0[xlo * 3 + c] * wx0 + r0[xhi * 3 + c] * wx1;
    const int bot = r1[xlo * 3 + c] * wx0 + r1[xhi * 3 + c] * wx1;
    int v = (((wy0 * (top >> 4)) >> 16) + ((wy1 * (bot >> 4)) >> 16) + 2) >> 2;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    dst[(long long)idx * 3 + c] = (uint8_t)v;
  }
}

// Plain byte mover for buffers that a copy engine would otherwise carry: `blocks` workgroups stride over the buffer with
// 16-byte accesses, four in flight per thread.  Either side may be pinned host memory (the GPU addresses it directly).
__global__ __launch_bounds__(256) void copy_bytes_kernel(const char *__restrict__ src, char *__restrict__ dst, size_t n16, size_t tail0,
                                                          size_t nbytes) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const u32x4 *s4 = reinterpret_cast<const u32x4 *>(src);
  u32x4 *d4 = reinterpret_cast<u32x4 *>(dst);
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const u32x4 a = s4[i], b = s4[i + stride], c = s4[i + 2 * stride], d = s4[i + 3 * stride];
    d4[i] = a; d4[i + stride] = b; d4[i + 2 * stride] = c; d4[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) d4[i] = s4[i];
  if (blockIdx.x == 0) for (size_t t = tail0 + threadIdx.x; t < nbytes; t += 256) dst[t] = src[t];
}
}  // namespace

extern "C" int y3_copy_bytes(const void *src, void *dst, size_t nbytes, int blocks, void *stream) {
  Y3_REQUIRE(src && dst, "y3_copy_bytes: null pointer argument");
  Y3_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "y3_copy_bytes: buffers must be 16-byte aligned");
  if (nbytes == 0) return Y3_OK;
  // a pageable host pointer would be a GPU memory fault inside the kernel, not an error code: both ends must be device
  // memory or pinned / registered host memory (what the runtime knows an address for)
  for (const void *ptr : {src, static_cast<const void *>(dst)}) {
    hipPointerAttribute_t attr;
    const hipError_t e = hipPointerGetAttributes(&attr, ptr);
    if (e != hipSuccess || attr.type == hipMemoryTypeUnregistered) {
      (void)hipGetLastError();
      y3_set_error("y3_copy_bytes: %p is neither device memory nor pinned (registered) host memory", ptr);
      return Y3_ERR_INVALID;
    }
  }
  if (blocks < 1) blocks = 32;
  if (blocks > 1024) blocks = 1024;
  Y3_LAUNCH(copy_bytes_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const char *>(src), static_cast<char *>(dst), nbytes / 16, nbytes / 16 * 16, nbytes);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}

extern "C" int y3_resize_bilinear_u8(const uint8_t *d_src, int src_h, int src_w, uint8_t *d_dst, int dst_h, int dst_w,
                                     const int32_t *d_ytab, const int32_t *d_xtab, void *stream) {
  Y3_REQUIRE(d_src && d_dst && d_ytab && d_xtab, "y3_resize_bilinear_u8: null pointer argument");
  Y3_REQUIRE(src_h > 0 && src_w > 0 && dst_h > 0 && dst_w > 0, "y3_resize_bilinear_u8: empty image");
  Y3_LAUNCH(resize_u8_kernel, dim3((dst_h * dst_w + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     d_src, src_h, src_w, d_dst, dst_h, dst_w, d_ytab, d_xtab);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}
