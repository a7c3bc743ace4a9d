// Convolutions that do not fit the MFMA implicit-GEMM kernel.
//
//  * conv_stem3x3: the network's first layer (Cin = 3, 3x3).  Reads the network input directly
//    -- float32 NCHW in [0,1] (what Darknet.forward receives, /root/reference/yolov3/darknet.py:351)
//    or uint8 NHWC BGR frames with the reference's BGR->RGB flip and /255.0 fused in
//    (/root/reference/yolov3/inference.py:332-333) -- and writes NHWC activations, so no
//    layout-conversion pass exists anywhere.  One thread per output pixel, 27 inputs in
//    registers, weights broadcast from LDS, 8 output channels per register block.
//  * conv_direct: any other shape (odd channel counts, big kernels).  One thread per output
//    element; correctness fallback, not a fast path.
// Both apply the same epilogue as the igemm kernel: *scale + bias, LeakyReLU(0.1), +residual.
#include "common.h"

namespace {

struct SmallArgs {
  const void *in;
  const float *wgt;  // [27][cout_pad] float32
  const float *scale;
  const float *bias;
  char *out;
  int B, H, W, Ho, Wo, Cout, cout_pad, out_ld, stride, pad, M;
  uint32_t flags;
};

// lut[v] = (float)v / 255.0f, the reference's uint8 -> float32 normalisation (inference.py:332-333);
// 256 correctly-rounded divisions per workgroup instead of 27 per thread.
template <int MODE>
__device__ __forceinline__ float load_input(const void *in, const float *lut, int b, int c, int y, int x,
                                            int H, int W) {
  if constexpr (MODE == 0) {  // float32 NCHW
    return static_cast<const float *>(in)[(((long long)b * 3 + c) * H + y) * W + x];
  } else {  // uint8 NHWC, BGR in memory; channel c of the RGB tensor is byte 2-c
    const uint8_t v = static_cast<const uint8_t *>(in)[(((long long)b * H + y) * W + x) * 3 + (2 - c)];
    return lut[v];
  }
}

template <typename TO, int MODE>
__global__ __launch_bounds__(256) void conv_stem3x3_kernel(SmallArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [27][cout_pad] then lut[256]
  const int nw = 27 * p.cout_pad;
  for (int i = threadIdx.x; i < nw; i += 256) sw[i] = p.wgt[i];
  float *lut = sw + nw;
  if constexpr (MODE == 1) lut[threadIdx.x] = (float)threadIdx.x / 255.0f;
  __syncthreads();
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= p.M) return;
  const int hw = p.Ho * p.Wo;
  const int b = m / hw;
  const int rem = m - b * hw;
  const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
  const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
  float x[27];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = iy0 + ky, ix = ix0 + kx;
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        x[(ky * 3 + kx) * 3 + c] = ok ? load_input<MODE>(p.in, lut, b, c, iy, ix, p.H, p.W) : 0.f;
    }
  const bool leaky = p.flags & Y3_F_LEAKY;
  TO *orow = reinterpret_cast<TO *>(p.out) + (long long)m * p.out_ld;
  for (int co = 0; co < p.Cout; co += 8) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const f32x4 w0 = *reinterpret_cast<const f32x4 *>(sw + k * p.cout_pad + co);
      const f32x4 w1 = *reinterpret_cast<const f32x4 *>(sw + k * p.cout_pad + co + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] += x[k] * w0[j];
        acc[4 + j] += x[k] * w1[j];
      }
    }
    const int nvalid = p.Cout - co < 8 ? p.Cout - co : 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = acc[j] * p.scale[co + j] + p.bias[co + j];
      if (leaky) t = t > 0.f ? t : Y3_LEAKY_SLOPE * t;
      v[j] = t;
    }
    if (nvalid == 8 && (p.out_ld % 8) == 0) {
      if constexpr (sizeof(TO) == 2) {
        *reinterpret_cast<u32x4 *>(orow + co) = y3_pack8<TO>(v);
      } else {
        *reinterpret_cast<f32x4 *>(orow + co) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4 *>(orow + co + 4) = f32x4{v[4], v[5], v[6], v[7]};
      }
    } else {
      for (int j = 0; j < nvalid; ++j) orow[co + j] = y3_from_float<TO>(v[j]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// MFMA stem for the throughput path (uint8 BGR frames in, bf16 NHWC out, 3x3 stride 1 pad 1, Cout <= 32).
// K = 27 fits one v_mfma_f32_16x16x32_bf16 step.  A workgroup owns an 8 x 32 pixel tile: the (8+2) x (32+2)
// input halo is read once (byte loads), normalised (v / 255.0f, correctly rounded, then bf16) into LDS, and
// every wave builds the im2col fragments of its two pixel rows straight from that image: for one filter row the
// 9 values (kx, c) of a pixel are 9 consecutive bf16, so K is ordered k = ky*9 + kx*3 + c_mem and the host
// permutes the weights to match (c_mem = 2 - c_rgb: the BGR->RGB flip costs nothing).
template <typename T>
struct StemMfmaArgs {
  const uint8_t *in;       // (B, H, W, 3) uint8 BGR
  const T *wgt;       // [32][32] bf16: row = output channel, k as above, zero padded
  const float *scale;
  const float *bias;
  T *out;
  int B, H, W, Cout, out_ld, tiles_x, tiles_y;
  uint32_t flags;
};

constexpr int kStemTH = 8, kStemTW = 32;
constexpr int kStemRow = (kStemTW + 2) * 3 + 2;   // bf16 elements per halo row (102 used, padded to 104)

template <typename T>
__global__ __launch_bounds__(256) void conv_stem_mfma_kernel(StemMfmaArgs<T> p) {
  typedef typename H16<T>::v8 V8;
  __shared__ __attribute__((aligned(16))) T tile[(kStemTH + 2) * kStemRow];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int tx = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int y0 = ty * kStemTH, x0 = tx * kStemTW;

  // ---- stage the normalised halo: (TH+2) rows x (TW+2)*3 bytes ------------------------------------
  constexpr int ROWB = (kStemTW + 2) * 3;
  for (int i = tid; i < (kStemTH + 2) * ROWB; i += 256) {
    const int r = i / ROWB, cb = i - r * ROWB;
    const int iy = y0 - 1 + r;
    const int ixb = (x0 - 1) * 3 + cb;            // byte column inside the image row
    float v = 0.f;
    if ((unsigned)iy < (unsigned)p.H && ixb >= 0 && ixb < p.W * 3)
      v = (float)p.in[((long long)b * p.H + iy) * p.W * 3 + ixb] / 255.0f;
    tile[r * kStemRow + cb] = (T)v;
  }

  // ---- weights: two A fragments (channels 0-15, 16-31), lane (co = lane&15, q = lane>>4) holds k = 8q..8q+7
  const int fr = lane & 15, fq = lane >> 4;
  const V8 w0 = *reinterpret_cast<const V8 *>(p.wgt + (0 + fr) * 32 + fq * 8);
  const V8 w1 = *reinterpret_cast<const V8 *>(p.wgt + (16 + fr) * 32 + fq * 8);
  // per-lane LDS element offsets of k = 8q + j relative to the pixel's first byte in halo row `row`
  int koff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = fq * 8 + j;
    const int ky = k < 27 ? k / 9 : 0, jj = k < 27 ? k - ky * 9 : 0;   // padded k: any valid element (weight is 0)
    koff[j] = ky * kStemRow + jj;
  }
  const int cq = fq * 4;   // this lane's 4 output channels inside a 16-channel fragment
  f32x4 sc[2], bi[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    sc[ni] = *reinterpret_cast<const f32x4 *>(p.scale + ni * 16 + cq);
    bi[ni] = *reinterpret_cast<const f32x4 *>(p.bias + ni * 16 + cq);
  }
  const bool leaky = p.flags & Y3_F_LEAKY;
  __syncthreads();

  // ---- each wave: 2 pixel rows x 2 groups of 16 pixels ------------------------------------------------
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int row = wave * 2 + rr;
    const int oy = y0 + row;
#pragma unroll
    for (int gx = 0; gx < kStemTW / 16; ++gx) {
      const int px = gx * 16 + fr;
      const T *base = tile + row * kStemRow + px * 3;
      V8 xf;
#pragma unroll
      for (int j = 0; j < 8; ++j) xf[j] = base[koff[j]];
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      acc0 = H16<T>::mfma(w0, xf, acc0);
      acc1 = H16<T>::mfma(w1, xf, acc1);
      const int ox = x0 + px;
      if (oy < p.H && ox < p.W) {
        T *op = p.out + (((long long)b * p.H + oy) * p.W + ox) * p.out_ld;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const f32x4 a = ni ? acc1 : acc0;
          const int co = ni * 16 + cq;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float t = a[r] * sc[ni][r] + bi[ni][r];
            if (leaky) t = t > 0.f ? t : Y3_LEAKY_SLOPE * t;
            v[r] = t;
          }
          if (co + 4 <= p.Cout) {
            *reinterpret_cast<u32x2 *>(op + co) = y3_pack4<T>(v[0], v[1], v[2], v[3]);
          } else {
            for (int r = 0; r < 4; ++r)
              if (co + r < p.Cout) op[co + r] = (T)v[r];
          }
        }
      }
    }
  }
}

struct DirectArgs {
  const void *in;
  const void *wgt;  // [cout_pad][k_ld] element type T
  const float *scale;
  const float *bias;
  const void *res;
  void *out;
  int B, H, W, Cin, in_ld, Ho, Wo, Cout, out_ld, res_ld, ks, stride, pad, k_ld;
  long long total;
  uint32_t flags;
};

template <typename T>
__global__ __launch_bounds__(256) void conv_direct_kernel(DirectArgs p) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.total) return;
  const int co = (int)(idx % p.Cout);
  const long long m = idx / p.Cout;
  const int hw = p.Ho * p.Wo;
  const int b = (int)(m / hw);
  const int rem = (int)(m - (long long)b * hw);
  const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
  const T *w = static_cast<const T *>(p.wgt) + (long long)co * p.k_ld;
  float acc = 0.f;
  for (int ky = 0; ky < p.ks; ++ky) {
    const int iy = oy * p.stride - p.pad + ky;
    if ((unsigned)iy >= (unsigned)p.H) continue;
    for (int kx = 0; kx < p.ks; ++kx) {
      const int ix = ox * p.stride - p.pad + kx;
      if ((unsigned)ix >= (unsigned)p.W) continue;
      const T *wk = w + (ky * p.ks + kx) * p.Cin;
      if (p.flags & Y3_F_IN_NCHW_F32) {
        for (int c = 0; c < p.Cin; ++c)
          acc += static_cast<const float *>(p.in)[(((long long)b * p.Cin + c) * p.H + iy) * p.W + ix] *
                 y3_to_float<T>(wk[c]);
      } else if (p.flags & Y3_F_IN_NHWC_U8BGR) {
        for (int c = 0; c < p.Cin; ++c) {
          const uint8_t u = static_cast<const uint8_t *>(p.in)[(((long long)b * p.H + iy) * p.W + ix) * p.Cin + (p.Cin - 1 - c)];
          acc += ((float)u / 255.0f) * y3_to_float<T>(wk[c]);
        }
      } else {
        const T *xp = static_cast<const T *>(p.in) + (((long long)b * p.H + iy) * p.W + ix) * p.in_ld;
        for (int c = 0; c < p.Cin; ++c) acc += y3_to_float<T>(xp[c]) * y3_to_float<T>(wk[c]);
      }
    }
  }
  float v = acc * p.scale[co] + p.bias[co];
  if (p.flags & Y3_F_LEAKY) v = v > 0.f ? v : Y3_LEAKY_SLOPE * v;
  if (p.flags & Y3_F_RESIDUAL) v += y3_to_float<T>(static_cast<const T *>(p.res)[m * p.res_ld + co]);
  if ((p.flags & Y3_F_OUT_F32) || sizeof(T) == 4)
    static_cast<float *>(p.out)[m * p.out_ld + co] = v;
  else
    static_cast<T *>(p.out)[m * p.out_ld + co] = y3_from_float<T>(v);
}

}  // namespace

int y3_launch_conv_small(const y3_op &op, const void *d_in, hipStream_t s, const char **kernel_name,
                         bool dry_run) {
  Y3_REQUIRE(op.in_c == 3 && op.ksize == 3, "conv block %d: stem kernel needs Cin=3, 3x3", op.block_idx);
  Y3_REQUIRE(op.flags & (Y3_F_IN_NCHW_F32 | Y3_F_IN_NHWC_U8BGR),
             "conv block %d: stem kernel reads the network input only", op.block_idx);
  Y3_REQUIRE(op.cout_pad % 8 == 0 && op.cout_pad >= op.out_c && op.cout_pad <= 256,
             "conv block %d: bad cout_pad %d", op.block_idx, op.cout_pad);
  Y3_REQUIRE(!(op.flags & Y3_F_RESIDUAL), "conv block %d: stem kernel has no residual input", op.block_idx);
  SmallArgs a;
  a.in = d_in;
  a.wgt = static_cast<const float *>(op.d_weight);
  a.scale = op.d_scale;
  a.bias = op.d_bias;
  a.out = static_cast<char *>(op.d_out);
  a.B = op.batch; a.H = op.in_h; a.W = op.in_w; a.Ho = op.out_h; a.Wo = op.out_w;
  a.Cout = op.out_c; a.cout_pad = op.cout_pad; a.out_ld = op.out_ld;
  a.stride = op.stride; a.pad = op.pad;
  a.M = op.batch * op.out_h * op.out_w;
  a.flags = op.flags;
  const bool u8 = op.flags & Y3_F_IN_NHWC_U8BGR;
  const int odt = (op.flags & Y3_F_OUT_F32) ? Y3_F32 : op.dtype;   // element type of the output
  *kernel_name = u8 ? Y3_KNAME(odt, "conv_stem3x3_u8_", "") : Y3_KNAME(odt, "conv_stem3x3_nchw_", "");
  if (dry_run) return Y3_OK;
  const dim3 grid(y3_ceil_div(a.M, 256)), block(256);
  const size_t lds = ((size_t)27 * op.cout_pad + 256) * sizeof(float);
  return y3_by_dtype(odt, [&](auto tag) {
    if (u8) Y3_LAUNCH((conv_stem3x3_kernel<decltype(tag), 1>), grid, block, lds, s, a);
    else Y3_LAUNCH((conv_stem3x3_kernel<decltype(tag), 0>), grid, block, lds, s, a);
    Y3_HIP_CHECK(hipGetLastError());
    return Y3_OK;
  });
}

// MFMA stem: needs weights in its own layout (bf16 [32][32], see yolov3/darknet.py) -> separate conv path (3)
bool y3_conv_stem_mfma_supported(const y3_op &op) {
  return (op.flags & Y3_F_IN_NHWC_U8BGR) && !(op.flags & (Y3_F_OUT_F32 | Y3_F_RESIDUAL)) && y3_is16(op.dtype) &&
         op.in_c == 3 && op.ksize == 3 && op.stride == 1 && op.pad == 1 && op.out_c <= 32 && op.out_ld % 4 == 0;
}

int y3_launch_conv_stem_mfma(const y3_op &op, const void *d_in, hipStream_t s, const char **kernel_name,
                             bool dry_run) {
  Y3_REQUIRE(y3_conv_stem_mfma_supported(op), "conv block %d: not a shape for the MFMA stem", op.block_idx);
  *kernel_name = op.dtype == Y3_F16 ? "conv_stem_mfma_u8_f16" : "conv_stem_mfma_u8_bf16";
  if (dry_run) return Y3_OK;
  return y3_by_dtype16(op.dtype, [&](auto tag) {
    typedef decltype(tag) T;
    StemMfmaArgs<T> a;
    a.in = static_cast<const uint8_t *>(d_in);
    a.wgt = static_cast<const T *>(op.d_weight);
    a.scale = op.d_scale; a.bias = op.d_bias;
    a.out = static_cast<T *>(op.d_out);
    a.B = op.batch; a.H = op.in_h; a.W = op.in_w; a.Cout = op.out_c; a.out_ld = op.out_ld;
    a.tiles_x = y3_ceil_div(op.in_w, kStemTW);
    a.tiles_y = y3_ceil_div(op.in_h, kStemTH);
    a.flags = op.flags;
    Y3_LAUNCH(conv_stem_mfma_kernel<T>, dim3(a.tiles_x * a.tiles_y * op.batch), dim3(256), 0, s, a);
    Y3_HIP_CHECK(hipGetLastError());
    return Y3_OK;
  });
}

int y3_launch_conv_direct(const y3_op &op, const void *d_in, hipStream_t s, const char **kernel_name,
                          bool dry_run) {
  DirectArgs a;
  a.in = d_in; a.wgt = op.d_weight; a.scale = op.d_scale; a.bias = op.d_bias; a.res = op.d_res;
  a.out = op.d_out;
  a.B = op.batch; a.H = op.in_h; a.W = op.in_w; a.Cin = op.in_c; a.in_ld = op.in_ld;
  a.Ho = op.out_h; a.Wo = op.out_w; a.Cout = op.out_c; a.out_ld = op.out_ld; a.res_ld = op.res_ld;
  a.ks = op.ksize; a.stride = op.stride; a.pad = op.pad; a.k_ld = op.k_ld;
  a.total = (long long)op.batch * op.out_h * op.out_w * op.out_c;
  a.flags = op.flags;
  Y3_REQUIRE(op.k_ld >= op.ksize * op.ksize * op.in_c, "conv block %d: k_ld too small", op.block_idx);
  *kernel_name = Y3_KNAME(op.dtype, "conv_direct_", "");
  if (dry_run) return Y3_OK;
  const dim3 grid((unsigned)((a.total + 255) / 256)), block(256);
  return y3_by_dtype(op.dtype, [&](auto tag) {
    Y3_LAUNCH(conv_direct_kernel<decltype(tag)>, grid, block, 0, s, a);
    Y3_HIP_CHECK(hipGetLastError());
    return Y3_OK;
  });
}
