// YOLO detection-head decode for gfx950.
//
// Replaces YOLOLayer.forward (/root/reference/yolov3/darknet.py:48-122) plus the head
// concatenation and the w,h / net-size division of Darknet.forward (darknet.py:389-399):
// every head writes straight into its row range of the concatenated (B, M, .) outputs, so no
// concat or scaling pass exists.
//
// Input : float32 head-conv output, NHWC (B, h, w, ld), channel = anchor * n_attr + attr
//         (the reference reshapes (B, A*n_attr, h, w) -> (B, A, n_attr, h, w), darknet.py:68).
// Output: row = row_offset + a*h*w + y*w + x   (darknet.py:118-120, heads in cfg order)
//   bbox[row] = ((sigmoid(tx)+x)/w, (sigmoid(ty)+y)/h, exp(tw)*Aw/net_w, exp(th)*Ah/net_h)
//   prob[row] = max_c softmax(tc)_c * sigmoid(to)       cls[row] = argmax (first on ties), int64
// Built with -ffp-contract=off: each operation rounds like the reference's separate torch ops.
#include "common.h"
#include "decode_core.h"

namespace {

struct YoloArgs {
  const float *in;
  float *bbox;
  float *prob;
  long long *cls;
  int B, h, w, ld, n_anchor, n_attr, row_offset, rows_total;
  float net_w, net_h;
  float aw[8], ah[8];
  long long total;
};

__device__ __forceinline__ float sigmoidf_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

// One workgroup decodes kPix consecutive pixels (all anchors).  The pixels' channel vectors are
// contiguous in NHWC memory, so they are staged into LDS with fully coalesced 16-byte loads
// (one wave instruction per pixel row of 256 floats); each (pixel, anchor) box is then decoded
// by one thread out of LDS.  The LDS pixel stride is ld+1 floats so that the per-box reads
// (stride n_attr floats between threads) spread over the banks.
constexpr int kPix = 32;

// LANES threads per box.  LANES == 1: the reference's sequential class loop (float32 parity path).  LANES == 4
// (bf16 throughput mode): decode_core.h -- the class range is split over four adjacent lanes (maxima, exp-sums and the
// arg-max combined with quad DPP moves), which quadruples the threads working out of the same LDS tile (the staging
// tile, not registers, bounds the boxes in flight per CU); the class exponentials use the hardware exp2 and the exp-sum
// is a tree of four partial sums, i.e. the score is equal to ~1e-6 relative, the arg-max and the box are identical.
template <int LANES>
__global__ __launch_bounds__(LANES == 1 ? 256 : 384) void yolo_decode_kernel(YoloArgs p) {
  constexpr int NT = LANES == 1 ? 256 : 384;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lds_ld = p.ld + 1;
  const long long npix = (long long)p.B * p.h * p.w;
  const long long pix0 = (long long)blockIdx.x * kPix;
  const int chunks_per_pix = p.ld >> 2;
  const int nchunks = kPix * chunks_per_pix;
  for (int c = threadIdx.x; c < nchunks; c += NT) {
    const int pl = c / chunks_per_pix, c4 = c - pl * chunks_per_pix;
    if (pix0 + pl < npix) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(p.in + (pix0 + pl) * p.ld + c4 * 4);
      float *d = sm + pl * lds_ld + c4 * 4;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
  }
  __syncthreads();
  const int nbox = kPix * p.n_anchor;
  // (all lanes of a box group stay in the loop together: nbox * LANES is a multiple of LANES and the shuffles below
  // only pair lanes of one group)
  for (int t = threadIdx.x; t < nbox * LANES; t += NT) {
    const int box = t / LANES, sub = t - box * LANES;
    const int pl = box / p.n_anchor, a = box - pl * p.n_anchor;
    const long long pix = pix0 + pl;
    const bool live = pix < npix;
    const long long pixc = live ? pix : npix - 1;
    const int x = (int)(pixc % p.w);
    const int y = (int)((pixc / p.w) % p.h);
    const int b = (int)(pixc / ((long long)p.w * p.h));
    const float *t_ = sm + pl * lds_ld + a * p.n_attr;

    if constexpr (LANES == 4) {
      // bf16 networks: decode_core.h (shared with the fused head kernel)
      float comp, score;
      int best_c;
      y3_decode_box4(t_, p.n_attr, sub, (float)x, (float)y, (float)p.w, (float)p.h, p.aw[a], p.ah[a], p.net_w, p.net_h, comp,
                     score, best_c);
      if (!live) continue;
      const long long row = (long long)b * p.rows_total + p.row_offset + (long long)a * p.h * p.w + (long long)y * p.w + x;
      p.bbox[row * 4 + sub] = comp;
      if (sub == 2) {
        p.prob[row] = score;
        p.cls[row] = best_c;
      }
      continue;
    }
    const int ncls = p.n_attr - 5;
    float mx = -INFINITY;
    for (int c = 0; c < ncls; ++c) mx = fmaxf(mx, t_[5 + c]);
    float sum = 0.f, best = -1.f;
    int best_c = 0;
    for (int c = 0; c < ncls; ++c) {
      const float e = expf(t_[5 + c] - mx);
      sum += e;
      if (e > best) {  // strict: first index wins ties, like torch.max
        best = e;
        best_c = c;
      }
    }
    if (!live) continue;
    const float bx = (sigmoidf_ref(t_[0]) + (float)x) / (float)p.w;
    const float by = (sigmoidf_ref(t_[1]) + (float)y) / (float)p.h;
    const float bw = (expf(t_[2]) * p.aw[a]) / p.net_w;
    const float bh = (expf(t_[3]) * p.ah[a]) / p.net_h;
    const float obj = sigmoidf_ref(t_[4]);
    const float score = (best / sum) * obj;

    const long long row = (long long)b * p.rows_total + p.row_offset + (long long)a * p.h * p.w + (long long)y * p.w + x;
    *reinterpret_cast<f32x4 *>(p.bbox + row * 4) = f32x4{bx, by, bw, bh};
    p.prob[row] = score;
    p.cls[row] = best_c;
  }
}

}  // namespace


int y3_launch_yolo(const y3_op &op, const void *d_in, hipStream_t s, const char **kernel_name,
                   bool dry_run) {
  Y3_REQUIRE(op.n_anchor >= 1 && op.n_anchor <= 8, "yolo block %d: 1..8 anchors per head", op.block_idx);
  Y3_REQUIRE(op.n_attr > 5 && op.n_anchor * op.n_attr <= op.in_ld, "yolo block %d: bad attribute count", op.block_idx);
  Y3_REQUIRE(dry_run || (op.d_bbox && op.d_prob && op.d_cls), "yolo block %d: missing output pointers", op.block_idx);
  Y3_REQUIRE(op.in_ld % 4 == 0 && op.in_ld <= 1024, "yolo block %d: pixel stride must be a multiple of 4 floats (<= 1024)", op.block_idx);
  YoloArgs a;
  a.in = static_cast<const float *>(d_in);
  a.bbox = op.d_bbox;
  a.prob = op.d_prob;
  a.cls = reinterpret_cast<long long *>(op.d_cls);
  a.B = op.batch; a.h = op.in_h; a.w = op.in_w; a.ld = op.in_ld;
  a.n_anchor = op.n_anchor; a.n_attr = op.n_attr;
  a.row_offset = op.row_offset; a.rows_total = op.rows_total;
  a.net_w = op.net_w; a.net_h = op.net_h;
  for (int i = 0; i < 8; ++i) { a.aw[i] = op.anchor_w[i]; a.ah[i] = op.anchor_h[i]; }
  a.total = (long long)op.batch * op.in_h * op.in_w * op.n_anchor;
  *kernel_name = "yolo_decode_f32";
  if (dry_run) return Y3_OK;
  const long long npix = (long long)op.batch * op.in_h * op.in_w;
  const size_t lds = (size_t)kPix * (op.in_ld + 1) * sizeof(float);
  const dim3 grid((unsigned)((npix + kPix - 1) / kPix));
  // bf16 networks (throughput mode) take the four-lanes-per-box form; float32 networks keep the sequential class loop
  if (y3_is16(op.dtype) && y3_opt().decode_lanes != 1) Y3_LAUNCH(yolo_decode_kernel<4>, grid, dim3(384), lds, s, a);
  else Y3_LAUNCH(yolo_decode_kernel<1>, grid, dim3(256), lds, s, a);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}
