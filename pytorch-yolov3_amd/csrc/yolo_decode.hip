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

__global__ __launch_bounds__(256) void yolo_decode_kernel(YoloArgs p) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.total) return;
  const int a = (int)(idx % p.n_anchor);
  long long cell = idx / p.n_anchor;
  const int x = (int)(cell % p.w);
  cell /= p.w;
  const int y = (int)(cell % p.h);
  const int b = (int)(cell / p.h);
  const float *t = p.in + (((long long)b * p.h + y) * p.w + x) * p.ld + a * p.n_attr;

  const float bx = (sigmoidf_ref(t[0]) + (float)x) / (float)p.w;
  const float by = (sigmoidf_ref(t[1]) + (float)y) / (float)p.h;
  const float bw = (expf(t[2]) * p.aw[a]) / p.net_w;
  const float bh = (expf(t[3]) * p.ah[a]) / p.net_h;
  const float obj = sigmoidf_ref(t[4]);

  const int ncls = p.n_attr - 5;
  float mx = -INFINITY;
  for (int c = 0; c < ncls; ++c) mx = fmaxf(mx, t[5 + c]);
  float sum = 0.f, best = -1.f;
  int best_c = 0;
  for (int c = 0; c < ncls; ++c) {
    const float e = expf(t[5 + c] - mx);
    sum += e;
    if (e > best) {  // strict: first index wins ties, like torch.max
      best = e;
      best_c = c;
    }
  }
  const float score = (best / sum) * obj;

  const long long row = (long long)b * p.rows_total + p.row_offset + (long long)a * p.h * p.w + (long long)y * p.w + x;
  *reinterpret_cast<f32x4 *>(p.bbox + row * 4) = f32x4{bx, by, bw, bh};
  p.prob[row] = score;
  p.cls[row] = best_c;
}

}  // namespace

int y3_launch_yolo(const y3_op &op, const void *d_in, hipStream_t s, const char **kernel_name,
                   bool dry_run) {
  Y3_REQUIRE(op.n_anchor >= 1 && op.n_anchor <= 8, "yolo block %d: 1..8 anchors per head", op.block_idx);
  Y3_REQUIRE(op.n_attr > 5 && op.n_anchor * op.n_attr <= op.in_ld, "yolo block %d: bad attribute count", op.block_idx);
  Y3_REQUIRE(op.d_bbox && op.d_prob && op.d_cls, "yolo block %d: missing output pointers", op.block_idx);
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
  hipLaunchKernelGGL(yolo_decode_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, s, a);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}
