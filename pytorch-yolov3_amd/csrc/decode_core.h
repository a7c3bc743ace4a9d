// Four-lanes-per-box YOLO decode, shared by yolo_decode_kernel<4> (yolo_decode.hip) and the fused head kernel
// (conv_igemm.hip, DECODE = true): the bf16 networks' form of YOLOLayer.forward's per-box arithmetic
// (/root/reference/yolov3/darknet.py:86-116).  One definition, so the fused and the two-kernel paths give the same bits
// for the same logits.
//
// The four lanes of a box (adjacent lanes, sub = lane & 3) split the class range; maxima, exp-sums and the arg-max are
// combined with quad DPP moves.  Differences from the sequential float32 form (yolo_decode_kernel<1>, the parity path):
//   * the class exponentials are exp2(x * log2(e)) on the hardware instruction (v_exp_f32, ~1 ulp; the argument's
//     rounding adds |x| * 6e-8) instead of expf -- 80 of the 85 transcendentals of a box; the soft-max denominator is
//     then within ~1e-6 relative of the sequential loop's (tests/test_gpu_parity.py::
//     test_split_class_decode_matches_sequential_decode), far below the bf16 logits' own noise;
//   * the exp-sum is a tree of four partial sums;
//   * box coordinates, objectness and the final product keep expf and IEEE division: identical to the sequential form.
// The box tail is spread over the four lanes (sub 0: x, 1: y, 2: w + objectness, 3: h + best / sum) with selects
// instead of branches, so no lane idles while one lane does five exponentials and seven divisions.
#pragma once
#include "common.h"

__device__ __forceinline__ float y3_quad_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float y3_quad_xor2(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ int y3_quad_xor1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ int y3_quad_xor2(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true); }
// lane 3 of the quad -> every lane of the quad
__device__ __forceinline__ float y3_quad_from3(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xFF, 0xF, 0xF, true));
}

// t_: the box's n_attr logits (LDS).  Called by all four lanes of the box together (EXEC must hold whole quads).
// cell_*: grid column / row of the box's cell; grid_*: grid size; anchor_*: the box's anchor (pixels); net_*: the network
// input size the anchors refer to.  (Scalars by value: selecting between members of a struct in memory by lane turns
// into an indexed load from scratch memory.)
// comp: this lane's box component (sub 0: bx, 1: by, 2: bw, 3: bh); score / cls are valid on the lane with sub == 2.
__device__ __forceinline__ void y3_decode_box4(const float *t_, int n_attr, int sub, float cell_x, float cell_y, float grid_w,
                                               float grid_h, float anchor_w, float anchor_h, float net_w, float net_h,
                                               float &comp, float &score, int &cls) {
#pragma clang fp contract(off)
  constexpr float kLog2e = 1.44269504088896340736f;
  const int ncls = n_attr - 5;
  const int per = (ncls + 3) >> 2;
  const int c_lo = sub * per;
  float mx = -INFINITY, sum = 0.f, best = -1.f;
  int best_c = 0;
  if (ncls == 80) {
    // COCO heads: the lane's 20 logits stay in registers between the two passes
    float v[20];
#pragma unroll
    for (int c = 0; c < 20; ++c) v[c] = t_[5 + c_lo + c];
#pragma unroll
    for (int c = 0; c < 20; ++c) mx = fmaxf(mx, v[c]);
    mx = fmaxf(mx, y3_quad_xor1(mx));
    mx = fmaxf(mx, y3_quad_xor2(mx));
#pragma unroll
    for (int c = 0; c < 20; ++c) {
      const float e = __builtin_amdgcn_exp2f((v[c] - mx) * kLog2e);
      sum += e;
      if (e > best) {  // strict: first index wins ties, like torch.max
        best = e;
        best_c = c_lo + c;
      }
    }
  } else {
    const int c_hi = c_lo + per < ncls ? c_lo + per : ncls;
    for (int c = c_lo; c < c_hi; ++c) mx = fmaxf(mx, t_[5 + c]);
    mx = fmaxf(mx, y3_quad_xor1(mx));
    mx = fmaxf(mx, y3_quad_xor2(mx));
    for (int c = c_lo; c < c_hi; ++c) {
      const float e = __builtin_amdgcn_exp2f((t_[5 + c] - mx) * kLog2e);
      sum += e;
      if (e > best) {
        best = e;
        best_c = c;
      }
    }
  }
  {
    const float os = y3_quad_xor1(sum), ob = y3_quad_xor1(best);
    const int oc = y3_quad_xor1(best_c);
    sum += os;                                       // both partners add the same two numbers: same result
    if (ob > best || (ob == best && oc < best_c)) {  // lower class index wins ties
      best = ob;
      best_c = oc;
    }
  }
  {
    const float os = y3_quad_xor2(sum), ob = y3_quad_xor2(best);
    const int oc = y3_quad_xor2(best_c);
    sum += os;
    if (ob > best || (ob == best && oc < best_c)) {
      best = ob;
      best_c = oc;
    }
  }
  // box component of this lane: sub 0 / 1: (sigmoid(t) + cell) / grid;  sub 2 / 3: exp(t) * anchor / net
  const bool is_xy = (sub & 2) == 0, second = (sub & 1) != 0;
  const float tk = t_[sub];
  const float e = expf(is_xy ? -tk : tk);
  const float sig = 1.0f / (1.0f + e);
  const float cell = second ? cell_y : cell_x;
  const float anchor = second ? anchor_h : anchor_w;
  const float grid = second ? grid_h : grid_w;
  const float net = second ? net_h : net_w;
  const float num = is_xy ? sig + cell : e * anchor;
  const float den = is_xy ? grid : net;
  comp = num / den;
  // sub 2: objectness = sigmoid(t4);  sub 3: best / sum;  score = (best / sum) * objectness on sub 2
  const float e4 = expf(-t_[4]);
  const float r = (sub == 3 ? best : 1.0f) / (sub == 3 ? sum : 1.0f + e4);
  const float q = y3_quad_from3(r);
  score = q * r;
  cls = best_c;
}

// YOLO decode of `rows` pixels whose float32 logits (conv sums with scale / bias already applied) are parked in LDS with
// a row stride of LD floats (LD odd in units of banks: the per-box reads of adjacent lanes spread over the banks): four
// lanes per box, y3_decode_box4 above -- the same code as yolo_decode_kernel<4>.  Args: the launch arguments of a fused head kernel
// (conv_igemm.hip: IgemmArgs, conv_1x1.hip: DwArgs) -- M, HoWo, Wo, Ho, the fast-division constants and the y_* fields of the decode.
template <int NT, int LD, typename Args>
__device__ __forceinline__ void y3_head_decode_rows(const Args &p, const float *sL, int mbase, int rows, int tid) {
  const int nbox = rows * p.y_anchors;
  const uint32_t inv_a = (65536u + (uint32_t)p.y_anchors - 1u) / (uint32_t)p.y_anchors;   // box / anchors for box < 8192
  for (int t = tid; t < nbox * 4; t += NT) {
    const int box = t >> 2, sub = t & 3;
    const int pl = (int)(((uint32_t)box * inv_a) >> 16), a = box - pl * p.y_anchors;
    const long long m = (long long)mbase + pl;
    const bool live = m < p.M;
    const uint32_t um = (uint32_t)(live ? m : p.M - 1);
    const uint32_t b = (__umulhi(um, p.mul_hw) + um) >> p.sh_hw;
    const uint32_t rem = um - b * (uint32_t)p.HoWo;
    const uint32_t y = (__umulhi(rem, p.mul_w) + rem) >> p.sh_w;
    const uint32_t x = rem - y * (uint32_t)p.Wo;
    float comp, score;
    int best_c;
    y3_decode_box4(sL + pl * LD + a * p.y_attr, p.y_attr, sub, (float)x, (float)y, (float)p.Wo, (float)p.Ho, p.y_aw[a],
                   p.y_ah[a], p.y_net_w, p.y_net_h, comp, score, best_c);
    if (!live) continue;
    const long long row = (long long)b * p.y_rows_total + p.y_row_offset + (long long)a * p.HoWo + (long long)y * p.Wo + x;
    p.y_bbox[row * 4 + sub] = comp;
    if (sub == 2) {
      p.y_prob[row] = score;
      p.y_cls[row] = best_c;
    }
  }
}
