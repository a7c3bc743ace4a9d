// Detection tail on device: score threshold + compaction, pixel scaling, integer truncation,
// centre/size -> corners, per-class greedy NMS, ordered gather.  One 1024-thread workgroup per
// frame runs all phases back to back in ONE launch (16 wavefronts; phases separated by
// workgroup barriers), so a batch costs a single kernel and only kept detections ever leave
// the GPU.
//
// Replaces /root/reference/yolov3/inference.py:342-366 (mask = prob >= thr; x,w *= orig_w;
// y,h *= orig_h; astype(int); cxywh_to_tlbr :269-283; non_max_suppression :220-266 with
// _non_max_suppression :161-217).  Semantics kept bit-for-bit:
//   * threshold compare and pixel scaling in float32, truncation toward zero to int64,
//     corners = centre -/+ (size // 2) in integers;
//   * areas / intersections with the "+1" pixel convention in int64, IoU = inter / union in
//     float64, suppress iff IoU > thr (strict), candidates visited in descending score;
//   * classes are independent.
// Output order is canonical: class ascending, score descending, then higher row first (the
// reference's order is Python-set / argsort dependent; tests compare as sets).
//
// Wavefront primitives: __ballot + popcount prefix sums for the ordered compactions; the NMS
// inner loop keeps one box per lane (64 boxes per wavefront chunk), broadcasts the current
// survivor with ds_bpermute shuffles and clears victims with a ballot mask.
// Built with -ffp-contract=off.
#include "common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kIt = 8;          // 1024-row chunks handled per compaction pass (kIt * kWaves == 128)
// elements sorted in LDS; larger frames sort in global memory.  (8192 -- 96 KiB, the heavy regime of bench.py has 4762 candidates per
// frame -- was tried in round 6: the detection tail of that regime stayed at 0.79 ms per batch, so the sort is not what it spends
// its time on, and a 100-KiB workgroup can no longer share a CU with a 64-KiB implicit-GEMM workgroup of another stream: kept at 4096.)
constexpr int kLdsSort = 4096;
constexpr int kMaxFlags = 1024;  // chunk flags in LDS; classes beyond that run on one wave

struct DetectArgs {
  // forward-output mode
  const float *bbox;
  const float *prob;
  const long long *cls;
  const int *orig_hw;
  // caller-boxes mode (y3_nms)
  const long long *in_tlbr;
  const float *in_prob;
  const long long *in_cls;
  int n_in;
  int rows;      // predictions per frame / capacity
  int rows_p2;   // next power of two >= rows
  float prob_thresh;
  double iou_thresh;
  // workspace, per frame
  long long *c_box;             // [rows][4]
  float *c_prob;                // [rows]
  int *c_cls;                   // [rows]
  int *c_row;                   // [rows]
  unsigned long long *s_key;    // [rows_p2]
  unsigned int *s_pos;          // [rows_p2]
  unsigned char *keep;          // [rows]
  int *seg;                     // [rows]
  // outputs
  int *det_count;
  long long *det_tlbr;
  float *det_prob;
  long long *det_cls;
  int *det_row;
  long long *keep_idx;
};

__device__ __forceinline__ unsigned int score_desc_bits(float p) {
  unsigned int u = __float_as_uint(p);
  u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;  // ascending-sortable
  return ~u;                                   // descending
}

// (key asc, pos desc)
__device__ __forceinline__ bool elem_less(unsigned long long ka, unsigned int pa, unsigned long long kb,
                                          unsigned int pb) {
  return ka < kb || (ka == kb && pa > pb);
}

__device__ __forceinline__ long long shfl_ll(long long v, int src) {
  int lo = (int)(v & 0xFFFFFFFFll), hi = (int)(v >> 32);
  lo = __shfl(lo, src, 64);
  hi = __shfl(hi, src, 64);
  return ((long long)hi << 32) | (unsigned int)lo;
}

// value of lane k (wave-uniform k) as a scalar: v_readlane_b32, no LDS traffic
__device__ __forceinline__ long long readlane_ll(long long v, int k) {
  const int lo = __builtin_amdgcn_readlane((int)(v & 0xFFFFFFFFll), k);
  const int hi = __builtin_amdgcn_readlane((int)(v >> 32), k);
  return ((long long)hi << 32) | (unsigned int)lo;
}

// Fast form of the suppression test for boxes whose coordinates all lie within +-16000 (widths <= 32001, areas
// <= 1.03e9, unions < 2^31: exact in 32-bit integers).  fl(inter / uni) > thr (float64 true division,
// inference.py:211-215) is decided without dividing: d = inter - thr * uni (one fma rounding) has the exact sign of
// inter / uni - thr, and the rounded quotient can disagree with that sign only inside (thr, thr + ulp(thr) / 2];
// pairs that close (d <= uni * |thr| * 2.3e-16), or with a non-positive union (degenerate boxes: the reference's
// inf / nan / negative quotient decides), take the division.  thr_m = |thr| * 2.3e-16.
__device__ __forceinline__ bool iou_exceeds_i32(int ax1, int ay1, int ax2, int ay2, int aarea, int x1, int y1, int x2,
                                                int y2, int area, double thr, double thr_m) {
  int iw = (ax2 < x2 ? ax2 : x2) - (ax1 > x1 ? ax1 : x1) + 1;
  int ih = (ay2 < y2 ? ay2 : y2) - (ay1 > y1 ? ay1 : y1) + 1;
  iw = iw > 0 ? iw : 0;
  ih = ih > 0 ? ih : 0;
  const int inter = iw * ih;
  const int uni = aarea + area - inter;
  const double a = (double)inter, u = (double)uni;
  const double d = fma(-thr, u, a);
  bool hit = d > u * thr_m;
  const bool unsure = !(uni > 0) || (d > 0.0 && !hit);
  if (__ballot(unsure) != 0ull) {
    if (unsure) hit = a / u > thr;
  }
  return hit;
}

__device__ __forceinline__ bool box_fits_i32(long long x1, long long y1, long long x2, long long y2) {
  const long long lim = 16000;
  return x1 > -lim && x1 < lim && y1 > -lim && y1 < lim && x2 > -lim && x2 < lim && y2 > -lim && y2 < lim;
}

// block-wide ordered compaction step: returns this thread's output slot (valid when flag) and
// adds the step's total to `running`.  Must be called by all threads.
__device__ __forceinline__ int ordered_slot(bool flag, int &running, int *wave_tot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long ballot = __ballot(flag);
  const int prefix = __popcll(ballot & ((1ull << lane) - 1ull));
  if (lane == 0) wave_tot[wave] = __popcll(ballot);
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    const int t = wave_tot[w];
    before += w < wave ? t : 0;
    total += t;
  }
  const int slot = running + before + prefix;
  running += total;
  __syncthreads();
  return slot;
}

template <bool NMS_MODE>
__global__ __launch_bounds__(kThreads) void detect_kernel(DetectArgs p) {
  __shared__ unsigned long long skey[kLdsSort];
  __shared__ unsigned int spos[kLdsSort];
  __shared__ int wave_tot[kWaves];
  __shared__ int chunk_tot[kIt * kWaves];
  __shared__ int nitem_sh, nflag_sh;
  __shared__ int flags_sh[kMaxFlags];   // "chunk is final" flags of classes that span several chunks

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long R = p.rows;

  const long long *c_box = NMS_MODE ? p.in_tlbr : p.c_box + (long long)b * R * 4;
  const float *c_prob = NMS_MODE ? p.in_prob : p.c_prob + (long long)b * R;
  unsigned long long *s_key = p.s_key + (long long)b * p.rows_p2;
  unsigned int *s_pos = p.s_pos + (long long)b * p.rows_p2;
  unsigned char *keep = p.keep + (long long)b * R;
  int *seg = p.seg + (long long)b * (R + R / 64 + 2) * 4;

  // ---- phase 1: threshold + ordered compaction + scale / truncate / corners -----------------
  int n = 0;
  if constexpr (!NMS_MODE) {
    long long *w_box = p.c_box + (long long)b * R * 4;
    float *w_prob = p.c_prob + (long long)b * R;
    int *w_cls = p.c_cls + (long long)b * R;
    int *w_row = p.c_row + (long long)b * R;
    const float oh = (float)p.orig_hw[b * 2 + 0], ow = (float)p.orig_hw[b * 2 + 1];
    // kIt row-chunks per pass: the kIt score loads of a thread fly together (one memory latency per pass instead
    // of one per chunk), and the ordered slots of all kIt x 16 wave-chunks come from ONE workgroup barrier plus a
    // wave-level scan of the 128 ballot counts (every wave scans them redundantly: no second barrier to publish).
    for (int base = 0; base < p.rows; base += kIt * kThreads) {
      float pr[kIt];
#pragma unroll
      for (int it = 0; it < kIt; ++it) {
        const int r = base + it * kThreads + tid;
        pr[it] = r < p.rows ? p.prob[(long long)b * R + r] : 0.f;
      }
      int pre[kIt];
      bool fl[kIt];
#pragma unroll
      for (int it = 0; it < kIt; ++it) {
        const int r = base + it * kThreads + tid;
        fl[it] = r < p.rows && pr[it] >= p.prob_thresh;  // float32 compare (inference.py:342)
        const unsigned long long ballot = __ballot(fl[it]);
        pre[it] = __popcll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) chunk_tot[it * kWaves + wave] = __popcll(ballot);
      }
      __syncthreads();
      const int t0 = chunk_tot[2 * lane], t1 = chunk_tot[2 * lane + 1];   // kIt * kWaves == 128 == 2 per lane
      int incl = t0 + t1;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
      }
      const int excl = incl - (t0 + t1);
      const int total = __shfl(incl, 63, 64);
#pragma unroll
      for (int it = 0; it < kIt; ++it) {
        const int e = it * kWaves + wave;                       // this thread's chunk in (pass, wave) order
        const int off = __shfl(excl, e >> 1, 64) + ((e & 1) ? __shfl(t0, e >> 1, 64) : 0);
        if (fl[it]) {
          const int r = base + it * kThreads + tid;
          const int slot = n + off + pre[it];
          const f32x4 bb = *reinterpret_cast<const f32x4 *>(p.bbox + ((long long)b * R + r) * 4);
          // float32 products, then truncation toward zero (inference.py:351-353)
          const long long cx = (long long)(bb[0] * ow), cy = (long long)(bb[1] * oh);
          const long long bw = (long long)(bb[2] * ow), bh = (long long)(bb[3] * oh);
          const long long hw = bw >> 1, hh = bh >> 1;  // floor division by 2 (inference.py:281-282)
          long long *o = w_box + (long long)slot * 4;
          o[0] = cx - hw; o[1] = cy - hh; o[2] = cx + hw; o[3] = cy + hh;
          w_prob[slot] = pr[it];
          w_cls[slot] = (int)p.cls[(long long)b * R + r];
          w_row[slot] = r;
        }
      }
      n += total;
      __syncthreads();   // chunk_tot is rewritten by the next pass
    }
    __syncthreads();
  } else {
    n = p.n_in;
  }
  if (n == 0) {
    if (tid == 0) p.det_count[b] = 0;
    return;
  }

  // ---- phase 2: sort by (class asc, score desc, position desc) ------------------------------
  int P = 2;
  while (P < n) P <<= 1;
  const bool lds_sort = P <= kLdsSort;
  auto make_key = [&](int i) -> unsigned long long {
    int c;
    if constexpr (NMS_MODE) c = p.in_cls ? (int)p.in_cls[i] : 0;
    else c = p.c_cls[(long long)b * R + i];
    const unsigned int cu = (unsigned int)c ^ 0x80000000u;  // signed order
    return ((unsigned long long)cu << 32) | score_desc_bits(c_prob[i]);
  };
  if (lds_sort) {
    for (int i = tid; i < P; i += kThreads) {
      skey[i] = i < n ? make_key(i) : ~0ull;
      spos[i] = i < n ? (unsigned int)i : 0u;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < P; i += kThreads) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const unsigned long long ka = skey[i], kb = skey[ixj];
            const unsigned int pa = spos[i], pb = spos[ixj];
            const bool up = (i & k) == 0;
            const bool sw = up ? elem_less(kb, pb, ka, pa) : elem_less(ka, pa, kb, pb);
            if (sw) {
              skey[i] = kb; skey[ixj] = ka;
              spos[i] = pb; spos[ixj] = pa;
            }
          }
        }
        __syncthreads();
      }
    for (int i = tid; i < n; i += kThreads) {
      s_key[i] = skey[i];
      s_pos[i] = spos[i];
    }
  } else {
    for (int i = tid; i < P; i += kThreads) {
      s_key[i] = i < n ? make_key(i) : ~0ull;
      s_pos[i] = i < n ? (unsigned int)i : 0u;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < P; i += kThreads) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const unsigned long long ka = s_key[i], kb = s_key[ixj];
            const unsigned int pa = s_pos[i], pb = s_pos[ixj];
            const bool up = (i & k) == 0;
            const bool sw = up ? elem_less(kb, pb, ka, pa) : elem_less(ka, pa, kb, pb);
            if (sw) {
              s_key[i] = kb; s_key[ixj] = ka;
              s_pos[i] = pb; s_pos[ixj] = pa;
            }
          }
        }
        __syncthreads();
      }
  }
  if (tid == 0) { nitem_sh = 0; nflag_sh = 0; }
  for (int i = tid; i < kMaxFlags; i += kThreads) flags_sh[i] = 0;
  __syncthreads();

  // ---- phase 3: class segments -> work items -----------------------------------------------------
  // A class of nc candidates is nch = ceil(nc / 64) chunks in score order.  Chunk j needs the survivors of chunks
  // 0..j-1, so chunks of one class form a dependency chain; the chain's links are handed to different wavefronts
  // (item index mod 16) and synchronise through per-chunk "final" flags in LDS: while one wave runs chunk j's
  // in-chunk greedy pass, the others are already suppressing their own chunks with the survivors of chunks < j.
  // (A frame whose candidates nearly all share one class -- the bench regime -- was serial on ONE wave before.)
  // item = {segment start, segment end, chunk j or -1 = whole segment on one wave, first flag of the segment}
  int *items = seg;
  for (int i = tid; i < n; i += kThreads) {
    const unsigned int c = (unsigned int)(s_key[i] >> 32);
    if (i == 0 || (unsigned int)(s_key[i - 1] >> 32) != c) {
      int lo = i + 1, hi = n;  // first index in (i, n] whose class differs
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((unsigned int)(s_key[mid] >> 32) == c) lo = mid + 1; else hi = mid;
      }
      const int nch = (lo - i + 63) >> 6;
      int fb = -1;
      if (nch > 1) {
        fb = atomicAdd(&nflag_sh, nch);
        if (fb + nch > kMaxFlags) fb = -1;            // out of flags: this class runs on one wave (as before)
      }
      const int cnt = fb >= 0 ? nch : 1;
      const int i0 = atomicAdd(&nitem_sh, cnt);
      for (int j = 0; j < cnt; ++j) {
        items[(i0 + j) * 4 + 0] = i;
        items[(i0 + j) * 4 + 1] = lo;
        items[(i0 + j) * 4 + 2] = fb >= 0 ? j : -1;
        items[(i0 + j) * 4 + 3] = fb;
      }
    }
  }
  __syncthreads();
  const int nitem = nitem_sh;

  // ---- phase 4: greedy NMS --------------------------------------------------------------------------
  const double thr_m = fabs(p.iou_thresh) * 2.3e-16;
  // one 64-candidate chunk of a class: suppress by the survivors of the earlier chunks, then greedy inside
  auto do_chunk = [&](int start, int end, int j, int fb) {
      const int c0 = start + 64 * j;
      const int idx = c0 + lane;
      const bool valid = idx < end;
      long long x1 = 0, y1 = 0, x2 = 0, y2 = 0;
      if (valid) {
        const long long *bp = c_box + (long long)s_pos[idx] * 4;
        x1 = bp[0]; y1 = bp[1]; x2 = bp[2]; y2 = bp[3];
      }
      const long long area = (x2 - x1 + 1) * (y2 - y1 + 1);
      // 32-bit fast path when every box of this chunk (and of the earlier chunk it is compared with) is small
      const bool fits_all = __ballot(box_fits_i32(x1, y1, x2, y2)) == ~0ull;
      const int ix1 = (int)x1, iy1 = (int)y1, ix2 = (int)x2, iy2 = (int)y2;
      const int iarea = (ix2 - ix1 + 1) * (iy2 - iy1 + 1);
      bool dead = !valid;
      // survivors of earlier chunks of this class
      for (int jj = 0; jj < j; ++jj) {
        if (fb >= 0) {
          while (__hip_atomic_load(&flags_sh[fb + jj], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0)
            __builtin_amdgcn_s_sleep(1);
        }
        const int p0 = start + 64 * jj;
        unsigned long long kept = __ballot(keep[p0 + lane] != 0);
        if (kept == 0ull) continue;
        const long long *qp = c_box + (long long)s_pos[p0 + lane] * 4;
        const long long qx1 = qp[0], qy1 = qp[1], qx2 = qp[2], qy2 = qp[3];
        if (fits_all && __ballot(box_fits_i32(qx1, qy1, qx2, qy2)) == ~0ull) {
          const int pqx1 = (int)qx1, pqy1 = (int)qy1, pqx2 = (int)qx2, pqy2 = (int)qy2;
          const int pqarea = (pqx2 - pqx1 + 1) * (pqy2 - pqy1 + 1);
          while (kept) {
            const int k = __ffsll((long long)kept) - 1;
            kept &= kept - 1ull;
            dead = dead || iou_exceeds_i32(__builtin_amdgcn_readlane(pqx1, k), __builtin_amdgcn_readlane(pqy1, k),
                                           __builtin_amdgcn_readlane(pqx2, k), __builtin_amdgcn_readlane(pqy2, k),
                                           __builtin_amdgcn_readlane(pqarea, k), ix1, iy1, ix2, iy2, iarea,
                                           p.iou_thresh, thr_m);
          }
          continue;
        }
        while (kept) {
          const int k = __ffsll((long long)kept) - 1;
          kept &= kept - 1ull;
          const long long ax1 = shfl_ll(qx1, k), ay1 = shfl_ll(qy1, k);
          const long long ax2 = shfl_ll(qx2, k), ay2 = shfl_ll(qy2, k);
          const long long aarea = (ax2 - ax1 + 1) * (ay2 - ay1 + 1);
          long long iw = (ax2 < x2 ? ax2 : x2) - (ax1 > x1 ? ax1 : x1) + 1;
          long long ih = (ay2 < y2 ? ay2 : y2) - (ay1 > y1 ? ay1 : y1) + 1;
          iw = iw > 0 ? iw : 0;
          ih = ih > 0 ? ih : 0;
          const long long inter = iw * ih;
          const double iou = (double)inter / (double)(aarea + area - inter);
          dead = dead || (iou > p.iou_thresh);
        }
      }
      // greedy inside the chunk, in score order (lane order)
      unsigned long long alive = __ballot(!dead);
      if (fits_all) {
        for (int k = 0; k < 64; ++k) {
          if (!((alive >> k) & 1ull)) continue;  // wave-uniform
          const bool hit = iou_exceeds_i32(__builtin_amdgcn_readlane(ix1, k), __builtin_amdgcn_readlane(iy1, k),
                                           __builtin_amdgcn_readlane(ix2, k), __builtin_amdgcn_readlane(iy2, k),
                                           __builtin_amdgcn_readlane(iarea, k), ix1, iy1, ix2, iy2, iarea,
                                           p.iou_thresh, thr_m);
          alive &= ~__ballot(lane > k && hit);
        }
      } else {
        for (int k = 0; k < 64; ++k) {
          if (!((alive >> k) & 1ull)) continue;  // wave-uniform
          const long long ax1 = shfl_ll(x1, k), ay1 = shfl_ll(y1, k);
          const long long ax2 = shfl_ll(x2, k), ay2 = shfl_ll(y2, k);
          const long long aarea = (ax2 - ax1 + 1) * (ay2 - ay1 + 1);
          long long iw = (ax2 < x2 ? ax2 : x2) - (ax1 > x1 ? ax1 : x1) + 1;
          long long ih = (ay2 < y2 ? ay2 : y2) - (ay1 > y1 ? ay1 : y1) + 1;
          iw = iw > 0 ? iw : 0;
          ih = ih > 0 ? ih : 0;
          const long long inter = iw * ih;
          const double iou = (double)inter / (double)(aarea + area - inter);
          const bool hit = lane > k && (iou > p.iou_thresh);
          alive &= ~__ballot(hit);
        }
      }
      if (valid) keep[idx] = (alive >> lane) & 1ull ? 1 : 0;
      // publish: the flags of this chunk's survivors first, then (release) the chunk's "final" flag -- every lane
      // stores the same word (no lane-dependent branch inside a loop that uses cross-lane reads)
      if (fb >= 0) __hip_atomic_store(&flags_sh[fb + j], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __threadfence_block();  // later chunks of this wavefront read these flags
  };
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  for (int it = wave_u; it < nitem; it += kWaves) {
    const int start = __builtin_amdgcn_readfirstlane(items[it * 4 + 0]);
    const int end = __builtin_amdgcn_readfirstlane(items[it * 4 + 1]);
    const int j = __builtin_amdgcn_readfirstlane(items[it * 4 + 2]);
    const int fb = __builtin_amdgcn_readfirstlane(items[it * 4 + 3]);
    if (j >= 0) {
      do_chunk(start, end, j, fb);
    } else {
      const int nch = (end - start + 63) >> 6;
      for (int jj = 0; jj < nch; ++jj) do_chunk(start, end, jj, -1);
    }
  }
  __syncthreads();

  // ---- phase 5: ordered gather of survivors ---------------------------------------------------
  int kept_n = 0;
  for (int base = 0; base < n; base += kThreads) {
    const int i = base + tid;
    const bool flag = i < n && keep[i] != 0;
    const int slot = ordered_slot(flag, kept_n, wave_tot);
    if (flag) {
      const unsigned int pos = s_pos[i];
      if constexpr (NMS_MODE) {
        p.keep_idx[slot] = (long long)pos;
      } else {
        const long long *bp = c_box + (long long)pos * 4;
        long long *o = p.det_tlbr + ((long long)b * R + slot) * 4;
        o[0] = bp[0]; o[1] = bp[1]; o[2] = bp[2]; o[3] = bp[3];
        p.det_prob[(long long)b * R + slot] = c_prob[pos];
        p.det_cls[(long long)b * R + slot] = (long long)p.c_cls[(long long)b * R + pos];
        p.det_row[(long long)b * R + slot] = p.c_row[(long long)b * R + pos];
      }
    }
  }
  if (tid == 0) p.det_count[b] = kept_n;
}

__global__ void cxywh_to_tlbr_kernel(const long long *in, long long *out, int n, int cols) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long *r = in + (long long)i * cols;
  long long *o = out + (long long)i * cols;
  const long long cx = r[0], cy = r[1], hw = r[2] >> 1, hh = r[3] >> 1;  // floor(w/2), floor(h/2)
  o[0] = cx - hw; o[1] = cy - hh; o[2] = cx + hw; o[3] = cy + hh;
  for (int c = 4; c < cols; ++c) o[c] = r[c];
}

// ------------------------------------------------------------------------------------------------
// non_max_suppression / cxywh_to_tlbr on FLOATING-POINT boxes (round 5): the reference's public functions take any numeric
// dtype (/root/reference/yolov3/inference.py:161-217, :269-283); its own inference() only ever passes integers (:353-355),
// which is what detect_kernel above is built for.  For float32 / float64 boxes numpy computes every step in the ARRAY's
// dtype -- area = ((x2 - x1) + 1) * ((y2 - y1) + 1), w = maximum(0, (min(x2) - max(x1)) + 1), inter = w * h,
// union = (area_i + area_j) - inter, iou = inter / union, removed iff iou > thr with the threshold taken to that dtype
// (a weak Python scalar) -- and so does this kernel, operation by operation (-ffp-contract=off).  Not a hot path: one
// workgroup, bitonic sort of (class, score desc, index desc) in global memory, then one wavefront per class walks its
// segment greedily, 64 later boxes per pass.  Scores arrive as float64 (exact for float32 / float64 inputs).
template <typename F>
__device__ __forceinline__ F np_maximum0(F x) { return (x != x) ? x : (x > (F)0 ? x : (F)0); }   // numpy.maximum(0, x): NaN propagates

__device__ __forceinline__ bool nmsf_before(int ca, double pa, int ia, int cb, double pb, int ib) {
  if (ca != cb) return ca < cb;
  if (pa != pb) return pa > pb;
  return ia > ib;
}

template <typename F>
__global__ __launch_bounds__(kThreads) void nms_float_kernel(const F *box, const double *prob, const long long *cls, int n, int np2,
                                                             double thr_d, int *order, volatile unsigned char *alive, long long *keep,
                                                             int *keep_count) {
  // (`alive` is volatile: a lane clears flags that other lanes of the same wave read in the next pass -- the accesses must
  // reach memory in program order and not be served from a stale L1 line)
  __shared__ int s_wave_tot[kWaves];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const F thr = (F)thr_d;
  for (int i = tid; i < np2; i += kThreads) order[i] = i < n ? i : -1;
  for (int i = tid; i < n; i += kThreads) alive[i] = 1;
  __syncthreads();
  // bitonic sort of `order` (padding entries -1 sort last)
  auto before = [&](int a, int b) {
    if (a < 0 || b < 0) return b < 0 && a >= 0;
    return nmsf_before(cls ? (int)cls[a] : 0, prob[a], a, cls ? (int)cls[b] : 0, prob[b], b);
  };
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < np2; i += kThreads) {
        const int l = i ^ j;
        if (l > i) {
          const int a = order[i], b = order[l];
          const bool up = (i & k) == 0;
          if (up ? before(b, a) : before(a, b)) { order[i] = b; order[l] = a; }
        }
      }
      __syncthreads();
    }
  // class segments, found by every wave on its own while it walks (all lanes read the same address: one request per wave);
  // wave w owns segments w, w + 16, ...
  int seg_index = -1;
  for (int s = 0; s < n;) {
    // find the end of the segment that starts at s (uniform across the workgroup: all threads scan alike)
    const int c0 = cls ? (int)cls[order[s]] : 0;
    int e = s + 1;
    while (e < n && (cls ? (int)cls[order[e]] : 0) == c0) ++e;
    ++seg_index;
    if ((seg_index & (kWaves - 1)) == wave) {
      for (int i = s; i < e; ++i) {
        const int bi = order[i];
        if (!alive[bi]) continue;                       // (written by this wave only, earlier in program order)
        const F x1 = box[4 * bi], y1 = box[4 * bi + 1], x2 = box[4 * bi + 2], y2 = box[4 * bi + 3];
        const F area_i = ((x2 - x1) + (F)1) * ((y2 - y1) + (F)1);
        for (int j0 = i + 1; j0 < e; j0 += 64) {
          const int j = j0 + lane;
          if (j < e) {
            const int bj = order[j];
            if (alive[bj]) {
              const F u1 = box[4 * bj], v1 = box[4 * bj + 1], u2 = box[4 * bj + 2], v2 = box[4 * bj + 3];
              const F area_j = ((u2 - u1) + (F)1) * ((v2 - v1) + (F)1);
              const F tlx = x1 > u1 ? x1 : u1, tly = y1 > v1 ? y1 : v1;      // numpy.maximum / minimum of two finite values
              const F brx = x2 < u2 ? x2 : u2, bry = y2 < v2 ? y2 : v2;
              const F w = np_maximum0<F>((brx - tlx) + (F)1), h = np_maximum0<F>((bry - tly) + (F)1);
              const F inter = w * h;
              const F uni = (area_i + area_j) - inter;
              const F iou = inter / uni;
              if (iou > thr) alive[bj] = 0;
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    s = e;
  }
  __syncthreads();
  // ordered gather of the survivors (canonical order: class asc, score desc, index desc)
  int running = 0;
  for (int base = 0; base < n; base += kThreads) {
    const int i = base + tid;
    const bool flag = i < n && alive[order[i]];
    const int slot = ordered_slot(flag, running, s_wave_tot);
    if (flag) keep[slot] = order[i];
  }
  if (tid == 0) *keep_count = running;
}

template <typename F>
__global__ void cxywh_to_tlbr_float_kernel(const F *in, F *out, int n, int cols) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const F *r = in + (long long)i * cols;
  F *o = out + (long long)i * cols;
  // numpy: wh // 2 on floats = floor(wh / 2) (division by two is exact)
  const F cx = r[0], cy = r[1], hw = floor(r[2] / (F)2), hh = floor(r[3] / (F)2);
  o[0] = cx - hw; o[1] = cy - hh; o[2] = cx + hw; o[3] = cy + hh;
  for (int c = 4; c < cols; ++c) o[c] = r[c];
}

// record = 8 x int32: x1 y1 x2 y2 | score bits | class | row | 1
__global__ void pack_records_kernel(const int *cnt, const long long *tlbr, const float *prob,
                                    const long long *cls, const int *row, int rows, int kmax,
                                    int *rec, int *rec_count) {
  const int b = blockIdx.x;
  const int total = cnt[b];
  const int n = total < kmax ? total : kmax;
  if (threadIdx.x == 0 && rec_count) rec_count[b] = total;
  for (int k = threadIdx.x; k < kmax; k += blockDim.x) {
    int *o = rec + ((long long)b * kmax + k) * 8;
    if (k < n) {
      const long long *t = tlbr + ((long long)b * rows + k) * 4;
      // pixel corners are int64 upstream (numpy's astype(int)); a record field is 32 bits: saturate instead of wrapping
      // (only untrained / adversarial weights produce boxes beyond +-2^31 px: exp(tw) * anchor * width)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long long v = t[j];
        o[j] = v > 2147483647ll ? 2147483647 : (v < -2147483648ll ? (int)-2147483648ll : (int)v);
      }
      o[4] = __float_as_int(prob[(long long)b * rows + k]);
      o[5] = (int)cls[(long long)b * rows + k];
      o[6] = row[(long long)b * rows + k];
      o[7] = total;   // the frame's true count rides in every valid record (0 = padding): one collective carries both
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = 0;
    }
  }
}

int next_pow2(int v) {
  int p = 2;
  while (p < v) p <<= 1;
  return p;
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct WsLayout {
  size_t box, prob, cls, row, key, pos, keep, seg, total;
};

WsLayout ws_layout(int batch, int rows) {
  WsLayout w;
  const size_t B = (size_t)batch, R = (size_t)rows, P = (size_t)next_pow2(rows);
  size_t off = 0;
  w.box = off; off = align_up(off + B * R * 4 * sizeof(long long));
  w.prob = off; off = align_up(off + B * R * sizeof(float));
  w.cls = off; off = align_up(off + B * R * sizeof(int));
  w.row = off; off = align_up(off + B * R * sizeof(int));
  w.key = off; off = align_up(off + B * P * sizeof(unsigned long long));
  w.pos = off; off = align_up(off + B * P * sizeof(unsigned int));
  w.keep = off; off = align_up(off + B * R);
  w.seg = off; off = align_up(off + B * (R + R / 64 + 2) * 4 * sizeof(int));   // work items: 4 ints each, <= R/64 + #classes
  w.total = off;
  return w;
}

}  // namespace

extern "C" size_t y3_detect_workspace_bytes(int batch, int rows) {
  if (batch <= 0 || rows <= 0) return 0;
  return ws_layout(batch, rows).total;
}

extern "C" size_t y3_nms_workspace_bytes(int n) {
  if (n <= 0) return 256;
  return ws_layout(1, n).total;
}

extern "C" int y3_detect(const float *d_bbox, const float *d_prob, const int64_t *d_cls, int batch, int rows,
                         const int32_t *d_orig_hw, float prob_thresh, double iou_thresh, void *d_workspace,
                         size_t workspace_bytes, int32_t *d_det_count, int64_t *d_det_tlbr, float *d_det_prob,
                         int64_t *d_det_cls, int32_t *d_det_row, void *stream) {
  Y3_REQUIRE(batch > 0 && rows > 0, "y3_detect: batch and rows must be positive");
  Y3_REQUIRE(d_bbox && d_prob && d_cls && d_orig_hw && d_workspace && d_det_count && d_det_tlbr && d_det_prob &&
                 d_det_cls && d_det_row, "y3_detect: null pointer argument");
  const WsLayout w = ws_layout(batch, rows);
  Y3_REQUIRE(workspace_bytes >= w.total, "y3_detect: workspace too small (%zu < %zu)", workspace_bytes, w.total);
  char *ws = static_cast<char *>(d_workspace);
  DetectArgs a = {};
  a.bbox = d_bbox; a.prob = d_prob; a.cls = reinterpret_cast<const long long *>(d_cls); a.orig_hw = d_orig_hw;
  a.rows = rows; a.rows_p2 = next_pow2(rows);
  a.prob_thresh = prob_thresh; a.iou_thresh = iou_thresh;
  a.c_box = reinterpret_cast<long long *>(ws + w.box);
  a.c_prob = reinterpret_cast<float *>(ws + w.prob);
  a.c_cls = reinterpret_cast<int *>(ws + w.cls);
  a.c_row = reinterpret_cast<int *>(ws + w.row);
  a.s_key = reinterpret_cast<unsigned long long *>(ws + w.key);
  a.s_pos = reinterpret_cast<unsigned int *>(ws + w.pos);
  a.keep = reinterpret_cast<unsigned char *>(ws + w.keep);
  a.seg = reinterpret_cast<int *>(ws + w.seg);
  a.det_count = d_det_count; a.det_tlbr = reinterpret_cast<long long *>(d_det_tlbr); a.det_prob = d_det_prob;
  a.det_cls = reinterpret_cast<long long *>(d_det_cls); a.det_row = d_det_row;
  Y3_LAUNCH(detect_kernel<false>, dim3(batch), dim3(kThreads), 0, static_cast<hipStream_t>(stream), a);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}

extern "C" int y3_nms(const int64_t *d_tlbr, const float *d_prob, const int64_t *d_cls, int n, double iou_thresh,
                      void *d_workspace, size_t workspace_bytes, int64_t *d_keep, int32_t *d_keep_count,
                      void *stream) {
  Y3_REQUIRE(n >= 0, "y3_nms: negative n");
  Y3_REQUIRE(d_keep_count, "y3_nms: null d_keep_count");
  if (n == 0) {
    Y3_HIP_CHECK(hipMemsetAsync(d_keep_count, 0, sizeof(int32_t), static_cast<hipStream_t>(stream)));
    return Y3_OK;
  }
  Y3_REQUIRE(d_tlbr && d_prob && d_workspace && d_keep, "y3_nms: null pointer argument");
  const WsLayout w = ws_layout(1, n);
  Y3_REQUIRE(workspace_bytes >= w.total, "y3_nms: workspace too small (%zu < %zu)", workspace_bytes, w.total);
  char *ws = static_cast<char *>(d_workspace);
  DetectArgs a = {};
  a.in_tlbr = reinterpret_cast<const long long *>(d_tlbr); a.in_prob = d_prob;
  a.in_cls = reinterpret_cast<const long long *>(d_cls); a.n_in = n;
  a.rows = n; a.rows_p2 = next_pow2(n);
  a.iou_thresh = iou_thresh;
  a.s_key = reinterpret_cast<unsigned long long *>(ws + w.key);
  a.s_pos = reinterpret_cast<unsigned int *>(ws + w.pos);
  a.keep = reinterpret_cast<unsigned char *>(ws + w.keep);
  a.seg = reinterpret_cast<int *>(ws + w.seg);
  a.det_count = d_keep_count;
  a.keep_idx = reinterpret_cast<long long *>(d_keep);
  Y3_LAUNCH(detect_kernel<true>, dim3(1), dim3(kThreads), 0, static_cast<hipStream_t>(stream), a);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}

extern "C" size_t y3_nms_float_workspace_bytes(int n) {
  if (n <= 0) return 256;
  return align_up((size_t)next_pow2(n) * sizeof(int)) + align_up((size_t)n);
}

extern "C" int y3_nms_float(const void *d_tlbr, int box_dtype, const double *d_prob, const int64_t *d_cls, int n, double iou_thresh,
                            void *d_workspace, size_t workspace_bytes, int64_t *d_keep, int32_t *d_keep_count, void *stream) {
  Y3_REQUIRE(n >= 0, "y3_nms_float: negative n");
  Y3_REQUIRE(d_keep_count, "y3_nms_float: null d_keep_count");
  Y3_REQUIRE(box_dtype == Y3_F32 || box_dtype == Y3_F64, "y3_nms_float: boxes must be Y3_F32 or Y3_F64");
  if (n == 0) {
    Y3_HIP_CHECK(hipMemsetAsync(d_keep_count, 0, sizeof(int32_t), static_cast<hipStream_t>(stream)));
    return Y3_OK;
  }
  Y3_REQUIRE(d_tlbr && d_prob && d_workspace && d_keep, "y3_nms_float: null pointer argument");
  Y3_REQUIRE(workspace_bytes >= y3_nms_float_workspace_bytes(n), "y3_nms_float: workspace too small");
  const int np2 = next_pow2(n);
  int *order = static_cast<int *>(d_workspace);
  unsigned char *alive = reinterpret_cast<unsigned char *>(static_cast<char *>(d_workspace) + align_up((size_t)np2 * sizeof(int)));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (box_dtype == Y3_F32)
    Y3_LAUNCH(nms_float_kernel<float>, dim3(1), dim3(kThreads), 0, s, static_cast<const float *>(d_tlbr), d_prob,
                       reinterpret_cast<const long long *>(d_cls), n, np2, iou_thresh, order, alive, reinterpret_cast<long long *>(d_keep), d_keep_count);
  else
    Y3_LAUNCH(nms_float_kernel<double>, dim3(1), dim3(kThreads), 0, s, static_cast<const double *>(d_tlbr), d_prob,
                       reinterpret_cast<const long long *>(d_cls), n, np2, iou_thresh, order, alive, reinterpret_cast<long long *>(d_keep), d_keep_count);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}

extern "C" int y3_cxywh_to_tlbr_float(const void *d_xywh, void *d_tlbr, int n, int cols, int dtype, void *stream) {
  Y3_REQUIRE(n >= 0 && cols >= 4, "y3_cxywh_to_tlbr_float: need n >= 0 and at least 4 columns");
  Y3_REQUIRE(dtype == Y3_F32 || dtype == Y3_F64, "y3_cxywh_to_tlbr_float: dtype must be Y3_F32 or Y3_F64");
  if (n == 0) return Y3_OK;
  Y3_REQUIRE(d_xywh && d_tlbr, "y3_cxywh_to_tlbr_float: null pointer argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == Y3_F32)
    Y3_LAUNCH(cxywh_to_tlbr_float_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, s, static_cast<const float *>(d_xywh),
                       static_cast<float *>(d_tlbr), n, cols);
  else
    Y3_LAUNCH(cxywh_to_tlbr_float_kernel<double>, dim3((n + 255) / 256), dim3(256), 0, s, static_cast<const double *>(d_xywh),
                       static_cast<double *>(d_tlbr), n, cols);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}

extern "C" int y3_cxywh_to_tlbr(const int64_t *d_xywh, int64_t *d_tlbr, int n, int cols, void *stream) {
  Y3_REQUIRE(n >= 0 && cols >= 4, "y3_cxywh_to_tlbr: need n >= 0 and at least 4 columns");
  if (n == 0) return Y3_OK;
  Y3_REQUIRE(d_xywh && d_tlbr, "y3_cxywh_to_tlbr: null pointer argument");
  Y3_LAUNCH(cxywh_to_tlbr_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const long long *>(d_xywh), reinterpret_cast<long long *>(d_tlbr), n, cols);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}

extern "C" int y3_pack_records(const int32_t *d_det_count, const int64_t *d_det_tlbr, const float *d_det_prob,
                               const int64_t *d_det_cls, const int32_t *d_det_row, int batch, int rows, int kmax,
                               int32_t *d_records, int32_t *d_rec_count, void *stream) {
  Y3_REQUIRE(batch > 0 && rows > 0 && kmax > 0, "y3_pack_records: sizes must be positive");
  Y3_REQUIRE(d_det_count && d_det_tlbr && d_det_prob && d_det_cls && d_det_row && d_records,
             "y3_pack_records: null pointer argument");
  Y3_LAUNCH(pack_records_kernel, dim3(batch), dim3(256), 0, static_cast<hipStream_t>(stream), d_det_count,
                     reinterpret_cast<const long long *>(d_det_tlbr), d_det_prob,
                     reinterpret_cast<const long long *>(d_det_cls), d_det_row, rows, kmax, d_records, d_rec_count);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}
