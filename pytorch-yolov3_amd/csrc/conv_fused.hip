// First two layers of Darknet-53 in one kernel (gfx950, bf16): uint8 BGR frames -> conv 3x3 s1 (3 -> 32) + BN + leaky
// -> conv 3x3 s2 (32 -> 64) + BN + leaky, NHWC bf16 out.  Replaces the first two conv blocks of
// /root/reference/yolov3/darknet.py:244-257 (models/yolov3.cfg:25-39) plus inference.py:332-333's flip / 255.
//
// Why: at 608x608 x 16 frames the stem's 32-channel output is 379 MB -- written by one kernel and read back (9/4 x
// through L2) by the next, both HBM-bound (0.125 + 0.18 ms, profiles/r01c_per_op.txt).  Here a workgroup owns a
// 16 x 16 tile of the SECOND conv's output, recomputes the 33 x 33 stem pixels it needs from a 35 x 35 x 3 byte
// input patch (6 % extra stem work), keeps them in LDS, and runs the stride-2 conv from there: HBM sees 18 MB of
// input bytes and the 189 MB output only.
//
// Two forms, same arithmetic and bits: conv_stem_s2_fused_kernel below runs the phases one after the other in all eight
// waves (kept as `fuse_stem` = 2: A/B runs and tests); conv_stem_s2_ws_kernel further down -- the default -- splits sixteen
// waves by role and overlaps the stem of tile t+1 with the conv of tile t.  The phases, per tile (first form: persistent
// workgroups, 8 waves, one per CU):
//   phase 1  stem: 69 fragments of 16 stem pixels.  The input patch is kept as 4 bf16 per pixel (B, G, R, 0), so a
//            filter row is 12 contiguous elements and a lane's 8 consecutive K values are two aligned 8-byte LDS
//            reads (k = ky*12 + kx*4 + c; K = 36 padded to 64 = two v_mfma_f32_16x16x32_bf16 steps per 16 channels,
//            the second one only carrying ky = 2, kx = 2); the zero channel and zero-padded K multiply zero
//            weights, so sums equal conv_stem_mfma_kernel's (which gathers 27 scalars per lane, 3x the LDS
//            instructions) up to the order of exact-zero terms; result ->
//            scale/bias/leaky -> bf16 -> stem image in LDS (80-byte pixel pitch: the stride-2 fragment reads of
//            phase 2 are then bank-conflict-free); stem pixels outside the frame are the second conv's zero padding
//   phase 2  stride-2 conv: wave w owns output rows 2w, 2w+1 (2 x 16 pixels) x 64 channels; one MFMA K-step per
//            filter tap (32 input channels = 64 bytes); weights [64][9][32] live in LDS (608-byte channel pitch)
//   phase 3  scale/bias/leaky -> bf16 -> staged through LDS (the stem image is dead by then) -> 16-byte NHWC stores
// The next tile's input patch is fetched into registers during phase 2 and converted into the other patch buffer
// afterwards, so no phase waits on HBM.  Operands, fp32 accumulation and the tap-major order of the second conv equal
// the unfused kernels'; the stem's 27 products sit at other K positions of the MFMA, so a few stem values differ by
// one bf16 ulp (tests/test_gpu_parity.py::test_fused_first_two_convs_output).
#include "common.h"
#include <type_traits>

namespace {

constexpr int kTO = 16;                 // output tile (second conv) is kTO x kTO
constexpr int kSR = 2 * kTO + 1;        // stem rows / cols per tile (33)
constexpr int kIR = kSR + 2;            // input rows / cols per tile (35)
constexpr int kInPitch = 144;           // bf16 elements per input-patch row: 35 pixels x 4 (B, G, R, zero), padded
constexpr int kStemPitch = 80;          // bytes per stem pixel in LDS (64 used)
constexpr int kW1Pitch = 608;           // bytes per output channel of the second conv's weights in LDS (576 used)
constexpr int kNStem = kSR * kSR;       // 1089
constexpr int kNFrag = (kNStem + 15) / 16;
constexpr int kThreads = 512;
constexpr int kInBytes = kIR * kInPitch * 2;                  // one input patch (bf16)
constexpr int kPatchElems = kIR * kIR * 3;                    // 3675 bytes of the frame per patch
constexpr int kPre = (kPatchElems + kThreads - 1) / kThreads; // bytes prefetched per thread (8)
constexpr int kLds = 2 * kInBytes + kNStem * kStemPitch + 64 * kW1Pitch + 512;   // + 256-entry byte -> bf16 table

template <typename T>
struct FusedArgs {
  const unsigned char *in;   // (B, H, W, 3) uint8 BGR
  int H, W, batch;
  const T *w0;          // stem weights [32][32]: k = ky*9 + kx*3 + byte channel, zero padded
  const float *sc0, *bi0;
  uint32_t flags0;
  const T *w1;          // second conv [>= 64][k_ld1], k = (ky*3 + kx)*32 + ci
  int k_ld1;
  const float *sc1, *bi1;
  uint32_t flags1;
  T *out;
  int out_ld, Ho, Wo;
  int tiles_x, tiles_y, n_tiles;
  int dbg;   // diagnostic: bit 0 skip phase 1, bit 1 skip phase 2, bit 2 skip phase 3 stores, bit 3 skip patch prefetch
};

// acc * scale + bias -> LeakyReLU(0.1) -> bf16, four channels at a time; written with 2-wide vectors so that hipcc
// emits v_pk_fma_f32 / v_pk_mul_f32 (the epilogues are VALU-bound: 35 k stem values per tile)
template <typename T>
__device__ __forceinline__ u32x2 bn_leaky_pack4(const f32x4 &a, const f32x4 &sc, const f32x4 &bi) {
  const f32x2 lo = f32x2{a[0], a[1]} * f32x2{sc[0], sc[1]} + f32x2{bi[0], bi[1]};
  const f32x2 hi = f32x2{a[2], a[3]} * f32x2{sc[2], sc[3]} + f32x2{bi[2], bi[3]};
  const f32x2 tl = lo * Y3_LEAKY_SLOPE, th = hi * Y3_LEAKY_SLOPE;
  return y3_pack4<T>(y3_vmax(lo[0], tl[0]), y3_vmax(lo[1], tl[1]), y3_vmax(hi[0], th[0]), y3_vmax(hi[1], th[1]));
}

template <typename T>
__global__ __launch_bounds__(kThreads) void conv_stem_s2_fused_kernel(FusedArgs<T> p) {
  typedef typename H16<T>::v8 V8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *in_tile = reinterpret_cast<T *>(smem);                 // [2][kIR][kInPitch]
  char *stem = smem + 2 * kInBytes;                                    // [kNStem][kStemPitch]
  char *w1s = stem + kNStem * kStemPitch;                              // [64][kW1Pitch]
  T *lut = reinterpret_cast<T *>(w1s + 64 * kW1Pitch);       // lut[v] = bf16(v / 255.0f): one IEEE division per
                                                                       // table entry instead of one per input byte

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;

  // ---- once per workgroup: second conv's weights -> LDS; stem weights / constants -> registers ----
  for (int i = tid; i < 64 * 36; i += kThreads) {                      // 36 chunks of 16 bytes per output channel
    const int co = i / 36, ch = i - co * 36;
    *reinterpret_cast<u32x4 *>(w1s + co * kW1Pitch + ch * 16) =
        *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(p.w1) + ((long long)co * p.k_ld1) * 2 + ch * 16);
  }
  // stem weights re-indexed to k = ky*12 + kx*4 + c (c == 3 and k >= 36: zero); A fragments for both K steps
  V8 w0a[2], w0b[2];
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = st * 32 + fq * 8 + j;
      const int ky = k / 12, rem = k - ky * 12, kx = rem >> 2, c = rem & 3;
      const bool live = k < 36 && c < 3;
      const int kold = live ? ky * 9 + kx * 3 + c : 0;
      const T za = p.w0[(0 + fr) * 32 + kold], zb = p.w0[(16 + fr) * 32 + kold];
      w0a[st][j] = live ? za : (T)0.f;
      w0b[st][j] = live ? zb : (T)0.f;
    }
  // element offsets (relative to the pixel's first element in its patch row) of this lane's two 4-element pieces
  // of K step 0, and of its piece of K step 1 (only fq == 0 carries live values there: ky = 2, kx = 2)
  int off_lo, off_hi;
  {
    const int k0 = fq * 8, k1 = fq * 8 + 4;
    off_lo = (k0 / 12) * kInPitch + (k0 % 12);
    off_hi = (k1 / 12) * kInPitch + (k1 % 12);
  }
  const int off_s1 = 2 * kInPitch + 8;
  const int cq = fq * 4;
  f32x4 sc0[2], bi0[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    sc0[ni] = *reinterpret_cast<const f32x4 *>(p.sc0 + ni * 16 + cq);
    bi0[ni] = *reinterpret_cast<const f32x4 *>(p.bi0 + ni * 16 + cq);
  }

  // bytes of the frame patch of a tile that this thread converts (kPre strided elements).  The patch geometry is the same
  // for every tile, so the element -> (row, byte column) split is done once: per tile only the (uniform) base moves.
  // Tiles that do not touch the frame border (80 % at 608 x 608) skip the bounds tests.
  int pr_row[kPre], pr_col[kPre], pr_dst[kPre];
  uint32_t pr_goff[kPre];
#pragma unroll
  for (int j = 0; j < kPre; ++j) {
    const int i = tid + j * kThreads;
    const int r = i / (kIR * 3), cb = i - r * (kIR * 3);
    pr_row[j] = r;
    pr_col[j] = cb;
    pr_goff[j] = (uint32_t)(r * p.W * 3 + cb);
    pr_dst[j] = r * kInPitch + (cb / 3) * 4 + (cb % 3);
  }
  const bool pr_last_live = tid + (kPre - 1) * kThreads < kPatchElems;   // only the last strided element can be past the end
  auto tile_coords = [&](int tile, int &tx, int &ty, int &b) {
    int t = tile;
    tx = t % p.tiles_x;
    t /= p.tiles_x;
    ty = t % p.tiles_y;
    b = t / p.tiles_y;
  };
  auto patch_fetch = [&](int tile, unsigned char (&pre)[kPre]) {
    int tx, ty, b;
    tile_coords(tile, tx, ty, b);
    const int iy0 = 2 * ty * kTO - 2, ixb0 = (2 * tx * kTO - 2) * 3;
    const bool interior = iy0 >= 0 && iy0 + kIR <= p.H && ixb0 >= 0 && ixb0 + kIR * 3 <= p.W * 3;
    if (interior) {
      const unsigned char *base = p.in + ((long long)b * p.H + iy0) * p.W * 3 + ixb0;   // uniform: scalar base + lane offset
#pragma unroll
      for (int j = 0; j < kPre; ++j) pre[j] = (j < kPre - 1 || pr_last_live) ? base[pr_goff[j]] : (unsigned char)0;
    } else {
#pragma unroll
      for (int j = 0; j < kPre; ++j) {
        const int iy = iy0 + pr_row[j], ixb = ixb0 + pr_col[j];
        unsigned char v = 0;
        if ((j < kPre - 1 || pr_last_live) && (unsigned)iy < (unsigned)p.H && ixb >= 0 && ixb < p.W * 3)
          v = p.in[((long long)b * p.H + iy) * p.W * 3 + ixb];
        pre[j] = v;
      }
    }
  };
  auto patch_store = [&](int buf, const unsigned char (&pre)[kPre]) {
    T *dst = in_tile + buf * (kIR * kInPitch);
#pragma unroll
    for (int j = 0; j < kPre; ++j)
      if (j < kPre - 1 || pr_last_live) dst[pr_dst[j]] = lut[pre[j]];
  };

  for (int i = tid; i < 2 * kIR * kInPitch / 2; i += kThreads) reinterpret_cast<uint32_t *>(in_tile)[i] = 0u;
  if (tid < 256) lut[tid] = (T)((float)tid / 255.0f);
  __syncthreads();   // the fourth channel of every patch pixel stays zero from here on
  int tile = blockIdx.x;
  int buf = 0;
  if (tile < p.n_tiles) {
    unsigned char pre[kPre];
    patch_fetch(tile, pre);
    patch_store(0, pre);
  }
  for (; tile < p.n_tiles; tile += gridDim.x) {
    int tx, ty, b;
    tile_coords(tile, tx, ty, b);
    const int oy0 = ty * kTO, ox0 = tx * kTO;
    __syncthreads();   // B1: this tile's input patch (and, first time, the weights) are in LDS; stem image is free

    // ---- phase 1: stem ------------------------------------------------------------------------------
    const T *patch = in_tile + buf * (kIR * kInPitch);
    // Three fragments per wave are in flight at a time: a fragment is one dependent chain LDS read -> two MFMAs ->
    // scale / bias / leaky -> LDS write, and with two waves per SIMD a chain at a time left every latency exposed
    // (690 cycles per fragment; the phase was 42 % of the kernel, profiles/r02m_stem_phases.txt).
    // All stem pixels of a tile lie inside the frame unless the tile touches the frame border.
    const bool stem_inside = 2 * oy0 - 1 >= 0 && 2 * oy0 - 1 + kSR <= p.H && 2 * ox0 - 1 >= 0 && 2 * ox0 - 1 + kSR <= p.W;
    constexpr int kFG = 3;                                               // fragments in flight per wave
    constexpr int kFragIters = (kNFrag + kThreads / 64 - 1) / (kThreads / 64);
    static_assert(kFragIters % kFG == 0, "fragment groups");
#pragma unroll 1
    for (int g = 0; g < ((p.dbg & 1) ? 0 : kFragIters); g += kFG) {
      u32x2 lo[kFG], hi[kFG], s1[kFG];
      int qv[kFG], syv[kFG], sxv[kFG];
#pragma unroll
      for (int j = 0; j < kFG; ++j) {
        const int q = (wave + (g + j) * (kThreads / 64)) * 16 + fr;
        const int qc = q < kNStem ? q : kNStem - 1;
        const int sy = qc / kSR, sx = qc - sy * kSR;
        const T *base = patch + sy * kInPitch + sx * 4;
        lo[j] = *reinterpret_cast<const u32x2 *>(base + off_lo);
        hi[j] = *reinterpret_cast<const u32x2 *>(base + off_hi);
        s1[j] = *reinterpret_cast<const u32x2 *>(base + off_s1);
        qv[j] = q;
        syv[j] = sy;
        sxv[j] = sx;
      }
      f32x4 acc0[kFG], acc1[kFG];
#pragma unroll
      for (int j = 0; j < kFG; ++j) {
        const V8 xf0 = __builtin_bit_cast(V8, u32x4{lo[j][0], lo[j][1], hi[j][0], hi[j][1]});
        acc0[j] = H16<T>::mfma(w0a[0], xf0, f32x4{0.f, 0.f, 0.f, 0.f});
        acc1[j] = H16<T>::mfma(w0b[0], xf0, f32x4{0.f, 0.f, 0.f, 0.f});
      }
#pragma unroll
      for (int j = 0; j < kFG; ++j) {
        const V8 xf1 = __builtin_bit_cast(V8, fq == 0 ? u32x4{s1[j][0], s1[j][1], 0u, 0u} : u32x4{0u, 0u, 0u, 0u});
        acc0[j] = H16<T>::mfma(w0a[1], xf1, acc0[j]);
        acc1[j] = H16<T>::mfma(w0b[1], xf1, acc1[j]);
      }
#pragma unroll
      for (int j = 0; j < kFG; ++j) {
        if (qv[j] < kNStem) {
          bool inside = true;
          if (!stem_inside) {
            const int gy = 2 * oy0 - 1 + syv[j], gx = 2 * ox0 - 1 + sxv[j];   // stem pixel in frame coordinates
            inside = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
          }
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            u32x2 o = bn_leaky_pack4<T>(ni ? acc1[j] : acc0[j], sc0[ni], bi0[ni]);
            if (!inside) o = u32x2{0u, 0u};                             // the second conv's zero padding
            *reinterpret_cast<u32x2 *>(stem + qv[j] * kStemPitch + (ni * 16 + cq) * 2) = o;
          }
        }
      }
    }
    // next tile's patch bytes start flying now; they are converted after phase 2
    const int next_tile = tile + gridDim.x;
    unsigned char pre[kPre];
    if (next_tile < p.n_tiles && !(p.dbg & 8)) patch_fetch(next_tile, pre);
    __syncthreads();   // B2: stem image complete

    // ---- phase 2: 3x3 stride-2 conv from the stem image ---------------------------------------------
    f32x4 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) acc[mi][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < ((p.dbg & 2) ? 0 : 9); ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      u32x4 xf[2], wf[4];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int q = (2 * (2 * wave + mi) + ky) * kSR + 2 * fr + kx;
        xf[mi] = *reinterpret_cast<const u32x4 *>(stem + q * kStemPitch + fq * 16);
      }
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
        wf[nf] = *reinterpret_cast<const u32x4 *>(w1s + (nf * 16 + fr) * kW1Pitch + tap * 64 + fq * 16);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
          acc[mi][nf] = H16<T>::mfma(__builtin_bit_cast(V8, wf[nf]),
                                                                __builtin_bit_cast(V8, xf[mi]), acc[mi][nf]);
    }
    if (next_tile < p.n_tiles) patch_store(buf ^ 1, pre);
    __syncthreads();   // B3: nobody reads the stem image any more

    // ---- phase 3: epilogue through LDS (256 pixels x 128 bytes, 16-byte chunks XOR-swizzled with the pixel) ----
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      const f32x4 s1 = *reinterpret_cast<const f32x4 *>(p.sc1 + nf * 16 + cq);
      const f32x4 b1 = *reinterpret_cast<const f32x4 *>(p.bi1 + nf * 16 + cq);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int px = (2 * wave + mi) * kTO + fr;
        const int co = nf * 16 + cq;                                   // 4 consecutive channels
        *reinterpret_cast<u32x2 *>(stem + px * 128 + (((co >> 3) ^ (px & 7)) << 4) + (co & 7) * 2) =
            bn_leaky_pack4<T>(acc[mi][nf], s1, b1);
      }
    }
    __syncthreads();   // B4
#pragma unroll
    for (int j = 0; j < kTO * kTO * 8 / kThreads; ++j) {
      const int i = tid + j * kThreads;
      const int px = i >> 3, ch = i & 7;
      const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
      if (oy < p.Ho && ox < p.Wo && !(p.dbg & 4)) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(stem + px * 128 + ((ch ^ (px & 7)) << 4));
        *reinterpret_cast<u32x4 *>(p.out + (((long long)b * p.Ho + oy) * p.Wo + ox) * p.out_ld + ch * 8) = v;
      }
    }
    buf ^= 1;
  }
}


// ------------------------------------------------------------------------------------------------
// Wave-specialised, pipelined form of the kernel above (the default).  Same arithmetic, same bits; what changes is WHO
// does what WHEN.  The kernel above runs its phases one after the other in every wave -- stem, stride-2 conv, write-out
// -- with four workgroup barriers per tile and two waves per SIMD: every LDS / MFMA / global latency of those dependent
// chains is exposed (profiles/r02m_stem_phases.txt: 7 k cycles per 256 outputs for ~1.1 k instructions per wave; with
// stem, conv, stores and patch handling all switched off 2 k cycles remain).  Here a workgroup owns 16 x 8 output
// tiles, keeps TWO stem images (17 x 33 pixels each) in LDS, and splits its waves by role:
//   waves 0-7   "conv":  stride-2 conv of tile t out of image t (one output row of 16 pixels x 64 channels each: 36
//               MFMAs), then a wave-private write-out (the wave stages its own 16 pixels x 128 bytes and stores them as
//               two 1 KiB runs);
//   waves 8..   "stem":  fetch the input patch of tile t+2 (bytes in registers), compute the stem image of tile t+1
//               from patch t+1 (MFMA + scale / bias / leaky, VALU bound), convert patch t+2 into LDS.
// The matrix work of one tile and the vector work of the next run on the same SIMDs at the same time, twice the waves
// hide each other's latencies, and ONE workgroup barrier per tile is left.  Patch geometry and the stem fragments' LDS
// addresses do not depend on the tile and are computed once per workgroup.
constexpr int kPX = 16, kPY = 8;                  // output tile: 16 wide, 8 tall (one output row per conv wave)
constexpr int kPSX = 2 * kPX + 1;                 // stem columns per tile (33)
constexpr int kPSY = 2 * kPY + 1;                 // stem rows per tile (17)
constexpr int kPIX = kPSX + 2, kPIY = kPSY + 2;   // input patch 35 x 19 pixels
constexpr int kPNStem = kPSX * kPSY;              // 561
constexpr int kPNFrag = (kPNStem + 15) / 16;      // 36 fragments of 16 stem pixels
constexpr int kPInBytes = kPIY * kInPitch * 2;    // one input patch (bf16): 5472 bytes
constexpr int kPStemBytes = kPNStem * kStemPitch; // 44880
constexpr int kPStage = 8 * 16 * 128;             // write-out staging: 8 conv waves x 16 pixels x 128 bytes
constexpr int kPLds = 2 * kPInBytes + 2 * kPStemBytes + 64 * kW1Pitch + 512 + kPStage;
static_assert(kPLds <= 160 * 1024, "LDS budget");
static_assert(kPX == 16 && kPY == 8, "one 16-pixel output row per conv wave");

template <typename T, int NC>   // conv waves (4: two output rows each, 8: one); the other 16 - NC waves are stem waves
__global__ __launch_bounds__(1024) void conv_stem_s2_ws_kernel(FusedArgs<T> p) {
  typedef typename H16<T>::v8 V8;
  constexpr int NS = 16 - NC, NT = 1024;
  constexpr int MR = kPY / NC;                                         // output rows per conv wave
  constexpr int kIters = (kPNFrag + NS - 1) / NS;                      // stem fragments per stem wave
  constexpr int kPairs = kPIY * ((kPIX * 3 + 1) / 2);                  // the patch as byte pairs: 19 rows x 53
  constexpr int kPre = (kPairs + 64 * NS - 1) / (64 * NS);             // pairs per stem thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *in_tile = smem;                                                // [2][kPIY][kInPitch] bf16
  char *stem0 = smem + 2 * kPInBytes;                                  // [2][kPNStem][kStemPitch]
  char *w1s = stem0 + 2 * kPStemBytes;                                 // [64][kW1Pitch]
  T *lut = reinterpret_cast<T *>(w1s + 64 * kW1Pitch);       // lut[v] = bf16(v / 255.0f)
  char *stage = reinterpret_cast<char *>(lut) + 512;                   // [8][16][128]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const bool is_stem = wave >= NC;

  // ---- once per workgroup ----
  for (int i = tid; i < 64 * 36; i += NT) {                            // second conv's weights: 36 chunks of 16 bytes per channel
    const int j = i / 36, ch = i - j * 36;
    const int co = y3_pair_perm(j);                                     // LDS row j holds channel co (common.h)
    *reinterpret_cast<u32x4 *>(w1s + j * kW1Pitch + ch * 16) =
        *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(p.w1) + ((long long)co * p.k_ld1) * 2 + ch * 16);
  }
  for (int i = tid; i < 2 * kPInBytes / 4; i += NT) reinterpret_cast<uint32_t *>(in_tile)[i] = 0u;
  if (tid < 256) lut[tid] = (T)((float)tid / 255.0f);
  __syncthreads();   // the fourth channel of every patch pixel stays zero from here on
  const int tile0 = blockIdx.x, stride = gridDim.x;
  if (tile0 >= p.n_tiles) return;
  // tile -> (tile column, tile row, frame): divisions once, then carried additions per step (a step advances by `stride`)
  struct TilePos { int tile, tx, ty, b; };
  auto pos_of = [&](int tile) {
    TilePos t;
    t.tile = tile;
    t.tx = tile % p.tiles_x;
    const int r = tile / p.tiles_x;
    t.ty = r % p.tiles_y;
    t.b = r / p.tiles_y;
    return t;
  };
  const TilePos dpos = pos_of(stride);
  auto advance = [&](TilePos &t) {
    t.tile += stride;
    t.tx += dpos.tx;
    int c = t.tx >= p.tiles_x ? 1 : 0;
    t.tx -= c ? p.tiles_x : 0;
    t.ty += dpos.ty + c;
    c = t.ty >= p.tiles_y ? 1 : 0;
    t.ty -= c ? p.tiles_y : 0;
    t.b += dpos.b + c;
  };

  if (is_stem) {
    // =============================== stem waves ===============================
    const int sw = wave - NC, stid = tid - 64 * NC;
    V8 w0a[2], w0b[2];                                             // stem weights, k = ky*12 + kx*4 + c (see above)
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = st * 32 + fq * 8 + j;
        const int ky = k / 12, rem = k - ky * 12, kx = rem >> 2, c = rem & 3;
        const bool live = k < 36 && c < 3;
        const int kold = live ? ky * 9 + kx * 3 + c : 0;
        // MFMA row fr of the first / second fragment is channel 8 (fr >> 2) + (fr & 3) (+ 4): y3_pair_perm
        const T za = p.w0[y3_pair_perm(fr) * 32 + kold], zb = p.w0[y3_pair_perm(16 + fr) * 32 + kold];
        w0a[st][j] = live ? za : (T)0.f;
        w0b[st][j] = live ? zb : (T)0.f;
      }
    int off_lo, off_hi;                                                 // bytes, relative to the pixel's first patch element
    {
      const int k0 = fq * 8, k1 = fq * 8 + 4;
      off_lo = ((k0 / 12) * kInPitch + (k0 % 12)) * 2;
      off_hi = ((k1 / 12) * kInPitch + (k1 % 12)) * 2;
    }
    constexpr int off_s1 = (2 * kInPitch + 8) * 2;
    f32x4 sc0[2], bi0[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      sc0[ni] = *reinterpret_cast<const f32x4 *>(p.sc0 + fq * 8 + ni * 4);
      bi0[ni] = *reinterpret_cast<const f32x4 *>(p.bi0 + fq * 8 + ni * 4);
    }
    // fragments of this wave: f = sw + NS j; lane fr handles stem pixel q = 16 f + fr
    int fr_patch[kIters];                                               // byte offset of the pixel's patch element 0
    uint32_t fr_live = 0;                                               // bit j: q < kPNStem
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const int q = (sw + NS * j) * 16 + fr;
      const int qc = q < kPNStem ? q : kPNStem - 1;
      const int sy = qc / kPSX, sx = qc - sy * kPSX;
      fr_patch[j] = (sy * kInPitch + sx * 4) * 2;
      fr_live |= (q < kPNStem ? 1u : 0u) << j;
    }
    int stem_wr = ((sw * 16 + fr) * kStemPitch) + fq * 16;             // + j * NS * 16 * kStemPitch; channels 8 fq .. 8 fq + 7
    asm volatile("" : "+v"(stem_wr));                                   // one live register (kept as its two terms it spilled)
    // input patch as byte PAIRS (half the registers per tile in flight): pair k of patch row r = bytes 2k, 2k + 1.
    // pr_dst: LDS byte offset of the first byte's bf16 element | bit 0: the second byte skips the zero channel (its
    // element is two further, not one); the last pair of a row only has its first byte inside the patch.
    constexpr int kRowPairs = (kPIX * 3 + 1) / 2;                       // 53
    int pr_dst[kPre];
    uint32_t pr_goff[kPre];
    uint32_t pr_one = 0;                                                // bit j: only the first byte of pair j is live
#pragma unroll
    for (int j = 0; j < kPre; ++j) {
      const int i = stid + j * 64 * NS;
      const int r = i / kRowPairs, cb = 2 * (i - r * kRowPairs);
      pr_goff[j] = (uint32_t)(r * p.W * 3 + cb);
      pr_dst[j] = (r * kInPitch + (cb / 3) * 4 + (cb % 3)) * 2 | (cb % 3 == 2 ? 1 : 0);
      pr_one |= (cb + 1 >= kPIX * 3 ? 1u : 0u) << j;
    }
    const bool pr_last_live = stid + (kPre - 1) * 64 * NS < kPairs;     // only the last strided pair can be past the end
    const bool pairs_aligned = ((p.W * 3) & 1) == 0 && (reinterpret_cast<size_t>(p.in) & 1) == 0;
    auto patch_fetch = [&](const TilePos &t, unsigned short (&pre)[kPre]) {
      const int tx = t.tx, ty = t.ty, b = t.b;
      const int iy0 = 2 * ty * kPY - 2, ixb0 = (2 * tx * kPX - 2) * 3;   // ixb0 is even
      // fast path: the whole patch (and the byte after each row) lies inside the frame, 2-byte loads are aligned
      const bool interior = pairs_aligned && iy0 >= 0 && iy0 + kPIY <= p.H && ixb0 >= 0 && ixb0 + 2 * kRowPairs <= p.W * 3;
      if (interior) {
        const unsigned char *base = p.in + ((long long)b * p.H + iy0) * p.W * 3 + ixb0;   // uniform base + lane offset
#pragma unroll
        for (int j = 0; j < kPre; ++j)
          pre[j] = (j < kPre - 1 || pr_last_live) ? *reinterpret_cast<const unsigned short *>(base + pr_goff[j]) : (unsigned short)0;
      } else {
#pragma unroll
        for (int j = 0; j < kPre; ++j) {
          const int i = stid + j * 64 * NS;
          const int r = i / kRowPairs, cb = 2 * (i - r * kRowPairs);
          const int iy = iy0 + r, ixb = ixb0 + cb;
          unsigned int v = 0;
          if ((j < kPre - 1 || pr_last_live) && (unsigned)iy < (unsigned)p.H) {
            const unsigned char *row = p.in + ((long long)b * p.H + iy) * p.W * 3;
            if (ixb >= 0 && ixb < p.W * 3) v = row[ixb];
            if (ixb + 1 >= 0 && ixb + 1 < p.W * 3 && cb + 1 < kPIX * 3) v |= (unsigned int)row[ixb + 1] << 8;
          }
          pre[j] = (unsigned short)v;
        }
      }
    };
    auto patch_store = [&](int buf, const unsigned short (&pre)[kPre]) {
      char *dst = in_tile + buf * kPInBytes;
#pragma unroll
      for (int j = 0; j < kPre; ++j)
        if (j < kPre - 1 || pr_last_live) {
          char *d = dst + (pr_dst[j] & ~1);
          *reinterpret_cast<T *>(d) = lut[pre[j] & 255];
          if (!((pr_one >> j) & 1u)) *reinterpret_cast<T *>(d + 2 + 2 * (pr_dst[j] & 1)) = lut[pre[j] >> 8];
        }
    };
    // stem image of one tile: kFG fragments in flight (LDS reads, two K steps each, scale / bias / leaky, LDS writes)
    auto stem_image = [&](const char *patch, char *img, const TilePos &t) {
      const int oy0 = t.ty * kPY, ox0 = t.tx * kPX;
      const bool all_inside = 2 * oy0 - 1 >= 0 && 2 * oy0 - 1 + kPSY <= p.H && 2 * ox0 - 1 >= 0 && 2 * ox0 - 1 + kPSX <= p.W;
      auto group = [&](auto g0, auto cnt) {
        constexpr int g = decltype(g0)::value, G = decltype(cnt)::value;
        u32x2 lo[G], hi[G], s1[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const char *base = patch + fr_patch[g + j];
          lo[j] = *reinterpret_cast<const u32x2 *>(base + off_lo);
          hi[j] = *reinterpret_cast<const u32x2 *>(base + off_hi);
          s1[j] = *reinterpret_cast<const u32x2 *>(base + off_s1);
        }
        f32x4 a0[G], a1[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const V8 xf0 = __builtin_bit_cast(V8, u32x4{lo[j][0], lo[j][1], hi[j][0], hi[j][1]});
          a0[j] = H16<T>::mfma(w0a[0], xf0, f32x4{0.f, 0.f, 0.f, 0.f});
          a1[j] = H16<T>::mfma(w0b[0], xf0, f32x4{0.f, 0.f, 0.f, 0.f});
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
          // K step 1 only carries tap (2, 2) in k = 32..34 (lanes fq == 0); every other weight of the step is zero, so
          // the data there only has to be finite: the same (valid) patch bytes again instead of selected zeros
          const V8 xf1 = __builtin_bit_cast(V8, u32x4{s1[j][0], s1[j][1], s1[j][0], s1[j][1]});
          a0[j] = H16<T>::mfma(w0a[1], xf1, a0[j]);
          a1[j] = H16<T>::mfma(w0b[1], xf1, a1[j]);
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
          u32x2 o0 = bn_leaky_pack4<T>(a0[j], sc0[0], bi0[0]);
          u32x2 o1 = bn_leaky_pack4<T>(a1[j], sc0[1], bi0[1]);
          if (!all_inside) {                                            // uniform: tiles on the frame border only
            const int q = (sw + NS * (g + j)) * 16 + fr;
            const int qc = q < kPNStem ? q : kPNStem - 1;
            const int sy = qc / kPSX, sx = qc - sy * kPSX;
            const int gy = 2 * oy0 - 1 + sy, gx = 2 * ox0 - 1 + sx;
            if (!((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)) {   // the second conv's zero padding
              o0 = u32x2{0u, 0u};
              o1 = u32x2{0u, 0u};
            }
          }
          if ((fr_live >> (g + j)) & 1u) {
            char *dst = img + stem_wr + (g + j) * (NS * 16 * kStemPitch);
            *reinterpret_cast<u32x4 *>(dst) = u32x4{o0[0], o0[1], o1[0], o1[1]};   // one 16-byte write: no bank conflicts at 80-byte pitch
          }
        }
      };
      using std::integral_constant;
      static_assert(kIters == 5 || kIters == 3, "fragment groups");
      group(integral_constant<int, 0>{}, integral_constant<int, 3>{});
      if constexpr (kIters == 5) group(integral_constant<int, 3>{}, integral_constant<int, 2>{});
    };

    // Input patches are fetched FOUR steps ahead of their conversion into LDS (a step is ~2 k cycles, an HBM miss under
    // load more than that: with one step of distance the stem waves sat on vmcnt and the step took the memory latency).
    // Four byte-register sets, one per tile in flight; the step loop is unrolled by four so that every set is addressed
    // statically (set = step index mod 4; a set is refilled in the step after it was converted).
    unsigned short pre[4][kPre];
    TilePos fpos = pos_of(tile0);                                       // next tile whose patch is fetched
    TilePos ipos = fpos;                                                // next tile whose stem image is computed
    {
      patch_fetch(fpos, pre[0]);
      patch_store(0, pre[0]);
      advance(fpos);
      if (fpos.tile < p.n_tiles) {
        patch_fetch(fpos, pre[1]);
        patch_store(1, pre[1]);
      }
      advance(fpos);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();   // S1 (stem waves write the patches; everybody waits)
#pragma unroll
    for (int k = 0; k < 3; ++k) {                                        // the patch of step i + 2 lives in set i % 4
      if (fpos.tile < p.n_tiles) patch_fetch(fpos, pre[k]);
      advance(fpos);
    }
    stem_image(in_tile, stem0, ipos);                                  // pipeline fill: image of the first tile
    advance(ipos);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();   // S2
    int par = 0;
    Y3_STAMP_DECL
    auto step = [&](auto setc) {                                        // false: no more tiles
      constexpr int set = decltype(setc)::value;
      // here: ipos = tile + stride (image to compute), fpos = tile + 5 stride (patch to fetch)
      if (fpos.tile < p.n_tiles) patch_fetch(fpos, pre[(set + 3) & 3]);
      advance(fpos);
      Y3_STAMP(0);
      const bool more = ipos.tile < p.n_tiles;
      if (more) stem_image(in_tile + (par ^ 1) * kPInBytes, stem0 + (par ^ 1) * kPStemBytes, ipos);
      advance(ipos);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      Y3_STAMP(1);
      if (ipos.tile < p.n_tiles) patch_store(par, pre[set]);           // patch buffer `par` held this tile's patch: free
      __builtin_amdgcn_s_waitcnt(0xC07F);
      Y3_STAMP(2);
      __builtin_amdgcn_s_barrier();   // T: image t+1 complete, image t read for the last time, patch t+2 stored
      Y3_STAMP(3);
      par ^= 1;
      return more;
    };
    using std::integral_constant;
    for (;;) {
      if (!step(integral_constant<int, 0>{})) break;
      if (!step(integral_constant<int, 1>{})) break;
      if (!step(integral_constant<int, 2>{})) break;
      if (!step(integral_constant<int, 3>{})) break;
    }
#ifdef Y3_STAMPS
    if (tid == 64 * NC) for (int _i = 0; _i < 4; ++_i) atomicAdd(&g_y3_stamps[_i], _st_acc[_i]);
#endif
    return;
  }

  // =============================== conv waves ===============================
  // operand addresses: output rows wave * MR + mr of the tile, pixel fr; tap (ky, kx) adds (ky * 33 + kx) * 80
  const int a_rd = ((2 * wave * MR) * kPSX + 2 * fr) * kStemPitch + fq * 16;   // + mr * 2 * 33 * 80
  const int b_rd = fr * kW1Pitch + fq * 16;                            // + nf * 16 * kW1Pitch + tap * 64
  char *my_stage = stage + wave * (MR * 16 * 128);
  f32x4 s1v[4], b1v[4];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) {
    s1v[nf] = *reinterpret_cast<const f32x4 *>(p.sc1 + (nf >> 1) * 32 + fq * 8 + (nf & 1) * 4);   // channels of acc[.][nf]
    b1v[nf] = *reinterpret_cast<const f32x4 *>(p.bi1 + (nf >> 1) * 32 + fq * 8 + (nf & 1) * 4);
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();     // S1
  __builtin_amdgcn_s_barrier();     // S2
  // write-out: lane -> (pixel, 16-byte channel chunk) of a row, as a 32-bit offset from the row's first pixel
  uint32_t st_off[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) st_off[k] = (uint32_t)((((lane >> 3) + 8 * k) * p.out_ld + (lane & 7) * 8) * 2);
  int par = 0;
  Y3_STAMP_DECL
  for (TilePos pos = pos_of(tile0); pos.tile < p.n_tiles; advance(pos)) {
    const int b = pos.b;
    const int oy0 = pos.ty * kPY, ox0 = pos.tx * kPX;
    const char *img = stem0 + par * kPStemBytes;
    f32x4 acc[MR][4];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) acc[mr][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      u32x4 xf[MR], wf[4];
#pragma unroll
      for (int mr = 0; mr < MR; ++mr)
        xf[mr] = *reinterpret_cast<const u32x4 *>(img + a_rd + ((2 * mr + ky) * kPSX + kx) * kStemPitch);
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
        wf[nf] = *reinterpret_cast<const u32x4 *>(w1s + b_rd + nf * 16 * kW1Pitch + tap * 64);
#pragma unroll
      for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
          acc[mr][nf] = H16<T>::mfma(__builtin_bit_cast(V8, wf[nf]),
                                                                __builtin_bit_cast(V8, xf[mr]), acc[mr][nf]);
    }
    Y3_STAMP(4);
    // write-out, wave-private: bn + leaky -> bf16 -> this wave's 16 pixels x 128 bytes per row -> two 1 KiB runs per row
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
      for (int k = 0; k < 2; ++k) {                                      // fragment pair k: channels 32 k + 8 fq .. + 7
        const u32x2 lo = bn_leaky_pack4<T>(acc[mr][2 * k], s1v[2 * k], b1v[2 * k]);
        const u32x2 hi = bn_leaky_pack4<T>(acc[mr][2 * k + 1], s1v[2 * k + 1], b1v[2 * k + 1]);
        *reinterpret_cast<u32x4 *>(my_stage + mr * 2048 + fr * 128 + (((k * 4 + fq) ^ (fr & 7)) << 4)) = u32x4{lo[0], lo[1], hi[0], hi[1]};
      }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
      const int oy = oy0 + wave * MR + mr;
      char *row = reinterpret_cast<char *>(p.out + (((long long)b * p.Ho + oy) * p.Wo + ox0) * p.out_ld);   // uniform
      const bool full = ox0 + kPX <= p.Wo;                              // uniform
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int px = (lane >> 3) + 8 * k, ch = lane & 7;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(my_stage + mr * 2048 + px * 128 + ((ch ^ (px & 7)) << 4));
        if (oy < p.Ho && (full || ox0 + px < p.Wo)) *reinterpret_cast<u32x4 *>(row + st_off[k]) = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    Y3_STAMP(5);
    __builtin_amdgcn_s_barrier();   // T
    Y3_STAMP(6);
    par ^= 1;
  }
#ifdef Y3_STAMPS
  if (tid == 0) {
    for (int _i = 4; _i < 7; ++_i) atomicAdd(&g_y3_stamps[_i], _st_acc[_i]);
    atomicAdd(&g_y3_stamps[7], 1ull);
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// One residual block of Darknet-53 in one kernel, for the block whose 3x3 weights fit in LDS (64 -> 32 -> 64 channels,
// the 304^2 block of yolov3@608):   z = leaky(bn(conv3x3(leaky(bn(conv1x1(x)))))) + x      (darknet.py:244-257, :376-379;
// models/yolov3.cfg:41-61).  Separately the 1x1 reads 189 MB and writes 95 MB, the 3x3 reads those 95 MB (9x through
// L2) plus the 189 MB shortcut operand and writes 189 MB (0.06 + 0.16 ms, both HBM-bound); here a workgroup loads an
// 18 x 18 x 64 patch of x once, keeps the 1x1's 32-channel output for it in LDS, runs the 3x3 from there and takes the
// shortcut operand from the same patch: HBM sees x (1.27x for the halo) and z only.
// Per 16 x 16 output tile (persistent workgroups, 8 waves):  phase A 1x1 over the 324 patch pixels (MFMA, pixels
// outside the frame give the 3x3's zero padding) -> 96-byte-pitch image;  phase B 3x3, one MFMA K-step per tap, wave w
// owns output rows 2w, 2w+1;  phase C bn + leaky + shortcut -> bf16 -> staged -> 16-byte NHWC stores.  The next tile's
// patch is fetched into registers during phase B and written to LDS after phase C.
constexpr int kRT = 16;                  // output tile
constexpr int kRP = kRT + 2;             // patch rows / cols (18)
constexpr int kRNP = kRP * kRP;          // 324 patch pixels
constexpr int kRFrag = (kRNP + 15) / 16; // 21
constexpr int kYPitch = 96;              // bytes per pixel of the 1x1's output image (64 used): conflict-free unit-stride reads
constexpr int kXBytes = kRNP * 128;
constexpr int kYBytes = 32768;           // >= kRNP * kYPitch, and holds the 256 x 128-byte staging tile
constexpr int kRLds = kXBytes + kYBytes + 64 * kW1Pitch;
constexpr int kXPre = (kRNP * 8 + kThreads - 1) / kThreads;   // 16-byte chunks of x prefetched per thread (6)
static_assert(kRNP * kYPitch <= kYBytes && kRT * kRT * 128 <= kYBytes, "image / staging tile must fit");

template <typename T>
struct ResArgs {
  const T *x;
  int H, W, batch, x_ld;
  const T *w2; int k_ld2; const float *sc2, *bi2;   // 1x1: [>= 32][k_ld2], k = ci
  const T *w3; int k_ld3; const float *sc3, *bi3;   // 3x3: [>= 64][k_ld3], k = (ky*3 + kx)*32 + ci
  T *out;
  int out_ld;
  int tiles_x, tiles_y, n_tiles;
};

template <typename T>
__global__ __launch_bounds__(kThreads) void conv_resblock_fused_kernel(ResArgs<T> p) {
  typedef typename H16<T>::v8 V8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *xt = smem;                        // [kRNP][128], 16-byte chunks XOR-swizzled with (pixel & 7)
  char *yt = smem + kXBytes;              // [kRNP][kYPitch]; later the staging tile
  char *w3s = yt + kYBytes;               // [64][kW1Pitch]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;

  for (int i = tid; i < 64 * 36; i += kThreads) {
    const int j = i / 36, ch = i - j * 36;
    const int co = y3_pair_perm(j);                                     // LDS row j holds channel co (common.h)
    *reinterpret_cast<u32x4 *>(w3s + j * kW1Pitch + ch * 16) =
        *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(p.w3) + ((long long)co * p.k_ld3) * 2 + ch * 16);
  }
  // 1x1 weights: A fragments [co-frag][K step], lane (MFMA row fr = channel y3_pair_perm(ni*16 + fr), k = ks*32 + fq*8 ..)
  u32x4 w2f[2][2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      w2f[ni][ks] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(p.w2) +
                                                    ((long long)y3_pair_perm(ni * 16 + fr) * p.k_ld2 + ks * 32 + fq * 8) * 2);
  f32x4 sc2[2], bi2[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    sc2[ni] = *reinterpret_cast<const f32x4 *>(p.sc2 + fq * 8 + ni * 4);
    bi2[ni] = *reinterpret_cast<const f32x4 *>(p.bi2 + fq * 8 + ni * 4);
  }

  auto tile_origin = [&](int tile, int &b, int &oy0, int &ox0) {
    const int tx = tile % p.tiles_x;
    tile /= p.tiles_x;
    oy0 = (tile % p.tiles_y) * kRT;
    b = tile / p.tiles_y;
    ox0 = tx * kRT;
  };
  // Patch chunk j of this thread: patch pixel (r, cc), 16-byte channel chunk c -- the same for every tile, so the patch
  // row / column, the byte offset relative to the tile's first patch pixel and the LDS address are computed ONCE; per tile
  // a chunk costs two range checks and a load from (wave-uniform tile base) + (32-bit lane offset).
  int xrc[kXPre];                                                        // (r << 8) | cc, or -1 past the patch
  uint32_t xoff[kXPre];                                                  // ((r * W + cc) * x_ld) * 2 + c * 16
  int xlds[kXPre];
#pragma unroll
  for (int j = 0; j < kXPre; ++j) {
    const int i = tid + j * kThreads;
    const int px = i >> 3, c = i & 7;
    const int r = px / kRP, cc = px - r * kRP;
    xrc[j] = i < kRNP * 8 ? (r << 8) | cc : -1;
    xoff[j] = ((uint32_t)(r * p.W + cc) * (uint32_t)p.x_ld) * 2u + (uint32_t)c * 16u;
    xlds[j] = px * 128 + ((c ^ (px & 7)) << 4);
  }
  // phase-A fragments of this wave (f = wave, wave + 8, wave + 16): patch read address, patch (r, cc), image write address
  constexpr int kAFr = (kRFrag + kThreads / 64 - 1) / (kThreads / 64);
  int a_rd[kAFr], a_rc[kAFr], a_wr[kAFr];
#pragma unroll
  for (int fs = 0; fs < kAFr; ++fs) {
    const int q = (wave + fs * (kThreads / 64)) * 16 + fr;
    const int qc = q < kRNP ? q : kRNP - 1;
    const int r = qc / kRP, cc = qc - r * kRP;
    a_rd[fs] = qc * 128 + ((fq ^ (qc & 7)) << 4);
    a_rc[fs] = (r << 8) | cc;
    a_wr[fs] = q < kRNP ? q * kYPitch + fq * 16 : -1;
  }
  // write-out item j of this thread: tile pixel opx, byte offset from the tile's first output pixel, staging address
  constexpr int kOutPer = kRT * kRT * 8 / kThreads;
  int opx[kOutPer], olds[kOutPer];
  uint32_t ooff[kOutPer];
#pragma unroll
  for (int j = 0; j < kOutPer; ++j) {
    const int i = tid + j * kThreads;
    const int px = i >> 3, ch = i & 7;
    opx[j] = px;
    ooff[j] = ((uint32_t)((px >> 4) * p.W + (px & 15)) * (uint32_t)p.out_ld + (uint32_t)ch * 8u) * 2u;
    olds[j] = px * 128 + ((ch ^ (px & 7)) << 4);
  }
  auto x_fetch = [&](int tile, u32x4 (&pre)[kXPre]) {
    int b, oy0, ox0;
    tile_origin(tile, b, oy0, ox0);
    // first patch pixel (oy0 - 1, ox0 - 1) of frame b: may lie outside the frame (then only in-range chunks are loaded)
    const char *base = reinterpret_cast<const char *>(p.x) + ((((long long)b * p.H + (oy0 - 1)) * p.W + (ox0 - 1)) * p.x_ld) * 2;
#pragma unroll
    for (int j = 0; j < kXPre; ++j) {
      const int gy = oy0 - 1 + (xrc[j] >> 8), gx = ox0 - 1 + (xrc[j] & 255);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (xrc[j] >= 0 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
        v = *reinterpret_cast<const u32x4 *>(base + xoff[j]);
      pre[j] = v;
    }
  };
  auto x_store = [&](const u32x4 (&pre)[kXPre]) {
#pragma unroll
    for (int j = 0; j < kXPre; ++j)
      if (xrc[j] >= 0) *reinterpret_cast<u32x4 *>(xt + xlds[j]) = pre[j];
  };

  int tile = blockIdx.x;
  if (tile < p.n_tiles) {
    u32x4 pre[kXPre];
    x_fetch(tile, pre);
    x_store(pre);
  }
  for (; tile < p.n_tiles; tile += gridDim.x) {
    int b, oy0, ox0;
    tile_origin(tile, b, oy0, ox0);
    __syncthreads();   // B1: x patch (and, first time, the weights) in LDS; image / staging region free

    // ---- phase A: 1x1 conv over the patch ------------------------------------------------------------
#pragma unroll
    for (int fs = 0; fs < kAFr; ++fs) {
      if (wave + fs * (kThreads / 64) >= kRFrag) break;                   // wave-uniform
      const u32x4 xa = *reinterpret_cast<const u32x4 *>(xt + a_rd[fs]);
      const u32x4 xb = *reinterpret_cast<const u32x4 *>(xt + (a_rd[fs] ^ 64));   // K half 1: chunk index + 4 under the XOR swizzle
      const bool inside = (unsigned)(oy0 - 1 + (a_rc[fs] >> 8)) < (unsigned)p.H && (unsigned)(ox0 - 1 + (a_rc[fs] & 255)) < (unsigned)p.W;
      u32x2 o[2];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        a = H16<T>::mfma(__builtin_bit_cast(V8, w2f[ni][0]), __builtin_bit_cast(V8, xa), a);
        a = H16<T>::mfma(__builtin_bit_cast(V8, w2f[ni][1]), __builtin_bit_cast(V8, xb), a);
        o[ni] = bn_leaky_pack4<T>(a, sc2[ni], bi2[ni]);
        if (!inside) o[ni] = u32x2{0u, 0u};                              // the 3x3's zero padding
      }
      // channels 8 fq .. 8 fq + 7 of pixel q: one 16-byte write
      if (a_wr[fs] >= 0) *reinterpret_cast<u32x4 *>(yt + a_wr[fs]) = u32x4{o[0][0], o[0][1], o[1][0], o[1][1]};
    }
    const int next_tile = tile + gridDim.x;
    u32x4 pre[kXPre];
    if (next_tile < p.n_tiles) x_fetch(next_tile, pre);
    __syncthreads();   // B2: image complete

    // ---- phase B: 3x3 stride-1 conv from the image -----------------------------------------------------
    f32x4 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) acc[mi][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      u32x4 xf[2], wf[4];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int q = (2 * wave + mi + ky) * kRP + fr + kx;
        xf[mi] = *reinterpret_cast<const u32x4 *>(yt + q * kYPitch + fq * 16);
      }
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
        wf[nf] = *reinterpret_cast<const u32x4 *>(w3s + (nf * 16 + fr) * kW1Pitch + tap * 64 + fq * 16);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
          acc[mi][nf] = H16<T>::mfma(__builtin_bit_cast(V8, wf[nf]),
                                                                __builtin_bit_cast(V8, xf[mi]), acc[mi][nf]);
    }
    __syncthreads();   // B3: nobody reads the image any more

    // ---- phase C: bn + leaky, + shortcut operand from the x patch, -> bf16 -> staging ----------------------
#pragma unroll
    for (int k = 0; k < 2; ++k) {                                        // fragment pair k: channels 32 k + 8 fq .. + 7
      f32x4 s3[2], b3[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        s3[h] = *reinterpret_cast<const f32x4 *>(p.sc3 + k * 32 + fq * 8 + h * 4);
        b3[h] = *reinterpret_cast<const f32x4 *>(p.bi3 + k * 32 + fq * 8 + h * 4);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int oyl = 2 * wave + mi;
        const int pp = (oyl + 1) * kRP + fr + 1;                          // this output pixel inside the patch
        const u32x4 xr = *reinterpret_cast<const u32x4 *>(xt + pp * 128 + (((k * 4 + fq) ^ (pp & 7)) << 4));
        float v[8];
        y3_bn_leaky8(v, acc[mi][2 * k], acc[mi][2 * k + 1], s3[0], s3[1], b3[0], b3[1], true);   // packed arithmetic
        y3_add8<T>(v, xr);
        const int px = oyl * kRT + fr;
        *reinterpret_cast<u32x4 *>(yt + px * 128 + (((k * 4 + fq) ^ (px & 7)) << 4)) = y3_pack8<T>(v);
      }
    }
    __syncthreads();   // B4: staging complete; the x patch is dead
    {
      char *obase = reinterpret_cast<char *>(p.out) + ((((long long)b * p.H + oy0) * p.W + ox0) * p.out_ld) * 2;   // uniform
#pragma unroll
      for (int j = 0; j < kOutPer; ++j) {
        const int oy = oy0 + (opx[j] >> 4), ox = ox0 + (opx[j] & 15);
        if (oy < p.H && ox < p.W) *reinterpret_cast<u32x4 *>(obase + ooff[j]) = *reinterpret_cast<const u32x4 *>(yt + olds[j]);
      }
    }
    if (next_tile < p.n_tiles) x_store(pre);
  }
}

}  // namespace


// op0: the MFMA stem conv (uint8 frames, 3 -> 32, bf16 out); op1: 3x3 stride-2 conv 32 -> 64 reading ONLY op0's output
bool y3_conv_fused_stem_s2_supported(const y3_op &op0, const y3_op &op1) {
  if (!y3_opt().fuse_stem) return false;
  if (!y3_conv_stem_mfma_supported(op0) || op0.out_c != 32 || (op0.flags & Y3_F_RESIDUAL)) return false;
  if (!(op0.flags & Y3_F_LEAKY) || !(op1.flags & Y3_F_LEAKY)) return false;   // the kernel hard-wires LeakyReLU(0.1)
  if (op1.kind != Y3_OP_CONV || op1.dtype != op0.dtype || op1.ksize != 3 || op1.stride != 2 || op1.pad != 1) return false;
  if (op1.in_c != 32 || op1.out_c != 64 || op1.out_ld % 8 != 0 || op1.out_ld < 64) return false;
  if (op1.flags & (Y3_F_RESIDUAL | Y3_F_OUT_F32 | Y3_F_IN_NCHW_F32 | Y3_F_IN_NHWC_U8BGR | Y3_F_PLAN_INPUT)) return false;
  if (op1.d_in != op0.d_out || op1.in_h != op0.out_h || op1.in_w != op0.out_w || op1.batch != op0.batch) return false;
  if (op1.k_ld < 288 || op1.cout_pad < 64) return false;
  if (op1.out_h != (op1.in_h + 2 - 3) / 2 + 1 || op1.out_w != (op1.in_w + 2 - 3) / 2 + 1) return false;
  return true;
}

int y3_launch_conv_fused_stem_s2(const y3_op &op0, const y3_op &op1, const void *d_in, hipStream_t s,
                                 const char **kernel_name, bool dry_run) {
  *kernel_name = op0.dtype == Y3_F16 ? "conv_stem_s2_fused_u8_f16" : "conv_stem_s2_fused_u8_bf16";
  if (dry_run) return Y3_OK;
  return y3_by_dtype16(op0.dtype, [&](auto tag) {
    typedef decltype(tag) T;
    FusedArgs<T> a;
    a.in = static_cast<const unsigned char *>(d_in);
    a.H = op0.in_h; a.W = op0.in_w; a.batch = op0.batch;
    a.w0 = static_cast<const T *>(op0.d_weight);
    a.sc0 = op0.d_scale; a.bi0 = op0.d_bias; a.flags0 = op0.flags;
    a.w1 = static_cast<const T *>(op1.d_weight);
    a.k_ld1 = op1.k_ld;
    a.sc1 = op1.d_scale; a.bi1 = op1.d_bias; a.flags1 = op1.flags;
    a.out = static_cast<T *>(op1.d_out);
    a.out_ld = op1.out_ld; a.Ho = op1.out_h; a.Wo = op1.out_w;
    a.dbg = 0;
    static Y3DeviceOnce once;                          // (one per element type: the lambda is instantiated per T)
    int n_cu = 0;
    {
      const int rc = once.run([]() -> int {
        Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_stem_s2_fused_kernel<T>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
        Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_stem_s2_ws_kernel<T, 4>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kPLds));
        return Y3_OK;
      }, &n_cu);
      if (rc != Y3_OK) return rc;
    }
    const bool pipelined = y3_opt().fuse_stem != 2;                      // fuse_stem 2: the phase-by-phase kernel (A/B)
    a.tiles_x = y3_ceil_div(a.Wo, pipelined ? kPX : kTO);
    a.tiles_y = y3_ceil_div(a.Ho, pipelined ? kPY : kTO);
    a.n_tiles = a.tiles_x * a.tiles_y * a.batch;
    const int grid = a.n_tiles < n_cu ? a.n_tiles : n_cu;
    if (pipelined) Y3_LAUNCH((conv_stem_s2_ws_kernel<T, 4>), dim3(grid), dim3(1024), kPLds, s, a);
    else Y3_LAUNCH(conv_stem_s2_fused_kernel<T>, dim3(grid), dim3(kThreads), kLds, s, a);
    Y3_HIP_CHECK(hipGetLastError());
    return Y3_OK;
  });
}

// op0: 1x1 conv 64 -> 32 whose output only op1 reads; op1: 3x3 stride-1 conv 32 -> 64 with the shortcut operand == op0's
// input (one Darknet-53 residual block, bf16, LeakyReLU on both)
bool y3_conv_fused_resblock_supported(const y3_op &op0, const y3_op &op1) {
  if (!y3_opt().fuse_stem) return false;
  if (op0.kind != Y3_OP_CONV || op1.kind != Y3_OP_CONV || !y3_is16(op0.dtype) || op1.dtype != op0.dtype) return false;
  if (op0.ksize != 1 || op0.stride != 1 || op0.in_c != 64 || op0.out_c != 32) return false;
  if (op1.ksize != 3 || op1.stride != 1 || op1.pad != 1 || op1.in_c != 32 || op1.out_c != 64) return false;
  const uint32_t bad = Y3_F_OUT_F32 | Y3_F_IN_NCHW_F32 | Y3_F_IN_NHWC_U8BGR | Y3_F_PLAN_INPUT;
  if ((op0.flags & (bad | Y3_F_RESIDUAL)) || (op1.flags & bad)) return false;
  if (!(op0.flags & Y3_F_LEAKY) || !(op1.flags & Y3_F_LEAKY) || !(op1.flags & Y3_F_RESIDUAL)) return false;
  if (op1.d_in != op0.d_out || op1.d_res != op0.d_in || op1.res_ld != op0.in_ld) return false;
  if (op0.in_h != op1.in_h || op0.in_w != op1.in_w || op0.batch != op1.batch) return false;
  if (op0.out_h != op0.in_h || op0.out_w != op0.in_w || op1.out_h != op1.in_h || op1.out_w != op1.in_w) return false;
  if (op0.in_ld % 8 != 0 || op1.out_ld % 8 != 0 || op0.in_ld < 64 || op1.out_ld < 64) return false;
  if (op0.k_ld < 64 || op1.k_ld < 288 || op0.cout_pad < 32 || op1.cout_pad < 64) return false;
  return true;
}

int y3_launch_conv_fused_resblock(const y3_op &op0, const y3_op &op1, hipStream_t s, const char **kernel_name,
                                  bool dry_run) {
  *kernel_name = Y3_KNAME(op0.dtype, "conv_resblock_fused_", "_64_32_64");
  if (dry_run) return Y3_OK;
  return y3_by_dtype16(op0.dtype, [&](auto tag) {
    typedef decltype(tag) T;
    ResArgs<T> a;
    a.x = static_cast<const T *>(op0.d_in);
    a.H = op0.in_h; a.W = op0.in_w; a.batch = op0.batch; a.x_ld = op0.in_ld;
    a.w2 = static_cast<const T *>(op0.d_weight); a.k_ld2 = op0.k_ld; a.sc2 = op0.d_scale; a.bi2 = op0.d_bias;
    a.w3 = static_cast<const T *>(op1.d_weight); a.k_ld3 = op1.k_ld; a.sc3 = op1.d_scale; a.bi3 = op1.d_bias;
    a.out = static_cast<T *>(op1.d_out);
    a.out_ld = op1.out_ld;
    a.tiles_x = y3_ceil_div(a.W, kRT);
    a.tiles_y = y3_ceil_div(a.H, kRT);
    a.n_tiles = a.tiles_x * a.tiles_y * a.batch;
    static Y3DeviceOnce once;
    int n_cu = 0;
    {
      const int rc = once.run([]() -> int {
        Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_resblock_fused_kernel<T>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kRLds));
        return Y3_OK;
      }, &n_cu);
      if (rc != Y3_OK) return rc;
    }
    const int grid = a.n_tiles < n_cu ? a.n_tiles : n_cu;
    Y3_LAUNCH(conv_resblock_fused_kernel<T>, dim3(grid), dim3(kThreads), kRLds, s, a);
    Y3_HIP_CHECK(hipGetLastError());
    return Y3_OK;
  });
}

Y3_STAMP_READER(y3_debug_stamps_fused)
