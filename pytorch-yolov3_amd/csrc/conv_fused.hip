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
// Per tile (persistent workgroups, 8 waves, one per CU):
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

struct FusedArgs {
  const unsigned char *in;   // (B, H, W, 3) uint8 BGR
  int H, W, batch;
  const bf16_t *w0;          // stem weights [32][32]: k = ky*9 + kx*3 + byte channel, zero padded
  const float *sc0, *bi0;
  uint32_t flags0;
  const bf16_t *w1;          // second conv [>= 64][k_ld1], k = (ky*3 + kx)*32 + ci
  int k_ld1;
  const float *sc1, *bi1;
  uint32_t flags1;
  bf16_t *out;
  int out_ld, Ho, Wo;
  int tiles_x, tiles_y, n_tiles;
  int dbg;   // diagnostic: bit 0 skip phase 1, bit 1 skip phase 2, bit 2 skip phase 3 stores, bit 3 skip patch prefetch
};

// acc * scale + bias -> LeakyReLU(0.1) -> bf16, four channels at a time; written with 2-wide vectors so that hipcc
// emits v_pk_fma_f32 / v_pk_mul_f32 (the epilogues are VALU-bound: 35 k stem values per tile)
__device__ __forceinline__ u32x2 bn_leaky_bf16x4(const f32x4 &a, const f32x4 &sc, const f32x4 &bi) {
  const f32x2 lo = f32x2{a[0], a[1]} * f32x2{sc[0], sc[1]} + f32x2{bi[0], bi[1]};
  const f32x2 hi = f32x2{a[2], a[3]} * f32x2{sc[2], sc[3]} + f32x2{bi[2], bi[3]};
  const f32x2 tl = lo * Y3_LEAKY_SLOPE, th = hi * Y3_LEAKY_SLOPE;
  const bf16x4 o = {(bf16_t)y3_vmax(lo[0], tl[0]), (bf16_t)y3_vmax(lo[1], tl[1]), (bf16_t)y3_vmax(hi[0], th[0]),
                    (bf16_t)y3_vmax(hi[1], th[1])};
  return __builtin_bit_cast(u32x2, o);
}

__global__ __launch_bounds__(kThreads) void conv_stem_s2_fused_kernel(FusedArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t *in_tile = reinterpret_cast<bf16_t *>(smem);                 // [2][kIR][kInPitch]
  char *stem = smem + 2 * kInBytes;                                    // [kNStem][kStemPitch]
  char *w1s = stem + kNStem * kStemPitch;                              // [64][kW1Pitch]
  bf16_t *lut = reinterpret_cast<bf16_t *>(w1s + 64 * kW1Pitch);       // lut[v] = bf16(v / 255.0f): one IEEE division per
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
  bf16x8 w0a[2], w0b[2];
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = st * 32 + fq * 8 + j;
      const int ky = k / 12, rem = k - ky * 12, kx = rem >> 2, c = rem & 3;
      const bool live = k < 36 && c < 3;
      const int kold = live ? ky * 9 + kx * 3 + c : 0;
      const bf16_t za = p.w0[(0 + fr) * 32 + kold], zb = p.w0[(16 + fr) * 32 + kold];
      w0a[st][j] = live ? za : (bf16_t)0.f;
      w0b[st][j] = live ? zb : (bf16_t)0.f;
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

  // bytes of the frame patch of `tile` that this thread converts (kPre strided elements)
  auto patch_fetch = [&](int tile, unsigned char (&pre)[kPre]) {
    int t = tile;
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int iy0 = 2 * ty * kTO - 2, ixb0 = (2 * tx * kTO - 2) * 3;
#pragma unroll
    for (int j = 0; j < kPre; ++j) {
      const int i = tid + j * kThreads;
      const int r = i / (kIR * 3), cb = i - r * (kIR * 3);
      const int iy = iy0 + r, ixb = ixb0 + cb;
      unsigned char v = 0;
      if (i < kPatchElems && (unsigned)iy < (unsigned)p.H && ixb >= 0 && ixb < p.W * 3)
        v = p.in[((long long)b * p.H + iy) * p.W * 3 + ixb];
      pre[j] = v;
    }
  };
  auto patch_store = [&](int buf, const unsigned char (&pre)[kPre]) {
    bf16_t *dst = in_tile + buf * (kIR * kInPitch);
#pragma unroll
    for (int j = 0; j < kPre; ++j) {
      const int i = tid + j * kThreads;
      const int r = i / (kIR * 3), cb = i - r * (kIR * 3);
      if (i < kPatchElems) dst[r * kInPitch + (cb / 3) * 4 + (cb % 3)] = lut[pre[j]];
    }
  };

  for (int i = tid; i < 2 * kIR * kInPitch / 2; i += kThreads) reinterpret_cast<uint32_t *>(in_tile)[i] = 0u;
  if (tid < 256) lut[tid] = (bf16_t)((float)tid / 255.0f);
  __syncthreads();   // the fourth channel of every patch pixel stays zero from here on
  int tile = blockIdx.x;
  int buf = 0;
  if (tile < p.n_tiles) {
    unsigned char pre[kPre];
    patch_fetch(tile, pre);
    patch_store(0, pre);
  }
  for (; tile < p.n_tiles; tile += gridDim.x) {
    int t = tile;
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int oy0 = ty * kTO, ox0 = tx * kTO;
    __syncthreads();   // B1: this tile's input patch (and, first time, the weights) are in LDS; stem image is free

    // ---- phase 1: stem ------------------------------------------------------------------------------
    const bf16_t *patch = in_tile + buf * (kIR * kInPitch);
    for (int f = wave; f < ((p.dbg & 1) ? 0 : kNFrag); f += kThreads / 64) {
      const int q = f * 16 + fr;
      const int qc = q < kNStem ? q : kNStem - 1;
      const int sy = qc / kSR, sx = qc - sy * kSR;
      const bf16_t *base = patch + sy * kInPitch + sx * 4;
      const u32x2 lo = *reinterpret_cast<const u32x2 *>(base + off_lo);
      const u32x2 hi = *reinterpret_cast<const u32x2 *>(base + off_hi);
      const u32x2 s1 = *reinterpret_cast<const u32x2 *>(base + off_s1);
      const bf16x8 xf0 = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
      const bf16x8 xf1 = __builtin_bit_cast(bf16x8, fq == 0 ? u32x4{s1[0], s1[1], 0u, 0u} : u32x4{0u, 0u, 0u, 0u});
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0a[0], xf0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0b[0], xf0, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0a[1], xf1, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0b[1], xf1, acc1, 0, 0, 0);
      const int gy = 2 * oy0 - 1 + sy, gx = 2 * ox0 - 1 + sx;          // stem pixel in frame coordinates
      const bool inside = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      if (q < kNStem) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          u32x2 o = bn_leaky_bf16x4(ni ? acc1 : acc0, sc0[ni], bi0[ni]);
          if (!inside) o = u32x2{0u, 0u};                               // the second conv's zero padding
          *reinterpret_cast<u32x2 *>(stem + q * kStemPitch + (ni * 16 + cq) * 2) = o;
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
          acc[mi][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[nf]),
                                                                __builtin_bit_cast(bf16x8, xf[mi]), acc[mi][nf], 0, 0, 0);
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
            bn_leaky_bf16x4(acc[mi][nf], s1, b1);
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

struct ResArgs {
  const bf16_t *x;
  int H, W, batch, x_ld;
  const bf16_t *w2; int k_ld2; const float *sc2, *bi2;   // 1x1: [>= 32][k_ld2], k = ci
  const bf16_t *w3; int k_ld3; const float *sc3, *bi3;   // 3x3: [>= 64][k_ld3], k = (ky*3 + kx)*32 + ci
  bf16_t *out;
  int out_ld;
  int tiles_x, tiles_y, n_tiles;
};

__global__ __launch_bounds__(kThreads) void conv_resblock_fused_kernel(ResArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *xt = smem;                        // [kRNP][128], 16-byte chunks XOR-swizzled with (pixel & 7)
  char *yt = smem + kXBytes;              // [kRNP][kYPitch]; later the staging tile
  char *w3s = yt + kYBytes;               // [64][kW1Pitch]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4, cq = fq * 4;

  for (int i = tid; i < 64 * 36; i += kThreads) {
    const int co = i / 36, ch = i - co * 36;
    *reinterpret_cast<u32x4 *>(w3s + co * kW1Pitch + ch * 16) =
        *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(p.w3) + ((long long)co * p.k_ld3) * 2 + ch * 16);
  }
  // 1x1 weights: A fragments [co-frag][K step], lane (co = fr, k = ks*32 + fq*8 ..)
  u32x4 w2f[2][2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      w2f[ni][ks] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(p.w2) +
                                                    ((long long)(ni * 16 + fr) * p.k_ld2 + ks * 32 + fq * 8) * 2);
  f32x4 sc2[2], bi2[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    sc2[ni] = *reinterpret_cast<const f32x4 *>(p.sc2 + ni * 16 + cq);
    bi2[ni] = *reinterpret_cast<const f32x4 *>(p.bi2 + ni * 16 + cq);
  }

  auto tile_origin = [&](int tile, int &b, int &oy0, int &ox0) {
    const int tx = tile % p.tiles_x;
    tile /= p.tiles_x;
    oy0 = (tile % p.tiles_y) * kRT;
    b = tile / p.tiles_y;
    ox0 = tx * kRT;
  };
  auto x_fetch = [&](int tile, u32x4 (&pre)[kXPre]) {
    int b, oy0, ox0;
    tile_origin(tile, b, oy0, ox0);
#pragma unroll
    for (int j = 0; j < kXPre; ++j) {
      const int i = tid + j * kThreads;
      const int px = i >> 3, c = i & 7;
      const int r = px / kRP, cc = px - r * kRP;
      const int gy = oy0 - 1 + r, gx = ox0 - 1 + cc;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (i < kRNP * 8 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
        v = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(p.x) +
                                             ((((long long)b * p.H + gy) * p.W + gx) * p.x_ld) * 2 + c * 16);
      pre[j] = v;
    }
  };
  auto x_store = [&](const u32x4 (&pre)[kXPre]) {
#pragma unroll
    for (int j = 0; j < kXPre; ++j) {
      const int i = tid + j * kThreads;
      const int px = i >> 3, c = i & 7;
      if (i < kRNP * 8) *reinterpret_cast<u32x4 *>(xt + px * 128 + ((c ^ (px & 7)) << 4)) = pre[j];
    }
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
    for (int f = wave; f < kRFrag; f += kThreads / 64) {
      const int q = f * 16 + fr;
      const int qc = q < kRNP ? q : kRNP - 1;
      const u32x4 xa = *reinterpret_cast<const u32x4 *>(xt + qc * 128 + (((0 + fq) ^ (qc & 7)) << 4));
      const u32x4 xb = *reinterpret_cast<const u32x4 *>(xt + qc * 128 + (((4 + fq) ^ (qc & 7)) << 4));
      const int r = qc / kRP, cc = qc - r * kRP;
      const bool inside = (unsigned)(oy0 - 1 + r) < (unsigned)p.H && (unsigned)(ox0 - 1 + cc) < (unsigned)p.W;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2f[ni][0]), __builtin_bit_cast(bf16x8, xa), a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2f[ni][1]), __builtin_bit_cast(bf16x8, xb), a, 0, 0, 0);
        u32x2 o = bn_leaky_bf16x4(a, sc2[ni], bi2[ni]);
        if (!inside) o = u32x2{0u, 0u};                                  // the 3x3's zero padding
        if (q < kRNP) *reinterpret_cast<u32x2 *>(yt + q * kYPitch + (ni * 16 + cq) * 2) = o;
      }
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
          acc[mi][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[nf]),
                                                                __builtin_bit_cast(bf16x8, xf[mi]), acc[mi][nf], 0, 0, 0);
    }
    __syncthreads();   // B3: nobody reads the image any more

    // ---- phase C: bn + leaky, + shortcut operand from the x patch, -> bf16 -> staging ----------------------
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      const f32x4 s3 = *reinterpret_cast<const f32x4 *>(p.sc3 + nf * 16 + cq);
      const f32x4 b3 = *reinterpret_cast<const f32x4 *>(p.bi3 + nf * 16 + cq);
      const int co = nf * 16 + cq;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int oyl = 2 * wave + mi;
        const int pp = (oyl + 1) * kRP + fr + 1;                          // this output pixel inside the patch
        const u32x2 xr = *reinterpret_cast<const u32x2 *>(xt + pp * 128 + (((co >> 3) ^ (pp & 7)) << 4) + (co & 7) * 2);
        const bf16x4 xv = __builtin_bit_cast(bf16x4, xr);
        const f32x4 a = acc[mi][nf];
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = a[r] * s3[r] + b3[r];
          v = y3_vmax(v, Y3_LEAKY_SLOPE * v);
          o[r] = (bf16_t)(v + (float)xv[r]);
        }
        const int px = oyl * kRT + fr;
        *reinterpret_cast<bf16x4 *>(yt + px * 128 + (((co >> 3) ^ (px & 7)) << 4) + (co & 7) * 2) = o;
      }
    }
    __syncthreads();   // B4: staging complete; the x patch is dead
#pragma unroll
    for (int j = 0; j < kRT * kRT * 8 / kThreads; ++j) {
      const int i = tid + j * kThreads;
      const int px = i >> 3, ch = i & 7;
      const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
      if (oy < p.H && ox < p.W) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(yt + px * 128 + ((ch ^ (px & 7)) << 4));
        *reinterpret_cast<u32x4 *>(p.out + (((long long)b * p.H + oy) * p.W + ox) * p.out_ld + ch * 8) = v;
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
  if (op1.kind != Y3_OP_CONV || op1.dtype != Y3_BF16 || op1.ksize != 3 || op1.stride != 2 || op1.pad != 1) return false;
  if (op1.in_c != 32 || op1.out_c != 64 || op1.out_ld % 8 != 0 || op1.out_ld < 64) return false;
  if (op1.flags & (Y3_F_RESIDUAL | Y3_F_OUT_F32 | Y3_F_IN_NCHW_F32 | Y3_F_IN_NHWC_U8BGR | Y3_F_PLAN_INPUT)) return false;
  if (op1.d_in != op0.d_out || op1.in_h != op0.out_h || op1.in_w != op0.out_w || op1.batch != op0.batch) return false;
  if (op1.k_ld < 288 || op1.cout_pad < 64) return false;
  if (op1.out_h != (op1.in_h + 2 - 3) / 2 + 1 || op1.out_w != (op1.in_w + 2 - 3) / 2 + 1) return false;
  return true;
}

int y3_launch_conv_fused_stem_s2(const y3_op &op0, const y3_op &op1, const void *d_in, hipStream_t s,
                                 const char **kernel_name, bool dry_run) {
  *kernel_name = "conv_stem_s2_fused_u8_bf16";
  if (dry_run) return Y3_OK;
  FusedArgs a;
  a.in = static_cast<const unsigned char *>(d_in);
  a.H = op0.in_h; a.W = op0.in_w; a.batch = op0.batch;
  a.w0 = static_cast<const bf16_t *>(op0.d_weight);
  a.sc0 = op0.d_scale; a.bi0 = op0.d_bias; a.flags0 = op0.flags;
  a.w1 = static_cast<const bf16_t *>(op1.d_weight);
  a.k_ld1 = op1.k_ld;
  a.sc1 = op1.d_scale; a.bi1 = op1.d_bias; a.flags1 = op1.flags;
  a.out = static_cast<bf16_t *>(op1.d_out);
  a.out_ld = op1.out_ld; a.Ho = op1.out_h; a.Wo = op1.out_w;
  a.tiles_x = y3_ceil_div(a.Wo, kTO);
  a.tiles_y = y3_ceil_div(a.Ho, kTO);
  a.n_tiles = a.tiles_x * a.tiles_y * a.batch;
  a.dbg = 0;
  static Y3DeviceOnce once;
  int n_cu = 0;
  {
    const int rc = once.run([]() -> int {
      Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_stem_s2_fused_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
      return Y3_OK;
    }, &n_cu);
    if (rc != Y3_OK) return rc;
  }
  const int grid = a.n_tiles < n_cu ? a.n_tiles : n_cu;
  hipLaunchKernelGGL(conv_stem_s2_fused_kernel, dim3(grid), dim3(kThreads), kLds, s, a);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}

// op0: 1x1 conv 64 -> 32 whose output only op1 reads; op1: 3x3 stride-1 conv 32 -> 64 with the shortcut operand == op0's
// input (one Darknet-53 residual block, bf16, LeakyReLU on both)
bool y3_conv_fused_resblock_supported(const y3_op &op0, const y3_op &op1) {
  if (!y3_opt().fuse_stem) return false;
  if (op0.kind != Y3_OP_CONV || op1.kind != Y3_OP_CONV || op0.dtype != Y3_BF16 || op1.dtype != Y3_BF16) return false;
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
  *kernel_name = "conv_resblock_fused_bf16_64_32_64";
  if (dry_run) return Y3_OK;
  ResArgs a;
  a.x = static_cast<const bf16_t *>(op0.d_in);
  a.H = op0.in_h; a.W = op0.in_w; a.batch = op0.batch; a.x_ld = op0.in_ld;
  a.w2 = static_cast<const bf16_t *>(op0.d_weight); a.k_ld2 = op0.k_ld; a.sc2 = op0.d_scale; a.bi2 = op0.d_bias;
  a.w3 = static_cast<const bf16_t *>(op1.d_weight); a.k_ld3 = op1.k_ld; a.sc3 = op1.d_scale; a.bi3 = op1.d_bias;
  a.out = static_cast<bf16_t *>(op1.d_out);
  a.out_ld = op1.out_ld;
  a.tiles_x = y3_ceil_div(a.W, kRT);
  a.tiles_y = y3_ceil_div(a.H, kRT);
  a.n_tiles = a.tiles_x * a.tiles_y * a.batch;
  static Y3DeviceOnce once;
  int n_cu = 0;
  {
    const int rc = once.run([]() -> int {
      Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_resblock_fused_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kRLds));
      return Y3_OK;
    }, &n_cu);
    if (rc != Y3_OK) return rc;
  }
  const int grid = a.n_tiles < n_cu ? a.n_tiles : n_cu;
  hipLaunchKernelGGL(conv_resblock_fused_kernel, dim3(grid), dim3(kThreads), kRLds, s, a);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}
