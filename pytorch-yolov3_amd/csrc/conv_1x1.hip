// 1x1 convolution with LDS-resident weights (gfx950, bf16, NHWC): persistent workgroups, streamed activations.
//
// Same contract as conv_igemm.hip for the layers it takes (1x1, stride 1, conv -> scale/bias -> LeakyReLU, bf16 in and
// out, no shortcut operand): the "bottleneck" convs of Darknet-53's residual blocks
// (/root/reference/yolov3/darknet.py:244-257; models/yolov3.cfg: 256 -> 128 at 76^2, 512 -> 256 at 38^2, ten each).
//
// Why: these layers are a GEMM with a short K (4 or 8 K-tiles of 64 channels) over a long M.  On the tiled implicit
// GEMM every 128 x 128 tile is its own workgroup that first waits for its operands, runs 4 K-steps with one step of
// prefetch, parks and writes its tile: 23 us per launch at 76^2 and 18 us at 38^2 for ~6 us of HBM traffic and ~3 us of
// MFMA work (profiles/r02e_per_op.txt) -- latency, not bandwidth.  Here a workgroup loads ITS weight panel
// (BN output channels x all K, <= 64 KiB) into LDS once and then streams pixel tiles through a 4-slot ring without ever
// stopping at a tile boundary: the loader waves run NS - 1 K-tiles ahead of the MFMA waves across tiles, so a tile's
// first MFMA never waits for a cold prologue, and the weights are read from HBM / L2 once per workgroup instead of once
// per tile.  The epilogue applies scale / bias / LeakyReLU in registers, parks the tile as bf16 (32 KiB, 16-byte slots
// XOR-swizzled with the pixel) and writes whole 16-byte NHWC chunks.
//
// Geometry: BM = 128 pixels, BN = 128 (K <= 256) or 64 (K <= 512) channels, 4 MFMA waves (2 x 2, wave tile 64 x BN/2)
// + 4 loader waves, LDS = K*BN*2 (weights) + 128*BN*2 (park) + 4 x 16 KiB (ring) <= 160 KiB, one workgroup per CU.
// Workgroup b owns channel tile b % n_tiles and the pixel tiles (b / n_tiles) + j * (grid / n_tiles).
#include <utility>

#include "common.h"
#include "decode_core.h"

namespace {

struct WresArgs {
  const char *in;
  const char *wgt;
  const float *scale;
  const float *bias;
  char *out;
  const char *zero;
  int M, Cin, in_ld, Cout, out_ld, k_ld;
  int n_kt;        // Cin / 64
  int n_tiles;     // Cout / BN
  int m_tiles;     // ceil(M / 128)
  uint32_t flags;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int N>
__device__ __forceinline__ void wres_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int BN>
__global__ __launch_bounds__(512, 2) void conv1x1_wres_kernel(WresArgs p) {
  constexpr int BM = 128, NS = 4;
  constexpr int NC = 256;                            // consumer threads (== loader threads)
  constexpr int TN = BN / 2;                         // channels per MFMA wave
  constexpr int MI = 4, NI = TN / 16;
  constexpr int A_CH = BM / 32;                      // LDS-DMA pieces per loader thread and K-tile (32 rows per pass)
  constexpr int SLOT = BM * 128;                     // one K-tile of activations: 128 pixels x 64 channels
  constexpr int OCT = BN / 8;                        // 16-byte chunks per output row
  constexpr int WR = BM * OCT / NC;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *sW = smem;                                   // [n_kt][BN][128 B]
  char *sP = smem + p.n_kt * BN * 128;               // [BM][BN * 2 B], 16-byte slots swizzled with the pixel
  char *sA = sP + BM * BN * 2;                       // [NS][BM][128 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= NC / 64;

  const int nt = blockIdx.x % p.n_tiles;             // this workgroup's channel tile, fixed: its weights stay in LDS
  const int n0 = nt * BN;
  const int mstride = gridDim.x / p.n_tiles;
  const int mfirst = blockIdx.x / p.n_tiles;
  const int my_tiles = mfirst < p.m_tiles ? (p.m_tiles - mfirst + mstride - 1) / mstride : 0;
  const int total = my_tiles * p.n_kt;               // K-tiles of activations this workgroup consumes

  if (loader) {
    __builtin_amdgcn_s_setprio(3);
    const int ltid = tid - NC;
    const int lwave = wave - NC / 64;
    const int slot = ltid & 7;
    const int row0 = ltid >> 3;                      // 0..31
    const int kc = slot ^ (row0 & 7);
    // ---- weights: n_kt x BN rows of 128 bytes, K-tile major
    for (int r = row0; r < p.n_kt * BN; r += 32) {
      const int kt = r / BN, j = r - kt * BN;
      // LDS row j holds output channel y3_pair_perm(j) (common.h): after the MFMAs a lane's accumulators of a fragment
      // pair are eight consecutive channels of its pixel -- one 16-byte park write.  The K loop reads the same LDS rows.
      const int co = y3_pair_perm(j);
      const char *src = p.wgt + ((long long)(n0 + co) * p.k_ld + kt * 64) * 2 + kc * 16;
      __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(sW + (r - row0) * 128 + lwave * 1024), 16, 0, 0);
    }
    // ---- activations: K-tile s of the stream = K-tile (s % n_kt) of pixel tile mfirst + (s / n_kt) * mstride
    int s_tile = 0, s_kt = 0;                        // position of the NEXT K-tile to issue
    auto issue = [&](int stage) {
      const int m0 = (mfirst + s_tile * mstride) * BM;
      const bool live = s_tile < my_tiles;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        const int m = m0 + row0 + 32 * i;
        const char *src = (live && m < p.M) ? p.in + ((long long)m * p.in_ld + s_kt * 64) * 2 + kc * 16 : p.zero;
        __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(sA + stage * SLOT + i * 4096 + lwave * 1024), 16, 0, 0);
      }
      if (++s_kt == p.n_kt) { s_kt = 0; ++s_tile; }
    };
    int stage = 0;
    for (int t = 0; t < NS - 1; ++t) {               // beyond the end of the stream the pieces read the zero page
      issue(stage);
      stage = stage + 1 == NS ? 0 : stage + 1;
    }
    int kt = 0;
    for (int s = 0; s < total; ++s) {
      wres_wait_vmcnt<(NS - 2) * A_CH>();            // K-tile s (and the weights before it) landed; NS - 2 tiles in flight
      __builtin_amdgcn_s_barrier();                  // B(s): tile s visible, slot of tile s - 1 free
      issue(stage);
      stage = stage + 1 == NS ? 0 : stage + 1;
      if (++kt == p.n_kt) {
        kt = 0;
        __builtin_amdgcn_s_barrier();                // P: the MFMA waves have parked the finished tile
      }
    }
    wres_wait_vmcnt<0>();
    return;
  }

  // ---------------- MFMA waves ----------------
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const bool leaky = p.flags & Y3_F_LEAKY;
  const float slope = leaky ? Y3_LEAKY_SLOPE : 1.0f;
  f32x4 sc[NI], bi[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int c = n0 + wn * TN + (ni >> 1) * 32 + fq * 8 + (ni & 1) * 4;   // channels of acc[.][ni] (row permutation above)
    sc[ni] = *reinterpret_cast<const f32x4 *>(p.scale + c);
    bi[ni] = *reinterpret_cast<const f32x4 *>(p.bias + c);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of this wave is in flight inside the loop
  int stage = 0;
  for (int j = 0; j < my_tiles; ++j) {
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < p.n_kt; ++kt) {
      __builtin_amdgcn_s_barrier();                  // B(s)
      const char *a = sA + stage * SLOT;
      const char *w = sW + kt * BN * 128;
      stage = stage + 1 == NS ? 0 : stage + 1;
      u32x4 xf[2][MI], wf[2][NI];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int row = wn * TN + ni * 16 + fr;
          wf[g][ni] = *reinterpret_cast<const u32x4 *>(w + row * 128 + (((g * 4 + fq) ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int row = wm * 64 + mi * 16 + fr;
          xf[g][mi] = *reinterpret_cast<const u32x4 *>(a + row * 128 + (((g * 4 + fq) ^ (row & 7)) << 4));
        }
      }
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = y3_mfma16<T>(wf[g][ni], xf[g][mi], acc[mi][ni]);
    }
    // ---- epilogue: scale / bias / LeakyReLU in registers (the arithmetic of y3_bn_leaky8), bf16, park, write out
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int pl = wm * 64 + mi * 16 + fr;
#pragma unroll
      for (int k = 0; k < NI / 2; ++k) {
        float o[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ni = 2 * k + h;
          const f32x2 t0 = f32x2{acc[mi][ni][0], acc[mi][ni][1]} * f32x2{sc[ni][0], sc[ni][1]} + f32x2{bi[ni][0], bi[ni][1]};
          const f32x2 t1 = f32x2{acc[mi][ni][2], acc[mi][ni][3]} * f32x2{sc[ni][2], sc[ni][3]} + f32x2{bi[ni][2], bi[ni][3]};
          const f32x2 s0 = t0 * slope, s1 = t1 * slope;
          o[4 * h + 0] = y3_vmax(t0[0], s0[0]);
          o[4 * h + 1] = y3_vmax(t0[1], s0[1]);
          o[4 * h + 2] = y3_vmax(t1[0], s1[0]);
          o[4 * h + 3] = y3_vmax(t1[1], s1[1]);
        }
        const int oc = wn * (TN / 8) + k * 4 + fq;                       // 16-byte chunk of the pixel's row
        *reinterpret_cast<u32x4 *>(sP + pl * (BN * 2) + ((oc ^ (pl & (OCT - 1))) << 4)) = y3_pack8<T>(o);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                    // P: the tile is parked
    const int m0 = (mfirst + j * mstride) * BM;
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      const int pl = tid / OCT + i * (NC / OCT);
      const int oc = tid % OCT;
      const int m = m0 + pl;
      const u32x4 v = *reinterpret_cast<const u32x4 *>(sP + pl * (BN * 2) + ((oc ^ (pl & (OCT - 1))) << 4));
      if (m < p.M) *reinterpret_cast<u32x4 *>(p.out + ((long long)m * p.out_ld + n0 + oc * 8) * 2) = v;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);              // park reads done before this wave can reach the next tile's park
  }
}

// ------------------------------------------------------------------------------------------------
// Direct-weights 1x1 kernel (round 6; 16-bit storage modes): for the SHORT-K, SMALL-MAP bottleneck layers (512 -> 256 at 38^2,
// 1024 -> 512 at 19^2, 768 -> 256 after the route) where the tiled implicit GEMM is latency-bound: 8 or 16 K-steps with ONE
// step of prefetch means every step waits out an L2 round trip (16.8 us per launch for 2.7 us of L2 traffic and 2.8 us of
// MFMA work, profiles/r06_*).  Here NOTHING in the K loop waits on a barrier or on a cold load:
//   * a workgroup owns BM (96 or 48) pixels x 256 channels = ONE tile per CU in one round (241 / 242 tiles at batch 16);
//   * its whole activation tile -- BM pixels x ALL Cin -- is staged in LDS in the prologue (<= 144 KiB, every LDS-DMA piece in
//     flight at once: one memory round trip instead of one per K-step);
//   * the weight fragments come straight from global memory / L2 into registers out of the FRAGMENT-ORDER copy of the
//     weights (conv_halo.hip: y3_conv_halo_dw_make_weights, rows y3_pair_perm'd), DEPTH K-steps ahead, as plain loads the
//     compiler counts itself (the loop is fully unrolled: NKT = Cin / 64 is a template parameter);
//   * eight waves, each BM pixels x 32 channels; no barrier after the prologue's; scale / bias / LeakyReLU / rounding in
//     registers, one 16-byte store per fragment pair (a lane holds eight consecutive channels of its pixel).
// Same K order as every other MFMA conv kernel (K ascending in steps of 32): same bits.
struct DwArgs {
  const char *in;
  const char *wgt;     // fragment order: block (channel block cb, K block kb) at ((cb * (k_ld / 32) + kb) << 10)
  const float *scale;
  const float *bias;
  char *out;
  const char *zero;
  int M, in_ld, out_ld, k_ld;
  int n_tiles;         // Cout / 256
  uint32_t flags;
  // detection-head form (HEAD = true): the YOLO decode of yolo_decode.hip runs on the workgroup's logit tile (see y3_head_decode_rows)
  int Ho, Wo, HoWo;
  uint32_t mul_hw, sh_hw, mul_w, sh_w;   // n / d == (umulhi(n, mul) + n) >> sh  (d = HoWo, Wo)
  float *y_bbox, *y_prob;
  long long *y_cls;
  int y_anchors, y_attr, y_row_offset, y_rows_total;
  float y_net_w, y_net_h, y_aw[8], y_ah[8];
};

template <int V>
struct StepC { static constexpr int value = V; };
template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F &&f) { (f(StepC<I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// s_waitcnt vmcnt(N) that NAMES the four registers it waits for (see dw_wait_vm in conv_halo.hip): the tie keeps the compiler
// from moving their uses above the wait and from re-using them while the load is in flight
template <int N>
__device__ __forceinline__ void dw1_wait_vm(u32x4 (&w)[2][2]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]) : "n"(N) : "memory");
}

// HEAD = true: the detection-head conv (bias, no activation, <= 256 logits per pixel = ONE channel tile) + YOLOLayer decode
// (/root/reference/yolov3/darknet.py:86-116) in one launch: after the K loop the float32 logits (sum * scale + bias, one fused rounding
// -- the arithmetic of the tiled head kernel, conv_igemm.hip) are parked in the LDS the activation tile no longer needs, rows of
// 260 floats, and decoded four lanes per box by the code every other decode path runs (decode_core.h): same bits.
template <typename T, int BM, int NKT, bool HEAD = false>
__global__ __launch_bounds__(512, 2) void conv1x1_dw_kernel(DwArgs p) {
  static_assert(sizeof(T) == 2 && (BM == 96 || BM == 48) && NKT % 2 == 0, "16-bit modes; 96- or 48-pixel tiles; Cin a multiple of 128");
  constexpr int MI = BM / 16, NI = 2;
  constexpr int RB = NKT * 128;                       // bytes of one pixel's Cin channels = LDS row pitch (a multiple of 256)
  constexpr int PIECES = BM * RB / 1024;              // 1-KiB LDS-DMA pieces of the tile
  constexpr int DEPTH = NKT < 4 ? NKT : 4;            // K-steps of weight fragments in flight (4 loads of 1 KiB per wave and step)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int tile = y3_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / p.n_tiles) * BM;             // channel tiles innermost: they share the activation tile in L2
  const int n0 = (tile % p.n_tiles) * 256;

  // ---- prologue: the whole activation tile.  A wave-instruction fills 1 KiB of consecutive LDS; the 16-byte chunk c of
  // pixel row r sits at chunk position c ^ (r & 15) of its row (every row starts on bank 0: the XOR spreads a fragment
  // read's sixteen rows over the sixteen 16-byte bank groups), applied on the SOURCE address of each lane.
#pragma unroll
  for (int j = 0; j < (PIECES + 7) / 8; ++j) {
    const int i = wave + 8 * j;
    if (i < PIECES) {
      const int o = i * 1024 + lane * 16;
      const int r = o / RB, cpos = (o - r * RB) >> 4;
      const int c = cpos ^ (r & 15);
      const long long m = (long long)m0 + r;
      const char *src = m < p.M ? p.in + (m * p.in_ld) * 2 + c * 16 : p.zero;
      __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(smem + i * 1024), 16, 0, 0);
    }
  }
  asm volatile("" ::: "memory");
  // ---- weight fragments of this wave's 32 channels (two 16-channel blocks), K block by K block: inline-asm loads with
  // hand-counted waits (left to the compiler the loads sink to just above their use: every K-step then waits out an L2 round trip)
  const uint32_t kblocks = (uint32_t)p.k_ld / 32u;
  const uint32_t w_voff = (uint32_t)lane * 16;
  const char *wb[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) wb[ni] = p.wgt + (((long long)((n0 + wave * 32) / 16 + ni) * kblocks) << 10);
  u32x4 wf[DEPTH][2][NI];
  auto load_w = [&](int kt, u32x4 (&w)[2][NI]) {     // 4 loads, in the order [kh][ni]
    const uint32_t voff = w_voff + ((uint32_t)kt << 11);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(w[0][ni]) : "v"(voff), "s"(wb[ni]));
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(w[1][ni]) : "v"(voff), "s"(wb[ni]));
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load_w(d, wf[d]);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (DEPTH - 1)) : "memory");   // the tile (this wave's pieces) and step 0's weights landed
  __builtin_amdgcn_s_barrier();                       // ... everyone's pieces: the only barrier of the kernel
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);       // the younger wave of each SIMD (see conv_halo_ws_kernel)

  typedef const __attribute__((address_space(3))) u32x4 lds_u32x4;
  const int a_base = (int)(size_t)(lds_void *)smem + fr * RB;
  const int swz = fr & 15;
  u32x4 xf[2][MI];
  auto read_x = [&](int kt, int kh, u32x4 (&x)[MI]) {
    const int c = ((kt * 8 + kh * 4 + fq) ^ swz) << 4;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) x[mi] = *reinterpret_cast<lds_u32x4 *>(a_base + mi * 16 * RB + c);
  };
  auto mma = [&](const u32x4 (&x)[MI], const u32x4 (&w)[NI]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = y3_mfma16<T>(w[ni], x[mi], acc[mi][ni]);
  };
  auto interleave = [&]() {                           // one fragment read of the NEXT K-half per two MFMAs of this one
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
  };
  read_x(0, 0, xf[0]);
  static_for<NKT>([&](auto ktc) {
    constexpr int kt = decltype(ktc)::value, slot = kt % DEPTH;
    // in flight behind step kt's four loads: those of steps kt + 1 .. min(kt + DEPTH, NKT) - 1
    constexpr int younger = 4 * ((kt + DEPTH < NKT ? kt + DEPTH : NKT) - kt - 1);
    dw1_wait_vm<younger>(wf[slot]);
    __builtin_amdgcn_sched_barrier(0);
    read_x(kt, 1, xf[1]);
    mma(xf[0], wf[slot][0]);
    interleave();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (kt + 1 < NKT) read_x(kt + 1, 0, xf[0]);
    mma(xf[1], wf[slot][1]);
    interleave();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (kt + DEPTH < NKT) load_w(kt + DEPTH, wf[slot]);
  });
  __builtin_amdgcn_s_setprio(0);

  // ---- epilogue in registers: lane (fr, fq) holds channels co .. co + 7 of pixels m0 + mi * 16 + fr
  const int co = n0 + wave * 32 + fq * 8;             // (y3_pair_perm'd weight rows)
  if constexpr (HEAD) {
    constexpr int LDL = 260;                          // 1040 bytes: consecutive pixels of a fragment shift by four banks
    const f32x4 hs_lo = *reinterpret_cast<const f32x4 *>(p.scale + co), hs_hi = *reinterpret_cast<const f32x4 *>(p.scale + co + 4);
    const f32x4 hb_lo = *reinterpret_cast<const f32x4 *>(p.bias + co), hb_hi = *reinterpret_cast<const f32x4 *>(p.bias + co + 4);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                     // nobody reads the activation tile any more: LDS holds the logits
    float *sL = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      f32x4 lo, hi;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        lo[r] = __builtin_fmaf(acc[mi][0][r], hs_lo[r], hb_lo[r]);
        hi[r] = __builtin_fmaf(acc[mi][1][r], hs_hi[r], hb_hi[r]);
      }
      float *row = sL + (mi * 16 + fr) * LDL + co;
      *reinterpret_cast<f32x4 *>(row) = lo;
      *reinterpret_cast<f32x4 *>(row + 4) = hi;
    }
    __syncthreads();
    y3_head_decode_rows<512, LDL>(p, sL, m0, BM, tid);
    return;
  }
  const f32x4 sc_lo = *reinterpret_cast<const f32x4 *>(p.scale + co), sc_hi = *reinterpret_cast<const f32x4 *>(p.scale + co + 4);
  const f32x4 bi_lo = *reinterpret_cast<const f32x4 *>(p.bias + co), bi_hi = *reinterpret_cast<const f32x4 *>(p.bias + co + 4);
  const bool leaky = p.flags & Y3_F_LEAKY;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m0 + mi * 16 + fr;
    float v[8];
    y3_bn_leaky8(v, acc[mi][0], acc[mi][1], sc_lo, sc_hi, bi_lo, bi_hi, leaky);
    if (m < p.M) *reinterpret_cast<u32x4 *>(p.out + ((long long)m * p.out_ld + co) * 2) = y3_pack8<T>(v);
  }
}

// tile height of the direct-weights 1x1 kernel for this op (96 / 48), or 0 when it does not take it
int dw1x1_bm(const y3_op &op) {
  if (op.kind != Y3_OP_CONV || !y3_is16(op.dtype) || op.ksize != 1 || op.stride != 1 || op.pad != 0) return 0;
  if (op.flags & (Y3_F_RESIDUAL | Y3_F_OUT_F32 | Y3_F_IN_NCHW_F32 | Y3_F_IN_NHWC_U8BGR | Y3_F_PLAN_INPUT)) return 0;
  if (op.out_c % 256 != 0 || op.in_ld % 8 != 0 || op.out_ld % 8 != 0 || op.k_ld % 32 != 0 || op.k_ld < op.in_c || op.cout_pad % 32 != 0) return 0;
  const int nkt = op.in_c / 64;
  if (op.in_c % 128 != 0 || !(nkt == 4 || nkt == 6 || nkt == 8 || nkt == 12 || nkt == 16)) return 0;
  const int M = op.batch * op.in_h * op.in_w, n_cu = y3_device_cus(), nt = op.out_c / 256;
  // one round of workgroups: 96-pixel tiles where they already give every CU (nearly) one, else 48-pixel tiles -- down to a
  // quarter of the CUs (512 -> 256 at 19^2 x 16 frames: 121 tiles, 5.8 against 8.1-8.9 us on the implicit GEMMs, whose eight
  // K-steps each wait out an L2 round trip whatever the grid; 38^2 x 8 frames 6.6 against 8.7-11.2: profiles/r06_conv1x1_dw.txt); smaller grids (one frame at a time) stay on the tiled kernels
  const long long t96 = (long long)y3_ceil_div(M, 96) * nt, t48 = (long long)y3_ceil_div(M, 48) * nt;
  if (op.in_c * 2 * 96 <= 144 * 1024 && t96 <= n_cu && 4 * t96 >= 3 * n_cu) return 96;
  if (op.in_c * 2 * 48 <= 144 * 1024 && t48 <= n_cu && 4 * t48 >= n_cu) return 48;
  return 0;
}

// tile height of the detection-head form for this head conv (96 / 48), or 0 when the tiled head kernel keeps it.  y3_options.fuse_head:
// 1 = 48 (measured at batch 16 / one frame, us per launch: 19^2 x 1024 13.8 / 10.9, 38^2 x 512 16.0 / 8.0, 76^2 x 256 32.9 / 6.7; 96-pixel
// tiles 13.8 (48: does not fit) / 15.8 / 38.1 and -- / 10.7 / 8.9; the tiled kernel 27.7 / 20.2 / 39.3 and 21.3 / 14.6 / 10.4:
// profiles/r06_head_dw.txt), 2 = never (the tiled kernel, A/B), 3 / 4 = 48 / 96 wherever the shape allows (A/B, tests)
int dw_head_bm(const y3_op &op) {
  const int mode = y3_opt().fuse_head;
  if (mode == 0 || mode == 2) return 0;
  if (op.kind != Y3_OP_CONV || !y3_is16(op.dtype) || op.ksize != 1 || op.stride != 1 || op.pad != 0 || !(op.flags & Y3_F_OUT_F32)) return 0;
  if (op.flags & (Y3_F_LEAKY | Y3_F_RESIDUAL | Y3_F_IN_NCHW_F32 | Y3_F_IN_NHWC_U8BGR | Y3_F_PLAN_INPUT)) return 0;
  if (op.out_c > 256 || op.cout_pad < 256 || op.cout_pad % 32 != 0 || op.in_ld % 8 != 0 || op.k_ld % 32 != 0 || op.k_ld < op.in_c) return 0;
  const int nkt = op.in_c / 64;
  if (op.in_c % 128 != 0 || !(nkt == 4 || nkt == 8 || nkt == 16)) return 0;
  const long long M = (long long)op.batch * op.in_h * op.in_w;
  if (M >= (1ll << 31)) return 0;
  const bool fits96 = nkt <= 8;                       // 96 pixels x 1024 channels would be 192 KiB
  return mode == 4 && fits96 ? 96 : 48;
}

int wres_bn(const y3_op &op) {
  if (op.in_c <= 256 && op.out_c % 128 == 0) return 128;
  if (op.in_c <= 512 && op.out_c % 64 == 0) return 64;
  return 0;
}

}  // namespace

// the layers the direct-weights 1x1 kernel takes: 16-bit, no shortcut, Cout a multiple of 256, Cin 256 .. 1024, and a map x
// batch that gives (nearly) every CU exactly one tile
bool y3_conv1x1_dw_pays(const y3_op &op) { return dw1x1_bm(op) != 0; }

int y3_launch_conv1x1_dw(const y3_op &op, const void *d_in, const void *d_zero, hipStream_t s, const char **kernel_name,
                         bool dry_run, const void *frag_w) {
  const int bm = dw1x1_bm(op);
  Y3_REQUIRE(bm != 0, "conv block %d: not a shape for the direct-weights 1x1 kernel", op.block_idx);
  *kernel_name = bm == 96 ? Y3_KNAME(op.dtype, "conv1x1_dw_", "_96x256") : Y3_KNAME(op.dtype, "conv1x1_dw_", "_48x256");
  if (dry_run) return Y3_OK;
  void *tmp = nullptr;
  if (!frag_w) {                                      // single-op calls without a shared copy: made here, stream-ordered
    Y3_HIP_CHECK(hipMallocAsync(&tmp, y3_conv_halo_dw_weight_bytes(op), s));
    const int rc = y3_conv_halo_dw_make_weights(op, tmp, s);
    if (rc != Y3_OK) { (void)hipFreeAsync(tmp, s); return rc; }
    frag_w = tmp;
  }
  DwArgs a;
  a.in = static_cast<const char *>(d_in);
  a.wgt = static_cast<const char *>(frag_w);
  a.scale = op.d_scale; a.bias = op.d_bias;
  a.out = static_cast<char *>(op.d_out);
  a.zero = static_cast<const char *>(d_zero);
  a.M = op.batch * op.in_h * op.in_w;
  a.in_ld = op.in_ld; a.out_ld = op.out_ld; a.k_ld = op.k_ld;
  a.n_tiles = op.out_c / 256;
  a.flags = op.flags;
  const int nkt = op.in_c / 64;
  const int rc = y3_by_dtype16(op.dtype, [&](auto tag) {
    typedef decltype(tag) T;
    static Y3DeviceOnce once;
    {
      const int rc1 = once.run([]() -> int {
#define Y3_DW1_ATTR(BM_, NKT_) Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_dw_kernel<T, BM_, NKT_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
        Y3_DW1_ATTR(96, 4); Y3_DW1_ATTR(96, 6); Y3_DW1_ATTR(96, 8); Y3_DW1_ATTR(96, 12);
        Y3_DW1_ATTR(48, 4); Y3_DW1_ATTR(48, 6); Y3_DW1_ATTR(48, 8); Y3_DW1_ATTR(48, 12); Y3_DW1_ATTR(48, 16);
#undef Y3_DW1_ATTR
        return Y3_OK;
      });
      if (rc1 != Y3_OK) return rc1;
    }
    const size_t lds = (size_t)bm * op.in_c * 2;
    const dim3 grid(y3_ceil_div(a.M, bm) * a.n_tiles);
#define Y3_DW1_GO(BM_, NKT_) Y3_LAUNCH((conv1x1_dw_kernel<T, BM_, NKT_>), grid, dim3(512), lds, s, a)
    if (bm == 96) {
      if (nkt == 4) Y3_DW1_GO(96, 4); else if (nkt == 6) Y3_DW1_GO(96, 6); else if (nkt == 8) Y3_DW1_GO(96, 8); else Y3_DW1_GO(96, 12);
    } else {
      if (nkt == 4) Y3_DW1_GO(48, 4); else if (nkt == 6) Y3_DW1_GO(48, 6); else if (nkt == 8) Y3_DW1_GO(48, 8);
      else if (nkt == 12) Y3_DW1_GO(48, 12); else Y3_DW1_GO(48, 16);
    }
#undef Y3_DW1_GO
    Y3_HIP_CHECK(hipGetLastError());
    return Y3_OK;
  });
  if (tmp) (void)hipFreeAsync(tmp, s);
  return rc;
}

// detection-head conv + YOLO decode on the direct-weights 1x1 kernel (op0: the head conv, op1: the Y3_OP_YOLO op reading it; the pair
// has passed y3_conv_head_decode_supported)
bool y3_conv_head_dw_fits(const y3_op &op0) { return dw_head_bm(op0) != 0; }

int y3_launch_conv_head_decode_dw(const y3_op &op0, const y3_op &op1, const void *d_zero, hipStream_t s, const char **kernel_name,
                                  bool dry_run, const void *frag_w) {
  const int bm = dw_head_bm(op0);
  Y3_REQUIRE(bm != 0, "conv block %d: not a shape for the direct-weights head kernel", op0.block_idx);
  *kernel_name = bm == 96 ? Y3_KNAME(op0.dtype, "conv_head_decode_dw_", "_96x256") : Y3_KNAME(op0.dtype, "conv_head_decode_dw_", "_48x256");
  if (dry_run) return Y3_OK;
  void *tmp = nullptr;
  if (!frag_w) {                                      // callers without a shared copy: made here, stream-ordered
    Y3_HIP_CHECK(hipMallocAsync(&tmp, y3_conv_halo_dw_weight_bytes(op0), s));
    const int rc = y3_conv_halo_dw_make_weights(op0, tmp, s);
    if (rc != Y3_OK) { (void)hipFreeAsync(tmp, s); return rc; }
    frag_w = tmp;
  }
  DwArgs a;
  a.in = static_cast<const char *>(op0.d_in);
  a.wgt = static_cast<const char *>(frag_w);
  a.scale = op0.d_scale; a.bias = op0.d_bias;
  a.out = nullptr;
  a.zero = static_cast<const char *>(d_zero);
  a.Ho = op0.out_h; a.Wo = op0.out_w; a.HoWo = op0.out_h * op0.out_w;
  a.M = op0.batch * a.HoWo;
  a.in_ld = op0.in_ld; a.out_ld = op0.out_ld; a.k_ld = op0.k_ld;
  a.n_tiles = 1;
  a.flags = op0.flags;
  y3_fast_div((uint32_t)a.HoWo, a.mul_hw, a.sh_hw);
  y3_fast_div((uint32_t)a.Wo, a.mul_w, a.sh_w);
  a.y_bbox = op1.d_bbox; a.y_prob = op1.d_prob; a.y_cls = reinterpret_cast<long long *>(op1.d_cls);
  a.y_anchors = op1.n_anchor; a.y_attr = op1.n_attr;
  a.y_row_offset = op1.row_offset; a.y_rows_total = op1.rows_total;
  a.y_net_w = op1.net_w; a.y_net_h = op1.net_h;
  for (int i = 0; i < 8; ++i) { a.y_aw[i] = op1.anchor_w[i]; a.y_ah[i] = op1.anchor_h[i]; }
  const int nkt = op0.in_c / 64;
  const int rc = y3_by_dtype16(op0.dtype, [&](auto tag) {
    typedef decltype(tag) T;
    static Y3DeviceOnce once;
    {
      const int rc1 = once.run([]() -> int {
#define Y3_DWH_ATTR(BM_, NKT_) Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_dw_kernel<T, BM_, NKT_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
        Y3_DWH_ATTR(96, 4); Y3_DWH_ATTR(96, 8); Y3_DWH_ATTR(48, 4); Y3_DWH_ATTR(48, 8); Y3_DWH_ATTR(48, 16);
#undef Y3_DWH_ATTR
        return Y3_OK;
      });
      if (rc1 != Y3_OK) return rc1;
    }
    const size_t tile = (size_t)bm * op0.in_c * 2, logits = (size_t)bm * 260 * 4;
    const size_t lds = tile > logits ? tile : logits;
    const dim3 grid(y3_ceil_div(a.M, bm));
#define Y3_DWH_GO(BM_, NKT_) Y3_LAUNCH((conv1x1_dw_kernel<T, BM_, NKT_, true>), grid, dim3(512), lds, s, a)
    if (bm == 96) {
      if (nkt == 4) Y3_DWH_GO(96, 4); else Y3_DWH_GO(96, 8);
    } else {
      if (nkt == 4) Y3_DWH_GO(48, 4); else if (nkt == 8) Y3_DWH_GO(48, 8); else Y3_DWH_GO(48, 16);
    }
#undef Y3_DWH_GO
    Y3_HIP_CHECK(hipGetLastError());
    return Y3_OK;
  });
  if (tmp) (void)hipFreeAsync(tmp, s);
  return rc;
}

// 1x1 stride-1 bf16 conv without shortcut operand whose weight panel fits LDS
bool y3_conv1x1_wres_supported(const y3_op &op) {
  if (op.kind != Y3_OP_CONV || !y3_is16(op.dtype) || op.ksize != 1 || op.stride != 1 || op.pad != 0) return false;
  if (op.flags & (Y3_F_RESIDUAL | Y3_F_OUT_F32 | Y3_F_IN_NCHW_F32 | Y3_F_IN_NHWC_U8BGR | Y3_F_PLAN_INPUT)) return false;
  if (op.in_c % 64 != 0 || op.in_c < 128 || op.in_ld % 8 != 0 || op.out_ld % 8 != 0 || op.k_ld < op.in_c) return false;
  if (((uintptr_t)op.d_in | (uintptr_t)op.d_out) % 16 != 0) return false;
  return wres_bn(op) != 0;
}

// ... where it measured faster than the tiled kernels (profiles/r02h_convbench_1x1.txt, batch 16): ONE channel tile (every
// workgroup streams the activations once: 256 -> 128 at 76^2 14.3 against 18.9 us, 128 -> 64 at 152^2 23.3 against 28.9 us
// = 6.1 TB/s) on a map large enough to keep every CU streaming at least two pixel tiles.  With several channel tiles
// (512 -> 256 at 38^2 as 4 x 64 channels, 384 -> 128 as 2 x 64) every tile's workgroups re-read the activations and the
// 64 x 32 wave tiles feed the matrix pipe worse: 17.9 against 14.2 us, 26.1 against 23.9 us.
bool y3_conv1x1_wres_pays(const y3_op &op) {
  const int bn = wres_bn(op);
  return bn != 0 && op.out_c == bn && (long long)y3_ceil_div(op.batch * op.in_h * op.in_w, 128) >= 512;
}

int y3_launch_conv1x1_wres(const y3_op &op, const void *d_in, const void *d_zero, hipStream_t s, const char **kernel_name,
                           bool dry_run) {
  Y3_REQUIRE(y3_conv1x1_wres_supported(op), "conv block %d: not a shape for the weights-resident 1x1 kernel", op.block_idx);
  const int bn = wres_bn(op);
  *kernel_name = bn == 128 ? Y3_KNAME(op.dtype, "conv1x1_wres_", "_128x128") : Y3_KNAME(op.dtype, "conv1x1_wres_", "_128x64");
  if (dry_run) return Y3_OK;
  WresArgs a;
  a.in = static_cast<const char *>(d_in);
  a.wgt = static_cast<const char *>(op.d_weight);
  a.scale = op.d_scale; a.bias = op.d_bias;
  a.out = static_cast<char *>(op.d_out);
  a.zero = static_cast<const char *>(d_zero);
  a.M = op.batch * op.in_h * op.in_w;
  a.Cin = op.in_c; a.in_ld = op.in_ld; a.Cout = op.out_c; a.out_ld = op.out_ld; a.k_ld = op.k_ld;
  a.n_kt = op.in_c / 64;
  a.n_tiles = op.out_c / bn;
  a.m_tiles = y3_ceil_div(a.M, 128);
  a.flags = op.flags;
  return y3_by_dtype16(op.dtype, [&](auto tag) {
    typedef decltype(tag) T;
    static Y3DeviceOnce once;
    int n_cu = 0;
    {
      const int rc = once.run([]() -> int {
        Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_wres_kernel<T, 128>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_wres_kernel<T, 64>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        return Y3_OK;
      }, &n_cu);
      if (rc != Y3_OK) return rc;
    }
    const size_t lds = (size_t)a.n_kt * bn * 128 + (size_t)128 * bn * 2 + (size_t)4 * 128 * 128;
    Y3_REQUIRE(lds <= 160 * 1024, "conv block %d: weight panel does not fit LDS", op.block_idx);
    // one workgroup per CU, a whole number of them per channel tile (at least one each: more channel tiles than CUs just
    // means more than one workgroup per CU in turn)
    int grid = a.n_tiles > n_cu ? a.n_tiles : n_cu - n_cu % a.n_tiles;
    const long long tiles = (long long)a.m_tiles * a.n_tiles;
    if (grid > tiles) grid = (int)tiles;
    if (bn == 128) Y3_LAUNCH((conv1x1_wres_kernel<T, 128>), dim3(grid), dim3(512), lds, s, a);
    else Y3_LAUNCH((conv1x1_wres_kernel<T, 64>), dim3(grid), dim3(512), lds, s, a);
    Y3_HIP_CHECK(hipGetLastError());
    return Y3_OK;
  });
}
