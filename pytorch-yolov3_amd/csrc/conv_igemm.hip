// Implicit-GEMM convolution for gfx950 (MI355X): conv -> per-channel scale/bias (folded BN or
// conv bias) -> LeakyReLU(0.1) -> optional residual add, NHWC activations.
//
// Replaces the reference's Conv2d + BatchNorm2d(eval) + LeakyReLU nn.Sequential
// (/root/reference/yolov3/darknet.py:244-257, run at :367-368) and the shortcut add that
// follows a conv (:376-379).
//
// GEMM view:  D[co][m] = sum_k  Wt[co][k] * A[m][k]
//   m  = (b, oy, ox) output pixel,  k = (ky, kx, ci)  -- ci fastest, so for one filter tap the
//   K-slice of a pixel is a contiguous run of input channels (NHWC), 128 B per K-tile row;
//   Wt = weights stored [Cout][ks*ks*Cin] (K contiguous), the MFMA "A" operand;
//   A  = im2col rows gathered on the fly (never materialised), the MFMA "B" operand, so each
//        lane ends up with 4 consecutive output channels of one pixel -> 8/16-byte NHWC stores.
//
// Tile: BM pixels x BN channels x 128 bytes of K per step, 256 threads = 4 waves,
// v_mfma_f32_16x16x32_bf16 (bf16) or v_mfma_f32_16x16x4_f32 (exact fp32 parity path).
// LDS image per operand: [rows][128 B], 16-byte chunks XOR-swizzled by (row & 7) so that the
// ds_read_b128 fragment reads (16 rows x one chunk column per lane group) are conflict-free.
// Padding taps and M-tail rows read a zero page instead of branching.
#include <string.h>

#include "common.h"
#include "decode_core.h"

namespace {

struct IgemmArgs {
  const char *in;
  const char *wgt;
  const float *scale;
  const float *bias;
  const char *res;
  char *out;
  const char *zero;
  int H, W, Cin, in_ld;
  int Ho, Wo, Cout, out_ld, res_ld;
  int ks, stride, pad;
  int M, HoWo;
  int k_ld, K;
  int n_ktiles, ktiles_per_tap, n_taps;
  int m_tiles, n_tiles;
  int n_major;         // tile order: 0 = channel tiles innermost (an XCD walks all weight panels for a few pixel tiles),
                       // 1 = pixel tiles innermost (an XCD owns a few weight panels): see igemm_tile_origin
  uint32_t mul_hw, sh_hw, mul_w, sh_w;   // n / d == (umulhi(n, mul) + n) >> sh  for n < 2^31 (d = HoWo, Wo)
  uint32_t flags;
  // detection-head fusion (conv_igemm2_kernel<..., DECODE = true>): the YOLO decode of yolo_decode.hip runs on the
  // parked fp32 tile instead of a second kernel reading it back from HBM
  float *y_bbox, *y_prob;
  long long *y_cls;
  int y_anchors, y_attr, y_row_offset, y_rows_total;
  float y_net_w, y_net_h, y_aw[8], y_ah[8];
};

template <typename T>
struct Mma {
  // 16-bit element types (bf16, IEEE half): one 128-byte K-tile = 64 elements = 2 MFMA k-steps of 32; a lane's 16-byte
  // chunk = 8 k values
  static __device__ __forceinline__ void run(f32x4 &acc, const u32x4 &w, const u32x4 &x) { acc = y3_mfma16<T>(w, x, acc); }
};

template <>
struct Mma<float> {
  // one 128-byte K-tile = 32 floats = 2 groups of 16; lane (r, q) holds floats 4q..4q+3 of the
  // group for row r.  MFMA j consumes element j of every lane: it sums k in {j, 4+j, 8+j, 12+j};
  // the same permutation is applied to both operands, so the four MFMAs cover the group.
  static __device__ __forceinline__ void run(f32x4 &acc, const u32x4 &w, const u32x4 &x) {
    const f32x4 wf = __builtin_bit_cast(f32x4, w), xf = __builtin_bit_cast(f32x4, x);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], xf[j], acc, 0, 0, 0);
  }
};

template <typename TO>
__device__ __forceinline__ void store4(char *dst, const float v[4], int nvalid) {
  TO *p = reinterpret_cast<TO *>(dst);
  if (nvalid >= 4) {
    if constexpr (sizeof(TO) == 4) {
      f32x4 o = {v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4 *>(dst) = o;
    } else {
      *reinterpret_cast<u32x2 *>(dst) = y3_pack4<TO>(v[0], v[1], v[2], v[3]);
    }
  } else {
    for (int r = 0; r < nvalid; ++r) p[r] = y3_from_float<TO>(v[r]);
  }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool GENERIC_K>
__global__ __launch_bounds__(256) void conv_igemm_kernel(IgemmArgs p) {
  constexpr int ES = sizeof(T);
  constexpr int CE = 16 / ES;    // elements per 16-byte chunk
  constexpr int BKE = 128 / ES;  // elements per K-tile
  constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int A_CH = BM / 32, B_CH = BN / 32;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  static_assert(MI >= 1 && NI >= 1 && A_CH >= 1 && B_CH >= 1, "tile too small");

  __shared__ __attribute__((aligned(16))) char smem[(BM + BN) * 128];
  char *sA = smem;             // pixels  [BM][128 B]
  char *sB = smem + BM * 128;  // weights [BN][128 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int tile = y3_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (p.n_major ? tile % p.m_tiles : tile / p.n_tiles) * BM;
  const int n0 = (p.n_major ? tile / p.m_tiles : tile % p.n_tiles) * BN;

  // ---- loader set-up: thread owns LDS chunk slot (tid&7) of rows (tid>>3) + 32*i ------------
  const int slot = tid & 7;
  const int row0 = tid >> 3;
  const int kc = slot ^ (row0 & 7);  // logical chunk held at this LDS position (source-side swizzle)

  const char *a_base[A_CH];
  uint32_t a_taps[A_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int m = m0 + row0 + 32 * i;
    a_base[i] = p.zero;
    a_taps[i] = 0u;
    if (m < p.M) {
      const int b = m / p.HoWo;
      const int rem = m - b * p.HoWo;
      const int oy = rem / p.Wo;
      const int ox = rem - oy * p.Wo;
      const int iy0 = oy * p.stride - p.pad;
      const int ix0 = ox * p.stride - p.pad;
      const long long pix = ((long long)b * p.H + iy0) * p.W + ix0;
      a_base[i] = p.in + pix * p.in_ld * ES + (GENERIC_K ? 0 : kc * 16);
      uint32_t mask = 0u;
      for (int ky = 0; ky < p.ks; ++ky)
        for (int kx = 0; kx < p.ks; ++kx) {
          const bool ok = (unsigned)(iy0 + ky) < (unsigned)p.H && (unsigned)(ix0 + kx) < (unsigned)p.W;
          mask |= (ok ? 1u : 0u) << (ky * p.ks + kx);
        }
      a_taps[i] = mask;
    }
  }
  const char *b_base[B_CH];
#pragma unroll
  for (int i = 0; i < B_CH; ++i)
    b_base[i] = p.wgt + ((long long)(n0 + row0 + 32 * i) * p.k_ld) * ES + kc * 16;

  u32x4 a_reg[A_CH], b_reg[B_CH];

  auto fetch = [&](int kt) {
    long long koff_k0 = 0;
    if constexpr (!GENERIC_K) {
      // whole K-tile lies inside one filter tap: tap and channel offset are wave-uniform
      // K order: channel chunk outermost, filter tap innermost -- the order of the halo / patch kernels (conv_halo.hip),
      // so that a layer sums in the same order whichever kernel the launcher picks for its grid size
      const int chunk = kt / p.n_taps;
      const int tap = kt - chunk * p.n_taps;
      const int ci0 = chunk * BKE;
      const int ky = tap / p.ks, kx = tap - ky * p.ks;
      const long long tap_off = ((long long)(ky * p.W + kx) * p.in_ld + ci0) * ES;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        const char *src = ((a_taps[i] >> tap) & 1u) ? a_base[i] + tap_off : p.zero;
        a_reg[i] = *reinterpret_cast<const u32x4 *>(src);
      }
      koff_k0 = ((long long)tap * p.Cin + ci0) * ES;
    } else {
      // per-chunk tap: Cin is only a multiple of the chunk width
      const int ke = kt * BKE + kc * CE;
      const int tap = ke / p.Cin;
      const int ci = ke - tap * p.Cin;
      const int ky = tap / p.ks, kx = tap - ky * p.ks;
      const long long tap_off = ((long long)(ky * p.W + kx) * p.in_ld + ci) * ES;
      const bool in_k = ke < p.K;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        bool ok = in_k && ((a_taps[i] >> tap) & 1u);
#ifdef Y3_X_S2BOUND
        // timing-only bound for a staged-once stride-2 kernel (`make variant NAME=s2bound FLAGS=-DY3_X_S2BOUND`, debug = 1):
        // only tap 0 of a stride-2 layer fetches pixels, the other eight read the zero page (results wrong)
        if ((p.flags & 0x40000000u) && p.stride == 2 && tap != 0) ok = false;
#endif
        const char *src = ok ? a_base[i] + tap_off : p.zero;
        a_reg[i] = *reinterpret_cast<const u32x4 *>(src);
      }
    }
    const long long koff = GENERIC_K ? (long long)kt * 128 : koff_k0;
#pragma unroll
    for (int i = 0; i < B_CH; ++i) b_reg[i] = *reinterpret_cast<const u32x4 *>(b_base[i] + koff);
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read addresses: lane (r = lane&15, q = lane>>4) reads chunk (g*4 + q) of row r
  const int fr = lane & 15, fq = lane >> 4;

  fetch(0);
  for (int kt = 0; kt < p.n_ktiles; ++kt) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) *reinterpret_cast<u32x4 *>(sA + tid * 16 + i * 4096) = a_reg[i];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) *reinterpret_cast<u32x4 *>(sB + tid * 16 + i * 4096) = b_reg[i];
    __syncthreads();
    if (kt + 1 < p.n_ktiles) fetch(kt + 1);  // next tile's global loads fly during the MFMAs
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      u32x4 xf[MI], wf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = wm * TM + mi * 16 + fr;
        xf[mi] = *reinterpret_cast<const u32x4 *>(sA + row * 128 + (((g * 4 + fq) ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int row = wn * TN + ni * 16 + fr;
        wf[ni] = *reinterpret_cast<const u32x4 *>(sB + row * 128 + (((g * 4 + fq) ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) Mma<T>::run(acc[mi][ni], wf[ni], xf[mi]);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds channels co..co+3 (registers) of pixel m (lane&15) -------------
  const bool leaky = p.flags & Y3_F_LEAKY;
  const bool has_res = p.flags & Y3_F_RESIDUAL;
  const bool out_f32 = (p.flags & Y3_F_OUT_F32) || sizeof(T) == 4;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int co = n0 + wn * TN + ni * 16 + fq * 4;
    const int nvalid = p.Cout - co;  // may be <= 0 for padded channels
    if (nvalid <= 0) continue;
    const f32x4 sc = *reinterpret_cast<const f32x4 *>(p.scale + co);
    const f32x4 bi = *reinterpret_cast<const f32x4 *>(p.bias + co);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + wm * TM + mi * 16 + fr;
      if (m >= p.M) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = acc[mi][ni][r] * sc[r] + bi[r];
        if (leaky) t = t > 0.f ? t : Y3_LEAKY_SLOPE * t;
        v[r] = t;
      }
      if (has_res) {
        const T *rp = reinterpret_cast<const T *>(p.res) + (long long)m * p.res_ld + co;
        if (nvalid >= 4) {
          if constexpr (sizeof(T) == 4) {
            const f32x4 rv = *reinterpret_cast<const f32x4 *>(rp);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += rv[r];
          } else {
            const typename H16<T>::v4 rv = *reinterpret_cast<const typename H16<T>::v4 *>(rp);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
          }
        } else {
          for (int r = 0; r < nvalid; ++r) v[r] += y3_to_float<T>(rp[r]);
        }
      }
      if (out_f32)
        store4<float>(p.out + ((long long)m * p.out_ld + co) * 4, v, nvalid);
      else
        store4<T>(p.out + ((long long)m * p.out_ld + co) * ES, v, nvalid);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// v2: LDS-DMA pipeline.  Same tile geometry and LDS image as above, but
//   * operands go global -> LDS directly (global_load_lds_dwordx4, 16 B per lane, 1 KiB per wave
//     instruction): no staging VGPRs, no ds_write pass; the XOR swizzle is applied on the per-lane
//     SOURCE address, the destination stays lane-linear (cdna_hip_programming.md rule 21);
//   * two LDS stages, one barrier per K-tile: tile k+1 streams in while tile k feeds the MFMAs;
//   * epilogue through LDS: the fp32 tile is parked in the (now idle) operand stages and written out
//     as whole 16-byte NHWC chunks (a pixel's channels are contiguous), residual read the same way.
// KMODE 0: Cin % BKE == 0 (one tap per K-tile)   1: any Cin % CE == 0 (per-chunk tap, slow)
//       2: BKE % Cin == 0, Cin < BKE (several whole taps per K-tile, e.g. Cin = 32 with bf16)
// (Measured and removed: a register-staged single-stage form, 64-byte K rows for three workgroups per CU, and a
// 256x128 eight-wave tile -- all within +-8 % of this one, none better: profiles/r01_convbench_variants.txt.)
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int KMODE, bool DECODE = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_igemm2_kernel(IgemmArgs p) {
  constexpr int RB = 128;                            // bytes of K per tile row
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int ES = sizeof(T);
  constexpr int CE = 16 / ES;
  constexpr int BKE = RB / ES;
  constexpr int CPRW = RB / 16;                      // 16-byte chunks per tile row
  constexpr int G = RB / 64;                         // 64-byte fragment groups per row
  constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int ROWS_PER_PASS = NT / CPRW;
  constexpr int A_CH = BM / ROWS_PER_PASS, B_CH = BN / ROWS_PER_PASS;
  constexpr int STAGE = (BM + BN) * RB;
  static_assert(MI >= 1 && NI >= 1 && A_CH >= 1 && B_CH >= 1, "tile too small");
  // the fp32 output tile is written out in EP passes of RP pixel rows each (it must fit in the stages)
  // (the detection-head form parks its whole padded logit tile at once, so that the decode's passes are full)
  constexpr int LDL = BN + 4;                        // row stride of the parked logit tile (floats), detection-head form
  constexpr int LDS_BYTES = DECODE && BM * LDL * 4 > 2 * STAGE ? BM * LDL * 4 : 2 * STAGE;
  constexpr int EP = (BM * BN * 4 + LDS_BYTES - 1) / LDS_BYTES <= 1 ? 1 : ((BM * BN * 4 + LDS_BYTES - 1) / LDS_BYTES <= 2 ? 2 : 4);
  constexpr int RP = BM / EP;
  static_assert(RP * BN * 4 <= LDS_BYTES && (RP % TM == 0 || TM % RP == 0), "epilogue tile must fit in the operand stages");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

  Y3_STAMP_DECL
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int tile = y3_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (p.n_major ? tile % p.m_tiles : tile / p.n_tiles) * BM;
  const int n0 = (p.n_major ? tile / p.m_tiles : tile % p.n_tiles) * BN;

  const int slot = tid % CPRW;
  const int row0 = tid / CPRW;
  // logical chunk held at this LDS position (source-side swizzle); rows of later passes keep the same low bits
  const int kc = slot ^ (row0 & 7);

  const char *a_base[A_CH];
  uint32_t a_taps[A_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int m = m0 + row0 + ROWS_PER_PASS * i;
    a_base[i] = p.zero;
    a_taps[i] = 0u;
    if (m < p.M) {
      const uint32_t um = (uint32_t)m;
      const uint32_t b = (__umulhi(um, p.mul_hw) + um) >> p.sh_hw;
      const uint32_t rem = um - b * (uint32_t)p.HoWo;
      const uint32_t oy = (__umulhi(rem, p.mul_w) + rem) >> p.sh_w;
      const uint32_t ox = rem - oy * (uint32_t)p.Wo;
      const int iy0 = (int)oy * p.stride - p.pad;
      const int ix0 = (int)ox * p.stride - p.pad;
      const long long pix = ((long long)b * p.H + iy0) * p.W + ix0;
      a_base[i] = p.in + pix * p.in_ld * ES + (KMODE == 0 ? kc * 16 : 0);
      // tap (ky,kx) is inside the image iff row ky and column kx both are: build the ks*ks mask from
      // two ks-bit masks (2*ks compares instead of ks*ks)
      uint32_t vx = 0u, mask = 0u;
      for (int kx = 0; kx < p.ks; ++kx) vx |= ((unsigned)(ix0 + kx) < (unsigned)p.W ? 1u : 0u) << kx;
      for (int ky = 0; ky < p.ks; ++ky)
        if ((unsigned)(iy0 + ky) < (unsigned)p.H) mask |= vx << (ky * p.ks);
      a_taps[i] = mask;
    }
  }
  const char *b_base[B_CH];
#pragma unroll
  for (int i = 0; i < B_CH; ++i)
    b_base[i] = p.wgt + ((long long)(n0 + row0 + ROWS_PER_PASS * i) * p.k_ld) * ES + kc * 16;

  // multi-tap mode: this lane's tap slot inside a K-tile and its channel offset inside the tap
  int tl = 0, cc = 0, tpt = 1;
  if constexpr (KMODE == 2) {
    const int cpt = p.Cin / CE;  // chunks per tap (power of two because BKE % Cin == 0)
    tpt = CPRW / cpt;
    tl = kc / cpt;
    cc = kc - tl * cpt;
  }

  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void gbl_void;

  // source address of this thread's i-th A chunk of K-tile kt (zero page for padding / tail rows)
  auto a_sources = [&](int kt, const char *(&src)[A_CH]) {
    long long tap_off;
    int tap;
    bool in_k = true;
    if constexpr (KMODE == 0) {
      const int chunk = kt / p.n_taps;                 // chunk outermost, tap innermost: the halo kernels' K order
      tap = kt - chunk * p.n_taps;
      const int ci0 = chunk * BKE;
      const int ky = tap / p.ks, kx = tap - ky * p.ks;
      tap_off = ((long long)(ky * p.W + kx) * p.in_ld + ci0) * ES;
    } else if constexpr (KMODE == 2) {
      tap = kt * tpt + tl;
      const int ky = tap / p.ks, kx = tap - ky * p.ks;
      tap_off = ((long long)(ky * p.W + kx) * p.in_ld + cc * CE) * ES;
      in_k = tap < p.ks * p.ks;
    } else {
      const int ke = kt * BKE + kc * CE;
      tap = ke / p.Cin;
      const int ci = ke - tap * p.Cin;
      const int ky = tap / p.ks, kx = tap - ky * p.ks;
      tap_off = ((long long)(ky * p.W + kx) * p.in_ld + ci) * ES;
      in_k = ke < p.K;
    }
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const bool ok = in_k && ((a_taps[i] >> tap) & 1u);
      src[i] = ok ? a_base[i] + tap_off : p.zero;
    }
  };
  auto issue = [&](int kt, int stage) {   // straight into LDS
    char *sA = smem + stage * STAGE;
    char *sB = sA + BM * RB;
    const char *src[A_CH];
    a_sources(kt, src);
#ifdef Y3_X_S2BOUND
    // debug = 2: the pixel loads of taps 1..8 are not even issued (this kernel waits vmcnt(0) per K-tile: any count is safe)
    const bool skip_a = (p.flags & 0x80000000u) && p.stride == 2 && KMODE == 0 && (kt % p.n_taps) != 0;
    if (!skip_a)
#endif
#pragma unroll
    for (int i = 0; i < A_CH; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void *)src[i], (lds_void *)(sA + wave * 1024 + i * (NT * 16)), 16, 0, Y3_AUX_A);
    long long koff = (long long)kt * RB;
    if constexpr (KMODE == 0) {
      const int chunk = kt / p.n_taps;
      koff = ((long long)(kt - chunk * p.n_taps) * p.Cin + chunk * BKE) * ES;
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void *)(b_base[i] + koff), (lds_void *)(sB + wave * 1024 + i * (NT * 16)), 16, 0, Y3_AUX_W);
  };
  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;

  // write-out role of this thread (epilogue): 8 channels [co, co+8) of pixels (tid / OCT_PER_ROW) + k*(NT / OCT_PER_ROW)
  constexpr int OCT_PER_ROW = BN / 8;
  constexpr int WR = RP * OCT_PER_ROW / NT;          // write-out steps per thread and pass
  static_assert(WR * NT == RP * OCT_PER_ROW && NT % OCT_PER_ROW == 0, "write-out must tile evenly");
  const int oc_mine = tid % OCT_PER_ROW;
  const int co = n0 + oc_mine * 8;
  const bool has_res = p.flags & Y3_F_RESIDUAL;
  // residual prefetch (16 bytes per step) only for the single-pass bf16 layout it was written for
  const bool res_fast = has_res && sizeof(T) == 2 && (p.res_ld % 8) == 0 && co + 8 <= p.Cout;
  u32x4 resv[EP * WR];
  f32x4 sc_lo, sc_hi, bi_lo, bi_hi;

  auto compute = [&](const char *sA, const char *sB) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      u32x4 xf[MI], wf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = wm * TM + mi * 16 + fr;
        const int ch = (g * 4 + fq) ^ (row & 7);
        xf[mi] = *reinterpret_cast<const u32x4 *>(sA + row * RB + (ch << 4));
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int row = wn * TN + ni * 16 + fr;
        const int ch = (g * 4 + fq) ^ (row & 7);
        wf[ni] = *reinterpret_cast<const u32x4 *>(sB + row * RB + (ch << 4));
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) Mma<T>::run(acc[mi][ni], wf[ni], xf[mi]);
      __builtin_amdgcn_s_setprio(0);
    }
  };

  Y3_STAMP(0);
  issue(0, 0);
  for (int kt = 0; kt < p.n_ktiles; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile kt has landed (this wave's pieces)
    __syncthreads();                                   // ... everyone's; stage cur^1 is free again
    if (kt == 0) Y3_STAMP(1);
    if (kt + 1 < p.n_ktiles) issue(kt + 1, cur ^ 1);
    compute(smem + cur * STAGE, smem + cur * STAGE + BM * RB);
  }
  // the epilogue's global reads go out first; a raw barrier (no vmcnt drain) lets them fly while the
  // accumulators are parked in LDS
  sc_lo = *reinterpret_cast<const f32x4 *>(p.scale + co);
  sc_hi = *reinterpret_cast<const f32x4 *>(p.scale + co + 4);
  bi_lo = *reinterpret_cast<const f32x4 *>(p.bias + co);
  bi_hi = *reinterpret_cast<const f32x4 *>(p.bias + co + 4);
  if (res_fast) {
#pragma unroll
    for (int h = 0; h < EP; ++h)
#pragma unroll
      for (int j = 0; j < WR; ++j) {
        const int m = m0 + h * RP + (tid / OCT_PER_ROW) + j * (NT / OCT_PER_ROW);
        const char *rp = p.res + ((long long)m * p.res_ld + co) * ES;
        resv[h * WR + j] = (m < p.M && co < p.Cout) ? *reinterpret_cast<const u32x4 *>(rp) : u32x4{0u, 0u, 0u, 0u};
      }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);   // this wave's fragment reads are done ...
  __builtin_amdgcn_s_barrier();         // ... and everyone's: the stages can hold the output tile
  Y3_STAMP(2);

  // ---- epilogue: raw fp32 accumulators -> LDS (pixel rows, 16-byte chunks XOR-swizzled with the pixel so
  // the 16 lanes holding the same channels of 16 pixels hit different banks); then every thread finishes 8
  // consecutive channels of one pixel per step: scale/bias, LeakyReLU, + residual, one 16-byte store.
  constexpr int CPR = BN / 4;  // 16-byte chunks per tile row
  constexpr int SWZ = (CPR < 16 ? CPR : 16) - 1;
  float *sC = reinterpret_cast<float *>(smem);
  const bool leaky = p.flags & Y3_F_LEAKY;
  const bool out_f32 = (p.flags & Y3_F_OUT_F32) || sizeof(T) == 4;
  const int nvalid = p.Cout - co < 8 ? p.Cout - co : 8;   // <= 0: this thread's channels are padding
#pragma unroll
  for (int h = 0; h < EP; ++h) {
    if (h > 0) __syncthreads();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int cl = wn * TN + ni * 16 + fq * 4;  // channel inside the tile (multiple of 4)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int prow = wm * TM + mi * 16;       // first pixel row of this 16-row fragment (compile-time per wm)
        if (prow / RP == h) {
          const int pl = prow + fr - h * RP;
          if constexpr (DECODE) {
            // logits = sum * scale + bias (the conv epilogue's single fused rounding).  Row stride BN + 4 floats: one
            // 16-byte write per fragment whose 8-lane groups cover a 128-byte bank window exactly (1040 B = 65 x 16), and
            // the decode's per-box reads collide 2-way instead of 3-way at BN + 1 (tools/lds_conflicts.py)
            const f32x4 hs = *reinterpret_cast<const f32x4 *>(p.scale + n0 + cl);
            const f32x4 hb = *reinterpret_cast<const f32x4 *>(p.bias + n0 + cl);
            f32x4 lg;
#pragma unroll
            for (int r = 0; r < 4; ++r) lg[r] = __builtin_fmaf(acc[mi][ni][r], hs[r], hb[r]);
            *reinterpret_cast<f32x4 *>(sC + pl * LDL + cl) = lg;
          } else {
            *reinterpret_cast<f32x4 *>(sC + pl * BN + (((cl >> 2) ^ (pl & SWZ)) << 2)) = acc[mi][ni];
          }
        }
      }
    }
    __syncthreads();
    if constexpr (DECODE) {
      static_assert(RP * LDL * 4 <= LDS_BYTES, "padded logit tile must fit in the operand stages");
      y3_head_decode_rows<NT, LDL>(p, sC, m0 + h * RP, RP, tid);
      continue;
    }
    if (nvalid <= 0) continue;
#pragma unroll
    for (int j = 0; j < WR; ++j) {
      const int pl = (tid / OCT_PER_ROW) + j * (NT / OCT_PER_ROW);
      const int m = m0 + h * RP + pl;
      if (m >= p.M) continue;
      const f32x4 lo = *reinterpret_cast<const f32x4 *>(sC + pl * BN + (((2 * oc_mine) ^ (pl & SWZ)) << 2));
      const f32x4 hi = *reinterpret_cast<const f32x4 *>(sC + pl * BN + (((2 * oc_mine + 1) ^ (pl & SWZ)) << 2));
      float v[8];
      y3_bn_leaky8(v, lo, hi, sc_lo, sc_hi, bi_lo, bi_hi, leaky);
      if (has_res) {
        if (res_fast) {
          if constexpr (sizeof(T) == 2) y3_add8<T>(v, resv[h * WR + j]);
        } else {
          const T *rp = reinterpret_cast<const T *>(p.res) + (long long)m * p.res_ld + co;
          for (int r = 0; r < nvalid; ++r) v[r] += y3_to_float<T>(rp[r]);
        }
      }
      if (out_f32) {
        float *op = reinterpret_cast<float *>(p.out) + (long long)m * p.out_ld + co;
        if (nvalid == 8) {
          *reinterpret_cast<f32x4 *>(op) = f32x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4 *>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
          for (int r = 0; r < nvalid; ++r) op[r] = v[r];
        }
      } else {
        T *op = reinterpret_cast<T *>(p.out) + (long long)m * p.out_ld + co;
        bool done = false;
        if constexpr (sizeof(T) == 2) {
          if (nvalid == 8 && (p.out_ld % 8) == 0) {
            *reinterpret_cast<u32x4 *>(op) = y3_pack8<T>(v);
            done = true;
          }
        }
        if (!done)
          for (int r = 0; r < nvalid; ++r) op[r] = y3_from_float<T>(v[r]);
      }
    }
  }
  Y3_STAMP(3);
  Y3_STAMP_COUNT();
}

// ------------------------------------------------------------------------------------------------
// v3: wave-specialised pipeline.  Same tile geometry, LDS image and epilogue as v2, but the workgroup
// carries twice the waves: waves [0, NC/64) only read fragments and issue MFMAs ("consumers"), waves
// [NC/64, 2*NC/64) only compute source addresses and issue the LDS-DMA loads ("loaders").  Measured on
// v2 (tools/conv_bench.py, stamps): one K-step of two co-resident 128x128 workgroups costs about the SUM
// of its LDS-DMA issue time, its fragment reads and its MFMAs -- a wave that is stuck issuing a 1 KiB
// LDS-DMA piece (60-185 cycles each, MI355X_MICROARCH.md constants table) issues no MFMA.  Splitting the
// roles takes the DMA issue out of the MFMA waves' instruction stream; the SIMD's other wave keeps the
// matrix pipe busy meanwhile.  NS LDS stages (3: 96 KiB, one workgroup per CU, two tiles in flight; 4: 128 KiB);
// one raw barrier per K-step pairs "tile kt has landed" with "stage kt-1 is free".
template <int N>
__device__ __forceinline__ void igemm_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int KMODE, int NS>
__global__ __launch_bounds__(128 * WAVES_M * WAVES_N,
                             NS * (BM + BN) * 128 <= 80 * 1024 ? (WAVES_M * WAVES_N) : (WAVES_M * WAVES_N) / 2)
void conv_igemm3_kernel(IgemmArgs p) {
  constexpr int NC = 64 * WAVES_M * WAVES_N;         // consumer threads (== loader threads)
  constexpr int NT = 2 * NC;
  constexpr int RB = 128;
  constexpr int ES = sizeof(T);
  constexpr int CE = 16 / ES;
  constexpr int BKE = RB / ES;
  constexpr int CPRW = RB / 16;
  constexpr int G = RB / 64;
  constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int ROWS_PER_PASS = NC / CPRW;
  constexpr int A_CH = BM / ROWS_PER_PASS, B_CH = BN / ROWS_PER_PASS;
  constexpr int PER = A_CH + B_CH;                   // LDS-DMA instructions per loader wave and K-tile
  constexpr int STAGE = (BM + BN) * RB;
  constexpr int LDS_BYTES = NS * STAGE;
  static_assert(MI >= 1 && NI >= 1 && A_CH >= 1 && B_CH >= 1, "tile too small");
  static_assert(BM * BN * 4 <= LDS_BYTES, "the fp32 output tile must fit in the operand stages");
  static_assert((NS - 2) * PER <= 63, "vmcnt is 6 bits");

  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: scalar role branch
  const bool loader = wave >= NC / 64;
  Y3_STAMP_DECL

  const int tile = y3_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (p.n_major ? tile % p.m_tiles : tile / p.n_tiles) * BM;
  const int n0 = (p.n_major ? tile / p.m_tiles : tile % p.n_tiles) * BN;
  const int n_kt = p.n_ktiles;

  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void gbl_void;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wm = (wave % (NC / 64)) / WAVES_N, wn = (wave % (NC / 64)) % WAVES_N;
  const int fr = lane & 15, fq = lane >> 4;

  if (loader) {
    // ---------------- loader waves: addresses + LDS-DMA only ----------------
    __builtin_amdgcn_s_setprio(3);   // youngest waves: keep their few instructions from starving behind the MFMA waves
    const int ltid = tid - NC;
    const int lwave = wave - NC / 64;
    const int slot = ltid % CPRW;
    const int row0 = ltid / CPRW;
    const int kc = slot ^ (row0 & 7);
    const char *a_base[A_CH];
    uint32_t a_taps[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int m = m0 + row0 + ROWS_PER_PASS * i;
      a_base[i] = p.zero;
      a_taps[i] = 0u;
      if (m < p.M) {
        const uint32_t um = (uint32_t)m;
        const uint32_t b = (__umulhi(um, p.mul_hw) + um) >> p.sh_hw;
        const uint32_t rem = um - b * (uint32_t)p.HoWo;
        const uint32_t oy = (__umulhi(rem, p.mul_w) + rem) >> p.sh_w;
        const uint32_t ox = rem - oy * (uint32_t)p.Wo;
        const int iy0 = (int)oy * p.stride - p.pad;
        const int ix0 = (int)ox * p.stride - p.pad;
        const long long pix = ((long long)b * p.H + iy0) * p.W + ix0;
        a_base[i] = p.in + pix * p.in_ld * ES + (KMODE == 0 ? kc * 16 : 0);
        uint32_t vx = 0u, mask = 0u;
        for (int kx = 0; kx < p.ks; ++kx) vx |= ((unsigned)(ix0 + kx) < (unsigned)p.W ? 1u : 0u) << kx;
        for (int ky = 0; ky < p.ks; ++ky)
          if ((unsigned)(iy0 + ky) < (unsigned)p.H) mask |= vx << (ky * p.ks);
        a_taps[i] = mask;
      }
    }
    const char *b_base[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      b_base[i] = p.wgt + ((long long)(n0 + row0 + ROWS_PER_PASS * i) * p.k_ld) * ES + kc * 16;
    int tl = 0, cc = 0, tpt = 1;
    if constexpr (KMODE == 2) {
      const int cpt = p.Cin / CE;
      tpt = CPRW / cpt;
      tl = kc / cpt;
      cc = kc - tl * cpt;
    }
    auto issue = [&](int kt, int stage) {
      long long tap_off;
      int tap;
      bool in_k = true;
      long long koff = (long long)kt * RB;
      if constexpr (KMODE == 0) {
        const int chunk = kt / p.n_taps;               // chunk outermost, tap innermost: the halo kernels' K order
        tap = kt - chunk * p.n_taps;
        const int ci0 = chunk * BKE;
        const int ky = tap / p.ks, kx = tap - ky * p.ks;
        tap_off = ((long long)(ky * p.W + kx) * p.in_ld + ci0) * ES;
        koff = ((long long)tap * p.Cin + ci0) * ES;
      } else if constexpr (KMODE == 2) {
        tap = kt * tpt + tl;
        const int ky = tap / p.ks, kx = tap - ky * p.ks;
        tap_off = ((long long)(ky * p.W + kx) * p.in_ld + cc * CE) * ES;
        in_k = tap < p.ks * p.ks;
      } else {
        const int ke = kt * BKE + kc * CE;
        tap = ke / p.Cin;
        const int ci = ke - tap * p.Cin;
        const int ky = tap / p.ks, kx = tap - ky * p.ks;
        tap_off = ((long long)(ky * p.W + kx) * p.in_ld + ci) * ES;
        in_k = ke < p.K;
      }
      char *sA = smem + stage * STAGE;
      char *sB = sA + BM * RB;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        bool ok = in_k && ((a_taps[i] >> tap) & 1u);
#ifdef Y3_X_S2BOUND
        // timing-only bound for a staged-once stride-2 kernel (`make variant NAME=s2bound FLAGS=-DY3_X_S2BOUND`, debug = 1):
        // only tap 0 of a stride-2 layer fetches pixels, the other eight read the zero page (results wrong)
        if ((p.flags & 0x40000000u) && p.stride == 2 && tap != 0) ok = false;
#endif
        const char *src = ok ? a_base[i] + tap_off : p.zero;
        __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(sA + lwave * 1024 + i * (NC * 16)), 16, 0, Y3_AUX_A);
      }
#pragma unroll
      for (int i = 0; i < B_CH; ++i)
        __builtin_amdgcn_global_load_lds((gbl_void *)(b_base[i] + koff), (lds_void *)(sB + lwave * 1024 + i * (NC * 16)), 16, 0, Y3_AUX_W);
    };
    int stage_next = 0;                               // stage that receives the next issued tile
    for (int t = 0; t < NS - 1 && t < n_kt; ++t) {
      issue(t, stage_next);
      stage_next = stage_next + 1 == NS ? 0 : stage_next + 1;
    }
    Y3_STAMP(6);
    for (int kt = 0; kt < n_kt; ++kt) {
      // tiles issued so far: 0 .. min(kt + NS - 2, n_kt - 1); tile kt must have landed
      if (NS > 2 && kt + NS - 2 < n_kt) igemm_wait_vmcnt<(NS - 2) * PER>();
      else if (NS > 3 && kt + NS - 3 < n_kt) igemm_wait_vmcnt<(NS > 3 ? NS - 3 : 0) * PER>();
      else igemm_wait_vmcnt<0>();
      Y3_STAMP(3);
      __builtin_amdgcn_s_barrier();                   // tile kt visible to the consumers; stage of tile kt-1 free
      Y3_STAMP(4);
      if (kt + NS - 1 < n_kt) {
        issue(kt + NS - 1, stage_next);
        stage_next = stage_next + 1 == NS ? 0 : stage_next + 1;
      }
      Y3_STAMP(5);
    }
#ifdef Y3_STAMPS
    if (tid == NC) for (int _i = 3; _i < 7; ++_i) atomicAdd(&g_y3_stamps[_i], _st_acc[_i]);
#endif
  } else {
    // ---------------- consumer waves: fragment reads + MFMAs only ----------------
    int stage = 0;
    Y3_STAMP(2);
    for (int kt = 0; kt < n_kt; ++kt) {
      __builtin_amdgcn_s_barrier();
      Y3_STAMP(0);
      const char *sA = smem + stage * STAGE;
      const char *sB = sA + BM * RB;
      stage = stage + 1 == NS ? 0 : stage + 1;
      u32x4 xf[G][MI], wf[G][NI];
#pragma unroll
      for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int row = wn * TN + ni * 16 + fr;
          wf[g][ni] = *reinterpret_cast<const u32x4 *>(sB + row * RB + ((((g * 4 + fq) ^ (row & 7))) << 4));
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int row = wm * TM + mi * 16 + fr;
          xf[g][mi] = *reinterpret_cast<const u32x4 *>(sA + row * RB + ((((g * 4 + fq) ^ (row & 7))) << 4));
        }
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) Mma<T>::run(acc[mi][ni], wf[g][ni], xf[g][mi]);
      __builtin_amdgcn_s_setprio(0);
      Y3_STAMP(1);
    }
#ifdef Y3_STAMPS
    if (tid == 0) {
      for (int _i = 0; _i < 3; ++_i) atomicAdd(&g_y3_stamps[_i], _st_acc[_i]);
      atomicAdd(&g_y3_stamps[7], 1ull);
    }
#endif
  }

  // ---------------- epilogue: all 2*NC threads write out ----------------
  constexpr int OCT_PER_ROW = BN / 8;
  constexpr int WR = BM * OCT_PER_ROW / NT;
  static_assert(WR * NT == BM * OCT_PER_ROW && NT % OCT_PER_ROW == 0, "write-out must tile evenly");
  const int oc_mine = tid % OCT_PER_ROW;
  const int co = n0 + oc_mine * 8;
  const bool has_res = p.flags & Y3_F_RESIDUAL;
  const bool res_fast = has_res && sizeof(T) == 2 && (p.res_ld % 8) == 0 && co + 8 <= p.Cout;
  u32x4 resv[WR];
  const f32x4 sc_lo = *reinterpret_cast<const f32x4 *>(p.scale + co);
  const f32x4 sc_hi = *reinterpret_cast<const f32x4 *>(p.scale + co + 4);
  const f32x4 bi_lo = *reinterpret_cast<const f32x4 *>(p.bias + co);
  const f32x4 bi_hi = *reinterpret_cast<const f32x4 *>(p.bias + co + 4);
  if (res_fast) {
#pragma unroll
    for (int j = 0; j < WR; ++j) {
      const int m = m0 + (tid / OCT_PER_ROW) + j * (NT / OCT_PER_ROW);
      const char *rp = p.res + ((long long)m * p.res_ld + co) * ES;
      resv[j] = (m < p.M && co < p.Cout) ? *reinterpret_cast<const u32x4 *>(rp) : u32x4{0u, 0u, 0u, 0u};
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();         // every consumer is done reading the last stage

  constexpr int CPR = BN / 4;
  constexpr int SWZ = (CPR < 16 ? CPR : 16) - 1;
  float *sC = reinterpret_cast<float *>(smem);
  const bool leaky = p.flags & Y3_F_LEAKY;
  const bool out_f32 = (p.flags & Y3_F_OUT_F32) || sizeof(T) == 4;
  const int nvalid = p.Cout - co < 8 ? p.Cout - co : 8;
  if (!loader) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int cl = wn * TN + ni * 16 + fq * 4;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int pl = wm * TM + mi * 16 + fr;
        *reinterpret_cast<f32x4 *>(sC + pl * BN + (((cl >> 2) ^ (pl & SWZ)) << 2)) = acc[mi][ni];
      }
    }
  }
  __syncthreads();
  if (nvalid <= 0) return;
#pragma unroll
  for (int j = 0; j < WR; ++j) {
    const int pl = (tid / OCT_PER_ROW) + j * (NT / OCT_PER_ROW);
    const int m = m0 + pl;
    if (m >= p.M) continue;
    const f32x4 lo = *reinterpret_cast<const f32x4 *>(sC + pl * BN + (((2 * oc_mine) ^ (pl & SWZ)) << 2));
    const f32x4 hi = *reinterpret_cast<const f32x4 *>(sC + pl * BN + (((2 * oc_mine + 1) ^ (pl & SWZ)) << 2));
    float v[8];
    y3_bn_leaky8(v, lo, hi, sc_lo, sc_hi, bi_lo, bi_hi, leaky);
    if (has_res) {
      if (res_fast) {
        if constexpr (sizeof(T) == 2) y3_add8<T>(v, resv[j]);
      } else {
        const T *rp = reinterpret_cast<const T *>(p.res) + (long long)m * p.res_ld + co;
        for (int r = 0; r < nvalid; ++r) v[r] += y3_to_float<T>(rp[r]);
      }
    }
    if (out_f32) {
      float *op = reinterpret_cast<float *>(p.out) + (long long)m * p.out_ld + co;
      if (nvalid == 8) {
        *reinterpret_cast<f32x4 *>(op) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4 *>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
      } else {
        for (int r = 0; r < nvalid; ++r) op[r] = v[r];
      }
    } else {
      T *op = reinterpret_cast<T *>(p.out) + (long long)m * p.out_ld + co;
      bool done = false;
      if constexpr (sizeof(T) == 2) {
        if (nvalid == 8 && (p.out_ld % 8) == 0) {
          *reinterpret_cast<u32x4 *>(op) = y3_pack8<T>(v);
          done = true;
        }
      }
      if (!done)
        for (int r = 0; r < nvalid; ++r) op[r] = y3_from_float<T>(v[r]);
    }
  }
}


// (Round 5: a v4 -- v3 with REGISTER-STAGED loader waves, global_load_dwordx4 -> ds_write_b128, on 256 x 128 and 128 x 128
// tiles -- was built, bit-identical, and measured 480 / 658 / 582 / 711 TFLOP/s on the four stride-2 layers against 568 / 772 /
// 846 / 797 for v2 / v3: removed again, profiles/r05b_igemm4_register_staged.txt; the history keeps it.)

// y3_options: igemm_version 1 = register-staged single buffer, 2 = LDS-DMA double buffer, 3 = wave-specialised;
// igemm_bm 64 = 64-pixel tiles for the wave-specialised kernel (bf16), else 128; igemm_ns = its LDS stages (3 or 4)

template <typename T, int BM, int BN, int WM, int WN, int NS>
int launch_cfg3x(IgemmArgs a, int kmode, hipStream_t s) {
  a.m_tiles = y3_ceil_div(a.M, BM);
  a.n_tiles = y3_ceil_div(a.Cout, BN);
  const dim3 grid(a.m_tiles * a.n_tiles), block(128 * WM * WN);
  if (kmode == 0) Y3_LAUNCH((conv_igemm3_kernel<T, BM, BN, WM, WN, 0, NS>), grid, block, 0, s, a);
  else if (kmode == 2) Y3_LAUNCH((conv_igemm3_kernel<T, BM, BN, WM, WN, 2, NS>), grid, block, 0, s, a);
  else Y3_LAUNCH((conv_igemm3_kernel<T, BM, BN, WM, WN, 1, NS>), grid, block, 0, s, a);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_cfg3(const IgemmArgs &a, int kmode, int ns, hipStream_t s) {
  // (a two-stage form existed; it needed more than the 128 VGPRs that two co-resident workgroups leave and measured
  // slower than every other variant, so 2 now means 3)
  if (ns == 4) return launch_cfg3x<T, BM, BN, WM, WN, 4>(a, kmode, s);
  return launch_cfg3x<T, BM, BN, WM, WN, 3>(a, kmode, s);
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_cfg2(IgemmArgs a, int kmode, hipStream_t s) {
  a.m_tiles = y3_ceil_div(a.M, BM);
  a.n_tiles = y3_ceil_div(a.Cout, BN);
  const dim3 grid(a.m_tiles * a.n_tiles), block(64 * WM * WN);
  if (kmode == 0) Y3_LAUNCH((conv_igemm2_kernel<T, BM, BN, WM, WN, 0>), grid, block, 0, s, a);
  else if (kmode == 2) Y3_LAUNCH((conv_igemm2_kernel<T, BM, BN, WM, WN, 2>), grid, block, 0, s, a);
  else Y3_LAUNCH((conv_igemm2_kernel<T, BM, BN, WM, WN, 1>), grid, block, 0, s, a);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_cfg(const IgemmArgs &a0, bool generic, hipStream_t s) {
  IgemmArgs a = a0;
  a.m_tiles = y3_ceil_div(a.M, BM);
  a.n_tiles = y3_ceil_div(a.Cout, BN);
  const dim3 grid(a.m_tiles * a.n_tiles), block(256);
  if (generic)
    Y3_LAUNCH((conv_igemm_kernel<T, BM, BN, WM, WN, true>), grid, block, 0, s, a);
  else
    Y3_LAUNCH((conv_igemm_kernel<T, BM, BN, WM, WN, false>), grid, block, 0, s, a);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}

}  // namespace

bool y3_conv_igemm_supported(const y3_op &op) {
  const int es = y3_elem_size(op.dtype);
  const int ce = 16 / es;
  if (op.flags & (Y3_F_IN_NCHW_F32 | Y3_F_IN_NHWC_U8BGR)) return false;
  if (op.ksize < 1 || op.ksize > 5) return false;
  if (op.in_c % ce != 0 || op.in_ld % ce != 0) return false;
  if (op.out_ld % 4 != 0) return false;
  if ((op.flags & Y3_F_RESIDUAL) && op.res_ld % 4 != 0) return false;
  if (op.cout_pad % 128 != 0 || op.cout_pad < op.out_c) return false;
  const int bke = 128 / es;
  if (op.k_ld % bke != 0 || op.k_ld < op.ksize * op.ksize * op.in_c) return false;
  return true;
}

// n / d == (umulhi(n, mul) + n) >> sh for 0 <= n < 2^31 (round-up method, d >= 1)
static void igemm_fast_div(uint32_t d, uint32_t &mul, uint32_t &sh) {
  if (d <= 1) { mul = 0; sh = 0; return; }
  sh = 0;
  while ((1u << sh) < d) ++sh;
  mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << sh) - d)) / d + 1);
}

int y3_launch_conv_igemm(const y3_op &op, const void *d_in, const void *d_zero, hipStream_t s,
                         const char **kernel_name, bool dry_run, int force_version, int force_ns, int force_bm) {
  const int version = force_version ? force_version : y3_opt().igemm_version;
  const int ns = force_ns ? force_ns : y3_opt().igemm_ns;
  const int bm_knob = force_bm ? force_bm : y3_opt().igemm_bm;
  Y3_REQUIRE(y3_conv_igemm_supported(op), "conv block %d: shape not supported by the igemm kernel",
             op.block_idx);
  const int es = y3_elem_size(op.dtype);
  const int bke = 128 / es;
  IgemmArgs a;
  a.in = static_cast<const char *>(d_in);
  a.wgt = static_cast<const char *>(op.d_weight);
  a.scale = op.d_scale;
  a.bias = op.d_bias;
  a.res = static_cast<const char *>(op.d_res);
  a.out = static_cast<char *>(op.d_out);
  a.zero = static_cast<const char *>(d_zero);
  a.H = op.in_h; a.W = op.in_w; a.Cin = op.in_c; a.in_ld = op.in_ld;
  a.Ho = op.out_h; a.Wo = op.out_w; a.Cout = op.out_c; a.out_ld = op.out_ld; a.res_ld = op.res_ld;
  a.ks = op.ksize; a.stride = op.stride; a.pad = op.pad;
  a.HoWo = op.out_h * op.out_w;
  a.M = op.batch * a.HoWo;
  a.k_ld = op.k_ld;
  a.K = op.ksize * op.ksize * op.in_c;
  // K-tiling mode: 0 one tap per tile, 2 several whole taps per tile, 1 per-chunk taps
  int kmode = 1;
  if (op.in_c % bke == 0) kmode = 0;
  else if (bke % op.in_c == 0) kmode = 2;
  a.ktiles_per_tap = kmode == 0 ? op.in_c / bke : 0;
  a.n_taps = op.ksize * op.ksize;
  if (kmode == 0) a.n_ktiles = op.ksize * op.ksize * a.ktiles_per_tap;
  else if (kmode == 2) a.n_ktiles = y3_ceil_div(op.ksize * op.ksize, bke / op.in_c);
  else a.n_ktiles = y3_ceil_div(a.K, bke);
  Y3_REQUIRE(a.n_ktiles * bke <= op.k_ld, "conv block %d: k_ld %d too small for %d K-tiles", op.block_idx, op.k_ld, a.n_ktiles);
  a.m_tiles = a.n_tiles = 0;
  // Each XCD gets one contiguous run of tile ids (y3_xcd_remap) and has its own L2.  Channel tiles innermost: the run covers a
  // few pixel tiles x ALL channel tiles, so every XCD fetches the whole weight matrix and 1/8 of the activations; pixel
  // tiles innermost: every XCD fetches all activations and 1/8 of the weights.  Whichever operand is bigger is the one to
  // split (yolov3-tiny's 13^2 layers at batch 8: 18.9 MB of float32 weights against 2.8 MB of input -- 45 MB of traffic per
  // launch for 16 MB algorithmic before this, profiles/r03_traffic.json).  Placement only: results do not change.
  a.n_major = (double)op.out_c * a.K > (double)op.batch * op.in_h * op.in_w * op.in_c ? 1 : 0;
  a.flags = op.flags | (y3_debug_flags() ? 0x40000000u : 0u) | (y3_debug_flags() == 2 ? 0x80000000u : 0u);
  Y3_REQUIRE((long long)op.batch * a.HoWo < (1ll << 31), "conv block %d: too many output pixels for the 32-bit tile index", op.block_idx);
  igemm_fast_div((uint32_t)a.HoWo, a.mul_hw, a.sh_hw);
  igemm_fast_div((uint32_t)a.Wo, a.mul_w, a.sh_w);

  const int dt = op.dtype;
  const bool bf = y3_is16(dt);                   // a 16-bit storage mode (bf16 / IEEE half): same tiles, same selection
  // channel-tile width follows Cout so narrow layers do not multiply zero padding ...
  int bn = op.out_c > 64 ? 128 : (op.out_c > 32 ? 64 : 32);
  // ... and shrinks while the grid would leave most CUs without a workgroup (small maps / small batches: 13^2 x 8 frames
  // of yolov3-tiny has 11 pixel tiles; 128-channel tiles of its 512 -> 1024 layer are 88 workgroups on 256 CUs, two per
  // CU resident).  Narrower tiles re-read the (small) activation tile more often and keep the weight bytes per FLOP.
  if (version == 2 && !(op.flags & Y3_F_OUT_F32) && !((unsigned)y3_opt().auto_mask & Y3_AM_NO_BN_SHRINK)) {
    const long long m_tiles = y3_ceil_div(a.M, 128);
    while (bn > 32 && m_tiles * y3_ceil_div(op.out_c, bn) < y3_device_cus()) bn >>= 1;   // 362 / 368 workgroups at 128 measured faster than twice as many at 64
  }
  // float32-output (detection head) convs: the direct epilogue of v1 measured faster
  if (version == 1 || (version == 2 && bf && (op.flags & Y3_F_OUT_F32))) {
    const bool generic = kmode != 0;
    if (generic) a.n_ktiles = y3_ceil_div(a.K, bke);
    if (bn == 128) {
      *kernel_name = Y3_KNAME(dt, "conv_igemm_", "_128x128");
      if (dry_run) return Y3_OK;
      return y3_by_dtype(dt, [&](auto tag) { return launch_cfg<decltype(tag), 128, 128, 2, 2>(a, generic, s); });
    } else if (bn == 64) {
      *kernel_name = Y3_KNAME(dt, "conv_igemm_", "_128x64");
      if (dry_run) return Y3_OK;
      return y3_by_dtype(dt, [&](auto tag) { return launch_cfg<decltype(tag), 128, 64, 2, 2>(a, generic, s); });
    }
    *kernel_name = Y3_KNAME(dt, "conv_igemm_", "_128x32");
    if (dry_run) return Y3_OK;
    return y3_by_dtype(dt, [&](auto tag) { return launch_cfg<decltype(tag), 128, 32, 4, 1>(a, generic, s); });
  }
  if (version == 3 && bn == 128 && bm_knob == 64 && bf && !(op.flags & Y3_F_OUT_F32)) {
    *kernel_name = Y3_KNAME(dt, "conv_igemm3_", "_64x128");   // 64-pixel tiles: twice the workgroups, two per CU at 3 stages
    if (dry_run) return Y3_OK;
    return y3_by_dtype16(dt, [&](auto tag) { return launch_cfg3<decltype(tag), 64, 128, 1, 4>(a, kmode, ns, s); });
  }
  if (version == 3 && bn == 128 && !(op.flags & Y3_F_OUT_F32)) {
    *kernel_name = Y3_KNAME(dt, "conv_igemm3_", "_128x128");
    if (dry_run) return Y3_OK;
    return y3_by_dtype(dt, [&](auto tag) { return launch_cfg3<decltype(tag), 128, 128, 2, 2>(a, kmode, ns, s); });
  }
  // 96 x 64 tiles where they fit the chip in ONE round of equal workgroups and the tile above does not: yolov3-tiny's big
  // float32 layers at batch 8 are 352 / 344 tiles of 128 x 32 on 256 CUs (the CUs that get two take twice as long: 39 % of
  // the float32 MFMA peak) against 240 / 228 of 96 x 64, with twice the weight bytes reused per pixel fragment.  Same K order,
  // same bits.  igemm_bm = 96 forces them (A/B), igemm_bm = 128 forbids them.
  if (version == 2 && !(op.flags & Y3_F_OUT_F32) && op.out_c % 64 == 0 && bm_knob != 128 && bm_knob != 64) {
    const int n_cu = y3_device_cus();
    const long long t96 = (long long)y3_ceil_div(a.M, 96) * (op.out_c / 64);
    const long long tcur = (long long)y3_ceil_div(a.M, 128) * y3_ceil_div(op.out_c, bn);
    const bool pays = t96 <= n_cu && t96 * 4 >= n_cu * 3 && tcur > n_cu && tcur < 2 * n_cu && a.n_ktiles >= 32;
    if (bm_knob == 96 || pays) {
      *kernel_name = Y3_KNAME(dt, "conv_igemm2_", "_96x64");
      if (dry_run) return Y3_OK;
      return y3_by_dtype(dt, [&](auto tag) { return launch_cfg2<decltype(tag), 96, 64, 2, 2>(a, kmode, s); });
    }
  }
  if (bn == 128) {
    *kernel_name = Y3_KNAME(dt, "conv_igemm2_", "_128x128");
    if (dry_run) return Y3_OK;
    return y3_by_dtype(dt, [&](auto tag) { return launch_cfg2<decltype(tag), 128, 128, 2, 2>(a, kmode, s); });
  } else if (bn == 64) {
    *kernel_name = Y3_KNAME(dt, "conv_igemm2_", "_128x64");
    if (dry_run) return Y3_OK;
    return y3_by_dtype(dt, [&](auto tag) { return launch_cfg2<decltype(tag), 128, 64, 2, 2>(a, kmode, s); });
  }
  *kernel_name = Y3_KNAME(dt, "conv_igemm2_", "_128x32");
  if (dry_run) return Y3_OK;
  return y3_by_dtype(dt, [&](auto tag) { return launch_cfg2<decltype(tag), 128, 32, 4, 1>(a, kmode, s); });
}

Y3_STAMP_READER(y3_debug_stamps_igemm)

// ---- detection head: 1x1 conv (bias, no activation, float32 logits) + YOLO decode in one launch -------------------
// op0: the head conv as the plan holds it (Y3_F_OUT_F32, Cout = anchors * attributes <= 256); op1: the Y3_OP_YOLO op
// reading it.  16-bit networks (bf16 / fp16) only: the float32 parity path keeps the two kernels (sequential class loop).

bool y3_conv_head_decode_supported(const y3_op &op0, const y3_op &op1) {
  if (!y3_opt().fuse_head) return false;
  if (op0.kind != Y3_OP_CONV || op1.kind != Y3_OP_YOLO || !y3_is16(op0.dtype)) return false;
  if (op0.ksize != 1 || op0.stride != 1 || !(op0.flags & Y3_F_OUT_F32)) return false;
  if (op0.flags & (Y3_F_LEAKY | Y3_F_RESIDUAL | Y3_F_IN_NCHW_F32 | Y3_F_IN_NHWC_U8BGR | Y3_F_PLAN_INPUT)) return false;
  if (!y3_conv_igemm_supported(op0) || op0.in_c % 64 != 0 || op0.out_c > 256 || op0.cout_pad < 256) return false;
  if (op1.d_in != op0.d_out || op1.in_ld != op0.out_ld || op1.in_h != op0.out_h || op1.in_w != op0.out_w) return false;
  if (op1.batch != op0.batch || op1.n_anchor < 1 || op1.n_anchor > 8 || op1.n_attr <= 5) return false;
  if (op1.n_anchor * op1.n_attr != op0.out_c) return false;
  return op1.d_bbox && op1.d_prob && op1.d_cls;
}

int y3_launch_conv_head_decode(const y3_op &op0, const y3_op &op1, const void *d_zero, hipStream_t s,
                               const char **kernel_name, bool dry_run, const void *frag_w) {
  // the direct-weights form (conv_1x1.hip) wherever its shape constraints hold: nothing in its K loop waits on a barrier or a cold
  // load; this tiled form (one K-step of prefetch) keeps the other shapes
  if (y3_conv_head_dw_fits(op0)) return y3_launch_conv_head_decode_dw(op0, op1, d_zero, s, kernel_name, dry_run, frag_w ? frag_w : op0.d_weight_frag);
  *kernel_name = Y3_KNAME(op0.dtype, "conv_head_decode_", "_64x256");
  if (dry_run) return Y3_OK;
  IgemmArgs a;
  a.in = static_cast<const char *>(op0.d_in);
  a.wgt = static_cast<const char *>(op0.d_weight);
  a.scale = op0.d_scale;
  a.bias = op0.d_bias;
  a.res = nullptr;
  a.out = nullptr;
  a.zero = static_cast<const char *>(d_zero);
  a.H = op0.in_h; a.W = op0.in_w; a.Cin = op0.in_c; a.in_ld = op0.in_ld;
  a.Ho = op0.out_h; a.Wo = op0.out_w; a.Cout = op0.out_c; a.out_ld = op0.out_ld; a.res_ld = 0;
  a.ks = 1; a.stride = 1; a.pad = 0;
  a.HoWo = op0.out_h * op0.out_w;
  a.M = op0.batch * a.HoWo;
  a.k_ld = op0.k_ld;
  a.K = op0.in_c;
  a.ktiles_per_tap = op0.in_c / 64;
  a.n_taps = 1;
  a.n_ktiles = a.ktiles_per_tap;
  a.flags = op0.flags;
  igemm_fast_div((uint32_t)a.HoWo, a.mul_hw, a.sh_hw);
  igemm_fast_div((uint32_t)a.Wo, a.mul_w, a.sh_w);
  a.y_bbox = op1.d_bbox; a.y_prob = op1.d_prob; a.y_cls = reinterpret_cast<long long *>(op1.d_cls);
  a.y_anchors = op1.n_anchor; a.y_attr = op1.n_attr;
  a.y_row_offset = op1.row_offset; a.y_rows_total = op1.rows_total;
  a.y_net_w = op1.net_w; a.y_net_h = op1.net_h;
  for (int i = 0; i < 8; ++i) { a.y_aw[i] = op1.anchor_w[i]; a.y_ah[i] = op1.anchor_h[i]; }
  // 64 pixels x all 255 channels per workgroup: 80 KiB of LDS (two operand stages; the padded logit tile of 64 x 260 floats
  // is parked in them), so two workgroups share a CU and one's decode runs under the other's loads (128-pixel tiles, one
  // per CU, measured equal in isolation: profiles/r02n_heads.txt)
  a.n_tiles = 1;
  a.n_major = 0;
  a.m_tiles = y3_ceil_div(a.M, 64);
  return y3_by_dtype16(op0.dtype, [&](auto tag) {
    Y3_LAUNCH((conv_igemm2_kernel<decltype(tag), 64, 256, 1, 4, 0, true>), dim3(a.m_tiles), dim3(256), 0, s, a);
    Y3_HIP_CHECK(hipGetLastError());
    return Y3_OK;
  });
}
