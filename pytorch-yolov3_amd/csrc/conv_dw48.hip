// Small-grid direct-weights kernel (round 6; 16-bit storage modes): 1x1 and 3x3 stride-1 conv -> scale/bias -> LeakyReLU (-> + shortcut)
// for layers whose map x batch is SMALL -- one frame at a time, the mode the reference's command line and video loop run
// (/root/reference/yolov3/__main__.py:157-165, inference.py:527-530), and batches of 2-8.  Same contract as conv_igemm.hip (replaces
// /root/reference/yolov3/darknet.py:244-257 and the shortcut at :376-379).
//
// Why: at batch 1 the 128 x 128 implicit GEMMs put 24-92 workgroups on 256 CUs and walk a LONG K (1152 .. 4608) with one or two K-steps
// of prefetch: 12.7 / 18.4 / 31.1 us per 3x3 layer at 76^2 / 38^2 / 19^2 for 3.4 GFLOP each, 0.56 of the 1.07 ms a frame's kernels take
// (profiles/r06_per_op_dispatch_times_batch1.txt).  Sums over K must stay sequential per output (every MFMA conv kernel here adds a
// layer's K in the same order -- a frame's bits do not depend on the batch it travels in), so the only way to use the idle CUs is MORE,
// SMALLER tiles; what makes small tiles affordable is the structure of conv1x1_dw (conv_1x1.hip):
//   * a WAVE owns 48 pixels x 32 channels (3 x 2 MFMA fragments, 24 accumulator registers); a workgroup is 1, 2, 4 or 8 such waves
//     side by side in the channel dimension (48 x 32..256 tiles), chosen so that the layer is ONE round of workgroups on the chip;
//   * the workgroup's whole input -- the 48 + 2W + 2 raster pixels all nine taps touch (48 for 1x1) x ALL Cin -- is staged in LDS by the
//     prologue, every LDS-DMA piece in flight at once; tap (ky, kx) of pixel p is halo row p + ky W + kx, border taps read a zero row;
//   * weight fragments come straight from the shared fragment-order copy (y3_pair_perm'd rows), DEPTH half-steps ahead, as inline-asm
//     loads with counted waits that name their registers; no barrier after the prologue's;
//   * the K loop is FULLY unrolled (Cin / 64 is a template parameter): chunk outermost, tap innermost, K-halves in order -- the order of
//     every other MFMA conv kernel: same bits;
//   * epilogue in registers: a lane holds eight consecutive channels of its pixel -> one 16-byte shortcut load, one 16-byte store.
#include <utility>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

struct Dw48Args {
  const char *in;
  const char *wgt;     // fragment order: block (channel block cb, K block kb) at ((cb * (k_ld / 32) + kb) << 10)
  const float *scale;
  const float *bias;
  const char *res;
  char *out;
  const char *zero;
  int M, H, W, HW, in_ld, out_ld, res_ld, k_ld;   // M, H, W, HW: the OUTPUT raster (stride 1: the input's as well)
  int in_h, in_w;      // input map (stride 2: 2 H x 2 W)
  int n_ctiles;        // channel tiles per pixel tile = Cout / (32 * nwc)
  int nwc;             // waves of the workgroup that compute (1 / 2 / 4 / 8); the rest only help to stage the halo image
  int m_tiles;         // pixel tiles = ceil(M / 48)
  int m_inner;         // tile order: 1 = pixel tiles innermost (consecutive tiles share a WEIGHT slice), 0 = channel tiles innermost (a halo)
  int hr;              // halo rows that hold pixels (48 for 1x1, 48 + 2W + 2 for 3x3, 194 + 2W for 3x3 stride 2); row hr is all zero
  uint32_t mul_hw, sh_hw, mul_w, sh_w;   // n / d == (umulhi(n, mul) + n) >> sh
  uint32_t flags;
};

template <int V>
struct HalfC { static constexpr int value = V; };
template <int... I, typename F>
__device__ __forceinline__ void dw48_for_impl(std::integer_sequence<int, I...>, F &&f) { (f(HalfC<I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void dw48_for(F &&f) { dw48_for_impl(std::make_integer_sequence<int, N>{}, f); }

// s_waitcnt vmcnt(N) that NAMES the two registers it waits for (see dw_wait_vm in conv_halo.hip)
template <int N>
__device__ __forceinline__ void dw48_wait_vm(u32x4 (&w)[2]) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w[0]), "+v"(w[1]) : "n"(N) : "memory");
}

#ifndef Y3_DW48_DEPTH
#define Y3_DW48_DEPTH 8
#endif
#ifndef Y3_DW48_HELPERS
#define Y3_DW48_HELPERS 1                              // 0: a workgroup has only its computing waves (A/B)
#endif

// Stride 2 (ST = 2, 3x3, even input maps): the same kernel over FOUR PARITY PLANES of the input.  With iy = 2 oy - 1 + ky and
// ix = 2 ox - 1 + kx every tap reads one of  EE = in[2i][2j],  EO = in[2i][2j+1],  OE = in[2i+1][2j],  OO = in[2i+1][2j+1],  and each
// plane is a raster in OUTPUT coordinates, so a tap is again a constant row offset from the lane's output pixel m:
//   (1,1) EE[m]      (1,0) EO[m-1]  (1,2) EO[m]      (0,1) OE[m-W]  (2,1) OE[m]      (0,0) OO[m-W-1]  (0,2) OO[m-W]  (2,0) OO[m-1]  (2,2) OO[m]
// LDS rows: EE [0, 48), EO [48, 97) from m0 - 1, OE [97, 145 + W) from m0 - W, OO [145 + W, 194 + 2W) from m0 - W - 1; only the top and
// left taps can fall outside the map.  The LDS-DMA gather places the planes (any source address per 16 bytes), nothing else changes.
// NH > 1: the image of ALL input channels does not fit in LDS (four input pixels per output pixel: 38^2 x 512 -> 19^2 needs 233 KiB),
// so the sum runs over NH images of Cin / NH channels one after the other -- K-tiles outermost as ever; the weight ring drains at the
// end of an image and is primed again behind the next image's pieces (the two latencies overlap).
template <typename T, int KS, int NKT, int ST = 1, int NH = 1>
__global__ __launch_bounds__(512) void conv_dw48_kernel(Dw48Args p) {
  static_assert(sizeof(T) == 2 && (KS == 1 || KS == 3) && (NKT % 2 == 0 || NKT == 1), "16-bit modes; 1x1 or 3x3; Cin 64 or a multiple of 128");
  constexpr int SWZ = NKT == 1 ? 7 : 15;               // a row of 64 channels has eight 16-byte chunks, not sixteen
  static_assert((ST == 1 && NH == 1) || (ST == 2 && KS == 3), "stride 2 and channel halves: the 3x3 form only");
  constexpr int BM = 48, MI = 3, NI = 2;
  constexpr int TAPS = KS * KS;
  constexpr int RB = NKT * 128;                        // bytes of one pixel's channels in an image = LDS row pitch (a multiple of 256)
  constexpr int S = NKT * TAPS * 2;                    // half-steps (32 K-elements each) per image: NH x S make the whole sum
  // half-steps of weight fragments in flight (2 loads of 1 KiB per wave each).  Only 2-8 waves run on a CU here and a half-step is six
  // MFMAs (96 cycles): six half-steps in flight covered a quarter of an L2 round trip (19^2 x 1 frame: 19.9 us for 6.6 us of MFMA work)
  constexpr int DEPTH = S < Y3_DW48_DEPTH ? S : Y3_DW48_DEPTH;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = blockDim.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int tile = y3_xcd_remap(blockIdx.x, gridDim.x);
  // Each XCD (own L2) gets one contiguous run of tile ids.  Whichever operand is larger should be fetched ONCE per XCD: with the
  // pixel tiles innermost an XCD's run covers all pixels x a few weight slices (19^2 x 1 frame: 9.4 MB of weights, 0.4 MB of pixels --
  // channel tiles innermost made every XCD stream all 9.4 MB: 75 MB per launch, 16.8 us), with the channel tiles innermost a few pixel
  // tiles x all weights (76^2: 1.5 MB of pixels, 0.6 MB of weights).  Placement only: results do not depend on it.
  const int mt = p.m_inner ? tile % p.m_tiles : tile / p.n_ctiles;
  const int ct = p.m_inner ? tile / p.m_tiles : tile % p.n_ctiles;
  const int m0 = mt * BM;
  const int n0 = (ct * p.nwc + wave) * 32;

  // ---- the halo image, rows 0 .. hr (row hr: zeros).  Stride 1: row r holds flattened input pixel q0 + r.  A wave-instruction fills
  // 1 KiB of consecutive LDS; the 16-byte chunk c of row r sits at chunk position c ^ (r & 15) (applied on the source address).
  const long long q0 = (long long)m0 - (KS == 3 ? p.W + 1 : 0);
  const int pieces = ((p.hr + 1) * RB + 1023) >> 10;
  auto stage = [&](int h) {
    for (int i = wave; i < pieces; i += nwaves) {
      const int o = i * 1024 + lane * 16;
      const int r = o / RB, cpos = (o - r * RB) >> 4;
      const int c = cpos ^ (r & SWZ);
      long long q;
      bool ok;
      if constexpr (ST == 1) {
        q = q0 + r;
        ok = r < p.hr && q >= 0 && q < p.M;
      } else {
        // plane of row r, and the output-raster index its row stands for
        const int e1 = 48, e2 = 97, e3 = 145 + p.W;
        const int pr = r >= e2 ? 1 : 0, pc = ((r >= e1 && r < e2) || r >= e3) ? 1 : 0;
        const int base = r < e1 ? 0 : (r < e2 ? e1 + 1 : (r < e3 ? e2 + p.W : e3 + p.W + 1));
        const long long mm = (long long)m0 + r - base;
        ok = r < p.hr && mm >= 0 && mm < p.M;
        const uint32_t um = ok ? (uint32_t)mm : 0u;
        const uint32_t img = (__umulhi(um, p.mul_hw) + um) >> p.sh_hw;
        const uint32_t rem = um - img * (uint32_t)p.HW;
        const uint32_t oy = (__umulhi(rem, p.mul_w) + rem) >> p.sh_w;
        const uint32_t ox = rem - oy * (uint32_t)p.W;
        q = ((long long)img * p.in_h + 2 * oy + pr) * p.in_w + 2 * ox + pc;
      }
      const char *src = ok ? p.in + (q * p.in_ld) * 2 + h * RB + c * 16 : p.zero;
      __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(smem + i * 1024), 16, 0, 0);
    }
  };

  // ---- weight fragments of this wave's 32 channels (two 16-channel blocks): K block kb = tap * (Cin / 32) + chunk * 2 + half
  // (chunk = K-tile of the whole layer: image h holds K-tiles h * NKT .. h * NKT + NKT - 1)
  const uint32_t kblocks = (uint32_t)p.k_ld / 32u;
  const uint32_t w_lane = (uint32_t)lane * 16;
  const char *wb[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) wb[ni] = p.wgt + (((long long)(n0 / 16 + ni) * kblocks) << 10);
  u32x4 wf[DEPTH][NI];
  uint32_t w_img = w_lane;                             // + the K-block offset of the image in LDS
  if (wave >= p.nwc) {
    // helper waves: a workgroup always has eight waves to issue the image's LDS-DMA pieces (a piece costs its wave 60-185 cycles, and
    // one or two waves alone would issue 45-70 of them); they take part in the barriers of the images and leave
#pragma unroll 1
    for (int h = 0; h < NH; ++h) {
      if (h > 0) __builtin_amdgcn_s_barrier();
      stage(h);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    return;
  }
  stage(0);
  asm volatile("" ::: "memory");
  auto load_w = [&](auto sc, u32x4 (&w)[NI]) {        // the two loads of half-step s, in the order [ni]
    constexpr int s = decltype(sc)::value;
    constexpr int kt = s / (2 * TAPS), tap = (s / 2) % TAPS, kh = s & 1;
    const uint32_t voff = w_img + (uint32_t)((tap * (NH * NKT * 2) + kt * 2 + kh) << 10);
    // (named copies: a variable that appears ONLY as an inline-asm operand inside a generic lambda is not captured by clang)
    const char *b0 = wb[0], *b1 = wb[1];
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(w[0]) : "v"(voff), "s"(b0));
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(w[1]) : "v"(voff), "s"(b1));
  };
  dw48_for<DEPTH>([&](auto sc) { load_w(sc, wf[decltype(sc)::value]); });

  // ---- fragment addresses, once per workgroup: tap (ky, kx) of this lane's pixel mi * 16 + fr is halo row mi * 16 + fr + ky W + kx; a
  // border tap (zero padding, row wrap of the raster strip) is redirected to the zero row here, so that the K loop's per-half-step
  // address work is ONE xor (the chunk position: (chunk ^ (row & 15)) << 4, and row & 15 does not depend on mi) and three adds --
  // a wave runs alone on its SIMD in this kernel (2-8 waves per CU): every vector instruction between two MFMAs is on its critical path
  typedef const __attribute__((address_space(3))) u32x4 lds_u32x4;
  const int lds0 = (int)(size_t)(lds_void *)smem;
  const int zrow = lds0 + p.hr * RB;                   // the all-zero row (any chunk of it)
  int abase[TAPS][MI], fk4[TAPS];
  {
    uint32_t tapmask[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      tapmask[mi] = 0x1FFu;
      if constexpr (KS == 3) {
        const uint32_t m = (uint32_t)(m0 + mi * 16 + fr);
        const uint32_t img = (__umulhi(m, p.mul_hw) + m) >> p.sh_hw;
        const uint32_t rem = m - img * (uint32_t)p.HW;
        const uint32_t oy = (__umulhi(rem, p.mul_w) + rem) >> p.sh_w;
        const uint32_t ox = rem - oy * (uint32_t)p.W;
        // (stride 2 on an even map: 2 oy + 1 and 2 ox + 1 are always inside)
        const uint32_t vx = (ox >= 1u ? 1u : 0u) | 2u | (ST == 2 || ox + 1u < (uint32_t)p.W ? 4u : 0u);
        tapmask[mi] = (oy >= 1u ? vx : 0u) | (vx << 3) | (ST == 2 || oy + 1u < (uint32_t)p.H ? vx << 6 : 0u);
      }
    }
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      int r0 = fr + (KS == 3 ? (tap / 3) * p.W + tap % 3 : 0);
      if constexpr (ST == 2) {                          // plane base + row offset inside the plane (see the table above)
        const int ky = tap / 3, kx = tap % 3;
        r0 = fr + (ky == 1 ? (kx == 1 ? 0 : 48) : (kx == 1 ? 97 : 145 + p.W)) + (kx == 2 ? 1 : 0) + (ky == 2 ? p.W : 0);
      }
      fk4[tap] = (fq ^ (r0 & SWZ)) << 4;                // chunk c of the row sits at position c ^ (row & 15); c = 8 kt + 4 kh + fq
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        abase[tap][mi] = ((tapmask[mi] >> tap) & 1u) ? lds0 + (mi * 16 + r0) * RB : zrow;
    }
  }

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 xf[3][MI];                                     // fragments of half-steps s, s + 1, s + 2: read TWO half-steps ahead
  auto read_x = [&](auto sc, u32x4 (&x)[MI]) {
    constexpr int s = decltype(sc)::value;
    constexpr int kt = s / (2 * TAPS), tap = (s / 2) % TAPS, kh = s & 1;
    const int off = ((kt * 8 + kh * 4) << 4) ^ fk4[tap];   // (8 kt + 4 kh) and fq share no bits: ((8 kt + 4 kh + fq) ^ key) << 4
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) x[mi] = *reinterpret_cast<lds_u32x4 *>(abase[tap][mi] + off);
  };
  auto mma = [&](const u32x4 (&x)[MI], const u32x4 (&w)[NI]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = y3_mfma16<T>(w[ni], x[mi], acc[mi][ni]);
  };
  auto run_image = [&]() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (DEPTH - 1)) : "memory");   // this wave's halo pieces and half-step 0's weights landed
    __builtin_amdgcn_s_barrier();                      // ... everyone's pieces: the only barrier of the K loop (per image)
    read_x(HalfC<0>{}, xf[0]);
    if constexpr (S > 1) read_x(HalfC<1>{}, xf[1]);
    dw48_for<S>([&](auto sc) {
      constexpr int s = decltype(sc)::value, slot = s % DEPTH;
      // in flight behind half-step s's two loads: those of half-steps s + 1 .. min(s + DEPTH, S) - 1
      constexpr int younger = 2 * ((s + DEPTH < S ? s + DEPTH : S) - s - 1);
      dw48_wait_vm<younger>(wf[slot]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (s + 2 < S) read_x(HalfC<s + 2>{}, xf[(s + 2) % 3]);
      mma(xf[s % 3], wf[slot]);
#pragma unroll
      for (int i = 0; i < MI; ++i) {                   // one fragment read (of half-step s + 2) per two MFMAs of this one
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (s + DEPTH < S) load_w(HalfC<s + DEPTH>{}, wf[slot]);
    });
  };
  if constexpr (NH == 1) {
    run_image();
  } else {
#pragma unroll 1
    for (int h = 0; h < NH; ++h) {
      // (the fragment addresses do not depend on h: opaque per iteration, or the compiler hoists every half-step's address
      // arithmetic out of the loop and runs out of registers)
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        asm volatile("" : "+v"(fk4[tap]));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) asm volatile("" : "+v"(abase[tap][mi]));
      }
      run_image();
      if (h + 1 < NH) {
        // the next image: every wave has its last fragments of this one in registers when it gets here
        __builtin_amdgcn_s_barrier();
        stage(h + 1);
        asm volatile("" ::: "memory");
        w_img = w_lane + (uint32_t)(((h + 1) * NKT * 2) << 10);
        dw48_for<DEPTH>([&](auto sc) { load_w(sc, wf[decltype(sc)::value]); });
      }
    }
  }

  // ---- epilogue in registers: lane (fr, fq) holds channels co .. co + 7 of pixels m0 + mi * 16 + fr
  const int co = n0 + fq * 8;
  const f32x4 sc_lo = *reinterpret_cast<const f32x4 *>(p.scale + co), sc_hi = *reinterpret_cast<const f32x4 *>(p.scale + co + 4);
  const f32x4 bi_lo = *reinterpret_cast<const f32x4 *>(p.bias + co), bi_hi = *reinterpret_cast<const f32x4 *>(p.bias + co + 4);
  const bool leaky = p.flags & Y3_F_LEAKY;
  const bool has_res = p.flags & Y3_F_RESIDUAL;
  u32x4 resv[MI];
  if (has_res) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + mi * 16 + fr;
      resv[mi] = m < p.M ? *reinterpret_cast<const u32x4 *>(p.res + ((long long)m * p.res_ld + co) * 2) : u32x4{0u, 0u, 0u, 0u};
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m0 + mi * 16 + fr;
    float v[8];
    y3_bn_leaky8(v, acc[mi][0], acc[mi][1], sc_lo, sc_hi, bi_lo, bi_hi, leaky);
    if (has_res) y3_add8<T>(v, resv[mi]);
    if (m < p.M) *reinterpret_cast<u32x4 *>(p.out + ((long long)m * p.out_ld + co) * 2) = y3_pack8<T>(v);
  }
}

void dw48_fast_div(uint32_t d, uint32_t &mul, uint32_t &sh) {
  if (d <= 1) { mul = 0; sh = 0; return; }
  sh = 0;
  while ((1u << sh) < d) ++sh;
  mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << sh) - d)) / d + 1);
}

// Shape of the halo image(s) for an op: rows that hold pixels, channels per image, images (NH); rows = 0: not a shape of this kernel
struct Dw48Shape { int rows, cin_img, nh; };
Dw48Shape dw48_shape(const y3_op &op) {
  Dw48Shape z = {0, 0, 0};
  if (op.kind != Y3_OP_CONV || !y3_is16(op.dtype)) return z;
  const bool k1 = op.ksize == 1 && op.pad == 0 && op.stride == 1, k3 = op.ksize == 3 && op.pad == 1 && op.stride == 1;
  const bool k3s2 = op.ksize == 3 && op.pad == 1 && op.stride == 2 && op.in_h % 2 == 0 && op.in_w % 2 == 0;
  if (!k1 && !k3 && !k3s2) return z;
  if (op.flags & (Y3_F_OUT_F32 | Y3_F_IN_NCHW_F32 | Y3_F_IN_NHWC_U8BGR | Y3_F_PLAN_INPUT)) return z;
  if (op.out_c % 32 != 0 || op.cout_pad % 32 != 0 || op.in_ld % 8 != 0 || op.out_ld % 8 != 0 || op.k_ld % 32 != 0) return z;
  if ((op.flags & Y3_F_RESIDUAL) && op.res_ld % 8 != 0) return z;
  if ((op.in_c % 128 != 0 && !(k3s2 && op.in_c == 64)) || op.k_ld < op.ksize * op.ksize * op.in_c) return z;
  Dw48Shape sh;
  sh.rows = k1 ? 48 : (k3 ? 48 + 2 * op.in_w + 2 : 194 + 2 * op.out_w);
  sh.cin_img = op.in_c;
  sh.nh = 1;
  // stride 2: four input pixels per output pixel -- when all channels do not fit, two images of half the channels one after the other
  if (k3s2 && (long long)(sh.rows + 1) * sh.cin_img * 2 + 1024 > 160 * 1024 && sh.cin_img % 256 == 0) { sh.cin_img /= 2; sh.nh = 2; }
  if ((long long)(sh.rows + 1) * sh.cin_img * 2 + 1024 > 160 * 1024) return z;
  const int nkt = sh.cin_img / 64;
  const bool inst = k1 ? (nkt == 2 || nkt == 4 || nkt == 6 || nkt == 8 || nkt == 12 || nkt == 16)
                       : (k3 ? (nkt == 2 || nkt == 4 || nkt == 8) : (nkt == 1 || nkt == 2 || nkt == 4));
  if (!inst) return z;
  if ((long long)op.batch * op.out_h * op.out_w >= (1ll << 31)) return z;
  return sh;
}

// waves per workgroup (1 / 2 / 4 / 8 = 32 .. 256 channels) for this op, or 0 when the kernel does not take it.
// grid: 0 = ONE round of workgroups (the rule the kernel was built for), 1 = a little over one round (Y3_AM_SMALL_DW_WIDE), 2 = any
// (tests / A-B: Y3_AM_SMALL_DW_ALWAYS -- the shape constraints only, the widest workgroup the channel count allows)
int dw48_waves(const y3_op &op, int grid) {
  const Dw48Shape sh = dw48_shape(op);
  if (sh.rows == 0) return 0;
  const long long M = (long long)op.batch * op.out_h * op.out_w;
  const long long mt = (M + 47) / 48;
  const int n_cu = y3_device_cus();
  int widest = 0;
  for (int nw = 8; nw >= 1 && !widest; nw >>= 1)
    if (op.out_c % (32 * nw) == 0) widest = nw;
  if (grid == 2 || widest == 0) return widest;
  if (grid == 0) {
    // ONE round of workgroups: the fewest waves per workgroup (most workgroups) that still fit the chip
    for (int nw = 1; nw <= 8; nw <<= 1) {
      if (op.out_c % (32 * nw) != 0) continue;
      // (not one computing wave per workgroup for a big image: with 256 workgroups each staging 64+ KiB the L2 traffic doubles for
      // nothing -- with or without the rule the layers it touches measure the same, profiles/r06_conv_dw48.txt)
      if (nw == 1 && (long long)(sh.rows + 1) * op.in_c * 2 > 64 * 1024) continue;
      if (mt * (op.out_c / (32 * nw)) <= n_cu) return nw;
    }
    return 0;
  }
  // A little over one round, widest workgroups (profiles/r06_conv_dw48.txt, "grids over one round": us per launch against what the layer
  // ran on).  3x3 (stride 1 and 2): up to 1.5 rounds -- 128 -> 256 @76^2 x 3 frames 19.1 against 26.0, 256 -> 512 @38^2 x 6 frames 30.0
  // against 39.3; at 1.9 rounds the strip kernel's 192-pixel tiles win (76^2 x 4 frames 17.3 against 21.0).  1x1 with ONE channel tile
  // per pixel tile (every pixel is staged once) and 128+ output channels: up to 8 rounds -- 256 -> 128 @76^2 x 3 / 4 / 8 frames 5.2 /
  // 5.6 / 9.1 against 6.5-7.8 / 6.8-8.1 / 10.7-11.1, 384 -> 128 @76^2 x 16 frames 22.7 against 24.0; the weights-resident kernel, where
  // it pays, is asked first (256 -> 128 @76^2 x 16 frames: 14.5 against 17.1).
  const long long tiles = mt * (op.out_c / (32 * widest));
  if (op.ksize == 3) return 2 * tiles <= 3 * (long long)n_cu ? widest : 0;
  return op.out_c >= 128 && op.out_c == 32 * widest && tiles <= 8ll * n_cu ? widest : 0;
}

int dw48_waves_for(const y3_op &op) {
  const unsigned am = (unsigned)y3_opt().auto_mask;
  if (am & Y3_AM_SMALL_DW_ALWAYS) return dw48_waves(op, 2);
  const int nw = dw48_waves(op, 0);
  return nw || !(am & Y3_AM_SMALL_DW_WIDE) ? nw : dw48_waves(op, 1);
}

}  // namespace

// one round of workgroups (or any grid under Y3_AM_SMALL_DW_ALWAYS)
bool y3_conv_dw48_fits(const y3_op &op) { return dw48_waves(op, ((unsigned)y3_opt().auto_mask & Y3_AM_SMALL_DW_ALWAYS) ? 2 : 0) != 0; }
// ... or a grid a little over one round (Y3_AM_SMALL_DW_WIDE): asked after the kernels that win on such grids where they apply
bool y3_conv_dw48_fits_wide(const y3_op &op) { return ((unsigned)y3_opt().auto_mask & Y3_AM_SMALL_DW_WIDE) && dw48_waves(op, 1) != 0; }

int y3_launch_conv_dw48(const y3_op &op, const void *d_in, const void *d_zero, hipStream_t s, const char **kernel_name,
                        bool dry_run, const void *frag_w) {
  const int nw = dw48_waves_for(op);
  Y3_REQUIRE(nw != 0, "conv block %d: not a shape for the small-grid direct-weights kernel", op.block_idx);
  *kernel_name = op.ksize == 3 ? (op.stride == 2 ? Y3_KNAME(op.dtype, "conv_dw48_k3s2_", "") : Y3_KNAME(op.dtype, "conv_dw48_k3_", ""))
                               : Y3_KNAME(op.dtype, "conv_dw48_k1_", "");
  if (dry_run) return Y3_OK;
  void *tmp = nullptr;
  if (!frag_w) {                                      // single-op calls without a shared copy: made here, stream-ordered
    Y3_HIP_CHECK(hipMallocAsync(&tmp, y3_conv_halo_dw_weight_bytes(op), s));
    const int rc = y3_conv_halo_dw_make_weights(op, tmp, s);
    if (rc != Y3_OK) { (void)hipFreeAsync(tmp, s); return rc; }
    frag_w = tmp;
  }
  Dw48Args a;
  a.in = static_cast<const char *>(d_in);
  a.wgt = static_cast<const char *>(frag_w);
  a.scale = op.d_scale; a.bias = op.d_bias;
  a.res = static_cast<const char *>(op.d_res);
  a.out = static_cast<char *>(op.d_out);
  a.zero = static_cast<const char *>(d_zero);
  a.H = op.out_h; a.W = op.out_w; a.HW = op.out_h * op.out_w;
  a.in_h = op.in_h; a.in_w = op.in_w;
  a.M = op.batch * a.HW;
  a.in_ld = op.in_ld; a.out_ld = op.out_ld; a.res_ld = op.res_ld; a.k_ld = op.k_ld;
  a.n_ctiles = op.out_c / (32 * nw);
  a.nwc = nw;
  a.m_tiles = y3_ceil_div(a.M, 48);
  a.m_inner = (double)op.ksize * op.ksize * op.in_c * op.out_c > (double)op.batch * op.in_h * op.in_w * op.in_c;   // weights outweigh the activations
  const Dw48Shape sh = dw48_shape(op);
  a.hr = sh.rows;
  dw48_fast_div((uint32_t)a.HW, a.mul_hw, a.sh_hw);
  dw48_fast_div((uint32_t)a.W, a.mul_w, a.sh_w);
  a.flags = op.flags;
  const int nkt = sh.cin_img / 64;
  const size_t lds = (size_t)(a.hr + 1) * sh.cin_img * 2 + 1024;   // (+ the tail of the last 1-KiB piece)
  const dim3 grid(y3_ceil_div(a.M, 48) * a.n_ctiles), block(Y3_DW48_HELPERS ? 512 : 64 * nw);
  const int ks = op.ksize, st = op.stride, nh = sh.nh;
  const int rc = y3_by_dtype16(op.dtype, [&](auto tag) {
    typedef decltype(tag) T;
    static Y3DeviceOnce once;
    {
      const int rc1 = once.run([]() -> int {
#define Y3_DW48_ATTR(KS_, NKT_) Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_dw48_kernel<T, KS_, NKT_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
        Y3_DW48_ATTR(3, 2); Y3_DW48_ATTR(3, 4); Y3_DW48_ATTR(3, 8);
        Y3_DW48_ATTR(1, 2); Y3_DW48_ATTR(1, 4); Y3_DW48_ATTR(1, 6); Y3_DW48_ATTR(1, 8); Y3_DW48_ATTR(1, 12); Y3_DW48_ATTR(1, 16);
#undef Y3_DW48_ATTR
#define Y3_DW48_ATTR2(NKT_, NH_) Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_dw48_kernel<T, 3, NKT_, 2, NH_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
        Y3_DW48_ATTR2(1, 1); Y3_DW48_ATTR2(2, 1); Y3_DW48_ATTR2(4, 1); Y3_DW48_ATTR2(2, 2); Y3_DW48_ATTR2(4, 2);
#undef Y3_DW48_ATTR2
        return Y3_OK;
      });
      if (rc1 != Y3_OK) return rc1;
    }
#define Y3_DW48_GO(KS_, NKT_) Y3_LAUNCH((conv_dw48_kernel<T, KS_, NKT_>), grid, block, lds, s, a)
    if (ks == 3 && st == 2) {
#define Y3_DW48_GO2(NKT_, NH_) Y3_LAUNCH((conv_dw48_kernel<T, 3, NKT_, 2, NH_>), grid, block, lds, s, a)
      if (nkt == 1) Y3_DW48_GO2(1, 1); else if (nkt == 2 && nh == 1) Y3_DW48_GO2(2, 1); else if (nkt == 4 && nh == 1) Y3_DW48_GO2(4, 1);
      else if (nkt == 2) Y3_DW48_GO2(2, 2); else Y3_DW48_GO2(4, 2);
#undef Y3_DW48_GO2
    } else if (ks == 3) {
      if (nkt == 2) Y3_DW48_GO(3, 2); else if (nkt == 4) Y3_DW48_GO(3, 4); else Y3_DW48_GO(3, 8);
    } else {
      if (nkt == 2) Y3_DW48_GO(1, 2); else if (nkt == 4) Y3_DW48_GO(1, 4); else if (nkt == 6) Y3_DW48_GO(1, 6);
      else if (nkt == 8) Y3_DW48_GO(1, 8); else if (nkt == 12) Y3_DW48_GO(1, 12); else Y3_DW48_GO(1, 16);
    }
#undef Y3_DW48_GO
    Y3_HIP_CHECK(hipGetLastError());
    return Y3_OK;
  });
  if (tmp) (void)hipFreeAsync(tmp, s);
  return rc;
}
