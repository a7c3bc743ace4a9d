// One Darknet-53 residual block (or one 1x1 -> 3x3 pair of a detection branch) in ONE kernel, for the 128-channel
// bottleneck stage (yolov3@608: the eight 76^2 blocks 256 -> 128 -> 256 and the three pairs of the 76^2 branch):
//     z = leaky(bn(conv3x3(leaky(bn(conv1x1(x))))))  [+ x]
// Replaces /root/reference/yolov3/darknet.py:244-257 (twice) and the shortcut at :376-379 (models/yolov3.cfg:197-283,
// :727-773).  The bottleneck tensor never leaves the CU: HBM sees x (1.22x for the border) and z only.
//
// STATUS (round 4): correct and bit-identical to the two separate launches, but only LEVEL with them --
// 0.067-0.075 ms against 0.069-0.072 ms per pair at batch 16, 6194-6229 against 6202-6216 frames/s end to end
// (profiles/r04_block_fused_AB.txt) -- so it is OFF by default (y3_options.fuse_block = 0; 1 = where it fills the chip,
// 2 = wherever supported: tests).  Why it does not win is recorded there and in profiles/HISTORY.md 3.1d: one workgroup per CU with
// all 160 KiB of LDS means (a) the chip reads x, computes, and writes z in lock step, so the 16 us of HBM time of a pair
// overlap with nothing, and (b) the weight ring has three slots, i.e. one K-step of lead, for tiles that all 256
// workgroups want in the same K-step.
//
// A workgroup owns a TW x TH rectangle of output pixels (<= 384 = 24 MFMA fragments of 16) of one frame and ALL output
// channels.  At 76 x 76 the rectangle is 19 x 19: sixteen per frame, so a batch of 16 is exactly one workgroup per CU
// of an MI355X -- no partial last round, no pixels computed beyond the frame, 1.22x recompute of the 1x1 for the
// one-pixel border the 3x3 needs.
//
//   phase A  1x1 over the (TW+2) x (TH+2) input patch (<= 448 rows = 28 fragments): x streams through LDS in 32-channel
//            sub-chunks (four 28-KiB buffers with 64-byte rows, LDS-DMA three sub-steps ahead, source-side XOR swizzle), the
//            1x1's weights in 8-KiB sub-chunks; one MFMA K-step and one barrier per sub-chunk; every wave keeps 7 x 4
//            accumulator tiles; then scale / bias / leaky -> bf16 -> the "mid" image in LDS (over the x buffers): two
//            planes of 64 channels, one 128-byte row per patch pixel, rows outside the frame are ZERO (the 3x3's padding).
//   phase B  3x3 from the mid image, 128 output channels at a time: tap (ky, kx) of output pixel (r, c) is mid row
//            (r + ky)(TW + 2) + c + kx -- rows never wrap, so nothing is masked.  Only the 16-KiB weight tile of a K-step
//            (one tap x 64 channels x 128 output channels) comes from memory: three-slot ring, one barrier per step,
//            weights(it + 2) issued between the K-halves of step it and waited for at its end; the first fragments of step
//            it + 1 are read during the second K-half of step it.  The 3x3's weights are pulled into L2 once, when phase A
//            is over (see warm_l2).
//            K order: channel chunk outermost, tap innermost -- the order of every other MFMA conv kernel here, so the
//            result equals the two separate launches bit for bit.
//   write-out straight from the accumulators: the weight rows are handed to the MFMA in y3_pair_perm order, so a lane holds
//            eight consecutive channels of its pixel -> scale / bias / leaky (+ shortcut operand, read from x) -> one
//            16-byte store.
//
// Eight waves, all of them load AND compute (no loader waves: 24 accumulator tiles + double-buffered fragments need more
// than the 168 registers a twelve-wave workgroup leaves a wave).  The mid image's swizzle key is the OUTPUT raster index
// mod 8 (TW * row + col of the patch pixel), not the patch row mod 8: the sixteen pixels of a fragment and all nine taps
// then see eight consecutive keys with alternating row parity whatever TW is, i.e. conflict-free ds_read_b128
// (MI355X_MICROARCH.md, LDS; measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 3 %, all from the mid image's writes).
#include "common.h"
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int kHaloRows = 448;                  // patch rows a workgroup can hold (28 fragments)
constexpr int kPlane = kHaloRows * 128;         // one x buffer / one mid plane: 57344 bytes
constexpr int kSlot = 128 * 128;                // one weight tile: 128 rows x 128 bytes
constexpr int kRingOff = 2 * kPlane;
constexpr int kBlockLds = kRingOff + 3 * kSlot; // 163840 = all of a CU's LDS
constexpr int kNT = 512;
constexpr int kMA = 7;                          // phase A: fragments per wave (4 pixel groups x 7 = 28)
constexpr int kMB = 6;                          // phase B: fragments per wave (4 pixel groups x 6 = 24 -> 384 pixels)
constexpr int kSub = kHaloRows * 64;            // one 32-channel sub-chunk of the x patch: 28672 bytes (four of them in flight)
static_assert(4 * kSub == 2 * kPlane, "the x buffers of phase A are the mid image of phase B");
static_assert(kBlockLds == 160 * 1024, "LDS budget");

struct BlockArgs {
  const char *x;      // 1x1 input (B, H, W, x_ld) bf16; also the shortcut operand when `res`
  int H, W, x_ld, ns; // ns = Cin / 64 channel chunks of the 1x1
  const char *w1; int k_ld1; const float *sc1, *bi1;   // 1x1: [>= 128][k_ld1]
  const char *w3; int k_ld3; const float *sc3, *bi3;   // 3x3: [Cout][k_ld3], k = tap * 128 + ci
  char *out; int out_ld, npass;                        // npass = Cout / 128
  int res;            // 1: add x (shortcut)
  int leaky1, leaky3;
  int TW, TH, tiles_x, tiles_y;
  uint32_t inv_pw, inv_tw;   // n / d == (n * inv) >> 16 for n < 512 (d = TW + 2, TW)
};

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ uint32_t lane32(uint32_t v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }

template <typename T>
__global__ __launch_bounds__(kNT) void conv_block_fused_kernel(BlockArgs p) {
  auto mma = [](const u32x4 &w, const u32x4 &x, const f32x4 &acc) { return y3_mfma16<T>(w, x, acc); };
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int pg = wave >> 1, cg = wave & 1;            // pixel group (4), channel group (2 x 64 channels)

  int tile = y3_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = tile % p.tiles_x;
  tile /= p.tiles_x;
  const int ty = tile % p.tiles_y, b = tile / p.tiles_y;
  const int oy0 = ty * p.TH, ox0 = tx * p.TW;
  const int PW = p.TW + 2;
  const int nhalo = PW * (p.TH + 2), npx = p.TW * p.TH;

  // ---- what this thread moves ----
  // phase A: x and the 1x1's weights in 32-channel sub-chunks, 64-byte LDS rows: a pass of the 512 threads fills 128
  // rows, thread t -> row t >> 2 of the pass, 16-byte piece (t & 3) ^ ((row >> 1) & 3) of the row's 64 bytes (source-side
  // swizzle: four consecutive rows are one 256-byte bank row; conflict-free ds_read_b128 of a fragment's 16 rows)
  const int arow = tid >> 2, apc16 = ((tid & 3) ^ ((tid >> 3) & 3)) << 4;
  constexpr int kAX = 4;                              // passes per x sub-chunk (the last one: rows 384..447, waves 0-3 only)
  const int n_ax = wave < 4 ? kAX : kAX - 1;
  uint32_t xoff[kAX];
#pragma unroll
  for (int i = 0; i < kAX; ++i) {
    int row = i * 128 + arow;
    row = row < nhalo ? row : nhalo - 1;              // rows past the patch: any valid address (their mid rows are unused)
    const int hr = (int)(((uint32_t)row * p.inv_pw) >> 16), hc = row - hr * PW;
    int gy = oy0 - 1 + hr, gx = ox0 - 1 + hc;         // pixels outside the frame: clamped (their mid rows are forced to zero)
    gy = gy < 0 ? 0 : (gy >= p.H ? p.H - 1 : gy);
    gx = gx < 0 ? 0 : (gx >= p.W ? p.W - 1 : gx);
    xoff[i] = (uint32_t)((b * p.H + gy) * p.W + gx) * (uint32_t)(p.x_ld * 2) + (uint32_t)apc16;
  }
  const uint32_t w1off = (uint32_t)(y3_pair_perm(arow) * p.k_ld1 * 2 + apc16);   // LDS row j carries channel y3_pair_perm(j)
  // phase B: 128-byte rows, thread t -> row t >> 3 (+ 64 in the second pass), piece (t & 7) ^ (row & 7)
  const int lrow = tid >> 3, kc16 = ((tid & 7) ^ (lrow & 7)) << 4;
  uint32_t w3off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) w3off[i] = (uint32_t)(y3_pair_perm(i * 64 + lrow) * p.k_ld3 * 2 + kc16);
  // sub-chunk j of x -> buffer j & 3 (28 KiB each), of the 1x1's weights -> 8-KiB slot j & 3 of the first 32 KiB of the ring
  auto issue_a = [&](int j) {
    // (lane32: the zero-extension of the lane's offset must happen in the block of the load -- hoisted out of a loop it
    // turns the load into its 64-bit-vector-address form, two registers per offset that end up spilled)
    const char *base = p.x + j * 64;
    char *dst = smem + (j & 3) * kSub + wave * 1024;
#pragma unroll
    for (int i = 0; i < kAX - 1; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void *)(base + lane32(xoff[i])), (lds_void *)(dst + i * (kNT * 16)), 16, 0, 0);
    if (wave < 4)
      __builtin_amdgcn_global_load_lds((gbl_void *)(base + lane32(xoff[kAX - 1])), (lds_void *)(dst + (kAX - 1) * (kNT * 16)), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void *)(p.w1 + j * 64 + lane32(w1off)), (lds_void *)(smem + kRingOff + (j & 3) * 8192 + wave * 1024), 16, 0, 0);
  };
  // K-step `it` of phase B: output-channel pass np = it / 18, then channel chunk (2), then tap (9); ring slot (it + 2) % 3
  auto issue_w3 = [&](int np, int c, int tap, int slot) {
    const char *base = p.w3 + ((long long)np * 128 * p.k_ld3 + tap * 128 + c * 64) * 2;
    char *dst = smem + kRingOff + slot * kSlot + wave * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void *)(base + lane32(w3off[i])), (lds_void *)(dst + i * (kNT * 16)), 16, 0, 0);
  };
  // at most n loads of this wave still in flight (n = whole sub-chunks: 5 loads each in waves 0-3, 4 in waves 4-7)
  auto wait_chunks = [&](int n) {
    if (n_ax == kAX) { if (n >= 2) wait_vm<10>(); else if (n == 1) wait_vm<5>(); else wait_vm<0>(); }
    else { if (n >= 2) wait_vm<8>(); else if (n == 1) wait_vm<4>(); else wait_vm<0>(); }
  };

  Y3_STAMP_DECL
  issue_w3(0, 0, 0, 2);
  issue_a(0);
  issue_a(1);
  issue_a(2);

  // L2 warm-up of the 3x3 weights (288 KiB per 128 output channels), issued when phase A's reads of x are over.  Every workgroup of the chip reads the same 16-KiB weight tile in
  // the same K-step, each tile once per launch: without this, each of those reads is an L2 miss that all 32 CUs of an XCD
  // wait for (2-3 k cycles against ~0.7 k for a hit: the K-step's lead of one step is then too short).  Nothing else
  // goes through L2 during phase B but those tiles and, at the end of a pass, the output stores.  The workgroups
  // of an XCD (dealt round-robin: XCD = blockIdx & 7) share the pass's 2304 cache lines; one dword per line, discarded.
  uint32_t sink = 0;   // destination of the warm-up loads: must stay allocated until they have returned (tied into the wait below)
  auto warm_l2 = [&](int np_w) {
    const int per_xcd = (gridDim.x + 7) >> 3, mine = blockIdx.x >> 3;
    const int lines = 128 * 18;                                  // 128 rows x 2304 bytes of K (9 taps x 128 channels)
    const int share = (lines + per_xcd - 1) / per_xcd;
    for (int l = mine * share + tid; l < (mine + 1) * share && l < lines; l += kNT) {
      const int row = l / 18, seg = l - row * 18;
      const uint32_t off = (uint32_t)((np_w * 128 + row) * p.k_ld3 * 2 + seg * 128);
      asm volatile("global_load_dword %0, %1, %2" : "+v"(sink) : "v"(off), "s"(p.w3) : "memory");
    }
  };

  // ---- phase A: mid = leaky(bn(conv1x1(x))) over the patch ----
  const int pA = pg * kMA * 16 + fr;                  // patch row of fragment 0; fragment mi is 16 mi rows further
  const int aswz = (fq ^ ((fr >> 1) & 3)) << 4;
  const int a_rd = pA * 64 + aswz;                    // + mi * 1024
  const int b_rdA = (cg * 64 + fr) * 64 + aswz;       // + ni * 1024
  const int b_row = cg * 64 + fr;
  const int b_rd = b_row * 128 + ((fq ^ (b_row & 7)) << 4);   // phase B
  // where this lane's results go in the mid image, and whether the pixel lies inside the frame
  int midw[kMA];
  uint32_t inside = 0;
#pragma unroll
  for (int mi = 0; mi < kMA; ++mi) {
    const int pr = pA + mi * 16;
    const int hr = (int)(((uint32_t)pr * p.inv_pw) >> 16), hc = pr - hr * PW;
    const int key = (p.TW * hr + hc) & 7;
    midw[mi] = cg * kPlane + pr * 128 + ((fq ^ key) << 4);
    const bool in = pr < nhalo && (unsigned)(oy0 - 1 + hr) < (unsigned)p.H && (unsigned)(ox0 - 1 + hc) < (unsigned)p.W;
    inside |= (in ? 1u : 0u) << mi;
  }
  {
    f32x4 acc[kMA][4];
#pragma unroll
    for (int mi = 0; mi < kMA; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nsub = 2 * p.ns;
#pragma unroll 1
    for (int j = 0; j < nsub; ++j) {
      // issue order: w3(0) c0 c1 c2 | c3 | c4 ...: sub-chunk j must have landed, j + 1 and j + 2 may fly
      const int after = nsub - 1 - j;
      wait_chunks(after < 2 ? after : 2);
      __builtin_amdgcn_s_barrier();                   // sub-chunk j is in LDS; everyone is done with sub-step j - 1
      Y3_STAMP(0);
      if (j + 3 < nsub) issue_a(j + 3);               // into the buffers of sub-step j - 1
      const char *xb = smem + (j & 3) * kSub, *wb = smem + kRingOff + (j & 3) * 8192;
      u32x4 xf[kMA], wf[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) wf[ni] = *reinterpret_cast<const u32x4 *>(wb + b_rdA + ni * 1024);
#pragma unroll
      for (int mi = 0; mi < kMA; ++mi) xf[mi] = *reinterpret_cast<const u32x4 *>(xb + a_rd + mi * 1024);
#pragma unroll
      for (int mi = 0; mi < kMA; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mma(wf[ni], xf[mi], acc[mi][ni]);
      wait_lgkm0();
      Y3_STAMP(1);
    }
    __builtin_amdgcn_s_barrier();                     // every read of the x buffers and of the 1x1's weights is done
    issue_w3(0, 0, 1, 0);                             // K-step 1 of phase B -> slot 0
    for (int npw = 0; npw < p.npass; ++npw) warm_l2(npw);   // (their latency runs under the mid image's write)
    // scale / bias / leaky -> bf16 -> mid image (over the x buffers)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int ch0 = (cg * 2 + k) * 32 + fq * 8;     // this lane's eight channels of fragment pair k
      const f32x4 sl = *reinterpret_cast<const f32x4 *>(p.sc1 + ch0), sh = *reinterpret_cast<const f32x4 *>(p.sc1 + ch0 + 4);
      const f32x4 bl = *reinterpret_cast<const f32x4 *>(p.bi1 + ch0), bh = *reinterpret_cast<const f32x4 *>(p.bi1 + ch0 + 4);
#pragma unroll
      for (int mi = 0; mi < kMA; ++mi) {
        float v[8];
        y3_bn_leaky8(v, acc[mi][2 * k], acc[mi][2 * k + 1], sl, sh, bl, bh, p.leaky1 != 0);
        u32x4 ov = y3_pack8<T>(v);
        if (!((inside >> mi) & 1u)) ov = u32x4{0u, 0u, 0u, 0u};   // the 3x3's zero padding
        *reinterpret_cast<u32x4 *>(smem + (midw[mi] ^ (k * 64))) = ov;
      }
    }
  }

  // ---- phase B: 3x3 from the mid image ----
  int p0b[kMB];        // byte offset of the mid row of tap (0, 0) of this lane's pixel of fragment mi
#pragma unroll
  for (int mi = 0; mi < kMB; ++mi) {
    const int q = (pg * kMB + mi) * 16 + fr;
    const int qc = q < npx ? q : npx - 1;
    const int r = (int)(((uint32_t)qc * p.inv_tw) >> 16), c = qc - r * p.TW;
    p0b[mi] = (r * PW + c) * 128;
  }
  // (b * H + oy) * W + ox of that pixel (of the tile's first pixel if it is not stored): recomputed at every write-out
  // rather than kept in six registers through the K loop
  auto out_px = [&](int mi, bool &ok) {
    const int q = (pg * kMB + mi) * 16 + fr;
    const int qc = q < npx ? q : npx - 1;
    const int r = (int)(((uint32_t)qc * p.inv_tw) >> 16), c = qc - r * p.TW;
    const int oy = oy0 + r, ox = ox0 + c;
    ok = q < npx && oy < p.H && ox < p.W;
    return ok ? (b * p.H + oy) * p.W + ox : (b * p.H + oy0) * p.W + ox0;
  };
  const char *ring = smem + kRingOff;
  const int nit = p.npass * 18;
  f32x4 acc[kMB][4];
  // pixel fragments of K-half g of a step come from mid plane c at tap shift (ky, kx); weight fragments from ring slot `slot`
  auto tap_off = [&](int c, int tap) {
    const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
    return ((fq ^ ((fr + p.TW * ky + kx) & 7)) << 4) + (ky * PW + kx) * 128 + c * kPlane;
  };
  auto read_x = [&](u32x4 (&xf)[kMB], int toff, int g) {
#pragma unroll
    for (int mi = 0; mi < kMB; ++mi) xf[mi] = *reinterpret_cast<const u32x4 *>(smem + ((p0b[mi] + toff) ^ (g * 64)));
  };
  auto read_x_part = [&](u32x4 (&xf)[kMB], int toff, auto lo, auto hi) {   // fragments lo .. hi - 1 of K-half 0
#pragma unroll
    for (int mi = decltype(lo)::value; mi < decltype(hi)::value; ++mi) xf[mi] = *reinterpret_cast<const u32x4 *>(smem + (p0b[mi] + toff));
  };
  auto read_w = [&](u32x4 (&wf)[4], int slot, int g) {
    const char *wb = ring + slot * kSlot + (b_rd ^ (g * 64));
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) wf[ni] = *reinterpret_cast<const u32x4 *>(wb + ni * 2048);
  };
  auto mma_all = [&](const u32x4 (&xf)[kMB], const u32x4 (&wf)[4]) {   // pixel fragment outermost: the first eight need xf[0], xf[1] only
#pragma unroll
    for (int mi = 0; mi < kMB; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mma(wf[ni], xf[mi], acc[mi][ni]);
  };
  // the younger wave of each SIMD loses every issue arbitration to its older partner: a static priority evens the pair out
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  Y3_STAMP(2);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) :: "memory");
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();                       // mid image complete; weights of K-steps 0 and 1 in LDS
  Y3_STAMP(3);
  u32x4 xf0[kMB], wf0[4], xf1[kMB], wf1[4];
  int np = 0, c = 0, tap = 0, slot = 2;
#pragma unroll 1
  for (int it = 0; it < nit; ++it) {
    if (it) __builtin_amdgcn_s_barrier();             // B(it): step it - 1 is over everywhere; weights(it + 1) are in LDS
    Y3_STAMP(4);
    const bool last = tap == 8 && c == 1;             // last K-step of an output-channel pass
    const bool first = tap == 0 && c == 0;
    int tap1 = tap + 1, c1 = c, np1 = np;
    if (tap1 == 9) { tap1 = 0; if (++c1 == 2) { c1 = 0; ++np1; } }
    int tap2 = tap1 + 1, c2 = c1, np2 = np1;
    if (tap2 == 9) { tap2 = 0; if (++c2 == 2) { c2 = 0; ++np2; } }
    const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
    const int toff = tap_off(c, tap);
    constexpr std::integral_constant<int, 0> I0{};
    constexpr std::integral_constant<int, 2> I2{};
    constexpr std::integral_constant<int, kMB> IM{};
    if (first) {                                      // first step of a pass: nothing was prefetched (registers: the write-out's)
      read_w(wf0, slot, 0);
      read_x_part(xf0, toff, I0, I2);
#pragma unroll
      for (int mi = 0; mi < kMB; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);
    read_x_part(xf0, toff, I2, IM);                   // (fragments 0, 1 and the weights of this K-half were read a step ago)
    read_x(xf1, toff, 1);
    read_w(wf1, slot, 1);
    mma_all(xf0, wf0);
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
    for (int i = 0; i < kMB + 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // the weight tile of step it + 2 goes out between the K-halves (right after B(it) it measured slower: the LDS-DMA issue
    // holds the wave 60-190 cycles per piece before its first MFMAs: profiles/r04i_block_fused_v6_*.txt)
    if (it + 2 < nit) issue_w3(np2, c2, tap2, slot2);  // slot2 == slot of step it - 1: free since B(it)
    __builtin_amdgcn_sched_barrier(0);
    if (!last) {                                      // next step's first K-half (its weights landed before B(it)): what its
      read_w(wf0, slot1, 0);                          // first eight MFMAs need
      read_x_part(xf0, tap_off(c1, tap1), I0, I2);
    }
    mma_all(xf1, wf1);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    Y3_STAMP(5);
    if (last) {
      // ---- write-out of output channels np * 128 .. + 127, straight from the accumulators ----
      // (one copy of the code per kind of block, so that the shortcut registers are defined and used in the same branch:
      // with separate `if (res)` tests around loads and uses they stay allocated through the whole K loop)
      auto write_out = [&](auto with_res) {
        constexpr bool RES = decltype(with_res)::value;
        int pxo[kMB];
        uint32_t stored = 0;
#pragma unroll
        for (int mi = 0; mi < kMB; ++mi) {
          bool ok;
          pxo[mi] = out_px(mi, ok);
          stored |= (ok ? 1u : 0u) << mi;
        }
        u32x4 rv[2][kMB];
        if constexpr (RES) {
          // shortcut operand (from x): both halves go out at once, into the registers this step's fragments have just left
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int mi = 0; mi < kMB; ++mi) {
              const uint32_t ro = (uint32_t)(pxo[mi] * p.x_ld + np * 128 + (cg * 2 + k) * 32 + fq * 8) * 2u;   // < 2^32: launcher
              asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rv[k][mi]) : "v"(ro), "s"(p.x) : "memory");
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int ch0 = np * 128 + (cg * 2 + k) * 32 + fq * 8;
          const f32x4 sl = *reinterpret_cast<const f32x4 *>(p.sc3 + ch0), sh = *reinterpret_cast<const f32x4 *>(p.sc3 + ch0 + 4);
          const f32x4 bl = *reinterpret_cast<const f32x4 *>(p.bi3 + ch0), bh = *reinterpret_cast<const f32x4 *>(p.bi3 + ch0 + 4);
          if constexpr (RES) {
            // younger than half k of the shortcut loads: at most the other half / the six stores of k == 0 and the four
            // scale / bias loads above (which the compiler waits for itself)
            if (k == 0) asm volatile("s_waitcnt vmcnt(10)" : "+v"(rv[0][0]), "+v"(rv[0][1]), "+v"(rv[0][2]), "+v"(rv[0][3]), "+v"(rv[0][4]), "+v"(rv[0][5]) :: "memory");
            else asm volatile("s_waitcnt vmcnt(10)" : "+v"(rv[1][0]), "+v"(rv[1][1]), "+v"(rv[1][2]), "+v"(rv[1][3]), "+v"(rv[1][4]), "+v"(rv[1][5]) :: "memory");
          }
#pragma unroll
          for (int mi = 0; mi < kMB; ++mi) {
            float v[8];
            y3_bn_leaky8(v, acc[mi][2 * k], acc[mi][2 * k + 1], sl, sh, bl, bh, p.leaky3 != 0);
            if constexpr (RES) y3_add8<T>(v, rv[k][mi]);
            if ((stored >> mi) & 1u) *reinterpret_cast<u32x4 *>(p.out + lane32((uint32_t)(pxo[mi] * p.out_ld + ch0) * 2u)) = y3_pack8<T>(v);
          }
        }
      };
      if (p.res) write_out(std::true_type{});
      else write_out(std::false_type{});
      // weights(it + 2) went out before the write-out's loads and stores.  A counted wait ("only the twelve stores may still
      // fly") would rest on how many stores the wave really issued -- they are predicated per lane, a wave whose pixels are
      // all outside the frame issues none -- and was right only because the compiler's waits on the scale / bias loads, which
      // return in order behind the weight pieces, happened to cover them.  One full drain per 18 K-steps instead.
      wait_vm<0>();
    } else {
      wait_vm<0>();                                   // weights(it + 2) (and everything older) landed: ready for B(it + 1)
    }
    Y3_STAMP(6);
    np = np1; c = c1; tap = tap1; slot = slot1;
  }
  Y3_STAMP_COUNT();
}

// rectangle of output pixels per workgroup: TW x TH <= 384, (TW + 2)(TH + 2) <= 448; the choice that needs the fewest
// rounds of workgroups (one per CU), then the one that computes the fewest pixels it does not store
bool choose_tile(int H, int W, int batch, int n_cu, int &TW, int &TH, double &eff) {
  long long best_rounds = 0, best_px = 0;
  TW = TH = 0;
  for (int tw = 8; tw <= 24; ++tw) {
    int th = 384 / tw;
    while (th > 0 && (tw + 2) * (th + 2) > kHaloRows) --th;
    if (th > H) th = H;
    if (tw > W || th < 4) continue;
    // no taller than needed for the same number of tile rows
    const int rows = y3_ceil_div(H, th);
    th = y3_ceil_div(H, rows);
    const long long tiles = (long long)y3_ceil_div(W, tw) * rows * batch;
    const long long rounds = (tiles + n_cu - 1) / n_cu;
    if (TW == 0 || rounds < best_rounds || (rounds == best_rounds && tiles < best_px)) {
      TW = tw; TH = th; best_rounds = rounds; best_px = tiles;
    }
  }
  if (TW == 0) return false;
  eff = (double)H * W * batch / ((double)best_rounds * n_cu * 384.0);
  return true;
}

}  // namespace

// op0: 1x1 conv Cin -> 128 whose output only op1 reads; op1: 3x3 stride-1 conv 128 -> Cout, optionally with the shortcut
// operand == op0's input.  fuse_block: 0 never [default], 1 where the rectangles fill the chip, 2 wherever supported.
bool y3_conv_block_fused_supported(const y3_op &op0, const y3_op &op1) {
  const int mode = y3_opt().fuse_block;
  if (!mode) return false;
  if (op0.kind != Y3_OP_CONV || op1.kind != Y3_OP_CONV || !y3_is16(op0.dtype) || op1.dtype != op0.dtype) return false;
  if (op0.ksize != 1 || op0.stride != 1 || op0.in_c % 64 != 0 || op0.in_c < 192 || op0.out_c != 128) return false;
  if (op1.ksize != 3 || op1.stride != 1 || op1.pad != 1 || op1.in_c != 128 || op1.out_c % 128 != 0) return false;
  const uint32_t bad = Y3_F_OUT_F32 | Y3_F_IN_NCHW_F32 | Y3_F_IN_NHWC_U8BGR | Y3_F_PLAN_INPUT;
  if ((op0.flags & (bad | Y3_F_RESIDUAL)) || (op1.flags & bad)) return false;
  if (op1.d_in != op0.d_out || op1.in_ld != op0.out_ld) return false;
  // z must not overlap x: a workgroup writes its rectangle of z while its neighbours still read those pixels of x as their
  // one-pixel border (phase A), and nothing orders workgroups of different rounds / streams.  The arena planner frees x after
  // the 1x1 of a pair WITHOUT shortcut and may hand the same bytes to z (unfused that is safe: the 1x1's output is a
  // separate buffer): such pairs run as two launches.  
  {
    const uintptr_t x0 = (uintptr_t)op0.d_in, z0 = (uintptr_t)op1.d_out;
    const uintptr_t xb = (uintptr_t)op0.batch * op0.in_h * op0.in_w * op0.in_ld * 2, zb = (uintptr_t)op1.batch * op1.out_h * op1.out_w * op1.out_ld * 2;
    if (x0 && z0 && x0 < z0 + zb && z0 < x0 + xb) return false;
  }
  if (op1.flags & Y3_F_RESIDUAL)
    if (op1.d_res != op0.d_in || op1.res_ld != op0.in_ld || op1.out_c != op0.in_c) return false;
  if (op0.in_h != op1.in_h || op0.in_w != op1.in_w || op0.batch != op1.batch) return false;
  if (op0.out_h != op0.in_h || op0.out_w != op0.in_w || op1.out_h != op1.in_h || op1.out_w != op1.in_w) return false;
  if (op0.in_ld % 8 != 0 || op1.out_ld % 8 != 0) return false;
  if (op0.k_ld < op0.in_c || op1.k_ld < 9 * 128 || op0.cout_pad < 128 || op1.cout_pad < op1.out_c) return false;
  // 32-bit byte offsets into x, the output and the weights
  if ((long long)op0.batch * op0.in_h * op0.in_w * op0.in_ld * 2 >= (1ll << 32)) return false;
  if ((long long)op1.batch * op1.out_h * op1.out_w * op1.out_ld * 2 >= (1ll << 32)) return false;
  if ((long long)op1.cout_pad * op1.k_ld * 2 >= (1ll << 31) || (long long)op0.cout_pad * op0.k_ld * 2 >= (1ll << 31)) return false;
  int tw, th;
  double eff;
  if (!choose_tile(op0.in_h, op0.in_w, op0.batch, y3_device_cus(), tw, th, eff)) return false;
  return mode >= 2 || eff >= 0.7;
}

int y3_launch_conv_block_fused(const y3_op &op0, const y3_op &op1, hipStream_t s, const char **kernel_name, bool dry_run) {
  *kernel_name = Y3_KNAME(op0.dtype, "conv_block_fused_", "_x128");
  if (dry_run) return Y3_OK;
  BlockArgs a;
  a.x = static_cast<const char *>(op0.d_in);
  a.H = op0.in_h; a.W = op0.in_w; a.x_ld = op0.in_ld; a.ns = op0.in_c / 64;
  a.w1 = static_cast<const char *>(op0.d_weight); a.k_ld1 = op0.k_ld; a.sc1 = op0.d_scale; a.bi1 = op0.d_bias;
  a.w3 = static_cast<const char *>(op1.d_weight); a.k_ld3 = op1.k_ld; a.sc3 = op1.d_scale; a.bi3 = op1.d_bias;
  a.out = static_cast<char *>(op1.d_out); a.out_ld = op1.out_ld; a.npass = op1.out_c / 128;
  a.res = (op1.flags & Y3_F_RESIDUAL) ? 1 : 0;
  a.leaky1 = (op0.flags & Y3_F_LEAKY) ? 1 : 0;
  a.leaky3 = (op1.flags & Y3_F_LEAKY) ? 1 : 0;
  return y3_by_dtype16(op0.dtype, [&](auto tag) {
    typedef decltype(tag) T;
    static Y3DeviceOnce once;
    int n_cu = 0;
    {
      const int rc = once.run([]() -> int {
        Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_block_fused_kernel<T>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kBlockLds));
        return Y3_OK;
      }, &n_cu);
      if (rc != Y3_OK) return rc;
    }
    double eff;
    Y3_REQUIRE(choose_tile(a.H, a.W, op0.batch, n_cu, a.TW, a.TH, eff), "conv block %d: no tile shape", op0.block_idx);
    a.tiles_x = y3_ceil_div(a.W, a.TW);
    a.tiles_y = y3_ceil_div(a.H, a.TH);
    a.inv_pw = (65536u + (uint32_t)(a.TW + 2) - 1u) / (uint32_t)(a.TW + 2);
    a.inv_tw = (65536u + (uint32_t)a.TW - 1u) / (uint32_t)a.TW;
    const int grid = a.tiles_x * a.tiles_y * op0.batch;
    Y3_LAUNCH(conv_block_fused_kernel<T>, dim3(grid), dim3(kNT), kBlockLds, s, a);
    Y3_HIP_CHECK(hipGetLastError());
    return Y3_OK;
  });
}

Y3_STAMP_READER(y3_debug_stamps_block)
