// 3x3 stride-1 convolution with input-halo reuse (gfx950), bf16 / fp32, NHWC.
//
// Same contract as conv_igemm.hip (conv -> scale/bias -> LeakyReLU -> +residual; replaces
// /root/reference/yolov3/darknet.py:244-257 and the shortcut at :376-379), specialised for the
// layers that carry ~75 % of Darknet-53's FLOPs: 3x3, stride 1, pad 1, Cin a multiple of the
// K-tile (64 bf16 / 32 fp32 channels = 128 bytes).
//
// Why: the generic implicit GEMM re-reads every input pixel once per filter tap (9x) and every
// weight tile once per 128-pixel tile; measured, its K loop is bound by global->LDS latency and
// traffic (~10 TB/s of LDS-DMA at 25-30 % MFMA busy), not by the matrix cores.  Here a workgroup
// owns BM consecutive output pixels in raster order (b, y, x flattened) and stages, per 128-byte
// channel chunk, the BM + 2W + 2 input pixels that ALL nine taps of those outputs touch -- once.
// Tap (ky,kx) of output pixel p is halo row  p + ky*W + kx, so the nine K-steps of a chunk read the
// same LDS image at nine row offsets.  Row-wrap / image-border taps (the zero padding) are cleared
// in registers with a per-pixel 9-bit mask after the fragment read.  Only the weight tile (BN x 128
// B) changes per K-step; it streams through a 3-slot LDS ring two steps ahead.  Bytes moved per FLOP
// drop ~3x versus the 128x128 implicit GEMM.
//
// Pipeline (all LDS-DMA, global_load_lds_dwordx4, counted vmcnt, one barrier per K-step):
//   step it = chunk*9 + tap:  wait(all but the loads issued in step it-1) ; barrier ;
//                             issue weights(it+2) [+ one slice of the next chunk's halo] ; MFMAs(it)
// Workgroup: (BM/64) x 2 waves, wave tile 64 x 64 (4x4 MFMA 16x16 accumulators), BN = 128.
#include "common.h"

namespace {

struct HaloArgs {
  const char *in;
  const char *wgt;
  const float *scale;
  const float *bias;
  const char *res;
  char *out;
  const char *zero;
  int H, W, Cin, in_ld;
  int Cout, out_ld, res_ld;
  int M, HW;           // B*H*W, H*W
  int k_ld;
  int nchunks;         // Cin / BKE
  int n_tiles;
  int hr_pad;          // halo rows, padded to a multiple of the loader's rows-per-pass
  int na;              // loader passes per halo (= glds per thread per halo)
  int a_bytes;         // hr_pad * 128
  uint32_t mul_hw, sh_hw, mul_w, sh_w;   // n / d == (umulhi(n, mul) + n) >> sh  for n < 2^31
  uint32_t flags;
};

template <typename T>
struct MmaH;
template <>
struct MmaH<bf16_t> {
  static __device__ __forceinline__ void run(f32x4 &acc, const u32x4 &w, const u32x4 &x) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x),
                                                  acc, 0, 0, 0);
  }
};
template <>
struct MmaH<float> {
  static __device__ __forceinline__ void run(f32x4 &acc, const u32x4 &w, const u32x4 &x) {
    const f32x4 wf = __builtin_bit_cast(f32x4, w), xf = __builtin_bit_cast(f32x4, x);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], xf[j], acc, 0, 0, 0);
  }
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_n() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int BM, int NSB>
__global__ __launch_bounds__(BM * 2) void conv_halo3x3_kernel(HaloArgs p) {
  constexpr int BN = 128;
  constexpr int WAVES_M = BM / 64, WAVES_N = 2;
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int ES = sizeof(T);
  constexpr int BKE = 128 / ES;
  constexpr int RPP = NT / 8;                         // rows filled per loader pass
  constexpr int NB = (BN + RPP - 1) / RPP;            // weight-tile passes (glds per thread per K-step)
  constexpr int B_ROWS = NB * RPP;
  constexpr int B_BYTES = B_ROWS * 128;
  constexpr int D = NSB - 2;                          // K-steps of load latency the ring tolerates
  constexpr int MI = 4, NI = 4;
  static_assert(NSB == 3 || NSB == 4, "weight ring has 3 or 4 slots");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *sB = smem;                                    // [NSB][B_ROWS][128]
  char *sA = smem + NSB * B_BYTES;                    // [2][hr_pad][128]

  Y3_STAMP_DECL
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int fr = lane & 15, fq = lane >> 4;

  const int tile = y3_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / p.n_tiles) * BM;
  const int n0 = (tile % p.n_tiles) * BN;

  // ---- loader geometry -----------------------------------------------------------------------
  const int slot = tid & 7;
  const int row0 = tid >> 3;
  const int kc = slot ^ (row0 & 7);                   // RPP % 8 == 0, so (row & 7) == (row0 & 7)
  const long long q0 = (long long)m0 - p.W - 1;       // flattened input pixel of halo row 0

  auto issue_halo_pass = [&](int chunk, int pass, bool live = true) {
    const int row = row0 + pass * RPP;
    const long long q = q0 + row;
    const bool ok = live && q >= 0 && q < p.M;
    const char *src = ok ? p.in + (q * p.in_ld + (long long)chunk * BKE) * ES + kc * 16 : p.zero;
    char *dst = sA + (chunk & 1) * p.a_bytes + pass * (NT * 16) + wave * 1024;
    __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)dst, 16, 0, 0);
  };
  const char *b_src[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int r = row0 + i * RPP;
    b_src[i] = r < BN ? p.wgt + ((long long)(n0 + r) * p.k_ld) * ES + kc * 16 : nullptr;
  }
  auto issue_weights = [&](int it, int slot_it = -1) {  // it = chunk*9 + tap; K offset = (tap*Cin + chunk*BKE) elements
    const int chunk = it / 9, tap = it - chunk * 9;
    const long long koff = ((long long)tap * p.Cin + (long long)chunk * BKE) * ES;
    char *dst = sB + ((slot_it < 0 ? it : slot_it) % NSB) * B_BYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const char *src = b_src[i] ? b_src[i] + koff : p.zero;
      __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(dst + i * (NT * 16)), 16, 0, 0);
    }
  };

  // ---- prologue: get the first operands moving before anything else ------------------------------
  const int nit = p.nchunks * 9;
  for (int pass = 0; pass < p.na; ++pass) issue_halo_pass(0, pass);
#pragma unroll
  for (int j = 0; j <= D; ++j)
    if (j < nit) issue_weights(j);

  // per-lane tap validity of the 4 pixels this lane feeds to the MFMAs (closed form, no loops)
  uint32_t tapmask[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const uint32_t m = (uint32_t)(m0 + wm * 64 + mi * 16 + fr);
    uint32_t mask = 0u;
    if (m < (uint32_t)p.M) {
      const uint32_t img = (__umulhi(m, p.mul_hw) + m) >> p.sh_hw;
      const uint32_t rem = m - img * (uint32_t)p.HW;
      const uint32_t oy = (__umulhi(rem, p.mul_w) + rem) >> p.sh_w;
      const uint32_t ox = rem - oy * (uint32_t)p.W;
      const uint32_t vx = (ox >= 1u ? 1u : 0u) | 2u | (ox + 1u < (uint32_t)p.W ? 4u : 0u);
      mask = (oy >= 1u ? vx : 0u) | (vx << 3) | (oy + 1u < (uint32_t)p.H ? vx << 6 : 0u);
    }
    tapmask[mi] = mask;
  }
  const int oc_mine = tid & 15;  // this thread's 8-channel group in the write-out phase

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addressing: lane (fr, fq) reads 16-byte chunk (g*4 + fq) of row (base + fr [+ shift]);
  // mi / ni steps are +16 rows = +2048 bytes and do not change (row & 7), so they are immediates
  const int a_lane_row = wm * 64 + fr;
  const int b_lane_row = wn * 64 + fr;
  const int b_off0 = b_lane_row * 128 + (((0 + fq) ^ (b_lane_row & 7)) << 4);
  const int b_off1 = b_lane_row * 128 + (((4 + fq) ^ (b_lane_row & 7)) << 4);

  auto read_frags = [&](u32x4 (&xf)[MI], u32x4 (&wf)[NI], const char *aBuf, const char *bBuf, int a_shift, int g) {
    const int r0 = a_lane_row + a_shift;
    const char *ap = aBuf + r0 * 128 + (((g * 4 + fq) ^ (r0 & 7)) << 4);
    const char *bp = bBuf + (g ? b_off1 : b_off0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) xf[mi] = *reinterpret_cast<const u32x4 *>(ap + mi * 2048);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) wf[ni] = *reinterpret_cast<const u32x4 *>(bp + ni * 2048);
  };
  auto mma_all = [&](u32x4 (&xf)[MI], const u32x4 (&wf)[NI], int tap) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      if (!((tapmask[mi] >> tap) & 1u)) xf[mi] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) MmaH<T>::run(acc[mi][ni], wf[ni], xf[mi]);
  };
  // scheduling recipe for one half-step: the 8 fragment reads of the OTHER register set ride in the gaps of
  // this set's 16 MFMAs (bf16: 2 MFMAs per read; fp32: 8), instead of being issued as one LDS burst up front
  auto interleave = [&]() {
#pragma unroll
    for (int i = 0; i < MI + NI; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                          // one DS read
      __builtin_amdgcn_sched_group_barrier(0x008, sizeof(T) == 2 ? 2 : 8, 0);     // its share of the MFMAs
    }
  };

  // ---- first operands: halo(0), weights(0), weights(1) must have landed ----------------------------
  Y3_COARSE(0);
  if (D == 2 && nit > 2) wait_vmcnt<NB>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  Y3_COARSE(1);
  u32x4 xf0[MI], wf0[NI], xf1[MI], wf1[NI];
  read_frags(xf0, wf0, sA, sB, 0, 0);
  __builtin_amdgcn_s_waitcnt(0xC07F);

  // LDS-DMA instructions this thread issued in the previous step (step "-1" = the prologue, whose
  // only possibly unfinished loads are weights(2) of the 4-slot ring)
  int issued_prev = (D == 2 && nit > 2) ? NB : 0;
  int tap = 0, chunk = 0;
#pragma unroll 1
  for (int it = 0; it < nit; ++it) {
    {
      // weights(it+1) -- and the next chunk's halo when the next step starts it -- must have landed
      // before this barrier; with the 4-slot ring only the loads of step it-1 may still be in flight.
      // (Step 0 repeats the prologue's barrier on purpose: every trip then enters with the same
      // "fragment set 0 complete" state and hipcc emits counted LDS waits inside the loop.)
      if (D == 2) {
        if (issued_prev == NB + 1) wait_vmcnt<NB + 1>();
        else if (issued_prev == NB) wait_vmcnt<NB>();
        else if (issued_prev == 1) wait_vmcnt<1>();
        else wait_vmcnt<0>();
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
    }
    Y3_FINE(0);   // vmcnt wait + barrier
    // This step's LDS-DMA (halo slice first, weights second: only later steps' loads are younger than these
    // weights) is issued INSIDE half-step A's scheduling region, so the up-to-NB+1 DMA instructions -- ~150 cycles
    // each when all eight waves issue at once -- ride in MFMA gaps instead of forming a DMA-only phase.
    const char *aBuf = sA + (chunk & 1) * p.a_bytes;
    const int ky = (tap * 11) >> 5, kx = tap - ky * 3;         // tap / 3, tap % 3 for tap in 0..8
    __builtin_amdgcn_sched_barrier(0);
    // Branch-free on purpose (one basic block = one scheduling region): when there is nothing left to fetch the
    // same number of DMA instructions is still issued -- a repeated halo slice / the last weight tile again --
    // into LDS that nobody reads any more (the next-chunk halo buffer, the ring slot that is free by invariant).
    // The DMA calls sit BETWEEN the fragment reads in program order: LDS-DMA and ds_read both touch LDS, so the
    // scheduler keeps their relative order, and this is what lets the group barriers below spread them.
    {
      const int r0 = a_lane_row + ky * p.W + kx;
      const char *ap = aBuf + r0 * 128 + (((4 + fq) ^ (r0 & 7)) << 4);
      const char *bp = sB + (it % NSB) * B_BYTES + b_off1;
      const bool live = chunk + 1 < p.nchunks;
      const int itw = it + 1 + D < nit ? it + 1 + D : nit - 1;
      xf1[0] = *reinterpret_cast<const u32x4 *>(ap);
      xf1[1] = *reinterpret_cast<const u32x4 *>(ap + 2048);
      issue_halo_pass(chunk + 1, tap < p.na ? tap : p.na - 1, live);
      xf1[2] = *reinterpret_cast<const u32x4 *>(ap + 4096);
      xf1[3] = *reinterpret_cast<const u32x4 *>(ap + 6144);
      wf1[0] = *reinterpret_cast<const u32x4 *>(bp);
      wf1[1] = *reinterpret_cast<const u32x4 *>(bp + 2048);
      issue_weights(itw, it + 1 + D);
      wf1[2] = *reinterpret_cast<const u32x4 *>(bp + 4096);
      wf1[3] = *reinterpret_cast<const u32x4 *>(bp + 6144);
      issued_prev = NB + 1;
    }
    mma_all(xf0, wf0, tap);                                    // the first half's MFMAs cover all of the above
    // 2 DS reads, 4 MFMAs, 1 DMA | 2 DS, 4 MFMA | 2 DS, 4 MFMA, NB DMA | 2 DS, 4 MFMA   (bf16 counts)
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, sizeof(T) == 2 ? 4 : 16, 0);
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, sizeof(T) == 2 ? 4 : 16, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, sizeof(T) == 2 ? 4 : 16, 0);
    __builtin_amdgcn_sched_group_barrier(0x010, NB, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, sizeof(T) == 2 ? 4 : 16, 0);
    __builtin_amdgcn_sched_barrier(0);
    Y3_FINE(2);   // half-step A issued (8 reads + masks + 16 MFMAs)
    __builtin_amdgcn_s_waitcnt(0xC07F);   // second-half fragments (16 MFMAs old); keeps <= 8 LDS reads in flight
    Y3_FINE(3);   // lgkmcnt(0)
    const int tap_n = tap == 8 ? 0 : tap + 1;
    const int chunk_n = tap == 8 ? chunk + 1 : chunk;
    {                                                          // first half of the next step (harmless
      const int ky_n = (tap_n * 11) >> 5, kx_n = tap_n - ky_n * 3;   // in-bounds read after the last one)
      read_frags(xf0, wf0, sA + (chunk_n & 1) * p.a_bytes, sB + ((it + 1) % NSB) * B_BYTES, ky_n * p.W + kx_n, 0);
    }
    mma_all(xf1, wf1, tap);
    interleave();
    __builtin_amdgcn_sched_barrier(0);
    Y3_FINE(4);   // half-step B issued
    // the prefetched fragments have had 16 MFMAs of time; retiring them here (lgkmcnt(0) only, in a form
    // hipcc's wait-count pass understands) lets it issue the next step's first MFMAs without a wait
    __builtin_amdgcn_s_waitcnt(0xC07F);
    Y3_FINE(5);   // lgkmcnt(0)
    tap = tap_n;
    chunk = chunk_n;
  }
  __syncthreads();  // all operand reads done: LDS can hold the output tile
  Y3_COARSE(2);  // main loop

  // ---- epilogue: raw fp32 accumulators -> LDS (pixel rows, XOR-swizzled 16-byte chunks) -> every thread
  // finishes 8 consecutive channels of one pixel: scale/bias/LeakyReLU, + residual, one 16-byte store
  constexpr int SWZ = 15;
  constexpr int OCT_PER_ROW = BN / 8;
  constexpr int WR = BM * OCT_PER_ROW / NT;   // write-out steps per thread (8)
  float *sC = reinterpret_cast<float *>(smem);
  const bool leaky = p.flags & Y3_F_LEAKY;
  const bool has_res = p.flags & Y3_F_RESIDUAL;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int cl = wn * 64 + ni * 16 + fq * 4;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int pl = wm * 64 + mi * 16 + fr;
      *reinterpret_cast<f32x4 *>(sC + pl * BN + (((cl >> 2) ^ (pl & SWZ)) << 2)) = acc[mi][ni];
    }
  }
  const int co = n0 + oc_mine * 8;
  const f32x4 sc_lo = *reinterpret_cast<const f32x4 *>(p.scale + co);
  const f32x4 sc_hi = *reinterpret_cast<const f32x4 *>(p.scale + co + 4);
  const f32x4 bi_lo = *reinterpret_cast<const f32x4 *>(p.bias + co);
  const f32x4 bi_hi = *reinterpret_cast<const f32x4 *>(p.bias + co + 4);
  u32x4 resv[WR];
  if (has_res) {
#pragma unroll
    for (int j = 0; j < WR; ++j) {
      const int m = m0 + (tid >> 4) + j * (NT / 16);
      const char *rp = p.res + ((long long)m * p.res_ld + co) * ES;
      if constexpr (sizeof(T) == 2) {
        resv[j] = m < p.M ? *reinterpret_cast<const u32x4 *>(rp) : u32x4{0u, 0u, 0u, 0u};
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < WR; ++j) {
    const int pl = (tid >> 4) + j * (NT / 16);
    const int m = m0 + pl;
    if (m >= p.M) continue;
    const f32x4 lo = *reinterpret_cast<const f32x4 *>(sC + pl * BN + (((2 * oc_mine) ^ (pl & SWZ)) << 2));
    const f32x4 hi = *reinterpret_cast<const f32x4 *>(sC + pl * BN + (((2 * oc_mine + 1) ^ (pl & SWZ)) << 2));
    float v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = lo[r] * sc_lo[r] + bi_lo[r];
      v[4 + r] = hi[r] * sc_hi[r] + bi_hi[r];
    }
    if (leaky) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : Y3_LEAKY_SLOPE * v[r];
    }
    if (has_res) {
      if constexpr (sizeof(T) == 2) {
        const bf16x8 rv = __builtin_bit_cast(bf16x8, resv[j]);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += (float)rv[r];
      } else {
        const float *rp = reinterpret_cast<const float *>(p.res) + (long long)m * p.res_ld + co;
        const f32x4 r0 = *reinterpret_cast<const f32x4 *>(rp), r1 = *reinterpret_cast<const f32x4 *>(rp + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] += r0[r]; v[4 + r] += r1[r]; }
      }
    }
    T *op = reinterpret_cast<T *>(p.out) + (long long)m * p.out_ld + co;
    if constexpr (sizeof(T) == 2) {
      bf16x8 ov;
#pragma unroll
      for (int r = 0; r < 8; ++r) ov[r] = (bf16_t)v[r];
      *reinterpret_cast<bf16x8 *>(op) = ov;
    } else {
      *reinterpret_cast<f32x4 *>(op) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4 *>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
  }
  Y3_COARSE(3);  // epilogue
  Y3_STAMP_COUNT();
}

// ------------------------------------------------------------------------------------------------
// Ping-pong variant (BM = 256, 8 waves = 2 per SIMD).  Measured on the kernel above: the two waves of
// a SIMD run in lockstep -- both issue LDS-DMA / LDS reads, then both want the matrix pipe -- so a
// K-step costs ~2100 cycles for 1024 cycles of MFMA.  Here every wave alternates a LOAD segment
// (its share of the LDS-DMA for step s+2, then the 16 fragment reads of step s into registers)
// with a COMPUTE segment (32 MFMAs of step s, operands already in registers); waves 4-7 run one
// segment behind waves 0-3 (they share SIMDs pairwise), and one workgroup barrier per segment keeps
// the alternation, so on every SIMD one wave feeds the matrix pipe while its partner loads.
// Ring: 3 weight slots; weights(s+2) are issued in load segment s; every wave drains its older loads
// at the end of each odd half-step (group 0: after compute(s), group 1: after load(s)).
template <typename T>
__global__ __launch_bounds__(512) void conv_halo3x3_pp_kernel(HaloArgs p) {
  constexpr int BM = 256, BN = 128, NT = 512, NSB = 3;
  constexpr int ES = sizeof(T);
  constexpr int BKE = 128 / ES;
  constexpr int RPP = NT / 8;
  constexpr int NB = BN / RPP;                        // 2 LDS-DMA instructions per thread per weight tile
  constexpr int B_BYTES = BN * 128;
  constexpr int MI = 4, NI = 4;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *sB = smem;                                    // [NSB][128][128]
  char *sA = smem + NSB * B_BYTES;                    // [2][hr_pad][128]

  Y3_STAMP_DECL
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  // waves w and w+4 share a SIMD; readfirstlane makes the group id provably wave-uniform, so the
  // group-dependent barriers below are real scalar branches (never executed under an empty EXEC mask)
  const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;

  const int tile = y3_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / p.n_tiles) * BM;
  const int n0 = (tile % p.n_tiles) * BN;

  const int slot = tid & 7;
  const int row0 = tid >> 3;
  const int kc = slot ^ (row0 & 7);
  const long long q0 = (long long)m0 - p.W - 1;

  auto issue_halo_pass = [&](int chunk, int pass) {
    const long long q = q0 + row0 + pass * RPP;
    const bool ok = q >= 0 && q < p.M;
    const char *src = ok ? p.in + (q * p.in_ld + (long long)chunk * BKE) * ES + kc * 16 : p.zero;
    char *dst = sA + (chunk & 1) * p.a_bytes + pass * (NT * 16) + wave * 1024;
    __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)dst, 16, 0, 0);
  };
  const char *b_src0 = p.wgt + ((long long)(n0 + row0) * p.k_ld) * ES + kc * 16;
  const char *b_src1 = b_src0 + (long long)RPP * p.k_ld * ES;
  auto issue_weights = [&](int chunk, int tap, int slot_idx) {
    const long long koff = ((long long)tap * p.Cin + (long long)chunk * BKE) * ES;
    char *dst = sB + slot_idx * B_BYTES + wave * 1024;
    __builtin_amdgcn_global_load_lds((gbl_void *)(b_src0 + koff), (lds_void *)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void *)(b_src1 + koff), (lds_void *)(dst + NT * 16), 16, 0, 0);
  };

  const int nit = p.nchunks * 9;
  for (int pass = 0; pass < p.na; ++pass) issue_halo_pass(0, pass);
  issue_weights(0, 0, 0);
  issue_weights(0, 1, 1);   // nit >= 18: the launcher requires at least two channel chunks

  uint32_t tapmask[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const uint32_t m = (uint32_t)(m0 + wm * 64 + mi * 16 + fr);
    uint32_t mask = 0u;
    if (m < (uint32_t)p.M) {
      const uint32_t img = (__umulhi(m, p.mul_hw) + m) >> p.sh_hw;
      const uint32_t rem = m - img * (uint32_t)p.HW;
      const uint32_t oy = (__umulhi(rem, p.mul_w) + rem) >> p.sh_w;
      const uint32_t ox = rem - oy * (uint32_t)p.W;
      const uint32_t vx = (ox >= 1u ? 1u : 0u) | 2u | (ox + 1u < (uint32_t)p.W ? 4u : 0u);
      mask = (oy >= 1u ? vx : 0u) | (vx << 3) | (oy + 1u < (uint32_t)p.H ? vx << 6 : 0u);
    }
    tapmask[mi] = mask;
  }
  // wave-uniform: does any of the 64 lanes need masking for this 16-pixel group at all?
  bool need_mask[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) need_mask[mi] = __any(tapmask[mi] != 0x1FFu);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int a_lane_row = wm * 64 + fr;
  const int b_lane_row = wn * 64 + fr;
  const int b_off0 = b_lane_row * 128 + (((0 + fq) ^ (b_lane_row & 7)) << 4);
  const int b_off1 = b_lane_row * 128 + (((4 + fq) ^ (b_lane_row & 7)) << 4);

  Y3_COARSE(0);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  Y3_COARSE(1);
  if (grp) __builtin_amdgcn_s_barrier();              // group 1 runs one segment behind group 0

  // write-out role of this thread (epilogue): 8 channels [co, co+8) of pixels (tid>>4) + 32*j
  constexpr int WR = BM * (BN / 8) / NT;
  const int oc_mine = tid & 15;
  const int co = n0 + oc_mine * 8;
  const bool has_res = p.flags & Y3_F_RESIDUAL;
  u32x4 resv[WR];
  f32x4 sc_lo, sc_hi, bi_lo, bi_hi;

  u32x4 xf[2][MI], wf[2][NI];
  int tap = 0, chunk = 0;        // step s
  int tap2 = 2, chunk2 = 0;      // step s + 2 (whose weights are issued in load segment s)
#pragma unroll 1
  for (int s = 0; s < nit; ++s) {
    // ================= load segment =================
    const char *aBuf = sA + (chunk & 1) * p.a_bytes;
    const char *bBuf = sB + (s % NSB) * B_BYTES;
    const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
    const int r0 = a_lane_row + ky * p.W + kx;
    const char *ap0 = aBuf + r0 * 128 + (((0 + fq) ^ (r0 & 7)) << 4);
    const char *ap1 = aBuf + r0 * 128 + (((4 + fq) ^ (r0 & 7)) << 4);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) xf[0][mi] = *reinterpret_cast<const u32x4 *>(ap0 + mi * 2048);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) wf[0][ni] = *reinterpret_cast<const u32x4 *>(bBuf + b_off0 + ni * 2048);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) xf[1][mi] = *reinterpret_cast<const u32x4 *>(ap1 + mi * 2048);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) wf[1][ni] = *reinterpret_cast<const u32x4 *>(bBuf + b_off1 + ni * 2048);
    if (s == nit - 1) {
      // last step: no LDS-DMA left to issue; start the epilogue's global reads now so that their latency
      // hides under this step's MFMAs (every counted vmcnt wait below is skipped for this step)
      sc_lo = *reinterpret_cast<const f32x4 *>(p.scale + co);
      sc_hi = *reinterpret_cast<const f32x4 *>(p.scale + co + 4);
      bi_lo = *reinterpret_cast<const f32x4 *>(p.bias + co);
      bi_hi = *reinterpret_cast<const f32x4 *>(p.bias + co + 4);
      if (has_res) {
#pragma unroll
        for (int j = 0; j < WR; ++j) {
          const int m = m0 + (tid >> 4) + j * (NT / 16);
          const char *rp = p.res + ((long long)m * p.res_ld + co) * ES;
          resv[j] = m < p.M ? *reinterpret_cast<const u32x4 *>(rp) : u32x4{0u, 0u, 0u, 0u};
        }
      }
    }
    Y3_FINE(0);   // [barrier wait that started this load segment ... fragment reads issued]
    // the fragment reads' latency hides under the LDS-DMA issue below
    int issued = 0;
    if (chunk + 1 < p.nchunks && tap < p.na) { issue_halo_pass(chunk + 1, tap); issued += 1; }
    if (s + 2 < nit) { issue_weights(chunk2, tap2, (s + 2) % NSB); issued += NB; }
    Y3_FINE(1);   // LDS-DMA issue
    if (grp == 1 && s != nit - 1) {                   // end of an odd half-step for group 1
      if (issued == NB + 1) wait_vmcnt<NB + 1>();
      else if (issued == NB) wait_vmcnt<NB>();
      else if (issued == 1) wait_vmcnt<1>();
      else wait_vmcnt<0>();
    }
    Y3_FINE(2);   // vmcnt wait (group 1 only)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    Y3_FINE(3);   // fragment reads landed
    __builtin_amdgcn_s_barrier();
    Y3_FINE(4);   // barrier (waiting for the partner group's compute segment)
    // ================= compute segment =================
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      if (need_mask[mi]) {   // scalar branch: interior pixel groups skip the 8 v_cndmask
        const bool dead = !((tapmask[mi] >> tap) & 1u);
#pragma unroll
        for (int g = 0; g < 2; ++g)
          if (dead) xf[g][mi] = u32x4{0u, 0u, 0u, 0u};
      }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) MmaH<T>::run(acc[mi][ni], wf[g][ni], xf[g][mi]);
    __builtin_amdgcn_s_setprio(0);
    Y3_FINE(5);   // masks + 32 MFMAs issued
    if (grp == 0 && s != nit - 1) {                   // end of an odd half-step for group 0
      if (issued == NB + 1) wait_vmcnt<NB + 1>();
      else if (issued == NB) wait_vmcnt<NB>();
      else if (issued == 1) wait_vmcnt<1>();
      else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    Y3_FINE(6);   // vmcnt wait (group 0) + barrier (waiting for the partner group's load segment)
    tap = tap == 8 ? 0 : tap + 1;
    chunk += tap == 0 ? 1 : 0;
    tap2 = tap2 == 8 ? 0 : tap2 + 1;
    chunk2 += tap2 == 0 ? 1 : 0;
  }
  if (!grp) __builtin_amdgcn_s_barrier();             // balance group 1's extra barrier
  __syncthreads();
  Y3_COARSE(2);

  // ---- epilogue (as above): raw fp32 tile -> LDS -> 8 channels of one pixel per thread-step ----------
  constexpr int SWZ = 15;
  constexpr int OCT_PER_ROW = BN / 8;
  float *sC = reinterpret_cast<float *>(smem);
  const bool leaky = p.flags & Y3_F_LEAKY;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int cl = wn * 64 + ni * 16 + fq * 4;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int pl = wm * 64 + mi * 16 + fr;
      *reinterpret_cast<f32x4 *>(sC + pl * BN + (((cl >> 2) ^ (pl & SWZ)) << 2)) = acc[mi][ni];
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < WR; ++j) {
    const int pl = (tid >> 4) + j * (NT / 16);
    const int m = m0 + pl;
    if (m >= p.M) continue;
    const f32x4 lo = *reinterpret_cast<const f32x4 *>(sC + pl * BN + (((2 * oc_mine) ^ (pl & SWZ)) << 2));
    const f32x4 hi = *reinterpret_cast<const f32x4 *>(sC + pl * BN + (((2 * oc_mine + 1) ^ (pl & SWZ)) << 2));
    float v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = lo[r] * sc_lo[r] + bi_lo[r];
      v[4 + r] = hi[r] * sc_hi[r] + bi_hi[r];
    }
    if (leaky) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : Y3_LEAKY_SLOPE * v[r];
    }
    if (has_res) {
      if constexpr (sizeof(T) == 2) {
        const bf16x8 rv = __builtin_bit_cast(bf16x8, resv[j]);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += (float)rv[r];
      } else {
        const float *rp = reinterpret_cast<const float *>(p.res) + (long long)m * p.res_ld + co;
        const f32x4 r0v = __builtin_bit_cast(f32x4, resv[j]), r1v = *reinterpret_cast<const f32x4 *>(rp + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] += r0v[r]; v[4 + r] += r1v[r]; }
      }
    }
    T *op = reinterpret_cast<T *>(p.out) + (long long)m * p.out_ld + co;
    if constexpr (sizeof(T) == 2) {
      bf16x8 ov;
#pragma unroll
      for (int r = 0; r < 8; ++r) ov[r] = (bf16_t)v[r];
      *reinterpret_cast<bf16x8 *>(op) = ov;
    } else {
      *reinterpret_cast<f32x4 *>(op) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4 *>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
  }
  Y3_COARSE(3);
  Y3_STAMP_COUNT();
}

// ------------------------------------------------------------------------------------------------
// 32-channel-chunk variant ("halo32"): 64-byte LDS rows.  The halo of a chunk is then 4x smaller, which buys
// what the 128-byte version cannot afford at W = 76: a 5-slot weight ring (loads issued 3 K-steps ahead, so
// the counted vmcnt wait at the top of a step is normally free) next to the double-buffered halo, in 112 KiB.
// One K-step = one filter tap x 32 channels = ONE MFMA k-step: 16 MFMAs per wave, whose gaps carry the 8
// fragment reads of the NEXT step (double-buffered registers, loop unrolled by two) and the step's two LDS-DMA
// instructions (branch-free issue; a dummy goes to a dump region when there is nothing to fetch).
template <typename T>
__global__ __launch_bounds__(512) void conv_halo32_kernel(HaloArgs p) {
  constexpr int BM = 256, BN = 128, NT = 512, NSB = 5, D = NSB - 2;
  constexpr int ES = sizeof(T);
  constexpr int RB = 64;                              // bytes per LDS row
  constexpr int BKE = RB / ES;                        // channels per chunk (32 bf16 / 16 fp32)
  constexpr int RPP = NT / 4;                         // rows per loader pass (4 chunks of 16 B per row)
  constexpr int B_BYTES = BN * RB;                    // 8 KiB per weight slot: exactly one DMA per thread
  constexpr int MI = 4, NI = 4;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *sB = smem;                                    // [NSB][128][64]
  char *sDump = smem + NSB * B_BYTES;                 // [8 waves][1 KiB] target of dummy DMAs
  char *sA = sDump + 8 * 1024;                        // [2][hr_pad][64]

  Y3_STAMP_DECL
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;

  const int tile = y3_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / p.n_tiles) * BM;
  const int n0 = (tile % p.n_tiles) * BN;

  const int slot = tid & 3;
  const int row0 = tid >> 2;                          // 0..127
  const int kc = slot ^ ((row0 >> 1) & 3);            // RPP % 8 == 0: later passes keep (row >> 1) & 3
  const long long q0 = (long long)m0 - p.W - 1;

  auto issue_halo_pass = [&](int chunk, int pass, bool live) {
    const long long q = q0 + row0 + pass * RPP;
    const bool ok = live && q >= 0 && q < p.M;
    const char *src = ok ? p.in + (q * p.in_ld + (long long)chunk * BKE) * ES + kc * 16 : p.zero;
    char *dst = live ? sA + (chunk & 1) * p.a_bytes + pass * (NT * 16) + wave * 1024 : sDump + wave * 1024;
    __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)dst, 16, 0, 0);
  };
  const char *b_src = p.wgt + ((long long)(n0 + row0) * p.k_ld) * ES + kc * 16;
  auto issue_weights = [&](int it, int slot_it) {     // it = chunk*9 + tap
    const int chunk = it / 9, tap = it - chunk * 9;
    const long long koff = ((long long)tap * p.Cin + (long long)chunk * BKE) * ES;
    __builtin_amdgcn_global_load_lds((gbl_void *)(b_src + koff), (lds_void *)(sB + (slot_it % NSB) * B_BYTES + wave * 1024), 16, 0, 0);
  };

  const int nit = p.nchunks * 9;                      // even (the launcher requires an even chunk count)
  for (int pass = 0; pass < p.na; ++pass) issue_halo_pass(0, pass, true);
#pragma unroll
  for (int j = 0; j <= D; ++j) issue_weights(j, j);

  uint32_t tapmask[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const uint32_t m = (uint32_t)(m0 + wm * 64 + mi * 16 + fr);
    uint32_t mask = 0u;
    if (m < (uint32_t)p.M) {
      const uint32_t img = (__umulhi(m, p.mul_hw) + m) >> p.sh_hw;
      const uint32_t rem = m - img * (uint32_t)p.HW;
      const uint32_t oy = (__umulhi(rem, p.mul_w) + rem) >> p.sh_w;
      const uint32_t ox = rem - oy * (uint32_t)p.W;
      const uint32_t vx = (ox >= 1u ? 1u : 0u) | 2u | (ox + 1u < (uint32_t)p.W ? 4u : 0u);
      mask = (oy >= 1u ? vx : 0u) | (vx << 3) | (oy + 1u < (uint32_t)p.H ? vx << 6 : 0u);
    }
    tapmask[mi] = mask;
  }
  bool need_mask[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) need_mask[mi] = __any(tapmask[mi] != 0x1FFu);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int a_lane_row = wm * 64 + fr;
  const int b_lane_row = wn * 64 + fr;
  const int b_off = b_lane_row * RB + ((fq ^ ((b_lane_row >> 1) & 3)) << 4);

  // fragment pointers of step `it` (tap, chunk): A rows shift with the tap, B comes from the ring slot
  auto a_ptr = [&](int chunk, int tap) -> const char * {
    const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
    const int r0 = a_lane_row + ky * p.W + kx;
    return sA + (chunk & 1) * p.a_bytes + r0 * RB + ((fq ^ ((r0 >> 1) & 3)) << 4);
  };

  Y3_COARSE(0);
  // halo(0) and weights(0..D-1) landed; only weights(D) may still fly.  (Waiting for just weights(1) would
  // break the steady-state count at step 1, whose weights(2) would then have D younger loads, not 2(D-1).)
  wait_vmcnt<1>();
  __builtin_amdgcn_s_barrier();
  Y3_COARSE(1);

  u32x4 xf0[MI], wf0[NI], xf1[MI], wf1[NI];
  {
    const char *ap = a_ptr(0, 0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) xf0[mi] = *reinterpret_cast<const u32x4 *>(ap + mi * 16 * RB);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) wf0[ni] = *reinterpret_cast<const u32x4 *>(sB + b_off + ni * 16 * RB);
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);

  // one K-step: MFMAs on (xc, wc) while (xn, wn_) receive the next step's fragments and the step's DMA goes out
  auto step = [&](int it, int tap, int chunk, u32x4 (&xc)[MI], u32x4 (&wc)[NI], u32x4 (&xn)[MI], u32x4 (&wn_)[NI]) {
    wait_vmcnt<2 * (D - 1)>();                        // weights(it+1) (+ a due halo) landed; D-1 steps' loads may fly
    __builtin_amdgcn_s_barrier();
    const int tap_n = tap == 8 ? 0 : tap + 1;
    const int chunk_n = tap == 8 ? chunk + 1 : chunk;
    const char *ap = a_ptr(chunk_n, tap_n);
    const char *bp = sB + ((it + 1) % NSB) * B_BYTES + b_off;
    const bool live = chunk + 1 < p.nchunks && tap < p.na;
    const int itw = it + 1 + D < nit ? it + 1 + D : nit - 1;
    __builtin_amdgcn_sched_barrier(0);
    xn[0] = *reinterpret_cast<const u32x4 *>(ap);
    xn[1] = *reinterpret_cast<const u32x4 *>(ap + 16 * RB);
    issue_halo_pass(chunk + 1, tap < p.na ? tap : 0, live);
    xn[2] = *reinterpret_cast<const u32x4 *>(ap + 32 * RB);
    xn[3] = *reinterpret_cast<const u32x4 *>(ap + 48 * RB);
    wn_[0] = *reinterpret_cast<const u32x4 *>(bp);
    wn_[1] = *reinterpret_cast<const u32x4 *>(bp + 16 * RB);
    issue_weights(itw, it + 1 + D);
    wn_[2] = *reinterpret_cast<const u32x4 *>(bp + 32 * RB);
    wn_[3] = *reinterpret_cast<const u32x4 *>(bp + 48 * RB);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      if (need_mask[mi] && !((tapmask[mi] >> tap) & 1u)) xc[mi] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) MmaH<T>::run(acc[mi][ni], wc[ni], xc[mi]);
    // 2 DS reads, 4 MFMAs, 1 DMA | 2 DS, 4 MFMA | 2 DS, 4 MFMA, 1 DMA | 2 DS, 4 MFMA   (bf16 counts)
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, sizeof(T) == 2 ? 4 : 16, 0);
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, sizeof(T) == 2 ? 4 : 16, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, sizeof(T) == 2 ? 4 : 16, 0);
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, sizeof(T) == 2 ? 4 : 16, 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);               // next fragments landed (16 MFMAs of cover)
  };

  int tap = 0, chunk = 0;
#pragma unroll 1
  for (int it = 0; it < nit; it += 2) {
    step(it, tap, chunk, xf0, wf0, xf1, wf1);
    const int tap1 = tap == 8 ? 0 : tap + 1;
    const int chunk1 = tap == 8 ? chunk + 1 : chunk;
    step(it + 1, tap1, chunk1, xf1, wf1, xf0, wf0);
    tap = tap1 == 8 ? 0 : tap1 + 1;
    chunk = tap1 == 8 ? chunk1 + 1 : chunk1;
  }
  wait_vmcnt<0>();                                    // dummy DMAs of the last steps
  __syncthreads();
  Y3_COARSE(2);

  // ---- epilogue (as in the ping-pong kernel) ---------------------------------------------------------------
  constexpr int SWZ = 15;
  constexpr int OCT_PER_ROW = BN / 8;
  constexpr int WR = BM * OCT_PER_ROW / NT;
  float *sC = reinterpret_cast<float *>(smem);
  const bool leaky = p.flags & Y3_F_LEAKY;
  const bool has_res = p.flags & Y3_F_RESIDUAL;
  const int oc_mine = tid & 15;
  const int co = n0 + oc_mine * 8;
  const f32x4 sc_lo = *reinterpret_cast<const f32x4 *>(p.scale + co);
  const f32x4 sc_hi = *reinterpret_cast<const f32x4 *>(p.scale + co + 4);
  const f32x4 bi_lo = *reinterpret_cast<const f32x4 *>(p.bias + co);
  const f32x4 bi_hi = *reinterpret_cast<const f32x4 *>(p.bias + co + 4);
  u32x4 resv[WR];
  if (has_res) {
#pragma unroll
    for (int j = 0; j < WR; ++j) {
      const int m = m0 + (tid >> 4) + j * (NT / 16);
      const char *rp = p.res + ((long long)m * p.res_ld + co) * ES;
      resv[j] = m < p.M ? *reinterpret_cast<const u32x4 *>(rp) : u32x4{0u, 0u, 0u, 0u};
    }
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int cl = wn * 64 + ni * 16 + fq * 4;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int pl = wm * 64 + mi * 16 + fr;
      *reinterpret_cast<f32x4 *>(sC + pl * BN + (((cl >> 2) ^ (pl & SWZ)) << 2)) = acc[mi][ni];
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < WR; ++j) {
    const int pl = (tid >> 4) + j * (NT / 16);
    const int m = m0 + pl;
    if (m >= p.M) continue;
    const f32x4 lo = *reinterpret_cast<const f32x4 *>(sC + pl * BN + (((2 * oc_mine) ^ (pl & SWZ)) << 2));
    const f32x4 hi = *reinterpret_cast<const f32x4 *>(sC + pl * BN + (((2 * oc_mine + 1) ^ (pl & SWZ)) << 2));
    float v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = lo[r] * sc_lo[r] + bi_lo[r];
      v[4 + r] = hi[r] * sc_hi[r] + bi_hi[r];
    }
    if (leaky) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : Y3_LEAKY_SLOPE * v[r];
    }
    if (has_res) {
      if constexpr (sizeof(T) == 2) {
        const bf16x8 rv = __builtin_bit_cast(bf16x8, resv[j]);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += (float)rv[r];
      } else {
        const float *rp = reinterpret_cast<const float *>(p.res) + (long long)m * p.res_ld + co;
        const f32x4 r0v = __builtin_bit_cast(f32x4, resv[j]), r1v = *reinterpret_cast<const f32x4 *>(rp + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] += r0v[r]; v[4 + r] += r1v[r]; }
      }
    }
    T *op = reinterpret_cast<T *>(p.out) + (long long)m * p.out_ld + co;
    if constexpr (sizeof(T) == 2) {
      bf16x8 ov;
#pragma unroll
      for (int r = 0; r < 8; ++r) ov[r] = (bf16_t)v[r];
      *reinterpret_cast<bf16x8 *>(op) = ov;
    } else {
      *reinterpret_cast<f32x4 *>(op) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4 *>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
  }
  Y3_COARSE(3);
  Y3_STAMP_COUNT();
}

// n / d == (umulhi(n, mul) + n) >> sh for 0 <= n < 2^31 (round-up method, d >= 1)
void fast_div(uint32_t d, uint32_t &mul, uint32_t &sh) {
  if (d <= 1) { mul = 0; sh = 0; return; }
  sh = 0;
  while ((1u << sh) < d) ++sh;
  mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << sh) - d)) / d + 1);
}


// ------------------------------------------------------------------------------------------------
// Wave-specialised variant (BM = 256, BN = 128): 8 consumer waves (wave tile 64 x 64, two per SIMD) that only
// read fragments and issue MFMAs + 4 loader waves (one per SIMD) that only issue LDS-DMA.  Measured on the
// kernels above and on conv_igemm3: a wave that issues a 1 KiB LDS-DMA piece is stuck for 60-185 cycles and
// feeds no MFMA meanwhile, and the per-CU LDS-DMA path takes ~22 cycles per KiB however many waves issue.  The
// halo image keeps the bytes per FLOP low (weights: 16 KiB per K-step, halo: ~52 KiB per nine K-steps, against
// 1024 MFMA cycles per SIMD and K-step); the role split keeps the DMA issue out of the MFMA waves' streams.
// One raw workgroup barrier per K-step:  loaders: wait(loads older than D-1 steps) ; barrier ; issue
// weights(it+D+1) + two halo slices of the next chunk.   consumers: barrier ; MFMAs(it) with the fragments of the
// second K-half and of step it+1's first K-half read in the MFMA gaps.  Ring: NSB = D + 2 weight slots.
template <typename T, int NSB>
__global__ __launch_bounds__(768, 3) void conv_halo_ws_kernel(HaloArgs p) {
  constexpr int BM = 256, BN = 128;
  constexpr int WAVES_N = 2;
  constexpr int NC = 512, NL = 256;                   // consumer / loader threads
  constexpr int ES = sizeof(T);
  constexpr int BKE = 128 / ES;
  constexpr int RPL = NL / 8;                         // rows filled per loader pass (32)
  constexpr int NBL = BN / RPL;                       // weight-tile passes per loader thread (4)
  constexpr int HPS = 2;                              // halo passes per K-step
  constexpr int PER = NBL + HPS;                      // LDS-DMA instructions per loader thread and K-step
  constexpr int B_BYTES = BN * 128;
  constexpr int D = NSB - 2;
  constexpr int MI = 4, NI = 4;
  static_assert(NSB == 3 || NSB == 4, "weight ring has 3 or 4 slots");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *sB = smem;                                    // [NSB][BN][128]
  char *sA = smem + NSB * B_BYTES;                    // [2][hr_pad][128]

  Y3_STAMP_DECL
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= NC / 64;

  const int tile = y3_xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / p.n_tiles) * BM;
  const int n0 = (tile % p.n_tiles) * BN;
  const int nit = p.nchunks * 9;

  const int wm = (wave & 7) / WAVES_N, wn = (wave & 7) % WAVES_N;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (loader) {
    // ---------------- loader waves ----------------
    // youngest waves of the workgroup: without a raised priority their few VALU / VMEM instructions lose every
    // issue arbitration against the two MFMA waves of the SIMD (stamps: ~190 cycles per LDS-DMA instruction)
    __builtin_amdgcn_s_setprio(3);
    const int ltid = tid - NC;
    const int lwave = wave - NC / 64;
    const int slot = ltid & 7;
    const int row0 = ltid >> 3;
    const int kc = slot ^ (row0 & 7);
    const long long q0 = (long long)m0 - p.W - 1;     // flattened input pixel of halo row 0
    auto issue_halo_pass = [&](int chunk, int pass, bool live) {
      const long long q = q0 + row0 + pass * RPL;
      const bool ok = live && q >= 0 && q < p.M;
      const char *src = ok ? p.in + (q * p.in_ld + (long long)chunk * BKE) * ES + kc * 16 : p.zero;
      char *dst = sA + (chunk & 1) * p.a_bytes + pass * (NL * 16) + lwave * 1024;
      __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)dst, 16, 0, 0);
    };
    const char *b_src[NBL];
#pragma unroll
    for (int i = 0; i < NBL; ++i) b_src[i] = p.wgt + ((long long)(n0 + row0 + i * RPL) * p.k_ld) * ES + kc * 16;
    auto issue_weights = [&](int it, int ring_slot) {  // it = chunk*9 + tap; K offset = tap*Cin + chunk*BKE elements
      const int chunk = it / 9, tap = it - chunk * 9;
      const long long koff = ((long long)tap * p.Cin + (long long)chunk * BKE) * ES;
      char *dst = sB + ring_slot * B_BYTES + lwave * 1024;
#pragma unroll
      for (int i = 0; i < NBL; ++i)
        __builtin_amdgcn_global_load_lds((gbl_void *)(b_src[i] + koff), (lds_void *)(dst + i * (NL * 16)), 16, 0, 0);
    };
    for (int pass = 0; pass < p.na; ++pass) issue_halo_pass(0, pass, true);
#pragma unroll
    for (int j = 0; j <= D; ++j) issue_weights(j < nit ? j : nit - 1, j);
    Y3_COARSE(6);
    int tap = 0, chunk = 0, ring = (D + 1) % NSB;
#pragma unroll 1
    for (int it = 0; it < nit; ++it) {
      // before barrier B(it): weights(it+1) (and everything older) landed; with the 4-slot ring the loads of the
      // previous step (or, at it == 0, the prologue's last weight tile) may still fly
      // Per step the weight tile goes out FIRST and the two halo slices after it, so the counted wait can leave the
      // youngest halo slices in flight (they are not needed before the next chunk, and a slice issued at step s is
      // covered by the wait of step s+1): only the weights' landing sits on the barrier's critical path.
      // (guaranteed landed at B(it): the weights issued D steps ago; halo slices issued D + 1 or more steps ago)
      if (D == 2) { if (it == 0) wait_vmcnt<NBL>(); else if (it == 1) wait_vmcnt_n<PER>(); else wait_vmcnt_n<PER + HPS>(); }
      else { if (it == 0) wait_vmcnt<0>(); else wait_vmcnt_n<HPS>(); }
      Y3_COARSE(3);
      __builtin_amdgcn_s_barrier();
      Y3_COARSE(4);
      const bool live = chunk + 1 < p.nchunks;
      const int p0 = 2 * tap < p.na ? 2 * tap : p.na - 1;
      const int p1 = 2 * tap + 1 < p.na ? 2 * tap + 1 : p.na - 1;
      const int itw = it + 1 + D < nit ? it + 1 + D : nit - 1;
      issue_weights(itw, ring);                       // ring == (it + 1 + D) % NSB: the slot of weights(it-1), free
      issue_halo_pass(chunk + 1, p0, live);
      issue_halo_pass(chunk + 1, p1, live);
      ring = ring + 1 == NSB ? 0 : ring + 1;
      if (++tap == 9) { tap = 0; ++chunk; }
      Y3_COARSE(5);
    }
    wait_vmcnt<0>();                                  // the tail's dummy loads must not land on the output tile
#if defined(Y3_STAMPS) && !defined(Y3_STAMPS_FINE)
    if (tid == NC) for (int _i = 3; _i < 7; ++_i) atomicAdd(&g_y3_stamps[_i], _st_acc[_i]);
#endif
  } else {
    // ---------------- consumer waves ----------------
    uint32_t tapmask[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const uint32_t m = (uint32_t)(m0 + wm * 64 + mi * 16 + fr);
      uint32_t mask = 0u;
      if (m < (uint32_t)p.M) {
        const uint32_t img = (__umulhi(m, p.mul_hw) + m) >> p.sh_hw;
        const uint32_t rem = m - img * (uint32_t)p.HW;
        const uint32_t oy = (__umulhi(rem, p.mul_w) + rem) >> p.sh_w;
        const uint32_t ox = rem - oy * (uint32_t)p.W;
        const uint32_t vx = (ox >= 1u ? 1u : 0u) | 2u | (ox + 1u < (uint32_t)p.W ? 4u : 0u);
        mask = (oy >= 1u ? vx : 0u) | (vx << 3) | (oy + 1u < (uint32_t)p.H ? vx << 6 : 0u);
      }
      tapmask[mi] = mask;
    }
    const int a_lane_row = wm * 64 + fr;
    const int b_lane_row = wn * 64 + fr;
    const int b_off0 = b_lane_row * 128 + (((0 + fq) ^ (b_lane_row & 7)) << 4);
    const int b_off1 = b_lane_row * 128 + (((4 + fq) ^ (b_lane_row & 7)) << 4);
    auto read_frags = [&](u32x4 (&xf)[MI], u32x4 (&wf)[NI], const char *aBuf, const char *bBuf, int a_shift, int g) {
      const int r0 = a_lane_row + a_shift;
      const char *ap = aBuf + r0 * 128 + (((g * 4 + fq) ^ (r0 & 7)) << 4);
      const char *bp = bBuf + (g ? b_off1 : b_off0);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) xf[mi] = *reinterpret_cast<const u32x4 *>(ap + mi * 2048);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[ni] = *reinterpret_cast<const u32x4 *>(bp + ni * 2048);
    };
    auto mma_all = [&](u32x4 (&xf)[MI], const u32x4 (&wf)[NI], int tap) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        if (!((tapmask[mi] >> tap) & 1u)) xf[mi] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) MmaH<T>::run(acc[mi][ni], wf[ni], xf[mi]);
    };
    auto interleave = [&]() {
#pragma unroll
      for (int i = 0; i < MI + NI; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, sizeof(T) == 2 ? 2 : 8, 0);
      }
    };
    Y3_STAMP(2);
    __builtin_amdgcn_s_barrier();                     // B(0): halo(0), weights(0), weights(1) are in LDS
    Y3_STAMP(0);
    u32x4 xf0[MI], wf0[NI], xf1[MI], wf1[NI];
    read_frags(xf0, wf0, sA, sB, 0, 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    int tap = 0, chunk = 0, ring = 0;
#pragma unroll 1
    for (int it = 0; it < nit; ++it) {
      if (it) {
        __builtin_amdgcn_s_barrier();                 // B(it): weights(it+1) landed; slot of weights(it-1) released
        Y3_STAMP(0);
      }
#if defined(Y3_STAMPS_FINE)
      const bool w0 = wave == 0;
#define Y3_W0(slot) do { if (w0) Y3_STAMP(slot); } while (0)
#else
#define Y3_W0(slot) do {} while (0)
#endif
      const char *aBuf = sA + (chunk & 1) * p.a_bytes;
      const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
      __builtin_amdgcn_sched_barrier(0);
      read_frags(xf1, wf1, aBuf, sB + ring * B_BYTES, ky * p.W + kx, 1);
      mma_all(xf0, wf0, tap);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      Y3_W0(2);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      Y3_W0(3);
      const int tap_n = tap == 8 ? 0 : tap + 1;
      const int chunk_n = tap == 8 ? chunk + 1 : chunk;
      const int ring_n = ring + 1 == NSB ? 0 : ring + 1;
      {
        const int ky_n = (tap_n * 11) >> 5, kx_n = tap_n - ky_n * 3;
        read_frags(xf0, wf0, sA + (chunk_n & 1) * p.a_bytes, sB + ring_n * B_BYTES, ky_n * p.W + kx_n, 0);
      }
      mma_all(xf1, wf1, tap);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      Y3_W0(4);
      __builtin_amdgcn_s_waitcnt(0xC07F);
#if defined(Y3_STAMPS_FINE)
      if (w0) Y3_STAMP(5); else Y3_STAMP(1);
#else
      Y3_STAMP(1);
#endif
      tap = tap_n;
      chunk = chunk_n;
      ring = ring_n;
    }
#if defined(Y3_STAMPS_FINE)
    if (tid == 0) {
      for (int _i = 0; _i < 6; ++_i) if (_i != 1) atomicAdd(&g_y3_stamps[_i], _st_acc[_i]);
      atomicAdd(&g_y3_stamps[7], 1ull);
    }
    if (tid == 256) {   // wave 4: slot 1 = its compute time, slot 6 = its barrier wait
      atomicAdd(&g_y3_stamps[1], _st_acc[1]);
      atomicAdd(&g_y3_stamps[6], _st_acc[0]);
    }
#elif defined(Y3_STAMPS)
    if (tid == 0) {
      for (int _i = 0; _i < 3; ++_i) atomicAdd(&g_y3_stamps[_i], _st_acc[_i]);
      atomicAdd(&g_y3_stamps[7], 1ull);
    }
#endif
  }
  __syncthreads();  // all operand reads and all LDS-DMA done: LDS can hold the output tile

  // ---- epilogue (the 512 consumer threads write out; the loaders only keep the barrier count) ----
  constexpr int SWZ = 15;
  constexpr int OCT_PER_ROW = BN / 8;
  constexpr int WR = BM * OCT_PER_ROW / NC;
  float *sC = reinterpret_cast<float *>(smem);
  const bool leaky = p.flags & Y3_F_LEAKY;
  const bool has_res = p.flags & Y3_F_RESIDUAL;
  const int oc_mine = tid & 15;
  const int co = n0 + oc_mine * 8;
  f32x4 sc_lo, sc_hi, bi_lo, bi_hi;
  u32x4 resv[WR];
  if (!loader) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int cl = wn * 64 + ni * 16 + fq * 4;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int pl = wm * 64 + mi * 16 + fr;
        *reinterpret_cast<f32x4 *>(sC + pl * BN + (((cl >> 2) ^ (pl & SWZ)) << 2)) = acc[mi][ni];
      }
    }
    sc_lo = *reinterpret_cast<const f32x4 *>(p.scale + co);
    sc_hi = *reinterpret_cast<const f32x4 *>(p.scale + co + 4);
    bi_lo = *reinterpret_cast<const f32x4 *>(p.bias + co);
    bi_hi = *reinterpret_cast<const f32x4 *>(p.bias + co + 4);
    if (has_res) {
#pragma unroll
      for (int j = 0; j < WR; ++j) {
        const int m = m0 + (tid >> 4) + j * (NC / 16);
        const char *rp = p.res + ((long long)m * p.res_ld + co) * ES;
        if constexpr (sizeof(T) == 2) {
          resv[j] = m < p.M ? *reinterpret_cast<const u32x4 *>(rp) : u32x4{0u, 0u, 0u, 0u};
        }
      }
    }
  }
  __syncthreads();
  if (loader) return;
#pragma unroll
  for (int j = 0; j < WR; ++j) {
    const int pl = (tid >> 4) + j * (NC / 16);
    const int m = m0 + pl;
    if (m >= p.M) continue;
    const f32x4 lo = *reinterpret_cast<const f32x4 *>(sC + pl * BN + (((2 * oc_mine) ^ (pl & SWZ)) << 2));
    const f32x4 hi = *reinterpret_cast<const f32x4 *>(sC + pl * BN + (((2 * oc_mine + 1) ^ (pl & SWZ)) << 2));
    float v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = lo[r] * sc_lo[r] + bi_lo[r];
      v[4 + r] = hi[r] * sc_hi[r] + bi_hi[r];
    }
    if (leaky) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : Y3_LEAKY_SLOPE * v[r];
    }
    if (has_res) {
      if constexpr (sizeof(T) == 2) {
        const bf16x8 rv = __builtin_bit_cast(bf16x8, resv[j]);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += (float)rv[r];
      } else {
        const float *rp = reinterpret_cast<const float *>(p.res) + (long long)m * p.res_ld + co;
        const f32x4 r0 = *reinterpret_cast<const f32x4 *>(rp), r1 = *reinterpret_cast<const f32x4 *>(rp + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] += r0[r]; v[4 + r] += r1[r]; }
      }
    }
    T *op = reinterpret_cast<T *>(p.out) + (long long)m * p.out_ld + co;
    if constexpr (sizeof(T) == 2) {
      bf16x8 ov;
#pragma unroll
      for (int r = 0; r < 8; ++r) ov[r] = (bf16_t)v[r];
      *reinterpret_cast<bf16x8 *>(op) = ov;
    } else {
      *reinterpret_cast<f32x4 *>(op) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4 *>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Persistent form of the wave-specialised kernel: one workgroup per CU walks a list of tiles
// (tile j of workgroup b = xcd_remap(b) + j * gridDim), and the operand streams never stop at a tile boundary:
// the halo image of the next tile's first chunk is simply "the next chunk" of the double-buffered halo, the weight
// ring runs D + 1 K-steps ahead straight into the next tile, so a tile's first MFMA does not wait for an 84 KiB
// prologue (stamps on the one-tile kernel: ~5 k of ~41 k cycles per workgroup) nor for a workgroup dispatch.
// The epilogue no longer parks the whole 256 x 128 fp32 tile over the operand buffers (they now hold the next tile's
// prefetch): after one extra barrier per tile ("everyone is done reading the last chunk's halo") each consumer wave
// parks 16 pixels x 64 channels at a time in a private 4 KiB slice of that -- now idle -- halo buffer and writes
// them out as 16-byte NHWC chunks; no further workgroup barrier, and the loaders keep streaming meanwhile.
template <typename T, int NSB>
__global__ __launch_bounds__(768, 3) void conv_halo_wsp_kernel(HaloArgs p, int n_tiles_total) {
  constexpr int BM = 256, BN = 128;
  constexpr int WAVES_N = 2;
  constexpr int NC = 512, NL = 256;
  constexpr int ES = sizeof(T);
  constexpr int BKE = 128 / ES;
  constexpr int RPL = NL / 8;
  constexpr int NBL = BN / RPL;
  constexpr int HPS = 2;
  constexpr int PER = NBL + HPS;
  constexpr int B_BYTES = BN * 128;
  constexpr int D = NSB - 2;
  constexpr int MI = 4, NI = 4;
  static_assert(NSB == 3 || NSB == 4, "weight ring has 3 or 4 slots");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *sB = smem;                                    // [NSB][BN][128]
  char *sA = smem + NSB * B_BYTES;                    // [2][hr_pad][128]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= NC / 64;
  const int nit = p.nchunks * 9;
  const int grid = gridDim.x;
  const int tile0 = y3_xcd_remap(blockIdx.x, grid);   // tiles of this workgroup: tile0, tile0 + grid, ...

  if (loader) {
    __builtin_amdgcn_s_setprio(3);
    const int ltid = tid - NC;
    const int lwave = wave - NC / 64;
    const int slot = ltid & 7;
    const int row0 = ltid >> 3;
    const int kc = slot ^ (row0 & 7);
    // halo slice `pass` of chunk `chunk` of the tile whose first halo pixel is q0, into halo buffer `buf`
    auto issue_halo_pass = [&](long long q0, int chunk, int pass, int buf, bool live) {
      const long long q = q0 + row0 + pass * RPL;
      const bool ok = live && q >= 0 && q < p.M;
      const char *src = ok ? p.in + (q * p.in_ld + (long long)chunk * BKE) * ES + kc * 16 : p.zero;
      char *dst = sA + buf * p.a_bytes + pass * (NL * 16) + lwave * 1024;
      __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)dst, 16, 0, 0);
    };
    auto issue_weights = [&](int n0, int it, int ring_slot) {
      const int chunk = it / 9, tap = it - chunk * 9;
      const long long koff = ((long long)tap * p.Cin + (long long)chunk * BKE) * ES;
      const char *src0 = p.wgt + ((long long)(n0 + row0) * p.k_ld) * ES + kc * 16 + koff;
      char *dst = sB + ring_slot * B_BYTES + lwave * 1024;
#pragma unroll
      for (int i = 0; i < NBL; ++i)
        __builtin_amdgcn_global_load_lds((gbl_void *)(src0 + (long long)i * RPL * p.k_ld * ES),
                                         (lds_void *)(dst + i * (NL * 16)), 16, 0, 0);
    };
    auto tile_m0 = [&](int tile) { return (tile / p.n_tiles) * BM; };
    auto tile_n0 = [&](int tile) { return (tile % p.n_tiles) * BN; };
    // prologue of the first tile only
    {
      const long long q0 = (long long)tile_m0(tile0) - p.W - 1;
      for (int pass = 0; pass < p.na; ++pass) issue_halo_pass(q0, 0, pass, 0, true);
#pragma unroll
      for (int j = 0; j <= D; ++j) issue_weights(tile_n0(tile0), j, j);
    }
    int ring = (D + 1) % NSB;                          // slot that receives the next weight tile
    int gchunk = 0;                                    // chunks consumed so far (halo buffer = gchunk & 1)
    int gstep = 0;                                     // K-steps issued so far, all tiles
    for (int tile = tile0; tile < n_tiles_total; tile += grid) {
      const int next_tile = tile + grid;
      const bool has_next = next_tile < n_tiles_total;
      const long long q0 = (long long)tile_m0(tile) - p.W - 1;
      const long long q0n = (long long)tile_m0(has_next ? next_tile : tile) - p.W - 1;
      const int n0 = tile_n0(tile), n0n = tile_n0(has_next ? next_tile : tile);
      int tap = 0, chunk = 0;
#pragma unroll 1
      for (int it = 0; it < nit; ++it) {
        // weights first, halo slices after: the counted wait leaves the youngest halo slices in flight (see above)
        if (D == 2) { if (gstep == 0) wait_vmcnt<NBL>(); else if (gstep == 1) wait_vmcnt_n<PER>(); else wait_vmcnt_n<PER + HPS>(); }
        else { if (gstep == 0) wait_vmcnt<0>(); else wait_vmcnt_n<HPS>(); }
        ++gstep;
        __builtin_amdgcn_s_barrier();
        // weight tile D + 1 steps ahead: this tile's, the next tile's, or (nothing left) the last one again
        const int itw = it + 1 + D;
        if (itw < nit) issue_weights(n0, itw, ring);
        else if (has_next) issue_weights(n0n, itw - nit, ring);
        else issue_weights(n0, nit - 1, ring);
        // two slices of the next chunk's halo: this tile's chunk + 1, or chunk 0 of the next tile
        const bool in_tile = chunk + 1 < p.nchunks;
        const bool live = in_tile || has_next;
        const int p0 = 2 * tap < p.na ? 2 * tap : p.na - 1;
        const int p1 = 2 * tap + 1 < p.na ? 2 * tap + 1 : p.na - 1;
        const int nbuf = (gchunk + 1) & 1;
        issue_halo_pass(in_tile ? q0 : q0n, in_tile ? chunk + 1 : 0, p0, nbuf, live);
        issue_halo_pass(in_tile ? q0 : q0n, in_tile ? chunk + 1 : 0, p1, nbuf, live);
        ring = ring + 1 == NSB ? 0 : ring + 1;
        if (++tap == 9) { tap = 0; ++chunk; ++gchunk; }
      }
      __builtin_amdgcn_s_barrier();                    // E: consumers are done with the last chunk's halo buffer
    }
    wait_vmcnt<0>();
    return;
  }

  // ---------------- consumer waves ----------------
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int fr = lane & 15, fq = lane >> 4;
  const int a_lane_row = wm * 64 + fr;
  const int b_lane_row = wn * 64 + fr;
  const int b_off0 = b_lane_row * 128 + (((0 + fq) ^ (b_lane_row & 7)) << 4);
  const int b_off1 = b_lane_row * 128 + (((4 + fq) ^ (b_lane_row & 7)) << 4);
  const bool leaky = p.flags & Y3_F_LEAKY;
  const bool has_res = p.flags & Y3_F_RESIDUAL;
  int ring = 0, gchunk = 0;
  for (int tile = tile0; tile < n_tiles_total; tile += grid) {
    const int m0 = (tile / p.n_tiles) * BM;
    const int n0 = (tile % p.n_tiles) * BN;
    uint32_t tapmask[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const uint32_t m = (uint32_t)(m0 + wm * 64 + mi * 16 + fr);
      uint32_t mask = 0u;
      if (m < (uint32_t)p.M) {
        const uint32_t img = (__umulhi(m, p.mul_hw) + m) >> p.sh_hw;
        const uint32_t rem = m - img * (uint32_t)p.HW;
        const uint32_t oy = (__umulhi(rem, p.mul_w) + rem) >> p.sh_w;
        const uint32_t ox = rem - oy * (uint32_t)p.W;
        const uint32_t vx = (ox >= 1u ? 1u : 0u) | 2u | (ox + 1u < (uint32_t)p.W ? 4u : 0u);
        mask = (oy >= 1u ? vx : 0u) | (vx << 3) | (oy + 1u < (uint32_t)p.H ? vx << 6 : 0u);
      }
      tapmask[mi] = mask;
    }
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto read_frags = [&](u32x4 (&xf)[MI], u32x4 (&wf)[NI], const char *aBuf, const char *bBuf, int a_shift, int g) {
      const int r0 = a_lane_row + a_shift;
      const char *ap = aBuf + r0 * 128 + (((g * 4 + fq) ^ (r0 & 7)) << 4);
      const char *bp = bBuf + (g ? b_off1 : b_off0);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) xf[mi] = *reinterpret_cast<const u32x4 *>(ap + mi * 2048);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[ni] = *reinterpret_cast<const u32x4 *>(bp + ni * 2048);
    };
    auto mma_all = [&](u32x4 (&xf)[MI], const u32x4 (&wf)[NI], int tap) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        if (!((tapmask[mi] >> tap) & 1u)) xf[mi] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) MmaH<T>::run(acc[mi][ni], wf[ni], xf[mi]);
    };
    auto interleave = [&]() {
#pragma unroll
      for (int i = 0; i < MI + NI; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, sizeof(T) == 2 ? 2 : 8, 0);
      }
    };
    __builtin_amdgcn_s_barrier();                      // B(0) of this tile
    u32x4 xf0[MI], wf0[NI], xf1[MI], wf1[NI];
    read_frags(xf0, wf0, sA + (gchunk & 1) * p.a_bytes, sB + ring * B_BYTES, 0, 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    int tap = 0;
#pragma unroll 1
    for (int it = 0; it < nit; ++it) {
      if (it) __builtin_amdgcn_s_barrier();
      const char *aBuf = sA + (gchunk & 1) * p.a_bytes;
      const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
      __builtin_amdgcn_sched_barrier(0);
      read_frags(xf1, wf1, aBuf, sB + ring * B_BYTES, ky * p.W + kx, 1);
      mma_all(xf0, wf0, tap);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      const int tap_n = tap == 8 ? 0 : tap + 1;
      const int gchunk_n = tap == 8 ? gchunk + 1 : gchunk;
      const int ring_n = ring + 1 == NSB ? 0 : ring + 1;
      {
        const int ky_n = (tap_n * 11) >> 5, kx_n = tap_n - ky_n * 3;
        read_frags(xf0, wf0, sA + (gchunk_n & 1) * p.a_bytes, sB + ring_n * B_BYTES, ky_n * p.W + kx_n, 0);
      }
      mma_all(xf1, wf1, tap);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      tap = tap_n;
      gchunk = gchunk_n;
      ring = ring_n;
    }
    __builtin_amdgcn_s_barrier();                      // E: every consumer is done reading the last chunk's halo
    // ---- epilogue, per wave: 16 pixels x 64 channels at a time through a private 4 KiB slice of that buffer ----
    float *sC = reinterpret_cast<float *>(sA + ((gchunk + 1) & 1) * p.a_bytes) + wave * 1024;
    const int oc = lane & 7;                           // 8-channel group of this lane's write-out items
    const int co = n0 + wn * 64 + oc * 8;
    const f32x4 sc_lo = *reinterpret_cast<const f32x4 *>(p.scale + co);
    const f32x4 sc_hi = *reinterpret_cast<const f32x4 *>(p.scale + co + 4);
    const f32x4 bi_lo = *reinterpret_cast<const f32x4 *>(p.bias + co);
    const f32x4 bi_hi = *reinterpret_cast<const f32x4 *>(p.bias + co + 4);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      // residual first: its latency hides behind the LDS round trip
      u32x4 resv[2];
      f32x4 resf[2][2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int m = m0 + wm * 64 + mi * 16 + (lane >> 3) + r * 8;
        if (has_res && m < p.M) {
          if constexpr (sizeof(T) == 2) {
            resv[r] = *reinterpret_cast<const u32x4 *>(p.res + ((long long)m * p.res_ld + co) * ES);
          } else {
            const float *rp = reinterpret_cast<const float *>(p.res) + (long long)m * p.res_ld + co;
            resf[r][0] = *reinterpret_cast<const f32x4 *>(rp);
            resf[r][1] = *reinterpret_cast<const f32x4 *>(rp + 4);
          }
        }
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)                  // pixel fr, channels ni*16 + fq*4 .. +3; 16-B chunks XOR-swizzled
        *reinterpret_cast<f32x4 *>(sC + fr * 64 + (((ni * 4 + fq) ^ fr) << 2)) = acc[mi][ni];
      __builtin_amdgcn_s_waitcnt(0xC07F);              // (same wave reads below: LDS ops of a wave complete in order)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int pl = (lane >> 3) + r * 8;            // pixel inside the 16-row group
        const int m = m0 + wm * 64 + mi * 16 + pl;
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(sC + pl * 64 + (((2 * oc) ^ pl) << 2));
        const f32x4 hi = *reinterpret_cast<const f32x4 *>(sC + pl * 64 + (((2 * oc + 1) ^ pl) << 2));
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q] = lo[q] * sc_lo[q] + bi_lo[q];
          v[4 + q] = hi[q] * sc_hi[q] + bi_hi[q];
        }
        if (leaky) {
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = v[q] > 0.f ? v[q] : Y3_LEAKY_SLOPE * v[q];
        }
        if (m < p.M) {
          if (has_res) {
            if constexpr (sizeof(T) == 2) {
              const bf16x8 rv = __builtin_bit_cast(bf16x8, resv[r]);
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q] += (float)rv[q];
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) { v[q] += resf[r][0][q]; v[4 + q] += resf[r][1][q]; }
            }
          }
          T *op = reinterpret_cast<T *>(p.out) + (long long)m * p.out_ld + co;
          if constexpr (sizeof(T) == 2) {
            bf16x8 ov;
#pragma unroll
            for (int q = 0; q < 8; ++q) ov[q] = (bf16_t)v[q];
            *reinterpret_cast<bf16x8 *>(op) = ov;
          } else {
            *reinterpret_cast<f32x4 *>(op) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4 *>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);              // reads done before the next group overwrites the slice
    }
  }
}

struct HaloGeom { int na, hr_pad, a_bytes, nsb; size_t lds; };

// geometry / LDS budget of one tile configuration; nsb == 0: does not fit
HaloGeom halo_geom(int bm, int w) {
  HaloGeom g = {0, 0, 0, 0, 0};
  const int nt = bm * 2, rpp = nt / 8, nb = (128 + rpp - 1) / rpp;
  const int hr = bm + 2 * w + 2;
  g.na = (hr + rpp - 1) / rpp;
  g.hr_pad = g.na * rpp;
  g.a_bytes = g.hr_pad * 128;
  const size_t epi = (size_t)bm * 128 * 4;
  for (int nsb = 4; nsb >= 3; --nsb) {
    // the next chunk's halo slices go out at taps 0..na-1 and must be older than the last allowed loads
    if (g.na > (nsb == 4 ? 7 : 8)) continue;
    size_t lds = (size_t)nsb * nb * rpp * 128 + (size_t)2 * g.a_bytes;
    if (lds < epi) lds = epi;
    if (lds <= 160 * 1024) { g.nsb = nsb; g.lds = lds; return g; }
  }
  return g;
}

template <typename T, int BM, int NSB>
int launch_halo(const HaloArgs &a0, const HaloGeom &g, hipStream_t s) {
  HaloArgs a = a0;
  a.na = g.na; a.hr_pad = g.hr_pad; a.a_bytes = g.a_bytes;
  const int m_tiles = y3_ceil_div(a.M, BM);
  static bool attr_set = false;
  if (!attr_set) {
    Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_halo3x3_kernel<T, BM, NSB>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_halo3x3_kernel<T, BM, NSB>), dim3(m_tiles * a.n_tiles), dim3(BM * 2), g.lds, s, a);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}

template <typename T>
int launch_halo_pp(const HaloArgs &a0, hipStream_t s) {
  HaloArgs a = a0;
  const int hr = 256 + 2 * a.W + 2;
  a.na = y3_ceil_div(hr, 64);
  a.hr_pad = a.na * 64;
  a.a_bytes = a.hr_pad * 128;
  size_t lds = (size_t)3 * 128 * 128 + (size_t)2 * a.a_bytes;
  if (lds < (size_t)256 * 128 * 4) lds = (size_t)256 * 128 * 4;
  Y3_REQUIRE(a.na <= 8 && lds <= 160 * 1024, "halo ping-pong kernel: row width %d does not fit", a.W);
  static bool attr_set = false;
  if (!attr_set) {
    Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_halo3x3_pp_kernel<T>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_halo3x3_pp_kernel<T>), dim3(y3_ceil_div(a.M, 256) * a.n_tiles), dim3(512), lds, s, a);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}

template <typename T>
int launch_halo32(const HaloArgs &a0, hipStream_t s) {
  HaloArgs a = a0;
  constexpr int ES = sizeof(T);
  const int hr = 256 + 2 * a.W + 2;
  a.na = y3_ceil_div(hr, 128);
  a.hr_pad = a.na * 128;
  a.a_bytes = a.hr_pad * 64;
  a.nchunks = a.Cin / (64 / ES);
  size_t lds = (size_t)5 * 128 * 64 + 8 * 1024 + (size_t)2 * a.a_bytes;
  if (lds < (size_t)256 * 128 * 4) lds = (size_t)256 * 128 * 4;
  Y3_REQUIRE(a.na <= 4 && lds <= 160 * 1024 && a.nchunks % 2 == 0 && a.nchunks >= 2,
             "halo32 kernel: shape does not fit (W %d, %d chunks)", a.W, a.nchunks);
  static bool attr_set = false;
  if (!attr_set) {
    Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_halo32_kernel<T>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_halo32_kernel<T>), dim3(y3_ceil_div(a.M, 256) * a.n_tiles), dim3(512), lds, s, a);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}


template <typename T>
int launch_halo_ws(const HaloArgs &a0, hipStream_t s) {
  HaloArgs a = a0;
  const int hr = 256 + 2 * a.W + 2;
  a.na = y3_ceil_div(hr, 32);
  a.hr_pad = a.na * 32;
  a.a_bytes = a.hr_pad * 128;
  // the next chunk's halo goes out two passes per K-step and must be older than the last loads allowed in flight
  int nsb = 0;
  size_t lds = 0;
  for (int c = 4; c >= 3; --c) {
    if (a.na > (c == 4 ? 12 : 14)) continue;   // all real halo slices out by tap 5 (4 slots) / 6 (3 slots)
    lds = (size_t)c * 128 * 128 + (size_t)2 * a.a_bytes;
    if (lds < (size_t)256 * 128 * 4) lds = (size_t)256 * 128 * 4;
    if (lds <= 160 * 1024) { nsb = c; break; }
  }
  Y3_REQUIRE(nsb != 0, "wave-specialised halo kernel: row width %d does not fit", a.W);
  static bool attr_set = false;
  if (!attr_set) {
    Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_halo_ws_kernel<T, 3>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_halo_ws_kernel<T, 4>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const dim3 grid(y3_ceil_div(a.M, 256) * a.n_tiles);
  if (nsb == 4) hipLaunchKernelGGL((conv_halo_ws_kernel<T, 4>), grid, dim3(768), lds, s, a);
  else hipLaunchKernelGGL((conv_halo_ws_kernel<T, 3>), grid, dim3(768), lds, s, a);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}

template <typename T>
int launch_halo_wsp(const HaloArgs &a0, hipStream_t s) {
  HaloArgs a = a0;
  const int hr = 256 + 2 * a.W + 2;
  a.na = y3_ceil_div(hr, 32);
  a.hr_pad = a.na * 32;
  a.a_bytes = a.hr_pad * 128;
  int nsb = 0;
  size_t lds = 0;
  for (int c = 4; c >= 3; --c) {
    if (a.na > (c == 4 ? 12 : 14)) continue;   // all real halo slices out by tap 5 (4 slots) / 6 (3 slots)
    lds = (size_t)c * 128 * 128 + (size_t)2 * a.a_bytes;
    if (lds <= 160 * 1024) { nsb = c; break; }
  }
  Y3_REQUIRE(nsb != 0 && a.a_bytes >= 32 * 1024, "persistent halo kernel: row width %d does not fit", a.W);
  static bool attr_set = false;
  static int n_cu = 0;
  if (!attr_set) {
    Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_halo_wsp_kernel<T, 3>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_halo_wsp_kernel<T, 4>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int dev = 0;
    Y3_HIP_CHECK(hipGetDevice(&dev));
    Y3_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    attr_set = true;
  }
  const int tiles = y3_ceil_div(a.M, 256) * a.n_tiles;
  const int grid = tiles < n_cu ? tiles : n_cu;
  if (nsb == 4) hipLaunchKernelGGL((conv_halo_wsp_kernel<T, 4>), dim3(grid), dim3(768), lds, s, a, tiles);
  else hipLaunchKernelGGL((conv_halo_wsp_kernel<T, 3>), dim3(grid), dim3(768), lds, s, a, tiles);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}

}  // namespace

int g_y3_halo_pp = 1;
int g_y3_halo_bm = 0;   // tuning knob "halo_pp": use the ping-pong schedule for 256-pixel tiles

// picks the pixel-tile height (256 or 192) that wastes the fewest CU rounds; 0 = not applicable
bool y3_conv_halo_eligible(const y3_op &op) {
  const int es = y3_elem_size(op.dtype);
  const int bke = 128 / es;
  if (op.ksize != 3 || op.stride != 1 || op.pad != 1) return false;
  if (op.flags & (Y3_F_OUT_F32 | Y3_F_IN_NCHW_F32 | Y3_F_IN_NHWC_U8BGR)) return false;
  if (op.in_c % bke != 0 || op.in_c / bke < 2) return false;
  if (op.out_c % 128 != 0 || op.out_ld % 8 != 0 || op.in_ld % (16 / es) != 0) return false;
  if ((op.flags & Y3_F_RESIDUAL) && op.res_ld % 8 != 0) return false;
  if (op.k_ld < 9 * op.in_c) return false;
  return true;
}

// wave-specialised 256x128 halo kernel: additionally the halo image must fit (2 * (258 + 2W) rows + 3 weight slots)
bool y3_conv_halo_ws_fits(const y3_op &op) {
  if (!y3_conv_halo_eligible(op)) return false;
  const int na = y3_ceil_div(256 + 2 * op.in_w + 2, 32);
  return na <= 14 && (size_t)3 * 128 * 128 + (size_t)2 * na * 32 * 128 <= 160 * 1024;
}

int y3_conv_halo_bm(const y3_op &op) {
  if (!y3_conv_halo_eligible(op)) return 0;
  const long long m = (long long)op.batch * op.out_h * op.out_w;
  const int n_tiles = op.out_c / 128;
  int best = 0;
  double best_eff = 0.0;
  const int cands[2] = {256, 192};
  const bool pp_only = g_y3_halo_pp && !g_y3_halo_bm;   // default: only the ping-pong 256-pixel tile, and only
                                                         // where it fills the 256 CUs evenly (else implicit GEMM)
  for (int bm : cands) {
    const HaloGeom g = halo_geom(bm, op.in_w);
    if (g.nsb == 0) continue;
    if (g_y3_halo_bm && bm != g_y3_halo_bm) continue;
    if (pp_only && bm != 256) continue;
    const double blocks = (double)((m + bm - 1) / bm) * n_tiles;
    const double rounds = blocks / 256.0;
    const double eff = rounds / (double)(long long)(rounds + 0.999999);
    const double score = eff * (bm == 256 ? 1.04 : 1.0) * (g.nsb == 4 ? 1.03 : 1.0);
    if (pp_only && eff < 0.9) continue;
    if (score > best_eff) { best_eff = score; best = bm; }
  }
  return best;
}

int y3_launch_conv_halo(const y3_op &op, int bm, const void *d_in, const void *d_zero, hipStream_t s,
                        const char **kernel_name, bool dry_run, int variant) {
  const int pp = variant >= 0 ? variant : g_y3_halo_pp;
  const int es = y3_elem_size(op.dtype);
  const bool bf = op.dtype == Y3_BF16;
  Y3_REQUIRE(bm == 256 || bm == 192, "conv block %d: bad halo tile %d", op.block_idx, bm);
  if (bm == 256) *kernel_name = bf ? "conv_halo3x3_bf16_256x128" : "conv_halo3x3_f32_256x128";
  else *kernel_name = bf ? "conv_halo3x3_bf16_192x128" : "conv_halo3x3_f32_192x128";
  const bool use32 = bm == 256 && pp == 2 && op.in_w <= 126 && (op.in_c / (64 / es)) % 2 == 0;
  if (use32) *kernel_name = bf ? "conv_halo32_bf16_256x128" : "conv_halo32_f32_256x128";
  const bool use_ws = bm == 256 && pp == 3;
  if (use_ws) *kernel_name = bf ? "conv_halo_ws_bf16_256x128" : "conv_halo_ws_f32_256x128";
  const bool use_wsp = bm == 256 && pp == 4;
  if (use_wsp) *kernel_name = bf ? "conv_halo_wsp_bf16_256x128" : "conv_halo_wsp_f32_256x128";
  if (dry_run) return Y3_OK;
  HaloArgs a;
  a.in = static_cast<const char *>(d_in);
  a.wgt = static_cast<const char *>(op.d_weight);
  a.scale = op.d_scale; a.bias = op.d_bias;
  a.res = static_cast<const char *>(op.d_res);
  a.out = static_cast<char *>(op.d_out);
  a.zero = static_cast<const char *>(d_zero);
  a.H = op.in_h; a.W = op.in_w; a.Cin = op.in_c; a.in_ld = op.in_ld;
  a.Cout = op.out_c; a.out_ld = op.out_ld; a.res_ld = op.res_ld;
  a.HW = op.in_h * op.in_w;
  a.M = op.batch * a.HW;
  a.k_ld = op.k_ld;
  a.nchunks = op.in_c / (128 / es);
  a.n_tiles = op.out_c / 128;
  a.hr_pad = a.na = a.a_bytes = 0;
  fast_div((uint32_t)a.HW, a.mul_hw, a.sh_hw);
  fast_div((uint32_t)a.W, a.mul_w, a.sh_w);
  a.flags = op.flags;
  Y3_REQUIRE((long long)op.batch * a.HW < (1ll << 31), "conv block %d: too many pixels for the 32-bit tile index", op.block_idx);
  const HaloGeom g = halo_geom(bm, op.in_w);
  Y3_REQUIRE(g.nsb != 0, "conv block %d: halo tile does not fit in LDS", op.block_idx);
  if (use32) return bf ? launch_halo32<bf16_t>(a, s) : launch_halo32<float>(a, s);
  if (use_ws) return bf ? launch_halo_ws<bf16_t>(a, s) : launch_halo_ws<float>(a, s);
  if (use_wsp) return bf ? launch_halo_wsp<bf16_t>(a, s) : launch_halo_wsp<float>(a, s);
  if (bm == 256 && pp) return bf ? launch_halo_pp<bf16_t>(a, s) : launch_halo_pp<float>(a, s);
  if (bf) {
    if (bm == 256) return g.nsb == 4 ? launch_halo<bf16_t, 256, 4>(a, g, s) : launch_halo<bf16_t, 256, 3>(a, g, s);
    return g.nsb == 4 ? launch_halo<bf16_t, 192, 4>(a, g, s) : launch_halo<bf16_t, 192, 3>(a, g, s);
  }
  if (bm == 256) return g.nsb == 4 ? launch_halo<float, 256, 4>(a, g, s) : launch_halo<float, 256, 3>(a, g, s);
  return g.nsb == 4 ? launch_halo<float, 192, 4>(a, g, s) : launch_halo<float, 192, 3>(a, g, s);
}

Y3_STAMP_READER(y3_debug_stamps_halo)
