// 3x3 stride-1 convolution with input-halo reuse (gfx950), bf16 / fp32, NHWC.
//
// Same contract as conv_igemm.hip (conv -> scale/bias -> LeakyReLU -> +residual; replaces
// /root/reference/yolov3/darknet.py:244-257 and the shortcut at :376-379), specialised for the
// layers that carry 70 % of Darknet-53's FLOPs: 3x3, stride 1, pad 1, Cin >= 2 K-tiles
// (K-tile = 64 bf16 / 32 fp32 channels = 128 bytes), Cout a multiple of 128.
//
// Why: the generic implicit GEMM re-reads every input pixel once per filter tap (9x) and every
// weight tile once per 128-pixel tile; measured, its K loop is bound by the LDS-DMA ingest path
// (1 KiB per ~16-22 cycles per CU whatever the number of issuing waves: a 128x128x64 tile needs as
// many cycles of it as of MFMA).  Here a workgroup owns 256 consecutive output pixels in raster
// order (b, y, x flattened) x 128 output channels and stages, per 128-byte channel chunk, the
// 256 + 2W + 2 input pixels that ALL nine taps of those outputs touch -- once.  Tap (ky,kx) of
// output pixel p is halo row  p + ky*W + kx, so the nine K-steps of a chunk read the same LDS
// image at nine row offsets.  Row-wrap / image-border taps (the zero padding): conv_halo_ws_kernel
// redirects the lane's fragment address to an all-zero row of the halo image (no data instructions).  Only the weight tile (128 rows x
// 128 B) changes per K-step; it streams through a 3- or 4-slot LDS ring.  Bytes through the LDS-DMA
// path per FLOP drop ~3x versus the 128x128 implicit GEMM.
//
// Two kernels: conv_halo_ws_kernel (raster strip, one tile per workgroup; default for rows of up to 128 pixels) and
// conv_patch_wsp_kernel (persistent, 8 x 32 output tiles with a 2-D input patch and no fragment masking; default for
// wider rows, where the strip's halo outgrows LDS).  Measured slower and removed (history keeps them): schedules where
// every wave loads and computes in lock step, ping-pong wave groups, 32-channel chunks with a deeper ring
// (profiles/r01_convbench_*.txt); two workgroups per CU with 128 x 128 tiles (profiles/HISTORY.md 3.1c); two persistent forms of
// the strip kernel, the second with the epilogue in registers and the finished tile drained by the loader waves
// (profiles/r03b_persistent_halo_wsq_investigation.txt).
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
  int ngrp_w, m_tiles; // strip kernel's tile order: channel tiles in groups of ngrp_w, pixel tiles inside a group (see launch_halo_ws)
  int hr;              // halo rows that hold pixels (256 + 2W + 2); strip kernel: row `hr` of each buffer is all zero
  int hr_pad;          // halo rows, padded to a multiple of the loader's rows-per-pass
  int na;              // loader passes per halo (= glds per thread per halo)
  int a_bytes;         // hr_pad * 128
  uint32_t mul_hw, sh_hw, mul_w, sh_w;   // n / d == (umulhi(n, mul) + n) >> sh  for n < 2^31
  uint32_t flags;
};

template <typename T>
struct MmaH {   // bf16 / IEEE half: one v_mfma_f32_16x16x32 per 64-byte K-half
  static __device__ __forceinline__ void run(f32x4 &acc, const u32x4 &w, const u32x4 &x) { acc = y3_mfma16<T>(w, x, acc); }
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

template <int V>
struct TapC { static constexpr int value = V; };

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
template <typename T, int NSB, int MI>
__global__ __launch_bounds__(768, 3) void conv_halo_ws_kernel(HaloArgs p) {
  // MI = 16-pixel fragments per MFMA wave: 4 -> 256-pixel tiles; 3 -> 192-pixel tiles (wave tile 48 x 64), chosen by the
  // launcher where 256-pixel tiles leave a large part of the last round of workgroups empty (19^2, 38^2 at batch 16)
  constexpr int WM = MI * 16;                         // pixels per MFMA wave
  constexpr int BM = 4 * WM, BN = 128;
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
  constexpr int NI = 4;
  static_assert(NSB == 3 || NSB == 4, "weight ring has 3 or 4 slots");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *sB = smem;                                    // [NSB][BN][128]
  char *sA = smem + NSB * B_BYTES;                    // [2][hr_pad][128]

#ifdef Y3_X_STAGGER
  // timing experiment: every other first-round workgroup of an XCD starts half a tile period late, so that the CUs' prologues
  // and epilogues (bursts of memory traffic) no longer coincide
  if ((p.flags & 0x40000000u) && blockIdx.x < 256 && ((blockIdx.x >> 3) & 1)) {
    __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);
  }
#endif
  Y3_STAMP_DECL
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= NC / 64;

  const int tile = y3_xcd_remap(blockIdx.x, gridDim.x);
  // linear order: channel-tile group (ngrp_w wide) outermost, then pixel tile, then channel tile inside the group
  const int grp = tile / (p.m_tiles * p.ngrp_w), rem = tile - grp * (p.m_tiles * p.ngrp_w);
  const int m0 = (rem / p.ngrp_w) * BM;
  const int n0 = (grp * p.ngrp_w + rem % p.ngrp_w) * BN;
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
    // Halo row of pass `pass`: row0 + 32 pass, input pixel q0 + that row.  The row is real (inside the halo image and the
    // tensor) for a per-lane RANGE of passes, and its address is the lane's pass-0 address plus a wave-uniform step: the
    // loop issues each slice with two compares, one 64-bit add and a select instead of recomputing row, pixel and product.
    const long long qr = q0 + row0;
    const int hi_rows = (p.hr - 1 - row0) >> 5;                                        // last pass with row < hr (-1: none)
    const long long hi_px = qr < p.M ? (p.M - 1 - qr) >> 5 : -1;                        // last pass with pixel < M
    const int pass_lo = qr >= 0 ? 0 : (int)((-qr + 31) >> 5);                           // first pass with pixel >= 0
    const int pass_hi = hi_px < hi_rows ? (int)hi_px : hi_rows;
    const char *hsrc0 = p.in + (qr * p.in_ld) * ES + kc * 16;                           // pass 0, chunk 0 (maybe out of range)
    const long long pass_step = (long long)RPL * p.in_ld * ES;
    auto issue_halo_pass = [&](int chunk, int pass, bool live) {
      const bool ok = live && pass >= pass_lo && pass <= pass_hi;   // other rows stay zero: row hr is the consumers' zero row
      const char *src = ok ? hsrc0 + (pass * pass_step + (long long)chunk * (BKE * ES)) : p.zero;
      char *dst = sA + (chunk & 1) * p.a_bytes + pass * (NL * 16) + lwave * 1024;
      __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)dst, 16, 0, Y3_AUX_H);
    };
    const bool has_res = (p.flags & Y3_F_RESIDUAL) != 0;
    // 4 KiB slice `sl` of this tile's 256 x 128-channel shortcut operand (row = pixel, 128 * ES bytes per row)
    auto issue_res_slice = [&](int sl, int chunk) {
      constexpr int ROWB = BN * ES;                    // bytes of the tile per pixel
      const int off = sl * 4096 + ltid * 16;
      const int px = off / ROWB, cb = off - px * ROWB;
      const long long m = (long long)m0 + px;
      const bool ok = px < BM && m < p.M;
      const char *src = ok ? p.res + (m * p.res_ld + n0) * ES + cb : p.zero;
      const int pass = sl < p.na - 1 ? sl : p.na - 2;  // any slice of the idle halo buffer but the last (zero row)
      char *dst = sA + (chunk & 1) * p.a_bytes + pass * (NL * 16) + lwave * 1024;
      __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)dst, 16, 0, 0);
    };
    // weight rows: wave-uniform base (channel tile, pass, K offset) + this lane's 32-bit offset (row, chunk): the compiler
    // picks the scalar-base addressing form, no vector arithmetic per piece
    const uint32_t b_lane = (uint32_t)row0 * (uint32_t)p.k_ld * ES + kc * 16;
    const char *b_tile = p.wgt + ((long long)n0 * p.k_ld) * ES;
    const long long b_pass = (long long)RPL * p.k_ld * ES;
    auto issue_weights = [&](int it, int ring_slot) {  // it = chunk*9 + tap; K offset = tap*Cin + chunk*BKE elements
      char *dst = sB + ring_slot * B_BYTES + lwave * 1024;
      if (it < nit) {
        const int chunk = it / 9, tap = it - chunk * 9;
        const long long koff = ((long long)tap * p.Cin + (long long)chunk * BKE) * ES;
#pragma unroll
        for (int i = 0; i < NBL; ++i)
          __builtin_amdgcn_global_load_lds((gbl_void *)(b_tile + (koff + i * b_pass) + b_lane), (lds_void *)(dst + i * (NL * 16)), 16, 0, Y3_AUX_W);
      } else {
        // past the last K-step: the instruction count per step stays (counted waits) but the pieces read the zero page --
        // a real weight tile here is 16 KiB nobody uses, and the epilogue's barrier waits for it to land
#pragma unroll
        for (int i = 0; i < NBL; ++i)
          __builtin_amdgcn_global_load_lds((gbl_void *)p.zero, (lds_void *)(dst + i * (NL * 16)), 16, 0, 0);
      }
    };
    for (int pass = 0; pass < p.na; ++pass) issue_halo_pass(0, pass, true);
#pragma unroll
    for (int j = 0; j <= D; ++j) issue_weights(j, j);
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
      issue_weights(it + 1 + D, ring);                // ring == (it + 1 + D) % NSB: the slot of weights(it-1), free
      if (live || !has_res) {
        issue_halo_pass(chunk + 1, p0, live);
        issue_halo_pass(chunk + 1, p1, live);
      } else {
        // Last chunk: there is no next halo, but the two LDS-DMA instructions are issued regardless (fixed count
        // per step for the counted waits).  They stream the tile's SHORTCUT operand instead -- into the idle halo
        // buffer, where nobody reads it: the point is that the epilogue's residual loads (issued after the K loop with
        // nothing left to hide their latency) then hit L2.  Residual layers ran 4-9 % below identical layers without.
        issue_res_slice(2 * tap, chunk + 1);
        issue_res_slice(2 * tap + 1, chunk + 1);
      }
      ring = ring + 1 == NSB ? 0 : ring + 1;
      if (++tap == 9) { tap = 0; ++chunk; }
      Y3_COARSE(5);
    }
    wait_vmcnt<0>();                                  // the tail's dummy loads must not land on the output tile
#if defined(Y3_STAMPS) && !defined(Y3_STAMPS_FINE) && !defined(Y3_STAMPS_CLOCK)
    if (tid == NC) for (int _i = 3; _i < 7; ++_i) atomicAdd(&g_y3_stamps[_i], _st_acc[_i]);
#endif
  } else {
    // ---------------- consumer waves ----------------
    // the younger MFMA wave of each SIMD loses every issue arbitration to its older partner (age order); a static
    // priority for that half evens the pair out (MI355X_MICROARCH.md, two waves per SIMD, item 4): +1-5 % measured
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    // Which border taps a lane's pixel lacks, as wave-wide LANE MASKS in scalar registers (one per fragment and border):
    // a tap's select below then costs one v_cndmask on a scalar pair instead of shift / and / compare in every lane and
    // step, and the centre tap needs no select at all.  Pixels beyond the tensor (m >= M) need no masking: their halo rows
    // were loaded as zeros or are other frames' pixels, and their accumulators are never stored.
    // (bf16 instantiations; float32 sits at the 168-VGPR / 102-SGPR caps and keeps the per-lane nine-bit mask)
    constexpr bool TAP_REGS = sizeof(T) == 2;
    unsigned long long mk_top[MI], mk_bot[MI], mk_left[MI], mk_right[MI];
    uint32_t tapmask[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const uint32_t m = (uint32_t)(m0 + wm * WM + mi * 16 + fr);
      const uint32_t img = (__umulhi(m, p.mul_hw) + m) >> p.sh_hw;
      const uint32_t rem = m - img * (uint32_t)p.HW;
      const uint32_t oy = (__umulhi(rem, p.mul_w) + rem) >> p.sh_w;
      const uint32_t ox = rem - oy * (uint32_t)p.W;
      if constexpr (TAP_REGS) {
        mk_top[mi] = __builtin_amdgcn_ballot_w64(oy >= 1u);
        mk_bot[mi] = __builtin_amdgcn_ballot_w64(oy + 1u < (uint32_t)p.H);
        mk_left[mi] = __builtin_amdgcn_ballot_w64(ox >= 1u);
        mk_right[mi] = __builtin_amdgcn_ballot_w64(ox + 1u < (uint32_t)p.W);
      } else {
        const uint32_t vx = (ox >= 1u ? 1u : 0u) | 2u | (ox + 1u < (uint32_t)p.W ? 4u : 0u);
        tapmask[mi] = (oy >= 1u ? vx : 0u) | (vx << 3) | (oy + 1u < (uint32_t)p.H ? vx << 6 : 0u);
      }
    }
    const int a_lane_row = wm * WM + fr;
    const int b_lane_row = wn * 64 + fr;
    const int b_off0 = b_lane_row * 128 + (((0 + fq) ^ (b_lane_row & 7)) << 4);
    const int b_off1 = b_lane_row * 128 + (((4 + fq) ^ (b_lane_row & 7)) << 4);
    // Border taps (zero padding, row wrap of the raster strip) are not cleared in registers: the fragment's LDS address
    // is redirected to the all-zero row `hr` of halo buffer 0 instead -- one select per fragment of K-half 0; K-half 1
    // of the same tap is the same address with bit 6 flipped (the XOR swizzle), for the zero row as well.  (The K loop
    // is VALU-issue bound: PMC, profiles/HISTORY.md section 6.)
    // (absolute LDS byte addresses, so that nothing but the immediate is added per read; the workgroup's LDS block and
    // the halo buffers are 128-byte aligned, which the bit-6 flip relies on)
    typedef const __attribute__((address_space(3))) u32x4 lds_u32x4;
    const int sA_lds = (int)(size_t)(lds_void *)sA;
    // The zero region is TWO rows (hr, hr + 1: one 256-byte bank row, hr is even) and a masked lane reads the spot of it
    // that has the bank of the address it would have read -- (ap & 255) | zero base, one v_and_or per fragment.  With a
    // single shared zero address the masked lanes of a fragment sat on one bank next to an unmasked lane's access:
    // 2-way conflicts on 11-20 % of the LDS cycles of this kernel (SQ_LDS_BANK_CONFLICT, profiles/), most at 19^2 where
    // nearly every 16-pixel fragment crosses an image row.
    int zalt[MI], sel[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      zalt[mi] = sA_lds + p.hr * 128 - mi * 2048;
      asm volatile("" : "+v"(zalt[mi]));               // lives in a VGPR: v_cndmask takes one scalar operand, the condition
    }
    int low8 = 255;
    asm volatile("" : "+s"(low8));                     // in an SGPR: no VOP3 literals on gfx9
    // fragment addresses of the nine taps at halo buffer 0 (the XOR swizzle makes them more than base + shift): nine
    // registers, computed once
    // (bf16 only: the float32 instantiation recomputes them, five instructions a step)
    int ap_tap[TAP_REGS ? 9 : 1];
    if constexpr (TAP_REGS) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int r0 = a_lane_row + (t / 3) * p.W + (t % 3);
        ap_tap[t] = ((r0 << 7) + sA_lds) + ((fq ^ (r0 & 7)) << 4);
      }
    }
    auto read_frags0 = [&](auto tapc, u32x4 (&xf)[MI], u32x4 (&wf)[NI], int a_off, const char *bBuf) {
      constexpr int tap = decltype(tapc)::value, ky = tap / 3, kx = tap % 3;
      int ap;
      if constexpr (TAP_REGS) {
        ap = ap_tap[tap] + a_off;
      } else {
        const int r0 = a_lane_row + ky * p.W + kx;
        ap = ((r0 << 7) + (a_off + sA_lds)) + ((fq ^ (r0 & 7)) << 4);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        int off = ap;
        if constexpr (tap != 4) {
          int zoff;
          asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(zoff) : "v"(ap), "s"(low8), "v"(zalt[mi]));
          if constexpr (TAP_REGS) {
            unsigned long long ok = ky == 0 ? mk_top[mi] : (ky == 2 ? mk_bot[mi] : ~0ull);
            if constexpr (kx == 0) ok &= mk_left[mi];
            if constexpr (kx == 2) ok &= mk_right[mi];
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(off) : "v"(zoff), "v"(ap), "s"(ok));
          } else {
            off = ((tapmask[mi] >> tap) & 1u) ? ap : zoff;
          }
        }
        asm volatile("" : "+v"(off));                  // keeps `+ mi * 2048` in the ds_read's immediate offset
        sel[mi] = off;
        xf[mi] = *reinterpret_cast<lds_u32x4 *>(off + mi * 2048);
      }
      const char *bp = bBuf + b_off0;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[ni] = *reinterpret_cast<const u32x4 *>(bp + ni * 2048);
    };
    // float32 instantiations: the tap is a run-time value (loop not unrolled, see below)
    auto read_frags0_rt = [&](u32x4 (&xf)[MI], u32x4 (&wf)[NI], int a_off, const char *bBuf, int a_shift, int tap) {
      const int r0 = a_lane_row + a_shift;
      const int ap = ((r0 << 7) + (a_off + sA_lds)) + ((fq ^ (r0 & 7)) << 4);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        int zoff;
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(zoff) : "v"(ap), "s"(low8), "v"(zalt[mi]));
        int off = ((tapmask[mi] >> tap) & 1u) ? ap : zoff;
        asm volatile("" : "+v"(off));                  // keeps `+ mi * 2048` in the ds_read's immediate offset
        sel[mi] = off;
        xf[mi] = *reinterpret_cast<lds_u32x4 *>(off + mi * 2048);
      }
      const char *bp = bBuf + b_off0;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[ni] = *reinterpret_cast<const u32x4 *>(bp + ni * 2048);
    };
    auto read_frags1 = [&](u32x4 (&xf)[MI], u32x4 (&wf)[NI], const char *bBuf) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        int off = sel[mi] ^ 64;
        asm volatile("" : "+v"(off));
        xf[mi] = *reinterpret_cast<lds_u32x4 *>(off + mi * 2048);
      }
      const char *bp = bBuf + b_off1;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[ni] = *reinterpret_cast<const u32x4 *>(bp + ni * 2048);
    };
    auto mma_all = [&](const u32x4 (&xf)[MI], const u32x4 (&wf)[NI]) {
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
    Y3_CLK_BEGIN();
    u32x4 xf0[MI], wf0[NI], xf1[MI], wf1[NI];
    if constexpr (TAP_REGS) {
    read_frags0(TapC<0>{}, xf0, wf0, 0, sB);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    int ring = 0, it = 0;
#if defined(Y3_STAMPS_FINE)
    const bool w0 = wave == 0;
#define Y3_W0(slot) do { if (w0) Y3_STAMP(slot); } while (0)
#else
#define Y3_W0(slot) do {} while (0)
#endif
    // one K-step; the tap is a compile-time constant (the chunk loop below is unrolled over its nine taps), so tap
    // shifts, mask choices and the next tap's address register are fixed per copy of the body
    auto kstep = [&](auto tapc, int chunk) {
      constexpr int tap = decltype(tapc)::value, tap_n = tap == 8 ? 0 : tap + 1;
      if (it) {
        __builtin_amdgcn_s_barrier();                 // B(it): weights(it+1) landed; slot of weights(it-1) released
        Y3_STAMP(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      read_frags1(xf1, wf1, sB + ring * B_BYTES);
      mma_all(xf0, wf0);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      Y3_W0(2);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      Y3_W0(3);
      const int chunk_n = tap == 8 ? chunk + 1 : chunk;
      const int ring_n = ring + 1 == NSB ? 0 : ring + 1;
      read_frags0(TapC<tap_n>{}, xf0, wf0, (chunk_n & 1) * p.a_bytes, sB + ring_n * B_BYTES);
      mma_all(xf1, wf1);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      Y3_W0(4);
      // (no s_waitcnt here: until round 6 every wave drained its fragment reads BEFORE the barrier -- the reads issued in this half
      // are of the NEXT step's slot and halo buffer, which the barrier does not release, and the compiler waits for each fragment
      // where its first MFMA needs it: the barrier's latency and the reads' now overlap.  +1.3 % on the layer mix of batch 16, +4-9 % on
      // the 19^2 layers alone; dropping the mid-step wait as well: level.  profiles/r06_ws_tail_wait.txt)
#if defined(Y3_STAMPS_FINE)
      if (w0) Y3_STAMP(5); else Y3_STAMP(1);
#else
      Y3_STAMP(1);
#endif
      ring = ring_n;
      ++it;
    };
#pragma unroll 1
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
      kstep(TapC<0>{}, chunk); kstep(TapC<1>{}, chunk); kstep(TapC<2>{}, chunk);
      kstep(TapC<3>{}, chunk); kstep(TapC<4>{}, chunk); kstep(TapC<5>{}, chunk);
      kstep(TapC<6>{}, chunk); kstep(TapC<7>{}, chunk); kstep(TapC<8>{}, chunk);
    }
    } else {
    read_frags0_rt(xf0, wf0, 0, sB, 0, 0);
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
      __builtin_amdgcn_sched_barrier(0);
      read_frags1(xf1, wf1, sB + ring * B_BYTES);
      mma_all(xf0, wf0);
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
        read_frags0_rt(xf0, wf0, (chunk_n & 1) * p.a_bytes, sB + ring_n * B_BYTES, ky_n * p.W + kx_n, tap_n);
      }
      mma_all(xf1, wf1);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      Y3_W0(4);
      // (no s_waitcnt before the barrier: see the 16-bit loop above)
#if defined(Y3_STAMPS_FINE)
      if (w0) Y3_STAMP(5); else Y3_STAMP(1);
#else
      Y3_STAMP(1);
#endif
      tap = tap_n;
      chunk = chunk_n;
      ring = ring_n;
    }
    }
    Y3_CLK_END();
#if defined(Y3_STAMPS_CLOCK)
#elif defined(Y3_STAMPS_FINE)
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
  // ---- epilogue: ALL twelve waves write out (round 5).  Until round 4 the four loader waves only kept the barrier count
  // and the eight MFMA waves wrote the tile, eight (six) 8-channel items per lane; now a lane of any wave takes the items
  // pl = (tid >> 4) + 48 j: 5.33 (4) per lane, and the loaders issue the scale / bias / shortcut loads of THEIR items while the
  // MFMA waves are still parking their accumulators.  Same arithmetic per item, same bits.
  constexpr int SWZ = 15;
  constexpr int NT = NC + NL;                        // 768
  constexpr int RPP = NT / 16;                       // pixel rows written per pass of the workgroup (48)
  constexpr int WR = (BM + RPP - 1) / RPP;           // 6 passes at 256 pixels (the last one: rows 240 .. 255, the first 256 lanes), 4 at 192
  float *sC = reinterpret_cast<float *>(smem);
  const bool leaky = p.flags & Y3_F_LEAKY;
  const bool has_res = p.flags & Y3_F_RESIDUAL;
  const int oc_mine = tid & 15;
  const int co = n0 + oc_mine * 8;
  f32x4 sc_lo, sc_hi, bi_lo, bi_hi;
  u32x4 resv[WR];
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();   // all operand reads and all LDS-DMA done: LDS can hold the output tile
  Y3_EPI(4);
  if (!loader) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int cl = wn * 64 + ni * 16 + fq * 4;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int pl = wm * WM + mi * 16 + fr;
        *reinterpret_cast<f32x4 *>(sC + pl * BN + (((cl >> 2) ^ (pl & SWZ)) << 2)) = acc[mi][ni];
      }
    }
  }
  // the epilogue's global reads go out AFTER the accumulators are on their way to LDS (MFMA waves; the loaders start at once):
  // issued before the barrier above (rounds 1-3) they held every wave back from it (stamps: 1740 cycles from the end of the K
  // loop to the barrier with a shortcut, 590 without)
  sc_lo = *reinterpret_cast<const f32x4 *>(p.scale + co);
  sc_hi = *reinterpret_cast<const f32x4 *>(p.scale + co + 4);
  bi_lo = *reinterpret_cast<const f32x4 *>(p.bias + co);
  bi_hi = *reinterpret_cast<const f32x4 *>(p.bias + co + 4);
  if (has_res) {
#pragma unroll
    for (int j = 0; j < WR; ++j) {
      const int pl = (tid >> 4) + j * RPP;
      const int m = m0 + pl;
      const char *rp = p.res + ((long long)m * p.res_ld + co) * ES;
      if constexpr (sizeof(T) == 2) {
        resv[j] = (pl < BM && m < p.M) ? *reinterpret_cast<const u32x4 *>(rp) : u32x4{0u, 0u, 0u, 0u};
      }
    }
  }
  __syncthreads();
  Y3_EPI(5);
#pragma unroll
  for (int j = 0; j < WR; ++j) {
    const int pl = (tid >> 4) + j * RPP;
    const int m = m0 + pl;
    if (pl >= BM || m >= p.M) continue;
    const f32x4 lo = *reinterpret_cast<const f32x4 *>(sC + pl * BN + (((2 * oc_mine) ^ (pl & SWZ)) << 2));
    const f32x4 hi = *reinterpret_cast<const f32x4 *>(sC + pl * BN + (((2 * oc_mine + 1) ^ (pl & SWZ)) << 2));
    float v[8];
    y3_bn_leaky8(v, lo, hi, sc_lo, sc_hi, bi_lo, bi_hi, leaky);
    if (has_res) {
      if constexpr (sizeof(T) == 2) {
        y3_add8<T>(v, resv[j]);
      } else {
        const float *rp = reinterpret_cast<const float *>(p.res) + (long long)m * p.res_ld + co;
        const f32x4 r0 = *reinterpret_cast<const f32x4 *>(rp), r1 = *reinterpret_cast<const f32x4 *>(rp + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] += r0[r]; v[4 + r] += r1[r]; }
      }
    }
    T *op = reinterpret_cast<T *>(p.out) + (long long)m * p.out_ld + co;
    if constexpr (sizeof(T) == 2) {
      *reinterpret_cast<u32x4 *>(op) = y3_pack8<T>(v);
    } else {
      *reinterpret_cast<f32x4 *>(op) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4 *>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
  }
  Y3_CLK_TAIL();
}

// ------------------------------------------------------------------------------------------------
// 2-D patch form of the persistent wave-specialised kernel, for wide feature maps (rows of more than 128 pixels,
// where the raster strip's halo of 2W + 2 rows no longer fits in LDS): a workgroup owns an 8 x 32 output tile and
// stages the 10 x 34 input patch per channel chunk (one LDS row per patch pixel, out-of-frame pixels as zeros).  Tap
// (ky, kx) of output (y, x) is patch row (y + ky) * 34 + x + kx: rows never wrap, so the fragment masking of the strip
// kernel disappears.  Loaders, weight ring and barriers follow conv_halo_ws_kernel, but persistently: one workgroup per
// CU walks a list of tiles, the patch of the next tile's first chunk is simply "the next chunk" of the double buffer
// and the weight ring runs straight into the next tile; after a tile's K loop one extra barrier ("everyone is done
// with the last chunk's buffer"), then each MFMA wave parks 16 pixels x 64 channels at a time in a private 4 KiB
// slice of that idle buffer and writes them out as 16-byte NHWC chunks.  Cost: tiles that hang over the right / bottom edge compute
// pixels nobody stores (152 = 4.75 x 32: 5 %), which is why narrow maps stay on the strip kernels.
template <typename T, int NSB>
__global__ __launch_bounds__(768, 3) void conv_patch_wsp_kernel(HaloArgs p, int n_tiles_total, int tiles_x, int tiles_y) {
  constexpr int BN = 128;
  constexpr int TY = 8, TX = 32;                      // output tile: 8 rows x 32 columns
  constexpr int PC = TX + 2, PROWS = (TY + 2) * PC;   // input patch: 10 x 34 pixels, one LDS row each
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
    // patch slice `pass` (32 LDS rows = 32 patch pixels) of chunk `chunk` of the tile at (b, oy0, ox0), into buffer `buf`
    auto issue_halo_pass = [&](int b, int oy0, int ox0, int chunk, int pass, int buf, bool live) {
      const int r = row0 + pass * RPL;
      const int pr = r / PC, pc = r - pr * PC;
      const int gy = oy0 - 1 + pr, gx = ox0 - 1 + pc;
      const bool ok = live && r < PROWS && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      const long long q = ((long long)b * p.H + gy) * p.W + gx;
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
    // tile id -> (frame, tile row, tile column, channel tile); channel tiles innermost (they share the patch in L2)
    auto tile_pos = [&](int tile, int &b, int &oy0, int &ox0) {
      int t = tile / p.n_tiles;
      ox0 = (t % tiles_x) * TX;
      t /= tiles_x;
      oy0 = (t % tiles_y) * TY;
      b = t / tiles_y;
    };
    auto tile_n0 = [&](int tile) { return (tile % p.n_tiles) * BN; };
    // prologue of the first tile only
    {
      int b0, oy00, ox00;
      tile_pos(tile0, b0, oy00, ox00);
      for (int pass = 0; pass < p.na; ++pass) issue_halo_pass(b0, oy00, ox00, 0, pass, 0, true);
#pragma unroll
      for (int j = 0; j <= D; ++j) issue_weights(tile_n0(tile0), j, j);
    }
    int ring = (D + 1) % NSB;                          // slot that receives the next weight tile
    int gchunk = 0;                                    // chunks consumed so far (halo buffer = gchunk & 1)
    int gstep = 0;                                     // K-steps issued so far, all tiles
    for (int tile = tile0; tile < n_tiles_total; tile += grid) {
      const int next_tile = tile + grid;
      const bool has_next = next_tile < n_tiles_total;
      int tb, toy, tox, nb, noy, nox;
      tile_pos(tile, tb, toy, tox);
      tile_pos(has_next ? next_tile : tile, nb, noy, nox);
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
        issue_halo_pass(in_tile ? tb : nb, in_tile ? toy : noy, in_tile ? tox : nox, in_tile ? chunk + 1 : 0, p0, nbuf, live);
        issue_halo_pass(in_tile ? tb : nb, in_tile ? toy : noy, in_tile ? tox : nox, in_tile ? chunk + 1 : 0, p1, nbuf, live);
        ring = ring + 1 == NSB ? 0 : ring + 1;
        if (++tap == 9) { tap = 0; ++chunk; ++gchunk; }
      }
      __builtin_amdgcn_s_barrier();                    // E: consumers are done with the last chunk's halo buffer
    }
    wait_vmcnt<0>();
    return;
  }

  // ---------------- consumer waves ----------------
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);   // the younger MFMA wave of each SIMD: static priority, +1-5 % measured
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int fr = lane & 15, fq = lane >> 4;
  const int a_lane_row = 2 * wm * PC + fr;            // patch row of (tile row 2*wm, column fr) at tap (0, 0)
  const int b_lane_row = wn * 64 + fr;
  const int b_off0 = b_lane_row * 128 + (((0 + fq) ^ (b_lane_row & 7)) << 4);
  const int b_off1 = b_lane_row * 128 + (((4 + fq) ^ (b_lane_row & 7)) << 4);
  const bool leaky = p.flags & Y3_F_LEAKY;
  const bool has_res = p.flags & Y3_F_RESIDUAL;
  int ring = 0, gchunk = 0;
  for (int tile = tile0; tile < n_tiles_total; tile += grid) {
    int tb, oy0, ox0;
    {
      int t = tile / p.n_tiles;
      ox0 = (t % tiles_x) * TX;
      t /= tiles_x;
      oy0 = (t % tiles_y) * TY;
      tb = t / tiles_y;
    }
    const int n0 = (tile % p.n_tiles) * BN;
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment mi of this wave: tile row 2*wm + (mi >> 1), columns (mi & 1) * 16 + fr; tap (ky, kx) shifts the patch
    // row by ky * PC + kx.  No masking anywhere: out-of-frame patch pixels were loaded as zeros, rows do not wrap.
    auto read_frags = [&](u32x4 (&xf)[MI], u32x4 (&wf)[NI], const char *aBuf, const char *bBuf, int a_shift, int g) {
      const int r0 = a_lane_row + a_shift, r1 = r0 + PC;
      const char *ap0 = aBuf + r0 * 128 + (((g * 4 + fq) ^ (r0 & 7)) << 4);
      const char *ap1 = aBuf + r1 * 128 + (((g * 4 + fq) ^ (r1 & 7)) << 4);
      const char *bp = bBuf + (g ? b_off1 : b_off0);
      xf[0] = *reinterpret_cast<const u32x4 *>(ap0);
      xf[1] = *reinterpret_cast<const u32x4 *>(ap0 + 2048);
      xf[2] = *reinterpret_cast<const u32x4 *>(ap1);
      xf[3] = *reinterpret_cast<const u32x4 *>(ap1 + 2048);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[ni] = *reinterpret_cast<const u32x4 *>(bp + ni * 2048);
    };
    auto mma_all = [&](u32x4 (&xf)[MI], const u32x4 (&wf)[NI], int tap) {
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
      read_frags(xf1, wf1, aBuf, sB + ring * B_BYTES, ky * PC + kx, 1);
      mma_all(xf0, wf0, tap);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      const int tap_n = tap == 8 ? 0 : tap + 1;
      const int gchunk_n = tap == 8 ? gchunk + 1 : gchunk;
      const int ring_n = ring + 1 == NSB ? 0 : ring + 1;
      {
        const int ky_n = (tap_n * 11) >> 5, kx_n = tap_n - ky_n * 3;
        read_frags(xf0, wf0, sA + (gchunk_n & 1) * p.a_bytes, sB + ring_n * B_BYTES, ky_n * PC + kx_n, 0);
      }
      mma_all(xf1, wf1, tap);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      // (no s_waitcnt before the next barrier: see conv_halo_ws_kernel)
      tap = tap_n;
      gchunk = gchunk_n;
      ring = ring_n;
    }
    __builtin_amdgcn_s_barrier();                      // E: every consumer is done reading the last chunk's halo
#ifdef Y3_X_NOEPI
    // timing-only experiment (y3_set_tuning("debug", 1) in this diagnostic build): no epilogue at all (results wrong)
    if (p.flags & 0x40000000u) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(acc[mi][ni]));
      continue;
    }
#endif
    // ---- epilogue, per wave: 16 pixels x 64 channels at a time through a private 4 KiB slice of that buffer ----
    float *sC = reinterpret_cast<float *>(sA + ((gchunk + 1) & 1) * p.a_bytes) + wave * 1024;
    const int oc = lane & 7;                           // 8-channel group of this lane's write-out items
    const int co = n0 + wn * 64 + oc * 8;
    const f32x4 sc_lo = *reinterpret_cast<const f32x4 *>(p.scale + co);
    const f32x4 sc_hi = *reinterpret_cast<const f32x4 *>(p.scale + co + 4);
    const f32x4 bi_lo = *reinterpret_cast<const f32x4 *>(p.bias + co);
    const f32x4 bi_hi = *reinterpret_cast<const f32x4 *>(p.bias + co + 4);
    // shortcut operand: fetched ONE 16-pixel group ahead of its use (two register sets), so that a group's memory
    // latency runs under the previous group's LDS round trip, arithmetic and stores instead of being waited out four
    // times per tile (a tile here has only nine K-steps: the epilogue is a third of its time)
    u32x4 resv[2][2];
    f32x4 resf[2][2][2];
    auto res_load = [&](int mi, int set) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int oy = oy0 + 2 * wm + (mi >> 1), ox = ox0 + (mi & 1) * 16 + (lane >> 3) + r * 8;
        const long long m = ((long long)tb * p.H + oy) * p.W + ox;
        if (has_res && oy < p.H && ox < p.W) {
          if constexpr (sizeof(T) == 2) {
            resv[set][r] = *reinterpret_cast<const u32x4 *>(p.res + ((long long)m * p.res_ld + co) * ES);
          } else {
            const float *rp = reinterpret_cast<const float *>(p.res) + (long long)m * p.res_ld + co;
            resf[set][r][0] = *reinterpret_cast<const f32x4 *>(rp);
            resf[set][r][1] = *reinterpret_cast<const f32x4 *>(rp + 4);
          }
        }
      }
    };
    constexpr bool AHEAD = sizeof(T) == 2;             // (float32: the second register set spills at the 168-VGPR cap)
    if constexpr (AHEAD) res_load(0, 0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (AHEAD) {
        if (mi + 1 < MI) res_load(mi + 1, (mi + 1) & 1);
      } else {
        res_load(mi, 0);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)                  // pixel fr, channels ni*16 + fq*4 .. +3; 16-B chunks XOR-swizzled
        *reinterpret_cast<f32x4 *>(sC + fr * 64 + (((ni * 4 + fq) ^ fr) << 2)) = acc[mi][ni];
      __builtin_amdgcn_s_waitcnt(0xC07F);              // (same wave reads below: LDS ops of a wave complete in order)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int pl = (lane >> 3) + r * 8;            // pixel inside the 16-row group
        const int oy = oy0 + 2 * wm + (mi >> 1), ox = ox0 + (mi & 1) * 16 + pl;
        const long long m = ((long long)tb * p.H + oy) * p.W + ox;
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(sC + pl * 64 + (((2 * oc) ^ pl) << 2));
        const f32x4 hi = *reinterpret_cast<const f32x4 *>(sC + pl * 64 + (((2 * oc + 1) ^ pl) << 2));
        float v[8];
        y3_bn_leaky8(v, lo, hi, sc_lo, sc_hi, bi_lo, bi_hi, leaky);
        if (oy < p.H && ox < p.W) {
          if (has_res) {
            if constexpr (sizeof(T) == 2) {
              y3_add8<T>(v, resv[AHEAD ? (mi & 1) : 0][r]);
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) { v[q] += resf[AHEAD ? (mi & 1) : 0][r][0][q]; v[4 + q] += resf[AHEAD ? (mi & 1) : 0][r][1][q]; }
            }
          }
          T *op = reinterpret_cast<T *>(p.out) + (long long)m * p.out_ld + co;
          if constexpr (sizeof(T) == 2) {
            *reinterpret_cast<u32x4 *>(op) = y3_pack8<T>(v);
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

// ------------------------------------------------------------------------------------------------
// Direct-weights strip kernel (round 5; 16-bit storage modes): a workgroup owns 192 raster pixels x 256 output channels.
// Eight waves, two per SIMD, no roles: each computes 96 pixels x 64 channels (96 accumulator registers), reads its pixel
// fragments from the halo image in LDS exactly like conv_halo_ws_kernel (tap = row offset, border taps redirected to the
// zero row) and takes its WEIGHT fragments straight from global memory into registers, out of a FRAGMENT-ORDER copy of the
// weights (the plan makes it once: 1-KiB blocks of 16 channels x 32 K-elements in the MFMA operand layout, so that one
// coalesced global_load_dwordx4 per fragment delivers the operand as it is).  No weight ring, no per-step barrier (one
// barrier per channel chunk, when the halo buffers swap), half the LDS fragment reads per MFMA of the 64 x 64 wave tile,
// and the halo is staged once for 256 channels instead of once per 128.  Per K-step (64 channels of one tap) a wave issues
// 48 MFMAs, 12 ds_read_b128, 8 global_load_dwordx4 (one K-step ahead, three register sets of four fragments in rotation)
// and at most one 1-KiB LDS-DMA piece of the next chunk's halo.  Same K order (chunk outermost, tap innermost, K-halves
// in order) as every other MFMA conv kernel here: same bits.  Development, measurements and the one hard bug
// (inline-asm loads the compiler cannot see in flight): profiles/r05s_halo_dw.txt.
// dw_wait_vm: s_waitcnt vmcnt(N) that NAMES the four registers it waits for -- the tie is what keeps the compiler from
// moving their uses above the wait, and from re-using them while the load is in flight.
// Y3_DW_EPI: the kernel's write-out.  0 = two channel halves of 128, each parked in LDS as float32 and written out by all
// threads (round 5).  1 = straight from the accumulators: the fragment-order copy of the weights carries y3_pair_perm'd rows,
// so a lane holds eight consecutive channels of its pixel per fragment pair -- scale / bias / LeakyReLU / shortcut / rounding
// in registers, one 16-byte store per pair; no LDS, no barrier, waves finish independently.  2 = the same arithmetic in
// registers, the ROUNDED tile (192 x 256 x 2 bytes) parked in LDS once and written out as whole 512-byte pixel rows.
// Same arithmetic per value in all three: same bits.
#ifndef Y3_DW_EPI
#define Y3_DW_EPI 0
#endif
// Y3_DW_SMASK: 1 = border-tap masks as wave-wide lane masks in scalar registers (one v_cndmask per select, as in
// conv_halo_ws_kernel); 0 = nine tap bits per fragment in vector registers (round 5)
#ifndef Y3_DW_SMASK
#define Y3_DW_SMASK 1
#endif

// Tried in round 6 and removed (profiles/r06_dw_epilogue.txt): THREE halo buffers with the images of chunks 0, 1 and 2 all issued in
// the prologue and chunk c + 2 fetched during chunk c (EXEC-masked inline-asm LDS-DMA, so that a 128-channel layer issues none
// from inside its K loop and a 256-channel layer one chunk's worth instead of three).  -8 % at 38^2, -12 % at 76^2, -5 % end to
// end: all 256 CUs start together and the prologue's burst (144 instead of 48 KiB per CU) is memory-bound; the loads of the
// second K-step cannot complete before it (vmcnt retires in order).  Spread over the K loop the same bytes cost less.
template <int N, typename V>
__device__ __forceinline__ void dw_wait_vm(V (&w)[4]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "n"(N) : "memory");
}

template <typename T, int NA>
__global__ __launch_bounds__(512, 2) void conv_halo_dw_kernel(HaloArgs p) {
  static_assert(sizeof(T) == 2 && NA >= 4 && NA <= 6, "16-bit storage modes only; 4 .. 6 halo passes per chunk");
  constexpr int MI = 6, NI = 4;                       // wave tile: 6 x 16 pixels, 4 x 16 channels (8 x 16 pixels needs 256+ VGPRs: spills)
  constexpr int WM = MI * 16, MH = MI / 2;
  constexpr int BM = 2 * WM, BN = 256;
  constexpr int NT = 512;
  constexpr int ES = 2, BKE = 64;
  constexpr int RPL = NT / 8;                         // halo rows filled per pass of the workgroup (64)

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *sA = smem;                                    // [2][hr_pad][128]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = y3_xcd_remap(blockIdx.x, gridDim.x);
  const int grp = tile / (p.m_tiles * p.ngrp_w), rem = tile - grp * (p.m_tiles * p.ngrp_w);
  const int m0 = (rem / p.ngrp_w) * BM;
  const int n0 = (grp * p.ngrp_w + rem % p.ngrp_w) * BN;
  const int wm = wave >> 2, wn = wave & 3;            // waves w and w + 4 share a SIMD and a channel slice
  const int fr = lane & 15, fq = lane >> 4;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- halo image: every thread fills its 16 bytes of each 64-row pass (see conv_halo_ws_kernel's loader) ----
  const int slot = tid & 7;
  const int row0 = tid >> 3;
  const int kc = slot ^ (row0 & 7);
  const long long qr = (long long)m0 - p.W - 1 + row0;
  const int hi_rows = (p.hr - 1 - row0) >> 6;
  const long long hi_px = qr < p.M ? (p.M - 1 - qr) >> 6 : -1;
  const int pass_lo = qr >= 0 ? 0 : (int)((-qr + 63) >> 6);
  const int pass_hi = hi_px < hi_rows ? (int)hi_px : hi_rows;
  const char *hsrc0 = p.in + (qr * p.in_ld) * ES + kc * 16;
  const long long pass_step = (long long)RPL * p.in_ld * ES;
  auto issue_halo_pass = [&](int chunk, int pass, bool live) {
    const bool ok = live && pass >= pass_lo && pass <= pass_hi;
    const char *src = ok ? hsrc0 + (pass * pass_step + (long long)chunk * (BKE * ES)) : p.zero;
    char *dst = sA + (chunk & 1) * p.a_bytes + pass * (NT * 16) + wave * 1024;
    __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)dst, 16, 0, Y3_AUX_H);
  };

  // ---- weight fragments, from the FRAGMENT-ORDER copy of the weights (y3_conv_halo_dw_layout): the 1 KiB that the 64 lanes
  // of a wave need for 16 channels x 32 K-elements is contiguous, lane l's 16 bytes at l * 16 -- a fully coalesced load (the
  // [Cout][K] layout makes every four lanes touch four different rows: measured 1.5x slower end of K-step, r05s).  Block
  // (channel block cb, K block kb of 64 bytes) sits at ((cb * KB + kb) << 10), KB = k_ld / 32.
  const uint32_t kblocks = (uint32_t)p.k_ld / 32u;
  const uint32_t b_voff = (uint32_t)lane * 16;
  const char *b_base[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) b_base[ni] = p.wgt + (((long long)((n0 + wn * 64) / 16 + ni) * kblocks) << 10);
  auto load_w0 = [&](u32x4 (&w)[NI], uint32_t voff) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(w[ni]) : "v"(voff), "s"(b_base[ni]));
  };
  auto load_w1 = [&](u32x4 (&w)[NI], uint32_t voff) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(w[ni]) : "v"(voff), "s"(b_base[ni]));
  };

  // ---- pixel fragments ----
  // border-tap masks of the wave's six fragments: lane masks in scalar registers (Y3_DW_SMASK) or nine tap bits per fragment,
  // three fragments per vector register
  [[maybe_unused]] uint32_t tm[(MI + 2) / 3] = {};
  unsigned long long mk_top[MI], mk_bot[MI], mk_left[MI], mk_right[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const uint32_t m = (uint32_t)(m0 + wm * WM + mi * 16 + fr);
    const uint32_t img = (__umulhi(m, p.mul_hw) + m) >> p.sh_hw;
    const uint32_t r = m - img * (uint32_t)p.HW;
    const uint32_t oy = (__umulhi(r, p.mul_w) + r) >> p.sh_w;
    const uint32_t ox = r - oy * (uint32_t)p.W;
#if Y3_DW_SMASK
    mk_top[mi] = __builtin_amdgcn_ballot_w64(oy >= 1u);
    mk_bot[mi] = __builtin_amdgcn_ballot_w64(oy + 1u < (uint32_t)p.H);
    mk_left[mi] = __builtin_amdgcn_ballot_w64(ox >= 1u);
    mk_right[mi] = __builtin_amdgcn_ballot_w64(ox + 1u < (uint32_t)p.W);
#else
    const uint32_t vx = (ox >= 1u ? 1u : 0u) | 2u | (ox + 1u < (uint32_t)p.W ? 4u : 0u);
    const uint32_t t9 = (oy >= 1u ? vx : 0u) | (vx << 3) | (oy + 1u < (uint32_t)p.H ? vx << 6 : 0u);
    tm[mi / 3] |= t9 << (9 * (mi % 3));
#endif
  }
  typedef const __attribute__((address_space(3))) u32x4 lds_u32x4;
  const int sA_lds = (int)(size_t)(lds_void *)sA;
  const int zbase = sA_lds + p.hr * 128;              // rows hr, hr + 1 of buffer 0: always zero
  const int a_lane_row = wm * WM + fr;
  int sel[MI];
  // K-half 0 of tap `tap`: fragment addresses (border taps redirected to the zero row) for all eight fragments, then the
  // reads of fragments [lo, lo + 4)
  auto frag_addrs = [&](auto tapc, int a_off) {
    constexpr int tap = decltype(tapc)::value, ky = tap / 3, kx = tap % 3;
    const int r0 = a_lane_row + ky * p.W + kx;
    const int ap = ((r0 << 7) + (a_off + sA_lds)) + ((fq ^ (r0 & 7)) << 4);
    const int zoff = (ap & 255) | zbase;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      int off = ap + mi * 2048;
      if constexpr (tap != 4) {
#if Y3_DW_SMASK
        unsigned long long ok = ky == 0 ? mk_top[mi] : (ky == 2 ? mk_bot[mi] : ~0ull);
        if constexpr (kx == 0) ok &= mk_left[mi];
        if constexpr (kx == 2) ok &= mk_right[mi];
        const int real = off;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(off) : "v"(zoff), "v"(real), "s"(ok));
#else
        off = ((tm[mi / 3] >> (9 * (mi % 3) + tap)) & 1u) ? off : zoff;
#endif
      }
      sel[mi] = off;
    }
  };
  auto read_half = [&](u32x4 (&xf)[MI], int lo, int flip) {
#pragma unroll
    for (int mi = 0; mi < MH; ++mi) xf[lo + mi] = *reinterpret_cast<lds_u32x4 *>(sel[lo + mi] ^ flip);
  };
  auto mma_half = [&](const u32x4 (&xf)[MI], int lo, const u32x4 (&wf)[NI]) {
#pragma unroll
    for (int mi = 0; mi < MH; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[lo + mi][ni] = y3_mfma16<T>(wf[ni], xf[lo + mi], acc[lo + mi][ni]);
  };

  // ---- prologue: halo of chunk 0, both K-halves of step 0's weights, step 0's first pixel fragments ----
  u32x4 wf[3][NI];
  u32x4 xf[MI];
  for (int pass = 0; pass < p.na; ++pass) issue_halo_pass(0, pass, true);
  load_w0(wf[0], b_voff);
  load_w1(wf[2], b_voff);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);       // the younger wave of each SIMD (see conv_halo_ws_kernel)
  frag_addrs(TapC<0>{}, 0);
  read_half(xf, 0, 0);
  read_half(xf, MH, 0);

  // One K-step.  Weight register sets: the step with index S = tap % 3 multiplies K-half 0 with wf[S] and K-half 1 with
  // wf[(S + 2) % 3]; the next step's K-half 0 is loaded into wf[(S + 1) % 3] at the top, its K-half 1 into wf[S] once this
  // step's K-half 0 MFMAs are issued.  VMEM order per step: 4 loads, at most 1 halo piece, 4 loads -- the counted waits rely on it.
  // Pixel fragments: eight registers sets of one fragment; a half (four fragments) is re-read for the next K-half as soon as
  // the MFMAs that use it are issued.  (A second register set with every read a whole K-half ahead measured 2 % slower: the
  // SIMD's other wave covers the read latency already, r05s.)
  auto kstep = [&](auto tapc, int chunk) {
    constexpr int tap = decltype(tapc)::value, S = tap % 3, tap_n = tap == 8 ? 0 : tap + 1;
    const int chunk_n = tap == 8 ? chunk + 1 : chunk;
    const uint32_t voff_n = b_voff + ((uint32_t)(tap_n * (p.Cin / 32) + chunk_n * 2) << 10);
    load_w0(wf[(S + 1) % 3], voff_n);
    // one 64-row pass of the next chunk's halo per step while there are any (4-6 of the 9 steps; a piece costs the issuing
    // wave 60-185 cycles: without them the launch is 10-15 % shorter, so none is issued that is not needed)
    // (NA = the number of passes, a template parameter: 4 .. 6 covers every map the wave-specialised kernel takes, rows of up
    // to 94 pixels; as a run-time test the branch costs registers the kernel does not have -- 96 bytes of scratch, 15 % slower)
    if constexpr (tap < NA) issue_halo_pass(chunk + 1, tap, chunk + 1 < p.nchunks);
    // younger than wf[S]'s loads: 4 loads of the previous step + 4 of this one for certain, up to two halo pieces maybe --
    // the count that is always safe is 8
    dw_wait_vm<8>(wf[S]);
    mma_half(xf, 0, wf[S]);
    read_half(xf, 0, 64);
    mma_half(xf, MH, wf[S]);
    read_half(xf, MH, 64);
    __builtin_amdgcn_sched_barrier(0);
    load_w1(wf[S], voff_n);
    dw_wait_vm<8>(wf[(S + 2) % 3]);                   // younger for certain: 4 + 4 loads of this step
    mma_half(xf, 0, wf[(S + 2) % 3]);
    if constexpr (tap == 8) {
      // chunk boundary: the next step reads the other halo buffer.  Its pieces (the youngest was issued before the last
      // four loads) have landed for this wave ... and, after the barrier, for all; the barrier also tells that everyone
      // is done with the buffer that the next chunk's pieces will overwrite -- EXCEPT the second half of this K-half's
      // fragments, which are in registers already
      mma_half(xf, MH, wf[(S + 2) % 3]);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      frag_addrs(TapC<tap_n>{}, (chunk_n & 1) * p.a_bytes);
      read_half(xf, 0, 0);
      read_half(xf, MH, 0);
    } else {
      frag_addrs(TapC<tap_n>{}, (chunk_n & 1) * p.a_bytes);
      read_half(xf, 0, 0);
        mma_half(xf, MH, wf[(S + 2) % 3]);
        read_half(xf, MH, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
#pragma unroll 1
  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    kstep(TapC<0>{}, chunk); kstep(TapC<1>{}, chunk); kstep(TapC<2>{}, chunk);
    kstep(TapC<3>{}, chunk); kstep(TapC<4>{}, chunk); kstep(TapC<5>{}, chunk);
    kstep(TapC<6>{}, chunk); kstep(TapC<7>{}, chunk); kstep(TapC<8>{}, chunk);
  }
  __builtin_amdgcn_s_setprio(0);
  // The run-ahead loads of the step after the last are still in flight and nobody uses their data: without the register ties
  // below the compiler considers their destination registers free from here on and may place the epilogue's pointer arithmetic
  // in them BEFORE the wait -- a load that lands late (memory contention: another kernel beside this one) then overwrites an
  // address, and the epilogue reads from nowhere (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION; profiles/r05s_halo_dw.txt).
  dw_wait_vm<0>(wf[0]);
  dw_wait_vm<0>(wf[1]);
  dw_wait_vm<0>(wf[2]);

  const bool leaky = p.flags & Y3_F_LEAKY;
  const bool has_res = p.flags & Y3_F_RESIDUAL;
#if Y3_DW_EPI != 0
  // ---- epilogue in registers (Y3_DW_EPI 1 / 2).  The weight rows were laid out with y3_pair_perm: after both fragments of a
  // pair (ni = 2 pr, 2 pr + 1) lane (fr, fq) holds channels  n0 + wn * 64 + pr * 32 + 8 fq .. + 7  of pixel  m0 + wm * 96 +
  // mi * 16 + fr.  All twelve shortcut loads of the wave (16 bytes each, the registers of the weight / pixel fragments are
  // free now) go out first, then per pair: scale / bias, LeakyReLU, + shortcut, round, store.
  {
    constexpr int NP = NI / 2;
    const int co0 = n0 + wn * 64 + fq * 8;
    const int mrow = m0 + wm * WM + fr;
    u32x4 resv[NP][MI];
    if (has_res) {
#pragma unroll
      for (int pr = 0; pr < NP; ++pr)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int m = mrow + mi * 16;
          const char *rp = p.res + ((long long)m * p.res_ld + co0 + pr * 32) * ES;
          resv[pr][mi] = m < p.M ? *reinterpret_cast<const u32x4 *>(rp) : u32x4{0u, 0u, 0u, 0u};
        }
    }
#if Y3_DW_EPI == 2
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                     // nobody reads the halo any more: LDS holds the rounded output tile
    char *sO = smem;                                  // [192 pixels][512 bytes], 16-byte pieces XOR-swizzled by pixel
#endif
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
      const int co = co0 + pr * 32;
      const f32x4 sc_lo = *reinterpret_cast<const f32x4 *>(p.scale + co), sc_hi = *reinterpret_cast<const f32x4 *>(p.scale + co + 4);
      const f32x4 bi_lo = *reinterpret_cast<const f32x4 *>(p.bias + co), bi_hi = *reinterpret_cast<const f32x4 *>(p.bias + co + 4);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int m = mrow + mi * 16;
        float v[8];
        y3_bn_leaky8(v, acc[mi][2 * pr], acc[mi][2 * pr + 1], sc_lo, sc_hi, bi_lo, bi_hi, leaky);
        if (has_res) y3_add8<T>(v, resv[pr][mi]);
        const u32x4 o = y3_pack8<T>(v);
#if Y3_DW_EPI == 1
        if (m < p.M) *reinterpret_cast<u32x4 *>(reinterpret_cast<T *>(p.out) + (long long)m * p.out_ld + co) = o;
#else
        const int pl = wm * WM + mi * 16 + fr;        // piece index inside the pixel's 512 bytes: (channel - n0) / 8
        *reinterpret_cast<u32x4 *>(sO + pl * 512 + ((((co - n0) >> 3) ^ (pl & 31)) << 4)) = o;
#endif
      }
    }
#if Y3_DW_EPI == 2
    __syncthreads();
    // 512 threads x 16 bytes = 16 pixel rows of 512 bytes per pass: every wave writes 2 KiB of consecutive output bytes
    const int piece = tid & 31;
#pragma unroll
    for (int j = 0; j < BM / 16; ++j) {
      const int pl = (tid >> 5) + j * 16;
      const int m = m0 + pl;
      const u32x4 o = *reinterpret_cast<const u32x4 *>(sO + pl * 512 + ((piece ^ (pl & 31)) << 4));
      if (m < p.M) *reinterpret_cast<u32x4 *>(reinterpret_cast<T *>(p.out) + (long long)m * p.out_ld + n0 + piece * 8) = o;
    }
#endif
  }
#else
  // ---- epilogue: two channel halves of 128; the four waves that own a half park it, all 512 threads write it out.  A half's
  // scale / bias / shortcut reads are issued one stage ahead: the first half's before the barrier that ends the K loop, the
  // second half's before the first half's write-out (all at once, before the loop's end, they cost 48 registers: spills) ----
  constexpr int SWZ = 15;
  constexpr int RPP = NT / 16;                        // 32 pixel rows per pass
  constexpr int WR = BM / RPP;                        // 6
  float *sC = reinterpret_cast<float *>(smem);
  const int oc_mine = tid & 15;
  f32x4 sc_lo[2], sc_hi[2], bi_lo[2], bi_hi[2];
  u32x4 resv[2][WR];
  auto epi_loads = [&](int h) {
    const int co = n0 + h * 128 + oc_mine * 8;
    sc_lo[h] = *reinterpret_cast<const f32x4 *>(p.scale + co);
    sc_hi[h] = *reinterpret_cast<const f32x4 *>(p.scale + co + 4);
    bi_lo[h] = *reinterpret_cast<const f32x4 *>(p.bias + co);
    bi_hi[h] = *reinterpret_cast<const f32x4 *>(p.bias + co + 4);
    if (has_res) {
#pragma unroll
      for (int j = 0; j < WR; ++j) {
        const int m = m0 + (tid >> 4) + j * RPP;
        const char *rp = p.res + ((long long)m * p.res_ld + co) * ES;
        resv[h][j] = m < p.M ? *reinterpret_cast<const u32x4 *>(rp) : u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  epi_loads(0);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();                       // nobody reads the halo any more: LDS holds the output tile
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if ((wn >> 1) == h) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int cl = (wn & 1) * 64 + (ni >> 1) * 32 + fq * 8 + (ni & 1) * 4;   // y3_pair_perm'd weight rows
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int pl = wm * WM + mi * 16 + fr;
          *reinterpret_cast<f32x4 *>(sC + pl * 128 + (((cl >> 2) ^ (pl & SWZ)) << 2)) = acc[mi][ni];
        }
      }
    }
    if (h == 0) epi_loads(1);
    const int co = n0 + h * 128 + oc_mine * 8;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < WR; ++j) {
      const int pl = (tid >> 4) + j * RPP;
      const int m = m0 + pl;
      if (m >= p.M) continue;
      const f32x4 lo = *reinterpret_cast<const f32x4 *>(sC + pl * 128 + (((2 * oc_mine) ^ (pl & SWZ)) << 2));
      const f32x4 hi = *reinterpret_cast<const f32x4 *>(sC + pl * 128 + (((2 * oc_mine + 1) ^ (pl & SWZ)) << 2));
      float v[8];
      y3_bn_leaky8(v, lo, hi, sc_lo[h], sc_hi[h], bi_lo[h], bi_hi[h], leaky);
      if (has_res) y3_add8<T>(v, resv[h][j]);
      T *op = reinterpret_cast<T *>(p.out) + (long long)m * p.out_ld + co;
      *reinterpret_cast<u32x4 *>(op) = y3_pack8<T>(v);
    }
    if (h == 0) __syncthreads();                      // the first half is out of LDS before the second is parked
  }
#endif
}

// n / d == (umulhi(n, mul) + n) >> sh for 0 <= n < 2^31 (round-up method, d >= 1)
void fast_div(uint32_t d, uint32_t &mul, uint32_t &sh) {
  if (d <= 1) { mul = 0; sh = 0; return; }
  sh = 0;
  while ((1u << sh) < d) ++sh;
  mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << sh) - d)) / d + 1);
}

// 192-pixel tiles (MI = 3) when they finish sooner than 256-pixel tiles: rounds of workgroups (one per CU) x per-tile
// time (K-steps x MFMA cycles of the step + ~300 cycles of barrier / LDS bubbles, + ~11 k cycles outside the loop:
// profiles/r01_halo_kernel_anatomy.txt).  At batch 16: 19^2 184 tiles -> 248 (one round either way, 3/4 of the work per
// tile), 38^2 364 tiles in two rounds -> 484 in two rounds of 3/4 tiles, 76^2 stays at 722 x 256 pixels.  Measured
// (profiles/r02d_*): 880 -> 1000 TF at 38^2, 960 -> 1090 TF at 19^2 per launch, the forward of one batch 3 % shorter.
// This is the LATENCY choice (one batch at a time: inference(), the video loop).  With several batches in flight on
// their own streams (bench.py) the idle CUs of a short last round are filled by the other batches' kernels anyway and
// what counts is CU time, which smaller tiles raise (the ~11 k cycles per tile are paid 248 instead of 184 times):
// 2.3 % fewer frames/s end to end.  Such callers set Y3_AM_HALO_TILE256 (256-pixel tiles only) in their plan options.
static int halo_tile_fragments(int M, int n_tiles, int nchunks, int n_cu) {
  if ((unsigned)y3_opt().auto_mask & Y3_AM_HALO_TILE256) return 4;
  double best = 0;
  int best_mi = 4;
  for (int mi = 4; mi >= 3; --mi) {
    const long long tiles = (long long)y3_ceil_div(M, 64 * mi) * n_tiles;
    const double rounds = (double)((tiles + n_cu - 1) / n_cu);
    const double cost = rounds * (nchunks * 9.0 * (256.0 * mi + 300.0) + 11000.0);
    if (best == 0 || cost < 0.97 * best) { best = cost; best_mi = mi; }
  }
  return best_mi;
}

template <typename T>
int launch_halo_ws(const HaloArgs &a0, hipStream_t s) {
  HaloArgs a = a0;
  static Y3DeviceOnce once;
  int n_cu = 256;
  {
    const int rc = once.run([]() -> int {
      Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_halo_ws_kernel<T, 3, 4>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_halo_ws_kernel<T, 4, 4>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_halo_ws_kernel<T, 3, 3>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_halo_ws_kernel<T, 4, 3>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      return Y3_OK;
    }, &n_cu);
    if (rc != Y3_OK) return rc;
  }
  const int mi = halo_tile_fragments(a.M, a.n_tiles, a.nchunks, n_cu);
  const int bm = 64 * mi;
  const int hr = bm + 2 * a.W + 2;
  a.hr = hr;
  a.na = y3_ceil_div(hr + 2, 32);                     // + the two zero rows (hr is even: rows hr, hr + 1 = one 256-byte bank row)
  a.hr_pad = a.na * 32;
  a.a_bytes = a.hr_pad * 128;
  // the next chunk's halo goes out two passes per K-step and must be older than the last loads allowed in flight
  int nsb = 0;
  size_t lds = 0;
  for (int c = 4; c >= 3; --c) {
    if (a.na > (c == 4 ? 12 : 14)) continue;   // all real halo slices out by tap 5 (4 slots) / 6 (3 slots)
    lds = (size_t)c * 128 * 128 + (size_t)2 * a.a_bytes;
    if (lds < (size_t)bm * 128 * 4) lds = (size_t)bm * 128 * 4;
    if (lds <= 160 * 1024) { nsb = c; break; }
  }
  Y3_REQUIRE(nsb != 0, "wave-specialised halo kernel: row width %d does not fit", a.W);
  a.m_tiles = y3_ceil_div(a.M, bm);
  // Tile order and HBM traffic.  Each XCD (own L2) gets one contiguous run of tile ids.  With all channel tiles innermost
  // (ngrp_w = n_tiles) a run covers a few pixel tiles x ALL weight panels: every XCD fetches the whole weight matrix -- fine
  // while the input outweighs it (76^2, 38^2), 75 MB of extra fetches per launch at 19^2 (9.4 MB of weights, 5.9 MB of
  // input).  With channel tiles in pn groups the XCDs split into pn sets of 8 / pn, each set owning one group's panels and
  // all pixels: extra fetches ~ pn x input + (8 / pn) x weights; pick the pn (1, 2, 4, 8 dividing n_tiles) that minimises
  // it.  Placement only: results and per-tile time do not change.
  {
    const double in_b = (double)a.M * a.Cin, w_b = 9.0 * a.Cin * a.Cout;
    int best = 1;
    double best_cost = 0;
    for (int pn = 1; pn <= 8; pn <<= 1) {
      if (a.n_tiles % pn) break;
      const double cost = pn * in_b + (8.0 / pn) * w_b;
      if (pn == 1 || cost < 0.9 * best_cost) { best = pn; best_cost = cost; }
    }
    a.ngrp_w = a.n_tiles / best;
  }
  const dim3 grid(a.m_tiles * a.n_tiles);
  if (mi == 3) {
    if (nsb == 4) Y3_LAUNCH((conv_halo_ws_kernel<T, 4, 3>), grid, dim3(768), lds, s, a);
    else Y3_LAUNCH((conv_halo_ws_kernel<T, 3, 3>), grid, dim3(768), lds, s, a);
  } else {
    if (nsb == 4) Y3_LAUNCH((conv_halo_ws_kernel<T, 4, 4>), grid, dim3(768), lds, s, a);
    else Y3_LAUNCH((conv_halo_ws_kernel<T, 3, 4>), grid, dim3(768), lds, s, a);
  }
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}

// [Cout_pad][k_ld] 16-bit weights -> fragment order: 1-KiB blocks of 16 channels x 32 K-elements, lane l = fq * 16 + fr of a
// wave holds channel fr, K-elements [8 fq, 8 fq + 8) of the block at l * 16 (the MFMA operand layout of y3_mfma16's first
// operand); block (cb, kb) at (cb * (k_ld / 32) + kb) << 10.  One thread per 16 bytes.
__global__ __launch_bounds__(256) void weights_to_fragment_order_kernel(const u32x4 *src, u32x4 *dst, int kblocks, long long n16) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n16) return;
  const int l = (int)(i & 63);
  const long long blk = i >> 6;
  const long long cb = blk / kblocks;
  const int kb = (int)(blk - cb * kblocks);
  const int fr = l & 15, fq = l >> 4;
  // MFMA row fr of channel block cb carries channel y3_pair_perm(cb * 16 + fr): after both fragments of a pair a lane holds
  // eight consecutive channels of its pixel (the epilogues work in registers: conv_halo_dw_kernel, conv1x1_dw_kernel)
  const long long ch = (long long)y3_pair_perm((int)(cb * 16 + fr));
  dst[i] = src[(ch * kblocks + kb) * 4 + fq];   // row ch: kblocks * 4 pieces of 16 bytes
}

constexpr int DW_BM = 192;

template <typename T>
int launch_halo_dw(const HaloArgs &a0, hipStream_t s) {
  HaloArgs a = a0;
  static Y3DeviceOnce once;
  int n_cu = 256;
  {
    const int rc = once.run([]() -> int {
      Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_halo_dw_kernel<T, 4>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_halo_dw_kernel<T, 5>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_halo_dw_kernel<T, 6>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      return Y3_OK;
    }, &n_cu);
    if (rc != Y3_OK) return rc;
  }
  a.hr = DW_BM + 2 * a.W + 2;
  a.na = y3_ceil_div(a.hr + 2, 64);                   // + the two zero rows
  a.hr_pad = a.na * 64;
  a.a_bytes = a.hr_pad * 128;
  size_t lds = (size_t)2 * a.a_bytes;
  if (lds < (size_t)DW_BM * 128 * 4) lds = (size_t)DW_BM * 128 * 4;
  Y3_REQUIRE(a.na >= 4 && a.na <= 6 && lds <= 160 * 1024, "direct-weights halo kernel: row width %d does not fit", a.W);
  a.m_tiles = y3_ceil_div(a.M, DW_BM);
  a.n_tiles = a.Cout / 256;
  {
    // tile order over the XCDs: as launch_halo_ws
    const double in_b = (double)a.M * a.Cin, w_b = 9.0 * a.Cin * a.Cout;
    int best = 1;
    double best_cost = 0;
    for (int pn = 1; pn <= 8; pn <<= 1) {
      if (a.n_tiles % pn) break;
      const double cost = pn * in_b + (8.0 / pn) * w_b;
      if (pn == 1 || cost < 0.9 * best_cost) { best = pn; best_cost = cost; }
    }
    a.ngrp_w = a.n_tiles / best;
  }
  const dim3 grid(a.m_tiles * a.n_tiles);
  // halo passes per chunk: 4 (rows of up to 30 pixels), 5 (up to 62), 6 (up to 94)
  if (a.na == 4) Y3_LAUNCH((conv_halo_dw_kernel<T, 4>), grid, dim3(512), lds, s, a);
  else if (a.na == 5) Y3_LAUNCH((conv_halo_dw_kernel<T, 5>), grid, dim3(512), lds, s, a);
  else Y3_LAUNCH((conv_halo_dw_kernel<T, 6>), grid, dim3(512), lds, s, a);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}

template <typename T>
int launch_patch_wsp(const HaloArgs &a0, hipStream_t s) {
  HaloArgs a = a0;
  constexpr int TY = 8, TX = 32, PROWS = (TY + 2) * (TX + 2);
  a.na = y3_ceil_div(PROWS, 32);
  a.hr_pad = a.na * 32;
  a.a_bytes = a.hr_pad * 128;
  const size_t lds = (size_t)4 * 128 * 128 + (size_t)2 * a.a_bytes;
  static_assert(PROWS <= 12 * 32, "all patch slices must be out by tap 5 (4-slot ring)");
  static Y3DeviceOnce once;
  int n_cu = 0;
  {
    const int rc = once.run([]() -> int {
      Y3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_patch_wsp_kernel<T, 4>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      return Y3_OK;
    }, &n_cu);
    if (rc != Y3_OK) return rc;
  }
  const int tiles_x = y3_ceil_div(a.W, TX), tiles_y = y3_ceil_div(a.H, TY);
  const int tiles = tiles_x * tiles_y * (a.M / a.HW) * a.n_tiles;
  const int grid = tiles < n_cu ? tiles : n_cu;
  Y3_LAUNCH((conv_patch_wsp_kernel<T, 4>), dim3(grid), dim3(768), lds, s, a, tiles, tiles_x, tiles_y);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}

}  // namespace

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

// additionally the halo image must fit: 2 x (258 + 2W rows, padded to 32) x 128 B + 3 weight slots <= 160 KiB, and all
// real halo slices must be out by tap 6 of the previous chunk (<= 14 passes of 32 rows)
bool y3_conv_halo_ws_fits(const y3_op &op) {
  if (!y3_conv_halo_eligible(op)) return false;
  const int na = y3_ceil_div(256 + 2 * op.in_w + 4, 32);
  return na <= 14 && (size_t)3 * 128 * 128 + (size_t)2 * na * 32 * 128 <= 160 * 1024;
}

// direct-weights strip kernel (round 5): 16-bit modes, Cout a multiple of 256, rows of up to 94 pixels (four to six 64-row halo
// passes per chunk: the maps the wave-specialised kernel takes)
bool y3_conv_halo_dw_fits(const y3_op &op) {
  if (!y3_conv_halo_eligible(op) || !y3_is16(op.dtype) || op.out_c % 256 != 0 || op.cout_pad % 32 != 0 || op.k_ld % 32 != 0) return false;
  const int na = y3_ceil_div(DW_BM + 2 * op.in_w + 4, 64);
  return na >= 4 && na <= 6 && y3_conv_halo_ws_fits(op);
}

// Where it is the better kernel.  Per workgroup the two strip kernels do the same work per cycle (PMC, profiles/r05s: 64 % of
// the matrix pipe over a workgroup's life either way), so what decides is how the tile count fills the chip: rounds of
// workgroups x time per tile (K-steps x cycles per step + cycles outside the loop; stamps / PMC of both kernels).  yolov3 @ 608,
// batch 16: 256 -> 512 at 38^2 is 364 tiles of 256 x 128 in two rounds against 242 of 192 x 256 in one: 58.5 -> 49.5 us,
// 930 -> 1100 TFLOP/s per launch (profiles/r05s_halo_dw.txt).  Layers with fewer than four channel chunks stay on the
// wave-specialised kernel (76^2: its shorter prologue / epilogue wins), and so does everything the model puts within 7 %.
bool y3_conv_halo_dw_pays(const y3_op &op) {
  if (!y3_conv_halo_dw_fits(op)) return false;
  const int es = y3_elem_size(op.dtype);
  const int nchunks = op.in_c / (128 / es), n_cu = y3_device_cus();
  if (nchunks < 4) return false;
  const int M = op.batch * op.in_h * op.in_w;
  const int mi = halo_tile_fragments(M, op.out_c / 128, nchunks, n_cu);
  const long long t_ws = (long long)y3_ceil_div(M, 64 * mi) * (op.out_c / 128), t_dw = (long long)y3_ceil_div(M, DW_BM) * (op.out_c / 256);
  const double c_ws = (double)((t_ws + n_cu - 1) / n_cu) * (nchunks * 9.0 * (256.0 * mi + 300.0) + 13000.0);
  const double c_dw = (double)((t_dw + n_cu - 1) / n_cu) * (nchunks * 9.0 * 1920.0 + 17000.0);
  return c_dw < 0.93 * c_ws;
}

size_t y3_conv_halo_dw_weight_bytes(const y3_op &op) { return (size_t)op.cout_pad * op.k_ld * 2; }

// the fragment-order copy of op's weights into `dst` (y3_conv_halo_dw_weight_bytes), on `s`
int y3_conv_halo_dw_make_weights(const y3_op &op, void *dst, hipStream_t s) {
  const long long n16 = (long long)op.cout_pad * op.k_ld * 2 / 16;
  Y3_LAUNCH(weights_to_fragment_order_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s,
                     static_cast<const u32x4 *>(op.d_weight), static_cast<u32x4 *>(dst), op.k_ld / 32, n16);
  Y3_HIP_CHECK(hipGetLastError());
  return Y3_OK;
}

// `frag_w`: the plan's fragment-order copy of the weights; nullptr (single-op calls, unit tests): made here, stream-ordered
int y3_launch_conv_halo_dw(const y3_op &op, const void *d_in, const void *d_zero, hipStream_t s,
                           const char **kernel_name, bool dry_run, const void *frag_w) {
  Y3_REQUIRE(y3_conv_halo_dw_fits(op), "conv block %d: shape not supported by the direct-weights halo kernel", op.block_idx);
  *kernel_name = Y3_KNAME(op.dtype, "conv_halo_dw_", "_192x256");
  if (dry_run) return Y3_OK;
  void *tmp = nullptr;
  if (!frag_w) {
    Y3_HIP_CHECK(hipMallocAsync(&tmp, y3_conv_halo_dw_weight_bytes(op), s));
    const int rc = y3_conv_halo_dw_make_weights(op, tmp, s);
    if (rc != Y3_OK) { (void)hipFreeAsync(tmp, s); return rc; }
    frag_w = tmp;
  }
  HaloArgs a;
  a.in = static_cast<const char *>(d_in);
  a.wgt = static_cast<const char *>(frag_w);
  a.scale = op.d_scale; a.bias = op.d_bias;
  a.res = static_cast<const char *>(op.d_res);
  a.out = static_cast<char *>(op.d_out);
  a.zero = static_cast<const char *>(d_zero);
  a.H = op.in_h; a.W = op.in_w; a.Cin = op.in_c; a.in_ld = op.in_ld;
  a.Cout = op.out_c; a.out_ld = op.out_ld; a.res_ld = op.res_ld;
  a.HW = op.in_h * op.in_w;
  a.M = op.batch * a.HW;
  a.k_ld = op.k_ld;
  a.nchunks = op.in_c / 64;
  a.n_tiles = op.out_c / 256;
  a.hr = a.hr_pad = a.na = a.a_bytes = 0;
  a.ngrp_w = a.n_tiles; a.m_tiles = 0;
  fast_div((uint32_t)a.HW, a.mul_hw, a.sh_hw);
  fast_div((uint32_t)a.W, a.mul_w, a.sh_w);
  a.flags = op.flags;
  int rc = Y3_OK;
  if ((long long)op.batch * a.HW >= (1ll << 31)) {
    y3_set_error("conv block %d: too many pixels for the 32-bit tile index", op.block_idx);
    rc = Y3_ERR_INVALID;
  } else {
    rc = y3_by_dtype16(op.dtype, [&](auto tag) { return launch_halo_dw<decltype(tag)>(a, s); });
  }
  if (tmp) (void)hipFreeAsync(tmp, s);
  return rc;
}

// 2-D patch kernel: same layer class, any number (>= 1) of channel chunks, any row width
bool y3_conv_patch_fits(const y3_op &op) {
  const int es = y3_elem_size(op.dtype);
  const int bke = 128 / es;
  if (op.ksize != 3 || op.stride != 1 || op.pad != 1) return false;
  if (op.flags & (Y3_F_OUT_F32 | Y3_F_IN_NCHW_F32 | Y3_F_IN_NHWC_U8BGR)) return false;
  if (op.in_c % bke != 0 || op.out_c % 128 != 0 || op.out_ld % 8 != 0 || op.in_ld % (16 / es) != 0) return false;
  if ((op.flags & Y3_F_RESIDUAL) && op.res_ld % 8 != 0) return false;
  return op.k_ld >= 9 * op.in_c;
}

int y3_launch_conv_patch(const y3_op &op, const void *d_in, const void *d_zero, hipStream_t s,
                         const char **kernel_name, bool dry_run) {
  const int es = y3_elem_size(op.dtype);
  Y3_REQUIRE(y3_conv_patch_fits(op), "conv block %d: shape not supported by the patch kernel", op.block_idx);
  *kernel_name = Y3_KNAME(op.dtype, "conv_patch_wsp_", "_8x32x128");
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
  a.ngrp_w = a.n_tiles; a.m_tiles = 0;
  a.mul_hw = a.sh_hw = a.mul_w = a.sh_w = 0;
  a.flags = op.flags | (y3_debug_flags() ? 0x40000000u : 0u);
  return y3_by_dtype(op.dtype, [&](auto tag) { return launch_patch_wsp<decltype(tag)>(a, s); });
}

int y3_launch_conv_halo(const y3_op &op, const void *d_in, const void *d_zero, hipStream_t s,
                        const char **kernel_name, bool dry_run) {
  const int es = y3_elem_size(op.dtype);
  Y3_REQUIRE(y3_conv_halo_ws_fits(op), "conv block %d: shape not supported by the halo kernel", op.block_idx);
  const int mi = halo_tile_fragments(op.batch * op.in_h * op.in_w, op.out_c / 128, op.in_c / (128 / es), y3_device_cus());
  if (mi == 3) *kernel_name = Y3_KNAME(op.dtype, "conv_halo_ws_", "_192x128");
  else *kernel_name = Y3_KNAME(op.dtype, "conv_halo_ws_", "_256x128");
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
  a.ngrp_w = a.n_tiles; a.m_tiles = 0;
  fast_div((uint32_t)a.HW, a.mul_hw, a.sh_hw);
  fast_div((uint32_t)a.W, a.mul_w, a.sh_w);
  a.flags = op.flags | (y3_debug_flags() ? 0x40000000u : 0u);
  Y3_REQUIRE((long long)op.batch * a.HW < (1ll << 31), "conv block %d: too many pixels for the 32-bit tile index", op.block_idx);
  return y3_by_dtype(op.dtype, [&](auto tag) { return launch_halo_ws<decltype(tag)>(a, s); });
}

Y3_STAMP_READER(y3_debug_stamps_halo)
