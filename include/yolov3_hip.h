/*
 * yolov3_hip.h -- C ABI of libyolov3_hip.so: the MI355X (gfx950) YOLOv3 inference hot path.
 *
 * The reference (nrsyed/pytorch-yolov3) has no FFI layer: its hot path is the Python
 * call chain Darknet.forward -> YOLOLayer.forward -> inference() tail -> non_max_suppression.
 * Each entry point below replaces the arithmetic of one of those call sites; the Python
 * package pytorch-yolov3_amd/yolov3 keeps the reference's signatures and binds these
 * symbols with ctypes.  Citations are into /root/reference.
 *
 * Conventions
 *   - plain C types only; every pointer named d_* is a DEVICE address (hipMalloc'd by the
 *     caller, e.g. torch tensor .data_ptr()); the library never allocates or frees caller-visible memory and never
 *     synchronises the device.  One exception, for callers that do not pass y3_op.d_weight_frag: a plan then makes and owns a
 *     private fragment-order copy of the weights of every layer it gives to the direct-weights strip kernel (hipMalloc at
 *     y3_plan_create on the device that owns d_weight, one stream synchronisation there, hipFree at y3_plan_destroy);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - every function returns 0 on success, a negative Y3_ERR_* code otherwise, and
 *     y3_last_error() then returns a human-readable message (thread-local);
 *   - activations are NHWC ("pixel-major"): element (b, y, x, c) of a tensor with pixel
 *     stride ld lives at ((b*H + y)*W + x)*ld + c;  ld >= C, and ld*sizeof(elem) as well
 *     as every channel-slice offset are multiples of 16 bytes on the MFMA path;
 *   - one plan / one stream per GPU; entry points are not re-entrant per plan.
 */
#ifndef YOLOV3_HIP_H
#define YOLOV3_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Y3_ABI_VERSION 6

/* error codes */
#define Y3_OK 0
#define Y3_ERR_INVALID (-1)   /* bad argument / unsupported shape */
#define Y3_ERR_HIP (-2)       /* a HIP runtime call failed        */
#define Y3_ERR_NODEVICE (-3)  /* no gfx950 device visible         */

/* element types of activations / weights */
#define Y3_F32 0
#define Y3_BF16 1
#define Y3_F16 2   /* IEEE half storage, float32 accumulation: every kernel of the bf16 mode instantiated on
                      v_mfma_f32_16x16x32_f16 (same rate, same bytes, 11 instead of 8 significand bits) */
#define Y3_F64 3   /* float64: box coordinates handed to y3_nms_float / y3_cxywh_to_tlbr_float only (no network runs in it) */

/* op kinds of a plan (one per Darknet block that does work) */
#define Y3_OP_CONV 1      /* conv -> (BN as per-channel scale/bias) -> LeakyReLU -> (+residual)  darknet.py:236-264, :376-379 */
#define Y3_OP_MAXPOOL 2   /* darknet.py:16-29 (zero-pad right/bottom quirk when stride==1)        */
#define Y3_OP_UPSAMPLE 3  /* nearest, integer factor: darknet.py:299-305                        */
#define Y3_OP_ADD 4       /* unfused shortcut: darknet.py:376-379                               */
#define Y3_OP_COPY 5      /* unfused route slice copy: darknet.py:369-375                       */
#define Y3_OP_YOLO 6      /* YOLOLayer.forward + head concat + wh/net size: darknet.py:48-122, :389-399 */

/* y3_op.flags */
#define Y3_F_LEAKY 1u          /* LeakyReLU(0.1) after scale/bias                        */
#define Y3_F_RESIDUAL 2u       /* add d_res (same dtype as output) after the activation  */
#define Y3_F_OUT_F32 4u        /* store float32 regardless of dtype (detection-head convs) */
#define Y3_F_IN_NCHW_F32 8u    /* conv input is the network input, float32 (B,C,H,W) in [0,1]  (inference.py:332-335) */
#define Y3_F_IN_NHWC_U8BGR 16u /* conv input is uint8 (B,H,W,3) BGR frames; kernel applies BGR->RGB and /255.0 (inference.py:332-333) */
#define Y3_F_PLAN_INPUT 32u    /* d_in is supplied at y3_plan_run time                  */
#define Y3_F_FUSE_NEXT 64u     /* plan hint: this conv's output is read by the NEXT op only, so the executor may run
                                  both as one kernel (stem + stride-2 conv) and never materialise d_out      */

/*
 * One unit of work.  POD, 8-byte aligned, zero-initialise unused fields.
 * Weight layouts (prepared on the host by yolov3/darknet.py:load_weights):
 *   igemm path  : d_weight[co][ (ky*ks + kx)*Cin + ci ]  row stride k_ld elements (zero padded),
 *                 rows padded with zeros up to a multiple of 128; dtype = op dtype
 *   small-Cin   : (Cin <= 4) d_weight as float32 [ (ky*ks+kx)*Cin + ci ][ co ] with row stride = cout_pad
 *   direct path : same as igemm layout
 * d_scale/d_bias: float32 per output channel (BN folded to y = conv*scale + bias; scale=1 for
 * convs with a bias and no BN), padded like the weight rows.
 */
typedef struct y3_op {
  int32_t kind;
  int32_t dtype;
  uint32_t flags;
  int32_t batch;
  int32_t in_h, in_w, in_c, in_ld;
  int32_t out_h, out_w, out_c, out_ld;
  int32_t ksize, stride, pad;     /* conv / maxpool / upsample factor in `stride` */
  int32_t res_ld;
  int32_t k_ld;                   /* weight row stride (elements)                  */
  int32_t cout_pad;               /* padded rows of weight / length of scale, bias */
  const void *d_in;
  void *d_out;
  const void *d_res;
  const void *d_weight;
  const float *d_scale;
  const float *d_bias;
  /* Y3_OP_YOLO: d_in is the float32 head tensor (B,h,w,ld) with channel = anchor*n_attr + attr */
  int32_t n_anchor, n_attr;
  float anchor_w[8], anchor_h[8]; /* pixels, already selected by `mask` (darknet.py:44) */
  int32_t row_offset, rows_total; /* this head's first row / total rows M of the concatenated output */
  float net_w, net_h;             /* cfg [net] width/height (darknet.py:395-399)          */
  float *d_bbox;                  /* (B, rows_total, 4) cx,cy in [0,1], w,h / net size     */
  float *d_prob;                  /* (B, rows_total)                                       */
  int64_t *d_cls;                 /* (B, rows_total)                                       */
  int32_t block_idx;              /* Darknet block this op implements (diagnostics)        */
  int32_t reserved;
  /* ABI 6: the conv's weights in MFMA-fragment order (y3_conv_make_fragment_weights), y3_conv_fragment_weight_bytes() bytes,
   * for the layers the direct-weights strip kernel takes; shared by every plan of the network.  NULL: a plan that needs the
   * copy makes a private one (see the conventions above).                                                              */
  const void *d_weight_frag;
} y3_op;

typedef struct y3_plan y3_plan;

/*
 * y3_options.auto_mask: per-layer kernel choice of the MFMA convolutions, one bit per rule (profiles/ holds the A/B tables
 * behind every default).  Results do not depend on it: all MFMA conv kernels sum a layer in the same K order (channel chunk
 * outermost, filter tap innermost), so a layer's output is bit-identical whichever of them runs.
 */
#define Y3_AM_HALO_WIDE 0x0001u      /* halo-reuse kernel for 3x3 stride-1 layers (Cin, Cout >= 128) with rows of more than 64 px */
#define Y3_AM_IGEMM3_MID 0x0002u     /* wave-specialised implicit GEMM for such layers with rows of 33..64 px (if not HALO_MID)   */
#define Y3_AM_HALO_NARROW 0x0004u    /* halo-reuse kernel for rows of <= 32 px                                                     */
#define Y3_AM_IGEMM3_1X1_DEEP 0x0008u /* wave-specialised implicit GEMM for 1x1 layers with Cin >= 1024                            */
#define Y3_AM_HALO_MID 0x0010u       /* halo-reuse kernel for rows of 33..64 px                                                    */
#define Y3_AM_IGEMM3_NARROW 0x0020u  /* wave-specialised implicit GEMM for rows of <= 32 px (if not HALO_NARROW)                   */
#define Y3_AM_IGEMM3_1X1_BM64 0x0040u /* 64-pixel tiles of it for bf16 1x1 layers with Cin >= 256 (A/B only)                       */
#define Y3_AM_PATCH_WIDE 0x0080u     /* 2-D patch kernel (8 x 32 output tiles) for 3x3 stride-1 layers with rows wider than 128 px */
#define Y3_AM_HALO_TILE256 0x0200u   /* halo kernel with 256-pixel tiles only: the THROUGHPUT choice of callers that keep several  */
                                     /* batches in flight on their own streams (default: 192-pixel tiles where they shorten one forward) */
#define Y3_AM_NO_BN_SHRINK 0x0400u   /* A/B: implicit-GEMM channel tiles by Cout only (no narrower tiles on small grids)           */
#define Y3_AM_NO_SMALL_GRID 0x0800u  /* A/B: no small-grid / stride-2 rerouting (the round-1 selection)                            */
#define Y3_AM_NO_WRES 0x1000u        /* A/B: never the weights-resident persistent 1x1 kernel                                      */
#define Y3_AM_HALO_DW 0x4000u        /* direct-weights strip kernel (192 x 256 tiles, weight fragments straight from global memory) */
                                     /* for 3x3 layers whose tile count fills the chip better that way (round 5)                   */
#define Y3_AM_HALO_DW_ALWAYS 0x8000u /* tests / A-B: that kernel wherever it fits                                                  */
#define Y3_AM_1X1_DW 0x10000u        /* direct-weights 1x1 kernel (whole activation tile in LDS, weight fragments straight from global */
                                     /* memory) for the short-K bottleneck layers whose map x batch gives every CU one tile (round 6) */
#define Y3_AM_SMALL_DW 0x20000u       /* small-grid direct-weights kernel (48-pixel x 32..256-channel tiles, whole halo in LDS) for 1x1 and   */
                                     /* 3x3 layers whose map x batch fits the chip in ONE round of such tiles: one frame at a time (round 6)  */
#define Y3_AM_SMALL_DW_ALWAYS 0x40000u /* tests / A-B: that kernel wherever its shape constraints hold, whatever the grid                       */
#define Y3_AM_SMALL_DW_WIDE 0x80000u  /* ... and for grids a little over one round where it measured faster than what the layer ran on: 3x3 layers  */
                                     /* up to 1.5 rounds of its widest tiles; 1x1 layers with ONE channel tile per pixel tile (128+ channels) up   */
                                     /* to 8 rounds, after the weights-resident kernel has had its say (batches of 3-8; two layers at batch 16)   */
#define Y3_AM_WRES_ALWAYS 0x2000u    /* tests: that kernel on every layer it supports, whatever the map size                       */
#define Y3_AM_DEFAULT (Y3_AM_HALO_WIDE | Y3_AM_HALO_NARROW | Y3_AM_IGEMM3_1X1_DEEP | Y3_AM_HALO_MID | Y3_AM_PATCH_WIDE | Y3_AM_HALO_DW | Y3_AM_1X1_DW | Y3_AM_SMALL_DW | Y3_AM_SMALL_DW_WIDE)   /* 0xb409d */
#define Y3_AM_IGEMM_ONLY 0u          /* LDS-DMA implicit GEMM (igemm_version) everywhere                                           */

/*
 * Kernel-selection options of ONE plan (fixed when the plan is created).  y3_options_default() fills in the
 * library's defaults -- the measured-best choices, profiles/ -- as modified by y3_set_tuning().
 *   auto_mask        Y3_AM_* bits, default Y3_AM_DEFAULT
 *   unused0          (was halo_persistent: the persistent strip kernels measured slower twice and were removed,
 *                    profiles/r03b_persistent_halo_wsq_investigation.txt; ignored)
 *   igemm_version    1 register-staged, 2 LDS-DMA double-buffered [default], 3 wave-specialised
 *   igemm_ns         LDS stages of version 3 (3 or 4; less means 3);  igemm_bm  64 = 64-pixel tiles for version 3 (bf16)
 *   use_graph        1: y3_plan_run replays a captured hipGraph (one launch per forward) on non-default streams;
 *                    0 [default]: every kernel is launched individually (measured 1 % faster at batch 16)
 *   fuse_stem        1 [default]: conv pairs flagged Y3_F_FUSE_NEXT run as one kernel where one exists; 2: same, with the
 *                    phase-by-phase form of the fused stem kernel (same bits, slower: A/B and tests); 0: never fused
 *   fuse_head        1 [default]: detection-head conv + YOLO decode in one launch (16-bit networks), on the direct-weights 1x1 kernel
 *                    where the head conv has 256 / 512 / 1024 input channels (it reads the fragment-order copy of the weights:
 *                    y3_op.d_weight_frag or a private copy; 48-pixel tiles), else on the tiled kernel; 2: the tiled kernel only; 3 / 4:
 *                    the direct-weights kernel with 48- / 96-pixel tiles wherever the shape allows (same bits all four: A/B and tests);
 *                    0: two launches
 *   fuse_spp         1 [default]: three stride-1 max-pools (5 / 9 / 13) of one tensor in one launch
 *   decode_lanes     4 [default]: four lanes per box in the bf16 decode; 1: sequential class loop everywhere
 *   fuse_block       0 [default]: off.  1: a 1x1 conv (-> 128 channels) + the 3x3 conv that is its only reader (+ the
 *                    shortcut add) run as ONE kernel with the 128-channel tensor kept in LDS (csrc/conv_block.hip), where map
 *                    and batch give enough workgroups to fill the chip (yolov3@608: the 76^2 blocks from batch 12 on); 2:
 *                    wherever the kernel supports the pair (tests).  Same bits either way; measured level with the two
 *                    launches (profiles/r04_block_fused_AB.txt), which is why it is not the default.
 */
typedef struct y3_options {
  int32_t auto_mask, unused0, igemm_version, igemm_ns, igemm_bm;
  int32_t use_graph, fuse_stem, fuse_head, fuse_spp, decode_lanes;
  int32_t fuse_block;
  int32_t reserved[5];
} y3_options;

/* library / device ------------------------------------------------------------------- */
int y3_abi_version(void);
const char *y3_last_error(void);
/* number of visible HIP devices whose arch is gfx950 (0 on a CPU-only machine) */
int y3_device_count(void);

/* plan executor: replaces the block loop of Darknet.forward (darknet.py:366-399) -------- */
/* copies `ops` (host array); `d_zero` = >= 256 bytes of zeroed device memory that outlives the plan */
int y3_plan_create(const y3_op *ops, int n_ops, const void *d_zero, y3_plan **out_plan);
/* same with explicit options (NULL = the current defaults); the plan keeps its own copy */
int y3_plan_create_ex(const y3_op *ops, int n_ops, const void *d_zero, const y3_options *options, y3_plan **out_plan);
void y3_options_default(y3_options *options);
void y3_plan_destroy(y3_plan *plan);
/* launches every op on `stream` (with the "use_graph" knob: from the second call on, on a non-default stream, as one
 * captured hipGraph per distinct d_input); `d_input` feeds ops flagged Y3_F_PLAN_INPUT */
int y3_plan_run(y3_plan *plan, const void *d_input, void *stream);
/* same, bracketing every op with HIP events; after the call ms_per_op[i] holds op i's device
 * time in milliseconds (synchronises the stream; for bench.py / profiling only)              */
int y3_plan_run_timed(y3_plan *plan, const void *d_input, void *stream, float *ms_per_op);
/* same run, but every kernel is launched with a start / stop event pair bound to its DISPATCH (hipExtLaunchKernel):
 * kernel_ms_per_op[i] is the device time of op i's kernel(s) from begin to end -- what rocprofv3's kernel trace reports --
 * without the dispatch / barrier-packet handling between two stream events that y3_plan_run_timed's figure includes
 * (~5 us per launch on an MI355X).  Synchronises the stream; for bench.py / profiling only.                       */
int y3_plan_run_profiled(y3_plan *plan, const void *d_input, void *stream, float *kernel_ms_per_op);
/* name of the kernel an op dispatches to, e.g. "conv_igemm_bf16_128x128" (static string)     */
const char *y3_plan_op_kernel(const y3_plan *plan, int op_index);
/* algorithmic FLOPs (2*k*k*Cin*Cout*Hout*Wout*B for convs, else 0) and compulsory bytes      */
double y3_plan_op_flops(const y3_plan *plan, int op_index);
double y3_plan_op_bytes(const y3_plan *plan, int op_index);

/* kernel family a Y3_OP_CONV op dispatches to: 0 = MFMA implicit GEMM (weights [cout_pad][k_ld] in
 * the op dtype), 1 = 3-channel stem (weights float32 [27][cout_pad]), 2 = direct fallback (same
 * layout as 0), 3 = MFMA stem for uint8 BGR frames -> bf16 (weights bf16 [32][32], k = ky*9 + kx*3 +
 * byte-channel, zero padded; scale/bias 32 floats); -1 if `op` is not a conv.  The host lays weights
 * out accordingly.                                                                              */
int y3_conv_path(const y3_op *op);

/* Fragment-order weights of the direct-weights strip kernel (csrc/conv_halo.hip: 1-KiB blocks of 16 output channels x 32
 * K-elements in the MFMA operand layout).  y3_conv_fragment_weight_bytes: size of the copy if a plan created with `options`
 * (NULL = the current defaults) runs `op` on that kernel, else 0 -- the host makes ONE copy per layer and device with
 * y3_conv_make_fragment_weights (reads op->d_weight, [cout_pad][k_ld]; stream-ordered on `stream`) and passes it to every
 * plan in y3_op.d_weight_frag.  No reference counterpart (a layout of the parameters darknet.py:415-476 loads).          */
size_t y3_conv_fragment_weight_bytes(const y3_op *op, const y3_options *options);
int y3_conv_make_fragment_weights(const y3_op *op, void *d_dst, void *stream);

/* A/B measurements only (tools/conv_bench.py, bench.py --tuning): changes ONE field of the process-wide DEFAULT
 * options by name ("auto_mask", "igemm_version", "igemm_ns", "igemm_bm", "use_graph", "fuse_stem",
 * "fuse_head", "fuse_spp", "decode_lanes", "fuse_block").  Plans created afterwards without explicit options pick it up; existing
 * plans keep the options they were created with.                                                                  */
int y3_set_tuning(const char *key, int value);

/* single op (unit tests): same dispatch as inside a plan */
int y3_op_run(const y3_op *op, const void *d_input, const void *d_zero, void *stream);

/* detection tail: replaces inference.py:342-366 (threshold, scale, int cast, cxywh_to_tlbr,
 * per-class non_max_suppression, gather) for a whole batch, on device ---------------------- */
size_t y3_detect_workspace_bytes(int batch, int rows);
/*
 * d_bbox (batch,rows,4) f32, d_prob (batch,rows) f32, d_cls (batch,rows) i64: Darknet.forward outputs.
 * d_orig_hw (batch,2) int32: original frame height,width (inference.py:351-352).
 * Outputs, capacity `rows` per frame, detections ordered by (class asc, score desc, row desc):
 *   d_det_count (batch) int32; d_det_tlbr (batch,rows,4) int64; d_det_prob (batch,rows) f32;
 *   d_det_cls (batch,rows) int64; d_det_row (batch,rows) int32 = row index into the `rows` predictions.
 */
int y3_detect(const float *d_bbox, const float *d_prob, const int64_t *d_cls, int batch, int rows,
              const int32_t *d_orig_hw, float prob_thresh, double iou_thresh, void *d_workspace,
              size_t workspace_bytes, int32_t *d_det_count, int64_t *d_det_tlbr, float *d_det_prob,
              int64_t *d_det_cls, int32_t *d_det_row, void *stream);

/* non_max_suppression (inference.py:161-266) on caller-provided integer boxes -------------- */
size_t y3_nms_workspace_bytes(int n);
/*
 * d_tlbr (n,4) int64, d_prob (n) f32, d_cls (n) int64 or NULL (class-agnostic).
 * d_keep (n) int64 receives the kept indices ordered by (class asc, score desc, index desc);
 * d_keep_count (1) int32.
 */
int y3_nms(const int64_t *d_tlbr, const float *d_prob, const int64_t *d_cls, int n, double iou_thresh,
           void *d_workspace, size_t workspace_bytes, int64_t *d_keep, int32_t *d_keep_count,
           void *stream);

/* the same on FLOATING-POINT boxes: the reference's non_max_suppression takes any numeric dtype (inference.py:161-217;
 * its own inference() only passes integers) and numpy then computes areas, intersections, IoU and the comparison with the
 * threshold in the array's dtype.  d_tlbr (n,4) float32 (box_dtype Y3_F32) or float64 (Y3_F64), d_prob (n) float64
 * (exact for float32 / float64 scores), d_cls (n) int64 or NULL; outputs and order as y3_nms. */
size_t y3_nms_float_workspace_bytes(int n);
int y3_nms_float(const void *d_tlbr, int box_dtype, const double *d_prob, const int64_t *d_cls, int n, double iou_thresh,
                 void *d_workspace, size_t workspace_bytes, int64_t *d_keep, int32_t *d_keep_count, void *stream);

/* cxywh_to_tlbr (inference.py:269-283) on int64 rows of `cols` >= 4 columns ---------------- */
int y3_cxywh_to_tlbr(const int64_t *d_xywh, int64_t *d_tlbr, int n, int cols, void *stream);
/* ... and on float32 / float64 rows (dtype Y3_F32 / Y3_F64): tl = c - floor(wh / 2), br = c + floor(wh / 2), numpy's `//` */
int y3_cxywh_to_tlbr_float(const void *d_xywh, void *d_tlbr, int n, int cols, int dtype, void *stream);

/* frame resize on device (replaces the host cv2.resize of inference.py:320-326): uint8
 * (src_h,src_w,3) -> (dst_h,dst_w,3) with OpenCV's 8-bit INTER_LINEAR arithmetic: per channel
 *   top = S[ylo][xlo]*ax0 + S[ylo][xhi]*ax1,  bot = the same on row yhi   (int, 11-bit coefficients)
 *   out = (((ay0 * (top >> 4)) >> 16) + ((ay1 * (bot >> 4)) >> 16) + 2) >> 2
 * d_ytab (dst_h,4) / d_xtab (dst_w,4) int32 rows = {lo index, hi index, weight_lo, weight_hi}: OpenCV's yofs / ibeta
 * and xofs / ialpha tables, computed on the host (yolov3/preprocess.py:axis_table). */
int y3_resize_bilinear_u8(const uint8_t *d_src, int src_h, int src_w, uint8_t *d_dst, int dst_h, int dst_w,
                          const int32_t *d_ytab, const int32_t *d_xtab, void *stream);

/* frames in / detections out without a copy engine: what yolov3/pipeline.py (the loop bench.py times and detect_in_frames
 * runs) uses in place of the `.to(device)` / `.cpu()` transfers around Darknet.forward (inference.py:335, :338-340).  A small
 * grid of `blocks` workgroups (<= 0: 32) moves `nbytes` bytes with 16-byte accesses.  `src` or `dst` may be PINNED (registered)
 * host memory, which the GPU addresses directly over PCIe; both must be 16-byte aligned, and a pointer the runtime does not
 * know (pageable host memory) is rejected with Y3_ERR_INVALID.  Unlike hipMemcpyAsync it needs no copy-engine hand-over on
 * the stream, so batches in flight on other streams keep the chip (measured: profiles/r03c_pcie_inclusive.txt). */
int y3_copy_bytes(const void *src, void *dst, size_t nbytes, int blocks, void *stream);

/* padded fixed-size detection records for the multi-GPU all-gather (no reference counterpart:
 * the reference is single device).  Record = 8 x int32: x1,y1,x2,y2 (int64 corners saturated to int32), score bits, class, row, and the
 * frame's TRUE detection count (0 in padding records: the field doubles as the valid flag; a count
 * above kmax means the frame was truncated to its kmax best-ordered detections).
 * d_records (batch, kmax, 8) int32; d_rec_count (batch) int32 also receives the true counts, may be NULL. */
int y3_pack_records(const int32_t *d_det_count, const int64_t *d_det_tlbr, const float *d_det_prob,
                    const int64_t *d_det_cls, const int32_t *d_det_row, int batch, int rows, int kmax,
                    int32_t *d_records, int32_t *d_rec_count, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLOV3_HIP_H */
