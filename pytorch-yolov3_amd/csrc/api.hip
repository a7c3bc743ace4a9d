// C-ABI glue: error string, device query, op dispatch and the plan executor that replaces the
// block loop of Darknet.forward (/root/reference/yolov3/darknet.py:366-399).
#include <stdarg.h>
#include <string.h>

#include <atomic>
#include <new>
#include <vector>

#include "common.h"

static_assert(sizeof(y3_op) == 248, "y3_op layout is part of the ABI (ctypes mirror in yolov3/_hip.py)");

namespace {
thread_local char g_err[512] = "";
}

// Default options (y3_options_default / y3_set_tuning); the auto_mask bits are named in include/yolov3_hip.h (Y3_AM_*).
static y3_options g_y3_defaults = {/*auto_mask*/ (int32_t)Y3_AM_DEFAULT, /*unused0*/ 0, /*igemm_version*/ 2, /*igemm_ns*/ 2,
                                   /*igemm_bm*/ 0, /*use_graph*/ 0, /*fuse_stem*/ 1, /*fuse_head*/ 1, /*fuse_spp*/ 1,
                                   /*decode_lanes*/ 4, /*fuse_block*/ 0, {0, 0, 0, 0, 0}};
static thread_local const y3_options *tl_y3_opt = nullptr;
const y3_options &y3_opt() { return tl_y3_opt ? *tl_y3_opt : g_y3_defaults; }
static thread_local Y3KernelTimer *tl_y3_timer = nullptr;
Y3KernelTimer *y3_kernel_timer() { return tl_y3_timer; }
// y3_set_tuning("debug", v): exists in DIAGNOSTIC builds only (`make variant FLAGS=-DY3_X_...`, `make stamps`); the product
// library rejects the key, so a benchmark line can never come from kernels that skip work
#if defined(Y3_X_NOEPI) || defined(Y3_X_S2BOUND) || defined(Y3_X_DEBUG) || defined(Y3_STAMPS)
#define Y3_HAS_DEBUG_KEY 1
static int g_y3_debug = 0;
int y3_debug_flags() { return g_y3_debug; }
#else
int y3_debug_flags() { return 0; }
#endif

// CU count of the current device (256 on an MI355X; 256 as well when no device is visible: dry runs on a CPU box)
int y3_device_cus() {
  static std::atomic<int> cus[32];                 // (zero-initialised; concurrent first calls store the same value)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) { (void)hipGetLastError(); return 256; }
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}
namespace {
struct OptScope {   // the launchers called below this frame see the plan's options
  const y3_options *prev;
  explicit OptScope(const y3_options *o) : prev(tl_y3_opt) { tl_y3_opt = o; }
  ~OptScope() { tl_y3_opt = prev; }
};
}  // namespace

void y3_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

struct y3_plan {
  std::vector<y3_op> ops;
  std::vector<const char *> kernel;
  const void *d_zero;
  std::vector<char> fuse;   // per op: 0 = launch normally, 1 / 3 / 4 / 6 = launch fused with the next op (stem pair /
                            // 64-32-64 residual block / head conv + decode / 128-channel bottleneck block), 5 = SPP
                            // pyramid with the next TWO ops, 2 = nothing (fused into a previous op)
  std::vector<void *> frag_w;   // per op: fragment-order copy of a conv's weights (direct-weights strip kernel), or null
  std::vector<char> frag_own;   // ... and whether the plan made (and frees) it: callers of ABI 6 pass a shared copy in the op
  std::vector<hipEvent_t> events;
  std::vector<hipEvent_t> kstart, kstop;   // y3_plan_run_profiled: event pairs bound to the dispatches (two per op)
  // hipGraph replay (one graph launch per forward instead of ~80 kernel launches): executable graphs keyed by the
  // input pointer they were captured with; the first run of a plan is always eager (one-time function attributes)
  struct GraphEntry { const void *input; hipGraphExec_t exec; };
  std::vector<GraphEntry> graphs;
  bool warmed = false, graph_ok = true;
  y3_options opt;           // fixed at creation
};

namespace {

// 0 = MFMA implicit GEMM, 1 = 3-channel stem kernel (VALU), 2 = direct fallback, 3 = MFMA stem (uint8 -> bf16)
int conv_path(const y3_op &op) {
  const bool net_input = op.flags & (Y3_F_IN_NCHW_F32 | Y3_F_IN_NHWC_U8BGR);
  if (y3_conv_stem_mfma_supported(op)) return 3;
  if (net_input && op.in_c == 3 && op.ksize == 3 && !(op.flags & Y3_F_RESIDUAL)) return 1;
  if (!net_input && y3_conv_igemm_supported(op)) return 0;
  return 2;
}

int dispatch(const y3_op &op, const void *d_input, const void *d_zero, hipStream_t s, const char **name,
             bool dry_run, const void *frag_w = nullptr) {
  const void *in = (op.flags & Y3_F_PLAN_INPUT) ? d_input : op.d_in;
  if (!dry_run) {
    Y3_REQUIRE(in != nullptr, "op for block %d has no input pointer", op.block_idx);
    Y3_REQUIRE(op.d_out != nullptr || op.kind == Y3_OP_YOLO, "op for block %d has no output pointer", op.block_idx);
  }
  Y3_REQUIRE(op.dtype == Y3_F32 || op.dtype == Y3_BF16 || op.dtype == Y3_F16, "op for block %d: unknown dtype %d", op.block_idx, op.dtype);
  Y3_REQUIRE(op.batch > 0 && op.in_h > 0 && op.in_w > 0 && op.in_c > 0, "op for block %d: empty input shape", op.block_idx);
  switch (op.kind) {
    case Y3_OP_CONV: {
      Y3_REQUIRE(op.out_h == (op.in_h + 2 * op.pad - op.ksize) / op.stride + 1 &&
                     op.out_w == (op.in_w + 2 * op.pad - op.ksize) / op.stride + 1,
                 "conv block %d: output size mismatch", op.block_idx);
      if (!dry_run)
        Y3_REQUIRE(op.d_weight && op.d_scale && op.d_bias, "conv block %d: missing parameters", op.block_idx);
      switch (conv_path(op)) {
        case 0: {
          if (y3_opt().auto_mask) {
            const unsigned am = (unsigned)y3_opt().auto_mask;
            const int w = op.in_w;
            const bool k3 = op.ksize == 3 && op.stride == 1 && op.in_c >= 128 && op.out_c >= 128 && !(op.flags & Y3_F_OUT_F32);
            const bool halo_ok = k3 && y3_conv_halo_ws_fits(op);
            bool want_halo = false, want_ws = false;
            if (k3 && w > 64) want_halo = am & Y3_AM_HALO_WIDE;
            else if (k3 && w > 32) { want_halo = am & Y3_AM_HALO_MID; want_ws = am & Y3_AM_IGEMM3_MID; }
            else if (k3) { want_halo = am & Y3_AM_HALO_NARROW; want_ws = am & Y3_AM_IGEMM3_NARROW; }
            else if (op.ksize == 1 && op.in_c >= 1024 && op.out_c >= 128 && !(op.flags & Y3_F_OUT_F32))
              // (not below 64 tiles of 128 x 128: yolov3-tiny's 1024 -> 256 at 13^2 x 8 frames is 22 of them; the LDS-DMA
              // version with narrower channel tiles has four times the workgroups)
              want_ws = (am & Y3_AM_IGEMM3_1X1_DEEP) &&
                        ((am & Y3_AM_NO_SMALL_GRID) || (long long)y3_ceil_div(op.batch * op.out_h * op.out_w, 128) * y3_ceil_div(op.out_c, 128) >= y3_device_cus() / 4);
            if ((am & Y3_AM_1X1_DW) && y3_conv1x1_dw_pays(op))
              return y3_launch_conv1x1_dw(op, in, d_zero, s, name, dry_run, frag_w ? frag_w : op.d_weight_frag);
            // small grids (one frame at a time, small batches): 48-pixel tiles that fill the chip in one round (conv_dw48.hip)
            const bool small_dw = (am & (Y3_AM_SMALL_DW | Y3_AM_SMALL_DW_ALWAYS)) && !(am & Y3_AM_NO_SMALL_GRID);
            if (small_dw && (y3_conv_dw48_fits(op) || (op.ksize == 3 && y3_conv_dw48_fits_wide(op))))
              return y3_launch_conv_dw48(op, in, d_zero, s, name, dry_run, frag_w ? frag_w : op.d_weight_frag);
            if (!(am & Y3_AM_NO_WRES) && y3_conv1x1_wres_supported(op) && ((am & Y3_AM_WRES_ALWAYS) || y3_conv1x1_wres_pays(op)))
              return y3_launch_conv1x1_wres(op, in, d_zero, s, name, dry_run);
            // 1x1 layers on grids of a few rounds, where the weights-resident kernel does not pay
            if (small_dw && op.ksize == 1 && y3_conv_dw48_fits_wide(op))
              return y3_launch_conv_dw48(op, in, d_zero, s, name, dry_run, frag_w ? frag_w : op.d_weight_frag);
            if ((am & Y3_AM_IGEMM3_1X1_BM64) && op.ksize == 1 && op.in_c >= 256 && op.out_c >= 128 && !(op.flags & Y3_F_OUT_F32) &&
                y3_is16(op.dtype))
              return y3_launch_conv_igemm(op, in, d_zero, s, name, dry_run, 3, 3, 64);
            if ((am & Y3_AM_PATCH_WIDE) && op.ksize == 3 && op.stride == 1 && w > 128 && op.out_c >= 128 && y3_conv_patch_fits(op))
              return y3_launch_conv_patch(op, in, d_zero, s, name, dry_run);
            // Small grids (small maps x small batches): a halo tile is 192+ pixels x 128 channels, and below ~3/4 of a
            // tile per CU most of the chip idles through its long K loop; the 128 x 128 implicit GEMMs have more, shorter
            // workgroups.  tools/conv_bench.py at batch 1 / 4 / 8 (profiles/r02f_convbench_small_batches.txt): 512 -> 1024
            // at 19^2 x 8 frames (128 tiles) 615 TFLOP/s on the halo kernel, 722 on the wave-specialised implicit GEMM;
            // in float32 (yolov3-tiny's 13^2 layers at batch 8) 66 against 93 on the LDS-DMA implicit GEMM.
            // (Which kernel runs changes speed only: every MFMA conv kernel sums in the same K order.)
            const long long halo_tiles = (long long)y3_ceil_div(op.batch * op.in_h * op.in_w, 192) * (op.out_c / 128);
            const bool small_grid = !(am & Y3_AM_NO_SMALL_GRID) && k3 && halo_tiles < (3 * y3_device_cus()) / 4;
            // direct-weights strip kernel (192 x 256 tiles) where its tile count fills the chip better (csrc/conv_halo.hip)
            if (k3 && !small_grid && halo_ok && (((am & Y3_AM_HALO_DW) && y3_conv_halo_dw_pays(op)) ||
                                                 ((am & Y3_AM_HALO_DW_ALWAYS) && y3_conv_halo_dw_fits(op))))
              return y3_launch_conv_halo_dw(op, in, d_zero, s, name, dry_run, frag_w ? frag_w : op.d_weight_frag);
            if (small_grid && y3_is16(op.dtype)) return y3_launch_conv_igemm(op, in, d_zero, s, name, dry_run, 3, 3);
            if (want_halo && halo_ok && !small_grid) return y3_launch_conv_halo(op, in, d_zero, s, name, dry_run);
            if (want_ws) return y3_launch_conv_igemm(op, in, d_zero, s, name, dry_run, 3, 3);
            // 3x3 stride-2 layer with 256 input channels (76^2 -> 38^2): 865 against 762 TFLOP/s on the wave-specialised
            // implicit GEMM at batch 16; the other stride-2 layers measured faster on the LDS-DMA version
            // (64 input channels -- 304^2 -> 152^2 -- at one or two frames: 9.6 against 12.3 us at one frame, level at four, 109 against
            // 97 at sixteen: profiles/r06_conv_dw48.txt)
            if (!(am & Y3_AM_NO_SMALL_GRID) && op.ksize == 3 && op.stride == 2 && op.out_c >= 128 && y3_is16(op.dtype) && !(op.flags & Y3_F_OUT_F32) &&
                ((op.in_c == 256 && (long long)op.batch * op.out_h * op.out_w >= 16384) ||
                 (op.in_c == 64 && (long long)op.batch * op.out_h * op.out_w <= 49152)))
              return y3_launch_conv_igemm(op, in, d_zero, s, name, dry_run, 3, 3);
          }
          return y3_launch_conv_igemm(op, in, d_zero, s, name, dry_run);
        }
        case 1: return y3_launch_conv_small(op, in, s, name, dry_run);
        case 3: return y3_launch_conv_stem_mfma(op, in, s, name, dry_run);
        default: return y3_launch_conv_direct(op, in, s, name, dry_run);
      }
    }
    case Y3_OP_MAXPOOL: return y3_launch_maxpool(op, in, s, name, dry_run);
    case Y3_OP_UPSAMPLE: return y3_launch_upsample(op, in, s, name, dry_run);
    case Y3_OP_ADD: return y3_launch_add(op, in, s, name, dry_run);
    case Y3_OP_COPY: return y3_launch_copy(op, in, s, name, dry_run);
    case Y3_OP_YOLO: return y3_launch_yolo(op, in, s, name, dry_run);
    default: break;
  }
  y3_set_error("op for block %d: unknown kind %d", op.block_idx, op.kind);
  return Y3_ERR_INVALID;
}

// one op of a plan, honouring fusion decisions taken at plan creation
int run_op(y3_plan *plan, size_t i, const void *d_input, hipStream_t s, const char **name) {
  if (plan->fuse[i] == 2) return Y3_OK;
  if (plan->fuse[i] == 3) return y3_launch_conv_fused_resblock(plan->ops[i], plan->ops[i + 1], s, name, false);
  if (plan->fuse[i] == 6) return y3_launch_conv_block_fused(plan->ops[i], plan->ops[i + 1], s, name, false);
  if (plan->fuse[i] == 4) return y3_launch_conv_head_decode(plan->ops[i], plan->ops[i + 1], plan->d_zero, s, name, false, plan->frag_w[i]);
  if (plan->fuse[i] == 5) return y3_launch_maxpool_spp(plan->ops[i], plan->ops[i + 1], plan->ops[i + 2], s, name, false);
  if (plan->fuse[i] == 1) {
    const y3_op &op0 = plan->ops[i];
    const void *in = (op0.flags & Y3_F_PLAN_INPUT) ? d_input : op0.d_in;
    Y3_REQUIRE(in != nullptr, "op for block %d has no input pointer", op0.block_idx);
    return y3_launch_conv_fused_stem_s2(op0, plan->ops[i + 1], in, s, name, false);
  }
  return dispatch(plan->ops[i], d_input, plan->d_zero, s, name, false, plan->frag_w[i]);
}

// the kernels that read their weights from the fragment-order copy
bool uses_fragment_weights(const char *kernel) {
  return strncmp(kernel, "conv_halo_dw_", 13) == 0 || strncmp(kernel, "conv1x1_dw_", 11) == 0 || strncmp(kernel, "conv_dw48_", 10) == 0 ||
         strncmp(kernel, "conv_head_decode_dw_", 20) == 0;
}

// A private fragment-order copy of op i's weights (callers that pass no y3_op.d_weight_frag).  Made on the device that OWNS
// the weights -- not on whichever device happens to be current (a network on cuda:1 compiled while device 0 is
// current put the copy on GPU 0) -- and on a stream of its own, so that neither the legacy null stream's implicit
// synchronisation nor a stream capture in progress elsewhere in the process is touched.
int make_private_fragment_weights(y3_plan *p, size_t i) {
  const y3_op &op = p->ops[i];
  int prev = -1, owner = -1;
  Y3_HIP_CHECK(hipGetDevice(&prev));
  hipPointerAttribute_t attr;
  if (op.d_weight && hipPointerGetAttributes(&attr, op.d_weight) == hipSuccess) owner = attr.device;
  else (void)hipGetLastError();
  if (owner >= 0 && owner != prev) Y3_HIP_CHECK(hipSetDevice(owner));
  int rc = Y3_OK;
  void *w = nullptr;
  hipStream_t s = nullptr;
  do {
    if (hipMalloc(&w, y3_conv_halo_dw_weight_bytes(op)) != hipSuccess) {
      (void)hipGetLastError();
      y3_set_error("y3_plan_create: no memory for the fragment-order weights of block %d", op.block_idx);
      rc = Y3_ERR_HIP;
      break;
    }
    p->frag_w[i] = w;
    p->frag_own[i] = 1;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { s = nullptr; rc = Y3_ERR_HIP; }
    if (rc == Y3_OK) rc = y3_conv_halo_dw_make_weights(op, w, s);
    if (rc == Y3_OK && hipStreamSynchronize(s) != hipSuccess) rc = Y3_ERR_HIP;
    if (rc == Y3_ERR_HIP) y3_set_error("y3_plan_create: fragment-order weights of block %d: %s", op.block_idx, hipGetErrorString(hipGetLastError()));
  } while (0);
  if (s) (void)hipStreamDestroy(s);
  if (owner >= 0 && owner != prev) (void)hipSetDevice(prev);
  return rc;
}

}  // namespace

extern "C" {

int y3_abi_version(void) { return Y3_ABI_VERSION; }

const char *y3_last_error(void) { return g_err; }

int y3_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, i) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
  }
  return ok;
}

// which kernel family a conv op will use: 0 igemm, 1 stem, 2 direct (the host side lays the
// weights out accordingly before creating the plan)
int y3_conv_path(const y3_op *op) {
  if (!op || op->kind != Y3_OP_CONV) return -1;
  return conv_path(*op);
}

size_t y3_conv_fragment_weight_bytes(const y3_op *op, const y3_options *options) {
  if (!op || op->kind != Y3_OP_CONV || conv_path(*op) != 0) return 0;
  y3_options o = options ? *options : g_y3_defaults;
  OptScope scope(&o);
  const char *name = "";
  // (a detection-head conv is dispatched with the YOLO op behind it, at plan creation: asked about by its shape here)
  if (y3_opt().fuse_head && y3_conv_head_dw_fits(*op)) return y3_conv_halo_dw_weight_bytes(*op);
  if (dispatch(*op, nullptr, nullptr, nullptr, &name, true) != Y3_OK) return 0;
  return uses_fragment_weights(name) ? y3_conv_halo_dw_weight_bytes(*op) : 0;
}

int y3_conv_make_fragment_weights(const y3_op *op, void *d_dst, void *stream) {
  Y3_REQUIRE(op && d_dst && op->d_weight, "y3_conv_make_fragment_weights: bad arguments");
  Y3_REQUIRE(y3_conv_halo_dw_fits(*op) || y3_conv1x1_dw_pays(*op) || y3_conv_dw48_fits(*op) || y3_conv_dw48_fits_wide(*op) || y3_conv_head_dw_fits(*op),
             "conv block %d: not a layer of a direct-weights kernel", op->block_idx);
  return y3_conv_halo_dw_make_weights(*op, d_dst, static_cast<hipStream_t>(stream));
}

void y3_options_default(y3_options *options) {
  if (options) *options = g_y3_defaults;
}

int y3_plan_create(const y3_op *ops, int n_ops, const void *d_zero, y3_plan **out_plan) {
  return y3_plan_create_ex(ops, n_ops, d_zero, nullptr, out_plan);
}

int y3_plan_create_ex(const y3_op *ops, int n_ops, const void *d_zero, const y3_options *options, y3_plan **out_plan) {
  Y3_REQUIRE(ops && n_ops > 0 && out_plan, "y3_plan_create: bad arguments");
  Y3_REQUIRE(d_zero, "y3_plan_create: d_zero (zeroed device page) is required");
  y3_plan *p = new (std::nothrow) y3_plan;
  Y3_REQUIRE(p, "y3_plan_create: out of host memory");
  p->opt = options ? *options : g_y3_defaults;
  OptScope scope(&p->opt);
  p->ops.assign(ops, ops + n_ops);
  p->kernel.assign(n_ops, "");
  p->frag_w.assign(n_ops, nullptr);
  p->frag_own.assign(n_ops, 0);
  p->d_zero = d_zero;
  p->fuse.assign(n_ops, 0);
  for (int i = 0; i + 2 < n_ops; ++i)
    if (y3_opt().fuse_spp && y3_maxpool_spp_supported(p->ops[i], p->ops[i + 1], p->ops[i + 2])) {
      p->fuse[i] = 5;           // SPP pyramid: pool 5 / 9 / 13 of one tensor in one launch
      p->fuse[i + 1] = p->fuse[i + 2] = 2;
    }
  for (int i = 0; i + 1 < n_ops; ++i)
    if (p->fuse[i] != 0) continue;
    else if ((p->ops[i].flags & Y3_F_FUSE_NEXT) && p->ops[i].kind == Y3_OP_CONV &&
        y3_conv_fused_stem_s2_supported(p->ops[i], p->ops[i + 1])) {
      p->fuse[i] = 1;
      p->fuse[i + 1] = 2;
    } else if ((p->ops[i].flags & Y3_F_FUSE_NEXT) && y3_conv_fused_resblock_supported(p->ops[i], p->ops[i + 1])) {
      p->fuse[i] = 3;           // whole residual block in one kernel
      p->fuse[i + 1] = 2;
    } else if ((p->ops[i].flags & Y3_F_FUSE_NEXT) && y3_conv_block_fused_supported(p->ops[i], p->ops[i + 1])) {
      p->fuse[i] = 6;           // 1x1 -> 3x3 (+ shortcut) with the 128-channel tensor in LDS
      p->fuse[i + 1] = 2;
    } else if (y3_conv_head_decode_supported(p->ops[i], p->ops[i + 1])) {
      p->fuse[i] = 4;           // head conv + YOLO decode (the float32 logits never leave the CU)
      p->fuse[i + 1] = 2;
    }
  for (int i = 0; i < n_ops; ++i) {
    if (p->fuse[i] == 2) { p->kernel[i] = "(fused into the previous op)"; continue; }
    if (p->fuse[i] == 1) {
      (void)y3_launch_conv_fused_stem_s2(p->ops[i], p->ops[i + 1], nullptr, nullptr, &p->kernel[i], true);
      continue;
    }
    if (p->fuse[i] == 3) {
      (void)y3_launch_conv_fused_resblock(p->ops[i], p->ops[i + 1], nullptr, &p->kernel[i], true);
      continue;
    }
    if (p->fuse[i] == 6) {
      (void)y3_launch_conv_block_fused(p->ops[i], p->ops[i + 1], nullptr, &p->kernel[i], true);
      continue;
    }
    if (p->fuse[i] == 5) {
      (void)y3_launch_maxpool_spp(p->ops[i], p->ops[i + 1], p->ops[i + 2], nullptr, &p->kernel[i], true);
      continue;
    }
    const int rc = p->fuse[i] == 4 ? y3_launch_conv_head_decode(p->ops[i], p->ops[i + 1], d_zero, nullptr, &p->kernel[i], true)
                                   : dispatch(p->ops[i], nullptr, d_zero, nullptr, &p->kernel[i], true);
    if (rc != Y3_OK) {
      y3_plan_destroy(p);
      return rc;
    }
    if (uses_fragment_weights(p->kernel[i])) {
      // this kernel reads its weights in MFMA-fragment order: the caller's shared copy (y3_op.d_weight_frag, ABI 6), or --
      // callers that pass none -- a private copy that the plan makes here and frees when it is destroyed
      if (p->ops[i].d_weight_frag) { p->frag_w[i] = const_cast<void *>(p->ops[i].d_weight_frag); continue; }
      const int rc2 = make_private_fragment_weights(p, i);
      if (rc2 != Y3_OK) {
        y3_plan_destroy(p);
        return rc2;
      }
    }
  }
  *out_plan = p;
  return Y3_OK;
}

void y3_plan_destroy(y3_plan *plan) {
  if (!plan) return;
  for (hipEvent_t e : plan->events) (void)hipEventDestroy(e);
  for (hipEvent_t e : plan->kstart) (void)hipEventDestroy(e);
  for (hipEvent_t e : plan->kstop) (void)hipEventDestroy(e);
  for (auto &g : plan->graphs) (void)hipGraphExecDestroy(g.exec);
  for (size_t i = 0; i < plan->frag_w.size(); ++i)
    if (plan->frag_w[i] && plan->frag_own[i]) (void)hipFree(plan->frag_w[i]);
  delete plan;
}

static int plan_run_eager(y3_plan *plan, const void *d_input, hipStream_t s) {
  const char *name = nullptr;
  for (size_t i = 0; i < plan->ops.size(); ++i) {
    const int rc = run_op(plan, i, d_input, s, &name);
    if (rc != Y3_OK) return rc;
  }
  return Y3_OK;
}

int y3_plan_run(y3_plan *plan, const void *d_input, void *stream) {
  Y3_REQUIRE(plan, "y3_plan_run: null plan");
  hipStream_t s = static_cast<hipStream_t>(stream);
  OptScope scope(&plan->opt);
  // graphs need a capturable (non-default) stream; the plan's pointers and options never change, only d_input may
  if (!plan->opt.use_graph || !plan->graph_ok || s == nullptr || !plan->warmed) {
    plan->warmed = true;
    return plan_run_eager(plan, d_input, s);
  }
  for (auto &g : plan->graphs)
    if (g.input == d_input) {
      Y3_HIP_CHECK(hipGraphLaunch(g.exec, s));
      return Y3_OK;
    }
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    plan->graph_ok = false;
    return plan_run_eager(plan, d_input, s);
  }
  const int rc = plan_run_eager(plan, d_input, s);
  const hipError_t e_end = hipStreamEndCapture(s, &graph);
  if (rc != Y3_OK || e_end != hipSuccess || graph == nullptr ||
      hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
    (void)hipGetLastError();
    if (graph) (void)hipGraphDestroy(graph);
    plan->graph_ok = false;                       // capture not possible here: stay eager from now on
    return rc != Y3_OK ? rc : plan_run_eager(plan, d_input, s);
  }
  (void)hipGraphDestroy(graph);
  if (plan->graphs.size() >= 8) {                 // callers cycle through a few input buffers at most
    (void)hipGraphExecDestroy(plan->graphs.front().exec);
    plan->graphs.erase(plan->graphs.begin());
  }
  plan->graphs.push_back({d_input, exec});
  Y3_HIP_CHECK(hipGraphLaunch(exec, s));
  return Y3_OK;
}

int y3_plan_run_timed(y3_plan *plan, const void *d_input, void *stream, float *ms_per_op) {
  Y3_REQUIRE(plan && ms_per_op, "y3_plan_run_timed: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  OptScope scope(&plan->opt);
  const size_t n = plan->ops.size();
  while (plan->events.size() < n + 1) {
    hipEvent_t e;
    Y3_HIP_CHECK(hipEventCreate(&e));
    plan->events.push_back(e);
  }
  const char *name = nullptr;
  Y3_HIP_CHECK(hipEventRecord(plan->events[0], s));
  for (size_t i = 0; i < n; ++i) {
    const int rc = run_op(plan, i, d_input, s, &name);
    if (rc != Y3_OK) return rc;
    Y3_HIP_CHECK(hipEventRecord(plan->events[i + 1], s));
  }
  Y3_HIP_CHECK(hipEventSynchronize(plan->events[n]));
  for (size_t i = 0; i < n; ++i) Y3_HIP_CHECK(hipEventElapsedTime(&ms_per_op[i], plan->events[i], plan->events[i + 1]));
  return Y3_OK;
}

int y3_plan_run_profiled(y3_plan *plan, const void *d_input, void *stream, float *kernel_ms_per_op) {
  Y3_REQUIRE(plan && kernel_ms_per_op, "y3_plan_run_profiled: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  OptScope scope(&plan->opt);
  const size_t n = plan->ops.size(), cap = 2 * n;       // (an op is one launch today; room for two)
  while (plan->kstart.size() < cap) {
    hipEvent_t a, b;
    Y3_HIP_CHECK(hipEventCreate(&a));
    plan->kstart.push_back(a);
    Y3_HIP_CHECK(hipEventCreate(&b));
    plan->kstop.push_back(b);
  }
  Y3KernelTimer timer = {plan->kstart.data(), plan->kstop.data(), 0, (int)cap};
  std::vector<int> first(n + 1, 0);
  const char *name = nullptr;
  int rc = Y3_OK;
  tl_y3_timer = &timer;
  for (size_t i = 0; i < n && rc == Y3_OK; ++i) {
    first[i] = timer.n;
    rc = run_op(plan, i, d_input, s, &name);
  }
  first[n] = timer.n;
  tl_y3_timer = nullptr;
  if (rc != Y3_OK) return rc;
  Y3_HIP_CHECK(hipStreamSynchronize(s));
  for (size_t i = 0; i < n; ++i) {
    float sum = 0.f;
    for (int k = first[i]; k < first[i + 1]; ++k) {
      float ms = 0.f;
      Y3_HIP_CHECK(hipEventElapsedTime(&ms, plan->kstart[k], plan->kstop[k]));
      sum += ms;
    }
    kernel_ms_per_op[i] = sum;
  }
  return Y3_OK;
}

const char *y3_plan_op_kernel(const y3_plan *plan, int op_index) {
  if (!plan || op_index < 0 || (size_t)op_index >= plan->ops.size()) return "";
  return plan->kernel[op_index];
}

double y3_plan_op_flops(const y3_plan *plan, int op_index) {
  if (!plan || op_index < 0 || (size_t)op_index >= plan->ops.size()) return 0.0;
  if (plan->fuse[op_index] == 2) return 0.0;
  auto conv_flops = [](const y3_op &o) {
    return o.kind != Y3_OP_CONV ? 0.0 : 2.0 * o.ksize * o.ksize * o.in_c * (double)o.out_c * o.out_h * o.out_w * o.batch;
  };
  double f = conv_flops(plan->ops[op_index]);
  if (plan->fuse[op_index] == 1 || plan->fuse[op_index] == 3 || plan->fuse[op_index] == 6) f += conv_flops(plan->ops[op_index + 1]);   // (4: the decode has no conv FLOPs)
  return f;
}

double y3_plan_op_bytes(const y3_plan *plan, int op_index) {
  if (!plan || op_index < 0 || (size_t)op_index >= plan->ops.size()) return 0.0;
  if (plan->fuse[op_index] == 2) return 0.0;
  if (plan->fuse[op_index] == 1) {   // frames in, second conv's output out, both weight sets; nothing in between
    const y3_op &a = plan->ops[op_index], &b = plan->ops[op_index + 1];
    return (double)a.batch * a.in_h * a.in_w * a.in_c + (double)b.batch * b.out_h * b.out_w * b.out_c * 2.0 +
           27.0 * a.out_c * 2.0 + 9.0 * b.in_c * b.out_c * 2.0;
  }
  if (plan->fuse[op_index] == 4) {   // activations + weights in, 28 bytes per box out
    const y3_op &a = plan->ops[op_index], &b = plan->ops[op_index + 1];
    return ((double)a.batch * a.in_h * a.in_w * a.in_c + (double)a.in_c * a.out_c) * 2.0 +
           (double)b.batch * b.in_h * b.in_w * b.n_anchor * 28.0;
  }
  if (plan->fuse[op_index] == 5) {   // the tensor in once, three pooled tensors out
    const y3_op &a = plan->ops[op_index];
    return 4.0 * a.batch * a.in_h * a.in_w * a.in_c * y3_elem_size(a.dtype);
  }
  if (plan->fuse[op_index] == 3 || plan->fuse[op_index] == 6) {   // x in (once), z out, both weight sets
    const y3_op &a = plan->ops[op_index], &b = plan->ops[op_index + 1];
    return ((double)a.batch * a.in_h * a.in_w * a.in_c + (double)b.batch * b.out_h * b.out_w * b.out_c) * 2.0 +
           ((double)a.in_c * a.out_c + 9.0 * b.in_c * b.out_c) * 2.0;
  }
  const y3_op &o = plan->ops[op_index];
  const double es = y3_elem_size(o.dtype);
  const double in_px = (double)o.batch * o.in_h * o.in_w, out_px = (double)o.batch * o.out_h * o.out_w;
  switch (o.kind) {
    case Y3_OP_CONV: {
      double in_es = es;
      if (o.flags & Y3_F_IN_NCHW_F32) in_es = 4;
      if (o.flags & Y3_F_IN_NHWC_U8BGR) in_es = 1;
      const double out_es = (o.flags & Y3_F_OUT_F32) ? 4 : es;
      double b = in_px * o.in_c * in_es + out_px * o.out_c * out_es + (double)o.ksize * o.ksize * o.in_c * o.out_c * es;
      if (o.flags & Y3_F_RESIDUAL) b += out_px * o.out_c * es;
      return b;
    }
    case Y3_OP_ADD: return 3.0 * out_px * o.out_c * es;
    case Y3_OP_YOLO: return in_px * o.n_anchor * (o.n_attr * 4.0 + 28.0);
    default: return (in_px * o.in_c + out_px * o.out_c) * es;
  }
}

int y3_set_tuning(const char *key, int value) {
  if (!key) return Y3_ERR_INVALID;
  struct { const char *name; int32_t *field; } fields[] = {
      {"auto_mask", &g_y3_defaults.auto_mask},
      {"igemm_version", &g_y3_defaults.igemm_version}, {"igemm_ns", &g_y3_defaults.igemm_ns},
      {"igemm_bm", &g_y3_defaults.igemm_bm}, {"use_graph", &g_y3_defaults.use_graph},
      {"fuse_stem", &g_y3_defaults.fuse_stem}, {"fuse_head", &g_y3_defaults.fuse_head},
      {"fuse_spp", &g_y3_defaults.fuse_spp}, {"decode_lanes", &g_y3_defaults.decode_lanes},
      {"fuse_block", &g_y3_defaults.fuse_block}};
  for (auto &f : fields)
    if (!strcmp(key, f.name)) { *f.field = value; return Y3_OK; }
#ifdef Y3_HAS_DEBUG_KEY
  if (!strcmp(key, "debug")) { g_y3_debug = value; return Y3_OK; }
#endif
  y3_set_error("y3_set_tuning: unknown key %s", key);
  return Y3_ERR_INVALID;
}

int y3_op_run(const y3_op *op, const void *d_input, const void *d_zero, void *stream) {
  Y3_REQUIRE(op, "y3_op_run: null op");
  const char *name = nullptr;
  return dispatch(*op, d_input, d_zero, static_cast<hipStream_t>(stream), &name, false);
}

}  // extern "C"
