"""ctypes binding of libyolov3_hip.so (C ABI: include/yolov3_hip.h).

The library is hand-written HIP for gfx950 and is the ONLY compute backend of this
package: if it cannot be loaded, or no MI355X is visible, the entry points raise --
there is deliberately no CPU or PyTorch fallback.

``torch`` is imported first on purpose: PyTorch-ROCm ships its own libamdhip64.so.7
and must be the HIP runtime of the process, so that tensor ``data_ptr()`` addresses
and ``torch.cuda`` streams are valid inside this library (same runtime instance).
"""
import ctypes
import os

# HIP maps its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams that share a queue run one after
# the other: the pipeline's copy stream next to its three compute streams needs more (yolov3/pipeline.py;
# profiles/r03c_pcie_inclusive.txt: 4.9 k against 6.3 k frames/s).  Read by the runtime when it initialises, i.e. at the first
# GPU call of the process -- importing this package before that is enough.
_HW_QUEUES_PRESET = os.environ.get("GPU_MAX_HW_QUEUES")          # what the caller's environment said (None: nothing)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: F401,E402  (must be loaded before libyolov3_hip.so, see above)

# If the process had ALREADY initialised HIP when this module was imported (``import torch; torch.cuda.*`` first), the
# runtime has read its environment and the setdefault above came too late: the pipeline then runs on the default 4 queues
# (~4.9 k instead of ~6.3 k frames/s) with nothing to show for it.  ``hw_queues()`` says what is in effect, as far as that
# can be known; yolov3/pipeline.py warns, bench.py records it.
_HIP_UP_AT_IMPORT = bool(torch.cuda.is_initialized())


def hw_queues():
    """(value of GPU_MAX_HW_QUEUES the HIP runtime saw or will see, whether that is certain).  Not certain -- and probably the
    runtime's default of 4 -- when HIP was initialised before this package was imported without the variable set."""
    if _HIP_UP_AT_IMPORT and _HW_QUEUES_PRESET is None:
        return 4, False
    try:
        return int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), True
    except ValueError:
        return 4, False

_HERE = os.path.dirname(os.path.abspath(__file__))
# Y3_HIP_LIB: developer override (e.g. the diagnostic build with in-kernel phase stamps)
LIB_PATH = os.environ.get("Y3_HIP_LIB") or os.path.join(_HERE, "..", "lib", "libyolov3_hip.so")

Y3_F32, Y3_BF16, Y3_F16, Y3_F64 = 0, 1, 2, 3
OP_CONV, OP_MAXPOOL, OP_UPSAMPLE, OP_ADD, OP_COPY, OP_YOLO = 1, 2, 3, 4, 5, 6
F_LEAKY, F_RESIDUAL, F_OUT_F32, F_IN_NCHW_F32, F_IN_NHWC_U8BGR, F_PLAN_INPUT, F_FUSE_NEXT = 1, 2, 4, 8, 16, 32, 64
PATH_IGEMM, PATH_STEM, PATH_DIRECT, PATH_STEM_MFMA = 0, 1, 2, 3


class Y3Op(ctypes.Structure):
    """Mirror of ``struct y3_op`` (include/yolov3_hip.h)."""
    _fields_ = [
        ("kind", ctypes.c_int32), ("dtype", ctypes.c_int32), ("flags", ctypes.c_uint32),
        ("batch", ctypes.c_int32),
        ("in_h", ctypes.c_int32), ("in_w", ctypes.c_int32), ("in_c", ctypes.c_int32), ("in_ld", ctypes.c_int32),
        ("out_h", ctypes.c_int32), ("out_w", ctypes.c_int32), ("out_c", ctypes.c_int32), ("out_ld", ctypes.c_int32),
        ("ksize", ctypes.c_int32), ("stride", ctypes.c_int32), ("pad", ctypes.c_int32),
        ("res_ld", ctypes.c_int32), ("k_ld", ctypes.c_int32), ("cout_pad", ctypes.c_int32),
        ("d_in", ctypes.c_void_p), ("d_out", ctypes.c_void_p), ("d_res", ctypes.c_void_p),
        ("d_weight", ctypes.c_void_p), ("d_scale", ctypes.c_void_p), ("d_bias", ctypes.c_void_p),
        ("n_anchor", ctypes.c_int32), ("n_attr", ctypes.c_int32),
        ("anchor_w", ctypes.c_float * 8), ("anchor_h", ctypes.c_float * 8),
        ("row_offset", ctypes.c_int32), ("rows_total", ctypes.c_int32),
        ("net_w", ctypes.c_float), ("net_h", ctypes.c_float),
        ("d_bbox", ctypes.c_void_p), ("d_prob", ctypes.c_void_p), ("d_cls", ctypes.c_void_p),
        ("block_idx", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("d_weight_frag", ctypes.c_void_p),
    ]


class Y3Options(ctypes.Structure):
    """Mirror of ``struct y3_options`` (include/yolov3_hip.h): kernel-selection options of one plan."""
    _fields_ = [(name, ctypes.c_int32) for name in (
        "auto_mask", "unused0", "igemm_version", "igemm_ns", "igemm_bm", "use_graph", "fuse_stem", "fuse_head",
        "fuse_spp", "decode_lanes", "fuse_block")] + [("reserved", ctypes.c_int32 * 5)]


# y3_options.auto_mask bits (include/yolov3_hip.h: Y3_AM_*)
AM_HALO_WIDE, AM_IGEMM3_MID, AM_HALO_NARROW, AM_IGEMM3_1X1_DEEP = 0x0001, 0x0002, 0x0004, 0x0008
AM_HALO_MID, AM_IGEMM3_NARROW, AM_IGEMM3_1X1_BM64, AM_PATCH_WIDE = 0x0010, 0x0020, 0x0040, 0x0080
AM_HALO_TILE256, AM_NO_BN_SHRINK, AM_NO_SMALL_GRID, AM_NO_WRES, AM_WRES_ALWAYS = 0x0200, 0x0400, 0x0800, 0x1000, 0x2000
AM_HALO_DW, AM_HALO_DW_ALWAYS = 0x4000, 0x8000                       # direct-weights strip kernel: where it pays / wherever it fits
AM_1X1_DW = 0x10000                                                  # direct-weights 1x1 kernel (short-K bottleneck layers, one tile per CU)
AM_SMALL_DW_ALWAYS = 0x40000                                         # tests / A-B: the small-grid kernel wherever its shape constraints hold
AM_SMALL_DW = 0x20000                                                # small-grid direct-weights kernel (48-pixel tiles, 1x1 and 3x3: one frame at a time)
AM_SMALL_DW_WIDE = 0x80000                                           # ... on grids a little over one round as well (3x3: 1.5 rounds; 1x1 with one channel tile: 8)
AM_DEFAULT = (AM_HALO_WIDE | AM_HALO_NARROW | AM_IGEMM3_1X1_DEEP | AM_HALO_MID | AM_PATCH_WIDE | AM_HALO_DW | AM_1X1_DW | AM_SMALL_DW |
              AM_SMALL_DW_WIDE)
AM_IGEMM_ONLY = 0
AM_HALO_ALL = AM_HALO_WIDE | AM_HALO_NARROW | AM_HALO_MID          # conv_bench: the halo kernel wherever it fits

ABI_VERSION = 6
_lib = None

# name -> (restype, argtypes); every symbol include/yolov3_hip.h declares
PROTOTYPES = {
    "y3_abi_version": (ctypes.c_int, []),
    "y3_last_error": (ctypes.c_char_p, []),
    "y3_device_count": (ctypes.c_int, []),
    "y3_plan_create": (ctypes.c_int, [ctypes.POINTER(Y3Op), ctypes.c_int, ctypes.c_void_p,
                                      ctypes.POINTER(ctypes.c_void_p)]),
    "y3_plan_create_ex": (ctypes.c_int, [ctypes.POINTER(Y3Op), ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(Y3Options),
                                         ctypes.POINTER(ctypes.c_void_p)]),
    "y3_options_default": (None, [ctypes.POINTER(Y3Options)]),
    "y3_plan_destroy": (None, [ctypes.c_void_p]),
    "y3_plan_run": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "y3_plan_run_timed": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.POINTER(ctypes.c_float)]),
    "y3_plan_run_profiled": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.POINTER(ctypes.c_float)]),
    "y3_plan_op_kernel": (ctypes.c_char_p, [ctypes.c_void_p, ctypes.c_int]),
    "y3_plan_op_flops": (ctypes.c_double, [ctypes.c_void_p, ctypes.c_int]),
    "y3_plan_op_bytes": (ctypes.c_double, [ctypes.c_void_p, ctypes.c_int]),
    "y3_conv_path": (ctypes.c_int, [ctypes.POINTER(Y3Op)]),
    "y3_conv_fragment_weight_bytes": (ctypes.c_size_t, [ctypes.POINTER(Y3Op), ctypes.POINTER(Y3Options)]),
    "y3_conv_make_fragment_weights": (ctypes.c_int, [ctypes.POINTER(Y3Op), ctypes.c_void_p, ctypes.c_void_p]),
    "y3_set_tuning": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    "y3_op_run": (ctypes.c_int, [ctypes.POINTER(Y3Op), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "y3_detect_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "y3_detect": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_void_p, ctypes.c_float, ctypes.c_double, ctypes.c_void_p,
                                 ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "y3_nms_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "y3_nms": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double,
                              ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                              ctypes.c_void_p]),
    "y3_cxywh_to_tlbr": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_void_p]),
    "y3_nms_float_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "y3_nms_float": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double,
                                    ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "y3_cxywh_to_tlbr_float": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_void_p]),
    "y3_resize_bilinear_u8": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "y3_copy_bytes": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    "y3_pack_records": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
}


class HipLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raise if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.abspath(LIB_PATH)
    if not os.path.exists(path):
        raise HipLibraryError(
            "libyolov3_hip.so not found at {} -- build it with "
            "`make -C pytorch-yolov3_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "This package has no CPU fallback.".format(path))
    try:
        handle = ctypes.CDLL(path)
    except OSError as exc:
        raise HipLibraryError("cannot load {}: {}".format(path, exc))
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(handle, name)
        fn.restype = restype
        fn.argtypes = argtypes
    if handle.y3_abi_version() != ABI_VERSION:
        raise HipLibraryError("libyolov3_hip.so ABI version {} != {} (rebuild: make -C pytorch-yolov3_amd/csrc)".format(
            handle.y3_abi_version(), ABI_VERSION))
    _lib = handle
    return _lib


def options(**overrides):
    """The library's default plan options (as modified by ``y3_set_tuning``) with ``overrides`` applied."""
    opt = Y3Options()
    lib().y3_options_default(ctypes.byref(opt))
    for key, val in overrides.items():
        if key not in dict(Y3Options._fields_) or key in ("reserved", "unused0"):
            raise KeyError("unknown plan option {!r}".format(key))
        setattr(opt, key, int(val))
    return opt


def check(rc):
    if rc != 0:
        msg = lib().y3_last_error()
        raise RuntimeError("libyolov3_hip: {} (code {})".format(msg.decode() if msg else "unknown error", rc))


def require_gpu():
    """Raise unless a gfx950 GPU is usable by both torch and the library."""
    if not torch.cuda.is_available():
        raise RuntimeError(
            "no HIP device visible to torch: this package runs its hot path only on MI355X (gfx950); "
            "there is no CPU fallback")
    if lib().y3_device_count() < 1:
        raise RuntimeError("libyolov3_hip: no gfx950 device found")


def stream_ptr(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)
