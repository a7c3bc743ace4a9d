"""``Darknet``: the reference's model object, executed by HIP kernels on MI355X.

Keeps the surface callers use (/root/reference/yolov3/darknet.py:318-476):
``Darknet(config_fpath, device)``, ``.load_weights(path) -> self``, ``.eval()``,
``.cuda(device)``, ``.forward(x) -> {"bbox_xywh", "class_prob", "class_idx"}`` plus the
attributes ``.blocks .net_info .device .blocks_to_cache .header .modules_``.

What is different underneath: no nn.Module graph.  ``forward`` compiles (once per input
shape) a flat op plan (yolov3/plan.py) and hands it to libyolov3_hip.so through the C ABI
(include/yolov3_hip.h); activations are NHWC in an arena, BN is a per-channel scale/bias in
the conv epilogue, shortcuts/routes/head-concat are fused away.  Extension over the
reference: ``dtype="bf16"`` / ``dtype="fp16"`` (16-bit storage, fp32 accumulate: the same kernels on the
bf16 / f16 MFMA, same rate; fp16 keeps 11 significand bits instead of 8) and uint8 BGR frame input with
the BGR->RGB, /255 preprocessing fused into the first conv (``forward_frames``).
"""
import ctypes

import numpy as np
import torch

from . import _hip
from .cfgparse import parse_config
from .plan import build_plan
from .weights import conv_layout, read_darknet_weights

BN_EPS = np.float32(1e-5)


def _round_up(v, m):
    return (v + m - 1) // m * m


def f32_to_bf16_bits(a):
    """Round-to-nearest-even float32 -> bfloat16 bit patterns (uint16)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u >> np.uint64(16)) & np.uint64(1)) + np.uint64(0x7FFF)
    return ((u + r) >> np.uint64(16)).astype(np.uint16)


def f32_to_f16_bits(a):
    """Round-to-nearest-even float32 -> IEEE half bit patterns (uint16); subnormals kept, overflow -> inf (numpy's
    conversion, the same rounding as v_cvt_f16_f32 and torch's ``.half()``)."""
    with np.errstate(over="ignore"):
        return np.ascontiguousarray(a, dtype=np.float32).astype(np.float16).view(np.uint16)


# storage modes: name -> (C ABI element type, bytes per element, torch view dtype, host rounding to bit patterns)
DTYPES = {
    "float32": (_hip.Y3_F32, 4, torch.float32, None),
    "bf16": (_hip.Y3_BF16, 2, torch.bfloat16, f32_to_bf16_bits),
    "fp16": (_hip.Y3_F16, 2, torch.float16, f32_to_f16_bits),
}
DTYPE_ALIASES = {"float32": "float32", "fp32": "float32", "f32": "float32", "bf16": "bf16", "bfloat16": "bf16",
                 "fp16": "fp16", "f16": "fp16", "float16": "fp16", "half": "fp16"}


class BlockInfo(object):
    """Lightweight stand-in for the reference's per-block nn.Sequential (``modules_[i]``)."""

    def __init__(self, index, block):
        self.index = index
        self.type = block["type"]
        self.block = block

    def __repr__(self):
        return "BlockInfo({}, {})".format(self.index, self.type)


class _CompiledPlan(object):
    def __init__(self):
        self.handle = None
        self.arena = None
        self.keep = []
        self.ops = None
        self.n_ops = 0
        self.rows_total = 0
        self.batch = 0

    def destroy(self):
        if self.handle is not None:
            _hip.lib().y3_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Darknet(object):
    def __init__(self, config_fpath, device="cpu", dtype="float32", keep_all=False, fuse=None, options=None):
        """
        Args:
            config_fpath (str): Darknet .cfg file.
            device (str): "cpu" [default, like the reference: darknet.py:319] only builds the description;
                ``forward`` needs "cuda" / "cuda:N" (an MI355X): pass it here or call ``.cuda()`` as the
                reference's command line does (__main__.py:119-120).
            dtype (str): "float32" (parity path, exact fp32 MFMA), "bf16" (bf16 activations/weights, fp32
                accumulation; the benchmarked throughput path) or "fp16" (IEEE half storage, same kernels and rate,
                eight times finer rounding: the throughput path closest to the reference's float32 results).
        """
        self.blocks, self.net_info = parse_config(config_fpath)
        if self.net_info is None:
            raise ValueError("cfg {!r} has no [net] section".format(config_fpath))
        self.config_fpath = config_fpath
        self.keep_all = bool(keep_all)   # debugging: no arena reuse, so block_output() works
        # conv-pair fusion (stem + stride-2 conv, residual blocks): default on, off with keep_all unless asked for
        # (then the tensor between a fused pair is never written and block_output() of it is meaningless)
        self.fuse = (not self.keep_all) if fuse is None else bool(fuse)
        # kernel-selection options of this network's plans (include/yolov3_hip.h: y3_options), e.g.
        # {"auto_mask": 0}; None = the library's defaults at the time a plan is compiled
        self.options = dict(options) if options else None
        self.device = device
        self.header = None
        self.training = False
        self.dtype = DTYPE_ALIASES[str(dtype).replace("torch.", "")]

        # absolute route indices + cache set, as the reference computes them (darknet.py:334-349)
        self.blocks_to_cache = set()
        for i, blk in enumerate(self.blocks):
            if blk["type"] == "route":
                blk["layers"] = [j if j >= 0 else i + j for j in blk["layers"]]
                self.blocks_to_cache.update(blk["layers"])
            elif blk["type"] == "shortcut":
                self.blocks_to_cache.add(i - 1)
                self.blocks_to_cache.add(i + blk["from"])
        self._out_channels, self._convs = conv_layout(self.blocks, self.net_info)
        if self.blocks and self.blocks[0]["type"] != "convolutional":
            raise ValueError("the first block must be [convolutional] (it reads the network input)")
        self.modules_ = [BlockInfo(i, b) for i, b in enumerate(self.blocks)]
        self._params = None          # host copies, list of dicts per conv
        self._dev_weights = {}       # (slot, path) -> dict of device tensors
        self._plans = {}
        self._zero = None

    # ------------------------------------------------------------------ nn.Module-like surface
    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("inference-only implementation (BatchNorm uses running statistics)")
        return self.eval()

    def _move(self, device):
        """Device weights and compiled plans hold addresses on the old device: drop them when the device changes
        (they are rebuilt lazily by the next forward)."""
        if str(device) != str(self.device):
            self.__dict__.pop("_pipelines", None)
            self._dev_weights = {}
            for plan in self._plans.values():
                plan.destroy()
            self._plans = {}
            self._zero = None
        self.device = device
        return self

    def cuda(self, device=None):
        if device is None:
            return self._move("cuda")
        if isinstance(device, int):
            return self._move("cuda:%d" % device)
        return self._move(str(device))

    def to(self, device):
        return self.cuda(device) if str(device).startswith("cuda") else self._set_cpu()

    def cpu(self):
        return self._set_cpu()

    def _set_cpu(self):
        return self._move("cpu")

    def __call__(self, x):
        return self.forward(x)

    # ------------------------------------------------------------------ weights
    def load_weights(self, weights_path):
        """Read a Darknet ``.weights`` file (same stream order as darknet.py:415-476)."""
        self.header, params = read_darknet_weights(weights_path, self.blocks, self.net_info)
        return self.set_params(params)

    def set_params(self, params):
        """Install per-conv parameter dicts (see weights.read_darknet_weights)."""
        if len(params) != len(self._convs):
            raise ValueError("expected {} conv parameter sets, got {}".format(len(self._convs), len(params)))
        self._params = params
        self.__dict__.pop("_f16_range_checked", None)
        self.__dict__.pop("_pipelines", None)      # (their plans hold the old weights' addresses)
        self._dev_weights = {}
        for plan in self._plans.values():
            plan.destroy()
        self._plans = {}
        return self

    def _fold_bn(self, slot):
        """BN(eval) as y = conv*scale + bias, float32 like torch's CPU batch_norm
        (alpha = gamma / sqrt(var + eps), beta' = beta - mean * alpha)."""
        p = self._params[slot]
        cout = self._convs[slot]["cout"]
        if "bn_gamma" in p:
            inv_std = np.float32(1.0) / np.sqrt(p["bn_var"].astype(np.float32) + BN_EPS)
            scale = (p["bn_gamma"].astype(np.float32) * inv_std).astype(np.float32)
            bias = (p["bn_beta"].astype(np.float32) - p["bn_mean"].astype(np.float32) * scale).astype(np.float32)
        else:
            scale = np.ones(cout, dtype=np.float32)
            bias = p["bias"].astype(np.float32)
        return scale, bias

    def _device_weights(self, slot, path, dtype, dev):
        """Device copies of one conv's parameters in the layout of kernel family ``path``, weights rounded to the
        storage type ``dtype`` ("float32" / "bf16" / "fp16")."""
        dtype = DTYPE_ALIASES[{True: "bf16", False: "float32"}.get(dtype, dtype)]      # (round 1-4 callers passed a bool: bf16 or not)
        key = (slot, path, dtype, str(dev))
        es, to_bits = DTYPES[dtype][1], DTYPES[dtype][3]
        if key in self._dev_weights:
            return self._dev_weights[key]
        c = self._convs[slot]
        cout, cin, k = c["cout"], c["cin"], c["k"]
        w = self._params[slot]["weight"].astype(np.float32)
        scale, bias = self._fold_bn(slot)
        if path == _hip.PATH_STEM_MFMA:
            # 16-bit [32][32]: row = output channel, k = ky*9 + kx*3 + c_mem with c_mem the BYTE order of the
            # uint8 BGR frame (c_mem = 2 - c_rgb), zero padded (csrc/conv_small.hip: conv_stem_mfma_kernel)
            cout_pad = 32
            k_ld = 32
            host = np.zeros((32, 32), dtype=np.float32)
            wk = w[:, ::-1, :, :].transpose(0, 2, 3, 1).reshape(cout, 27)      # (co, ky, kx, c_mem)
            host[:cout, :27] = wk
            dw = torch.from_numpy(to_bits(host).view(np.int16)).to(dev)
        elif path == _hip.PATH_STEM:
            cout_pad = _round_up(cout, 8)
            k_ld = cout_pad
            host = np.zeros((k * k * cin, cout_pad), dtype=np.float32)
            host[:, :cout] = w.transpose(2, 3, 1, 0).reshape(k * k * cin, cout)
            dw = torch.from_numpy(host).to(dev)
        else:
            cout_pad = _round_up(cout, 128)
            k_ld = _round_up(k * k * cin, 128 // es)
            host = np.zeros((cout_pad, k_ld), dtype=np.float32)
            host[:cout, :k * k * cin] = w.transpose(0, 2, 3, 1).reshape(cout, k * k * cin)
            if to_bits is not None:
                dw = torch.from_numpy(to_bits(host).view(np.int16)).to(dev)
            else:
                dw = torch.from_numpy(host).to(dev)
        sc = np.zeros(cout_pad, dtype=np.float32)
        bi = np.zeros(cout_pad, dtype=np.float32)
        sc[:cout] = scale
        bi[:cout] = bias
        entry = dict(weight=dw, scale=torch.from_numpy(sc).to(dev), bias=torch.from_numpy(bi).to(dev),
                     cout_pad=cout_pad, k_ld=k_ld)
        self._dev_weights[key] = entry
        return entry

    # ------------------------------------------------------------------ plan compilation
    def _torch_device(self):
        if not str(self.device).startswith("cuda"):
            raise RuntimeError(
                "Darknet.forward runs only on an MI355X GPU (device={!r}); call .cuda() -- this build "
                "has no CPU execution path".format(self.device))
        _hip.require_gpu()
        dev = torch.device(self.device)
        return dev if dev.index is not None else torch.device("cuda", torch.cuda.current_device())

    def _fragment_weights(self, slot, dev, op, nbytes):
        """The conv's weights in MFMA-fragment order (direct-weights strip kernel), ONE copy per (layer, storage type, device)
        next to the other device layouts: every plan of the network shares it (until round 6 each plan made its own with
        hipMalloc, ~40 MB per plan for yolov3) and ``set_params`` invalidates it with them.  Made on torch's current stream."""
        key = (slot, "fragment", self.dtype, str(dev))
        frag = self._dev_weights.get(key)
        if frag is None:
            frag = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            _hip.check(_hip.lib().y3_conv_make_fragment_weights(ctypes.byref(op), frag.data_ptr(), _hip.stream_ptr()))
            self._dev_weights[key] = frag
            self._made_fragments = True
        return frag

    def _compile(self, batch, height, width, input_mode, options=None):
        # every allocation and launch of plan compilation happens on the NETWORK's device, whichever is current
        with torch.cuda.device(self._torch_device()):
            return self._compile_on_device(batch, height, width, input_mode, options)

    def _compile_on_device(self, batch, height, width, input_mode, options=None):
        if self._params is None:
            raise RuntimeError("call load_weights() / set_params() before forward()")
        dev = self._torch_device()
        lib = _hip.lib()
        options = options if options is not None else self.options
        opt = _hip.options(**options) if options else None
        self._made_fragments = False
        c_dtype, es = DTYPES[self.dtype][0], DTYPES[self.dtype][1]
        bf16 = es == 2                       # a 16-bit storage mode (bf16 or fp16)
        desc = build_plan(self.blocks, self.net_info, batch, height, width, es, reuse=not self.keep_all, fuse=self.fuse)
        cp = _CompiledPlan()
        cp.batch = batch
        cp.rows_total = desc["rows_total"]
        cp.arena = torch.empty(desc["arena_bytes"], dtype=torch.uint8, device=dev)
        # The zero page (source of padding taps and tile tails in the conv kernels) must outlive every plan that holds its
        # address: one per device index, and every plan keeps a reference.  (Until round 4 the test below compared
        # ``cuda:0`` with ``cuda``, so EVERY compile made a new page and dropped the old one under the plans compiled before:
        # once the allocator reused those 4 KiB, border pixels of the older plans read garbage.  Found by
        # tests/test_gpu_pipeline.py, which is the first test to check outputs of several plans of one network.)
        if self._zero is None or (self._zero.device.index or 0) != (dev.index if dev.index is not None else torch.cuda.current_device()):
            self._zero = torch.zeros(4096, dtype=torch.uint8, device=dev)
        cp.keep.append(self._zero)
        base = cp.arena.data_ptr()

        def addr(t, elem):
            if t is None:
                return None
            if t.buf == "input":
                return None
            return base + desc["offsets"][t.buf] + t.off * elem

        ops = (_hip.Y3Op * len(desc["ops"]))()
        for n, od in enumerate(desc["ops"]):
            op = ops[n]
            kind = od["kind"]
            op.dtype = c_dtype
            op.batch = batch
            op.block_idx = od["block"]
            tin = od["inp"]
            op.in_h, op.in_w, op.in_c, op.in_ld = tin.h, tin.w, tin.c, tin.ld
            in_es = 4 if tin.f32 else es
            if tin.buf == "input":
                op.flags |= _hip.F_PLAN_INPUT
                op.flags |= _hip.F_IN_NHWC_U8BGR if input_mode == "u8" else _hip.F_IN_NCHW_F32
            else:
                op.d_in = addr(tin, in_es)
            tout = od.get("out")
            if tout is not None:
                op.out_h, op.out_w, op.out_c, op.out_ld = tout.h, tout.w, tout.c, tout.ld
                op.d_out = addr(tout, 4 if tout.f32 else es)
                if tout.f32 and bf16:
                    op.flags |= _hip.F_OUT_F32
            res = od.get("res")
            if res is not None:
                op.d_res = addr(res, es)
                op.res_ld = res.ld
            if kind == "conv":
                op.kind = _hip.OP_CONV
                op.ksize, op.stride, op.pad = od["ksize"], od["stride"], od["pad"]
                if od["leaky"]:
                    op.flags |= _hip.F_LEAKY
                if od.get("fuse_next"):
                    op.flags |= _hip.F_FUSE_NEXT
                if res is not None:
                    op.flags |= _hip.F_RESIDUAL
                c = self._convs[od["slot"]]
                # provisional padded sizes so that y3_conv_path can judge the shape
                op.cout_pad = _round_up(c["cout"], 128)
                op.k_ld = _round_up(c["k"] * c["k"] * c["cin"], 128 // es)
                path = lib.y3_conv_path(ctypes.byref(op))     # (does not depend on the plan options)
                wts = self._device_weights(od["slot"], path, self.dtype, dev)
                op.cout_pad, op.k_ld = wts["cout_pad"], wts["k_ld"]
                op.d_weight = wts["weight"].data_ptr()
                op.d_scale = wts["scale"].data_ptr()
                op.d_bias = wts["bias"].data_ptr()
                cp.keep.append(wts)
                nfrag = lib.y3_conv_fragment_weight_bytes(ctypes.byref(op), ctypes.byref(opt) if opt is not None else None)
                if nfrag:
                    frag = self._fragment_weights(od["slot"], dev, op, nfrag)
                    op.d_weight_frag = frag.data_ptr()
                    cp.keep.append(frag)
            elif kind == "maxpool":
                op.kind = _hip.OP_MAXPOOL
                op.ksize, op.stride = od["ksize"], od["stride"]
            elif kind == "upsample":
                op.kind = _hip.OP_UPSAMPLE
                op.ksize, op.stride = 1, od["stride"]
            elif kind == "add":
                op.kind = _hip.OP_ADD
                op.ksize = op.stride = 1
            elif kind == "copy":
                op.kind = _hip.OP_COPY
                op.ksize = op.stride = 1
            elif kind == "yolo":
                op.kind = _hip.OP_YOLO
                op.n_anchor = len(od["anchors"])
                op.n_attr = od["n_attr"]
                for a, (aw, ah) in enumerate(od["anchors"]):
                    op.anchor_w[a] = float(aw)
                    op.anchor_h[a] = float(ah)
                op.row_offset, op.rows_total = od["row_offset"], od["rows_total"]
                op.net_w, op.net_h = float(self.net_info["width"]), float(self.net_info["height"])
            else:
                raise AssertionError(kind)
        cp.ops = ops
        cp.n_ops = len(desc["ops"])
        cp.desc = desc
        # persistent output buffers: every head's decode kernel writes its row range in place
        m = cp.rows_total
        cp.bbox = torch.empty((batch, m, 4), dtype=torch.float32, device=dev)
        cp.prob = torch.empty((batch, m), dtype=torch.float32, device=dev)
        cp.cls = torch.empty((batch, m), dtype=torch.int64, device=dev)
        for n in range(cp.n_ops):
            if ops[n].kind == _hip.OP_YOLO:
                ops[n].d_bbox, ops[n].d_prob, ops[n].d_cls = (
                    cp.bbox.data_ptr(), cp.prob.data_ptr(), cp.cls.data_ptr())
        handle = ctypes.c_void_p()
        if self._made_fragments:          # plans run on other streams than the one the copies were made on
            torch.cuda.current_stream().synchronize()
        _hip.check(lib.y3_plan_create_ex(ops, cp.n_ops, self._zero.data_ptr(),
                                         ctypes.byref(opt) if opt is not None else None, ctypes.byref(handle)))
        cp.handle = handle
        return cp

    def _get_plan(self, batch, height, width, input_mode, slot=0, options=None):
        # `slot` selects an independent arena + output buffers (one per in-flight batch when the caller
        # pipelines batches over several HIP streams); `options` (a dict, see __init__) override the network's
        # plan options for this plan only (yolov3/pipeline.py asks for the throughput tile choice)
        okey = tuple(sorted(options.items())) if options else None
        key = (batch, height, width, input_mode, self.dtype, str(self.device), slot, okey)
        cp = self._plans.get(key)
        if cp is None:
            cp = self._compile(batch, height, width, input_mode, options)
            self._plans[key] = cp
        return cp

    def _run(self, x, input_mode, timed=False, fresh=True, slot=0, options=None):
        """Launch the plan on torch's current stream.  ``fresh=False`` returns the plan's own
        output buffers (overwritten by the next call with the same shape) -- used by
        ``inference()`` and the benchmark, which consume them immediately."""
        dev = self._torch_device()
        lib = _hip.lib()
        if input_mode == "u8":
            batch, height, width, ch = x.shape
        else:
            batch, ch, height, width = x.shape
        if ch != self.net_info["channels"]:
            raise ValueError("input has {} channels, cfg says {}".format(ch, self.net_info["channels"]))
        cp = self._get_plan(batch, height, width, input_mode, slot, options)
        with torch.cuda.device(dev):
            if timed:
                # "kernel": events bound to every dispatch (the kernels' own begin -> end); True: events recorded on the
                # stream around every launch (includes dispatch handling) -- include/yolov3_hip.h
                ms = (ctypes.c_float * cp.n_ops)()
                run = lib.y3_plan_run_profiled if timed == "kernel" else lib.y3_plan_run_timed
                _hip.check(run(cp.handle, x.data_ptr(), _hip.stream_ptr(), ms))
                self.last_op_ms = list(ms)
            else:
                _hip.check(lib.y3_plan_run(cp.handle, x.data_ptr(), _hip.stream_ptr()))
        self._last_plan = cp
        if self.dtype == "fp16" and not timed and not self.__dict__.get("_f16_range_checked"):
            # IEEE half storage has no saturation: a stored activation above 65504 becomes inf and NaN downstream.
            # The procedural weights stay below 10 (profiles/r05_f16_overflow_audit.txt); real checkpoints are checked here, on
            # the FIRST forward after new parameters (one synchronising reduction, then never again): scores that are not finite
            # mean the half range was exceeded -- warn and point at bf16, which has float32's range.
            self._f16_range_checked = True
            if not bool(torch.isfinite(cp.prob).all()):     # (box sizes may be inf in any dtype: exp(tw) is unclamped, darknet.py:89-101)
                import warnings
                warnings.warn("dtype='fp16': non-finite outputs -- an activation exceeded the IEEE half range (65504); "
                              "use dtype='bf16' (float32 range) or 'float32' for these weights", RuntimeWarning)
        if fresh:
            return {"bbox_xywh": cp.bbox.clone(), "class_prob": cp.prob.clone(), "class_idx": cp.cls.clone()}
        return {"bbox_xywh": cp.bbox, "class_prob": cp.prob, "class_idx": cp.cls}

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        """x: (B,3,H,W) float32 RGB in [0,1] (torch tensor, any device).  Returns the reference's
        dict of (B,M,4) / (B,M) / (B,M) tensors on the GPU (darknet.py:401-405)."""
        dev = self._torch_device()
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(x)
        if x.dim() != 4:
            raise ValueError("expected a (B,C,H,W) tensor, got shape {}".format(tuple(x.shape)))
        x = x.to(device=dev, dtype=torch.float32).contiguous()
        return self._run(x, "f32")

    def forward_frames(self, frames_u8, fresh=True, slot=0, options=None):
        """frames_u8: (B,H,W,3) uint8 BGR (numpy or torch).  Same outputs as ``forward`` on
        ``flip(frames)/255`` transposed to NCHW (inference.py:332-333), preprocessing fused
        into the first conv kernel."""
        dev = self._torch_device()
        if not isinstance(frames_u8, torch.Tensor):
            frames_u8 = torch.from_numpy(np.ascontiguousarray(frames_u8))
        if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4:
            raise ValueError("expected uint8 (B,H,W,3) frames")
        return self._run(frames_u8.to(dev).contiguous(), "u8", fresh=fresh, slot=slot, options=options)

    def block_output(self, i):
        """(B,C,H,W) float32 copy of block i's output from the last forward.  Needs
        ``keep_all=True`` (otherwise the arena slot may have been reused)."""
        if not self.keep_all:
            raise RuntimeError("construct Darknet(..., keep_all=True) to inspect intermediate tensors")
        cp = self._last_plan
        t = cp.desc["tensor_of"][i]
        if t is None:
            raise ValueError("block {} is fused into its consumer and never materialised".format(i))
        es = 4 if t.f32 else DTYPES[self.dtype][1]
        start = cp.desc["offsets"][t.buf]
        nelem = cp.batch * t.h * t.w * t.ld
        raw = cp.arena[start:start + nelem * es]
        arr = raw.view(torch.float32 if es == 4 else DTYPES[self.dtype][2]).reshape(cp.batch, t.h, t.w, t.ld)
        return arr[:, :, :, t.off:t.off + t.c].permute(0, 3, 1, 2).float().contiguous()

    def plan_report(self):
        """Per-op (kernel name, flops, bytes, block) of the last executed plan (for bench/profiling)."""
        cp = self._last_plan
        lib = _hip.lib()
        return [dict(kernel=lib.y3_plan_op_kernel(cp.handle, i).decode(),
                     flops=lib.y3_plan_op_flops(cp.handle, i), bytes=lib.y3_plan_op_bytes(cp.handle, i),
                     block=int(cp.ops[i].block_idx)) for i in range(cp.n_ops)]
