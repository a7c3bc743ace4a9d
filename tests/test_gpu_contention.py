"""Every kernel family that overlaps its loads with counted waits (``s_waitcnt vmcnt(N)`` with loads still in flight: the
LDS-DMA rings of conv_halo_ws / conv_igemm2 / conv_igemm3 / conv1x1_wres / conv_patch_wsp / conv_block_fused and the
register-destination run-ahead loads of conv_halo_dw / conv1x1_dw / conv_dw48) run BESIDE a copy kernel that saturates the memory system, against
its own output when it runs alone.

Why: the waits are hand-counted.  Alone, loads land in issue order and early; under contention one can land late -- in the
direct-weights kernel that showed as a fault (round 5), in an LDS-DMA kernel it would be a SILENTLY wrong tile, which no
oracle comparison of a kernel running alone can see.  Each case: a single-op plan (two ops for the fused block) through the C
ABI, its output alone, then 20 x [y3_copy_bytes of 17.7 MB from pinned host memory on stream A, 6 launches on stream B], bit-equal
every fifth round.  These kernels replace /root/reference/yolov3/darknet.py:244-257 (+ the shortcut at :376-379).
Need an MI355X: -m gpu."""
import ctypes
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))

pytestmark = pytest.mark.gpu

TDT = {"bf16": torch.bfloat16, "fp16": torch.float16}


def _conv_op(dev, gen, dtype, B, h, cin, cout, k, stride, res):
    """One conv op on random data: (y3_op, tensors to keep alive, output tensor)."""
    from yolov3 import _hip
    tdt = TDT[dtype]
    pad = (k - 1) // 2
    ho = (h + 2 * pad - k) // stride + 1
    kk = k * k * cin
    k_ld = (kk + 63) // 64 * 64
    cp = (cout + 127) // 128 * 128
    keep = {}
    keep["x"] = (torch.rand((B, h, h, cin), generator=gen) - 0.5).to(tdt).to(dev)
    w = torch.zeros((cp, k_ld), dtype=torch.float32)
    w[:cout, :kk] = (torch.rand((cout, kk), generator=gen) - 0.5) * (6.0 / kk) ** 0.5
    keep["w"] = w.to(tdt).to(dev)
    keep["sc"] = (torch.rand(cp, generator=gen) + 0.5).to(dev)
    keep["bi"] = (torch.rand(cp, generator=gen) - 0.5).to(dev)
    out = torch.zeros((B, ho, ho, cout), dtype=tdt, device=dev)
    op = _hip.Y3Op()
    op.kind, op.dtype, op.flags = _hip.OP_CONV, {"bf16": _hip.Y3_BF16, "fp16": _hip.Y3_F16}[dtype], _hip.F_LEAKY
    op.batch, op.in_h, op.in_w, op.in_c, op.in_ld = B, h, h, cin, cin
    op.out_h, op.out_w, op.out_c, op.out_ld = ho, ho, cout, cout
    op.ksize, op.stride, op.pad, op.k_ld, op.cout_pad = k, stride, pad, k_ld, cp
    op.d_in, op.d_out = keep["x"].data_ptr(), out.data_ptr()
    op.d_weight, op.d_scale, op.d_bias = keep["w"].data_ptr(), keep["sc"].data_ptr(), keep["bi"].data_ptr()
    if res:
        keep["r"] = (torch.rand((B, ho, ho, cout), generator=gen) - 0.5).to(tdt).to(dev)
        op.d_res, op.res_ld = keep["r"].data_ptr(), cout
        op.flags |= _hip.F_RESIDUAL
    return op, keep, out


def _beside_a_copy(handle, out, rounds=20, launches=6, more=()):
    """`out` (and the tensors in `more`) after launches beside the copy kernel == after a launch alone."""
    from yolov3 import _hip
    lib = _hip.lib()
    dev = out.device
    _hip.check(lib.y3_plan_run(handle, None, None))
    torch.cuda.synchronize()
    alone = out.clone()
    alone_more = [t.clone() for t in more]
    assert torch.isfinite(alone.float()).all() and float(alone.float().abs().max()) > 0
    out.zero_()
    for t in more:
        t.zero_()
    host = torch.zeros(16 * 608 * 608 * 3, dtype=torch.uint8).pin_memory()
    dst = torch.zeros(16 * 608 * 608 * 3, dtype=torch.uint8, device=dev)
    cs, st = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for i in range(rounds):
        _hip.check(lib.y3_copy_bytes(host.data_ptr(), dst.data_ptr(), host.numel(), 8, _hip.stream_ptr(cs)))
        for _ in range(launches):
            _hip.check(lib.y3_plan_run(handle, None, _hip.stream_ptr(st)))
        if i % 5 == 4:
            torch.cuda.synchronize()
            assert torch.equal(out, alone), "round %d: %d values differ from the kernel's own quiet output" % (
                i, int((out != alone).sum()))
            for t, a in zip(more, alone_more):
                assert torch.equal(t, a), "round %d: a second output differs from the kernel's own quiet output" % i
    torch.cuda.synchronize()


def _am():
    from yolov3 import _hip
    return _hip


# (id, expected kernel-name prefix incl. tile, options, B, h, cin, cout, k, stride, shortcut)
def _cases():
    from yolov3 import _hip as H
    halo = H.AM_HALO_ALL | H.AM_NO_SMALL_GRID
    return [
        # wave-specialised strip kernel: 256-pixel tiles with the 4-slot ring (19^2, 38^2) and the 3-slot ring (76^2: the halo
        # leaves no room for a fourth slot), 192-pixel tiles (the latency choice at 19^2 / 38^2)
        ("ws256_ring4_19", "conv_halo_ws_%s_256x128", dict(auto_mask=halo | H.AM_HALO_TILE256), 16, 19, 512, 1024, 3, 1, True),
        ("ws256_ring3_76", "conv_halo_ws_%s_256x128", dict(auto_mask=halo | H.AM_HALO_TILE256), 16, 76, 128, 256, 3, 1, True),
        ("ws192_38", "conv_halo_ws_%s_192x128", dict(auto_mask=halo), 16, 38, 256, 512, 3, 1, True),
        ("ws192_19_nores", "conv_halo_ws_%s_192x128", dict(auto_mask=halo), 16, 19, 512, 1024, 3, 1, False),
        # LDS-DMA double-buffered implicit GEMM: stride-2 3x3 (kmode: nine taps), 1x1, 96 x 64 tiles
        ("igemm2_s2", "conv_igemm2_%s_128x128", dict(auto_mask=0), 16, 76, 256, 512, 3, 2, False),
        ("igemm2_1x1_38", "conv_igemm2_%s_128x128", dict(auto_mask=0), 16, 38, 512, 256, 1, 1, False),
        ("igemm2_96x64", "conv_igemm2_%s_96x64", dict(auto_mask=0, igemm_bm=96), 8, 26, 384, 256, 1, 1, False),
        # wave-specialised implicit GEMM, 3 and 4 LDS stages
        ("igemm3_ns3_1x1_19", "conv_igemm3_%s_128x128", dict(auto_mask=0, igemm_version=3, igemm_ns=3), 16, 19, 1024, 512, 1, 1, False),
        ("igemm3_ns4_s2", "conv_igemm3_%s_128x128", dict(auto_mask=0, igemm_version=3, igemm_ns=4), 16, 76, 256, 512, 3, 2, False),
        # weights-resident persistent 1x1 (4-slot pixel ring)
        ("wres_76", "conv1x1_wres_%s_128x128", dict(auto_mask=H.AM_WRES_ALWAYS), 16, 76, 256, 128, 1, 1, False),
        ("wres_152", "conv1x1_wres_%s_128x64", dict(auto_mask=H.AM_WRES_ALWAYS), 16, 152, 128, 64, 1, 1, False),
        # direct-weights 1x1 kernel (round 6): register-destination run-ahead weight loads, 4 K-steps deep, behind the tile's LDS-DMA
        ("dw1x1_96", "conv1x1_dw_%s_96x256", dict(auto_mask=H.AM_1X1_DW), 16, 38, 512, 256, 1, 1, False),
        ("dw1x1_48", "conv1x1_dw_%s_48x256", dict(auto_mask=H.AM_1X1_DW), 16, 19, 1024, 512, 1, 1, False),
        ("dw1x1_96_k768", "conv1x1_dw_%s_96x256", dict(auto_mask=H.AM_1X1_DW), 16, 38, 768, 256, 1, 1, False),
        # small-grid direct-weights kernel (round 6): 1 .. 8 waves per workgroup, run-ahead weight loads behind the halo's LDS-DMA
        ("dw48_k3_76", "conv_dw48_k3_%s", dict(auto_mask=H.AM_SMALL_DW), 1, 76, 128, 256, 3, 1, True),
        ("dw48_k3_19", "conv_dw48_k3_%s", dict(auto_mask=H.AM_SMALL_DW), 2, 19, 512, 1024, 3, 1, True),
        ("dw48_k1_38", "conv_dw48_k1_%s", dict(auto_mask=H.AM_SMALL_DW), 1, 38, 512, 256, 1, 1, False),
        # ... its stride-2 form (four parity planes; 512 channels as two LDS images, re-staged by helper and computing waves in the K loop)
        ("dw48_k3s2_76", "conv_dw48_k3s2_%s", dict(auto_mask=H.AM_SMALL_DW), 1, 76, 256, 512, 3, 2, False),
        ("dw48_k3s2_38", "conv_dw48_k3s2_%s", dict(auto_mask=H.AM_SMALL_DW), 2, 38, 512, 1024, 3, 2, False),
        # persistent 2-D patch kernel (rows wider than 128 px)
        ("patch_152", "conv_patch_wsp_%s_8x32x128", dict(auto_mask=halo | H.AM_PATCH_WIDE), 16, 152, 64, 128, 3, 1, True),
    ]


def _case_ids():
    try:
        return [c[0] for c in _cases()]
    except Exception:           # the library is missing: collection must not fail on a CPU box (the tests are -m gpu)
        return []


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", _case_ids())
def test_counted_wait_kernel_beside_a_copy_kernel(case, dtype):
    from yolov3 import _hip
    lib = _hip.lib()
    _hip.require_gpu()
    name, kernel, options, B, h, cin, cout, k, stride, res = next(c for c in _cases() if c[0] == case)
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(1000 + h + cin)
    op, keep, out = _conv_op(dev, gen, dtype, B, h, cin, cout, k, stride, res)
    zero = torch.zeros(4096, dtype=torch.uint8, device=dev)
    opts = _hip.options(**options)
    handle = ctypes.c_void_p()
    _hip.check(lib.y3_plan_create_ex((_hip.Y3Op * 1)(op), 1, zero.data_ptr(), ctypes.byref(opts), ctypes.byref(handle)))
    try:
        want = kernel % {"bf16": "bf16", "fp16": "f16"}[dtype]
        assert lib.y3_plan_op_kernel(handle, 0).decode() == want, (lib.y3_plan_op_kernel(handle, 0).decode(), want)
        _beside_a_copy(handle, out)
    finally:
        lib.y3_plan_destroy(handle)


@pytest.mark.parametrize("res", [True, False])
def test_fused_bottleneck_block_beside_a_copy_kernel(res):
    """conv_block_fused (1x1 -> 3x3 + shortcut, bottleneck tensor in LDS; bf16 only, opt-in): the same check."""
    import block_bench as bb
    from yolov3 import _hip
    lib = _hip.lib()
    _hip.require_gpu()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(7)
    t, ops = bb.make_pair(dev, 16, 76, 256, 256, res, gen)
    out = torch.zeros((16, 76, 76, 256), dtype=torch.bfloat16, device=dev)
    handle = bb.make_plan(ops, t["zero"], out, fuse_block=2)
    try:
        assert lib.y3_plan_op_kernel(handle, 0).decode() == "conv_block_fused_bf16_x128"
        _beside_a_copy(handle, out)
    finally:
        lib.y3_plan_destroy(handle)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("mode,kernel,B,h,cin", [
    (2, "conv_head_decode_%s_64x256", 16, 38, 512),            # tiled head kernel: LDS-DMA double buffer
    (4, "conv_head_decode_dw_%s_96x256", 16, 38, 512),         # direct-weights head kernel (round 6): run-ahead weight loads, 96-pixel tiles
    (1, "conv_head_decode_dw_%s_48x256", 16, 19, 1024),        # ... 48-pixel tiles (the default), sixteen K-steps
    (1, "conv_head_decode_dw_%s_48x256", 1, 76, 256)])         # ... one frame
def test_head_kernels_beside_a_copy_kernel(mode, kernel, B, h, cin, dtype):
    """The fused detection-head kernels (1x1 conv + YOLO decode in one launch; /root/reference/yolov3/darknet.py:244-257 + :86-116):
    boxes, scores and classes beside a copy kernel == alone."""
    from yolov3 import _hip
    lib = _hip.lib()
    _hip.require_gpu()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(50 + h)
    op, keep, _ = _conv_op(dev, gen, dtype, B, h, cin, 255, 1, 1, False)
    op.flags = _hip.F_OUT_F32                                   # a head conv: bias only, float32 logits
    logits = torch.zeros((B, h, h, 256), dtype=torch.float32, device=dev)
    op.d_out, op.out_ld = logits.data_ptr(), 256
    rows = 3 * h * h
    bbox = torch.zeros((B, rows, 4), dtype=torch.float32, device=dev)
    prob = torch.zeros((B, rows), dtype=torch.float32, device=dev)
    cls = torch.zeros((B, rows), dtype=torch.int64, device=dev)
    yo = _hip.Y3Op()
    yo.kind, yo.dtype = _hip.OP_YOLO, op.dtype
    yo.batch, yo.in_h, yo.in_w, yo.in_c, yo.in_ld = B, h, h, 255, 256
    yo.out_h, yo.out_w = h, h
    yo.d_in = logits.data_ptr()
    yo.n_anchor, yo.n_attr = 3, 85
    for a, (aw, ah) in enumerate(((116, 90), (156, 198), (373, 326))):
        yo.anchor_w[a], yo.anchor_h[a] = float(aw), float(ah)
    yo.row_offset, yo.rows_total = 0, rows
    yo.net_w = yo.net_h = 608.0
    yo.d_bbox, yo.d_prob, yo.d_cls = bbox.data_ptr(), prob.data_ptr(), cls.data_ptr()
    zero = torch.zeros(4096, dtype=torch.uint8, device=dev)
    opts = _hip.options(fuse_head=mode)
    handle = ctypes.c_void_p()
    _hip.check(lib.y3_plan_create_ex((_hip.Y3Op * 2)(op, yo), 2, zero.data_ptr(), ctypes.byref(opts), ctypes.byref(handle)))
    try:
        want = kernel % {"bf16": "bf16", "fp16": "f16"}[dtype]
        assert lib.y3_plan_op_kernel(handle, 0).decode() == want, (lib.y3_plan_op_kernel(handle, 0).decode(), want)
        _beside_a_copy(handle, bbox, more=(prob, cls))
    finally:
        lib.y3_plan_destroy(handle)
