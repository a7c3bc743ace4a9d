"""The fused 1x1 -> 3x3 (+ shortcut) kernel of the 128-channel bottleneck stage (csrc/conv_block.hip; replaces
/root/reference/yolov3/darknet.py:244-257 twice and the shortcut at :376-379) against the two separate launches: same
operands, same K order (channel chunk outermost, tap innermost), same epilogue arithmetic -> the same bits.
Need an MI355X: -m gpu."""
import ctypes
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))

pytestmark = pytest.mark.gpu


def _run_pair(h, batch, cin, cout, res, seed):
    import block_bench as bb
    from yolov3 import _hip
    lib = _hip.lib()
    _hip.require_gpu()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(seed)
    t, ops = bb.make_pair(dev, batch, h, cin, cout, res, gen)
    outs, names = [], []
    for mode in (2, 0):
        out = torch.full((batch, h, h, cout), 7.0, dtype=torch.bfloat16, device=dev)
        plan = bb.make_plan(ops, t["zero"], out, fuse_block=mode)
        _hip.check(lib.y3_plan_run(plan, None, _hip.stream_ptr()))
        torch.cuda.synchronize()
        names.append(lib.y3_plan_op_kernel(plan, 0).decode())
        lib.y3_plan_destroy(plan)
        outs.append(out)
    return outs, names


# 76 = the real stage (19 x 19 rectangles, one per CU at batch 16); 20 / 33 / 50: rectangles that hang over the right and
# bottom edges, several tile rows; 384 input channels = the first pair of the 76^2 detection branch (after the route)
@pytest.mark.parametrize("h,batch,cin,cout,res", [
    (76, 2, 256, 256, True), (76, 1, 384, 256, False), (20, 2, 256, 256, True), (33, 1, 256, 256, False),
    (50, 1, 192, 128, False), (9, 3, 256, 256, True), (76, 16, 256, 256, True)])
def test_fused_bottleneck_block_is_bit_identical(h, batch, cin, cout, res):
    (fused, plain), (n_f, n_p) = _run_pair(h, batch, cin, cout, res, seed=h * 131 + cin)
    assert n_f == "conv_block_fused_bf16_x128", n_f
    assert not n_p.startswith("conv_block_fused"), n_p
    assert torch.isfinite(plain.float()).all()
    assert torch.equal(fused, plain), float((fused.float() - plain.float()).abs().max())


def test_fused_bottleneck_block_is_chosen_only_where_it_fills_the_chip():
    """fuse_block = 1: the pair is fused at 76^2 x 16 frames (256 rectangles on 256 CUs), not at batch 2; default options
    (fuse_block = 0: the kernel is level with the two launches, profiles/r04_block_fused_AB.txt): never."""
    import block_bench as bb
    from yolov3 import _hip
    lib = _hip.lib()
    dev = torch.device("cuda:0")
    for batch, want in ((16, True), (2, False)):
        t, ops = bb.make_pair(dev, batch, 76, 256, 256, True, torch.Generator().manual_seed(1))
        out = torch.zeros((batch, 76, 76, 256), dtype=torch.bfloat16, device=dev)
        for options, expect in ((dict(fuse_block=1), want), (dict(), False)):
            plan = bb.make_plan(ops, t["zero"], out, **options)
            name = lib.y3_plan_op_kernel(plan, 0).decode()
            lib.y3_plan_destroy(plan)
            assert name.startswith("conv_block_fused") == expect, (batch, options, name)
