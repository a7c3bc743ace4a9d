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


def test_whole_network_with_the_fused_block_equals_the_two_launches():
    """inside a network plan the arena planner may give z the bytes of x for the pairs WITHOUT a shortcut
    (x dies after the 1x1: blocks 101 / 103 of yolov3@608), and a fused workgroup writing its rectangle of z would race with its
    neighbours still reading those pixels of x as their border.  ``y3_conv_block_fused_supported`` refuses such pairs (they run
    as two launches); the shortcut pairs are fused.  yolov3 608 bf16 at batch 32 -- 512 rectangles on 256 CUs, two rounds of
    workgroups, what makes the race possible -- with fuse_block = 2 must give the bits of fuse_block = 0, run after run."""
    import numpy as np
    import yolov3
    from yolov3 import weights as W
    from yolov3.synthdata import synth_frames
    from golden_util import MODELS
    frames = torch.from_numpy(synth_frames(4321, 32, 608, 608)).cuda()
    outs, fused_blocks = {}, {}
    for mode in (0, 2):
        net = yolov3.Darknet(MODELS["yolov3"], device="cuda", dtype="bf16", options={"fuse_block": mode}).eval()
        net.set_params(W.synth_params(net.blocks, net.net_info, seed=0, obj_bias=-8.5, calib=W.load_calibration("yolov3")))
        runs = []
        for _ in range(3):
            o = net.forward_frames(frames)
            torch.cuda.synchronize()
            runs.append({k: v.cpu() for k, v in o.items()})
        for k in runs[0]:
            assert torch.equal(runs[0][k], runs[1][k]) and torch.equal(runs[0][k], runs[2][k]), (mode, k)
        outs[mode] = runs[0]
        fused_blocks[mode] = [r["block"] for r in net.plan_report() if r["kernel"].startswith("conv_block_fused")]
        # no pair that runs fused may write over its own input
        cp = net._last_plan
        for n in range(cp.n_ops - 1):
            if int(cp.ops[n].block_idx) in fused_blocks[mode]:
                a, b = cp.ops[n], cp.ops[n + 1]
                x0, xb = a.d_in, a.batch * a.in_h * a.in_w * a.in_ld * 2
                z0, zb = b.d_out, b.batch * b.out_h * b.out_w * b.out_ld * 2
                assert not (x0 < z0 + zb and z0 < x0 + xb), "fused pair at block %d runs in place" % a.block_idx
        del net
    assert fused_blocks[0] == [] and len(fused_blocks[2]) >= 8, fused_blocks          # the eight 76^2 residual blocks at least
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[2][k]), k
    assert np.isfinite(outs[0]["class_prob"].numpy()).all()
