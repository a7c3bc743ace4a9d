#!/usr/bin/env python3
"""A/B of the fused 1x1 -> 3x3 (+ shortcut) kernel (csrc/conv_block.hip) against the two separate launches
(runs ON THE GPU BOX).  For every shape: a two-op plan through the C ABI with fuse_block = 2 / 0, bit-equality of the
outputs, HIP-event time per plan run (interleaved rounds, best of).
Usage: python tools/block_bench.py [--batch 16] [--iters 30] [--rounds 3]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "pytorch-yolov3_amd"))

import torch  # noqa: E402

from yolov3 import _hip  # noqa: E402

# name, H, Cin, Cout, shortcut
SHAPES = [
    ("s76_256-128-256_res", 76, 256, 256, True),
    ("s76_256-128-256", 76, 256, 256, False),
    ("s76_384-128-256", 76, 384, 256, False),
]


def round_up(v, m):
    return (v + m - 1) // m * m


def make_pair(dev, B, h, cin, cout, res, gen):
    """Tensors + two y3_op structs of a 1x1 (cin -> 128) + 3x3 (128 -> cout) pair."""
    t = {}
    t["x"] = (torch.randn((B, h, h, cin), generator=gen) * 0.7).to(torch.bfloat16).to(dev)
    t["mid"] = torch.zeros((B, h, h, 128), dtype=torch.bfloat16, device=dev)
    t["zero"] = torch.zeros(4096, dtype=torch.uint8, device=dev)
    ops = (_hip.Y3Op * 2)()
    for n, (ci, co, k) in enumerate(((cin, 128, 1), (128, cout, 3))):
        kk = k * k * ci
        k_ld = round_up(kk, 64)
        cp = round_up(co, 128)
        w = torch.zeros((cp, k_ld), dtype=torch.float32)
        w[:co, :kk] = torch.randn((co, kk), generator=gen) * (2.0 / kk) ** 0.5
        sc = torch.zeros(cp)
        bi = torch.zeros(cp)
        sc[:co] = torch.rand(co, generator=gen) + 0.5
        bi[:co] = torch.rand(co, generator=gen) - 0.5
        t["w%d" % n], t["sc%d" % n], t["bi%d" % n] = w.to(torch.bfloat16).to(dev), sc.to(dev), bi.to(dev)
        op = ops[n]
        op.kind, op.dtype, op.batch = _hip.OP_CONV, _hip.Y3_BF16, B
        op.ksize, op.stride, op.pad = k, 1, (k - 1) // 2
        op.in_c, op.out_c, op.in_ld, op.out_ld = ci, co, ci, co
        op.in_h = op.in_w = op.out_h = op.out_w = h
        op.cout_pad, op.k_ld = cp, k_ld
        op.d_weight, op.d_scale, op.d_bias = t["w%d" % n].data_ptr(), t["sc%d" % n].data_ptr(), t["bi%d" % n].data_ptr()
        op.flags = _hip.F_LEAKY
        op.block_idx = n
    ops[0].d_in, ops[0].d_out = t["x"].data_ptr(), t["mid"].data_ptr()
    ops[0].flags |= _hip.F_FUSE_NEXT
    ops[1].d_in = t["mid"].data_ptr()
    if res:
        ops[1].d_res, ops[1].res_ld = t["x"].data_ptr(), cin
        ops[1].flags |= _hip.F_RESIDUAL
    return t, ops


def make_plan(ops, zero, out, **options):
    lib = _hip.lib()
    ops[1].d_out = out.data_ptr()
    handle = ctypes.c_void_p()
    opt = _hip.options(**options)
    _hip.check(lib.y3_plan_create_ex(ops, 2, zero.data_ptr(), ctypes.byref(opt), ctypes.byref(handle)))
    return handle


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", default=None)
    ap.add_argument("--stamps", action="store_true", help="library built with `make stamps` (Y3_HIP_LIB): phase cycles of the fused kernel")
    args = ap.parse_args()
    lib = _hip.lib()
    _hip.require_gpu()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    stream = _hip.stream_ptr()
    for name, h, cin, cout, res in SHAPES:
        if args.only and args.only not in name:
            continue
        B = args.batch
        t, ops = make_pair(dev, B, h, cin, cout, res, gen)
        outs = [torch.zeros((B, h, h, cout), dtype=torch.bfloat16, device=dev) for _ in range(2)]
        plans = [make_plan(ops, t["zero"], outs[0], fuse_block=2), make_plan(ops, t["zero"], outs[1], fuse_block=0)]
        names = [[lib.y3_plan_op_kernel(pl, i).decode() for i in range(2)] for pl in plans]
        best = [1e9, 1e9]
        for _ in range(args.rounds):
            for vi, pl in enumerate(plans):
                for _ in range(3):
                    _hip.check(lib.y3_plan_run(pl, None, stream))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    _hip.check(lib.y3_plan_run(pl, None, stream))
                e1.record()
                torch.cuda.synchronize()
                best[vi] = min(best[vi], e0.elapsed_time(e1) / args.iters)
                if args.stamps and vi == 0:
                    buf = (ctypes.c_ulonglong * 8)()
                    lib.y3_debug_stamps_block(buf)
                    n = float(buf[7]) or 1.0
                    print("    [stamps] workgroups/launch %.0f  cycles/workgroup: wait-A %.0f  mma-A %.0f  mid-write %.0f  (unused %.0f)  "
                          "wait+barrier-B %.0f  mma-B %.0f  write-out %.0f" % (
                              n / (args.iters + 3), *[buf[i] / n for i in range(7)]))
        flops = 2.0 * (cin * 128 + 9 * 128 * cout) * h * h * B
        same = torch.equal(outs[0], outs[1])
        diff = float((outs[0].float() - outs[1].float()).abs().max())
        print("%-22s b%-3d fused %.4f ms %6.1f TF [%s] | separate %.4f ms %6.1f TF [%s + %s] | bit-identical %s (max diff %.3g)" % (
            name, B, best[0], flops / best[0] / 1e9, names[0][0], best[1], flops / best[1] / 1e9, names[1][0], names[1][1],
            same, diff), flush=True)
        for pl in plans:
            lib.y3_plan_destroy(pl)


if __name__ == "__main__":
    main()
