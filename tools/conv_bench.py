#!/usr/bin/env python3
"""Micro-benchmark of the conv kernels on the Darknet-53 layer shapes (runs ON THE GPU BOX).

For every (layer shape, kernel variant) it launches the single op through the C ABI (y3_op_run),
times it (a one-op plan, y3_plan_run) with HIP events over many iterations on random data, and checks that all variants agree
with variant 0.  Interleaved rounds in one process (cdna_hip_programming.md rule 24).
Usage: python tools/conv_bench.py [--batch 16] [--dtype bf16] [--iters 20] [--out file]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "pytorch-yolov3_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from yolov3 import _hip  # noqa: E402

# name, H(in), Cin, Cout, k, stride, residual, out_f32
LAYERS = [
    ("s76_128-256_k3", 76, 128, 256, 3, 1, True, False),
    ("s38_256-512_k3", 38, 256, 512, 3, 1, True, False),
    ("s19_512-1024_k3", 19, 512, 1024, 3, 1, True, False),
    ("s152_64-128_k3", 152, 64, 128, 3, 1, True, False),
    ("s76_128-256_k3_nores", 76, 128, 256, 3, 1, False, False),
    ("s38_256-512_k3_b3", 38, 256, 512, 3, 1, True, False),
    ("s304_32-64_k3", 304, 32, 64, 3, 1, True, False),
    ("s608_32-64_k3s2", 608, 32, 64, 3, 2, False, False),
    ("s152_128-256_k3s2", 152, 128, 256, 3, 2, False, False),
    ("s76_256-512_k3s2", 76, 256, 512, 3, 2, False, False),
    ("s38_512-1024_k3s2", 38, 512, 1024, 3, 2, False, False),
    ("s304_64-128_k3s2", 304, 64, 128, 3, 2, False, False),
    ("s76_256-128_k1", 76, 256, 128, 1, 1, False, False),
    ("s38_512-256_k1", 38, 512, 256, 1, 1, False, False),
    ("s19_1024-512_k1", 19, 1024, 512, 1, 1, False, False),
    ("s152_128-64_k1", 152, 128, 64, 1, 1, False, False),
    ("s19_512-256_k1", 19, 512, 256, 1, 1, False, False),
    ("s38_768-256_k1", 38, 768, 256, 1, 1, False, False),
    ("s76_384-128_k1", 76, 384, 128, 1, 1, False, False),
    ("s76_256-255_k1_head", 76, 256, 255, 1, 1, False, True),
]

# knob sets (y3_set_tuning) compared per layer (auto_mask bits: include/yolov3_hip.h, Y3_AM_*)
AM = _hip
BASE = {"igemm_version": 2, "igemm_bm": 0, "igemm_ns": 2, "auto_mask": AM.AM_IGEMM_ONLY}   # no rerouting at all
VARIANTS = [
    ("igemm_v2", dict(BASE)),
    ("halo_ws", dict(BASE, auto_mask=AM.AM_HALO_ALL)),
    ("patch_8x32", dict(BASE, auto_mask=AM.AM_HALO_ALL | AM.AM_PATCH_WIDE)),
    ("igemm_v3_ns3", dict(BASE, igemm_version=3, igemm_ns=3)),
    ("igemm_v3_ns4", dict(BASE, igemm_version=3, igemm_ns=4)),
    ("igemm_v3_bm64_ns4", dict(BASE, igemm_version=3, igemm_ns=4, igemm_bm=64)),
    ("igemm_v3_bm64_ns3", dict(BASE, igemm_version=3, igemm_ns=3, igemm_bm=64)),
    ("halo_ws_256", dict(BASE, auto_mask=AM.AM_HALO_ALL | AM.AM_HALO_TILE256)),
    ("igemm_v2_bn128", dict(BASE, auto_mask=AM.AM_NO_BN_SHRINK)),
    ("halo_dw", dict(BASE, auto_mask=AM.AM_HALO_ALL | AM.AM_HALO_DW_ALWAYS)),   # direct-weights strip kernel (192 x 256 tiles) wherever it fits
    ("halo_dw_auto", dict(BASE, auto_mask=AM.AM_HALO_ALL | AM.AM_HALO_TILE256 | AM.AM_HALO_DW)),   # ... where the launcher's model says it pays
    ("dw_1x1", dict(BASE, auto_mask=AM.AM_1X1_DW)),                  # direct-weights 1x1 kernel (whole activation tile in LDS) where it takes the layer
    ("dw48", dict(BASE, auto_mask=AM.AM_SMALL_DW)),                  # small-grid direct-weights kernel (48-pixel tiles) where the layer fits one round
    ("dw48_always", dict(BASE, auto_mask=AM.AM_SMALL_DW_ALWAYS)),    # ... wherever its shape constraints hold, whatever the grid (A/B)
    ("wres_1x1", dict(BASE, auto_mask=AM.AM_WRES_ALWAYS)),           # weights-resident persistent 1x1 kernel wherever it is supported
]


def round_up(v, m):
    return (v + m - 1) // m * m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", default=None)
    ap.add_argument("--variants", default=None, help="comma-separated variant names (default: all)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--stamps", action="store_true", help="library built with `make stamps`: print phase cycles")
    args = ap.parse_args()
    global VARIANTS
    if args.variants:
        VARIANTS = [v for v in VARIANTS if v[0] in args.variants.split(",")]
    lib = _hip.lib()
    _hip.require_gpu()
    if os.environ.get("Y3_CONV_BENCH_DEBUG"):          # diagnostic libraries only (timing experiments)
        _hip.check(lib.y3_set_tuning(b"debug", int(os.environ["Y3_CONV_BENCH_DEBUG"])))
    dev = torch.device("cuda:0")
    bf = args.dtype in ("bf16", "fp16")                  # a 16-bit storage mode
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(args.dtype, torch.float32)
    es = 2 if bf else 4
    zero = torch.zeros(4096, dtype=torch.uint8, device=dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    results = []
    for name, h, cin, cout, k, s, res, f32out in LAYERS:
        if args.only and not any(o in name for o in args.only.split(",")):
            continue
        pad = (k - 1) // 2
        ho = (h + 2 * pad - k) // s + 1
        B = args.batch
        x = (torch.rand((B, h, h, cin), generator=g) * 2 - 1).to(tdt).to(dev)
        cout_pad = round_up(cout, 128)
        kk = k * k * cin
        k_ld = round_up(kk, 128 // es) + 128 // es      # one spare zero K-tile
        w = torch.zeros((cout_pad, k_ld), dtype=torch.float32)
        w[:cout, :kk] = (torch.rand((cout, kk), generator=g) * 2 - 1) * (2.0 / kk) ** 0.5
        w = w.to(tdt).to(dev)
        scale = torch.zeros(cout_pad, dtype=torch.float32)
        bias = torch.zeros(cout_pad, dtype=torch.float32)
        scale[:cout] = torch.rand(cout, generator=g) + 0.5
        bias[:cout] = torch.rand(cout, generator=g) - 0.5
        scale, bias = scale.to(dev), bias.to(dev)
        out_ld = round_up(cout, 8)
        r = (torch.rand((B, ho, ho, out_ld), generator=g) * 2 - 1).to(tdt).to(dev) if res else None
        outs = []
        for _ in VARIANTS:
            outs.append(torch.zeros((B, ho, ho, out_ld), dtype=torch.float32 if (f32out or not bf) else tdt, device=dev))
        op = _hip.Y3Op()
        op.kind = _hip.OP_CONV
        op.dtype = {"bf16": _hip.Y3_BF16, "fp16": _hip.Y3_F16}.get(args.dtype, _hip.Y3_F32)
        op.flags = _hip.F_LEAKY | (_hip.F_RESIDUAL if res else 0) | (_hip.F_OUT_F32 if (f32out and bf) else 0)
        op.batch = B
        op.in_h = op.in_w = h
        op.in_c = op.in_ld = cin
        op.out_h = op.out_w = ho
        op.out_c, op.out_ld = cout, out_ld
        op.ksize, op.stride, op.pad = k, s, pad
        op.res_ld = out_ld
        op.k_ld, op.cout_pad = k_ld, cout_pad
        op.d_in = x.data_ptr()
        op.d_res = r.data_ptr() if res else None
        op.d_weight, op.d_scale, op.d_bias = w.data_ptr(), scale.data_ptr(), bias.data_ptr()
        flops = 2.0 * kk * cout * ho * ho * B
        nbytes = (x.numel() + (r.numel() if res else 0)) * es + outs[0].numel() * outs[0].element_size() + kk * cout * es
        best = [1e9] * len(VARIANTS)
        kernel_of = [""] * len(VARIANTS)
        stream = _hip.stream_ptr()
        for rnd in range(args.rounds):
            for vi, (vname, knobs) in enumerate(VARIANTS):
                for key, val in knobs.items():
                    _hip.check(lib.y3_set_tuning(key.encode(), val))
                op.d_out = outs[vi].data_ptr()
                # a one-op plan: the product path (a plan makes its kernel's private weight layout once, at creation)
                handle = ctypes.c_void_p()
                _hip.check(lib.y3_plan_create((_hip.Y3Op * 1)(op), 1, zero.data_ptr(), ctypes.byref(handle)))
                kernel_of[vi] = lib.y3_plan_op_kernel(handle, 0).decode()
                for _ in range(2):
                    _hip.check(lib.y3_plan_run(handle, None, stream))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    _hip.check(lib.y3_plan_run(handle, None, stream))
                e1.record()
                torch.cuda.synchronize()
                lib.y3_plan_destroy(handle)
                best[vi] = min(best[vi], e0.elapsed_time(e1) / args.iters)
                if args.stamps and rnd == args.rounds - 1:
                    import ctypes as C
                    for rd in ("y3_debug_stamps_halo", "y3_debug_stamps_igemm"):
                        buf = (C.c_ulonglong * 8)()
                        getattr(lib, rd)(buf)
                        if buf[7]:
                            n = float(buf[7])
                            print("    [%s %s] blocks/launch %.0f  cycles/block slots 0-6: %s" % (
                                vname, rd[16:], n / (args.iters + 2), " ".join("%.0f" % (buf[i] / n) for i in range(7))))
        ref = outs[0].float()
        row = dict(layer=name, gflop=flops / 1e9)
        line = "%-22s %7.1f GF |" % (name, flops / 1e9)
        for vi, (vname, _) in enumerate(VARIANTS):
            diff = float((outs[vi].float() - ref).abs().max())
            tf = flops / best[vi] / 1e9
            row[vname] = dict(ms=best[vi], tflops=tf, gbps=nbytes / best[vi] / 1e6, maxdiff_vs_v0=diff, kernel=kernel_of[vi])
            line += " %s [%s] %.4f ms %6.1f TF %6.0f GB/s d=%.1e |" % (vname, kernel_of[vi].replace("conv_", ""), best[vi], tf,
                                                                    nbytes / best[vi] / 1e6, diff)
        print(line, flush=True)
        results.append(row)
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(results, fh, indent=1)


if __name__ == "__main__":
    main()
