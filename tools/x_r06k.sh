mkdir -p gpurun_out/r06k
L="s76_128-256_k3,s38_256-512_k3,s19_512-1024_k3,s38_512-256_k1,s19_1024-512_k1,s19_512-256_k1,s76_256-128_k1,s38_768,s76_384"
for b in 1 2 4 8; do
python tools/conv_bench.py --batch $b --only $L --variants igemm_v2,igemm_v3_ns3,halo_ws,dw48 > gpurun_out/r06k/cb_b$b.txt 2>&1
done
python tools/conv_bench.py --batch 1 --dtype fp16 --only s76_128-256_k3,s19_512-1024_k3,s19_1024-512_k1 --variants igemm_v2,dw48 > gpurun_out/r06k/cb_b1_f16.txt 2>&1
python - > gpurun_out/r06k/net_b1.txt 2>&1 <<'PY'
import sys, os, time
sys.path.insert(0, "pytorch-yolov3_amd")
import numpy as np, torch, yolov3
from yolov3 import weights as W, _hip
from yolov3.synthdata import synth_frames
for model in ("yolov3", "yolov3-spp", "yolov3-tiny"):
    cfg = "pytorch-yolov3_amd/models/%s.cfg" % model
    dim = 416 if model == "yolov3-tiny" else 608
    for dtype in ("bf16", "fp16"):
        for batch in (1, 2, 3, 5, 8):
            frames = synth_frames(7 + batch, batch, dim, dim)
            outs = []
            for mask in (_hip.AM_DEFAULT, _hip.AM_DEFAULT & ~_hip.AM_SMALL_DW):
                net = yolov3.Darknet(cfg, device="cuda:0", dtype=dtype, options={"auto_mask": mask}).eval()
                net.set_params(W.synth_params(net.blocks, net.net_info, seed=0, obj_bias=-5.0, calib=W.load_calibration(model)))
                o = {k: v.clone() for k, v in net.forward_frames(frames).items()}
                n = sum(r["kernel"].startswith("conv_dw48") for r in net.plan_report())
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(30): net.forward_frames(frames, fresh=False)
                torch.cuda.synchronize()
                outs.append((o, n, (time.perf_counter() - t0) / 30 * 1e3))
            same = all(torch.equal(outs[0][0][k], outs[1][0][k]) for k in outs[0][0])
            print("%-12s %s batch %d: dw48 ops %2d / %2d  forward %.3f ms vs %.3f ms  bit-equal %s" % (model, dtype, batch, outs[0][1], outs[1][1], outs[0][2], outs[1][2], same))
PY
