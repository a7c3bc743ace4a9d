"""Callers of the hot path: COCO export, id matching, box drawing, file
listing, command line.  CPU tests pin the host logic against vectors produced by the reference
(tests/golden/coco_export.json, tools/make_goldens.py g9); the GPU tests check that the batched
loops return exactly what per-frame ``inference()`` returns."""
import copy
import json
import os

import numpy as np
import pytest

import yolov3
from yolov3 import stream
from yolov3.__main__ import build_parser, main as cli_main
from yolov3.devtools import coco_util

from yolov3.synthdata import synth_frames

from golden_util import GOLDEN, MODELS, golden_weights_path, load_jpeg_bgr


def _coco_case():
    with open(os.path.join(GOLDEN, "coco_export.json")) as fh:
        g = json.load(fh)
    output = [[np.array(b, dtype=np.int64).reshape(-1, 4), np.array(p, dtype=np.uint32).view(np.float32),
               np.array(c, dtype=np.int64)] for b, p, c in g["inference_output"]]
    return g, output


def test_to_coco_matches_reference():
    g, output = _coco_case()
    got = yolov3.to_coco(g["image_filenames"], output, g["class_names"])
    assert got == g["to_coco"]
    json.dumps(got)   # plain Python numbers only


def test_match_ids_matches_reference():
    g, output = _coco_case()
    dataset = yolov3.to_coco(g["image_filenames"], output, g["class_names"])
    coco_util.match_ids(dataset, copy.deepcopy(g["reference_dataset"]))
    assert dataset == g["match_ids"]
    with pytest.raises(KeyError):
        bad = copy.deepcopy(g["reference_dataset"])
        bad["images"].append({"file_name": "missing.jpg", "id": 1, "height": 1, "width": 1})
        coco_util.match_ids(yolov3.to_coco(g["image_filenames"], output, g["class_names"]), bad)


def test_draw_boxes_outlines_and_clipping():
    img = np.zeros((40, 60, 3), dtype=np.uint8)
    boxes = np.array([[10, 8, 30, 20], [-5, -5, 70, 50]], dtype=np.int64)      # second one leaves the image
    out = yolov3.draw_boxes(img, boxes)
    assert out is img
    assert (img[8, 10:31] == (0, 255, 0)).all() and (img[20, 10:31] == (0, 255, 0)).all()
    assert (img[8:21, 10] == (0, 255, 0)).all() and (img[8:21, 30] == (0, 255, 0)).all()
    assert (img[14, 20] == 0).all()                                            # interior untouched
    # per-class colours + labels: no exception, label patch drawn inside the image
    img2 = np.zeros((64, 64, 3), dtype=np.uint8)
    yolov3.draw_boxes(img2, boxes[:1], class_prob=np.array([0.5]), class_idx=np.array([1]), class_names=["a", "b"])
    assert img2.any()
    assert len(set(stream.unique_colors(5))) == 5


def test_list_image_files_sorted_and_filtered(tmp_path):
    for name in ("b.jpg", "a.png", "notes.txt", "c.JPG"):
        (tmp_path / name).write_bytes(b"x")
    directory, names = stream.list_image_files(tmp_path)
    assert directory == str(tmp_path) and names == ["a.png", "b.jpg", "c.JPG"]
    directory, names = stream.list_image_files(tmp_path / "b.jpg")
    assert directory == str(tmp_path) and names == ["b.jpg"]


def test_cli_flags_mirror_the_reference():
    p = build_parser()
    a = p.parse_args(["-V", "v.mp4", "-c", "m.cfg", "-w", "m.weights", "-i", "0.4", "-p", "0.2", "-n", "names.txt",
                      "-o", "out", "--show-fps", "-v"])
    assert a.iou_thresh == 0.4 and a.prob_thresh == 0.2 and a.device == "cuda" and a.show_fps and a.verbose
    assert p.parse_args(["-C", "-c", "m.cfg", "-w", "w"]).cam == 0           # bare -C: device 0
    with pytest.raises(SystemExit):
        p.parse_args(["-c", "m.cfg", "-w", "w"])                              # an input source is required
    with pytest.raises(SystemExit):
        p.parse_args(["-I", "a", "-V", "b", "-c", "m.cfg", "-w", "w"])        # ... and exactly one
    with pytest.raises(SystemExit):
        cli_main(["-I", "a", "-c", "m.cfg", "-w", "w", "-d", "cpu"])          # no CPU path: refused loudly


def test_video_and_camera_need_opencv_and_say_so(tmp_path):
    if stream._cv2() is not None:
        pytest.skip("OpenCV present")
    clip = tmp_path / "clip.mp4"
    clip.write_bytes(b"\x00\x00\x00\x18ftypmp42" + b"\x00" * 64)
    with pytest.raises(ValueError, match="OpenCV"):         # an .mp4 needs a codec library; .avi (MJPEG) / .y4m do not
        list(stream._video_frames(str(clip)))
    with pytest.raises(RuntimeError, match="OpenCV"):
        stream.detect_in_cam(None)


# ---------------------------------------------------------------------------------------------- GPU
IMAGES = ["000000000785.jpg", "000000017627.jpg", "000000024919.jpg", "000000035279.jpg", "000000037777.jpg"]


def _net(dtype="float32"):
    net = yolov3.Darknet(MODELS["yolov3-tiny"], device="cuda", dtype=dtype)
    net.load_weights(golden_weights_path("yolov3-tiny"))
    return net.eval()


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.gpu
def test_batched_frame_loop_equals_per_frame_inference():
    names = sorted(n for n in os.listdir(os.path.join(GOLDEN, "images")) if n.endswith(".jpg"))
    frames = [load_jpeg_bgr(n) for n in names]          # different sizes: resized on the GPU per frame
    net = _net()
    want = [yolov3.inference(net, f, prob_thresh=0.2, nms_iou_thresh=0.3)[0] for f in frames]
    for batch_size, in_flight in ((4, 2), (2, 3), (16, 1), (1, 2)):
        got = list(yolov3.detect_in_frames(net, iter(frames), batch_size=batch_size, prob_thresh=0.2,
                                           nms_iou_thresh=0.3, in_flight=in_flight))
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert _same(g, w), (batch_size, in_flight)
    assert list(yolov3.detect_in_frames(net, [], batch_size=4)) == []


@pytest.mark.gpu
def test_video_loop_over_a_frame_directory_and_cli(tmp_path):
    img_dir = os.path.join(GOLDEN, "images")
    net = _net()
    collected = []
    results = yolov3.detect_in_video(net, img_dir, prob_thresh=0.2, frames=collected, batch_size=4)
    names_sorted = stream.list_image_files(img_dir)[1]
    assert len(results) == len(names_sorted) == len(collected)
    first = yolov3.inference(net, load_jpeg_bgr(names_sorted[0]), prob_thresh=0.2)[0]
    assert _same(results[0], first)
    assert collected[0].shape == load_jpeg_bgr(names_sorted[0]).shape

    out_json, out_dir = tmp_path / "det.json", tmp_path / "frames"
    rc = cli_main(["-I", img_dir, "-c", MODELS["yolov3-tiny"], "-w", golden_weights_path("yolov3-tiny"), "-p", "0.2",
                   "--dtype", "float32", "-b", "4", "--json", str(out_json), "-o", str(out_dir)])
    assert rc == 0
    with open(out_json) as fh:
        ds = json.load(fh)
    assert [im["file_name"] for im in ds["images"]] == names_sorted
    assert len(ds["annotations"]) == sum(len(r[1]) for r in results)
    assert len(os.listdir(out_dir)) == len(names_sorted)


@pytest.mark.gpu
def test_bench_prints_one_contract_json_line():
    """bench.py's output contract (one JSON line with the driver's keys plus roofline / cpu_baseline objects), on a
    small workload so the test stays short."""
    import subprocess
    import sys
    from golden_util import ROOT
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "yolov3-tiny", "--dim", "416",
                           "--batch", "2", "--steps", "3", "--warmup", "1", "--cpu-budget", "4"],
                          capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and "traffic" in r
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert c["cpu_model"] and c["batch1"]["fps"] > 0 and c["batch16"]["fps"] > 0
    # the headline is the PCIe-inclusive rate of the package's own loop; measured in the same run: the resident rate, the
    # other BASELINE configurations, bf16 detection agreement
    assert "pinned host memory" in d["config"]["workload"] and d["config"]["timed_loop"] == "yolov3.pipeline.Pipeline.submit"
    assert d["use_graph"] is False
    assert d["resident"]["value"] > 0
    assert {o["workload"] for o in d["other_configs"]} == {"yolov3-tiny 416x416 batch=8 float32", "yolov3-spp 608x608 batch=16 bf16",
                                                           "yolov3 608x608 batch=16 fp16", "yolov3 608x608 batch=16 bf16",
                                                           "yolov3 608x608 batch=16 float32"}
    for o in d["other_configs"]:
        assert o["value"] > 0 and abs(o["roofline"]["frac"] - o["roofline"]["achieved"] / o["roofline"]["peak"]) < 1e-3
    # the headline workload at the reference test's thresholds and in a heavy regime (more candidates, all kept
    # boxes still inside the records)
    regimes = d["detection_regimes"]
    assert [(g["prob_thresh"], g["nms_iou_thresh"]) for g in regimes] == [(0.2, 0.3), (0.05, 0.3)]
    assert all(g["value"] > 0 and g["candidates_per_frame"] >= g["kept_per_frame"] for g in regimes)
    assert regimes[1]["candidates_per_frame"] > regimes[0]["candidates_per_frame"] and regimes[1]["kept_per_frame"] <= regimes[1]["kmax"]
    agree = d["bf16_agreement"]["thr_0.05_iou_0.3"]
    assert len(agree["keep_set_jaccard"]) == 3 and min(agree["keep_set_jaccard"]) > 0.4
    assert agree["score_abs_diff"]["median"] < 0.01
    # round 5: the fp16 storage mode beside it (same kernels on the f16 MFMA): far closer to the reference's float32 lists
    f16 = d["f16_agreement"]
    assert min(f16["thr_0.05_iou_0.3"]["keep_set_jaccard"]) > 0.85 and f16["thr_0.05_iou_0.3"]["score_abs_diff"]["median"] < 1e-3
    assert f16["bench_regime"]["thr_0.05_iou_0.3"]["keep_set_jaccard"] > 0.9
    # R timed windows, the median is the value; the CPU baseline ran in a pinned child before any GPU call; phases
    assert d["repeats"]["windows"] == 5 and d["repeats"]["ms_per_step_min"] <= d["ms_per_step"] <= d["repeats"]["ms_per_step_max"]
    assert "child process" in c["process"] and len(c["pinned_cpus"]) == c["cores"] and len(c["loadavg_before"]) == 3
    assert d["phases_s"]["total_s"] > 0 and "cpu_baseline_s" in d["phases_s"]
    assert d["per_rank"]["placement"][0]["cpus"] >= 1


@pytest.mark.gpu
def test_bench_resident_variant_runs():
    """`bench.py --resident` (frames already in HBM, records left there: kernel A/B runs): runs and says in
    config.workload that it is not the headline configuration."""
    import subprocess
    import sys
    from golden_util import ROOT
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "yolov3-tiny", "--dim", "416",
                           "--batch", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--resident"],
                          capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    d = json.loads([l for l in proc.stdout.splitlines() if l.startswith("{")][0])
    assert d["value"] > 0 and "NOT the headline" in d["config"]["workload"] and "resident" not in d


@pytest.mark.gpu
def test_bench_under_torchrun_runs_the_rccl_plumbing():
    """One rank under torchrun with Y3_BENCH_FORCE_DIST=1: process-group init on RCCL, barriers, the per-step
    all-gather of detection records issued from three streams, the max all-reduce and the teardown all execute on a
    real GPU (multi-rank correctness of the sharding / gather itself is covered by the gloo test on CPU)."""
    import subprocess
    import sys
    from golden_util import ROOT
    env = dict(os.environ, Y3_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                           "--gpus", "1", "--model", "yolov3-tiny", "--dim", "416", "--batch", "4", "--steps", "6",
                           "--warmup", "3", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0, (proc.stdout[-1500:], proc.stderr[-1500:])
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "all_gather_into_tensor" in d["config"]["collective"]
    assert d["config"]["collective"].startswith("1 x ") and d["config"]["ranks_seen_by_collective"] == 1


@pytest.mark.gpu
def test_bench_refuses_more_gpus_than_visible_on_the_gpu_box():
    """`python bench.py --gpus N` (no torchrun) with fewer than N GPUs: non-zero exit and a clear message, never a
    silent one-rank run."""
    import subprocess
    import sys
    import torch
    from golden_util import ROOT
    want = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--steps", "1", "--warmup", "0"],
                          capture_output=True, text=True, timeout=300, env=env)
    assert proc.returncode != 0
    assert "%d GPUs requested, %d visible" % (want, want - 1) in proc.stderr
    assert not [l for l in proc.stdout.splitlines() if l.startswith("{")]


def test_video_io_without_opencv(tmp_path):
    """Motion-JPEG AVI and YUV4MPEG2, the two containers readable without a codec library (yolov3/videoio.py): what
    the writer produces the reader returns frame for frame (AVI: exactly the JPEG decode of every frame; y4m: YCbCr
    4:2:0 round trip of smooth frames within a few grey levels), frame rates survive, other files are refused."""
    import io
    from PIL import Image
    from yolov3 import videoio
    yy, xx = np.mgrid[0:46, 0:62]
    frames = [np.stack([(xx * 3 + i * 7) % 256, (yy * 5) % 256, (xx + yy + i) % 256], axis=-1).astype(np.uint8) for i in range(5)]
    avi = str(tmp_path / "clip.avi")
    videoio.write_avi_mjpeg(avi, frames, fps=30, quality=95)
    fps, got = videoio.open_video(avi)
    got = list(got)
    assert fps == 30 and len(got) == 5
    for src, g in zip(frames, got):
        buf = io.BytesIO()
        Image.fromarray(src[:, :, ::-1]).save(buf, format="JPEG", quality=95)
        want = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))[:, :, ::-1]
        assert g.shape == src.shape and np.array_equal(g, want)
    ys, xs = np.mgrid[0:40, 0:60]
    smooth = [np.stack([np.full((40, 60), 40 + 10 * i), (xs * 2 + 30) % 200, (ys * 3 + 20) % 200],
                       axis=-1).astype(np.uint8) for i in range(3)]
    y4m = str(tmp_path / "clip.y4m")
    videoio.write_video(y4m, smooth, fps=24)
    fps, got = videoio.open_video(y4m)
    got = list(got)
    assert fps == 24 and len(got) == 3
    for src, g in zip(smooth, got):
        assert g.shape == src.shape and np.abs(g.astype(int) - src.astype(int)).mean() < 4
    grey = np.full((8, 8, 3), 128, np.uint8)          # a grey frame survives YCbCr exactly to +-1
    videoio.write_y4m(y4m, [grey], 25)
    assert np.abs(list(videoio.open_video(y4m)[1])[0].astype(int) - 128).max() <= 1
    bad = tmp_path / "clip.mp4"
    bad.write_bytes(b"\x00\x00\x00\x18ftypmp42" + b"\x00" * 64)
    with pytest.raises(ValueError):
        videoio.open_video(str(bad))
    assert stream.video_fps(avi) == 30


@pytest.mark.gpu
def test_cli_video_file_round_trip(tmp_path):
    """`yolov3 --video clip.avi -o out.avi` without OpenCV: detections equal running the decoded frames through
    detect_in_frames, and the annotated output video has as many frames as the input."""
    from yolov3 import videoio
    from yolov3.__main__ import main as cli_main
    frames = [f for f in synth_frames(11, 5, 208, 256)]
    clip = str(tmp_path / "in.avi")
    videoio.write_avi_mjpeg(clip, frames, fps=12)
    decoded = list(videoio.open_video(clip)[1])
    net = _net()
    want = list(stream.detect_in_frames(net, decoded, batch_size=2, prob_thresh=0.2))
    out = str(tmp_path / "out.avi")
    out_json = str(tmp_path / "out.json")
    rc = cli_main(["-V", clip, "-c", MODELS["yolov3-tiny"], "-w", golden_weights_path("yolov3-tiny"), "-p", "0.2",
                   "--dtype", "float32", "-b", "2", "-o", out, "--json", out_json])
    assert rc == 0
    fps, annotated = videoio.open_video(out)
    assert fps == 12 and len(list(annotated)) == 5
    with open(out_json) as fh:
        ds = json.load(fh)
    assert len(ds["images"]) == 5 and len(ds["annotations"]) == sum(len(r[1]) for r in want)


@pytest.mark.gpu
def test_bench_self_launch_on_a_real_gpu():
    """`python bench.py --gpus N` as the driver calls it, with the one GPU this box has (Y3_BENCH_FORCE_LAUNCH=1): the parent
    counts devices without touching the GPU, starts the torchrun child job under its watchdog, the rank initialises RCCL, verifies
    the ranks with a real all-gather, runs the pipelined steps with one all-gather per step, and the parent relays exactly one
    JSON line (it arrives tee'd through torchrun's per-rank prefix: a 10-KiB line must survive that)."""
    import subprocess
    import sys
    from golden_util import ROOT
    env = dict(os.environ, Y3_BENCH_FORCE_LAUNCH="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--model", "yolov3-tiny", "--dim", "416",
                           "--batch", "4", "--steps", "8", "--warmup", "3", "--no-cpu-baseline", "--launch-timeout", "400"],
                          capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0, (proc.stdout[-1500:], proc.stderr[-2500:])
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["ranks_seen_by_collective"] == 1
    assert "all_gather_into_tensor" in d["config"]["collective"] and len(d["per_rank"]["ms_per_step_by_rank"]) == 1
