"""``python -m yolov3``: the reference's command line (/root/reference/yolov3/__main__.py:36-212)
over the HIP path -- same flags and meanings; frames are processed in batches
(``--batch-size``), detections can be dumped as COCO JSON (``--json``), and annotated frames
are written to ``--output`` (a directory of PNGs; an .mp4 needs OpenCV).  There is no window
output without OpenCV and no CPU mode: ``-d cpu`` is refused instead of silently running
something else than the GPU path.
"""
import argparse
import json
import os
import pathlib
import sys
import time


# (flags, keyword arguments) per option; the flag names and defaults are the reference CLI's -- they are the interface a
# user of `yolov3 ...` already knows -- the batching / dtype / json options are this build's additions
_SOURCE_OPTIONS = [
    (("-C", "--cam"), dict(metavar="cam_id", nargs="?", const=0,
                           help="live capture: device number (0 when given bare) or a stream path")),
    (("-I", "--image"), dict(type=pathlib.Path, metavar="<path>", help="one image, or a folder whose images are all run")),
    (("-V", "--video"), dict(type=pathlib.Path, metavar="<path>", help="a video file, or a folder of frames")),
]
_MODEL_OPTIONS = [
    (("-c", "--config"), dict(type=pathlib.Path, required=True, metavar="<path>", help="Darknet .cfg of the network (required)")),
    (("-w", "--weights"), dict(type=pathlib.Path, required=True, metavar="<path>", help="Darknet .weights file (required)")),
    (("-d", "--device"), dict(type=str, default="cuda", metavar="<device>", help="'cuda' or 'cuda:N' (default cuda)")),
    (("-p", "--prob-thresh"), dict(type=float, default=0.05, metavar="<prob>", help="keep detections scoring at least this (default 0.05)")),
    (("-i", "--iou-thresh"), dict(type=float, default=0.3, metavar="<iou>", help="NMS overlap above which the weaker box goes (default 0.3)")),
    (("-n", "--class-names"), dict(type=pathlib.Path, metavar="<path>", help="text file, one class name per line; labels show indices without it")),
    (("--dtype",), dict(default="float32", choices=["float32", "fp16", "bf16"], help="conv arithmetic: float32 (the reference's; default, boxes and scores within 1e-3 of its CPU path), fp16 or bf16 storage with float32 accumulation (about 7x the frames/s; scores within ~1e-3 / ~1e-2)")),
    (("-b", "--batch-size"), dict(type=int, default=16, metavar="<n>", help="frames per GPU batch for folders and videos (default 16)")),
]
_OUTPUT_OPTIONS = [
    (("-o", "--output"), dict(type=pathlib.Path, metavar="<path>", help="annotated frames: a folder of PNGs, or an .mp4 when OpenCV is installed")),
    (("--json",), dict(type=pathlib.Path, metavar="<path>", help="dump every detection as COCO-format JSON")),
    (("--show-fps",), dict(action="store_true", help="overlay the frame rate (camera mode)")),
    (("-v", "--verbose"), dict(action="store_true", help="print device and throughput")),
]


def build_parser():
    parser = argparse.ArgumentParser(prog="yolov3", description="YOLOv3 detection on an MI355X (HIP path)")
    source = parser.add_argument_group(title="input (exactly one)").add_mutually_exclusive_group(required=True)
    for flags, kw in _SOURCE_OPTIONS:
        source.add_argument(*flags, **kw)
    model = parser.add_argument_group(title="model")
    for flags, kw in _MODEL_OPTIONS:
        model.add_argument(*flags, **kw)
    output = parser.add_argument_group(title="output")
    for flags, kw in _OUTPUT_OPTIONS:
        output.add_argument(*flags, **kw)
    return parser


def _abspath(p):
    return None if p is None else str(pathlib.Path(p).expanduser().absolute())


def _write_frames(frames, fps, path):
    """``-o``: .mp4 like the reference's write_mp4 (__main__.py:13-33) when OpenCV is there; .avi (Motion JPEG) / .y4m
    through this package's own writers; anything else is taken as a directory of PNG frames."""
    from .stream import _cv2
    cv2 = _cv2()
    if path.endswith(".mp4"):
        if cv2 is None:
            raise RuntimeError("writing .mp4 needs OpenCV (cv2), which is not installed; give a .avi (Motion JPEG) or "
                               ".y4m file name or a directory instead")
        h, w = frames[0].shape[:2]
        writer = cv2.VideoWriter(path, cv2.VideoWriter_fourcc(*"mp4v"), int(fps), (w, h))
        for frame in frames:
            writer.write(frame)
        writer.release()
        return
    if path.lower().endswith((".avi", ".y4m")):
        from .videoio import write_video
        write_video(path, frames, fps)
        return
    from PIL import Image
    os.makedirs(path, exist_ok=True)
    for i, frame in enumerate(frames):
        Image.fromarray(frame[:, :, ::-1]).save(os.path.join(path, "frame_%06d.png" % i))


def main(argv=None):
    args = vars(build_parser().parse_args(argv))
    for key in ("class_names", "config", "weights", "image", "video", "output", "json"):
        args[key] = _abspath(args[key])
    device = args["device"]
    if not device.startswith("cuda"):
        raise SystemExit("yolov3: device %r refused -- this build has no CPU path (the reference's `-d cpu` mode "
                         "is PyTorch-CPU; use the reference for that)" % device)

    import yolov3
    from yolov3 import stream

    net = yolov3.Darknet(args["config"], device=device, dtype=args["dtype"])
    net.load_weights(args["weights"])
    net.eval()
    net.cuda(device=device)
    if args["verbose"]:
        import torch
        print("Running model on %s" % torch.cuda.get_device_name(net._torch_device()))

    class_names = None
    if args["class_names"] is not None and os.path.isfile(args["class_names"]):
        with open(args["class_names"], "r") as fh:
            class_names = [line.strip() for line in fh.readlines()]

    frames = [] if args["output"] else None
    names, results, fps = None, None, 25.0
    t0 = time.time()
    if args["image"]:
        directory, names = stream.list_image_files(args["image"])
        images = [stream.load_image_bgr(os.path.join(directory, n)) for n in names]
        results = list(stream.detect_in_frames(net, images, batch_size=args["batch_size"],
                                               prob_thresh=args["prob_thresh"], nms_iou_thresh=args["iou_thresh"]))
        if frames is not None:
            for image, (bbox_tlbr, class_prob, class_idx) in zip(images, results):
                stream.draw_boxes(image, bbox_tlbr, class_idx=class_idx, class_names=class_names)
                frames.append(image)
    elif args["video"]:
        fps = stream.video_fps(args["video"], fps)
        results = stream.detect_in_video(net, args["video"], device=device, prob_thresh=args["prob_thresh"],
                                         nms_iou_thresh=args["iou_thresh"], class_names=class_names,
                                         frames=frames, show_video=False, batch_size=args["batch_size"])
        names = ["frame_%06d" % i for i in range(len(results))]
    else:
        cam = args["cam"]
        if isinstance(cam, str) and cam.isdigit():
            cam = int(cam)
        stream.detect_in_cam(net, cam_id=cam, device=device, prob_thresh=args["prob_thresh"],
                             nms_iou_thresh=args["iou_thresh"], class_names=class_names,
                             show_fps=args["show_fps"], frames=frames)
    elapsed = time.time() - t0
    if results is not None and args["verbose"]:
        kept = sum(len(r[1]) for r in results)
        print("%d frames, %d detections, %.1f frames/s (decode + upload + GPU + fetch)" % (
            len(results), kept, len(results) / max(elapsed, 1e-9)))
    if args["json"] and results is not None:
        categories = class_names
        if categories is None:
            top = max([int(r[2].max()) for r in results if len(r[2])] + [0])
            categories = [str(i) for i in range(top + 1)]
        with open(args["json"], "w") as fh:
            json.dump(stream.to_coco(names, results, categories), fh)
    if args["output"] and frames:
        _write_frames(frames, fps, args["output"])
    return 0


if __name__ == "__main__":
    sys.exit(main())
