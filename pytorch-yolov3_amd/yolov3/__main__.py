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


def build_parser():
    parser = argparse.ArgumentParser(prog="yolov3")
    source_ = parser.add_argument_group(title="input source [required]")
    source_args = source_.add_mutually_exclusive_group(required=True)
    source_args.add_argument("-C", "--cam", metavar="cam_id", nargs="?", const=0,
                             help="Camera or video capture device ID or path. [Default 0]")
    source_args.add_argument("-I", "--image", type=pathlib.Path, metavar="<path>",
                             help="Path to image file or directory of images.")
    source_args.add_argument("-V", "--video", type=pathlib.Path, metavar="<path>",
                             help="Path to video file (or a directory of frames).")

    model_args = parser.add_argument_group(title="model parameters")
    model_args.add_argument("-c", "--config", type=pathlib.Path, required=True, metavar="<path>",
                            help="[Required] Path to Darknet model config file.")
    model_args.add_argument("-d", "--device", type=str, default="cuda", metavar="<device>",
                            help="Device for inference ('cuda', 'cuda:N'). [Default 'cuda']")
    model_args.add_argument("-i", "--iou-thresh", type=float, default=0.3, metavar="<iou>",
                            help="Non-maximum suppression IOU threshold. [Default 0.3]")
    model_args.add_argument("-n", "--class-names", type=pathlib.Path, metavar="<path>",
                            help="Path to text file of class names. If omitted, class index is displayed "
                                 "instead of name.")
    model_args.add_argument("-p", "--prob-thresh", type=float, default=0.05, metavar="<prob>",
                            help="Detection probability threshold. [Default 0.05]")
    model_args.add_argument("-w", "--weights", type=pathlib.Path, required=True, metavar="<path>",
                            help="[Required] Path to Darknet model weights file.")
    model_args.add_argument("--dtype", default="bf16", choices=["bf16", "float32"],
                            help="Arithmetic of the conv path: bf16 (fast) or float32 (reference parity). "
                                 "[Default bf16]")
    model_args.add_argument("-b", "--batch-size", type=int, default=16, metavar="<n>",
                            help="Frames per GPU batch for --image directories and --video. [Default 16]")

    other_args = parser.add_argument_group(title="Output/display options")
    other_args.add_argument("-o", "--output", type=pathlib.Path, metavar="<path>",
                            help="Where annotated frames go: a directory (PNG per frame) or, with OpenCV "
                                 "installed, an .mp4 file.")
    other_args.add_argument("--json", type=pathlib.Path, metavar="<path>",
                            help="Write all detections as a COCO-format JSON file.")
    other_args.add_argument("--show-fps", action="store_true",
                            help="Display frames processed per second (for --cam input).")
    other_args.add_argument("-v", "--verbose", action="store_true", help="Verbose output")
    return parser


def _abspath(p):
    return None if p is None else str(pathlib.Path(p).expanduser().absolute())


def _write_frames(frames, fps, path):
    if path.endswith(".mp4"):
        from .stream import _cv2
        cv2 = _cv2()
        if cv2 is None:
            raise RuntimeError("writing .mp4 needs OpenCV (cv2), which is not installed; give a directory instead")
        h, w = frames[0].shape[:2]
        writer = cv2.VideoWriter(path, cv2.VideoWriter_fourcc(*"mp4v"), int(fps), (w, h))
        for frame in frames:
            writer.write(frame)
        writer.release()
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
