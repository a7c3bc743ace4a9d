"""Callers of the hot path: batched image / video / frame-stream loops, COCO export, box drawing.

Reference counterparts: ``to_coco`` (/root/reference/yolov3/inference.py:371-432), ``draw_boxes``
(:97-158), ``detect_in_video`` (:496-544), ``detect_in_cam`` (:435-493) and the ``--image`` loop
of the CLI (/root/reference/yolov3/__main__.py:161-187, which runs one frame at a time and leaves
batching as a TODO).  Here frames are grouped into batches and three batches are kept in flight on
separate HIP streams (each with its own activation arena and detection buffers; yolov3/pipeline.py), because a GPU
only earns its throughput on batches; results are yielded in frame order and are identical to
calling ``inference()`` on every frame by itself.

Decoding / display are not part of the GPU path: images are decoded with PIL; video files are read with OpenCV
when it is installed (any codec) and otherwise by ``videoio.py`` (Motion-JPEG AVI and YUV4MPEG2, pure Python);
cameras and windows need OpenCV, which this image does not ship -- those entry points raise a RuntimeError
that says so instead of silently doing something else.
"""
import colorsys
import os

import numpy as np
import torch

from . import _hip
from .preprocess import prepare_frames_device

IMAGE_EXTENSIONS = (".jpg", ".jpeg", ".png", ".bmp", ".ppm", ".webp", ".tif", ".tiff")


def _cv2():
    try:
        import cv2  # noqa: WPS433 (optional dependency)
        return cv2
    except ImportError:
        return None


def load_image_bgr(path):
    """Decode an image file to an HxWx3 uint8 BGR array (what ``cv2.imread`` returns)."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def list_image_files(path):
    """``path``: an image file or a directory.  Returns (directory, [file names]); directory
    listings are sorted (the reference uses ``os.listdir`` order, which is arbitrary) and
    restricted to image extensions."""
    path = str(path)
    if os.path.isdir(path):
        names = sorted(n for n in os.listdir(path) if n.lower().endswith(IMAGE_EXTENSIONS))
        return path, names
    directory, name = os.path.split(path)
    return directory, [name]


def _batches(frames, batch_size):
    batch = []
    for frame in frames:
        batch.append(frame)
        if len(batch) == batch_size:
            yield batch
            batch = []
    if batch:
        yield batch


def detect_in_frames(net, frames, batch_size=16, prob_thresh=0.05, nms_iou_thresh=0.3, resize=True,
                     in_flight=3, kmax=512):
    """Generator over ``[bbox_tlbr, class_prob, class_idx]`` for every frame of the iterable
    ``frames`` (HxWx3 uint8 BGR arrays; sizes may differ when ``resize``), in order.

    The frames run through :class:`yolov3.pipeline.Pipeline` -- the loop ``bench.py`` times -- in batches of
    ``batch_size`` with ``in_flight`` batches on their own HIP streams: a batch's detections are taken only when its
    slot is needed again (or at the end), so decoding / uploading batch k+1 overlaps the kernels of batch k.
    Net-sized frames are stacked straight into one of the pipeline's pinned host buffers and uploaded from there;
    other sizes are uploaded one by one and resized on the GPU.  ``frames`` may also yield whole (B, H, W, 3) batches
    of net-sized frames: a pinned torch tensor is then uploaded from where it lies (no host copy at all), which is how
    a decoder that fills ``Pipeline.host_frames`` reaches the benchmark's rate.
    """
    from .pipeline import Pipeline
    _hip.require_gpu()
    if batch_size < 1 or in_flight < 1:
        raise ValueError("batch_size and in_flight must be positive")
    if not str(net.device).startswith("cuda"):
        net.cuda()
    dev = net._torch_device()
    height, width = net.net_info["height"], net.net_info["width"]
    state = {"pipe": None, "keep": {}, "filled": 0}

    def pipeline(batch):
        # Pipelines are kept on the network (streams, pinned and device buffers, detector workspaces: ~50 ms to set up, half a
        # second of video at the rate they run at) and handed to one generator at a time
        if state["pipe"] is None:
            cache = net.__dict__.setdefault("_pipelines", {})
            key = (batch, height, width, in_flight, kmax, str(net.device), net.dtype)
            pipe = cache.get(key)
            if pipe is None or pipe.busy:
                pipe = Pipeline(net, batch, height, width, in_flight=in_flight, kmax=kmax)
                if key not in cache or not cache[key].busy:
                    cache[key] = pipe
            pipe.busy = True
            pipe.prob_thresh, pipe.nms_iou_thresh = float(np.float32(prob_thresh)), float(nms_iou_thresh)
            state["pipe"] = pipe
        return state["pipe"]

    def submit(batch):
        """batch: list of frames, or one (B, H, W, 3) array / tensor.  Returns the ticket."""
        whole = not isinstance(batch, list)
        n = int(batch.shape[0]) if whole else len(batch)
        pipe = pipeline(n)
        if n > pipe.batch:
            raise ValueError("a batch of {} frames after the pipeline was sized for {}".format(n, pipe.batch))
        if whole:
            if tuple(batch.shape[1:]) != (height, width, 3):
                raise ValueError("whole batches must be net-sized: (B, {}, {}, 3)".format(height, width))
            if n == pipe.batch:
                return pipe.submit(batch, n_frames=n)
            batch = list(batch)                               # a short last batch: through the staging path below
        if all(tuple(f.shape[:2]) == (height, width) for f in batch):
            j = state["filled"] % pipe.nbuf
            state["filled"] += 1
            pipe.upload_done(j)                               # the upload that last read this pinned buffer is over
            host = pipe.host_frames(j).numpy()
            for i, f in enumerate(batch):
                host[i] = f.numpy() if isinstance(f, torch.Tensor) else f
            return pipe.submit(pipe.host_frames(j), n_frames=n)
        # frames of other sizes: uploaded one by one and resized on the GPU, on the stream the pipeline will run the batch on
        stream, slot = pipe.next_slot()
        pad = batch + [batch[-1]] * (pipe.batch - n)
        with torch.cuda.device(dev), torch.cuda.stream(stream):
            dev_frames, shapes = prepare_frames_device(pad, height, width, dev, resize)
        state["keep"][slot] = dev_frames                      # alive until the ticket's buffers are reused
        orig_hw = np.array([[s[0], s[1]] for s in shapes], dtype=np.int32)
        return pipe.submit(dev_frames, orig_hw=orig_hw, n_frames=n)

    def batches():
        group = []
        for item in frames:
            if getattr(item, "ndim", 3) == 4:
                if group:
                    yield group
                    group = []
                yield item
                continue
            group.append(item)
            if len(group) == batch_size:
                yield group
                group = []
        if group:
            yield group

    open_tickets = []
    try:
        for batch in batches():
            if state["pipe"] is not None and len(open_tickets) == state["pipe"].max_open:   # the buffers this ticket will take
                for result in state["pipe"].results(open_tickets.pop(0)):
                    yield result
            open_tickets.append(submit(batch))
        for ticket in open_tickets:
            for result in state["pipe"].results(ticket):
                yield result
    finally:
        if state["pipe"] is not None:
            state["pipe"].synchronize()                       # (a generator abandoned half way leaves nothing in flight)
            state["pipe"].busy = False


def detect_in_images(net, path, batch_size=16, prob_thresh=0.05, nms_iou_thresh=0.3):
    """The CLI's ``--image`` mode: ``path`` is a file or a directory.  Returns (file names, results)."""
    directory, names = list_image_files(path)
    frames = (load_image_bgr(os.path.join(directory, n)) for n in names)
    results = list(detect_in_frames(net, frames, batch_size=batch_size, prob_thresh=prob_thresh,
                                    nms_iou_thresh=nms_iou_thresh))
    return names, results


def video_fps(filepath, default=25.0):
    """Frame rate of a video file (``default`` for a directory of frames or when the container does not say)."""
    if os.path.isdir(filepath):
        return default
    cv2 = _cv2()
    if cv2 is not None:
        cap = cv2.VideoCapture(filepath)
        fps = cap.get(cv2.CAP_PROP_FPS)
        cap.release()
        return fps or default
    from .videoio import open_video
    fps, frames = open_video(filepath)
    frames.close()
    return fps or default


def _video_frames(filepath):
    """Frames of ``filepath``: a directory of images (sorted) or a video file (any codec with OpenCV, Motion-JPEG
    AVI / YUV4MPEG2 without)."""
    if os.path.isdir(filepath):
        directory, names = list_image_files(filepath)
        for n in names:
            yield load_image_bgr(os.path.join(directory, n))
        return
    cv2 = _cv2()
    if cv2 is None:
        from .videoio import open_video           # Motion-JPEG .avi / .y4m; raises ValueError for anything else
        _, frames = open_video(filepath)
        for frame in frames:
            yield frame
        return
    cap = cv2.VideoCapture(filepath)
    try:
        while True:
            grabbed, frame = cap.read()
            if not grabbed:
                break
            yield frame
    finally:
        cap.release()


def detect_in_video(net, filepath, device="cuda", prob_thresh=0.05, nms_iou_thresh=0.3, class_names=None,
                    frames=None, show_video=False, batch_size=16):
    """Run detection over a video (or a directory of frames), draw the boxes on every frame and
    append the frames to ``frames`` when a list is given -- the reference's contract, batched.
    Returns the list of per-frame results."""
    if show_video and _cv2() is None:
        raise RuntimeError("show_video needs OpenCV (cv2), which is not installed")
    if str(device).startswith("cuda") and not str(net.device).startswith("cuda"):
        net.cuda(device)
    kept = []

    def tap():
        for frame in _video_frames(str(filepath)):
            kept.append(frame)
            yield frame

    results = []
    for i, (bbox_tlbr, class_prob, class_idx) in enumerate(
            detect_in_frames(net, tap(), batch_size=batch_size, prob_thresh=prob_thresh,
                             nms_iou_thresh=nms_iou_thresh)):
        frame = kept[i]
        kept[i] = None
        draw_boxes(frame, bbox_tlbr, class_idx=class_idx, class_names=class_names)
        if frames is not None:
            frames.append(frame)
        if show_video:
            cv2 = _cv2()
            cv2.imshow("YOLOv3", frame)
            if cv2.waitKey(1) == ord("q"):
                break
        results.append([bbox_tlbr, class_prob, class_idx])
    return results


def detect_in_cam(net, cam_id=0, device="cuda", prob_thresh=0.05, nms_iou_thresh=0.3, class_names=None,
                  show_fps=False, frames=None):
    """Live camera loop (latency-bound, one frame per step like the reference).  Needs OpenCV for
    capture and display."""
    cv2 = _cv2()
    if cv2 is None:
        raise RuntimeError("camera capture needs OpenCV (cv2), which is not installed")
    from .inference import inference
    import time
    cap = cv2.VideoCapture(cam_id)
    try:
        while True:
            t0 = time.time()
            grabbed, frame = cap.read()
            if not grabbed:
                break
            bbox_tlbr, _, class_idx = inference(net, frame, device=device, prob_thresh=prob_thresh,
                                                nms_iou_thresh=nms_iou_thresh)[0]
            draw_boxes(frame, bbox_tlbr, class_idx=class_idx, class_names=class_names)
            if show_fps:
                cv2.putText(frame, "%d fps" % int(1.0 / max(time.time() - t0, 1e-6)), (2, 20),
                            cv2.FONT_HERSHEY_COMPLEX_SMALL, 0.9, (255, 255, 255))
            if frames is not None:
                frames.append(frame)
            cv2.imshow("YOLOv3", frame)
            if cv2.waitKey(1) == ord("q"):
                break
    finally:
        cap.release()
        cv2.destroyAllWindows()


def to_coco(image_filenames, inference_output, class_names):
    """``inference()`` / ``detect_in_frames()`` results -> COCO detection dataset (dict).

    Same layout as the reference: categories are (index, name) pairs, image ids are list
    positions, boxes are [x, y, w, h] with w = x2 - x1, h = y2 - y1, annotation ids count up
    from 0 across images; all numbers are plain Python ints / floats so the dict goes
    straight into ``json.dump``.
    """
    dataset = {
        "info": [],
        "licenses": [],
        "categories": [{"id": i, "name": name} for i, name in enumerate(class_names)],
        "images": [],
        "annotations": [],
    }
    for image_id, (bbox_tlbr, class_prob, class_idx) in enumerate(inference_output):
        dataset["images"].append({"file_name": image_filenames[image_id], "id": image_id})
        for j, (x1, y1, x2, y2) in enumerate(np.asarray(bbox_tlbr).reshape(-1, 4).tolist()):
            w, h = int(x2) - int(x1), int(y2) - int(y1)
            dataset["annotations"].append({
                "image_id": image_id,
                "bbox": [int(x1), int(y1), w, h],
                "category_id": int(class_idx[j]),
                "id": len(dataset["annotations"]),
                "score": float(class_prob[j]),
                "area": w * h,
            })
    return dataset


def unique_colors(num_colors):
    """``num_colors`` evenly spaced hues as 8-bit BGR tuples, the channel order of the frames they are drawn into
    (reference: inference.py:80-93)."""
    colors = []
    for h in np.linspace(0, 1, num_colors, endpoint=False):
        r, g, b = colorsys.hsv_to_rgb(h, 1.0, 1.0)
        colors.append((int(255 * b), int(255 * g), int(255 * r)))
    return colors


def _rect(img, x1, y1, x2, y2, color, thickness):
    h, w = img.shape[:2]
    half_lo, half_hi = thickness // 2, (thickness - 1) // 2

    def band(a0, a1, limit):
        return max(0, min(a0, limit)), max(0, min(a1, limit))
    ya, yb = band(y1 - half_lo, y2 + half_hi + 1, h)
    xa, xb = band(x1 - half_lo, x2 + half_hi + 1, w)
    for (r0, r1) in (band(y1 - half_lo, y1 + half_hi + 1, h), band(y2 - half_lo, y2 + half_hi + 1, h)):
        img[r0:r1, xa:xb] = color
    for (c0, c1) in (band(x1 - half_lo, x1 + half_hi + 1, w), band(x2 - half_lo, x2 + half_hi + 1, w)):
        img[ya:yb, c0:c1] = color


def draw_boxes(img, bbox_tlbr, class_prob=None, class_idx=None, class_names=None):
    """Draw the boxes in place (2-pixel outlines; one colour per class when ``class_names`` is
    given, else green) and, where PIL is available, the reference's label text (class name or
    index, then "(prob)").  Boxes may leave the image; they are clipped while drawing."""
    colors = unique_colors(len(class_names)) if class_names is not None else None
    labels = []
    for i, (x1, y1, x2, y2) in enumerate(np.asarray(bbox_tlbr).reshape(-1, 4).tolist()):
        color = colors[int(class_idx[i])] if colors is not None else (0, 255, 0)
        _rect(img, int(x1), int(y1), int(x2), int(y2), color, 2)
        text = []
        if class_names is not None:
            text.append(str(class_names[int(class_idx[i])]))
        elif class_idx is not None:
            text.append(str(int(class_idx[i])))
        if class_prob is not None:
            text.append("({:.2f})".format(float(class_prob[i])))
        if text:
            labels.append((int(x1), int(y1), " ".join(text)))
    if labels:
        try:
            from PIL import Image, ImageDraw
        except ImportError:  # pragma: no cover
            return img
        for x, y, text in labels:
            h, w = img.shape[:2]
            x0, y0 = max(0, min(x + 1, w - 1)), max(0, min(y + 1, h - 1))
            x1, y1 = max(0, min(x + 8 * len(text), w)), max(0, min(y + 18, h))
            if x1 <= x0 or y1 <= y0:
                continue
            patch = Image.new("RGB", (x1 - x0, y1 - y0), (20, 20, 20))
            ImageDraw.Draw(patch).text((1, 3), text, fill=(255, 255, 255))
            img[y0:y1, x0:x1] = np.asarray(patch)[:, :, ::-1]
    return img
