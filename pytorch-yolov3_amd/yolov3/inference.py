"""Post-processing entry points with the reference's signatures, computed on the GPU.

``cxywh_to_tlbr``, ``non_max_suppression`` and ``inference`` mirror
/root/reference/yolov3/inference.py:220-368.  Differences that callers can observe:

* kept indices / detections come out in a canonical order (class ascending, score
  descending, then higher index first) instead of the reference's Python-``set`` /
  ``argsort`` dependent order -- the *set* of kept boxes is identical;
* ``inference()`` runs threshold, scaling, integer truncation, corner conversion and
  per-class NMS in one device kernel per batch (libyolov3_hip ``y3_detect``) and copies
  only the surviving detections back to the host;
* frames that are not net-sized are resized on the GPU (``y3_resize_bilinear_u8``, bit-identical to
  ``preprocess.resize_bilinear_u8``; the reference uses ``cv2.resize``, see that module's docstring).

There is no CPU fallback: without the HIP library / a GPU these functions raise.
"""
import ctypes

import numpy as np
import torch

from . import _hip
from .preprocess import prepare_frames_device


def _device(device=None):
    _hip.require_gpu()
    if device is None or str(device) == "cuda":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device(device)


def _float_kind(dtype):
    """numpy float dtype -> (C ABI element type, numpy dtype the kernel computes in): float32 and float64 boxes are computed
    in their own type, step by step as numpy does.  float16 boxes are REFUSED: the reference would compute areas and IoU in
    float16 (a 300 x 300 box's area overflows to inf there), which the float32 kernel would silently not reproduce
    (ADVICE r05); wider floats (longdouble) go through float64."""
    if dtype == np.float16:
        raise TypeError("float16 boxes are not supported: numpy would compute areas and IoU in float16 (overflow above 255 x 255 "
                        "pixels); pass float32 / float64 or integer boxes")
    if dtype == np.float64 or dtype.itemsize > 8:
        return _hip.Y3_F64, np.float64
    return _hip.Y3_F32, np.float32


def _require_finite(arr, what):
    """NaN / inf coordinates: numpy's maximum / minimum propagate NaN and ``inf // 2`` is NaN there, the device kernels
    compare and floor instead -- refuse rather than return a different keep set."""
    if not np.isfinite(arr).all():
        raise ValueError("{} holds non-finite values (NaN / inf): not supported on float boxes".format(what))


def cxywh_to_tlbr(bbox_xywh):
    """(n, >=4) array [cx, cy, w, h, ...] -> [x1, y1, x2, y2, ...] with ``x1 = cx - w//2`` etc. (floor division; extra
    columns pass through).  Integer pixel boxes (what ``inference()`` uses) and float32 / float64 boxes, like the reference
    (inference.py:269-283: ``//`` is numpy's floor division in the array's dtype)."""
    arr = np.asarray(bbox_xywh)
    if arr.ndim != 2 or arr.shape[1] < 4:
        raise ValueError("expected an (n, >=4) array")
    if np.issubdtype(arr.dtype, np.floating):
        if arr.shape[0] == 0:
            return arr.copy()
        code, work = _float_kind(arr.dtype)
        _require_finite(arr, "bbox_xywh")
        dev = _device()
        src = torch.from_numpy(np.ascontiguousarray(arr, dtype=work)).to(dev)
        dst = torch.empty_like(src)
        _hip.check(_hip.lib().y3_cxywh_to_tlbr_float(src.data_ptr(), dst.data_ptr(), arr.shape[0], arr.shape[1], code, _hip.stream_ptr()))
        return dst.cpu().numpy().astype(arr.dtype, copy=False)
    if not np.issubdtype(arr.dtype, np.integer):
        raise TypeError("cxywh_to_tlbr works on integer or floating-point boxes, got dtype {}".format(arr.dtype))
    if arr.shape[0] == 0:
        return arr.copy()
    dev = _device()
    src = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64)).to(dev)
    dst = torch.empty_like(src)
    _hip.check(_hip.lib().y3_cxywh_to_tlbr(src.data_ptr(), dst.data_ptr(), arr.shape[0], arr.shape[1],
                                           _hip.stream_ptr()))
    return dst.cpu().numpy().astype(arr.dtype, copy=False)


def non_max_suppression(bbox_tlbr, class_prob, class_idx=None, iou_thresh=0.3):
    """Greedy NMS; per class when ``class_idx`` is given.  Returns a list of kept indices.

    Same decision rule as the reference (areas with +1, suppress iff IoU > iou_thresh, highest score first).  INTEGER pixel
    corners -- what ``inference()`` feeds it (inference.py:353-355) -- take the int64 device kernel (IoU = int / int in
    float64, like numpy); float32 / float64 boxes (normalised or sub-pixel coordinates: any caller of the public function)
    take ``y3_nms_float``, which computes every step in the array's dtype as numpy does (round 5).
    """
    barr = np.asarray(bbox_tlbr)
    if barr.size and np.issubdtype(barr.dtype, np.floating):
        return _nms_float(barr, class_prob, class_idx, iou_thresh)
    if barr.size and not np.issubdtype(barr.dtype, np.integer):
        raise TypeError("non_max_suppression works on integer or floating-point boxes, got dtype {}".format(barr.dtype))
    boxes = np.ascontiguousarray(np.asarray(bbox_tlbr)[:, :4] if np.asarray(bbox_tlbr).size else
                                 np.zeros((0, 4)), dtype=np.int64)
    prob = np.ascontiguousarray(class_prob, dtype=np.float32)
    n = boxes.shape[0]
    if prob.shape[0] != n:
        raise ValueError("bbox_tlbr and class_prob disagree on the number of boxes")
    if n == 0:
        return []
    dev = _device()
    lib = _hip.lib()
    d_box = torch.from_numpy(boxes).to(dev)
    d_prob = torch.from_numpy(prob).to(dev)
    d_cls = None
    if class_idx is not None:
        cls = np.ascontiguousarray(class_idx, dtype=np.int64)
        if cls.shape[0] != n:
            raise ValueError("class_idx has the wrong length")
        d_cls = torch.from_numpy(cls).to(dev)
    ws_bytes = lib.y3_nms_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    _hip.check(lib.y3_nms(d_box.data_ptr(), d_prob.data_ptr(), d_cls.data_ptr() if d_cls is not None else None,
                          n, float(iou_thresh), ws.data_ptr(), ws_bytes, keep.data_ptr(), count.data_ptr(),
                          _hip.stream_ptr()))
    k = int(count.cpu()[0])
    return keep[:k].cpu().numpy().tolist()


def _nms_float(barr, class_prob, class_idx, iou_thresh):
    code, work = _float_kind(barr.dtype)
    _require_finite(barr[:, :4], "bbox_tlbr")
    boxes = np.ascontiguousarray(barr[:, :4], dtype=work)
    prob = np.ascontiguousarray(class_prob, dtype=np.float64)
    n = boxes.shape[0]
    if prob.shape[0] != n:
        raise ValueError("bbox_tlbr and class_prob disagree on the number of boxes")
    dev = _device()
    lib = _hip.lib()
    d_box = torch.from_numpy(boxes).to(dev)
    d_prob = torch.from_numpy(prob).to(dev)
    d_cls = None
    if class_idx is not None:
        cls = np.ascontiguousarray(class_idx, dtype=np.int64)
        if cls.shape[0] != n:
            raise ValueError("class_idx has the wrong length")
        d_cls = torch.from_numpy(cls).to(dev)
    ws_bytes = lib.y3_nms_float_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    _hip.check(lib.y3_nms_float(d_box.data_ptr(), code, d_prob.data_ptr(), d_cls.data_ptr() if d_cls is not None else None, n,
                                float(iou_thresh), ws.data_ptr(), ws_bytes, keep.data_ptr(), count.data_ptr(), _hip.stream_ptr()))
    return keep[:int(count.cpu()[0])].cpu().numpy().tolist()


class Detector(object):
    """Reusable device buffers for the detection tail of one (batch, rows) shape."""

    def __init__(self, batch, rows, device):
        lib = _hip.lib()
        self.batch, self.rows, self.device = batch, rows, device
        self.ws_bytes = lib.y3_detect_workspace_bytes(batch, rows)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)
        self.count = torch.zeros(batch, dtype=torch.int32, device=device)
        self.tlbr = torch.empty((batch, rows, 4), dtype=torch.int64, device=device)
        self.prob = torch.empty((batch, rows), dtype=torch.float32, device=device)
        self.cls = torch.empty((batch, rows), dtype=torch.int64, device=device)
        self.row = torch.empty((batch, rows), dtype=torch.int32, device=device)
        self.orig_hw = torch.empty((batch, 2), dtype=torch.int32, device=device)
        self._records = {}      # kmax -> (batch, kmax, 8) int32 staging buffer of fetch()

    def run(self, out, orig_hw, prob_thresh, iou_thresh):
        """out: Darknet.forward dict (device tensors).  orig_hw: (batch,2) int32 tensor/array."""
        if not isinstance(orig_hw, torch.Tensor):
            orig_hw = torch.from_numpy(np.ascontiguousarray(orig_hw, dtype=np.int32))
        if (orig_hw.device == self.orig_hw.device and orig_hw.dtype == torch.int32 and orig_hw.is_contiguous()
                and tuple(orig_hw.shape) == (self.batch, 2)):
            hw = orig_hw                       # already resident: no staging copy on the launch path
        else:
            self.orig_hw.copy_(orig_hw, non_blocking=True)
            hw = self.orig_hw
        bbox, prob, cls = out["bbox_xywh"], out["class_prob"], out["class_idx"]
        _hip.check(_hip.lib().y3_detect(
            bbox.data_ptr(), prob.data_ptr(), cls.data_ptr(), self.batch, self.rows, hw.data_ptr(),
            ctypes.c_float(prob_thresh), ctypes.c_double(iou_thresh), self.ws.data_ptr(), self.ws_bytes,
            self.count.data_ptr(), self.tlbr.data_ptr(), self.prob.data_ptr(), self.cls.data_ptr(),
            self.row.data_ptr(), _hip.stream_ptr()))

    def fetch(self, return_rows=False, kmax=1024):
        """Detections of the last ``run`` on the host: ONE device-to-host copy per batch.  The device packs every
        frame's first ``kmax`` detections into fixed-size records that also carry the frame's true count
        (``y3_pack_records``, the multi-GPU gather's format); only if some frame kept more than ``kmax`` boxes is a
        second, larger copy made."""
        from .dist import unpack_records
        lib = _hip.lib()
        kmax = max(1, min(int(kmax), self.rows))
        while True:
            rec = self._records.get(kmax)
            if rec is None:
                rec = self._records[kmax] = torch.empty((self.batch, kmax, 8), dtype=torch.int32, device=self.device)
            _hip.check(lib.y3_pack_records(self.count.data_ptr(), self.tlbr.data_ptr(), self.prob.data_ptr(),
                                           self.cls.data_ptr(), self.row.data_ptr(), self.batch, self.rows, kmax,
                                           rec.data_ptr(), None, _hip.stream_ptr()))
            host = rec.cpu().numpy()                         # the one synchronising copy
            most = int(host[:, 0, 7].max()) if host.size else 0
            if most <= kmax:
                break
            kmax = min(self.rows, max(most, 2 * kmax))
        results = []
        for item in unpack_records(host):
            results.append(item[:4] if return_rows else item[:3])
        return results


_detectors = {}


def get_detector(batch, rows, device):
    key = (batch, rows, str(device))
    det = _detectors.get(key)
    if det is None:
        det = Detector(batch, rows, device)
        _detectors[key] = det
    return det


def inference(net, images, device="cuda", prob_thresh=0.05, nms_iou_thresh=0.3, resize=True,
              return_rows=False):
    """Run detection on one frame or a list of HxWx3 uint8 BGR frames.

    Returns, per frame, ``[bbox_tlbr int64 (K,4), class_prob float32 (K,), class_idx int64 (K,)]``
    (plus the prediction-row index of every detection when ``return_rows``), in original-frame
    pixel coordinates -- same contract as the reference's ``inference()``.
    """
    if not isinstance(images, (list, tuple)):
        images = [images]
    if str(device).startswith("cuda") and not str(net.device).startswith("cuda"):
        net.cuda(device)
    dev = net._torch_device()
    frames, shapes = prepare_frames_device(list(images), net.net_info["height"], net.net_info["width"], dev, resize)
    out = net.forward_frames(frames, fresh=False)
    batch, rows = out["class_prob"].shape
    det = get_detector(batch, rows, dev)
    orig_hw = np.array([[s[0], s[1]] for s in shapes], dtype=np.int32)
    with torch.cuda.device(dev):
        det.run(out, orig_hw, float(np.float32(prob_thresh)), float(nms_iou_thresh))
        return det.fetch(return_rows=return_rows)
