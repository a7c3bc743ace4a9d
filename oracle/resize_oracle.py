"""TEST INFRASTRUCTURE (oracle/): an independent float bilinear resize, the bound for SURVEY.md 8(f) n1.

The reference resizes non-net-sized frames with ``cv2.resize(image, (net_h, net_w))`` -- OpenCV's INTER_LINEAR, half-pixel
centres, edge pixels replicated, no antialiasing (/root/reference/yolov3/inference.py:320-326).  OpenCV is not in this image,
so the product's integer restatement of its 8-bit arithmetic (yolov3/preprocess.py, csrc/layers.hip) cannot be checked
against cv2 itself.  What CAN be checked is that it is a bilinear resize with those conventions to within its fixed-point
error: this module computes the same interpolation in floating point with ``torch.nn.functional.interpolate(mode="bilinear",
align_corners=False, antialias=False)`` -- code that shares nothing with the product -- and the tests require
|product - round(float)| <= 1 LSB on every byte (11-bit coefficients and two truncating stages stay inside that).

Only tests/ may import this module.
"""
import numpy as np
import torch
import torch.nn.functional as F


def resize_bilinear_float(img, out_h, out_w):
    """uint8 (H, W, C) -> float64 (out_h, out_w, C): bilinear with half-pixel centres, edges replicated, no antialiasing."""
    x = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1)[None].to(torch.float64)
    y = F.interpolate(x, size=(int(out_h), int(out_w)), mode="bilinear", align_corners=False, antialias=False)
    return y[0].permute(1, 2, 0).contiguous().numpy()


def compare_u8(got_u8, want_float):
    """Byte-wise comparison of an 8-bit resize with the float one: (max |difference| in LSB against the rounded float
    result, share of bytes equal to it, max distance from the unrounded float value)."""
    want = np.clip(np.rint(want_float), 0, 255)
    d = np.abs(got_u8.astype(np.float64) - want)
    return float(d.max()), float((d == 0).mean()), float(np.abs(got_u8.astype(np.float64) - want_float).max())
