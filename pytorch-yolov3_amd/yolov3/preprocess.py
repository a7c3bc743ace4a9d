"""Host-side frame preparation for ``inference()`` (reference inference.py:314-335).

The reference resizes with ``cv2.resize(image, (net_h, net_w))`` (bilinear,
no letterbox, aspect not preserved) when a frame is not net-sized.  OpenCV is
not available here, so :func:`resize_bilinear_u8` is this build's own
fixed-point bilinear (half-pixel centres, 11-bit coefficients like OpenCV's
INTER_LINEAR path); agreement with cv2 is expected to +-1 LSB but cannot be
pinned without cv2.  Net-sized frames -- the case every
golden vector and the benchmark use -- skip the resize exactly like the
reference does (inference.py:322-326).
"""
import numpy as np

_COEF_BITS = 11
_COEF_ONE = 1 << _COEF_BITS


def _axis_taps(src_len, dst_len):
    scale = src_len / float(dst_len)
    pos = (np.arange(dst_len, dtype=np.float64) + 0.5) * scale - 0.5
    lo = np.floor(pos).astype(np.int64)
    frac = pos - lo
    frac[lo < 0] = 0.0
    lo = np.clip(lo, 0, src_len - 1)
    hi = np.clip(lo + 1, 0, src_len - 1)
    frac[lo >= src_len - 1] = 0.0
    w_hi = np.rint(frac * _COEF_ONE).astype(np.int64)
    return lo, hi, _COEF_ONE - w_hi, w_hi


def resize_bilinear_u8(img, out_h, out_w):
    """uint8 (H,W,C) -> uint8 (out_h,out_w,C), integer arithmetic only."""
    img = np.asarray(img)
    if img.shape[0] == out_h and img.shape[1] == out_w:
        return img
    ylo, yhi, wy0, wy1 = _axis_taps(img.shape[0], out_h)
    xlo, xhi, wx0, wx1 = _axis_taps(img.shape[1], out_w)
    src = img.astype(np.int64)
    rows = src[:, xlo, :] * wx0[None, :, None] + src[:, xhi, :] * wx1[None, :, None]
    acc = rows[ylo] * wy0[:, None, None] + rows[yhi] * wy1[:, None, None]
    out = (acc + (1 << (2 * _COEF_BITS - 1))) >> (2 * _COEF_BITS)
    return np.clip(out, 0, 255).astype(np.uint8)


def prepare_frames(images, net_h, net_w, resize=True):
    """list of HxWx3 uint8 BGR -> (uint8 (B,net_h,net_w,3) BGR, list of original shapes)."""
    if not isinstance(images, (list, tuple)):
        images = [images]
    shapes = [tuple(im.shape) for im in images]
    if resize:
        images = [resize_bilinear_u8(im, net_h, net_w) for im in images]
    return np.ascontiguousarray(np.stack(images)), shapes
