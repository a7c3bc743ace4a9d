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


def axis_table(src_len, dst_len):
    """(dst_len, 4) int32 rows {lo, hi, weight_lo, weight_hi}: the tap table both the host resize above and
    the device kernel ``y3_resize_bilinear_u8`` use (so they agree bit for bit)."""
    lo, hi, w0, w1 = _axis_taps(src_len, dst_len)
    return np.ascontiguousarray(np.stack([lo, hi, w0, w1], axis=1).astype(np.int32))


_device_tables = {}


def resize_on_device(frame, out_h, out_w, device, out=None):
    """uint8 (H,W,3) numpy / torch frame -> uint8 (out_h,out_w,3) torch tensor on ``device`` (HIP kernel;
    identical result to :func:`resize_bilinear_u8`).  ``out`` may be a preallocated slice of a batch tensor."""
    import torch
    from . import _hip
    src = frame if isinstance(frame, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(frame))
    src = src.to(device).contiguous()
    sh, sw = int(src.shape[0]), int(src.shape[1])
    if out is None:
        out = torch.empty((out_h, out_w, 3), dtype=torch.uint8, device=device)
    if (sh, sw) == (out_h, out_w):
        out.copy_(src)
        return out
    key = (sh, sw, out_h, out_w, str(device))
    if key not in _device_tables:
        _device_tables[key] = (torch.from_numpy(axis_table(sh, out_h)).to(device),
                               torch.from_numpy(axis_table(sw, out_w)).to(device))
    ytab, xtab = _device_tables[key]
    with torch.cuda.device(device):
        _hip.check(_hip.lib().y3_resize_bilinear_u8(src.data_ptr(), sh, sw, out.data_ptr(), out_h, out_w,
                                                    ytab.data_ptr(), xtab.data_ptr(), _hip.stream_ptr()))
    return out


def prepare_frames_device(images, net_h, net_w, device, resize=True):
    """Like :func:`prepare_frames` but uploads every original frame once and resizes on the GPU."""
    import torch
    if not isinstance(images, (list, tuple)):
        images = [images]
    shapes = [tuple(im.shape) for im in images]
    if not resize:
        # the reference then stacks the frames as they are (np.stack needs equal sizes) and runs the net at that size
        net_h, net_w = shapes[0][0], shapes[0][1]
        if any(s[:2] != (net_h, net_w) for s in shapes):
            raise ValueError("resize=False needs frames of one size, got {}".format(sorted(set(s[:2] for s in shapes))))
    batch = torch.empty((len(images), net_h, net_w, 3), dtype=torch.uint8, device=device)
    for i, im in enumerate(images):
        resize_on_device(im, net_h, net_w, device, out=batch[i])
    return batch, shapes


def prepare_frames(images, net_h, net_w, resize=True):
    """list of HxWx3 uint8 BGR -> (uint8 (B,net_h,net_w,3) BGR, list of original shapes)."""
    if not isinstance(images, (list, tuple)):
        images = [images]
    shapes = [tuple(im.shape) for im in images]
    if resize:
        images = [resize_bilinear_u8(im, net_h, net_w) for im in images]
    return np.ascontiguousarray(np.stack(images)), shapes
