"""Host-side frame preparation for ``inference()`` (reference inference.py:314-335).

The reference resizes with ``cv2.resize(image, (net_h, net_w))`` (bilinear, no letterbox, aspect not
preserved) when a frame is not net-sized.  OpenCV is not available in this image, so
:func:`resize_bilinear_u8` RESTATES OpenCV's published algorithm for 8-bit ``INTER_LINEAR``
(opencv/modules/imgproc/src/resize.cpp, 4.x: ``resizeGeneric_`` with ``HResizeLinear<uchar,int,short,2048>``
and the 8-bit specialisation of ``VResizeLinear``; IPP is not used for 8-bit linear unless
``useIPP_NotExact``):

* per destination column ``fx = (float)((dx + 0.5) * scale_x - 0.5)`` with ``scale_x = 1 / (dst_w / src_w)`` in
  double, ``sx = floor(fx)``, ``fx -= sx``; ``sx < 0 -> sx = 0, fx = 0``; ``sx >= src_w - 1 -> sx = src_w - 1,
  fx = 0`` (that column then reads the single pixel times 2048); coefficients
  ``saturate_cast<short>((1 - fx) * 2048)``, ``saturate_cast<short>(fx * 2048)`` (round half to even);
* rows likewise, except that the two source rows are clamped into the image instead of the weight being zeroed;
* horizontal pass to int: ``S[sx] * a0 + S[sx + 1] * a1``; vertical pass
  ``(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2``  -- two stages, each truncating;
* an exact 2:1 reduction, which OpenCV routes to its INTER_AREA fast path ``(s00 + s01 + s10 + s11 + 2) >> 2``,
  comes out of the formulas above unchanged (all four weights are 1024).

PARITY against OpenCV ITSELF is unpinned for this function (no cv2 in this image: INTEGRATION.md).  What is pinned
(tests/test_resize_pin.py): host (:func:`resize_bilinear_u8`) and device (``y3_resize_bilinear_u8``) are within 1 LSB on
every byte -- 88-91 % of the bytes equal -- of an independent float bilinear with the same conventions
(``torch.nn.functional.interpolate``, kept with the test infrastructure) on the nine sample images, up- and down-scaling, and
bit-identical to each other; net-sized frames -- the benchmark's case, and the crop goldens tests/golden/inference_crops_* --
skip the resize exactly like the reference does (inference.py:322-326).
Like the reference, ``dsize`` is passed as ``(net_h, net_w)`` although cv2 reads it as (width, height): for the
square networks shipped here that is the same thing, and :func:`reference_dsize` keeps the quirk for others.
"""
import numpy as np

_COEF_BITS = 11
_COEF_ONE = 1 << _COEF_BITS          # INTER_RESIZE_COEF_SCALE


def _round_half_even_to_short(v):
    return np.clip(np.rint(v), -32768, 32767).astype(np.int64)     # cvRound + saturate_cast<short>


def _axis_taps(src_len, dst_len, clamp_weights):
    """OpenCV's xofs / ialpha (``clamp_weights=True``) or yofs / ibeta (False) tables for one axis:
    (lo index, hi index, weight_lo, weight_hi)."""
    inv_scale = float(dst_len) / float(src_len)
    scale = 1.0 / inv_scale
    f = ((np.arange(dst_len, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_weights:
        low = s < 0
        f[low] = 0.0
        s[low] = 0
        high = s >= src_len - 1
        f[high] = 0.0
        s[high] = src_len - 1
    w_lo = _round_half_even_to_short((np.float32(1.0) - f) * np.float32(_COEF_ONE))
    w_hi = _round_half_even_to_short(f * np.float32(_COEF_ONE))
    lo = np.clip(s, 0, src_len - 1)
    hi = np.clip(s + 1, 0, src_len - 1)
    if clamp_weights:
        # columns at / past the last pixel read that pixel alone, times ONE (HResizeLinear's dx >= xmax loop)
        w_hi = np.where(high, 0, w_hi)
        w_lo = np.where(high, _COEF_ONE, w_lo)
    return lo, hi, w_lo, w_hi


def resize_bilinear_u8(img, out_h, out_w):
    """uint8 (H,W,C) -> uint8 (out_h,out_w,C): OpenCV's 8-bit INTER_LINEAR arithmetic, integers only."""
    img = np.asarray(img)
    if img.shape[0] == out_h and img.shape[1] == out_w:
        return img
    ylo, yhi, wy0, wy1 = _axis_taps(img.shape[0], out_h, False)
    xlo, xhi, wx0, wx1 = _axis_taps(img.shape[1], out_w, True)
    src = img.astype(np.int64)
    rows = src[:, xlo, :] * wx0[None, :, None] + src[:, xhi, :] * wx1[None, :, None]      # HResizeLinear -> int
    top, bot = rows[ylo] >> 4, rows[yhi] >> 4
    out = (((wy0[:, None, None] * top) >> 16) + ((wy1[:, None, None] * bot) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def reference_dsize(net_h, net_w):
    """Rows, columns of the frame the reference feeds the network: it calls ``cv2.resize(image, (net_h, net_w))``
    (inference.py:323-325) and cv2 reads dsize as (width, height), so the result has net_w rows and net_h columns."""
    return net_w, net_h


def axis_table(src_len, dst_len, clamp_weights):
    """(dst_len, 4) int32 rows {lo, hi, weight_lo, weight_hi}: the tap table both the host resize above and
    the device kernel ``y3_resize_bilinear_u8`` use (so they agree bit for bit)."""
    lo, hi, w0, w1 = _axis_taps(src_len, dst_len, clamp_weights)
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
        _device_tables[key] = (torch.from_numpy(axis_table(sh, out_h, False)).to(device),
                               torch.from_numpy(axis_table(sw, out_w, True)).to(device))
    ytab, xtab = _device_tables[key]
    with torch.cuda.device(device):
        _hip.check(_hip.lib().y3_resize_bilinear_u8(src.data_ptr(), sh, sw, out.data_ptr(), out_h, out_w,
                                                    ytab.data_ptr(), xtab.data_ptr(), _hip.stream_ptr()))
    return out


def _target_shapes(shapes, net_h, net_w, resize):
    """Rows, columns every frame has when it enters the network, following inference.py:320-326: a frame that
    already is (net_h, net_w) stays, any other goes through ``cv2.resize(image, (net_h, net_w))`` and comes out with
    ``reference_dsize`` rows / columns.  Like ``np.stack`` there, a batch must end up with one size."""
    if resize:
        out = [s[:2] if tuple(s[:2]) == (net_h, net_w) else reference_dsize(net_h, net_w) for s in shapes]
    else:
        out = [tuple(s[:2]) for s in shapes]
    if len(set(out)) != 1:
        raise ValueError("frames of different sizes cannot form one batch: {}".format(sorted(set(out))))
    return out[0]


def prepare_frames_device(images, net_h, net_w, device, resize=True):
    """Like :func:`prepare_frames` but uploads every original frame once and resizes on the GPU."""
    import torch
    if not isinstance(images, (list, tuple)):
        images = [images]
    shapes = [tuple(im.shape) for im in images]
    out_h, out_w = _target_shapes(shapes, net_h, net_w, resize)
    batch = torch.empty((len(images), out_h, out_w, 3), dtype=torch.uint8, device=device)
    for i, im in enumerate(images):
        resize_on_device(im, out_h, out_w, device, out=batch[i])
    return batch, shapes


def prepare_frames(images, net_h, net_w, resize=True):
    """list of HxWx3 uint8 BGR -> (uint8 (B,h,w,3) BGR, list of original shapes)."""
    if not isinstance(images, (list, tuple)):
        images = [images]
    shapes = [tuple(im.shape) for im in images]
    out_h, out_w = _target_shapes(shapes, net_h, net_w, resize)
    images = [resize_bilinear_u8(im, out_h, out_w) for im in images]
    return np.ascontiguousarray(np.stack(images)), shapes
