"""Procedural uint8 BGR frames (no dataset access on the GPU box).

Pure integer-hash scenes: a colour gradient, a few dozen flat rectangles and
per-pixel noise.  Unlike i.i.d. noise they have spatial structure, so the
detection heads see varied activations (many classes, overlapping boxes for
NMS).  Bit-reproducible everywhere (see weights.hash_uniform).
"""
import numpy as np

from .weights import hash_uniform


def synth_frames(seed, n, height, width, rects=48):
    """Return uint8 array (n, height, width, 3), BGR like ``cv2.imread``."""
    frames = np.empty((n, height, width, 3), dtype=np.uint8)
    yy = np.arange(height, dtype=np.int64)[:, None]
    xx = np.arange(width, dtype=np.int64)[None, :]
    for f in range(n):
        s = seed * 1000 + f
        g = (hash_uniform(s, 1, 9) * 256).astype(np.int64)
        img = np.empty((height, width, 3), dtype=np.int64)
        for c in range(3):
            img[:, :, c] = (g[c] + (g[3 + c] * yy) // height + (g[6 + c] * xx) // width) % 256
        r = hash_uniform(s, 2, rects * 7)
        for k in range(rects):
            cx, cy, hw, hh = r[7 * k:7 * k + 4]
            x0 = int(cx * width)
            y0 = int(cy * height)
            w2 = 4 + int(hw * hw * width * 0.3)
            h2 = 4 + int(hh * hh * height * 0.3)
            col = (r[7 * k + 4:7 * k + 7] * 256).astype(np.int64)
            img[max(0, y0 - h2):y0 + h2, max(0, x0 - w2):x0 + w2, :] = col
        noise = (hash_uniform(s, 3, height * width * 3) * 48).astype(np.int64).reshape(height, width, 3)
        frames[f] = np.clip(img + noise - 24, 0, 255).astype(np.uint8)
    return frames
