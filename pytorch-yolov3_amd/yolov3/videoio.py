"""Video files without OpenCV: MJPEG-in-AVI and YUV4MPEG2 readers / writers in pure Python (numpy + PIL).

The reference reads videos with ``cv2.VideoCapture`` (/root/reference/yolov3/inference.py:496-544) and writes
``.mp4`` with ``cv2.VideoWriter`` (/root/reference/yolov3/__main__.py:13-33).  OpenCV is not part of this image,
and decoding is not part of the GPU hot path, so this module covers the two container formats that need no codec
library -- every frame of a Motion-JPEG AVI is a JPEG (PIL decodes it), and a ``.y4m`` file is raw planar YCbCr --
so that ``yolov3 --video clip.avi -o out.avi`` works end to end.  With OpenCV installed, ``stream.py`` prefers it
(any codec, and ``.mp4`` output exactly like the reference).

Frames are HxWx3 uint8 BGR arrays, like ``cv2.VideoCapture.read`` returns them.
"""
import io
import struct

import numpy as np

# BT.601 "studio swing" 8-bit YCbCr <-> RGB in 16.16 fixed point (ITU-R BT.601-7, section 2.5)


def _ycbcr_to_bgr(y, cb, cr):
    c = y.astype(np.int32) - 16
    d = cb.astype(np.int32) - 128
    e = cr.astype(np.int32) - 128
    r = (76309 * c + 104597 * e + 32768) >> 16
    g = (76309 * c - 25675 * d - 53279 * e + 32768) >> 16
    b = (76309 * c + 132201 * d + 32768) >> 16
    return np.clip(np.stack([b, g, r], axis=-1), 0, 255).astype(np.uint8)


def _bgr_to_ycbcr(frame):
    b = frame[:, :, 0].astype(np.int32)
    g = frame[:, :, 1].astype(np.int32)
    r = frame[:, :, 2].astype(np.int32)
    y = ((16829 * r + 33039 * g + 6416 * b + 32768) >> 16) + 16
    cb = ((-9714 * r - 19070 * g + 28784 * b + 32768) >> 16) + 128
    cr = ((28784 * r - 24103 * g - 4681 * b + 32768) >> 16) + 128
    return (np.clip(y, 0, 255).astype(np.uint8), np.clip(cb, 0, 255).astype(np.uint8),
            np.clip(cr, 0, 255).astype(np.uint8))


# ---------------------------------------------------------------------------------------------------
# YUV4MPEG2
# ---------------------------------------------------------------------------------------------------

def read_y4m(path):
    """(fps, generator of BGR frames) of a YUV4MPEG2 file (8-bit 4:2:0 or 4:4:4, progressive)."""
    fh = open(path, "rb")
    header = fh.readline()
    if not header.startswith(b"YUV4MPEG2"):
        fh.close()
        raise ValueError("%s is not a YUV4MPEG2 file" % path)
    width = height = None
    fps, chroma = 25.0, "420"
    for tok in header.split()[1:]:
        tag, val = tok[:1], tok[1:].decode("ascii", "replace")
        if tag == b"W":
            width = int(val)
        elif tag == b"H":
            height = int(val)
        elif tag == b"F":
            num, den = val.split(":")
            fps = float(num) / max(float(den), 1.0)
        elif tag == b"C":
            chroma = val
    if not width or not height:
        fh.close()
        raise ValueError("%s: no frame size in the YUV4MPEG2 header" % path)
    if chroma.startswith("444") and "p" not in chroma:
        cw, ch = width, height
    elif chroma.startswith("420") and "p1" not in chroma:          # 420, 420jpeg, 420mpeg2, 420paldv: 8 bit
        cw, ch = (width + 1) // 2, (height + 1) // 2
    else:
        fh.close()
        raise ValueError("%s: chroma format C%s is not supported (8-bit 4:2:0 and 4:4:4 only)" % (path, chroma))

    def frames():
        try:
            while True:
                line = fh.readline()
                if not line:
                    return
                if not line.startswith(b"FRAME"):
                    raise ValueError("%s: bad frame marker %r" % (path, line[:16]))
                raw = fh.read(width * height + 2 * cw * ch)
                if len(raw) < width * height + 2 * cw * ch:
                    return
                y = np.frombuffer(raw, np.uint8, width * height).reshape(height, width)
                cb = np.frombuffer(raw, np.uint8, cw * ch, width * height).reshape(ch, cw)
                cr = np.frombuffer(raw, np.uint8, cw * ch, width * height + cw * ch).reshape(ch, cw)
                if (cw, ch) != (width, height):
                    cb = np.repeat(np.repeat(cb, 2, axis=0), 2, axis=1)[:height, :width]
                    cr = np.repeat(np.repeat(cr, 2, axis=0), 2, axis=1)[:height, :width]
                yield _ycbcr_to_bgr(y, cb, cr)
        finally:
            fh.close()

    return fps, frames()


def write_y4m(path, frames, fps=25):
    """Write BGR frames as 8-bit 4:2:0 YUV4MPEG2 (chroma = mean of each 2x2 block)."""
    with open(path, "wb") as fh:
        first = True
        for frame in frames:
            h, w = frame.shape[:2]
            if first:
                fh.write(("YUV4MPEG2 W%d H%d F%d:1 Ip A1:1 C420jpeg\n" % (w, h, int(round(fps)))).encode("ascii"))
                first = False
            y, cb, cr = _bgr_to_ycbcr(frame)
            ph, pw = (h + 1) // 2 * 2, (w + 1) // 2 * 2

            def sub(p):
                q = np.pad(p, ((0, ph - h), (0, pw - w)), mode="edge").astype(np.uint16)
                return ((q[0::2, 0::2] + q[0::2, 1::2] + q[1::2, 0::2] + q[1::2, 1::2] + 2) >> 2).astype(np.uint8)
            fh.write(b"FRAME\n")
            fh.write(y.tobytes())
            fh.write(sub(cb).tobytes())
            fh.write(sub(cr).tobytes())


# ---------------------------------------------------------------------------------------------------
# Motion-JPEG AVI (RIFF)
# ---------------------------------------------------------------------------------------------------

def _chunks(fh, end):
    """(fourcc, data offset, size) of the RIFF chunks between the current position and ``end``."""
    while fh.tell() + 8 <= end:
        head = fh.read(8)
        if len(head) < 8:
            return
        fourcc, size = head[:4], struct.unpack("<I", head[4:])[0]
        pos = fh.tell()
        yield fourcc, pos, size
        fh.seek(pos + size + (size & 1))


def read_avi_mjpeg(path):
    """(fps, generator of BGR frames) of an AVI file whose video stream is Motion JPEG."""
    from PIL import Image
    fh = open(path, "rb")
    riff = fh.read(12)
    if len(riff) < 12 or riff[:4] != b"RIFF" or riff[8:12] != b"AVI ":
        fh.close()
        raise ValueError("%s is not an AVI file" % path)
    file_end = 8 + struct.unpack("<I", riff[4:8])[0]
    fps, movi = 25.0, None
    handler = None
    for fourcc, pos, size in _chunks(fh, file_end):
        if fourcc != b"LIST":
            continue
        kind = fh.read(4)
        if kind == b"hdrl":
            for f2, p2, s2 in _chunks(fh, pos + size):
                if f2 == b"avih":
                    usec = struct.unpack("<I", fh.read(4))[0]
                    if usec:
                        fps = 1e6 / usec
                elif f2 == b"LIST":
                    if fh.read(4) == b"strl":
                        for f3, p3, s3 in _chunks(fh, p2 + s2):
                            if f3 == b"strh":
                                strh = fh.read(min(s3, 56))
                                if strh[:4] == b"vids":
                                    handler = strh[4:8]
                                    scale, rate = struct.unpack("<II", strh[20:28])
                                    if scale and rate:
                                        fps = rate / float(scale)
        elif kind == b"movi":
            movi = (pos + 4, pos + size)
    if movi is None:
        fh.close()
        raise ValueError("%s: no 'movi' list" % path)
    if handler is not None and handler.upper() not in (b"MJPG", b"JPEG", b"\x00\x00\x00\x00"):
        fh.close()
        raise ValueError("%s: video stream is %r, only Motion JPEG can be read without OpenCV" % (path, handler))

    def frames():
        try:
            fh.seek(movi[0])
            for fourcc, pos, size in _chunks(fh, movi[1]):
                if fourcc[2:] in (b"dc", b"db") and size > 0:
                    data = fh.read(size)
                    with Image.open(io.BytesIO(data)) as im:
                        rgb = np.asarray(im.convert("RGB"))
                    yield np.ascontiguousarray(rgb[:, :, ::-1])
        finally:
            fh.close()

    return fps, frames()


def write_avi_mjpeg(path, frames, fps=25, quality=92):
    """Write BGR frames as a Motion-JPEG AVI (one video stream, 'MJPG', index chunk included)."""
    from PIL import Image
    encoded = []
    width = height = 0
    for frame in frames:
        height, width = frame.shape[:2]
        buf = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(frame[:, :, ::-1])).save(buf, format="JPEG", quality=quality)
        encoded.append(buf.getvalue())
    if not encoded:
        raise ValueError("no frames to write")
    rate = max(1, int(round(fps)))
    movi = io.BytesIO()
    index = []
    for data in encoded:
        index.append((4 + movi.tell(), len(data)))          # offset relative to the 'movi' fourcc
        movi.write(b"00dc" + struct.pack("<I", len(data)) + data + (b"\x00" if len(data) & 1 else b""))
    movi_bytes = movi.getvalue()
    biggest = max(len(d) for d in encoded)
    avih = struct.pack("<IIIIIIIIIIIIII", 1000000 // rate, biggest * rate, 0, 0x10, len(encoded), 0, 1, biggest,
                       width, height, 0, 0, 0, 0)
    strh = struct.pack("<4s4sIHHIIIIIIIIhhhh", b"vids", b"MJPG", 0, 0, 0, 0, 1, rate, 0, len(encoded), biggest,
                       0xFFFFFFFF, 0, 0, 0, width, height)
    strf = struct.pack("<IiiHH4sIiiII", 40, width, height, 1, 24, b"MJPG", width * height * 3, 0, 0, 0, 0)

    def chunk(fourcc, data):
        return fourcc + struct.pack("<I", len(data)) + data + (b"\x00" if len(data) & 1 else b"")

    def lst(kind, data):
        return b"LIST" + struct.pack("<I", 4 + len(data)) + kind + data

    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    idx1 = b"".join(b"00dc" + struct.pack("<III", 0x10, off, size) for off, size in index)
    body = b"AVI " + hdrl + lst(b"movi", movi_bytes) + chunk(b"idx1", idx1)
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", len(body)) + body)


# ---------------------------------------------------------------------------------------------------

def open_video(path):
    """(fps, frame generator) for ``path`` by its magic bytes: YUV4MPEG2 or Motion-JPEG AVI."""
    with open(path, "rb") as fh:
        magic = fh.read(12)
    if magic.startswith(b"YUV4MPEG2"):
        return read_y4m(path)
    if magic[:4] == b"RIFF" and magic[8:12] == b"AVI ":
        return read_avi_mjpeg(path)
    raise ValueError("%s: only Motion-JPEG .avi and .y4m files can be read without OpenCV (cv2 is not installed)" % path)


def write_video(path, frames, fps=25):
    """Write ``frames`` to ``path``: ``.y4m`` -> YUV4MPEG2, anything else -> Motion-JPEG AVI."""
    if str(path).lower().endswith(".y4m"):
        write_y4m(path, frames, fps)
    else:
        write_avi_mjpeg(path, frames, fps)
