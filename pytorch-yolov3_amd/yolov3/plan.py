"""Layer-plan compiler: Darknet blocks -> flat list of device ops + buffer arena.

Replaces the reference's ``blocks2modules`` / ``forward`` dispatch
(/root/reference/yolov3/darknet.py:218-315, :351-405).  The reference executes
one nn.Module per block and materialises every route (``torch.cat``) and
shortcut (``+``) as a new tensor.  Here the graph is resolved once, on the host:

* shortcut   -> residual input of the preceding conv's epilogue (no add pass);
* route [a]  -> alias of block a's tensor (no copy);
* route [a,b,..] -> one concat buffer; producers a, b, .. write straight into
  their channel slice (pixel stride = total channels), so no concat pass;
* yolo heads -> decode kernels write into row ranges of the final (B, M, .)
  outputs, so no head concat and no w,h rescale pass;
* every other tensor lives in an arena whose slots are reused as soon as their
  last reader has run (keeps the working set small enough to stay in the
  256 MiB Infinity Cache for the deeper stages).

Pure Python, no GPU needed: unit-tested on CPU.
"""

ALIGN = 256          # bytes, arena slot alignment
CH_ALIGN = 8         # channel-slice / pixel-stride granularity (elements): 16 B for bf16


def _round_up(v, m):
    return (v + m - 1) // m * m


class Tensor(object):
    """A (B,H,W,C) NHWC view: channels [off, off+c) of buffer `buf` with pixel stride `ld`."""
    __slots__ = ("buf", "off", "ld", "c", "h", "w", "f32")

    def __init__(self, buf, off, ld, c, h, w, f32=False):
        self.buf, self.off, self.ld, self.c, self.h, self.w, self.f32 = buf, off, ld, c, h, w, f32

    def __repr__(self):
        return "T(buf={} off={} ld={} c={} {}x{}{})".format(
            self.buf, self.off, self.ld, self.c, self.h, self.w, " f32" if self.f32 else "")


def infer_shapes(blocks, net_info, height, width):
    """(C,H,W) of every block output for an input of size (height,width); conv arithmetic
    as torch.nn.Conv2d / MaxPool2d / Upsample compute it."""
    shapes = []
    c, h, w = net_info["channels"], height, width
    for i, blk in enumerate(blocks):
        kind = blk["type"]
        if kind == "convolutional":
            k, s = blk["size"], blk["stride"]
            pad = (k - 1) // 2 if "pad" in blk else 0
            h = (h + 2 * pad - k) // s + 1
            w = (w + 2 * pad - k) // s + 1
            c = blk["filters"]
        elif kind == "maxpool":
            k, s = blk["size"], blk["stride"]
            if not (k > 1 and s == 1):
                h = (h - k) // s + 1
                w = (w - k) // s + 1
        elif kind == "upsample":
            h, w = h * blk["stride"], w * blk["stride"]
        elif kind == "route":
            srcs = [shapes[j] for j in blk["layers"]]
            if any((s_[1], s_[2]) != (srcs[0][1], srcs[0][2]) for s_ in srcs):
                raise ValueError("route block {} joins tensors of different sizes: {}".format(i, srcs))
            c, h, w = sum(s_[0] for s_ in srcs), srcs[0][1], srcs[0][2]
        elif kind == "shortcut":
            a, b = shapes[i - 1], shapes[i + blk["from"]]
            if a != b:
                raise ValueError("shortcut block {} adds {} and {}".format(i, a, b))
            c, h, w = a
        elif kind == "yolo":
            pass
        else:
            raise ValueError("unsupported block type {!r} (block {})".format(kind, i))
        if h <= 0 or w <= 0:
            raise ValueError("block {} produces an empty tensor".format(i))
        shapes.append((c, h, w))
    return shapes


def build_plan(blocks, net_info, batch, height, width, elem_size, reuse=True, fuse=None):
    """Resolve the graph.  ``blocks`` must already carry absolute route indices.

    Returns dict(ops=[...], buffers={id: nbytes}, offsets={id: arena offset}, arena_bytes,
    rows_total, shapes).  Each op is a dict; tensors are :class:`Tensor`.
    ``fuse`` (default: same as ``reuse``): mark conv pairs the executor may run as one kernel; with
    ``reuse=False, fuse=True`` (per-block parity tests) the intermediate tensor of a fused pair keeps
    its arena slot but is never written.
    """
    if fuse is None:
        fuse = reuse
    n = len(blocks)
    shapes = infer_shapes(blocks, net_info, height, width)
    kinds = [b["type"] for b in blocks]

    # ---- consumers of every block output -------------------------------------------------
    readers = [[] for _ in range(n)]      # (consumer block, role)
    for i, blk in enumerate(blocks):
        kind = kinds[i]
        if kind in ("convolutional", "maxpool", "upsample", "yolo"):
            if i > 0:
                readers[i - 1].append((i, "in"))
        elif kind == "route":
            for j in blk["layers"]:
                readers[j].append((i, "route"))
        elif kind == "shortcut":
            readers[i - 1].append((i, "sc_prev"))
            readers[i + blk["from"]].append((i, "sc_from"))

    # ---- shortcut fusion: conv (i-1) + shortcut (i) when nobody else reads conv i-1 -------
    fused_into = {}      # shortcut block -> conv block
    for i, blk in enumerate(blocks):
        if kinds[i] == "shortcut" and kinds[i - 1] == "convolutional" and i + blk["from"] != i - 1:
            if [r for r in readers[i - 1] if r != (i, "sc_prev")] == []:
                fused_into[i] = i - 1
    conv_fused = {v: k for k, v in fused_into.items()}

    # ---- concat placement ------------------------------------------------------------------
    def resolve(j):
        """follow single-source routes down to the block that really produces the data"""
        while kinds[j] == "route" and len(blocks[j]["layers"]) == 1:
            j = blocks[j]["layers"][0]
        return j

    buffers = {}            # id -> dict(bytes, first, last)
    tensor_of = [None] * n
    placed = {}             # producing block -> Tensor inside a concat buffer
    copies = {}             # route block -> list of (src block, Tensor dst) needing a copy op
    head_of = {}            # conv block feeding a yolo block
    for i in range(n):
        if kinds[i] == "yolo" and kinds[i - 1] == "convolutional":
            head_of[i - 1] = i

    for i, blk in enumerate(blocks):
        if kinds[i] == "route" and len(blk["layers"]) > 1:
            c_tot, h, w = shapes[i]
            ld = _round_up(c_tot, CH_ALIGN)
            buf = "cat%d" % i
            buffers[buf] = dict(elems=batch * h * w * ld, es=elem_size)
            tensor_of[i] = Tensor(buf, 0, ld, c_tot, h, w)
            off = 0
            copies[i] = []
            for j in blk["layers"]:
                src = resolve(j)
                # the data of block `src` is produced by conv src-1... if src is a fused shortcut
                cj = shapes[j][0]
                dst = Tensor(buf, off, ld, cj, h, w)
                ok = (src not in placed and off % CH_ALIGN == 0 and src < i
                      and kinds[src] in ("convolutional", "maxpool", "upsample", "shortcut")
                      and src not in head_of and kinds[src] != "route")
                if ok:
                    placed[src] = dst
                else:
                    copies[i].append((j, dst))
                off += cj

    def own_tensor(i, f32=False):
        c, h, w = shapes[i]
        ld = _round_up(c, CH_ALIGN)
        buf = "t%d" % i
        buffers[buf] = dict(elems=batch * h * w * ld, es=4 if f32 else elem_size)
        return Tensor(buf, 0, ld, c, h, w, f32)

    # ---- emit ops ---------------------------------------------------------------------------
    ops = []
    def mask_of(blk):
        m = blk["mask"]
        return m if isinstance(m, list) else [m]      # "mask=0" parses to a bare int

    rows_total = sum(len(mask_of(blocks[i])) * shapes[i][1] * shapes[i][2] for i in range(n) if kinds[i] == "yolo")
    row_offset = 0
    conv_slot = 0
    in_tensor = Tensor("input", 0, net_info["channels"], net_info["channels"], height, width)

    def prev_tensor(i):
        return in_tensor if i == 0 else tensor_of[i - 1]

    for i, blk in enumerate(blocks):
        kind = kinds[i]
        if kind == "convolutional":
            k, s = blk["size"], blk["stride"]
            target = conv_fused.get(i, i)          # block whose tensor this conv produces
            if target in placed:
                out = placed[target]
            else:
                out = own_tensor(target, f32=(i in head_of))
            res = None
            if i in conv_fused:
                sc = conv_fused[i]
                res = tensor_of[sc + blocks[sc]["from"]]
            ops.append(dict(kind="conv", block=i, inp=prev_tensor(i), out=out, res=res, ksize=k, stride=s,
                            pad=(k - 1) // 2 if "pad" in blk else 0, leaky=blk["activation"] == "leaky",
                            slot=conv_slot, bn=bool(blk.get("batch_normalize", 0)), net_input=(i == 0),
                            # hint for the executor: the next op is a conv and the ONLY reader of this conv's
                            # output, so the pair may run as one kernel that never writes this tensor (stem + stride-2
                            # conv, 1x1 + 3x3 of a residual block); needs arena reuse semantics, i.e. not the
                            # keep-every-tensor debugging mode.  Whether a fused kernel exists is the executor's call.
                            fuse_next=bool(fuse and i not in conv_fused and i + 1 < n and
                                           kinds[i + 1] == "convolutional" and readers[i] == [(i + 1, "in")])))
            conv_slot += 1
            if i in conv_fused:
                tensor_of[i] = None                # never materialised
                tensor_of[conv_fused[i]] = out
            else:
                tensor_of[i] = out
        elif kind in ("maxpool", "upsample"):
            out = placed[i] if i in placed else own_tensor(i)
            ops.append(dict(kind=kind, block=i, inp=prev_tensor(i), out=out, ksize=blk.get("size", 1),
                            stride=blk["stride"]))
            tensor_of[i] = out
        elif kind == "shortcut":
            if i in fused_into:
                pass                               # produced by the conv's epilogue
            else:
                out = placed[i] if i in placed else own_tensor(i)
                ops.append(dict(kind="add", block=i, inp=tensor_of[i - 1], res=tensor_of[i + blk["from"]], out=out))
                tensor_of[i] = out
        elif kind == "route":
            if len(blk["layers"]) == 1:
                tensor_of[i] = tensor_of[blk["layers"][0]]
            else:
                for j, dst in copies[i]:
                    ops.append(dict(kind="copy", block=i, inp=tensor_of[j], out=dst))
        elif kind == "yolo":
            src = tensor_of[i - 1]
            c, h, w = shapes[i]
            na = len(mask_of(blk))
            if c % na != 0 or c // na <= 5:
                raise ValueError("yolo block {}: {} channels do not split into {} anchors".format(i, c, na))
            anchors = [blk["anchors"][m] for m in mask_of(blk)]
            ops.append(dict(kind="yolo", block=i, inp=src, anchors=anchors, n_attr=c // na,
                            row_offset=row_offset, rows_total=rows_total))
            row_offset += na * h * w
            tensor_of[i] = src
        if tensor_of[i] is None and kind != "convolutional":
            raise AssertionError("block {} has no tensor".format(i))

    # ---- liveness + arena ---------------------------------------------------------------------
    first, last = {}, {}
    for t, op in enumerate(ops):
        for key in ("inp", "res"):
            tt = op.get(key)
            if tt is not None and tt.buf != "input":
                last[tt.buf] = t
                first.setdefault(tt.buf, t)     # read before written would be a planner bug
        tt = op.get("out")
        if tt is not None:
            first.setdefault(tt.buf, t)
            last[tt.buf] = max(last.get(tt.buf, t), t)
    for buf in buffers:
        if buf not in first:
            raise AssertionError("buffer {} never produced".format(buf))

    nbytes = {buf: _round_up(d["elems"] * d["es"], ALIGN) for buf, d in buffers.items()}
    offsets = {}
    free = []          # (offset, size) sorted by offset
    top = 0

    def alloc(size):
        nonlocal top
        best = None
        for idx, (o, s) in enumerate(free):
            if s >= size and (best is None or s < free[best][1]):
                best = idx
        if best is not None:
            o, s = free.pop(best)
            if s > size:
                free.append((o + size, s - size))
                free.sort()
            return o
        o = top
        top += size
        return o

    def release(o, size):
        free.append((o, size))
        free.sort()
        merged = []
        for fo, fs in free:
            if merged and merged[-1][0] + merged[-1][1] == fo:
                merged[-1] = (merged[-1][0], merged[-1][1] + fs)
            else:
                merged.append((fo, fs))
        free[:] = merged

    by_first = {}
    by_last = {}
    for buf in buffers:
        by_first.setdefault(first[buf], []).append(buf)
        by_last.setdefault(last[buf], []).append(buf)
    for t in range(len(ops)):
        for buf in by_first.get(t, []):
            offsets[buf] = alloc(nbytes[buf])
        if reuse:
            for buf in by_last.get(t, []):
                release(offsets[buf], nbytes[buf])

    return dict(ops=ops, buffers=nbytes, offsets=offsets, arena_bytes=max(top, ALIGN),
                rows_total=rows_total, shapes=shapes, n_convs=conv_slot, live=(first, last),
                tensor_of=tensor_of)
