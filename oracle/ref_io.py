"""Oracle-side readers of the reference's two file formats -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The oracle must not share host code with the product it checks (a parser bug would be common to both and cancel out in every
comparison), so these are written independently of pytorch-yolov3_amd/yolov3/cfgparse.py and weights.py.  Both restate
what the reference does, citing it; tests/test_oracle_golden.py pins them to the JSON dumps of the reference's own
``parse_config`` output (tests/golden/parse_config_*.json) and to the product's readers on the same files.

  read_cfg        /root/reference/yolov3/darknet.py:125-215  (``parse_config``)
  conv_shapes     /root/reference/yolov3/darknet.py:229-313  (channel bookkeeping of ``blocks2modules``)
  read_weights    /root/reference/yolov3/darknet.py:407-476  (``Darknet.load_weights``)
"""
import numpy as np


def _scalar(text):
    # darknet.py:169-181: an int if int() takes it, else a float if float() takes it, else the string itself
    try:
        return int(text)
    except ValueError:
        try:
            return float(text)
        except ValueError:
            return text


def read_cfg(path):
    """-> (blocks, net_info).  A line counts if it is not all whitespace and its first character (BEFORE stripping) is not
    '#' (darknet.py:145-148); a line starting with '[' opens a section named by what is between its first and last character
    (:153-161, :184); every other line must split on '=' into exactly two parts (:186); a right-hand side containing a
    comma becomes a list of scalars (:190-193); a route's ``layers`` is always a list (:198-203); ``anchors`` are paired
    (:208-209); the LAST ``[net]`` section is ``net_info``, every other section is a block, in file order (:213-216)."""
    sections = []
    with open(path, "r") as fh:
        for raw in fh.readlines():
            if raw.isspace() or raw.startswith("#"):
                continue
            text = raw.strip()
            if text.startswith("["):
                sections.append({"type": text[1:-1]})
                continue
            if not sections:
                continue                      # text before the first section belongs to no chunk (:153-161)
            left, right = text.split("=")     # ValueError unless there is exactly one '=' -- like the reference's unpacking
            key = left.strip()
            parts = [_scalar(p.strip()) for p in right.split(",")]
            value = parts if "," in right else parts[0]
            owner = sections[-1]
            if owner["type"] == "route" and key == "layers" and isinstance(value, int):
                value = [value]
            if key == "anchors":
                value = [value[k:k + 2] for k in range(0, len(value), 2)]
            owner[key] = value
    net = None
    blocks = []
    for sec in sections:
        if sec["type"] == "net":
            net = sec
        else:
            blocks.append(sec)
    return blocks, net


def conv_shapes(blocks, in_channels):
    """(cout, cin, k, has_bn_params) of every convolutional block in file order.  Channels follow the reference's module
    builder: a conv outputs ``filters`` (darknet.py:244-250); a route outputs the sum of its sources' channels, negative
    indices relative to the route (:283-290, absolute ones after Darknet.__init__ :338-343); shortcut / maxpool /
    upsample / yolo keep the count (:265-313).  ``has_bn_params``: the LOADER's test, key present AND truthy (:428)."""
    out_channels = []
    convs = []
    current = in_channels
    for pos, blk in enumerate(blocks):
        kind = blk["type"]
        if kind == "convolutional":
            convs.append((blk["filters"], current, blk["size"], bool("batch_normalize" in blk and blk["batch_normalize"])))
            current = blk["filters"]
        elif kind == "route":
            current = sum(out_channels[src if src >= 0 else pos + src] for src in blk["layers"])
        out_channels.append(current)
    return convs


def read_weights(path, blocks, in_channels):
    """-> (header int32[5], [per-conv dict]).  File = 5 int32 (darknet.py:416-417) then float32 values consumed in block
    order: for a conv with batch-norm parameters bn bias, bn weight, running mean, running var (``cout`` each, :433-461),
    else the conv bias (:466-474); then the conv weight, ``cout*cin*k*k`` values viewed as (cout, cin, k, k) (:476-481).
    A short file fails (the reference's ``view_as`` raises RuntimeError); trailing values are ignored."""
    with open(path, "rb") as fh:
        header = np.fromfile(fh, dtype=np.int32, count=5)
        values = np.fromfile(fh, dtype=np.float32)
    cursor = 0
    out = []
    for cout, cin, k, bn in conv_shapes(blocks, in_channels):
        names = ["bn_beta", "bn_gamma", "bn_mean", "bn_var"] if bn else ["bias"]
        entry = {}
        for name in names + ["weight"]:
            count = cout * cin * k * k if name == "weight" else cout
            piece = values[cursor:cursor + count]
            if piece.size != count:
                raise RuntimeError("weights file %r ends inside %s of a %dx%dx%dx%d conv" % (path, name, cout, cin, k, k))
            entry[name] = piece.reshape(cout, cin, k, k).copy() if name == "weight" else piece.copy()
            cursor += count
        out.append(entry)
    return header, out
