"""Darknet ``.cfg`` reader.

Mirrors the observable behaviour of the reference's ``parse_config``
(/root/reference/yolov3/darknet.py:125-215): same block list, same ``net_info``
dict, same scalar coercion (int, then float, then str), comma lists, paired
``anchors`` and list-valued route ``layers``.  The output is pinned against
JSON dumps of the reference's result in tests/golden/parse_config_*.json.
"""

_SECTION_OPEN = "["


def _coerce(token):
    """int if it parses as int, else float if it parses as float, else str."""
    for cast in (int, float):
        try:
            return cast(token)
        except ValueError:
            continue
    return token


def _read_logical_lines(fpath):
    # A line is dropped when it is all whitespace or when its FIRST character
    # (before stripping) is '#' -- reference darknet.py:145-148.
    out = []
    with open(fpath, "r") as fh:
        for raw in fh:
            if raw.isspace() or raw[:1] == "#":
                continue
            out.append(raw.strip())
    return out


def _parse_assignment(section_type, line):
    pieces = line.split("=")
    if len(pieces) != 2:
        # the reference unpacks `key, raw_val = line.split("=")` -> ValueError
        raise ValueError(
            "cfg line must have exactly one '=': {!r}".format(line))
    key = pieces[0].strip()
    rhs = pieces[1]
    if "," in rhs:
        value = [_coerce(tok.strip()) for tok in rhs.split(",")]
    else:
        value = _coerce(rhs.strip())
    if section_type == "route" and key == "layers" and isinstance(value, int):
        value = [value]
    if key == "anchors":
        value = [value[j:j + 2] for j in range(0, len(value), 2)]
    return key, value


def parse_config(fpath):
    """Return ``(blocks, net_info)`` for a Darknet cfg file.

    ``blocks`` is the list of non-``[net]`` sections in file order, each a dict
    with a ``"type"`` entry; ``net_info`` is the (last) ``[net]`` section.
    """
    blocks = []
    net_info = None
    current = None
    for line in _read_logical_lines(fpath):
        if line.startswith(_SECTION_OPEN):
            current = {"type": line[1:-1]}
            if current["type"] == "net":
                net_info = current
            else:
                blocks.append(current)
            continue
        if current is None:
            # text before the first section header is ignored by the reference
            # (its first chunk starts at the first '[' line).
            continue
        key, value = _parse_assignment(current["type"], line)
        current[key] = value
    return blocks, net_info
