"""Import the reference package (read-only, /root/reference) in THIS container.

Only used by tools/make_goldens.py to produce tests/golden/*; never shipped to
the GPU box and never imported by the product, tests, smoke() or bench.py.
Two shims are needed: the reference imports ``cv2`` at
module scope (absent here; only ``cv2.resize`` is on the path and only for
frames that are not net-sized) and uses the removed alias ``np.int``.
"""
import sys
import types

import numpy as np


def load_reference(path="/root/reference"):
    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")

        def _no_resize(*a, **k):
            raise RuntimeError("golden generator feeds net-sized frames only")
        cv2.resize = _no_resize
        sys.modules["cv2"] = cv2
    if not hasattr(np, "int"):
        np.int = int
    if path not in sys.path:
        sys.path.insert(0, path)
    import yolov3 as ref_yolov3  # noqa: E402  (this is the REFERENCE package)
    assert ref_yolov3.__file__.startswith(path), ref_yolov3.__file__
    return ref_yolov3
