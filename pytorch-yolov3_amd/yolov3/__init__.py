"""MI355X-native YOLOv3 inference path with the reference's Python surface.

Drop-in names (reference yolov3/__init__.py:1-12): ``Darknet``,
``non_max_suppression``, ``cxywh_to_tlbr``, ``inference``.  All arithmetic on
the path runs in hand-written HIP kernels for gfx950 behind the C ABI declared
in include/yolov3_hip.h (loaded with ctypes by ``yolov3._hip``); there is no
CPU fallback -- if the shared library or a GPU is missing the calls raise.
"""
from .cfgparse import parse_config
from .darknet import Darknet
from .inference import cxywh_to_tlbr, inference, non_max_suppression

__all__ = ["Darknet", "parse_config", "cxywh_to_tlbr", "non_max_suppression", "inference"]
__version__ = "0.1.0"
