"""MI355X-native YOLOv3 inference path with the reference's Python surface.

Drop-in names (reference yolov3/__init__.py:1-12): ``Darknet``,
``non_max_suppression``, ``cxywh_to_tlbr``, ``inference``, ``to_coco``, ``draw_boxes``,
``detect_in_video``, ``detect_in_cam``, ``devtools``; ``detect_in_frames`` / ``detect_in_images``
are the batched loops the reference's CLI leaves as a TODO.  All arithmetic on
the path runs in hand-written HIP kernels for gfx950 behind the C ABI declared
in include/yolov3_hip.h (loaded with ctypes by ``yolov3._hip``); there is no
CPU fallback -- if the shared library or a GPU is missing the calls raise.
"""
from .cfgparse import parse_config
from .darknet import Darknet
from .inference import cxywh_to_tlbr, inference, non_max_suppression
from .stream import (detect_in_cam, detect_in_frames, detect_in_images, detect_in_video, draw_boxes,
                     to_coco)
from . import devtools

__all__ = ["Darknet", "parse_config", "cxywh_to_tlbr", "draw_boxes", "non_max_suppression", "inference",
           "to_coco", "detect_in_cam", "detect_in_video", "detect_in_frames", "detect_in_images", "devtools"]
__version__ = "0.1.0"
