"""Helpers around the detection path (reference: yolov3/devtools/)."""
from . import coco_util

__all__ = ["coco_util"]
