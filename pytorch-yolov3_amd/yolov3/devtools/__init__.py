"""Tooling around the detection path: COCO id bookkeeping and the mAP hook (the reference keeps its
counterparts under yolov3/devtools/)."""
from . import coco_util  # noqa: F401

__all__ = ("coco_util",)
