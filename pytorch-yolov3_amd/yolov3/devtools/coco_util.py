"""COCO-format bookkeeping for detections (host-side dictionaries only, nothing here touches the GPU).

``match_ids`` follows /root/reference/yolov3/devtools/coco_util.py:110-150: category ids are
re-keyed by category *name* and image ids by *file name* to those of a reference (ground
truth) dataset, images gain the reference's height / width, and the category table is
replaced by the reference's.  ``evaluate_bbox_map`` is the hook for the reference's mAP test
(tests/test_inference.py:61-87); it needs pycocotools, which this image does not ship.
"""


def match_ids(dataset, reference_dataset):
    """Rewrite ``dataset`` in place so its category / image ids are those of ``reference_dataset``.

    Raises ``KeyError`` (like the reference) when the reference names a category or image
    file that ``dataset`` does not contain.
    """
    cat_id_by_name = {cat["name"]: cat["id"] for cat in dataset["categories"]}
    new_cat_id = {cat_id_by_name[cat["name"]]: cat["id"] for cat in reference_dataset["categories"]}

    image_id_by_fname = {im["file_name"]: im["id"] for im in dataset["images"]}
    new_image_id = {image_id_by_fname[im["file_name"]]: im["id"] for im in reference_dataset["images"]}
    ref_image = {im["id"]: im for im in reference_dataset["images"]}

    for im in dataset["images"]:
        ref = ref_image[new_image_id[im["id"]]]
        im["id"] = ref["id"]
        im["height"] = ref["height"]
        im["width"] = ref["width"]
    for ann in dataset["annotations"]:
        ann["category_id"] = new_cat_id[ann["category_id"]]
        ann["image_id"] = new_image_id[ann["image_id"]]
    dataset["categories"] = reference_dataset["categories"]


def evaluate_bbox_map(detections, ground_truth_json):
    """COCO bbox mAP@[.5:.95] of a ``to_coco`` dataset already passed through ``match_ids``.

    Thin wrapper over pycocotools (COCO.loadRes + COCOeval), the same calls the reference's
    test makes; raises ImportError with a clear message where pycocotools is missing.
    """
    try:
        from pycocotools.coco import COCO
        from pycocotools.cocoeval import COCOeval
    except ImportError as exc:  # pragma: no cover - not installable in the build image
        raise ImportError("evaluate_bbox_map needs pycocotools (not shipped in this image)") from exc
    gt = COCO(ground_truth_json)
    dt = gt.loadRes(detections["annotations"])
    ev = COCOeval(gt, dt, "bbox")
    ev.evaluate()
    ev.accumulate()
    ev.summarize()
    return float(ev.stats[0])
