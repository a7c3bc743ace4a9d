"""The reference's one published number: bbox mAP@[.50:.95] = 0.33983872 (atol 0.0015) of yolov3 (608 x 608 cfg) on the nine
sample_dataset images, prob_thresh 0.2, nms_iou_thresh 0.3, images run ONE AT A TIME
(/root/reference/tests/test_inference.py:26-87).  It needs the real Darknet checkpoint (248 MB, downloaded by
/root/reference/get_weights.sh) and pycocotools, neither of which exists in the build image or on the GPU boxes, so the test
is skipped unless both are provided:

    Y3_WEIGHTS=/path/to/yolov3.weights  [Y3_COCO_ANNOTATIONS=/path/to/sample.json]  pytest tests/test_map_hook.py -m gpu

Path under test: yolov3.Darknet (HIP kernels, float32) -> yolov3.inference -> yolov3.to_coco -> devtools.coco_util.match_ids
-> pycocotools COCOeval, i.e. the reference test with this package's API in place of the reference's.
"""
import json
import os

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
WEIGHTS = os.environ.get("Y3_WEIGHTS")
# the nine images' COCO annotations: a committed fixture (tools/make_goldens.py g10 copies the reference's data file
# sample_dataset/sample.json): only the weights and pycocotools are missing on the GPU boxes
ANNOTATIONS = os.environ.get("Y3_COCO_ANNOTATIONS", os.path.join(ROOT, "tests", "golden", "sample_annotations.json"))
EXPECTED_MAP = 0.33983872        # tests/test_inference.py:28-32
ATOL = 0.0015                    # tests/test_inference.py:87

try:
    import pycocotools  # noqa: F401
    HAVE_COCO = True
except ImportError:
    HAVE_COCO = False


@pytest.mark.gpu
@pytest.mark.skipif(not (WEIGHTS and os.path.exists(WEIGHTS)), reason="set Y3_WEIGHTS to the real yolov3.weights (248 MB, not redistributable here)")
@pytest.mark.skipif(not HAVE_COCO, reason="pycocotools is not installed")
@pytest.mark.skipif(not os.path.exists(ANNOTATIONS), reason="COCO annotations of the nine sample images not found (Y3_COCO_ANNOTATIONS)")
def test_map_on_sample_dataset_matches_the_reference(tmp_path):
    from PIL import Image
    from pycocotools.coco import COCO
    from pycocotools.cocoeval import COCOeval

    import yolov3
    from yolov3.devtools import coco_util
    from golden_util import GOLDEN, SAMPLE_IMAGES

    net = yolov3.Darknet(os.path.join(ROOT, "pytorch-yolov3_amd", "models", "yolov3.cfg"), device="cuda", dtype="float32")
    net.load_weights(WEIGHTS).eval()
    with open(os.path.join(ROOT, "pytorch-yolov3_amd", "models", "coco.names")) as fh:
        class_names = [ln.strip() for ln in fh if ln.strip()]
    results = []
    for name in SAMPLE_IMAGES:                       # one image per call, like the reference test
        img = np.ascontiguousarray(np.asarray(Image.open(os.path.join(GOLDEN, "images", name)).convert("RGB"))[:, :, ::-1])
        results.extend(yolov3.inference(net, img, device="cuda", prob_thresh=0.2, nms_iou_thresh=0.3))
    pred = yolov3.to_coco(SAMPLE_IMAGES, results, class_names)
    with open(ANNOTATIONS) as fh:
        truth = json.load(fh)
    coco_util.match_ids(pred, truth)
    pred_path = str(tmp_path / "pred.json")
    with open(pred_path, "w") as fh:
        json.dump(pred["annotations"], fh)
    gt = COCO(ANNOTATIONS)
    dt = gt.loadRes(pred_path)
    ev = COCOeval(gt, dt, "bbox")
    ev.evaluate()
    ev.accumulate()
    ev.summarize()
    assert np.isclose(ev.stats[0], EXPECTED_MAP, atol=ATOL), ev.stats[0]


def test_map_hook_is_wired_to_the_package_api():
    """Runs everywhere: the names the hook calls exist with the reference's signatures (to_coco / match_ids are pinned to the
    reference's output by tests/test_callers.py), and the skip conditions name what is missing."""
    import inspect
    import yolov3
    from yolov3.devtools import coco_util
    assert list(inspect.signature(yolov3.to_coco).parameters)[:3] == ["image_filenames", "inference_output", "class_names"]
    assert list(inspect.signature(coco_util.match_ids).parameters)[:2] == ["dataset", "reference_dataset"]
    assert list(inspect.signature(yolov3.inference).parameters)[:2] == ["net", "images"]
    assert EXPECTED_MAP == 0.33983872 and ATOL == 0.0015
    # the ground truth is a committed fixture: nine images, every file name one of the golden JPEGs
    from golden_util import SAMPLE_IMAGES
    with open(ANNOTATIONS) as fh:
        truth = json.load(fh)
    assert sorted(im["file_name"] for im in truth["images"]) == sorted(SAMPLE_IMAGES)
    assert len(truth["annotations"]) == 81 and len(truth["categories"]) == 80
