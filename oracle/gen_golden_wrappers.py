"""Golden vectors for the Python wrappers AROUND the three C operators, produced by RUNNING the reference's own
Python with the native extension call replaced by a recorder (TEST INFRASTRUCTURE; build container only):

  * `RoIAlign.forward` (lib/roi_align/roi_align.py:17-48): its `CropAndResizeFunction` is replaced by a class
    that returns the boxes it was handed -> the normalised (y1, x1, y2, x2) boxes the reference passes to the
    crop kernel, for transform_fpcoor True / False and crops 7 / 14 / 1;
  * `pth_nms` + `nms` (lib/nms/pth_nms.py:5-19, lib/nms/nms_wrapper.py:14-34): `_ext.nms.cpu_nms` is replaced
    by a recorder that captures the `order` / `areas` prelude and completes the call with the oracle's
    restatement of the C kernel -> areas, order, and the wrapper's int32 [bs, min_keep] result;
  * `conduct_nms` (lib/layers.py:664-718: the per-sample half of the detection layer -- per-class NMS through the `nms`
    wrapper on that recorder, union, top DET_MAX_INSTANCES by score, output rows) -> detections [<=100, 6] and the
    indices of the surviving RoIs;
  * `pyramid_roi_align` (lib/layers.py:143-218): `CropAndResizeFunction` is replaced by the oracle's crop ->
    which level every RoI is sent to (tools/utils.py:50-55 `log2` + the level formula :168-181, identical to
    lib/sub_module.py:405-410), the per-level box order, and the scatter back into RoI order.

What this pins: everything the reference does in Python on either side of the C kernels (box transform, areas,
ordering, truncation, level routing).  What it does not pin: the C kernels proper (orc_crop_*, orc_nms), which
stay "parity unpinned" (oracle/fi_oracle.c header) -- they sit in the middle of two of the three chains here.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_wrappers.py   ->  tests/golden/wrappers.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import _ref_import  # noqa: E402

_ref_import.install()
OUT = os.path.join(HERE, "..", "tests", "golden")

import lib.layers as RL  # noqa: E402
import lib.nms.nms_wrapper as RW  # noqa: E402
import lib.nms.pth_nms as RN  # noqa: E402
import lib.roi_align.roi_align as RA  # noqa: E402

from helpers import golden_wrapper_inputs  # noqa: E402
import oracle as O  # noqa: E402  (oracle/oracle.py: HERE is first on sys.path)


def gen():
    O.build()
    gi = golden_wrapper_inputs()
    out = {}

    # ---- RoIAlign.forward's box transform
    class ReturnBoxes:
        def __init__(self, *a):
            pass

        def __call__(self, featuremap, boxes, box_ind):
            return boxes
    RA.CropAndResizeFunction = ReturnBoxes
    fmap = torch.zeros(2, 1, 50, 68)
    px = torch.from_numpy(gi["roialign_boxes_px"])
    ind = torch.from_numpy(gi["roialign_box_ind"])
    for crop in (7, 14, 1):
        for fp in (True, False):
            got = RA.RoIAlign(crop, crop, 0, fp)(fmap, px, ind)
            out["roialign_norm_c%d_%s" % (crop, "fp" if fp else "plain")] = got.numpy().astype(np.float32)

    # ---- pth_nms prelude + nms wrapper
    rec = []

    def cpu_nms(keep, num_out, dets, order, areas, thresh):
        rec.append((order.numpy().copy(), areas.numpy().copy()))
        k = O.nms_core(dets.numpy(), order.numpy(), areas.numpy(), thresh)
        keep[:len(k)] = torch.from_numpy(k)
        num_out[0] = len(k)
        return 1
    RN.nms.cpu_nms = cpu_nms
    dets = torch.from_numpy(gi["nms_dets"])
    for t in (0.3, 0.7):
        del rec[:]
        ko = RW.nms(dets, t)
        assert ko.dtype == np.int32 and ko.ndim == 2
        out["nms_keep_out_%d" % int(t * 10)] = ko
        single = RN.pth_nms(dets[0], t)
        out["pth_nms_keep_%d" % int(t * 10)] = single.numpy().astype(np.int64)
    out["nms_order"] = np.stack([r[0] for r in rec[:3]]).astype(np.int64)
    out["nms_areas"] = np.stack([r[1] for r in rec[:3]]).astype(np.float32)

    # ---- conduct_nms (detection layer, one sample at a time)
    from types import SimpleNamespace as NS
    dcfg = NS(TEST=NS(DET_NMS_THRESHOLD=0.3, DET_MAX_INSTANCES=100, DET_MIN_CONFIDENCE=0.5))
    for i, (cls, boxes, scores) in enumerate(gi["det_samples"]):
        cls_t, box_t, sc_t = torch.from_numpy(cls), torch.from_numpy(boxes), torch.from_numpy(scores)
        area = (box_t[:, 0] - box_t[:, 2]) * (box_t[:, 1] - box_t[:, 3])
        keep = (cls_t > 0) & (sc_t >= dcfg.TEST.DET_MIN_CONFIDENCE) & (area > 0)          # lib/layers.py:768-769
        det, final_index = RL.conduct_nms(cls_t, box_t, sc_t, keep, dcfg)
        out["det_rows_%d" % i] = det.numpy().astype(np.float32)
        out["det_index_%d" % i] = final_index.numpy().astype(np.int64)

    # ---- pyramid_roi_align: level routing + scatter back
    calls = []

    class OracleCrop:
        def __init__(self, ch, cw, extrapolation_value=0):
            self.ch, self.cw = ch, cw

        def __call__(self, image, boxes, box_ind):
            calls.append((tuple(image.shape[2:]), boxes.numpy().copy(), box_ind.numpy().copy()))
            return torch.from_numpy(O.crop_and_resize_forward(image.numpy(), boxes.numpy(), box_ind.numpy(), self.ch, self.cw))
    RL.CropAndResizeFunction = OracleCrop
    maps = [torch.from_numpy(m) for m in gi["pyr_maps"]]
    rois = torch.from_numpy(gi["pyr_rois"])
    for pool in (7, 14):
        del calls[:]
        pooled = RL.pyramid_roi_align([rois] + maps, pool, gi["pyr_image_shape"])
        out["pyr_pooled_%d" % pool] = pooled.numpy().astype(np.float32)
    # the level of every RoI, recovered from which map's call carried it
    level = np.zeros(rois.shape[:2], np.int32)
    flat = gi["pyr_rois"]
    for (hw, boxes, bi) in calls:
        lvl = {64: 2, 32: 3, 16: 4, 8: 5}[hw[0]]
        for b, i in zip(boxes, bi):
            j = np.nonzero((flat[i] == b).all(1))[0]
            level[i, j] = lvl
        out["pyr_call_boxes_l%d" % lvl] = boxes
        out["pyr_call_ind_l%d" % lvl] = bi.astype(np.int32)
    assert (level >= 2).all()
    out["pyr_level"] = level
    np.savez_compressed(os.path.join(OUT, "wrappers.npz"), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})
    print("levels:", np.bincount(level.ravel()))


if __name__ == "__main__":
    gen()
