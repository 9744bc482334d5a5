"""The Python wrappers around the three C operators against golden vectors produced by RUNNING the reference's own
Python (oracle/gen_golden_wrappers.py -> tests/golden/wrappers.npz): RoIAlign.forward's box transform
(lib/roi_align/roi_align.py:26-45), pth_nms's areas / order prelude and nms's truncation to int32 [bs, min_keep]
(lib/nms/pth_nms.py:8-17, lib/nms/nms_wrapper.py:23-34), pyramid_roi_align's level formula, per-level routing and
scatter back into RoI order (lib/layers.py:168-216; the same formula as lib/sub_module.py:405-410).

CPU tests hold the ORACLE's restatements to the goldens; gpu tests hold the PRODUCT (HIP kernels + host wrappers)."""
import os

import numpy as np
import pytest
import torch

from helpers import golden_wrapper_inputs

G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(G, "wrappers.npz"))


@pytest.fixture(scope="module")
def gi():
    return golden_wrapper_inputs()


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


CASES = [(7, True), (7, False), (14, True), (14, False), (1, True), (1, False)]


def _key(crop, fp):
    return "roialign_norm_c%d_%s" % (crop, "fp" if fp else "plain")


# ------------------------------------------------------------------------------------------- oracle vs reference
@pytest.mark.parametrize("crop,fp", CASES)
def test_oracle_roi_align_boxes_equal_the_reference_transform(oracle, gold, gi, crop, fp):
    got = oracle.roi_align_boxes(gi["roialign_boxes_px"], 50, 68, crop, crop, transform_fpcoor=fp)
    assert np.array_equal(_bits(got), _bits(gold[_key(crop, fp)]))


def test_oracle_nms_prelude_and_wrapper_equal_the_reference(oracle, gold, gi):
    dets = gi["nms_dets"]
    one = np.float32(1)
    for i in range(3):
        d = dets[i]
        areas = (d[:, 3] - d[:, 1] + one) * (d[:, 2] - d[:, 0] + one)          # as oracle.pth_nms computes them
        assert np.array_equal(_bits(areas), _bits(gold["nms_areas"][i]))
        assert np.array_equal(np.argsort(-d[:, 4], kind="stable"), gold["nms_order"][i])
    for t in (0.3, 0.7):
        ko = oracle.nms(dets, t)
        assert ko.dtype == np.int32 and np.array_equal(ko, gold["nms_keep_out_%d" % int(t * 10)])
        assert np.array_equal(oracle.pth_nms(dets[0], t), gold["pth_nms_keep_%d" % int(t * 10)])


def test_oracle_level_formula_and_routing_equal_the_reference(oracle, gold, gi):
    rois = gi["pyr_rois"]
    h, w, _ = gi["pyr_image_shape"]
    level = oracle.roi_level(rois, float(h * w))
    assert np.array_equal(level, gold["pyr_level"])
    assert set(np.unique(level)) == {2, 3, 4, 5}
    # per-level calls: boxes in (image, RoI) order of that level -- what dev_stage_groups / the callers rely on
    for lvl in (2, 3, 4, 5):
        img, idx = np.nonzero(level == lvl)
        assert np.array_equal(rois[img, idx], gold["pyr_call_boxes_l%d" % lvl])
        assert np.array_equal(img.astype(np.int32), gold["pyr_call_ind_l%d" % lvl])
    # the whole function on the oracle's crop: pooled rows in RoI order
    for pool in (7, 14):
        exp = gold["pyr_pooled_%d" % pool]
        got = np.zeros_like(exp)
        for lvl in (2, 3, 4, 5):
            img, idx = np.nonzero(level == lvl)
            got[img * rois.shape[1] + idx] = oracle.crop_and_resize_forward(gi["pyr_maps"][lvl - 2], rois[img, idx],
                                                                            img.astype(np.int32), pool, pool)
        assert np.array_equal(_bits(got), _bits(exp))


# ------------------------------------------------------------------------------------------ product vs reference
@pytest.mark.gpu
@pytest.mark.parametrize("crop,fp", CASES)
def test_roi_align_module_hands_the_reference_boxes_to_the_kernel(oracle, gold, gi, crop, fp):
    """RoIAlign (the product module) on the HIP kernel == the oracle's crop on the boxes the REFERENCE's RoIAlign.forward
    computed; the module's own transform bit-equal to them as well."""
    from feature_intertwiner_amd.roi_align.roi_align import RoIAlign, to_crop_boxes
    rs = np.random.RandomState(4)
    fmap = rs.standard_normal((2, 8, 50, 68)).astype(np.float32)
    px, ind = gi["roialign_boxes_px"], gi["roialign_box_ind"]
    norm = to_crop_boxes(torch.from_numpy(px).to(DEV), 50, 68, crop, crop, bin_centres=fp).cpu().numpy()
    assert np.array_equal(_bits(norm), _bits(gold[_key(crop, fp)]))
    got = RoIAlign(crop, crop, 0, fp)(torch.from_numpy(fmap).to(DEV), torch.from_numpy(px).to(DEV),
                                      torch.from_numpy(ind).to(DEV)).cpu().numpy()
    exp = oracle.crop_and_resize_forward(fmap, gold[_key(crop, fp)], ind, crop, crop)
    assert np.array_equal(_bits(got), _bits(exp))


@pytest.mark.gpu
def test_nms_wrappers_return_the_reference_results(gold, gi):
    from feature_intertwiner_amd.nms.nms_wrapper import nms
    from feature_intertwiner_amd.nms.pth_nms import pth_nms
    dets = torch.from_numpy(gi["nms_dets"]).to(DEV)
    for t in (0.3, 0.7):
        ko = nms(dets, t)
        assert isinstance(ko, np.ndarray) and ko.dtype == np.int32
        assert np.array_equal(ko, gold["nms_keep_out_%d" % int(t * 10)])
        keep = pth_nms(dets[0], t)
        assert keep.dtype == torch.int64 and np.array_equal(keep.cpu().numpy(), gold["pth_nms_keep_%d" % int(t * 10)])


@pytest.mark.gpu
@pytest.mark.parametrize("channels_last", [False, True])
def test_pyramid_crop_equals_the_reference_pyramid_roi_align(gold, gi, channels_last):
    """roi_level + pyramid_crop_and_resize (one launch over the levels) == lib/layers.py:143-218 run by the reference
    (level formula, routing, scatter back) around the oracle's crop: levels identical, pooled rows bit-identical."""
    from feature_intertwiner_amd.intertwiner import roi_level
    from feature_intertwiner_amd.roi_align.crop_and_resize import pyramid_crop_and_resize
    rois = torch.from_numpy(gi["pyr_rois"]).to(DEV)
    h, w, _ = gi["pyr_image_shape"]
    level = roi_level(rois, float(h * w))
    assert np.array_equal(level.cpu().numpy(), gold["pyr_level"])
    maps = [torch.from_numpy(m).to(DEV) for m in gi["pyr_maps"]]
    if channels_last:
        maps = [m.contiguous(memory_format=torch.channels_last) for m in maps]
    bs, R = rois.shape[:2]
    box_ind = torch.arange(bs, device=DEV, dtype=torch.int32).repeat_interleave(R)
    for pool in (7, 14):
        got = pyramid_crop_and_resize(maps, rois.reshape(-1, 4), box_ind, level.reshape(-1), pool, pool).cpu().numpy()
        assert np.array_equal(_bits(got), _bits(gold["pyr_pooled_%d" % pool]))


@pytest.mark.gpu
@pytest.mark.parametrize("sample", [0, 1])
def test_detection_layer_returns_what_the_reference_conduct_nms_returned(gold, gi, sample):
    """detection_layer (one sorted NMS launch per image with class-offset boxes, no host synchronisation) against the
    reference's per-sample conduct_nms (lib/layers.py:664-718: a python loop over the classes present, each through the
    `nms` wrapper), run by oracle/gen_golden_wrappers.py: the same rows in the same order -- boxes, class ids, scores --
    and, through `feature`, the same surviving RoI indices; zero padded to DET_MAX_INSTANCES."""
    from types import SimpleNamespace as NS
    from feature_intertwiner_amd.layers import detection_layer
    cls, boxes, scores = gi["det_samples"][sample]
    N, K, S = len(cls), 81, 256.0
    probs = np.full((N, K), 0.001, np.float32)
    probs[np.arange(N), cls] = scores                       # arg max = the class, max = the score
    cfg = NS(TEST=NS(DET_NMS_THRESHOLD=0.3, DET_MAX_INSTANCES=100, DET_MIN_CONFIDENCE=0.5),
             DATA=NS(IMAGE_SHAPE=np.array([256, 256, 3]), BBOX_STD_DEV=np.array([0.1, 0.1, 0.2, 0.2], np.float32)))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    feat = np.arange(N, dtype=np.float32).reshape(N, 1) + 1.0           # row id + 1 (0 = padding)
    det, out_feat = detection_layer(T((boxes / S).reshape(1, N, 4)), T(probs), T(np.zeros((N, K, 4), np.float32)),
                                    T(np.array([[0, 0, S, S]], np.float32)), cfg, feature=T(feat))
    det, out_feat = det.cpu().numpy()[0], out_feat.cpu().numpy()[0, :, 0]
    exp, idx = gold["det_rows_%d" % sample], gold["det_index_%d" % sample]
    n = exp.shape[0]
    assert det.shape == (100, 6) and n <= 100
    assert np.array_equal(det[:n], exp), np.abs(det[:n] - exp).max()
    assert np.all(det[n:] == 0)
    assert np.array_equal(out_feat[:n], idx.astype(np.float32) + 1.0) and np.all(out_feat[n:] == 0)
