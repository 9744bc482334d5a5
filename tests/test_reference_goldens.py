"""Callers of the hot path vs golden vectors produced by RUNNING the reference's Python
(oracle/gen_golden_layers.py): anchors, box arithmetic, the five detector losses, and the
state-dict names/shapes of every module (weight-file compatibility).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from helpers import golden_loss_inputs

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(G, "layers.npz"))


def test_anchors_match_reference(gold):
    from feature_intertwiner_amd import layers as L
    scales, ratios, strides = (32, 64, 128, 256, 512), [0.5, 1, 2], [4, 8, 16, 32, 64]
    a = L.generate_pyramid_priors(scales, ratios, np.array([[128 // s] * 2 for s in strides]), strides, 1)
    assert a.shape == gold["anchors_128"].shape
    assert np.array_equal(np.asarray(a, np.float64), gold["anchors_128"])           # reference: lib/layers.py:9-65
    b = L.generate_pyramid_priors(scales, ratios, np.array([[1024 // s] * 2 for s in strides]), strides, 1)
    assert tuple(b.shape) == tuple(gold["anchors_1024_shape"]) == (261888, 4)
    assert np.array_equal(b[:6], gold["anchors_1024_head"]) and np.array_equal(b[-6:], gold["anchors_1024_tail"])
    digest = np.frombuffer(hashlib.sha256(np.asarray(b, np.float32).tobytes()).digest(), np.uint8)
    assert np.array_equal(digest, gold["anchors_1024_sha256_f32"])


def test_box_arithmetic_matches_reference(gold):
    from feature_intertwiner_amd import layers as L
    T = torch.from_numpy
    boxes, deltas = T(gold["boxes"]), T(gold["deltas"])
    got = L.apply_box_deltas(boxes.clone(), deltas)                                   # tools/box_utils.py:7-29
    assert np.array_equal(got.numpy(), gold["apply_box_deltas"])
    for b in range(2):                                                                # un-batched use
        assert np.array_equal(L.apply_box_deltas(boxes[b].clone(), deltas[b]).numpy(), gold["apply_box_deltas"][b])
    w = gold["clip_window"]
    got = L.clip_boxes(T(gold["clip_in"]), tuple(float(v) for v in w))                # :32-60
    assert np.array_equal(got.numpy(), gold["clip_boxes"])
    got = L.box_refinement(boxes[0], T(gold["gt"]))                                   # :89-110
    assert np.array_equal(got.numpy(), gold["box_refinement"])
    got = L.bbox_overlaps(boxes[0], T(gold["gt"][:20]))                               # :113-196
    assert np.allclose(got.numpy(), gold["bbox_overlaps"], rtol=1e-6, atol=1e-7)


def test_losses_match_reference(gold):
    """lib/layers.py:808-934 (nonzero-gather + python loops) vs the masked static-shape forms."""
    from feature_intertwiner_amd import layers as L
    li = {k: torch.from_numpy(v) for k, v in golden_loss_inputs().items()}
    # the reference packs the positive anchors' targets into the first rows of [B,256,4]
    # (lib/layers.py:845-852); the static-shape form keeps them at their anchor: [B,A,4]
    per_anchor = torch.zeros_like(li["rpn_bbox_pred"])
    for b in range(per_anchor.size(0)):
        pos = torch.nonzero(li["rpn_match"][b] == 1).squeeze(1)
        per_anchor[b, pos] = li["rpn_bbox_target"][b, :len(pos)]
    got = dict(
        loss_rpn_class=L.compute_rpn_class_loss(li["rpn_match"], li["rpn_logits"]),
        loss_rpn_bbox=L.compute_rpn_bbox_loss(per_anchor, li["rpn_match"], li["rpn_bbox_pred"]),
        loss_mrcnn_class=L.compute_mrcnn_class_loss(li["cls_ids"], li["cls_logits"]),
        loss_mrcnn_bbox=L.compute_mrcnn_bbox_loss(li["bbox_target"], li["cls_ids"], li["bbox_pred"]),
        loss_mrcnn_mask=L.compute_mrcnn_mask_loss(li["mask_target"], li["cls_ids"], li["mask_pred"]),
    )
    for k, v in got.items():
        assert abs(float(v) - float(gold[k])) <= 2e-6 * max(1.0, abs(float(gold[k]))), (k, float(v), float(gold[k]))


@pytest.mark.parametrize("arch", ["resnet50", "resnet101"])
def test_state_dict_names_and_shapes_match_reference(arch):
    """Every parameter/buffer name, shape AND registration order of the reference modules
    (lib/sub_module.py) -- what tools/utils.py:263-452 keys checkpoints by."""
    from feature_intertwiner_amd import sub_module as M
    from feature_intertwiner_amd.config import make_config
    cfg = make_config(backbone=arch, image_size=128, batch_size=1, train_rois_per_image=16, dev_switch=True)
    ref = json.load(open(os.path.join(G, "state_dict_keys.json")))[arch]
    r = M.ResNet(arch, stage5=True)
    mods = dict(fpn=M.FPN(cfg, *r.stages(), out_channels=256), rpn=M.RPN(3, 1, 256),
                classifier=M.Classifier(256, 81, 7, cfg), mask=M.Mask(256, 81), dev=M.Dev(cfg, 256))
    for k, m in mods.items():
        mine = [[n, list(v.shape)] for n, v in m.state_dict().items()]
        assert mine == ref[k], k
