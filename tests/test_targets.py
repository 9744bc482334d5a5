"""Target generation against the oracle's line-by-line restatement of the reference
(oracle.generate_rpn_target = lib/layers.py:439-604, oracle.generate_roi = :224-376).

The reference draws random sub-samples (torch.randperm / np.random.permutation); the build draws them on
the device, so the streams differ.  The oracle takes the permutations as arguments: each test recovers
WHICH candidates the build kept, replays exactly that choice through the oracle, and then demands equal
outputs -- the candidate sets, the sampling rules (positives first, ROI_POSITIVE_RATIO, crowd handling)
and every derived value are the reference's."""
import numpy as np
import pytest
import torch


def _cfg(**kw):
    from feature_intertwiner_amd.config import make_config
    return make_config(**kw)


def _perm_dropping(candidates, kept):
    """A permutation of range(len(candidates)) whose leading entries select exactly candidates \\ kept."""
    kept = set(int(k) for k in kept)
    drop = [i for i, c in enumerate(candidates) if int(c) not in kept]
    keep = [i for i, c in enumerate(candidates) if int(c) in kept]
    return np.array(drop + keep, np.int64)


def _gt(rs, b, G, size, n_valid, crowd=False):
    side = np.exp(rs.uniform(np.log(12), np.log(size / 2), (b, G)))
    asp = np.exp(rs.uniform(np.log(0.5), np.log(2.0), (b, G)))
    h, w = side / np.sqrt(asp), side * np.sqrt(asp)
    y1, x1 = rs.uniform(0, size - h), rs.uniform(0, size - w)
    boxes = np.stack([y1, x1, y1 + h, x1 + w], 2).astype(np.float32)
    cls = rs.randint(1, 81, (b, G)).astype(np.int64)
    for i in range(b):
        cls[i, n_valid[i]:] = 0
        boxes[i, n_valid[i]:] = 0
    if crowd:
        cls[0, 1] = -1                      # a COCO crowd box in image 0 (lib/layers.py:229-246, 455-472)
    return cls, boxes


@pytest.mark.parametrize("crowd,pos_thres", [(False, 0.7), (True, 0.7), (False, 0.4)])
def test_rpn_targets_equal_the_reference_rules(oracle, crowd, pos_thres):
    """CPU (prepare_rpn_target is plain tensor arithmetic): match vector identical, positive deltas equal.
    pos_thres 0.4 produces more than TRAIN_ANCHORS_PER_IMAGE / 2 positive candidates, i.e. the
    positive-reduction branch (lib/layers.py:512-527)."""
    from feature_intertwiner_amd import layers as L
    cfg = _cfg(backbone="resnet50", image_size=256)
    cfg.RPN.TARGET_POS_THRES = pos_thres
    anchors = L.generate_pyramid_priors(cfg.RPN.ANCHOR_SCALES, cfg.RPN.ANCHOR_RATIOS, cfg.MODEL.BACKBONE_SHAPES,
                                        cfg.MODEL.BACKBONE_STRIDES, 1).astype(np.float32)
    rs = np.random.RandomState(3)
    cls, boxes = _gt(rs, 3, 12, 256, [12, 7, 0], crowd)
    g = torch.Generator().manual_seed(5)
    match, deltas = L.prepare_rpn_target(torch.from_numpy(anchors), torch.from_numpy(cls), torch.from_numpy(boxes), cfg, g)
    match, deltas = match.numpy(), deltas.numpy()
    for i in range(3):
        m0 = _candidates(oracle, anchors, cls[i], boxes[i], cfg)
        pos_c, neg_c = np.nonzero(m0 == 1)[0], np.nonzero(m0 == -1)[0]
        kept_pos, kept_neg = np.nonzero(match[i] == 1)[0], np.nonzero(match[i] == -1)[0]
        assert set(kept_pos) <= set(pos_c) and set(kept_neg) <= set(neg_c)
        if pos_thres < 0.5 and i == 0:
            assert len(pos_c) > 128 and len(kept_pos) == 128          # the reduction really happened
        exp_match, exp_bbox = oracle.generate_rpn_target(anchors, cls[i], boxes[i], cfg,
                                                         _perm_dropping(pos_c, kept_pos), _perm_dropping(neg_c, kept_neg))
        assert np.array_equal(match[i], exp_match), i
        n_pos = int((exp_match == 1).sum())
        assert n_pos <= 128 and n_pos + int((exp_match == -1).sum()) <= 256
        if i < 2:
            assert n_pos > 0 and n_pos + int((exp_match == -1).sum()) == 256
        got = deltas[i][exp_match == 1]                                    # per-anchor layout -> anchor order
        exp = exp_bbox[:n_pos] / np.asarray(cfg.DATA.BBOX_STD_DEV, np.float32)
        assert np.allclose(got, exp, rtol=1e-5, atol=1e-6)
        assert np.all(deltas[i][exp_match != 1] == 0)
    assert np.all(match[2] != 1)                                           # image without objects


def _candidates(oracle, anchors, cls, boxes, cfg):
    """match vector before the balancing step: the oracle with a budget that never binds."""
    from copy import deepcopy
    big = deepcopy(cfg)
    big.RPN.TRAIN_ANCHORS_PER_IMAGE = 4 * anchors.shape[0]
    m, _ = oracle.generate_rpn_target(anchors, cls, boxes, big)
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("crowd", [False, True])
def test_detection_targets_equal_the_reference_rules(oracle, crowd):
    """GPU (mask targets go through the HIP crop_and_resize): RoIs, class ids, deltas and 28x28 mask
    targets identical to generate_roi replayed with the build's own random choice."""
    from feature_intertwiner_amd import layers as L
    DEV = "cuda:0"
    cfg = _cfg(backbone="resnet50", image_size=512, batch_size=3, train_rois_per_image=96)
    rs = np.random.RandomState(9)
    b, G, P = 3, 14, 700
    cls, boxes = _gt(rs, b, G, 512, [14, 9, 0], crowd)
    yy, xx = np.meshgrid(np.linspace(-1, 1, 56), np.linspace(-1, 1, 56), indexing="ij")
    masks = np.stack([[((yy / rs.uniform(0.5, 1)) ** 2 + (xx / rs.uniform(0.5, 1)) ** 2 <= 1).astype(np.float32)
                       for _ in range(G)] for _ in range(b)])
    gtn = boxes / 512.0
    props = np.zeros((b, P, 4), np.float32)
    num = np.array([P, 500, 300], np.int32)
    for i in range(b):
        n = int(num[i])
        k = n // 2 if (cls[i] != 0).any() else 0
        src = gtn[i][rs.randint(0, max(int((cls[i] != 0).sum()), 1), k)]
        jit = src * (1 + 0.12 * (rs.uniform(size=(k, 4)) - 0.5)).astype(np.float32)
        y1x1 = rs.uniform(0, 0.8, (n - k, 2))
        rnd = np.concatenate([y1x1, y1x1 + rs.uniform(0.02, 0.2, (n - k, 2))], 1)
        p = np.clip(np.concatenate([jit, rnd], 0), 0, 1).astype(np.float32)
        props[i, :n] = p[rs.permutation(n)]
    T = lambda a: torch.from_numpy(a).to(DEV)
    gen = torch.Generator(device=DEV).manual_seed(1)
    rois, tcls, tdel, tmask = L.prepare_det_target(T(props), T(num), T(cls), T(gtn.astype(np.float32)), T(masks), cfg, gen)
    rois, tcls, tdel, tmask = rois.cpu().numpy(), tcls.cpu().numpy(), tdel.cpu().numpy(), tmask.cpu().numpy()
    for i in range(b):
        n = int(num[i])
        used = int((np.abs(rois[i]).sum(1) > 0).sum())
        n_pos = int((tcls[i] > 0).sum())
        # which proposals did the build pick, in output order
        d = np.abs(rois[i][:used, None, :] - props[i][None, :n, :]).sum(2)
        sel = d.argmin(1)
        assert np.all(d[np.arange(used), sel] == 0) and len(set(sel.tolist())) == used
        # candidate sets from the oracle's own arithmetic
        keep, no_crowd = oracle._split_crowd(cls[i], gtn[i].astype(np.float32), props[i][:n])
        gb = gtn[i].astype(np.float32)[keep] if (cls[i] < 0).any() else gtn[i].astype(np.float32)
        iou = oracle.compute_iou(props[i][:n], gb)
        mx = iou.max(1)
        pos_c = np.nonzero(mx >= np.float32(0.5))[0]
        neg_c = np.nonzero((mx < np.float32(0.5)) & no_crowd)[0]
        perm_pos = np.array([int(np.nonzero(pos_c == s)[0][0]) for s in sel[:n_pos]] +
                            [j for j, c in enumerate(pos_c) if c not in set(sel[:n_pos].tolist())], np.int64)
        perm_neg = np.array([int(np.nonzero(neg_c == s)[0][0]) for s in sel[n_pos:]] +
                            [j for j, c in enumerate(neg_c) if c not in set(sel[n_pos:].tolist())], np.int64)
        exp = oracle.generate_roi(cfg, props[i], cls[i], gtn[i].astype(np.float32), masks[i], perm_pos, perm_neg, num_valid=n)
        if exp is None:
            assert used == 0 and n_pos == 0
            continue
        e_rois, e_cls, e_del, e_mask = exp
        assert len(e_rois) == used and int((e_cls > 0).sum()) == n_pos      # sampling rule: counts
        assert np.array_equal(rois[i][:used], e_rois)
        assert np.array_equal(tcls[i][:used], e_cls) and np.all(tcls[i][used:] == 0)
        assert np.allclose(tdel[i][:used], e_del, rtol=1e-5, atol=1e-5) and np.all(tdel[i][used:] == 0)
        assert np.array_equal(tmask[i][:used], e_mask) and np.all(tmask[i][used:] == 0)
        assert n_pos == min(len(pos_c), int(96 * 0.33))
    assert int((tcls[0] > 0).sum()) > 5 and int((tcls[2] > 0).sum()) == 0
