"""CPU: host-side logic of the callers around the hot operators (box arithmetic, anchors,
RPN target assignment, losses) against straightforward per-image / per-item restatements
of lib/layers.py and tools/box_utils.py."""
import numpy as np
import torch
import torch.nn.functional as F

from feature_intertwiner_amd import layers as L
from feature_intertwiner_amd.config import make_config


def test_anchor_count_and_layout():
    cfg = make_config("resnet101", 1024)
    pri = L.generate_pyramid_priors(cfg.RPN.ANCHOR_SCALES, cfg.RPN.ANCHOR_RATIOS, cfg.MODEL.BACKBONE_SHAPES,
                                    cfg.MODEL.BACKBONE_STRIDES, cfg.RPN.ANCHOR_STRIDE)
    assert pri.shape == (261888, 4)
    # first anchor: scale 32, ratio 0.5 centred at (0,0): h = 32/sqrt(.5), w = 32*sqrt(.5)
    h, w = 32 / np.sqrt(0.5), 32 * np.sqrt(0.5)
    assert np.allclose(pri[0], [-h / 2, -w / 2, h / 2, w / 2])
    assert np.allclose(pri[3], [-h / 2, 4 - w / 2, h / 2, 4 + w / 2])     # next location, stride 4
    assert np.allclose(pri[-1, [0, 1]] + pri[-1, [2, 3]], [2 * 15 * 64, 2 * 15 * 64])  # last P6 cell centre


def test_box_arithmetic_round_trip():
    g = torch.Generator().manual_seed(0)
    a = torch.rand(3, 50, 2, generator=g) * 500
    boxes = torch.cat([a, a + 5 + torch.rand(3, 50, 2, generator=g) * 300], 2)
    b = torch.rand(3, 50, 2, generator=g) * 500
    gt = torch.cat([b, b + 5 + torch.rand(3, 50, 2, generator=g) * 300], 2)
    d = L.box_refinement(boxes, gt)
    assert torch.allclose(L.apply_box_deltas(boxes, d), gt, rtol=1e-4, atol=1e-2)
    c = L.clip_boxes(torch.tensor([[-5.0, 10.0, 2000.0, 900.0]]), (0.0, 0.0, 1024.0, 1024.0))
    assert c.tolist() == [[0.0, 10.0, 1024.0, 900.0]]
    iou = L.bbox_overlaps(torch.tensor([[0.0, 0.0, 10.0, 10.0]]), torch.tensor([[0.0, 5.0, 10.0, 15.0], [20.0, 20.0, 30.0, 30.0]]))
    assert torch.allclose(iou, torch.tensor([[50.0 / 150.0, 0.0]]))


def _rpn_target_reference(anchors, gt_cls, gt_boxes, cfg):
    """per-image restatement of lib/layers.py:485-509 (before the random sub-sampling)."""
    valid = gt_cls > 0
    ov = L.bbox_overlaps(anchors, gt_boxes[valid])
    m = torch.zeros(anchors.size(0))
    iou_max, arg = ov.max(1)
    m[iou_max < cfg.RPN.TARGET_NEG_THRES] = -1
    m[ov.argmax(0)] = 1
    m[iou_max >= cfg.RPN.TARGET_POS_THRES] = 1
    return m, arg


def test_rpn_targets_match_reference_rules():
    cfg = make_config("resnet101", 256)
    pri = torch.from_numpy(L.generate_pyramid_priors(cfg.RPN.ANCHOR_SCALES, cfg.RPN.ANCHOR_RATIOS,
                                                     cfg.MODEL.BACKBONE_SHAPES, cfg.MODEL.BACKBONE_STRIDES, 1)).float()
    g = torch.Generator().manual_seed(3)
    b, G = 3, 12
    a = torch.rand(b, G, 2, generator=g) * 180
    gt_boxes = torch.cat([a, a + 10 + torch.rand(b, G, 2, generator=g) * 60], 2)
    gt_cls = torch.randint(1, 81, (b, G), generator=g)
    gt_cls[0, 8:] = 0
    gt_boxes[0, 8:] = 0            # zero padding
    match, deltas = L.prepare_rpn_target(pri, gt_cls, gt_boxes, cfg, g)
    assert match.shape == (b, pri.size(0)) and deltas.shape == (b, pri.size(0), 4)
    for i in range(b):
        ref, arg = _rpn_target_reference(pri, gt_cls[i], gt_boxes[i], cfg)
        m = match[i]
        # sub-sampling only ever resets anchors to neutral
        assert torch.all((m == 1) <= (ref == 1)) and torch.all((m == -1) <= (ref == -1))
        n_pos, n_neg = int((m == 1).sum()), int((m == -1).sum())
        assert n_pos == min(int((ref == 1).sum()), 128)
        assert n_pos + n_neg == min(256, n_pos + int((ref == -1).sum()))
        pos = torch.nonzero(m == 1).view(-1)
        valid_gt = gt_boxes[i][gt_cls[i] > 0]
        exp = L.box_refinement(pri[pos], valid_gt[arg[pos]]) / torch.tensor(cfg.DATA.BBOX_STD_DEV)
        assert torch.allclose(deltas[i, pos], exp, rtol=1e-5, atol=1e-5)
        assert torch.all(deltas[i][m != 1] == 0)


def test_losses_match_gather_formulations():
    g = torch.Generator().manual_seed(5)
    b, A = 2, 300
    match = torch.randint(-1, 2, (b, A), generator=g).float()
    logits = torch.randn(b, A, 2, generator=g)
    sel = match != 0
    ref = F.cross_entropy(logits[sel], (match[sel] == 1).long())
    assert torch.allclose(L.compute_rpn_class_loss(match, logits), ref, rtol=1e-5)
    tgt = torch.randn(b, A, 4, generator=g)
    pred = torch.randn(b, A, 4, generator=g)
    ref = F.smooth_l1_loss(pred[match == 1], tgt[match == 1])
    assert torch.allclose(L.compute_rpn_bbox_loss(tgt, match, pred), ref, rtol=1e-5)
    R, K = 40, 7
    cls = torch.randint(0, K, (b, R), generator=g).int()
    cls[:, 20:] = 0
    cl = torch.randn(b, R, K, generator=g)
    assert torch.allclose(L.compute_mrcnn_class_loss(cls, cl), F.cross_entropy(cl.view(-1, K), cls.long().view(-1)))
    pb = torch.randn(b, R, K, 4, generator=g)
    tb = torch.randn(b, R, 4, generator=g)
    idx = torch.nonzero(cls > 0)
    ref = F.smooth_l1_loss(torch.stack([pb[i, j, cls[i, j]] for i, j in idx.tolist()]),
                           torch.stack([tb[i, j] for i, j in idx.tolist()]))
    assert torch.allclose(L.compute_mrcnn_bbox_loss(tb, cls, pb), ref, rtol=1e-5)
    pm = torch.rand(b, R, K, 6, 6, generator=g)
    tm = (torch.rand(b, R, 6, 6, generator=g) > 0.5).float()
    ref = F.binary_cross_entropy(torch.stack([pm[i, j, cls[i, j]] for i, j in idx.tolist()]),
                                 torch.stack([tm[i, j] for i, j in idx.tolist()]))
    assert torch.allclose(L.compute_mrcnn_mask_loss(tm, cls, pm), ref, rtol=1e-5)
    zero = torch.zeros(b, R).int()
    assert L.compute_mrcnn_class_loss(zero, cl).item() == 0
    assert L.compute_mrcnn_bbox_loss(tb, zero, pb).item() == 0


def test_config_and_state_dict_names():
    from feature_intertwiner_amd.model import MaskRCNN
    cfg = make_config("resnet50", 128, 1, 8, ot_L=5)
    m = MaskRCNN(cfg)
    keys = set(m.state_dict().keys())
    # names the reference's checkpoints use (lib/model.py:121-131, lib/config.py LAYER_REGEX)
    for k in ("fpn.C1.0.weight", "fpn.C2.0.conv1.weight", "fpn.C5.0.downsample.0.weight", "fpn.C5.0.bn3.bias",
              "fpn.P5_conv1.bias", "fpn.P5_conv2.1.weight", "rpn.conv_shared.weight", "rpn.conv_class.weight",
              "dev_roi.upsample.0.0.weight", "dev_roi.feat_extract.0.weight", "dev_roi.feat_extract.6.bias",
              "classifier.conv1.weight", "classifier.linear_bbox.weight", "mask.deconv.weight", "mask.conv5.bias",
              "ot_loss.G_net.0.weight", "ot_loss.critic.0.bias"):
        assert k in keys, k
    assert m.classifier.conv1.weight.shape == (1024, 256, 7, 7)
    assert m.mask.deconv.weight.shape == (256, 256, 2, 2)
    assert m.priors.shape == (3 * (32 * 32 + 16 * 16 + 8 * 8 + 4 * 4 + 2 * 2), 4)
    assert not m.training or True


def test_unshuffled_mask_loss_equals_standard_form():
    """compute_mrcnn_mask_loss_unshuffled on the mask head's pre-shuffle layout == the standard loss
    on the shuffled tensor (value and gradient)."""
    from feature_intertwiner_amd import layers as L
    torch.manual_seed(0)
    b, R, K, h, w = 2, 6, 5, 3, 4
    u = torch.rand(b, R, 2, 2, K, h, w).clamp(0.05, 0.95).requires_grad_(True)
    std = u.detach().permute(0, 1, 4, 5, 2, 6, 3).reshape(b, R, K, 2 * h, 2 * w).requires_grad_(True)
    cls = torch.tensor([[1, 3, 0, 0, 4, 2], [2, 0, 0, 1, 1, 0]], dtype=torch.int32)
    tgt = (torch.rand(b, R, 2 * h, 2 * w) > 0.5).float()
    l1 = L.compute_mrcnn_mask_loss_unshuffled(tgt, cls, u)
    l2 = L.compute_mrcnn_mask_loss(tgt, cls, std)
    assert abs(float(l1) - float(l2)) < 1e-6
    l1.backward()
    l2.backward()
    g2 = std.grad.view(b, R, K, h, 2, w, 2).permute(0, 1, 4, 6, 2, 3, 5)
    assert torch.allclose(u.grad, g2, atol=1e-7)


def test_mask_loss_from_logits_equals_sigmoid_before_the_gather():
    """The training path hands the loss the mask head's LOGITS and the sigmoid (lib/sub_module.py:786) is applied
    behind the class gather: same loss value bit for bit, same gradient w.r.t. the logits (non-target channels: 0,
    exactly what the sigmoid -> gather chain gives them)."""
    from feature_intertwiner_amd import layers as L
    torch.manual_seed(1)
    b, R, K, h, w = 2, 6, 5, 3, 4
    z1 = torch.randn(b, R, 2, 2, K, h, w).requires_grad_(True)
    z2 = z1.detach().clone().requires_grad_(True)
    cls = torch.tensor([[1, 3, 0, 0, 4, 2], [2, 0, 0, 1, 1, 0]], dtype=torch.int32)
    tgt = (torch.rand(b, R, 2 * h, 2 * w) > 0.5).float()
    l1 = L.compute_mrcnn_mask_loss_unshuffled(tgt, cls, z1, from_logits=True)
    l2 = L.compute_mrcnn_mask_loss_unshuffled(tgt, cls, torch.sigmoid(z2))
    assert float(l1) == float(l2)
    l1.backward()
    l2.backward()
    assert torch.allclose(z1.grad, z2.grad, atol=1e-8, rtol=1e-6)
