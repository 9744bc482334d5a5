"""The callers either side of the hot-path operators: anchors, the proposal layer around
NMS, RPN / detection target generation and the five detector losses.  Counterpart of
lib/layers.py and tools/box_utils.py of the reference.

Same arithmetic, different execution: everything is batched over the minibatch with
static shapes and validity masks on the GPU -- no per-image Python loops, no per-positive
loops (lib/layers.py:599-604, 892-895, 923-926), no host round trips (the reference reads
counts back with .data[0] / .cpu().numpy() at lib/layers.py:453, 505-507, 866 and
lib/nms/nms_wrapper.py:33).  Random sub-sampling uses the device generator instead of
np.random / torch.randperm on the host (statistically equivalent, not stream-identical).
"""
import numpy as np
import torch
import torch.nn.functional as F

from ._lib import const_tensor
from .nms.pth_nms import nms_sorted
from .roi_align.crop_and_resize import CropAndResizeFunction

EPS_IOU = 10e-20     # tools/box_utils.py:4


# --------------------------------------------------------------------------------------
# anchors (lib/layers.py:9-65) -- executed once, on the host
# --------------------------------------------------------------------------------------
def generate_priors(scales, ratios, shape, feature_stride, anchor_stride):
    scales, ratios = np.meshgrid(np.array(scales), np.array(ratios))
    scales, ratios = scales.flatten(), ratios.flatten()
    heights = scales / np.sqrt(ratios)
    widths = scales * np.sqrt(ratios)
    shifts_y = np.arange(0, shape[0], anchor_stride) * feature_stride
    shifts_x = np.arange(0, shape[1], anchor_stride) * feature_stride
    shifts_x, shifts_y = np.meshgrid(shifts_x, shifts_y)
    box_widths, box_centers_x = np.meshgrid(widths, shifts_x)
    box_heights, box_centers_y = np.meshgrid(heights, shifts_y)
    box_centers = np.stack([box_centers_y, box_centers_x], axis=2).reshape([-1, 2])
    box_sizes = np.stack([box_heights, box_widths], axis=2).reshape([-1, 2])
    return np.concatenate([box_centers - 0.5 * box_sizes, box_centers + 0.5 * box_sizes], axis=1)


def generate_pyramid_priors(scales, ratios, feature_shapes, feature_strides, anchor_stride):
    """[N, (y1, x1, y2, x2)] pixel anchors, level-major (261 888 x 4 at 1024^2)."""
    return np.concatenate([generate_priors(scales[i], ratios, feature_shapes[i], feature_strides[i],
                                           anchor_stride) for i in range(len(scales))], axis=0)


# --------------------------------------------------------------------------------------
# box arithmetic (tools/box_utils.py:7-60, 89-140)
# --------------------------------------------------------------------------------------
def apply_box_deltas(boxes, deltas):
    height = boxes[..., 2] - boxes[..., 0]
    width = boxes[..., 3] - boxes[..., 1]
    center_y = boxes[..., 0] + 0.5 * height
    center_x = boxes[..., 1] + 0.5 * width
    center_y = center_y + deltas[..., 0] * height
    center_x = center_x + deltas[..., 1] * width
    height = height * torch.exp(deltas[..., 2])
    width = width * torch.exp(deltas[..., 3])
    y1 = center_y - 0.5 * height
    x1 = center_x - 0.5 * width
    return torch.stack([y1, x1, y1 + height, x1 + width], dim=-1)


def clip_boxes(boxes, window):
    """window = (y1, x1, y2, x2) python floats."""
    return torch.stack([boxes[..., 0].clamp(window[0], window[2]), boxes[..., 1].clamp(window[1], window[3]),
                        boxes[..., 2].clamp(window[0], window[2]), boxes[..., 3].clamp(window[1], window[3])], -1)


def box_refinement(box, gt_box):
    height = box[..., 2] - box[..., 0]
    width = box[..., 3] - box[..., 1]
    center_y = box[..., 0] + 0.5 * height
    center_x = box[..., 1] + 0.5 * width
    gt_height = gt_box[..., 2] - gt_box[..., 0]
    gt_width = gt_box[..., 3] - gt_box[..., 1]
    gt_center_y = gt_box[..., 0] + 0.5 * gt_height
    gt_center_x = gt_box[..., 1] + 0.5 * gt_width
    dy = (gt_center_y - center_y) / height
    dx = (gt_center_x - center_x) / width
    dh = torch.log(gt_height / height)
    dw = torch.log(gt_width / width)
    return torch.stack([dy, dx, dh, dw], dim=-1)


def bbox_overlaps(boxes1, boxes2):
    """IoU [.., N, M] of boxes1 [.., N, 4] vs boxes2 [.., M, 4] (no +1; union + 1e-19)."""
    b1 = boxes1.unsqueeze(-2)
    b2 = boxes2.unsqueeze(-3)
    y1 = torch.maximum(b1[..., 0], b2[..., 0])
    x1 = torch.maximum(b1[..., 1], b2[..., 1])
    y2 = torch.minimum(b1[..., 2], b2[..., 2])
    x2 = torch.minimum(b1[..., 3], b2[..., 3])
    inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0)
    a1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    a2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    return inter / (a1 + a2 - inter + EPS_IOU)


# --------------------------------------------------------------------------------------
# proposal layer (lib/layers.py:71-139)
# --------------------------------------------------------------------------------------
PRE_NMS_KERNEL_LIMIT = 8192       # fi_proposal_candidates sorts its winners in 64 KB of LDS
import os as _os
PROPOSAL_MULTI_WG = _os.environ.get("FI_PROPOSAL_MULTI_WG", "1") != "0"      # A/B switch: fi_proposal_candidates_ws
_PROPOSAL_WS = {}                 # device -> workspace of the multi-workgroup selection (one per device; stream-ordered use)


def _proposal_candidates_tensors(probs, deltas, anchors, extra_dets, pre_nms_limit, std, height, width):
    """fi_proposal_candidates as tensor operations on the device, for a PRE_NMS_LIMIT the kernel's LDS sort does not hold
    (> 8192; the reference's own full sort, lib/layers.py:99-127, has no such bound): the same candidates in the same
    order (descending score, ties to the lower index, external rows ahead of anchors), the same float32 operation order
    for deltas * BBOX_STD_DEV, apply_box_deltas and clip_boxes (tools/box_utils.py:7-60)."""
    b, A = probs.size(0), probs.size(1)
    scores = probs[:, :, 1]
    E = 0 if extra_dets is None else extra_dets.size(1)
    if E:
        scores = torch.cat((extra_dets[:, :, 4], scores), 1)
    val, order = torch.sort(scores, dim=1, descending=True, stable=True)
    val, order = val[:, :pre_nms_limit], order[:, :pre_nms_limit]
    a_idx = (order - E).clamp(min=0)
    box = anchors[a_idx]                                                     # [b, K, 4]
    d = torch.gather(deltas, 1, a_idx.unsqueeze(2).expand(-1, -1, 4)) * torch.tensor(list(std), device=probs.device).view(1, 1, 4)
    h = box[..., 2] - box[..., 0]
    w = box[..., 3] - box[..., 1]
    cy = box[..., 0] + 0.5 * h
    cx = box[..., 1] + 0.5 * w
    cy = cy + d[..., 0] * h
    cx = cx + d[..., 1] * w
    h = h * torch.exp(d[..., 2])
    w = w * torch.exp(d[..., 3])
    y1 = cy - 0.5 * h
    x1 = cx - 0.5 * w
    rows = torch.stack((y1, x1, y1 + h, x1 + w), 2)
    if E:
        ext = torch.gather(extra_dets[:, :, :4], 1, order.clamp(max=E - 1).unsqueeze(2).expand(-1, -1, 4))
        rows = torch.where((order < E).unsqueeze(2), ext, rows)
    lim = torch.tensor([height, width, height, width], device=probs.device).view(1, 1, 4)
    rows = torch.minimum(torch.maximum(rows, torch.zeros_like(rows)), lim)
    return torch.cat((rows, val.unsqueeze(2)), 2).contiguous()


def proposal_layer(inputs, proposal_count, nms_threshold, priors, config, extra_dets=None):
    """rpn_probs [b, A, 2], rpn_bbox [b, A, 4] -> normalised proposals [b, proposal_count, 4]
    (zero rows past each image's count) and the per-image counts [b] (int32, on the GPU).

    Three launches: fi_proposal_candidates (selection of the PRE_NMS_LIMIT best foreground scores in descending
    order -- ties: lower anchor index --, deltas * BBOX_STD_DEV, apply_box_deltas, clip_boxes: lib/layers.py:99-127
    as one kernel, no sort of all 261 888 anchors), fi_nms_sorted, fi_proposal_gather (kept boxes / image size).
    The reference truncates every image to the shortest keep list, which needs the counts on
    the host (lib/nms/nms_wrapper.py:29-33); here shapes are static and the count
    travels with the tensor.  `extra_dets` [b, E, 5] = (y1, x1, y2, x2, score) in pixels: external proposals that
    compete with the RPN's candidates on their scores (precomputed proposals; the synthetic benchmark's
    object-like boxes, synthetic.SyntheticProposals) -- None in normal use.
    """
    import ctypes
    from . import _lib
    probs, deltas = inputs[0].detach(), inputs[1].detach()
    _lib.require_cuda(probs, deltas)
    L = _lib.load()
    probs = probs.contiguous().float()
    deltas = deltas.contiguous().float()
    anchors = priors.to(probs.device).contiguous().float()
    b, A = probs.size(0), probs.size(1)
    E = 0
    if extra_dets is not None:
        extra_dets = extra_dets.detach().contiguous().float()
        E = extra_dets.size(1)
    pre_nms_limit = min(config.RPN.PRE_NMS_LIMIT, A + E)
    height, width = float(config.DATA.IMAGE_SHAPE[0]), float(config.DATA.IMAGE_SHAPE[1])
    if pre_nms_limit > PRE_NMS_KERNEL_LIMIT:
        dets = _proposal_candidates_tensors(probs, deltas, anchors, extra_dets, pre_nms_limit,
                                            [float(v) for v in config.DATA.BBOX_STD_DEV], height, width)
    else:
        dets = torch.empty((b, pre_nms_limit, 5), device=probs.device, dtype=torch.float32)
        std = (ctypes.c_float * 4)(*[float(v) for v in config.DATA.BBOX_STD_DEV])
        with torch.cuda.device(probs.device):
            if PROPOSAL_MULTI_WG:
                # eight small launches with the state in a workspace (32 workgroups per image for the radix passes and the
                # compaction) instead of one workgroup per image walking every score four times
                need = int(L.fi_proposal_workspace_bytes(b))
                ws = _PROPOSAL_WS.get(probs.device)
                if ws is None or ws.numel() < need:
                    ws = _PROPOSAL_WS[probs.device] = torch.empty(need, device=probs.device, dtype=torch.uint8)
                ws.record_stream(torch.cuda.current_stream(probs.device))
                _lib.check(L.fi_proposal_candidates_ws(_lib.ptr(probs), probs.size(2), 1, _lib.ptr(deltas), _lib.ptr(anchors),
                                                       _lib.ptr(extra_dets), b, A, E, pre_nms_limit, std, height, width,
                                                       _lib.ptr(dets), _lib.ptr(ws), need, _lib.current_stream()),
                           "fi_proposal_candidates_ws")
            else:
                _lib.check(L.fi_proposal_candidates(_lib.ptr(probs), probs.size(2), 1, _lib.ptr(deltas), _lib.ptr(anchors),
                                                    _lib.ptr(extra_dets), b, A, E, pre_nms_limit, std, height, width,
                                                    _lib.ptr(dets), _lib.current_stream()), "fi_proposal_candidates")
    if _lib.TAP is not None:
        _lib.TAP("proposal_candidates", probs=probs, deltas=deltas, anchors=anchors, extra=extra_dets, dets=dets)
    keep, num = nms_sorted(dets, nms_threshold, max_keep=proposal_count)
    out = torch.empty((b, proposal_count, 4), device=probs.device, dtype=torch.float32)
    with torch.cuda.device(probs.device):
        _lib.check(L.fi_proposal_gather(_lib.ptr(dets), pre_nms_limit, 5, _lib.ptr(keep), keep.size(1), _lib.ptr(num), b,
                                        proposal_count, height, width, _lib.ptr(out), _lib.current_stream()),
                   "fi_proposal_gather")
    return out, num


# --------------------------------------------------------------------------------------
# Detection layer, inference (lib/layers.py:664-802), batched, static shapes
# --------------------------------------------------------------------------------------
_CLASS_STRIDE = 8192.0      # > any pixel coordinate + 1; class*stride and the sums stay exact in fp32


def detection_layer(rois, probs, deltas, windows, config, feature=None):
    """rois [bs, N, 4] normalised, probs [bs*N, K], deltas [bs*N, K, 4], windows [bs, 4] (pixels)
    -> detections [bs, DET_MAX_INSTANCES, (y1, x1, y2, x2, class_id, score)] in pixels, zero
    padded, sorted by descending score (and the matching rows of `feature` if given).

    The reference loops over images and over the classes present, sorting and calling NMS per
    class (conduct_nms, :664-718).  Here every image is ONE sorted NMS launch: boxes are rounded
    pixel integers (:765), so shifting a box by class_id * 8192 along x keeps every coordinate,
    area and intersection exact in fp32 -- IoUs inside a class are unchanged and IoUs across
    classes are 0, which is precisely per-class NMS.  Rows failing the filter (:768-769) are
    sorted last and collapse onto one dummy box.  No host synchronisation."""
    bs, N = rois.size(0), rois.size(1)
    max_det = int(config.TEST.DET_MAX_INSTANCES)
    class_scores, class_ids = torch.max(probs, dim=1)
    idx = torch.arange(class_ids.size(0), device=probs.device)
    std = const_tensor(config.DATA.BBOX_STD_DEV, probs.device).view(1, 4)
    deltas_specific = deltas[idx, class_ids] * std
    refined = apply_box_deltas(rois.reshape(1, -1, 4), deltas_specific.unsqueeze(0)).view(bs, N, 4)
    h, w = float(config.DATA.IMAGE_SHAPE[0]), float(config.DATA.IMAGE_SHAPE[1])
    refined = refined * const_tensor([h, w, h, w], probs.device)
    win = windows.to(refined.dtype).view(bs, 1, 4)
    refined = torch.stack([torch.maximum(torch.minimum(refined[..., 0], win[..., 2]), win[..., 0]),
                           torch.maximum(torch.minimum(refined[..., 1], win[..., 3]), win[..., 1]),
                           torch.maximum(torch.minimum(refined[..., 2], win[..., 2]), win[..., 0]),
                           torch.maximum(torch.minimum(refined[..., 3], win[..., 3]), win[..., 1])], 2)
    refined = torch.round(refined)
    class_ids = class_ids.view(bs, N)
    class_scores = class_scores.view(bs, N)
    area = (refined[..., 0] - refined[..., 2]) * (refined[..., 1] - refined[..., 3])
    ok = (class_ids > 0) & (class_scores >= config.TEST.DET_MIN_CONFIDENCE) & (area > 0)
    n_ok = ok.sum(1)
    key = torch.where(ok, class_scores, torch.full_like(class_scores, -1.0))
    key, order = torch.sort(key, dim=1, descending=True, stable=True)
    g = lambda t: torch.gather(t, 1, order)
    ok_s, cls_s = g(ok), g(class_ids)
    box_s = torch.gather(refined, 1, order.unsqueeze(2).expand(-1, -1, 4))
    shift = torch.where(ok_s, cls_s.to(box_s.dtype) * _CLASS_STRIDE, torch.zeros_like(key))
    dummy = const_tensor([0.0, 0.0, 1.0, 1.0], probs.device)     # class-0 slot: no real box lives there
    nms_in = torch.where(ok_s.unsqueeze(2), box_s, dummy.expand_as(box_s)).clone()
    nms_in[..., 1] += shift
    nms_in[..., 3] += shift
    dets = torch.cat((nms_in, key.unsqueeze(2)), 2)
    keep, num = nms_sorted(dets, config.TEST.DET_NMS_THRESHOLD, max_keep=max_det + 1)
    keep = keep[:, :max_det]
    if keep.size(1) < max_det:
        keep = F.pad(keep, (0, max_det - keep.size(1)))
    slot = torch.arange(max_det, device=keep.device).unsqueeze(0)
    valid = (slot < num.unsqueeze(1)) & (keep < n_ok.unsqueeze(1))
    keep = torch.where(valid, keep, torch.zeros_like(keep))
    out_box = torch.gather(box_s, 1, keep.unsqueeze(2).expand(-1, -1, 4))
    out = torch.cat((out_box, torch.gather(cls_s, 1, keep).unsqueeze(2).to(out_box.dtype),
                     torch.gather(key, 1, keep).unsqueeze(2)), 2) * valid.unsqueeze(2).to(out_box.dtype)
    if feature is None:
        return out
    src = torch.gather(order, 1, keep)                                   # row of the RoI inside its image
    feat = feature.view(bs, N, -1)
    out_feat = torch.gather(feat, 1, src.unsqueeze(2).expand(-1, -1, feat.size(2))) * valid.unsqueeze(2).to(feat.dtype)
    return out, out_feat


# --------------------------------------------------------------------------------------
# RPN targets (lib/layers.py:439-658), batched
# --------------------------------------------------------------------------------------
def _random_subset(mask, limit, kmax, key):
    """Keep at most limit[b] of the True entries of mask [b, n], uniformly at random: the entries with the largest
    `key` [b, n] (uniform in [1, 2)).  limit: python int or int tensor [b]; kmax: static upper bound of limit."""
    b, n = mask.shape
    key = torch.where(mask, key, torch.zeros_like(key))
    k = min(n, int(kmax))
    top, idx = torch.topk(key, k, dim=1, sorted=True)
    lim = limit if torch.is_tensor(limit) else torch.full((b,), int(limit), device=mask.device)
    take = (torch.arange(k, device=mask.device).unsqueeze(0) < lim.unsqueeze(1)) & (top > 0)
    out = torch.zeros_like(mask)
    out.scatter_(1, idx, take)
    return out


import os as _os
# CUDA: fi_rpn_targets / fi_detection_targets (csrc/targets.hip); False: the tensor formulation (A/B switch)
TARGET_KERNELS = _os.environ.get('FI_TARGET_KERNELS', '1') != '0'


def prepare_rpn_target(anchors, gt_class_ids, gt_boxes, config, generator=None):
    """anchors [A,4] pixels; gt_class_ids [b,G] (0 = padding, <0 = crowd); gt_boxes [b,G,4]
    pixels.  Returns target_rpn_match [b,A] in {1,-1,0} and target_rpn_deltas [b,A,4] (the
    refinement of every anchor towards its best GT, already divided by BBOX_STD_DEV; only
    rows with match == 1 are used).  The reference packs the positives' deltas into
    [b, 256, 4] in anchor order (:599-604) and the loss re-aligns them (:854-861); keeping
    them per anchor is the same pairing without the packing.
    The random sub-samples are drawn as one uniform key per anchor (positives first, then negatives: two draws from
    `generator`); which anchors are kept is a function of the keys (rpn_target_from_keys)."""
    anchors = anchors.to(gt_boxes.device)
    b, A = gt_class_ids.size(0), anchors.size(0)
    key_pos = torch.rand((b, A), device=gt_boxes.device, generator=generator) + 1.0
    key_neg = torch.rand((b, A), device=gt_boxes.device, generator=generator) + 1.0
    return rpn_target_from_keys(anchors, gt_class_ids, gt_boxes, config, key_pos, key_neg)


def rpn_target_from_keys(anchors, gt_class_ids, gt_boxes, config, key_pos, key_neg, kernels=None):
    """prepare_rpn_target given the sampling keys.  CUDA: two kernels (fi_rpn_targets) instead of ~140 framework
    launches; the tensor formulation below is the CPU path and the kernels' test reference (same results bit for bit
    when no two candidate keys tie at a selection boundary)."""
    use = TARGET_KERNELS if kernels is None else kernels
    G = gt_class_ids.size(1)
    if use and gt_boxes.is_cuda and G <= 256 and 2 <= config.RPN.TRAIN_ANCHORS_PER_IMAGE <= 4096:
        return _rpn_target_kernels(anchors, gt_class_ids, gt_boxes, config, key_pos, key_neg)
    return _rpn_target_tensors(anchors, gt_class_ids, gt_boxes, config, key_pos, key_neg)


def _rpn_target_kernels(anchors, gt_class_ids, gt_boxes, config, key_pos, key_neg):
    import ctypes
    from . import _lib
    L = _lib.load()
    dev = gt_boxes.device
    b, G = gt_class_ids.shape
    A = anchors.size(0)
    n_total = int(config.RPN.TRAIN_ANCHORS_PER_IMAGE)
    anchors = anchors.contiguous().float()
    ids = gt_class_ids.to(torch.int64).contiguous()
    boxes = gt_boxes.contiguous().float()
    match = torch.empty((b, A), device=dev, dtype=torch.float32)
    deltas = torch.empty((b, A, 4), device=dev, dtype=torch.float32)
    row_image = torch.empty((b * n_total,), device=dev, dtype=torch.int64)
    row_anchor = torch.empty((b * n_total,), device=dev, dtype=torch.int64)
    ws = torch.empty((int(L.fi_rpn_targets_workspace_bytes(b, A, G)) + 3) // 4, device=dev, dtype=torch.float32)
    std = (ctypes.c_float * 4)(*[float(v) for v in config.DATA.BBOX_STD_DEV])
    with torch.cuda.device(dev):
        _lib.check(L.fi_rpn_targets(_lib.ptr(anchors), _lib.ptr(ids), _lib.ptr(boxes), _lib.ptr(key_pos.contiguous()),
                                    _lib.ptr(key_neg.contiguous()), b, A, G, float(config.RPN.TARGET_NEG_THRES),
                                    float(config.RPN.TARGET_POS_THRES), n_total, std, _lib.ptr(match), _lib.ptr(deltas),
                                    _lib.ptr(row_image), _lib.ptr(row_anchor), _lib.ptr(ws), _lib.current_stream()),
                   "fi_rpn_targets")
    match._fi_rows = (n_total, row_image, row_anchor)        # select_rpn_rows: the kernel has listed them already
    return match, deltas


def _rpn_target_tensors(anchors, gt_class_ids, gt_boxes, config, key_pos, key_neg):
    b, G = gt_class_ids.shape
    A = anchors.size(0)
    valid_gt = gt_class_ids > 0
    crowd = gt_class_ids < 0
    # IoU as [b, G, A]: the reduction over the 261 888 anchors (each GT's best anchor) then runs along the
    # contiguous axis, and the per-anchor max over <= 100 GTs is an elementwise sweep (the [b, A, G] form
    # spent 0.9 ms in one strided argmax)
    overlaps = bbox_overlaps(gt_boxes, anchors.unsqueeze(0))                 # [b, G, A]
    ov_gt = torch.where(valid_gt.unsqueeze(2), overlaps, torch.zeros_like(overlaps))
    no_crowd = torch.where(crowd.unsqueeze(2), overlaps, torch.zeros_like(overlaps)).amax(1) < 0.001
    iou_max, iou_argmax = ov_gt.max(dim=1)                                    # [b, A]
    match = torch.zeros(b, A, device=gt_boxes.device)
    match = torch.where((iou_max < config.RPN.TARGET_NEG_THRES) & no_crowd, -torch.ones_like(match), match)
    # every valid GT claims its best anchor (:495-497)
    gt_best = ov_gt.argmax(dim=2)                                             # [b, G]
    claim = torch.zeros(b, A, device=gt_boxes.device, dtype=torch.int32)
    claim.scatter_add_(1, gt_best, valid_gt.to(torch.int32))      # padded GTs add 0
    claim = claim > 0
    match = torch.where(claim, torch.ones_like(match), match)
    match = torch.where(iou_max >= config.RPN.TARGET_POS_THRES, torch.ones_like(match), match)
    # balance: at most half positives, negatives fill the rest (:512-548)
    n_total = config.RPN.TRAIN_ANCHORS_PER_IMAGE
    pos = _random_subset(match == 1, n_total // 2, n_total // 2, key_pos)
    n_pos = pos.sum(1)
    neg = _random_subset(match == -1, (n_total - n_pos).clamp(min=0), n_total, key_neg)
    match = pos.float() - neg.float()
    gt_for_anchor = torch.gather(gt_boxes, 1, iou_argmax.unsqueeze(2).expand(-1, -1, 4))
    deltas = box_refinement(anchors.unsqueeze(0).expand(b, -1, -1), gt_for_anchor)
    deltas = deltas / const_tensor(config.DATA.BBOX_STD_DEV, deltas.device)
    deltas = torch.where(pos.unsqueeze(2), deltas, torch.zeros_like(deltas))
    return match, deltas


# --------------------------------------------------------------------------------------
# detection targets (lib/layers.py:224-433), batched
# --------------------------------------------------------------------------------------
def prepare_det_target(proposals, num_proposals, gt_class_ids, gt_boxes, gt_masks, config, generator=None):
    """proposals [b,P,4] normalised (zero rows past num_proposals[b]); gt_* zero padded,
    gt_boxes normalised, gt_masks [b,G,56,56] mini-masks.
    Returns rois [b,R,4], target_class_ids [b,R] int32, target_deltas [b,R,4],
    target_mask [b,R,28,28] -- positives first, then negatives, then zero padding.
    The two random rankings are one uniform key per proposal each (positives, then negatives: two draws from
    `generator`); see det_target_from_keys."""
    b, P, _ = proposals.shape
    key_pos = torch.rand((b, P), device=proposals.device, generator=generator) + 1.0
    key_neg = torch.rand((b, P), device=proposals.device, generator=generator) + 1.0
    return det_target_from_keys(proposals, num_proposals, gt_class_ids, gt_boxes, gt_masks, config, key_pos, key_neg)


def det_target_from_keys(proposals, num_proposals, gt_class_ids, gt_boxes, gt_masks, config, key_pos, key_neg, kernels=None):
    """prepare_det_target given the ranking keys.  CUDA: one kernel (fi_detection_targets: IoU, ranking by an LDS sort,
    slots, class ids, refinements, mask-crop boxes) + the mask-target crop, instead of ~130 framework launches."""
    b, P, _ = proposals.shape
    G = gt_class_ids.size(1)
    dev = proposals.device
    R = config.ROIS.TRAIN_ROIS_PER_IMAGE
    mh, mw = config.MRCNN.MASK_SHAPE
    use = TARGET_KERNELS if kernels is None else kernels
    pos_cap = int(R * config.ROIS.ROI_POSITIVE_RATIO)
    ratio = 1.0 / config.ROIS.ROI_POSITIVE_RATIO
    if use and proposals.is_cuda and P <= 2048 and G <= 256:
        import ctypes
        from . import _lib
        L = _lib.load()
        rois = torch.empty((b, R, 4), device=dev, dtype=torch.float32)
        target_class_ids = torch.empty((b, R), device=dev, dtype=torch.int32)
        target_deltas = torch.empty((b, R, 4), device=dev, dtype=torch.float32)
        boxes = torch.empty((b, R, 4), device=dev, dtype=torch.float32)
        box_ids = torch.empty((b, R), device=dev, dtype=torch.int32)
        is_pos_f = torch.empty((b, R), device=dev, dtype=torch.float32)
        std = (ctypes.c_float * 4)(*[float(v) for v in config.DATA.BBOX_STD_DEV])
        props = proposals.contiguous().float()
        with torch.cuda.device(dev):
            _lib.check(L.fi_detection_targets(_lib.ptr(props), _lib.ptr(num_proposals.to(torch.int64).contiguous()),
                                              _lib.ptr(gt_class_ids.to(torch.int64).contiguous()),
                                              _lib.ptr(gt_boxes.contiguous().float()), _lib.ptr(key_pos.contiguous()),
                                              _lib.ptr(key_neg.contiguous()), b, P, G, R, pos_cap, ratio,
                                              1 if config.MRCNN.USE_MINI_MASK else 0, std, _lib.ptr(rois),
                                              _lib.ptr(target_class_ids), _lib.ptr(target_deltas), _lib.ptr(boxes),
                                              _lib.ptr(box_ids), _lib.ptr(is_pos_f), _lib.current_stream()),
                       "fi_detection_targets")
        masks = CropAndResizeFunction(mh, mw)(gt_masks.reshape(b * G, 1, gt_masks.size(2), gt_masks.size(3)).float(),
                                              boxes.reshape(-1, 4), box_ids.reshape(-1))
        target_mask = torch.round(masks.view(b, R, mh, mw)) * is_pos_f.view(b, R, 1, 1)
        return rois, target_class_ids, target_deltas, target_mask
    valid_prop = torch.arange(P, device=dev).unsqueeze(0) < num_proposals.unsqueeze(1)
    valid_gt = gt_class_ids > 0
    crowd = gt_class_ids < 0
    overlaps = bbox_overlaps(proposals, gt_boxes)                              # [b, P, G]
    ov_gt = torch.where(valid_gt.unsqueeze(1), overlaps, torch.zeros_like(overlaps))
    no_crowd = torch.where(crowd.unsqueeze(1), overlaps, torch.zeros_like(overlaps)).amax(2) < 0.001
    roi_iou_max, assign = ov_gt.max(dim=2)
    pos_bool = (roi_iou_max >= 0.5) & valid_prop
    neg_bool = (roi_iou_max < 0.5) & no_crowd & valid_prop

    def ranked(mask, k, key):
        key = torch.where(mask, key, torch.zeros_like(key))
        top, idx = torch.topk(key, min(k, P), dim=1, sorted=True)
        return idx, (top > 0).sum(1)

    pos_idx, n_pos_avail = ranked(pos_bool, pos_cap, key_pos)
    pos_cnt = n_pos_avail.clamp(max=pos_cap)
    neg_want = torch.floor(ratio * pos_cnt.double() - pos_cnt.double()).long()     # int(r*pos - pos)
    neg_idx, n_neg_avail = ranked(neg_bool, R, key_neg)
    neg_cnt = torch.minimum(neg_want, n_neg_avail).clamp(max=R)
    neg_cnt = torch.minimum(neg_cnt, (R - pos_cnt))

    slot = torch.arange(R, device=dev).unsqueeze(0).expand(b, R)
    is_pos = slot < pos_cnt.unsqueeze(1)
    is_neg = (~is_pos) & (slot < (pos_cnt + neg_cnt).unsqueeze(1))
    pi = torch.gather(pos_idx, 1, slot.clamp(max=pos_idx.size(1) - 1))
    ni = torch.gather(neg_idx, 1, (slot - pos_cnt.unsqueeze(1)).clamp(min=0, max=neg_idx.size(1) - 1))
    sel = torch.where(is_pos, pi, ni)
    used = (is_pos | is_neg)
    rois = torch.gather(proposals, 1, sel.unsqueeze(2).expand(-1, -1, 4)) * used.unsqueeze(2).float()

    roi_assign = torch.gather(assign, 1, sel)
    roi_gt_boxes = torch.gather(gt_boxes, 1, roi_assign.unsqueeze(2).expand(-1, -1, 4))
    cls = torch.gather(gt_class_ids, 1, roi_assign)
    target_class_ids = torch.where(is_pos, cls, torch.zeros_like(cls)).to(torch.int32)
    deltas = box_refinement(rois, roi_gt_boxes) / const_tensor(config.DATA.BBOX_STD_DEV, dev)
    target_deltas = torch.where(is_pos.unsqueeze(2), deltas, torch.zeros_like(deltas))

    # mask targets: crop the GT mini-mask with the RoI expressed in mini-mask space (:301-322)
    boxes = rois
    if config.MRCNN.USE_MINI_MASK:
        gh = (roi_gt_boxes[..., 2] - roi_gt_boxes[..., 0])
        gw = (roi_gt_boxes[..., 3] - roi_gt_boxes[..., 1])
        boxes = torch.stack([(rois[..., 0] - roi_gt_boxes[..., 0]) / gh, (rois[..., 1] - roi_gt_boxes[..., 1]) / gw,
                             (rois[..., 2] - roi_gt_boxes[..., 0]) / gh, (rois[..., 3] - roi_gt_boxes[..., 1]) / gw], -1)
    boxes = torch.where(is_pos.unsqueeze(2), boxes, torch.zeros_like(boxes))
    box_ids = (roi_assign + torch.arange(b, device=dev).unsqueeze(1) * G).to(torch.int32)
    masks = CropAndResizeFunction(mh, mw)(gt_masks.reshape(b * G, 1, gt_masks.size(2), gt_masks.size(3)).float(),
                                          boxes.reshape(-1, 4), box_ids.reshape(-1))
    masks = torch.round(masks.view(b, R, mh, mw))
    target_mask = torch.where(is_pos.view(b, R, 1, 1), masks, torch.zeros_like(masks))
    return rois.detach(), target_class_ids, target_deltas.detach(), target_mask.detach()


# --------------------------------------------------------------------------------------
# losses (lib/layers.py:808-934) -- masked means instead of nonzero() gathers
# --------------------------------------------------------------------------------------
def _masked_mean(values, mask, per_item=1):
    n = mask.sum()
    return (values * mask).sum() / (n * per_item).clamp(min=1)


def compute_rpn_class_loss(target_rpn_match, rpn_class_logits):
    anchor_class = (target_rpn_match == 1).long()
    ce = F.cross_entropy(rpn_class_logits.reshape(-1, 2), anchor_class.reshape(-1), reduction='none')
    return _masked_mean(ce, (target_rpn_match != 0).reshape(-1).float())


def select_rpn_rows(target_rpn_match, rows_per_image):
    """(image [R], anchor [R], valid [R]), R = b * rows_per_image: the anchors with target_rpn_match != 0 in (image,
    anchor) order, padded with invalid rows -- prepare_rpn_target samples at most RPN.TRAIN_ANCHORS_PER_IMAGE of them
    per image, so the shape is static and nothing is read back."""
    b = target_rpn_match.size(0)
    rows = getattr(target_rpn_match, "_fi_rows", None)
    if rows is not None and rows[0] == rows_per_image:          # fi_rpn_targets listed them (per image, -1 padded)
        if rows[1].is_cuda:
            # allocated on the stream the targets ran on (run_on_side_stream marks its OUTPUTS for the consumer's stream;
            # these two ride on a python attribute): mark them for the stream that reads them from here on
            cur = torch.cuda.current_stream(rows[1].device)
            rows[1].record_stream(cur)
            rows[2].record_stream(cur)
        return rows[1], rows[2], rows[1] >= 0
    sel = torch.nonzero_static(target_rpn_match != 0, size=b * rows_per_image, fill_value=-1)
    return sel[:, 0], sel[:, 1], sel[:, 0] >= 0


def compute_rpn_losses_on_rows(target_rpn_match, target_rpn_deltas, image, anchor, valid, logits, bbox):
    """compute_rpn_class_loss and compute_rpn_bbox_loss on the rows of select_rpn_rows (every anchor with a non-zero
    match is one of them, every other anchor contributes exactly zero to either loss): same values."""
    i, a = image.clamp(min=0), anchor.clamp(min=0)
    match = torch.where(valid, target_rpn_match[i, a], torch.zeros_like(target_rpn_match[i, a]))
    ce = F.cross_entropy(logits, (match == 1).long(), reduction='none')
    cls_loss = _masked_mean(ce, (match != 0).float())
    pos = (match == 1).float().unsqueeze(1)
    l = F.smooth_l1_loss(bbox, target_rpn_deltas[i, a], reduction='none')
    return cls_loss, (l * pos).sum() / (pos.sum() * 4).clamp(min=1)


def compute_rpn_bbox_loss(target_rpn_deltas, target_rpn_match, rpn_bbox):
    pos = (target_rpn_match == 1).float().unsqueeze(2)
    l = F.smooth_l1_loss(rpn_bbox, target_rpn_deltas, reduction='none')
    return (l * pos).sum() / (pos.sum() * 4).clamp(min=1)


def compute_mrcnn_class_loss(target_class_ids, pred_class_logits):
    has_fg = (target_class_ids.sum() != 0).float()
    loss = F.cross_entropy(pred_class_logits.reshape(-1, pred_class_logits.size(-1)),
                           target_class_ids.long().reshape(-1))
    return loss * has_fg


def compute_mrcnn_bbox_loss(target_bbox, target_class_ids, pred_bbox):
    """pred_bbox [b, R, num_classes, 4]: positives only, class-specific row."""
    cls = target_class_ids.long()
    pos = (cls > 0).float().unsqueeze(2)
    pred = torch.gather(pred_bbox, 2, cls.view(cls.size(0), cls.size(1), 1, 1).expand(-1, -1, 1, 4)).squeeze(2)
    l = F.smooth_l1_loss(pred, target_bbox, reduction='none')
    return (l * pos).sum() / (pos.sum() * 4).clamp(min=1)


def compute_mrcnn_mask_loss_unshuffled(target_masks, target_class_ids, pred_u, from_logits=False):
    """compute_mrcnn_mask_loss for the mask head's un-shuffled output pred_u [b, R, 2, 2, K, h, w]
    (pred[.., k, 2y+a, 2x+b] == pred_u[.., a, b, k, y, x]): the class channel is gathered first and only
    that [b, R, 2, 2, h, w] slice is pixel-shuffled.  Same value and gradient.  from_logits: pred_u holds the
    mask head's logits (Mask.forward(activate=False)) and the sigmoid is applied to the gathered slice."""
    cls = target_class_ids.long()
    b, R, _, _, K, h, w = pred_u.shape
    idx = cls.view(b, R, 1, 1, 1, 1, 1).expand(-1, -1, 2, 2, 1, h, w)
    pred = torch.gather(pred_u, 4, idx).squeeze(4)                                  # [b, R, 2, 2, h, w]
    if from_logits:
        pred = torch.sigmoid(pred)
    pred = pred.permute(0, 1, 4, 2, 5, 3).reshape(b, R, 2 * h, 2 * w)
    pos = (cls > 0).float().view(b, R, 1, 1)
    l = F.binary_cross_entropy(pred, target_masks, reduction='none')
    return (l * pos).sum() / (pos.sum() * 4 * h * w).clamp(min=1)


def compute_mrcnn_mask_loss_selected(target_masks, target_class_ids, logits):
    """compute_mrcnn_mask_loss for logits [b, R, 2, 2, h, w] that ALREADY are the target class's channel of the mask
    head's un-shuffled output (Mask.forward(select_class=...)): sigmoid, pixel shuffle, BCE on the positives.  Same
    value and gradient as compute_mrcnn_mask_loss_unshuffled(from_logits=True) on the full output."""
    cls = target_class_ids.long()
    b, R, _, _, h, w = logits.shape
    pred = torch.sigmoid(logits).permute(0, 1, 4, 2, 5, 3).reshape(b, R, 2 * h, 2 * w)
    pos = (cls > 0).float().view(b, R, 1, 1)
    l = F.binary_cross_entropy(pred, target_masks, reduction='none')
    return (l * pos).sum() / (pos.sum() * 4 * h * w).clamp(min=1)


def compute_mrcnn_mask_loss(target_masks, target_class_ids, pred_masks):
    """pred_masks [b, R, num_classes, h, w] probabilities: positives only, class-specific mask."""
    cls = target_class_ids.long()
    b, R, _, h, w = pred_masks.shape
    pos = (cls > 0).float().view(b, R, 1, 1)
    pred = torch.gather(pred_masks, 2, cls.view(b, R, 1, 1, 1).expand(-1, -1, 1, h, w)).squeeze(2)
    l = F.binary_cross_entropy(pred, target_masks, reduction='none')
    return (l * pos).sum() / (pos.sum() * h * w).clamp(min=1)


# --------------------------------------------------------------------------------------
# the five losses in one pass (csrc/losses.hip)
# --------------------------------------------------------------------------------------
LOSS_KERNEL = _os.environ.get('FI_LOSS_KERNEL', '1') != '0'      # A/B switch


class _DetectorLossesFn(torch.autograd.Function):
    """[rpn_class, rpn_bbox, mrcnn_class, mrcnn_bbox, mrcnn_mask] (fi_detector_losses): the kernel leaves every loss's
    gradient with respect to its network output up to a factor 1 / count, so backward is one scaling per tensor."""

    @staticmethod
    def forward(ctx, row_logits, row_bbox, cls_logits, roi_bbox, mask_logits, rpn_match, rpn_deltas, row_image, row_anchor,
                roi_cls, roi_deltas, mask_cls, mask_targets):
        from . import _lib
        L = _lib.load()
        dev = cls_logits.device
        c = lambda t: t.contiguous().float()
        row_logits, row_bbox, cls_logits, roi_bbox, mask_logits = (c(row_logits), c(row_bbox), c(cls_logits), c(roi_bbox),
                                                                   c(mask_logits))
        Rr, N, K = row_logits.size(0), cls_logits.size(0), cls_logits.size(1)
        Nm, h, w = mask_logits.size(0), mask_logits.size(-2), mask_logits.size(-1)
        grads = [torch.empty_like(t) for t in (row_logits, row_bbox, cls_logits, roi_bbox, mask_logits)]
        out = torch.empty(10, device=dev, dtype=torch.float32)
        ws = torch.empty((int(L.fi_detector_losses_workspace_bytes(Rr, N, Nm)) + 3) // 4, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(L.fi_detector_losses(_lib.ptr(c(rpn_match)), _lib.ptr(c(rpn_deltas)),
                                            _lib.ptr(row_image.to(torch.int64).contiguous()),
                                            _lib.ptr(row_anchor.to(torch.int64).contiguous()), _lib.ptr(row_logits),
                                            _lib.ptr(row_bbox), Rr, rpn_match.size(1),
                                            _lib.ptr(roi_cls.to(torch.int32).contiguous()), _lib.ptr(cls_logits),
                                            _lib.ptr(c(roi_deltas)), _lib.ptr(roi_bbox), N, K,
                                            _lib.ptr(mask_cls.to(torch.int32).contiguous()), _lib.ptr(mask_logits),
                                            _lib.ptr(c(mask_targets)), Nm, h, w, *[_lib.ptr(g) for g in grads],
                                            _lib.ptr(out), _lib.ptr(ws), _lib.current_stream()), "fi_detector_losses")
        ctx.save_for_backward(out, *grads)
        return out[:5].clone()

    @staticmethod
    def backward(ctx, g):
        out, g0, g1, g2, g3, g4 = ctx.saved_tensors
        f = g.float() * out[5:]                                   # [5]
        need = ctx.needs_input_grad
        return tuple((G * f[k]) if need[k] else None for k, G in enumerate((g0, g1, g2, g3, g4))) + (None,) * 8


def detector_losses(row_logits, row_bbox, r_img, r_anchor, target_rpn_match, target_rpn_deltas, class_logits, roi_bbox,
                    target_class_ids, target_deltas, mask_logits, mask_ids, mask_targets):
    """The five losses [5] of compute_rpn_losses_on_rows / compute_mrcnn_class_loss / compute_mrcnn_bbox_loss /
    compute_mrcnn_mask_loss_selected in ONE kernel pass (CUDA; None when the inputs do not fit it).  class_logits
    [b, R, K], roi_bbox [b, R, K, 4], mask_logits [b, P, 2, 2, h, w] (target-class channel), mask_targets [b, P, 2h, 2w]."""
    if not (LOSS_KERNEL and class_logits.is_cuda and mask_logits.dim() == 6 and row_logits.dim() == 2):
        return None
    K = class_logits.size(-1)
    return _DetectorLossesFn.apply(row_logits, row_bbox, class_logits.reshape(-1, K), roi_bbox.reshape(-1, K, 4),
                                   mask_logits.reshape(-1, *mask_logits.shape[2:]), target_rpn_match, target_rpn_deltas,
                                   r_img, r_anchor, target_class_ids.reshape(-1), target_deltas.reshape(-1, 4),
                                   mask_ids.reshape(-1), mask_targets.reshape(-1, *mask_targets.shape[2:]))
