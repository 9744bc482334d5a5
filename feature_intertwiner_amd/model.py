"""MaskRCNN with the feature intertwiner: the caller of every hot-path operator.
Counterpart of lib/model.py of the reference (train path; same sub-module names, so
state dicts line up: fpn.*, rpn.*, dev_roi.*, ot_loss.*, classifier.*, mask.*).
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import conv as _conv
import os as _os
# the mask head's batch whose results nothing reads runs on the second stream (same-box A/B 120.3 -> 117.2 ms/step)
_DEAD_SIDE = _os.environ.get('FI_DEAD_SIDE', '1') != '0'
# the proposal layer on the second stream, under the Dev stage's make-up convolutions (A/B switch)
_PROPOSAL_SIDE = _os.environ.get('FI_PROPOSAL_SIDE', '1') != '0'
# the 14 x 14 crops come back with the positive RoI slots first: the mask head's two batches are views (A/B switch)
_MASK_FRONT = _os.environ.get('FI_MASK_FRONT', '1') != '0'
from ._lib import const_tensor
from .conv import conv_precision, prepare_step, set_conv_precision
from .intertwiner import FeatureBuffer, meta_loss
from .layers import (compute_mrcnn_bbox_loss, compute_mrcnn_class_loss, compute_mrcnn_mask_loss_selected,
                     compute_mrcnn_mask_loss_unshuffled,
                     compute_rpn_bbox_loss, compute_rpn_class_loss, compute_rpn_losses_on_rows, detection_layer, detector_losses,
                     generate_pyramid_priors, prepare_det_target, prepare_rpn_target, proposal_layer, select_rpn_rows)
from .OT_module import OptTrans
from .sub_module import FPN, RPN, Classifier, Dev, Mask, ResNet

EPS = 1e-20


class MaskRCNN(nn.Module):
    def __init__(self, config):
        super(MaskRCNN, self).__init__()
        self.config = config
        self._build(config)
        self._initialize_weights()
        self.feature_buffer = None
        self.external_proposals = None   # optional callable -> [b, E, 5] rows that compete with the RPN's candidates
                                         # before NMS (precomputed proposals; synthetic.SyntheticProposals)
        self.generator = None         # optional torch.Generator for target sub-sampling

    def _build(self, config):
        resnet = ResNet(config.MODEL.BACKBONE, stage5=True)
        C1, C2, C3, C4, C5 = resnet.stages()
        self.fpn = FPN(config, C1, C2, C3, C4, C5, out_channels=256)
        priors = generate_pyramid_priors(config.RPN.ANCHOR_SCALES, config.RPN.ANCHOR_RATIOS,
                                         config.MODEL.BACKBONE_SHAPES, config.MODEL.BACKBONE_STRIDES,
                                         config.RPN.ANCHOR_STRIDE)
        self.register_buffer("priors", torch.from_numpy(priors).float(), persistent=False)
        self.rpn = RPN(len(config.RPN.ANCHOR_RATIOS), config.RPN.ANCHOR_STRIDE, input_ch=256)
        self.dev_roi = Dev(config, depth=256)
        if config.DEV.SWITCH and config.DEV.LOSS_CHOICE == 'ot':
            self.ot_loss = OptTrans(config, ch_x=1024, epsilon=config.DEV.OT_EPSILON, L=config.DEV.OT_L)
        self.classifier = Classifier(depth=256, num_classes=config.DATASET.NUM_CLASSES,
                                     pool_size=config.MRCNN.POOL_SIZE, config=config)
        self.mask = Mask(depth=256, num_classes=config.DATASET.NUM_CLASSES)

    def _initialize_weights(self):
        """lib/model.py:84-103."""
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv1d)):   # conv.Conv2d / conv.Conv1d subclass these
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, (nn.ConvTranspose2d, nn.ConvTranspose1d)):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                m.weight.data.normal_(0, 0.01)
                m.bias.data.zero_()

    def initialize_buffer(self, device=None):
        device = device or next(self.parameters()).device
        self.feature_buffer = FeatureBuffer(self.config.DEV.BUFFER_SIZE, 1024, self.config.DATASET.NUM_CLASSES,
                                            device)

    # ------------------------------------------------------------------ forward (train)
    def forward(self, input, mode='train'):
        """Runs _forward with the convolution arithmetic this model is configured for
        (config.MODEL.CONV_PRECISION); the process-wide setting is restored afterwards (the backward pass
        of every layer uses the precision its forward ran with)."""
        prev = conv_precision()
        set_conv_precision(getattr(self.config.MODEL, "CONV_PRECISION", "fp32"))
        try:
            return self._forward(input, mode)
        finally:
            set_conv_precision(prev)

    def _forward(self, input, mode='train'):
        """input = [images [b,3,S,S], gt_class_ids [b,G], gt_boxes [b,G,4] pixels, gt_masks [b,G,56,56]].
        Returns (loss_merge [1,5], big_feat, big_cnt, small_feat, small_cnt, big_loss,
        small_output_all, small_gt_all, fpn_ot_loss) as lib/model.py:466-469."""
        if mode not in ('train', 'inference'):
            raise NotImplementedError("modes: 'train' (SURVEY 8a) and 'inference' (8f-3); 'visualize' is out of scope")
        cfg = self.config
        images = input[0]
        bs = images.size(0)
        if mode == 'inference':
            return self._inference(images, input[1])
        gt_class_ids, gt_boxes, gt_masks = input[1], input[2], input[3]
        self.eval()   # the reference always runs BN (and everything else) in eval mode
        join, self._side_join = getattr(self, "_side_join", None), None
        if join is not None:      # side-stream work of a previous pass that was not joined by workflow.train_step
            join()
        prepare_step(self)   # BN folds, weight layouts and the zeroed gradient arena: a handful of launches
        proposal_cnt = cfg.RPN.POST_NMS_ROIS_INFERENCE   # also Q1

        # RPN targets depend on the anchors and the ground truth only: generated on a second stream, their ~150
        # small kernels interleave with the backbone instead of running alone (1.5 ms of the step)
        with torch.no_grad():
            rpn_target_ready = _lib.run_on_side_stream(prepare_rpn_target, self.priors, gt_class_ids, gt_boxes, cfg,
                                                       self.generator)
        p2, p3, p4, p5, p6, fpn_ot_loss = self.fpn(images, mode=mode)
        rpn_maps = [p2, p3, p4, p5, p6]
        mrcnn_maps = [p2, p3, p4, p5]
        # P2..P5 have three readers -- the RPN, the Dev make-up layer and (raw) the Dev stage's big-box crop.  Backward
        # visits them in the reverse order; each leaves its gradient for the map with the next one, whose data-gradient
        # kernel adds it, and only the RPN's shared convolution reports to autograd (conv.GradBox; off with conv.GATES)
        chain = bool(cfg.DEV.SWITCH) and images.is_cuda and torch.is_grad_enabled() and _conv.GATES
        to_rpn = [_conv.GradBox() if chain else None for _ in mrcnn_maps]
        to_make_up = [_conv.GradBox() if chain else None for _ in mrcnn_maps]
        # (the 14 x 14 crops likewise: mask head -> the row gather in front of the Dev stage's feature extractor)
        # Only the positive RoIs' masks enter the mask loss, and prepare_det_target puts an image's positives in its first
        # int(R * ROI_POSITIVE_RATIO) slots: with conv.GATES the mask head runs as two batches -- those slots WITH a graph,
        # the other slots without one (their outputs are computed, as the reference does, and their gradient is zero)
        split_mask = images.is_cuda and torch.is_grad_enabled() and _conv.GATES and \
            not cfg.MRCNN.MASK_HEAD_ON_POSITIVE_SLOTS
        P_front = int(cfg.ROIS.TRAIN_ROIS_PER_IMAGE * cfg.ROIS.ROI_POSITIVE_RATIO)
        # with the split, the Dev stage returns the 14 x 14 crops with the positive slots first (Dev.forward(mask_front)):
        # both batches are views, and the graph batch's input gradient (the first rows) goes through the box as well
        front = P_front if (split_mask and chain and _MASK_FRONT and 0 < P_front < cfg.ROIS.TRAIN_ROIS_PER_IMAGE) else None
        mask_box = None if ((split_mask and not front) or cfg.MRCNN.MASK_HEAD_ON_POSITIVE_SLOTS or not chain) \
            else _conv.GradBox()
        # The RPN losses read RPN.TRAIN_ANCHORS_PER_IMAGE sampled anchors per image: with conv.GATES the dense RPN runs
        # without a graph (the proposal layer needs every anchor) and the losses' graph is RPN.forward_rows on those rows
        rows = images.is_cuda and torch.is_grad_enabled() and _conv.GATES and self.rpn.anchor_stride == 1
        if rows:
            with torch.no_grad():
                outs = [self.rpn(p) for p in rpn_maps]
                target_rpn_match, target_rpn_deltas = rpn_target_ready()
                r_img, r_anchor, r_valid = select_rpn_rows(target_rpn_match, cfg.RPN.TRAIN_ANCHORS_PER_IMAGE)
            for b_ in to_rpn:
                if b_ is not None:
                    b_.taker = True                     # _PatchRowsFn takes what the make-up layer leaves
            row_logits, row_bbox = self.rpn.forward_rows(rpn_maps, r_img, r_anchor, r_valid,
                                                         grad_boxes=list(to_rpn) + [None] * (len(rpn_maps) - len(to_rpn)))
        else:
            outs = [self.rpn(p, to_rpn[i] if i < len(to_rpn) else None) for i, p in enumerate(rpn_maps)]
        rpn_logits, rpn_probs, rpn_bbox = [torch.cat(list(o), dim=1) for o in zip(*outs)]

        with torch.no_grad():
            h, w = float(cfg.DATA.IMAGE_SHAPE[0]), float(cfg.DATA.IMAGE_SHAPE[1])
            scale = const_tensor([h, w, h, w], images.device)
            # The proposal layer (candidate selection, NMS: a chain of latency-bound kernels on a handful of CUs, ~0.7 ms)
            # and the detection targets (IoU matching, sampling, mask-target crops: ~130 small kernels) on the second
            # stream; the Dev stage's make-up convolutions below do not depend on the RoIs and run meanwhile
            def propose(probs, bbox):
                extra = self.external_proposals() if self.external_proposals is not None else None
                return proposal_layer([probs, bbox], proposal_cnt, cfg.RPN.NMS_THRESHOLD, self.priors, cfg, extra)
            early = None if _PROPOSAL_SIDE else propose(rpn_probs, rpn_bbox)      # A/B: on the main stream

            def proposals_targets_levels(probs, bbox, *a):
                proposals, num_prop = early if early is not None else propose(probs, bbox)
                t = prepare_det_target(proposals, num_prop, *a)
                level, counts_ready = self.dev_roi.level_info(t[0])      # starts the one host read of the RoI stage
                return t + (level,), counts_ready

            gtb = gt_boxes / scale
            if images.is_cuda:
                side = _lib.run_on_side_stream(proposals_targets_levels, rpn_probs, rpn_bbox, gt_class_ids, gtb,
                                               gt_masks, cfg, self.generator)
            else:
                side = (lambda r=proposals_targets_levels(rpn_probs, rpn_bbox, gt_class_ids, gtb, gt_masks, cfg,
                                                          self.generator): r)
        up_maps = self.dev_roi.make_up_maps(mrcnn_maps, give_to=to_rpn, take_from=to_make_up) \
            if (cfg.DEV.SWITCH and images.is_cuda) else None
        with torch.no_grad():
            (rois, target_class_ids, target_deltas, target_mask, roi_lvl), counts_ready = side()
            if not rows:
                target_rpn_match, target_rpn_deltas = rpn_target_ready()

        K = cfg.DATASET.NUM_CLASSES
        pooled_cls, pooled_mask, feat_out = self.dev_roi(mrcnn_maps, rois, target_class_ids, up_maps=up_maps,
                                                         level_info=(roi_lvl, counts_ready),
                                                         raw_grad_boxes=to_make_up if chain else None,
                                                         mask_grad_box=mask_box, mask_front=front)
        # the statistics the meta loss reads are complete here: workflow.compute_loss evaluates it on another stream from
        # this point on, next to the box and mask heads (its ~70 small kernels and the Sinkhorn launch are latency bound)
        self._stats_ready = None
        if images.is_cuda:
            self._stats_ready = torch.cuda.Event()
            self._stats_ready.record(torch.cuda.current_stream(images.device))
        scale_num = 3
        if cfg.DEV.SWITCH and not cfg.DEV.BASELINE:
            big_feat, big_cnt, small_feat, small_cnt, big_loss, small_output_all, small_gt_all = feat_out
        else:
            z = images.new_zeros
            big_feat, small_feat = z(1, scale_num, 1024, K), z(1, scale_num, 1024, K)
            big_cnt, small_cnt = z(1, scale_num, 1, K), z(1, scale_num, 1, K)
            big_loss, small_output_all, small_gt_all = z(1, scale_num, 1), z(1, 1024), z(1)

        mrcnn_class_logits, _, mrcnn_bbox = self.classifier(pooled_cls, small_output_all, small_gt_all)
        mask_ids, mask_tgt = target_class_ids, target_mask
        if cfg.MRCNN.MASK_HEAD_ON_POSITIVE_SLOTS or split_mask:
            # The reference runs the mask head on every RoI (lib/model.py:442) although only positive
            # RoIs enter the mask loss (lib/layers.py:905-934).  prepare_det_target puts the positives
            # of an image in its first slots, at most ROI_POSITIVE_RATIO * R of them, so the head's
            # output on the other slots is never read.  MASK_HEAD_ON_POSITIVE_SLOTS (not the reference's schedule) drops
            # them: same loss, same gradients, 1/3 of the work.  The default evaluates them too, in a second batch that
            # builds no graph: every RoI's masks are computed exactly as before (the layers act per RoI), and the
            # backward pass of the head -- the largest convolutions of the step -- covers the third of the RoIs whose
            # gradient is not identically zero.
            P = int(cfg.ROIS.TRAIN_ROIS_PER_IMAGE * cfg.ROIS.ROI_POSITIVE_RATIO)
            R = rois.size(1)
            per_image = pooled_mask.view(bs, R, *pooled_mask.shape[1:]) if not front else None
            if split_mask and P < R:
                with torch.no_grad():
                    rest = pooled_mask[bs * P:] if front else per_image[:, P:].reshape(bs * (R - P), *pooled_mask.shape[1:])
                    if _DEAD_SIDE:
                        # nothing reads these masks: the batch runs on the second stream next to the rest of the step
                        # and is joined before the optimiser touches the weights (workflow.train_step: join_side_work)
                        dead = lambda t: (self.mask(t, shuffled=False, activate=False), None)[1]
                        self._side_join = _lib.run_on_side_stream(dead, rest)
                        rest.record_stream(_lib.side_stream(rest.device))
                    else:
                        self.mask(rest, shuffled=False, activate=False)
            pooled_mask = pooled_mask[:bs * P] if front else per_image[:, :P].reshape(bs * P, *pooled_mask.shape[1:])
            mask_ids, mask_tgt = target_class_ids[:, :P], target_mask[:, :P]
        # logits of every RoI's TARGET class [bs*R', 2, 2, 14, 14] (all K classes are evaluated; see Mask.forward), or of
        # all classes [bs*R', 2, 2, K, 14, 14] on a CPU tensor / with conv.GATES off
        mask_u = self.mask(pooled_mask, shuffled=False, activate=False, input_grad_box=mask_box,
                           select_class=mask_ids.reshape(-1))
        mrcnn_class_logits = mrcnn_class_logits.view(bs, -1, mrcnn_class_logits.size(1))
        mrcnn_bbox = mrcnn_bbox.view(bs, -1, mrcnn_bbox.size(1), mrcnn_bbox.size(2))
        mask_u = mask_u.view(bs, -1, *mask_u.shape[1:])

        fused = detector_losses(row_logits, row_bbox, r_img, r_anchor, target_rpn_match, target_rpn_deltas,
                                mrcnn_class_logits, mrcnn_bbox, target_class_ids, target_deltas, mask_u, mask_ids,
                                mask_tgt) if rows else None
        if fused is not None:
            big_done = getattr(self.dev_roi, "big_done", None)
            if big_done is not None:
                cur = torch.cuda.current_stream(images.device)
                cur.wait_event(big_done)
                for t in (big_feat, big_cnt, big_loss):
                    t.record_stream(cur)
            return (fused.view(1, 5), big_feat, big_cnt, small_feat, small_cnt, big_loss, small_output_all, small_gt_all,
                    fpn_ot_loss)
        if rows:
            rpn_cls_loss, rpn_box_loss = compute_rpn_losses_on_rows(target_rpn_match, target_rpn_deltas, r_img, r_anchor,
                                                                    r_valid, row_logits, row_bbox)
        else:
            rpn_cls_loss = compute_rpn_class_loss(target_rpn_match, rpn_logits)
            rpn_box_loss = compute_rpn_bbox_loss(target_rpn_deltas, target_rpn_match, rpn_bbox)
        losses = torch.stack((
            rpn_cls_loss,
            rpn_box_loss,
            compute_mrcnn_class_loss(target_class_ids, mrcnn_class_logits),
            compute_mrcnn_bbox_loss(target_deltas, target_class_ids, mrcnn_bbox),
            compute_mrcnn_mask_loss_selected(mask_tgt, mask_ids, mask_u) if mask_u.dim() == 6 else
            compute_mrcnn_mask_loss_unshuffled(mask_tgt, mask_ids, mask_u, from_logits=True))).view(1, 5)
        big_done = getattr(self.dev_roi, "big_done", None)
        if big_done is not None:
            # the Dev stage's big branch ran on the third stream: joined here, behind the heads (it finished long ago in
            # device time), so that whatever reads the statistics on this stream is ordered behind it
            cur = torch.cuda.current_stream(images.device)
            cur.wait_event(big_done)
            for t in (big_feat, big_cnt, big_loss):
                t.record_stream(cur)
        return (losses, big_feat, big_cnt, small_feat, small_cnt, big_loss, small_output_all, small_gt_all,
                fpn_ot_loss)

    # ------------------------------------------------------------------ inference
    @torch.no_grad()
    def _inference(self, images, image_metas):
        """lib/model.py:265-345, mode == 'inference': [detections [bs, 100, 6] (pixels, class, score),
        mrcnn_mask [bs, 100, K, 28, 28]].  `image_metas` is the reference's [bs, 8+K+1] meta array
        (window in columns 4:8, tools/image_utils.py:31-40) or just the windows [bs, 4]."""
        cfg = self.config
        bs = images.size(0)
        self.eval()
        prepare_step(self)
        p2, p3, p4, p5, p6, _ = self.fpn(images, mode='inference')
        mrcnn_maps = [p2, p3, p4, p5]
        outs = [self.rpn(p) for p in (p2, p3, p4, p5, p6)]
        _, rpn_probs, rpn_bbox = [torch.cat(list(o), dim=1) for o in zip(*outs)]
        extra = self.external_proposals() if self.external_proposals is not None else None
        proposals, _ = proposal_layer([rpn_probs, rpn_bbox], cfg.RPN.POST_NMS_ROIS_INFERENCE,
                                      cfg.RPN.NMS_THRESHOLD, self.priors, cfg, extra)
        pooled_cls, _, feat_out = self.dev_roi(mrcnn_maps, proposals)
        small_output_all, small_gt_all = feat_out if feat_out else (None, None)
        _, mrcnn_class, mrcnn_bbox = self.classifier(pooled_cls, small_output_all, small_gt_all)
        meta = torch.as_tensor(image_metas, device=images.device, dtype=torch.float32)
        windows = meta if meta.size(1) == 4 else meta[:, 4:8]
        detections = detection_layer(proposals, mrcnn_class, mrcnn_bbox, windows, cfg)
        h, w = float(cfg.DATA.IMAGE_SHAPE[0]), float(cfg.DATA.IMAGE_SHAPE[1])
        scale = const_tensor([h, w, h, w], images.device)
        _, pooled_mask, _ = self.dev_roi(mrcnn_maps, detections[:, :, :4] / scale)
        mrcnn_mask = self.mask(pooled_mask)
        mrcnn_mask = mrcnn_mask.view(bs, -1, mrcnn_mask.size(1), mrcnn_mask.size(2), mrcnn_mask.size(3))
        return [detections, mrcnn_mask]

    # ------------------------------------------------------------------ meta loss
    def meta_loss(self, feat_input, reduce_fn=None):
        """lib/model.py:143-210; see intertwiner.meta_loss for the static-shape formulation and the
        quirks it mirrors or decides.  `reduce_fn` is the data-parallel statistics all-reduce."""
        if self.feature_buffer is None:
            self.initialize_buffer(feat_input[0].device)
        return meta_loss(self.config, self.feature_buffer, getattr(self, "ot_loss", None), feat_input, reduce_fn)
