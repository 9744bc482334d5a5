"""Kernel time of the RPN target generation (fi_rpn_targets: rpn_iou_kernel + rpn_sample_kernel) alone on the chip, at the
bench workload's size (4 x 261 888 anchors, 20 GT boxes per image).  HIP events around the call."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_intertwiner_amd import layers as L  # noqa: E402
from feature_intertwiner_amd.config import make_config  # noqa: E402
from feature_intertwiner_amd.synthetic import synthetic_batch  # noqa: E402

DEV = "cuda:0"
for bs in (4, 2):
    cfg = make_config("resnet101", 1024, bs, 512, dev_switch=True)
    anchors = torch.from_numpy(L.generate_pyramid_priors(cfg.RPN.ANCHOR_SCALES, cfg.RPN.ANCHOR_RATIOS, cfg.MODEL.BACKBONE_SHAPES,
                                                         cfg.MODEL.BACKBONE_STRIDES, 1).astype(np.float32)).to(DEV)
    batch = synthetic_batch(bs, 1024, device=DEV, seed=2000)
    g = torch.Generator(device=DEV).manual_seed(1)
    for _ in range(3):
        L.prepare_rpn_target(anchors, batch[1], batch[2], cfg, g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.prepare_rpn_target(anchors, batch[1], batch[2], cfg, g)
    e1.record()
    torch.cuda.synchronize()
    print("prepare_rpn_target, %d images x %d anchors: %.1f us per call (2 torch.rand + fi_rpn_targets)" % (
        bs, anchors.size(0), e0.elapsed_time(e1) * 50))
