"""Where the channels-last pyramid crop backward spends its time, by level: the bench workload's 2048 RoIs with all but one
level's boxes disabled (level 0 = skipped by the kernel).  In-library HIP events around the backward kernel."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import training_rois  # noqa: E402
from feature_intertwiner_amd import _lib  # noqa: E402
from feature_intertwiner_amd.intertwiner import roi_level  # noqa: E402
from feature_intertwiner_amd.roi_align.crop_and_resize import pyramid_crop_and_resize  # noqa: E402

DEV = "cuda:0"
rs = np.random.RandomState(1)
B, C = 4, 256
maps = [torch.randn(B, C, s, s, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        for s in (256, 128, 64, 32)]
rois = torch.from_numpy(training_rois(rs, B, 512).reshape(-1, 4)).to(DEV)
ind = torch.arange(B, device=DEV, dtype=torch.int32).repeat_interleave(512)
level = roi_level(rois, 1024.0 * 1024.0)
print("boxes per level", [int((level == l).sum()) for l in (2, 3, 4, 5)])
for crop in (7, 14):
    for only in (None, 2, 3, 4, 5):
        lv = level if only is None else torch.where(level == only, level, torch.zeros_like(level))
        out = pyramid_crop_and_resize(maps, rois, ind, lv, crop, crop)
        g = torch.randn_like(out)
        for _ in range(3):
            out.backward(g, retain_graph=True)
        torch.cuda.synchronize()
        _lib.prof_reset()
        _lib.prof_enable(True)
        for _ in range(10):
            for m in maps:
                m.grad = None
            out.backward(g, retain_graph=True)
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        key = "crop_bwd_nhwc_7x7" if crop == 7 else "crop_bwd_nhwc_14x14"
        n, ms = _lib.prof_get(key)
        print("crop %2d  level %s: %.1f us per launch (%d launches)" % (crop, only or "all", ms / max(n, 1) * 1e3, n))
