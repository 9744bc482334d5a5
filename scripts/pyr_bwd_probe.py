import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import training_rois
from feature_intertwiner_amd.roi_align.crop_and_resize import pyramid_crop_and_resize
from feature_intertwiner_amd.intertwiner import roi_level
DEV = 'cuda:0'
rs = np.random.RandomState(1)
B, C = 4, 256
maps = [torch.randn(B, C, s, s, device=DEV, requires_grad=True) for s in (256, 128, 64, 32)]
rois = torch.from_numpy(training_rois(rs, B, 512).reshape(-1, 4)).to(DEV)
ind = torch.arange(B, device=DEV, dtype=torch.int32).repeat_interleave(512)
level = roi_level(rois, 1024.0 * 1024.0)
for crop in (7, 14):
    out = pyramid_crop_and_resize(maps, rois, ind, level, crop, crop)
    g = torch.randn_like(out)
    for _ in range(3):
        out.backward(g, retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        for m in maps: m.grad = None
        out.backward(g, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    print("NCHW pyramid backward 2048 x 256, crop", crop, "scatter" if os.environ.get("FI_CROP_BWD_SCATTER") else "tiles", "%.1f us per call (incl. autograd + alloc)" % (e0.elapsed_time(e1) * 100))
