"""RoIPool forward: time vs RoI size (uniform populations) to separate per-element from per-RoI cost."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from feature_intertwiner_amd import _lib
from feature_intertwiner_amd.roi_pooling.functions.roi_pool import RoIPoolFunction
DEV = "cuda:0"
image = torch.randn(2, 256, 256, 256, device=DEV)
rs = np.random.RandomState(0)
for pool in (7, 14):
    for side in (4, 8, 16, 32, 64, 128, 250):
        n = 512
        x1 = rs.uniform(0, 256 - side, n); y1 = rs.uniform(0, 256 - side, n)
        rois = np.stack([rs.randint(0, 2, n), x1, y1, x1 + side - 1, y1 + side - 1], 1).astype(np.float32)
        r = torch.from_numpy(rois).to(DEV)
        fn = RoIPoolFunction(pool, pool, 1.0)
        for _ in range(5): fn(image, r)
        torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(20): fn(image, r)
        torch.cuda.synchronize(); _lib.prof_enable(False)
        k, ms = _lib.prof_get("roipool_fwd")
        us = ms / k * 1e3
        elems = n * 256 * side * side
        print(json.dumps({"pool": pool, "side": side, "us": round(us, 1), "ns_per_roi_channel": round(us * 1e3 / (n * 256), 2),
                          "GB/s_read": round(elems * 4 / us / 1e3, 1)}))
print("-- identical RoIs (one 250x250 window on image 0): everything after the first touch is an L2 hit")
for pool in (1, 7, 14):
    rois = np.tile(np.array([[0, 3, 3, 252, 252]], np.float32), (512, 1))
    r = torch.from_numpy(rois).to(DEV)
    fn = RoIPoolFunction(pool, pool, 1.0)
    for _ in range(3): fn(image, r)
    torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(10): fn(image, r)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    k, ms = _lib.prof_get("roipool_fwd")
    us = ms / k * 1e3
    print(json.dumps({"pool": pool, "us": round(us, 1), "GB/s_read": round(512 * 256 * 250 * 250 * 4 / us / 1e3, 1)}))
