"""nms_mask_kernel / nms_scan kernel alone on the chip (HIP events around each launch), 6000 boxes @0.7."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from helpers import clustered_dets
from feature_intertwiner_amd import _lib
from feature_intertwiner_amd.nms.pth_nms import nms_sorted
rs = np.random.RandomState(99)
for bs in (1, 4):
    dets = torch.from_numpy(np.stack([clustered_dets(rs, 6000, 1024) for _ in range(bs)])).cuda()
    for mk in (0, 2000, 1000):
        for _ in range(3):
            nms_sorted(dets, 0.7, max_keep=mk)
        torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(20):
            keep, num = nms_sorted(dets, 0.7, max_keep=mk)
        torch.cuda.synchronize(); _lib.prof_enable(False)
        out = {"batch": bs, "max_keep": mk, "kept": num.tolist(), "scan": "narrow" if os.environ.get("FI_NMS_SCAN_NARROW") else "wide"}
        for k in ("nms_mask", "nms_scan"):
            n, ms = _lib.prof_get(k)
            out[k + "_us"] = round(ms / max(n, 1) * 1e3, 1)
        print(json.dumps(out), flush=True)
