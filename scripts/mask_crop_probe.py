"""The mask-target crop (depth 1, 28 x 28, lib/layers.py:301-322) alone on the chip: crop_fwd_c1_kernel vs the channel-chunk
kernel (FI_CROP_NO_C1=1)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_intertwiner_amd import _lib
from feature_intertwiner_amd.roi_align.crop_and_resize import CropAndResizeFunction
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
masks = (torch.rand(400, 1, 56, 56, device=dev, generator=g) > 0.5).float()
y1 = torch.rand(2048, 1, device=dev, generator=g) * 0.5
x1 = torch.rand(2048, 1, device=dev, generator=g) * 0.5
boxes = torch.cat([y1, x1, y1 + 0.4, x1 + 0.4], 1)
boxes[683:] = 0.0                      # the non-positive slots carry zero boxes
ind = torch.randint(0, 400, (2048,), device=dev, generator=g, dtype=torch.int32)
fn = CropAndResizeFunction(28, 28)
with torch.no_grad():
    for _ in range(5):
        fn(masks, boxes, ind)
    torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(50):
        fn(masks, boxes, ind)
    torch.cuda.synchronize(); _lib.prof_enable(False)
n, ms = _lib.prof_get("crop_fwd_28x28")
print(json.dumps({"kernel": "c1" if not os.environ.get("FI_CROP_NO_C1") else "chunk", "launches": n, "us": round(ms / max(n, 1) * 1e3, 1)}))
