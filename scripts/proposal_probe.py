"""The proposal selection alone on the chip (HIP events around its launches): 4 images x 261 888 anchors, 6000 candidates.
FI_PROPOSAL_LDS_SORT=1: the all-LDS bitonic sort; FI_PROPOSAL_MULTI_WG=0: the single kernel."""
import json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from feature_intertwiner_amd import _lib, layers as L
from feature_intertwiner_amd.config import make_config
dev = "cuda:0"
cfg = make_config("resnet101", 1024, 4, 512, dev_switch=True, loss_choice="ot", ot_L=50)
pri = torch.from_numpy(L.generate_pyramid_priors(cfg.RPN.ANCHOR_SCALES, cfg.RPN.ANCHOR_RATIOS, cfg.MODEL.BACKBONE_SHAPES,
                                                 cfg.MODEL.BACKBONE_STRIDES, 1)).float().to(dev)
g = torch.Generator(device=dev).manual_seed(3)
A = pri.size(0)
probs = torch.rand(4, A, 2, device=dev, generator=g)
bbox = torch.randn(4, A, 4, device=dev, generator=g) * 0.3
with torch.no_grad():
    for _ in range(3):
        L.proposal_layer([probs, bbox], 2000, 0.7, pri, cfg)
    torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(20):
        L.proposal_layer([probs, bbox], 2000, 0.7, pri, cfg)
    torch.cuda.synchronize(); _lib.prof_enable(False)
out = {"anchors": A, "multi_wg": L.PROPOSAL_MULTI_WG, "lds_sort": bool(os.environ.get("FI_PROPOSAL_LDS_SORT"))}
for k in ("proposal_select", "nms_mask", "nms_scan", "proposal_gather"):
    n, ms = _lib.prof_get(k)
    out[k + "_us"] = round(ms / max(n, 1) * 1e3, 1)
print(json.dumps(out))
