"""The RPN ReLU-boundary events of workflow.compare_backward_forms at BASELINE configs[2], with their evidence
(workflow._relu_boundary_evidence): for several random draws, which channels of rpn.conv_shared differ between the default
and the dense backward, and what the two evaluations of the shared convolution computed at the sampled anchors there.
    python scripts/relu_boundary_probe.py [n_draws]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from feature_intertwiner_amd.config import make_config  # noqa: E402
from feature_intertwiner_amd.model import MaskRCNN  # noqa: E402
from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch  # noqa: E402
from feature_intertwiner_amd.workflow import compare_backward_forms, set_optimizer, train_step  # noqa: E402

DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
torch.manual_seed(2000)
cfg = make_config("resnet101", 1024, 4, 512, dev_switch=True, loss_choice="ot", ot_L=50)
model = MaskRCNN(cfg).to(DEV)
opt = set_optimizer(model, cfg.TRAIN)
batch = synthetic_batch(4, 1024, device=DEV, seed=2000)
model.external_proposals = SyntheticProposals(batch[2], 1024, seed=7)
model.generator = torch.Generator(device=DEV).manual_seed(11)
for _ in range(2):
    train_step(model, opt, list(batch))
for k in range(n):
    r = compare_backward_forms(model, batch, generator_seed=3 + k)
    print(json.dumps({"seed": 3 + k, "max_rel_dev": r["max_rel_dev"], "worst": r["worst"],
                      "channels": r.get("rpn_relu_boundary_which"), "evidence": r.get("rpn_relu_boundary_evidence")}))
    if "--mask" in sys.argv and r.get("rpn_relu_boundary_channels"):
        r2 = compare_backward_forms(model, batch, generator_seed=3 + k, rpn_mask_from_dense=True)
        print(json.dumps({"seed": 3 + k, "with_dense_mask": True, "max_rel_dev": r2["max_rel_dev"], "worst": r2["worst"],
                          "channels": r2.get("rpn_relu_boundary_which")}))
