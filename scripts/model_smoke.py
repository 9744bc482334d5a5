"""Run a few training steps of the synthetic workload and print losses/timings (dev aid)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_intertwiner_amd.config import make_config
from feature_intertwiner_amd.model import MaskRCNN
from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
from feature_intertwiner_amd.workflow import set_optimizer, train_step

ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="resnet50")
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--rois", type=int, default=64)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--L", type=int, default=5)
ap.add_argument("--dev", type=int, default=1)
ap.add_argument("--find", type=int, default=0)
a = ap.parse_args()
torch.manual_seed(2000)
torch.backends.cudnn.benchmark = bool(a.find)
cfg = make_config(a.backbone, a.size, a.batch, a.rois, dev_switch=bool(a.dev), ot_L=a.L)
model = MaskRCNN(cfg).cuda()
print("params", sum(p.numel() for p in model.parameters()) / 1e6, "M")
opt = set_optimizer(model, cfg.TRAIN)
batch = synthetic_batch(a.batch, a.size)
model.external_proposals = SyntheticProposals(batch[2], a.size)
for i in range(a.steps):
    torch.cuda.synchronize()
    t = time.time()
    terms = train_step(model, opt, list(batch))
    torch.cuda.synchronize()
    print(i, "%.1f ms" % ((time.time() - t) * 1e3), {k: round(float(v), 5) for k, v in terms.items()},
          "mem %.1f GB" % (torch.cuda.max_memory_allocated() / 1e9))
