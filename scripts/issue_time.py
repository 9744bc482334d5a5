"""How long the host needs to ISSUE one train step (no synchronisation) vs the GPU time of the step:
the headroom that keeps 8 processes on one node from becoming launch-bound."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_intertwiner_amd.config import make_config
from feature_intertwiner_amd.model import MaskRCNN
from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
from feature_intertwiner_amd.workflow import set_optimizer, train_step

dev = "cuda:0"
torch.manual_seed(2000)
cfg = make_config("resnet101", 1024, 4, 512, dev_switch=True, loss_choice="ot", ot_L=50)
model = MaskRCNN(cfg).to(dev)
opt = set_optimizer(model, cfg.TRAIN)
batch = synthetic_batch(4, 1024, device=dev, seed=2000)
model.external_proposals = SyntheticProposals(batch[2], 1024, seed=7)
model.generator = torch.Generator(device=dev).manual_seed(2000)
for _ in range(3):
    train_step(model, opt, list(batch))
torch.cuda.synchronize()
issue, total = [], []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_step(model, opt, list(batch))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    issue.append((t1 - t0) * 1e3)
    total.append((t2 - t0) * 1e3)
print("host issue ms per step:", [round(v, 1) for v in issue])
print("step ms (issue + drain):", [round(v, 1) for v in total])
