import os, sys, collections
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from feature_intertwiner_amd.config import make_config
from feature_intertwiner_amd.model import MaskRCNN
from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
from feature_intertwiner_amd.workflow import set_optimizer, train_step
from torch.profiler import profile, ProfilerActivity
dev = "cuda:0"
torch.manual_seed(2000)
cfg = make_config("resnet101", 1024, 4, 512, dev_switch=True, loss_choice="ot", ot_L=50)
model = MaskRCNN(cfg).to(dev)
opt = set_optimizer(model, cfg.TRAIN)
batch = synthetic_batch(4, 1024, device=dev, seed=2000)
model.external_proposals = SyntheticProposals(batch[2], 1024, seed=7)
model.generator = torch.Generator(device=dev).manual_seed(11)
for _ in range(3):
    train_step(model, opt, list(batch))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    train_step(model, opt, list(batch))
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name not in ("aten::add_", "aten::add", "aten::copy_", "aten::fill_", "aten::zero_", "aten::mul", "aten::clone", "aten::contiguous"):
        continue
    par = ev.cpu_parent
    pn = par.name[:60] if par is not None else "-"
    us = sum(k.duration for k in (ev.kernels or []))
    if us < 8:
        continue
    key = (ev.name, pn, str(ev.input_shapes)[:80])
    agg[key][0] += 1
    agg[key][1] += us
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%4d %8.1f us  %-12s %-50s %s" % (v[0], v[1], k[0], k[1], k[2]))
