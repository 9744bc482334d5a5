"""Host-side time of the step's phases (no synchronisation inside the step): where the host cannot run ahead of the
device -- the step boundary: optimizer table, zero_grad, prepare_step -- its time is a device bubble."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from feature_intertwiner_amd import conv, optim, workflow
from feature_intertwiner_amd.config import make_config
from feature_intertwiner_amd.model import MaskRCNN
from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch

dev = "cuda:0"
torch.manual_seed(2000)
cfg = make_config("resnet101", 1024, 4, 512, dev_switch=True, loss_choice="ot", ot_L=50)
model = MaskRCNN(cfg).to(dev)
opt = workflow.set_optimizer(model, cfg.TRAIN)
batch = synthetic_batch(4, 1024, device=dev, seed=2000)
model.external_proposals = SyntheticProposals(batch[2], 1024, seed=7)
model.generator = torch.Generator(device=dev).manual_seed(11)
for _ in range(3):
    workflow.train_step(model, opt, list(batch))
torch.cuda.synchronize()

T = {}
def timed(name, fn):
    def inner(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        T[name] = T.get(name, 0.0) + (time.perf_counter() - t)
        return r
    return inner

conv.refresh_bn_folds = timed("refresh_bn_folds", conv.refresh_bn_folds)
conv._prepare_step = timed("prepare_step(total)", conv._prepare_step)
optim.clip_and_step = timed("clip_and_step", optim.clip_and_step)
workflow.compute_loss = timed("compute_loss(forward, host)", workflow.compute_loss)
orig_zero = opt.zero_grad
opt.zero_grad = timed("zero_grad", orig_zero)
N = 5
t0 = time.perf_counter()
for _ in range(N):
    t = time.perf_counter()
    workflow.train_step(model, opt, list(batch))
    T["train_step(host)"] = T.get("train_step(host)", 0.0) + time.perf_counter() - t
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N
print("wall per step %.2f ms" % (wall * 1e3))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("%-32s %8.3f ms / step" % (k, v / N * 1e3))
