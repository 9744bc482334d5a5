"""Which (N, C, HW) shapes fi_bn_act_backward sees in one headline step, and the device time of each."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from feature_intertwiner_amd import _lib
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
model.generator = torch.Generator(device=dev).manual_seed(11)
for _ in range(2):
    train_step(model, opt, list(batch))
L = _lib.load()
orig = L.fi_bn_act_backward
rec = []


def wrapped(dy, y, scale, gamma, beta, res, N, C, HW, relu, dz, g, dshift, dgamma, dbias, layout, flags, stream):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    rc = orig(dy, y, scale, gamma, beta, res, N, C, HW, relu, dz, g, dshift, dgamma, dbias, layout, flags, stream)
    b.record()
    rec.append(((N, C, HW, res is not None and res.value is not None, g is not None and g.value is not None, layout), a, b))
    return rc


L.fi_bn_act_backward = wrapped
train_step(model, opt, list(batch))
torch.cuda.synchronize()
L.fi_bn_act_backward = orig
agg = collections.defaultdict(list)
for key, a, b in rec:
    agg[key].append(a.elapsed_time(b) * 1e3)
tot = 0.0
print("%6s %6s %7s %5s %5s %3s %4s %9s %9s %8s" % ("N", "C", "HW", "res", "g", "cl", "n", "mean_us", "GB/s", "ms/step"))
for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    N, C, HW, r, g, cl = key
    nbytes = 4.0 * N * C * HW * (3 + int(r) + int(g))
    m = sum(v) / len(v)
    tot += sum(v)
    print("%6d %6d %7d %5s %5s %3d %4d %9.1f %9.0f %8.3f" % (N, C, HW, r, g, cl, len(v), m, nbytes / m / 1e3, sum(v) / 1e3))
print("total %.3f ms/step over %d launches" % (tot / 1e3, len(rec)))
