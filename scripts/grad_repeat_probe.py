"""Run-to-run repeatability of the gradients the meta loss sends into the Dev stage (headline model, default backward
form, the same weights / inputs / draws every pass): d loss / d small_feat, d loss / d small_output_all (hooks) and the
feature extractor's parameter gradients, compared bit for bit with the first pass.
    python scripts/grad_repeat_probe.py [--passes 8]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from feature_intertwiner_amd.config import make_config
from feature_intertwiner_amd.model import MaskRCNN
from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
from feature_intertwiner_amd.workflow import compute_loss, set_optimizer, train_step

ap = argparse.ArgumentParser()
ap.add_argument("--passes", type=int, default=8)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--pairs", action="store_true", help="a real train step before every PAIR of passes; pairs compared")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(2000)
cfg = make_config("resnet101", 1024, 4, 512, dev_switch=True, loss_choice="ot", ot_L=50)
model = MaskRCNN(cfg).to(dev)
opt = set_optimizer(model, cfg.TRAIN)
batch = synthetic_batch(4, 1024, device=dev, seed=2000)
model.external_proposals = SyntheticProposals(batch[2], 1024, seed=7)
model.generator = torch.Generator(device=dev).manual_seed(11)
for k in range(a.steps):
    train_step(model, opt, list(batch))
torch.cuda.synchronize()
fb = model.feature_buffer
saved = (fb.buffer.clone(), fb.buffer_cnt.clone())
ext = model.external_proposals.gen
ext_state = ext.get_state()
seen = {}
orig_meta = model.meta_loss


def meta_with_hooks(feats, reduce_fn=None):
    for i, n in ((2, "d_small_feat"), (4, "d_small_output_all")):
        if feats[i].requires_grad:
            feats[i].register_hook(lambda g, n=n: seen.__setitem__(n, g.detach().clone()))
    seen["in_small_feat"] = feats[2].detach().clone()
    seen["in_big_feat"] = feats[0].detach().clone()
    out = orig_meta(feats, reduce_fn=reduce_fn)
    seen["meta"] = out.detach().clone()
    return out


model.meta_loss = meta_with_hooks
first = None
for p in range(a.passes):
    if a.pairs and p % 2 == 0:
        model.meta_loss = orig_meta
        train_step(model, opt, list(batch))
        model.meta_loss = meta_with_hooks
        torch.cuda.synchronize()
        saved = (fb.buffer.clone(), fb.buffer_cnt.clone())
        ext_state = ext.get_state()
        first = None
    fb.buffer.copy_(saved[0]); fb.buffer_cnt.copy_(saved[1])
    ext.set_state(ext_state)
    model.generator = torch.Generator(device=dev).manual_seed(3)
    for q in model.parameters():
        q.grad = None
    seen.clear()
    loss, _ = compute_loss(model, list(batch), True, 1, None)
    loss.backward()
    join = getattr(model, "_side_join", None)
    if join is not None:
        model._side_join = None
        join()
    torch.cuda.synchronize()
    cur = dict(seen)
    for n, q in model.named_parameters():
        if n.startswith("dev_roi.feat_extract") or n.startswith("ot_loss"):
            cur["grad " + n] = None if q.grad is None else q.grad.detach().clone()
    cur["grad fpn.C4.3.conv2.weight"] = dict(model.named_parameters())["fpn.C4.3.conv2.weight"].grad.detach().clone()
    if first is None:
        first = cur
        print("pass %d:" % p, ", ".join("%s %.3g" % (n, float(t.abs().max())) for n, t in cur.items() if t is not None and (not a.pairs or "ot_loss" in n or "0.bias" in n or n in ("meta", "d_small_feat"))), flush=True)
        continue
    diffs = []
    for n in cur:
        if cur[n] is None or first[n] is None:
            continue
        if not torch.equal(cur[n], first[n]):
            d = (cur[n].double() - first[n].double()).abs()
            diffs.append("%s: max|.| %.3g vs %.3g, max diff %.3g" % (n, float(cur[n].abs().max()), float(first[n].abs().max()),
                                                                  float(d.max())))
    print("pass", p, "identical" if not diffs else "\n      ".join([""] + diffs), flush=True)
