"""Run-to-run repeatability of the intertwiner statistics of the headline model: the same weights, inputs and random
draws, the forward pass repeated; every statistic compared bit for bit with the first pass.
    python scripts/stats_repeat_probe.py [--passes 8]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from feature_intertwiner_amd.config import make_config
from feature_intertwiner_amd.model import MaskRCNN
from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
from feature_intertwiner_amd.workflow import set_optimizer, train_step

ap = argparse.ArgumentParser()
ap.add_argument("--passes", type=int, default=8)
ap.add_argument("--steps", type=int, default=4)
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
names = ["merged_loss", "big_feat", "big_cnt", "small_feat", "small_cnt", "big_loss", "small_output_all", "small_gt_all",
         "fpn_ot_loss"]
first = None
for p in range(a.passes):
    fb.buffer.copy_(saved[0]); fb.buffer_cnt.copy_(saved[1])
    ext.set_state(ext_state)
    model.generator = torch.Generator(device=dev).manual_seed(3)
    with torch.no_grad():
        out = model(list(batch), 'train')
        big_done = getattr(model.dev_roi, "big_done", None)
        if big_done is not None:
            torch.cuda.current_stream().wait_event(big_done)
        model._stats_ready = None
        meta = model.meta_loss([out[1], out[2], out[3], out[4], out[6], out[7]], reduce_fn=None)
    join = getattr(model, "_side_join", None)
    if join is not None:
        model._side_join = None
        join()
    torch.cuda.synchronize()
    cur = {n: t.detach().clone() for n, t in zip(names, out)}
    cur["meta"] = meta.detach().clone()
    if first is None:
        first = cur
        print("pass 0: meta %.9g  big rows %d" % (float(meta), int(cur["big_cnt"].sum())), flush=True)
        continue
    diffs = []
    for n in cur:
        if not torch.equal(cur[n], first[n]):
            d = (cur[n].double() - first[n].double()).abs()
            diffs.append("%s: %d of %d differ, max %.3g" % (n, int((d > 0).sum()), d.numel(), float(d.max())))
    print("pass", p, "identical" if not diffs else "; ".join(diffs), flush=True)
