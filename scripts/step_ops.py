"""Which ATen ops (and from where) make up the non-library part of a train step: torch.profiler over ONE
step of the bench workload, grouped by op and by Python call site."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train_step(model, opt, list(batch))
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
print("%-46s %7s %12s %12s" % ("op", "count", "cpu_ms", "device_ms"))
for e in rows[:45]:
    print("%-46s %7d %12.2f %12.2f" % (e.key[:46], e.count, e.cpu_time_total / 1e3, getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0)) / 1e3))
# call sites of the frequent small ops: nearest python frame of this package (forward) or the autograd
# node that issued them (backward)
want = ("aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::fill_", "aten::zero_", "aten::clone",
        "aten::contiguous", "hipMemcpyAsync", "hipMemcpyWithStream", "aten::_to_copy", "aten::sub", "aten::div")
sites = collections.Counter()
site_dev = collections.Counter()
for ev in prof.events():
    if ev.name not in want:
        continue
    site = None
    for f in (ev.stack or []):
        if "feature_intertwiner" in f or "bench.py" in f or "workflow" in f:
            site = f.split("/")[-1][:90]
            break
    if site is None:
        p_ = ev.cpu_parent
        while p_ is not None:
            if "evaluate_function" in p_.name or "Backward" in p_.name or "Optimizer" in p_.name:
                site = p_.name[:90]
                break
            p_ = p_.cpu_parent
    sites[(ev.name, site or "?")] += 1
    site_dev[(ev.name, site or "?")] += sum(k.duration for k in (ev.kernels or []))
print("%5s %9s  %-16s %s" % ("count", "device_us", "op", "call site"))
for (name, site), us in site_dev.most_common(70):
    print("%5d %9.1f  %-16s %s" % (sites[(name, site)], us, name, site))

# GPU busy vs wall: union of device-side kernel intervals inside the profiled step
iv = []
named = []
for ev in prof.events():
    dt = getattr(ev, "device_type", None)
    if str(dt).endswith("CUDA") and ev.time_range is not None:
        iv.append((ev.time_range.start, ev.time_range.end))
        named.append((ev.time_range.start, ev.time_range.end, ev.name))
iv.sort()
named.sort()
busy, cur_s, cur_e = 0.0, None, None
gaps = []
for s_, e_ in iv:
    if cur_e is None or s_ > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append((s_ - cur_e, cur_e))
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
if cur_e is not None:
    busy += cur_e - cur_s
span = iv[-1][1] - iv[0][0] if iv else 0
print("device kernels: %d, span %.2f ms, busy %.2f ms, idle %.2f ms" % (len(iv), span / 1e3, busy / 1e3, (span - busy) / 1e3))
gaps.sort(reverse=True)
for gap, at in gaps[:6]:
    before = [n for s0, e0, n in named if abs(e0 - at) < 1e-3 or (e0 <= at and at - e0 < 5)][-1:]
    after = [n for s0, e0, n in named if s0 >= at][:1]
    print("gap %8.1f us  after %-60s before %-60s" % (gap, (before or ["?"])[0][:60], (after or ["?"])[0][:60]))
print("largest idle gaps (us):", [round(g[0], 1) for g in gaps[:25]])
print("gaps > 20us: %d totalling %.2f ms; gaps <= 20us: %d totalling %.2f ms" % (
    sum(1 for g in gaps if g[0] > 20), sum(g[0] for g in gaps if g[0] > 20) / 1e3,
    sum(1 for g in gaps if g[0] <= 20), sum(g[0] for g in gaps if g[0] <= 20) / 1e3))
