"""What is on the MAIN stream of the train step besides the MFMA kernels: per stream busy time, then for the main stream the
non-convolution kernels (by name, time per step) and the gaps between consecutive kernels (launch gaps and waits for side
streams), with the kernels on either side of the largest gaps.  torch.profiler, 3 traced steps of the bench workload.
    python scripts/main_stream_report.py [--batch 2]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from feature_intertwiner_amd.config import make_config  # noqa: E402
from feature_intertwiner_amd.model import MaskRCNN  # noqa: E402
from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch  # noqa: E402
from feature_intertwiner_amd.workflow import set_optimizer, train_step  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 4
torch.manual_seed(2000)
cfg = make_config("resnet101", 1024, B, 512, dev_switch=True, loss_choice="ot", ot_L=50)
model = MaskRCNN(cfg).to(dev)
opt = set_optimizer(model, cfg.TRAIN)
batch = synthetic_batch(B, 1024, device=dev, seed=2000)
model.external_proposals = SyntheticProposals(batch[2], 1024, seed=7, cycle=16)
model.generator = torch.Generator(device=dev).manual_seed(11)
for _ in range(4):
    train_step(model, opt, list(batch))
torch.cuda.synchronize()
STEPS = 3
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(STEPS):
        train_step(model, opt, list(batch))
    torch.cuda.synchronize()
import json, tempfile
path = os.path.join(tempfile.mkdtemp(), "t.json")
prof.export_chrome_trace(path)
tr = json.load(open(path))
ev = [e for e in tr["traceEvents"] if e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset") and "dur" in e]
by_stream = collections.defaultdict(list)
for e in ev:
    by_stream[e["args"].get("stream", e.get("tid"))].append((e["ts"], e["ts"] + e["dur"], e["name"]))
print("streams (busy ms/step, kernels/step):")
for sid, lst in sorted(by_stream.items(), key=lambda kv: -sum(b - a for a, b, _ in kv[1])):
    print("  stream %-6s %8.2f ms  %5d" % (sid, sum(b - a for a, b, _ in lst) / STEPS / 1e3, len(lst) // STEPS))
main = max(by_stream.items(), key=lambda kv: sum(b - a for a, b, _ in kv[1]))[1]
main.sort()
span = (main[-1][1] - main[0][0]) / STEPS / 1e3
busy = sum(b - a for a, b, _ in main) / STEPS / 1e3
print("main stream: span %.2f ms/step, busy %.2f, gaps %.2f" % (span, busy, span - busy))


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    return n[:70]


conv = collections.Counter()
other = collections.Counter()
cnt = collections.Counter()
for a, b, n in main:
    k = short(n)
    (conv if ("conv" in k or "gemm" in k) and "at::" not in k else other)[k] += (b - a)
    cnt[k] += 1
print("main stream, convolution / GEMM kernels: %.2f ms/step" % (sum(conv.values()) / STEPS / 1e3))
print("main stream, everything else: %.2f ms/step" % (sum(other.values()) / STEPS / 1e3))
for k, v in other.most_common(40):
    print("   %8.1f us/step  x%-4d %s" % (v / STEPS, cnt[k] // STEPS, k))
gaps = []
for (a0, b0, n0), (a1, b1, n1) in zip(main, main[1:]):
    if a1 > b0:
        gaps.append((a1 - b0, short(n0), short(n1)))
gaps.sort(reverse=True)
tot = sum(g for g, _, _ in gaps) / STEPS / 1e3
print("gaps on the main stream: %.2f ms/step in %d gaps/step" % (tot, len(gaps) // STEPS))
for lim in (5, 10, 20, 50, 200, 1e9):
    sel = [g for g, _, _ in gaps if g <= lim]
    print("   <= %6.0f us: %5d per step, %.3f ms/step" % (lim, len(sel) // STEPS, sum(sel) / STEPS / 1e3))
print("largest gaps (us): after -> before")
for g, n0, n1 in gaps[:30]:
    print("   %8.1f  %s  ->  %s" % (g, n0[:48], n1[:48]))
