"""Which flow of tests/test_gpu_data_parallel.py's `full` case disagrees between the persistent 1x1 kernel and the reg kernel:
A = one forward + backward of a fresh model (a data-parallel worker's first step: BatchNorm fold pairs unknown at prepare
time), B = two forwards + one backward (the single-process statement of the rule)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import test_gpu_data_parallel as T
from feature_intertwiner_amd import conv

size = sys.argv[1] if len(sys.argv) > 1 else "full"
variant = "ot_l2cost"


def run(flow, ring):
    conv.RING_1X1 = ring
    conv.invalidate_step_state()
    cfg, model, _ = T._make(0, variant, size)
    outs = []
    for g in range(2 if flow == "B" else 1):
        batch, hook, gen = T._shard(g, T.SIZES[size][1])
        model.external_proposals, model.generator = hook, gen
        outs.append(model(list(batch), 'train'))
    merged = [torch.cat([o[i] for o in outs], 0) for i in range(9)]
    detailed = merged[0].mean(0)
    meta = model.meta_loss([merged[1], merged[2], merged[3], merged[4], merged[6], merged[7]])
    meta = torch.where(meta < 0, torch.zeros_like(meta), meta) * cfg.DEV.LOSS_FAC
    total = detailed.sum() + meta
    total.backward()
    torch.cuda.synchronize()
    return float(total), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


for flow in ("A", "B"):
    l1, g1 = run(flow, True)
    l0, g0 = run(flow, False)
    l0b, g0b = run(flow, False)
    for tag, (la, ga) in (("ring vs reg", (l1, g1)), ("reg vs reg ", (l0b, g0b))):
        dev = {n: ((ga[n] - g0[n]).abs().max() / (g0[n].abs().max() + 1e-12)).item() for n in g0}
        worst = sorted(dev.items(), key=lambda kv: -kv[1])[:6]
        print("flow %s %s: loss %.6f vs %.6f; params over 1e-3: %d of %d; worst %s"
              % (flow, tag, la, l0, sum(v > 1e-3 for v in dev.values()), len(dev), [(n, "%.2e" % v) for n, v in worst]), flush=True)
