"""Soak: N train steps of the bench workload; step time, allocated / reserved memory and loss trend
(checks for leaks, allocator growth and numerical blow-ups over a longer run than bench.py)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_intertwiner_amd.config import make_config
from feature_intertwiner_amd.model import MaskRCNN
from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
from feature_intertwiner_amd.workflow import set_optimizer, train_step
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = "cuda:0"; torch.manual_seed(2000)
cfg = make_config("resnet101", 1024, 4, 512, dev_switch=True, loss_choice="ot", ot_L=50,
                  conv_precision=os.environ.get("FI_SOAK_PRECISION", "fp32"))
m = MaskRCNN(cfg).to(dev); opt = set_optimizer(m, cfg.TRAIN)
b = synthetic_batch(4, 1024, device=dev, seed=2000); m.external_proposals = SyntheticProposals(b[2], 1024, seed=7)
m.generator = torch.Generator(device=dev).manual_seed(1)
t0 = time.perf_counter()
for i in range(steps):
    t = train_step(m, opt, list(b))
    if i % 10 == 9 or i == 0:
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("step %3d  total %.4f meta %.4f  alloc %.2f GB  reserved %.2f GB  max %.2f GB  %.1f ms/step" % (
            i + 1, float(t["total"]), float(t["meta"]), torch.cuda.memory_allocated() / 2**30,
            torch.cuda.memory_reserved() / 2**30, torch.cuda.max_memory_allocated() / 2**30,
            dt / (10 if i else 1) * 1e3))
        assert all(torch.isfinite(v) for v in t.values())
        t0 = time.perf_counter()
