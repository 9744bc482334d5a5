"""300 train steps of the headline workload: step time and device memory every 50 steps (no growth expected: gradients live
in one arena, side-stream tensors are released when their readers' events complete)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from feature_intertwiner_amd.config import make_config
from feature_intertwiner_amd.model import MaskRCNN
from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
from feature_intertwiner_amd.workflow import set_optimizer, train_step
dev = torch.device("cuda:0")
torch.manual_seed(2000)
cfg = make_config("resnet101", 1024, 4, 512, dev_switch=True, loss_choice="ot", ot_L=50)
model = MaskRCNN(cfg).to(dev)
opt = set_optimizer(model, cfg.TRAIN)
batch = synthetic_batch(4, 1024, device=dev, seed=2000)
model.external_proposals = SyntheticProposals(batch[2], 1024, seed=7, cycle=16)
model.generator = torch.Generator(device=dev).manual_seed(11)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = time.perf_counter()
for i in range(n):
    terms = train_step(model, opt, list(batch))
    if (i + 1) % 50 == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print("step %4d  %.2f ms/step  allocated %.1f MB  reserved %.1f MB  max allocated %.1f MB  total loss %.4f" % (
            i + 1, (t1 - t0) / 50 * 1e3, torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20,
            torch.cuda.max_memory_allocated() / 2**20, float(terms["total"])), flush=True)
        t0 = time.perf_counter()
