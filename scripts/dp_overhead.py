"""The data-parallel engine's own cost: the bench step with and without GradientBuckets (hooks, in-place bucket
all-reduce over RCCL with ONE rank, statistics all-reduce), alternating inside one process."""
import os, sys, time, torch
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29533"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from feature_intertwiner_amd import workflow
from feature_intertwiner_amd.config import make_config
from feature_intertwiner_amd.model import MaskRCNN
from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
from feature_intertwiner_amd.data_parallel import GradientBuckets, all_reduce_statistics
dev = "cuda:0"
dist.init_process_group("nccl", rank=0, world_size=1)
torch.manual_seed(2000)
cfg = make_config("resnet101", 1024, 4, 512, dev_switch=True, loss_choice="ot", ot_L=50)
model = MaskRCNN(cfg).to(dev)
opt = workflow.set_optimizer(model, cfg.TRAIN)
batch = synthetic_batch(4, 1024, device=dev, seed=2000)
model.external_proposals = SyntheticProposals(batch[2], 1024, seed=7)
model.generator = torch.Generator(device=dev).manual_seed(11)
sync = GradientBuckets(model)
def step(s):
    return workflow.train_step(model, opt, list(batch), True, s, 1, all_reduce_statistics if s else None)
for use in (None, sync, None, sync):
    for _ in range(3):
        step(use)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        step(use)
    torch.cuda.synchronize()
    print("sync" if use else "plain", (time.perf_counter() - t0) / 10 * 1e3, flush=True)
dist.destroy_process_group()
