"""Experiment: the whole train step as ONE hipGraph (torch.cuda.CUDAGraph) -- possible since the step has no host
synchronisation (Dev.static_shapes).  Prints ms/step eager vs replayed and checks that the replayed steps train (losses
move, parameters change)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_intertwiner_amd import workflow
from feature_intertwiner_amd.config import make_config
from feature_intertwiner_amd.model import MaskRCNN
from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
dev = torch.device("cuda", 0)
small = "--small" in sys.argv
BATCH = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 4      # 2 = BASELINE configs[3]'s per-GPU shape
torch.manual_seed(2000)
if "--cfg5" in sys.argv:          # the single-GPU slice of BASELINE configs[4] on the bf16 kernels
    cfg = make_config("resnet101", 1344, 2, 1000, dev_switch=True, loss_choice="ot", ot_L=50, conv_precision="bf16")
else:
    cfg = make_config("resnet50" if small else "resnet101", 512 if small else 1024, 2 if small else BATCH, 128 if small else 512,
                      dev_switch=True, loss_choice="ot", ot_L=50)
model = MaskRCNN(cfg).to(dev)
opt = workflow.set_optimizer(model, cfg.TRAIN)
batch = synthetic_batch(cfg.TRAIN.BATCH_SIZE, cfg.DATA.IMAGE_MAX_DIM, device=dev, seed=2000)
model.external_proposals = SyntheticProposals(batch[2], cfg.DATA.IMAGE_MAX_DIM, seed=7)
model.generator = torch.Generator(device=dev).manual_seed(11)
def step():
    return workflow.train_step(model, opt, list(batch))
def timeit(fn, n=12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(4): step()
torch.cuda.current_stream().wait_stream(s)
ms, terms = timeit(step)
print("eager ms/step", round(ms, 2), {k: round(float(v), 4) for k, v in terms.items()}, flush=True)
g = torch.cuda.CUDAGraph()
for gen in (model.generator, model.external_proposals.gen):
    g.register_generator_state(gen)
t0 = time.perf_counter()
with torch.cuda.graph(g, stream=s):
    static_terms = step()
torch.cuda.synchronize()
print("captured in %.1f s" % (time.perf_counter() - t0), flush=True)
p0 = next(model.parameters()).detach().clone()
ms, _ = timeit(g.replay)
print("graph ms/step", round(ms, 2), {k: round(float(v), 4) for k, v in static_terms.items()}, flush=True)
print("parameters moved:", float((next(model.parameters()).detach() - p0).abs().max()) > 0)

if "--replay" in sys.argv:        # for an outside tracer (rocprofv3 --kernel-trace + scripts/gpu_idle.py): replays only
    n = int(sys.argv[sys.argv.index("--replay") + 1])
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
