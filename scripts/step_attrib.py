"""Where the NON-library device kernels of a train step come from: torch.profiler over ONE step of the bench
workload, every device kernel attributed to (a) the forward stage that launched it -- model stages are wrapped
in record_function ranges by this script -- or (b) the autograd node that launched it in backward.
Prints, per origin: launches, device microseconds, and the kernel names; library (fi) kernels are summarised
in one line per origin."""
import collections
import functools
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FI_DEAD_SIDE", "0")
os.environ.setdefault("FI_WGRAD_SIDE_PIXELS", "0")      # one stream: kernel durations are exclusive and add up to the step
from torch.profiler import ProfilerActivity, profile, record_function  # noqa: E402

from feature_intertwiner_amd import intertwiner, layers, model as model_mod, optim, sub_module, workflow  # noqa: E402
from feature_intertwiner_amd.config import make_config  # noqa: E402
from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch  # noqa: E402


def wrap(mod, name, label=None):
    fn = getattr(mod, name)

    @functools.wraps(fn)
    def inner(*a, **k):
        with record_function("stage:" + (label or name)):
            return fn(*a, **k)
    setattr(mod, name, inner)


for n in ("proposal_layer", "prepare_det_target", "prepare_rpn_target", "compute_rpn_class_loss", "compute_rpn_bbox_loss",
          "compute_mrcnn_class_loss", "compute_mrcnn_bbox_loss", "compute_mrcnn_mask_loss_unshuffled"):
    wrap(layers, n)
    if hasattr(model_mod, n):
        wrap(model_mod, n)
wrap(model_mod, "meta_loss")
wrap(model_mod, "prepare_step")
wrap(optim, "clip_and_step")
for cls, label in ((sub_module.FPN, "fpn"), (sub_module.RPN, "rpn"), (sub_module.Dev, "dev_roi"),
                   (sub_module.Classifier, "classifier"), (sub_module.Mask, "mask")):
    wrap(cls, "forward", label)

dev = "cuda:0"
torch.manual_seed(2000)
cfg5 = "--cfg5" in sys.argv          # the single-GPU slice of BASELINE configs[4] on the bf16 kernels
SIZE, BATCH, ROIS = (1344, 2, 1000) if cfg5 else (1024, 4, 512)
cfg = make_config("resnet101", SIZE, BATCH, ROIS, dev_switch=True, loss_choice="ot", ot_L=50,
                  conv_precision="bf16" if cfg5 else "fp32")
model = model_mod.MaskRCNN(cfg).to(dev)
opt = workflow.set_optimizer(model, cfg.TRAIN)
batch = synthetic_batch(BATCH, SIZE, device=dev, seed=2000)
model.external_proposals = SyntheticProposals(batch[2], SIZE, seed=7, cycle=16)   # as bench.py: drawn before the measured steps
model.generator = torch.Generator(device=dev).manual_seed(11)
for _ in range(3):
    workflow.train_step(model, opt, list(batch))
torch.cuda.synchronize()
SHAPES = "--shapes" in sys.argv     # input shapes of the aten op behind every framework launch
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=SHAPES) as prof:
    workflow.train_step(model, opt, list(batch))
    torch.cuda.synchronize()

LIB = ("(anonymous namespace)::", "_GLOBAL__N_", "fi_calib")
# ... and every __global__ function of csrc/ by name: a kernel defined outside an anonymous namespace (the split-K
# reduction of fi_gemm_nt) carries no such marker and used to be counted as a framework launch
import glob, re
LIB_KERNELS = set()
for _f in glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "feature_intertwiner_amd", "csrc", "*.h*")):
    for _m in re.finditer(r"__global__[^;{]*?void\s+(\w+)\s*\(", open(_f).read(), re.S):
        LIB_KERNELS.add(_m.group(1))


def origin(ev):
    """stage:<name> (forward) or the autograd node (backward) or '?'; plus the aten op that launched."""
    op = None
    p = ev
    stage = None
    while p is not None:
        n = p.name
        if n.startswith("aten::") and op is None:
            op = n + ((" " + str([list(x) for x in p.input_shapes if x])[:90]) if SHAPES and p.input_shapes else "")
        if n.startswith("stage:"):
            stage = n[6:] if stage is None else stage
        if "evaluate_function: " in n:
            return "bwd:" + n.split("evaluate_function: ")[1], op
        p = p.cpu_parent
    top = op
    p = ev
    while p is not None:            # outermost aten op: what the Python code called
        if p.name.startswith("aten::"):
            top = p.name
        p = p.cpu_parent
    return ("fwd:" + stage) if stage else "?", top


acc = collections.OrderedDict()
for ev in prof.events():
    ks = ev.kernels or []
    if not ks:
        continue
    org, op = origin(ev)
    for k in ks:
        base = k.name.split("(")[0].split("<")[0].replace("void ", "").split("::")[-1].strip()
        lib = (any(s in k.name for s in LIB) or base in LIB_KERNELS) and "at::native" not in k.name and "rocprim" not in k.name
        key = (org, "<library kernels>" if lib else (op or "?") + " -> " + k.name.split("(")[0].replace("void ", "")[:70])
        e = acc.setdefault(key, [0, 0.0])
        e[0] += 1
        e[1] += k.duration

by_origin = collections.OrderedDict()
for (org, what), (n, us) in acc.items():
    o = by_origin.setdefault(org, {"lib": [0, 0.0], "other": [0, 0.0], "rows": []})
    tgt = o["lib"] if what == "<library kernels>" else o["other"]
    tgt[0] += n
    tgt[1] += us
    if what != "<library kernels>":
        o["rows"].append((us, n, what))
tot_n = sum(o["other"][0] for o in by_origin.values())
tot_us = sum(o["other"][1] for o in by_origin.values())
lib_n = sum(o["lib"][0] for o in by_origin.values())
lib_us = sum(o["lib"][1] for o in by_origin.values())
print("non-library kernels: %d launches, %.2f ms; library kernels: %d launches, %.2f ms" % (tot_n, tot_us / 1e3, lib_n, lib_us / 1e3))
for org, o in sorted(by_origin.items(), key=lambda kv: -kv[1]["other"][1]):
    if o["other"][0] == 0:
        continue
    print("\n%-60s non-lib %4d launches %9.1f us   (library: %d launches %.1f us)" % (org[:60], o["other"][0], o["other"][1], o["lib"][0], o["lib"][1]))
    for us, n, what in sorted(o["rows"], reverse=True)[:14]:
        print("      %4d x %9.1f us  %s" % (n, us, what))
