"""Every convolution launch of ONE train step of the bench workload, grouped by (entry point, shape): launches,
mean time (HIP events around the C call), algorithmic TFLOP/s, and the time that would be saved at a target
rate -- the list to work down when raising the conv stack's MFMA efficiency.
    python scripts/conv_shapes.py [--bf16] [--target 135]"""
import collections
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FI_DEAD_SIDE", "0")
os.environ.setdefault("FI_WGRAD_SIDE_PIXELS", "0")      # weight gradients on the main stream: the events bracket the kernel
# ... and so is every other side stream (round 5's file had three GEMMs of the OT module's backward at 0.1-0.9 TFLOP/s: 8
# workgroups of the meta loss on the THIRD stream waiting for CU slots next to a main-stream convolution -- the events
# bracketed the wait, not the kernel)
os.environ.setdefault("FI_META_SIDE", "0")
os.environ.setdefault("FI_BIG_SIDE", "0")
os.environ.setdefault("FI_PROPOSAL_SIDE", "0")
from feature_intertwiner_amd import _lib  # noqa: E402
from feature_intertwiner_amd.config import make_config  # noqa: E402
from feature_intertwiner_amd.model import MaskRCNN  # noqa: E402
from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch  # noqa: E402
from feature_intertwiner_amd.workflow import set_optimizer, train_step  # noqa: E402

bf16 = "--bf16" in sys.argv
target = float(sys.argv[sys.argv.index("--target") + 1]) if "--target" in sys.argv else (600.0 if bf16 else 135.0)
dev = "cuda:0"
torch.manual_seed(2000)
cfg5 = "--cfg5" in sys.argv          # the single-GPU slice of BASELINE configs[4]: 2 x 1344^2, 1000 RoIs/image
SIZE, BATCH, ROIS = (1344, 2, 1000) if cfg5 else (1024, 4, 512)
cfg = make_config("resnet101", SIZE, BATCH, ROIS, dev_switch=True, loss_choice="ot", ot_L=50,
                  conv_precision="bf16" if bf16 else "fp32")
model = MaskRCNN(cfg).to(dev)
opt = set_optimizer(model, cfg.TRAIN)
batch = synthetic_batch(BATCH, SIZE, device=dev, seed=2000)
model.external_proposals = SyntheticProposals(batch[2], SIZE, seed=7)
model.generator = torch.Generator(device=dev).manual_seed(11)
for _ in range(3):
    train_step(model, opt, list(batch))
torch.cuda.synchronize()

L = _lib.load()
records = []


class Wrapped(object):
    """Stands in for the CDLL: attribute access returns the real function or a timing wrapper."""

    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if not name.startswith("fi_conv") or name.endswith("_eligible"):       # (a host-side query, not a launch)
            return fn

        def timed(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            ints = tuple(int(v) if isinstance(v, int) else None for v in a)
            records.append((name, ints, e0, e1))
            return rc
        return timed


from feature_intertwiner_amd import conv as _conv  # noqa: E402
_lib._lib = Wrapped(L)
_conv.FLOP_LOG, _conv.LIVE_LOG = {}, []      # static-capacity batches: the live share of each launch is read back (in order)
train_step(model, opt, list(batch))
torch.cuda.synchronize()
_lib._lib = L
live_shares = [sh for tag, sh in _conv.LIVE_LOG if tag == "conv"]
_conv.FLOP_LOG = _conv.LIVE_LOG = None


def describe(name, a):
    v = [x for x in a if x is not None]
    gated = "_gated" in name
    name = name.replace("_gated", "")
    if name.endswith("_live") or "_live_" in name:
        # (same leading arguments as the plain entry point; flops = the live rows only, as the bench counts them)
        share = live_shares.pop(0) if live_shares else 1.0
        key, fl = describe(name.replace("_live", ""), a)
        return key + " [static capacity, %.0f %% live]" % (100 * share), fl * share
    if name in ("fi_conv2d_forward", "fi_conv2d_forward_bf16", "fi_conv2d_forward_f16"):
        N, Cin, H, W, Cout, R, S, sh, sw, ph, pw, relu, layout, oh, ow, ocl = v[:16]
        OH = oh or (H + 2 * ph - R) // sh + 1
        OW = ow or (W + 2 * pw - S) // sw + 1
        fl = 2.0 * N * Cout * OH * OW * Cin * R * S
        return "fwd%s N%d %dx%d Cin%d->Cout%d k%dx%d s%d p%d lay%d%s" % (
            "_bf16" if "bf16" in name else "", N, H, W, Cin, Cout, R, S, sh, ph, layout, " cl" if ocl else ""), fl
    if name in ("fi_conv3x3_forward_bf16w",):
        N, Cin, H, W, Cout = v[:5]
        return "fwd_bf16w3x3 N%d %dx%d Cin%d->Cout%d" % (N, H, W, Cin, Cout), 2.0 * N * Cout * H * W * Cin * 9
    if name in ("fi_conv1x1_forward_bf16w",):
        N, Cin, HW, Cout = v[:4]
        return "fwd_bf16w1x1 N%d HW%d Cin%d->Cout%d" % (N, HW, Cin, Cout), 2.0 * N * Cout * HW * Cin
    if name in ("fi_conv2d_weight_grad_batch", "fi_conv2d_weight_grad_batch_bf16", "fi_conv2d_weight_grad_batch_f16"):
        # (x[], dy[], dw[], db[], n, N, Cin, H, W, Cout, R, S, 1, 1, ph, pw, ...) -- the 16-bit entry points take the same
        # arguments (round 5's cfg5 file had this row at 0.0 TFLOP/s: the script did not know the name)
        n, N, Cin, H, W, Cout, R, S, sh, sw, ph, pw = v[:12]
        fl = 2.0 * n * N * Cout * H * W * Cin * R * S
        return "wgrad%s x%d (one launch) N%d %dx%d Cin%d->Cout%d k%dx%d s%d p%d" % (
            "_bf16" if ("bf16" in name or "f16" in name) else "", n, N, H, W, Cin, Cout, R, S, sh, ph), fl
    name = name.replace("_db_", "_")
    if name in ("fi_conv2d_weight_grad", "fi_conv2d_weight_grad_bf16"):
        N, Cin, H, W, Cout, R, S, sh, sw, ph, pw = v[:11]
        OH = (H + 2 * ph - R) // sh + 1
        OW = (W + 2 * pw - S) // sw + 1
        fl = 2.0 * N * Cout * OH * OW * Cin * R * S
        return "wgrad%s N%d %dx%d Cin%d->Cout%d k%dx%d s%d p%d" % ("_bf16" if "bf16" in name else "", N, H, W, Cin, Cout, R, S, sh, ph), fl
    return name, 0.0


agg = collections.OrderedDict()
for name, a, e0, e1 in records:
    key, fl = describe(name, a)
    e = agg.setdefault(key, [0, 0.0, 0.0])
    e[0] += 1
    e[1] += e0.elapsed_time(e1) * 1e3
    e[2] += fl
rows = []
for key, (n, us, fl) in agg.items():
    tf = fl / (us * 1e-6) / 1e12 if us > 0 else 0.0
    lost = us - fl / (target * 1e12) * 1e6
    rows.append((lost, key, n, us, fl, tf))
tot_us = sum(r[3] for r in rows)
tot_fl = sum(r[4] for r in rows)
print("conv launches: %d, %.2f ms, %.2f TFLOP, %.1f TFLOP/s; time above %.0f TFLOP/s: %.2f ms" % (
    sum(r[2] for r in rows), tot_us / 1e3, tot_fl / 1e12, tot_fl / tot_us / 1e6, target, sum(max(r[0], 0) for r in rows) / 1e3))
print("%9s %5s %10s %8s  %s" % ("lost_us", "n", "total_us", "TFLOP/s", "shape"))
for lost, key, n, us, fl, tf in sorted(rows, reverse=True)[:70]:
    print("%9.1f %5d %10.1f %8.1f  %s" % (lost, n, us, tf, key))
