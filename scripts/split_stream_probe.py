"""Does the backbone run faster as two half-batches on two streams than as one batch (the fixed ~15 us of a one-round conv
launch hidden under the other stream's kernels)?  Forward only and forward + backward of the FPN backbone, 4 images of 1024^2."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from feature_intertwiner_amd import conv
from feature_intertwiner_amd.config import make_config
from feature_intertwiner_amd.model import MaskRCNN
dev = torch.device("cuda:0")
torch.manual_seed(1)
cfg = make_config("resnet101", 1024, 4, 512, dev_switch=True, loss_choice="ot", ot_L=50)
model = MaskRCNN(cfg).to(dev)
fpn = model.fpn
x = torch.randn(4, 3, 1024, 1024, device=dev)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def whole(train):
    conv.prepare_step(model)
    outs = fpn(x)
    if train:
        sum(o.sum() for o in outs).backward()


def halves(train):
    conv.prepare_step(model)
    cur = torch.cuda.current_stream(dev)
    res = []
    for s, part in ((s1, x[:2]), (s2, x[2:])):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            res.append(fpn(part))
    for s in (s1, s2):
        cur.wait_stream(s)
    if train:
        sum(o.sum() for r in res for o in r).backward()


def timed(fn, train, n=10):
    ctx = torch.enable_grad() if train else torch.no_grad()
    with ctx:
        for _ in range(3):
            model.zero_grad(set_to_none=True); fn(train)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            model.zero_grad(set_to_none=True); fn(train)
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for train in (False, True):
    a = timed(whole, train); b = timed(halves, train); a2 = timed(whole, train); b2 = timed(halves, train)
    print("%s: one batch of 4: %.2f / %.2f ms   two half-batches on two streams: %.2f / %.2f ms" % (
        "forward + backward" if train else "forward only", a, a2, b, b2), flush=True)
