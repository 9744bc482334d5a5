"""What a per-stage batched weight-gradient launch could reach: the C4 shapes at batch 4 (one launch per layer, as the
step runs them) against the same layer at batch 4 x 23 (the arithmetic of 23 layers in ONE launch: same tiles, 23x the
reduction length, few pixel splits)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_intertwiner_amd import _lib
L = _lib.load()
DEV = "cuda:0"
for name, Cin, Cout, R in (("C4 1x1 256->1024", 256, 1024, 1), ("C4 1x1 1024->256", 1024, 256, 1), ("C4 3x3 256->256", 256, 256, 3),
                           ("C3 1x1 128->512", 128, 512, 1), ("C3 3x3 128", 128, 128, 3)):
    H = 64 if "C4" in name else 128
    for N in (4, 16, 92):
        if "C3" in name and N == 92:
            N = 16
        x = torch.randn(N, Cin, H, H, device=DEV)
        dy = torch.randn(N, Cout, H, H, device=DEV)
        dw = torch.zeros(Cout, R, R, Cin, device=DEV)
        pad = R // 2
        hwc = 1 if Cin % 128 == 0 else 0
        def run():
            _lib.check(L.fi_conv2d_weight_grad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), N, Cin, H, H, Cout, R, R, 1, 1, pad, pad,
                                               hwc, None, _lib.OUTPUTS_ZEROED, _lib.current_stream()), "wgrad")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            run()
        b.record(); b.synchronize()
        us = a.elapsed_time(b) * 100
        fl = 2.0 * N * H * H * Cin * Cout * R * R
        print(json.dumps({"layer": name, "N": N, "us": round(us, 1), "us_per_4_images": round(us * 4 / N, 1), "TFLOPs": round(fl / us / 1e6, 1)}))
