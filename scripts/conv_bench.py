"""Per-shape throughput of the MFMA implicit-GEMM conv kernels (forward, dgrad, wgrad) on the
layer shapes of the BASELINE workload (ResNet-101-FPN, 4 x 1024^2, 512 RoIs/img)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_intertwiner_amd import _lib
from feature_intertwiner_amd.conv import _conv_fwd

DEV = "cuda:0"
SHAPES = [
    # name, N, Cin, H, W, Cout, R, stride, pad
    ("mask_head 3x3", 2048, 256, 14, 14, 256, 3, 1, 1),
    ("C4 conv2 3x3", 4, 256, 64, 64, 256, 3, 1, 1),
    ("C2 conv2 3x3", 4, 64, 256, 256, 64, 3, 1, 1),
    ("C3 conv2 3x3", 4, 128, 128, 128, 128, 3, 1, 1),
    ("C5 conv2 3x3", 4, 512, 32, 32, 512, 3, 1, 1),
    ("FPN P2 smooth 3x3", 4, 256, 256, 256, 256, 3, 1, 1),
    ("RPN shared P2 3x3", 4, 256, 256, 256, 512, 3, 1, 1),
    ("feat_extract 3x3 s2", 2048, 256, 14, 14, 512, 3, 2, 1),
    ("C4 conv1 1x1", 4, 1024, 64, 64, 256, 1, 1, 0),
    ("C4 conv3 1x1", 4, 256, 64, 64, 1024, 1, 1, 0),
    ("C2 conv3 1x1", 4, 64, 256, 256, 256, 1, 1, 0),
    ("mask deconv as 1x1", 2048, 256, 14, 14, 1024, 1, 1, 0),
    ("stem 7x7 s2", 4, 3, 1024, 1024, 64, 7, 2, 3),
    ("C3 conv1 1x1", 4, 512, 128, 128, 128, 1, 1, 0),
    ("C3 conv3 1x1", 4, 128, 128, 128, 512, 1, 1, 0),
    ("C5 conv1 1x1", 4, 2048, 32, 32, 512, 1, 1, 0),
    ("C5 conv3 1x1", 4, 512, 32, 32, 2048, 1, 1, 0),
]


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e-3 / iters


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only = args[0] if args else None
    bf16 = "--bf16" in sys.argv
    L = _lib.load()
    prec = "bf16" if bf16 else "fp32"
    for name, N, Cin, H, W, Cout, R, st, pd in SHAPES:
        if only and only not in name:
            continue
        x = torch.randn(N, Cin, H, W, device=DEV)
        w = torch.randn(Cout, Cin, R, R, device=DEV) * 0.05
        b = torch.randn(Cout, device=DEV)
        y = _conv_fwd(x, w, b, (st, st), (pd, pd), precision=prec)
        OH, OW = y.shape[2], y.shape[3]
        flops = 2.0 * N * Cout * OH * OW * Cin * R * R
        t_f = timeit(lambda: _conv_fwd(x, w, b, (st, st), (pd, pd), precision=prec))
        # the fused epilogue of a bottleneck's last convolution: eval-BN scale/shift + shortcut + ReLU
        sc = torch.rand(Cout, device=DEV) + 0.5
        res = torch.randn_like(y)
        t_e = timeit(lambda: _conv_fwd(x, w, b, (st, st), (pd, pd), relu=True, scale=sc, residual=res, precision=prec))
        dw = torch.empty_like(w)
        dy = torch.randn_like(y)
        same = st == 1 and OH == H and OW == W and (H * W) % 4 == 0 and W >= 4
        hwc = 1 if (Cin % 128 == 0 or (Cin == 64 and same)) else 0

        dwt = torch.empty(Cout, R, R, Cin, device=DEV)

        def wg():
            if bf16:
                _lib.check(L.fi_conv2d_weight_grad_bf16(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dwt), N, Cin, H, W, Cout, R, R,
                                                        st, st, pd, pd, 0, _lib.current_stream()), "wgrad")
                return
            _lib.check(L.fi_conv2d_weight_grad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), N, Cin, H, W, Cout, R, R,
                                               st, st, pd, pd, hwc, None, 0, _lib.current_stream()), "wgrad")
        t_w = timeit(wg)
        print(json.dumps({"precision": prec, "layer": name, "GFLOP": round(flops / 1e9, 1), "fwd_us": round(t_f * 1e6, 1),
                          "fwd_TFLOPs": round(flops / t_f / 1e12, 1), "fwd_fused_us": round(t_e * 1e6, 1), "wgrad_us": round(t_w * 1e6, 1),
                          "wgrad_TFLOPs": round(flops / t_w / 1e12, 1)}))


if __name__ == "__main__":
    main()
