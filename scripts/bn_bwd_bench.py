"""fi_bn_act_backward on the ResNet-101 layer shapes of the headline step (N=4, 1024^2): time and achieved
HBM rate per shape, narrow (conv1/conv2: dy,y read, dz written) and wide (conv3: + shortcut read and the
masked gradient for the shortcut written)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from feature_intertwiner_amd import _lib

dev = "cuda:0"
L = _lib.load()
N = 4
SHAPES = [(64, 256, 6, False), (256, 256, 3, True), (128, 128, 8, False), (512, 128, 4, True),
          (256, 64, 46, False), (1024, 64, 23, True), (512, 32, 6, False), (2048, 32, 3, True)]
total = 0.0
for C, S, per_step, wide in SHAPES:
    HW = S * S
    # enough distinct tensor sets that a launch never finds its operands in the 256 MB Infinity Cache
    per_set = 4 * N * C * HW * (5 if wide else 3)
    sets = max(2, min(64, int(1.5e9 // per_set)))
    DY = [torch.randn(N, C, S, S, device=dev) for _ in range(sets)]
    Y = [torch.randn(N, C, S, S, device=dev).relu_() for _ in range(sets)]
    RES = [torch.randn(N, C, S, S, device=dev) if wide else None for _ in range(sets)]
    DZ = [torch.empty(N, C, S, S, device=dev) for _ in range(sets)]
    G = [torch.empty(N, C, S, S, device=dev) if wide else None for _ in range(sets)]
    turn = [0]
    scale = torch.rand(C, device=dev) + 0.5
    gamma = torch.rand(C, device=dev) + 0.5
    beta = torch.randn(C, device=dev)
    sums = torch.zeros(3 * C, device=dev)

    def run():
        k = turn[0] = (turn[0] + 1) % sets
        dy, y, res, dz, g = DY[k], Y[k], RES[k], DZ[k], G[k]
        _lib.check(L.fi_bn_act_backward(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(scale), _lib.ptr(gamma), _lib.ptr(beta),
                                        _lib.ptr(res), N, C, HW, 1, _lib.ptr(dz), _lib.ptr(g), _lib.ptr(sums[:C]),
                                        _lib.ptr(sums[C:2 * C]), None, 0, _lib.OUTPUTS_ZEROED, _lib.current_stream()), "bn")
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(50, 2 * sets)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    nbytes = 4 * N * C * HW * (5 if wide else 3)
    total += us * per_step
    print("C=%5d %3dx%-3d %-6s %8.1f us  %7.0f GB/s   x%d/step = %.2f ms" % (C, S, S, "wide" if wide else "narrow", us,
                                                                             nbytes / us / 1e3, per_step, us * per_step / 1e3))
print("per step: %.2f ms" % (total / 1e3))
