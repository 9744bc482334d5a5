"""conv3x3_patch_kernel alone on the chip: TFLOP/s by shape (in-library HIP events)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_intertwiner_amd import _lib
from feature_intertwiner_amd.conv import _conv_fwd
DEV = "cuda:0"
def run(name, N, Cin, H, W, Cout, key, iters=20):
    x = torch.randn(N, Cin, H, W, device=DEV)
    w = (torch.randn(Cout, Cin, 3, 3, device=DEV) * 0.05).contiguous(memory_format=torch.channels_last)
    sc = torch.rand(Cout, device=DEV) + 0.5; b = torch.randn(Cout, device=DEV)
    for _ in range(3): _conv_fwd(x, w, b, (1, 1), (1, 1), relu=True, scale=sc)
    torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(iters): _conv_fwd(x, w, b, (1, 1), (1, 1), relu=True, scale=sc)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    n, ms = _lib.prof_get(key); us = ms / max(n, 1) * 1e3; fl = 2.0 * N * H * W * Cin * Cout * 9
    print(json.dumps({"layer": name, "us": round(us, 1), "TFLOPs": round(fl / us / 1e6, 1), "n": n,
                      "tiles": ((N * H * W + 127) // 128) * ((Cout + 127) // 128)}))
run("C4 3x3 256 N4 (256 tiles)", 4, 256, 64, 64, 256, "conv3x3_patch")
run("C4 3x3 256 N8 (512 tiles)", 8, 256, 64, 64, 256, "conv3x3_patch")
run("C4 3x3 256 N12 (768 tiles)", 12, 256, 64, 64, 256, "conv3x3_patch")
run("C4 3x3 256 N16 (1024 tiles)", 16, 256, 64, 64, 256, "conv3x3_patch")
run("C5 3x3 512 N4 (128 tiles)", 4, 512, 32, 32, 512, "conv3x3_patch")
run("C3 3x3 128 N4 (512 tiles)", 4, 128, 128, 128, 128, "conv3x3_patch")
run("FPN P2 3x3 256", 4, 256, 256, 256, 256, "conv3x3_patch")
run("mask 14x14 256 N1376", 1376, 256, 14, 14, 256, "conv3x3_patch_flat")
