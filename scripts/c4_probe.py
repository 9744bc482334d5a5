"""Probe of the short-K 1x1 layers (C4 stage): kernel time of conv1x1_reg_kernel with the epilogue operand sets the
step uses, from the in-library HIP events.  FI_DBG_1X1 (probe builds: -DFI_PROBE_1X1) adds a start delay for half of the
workgroups / removes the epilogue."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_intertwiner_amd import _lib
from feature_intertwiner_amd.conv import _conv_fwd

DEV = "cuda:0"


def run(name, N, Cin, H, W, Cout, mode, iters=40):
    x = torch.randn(N, Cin, H, W, device=DEV)
    w = torch.randn(Cout, Cin, 1, 1, device=DEV) * 0.05
    sc = torch.rand(Cout, device=DEV) + 0.5
    b = torch.randn(Cout, device=DEV)
    res = torch.randn(N, Cout, H, W, device=DEV)
    gate = torch.randn(N, Cout, H, W, device=DEV)
    kw = {"plain": {}, "fwd": dict(relu=True, scale=sc, residual=res), "dgrad": dict(residual=res, gate=gate),
          "gate": dict(gate=gate)}[mode]
    bb = b if mode in ("plain", "fwd") else None
    for _ in range(5):
        _conv_fwd(x, w, bb, (1, 1), (0, 0), **kw)
    torch.cuda.synchronize()
    _lib.prof_reset()
    _lib.prof_enable(True)
    for _ in range(iters):
        _conv_fwd(x, w, bb, (1, 1), (0, 0), **kw)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    n, ms = _lib.prof_get("conv1x1_reg")
    us = ms / max(n, 1) * 1e3
    fl = 2.0 * N * H * W * Cin * Cout
    print(json.dumps({"layer": name, "mode": mode, "dbg": os.environ.get("FI_DBG_1X1"), "us": round(us, 1),
                      "TFLOPs": round(fl / us / 1e6, 1)}))


if __name__ == "__main__":
    for mode in ("plain", "fwd", "dgrad"):
        run("C4 256->1024", 4, 256, 64, 64, 1024, mode)
    for mode in ("plain", "fwd", "gate"):
        run("C4 1024->256", 4, 1024, 64, 64, 256, mode)
    for mode in ("fwd", "dgrad"):
        run("C3 128->512", 4, 128, 128, 128, 512, mode)
