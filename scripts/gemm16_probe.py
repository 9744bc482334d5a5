"""conv.linear's forward product on the 16-bit kernels (the weight-gradient kernel as a GEMM with an atomic split over K)
for the fully connected shapes of the step: microseconds and TFLOP/s.   FI_WG16_TARGET=<workgroups> python scripts/gemm16_probe.py
--fp32: fi_gemm_nt (deterministic split: slabs + ordered reduction) instead; its knob is FI_GEMM_TARGET."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from feature_intertwiner_amd import conv as C
dev = "cuda:0"
PREC = "fp32" if "--fp32" in sys.argv else "bf16"
C.set_conv_precision(PREC)
shapes = [(2048, 12544, 1024), (2048, 1024, 1024), (2048, 25088, 1024), (3008, 25088, 1024), (1408, 25088, 1024), (1024, 2304, 512),
          (2048, 1024, 512)]
for M, K, N in shapes:
    a = torch.randn(M, K, device=dev)
    b = torch.randn(N, K, device=dev)
    for _ in range(3):
        C._gemm_nt(a, b, None, precision=PREC)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        C._gemm_nt(a, b, None, precision=PREC)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    print(json.dumps({"M": M, "K": K, "N": N, "target": os.environ.get("FI_GEMM_TARGET" if PREC == "fp32" else "FI_WG16_TARGET", "default"), "us": round(us, 1),
                      "TFLOPs": round(2.0 * M * N * K / us / 1e6, 1)}), flush=True)
