"""fi_bn_act_backward on the big activation shapes of the step (GB/s = (reads + writes) / time)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_intertwiner_amd import _lib
L = _lib.load()
for (N, C, H, W, res) in [(4, 256, 256, 256, True), (4, 64, 256, 256, False), (4, 1024, 64, 64, True), (2048, 256, 14, 14, False)]:
    y = torch.randn(N, C, H, W, device="cuda").relu_(); dy = torch.randn_like(y)
    r = torch.randn_like(y) if res else None
    sc = torch.rand(C, device="cuda") + 0.5; ga = torch.rand(C, device="cuda") + 0.5; be = torch.randn(C, device="cuda")
    dz = torch.empty_like(y); gres = torch.empty_like(y) if res else None
    sums = torch.empty(2, C, device="cuda")
    def run():
        _lib.check(L.fi_bn_act_backward(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(sc), _lib.ptr(ga), _lib.ptr(be), _lib.ptr(r),
                                        N, C, H * W, 1, _lib.ptr(dz), _lib.ptr(gres), _lib.ptr(sums[0]), _lib.ptr(sums[1]), None, 0, 0,
                                        _lib.current_stream()), "bn")
    for _ in range(5): run()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): run()
    b.record(); b.synchronize()
    t = a.elapsed_time(b) / 20 * 1e-3
    nbytes = y.numel() * 4 * (3 + (2 if res else 0))
    print(json.dumps({"shape": [N, C, H, W], "residual": res, "us": round(t * 1e6, 1), "GBps": round(nbytes / t / 1e9, 1)}))
