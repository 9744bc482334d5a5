"""bf16-input / fp32-accumulate MFMA convolution (csrc/conv_bf16.hip, BASELINE configs[4]'s reduced-
precision conv path) against float64 torch references.

Two bars per quantity:
  * TIGHT, against the same op evaluated in float64 on operands ROUNDED TO BF16 first: the kernel rounds
    exactly those operands and accumulates in fp32, so only fp32 summation order remains --
    |d| <= 2e-5 * sqrt(K) * max|ref| (the bar of the exact fp32 kernel, tests/test_gpu_conv.py).  A wrong
    MFMA operand layout, tap order or halo would miss this by orders of magnitude.
  * LOOSE, against the unrounded float64 op: the price of bf16 operands, 2 * 2^-9 relative per product:
    |d| <= 1.6e-2 * sqrt(K) * rms(x) * rms(w)  (stated so that the fp32-vs-bf16 gap is a documented number; the
    constant covers the maximum over the 5 M outputs of the largest case, ~10 standard deviations).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [
    # N, Cin, H, W, Cout, R, S, stride, pad
    (2, 64, 20, 24, 64, 1, 1, 1, 0),       # BM = 64 tile
    (2, 256, 14, 14, 256, 3, 3, 1, 1),     # rows of 14: quads straddle output rows
    (3, 64, 19, 23, 96, 3, 3, 1, 1),       # odd sizes, Cout tail, pixel tail
    (2, 128, 12, 20, 81, 1, 1, 1, 0),      # Cout = 81 (mask conv5)
    (2, 256, 13, 17, 128, 1, 1, 2, 0),     # 1x1 stride 2 (scalar gather path)
    (2, 256, 14, 14, 512, 3, 3, 2, 1),     # 3x3 stride 2
    (1, 32, 9, 8, 40, 5, 5, 1, 2),         # generic window
    (4, 96, 8, 8, 130, 3, 3, 1, 1),        # Cin = 3 K-tiles per tap
    (1, 1024, 8, 8, 256, 1, 1, 1, 0),      # long K
    (2, 64, 128, 128, 160, 3, 3, 1, 1),    # conv3x3_patch_bf16_kernel (>= 256 tiles, W % 16 == 0), partial Cout tile
    (36, 32, 20, 16, 256, 3, 3, 1, 1),     # patch kernel: tiles cross image boundaries (H = 20), one column tile
    (9, 160, 32, 32, 160, 3, 3, 1, 1),     # patch kernel for forward AND data gradient, 5 channel blocks
    (171, 64, 14, 14, 256, 3, 3, 1, 1),    # flat patch tiles (14-wide RoI maps, odd tail tile); flat-space weight gradient
    (90, 256, 14, 14, 272, 3, 3, 1, 1),    # flat tiles forward AND data gradient, partial third Cout tile
    (70, 128, 12, 12, 256, 3, 3, 1, 1),    # flat tiles, 12 columns: the last staged column group lies outside the map
    (3, 128, 20, 24, 128, 3, 3, 1, 1),     # flat-space weight gradient: groups continue on the next row, tail tile
    (5, 64, 10, 10, 128, 1, 1, 1, 0),      # flat-space weight gradient, 1x1, 64-channel tiles, pixel tail
    (14, 64, 40, 84, 160, 3, 3, 1, 1),     # patch kernel, W = 84: the sixth column tile holds 4 of 16 columns
    (16, 64, 32, 32, 256, 1, 1, 1, 0),     # conv1x1_bf16_kernel (>= 192 tiles): one 64-channel stage
    (10, 192, 58, 58, 130, 1, 1, 1, 0),    # conv1x1_bf16_kernel: 3 stages, partial Cout tile, pixel tail, tiles across images
]


# the two 16-bit operand types: (torch dtype, operand rounding 2^-(significand bits + 1)); bf16 is csrc/conv_bf16.hip,
# fp16 the same kernels on v_mfma_f32_32x32x16_f16 (csrc/conv_f16.hip, BASELINE configs[4]'s "fp16 MFMA conv path")
LOWP = {"bf16": (torch.bfloat16, 2.0 ** -9), "fp16": (torch.float16, 2.0 ** -12)}


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("case", CASES)
def test_bf16_conv_forward_backward(case, precision):
    from feature_intertwiner_amd import conv as C
    N, Cin, H, W, Cout, R, S, st, pd = case
    dtype, ulp = LOWP[precision]
    _bf = lambda t: t.to(dtype).double()
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, R, S, generator=g) / math.sqrt(Cin * R * S)
    b = torch.randn(Cout, generator=g)
    gy = torch.randn(N, Cout, (H + 2 * pd - R) // st + 1, (W + 2 * pd - S) // st + 1, generator=g)

    def ref(xd, wd, gyd):
        xd, wd = xd.clone().requires_grad_(True), wd.clone().requires_grad_(True)
        y = F.conv2d(xd, wd, b.double(), stride=st, padding=pd)
        return y.detach(), xd, wd, y
    # forward reference on bf16-rounded operands; gradient references with the operands THEIR kernel rounds:
    # dX = conv_T(bf16(dY), bf16(W)),  dW = corr(bf16(dY), bf16(X))
    y_t = F.conv2d(_bf(x), _bf(w), b.double(), stride=st, padding=pd)
    xd = _bf(x).requires_grad_(True)
    wd = _bf(w).requires_grad_(True)
    F.conv2d(xd, wd, None, stride=st, padding=pd).backward(_bf(gy))
    y_full = F.conv2d(x.double(), w.double(), b.double(), stride=st, padding=pd)

    C.set_conv_precision(precision)
    try:
        xg = x.to(DEV).requires_grad_(True)
        wg = w.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)   # as parameters are stored
        bg = b.to(DEV).requires_grad_(True)
        C.FLOP_LOG = {}
        y = C.conv2d(xg, wg, bg, (st, st), (pd, pd))
        y.backward(gy.to(DEV))
        used = dict(C.FLOP_LOG)
    finally:
        C.set_conv_precision("fp32")
        C.FLOP_LOG = None
    assert "conv_bf16_fwd" in used and "conv_bf16_wgrad" in used, used        # the bf16 kernels really ran
    K = Cin * R * S
    tight = lambda r, k: 2e-5 * math.sqrt(k) * (r.abs().max().item() + 1e-6)
    assert (y.detach().cpu().double() - y_t).abs().max().item() <= tight(y_t, K)
    if Cout % 32 == 0:
        assert (xg.grad.cpu().double() - xd.grad).abs().max().item() <= tight(xd.grad, Cout * R * S)
    else:
        # the data gradient contracts over Cout: not a multiple of 32 -> the exact fp32 kernel ran (unrounded)
        xf, wf = x.double().requires_grad_(True), w.double()
        F.conv2d(xf, wf, None, stride=st, padding=pd).backward(gy.double())
        assert (xg.grad.cpu().double() - xf.grad).abs().max().item() <= tight(xf.grad, Cout * R * S)
    P = N * gy.shape[2] * gy.shape[3]
    assert (wg.grad.cpu().double() - wd.grad).abs().max().item() <= tight(wd.grad, P)
    assert (bg.grad.cpu().double() - gy.double().sum((0, 2, 3))).abs().max().item() <= tight(gy.double().sum((0, 2, 3)), P)
    # the documented cost of 16-bit operands (1.6e-2 = 8 * 2^-9 for bf16)
    loose = 8 * ulp * math.sqrt(K) * x.double().pow(2).mean().sqrt().item() * w.double().pow(2).mean().sqrt().item()
    assert (y.detach().cpu().double() - y_full).abs().max().item() <= loose


def test_bf16_fused_epilogue_and_layouts():
    """conv + eval-BN + shortcut + ReLU in the bf16 kernel (NCHW) and the channels-last output used by the
    Dev make-up layer; values equal the fp32 kernel run on bf16-rounded operands."""
    from feature_intertwiner_amd import conv as C
    torch.manual_seed(5)
    conv = C.Conv2d(64, 128, 3, padding=1).to(DEV)
    bn = torch.nn.BatchNorm2d(128).to(DEV).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.2)
        bn.running_var.uniform_(0.5, 1.5)
        bn.weight.normal_(1, 0.1)
        bn.bias.normal_(0, 0.1)
    x = torch.randn(2, 64, 18, 22, device=DEV)
    res = torch.randn(2, 128, 18, 22, device=DEV)
    rounded = C.Conv2d(64, 128, 3, padding=1).to(DEV)
    with torch.no_grad():
        rounded.weight.copy_(conv.weight.to(torch.bfloat16).float())
        rounded.bias.copy_(conv.bias)
    xr = x.to(torch.bfloat16).float()
    with torch.no_grad():
        exp = C.conv_bn_act(xr, rounded, bn, relu=True, residual=res)                     # exact fp32 kernel
        exp_cl = C.conv_bn_act(xr, rounded, bn, relu=True, channels_last_out=True)
        C.set_conv_precision("bf16")
        try:
            got = C.conv_bn_act(x, conv, bn, relu=True, residual=res)
            got_cl = C.conv_bn_act(x, conv, bn, relu=True, channels_last_out=True)
        finally:
            C.set_conv_precision("fp32")
    assert got_cl.is_contiguous(memory_format=torch.channels_last)
    for a, e in ((got, exp), (got_cl, exp_cl)):
        assert (a - e).abs().max().item() <= 2e-5 * math.sqrt(64 * 9) * e.abs().max().item()


def test_bf16_train_step_tracks_fp32():
    """The detector's train step with the bf16 conv path: same losses as the fp32 path within the bf16
    operand error accumulated through ~100 layers (5 % on each term), and it learns."""
    from feature_intertwiner_amd import conv as C
    from feature_intertwiner_amd.config import make_config
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import compute_loss, set_optimizer, train_step
    batch = synthetic_batch(2, 256, device=DEV)
    terms = {}
    for mode in ("fp32", "bf16"):
        torch.manual_seed(7)
        cfg = make_config("resnet50", 256, 2, 64, dev_switch=True, loss_choice="l2", conv_precision=mode)
        model = MaskRCNN(cfg).to(DEV)
        model.external_proposals = SyntheticProposals(batch[2], 256, seed=7)
        model.generator = torch.Generator(device=DEV).manual_seed(5)
        with torch.no_grad():
            _, terms[mode] = compute_loss(model, list(batch))
        if mode == "bf16":
            opt = set_optimizer(model, cfg.TRAIN)
            hist = [float(train_step(model, opt, list(batch))["total"]) for _ in range(5)]
            assert hist[-1] < hist[0] and all(math.isfinite(h) for h in hist)
    assert C.conv_precision() == "fp32"            # the model restores the process-wide default after its pass
    for k in ("rpn_cls", "rpn_bbox", "mrcnn_cls", "mrcnn_bbox", "mrcnn_mask"):
        a, b = float(terms["fp32"][k]), float(terms["bf16"][k])
        assert abs(a - b) <= 0.05 * abs(a) + 1e-3, (k, a, b)


@pytest.mark.parametrize("shape", [
    (4, 256, 256, 256, 256, 3, 1),      # P2-level 3x3 of configs[2]: 2-D patch tiles, flat-space weight gradient on 2016 workgroups
    (2048, 256, 14, 14, 256, 3, 1),     # mask-head 3x3: flat patch tiles across 2048 RoI maps
    (4, 1024, 64, 64, 256, 1, 0),       # C4 conv1 1x1: conv1x1_bf16_kernel, 16 stages
])
def test_bf16_kernels_at_full_layer_sizes_against_fp32(shape):
    """The bf16 kernels at the layer sizes of BASELINE configs[2] (XCD-aware launches with thousands of workgroups,
    pixel splits, tail tiles) against the exact fp32 kernels on the same inputs: the relative error of forward, data
    gradient and weight gradient stays at the rounding of the two bf16 operands (2^-9 each, averaged down by the
    reduction) -- a mis-mapped tile or split shows up as an O(1) error."""
    from feature_intertwiner_amd import conv as C
    N, Cin, H, W, Cout, R, pd = shape
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.randn(N, Cin, H, W, device=DEV, generator=g)
    w = (torch.randn(Cout, Cin, R, R, device=DEV, generator=g) / math.sqrt(Cin * R * R)).contiguous(
        memory_format=torch.channels_last)
    gy = torch.randn(N, Cout, H, W, device=DEV, generator=g)
    out = {}
    for prec in ("fp32", "bf16"):
        C.set_conv_precision(prec)
        try:
            xg = x.clone().requires_grad_(True)
            wg = w.clone(memory_format=torch.channels_last).requires_grad_(True)
            y = C.conv2d(xg, wg, None, (1, 1), (pd, pd))
            y.backward(gy)
            out[prec] = (y.detach(), xg.grad, wg.grad)
        finally:
            C.set_conv_precision("fp32")
    for name, a, b in zip(("y", "dx", "dw"), out["fp32"], out["bf16"]):
        rel = ((a.double() - b.double()).norm() / a.double().norm()).item()
        assert rel < 6e-3, (name, rel)                      # 2^-9 * sqrt(2) = 2.8e-3 for independent roundings
        assert rel > 1e-5, (name, rel)                      # ... and the bf16 kernels really ran


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("M,K,N", [(128, 1024, 1024), (2000, 12544, 1024), (192, 25088, 1024), (96, 1024, 81)])
def test_linear_on_the_16bit_kernels(M, K, N, precision):
    """conv.linear under MODEL.CONV_PRECISION bf16 / fp16: the three products of a head FC (y = x W^T, dX, dW) on the
    16-bit MFMA kernels.  Tight bar against float64 on operands rounded to the 16-bit type (what each product's
    kernel rounds), as for the convolutions above."""
    from feature_intertwiner_amd import conv as C
    dtype, ulp = LOWP[precision]
    r = lambda t: t.to(dtype).double()
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    gy = torch.randn(M, N, generator=g)
    y_t = r(x) @ r(w).t() + b.double()
    dx_t = r(gy) @ r(w)
    dw_t = r(gy).t() @ r(x)
    C.set_conv_precision(precision)
    try:
        C.FLOP_LOG = {}
        xg, wg, bg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        y = C.linear(xg, wg, bg)
        y.backward(gy.to(DEV))
        used = dict(C.FLOP_LOG)
    finally:
        C.set_conv_precision("fp32")
        C.FLOP_LOG = None
    assert "conv_bf16_fwd" in used and "conv_bf16_wgrad" in used, used
    tight = lambda ref, k: 2e-5 * math.sqrt(k) * (ref.abs().max().item() + 1e-6)
    assert (y.detach().cpu().double() - y_t).abs().max().item() <= tight(y_t, K)
    assert (xg.grad.cpu().double() - dx_t).abs().max().item() <= tight(dx_t, N)
    assert (wg.grad.cpu().double() - dw_t).abs().max().item() <= tight(dw_t, M)
    assert (bg.grad.cpu().double() - gy.double().sum(0)).abs().max().item() <= tight(gy.double().sum(0), M)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("case", [
    (4, 256, 64, 64, 128, 1, True),      # conv1x1 weights-in-registers kernel
    (2, 128, 64, 64, 128, 3, False),     # 3x3 patch kernel, 8 x 16 tiles
    (640, 128, 14, 14, 128, 3, True),    # 3x3 patch kernel, flat tiles (RoI maps)
    (3, 64, 13, 11, 40, 3, True),        # general kernel, ragged rows
    (2, 64, 32, 32, 64, 1, False),       # general kernel, full tiles
])
def test_gated_epilogue_of_the_16bit_kernels(case, precision):
    """fi_conv*_forward_gated_{bf16,f16}: the gate is a select after the unchanged arithmetic."""
    from feature_intertwiner_amd import conv as C
    N, Cin, H, W, Cout, k, res = case
    g = torch.Generator(device="cpu").manual_seed(N + Cin + H + k)
    x = torch.randn(N, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (1.0 / (Cin * k * k) ** 0.5)).to(DEV)
    r = torch.randn(N, Cout, H, W, generator=g).to(DEV) if res else None
    gate = torch.randn(N, Cout, H, W, generator=g).to(DEV)
    gate[gate.abs() < 0.2] = 0.0
    wt = w.permute(0, 2, 3, 1).contiguous()
    y = C._conv_fwd(x, wt, None, (1, 1), (k // 2, k // 2), residual=r, w_tap_major=True, gate=gate, precision=precision)
    plain = C._conv_fwd(x, wt, None, (1, 1), (k // 2, k // 2), residual=r, w_tap_major=True, precision=precision)
    assert torch.equal(y, plain * (gate > 0))
    ref = F.conv2d(x.double(), w.double(), padding=k // 2) + (r.double() if res else 0.0)
    assert float((plain.double() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_stage_gradients_on_the_16bit_kernels_track_fp32(precision):
    """The unscaled-gradient backward (scaled W^T, gates, BatchNorm sums from the weight gradient, bias sums inside the
    16-bit weight-gradient kernel) on a ResNet stage against the fp32 run and against the older form of the backward
    (fi_bn_act_backward) on the same 16-bit kernels."""
    import torch.nn as nn
    from feature_intertwiner_amd import conv as C
    from feature_intertwiner_amd.sub_module import ResNet
    torch.manual_seed(5)
    net = ResNet("resnet50")
    net.inplanes = 128
    stage = net.make_layer(net.block, 64, 3, stride=2)
    for m in stage.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    stage = stage.eval().to(DEV)
    x = torch.randn(2, 128, 32, 48, device=DEV)
    gy = None
    grads = {}
    try:
        for name, prec, unscaled in (("fp32", "fp32", True), ("new", precision, True), ("old", precision, False)):
            C.set_conv_precision(prec)
            C._UNSCALED_BACKWARD = unscaled
            for p in stage.parameters():
                p.grad = None
            C.prepare_step(stage)
            C.prepare_step(stage)
            xg = x.clone().requires_grad_(True)
            y = stage(xg * 1.0)
            gy = torch.randn_like(y) if gy is None else gy
            y.backward(gy)
            torch.cuda.synchronize()
            grads[name] = {"x": xg.grad.clone(), **{k: p.grad.clone() for k, p in stage.named_parameters()}}
    finally:
        C.set_conv_precision("fp32")
        C._UNSCALED_BACKWARD = True
        C.invalidate_step_state()
    # 16-bit operands through 9 layers forward and backward: ~0.1 (bf16) / ~0.06 (fp16) of the largest element away from
    # fp32, in EITHER form of the backward; the two forms differ from each other by much less (where the BatchNorm scale
    # is rounded into the operands), and neither is biased
    far = 0.15 if precision == "bf16" else 0.09
    for k, ref in grads["fp32"].items():
        new, old = grads["new"][k], grads["old"][k]
        top = float(ref.abs().max()) + 1e-6
        assert float((new - ref).abs().max()) <= far * top, k
        assert float((new - old).abs().max()) <= 0.4 * far * top, k
        ratio = float((new * ref).sum() / (ref * ref).sum().clamp(min=1e-30))
        assert abs(ratio - 1.0) < 3e-2, (k, ratio)


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_weight_grad_batch_16bit_equals_separate_launches(prec):
    """fi_conv2d_weight_grad_batch_{bf16,f16}: n problems of one geometry in one launch of the flat 16-bit weight-gradient
    kernel against n separate launches (same operand rounding; the fp32 atomics add the pixel splits in another order) --
    1x1 and 3x3, with the bias sums, more problems than FI_WGRAD_BATCH_MAX, and a geometry that falls back to the loop."""
    import ctypes
    from feature_intertwiner_amd import _lib
    L = _lib.load()
    torch.manual_seed(5)
    single = getattr(L, "fi_conv2d_weight_grad_db_%s" % prec)
    batch = getattr(L, "fi_conv2d_weight_grad_batch_%s" % prec)
    for N, Cin, H, W, Cout, R, n, with_db in [(2, 256, 16, 16, 512, 1, 5, True), (2, 128, 16, 16, 128, 3, 4, True),
                                              (1, 128, 8, 8, 64, 3, 26, False), (2, 128, 6, 6, 128, 3, 3, True)]:
        pad = R // 2
        xs = [torch.randn(N, Cin, H, W, device=DEV) for _ in range(n)]
        dys = [torch.randn(N, Cout, H, W, device=DEV) for _ in range(n)]
        got = [torch.zeros(Cout, R, R, Cin, device=DEV) for _ in range(n)]
        gdb = [torch.zeros(Cout, device=DEV) for _ in range(n)]
        arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        _lib.check(batch(arr(xs), arr(dys), arr(got), arr(gdb) if with_db else None, n, N, Cin, H, W, Cout, R, R, 1, 1, pad,
                         pad, 1, _lib.OUTPUTS_ZEROED, _lib.current_stream()), "batch")
        for i in range(n):
            one = torch.zeros(Cout, R, R, Cin, device=DEV)
            odb = torch.zeros(Cout, device=DEV)
            _lib.check(single(_lib.ptr(xs[i]), _lib.ptr(dys[i]), _lib.ptr(one), _lib.ptr(odb) if with_db else None, N, Cin,
                              H, W, Cout, R, R, 1, 1, pad, pad, _lib.OUTPUTS_ZEROED, _lib.current_stream()), "single")
            scale = float(one.abs().max())
            assert float((got[i] - one).abs().max()) <= 2e-5 * (N * H * W) ** 0.5 * scale, (N, Cin, Cout, R, i)
            if with_db:
                assert float((gdb[i] - odb).abs().max()) <= 1e-3
