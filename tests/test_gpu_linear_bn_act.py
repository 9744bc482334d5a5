"""conv.linear_bn_act: a convolution that is a matrix product (full-window kernel, or 1x1 on a 1x1 map) + eval-mode
BatchNorm + ReLU as one forward pass and a three-kernel backward (fi_gemm_nt_affine, fi_rows_mask_scale, fi_bn_fold_grad)
-- nn.Conv2d + nn.BatchNorm2d + nn.ReLU in the reference's heads (lib/sub_module.py:333-340, :707-716).  Checked against
the same three torch modules in float64."""
import math

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _layers(cin, cout, k, bias, g):
    from feature_intertwiner_amd import conv as C
    conv = C.Conv2d(cin, cout, k, bias=bias)
    conv.full_window = k > 1
    bn = nn.BatchNorm2d(cout)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / math.sqrt(cin * k * k))
        if bias:
            conv.bias.copy_(torch.randn(cout, generator=g) * 0.3)
        bn.weight.copy_(torch.rand(cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(cout, generator=g) * 0.2)
        bn.running_mean.copy_(torch.randn(cout, generator=g) * 0.2)
        bn.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
    return conv, bn.eval()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("M,cin,k,cout,bias,relu", [
    (128, 64, 7, 256, True, True),       # full-window "fc" convolution, whole tiles
    (100, 32, 7, 192, True, True),       # rows and columns that need padding
    (96, 1024, 1, 1024, False, True),    # 1x1 convolution on a 1x1 map
    (64, 256, 1, 128, True, False),      # no ReLU
    (2048, 256, 7, 1024, True, True),    # the box head's first layer at the headline size
])
def test_linear_bn_act_matches_conv_bn_relu_in_float64(M, cin, k, cout, bias, relu, precision):
    from feature_intertwiner_amd import conv as C
    g = torch.Generator().manual_seed(M + cin + cout)
    conv, bn = _layers(cin, cout, k, bias, g)
    x = torch.randn(M, cin, k, k, generator=g)
    gy = torch.randn(M, cout, 1, 1, generator=g)
    lowp = precision != "fp32"
    r = (lambda t: t.to(torch.bfloat16).double()) if lowp else (lambda t: t.double())
    # float64 reference on the operands the kernels see (the 16-bit kernels round x and W on their way in)
    ref_conv = nn.Conv2d(cin, cout, k, bias=bias).double()
    ref_bn = nn.BatchNorm2d(cout).double().eval()
    with torch.no_grad():
        ref_conv.weight.copy_(r(conv.weight))
        if bias:
            ref_conv.bias.copy_(conv.bias.double())
        for name in ("weight", "bias", "running_mean", "running_var"):
            getattr(ref_bn, name).copy_(getattr(bn, name).double())
    conv, bn = conv.to(DEV), bn.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    C.set_conv_precision(precision)
    try:
        for step in range(2):           # the second pass runs on the cached (scale, shift) fold
            for p in list(conv.parameters()) + list(bn.parameters()) + [xg]:
                p.grad = None
            C.refresh_bn_folds()
            y = C.conv_bn_act(xg, conv, bn, relu=relu)
            assert type(y.grad_fn).__name__ in ("ViewBackward0", "UnsafeViewBackward0", "_LinearBnActFnBackward"), y.grad_fn
            y.backward(gy.to(DEV))
    finally:
        C.set_conv_precision("fp32")
    torch.cuda.synchronize()
    K = cin * k * k
    # the reference takes its ReLU mask from the kernel's output: a pre-activation within rounding of zero may sit on the
    # other side in float64, and one such element moves a whole row of W in dx
    xr = r(x).requires_grad_(True)
    yr = ref_bn(ref_conv(xr))
    if relu:
        yr = yr * (y.detach().cpu() > 0).double()
    yr.backward(gy.double())

    def close(got, ref, n, what):
        # fp32 products of n terms; the 16-bit kernels also round g = dy * mask (and g * scale) as an operand: 2^-9
        bar = (4e-3 if lowp and what in ("dx", "dw", "dgamma") else 0) * ref.abs().max().item() + \
            3e-6 * math.sqrt(n) * (ref.abs().max().item() + 1e-6)
        err = (got.detach().cpu().double() - ref).abs().max().item()
        assert err <= bar, (what, err, bar)

    close(y, yr.detach(), K, "y")
    close(xg.grad, xr.grad, cout, "dx")
    close(conv.weight.grad, ref_conv.weight.grad, M, "dw")
    if bias:
        close(conv.bias.grad, ref_conv.bias.grad, M, "dbias")
    close(bn.weight.grad, ref_bn.weight.grad, M * 4, "dgamma")
    close(bn.bias.grad, ref_bn.bias.grad, M, "dbeta")


def test_linear_bn_act_with_a_device_row_count():
    from feature_intertwiner_amd import conv as C
    g = torch.Generator().manual_seed(3)
    conv, bn = _layers(64, 256, 7, True, g)
    conv, bn = conv.to(DEV), bn.to(DEV)
    x = torch.randn(256, 64, 7, 7, generator=g).to(DEV)
    with torch.no_grad():
        full = C.conv_bn_act(x, conv, bn, relu=True)
        for n_live in (0, 70, 256):
            xx = x.clone()
            xx[n_live:] = float("nan")
            y = C.conv_bn_act(xx, conv, bn, relu=True, live=torch.tensor([n_live], dtype=torch.int32, device=DEV))
            torch.cuda.synchronize()
            assert torch.equal(y[:n_live], full[:n_live])
            assert torch.isfinite(y[(n_live + 127) // 128 * 128:]).all()
