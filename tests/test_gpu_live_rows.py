"""The live-count entry points (include/fi_capi.h: fi_conv2d_forward_live, fi_gemm_nt_rows and their 16-bit twins): a batch
of static capacity whose real size is a DEVICE integer -- the Dev stage without its host read (Dev.static_shapes,
lib/sub_module.py:437-540 of the reference is the data-dependent original).  The live part must equal the plain call on the
live part alone; what lies past it is unspecified (tiles that are completely past it are not even computed)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bn(c, g):
    bn = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(c, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
    return bn.to(DEV).eval()


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("n_live", [0, 1, 37, 130, 192])
def test_feature_extractor_stages_with_a_device_row_count(precision, n_live):
    """The three stages of Dev.feat_extract (3x3 stride-2 conv, full-window conv, 1x1 conv, each + eval BatchNorm + ReLU)
    on a 192-row batch with n_live real rows and NaN bit patterns behind them."""
    from feature_intertwiner_amd import conv as C
    g = torch.Generator().manual_seed(5 + n_live)
    cap = 192
    convs = [C.Conv2d(64, 128, 3, stride=2, padding=1), C.Conv2d(128, 256, 7), C.Conv2d(256, 128, 1)]
    convs[1].full_window = True
    convs = [c.to(DEV) for c in convs]
    bns = [_bn(128, g), _bn(256, g), _bn(128, g)]
    x = torch.randn(cap, 64, 14, 14, generator=g).to(DEV)
    x[n_live:] = float("nan")
    live = torch.tensor([n_live], dtype=torch.int32, device=DEV)
    C.set_conv_precision(precision)
    try:
        with torch.no_grad():
            v, ref = x, x[:max(n_live, 1)].clone() if n_live else None
            for c, b in zip(convs, bns):
                v = C.conv_bn_act(v, c, b, relu=True, live=live)
            if n_live:
                r = x[:n_live].contiguous()
                if precision != "fp32" and n_live % 64:      # the 16-bit GEMM wants row multiples of 64: pad with zeros
                    r = torch.cat([r, torch.zeros(64 - n_live % 64, 64, 14, 14, device=DEV)])
                for c, b in zip(convs, bns):
                    r = C.conv_bn_act(r, c, b, relu=True)
                r = r[:n_live]
    finally:
        C.set_conv_precision("fp32")
    torch.cuda.synchronize()
    assert v.shape == (cap, 128, 1, 1)
    if n_live:
        got = v[:n_live]
        assert torch.isfinite(got).all()
        if precision == "fp32":
            assert torch.equal(got, r)               # the same kernels on the same rows, deterministic split
        else:
            # fp32 atomics over the K split: the order of the partial sums is not fixed, and a last-bit difference of a
            # stage's output can round to the other 16-bit neighbour as the next stage's operand (2^-9 of that operand)
            assert (got - r).abs().max().item() <= 2e-3 * (r.abs().max().item() + 1e-6)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_linear_with_a_device_row_count(precision):
    from feature_intertwiner_amd import conv as C
    g = torch.Generator().manual_seed(9)
    M, K, N = 384, 1024, 256
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    C.set_conv_precision(precision)
    try:
        with torch.no_grad():
            full = C.linear(x, w, b)
            for n_live in (0, 5, 64, 200, 384):
                xx = x.clone()
                xx[n_live:] = float("inf")
                y = C.linear(xx, w, b, torch.tensor([n_live], dtype=torch.int32, device=DEV))
                torch.cuda.synchronize()
                if precision == "fp32":
                    assert torch.equal(y[:n_live], full[:n_live]), n_live
                else:
                    assert n_live == 0 or (y[:n_live] - full[:n_live]).abs().max().item() <= 1e-4 * full.abs().max().item()
                # rows of the tiles that lie completely behind the count were not computed: no Inf / NaN from them
                tile = 128
                behind = (n_live + tile - 1) // tile * tile
                assert torch.isfinite(y[behind:]).all(), n_live
    finally:
        C.set_conv_precision("fp32")
