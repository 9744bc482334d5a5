"""fp32 MFMA implicit-GEMM convolution vs a float64 torch reference of the same op
(forward, input gradient, weight gradient, bias gradient).  Tolerance: fp32 round-off of
a K-term dot product, |d| <= 2e-5 * sqrt(K) * max|ref| (the MFMA path is an exact fp32 fma
chain; only the summation order differs from the reference)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [
    # N, Cin, H, W, Cout, R, S, stride, pad
    (2, 64, 20, 24, 64, 1, 1, 1, 0),
    (2, 256, 13, 17, 128, 1, 1, 2, 0),
    (3, 64, 19, 23, 64, 3, 3, 1, 1),
    (2, 70, 16, 16, 200, 3, 3, 1, 1),
    (2, 256, 14, 14, 512, 3, 3, 2, 1),
    (2, 3, 64, 64, 64, 7, 7, 2, 3),
    (1, 5, 11, 9, 7, 5, 3, 1, 2),
    (2, 8, 15, 15, 130, 3, 3, 2, 0),
    (4, 256, 7, 7, 1024, 7, 7, 1, 0),      # full-window -> GEMM path
    (1, 1024, 8, 8, 2048, 1, 1, 1, 0),
    # same-size stride-1 layers with Cin % 128 == 0: the row-major (16-byte) weight-gradient kernel
    (2, 128, 14, 14, 256, 3, 3, 1, 1),     # rows of 14: pixel groups continue on the next row
    (3, 256, 12, 20, 64, 3, 3, 1, 1),      # 64-row tiles, H != W
    (1, 128, 4, 4, 128, 3, 3, 1, 1),       # narrowest map the kernel accepts
    (2, 128, 6, 6, 96, 3, 3, 1, 1),        # Cout not a multiple of the tile
    (2, 128, 5, 5, 64, 3, 3, 1, 1),        # H*W % 4 != 0 -> scalar kernel
    (5, 128, 10, 10, 128, 1, 1, 1, 0),     # pixel count not a multiple of the K-step
    (2, 128, 9, 8, 128, 5, 5, 1, 2),       # generic window
    (33, 16, 64, 64, 128, 3, 3, 1, 1),     # 1056 tiles on 1024 slots: main launch + 64x64-tile tail launch
    (2, 64, 14, 14, 64, 3, 3, 1, 1),       # Cin = 64: two taps per weight-gradient column tile (9 taps: last tile half empty)
    (3, 64, 12, 20, 256, 1, 1, 1, 0),      # Cin = 64, 1x1: second half of the tile is past K
    (2, 64, 16, 16, 96, 5, 5, 1, 2),       # Cin = 64, generic window (25 taps)
    # 3x3 / stride 1 / pad 1 with >= 512 tiles: conv3x3_patch_kernel (input patch in LDS)
    (2, 32, 128, 128, 160, 3, 3, 1, 1),    # second Cout tile partial -> general epilogue
    (171, 48, 14, 14, 256, 3, 3, 1, 1),    # RoI maps: flat 128-pixel tiles across image boundaries (W = 14), odd tail tile
    (90, 256, 14, 14, 272, 3, 3, 1, 1),    # flat tiles, forward AND data gradient, partial third Cout tile
    (52, 144, 20, 20, 144, 3, 3, 1, 1),    # forward AND data gradient (flipped taps) on the patch kernel; W = 20: partial column tile
]


@pytest.mark.parametrize("case", CASES)
def test_conv_forward_backward(case):
    from feature_intertwiner_amd.conv import conv2d
    N, Cin, H, W, Cout, R, S, st, pd = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, R, S, generator=g) / math.sqrt(Cin * R * S)
    b = torch.randn(Cout, generator=g)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    yd = F.conv2d(xd, wd, bd, stride=st, padding=pd)
    gy = torch.randn(yd.shape, generator=g)
    yd.backward(gy.double())
    xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = conv2d(xg, wg, bg, (st, st), (pd, pd))
    assert y.shape == yd.shape
    y.backward(gy.to(DEV))
    K = Cin * R * S
    tol = lambda ref, k: 2e-5 * math.sqrt(k) * (ref.abs().max().item() + 1e-6)
    assert (y.detach().cpu().double() - yd.detach()).abs().max().item() <= tol(yd.detach(), K)
    assert (xg.grad.cpu().double() - xd.grad).abs().max().item() <= tol(xd.grad, Cout * R * S)
    assert (wg.grad.cpu().double() - wd.grad).abs().max().item() <= tol(wd.grad, N * yd.shape[2] * yd.shape[3])
    assert (bg.grad.cpu().double() - bd.grad).abs().max().item() <= tol(bd.grad, N * yd.shape[2] * yd.shape[3])


def test_modules_match_torch_modules():
    from feature_intertwiner_amd.conv import Conv1d, Conv2d, ConvTranspose2x2
    torch.manual_seed(0)
    ref = torch.nn.Conv2d(32, 48, 3, stride=1, padding=1)
    m = Conv2d(32, 48, 3, stride=1, padding=1)
    m.load_state_dict(ref.state_dict())
    x = torch.randn(2, 32, 10, 12)
    assert torch.allclose(m.to(DEV)(x.to(DEV)).cpu(), ref(x), rtol=1e-4, atol=1e-4)
    rt = torch.nn.ConvTranspose2d(16, 24, kernel_size=2, stride=2)
    mt = ConvTranspose2x2(16, 24, kernel_size=2, stride=2)
    mt.load_state_dict(rt.state_dict())
    x = torch.randn(3, 16, 7, 5)
    xg = x.to(DEV).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    out = mt.to(DEV)(xg)
    outr = rt(xr)
    assert torch.allclose(out.cpu(), outr, rtol=1e-4, atol=1e-4)
    gy = torch.randn_like(outr)
    out.backward(gy.to(DEV))
    outr.backward(gy)
    assert torch.allclose(xg.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(mt.weight.grad.cpu(), rt.weight.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(mt.bias.grad.cpu(), rt.bias.grad, rtol=1e-4, atol=1e-4)
    r1 = torch.nn.Conv1d(20, 12, 3, padding=1)
    m1 = Conv1d(20, 12, 3, padding=1)
    m1.load_state_dict(r1.state_dict())
    for L in (1, 6):
        x = torch.randn(4, 20, L)
        assert torch.allclose(m1.to(DEV)(x.to(DEV)).cpu(), r1(x), rtol=1e-4, atol=1e-5)


def test_relu_epilogue_and_large_shape_property():
    """fused ReLU == relu(conv); a P2-sized 3x3 conv is linear in its input (size-independent)."""
    from feature_intertwiner_amd.conv import _conv_fwd
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 256, 128, 128, generator=g).to(DEV)
    x2 = torch.randn(1, 256, 128, 128, generator=g).to(DEV)
    w = (torch.randn(256, 256, 3, 3, generator=g) / 48).to(DEV)
    b = torch.randn(256, generator=g).to(DEV)
    y = _conv_fwd(x, w, b, (1, 1), (1, 1))
    yr = _conv_fwd(x, w, b, (1, 1), (1, 1), relu=True)
    assert torch.equal(yr, torch.relu(y))
    y2 = _conv_fwd(x2, w, None, (1, 1), (1, 1))
    y12 = _conv_fwd(x + x2, w, b, (1, 1), (1, 1))
    assert torch.allclose(y12, y + y2, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("relu,res", [(True, False), (False, False), (True, True)])
def test_fused_conv_bn_act_matches_separate_ops(relu, res):
    """conv_bn_act (one launch: conv + eval-BN + shortcut + ReLU, fused backward) vs torch modules."""
    from feature_intertwiner_amd.conv import Conv2d, conv_bn_act
    torch.manual_seed(3)
    conv = Conv2d(32, 48, 3, stride=1, padding=1).to(DEV)
    bn = torch.nn.BatchNorm2d(48, eps=0.001).to(DEV).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 2.0)
    rconv = torch.nn.Conv2d(32, 48, 3, stride=1, padding=1).double()
    rbn = torch.nn.BatchNorm2d(48, eps=0.001).double().eval()
    rconv.load_state_dict({k: v.cpu().double() for k, v in conv.state_dict().items()})
    rbn.load_state_dict({k: v.cpu().double() if v.dtype.is_floating_point else v.cpu() for k, v in bn.state_dict().items()})
    x = torch.randn(3, 32, 12, 10)
    r = torch.randn(3, 48, 12, 10)
    xg = x.to(DEV).requires_grad_(True)
    rg = r.to(DEV).requires_grad_(True) if res else None
    y = conv_bn_act(xg, conv, bn, relu=relu, residual=rg)
    xd = x.double().requires_grad_(True)
    rd = r.double().requires_grad_(True) if res else None
    yd = rbn(rconv(xd))
    if res:
        yd = yd + rd
    if relu:
        yd = torch.relu(yd)
    assert torch.allclose(y.detach().cpu().double(), yd.detach(), rtol=1e-4, atol=1e-4)
    gy = torch.randn(yd.shape)
    y.backward(gy.to(DEV))
    yd.backward(gy.double())
    pairs = [(xg.grad, xd.grad), (conv.weight.grad, rconv.weight.grad), (conv.bias.grad, rconv.bias.grad),
             (bn.weight.grad, rbn.weight.grad), (bn.bias.grad, rbn.bias.grad)]
    if res:
        pairs.append((rg.grad, rd.grad))
    for a, b in pairs:
        assert torch.allclose(a.cpu().double(), b, rtol=2e-4, atol=2e-4), (a.shape,)


@pytest.mark.parametrize("shape", [(2, 32, 24, 20, 64), (3, 16, 9, 7, 36), (1, 64, 16, 16, 256)])
def test_channels_last_output_matches_nchw(shape):
    """conv + eval-BN + ReLU written channels-last by the epilogue (output_layout = 1) and its
    backward through fi_bn_act_backward(layout = 1): values identical to the NCHW path, gradients
    equal up to the summation order of the per-channel reductions."""
    from feature_intertwiner_amd.conv import Conv2d, conv_bn_act
    N, Cin, H, W, Cout = shape
    torch.manual_seed(sum(shape))
    conv = Conv2d(Cin, Cout, 3, padding=1).to(DEV)
    bn = torch.nn.BatchNorm2d(Cout).to(DEV).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.2)
        bn.running_var.uniform_(0.5, 2.0)
        bn.weight.normal_(1, 0.2)
        bn.bias.normal_(0, 0.2)
    x = torch.randn(N, Cin, H, W, device=DEV)
    g = torch.randn(N, Cout, H, W, device=DEV)
    res = {}
    for cl in (False, True):
        xi = x.clone().requires_grad_(True)
        for p in list(conv.parameters()) + list(bn.parameters()):
            p.grad = None
        y = conv_bn_act(xi, conv, bn, relu=True, channels_last_out=cl)
        assert y.is_contiguous(memory_format=torch.channels_last) == cl or (H * W == 1)
        y.backward(g.contiguous(memory_format=torch.channels_last) if cl else g)
        res[cl] = (y.detach().contiguous(), xi.grad, conv.weight.grad.clone(), conv.bias.grad.clone(),
                   bn.weight.grad.clone(), bn.bias.grad.clone())
    assert torch.equal(res[False][0].view(torch.int32), res[True][0].view(torch.int32))
    for a, b in zip(res[False][1:], res[True][1:]):
        assert (a - b).abs().max().item() <= 2e-5 * (a.abs().max().item() + 1e-6)


def test_bn_fold_cache_tracks_parameter_versions():
    from feature_intertwiner_amd import conv as C
    torch.manual_seed(3)
    conv = C.Conv2d(16, 32, 3, padding=1).to(DEV)
    bn = torch.nn.BatchNorm2d(32, eps=0.001).to(DEV).eval()
    with torch.no_grad():
        bn.running_var.uniform_(0.5, 2.0)
        bn.running_mean.normal_()
    x = torch.randn(2, 16, 8, 8, device=DEV)
    y0 = C.conv_bn_act(x, conv, bn)                       # registers the pair, folds in-layer
    assert C._cached_fold(conv, bn) is None
    C.refresh_bn_folds()
    sc, sh = C._cached_fold(conv, bn)
    exp_sc = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    assert torch.allclose(sc, exp_sc, rtol=1e-6) and torch.allclose(
        sh, bn.bias + (conv.bias - bn.running_mean) * exp_sc, rtol=1e-5, atol=1e-6)
    assert torch.allclose(C.conv_bn_act(x, conv, bn), y0, rtol=1e-5, atol=1e-5)
    with torch.no_grad():
        bn.weight.mul_(2.0)                               # optimizer-style in-place update -> stale
    assert C._cached_fold(conv, bn) is None
    y1 = C.conv_bn_act(x, conv, bn)
    C.refresh_bn_folds()
    assert torch.allclose(C.conv_bn_act(x, conv, bn), y1, rtol=1e-5, atol=1e-5)


def test_conv_bias_relu_and_unshuffled_deconv_match_torch():
    from feature_intertwiner_amd.conv import ConvTranspose2x2, conv_bias_relu
    torch.manual_seed(5)
    x = torch.randn(3, 32, 9, 11, device=DEV)
    w = torch.randn(48, 32, 3, 3, device=DEV) * 0.1
    b = torch.randn(48, device=DEV)
    g = torch.randn(3, 48, 9, 11, device=DEV)
    res = []
    for fused in (False, True):
        xi, wi, bi = (t.clone().requires_grad_(True) for t in (x, w, b))
        y = conv_bias_relu(xi, wi, bi, (1, 1), (1, 1)) if fused else F.relu(F.conv2d(xi, wi, bi, 1, 1))
        y.backward(g)
        res.append((y.detach(), xi.grad, wi.grad, bi.grad))
    for a, c in zip(*res):
        assert (a - c).abs().max().item() <= 1e-4 * (a.abs().max().item() + 1e-6)
    # deconv: un-shuffled form (+ fused ReLU) vs torch's ConvTranspose2d + ReLU
    ref = torch.nn.ConvTranspose2d(32, 24, 2, stride=2).to(DEV)
    m = ConvTranspose2x2(32, 24, kernel_size=2, stride=2).to(DEV)
    m.load_state_dict(ref.state_dict())
    exp = F.relu(ref(x))
    u = m.forward_unshuffled(x, relu=True)
    assert u.shape == (3, 2, 2, 24, 9, 11)
    got = u.permute(0, 3, 4, 1, 5, 2).reshape(3, 24, 18, 22)
    assert torch.allclose(got, exp, rtol=1e-4, atol=1e-5)
    assert torch.allclose(F.relu(m(x)), exp, rtol=1e-4, atol=1e-5)


def test_derived_state_follows_data_writes():
    """Writes through `.data` (broadcast, checkpoint surgery) bypass tensor version counters: the cached
    eval-BN folds and the per-step W^T copies must be dropped by invalidate_derived_state, after which the
    next pass uses the new values (stale folds after dist.broadcast(t.data))."""
    from feature_intertwiner_amd import conv as C
    from feature_intertwiner_amd.data_parallel import invalidate_derived_state
    torch.manual_seed(3)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = C.Conv2d(32, 48, 3, padding=1)
            self.bn = torch.nn.BatchNorm2d(48)

        def forward(self, x):
            C.prepare_step(self)
            return C.conv_bn_act(x, self.conv, self.bn, relu=True)
    net = Net().to(DEV).eval()
    x = torch.randn(2, 32, 12, 12, device=DEV, requires_grad=True)

    def ref():
        y = F.relu(F.batch_norm(F.conv2d(x, net.conv.weight, net.conv.bias, padding=1), net.bn.running_mean,
                                net.bn.running_var, net.bn.weight, net.bn.bias, False, 0.0, net.bn.eps))
        return y
    y0 = net(x)
    assert torch.allclose(y0, ref(), rtol=1e-4, atol=1e-4)
    net.bn.running_mean.data.fill_(0.37)            # no version bump
    net.conv.weight.data.mul_(1.5)
    invalidate_derived_state(net)
    y1 = net(x)
    assert torch.allclose(y1, ref(), rtol=1e-4, atol=1e-4) and not torch.allclose(y1, y0, atol=1e-3)
    gx, = torch.autograd.grad(y1.sum(), x)          # the data gradient uses the refreshed W^T
    gr, = torch.autograd.grad(ref().sum(), x)
    assert torch.allclose(gx, gr, rtol=1e-3, atol=1e-4)


def test_patch_kernel_fused_epilogue():
    """conv3x3_patch_kernel with the fused scale / shift / shortcut / ReLU epilogue (a bottleneck's conv + eval-BN)."""
    from feature_intertwiner_amd.conv import _conv_fwd
    g = torch.Generator().manual_seed(5)
    N, Cin, H, W, Cout = 8, 32, 64, 64, 256
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b, sc = torch.randn(Cout, generator=g), torch.rand(Cout, generator=g) + 0.5
    res = torch.randn(N, Cout, H, W, generator=g)
    ref = F.relu(F.conv2d(x.double(), w.double(), None, padding=1) * sc.double().view(1, -1, 1, 1) +
                 b.double().view(1, -1, 1, 1) + res.double())
    for kw in (dict(relu=True, scale=sc.to(DEV), residual=res.to(DEV)), dict(relu=False, scale=None, residual=None)):
        y = _conv_fwd(x.to(DEV), w.to(DEV), b.to(DEV), (1, 1), (1, 1), **kw)
        r = ref if kw["relu"] else F.conv2d(x.double(), w.double(), b.double(), padding=1)
        assert (y.cpu().double() - r).abs().max().item() <= 2e-5 * math.sqrt(Cin * 9) * r.abs().max().item()
    # channels-last output of the 2-D patch kernel (the make-up layers of the RoI stage)
    y = _conv_fwd(x.to(DEV), w.to(DEV), b.to(DEV), (1, 1), (1, 1), relu=True, scale=sc.to(DEV), out_channels_last=True)
    assert y.is_contiguous(memory_format=torch.channels_last)
    r = F.relu(F.conv2d(x.double(), w.double(), None, padding=1) * sc.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1))
    assert (y.cpu().double() - r).abs().max().item() <= 2e-5 * math.sqrt(Cin * 9) * r.abs().max().item()
    # flat tiles (14 x 14 maps) with the same epilogue
    N, Cin, H, W, Cout = 400, 16, 14, 14, 200
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b, sc = torch.randn(Cout, generator=g), torch.rand(Cout, generator=g) + 0.5
    res = torch.randn(N, Cout, H, W, generator=g)
    ref = F.relu(F.conv2d(x.double(), w.double(), None, padding=1) * sc.double().view(1, -1, 1, 1) +
                 b.double().view(1, -1, 1, 1) + res.double())
    y = _conv_fwd(x.to(DEV), w.to(DEV), b.to(DEV), (1, 1), (1, 1), relu=True, scale=sc.to(DEV), residual=res.to(DEV))
    assert (y.cpu().double() - ref).abs().max().item() <= 2e-5 * math.sqrt(Cin * 9) * ref.abs().max().item()


@pytest.mark.parametrize("case", [
    (4, 256, 256, 256, 256, 3, 1),      # P2-level 3x3: conv3x3_patch_kernel<false> forward + data gradient, vec weight gradient
    (2006, 256, 14, 14, 256, 3, 1),     # mask-head 3x3 at the step's RoI count: flat patch tiles
    (4, 1024, 64, 64, 256, 1, 0),       # C4 1x1: conv1x1_reg_kernel
    (4, 256, 64, 64, 1024, 1, 0),       # C4 conv3 1x1
])
def test_full_size_adjoint_identities(case):
    """Size-independent properties at the layer sizes of BASELINE configs[2] (no CPU reference at these sizes):
    the data gradient is the adjoint of the forward map in x, the weight gradient its adjoint in w,
        <dy, conv(x, w)> == <dgrad(dy), x> == <wgrad(dy, x), w>,
    and the forward map is linear in x.  Sums in float64 on the device; bar 2e-6 relative to the products' scale."""
    from feature_intertwiner_amd.conv import conv2d
    N, Cin, H, W, Cout, R, pd = case
    g = torch.Generator(device=DEV).manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, device=DEV, generator=g).requires_grad_(True)
    w = (torch.randn(Cout, Cin, R, R, device=DEV, generator=g) / math.sqrt(Cin * R * R))
    w = (w.contiguous(memory_format=torch.channels_last) if R > 1 else w).requires_grad_(True)
    y = conv2d(x, w, None, (1, 1), (pd, pd))
    dy = torch.randn(y.shape, device=DEV, generator=g)
    y.backward(dy)
    dot = lambda a, b: float((a.double() * b.double()).sum())
    lhs = dot(dy, y.detach())
    scale = math.sqrt(float(dy.double().pow(2).sum()) * float(y.detach().double().pow(2).sum()))
    assert abs(lhs - dot(x.grad, x.detach())) <= 2e-6 * scale
    assert abs(lhs - dot(w.grad, w.detach())) <= 2e-6 * scale
    with torch.no_grad():
        x2 = torch.randn(N, Cin, H, W, device=DEV, generator=g)
        ysum = conv2d(x.detach() + x2, w.detach(), None, (1, 1), (pd, pd))
        y2 = conv2d(x2, w.detach(), None, (1, 1), (pd, pd))
        assert float((ysum - y.detach() - y2).abs().max()) <= 2e-5 * math.sqrt(Cin * R * R) * float(ysum.abs().max())


@pytest.mark.parametrize("inplanes,planes,stride,proj,hw", [(64, 64, 1, True, (16, 16)), (256, 128, 2, True, (16, 24)),
                                                            (256, 128, 2, True, (15, 13)), (256, 64, 1, False, (12, 12)),
                                                            (512, 256, 2, True, (28, 28))])
def test_bottleneck_block_matches_torch_reference(inplanes, planes, stride, proj, hw):
    """A whole residual block (lib/sub_module.py:84-128) incl. the gradient hand-offs between its autograd nodes:
    identity shortcut (conv3's shortcut gradient added in conv1's data-gradient epilogue) and projection shortcut
    (the projection's data gradient -- compact for stride 2 -- added in conv1's, one interleave pass).  Reference:
    the same block from stock torch modules in float64."""
    import torch.nn as nn
    from feature_intertwiner_amd.conv import Conv2d
    from feature_intertwiner_amd.sub_module import Bottleneck
    torch.manual_seed(inplanes + planes + stride + hw[0])
    ds = None
    if proj:
        ds = nn.Sequential(Conv2d(inplanes, planes * 4, kernel_size=1, stride=stride), nn.BatchNorm2d(planes * 4, eps=0.001))
    blk = Bottleneck(inplanes, planes, stride, ds)
    for m in blk.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    blk.eval()

    def ref_forward(x, sd):
        c = lambda name, inp, st=1, pd=0: F.conv2d(inp, sd[name + ".weight"], sd[name + ".bias"], stride=st, padding=pd)
        bn = lambda name, inp, eps=0.001: F.batch_norm(inp, sd[name + ".running_mean"], sd[name + ".running_var"],
                                                        sd[name + ".weight"], sd[name + ".bias"], False, 0.0, eps)
        out = F.relu(bn("bn1", c("conv1", x, stride)))
        out = F.relu(bn("bn2", c("conv2", out, 1, 1)))
        out = bn("bn3", c("conv3", out))
        res = bn("downsample.1", c("downsample.0", x, stride)) if proj else x
        return F.relu(out + res)

    x = torch.randn(3, inplanes, *hw)
    pre = torch.nn.Conv2d(inplanes, inplanes, 1)          # makes the block's input a non-leaf with a consumer chain
    sd = {k: v.detach().double().requires_grad_(v.dtype.is_floating_point and "running" not in k)
          for k, v in blk.state_dict().items() if v.dtype.is_floating_point}
    xd = x.double().requires_grad_(True)
    yd = ref_forward(xd * 1.0, sd)
    gy = torch.randn(yd.shape)
    yd.backward(gy.double())

    blk = blk.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    y = blk(xg * 1.0)
    y.backward(gy.to(DEV))
    tol = lambda ref: 3e-4 * (ref.abs().max().item() + 1e-6)
    assert (y.detach().cpu().double() - yd.detach()).abs().max().item() <= tol(yd.detach())
    assert (xg.grad.cpu().double() - xd.grad).abs().max().item() <= tol(xd.grad)
    for k, p in blk.named_parameters():
        assert p.grad is not None, k
        assert (p.grad.cpu().double() - sd[k].grad).abs().max().item() <= tol(sd[k].grad), k


@pytest.mark.parametrize("M,K,N,bias", [(64, 1024, 1024, True), (2048, 12544, 1024, True), (96, 1024, 81, True),
                                       (50, 1024, 324, False), (224, 25088, 1024, True), (7, 256, 130, True)])
def test_linear_on_the_library_kernels_matches_float64(M, K, N, bias):
    """conv.linear = F.linear on the library's own fp32 MFMA kernels (no vendor GEMM): y = x W^T through the
    weight-gradient kernel (split over K), dx and dW through the 1x1 convolution kernel; rows / columns that are not
    multiples of 32 / 128 are padded.  The head FCs of lib/sub_module.py:698-747 and :333."""
    from feature_intertwiner_amd import _lib
    from feature_intertwiner_amd.conv import linear
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) if bias else None
    gy = torch.randn(M, N, generator=g)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    bd = b.double().requires_grad_(True) if bias else None
    yd = F.linear(xd, wd, bd)
    yd.backward(gy.double())
    xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    bg = b.to(DEV).requires_grad_(True) if bias else None
    _lib.prof_reset()
    _lib.prof_enable(True)
    y = linear(xg, wg, bg)
    y.backward(gy.to(DEV))
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    ran = sum(_lib.prof_get(k)[0] for k in _lib.KERNEL_IDS if k.startswith("conv"))
    assert ran >= 3, "forward, dx and dW must run on the library's kernels"
    tol = lambda ref, k: 2e-5 * math.sqrt(k) * (ref.abs().max().item() + 1e-6)
    assert y.shape == (M, N)
    assert (y.detach().cpu().double() - yd.detach()).abs().max().item() <= tol(yd.detach(), K)
    assert (xg.grad.cpu().double() - xd.grad).abs().max().item() <= tol(xd.grad, N)
    assert (wg.grad.cpu().double() - wd.grad).abs().max().item() <= tol(wd.grad, M)
    if bias:
        assert (bg.grad.cpu().double() - bd.grad).abs().max().item() <= tol(bd.grad, M)


@pytest.mark.parametrize("case", [
    # N, Cin, H, W, Cout, k, residual -- which forward kernel serves the (data-gradient shaped) call
    (4, 256, 64, 64, 128, 1, True),      # conv1x1_reg_kernel, full tiles, shortcut + gate
    (4, 256, 64, 64, 128, 1, False),
    (2, 128, 64, 64, 128, 3, False),     # conv3x3_patch_kernel<false> (8 x 16 tiles)
    (2, 128, 64, 64, 128, 3, True),
    (640, 128, 14, 14, 128, 3, False),   # conv3x3_patch_kernel<true> (flat tiles, the RoI maps)
    (3, 48, 13, 11, 40, 3, True),        # conv_fwd_kernel, ragged tile, scalar epilogue
    (2, 64, 32, 32, 64, 1, True),        # conv_fwd_kernel, 16-byte epilogue
    (2, 128, 20, 12, 256, 1, False),     # 1x1, pixel count not a multiple of the tile
])
def test_gated_epilogue_matches_masked_reference(case):
    """fi_conv2d_forward_gated: y = (conv(x, w) [+ residual]) * (gate > 0) in every forward kernel's epilogue."""
    from feature_intertwiner_amd.conv import _conv_fwd
    N, Cin, H, W, Cout, k, res = case
    g = torch.Generator(device="cpu").manual_seed(N + Cin + H + k)
    x = torch.randn(N, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (1.0 / (Cin * k * k) ** 0.5)).to(DEV)
    r = torch.randn(N, Cout, H, W, generator=g).to(DEV) if res else None
    gate = torch.randn(N, Cout, H, W, generator=g).to(DEV)
    gate[gate.abs() < 0.2] = 0.0                                       # exact zeros are closed gates
    wt = w.permute(0, 2, 3, 1).contiguous() if Cin % 16 == 0 else None
    if wt is not None:
        y = _conv_fwd(x, wt, None, (1, 1), (k // 2, k // 2), residual=r, w_tap_major=True, gate=gate, precision="fp32")
        plain = _conv_fwd(x, wt, None, (1, 1), (k // 2, k // 2), residual=r, w_tap_major=True, precision="fp32")
    else:
        y = _conv_fwd(x, w, None, (1, 1), (k // 2, k // 2), residual=r, gate=gate, precision="fp32")
        plain = _conv_fwd(x, w, None, (1, 1), (k // 2, k // 2), residual=r, precision="fp32")
    assert torch.equal(y, plain * (gate > 0))                          # the same arithmetic, then a select
    ref = F.conv2d(x.double(), w.double(), padding=k // 2)
    if res:
        ref = ref + r.double()
    ref = ref * (gate > 0)
    assert float((y.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) * (Cin * k * k) ** 0.5


@pytest.mark.parametrize("cin,cout,k,cl_weight,bias", [(128, 64, 3, False, True), (128, 64, 3, True, False),
                                                      (48, 40, 3, True, True), (64, 256, 1, False, True),
                                                      (20, 24, 3, False, True)])
def test_batchnorm_gradients_from_the_weight_gradient(cin, cout, k, cl_weight, bias):
    """The fp32 backward of conv + eval-BN + ReLU takes d gamma from <W, dW'> (fi_bn_fold_grad) instead of a pass over
    the activations: every gradient against float64 autograd, for both memory orders of W and of dW'."""
    from feature_intertwiner_amd import conv as C
    torch.manual_seed(cin + cout + k)
    conv = C.Conv2d(cin, cout, k, padding=k // 2, bias=bias).to(DEV)
    if cl_weight:
        conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(cout, eps=0.001).to(DEV).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.5); bn.running_var.uniform_(0.5, 2.0)
        bn.weight[3] = 0.0                       # a dead channel: d beta and d gamma must still be exact
    x = torch.randn(4, cin, 16, 12, device=DEV)
    xg = x.clone().requires_grad_(True)
    y = C.conv_bn_act(xg, conv, bn, relu=True)
    gy = torch.randn_like(y)
    y.backward(gy)
    rc = torch.nn.Conv2d(cin, cout, k, padding=k // 2, bias=bias).double()
    rb = torch.nn.BatchNorm2d(cout, eps=0.001).double().eval()
    rc.load_state_dict({n: v.cpu().double() for n, v in conv.state_dict().items()})
    rb.load_state_dict({n: v.cpu().double() if v.dtype.is_floating_point else v.cpu() for n, v in bn.state_dict().items()})
    xd = x.cpu().double().requires_grad_(True)
    yd = torch.relu(rb(rc(xd)))
    yd.backward(gy.cpu().double())
    pairs = [("dx", xg.grad, xd.grad), ("dw", conv.weight.grad, rc.weight.grad), ("dgamma", bn.weight.grad, rb.weight.grad),
             ("dbeta", bn.bias.grad, rb.bias.grad)]
    if bias:
        pairs.append(("dbias", conv.bias.grad, rc.bias.grad))
    for name, a, b in pairs:
        assert float((a.cpu().double() - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-9, name


def test_stage_of_bottlenecks_with_gated_data_gradients():
    """A stage as ResNet.make_layer builds it (projection block + identity blocks in an nn.Sequential): the blocks after
    the first are the only readers of their inputs, so every ReLU mask inside the stage is applied in the consuming
    layer's data-gradient epilogue (conv.Gate) -- values and ALL gradients against the same stage in float64, with and
    without the gradient arena of a model-level prepare_step."""
    import torch.nn as nn
    from feature_intertwiner_amd import conv as C
    from feature_intertwiner_amd.sub_module import ResNet
    torch.manual_seed(5)
    net = ResNet("resnet50")
    net.inplanes = 64
    stage = net.make_layer(net.block, 32, 3, stride=2)
    for m in stage.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    stage.eval()
    assert [b.input_sole for b in stage] == [False, True, True]

    def ref_block(x, sd, p, stride, proj):
        c = lambda name, inp, st=1, pd=0: F.conv2d(inp, sd[p + name + ".weight"], sd[p + name + ".bias"], stride=st, padding=pd)
        bn = lambda name, inp: F.batch_norm(inp, sd[p + name + ".running_mean"], sd[p + name + ".running_var"],
                                            sd[p + name + ".weight"], sd[p + name + ".bias"], False, 0.0, 0.001)
        out = F.relu(bn("bn1", c("conv1", x, stride)))
        out = F.relu(bn("bn2", c("conv2", out, 1, 1)))
        out = bn("bn3", c("conv3", out))
        res = bn("downsample.1", c("downsample.0", x, stride)) if proj else x
        return F.relu(out + res)

    x = torch.randn(2, 64, 32, 48)
    sd = {k: v.detach().double().requires_grad_("running" not in k) for k, v in stage.state_dict().items()
          if v.dtype.is_floating_point}
    xd = x.double().requires_grad_(True)
    yd = xd * 1.0
    for i in range(3):
        yd = ref_block(yd, sd, "%d." % i, 2 if i == 0 else 1, i == 0)
    gy = torch.randn(yd.shape)
    yd.backward(gy.double())
    stage = stage.to(DEV)
    tol = lambda ref: 3e-4 * (ref.abs().max().item() + 1e-6)
    for arena in (False, True):
        for p in stage.parameters():
            p.grad = None
        if arena:
            C.prepare_step(stage)
            C.prepare_step(stage)          # the second call knows the (conv, bn) pairs: scaled W^T from the batched transpose
        xg = x.to(DEV).requires_grad_(True)
        y = stage(xg * 1.0)
        claimed = [getattr(t, "claimed", None) for t in ()]
        y.backward(gy.to(DEV))
        torch.cuda.synchronize()
        assert (y.detach().cpu().double() - yd.detach()).abs().max().item() <= tol(yd.detach())
        assert (xg.grad.cpu().double() - xd.grad).abs().max().item() <= tol(xd.grad), arena
        for k, p in stage.named_parameters():
            assert p.grad is not None, k
            assert (p.grad.cpu().double() - sd[k].grad).abs().max().item() <= tol(sd[k].grad), (k, arena)
    C.invalidate_step_state()


def test_gate_between_conv_bias_relu_and_its_only_reader():
    """conv + bias + ReLU whose output feeds ONE plain convolution (the RPN's shared conv -> stacked heads, the mask
    head's deconv -> conv5 through a view): the reader masks its data gradient, the producer takes the bias gradient
    from its weight-gradient kernel.  Against float64 autograd, single use and a layer applied twice (two pyramid levels)."""
    from feature_intertwiner_amd import conv as C
    torch.manual_seed(9)
    shared = C.Conv2d(32, 64, 3, padding=1).to(DEV)
    head = C.Conv2d(64, 18, 1).to(DEV)
    rs, rh = torch.nn.Conv2d(32, 64, 3, padding=1).double(), torch.nn.Conv2d(64, 18, 1).double()
    rs.load_state_dict({k: v.cpu().double() for k, v in shared.state_dict().items()})
    rh.load_state_dict({k: v.cpu().double() for k, v in head.state_dict().items()})
    xs = [torch.randn(2, 32, 24, 16), torch.randn(2, 32, 12, 8)]
    xg = [x.to(DEV).requires_grad_(True) for x in xs]
    xd = [x.double().requires_grad_(True) for x in xs]
    tot, totd, gates = 0.0, 0.0, []
    for a, b in zip(xg, xd):
        y = C.conv_bias_relu(a, shared.weight, shared.bias, (1, 1), (1, 1))
        gates.append(y._fi_gate)
        o = C.conv2d(y.view(y.shape), head.weight, head.bias, gate_dx=y._fi_gate)       # through a view, like the mask head
        gy = torch.randn(o.shape)
        tot = tot + (o * gy.to(DEV)).sum()
        totd = totd + (rh(torch.relu(rs(b))) * gy.double()).sum()
    assert all(g.claimed for g in gates)
    tot.backward()
    totd.backward()
    pairs = [(xg[0].grad, xd[0].grad), (xg[1].grad, xd[1].grad), (shared.weight.grad, rs.weight.grad),
             (shared.bias.grad, rs.bias.grad), (head.weight.grad, rh.weight.grad), (head.bias.grad, rh.bias.grad)]
    for i, (a, b) in enumerate(pairs):
        assert float((a.cpu().double() - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-9, i


@pytest.mark.parametrize("shape", [(4, 256, 64, 64), (2, 3, 5, 6), (1, 16, 33, 128)])
def test_upsample2x_backward_is_bit_identical_to_the_framework(shape):
    """conv.upsample2x: F.interpolate(scale 2, nearest) forward; backward fi_sum2x2 adds the 2 x 2 source window in the
    order of the framework's kernel -- same bits."""
    from feature_intertwiner_amd.conv import upsample2x
    torch.manual_seed(shape[1])
    x = torch.randn(*shape, device=DEV)
    a, b = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = upsample2x(a), F.interpolate(b, scale_factor=2, mode="nearest")
    assert torch.equal(ya, yb)
    gy = torch.randn_like(ya)
    ya.backward(gy)
    yb.backward(gy)
    assert torch.equal(a.grad, b.grad)


@pytest.mark.parametrize("shape", [(2, 8, 64, 64), (1, 3, 13, 12), (2, 4, 9, 8), (1, 2, 512, 512), (3, 5, 16, 20)])
def test_stem_maxpool_kernels_match_the_framework_bit_for_bit(shape):
    """fi_maxpool3x3s2_*: max_pool2d(3, 2, ceil_mode) without an index tensor.  Inputs with MANY exact ties (a ReLU
    output is mostly zeros; values drawn from a small set) pin the first-maximum rule and the order in which the
    overlapping windows' gradients are added; `masked` = the gradient times (x > 0)."""
    from feature_intertwiner_amd.conv import _MaxPool3x3s2Fn
    g = torch.Generator(device="cpu").manual_seed(shape[2])
    x = torch.relu(torch.randint(-3, 4, shape, generator=g).float() * 0.5).to(DEV)          # ties everywhere
    x2 = torch.relu(torch.randn(shape, generator=g)).to(DEV)
    for inp in (x, x2):
        a, b = inp.clone().requires_grad_(True), inp.clone().requires_grad_(True)
        ya = _MaxPool3x3s2Fn.apply(a, False)
        yb = F.max_pool2d(b, 3, 2, 0, ceil_mode=True)
        assert ya.shape == yb.shape and torch.equal(ya, yb)
        gy = torch.randn(ya.shape, generator=g).to(DEV)
        ya.backward(gy)
        yb.backward(gy)
        assert torch.equal(a.grad, b.grad)
        c = inp.clone().requires_grad_(True)
        _MaxPool3x3s2Fn.apply(c, True).backward(gy)
        assert torch.equal(c.grad, b.grad * (inp > 0))


def test_take_rows_gathers_and_hands_the_second_gradient_over():
    """conv.take_rows: rows of a tensor with a second reader (the Dev stage's 14 x 14 crops: feature extractor + mask
    head).  Values and gradients against plain indexing, with and without the GradBox hand-off."""
    from feature_intertwiner_amd import conv as C
    torch.manual_seed(4)
    x = torch.randn(40, 8, 6, 6, device=DEV)
    order = torch.randperm(40, device=DEV)[:25]
    w = torch.randn(25, 8, 6, 6, device=DEV)
    v = torch.randn(40, 8, 6, 6, device=DEV)
    a = x.clone().requires_grad_(True)
    (a[order] * w).sum().backward()
    ref_alone = a.grad.clone()
    for with_box in (False, True):
        b = x.clone().requires_grad_(True)
        box = C.GradBox() if with_box else None
        rows = C.take_rows(b, order, box)
        assert torch.equal(rows, x[order])
        if with_box:
            assert box.taker
            box.value = v.clone()                 # what the second reader's backward would leave
        (rows * w).sum().backward()
        assert torch.equal(b.grad, ref_alone + v if with_box else ref_alone)


@pytest.mark.parametrize("N,C,K,hw,gated", [(64, 256, 81, 14, True), (37, 48, 5, 7, False), (16, 256, 81, 14, False)])
def test_class_row_conv1x1_matches_the_dense_backward(N, C, K, hw, gated):
    """conv.conv1x1_class_rows: the mask head's conv5 + the loss's class gather.  Forward = the dense 1x1 convolution's
    channel cls[n]; backward (fi_class_row_conv1x1_backward) = what the dense data- and weight-gradient produce for a
    gradient that is zero outside channel cls[n] -- against float64 autograd of the dense formulation; `gated`: x is a
    ReLU output whose mask the layer applies to dx (conv.Gate)."""
    from feature_intertwiner_amd import conv as C_
    torch.manual_seed(N + K)
    pre = torch.randn(N, C, hw, hw, device=DEV)
    w = (torch.randn(K, C, 1, 1, device=DEV) * 0.1).requires_grad_(True)
    b = torch.randn(K, device=DEV).requires_grad_(True)
    cls = torch.randint(0, K, (N,), device=DEV)
    gy = torch.randn(N, hw, hw, device=DEV)
    # library path: x = relu(pre) produced by a fused conv so that it carries a Gate when `gated`
    x_leaf = pre.clone().requires_grad_(True)
    if gated:
        eye = torch.eye(C, device=DEV).view(C, C, 1, 1)
        x = C_.conv_bias_relu(x_leaf, eye, None)                      # relu(x_leaf), with a Gate
        out = C_.conv1x1_class_rows(x, w, b, cls, gate_dx=True)
        assert x._fi_gate.claimed
    else:
        x = x_leaf
        out = C_.conv1x1_class_rows(x, w, b, cls)
    out.backward(gy)
    # float64 dense reference
    xd = pre.double().cpu().requires_grad_(True)
    wd, bd = w.detach().double().cpu().requires_grad_(True), b.detach().double().cpu().requires_grad_(True)
    xin = torch.relu(xd) if gated else xd
    y = F.conv2d(xin, wd, bd)
    ref = y[torch.arange(N), cls.cpu()]
    ref.backward(gy.double().cpu())
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) <= 1e-4 * float(ref.abs().max())
    for name, got, want in (("dx", x_leaf.grad, xd.grad), ("dw", w.grad, wd.grad), ("db", b.grad, bd.grad)):
        assert float((got.cpu().double() - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-9, name


def test_weight_grad_batch_equals_separate_launches():
    """fi_conv2d_weight_grad_batch (n problems of one geometry in ONE launch, operand pointers in the kernel arguments)
    against n calls of fi_conv2d_weight_grad and against float64: the C4-stage shapes of ResNet-101 (1x1 256 -> 1024,
    1x1 1024 -> 256, 3x3 256 -> 256 on 64 x 64 maps), with and without the bias sums, incl. a batch larger than
    FI_WGRAD_BATCH_MAX and a geometry the one-launch path does not take (falls back to a loop)."""
    import ctypes
    from feature_intertwiner_amd import _lib
    L = _lib.load()
    torch.manual_seed(3)
    cases = [(2, 256, 16, 16, 512, 1, 5, True), (2, 512, 16, 16, 128, 1, 3, False), (2, 128, 16, 16, 128, 3, 4, True),
             (1, 128, 8, 8, 128, 3, 26, True), (2, 64, 16, 16, 64, 3, 3, False)]
    for N, Cin, H, W, Cout, R, n, with_db in cases:
        pad = R // 2
        xs = [torch.randn(N, Cin, H, W, device=DEV) for _ in range(n)]
        dys = [torch.randn(N, Cout, H, W, device=DEV) for _ in range(n)]
        lay = 1 if Cin % 128 == 0 else 0
        shape = (Cout, R, R, Cin) if lay else (Cout, Cin, R, R)
        got = [torch.zeros(shape, device=DEV) for _ in range(n)]
        gdb = [torch.zeros(Cout, device=DEV) for _ in range(n)]
        arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        _lib.check(L.fi_conv2d_weight_grad_batch(arr(xs), arr(dys), arr(got), arr(gdb) if with_db else None, n, N, Cin, H, W,
                                                 Cout, R, R, 1, 1, pad, pad, lay, _lib.OUTPUTS_ZEROED,
                                                 _lib.current_stream()), "batch")
        for i in range(n):
            one = torch.zeros(shape, device=DEV)
            odb = torch.zeros(Cout, device=DEV)
            _lib.check(L.fi_conv2d_weight_grad(_lib.ptr(xs[i]), _lib.ptr(dys[i]), _lib.ptr(one), N, Cin, H, W, Cout, R, R,
                                               1, 1, pad, pad, lay, _lib.ptr(odb) if with_db else None,
                                               _lib.OUTPUTS_ZEROED, _lib.current_stream()), "single")
            ref = torch.nn.grad.conv2d_weight(xs[i].double(), (Cout, Cin, R, R), dys[i].double(), padding=pad)
            g = got[i].permute(0, 3, 1, 2) if lay else got[i]
            o = one.permute(0, 3, 1, 2) if lay else one
            bar = 2e-5 * (N * H * W) ** 0.5 * float(ref.abs().max())
            assert float((g.double() - ref).abs().max()) <= bar, (N, Cin, Cout, R, i)
            assert float((g - o).abs().max()) <= bar
            if with_db:
                rdb = dys[i].double().sum((0, 2, 3))
                assert float((gdb[i].double() - rdb).abs().max()) <= 2e-5 * (N * H * W) ** 0.5 * float(rdb.abs().max()) + 1e-4
                assert float((gdb[i] - odb).abs().max()) <= 1e-3


def test_take_rows_with_a_front_block_in_the_box():
    """conv.take_rows whose GradBox holds the gradient of the FIRST rows only (the mask head's graph batch reads the
    14 x 14 crops [: bs * P] as a view, Dev.forward(mask_front)): one fi_rows_combine pass must equal zeros + front block +
    index_add; also with an empty box."""
    from feature_intertwiner_amd import conv as C
    torch.manual_seed(0)
    x = torch.randn(40, 8, 6, 6, device=DEV)
    idx = torch.randperm(40, device=DEV)[:24]
    gy = torch.randn(24, 8, 6, 6, device=DEV)
    front = torch.randn(16, 8, 6, 6, device=DEV)
    for with_front in (True, False):
        xg = x.clone().requires_grad_(True)
        box = C.GradBox()
        y = C.take_rows(xg, idx, box)
        assert torch.equal(y, x[idx])
        if with_front:
            box.value = front.clone()
        y.backward(gy)
        ref = torch.zeros_like(x)
        if with_front:
            ref[:16] = front
        ref.index_add_(0, idx, gy)
        assert torch.equal(xg.grad, ref)
