"""The persistent 1x1 / stride-1 kernel (csrc/conv1x1_ring.hip, weight_layout 3) against the round-1..5 kernel
(conv1x1_reg_kernel, weight_layout 1) through the C ABI: the two accumulate every output element in the same order (channel
pairs ascending through v_mfma_f32_32x32x2_f32), so the results must be BIT-IDENTICAL for every epilogue operand set; and
against float64 for an independent check.  Shapes: the C3 / C4 / C5 bottleneck layers of ResNet-101 (lib/sub_module.py:90-128),
a RoI-map layer whose 128-pixel tiles straddle images, a partial last tile, one tile per workgroup and many."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fragment_major(w2d, scale_rows=None, transpose=False):
    """What fi_weight_transpose_batch writes with flag 1 (+2): D [M][K] per block of 32 rows x 16 columns as
    [2 halves][64 lanes = (k / 8, row)][4 k]."""
    D = w2d.double()
    if transpose:
        if scale_rows is not None:
            D = D * scale_rows.double()[:, None]
        D = D.t()
    D = D.float().contiguous()
    M, K = D.shape
    blk = D.view(M // 32, 32, K // 16, 2, 2, 4)               # mb, row, kg, khalf, half, k4
    return blk.permute(0, 2, 4, 3, 1, 5).contiguous().view(-1)  # mb, kg, half, khalf, row, k4


def _run(L, _lib, x, w, layout, bias, scale, res, gate, relu, N, Cin, H, W, Cout):
    y = torch.empty(N, Cout, H, W, device=DEV)
    _lib.check(L.fi_conv2d_forward_gated(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(scale), _lib.ptr(res), _lib.ptr(gate),
                                         _lib.ptr(y), N, Cin, H, W, Cout, 1, 1, 1, 1, 0, 0, 1 if relu else 0, layout, 0, 0, 0,
                                         _lib.current_stream()), "conv")
    torch.cuda.synchronize()
    return y


SHAPES = [(4, 256, 64, 64, 1024), (4, 1024, 64, 64, 256), (4, 128, 128, 128, 512), (4, 512, 32, 32, 2048),
          (300, 256, 14, 14, 256), (37, 128, 14, 14, 128 * 9), (4, 160, 90, 92, 128)]


@pytest.mark.parametrize("shape", SHAPES)
def test_ring_kernel_equals_reg_kernel_bitwise(shape):
    from feature_intertwiner_amd import _lib
    L = _lib.load()
    N, Cin, H, W, Cout = shape
    g = torch.Generator(device="cpu").manual_seed(sum(shape))
    x = torch.randn(N, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, generator=g) * 0.05).to(DEV)
    wF = _fragment_major(w).to(DEV)
    bias, scale = torch.randn(Cout, generator=g).to(DEV), (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    res, gate = torch.randn(N, Cout, H, W, generator=g).to(DEV), torch.randn(N, Cout, H, W, generator=g).to(DEV)
    assert L.fi_conv1x1_ring_eligible(N, Cin, H, W, Cout, 1, 1, 1, 1, 0, 0, 0, _lib.ptr(x), _lib.ptr(res), _lib.ptr(res),
                                      _lib.ptr(gate)) == 1
    combos = [(bias, scale, res, None, True), (None, None, res, gate, False), (None, None, None, gate, False),
              (bias, None, None, None, False), (None, None, None, None, False), (bias, scale, None, None, True)]
    for b, s, r, gt, relu in combos:
        ref = _run(L, _lib, x, w, 1, b, s, r, gt, relu, N, Cin, H, W, Cout)
        got = _run(L, _lib, x, wF, 3, b, s, r, gt, relu, N, Cin, H, W, Cout)
        assert torch.equal(got.view(torch.int32), ref.view(torch.int32)), (shape, b is not None, r is not None, gt is not None)
    # ... and against float64 (the plain case)
    y = _run(L, _lib, x, wF, 3, None, None, None, None, False, N, Cin, H, W, Cout)
    exp = torch.einsum("nchw,kc->nkhw", x[:2].double(), w.double())
    assert float((y[:2].double() - exp).abs().max()) <= 2e-5 * np.sqrt(Cin) * float(exp.abs().max())


def test_ineligible_calls_are_refused_not_miscomputed():
    from feature_intertwiner_amd import _lib
    L = _lib.load()
    x = torch.randn(2, 64, 16, 16, device=DEV)
    w = torch.randn(64 * 64, device=DEV)
    y = torch.empty(2, 64, 16, 16, device=DEV)
    rc = L.fi_conv2d_forward_gated(_lib.ptr(x), _lib.ptr(w), None, None, None, None, _lib.ptr(y), 2, 64, 16, 16, 64, 1, 1, 1, 1, 0, 0,
                                   0, 3, 0, 0, 0, _lib.current_stream())
    assert rc != 0


def test_fragment_major_weights_from_the_batched_transpose():
    """fi_weight_transpose_batch flag 1 (W^T * row_scale, the data gradient's operand) and flag 1 + 2 (W itself)."""
    import numpy as np
    from feature_intertwiner_amd import _lib
    from feature_intertwiner_amd.conv import _TR_DESC
    L = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(3)
    co, ci = 256, 384
    w = torch.randn(co, ci, generator=g).to(DEV)
    rs = (torch.rand(co, generator=g) + 0.5).to(DEV)
    d_fwd, d_bwd, d_plain = (torch.empty(co * ci, device=DEV) for _ in range(3))
    wt = torch.empty(ci, co, device=DEV)
    tiles = (co // 32) * (ci // 32)
    desc = np.zeros(3, dtype=_TR_DESC)
    desc[0] = (w.data_ptr(), d_fwd.data_ptr(), co, ci, 1, 3, 0, 0)
    desc[1] = (w.data_ptr(), d_bwd.data_ptr(), co, ci, 1, 1, tiles, rs.data_ptr())
    desc[2] = (w.data_ptr(), wt.data_ptr(), co, ci, 1, 0, 2 * tiles, rs.data_ptr())
    table = torch.from_numpy(desc.view(np.uint8).copy()).to(DEV)
    _lib.check(L.fi_weight_transpose_batch(_lib.ptr(table), 3, 3 * tiles, _lib.current_stream()), "transpose")
    torch.cuda.synchronize()
    assert torch.equal(d_fwd, _fragment_major(w).to(DEV))
    assert torch.equal(wt, (w * rs[:, None]).t().contiguous())
    assert torch.equal(d_bwd, _fragment_major(w, rs, transpose=True).to(DEV))


def test_model_layers_take_the_ring_kernel_and_match_the_reg_kernel(monkeypatch):
    """conv.prepare_step makes the fragment-major copies; a bottleneck-shaped conv + BN + ReLU forward / backward through the
    Python layer gives the same tensors with FI_NO_RING1X1-equivalent routing (the copies dropped)."""
    import torch.nn as nn
    from feature_intertwiner_amd import conv as C
    torch.manual_seed(5)

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = C.Conv2d(512, 128, kernel_size=1)
            self.bn1 = nn.BatchNorm2d(128)
            self.conv3 = C.Conv2d(128, 512, kernel_size=1)
            self.bn3 = nn.BatchNorm2d(512)

        def forward(self, x):
            h = C.conv_bn_act(x, self.conv1, self.bn1, relu=True)
            return C.conv_bn_act(h, self.conv3, self.bn3, relu=True, residual=x)

    net = Block().to(DEV).eval()
    for bn in (net.bn1, net.bn3):
        bn.running_var.uniform_(0.5, 1.5)
        bn.running_mean.normal_(0, 0.1)
    x = torch.randn(4, 512, 64, 64, device=DEV)

    def step():
        xx = x.clone().requires_grad_(True)
        net.zero_grad(set_to_none=True)
        C.prepare_step(net)
        return xx

    xx = step()                     # warm-up: the (conv, BatchNorm) pairs are known from the first forward pass on
    net(xx).sum().backward()
    outs = []
    for ring in (True, False):
        C.invalidate_step_state()
        xx = step()
        if ring:
            assert len(C._WF) == 4          # two layers x (forward, data gradient)
        else:
            C._WF.clear()
        y = net(xx)
        (y * torch.linspace(0.5, 1.5, y.numel(), device=DEV).view_as(y)).sum().backward()
        torch.cuda.synchronize()
        outs.append((y.detach().clone(), xx.grad.clone(), net.conv1.weight.grad.clone(), net.conv3.weight.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0])          # forward: bit-identical
    assert torch.equal(outs[0][1], outs[1][1])          # data gradient: bit-identical
    for a, b in zip(outs[0][2:], outs[1][2:]):          # weight gradients: the same kernel, fp32 atomics (order not fixed)
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))


def test_a_collected_models_fragment_copies_are_not_used_for_the_next_model():
    """The caches are keyed by address and the plan is weakly keyed by the model: when a model is garbage-collected its
    entries stay, and the allocator hands the freed addresses to the next model's weights (same shape, same version
    counter).  Seen as a 9 % gradient error in the two-rank `full` test whenever a smaller detector had run before it in
    the same process.  Entries carry a weak reference to the tensor that owns their address and are void without it."""
    import gc
    import torch.nn as nn
    from feature_intertwiner_amd import conv as C

    class One(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = C.Conv2d(256, 256, kernel_size=1)

    C.invalidate_step_state()
    x = torch.randn(4, 256, 64, 64, device=DEV)
    a = One().to(DEV)
    C.prepare_step(a)
    torch.cuda.synchronize()
    key = a.conv.weight.data_ptr()
    assert key in C._WF
    del a
    gc.collect()
    # a tensor of the same shape at the freed address, not a parameter of any planned model
    w = None
    for _ in range(64):
        t = torch.randn(256, 256, 1, 1, device=DEV)
        if t.data_ptr() == key:
            w = t
            break
    if w is None:
        pytest.skip("the allocator did not hand the freed block back")
    assert C._WF[key][3]() is None                      # the owner is gone
    while w._version < C._WF[key][1]:
        w.add_(0)                                       # (a parameter's counter after its initialisation)
    assert C._WF[key][1] == w._version                  # version / shape alone would have matched
    y = C._conv_fwd(x, w, None, (1, 1), (0, 0))
    ref = torch.nn.functional.conv2d(x.double(), w.double()).float()
    assert torch.allclose(y, ref, rtol=1e-4, atol=1e-3)
    b = One().to(DEV)
    C.prepare_step(b)                                   # a new plan sweeps the dead entries
    assert all(e[3]() is not None for e in C._WF.values())
    C.invalidate_step_state()
