"""HIP RoIPool vs the CPU oracle (which follows the reference CUDA kernel): outputs and
argmax indices EXACT; backward within a stated tolerance (atomics)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rois(rs, n, B, size, malformed=True):
    xy = rs.uniform(-20, size * 0.9, (n, 2))
    wh = np.exp(rs.uniform(np.log(2), np.log(size), (n, 2)))
    r = np.concatenate([rs.randint(0, B, (n, 1)).astype(np.float64), xy, xy + wh], 1)
    if malformed and n >= 8:
        r[3, 3:] = r[3, 1:3] - 30          # end < start: forced to 1x1 by the forward pass
        r[5, 1:] = [size + 50, size + 50, size + 90, size + 90]   # fully outside -> empty bins
        r[7, 1:] = [10.5, 20.5, 10.5, 20.5]                      # single (rounded) pixel
    return r.astype(np.float32)


@pytest.mark.parametrize("pool", [(7, 7), (14, 14), (3, 5), (1, 1)])
@pytest.mark.parametrize("scale", [1.0 / 4, 1.0 / 16, 1.0])
def test_forward_exact(oracle, pool, scale):
    from feature_intertwiner_amd.roi_pooling.functions.roi_pool import RoIPoolFunction
    rs = np.random.RandomState(17)
    B, C, H, W = 2, 9, 48, 40
    feats = rs.standard_normal((B, C, H, W)).astype(np.float32)
    feats[0, 0, 5:9, 5:9] = 2.5    # ties: first maximum in (h, w) order must win
    rois = _rois(rs, 60, B, max(H, W) / scale)
    exp_out, exp_arg = oracle.roi_pool_forward(feats, rois, pool[0], pool[1], scale)
    fn = RoIPoolFunction(pool[0], pool[1], scale)
    got = fn(torch.from_numpy(feats).to(DEV), torch.from_numpy(rois).to(DEV))
    assert np.array_equal(got.cpu().numpy().view(np.uint32), exp_out.view(np.uint32))
    assert fn.argmax.dtype == torch.int32
    assert np.array_equal(fn.argmax.cpu().numpy(), exp_arg)


@pytest.mark.parametrize("pool", [(7, 7), (14, 14), (28, 28), (30, 30)])
def test_forward_exact_wide_windows_and_ties(oracle, pool):
    """Windows wider than one 64-column slab, RoIs larger than the map, RoIs shorter than the pooled
    height (bin_h < 1), and a map quantised to 5 values so that almost every bin has tied maxima:
    the first maximum in (h, w) scan order must win everywhere.  (30x30 takes the simple kernel.)"""
    from feature_intertwiner_amd.roi_pooling.functions.roi_pool import RoIPoolFunction
    rs = np.random.RandomState(23)
    B, C, H, W = 2, 11, 90, 200
    feats = np.round(rs.standard_normal((B, C, H, W)) * 1.5).clip(-2, 2).astype(np.float32)
    feats[1, 3] = -0.0
    feats[1, 3, ::7, ::5] = 0.0                                    # signed zeros compare equal
    n = 120
    xy = rs.uniform(-30, 180, (n, 2))
    wh = np.exp(rs.uniform(np.log(1), np.log(260), (n, 2)))
    rois = np.concatenate([rs.randint(0, B, (n, 1)).astype(np.float64), xy, xy + wh], 1).astype(np.float32)
    rois[0, 1:] = [-50, -50, 400, 300]                             # covers the whole map and more
    rois[1, 1:] = [0, 0, 199, 89]                                  # exactly the map
    rois[2, 1:] = [10, 85, 190, 88]                                # 4 rows high: bin_h < 1 for every pool here
    rois[3, 1:] = [150, -3, 260, 2]                                # clipped on two sides
    exp_out, exp_arg = oracle.roi_pool_forward(feats, rois, pool[0], pool[1], 1.0)
    fn = RoIPoolFunction(pool[0], pool[1], 1.0)
    got = fn(torch.from_numpy(feats).to(DEV), torch.from_numpy(rois).to(DEV))
    assert np.array_equal(fn.argmax.cpu().numpy(), exp_arg)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), exp_out.view(np.uint32))


def test_known_answers():
    """hand-computed cases (the reference ships no RoIPool tests -- parity unpinned there)."""
    from feature_intertwiner_amd.roi_pooling.functions.roi_pool import RoIPoolFunction
    f = torch.arange(36, dtype=torch.float32, device=DEV).view(1, 1, 6, 6)
    # RoI (x1,y1,x2,y2) = (0,0,5,5), scale 1, 2x2 pooling: bins are 3x3 -> maxima at their corners
    fn = RoIPoolFunction(2, 2, 1.0)
    out = fn(f, torch.tensor([[0, 0, 0, 5, 5]], dtype=torch.float32, device=DEV))
    assert out.view(-1).tolist() == [14.0, 17.0, 32.0, 35.0]
    assert fn.argmax.view(-1).tolist() == [14, 17, 32, 35]
    # 1x1 RoI at (x=2, y=3): every pooled cell sees the same pixel 3*6+2 = 20
    out = fn(f, torch.tensor([[0, 2, 3, 2, 3]], dtype=torch.float32, device=DEV))
    assert out.view(-1).tolist() == [20.0] * 4
    # RoI fully outside: empty bins -> 0 and argmax -1
    out = fn(f, torch.tensor([[0, 50, 50, 60, 60]], dtype=torch.float32, device=DEV))
    assert out.view(-1).tolist() == [0.0] * 4 and fn.argmax.view(-1).tolist() == [-1] * 4
    # round half away from zero: 2.5 * 1 -> 3
    out = RoIPoolFunction(1, 1, 1.0)(f, torch.tensor([[0, 2.5, 2.5, 2.5, 2.5]], dtype=torch.float32, device=DEV))
    assert out.item() == 21.0


@pytest.mark.parametrize("pool", [(7, 7), (2, 3)])
def test_backward_vs_oracle(oracle, pool):
    from feature_intertwiner_amd.roi_pooling.functions.roi_pool import RoIPoolFunction
    rs = np.random.RandomState(29)
    B, C, H, W = 2, 4, 24, 20
    scale = 0.25
    feats = rs.standard_normal((B, C, H, W)).astype(np.float32)
    rois = _rois(rs, 40, B, max(H, W) / scale)
    _, arg = oracle.roi_pool_forward(feats, rois, pool[0], pool[1], scale)
    top = rs.standard_normal((40, C, pool[0], pool[1])).astype(np.float32)
    exp = oracle.roi_pool_backward(top, arg, rois, feats.shape, scale)
    x = torch.from_numpy(feats).to(DEV).requires_grad_(True)
    out = RoIPoolFunction(pool[0], pool[1], scale)(x, torch.from_numpy(rois).to(DEV))
    out.backward(torch.from_numpy(top).to(DEV))
    got = x.grad.cpu().numpy()
    assert np.max(np.abs(got - exp)) <= 2e-5 * (np.abs(exp).max() + 1e-6)
    # the malformed RoI (row 3) must not receive gradient (reference in_roi test)
    only3 = np.zeros_like(top)
    only3[3] = 1.0
    x.grad = None
    out = RoIPoolFunction(pool[0], pool[1], scale)(x, torch.from_numpy(rois).to(DEV))
    out.backward(torch.from_numpy(only3).to(DEV))
    assert np.array_equal(x.grad.cpu().numpy(), oracle.roi_pool_backward(only3, arg, rois, feats.shape, scale))


def test_module_and_full_size_property():
    """512 RoIs x 256 ch x 7x7 on a 2x256x128x128 map: every output equals the feature at
    its argmax, and the argmax lies on the RoI's own image."""
    from feature_intertwiner_amd.roi_pooling.modules.roi_pool import _RoIPooling
    from feature_intertwiner_amd.roi_pooling.functions.roi_pool import RoIPoolFunction
    rs = np.random.RandomState(41)
    B, C, H, W = 2, 256, 128, 128
    feats = torch.from_numpy(rs.standard_normal((B, C, H, W)).astype(np.float32)).to(DEV)
    rois = torch.from_numpy(_rois(rs, 512, B, 1024, malformed=False)).to(DEV)
    fn = RoIPoolFunction(7, 7, 1.0 / 8)
    out = fn(feats, rois)
    arg = fn.argmax.long()
    valid = arg >= 0
    assert torch.equal(out[valid], feats.view(-1)[arg[valid]])
    assert torch.all(out[~valid] == 0)
    img = (arg // (C * H * W))
    assert torch.all(img[valid] == rois[:, 0].long().view(-1, 1, 1, 1).expand_as(arg)[valid])
    out2 = _RoIPooling(7, 7, 1.0 / 8)(feats, rois)
    assert torch.equal(out, out2)
