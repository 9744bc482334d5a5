"""HIP crop_and_resize (RoIAlign) vs the CPU oracle -- through the C ABI and through
the reference-shaped Python operator.  Bar: bin assignment and forward values
BIT-EXACT; backward (fp32 atomics, order not fixed) within a stated tolerance."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import adversarial_boxes, training_rois

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _fwd(image, boxes, ind, ch, cw, extrap=0.0):
    from feature_intertwiner_amd.roi_align.crop_and_resize import CropAndResizeFunction
    out = CropAndResizeFunction(ch, cw, extrap)(
        torch.from_numpy(image).to(DEV), torch.from_numpy(boxes).to(DEV), torch.from_numpy(ind).to(DEV))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("crop", [(7, 7), (14, 14), (28, 28), (1, 1), (5, 3), (1, 9), (64, 64)])
def test_taps_bit_exact(oracle, crop):
    from feature_intertwiner_amd import _lib
    L = _lib.load()
    ch, cw = crop
    rs = np.random.RandomState(3)
    for (H, W) in ((64, 64), (37, 91), (256, 256), (2, 2), (1, 5)):
        boxes = adversarial_boxes(rs, 400, max(H, 9), max(W, 9))
        exp = oracle.crop_taps(boxes, H, W, ch, cw)
        tb = torch.from_numpy(boxes).to(DEV)
        N = boxes.shape[0]
        o = {k: torch.empty((N, ch if k[0] == "y" else cw), device=DEV,
                            dtype=torch.float32 if k.endswith("frac") else torch.int32)
             for k in ("y_valid", "y0", "y1", "y_frac", "x_valid", "x0", "x1", "x_frac")}
        _lib.check(L.fi_crop_and_resize_taps(_lib.ptr(tb), N, H, W, ch, cw, _lib.ptr(o["y_valid"]),
                                             _lib.ptr(o["y0"]), _lib.ptr(o["y1"]), _lib.ptr(o["y_frac"]),
                                             _lib.ptr(o["x_valid"]), _lib.ptr(o["x0"]), _lib.ptr(o["x1"]),
                                             _lib.ptr(o["x_frac"]), _lib.current_stream()), "taps")
        torch.cuda.synchronize()
        for k, v in o.items():
            got = v.cpu().numpy()
            if k.endswith("frac"):
                assert np.array_equal(_bits(got), _bits(exp[k])), (k, H, W)
            else:
                assert np.array_equal(got, exp[k]), (k, H, W)


@pytest.mark.parametrize("crop", [(7, 7), (14, 14), (28, 28), (1, 1), (5, 3), (3, 11)])
@pytest.mark.parametrize("shape", [(2, 8, 64, 64), (3, 5, 37, 91), (1, 70, 16, 16)])
def test_forward_bit_exact_adversarial(oracle, crop, shape):
    rs = np.random.RandomState(11)
    B, C, H, W = shape
    image = rs.standard_normal(shape).astype(np.float32)
    boxes = adversarial_boxes(rs, 203, H, W)
    ind = rs.randint(0, B, boxes.shape[0]).astype(np.int32)
    for extrap in (0.0, -3.5):
        exp = oracle.crop_and_resize_forward(image, boxes, ind, crop[0], crop[1], extrap)
        got = _fwd(image, boxes, ind, crop[0], crop[1], extrap)
        assert got.shape == exp.shape
        assert np.array_equal(_bits(got), _bits(exp))


def test_forward_full_size_bit_exact(oracle):
    """north-star shape: 512 RoIs x 256 ch x 7x7 (and 14x14) from a [2,256,256,256] map."""
    rs = np.random.RandomState(2000)
    image = rs.standard_normal((2, 256, 256, 256)).astype(np.float32)
    rois = training_rois(rs, 2, 256).reshape(-1, 4)
    ind = np.repeat(np.arange(2, dtype=np.int32), 256)
    for crop in (7, 14):
        exp = oracle.crop_and_resize_forward(image, rois, ind, crop, crop, 0.0)
        got = _fwd(image, rois, ind, crop, crop)
        assert np.array_equal(_bits(got), _bits(exp))


def test_identity_crop_property():
    """size-independent property: the box (0,0,1,1) with crop == map size is the identity."""
    rs = np.random.RandomState(5)
    image = rs.standard_normal((3, 40, 64, 48)).astype(np.float32)
    boxes = np.tile(np.array([[0, 0, 1, 1]], np.float32), (3, 1))
    ind = np.arange(3, dtype=np.int32)
    got = _fwd(image, boxes, ind, 64, 48)
    assert np.array_equal(_bits(got), _bits(image))


def test_bad_box_index_gives_zeros_and_status():
    from feature_intertwiner_amd import _lib
    L = _lib.load()
    rs = np.random.RandomState(1)
    image = torch.from_numpy(rs.standard_normal((2, 4, 16, 16)).astype(np.float32)).to(DEV)
    boxes = torch.tensor([[0.1, 0.1, 0.5, 0.5]] * 3, device=DEV)
    ind = torch.tensor([0, 7, -1], device=DEV, dtype=torch.int32)
    crops = torch.full((3, 4, 7, 7), 9.0, device=DEV)
    status = torch.zeros(1, device=DEV, dtype=torch.int32)
    _lib.check(L.fi_crop_and_resize_forward(_lib.ptr(image), _lib.ptr(boxes), _lib.ptr(ind), 3, 2, 4, 16, 16,
                                            7, 7, 0.0, _lib.ptr(crops), _lib.ptr(status),
                                            _lib.current_stream()), "fwd")
    torch.cuda.synchronize()
    assert status.item() == 1
    assert crops[1].abs().max().item() == 0 and crops[2].abs().max().item() == 0
    assert crops[0].abs().max().item() > 0


def test_empty_and_invalid_args():
    from feature_intertwiner_amd import _lib
    from feature_intertwiner_amd.roi_align.crop_and_resize import CropAndResizeFunction
    image = torch.randn(1, 3, 8, 8, device=DEV)
    out = CropAndResizeFunction(7, 7)(image, torch.zeros(0, 4, device=DEV),
                                      torch.zeros(0, dtype=torch.int32, device=DEV))
    assert out.shape == (0, 3, 7, 7)
    with pytest.raises(_lib.FiError):
        CropAndResizeFunction(65, 7)(image, torch.zeros(1, 4, device=DEV),
                                     torch.zeros(1, dtype=torch.int32, device=DEV))
    with pytest.raises(_lib.FiError):   # no CPU fallback
        CropAndResizeFunction(7, 7)(image.cpu(), torch.zeros(1, 4), torch.zeros(1, dtype=torch.int32))


@pytest.mark.parametrize("crop", [(7, 7), (14, 14), (4, 9), (1, 1)])
def test_backward_vs_oracle(oracle, crop):
    from feature_intertwiner_amd.roi_align.crop_and_resize import CropAndResizeFunction
    rs = np.random.RandomState(23)
    shape = (2, 6, 33, 47)
    B, C, H, W = shape
    boxes = adversarial_boxes(rs, 150, H, W)
    ind = rs.randint(0, B, boxes.shape[0]).astype(np.int32)
    grads = rs.standard_normal((boxes.shape[0], C, crop[0], crop[1])).astype(np.float32)
    exp = oracle.crop_and_resize_backward(grads, boxes, ind, shape)
    image = torch.zeros(shape, device=DEV, requires_grad=True)
    out = CropAndResizeFunction(crop[0], crop[1])(image, torch.from_numpy(boxes).to(DEV),
                                                  torch.from_numpy(ind).to(DEV))
    out.backward(torch.from_numpy(grads).to(DEV))
    got = image.grad.cpu().numpy()
    # tolerance: fp32 sums of <= ~150*crop terms in a different order
    scale = np.abs(exp).max() + 1e-6
    assert np.max(np.abs(got - exp)) <= 2e-5 * scale
    # run twice: atomics may reorder, result must stay within the same tolerance
    image.grad = None
    out2 = CropAndResizeFunction(crop[0], crop[1])(image, torch.from_numpy(boxes).to(DEV),
                                                   torch.from_numpy(ind).to(DEV))
    out2.backward(torch.from_numpy(grads).to(DEV))
    assert np.max(np.abs(image.grad.cpu().numpy() - exp)) <= 2e-5 * scale


def test_backward_adjoint_property_full_size():
    """<crop(I), G> == <I, crop^T(G)> at the north-star size (size-independent property)."""
    from feature_intertwiner_amd.roi_align.crop_and_resize import CropAndResizeFunction
    g = torch.Generator(device="cpu").manual_seed(7)
    image = torch.randn(2, 256, 128, 128, generator=g).to(DEV).requires_grad_(True)
    rs = np.random.RandomState(9)
    rois = torch.from_numpy(training_rois(rs, 2, 256).reshape(-1, 4)).to(DEV)
    ind = torch.arange(2, dtype=torch.int32, device=DEV).repeat_interleave(256)
    out = CropAndResizeFunction(7, 7)(image, rois, ind)
    G = torch.randn(out.shape, generator=g).to(DEV)
    lhs = (out.double() * G.double()).sum()
    out.backward(G)
    rhs = (image.detach().double() * image.grad.double()).sum()
    assert abs(lhs.item() - rhs.item()) <= 1e-5 * max(1.0, abs(lhs.item()))


def test_pyramid_matches_per_level_oracle(oracle):
    from feature_intertwiner_amd.roi_align.crop_and_resize import pyramid_crop_and_resize
    rs = np.random.RandomState(31)
    B, C = 2, 16
    maps = [rs.standard_normal((B, C, s, s)).astype(np.float32) for s in (64, 32, 16, 8)]
    N = 300
    boxes = adversarial_boxes(rs, N, 64, 64)
    ind = rs.randint(0, B, N).astype(np.int32)
    level = rs.randint(1, 7, N).astype(np.int32)   # includes out-of-pyramid levels 1 and 6
    tm = [torch.from_numpy(m).to(DEV).requires_grad_(True) for m in maps]
    for crop in (7, 14):
        out = pyramid_crop_and_resize(tm, torch.from_numpy(boxes).to(DEV), torch.from_numpy(ind).to(DEV),
                                      torch.from_numpy(level).to(DEV), crop, crop)
        got = out.detach().cpu().numpy()
        exp = np.zeros_like(got)
        for l in range(2, 6):
            sel = np.nonzero(level == l)[0]
            if len(sel):
                exp[sel] = oracle.crop_and_resize_forward(maps[l - 2], boxes[sel], ind[sel], crop, crop, 0.0)
        assert np.array_equal(_bits(got), _bits(exp))
        # backward: per-level oracle on the selected rows
        G = rs.standard_normal(got.shape).astype(np.float32)
        for t in tm:
            t.grad = None
        out.backward(torch.from_numpy(G).to(DEV))
        for l in range(2, 6):
            sel = np.nonzero(level == l)[0]
            e = oracle.crop_and_resize_backward(G[sel], boxes[sel], ind[sel], maps[l - 2].shape)
            g = tm[l - 2].grad.cpu().numpy()
            assert np.max(np.abs(g - e)) <= 2e-5 * (np.abs(e).max() + 1e-6)


@pytest.mark.parametrize("C,crops", [(16, (7, 14)), (256, (7, 14)), (200, (7, 5)), (64, (1, 12)), (40, (14, 10))])
def test_pyramid_channels_last_matches_per_level_oracle(oracle, C, crops):
    """Maps in torch.channels_last memory format ([B,H,W,C]) take fi_pyramid_crop_*_nhwc: forward
    BIT-EXACT vs the oracle (and hence vs the NCHW kernels), backward within the atomics tolerance -- and EQUAL to the
    oracle in the deterministic tile-owner form (FI_CROP_BWD_CL_TILES=1: every cell's contributions are added in the
    reference's serial order -- box, bin row, bin column, TL TR BL BR -- with the reference's fp32 products);
    gradients returned in channels_last."""
    import os
    from feature_intertwiner_amd.roi_align.crop_and_resize import LAUNCH_LOG, pyramid_crop_and_resize
    import feature_intertwiner_amd.roi_align.crop_and_resize as mod
    rs = np.random.RandomState(32 + C)
    B = 2
    maps = [rs.standard_normal((B, C, s, s)).astype(np.float32) for s in (64, 32, 16, 8)]
    N = 200
    boxes = adversarial_boxes(rs, N, 64, 64)
    ind = rs.randint(0, B, N).astype(np.int32)
    ind[5] = B + 3                                  # bad image index -> zero row
    level = rs.randint(1, 7, N).astype(np.int32)    # includes out-of-pyramid levels 1 and 6
    tm = [torch.from_numpy(m).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True) for m in maps]
    assert not tm[0].is_contiguous()
    for crop in crops:
        mod.LAUNCH_LOG = []
        out = pyramid_crop_and_resize(tm, torch.from_numpy(boxes).to(DEV), torch.from_numpy(ind).to(DEV),
                                      torch.from_numpy(level).to(DEV), crop, crop)
        assert mod.LAUNCH_LOG[-1]["nhwc"] is True
        mod.LAUNCH_LOG = None
        assert out.is_contiguous() and out.shape == (N, C, crop, crop)
        got = out.detach().cpu().numpy()
        exp = np.zeros_like(got)
        good = (ind >= 0) & (ind < B)
        for l in range(2, 6):
            sel = np.nonzero((level == l) & good)[0]
            if len(sel):
                exp[sel] = oracle.crop_and_resize_forward(maps[l - 2], boxes[sel], ind[sel], crop, crop, 0.0)
        assert np.array_equal(_bits(got), _bits(exp))
        G = rs.standard_normal(got.shape).astype(np.float32)
        for tiles in ("0", "1"):
            for t in tm:
                t.grad = None
            os.environ["FI_CROP_BWD_CL_TILES"] = tiles
            try:
                out.backward(torch.from_numpy(G).to(DEV), retain_graph=True)
                torch.cuda.synchronize()
            finally:
                del os.environ["FI_CROP_BWD_CL_TILES"]
            for l in range(2, 6):
                sel = np.nonzero((level == l) & good)[0]
                e = oracle.crop_and_resize_backward(G[sel], boxes[sel], ind[sel], maps[l - 2].shape)
                assert tm[l - 2].grad.is_contiguous(memory_format=torch.channels_last)
                g = tm[l - 2].grad.cpu().numpy()
                if tiles == "1":
                    assert np.array_equal(g, e), (crop, l, float(np.max(np.abs(g - e))))
                else:
                    assert np.max(np.abs(g - e)) <= 2e-5 * (np.abs(e).max() + 1e-6)


def test_channels_last_full_size_equals_nchw_bitwise():
    """North-star shape 512 x 256 x 7 x 7 (and 14 x 14): the two layouts give identical bits."""
    from feature_intertwiner_amd.roi_align.crop_and_resize import pyramid_crop_and_resize
    rs = np.random.RandomState(9)
    B, C = 2, 256
    maps = [torch.randn(B, C, s, s, device=DEV) for s in (128, 64, 32, 16)]
    maps_cl = [m.contiguous(memory_format=torch.channels_last) for m in maps]
    rois = torch.from_numpy(training_rois(rs, B, 256).reshape(-1, 4)).to(DEV)
    ind = torch.arange(B, device=DEV, dtype=torch.int32).repeat_interleave(256)
    from feature_intertwiner_amd.intertwiner import roi_level
    level = roi_level(rois, 512.0 * 512.0)
    for crop in (7, 14):
        a = pyramid_crop_and_resize(maps, rois, ind, level, crop, crop)
        b = pyramid_crop_and_resize(maps_cl, rois, ind, level, crop, crop)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_single_level_operator_dispatches_on_memory_format(oracle):
    """CropAndResizeFunction (the reference-shaped operator, lib/roi_align/crop_and_resize.py:14-54) handed a map in
    torch.channels_last format does not transpose it: it runs the channels-last kernels (no NCHW crop launch is
    timed), the forward is bit-identical to the NCHW path and to the oracle, the gradient comes back
    channels-last."""
    from feature_intertwiner_amd import _lib
    from feature_intertwiner_amd.roi_align.crop_and_resize import CropAndResizeFunction
    rs = np.random.RandomState(77)
    B, C, H, W = 2, 64, 48, 40
    fm = rs.standard_normal((B, C, H, W)).astype(np.float32)
    boxes = adversarial_boxes(rs, 150, H, W)
    ind = rs.randint(0, B, 150).astype(np.int32)
    tb, ti = torch.from_numpy(boxes).to(DEV), torch.from_numpy(ind).to(DEV)
    for crop in (7, 14):
        x_nchw = torch.from_numpy(fm).to(DEV).requires_grad_(True)
        x_cl = torch.from_numpy(fm).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        fn = CropAndResizeFunction(crop, crop)
        a = fn(x_nchw, tb, ti)
        _lib.prof_reset()
        _lib.prof_enable(True)
        b = fn(x_cl, tb, ti)
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        key = {7: "7x7", 14: "14x14"}[crop]
        assert _lib.prof_get("crop_fwd_nhwc_" + key)[0] == 1 and _lib.prof_get("crop_fwd_" + key)[0] == 0
        exp = oracle.crop_and_resize_forward(fm, boxes, ind, crop, crop, 0.0)
        assert np.array_equal(_bits(a.detach().cpu().numpy()), _bits(exp))
        assert np.array_equal(_bits(b.detach().cpu().numpy()), _bits(exp))
        G = torch.from_numpy(rs.standard_normal(exp.shape).astype(np.float32)).to(DEV)
        a.backward(G)
        b.backward(G)
        assert x_cl.grad.is_contiguous(memory_format=torch.channels_last) and x_nchw.grad.is_contiguous()
        e = oracle.crop_and_resize_backward(G.cpu().numpy(), boxes, ind, fm.shape)
        for g in (x_nchw.grad, x_cl.grad):
            assert np.max(np.abs(g.cpu().numpy() - e)) <= 2e-5 * (np.abs(e).max() + 1e-6)


def test_roi_align_module_matches_oracle(oracle):
    from feature_intertwiner_amd.roi_align.roi_align import RoIAlign
    rs = np.random.RandomState(4)
    fm = rs.standard_normal((2, 8, 40, 56)).astype(np.float32)
    xy = rs.uniform(0, 30, (50, 2)).astype(np.float32)
    wh = rs.uniform(1, 25, (50, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)   # x1,y1,x2,y2 pixels
    ind = rs.randint(0, 2, 50).astype(np.int32)
    for tf in (True, False):
        nb = oracle.roi_align_boxes(boxes, 40, 56, 7, 7, tf)
        exp = oracle.crop_and_resize_forward(fm, nb, ind, 7, 7, 0.0)
        got = RoIAlign(7, 7, 0, tf)(torch.from_numpy(fm).to(DEV), torch.from_numpy(boxes).to(DEV),
                                    torch.from_numpy(ind).to(DEV)).cpu().numpy()
        # the box transform runs in torch fp32 on the GPU: division may differ by 1 ulp from
        # numpy, so compare values with a tolerance here (bin assignment itself is tested above)
        assert np.allclose(got, exp, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("channels_last", [False, True])
def test_crop_grad_group_shares_buffers_within_one_backward_pass_only(channels_last):
    """Two pyramid crops of the same maps in one CropGradGroup (the Dev stage's 7x7 and 14x14 crops,
    lib/sub_module.py:549-577): their map gradients accumulate in ONE set of buffers -- equal to two independent crops --
    and a SECOND backward pass through the same graph starts with fresh buffers (the gradients double; they are neither
    dropped nor added into the first pass's tensors)."""
    from feature_intertwiner_amd.roi_align.crop_and_resize import CropGradGroup, pyramid_crop_and_resize
    rs = np.random.RandomState(5)
    B, C, N = 2, 64, 96
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    maps = [torch.from_numpy(rs.standard_normal((B, C, s, s)).astype(np.float32)).to(DEV).contiguous(memory_format=fmt)
            for s in (64, 32, 16, 8)]
    boxes = torch.from_numpy(adversarial_boxes(rs, N, 64, 64)).to(DEV)
    ind = torch.from_numpy(rs.randint(0, B, N).astype(np.int32)).to(DEV)
    level = torch.from_numpy(rs.randint(2, 6, N).astype(np.int32)).to(DEV)
    g7 = torch.from_numpy(rs.standard_normal((N, C, 7, 7)).astype(np.float32)).to(DEV)
    g14 = torch.from_numpy(rs.standard_normal((N, C, 14, 14)).astype(np.float32)).to(DEV)

    def run(shared, passes):
        tm = [m.clone().requires_grad_(True) for m in maps]
        group = CropGradGroup() if shared else None
        a = pyramid_crop_and_resize(tm, boxes, ind, level, 7, 7, grad_group=group)
        b = pyramid_crop_and_resize(tm, boxes, ind, level, 14, 14, grad_group=CropGradGroup() if not shared else group)
        loss = (a * g7).sum() + (b * g14).sum()
        for _ in range(passes):
            loss.backward(retain_graph=True)
        torch.cuda.synchronize()
        return [t.grad.clone() for t in tm]

    ref = run(False, 1)
    one = run(True, 1)
    two = run(True, 2)
    for r, a, b in zip(ref, one, two):
        tol = 2e-5 * (r.abs().max().item() + 1e-6)           # fp32 atomics: the order of the additions differs
        assert (a - r).abs().max().item() <= tol
        assert (b - 2 * r).abs().max().item() <= 2 * tol


@pytest.mark.parametrize("crop", [(28, 28), (7, 7), (14, 14), (5, 3), (1, 1)])
def test_single_channel_maps_bit_exact(oracle, crop):
    """depth 1 (the mask-target crop, lib/layers.py:301-322) takes crop_fwd_c1_kernel (a thread per bin): bit-identical to the
    oracle on adversarial boxes, both extrapolation values, bad box indices flagged and zero-filled."""
    from feature_intertwiner_amd import _lib
    rs = np.random.RandomState(77)
    for (B, H, W) in ((5, 56, 56), (3, 37, 91), (2, 2, 2)):
        image = rs.standard_normal((B, 1, H, W)).astype(np.float32)
        boxes = adversarial_boxes(rs, 211, max(H, 9), max(W, 9))
        ind = rs.randint(0, B, boxes.shape[0]).astype(np.int32)
        for extrap in (0.0, 2.5):
            exp = oracle.crop_and_resize_forward(image, boxes, ind, crop[0], crop[1], extrap)
            got = _fwd(image, boxes, ind, crop[0], crop[1], extrap)
            assert np.array_equal(_bits(got), _bits(exp)), (crop, H, W, extrap)
    L = _lib.load()
    image = torch.from_numpy(rs.standard_normal((2, 1, 16, 16)).astype(np.float32)).to(DEV)
    boxes = torch.tensor([[0.1, 0.1, 0.5, 0.5]] * 3, device=DEV)
    ind = torch.tensor([0, 7, -1], device=DEV, dtype=torch.int32)
    crops = torch.full((3, 1, 28, 28), 9.0, device=DEV)
    status = torch.zeros(1, device=DEV, dtype=torch.int32)
    _lib.check(L.fi_crop_and_resize_forward(_lib.ptr(image), _lib.ptr(boxes), _lib.ptr(ind), 3, 2, 1, 16, 16, 28, 28, 0.0,
                                            _lib.ptr(crops), _lib.ptr(status), _lib.current_stream()), "fwd")
    torch.cuda.synchronize()
    assert status.item() == 1 and crops[1].abs().max().item() == 0 and crops[2].abs().max().item() == 0
    assert crops[0].abs().max().item() > 0
