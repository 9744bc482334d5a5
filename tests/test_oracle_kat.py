"""CPU: hand-derived known-answer tests and independent cross-checks for the oracle's
crop_and_resize / NMS / RoIPool restatements.  The reference ships no tests or golden
vectors for these and its C sources cannot be built in this image (they need the
PyTorch-0.3 <TH/TH.h>), so this -- not reference execution -- is what pins them
("parity unpinned" in oracle/fi_oracle.c and DESIGN.md)."""
import numpy as np
import torch
import torch.nn.functional as F

from helpers import adversarial_boxes, clustered_dets


# ---------------------------------------------------------------- crop_and_resize
def test_crop_known_answers(oracle):
    img = np.arange(16, dtype=np.float32).reshape(1, 1, 4, 4)
    # whole image, 4x4 crop: identity (sample grid == pixel grid)
    out = oracle.crop_and_resize_forward(img, np.array([[0, 0, 1, 1]], np.float32), np.array([0], np.int32), 4, 4)
    assert np.array_equal(out[0, 0], img[0, 0])
    # whole image, 2x2 crop: corners (corner-aligned sampling, no averaging)
    out = oracle.crop_and_resize_forward(img, np.array([[0, 0, 1, 1]], np.float32), np.array([0], np.int32), 2, 2)
    assert out[0, 0].tolist() == [[0.0, 3.0], [12.0, 15.0]]
    # 1x1 crop: the box centre -> (1.5, 1.5) -> mean of pixels 5,6,9,10 = 7.5
    out = oracle.crop_and_resize_forward(img, np.array([[0, 0, 1, 1]], np.float32), np.array([0], np.int32), 1, 1)
    assert out.item() == 7.5
    # 3x3 crop of rows/cols [0,2]: exact grid points 0,1,2
    b = np.array([[0, 0, 2 / 3, 2 / 3]], np.float32)
    out = oracle.crop_and_resize_forward(img, b, np.array([0], np.int32), 3, 3)
    assert np.allclose(out[0, 0], img[0, 0, :3, :3], atol=1e-5)
    # half-pixel sample: box y in [0.5/3, 0.5/3] x [0,1] -> rows blend 0.5/0.5
    b = np.array([[0.5 / 3, 0, 0.5 / 3, 1]], np.float32)
    out = oracle.crop_and_resize_forward(img, b, np.array([0], np.int32), 1, 4)
    # crop_h == 1 -> in_y = 0.5*(y1+y2)*(H-1) = 0.5 ; crop_w == 4 -> x = 0..3
    assert np.allclose(out[0, 0, 0], [2.0, 3.0, 4.0, 5.0], atol=1e-5)
    # outside the image -> extrapolation value, rows and columns independently
    b = np.array([[-1.0, 0.0, 1.0, 1.0]], np.float32)       # y samples: -3, 0, 3 for crop 3
    out = oracle.crop_and_resize_forward(img, b, np.array([0], np.int32), 3, 2, extrapolation_value=-7.0)
    assert out[0, 0].tolist() == [[-7.0, -7.0], [0.0, 3.0], [12.0, 15.0]]


def test_crop_matches_grid_sample(oracle):
    """independent formulation: F.grid_sample(align_corners=True, bilinear, zeros) on the
    same sample points (valid boxes only; tolerance, different operation order)."""
    rs = np.random.RandomState(4)
    B, C, H, W = 2, 5, 23, 31
    img = rs.standard_normal((B, C, H, W)).astype(np.float32)
    N = 40
    y1, x1 = rs.uniform(0, 0.6, (2, N))
    boxes = np.stack([y1, x1, y1 + rs.uniform(0.05, 0.4, N), x1 + rs.uniform(0.05, 0.4, N)], 1).astype(np.float32)
    ind = rs.randint(0, B, N).astype(np.int32)
    ch, cw = 7, 5
    out = oracle.crop_and_resize_forward(img, boxes, ind, ch, cw)
    ys = boxes[:, 0:1] + (boxes[:, 2:3] - boxes[:, 0:1]) * np.linspace(0, 1, ch)[None]     # normalised
    xs = boxes[:, 1:2] + (boxes[:, 3:4] - boxes[:, 1:2]) * np.linspace(0, 1, cw)[None]
    grid = np.stack(np.broadcast_arrays(xs[:, None, :] * 2 - 1, ys[:, :, None] * 2 - 1), -1).astype(np.float32)
    ref = F.grid_sample(torch.from_numpy(img[ind]), torch.from_numpy(grid), mode="bilinear",
                        padding_mode="zeros", align_corners=True).numpy()
    assert np.allclose(out, ref, rtol=1e-4, atol=1e-4)


def test_crop_taps_and_backward_adjoint(oracle):
    rs = np.random.RandomState(8)
    shape = (2, 3, 19, 27)
    boxes = adversarial_boxes(rs, 64, 19, 27)
    ind = rs.randint(0, 2, 64).astype(np.int32)
    taps = oracle.crop_taps(boxes, 19, 27, 7, 7)
    v = taps["y_valid"].astype(bool)
    assert np.all(taps["y0"][v] >= 0) and np.all(taps["y1"][v] <= 18)
    assert np.all((taps["y1"] - taps["y0"])[v] >= 0) and np.all((taps["y1"] - taps["y0"])[v] <= 1)
    assert np.all(taps["y_frac"][v] >= 0) and np.all(taps["y_frac"][v] < 1)
    # <crop(I), G> == <I, crop^T(G)>
    img = rs.standard_normal(shape).astype(np.float32)
    out = oracle.crop_and_resize_forward(img, boxes, ind, 7, 7)
    G = rs.standard_normal(out.shape).astype(np.float32)
    gi = oracle.crop_and_resize_backward(G, boxes, ind, shape)
    lhs = float((out.astype(np.float64) * G).sum())
    rhs = float((img.astype(np.float64) * gi).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


def test_crop_bad_index_reported(oracle):
    import pytest
    img = np.zeros((1, 1, 4, 4), np.float32)
    with pytest.raises(RuntimeError):
        oracle.crop_and_resize_forward(img, np.array([[0, 0, 1, 1]], np.float32), np.array([3], np.int32), 2, 2)


def test_roi_align_box_transform(oracle):
    # a 7-pixel-wide box with crop 7 samples bin centres x1+0.5 ... x1+6.5 minus 0.5 -> x1 .. x1+6
    b = oracle.roi_align_boxes(np.array([[4.0, 2.0, 11.0, 9.0]], np.float32), 21, 31, 7, 7, True)
    assert np.allclose(b[0] * np.array([20, 30, 20, 30]), [2.0, 4.0, 8.0, 10.0], atol=1e-5)
    b = oracle.roi_align_boxes(np.array([[3.0, 5.0, 30.0, 20.0]], np.float32), 21, 31, 7, 7, False)
    assert np.allclose(b[0], [5 / 20, 3 / 30, 20 / 20, 30 / 30], atol=1e-6)


# ---------------------------------------------------------------------------- NMS
def _brute_nms(dets, thresh, strict):
    """independent float64 greedy NMS with the +1 convention (differs from fp32 only at
    exact ties, which the inputs below avoid)."""
    order = np.argsort(-dets[:, 4], kind="stable")
    b = dets[:, :4].astype(np.float64)
    area = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    alive = np.ones(len(dets), bool)
    keep = []
    for i in order:
        if not alive[i]:
            continue
        keep.append(i)
        lt = np.maximum(b[i, :2], b[:, :2])
        rb = np.minimum(b[i, 2:], b[:, 2:])
        wh = np.clip(rb - lt + 1, 0, None)
        inter = wh[:, 0] * wh[:, 1]
        iou = inter / (area[i] + area - inter)
        hit = (iou > thresh) if strict else (iou >= thresh)
        hit[i] = False
        alive &= ~hit
    return np.array(keep, np.int64)


def test_nms_known_answers(oracle):
    dets = np.array([[0, 0, 9, 9, 0.9],      # kept
                     [0, 0, 9, 4, 0.8],      # IoU with #0 = 50/100 = 0.5 exactly
                     [1, 1, 10, 10, 0.7],    # IoU with #0 = 81/119 = 0.68
                     [50, 50, 60, 60, 0.6]], np.float32)
    assert oracle.pth_nms(dets, 0.5, strict=False).tolist() == [0, 3]
    assert oracle.pth_nms(dets, 0.5, strict=True).tolist() == [0, 1, 3]     # 0.5 > 0.5 is false
    assert oracle.pth_nms(dets, 0.7).tolist() == [0, 1, 2, 3]
    # touching boxes overlap by one pixel column under the +1 convention: inter = 10, union = 190
    d = np.array([[0, 0, 9, 9, 0.9], [0, 9, 9, 18, 0.8]], np.float32)
    assert oracle.pth_nms(d, 10 / 190 - 1e-3).tolist() == [0]
    assert oracle.pth_nms(d, 10 / 190 + 1e-3).tolist() == [0, 1]
    # suppression is not transitive: #1 (suppressed by #0) must not suppress #2
    d = np.array([[0, 0, 10, 10, 0.9], [0, 4, 10, 14, 0.8], [0, 8, 10, 18, 0.7]], np.float32)
    assert oracle.pth_nms(d, 0.4).tolist() == [0, 2]


def test_nms_vs_bruteforce_and_batch_wrapper(oracle):
    rs = np.random.RandomState(12)
    for n in (1, 64, 65, 1000, 3000):
        dets = clustered_dets(rs, n, 1024)
        for thr in (0.3, 0.5, 0.7):
            for strict in (False, True):
                assert np.array_equal(oracle.pth_nms(dets, thr, strict), _brute_nms(dets, thr, strict))
    batch = np.stack([clustered_dets(rs, 500, 512, n_clusters=c) for c in (3, 30)])
    out = oracle.nms(batch, 0.7)
    ks = [oracle.pth_nms(b, 0.7) for b in batch]
    m = min(len(k) for k in ks)
    assert out.dtype == np.int32 and out.shape == (2, m)
    assert all(np.array_equal(out[i], ks[i][:m]) for i in range(2))


# ------------------------------------------------------------------------ RoIPool
def test_roipool_known_answers(oracle):
    f = np.arange(36, dtype=np.float32).reshape(1, 1, 6, 6)
    out, arg = oracle.roi_pool_forward(f, np.array([[0, 0, 0, 5, 5]], np.float32), 2, 2, 1.0)
    assert out.ravel().tolist() == [14.0, 17.0, 32.0, 35.0] and arg.ravel().tolist() == [14, 17, 32, 35]
    out, arg = oracle.roi_pool_forward(f, np.array([[0, 2, 3, 2, 3]], np.float32), 2, 2, 1.0)
    assert out.ravel().tolist() == [20.0] * 4 and arg.ravel().tolist() == [20] * 4
    out, arg = oracle.roi_pool_forward(f, np.array([[0, 50, 50, 60, 60]], np.float32), 2, 2, 1.0)
    assert out.ravel().tolist() == [0.0] * 4 and arg.ravel().tolist() == [-1] * 4
    # scale 0.5 and rounding half away from zero: x1 = 5*0.5 = 2.5 -> 3
    out, arg = oracle.roi_pool_forward(f, np.array([[0, 5, 5, 5, 5]], np.float32), 1, 1, 0.5)
    assert out.item() == 21.0
    # ties: strict '>' keeps the first maximum in (h, w) order
    g = np.zeros((1, 1, 4, 4), np.float32)
    out, arg = oracle.roi_pool_forward(g, np.array([[0, 0, 0, 3, 3]], np.float32), 1, 1, 1.0)
    assert arg.item() == 0
    # backward: gradient lands on the argmax; the malformed RoI gets none
    rois = np.array([[0, 0, 0, 5, 5], [0, 4, 4, 1, 1]], np.float32)
    out, arg = oracle.roi_pool_forward(f, rois, 2, 2, 1.0)
    top = np.ones_like(out)
    bg = oracle.roi_pool_backward(top, arg, rois, f.shape, 1.0)
    exp = np.zeros(36, np.float32)
    exp[[14, 17, 32, 35]] = 1.0
    assert np.array_equal(bg.ravel(), exp)


def test_roipool_matches_torch_maxpool_on_aligned_rois(oracle):
    """independent cross-check: RoIs aligned to a k*ph grid reduce to max_pool2d."""
    rs = np.random.RandomState(2)
    f = rs.standard_normal((1, 3, 28, 28)).astype(np.float32)
    out, arg = oracle.roi_pool_forward(f, np.array([[0, 0, 0, 27, 27]], np.float32), 7, 7, 1.0)
    ref = F.max_pool2d(torch.from_numpy(f), 4).numpy()
    assert np.array_equal(out[0], ref[0])
    assert np.array_equal(f.ravel()[arg.ravel()], out.ravel())


def test_class_mean_and_level(oracle):
    feats = np.array([[1, 2], [3, 4], [5, 6], [7, 8]], np.float32)
    gt = np.array([2, 0, 2, 1], np.int32)
    feat, cnt = oracle.class_mean(feats, gt, 4)
    assert cnt.tolist() == [[0, 1, 2, 0]]
    assert feat[:, 2].tolist() == [3.0, 4.0] and feat[:, 1].tolist() == [7.0, 8.0]
    assert np.all(feat[:, 0] == 0) and np.all(feat[:, 3] == 0)
    # 224-pixel square box on a 1024^2 image -> level 4; 112 -> 3; 448 -> 5; tiny -> 2; huge -> 5
    s = np.array([224, 112, 448, 8, 1000], np.float32) / 1024
    rois = np.stack([np.zeros(5, np.float32), np.zeros(5, np.float32), s, s], 1)
    assert oracle.roi_level(rois, 1024 * 1024).tolist() == [4, 3, 5, 2, 5]


def test_dev_stage_groups_known_answer(oracle):
    """oracle.dev_stage_groups on a hand-worked case (lib/sub_module.py:366-378, 437-598): levels 5,2,3,3,4,2,5,4."""
    r = oracle.dev_stage_groups([5, 2, 3, 3, 4, 2, 5, 4], [1, 0, 2, 3, 4, 5, 6, 0])
    assert r["order"].tolist() == [1, 5, 2, 3, 4, 7, 0, 6]
    assert r["small"][2].tolist() == [1, 5] and r["small"][3].tolist() == [2, 3] and r["small"][4].tolist() == [4, 7]
    assert r["big"][2].tolist() == [0, 2, 3, 4, 6, 7] and r["big"][3].tolist() == [0, 4, 6, 7] and r["big"][4].tolist() == [0, 6]
    assert r["small_gt_all"].tolist() == [0, 5, 2, 3, 4, 0, 0, 0]
    # a level without small boxes has no big statistics either (:456-467)
    r = oracle.dev_stage_groups([5, 3, 3, 5], None)
    assert r["big"][2].tolist() == [] and r["big"][3].tolist() == [0, 3] and r["big"][4].tolist() == []


def test_static_index_tensor_formulation_equals_the_oracle(oracle):
    """Dev._static_index_tensors (the CPU / reference form of fi_dev_stage_index) against oracle.dev_stage_groups."""
    import numpy as np
    import torch
    from feature_intertwiner_amd.sub_module import Dev
    rs = np.random.RandomState(3)
    for N in (8, 100, 513):
        level = rs.randint(2, 6, N).astype(np.int32)
        gt = rs.randint(0, 11, N).astype(np.int32)
        ref = oracle.dev_stage_groups(level, gt)
        cap = (3 * N + 63) // 64 * 64
        order, small_cls, small_gt, small_on, big_idx, big_level, big_cls, live = [
            t.numpy() for t in Dev._static_index_tensors(torch.from_numpy(level), torch.from_numpy(gt), 11, cap)]
        assert (order == ref["order"]).all() and (small_gt == ref["small_gt_all"]).all()
        bpos = 0
        for l in (2, 3, 4):
            above = np.nonzero(level > l)[0]
            assert (big_idx[bpos:bpos + len(above)] == above).all()
            counted = len(ref["big"][l]) > 0
            want = np.where((gt[above] > 0) & counted, (l - 2) * 11 + gt[above], 0)
            assert (big_cls[bpos:bpos + len(above)] == want).all()
            bpos += len(above)
        assert int(live[0]) == bpos
