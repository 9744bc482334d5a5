"""HIP NMS vs the CPU oracle: keep indices must match EXACTLY."""
import numpy as np
import pytest
import torch

from helpers import clustered_dets

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 200, 1000, 3000, 6000])
@pytest.mark.parametrize("thresh", [0.3, 0.5, 0.7])
def test_pth_nms_matches_oracle(oracle, n, thresh):
    from feature_intertwiner_amd.nms.pth_nms import pth_nms
    rs = np.random.RandomState(n + int(thresh * 10))
    for pixel in (False, True):
        dets = clustered_dets(rs, n, 1024, pixel_round=pixel)
        for strict in (False, True):
            exp = oracle.pth_nms(dets, thresh, strict)
            got = pth_nms(torch.from_numpy(dets).to(DEV), thresh, strict=strict).cpu().numpy()
            assert got.dtype == np.int64
            assert np.array_equal(got, exp), (n, thresh, pixel, strict, len(got), len(exp))


@pytest.mark.parametrize("n", [7104, 7168, 7169, 9000, 14336, 14400])
def test_scan_kernel_variants_by_size(oracle, n):
    """nms_scan_wide_kernel<8> up to 112 column words (7168 boxes), <4> up to 224 (14336), the one-word-per-thread scan
    above; the last block partly filled or full."""
    from feature_intertwiner_amd.nms.pth_nms import pth_nms
    rs = np.random.RandomState(n)
    dets = clustered_dets(rs, n, 1024, n_clusters=40)
    exp = oracle.pth_nms(dets, 0.7, False)
    got = pth_nms(torch.from_numpy(dets).to(DEV), 0.7, strict=False).cpu().numpy()
    assert np.array_equal(got, exp), (n, len(got), len(exp))


def test_threshold_tie_ge_vs_gt(oracle):
    """IoU exactly equal to the threshold: `>=` (CPU spec) suppresses, `>` (CUDA) keeps."""
    from feature_intertwiner_amd.nms.pth_nms import pth_nms
    # boxes [0,0,9,9] and [0,0,9,4] (+1 convention): inter 50, union 100 -> IoU 0.5 exactly
    dets = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 4, 0.8], [50, 50, 60, 60, 0.7]], np.float32)
    t = torch.from_numpy(dets).to(DEV)
    assert pth_nms(t, 0.5, strict=False).cpu().tolist() == [0, 2] == oracle.pth_nms(dets, 0.5, False).tolist()
    assert pth_nms(t, 0.5, strict=True).cpu().tolist() == [0, 1, 2] == oracle.pth_nms(dets, 0.5, True).tolist()


def test_thresholds_at_and_next_to_attained_ious(oracle):
    """nms_mask_kernel decides a pair from inter * rcp(union) unless that lies within 2^-20 of the threshold, where it
    takes the IEEE division the reference makes: thresholds EQUAL to an IoU the boxes attain, and one float above / below."""
    from feature_intertwiner_amd.nms.pth_nms import pth_nms
    rs = np.random.RandomState(31)
    checked = 0
    for trial in range(12):
        dets = clustered_dets(rs, 96, 256, n_clusters=6, pixel_round=bool(trial & 1))
        order = np.argsort(-dets[:, 4], kind="stable")
        d = dets[order]
        x1, y1, x2, y2 = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
        area = (x2 - x1 + np.float32(1)) * (y2 - y1 + np.float32(1))
        ious = []
        for i in range(0, 40, 3):
            for j in range(i + 1, min(i + 6, 96)):
                w = max(np.float32(0), min(x2[i], x2[j]) - max(x1[i], x1[j]) + np.float32(1))
                h = max(np.float32(0), min(y2[i], y2[j]) - max(y1[i], y1[j]) + np.float32(1))
                inter = np.float32(w * h)
                if inter > 0:
                    ious.append(np.float32(inter / np.float32(np.float32(area[i] + area[j]) - inter)))
        t = torch.from_numpy(dets).to(DEV)
        for iou in ious[:6]:
            for thr in (iou, np.nextafter(iou, np.float32(2)), np.nextafter(iou, np.float32(0))):
                for strict in (False, True):
                    exp = oracle.pth_nms(dets, float(thr), strict)
                    got = pth_nms(t, float(thr), strict=strict).cpu().numpy()
                    assert np.array_equal(got, exp), (trial, float(thr), strict)
                    checked += 1
    assert checked >= 200


def test_unsorted_input_and_duplicates(oracle):
    from feature_intertwiner_amd.nms.pth_nms import pth_nms
    rs = np.random.RandomState(77)
    dets = clustered_dets(rs, 500, 512)
    perm = rs.permutation(500)
    dets = dets[perm]                       # caller did not pre-sort
    dets[100:140, :4] = dets[100, :4]       # identical boxes
    exp = oracle.pth_nms(dets, 0.7)
    got = pth_nms(torch.from_numpy(dets).to(DEV), 0.7).cpu().numpy()
    assert np.array_equal(got, exp)


def test_batched_wrapper_matches_oracle(oracle):
    from feature_intertwiner_amd.nms.nms_wrapper import nms
    rs = np.random.RandomState(5)
    dets = np.stack([clustered_dets(rs, 2000, 1024, n_clusters=c) for c in (5, 20, 60)])
    exp = oracle.nms(dets, 0.7)
    got = nms(torch.from_numpy(dets).to(DEV), 0.7)
    assert got.dtype == np.int32 and got.shape == exp.shape
    assert np.array_equal(got, exp)


def test_max_keep_is_a_prefix(oracle):
    from feature_intertwiner_amd.nms.pth_nms import nms_sorted
    rs = np.random.RandomState(8)
    dets = clustered_dets(rs, 6000, 1024)
    exp = oracle.pth_nms(dets, 0.7)
    t = torch.from_numpy(dets).to(DEV)
    for mk in (1, 64, 65, 1000, 5999):
        keep, num = nms_sorted(t, 0.7, max_keep=mk)
        n = int(num.item())
        assert n == min(mk, len(exp))
        assert np.array_equal(keep[:n].cpu().numpy(), exp[:n])


def test_properties_full_size():
    """N = 6000 (config RPN.PRE_NMS_LIMIT), batch 4: survivors are pairwise below the
    threshold, every suppressed box overlaps an earlier survivor (greedy invariant),
    and NMS of the survivors is the identity (idempotence)."""
    from feature_intertwiner_amd.nms.pth_nms import nms_sorted
    rs = np.random.RandomState(99)
    dets = np.stack([clustered_dets(rs, 6000, 1024) for _ in range(4)])
    t = torch.from_numpy(dets).to(DEV)
    keep, num = nms_sorted(t, 0.7)
    for b in range(4):
        n = int(num[b].item())
        k = keep[b, :n]
        assert torch.all(k[1:] > k[:-1])                      # visit order
        kb = t[b, k, :4].double()
        area = (kb[:, 2] - kb[:, 0] + 1) * (kb[:, 3] - kb[:, 1] + 1)
        lt = torch.maximum(kb[:, None, :2], kb[None, :, :2])
        rb = torch.minimum(kb[:, None, 2:], kb[None, :, 2:])
        wh = (rb - lt + 1).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        iou = inter / (area[:, None] + area[None, :] - inter)
        iou.fill_diagonal_(0)
        assert iou.max().item() < 0.7 + 1e-6
        keep2, num2 = nms_sorted(t[b, k].contiguous(), 0.7)
        assert int(num2.item()) == n and torch.equal(keep2[:n], torch.arange(n, device=DEV))


def test_empty_input():
    from feature_intertwiner_amd.nms.pth_nms import nms_sorted
    keep, num = nms_sorted(torch.zeros(2, 0, 5, device=DEV), 0.7)
    assert num.tolist() == [0, 0] and keep.shape == (2, 0)
