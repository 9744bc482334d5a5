"""GPU: the target-generation kernels (csrc/targets.hip: fi_rpn_targets, fi_detection_targets) against the tensor
formulation of feature_intertwiner_amd/layers.py -- which tests/test_targets.py holds to the oracle's restatement of
lib/layers.py:224-376, 439-604 -- given the SAME sampling keys: bit-identical outputs (match vectors, kept rows, RoIs,
class ids, refinements, mask targets), incl. crowd boxes, images without objects, the positive-reduction branch,
fewer candidates than the budget, and keys with exact ties (lower index first)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(**kw):
    from feature_intertwiner_amd.config import make_config
    return make_config(**kw)


def _gt(rs, b, G, size, n_valid, crowd=False):
    side = np.exp(rs.uniform(np.log(12), np.log(size / 2), (b, G)))
    asp = np.exp(rs.uniform(np.log(0.5), np.log(2.0), (b, G)))
    h, w = side / np.sqrt(asp), side * np.sqrt(asp)
    y1, x1 = rs.uniform(0, size - h), rs.uniform(0, size - w)
    boxes = np.stack([y1, x1, y1 + h, x1 + w], 2).astype(np.float32)
    cls = rs.randint(1, 81, (b, G)).astype(np.int64)
    for i in range(b):
        cls[i, n_valid[i]:] = 0
        boxes[i, n_valid[i]:] = 0
    if crowd:
        cls[0, 1] = -1
    return torch.from_numpy(cls).to(DEV), torch.from_numpy(boxes).to(DEV)


def _unique_keys(b, n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.stack([1.0 + (torch.randperm(n, generator=g).float() + 0.5) / n for _ in range(b)]).to(DEV)


@pytest.mark.parametrize("size,crowd,pos_thres,ties", [(256, False, 0.7, False), (256, True, 0.7, False),
                                                       (256, False, 0.4, False), (1024, True, 0.7, False),
                                                       (256, False, 0.4, True)])
def test_rpn_target_kernels_equal_the_tensor_formulation(size, crowd, pos_thres, ties):
    from feature_intertwiner_amd import layers as L
    cfg = _cfg(backbone="resnet50", image_size=size)
    cfg.RPN.TARGET_POS_THRES = pos_thres
    anchors = torch.from_numpy(L.generate_pyramid_priors(cfg.RPN.ANCHOR_SCALES, cfg.RPN.ANCHOR_RATIOS,
                                                         cfg.MODEL.BACKBONE_SHAPES, cfg.MODEL.BACKBONE_STRIDES,
                                                         1).astype(np.float32)).to(DEV)
    A = anchors.size(0)
    rs = np.random.RandomState(3)
    cls, boxes = _gt(rs, 4, 12, size, [12, 7, 0, 1], crowd)
    kp, kn = _unique_keys(4, A, 1), _unique_keys(4, A, 2)
    if ties:            # 64 distinct key values: every selection boundary falls inside a run of equal keys
        kp, kn = 1.0 + torch.floor((kp - 1.0) * 64) / 64, 1.0 + torch.floor((kn - 1.0) * 64) / 64
        kp[0, :5] = 2.0                                        # rand + 1 can round up to exactly 2
    match, deltas = L.rpn_target_from_keys(anchors, cls, boxes, cfg, kp, kn, kernels=True)
    torch.cuda.synchronize()
    n_total = cfg.RPN.TRAIN_ANCHORS_PER_IMAGE
    if not ties:
        rm, rd = L.rpn_target_from_keys(anchors, cls, boxes, cfg, kp, kn, kernels=False)
        assert torch.equal(match, rm)
        assert torch.equal(deltas.view(torch.int32), rd.view(torch.int32))
    else:
        # reference with the kernels' tie rule: largest key first, lower index first (a stable sort of the negated keys)
        big = cfg.RPN.TRAIN_ANCHORS_PER_IMAGE
        cfg.RPN.TRAIN_ANCHORS_PER_IMAGE = 2 * A
        cand, rd = L.rpn_target_from_keys(anchors, cls, boxes, cfg, kp, kn, kernels=False)       # budget never binds
        cfg.RPN.TRAIN_ANCHORS_PER_IMAGE = big
        cand, kpn, knn = cand.cpu().numpy(), kp.cpu().numpy(), kn.cpu().numpy()
        exp = np.zeros_like(cand)
        for i in range(4):
            pos = np.nonzero(cand[i] == 1)[0]
            pos = pos[np.argsort(-kpn[i][pos], kind="stable")][:n_total // 2]
            neg = np.nonzero(cand[i] == -1)[0]
            neg = neg[np.argsort(-knn[i][neg], kind="stable")][:max(n_total - len(pos), 0)]
            exp[i][pos], exp[i][neg] = 1, -1
        assert np.array_equal(match.cpu().numpy(), exp)
        assert (match[0, :5] == 1).sum() == (torch.from_numpy(cand[0, :5]) == 1).sum()       # the 2.0 keys rank first
    # the compact rows the kernel lists = the non-zero anchors in anchor order, -1 padded per image
    n, ri, ra = match._fi_rows
    assert n == n_total
    ri, ra = ri.view(4, n_total).cpu().numpy(), ra.view(4, n_total).cpu().numpy()
    mm = match.cpu().numpy()
    for i in range(4):
        nz = np.nonzero(mm[i])[0]
        assert np.array_equal(ra[i, :len(nz)], nz) and np.all(ri[i, :len(nz)] == i)
        assert np.all(ra[i, len(nz):] == -1) and np.all(ri[i, len(nz):] == -1)
    assert (mm[2] == 1).sum() == 0 and (mm[0] == 1).sum() > 0
    if pos_thres < 0.5:
        assert (mm[0] == 1).sum() == n_total // 2              # the positive reduction really happened


@pytest.mark.parametrize("P,R,crowd,ties", [(1000, 512, False, False), (1000, 200, True, False), (300, 64, False, False),
                                            (2048, 512, True, True)])
def test_detection_target_kernel_equals_the_tensor_formulation(P, R, crowd, ties):
    from feature_intertwiner_amd import layers as L
    cfg = _cfg(backbone="resnet50", image_size=256, train_rois_per_image=R)
    rs = np.random.RandomState(5)
    b, G = 4, 16
    cls, boxes = _gt(rs, b, G, 1.0, [16, 5, 0, 1], crowd)
    g = torch.Generator().manual_seed(9)
    # proposals: jittered copies of the objects + background, normalised
    src = boxes[:, torch.randint(0, G, (P,), generator=g)].clone()
    jit = (torch.rand(b, P, 4, generator=g).to(DEV) - 0.5) * 0.1
    props = torch.where(torch.rand(b, P, 1, generator=g).to(DEV) < 0.5, src + jit,
                        torch.rand(b, P, 4, generator=g).to(DEV).sort(dim=2)[0][..., [0, 1, 2, 3]])
    y = torch.stack([torch.minimum(props[..., 0], props[..., 2]), torch.minimum(props[..., 1], props[..., 3]),
                     torch.maximum(props[..., 0], props[..., 2]), torch.maximum(props[..., 1], props[..., 3])], 2)
    props = y.clamp(0, 1).contiguous()
    num = torch.tensor([P, P - 17, P, 40], device=DEV)
    masks = (torch.rand(b, G, 56, 56, generator=g) > 0.5).float().to(DEV)
    kp, kn = _unique_keys(b, P, 3), _unique_keys(b, P, 4)
    if ties:
        kp, kn = 1.0 + torch.floor((kp - 1.0) * 16) / 16, 1.0 + torch.floor((kn - 1.0) * 16) / 16
    got = L.det_target_from_keys(props, num, cls, boxes, masks, cfg, kp, kn, kernels=True)
    torch.cuda.synchronize()
    if not ties:
        ref = L.det_target_from_keys(props, num, cls, boxes, masks, cfg, kp, kn, kernels=False)
        for name, a, r in zip(("rois", "class ids", "deltas", "masks"), got, ref):
            assert a.shape == r.shape and a.dtype == r.dtype, name
            assert torch.equal(a, r), name
        assert int((got[1] > 0).sum()) > 0 and int((got[1][2] > 0).sum()) == 0
    else:
        # the slots follow (key descending, index ascending): recover the order from the RoIs
        rois, ids = got[0].cpu().numpy(), got[1].cpu().numpy()
        pc = int(R * cfg.ROIS.ROI_POSITIVE_RATIO)
        assert ((ids > 0).sum(1) <= pc).all() and (ids[:, pc:] == 0).all()
