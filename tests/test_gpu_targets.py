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


@pytest.mark.parametrize("with_fg", [True, False])
def test_detector_losses_kernel_matches_the_loss_functions(with_fg):
    """fi_detector_losses (the five losses and their gradients in one pass) against the loss functions of layers.py --
    which tests/test_reference_goldens.py holds to the reference's outputs -- values to 2e-6 relative, gradients with
    respect to every network output to 1e-6 of their largest element; also a batch without any foreground (the class
    loss is then switched off, lib/layers.py:866-870)."""
    from feature_intertwiner_amd import layers as L
    torch.manual_seed(4)
    b, A, R, K, P, h = 2, 3000, 64, 81, 21, 14
    n_total = 256
    match = torch.zeros(b, A, device=DEV)
    deltas = torch.zeros(b, A, 4, device=DEV)
    rows_i, rows_a = [], []
    for i in range(b):
        idx = torch.randperm(A)[:200 + 30 * i].sort()[0].to(DEV)
        match[i, idx[:60]] = 1.0
        match[i, idx[60:]] = -1.0
        deltas[i, idx[:60]] = torch.randn(60, 4, device=DEV)
        pad = n_total - idx.numel()
        rows_i.append(torch.cat((torch.full((idx.numel(),), i, device=DEV), torch.full((pad,), -1, device=DEV))))
        rows_a.append(torch.cat((idx, torch.full((pad,), -1, device=DEV))))
    r_img, r_anchor = torch.cat(rows_i).long(), torch.cat(rows_a).long()
    r_valid = r_img >= 0
    ids = torch.zeros(b, R, dtype=torch.int32, device=DEV)
    if with_fg:
        ids[0, :15] = torch.randint(1, K, (15,), dtype=torch.int32).to(DEV)
        ids[1, :7] = torch.randint(1, K, (7,), dtype=torch.int32).to(DEV)
    tdel = torch.randn(b, R, 4, device=DEV) * (ids > 0).unsqueeze(2)
    tmask = (torch.rand(b, P, 2 * h, 2 * h, device=DEV) > 0.5).float()
    mk = lambda *s: (torch.randn(*s, device=DEV) * 2).requires_grad_(True)
    outs = {}
    for form in ("kernel", "functions"):
        torch.manual_seed(7)
        row_logits, row_bbox = mk(b * n_total, 2), mk(b * n_total, 4)
        cls_logits, roi_bbox, mask_logits = mk(b, R, K), mk(b, R, K, 4), mk(b, P, 2, 2, h, h)
        if form == "kernel":
            five = L.detector_losses(row_logits, row_bbox, r_img, r_anchor, match, deltas, cls_logits, roi_bbox, ids, tdel,
                                     mask_logits, ids[:, :P], tmask)
        else:
            rc, rb = L.compute_rpn_losses_on_rows(match, deltas, r_img, r_anchor, r_valid, row_logits, row_bbox)
            five = torch.stack((rc, rb, L.compute_mrcnn_class_loss(ids, cls_logits),
                                L.compute_mrcnn_bbox_loss(tdel, ids, roi_bbox),
                                L.compute_mrcnn_mask_loss_selected(tmask, ids[:, :P], mask_logits)))
        wts = torch.tensor([1.0, 0.7, 1.3, 0.9, 1.1], device=DEV)
        (five * wts).sum().backward()
        outs[form] = (five.detach(), [t.grad for t in (row_logits, row_bbox, cls_logits, roi_bbox, mask_logits)])
    a, r = outs["kernel"], outs["functions"]
    assert float((a[0] - r[0]).abs().max()) <= 2e-6 * float(r[0].abs().max()) + 1e-9, (a[0], r[0])
    if not with_fg:
        assert float(a[0][2]) == 0.0 and float(a[0][3]) == 0.0 and float(a[0][4]) == 0.0
    for name, ga, gr in zip(("rpn logits", "rpn bbox", "class logits", "roi bbox", "mask logits"), a[1], r[1]):
        assert float((ga - gr).abs().max()) <= 1e-6 * float(gr.abs().max()) + 1e-12, name


# ---------------------------------------------------------------------------------------------------------------
# the chain closed ON the GPU: the kernels against the oracle's restatement of the reference and the reference-run goldens
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("crowd,pos_thres", [(False, 0.7), (True, 0.7), (False, 0.4)])
def test_rpn_target_kernels_equal_the_reference_rules(oracle, crowd, pos_thres):
    """prepare_rpn_target on the GPU with the kernels on (fi_rpn_targets), replayed through oracle.generate_rpn_target
    (= lib/layers.py:439-604 line by line) with the build's own random choice: candidate sets, the balancing rules and
    every derived value are the reference's.  The gpu twin of tests/test_targets.py::test_rpn_targets_equal_the_reference_rules."""
    from test_targets import _candidates, _perm_dropping, _gt as _gt_np
    from feature_intertwiner_amd import layers as L
    assert L.TARGET_KERNELS
    cfg = _cfg(backbone="resnet50", image_size=256)
    cfg.RPN.TARGET_POS_THRES = pos_thres
    anchors = L.generate_pyramid_priors(cfg.RPN.ANCHOR_SCALES, cfg.RPN.ANCHOR_RATIOS, cfg.MODEL.BACKBONE_SHAPES,
                                        cfg.MODEL.BACKBONE_STRIDES, 1).astype(np.float32)
    rs = np.random.RandomState(3)
    cls, boxes = _gt_np(rs, 3, 12, 256, [12, 7, 0], crowd)
    g = torch.Generator(device=DEV).manual_seed(5)
    match, deltas = L.prepare_rpn_target(torch.from_numpy(anchors).to(DEV), torch.from_numpy(cls).to(DEV),
                                         torch.from_numpy(boxes).to(DEV), cfg, g)
    assert getattr(match, "_fi_rows", None) is not None                  # the kernel path really ran
    n_rows, row_image, row_anchor = match._fi_rows
    match, deltas = match.cpu().numpy(), deltas.cpu().numpy()
    row_anchor = row_anchor.view(3, n_rows).cpu().numpy()
    for i in range(3):
        m0 = _candidates(oracle, anchors, cls[i], boxes[i], cfg)
        pos_c, neg_c = np.nonzero(m0 == 1)[0], np.nonzero(m0 == -1)[0]
        kept_pos, kept_neg = np.nonzero(match[i] == 1)[0], np.nonzero(match[i] == -1)[0]
        assert set(kept_pos) <= set(pos_c) and set(kept_neg) <= set(neg_c)
        if pos_thres < 0.5 and i == 0:
            assert len(pos_c) > 128 and len(kept_pos) == 128
        exp_match, exp_bbox = oracle.generate_rpn_target(anchors, cls[i], boxes[i], cfg,
                                                         _perm_dropping(pos_c, kept_pos), _perm_dropping(neg_c, kept_neg))
        assert np.array_equal(match[i], exp_match), i
        n_pos = int((exp_match == 1).sum())
        if i < 2:
            assert n_pos > 0 and n_pos + int((exp_match == -1).sum()) == 256
        got = deltas[i][exp_match == 1]
        exp = exp_bbox[:n_pos] / np.asarray(cfg.DATA.BBOX_STD_DEV, np.float32)
        assert np.allclose(got, exp, rtol=1e-5, atol=1e-6)
        assert np.all(deltas[i][exp_match != 1] == 0)
        nz = np.nonzero(exp_match)[0]                                     # the rows the losses read = the reference's
        assert np.array_equal(row_anchor[i, :len(nz)], nz) and np.all(row_anchor[i, len(nz):] == -1)
    assert np.all(match[2] != 1)


def test_detector_losses_kernel_returns_the_reference_values(golden_dir):
    """fi_detector_losses on the inputs of tests/golden/layers.npz: the five values the REFERENCE's loss functions
    (lib/layers.py:808-934, run by oracle/gen_golden_layers.py) returned, to 2e-6.  The kernel reads the RPN outputs as
    rows of the non-zero anchors and the mask head's target-class logits in the un-shuffled layout, so the golden
    inputs are re-laid here (no arithmetic beyond logit(p) for the mask probabilities)."""
    import os
    from helpers import golden_loss_inputs
    from feature_intertwiner_amd import layers as L
    assert L.LOSS_KERNEL
    gold = np.load(os.path.join(golden_dir, "layers.npz"))
    li = golden_loss_inputs()
    match = li["rpn_match"]
    B, A = match.shape
    n_total = 256
    per_anchor = np.zeros((B, A, 4), np.float32)
    r_img = np.full((B, n_total), -1, np.int64)
    r_anchor = np.full((B, n_total), -1, np.int64)
    row_logits = np.zeros((B, n_total, 2), np.float32)
    row_bbox = np.zeros((B, n_total, 4), np.float32)
    for b in range(B):
        pos = np.nonzero(match[b] == 1)[0]
        per_anchor[b, pos] = li["rpn_bbox_target"][b, :len(pos)]          # the reference packs them (lib/layers.py:845-852)
        nz = np.nonzero(match[b])[0]
        assert len(nz) <= n_total
        r_img[b, :len(nz)], r_anchor[b, :len(nz)] = b, nz
        row_logits[b, :len(nz)] = li["rpn_logits"][b, nz]
        row_bbox[b, :len(nz)] = li["rpn_bbox_pred"][b, nz]
    ids = li["cls_ids"]
    R = ids.shape[1]
    p = li["mask_pred"].astype(np.float64)                                # [B, R, NC, 28, 28] probabilities
    sel = np.take_along_axis(p, ids.reshape(B, R, 1, 1, 1).astype(np.int64), 2)[:, :, 0]     # [B, R, 28, 28]
    logit = np.log(sel) - np.log1p(-sel)
    un = logit.reshape(B, R, 14, 2, 14, 2).transpose(0, 1, 3, 5, 2, 4).astype(np.float32)    # [.., a, b, y, x]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    five = L.detector_losses(T(row_logits.reshape(-1, 2)), T(row_bbox.reshape(-1, 4)), T(r_img.reshape(-1)),
                             T(r_anchor.reshape(-1)), T(match.astype(np.float32)), T(per_anchor), T(li["cls_logits"]),
                             T(li["bbox_pred"]), T(ids.astype(np.int32)), T(li["bbox_target"]), T(un),
                             T(ids.astype(np.int32)), T(li["mask_target"]))
    assert five is not None
    five = five.cpu().numpy()
    names = ("loss_rpn_class", "loss_rpn_bbox", "loss_mrcnn_class", "loss_mrcnn_bbox", "loss_mrcnn_mask")
    for v, k in zip(five, names):
        assert abs(float(v) - float(gold[k])) <= 2e-6 * max(1.0, abs(float(gold[k]))), (k, float(v), float(gold[k]))
