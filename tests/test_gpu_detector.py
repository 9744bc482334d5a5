"""GPU: the callers wired to the HIP operators -- proposal layer (NMS), detection targets
(RoIAlign on GT masks), the Dev pyramid stage vs a per-level restatement on the oracle,
and one full train step."""
import numpy as np
import pytest
import torch

from helpers import training_rois

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(**kw):
    from feature_intertwiner_amd.config import make_config
    return make_config(**kw)


def test_det_targets_invariants():
    from feature_intertwiner_amd import layers as L
    from feature_intertwiner_amd.synthetic import synthetic_batch
    cfg = _cfg(backbone="resnet50", image_size=512, batch_size=3, train_rois_per_image=128)
    _, gt_cls, gt_boxes, gt_masks = synthetic_batch(3, 512, device=DEV)
    gt_cls[1, 15:] = 0
    gt_boxes[1, 15:] = 0
    gen = torch.Generator(device=DEV).manual_seed(1)
    scale = torch.tensor([512.0] * 4, device=DEV)
    # proposals: jittered GT copies + random boxes, image 2 has only 300 valid rows
    jit = gt_boxes.repeat(1, 20, 1) * (1 + 0.1 * (torch.rand(3, 400, 4, device=DEV, generator=gen) - 0.5))
    rnd = torch.rand(3, 600, 2, device=DEV, generator=gen) * 400
    rnd = torch.cat([rnd, rnd + 8 + torch.rand(3, 600, 2, device=DEV, generator=gen) * 100], 2)
    perm = torch.randperm(1000, device=DEV, generator=gen)
    props = (torch.cat([jit, rnd], 1)[:, perm] / scale).clamp(0, 1)
    zero_rows = props.abs().sum(2) == 0          # copies of the zero-padded GT rows of image 1
    props[zero_rows] = torch.tensor([0.9, 0.9, 0.95, 0.95], device=DEV)
    num = torch.tensor([1000, 1000, 300], device=DEV, dtype=torch.int32)
    props[2, 300:] = 0
    rois, cls, deltas, masks = L.prepare_det_target(props, num, gt_cls, gt_boxes / scale, gt_masks, cfg, gen)
    assert rois.shape == (3, 128, 4) and cls.dtype == torch.int32 and masks.shape == (3, 128, 28, 28)
    assert torch.isfinite(deltas).all()
    gtn = gt_boxes / scale
    for b in range(3):
        n_pos = int((cls[b] > 0).sum())
        used = int((rois[b].abs().sum(1) > 0).sum())
        assert 0 < n_pos <= int(128 * 0.33)
        assert torch.all(cls[b, :n_pos] > 0) and torch.all(cls[b, n_pos:] == 0)        # positives first
        assert used - n_pos <= int(n_pos / 0.33 - n_pos)                               # reference ratio rule
        valid_gt = gt_cls[b] > 0
        iou = L.bbox_overlaps(rois[b], gtn[b][valid_gt])
        best, arg = iou.max(1)
        assert torch.all(best[:n_pos] >= 0.5) and torch.all(best[n_pos:used] < 0.5)
        assert torch.equal(cls[b, :n_pos].long(), gt_cls[b][valid_gt][arg[:n_pos]])
        exp = L.box_refinement(rois[b, :n_pos], gtn[b][valid_gt][arg[:n_pos]]) / torch.tensor(
            cfg.DATA.BBOX_STD_DEV, device=DEV)
        assert torch.allclose(deltas[b, :n_pos], exp, rtol=1e-4, atol=1e-4)
        assert torch.all(deltas[b, n_pos:] == 0) and torch.all(rois[b, used:] == 0)
        assert torch.all((masks[b] == 0) | (masks[b] == 1)) and torch.all(masks[b, n_pos:] == 0)
        assert masks[b, :n_pos].sum() > 0
        # every RoI is one of the valid proposals of its own image
        d = (rois[b, :used, None, :] - props[b, None, :int(num[b]), :]).abs().sum(2).min(1)[0]
        assert torch.all(d == 0)


def _priors(cfg):
    from feature_intertwiner_amd import layers as L
    return torch.from_numpy(L.generate_pyramid_priors(cfg.RPN.ANCHOR_SCALES, cfg.RPN.ANCHOR_RATIOS,
                                                      cfg.MODEL.BACKBONE_SHAPES, cfg.MODEL.BACKBONE_STRIDES, 1)).float().to(DEV)


def _check_candidates(oracle, tap, cfg, size):
    """fi_proposal_candidates vs the oracle restatement, per image: the SAME candidates in the SAME order (scores
    identical, which pins the selection and the tie rule), boxes to 2 ulp of the image size (the device exp and
    numpy's differ in the last bit)."""
    probs, deltas, anchors, extra, dets = (tap[k] for k in ("probs", "deltas", "anchors", "extra", "dets"))
    K = dets.shape[1]
    for b in range(probs.shape[0]):
        exp, order = oracle.proposal_candidates(probs[b].cpu().numpy(), deltas[b].cpu().numpy(), anchors.cpu().numpy(), K,
                                                cfg.DATA.BBOX_STD_DEV, (size, size),
                                                None if extra is None else extra[b].cpu().numpy())
        got = dets[b].cpu().numpy()
        assert np.array_equal(got[:, 4], exp[:, 4]), "selection / order differs"
        assert np.all(np.diff(got[:, 4]) <= 0)
        assert np.max(np.abs(got[:, :4] - exp[:, :4])) <= 3e-7 * size * 8, np.max(np.abs(got[:, :4] - exp[:, :4]))
    return True


@pytest.mark.parametrize("multi_wg", [True, False])
def test_proposal_layer_matches_oracle(oracle, multi_wg, monkeypatch):
    """proposal_layer = fi_proposal_candidates(_ws) + fi_nms_sorted + fi_proposal_gather.  Candidates vs the oracle
    (selection exact); keep list = the oracle's NMS on the kernel's own dets (exact); gathered proposals exact.
    multi_wg: the eight-launch selection with its state in a workspace (round 6) / the single kernel."""
    from feature_intertwiner_amd import _lib
    from feature_intertwiner_amd import layers as L
    monkeypatch.setattr(L, "PROPOSAL_MULTI_WG", multi_wg)
    cfg = _cfg(backbone="resnet50", image_size=256)
    pri = _priors(cfg)
    g = torch.Generator(device=DEV).manual_seed(2)
    A = pri.size(0)
    probs = torch.rand(2, A, 2, device=DEV, generator=g)
    bbox = torch.randn(2, A, 4, device=DEV, generator=g) * 0.5
    taps = {}
    _lib.TAP = lambda name, **kw: taps.setdefault(name, kw)
    try:
        props, num = L.proposal_layer([probs, bbox], 1000, 0.7, pri, cfg)
    finally:
        _lib.TAP = None
    assert props.shape == (2, 1000, 4)
    tap = taps["proposal_candidates"]
    assert tap["dets"].shape == (2, min(6000, A), 5)
    _check_candidates(oracle, tap, cfg, 256.0)
    for b in range(2):
        dets = tap["dets"][b].cpu().numpy()
        keep = oracle.pth_nms(dets, 0.7)[:1000]
        n = int(num[b])
        assert n == len(keep)
        assert np.array_equal(props[b, :n].cpu().numpy(), dets[keep, :4] / np.float32(256.0))
        assert torch.all(props[b, n:] == 0)


@pytest.mark.parametrize("multi_wg", [True, False])
@pytest.mark.parametrize("case", ["ties", "extra", "few", "all_equal", "negative_and_nan_free"])
def test_proposal_candidates_edge_cases(oracle, case, multi_wg, monkeypatch):
    """Tie rule (lower index first; more ties at the threshold than places), external candidates merged by score
    (before anchors at equal score), fewer candidates than PRE_NMS_LIMIT, constant scores, scores of both signs."""
    from feature_intertwiner_amd import _lib
    from feature_intertwiner_amd import layers as L
    monkeypatch.setattr(L, "PROPOSAL_MULTI_WG", multi_wg)
    cfg = _cfg(backbone="resnet50", image_size=256)
    pri = _priors(cfg)
    A = pri.size(0)
    g = torch.Generator(device=DEV).manual_seed(5)
    probs = torch.rand(3, A, 2, device=DEV, generator=g)
    bbox = torch.randn(3, A, 4, device=DEV, generator=g) * 0.3
    extra = None
    if case == "ties":
        probs[:, :, 1] = torch.round(probs[:, :, 1] * 50) / 50          # ~51 distinct values: thousands of ties
    elif case == "extra":
        e_box = torch.rand(3, 700, 4, device=DEV, generator=g) * 200
        e_sc = torch.rand(3, 700, 1, device=DEV, generator=g)
        e_sc[:, :10, 0] = probs[:, :10, 1]                                # exact ties between extra and anchors
        e_sc[:, 10:40] = 1.5
        extra = torch.cat([e_box, e_sc], 2)
    elif case == "few":
        pri, probs, bbox = pri[:3000], probs[:, :3000], bbox[:, :3000]
    elif case == "all_equal":
        probs[:, :, 1] = 0.25
    else:
        probs[:, :, 1] = torch.randn(3, probs.size(1), device=DEV, generator=g)
    taps = {}
    _lib.TAP = lambda name, **kw: taps.setdefault(name, kw)
    try:
        props, num = L.proposal_layer([probs, bbox], 1000, 0.7, pri, cfg, extra)
    finally:
        _lib.TAP = None
    tap = taps["proposal_candidates"]
    assert tap["dets"].shape[1] == min(6000, pri.size(0) + (0 if extra is None else 700))
    _check_candidates(oracle, tap, cfg, 256.0)
    assert int(num.min()) >= 1


def test_dev_stage_matches_per_level_restatement(oracle):
    """Dev.forward (one pyramid launch per crop size) vs the reference's per-level procedure
    (lib/sub_module.py:429-662) restated on the CPU oracle: pooled 7x7 / 14x14 outputs."""
    from feature_intertwiner_amd.sub_module import Dev
    cfg = _cfg(backbone="resnet50", image_size=256, batch_size=2, train_rois_per_image=64, dev_switch=True)
    torch.manual_seed(0)
    dev = Dev(cfg, 256).to(DEV).eval()
    maps = [torch.randn(2, 256, s, s, device=DEV) for s in (64, 32, 16, 8)]
    rs = np.random.RandomState(1)
    rois_np = training_rois(rs, 2, 64)
    rois_np[1, 60:] = 0                                   # zero-padded rows
    rois = torch.from_numpy(rois_np).to(DEV)
    cls = torch.randint(0, 81, (2, 64), device=DEV, dtype=torch.int32)
    with torch.no_grad():
        pooled, mask, feat_out = dev(maps, rois, cls)
        from feature_intertwiner_amd.conv import conv_bn_act
        up = [conv_bn_act(m, dev.upsample[0][0], dev.upsample[0][1], relu=True).cpu().numpy() for m in maps]
    level = oracle.roi_level(rois_np.reshape(-1, 4), 256 * 256)
    ind = np.repeat(np.arange(2, dtype=np.int32), 64)
    for size, got in ((7, pooled), (14, mask)):
        exp = np.zeros((128, 256, size, size), np.float32)
        for l in range(2, 6):
            sel = np.nonzero(level == l)[0]
            if len(sel):
                exp[sel] = oracle.crop_and_resize_forward(up[l - 2], rois_np.reshape(-1, 4)[sel], ind[sel], size, size)
        assert np.array_equal(got.cpu().numpy().view(np.uint32), exp.view(np.uint32))
    big_feat, big_cnt, small_feat, small_cnt = feat_out[:4]
    assert big_feat.shape == (1, 3, 1024, 81) and small_cnt.shape == (1, 3, 1, 81)
    cls_np = cls.cpu().numpy().reshape(-1)
    for i, l in enumerate((2, 3, 4)):
        exp_cnt = np.bincount(cls_np[(level == l) & (cls_np > 0)], minlength=81)
        assert np.array_equal(small_cnt[0, i, 0].cpu().numpy(), exp_cnt.astype(np.float32))
        exp_big = np.bincount(cls_np[(level > l) & (cls_np > 0)], minlength=81) * float((level == l).any())
        assert np.array_equal(big_cnt[0, i, 0].cpu().numpy(), exp_big.astype(np.float32))
    assert not big_feat.requires_grad


def test_train_step_runs_and_learns():
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import set_optimizer, train_step
    torch.manual_seed(2000)
    cfg = _cfg(backbone="resnet50", image_size=256, batch_size=2, train_rois_per_image=64, ot_L=5)
    model = MaskRCNN(cfg).to(DEV)
    opt = set_optimizer(model, cfg.TRAIN)
    batch = synthetic_batch(2, 256, device=DEV)
    model.external_proposals = SyntheticProposals(batch[2], 256)
    model.generator = torch.Generator(device=DEV).manual_seed(3)
    hist = []
    for _ in range(6):
        t = train_step(model, opt, list(batch))
        assert all(torch.isfinite(v) for v in t.values()), t
        hist.append(float(t["total"]))
    assert hist[-1] < hist[0]
    assert all(p.grad is not None for n, p in model.named_parameters() if not n.startswith("ot_loss"))


def _detection_layer_restated(rois, probs, deltas, windows, cfg, oracle):
    """lib/layers.py:664-802 step by step on the host: per image, per class present, sort by
    score, NMS (oracle, `>=`), union, top DET_MAX_INSTANCES by score."""
    bs, N = rois.shape[:2]
    K = probs.shape[1]
    out = np.zeros((bs, cfg.TEST.DET_MAX_INSTANCES, 6), np.float32)
    ids = probs.argmax(1)
    sc = probs.max(1)
    d = deltas[np.arange(bs * N), ids] * np.asarray(cfg.DATA.BBOX_STD_DEV, np.float32)
    b = rois.reshape(-1, 4)
    hgt, wid = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    cy, cx = b[:, 0] + np.float32(0.5) * hgt, b[:, 1] + np.float32(0.5) * wid
    cy, cx = cy + d[:, 0] * hgt, cx + d[:, 1] * wid
    hgt, wid = hgt * np.exp(d[:, 2]), wid * np.exp(d[:, 3])
    y1, x1 = cy - np.float32(0.5) * hgt, cx - np.float32(0.5) * wid
    ref = np.stack([y1, x1, y1 + hgt, x1 + wid], 1) * np.float32(cfg.DATA.IMAGE_SHAPE[0])
    for i in range(bs):
        w = windows[i]
        r = ref[i * N:(i + 1) * N]
        r = np.round(np.stack([r[:, 0].clip(w[0], w[2]), r[:, 1].clip(w[1], w[3]),
                               r[:, 2].clip(w[0], w[2]), r[:, 3].clip(w[1], w[3])], 1))
        cid, s = ids[i * N:(i + 1) * N], sc[i * N:(i + 1) * N]
        ok = (cid > 0) & (s >= cfg.TEST.DET_MIN_CONFIDENCE) & ((r[:, 0] - r[:, 2]) * (r[:, 1] - r[:, 3]) > 0)
        kept = []
        for c in np.unique(cid[ok]):
            ix = np.nonzero(ok & (cid == c))[0]
            ix = ix[np.argsort(-s[ix], kind="stable")]
            dets = np.concatenate([r[ix], s[ix, None]], 1).astype(np.float32)
            kept.extend(ix[oracle.pth_nms(dets, cfg.TEST.DET_NMS_THRESHOLD)].tolist())
        kept = np.array(sorted(kept), np.int64)
        top = kept[np.argsort(-s[kept], kind="stable")][:cfg.TEST.DET_MAX_INSTANCES]
        out[i, :len(top)] = np.concatenate([r[top], cid[top, None].astype(np.float32), s[top, None]], 1)
    return out


@pytest.mark.parametrize("min_conf", [0.0, 0.5])
def test_detection_layer_matches_per_class_restatement(oracle, min_conf):
    from feature_intertwiner_amd import layers as L
    cfg = _cfg(backbone="resnet50", image_size=512)
    cfg.TEST.DET_MIN_CONFIDENCE = min_conf
    rs = np.random.RandomState(5)
    bs, N, K = 3, 400, 81
    # clustered proposals so that NMS has work: 12 centres, jittered copies; a few degenerate rows
    ctr = rs.uniform(0.15, 0.85, (bs, 12, 2))
    sz = rs.uniform(0.05, 0.3, (bs, 12, 2))
    pick = rs.randint(0, 12, (bs, N))
    c = np.take_along_axis(ctr, pick[..., None], 1) + rs.normal(0, 0.01, (bs, N, 2))
    s = np.take_along_axis(sz, pick[..., None], 1) * rs.uniform(0.9, 1.1, (bs, N, 2))
    rois = np.concatenate([c - s / 2, c + s / 2], 2).clip(0, 1).astype(np.float32)
    rois[1, 390:] = 0
    logits = rs.standard_normal((bs * N, K)).astype(np.float32)
    cls_of_cluster = rs.randint(0, 6, (bs, 12))            # few classes (incl. background 0) => same-class clusters
    fav = np.take_along_axis(cls_of_cluster, pick, 1).reshape(-1)
    logits[np.arange(bs * N), fav] += rs.uniform(2, 6, bs * N).astype(np.float32)
    probs = torch.softmax(torch.from_numpy(logits), 1).numpy()
    deltas = (rs.standard_normal((bs * N, K, 4)) * 0.5).astype(np.float32)
    windows = np.array([[0, 0, 512, 512], [32, 0, 480, 512], [0, 64, 512, 448]], np.float32)
    T = lambda a: torch.from_numpy(a).to(DEV)
    feat = rs.standard_normal((bs * N, 16)).astype(np.float32)
    got, got_feat = L.detection_layer(T(rois), T(probs), T(deltas), T(windows), cfg, T(feat))
    exp = _detection_layer_restated(rois, probs, deltas, windows, cfg, oracle)
    got = got.cpu().numpy()
    n_exp = (exp[:, :, 4] > 0).sum(1)
    assert np.array_equal((got[:, :, 4] > 0).sum(1), n_exp) and n_exp.min() > 5
    assert np.array_equal(got[:, :, 4], exp[:, :, 4])                         # classes, in score order
    assert np.array_equal(got[:, :, 5], exp[:, :, 5])                         # scores: same elements of `probs`
    assert np.abs(got[:, :, :4] - exp[:, :, :4]).max() <= 1.0                 # rounded pixels; exp() differs by ulps host/device
    assert (np.abs(got[:, :, :4] - exp[:, :, :4]) > 0).mean() < 0.01
    # feature rows follow their detections
    ids = probs.argmax(1).reshape(bs, N)
    sc = probs.max(1).reshape(bs, N)
    for i in range(bs):
        for j in range(int(n_exp[i])):
            src = np.nonzero((sc[i] == got[i, j, 5]) & (ids[i] == got[i, j, 4]))[0]
            assert len(src) == 1 and np.array_equal(got_feat[i, j].cpu().numpy(), feat[i * N + src[0]])
    assert torch.all(got_feat[0, int(n_exp[0]):] == 0)


def test_inference_path_runs():
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    torch.manual_seed(1)
    cfg = _cfg(backbone="resnet50", image_size=256, batch_size=2)
    model = MaskRCNN(cfg).to(DEV)
    batch = synthetic_batch(2, 256, device=DEV)
    model.external_proposals = SyntheticProposals(batch[2], 256)
    windows = torch.tensor([[0, 0, 256, 256], [0, 16, 256, 240]], dtype=torch.float32)
    det, masks = model([batch[0], windows], mode='inference')
    assert det.shape == (2, 100, 6) and masks.shape == (2, 100, 81, 28, 28)
    n = (det[:, :, 4] > 0).sum(1)
    assert torch.all(n > 0)
    for b in range(2):
        d = det[b, :int(n[b])]
        assert torch.all(d[:-1, 5] >= d[1:, 5])                                # descending score
        assert torch.all((d[:, 0] >= windows[b, 0]) & (d[:, 2] <= windows[b, 2]) &
                         (d[:, 1] >= windows[b, 1]) & (d[:, 3] <= windows[b, 3]))
        assert torch.all(det[b, int(n[b]):] == 0)
    assert torch.isfinite(masks).all() and masks.min() >= 0 and masks.max() <= 1


def test_mask_head_on_positive_slots_is_exact():
    """MRCNN.MASK_HEAD_ON_POSITIVE_SLOTS skips only work whose results are never read: same loss
    terms, same gradients (up to the summation order of atomics)."""
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import compute_loss
    batch = synthetic_batch(2, 256, device=DEV)
    res = {}
    for flag in (False, True):
        torch.manual_seed(11)
        cfg = _cfg(backbone="resnet50", image_size=256, batch_size=2, train_rois_per_image=64, ot_L=5)
        cfg.MRCNN.MASK_HEAD_ON_POSITIVE_SLOTS = flag
        model = MaskRCNN(cfg).to(DEV)
        model.external_proposals = SyntheticProposals(batch[2], 256)
        model.generator = torch.Generator(device=DEV).manual_seed(3)
        loss, terms = compute_loss(model, list(batch), True, 1, None)
        loss.backward()
        res[flag] = (float(loss.detach()), {k: float(v) for k, v in terms.items()},
                     {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    assert abs(res[False][0] - res[True][0]) <= 1e-5 * abs(res[False][0])
    for k in res[False][1]:
        assert abs(res[False][1][k] - res[True][1][k]) <= 1e-5 * max(1.0, abs(res[False][1][k])), k
    assert res[False][2].keys() == res[True][2].keys()
    for n, g in res[False][2].items():
        d = (g - res[True][2][n]).abs().max().item()
        assert d <= 2e-4 * (g.abs().max().item() + 1e-6), (n, d)


@pytest.mark.parametrize("name,kw,size,bs", [
    ("configs[0]: R50, 2 x 512^2, 64 RoIs, OT off", dict(backbone="resnet50", image_size=512, batch_size=2,
                                                         train_rois_per_image=64, dev_switch=False), 512, 2),
    ("configs[1]: R50, 4 x 1024^2, 256 RoIs", dict(backbone="resnet50", image_size=1024, batch_size=4,
                                                   train_rois_per_image=256, ot_L=5), 1024, 4),
])
def test_baseline_configs_run(name, kw, size, bs):
    """The other BASELINE.json configurations as plain 'does a train step run and stay finite' cases
    (configs[2] is the bench workload; configs[0] exercises the non-intertwiner pyramid_roi_align path)."""
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import set_optimizer, train_step
    torch.manual_seed(2000)
    cfg = _cfg(**kw)
    model = MaskRCNN(cfg).to(DEV)
    opt = set_optimizer(model, cfg.TRAIN)
    batch = synthetic_batch(bs, size, device=DEV)
    model.external_proposals = SyntheticProposals(batch[2], size)
    model.generator = torch.Generator(device=DEV).manual_seed(3)
    first = last = None
    for _ in range(3):
        t = train_step(model, opt, list(batch), do_meta=cfg.DEV.SWITCH)
        assert all(torch.isfinite(v) for v in t.values()), (name, t)
        first = first if first is not None else float(t["total"])
        last = float(t["total"])
    assert last < first, (name, first, last)
    del model, opt
    torch.cuda.empty_cache()


def test_dev_stage_roi_pool_method_matches_oracle(oracle):
    """ROIS.METHOD = 'roi_pool' (lib/sub_module.py:515-577 with RoIPoolFunction instead of
    crop_and_resize): Dev's pooled 7x7 / 14x14 outputs vs the oracle's RoIPool per level, and one
    train step with that method."""
    from feature_intertwiner_amd.conv import conv_bn_act
    from feature_intertwiner_amd.sub_module import Dev
    cfg = _cfg(backbone="resnet50", image_size=256, batch_size=2, train_rois_per_image=48, dev_switch=True,
               roi_method="roi_pool")
    torch.manual_seed(0)
    dev = Dev(cfg, 256).to(DEV).eval()
    maps = [torch.randn(2, 256, s, s, device=DEV) for s in (64, 32, 16, 8)]
    rs = np.random.RandomState(3)
    rois_np = training_rois(rs, 2, 48)
    rois = torch.from_numpy(rois_np).to(DEV)
    cls = torch.randint(0, 81, (2, 48), device=DEV, dtype=torch.int32)
    with torch.no_grad():
        pooled, mask, _ = dev(maps, rois, cls)
        up = [conv_bn_act(m, dev.upsample[0][0], dev.upsample[0][1], relu=True).cpu().numpy() for m in maps]
    flat = rois_np.reshape(-1, 4)
    level = oracle.roi_level(flat, 256 * 256)
    ind = np.repeat(np.arange(2, dtype=np.float32), 48)
    pix = np.stack([ind, flat[:, 1] * 256, flat[:, 0] * 256, flat[:, 3] * 256, flat[:, 2] * 256], 1).astype(np.float32)
    for size, got in ((7, pooled), (14, mask)):
        exp = np.zeros((96, 256, size, size), np.float32)
        for l in range(2, 6):
            sel = np.nonzero(level == l)[0]
            if len(sel):
                exp[sel] = oracle.roi_pool_forward(up[l - 2], pix[sel], size, size, 1.0 / (4 << (l - 2)))[0]
        assert np.array_equal(got.cpu().numpy(), exp)

    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import set_optimizer, train_step
    torch.manual_seed(4)
    model = MaskRCNN(cfg).to(DEV)
    opt = set_optimizer(model, cfg.TRAIN)
    batch = synthetic_batch(2, 256, device=DEV)
    model.external_proposals = SyntheticProposals(batch[2], 256)
    t = train_step(model, opt, list(batch))
    assert all(torch.isfinite(v) for v in t.values()), t


@pytest.mark.parametrize("choice", ["l2", "l1", "kl"])
def test_meta_loss_choices_run(choice):
    """DEV.LOSS_CHOICE other than 'ot' (lib/model.py:197-204: mse / l1 / kl on sigmoid- resp.
    softmax-activated class features): a train step runs, the meta term is finite and non-negative."""
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import set_optimizer, train_step
    torch.manual_seed(5)
    cfg = _cfg(backbone="resnet50", image_size=256, batch_size=2, train_rois_per_image=64, loss_choice=choice)
    model = MaskRCNN(cfg).to(DEV)
    assert not hasattr(model, "ot_loss")
    opt = set_optimizer(model, cfg.TRAIN)
    batch = synthetic_batch(2, 256, device=DEV)
    model.external_proposals = SyntheticProposals(batch[2], 256)
    model.generator = torch.Generator(device=DEV).manual_seed(3)
    for _ in range(2):
        t = train_step(model, opt, list(batch))
        assert all(torch.isfinite(v) for v in t.values()), (choice, t)
        assert float(t["meta"]) >= 0.0


@pytest.mark.parametrize("tweak", ["BASELINE", "MULTI_UPSAMPLER", "UPSAMPLE_FAC2"])
def test_dev_config_branches_run(tweak):
    """Dev-stage switches of lib/config.py that change the graph: DEV.BASELINE (no intertwiner
    statistics), DEV.MULTI_UPSAMPLER (one make-up layer per level), DEV.UPSAMPLE_FAC = 2 (transposed
    3x3 conv as make-up layer, lib/sub_module.py:312-314)."""
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import set_optimizer, train_step
    torch.manual_seed(6)
    cfg = _cfg(backbone="resnet50", image_size=256, batch_size=2, train_rois_per_image=48, ot_L=5)
    if tweak == "BASELINE":
        cfg.DEV.BASELINE = True
    elif tweak == "MULTI_UPSAMPLER":
        cfg.DEV.MULTI_UPSAMPLER = True
    else:
        cfg.DEV.UPSAMPLE_FAC = 2.0
    model = MaskRCNN(cfg).to(DEV)
    opt = set_optimizer(model, cfg.TRAIN)
    batch = synthetic_batch(2, 256, device=DEV)
    model.external_proposals = SyntheticProposals(batch[2], 256)
    model.generator = torch.Generator(device=DEV).manual_seed(3)
    t = train_step(model, opt, list(batch), do_meta=not cfg.DEV.BASELINE)
    assert all(torch.isfinite(v) for v in t.values()), (tweak, t)


def test_fpn_ot_loss_branch_runs():
    """TRAIN.FPN_OT_LOSS (lib/sub_module.py:179-212: 2-D OptTrans between adjacent pyramid levels,
    off by default): one train step with it enabled."""
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import set_optimizer, train_step
    torch.manual_seed(7)
    cfg = _cfg(backbone="resnet50", image_size=128, batch_size=2, train_rois_per_image=32, ot_L=5)
    cfg.TRAIN.FPN_OT_LOSS = True
    model = MaskRCNN(cfg).to(DEV)
    opt = set_optimizer(model, cfg.TRAIN)
    batch = synthetic_batch(2, 128, device=DEV)
    model.external_proposals = SyntheticProposals(batch[2], 128)
    model.generator = torch.Generator(device=DEV).manual_seed(3)
    t = train_step(model, opt, list(batch))
    assert all(torch.isfinite(v) for v in t.values()), t


def test_train_step_edge_batches():
    """An image without any ground-truth object (all rows zero padding) and a history buffer longer
    than one step (DEV.BUFFER_SIZE > 1, lib/model.py:159-166): losses stay finite."""
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import set_optimizer, train_step
    torch.manual_seed(8)
    cfg = _cfg(backbone="resnet50", image_size=256, batch_size=2, train_rois_per_image=48, ot_L=5, buffer_size=3)
    model = MaskRCNN(cfg).to(DEV)
    opt = set_optimizer(model, cfg.TRAIN)
    images, gt_cls, gt_boxes, gt_masks = synthetic_batch(2, 256, device=DEV)
    hook = SyntheticProposals(gt_boxes.clone(), 256)
    gt_cls[1] = 0
    gt_boxes[1] = 0
    gt_masks[1] = 0
    model.external_proposals = hook
    model.generator = torch.Generator(device=DEV).manual_seed(3)
    for _ in range(4):
        t = train_step(model, opt, [images, gt_cls, gt_boxes, gt_masks])
        assert all(torch.isfinite(v) for v in t.values()), t
    assert model.feature_buffer.buffer.shape[0] == 3


def test_detection_layer_without_foreground():
    """Every RoI classified as background (or below DET_MIN_CONFIDENCE): all-zero detections, as the
    reference's early return (lib/layers.py:771-773)."""
    from feature_intertwiner_amd import layers as L
    cfg = _cfg(backbone="resnet50", image_size=256)
    rois = torch.rand(2, 50, 4, device=DEV).sort(2)[0][:, :, [0, 1, 2, 3]]
    probs = torch.zeros(100, 81, device=DEV)
    probs[:, 0] = 1.0
    deltas = torch.zeros(100, 81, 4, device=DEV)
    win = torch.tensor([[0, 0, 256, 256]] * 2, device=DEV, dtype=torch.float32)
    det = L.detection_layer(rois, probs, deltas, win, cfg)
    assert det.shape == (2, 100, 6) and torch.all(det == 0)
    cfg.TEST.DET_MIN_CONFIDENCE = 0.9
    probs = torch.full((100, 81), 0.5 / 80, device=DEV)
    probs[:, 3] = 0.5
    assert torch.all(L.detection_layer(rois, probs, deltas, win, cfg) == 0)


def test_rpn_targets_on_the_side_stream_equal_the_inline_call():
    """MaskRCNN.forward generates the RPN targets on a second stream while the backbone runs
    (_lib.run_on_side_stream): same values, same generator consumption as the inline call."""
    from feature_intertwiner_amd import _lib
    from feature_intertwiner_amd.config import make_config
    from feature_intertwiner_amd.layers import prepare_rpn_target
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import synthetic_batch
    torch.manual_seed(3)
    cfg = make_config("resnet50", 256, 2, 64, dev_switch=False)
    model = MaskRCNN(cfg).to(DEV)
    batch = synthetic_batch(2, 256, device=DEV, seed=9)
    g1 = torch.Generator(device=DEV).manual_seed(5)
    g2 = torch.Generator(device=DEV).manual_seed(5)
    with torch.no_grad():
        inline = prepare_rpn_target(model.priors, batch[1], batch[2], cfg, g1)
        # keep the main stream busy so that the side stream really overlaps something
        busy = torch.randn(4096, 4096, device=DEV)
        ready = _lib.run_on_side_stream(prepare_rpn_target, model.priors, batch[1], batch[2], cfg, g2)
        for _ in range(4):
            busy = busy @ busy * 1e-4
        side = ready()
    torch.cuda.synchronize()
    for a, b in zip(inline, side):
        assert torch.equal(a, b)
    assert torch.equal(g1.get_state(), g2.get_state())


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_detector_gradients_agree_between_backward_forms(precision):
    """The whole detector's backward in its round-3 form -- ReLU masks in the readers' data-gradient epilogues (conv.Gate),
    BatchNorm scale in W^T, BatchNorm sums from the weight gradient, FPN-lateral / shortcut gradients handed through
    GradBoxes, stem pooling and top-down upsampling on own kernels -- against the round-2 form (one fused
    elementwise/reduction pass per layer, autograd accumulating every multi-reader gradient).  The forward passes are
    the same kernels on the same inputs (the RPN's sampled anchors apart, see below), so every mask is identical and the
    two gradients may differ only by summation order and by where the BatchNorm scale is multiplied in: 2e-5 of each
    parameter's largest gradient element in fp32
    (the 16-bit kernels round the scale into different operands: the bar of test_gpu_conv_bf16's stage test)."""
    from feature_intertwiner_amd import conv as C
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import compute_loss
    cfg = _cfg(backbone="resnet50", image_size=256, batch_size=2, train_rois_per_image=64, ot_L=5,
               conv_precision=precision)
    grads, losses = {}, {}
    try:
        for form in ("round3", "round2"):
            C._UNSCALED_BACKWARD = C.GATES = form == "round3"
            torch.manual_seed(2000)
            model = MaskRCNN(cfg).to(DEV)
            batch = synthetic_batch(2, 256, device=DEV)
            model.external_proposals = SyntheticProposals(batch[2], 256)
            for step in range(2):                       # the second step runs on the cached folds / scaled W^T
                model.generator = torch.Generator(device=DEV).manual_seed(3)
                for p in model.parameters():
                    p.grad = None
                loss, terms = compute_loss(model, list(batch), True, 1, None)
                loss.backward()
            torch.cuda.synchronize()
            losses[form] = float(loss.detach())
            grads[form] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
            del model
    finally:
        C._UNSCALED_BACKWARD = C.GATES = True
        C.set_conv_precision("fp32")
        C.invalidate_step_state()
    # the same forward kernels on the same inputs, except the RPN's outputs at the sampled anchors, which the round-3 form
    # evaluates as matrix products on the anchors' 3 x 3 patches (RPN.forward_rows: another summation order)
    assert abs(losses["round3"] - losses["round2"]) <= (1e-6 if precision == "fp32" else 1e-5) * abs(losses["round2"])
    assert grads["round3"].keys() == grads["round2"].keys()
    bar = 2e-5 if precision == "fp32" else 6e-2
    gmax = max(float(g.abs().max()) for g in grads["round2"].values())
    worst = {}
    for n, ref in grads["round2"].items():
        diff = float((grads["round3"][n] - ref).abs().max())
        if precision != "fp32" and (n.startswith("ot_loss") or n.startswith("dev_roi.feat_extract")):
            # reached only through the 1-D cosine OT term, whose gradient is numerically degenerate (SURVEY Q6;
            # tests/test_gpu_data_parallel.py VARIANTS): ~0 next to the detector gradients in either form, and the
            # 16-bit forward is not bit-reproducible (atomic split-K in the head FCs)
            if diff > 1e-3 * gmax:
                worst[n] = diff / gmax
            continue
        err = diff / (float(ref.abs().max()) + 1e-20)
        if err > bar:
            worst[n] = err
    assert not worst, sorted(worst.items(), key=lambda kv: -kv[1])[:8]


def test_rpn_row_form_equals_the_dense_rpn_at_the_selected_anchors():
    """RPN.forward_rows (3 x 3 patches of the selected anchors' pixels -> conv.linear x 2) against the dense RPN read at
    the same anchors: outputs, and the gradients of every weight and of every pyramid level -- including a level whose
    gradient box already holds another reader's gradient, anchors on map borders and padding rows."""
    from feature_intertwiner_amd import conv as C
    from feature_intertwiner_amd.sub_module import RPN
    torch.manual_seed(11)
    rpn = RPN(3, 1, 256).to(DEV)
    shapes = [(2, 256, 16, 16), (2, 256, 8, 8), (2, 256, 4, 4)]
    maps = [torch.randn(s, device=DEV) for s in shapes]
    A = sum(s[2] * s[3] * 3 for s in shapes)
    R = 96
    g = torch.Generator(device="cpu").manual_seed(3)
    image = torch.randint(0, 2, (R,), generator=g).to(DEV)
    anchor = torch.randint(0, A, (R,), generator=g).to(DEV)
    anchor[:6] = torch.tensor([0, 2, 16 * 16 * 3 - 1, 16 * 16 * 3, A - 1, 15 * 3], device=DEV)      # corners / level starts
    valid = torch.ones(R, dtype=torch.bool, device=DEV)
    valid[-7:] = False
    image_p, anchor_p = image.clone(), anchor.clone()
    image_p[-7:] = -1
    anchor_p[-7:] = -1
    w_l, w_b = torch.randn(R, 2, device=DEV), torch.randn(R, 4, device=DEV)
    extra = torch.randn(shapes[1], device=DEV)                      # what another reader left for level 1

    def run(row_form):
        for p in rpn.parameters():
            p.grad = None
        ms = [m.clone().requires_grad_(True) for m in maps]
        if row_form:
            boxes = [None, C.GradBox(), None]
            boxes[1].taker = True
            boxes[1].value = extra.clone()
            logits, bbox = rpn.forward_rows(ms, image_p, anchor_p, valid, grad_boxes=boxes)
        else:
            outs = [rpn(m) for m in ms]
            lg = torch.cat([o[0] for o in outs], 1)
            bb = torch.cat([o[2] for o in outs], 1)
            logits = lg[image, anchor] * valid.unsqueeze(1)
            bbox = bb[image, anchor] * valid.unsqueeze(1)
        ((logits * w_l).sum() + (bbox * w_b).sum()).backward()
        gm = [m.grad.clone() for m in ms]
        if not row_form:
            gm[1] = gm[1] + extra
        return logits.detach(), bbox.detach(), gm, {n: p.grad.clone() for n, p in rpn.named_parameters()}

    dl, db, dgm, dgp = run(False)
    rl, rb, rgm, rgp = run(True)
    close = lambda a, b: float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-7
    assert close(rl, dl) and close(rb, db)
    assert torch.equal(rl[-7:], torch.zeros_like(rl[-7:]))
    for a, b in zip(rgm, dgm):
        assert close(a, b)
    for n in dgp:
        assert close(rgp[n], dgp[n]), n


@pytest.mark.parametrize("with_extra", [False, True])
def test_proposal_layer_with_a_pre_nms_limit_above_the_kernels_sort_capacity(oracle, with_extra):
    """RPN.PRE_NMS_LIMIT > 8192 (the reference's full sort has no bound, lib/layers.py:99-106): the candidates come from
    the tensor path (layers._proposal_candidates_tensors) -- against the oracle like the kernel's, and equal to the kernel
    where both apply (limit 6000 on the same inputs, tie rule included)."""
    from feature_intertwiner_amd import _lib
    from feature_intertwiner_amd import layers as L
    cfg = _cfg(backbone="resnet50", image_size=512)
    pri = _priors(cfg)
    g = torch.Generator(device=DEV).manual_seed(4)
    A = pri.size(0)
    probs = (torch.rand(2, A, 2, device=DEV, generator=g) * 64).round() / 64        # many ties
    bbox = torch.randn(2, A, 4, device=DEV, generator=g) * 0.5
    extra = None
    if with_extra:
        extra = torch.rand(2, 300, 5, device=DEV, generator=g) * 400
        extra[..., 2:4] += extra[..., 0:2]
        extra[..., 4] = (torch.rand(2, 300, device=DEV, generator=g) * 64).round() / 64
    out = {}
    for limit in (6000, 12000):
        cfg.RPN.PRE_NMS_LIMIT = limit
        taps = {}
        _lib.TAP = lambda name, **kw: taps.setdefault(name, kw)
        try:
            props, num = L.proposal_layer([probs, bbox], 1000, 0.7, pri, cfg, extra)
        finally:
            _lib.TAP = None
        tap = taps["proposal_candidates"]
        assert tap["dets"].shape == (2, limit, 5)
        _check_candidates(oracle, tap, cfg, 512.0)
        out[limit] = (tap["dets"], props, num)
    # the first 6000 candidates of the tensor path ARE the kernel's (same scores in the same order; boxes to 2 ulp)
    a, b = out[6000][0], out[12000][0][:, :6000]
    assert torch.equal(a[..., 4], b[..., 4])
    assert (a[..., :4] - b[..., :4]).abs().max().item() <= 3e-7 * 512 * 8
    for limit in (6000, 12000):
        dets, props, num = out[limit]
        for i in range(2):
            keep = oracle.pth_nms(dets[i].cpu().numpy(), 0.7)[:1000]
            assert int(num[i]) == len(keep)
