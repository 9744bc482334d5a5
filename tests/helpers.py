"""Seeded synthetic inputs shared by the tests, smoke() and bench.py."""
import numpy as np


def adversarial_boxes(rs, n, H, W):
    """Normalised (y1,x1,y2,x2) boxes that stress the bin assignment: inside, partly and
    fully outside, degenerate, flipped, grid-aligned, touching the last row/column."""
    f = np.float32
    boxes = []
    for i in range(n):
        k = i % 8
        if k == 0:      # ordinary box
            y1, x1 = rs.uniform(0, 0.8, 2)
            h, w = rs.uniform(0.02, 0.2, 2)
            b = [y1, x1, y1 + h, x1 + w]
        elif k == 1:    # straddles the border
            y1, x1 = rs.uniform(-0.2, 0.95, 2)
            b = [y1, x1, y1 + rs.uniform(0.1, 0.5), x1 + rs.uniform(0.1, 0.5)]
        elif k == 2:    # exactly on the pixel grid
            r0, c0 = rs.randint(0, H - 8), rs.randint(0, W - 8)
            r1, c1 = r0 + rs.randint(1, 8), c0 + rs.randint(1, 8)
            b = [r0 / (H - 1), c0 / (W - 1), r1 / (H - 1), c1 / (W - 1)]
        elif k == 3:    # touches the last row / column (in == H-1)
            b = [rs.uniform(0.5, 0.9), rs.uniform(0.5, 0.9), 1.0, 1.0]
        elif k == 4:    # degenerate (zero area)
            y, x = rs.uniform(0, 1, 2)
            b = [y, x, y, x]
        elif k == 5:    # flipped
            y1, x1 = rs.uniform(0.3, 0.9, 2)
            b = [y1, x1, y1 - rs.uniform(0.05, 0.3), x1 - rs.uniform(0.05, 0.3)]
        elif k == 6:    # completely outside
            b = [1.2, 1.3, 1.6, 1.7] if i % 16 == 6 else [-0.7, -0.6, -0.2, -0.1]
        else:           # whole image
            b = [0.0, 0.0, 1.0, 1.0]
        boxes.append(b)
    return np.asarray(boxes, f)


def clustered_dets(rs, n, size, n_clusters=20, pixel_round=False):
    """[n,5] = (y1,x1,y2,x2,score) proposals: jittered copies of a few 'objects' plus
    background, clipped to [0,size], unique scores sorted descending."""
    f = np.float32
    cy, cx = rs.uniform(0.1 * size, 0.9 * size, (2, n_clusters))
    side = np.exp(rs.uniform(np.log(16), np.log(size / 2), n_clusters))
    asp = np.exp(rs.uniform(np.log(0.5), np.log(2.0), n_clusters))
    n_fg = int(0.7 * n)
    which = rs.randint(0, n_clusters, n_fg)
    h = side[which] / np.sqrt(asp[which]) * np.exp(rs.normal(0, 0.15, n_fg))
    w = side[which] * np.sqrt(asp[which]) * np.exp(rs.normal(0, 0.15, n_fg))
    y = cy[which] + rs.normal(0, 0.1, n_fg) * h
    x = cx[which] + rs.normal(0, 0.1, n_fg) * w
    fg = np.stack([y - h / 2, x - w / 2, y + h / 2, x + w / 2], 1)
    n_bg = n - n_fg
    y1, x1 = rs.uniform(0, size * 0.9, (2, n_bg))
    hh, ww = np.exp(rs.uniform(np.log(8), np.log(size / 2), (2, n_bg)))
    bg = np.stack([y1, x1, y1 + hh, x1 + ww], 1)
    boxes = np.clip(np.concatenate([fg, bg], 0), 0, size)
    if pixel_round:
        boxes = np.round(boxes)
    scores = rs.permutation(n).astype(np.float64) / n + rs.uniform(0, 0.1 / n, n)
    order = np.argsort(-scores, kind="stable")
    dets = np.concatenate([boxes, scores[:, None]], 1)[order]
    return dets.astype(f)


def training_rois(rs, batch, per_image, n_gt=20):
    """Normalised (y1,x1,y2,x2) RoIs per image: half jittered GT copies, half background."""
    f = np.float32
    out = np.zeros((batch, per_image, 4), f)
    for b in range(batch):
        side = np.exp(rs.uniform(np.log(16 / 1024), np.log(0.5), n_gt))
        asp = np.exp(rs.uniform(np.log(0.5), np.log(2.0), n_gt))
        gh, gw = side / np.sqrt(asp), side * np.sqrt(asp)
        gy, gx = rs.uniform(0, 1 - gh), rs.uniform(0, 1 - gw)
        k = per_image // 2
        w = rs.randint(0, n_gt, k)
        jy = gy[w] + rs.uniform(-0.2, 0.2, k) * gh[w]
        jx = gx[w] + rs.uniform(-0.2, 0.2, k) * gw[w]
        jh = gh[w] * rs.uniform(0.8, 1.2, k)
        jw = gw[w] * rs.uniform(0.8, 1.2, k)
        fg = np.stack([jy, jx, jy + jh, jx + jw], 1)
        m = per_image - k
        s2 = np.exp(rs.uniform(np.log(16 / 1024), np.log(0.6), m))
        a2 = np.exp(rs.uniform(np.log(0.5), np.log(2.0), m))
        bh, bw = s2 / np.sqrt(a2), s2 * np.sqrt(a2)
        by, bx = rs.uniform(0, 1, m) * (1 - np.minimum(bh, 1)), rs.uniform(0, 1, m) * (1 - np.minimum(bw, 1))
        bg = np.stack([by, bx, by + bh, bx + bw], 1)
        out[b] = np.clip(np.concatenate([fg, bg], 0), 0, 1)
    return out


def filled_state(named_shapes, seed):
    """Deterministic values for a state dict, by position: numpy legacy generator (stable across
    versions).  Weights ~ N(0, 1/fan_in), biases small, BN statistics positive."""
    out = {}
    for i, (name, shape) in enumerate(named_shapes):
        rs = np.random.RandomState(seed + i)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = np.zeros(shape, np.int64)
        elif leaf == "running_var":
            out[name] = (1.0 + 0.2 * np.abs(rs.standard_normal(shape))).astype(np.float32)
        elif leaf == "running_mean":
            out[name] = (0.1 * rs.standard_normal(shape)).astype(np.float32)
        elif len(shape) == 1 and leaf == "weight":                       # BN gamma
            out[name] = (1.0 + 0.1 * rs.standard_normal(shape)).astype(np.float32)
        elif leaf == "bias":
            out[name] = (0.05 * rs.standard_normal(shape)).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
            out[name] = (rs.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
    return out



def golden_loss_inputs(seed=13):
    """Seeded inputs for the five detector losses (shared by oracle/gen_golden_layers.py and the
    tests, so the fixture only stores the reference's outputs)."""
    rs = np.random.RandomState(seed)
    B, A, R, NC = 2, 300, 24, 81
    match = rs.choice([-1, 0, 1], size=(B, A), p=[0.3, 0.5, 0.2]).astype(np.int64)
    rpn_logits = rs.standard_normal((B, A, 2)).astype(np.float32)
    rpn_bbox = rs.standard_normal((B, A, 4)).astype(np.float32)
    tgt_rpn_bbox = np.zeros((B, 256, 4), np.float32)
    for b in range(B):
        n = int((match[b] == 1).sum())
        tgt_rpn_bbox[b, :n] = rs.standard_normal((n, 4))
    cls_ids = np.zeros((B, R), np.int64)
    cls_ids[0, :7] = rs.randint(1, NC, 7)
    cls_ids[1, :3] = rs.randint(1, NC, 3)
    cls_logits = rs.standard_normal((B, R, NC)).astype(np.float32)
    tgt_bbox = (rs.standard_normal((B, R, 4)) * (cls_ids > 0)[..., None]).astype(np.float32)
    pred_bbox = rs.standard_normal((B, R, NC, 4)).astype(np.float32)
    tgt_masks = ((rs.uniform(size=(B, R, 28, 28)) > 0.5) * (cls_ids > 0)[..., None, None]).astype(np.float32)
    pred_masks = rs.uniform(0.02, 0.98, (B, R, NC, 28, 28)).astype(np.float32)
    return dict(rpn_match=match, rpn_logits=rpn_logits, rpn_bbox_pred=rpn_bbox, rpn_bbox_target=tgt_rpn_bbox,
                cls_ids=cls_ids, cls_logits=cls_logits, bbox_target=tgt_bbox, bbox_pred=pred_bbox,
                mask_target=tgt_masks, mask_pred=pred_masks)


def golden_module_inputs(seed=7):
    """Inputs of tests/golden/modules_r50_128.npz: image, 7x7 pooled RoIs, 14x14 pooled RoIs."""
    rs = np.random.RandomState(seed)
    image = rs.standard_normal((1, 3, 128, 128)).astype(np.float32)
    pooled = np.maximum(rs.standard_normal((6, 256, 7, 7)), 0).astype(np.float32)
    pooled14 = np.maximum(rs.standard_normal((3, 256, 14, 14)), 0).astype(np.float32)
    return image, pooled, pooled14


GOLDEN_MODULE_SEEDS = dict(fpn=100, rpn=2000, classifier=3000, mask=4000, dev=5000)


def golden_meta_inputs(step, K=11, F=1024, G=2, S=3, seed=31, activation="relu"):
    """Seeded (big_feat, big_cnt, small_feat, small_cnt) stacks [G,S,F,K] / [G,S,1,K] of one training
    step, as Dev.forward would return them (class means are 0 where the count is 0; background column
    empty), shared by oracle/gen_golden_meta.py and the tests.  `activation`: 'relu' (OT features),
    'sigmoid' (l1/l2) or 'softmax' (kl, over the feature axis).  Step 2 has NO small objects at all
    (the reference then skips meta_loss and leaves the buffer alone, lib/workflow.py:190-194)."""
    rs = np.random.RandomState(seed + 17 * step)

    def stack(p_present):
        cnt = (rs.randint(1, 6, (G, S, 1, K)) * (rs.uniform(size=(G, S, 1, K)) < p_present)).astype(np.float32)
        cnt[..., 0] = 0
        raw = rs.standard_normal((G, S, F, K)).astype(np.float32)
        if activation == "relu":
            f = np.maximum(raw, 0)
        elif activation == "sigmoid":
            f = 1.0 / (1.0 + np.exp(-raw))
        else:
            e = np.exp(raw - raw.max(2, keepdims=True))
            f = e / e.sum(2, keepdims=True)
        return (f * (cnt > 0)).astype(np.float32), cnt

    bf, bc = stack(0.5)
    sf, sc = stack(0.0 if step == 2 else 0.45)
    return bf, bc, sf, sc


def golden_meta_instances(step, n=48, K=11, F=1024, seed=77, activation="sigmoid"):
    """Seeded (small_output_all [n,F], small_gt_all [n]) for the DEV.INST_LOSS branch; rows past the
    small boxes are zero with class 0, as Dev.forward leaves them."""
    rs = np.random.RandomState(seed + 13 * step)
    raw = rs.standard_normal((n, F)).astype(np.float32)
    out = np.maximum(raw, 0) if activation == "relu" else (1.0 / (1.0 + np.exp(-raw))).astype(np.float32)
    gt = rs.randint(0, K, n).astype(np.int64)
    gt[n - 10:] = 0
    out[n - 10:] = 0
    return out.astype(np.float32), gt


def ot_full_weights(seed, ch=1024):
    """Deterministic weights for a full-size 1-D OptTrans (G_net ch->ch, critic ch->ch/4, k=3):
    same recipe as oracle/gen_golden_ot.full_weights."""
    rs = np.random.RandomState(seed)
    g_w = (rs.standard_normal((ch, ch, 3)) * (1.0 / np.sqrt(3 * ch))).astype(np.float32)
    g_b = (rs.standard_normal((ch,)) * 0.05).astype(np.float32)
    c_w = (rs.standard_normal((ch // 4, ch, 3)) * (1.0 / np.sqrt(3 * ch))).astype(np.float32)
    c_b = (rs.standard_normal((ch // 4,)) * 0.05).astype(np.float32)
    return g_w, g_b, c_w, c_b


def keras_named_arrays(arch="resnet101", num_classes=81, seed=5, small=True):
    """A Keras (matterport Mask R-CNN layout) weight dictionary: {'<layer>.<weight>:0': array} with conv
    kernels HWIO, transposed-conv kernels HWOI, dense kernels (in, out) -- the input of
    tools/convert_from_keras.py.  small=True divides every channel count by 16 (the name mapping and the
    transposes are what is under test, not 250 MB of values).  Shared by oracle/gen_golden_keras.py and
    tests/test_checkpoint_formats.py."""
    rs = np.random.RandomState(seed)
    d = 16 if small else 1
    ch = lambda c: max(c // d, 1)
    out = {}

    def conv(name, kh, kw, cin, cout, bias=True):
        out[name + ".kernel:0"] = rs.standard_normal((kh, kw, cin, cout)).astype(np.float32)
        if bias:
            out[name + ".bias:0"] = rs.standard_normal((cout,)).astype(np.float32)

    def bn(name, c):
        for w in ("gamma:0", "beta:0", "moving_mean:0", "moving_variance:0"):
            out[name + "." + w] = rs.standard_normal((c,)).astype(np.float32)

    conv("conv1", 7, 7, 3, ch(64))
    bn("bn_conv1", ch(64))
    blocks = {"resnet50": [3, 4, 6, 3], "resnet101": [3, 4, 23, 3]}[arch]
    inpl = ch(64)
    for stage, (nb, planes) in enumerate(zip(blocks, (64, 128, 256, 512)), start=2):
        p = ch(planes)
        for b in range(nb):
            blk = "abcdefghijklmnopqrstuvwxyz"[b]
            conv("res%d%s_branch2a" % (stage, blk), 1, 1, inpl, p)
            bn("bn%d%s_branch2a" % (stage, blk), p)
            conv("res%d%s_branch2b" % (stage, blk), 3, 3, p, p)
            bn("bn%d%s_branch2b" % (stage, blk), p)
            conv("res%d%s_branch2c" % (stage, blk), 1, 1, p, 4 * p)
            bn("bn%d%s_branch2c" % (stage, blk), 4 * p)
            if b == 0:
                conv("res%d%s_branch1" % (stage, blk), 1, 1, inpl, 4 * p)
                bn("bn%d%s_branch1" % (stage, blk), 4 * p)
            inpl = 4 * p
    f = ch(256)
    for lvl, cin in zip((2, 3, 4, 5), (256, 512, 1024, 2048)):
        conv("fpn_c%dp%d" % (lvl, lvl), 1, 1, ch(cin), f)
        conv("fpn_p%d" % lvl, 3, 3, f, f)
    conv("rpn_conv_shared", 3, 3, f, ch(512))
    conv("rpn_class_raw", 1, 1, ch(512), 6)
    conv("rpn_bbox_pred", 1, 1, ch(512), 12)
    conv("mrcnn_class_conv1", 7, 7, f, ch(1024))
    bn("mrcnn_class_bn1", ch(1024))
    conv("mrcnn_class_conv2", 1, 1, ch(1024), ch(1024))
    bn("mrcnn_class_bn2", ch(1024))
    out["mrcnn_class_logits.kernel:0"] = rs.standard_normal((ch(1024), num_classes)).astype(np.float32)
    out["mrcnn_class_logits.bias:0"] = rs.standard_normal((num_classes,)).astype(np.float32)
    out["mrcnn_bbox_fc.kernel:0"] = rs.standard_normal((ch(1024), num_classes * 4)).astype(np.float32)
    out["mrcnn_bbox_fc.bias:0"] = rs.standard_normal((num_classes * 4,)).astype(np.float32)
    for i in range(1, 5):
        conv("mrcnn_mask_conv%d" % i, 3, 3, f, f)
        bn("mrcnn_mask_bn%d" % i, f)
    out["mrcnn_mask_deconv.kernel:0"] = rs.standard_normal((2, 2, f, f)).astype(np.float32)     # (kh, kw, out, in)
    out["mrcnn_mask_deconv.bias:0"] = rs.standard_normal((f,)).astype(np.float32)
    conv("mrcnn_mask", 1, 1, f, num_classes)
    return out


def golden_wrapper_inputs(seed=23):
    """Seeded inputs of tests/golden/wrappers.npz (oracle/gen_golden_wrappers.py): pixel boxes for RoIAlign.forward's box
    transform, score-carrying boxes for pth_nms / nms, and a small pyramid with normalised RoIs for pyramid_roi_align
    (incl. RoIs exactly on a level boundary's far sides, tiny and whole-image ones).  Shared with the tests, so the
    fixture holds the reference's outputs only."""
    rs = np.random.RandomState(seed)
    f = np.float32
    out = {}
    # (a) RoIAlign pixel boxes (x1, y1, x2, y2) on a 50 x 68 map, some outside / degenerate
    x1y1 = rs.uniform(-4, 60, (96, 2))
    wh = np.exp(rs.uniform(np.log(0.5), np.log(40), (96, 2)))
    px = np.concatenate([x1y1, x1y1 + wh], 1).astype(f)
    px[::13, 2:] = px[::13, :2]                      # zero-area
    px[5] = [0, 0, 67, 49]                           # the whole map
    out["roialign_boxes_px"] = px
    out["roialign_box_ind"] = rs.randint(0, 2, 96).astype(np.int32)
    # (b) NMS: 3 images x 400 clustered boxes (y1, x1, y2, x2, score), scores unique but NOT pre-sorted
    dets = np.stack([clustered_dets(rs, 400, 256, n_clusters=6, pixel_round=(i == 1)) for i in range(3)])
    for i in range(3):
        dets[i] = dets[i][rs.permutation(400)]
    out["nms_dets"] = dets.astype(f)
    # (c) pyramid: 2 images, 4 channels, maps 64/32/16/8, 60 RoIs per image; the level formula reads the IMAGE shape
    # only (1024 x 1024 here), the crops read the maps
    maps = [rs.standard_normal((2, 4, s, s)).astype(f) for s in (64, 32, 16, 8)]
    rois = training_rois(rs, 2, 60, n_gt=6)
    side = np.array([8, 40, 56, 112, 224, 448, 1023]) / 1024.0   # spans every level incl. both clamps
    for j, s in enumerate(side):
        rois[0, j] = [0.0, 0.0, s, s]
        rois[1, j] = np.clip([0.3, 0.2, 0.3 + s / 2, 0.2 + 2 * s], 0, 1)
    # (d) detection layer, per sample: integer pixel boxes, top class and its score per RoI (conduct_nms's inputs)
    det = []
    for n_roi, n_cls, lo in ((400, 6, 0.2), (120, 8, 0.45)):           # > 100 survivors / fewer than 100 (every class keeps >= 2 boxes:
                                                                        # the reference indexes a 1-element class as a scalar and fails)
        cls = rs.randint(0, n_cls, n_roi).astype(np.int64)
        y1x1 = rs.randint(0, 200, (n_roi, 2))
        hw = rs.randint(1, 56, (n_roi, 2))
        hw[::17] = 0                                                    # zero-area boxes (filtered)
        boxes = np.concatenate([y1x1, y1x1 + hw], 1).astype(f)
        scores = (lo + (1 - lo) * (rs.permutation(n_roi) + 0.5) / n_roi).astype(f)      # unique
        det.append((cls, boxes, scores))
    out["det_samples"] = det
    out["pyr_maps"] = maps
    out["pyr_rois"] = rois.astype(f)
    out["pyr_image_shape"] = (1024, 1024, 3)
    return out


def golden_crop_cases():
    """Seeded inputs for tests/golden/crop_fwd.npz (oracle/gen_golden_crop.py runs the reference's own
    `CropAndResizePerBox` on them; the tests regenerate the same inputs and compare bit for bit).
    Yields (name, keep_full_output, image[B,C,H,W], boxes[N,4], box_ind[N], crop_h, crop_w, extrapolation)."""
    small_crops = ((7, 7), (1, 1), (5, 3), (1, 9), (3, 11), (14, 14))
    all_crops = small_crops + ((28, 28), (64, 64))
    shapes = ((2, 4, 64, 64), (3, 5, 37, 91), (1, 6, 16, 16), (1, 3, 2, 2), (2, 3, 1, 5))
    for si, shape in enumerate(shapes):
        B, C, H, W = shape
        for ci, (ch, cw) in enumerate(all_crops):
            rs = np.random.RandomState(7000 + 31 * si + ci)
            image = rs.standard_normal(shape).astype(np.float32)
            for extrap in (0.0, -3.5):
                keep = (ch, cw) in small_crops and extrap == 0.0
                n = 48 if keep else 203
                boxes = adversarial_boxes(rs, n, max(H, 9), max(W, 9))
                ind = rs.randint(0, B, n).astype(np.int32)
                yield ("s%d_c%dx%d_e%s" % (si, ch, cw, "0" if extrap == 0.0 else "m"), keep,
                       image, boxes, ind, ch, cw, extrap)
    # the shapes of tests/test_gpu_crop.py::test_forward_bit_exact_adversarial (70 channels: odd channel chunks)
    rs = np.random.RandomState(7900)
    image = rs.standard_normal((1, 70, 16, 16)).astype(np.float32)
    boxes = adversarial_boxes(rs, 203, 16, 16)
    yield ("c70_7x7", False, image, boxes, np.zeros(203, np.int32), 7, 7, 0.0)
    # north star: 512 RoIs x 256 channels x 7x7 (and 14x14) from a [2,256,256,256] map
    rs = np.random.RandomState(2000)
    image = rs.standard_normal((2, 256, 256, 256)).astype(np.float32)
    rois = training_rois(rs, 2, 256).reshape(-1, 4)
    ind = np.repeat(np.arange(2, dtype=np.int32), 256)
    for crop in (7, 14):
        yield ("northstar_%dx%d" % (crop, crop), False, image, rois, ind, crop, crop, 0.0)
    adv = adversarial_boxes(np.random.RandomState(2001), 512, 256, 256)
    yield ("northstar_adversarial_7x7", False, image, adv, ind, 7, 7, 0.0)
