"""NumPy front-end to the CPU oracle (oracle/fi_oracle.c) plus restatements of the
Python-level wrappers of the reference hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.

Reference citations are relative to the reference checkout (lib/...).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfi_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile oracle/fi_oracle.c with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "fi_oracle.c")
    if force or (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.orc_crop_and_resize_forward.restype = ctypes.c_int
        L.orc_crop_and_resize_backward.restype = ctypes.c_int
        L.orc_crop_taps.restype = ctypes.c_int
        L.orc_roi_pool_forward.restype = ctypes.c_int
        L.orc_roi_pool_backward.restype = ctypes.c_int
        L.orc_nms.restype = ctypes.c_long
        L.orc_sinkhorn.restype = ctypes.c_float
        L.orc_class_mean.restype = ctypes.c_int
        L.orc_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(t)


def num_threads():
    return int(lib().orc_num_threads())


# ----------------------------------------------------------------------------
# crop_and_resize (lib/roi_align/src/crop_and_resize.c)
# ----------------------------------------------------------------------------
def crop_and_resize_forward(image, boxes, box_ind, crop_h, crop_w, extrapolation_value=0.0):
    image, boxes, box_ind = _f32(image), _f32(boxes), _i32(box_ind)
    B, C, H, W = image.shape
    N = boxes.shape[0]
    crops = np.empty((N, C, crop_h, crop_w), np.float32)
    rc = lib().orc_crop_and_resize_forward(
        _p(image, _f32p), B, C, H, W, _p(boxes, _f32p), _p(box_ind, _i32p), N,
        int(crop_h), int(crop_w), ctypes.c_float(extrapolation_value), _p(crops, _f32p))
    if rc != 0:
        raise RuntimeError("oracle crop_and_resize_forward: status %d" % rc)
    return crops


def crop_and_resize_backward(grads, boxes, box_ind, image_shape):
    grads, boxes, box_ind = _f32(grads), _f32(boxes), _i32(box_ind)
    B, C, H, W = image_shape
    N, C2, ch, cw = grads.shape
    assert C2 == C
    out = np.empty((B, C, H, W), np.float32)
    rc = lib().orc_crop_and_resize_backward(
        _p(grads, _f32p), _p(boxes, _f32p), _p(box_ind, _i32p), N, B, C, H, W, ch, cw,
        _p(out, _f32p))
    if rc != 0:
        raise RuntimeError("oracle crop_and_resize_backward: status %d" % rc)
    return out


def crop_taps(boxes, H, W, crop_h, crop_w):
    """Bin assignment only: dict of [N,crop] arrays per axis."""
    boxes = _f32(boxes)
    N = boxes.shape[0]
    yv, y0, y1 = (np.empty((N, crop_h), np.int32) for _ in range(3))
    xv, x0, x1 = (np.empty((N, crop_w), np.int32) for _ in range(3))
    yf = np.empty((N, crop_h), np.float32)
    xf = np.empty((N, crop_w), np.float32)
    rc = lib().orc_crop_taps(_p(boxes, _f32p), N, H, W, crop_h, crop_w,
                             _p(yv, _i32p), _p(y0, _i32p), _p(y1, _i32p), _p(yf, _f32p),
                             _p(xv, _i32p), _p(x0, _i32p), _p(x1, _i32p), _p(xf, _f32p))
    assert rc == 0
    return dict(y_valid=yv, y0=y0, y1=y1, y_frac=yf, x_valid=xv, x0=x0, x1=x1, x_frac=xf)


def roi_align_boxes(boxes_xyxy, H, W, crop_h, crop_w, transform_fpcoor=True):
    """Pixel (x1,y1,x2,y2) -> normalised (y1,x1,y2,x2) exactly as RoIAlign.forward,
    lib/roi_align/roi_align.py:26-45 (fp32 elementwise arithmetic in that order)."""
    b = _f32(boxes_xyxy)
    x1, y1, x2, y2 = (b[:, i:i + 1] for i in range(4))
    f = np.float32
    if transform_fpcoor:
        sw = (x2 - x1) / f(crop_w)
        sh = (y2 - y1) / f(crop_h)
        nx0 = (x1 + sw / f(2) - f(0.5)) / f(W - 1)
        ny0 = (y1 + sh / f(2) - f(0.5)) / f(H - 1)
        nw = sw * f(crop_w - 1) / f(W - 1)
        nh = sh * f(crop_h - 1) / f(H - 1)
        out = np.concatenate((ny0, nx0, ny0 + nh, nx0 + nw), 1)
    else:
        out = np.concatenate((y1 / f(H - 1), x1 / f(W - 1), y2 / f(H - 1), x2 / f(W - 1)), 1)
    return out.astype(np.float32)


# ----------------------------------------------------------------------------
# RoIPool (lib/roi_pooling/src/roi_pooling_kernel.cu)
# ----------------------------------------------------------------------------
def roi_pool_forward(features, rois, ph, pw, scale):
    features, rois = _f32(features), _f32(rois)
    B, C, H, W = features.shape
    N = rois.shape[0]
    assert rois.shape[1] == 5
    out = np.empty((N, C, ph, pw), np.float32)
    arg = np.empty((N, C, ph, pw), np.int32)
    rc = lib().orc_roi_pool_forward(_p(features, _f32p), B, C, H, W, _p(rois, _f32p), N,
                                    ph, pw, ctypes.c_float(scale), _p(out, _f32p), _p(arg, _i32p))
    assert rc == 0
    return out, arg


def roi_pool_backward(top_grad, argmax, rois, feature_shape, scale):
    top_grad, rois, argmax = _f32(top_grad), _f32(rois), _i32(argmax)
    B, C, H, W = feature_shape
    N, _, ph, pw = top_grad.shape
    out = np.empty((B, C, H, W), np.float32)
    rc = lib().orc_roi_pool_backward(_p(top_grad, _f32p), _p(argmax, _i32p), _p(rois, _f32p), N,
                                     B, C, H, W, ph, pw, ctypes.c_float(scale), _p(out, _f32p))
    assert rc == 0
    return out


# ----------------------------------------------------------------------------
# NMS (lib/nms/src/nms.c, lib/nms/pth_nms.py, lib/nms/nms_wrapper.py)
# ----------------------------------------------------------------------------
def nms_core(boxes, order, areas, thresh, strict=False):
    boxes = _f32(boxes)
    order = np.ascontiguousarray(order, dtype=np.int64)
    areas = _f32(areas)
    N, dim = boxes.shape
    keep = np.empty((N,), np.int64)
    n = lib().orc_nms(_p(boxes, _f32p), N, dim, _p(order, _i64p), _p(areas, _f32p),
                      ctypes.c_float(thresh), int(bool(strict)), _p(keep, _i64p))
    return keep[:n].copy()


def pth_nms(dets, thresh, strict=False):
    """dets [N,5] = (y1,x1,y2,x2,score): lib/nms/pth_nms.py:5-19 (CPU branch): areas
    in fp32 with the +1 convention, order = scores sorted descending, then cpu_nms.
    A stable descending sort is used; both reference call sites pass unique,
    pre-sorted scores (lib/layers.py:103-105, 690-691)."""
    dets = _f32(dets)
    x1, y1, x2, y2 = dets[:, 1], dets[:, 0], dets[:, 3], dets[:, 2]
    one = np.float32(1)
    areas = (x2 - x1 + one) * (y2 - y1 + one)
    order = np.argsort(-dets[:, 4], kind="stable")
    return nms_core(dets, order, areas, thresh, strict)


def nms(dets, thresh, strict=False):
    """Batched wrapper, lib/nms/nms_wrapper.py:14-34: per image, truncate every
    image to the shortest keep list, int32 [bs, min_keep]."""
    keeps = [pth_nms(d, thresh, strict) for d in dets]
    m = min(len(k) for k in keeps) if keeps else 0
    out = np.zeros((len(keeps), m), np.int32)
    for i, k in enumerate(keeps):
        out[i] = k[:m]
    return out


# ----------------------------------------------------------------------------
# OT intertwiner loss (lib/OT_module.py)
# ----------------------------------------------------------------------------
def sinkhorn(x, y, eps_inv=1.0, L=5, c_form="cosine", return_plan=False):
    x, y = _f32(x), _f32(y)
    S, D = x.shape
    assert y.shape == (S, D)
    plan = np.empty((S, S), np.float32) if return_plan else None
    v = lib().orc_sinkhorn(_p(x, _f32p), _p(y, _f32p), S, D, ctypes.c_float(eps_inv), int(L),
                           1 if c_form == "l2" else 0,
                           _p(plan, _f32p) if return_plan else None)
    return (np.float32(v), plan) if return_plan else np.float32(v)


def conv1d_same_k3(x, weight, bias):
    """nn.Conv1d(k=3, padding=1, stride=1) on x [n, Cin, Lx] (lib/OT_module.py:37-41,
    58-63).  float64 accumulation, fp32 result."""
    x = _f32(x)
    n, cin, lx = x.shape
    xp = np.zeros((n, cin, lx + 2), np.float64)
    xp[:, :, 1:-1] = x
    w = weight.astype(np.float64)
    out = np.zeros((n, w.shape[0], lx), np.float64)
    for k in range(3):
        out += np.einsum("oc,ncl->nol", w[:, :, k], xp[:, :, k:k + lx])
    out += bias.astype(np.float64)[None, :, None]
    return out.astype(np.float32)


def opttrans_1d_forward(x, y, g_w, g_b, c_w, c_b, epsilon=1.0, L=5, c_form="cosine",
                        remove_bias=False, return_terms=False):
    """OptTrans.forward for the 1-D ('conv') form, lib/OT_module.py:67-102:
    x_up = relu(G_net(x)); T(p,q) = [sinkhorn(critic(p)[i], critic(q)[i]) for i];
    loss = 2 T(x_up,y) - T(x_up,x_up) - T(y,y)  (or T(x_up,y) if remove_bias)."""
    relu = lambda t: np.maximum(t, np.float32(0))
    x_up = relu(conv1d_same_k3(x, g_w, g_b))

    def term(p, q):
        cp = relu(conv1d_same_k3(p, c_w, c_b))   # [n, ch/4, L]  -> samples = channels
        cq = relu(conv1d_same_k3(q, c_w, c_b))
        return np.array([sinkhorn(cp[i], cq[i], 1.0 / epsilon, L, c_form) for i in range(cp.shape[0])],
                        np.float32)

    t_xy = term(x_up, y)
    if remove_bias:
        return (t_xy, (t_xy,)) if return_terms else t_xy
    t_xx = term(x_up, x_up)
    t_yy = term(_f32(y), _f32(y))
    loss = np.float32(2) * t_xy - t_xx - t_yy
    return (loss, (t_xy, t_xx, t_yy)) if return_terms else loss


def class_mean(features, gt, num_classes):
    features, gt = _f32(features), _i32(gt)
    N, F = features.shape
    feat = np.empty((F, num_classes), np.float32)
    cnt = np.empty((num_classes,), np.float32)
    rc = lib().orc_class_mean(_p(features, _f32p), _p(gt, _i32p), N, F, num_classes,
                              _p(feat, _f32p), _p(cnt, _f32p))
    assert rc == 0
    return feat, cnt.reshape(1, num_classes)


def roi_level(rois, image_area, base=224.0):
    """Pyramid level assignment, lib/sub_module.py:396-410 == lib/layers.py:168-181:
    clamp(round(4 + log2(sqrt(h*w) / (base / sqrt(image_area)))), 2, 5) in fp32.
    torch.round is round-half-to-even (as np.rint)."""
    r = _f32(rois)
    h = r[..., 2] - r[..., 0]
    w = r[..., 3] - r[..., 1]
    f = np.float32
    ratio = np.sqrt(h * w) / (f(base) / np.sqrt(f(image_area)))
    with np.errstate(divide="ignore"):
        lvl = f(4) + (np.log(ratio) / np.log(f(2))).astype(np.float32)
    lvl = np.rint(lvl)
    lvl = np.where(np.isfinite(lvl), lvl, -1e9)
    return np.clip(lvl, 2, 5).astype(np.int32)


def dev_stage_groups(level, gt=None):
    """Which RoIs the intertwiner stage feeds where, per pyramid level, as the reference's loop over levels 2..5 builds it
    (lib/sub_module.py:437-598, structure 'beta'): level ell's SMALL boxes are the RoIs with roi_level == ell, in RoI
    order (small_ix, :440-454 with DEV.ASSIGN_BOX_ON_ALL_SCALE off); its BIG boxes are those of the levels above,
    _find_big_box2 (:366-378: 2 -> {3,4,5}, 3 -> {4,5}, 4 -> {5}, 5 -> none), and they contribute statistics only when
    the level has small boxes at all (:456-467).  Statistics exist for levels 2..4 (:434-435).  The small rows are
    written level-major into small_output_all / small_gt_all (:583-598).
    level [N] int (2..5), gt [N] class ids or None.  Returns dict(order = RoI indices in level-major order (levels 2..5),
    small = {ell: indices}, big = {ell: indices} for ell in 2..4 (empty when the level has no small box),
    small_gt_all [N] float (class id of the small rows of levels 2..4 in level-major order, 0 behind them))."""
    level = np.asarray(level).astype(np.int64).reshape(-1)
    N = level.size
    gt = np.zeros(N, np.int64) if gt is None else np.asarray(gt).astype(np.int64).reshape(-1)
    small, big, order = {}, {}, []
    small_gt_all = np.zeros(N, np.float32)
    pos = 0
    for ell in (2, 3, 4, 5):
        idx = np.nonzero(level == ell)[0]
        order.append(idx)
        if ell <= 4:
            small[ell] = idx
            above = np.nonzero(level > ell)[0]
            big[ell] = above if idx.size else above[:0]
            small_gt_all[pos:pos + idx.size] = gt[idx]
            pos += idx.size
    return {"order": np.concatenate(order), "small": small, "big": big, "small_gt_all": small_gt_all}


# ----------------------------------------------------------------------------
# Intertwiner meta loss + history buffer (lib/model.py:143-224, lib/workflow.py:190-203)
# ----------------------------------------------------------------------------
META_EPS = np.float32(1e-20)


def merge_feat_vec(box_feat, box_cnt):
    """MaskRCNN._merge_feat_vec, lib/model.py:217-224: count-weighted mean over (gpu, scale)."""
    s = (_f32(box_feat) * _f32(box_cnt)).sum(0).sum(0)
    c = _f32(box_cnt).sum(0).sum(0)
    return (s / (c + META_EPS)).astype(np.float32), c.astype(np.float32)


class MetaLoss(object):
    """Step-by-step restatement of MaskRCNN.meta_loss with its buffer state.

    choice in {'l2','l1','kl','ot'}; `ot` = dict(g_w, g_b, c_w, c_b, epsilon, L) of the 1-D OptTrans.
    __call__ returns the reference's return value: a scalar for l2/l1/kl, the per-row vector for
    'ot' (:207), and 0 when nothing is selected or (lib/workflow.py:190-194) when the step has no
    small-object statistics -- in which case the buffer is NOT updated."""

    def __init__(self, choice, buffer_size, feat_dim, num_classes, inst_loss=False, ot=None):
        self.choice, self.inst, self.ot = choice, inst_loss, ot
        self.buffer = np.zeros((buffer_size, feat_dim, num_classes), np.float32)
        self.buffer_cnt = np.zeros((buffer_size, 1, num_classes), np.float32)

    def _update(self, big_feat, big_cnt):
        bf, bc = merge_feat_vec(big_feat, big_cnt)
        if self.buffer.shape[0] == 1:                                         # :153-158 running mean
            feat_sum = self.buffer * self.buffer_cnt + bf[None] * bc[None]
            self.buffer_cnt = self.buffer_cnt + bc[None]
            self.buffer = (feat_sum / (self.buffer_cnt + META_EPS)).astype(np.float32)
            return self.buffer[0]
        self.buffer[:-1] = self.buffer[1:].copy()                             # :159-166 FIFO
        self.buffer[-1] = bf
        self.buffer_cnt[:-1] = self.buffer_cnt[1:].copy()
        self.buffer_cnt[-1] = bc
        return ((self.buffer * self.buffer_cnt).sum(0) / (self.buffer_cnt.sum(0) + META_EPS)).astype(np.float32)

    def _pair(self, SMALL, BIG):
        if self.choice == 'l2':
            return np.float32(np.mean((SMALL.astype(np.float64) - BIG) ** 2))
        if self.choice == 'l1':
            return np.float32(np.mean(np.abs(SMALL.astype(np.float64) - BIG)))
        if self.choice == 'kl':                         # F.kl_div(log SMALL, BIG), elementwise mean
            b, s = BIG.astype(np.float64), SMALL.astype(np.float64)
            with np.errstate(divide="ignore", invalid="ignore"):
                t = np.where(b > 0, b * (np.log(b) - np.log(s)), 0.0)
            return np.float32(t.mean())
        o = self.ot
        return opttrans_1d_forward(SMALL[:, :, None], BIG[:, :, None], o["g_w"], o["g_b"], o["c_w"], o["c_b"],
                                   o.get("epsilon", 1.0), o.get("L", 5))

    def __call__(self, big_feat, big_cnt, small_feat, small_cnt, small_output_all=None, small_gt_all=None):
        if float(_f32(small_feat).sum()) == 0.0:                              # workflow.py:190-194
            return np.float32(0)
        final_big = self._update(big_feat, big_cnt)                           # [F, K]
        buf_has = self.buffer_cnt.sum(0).reshape(-1) > 0   # == buffer_cnt.squeeze() > 0 when BUFFER_SIZE == 1 (:180)
        if self.inst:                                                         # :168-174, 184-186
            gt = np.asarray(small_gt_all).astype(np.int64)
            idx = np.array([i for i in np.nonzero(gt)[0] if buf_has[gt[i]]], np.int64)
            if idx.size == 0:
                return np.float32(0)
            return self._pair(_f32(small_output_all)[idx], final_big[:, gt[idx]].T.copy())
        s_feat, s_cnt = merge_feat_vec(small_feat, small_cnt)
        s_cnt = s_cnt.copy()
        s_cnt[0, 0] = 0                                                       # :178 no background
        idx = np.nonzero((s_cnt.reshape(-1) > 0) & buf_has)[0]
        if idx.size == 0:
            return np.float32(0)
        return self._pair(s_feat[:, idx].T.copy(), final_big[:, idx].T.copy())


# ----------------------------------------------------------------------------
# Target generation (lib/layers.py:224-433 generate_roi, :439-604 generate_target;
# tools/box_utils.py:89-140 box_refinement, compute_iou).  The reference draws its random
# sub-samples with torch.randperm / np.random.permutation; here the permutations are ARGUMENTS,
# so that a test can replay the choice another implementation made and demand identical outputs.
# ----------------------------------------------------------------------------
IOU_EPS = np.float32(10e-20)     # tools/box_utils.py:4


def proposal_candidates(probs, deltas, anchors, pre_nms, bbox_std, window_hw, extra=None):
    """The pre-NMS half of proposal_layer (lib/layers.py:99-127) for ONE image: foreground scores probs[:, 1],
    the pre_nms best in descending order -- the reference's full sort (:99-106) made deterministic: ties go to the
    lower index, `extra` rows [E, 5] rank before anchors at equal score --, deltas * BBOX_STD_DEV (:96),
    apply_box_deltas (tools/box_utils.py:7-33) and clip_boxes (:36-60), every operation in float32.
    Returns (dets [pre_nms, 5], chosen [pre_nms] indices into the concatenation (extra, anchors))."""
    f = np.float32
    scores = np.asarray(probs, f)[:, 1]
    E = 0 if extra is None else len(extra)
    if E:
        scores = np.concatenate([np.asarray(extra, f)[:, 4], scores])
    order = np.argsort(-scores.astype(np.float64), kind="stable")[:pre_nms]
    out = np.zeros((len(order), 5), f)
    an = order >= E
    a_idx = order[an] - E
    b = np.asarray(anchors, f)[a_idx]
    d = np.asarray(deltas, f)[a_idx] * np.asarray(bbox_std, f)[None, :]
    h = b[:, 2] - b[:, 0]
    w = b[:, 3] - b[:, 1]
    cy = b[:, 0] + f(0.5) * h
    cx = b[:, 1] + f(0.5) * w
    cy = cy + d[:, 0] * h
    cx = cx + d[:, 1] * w
    h = h * np.exp(d[:, 2])
    w = w * np.exp(d[:, 3])
    y1 = cy - f(0.5) * h
    x1 = cx - f(0.5) * w
    out[an, 0], out[an, 1], out[an, 2], out[an, 3] = y1, x1, y1 + h, x1 + w
    if E:
        out[~an, :4] = np.asarray(extra, f)[order[~an], :4]
    H, W = f(window_hw[0]), f(window_hw[1])
    out[:, 0] = np.minimum(np.maximum(out[:, 0], f(0)), H)
    out[:, 1] = np.minimum(np.maximum(out[:, 1], f(0)), W)
    out[:, 2] = np.minimum(np.maximum(out[:, 2], f(0)), H)
    out[:, 3] = np.minimum(np.maximum(out[:, 3], f(0)), W)
    out[:, 4] = scores[order]
    return out, order


def compute_iou(boxes1, boxes2):
    """tools/box_utils.py:112-140, fp32 in the reference's operation order; [N1, N2]."""
    b1, b2 = _f32(boxes1)[:, None, :], _f32(boxes2)[None, :, :]
    y1 = np.maximum(b1[..., 0], b2[..., 0])
    x1 = np.maximum(b1[..., 1], b2[..., 1])
    y2 = np.minimum(b1[..., 2], b2[..., 2])
    x2 = np.minimum(b1[..., 3], b2[..., 3])
    zero = np.float32(0)
    inter = np.maximum(x2 - x1, zero) * np.maximum(y2 - y1, zero)
    a1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    a2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    union = a1 + a2 - inter
    return (inter / (union + IOU_EPS)).astype(np.float32)


def box_refinement(box, gt_box):
    """tools/box_utils.py:89-109."""
    box, gt_box = _f32(box), _f32(gt_box)
    half = np.float32(0.5)
    h = box[:, 2] - box[:, 0]
    w = box[:, 3] - box[:, 1]
    cy = box[:, 0] + half * h
    cx = box[:, 1] + half * w
    gh = gt_box[:, 2] - gt_box[:, 0]
    gw = gt_box[:, 3] - gt_box[:, 1]
    gcy = gt_box[:, 0] + half * gh
    gcx = gt_box[:, 1] + half * gw
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.stack([(gcy - cy) / h, (gcx - cx) / w, np.log(gh / h), np.log(gw / w)], 1).astype(np.float32)


def _split_crowd(gt_class_ids, gt_boxes, boxes_for_crowd):
    """lib/layers.py:229-246 == :455-472: crowd rows (class < 0) are removed together with the zero
    padding (only class > 0 rows are kept) and boxes overlapping a crowd by >= 0.001 are flagged."""
    ids = np.asarray(gt_class_ids)
    if (ids < 0).any():
        crowd = gt_boxes[ids < 0]
        keep = ids > 0
        no_crowd = compute_iou(boxes_for_crowd, crowd).max(-1) < np.float32(0.001)
        return keep, no_crowd
    return np.ones(len(ids), bool), np.ones(len(boxes_for_crowd), bool)


def generate_rpn_target(anchors, gt_class_ids, gt_boxes, cfg, perm_pos=None, perm_neg=None):
    """generate_target, lib/layers.py:439-604 (one image).  Returns (match [A] in {1,-1,0},
    bbox [TRAIN_ANCHORS_PER_IMAGE, 4] = refinements of the positive anchors in anchor order, NOT yet
    divided by BBOX_STD_DEV -- prepare_rpn_target does that, :655).  perm_pos / perm_neg play the role
    of np.random.permutation(len(ids)) at :521 / :543."""
    anchors, gt_boxes = _f32(anchors), _f32(gt_boxes)
    ids = np.asarray(gt_class_ids).astype(np.int64)
    keep, no_crowd = _split_crowd(ids, gt_boxes, anchors)
    had_crowd = (ids < 0).any()
    if had_crowd:
        gt_boxes, ids = gt_boxes[keep], ids[keep]
    actual_gt_num = int((ids > 0).sum())
    A = anchors.shape[0]
    n_total = cfg.RPN.TRAIN_ANCHORS_PER_IMAGE
    match = np.zeros(A, np.float32)
    bbox = np.zeros((n_total, 4), np.float32)
    overlaps = compute_iou(anchors, gt_boxes)                                     # [A, G]
    iou_max, iou_argmax = overlaps.max(-1), overlaps.argmax(-1)
    match[(iou_max < np.float32(cfg.RPN.TARGET_NEG_THRES)) & no_crowd] = -1       # :491
    gt_iou_argmax = overlaps.argmax(0)                                            # :494
    match[gt_iou_argmax[:actual_gt_num]] = 1                                      # :495 (valid GTs come first)
    match[iou_max >= np.float32(cfg.RPN.TARGET_POS_THRES)] = 1                    # :498
    pos_ids = np.nonzero(match == 1)[0]
    pos_extra = len(pos_ids) - n_total // 2
    if pos_extra > 0:                                                             # :516-527
        match[pos_ids[np.asarray(perm_pos)[:pos_extra]]] = 0
    neg_ids = np.nonzero(match == -1)[0]
    neg_extra = len(neg_ids) - (n_total - int((match == 1).sum()))
    if neg_extra > 0:                                                             # :541-546
        match[neg_ids[np.asarray(perm_neg)[:neg_extra]]] = 0
    pos_ids = np.nonzero(match == 1)[0]                                           # :597-604
    bbox[:len(pos_ids)] = box_refinement(anchors[pos_ids], gt_boxes[iou_argmax[pos_ids]])
    return match, bbox


def generate_roi(cfg, proposals, gt_class_ids, gt_boxes, gt_masks, perm_pos, perm_neg, num_valid=None):
    """generate_roi, lib/layers.py:224-376 (one image; boxes normalised).  Returns (rois, class_ids,
    deltas, masks) -- positives first, then negatives -- or None where the reference returns None.
    perm_pos / perm_neg stand for torch.randperm at :273 / :326.  num_valid: only the first num_valid
    proposals are real (the build carries a count instead of zero rows, DESIGN quirk Q3; the reference
    would let all-zero padding rows be drawn as negatives)."""
    proposals, gt_boxes = _f32(proposals), _f32(gt_boxes)
    ids = np.asarray(gt_class_ids).astype(np.int64)
    gt_masks = _f32(gt_masks)
    if num_valid is not None:
        proposals = proposals[:num_valid]
    keep, no_crowd = _split_crowd(ids, gt_boxes, proposals)
    if (ids < 0).any():
        ids, gt_boxes, gt_masks = ids[keep], gt_boxes[keep], gt_masks[keep]
    overlaps = compute_iou(proposals, gt_boxes)
    roi_iou_max = overlaps.max(-1) if overlaps.shape[1] else np.zeros(len(proposals), np.float32)
    pos_bool = roi_iou_max >= np.float32(0.5)
    neg_bool = (roi_iou_max < np.float32(0.5)) & no_crowd
    R = cfg.ROIS.TRAIN_ROIS_PER_IMAGE
    mh, mw = cfg.MRCNN.MASK_SHAPE
    pos_cnt = neg_cnt = 0
    if pos_bool.any():
        pos_ind = np.nonzero(pos_bool)[0]
        pos_cap = int(R * cfg.ROIS.ROI_POSITIVE_RATIO)
        pos_ind = pos_ind[np.asarray(perm_pos)[:pos_cap]]
        pos_cnt = len(pos_ind)
        pos_rois = proposals[pos_ind]
        assign = overlaps[pos_ind].argmax(1)
        roi_gt = gt_boxes[assign]
        cls = ids[assign].astype(np.int32)
        deltas = box_refinement(pos_rois, roi_gt) / _f32(cfg.DATA.BBOX_STD_DEV)
        boxes = pos_rois
        if cfg.MRCNN.USE_MINI_MASK:                                               # :301-312
            gh = roi_gt[:, 2:3] - roi_gt[:, 0:1]
            gw = roi_gt[:, 3:4] - roi_gt[:, 1:2]
            boxes = np.concatenate([(pos_rois[:, 0:1] - roi_gt[:, 0:1]) / gh, (pos_rois[:, 1:2] - roi_gt[:, 1:2]) / gw,
                                    (pos_rois[:, 2:3] - roi_gt[:, 0:1]) / gh, (pos_rois[:, 3:4] - roi_gt[:, 1:2]) / gw], 1)
        masks = crop_and_resize_forward(gt_masks[assign][:, None], boxes, np.arange(pos_cnt, dtype=np.int32), mh, mw)
        masks = np.rint(masks[:, 0])                                              # torch.round: half to even
    if neg_bool.any() and pos_cnt > 0:
        neg_ind = np.nonzero(neg_bool)[0]
        r = 1.0 / cfg.ROIS.ROI_POSITIVE_RATIO
        neg_want = int(r * pos_cnt - pos_cnt)
        neg_ind = neg_ind[np.asarray(perm_neg)[:neg_want]]
        neg_cnt = len(neg_ind)
        neg_rois = proposals[neg_ind]
    if pos_cnt == 0:
        return None          # (:359-371 can only be reached with pos_cnt > 0 because of the guard at :322)
    if neg_cnt > 0:
        return (np.concatenate([pos_rois, neg_rois], 0), np.concatenate([cls, np.zeros(neg_cnt, np.int32)]),
                np.concatenate([deltas, np.zeros((neg_cnt, 4), np.float32)], 0),
                np.concatenate([masks, np.zeros((neg_cnt, mh, mw), np.float32)], 0))
    return pos_rois, cls, deltas, masks
