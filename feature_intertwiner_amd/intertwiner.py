"""Intertwiner statistics: per-class feature means, the count-weighted merge over
(GPU, scale) and the history buffer -- the pieces of lib/sub_module.py:664-684 and
lib/model.py:143-224 that sit between the RoI operators and the OT loss.
"""
import torch

from . import _lib

EPS = 1e-20


class _ClassMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, gt, num_classes):
        _lib.require_cuda(features, gt)
        L = _lib.load()
        n_rows = features.shape[0]
        n_feat = 1
        for d in features.shape[1:]:
            n_feat *= int(d)
        f2 = features.reshape(n_rows, n_feat).contiguous().float()
        gt_c = gt.detach().reshape(-1).contiguous().to(torch.int32)
        N, F = f2.shape
        feat = torch.empty((F, num_classes), device=f2.device, dtype=torch.float32)
        cnt = torch.empty((num_classes,), device=f2.device, dtype=torch.float32)
        ws = torch.empty((max(L.fi_class_mean_workspace_bytes(N, F, int(num_classes)), 4) // 4,),
                         device=f2.device, dtype=torch.float32)
        with torch.cuda.device(f2.device):
            _lib.check(L.fi_class_mean_forward(_lib.ptr(f2), _lib.ptr(gt_c), N, F, int(num_classes),
                                               _lib.ptr(feat), _lib.ptr(cnt), _lib.ptr(ws),
                                               _lib.current_stream()),
                       "fi_class_mean_forward")
        ctx.save_for_backward(gt_c, cnt)
        ctx.shape = tuple(features.shape)
        ctx.nf = (N, F, int(num_classes))
        ctx.mark_non_differentiable(cnt)
        return feat, cnt

    @staticmethod
    def backward(ctx, grad_feat, _grad_cnt):
        gt_c, cnt = ctx.saved_tensors
        L = _lib.load()
        N, F, K = ctx.nf
        g = grad_feat.contiguous().float()
        out = torch.empty((N, F), device=g.device, dtype=torch.float32)
        with torch.cuda.device(g.device):
            _lib.check(L.fi_class_mean_backward(_lib.ptr(g), _lib.ptr(gt_c), _lib.ptr(cnt), N, F, K,
                                                _lib.ptr(out), _lib.current_stream()),
                       "fi_class_mean_backward")
        return out.view(ctx.shape), None, None


def class_mean(features, gt, num_classes):
    """Dev._assign_feat2cls (lib/sub_module.py:664-684): features [N, F(,1,1)], gt [N] ->
    feat [F, num_classes] (column c = mean of the rows of foreground class c, zero when
    absent; background skipped) and cnt [1, num_classes]."""
    feat, cnt = _ClassMean.apply(features, gt, num_classes)
    return feat, cnt.view(1, num_classes)


def merge_feat_vec(box_feat, box_cnt):
    """MaskRCNN._merge_feat_vec (lib/model.py:217-224): count-weighted mean over the
    leading (gpu, scale) axes.  box_feat [G, S, F, K], box_cnt [G, S, 1, K]."""
    feat_sum = (box_feat * box_cnt).sum(0).sum(0)
    cnt_sum = box_cnt.sum(0).sum(0)
    return feat_sum / (cnt_sum + EPS), cnt_sum


def roi_level(rois, image_area, base=224.0):
    """FPN level of each RoI (lib/sub_module.py:396-410 == lib/layers.py:168-181), int32 in
    [2, 5].  rois [..., 4] normalised (y1, x1, y2, x2)."""
    h = rois[..., 2] - rois[..., 0]
    w = rois[..., 3] - rois[..., 1]
    area = torch.tensor(float(image_area), device=rois.device, dtype=torch.float32)
    ln2 = torch.log(torch.tensor(2.0, device=rois.device))
    lvl = 4 + torch.log(torch.sqrt(h * w) / (base / torch.sqrt(area))) / ln2
    lvl = torch.nan_to_num(lvl.round(), nan=2.0, posinf=5.0, neginf=2.0)
    return lvl.clamp(2, 5).to(torch.int32)


class FeatureBuffer(object):
    """History buffer of big-object class features (lib/model.py:106-115, 150-166)."""

    def __init__(self, buffer_size, feat_dim, num_classes, device):
        self.buffer = torch.zeros(buffer_size, feat_dim, num_classes, device=device)
        self.buffer_cnt = torch.zeros(buffer_size, 1, num_classes, device=device)

    @torch.no_grad()
    def update(self, big_feat, big_cnt):
        """big_feat [F, K], big_cnt [1, K] (already merged).  Returns the class features the
        meta loss compares against, [F, K]."""
        if self.buffer.size(0) == 1:
            feat_sum = self.buffer * self.buffer_cnt + big_feat.unsqueeze(0) * big_cnt.unsqueeze(0)
            self.buffer_cnt += big_cnt.unsqueeze(0)
            self.buffer = feat_sum / (self.buffer_cnt + EPS)
            return self.buffer[0]
        self.buffer = torch.roll(self.buffer, -1, 0)
        self.buffer_cnt = torch.roll(self.buffer_cnt, -1, 0)
        self.buffer[-1] = big_feat
        self.buffer_cnt[-1] = big_cnt
        return (self.buffer * self.buffer_cnt).sum(0) / (self.buffer_cnt.sum(0) + EPS)
