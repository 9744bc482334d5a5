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
    area = _lib.const_tensor([float(image_area)], rois.device)[0]
    ln2 = torch.log(_lib.const_tensor([2.0], rois.device)[0])
    lvl = 4 + torch.log(torch.sqrt(h * w) / (base / torch.sqrt(area))) / ln2
    lvl = torch.nan_to_num(lvl.round(), nan=2.0, posinf=5.0, neginf=2.0)
    return lvl.clamp(2, 5).to(torch.int32)


class FeatureBuffer(object):
    """History buffer of big-object class features (lib/model.py:106-115, 150-166)."""

    def __init__(self, buffer_size, feat_dim, num_classes, device):
        self.buffer = torch.zeros(buffer_size, feat_dim, num_classes, device=device)
        self.buffer_cnt = torch.zeros(buffer_size, 1, num_classes, device=device)

    @torch.no_grad()
    def update(self, big_feat, big_cnt, active=None):
        """big_feat [F, K], big_cnt [1, K] (already merged).  Returns the class features the
        meta loss compares against, [F, K].  `active` (0-dim bool tensor, optional): when False
        the buffer is left untouched -- the reference calls meta_loss, and with it this update,
        only on steps that have small-object statistics (lib/workflow.py:190-194); a device-side
        select keeps the step free of host synchronisation."""
        if self.buffer.size(0) == 1:
            feat_sum = self.buffer * self.buffer_cnt + big_feat.unsqueeze(0) * big_cnt.unsqueeze(0)
            new_cnt = self.buffer_cnt + big_cnt.unsqueeze(0)
            new_buf = feat_sum / (new_cnt + EPS)
        else:
            new_buf = torch.roll(self.buffer, -1, 0)
            new_cnt = torch.roll(self.buffer_cnt, -1, 0)
            new_buf[-1] = big_feat
            new_cnt[-1] = big_cnt
        if active is not None:
            new_buf = torch.where(active, new_buf, self.buffer)
            new_cnt = torch.where(active, new_cnt, self.buffer_cnt)
        # in place: the tensors keep their addresses (a captured step -- hipGraph -- reads and writes the same buffers at
        # every replay; rebinding would leave the replays on the first step's tensors)
        self.buffer.copy_(new_buf)
        self.buffer_cnt.copy_(new_cnt)
        if self.buffer.size(0) == 1:
            # a copy: the caller saves it for the pair loss's backward, and the next update writes the buffer in place
            # (two forward passes before one backward are supported, conv.py)
            return self.buffer[0].clone()
        return (self.buffer * self.buffer_cnt).sum(0) / (self.buffer_cnt.sum(0) + EPS)


class _MetaStatsFn(torch.autograd.Function):
    """The statistics side of meta_loss below as three launches (fi_meta_stats_forward): merged big / small class
    features, history update in place, class selection, the transposed operands of the pair loss.  Differentiable in the
    small class features only (the big ones are detached in the reference too)."""

    @staticmethod
    def forward(ctx, small_feat, small_cnt, big_feat, big_cnt, buffer, buffer_cnt):
        G, S, F_, K = small_feat.shape
        lib = _lib.load()
        dev = small_feat.device
        sc = small_cnt.reshape(G * S, K).contiguous().float()
        bc = big_cnt.reshape(G * S, K).contiguous().float()
        SMALL = torch.empty((K - 1, F_), device=dev, dtype=torch.float32)
        BIG = torch.empty((K - 1, F_), device=dev, dtype=torch.float32)
        on = torch.empty(K - 1, device=dev, dtype=torch.float32)
        s_cnt = torch.empty(K, device=dev, dtype=torch.float32)
        active = torch.empty(1, device=dev, dtype=torch.float32)
        ws = torch.empty(int(lib.fi_meta_stats_workspace_bytes(F_, K)) // 4 + 1, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.fi_meta_stats_forward(_lib.ptr(big_feat), _lib.ptr(bc), big_feat.stride(2), big_feat.stride(1),
                                                 big_feat.stride(0), _lib.ptr(small_feat), _lib.ptr(sc), small_feat.stride(2),
                                                 small_feat.stride(1), small_feat.stride(0), G, S, F_, K, _lib.ptr(buffer),
                                                 _lib.ptr(buffer_cnt), _lib.ptr(s_cnt), _lib.ptr(SMALL), _lib.ptr(BIG),
                                                 _lib.ptr(on), _lib.ptr(active), _lib.ptr(ws), _lib.current_stream()),
                       "fi_meta_stats_forward")
        ctx.save_for_backward(s_cnt, sc)
        ctx.layout = (tuple(small_feat.shape), tuple(small_feat.stride()))
        ctx.mark_non_differentiable(BIG, on, active)
        return SMALL, BIG, on, active

    @staticmethod
    def backward(ctx, d_small, _d_big, _d_on, _d_active):
        s_cnt, sc = ctx.saved_tensors
        shape, stride = ctx.layout
        G, S, F_, K = shape
        g = d_small.contiguous().float()
        out = torch.empty_strided(shape, stride, device=g.device, dtype=torch.float32)
        with torch.cuda.device(g.device):
            _lib.check(_lib.load().fi_meta_stats_backward(_lib.ptr(g), _lib.ptr(s_cnt), _lib.ptr(sc), G, S, F_, K, stride[2],
                                                          stride[1], stride[0], _lib.ptr(out), _lib.current_stream()),
                       "fi_meta_stats_backward")
        return out, None, None, None, None, None


class _MetaStatsReducedFn(torch.autograd.Function):
    """_MetaStatsFn under data parallelism: this rank's count-weighted sums in one flat vector (fi_meta_stats_sums), ONE
    all-reduce of it (`flat_sum`: in place, returns the world size), and the rest of the statistics side from the reduced
    vector (fi_meta_stats_from_sums) -- 1 + 3 launches around the collective instead of the ~60 of the tensor formulation.
    Backward: every rank back-propagates the same meta loss through its own contribution only and gradients are averaged
    afterwards, so the path into the local statistics carries the world size (data_parallel._AllReduceSumIdentityGrad)."""

    @staticmethod
    def forward(ctx, small_feat, small_cnt, big_feat, big_cnt, buffer, buffer_cnt, flat_sum):
        G, S, F_, K = small_feat.shape
        lib = _lib.load()
        dev = small_feat.device
        sc = small_cnt.reshape(G * S, K).contiguous().float()
        bc = big_cnt.reshape(G * S, K).contiguous().float()
        sums = torch.empty(2 * F_ * K + 2 * K, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.fi_meta_stats_sums(_lib.ptr(big_feat), _lib.ptr(bc), big_feat.stride(2), big_feat.stride(1),
                                              big_feat.stride(0), _lib.ptr(small_feat), _lib.ptr(sc), small_feat.stride(2),
                                              small_feat.stride(1), small_feat.stride(0), G, S, F_, K, _lib.ptr(sums),
                                              _lib.current_stream()), "fi_meta_stats_sums")
        world = flat_sum(sums)
        SMALL = torch.empty((K - 1, F_), device=dev, dtype=torch.float32)
        BIG = torch.empty((K - 1, F_), device=dev, dtype=torch.float32)
        on = torch.empty(K - 1, device=dev, dtype=torch.float32)
        s_cnt = torch.empty(K, device=dev, dtype=torch.float32)
        active = torch.empty(1, device=dev, dtype=torch.float32)
        ws = torch.empty(int(lib.fi_meta_stats_workspace_bytes(F_, K)) // 4 + 1, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.fi_meta_stats_from_sums(_lib.ptr(sums), F_, K, _lib.ptr(buffer), _lib.ptr(buffer_cnt),
                                                   _lib.ptr(s_cnt), _lib.ptr(SMALL), _lib.ptr(BIG), _lib.ptr(on),
                                                   _lib.ptr(active), _lib.ptr(ws), _lib.current_stream()),
                       "fi_meta_stats_from_sums")
        ctx.save_for_backward(s_cnt, sc)
        ctx.layout = (tuple(small_feat.shape), tuple(small_feat.stride()))
        ctx.world = float(world)
        ctx.mark_non_differentiable(BIG, on, active)
        return SMALL, BIG, on, active

    @staticmethod
    def backward(ctx, d_small, _d_big, _d_on, _d_active):
        s_cnt, sc = ctx.saved_tensors                      # the GLOBAL small counts, this rank's own counts
        shape, stride = ctx.layout
        G, S, F_, K = shape
        g = d_small.contiguous().float()
        if ctx.world != 1.0:
            g = g * ctx.world
        out = torch.empty_strided(shape, stride, device=g.device, dtype=torch.float32)
        with torch.cuda.device(g.device):
            _lib.check(_lib.load().fi_meta_stats_backward(_lib.ptr(g), _lib.ptr(s_cnt), _lib.ptr(sc), G, S, F_, K, stride[2],
                                                          stride[1], stride[0], _lib.ptr(out), _lib.current_stream()),
                       "fi_meta_stats_backward")
        return out, None, None, None, None, None, None


def _dense_levels(t):
    """[G, S, F, K] whose layout covers its storage exactly once with unit stride along K: the stacked (contiguous) tensor,
    or -- G = 1 -- the [F, S K] result of one class-mean launch viewed per level."""
    if t.dim() != 4 or t.stride(3) != 1:
        return False
    G, S, F_, K = t.shape
    if t.is_contiguous():
        return True
    return G == 1 and (t.stride(1), t.stride(2)) == (K, S * K)


STATS_KERNEL = __import__("os").environ.get("FI_META_STATS_KERNEL", "1") != "0"      # A/B switch


def _masked_mean(per_row, weight):
    return (per_row * weight).sum() / weight.sum().clamp(min=1)


def _pair_loss(choice, SMALL, BIG, on, ot_module):
    """Per-row loss between SMALL [n, F] (carries the gradient) and BIG [n, F] (constant) for the
    rows with on[n] > 0 (lib/model.py:197-207).  l2 / l1 / kl are means over the features, so the
    mean over the selected rows equals F.mse_loss / F.l1_loss / F.kl_div over the selected block."""
    if choice == 'ot':
        return ot_module(SMALL.unsqueeze(-1), BIG.unsqueeze(-1).contiguous())
    if choice == 'l2':
        return ((SMALL - BIG) ** 2).mean(1)
    if choice == 'l1':
        return (SMALL - BIG).abs().mean(1)
    if choice == 'kl':
        # rows outside the selection have all-zero statistics: keep them out of the logarithms (the
        # reference indexes the selected rows first, lib/model.py:187-201).  F.kl_div(log q, p) is
        # p * (log p - log q) with 0 where p == 0.
        m = on.view(-1, 1) > 0
        sm = torch.where(m, SMALL, torch.ones_like(SMALL)).clamp_min(1e-38)
        bg = torch.where(m, BIG, torch.zeros_like(BIG))
        return (bg * (torch.log(bg.clamp_min(1e-38)) - torch.log(sm))).mean(1)
    raise ValueError(choice)


def meta_loss(cfg, feature_buffer, ot_module, feat_input, reduce_fn=None):
    """MaskRCNN.meta_loss (lib/model.py:143-210) on static shapes, with no host synchronisation.

    feat_input = [big_feat, big_cnt, small_feat, small_cnt, small_output_all, small_gt_all];
    big_*/small_* are [G, S, F, K] / [G, S, 1, K] stacks over (gpu, scale) as Dev.forward returns
    them.  `reduce_fn(sum_feat, sum_cnt)` -- the data-parallel path -- all-reduces count-weighted
    sums across ranks: algebraically the reference's gather-to-GPU-0 + _merge_feat_vec (:217-224).

    * The history buffer is updated only when the step has small-object statistics, as the
      reference guards the whole call with `small_feat.sum() != 0` (lib/workflow.py:190-194);
      otherwise the loss is 0 and the buffer is untouched.
    * Class selection (:176-181): foreground classes with a small count > 0 AND a buffer count > 0.
      With BUFFER_SIZE > 1 the reference's `buffer_cnt.squeeze()` is [BUFFER_SIZE, K] and its
      selection line does not execute (2-D nonzero); the count summed over the history is used.
    * 'ot' returns one value per selected class in the reference (:207), which loss.backward()
      cannot reduce (quirk Q10): the mean over the selected classes is taken, as l1/l2/kl do.
    * DEV.INST_LOSS (:168-174, 184-186): rows of small_output_all whose class is in the buffer are
      compared with the buffer column of their class.
    """
    big_feat, big_cnt, small_feat, small_cnt = feat_input[:4]
    flat_sum = getattr(reduce_fn, "flat_sum", None)       # data_parallel.all_reduce_statistics carries one
    if STATS_KERNEL and (reduce_fn is None or flat_sum is not None) and not cfg.DEV.INST_LOSS and small_feat.is_cuda and \
            feature_buffer.buffer.size(0) == 1 and small_feat.dtype == torch.float32 and big_feat.dtype == torch.float32 and \
            _dense_levels(small_feat) and _dense_levels(big_feat) and small_feat.shape == big_feat.shape and \
            feature_buffer.buffer.is_contiguous() and feature_buffer.buffer_cnt.is_contiguous():
        # a history of one step (every shipped configuration): the statistics side as three launches on one rank, as
        # 1 + 3 launches around ONE all-reduce under data parallelism
        if reduce_fn is None:
            SMALL, BIG, on, active = _MetaStatsFn.apply(small_feat, small_cnt.detach(), big_feat.detach(), big_cnt.detach(),
                                                        feature_buffer.buffer, feature_buffer.buffer_cnt)
        else:
            SMALL, BIG, on, active = _MetaStatsReducedFn.apply(small_feat, small_cnt.detach(), big_feat.detach(),
                                                               big_cnt.detach(), feature_buffer.buffer,
                                                               feature_buffer.buffer_cnt, flat_sum)
        loss = _masked_mean(_pair_loss(cfg.DEV.LOSS_CHOICE, SMALL, BIG, on, ot_module), on)
        return torch.where(active[0] > 0, loss, torch.zeros_like(loss))

    def merged(feat, cnt):
        s = (feat * cnt).sum(0).sum(0)
        c = cnt.sum(0).sum(0)
        if reduce_fn is not None:
            s, c = reduce_fn(s, c)
        return s / (c + EPS), c, s

    b_feat, b_cnt, _ = merged(big_feat.detach(), big_cnt.detach())
    s_feat, s_cnt, s_sum = merged(small_feat, small_cnt.detach())
    # lib/workflow.py:190: `small_feat.sum() != 0` (class means are >= 0 after ReLU / sigmoid /
    # softmax, so the count-weighted sums vanish exactly when the means do)
    active = s_sum.detach().sum() != 0
    final_big = feature_buffer.update(b_feat, b_cnt, active)                  # [F, K]
    buf_cnt = feature_buffer.buffer_cnt.sum(0)                                # [1, K]
    choice = cfg.DEV.LOSS_CHOICE
    if cfg.DEV.INST_LOSS:
        rows, gt = feat_input[4], feat_input[5]
        gt_l = gt.detach().long()
        in_buf = (buf_cnt.view(-1) > 0)[gt_l]
        on = ((gt_l != 0) & in_buf).float()
        BIG = final_big.t()[gt_l].detach()                                    # [N, F]
        per_row = _pair_loss(choice, rows, BIG, on, ot_module)
        num, den = (per_row * on).sum(), on.sum()
        if reduce_fn is not None:      # global mean over the instances of every rank
            num, den = reduce_fn(num.view(1), den.view(1))
            num, den = num.view(()), den.view(())
        loss = num / den.clamp(min=1)
    else:
        # no background class (a scalar assignment `s_cnt[0, 0] = 0` is a synchronising host-to-device copy)
        s_cnt = s_cnt * _lib.const_tensor([0.0] + [1.0] * (s_cnt.size(1) - 1), s_cnt.device).view(1, -1)
        on = ((s_cnt > 0) & (buf_cnt > 0)).view(-1)[1:].float()               # foreground classes
        SMALL = s_feat[:, 1:].t()                                             # [K-1, F]
        BIG = final_big[:, 1:].t().detach()
        loss = _masked_mean(_pair_loss(choice, SMALL, BIG, on, ot_module), on)
    return torch.where(active, loss, torch.zeros_like(loss))
