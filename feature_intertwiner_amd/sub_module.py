"""Network modules around the hot-path operators: ResNet-FPN backbone, RPN, the Dev
(intertwiner RoI) stage, classifier and mask heads.  Counterpart of lib/sub_module.py
of the reference, with identical parameter names (state-dict compatible) and the same
layer arithmetic; the RoI stage is re-designed around the one-launch pyramid RoIAlign
kernel instead of per-level nonzero / gather / crop / cat / scatter loops.

Dense convolutions run on the fp32 MFMA implicit-GEMM kernels (conv.py / csrc/conv_igemm.hip)
with eval-BN, shortcut and ReLU folded into the epilogue.  BatchNorm is always evaluated with
running statistics, as in the reference where `if mode == 'inference' or 'visualize'` is always
true (lib/model.py:265-267).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import conv as _conv
from .conv import (Conv2d, ConvTranspose2x2, GradBox, conv1x1_class_rows, conv2d, conv_bias_relu, conv_bn_act, linear,
                   maxpool3x3s2, take_rows, upsample2x)
from .intertwiner import class_mean, roi_level
from .roi_align.crop_and_resize import CropAndResizeFunction, CropGradGroup, pyramid_crop_and_resize
from .roi_pooling.functions.roi_pool import RoIPoolFunction


import os as _os
_BIG_SIDE = _os.environ.get('FI_BIG_SIDE', '1') != '0'
_INDEX_KERNEL = _os.environ.get('FI_INDEX_KERNEL', '1') != '0'      # fi_dev_stage_index vs its tensor formulation (A/B switch)
# the Dev stage without its host read (Dev.static_shapes): True / False, or None = on the fp32 kernels only -- on the 16-bit
# kernels the extra rows of the static batches cost more than the read (cfg5: 46.4 vs 45.6 ms/step, same box)
_STATIC_DEV = {'1': True, '0': False}.get(_os.environ.get('FI_STATIC_DEV', ''), None)

class SamePad2d(nn.Module):
    """TensorFlow 'SAME' padding (lib/sub_module.py:9-33).  `folded=True` means the following
    convolution carries the (symmetric) padding itself, so no padded copy is materialised."""

    def __init__(self, kernel_size, stride, folded=False):
        super(SamePad2d, self).__init__()
        self.kernel_size = nn.modules.utils._pair(kernel_size)
        self.stride = nn.modules.utils._pair(stride)
        self.folded = folded

    def pads(self, in_h, in_w):
        out_h = math.ceil(float(in_h) / float(self.stride[0]))
        out_w = math.ceil(float(in_w) / float(self.stride[1]))
        pad_h = max((out_h - 1) * self.stride[0] + self.kernel_size[0] - in_h, 0)
        pad_w = max((out_w - 1) * self.stride[1] + self.kernel_size[1] - in_w, 0)
        return pad_w // 2, pad_w - pad_w // 2, pad_h // 2, pad_h - pad_h // 2

    def forward(self, x):
        if self.folded:
            return x
        l, r, t, b = self.pads(x.size(2), x.size(3))
        return F.pad(x, (l, r, t, b), 'constant', 0)

    def __repr__(self):
        return self.__class__.__name__


def _pad_maxpool(x, pad, pool):
    """SamePad2d + MaxPool2d of the stem (lib/sub_module.py:44-45).  The padding is on the right / bottom only
    and the input comes out of a ReLU, so the zero column it adds never changes a maximum: pooling with
    ceil_mode (the overhanging window ignores what is outside) gives the same values without materialising the
    padded copy (0.25 GB at 4 x 1024^2), forward and backward."""
    l, r, t, b = pad.pads(x.size(2), x.size(3))
    k, s_ = nn.modules.utils._pair(pool.kernel_size), nn.modules.utils._pair(pool.stride)
    if l == 0 and t == 0 and r < k[1] and b < k[0] and tuple(nn.modules.utils._pair(pool.padding)) == (0, 0):
        if k == (3, 3) and s_ == (2, 2):
            y = maxpool3x3s2(x)            # own kernels: no index tensor, the stem's ReLU mask applied on the way back
            if y is not None and y.shape[2] == (x.size(2) + b - 3) // 2 + 1 and y.shape[3] == (x.size(3) + r - 3) // 2 + 1:
                return y
        y = F.max_pool2d(x, k, s_, 0, ceil_mode=True)
        if y.shape[2] == (x.size(2) + b - k[0]) // s_[0] + 1 and y.shape[3] == (x.size(3) + r - k[1]) // s_[1] + 1:
            return y
    return pool(pad(x))


def _bn(ch, eps=0.001, momentum=0.01):
    return nn.BatchNorm2d(ch, eps=eps, momentum=momentum)


def _fused_path(x, *bns):
    """conv_bn_act takes its one-launch path for these layers (eval-mode BN with running statistics, a
    real spatial extent)."""
    return x.is_cuda and x.shape[2] * x.shape[3] > 1 and all((not b.training) and b.track_running_stats for b in bns)


def _conv_bn(owner, index, cin, cout, kernel, stride=1, padding=0, eps=0.001, momentum=0.01):
    """Register `conv<index>` / `bn<index>` on `owner` (the reference's parameter names, hence its state-dict
    keys: lib/sub_module.py:90-98, 704-710, 757-768)."""
    owner.add_module("conv%d" % index, Conv2d(cin, cout, kernel_size=kernel, stride=stride, padding=padding))
    owner.add_module("bn%d" % index, nn.BatchNorm2d(cout, eps=eps, momentum=momentum))


class Bottleneck(nn.Module):
    """1x1 (strided) -> 3x3 -> 1x1 (x4 channels) with a projection shortcut where the shape changes
    (lib/sub_module.py:84-128)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(Bottleneck, self).__init__()
        for index, (cin, cout, kernel, st, pad) in enumerate(((inplanes, planes, 1, stride, 0),
                                                              (planes, planes, 3, 1, 1),
                                                              (planes, planes * self.expansion, 1, 1, 0)), start=1):
            if kernel == 3:
                self.padding2 = SamePad2d(kernel_size=3, stride=1, folded=True)
            _conv_bn(self, index, cin, cout, kernel, st, pad)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.input_sole = False       # True: nothing but this block reads its input (the inner blocks of a stage)
        self.lateral_box = None       # set for ONE forward call by FPN.forward on the first block of C3..C5

    def forward(self, x):
        # conv + eval-BN (+ shortcut) + ReLU are one kernel launch each (conv.conv_bn_act)
        if self.downsample is None and _fused_path(x, self.bn1, self.bn3) and x.requires_grad and torch.is_grad_enabled():
            # identity shortcut: x receives two gradients, conv1's data gradient and the shortcut's.  The
            # shortcut gradient (produced by conv3's backward, which always runs first) is handed to
            # conv1's backward and added inside its data-gradient kernel, instead of a separate pass over
            # two block-sized tensors (33 blocks in ResNet-101).
            # gate_dx: conv2 / conv3 are the only readers of their inputs, and with `input_sole` (set by
            # ResNet.make_layer for the blocks after the first of a stage) this block is the only reader of x --
            # conv1's kernel sees x's whole gradient (its own + the shortcut's through the box): the ReLU masks of
            # the three producing layers are applied in these layers' data-gradient epilogues (conv.Gate).
            box = GradBox()
            out = conv_bn_act(x, self.conv1, self.bn1, relu=True, dx_add_from=box, gate_dx=self.input_sole)
            out = conv_bn_act(out, self.conv2, self.bn2, relu=True, gate_dx=True)
            return conv_bn_act(out, self.conv3, self.bn3, relu=True, residual=x, res_grad_to=box, gate_dx=True)
        if self.downsample is not None and _fused_path(x, self.bn1, self.bn3, self.downsample[1]) and \
                x.requires_grad and torch.is_grad_enabled():
            # projection shortcut: x receives the data gradients of conv1 and of the projection.  The projection
            # is applied after conv1, so its backward runs first; it leaves its gradient in the box (for the
            # stride-2 blocks in compact form: only even positions are non-zero) and conv1's backward adds it
            # inside its own data-gradient kernel -- no zero fill, strided scatter and add over block-sized tensors.
            # `lateral_box` (FPN.forward): x -- the previous stage's output -- has a third reader, the FPN's lateral
            # convolution, whose backward runs before this block's.  It leaves its data gradient in that box too, conv1's
            # interleave pass adds both and, now that it sees x's WHOLE gradient, applies x's ReLU mask (gate_dx).
            box = GradBox()
            lateral, self.lateral_box = self.lateral_box, None
            if lateral is not None:
                lateral.taker = True
            out = conv_bn_act(x, self.conv1, self.bn1, relu=True, dx_add_from=box if lateral is None else (box, lateral),
                              gate_dx=lateral is not None)
            out = conv_bn_act(out, self.conv2, self.bn2, relu=True, gate_dx=True)
            residual = conv_bn_act(x, self.downsample[0], self.downsample[1], relu=False, dx_give_to=box)
            return conv_bn_act(out, self.conv3, self.bn3, relu=True, residual=residual, gate_dx=True)
        out = conv_bn_act(x, self.conv1, self.bn1, relu=True)
        out = conv_bn_act(out, self.conv2, self.bn2, relu=True, gate_dx=True)
        residual = x
        if self.downsample is not None:
            residual = conv_bn_act(x, self.downsample[0], self.downsample[1], relu=False)
        return conv_bn_act(out, self.conv3, self.bn3, relu=True, residual=residual, gate_dx=True)


class ResNet(nn.Module):
    """Trunk C1..C5 (lib/sub_module.py:38-82).  One row per stage: (name, planes, stride of its first block);
    block counts per architecture."""
    DEPTHS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3)}
    STAGES = (("C2", 64, 1), ("C3", 128, 2), ("C4", 256, 2), ("C5", 512, 2))

    def __init__(self, architecture, stage5=False):
        super(ResNet, self).__init__()
        if architecture not in self.DEPTHS:
            raise ValueError("architecture must be one of %s" % sorted(self.DEPTHS))
        self.stage5 = stage5
        self.block = Bottleneck
        self.layers = list(self.DEPTHS[architecture])
        self.inplanes = 64
        self.C1 = nn.Sequential(Conv2d(3, 64, kernel_size=7, stride=2, padding=3), _bn(64), nn.ReLU(inplace=True),
                                SamePad2d(kernel_size=3, stride=2), nn.MaxPool2d(kernel_size=3, stride=2))
        for (name, planes, stride), depth in zip(self.STAGES, self.layers):
            built = self.make_layer(Bottleneck, planes, depth, stride) if (name != "C5" or stage5) else None
            setattr(self, name, built)

    def stages(self):
        return [self.C1, self.C2, self.C3, self.C4, self.C5]

    def make_layer(self, block, planes, blocks, stride=1):
        out_ch = planes * block.expansion
        project = None
        if stride != 1 or self.inplanes != out_ch:
            project = nn.Sequential(Conv2d(self.inplanes, out_ch, kernel_size=1, stride=stride), _bn(out_ch))
        chain = [block(self.inplanes, planes, stride, project)]
        chain += [block(out_ch, planes) for _ in range(blocks - 1)]
        for b in chain[1:]:
            b.input_sole = True       # nn.Sequential: the previous block's output goes to this block only
        self.inplanes = out_ch
        return nn.Sequential(*chain)


class FPN(nn.Module):
    def __init__(self, config, C1, C2, C3, C4, C5, out_channels):
        super(FPN, self).__init__()
        self.config = config
        self.out_channels = out_channels
        self.C1, self.C2, self.C3, self.C4, self.C5 = C1, C2, C3, C4, C5
        self.P6 = nn.MaxPool2d(kernel_size=1, stride=2)
        oc = out_channels

        def smooth():
            return nn.Sequential(SamePad2d(kernel_size=3, stride=1, folded=True),
                                 Conv2d(oc, oc, kernel_size=3, stride=1, padding=1))

        self.P5_conv1 = Conv2d(2048, oc, kernel_size=1, stride=1)
        self.P5_conv2 = smooth()
        self.P4_conv1 = Conv2d(1024, oc, kernel_size=1, stride=1)
        self.P4_conv2 = smooth()
        self.P3_conv1 = Conv2d(512, oc, kernel_size=1, stride=1)
        self.P3_conv2 = smooth()
        self.P2_conv1 = Conv2d(256, oc, kernel_size=1, stride=1)
        self.P2_conv2 = smooth()
        self.ot = False
        if getattr(config.TRAIN, "FPN_OT_LOSS", False):
            from .OT_module import OptTrans
            self.ot = True
            base = int(config.DATA.IMAGE_SHAPE[0] / 4)
            self.p2_ot = OptTrans(config, ch_x=256, spatial_x=base / 2, spatial_y=base)
            self.p3_ot = OptTrans(config, ch_x=256, spatial_x=base / 4, spatial_y=base / 2)
            self.p4_ot = OptTrans(config, ch_x=256, spatial_x=base / 8, spatial_y=base / 4)

    def forward(self, x, mode='train'):
        bs = x.size(0)
        ot_loss = x.new_zeros(bs, 3)
        x = conv_bn_act(x, self.C1[0], self.C1[1], relu=True)       # C1 = conv, bn, relu, pad, maxpool
        x = _pad_maxpool(x, self.C1[3], self.C1[4])
        # c2..c4 are read by the next stage AND by their lateral convolution: the lateral hands its data gradient to the
        # next stage's first block (GradBox; see Bottleneck.forward), which is the only one left talking to autograd
        fuse = mode == 'train' and not self.ot and x.is_cuda and torch.is_grad_enabled() and _conv.GATES
        boxes = {}
        c2 = self.C2(x)
        c = {2: c2}
        for lvl, stage in ((3, self.C3), (4, self.C4), (5, self.C5)):
            first = stage[0] if (fuse and stage is not None and len(stage)) else None
            if isinstance(first, Bottleneck) and first.downsample is not None and tuple(first.conv1.stride) == (2, 2):
                boxes[lvl - 1] = first.lateral_box = GradBox()
            c[lvl] = stage(c[lvl - 1])
            if first is not None:
                first.lateral_box = None           # not picked up (the block took another path): nothing is given
        c3, c4, c5 = c[3], c[4], c[5]
        p5 = self.P5_conv1(c5)
        up = upsample2x
        if self.ot and mode == 'train':
            t4 = self.P4_conv1(c4)
            l0 = self.p4_ot(p5, t4)
            p4 = t4 + up(p5)
            t3 = self.P3_conv1(c3)
            l1 = self.p3_ot(p4, t3)
            p3 = t3 + up(p4)
            t2 = self.P2_conv1(c2)
            l2 = self.p2_ot(p3, t2)
            p2 = t2 + up(p3)
            ot_loss = torch.stack((l0, l1, l2), 1)
        else:
            # lateral 1x1 conv + top-down map in the conv epilogue (no separate add pass)
            lat = lambda m, c, top, box: conv2d(c, m.weight, m.bias, m.stride, m.padding, residual=up(top),
                                                dx_give_to=box)
            p4 = lat(self.P4_conv1, c4, p5, boxes.get(4))
            p3 = lat(self.P3_conv1, c3, p4, boxes.get(3))
            p2 = lat(self.P2_conv1, c2, p3, boxes.get(2))
        p5 = self.P5_conv2(p5)
        p4 = self.P4_conv2(p4)
        p3 = self.P3_conv2(p3)
        p2 = self.P2_conv2(p2)
        p6 = self.P6(p5)
        return [p2, p3, p4, p5, p6, ot_loss]


class _PatchRowsFn(torch.autograd.Function):
    """rows[r] = the 3 x 3 x C patch (tap-major: (dy, dx, c)) around the pixel of anchor (image[r], anchor[r]) on its
    pyramid level -- zeros outside the map and for padding rows (image[r] < 0): fi_pyramid_patch_rows_forward.  The
    training path of the RPN reads the pyramid ONLY here (RPN.forward_rows): backward adds the patch gradients into the
    level maps' gradients (a few thousand atomics per level) instead of running dense convolution gradients over maps
    whose output gradient is zero except at the sampled anchors.
    boxes[l] (a conv.GradBox or None): another reader of level l (the Dev make-up layer) leaves ITS gradient for the map
    there; this op is the taker and returns the sum to autograd."""

    @staticmethod
    def _levels(shapes):
        import ctypes
        n = len(shapes)
        return (ctypes.c_int * n)(*[s[2] for s in shapes]), (ctypes.c_int * n)(*[s[3] for s in shapes])

    @staticmethod
    def forward(ctx, image, anchor, per_loc, boxes, *maps):
        import ctypes
        _lib.require_cuda(image, anchor, *maps)
        maps = [m.contiguous().float() for m in maps]
        image, anchor = image.to(torch.int64).contiguous(), anchor.to(torch.int64).contiguous()
        R, C = image.numel(), maps[0].shape[1]
        out = torch.empty((R, 9 * C), device=image.device, dtype=torch.float32)
        shapes = [tuple(m.shape) for m in maps]
        hs, ws = _PatchRowsFn._levels(shapes)
        ptrs = (ctypes.c_void_p * len(maps))(*[m.data_ptr() for m in maps])
        with torch.cuda.device(image.device):
            _lib.check(_lib.load().fi_pyramid_patch_rows_forward(ptrs, hs, ws, len(maps), int(per_loc), _lib.ptr(image),
                                                                 _lib.ptr(anchor), R, C, _lib.ptr(out),
                                                                 _lib.current_stream()), "fi_pyramid_patch_rows_forward")
        ctx.save_for_backward(image, anchor)
        ctx.shapes, ctx.boxes, ctx.per_loc = shapes, boxes, int(per_loc)
        return out

    @staticmethod
    def backward(ctx, d):
        import ctypes
        image, anchor = ctx.saved_tensors
        d = d.contiguous().float()
        grads = []
        for l, shape in enumerate(ctx.shapes):
            base = None
            if ctx.boxes is not None and ctx.boxes[l] is not None:
                base, ctx.boxes[l].value = ctx.boxes[l].value, None
            if not (torch.is_tensor(base) and tuple(base.shape) == shape and base.is_contiguous() and
                    base.dtype == torch.float32):
                extra, base = base, d.new_zeros(shape)
                if torch.is_tensor(extra):
                    base += extra
            grads.append(base)
        hs, ws = _PatchRowsFn._levels(ctx.shapes)
        ptrs = (ctypes.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
        with torch.cuda.device(d.device):
            _lib.check(_lib.load().fi_pyramid_patch_rows_backward(_lib.ptr(d), ptrs, hs, ws, len(grads), ctx.per_loc,
                                                                  _lib.ptr(image), _lib.ptr(anchor), image.numel(),
                                                                  ctx.shapes[0][1], _lib.current_stream()),
                       "fi_pyramid_patch_rows_backward")
        return (None,) * 4 + tuple(grads)


class RPN(nn.Module):
    """Returns [rpn_class_logits [b, anchors, 2], rpn_probs, rpn_bbox [b, anchors, 4]]."""

    _probe = None       # a dict while workflow.compare_backward_forms records what decides the ReLU masks of the two forms

    def __init__(self, anchors_per_location, anchor_stride, input_ch):
        super(RPN, self).__init__()
        self.anchor_stride = anchor_stride
        self.input_ch = input_ch
        self.padding = SamePad2d(kernel_size=3, stride=anchor_stride, folded=(anchor_stride == 1))
        self.conv_shared = Conv2d(input_ch, 512, kernel_size=3, stride=anchor_stride,
                                     padding=1 if anchor_stride == 1 else 0)
        self.relu = nn.ReLU(inplace=True)
        self.conv_class = Conv2d(512, 2 * anchors_per_location, kernel_size=1, stride=1)
        self.softmax = nn.Softmax(dim=2)
        self.conv_bbox = Conv2d(512, 4 * anchors_per_location, kernel_size=1, stride=1)

    def forward(self, x, grad_box=None):
        """grad_box: a GradBox in which a later reader of x (the Dev make-up layer) leaves ITS data gradient for x;
        the shared convolution's data-gradient kernel adds it (only when the padding is folded: x itself is read)."""
        c = self.conv_shared
        xp = self.padding(x)
        x = conv_bias_relu(xp, c.weight, c.bias, c.stride, c.padding,
                           dx_add_from=grad_box if xp is x else None)                    # conv + bias + ReLU, one launch
        if self._probe is not None:       # workflow.compare_backward_forms: the dense kernel's ReLU output per level
            self._probe.setdefault("dense_y", []).append(x.detach())
        # the two 1x1 heads as ONE convolution over their stacked filters (every output channel is computed
        # exactly as before): the 512-channel map is read once instead of twice in forward, data gradient and
        # weight gradient, and autograd has no two data gradients to add
        ncls = self.conv_class.weight.shape[0]
        both = conv2d(x, torch.cat((self.conv_class.weight, self.conv_bbox.weight), 0),
                      torch.cat((self.conv_class.bias, self.conv_bbox.bias), 0), gate_dx=True)    # x's only reader
        logits = both[:, :ncls].permute(0, 2, 3, 1).contiguous().view(x.size(0), -1, 2)
        probs = self.softmax(logits)
        bbox = both[:, ncls:].permute(0, 2, 3, 1).contiguous().view(x.size(0), -1, 4)
        return [logits, probs, bbox]

    @staticmethod
    def dense_at_rows(dense_maps, image, anchor, valid, per_loc):
        """[R, C]: the per-level maps `dense_maps` ([B, C, H_l, W_l], pyramid order) at rows (image[r], anchor[r]) of the
        level-major anchor list with per_loc anchors per pixel (lib/layers.py:41-44); zeros where valid[r] is False."""
        a = anchor.clamp(min=0)
        out = dense_maps[0].new_zeros((a.numel(), dense_maps[0].shape[1]))
        base = 0
        for y in dense_maps:
            H, W = y.shape[2], y.shape[3]
            n = H * W * per_loc
            here = valid & (a >= base) & (a < base + n)
            pix = ((a - base) // per_loc).clamp(0, H * W - 1)
            got = y[image.clamp(min=0), :, pix // W, pix % W]
            out = torch.where(here.unsqueeze(1), got, out)
            base += n
        return out

    def forward_rows(self, maps, image, anchor, valid, grad_boxes=None):
        """The RPN's outputs at SELECTED anchors only: (logits [R, 2], bbox [R, 4]) for rows (image[r], anchor[r]) of the
        level-major anchor list (lib/layers.py:41-44), zeros where valid[r] is False.

        The RPN losses read 256 sampled anchors per image out of ~262 000 (lib/layers.py:808-861), so the gradient of
        the dense outputs is zero everywhere else and the dense backward of the shared 3x3 convolution -- the largest
        convolution of the step next to the mask head's -- multiplies zeros (13 of 151 ms).  The dense forward still runs
        (under no_grad: the proposal layer needs every anchor's score); for the losses the same two layers are evaluated
        on the selected pixels' 3 x 3 patches as matrix products (conv.linear: [R, 9*256] . W_shared^T -> ReLU -> . W_heads^T),
        whose backward IS the convolution's backward restricted to the rows that have a gradient; the patch gather's
        backward scatters the input gradient into the level maps (_PatchRowsFn)."""
        assert self.anchor_stride == 1, "row form: stride-1 RPN only"
        per_loc = self.conv_class.weight.shape[0] // 2
        rows_image = torch.where(valid, image, torch.full_like(image, -1))
        patches = _PatchRowsFn.apply(rows_image, anchor.clamp(min=0), per_loc, grad_boxes, *maps)
        k = anchor.clamp(min=0) % per_loc          # every level's first anchor index is a multiple of per_loc
        a = anchor
        cs = self.conv_shared
        ws = cs.weight.permute(0, 2, 3, 1).reshape(cs.weight.shape[0], -1)           # [512, 9*256]: a view of the
        z = linear(patches, ws, cs.bias)                                              # channels-last parameter
        if self._probe is not None:       # workflow.compare_backward_forms: the row form's pre-activations
            self._probe.update(rows=(rows_image.detach(), anchor.detach(), valid.detach(), per_loc),
                               patches=patches.detach(), z_rows=z.detach())
        if self._probe is not None and self._probe.get("mask_from_dense"):
            # check_backward_forms' replay of a verified ReLU-boundary event: the mask bits of the DENSE kernel (whose
            # output the dense form's backward masks with) instead of the row form's own -- the two forms then differ
            # by rounding only, whatever side of zero a pre-activation at rounding distance fell on
            y = z * (self.dense_at_rows(self._probe["dense_y"], rows_image, anchor, valid, per_loc) > 0).to(z.dtype)
        else:
            y = torch.relu(z)
        heads = linear(y, torch.cat((self.conv_class.weight, self.conv_bbox.weight), 0).flatten(1),
                       torch.cat((self.conv_class.bias, self.conv_bbox.bias), 0))     # [R, 2*per_loc + 4*per_loc]
        ncls = 2 * per_loc
        col = torch.arange(2, device=a.device).unsqueeze(0)
        logits = torch.gather(heads, 1, 2 * k.unsqueeze(1) + col)
        bbox = torch.gather(heads, 1, ncls + 4 * k.unsqueeze(1) + torch.arange(4, device=a.device).unsqueeze(0))
        keep = valid.unsqueeze(1).float()
        return logits * keep, bbox * keep


class Dev(nn.Module):
    """The intertwiner RoI stage (lib/sub_module.py:286-692), 'beta' structure.

    forward(x, rois, roi_cls_gt) -> pooled [bs*R,256,7,7], mask [bs*R,256,14,14], feat_out.
    One pyramid launch per crop size replaces the per-level loops; outputs are in the
    original RoI order by construction (the reference scatters them back, :644-662).
    """

    def __init__(self, config, depth):
        super(Dev, self).__init__()
        self.depth = depth
        self.use_dev = config.DEV.SWITCH
        self.pool_size = config.MRCNN.POOL_SIZE
        self.mask_pool_size = config.MRCNN.MASK_POOL_SIZE
        self.image_shape = config.DATA.IMAGE_SHAPE
        self.num_classs = config.DATASET.NUM_CLASSES
        self.config = config
        self.structure = config.DEV.STRUCTURE
        self.roi_type = config.ROIS.METHOD
        self.roi_spatial_scale = [1. / 4, 1. / 8, 1. / 16, 1. / 32]
        if self.use_dev:
            self.feat_pool_size = config.DEV.FEAT_BRANCH_POOL_SIZE
            assert self.feat_pool_size % 2 == 0, 'pool size of feature branch has to be even'
            if not config.DEV.DIS_UPSAMPLER:
                if config.DEV.UPSAMPLE_FAC == 1.:
                    conv_opt = Conv2d(depth, depth, kernel_size=3, padding=1)
                elif config.DEV.UPSAMPLE_FAC == 2.:
                    conv_opt = nn.ConvTranspose2d(depth, depth, kernel_size=3, stride=2, padding=1,
                                                  output_padding=1)
                n_up = 4 if config.DEV.MULTI_UPSAMPLER else 1
                self.upsample = nn.ModuleList()
                for _ in range(n_up):
                    self.upsample.append(nn.Sequential(conv_opt, nn.BatchNorm2d(depth), nn.ReLU(inplace=True)))
            if not config.DEV.BASELINE:
                k = int(self.feat_pool_size / 2)
                self.feat_extract = nn.Sequential(
                    Conv2d(depth, 512, kernel_size=3, padding=1, stride=2), nn.BatchNorm2d(512),
                    nn.ReLU(inplace=True),
                    Conv2d(512, 1024, kernel_size=k, stride=1), nn.BatchNorm2d(1024), nn.ReLU(inplace=True),
                    Conv2d(1024, 1024, kernel_size=1, stride=1), nn.BatchNorm2d(1024), nn.ReLU(inplace=True),
                )
                self.feat_extract[3].full_window = True          # k x k kernel on the k x k map: a GEMM
                if config.DEV.LOSS_CHOICE in ('l2', 'l1'):
                    self.last_op = nn.Sigmoid()
                elif config.DEV.LOSS_CHOICE == 'kl':
                    self.last_op = nn.Softmax(dim=1)
                if config.DEV.BIG_SUPERVISE:                 # lib/sub_module.py:352-353
                    self.big_fc_layer = nn.Linear(1024, self.num_classs)
            if config.DEV.DIS_UPSAMPLER:
                raise NotImplementedError("DEV.DIS_UPSAMPLER=True builds no make-up layer, and the 'beta' forward "
                                          "(lib/sub_module.py:550) then fails in the reference as well")
            if config.DEV.ASSIGN_BOX_ON_ALL_SCALE:
                raise NotImplementedError("DEV.ASSIGN_BOX_ON_ALL_SCALE (area-threshold level assignment, "
                                          "lib/sub_module.py:441-454) is not built; every shipped config sets it False")

    def _feat_extract(self, v):
        fe = self.feat_extract
        v = conv_bn_act(v, fe[0], fe[1], relu=True)
        v = conv_bn_act(v, fe[3], fe[4], relu=True)      # full-window conv: library GEMM + affine BN
        return conv_bn_act(v, fe[6], fe[7], relu=True)

    @staticmethod
    def _find_big_box2(level, roi_lvl):
        """RoIs that act as 'big' supervision at pyramid level `level` (:366-378)."""
        return roi_lvl > level if level < 5 else torch.zeros_like(roi_lvl, dtype=torch.bool)

    def _crop(self, maps, boxes, box_ind, level, size, grad_group=None):
        if self.roi_type == 'roi_align':
            return pyramid_crop_and_resize(maps, boxes, box_ind, level, size, size, grad_group=grad_group)
        # roi_pool: per level (the RoIPool kernel takes pixel boxes and a per-level scale)
        out = boxes.new_zeros(boxes.size(0), maps[0].size(1), size, size)
        pix = self._make_roi_pool_box_input(boxes, box_ind)
        for i, lvl in enumerate(range(2, 6)):
            sel = torch.nonzero(level == lvl).view(-1)
            if sel.numel():
                out[sel] = RoIPoolFunction(size, size, self.roi_spatial_scale[i])(maps[i], pix[sel])
        return out

    def _make_roi_pool_box_input(self, boxes, box_ind):
        b = boxes * float(self.image_shape[0])      # square images only
        return torch.stack([box_ind.float(), b[:, 1], b[:, 0], b[:, 3], b[:, 2]], dim=1)

    def make_up_maps(self, x, give_to=None, take_from=None):
        """The make-up layer (lib/sub_module.py:308-325, applied at :549-557) on every pyramid level.  It does not
        depend on the RoIs, so the caller may run it while the RoIs are still being generated.
        give_to / take_from: per level, GradBoxes of the level map's other readers (MaskRCNN.forward): the layer's data
        gradient for the map -- plus what the big-box crop of forward() left in take_from -- goes to give_to's taker
        (the RPN's shared convolution) instead of to autograd: one kernel writes the map's whole gradient."""
        cfg = self.config

        def make_up(i, m):
            seq = self.upsample[i if cfg.DEV.MULTI_UPSAMPLER else 0]
            if isinstance(seq[0], Conv2d):
                # these maps feed only the two crops of forward(): written channels-last by the conv epilogue so
                # that RoIAlign reads (and its backward adds) whole cache lines per tap
                fused = m.is_cuda and not seq[1].training and seq[1].track_running_stats and m.requires_grad and \
                    torch.is_grad_enabled() and m.shape[2] * m.shape[3] > 1
                give = give_to[i] if (give_to and fused and give_to[i] is not None and give_to[i].taker) else None
                take = take_from[i] if (take_from and fused and take_from[i] is not None) else None
                if take is not None:
                    take.taker = True
                return conv_bn_act(m, seq[0], seq[1], relu=True, channels_last_out=(self.roi_type == 'roi_align'),
                                   dx_add_from=take, dx_give_to=give)
            return seq(m)
        return [make_up(i, m) for i, m in enumerate(x)]

    def level_info(self, rois):
        """(level [bs*R] int, counts_ready): the pyramid level of every RoI (lib/sub_module.py:405-410) and a
        function returning the RoI counts of levels 2..5 on the HOST -- the only data-dependent SHAPES of the stage
        (how many 'small' rows feed feat_extract, how many 'big' boxes each level sees).  They come back in ONE small
        read; the copy is started here on a side stream and awaited in forward() only after work that does not
        depend on it has been enqueued, so the device keeps working while the host waits.  (The reference
        synchronises per level: nonzero / .any() at lib/sub_module.py:456, 475, 483, 541.)"""
        boxes = rois.reshape(-1, 4)
        level = roi_level(boxes, float(self.image_shape[0] * self.image_shape[1]), self.config.ROIS.ASSIGN_ANCHOR_BASE)
        lv = torch.arange(2, 6, device=level.device, dtype=level.dtype)
        per_level = (level.unsqueeze(0) == lv.unsqueeze(1)).sum(1)                 # [n2, n3, n4, n5]
        if self.static_shapes(rois):
            return level, per_level           # stays on the device: forward() never learns the counts (round 4)
        counts_ready = _lib.async_host_read(per_level) if level.is_cuda else (lambda: per_level)
        return level, counts_ready

    def static_shapes(self, rois):
        """True: the stage runs without its one host read (round 4).  The RoI counts per level size the feature
        extractor's two batches in the reference (nonzero / .any() per level, lib/sub_module.py:456-541).  Here the small
        branch takes ALL RoIs (the level-5 rows ride along and are masked: at most one batch tile more than before), and
        the big branch a batch of static capacity 3 * RoIs -- every (RoI, lower level) pair there can be -- whose live
        count n3 + 2 n4 + 3 n5 stays on the device: its kernels skip the tiles past it (fi_conv2d_forward_live,
        fi_gemm_nt_rows and their 16-bit twins), the filler rows are never written and carry class 0.  Needs the graph-less big
        branch (the defaults DEV.BIG_FEAT_DETACH, no BIG_SUPERVISE)."""
        cfg = self.config
        on = _STATIC_DEV if _STATIC_DEV is not None else getattr(cfg.MODEL, "CONV_PRECISION", "fp32") == "fp32"
        # (the class means of the three levels are ONE launch over 3 K classes; fi_class_mean holds at most 248)
        return bool(on and rois.is_cuda and self.use_dev and not cfg.DEV.BASELINE and cfg.DEV.BIG_FEAT_DETACH and
                    not cfg.DEV.BIG_SUPERVISE and self.roi_type == 'roi_align' and 3 * cfg.DATASET.NUM_CLASSES <= 248)

    _PERM = {}

    @classmethod
    def _front_permutation(cls, bs, R, P, device):
        """(perm, inv): RoI slots (b, s < P) of every image first -- in (b, s) order --, then the others; inv[perm[j]] = j."""
        key = (bs, R, P, str(device))
        t = cls._PERM.get(key)
        if t is None:
            slot = torch.arange(bs * R).view(bs, R)
            perm = torch.cat((slot[:, :P].reshape(-1), slot[:, P:].reshape(-1)))
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(bs * R)
            t = cls._PERM[key] = (perm.to(device), inv.to(device))
        return t

    def forward(self, x, rois, roi_cls_gt=None, up_maps=None, level_info=None, raw_grad_boxes=None, mask_grad_box=None,
                mask_front=None):
        """raw_grad_boxes: per level, the GradBox make_up_maps(take_from=...) takes from -- the big-box crop of the RAW
        level maps leaves its map gradients there.  mask_grad_box: the mask head (the second reader of the 14 x 14
        crops, applied after this stage) leaves its input gradient there; the row gather in front of the feature
        extractor adds its own rows into it (conv.take_rows)."""
        cfg = self.config
        bs, R = rois.size(0), rois.size(1)
        boxes = rois.reshape(-1, 4)
        box_ind = torch.arange(bs, device=rois.device, dtype=torch.int32).repeat_interleave(R)
        level, counts_ready = level_info if level_info is not None else self.level_info(rois)
        # mask_front = P: the 14 x 14 crops are RETURNED with the RoI slots s < P of every image first (rows b*P + s), the
        # other slots behind them: the mask head's two batches (MaskRCNN.forward) are then two contiguous views of the
        # crop tensor instead of two strided copies of it (and of a zero-filled, copied gradient on the way back)
        front = mask_front if (mask_front and self.use_dev and 0 < mask_front < R and rois.is_cuda) else None

        if not self.use_dev:
            pooled = self._crop(x, boxes, box_ind, level, self.pool_size)
            mask = self._crop(x, boxes, box_ind, level, self.mask_pool_size)
            return pooled, mask, None
        if self.structure != 'beta':
            raise NotImplementedError("only DEV.STRUCTURE='beta' executes in the reference (SURVEY Q9)")

        train_phase = roi_cls_gt is not None
        # make-up layer on every level (unless the caller already ran it), then ONE launch per crop size over all levels
        if up_maps is None:
            up_maps = self.make_up_maps(x)
        static = torch.is_tensor(counts_ready)          # level_info(): static_shapes() -- the counts stay on the device
        if not static:
            n2, n3, n4, n5 = (int(v) for v in counts_ready().tolist())
        group = CropGradGroup()      # both crops' gradients accumulate in ONE set of buffers (no add pass per level)
        pooled = self._crop(up_maps, boxes, box_ind, level, self.pool_size, group)
        if front:
            perm, inv = self._front_permutation(bs, R, front, rois.device)
            mask_and_feat = self._crop(up_maps, boxes[perm], box_ind[perm], level[perm], self.mask_pool_size, group)
        else:
            inv = None
            mask_and_feat = self._crop(up_maps, boxes, box_ind, level, self.mask_pool_size, group)
        if cfg.DEV.BASELINE:
            return pooled, mask_and_feat, []

        # 'small' features of the RoIs on levels 2..4 (levels with meta loss, :434-435), written
        # level-major like the reference's small_output_all / small_gt_all (:583-598): a stable sort by
        # level puts them first, in (level, original index) order
        total_box = bs * R
        if static:
            return self._forward_static(x, boxes, box_ind, level, counts_ready, roi_cls_gt, pooled, mask_and_feat, inv,
                                        mask_grad_box, total_box)
        n_small = n2 + n3 + n4
        # The feature extractor's fully connected stages are matrix products over the rows fed to it; their kernels
        # take row counts that are multiples of 64, so up to 63 further RoIs (the first level-5 ones in the sorted
        # order) ride along.  Their outputs are never read: every use below is restricted to the first n_small rows
        # or masked by level.
        n_rows = min((n_small + 63) // 64 * 64, total_box)
        order = torch.sort(level, stable=True)[1][:n_rows]
        small_output = self._feat_extract(take_rows(mask_and_feat, order if inv is None else inv[order], mask_grad_box))
        if cfg.DEV.LOSS_CHOICE != 'ot':
            small_output = self.last_op(small_output)
        small_output = small_output.view(n_rows, -1)
        small_output_all = small_output.new_zeros(total_box, small_output.size(1))
        small_output_all[:n_small] = small_output[:n_small]
        small_gt_all = small_output.new_zeros(total_box)
        if not train_phase:
            small_gt_all[:n_small] = 1
            return pooled, mask_and_feat, [small_output_all, small_gt_all]

        gt = roi_cls_gt.reshape(-1).to(torch.int32)
        small_gt_all[:n_small] = gt[order[:n_small]].float()
        lvl_o = level[order]
        gt_o = gt[order]
        K = self.num_classs
        small_feat, small_cnt, big_feat, big_cnt = [], [], [], []
        for lvl in (2, 3, 4):
            f, c = class_mean(small_output, torch.where(lvl_o == lvl, gt_o, torch.zeros_like(gt_o)), K)
            small_feat.append(f)
            small_cnt.append(c)
        # 'big' boxes: RoIs of higher levels pooled 14x14 from the RAW level map (:498-507); level l sees
        # the RoIs of every higher level, so an RoI of level 5 appears three times
        def big_branch():
            big_feat, big_cnt = [], []
            n_big = {2: n3 + n4 + n5, 3: n4 + n5, 4: n5}
            has_small = {2: n2 > 0, 3: n3 > 0, 4: n4 > 0}
            big_sel, big_lvl = [], []
            for lvl in (2, 3, 4):
                idx = torch.nonzero_static(level > lvl, size=n_big[lvl]).view(-1)
                big_sel.append(idx)
                big_lvl.append(torch.full_like(idx, lvl, dtype=torch.int32))
            n_big_rows = sum(n_big.values())
            n_pad = (-n_big_rows) % 64
            if n_big_rows and n_pad:
                # row count of the big branch rounded up to a multiple of 64 (see above): the filler rows carry level 0,
                # which no pyramid level matches -- zero crops, no statistics, no loss
                big_sel.append(torch.zeros(n_pad, dtype=big_sel[0].dtype, device=level.device))
                big_lvl.append(torch.zeros(n_pad, dtype=torch.int32, device=level.device))
            big_idx = torch.cat(big_sel)
            big_level = torch.cat(big_lvl)
            big_loss = []
            # the big branch needs a graph only when its class means are not detached or when it is
            # supervised by its own classifier (:531-535: the CE loss reaches feat_extract either way)
            with torch.set_grad_enabled(torch.is_grad_enabled() and
                                        (not cfg.DEV.BIG_FEAT_DETACH or cfg.DEV.BIG_SUPERVISE)):
                if big_idx.numel():
                    big_pooled = self._crop(x, boxes[big_idx], box_ind[big_idx], big_level, self.feat_pool_size,
                                            CropGradGroup(give_to=raw_grad_boxes) if raw_grad_boxes else None)
                    big_raw = self._feat_extract(big_pooled)
                    big_out = self.last_op(big_raw) if cfg.DEV.LOSS_CHOICE != 'ot' else big_raw
                    big_out = big_out.view(big_idx.numel(), -1)
                    big_raw = big_raw.view(big_idx.numel(), -1)
                else:
                    big_out = big_raw = small_output.new_zeros(0, small_output.size(1))
                big_gt = gt[big_idx]
                if cfg.DEV.BIG_SUPERVISE:
                    ce = F.cross_entropy(linear(big_raw, self.big_fc_layer.weight, self.big_fc_layer.bias), big_gt.long(),
                                         reduction='none') \
                        if big_idx.numel() else big_raw.new_zeros(0)
                for i, lvl in enumerate((2, 3, 4)):
                    # a level without small boxes contributes no big statistics either (:456-467)
                    at_lvl = big_level == lvl
                    g = torch.where(at_lvl, big_gt, torch.zeros_like(big_gt)) if has_small[lvl] else torch.zeros_like(big_gt)
                    f, c = class_mean(big_out, g, K)
                    big_feat.append(f)
                    big_cnt.append(c)
                    if cfg.DEV.BIG_SUPERVISE and has_small[lvl]:     # mean CE over the level's big boxes (:531-535)
                        w = at_lvl.float()
                        big_loss.append(((ce * w).sum() / w.sum().clamp(min=1)).view(1))
                    else:
                        big_loss.append(small_output.new_zeros(1))
            bf = torch.stack(big_feat).unsqueeze(0)
            if cfg.DEV.BIG_FEAT_DETACH:
                bf = bf.detach()
            return bf, torch.stack(big_cnt).unsqueeze(0), torch.stack(big_loss).unsqueeze(0)

        # The big branch builds no graph when its class means are detached and it has no classifier of its own (the
        # defaults): nothing on the main stream reads it before the meta loss, so it runs on the third stream, next to
        # the small branch, the box head and the mask head (crop + feat_extract on ~1300 rows: ~3 ms of kernels).
        # workflow.compute_loss evaluates the meta loss on that same stream, behind it; a consumer on another stream
        # waits for self.big_done.
        self.big_done = None
        if _BIG_SIDE and level.is_cuda and cfg.DEV.BIG_FEAT_DETACH and not cfg.DEV.BIG_SUPERVISE:
            fork = torch.cuda.Event()
            fork.record(torch.cuda.current_stream(level.device))
            side3 = _lib.side_stream3(level.device)
            for t in (boxes, box_ind, level, gt):
                t.record_stream(side3)
            ready = _lib.run_on_side_stream(big_branch, after=fork)
            bf, bcnt, bloss = ready.out
            self.big_done = ready.done
        else:
            bf, bcnt, bloss = big_branch()
        feat_out = [bf, bcnt,
                    torch.stack(small_feat).unsqueeze(0), torch.stack(small_cnt).unsqueeze(0),
                    bloss, small_output_all, small_gt_all]
        return pooled, mask_and_feat, feat_out


    def _feat_extract_live(self, v, live):
        """_feat_extract on a batch of static capacity whose first live[0] rows are real (no graph)."""
        fe = self.feat_extract
        v = conv_bn_act(v, fe[0], fe[1], relu=True, live=live)
        v = conv_bn_act(v, fe[3], fe[4], relu=True, live=live)
        return conv_bn_act(v, fe[6], fe[7], relu=True, live=live)

    @staticmethod
    def _static_index_tensors(level, gt, K, cap):
        """The index side of _forward_static as tensor operations (the specification of fi_dev_stage_index; CPU path and
        the tests' reference): see include/fi_capi.h for the outputs."""
        N = level.numel()
        order = torch.sort(level, stable=True)[1]                       # level-major, original index inside a level
        lvl_o = level[order]
        small_on = lvl_o <= 4                                           # the rows the reference feeds (:583-598)
        gt = gt if gt is not None else torch.zeros_like(level)
        gt_o = gt[order]
        zero = torch.zeros_like(gt_o)
        small_gt = torch.where(small_on, gt_o, zero).float()
        small_cls = torch.where(small_on & (gt_o > 0), (lvl_o - 2) * K + gt_o, zero).to(torch.int32)
        # every (RoI, lower level) pair in (level, RoI) order, compacted to the front of the big batch
        pairs = torch.stack([level > lvl for lvl in (2, 3, 4)])                              # [3, N]
        flat = torch.nonzero_static(pairs.reshape(-1), size=cap, fill_value=-1).view(-1)
        valid = flat >= 0
        flat_c = flat.clamp(min=0)
        big_idx = flat_c % N
        # (level -1: a row past the live count -- the crop does not even write it, nothing downstream reads it)
        big_level = torch.where(valid, 2 + flat_c // N, torch.full_like(flat_c, -1)).to(torch.int32)
        counts = torch.stack([(level == lvl).sum() for lvl in (2, 3, 4, 5)]).to(torch.int32)
        has_small = counts[:3] > 0                                      # a level without small boxes contributes no big
        big_gt = gt[big_idx]                                            # statistics either (:456-467)
        lv = (big_level - 2).clamp(0, 2).long()
        use = valid & has_small[lv] & (big_gt > 0)
        big_cls = torch.where(use, lv.to(torch.int32) * K + big_gt.to(torch.int32), torch.zeros_like(big_level))
        live = pairs.sum().to(torch.int32).view(1)
        return order, small_cls, small_gt, small_on, big_idx, big_level, big_cls, live

    def _static_index(self, level, gt, K, cap):
        if not (_INDEX_KERNEL and level.is_cuda):
            return self._static_index_tensors(level, gt, K, cap)
        N, dev = level.numel(), level.device
        level = level.contiguous()
        order = torch.empty(N, dtype=torch.int64, device=dev)
        small_cls = torch.empty(N, dtype=torch.int32, device=dev)
        small_gt = torch.empty(N, dtype=torch.float32, device=dev)
        small_on = torch.empty(N, dtype=torch.bool, device=dev)
        big_idx = torch.empty(cap, dtype=torch.int64, device=dev)
        big_level = torch.empty(cap, dtype=torch.int32, device=dev)
        big_cls = torch.empty(cap, dtype=torch.int32, device=dev)
        counts = torch.empty(5, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().fi_dev_stage_index(_lib.ptr(level), _lib.ptr(gt.contiguous() if gt is not None else None), N, K,
                                                      cap, _lib.ptr(order), _lib.ptr(small_cls), _lib.ptr(small_gt),
                                                      _lib.ptr(small_on), _lib.ptr(big_idx), _lib.ptr(big_level),
                                                      _lib.ptr(big_cls), _lib.ptr(counts), _lib.current_stream()),
                       "fi_dev_stage_index")
        return order, small_cls, small_gt, small_on, big_idx, big_level, big_cls, counts[4:5]

    def _forward_static(self, x, boxes, box_ind, level, per_level, roi_cls_gt, pooled, mask_and_feat, inv, mask_grad_box,
                        total_box):
        """The part of forward() behind the two crops with shapes that do not depend on the RoIs (see static_shapes).
        One launch (fi_dev_stage_index) makes every index the stage needs; the class means of the three levels are ONE
        launch over 3 K classes (class = (level - 2) K + gt: the same rows in the same order per class as three launches)."""
        cfg = self.config
        K = self.num_classs
        dev = level.device
        train_phase = roi_cls_gt is not None
        gt = roi_cls_gt.reshape(-1).to(torch.int32) if train_phase else None
        cap = (3 * total_box + 63) // 64 * 64            # whole row tiles of the fully connected stages: no padding copy
        order, small_cls, small_gt_all, small_on, big_idx, big_level, big_cls, live = self._static_index(level, gt, K, cap)
        small_output = self._feat_extract(take_rows(mask_and_feat, order if inv is None else inv[order], mask_grad_box))
        if cfg.DEV.LOSS_CHOICE != 'ot':
            small_output = self.last_op(small_output)
        small_output = small_output.view(total_box, -1)
        small_output_all = torch.where(small_on.unsqueeze(1), small_output, torch.zeros_like(small_output))
        if not train_phase:
            return pooled, mask_and_feat, [small_output_all, small_on.float()]
        F_ = small_output.size(1)

        def per_level_stats(feat, cnt):          # [F, 3K], [1, 3K] -> [1, 3, F, K], [1, 3, 1, K] (views)
            return feat.view(F_, 3, K).permute(1, 0, 2).unsqueeze(0), cnt.view(1, 3, 1, K)

        small_feat, small_cnt = per_level_stats(*class_mean(small_output, small_cls, 3 * K))

        def big_branch():
            with torch.no_grad():
                big_pooled = self._crop(x, boxes[big_idx], box_ind[big_idx], big_level, self.feat_pool_size, None)
                big_raw = self._feat_extract_live(big_pooled, live)
                big_out = self.last_op(big_raw) if cfg.DEV.LOSS_CHOICE != 'ot' else big_raw
                bf, bcnt = per_level_stats(*class_mean(big_out.view(cap, -1), big_cls, 3 * K))
                return bf.detach(), bcnt, small_output.new_zeros(1, 3, 1)

        self.big_done = None
        if _BIG_SIDE and level.is_cuda:
            fork = torch.cuda.Event()
            fork.record(torch.cuda.current_stream(dev))
            # (reads=: every tensor of THIS stream that the closure reads on the third one -- without the mark the allocator
            # may hand a freed block to the next main-stream tensor while the third stream still reads it)
            ready = _lib.run_on_side_stream(big_branch, after=fork, reads=(boxes, box_ind, big_idx, big_level, big_cls, live))
            bf, bcnt, bloss = ready.out
            self.big_done = ready.done
        else:
            bf, bcnt, bloss = big_branch()
        feat_out = [bf, bcnt, small_feat, small_cnt, bloss, small_output_all, small_gt_all]
        return pooled, mask_and_feat, feat_out


class Classifier(nn.Module):
    """Box head (lib/sub_module.py:698-747): full-window 7x7 conv, 1x1 conv, class and box linears."""

    def __init__(self, depth, num_classes, pool_size, config):
        super(Classifier, self).__init__()
        self.depth, self.pool_size, self.num_classes, self.config = depth, pool_size, num_classes, config
        _conv_bn(self, 1, depth, 1024, pool_size)
        self.conv1.full_window = pool_size > 1
        _conv_bn(self, 2, 1024, 1024, 1)
        self.relu = nn.ReLU(inplace=True)
        self.linear_class = nn.Linear(1024, num_classes)
        self.softmax = nn.Softmax(dim=1)
        self.linear_bbox = nn.Linear(1024, num_classes * 4)

    def forward(self, x, small_feat_input=None, small_gt_index=None, mode='train'):
        x = conv_bn_act(x, self.conv1, self.bn1, relu=True)
        cfg = self.config
        if cfg.DEV.SWITCH and cfg.DEV.CLS_MERGE_FEAT and cfg.DEV.STRUCTURE == 'beta' and small_feat_input is not None:
            # lib/sub_module.py:724-730.  Row i of small_feat_input is the i-th SMALL box in level-major
            # order, not RoI i -- the reference adds them row by row all the same; mirrored.
            on = (small_gt_index > 0).float()
            if cfg.DEV.CLS_MERGE_MANNER == 'simple_add':
                x = x + (small_feat_input * on.unsqueeze(1)).view(x.size(0), x.size(1), 1, 1)
            else:                                           # 'linear_add'
                wgt = (on * cfg.DEV.CLS_MERGE_FAC).view(x.size(0), 1, 1, 1)
                x = (1 - wgt) * x + wgt * small_feat_input.view(x.size(0), x.size(1), 1, 1)
        x = conv_bn_act(x, self.conv2, self.bn2, relu=True)
        x = x.view(-1, 1024)
        if x.is_cuda:
            # the two heads read the same rows: ONE matrix product over the stacked weights (81 + 324 output columns,
            # padded once to the kernel's 128-column tiles instead of 128 + 384), split afterwards
            nc = self.linear_class.weight.shape[0]
            heads = linear(x, torch.cat((self.linear_class.weight, self.linear_bbox.weight), 0),
                           torch.cat((self.linear_class.bias, self.linear_bbox.bias), 0))
            logits, bbox = heads[:, :nc].contiguous(), heads[:, nc:].contiguous()
        else:
            logits = linear(x, self.linear_class.weight, self.linear_class.bias)
            bbox = linear(x, self.linear_bbox.weight, self.linear_bbox.bias)
        probs = self.softmax(logits)
        bbox = bbox.view(bbox.size(0), -1, 4)
        return [logits, probs, bbox]


class Mask(nn.Module):
    """Mask head (lib/sub_module.py:750-787): four 3x3 conv+BN+ReLU, 2x2 stride-2 deconv, 1x1 conv to classes."""

    def __init__(self, depth, num_classes):
        super(Mask, self).__init__()
        self.depth, self.num_classes = depth, num_classes
        self.padding = SamePad2d(kernel_size=3, stride=1, folded=True)
        for index in (1, 2, 3, 4):
            _conv_bn(self, index, depth if index == 1 else 256, 256, 3, 1, 1, eps=0.001, momentum=0.1)
        self.deconv = ConvTranspose2x2(256, 256, kernel_size=2, stride=2)
        self.conv5 = Conv2d(256, num_classes, kernel_size=1, stride=1)
        self.sigmoid = nn.Sigmoid()
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x, shuffled=True, activate=True, input_grad_box=None, select_class=None):
        """shuffled=True: [N, K, 28, 28] as the reference (lib/sub_module.py:769-787).
        shuffled=False: the same values as [N, 2, 2, K, 14, 14] with out[n,k,2h+a,2w+b] = u[n,a,b,k,h,w]
        (training: the loss gathers the class channel first and shuffles only that).
        activate=False (training, with shuffled=False): conv5's LOGITS; the sigmoid (:786) is applied by the
        loss to the one class channel per RoI it reads (compute_mrcnn_mask_loss_unshuffled(from_logits=True)) --
        an elementwise op commuted with a gather: same values, 1/81 of the elements.
        select_class [N] (training, with shuffled=False and activate=False): return only the logits of class
        select_class[n] for RoI n, [N, 2, 2, 14, 14] -- conv5 still evaluates every class (the reference's schedule), but
        its backward then knows that the gradient has one non-zero channel per RoI (conv.conv1x1_class_rows)."""
        # input_grad_box: x has another reader whose backward runs later and takes the gradient from there (Dev.forward)
        give = input_grad_box if (input_grad_box is not None and input_grad_box.taker and _fused_path(x, self.bn1)) else None
        x = conv_bn_act(x, self.conv1, self.bn1, relu=True, dx_give_to=give)
        x = conv_bn_act(x, self.conv2, self.bn2, relu=True, gate_dx=True)      # each layer is the only reader of the
        x = conv_bn_act(x, self.conv3, self.bn3, relu=True, gate_dx=True)      # previous one's output (conv.Gate)
        x = conv_bn_act(x, self.conv4, self.bn4, relu=True, gate_dx=True)
        # deconv(k2,s2) = 1x1 conv to (a, b, c) channels; ReLU, conv5 (1x1) and sigmoid are per pixel, so
        # they run before the pixel shuffle and only 81 channels are ever moved
        u = self.deconv.forward_unshuffled(x, relu=True, gate_dx=True)   # [N, 2, 2, 256, H, W]; conv4's only reader
        n, h, w = u.shape[0], u.shape[4], u.shape[5]
        K = self.conv5.weight.shape[0]
        if select_class is not None and not activate and not shuffled and u.is_cuda and _conv.GATES:
            cls4 = select_class.reshape(-1, 1).expand(-1, 4).reshape(-1)             # rows of u are (RoI, a, b)
            sel = conv1x1_class_rows(u.view(n * 4, u.shape[3], h, w), self.conv5.weight, self.conv5.bias, cls4,
                                     gate_dx=getattr(u, "_fi_gate", None))           # u's only reader
            return sel.view(n, 2, 2, h, w)
        if activate or K % 16 == 0 or not u.is_cuda:
            y = self.conv5(u.view(n * 4, u.shape[3], h, w))
        else:
            # training: conv5 with its filter bank zero-padded to a multiple of 16 output channels (81 -> 96).  The
            # forward tile covers 128 output channels either way; the data gradient's reduction then runs over
            # 96 = 6 x 16 channels on the tap-major kernel instead of 81 on the scalar-gather path (1.3 -> 0.5 ms),
            # and the weight gradient gets whole rows.  The padded logits are zero, and nothing reads them: the loss
            # gathers the target class (< K) of every RoI.
            pad = (-K) % 16
            w5 = torch.cat((self.conv5.weight, self.conv5.weight.new_zeros(pad, *self.conv5.weight.shape[1:])), 0)
            b5 = torch.cat((self.conv5.bias, self.conv5.bias.new_zeros(pad))) if self.conv5.bias is not None else None
            y = conv2d(u.view(n * 4, u.shape[3], h, w), w5, b5, gate_dx=getattr(u, "_fi_gate", None))    # u's only reader
        if activate:
            y = self.sigmoid(y)
        y = y.view(n, 2, 2, y.shape[1], h, w)
        if not shuffled:
            return y
        return y.permute(0, 3, 4, 1, 5, 2).reshape(n, y.shape[3], 2 * h, 2 * w)
