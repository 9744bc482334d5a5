"""Drop-in for lib/roi_pooling/functions/roi_pool.py:6-38 of the reference."""
import torch

from ... import _lib


class _RoIPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, rois, pooled_height, pooled_width, spatial_scale):
        _lib.require_cuda(features, rois)
        L = _lib.load()
        features = features.contiguous().float()
        rois_c = rois.detach().contiguous().float()
        if rois_c.dim() != 2 or rois_c.shape[1] != 5:
            raise _lib.FiError("rois must be [num_rois, 5] = (batch, x1, y1, x2, y2)")
        B, C, H, W = features.shape
        N = rois_c.shape[0]
        out = torch.empty((N, C, pooled_height, pooled_width), device=features.device,
                          dtype=torch.float32)
        argmax = torch.empty((N, C, pooled_height, pooled_width), device=features.device,
                             dtype=torch.int32)
        with torch.cuda.device(features.device):
            _lib.check(L.fi_roi_pool_forward(
                _lib.ptr(features), _lib.ptr(rois_c), N, B, C, H, W, int(pooled_height),
                int(pooled_width), float(spatial_scale), _lib.ptr(out), _lib.ptr(argmax),
                _lib.current_stream()), "fi_roi_pool_forward")
        ctx.feature_size = (B, C, H, W)
        ctx.pool = (int(pooled_height), int(pooled_width), float(spatial_scale))
        ctx.save_for_backward(rois_c, argmax)
        ctx.mark_non_differentiable(argmax)
        return out, argmax

    @staticmethod
    def backward(ctx, grad_output, _grad_argmax):
        rois_c, argmax = ctx.saved_tensors
        L = _lib.load()
        g = grad_output.contiguous().float()
        B, C, H, W = ctx.feature_size
        ph, pw, scale = ctx.pool
        grad_input = torch.empty((B, C, H, W), device=g.device, dtype=torch.float32)
        with torch.cuda.device(g.device):
            _lib.check(L.fi_roi_pool_backward(
                _lib.ptr(g), _lib.ptr(rois_c), _lib.ptr(argmax), rois_c.shape[0], B, C, H, W, ph,
                pw, scale, _lib.ptr(grad_input), _lib.current_stream()), "fi_roi_pool_backward")
        return grad_input, None, None, None, None


class RoIPoolFunction(object):
    """`RoIPoolFunction(ph, pw, spatial_scale)(features, rois)`; rois [N,5] =
    (batch index, x1, y1, x2, y2) in pixels.  After a call, `.argmax` holds the int32
    flat-index tensor as in the reference (ctx.argmax, roi_pool.py:18)."""

    def __init__(self, pooled_height, pooled_width, spatial_scale):
        self.pooled_width = pooled_width
        self.pooled_height = pooled_height
        self.spatial_scale = spatial_scale
        self.feature_size = None
        self.argmax = None

    def __call__(self, features, rois):
        self.feature_size = features.size()
        out, self.argmax = _RoIPool.apply(features, rois, self.pooled_height, self.pooled_width,
                                          self.spatial_scale)
        return out

    forward = __call__
