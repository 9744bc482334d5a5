"""`_RoIPooling(pooled_height, pooled_width, spatial_scale)(features, rois)` -- the nn.Module
face of RoIPoolFunction, call-compatible with lib/roi_pooling/modules/roi_pool.py:5-14 of the
reference.  rois are [K, 5] = (batch index, x1, y1, x2, y2) in image pixels; `spatial_scale`
maps them onto the feature map (1/4 .. 1/32 for the FPN levels)."""
from torch import nn

from ..functions.roi_pool import RoIPoolFunction


class _RoIPooling(nn.Module):
    def __init__(self, pooled_height, pooled_width, spatial_scale):
        super().__init__()
        self.pooled_height, self.pooled_width = int(pooled_height), int(pooled_width)
        self.spatial_scale = float(spatial_scale)
        self.last_argmax = None          # flat NCHW indices of the maxima of the most recent call

    def extra_repr(self):
        return "%dx%d bins, scale %g" % (self.pooled_height, self.pooled_width, self.spatial_scale)

    def forward(self, features, rois):
        op = RoIPoolFunction(self.pooled_height, self.pooled_width, self.spatial_scale)
        pooled = op(features, rois)
        self.last_argmax = getattr(op, "argmax", None)
        return pooled
