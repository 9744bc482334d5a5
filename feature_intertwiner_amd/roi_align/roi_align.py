"""RoIAlign on pixel boxes -- call-compatible with lib/roi_align/roi_align.py:6-48 of the
reference: `RoIAlign(crop_height, crop_width, extrapolation_value=0, transform_fpcoor=True)
(featuremap, boxes, box_ind)` with boxes [M, 4] = (x1, y1, x2, y2) in feature-map pixels.

The module only converts the boxes to the normalised (y1, x1, y2, x2) form of
crop_and_resize and launches the HIP operator.  With `transform_fpcoor` the sampling grid is
moved to bin centres (tensorpack's convention): for a crop of `c` bins over [a, b] the first
sample sits at a + (b - a)/(2c) - 0.5 and the last one (c - 1) bins further.  The order of the
floating-point operations below is part of the specification (oracle.roi_align_boxes pins it, and
tests/golden/wrappers.npz holds what the reference's own RoIAlign.forward computed).  Every division is a true
fp32 division: the divisors are device tensors, because `tensor / python_number` on the GPU is a multiplication
by the rounded reciprocal -- 1 ulp away from the reference's CPU result for divisors like 7 or W - 1
(found by tests/test_reference_wrappers.py on the MI355X).
"""
import torch
from torch import nn

from .. import _lib
from .crop_and_resize import CropAndResizeFunction


def to_crop_boxes(boxes, map_h, map_w, crop_h, crop_w, bin_centres=True):
    """[M, 4] pixel (x1, y1, x2, y2) -> [M, 4] normalised (y1, x1, y2, x2) for crop_and_resize."""
    left, top, right, bottom = boxes.unbind(dim=1)
    k = _lib.const_tensor([float(map_w - 1), float(map_h - 1), float(crop_w), float(crop_h)], boxes.device, boxes.dtype)
    span_x, span_y, n_x, n_y = k[0], k[1], k[2], k[3]
    if not bin_centres:
        return torch.stack((top / span_y, left / span_x, bottom / span_y, right / span_x), dim=1)
    bin_w = (right - left) / n_x
    bin_h = (bottom - top) / n_y
    first_x = (left + bin_w / 2 - 0.5) / span_x           # (/ 2 is exact either way)
    first_y = (top + bin_h / 2 - 0.5) / span_y
    reach_x = bin_w * float(crop_w - 1) / span_x
    reach_y = bin_h * float(crop_h - 1) / span_y
    return torch.stack((first_y, first_x, first_y + reach_y, first_x + reach_x), dim=1)


class RoIAlign(nn.Module):
    def __init__(self, crop_height, crop_width, extrapolation_value=0, transform_fpcoor=True):
        super().__init__()
        self.crop_height, self.crop_width = crop_height, crop_width
        self.extrapolation_value = extrapolation_value
        self.transform_fpcoor = transform_fpcoor

    def forward(self, featuremap, boxes, box_ind):
        """featuremap [N, C, H, W]; boxes [M, 4]; box_ind [M] -> [M, C, crop_height, crop_width]."""
        norm = to_crop_boxes(boxes, featuremap.size(2), featuremap.size(3), self.crop_height, self.crop_width,
                             bin_centres=self.transform_fpcoor)
        op = CropAndResizeFunction(self.crop_height, self.crop_width, self.extrapolation_value)
        return op(featuremap, norm.detach().contiguous(), box_ind.detach())
