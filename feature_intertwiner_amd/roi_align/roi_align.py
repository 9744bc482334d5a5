"""RoIAlign on pixel boxes -- call-compatible with lib/roi_align/roi_align.py:6-48 of the
reference: `RoIAlign(crop_height, crop_width, extrapolation_value=0, transform_fpcoor=True)
(featuremap, boxes, box_ind)` with boxes [M, 4] = (x1, y1, x2, y2) in feature-map pixels.

The module only converts the boxes to the normalised (y1, x1, y2, x2) form of
crop_and_resize and launches the HIP operator.  With `transform_fpcoor` the sampling grid is
moved to bin centres (tensorpack's convention): for a crop of `c` bins over [a, b] the first
sample sits at a + (b - a)/(2c) - 0.5 and the last one (c - 1) bins further.  The order of the
floating-point operations below is part of the specification (oracle.roi_align_boxes pins it).
"""
import torch
from torch import nn

from .crop_and_resize import CropAndResizeFunction


def to_crop_boxes(boxes, map_h, map_w, crop_h, crop_w, bin_centres=True):
    """[M, 4] pixel (x1, y1, x2, y2) -> [M, 4] normalised (y1, x1, y2, x2) for crop_and_resize."""
    left, top, right, bottom = boxes.unbind(dim=1)
    span_x, span_y = float(map_w - 1), float(map_h - 1)
    if not bin_centres:
        return torch.stack((top / span_y, left / span_x, bottom / span_y, right / span_x), dim=1)
    bin_w = (right - left) / float(crop_w)
    bin_h = (bottom - top) / float(crop_h)
    first_x = (left + bin_w / 2 - 0.5) / span_x
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
