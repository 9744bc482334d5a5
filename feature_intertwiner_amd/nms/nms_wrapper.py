"""Drop-in for lib/nms/nms_wrapper.py:14-34 of the reference."""
import numpy as np
import torch

from .pth_nms import nms_sorted


def nms(dets, thresh, strict=False):
    """dets [bs, N, 5] = (y1, x1, y2, x2, score) -> np.int32 [bs, min_keep_num]: per-image
    greedy NMS, every image truncated to the shortest keep list (reference :29-33).
    All images are processed by one launch pair; one host copy at the end (the
    reference API returns a NumPy array)."""
    bs = dets.size(0)
    order = torch.sort(dets[:, :, 4], dim=1, descending=True, stable=True)[1]
    sorted_dets = torch.gather(dets, 1, order.unsqueeze(2).expand(-1, -1, dets.size(2))).contiguous()
    keep, num_out = nms_sorted(sorted_dets, thresh, 0, strict)
    keep = torch.gather(order, 1, keep)  # back to the caller's row numbering
    counts = num_out.cpu().numpy()
    min_keep = int(counts.min()) if bs > 0 else 0
    return keep[:, :min_keep].to(torch.int32).cpu().numpy().astype(np.int32)
