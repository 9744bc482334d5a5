"""Drop-in for lib/nms/pth_nms.py:5-46 of the reference, plus the static-shape
entry point the training step uses (no host synchronisation)."""
import torch

from .. import _lib


def nms_sorted(boxes, thresh, max_keep=0, strict=False):
    """Greedy NMS on rows already sorted by descending score.

    boxes [bs, N, >=4] (or [N, >=4]) fp32 on the GPU.  Returns (keep [bs, N] int64 --
    indices into the sorted rows, valid up to num_out -- and num_out [bs] int32), both
    on the GPU; nothing is copied to the host.
    """
    _lib.require_cuda(boxes)
    L = _lib.load()
    squeeze = boxes.dim() == 2
    b = (boxes.unsqueeze(0) if squeeze else boxes).contiguous().float()
    bs, N, stride = b.shape
    keep = torch.zeros((bs, N), device=b.device, dtype=torch.int64)
    num_out = torch.zeros((bs,), device=b.device, dtype=torch.int32)
    ws_bytes = L.fi_nms_workspace_bytes(bs, N)
    ws = torch.empty((max(ws_bytes, 8) // 8,), device=b.device, dtype=torch.int64)
    with torch.cuda.device(b.device):
        _lib.check(L.fi_nms_sorted(_lib.ptr(b), bs, N, stride, float(thresh), 1 if strict else 0,
                                   int(max_keep), _lib.ptr(keep), _lib.ptr(num_out), _lib.ptr(ws),
                                   _lib.current_stream()), "fi_nms_sorted")
    if _lib.TAP is not None:
        _lib.TAP("nms_sorted", boxes=b, thresh=float(thresh), strict=bool(strict), max_keep=int(max_keep), keep=keep,
                 num_out=num_out)
    if squeeze:
        return keep[0], num_out[0]
    return keep, num_out


def pth_nms(dets, thresh, strict=False):
    """dets [N,5] = (y1, x1, y2, x2, score) -> LongTensor of kept row indices, in
    descending-score order.  Comparison is the reference CPU path's `>=`
    (lib/nms/src/nms.c:59); strict=True selects the CUDA kernel's `>`
    (lib/nms/src/cuda/nms_kernel.cu:63).  Returning a variable-length tensor costs one
    device->host read of the count, exactly like the reference API."""
    _lib.require_cuda(dets)
    scores = dets[:, 4]
    order = torch.sort(scores, dim=0, descending=True, stable=True)[1]
    sorted_dets = dets[order].contiguous()
    keep, num_out = nms_sorted(sorted_dets, thresh, 0, strict)
    n = int(num_out.item())
    return order[keep[:n]].contiguous()
