"""Gradient clipping + SGD step of the training iteration in three launches (csrc/sgd.hip).

`clip_and_step(optimizer, max_norm)` does what lib/workflow.py:226-230 does with
`torch.nn.utils.clip_grad_norm_(params, max_norm)` + `optimizer.step()` for a `torch.optim.SGD` built by
`workflow.set_optimizer` (tools/utils.py:474-501: momentum, weight decay on the non-BatchNorm group).  The
optimizer object stays the owner of the hyper-parameters and of the momentum buffers
(`optimizer.state[p]['momentum_buffer']`), so its state dict -- and the reference's checkpoint file -- is
unchanged; only the arithmetic moves into one pass over the parameters.
"""
import weakref

import numpy as np
import torch

from . import _lib

_DESC = np.dtype([("param", "<u8"), ("grad", "<u8"), ("buf", "<u8"), ("numel", "<i8"), ("chunk_base", "<i8"),
                  ("weight_decay", "<f4"), ("lr", "<f4"), ("momentum", "<f4"), ("pad", "<f4")])
_CACHE = weakref.WeakKeyDictionary()      # optimizer -> {"key", "table", "partial", "out", "chunks", "n"}


def _dense_same_layout(p, g):
    if g.dtype != torch.float32 or p.dtype != torch.float32 or g.shape != p.shape or g.stride() != p.stride():
        return False
    return p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))


def supported(optimizer):
    """Plain SGD (no dampening / nesterov / maximize) on fp32 CUDA parameters."""
    if type(optimizer) is not torch.optim.SGD:
        return False
    for grp in optimizer.param_groups:
        if grp.get("dampening", 0) != 0 or grp.get("nesterov", False) or grp.get("maximize", False):
            return False
        for p in grp["params"]:
            if not p.is_cuda or p.dtype != torch.float32:
                return False
    return True


def clip_and_step(optimizer, max_norm):
    """Returns the total gradient norm (0-d device tensor), like clip_grad_norm_.  max_norm None/<=0: no clip."""
    L = _lib.load()
    entries = []
    for grp in optimizer.param_groups:
        mom, wd, lr = float(grp["momentum"]), float(grp["weight_decay"]), float(grp["lr"])
        for p in grp["params"]:
            g = p.grad
            if g is None:
                continue
            if not _dense_same_layout(p, g):
                raise _lib.FiError("clip_and_step: gradient of a %s parameter does not share its memory layout "
                                   "(strides %s vs %s)" % (tuple(p.shape), g.stride(), p.stride()))
            buf = None
            if mom != 0.0:
                st = optimizer.state[p]
                buf = st.get("momentum_buffer")
                if buf is None:     # torch's first step sets buf = g; momentum * 0 + g is the same value
                    buf = st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if buf.stride() != p.stride():
                    raise _lib.FiError("clip_and_step: momentum buffer layout differs from its parameter")
            entries.append((p, g, buf, wd, lr, mom))
    if not entries:
        return None
    dev = entries[0][0].device
    key = tuple((p.data_ptr(), g.data_ptr(), 0 if b is None else b.data_ptr(), p.numel(), wd, lr, mom)
                for p, g, b, wd, lr, mom in entries)
    c = _CACHE.get(optimizer)
    if c is None or c["key"] != key:
        desc = np.zeros(len(entries), dtype=_DESC)
        base = 0
        for i, (p, g, b, wd, lr, mom) in enumerate(entries):
            desc[i] = (p.data_ptr(), g.data_ptr(), 0 if b is None else b.data_ptr(), p.numel(), base, wd, lr, mom, 0.0)
            base += int(L.fi_sgd_chunks(p.numel()))
        host = torch.from_numpy(desc.view(np.uint8).copy()).pin_memory()
        table = host.to(dev, non_blocking=True)
        c = _CACHE[optimizer] = {"key": key, "table": table, "host": host, "chunks": base, "n": len(entries),
                                     "partial": torch.empty(max(base, 1), device=dev, dtype=torch.float32),
                                     "out": torch.empty(2, device=dev, dtype=torch.float32)}
    with torch.cuda.device(dev):
        _lib.check(L.fi_sgd_clip_step(_lib.ptr(c["table"]), c["n"], c["chunks"],
                                      float(max_norm) if max_norm else 0.0, _lib.ptr(c["partial"]), _lib.ptr(c["out"]),
                                      _lib.current_stream()), "fi_sgd_clip_step")
    # the kernel wrote parameters, buffers and gradients behind autograd's back: bump their version counters
    # (the per-step caches of W^T and of the eval-BN folds are keyed on parameter versions)
    torch.autograd.graph.increment_version([p for p, _, _, _, _, _ in entries])
    torch.autograd.graph.increment_version([b for _, _, b, _, _, _ in entries if b is not None])
    torch.autograd.graph.increment_version([g for _, g, _, _, _, _ in entries])
    return c["out"][0]
