"""Gradient clipping + SGD step of the training iteration in three launches (csrc/sgd.hip).

`clip_and_step(optimizer, max_norm)` does what lib/workflow.py:226-230 does with
`torch.nn.utils.clip_grad_norm_(params, max_norm)` + `optimizer.step()` for a `torch.optim.SGD` built by
`workflow.set_optimizer` (tools/utils.py:474-501: momentum, weight decay on the non-BatchNorm group).  The
optimizer object stays the owner of the hyper-parameters and of the momentum buffers
(`optimizer.state[p]['momentum_buffer']`), so its state dict -- and the reference's checkpoint file -- is
unchanged; only the arithmetic moves into one pass over the parameters.
"""
import operator
import weakref

import numpy as np
import torch

from . import _lib

_DESC = np.dtype([("param", "<u8"), ("grad", "<u8"), ("buf", "<u8"), ("numel", "<i8"), ("chunk_base", "<i8"),
                  ("weight_decay", "<f4"), ("lr", "<f4"), ("momentum", "<f4"), ("pad", "<f4")])
_CACHE = weakref.WeakKeyDictionary()      # optimizer -> the descriptor table and what it was built from


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


def _launch(L, c, max_norm):
    fn = L.fi_sgd_clip_step_guarded if c.get("guard") else L.fi_sgd_clip_step
    with torch.cuda.device(c["dev"]):
        _lib.check(fn(_lib.ptr(c["table"]), c["n"], c["chunks"],
                      float(max_norm) if max_norm else 0.0, _lib.ptr(c["partial"]), _lib.ptr(c["out"]),
                      _lib.current_stream()), "fi_sgd_clip_step")
    # the kernel wrote parameters, buffers and gradients behind autograd's back: bump their version counters
    # (the per-step caches of W^T and of the eval-BN folds are keyed on parameter versions)
    torch.autograd.graph.increment_version(c["params"])
    if c["bufs"]:
        torch.autograd.graph.increment_version(c["bufs"])
    return c["out"][0]


def _hyper(optimizer):
    return tuple((float(g["momentum"]), float(g["weight_decay"]), float(g["lr"]), len(g["params"]))
                 for g in optimizer.param_groups)


def skipped_steps(optimizer):
    """Steps `clip_and_step(..., skip_nonfinite=True)` skipped so far because the gradient norm was inf / NaN
    (host-synchronising; for logs and tests)."""
    c = _CACHE.get(optimizer)
    return 0 if c is None else int(c["out"][3].item())


def clip_and_step(optimizer, max_norm, skip_nonfinite=False):
    """Returns the total gradient norm (0-d device tensor), like clip_grad_norm_.  max_norm None/<=0: no clip.
    skip_nonfinite (the 16-bit paths, workflow.train_step under a loss scale): a step whose gradient norm is inf / NaN
    leaves parameters and momentum buffers untouched (decided on the device, no host synchronisation);
    `skipped_steps(optimizer)` counts them.

    The descriptor table is static across steps when the gradients live where they lived last step (the gradient
    arena, grad_arena.py): the per-step host work is then one pass collecting (parameter, gradient) addresses and
    contiguity flags and comparing them with the cached ones (~0.2 ms for ~430 parameters instead of ~2 ms of
    validation + table building, at a point of the step where the device has nothing queued)."""
    L = _lib.load()
    c = _CACHE.get(optimizer)
    if c is not None:
        c["guard"] = bool(skip_nonfinite)
    if c is not None and c["state"] is optimizer.state and c["hyper"] == _hyper(optimizer):
        grads = [p.grad for p in c["all"]]
        sig = [0 if g is None else g.data_ptr() for g in grads]
        st = optimizer.state
        if ([p.data_ptr() for p in c["params"]] == c["psig"] and [g is None for g in grads] == c["absent"]
                and (not c["bufs"] or all(map(operator.is_, c["bufs"], [st[p].get("momentum_buffer") for p in c["mom"]])))):
            live = [g for g in grads if g is not None]
            if sig != c["gsig"]:
                # some gradients were allocated elsewhere this step (the ones autograd itself materialises):
                # validate those, patch their addresses into the next staging copy of the table and upload it
                lsig = [x for x in sig if x]
                moved = [i for i, (a, b) in enumerate(zip(lsig, c["lsig"])) if a != b]
                for i in moved:
                    if not _dense_same_layout(c["params"][i], live[i]):
                        raise _lib.FiError("clip_and_step: gradient of a %s parameter does not share its memory layout"
                                           % (tuple(c["params"][i].shape),))
                    c["contig"][i] = live[i].is_contiguous()
                if torch.cuda.is_current_stream_capturing():
                    # a step being captured into a hipGraph: its upload becomes a copy node that every replay repeats, so
                    # the staging buffer must stay what it is now -- a buffer of its own, kept for the life of the cache
                    # entry (and no event query: not permitted on events of a capturing stream)
                    if not c["spare"]:
                        raise _lib.FiError("clip_and_step: more graph captures of one optimiser than spare staging buffers")
                    host = c["spare"].pop()          # (pinned memory cannot be allocated while a stream is capturing)
                    desc = host.numpy().view(c["desc"].dtype)
                    c.setdefault("captured", []).append(host)
                    ev = None
                else:
                    host, desc, ev = c["ring"][c["turn"] % len(c["ring"])]
                    c["turn"] += 1
                    ev.synchronize()             # its previous upload (len(ring) steps ago) has long completed
                desc[:] = c["desc"]
                desc["grad"] = lsig
                c["desc"] = desc
                with torch.cuda.device(c["dev"]):
                    c["table"].copy_(host, non_blocking=True)
                    if ev is not None:
                        ev.record()
                c["gsig"], c["lsig"] = sig, lsig
            if [g.is_contiguous() for g in live] == c["contig"]:
                out = _launch(L, c, max_norm)
                torch.autograd.graph.increment_version(live)
                return out
    every, entries = [], []
    for grp in optimizer.param_groups:
        mom, wd, lr = float(grp["momentum"]), float(grp["weight_decay"]), float(grp["lr"])
        for p in grp["params"]:
            every.append(p)
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
    desc = np.zeros(len(entries), dtype=_DESC)
    base = 0
    for i, (p, g, b, wd, lr, mom) in enumerate(entries):
        desc[i] = (p.data_ptr(), g.data_ptr(), 0 if b is None else b.data_ptr(), p.numel(), base, wd, lr, mom, 0.0)
        base += int(L.fi_sgd_chunks(p.numel()))
    old = c
    ring = []
    for _ in range(3):
        host = torch.empty(desc.nbytes, dtype=torch.uint8).pin_memory()
        ring.append((host, host.numpy().view(_DESC), torch.cuda.Event()))
    ring[0][1][:] = desc
    table = torch.empty(desc.nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        table.copy_(ring[0][0], non_blocking=True)
        ring[0][2].record()
    c = _CACHE[optimizer] = {
        "guard": bool(skip_nonfinite),
        "spare": [torch.empty(desc.nbytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
        if not torch.cuda.is_current_stream_capturing() else [],
        "dev": dev, "table": table, "ring": ring, "turn": 1, "desc": ring[0][1], "chunks": base, "n": len(entries),
        "state": optimizer.state, "hyper": _hyper(optimizer), "all": every, "absent": [p.grad is None for p in every],
        "params": [e[0] for e in entries], "bufs": [e[2] for e in entries if e[2] is not None],
        "mom": [e[0] for e in entries if e[2] is not None],
        "gsig": [0 if p.grad is None else p.grad.data_ptr() for p in every], "lsig": [e[1].data_ptr() for e in entries],
        "psig": [e[0].data_ptr() for e in entries], "contig": [e[1].is_contiguous() for e in entries]}
    if old is not None and old["dev"] == dev and old["chunks"] >= base:
        c["partial"], c["out"] = old["partial"], old["out"]
    else:
        c["partial"] = torch.empty(max(base, 1), device=dev, dtype=torch.float32)
        c["out"] = torch.zeros(4, device=dev, dtype=torch.float32)     # norm, clip factor, skipped now, skipped so far
    out = _launch(L, c, max_norm)
    torch.autograd.graph.increment_version([e[1] for e in entries])
    return out
