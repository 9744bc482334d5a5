"""One persistent fp32 buffer that holds every parameter gradient of a model.

Why.  A training step of the detector produces ~500 gradient tensors (386 MB for R101-FPN + the OT
module).  Laid out in ONE buffer, in the order autograd produces them (reverse registration order),
  * the weight-gradient / fused-BN-backward kernels accumulate straight into their slots, so the ~300
    per-layer memsets of a backward pass are ONE fill per step (conv.prepare_step);
  * a data-parallel gradient bucket is a CONTIGUOUS slice of the buffer: RCCL all-reduces it in place --
    no packing copy into a flat bucket and no copy back (data_parallel.GradientBuckets; the reference's
    nn.DataParallel, tools/utils.py:645-654, reduces per-parameter tensors onto GPU 0);
  * gradient addresses are the same every step, so the descriptor table of the fused clip + SGD kernel
    (optim.clip_and_step) is built once.

Layout.  Units are placed in reverse registration order.  A unit is one parameter, or -- for a
convolution directly followed by its BatchNorm2d in the owning module -- the triple
(bn.bias, bn.weight, conv.bias) of C floats each, which is exactly the (d shift, d gamma, d conv-bias)
block fi_bn_act_backward writes.  Units start on 16-byte boundaries.  Consecutive units form a bucket
once they reach `bucket_bytes`; each bucket ends with one float per parameter (the "had a gradient"
flags that ride through the all-reduce for the cross-rank consistency check).
"""
import os as _os
import weakref

import torch
import torch.nn as nn

DEFAULT_BUCKET_BYTES = int(float(_os.environ.get("FI_DP_BUCKET_MB", "64")) * 1024 * 1024)


def _align4(n):
    return (n + 3) // 4 * 4


class Bucket(object):
    """[start, end) of the arena (floats).  `params`: the parameters whose slots END in this bucket -- their "had a
    gradient" flags ride at flag_off.  `wait`: the parameters whose gradient must have arrived before the slice may
    be reduced -- `params` plus, for a bucket that is a piece of ONE large parameter (split_large), that parameter."""
    __slots__ = ("params", "start", "end", "flag_off", "wait")

    def __init__(self, params, start, end, flag_off, wait=None):
        self.params, self.start, self.end, self.flag_off = params, start, end, flag_off
        self.wait = list(params) if wait is None else wait


class ArenaLayout(object):
    """Offsets (in floats) of every trainable parameter's gradient slot; see the module docstring."""

    def __init__(self, module, bucket_bytes=DEFAULT_BUCKET_BYTES):
        self.bucket_bytes = int(bucket_bytes)
        params = [p for p in module.parameters() if p.requires_grad]
        trainable = set(params)
        # (conv, bn) pairs by adjacency among a module's children: Bottleneck's conv1/bn1..., Sequential(conv, bn, ...)
        group_of = {}
        for m in module.modules():
            kids = list(m._modules.values())
            for a, b in zip(kids, kids[1:]):
                if isinstance(a, nn.Conv2d) and isinstance(b, nn.BatchNorm2d) and b.affine and \
                        b.num_features == a.out_channels and b.num_features % 4 == 0 and b not in group_of:
                    members = [b.bias, b.weight, a.bias]
                    members = [p if (p is not None and p in trainable) else None for p in members]
                    if any(p is not None for p in members) and not any(p in group_of for p in members if p is not None):
                        g = (b, members)
                        group_of[b] = g
                        for p in members:
                            if p is not None:
                                group_of[p] = g
        self.slot = {}              # parameter -> (offset, numel)
        self.bn_slot = {}           # BatchNorm2d module -> offset of its 3*C block
        self.bn_partner = {}        # BatchNorm2d module -> the convolution bias whose gradient is the block's third part
        self.buckets = []
        off, start, cur, done = 0, 0, [], set()

        def close():
            nonlocal off, start, cur
            flag_off = off
            off = _align4(off + len(cur))
            # A bucket boundary may fall INSIDE a parameter (the slice is contiguous either way): a parameter larger
            # than ~1.5 buckets (the heads' two full-window convolutions: 103 MB and 51 MB of gradient) is cut into
            # bucket-sized pieces, each its own collective, so that the exchange of the first piece starts while
            # the later ones are still queued and no single 100 MB all-reduce sits at the end of backward.  The
            # pieces wait for the same gradient; the flags ride in the last piece.
            piece = self.bucket_bytes // 4
            last = cur[-1] if cur else None
            lo, n = self.slot.get(last, (0, 0)) if last is not None else (0, 0)
            if n > piece + piece // 2 and lo + _align4(n) == flag_off:      # the oversize is the LAST unit, a plain parameter
                cut = start
                while flag_off - cut > piece + piece // 2:
                    nxt = _align4(max(cut + piece, lo + 4))
                    if nxt >= flag_off:
                        break
                    self.buckets.append(Bucket([], cut, nxt, nxt, wait=list(cur)))      # no flags of its own
                    cut = nxt
                self.buckets.append(Bucket(cur, cut, off, flag_off))
                start, cur = off, []
                return
            self.buckets.append(Bucket(cur, start, off, flag_off))
            start, cur = off, []

        for p in reversed(params):
            if p in done:
                continue
            g = group_of.get(p)
            if g is not None:
                bn, members = g
                C = bn.num_features
                self.bn_slot[bn] = off
                self.bn_partner[bn] = members[2]
                for i, q in enumerate(members):
                    if q is not None:
                        self.slot[q] = (off + i * C, C)
                        done.add(q)
                        cur.append(q)
                off = _align4(off + 3 * C)
            else:
                self.slot[p] = (off, p.numel())
                done.add(p)
                cur.append(p)
                off = _align4(off + p.numel())
            if (off - start) * 4 >= self.bucket_bytes:
                close()
        if cur:
            close()
        self.total = off
        self.params = [p for b in self.buckets for p in b.params]
        self._buf = {}              # device -> tensor
        self.fresh = False          # zeroed and not yet consumed by a gradient exchange
        self.superseded = False     # get_layout() built a newer layout for the same module: holders must re-fetch

    def buffer(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        b = self._buf.get(device)
        if b is None:
            self._buf.clear()       # a model lives on one device: drop the buffer of the device it left
            b = self._buf[device] = torch.zeros(max(self.total, 1), device=device, dtype=torch.float32)
            self.fresh = True
        return b

    def zero(self, device):
        b = self.buffer(device)
        b.zero_()
        self.fresh = True
        return b

    def view(self, p, buf):
        """The slot of `p` as a tensor with p's shape AND p's memory layout (channels-last parameters get a
        channels-last gradient, which is what AccumulateGrad and optim.clip_and_step expect)."""
        off, n = self.slot[p]
        flat = buf[off:off + n]
        if p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last):
            co, ci, r, s = p.shape
            return flat.view(co, r, s, ci).permute(0, 3, 1, 2)
        if p.is_contiguous():
            return flat.view(p.shape)
        return None                 # exotic strides: the caller falls back to a plain gradient tensor

    def holds(self, p, g, buf):
        off, _ = self.slot[p]
        return g.data_ptr() == buf.data_ptr() + 4 * off and g.dtype == torch.float32 and g.stride() == p.stride()

    def live_gradients(self, buf):
        """True if some parameter's .grad currently lives in `buf` (gradient accumulation over several
        backward passes without zero_grad): the buffer must then not be cleared."""
        lo, hi = buf.data_ptr(), buf.data_ptr() + 4 * buf.numel()
        for p in self.params:
            g = p.grad
            if g is not None and lo <= g.data_ptr() < hi:
                return True
        return False


_LAYOUTS = weakref.WeakKeyDictionary()       # module -> ArenaLayout


def get_layout(module, bucket_bytes=None):
    """The layout of `module` (created on first use).  A different explicit `bucket_bytes`, or a change of
    the set of trainable parameters, rebuilds it."""
    lay = _LAYOUTS.get(module)
    if lay is not None:
        same_params = len(lay.params) == sum(1 for p in module.parameters() if p.requires_grad) and \
            all(p in lay.slot for p in module.parameters() if p.requires_grad)
        if same_params and (bucket_bytes is None or int(bucket_bytes) == lay.bucket_bytes):
            return lay
    if lay is not None:
        lay.superseded = True       # conv._prepare_step / GradientBuckets compare and rebuild (or refuse)
    lay = ArenaLayout(module, DEFAULT_BUCKET_BYTES if bucket_bytes is None else bucket_bytes)
    _LAYOUTS[module] = lay
    return lay
