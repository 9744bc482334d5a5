"""Data-parallel training, one process per GPU over RCCL (torch.distributed backend "nccl"
is RCCL on ROCm).  Replaces the reference's single-process nn.DataParallel
(tools/utils.py:645-654), which re-broadcasts ~386 MB of parameters to every GPU each
step, gathers all outputs to GPU 0 and runs the optimizer there.

Here every rank owns a full replica and its shard of the minibatch (images are
independent through the whole detector), so the only exchanges per step are
  1. an all-reduce (mean) of the gradients, bucketed (~25 MB flat buffers filled in
     reverse parameter order as autograd produces the gradients) and issued from a side
     HIP stream so that RCCL traffic over xGMI overlaps the rest of backward;
  2. one small all-reduce (sum) of the intertwiner class statistics (feat*cnt, cnt),
     ~1 MB -- algebraically the reference's gather-to-GPU-0 + _merge_feat_vec
     (lib/model.py:217-224).
There is no data-path collective besides these.  The same code runs on CPU tensors with
the gloo backend (tests/test_data_parallel_gloo.py).
"""
import torch
import torch.distributed as dist


class _AllReduceSumIdentityGrad(torch.autograd.Function):
    """y = sum over ranks of x; backward returns world_size * g.

    Every rank evaluates the same function M of the reduced statistics and back-propagates
    only through its own contribution s_g.  Gradients are AVERAGED over ranks afterwards, so
    the path back into the local statistics carries the factor world_size:
        (1/W) sum_g W * dM/ds * ds_g/dtheta = dM/dtheta   (the reference's single meta loss,
    lib/workflow.py:180, 221), while parameters M owns itself (ot_loss.G_net / critic) receive
    the same gradient on every rank and average to 1x.  (Scaling the loss term by W instead
    would multiply the gradient of those parameters by W.)"""

    @staticmethod
    def forward(ctx, x, group):
        ctx.world = dist.get_world_size(group)
        y = x.detach().clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
        return y

    @staticmethod
    def backward(ctx, g):
        return g * float(ctx.world), None


def _force_collectives():
    """FI_DP_FORCE=1: run every collective even in a 1-rank group (tests/test_gpu_rccl_single_rank.py drives
    the RCCL code path -- process-group init, side stream, async work handles -- on a one-GPU box)."""
    import os
    return os.environ.get("FI_DP_FORCE") == "1"


def all_reduce_statistics(feat_sum, cnt_sum, group=None):
    """reduce_fn for MaskRCNN.meta_loss: one collective for both tensors."""
    if not (dist.is_available() and dist.is_initialized()) or \
            (dist.get_world_size(group) == 1 and not _force_collectives()):
        return feat_sum, cnt_sum
    flat = torch.cat([feat_sum.reshape(-1), cnt_sum.reshape(-1)])
    flat = _AllReduceSumIdentityGrad.apply(flat, group)
    n = feat_sum.numel()
    return flat[:n].view_as(feat_sum), flat[n:].view_as(cnt_sum)


def _mem_flat(t):
    """1-D view of a dense tensor in MEMORY order (no copy): gradients of channels-last parameters are
    channels-last themselves, and flattening them in logical order would cost a strided copy per tensor
    on the way into and out of every bucket."""
    if t.is_contiguous():
        return t.view(-1)
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return t.permute(0, 2, 3, 1).reshape(-1)          # a view for channels-last strides
    return t.reshape(-1)


class GradientBuckets(object):
    """Bucketed, overlapped gradient all-reduce.

        sync = GradientBuckets(model)
        ...
        sync.begin(key)      # optional: names the graph variant of this step (e.g. do_meta on/off)
        loss.backward()      # buckets launch from autograd hooks as they fill
        sync()               # waits, writes the rank-mean gradients back into p.grad

    Parameters without a gradient.  The autograd graph has static shapes and the same structure on
    every rank, so which parameters receive a gradient depends only on the graph variant `key`.
      * A parameter whose gradient is None at the end of backward keeps `.grad is None` (it rides
        through the collective as zeros: the bucket layout is fixed) -- SGD then applies neither weight
        decay nor momentum to it, exactly as on one GPU and in the reference.
      * Buckets are issued strictly in order, so a bucket holding such a parameter would stall every
        later bucket until the end of backward.  The first step of a `key` waits for everything; later
        steps do not wait for parameters that had no gradient under the same `key`.  A gradient that
        shows up for a parameter that was not waited for is a programming error and raises.
      * Cross-rank consistency (a parameter with a gradient on one rank and none on another) cannot be
        acted on without a host synchronisation; one flag per parameter travels with each bucket, a
        device-side counter accumulates disagreements, and `check()` (tests, end of a run) raises on it.
    """

    def __init__(self, module, bucket_bytes=25 * 1024 * 1024, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        params = [p for p in module.parameters() if p.requires_grad]
        params.reverse()                      # gradients arrive roughly in reverse registration order
        self.buckets = []
        cur, size = [], 0
        for p in params:
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self.bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self.bucket_of[p] = bi
        self.device = params[0].device if params else torch.device("cpu")
        self.use_stream = self.device.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.use_stream else None
        self.launch_log = None       # set to [] to record when each bucket is issued (tests)
        self._absent = {}            # key -> set of parameters that produced no gradient under that key
        self._key = None
        self._violations = torch.zeros((), device=self.device)
        self._flag_cache = {}
        self._reset()
        self.active = self.world > 1 or (dist.is_initialized() and _force_collectives())
        if self.active:
            for p in params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def begin(self, key=None):
        """Call before backward.  `key` identifies the graph variant (anything hashable)."""
        self._key = key
        self._reset()

    def _reset(self):
        absent = self._absent.get(self._key, ())
        self.pending = [sum(1 for p in b if p not in absent) for b in self.buckets]
        self.next_to_launch = 0
        self.inflight = []        # (bucket index, flat buffer, work handle, which params had a gradient)
        self._launched = set()

    def _on_grad(self, p):
        bi = self.bucket_of[p]
        if bi in self._launched:
            raise RuntimeError("GradientBuckets: a gradient arrived for a parameter of bucket %d after the bucket was "
                               "issued -- the set of parameters that receive gradients changed without a new "
                               "begin(key)" % bi)
        if p not in self._absent.get(self._key, ()):
            self.pending[bi] -= 1
        self._launch_ready()

    def _launch_ready(self, force=False):
        # strictly in bucket order, so every rank issues the same sequence of collectives
        while self.next_to_launch < len(self.buckets) and (force or self.pending[self.next_to_launch] <= 0):
            bi = self.next_to_launch
            self.next_to_launch += 1
            self._launched.add(bi)
            if self.launch_log is not None:      # (bucket, gradients still to come when it was issued)
                self.launch_log.append((bi, sum(max(n, 0) for n in self.pending[bi + 1:])))
            # a parameter without a gradient contributes zeros (the bucket layout is fixed); one flag per
            # parameter rides along for the cross-rank consistency counter
            had = [p.grad is not None for p in self.buckets[bi]]
            grads = [p.grad if h else torch.zeros_like(p) for p, h in zip(self.buckets[bi], had)]
            # the flag vector of a (bucket, pattern) pair is built once and kept on the device: a host-to-device
            # copy from a hook would synchronise the host with the main stream at every bucket
            fkey = (bi, tuple(had))
            flags = self._flag_cache.get(fkey)
            if flags is None:
                flags = torch.tensor([1.0 if h else 0.0 for h in had], dtype=grads[0].dtype, device=self.device)
                self._flag_cache[fkey] = flags
            pieces = [_mem_flat(g) for g in grads] + [flags]
            if self.use_stream:
                self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self.comm_stream):
                    flat = torch.cat(pieces)
                    for g in grads:
                        g.record_stream(self.comm_stream)
                    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            else:
                flat = torch.cat(pieces)
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.inflight.append((bi, flat, work, had, flags))

    def __call__(self):
        if not self.active:
            return
        self._launch_ready(force=True)        # whatever is left (first step of a key; trailing bucket)
        inv = 1.0 / float(self.world)
        absent = set()
        for bi, flat, work, had, flags in self.inflight:
            work.wait()
            if self.use_stream:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_stream(self.comm_stream)
                flat.record_stream(cur)
            n = len(had)
            # some rank had a gradient where this one had none (or vice versa): replicas would diverge
            self._violations += ((flat[-n:] > 0) & (flat[-n:] < self.world)).sum()
            flat[:-n].mul_(inv)
            dst, src, off = [], [], 0
            for p, h in zip(self.buckets[bi], had):
                k = p.numel()
                if h:
                    dst.append(_mem_flat(p.grad))             # memory order, as it was packed
                    src.append(flat[off:off + k])
                else:
                    # no gradient here: .grad stays None, exactly as on one GPU
                    if p.grad is not None:
                        raise RuntimeError("GradientBuckets: gradient produced after its bucket was issued")
                    absent.add(p)
                off += k
            if dst:
                torch._foreach_copy_(dst, src)                # a few launches per bucket, not one per parameter
        self._absent[self._key] = absent
        self._reset()

    def check(self):
        """Host-synchronising: raises if any parameter ever had a gradient on some ranks only."""
        n = int(self._violations.item())
        if n:
            raise RuntimeError("GradientBuckets: %d parameter-steps had a gradient on some ranks but not on others; "
                               "the replicas have diverged" % n)


def invalidate_derived_state(module):
    """Writes through `.data` (broadcast, checkpoint load, weight surgery) do not bump tensor
    version counters; drop everything cached from parameter values (eval-BN folds, conv.py)."""
    from .conv import invalidate_bn_folds
    invalidate_bn_folds(module)


def broadcast_parameters(module, src=0, group=None):
    """Make every replica start from rank `src`'s parameters and buffers (done once; the
    reference re-broadcast them every step)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
    invalidate_derived_state(module)
