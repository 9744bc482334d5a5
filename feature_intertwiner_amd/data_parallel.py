"""Data-parallel training, one process per GPU over RCCL (torch.distributed backend "nccl"
is RCCL on ROCm).  Replaces the reference's single-process nn.DataParallel
(tools/utils.py:645-654), which re-broadcasts ~386 MB of parameters to every GPU each
step, gathers all outputs to GPU 0 and runs the optimizer there.

Here every rank owns a full replica and its shard of the minibatch (images are
independent through the whole detector), so the only exchanges per step are
  1. an all-reduce (mean) of the gradients, bucketed (~64 MB contiguous slices of the
     model's gradient arena, laid out in the order autograd produces the gradients, reduced
     IN PLACE) and issued from a side HIP stream so that RCCL traffic over xGMI overlaps
     the rest of backward;
  2. one small all-reduce (sum) of the intertwiner class statistics (feat*cnt, cnt),
     ~1 MB -- algebraically the reference's gather-to-GPU-0 + _merge_feat_vec
     (lib/model.py:217-224).
There is no data-path collective besides these.  The same code runs on CPU tensors with
the gloo backend (tests/test_data_parallel_gloo.py).
"""
import torch
import torch.distributed as dist

from . import grad_arena


def conv_wgrad_stream(device):
    from .conv import wgrad_stream          # (conv imports nothing from here; late import keeps CPU-only use light)
    return wgrad_stream(device)


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


def all_reduce_flat_sum(flat, group=None):
    """In place: flat <- sum over the ranks of flat (no autograd; the caller's backward applies the world-size factor,
    see _AllReduceSumIdentityGrad).  Returns the world size.  The kernel form of the statistics exchange
    (intertwiner._MetaStatsReducedFn) reduces its one flat vector of sums with this."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    world = dist.get_world_size(group)
    if world > 1 or _force_collectives():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return world


all_reduce_statistics.flat_sum = all_reduce_flat_sum


class GradientBuckets(object):
    """Bucketed, overlapped gradient all-reduce, IN PLACE on the model's gradient arena.

        sync = GradientBuckets(model)
        ...
        sync.begin(key)      # optional: names the graph variant of this step (e.g. do_meta on/off)
        loss.backward()      # buckets launch from autograd hooks as they fill
        sync()               # waits; p.grad (views of the arena) now hold the rank-mean gradients

    Every trainable parameter owns a slot of ONE persistent buffer (grad_arena.ArenaLayout), laid out in the
    order autograd produces gradients; a bucket is a contiguous ~64 MB slice of it (FI_DP_BUCKET_MB).  The weight-gradient and
    fused-BN-backward kernels accumulate straight into their slots, so for nearly all bytes there is nothing
    to pack: when the last gradient of a bucket has arrived, the few gradients autograd allocated elsewhere
    (library GEMMs of the head FCs, the OT module) are moved into their slots with one multi-tensor copy and
    `.grad` is re-pointed at the slot, then the slice is all-reduced where it lies, on a side HIP stream, while
    backward continues.  `sync()` scales the buffer by 1/world once -- no flat staging buffers, no copy back.

    Parameters without a gradient.  The autograd graph has static shapes and the same structure on
    every rank, so which parameters receive a gradient depends only on the graph variant `key`.
      * A parameter whose gradient is None at the end of backward keeps `.grad is None` (its slot stays
        zero and rides through the collective: the layout is fixed) -- SGD then applies neither weight
        decay nor momentum to it, exactly as on one GPU and in the reference.
      * Buckets are issued strictly in order, so a bucket holding such a parameter would stall every
        later bucket until the end of backward.  A parameter that had no gradient in ABSENT_STEPS consecutive
        steps of a `key` is no longer waited for under that key.  A gradient that shows up for such a parameter
        after its bucket has been issued travels in a LATE extra collective at the end of backward (same sequence
        on every rank: the graph has the same structure everywhere), `late_gradients` counts the event, a warning is
        printed once, and the parameter is waited for again under that key from the next step on.  (Only a gradient
        that a kernel accumulated straight into the arena slot WHILE the slot's collective was in flight cannot be
        recovered; that still raises.)
      * Cross-rank consistency (a parameter with a gradient on one rank and none on another) cannot be
        acted on without a host synchronisation; one flag per parameter travels at the end of each bucket, a
        device-side counter accumulates disagreements, and `check()` (tests, end of a run) raises on it.
    """

    ABSENT_STEPS = 2

    def __init__(self, module, bucket_bytes=None, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.layout = grad_arena.get_layout(module, bucket_bytes)
        self.buckets = [b.params for b in self.layout.buckets]
        self.waits = [b.wait for b in self.layout.buckets]      # == params, except for the pieces of one large parameter
        params = self.layout.params
        self.buckets_of = {}
        for bi, w in enumerate(self.waits):
            for p in w:
                self.buckets_of.setdefault(p, []).append(bi)
        self.device = params[0].device if params else torch.device("cpu")
        import os
        self.use_stream = self.device.type == "cuda" and os.environ.get("FI_DP_COMM_STREAM", "1") != "0"
        self._comm_stream = None     # picked at the first bucket (_lib.pick_stream: next to the streams the step uses)
        self.stream_ordered = False  # RCCL: completion is a stream dependency; other backends: host wait in __call__
        self.launch_log = None       # set to [] to record when each bucket is issued (tests)
        self.profile = None          # set to [] to record HIP events around every collective (bench.py, RCCL only)
        self._absent_run = {}        # key -> {parameter: consecutive steps without a gradient}
        self._key = None
        self._violations = torch.zeros((), device=self.device)
        self.stream_choice = None    # recheck_streams(): the measured concurrency of the picked streams
        self.late_gradients = 0      # gradients that arrived after their bucket had been issued (see the class docstring)
        self._late_warned = False
        self._flag_cache = {}
        self._flag_index = None
        self._started = False
        self._reset()
        self.active = self.world > 1 or (dist.is_initialized() and _force_collectives())
        if self.active:
            self.stream_ordered = self.use_stream and dist.get_backend(group) == "nccl"
            self.arena = self.layout.buffer(self.device)
            idx = [i for b in self.layout.buckets for i in range(b.flag_off, b.flag_off + len(b.params))]
            self._flag_index = torch.tensor(idx, dtype=torch.long, device=self.device)
            for p in params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    @property
    def comm_stream(self):
        if self._comm_stream is None and self.use_stream:
            from . import _lib
            with torch.cuda.device(self.device):
                self._comm_stream = _lib.pick_stream(self.device)
                _lib.name_stream(self._comm_stream, "communication", self.device)
        return self._comm_stream

    def recheck_streams(self):
        """Call at a step boundary AFTER the first step (i.e. after the first real collectives: RCCL has created its own
        streams and channels by then): re-measures which of the process's picked streams still run next to each other and
        replaces those that do not (_lib.stream_report(repick=True)); follows a replaced communication stream.  The
        report is kept in `stream_choice` (bench.py prints it).  Host-synchronising, a few ms: once, not per step."""
        if not (self.active and self.use_stream):
            return None
        from . import _lib
        with torch.cuda.device(self.device):
            rep = _lib.stream_report(self.device, repick=True)
        new = rep["replaced"].get(id(self._comm_stream)) if self._comm_stream is not None else None
        if new is not None:
            self._comm_stream = new
        self.stream_choice = {"GPU_MAX_HW_QUEUES": rep["GPU_MAX_HW_QUEUES"],
                              "streams": [{k: v for k, v in r.items() if k != "detail"} for r in rep["streams"]]}
        return self.stream_choice

    def bucket_bytes(self):
        return [4 * (b.end - b.start) for b in self.layout.buckets]

    def begin(self, key=None):
        """Call before backward.  `key` identifies the graph variant (anything hashable)."""
        self._key = key
        self._reset()
        self._start()

    def _start(self):
        """First touch of a backward pass: the arena must be clear (conv.prepare_step did it at the start of the
        forward pass of a GPU model; models without that step -- the CPU tests -- are cleared here)."""
        if self.active and not self._started:
            self._started = True
            if self.layout.superseded:
                # grad_arena.get_layout() replaced the layout (another bucket size, a changed trainable set): the
                # kernels and the optimiser follow the new one, so must the buckets
                raise RuntimeError("GradientBuckets: the model's gradient layout was rebuilt after this object was "
                                   "created; construct a new GradientBuckets")
            self.arena = self.layout.buffer(self.device)
            # Gradients of an earlier backward pass that are still attached LIVE in the arena (accumulation over
            # micro-batches without zero_grad; zero_grad(set_to_none=False) with no begin(), where AccumulateGrad
            # has already added into the slot when the first hook runs): clearing would discard them.  The reduced
            # means of the earlier passes are identical on every rank, so summing them again over the ranks and
            # dividing by the world size in __call__ leaves them unchanged.
            if not self.layout.fresh and not self.layout.live_gradients(self.arena):
                self.layout.zero(self.device)
            if self.profile is not None and self.use_stream:
                e = torch.cuda.Event(enable_timing=True)
                e.record(torch.cuda.current_stream(self.device))
                self.profile.append(("backward_start", e))

    def _not_waited(self):
        run = self._absent_run.get(self._key)
        if not run:
            return ()
        return {p for p, n in run.items() if n >= self.ABSENT_STEPS}

    def _reset(self):
        skip = self._not_waited()
        self._skip = skip
        self.pending = [sum(1 for p in w if p not in skip) for w in self.waits]
        self.next_to_launch = 0
        self.inflight = []        # (bucket index, work handle, which params had a gradient)
        self._launched = set()
        self._started = False
        self._late = []           # parameters whose gradient arrived after their bucket was issued, in arrival order
        # streams (other than the issuing one) on which gradients of a bucket were produced: the meta loss / OT module
        # runs on the third stream, and autograd calls a parameter's hook with the stream of its forward pass current
        self._grad_streams = [set() for _ in self.waits]

    def _on_grad(self, p):
        self._start()
        cur = torch.cuda.current_stream(self.device) if self.use_stream else None
        late = []                 # this parameter's buckets that had already been issued when the gradient arrived
        for bi in self.buckets_of[p]:
            if bi in self._launched:
                late.append(bi)
                continue
            if cur is not None:
                self._grad_streams[bi].add(cur)
            if p not in self._skip:
                self.pending[bi] -= 1
        if late:
            self._late_gradient(p, cur, late)
        self._launch_ready()

    def _late_gradient(self, p, cur, late_buckets):
        if p.grad is not None and self.layout.holds(p, p.grad, self.arena):
            raise RuntimeError("GradientBuckets: a kernel accumulated the gradient of a parameter (shape %s) into its arena "
                               "slot after the slot's bucket had been issued -- the set of parameters that receive "
                               "gradients changed without a new begin(key)" % (tuple(p.shape),))
        if all(q is not p for q, _, _ in self._late):
            self._late.append((p, cur, tuple(late_buckets)))
        self.late_gradients += 1
        run = self._absent_run.setdefault(self._key, {})
        run.pop(p, None)                      # waited for again from the next step on
        if not self._late_warned:
            self._late_warned = True
            import warnings
            warnings.warn("GradientBuckets: a gradient (parameter of shape %s) arrived after its bucket had been issued; "
                          "it is reduced in a late extra collective (counter: late_gradients)" % (tuple(p.shape),))

    @torch.no_grad()
    def _launch_ready(self, force=False):
        # strictly in bucket order, so every rank issues the same sequence of collectives
        lay, buf = self.layout, self.arena
        while self.next_to_launch < len(self.buckets) and (force or self.pending[self.next_to_launch] <= 0):
            bi = self.next_to_launch
            self.next_to_launch += 1
            self._launched.add(bi)
            b = lay.buckets[bi]
            if self.launch_log is not None:      # (bucket, gradients still to come when it was issued)
                self.launch_log.append((bi, sum(max(n, 0) for n in self.pending[bi + 1:])))
            had = [p.grad is not None for p in b.params]
            # gradients that autograd allocated outside the arena move into their slots (one multi-tensor copy);
            # a parameter without a gradient leaves its slot zero.  (b.wait: a piece of a split parameter moves the
            # whole parameter in before the first piece leaves)
            if self.use_stream:
                cur = torch.cuda.current_stream(self.device)
                for s in self._grad_streams[bi]:
                    if s != cur:
                        cur.wait_stream(s)        # the move below (and the flag write) run on `cur`
            dst, src, moved = [], [], []
            late_now = {id(q) for q, _, _ in self._late}
            for p in b.wait:
                h = p.grad is not None
                if h and not lay.holds(p, p.grad, buf):
                    if id(p) in late_now and self.use_stream:
                        # a split parameter whose EARLIER piece has already been issued: the whole gradient moves in
                        # here, over a slice that piece's collective may still be writing
                        torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
                    elif id(p) in late_now:
                        for _, w, _ in self.inflight:
                            if w is not None:
                                w.wait()
                        self.inflight = [(i, None, hd) for i, _, hd in self.inflight]
                    v = lay.view(p, buf)
                    if v is None:
                        raise RuntimeError("GradientBuckets: parameter with unsupported strides %s" % (p.stride(),))
                    dst.append(v)
                    src.append(p.grad)
                    moved.append((p, v))
            if dst:
                torch._foreach_copy_(dst, src)
                if self.use_stream:
                    # `p.grad = v` drops the last reference to the autograd-allocated gradient; it may have been
                    # allocated on another stream (the OT module's on the third one) and would return to THAT stream's
                    # pool with the copy on `cur` still queued: tell the allocator `cur` uses the block
                    for g in src:
                        g.record_stream(cur)
                for p, v in moved:
                    p.grad = v
            # one flag per parameter rides along for the cross-rank consistency counter; the flag vector of a
            # (bucket, pattern) pair is built once and kept on the device: a host-to-device copy from a hook
            # would synchronise the host with the main stream at every bucket
            fkey = (bi, tuple(had))
            flags = self._flag_cache.get(fkey)
            if flags is None:
                flags = torch.tensor([1.0 if h else 0.0 for h in had], dtype=torch.float32, device=self.device)
                self._flag_cache[fkey] = flags
            buf[b.flag_off:b.flag_off + len(had)].copy_(flags)
            piece = buf[b.start:b.end]
            if self.use_stream:
                from .conv import flush_deferred_wgrads
                flush_deferred_wgrads()                    # queued (batched) weight gradients of this bucket's layers:
                                                           # BEFORE the communication stream takes its dependency
                self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))
                for s in self._grad_streams[bi]:           # gradients produced on other streams (see _reset)
                    self.comm_stream.wait_stream(s)
                wg = conv_wgrad_stream(self.device)        # weight gradients on their own stream (conv.WGRAD_SIDE_STREAM...)
                if wg is not None:
                    self.comm_stream.wait_stream(wg)
                with torch.cuda.stream(self.comm_stream):
                    ev = None
                    if self.profile is not None and self.stream_ordered:
                        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                        ev[0].record(self.comm_stream)
                    work = dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                    if self.stream_ordered:
                        work.wait()              # RCCL: makes comm_stream wait for the collective; the host goes on
                        work = None
                        if ev is not None:
                            ev[1].record(self.comm_stream)
                            self.profile.append(("bucket", bi, 4 * (b.end - b.start), ev[0], ev[1]))
            else:
                work = dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.inflight.append((bi, work, had))

    @torch.no_grad()
    def __call__(self):
        if not self.active:
            return
        self._start()
        if self.profile is not None and self.use_stream:
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(self.device))
            self.profile.append(("backward_end", e))
        self._launch_ready(force=True)        # whatever is left (first steps of a key; trailing bucket)
        late_views = self._reduce_late()
        for bi, work, had in self.inflight:
            if work is not None:
                work.wait()
        if self.use_stream:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
            if self.profile is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record(torch.cuda.current_stream(self.device))
                self.profile.append(("reduced", e))
        buf = self.arena
        # some rank had a gradient where this one had none (or vice versa): replicas would diverge
        f = buf[self._flag_index]
        self._violations += ((f > 0.5) & (f < self.world - 0.5)).sum()
        if self.world > 1:
            buf.mul_(1.0 / float(self.world))         # ONE pass over the arena: sums -> means, in place
        run = self._absent_run.setdefault(self._key, {})
        late = {id(p) for p, _ in late_views}
        for p, v in late_views:
            p.grad = v                         # the slot now holds the rank sum (scaled with the rest above)
        for bi, work, had in self.inflight:
            for p, h in zip(self.buckets[bi], had):
                if h or id(p) in late:
                    run.pop(p, None)
                else:
                    # no gradient here: .grad stays None, exactly as on one GPU
                    if p.grad is not None:
                        raise RuntimeError("GradientBuckets: gradient produced after its bucket was issued")
                    run[p] = run.get(p, 0) + 1
        self.layout.fresh = False
        self._reset()

    def _reduce_late(self):
        """Gradients that arrived after their bucket had left (autograd-allocated, outside the arena): moved into their
        slots -- which the issued bucket left holding the rank sum of zeros -- and all-reduced there, one collective per
        parameter in arrival order (the same on every rank), before the 1/world scaling.  Returns [(param, slot view)]."""
        out = []
        if not self._late:
            return out
        lay, buf = self.layout, self.arena
        for p, s, late_buckets in self._late:
            if p.grad is None:
                continue
            v = lay.view(p, buf)
            if v is None:
                raise RuntimeError("GradientBuckets: parameter with unsupported strides %s" % (p.stride(),))
            # only the slices whose buckets had left when the gradient arrived: a piece of a split parameter whose
            # bucket was still to come has been reduced by that bucket (which also moved the whole gradient in)
            off, n = lay.slot[p]
            slices = []
            for bi in late_buckets:
                b = lay.buckets[bi]
                lo, hi = max(off, b.start), min(off + n, b.end)
                if hi > lo:
                    slices.append(buf[lo:hi])
            moved_in = lay.holds(p, p.grad, buf)
            if self.use_stream:
                cur = torch.cuda.current_stream(self.device)
                if s is not None and s != cur:
                    cur.wait_stream(s)                # the producer of the late gradient
                cur.wait_stream(self.comm_stream)     # the slot's own bucket must have landed before the move
                if not moved_in:
                    v.copy_(p.grad)
                    p.grad.record_stream(cur)         # allocated on the producer's stream, read here on `cur`
                self.comm_stream.wait_stream(cur)
                works = []
                with torch.cuda.stream(self.comm_stream):
                    for piece in slices:
                        work = dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                        if self.stream_ordered:
                            work.wait()
                        else:
                            works.append(work)
            else:
                # the slot's own bucket may still be in flight on a host-waited backend: finish everything first
                for _, w, _ in self.inflight:
                    if w is not None:
                        w.wait()
                self.inflight = [(bi, None, had) for bi, _, had in self.inflight]
                if not moved_in:
                    v.copy_(p.grad)
                works = [dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                         for piece in slices]
            for work in works:
                work.wait()
            out.append((p, v))
        return out

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
