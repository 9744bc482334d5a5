"""One training iteration -- the loop body of train_epoch (lib/workflow.py:152-230 of the
reference): forward, meta (intertwiner) loss, loss composition, backward, gradient
clipping, SGD step.  The reference's epoch/stage loops, logging, visdom and checkpointing
are out of scope."""
import torch

import os

from . import optim
from ._lib import const_tensor
from ._lib import run_on_side_stream as _run_on_side_stream

META_SIDE_STREAM = os.environ.get("FI_META_SIDE", "1") != "0"      # A/B switch (scripts/ab_env.sh)
META_SIDE_WITH_REDUCE = os.environ.get("FI_META_SIDE_DP", "1") != "0"   # ... also with the statistics all-reduce in it


def set_optimizer(net, opt):
    """tools/utils.py:474-501: SGD, weight decay on everything except BatchNorm affine
    parameters (names containing 'bn')."""
    wo_bn = [p for n, p in net.named_parameters() if p.requires_grad and 'bn' not in n]
    only_bn = [p for n, p in net.named_parameters() if p.requires_grad and 'bn' in n]
    return torch.optim.SGD([{'params': wo_bn, 'weight_decay': opt.WEIGHT_DECAY}, {'params': only_bn}],
                           lr=opt.INIT_LR, momentum=opt.MOMENTUM)


def compute_loss(model, inputs, do_meta=True, world_size=1, reduce_fn=None):
    """Returns (loss to back-propagate, detached dict of its terms) -- lib/workflow.py:180-221.

    Data parallel: the reference total is mean_g(L_det,g) + meta(global statistics).
    With one process per GPU and gradients AVERAGED over ranks, every rank adds the SAME meta
    term evaluated from all-reduced statistics; the factor world_size that the path back into the
    local statistics needs is applied inside `reduce_fn`'s backward (data_parallel.py), so that
    parameters owned by the meta loss itself (ot_loss.*) are not over-weighted."""
    cfg = model.config
    (merged_loss, big_feat, big_cnt, small_feat, small_cnt, big_loss, small_output_all, small_gt_all,
     fpn_ot_loss) = model(inputs, 'train')
    detailed = merged_loss.mean(0)
    if cfg.DEV.SWITCH and not cfg.DEV.BASELINE:
        if cfg.DEV.DIS_REG_LOSS:
            # workflow.py:184-187 zeroes the VALUES (`.data[i] = 0`) of rpn_bbox, mrcnn_bbox and mask;
            # torch.sum's backward does not look at values, so their gradients still flow -- mirrored
            off = const_tensor([0., 1., 0., 1., 1.], detailed.device)
            detailed = detailed - detailed.detach() * off
        feats = [big_feat, big_cnt, small_feat, small_cnt, small_output_all, small_gt_all]
        fork = getattr(model, "_stats_ready", None)
        if fork is not None and META_SIDE_STREAM and (reduce_fn is None or META_SIDE_WITH_REDUCE) and big_feat.is_cuda:
            # latency-bound (~70 small kernels + one Sinkhorn launch) and independent of the box / mask heads the main
            # stream still has queued: evaluated from the point where the statistics were complete (autograd runs the
            # backward of these ops on the stream their forward ran on, with the joins it needs).  Data parallel: the
            # statistics all-reduce is issued from that stream too -- every rank issues its collectives in the same
            # program order (this one in forward, the gradient buckets in backward), and the process group runs them
            # in issue order whatever stream they were issued from.
            model._stats_ready = None
            # (`reads`: the statistics were allocated on this stream; the side stream's ops keep some for their backward)
            meta = _run_on_side_stream(lambda: model.meta_loss(feats, reduce_fn=reduce_fn), after=fork, reads=feats)()
        else:
            big_done = getattr(getattr(model, "dev_roi", None), "big_done", None)
            if big_done is not None:          # the big branch ran on the third stream (Dev.forward)
                here = torch.cuda.current_stream(big_feat.device)
                here.wait_event(big_done)
                big_feat.record_stream(here)
                big_cnt.record_stream(here)
            meta = model.meta_loss(feats, reduce_fn=reduce_fn)
        meta = torch.where(meta < 0, torch.zeros_like(meta), meta)       # workflow.py:196-200
        meta = meta * cfg.DEV.LOSS_FAC if do_meta else torch.zeros_like(meta)
    else:
        meta = detailed.new_zeros(())
    if cfg.DEV.SWITCH and cfg.DEV.BIG_SUPERVISE:                          # workflow.py:212-217
        big = big_loss.mean() * cfg.DEV.BIG_LOSS_FAC
    else:
        big = detailed.new_zeros(())
    fpn_ot = cfg.TRAIN.FPN_OT_LOSS_FAC * fpn_ot_loss.mean()
    loss = detailed.sum() + meta + big + fpn_ot
    terms = {"rpn_cls": detailed[0], "rpn_bbox": detailed[1], "mrcnn_cls": detailed[2],
             "mrcnn_bbox": detailed[3], "mrcnn_mask": detailed[4], "meta": meta, "big": big,
             "total": detailed.sum() + meta + big}
    return loss, {k: v.detach() for k, v in terms.items()}


def backward_scaled(model, loss):
    """loss.backward() with the static loss scale of the 16-bit paths (cfg.TRAIN.LOSS_SCALE; 1 = plain backward).  Returns
    the function that divides the scale out of the gradients again (one multi-tensor pass; call it after the gradient
    exchange, before clipping).  Custom training loops on the 16-bit paths must go through this function (a plain
    `loss.backward()` silently loses the scale) and should pass `skip_nonfinite=True` to `optim.clip_and_step`, as
    `train_step` does: a data gradient above 65504 / LOSS_SCALE gives an inf norm, and that step must not reach the
    weights."""
    s = float(getattr(model.config.TRAIN, "LOSS_SCALE", 1.0) or 1.0)
    if s == 1.0:
        loss.backward()
        return lambda: None
    (loss * s).backward()

    def unscale():
        grads = [p.grad for p in model.parameters() if p.grad is not None]
        if grads:
            torch._foreach_mul_(grads, 1.0 / s)
    return unscale


def train_step(model, optimizer, inputs, do_meta=True, grad_sync=None, world_size=1, reduce_fn=None):
    """zero_grad / backward / clip_grad_norm(MAX_GRAD_NORM) / step (lib/workflow.py:226-230).
    `grad_sync` (data parallel) is called after backward and must leave the rank-averaged
    gradients in place BEFORE clipping -- the reference clips the reduced gradient."""
    cfg = model.config
    optimizer.zero_grad(set_to_none=True)
    loss, terms = compute_loss(model, inputs, do_meta, world_size, reduce_fn)
    if grad_sync is not None and hasattr(grad_sync, "begin"):
        grad_sync.begin(("do_meta", bool(do_meta)))     # the graph variant decides which parameters get gradients
    unscale = backward_scaled(model, loss)
    join = getattr(model, "_side_join", None)
    if join is not None:          # work of the forward pass that nothing reads (MaskRCNN.forward) ends before the weights move
        model._side_join = None
        join()
    if grad_sync is not None:
        grad_sync()
    unscale()
    if optim.supported(optimizer):
        # clip_grad_norm_ + SGD step in three launches (csrc/sgd.hip) instead of ~8 passes over all parameters
        # under a loss scale (fp16 operands) a step whose gradient norm overflowed is skipped on the device
        scaled = float(getattr(cfg.TRAIN, "LOSS_SCALE", 1.0) or 1.0) != 1.0
        optim.clip_and_step(optimizer, cfg.TRAIN.MAX_GRAD_NORM if cfg.TRAIN.CLIP_GRAD else None, skip_nonfinite=scaled)
        return terms
    if cfg.TRAIN.CLIP_GRAD:
        torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None],
                                       cfg.TRAIN.MAX_GRAD_NORM)
    optimizer.step()
    return terms


def compare_backward_forms(model, inputs, do_meta=True, generator_seed=3, skip=None, detail=0, forms=("default", "dense"),
                           keep_gradients=False, rpn_mask_from_dense=False):
    """Differential check of the default backward pass against its DENSE form on the same weights, the same inputs and
    the same random draws (no optimiser step; the intertwiner history buffer is restored between the two passes).

    Default form: ReLU masks in the readers' data-gradient epilogues, BatchNorm scale in W^T,
    gradients that are identically zero not computed (the RPN's graph on the sampled anchors, the mask head's on the
    slots that can hold positives -- lib/layers.py:808-934, lib/model.py:442).  Dense form (`conv.GATES =
    conv._UNSCALED_BACKWARD = False`): what the reference's autograd does -- dense RPN / mask-head gradients, one
    elementwise pass per BatchNorm layer, autograd accumulating every multi-reader gradient.

    Returns {"loss": (default, dense), "loss_rel": |difference| / |dense|, "max_rel_dev": the largest, over parameters,
    of max|g_default - g_dense| / max|g_dense|, "worst": its parameter, "params": parameters compared,
    "none_sets_equal": the same parameters have no gradient in both forms}.  skip(name) -> True leaves a parameter out
    of max_rel_dev; detail=k adds "table": the k largest deviations as (name, deviation, max|g_default|, max|g_dense|)."""
    from . import conv as C
    keep = (C.GATES, C._UNSCALED_BACKWARD)
    fb = model.feature_buffer
    saved_fb = None if fb is None else (fb.buffer.clone(), fb.buffer_cnt.clone())
    saved_gen = model.generator
    # a stateful external proposal source (synthetic.SyntheticProposals draws new jitter at every call) must hand both
    # passes the same rows
    ext_gen = getattr(model, "external_proposals", None)
    if ext_gen is not None and not hasattr(ext_gen, "get_state"):
        ext_gen = getattr(ext_gen, "gen", None)
    ext_state = ext_gen.get_state() if ext_gen is not None else None
    dev = next(model.parameters()).device
    out = {}
    rpn = getattr(model, "rpn", None)
    try:
        for slot, form in zip(("default", "dense"), forms):     # (forms=("default", "default"): run-to-run repeatability)
            C.GATES = C._UNSCALED_BACKWARD = (form == "default")
            if rpn is not None:     # the default form's pass computes BOTH evaluations of the shared convolution (the dense
                rpn._probe = {} if (slot == "default" and form == "default") else None       # kernel and the row form)
                if rpn._probe is not None and rpn_mask_from_dense:
                    rpn._probe["mask_from_dense"] = True      # replay of a verified event (check_backward_forms)
            if model.feature_buffer is not None and saved_fb is not None:
                model.feature_buffer.buffer, model.feature_buffer.buffer_cnt = saved_fb[0].clone(), saved_fb[1].clone()
            elif saved_fb is None:
                model.feature_buffer = None
            model.generator = torch.Generator(device=dev).manual_seed(generator_seed)
            if ext_state is not None:
                ext_gen.set_state(ext_state)
            for p in model.parameters():
                p.grad = None
            loss, _ = compute_loss(model, list(inputs), do_meta, 1, None)
            unscale = backward_scaled(model, loss)
            join = getattr(model, "_side_join", None)
            if join is not None:
                model._side_join = None
                join()
            unscale()
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            out[slot] = (float(loss.detach()),
                         {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in model.named_parameters()})
            if slot == "default" and rpn is not None and rpn._probe and "z_rows" in rpn._probe:
                out["probe"] = rpn._probe
    finally:
        C.GATES, C._UNSCALED_BACKWARD = keep
        if rpn is not None:
            rpn._probe = None
        model.generator = saved_gen
        if saved_fb is not None and model.feature_buffer is not None:
            model.feature_buffer.buffer, model.feature_buffer.buffer_cnt = saved_fb
        for p in model.parameters():
            p.grad = None
    (l_a, g_a), (l_b, g_b) = out["default"], out["dense"]
    probe_default = out.get("probe")
    worst, worst_name, n, table = 0.0, None, 0, []
    for name, ref in g_b.items():
        got = g_a[name]
        if ref is None or got is None or (skip is not None and skip(name)):
            continue
        n += 1
        dev_rel = float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
        if detail:
            table.append((name, dev_rel, float(got.abs().max()), float(ref.abs().max())))
        if dev_rel > worst:
            worst, worst_name = dev_rel, name
    none_equal = {k for k, v in g_a.items() if v is None} == {k for k, v in g_b.items() if v is None}
    res = {"loss": (l_a, l_b), "loss_rel": abs(l_a - l_b) / (abs(l_b) + 1e-30), "max_rel_dev": worst,
           "worst": worst_name, "params": n, "none_sets_equal": none_equal}
    # The one legitimate discontinuity between the forms: the default form evaluates the RPN's shared 3 x 3 convolution
    # at the sampled anchors as a matrix product (RPN.forward_rows: another summation order than the dense kernel), so a
    # pre-activation within rounding of zero can fall on the other side of its ReLU.  One (anchor, channel) mask bit
    # then differs: the whole gradient of that channel at that anchor -- 1e-3..1e-2 of the channel's bias gradient, and
    # what it sends down the backbone.  Recognisable by its footprint: single channels of rpn.conv_shared.
    for name, ref in g_b.items():
        if name.endswith("rpn.conv_shared.bias") and ref is not None and g_a[name] is not None:
            d = (g_a[name] - ref).abs() / (ref.abs().max() + 1e-30)
            res["rpn_relu_boundary_channels"] = int((d > 1e-4).sum())
            res["rpn_relu_boundary_which"] = [int(c) for c in torch.nonzero(d > 1e-4).flatten().tolist()[:8]]
            if res["rpn_relu_boundary_channels"] and probe_default is not None:
                res["rpn_relu_boundary_evidence"] = _relu_boundary_evidence(
                    probe_default, model.rpn, res["rpn_relu_boundary_which"])
    if detail:
        res["table"] = sorted(table, key=lambda r: -r[1])[:detail]
    if keep_gradients:
        res["gradients"] = (g_a, g_b)
    return res


def _dense_at_rows(maps, image, anchor, valid, per_loc):
    from .sub_module import RPN
    return RPN.dense_at_rows(maps, image, anchor, valid, per_loc)


def _relu_boundary_evidence(probe, rpn, channels, ulps=16.0):
    """Is a differing channel of rpn.conv_shared really a ReLU-boundary event?  For every channel c in `channels`: the
    sampled rows (image, anchor) at which the two evaluations of the shared 3 x 3 convolution -- the dense kernel
    (`RPN.forward`, whose output the dense form's backward masks with) and the row form (`RPN.forward_rows`) --
    DISAGREE on the sign of the pre-activation, and for each such row the float64 pre-activation z (from the row's own
    3 x 3 patch) against the rounding scale of its summands, S = sum|patch . w| + |bias|.  An event requires
    |z| <= ulps * 2^-24 * S: at that distance from zero two fp32 summation orders may land on different sides; a row
    that disagrees with |z| far above it is a bug in one of the two evaluations.
    Returns [{"channel", "rows_disagreeing", "max_abs_z_over_scale" (in units of 2^-24 * S), "within_rounding"}]."""
    img, anchor, valid, per_loc = probe["rows"]
    patches, z_rows, maps = probe["patches"], probe["z_rows"], probe.get("dense_y", [])
    cs = rpn.conv_shared
    ws = cs.weight.detach().permute(0, 2, 3, 1).reshape(cs.weight.shape[0], -1)
    dense = type(rpn).dense_at_rows(maps, img, anchor, valid, per_loc) if hasattr(type(rpn), "dense_at_rows") else \
        _dense_at_rows(maps, img, anchor, valid, per_loc)
    out = []
    for c in channels:
        dis = valid & ((z_rows[:, c] > 0) != (dense[:, c] > 0))
        rec = {"channel": int(c), "rows_disagreeing": int(dis.sum()), "max_abs_z_over_scale": None,
               "within_rounding": False}
        if rec["rows_disagreeing"]:
            pr = patches[dis].double()
            w = ws[c].double()
            z64 = pr @ w + float(cs.bias[c])
            scale = pr.abs() @ w.abs() + abs(float(cs.bias[c]))
            ratio = (z64.abs() / (scale * 2.0 ** -24 + 1e-300)).max()
            rec["max_abs_z_over_scale"] = float(ratio)
            rec["within_rounding"] = bool(ratio <= ulps)
            idx = torch.nonzero(dis).flatten()[:4]
            rec["rows"] = [{"image": int(img[i]), "anchor": int(anchor[i]), "z_row_form": float(z_rows[i, c]),
                            "y_dense_kernel": float(dense[i, c]), "z_float64": float(z64[j]), "summand_scale": float(scale[j])}
                           for j, i in enumerate(idx.tolist())]
        out.append(rec)
    return out


def check_backward_forms(model, inputs, bar=2e-5, attempts=3, **kw):
    """compare_backward_forms with the RPN's ReLU-boundary events (see there) taken out -- by VERIFYING them, not by
    their footprint: when the comparison misses `bar` and single channels of rpn.conv_shared carry the deviation,
      1. `_relu_boundary_evidence` must show, at EVERY differing channel, a sampled (image, anchor) row at which the dense
         kernel and the row form disagree on the sign of the pre-activation AND whose float64 pre-activation is within
         16 x 2^-24 of the magnitude of its summands (otherwise: "boundary_unverified", the comparison stands as failed);
      2. the comparison is then REPLAYED on the same draws with the row form taking its ReLU mask bits from the dense
         kernel's output (`rpn_mask_from_dense`): the only thing that changes is on which side of zero those
         pre-activations count, and the replay must meet `bar` like any other pass.
    (Rounds 3-5 retried with other random anchors instead and accepted an event by its footprint; an event at a POSITIVE
    anchor -- all of them are sampled under every draw -- came back on every retry.)  Returns the last result plus
    "attempts" (1 or 2) and "boundary_events" ([{max_rel_dev, evidence}] of the pass that was set aside)."""
    events = []
    r = compare_backward_forms(model, inputs, generator_seed=3, **kw)
    if r["max_rel_dev"] > bar and r.get("rpn_relu_boundary_channels", 0) >= 1:
        ev = r.get("rpn_relu_boundary_evidence")
        if not ev or not all(e["rows_disagreeing"] >= 1 and e["within_rounding"] for e in ev):
            r["boundary_unverified"] = True
        else:
            events.append({"max_rel_dev": r["max_rel_dev"], "evidence": ev})
            r = compare_backward_forms(model, inputs, generator_seed=3, rpn_mask_from_dense=True, **kw)
    r["attempts"] = len(events) + 1
    r["boundary_events"] = events
    return r
