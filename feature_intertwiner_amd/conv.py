"""Dense convolution layers of the detector on the MI355X matrix cores.

`Conv2d`, `ConvTranspose2x2` and `Conv1d` are drop-in subclasses of the torch modules
(same constructor, same parameter names and shapes -> state dicts of the reference load
unchanged) whose forward/backward run the hand-written fp32 MFMA implicit-GEMM kernels of
csrc/conv_igemm.hip:
    forward        fi_conv2d_forward
    grad input     fi_conv2d_forward on dY with the flipped/transposed weight (stride 1), or
                   on a zero-stuffed dY (stride > 1)
    grad weight    fi_conv2d_weight_grad
    grad bias      a plain reduction
"Full-window" convolutions whose kernel covers the whole input (the 7x7 classifier conv on
7x7 crops, lib/sub_module.py:707; the 7x7 conv of feat_extract on 7x7 maps, :333) are
plain matrix products and go to the library GEMM.
"""
import contextlib
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, grad_arena

# bench.py sets this to a dict to accumulate the algorithmic flops (2*N*Cout*OH*OW*Cin*R*S) of
# every launch, keyed by the device kernel instance (see _lib.conv_kernel_key).
FLOP_LOG = None
LIVE_LOG = None
# bench.py sets this to a list to record the (N, Cin, H, W, Cout, R, S, stride, padding) of every
# forward convolution of a step (used to time the same stack on the host CPU for cpu_baseline).
SHAPE_LOG = None


# Arithmetic of the MFMA convolutions: "fp32" (exact, v_mfma_f32_32x32x2_f32 -- the headline), "bf16" or "fp16"
# (operands rounded to the 16-bit type, fp32 accumulation, v_mfma_f32_32x32x16_{bf16,f16} -- BASELINE configs[4]'s reduced-
# precision conv path).  Read when a forward pass runs; the backward of that pass uses the same setting.
_PRECISION = "fp32"
RING_1X1 = True        # A/B switch of the persistent 1x1 kernel (conv1x1_ring_kernel); FI_NO_RING1X1=1 does the same in C


def set_conv_precision(mode):
    global _PRECISION
    if mode not in ("fp32", "bf16", "fp16"):
        raise ValueError("conv precision must be 'fp32', 'bf16' or 'fp16'")
    _PRECISION = mode


def conv_precision():
    return _PRECISION


_LOWP = {"bf16": ("bf16", torch.bfloat16), "fp16": ("f16", torch.float16)}     # precision -> (entry-point suffix, dtype)


def _lowp_fn(L, stem, precision, tail=""):
    """fi_<stem>_{bf16,f16}<tail> of the loaded library."""
    return getattr(L, "fi_%s_%s%s" % (stem, _LOWP[precision][0], tail))


def _log_shape(x, w, stride, padding):
    if SHAPE_LOG is not None:
        SHAPE_LOG.append((x.shape[0], x.shape[1], x.shape[2], x.shape[3], w.shape[0], w.shape[2], w.shape[3],
                          tuple(stride), tuple(padding)))


def _log_flops(kind, cout, R, S, flops, pixels=None, cin=None, patch=0, batch=1):
    if FLOP_LOG is not None:
        if patch:
            k = {1: "conv3x3_patch", 2: "conv3x3_patch_flat", 3: "conv1x1_reg"}[patch]
        else:
            k = ("conv_" + kind) if kind.startswith("bf16") else _lib.conv_kernel_key(kind, cout, R, S, pixels, cin, batch)
        e = FLOP_LOG.setdefault(k, [0, 0])
        e[0] += 1
        e[1] += flops


def _live_share(live, capacity, tag="gemm"):
    """Share of a static-capacity batch that is real, for the flop log only (a profiled pass may read the device count;
    the timed step never does: FLOP_LOG is None there)."""
    if FLOP_LOG is None or live is None or capacity <= 0:
        return 1.0
    share = min(1.0, float(int(live.reshape(-1)[0].item())) / capacity)
    if LIVE_LOG is not None:                 # scripts/conv_shapes.py: the share of every live-count launch, in order
        LIVE_LOG.append((tag, share))
    return share


def _dense(w):
    """A weight as the kernels can read it: NCHW-contiguous or channels-last-contiguous, no copy if it
    already is one of the two (parameters with Cin % 16 == 0 are stored channels-last, see
    prepare_step)."""
    if w.is_contiguous() or (w.dim() == 4 and w.is_contiguous(memory_format=torch.channels_last)):
        return w
    return w.contiguous()


def _conv_fwd(x, w, b, stride, padding, relu=False, scale=None, residual=None, out_hw=None,
              out_channels_last=False, w_tap_major=False, flip_taps=False, precision=None, gate=None, live=None):
    """w is [Cout, Cin, R, S].  When Cin % 16 == 0 the kernel's tap-major fast path is used: the
    weight is handed over channels-last ([Cout, R, S, Cin]; a copy of at most a few MB).
    out_channels_last: y is returned in torch.channels_last memory format ([N,OH,OW,Cout] in
    memory, written that way by the kernel epilogue)."""
    L = _lib.load()
    N, Cin, H, W = x.shape
    if w_tap_major:                       # w is already [Cout, R, S, Cin] contiguous
        Cout, R, S, _ = w.shape
        layout = 2 if flip_taps else 1    # 2: taps applied in reverse order (data gradient)
        assert Cin % 16 == 0 and R * S <= 64 and w.is_contiguous()
    else:
        Cout, _, R, S = w.shape
        layout = 0
        if Cin % 16 == 0 and R * S <= 64:
            layout = 1
            if R * S > 1:
                w = w.permute(0, 2, 3, 1).contiguous()       # free for a channels-last parameter
        else:
            w = w.contiguous()
    OH = (H + 2 * padding[0] - R) // stride[0] + 1
    OW = (W + 2 * padding[1] - S) // stride[1] + 1
    if out_hw is not None:
        OH, OW = out_hw
    prec = _PRECISION if precision is None else precision
    share = _live_share(live, N, "conv")    # (algorithmic flops: the live images only)
    if not (prec in _LOWP and layout >= 1 and Cin % 32 == 0):
        _log_flops("fwd", Cout, R, S, share * 2 * N * Cout * OH * OW * Cin * R * S, N * OH * OW,
                   patch=(_lib.patch_mode(N, Cin, H, W, Cout, R, S, stride, padding, layout >= 1,
                                          (OH, OW) == (H, W), out_channels_last) or
                          (3 if _lib.reg1x1_mode(N, Cin, H, W, Cout, R, S, stride, padding, out_channels_last) else 0))
                   if FLOP_LOG is not None else 0)
    y = torch.empty((N, Cout, OH, OW), device=x.device, dtype=torch.float32,
                    memory_format=torch.channels_last if out_channels_last else torch.contiguous_format)
    bf16 = prec in _LOWP and layout >= 1 and Cin % 32 == 0          # "bf16" here and below: either 16-bit operand type
    if gate is not None:
        # y * (gate > 0) in the epilogue: the data gradient of a layer whose input is a ReLU output
        assert not out_channels_last and gate.shape == y.shape and gate.is_contiguous()
    fn = _lowp_fn(L, "conv2d_forward_gated", prec) if bf16 else L.fi_conv2d_forward_gated
    if bf16:
        _log_flops("bf16_fwd", Cout, R, S, share * 2 * N * Cout * OH * OW * Cin * R * S)
        # 3x3 / stride 1 / pad 1 on maps whose width is a multiple of 16 (or 14-wide RoI maps): patch kernel with the
        # weights converted to bf16 once per step (cached like W^T) instead of inside every workgroup
        mt = (Cout + 127) // 128
        tiled = W % 4 == 0 and W >= 16 and ((N * H + 7) // 8) * ((W + 15) // 16) * mt >= 192
        # flat 128-pixel tiles for the 14 x 14 RoI maps (even widths below 16)
        flat = W % 16 != 0 and W < 16 and W % 2 == 0 and (W + 126) // W + 2 <= 13 and \
            ((N * H * W + 127) // 128) * mt >= 512
        if (R, S) == (3, 3) and tuple(stride) == (1, 1) and tuple(padding) == (1, 1) and (OH, OW) == (H, W) and \
                (tiled or flat) and Cout > 64 and not out_channels_last and out_hw is None and \
                (residual is None or residual.data_ptr() % 16 == 0) and (gate is None or gate.data_ptr() % 16 == 0):
            wb = _cached_bf16(w, _LOWP[prec][1])
            with torch.cuda.device(x.device):
                _lib.check(_lowp_fn(L, "conv3x3_forward_gated", prec, "w")(_lib.ptr(x), _lib.ptr(wb), _lib.ptr(b), _lib.ptr(scale),
                                                      _lib.ptr(residual), _lib.ptr(gate), _lib.ptr(y), N, Cin, H, W, Cout,
                                                      1 if relu else 0, 1 if layout == 2 else 0, _lib.current_stream()),
                           "fi_conv3x3_forward_bf16w")
            return y
        # 1x1 / stride 1 with whole quads and 64-channel stages: weights-in-registers kernel, bf16 weights cached
        if (R, S) == (1, 1) and tuple(stride) == (1, 1) and tuple(padding) == (0, 0) and (H * W) % 4 == 0 and \
                Cin % 64 == 0 and Cout > 64 and not out_channels_last and out_hw is None and \
                ((N * H * W + 127) // 128) * ((Cout + 127) // 128) >= 192 and \
                (residual is None or residual.data_ptr() % 16 == 0) and (gate is None or gate.data_ptr() % 16 == 0):
            wb = _cached_bf16(w, _LOWP[prec][1])
            with torch.cuda.device(x.device):
                _lib.check(_lowp_fn(L, "conv1x1_forward_gated", prec, "w")(_lib.ptr(x), _lib.ptr(wb), _lib.ptr(b), _lib.ptr(scale),
                                                      _lib.ptr(residual), _lib.ptr(gate), _lib.ptr(y), N, Cin, H * W, Cout,
                                                      1 if relu else 0, _lib.current_stream()),
                           "fi_conv1x1_forward_bf16w")
            return y
    if RING_1X1 and not bf16 and layout >= 1 and R * S == 1 and live is None and out_hw is None and not out_channels_last and _WF:
        e = _WF.get(w.data_ptr())
        if e is not None and e[3]() is not None and e[2] == (Cout, Cin) and (e[1] is None or e[1] == w._version) and \
                L.fi_conv1x1_ring_eligible(N, Cin, H, W, Cout, 1, 1, stride[0], stride[1], padding[0], padding[1], 0,
                                           _lib.ptr(x), _lib.ptr(y), _lib.ptr(residual), _lib.ptr(gate)) == 1:
            w, layout = e[0], 3         # the persistent 1x1 kernel (csrc/conv1x1_ring.hip)
    with torch.cuda.device(x.device):
        if live is not None and bf16:
            _lib.check(_lowp_fn(L, "conv2d_forward_live", prec)(
                _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(scale), _lib.ptr(residual), _lib.ptr(gate), _lib.ptr(y), N, Cin,
                H, W, Cout, R, S, stride[0], stride[1], padding[0], padding[1], 1 if relu else 0, layout,
                OH if out_hw is not None else 0, OW if out_hw is not None else 0, 1 if out_channels_last else 0,
                _lib.ptr(live), _lib.current_stream()), "fi_conv2d_forward_live_16")
            return y
        if live is not None and not bf16:
            # live [1] int32 on the device: only the first live[0] images are real (fi_conv2d_forward_live)
            _lib.check(L.fi_conv2d_forward_live(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(scale), _lib.ptr(residual),
                                                _lib.ptr(gate), _lib.ptr(y), N, Cin, H, W, Cout, R, S, stride[0], stride[1],
                                                padding[0], padding[1], 1 if relu else 0, layout,
                                                OH if out_hw is not None else 0, OW if out_hw is not None else 0,
                                                1 if out_channels_last else 0, _lib.ptr(live), _lib.current_stream()),
                       "fi_conv2d_forward_live")
            return y
        _lib.check(fn(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(scale), _lib.ptr(residual), _lib.ptr(gate),
                                       _lib.ptr(y), N, Cin, H, W, Cout,
                                       R, S, stride[0], stride[1], padding[0], padding[1], 1 if relu else 0,
                                       layout, OH if out_hw is not None else 0, OW if out_hw is not None else 0,
                                       1 if out_channels_last else 0,
                                       _lib.current_stream()), "fi_conv2d_forward")
    return y


class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, padding, residual=None, dx_gate=False, dx_give_to=None):
        _lib.require_cuda(x, w)
        x = x.contiguous().float()
        w = _dense(w.float())
        bc = b.contiguous().float() if b is not None else None
        ctx.save_for_backward(x, w)
        ctx.dx_gate = bool(dx_gate)
        ctx.dx_give_to = dx_give_to
        ctx.conf = (tuple(stride), tuple(padding), b is not None)
        ctx.bias_ptr = b.data_ptr() if b is not None else 0
        ctx.precision = _PRECISION
        _log_shape(x, w, stride, padding)
        res = residual.contiguous().float() if residual is not None else None      # y = conv(x) + b + residual
        return _conv_fwd(x, w, bc, stride, padding, residual=res)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, has_bias = ctx.conf
        dy = dy.contiguous().float()
        if has_bias and ctx.needs_input_grad[2]:
            dx, dw, db = _conv_backward(ctx.needs_input_grad, x, w, dy, stride, padding, want_db=True,
                                        precision=ctx.precision, bias_ptr=ctx.bias_ptr, gate=x if ctx.dx_gate else None)
        else:
            dx, dw = _conv_backward(ctx.needs_input_grad, x, w, dy, stride, padding, precision=ctx.precision,
                                    gate=x if ctx.dx_gate else None)
            db = None
        if ctx.dx_give_to is not None and dx is not None:
            ctx.dx_give_to.value = dx         # added (and masked) inside the data-gradient kernel of x's other reader
            dx = None
        return dx, dw, db, None, None, (dy if ctx.needs_input_grad[5] else None), None, None


def _class_taps(a, size_k, s, p):
    """Taps of a stride-s convolution that reach input positions ih = s*q + a:
    r = r0 + s*t (t < T); the output index is q + c0 - t."""
    r0 = (a + p) % s
    T = 0 if r0 >= size_k else (size_k - r0 + s - 1) // s
    c0 = (a + p - r0) // s
    return r0, T, c0


class _Compact(object):
    """The data gradient of a 1x1 / stride-2 convolution in COMPACT form: only the input positions (2i, 2j) can be
    non-zero, `t` [N, Cin, ceil(H/2), ceil(W/2)] holds them.  Handed from a projection shortcut's backward to the
    backward of the block's first convolution (same input, same stride), which adds it inside its own compact
    data-gradient kernel before ONE interleave pass writes the full tensor."""
    __slots__ = ("t", "hw")

    def __init__(self, t, hw):
        self.t, self.hw = t, hw

    def expand(self, add=None):
        return _interleave2({(0, 0): self.t}, add, self.t.shape[0], self.t.shape[1], self.hw[0], self.hw[1])


def _interleave2(classes, add, N, C, H, W, gate=None):
    """dx [N,C,H,W] from the residue-class results of a stride-2 data gradient (fi_stride2_interleave): one pass,
    every element written once; `add` [N,C,H,W] is added on the way, the result multiplied by (gate > 0)."""
    L = _lib.load()
    some = next(iter(classes.values()))
    dx = torch.empty((N, C, H, W), device=some.device, dtype=torch.float32)
    g = lambda a, b: _lib.ptr(classes[(a, b)].contiguous()) if (a, b) in classes else None
    keep = [classes[k].contiguous() for k in classes]      # noqa: F841  (alive until the launch is enqueued)
    with torch.cuda.device(dx.device):
        _lib.check(L.fi_stride2_interleave_gated(g(0, 0), g(0, 1), g(1, 0), g(1, 1), _lib.ptr(add), _lib.ptr(gate),
                                                 _lib.ptr(dx), N * C, H, W, _lib.current_stream()),
                   "fi_stride2_interleave")
    return dx


def _strided_dgrad(dz, w, in_hw, stride, padding, precision=None, add=None):
    """Data gradient of a strided convolution WITHOUT zero-stuffing: the input positions split into
    stride_h*stride_w residue classes; each class is a stride-1 correlation of dz with the sub-kernel
    of the taps that reach it (e.g. 3x3/stride 2/pad 1: 1, 2, 2 and 4 taps instead of 9 everywhere).
    Stride 2: the class results are interleaved (and `add` added) by one kernel; other strides: strided copies."""
    N, Cout = dz.shape[0], dz.shape[1]
    Cin, R, S = w.shape[1], w.shape[2], w.shape[3]
    H, W = in_hw
    sh, sw = stride
    two = (sh, sw) == (2, 2)
    dx = None if two else dz.new_zeros(N, Cin, H, W)
    classes = {}
    for a in range(min(sh, H)):
        r0, Th, ch0 = _class_taps(a, R, sh, padding[0])
        qa = (H - a + sh - 1) // sh
        for b in range(min(sw, W)):
            s0, Tw, cw0 = _class_taps(b, S, sw, padding[1])
            qb = (W - b + sw - 1) // sw
            if Th == 0 or Tw == 0:
                continue
            pad_h, pad_w = Th - 1 - ch0, Tw - 1 - cw0
            assert pad_h >= 0 and pad_w >= 0, "unsupported stride/padding combination"
            # taps r0 + sh*t, t = Th-1 .. 0 (and the same along the width): a strided slice, reversed.
            # (Indexing with Python lists would build index tensors on the host: a synchronising copy each.)
            k = w[:, :, r0:r0 + sh * (Th - 1) + 1:sh, s0:s0 + sw * (Tw - 1) + 1:sw].flip(2, 3)
            k = k.transpose(0, 1).contiguous()                                  # [Cin, Cout, Th, Tw]
            out = _conv_fwd(dz, k, None, (1, 1), (pad_h, pad_w), out_hw=(qa, qb), precision=precision)
            if two:
                classes[(a, b)] = out
            else:
                dx[:, :, a::sh, b::sw] = out
    if two:
        if not classes:
            dx = dz.new_zeros(N, Cin, H, W)
            return dx if add is None else dx + add
        return _interleave2(classes, add, N, Cin, H, W)
    return dx if add is None else dx + add


def _relu_mask(dy, y, out=None):
    """dy * (y > 0) (fi_relu_mask); out may be dy itself."""
    out = torch.empty_like(dy) if out is None else out
    with torch.cuda.device(dy.device):
        _lib.check(_lib.load().fi_relu_mask(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(out), dy.numel(), _lib.current_stream()),
                   "fi_relu_mask")
    return out


class _Upsample2xFn(torch.autograd.Function):
    """F.interpolate(x, scale_factor=2, mode='nearest') whose backward is ONE pass (fi_sum2x2: the framework's kernel
    takes 0.5 ms per step on the FPN's three top-down maps, 5x the time of reading them)."""

    @staticmethod
    def forward(ctx, x):
        return F.interpolate(x, scale_factor=2, mode='nearest')

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous().float()
        N, C, H2, W2 = dy.shape
        if not dy.is_cuda or (W2 // 2) % 2 or dy.data_ptr() % 16:
            return dy.view(N, C, H2 // 2, 2, W2 // 2, 2).sum((3, 5))
        out = torch.empty((N, C, H2 // 2, W2 // 2), device=dy.device, dtype=torch.float32)
        with torch.cuda.device(dy.device):
            _lib.check(_lib.load().fi_sum2x2(_lib.ptr(dy), _lib.ptr(out), N * C, H2 // 2, W2 // 2, _lib.current_stream()),
                       "fi_sum2x2")
        return out


class _MaxPool3x3s2Fn(torch.autograd.Function):
    """The stem's max-pooling (3 x 3, stride 2, windows clipped at the border) without an index tensor: backward
    recomputes the arg-max from the input, which the stem keeps anyway (fi_maxpool3x3s2_*).  masked: the input is a
    ReLU output whose Gate this op claimed -- the gradient leaves multiplied by (x > 0)."""

    @staticmethod
    def forward(ctx, x, masked):
        N, C, H, W = x.shape
        y = torch.empty((N, C, (H - 2) // 2 + 1, (W - 2) // 2 + 1), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().fi_maxpool3x3s2_forward(_lib.ptr(x), _lib.ptr(y), N * C, H, W, _lib.current_stream()),
                       "fi_maxpool3x3s2_forward")
        ctx.save_for_backward(x)
        ctx.masked = bool(masked)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        N, C, H, W = x.shape
        dy = dy.contiguous().float()
        dx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().fi_maxpool3x3s2_backward(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(dx), N * C, H, W,
                                                            1 if ctx.masked else 0, _lib.current_stream()),
                       "fi_maxpool3x3s2_backward")
        return dx, None


def maxpool3x3s2(x):
    """max_pool2d(x, 3, 2, ceil_mode=True) for a contiguous fp32 CUDA map whose width is a multiple of 4 (None if x
    is something else: the caller falls back to the framework).  Claims x's Gate (the pooling is the stem's only
    reader)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and x.shape[3] % 4 == 0 and
            x.shape[2] >= 3 and x.data_ptr() % 16 == 0):
        return None
    return _MaxPool3x3s2Fn.apply(x, _claim_gate(x, True))


class _Conv1x1ClassRowsFn(torch.autograd.Function):
    """out[n] = conv1x1(x, W, b)[n, cls[n]]  ([N, H, W] from x [N, C, H, W]): the 1x1 convolution is computed for ALL
    output channels (the reference's schedule: lib/sub_module.py:783-786 evaluates every class's mask), the caller reads
    one of them per row.  Backward exploits what that implies -- the output gradient has one non-zero channel per row --
    instead of running the dense data- and weight-gradient kernels over K - 1 channels of zeros
    (fi_class_row_conv1x1_backward: same gradients, one pass over x and one over dx)."""

    @staticmethod
    def forward(ctx, x, w, b, cls, gated):
        _lib.require_cuda(x, w)
        x = x.contiguous().float()
        w2 = w.reshape(w.shape[0], -1).contiguous().float()
        y = _conv_fwd(x, w2.view(w.shape[0], x.shape[1], 1, 1), b.contiguous().float() if b is not None else None,
                      (1, 1), (0, 0))
        cls = cls.to(torch.int64).contiguous()
        ctx.save_for_backward(x, w2, cls)
        ctx.gated, ctx.has_bias, ctx.wshape = bool(gated), b is not None, tuple(w.shape)
        return y[torch.arange(x.shape[0], device=x.device), cls]

    @staticmethod
    def backward(ctx, d):
        x, w2, cls = ctx.saved_tensors
        N, C, H, W = x.shape
        K = w2.shape[0]
        d = d.contiguous().float()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.zeros_like(w2) if ctx.needs_input_grad[1] else None
        db = torch.zeros(K, device=x.device, dtype=torch.float32) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        L = _lib.load()
        ws = torch.empty(int(L.fi_class_row_conv1x1_workspace_bytes(N, C)) // 4, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(L.fi_class_row_conv1x1_backward(_lib.ptr(d), _lib.ptr(x), _lib.ptr(w2), _lib.ptr(cls),
                                                       _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), N, C, H * W, K,
                                                       1 if ctx.gated else 0, _lib.ptr(ws), _lib.current_stream()),
                       "fi_class_row_conv1x1_backward")
        return dx, (dw.view(ctx.wshape) if dw is not None else None), db, None, None


def conv1x1_class_rows(x, weight, bias, cls, gate_dx=False):
    """conv2d(x, weight [K, C, 1, 1], bias)[n, cls[n]] for every row n -> [N, H, W]; gate_dx as conv2d's."""
    if weight.shape[0] > 240:          # the backward keeps a [K][64] table in LDS: dense path beyond that
        return conv2d(x, weight, bias, gate_dx=gate_dx)[torch.arange(x.shape[0], device=x.device), cls.long()]
    return _Conv1x1ClassRowsFn.apply(x, weight, bias, cls, _claim_gate(x, gate_dx))


class _TakeRowsFn(torch.autograd.Function):
    """x[index] along dim 0 for a tensor with a second reader: backward adds dy's rows INTO the gradient that reader
    left in the GradBox (one index_add) instead of zeros + index_put followed by autograd's accumulation pass."""

    @staticmethod
    def _fast(x, index):
        n = x[0].numel() if x.shape[0] else 0
        return (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and index.dtype == torch.int64 and
                index.is_contiguous() and n >= 4 and n % 4 == 0 and 0 < index.numel() <= 65535 and x.data_ptr() % 16 == 0)

    @staticmethod
    def forward(ctx, x, index, box):
        ctx.save_for_backward(index)
        ctx.shape, ctx.box = tuple(x.shape), box
        if not _TakeRowsFn._fast(x, index):
            return x.index_select(0, index)
        out = torch.empty((index.numel(),) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().fi_rows_gather(_lib.ptr(x), _lib.ptr(index), _lib.ptr(out), index.numel(),
                                                  x[0].numel(), _lib.current_stream()), "fi_rows_gather")
        return out

    @staticmethod
    def backward(ctx, dy):
        index, = ctx.saved_tensors
        base = _take_boxes(ctx.box)
        dy = dy.contiguous()
        if (base is None or (torch.is_tensor(base) and base.dim() == len(ctx.shape) and base.shape[0] < ctx.shape[0] and
                             tuple(base.shape[1:]) == ctx.shape[1:] and base.is_contiguous() and
                             base.dtype == torch.float32)) and \
                dy.is_cuda and dy.dtype == torch.float32 and index.dtype == torch.int64 and 0 < ctx.shape[0] <= 65535 and \
                dy[0].numel() % 4 == 0 and dy.data_ptr() % 16 == 0 and (base is None or base.data_ptr() % 16 == 0):
            # the other reader took the FIRST rows of x as a view (the mask head's graph batch) or nothing at all: the
            # whole gradient in one pass (fi_rows_combine)
            rows = ctx.shape[0]
            pos = torch.full((rows,), -1, device=dy.device, dtype=torch.int64)
            pos[index] = torch.arange(index.numel(), device=dy.device, dtype=torch.int64)
            out = torch.empty(ctx.shape, device=dy.device, dtype=torch.float32)
            with torch.cuda.device(dy.device):
                _lib.check(_lib.load().fi_rows_combine(_lib.ptr(base), 0 if base is None else base.shape[0], _lib.ptr(dy),
                                                       _lib.ptr(pos), _lib.ptr(out), rows, dy[0].numel(),
                                                       _lib.current_stream()), "fi_rows_combine")
            return out, None, None
        if not (torch.is_tensor(base) and tuple(base.shape) == ctx.shape and base.is_contiguous() and
                base.dtype == dy.dtype):
            extra, base = base, torch.zeros(ctx.shape, device=dy.device, dtype=dy.dtype)
            if torch.is_tensor(extra):
                base[:extra.shape[0]] += extra
        if _TakeRowsFn._fast(base, index) and dy.dtype == torch.float32 and dy.data_ptr() % 16 == 0:
            # the index is a prefix of a permutation (distinct rows): plain read-modify-write
            with torch.cuda.device(dy.device):
                _lib.check(_lib.load().fi_rows_scatter_add(_lib.ptr(dy), _lib.ptr(index), _lib.ptr(base), index.numel(),
                                                           base[0].numel(), _lib.current_stream()), "fi_rows_scatter_add")
        else:
            base.index_add_(0, index, dy)
        return base, None, None


def take_rows(x, index, grad_box=None):
    """x.index_select(0, index) for an index vector with DISTINCT entries; grad_box: a GradBox in which x's OTHER reader (applied later in forward) leaves its
    gradient for x -- this op becomes the box's taker and returns the sum to autograd."""
    if grad_box is not None and GATES and x.is_cuda and x.requires_grad and torch.is_grad_enabled():
        grad_box.taker = True
        return _TakeRowsFn.apply(x, index, grad_box)
    return x.index_select(0, index)


def upsample2x(x):
    """x2 nearest-neighbour upsampling (the FPN's top-down path)."""
    return _Upsample2xFn.apply(x) if (x.is_cuda and x.requires_grad and torch.is_grad_enabled()) else \
        F.interpolate(x, scale_factor=2, mode='nearest')


def _conv_backward(ctx_needs, x, w, dz, stride, padding, want_db=False, add_to_dx=None, precision=None,
                   give_compact=False, bias_ptr=0, gate=None, w_scale=None, db_into=None, after_wgrad=None):
    """dX and dW of z = conv(x, w) given dz (shared by the plain and the fused functions).
    want_db: also return sum(dz) over images and pixels (the bias gradient), accumulated by the
    weight-gradient kernel from the dY tiles it stages anyway.
    add_to_dx: a tensor shaped like x -- or a _Compact -- that is added to dX inside the data-gradient kernel's
    epilogue (the shortcut gradient of a bottleneck, instead of a separate add pass).
    give_compact: a 1x1 / stride-2 layer may return dX as a _Compact (see there) instead of the full tensor.
    gate: a tensor shaped like x; dX (with add_to_dx) is multiplied by (gate > 0) -- inside the data-gradient kernel's
    epilogue where that kernel has one, by a separate pass otherwise.
    w_scale [Cout]: the data gradient uses W * w_scale[co] (conv + eval-BatchNorm given the UNSCALED masked gradient:
    _ConvBnActFn.backward); the weight gradient is the caller's to scale (after_wgrad).
    db_into: where sum(dz) is accumulated (a zeroed [Cout] tensor) instead of the bias's arena slot.
    after_wgrad(dw_flat, tap_major): called right after the weight-gradient launch, on the stream it ran on."""
    L = _lib.load()
    N, Cin, H, W = x.shape
    Cout, _, R, S = w.shape
    dx = dw = db = None
    dz_ready = None
    if ctx_needs[0] and ctx_needs[1] and WGRAD_SIDE_STREAM_MAX_PIXELS and \
            N * dz.shape[2] * dz.shape[3] <= WGRAD_SIDE_STREAM_MAX_PIXELS and precision not in _LOWP:
        # the weight gradient may run on the second stream (below): it depends on dz as it is NOW, not on the data
        # gradient that is enqueued first
        dz_ready = torch.cuda.Event()
        dz_ready.record(torch.cuda.current_stream(x.device))
    if ctx_needs[0]:
        compact_path = stride == (2, 2) and R * S == 1 and padding == (0, 0) and Cout % 16 == 0 and Cin % 16 == 0
        if isinstance(add_to_dx, tuple):
            # (compact or None, full tensor or None): two gradients handed over (the projection shortcut's and the FPN
            # lateral's); only the compact 1x1 / stride-2 path takes both inside its kernels
            cpt, full = add_to_dx
            if compact_path:
                add_to_dx = (cpt, full) if (cpt is not None and full is not None) else (cpt if cpt is not None else full)
            else:
                parts = [t.expand() if isinstance(t, _Compact) else t for t in (cpt, full) if t is not None]
                add_to_dx = parts[0] + parts[1] if len(parts) == 2 else (parts[0] if parts else None)
        if isinstance(add_to_dx, _Compact) and not compact_path:
            add_to_dx = add_to_dx.expand()
        scaled = w_scale is not None
        weff = None            # W * w_scale, made on demand where no cached transposed copy exists

        def w_eff():
            return w * w_scale.view(-1, 1, 1, 1) if scaled else w
        gate_in_kernel = gate is not None and gate.is_contiguous()
        if stride == (1, 1):
            if Cout % 16 == 0 and R * S <= 64:
                # transposed weight in the kernel's tap-major layout [Cin, R, S, Cout]: re-laid-out for all
                # layers by one launch per step (prepare_step); otherwise one copy here.  The tap flip is
                # done by the kernel's weight indexing (weight_layout 2)
                wt = _cached_wt(w, scaled)
                if wt is None:
                    wt = w_eff().permute(1, 2, 3, 0).contiguous()
                dx = _conv_fwd(dz, wt, None, (1, 1), (R - 1 - padding[0], S - 1 - padding[1]), w_tap_major=True,
                               flip_taps=True, residual=add_to_dx, precision=precision,
                               gate=gate if gate_in_kernel else None)
                add_to_dx = None
                if gate_in_kernel:
                    gate = None
            else:
                wt = w_eff().flip(2, 3).transpose(0, 1).contiguous()          # [Cin, Cout, R, S]
                fuse = gate_in_kernel and (add_to_dx is None or torch.is_tensor(add_to_dx))
                dx = _conv_fwd(dz, wt, None, (1, 1), (R - 1 - padding[0], S - 1 - padding[1]), precision=precision,
                               residual=add_to_dx if fuse else None, gate=gate if fuse else None)
                if fuse:
                    add_to_dx = gate = None
            if add_to_dx is not None:
                dx = dx + add_to_dx
        elif stride == (2, 2) and R * S == 1 and padding == (0, 0) and Cout % 16 == 0 and Cin % 16 == 0:
            # 1x1 / stride 2 (the first block of C3..C5: conv1 and the projection shortcut): only the even input
            # positions receive a gradient -- ONE 1x1 correlation with the cached W^T on the half-size map,
            # a second compact gradient for the same input added in its epilogue, then one interleave pass
            wt = _cached_wt(w, scaled)
            if wt is None:
                wt = w_eff().permute(1, 2, 3, 0).contiguous()
            both = add_to_dx if isinstance(add_to_dx, tuple) else (add_to_dx if isinstance(add_to_dx, _Compact) else None,
                                                                    add_to_dx if torch.is_tensor(add_to_dx) else None)
            comp = both[0].t if both[0] is not None else None
            c = _conv_fwd(dz, wt, None, (1, 1), (0, 0), w_tap_major=True, flip_taps=True, residual=comp,
                          precision=precision)
            full_add = both[1]
            if give_compact and full_add is None and gate is None:
                dx = _Compact(c, (H, W))
            else:
                dx = _interleave2({(0, 0): c}, full_add, N, Cin, H, W,
                                  gate=gate if (gate is not None and gate.is_contiguous()) else None)
                if gate is not None and gate.is_contiguous():
                    gate = None
        else:
            dx = _strided_dgrad(dz, w_eff(), (H, W), stride, padding, precision, add=add_to_dx)
        if gate is not None:
            if isinstance(dx, _Compact):
                dx = dx.expand()
            dx = _relu_mask(dx, gate.contiguous(), dx)
    if ctx_needs[1]:
        # tap-major dW ([Cout,R,S,Cin]): 128 channels of one tap per column tile, or -- same-size stride-1
        # layers with Cin == 64 (the C2 stage) -- 64 channels of two taps (mirrors wgrad_same_size())
        same = (stride == (1, 1) and dz.shape[2] == H and dz.shape[3] == W and (H * W) % 4 == 0 and W >= 4 and
                x.numel() * 4 < 0x7fffff00 and dz.numel() * 4 < 0x7fffff00 and
                x.data_ptr() % 16 == 0 and dz.data_ptr() % 16 == 0)
        hwc = 1 if (Cin % 128 == 0 or (Cin == 64 and same)) else 0
        # bf16 weight gradient: always tap-major, so the parameter must be stored that way (or be 1x1)
        bf16 = precision in _LOWP and (R * S == 1 or (Cin % 16 == 0 and w.is_contiguous(memory_format=torch.channels_last)))
        if bf16:
            hwc = 1
        shape = (Cout, R, S, Cin) if (hwc and R * S > 1) else (Cout, Cin, R, S)
        # pre-zeroed slice of the step's gradient arena (one fill per step instead of one per layer); a repeated
        # use of the layer accumulates into the same slice.  (With want_db the kernel clears dW and db itself.)
        # With want_db the fp32 kernel accumulates the bias gradient too: both outputs must be pre-zeroed slots, or
        # the kernel clears both itself.
        db_slot = None
        if want_db:
            if db_into is not None:
                db_slot = db_into
            else:
                db_slot, _ = _arena_take(("db", bias_ptr), Cout) if bias_ptr else (None, False)
        if want_db and db_slot is None:
            dw, first = None, False
        else:
            dw, first = _arena_take(("dw", w.data_ptr()), Cout * Cin * R * S)
            if dw is None:
                db_slot = None
            elif after_wgrad is not None and not first:
                # a repeated use of the layer would add unscaled sums onto the scaled ones of the first use:
                # _ConvBnActFn.backward routes repeated uses to the path that has no after_wgrad
                raise _lib.FiError("after_wgrad on a layer applied more than once per step")
        flags = _lib.OUTPUTS_ZEROED if dw is not None else 0
        dw = dw.view(shape) if dw is not None else torch.empty(shape, device=x.device, dtype=torch.float32)
        hand_over = first or not flags
        if want_db:
            db = db_slot if db_slot is not None else torch.empty(Cout, device=x.device, dtype=torch.float32)
        side = None
        # second stream only if autograd will ADOPT dw as the parameter's gradient: a dw whose memory order differs
        # from the parameter's is cloned by AccumulateGrad, on the main stream, while the kernel may still be running
        adopt = R * S == 1 or bool(hwc) == (not w.is_contiguous())
        if flags and dz_ready is not None and not bf16 and adopt:
            main = torch.cuda.current_stream(x.device)
            side = _wgrad_side_stream(x.device)
            side.wait_event(dz_ready)                      # dz (and x) were complete on the main stream there
            x.record_stream(side)
            dz.record_stream(side)
        if flags and not first and _WGQ["queues"]:
            # a further use of a layer in this backward pass (two forward passes before one backward; the dense RPN on five
            # levels): its first use may still be queued together with the pass that scales dW in place -- run the queue
            # before this use adds to the slot, and do not queue this one
            flush_deferred_wgrads()
        if flags and first and adopt and WGRAD_BATCH > 1 and hwc and same and Cin % 128 == 0 and \
                (not bf16 or (Cout % 64 == 0 and (R * S == 1 or (R, S, padding) == (3, 3, (1, 1))))) and \
                N * dz.shape[2] * dz.shape[3] <= WGRAD_BATCH_MAX_PIXELS and _defer_wgrad(
                    (N, Cin, H, W, Cout, R, S, padding, want_db and db is not None, precision if bf16 else None),
                    x, dz, dw, db if want_db else None, after_wgrad, bool(hwc and R * S > 1), dz_ready,
                    torch.cuda.current_stream(x.device), side):
            side = None                    # queued: launched with the other layers of its geometry (_flush_wgrads)
            deferred = True
        else:
            deferred = False
        with torch.cuda.device(x.device), (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            if deferred:
                pass
            elif bf16:
                _log_flops("bf16_wgrad", Cout, R, S, 2 * N * Cout * dz.shape[2] * dz.shape[3] * Cin * R * S)
                _lib.check(_lowp_fn(L, "conv2d_weight_grad_db", precision)(_lib.ptr(x), _lib.ptr(dz), _lib.ptr(dw), _lib.ptr(db),
                                                        N, Cin, H, W, Cout,
                                                        R, S, stride[0], stride[1], padding[0], padding[1], flags,
                                                        _lib.current_stream()), "fi_conv2d_weight_grad_bf16")
            else:
                _log_flops("wgrad", Cout, R, S, 2 * N * Cout * dz.shape[2] * dz.shape[3] * Cin * R * S,
                           N * dz.shape[2] * dz.shape[3], Cin)
                _lib.check(L.fi_conv2d_weight_grad(_lib.ptr(x), _lib.ptr(dz), _lib.ptr(dw), N, Cin, H, W, Cout,
                                                   R, S, stride[0], stride[1], padding[0], padding[1], hwc,
                                                   _lib.ptr(db), flags, _lib.current_stream()), "fi_conv2d_weight_grad")
            if after_wgrad is not None and not deferred:
                after_wgrad(dw, db, bool(hwc and R * S > 1))        # (the 16-bit kernels write tap-major: hwc is set)
        if side is not None:
            _queue_wgrad_join(main, side)
        _age_wgrad_queues()
        if hwc and R * S > 1:
            dw = dw.permute(0, 3, 1, 2)
        if not hand_over:
            dw = None                 # accumulated into the slot autograd already holds
            if want_db:
                db = None
    elif want_db:
        db = dz.sum((0, 2, 3))
    return (dx, dw, db) if want_db else (dx, dw)


# ---- weight gradients of small layers on a second stream ---------------------------------------------------------
# The data gradient and the weight gradient of a layer are independent (both read dz).  For layers whose grids are a
# single round of workgroups (the C3..C5 stages at batch 4) each kernel leaves the chip part-empty while its
# workgroups start up and while they drain (prologue, short K loops, atomic epilogue); issued on two streams the
# tail of one kernel overlaps the head of the other.  Only for gradients that land in the persistent arena (no
# allocator involvement); the main stream re-joins at the end of the backward pass (an autograd engine callback) and
# data_parallel.GradientBuckets waits for this stream before it reduces a bucket.  Measured in one box (scripts/ab_env.sh):
# fp32 step 157.9 -> 155.1 ms with every layer on the second stream (155.3 with only layers <= 65 536 pixels); the
# 16-bit kernels are bound by operand traffic through L2, not by their tails -- there it costs 0.5-1 ms and stays off.
import os as _os
# Round 4: OFF by default.  With the weight gradients of a stage's identical layers batched into few long launches
# (WGRAD_BATCH below) their tails no longer need hiding, and two MFMA-bound kernels sharing the chip only slow each other:
# same-box A/B 112.7 (second stream) vs 110.9 ms/step (main stream), profiles/r04_ab_wgrad_stream.txt.
WGRAD_SIDE_STREAM_MAX_PIXELS = int(_os.environ.get("FI_WGRAD_SIDE_PIXELS", "0"))   # largest layer (pixels) that takes the second stream; 0 disables
_WG_STREAM = {}
def wgrad_stream(device):
    """The second stream weight gradients may run on (None before first use)."""
    return _WG_STREAM.get(_lib.device_key(device))


def _wgrad_side_stream(dev):
    key = _lib.device_key(dev)
    st = _WG_STREAM.get(key)
    if st is None:
        st = _WG_STREAM[key] = torch.cuda.Stream(device=dev)
    return st


def _queue_wgrad_join(main, side):
    """Make `main` (the stream the layer's backward runs on) and the stream backward() was called from wait for the
    weight gradients on `side` when the running backward pass ends.  One engine callback per launch: a stream wait is
    a few microseconds of host time, and no state survives a backward pass that raises."""
    def join():
        main.wait_stream(side)
        torch.cuda.current_stream(side.device).wait_stream(side)
    try:
        torch.autograd.Variable._execution_engine.queue_callback(join)
    except RuntimeError:          # not inside a backward pass: join right away
        join()


# ---- weight gradients of identical layers in ONE launch ---------------------------------------------------------------
# A layer of the C4 stage at batch 4 ([4, 256|1024, 64, 64]) gives the weight-gradient kernel 16..36 tiles: to fill the
# chip it is cut into up to 64 pixel splits, one round of short workgroups whose fixed cost -- prologue, atomic epilogue of
# 64 KB per workgroup -- is a third of the launch (86..95 us for 8.6 GFLOP whatever the shape; scripts/wg_batch_probe.py:
# the same arithmetic with 4x / 23x the pixels per launch runs at 123 / 130 TFLOP/s instead of 90..98).  ResNet-101 has 23
# such blocks in a row.  Their weight gradients already run on the second stream and nothing reads them before the
# optimiser, so they are QUEUED per geometry and launched WGRAD_BATCH at a time (fi_conv2d_weight_grad_batch: the operand
# pointers travel in the kernel arguments), each problem with fewer, longer splits.  A queue is also flushed when its
# geometry has not been seen for WGRAD_BATCH_AGE convolution backward calls (the stage is over), when a data-parallel
# bucket is about to be reduced (data_parallel.GradientBuckets), and at the end of the backward pass.
WGRAD_BATCH = int(_os.environ.get("FI_WGRAD_BATCH", "12"))              # problems per launch; <= 1 disables the queue
WGRAD_BATCH_AGE = 9
WGRAD_BATCH_MAX_PIXELS = 65536                                         # larger layers fill the chip on their own
_WGQ = {"queues": {}, "tick": 0, "armed": False, "streams": {}}


def _defer_wgrad(key, x, dz, dw, db, after, tap_major, ev, main, side):
    """Queue one weight gradient (True), or decline (False: not inside a backward pass -- nobody would flush)."""
    q = _WGQ
    if not q["armed"]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_finish_wgrads)
        except RuntimeError:
            return False
        q["armed"] = True
    if side is not None:
        q["streams"][(id(main), id(side))] = (main, side)
    e = q["queues"].setdefault(key, {"items": [], "last": 0, "main": main, "side": side})
    # (an alias of dw: AccumulateGrad adopts a gradient only while nobody else holds the tensor it was handed -- a second
    # reference to that very object makes it clone the still-empty slot)
    e["items"].append((x, dz, dw.view(dw.shape), None if db is None else db.view(db.shape), after, tap_major, ev))
    e["last"] = q["tick"]
    if len(e["items"]) >= min(WGRAD_BATCH, 24):
        _flush_wgrads(key)
    return True


def _age_wgrad_queues():
    q = _WGQ
    q["tick"] += 1
    if q["queues"]:
        for key in [k for k, e in q["queues"].items() if q["tick"] - e["last"] >= WGRAD_BATCH_AGE]:
            _flush_wgrads(key)


def _flush_wgrads(key):
    import ctypes
    e = _WGQ["queues"].pop(key, None)
    if not e or not e["items"]:
        return
    items, side = e["items"], e["side"]
    N, Cin, H, W, Cout, R, S, padding, has_db, lowp = key
    n = len(items)
    L = _lib.load()
    arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    xs, dzs, dws = [it[0] for it in items], [it[1] for it in items], [it[2] for it in items]
    if side is not None:                   # (None: no second stream for weight gradients -- launched where we are)
        side.wait_event(items[-1][6])      # the events were recorded on one stream, in order: the last covers all
        for t in xs + dzs:
            t.record_stream(side)
    with torch.cuda.device(xs[0].device), (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        if lowp:
            _log_flops("bf16_wgrad", Cout, R, S, 2.0 * n * N * H * W * Cout * Cin * R * S)
        else:
            _log_flops("wgrad", Cout, R, S, 2.0 * n * N * H * W * Cout * Cin * R * S, N * H * W, Cin, batch=n)
        fn = _lowp_fn(L, "conv2d_weight_grad_batch", lowp) if lowp else L.fi_conv2d_weight_grad_batch
        _lib.check(fn(arr(xs), arr(dzs), arr(dws),
                                                 arr([it[3] for it in items]) if has_db else None, n, N, Cin, H, W, Cout,
                                                 R, S, 1, 1, padding[0], padding[1], 1, _lib.OUTPUTS_ZEROED,
                                                 _lib.current_stream()), "fi_conv2d_weight_grad_batch")
        folds = [getattr(it[4], "fold", None) for it in items]
        if n > 1 and all(f is not None for f in folds) and len({(f[4], f[5] is None, f[6] is None, f[7] is None,
                                                                 f[0].is_contiguous()) for f in folds}) == 1:
            # every layer of the batch is conv + eval-BatchNorm: their fi_bn_fold_grad passes as one launch too
            f0 = folds[0]
            w_tap_major = R * S > 1 and not f0[0].is_contiguous()
            here = torch.cuda.current_stream(xs[0].device)
            for f in folds:
                f[8].record_stream(here)
                f[1].record_stream(here)
            col = lambda k: arr([f[k] for f in folds]) if f0[k] is not None else None
            _lib.check(L.fi_bn_fold_grad_batch(arr(dws), col(0), arr([it[3] for it in items]), col(1), col(2), col(3), f0[4],
                                               col(5), col(6), col(7), n, Cout, Cin, R * S, 1 if items[0][5] else 0,
                                               1 if w_tap_major else 0, _lib.current_stream()), "fi_bn_fold_grad_batch")
        else:
            for x, dz, dw, db, after, tap_major, ev in items:
                if after is not None:
                    after(dw, db, tap_major)


def flush_deferred_wgrads():
    """Launch every queued weight gradient now (on the weight-gradient stream)."""
    for key in list(_WGQ["queues"].keys()):
        _flush_wgrads(key)


def _finish_wgrads():
    """Engine callback at the end of the backward pass that queued something: flush, then re-join the streams."""
    q = _WGQ
    q["armed"] = False
    flush_deferred_wgrads()
    joins, q["streams"] = list(q["streams"].values()), {}
    for main, side in joins:
        main.wait_stream(side)
        torch.cuda.current_stream(side.device).wait_stream(side)


# ---- per-step derived state: W^T for the data gradient, zeroed gradient arena ---------------------
_WT = {}           # weight data_ptr -> (W^T [Cin,R,S,Cout], weight version it was made from)
_WF = {}           # data_ptr of a 1x1 weight (or of its W^T) -> (fragment-major copy, version or None, (Cout, Cin) of the matrix,
                   # weak reference to the tensor that owns the address).  An entry is valid only while its owner lives: a
                   # model that is garbage-collected leaves its entries behind (the plan is weakly keyed), and the allocator
                   # hands the freed addresses to the next model's tensors -- same shape, same version counter
_ARENA = {"buf": None, "slots": {}, "used": set()}
_PLAN = weakref.WeakKeyDictionary()      # model -> cached layer lists / descriptor table


_WB = {}           # tap-major fp32 weight data_ptr -> (bf16 copy, version, shape): refreshed once per step


def _cached_bf16(w, dtype=torch.bfloat16):
    """16-bit copy of a (contiguous, tap-major) fp32 weight, made once per (tensor, version).  Model weights and
    their W^T are converted for ALL layers by one multi-tensor copy per step (_prepare_step fills the cache); anything
    else (temporaries) is converted here.  The entry keeps `w` alive, so its address cannot be handed to another
    tensor while the entry exists."""
    e = _WB.get(w.data_ptr())
    if e is not None and e[1] == w._version and e[0].numel() == w.numel() and e[0].dtype == dtype:
        return e[0].view(w.shape) if e[0].shape != w.shape else e[0]
    wb = w.to(dtype)
    _WB[w.data_ptr()] = (wb, w._version, tuple(w.shape), w)
    return wb


def _cached_wt(w, scaled=False):
    """This step's W^T [Cin][R][S][Cout] (prepare_step); scaled: the copy with the layer's eval-BatchNorm scale folded
    in (W^T[...][co] * scale[co]) -- a layer has one or the other."""
    e = _WT.get(w.data_ptr())
    if e is not None and e[3]() is not None and e[1] == w._version and e[2] == bool(scaled) and \
            e[0].shape == (w.shape[1], w.shape[2], w.shape[3], w.shape[0]):
        return e[0]
    return None


def _arena_take(key, numel, partner=None):
    """(slot, first): the slot reserved for `key` in this step's arena (zeroed at the start of the step) and
    whether this is its first use in the step.  A layer applied several times per step (the RPN on 5 levels, the
    feature extractor on small and big boxes) ACCUMULATES all its gradient contributions in the one slot -- the
    kernels add with atomics anyway -- and hands the slot to autograd only the first time (later calls return no
    gradient for the parameter), so there is no per-use buffer, fill and add.  (None, False): no slot."""
    slot = _ARENA["slots"].get(key)
    if slot is None or _ARENA["buf"] is None or slot[1] != numel:
        return None, False
    owner = slot[2]()
    if owner is None or owner.data_ptr() != key[1]:
        return None, False        # the model this arena belongs to is gone and its address was reused
    if partner is not None and len(slot) > 3 and slot[3] != partner:
        return None, False        # a BatchNorm block laid out for another convolution's bias: private buffers instead
    first = key not in _ARENA["used"]
    _ARENA["used"].add(key)
    return _ARENA["buf"][slot[0]:slot[0] + numel], first


import numpy as _np
_TR_DESC = _np.dtype([("src", "<u8"), ("dst", "<u8"), ("rows", "<i4"), ("cols", "<i4"), ("taps", "<i4"), ("pad", "<i4"),
                      ("tile_base", "<i8"), ("row_scale", "<u8")])      # FiTransposeDesc


def invalidate_step_state():
    _WT.clear()
    _WF.clear()
    _WB.clear()
    _ARENA["buf"] = None
    _ARENA["slots"] = {}
    _ARENA["used"] = set()


def prepare_step(model):
    """See _prepare_step; the caller's grad mode decides whether a gradient arena is laid out."""
    _prepare_step(model, torch.is_grad_enabled())


@torch.no_grad()
def _prepare_step(model, grad_on):
    """Once per forward pass of `model` (MaskRCNN.forward calls it):
      * eval-BN folds for every (conv, bn) pair (refresh_bn_folds);
      * convolution weights with Cin % 16 == 0 are STORED channels-last ([Cout][R][S][Cin] in memory, same
        logical shape and state-dict contents), which is the forward kernel's layout and the layout the
        weight-gradient kernel writes -- no per-call re-layout, no layout-contract clone in AccumulateGrad;
      * W^T [Cin][R][S][Cout] of every stride-1 layer for the data-gradient kernel, all layers in ONE launch;
      * when gradients are enabled: one zero-filled arena with a slot per weight gradient and per BatchNorm's
        (d shift, d gamma, d conv-bias) sums, so that the ~300 per-layer fills of a backward pass become one.
    """
    refresh_bn_folds()
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        return
    if _WGQ["queues"] or _WGQ["armed"]:
        # leftovers of a backward pass that raised before its end-of-pass callback ran: their slots are about to be cleared
        _WGQ["queues"].clear()
        _WGQ["streams"].clear()
        _WGQ["armed"] = False
    if len(_WB) > 4096:
        # bf16 copies of weights that were temporaries (the deconv's reshaped weight, transposes made on the fly)
        # keep their source alive; inference loops never bump a parameter version, so bound the cache here
        _WB.clear()
    plan = _PLAN.get(model)
    ptrs = tuple(p.data_ptr() for p in model.parameters())
    if plan is None or plan["ptrs"] != ptrs or plan["layout"].superseded:
        if plan is not None:
            for k in plan.get("wf_keys", ()):          # the replaced plan's fragment-major copies
                _WF.pop(k, None)
        for cache in (_WT, _WF):                       # entries of models that no longer exist (W^T first: it owns the
                                                       # tensors the data-gradient entries of _WF are keyed by)
            for k in [k for k, e in cache.items() if e[3]() is None]:
                del cache[k]
        convs = [m for m in model.modules() if isinstance(m, Conv2d) and not m.full_window]
        for m in convs:
            w = m.weight
            if w.shape[1] % 16 == 0 and w.shape[2] * w.shape[3] > 1 and \
                    not w.is_contiguous(memory_format=torch.channels_last):
                w.data = w.data.contiguous(memory_format=torch.channels_last)
        tr = [m for m in convs if (tuple(m.stride) == (1, 1) or m.weight.shape[2] * m.weight.shape[3] == 1) and
              m.weight.shape[0] % 16 == 0 and
              m.weight.shape[1] % 16 == 0 and m.weight.shape[2] * m.weight.shape[3] <= 64 and m.weight.requires_grad]
        import numpy as np
        desc = np.zeros(len(tr), dtype=_TR_DESC)
        wts, base = [], 0
        for i, m in enumerate(tr):
            co, ci, r, s_ = m.weight.shape
            wt = torch.empty((ci, r, s_, co), device=dev, dtype=torch.float32)
            wts.append(wt)
            desc[i] = (m.weight.data_ptr(), wt.data_ptr(), co, ci, r * s_, 0, base, 0)
            base += r * s_ * ((co + 31) // 32) * ((ci + 31) // 32)
        # fragment-major copies of the 1x1 / stride-1 weights for the persistent 1x1 kernel (conv1x1_ring_kernel,
        # weight_layout 3): W itself for the forward pass, W^T (with the eval-BatchNorm scale, like the plain W^T) for the
        # data gradient -- extra descriptors of the SAME launch (flag 1 = fragment-major, flag 2 = no transpose)
        frag = []           # (module, "fwd" | "dgrad", tensor, index of the module in tr or -1)
        tr_index = {m: i for i, m in enumerate(tr)}
        for m in convs:
            co, ci, r, s_ = m.weight.shape
            if r * s_ != 1 or tuple(m.stride) != (1, 1) or tuple(m.padding) != (0, 0) or _PRECISION in _LOWP:
                continue
            if co % 128 == 0 and ci % 32 == 0 and ci >= 128:
                frag.append((m, "fwd", torch.empty(co * ci, device=dev, dtype=torch.float32), -1))
            if m in tr_index and ci % 128 == 0 and co % 32 == 0 and co >= 128:
                frag.append((m, "dgrad", torch.empty(co * ci, device=dev, dtype=torch.float32), tr_index[m]))
        if frag:
            extra = np.zeros(len(frag), dtype=_TR_DESC)
            for i, (m, kind, t, _) in enumerate(frag):
                co, ci = m.weight.shape[0], m.weight.shape[1]
                extra[i] = (m.weight.data_ptr(), t.data_ptr(), co, ci, 1, 3 if kind == "fwd" else 1, base, 0)
                base += ((co + 31) // 32) * ((ci + 31) // 32)
            desc = np.concatenate([desc, extra])
        table = torch.from_numpy(desc.view(np.uint8).copy()).to(dev) if len(desc) else None
        # gradient slots: the model-wide arena layout (grad_arena.py) -- every trainable parameter has one, in
        # bucket order; the kernels' keys are the weight's / the BatchNorm gamma's address
        layout = grad_arena.get_layout(model)
        slots = {}
        for m in convs:
            if m.weight.requires_grad and m.weight in layout.slot:
                slots[("dw", m.weight.data_ptr())] = layout.slot[m.weight] + (weakref.ref(m.weight),)
            if m.bias is not None and m.bias.requires_grad and m.bias in layout.slot:
                slots[("db", m.bias.data_ptr())] = layout.slot[m.bias] + (weakref.ref(m.bias),)
        for bn in [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]:
            if bn in layout.bn_slot:
                # (4th entry: the address of the convolution bias the layout paired with this BatchNorm -- by adjacency
                # among a module's children; a caller that combines the BatchNorm with ANOTHER convolution must not
                # have that one's bias gradient adopted as a view of this block)
                partner = layout.bn_partner.get(bn)
                slots[("bn", bn.weight.data_ptr())] = (layout.bn_slot[bn], 3 * bn.num_features, weakref.ref(bn.weight),
                                                       partner.data_ptr() if partner is not None else 0)
        plan = {"ptrs": tuple(p.data_ptr() for p in model.parameters()), "tr": tr, "wts": wts, "table": table, "frag": frag,
                "desc": desc, "tiles": base, "slots": slots, "layout": layout, "versions": None, "convs": convs,
                "scales": None}
        _PLAN[model] = plan
    # fp32 training: W^T of a layer that is followed by an eval-mode BatchNorm carries the BatchNorm's scale
    # (_ConvBnActFn.backward feeds the data-gradient kernel the unscaled masked gradient); the 16-bit kernels keep the
    # plain W^T.  The pairs are known after the first forward pass (conv_bn_act registers them).
    scales = [None] * len(plan["tr"])
    if grad_on and _FOLD_PAIRS and plan["table"] is not None:
        by_conv = {c: b for b, c in _FOLD_PAIRS.items()}
        for i, m in enumerate(plan["tr"]):
            b = by_conv.get(m)
            if b is not None and not b.training and getattr(b, "_fi_fold", None) is not None:
                scales[i] = b
    sig = tuple(0 if b is None else b._fi_fold[0].data_ptr() for b in scales)
    sig = sig + tuple(sig[ti] if ti >= 0 else 0 for _, _, _, ti in plan["frag"])
    if plan["table"] is not None and plan["scales"] != sig:
        desc = plan["desc"]
        desc["row_scale"] = sig
        plan["table"] = torch.from_numpy(desc.view(_np.uint8).copy()).to(dev)
        plan["scales"] = sig
        plan["versions"] = None
    versions = tuple(m.weight._version for m in plan["tr"]) + tuple(b._fi_fold[2] for b in scales if b is not None)
    versions = versions + tuple(m.weight._version for m, _, _, ti in plan["frag"] if ti < 0)
    if plan["table"] is not None and (plan["versions"] != versions or not all(
            _cached_wt(m.weight, sig[0] != 0) is not None for m in plan["tr"][:1])):
        L = _lib.load()
        with torch.cuda.device(dev):
            _lib.check(L.fi_weight_transpose_batch(_lib.ptr(plan["table"]), len(plan["desc"]), plan["tiles"],
                                                   _lib.current_stream()), "fi_weight_transpose_batch")
        for m, wt, sp in zip(plan["tr"], plan["wts"], sig):
            _WT[m.weight.data_ptr()] = (wt, m.weight._version, sp != 0, weakref.ref(m.weight))
        keys = []
        for m, kind, t, ti in plan["frag"]:
            # forward: keyed by the parameter; data gradient: keyed by the W^T tensor _conv_fwd is handed
            if kind == "fwd":
                keys.append(m.weight.data_ptr())
                _WF[keys[-1]] = (t, m.weight._version, tuple(m.weight.shape[:2]), weakref.ref(m.weight))
            else:
                keys.append(plan["wts"][ti].data_ptr())
                _WF[keys[-1]] = (t, None, (m.weight.shape[1], m.weight.shape[0]), weakref.ref(plan["wts"][ti]))
        plan["wf_keys"] = keys
        plan["versions"] = versions
        _WB.clear()          # the W^T tensors were rewritten in place (no version bump): drop their bf16 copies
    if _PRECISION in _LOWP:
        # 16-bit copies of every layer's weight and W^T for this step: one multi-tensor copy (per-layer .to() calls were
        # ~190 launches / 0.8 ms of the bf16 step)
        dtype = _LOWP[_PRECISION][1]
        lw = plan.get("lowp")
        if lw is None or lw["dtype"] != dtype:
            srcs = [m.weight for m in plan["convs"] if m.weight.shape[1] % 32 == 0] + list(plan["wts"])
            lw = plan["lowp"] = {"dtype": dtype, "srcs": srcs, "versions": None,
                                 "dsts": [torch.empty(t.numel(), device=dev, dtype=dtype) for t in srcs]}
        vers = tuple(t._version for t in lw["srcs"]) + (plan["versions"],)
        stale = lw["versions"] != vers or any(_WB.get(t.data_ptr(), (None,))[0] is not d
                                               for t, d in zip(lw["srcs"][:1], lw["dsts"][:1]))
        if stale and lw["srcs"]:
            # memory order of a channels-last parameter IS the tap-major [Cout][R][S][Cin] order the kernels read
            flat = [t.permute(0, 2, 3, 1).reshape(-1) if (t.dim() == 4 and not t.is_contiguous()) else t.reshape(-1)
                    for t in lw["srcs"]]
            torch._foreach_copy_(lw["dsts"], flat)
            for t, d in zip(lw["srcs"], lw["dsts"]):
                _WB[t.data_ptr()] = (d, t._version, tuple(t.shape), t)
            lw["versions"] = vers
    layout = plan["layout"]
    if grad_on and layout.total:
        buf = layout.buffer(dev)
        if layout.live_gradients(buf):
            # gradients of an earlier backward pass are still attached (accumulation without zero_grad): they live
            # in the arena, so it cannot be cleared -- this pass's kernels allocate their own outputs and autograd
            # adds them to the attached gradients
            _ARENA["buf"] = None
        else:
            _ARENA["buf"] = layout.zero(dev)         # ONE fill per step, same addresses every step
        _ARENA["slots"] = plan["slots"]
        _ARENA["used"] = set()
    else:
        _ARENA["buf"] = None


_UNSCALED_BACKWARD = not _os.environ.get("FI_BN_BWD_OLD")      # A/B switch (scripts/ab_env.sh)
# Gates and the FPN-lateral hand-off on / off.  Off + _UNSCALED_BACKWARD off = the round-2 form of the backward pass, kept
# as the differential reference of tests/test_gpu_detector.py::test_detector_gradients_agree_between_backward_forms
GATES = True


class GradBox(object):
    """Hands a gradient from one autograd node to another: a bottleneck's last convolution leaves the
    gradient of its identity shortcut here -- or its projection shortcut leaves its data gradient here -- and the
    block's first convolution, whose data gradient flows into the same tensor, adds it inside its kernel
    epilogue (see Bottleneck.forward).  The giver must run first in backward: it must be applied AFTER the
    taker in forward (autograd executes ready nodes in reverse order of creation)."""
    __slots__ = ("value", "taker")

    def __init__(self):
        self.value = None
        self.taker = False      # set by the layer that WILL pick the value up (a giver must not give otherwise)


def _take_boxes(boxes):
    """The value(s) left in a GradBox, or in a pair of them (the second gradient then becomes the second element of a
    (compact, full) tuple for _conv_backward); the boxes are emptied."""
    if boxes is None:
        return None
    if isinstance(boxes, GradBox):
        v, boxes.value = boxes.value, None
        return v
    vals = []
    for b in boxes:
        v, b.value = b.value, None
        if v is not None:
            vals.append(v)
    if not vals:
        return None
    if len(vals) == 1:
        return vals[0]
    cpt = [v for v in vals if isinstance(v, _Compact)]
    full = [v for v in vals if not isinstance(v, _Compact)]
    if len(cpt) > 1 or len(full) > 1:
        parts = [v.expand() if isinstance(v, _Compact) else v for v in vals]
        return parts[0] + parts[1]
    return (cpt[0] if cpt else None, full[0] if full else None)


class Gate(object):
    """Travels with the output y of a fused conv + BN + ReLU (attribute `_fi_gate` of the tensor).  A consumer that is
    told it is y's ONLY reader (conv_bn_act(gate_dx=True)) claims it: its data gradient -- together with whatever it
    adds through a GradBox -- leaves its kernel multiplied by (y > 0), and the producer's backward skips its own
    masking pass over (dy, y)."""
    __slots__ = ("claimed",)

    def __init__(self):
        self.claimed = False


class _ConvBnActFn(torch.autograd.Function):
    """y = act(BN_eval(conv(x)) [+ residual]) in ONE kernel launch.

    Backward, fp32 kernels (round 3): with g = dy * (y > 0) the UNSCALED masked gradient,
        dX  = conv_T(g, W * scale[co])            W^T with the scale folded in: one transpose launch per step
        dW' = g (x) patches(x), s = sum_p g       the weight-gradient kernel and its bias sums
        dW = scale * dW', d beta = s, d gamma = inv_std * (<W, dW'> + (bias - mean) * s), d bias = scale * s
                                                  (fi_bn_fold_grad: sum_p g * conv(x, W) = <W, dW'>)
    so no pass over the activations computes the BatchNorm sums, the shortcut gradient of a bottleneck IS g, and g
    itself arrives ready-made when the consumer of y claimed its Gate (otherwise: one fi_relu_mask pass).
    16-bit kernels, channels-last outputs and layers applied more than once per step keep the older form: one fused
    elementwise/reduction pass (fi_bn_act_backward) producing dz = g * scale and the sums, then dgrad / wgrad on dz."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, mean, var, eps, residual, relu, stride, padding, out_cl=False,
                fold=None, res_grad_to=None, dx_add_from=None, dx_give_to=None, dx_gate=False, out_gate=None, live=None):
        _lib.require_cuda(x, w)
        x = x.contiguous().float()
        w = _dense(w.float())
        if fold is not None:              # precomputed for the whole model by refresh_bn_folds()
            scale, shift = fold
        else:
            scale = gamma * torch.rsqrt(var + eps)
            shift = beta - mean * scale
            if b is not None:
                shift = shift + b * scale
        res = residual.contiguous().float() if residual is not None else None
        _log_shape(x, w, stride, padding)
        y = _conv_fwd(x, w, shift.contiguous(), stride, padding, relu=relu, scale=scale.contiguous(), residual=res,
                      out_channels_last=out_cl, live=live)
        ctx.out_cl = bool(out_cl)
        ctx.precision = _PRECISION
        ctx.res_grad_to, ctx.dx_add_from, ctx.dx_give_to = res_grad_to, dx_add_from, dx_give_to
        ctx.dx_gate, ctx.out_gate = bool(dx_gate), out_gate
        ctx.save_for_backward(x, w, y, scale, gamma, beta, res, b)
        ctx.conf = (tuple(stride), tuple(padding), b is not None, bool(relu), residual is not None, eps, mean, var)
        return y

    @staticmethod
    def _backward_unscaled(ctx, dy):
        """The fp32 form of the class docstring; None if this use of the layer has to take the older form."""
        x, w, y, scale, gamma, beta, res, b = ctx.saved_tensors
        stride, padding, has_bias, relu, has_res, eps, mean, var = ctx.conf
        N, C, OH, OW = y.shape
        if not ctx.needs_input_grad[1]:
            return None                          # frozen weights: no weight gradient to take the sums from
        sums, first = _arena_take(("bn", gamma.data_ptr()), 3 * C, partner=b.data_ptr() if has_bias else 0)
        if sums is not None and not first:
            return None                          # applied more than once this step: sums and dW accumulate
        L = _lib.load()
        dy = dy.contiguous().float()
        if sums is None:
            sums = torch.zeros(3 * C, device=y.device, dtype=torch.float32)
        premasked = ctx.out_gate is not None and ctx.out_gate.claimed
        g = _relu_mask(dy, y) if (relu and not premasked) else dy
        g_res = g if (has_res and ctx.needs_input_grad[8]) else None
        if g_res is not None and ctx.res_grad_to is not None:
            ctx.res_grad_to.value = g_res           # picked up by the block's first convolution
            g_res = None
        add = _take_boxes(ctx.dx_add_from)
        want_gamma, want_beta = ctx.needs_input_grad[3], ctx.needs_input_grad[4]
        want_db = has_bias and ctx.needs_input_grad[2]
        out = {}

        def finish(dw_flat, s, tap_major):
            Cin, R, S = w.shape[1], w.shape[2], w.shape[3]
            w_tap_major = R * S > 1 and not w.is_contiguous()       # _dense(): contiguous or channels-last
            out["s"] = s
            here = torch.cuda.current_stream(sums.device)                  # the weight gradient's stream
            sums.record_stream(here)
            scale.record_stream(here)       # folded in-layer (first step, no prepare_step): a temporary of the forward
            _lib.check(L.fi_bn_fold_grad(_lib.ptr(dw_flat), _lib.ptr(w), _lib.ptr(s), _lib.ptr(scale), _lib.ptr(mean),
                                         _lib.ptr(var), float(eps), _lib.ptr(b) if has_bias else None,
                                         _lib.ptr(sums[C:2 * C]) if want_gamma else None,
                                         _lib.ptr(sums[2 * C:]) if want_db else None, C, Cin, R * S,
                                         1 if tap_major else 0, 1 if w_tap_major else 0, _lib.current_stream()),
                       "fi_bn_fold_grad")
        # what a batched flush needs to run the folds of n queued layers as ONE launch (_flush_wgrads)
        finish.fold = (w, scale, mean, var, float(eps), b if has_bias else None, sums[C:2 * C] if want_gamma else None,
                       sums[2 * C:] if want_db else None, sums)
        dx, dw, _ = _conv_backward(ctx.needs_input_grad, x, w, g, stride, padding, want_db=True, add_to_dx=add,
                                   precision=ctx.precision, give_compact=ctx.dx_give_to is not None,
                                   gate=x if ctx.dx_gate else None, w_scale=scale, db_into=sums[:C],
                                   after_wgrad=finish)
        # the weight-gradient kernel's bias sums: what finish() was handed, or -- the launch is still queued -- db_into
        dbeta = out.get("s", sums[:C]) if want_beta else None
        if ctx.dx_give_to is not None and dx is not None:
            ctx.dx_give_to.value = dx               # picked up (and added) by the backward of the block's first conv
            dx = None
        return (dx, dw, sums[2 * C:] if want_db else None, sums[C:2 * C] if want_gamma else None,
                dbeta if want_beta else None, None, None, None, g_res) + (None,) * 11

    @staticmethod
    def backward(ctx, dy):
        if not ctx.out_cl and _UNSCALED_BACKWARD:
            r = _ConvBnActFn._backward_unscaled(ctx, dy)
            if r is not None:
                return r
        x, w, y, scale, gamma, beta, res, b = ctx.saved_tensors
        stride, padding, has_bias, relu, has_res, eps, mean, var = ctx.conf
        L = _lib.load()
        # channels-last output: the gradient comes back channels-last from the RoIAlign backward and is
        # transposed to NCHW inside the fused pass (dz feeds the NCHW dgrad / wgrad kernels)
        dy = dy.float().contiguous(memory_format=torch.channels_last) if ctx.out_cl else dy.contiguous().float()
        N, C, OH, OW = y.shape
        dz = torch.empty(y.shape, device=y.device, dtype=torch.float32)
        g_res = torch.empty_like(y) if (has_res and ctx.needs_input_grad[8]) else None
        # (d shift, d gamma, d conv-bias) adjacent: a slot of the step's zeroed arena, or ONE fill in the call
        sums, first = _arena_take(("bn", gamma.data_ptr()), 3 * C, partner=b.data_ptr() if has_bias else 0)
        flags = _lib.OUTPUTS_ZEROED if sums is not None else 0
        if sums is None:
            sums, first = torch.empty(3 * C, device=y.device, dtype=torch.float32), True
        dshift = sums[:C]
        dgamma = sums[C:2 * C] if ctx.needs_input_grad[3] else None
        want_db = has_bias and ctx.needs_input_grad[2]
        db = sums[2 * C:] if want_db else None
        with torch.cuda.device(y.device):
            _lib.check(L.fi_bn_act_backward(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(scale), _lib.ptr(gamma),
                                            _lib.ptr(beta), _lib.ptr(res), N, C, OH * OW, 1 if relu else 0,
                                            _lib.ptr(dz),
                                            _lib.ptr(g_res), _lib.ptr(dshift), _lib.ptr(dgamma), _lib.ptr(db),
                                            1 if ctx.out_cl else 0, flags,
                                            _lib.current_stream()), "fi_bn_act_backward")
        if g_res is not None and ctx.res_grad_to is not None:
            ctx.res_grad_to.value = g_res           # picked up by the block's first convolution
            g_res = None
        add = _take_boxes(ctx.dx_add_from)
        dx, dw = _conv_backward(ctx.needs_input_grad, x, w, dz, stride, padding, add_to_dx=add, precision=ctx.precision,
                                give_compact=ctx.dx_give_to is not None, gate=x if ctx.dx_gate else None)
        if ctx.dx_give_to is not None and dx is not None:
            ctx.dx_give_to.value = dx               # picked up (and added) by the backward of the block's first conv
            dx = None
        dbeta = dshift if ctx.needs_input_grad[4] else None
        if not first:                 # a repeated use of the layer: accumulated into the slices autograd already holds
            db = dgamma = dbeta = None
        return (dx, dw, db, dgamma, dbeta, None, None, None, g_res) + (None,) * 11


class _ConvBiasActFn(torch.autograd.Function):
    """y = relu(conv(x) + bias) in one launch (layers without a BatchNorm: the RPN's shared conv,
    lib/sub_module.py:256-259; the mask head's deconv, :779-780).  Backward reuses the fused
    elementwise pass with a unit scale: it yields the ReLU-masked gradient and the bias gradient
    together (instead of threshold_backward + a separate reduction)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, padding, out_gate=None, dx_gate=False, dx_add_from=None):
        _lib.require_cuda(x, w)
        x = x.contiguous().float()
        w = _dense(w.float())
        bc = b.contiguous().float() if b is not None else None
        _log_shape(x, w, stride, padding)
        y = _conv_fwd(x, w, bc, stride, padding, relu=True)
        ctx.save_for_backward(x, w, y)
        ctx.precision = _PRECISION
        ctx.out_gate, ctx.dx_gate, ctx.dx_add_from = out_gate, bool(dx_gate), dx_add_from
        ctx.conf = (tuple(stride), tuple(padding), b is not None)
        ctx.bias_ptr = b.data_ptr() if b is not None else 0
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, padding, has_bias = ctx.conf
        L = _lib.load()
        dy = dy.contiguous().float()
        add = _take_boxes(ctx.dx_add_from)          # another reader's gradient for x: added in the data-gradient epilogue
        if ctx.out_gate is not None and ctx.out_gate.claimed:
            # the only reader of y applied the ReLU mask in its data-gradient epilogue (Gate): dy IS the masked
            # gradient; the bias gradient comes out of the weight-gradient kernel's pass over it
            if has_bias and ctx.needs_input_grad[2] and ctx.needs_input_grad[1]:
                return _conv_backward(ctx.needs_input_grad, x, w, dy, stride, padding, want_db=True,
                                      precision=ctx.precision, bias_ptr=ctx.bias_ptr, add_to_dx=add,
                                      gate=x if ctx.dx_gate else None) + (None, None, None, None, None)
            dx, dw = _conv_backward(ctx.needs_input_grad, x, w, dy, stride, padding, precision=ctx.precision,
                                    add_to_dx=add, gate=x if ctx.dx_gate else None)
            db = dy.sum((0, 2, 3)) if (has_bias and ctx.needs_input_grad[2]) else None
            return dx, dw, db, None, None, None, None, None
        N, C, OH, OW = y.shape
        dz = torch.empty_like(y)
        ones = _ones(C, y.device)
        # the bias gradient accumulates in the bias's arena slot (a layer applied on 5 pyramid levels: one slot)
        dshift, first = _arena_take(("db", ctx.bias_ptr), C) if ctx.bias_ptr else (None, False)
        flags = _lib.OUTPUTS_ZEROED if dshift is not None else 0
        if dshift is None:
            dshift, first = torch.empty(C, device=y.device, dtype=torch.float32), True
        with torch.cuda.device(y.device):
            _lib.check(L.fi_bn_act_backward(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(ones), None, None, None, N, C, OH * OW, 1,
                                            _lib.ptr(dz), None, _lib.ptr(dshift), None, None, 0, flags,
                                            _lib.current_stream()), "fi_bn_act_backward")
        dx, dw = _conv_backward(ctx.needs_input_grad, x, w, dz, stride, padding, precision=ctx.precision,
                                add_to_dx=add, gate=x if ctx.dx_gate else None)
        return dx, dw, (dshift if (has_bias and ctx.needs_input_grad[2] and first) else None), None, None, None, None, None


_ONES = {}


def _ones(n, device):
    """Shared, read-only (see _lib.const_tensor)."""
    key = (n, _lib.device_key(device))
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones(n, device=device, dtype=torch.float32)
    return t


def conv_bias_relu(x, weight, bias=None, stride=(1, 1), padding=(0, 0), gate_dx=False, dx_add_from=None):
    """relu(conv2d(x, weight, bias)) fused (weight [Cout, Cin, R, S]).  The result carries a Gate (see there);
    gate_dx as conv_bn_act's.  dx_add_from: a GradBox another reader of x (applied LATER in forward) leaves its data
    gradient in; this layer becomes the box's taker and adds the value inside its data-gradient kernel."""
    gate = Gate() if (torch.is_grad_enabled() and x.is_cuda) else None
    take = dx_add_from if (dx_add_from is not None and GATES and torch.is_grad_enabled() and x.is_cuda and
                           x.requires_grad) else None
    if take is not None:
        take.taker = True
    y = _ConvBiasActFn.apply(x, weight, bias, tuple(stride), tuple(padding), gate, _claim_gate(x, gate_dx), take)
    if gate is not None:
        y._fi_gate = gate
    return y


def _claim_gate(x, gate_dx):
    """gate_dx: True (x itself carries the Gate) or the Gate of the tensor x is a view of.  Claims it; returns whether
    this layer has to apply the mask (x > 0) to its data gradient."""
    if not GATES or not gate_dx or not (torch.is_grad_enabled() and x.is_cuda and x.requires_grad):
        return False
    tok = gate_dx if isinstance(gate_dx, Gate) else getattr(x, "_fi_gate", None)
    if tok is None:
        return False
    tok.claimed = True
    return True


# ---- eval-BN fold (scale, shift) for every (conv, bn) pair, refreshed once per step ------------
# Folding inside each layer costs ~5 tiny launches per layer (500 per step); refresh_bn_folds()
# does all layers with a handful of torch._foreach_* launches.  A cached fold is used only while
# the versions of the five tensors it was computed from are unchanged.
_FOLD_PAIRS = weakref.WeakKeyDictionary()      # bn module -> conv module (filled by conv_bn_act)


def _fold_key(conv, bn):
    return (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
            -1 if conv.bias is None else conv.bias._version, bn.weight.data_ptr())


def _cached_fold(conv, bn):
    c = getattr(bn, "_fi_fold", None)
    if c is not None and c[2] == _fold_key(conv, bn):
        return c[0], c[1]
    return None


_FOLD_TABLE = {"key": None, "refs": []}      # descriptor table of fi_bn_fold_batch for the current set of (conv, bn) pairs


@torch.no_grad()
def refresh_bn_folds():
    """Recompute scale = gamma/sqrt(var+eps), shift = beta + (conv_bias - mean)*scale for every
    eval-mode (conv, bn) pair seen so far whose parameters changed.  Call once per step before the forward pass.
    ONE launch (fi_bn_fold_batch) over a descriptor table that is rebuilt only when the set of pairs or an address
    changes; the outputs are persistent per BatchNorm."""
    pairs, keys = [], []
    for b, c in _FOLD_PAIRS.items():
        if b.training or not b.weight.is_cuda:
            continue
        fk = _fold_key(c, b)
        cur = getattr(b, "_fi_fold", None)
        if cur is None or cur[2] != fk:
            pairs.append((c, b))
            keys.append(fk)
    if not pairs:
        return
    import numpy as np
    L = _lib.load()
    dev = pairs[0][1].weight.device
    key = tuple((id(b), fk[-1]) for (_, b), fk in zip(pairs, keys))
    t = _FOLD_TABLE
    # ids and addresses are reused once a model is freed: the table also remembers (weakly) WHICH modules it is for
    same = t["key"] == key and all(r() is b for r, (_, b) in zip(t["refs"], pairs))
    if not same:
        desc = np.zeros(len(pairs), dtype=np.dtype([("gamma", "<u8"), ("beta", "<u8"), ("mean", "<u8"), ("var", "<u8"),
                                                    ("cb", "<u8"), ("scale", "<u8"), ("shift", "<u8"), ("ch", "<i4"),
                                                    ("eps", "<f4")]))
        outs = []
        for i, (c, b) in enumerate(pairs):
            C = b.num_features
            # one (scale, shift) pair per BatchNorm for its lifetime: this step's W^T table (prepare_step) holds
            # the address of the scale
            keep = getattr(b, "_fi_fold_out", None)
            if keep is None or keep[0].device != dev or keep[0].numel() != C:
                keep = b._fi_fold_out = (torch.empty(C, device=dev, dtype=torch.float32),
                                         torch.empty(C, device=dev, dtype=torch.float32))
            sc, sh = keep
            outs.append((sc, sh))
            desc[i] = (b.weight.data_ptr(), b.bias.data_ptr(), b.running_mean.data_ptr(), b.running_var.data_ptr(),
                       0 if c.bias is None else c.bias.data_ptr(), sc.data_ptr(), sh.data_ptr(), C, float(b.eps))
        t.update(key=key, outs=outs, n=len(pairs), maxc=max(b.num_features for _, b in pairs),
                 refs=[weakref.ref(b) for _, b in pairs],
                 table=torch.from_numpy(desc.view(np.uint8).copy()).to(dev))
    with torch.cuda.device(dev):
        _lib.check(L.fi_bn_fold_batch(_lib.ptr(t["table"]), t["n"], t["maxc"], _lib.current_stream()), "fi_bn_fold_batch")
    for (c, b), (sc, sh), fk in zip(pairs, t["outs"], keys):
        b._fi_fold = (sc, sh, fk)


def invalidate_bn_folds(module=None):
    invalidate_step_state()
    _invalidate_bn_folds(module)


def _invalidate_bn_folds(module=None):
    """Forget cached (scale, shift) folds -- of `module`'s BatchNorms, or of every BatchNorm seen
    so far.  Needed after writes that bypass the tensor version counters the cache is keyed on
    (`.data.copy_`, dist.broadcast(t.data), load_state_dict under torch.no_grad is counted, but
    `.data` surgery is not)."""
    mods = module.modules() if module is not None else list(_FOLD_PAIRS.keys())
    for m in mods:
        if getattr(m, "_fi_fold", None) is not None:
            m._fi_fold = None


FUSED_FC = _os.environ.get("FI_FUSED_FC", "1") != "0"       # A/B switch: linear_bn_act vs conv, affine, ReLU as torch ops


def conv_bn_act(x, conv, bn, relu=True, residual=None, channels_last_out=False, res_grad_to=None,
                dx_add_from=None, dx_give_to=None, gate_dx=False, live=None):
    """act(bn(conv(x)) [+ residual]) for an eval-mode BatchNorm2d (the reference always evaluates
    BN with running statistics, lib/model.py:265-267).  Falls back to separate ops for a BN in
    training mode or a full-window (GEMM) convolution.  channels_last_out: return the result in
    torch.channels_last memory format (for maps that only the channels-last RoIAlign reads).
    gate_dx: the caller's promise that NOTHING else reads x (or that every other gradient into x reaches this layer
    through dx_add_from).  If x is the output of a fused conv + BN + ReLU, this layer's data-gradient kernel then
    applies that ReLU's mask in its epilogue and the producer skips its masking pass (see Gate)."""
    R, S = conv.weight.shape[2], conv.weight.shape[3]
    gemm_path = ((x.shape[2], x.shape[3]) == (R, S) and tuple(conv.padding) == (0, 0)) or x.shape[2] * x.shape[3] == 1
    if bn.training or gemm_path or not bn.track_running_stats:
        assert res_grad_to is None and dx_add_from is None and dx_give_to is None, \
            "gradient hand-off needs the fused conv+BN path"
        if gemm_path and not bn.training and bn.track_running_stats and x.is_cuda and x.dtype == torch.float32 and \
                residual is None and FUSED_FC and (x.shape[1] * R * S) % 4 == 0 and conv.weight.shape[0] % 4 == 0 and \
                x.shape[0] > 0:
            y = linear_bn_act(x, conv, bn, relu, live)
            return y.contiguous(memory_format=torch.channels_last) if channels_last_out else y
        if gemm_path and not bn.training and bn.track_running_stats:
            # [N,C,1,1]: MIOpen's spatial inference BN takes ~0.4 ms on 8 MB here; the affine form is ~10 us
            scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
            z = conv(x) if live is None else conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding, live=live)
            y = z * scale.view(1, -1, 1, 1) + (bn.bias - bn.running_mean * scale).view(1, -1, 1, 1)
        else:
            y = bn(conv(x))
        if residual is not None:
            y = y + residual
        y = F.relu(y) if relu else y
        return y.contiguous(memory_format=torch.channels_last) if channels_last_out else y
    out_cl = bool(channels_last_out) and residual is None and conv.weight.shape[0] % 4 == 0
    _FOLD_PAIRS[bn] = conv
    track = torch.is_grad_enabled() and x.is_cuda
    claimed = _claim_gate(x, gate_dx)
    out_gate = Gate() if (relu and track and not out_cl) else None
    y = _ConvBnActFn.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                           bn.eps, residual, relu, tuple(conv.stride), tuple(conv.padding), out_cl,
                           _cached_fold(conv, bn), res_grad_to, dx_add_from, dx_give_to, claimed, out_gate, live)
    if out_gate is not None:
        y._fi_gate = out_gate
    return y.contiguous(memory_format=torch.channels_last) if (channels_last_out and not out_cl) else y



# ---- fully connected layers on the convolution kernels -----------------------------------------------------------
# The heads' full-window 7x7 "fc" convolutions (lib/sub_module.py:707, :333), their 1x1 convolutions on 1x1 maps,
# the class / box nn.Linear layers (:744-747) and the OT module's centre-tap Conv1d (lib/OT_module.py:37-41) are
# plain matrix products y = x W^T.  They run on the library's own fp32 MFMA kernels (no vendor GEMM):
#   y  [M,N] = x [M,K] . W[N,K]^T   both operands contiguous along the reduction  -> the weight-gradient kernel
#              (dW[co][ci] = sum_p dz[co][p] x[ci][p] with co = m, ci = n, p = k), split over K, fp32 atomics;
#   dx [M,K] = dy[M,N] . W[N,K]     reduction over rows of W                       -> the 1x1 convolution kernel
#              (Y[co][p] = sum_ci A[co][ci] X[ci][p] with A = dy, X = W, p = k);
#   dW [N,K] = dy^T[N,M] . x[M,K]   reduction over rows of x                       -> the 1x1 convolution kernel
#              with A = dy^T (a small transposed copy), X = x.
# The kernels want the reduction length of the second and third product (N, M) to be a multiple of 32 (64 on the
# 16-bit kernels) and the output columns of the first (N) a multiple of 128: N is padded with zero rows of W / zero
# columns of dy (small copies), M by zero rows of x and dy (only when needed: the callers keep M a multiple of 64).
# With MODEL.CONV_PRECISION 'bf16' / 'fp16' the three products run on the 16-bit MFMA kernels like every other
# convolution of the step (the full-window "fc" layers ARE convolutions in the reference).
def _pad_rows(t, rows):
    if t.shape[0] == rows:
        return t.contiguous()
    out = t.new_zeros((rows,) + tuple(t.shape[1:]))
    out[:t.shape[0]] = t
    return out


def _gemm_nt(a, b, bias=None, relu=False, precision="fp32", live=None, scale=None):
    """act(a [M,K] . b[N,K]^T + bias) -> [M,N]; N % 128 == 0, K % 4 == 0.  fp32: fi_gemm_nt (deterministic split over
    K).  16-bit precisions: the 16-bit weight-gradient kernel (operands rounded on their way into LDS, fp32 atomics
    over the split) when M % 64 == 0, then bias / ReLU."""
    L = _lib.load()
    M, K = a.shape
    N = b.shape[0]
    if precision in _LOWP and M % 64 == 0 and K >= 64:
        y = torch.empty((M, N), device=a.device, dtype=torch.float32)
        _log_flops("bf16_wgrad", M, 1, 1, _live_share(live, M) * 2.0 * M * N * K)
        with torch.cuda.device(a.device):
            if live is not None:
                _lib.check(_lowp_fn(L, "conv2d_weight_grad_rows", precision)(_lib.ptr(b), _lib.ptr(a), _lib.ptr(y), 1, N, 1, K,
                                                                             M, 1, 1, 1, 1, 0, 0, 0, _lib.ptr(live),
                                                                             _lib.current_stream()),
                           "fi_conv2d_weight_grad_rows_16 (gemm)")
            else:
                _lib.check(_lowp_fn(L, "conv2d_weight_grad", precision)(_lib.ptr(b), _lib.ptr(a), _lib.ptr(y), 1, N, 1, K, M,
                                                                        1, 1, 1, 1, 0, 0, 0, _lib.current_stream()),
                           "fi_conv2d_weight_grad_16 (gemm)")
        if scale is not None:         # (fully connected + eval BatchNorm + ReLU: one in-place pass behind the atomics)
            with torch.cuda.device(a.device):
                _lib.check(L.fi_rows_affine_act(_lib.ptr(y), _lib.ptr(scale), _lib.ptr(bias), M, N, 1 if relu else 0,
                                                _lib.current_stream()), "fi_rows_affine_act")
            return y
        if bias is not None:
            y = y + bias
        return torch.relu_(y) if relu else y
    y = torch.empty((M, N), device=a.device, dtype=torch.float32)
    ws = torch.empty((int(L.fi_gemm_nt_workspace_bytes(M, N, K)) + 3) // 4, device=a.device, dtype=torch.float32)
    _log_flops("wgrad", M, 1, 1, _live_share(live, M) * 2.0 * M * N * K, K, N)
    with torch.cuda.device(a.device):
        _lib.check(L.fi_gemm_nt_affine(_lib.ptr(a), _lib.ptr(b), _lib.ptr(scale), _lib.ptr(bias), _lib.ptr(y), M, N, K,
                                       1 if relu else 0, _lib.ptr(ws), _lib.ptr(live), _lib.current_stream()), "fi_gemm_nt")
    return y


def _gemm_nn(a, b, precision="fp32"):
    """a [M,R] . b [R,K] -> [M,K]; R % 32 == 0, K % 4 == 0."""
    M, R = a.shape
    K = b.shape[1]
    return _conv_fwd(b.view(1, R, 1, K), a.view(M, R, 1, 1), None, (1, 1), (0, 0), precision=precision).view(M, K)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, live=None):
        _lib.require_cuda(x, w)
        M, K = x.shape
        N = w.shape[0]
        prec = _PRECISION
        rows = 64 if prec in _LOWP else 32
        Mp, Np = (M + rows - 1) // rows * rows, (N + 127) // 128 * 128
        xp = _pad_rows(x.float(), Mp)
        wp = _pad_rows(w.float(), Np)
        bp = _pad_rows(b.float(), Np) if b is not None else None
        y = _gemm_nt(xp, wp, bp, precision=prec, live=live)[:M, :N]      # bias added by the split reduction
        ctx.save_for_backward(xp, wp)
        ctx.dims = (M, N, K, b is not None)
        ctx.precision = prec
        return y.contiguous()

    @staticmethod
    def backward(ctx, dy):
        xp, wp = ctx.saved_tensors
        M, N, K, has_b = ctx.dims
        Mp, Np = xp.shape[0], wp.shape[0]
        dy = dy.float()
        dyp = dy.new_zeros((Mp, Np))
        dyp[:M, :N] = dy
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = _gemm_nn(dyp, wp, ctx.precision)[:M]        # reduction over the padded N (zero columns x zero rows)
        if ctx.needs_input_grad[1]:
            dw = _gemm_nn(dyp.t().contiguous(), xp, ctx.precision)[:N]      # reduction over the padded M
        db = dy.sum(0) if (has_b and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None


class _LinearBnActFn(torch.autograd.Function):
    """act((x W^T + conv_bias - mean) * gamma / sqrt(var + eps) + beta): a fully connected layer with its eval-mode
    BatchNorm and ReLU (the heads' full-window "fc" convolutions and their 1x1 convolutions on 1x1 maps:
    nn.Conv2d + nn.BatchNorm2d + nn.ReLU in the reference, lib/sub_module.py:333-340, :707-716).  Forward: the affine
    map and the ReLU ride in the GEMM's reduction pass (fi_gemm_nt_affine).  Backward, as for the convolutions
    (_ConvBnActFn): ONE pass makes g = dy * (y > 0), g * scale and the column sums of g (fi_rows_mask_scale); the data
    gradient is (g * scale) W, the weight gradient of the unscaled g gives dW, d gamma, d conv_bias through
    fi_bn_fold_grad (sum_m g * (x W^T) = <W, dW'>), d beta is the column sums."""

    @staticmethod
    def forward(ctx, x, w, cb, gamma, beta, mean, var, eps, relu, fold, live):
        _lib.require_cuda(x, w)
        M, K = x.shape
        N = w.shape[0]
        prec = _PRECISION
        rows = 64 if prec in _LOWP else 32
        Mp, Np = (M + rows - 1) // rows * rows, (N + 127) // 128 * 128
        if fold is not None:
            scale, shift = fold
        else:
            scale = gamma * torch.rsqrt(var + eps)
            shift = beta - mean * scale if cb is None else beta + (cb - mean) * scale
        xp = _pad_rows(x.float(), Mp)
        wp = _pad_rows(w.float(), Np)
        # (zero scale and shift in the padding columns: they come out as zeros)
        y = _gemm_nt(xp, wp, _pad_rows(shift.float(), Np), relu=relu, precision=prec, live=live,
                     scale=_pad_rows(scale.float(), Np))[:M, :N].contiguous()
        ctx.save_for_backward(xp, wp, y, scale, mean, var, cb)
        ctx.dims = (M, N, K, float(eps), bool(relu))
        ctx.precision = prec
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, wp, y, scale, mean, var, cb = ctx.saved_tensors
        M, N, K, eps, relu = ctx.dims
        Mp, Np = xp.shape[0], wp.shape[0]
        L = _lib.load()
        dy = dy.contiguous().float()
        need = ctx.needs_input_grad
        want_w = need[1] or need[2] or need[3]
        padded = (Mp, Np) != (M, N)
        new = dy.new_zeros if padded else dy.new_empty
        g = new((Mp, Np)) if want_w else None
        gs = new((Mp, Np)) if need[0] else None
        s = dy.new_zeros(N)
        with torch.cuda.device(dy.device):
            _lib.check(L.fi_rows_mask_scale(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(scale), _lib.ptr(g), _lib.ptr(gs), _lib.ptr(s),
                                            M, N, Np, 1 if relu else 0, _lib.OUTPUTS_ZEROED, _lib.current_stream()),
                       "fi_rows_mask_scale")
        dx = _gemm_nn(gs, wp, ctx.precision)[:M] if need[0] else None
        dw = dcb = dgamma = None
        if want_w:
            dw = _gemm_nn(g.t().contiguous(), xp, ctx.precision)[:N]        # dW' of the unscaled g; [N, K] contiguous
            dgamma = dy.new_zeros(N) if need[3] else None
            dcb = dy.new_zeros(N) if (cb is not None and need[2]) else None
            with torch.cuda.device(dy.device):
                _lib.check(L.fi_bn_fold_grad(_lib.ptr(dw), _lib.ptr(wp), _lib.ptr(s), _lib.ptr(scale), _lib.ptr(mean),
                                             _lib.ptr(var), eps, _lib.ptr(cb), _lib.ptr(dgamma), _lib.ptr(dcb), N, K, 1, 0, 0,
                                             _lib.current_stream()), "fi_bn_fold_grad")
            if not need[1]:
                dw = None
        return dx, dw, dcb, dgamma, (s if need[4] else None), None, None, None, None, None, None


def linear_bn_act(x, conv, bn, relu=True, live=None):
    """conv_bn_act for a convolution that is a matrix product (full window, or a 1x1 layer on a 1x1 map) behind an
    eval-mode BatchNorm: x [M, Cin, R, S] -> [M, Cout, 1, 1]."""
    M, N = x.shape[0], conv.weight.shape[0]
    _FOLD_PAIRS[bn] = conv
    y = _LinearBnActFn.apply(x.reshape(M, -1), conv.weight.reshape(N, -1), conv.bias, bn.weight, bn.bias, bn.running_mean,
                             bn.running_var, bn.eps, relu, _cached_fold(conv, bn), live)
    return y.view(M, N, 1, 1)


def linear(x, weight, bias=None, live=None):
    """F.linear(x [M,K], weight [N,K], bias) on the library's MFMA kernels (see above).  Shapes the kernels do not
    take (K % 4 != 0, a CPU tensor) fall back to F.linear."""
    if x.dim() != 2 or not x.is_cuda or x.shape[1] % 4 != 0 or x.shape[0] == 0 or x.dtype != torch.float32:
        return F.linear(x, weight, bias)
    # live [1] int32 (device): only the first live[0] rows of x are real -- the others' outputs are zeros (no graph)
    return _LinearFn.apply(x, weight, bias, live)


def conv2d(x, weight, bias=None, stride=(1, 1), padding=(0, 0), residual=None, gate_dx=False, dx_give_to=None, live=None):
    """Functional form: conv(x) + bias [+ residual, added in the kernel epilogue].  Full-window kernels are
    matrix products (linear() above).  gate_dx: as conv_bn_act's (True, or the Gate of the tensor x is a view of).
    dx_give_to: a GradBox whose taker flag is set -- the data gradient goes there instead of to autograd (the taker,
    another reader of x whose backward runs later, adds it inside its own data-gradient kernel)."""
    R, S = weight.shape[2], weight.shape[3]
    gemm = ((x.shape[2], x.shape[3]) == (R, S) and tuple(padding) == (0, 0) and R * S > 1) or \
        (x.shape[2] * x.shape[3] == 1 and R * S == 1)
    if gemm:
        y = linear(x.reshape(x.shape[0], -1), weight.reshape(weight.shape[0], -1), bias, live)
        y = y.view(x.shape[0], weight.shape[0], 1, 1)
        return y if residual is None else y + residual
    give = dx_give_to if (dx_give_to is not None and dx_give_to.taker and x.is_cuda and torch.is_grad_enabled()) else None
    return _Conv2dFn.apply(x, weight, bias, tuple(stride), tuple(padding), residual, _claim_gate(x, gate_dx), give)


class Conv2d(nn.Conv2d):
    """nn.Conv2d (groups=1, dilation=1, zero padding) on the MFMA implicit-GEMM kernels."""

    # set by the owner for a layer that only ever sees inputs of its kernel's size (the heads' 7x7 "fc"
    # convolutions): it runs as a GEMM, so its weight stays row-major and gets no tap-major extras
    full_window = False

    def forward(self, x):
        if self.groups != 1 or self.dilation != (1, 1) or self.padding_mode != 'zeros' or \
                isinstance(self.padding, str):
            raise NotImplementedError("fi Conv2d supports groups=1, dilation=1, numeric zero padding")
        return conv2d(x, self.weight, self.bias, self.stride, self.padding)


class ConvTranspose2x2(nn.ConvTranspose2d):
    """ConvTranspose2d(kernel_size=2, stride=2) (the mask head's deconv, lib/sub_module.py:763):
    a 1x1 convolution to 4*Cout channels followed by a pixel shuffle."""

    def forward(self, x):
        assert self.kernel_size == (2, 2) and self.stride == (2, 2) and self.padding == (0, 0) and \
            self.output_padding == (0, 0) and self.groups == 1
        cin, cout = self.weight.shape[0], self.weight.shape[1]
        w = self.weight.permute(1, 2, 3, 0).reshape(cout * 4, cin, 1, 1)
        b = self.bias.repeat_interleave(4) if self.bias is not None else None
        return F.pixel_shuffle(conv2d(x, w, b), 2)

    def forward_unshuffled(self, x, relu=False, gate_dx=False):
        """The same values BEFORE the pixel shuffle, as [N, 2, 2, Cout, H, W] with
        out[n, c, 2h+a, 2w+b] == u[n, a, b, c, h, w] (optionally with ReLU fused into the 1x1 conv).
        A following 1x1 convolution / elementwise op can consume u viewed as [N*4, Cout, H, W] and
        only its (smaller) result needs shuffling -- the mask head moves 81 channels instead of 256."""
        assert self.kernel_size == (2, 2) and self.stride == (2, 2) and self.padding == (0, 0) and \
            self.output_padding == (0, 0) and self.groups == 1
        cin, cout = self.weight.shape[0], self.weight.shape[1]
        w = self.weight.permute(2, 3, 1, 0).reshape(4 * cout, cin, 1, 1)       # output channels ordered (a, b, c)
        b = self.bias.repeat(4) if self.bias is not None else None
        y = conv_bias_relu(x, w, b, gate_dx=gate_dx) if relu else conv2d(x, w, b, gate_dx=gate_dx)
        u = y.view(x.shape[0], 2, 2, cout, x.shape[2], x.shape[3])
        if getattr(y, "_fi_gate", None) is not None:
            u._fi_gate = y._fi_gate           # a view of y: its only reader may claim the Gate (conv2d(gate_dx=...))
        return u


class Conv1d(nn.Conv1d):
    """The OT module's Conv1d(k=3, padding=1) layers (lib/OT_module.py:37-41, 58-63).  On the
    length-1 inputs the intertwiner feeds them only the centre tap contributes, which is a
    library GEMM; other lengths fall through to the stock implementation."""

    def forward(self, x):
        if x.size(2) == 1 and self.kernel_size == (3,) and self.padding == (1,) and self.stride == (1,) \
                and self.dilation == (1,) and self.groups == 1:
            return linear(x[:, :, 0].contiguous(), self.weight[:, :, 1], self.bias).unsqueeze(2)
        return super(Conv1d, self).forward(x)
