"""Drop-in for lib/roi_align/crop_and_resize.py:14-54 of the reference.

`CropAndResizeFunction(crop_height, crop_width, extrapolation_value=0)(image, boxes,
box_ind)` keeps the reference's call shape (an instance that is then called); the
autograd part is a new-style torch.autograd.Function because old-style instance
Functions no longer exist.  Gradients flow to `image` only (reference :54).
"""
import torch

from .. import _lib

# bench.py sets this to a list to record, per forward launch, what is needed to compute the
# launch's algorithmic bytes (references only; no copies, no synchronisation).
LAUNCH_LOG = None


def _is_channels_last(t):
    """[B,C,H,W] tensor whose memory is [B,H,W,C] (and not also plain-contiguous, as when C or H*W is 1)."""
    return t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last)


class _CropAndResize(torch.autograd.Function):
    """Dispatches on the memory format of `image`: an NCHW-contiguous map takes the reference-shaped entry
    point fi_crop_and_resize_forward; a map in torch.channels_last format ([B,H,W,C] in memory) is NOT
    transposed -- it takes the channels-last kernels (fi_pyramid_crop_*_nhwc with one level), where one tap is
    C contiguous floats, and its gradient comes back channels-last.  Results are bit-identical."""

    @staticmethod
    def forward(ctx, image, boxes, box_ind, crop_height, crop_width, extrapolation_value):
        import ctypes
        _lib.require_cuda(image, boxes, box_ind)
        L = _lib.load()
        cl = _is_channels_last(image) and int(crop_height) * int(crop_width) <= 220
        image = image.float() if cl else image.contiguous().float()
        boxes_c = boxes.detach().contiguous().float()
        ind_c = box_ind.detach().contiguous().to(torch.int32)
        B, C, H, W = image.shape
        N = boxes_c.shape[0]
        crops = torch.empty((N, C, crop_height, crop_width), device=image.device, dtype=torch.float32)
        with torch.cuda.device(image.device):
            if cl:
                _lib.check(L.fi_pyramid_crop_forward_nhwc(
                    (ctypes.c_void_p * 1)(image.data_ptr()), (ctypes.c_int * 1)(H), (ctypes.c_int * 1)(W), 1,
                    _lib.ptr(boxes_c), _lib.ptr(ind_c), None, N, B, C, int(crop_height), int(crop_width),
                    float(extrapolation_value), _lib.ptr(crops), _lib.current_stream()), "fi_pyramid_crop_forward_nhwc")
            else:
                _lib.check(L.fi_crop_and_resize_forward(
                    _lib.ptr(image), _lib.ptr(boxes_c), _lib.ptr(ind_c), N, B, C, H, W,
                    int(crop_height), int(crop_width), float(extrapolation_value), _lib.ptr(crops),
                    None, _lib.current_stream()), "fi_crop_and_resize_forward")
        ctx.im_size = (B, C, H, W)
        ctx.channels_last = cl
        ctx.crop = (int(crop_height), int(crop_width))
        ctx.save_for_backward(boxes_c, ind_c)
        return crops

    @staticmethod
    def backward(ctx, grad_outputs):
        import ctypes
        boxes_c, ind_c = ctx.saved_tensors
        L = _lib.load()
        g = grad_outputs.contiguous().float()
        B, C, H, W = ctx.im_size
        grad_image = torch.empty((B, C, H, W), device=g.device, dtype=torch.float32,
                                 memory_format=torch.channels_last if ctx.channels_last else torch.contiguous_format)
        with torch.cuda.device(g.device):
            if ctx.channels_last:
                _lib.check(L.fi_pyramid_crop_backward_nhwc(
                    _lib.ptr(g), (ctypes.c_void_p * 1)(grad_image.data_ptr()), (ctypes.c_int * 1)(H),
                    (ctypes.c_int * 1)(W), 1, _lib.ptr(boxes_c), _lib.ptr(ind_c), None, boxes_c.shape[0], B, C,
                    ctx.crop[0], ctx.crop[1], _lib.current_stream()), "fi_pyramid_crop_backward_nhwc")
            else:
                _lib.check(L.fi_crop_and_resize_backward(
                    _lib.ptr(g), _lib.ptr(boxes_c), _lib.ptr(ind_c), boxes_c.shape[0], B, C, H, W,
                    ctx.crop[0], ctx.crop[1], _lib.ptr(grad_image), _lib.current_stream()),
                    "fi_crop_and_resize_backward")
        return grad_image, None, None, None, None, None


class CropAndResizeFunction(object):
    """Single-tap bilinear crop (tf.image.crop_and_resize semantics)."""

    def __init__(self, crop_height, crop_width, extrapolation_value=0):
        self.crop_height = crop_height
        self.crop_width = crop_width
        self.extrapolation_value = extrapolation_value

    def __call__(self, image, boxes, box_ind):
        return _CropAndResize.apply(image, boxes, box_ind, self.crop_height, self.crop_width,
                                    self.extrapolation_value)

    forward = __call__


class CropGradGroup(object):
    """Shared by the pyramid crops of ONE set of maps (the Dev stage pools 7x7 and 14x14 from the same four maps):
    the first backward of the group allocates and clears the maps' gradients and returns them to autograd, every
    later one ADDS into the same buffers (fi_pyramid_crop_backward*_accumulate) and returns no gradient -- instead
    of one cleared set of buffers per crop and an add pass per level."""
    __slots__ = ("grads", "give_to")

    def __init__(self, give_to=None):
        self.grads = None
        # give_to: per map, a conv.GradBox (or None): the map's gradient is left there for another reader of the map,
        # whose data-gradient kernel adds it, instead of going to autograd (only boxes whose taker flag is set)
        self.give_to = give_to


class _PyramidCrop(torch.autograd.Function):
    """All FPN levels in one launch (the callers' per-level loops, lib/layers.py:183-216 and
    lib/sub_module.py:429-662, collapse into a precomputed `level` vector)."""

    @staticmethod
    def forward(ctx, boxes, box_ind, level, crop_height, crop_width, extrapolation_value, group, *maps):
        import ctypes
        _lib.require_cuda(boxes, box_ind, level, *maps)
        L = _lib.load()
        # maps in torch.channels_last memory format ([B,H,W,C] in memory) take the channels-last kernels
        cl = all(_is_channels_last(m) for m in maps)
        maps = [m.float() if cl else m.contiguous().float() for m in maps]
        nl = len(maps)
        B, C = maps[0].shape[:2]
        boxes_c = boxes.detach().contiguous().float()
        ind_c = box_ind.detach().contiguous().to(torch.int32)
        lvl_c = level.detach().contiguous().to(torch.int32)
        N = boxes_c.shape[0]
        crops = torch.empty((N, C, crop_height, crop_width), device=boxes_c.device,
                            dtype=torch.float32)
        ptrs = (ctypes.c_void_p * nl)(*[m.data_ptr() for m in maps])
        hs = (ctypes.c_int * nl)(*[m.shape[2] for m in maps])
        ws = (ctypes.c_int * nl)(*[m.shape[3] for m in maps])
        fn = L.fi_pyramid_crop_forward_nhwc if cl else L.fi_pyramid_crop_forward
        with torch.cuda.device(boxes_c.device):
            _lib.check(fn(
                ptrs, hs, ws, nl, _lib.ptr(boxes_c), _lib.ptr(ind_c), _lib.ptr(lvl_c), N, B, C,
                int(crop_height), int(crop_width), float(extrapolation_value), _lib.ptr(crops),
                _lib.current_stream()), "fi_pyramid_crop_forward")
        ctx.channels_last = cl
        ctx.group = group
        if _lib.TAP is not None:
            _lib.TAP("pyramid_crop", maps=maps, boxes=boxes_c, box_ind=ind_c, level=lvl_c, crops=crops,
                     crop=int(crop_height))
        if LAUNCH_LOG is not None:
            LAUNCH_LOG.append({"pyramid": True, "nhwc": cl, "crop": int(crop_height), "depth": int(C), "boxes": boxes_c,
                               "level": lvl_c, "box_ind": ind_c, "shapes": [(m.shape[2], m.shape[3]) for m in maps]})
        ctx.shapes = [tuple(m.shape) for m in maps]
        ctx.crop = (int(crop_height), int(crop_width))
        ctx.save_for_backward(boxes_c, ind_c, lvl_c)
        return crops

    @staticmethod
    def backward(ctx, grad_outputs):
        import ctypes
        boxes_c, ind_c, lvl_c = ctx.saved_tensors
        L = _lib.load()
        g = grad_outputs.contiguous().float()
        nl = len(ctx.shapes)
        B, C = ctx.shapes[0][:2]
        fmt = torch.channels_last if ctx.channels_last else torch.contiguous_format
        group = ctx.group
        accumulate = group is not None and group.grads is not None
        if accumulate:
            grads = group.grads
        else:
            grads = [torch.empty(s, device=g.device, dtype=torch.float32, memory_format=fmt) for s in ctx.shapes]
            if group is not None:
                group.grads = grads
                # the sharing holds for ONE backward pass: a further pass through the same graph (retain_graph, a second
                # loss) must start with fresh buffers and hand them to autograd again
                try:
                    torch.autograd.Variable._execution_engine.queue_callback(lambda g=group: setattr(g, "grads", None))
                except RuntimeError:          # called outside a backward pass
                    pass
        ptrs = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in grads])
        hs = (ctypes.c_int * nl)(*[s[2] for s in ctx.shapes])
        ws = (ctypes.c_int * nl)(*[s[3] for s in ctx.shapes])
        if accumulate:
            fn = L.fi_pyramid_crop_backward_nhwc_accumulate if ctx.channels_last else L.fi_pyramid_crop_backward_accumulate
        else:
            fn = L.fi_pyramid_crop_backward_nhwc if ctx.channels_last else L.fi_pyramid_crop_backward
        with torch.cuda.device(g.device):
            _lib.check(fn(
                _lib.ptr(g), ptrs, hs, ws, nl, _lib.ptr(boxes_c), _lib.ptr(ind_c), _lib.ptr(lvl_c),
                boxes_c.shape[0], B, C, ctx.crop[0], ctx.crop[1], _lib.current_stream()),
                "fi_pyramid_crop_backward")
        if accumulate:
            return (None,) * (7 + nl)
        out = list(grads)
        if group is not None and group.give_to:
            for i, box in enumerate(group.give_to):
                if box is not None and box.taker:
                    box.value, out[i] = out[i], None
        return (None, None, None, None, None, None, None) + tuple(out)


def pyramid_crop_and_resize(feature_maps, boxes, box_ind, level, crop_height, crop_width,
                            extrapolation_value=0.0, grad_group=None):
    """crops[i] = crop of box i from feature_maps[level[i] - 2]; rows with a level
    outside the pyramid are zero.  Output rows are in the order of `boxes`.
    grad_group: a CropGradGroup shared by every crop of the SAME maps (same tensors, same memory format) whose
    results all take part in the same backward pass."""
    return _PyramidCrop.apply(boxes, box_ind, level, crop_height, crop_width, extrapolation_value, grad_group,
                              *feature_maps)
