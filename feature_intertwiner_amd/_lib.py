"""ctypes binding of libfi_hip.so (the C ABI declared in include/fi_capi.h).

The product path has NO fallback: if the shared library is missing or fails to
load, importing any operator raises.  torch is imported first on purpose so that
the HIP runtime already resident in the process (torch's bundled libamdhip64,
soname libamdhip64.so.7) is the one our library binds to -- device pointers and
streams are then shared between torch and the kernels.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FI_LIB_PATH", os.path.join(_HERE, "libfi_hip.so"))   # override: kernel A/B builds

c_int = ctypes.c_int
c_float = ctypes.c_float
c_void_p = ctypes.c_void_p
c_size_t = ctypes.c_size_t
c_char_p = ctypes.c_char_p
_pp = ctypes.POINTER(ctypes.c_void_p)
_ip = ctypes.POINTER(ctypes.c_int)

# name -> (restype, argtypes); mirrors include/fi_capi.h one to one
SIGNATURES = {
    "fi_version": (c_char_p, []),
    "fi_last_error": (c_char_p, []),
    "fi_crop_and_resize_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                           c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                           c_void_p]),
    "fi_crop_and_resize_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                            c_int, c_int, c_int, c_void_p, c_void_p]),
    "fi_crop_and_resize_taps": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int] +
                                [c_void_p] * 8 + [c_void_p]),
    "fi_pyramid_crop_forward": (c_int, [_pp, _ip, _ip, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                        c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "fi_pyramid_crop_backward": (c_int, [c_void_p, _pp, _ip, _ip, c_int, c_void_p, c_void_p,
                                         c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fi_pyramid_crop_forward_nhwc": (c_int, [_pp, _ip, _ip, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                             c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "fi_pyramid_crop_backward_nhwc": (c_int, [c_void_p, _pp, _ip, _ip, c_int, c_void_p, c_void_p,
                                              c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fi_pyramid_crop_backward_accumulate": (c_int, [c_void_p, _pp, _ip, _ip, c_int, c_void_p, c_void_p,
                                                    c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fi_pyramid_crop_backward_nhwc_accumulate": (c_int, [c_void_p, _pp, _ip, _ip, c_int, c_void_p, c_void_p,
                                                         c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fi_roi_pool_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                    c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "fi_roi_pool_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                     c_int, c_int, c_float, c_void_p, c_void_p]),
    "fi_nms_workspace_bytes": (c_size_t, [c_int, c_int]),
    "fi_nms_sorted": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_int, c_int, c_void_p,
                              c_void_p, c_void_p, c_void_p]),
    "fi_sinkhorn_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_int,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fi_class_mean_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "fi_class_mean_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                      c_void_p, c_void_p]),
    "fi_class_mean_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                       c_void_p]),
    "fi_conv2d_forward": (c_int, [c_void_p] * 6 + [c_int] * 16 + [c_void_p]),
    "fi_bn_act_backward": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_void_p] * 5 + [c_int, c_int, c_void_p]),
    "fi_conv2d_weight_grad": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 12 + [c_void_p, c_int, c_void_p]),
    "fi_detector_losses_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "fi_detector_losses": (c_int, [c_void_p] * 6 + [c_int, c_int] + [c_void_p] * 4 + [c_int, c_int] + [c_void_p] * 3 +
                           [c_int, c_int, c_int] + [c_void_p] * 7 + [c_void_p]),
    "fi_rpn_targets_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "fi_rpn_targets": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_float, c_float, c_int, ctypes.POINTER(c_float)] +
                       [c_void_p] * 5 + [c_void_p]),
    "fi_detection_targets": (c_int, [c_void_p] * 6 + [c_int] * 5 + [ctypes.c_double, c_int, ctypes.POINTER(c_float)] +
                             [c_void_p] * 6 + [c_void_p]),
    "fi_conv2d_weight_grad_batch": (c_int, [c_void_p] * 4 + [c_int] * 14 + [c_void_p]),
    "fi_conv2d_weight_grad_batch_bf16": (c_int, [c_void_p] * 4 + [c_int] * 14 + [c_void_p]),
    "fi_conv2d_weight_grad_batch_f16": (c_int, [c_void_p] * 4 + [c_int] * 14 + [c_void_p]),
    "fi_conv2d_forward_gated_bf16": (c_int, [c_void_p] * 7 + [c_int] * 16 + [c_void_p]),
    "fi_conv3x3_forward_gated_bf16w": (c_int, [c_void_p] * 7 + [c_int] * 7 + [c_void_p]),
    "fi_conv1x1_forward_gated_bf16w": (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p]),
    "fi_conv2d_weight_grad_db_bf16": (c_int, [c_void_p] * 4 + [c_int] * 12 + [c_void_p]),
    "fi_conv2d_forward_gated_f16": (c_int, [c_void_p] * 7 + [c_int] * 16 + [c_void_p]),
    "fi_conv3x3_forward_gated_f16w": (c_int, [c_void_p] * 7 + [c_int] * 7 + [c_void_p]),
    "fi_conv1x1_forward_gated_f16w": (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p]),
    "fi_conv2d_weight_grad_db_f16": (c_int, [c_void_p] * 4 + [c_int] * 12 + [c_void_p]),
    "fi_conv2d_forward_bf16": (c_int, [c_void_p] * 6 + [c_int] * 16 + [c_void_p]),
    "fi_conv2d_weight_grad_bf16": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 12 + [c_void_p]),
    "fi_conv2d_forward_f16": (c_int, [c_void_p] * 6 + [c_int] * 16 + [c_void_p]),
    "fi_conv2d_weight_grad_f16": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 12 + [c_void_p]),
    "fi_conv3x3_forward_f16w": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p]),
    "fi_conv1x1_forward_f16w": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p]),
    "fi_gemm_nt_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "fi_gemm_nt": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fi_conv2d_forward_live_bf16": (c_int, [c_void_p] * 7 + [c_int] * 16 + [c_void_p, c_void_p]),
    "fi_conv2d_forward_live_f16": (c_int, [c_void_p] * 7 + [c_int] * 16 + [c_void_p, c_void_p]),
    "fi_conv2d_weight_grad_rows_bf16": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 12 + [c_void_p, c_void_p]),
    "fi_conv2d_weight_grad_rows_f16": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 12 + [c_void_p, c_void_p]),
    "fi_gemm_nt_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "fi_meta_stats_workspace_bytes": (ctypes.c_size_t, [c_int, c_int]),
    "fi_meta_stats_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                      c_int, c_int, c_int, c_int] + [c_void_p] * 9),
    "fi_meta_stats_backward": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p, c_void_p]),
    "fi_meta_stats_sums": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fi_meta_stats_from_sums": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 9),
    "fi_dev_stage_index": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int] + [c_void_p] * 9),
    "fi_gemm_nt_affine": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p, c_void_p, c_void_p]),
    "fi_rows_mask_scale": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p]),
    "fi_rows_affine_act": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p]),
    "fi_conv2d_forward_live": (c_int, [c_void_p] * 7 + [c_int] * 16 + [c_void_p, c_void_p]),
    "fi_weight_transpose_batch": (c_int, [c_void_p, c_int, ctypes.c_long, c_void_p]),
    "fi_conv3x3_forward_bf16w": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p]),
    "fi_conv1x1_forward_bf16w": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p]),
    "fi_proposal_candidates": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                       ctypes.POINTER(c_float), c_float, c_float, c_void_p, c_void_p]),
    "fi_proposal_workspace_bytes": (ctypes.c_size_t, [c_int]),
    "fi_proposal_candidates_ws": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                          ctypes.POINTER(c_float), c_float, c_float, c_void_p, c_void_p, ctypes.c_size_t,
                                          c_void_p]),
    "fi_proposal_gather": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_float,
                                   c_void_p, c_void_p]),
    "fi_conv2d_forward_gated": (c_int, [c_void_p] * 7 + [c_int] * 16 + [c_void_p]),
    "fi_class_row_conv1x1_workspace_bytes": (ctypes.c_size_t, [ctypes.c_long, c_int]),
    "fi_class_row_conv1x1_backward": (c_int, [c_void_p] * 7 + [ctypes.c_long, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fi_pyramid_patch_rows_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, ctypes.c_long, c_int, c_void_p, c_void_p]),
    "fi_pyramid_patch_rows_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, ctypes.c_long, c_int, c_void_p]),
    "fi_rows_gather": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_long, ctypes.c_long, c_void_p]),
    "fi_rows_scatter_add": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_long, ctypes.c_long, c_void_p]),
    "fi_rows_combine": (c_int, [c_void_p, ctypes.c_long, c_void_p, c_void_p, c_void_p, ctypes.c_long, ctypes.c_long, c_void_p]),
    "fi_maxpool3x3s2_forward": (c_int, [c_void_p, c_void_p, ctypes.c_long, c_int, c_int, c_void_p]),
    "fi_maxpool3x3s2_backward": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_long, c_int, c_int, c_int, c_void_p]),
    "fi_sum2x2": (c_int, [c_void_p, c_void_p, ctypes.c_long, c_int, c_int, c_void_p]),
    "fi_relu_mask": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_long, c_void_p]),
    "fi_bn_fold_grad": (c_int, [c_void_p] * 6 + [ctypes.c_float] + [c_void_p] * 3 + [c_int] * 5 + [c_void_p]),
    "fi_bn_fold_batch": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "fi_bn_fold_grad_batch": (c_int, [c_void_p] * 6 + [ctypes.c_float] + [c_void_p] * 3 + [c_int] * 6 + [c_void_p]),
    "fi_stride2_interleave": (c_int, [c_void_p] * 6 + [ctypes.c_long, c_int, c_int, c_void_p]),
    "fi_stride2_interleave_gated": (c_int, [c_void_p] * 7 + [ctypes.c_long, c_int, c_int, c_void_p]),
    "fi_sgd_chunks": (ctypes.c_long, [ctypes.c_long]),
    "fi_conv1x1_ring_eligible": (c_int, [c_int] * 12 + [c_void_p] * 4),
    "fi_sgd_clip_step": (c_int, [c_void_p, c_int, ctypes.c_long, c_float, c_void_p, c_void_p, c_void_p]),
    "fi_sgd_clip_step_guarded": (c_int, [c_void_p, c_int, ctypes.c_long, c_float, c_void_p, c_void_p, c_void_p]),
    "fi_calib_copy": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "fi_prof_enable": (None, [c_int]),
    "fi_prof_reset": (None, []),
    "fi_prof_get": (c_int, [c_int, _ip, ctypes.POINTER(c_float)]),
    "fi_prof_kernel_name": (c_char_p, [c_int]),
}

OUTPUTS_ZEROED = 1      # FI_OUTPUTS_ZEROED

KERNEL_IDS = {
    "crop_fwd_7x7": 0, "crop_fwd_14x14": 1, "crop_fwd_28x28": 2, "crop_fwd_generic": 3,
    "crop_bwd_7x7": 4, "crop_bwd_14x14": 5, "crop_bwd_28x28": 6, "crop_bwd_generic": 7,
    "roipool_fwd": 8, "roipool_bwd": 9, "nms_mask": 10, "nms_scan": 11, "sinkhorn": 12,
    "class_mean": 13, "bn_act_bwd": 30,
    "crop_fwd_nhwc_7x7": 31, "crop_fwd_nhwc_14x14": 32, "crop_fwd_nhwc_generic": 33,
    "crop_bwd_nhwc_7x7": 34, "crop_bwd_nhwc_14x14": 35, "crop_bwd_nhwc_generic": 36,
    "conv_bf16_fwd": 37, "conv_bf16_wgrad": 38, "conv3x3_patch": 39, "conv3x3_patch_flat": 40, "conv1x1_reg": 41,
    "proposal_select": 42, "proposal_gather": 43, "stride2_interleave": 44, "gemm_reduce": 45,
}
for _i, _bm in enumerate((64, 128)):
    for _j, _w in enumerate(("1x1", "3x3", "7x7", "other")):
        KERNEL_IDS["conv_fwd_bm%d_%s" % (_bm, _w)] = 14 + 4 * _i + _j
        KERNEL_IDS["conv_wgrad_bm%d_%s" % (_bm, _w)] = 22 + 4 * _i + _j


def patch_mode(N, Cin, H, W, Cout, R, S, stride, padding, tap_major, out_hw_same, out_channels_last):
    """Mirrors patch_eligible() in csrc/conv_igemm.hip: 0 = conv_fwd_kernel, 1 = conv3x3_patch_kernel<false>
    (2-D tiles), 2 = conv3x3_patch_kernel<true> (flat tiles of 14-wide RoI maps)."""
    if (R, S) != (3, 3) or tuple(stride) != (1, 1) or tuple(padding) != (1, 1):
        return 0
    if not tap_major or Cout <= 64 or not out_hw_same:
        return 0
    if out_channels_last and not (W % 16 == 0 and Cout % 128 == 0):
        return 0
    mt = (Cout + 127) // 128
    if W % 16 == 0:
        return 1 if ((N * H + 7) // 8) * (W // 16) * mt >= 256 else 0
    if W < 16 and W % 2 == 0 and (W + 126) // W + 2 <= 13 and not out_channels_last:
        return 2 if ((N * H * W + 127) // 128) * mt >= 512 else 0
    return 0


def reg1x1_mode(N, Cin, H, W, Cout, R, S, stride, padding, out_channels_last):
    """Mirrors the conv1x1_reg_kernel dispatch in fi_conv2d_forward (csrc/conv_igemm.hip)."""
    return ((R, S) == (1, 1) and tuple(stride) == (1, 1) and tuple(padding) == (0, 0) and not out_channels_last and
            Cin % 32 == 0 and Cin >= 128 and Cout > 64 and (H * W) % 4 == 0 and
            ((N * H * W + 127) // 128) * ((Cout + 127) // 128) >= 256)


def conv_kernel_key(kind, cout, R, S, pixels=None, cin=None, batch=1):
    """Name (KERNEL_IDS key) of the device kernel instance a convolution launch uses (mirrors
    use_bm64() and the tile choice of fi_conv2d_weight_grad in csrc/conv_igemm.hip: 64-row tiles for
    narrow layers and under-filled grids)."""
    w = {(1, 1): "1x1", (3, 3): "3x3", (7, 7): "7x7"}.get((R, S), "other")
    bm64 = cout <= 64
    if kind == "fwd" and not bm64 and pixels is not None:
        bm64 = ((pixels + 127) // 128) * ((cout + 127) // 128) < 512
    if kind == "wgrad" and not bm64 and pixels is not None and cin is not None:
        tiles128 = ((cin * R * S + 127) // 128) * ((cout + 127) // 128)
        max_splits = (pixels + 511) // 512
        bm64 = batch * tiles128 * min(max(1, 1024 // tiles128), max_splits) < 768      # batch: fi_conv2d_weight_grad_batch
    return "conv_%s_bm%d_%s" % (kind, 64 if bm64 else 128, w)


def kernel_name(key):
    return load().fi_prof_kernel_name(KERNEL_IDS[key]).decode()

_lib = None

# Test hook: `TAP(name, **tensors)` is called by the pyramid RoIAlign, NMS and Sinkhorn operators
# right after their launch with their input and output tensors (references, no copies, no
# synchronisation), so that a test can re-check what ran INSIDE a train step against the oracle.
# None in production.
TAP = None


class FiError(RuntimeError):
    pass


def load():
    """Load libfi_hip.so and attach the signatures.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FiError(
            "libfi_hip.so not found at %s -- build it with "
            "`python -m feature_intertwiner_amd.build` (there is no CPU/PyTorch fallback)" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = load().fi_last_error().decode("utf-8", "replace")
        raise FiError("%s failed (status %d): %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream():
    """hipStream_t of torch's current stream on the current device."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise FiError("feature_intertwiner_amd operators run on the GPU only "
                          "(got a %s tensor); there is no CPU fallback" % t.device)


_CONST = {}


def device_key(device):
    """'cuda' without an index means "the current device": resolve it, so that a cache keyed on the result
    cannot hand a tensor of another GPU to a later caller."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return str(device)


def const_tensor(values, device, dtype=torch.float32):
    """Small constant (python numbers / numpy array) as a device tensor, created ONCE per (values, device):
    torch.tensor(list, device=gpu) inside the step is a pageable host-to-device copy, i.e. a full stream
    synchronisation every time.  The tensor is SHARED by every caller: read-only by contract (never the
    target of an in-place operation)."""
    try:
        key = (tuple(float(v) for v in values), device_key(device), dtype)
    except TypeError:
        key = ((float(values),), device_key(device), dtype)
        values = [values]
    t = _CONST.get(key)
    if t is None:
        t = torch.tensor([float(v) for v in values], dtype=dtype, device=device)
        _CONST[key] = t
    return t


def async_host_read(t):
    """Start copying a small device tensor to pinned host memory on a side stream and return a function
    that waits for JUST that copy: the caller keeps enqueueing independent work on the current stream in
    between, so the device does not drain while the host waits for a count."""
    dev = t.device
    side = _SIDE.get(dev)
    if side is None:
        side = _SIDE[dev] = torch.cuda.Stream(device=dev)
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        side.wait_event(ready)
        host.copy_(t, non_blocking=True)
        done = torch.cuda.Event()
        done.record(side)
    t.record_stream(side)

    def wait():
        done.synchronize()
        return host
    return wait


_SIDE = {}
_SIDE2 = {}
_SIDE3 = {}

# ---- streams that really run next to each other -----------------------------------------------------------------------
# A process gets GPU_MAX_HW_QUEUES hardware queues (default 4); a HIP stream is bound to one of them at its first use,
# the least referenced one once all exist, and kernels of two streams on ONE queue run one after the other -- worse, a
# stream that mostly holds waits (the gradient buckets' communication stream) parks its barrier packets in front of
# whatever shares its queue.  Which streams collide depends on how many streams the process group, RCCL and torch used
# before (scripts/stream_queues.py prints the map): with the data-parallel engine attached the third stream used to land
# on the communication stream's queue (+3 ms per step).  So a stream is PICKED: candidates are bound by a first launch,
# then a short spin kernel on each stream already in use must be overtaken by a kernel on the candidate.
PICK_STREAMS = os.environ.get("FI_PICK_STREAMS", "1") != "0"       # A/B switch
_CANDIDATES = {}
_IN_USE = {}
_RETIRED = {}       # device -> streams stream_report(repick=True) replaced: never handed out again (holders of the old
                    # reference -- events, captured graphs -- may still use them), kept alive for the process
_FRESH = {}         # device -> extra candidates created by a repick (created once, not four per failing entry per call)


def _overtakes(a, b, probe):
    """True when a kernel on stream b finishes before a ~1 ms spin that was launched on stream a just before it."""
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(a):
        e0.record(a)
        torch.cuda._sleep(600000)
        ea.record(a)
    with torch.cuda.stream(b):
        probe.add_(1.0)
        eb.record(b)
    ea.synchronize()
    eb.synchronize()
    return e0.elapsed_time(eb) < 0.5 * e0.elapsed_time(ea)


def pick_stream(device=None):
    """A new stream that runs concurrently with the current stream and with every stream picked before (when the
    process has a hardware queue left for it; the best candidate otherwise)."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if not PICK_STREAMS or torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream(device=dev)
    with torch.cuda.device(dev):
        torch.cuda.synchronize(dev)
        probe = torch.zeros(64, device="cuda:%d" % dev)
        cands = _CANDIDATES.get(dev)
        if cands is None:
            cands = _CANDIDATES[dev] = [torch.cuda.Stream(device=dev) for _ in range(8)]
            for c in cands:                       # first use binds the stream to its hardware queue (and may create it)
                with torch.cuda.stream(c):
                    probe.add_(1.0)
            torch.cuda.synchronize(dev)
        used = _IN_USE.setdefault(dev, [])
        busy = [torch.cuda.current_stream(dev)] + used
        best, best_hits = None, -1
        retired = _RETIRED.get(dev, [])
        for c in cands + _FRESH.get(dev, []):
            if any(c is u for u in used) or any(c is r for r in retired):
                continue
            hits = sum(1 for a in busy if _overtakes(a, c, probe))
            if hits > best_hits:
                best, best_hits = c, hits
            if hits == len(busy):
                break
        torch.cuda.synchronize(dev)
        if best is None:
            best = torch.cuda.Stream(device=dev)
        used.append(best)
        return best


def side_stream(device=None):
    """The second stream of run_on_side_stream (picked on first use)."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    side = _SIDE2.get(dev)
    if side is None:
        side = _SIDE2[dev] = pick_stream(dev)
        _ROLE[(dev, id(side))] = "second"
    return side


def side_stream3(device=None):
    """The third stream (run_on_side_stream(after=...)), picked on first use."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    side = _SIDE3.get(dev)
    if side is None:
        side = _SIDE3[dev] = pick_stream(dev)
        _ROLE[(dev, id(side))] = "third"
    return side


_ROLE = {}          # (device, id(stream)) -> "second" | "third" | "communication" | ...


def name_stream(stream, role, device=None):
    """Give a picked stream a name for stream_report (the gradient buckets call this for their communication stream)."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    _ROLE[(dev, id(stream))] = role


def stream_report(device=None, repick=False):
    """The measured concurrency of the streams this process picked on `device`, in pick order: for every stream, whether a
    kernel on it overtakes a spin on the current stream and on every stream picked before it (the property pick_stream
    selected it for).  A process group created later, RCCL's own streams at its first collective, or another library can
    take hardware queues and change the map: with repick=True a stream that lost the property is REPLACED by a candidate
    that has it (the second / third stream caches are updated; the returned "replaced" maps id(old) -> new stream, so an
    owner like GradientBuckets can follow).  Host-synchronising (a handful of ~1 ms spins): for start-up and the first
    step boundary after the first real collective, not for the step.
    A replaced stream is RETIRED: it is never a candidate again (someone may still hold it), and blocks the caching
    allocator handed out on it stay in its pool (a one-off cost of the few tensors of the first step).  Call this before
    any graph capture and before caching a side stream or an event recorded on one (dev_roi.big_done, hipGraphs):
    consumers that cached the old stream keep working on it, but no longer next to the others."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    out = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "unset (runtime default 4)"), "pick_streams": PICK_STREAMS,
           "streams": [], "replaced": {}}
    used = _IN_USE.get(dev, [])
    if not used or torch.cuda.is_current_stream_capturing():
        return out
    with torch.cuda.device(dev):
        torch.cuda.synchronize(dev)
        probe = torch.zeros(64, device="cuda:%d" % dev)
        cur = torch.cuda.current_stream(dev)
        for i in range(len(used)):
            u = used[i]
            busy = [cur] + used[:i]
            ok = [bool(_overtakes(a, u, probe)) for a in busy]
            role = _ROLE.get((dev, id(u)), "picked-%d" % i)
            rec = {"role": role, "concurrent_with_earlier": all(ok), "detail": ok}
            if repick and not all(ok) and PICK_STREAMS:
                best, best_hits = None, sum(ok)
                fresh = _FRESH.get(dev)
                if fresh is None:
                    fresh = _FRESH[dev] = [torch.cuda.Stream(device=dev) for _ in range(4)]
                retired = _RETIRED.setdefault(dev, [])
                for c in _CANDIDATES.get(dev, []) + fresh:
                    if any(c is x for x in used) or any(c is x for x in retired):
                        continue
                    with torch.cuda.stream(c):
                        probe.add_(1.0)                   # binds a fresh stream to its hardware queue
                    hits = sum(1 for a in busy if _overtakes(a, c, probe))
                    if hits > best_hits:
                        best, best_hits = c, hits
                    if hits == len(busy):
                        break
                if best is not None:
                    out["replaced"][id(u)] = best
                    retired.append(u)
                    used[i] = best
                    _ROLE[(dev, id(best))] = role
                    if _SIDE2.get(dev) is u:
                        _SIDE2[dev] = best
                    if _SIDE3.get(dev) is u:
                        _SIDE3[dev] = best
                    rec["repicked"] = True
                    rec["concurrent_with_earlier"] = best_hits == len(busy)
            out["streams"].append(rec)
        torch.cuda.synchronize(dev)
    return out


def _mark_stream(o, stream):
    """record_stream on every CUDA tensor in o (nested tuples / lists): a tensor that a stream OTHER than the one it
    was allocated on reads must be marked, or the caching allocator may hand its block to the next tensor of the
    allocating stream the moment Python drops it -- while the reader's kernels are still queued."""
    if torch.is_tensor(o):
        if o.is_cuda:
            o.record_stream(stream)
    elif isinstance(o, (tuple, list)):
        for e in o:
            _mark_stream(e, stream)


def run_on_side_stream(fn, *args, after=None, reads=()):
    """Run fn(*args) (network-independent small kernels, e.g. RPN target generation) on a second stream so that
    its launch-latency-bound kernels interleave with the convolutions of the current stream.  Returns a
    function that makes the current stream wait for the result and returns it (a tuple/list of tensors or a
    tensor).  Inputs must already be complete on the current stream when this is called -- or, with `after` (an
    event recorded on the current stream EARLIER), at that event: the work then starts there, next to whatever the
    current stream has queued behind the event (a third stream, so that it does not queue behind run_on_side_stream
    work either).  Tensor arguments, and the tensors a closure reads (`reads`), are marked as used on the side stream:
    autograd keeps some of them for the backward pass of the side stream's ops and frees them, at CPU time, when it
    enqueues that op -- possibly before the side stream has run it."""
    dev = torch.cuda.current_device()
    cur = torch.cuda.current_stream(dev)
    side = side_stream(dev) if after is None else side_stream3(dev)
    if after is None:
        side.wait_stream(cur)
    else:
        side.wait_event(after)
    _mark_stream(args, side)
    _mark_stream(reads, side)
    with torch.cuda.stream(side):
        out = fn(*args)
    done = torch.cuda.Event()
    done.record(side)

    def wait():
        now = torch.cuda.current_stream(dev)
        now.wait_event(done)
        _mark_stream(out, now)          # allocated from the side stream's pool, consumed here
        return out
    wait.out, wait.done = out, done          # for consumers on the SAME side stream (ordered there): no wait needed
    return wait


def prof_enable(on=True):
    load().fi_prof_enable(1 if on else 0)


def prof_reset():
    load().fi_prof_reset()


def prof_get(name):
    n = ctypes.c_int(0)
    ms = ctypes.c_float(0.0)
    check(load().fi_prof_get(KERNEL_IDS[name], ctypes.byref(n), ctypes.byref(ms)), "fi_prof_get")
    return n.value, ms.value
