"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly the symbols that
include/fi_capi.h declares; the Python mirror keeps the reference's names and call
shapes; the product path has no CPU fallback.  No kernel is launched here."""
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "fi_capi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fi_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from feature_intertwiner_amd import build
    path = build.build_hip()
    assert os.path.exists(path)
    nm = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    exported = set(re.findall(r" T (fi_[a-z0-9_]+)", nm))
    declared = _declared()
    assert len(declared) >= 18
    assert set(declared) <= exported, sorted(set(declared) - exported)
    # nothing but the declared API (and no torch types) leaks out with the fi_ prefix
    assert exported <= set(declared), sorted(exported - set(declared))
    # device code for gfx950 is embedded
    assert b"gfx950" in open(path, "rb").read()


def test_ctypes_binding_covers_the_header_and_loads():
    from feature_intertwiner_amd import _lib
    assert sorted(_lib.SIGNATURES.keys()) == _declared()
    L = _lib.load()
    assert L.fi_version().startswith(b"fi_hip")
    assert L.fi_nms_workspace_bytes(4, 6000) == 4 * 6000 * 94 * 8
    assert L.fi_nms_workspace_bytes(1, 64) == 64 * 8
    assert L.fi_prof_kernel_name(0) == b"crop_fwd_flat_kernel<7, 7, 8>" and L.fi_prof_kernel_name(45) == b"gemm_slab_reduce_kernel"


def test_argument_validation_without_a_gpu():
    """Invalid sizes are rejected on the host before any HIP call."""
    from feature_intertwiner_amd import _lib
    L = _lib.load()
    assert L.fi_crop_and_resize_forward(None, None, None, 1, 1, 1, 8, 8, 0, 7, 0.0, None, None, None) == -1
    assert b"crop size" in L.fi_last_error()
    assert L.fi_crop_and_resize_forward(None, None, None, 1, 1, 1, 8, 8, 65, 7, 0.0, None, None, None) == -3
    assert L.fi_sinkhorn_forward(None, None, 1, 300, 1, 1.0, 5, 0, None, None, None, None, None) == -3
    assert L.fi_sinkhorn_forward(None, None, 1, 16, 1, 1.0, 0, 0, None, None, None, None, None) == -1
    assert L.fi_nms_sorted(None, 1, 10, 3, 0.5, 0, 0, None, None, None, None) == -1
    assert L.fi_roi_pool_forward(None, None, 1, 1, 1, 4, 4, 0, 7, 1.0, None, None, None) == -1
    assert L.fi_class_mean_forward(None, None, 0, 8, 500, None, None, None, None) == -3
    assert L.fi_class_mean_workspace_bytes(2048, 1024, 81) == 32 * 81 * 1024 * 4
    # channels-last pyramid crop: level array required with several maps, crop bounded by the LDS tile
    import ctypes
    ptrs = (ctypes.c_void_p * 2)(16, 32)
    hs = (ctypes.c_int * 2)(8, 4)
    assert L.fi_pyramid_crop_forward_nhwc(ptrs, hs, hs, 2, 16, 16, None, 1, 1, 64, 7, 7, 0.0, 16, None) == -1
    assert b"level" in L.fi_last_error()
    assert L.fi_pyramid_crop_forward_nhwc(ptrs, hs, hs, 0, 16, 16, 16, 1, 1, 64, 7, 7, 0.0, 16, None) == -1
    # bf16-weight patch convolution and the optimiser step validate before touching the device
    assert L.fi_conv3x3_forward_bf16w(16, 16, None, None, None, 16, 1, 32, 8, 18, 128, 0, 0, None) == -3
    assert b"W % 4" in L.fi_last_error()
    assert L.fi_conv3x3_forward_bf16w(None, 16, None, None, None, 16, 1, 32, 8, 16, 128, 0, 0, None) == -1
    assert L.fi_conv1x1_forward_bf16w(16, 16, None, None, None, 16, 1, 96, 64, 128, 0, None) == -3     # Cin % 64
    assert L.fi_conv1x1_forward_bf16w(16, 16, None, None, None, 16, 1, 64, 50, 128, 0, None) == -3     # H*W % 4
    assert L.fi_conv1x1_forward_bf16w(None, 16, None, None, None, 16, 1, 64, 64, 128, 0, None) == -1
    assert L.fi_sgd_chunks(0) == 0 and L.fi_sgd_chunks(1) == 1 and L.fi_sgd_chunks(8193) == 2
    assert L.fi_sgd_clip_step(None, 0, 0, 1.0, None, None, None) == 0
    assert L.fi_sgd_clip_step(None, 3, 5, 1.0, None, None, None) == -1
    # conv: layouts are validated before anything is launched
    a = [16, 16, None, None, None, 16]
    assert L.fi_conv2d_forward(*a, 1, 32, 8, 8, 64, 3, 3, 1, 1, 1, 1, 0, 5, 0, 0, 0, None) == -1
    assert b"weight_layout" in L.fi_last_error()
    assert L.fi_conv2d_forward(*a, 1, 32, 8, 8, 64, 3, 3, 1, 1, 1, 1, 0, 1, 0, 0, 2, None) == -1
    assert b"output_layout" in L.fi_last_error()
    assert L.fi_conv2d_forward(*a, 1, 30, 8, 8, 64, 3, 3, 1, 1, 1, 1, 0, 1, 0, 0, 0, None) == -1      # tap-major needs Cin % 16
    assert L.fi_conv2d_forward(*a, 1, 32, 8, 8, 62, 3, 3, 1, 1, 1, 1, 0, 1, 0, 0, 1, None) == -1      # NHWC needs Cout % 4
    assert L.fi_bn_act_backward(16, 16, 16, None, None, None, 1, 8, 16, 1, 16, None, 16, None, None, 3, 0, None) == -1


def test_round3_entry_points_validate_and_gradient_handoffs_on_the_host():
    """The entry points added for the unscaled-gradient backward reject bad arguments before any HIP call, and the
    host-side hand-off logic (GradBox pairs, Gate claims) behaves on CPU tensors: nothing is claimed, nothing is given."""
    import torch
    from feature_intertwiner_amd import _lib, conv as C
    L = _lib.load()
    a = [16, 16, None, None, None, 32, 16]                   # x, w, bias, scale, residual, gate, y
    assert L.fi_conv2d_forward_gated(*a, 1, 32, 8, 8, 64, 1, 1, 1, 1, 0, 0, 0, 1, 0, 0, 1, None) == -1      # gate + NHWC output
    assert b"gate" in L.fi_last_error()
    assert L.fi_relu_mask(None, None, None, -1, None) == -1
    assert L.fi_relu_mask(None, None, None, 0, None) == 0
    assert L.fi_relu_mask(None, 16, 16, 8, None) == -1
    assert L.fi_bn_fold_grad(None, 16, 16, 16, 16, 16, 0.001, None, None, None, 8, 8, 1, 0, 0, None) == -1
    assert L.fi_bn_fold_grad(16, 16, 16, 16, 16, 16, 0.001, None, None, None, 0, 8, 1, 0, 0, None) == -1
    assert L.fi_bn_fold_batch(None, 0, 0, None) == 0 and L.fi_bn_fold_batch(None, 3, 64, None) == -1
    assert L.fi_sum2x2(16, 16, 1, 4, 5, None) == -1                                   # odd width
    assert L.fi_sum2x2(None, None, 0, 4, 4, None) == 0
    assert L.fi_maxpool3x3s2_forward(16, 16, 1, 8, 6, None) == -1                     # width % 4
    assert L.fi_maxpool3x3s2_backward(16, 16, 16, 1, 2, 8, 0, None) == -1             # height < 3
    assert L.fi_stride2_interleave_gated(None, None, None, None, None, None, None, 1, 4, 4, None) == -1
    # GradBox pairs: (compact, full) for the compact 1x1 / stride-2 path, one value alone, nothing -> None
    b1, b2 = C.GradBox(), C.GradBox()
    assert C._take_boxes(None) is None and C._take_boxes((b1, b2)) is None
    t = torch.ones(1, 2, 4, 4)
    cp = C._Compact(torch.ones(1, 2, 2, 2), (4, 4))
    b1.value = cp
    assert C._take_boxes((b1, b2)) is cp and b1.value is None
    b1.value, b2.value = cp, t
    got = C._take_boxes((b1, b2))
    assert isinstance(got, tuple) and got[0] is cp and got[1] is t and b1.value is None and b2.value is None
    b1.value = t
    assert C._take_boxes(b1) is t
    # a Gate is only ever claimed for a CUDA tensor that requires grad, and only a taker's box is given to
    x = torch.ones(1, 2, 4, 4, requires_grad=True)
    x._fi_gate = C.Gate()
    assert C._claim_gate(x, True) is False and not x._fi_gate.claimed
    assert C.GradBox().taker is False


def test_reference_shaped_python_surface():
    import inspect
    from feature_intertwiner_amd.roi_align.crop_and_resize import CropAndResizeFunction
    from feature_intertwiner_amd.roi_align.roi_align import RoIAlign
    from feature_intertwiner_amd.roi_pooling.functions.roi_pool import RoIPoolFunction
    from feature_intertwiner_amd.roi_pooling.modules.roi_pool import _RoIPooling
    from feature_intertwiner_amd.nms.nms_wrapper import nms
    from feature_intertwiner_amd.nms.pth_nms import pth_nms
    from feature_intertwiner_amd.OT_module import OptTrans
    assert list(inspect.signature(CropAndResizeFunction.__init__).parameters)[1:] == \
        ["crop_height", "crop_width", "extrapolation_value"]
    assert list(inspect.signature(RoIAlign.__init__).parameters)[1:] == \
        ["crop_height", "crop_width", "extrapolation_value", "transform_fpcoor"]
    assert list(inspect.signature(RoIPoolFunction.__init__).parameters)[1:] == \
        ["pooled_height", "pooled_width", "spatial_scale"]
    assert list(inspect.signature(_RoIPooling.__init__).parameters)[1:] == \
        ["pooled_height", "pooled_width", "spatial_scale"]
    assert list(inspect.signature(nms).parameters)[:2] == ["dets", "thresh"]
    assert list(inspect.signature(pth_nms).parameters)[:2] == ["dets", "thresh"]
    assert list(inspect.signature(OptTrans.__init__).parameters)[1:] == [
        "config", "ch_x", "spatial_x", "ch_y", "spatial_y", "epsilon", "L", "remove_bias", "C_form",
        "no_bp_P_L", "skip_critic"]
    import types
    cfg = types.SimpleNamespace(DEV=types.SimpleNamespace(OT_ONE_DIM_FORM="conv"))
    m = OptTrans(cfg, ch_x=1024)
    assert sorted(m.state_dict().keys()) == ["G_net.0.bias", "G_net.0.weight", "critic.0.bias", "critic.0.weight"]
    assert m.critic[0].weight.shape == (256, 1024, 3) and m.epsilon == 1.0 and m.L == 5
    m2 = OptTrans(cfg, ch_x=256, spatial_x=32, spatial_y=64)
    assert m2.two_dim and m2.G_net[0].stride == (2, 2) and m2.critic[3].out_channels == 64


def test_no_cpu_fallback():
    from feature_intertwiner_amd import _lib
    from feature_intertwiner_amd.roi_align.crop_and_resize import CropAndResizeFunction
    from feature_intertwiner_amd.nms.pth_nms import pth_nms
    from feature_intertwiner_amd.OT_module import sinkhorn_loss
    with pytest.raises(_lib.FiError):
        CropAndResizeFunction(7, 7)(torch.zeros(1, 1, 8, 8), torch.zeros(1, 4), torch.zeros(1, dtype=torch.int32))
    with pytest.raises(_lib.FiError):
        pth_nms(torch.zeros(4, 5), 0.5)
    with pytest.raises(_lib.FiError):
        sinkhorn_loss(torch.zeros(1, 4, 1), torch.zeros(1, 4, 1))


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "feature_intertwiner_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(d, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "libfi_oracle" not in text, f


def test_level_assignment_matches_oracle(oracle):
    import numpy as np
    from feature_intertwiner_amd.intertwiner import roi_level, merge_feat_vec
    rs = np.random.RandomState(0)
    y1, x1 = rs.uniform(0, 0.7, (2, 500))
    rois = np.stack([y1, x1, y1 + np.exp(rs.uniform(-6, -0.3, 500)), x1 + np.exp(rs.uniform(-6, -0.3, 500))], 1)
    rois = rois.astype(np.float32)
    got = roi_level(torch.from_numpy(rois), 1024 * 1024).numpy()
    exp = oracle.roi_level(rois, 1024 * 1024)
    # torch.log vs np.log may differ in the last ulp exactly at a .5 boundary: allow none here
    assert np.array_equal(got, exp)
    f = torch.rand(2, 3, 8, 5)
    c = torch.randint(0, 4, (2, 3, 1, 5)).float()
    m, cs = merge_feat_vec(f, c)
    assert torch.allclose(cs, c.sum((0, 1)))
    assert torch.allclose(m, (f * c).sum((0, 1)) / (cs + 1e-20))


def test_checkpoint_file_round_trip(tmp_path):
    """tools/utils.py:567-586 file layout; resume restores weights, counters and the history buffer."""
    import numpy as np
    import torch
    from feature_intertwiner_amd.checkpoint import load_model, save_model
    from feature_intertwiner_amd.config import make_config
    from feature_intertwiner_amd.model import MaskRCNN
    cfg = make_config(backbone="resnet50", image_size=128, batch_size=1, train_rois_per_image=16)
    torch.manual_seed(0)
    a = MaskRCNN(cfg)
    a.initialize_buffer(torch.device("cpu"))
    a.feature_buffer.buffer.normal_()
    a.feature_buffer.buffer_cnt.fill_(3)
    path = str(tmp_path / "mask_rcnn_ep_0002_iter_000123.pth")
    save_model(a, path, epoch=2, iter=123, loss_data={"x": [1.0]})
    raw = torch.load(path, weights_only=False)
    assert sorted(raw) == ['buffer', 'buffer_cnt', 'epoch', 'iter', 'loss_data', 'state_dict']
    assert isinstance(raw['buffer'], np.ndarray) and raw['buffer'].shape == (1, 1024, 81)
    torch.manual_seed(1)
    b = MaskRCNN(cfg)
    ep, it, loss_data, missing, unexpected = load_model(b, path)
    assert (ep, it) == (2, 124) and loss_data == {"x": [1.0]} and not missing and not unexpected
    for (k, v), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert k == k2 and torch.equal(v, v2)
    assert torch.equal(a.feature_buffer.buffer, b.feature_buffer.buffer)
    assert torch.equal(a.feature_buffer.buffer_cnt, b.feature_buffer.buffer_cnt)
    # bare state dict == pretrain model
    torch.save(a.state_dict(), path)
    assert load_model(MaskRCNN(cfg), path)[:2] == (1, 1)


def test_relu_boundary_evidence_tells_an_event_from_a_bug():
    """workflow._relu_boundary_evidence (advisor: a deviating rpn.conv_shared channel was classified as a ReLU-boundary
    event by footprint only): a sign disagreement between the dense kernel and the row form counts as an event only when
    the float64 pre-activation is within rounding of zero."""
    import types
    import torch
    from feature_intertwiner_amd.workflow import _relu_boundary_evidence
    g = torch.Generator().manual_seed(5)
    per_loc, H, W, Cin, Cout, R = 3, 4, 4, 2, 8, 6
    weight = torch.randn(Cout, Cin, 3, 3, generator=g)
    bias = torch.randn(Cout, generator=g)
    ws = weight.permute(0, 2, 3, 1).reshape(Cout, -1)
    patches = torch.randn(R, 9 * Cin, generator=g)
    # row 2, channel 5: a pre-activation at rounding distance from zero
    bias[5] = -float(patches[2].double() @ ws[5].double())
    z = patches @ ws.t() + bias
    anchor = torch.tensor([0, 7, 13, 20, 31, 47])             # pixel = anchor // per_loc
    img = torch.zeros(R, dtype=torch.long)
    valid = torch.ones(R, dtype=torch.bool)
    dense = torch.zeros(1, Cout, H, W)
    pix = anchor // per_loc
    dense[0, :, pix // W, pix % W] = torch.relu(z).t()
    rpn = types.SimpleNamespace(conv_shared=types.SimpleNamespace(weight=weight, bias=bias))
    probe = {"rows": (img, anchor, valid, per_loc), "patches": patches, "z_rows": z.clone(), "dense_y": [dense.clone()]}
    assert all(e["rows_disagreeing"] == 0 for e in _relu_boundary_evidence(probe, rpn, [5, 1]))
    # the event: the row form lands just above zero, the dense kernel on zero
    probe["z_rows"][2, 5] = 1e-9
    probe["dense_y"][0][0, 5, pix[2] // W, pix[2] % W] = 0.0
    ev = _relu_boundary_evidence(probe, rpn, [5])[0]
    assert ev["rows_disagreeing"] == 1 and ev["within_rounding"] and ev["max_abs_z_over_scale"] <= 16
    # a bug: the evaluations disagree where |z| is far from zero
    r = int(torch.argmax(z[:, 1].abs()))
    probe["z_rows"][r, 1] = -z[r, 1]
    ev = _relu_boundary_evidence(probe, rpn, [1])[0]
    assert ev["rows_disagreeing"] == 1 and not ev["within_rounding"]
