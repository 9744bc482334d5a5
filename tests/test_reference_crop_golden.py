"""a1 pinned by reference execution: tests/golden/crop_fwd.npz holds what the reference's own C
`CropAndResizePerBox` (lib/roi_align/src/crop_and_resize.c:6-112), compiled unmodified in the build container by
oracle/gen_golden_crop.py, returned on the seeded inputs of helpers.golden_crop_cases.

  * CPU: the oracle's restatement (orc_crop_forward) equals it BIT FOR BIT on every case -> the oracle's bin
    assignment (floorf / ceilf / range test / lerp order) is the reference's, by execution, not by reading;
  * GPU: `fi_crop_and_resize_forward` -- NCHW (the drop-in layout) and channels-last maps -- equals it bit for bit.
"""
import hashlib
import os

import numpy as np
import pytest

from helpers import golden_crop_cases


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, "crop_fwd.npz"))


def _check(name, got, G):
    got = np.ascontiguousarray(got, np.float32)
    if "full/" + name in G.files:
        exp = G["full/" + name]
        assert got.shape == exp.shape, name
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), name
    digest = np.frombuffer(hashlib.sha256(got.tobytes()).digest(), np.uint8)
    assert np.array_equal(digest, G["sha256/" + name]), name


def test_fixture_covers_every_case(golden_dir):
    G = _golden(golden_dir)
    names = [c[0] for c in golden_crop_cases()]
    assert len(names) == len(set(names)) == 84
    assert sorted("sha256/" + n for n in names) == sorted(k for k in G.files if k.startswith("sha256/"))
    assert sum(k.startswith("full/") for k in G.files) == 30


def test_oracle_equals_reference_c_bit_for_bit(oracle, golden_dir):
    G = _golden(golden_dir)
    n = 0
    for name, _, image, boxes, ind, ch, cw, extrap in golden_crop_cases():
        _check(name, oracle.crop_and_resize_forward(image, boxes, ind, ch, cw, extrap), G)
        n += 1
    assert n == 84


@pytest.mark.gpu
@pytest.mark.parametrize("channels_last", [False, True])
def test_hip_equals_reference_c_bit_for_bit(golden_dir, channels_last):
    import torch
    from feature_intertwiner_amd.roi_align.crop_and_resize import CropAndResizeFunction
    G = _golden(golden_dir)
    dev = "cuda:0"
    for name, _, image, boxes, ind, ch, cw, extrap in golden_crop_cases():
        x = torch.from_numpy(image).to(dev)
        if channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        out = CropAndResizeFunction(ch, cw, extrap)(x, torch.from_numpy(boxes).to(dev), torch.from_numpy(ind).to(dev))
        torch.cuda.synchronize()
        _check(name, out.contiguous().cpu().numpy(), G)


@pytest.mark.gpu
def test_c_abi_entry_point_equals_reference_c(golden_dir):
    """The raw launcher a maintainer binds (include/fi_capi.h: fi_crop_and_resize_forward), not the Python operator."""
    import torch
    from feature_intertwiner_amd import _lib
    L = _lib.load()
    G = _golden(golden_dir)
    dev = "cuda:0"
    for name, _, image, boxes, ind, ch, cw, extrap in golden_crop_cases():
        if not (name.startswith("northstar") or name.startswith("s1_") or name.startswith("c70")):
            continue
        x = torch.from_numpy(image).to(dev)
        tb, ti = torch.from_numpy(boxes).to(dev), torch.from_numpy(ind).to(dev)
        B, C, H, W = image.shape
        N = boxes.shape[0]
        crops = torch.empty((N, C, ch, cw), device=dev)
        status = torch.zeros(1, device=dev, dtype=torch.int32)
        _lib.check(L.fi_crop_and_resize_forward(_lib.ptr(x), _lib.ptr(tb), _lib.ptr(ti), N, B, C, H, W, ch, cw,
                                                float(extrap), _lib.ptr(crops), _lib.ptr(status),
                                                _lib.current_stream()), "crop")
        torch.cuda.synchronize()
        assert int(status.item()) == 0
        _check(name, crops.cpu().numpy(), G)
