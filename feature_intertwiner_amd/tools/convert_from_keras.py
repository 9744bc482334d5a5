"""Keras (matterport Mask R-CNN) weights -> this package's state dict: the counterpart of
tools/convert_from_keras.py:18-110 of the reference, which the authors used to import the COCO-pretrained
Keras model (`mask_rcnn_coco.h5`).

The reference rewrites names with ~140 ordered substring replacements over a flat script; here the
mapping is a small grammar over the Keras layer names (`res4f_branch2b` -> stage 4, block 5, conv 2):

    conv1 / bn_conv1                      fpn.C1.0 / fpn.C1.1
    res<S><b>_branch2<abc> / bn...        fpn.C<S>.<block>.conv<123> / .bn<123>
    res<S>a_branch1 / bn<S>a_branch1      fpn.C<S>.0.downsample.0 / .1
    fpn_c<L>p<L> / fpn_p<L>               fpn.P<L>_conv1 / fpn.P<L>_conv2.1
    rpn_conv_shared, rpn_class_raw, rpn_bbox_pred      rpn.conv_shared / conv_class / conv_bbox
    mrcnn_class_{conv,bn}<i>, mrcnn_class_logits, mrcnn_bbox_fc   classifier.* / linear_class / linear_bbox
    mrcnn_mask_{conv,bn}<i>, mrcnn_mask_deconv, mrcnn_mask        mask.* / mask.conv5
    kernel:0 / bias:0 / gamma:0 / beta:0 / moving_mean:0 / moving_variance:0
                                          weight / bias / weight / bias / running_mean / running_var

Layouts: 4-D kernels (kh, kw, in, out) -> (out, in, kh, kw) -- which also turns a Keras Conv2DTranspose
kernel (kh, kw, out, in) into torch's ConvTranspose2d (in, out, kh, kw); 2-D dense kernels (in, out) ->
(out, in).  `convert(named_arrays)` works on any {"<layer>.<weight>:0": ndarray} mapping; `read_h5`
needs h5py, which this image does not ship (the CLI says so instead of failing obscurely).
tests/test_checkpoint_formats.py holds the result to what the reference's own script produces.
"""
import collections
import re

import numpy as np
import torch

_WEIGHT = {"kernel:0": "weight", "bias:0": "bias", "gamma:0": "weight", "beta:0": "bias",
           "moving_mean:0": "running_mean", "moving_variance:0": "running_var"}
_FIXED = {"conv1": "fpn.C1.0", "bn_conv1": "fpn.C1.1", "rpn_conv_shared": "rpn.conv_shared",
          "rpn_class_raw": "rpn.conv_class", "rpn_bbox_pred": "rpn.conv_bbox",
          "mrcnn_class_logits": "classifier.linear_class", "mrcnn_bbox_fc": "classifier.linear_bbox",
          "mrcnn_mask_deconv": "mask.deconv", "mrcnn_mask": "mask.conv5"}
_RES = re.compile(r"^(res|bn)([2-5])([a-z])_branch(1|2[abc])$")
_FPN = re.compile(r"^fpn_(c([2-5])p\2|p([2-5]))$")
_HEAD = re.compile(r"^mrcnn_(class|mask)_(conv|bn)([1-4])$")


def torch_module_name(layer):
    """Keras layer name -> module path in MaskRCNN (None: not part of the detector, e.g. optimiser slots)."""
    if layer in _FIXED:
        return _FIXED[layer]
    m = _RES.match(layer)
    if m:
        kind, stage, blk, branch = m.groups()
        block = ord(blk) - ord("a")
        if branch == "1":
            return "fpn.C%s.%d.downsample.%d" % (stage, block, 0 if kind == "res" else 1)
        return "fpn.C%s.%d.%s%d" % (stage, block, "conv" if kind == "res" else "bn", "abc".index(branch[1]) + 1)
    m = _FPN.match(layer)
    if m:
        return "fpn.P%s_conv1" % m.group(2) if m.group(2) else "fpn.P%s_conv2.1" % m.group(3)
    m = _HEAD.match(layer)
    if m:
        return "%s.%s%s" % ("classifier" if m.group(1) == "class" else "mask", m.group(2), m.group(3))
    return None


def convert(named_arrays, strict=True):
    """{'<keras layer>.<weight>:0': ndarray} -> OrderedDict of torch tensors under this package's names."""
    out = collections.OrderedDict()
    for key, arr in named_arrays.items():
        layer, wname = key.split(".", 1)
        mod = torch_module_name(layer)
        if mod is None or wname not in _WEIGHT:
            if strict:
                raise KeyError("no counterpart for Keras weight %r" % key)
            continue
        a = np.asarray(arr)
        if a.ndim == 4:
            a = a.transpose(3, 2, 0, 1)
        elif a.ndim == 2:
            a = a.transpose(1, 0)
        out[mod + "." + _WEIGHT[wname]] = torch.from_numpy(np.ascontiguousarray(a))
    return out


def read_h5(path):
    """Flatten a Keras weight file into {'<layer>.<weight>:0': ndarray} (tools/convert_from_keras.py:24-30)."""
    try:
        import h5py
    except ImportError as e:
        raise ImportError("reading Keras .h5 files needs h5py, which is not installed in this image; "
                          "convert(named_arrays) works on any mapping of arrays") from e
    out = collections.OrderedDict()
    with h5py.File(path, mode="r") as f:
        for _, group in f.items():
            for layer_name, layer in group.items():
                for weight_name, weight in layer.items():
                    out[layer_name + "." + weight_name] = np.asarray(weight)
    return out


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Convert keras-mask-rcnn weights to a state dict for this package")
    ap.add_argument("--keras_model", required=True)
    ap.add_argument("--pytorch_model", required=True)
    args = ap.parse_args(argv)
    torch.save(convert(read_h5(args.keras_model)), args.pytorch_model)


if __name__ == "__main__":
    main()
