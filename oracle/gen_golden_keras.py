"""Golden vectors for the Keras-h5 -> state-dict weight converter, produced by RUNNING the reference's
tools/convert_from_keras.py (a flat script) under runpy.

Runs only in the build container (needs /root/reference).  The script reads an .h5 file with h5py and
writes with torch.save; neither a Keras checkpoint nor h5py exists here, so the two I/O calls are given
in-memory doubles that carry no conversion logic: `h5py.File(path)` returns the nested mapping
{group: {layer: {weight_name: object with .value}}} built from tests/helpers.keras_named_arrays, and
`torch.save` hands the finished state dict back.  Everything between -- the ordered name rewriting
(:32-99) and the HWIO->OIHW / (in,out)->(out,in) transposes (:101-108) -- is the reference's own code.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_keras.py
writes tests/golden/keras_convert.json: for resnet50 and resnet101, [torch name, shape, sha256/16 of the
float32 bytes] per tensor, in the reference's output order.
"""
import hashlib
import json
import os
import runpy
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
from helpers import keras_named_arrays  # noqa: E402

REF = os.environ.get("FI_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "..", "tests", "golden", "keras_convert.json")
sys.dont_write_bytecode = True


class _Weight(object):
    def __init__(self, a):
        self.value = a


def run(arch):
    arrays = keras_named_arrays(arch)
    groups = {}
    for k, a in arrays.items():                      # one h5 group per layer, as Keras writes them
        layer, wname = k.split(".", 1)
        groups.setdefault(layer, {}).setdefault(layer, {})[wname] = _Weight(a)
    fake = types.ModuleType("h5py")
    fake.File = lambda path, mode="r": groups
    sys.modules["h5py"] = fake
    captured = {}
    real_save = torch.save
    torch.save = lambda obj, path: captured.setdefault("sd", obj)
    argv = sys.argv
    sys.argv = ["convert_from_keras.py", "--keras_model", "in.h5", "--pytorch_model", "out.pth"]
    try:
        runpy.run_path(os.path.join(REF, "tools", "convert_from_keras.py"), run_name="__main__")
    finally:
        torch.save = real_save
        sys.argv = argv
        del sys.modules["h5py"]
    return [[k, list(v.shape), hashlib.sha256(v.numpy().astype(np.float32).tobytes()).hexdigest()[:16]]
            for k, v in captured["sd"].items()]


if __name__ == "__main__":
    out = {arch: run(arch) for arch in ("resnet50", "resnet101")}
    with open(OUT, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print({a: len(v) for a, v in out.items()}, out["resnet50"][:3], os.path.getsize(OUT))
