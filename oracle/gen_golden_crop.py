"""Golden vectors for a1 (RoIAlign forward = `CropAndResizePerBox`) produced by the REFERENCE'S OWN C compiled here.
TEST INFRASTRUCTURE; build container only (reads /root/reference, which does not exist on the GPU box).

`lib/roi_align/src/crop_and_resize.c` as a file needs <TH/TH.h> (PyTorch 0.3, absent) -- but its lines 2-112, the
includes of <stdio.h>/<math.h> and the whole of `CropAndResizePerBox` (:6-112), touch no TH symbol.  This script

  1. copies exactly those lines, unmodified, into a temporary directory OUTSIDE the repository (nothing of the
     reference's text or object code ever enters /root/repo; the directory is deleted at exit),
  2. compiles them as they are: `gcc -std=gnu99 -O2 -fopenmp -fPIC -shared -include stdlib.h` (`-include stdlib.h`
     only declares the `exit` the function calls; no stand-in header, no edit),
  3. calls `CropAndResizePerBox` through ctypes on the seeded inputs of `tests/helpers.golden_crop_cases`
     (adversarial boxes: inside / straddling / grid-aligned / last row / degenerate / flipped / outside / whole image;
     every crop size of tests/test_gpu_crop.py; both extrapolation values; 2x2 and 1x5 maps; the north-star
     512 x 256 x 7x7 and 14x14), and
  4. writes tests/golden/crop_fwd.npz: for the small cases the full fp32 output; for every case the SHA-256 of the
     output bytes (bit-exact comparison at sizes that would not fit a fixture).  Inputs are not stored: the tests
     regenerate them from the same seeds through the same helper.

The compile flags do not pin -ffp-contract: on x86-64 without -march flags gcc emits no FMA, so every multiply and add
is rounded separately, which is what oracle/Makefile forces for the restatement.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_crop.py   ->  tests/golden/crop_fwd.npz
"""
import atexit
import ctypes
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
from helpers import golden_crop_cases  # noqa: E402

REF = os.environ.get("FI_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "lib", "roi_align", "src", "crop_and_resize.c")
FIRST, LAST = 2, 112            # <stdio.h>, <math.h>, CropAndResizePerBox; line 1 is the TH include
OUT = os.path.join(HERE, "..", "tests", "golden", "crop_fwd.npz")


def build_reference():
    tmp = tempfile.mkdtemp(prefix="fi_ref_crop_")
    assert not os.path.abspath(tmp).startswith(os.path.abspath(os.path.join(HERE, ".."))), tmp
    atexit.register(shutil.rmtree, tmp, True)
    with open(SRC) as f:
        lines = f.readlines()
    assert "TH/TH.h" in lines[0] and "void CropAndResizePerBox(" in lines[5], "reference layout changed"
    assert lines[LAST - 1].strip() == "}" and "".join(lines[LAST:LAST + 3]).strip().startswith("void crop_and_resize_forward")
    body = "".join(lines[FIRST - 1:LAST])
    assert "TH" not in body.replace("THE", "")
    c = os.path.join(tmp, "per_box.c")
    with open(c, "w") as f:
        f.write(body)
    so = os.path.join(tmp, "per_box.so")
    subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-fopenmp", "-fPIC", "-shared", "-include", "stdlib.h",
                           "-o", so, c, "-lm"])
    L = ctypes.CDLL(so)
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    L.CropAndResizePerBox.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, ip,
                                      ctypes.c_int, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int, ctypes.c_float]
    L.CropAndResizePerBox.restype = None
    return L


def reference_crop(L, image, boxes, ind, ch, cw, extrap):
    """= what `crop_and_resize_forward` (crop_and_resize.c:115-154) does around the per-box function: a zeroed
    [N, C, ch, cw] output, one call over all boxes."""
    image = np.ascontiguousarray(image, np.float32)
    boxes = np.ascontiguousarray(boxes, np.float32)
    ind = np.ascontiguousarray(ind, np.int32)
    B, C, H, W = image.shape
    N = boxes.shape[0]
    out = np.zeros((N, C, ch, cw), np.float32)
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    L.CropAndResizePerBox(image.ctypes.data_as(fp), B, C, H, W, boxes.ctypes.data_as(fp), ind.ctypes.data_as(ip),
                          0, N, out.ctypes.data_as(fp), ch, cw, ctypes.c_float(extrap))
    return out


def gen():
    L = build_reference()
    out = {}
    n_el = 0
    for name, keep, image, boxes, ind, ch, cw, extrap in golden_crop_cases():
        got = reference_crop(L, image, boxes, ind, ch, cw, extrap)
        n_el += got.size
        out["sha256/" + name] = np.frombuffer(hashlib.sha256(got.tobytes()).digest(), np.uint8)
        if keep:
            out["full/" + name] = got
    np.savez_compressed(OUT, **out)
    print("wrote %s: %d cases, %d full, %.1f M elements through the reference's CropAndResizePerBox"
          % (OUT, sum(k.startswith("sha256/") for k in out), sum(k.startswith("full/") for k in out), n_el / 1e6))


if __name__ == "__main__":
    gen()
