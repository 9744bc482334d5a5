"""Import helpers for the golden generators: make /root/reference importable in the build
container.  TEST INFRASTRUCTURE ONLY (used by oracle/gen_golden_*.py, never by the product).

The reference targets PyTorch 0.3 + CUDA: it imports `past.builtins` (python-future, absent
here) and its three cffi `_ext` packages (never built) at module import time, and calls
`.cuda()` unconditionally.  The placeholders below let the *Python* modules import; they
contain no behaviour (empty modules, `basestring = str`, `.cuda()` = identity), so any
function that would actually reach a native extension fails.  Only pure-Python/torch code of
the reference is executed by the generators (anchors, box arithmetic, nn.Module stacks, OT)."""
import os
import sys
import types

import torch

REF = os.environ.get("FI_REFERENCE", "/root/reference")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    _stub("past")
    _stub("past.builtins", basestring=str)
    for pkg, leaf in (("lib.roi_align._ext", "crop_and_resize"), ("lib.nms._ext", "nms"),
                      ("lib.roi_pooling._ext", "roi_pooling")):
        p = _stub(pkg)
        p.__path__ = []
        setattr(p, leaf, _stub(pkg + "." + leaf))
