"""Generate golden vectors for the intertwiner meta loss and its history buffer by RUNNING THE
REFERENCE's `MaskRCNN.meta_loss` / `_merge_feat_vec` (lib/model.py:143-224).

Runs only in the build container (needs /root/reference, see oracle/_ref_import.py); writes
tests/golden/meta_loss.npz = the reference's outputs only (loss per step, buffer / buffer_cnt after
each step); the inputs are regenerated from seeds by tests/helpers.golden_meta_inputs.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_meta.py

How the reference method is executed: `meta_loss` only touches `self.buffer`, `self.buffer_cnt`,
`self.config`, `self.ot_loss` and two static methods, so it is called unbound on a namespace carrying
exactly those (building the whole MaskRCNN needs the never-built cffi extensions).  Two PyTorch-0.3
conventions are restored for the duration of the call, neither adds behaviour of its own:
  * comparisons return ByteTensor (uint8): the class selection (:180-181) adds two comparison
    results and tests `== 2`; with today's bool tensors `True + True` is `True`.
  * `numpy_scalar in tensor` (:173, INST_LOSS): today's Tensor.__contains__ rejects numpy values;
    they are converted to Python scalars first.
The OT choice runs under no_grad (the reference normalises in place).  Sequences:
BUFFER_SIZE == 1 (every shipped config, configs/10x/*.yaml) for l2 / l1 / kl / ot over 4 steps, step 2
without any small object; and DEV.INST_LOSS with int64 class ids (the float ids the reference's
Dev.forward produces cannot index a tensor under torch 2.x; the values are the same).
BUFFER_SIZE > 1 with INST_LOSS False does not execute in the reference (2-D nonzero at :181).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import _ref_import  # noqa: E402

_ref_import.install()
import lib.model as RM  # noqa: E402
from lib.OT_module import OptTrans  # noqa: E402
from helpers import golden_meta_inputs, golden_meta_instances, ot_full_weights  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")
K, F = 11, 1024
ACT = dict(l2="sigmoid", l1="sigmoid", kl="softmax", ot="relu")


class torch03:
    """PyTorch-0.3 conventions needed by lib/model.py:173-181 (see module docstring)."""

    def __enter__(self):
        self.gt, self.contains = torch.Tensor.__gt__, torch.Tensor.__contains__
        gt, contains = self.gt, self.contains
        torch.Tensor.__gt__ = lambda a, b: gt(a, b).to(torch.uint8)
        torch.Tensor.__contains__ = lambda a, v: contains(a, v.item() if isinstance(v, np.ndarray) else v)

    def __exit__(self, *exc):
        torch.Tensor.__gt__, torch.Tensor.__contains__ = self.gt, self.contains


def fake_self(choice, inst, buffer_size=1):
    cfg = types.SimpleNamespace(DEV=types.SimpleNamespace(INST_LOSS=inst, LOSS_CHOICE=choice, OT_ONE_DIM_FORM="conv"))
    s = types.SimpleNamespace(config=cfg, buffer=torch.zeros(buffer_size, F, K), buffer_cnt=torch.zeros(buffer_size, 1, K),
                              _merge_feat_vec=RM.MaskRCNN._merge_feat_vec,
                              _assign_from_buffer=RM.MaskRCNN._assign_from_buffer)
    if choice == "ot":
        ot = OptTrans(cfg, ch_x=F, epsilon=1.0, L=5)
        g_w, g_b, c_w, c_b = ot_full_weights(4321, F)
        ot.load_state_dict({"G_net.0.weight": torch.from_numpy(g_w), "G_net.0.bias": torch.from_numpy(g_b),
                            "critic.0.weight": torch.from_numpy(c_w), "critic.0.bias": torch.from_numpy(c_b)})
        s.ot_loss = ot
    return s


def main():
    out = {}
    T = torch.from_numpy
    # _merge_feat_vec on its own (static method, runs as is)
    bf, bc, _, _ = golden_meta_inputs(0, K, F)
    m, c = RM.MaskRCNN._merge_feat_vec(T(bf.copy()), T(bc.copy()))
    out["merge_feat"], out["merge_cnt"] = m.numpy(), c.numpy()
    for choice in ("l2", "l1", "kl", "ot"):
        me = fake_self(choice, False)
        for step in range(4):
            bf, bc, sf, sc = golden_meta_inputs(step, K, F, activation=ACT[choice])
            if float(sf.sum()) != 0:           # the guard of lib/workflow.py:190-194
                with torch03(), torch.no_grad():
                    loss = RM.MaskRCNN.meta_loss(me, [T(bf), T(bc), T(sf), T(sc), None, None])
                loss = loss.numpy().reshape(-1)
            else:
                loss = np.zeros(1, np.float32)
            out["%s_loss_%d" % (choice, step)] = loss
            out["%s_buffer_cnt_%d" % (choice, step)] = me.buffer_cnt.numpy().copy()
            if choice in ("l2", "ot"):
                out["%s_buffer_%d" % (choice, step)] = me.buffer.numpy().copy()
    for choice in ("l2", "l1", "ot"):
        me = fake_self(choice, True)
        for step in range(2):
            bf, bc, sf, sc = golden_meta_inputs(step, K, F, activation=ACT[choice])
            rows, gt = golden_meta_instances(step, 48, K, F, activation=ACT[choice])
            with torch03(), torch.no_grad():
                loss = RM.MaskRCNN.meta_loss(me, [T(bf), T(bc), T(sf), T(sc), T(rows), T(gt)])
            out["inst_%s_loss_%d" % (choice, step)] = loss.numpy().reshape(-1)
    np.savez_compressed(os.path.join(OUT, "meta_loss.npz"), **out)
    print({k: (v.shape, float(np.asarray(v).reshape(-1)[0])) for k, v in out.items() if "loss" in k})
    print("bytes:", os.path.getsize(os.path.join(OUT, "meta_loss.npz")))


if __name__ == "__main__":
    main()
