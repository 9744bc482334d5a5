"""CPU: the product's static-shape meta loss (feature_intertwiner_amd/intertwiner.meta_loss +
FeatureBuffer) for the choices that are plain tensor arithmetic (l2 / l1 / kl) against the goldens
produced by RUNNING the reference's MaskRCNN.meta_loss (oracle/gen_golden_meta.py), over 4 consecutive
steps incl. one without small objects; BUFFER_SIZE > 1 against the oracle restatement; gradients of
the static form vs the reference's index-then-reduce form.  The 'ot' choice needs the HIP Sinkhorn
kernel and is covered by tests/test_gpu_meta.py."""
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from helpers import golden_meta_inputs, golden_meta_instances

K, F = 11, 1024
ACT = dict(l2="sigmoid", l1="sigmoid", kl="softmax", ot="relu")


def _cfg(choice, inst=False):
    return NS(DEV=NS(LOSS_CHOICE=choice, INST_LOSS=inst))


@pytest.mark.parametrize("choice", ["l2", "l1", "kl"])
def test_meta_loss_matches_reference_goldens(golden_dir, choice):
    from feature_intertwiner_amd.intertwiner import FeatureBuffer, meta_loss
    gold = np.load(os.path.join(golden_dir, "meta_loss.npz"))
    buf = FeatureBuffer(1, F, K, "cpu")
    for step in range(4):
        inp = [torch.from_numpy(a) for a in golden_meta_inputs(step, K, F, activation=ACT[choice])]
        got = float(meta_loss(_cfg(choice), buf, None, inp + [None, None]))
        exp = float(gold["%s_loss_%d" % (choice, step)][0])
        assert abs(got - exp) <= 2e-5 * abs(exp) + 1e-9, (choice, step, got, exp)
        assert np.array_equal(buf.buffer_cnt.numpy(), gold["%s_buffer_cnt_%d" % (choice, step)])
        if choice == "l2":
            assert np.allclose(buf.buffer.numpy(), gold["l2_buffer_%d" % step], rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("choice", ["l2", "l1"])
def test_inst_loss_matches_reference_goldens(golden_dir, choice):
    from feature_intertwiner_amd.intertwiner import FeatureBuffer, meta_loss
    gold = np.load(os.path.join(golden_dir, "meta_loss.npz"))
    buf = FeatureBuffer(1, F, K, "cpu")
    for step in range(2):
        inp = [torch.from_numpy(a) for a in golden_meta_inputs(step, K, F, activation=ACT[choice])]
        rows, gt = golden_meta_instances(step, 48, K, F, activation=ACT[choice])
        got = float(meta_loss(_cfg(choice, True), buf, None, inp + [torch.from_numpy(rows), torch.from_numpy(gt).float()]))
        exp = float(gold["inst_%s_loss_%d" % (choice, step)][0])
        assert abs(got - exp) <= 2e-5 * abs(exp), (choice, step, got, exp)


def test_fifo_buffer_matches_oracle(oracle):
    from feature_intertwiner_amd.intertwiner import FeatureBuffer, meta_loss
    buf = FeatureBuffer(3, F, K, "cpu")
    ml = oracle.MetaLoss("l2", 3, F, K)
    for step in (0, 1, 2, 3, 4, 5):             # step 2: no small objects -> no roll on either side
        a = golden_meta_inputs(step, K, F, activation="sigmoid")
        got = float(meta_loss(_cfg("l2"), buf, None, [torch.from_numpy(t) for t in a] + [None, None]))
        exp = float(ml(*a))
        assert abs(got - exp) <= 2e-5 * abs(exp) + 1e-9, (step, got, exp)
        assert np.array_equal(buf.buffer_cnt.numpy(), ml.buffer_cnt)
        assert np.array_equal(buf.buffer.numpy(), ml.buffer)


@pytest.mark.parametrize("choice", ["l2", "l1", "kl"])
def test_static_form_gradient_equals_indexed_form(choice):
    """d loss / d small_feat of the masked static-shape evaluation == the reference's
    'index the selected classes, then F.mse_loss / l1_loss / kl_div' (lib/model.py:187-204)."""
    import torch.nn.functional as Fn
    from feature_intertwiner_amd.intertwiner import EPS, FeatureBuffer, meta_loss
    buf = FeatureBuffer(1, F, K, "cpu")
    bf, bc, sf, sc = [torch.from_numpy(a) for a in golden_meta_inputs(0, K, F, activation=ACT[choice])]
    sf1 = sf.clone().requires_grad_(True)
    meta_loss(_cfg(choice), buf, None, [bf, bc, sf1, sc, None, None]).backward()
    sf2 = sf.clone().requires_grad_(True)
    s = (sf2 * sc).sum(0).sum(0) / (sc.sum(0).sum(0) + EPS)
    cnt = sc.sum(0).sum(0).view(-1).clone()
    cnt[0] = 0
    idx = torch.nonzero((cnt > 0) & (buf.buffer_cnt.view(-1) > 0)).view(-1)
    SMALL, BIG = s[:, idx].t(), buf.buffer[0][:, idx].t()
    ref = {"l2": lambda: Fn.mse_loss(SMALL, BIG), "l1": lambda: Fn.l1_loss(SMALL, BIG),
           "kl": lambda: Fn.kl_div(torch.log(SMALL), BIG, reduction="mean")}[choice]()
    ref.backward()
    assert torch.allclose(sf1.grad, sf2.grad, rtol=1e-4, atol=1e-9)
    assert sf1.grad.abs().sum() > 0


def test_unsupported_switches_fail_loudly():
    from feature_intertwiner_amd.config import make_config
    from feature_intertwiner_amd.sub_module import Dev
    for flag in ("DIS_UPSAMPLER", "ASSIGN_BOX_ON_ALL_SCALE"):
        cfg = make_config("resnet50", 128, 1, 16)
        setattr(cfg.DEV, flag, True)
        with pytest.raises(NotImplementedError):
            Dev(cfg, 256)
