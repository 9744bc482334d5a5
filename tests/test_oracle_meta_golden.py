"""CPU: the NumPy restatement of MaskRCNN.meta_loss / the history buffer (oracle.MetaLoss) against
tests/golden/meta_loss.npz, which oracle/gen_golden_meta.py produced by running the reference's own
lib/model.py:143-224 -- this pins the oracle the GPU parity test (tests/test_gpu_meta.py) uses."""
import os

import numpy as np
import pytest

from helpers import golden_meta_inputs, golden_meta_instances, ot_full_weights

K, F = 11, 1024
OT_ABS = 1e-5 * 0.7
ACT = dict(l2="sigmoid", l1="sigmoid", kl="softmax", ot="relu")


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "meta_loss.npz"))


def _ot():
    g_w, g_b, c_w, c_b = ot_full_weights(4321, F)
    return dict(g_w=g_w, g_b=g_b, c_w=c_w, c_b=c_b, epsilon=1.0, L=5)


def test_merge_feat_vec(oracle, gold):
    bf, bc, _, _ = golden_meta_inputs(0, K, F)
    m, c = oracle.merge_feat_vec(bf, bc)
    assert np.array_equal(c, gold["merge_cnt"])
    assert np.allclose(m, gold["merge_feat"], rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("choice", ["l2", "l1", "kl", "ot"])
def test_meta_loss_sequence(oracle, gold, choice):
    ml = oracle.MetaLoss(choice, 1, F, K, ot=_ot() if choice == "ot" else None)
    for step in range(4):
        got = np.asarray(ml(*golden_meta_inputs(step, K, F, activation=ACT[choice]))).reshape(-1)
        exp = gold["%s_loss_%d" % (choice, step)]
        assert got.shape == exp.shape, (choice, step)
        if choice == "ot":
            # debiased OT = 2 T(x,y) - T(x,x) - T(y,y) is a 1e-3 difference of terms of size ~0.7:
            # the bar is 1e-5 relative to the TERMS, as in tests/test_oracle_ot_golden.py
            assert np.abs(got - exp).max() <= OT_ABS, (step, np.abs(got - exp).max())
        else:
            assert np.allclose(got, exp, rtol=2e-5, atol=1e-9), (choice, step, got, exp)
        assert np.array_equal(ml.buffer_cnt, gold["%s_buffer_cnt_%d" % (choice, step)])
        if choice in ("l2", "ot"):
            assert np.allclose(ml.buffer, gold["%s_buffer_%d" % (choice, step)], rtol=2e-6, atol=1e-7)
    # step 2 has no small objects: loss 0 and the buffer is left alone (lib/workflow.py:190-194)
    assert float(gold["%s_loss_2" % choice][0]) == 0.0
    assert np.array_equal(gold["%s_buffer_cnt_2" % choice], gold["%s_buffer_cnt_1" % choice])


@pytest.mark.parametrize("choice", ["l2", "l1", "ot"])
def test_inst_loss_sequence(oracle, gold, choice):
    ml = oracle.MetaLoss(choice, 1, F, K, inst_loss=True, ot=_ot() if choice == "ot" else None)
    for step in range(2):
        rows, gt = golden_meta_instances(step, 48, K, F, activation=ACT[choice])
        got = np.asarray(ml(*golden_meta_inputs(step, K, F, activation=ACT[choice]), rows, gt)).reshape(-1)
        exp = gold["inst_%s_loss_%d" % (choice, step)]
        assert got.shape == exp.shape
        if choice == "ot":
            assert np.abs(got - exp).max() <= OT_ABS
        else:
            assert np.allclose(got, exp, rtol=2e-5)


def test_fifo_buffer_rule(oracle):
    """BUFFER_SIZE > 1 (lib/model.py:159-166): slots shift, the newest step goes last, the comparison
    target is the count-weighted mean over the history.  (Unpinned: the reference's class selection
    does not execute for BUFFER_SIZE > 1, see oracle/gen_golden_meta.py.)"""
    ml = oracle.MetaLoss("l2", 3, F, K)
    seen = []
    for step in (0, 1, 3, 4, 5):
        bf, bc, sf, sc = golden_meta_inputs(step, K, F, activation="sigmoid")
        ml(bf, bc, sf, sc)
        seen.append(oracle.merge_feat_vec(bf, bc))
    for slot, (f, c) in zip(range(3), seen[-3:]):
        assert np.array_equal(ml.buffer[slot], f) and np.array_equal(ml.buffer_cnt[slot], c)
