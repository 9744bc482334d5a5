"""GPU: the intertwiner meta loss on the device -- 'ot' (HIP Sinkhorn) and the tensor-arithmetic
choices -- against the goldens produced by running the reference's MaskRCNN.meta_loss, and the
model-level path (MaskRCNN.meta_loss inside 3 consecutive train steps) against the oracle
restatement replayed on the captured statistics."""
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from helpers import golden_meta_inputs, golden_meta_instances, ot_full_weights

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
K, F = 11, 1024
ACT = dict(l2="sigmoid", l1="sigmoid", kl="softmax", ot="relu")
# debiased OT = 2 T(x,y) - T(x,x) - T(y,y): a ~1e-3 difference of terms of size ~0.7.  The
# north star holds the fp32 OT loss to 1e-4 relative -- per TERM; on the combination that is
OT_ABS = 1e-4 * 0.7


def _cfg(choice, inst=False):
    return NS(DEV=NS(LOSS_CHOICE=choice, INST_LOSS=inst, OT_ONE_DIM_FORM="conv"))


def _ot_module(L=5):
    from feature_intertwiner_amd.OT_module import OptTrans
    ot = OptTrans(_cfg("ot"), ch_x=F, epsilon=1.0, L=L)
    g_w, g_b, c_w, c_b = ot_full_weights(4321, F)
    ot.load_state_dict({"G_net.0.weight": torch.from_numpy(g_w), "G_net.0.bias": torch.from_numpy(g_b),
                        "critic.0.weight": torch.from_numpy(c_w), "critic.0.bias": torch.from_numpy(c_b)})
    return ot.to(DEV)


@pytest.mark.parametrize("choice", ["ot", "l2", "l1", "kl"])
def test_meta_loss_sequence_vs_reference_goldens(golden_dir, choice):
    from feature_intertwiner_amd.intertwiner import FeatureBuffer, meta_loss
    gold = np.load(os.path.join(golden_dir, "meta_loss.npz"))
    buf = FeatureBuffer(1, F, K, DEV)
    ot = _ot_module() if choice == "ot" else None
    for step in range(4):
        inp = [torch.from_numpy(a).to(DEV) for a in golden_meta_inputs(step, K, F, activation=ACT[choice])]
        with torch.no_grad():
            got = float(meta_loss(_cfg(choice), buf, ot, inp + [None, None]))
        exp_v = gold["%s_loss_%d" % (choice, step)]
        exp = float(exp_v.mean())           # 'ot': the reference returns one value per selected class (Q10)
        if choice == "ot":
            assert abs(got - exp) <= OT_ABS, (step, got, exp)
        else:
            assert abs(got - exp) <= 2e-5 * abs(exp) + 1e-9, (choice, step, got, exp)
        assert np.array_equal(buf.buffer_cnt.cpu().numpy(), gold["%s_buffer_cnt_%d" % (choice, step)])
        if choice in ("l2", "ot"):
            assert np.allclose(buf.buffer.cpu().numpy(), gold["%s_buffer_%d" % (choice, step)], rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("choice", ["ot", "l2", "l1"])
def test_inst_loss_vs_reference_goldens(golden_dir, choice):
    from feature_intertwiner_amd.intertwiner import FeatureBuffer, meta_loss
    gold = np.load(os.path.join(golden_dir, "meta_loss.npz"))
    buf = FeatureBuffer(1, F, K, DEV)
    ot = _ot_module() if choice == "ot" else None
    for step in range(2):
        inp = [torch.from_numpy(a).to(DEV) for a in golden_meta_inputs(step, K, F, activation=ACT[choice])]
        rows, gt = golden_meta_instances(step, 48, K, F, activation=ACT[choice])
        with torch.no_grad():
            got = float(meta_loss(_cfg(choice, True), buf, ot, inp + [torch.from_numpy(rows).to(DEV),
                                                                       torch.from_numpy(gt).float().to(DEV)]))
        exp = float(gold["inst_%s_loss_%d" % (choice, step)].mean())
        assert abs(got - exp) <= (OT_ABS if choice == "ot" else 2e-5 * abs(exp)), (choice, step, got, exp)


def test_ot_per_class_values_vs_reference_goldens(golden_dir):
    """The per-class vector the reference returns (lib/model.py:207), class by class."""
    from feature_intertwiner_amd.intertwiner import EPS
    gold = np.load(os.path.join(golden_dir, "meta_loss.npz"))
    ot = _ot_module()
    bf, bc, sf, sc = golden_meta_inputs(0, K, F, activation="relu")
    T = lambda a: torch.from_numpy(a).to(DEV)
    s = (T(sf) * T(sc)).sum(0).sum(0) / (T(sc).sum(0).sum(0) + EPS)
    b = torch.from_numpy(gold["ot_buffer_0"][0]).to(DEV)
    cnt = sc.sum(0).sum(0).reshape(-1)
    idx = np.nonzero((cnt > 0) & (gold["ot_buffer_cnt_0"].reshape(-1) > 0))[0]
    idx = idx[idx > 0]
    with torch.no_grad():
        got = ot(s[:, idx].t().unsqueeze(-1).contiguous(), b[:, idx].t().unsqueeze(-1).contiguous()).cpu().numpy()
    assert got.shape == gold["ot_loss_0"].shape
    assert np.abs(got - gold["ot_loss_0"]).max() <= OT_ABS


@pytest.mark.parametrize("choice,buffer_size", [("ot", 1), ("l2", 3)])
def test_model_meta_loss_over_consecutive_steps_vs_oracle(oracle, choice, buffer_size):
    """3 consecutive train steps of the detector; the (big, small) statistics Dev.forward produced in
    each step are captured and replayed through the oracle's MetaLoss (pinned by the reference-run
    goldens, tests/test_oracle_meta_golden.py) with the model's CURRENT ot_loss weights; the model's
    meta term and history buffer must follow it step by step."""
    from feature_intertwiner_amd.config import make_config
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import set_optimizer, train_step
    torch.manual_seed(21)
    cfg = make_config("resnet50", 256, 2, 64, dev_switch=True, loss_choice=choice, ot_L=5, buffer_size=buffer_size,
                      loss_fac=1.0)
    model = MaskRCNN(cfg).to(DEV)
    opt = set_optimizer(model, cfg.TRAIN)
    batch = synthetic_batch(2, 256, device=DEV)
    model.external_proposals = SyntheticProposals(batch[2], 256)
    model.generator = torch.Generator(device=DEV).manual_seed(3)
    captured = {}
    inner = model.meta_loss

    def spy(feat_input, reduce_fn=None):
        captured["in"] = [t.detach().cpu().numpy() for t in feat_input[:4]]
        if choice == "ot":
            sd = model.ot_loss.state_dict()
            captured["ot"] = dict(g_w=sd["G_net.0.weight"].cpu().numpy(), g_b=sd["G_net.0.bias"].cpu().numpy(),
                                  c_w=sd["critic.0.weight"].cpu().numpy(), c_b=sd["critic.0.bias"].cpu().numpy(),
                                  epsilon=1.0, L=5)
        return inner(feat_input, reduce_fn)
    model.meta_loss = spy
    ml = oracle.MetaLoss(choice, buffer_size, 1024, 81)
    n_selected = 0
    for step in range(3):
        terms = train_step(model, opt, list(batch))
        ml.ot = captured.get("ot")
        exp_v = np.asarray(ml(*captured["in"])).reshape(-1)
        exp = max(float(exp_v.mean()), 0.0)                       # lib/workflow.py:196-200: negative -> 0
        got = float(terms["meta"])
        n_selected += exp_v.size if exp != 0 else 0
        if choice == "ot":
            assert abs(got - exp) <= OT_ABS, (step, got, exp)
        else:
            assert abs(got - exp) <= 1e-4 * abs(exp) + 1e-9, (step, got, exp)
        assert np.array_equal(model.feature_buffer.buffer_cnt.cpu().numpy(), ml.buffer_cnt)
        assert np.allclose(model.feature_buffer.buffer.cpu().numpy(), ml.buffer, rtol=1e-5, atol=1e-7)
    assert n_selected > 0          # the comparison was not vacuous


@pytest.mark.parametrize("choice", ["l2", "kl", "ot"])
@pytest.mark.parametrize("layout", ["stacked", "one_launch_view"])
def test_statistics_kernels_equal_the_tensor_formulation(choice, layout):
    """fi_meta_stats_forward / _backward (merge, history update, selection, transposes: three launches) against the
    tensor formulation of intertwiner.meta_loss on the same inputs over three steps (the second without small-object
    statistics: history untouched, loss 0): loss, history buffer and counts, and the gradient with respect to the small
    class features.  Both memory layouts Dev.forward produces."""
    from feature_intertwiner_amd import intertwiner as IT
    G, S = (2, 3) if layout == "stacked" else (1, 3)
    ot = _ot_module() if choice == "ot" else None
    out = {}
    keep = IT.STATS_KERNEL
    try:
        for on in (True, False):
            IT.STATS_KERNEL = on
            buf = IT.FeatureBuffer(1, F, K, DEV)
            res = []
            for step in (0, 2, 1):
                bf, bc, sf, sc = [torch.from_numpy(a).to(DEV) for a in golden_meta_inputs(step, K, F, G=G, activation=ACT[choice])]
                if layout == "one_launch_view":         # [F, S K] storage viewed as [1, S, F, K]
                    as_view = lambda t: t[0].permute(1, 0, 2).reshape(F, S * K).contiguous().view(F, S, K).permute(1, 0, 2).unsqueeze(0)
                    bf, sf = as_view(bf), as_view(sf)
                    assert not sf.is_contiguous()
                sf = sf.detach().requires_grad_(True)
                loss = IT.meta_loss(_cfg(choice), buf, ot, [bf, bc, sf, sc, None, None])
                loss.backward()
                res.append((float(loss.detach()), sf.grad.clone(), buf.buffer.clone(), buf.buffer_cnt.clone()))
            out[on] = res
    finally:
        IT.STATS_KERNEL = keep
    for (la, ga, ba, ca), (lb, gb, bb, cb) in zip(out[True], out[False]):
        assert torch.equal(ca, cb)
        assert (ba - bb).abs().max().item() <= 1e-6 * bb.abs().max().item()
        if choice == "ot":
            assert abs(la - lb) <= OT_ABS
        else:
            assert abs(la - lb) <= 2e-6 * abs(lb) + 1e-12
            assert (ga - gb).abs().max().item() <= 1e-5 * gb.abs().max().item() + 1e-12
    assert out[True][1][0] == 0.0 and torch.equal(out[True][1][2], out[True][0][2])      # the step without statistics


@pytest.mark.parametrize("choice", ["ot", "l2"])
def test_meta_loss_split_over_two_ranks_equals_the_reference_goldens(golden_dir, choice):
    """The data-parallel form of the statistics side (fi_meta_stats_sums -> ONE all-reduce of the flat sums ->
    fi_meta_stats_from_sums): the golden stacks hold the statistics of the reference's two nn.DataParallel replicas;
    here each replica is a 'rank' with its own history buffer, and the all-reduce is played by adding the other rank's
    sums.  Every rank must end with the reference's loss and history (the goldens of
    test_meta_loss_sequence_vs_reference_goldens), and its gradient into its own statistics must be world-size times
    the single-process gradient's slice (gradients are averaged over the ranks afterwards)."""
    from feature_intertwiner_amd import _lib
    from feature_intertwiner_amd.intertwiner import FeatureBuffer, meta_loss
    gold = np.load(os.path.join(golden_dir, "meta_loss.npz"))
    ot = _ot_module() if choice == "ot" else None
    L = _lib.load()
    bufs = [FeatureBuffer(1, F, K, DEV) for _ in range(2)]
    ref_buf = FeatureBuffer(1, F, K, DEV)
    launches = []
    for step in range(4):
        full = [torch.from_numpy(a).to(DEV) for a in golden_meta_inputs(step, K, F, activation=ACT[choice])]
        S = full[0].size(1)

        def local_sums(r):
            bf, bc, sf, sc = [t[r:r + 1].contiguous() for t in full]
            out = torch.empty(2 * F * K + 2 * K, device=DEV)
            _lib.check(L.fi_meta_stats_sums(_lib.ptr(bf), _lib.ptr(bc.reshape(S, K).contiguous()), bf.stride(2), bf.stride(1),
                                            bf.stride(0), _lib.ptr(sf), _lib.ptr(sc.reshape(S, K).contiguous()), sf.stride(2),
                                            sf.stride(1), sf.stride(0), 1, S, F, K, _lib.ptr(out), _lib.current_stream()), "sums")
            return out
        # single process on the full stack: value, history and the gradient to compare with
        sf_all = full[2].clone().requires_grad_(True)
        ref = meta_loss(_cfg(choice), ref_buf, ot, [full[0], full[1], sf_all, full[3], None, None])
        ref.backward()
        for r in range(2):
            other = local_sums(1 - r)

            def reduce_fn(s, c):                     # the tensor form is not used on this path
                raise AssertionError("the kernel path must take flat_sum")

            def flat_sum(t, other=other):
                launches.append(1)
                t += other
                return 2
            reduce_fn.flat_sum = flat_sum
            sf = full[2][r:r + 1].clone().requires_grad_(True)
            inp = [full[0][r:r + 1], full[1][r:r + 1], sf, full[3][r:r + 1], None, None]
            before = (bufs[r].buffer.clone(), bufs[r].buffer_cnt.clone())
            got = meta_loss(_cfg(choice), bufs[r], ot, inp, reduce_fn=reduce_fn)
            got.backward()
            exp = float(gold["%s_loss_%d" % (choice, step)].mean())
            tol = OT_ABS if choice == "ot" else 2e-5 * abs(exp) + 1e-9
            assert abs(float(got.detach()) - exp) <= tol, (choice, step, r, float(got.detach()), exp)
            assert np.array_equal(bufs[r].buffer_cnt.cpu().numpy(), gold["%s_buffer_cnt_%d" % (choice, step)])
            assert np.allclose(bufs[r].buffer.cpu().numpy(), gold["%s_buffer_%d" % (choice, step)], rtol=2e-6, atol=1e-7)
            # gradient: against the TENSOR formulation of the same data-parallel exchange (same association of the sums:
            # levels first, then ranks), history taken from before this step.  (Against the single-process stack the
            # debiased OT term -- a 1e-3 difference of 0.7-sized terms, SURVEY Q6 -- amplifies the 1e-7 re-association
            # of the merged features beyond any useful bar; l2 is also held to the single-process gradient.)
            class AddOther(torch.autograd.Function):
                @staticmethod
                def forward(ctx, t, o):
                    return t + o

                @staticmethod
                def backward(ctx, g):
                    return g * 2.0, None
            n = F * K
            o_s, o_c = (other[:n], other[2 * n:2 * n + K]), (other[n:2 * n], other[2 * n + K:])
            calls = []

            def tensor_reduce(sm, c):
                big = not calls
                calls.append(1)
                src = o_s if big else o_c
                return AddOther.apply(sm, src[0].view_as(sm)), c + src[1].view_as(c)
            sf_t = full[2][r:r + 1].clone().requires_grad_(True)
            tb = FeatureBuffer(1, F, K, DEV)
            tb.buffer.copy_(before[0]), tb.buffer_cnt.copy_(before[1])
            ten = meta_loss(_cfg(choice), tb, ot, [full[0][r:r + 1], full[1][r:r + 1], sf_t, full[3][r:r + 1], None, None],
                            reduce_fn=tensor_reduce)
            ten.backward()
            assert abs(float(got) - float(ten)) <= (1e-5 if choice == "ot" else 1e-6) * max(abs(float(ten)), 1e-3)
            scale = float(sf_t.grad.abs().max())
            assert float((sf.grad - sf_t.grad).abs().max()) <= (2e-3 if choice == "ot" else 1e-5) * scale + 1e-30, (choice, step, r)
            if choice == "l2":
                g_ref = sf_all.grad[r:r + 1] * 2.0
                assert float((sf.grad - g_ref).abs().max()) <= 2e-5 * float(g_ref.abs().max()) + 1e-30, (choice, step, r)
    assert len(launches) == 8                        # one collective per rank and step


def test_data_parallel_meta_loss_launches_what_the_single_rank_one_does():
    """Round-4 gap: with a reduce_fn the statistics side fell back to ~60 tensor launches.  Device kernels of
    meta_loss forward + backward ('l2': no OT module in the way), counted by torch.profiler: the data-parallel form may
    add the split of the merge (sums | means), the world-size scaling of the gradient and the collective itself."""
    from torch.profiler import ProfilerActivity, profile
    from feature_intertwiner_amd.intertwiner import FeatureBuffer, meta_loss

    def count(reduce_fn):
        buf = FeatureBuffer(1, F, K, DEV)
        full = [torch.from_numpy(a).to(DEV) for a in golden_meta_inputs(0, K, F, activation="sigmoid")]
        n = []
        for rep in range(2):                           # first pass: allocator / constant-tensor warm-up
            sf = full[2].clone().requires_grad_(True)
            torch.cuda.synchronize()
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                meta_loss(_cfg("l2"), buf, None, [full[0], full[1], sf, full[3], None, None], reduce_fn=reduce_fn).backward()
                torch.cuda.synchronize()
            n.append(sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA
                         and "memcpy" not in e.name.lower() and "memset" not in e.name.lower()))
        return n[-1]

    def reduce_fn(s, c):
        raise AssertionError("the kernel path must take flat_sum")
    reduce_fn.flat_sum = lambda t: 2                   # the collective itself is not a launch of ours
    single, dp = count(None), count(reduce_fn)
    assert single <= 30, single                        # (3 statistics launches + the pair loss, the masked mean and their backward: 26)
    assert dp <= single + 2, (single, dp)
