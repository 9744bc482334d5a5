"""Fused clip_grad_norm_ + SGD step (csrc/sgd.hip) against torch.nn.utils.clip_grad_norm_ + torch.optim.SGD
(the reference's lib/workflow.py:226-230 with tools/utils.py:474-501's two parameter groups)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _params(seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    shapes = [(64, 32, 3, 3), (8200,), (16, 16, 1, 1), (3, 5), (1,), (33000,), (128, 64, 3, 3), (7,)]
    ps = []
    for i, s in enumerate(shapes):
        t = torch.randn(s, generator=g).to(DEV)
        if len(s) == 4 and i != 2:
            t = t.contiguous(memory_format=torch.channels_last)      # parameters stored channels-last (conv.prepare_step)
        ps.append(torch.nn.Parameter(t))
    return ps


def _make(ps, lr=0.02, momentum=0.9, wd=1e-4):
    return torch.optim.SGD([{"params": ps[:5], "weight_decay": wd}, {"params": ps[5:]}], lr=lr, momentum=momentum)


def _set_grads(ps, seed, scale, skip=()):
    g = torch.Generator(device="cpu").manual_seed(seed)
    for i, p in enumerate(ps):
        if i in skip:
            p.grad = None
            continue
        t = (torch.randn(p.shape, generator=g) * scale).to(DEV)
        p.grad = t.contiguous(memory_format=torch.channels_last) if (p.dim() == 4 and not p.is_contiguous()) else t


@pytest.mark.parametrize("momentum", [0.9, 0.0])
def test_matches_torch_clip_and_sgd(momentum):
    from feature_intertwiner_amd import optim
    a, b = _params(1), _params(1)
    oa, ob = _make(a, momentum=momentum), _make(b, momentum=momentum)
    assert optim.supported(oa)
    # step 0: small gradients (no clipping), 1: large (clipped), 2: one parameter without a gradient, 3: lr change
    for step, (scale, skip) in enumerate([(1e-4, ()), (3.0, ()), (0.5, (3, 6)), (2.0, ())]):
        _set_grads(a, 100 + step, scale, skip)
        _set_grads(b, 100 + step, scale, skip)
        if step == 3:
            for o in (oa, ob):
                for grp in o.param_groups:
                    grp["lr"] = 0.005
        before = [p._version for p in a]
        snap = [p.detach().clone() for p in a]
        norm = optim.clip_and_step(oa, 5.0)
        ref_norm = torch.nn.utils.clip_grad_norm_([p for p in b if p.grad is not None], 5.0)
        ob.step()
        torch.cuda.synchronize()
        assert abs(float(norm) - float(ref_norm)) <= 2e-6 * float(ref_norm)
        for i, (p, q) in enumerate(zip(a, b)):
            if i in skip:
                assert p.grad is None and torch.equal(p.detach(), snap[i])      # untouched, like torch's step
                continue
            assert p._version > before[i]                          # caches keyed on versions must notice
            assert p.stride() == q.stride()
            tol = 2e-6 * float(q.detach().abs().max()) + 1e-9
            assert float((p - q).detach().abs().max()) <= tol, (step, i)
            assert float((p.grad - q.grad).abs().max()) <= 2e-6 * float(q.grad.abs().max()) + 1e-12
            if momentum:
                ba, bb = oa.state[p]["momentum_buffer"], ob.state[q]["momentum_buffer"]
                # b = m*b + g cancels: the bar is relative to the terms, not to the (possibly tiny) result
                assert float((ba - bb).abs().max()) <= 4e-6 * float(bb.abs().max() + q.grad.abs().max()) + 1e-12


def test_state_dict_continues_in_torch():
    """The optimizer object still owns the state: its state dict loads into a stock SGD that continues alike."""
    from feature_intertwiner_amd import optim
    a = _params(2)
    oa = _make(a)
    for step in range(2):
        _set_grads(a, 7 + step, 1.0)
        optim.clip_and_step(oa, 5.0)
    b = [torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in a]
    ob = _make(b)
    ob.load_state_dict(copy.deepcopy(oa.state_dict()))
    _set_grads(a, 50, 1.0)
    _set_grads(b, 50, 1.0)
    optim.clip_and_step(oa, 5.0)
    torch.nn.utils.clip_grad_norm_(b, 5.0)
    ob.step()
    for p, q in zip(a, b):
        assert float((p - q).detach().abs().max()) <= 2e-6 * float(q.detach().abs().max()) + 1e-9


def test_deterministic_norm_and_no_clip():
    from feature_intertwiner_amd import optim
    norms = []
    for _ in range(2):
        a = _params(3)
        oa = _make(a)
        _set_grads(a, 9, 1.0)
        norms.append(float(optim.clip_and_step(oa, None)))          # no clipping: gradients untouched
        g = torch.Generator(device="cpu").manual_seed(9)
        first = (torch.randn(a[0].shape, generator=g)).to(DEV)
        assert torch.equal(a[0].grad.contiguous(), first)
    assert norms[0] == norms[1]                                     # fixed reduction order


def test_train_step_uses_the_fused_step():
    from feature_intertwiner_amd import optim
    from feature_intertwiner_amd.config import make_config
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import set_optimizer, train_step
    torch.manual_seed(0)
    cfg = make_config("resnet50", 256, 2, 64, dev_switch=True, loss_choice="ot", ot_L=5)
    models, outs = [], []
    for fused in (True, False):
        torch.manual_seed(0)
        model = MaskRCNN(cfg).to(DEV)
        opt = set_optimizer(model, cfg.TRAIN)
        batch = synthetic_batch(2, 256, device=DEV, seed=5)
        model.external_proposals = SyntheticProposals(batch[2], 256, seed=7)
        model.generator = torch.Generator(device=DEV).manual_seed(11)
        old = optim.supported
        if not fused:
            optim.supported = lambda o: False
        try:
            terms = train_step(model, opt, list(batch))
        finally:
            optim.supported = old
        models.append(model)
        outs.append({k: float(v) for k, v in terms.items()})
    assert outs[0] == outs[1]
    worst = 0.0
    for (n, p), (_, q) in zip(models[0].named_parameters(), models[1].named_parameters()):
        worst = max(worst, float((p - q).detach().abs().max()) / (float(q.detach().abs().max()) + 1e-12))
    assert worst <= 1e-5, worst


def test_static_table_path_with_persistent_gradients():
    """Gradients that stay where they were (the gradient arena): the descriptor table is reused, and every change the
    fast path has to notice -- a learning-rate change, a replaced momentum buffer, a gradient dropped, a gradient
    re-allocated -- still produces torch's result."""
    from feature_intertwiner_amd import optim
    a, b = _params(4), _params(4)
    oa, ob = _make(a), _make(b)
    _set_grads(a, 1, 1.0)
    _set_grads(b, 1, 1.0)
    slots = [p.grad for p in a]
    tables = []
    for step in range(7):
        g = torch.Generator(device="cpu").manual_seed(200 + step)
        for i, (p, q) in enumerate(zip(a, b)):
            t = torch.randn(p.shape, generator=g).to(DEV) * (3.0 if step % 2 else 0.01)
            q.grad = t.contiguous(memory_format=torch.channels_last) if (p.dim() == 4 and not p.is_contiguous()) else t
            if step == 4 and i == 2:
                p.grad = None                                   # dropped this step
                q.grad = None
            elif step == 6 and i == 1:
                p.grad = q.grad.clone()                         # re-allocated elsewhere
            else:
                slots[i].copy_(q.grad)
                p.grad = slots[i].view_as(slots[i])             # a NEW tensor object over the same memory, like an arena view
        if step == 2:
            for o in (oa, ob):
                o.param_groups[1]["lr"] = 0.004
        if step == 3:
            oa.state[a[5]]["momentum_buffer"] = oa.state[a[5]]["momentum_buffer"].clone()
        versions = [p._version for p in a]
        optim.clip_and_step(oa, 5.0)
        tables.append(optim._CACHE[oa]["table"].data_ptr())
        torch.nn.utils.clip_grad_norm_([q for q in b if q.grad is not None], 5.0)
        ob.step()
        for i, (p, q) in enumerate(zip(a, b)):
            if q.grad is None:
                continue
            assert p._version > versions[i]
            assert float((p - q).detach().abs().max()) <= 2e-6 * float(q.detach().abs().max()) + 1e-9, (step, i)
            ba, bb = oa.state[p]["momentum_buffer"], ob.state[q]["momentum_buffer"]
            assert float((ba - bb).abs().max()) <= 4e-6 * float(bb.abs().max() + q.grad.abs().max()) + 1e-12, (step, i)
    # same table for steps 0-1; rebuilt at 2 (lr), 3 (buffer), 4 (gradient dropped), 5 (it is back); patched in place at 6 (moved)
    assert tables[0] == tables[1] and tables[5] == tables[6]
    torch.cuda.synchronize()


def test_a_step_with_a_non_finite_gradient_norm_is_skipped_under_the_guard():
    """16-bit paths (static loss scale): one overflowed data gradient gives an inf norm; the guarded form leaves weights
    and momentum buffers untouched and counts the event -- decided on the device."""
    from feature_intertwiner_amd import optim
    a, b = _params(4), _params(4)
    oa, ob = _make(a), _make(b)
    for step, bad in enumerate([None, float("inf"), float("nan"), None]):
        _set_grads(a, 300 + step, 0.5)
        _set_grads(b, 300 + step, 0.5)
        if bad is not None:
            a[5].grad.view(-1)[17] = bad
        snap = [p.detach().clone() for p in a]
        bufs = [oa.state[p]["momentum_buffer"].clone() for p in a] if step else None
        optim.clip_and_step(oa, 5.0, skip_nonfinite=True)
        torch.cuda.synchronize()
        if bad is not None:
            for p, s in zip(a, snap):
                assert torch.equal(p.detach(), s)
            for p, s in zip(a, bufs):
                assert torch.equal(oa.state[p]["momentum_buffer"], s)
        else:
            torch.nn.utils.clip_grad_norm_(b, 5.0)
            ob.step()
            for p, q in zip(a, b):
                assert torch.allclose(p.detach(), q.detach(), rtol=2e-6, atol=1e-7)
    assert optim.skipped_steps(oa) == 2
    # the plain form on a NaN gradient poisons every weight (what the guard is for; an inf norm gives a clip factor of 0
    # and a NaN only where inf * 0 is formed)
    _set_grads(a, 400, 0.5)
    a[5].grad.view(-1)[3] = float("nan")
    optim.clip_and_step(oa, 5.0)
    torch.cuda.synchronize()
    assert not torch.isfinite(a[0].detach()).all()
