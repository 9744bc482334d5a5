"""The Dev stage without its host read (Dev.static_shapes / Dev._forward_static: all RoIs through the small branch, a big
branch of static capacity whose live count stays on the device) against the stage WITH the read (FI_STATIC_DEV=0: batches
sized by the RoI counts per level, as lib/sub_module.py:437-540 of the reference sizes them with nonzero / .any()).  Same
weights, inputs and random draws: the same RoIs reach the same layers, so losses and gradients agree up to the summation
order of the fully connected stages (their K split depends on the row count)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("rois,choice,precision", [(64, "l2", "fp32"), (100, "l2", "fp32"), (100, "kl", "fp32"),
                                                   (100, "ot", "fp32"), (100, "l2", "bf16")])
def test_static_dev_stage_equals_the_stage_with_the_host_read(rois, choice, precision):
    from feature_intertwiner_amd import conv as C
    from feature_intertwiner_amd import sub_module as SM
    from feature_intertwiner_amd.config import make_config
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import compute_loss
    cfg = make_config(backbone="resnet50", image_size=256, batch_size=2, train_rois_per_image=rois, loss_choice=choice,
                      ot_L=5, conv_precision=precision)
    keep = SM._STATIC_DEV
    out = {}
    try:
        for static in (True, False):
            SM._STATIC_DEV = static
            torch.manual_seed(2000)
            model = MaskRCNN(cfg).to(DEV)
            batch = synthetic_batch(2, 256, device=DEV)
            model.external_proposals = SyntheticProposals(batch[2], 256)
            assert model.dev_roi.static_shapes(torch.zeros(2, rois, 4, device=DEV)) == static
            for step in range(2):                   # the second pass runs on a filled history buffer
                model.generator = torch.Generator(device=DEV).manual_seed(3)
                for p in model.parameters():
                    p.grad = None
                loss, terms = compute_loss(model, list(batch), True, 1, None)
                loss.backward()
                join, model._side_join = getattr(model, "_side_join", None), None
                if join is not None:
                    join()
            torch.cuda.synchronize()
            out[static] = ({k: float(v) for k, v in terms.items()},
                           {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None},
                           model.feature_buffer.buffer.clone(), model.feature_buffer.buffer_cnt.clone())
            del model
    finally:
        SM._STATIC_DEV = keep
        C.set_conv_precision("fp32")
        C.invalidate_step_state()
    (ta, ga, ba, ca), (tb, gb, bb, cb) = out[True], out[False]
    lowp = precision != "fp32"
    assert torch.equal(ca, cb)                                     # the class counts of the history buffer: exact
    tol = 2e-2 if lowp else 1e-5
    assert (ba - bb).abs().max().item() <= tol * bb.abs().max().item()
    for k in tb:
        if k in ("meta", "total") and choice == "ot":
            continue        # the 1-D cosine OT value is rounding noise around its operands' last bits
        assert abs(ta[k] - tb[k]) <= (2e-3 if lowp else 1e-5) * max(abs(tb[k]), 1e-3), (k, ta[k], tb[k])
    assert ga.keys() == gb.keys()
    gmax = max(float(g.abs().max()) for g in gb.values())
    for n, ref in gb.items():
        if choice == "ot" and (n.startswith("ot_loss") or n.startswith("dev_roi.feat_extract")):
            continue
        diff = float((ga[n] - ref).abs().max())
        bar = (6e-2 if lowp else 1e-4) * float(ref.abs().max()) + 1e-7 * gmax
        assert diff <= bar, (n, diff, float(ref.abs().max()))


@pytest.mark.parametrize("N,probs", [(8, (1, 1, 1, 1)), (2048, (4, 3, 2, 1)), (2000, (1, 0, 2, 5)), (3000, (0, 0, 0, 1)),
                                     (1024, (1, 0, 0, 0)), (5000, (3, 1, 0, 2))])
def test_index_kernel_equals_its_tensor_formulation(N, probs):
    """fi_dev_stage_index against Dev._static_index_tensors (torch.sort / nonzero_static / where): every output, bit for
    bit -- levels that are absent, all rows on one level, row counts that are not multiples of the workgroup size."""
    from feature_intertwiner_amd import sub_module as SM
    g = torch.Generator().manual_seed(N)
    level = (2 + torch.multinomial(torch.tensor(probs, dtype=torch.float), N, replacement=True, generator=g)).to(torch.int32).to(DEV)
    gt = torch.randint(0, 81, (N,), generator=g).to(torch.int32).to(DEV)
    gt[torch.rand(N, generator=g).to(DEV) < 0.3] = 0
    cap = (3 * N + 63) // 64 * 64
    dev_stage = SM.Dev.__new__(SM.Dev)          # the two methods under test use no module state
    keep = SM._INDEX_KERNEL
    try:
        SM._INDEX_KERNEL = True
        got = SM.Dev._static_index(dev_stage, level, gt, 81, cap)
        got_inf = SM.Dev._static_index(dev_stage, level, None, 81, cap)
    finally:
        SM._INDEX_KERNEL = keep
    ref = SM.Dev._static_index_tensors(level, gt, 81, cap)
    ref_inf = SM.Dev._static_index_tensors(level, None, 81, cap)
    torch.cuda.synchronize()
    names = ["order", "small_cls", "small_gt", "small_on", "big_idx", "big_level", "big_cls", "live"]
    for name, a, b in zip(names, got, ref):
        if name == "big_idx":           # behind the live count the row index is a don't-care (level -1)
            n_live = int(ref[7])
            a, b = a[:n_live], b[:n_live]
        assert a.dtype == b.dtype and torch.equal(a, b), name
    for name, a, b in zip(names, got_inf, ref_inf):
        if name in ("order", "small_on", "big_level", "live"):
            assert torch.equal(a, b), name


def test_picked_streams_run_next_to_each_other():
    """_lib.pick_stream: HIP binds a process's streams to at most GPU_MAX_HW_QUEUES hardware queues,
    and two streams on one queue run their kernels one after the other.  The streams the step uses are picked by
    measurement; here the measurement is repeated on the picked ones: a kernel on each overtakes a spin kernel on the
    current stream and on the other picked streams."""
    from feature_intertwiner_amd import _lib
    torch.cuda.synchronize()
    a, b = _lib.side_stream(0), _lib.side_stream3(0)
    main = torch.cuda.current_stream()
    assert a is not b and a != main and b != main
    probe = torch.zeros(64, device=DEV)
    for x, y in ((main, a), (main, b), (a, b), (b, a)):
        assert _lib._overtakes(x, y, probe), "two of the step's streams share a hardware queue"


@pytest.mark.parametrize("N,probs", [(2048, (4, 3, 2, 1)), (2000, (1, 0, 2, 5)), (777, (0, 1, 1, 1)), (64, (1, 1, 1, 0))])
def test_index_kernel_against_the_oracles_level_loop(oracle, N, probs):
    """fi_dev_stage_index against oracle.dev_stage_groups (the reference's loop over the levels, lib/sub_module.py:437-598):
    the level-major order, which rows are small rows of which level and class, and every level's big rows in RoI order
    with their classes -- dropped (class index 0) when the level has no small box."""
    from feature_intertwiner_amd import sub_module as SM
    K = 81
    g = torch.Generator().manual_seed(100 + N)
    level = (2 + torch.multinomial(torch.tensor(probs, dtype=torch.float), N, replacement=True, generator=g)).to(torch.int32)
    gt = torch.randint(0, K, (N,), generator=g).to(torch.int32)
    ref = oracle.dev_stage_groups(level.numpy(), gt.numpy())
    cap = (3 * N + 63) // 64 * 64
    keep = SM._INDEX_KERNEL
    try:
        SM._INDEX_KERNEL = True
        order, small_cls, small_gt, small_on, big_idx, big_level, big_cls, live = [
            t.cpu().numpy() for t in SM.Dev._static_index(SM.Dev.__new__(SM.Dev), level.to(DEV), gt.to(DEV), K, cap)]
    finally:
        SM._INDEX_KERNEL = keep
    assert (order == ref["order"]).all()
    assert (small_gt == ref["small_gt_all"]).all()
    n_small = sum(len(ref["small"][l]) for l in (2, 3, 4))
    assert small_on[:n_small].all() and not small_on[n_small:].any()
    pos, bpos = 0, 0
    for l in (2, 3, 4):
        idx = ref["small"][l]
        want = np.where(gt.numpy()[idx] > 0, (l - 2) * K + gt.numpy()[idx], 0)
        assert (small_cls[pos:pos + len(idx)] == want).all()
        pos += len(idx)
        above = np.nonzero(level.numpy() > l)[0]                 # the stage crops them whether or not they count
        assert (big_idx[bpos:bpos + len(above)] == above).all() and (big_level[bpos:bpos + len(above)] == l).all()
        counted = len(ref["big"][l]) > 0 or len(above) == 0
        want = np.where(gt.numpy()[above] > 0, (l - 2) * K + gt.numpy()[above], 0) if counted else np.zeros(len(above), np.int64)
        assert (big_cls[bpos:bpos + len(above)] == want).all()
        bpos += len(above)
    assert int(live[0]) == bpos and (big_level[bpos:] == -1).all() and (small_cls[pos:] == 0).all()
