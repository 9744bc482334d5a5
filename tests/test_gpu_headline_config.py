"""GPU: BASELINE configs[2] -- the configuration the headline metric is quoted on (ResNet-101-FPN,
4 x 1024^2, 512 RoIs/image, OT intertwiner on, Sinkhorn L=50, 80 classes) -- for 2 full train steps,
with the inputs and outputs of every RoIAlign, NMS and Sinkhorn launch INSIDE the step captured
(feature_intertwiner_amd._lib.TAP) and re-checked against the CPU oracle: RoIAlign bit-exact, NMS keep
indices exact, OT terms within 1e-4 relative (the north star's bars)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_configs2_full_size_two_steps_operators_vs_oracle(oracle):
    from feature_intertwiner_amd import _lib
    from feature_intertwiner_amd.config import make_config
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import set_optimizer, train_step
    torch.manual_seed(2000)
    cfg = make_config("resnet101", 1024, 4, 512, dev_switch=True, loss_choice="ot", ot_L=50)
    model = MaskRCNN(cfg).to(DEV)
    opt = set_optimizer(model, cfg.TRAIN)
    batch = synthetic_batch(4, 1024, device=DEV, seed=2000)
    model.external_proposals = SyntheticProposals(batch[2], 1024, seed=7)
    model.generator = torch.Generator(device=DEV).manual_seed(11)
    first = float(train_step(model, opt, list(batch))["total"])       # step 1 (fills the history buffer)

    taps = []
    _lib.TAP = lambda name, **kw: taps.append((name, {k: (v.detach().clone() if torch.is_tensor(v) else
                                                         ([m.detach() for m in v] if isinstance(v, list) else v))
                                                      for k, v in kw.items()}))
    from feature_intertwiner_amd import conv as C
    C.FLOP_LOG = {}
    _lib.prof_reset()
    _lib.prof_enable(True)
    try:
        terms = train_step(model, opt, list(batch))                   # step 2, captured
    finally:
        _lib.TAP = None
        _lib.prof_enable(False)
        used, C.FLOP_LOG = dict(C.FLOP_LOG), None
    torch.cuda.synchronize()
    assert all(torch.isfinite(v) for v in terms.values()), terms
    assert float(terms["total"]) < first
    # kernel selection of the headline step: the 3x3 / stride 1 layers run on the patch kernels (2-D tiles on the
    # pyramid maps, flat tiles on the 14 x 14 RoI maps), the wide 1x1 layers on conv1x1_reg_kernel -- and the
    # Python-side bookkeeping (bench.py's per-kernel flops) agrees with what the library launched
    for key in ("conv3x3_patch", "conv3x3_patch_flat", "conv1x1_reg"):
        assert key in used and used[key][0] > 0, (key, sorted(used))
        launches, _ = _lib.prof_get(key)
        assert launches == used[key][0], (key, launches, used[key][0])
    flops = {k: v[1] for k, v in used.items()}
    assert flops["conv3x3_patch"] + flops["conv3x3_patch_flat"] > 0.4 * sum(flops.values())

    names = [n for n, _ in taps]
    assert names.count("nms_sorted") == 1 and names.count("sinkhorn") == 1 and names.count("proposal_candidates") == 1
    # the fused pre-NMS stage inside the step: 4 x 6000 candidates selected from 261 888 anchors + the external
    # proposals, same rows in the same order as the oracle restatement, and they ARE what NMS received
    cand = [t for n, t in taps if n == "proposal_candidates"][0]
    nms_in = [t for n, t in taps if n == "nms_sorted"][0]
    assert torch.equal(cand["dets"], nms_in["boxes"])
    for b in range(4):
        exp, _ = oracle.proposal_candidates(cand["probs"][b].cpu().numpy(), cand["deltas"][b].cpu().numpy(),
                                            cand["anchors"].cpu().numpy(), 6000, cfg.DATA.BBOX_STD_DEV, (1024.0, 1024.0),
                                            cand["extra"][b].cpu().numpy())
        got = cand["dets"][b].cpu().numpy()
        assert np.array_equal(got[:, 4], exp[:, 4])
        assert np.max(np.abs(got[:, :4] - exp[:, :4])) <= 1e-3
    crops = [t for n, t in taps if n == "pyramid_crop"]
    # Dev stage: 7x7 and 14x14 over all 2048 RoIs (channels-last make-up maps) + the 'big' 14x14 crop
    kinds = sorted((c["crop"], c["boxes"].shape[0] == 2048) for c in crops)
    assert kinds == [(7, True), (14, False), (14, True)], kinds

    # ---- RoIAlign: bit-exact, every launch of the step ------------------------------------
    for c in crops:
        maps = [m.cpu().numpy() for m in c["maps"]]                   # logical NCHW whatever the memory format
        boxes, ind, level = c["boxes"].cpu().numpy(), c["box_ind"].cpu().numpy(), c["level"].cpu().numpy()
        got = c["crops"].cpu().numpy()
        # level 0 = the filler rows that round the big branch's row count up to a multiple of 64 (Dev.forward):
        # no pyramid level matches them, their crops must be zero
        # (or -- the stage without its host read, Dev.static_shapes -- the unused part of the big branch's static capacity
        # of 3 * RoIs rows)
        # of 3 * RoIs rows: level -1, rows that are not even written)
        assert (((level >= 2) & (level <= 5)) | (level == 0) | (level == -1)).all()
        assert (level == 0).sum() < 64 and ((level == -1).sum() == 0 or len(level) == 3 * 2048)
        assert not got[level == 0].any()
        for l in range(2, 6):
            sel = np.nonzero(level == l)[0]
            if len(sel) == 0:
                continue
            exp = oracle.crop_and_resize_forward(maps[l - 2], boxes[sel], ind[sel], c["crop"], c["crop"])
            assert np.array_equal(got[sel].view(np.uint32), exp.view(np.uint32)), (c["crop"], l)

    # ---- NMS: keep indices exact (4 images x 6000 candidates, threshold 0.7, first 1000 kept) ----
    t = dict(taps)["nms_sorted"]
    dets = t["boxes"].cpu().numpy()
    assert dets.shape == (4, 6000, 5) and t["thresh"] == pytest.approx(0.7) and not t["strict"]
    keep, num = t["keep"].cpu().numpy(), t["num_out"].cpu().numpy()
    for b in range(4):
        exp = oracle.pth_nms(dets[b], 0.7)[:t["max_keep"]]
        assert int(num[b]) == len(exp) and np.array_equal(keep[b, :len(exp)], exp), b
    assert num.min() > 100

    # ---- Sinkhorn: 3 x 80 problems of 256 x 256, L = 50, each term within 1e-4 relative ----------
    t = dict(taps)["sinkhorn"]
    x, y, got = t["x"].cpu().numpy(), t["y"].cpu().numpy(), t["loss"].cpu().numpy()
    assert x.shape == (240, 256, 1) and t["L"] == 50 and t["C_form"] == "cosine"
    exp = np.array([oracle.sinkhorn(x[p], y[p], t["eps_inv"], 50) for p in range(240)])
    assert np.all(np.abs(got - exp) <= 1e-4 * np.abs(exp) + 1e-7), np.abs(got - exp).max()
    del model, opt
    torch.cuda.empty_cache()


def test_configs2_backward_forms_agree_at_full_size():
    """The headline rests on the backward pass that skips structural zeros: at BASELINE configs[2]
    itself -- R101, 23 C4 blocks, 4 x 1024^2, 512 RoIs per image (170 positive slots), L = 50, 261 888 anchors with the
    border / level-start anchors at real scale -- one backward pass in the default form and one in the dense form
    (conv.GATES = conv._UNSCALED_BACKWARD = False) from identical weights, inputs and random draws: losses equal to
    1e-6, every parameter's gradient within 2e-5 of its largest element, the same parameters without a gradient.
    lib/layers.py:808-934 (which rows the losses read), lib/model.py:442 (the mask head on all RoIs)."""
    from feature_intertwiner_amd.config import make_config
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import check_backward_forms, compare_backward_forms, set_optimizer, train_step
    torch.manual_seed(2000)
    cfg = make_config("resnet101", 1024, 4, 512, dev_switch=True, loss_choice="ot", ot_L=50)
    model = MaskRCNN(cfg).to(DEV)
    opt = set_optimizer(model, cfg.TRAIN)
    batch = synthetic_batch(4, 1024, device=DEV, seed=2000)
    model.external_proposals = SyntheticProposals(batch[2], 1024, seed=7)
    model.generator = torch.Generator(device=DEV).manual_seed(11)
    for _ in range(2):                       # two real steps first: history buffer filled, weights off their initial values
        train_step(model, opt, list(batch))
    # (check_: a pre-activation of the RPN's shared convolution within rounding of zero may fall on the other side of its
    # ReLU in the default form, which evaluates it at the sampled anchors as a matrix product; one mask bit then moves
    # 1e-3..1e-2 of a channel's gradient.  Such a pass is set aside only with evidence -- sign disagreement at a sampled row
    # whose float64 pre-activation is within rounding of zero -- and replayed with the dense kernel's mask bits)
    r = check_backward_forms(model, batch, bar=2e-5, detail=6)
    assert not r.get("boundary_unverified"), r
    assert r["none_sets_equal"], r
    assert r["params"] > 400, r
    assert r["loss_rel"] <= 1e-6, r
    assert r["max_rel_dev"] <= 2e-5, r
    # ... and the default form against ITSELF after further real steps: the step runs on three streams, and a tensor that
    # crosses between them without being marked for its reader (Tensor.record_stream) shows up here as a gradient that
    # changes from run to run (round 4: the class counts the meta loss keeps for its backward pass were such a tensor --
    # one pass in four lost the feature extractor's gradient; scripts/bfp_sweep.sh)
    for _ in range(6):
        train_step(model, opt, list(batch))
        r = compare_backward_forms(model, batch, forms=("default", "default"), detail=4)
        assert r["loss_rel"] == 0.0, r
        assert r["max_rel_dev"] <= 2e-5, r
    del model, opt
    torch.cuda.empty_cache()


def test_configs4_slice_backward_forms_agree_bf16():
    """The same differential check on the configs[4] slice (R101, 2 x 1344^2, 1000 RoIs, bf16 MFMA convolutions) at the
    16-bit bar of tests/test_gpu_detector.py::test_detector_gradients_agree_between_backward_forms: the two forms round
    the BatchNorm scale into different operands, and the 1-D cosine OT gradient is numerically degenerate,
    so parameters reached only through it are compared against the largest detector gradient."""
    from feature_intertwiner_amd.config import make_config
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import check_backward_forms, compare_backward_forms, set_optimizer, train_step
    torch.manual_seed(2000)
    cfg = make_config("resnet101", 1344, 2, 1000, dev_switch=True, loss_choice="ot", ot_L=50, conv_precision="bf16")
    model = MaskRCNN(cfg).to(DEV)
    opt = set_optimizer(model, cfg.TRAIN)
    batch = synthetic_batch(2, 1344, device=DEV, seed=2000)
    model.external_proposals = SyntheticProposals(batch[2], 1344, seed=7)
    model.generator = torch.Generator(device=DEV).manual_seed(11)
    train_step(model, opt, list(batch))
    r = check_backward_forms(model, batch, bar=6e-2, skip=lambda n: n.startswith("ot_loss") or n.startswith("dev_roi.feat_extract"))
    assert r["none_sets_equal"], r
    assert r["loss_rel"] <= 1e-5, r
    assert r["max_rel_dev"] <= 6e-2, r
    del model, opt
    torch.cuda.empty_cache()


def test_configs4_slice_full_size_bf16(oracle):
    """The single-GPU slice of BASELINE configs[4]: ResNet-101-FPN, 1333x800 padded to 1344^2,
    2 images per GPU, 1000 RoIs per image with the mask head, bf16-input MFMA convolutions.  Two full train
    steps; the RoIAlign and NMS launches inside the step stay bit-exact / index-exact against the oracle
    (they do not depend on the conv arithmetic), the losses are finite and go down."""
    from feature_intertwiner_amd import _lib, conv as C
    from feature_intertwiner_amd.config import make_config
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import set_optimizer, train_step
    torch.manual_seed(2000)
    cfg = make_config("resnet101", 1344, 2, 1000, dev_switch=True, loss_choice="ot", ot_L=50, conv_precision="bf16")
    model = MaskRCNN(cfg).to(DEV)
    opt = set_optimizer(model, cfg.TRAIN)
    batch = synthetic_batch(2, 1344, device=DEV, seed=2000)
    model.external_proposals = SyntheticProposals(batch[2], 1344, seed=7)
    model.generator = torch.Generator(device=DEV).manual_seed(11)
    first = float(train_step(model, opt, list(batch))["total"])
    taps = []
    C.FLOP_LOG = {}
    _lib.TAP = lambda name, **kw: taps.append((name, {k: (v.detach().clone() if torch.is_tensor(v) else
                                                         ([m.detach() for m in v] if isinstance(v, list) else v))
                                                      for k, v in kw.items()}))
    try:
        terms = train_step(model, opt, list(batch))
    finally:
        _lib.TAP = None
        used, C.FLOP_LOG = dict(C.FLOP_LOG), None
    torch.cuda.synchronize()
    assert all(torch.isfinite(v) for v in terms.values()), terms
    assert float(terms["total"]) < first
    # the bf16 kernels carried the conv stack (the 3-channel stem, a few narrow layers and the heads' fully
    # connected layers -- 3.6 % of the flops, conv.linear -- stay on the fp32 kernels)
    bf = sum(f for k, (n, f) in used.items() if "bf16" in k)
    assert bf > 0.95 * sum(f for n, f in used.values()), {k: v[1] / 1e9 for k, v in used.items()}
    assert C.conv_precision() == "fp32"
    crops = [t for n, t in taps if n == "pyramid_crop"]
    assert sorted((c["crop"], c["boxes"].shape[0] == 2000) for c in crops) == [(7, True), (14, False), (14, True)]
    c7 = [c for c in crops if c["crop"] == 7][0]
    assert [tuple(m.shape[2:]) for m in c7["maps"]] == [(336, 336), (168, 168), (84, 84), (42, 42)]
    for c in crops:
        maps = [m.cpu().numpy() for m in c["maps"]]
        boxes, ind, level = c["boxes"].cpu().numpy(), c["box_ind"].cpu().numpy(), c["level"].cpu().numpy()
        got = c["crops"].cpu().numpy()
        for l in range(2, 6):
            sel = np.nonzero(level == l)[0]
            if len(sel):
                exp = oracle.crop_and_resize_forward(maps[l - 2], boxes[sel], ind[sel], c["crop"], c["crop"])
                assert np.array_equal(got[sel].view(np.uint32), exp.view(np.uint32)), (c["crop"], l)
    t = dict(taps)["nms_sorted"]
    dets, keep, num = t["boxes"].cpu().numpy(), t["keep"].cpu().numpy(), t["num_out"].cpu().numpy()
    assert dets.shape == (2, 6000, 5)
    for b in range(2):
        exp = oracle.pth_nms(dets[b], 0.7)[:t["max_keep"]]
        assert int(num[b]) == len(exp) and np.array_equal(keep[b, :len(exp)], exp)
    del model, opt
    torch.cuda.empty_cache()


def test_train_step_has_no_host_synchronisation():
    """Round 4: the step reads nothing back -- the RoI counts per level that sized the Dev stage's batches stay on the
    device (Dev.static_shapes).  torch's sync debug mode turns every synchronising call of the framework (.item(),
    .tolist(), .cpu(), pageable host-to-device copies, nonzero, ...) into an error; three steps of the headline model
    class (R50 here: the code path does not depend on the depth) run under it after two warm-up steps."""
    from feature_intertwiner_amd.config import make_config
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import set_optimizer, train_step
    torch.manual_seed(2000)
    cfg = make_config("resnet50", 512, 2, 128, dev_switch=True, loss_choice="ot", ot_L=10)
    model = MaskRCNN(cfg).to(DEV)
    opt = set_optimizer(model, cfg.TRAIN)
    batch = synthetic_batch(2, 512, device=DEV, seed=2000)
    model.external_proposals = SyntheticProposals(batch[2], 512, seed=7)
    model.generator = torch.Generator(device=DEV).manual_seed(11)
    assert model.dev_roi.static_shapes(torch.zeros(2, 128, 4, device=DEV))
    for _ in range(2):
        train_step(model, opt, list(batch))
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        for _ in range(3):
            terms = train_step(model, opt, list(batch))
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert all(torch.isfinite(v) for v in terms.values()), terms


def test_whole_train_step_replays_as_one_hip_graph():
    """With no host synchronisation left, the WHOLE train step -- forward on three streams, target kernels, losses,
    backward, clip + SGD -- captures into one hipGraph (torch.cuda.CUDAGraph) and replays: the replayed steps train (the
    loss falls, the parameters move) and track the same number of eager steps from the same start to the tolerance of
    fp32 atomics.  (On this machine the replay is not faster than eager launching -- 112.4 vs 110.2 ms/step at the
    headline size, scripts/graph_probe.py -- so bench.py launches eagerly; the test pins the property.)"""
    from feature_intertwiner_amd.config import make_config
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import set_optimizer, train_step

    def build():
        torch.manual_seed(2000)
        cfg = make_config("resnet50", 512, 2, 128, dev_switch=True, loss_choice="ot", ot_L=10)
        model = MaskRCNN(cfg).to(DEV)
        opt = set_optimizer(model, cfg.TRAIN)
        batch = synthetic_batch(2, 512, device=DEV, seed=2000)
        model.external_proposals = SyntheticProposals(batch[2], 512, seed=7)
        model.generator = torch.Generator(device=DEV).manual_seed(11)
        return model, opt, batch

    model, opt, batch = build()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            first = train_step(model, opt, list(batch))
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    graph.register_generator_state(model.generator)
    graph.register_generator_state(model.external_proposals.gen)
    with torch.cuda.graph(graph, stream=side):
        terms = train_step(model, opt, list(batch))
    before = [p.detach().clone() for p in model.parameters()]
    for _ in range(8):
        graph.replay()
    torch.cuda.synchronize()
    assert all(torch.isfinite(v) for v in terms.values()), terms
    assert float(terms["total"]) < float(first["total"])
    moved = sum(float((p.detach() - b).abs().max()) > 0 for p, b in zip(model.parameters(), before))
    assert moved > 100
