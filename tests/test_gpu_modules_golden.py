"""GPU: the conv stacks around the RoI operators (ResNet-50-FPN, RPN, Classifier, Mask, Dev
make-up layer + feat_extract) vs outputs of the REFERENCE's modules run on CPU with the same
deterministic weights (tests/golden/modules_r50_128.npz, oracle/gen_golden_layers.py).
Tolerance: fp32 convs in a different summation order through ~50 layers: 2e-4 of the tensor's
max magnitude (SURVEY 8c pins convs at 1e-5 rel per layer)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN_MODULE_SEEDS, filled_state, golden_module_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(got, exp, tol=2e-4):
    got = got.detach().cpu().numpy()
    assert got.shape == exp.shape
    err = np.abs(got - exp).max()
    assert err <= tol * max(np.abs(exp).max(), 1e-3), (err, np.abs(exp).max())


def test_module_stacks_match_reference_outputs():
    from feature_intertwiner_amd import sub_module as M
    from feature_intertwiner_amd.config import make_config
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "modules_r50_128.npz"))
    cfg = make_config(backbone="resnet50", image_size=128, batch_size=1, train_rois_per_image=16, dev_switch=True)
    r = M.ResNet("resnet50", stage5=True)
    mods = dict(fpn=M.FPN(cfg, *r.stages(), out_channels=256), rpn=M.RPN(3, 1, 256),
                classifier=M.Classifier(256, 81, 7, cfg), mask=M.Mask(256, 81), dev=M.Dev(cfg, 256))
    for k, m in mods.items():
        st = filled_state([(n, tuple(v.shape)) for n, v in m.state_dict().items()], GOLDEN_MODULE_SEEDS[k])
        m.load_state_dict({n: torch.from_numpy(v) for n, v in st.items()})
        m.to(DEV).eval()
    image, pooled, pooled14 = (torch.from_numpy(a).to(DEV) for a in golden_module_inputs())
    with torch.no_grad():
        p2, p3, p4, p5, p6, _ = mods["fpn"](image, "inference")
        _close(p2[:, ::16], gold["p2_sub"]); _close(p3[:, ::8], gold["p3_sub"])
        _close(p4, gold["p4"]); _close(p5, gold["p5"]); _close(p6, gold["p6"])
        p4_ref = torch.from_numpy(gold["p4"]).to(DEV)
        logits, probs, bbox = mods["rpn"](p4_ref)
        _close(logits, gold["rpn_logits"]); _close(probs, gold["rpn_probs"]); _close(bbox, gold["rpn_bbox"])
        cl, cp, cb = mods["classifier"](pooled, None, None)
        _close(cl, gold["cls_logits"]); _close(cp, gold["cls_probs"]); _close(cb, gold["cls_bbox"])
        _close(mods["mask"](pooled14)[:, ::8], gold["mask_sub"])
        from feature_intertwiner_amd.conv import conv_bn_act
        up = mods["dev"].upsample[0]
        _close(conv_bn_act(p4_ref, up[0], up[1], relu=True), gold["dev_upsample_p4"])
        _close(mods["dev"]._feat_extract(pooled14), gold["dev_feat_extract"])
