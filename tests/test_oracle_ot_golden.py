"""CPU: the oracle's OT restatement vs golden vectors produced by RUNNING THE REFERENCE
(oracle/gen_golden_ot.py imports lib/OT_module.py).  This is what pins the oracle."""
import os

import numpy as np

REL = 1e-5


def test_sinkhorn_terms(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "ot_sinkhorn.npz"))
    n = 0
    for k in g.files:
        if "_L" not in k:
            continue
        name, rest = k.split("_L")
        L, eps = rest.split("_eps")
        form = "l2" if name.startswith("l2") else "cosine"
        v = oracle.sinkhorn(g[name + "_x"], g[name + "_y"], 1.0 / float(eps), int(L), form)
        assert abs(v - g[k]) <= REL * abs(g[k]), (k, v, g[k])
        n += 1
    assert n == 24


def test_opttrans_1d_small(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "ot_1d_small.npz"))
    for L in (5, 50):
        pre = "L%d_" % L
        sd = {k[len(pre) + 3:]: g[k] for k in g.files if k.startswith(pre + "sd_")}
        loss, (t_xy, t_xx, t_yy) = oracle.opttrans_1d_forward(
            g[pre + "x"], g[pre + "y"], sd["G_net.0.weight"], sd["G_net.0.bias"],
            sd["critic.0.weight"], sd["critic.0.bias"], 1.0, L, return_terms=True)
        for got, key in ((t_xy, "t_xy"), (t_xx, "t_xx"), (t_yy, "t_yy")):
            assert np.all(np.abs(got - g[pre + key]) <= REL * np.abs(g[pre + key])), key
        # combination: absolute (cancellation)
        assert np.all(np.abs(loss - g[pre + "loss"]) <= REL * np.abs(g[pre + "t_xy"]).max())
        lb = oracle.opttrans_1d_forward(g[pre + "x"], g[pre + "y"], sd["G_net.0.weight"], sd["G_net.0.bias"],
                                        sd["critic.0.weight"], sd["critic.0.bias"], 1.0, L, remove_bias=True)
        assert np.all(np.abs(lb - g[pre + "loss_remove_bias"]) <= REL * np.abs(g[pre + "loss_remove_bias"]))


def test_centre_tap_only_for_length_one(oracle):
    """On [n, C, 1] inputs Conv1d(k=3, p=1) reduces to the centre tap."""
    rs = np.random.RandomState(0)
    x = rs.standard_normal((3, 8, 1)).astype(np.float32)
    w = rs.standard_normal((4, 8, 3)).astype(np.float32)
    b = rs.standard_normal(4).astype(np.float32)
    full = oracle.conv1d_same_k3(x, w, b)
    centre = np.einsum("oc,ncl->nol", w[:, :, 1], x) + b[None, :, None]
    assert np.allclose(full, centre, rtol=1e-6, atol=1e-6)
