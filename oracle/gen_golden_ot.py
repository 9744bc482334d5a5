"""Generate golden vectors for the OT intertwiner loss by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference); the fixtures it writes
(tests/golden/ot_*.npz) are data -- inputs, weights (or the seed that regenerates
them) and the reference's outputs.  No reference source is copied.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_ot.py

What is captured
  ot_sinkhorn.npz : OptTrans._sinkhorn_iterate (lib/OT_module.py:104-135) on raw
                    [S,D] inputs -- cosine and l2 cost, D=1 and D>1, L in {5,50}.
  ot_1d_small.npz : OptTrans.forward (1-D 'conv' form, :67-102) with ch_x=64 and the
                    full state_dict stored.
  ot_1d_full.npz  : OptTrans.forward at the model's size (ch_x=1024, critic 256) for
                    n in {1,12} and L in {5,50}; weights/inputs are regenerated from a
                    numpy RandomState seed (legacy generator: stable across versions)
                    so the fixture stays small.
  ot_2d.npz       : two-dim form (ConvTranspose2d/BN/ReLU G_net, 2x conv-BN-ReLU critic)
                    in eval mode, state_dict stored.
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("FI_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

# the reference calls .cuda() unconditionally (lib/OT_module.py:118-119)
torch.Tensor.cuda = lambda self, *a, **k: self

from lib.OT_module import OptTrans  # noqa: E402


def cfg(form="conv"):
    c = types.SimpleNamespace()
    c.DEV = types.SimpleNamespace(OT_ONE_DIM_FORM=form)
    return c


def full_weights(seed, ch=1024):
    """Deterministic weights for the full-size 1-D module (see tests/helpers)."""
    rs = np.random.RandomState(seed)
    g_w = (rs.standard_normal((ch, ch, 3)) * (1.0 / np.sqrt(3 * ch))).astype(np.float32)
    g_b = (rs.standard_normal((ch,)) * 0.05).astype(np.float32)
    c_w = (rs.standard_normal((ch // 4, ch, 3)) * (1.0 / np.sqrt(3 * ch))).astype(np.float32)
    c_b = (rs.standard_normal((ch // 4,)) * 0.05).astype(np.float32)
    return g_w, g_b, c_w, c_b


def full_inputs(seed, n, ch=1024):
    rs = np.random.RandomState(seed)
    x = np.maximum(rs.standard_normal((n, ch, 1)), 0).astype(np.float32)
    y = np.maximum(rs.standard_normal((n, ch, 1)), 0).astype(np.float32)
    return x, y


def run_terms(m, x, y):
    with torch.no_grad():
        x_up = m.G_net(x)
        t_xy = m._basic_compute_loss(x_up, y)
        t_xx = m._basic_compute_loss(x_up, x_up)
        t_yy = m._basic_compute_loss(y, y)
        loss = m(x, y)
    return [t.numpy().astype(np.float32) for t in (t_xy, t_xx, t_yy, loss)]


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(2000)
    torch.set_num_threads(4)

    # ---- raw sinkhorn ------------------------------------------------------
    sk = {}
    rs = np.random.RandomState(11)
    cases = [("cos_S32_D1", 32, 1, "cosine"), ("cos_S256_D1", 256, 1, "cosine"),
             ("cos_S16_D9", 16, 9, "cosine"), ("cos_S64_D64", 64, 64, "cosine"),
             ("l2_S16_D9", 16, 9, "l2"), ("l2_S32_D1", 32, 1, "l2")]
    for name, S, D, form in cases:
        x = np.maximum(rs.standard_normal((S, D)), 0).astype(np.float32)
        y = np.maximum(rs.standard_normal((S, D)), 0).astype(np.float32)
        if D > 1:  # signed features exercise negative cosine too
            x = rs.standard_normal((S, D)).astype(np.float32)
            y = rs.standard_normal((S, D)).astype(np.float32)
        sk[name + "_x"], sk[name + "_y"] = x, y
        for L in (5, 50):
            for eps in (1.0, 0.5):
                m = OptTrans(cfg(), ch_x=8, epsilon=eps, L=L, C_form=form)
                with torch.no_grad():
                    v = m._sinkhorn_iterate(torch.from_numpy(x.copy()), torch.from_numpy(y.copy()))
                sk["%s_L%d_eps%g" % (name, L, eps)] = np.float32(v.item())
    np.savez_compressed(os.path.join(OUT, "ot_sinkhorn.npz"), **sk)

    # ---- 1-D small, state dict stored ---------------------------------------
    d = {}
    for L in (5, 50):
        m = OptTrans(cfg(), ch_x=64, L=L).eval()
        for p in m.parameters():
            torch.nn.init.normal_(p, std=0.08)
        x = torch.relu(torch.randn(7, 64, 1))
        y = torch.relu(torch.randn(7, 64, 1))
        t_xy, t_xx, t_yy, loss = run_terms(m, x, y)
        for k, v in m.state_dict().items():
            d["L%d_sd_%s" % (L, k)] = v.numpy()
        d["L%d_x" % L], d["L%d_y" % L] = x.numpy(), y.numpy()
        d["L%d_t_xy" % L], d["L%d_t_xx" % L], d["L%d_t_yy" % L], d["L%d_loss" % L] = t_xy, t_xx, t_yy, loss
        mb = OptTrans(cfg(), ch_x=64, L=L, remove_bias=True).eval()
        mb.load_state_dict(m.state_dict())
        with torch.no_grad():
            d["L%d_loss_remove_bias" % L] = mb(x, y).numpy()
    np.savez_compressed(os.path.join(OUT, "ot_1d_small.npz"), **d)

    # ---- 1-D full size, seed-regenerated ------------------------------------
    d = {"weight_seed": np.int64(77)}
    g_w, g_b, c_w, c_b = full_weights(77)
    for n, in_seed in ((1, 5), (12, 6)):
        x, y = full_inputs(in_seed, n)
        for L in (5, 50):
            m = OptTrans(cfg(), ch_x=1024, L=L).eval()
            with torch.no_grad():
                m.G_net[0].weight.copy_(torch.from_numpy(g_w)); m.G_net[0].bias.copy_(torch.from_numpy(g_b))
                m.critic[0].weight.copy_(torch.from_numpy(c_w)); m.critic[0].bias.copy_(torch.from_numpy(c_b))
            t_xy, t_xx, t_yy, loss = run_terms(m, torch.from_numpy(x), torch.from_numpy(y))
            key = "n%d_L%d" % (n, L)
            d[key + "_input_seed"] = np.int64(in_seed)
            d[key + "_t_xy"], d[key + "_t_xx"], d[key + "_t_yy"], d[key + "_loss"] = t_xy, t_xx, t_yy, loss
    np.savez_compressed(os.path.join(OUT, "ot_1d_full.npz"), **d)

    # ---- 2-D form -----------------------------------------------------------
    d = {}
    for tag, sx, sy in (("up", 4, 8), ("same", 8, 8)):
        m = OptTrans(cfg(), ch_x=16, spatial_x=sx, spatial_y=sy, L=5).eval()
        for p in m.parameters():
            torch.nn.init.normal_(p, std=0.1)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.1)
                mod.running_var.uniform_(0.5, 1.5)
        x = torch.randn(3, 16, sx, sx)
        y = torch.randn(3, 16, sy, sy)
        t_xy, t_xx, t_yy, loss = run_terms(m, x, y)
        for k, v in m.state_dict().items():
            d["%s_sd_%s" % (tag, k)] = v.numpy()
        d[tag + "_x"], d[tag + "_y"] = x.numpy(), y.numpy()
        d[tag + "_t_xy"], d[tag + "_t_xx"], d[tag + "_t_yy"], d[tag + "_loss"] = t_xy, t_xx, t_yy, loss
    np.savez_compressed(os.path.join(OUT, "ot_2d.npz"), **d)
    print("wrote fixtures to", os.path.abspath(OUT))


if __name__ == "__main__":
    main()
