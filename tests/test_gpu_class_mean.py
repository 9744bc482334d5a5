import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("N,F", [(0, 64), (1, 64), (57, 100), (473, 1024), (2048, 1024)])
def test_class_mean_vs_oracle(oracle, N, F):
    from feature_intertwiner_amd.intertwiner import class_mean
    rs = np.random.RandomState(N + F)
    feats = rs.standard_normal((N, F)).astype(np.float32)
    gt = rs.randint(0, 81, N).astype(np.int32)
    if N > 10:
        gt[:5] = 0
    ef, ec = oracle.class_mean(feats, gt, 81)
    x = torch.from_numpy(feats).to(DEV).requires_grad_(True)
    f, c = class_mean(x, torch.from_numpy(gt).to(DEV), 81)
    assert f.shape == (F, 81) and c.shape == (1, 81)
    assert np.array_equal(c.cpu().numpy(), ec)
    assert np.allclose(f.detach().cpu().numpy(), ef, rtol=1e-5, atol=1e-6)
    assert torch.all(f[:, 0] == 0)                      # background skipped
    if N:
        w = torch.from_numpy(rs.standard_normal((F, 81)).astype(np.float32)).to(DEV)
        (f * w).sum().backward()
        cnt = np.maximum(ec[0], 1)
        exp = (w.cpu().numpy()[:, gt] / cnt[gt]).T * (gt > 0)[:, None]
        assert np.allclose(x.grad.cpu().numpy(), exp, rtol=1e-5, atol=1e-7)
