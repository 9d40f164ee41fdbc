"""CPU tests of the training-step oracle (oracle/train_ref.py) and of the host-side schedule mirror."""
import math
import sys

import numpy as np
import pytest
import torch

from oracle import train_ref


def torchmetrics_ssim_restated(preds, target, data_range=1.0, k1=0.01, k2=0.03, sigma=1.5, size=11):
    """Literal restatement of torchmetrics.functional.image.ssim._ssim_update (+ elementwise_mean) in torch.
    preds/target: [B, C, H, W]."""
    import torch.nn.functional as F

    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    channel = preds.size(1)
    dist = torch.arange((1 - size) / 2, (1 + size) / 2, 1, dtype=preds.dtype)
    gauss = torch.exp(-torch.pow(dist / sigma, 2) / 2)
    g1 = (gauss / gauss.sum()).unsqueeze(0)
    kernel = torch.matmul(g1.t(), g1).expand(channel, 1, size, size)
    pad = (size - 1) // 2
    preds = F.pad(preds, (pad, pad, pad, pad), mode="reflect")
    target = F.pad(target, (pad, pad, pad, pad), mode="reflect")
    inp = torch.cat((preds, target, preds * preds, target * target, preds * target))
    out = F.conv2d(inp, kernel, groups=channel).split(preds.shape[0])
    mu_p2, mu_t2, mu_pt = out[0].pow(2), out[1].pow(2), out[0] * out[1]
    s_p2 = torch.clamp(out[2] - mu_p2, min=0.0)
    s_t2 = torch.clamp(out[3] - mu_t2, min=0.0)
    s_pt = out[4] - mu_pt
    upper, lower = 2 * s_pt + c2, s_p2 + s_t2 + c2
    full = ((2 * mu_pt + c1) * upper) / ((mu_p2 + mu_t2 + c1) * lower)
    idx = full[..., pad:-pad, pad:-pad]
    return idx.reshape(idx.shape[0], -1).mean(-1).mean()


@pytest.mark.parametrize("h,w,weight", [(24, 31, 0.1), (40, 40, 1.0), (17, 50, 0.5)])
def test_loss_oracle_matches_autograd_of_restated_torchmetrics(h, w, weight):
    rng = np.random.default_rng(h * 100 + w)
    x = rng.uniform(0, 1, (h, w, 3))
    y = np.clip(x + rng.normal(0, 0.1, x.shape), 0, 1)
    loss, l1, ssim, grad = train_ref.l1_ssim_loss(x, y, weight)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    yt = torch.tensor(y, dtype=torch.float64)
    l1_t = (xt - yt).abs().mean()
    ssim_t = torchmetrics_ssim_restated(xt.unsqueeze(0).permute(0, 3, 1, 2), yt.unsqueeze(0).permute(0, 3, 1, 2))
    loss_t = (1 - weight) * l1_t + weight * (1.0 - ssim_t)  # train.py:100-107
    loss_t.backward()
    assert abs(l1 - l1_t.item()) < 1e-14 and abs(ssim - ssim_t.item()) < 1e-12
    assert abs(loss - loss_t.item()) < 1e-12
    assert np.abs(grad - xt.grad.numpy()).max() < 1e-13


def test_ssim_known_answers():
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 1, (30, 30, 3))
    _, l1, ssim, grad = train_ref.l1_ssim_loss(x, x, 0.1)
    assert l1 == 0 and abs(ssim - 1.0) < 1e-14  # identical images
    assert np.abs(grad).max() < 1e-12           # SSIM is stationary there and sign(0) = 0
    _, _, ssim2, _ = train_ref.l1_ssim_loss(np.full((30, 30, 3), 0.5), np.full((30, 30, 3), 0.25), 1.0)
    c1 = 1e-4
    assert abs(ssim2 - (2 * 0.5 * 0.25 + c1) / (0.25 + 0.0625 + c1)) < 1e-12  # constant images: luminance term only
    g = train_ref.gaussian_window()
    assert len(g) == 11 and abs(g.sum() - 1) < 1e-15 and np.allclose(g, g[::-1])


def test_adam_oracle_matches_torch_optim():
    rng = np.random.default_rng(11)
    shapes = [(37, 4), (37, 3), (37,), (37, 3)]
    lrs = [0.03, 0.003, 0.0005, 0.01]
    ps = [torch.tensor(rng.normal(size=s).astype(np.float32), requires_grad=True) for s in shapes]
    opt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ps, lrs)], betas=(0.9, 0.99), foreach=False)
    state = [(p.detach().numpy().copy(), np.zeros(s, np.float32), np.zeros(s, np.float32)) for p, s in zip(ps, shapes)]
    for step in range(1, 8):
        grads = [(rng.normal(size=s) * 10.0 ** rng.integers(-4, 1)).astype(np.float32) for s in shapes]
        for p, g in zip(ps, grads):
            p.grad = torch.tensor(g)
        opt.step()
        state = [train_ref.adam_step(p0, g, m, v, lr, 0.9, 0.99, 1e-8, step)
                 for (p0, m, v), g, lr in zip(state, grads, lrs)]
        for p, (p0, m, v), grp in zip(ps, state, opt.param_groups):
            st = opt.state[grp["params"][0]]
            assert np.allclose(st["exp_avg"].numpy(), m, rtol=1e-5, atol=3e-7 * np.abs(m).max())
            assert np.allclose(st["exp_avg_sq"].numpy(), v, rtol=1e-5, atol=3e-7 * np.abs(v).max())
            assert np.allclose(p.detach().numpy(), p0, rtol=0, atol=1e-6 * max(1.0, np.abs(p0).max()))  # a few ulps over 7 steps


def test_lr_schedule_mirror():
    """gs_train.lr_lambdas / base_lrs against the formulas of train.py:20-58 written out independently."""
    sys.path.insert(0, "3d-gaussian-splatting_amd")
    import importlib

    # gs_train imports the HIP library loader; the schedule itself is pure Python
    gs_train = importlib.import_module("gs_train")
    opt = gs_train.TrainOptions()
    assert gs_train.base_lrs(opt) == [0.03, 0.03, 0.003, 0.003, 0.003]
    gamma = 0.01 ** (1 / (7001 - 300))
    for mode, expect in (("exp", [True] * 5), ("official", [True, False, True, False, False])):
        opt.lr_decay = mode
        fs = gs_train.lr_lambdas(opt)
        for f, decays in zip(fs, expect):
            assert f(0) == 0 and f(150) == 0.5 and f(300) == 1.0
            assert math.isclose(f(1300), gamma ** 1000 if decays else 1.0, rel_tol=1e-12)
    opt.lr_decay = "none"
    f = gs_train.lr_lambdas(opt)[2]
    assert f(300) == 1 and f(2299) == 1 and math.isclose(f(2300), 0.2) and math.isclose(f(4300), 0.04)
