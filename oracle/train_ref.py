"""CPU restatement of the per-iteration training arithmetic around the rasterizer.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline may use it; the
product path never does).

* ``adam_step``      -- torch.optim.Adam (single-tensor path, no weight decay / amsgrad), the optimizer of
                        the reference trainer (train.py:59-67, :113), restated from torch/optim/adam.py
                        ``_single_tensor_adam``.
* ``l1_ssim_loss``   -- the loss of train.py:99-107 with torchmetrics' StructuralSimilarityIndexMeasure
                        (train.py:71).  torchmetrics is an unpinned third-party dependency that is absent from
                        /root/reference and from this image; the algorithm is restated from its published
                        source (functional/image/ssim.py ``_ssim_update``): Gaussian window size 11 /
                        sigma 1.5, reflect padding by 5, VALID convolution, crop by 5, k1 0.01, k2 0.03,
                        variances clamped at 0, mean over the remaining pixels and channels.  Pinned in
                        tests/test_train_ref.py against a literal torch restatement of that function run
                        through torch.autograd (torch is available; torchmetrics is not): parity with
                        torchmetrics itself is therefore UNPINNED beyond the restated source.
"""
from __future__ import annotations

import numpy as np


def adam_step(p, g, m, v, lr, beta1=0.9, beta2=0.99, eps=1e-8, step=1):
    """One step in fp32 with torch's operation order; returns new (p, m, v)."""
    f = np.float32
    p, g, m, v = (np.asarray(a, f) for a in (p, g, m, v))
    m = m + (g - m) * f(1.0 - beta1)                      # exp_avg.lerp_(grad, 1 - beta1)
    v = v * f(beta2) + f(1.0 - beta2) * (g * g)           # exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2_sqrt = np.sqrt(1.0 - beta2 ** step)
    denom = np.sqrt(v) / f(bc2_sqrt) + f(eps)             # (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    p = p - f(lr / bc1) * (m / denom)                     # param.addcdiv_(exp_avg, denom, value=-step_size)
    return p.astype(f), m.astype(f), v.astype(f)


def gaussian_window(size=11, sigma=1.5, dtype=np.float64):
    d = np.arange((1 - size) / 2, (1 + size) / 2, 1, dtype=dtype)
    g = np.exp(-((d / sigma) ** 2) / 2)
    return g / g.sum()


def _filt_valid(a, g):
    """Separable VALID correlation of a [H,W,C] array with the 1-D window g along H and W."""
    k = len(g)
    H, W = a.shape[:2]
    tmp = sum(g[i] * a[:, i:W - k + 1 + i] for i in range(k))
    return sum(g[i] * tmp[i:H - k + 1 + i] for i in range(k))


def _filt_full_adjoint(d, g, H, W):
    """Adjoint of _filt_valid: scatters d [H-k+1, W-k+1, C] back onto [H, W, C]."""
    k = len(g)
    h, w = d.shape[:2]
    tmp = np.zeros((H, w) + d.shape[2:], d.dtype)
    for i in range(k):
        tmp[i:i + h] += g[i] * d
    out = np.zeros((H, W) + d.shape[2:], d.dtype)
    for i in range(k):
        out[:, i:i + w] += g[i] * tmp
    return out


def l1_ssim_loss(pred, target, ssim_weight=0.1, dtype=np.float64):
    """Returns (loss, l1, ssim, dloss/dpred) for [H,W,3] images."""
    x, y = np.asarray(pred, dtype), np.asarray(target, dtype)
    H, W, Cn = x.shape
    l1 = np.abs(x - y).mean()
    grad = (1.0 - ssim_weight) * np.sign(x - y) / x.size
    ssim = 0.0
    if ssim_weight > 0:
        g = gaussian_window(dtype=dtype)
        c1, c2 = 0.01 ** 2, 0.03 ** 2
        mu, nu = _filt_valid(x, g), _filt_valid(y, g)
        exx, eyy, exy = _filt_valid(x * x, g), _filt_valid(y * y, g), _filt_valid(x * y, g)
        vx_raw, vy_raw = exx - mu * mu, eyy - nu * nu
        vx, vy = np.maximum(vx_raw, 0), np.maximum(vy_raw, 0)
        A1, A2 = 2 * mu * nu + c1, 2 * (exy - mu * nu) + c2
        B1, B2 = mu * mu + nu * nu + c1, vx + vy + c2
        S = A1 * A2 / (B1 * B2)
        ssim = S.mean()
        dExx = np.where(vx_raw > 0, -S / B2, 0.0)
        dmu = 2 * nu * (A2 - A1) / (B1 * B2) - 2 * mu * S / B1 - 2 * mu * dExx
        dExy = 2 * A1 / (B1 * B2)
        dS_dx = _filt_full_adjoint(dmu, g, H, W) + 2 * x * _filt_full_adjoint(dExx, g, H, W) \
            + y * _filt_full_adjoint(dExy, g, H, W)
        grad = grad - ssim_weight * dS_dx / S.size
    loss = (1.0 - ssim_weight) * l1 + (ssim_weight * (1.0 - ssim) if ssim_weight > 0 else 0.0)
    return float(loss), float(l1), float(ssim), grad
