"""CPU restatement of the reference's densification step.  TEST INFRASTRUCTURE ONLY.

``adaptive_control`` follows Gaussian3ds.adaptive_control (splatter.py:122-228) line by line:
  1. prune:   keep = (opa > logit(0.02)) & (||act(scale)|| < delete_thresh)                (:143-155)
  2. select:  densify = aggregate(|grad|) > grad_thresh over the kept Gaussians            (:158-162)
  3. clone:   small ones (||act(scale)|| <= taus): copy with pos - grad * clone_dt          (:176-188)
  4. split:   large ones: scale /= 1.6 (abs) or -= log 1.6 (exp), the original moves to a first
              sample and a second sample is appended; samples are N(pos, R S^2 R^T) drawn as
              pos + chol(Sigma) @ eps, which is what torch's MultivariateNormal.sample() does
              (utils.py:391-402); the two normal draws eps1, eps2 are INPUTS here             (:190-222)
  output order: kept (split ones modified), clones, second split samples                      (:223-227)
Pinned against the reference's own Python code run in the build container
(tests/golden/make_golden.py::densify -> tests/golden/densify.npz).
"""
from __future__ import annotations

import math

import numpy as np

EPS = 1e-4  # splatter.py: `_scale = self.scale.abs() + EPS`


def inverse_sigmoid(y):  # utils.py:350-351
    return -math.log(1 / y - 1)


def q2r(q):  # utils.py:318-333
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    r = [1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y,
         2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x,
         2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]
    return np.stack(r, axis=1).reshape(-1, 3, 3)


def cov3d(quat, scale, scale_activation):  # splatter.py:100-114 (trunc_exp forward = exp)
    s = np.abs(scale) + np.float32(EPS) if scale_activation == "abs" else np.exp(scale)
    RS = q2r(quat) * s[:, None, :]
    return RS @ RS.transpose(0, 2, 1)


def adaptive_control(pos, quat, scale, opa, rgb, grad, taus, delete_thresh, eps1, eps2, scale_activation="abs",
                     grad_thresh=0.0002, grad_aggregation="max", use_clone=True, use_split=True, clone_dt=0.01,
                     dtype=np.float32):
    """Returns (pos, quat, scale, opa, rgb, counts) with counts = (kept, cloned, split)."""
    pos, quat, scale, opa, rgb, grad = (np.asarray(a, dtype) for a in (pos, quat, scale, opa, rgb, grad))
    act = np.abs(scale) if scale_activation == "abs" else np.exp(scale)
    # NB the prune / size tests use ||scale|| WITHOUT the +EPS of the covariance (splatter.py:143-145, 170)
    norm = np.sqrt((act * act).sum(-1, dtype=dtype))
    keep = (opa > dtype(inverse_sigmoid(0.02))) & (norm < dtype(delete_thresh))
    pos, quat, scale, opa, rgb, grad, norm = (a[keep] for a in (pos, quat, scale, opa, rgb, grad, norm))
    agg = np.abs(grad).max(-1) if grad_aggregation == "max" else np.abs(grad).mean(-1, dtype=dtype)
    densify = agg > dtype(grad_thresh)
    out = [[pos.copy()], [quat.copy()], [scale.copy()], [opa.copy()], [rgb.copy()]]
    n_clone = n_split = 0
    if densify.any():
        split = (norm > dtype(taus)) & densify
        clone = (norm <= dtype(taus)) & densify
        if clone.any() and use_clone:
            n_clone = int(clone.sum())
            out[0].append(pos[clone] - grad[clone] * dtype(clone_dt))
            for k, a in zip((1, 2, 3, 4), (quat, scale, opa, rgb)):
                out[k].append(a[clone].copy())
        if split.any() and use_split:
            n_split = int(split.sum())
            new_scale = scale.copy()
            if scale_activation == "abs":
                new_scale[split] = new_scale[split] / dtype(1.6)
            else:
                new_scale[split] = new_scale[split] - dtype(math.log(1.6))
            out[2][0] = new_scale
            # the covariance is built from the ORIGINAL scale (get_gaussian_3d_cov reads self.scale, :202)
            L = np.linalg.cholesky(cov3d(quat[split], scale[split], scale_activation).astype(dtype))
            p1 = pos[split] + np.einsum("nij,nj->ni", L, np.asarray(eps1, dtype)[:n_split])
            p2 = pos[split] + np.einsum("nij,nj->ni", L, np.asarray(eps2, dtype)[:n_split])
            out[0][0][split] = p1
            out[0].append(p2)
            out[1].append(quat[split].copy())
            out[2].append(new_scale[split].copy())
            out[3].append(opa[split].copy())
            out[4].append(rgb[split].copy())
    res = [np.concatenate(o).astype(dtype) for o in out]
    return (*res, (int(keep.sum()), n_clone, n_split))


def reset_opa(n, dtype=np.float32):  # splatter.py:119-120
    return np.full(n, inverse_sigmoid(0.01), dtype)
