"""CPU: oracle/densify_ref.py against the reference's own adaptive_control, recorded in tests/golden/densify.npz
(tests/golden/make_golden.py::densify ran Gaussian3ds.adaptive_control of /root/reference/splatter.py as is)."""
import os

import numpy as np
import pytest
import torch

from oracle import densify_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden", "densify.npz")


def reference_draws(seed, n_split):
    """The two standard-normal blocks MultivariateNormal.sample() consumed in the recorded run."""
    torch.manual_seed(int(seed))
    return [torch.normal(torch.zeros(n_split, 3), torch.ones(n_split, 3)).numpy() for _ in range(2)]


def run_case(g, ci, fn=densify_ref.adaptive_control):
    act, agg, use_clone, use_split = g[f"c{ci}_cfg"]
    args = [g[f"c{ci}_{k}"] for k in ("pos", "quat", "scale", "opa", "rgb", "grad")]
    kw = dict(scale_activation=str(act), grad_thresh=0.0002, grad_aggregation=str(agg), use_clone=bool(int(use_clone)),
              use_split=bool(int(use_split)), clone_dt=0.01)
    # first pass with dummy draws to learn n_split, then the recorded draws of that shape
    n = len(args[0])
    *_, counts = fn(*args, 0.05, 0.17, np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), **kw)
    e1, e2 = reference_draws(g[f"c{ci}_seed"], counts[2]) if counts[2] else (np.zeros((0, 3)), np.zeros((0, 3)))
    return fn(*args, 0.05, 0.17, e1, e2, **kw)


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_oracle_matches_reference_adaptive_control(ci):
    g = np.load(GOLD)
    pos, quat, scale, opa, rgb, counts = run_case(g, ci)
    want = [g[f"c{ci}_out_{k}"] for k in ("pos", "quat", "scale", "opa", "rgb")]
    assert len(pos) == len(want[0]) == sum(counts)
    for got, w, name in zip((pos, quat, scale, opa, rgb), want, ("pos", "quat", "scale", "opa", "rgb")):
        assert got.shape == w.shape, name
        if name == "pos":  # samples go through a 3x3 Cholesky factor: a few ulps
            assert np.abs(got - w).max() < 2e-6 * max(1.0, np.abs(w).max()), name
        else:
            assert np.array_equal(got, w), name
    assert counts[2] > 0 or not bool(int(g[f"c{ci}_cfg"][3]))


def test_reset_opa_value():
    assert np.allclose(densify_ref.reset_opa(3), -np.log(1 / 0.01 - 1))
