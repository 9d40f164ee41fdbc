"""GPU: gs_densify_classify / gs_densify_apply against the reference's own adaptive_control (golden vectors
recorded from /root/reference/splatter.py) and against the oracle on larger random sets."""

import numpy as np
import pytest
import torch

from oracle import densify_ref
from test_densify_ref import GOLD, reference_draws

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def run_gpu(gpu, arrays, grad, draws, **kw):
    from gs_densify import adaptive_control

    params = [torch.from_numpy(np.ascontiguousarray(a)).to(gpu) for a in arrays]
    d = tuple(torch.from_numpy(np.ascontiguousarray(e, dtype=np.float32)).to(gpu) for e in draws)
    out, counts = adaptive_control(params, torch.from_numpy(grad).to(gpu), 0.05, 0.17, draws=d, **kw)
    return [t.cpu().numpy() for t in out], counts


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_matches_reference_golden(gpu, ci):
    g = np.load(GOLD)
    act, agg, use_clone, use_split = g[f"c{ci}_cfg"]
    kw = dict(scale_activation=str(act), grad_thresh=0.0002, grad_aggregation=str(agg), use_clone=bool(int(use_clone)),
              use_split=bool(int(use_split)), clone_dt=0.01)
    arrays = [g[f"c{ci}_{k}"] for k in ("pos", "quat", "scale", "opa", "rgb")]
    want = [g[f"c{ci}_out_{k}"] for k in ("pos", "quat", "scale", "opa", "rgb")]
    # the recorded run consumed two (n_split, 3) normal blocks; n_split from the oracle
    *_, counts = densify_ref.adaptive_control(*arrays, g[f"c{ci}_grad"], 0.05, 0.17, np.zeros((600, 3)),
                                              np.zeros((600, 3)), **kw)
    draws = reference_draws(g[f"c{ci}_seed"], counts[2]) if counts[2] else [np.zeros((1, 3)), np.zeros((1, 3))]
    got, gcounts = run_gpu(gpu, arrays, g[f"c{ci}_grad"], draws, **kw)
    assert gcounts == counts
    for a, w, name in zip(got, want, ("pos", "quat", "scale", "opa", "rgb")):
        assert a.shape == w.shape, name
        if name == "pos":
            assert np.abs(a - w).max() < 3e-6 * max(1.0, np.abs(w).max())
        else:
            assert np.array_equal(a, w), name


@pytest.mark.parametrize("n,color_dim,act", [(1, 3, "abs"), (257, 3, "abs"), (100_003, 3, "exp"), (20_000, 27, "abs"),
                                              (5_000, 48, "abs")])
def test_matches_oracle_random(gpu, n, color_dim, act):
    rng = np.random.default_rng(n)
    pos = rng.normal(size=(n, 3)).astype(np.float32)
    quat = rng.normal(size=(n, 4)).astype(np.float32)
    scale = (rng.uniform(0.005, 0.12, size=(n, 3)) * rng.choice([-1, 1], size=(n, 3))).astype(np.float32)
    if act == "exp":
        scale = np.log(np.abs(scale)).astype(np.float32)
    opa = rng.normal(-2.0, 2.5, size=n).astype(np.float32)
    rgb = rng.normal(size=(n, color_dim)).astype(np.float32)
    grad = (rng.normal(size=(n, 3)) * 3e-4).astype(np.float32)
    e1, e2 = rng.normal(size=(n, 3)).astype(np.float32), rng.normal(size=(n, 3)).astype(np.float32)
    kw = dict(scale_activation=act, grad_aggregation="max")
    want = densify_ref.adaptive_control(pos, quat, scale, opa, rgb, grad, 0.05, 0.17, e1, e2, **kw)
    got, counts = run_gpu(gpu, [pos, quat, scale, opa, rgb], grad, (e1, e2), **kw)
    assert counts == want[5] and sum(counts) == len(got[0])
    for a, w, name in zip(got, want[:5], ("pos", "quat", "scale", "opa", "rgb")):
        if name == "pos":
            assert np.abs(a - w).max() < 3e-6 * max(1.0, np.abs(w).max())
        elif name == "scale" and act == "exp":
            assert np.abs(a - w).max() < 1e-6  # expf in the size test may flip a borderline class? no: values only
        else:
            assert np.array_equal(a, w), name


def test_all_deleted_and_reset_opa(gpu):
    from gs_densify import adaptive_control, inverse_sigmoid, reset_opa

    n = 1000
    params = [torch.randn(n, 3, device=gpu), torch.randn(n, 4, device=gpu), torch.rand(n, 3, device=gpu) * 0.01,
              torch.full((n,), -10.0, device=gpu), torch.randn(n, 3, device=gpu)]
    out, counts = adaptive_control(params, torch.zeros(n, 3, device=gpu), 0.05, 0.17)
    assert counts == (0, 0, 0) and all(t.shape[0] == 0 for t in out)
    o = reset_opa(torch.randn(50, device=gpu))
    assert float((o - inverse_sigmoid(0.01)).abs().max()) < 1e-6
