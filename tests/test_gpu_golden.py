"""GPU tier: the HIP kernels (through the `gaussian` drop-in, i.e. the C ABI) DIRECTLY against
golden vectors produced by the reference's own kernels (tests/golden/, see make_golden.py) --
no oracle in between the two RESULTS.  Same tolerances as test_gpu_kernels.py: the gradients element by element in
units of each element's conditioning scale (the oracle supplies that scale -- a property of the inputs -- only)."""
import os

import numpy as np
import pytest
import torch

import oracle
from gs_testutil import assert_rows_close, grad_close

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dev(a, gpu, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t.to(dtype) if dtype is not None else t).to(gpu)


@pytest.fixture(scope="module", params=["nosh", "sh", "dense_fwd"])
def gold(request):
    return np.load(os.path.join(GOLD, f"kernels_{request.param}.npz"))


def test_global_culling_vs_reference_kernel(gpu, gold):
    import gaussian

    n = gold["pos"].shape[0]
    res_pos, res_cov = torch.zeros(n, 3, device=gpu), torch.zeros(n, 2, 2, device=gpu)
    mask = torch.zeros(n, dtype=torch.long, device=gpu)
    args = [dev(gold[k], gpu) for k in ("pos", "quat", "scale", "rot", "tran")]
    gaussian.global_culling(*args, res_pos, res_cov, mask, float(gold["near"]), float(gold["half_w"]),
                            float(gold["half_h"]))
    assert np.array_equal(mask.cpu().numpy(), gold["k1_mask"])
    assert np.array_equal(res_pos.cpu().numpy().view(np.uint32), gold["k1_pos"].view(np.uint32))
    assert np.array_equal(res_cov.cpu().numpy().view(np.uint32), gold["k1_cov"].view(np.uint32))
    outs = [torch.zeros(n, k, device=gpu) for k in (3, 4, 3)]
    gaussian.global_culling_backward(*args, dev(gold["k2_gop"], gpu), dev(gold["k2_goc"], gpu), mask, *outs)
    S = oracle.global_culling_backward_scale(gold["pos"], gold["quat"], gold["scale"], gold["rot"], gold["tran"],
                                             np.abs(gold["k2_gop"]), np.abs(gold["k2_goc"]).reshape(-1, 4), gold["k1_mask"])
    for o, sc, key in zip(outs, S, ("k2_gpos", "k2_gquat", "k2_gscale")):
        ok, worst, where, _ = grad_close(o.cpu().numpy(), gold[key], sc, rtol=1e-5, kappa=2e-6)
        assert ok, (key, "worst err/tol", worst, "at", where)


def test_tile_binning_vs_reference_kernel(gpu, gold):
    import gaussian

    keep = gold["k1_mask"].astype(bool)
    g3 = gaussian.Gaussian3ds()
    g3.pos, g3.cov = dev(gold["k1_pos"][keep], gpu), dev(gold["k1_cov"][keep], gpu)
    tlx, tly, ntx, nty, leftmost, topmost = gold["k3_geom"]
    T, maxp = int(ntx * nty), int(gold["k3_maxp"])
    cnt = torch.zeros(T, dtype=torch.int32, device=gpu)
    lst = torch.full((T, maxp), -1, dtype=torch.int32, device=gpu)
    gaussian.calc_tile_list(g3, gaussian.Tiles(), cnt, lst, 0.05, 2, tlx, tly, int(ntx), int(nty), leftmost, topmost)
    c, l = cnt.cpu().numpy(), lst.cpu().numpy()
    ref_c, ref_l = gold["k3_m2_count"], gold["k3_m2_list"]
    assert np.array_equal(np.minimum(c, maxp), np.minimum(ref_c, maxp))
    for t in range(T):
        if ref_c[t] < maxp:
            assert np.array_equal(np.sort(l[t, :c[t]]), np.sort(ref_l[t, :ref_c[t]])), t


def test_draw_vs_reference_kernel(gpu, gold):
    from renderer import draw

    use_sh = bool(gold["use_sh"])
    img_ref = gold["k7_image"]
    t = [dev(gold[k], gpu).requires_grad_(True) for k in ("k7_pos", "k7_rgb", "k7_opa")]
    cov = dev(gold["k7_cov"].reshape(-1, 2, 2), gpu).requires_grad_(True)
    rays = [dev(gold[k], gpu) for k in ("rays_o", "lefttop", "vdx", "vdy")]
    img = draw(*t, cov, dev(gold["k7_accum"], gpu), img_ref.shape[0], img_ref.shape[1], float(gold["fx"]),
               float(gold["fy"]), False, False, use_sh, True, *rays)
    assert np.abs(img.detach().cpu().numpy() - img_ref).max() < 5e-5
    if "k8_gpos" not in gold.files:
        return
    img.backward(dev(gold["k8_grad_output"], gpu))
    _, scale = oracle.draw_backward(gold["k7_pos"], gold["k7_rgb"], gold["k7_opa"], gold["k7_cov"], gold["k7_accum"],
                                    img_ref, gold["k8_grad_output"], float(gold["fx"]), float(gold["fy"]),
                                    use_sh=use_sh, fast=True, rays_o=gold["rays_o"], lefttop=gold["lefttop"],
                                    vdx=gold["vdx"], vdy=gold["vdy"], with_scale=True)
    want = [gold[k] for k in ("k8_gpos", "k8_grgb", "k8_gopa", "k8_gcov")]
    assert_rows_close([x.grad.cpu().numpy() for x in (*t, cov)], want, scale, "vs reference kernel")
