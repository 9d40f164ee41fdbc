"""GPU tier: the reference's Splatter / Gaussian3ds class API (splatter.py) on the fused path, driven the way the
reference's own train.py drives it: torch.optim.Adam over gaussian_3ds' nn.Parameters, an autograd loss on
``splatter(camera_id)``, ``adaptive_control`` + a fresh optimizer, ``switch_resolution``, the viewer call."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", "tools"), os.path.join(HERE, "..", "3d-gaussian-splatting_amd")]

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capture(tmp_path_factory):
    import make_synthetic_colmap as msc

    root = str(tmp_path_factory.mktemp("capture"))
    scene, cams = msc.build(root, n=5000, width=160, height=112, views=9, points=2000, downsample=(1, 2), seed=8)
    return root, scene, cams


def make(capture, **kw):
    from splatter import Splatter

    root = capture[0]
    args = dict(render_downsample=2, opa_init_value=0.3, scale_init_value=1, tile_culling_prob_thresh=0.05)
    args.update(kw)
    return Splatter(os.path.join(root, "sparse", "0"), os.path.join(root, "images_2"), **args)


def test_constructor_state_matches_the_reference_conventions(capture):
    import gs_colmap

    sp = make(capture)
    g = sp.gaussian_3ds
    assert all(isinstance(t, torch.nn.Parameter) for t in (g.pos, g.rgb, g.opa, g.quat, g.scale))
    assert sp.n_gaussians == 2000 and g.rgb.shape == (2000, 3) and len(sp.imgs) == 9
    assert sp.imgs[0].dtype == torch.uint8 and sp.imgs[0].shape == (56, 80, 3)
    assert sp.ground_truth.dtype == torch.float16 and sp.ground_truth.shape == (56, 80, 3)
    want = gs_colmap.initial_gaussians(sp.points3d, 1, 0.3, "abs", False)
    for t, w in zip((g.pos, g.quat, g.scale, g.opa, g.rgb), want):
        assert np.array_equal(t.detach().cpu().numpy(), w)
    assert (sp.tile_info.width, sp.tile_info.height) == (80, 56) and sp.tile_info.focal_x == 0.75 * 160 / 2
    assert make(capture, use_sh_coeff=True).gaussian_3ds.rgb.shape == (2000, 27)
    assert make(capture, tile_culling_method="dist")._renderer.tile_culling_method == 0  # Splatter.__init__'s default
    with pytest.raises(ValueError):
        make(capture, tile_culling_method="nearest")


def test_forward_is_the_oracle_frame_and_backward_reaches_the_parameters(capture):
    from gs_scene import Scene
    from gs_testutil import OracleFrame

    sp = make(capture)
    img = sp(3)
    assert img.shape == (56, 80, 3) and img.requires_grad
    g = sp.gaussian_3ds
    sc = Scene(*(t.detach().cpu().numpy() for t in (g.pos, g.quat, g.scale, g.opa, g.rgb)))
    of = OracleFrame(sc, sp._camera)
    assert np.abs(img.detach().cpu().numpy() - of.image).max() < 5e-5
    assert np.array_equal(sp.culling_mask.cpu().numpy(), of.mask) and sp.n_tile_gaussians == len(of.ids)
    w = torch.randn_like(img)
    (img * w).sum().backward()
    ref = of.backward(w.cpu().numpy())
    for t, name in ((g.pos, "pos"), (g.quat, "quat"), (g.scale, "scale"), (g.opa, "opa"), (g.rgb, "rgb")):
        err = np.abs(t.grad.cpu().numpy() - ref[name]).max() / (np.abs(ref[name]).max() + 1e-30)
        assert err < 3e-4, (name, err)


def test_reference_style_training_loop(capture):
    """train.py:59-67, 84-185 in miniature with torch's own Adam and an autograd L1 loss."""
    sp = make(capture)
    g = sp.gaussian_3ds

    def optimizer():
        return torch.optim.Adam([{"params": g.opa, "lr": 0.03}, {"params": g.rgb, "lr": 0.03},
                                 {"params": g.pos, "lr": 0.003}, {"params": g.scale, "lr": 0.003},
                                 {"params": g.quat, "lr": 0.003}], betas=(0.9, 0.99))

    opt = optimizer()
    rng = np.random.default_rng(0)
    accum = torch.zeros_like(g.pos)
    losses = []
    for it in range(150):
        opt.zero_grad()
        cid = int(rng.integers(1, 8))
        loss = (sp(cid) - sp.ground_truth).abs().mean()
        loss.backward()
        opt.step()
        accum = torch.max(g.pos.grad.abs(), accum)
        losses.append(float(loss.detach()))
        if it == 100:
            n0 = sp.n_gaussians
            kept, cloned, split = g.adaptive_control(accum, taus=0.02, delete_thresh=1.5, grad_thresh=1e-7,
                                                     use_clone=True, use_split=True)
            assert sp.n_gaussians == kept + cloned + split != n0 and isinstance(g.pos, torch.nn.Parameter)
            opt, accum = optimizer(), torch.zeros_like(g.pos)
        if it == 120:
            sp.switch_resolution(1)
            assert sp.ground_truth.shape == (112, 160, 3)
    assert np.isfinite(losses).all() and np.mean(losses[-10:]) < 0.6 * np.mean(losses[:10])
    # held-out view 0 and the viewer call (visergui.py:137-149) at a size that is not a multiple of 16
    with torch.no_grad():
        test_img = sp(0)
        mse = torch.mean((test_img - sp.ground_truth) ** 2).item()
        assert 10 * np.log10(1 / mse) > 18
        cam = sp._camera
        view = sp(None, extrinsics={"rot": cam.rot, "tran": cam.tran},
                  intrinsics={"width": 150, "height": 90, "focal_x": 100.0, "focal_y": 100.0})
        assert view.shape == (90, 150, 3) and not view.requires_grad and bool(torch.isfinite(view).all())
    g.reset_opa()
    assert float((g.opa.detach() - float(np.log(0.01 / 0.99))).abs().max()) < 1e-6
