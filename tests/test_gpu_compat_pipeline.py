"""GPU: the zero-change integration mode end to end.

The reference's per-frame flow (splatter.py:513-655: activations -> renderer.global_culling -> mask gathers ->
gaussian.calc_tile_list -> clamp / cumsum -> gaussian.gather_gaussians -> attribute gather -> torch.sort on the
fp32 composite key -> attribute gather -> renderer.draw -> clamp -> crop) is driven here through the drop-in
``gaussian`` and ``renderer`` modules exactly as splatter.py would call them, with plain torch ops in between, and
compared with the fused frame path and with the oracle -- image and parameter gradients."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

from gs_testutil import OracleFrame, assert_grads_close, to_torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


from compat_fps import reference_style_frame  # noqa: E402  (tools/compat_fps.py: also bench.py's compat_mode leg)


def test_reference_call_sequence_matches_frame_path_and_oracle(gpu):
    from gs_frame import FrameRenderer
    from gs_scene import make_camera, make_scene
    from gs_testutil import frame_scalars

    W, H = 250, 186  # not multiples of 16: padded to 256 x 192 (192 tiles), centred crop
    scene, cam = make_scene(3000, W, H, seed=5), make_camera(W, H, yaw_deg=2.0)
    grid, _, _, rays = frame_scalars(cam)
    of = OracleFrame(scene, cam)
    params = to_torch(scene, gpu, requires_grad=True)
    img, max_tile, maxp = reference_style_frame(params, cam, grid, rays)
    assert max_tile <= maxp, "keep the test below the reference's per-tile cap (it would drop Gaussians)"
    got = img.detach().cpu().numpy()
    assert np.abs(got - of.image).max() < 2e-4
    fused = FrameRenderer(gpu, max_pairs=1 << 16).forward(*[p.detach() for p in params], cam)[0]
    assert float((fused - img.detach()).abs().max()) < 2e-4
    gimg = np.random.default_rng(2).normal(size=of.image.shape).astype(np.float32)
    gimg, _ = of.robust_grad_image(gimg)
    img.backward(torch.from_numpy(gimg).to(gpu))
    ref, scale = of.backward(gimg, with_scale=True)
    for t, name in zip(params, ("pos", "quat", "scale", "opa", "rgb")):
        assert t.grad is not None and bool(torch.isfinite(t.grad).all()), name
    # element by element, the frame path's standard (the reference's torch.sort on the fp32 composite key may order
    # equal-depth neighbours differently from the canonical order: the sums are the same terms in another order)
    print(assert_grads_close([t.grad.cpu().numpy() for t in params], ref, scale, "zero-change sequence"))
