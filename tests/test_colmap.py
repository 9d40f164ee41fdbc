"""CPU: gs_colmap readers / initial Gaussians against the reference's own readers (utils.py) and helpers, recorded
in tests/golden/colmap.npz from the synthetic binaries under tests/golden/colmap/ (make_golden.py::colmap)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "3d-gaussian-splatting_amd"))

HERE = os.path.dirname(__file__)
D = os.path.join(HERE, "golden", "colmap")
G = np.load(os.path.join(HERE, "golden", "colmap.npz"))


def test_cameras_images_points_match_reference_readers():
    import gs_colmap

    cams = gs_colmap.read_cameras_binary(os.path.join(D, "cameras.bin"))
    assert sorted(cams) == G["cam_ids"].tolist()
    for k, c in cams.items():
        assert c.model == str(G[f"cam{k}_model"]) and [c.width, c.height] == G[f"cam{k}_wh"].tolist()
        assert np.array_equal(c.params, G[f"cam{k}_params"])
    imgs = gs_colmap.read_images_binary(os.path.join(D, "images.bin"))
    assert sorted(imgs) == G["img_ids"].tolist()
    for k, im in imgs.items():
        assert np.array_equal(np.concatenate([im.qvec, im.tvec]), G[f"img{k}_pose"])
        assert im.camera_id == int(G[f"img{k}_cam"]) and im.name == str(G[f"img{k}_name"])
        assert np.array_equal(im.xys, G[f"img{k}_xys"]) and np.array_equal(im.point3D_ids, G[f"img{k}_pids"])
        assert np.allclose(im.qvec2rotmat(), G[f"img{k}_rot"], rtol=0, atol=1e-15)
    pts = gs_colmap.read_points3d_binary(os.path.join(D, "points3D.bin"))
    assert list(pts) == G["pt_ids"].tolist()  # file order, as the reference's dict
    assert np.array_equal(np.stack([p.xyz for p in pts.values()]), G["pt_xyz"])
    assert np.array_equal(np.stack([p.rgb for p in pts.values()]), G["pt_rgb"])
    assert np.array_equal(np.array([p.error for p in pts.values()]), G["pt_err"])
    assert np.array_equal(np.concatenate([p.image_ids for p in pts.values()]), G["pt_image_ids"])
    assert np.array_equal(np.concatenate([p.point2D_idxs for p in pts.values()]), G["pt_p2d"])


def test_initial_gaussians():
    import gs_colmap

    pts = gs_colmap.read_points3d_binary(os.path.join(D, "points3D.bin"))
    pos, quat, scale, opa, rgb = gs_colmap.initial_gaussians(pts, scale_init_value=1.0, opa_init_value=0.3)
    n = len(pts)
    assert pos.shape == (n, 3) and pos.dtype == np.float32 and np.array_equal(pos, G["pt_xyz"].astype(np.float32))
    assert np.allclose(rgb, G["init_rgb"], rtol=2e-7, atol=0)  # logit(rgb / 255), utils.inverse_sigmoid_torch
    assert np.array_equal(quat, np.tile([1, 0, 0, 0], (n, 1))) and np.allclose(opa, -np.log(1 / 0.3 - 1))
    # scale: mean distance to the three nearest neighbours (brute force check), isotropic
    d = np.linalg.norm(pos[:, None, :].astype(np.float64) - pos[None].astype(np.float64), axis=-1)
    want = np.sort(d, axis=1)[:, 1:4].mean(axis=1)
    assert np.allclose(scale, np.repeat(want[:, None], 3, 1), rtol=1e-5)
    *_, rgb_sh = gs_colmap.initial_gaussians(pts, use_sh_coeff=True)
    assert np.allclose(rgb_sh, G["init_sh"], rtol=2e-7, atol=0) and rgb_sh.shape == (n, 27)
    *_, s_exp, _, _ = gs_colmap.initial_gaussians(pts, scale_activation="exp")
    assert np.allclose(np.exp(s_exp), scale, rtol=1e-6)
