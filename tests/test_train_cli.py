"""CPU tier: the command line of the trainer and the dataset layer under it.

* every flag of the reference's argparse (train.py:296-363) exists here with the same default and choices --
  compared with tests/golden/train_cli.json, extracted from the reference's source by tests/golden/make_golden.py;
* COLMAP writers -> readers round trip, and ``load_scene`` on a small capture written on the fly (poses, intrinsics
  divided by the downsample factor, sizes taken from the loaded images, fp16-rounded targets, missing files skipped).
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

import gs_colmap as gc

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "3d-gaussian-splatting_amd"))


def test_cli_flags_and_defaults_match_the_reference():
    import train as cli

    gold = json.load(open(os.path.join(HERE, "golden", "train_cli.json")))
    ours = {a.dest: a for a in cli.build_parser()._actions if a.dest != "help"}
    assert sorted(ours) == sorted(gold), (sorted(set(gold) - set(ours)), sorted(set(ours) - set(gold)))
    for name, g in gold.items():
        a = ours[name]
        assert a.type.__name__ == g["type"], name
        assert a.default == g["default"] and type(a.default) is type(g["default"]), (name, a.default, g["default"])
        assert (list(a.choices) if a.choices else None) == g["choices"], name


def test_cli_options_reach_the_trainer():
    import train as cli

    opt = cli.build_parser().parse_args(["--lr", "0.002", "--grad_accum_method", "mean", "--use_clone", "1",
                                         "--scale_reg", "0.01", "--n_iters", "3000"])
    o = cli.train_options(opt)
    assert (o.lr, o.grad_accum_method, o.use_clone, o.scale_reg, o.n_iters) == (0.002, "mean", 1, 0.01, 3000)
    assert o.n_iters_warmup == 300 and o.ssim_weight == 0.1 and o.adaptive_control_start_iter == 600


def _tiny_capture(root, with_missing=True):
    from PIL import Image

    rng = np.random.default_rng(3)
    os.makedirs(os.path.join(root, "sparse", "0"))
    cams = {1: gc.ColmapCamera(1, "PINHOLE", 64, 48, np.array([70.0, 72.0, 32.0, 24.0])),
            7: gc.ColmapCamera(7, "SIMPLE_PINHOLE", 80, 40, np.array([90.0, 40.0, 20.0]))}
    imgs, truth = {}, {}
    os.makedirs(os.path.join(root, "images_2"))
    for k, image_id in enumerate((5, 3, 9)):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        cam = cams[1 if k != 1 else 7]
        imgs[image_id] = gc.ColmapImage(image_id, q, rng.normal(size=3), cam.id, f"im_{image_id}.png",
                                        rng.uniform(0, 60, (4, 2)), np.array([1, -1, 2, 3], np.int64))
        if with_missing and image_id == 9:
            continue  # listed in images.bin, absent on disk: skipped like splatter.py:440-441
        arr = rng.integers(0, 256, (cam.height // 2, cam.width // 2, 3), dtype=np.uint8)
        Image.fromarray(arr, "RGB").save(os.path.join(root, "images_2", f"im_{image_id}.png"))
        truth[image_id] = arr
    pts = {i: gc.ColmapPoint3D(i, rng.normal(size=3), rng.integers(1, 255, 3).astype(np.uint8), 0.1 * i,
                               np.array([3, 5], np.int32), np.array([0, 1], np.int32)) for i in (11, 12, 13, 14)}
    gc.write_cameras_binary(os.path.join(root, "sparse", "0", "cameras.bin"), cams)
    gc.write_images_binary(os.path.join(root, "sparse", "0", "images.bin"), imgs)
    gc.write_points3d_binary(os.path.join(root, "sparse", "0", "points3D.bin"), pts)
    return cams, imgs, pts, truth


def test_colmap_writers_round_trip(tmp_path):
    cams, imgs, pts, _ = _tiny_capture(str(tmp_path))
    d = tmp_path / "sparse" / "0"
    rc, ri, rp = gc.read_cameras_binary(d / "cameras.bin"), gc.read_images_binary(d / "images.bin"), \
        gc.read_points3d_binary(d / "points3D.bin")
    assert sorted(rc) == sorted(cams) and all(rc[k].model == cams[k].model and np.array_equal(rc[k].params, cams[k].params)
                                              and (rc[k].width, rc[k].height) == (cams[k].width, cams[k].height) for k in cams)
    for k, im in imgs.items():
        assert np.array_equal(ri[k].qvec, im.qvec) and np.array_equal(ri[k].tvec, im.tvec) and ri[k].name == im.name
        assert ri[k].camera_id == im.camera_id and np.array_equal(ri[k].xys, im.xys)
        assert np.array_equal(ri[k].point3D_ids, im.point3D_ids)
    for k, p in pts.items():
        assert np.array_equal(rp[k].xyz, p.xyz) and np.array_equal(rp[k].rgb, p.rgb) and rp[k].error == p.error
        assert np.array_equal(rp[k].image_ids, p.image_ids) and np.array_equal(rp[k].point2D_idxs, p.point2D_idxs)
    for q in (np.array([1.0, 0, 0, 0]), imgs[5].qvec, imgs[3].qvec):
        back = gc.rotmat2qvec(gc.qvec2rotmat(q))
        assert np.allclose(back, q if q[0] >= 0 else -q, atol=1e-12)


def test_load_scene_follows_the_reference_conventions(tmp_path):
    cams, imgs, pts, truth = _tiny_capture(str(tmp_path))
    sc = gc.load_scene(str(tmp_path), render_downsample=2)
    assert sc.names == ["im_3.png", "im_5.png"]  # sorted by image id, the missing file skipped
    assert sorted(sc.points3d) == sorted(pts)
    for cam, tgt, image_id in zip(sc.cameras, sc.targets, (3, 5)):
        c = cams[imgs[image_id].camera_id]
        assert (cam.width, cam.height) == (c.width // 2, c.height // 2) == (tgt.shape[1], tgt.shape[0])
        assert cam.focal_x == c.params[0] / 2 and cam.focal_y == c.params[1] / 2  # params[0:2], whatever the model
        assert np.allclose(cam.rot, gc.qvec2rotmat(imgs[image_id].qvec), atol=1e-6)
        assert np.allclose(cam.tran, imgs[image_id].tvec, atol=1e-6)
        want = (torch.from_numpy(truth[image_id]).to(torch.float16) / 255.0).to(torch.float32)
        assert tgt.dtype == torch.float32 and torch.equal(tgt, want)
    with pytest.raises(RuntimeError):
        gc.load_scene(str(tmp_path), render_downsample=8)  # no images_8 folder
