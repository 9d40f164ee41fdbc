"""GPU tier: the reference's command line end to end on a synthetic capture (there is no real dataset offline):
COLMAP binaries + images on disk -> train.py main() -> checkpoint, images, test-split metrics."""
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
    msc.build(root, n=6000, width=192, height=128, views=17, points=2500, downsample=(1, 2), seed=5)
    return root


def test_train_cli_end_to_end(capture, tmp_path, capsys):
    import train as cli

    exp = str(tmp_path / "exp")
    common = ["--data", capture, "--exp", exp, "--render_downsample_start", "2", "--render_downsample", "1",
              "--n_iters_warmup", "20", "--n_history_track", "50", "--seed", "3"]
    # the initial point cloud alone (an evaluation-only run before any training)
    out = cli.main(common + ["--n_iters", "1", "--n_iters_test", "1000000", "--n_save_train_img", "1000000"])
    res = cli.main(common + ["--n_iters", "701", "--n_iters_test", "350", "--n_save_train_img", "350",
                             "--n_adaptive_control", "100", "--grad_accum_iters", "20"])
    log = capsys.readouterr().out
    assert "TEST SPLIT PSNR" in log and "REDNDERING SPEED" in log
    # artefacts of train.py:222-228, 236-254, 283-291
    assert os.path.exists(os.path.join(exp, "ckpt.pth"))
    assert {"train_0.png", "train_350.png", "train_700.png"} <= set(os.listdir(os.path.join(exp, "imgs")))
    tests = os.listdir(os.path.join(exp, "test_imgs"))
    assert {f"iter_700_cid_{c}.png" for c in (0, 8, 16)} <= set(tests)
    ck = torch.load(os.path.join(exp, "ckpt.pth"))
    assert sorted(ck) == ["opa", "pos", "quat", "rgb", "scale"] and ck["pos"].shape[1] == 3
    assert all(torch.isfinite(v).all() for v in ck.values())
    # the resolution switch at iteration 400 happened: the last test renders are full size
    from PIL import Image

    assert Image.open(os.path.join(exp, "test_imgs", "iter_700_cid_8.png")).size == (192, 128)
    assert Image.open(os.path.join(exp, "test_imgs", "iter_350_cid_8.png")).size == (96, 64)
    # training from the sparse cloud improves the held-out views
    assert res["test"]["psnr"] > 17.0 and np.isfinite(res["loss"]) and res["n_gaussians"] > 0
    # --test 1 --ckpt: evaluation of a saved model gives the numbers of the last evaluation
    ev = cli.main(["--data", capture, "--exp", exp, "--render_downsample_start", "1", "--test", "1", "--ckpt",
                   os.path.join(exp, "ckpt.pth")])
    assert abs(ev["test"]["psnr"] - res["test"]["psnr"]) < 1e-3
    assert out is not None


def test_train_cli_under_a_torchrun_environment(capture, tmp_path, monkeypatch):
    """RANK / WORLD_SIZE / MASTER_* set as torchrun does (one rank: there is one GPU here): the script goes through
    torch.distributed with the RCCL backend, draws its views from the shared generator and still writes rank 0's
    artefacts."""
    import torch.distributed as dist

    import train as cli

    for k, v in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"),
                 ("MASTER_PORT", str(29600 + os.getpid() % 300))):
        monkeypatch.setenv(k, v)
    exp = str(tmp_path / "exp_dp")
    res = cli.main(["--data", capture, "--exp", exp, "--render_downsample_start", "2", "--render_downsample", "2",
                    "--n_iters", "61", "--n_iters_warmup", "10", "--n_iters_test", "60", "--n_save_train_img", "60",
                    "--n_history_track", "30"])
    assert not dist.is_initialized()  # torn down at the end
    assert os.path.exists(os.path.join(exp, "ckpt.pth")) and np.isfinite(res["loss"]) and "test" in res


def test_train_cli_rejects_what_it_does_not_provide(capture, tmp_path):
    import train as cli

    for bad in (["--gui", "1", "--test", "1"],):
        with pytest.raises(SystemExit):
            cli.main(["--data", capture, "--exp", str(tmp_path / "x")] + bad)
