"""Command line of the reference trainer (train.py:294-420) on the MI355X-native path.

    python train.py --data colmap_garden/ --exp garden --render_downsample 4 ...

Same flags and defaults as the reference's argparse (tests/test_train_cli.py compares them with the reference's
source), same dataset layout (``<data>/sparse/0/*.bin`` + ``<data>/images_<k>/``), same schedule (LR lambdas,
adaptive control, opacity reset, every-8th-view test split, resolution switch at iteration 400), same artefacts
(``<exp>/ckpt.pth`` in the reference's dict format, ``<exp>/imgs/train_<i>.png``, ``<exp>/test_imgs/...``).
What differs: one fused HIP frame call per direction instead of ~25 torch kernels and >= 8 host synchronisations
(gs_frame.FrameRenderer), fused loss / Adam / densification kernels (gs_train.Trainer), and logging that reads
the device every ``--n_history_track`` iterations instead of three ``.item()`` calls per iteration.

New: under ``torchrun --nproc-per-node N`` the same script trains view-parallel (one view per GPU per step, RCCL
all-reduce of the gradients, rank-consistent densification).

Not provided: ``--gui`` (the viser viewer; its per-frame hook ``Trainer.test(None, extrinsics, intrinsics)`` is),
``--jacobian_track`` / ``--adaptive_lr`` / ``--debug`` (accepted, ignored: they select debugging code paths of the
reference).  Deliberate deviation: with ``--render_downsample_start != --render_downsample`` the focal lengths of
the start resolution are those of ``images_<start>`` and a real switch happens at iteration 400; the reference divides
the focal lengths by ``render_downsample`` from step 0 (its switch at 400 is then a no-op on the intrinsics).
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def build_parser() -> argparse.ArgumentParser:
    """train.py:296-363, flag for flag."""
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    a = p.add_argument
    a("--n_iters", type=int, default=7001)
    a("--n_iters_warmup", type=int, default=300)
    a("--n_iters_test", type=int, default=200)
    a("--n_history_track", type=int, default=100)
    a("--n_save_train_img", type=int, default=100)
    a("--n_adaptive_control", type=int, default=100)
    a("--render_downsample_start", type=int, default=4)
    a("--render_downsample", type=int, default=4)
    a("--jacobian_track", type=int, default=0)
    a("--data", type=str, default="colmap_garden/")
    a("--scale_init_value", type=float, default=1)
    a("--opa_init_value", type=float, default=0.3)
    a("--tile_culling_dist_thresh", type=float, default=0.5)
    a("--tile_culling_prob_thresh", type=float, default=0.05)
    a("--tile_culling_method", type=str, default="prob2", choices=["dist", "prob", "prob2"])
    # learning rate
    a("--lr", type=float, default=0.003)
    a("--lr_factor_for_scale", type=float, default=1)
    a("--lr_factor_for_rgb", type=float, default=10)
    a("--lr_factor_for_opa", type=float, default=10)
    a("--lr_factor_for_quat", type=float, default=1)
    a("--lr_decay", type=str, default="exp", choices=["none", "official", "exp"])
    a("--delete_thresh", type=float, default=1.5)
    a("--n_opa_reset", type=int, default=10000000)
    a("--reset_interval", type=int, default=500)
    a("--split_thresh", type=float, default=0.05)
    a("--ssim_weight", type=float, default=0.1)
    a("--debug", type=int, default=0)
    a("--use_sh_coeff", type=int, default=0)
    a("--scale_reg", type=float, default=0)
    a("--opa_reg", type=float, default=0)
    a("--cudaculling", type=int, default=1)
    a("--adaptive_lr", type=int, default=0)
    a("--seed", type=int, default=2023)
    a("--ckpt", type=str, default="")
    a("--scale_activation", type=str, default="abs", choices=["abs", "exp"])
    a("--fast_drawing", type=int, default=1)
    a("--exp", type=str, default="default")
    # adaptive control
    a("--grad_accum_iters", type=int, default=50)
    a("--grad_accum_method", type=str, default="max", choices=["mean", "max"])
    a("--grad_thresh", type=float, default=0.0002)
    a("--use_clone", type=int, default=0)
    a("--use_split", type=int, default=1)
    a("--clone_dt", type=float, default=0.01)
    a("--grad_aggregation", type=str, default="max", choices=["max", "mean"])
    a("--adaptive_control_end_iter", type=int, default=1000000000)
    # GUI related (parsed for compatibility; the viewer itself is not part of this package)
    a("--gui", default=0, type=int)
    a("--test", default=0, type=int)
    a("--H", default=768, type=int)
    a("--W", default=1024, type=int)
    a("--radius", default=5.0, type=float)
    a("--fovy", type=float, default=50)
    a("--max_spp", type=int, default=1)
    a("--dt_gamma", type=float, default=0)
    a("--max_steps", type=int, default=1024)
    a("--bound", type=float, default=10)
    return p


def train_options(opt):
    from gs_train import TrainOptions

    return TrainOptions(
        lr=opt.lr, lr_factor_for_scale=opt.lr_factor_for_scale, lr_factor_for_rgb=opt.lr_factor_for_rgb,
        lr_factor_for_opa=opt.lr_factor_for_opa, lr_factor_for_quat=opt.lr_factor_for_quat, lr_decay=opt.lr_decay,
        n_iters=opt.n_iters, n_iters_warmup=opt.n_iters_warmup, ssim_weight=opt.ssim_weight,
        scale_reg=opt.scale_reg, opa_reg=opt.opa_reg, grad_accum_method=opt.grad_accum_method,
        n_adaptive_control=opt.n_adaptive_control, adaptive_control_end_iter=opt.adaptive_control_end_iter,
        grad_accum_iters=opt.grad_accum_iters, n_opa_reset=opt.n_opa_reset, reset_interval=opt.reset_interval,
        split_thresh=opt.split_thresh, delete_thresh=opt.delete_thresh, grad_thresh=opt.grad_thresh,
        grad_aggregation=opt.grad_aggregation, use_clone=opt.use_clone, use_split=opt.use_split,
        clone_dt=opt.clone_dt)


def save_png(path: str, image):
    """cv2.imwrite(path, (img.clip(0, 1) * 255).astype(uint8)[..., ::-1]) of train.py:227, with Pillow."""
    from PIL import Image

    arr = (image.detach().clamp(0, 1).cpu().numpy() * 255).astype(np.uint8)
    Image.fromarray(arr, "RGB").save(path)


def evaluate(trainer, test_split, out_dir=None, tag=""):
    """The test-split block of train.py:236-254: PSNR, SSIM and rendering speed over every 8th view."""
    psnrs, ssims, elapsed = [], [], 0.0
    for cid in test_split:
        out = trainer.test(int(cid))
        elapsed += out["render_time"]
        psnrs.append(out["psnr"])
        ssims.append(out.get("ssim", float("nan")))
        if out_dir:
            os.makedirs(out_dir, exist_ok=True)
            save_png(os.path.join(out_dir, f"{tag}_cid_{cid}.png"), out["image"])
    res = {"psnr": float(np.mean(psnrs)), "ssim": float(np.mean(ssims)), "fps": len(test_split) / max(elapsed, 1e-9)}
    print("TEST SPLIT PSNR: {:.4f}".format(res["psnr"]))
    print("TEST SPLIT SSIM: {:.4f}".format(res["ssim"]))
    print("REDNDERING SPEED: {:.4f}".format(res["fps"]))  # (sic) the reference's wording
    return res


def main(argv=None) -> dict:
    opt = build_parser().parse_args(argv)
    import torch

    import gs_colmap
    from gs_train import Trainer

    if opt.gui:
        raise SystemExit("--gui: the viser viewer is not part of this package; drive Trainer.test(None, extrinsics, "
                         "intrinsics) from your viewer instead (see INTEGRATION.md)")
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs a HIP device (there is no CPU fallback)")
    np.random.seed(opt.seed)  # train.py:365: the view order is drawn from numpy's global generator
    torch.manual_seed(opt.seed)
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    # View-parallel training (new; the reference is single-GPU): under torchrun every rank renders its own view of a
    # step's batch, the parameter gradients are averaged with one RCCL all-reduce (gs_dp.py) and the densification
    # decisions are kept rank-consistent (gs_dp.ViewParallelGradStat).  Rank 0 logs, evaluates and writes files.
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    scene = gs_colmap.load_scene(opt.data, opt.render_downsample_start, dev)
    if opt.ckpt:
        ck = torch.load(opt.ckpt, map_location=dev)
        params = [ck[k].detach().to(torch.float32).contiguous() for k in ("pos", "quat", "scale", "opa", "rgb")]
    else:
        init = gs_colmap.initial_gaussians(scene.points3d, opt.scale_init_value, opt.opa_init_value,
                                           opt.scale_activation, bool(opt.use_sh_coeff))
        params = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in init]
    trainer = Trainer(params, scene.cameras, scene.targets, train_options(opt), world_size=world,
                      scale_activation=opt.scale_activation, densify=True,
                      generator=torch.Generator(dev).manual_seed(opt.seed))
    trainer.renderer.thresh = float(opt.tile_culling_prob_thresh)
    from gs_frame import TILE_CULLING

    trainer.renderer.tile_culling_method = TILE_CULLING[opt.tile_culling_method]
    trainer.renderer.tile_culling_dist_thresh = float(opt.tile_culling_dist_thresh)
    n_cameras = len(scene.cameras)
    test_split = np.arange(0, n_cameras, 8)  # train.py:69-71
    train_split = np.array(sorted(set(range(n_cameras)) - set(test_split.tolist())))
    os.makedirs(opt.exp, exist_ok=True)
    if opt.test:
        out = {"test": evaluate(trainer, test_split, os.path.join(opt.exp, "test_imgs"), "test")} if rank == 0 else {}
        if world > 1 or "RANK" in os.environ:  # every rank leaves together, the process group is torn down
            import torch.distributed as dist

            dist.barrier()
            dist.destroy_process_group()
        return out
    if len(train_split) == 0:
        raise SystemExit("the dataset has a single image: nothing left to train on after the every-8th test split")

    hist, summary, t_start = [], {}, time.perf_counter()
    # train.py:94; with several ranks the same draw of `world` views happens on every rank (same generator state
    # everywhere) and rank r takes the r-th.  The draw of iteration i + 1 is made before step i runs (nothing else consumes
    # this generator, so the sequence of views is the reference's): the view-parallel step can then issue the next frame's
    # project stage behind its optimizer (gs_train.Trainer.train_step, next_camera_id)
    next_id = int(np.random.choice(train_split, world)[rank])
    for i_iter in range(opt.n_iters):
        camera_id = next_id
        next_id = int(np.random.choice(train_split, world)[rank]) if i_iter + 1 < opt.n_iters else None
        hist.append(trainer.train_step(i_iter, camera_id, next_camera_id=next_id if world > 1 else None).clone())
        if rank != 0:
            if i_iter == 400 and opt.render_downsample != opt.render_downsample_start:
                scene = gs_colmap.load_scene(opt.data, opt.render_downsample, dev)
                trainer.cameras, trainer.targets = scene.cameras, scene.targets
            del hist[:-1]
            continue
        if i_iter % opt.n_save_train_img == 0:  # train.py:222-228
            os.makedirs(os.path.join(opt.exp, "imgs"), exist_ok=True)
            save_png(os.path.join(opt.exp, "imgs", f"train_{i_iter}.png"),
                     trainer.renderer.forward(*trainer.flat.params, trainer.cameras[camera_id], training=False)[0])
            trainer.save_checkpoint(os.path.join(opt.exp, "ckpt.pth"))
        if i_iter % opt.n_history_track == 0 or i_iter == opt.n_iters - 1:
            v = torch.stack(hist[-opt.n_history_track:]).mean(0).cpu().numpy()  # the one host read of this window
            del hist[:-opt.n_history_track]
            rate = (i_iter + 1) / (time.perf_counter() - t_start)
            print(f"iter {i_iter}: loss {v[0]:.6f} l1 {v[1]:.6f} ssim {v[2]:.4f} [{trainer.n_gaussians} Gaussians] "
                  f"{rate:.1f} it/s", flush=True)
            summary.update(loss=float(v[0]), l1=float(v[1]), iters_per_s=rate, n_gaussians=trainer.n_gaussians)
        if i_iter == 400 and opt.render_downsample != opt.render_downsample_start:  # train.py:233-234
            scene = gs_colmap.load_scene(opt.data, opt.render_downsample, dev)
            trainer.cameras, trainer.targets = scene.cameras, scene.targets
        if i_iter % opt.n_iters_test == 0:
            summary["test"] = evaluate(trainer, test_split, os.path.join(opt.exp, "test_imgs"), f"iter_{i_iter}")
    if rank == 0:
        trainer.save_checkpoint(os.path.join(opt.exp, "ckpt.pth"))
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    return summary


if __name__ == "__main__":
    main()
