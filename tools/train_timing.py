"""Timing protocol of the training step, shared by bench.py and tools/sweep_n.py (round 4).

Round 3 timed the step two ways that disagreed by 15 % (693 it/s in bench.py, 798 it/s in sweep_n.py for the same
2.4 M-Gaussian scene): bench.py warmed up for 30 iterations against a noisy target and timed ONE block, sweep_n.py timed
iterations 5 .. 25 against a random target inside the learning-rate warm-up.  Every optimizer step moves the scene
(lower opacities -> tiles composite more Gaussians before they saturate -> more work), so the two harnesses timed
different scenes, and neither could repeat its block.

Here: `warm` iterations of the real step, then a SNAPSHOT of everything the step changes (parameters, both Adam moments,
the step counter, the densification statistic); every timed block restores the snapshot (outside the timed region) and
runs the same `k` iterations [warm, warm + k) again -- same scene, same work, R >= 15 blocks with min / median / max;
k = 25 or what ~30 ms of GPU work take, whichever is more (`block_length`).
Nothing inside the timed region differs from training: loss, backward, exchange, fused Adam all run in full.

One more thing the callers decide (bench.py `fixed_scene`, sweep_n.py): with the reference's learning rates the timed
iterations themselves move the scene -- and the longer the block, the further (100 k Gaussians: 3,300 it/s with
25-iteration blocks, 2,980 with 100-iteration blocks, 5,050 for iterations 5 .. 25 of round 3's sweep, one code:
profiles/r04_o_*, r04_n_*).  The headline training figures therefore run the step with learning rate 0: every kernel
does its full work, Adam included, and the parameters stay where the scene generator put them -- the scene the render
figures are quoted on; bench.py reports the moving-scene figure next to it.
"""
import statistics
import time

import torch


def snapshot(tr):
    o = tr.optimizer
    return {"param": tr.flat.flat_param.clone(), "m": o.exp_avg.clone(), "v": o.exp_avg_sq.clone(),
            "step": o.step_count, "stat": None if o.accum_grad is None else o.accum_grad.clone(),
            "lr": list(o._lr)}


def restore(tr, snap):
    o = tr.optimizer
    tr.flat.finish_gather()
    tr.renderer.forward_abandon()  # a frame projected ahead saw the parameters of the block's last step
    tr.flat.flat_param.copy_(snap["param"])
    o.exp_avg.copy_(snap["m"])
    o.exp_avg_sq.copy_(snap["v"])
    o.step_count = snap["step"]
    if snap["stat"] is not None:
        o.accum_grad.copy_(snap["stat"])
    for i, v in enumerate(snap["lr"]):
        o._lr[i] = v


def block_length(step_seconds, floor=25, ceiling=200, block_seconds=0.03):
    """Iterations per timed block: at least `floor`, and enough of them for ~30 ms of GPU work -- a block starts from an
    idle device (the snapshot is restored and the ranks meet in front of it), and a 5-ms block of a small scene spends a
    visible part of itself getting the pipeline and the clocks going: round 4 first timed 25 iterations everywhere and
    read 3,300 it/s at 100 k Gaussians where 20 free-running iterations give 5,050 (profiles/r04_n_*)."""
    return int(min(ceiling, max(floor, -(-block_seconds // max(step_seconds, 1e-6)))))


def time_training(tr, k=None, warm=30, repeats=15, barrier=None, max_over_ranks=None, camera_id=0, ahead=True):
    """(median seconds per block, [all blocks], k) -- see the module docstring.  `k`: iterations per block (None: chosen by
    `block_length` from the warm-up's own step time, the same on every rank).  `barrier` / `max_over_ranks`: the
    multi-rank hooks of bench.py (defaults: single process).  `ahead`: tell the step which view comes next (the same
    one), so that the view-parallel trainer can project the next frame behind its optimizer."""
    barrier = barrier or torch.cuda.synchronize
    max_over_ranks = max_over_ranks or (lambda x: x)
    nxt = camera_id if ahead else None
    for i in range(warm - 10):
        tr.train_step(i, camera_id, next_camera_id=nxt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warm - 10, warm):
        tr.train_step(i, camera_id, next_camera_id=nxt)
    torch.cuda.synchronize()
    if k is None:
        k = block_length(max_over_ranks((time.perf_counter() - t0) / 10))
    snap = snapshot(tr)
    blocks = []
    for _ in range(repeats):
        restore(tr, snap)
        barrier()
        t0 = time.perf_counter()
        for i in range(warm, warm + k):
            tr.train_step(i, camera_id, next_camera_id=nxt)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        barrier()
        blocks.append(max_over_ranks(dt))
    return statistics.median(blocks), blocks, k
