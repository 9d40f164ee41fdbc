"""CPU tier: tools/step_timeline.py on a synthetic kernel trace -- the step boundaries, the idle gap and the per-queue busy times
are the figures DESIGN.md quotes from it, so the arithmetic is pinned here."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_step_timeline_reports_gaps_and_queues(tmp_path):
    rows = ["Kind,Agent_Id,Queue_Id,Kernel_Name,Start_Timestamp,End_Timestamp"]
    t = 1_000_000
    for step in range(6):  # six identical steps of 100 us: project 10, forward 30, [gap 4], loss 20 (+ side kernel 8 under it), backward 30, 6 idle
        base = t + step * 100_000
        rows += [f"KERNEL_DISPATCH,1,1,frame_project_count_kernel<false>(),{base},{base + 10_000}",
                 f"KERNEL_DISPATCH,1,1,raster_forward_kernel<3>(),{base + 10_000},{base + 40_000}",
                 f"KERNEL_DISPATCH,1,1,loss_fused_kernel(),{base + 44_000},{base + 64_000}",
                 f"KERNEL_DISPATCH,1,2,bucket_scan_kernel(),{base + 45_000},{base + 53_000}",
                 f"KERNEL_DISPATCH,1,1,raster_backward_rows_kernel(),{base + 64_000},{base + 94_000}"]
    path = tmp_path / "t_kernel_trace.csv"
    path.write_text("\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_timeline.py"), str(path)], capture_output=True,
                         text=True, check=True).stdout
    lines = out.strip().splitlines()
    assert len(lines) == 6  # five dispatches + the summary
    assert "frame_project_count_kernel" in lines[0] and "+     0.0 us" in lines[0]
    assert "loss_fused_kernel" in lines[2] and "[device idle 4.0 us before]" in lines[2]
    assert "queue   2" in lines[3] and "bucket_scan_kernel" in lines[3]
    assert "wall 100.0 us, some kernel running 90.0 us, idle 10.0 us" in lines[5]
    assert "1: 90.0 us, 2: 8.0 us" in lines[5]
