#!/usr/bin/env python3
"""Timeline of ONE training step out of a rocprofv3 kernel trace: which kernels ran when, on which queue, and where the device idled.

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python tools/prof_target.py cfg5 --train --frames 40
    python tools/step_timeline.py out/*/t_kernel_trace.csv [marker kernel substring, default raster_forward_kernel] [step from the end, default 3]

A step is the interval between two consecutive dispatches of the marker kernel.  Prints every dispatch of that interval (start
relative to the step's first dispatch, duration, queue) and the step's summary: wall time, time with at least one kernel running,
idle time, and the per-queue busy time -- kernels on a second queue are the renderer's side-stream work (the backward's work
lists underneath the caller's loss)."""
import csv
import sys

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "raster_forward_kernel"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = []
with open(path, newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
if len(marks) < back + 1:
    raise SystemExit(f"only {len(marks)} dispatches of {marker}")
# a step starts with the first kernel of its forward: walk back from the marker to the previous step's last kernel
lo_m, hi_m = marks[-back - 1], marks[-back]


def step_start(m):  # the project stage leads the forward: the nearest frame_project* dispatch in front of the marker
    i = m
    while i > 0 and "frame_project_count" not in rows[i][2] and "frame_project_kernel" not in rows[i][2]:
        i -= 1
    return i


a, b = step_start(lo_m), step_start(hi_m)
step = rows[a:b]
t0 = step[0][0]
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in sorted(step):
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = rows[b][0] - t0
per_q = {}
prev_end = t0
for s, e, n, q in step:
    short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:64]
    gap = s - prev_end
    print(f"  +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  queue {q:>3}  {short}" + (f"   [device idle {gap / 1e3:.1f} us before]" if gap > 1500 else ""))
    prev_end = max(prev_end, e)
    per_q[q] = per_q.get(q, 0) + e - s
print(f"step: wall {wall / 1e3:.1f} us, some kernel running {busy / 1e3:.1f} us, idle {(wall - busy) / 1e3:.1f} us; busy per queue: "
      + ", ".join(f"{q}: {v / 1e3:.1f} us" for q, v in sorted(per_q.items())))
