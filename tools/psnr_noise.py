#!/usr/bin/env python3
"""Is the held-out PSNR of the 7,001-iteration synthetic fit (tools/train_demo.py) a property of the build or of the run?

    python tools/psnr_noise.py [--seeds 1,2,3,4,5] [--trees .,build/r03]  >  gpurun_out/<tag>/psnr_noise.jsonl

Round 3 reported 32.9 ... 34.7 dB for the same run over seven builds (its final tree the lowest) and could not say whether
that was noise.  This runs the fit for several seeds (start perturbation + view order) on two checkouts of the repository
side by side on the same box -- the current tree and round 3's final tree (build/r03 = commit 3fe0e17, built with its own
gs_build.py) -- one fresh process per run; the last line is the table (mean, min, max per tree)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
a = sys.argv[1:]
seeds = [int(x) for x in (a[a.index("--seeds") + 1] if "--seeds" in a else "1,2,3,4,5").split(",")]
trees = (a[a.index("--trees") + 1] if "--trees" in a else ".,build/r03").split(",")
rows = []
for seed in seeds:
    for tree in trees:
        path = os.path.join(ROOT, tree)
        if not os.path.exists(os.path.join(path, "3d-gaussian-splatting_amd", "csrc", "libgs_amd.so")):
            print(json.dumps({"tree": tree, "error": "not built"}), flush=True)
            continue
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_demo.py"), "7001", "--seed", str(seed),
                            "--tree", path], capture_output=True, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode or not line:
            print(json.dumps({"tree": tree, "seed": seed, "error": p.stderr[-300:]}), flush=True)
            continue
        d = json.loads(line[-1])
        d["tree"] = tree
        rows.append(d)
        print(json.dumps({k: d[k] for k in ("tree", "seed", "iters_per_s", "test_psnr_before_dB", "test_psnr_after_dB",
                                            "test_ssim_after", "final_train_loss")}), flush=True)
table = {}
for tree in trees:
    ps = [r["test_psnr_after_dB"] for r in rows if r["tree"] == tree]
    if ps:
        table[tree] = {"runs": len(ps), "psnr_mean_dB": round(sum(ps) / len(ps), 2), "psnr_min_dB": min(ps),
                       "psnr_max_dB": max(ps)}
print(json.dumps({"table": table}))
