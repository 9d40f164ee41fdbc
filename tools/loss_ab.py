#!/usr/bin/env python3
"""A/B timing of compile-time variants of the image-loss kernels (loss.hip) in one gpurun call: every library under
build/variants/ (tools/ab_variants.py build ...) and the in-tree build run tools/loss_profile.py in a fresh process."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VAR = os.path.join(ROOT, "build", "variants")
names = ["base"] + (sorted(os.listdir(VAR)) if os.path.isdir(VAR) else [])
for rep in range(2):
    for name in names:
        env = dict(os.environ)
        if name != "base":
            env["GS_AMD_LIB"] = os.path.join(VAR, name, "libgs_amd.so")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "loss_profile.py")], env=env, capture_output=True,
                           text=True)
        print(f"[{name} #{rep}] {p.stdout.strip()}" + (f" FAILED: {p.stderr[-300:]}" if p.returncode else ""), flush=True)
