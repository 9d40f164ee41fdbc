#!/usr/bin/env python3
"""A/B timing of compile-time variants of libgs_amd.so in ONE gpurun call.

    python tools/ab_variants.py build nolegacy="-DGS_FWD_LEGACY_MUL=0" pf1="-DBIN_PF=1"     # here (CPU, hipcc)
    python tools/ab_variants.py run cfg5 cfg2 > gpurun_out/ab.txt                            # on the GPU box

`build` compiles every variant into build/variants/<name>/libgs_amd.so (git-ignored, but it travels with the gpurun
snapshot); `run` executes tools/stage_profile.py once per variant (and once for the in-tree product build, "base")
in a fresh process with GS_AMD_LIB pointing at the variant.  Experiments only: the product never sets GS_AMD_LIB.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VAR = os.path.join(ROOT, "build", "variants")
sys.path.insert(0, os.path.join(ROOT, "3d-gaussian-splatting_amd"))

if sys.argv[1] == "build":
    import gs_build

    for spec in sys.argv[2:]:
        name, flags = spec.split("=", 1)
        print(name, gs_build.build(outdir=os.path.join(VAR, name), defines=flags.split()), flush=True)
elif sys.argv[1] == "run":
    cfgs = sys.argv[2:] or ["cfg5"]
    names = ["base"] + (sorted(os.listdir(VAR)) if os.path.isdir(VAR) else [])
    for rep in range(2):  # two rounds: box-to-box and run-to-run noise is visible
        for name in names:
            env = dict(os.environ)
            if name != "base":
                env["GS_AMD_LIB"] = os.path.join(VAR, name, "libgs_amd.so")
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stage_profile.py"), *cfgs], env=env,
                               capture_output=True, text=True)
            for line in p.stdout.splitlines():
                print(f"[{name} #{rep}] {line}", flush=True)
            if p.returncode:
                print(f"[{name} #{rep}] FAILED: {p.stderr[-400:]}", flush=True)
else:
    raise SystemExit(__doc__)
