"""Compile the REFERENCE's own CUDA kernels for the CPU (oracle/_ref/libgs_ref.so).

TEST INFRASTRUCTURE ONLY -- runs where /root/reference exists (the build container); the GPU
box only ever sees the resulting .so (git-ignored, not gpurun-ignored) and the golden fixtures
generated from it (tests/golden/, tests/golden/make_golden.py).

The reference cannot be built with its own toolchain here (no nvcc; hipified source fails on
ROCm 7.2) and its host launchers need libtorch.  But its DEVICE code is
plain C++ once a handful of CUDA keywords and intrinsics exist, so this recipe

  1. reads /root/reference/src/gaussian.cu where it lies,
  2. keeps only the top-level items that are device code (`__global__` / `__device__`
     functions, `__constant__` tables, `#define FULL_MASK`) -- the host launchers (anything
     mentioning torch::Tensor or a <<<...>>> launch) are dropped,
  3. pipes   #include "cuda_cpu_shim.h"  +  that device code  +  #include "ref_harness.inc"
     to g++ on STDIN (-x c++ -), so no reference source text is ever written into the repo,
  4. writes only oracle/_ref/libgs_ref.so.

oracle/cuda_cpu_shim.h is a deterministic SIMT emulator (fibers + barrier/shuffle/activemask
scheduling); oracle/ref_harness.inc holds the launch shapes of the reference host code.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src/gaussian.cu"
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "libgs_ref.so")


def _strip_line_comment(line: str) -> str:
    i = line.find("//")
    return line if i < 0 else line[:i]


def device_code(text: str) -> str:
    """Top-level items of a .cu file that are device code."""
    lines = text.split("\n")
    items, cur, depth = [], [], 0
    for raw in lines:
        code = _strip_line_comment(raw)
        if depth == 0 and not cur:
            s = code.strip()
            if not s:
                continue
            if s.startswith("#"):
                items.append(raw)  # preprocessor line, classified below
                continue
        cur.append(raw)
        depth += code.count("{") - code.count("}")
        ended = depth == 0 and (code.rstrip().endswith("}") or code.rstrip().endswith(";"))
        if ended:
            items.append("\n".join(cur))
            cur = []
    keep = []
    for it in items:
        s = it.strip()
        if s.startswith("#"):
            if s.startswith("#define FULL_MASK"):
                keep.append(it)
            continue  # drops #include lines (torch, cuda_runtime, common.hpp) and other macros
        code = "\n".join(_strip_line_comment(l) for l in it.split("\n"))
        if "torch::Tensor" in code or "<<<" in code or "Gaussian3ds" in code:
            continue  # host launchers
        if "__global__" in code or "__device__" in code:
            keep.append(it)
    return "\n".join(keep) + "\n"


def build(force: bool = False) -> str | None:
    if not os.path.exists(REF_SRC):
        return OUT if os.path.exists(OUT) else None
    deps = [REF_SRC, os.path.join(HERE, "cuda_cpu_shim.h"), os.path.join(HERE, "ref_harness.inc"), __file__]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(REF_SRC) as f:
        dev = device_code(f.read())
    unit = '#include "cuda_cpu_shim.h"\n' + dev + '\n#include "ref_harness.inc"\n'
    cmd = ["g++", "-x", "c++", "-", "-std=c++17", "-O1", "-ffp-contract=off", "-fPIC", "-shared", "-w",
           "-I", HERE, "-o", OUT]
    subprocess.run(cmd, input=unit.encode(), check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
