"""Build recipe for libgs_amd.so (hipcc, gfx950 only, in-tree).

``python gs_build.py`` or ``build()``: every ``csrc/*.hip`` is compiled to an object with
``hipcc --offload-arch=gfx950 -O3`` and linked into ``csrc/libgs_amd.so``.  Files whose results
feed the integer side of the pipeline (depth bits, tile rectangles) are compiled with
``-ffp-contract=off`` so they match the oracle bit for bit (see cull_project.hip).
No torch / pybind dependency: the library is a plain C ABI (include/gs_abi.h).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libgs_amd.so")

SOURCES = {
    # no SLP packing: the packer turns the 3x3 products into v_pk_mul/v_pk_add + ~140 v_mov shuffles per Gaussian;
    # a packed op has no rate advantage on gfx950, so that is pure issue time (and 66 instead of 45 VGPRs)
    "cull_project.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],
    "binning.hip": ["-ffp-contract=off"],
    "radix_sort.hip": [],
    "tile_sort.hip": [],
    "tile_bin.hip": [],
    "strip_bin.hip": [],
    # no SLP packing: v_pk_*_f32 has no rate advantage on gfx950 and costs extra v_mov shuffles
    "raster_fwd.hip": ["-fno-slp-vectorize"],
    "raster_bwd.hip": ["-fno-slp-vectorize"],
    "gs_frame.hip": [],
    "adam.hip": ["-ffp-contract=off"],  # same roundings as torch's unfused elementwise kernels
    "loss.hip": ["-fno-slp-vectorize"],  # as above: the packer costs ~50 v_mov per row step of the fused kernel
    "densify.hip": ["-ffp-contract=off"],
}
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall",
          "-Wno-unused-function", "-munsafe-fp-atomics",
          *os.environ.get("GS_EXTRA_HIPCC_FLAGS", "").split()]  # experiments only (-D switches)
HEADERS = ["gs_common.h", "gs_frame_layout.h", "raster_common.h", "strip_common.h", "tile_bin_common.h", os.path.join("..", "..", "include", "gs_abi.h")]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, outdir: str = None, defines=()) -> str:
    """Default: csrc/libgs_amd.so.  ``outdir`` + ``defines`` (-D switches) build an experiment VARIANT of the library
    somewhere else (tools/ab_variants.py; load it with GS_AMD_LIB=<path>): never the product build."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs, jobs = [], []
    odir = outdir or CSRC
    os.makedirs(odir, exist_ok=True)
    lib = os.path.join(odir, "libgs_amd.so")
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(odir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc, *COMMON, *extra, *defines, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
