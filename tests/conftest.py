"""pytest configuration: `gpu` marker, import paths, oracle build.

CPU tier  (`-m "not gpu"`): oracle vs known answers / golden vectors, host logic, ABI symbols.
GPU tier  (`-m gpu`)      : HIP kernels (through the C ABI) vs the oracle.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "3d-gaussian-splatting_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import oracle

    oracle.build()
    try:  # the suite's scenes are small: keep them on the strip variant of the binning (the product picks by scene size;
        # tests/test_gpu_frame.py::test_binning_variant_is_chosen_by_scene_size covers that choice)
        from gs_frame import FrameRenderer

        FrameRenderer.default_force_strips = True
    except ImportError:  # libgs_amd.so not built: the CPU tier that needs it reports that itself
        pass


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
