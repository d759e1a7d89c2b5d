import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def omx():
    """The product package (ctypes binding over libomx_hip.so)."""
    import omx_import
    return omx_import.load_package()


def needs_experiments(omx):
    """skip unless the loaded library is the `make EXPERIMENTS=1` build (OMX_LIB_VARIANT=exp): the measured-negative engines of
    EXPERIMENTS.md -- persistent decode step, AQL replay, 32x32 two-phase and stream-K attention -- are not in the default library"""
    import ctypes
    omx.lib.omx_experiments_built.restype = ctypes.c_int
    if not omx.lib.omx_experiments_built():
        pytest.skip("experimental engine: build with `make -C ominix-mlx_amd/csrc EXPERIMENTS=1`, run with OMX_LIB_VARIANT=exp")
