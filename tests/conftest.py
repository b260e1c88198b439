"""pytest configuration: the `gpu` marker and import paths.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol check (runs
without a GPU). `-m gpu`: parity tests proper, calling the HIP kernels through
the C ABI on a real MI355X.
"""
import os
import sys

import pytest

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for path in (os.path.join(REPO, "accessor-blas_amd", "python"),
             os.path.join(REPO, "tests")):
    if path not in sys.path:
        sys.path.insert(0, path)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
