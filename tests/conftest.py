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
    config.addinivalue_line("markers", "perf: asserts a ratio of measured times (GPU tests; "
                                       "deselect with -m 'gpu and not perf')")


def pytest_sessionstart(session):
    """Builds the binaries when they are missing OR stale (hipcc cross-compiles
    without a GPU): `make` stamps a hash of the sources beside the libraries
    (tools/source_hash.py) and a tree whose hash differs is rebuilt, so the
    prebuilt files that travel to the GPU box with the snapshot are always the
    ones of this checkout."""
    import subprocess
    pkg = os.path.join(REPO, "accessor-blas_amd")
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from source_hash import source_hash
    stamp = os.path.join(pkg, "lib", "SOURCE_HASH")
    fresh = os.path.exists(stamp) and open(stamp).read().strip() == source_hash()
    wanted = [os.path.join(pkg, "lib", "libaccblas_hip.so"),
              os.path.join(pkg, "lib", "libaccblas_vendor.so"),
              os.path.join(pkg, "lib", "libaccblas_rccl.so"),
              os.path.join(pkg, "bin", "multi_gpu_benchmark"),
              os.path.join(pkg, "bin", "gemv_benchmark")]
    if not fresh or not all(os.path.exists(p) for p in wanted):
        subprocess.run(["make", "-C", pkg, "-j", str(min(8, os.cpu_count() or 1)), "all"],
                       check=True, stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(REPO, "oracle", "liboracle.so")):
        subprocess.run(["make", "-C", os.path.join(REPO, "oracle")], check=True,
                       stdout=subprocess.DEVNULL)


def pytest_sessionfinish(session, exitstatus):
    """The TRSV parity errors this session measured (tests/trsv_margins.py)."""
    import trsv_margins
    trsv_margins.dump()


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
