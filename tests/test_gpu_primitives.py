"""Runs the GPU self-test of the gfx950 wave primitives (SURVEY.md §8a rows A4 and
A7: reductions replacing `cuda/kernel_utils.cuh`, atomics replacing
`cuda/atomics.cuh` incl. the compare-and-swap fallback). Run with -m gpu."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
EXE = os.path.join(REPO, "accessor-blas_amd", "bin", "selftest_primitives")


def test_wave_primitives_selftest():
    assert os.path.exists(EXE), "build it with make -C accessor-blas_amd"
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "primitives ok" in out.stdout
