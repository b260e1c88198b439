"""the HIP kernels driven through the OTHER accessor kinds
(`row_major`, `scaled_reduced_row_major`). There is no C-ABI entry for them (the
reference's launchers only build `reduced_row_major` ranges), so this compiles a
small HIP program against the kernel headers and runs it on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
PKG = os.path.join(REPO, "accessor-blas_amd")


def test_kernels_accept_other_accessor_kinds(tmp_path):
    exe = tmp_path / "device_accessor_kinds"
    subprocess.run(
        ["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "--offload-arch=gfx950",
         "-ffp-contract=fast", "-I", os.path.join(PKG, "include"), "-I",
         os.path.join(PKG, "csrc"), os.path.join(REPO, "tests", "cpp",
                                                 "device_accessor_kinds.hip"),
         "-o", str(exe)], check=True, timeout=900)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "device accessor kinds ok" in out.stdout
    assert out.stdout.count(" ok\n") >= 6
