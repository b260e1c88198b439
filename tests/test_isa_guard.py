"""Guards two properties of the generated gfx950 code that no functional test
can see (hipcc cross-compiles without a GPU): the TRSV kernels stage their
diagonal tiles with ONE batch of loads (a full wait behind each load cost
5-7.5 us per block until round 2), and the 128-row kernel for
fp32 storage / fp64 arithmetic keeps everything in registers (no scratch)."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
PKG = os.path.join(REPO, "accessor-blas_amd")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

SOURCE = r"""
#include "trsv_wide_kernels.hpp"
using namespace accblas::dev;
using acc = gko::acc::reduced_row_major<2, double, float>;
using range = gko::acc::range<acc>;
using c_range = gko::acc::range<acc::const_accessor>;
hipError_t wide(const c_range &m, const range &x, std::uint32_t *helper, const trsv_session &s)
{
    return launch_trsv_wide<double>(false, true, m, x, helper, s, nullptr, 256);
}
hipError_t narrow(const c_range &m, const range &x, std::uint32_t *helper, const trsv_session &s)
{
    return launch_trsv<double>(false, true, m, x, helper, s, nullptr, 0);
}
"""


def kernels(asm):
    """name -> body of every kernel in the assembly listing."""
    out = {}
    for match in re.finditer(r"^(_Z\w+):.*?\n(.*?)s_endpgm", asm, flags=re.S | re.M):
        out[match.group(1)] = match.group(2)
    return out


def longest_load_batch(body):
    """Most `global_load_dword` (one element per lane) instructions issued in a
    row with no `s_waitcnt vmcnt` between them."""
    best = run = 0
    for line in body.splitlines():
        line = line.strip()
        if line.startswith("global_load_dword ") or line.startswith("global_load_dword\t"):
            run += 1
            best = max(best, run)
        elif line.startswith("s_waitcnt") and "vmcnt" in line:
            run = 0
    return best


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_trsv_tile_staging_is_one_batch_of_loads_and_nothing_spills(tmp_path):
    src = tmp_path / "isa_guard.hip"
    src.write_text(SOURCE)
    asm_path = tmp_path / "isa_guard.s"
    subprocess.run([HIPCC, "-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=fast",
                    "-I", os.path.join(PKG, "include"), "-I", os.path.join(PKG, "csrc"),
                    "--offload-device-only", "-S", str(src), "-o", str(asm_path)],
                   check=True, capture_output=True, timeout=600)
    asm = asm_path.read_text()
    found = kernels(asm)
    wide = {k: v for k, v in found.items() if "9trsv_wide" in k}
    narrow = {k: v for k, v in found.items() if "kernel4trsv" in k}
    assert len(wide) == 4 and narrow, sorted(found)
    for name, body in wide.items():
        # between the ticket barrier and the "tiles staged" barrier: 3 tiles x
        # 8 rows per wave plus the block's own right-hand side, one batch
        staging = body.split("s_barrier")[1]
        assert longest_load_batch(staging) >= 24, (name, longest_load_batch(staging))
    for name, body in narrow.items():
        assert longest_load_batch(body) >= 16, (name, longest_load_batch(body))
    # register budget of the 128-row kernel: two waves per SIMD, no scratch
    for name in wide:
        meta = re.search(r"\.name:\s+" + re.escape(name) + r"\n(.*?)\.wavefront_size", asm, flags=re.S)
        block = asm[asm.index(".amdhsa_kernel " + name):]
        block = block[:block.index(".end_amdhsa_kernel")]
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", block).group(1))
        assert scratch == 0, (name, scratch)
        assert meta is None or ".vgpr_spill_count: 0" in meta.group(1) or "vgpr_spill_count" not in meta.group(1)
