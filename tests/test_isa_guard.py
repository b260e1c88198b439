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
// fp32 arithmetic on fp32 storage (the reference's plain `trsv<float>` column)
using acc32 = gko::acc::reduced_row_major<2, float, float>;
using range32 = gko::acc::range<acc32>;
using c_range32 = gko::acc::range<acc32::const_accessor>;
hipError_t wide32(const c_range32 &m, const range32 &x, std::uint32_t *helper, const trsv_session &s)
{
    return launch_trsv_wide<float>(true, true, m, x, helper, s, nullptr, 256);
}
// 2-byte storage (general sweep, ring of three tiles)
using acc16 = gko::acc::reduced_row_major<2, double, _Float16>;
using range16 = gko::acc::range<acc16>;
using c_range16 = gko::acc::range<acc16::const_accessor>;
hipError_t wide16(const c_range16 &m, const range16 &x, std::uint32_t *helper, const trsv_session &s)
{
    return launch_trsv_wide<double>(false, true, m, x, helper, s, nullptr, 256);
}
hipError_t narrow(const c_range &m, const range &x, std::uint32_t *helper, const trsv_session &s)
{
    return launch_trsv<double>(false, true, m, x, helper, s, nullptr, 0);
}
// the relay variant (a loader wave feeds x through LDS), 8-byte storage,
// one workgroup per CU drawing tickets: what solves of 17000 ... 53000 fp64
// unknowns run
using acc64 = gko::acc::reduced_row_major<2, double, double>;
using range64 = gko::acc::range<acc64>;
using c_range64 = gko::acc::range<acc64::const_accessor>;
hipError_t relay(const c_range64 &m, const range64 &x, std::uint32_t *helper, const trsv_session &s)
{
    return launch_trsv<double>(false, true, m, x, helper, s, nullptr, 2, 256);
}
"""


def refills_without_a_full_wait(body):
    """Batches of non-temporal matrix loads (`global_load_dwordx4 ... nt`) after
    which the wave reaches its next FMAs WITHOUT an `s_waitcnt vmcnt(0)`: the
    refill of a ring slot stays in flight while the next panel is consumed."""
    count = 0
    pending = False  # a refill has been issued and no FMA has followed yet
    for line in (raw.strip() for raw in body.splitlines()):
        if line.startswith("global_load_dwordx4") and line.endswith(" nt"):
            pending = True
        elif line.startswith("s_waitcnt") and "vmcnt(0)" in line:
            pending = False
        elif line.startswith("v_fma") and pending:
            count += 1
            pending = False
    return count


def kernels(asm):
    """name -> body of every kernel in the assembly listing."""
    out = {}
    # (kernels only: device functions that are not inlined have labels of the
    # same form and end in s_setpc_b64)
    for name in re.findall(r"^\s*\.amdhsa_kernel (\S+)", asm, flags=re.M):
        match = re.search(r"^" + re.escape(name) + r":.*?\n(.*?)s_endpgm", asm, flags=re.S | re.M)
        out[name] = match.group(1)
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
    # the loop-free form (what the library launches by default) and the ticket
    # loop's form (option "trsv_wide_persistent") of every 128-row kernel
    looped = {k: v for k, v in found.items() if "9trsv_wide" in k and re.search(r"E[df]Lb1EN3gko", k)}
    wide = {k: v for k, v in found.items() if "9trsv_wide" in k and k not in looped}
    narrow = {k: v for k, v in found.items() if "kernel4trsv" in k}
    # (uplo x diag) x (fp64 / fp32 arithmetic on fp32 storage, fp64 arithmetic on fp16 storage)
    assert len(wide) == 12 and len(looped) == 12 and narrow, sorted(found)
    for name, body in wide.items():
        if "KDF16_" in name:
            continue  # (2-byte storage stages with global_load_ushort / short_d16)
        # between the ticket barrier and the "tiles staged" barrier: 3 tiles x
        # 8 rows per wave plus the block's own right-hand side, one batch
        # (the staging code sits between two barriers; which ones, in the order
        # of the listing, depends on where the compiler lays the cold paths)
        staging = max(longest_load_batch(part) for part in body.split("s_barrier"))
        assert staging >= 24, (name, staging)
    narrow_f32 = {k: v for k, v in narrow.items() if "EdKf" in k}  # 4-byte storage
    assert narrow_f32
    for name, body in narrow_f32.items():
        assert longest_load_batch(body) >= 16, (name, longest_load_batch(body))
    # The relay variant's workers keep their ring of matrix panels in flight
    # (round 4): their waits on the loader look at LDS with LDS instructions - a
    # FLAT load of an LDS flag counts in vmcnt and waits for every matrix load in
    # flight -, a slot is refilled into its own registers, and most refills are
    # followed by the next panel's FMAs without a full wait.
    relay = {k: v for k, v in found.items() if "kernel4trsv" in k and "Lb1EdLb" in k}
    deep = [k for k in relay if "ELi2ELb1EdLb1E" in k]  # 16-byte loads of fp64, persistent
    assert len(relay) >= 16 and len(deep) == 4, sorted(found)  # (uplo x diag of each)
    for name, body in relay.items():
        assert "flat_load" not in body and "flat_store" not in body, name
    for name in deep:
        assert refills_without_a_full_wait(relay[name]) >= 3, \
            (name, refills_without_a_full_wait(relay[name]))
    # register budget of the 128-row kernel, every instantiation the library
    # dispatches to by default: two waves per SIMD (at most 256 registers; the
    # fp64-arithmetic kernels of 4-byte storage sit at 253-255), nothing spilled,
    # no private segment
    for name in wide:
        meta = re.search(r"\.name:\s+" + re.escape(name) + r"\n(.*?)\.wavefront_size", asm, flags=re.S)
        block = asm[asm.index(".amdhsa_kernel " + name):]
        block = block[:block.index(".end_amdhsa_kernel")]
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", block).group(1))
        vgprs = int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", block).group(1))
        assert scratch == 0, (name, scratch)
        assert vgprs <= 256, (name, vgprs)
        assert meta is None or ".vgpr_spill_count: 0" in meta.group(1) or "vgpr_spill_count" not in meta.group(1)
    # The ticket loop calls the row block as an out-of-line function. A pointer
    # that reaches a device function as a plain `T *` is a flat pointer: every
    # matrix load a flat_load, which counts in lgkmcnt as well and makes each
    # wait for an LDS read a wait for the ring of tiles (round 5: 6-12 % slower).
    # The parameters are typed `global_ptr<T>`: no flat access anywhere.
    functions = re.findall(r"^(_Z\w*trsv_wide_ticket_call\w*):.*?\n(.*?)s_setpc_b64 s\[30:31\]", asm,
                           flags=re.S | re.M)
    assert len(functions) == 12, [name for name, _ in functions]
    for name, body in functions:
        assert "flat_load" not in body and "flat_store" not in body and "flat_atomic" not in body, name
