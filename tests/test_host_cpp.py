"""Builds and runs the C++14 host unit tests (tests/cpp/host_tests.cpp) with g++:
accessor contract (the static_assert identities of the reference kernels, proxy
semantics), float16 conversions, matrix_info / reduce / compare / Memory /
gen_mtx. Also checks the benchmark CLIs' flag handling and the CPU-only config
C1 (`dot_benchmark --cpu-baseline --size=1000000`). No GPU needed."""
import os
import subprocess

import pytest

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
PKG = os.path.join(REPO, "accessor-blas_amd")
BIN = os.path.join(PKG, "bin")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.run(["make", "-C", PKG, "-j", "4", "all"], check=True,
                   stdout=subprocess.DEVNULL)


def test_host_unit_tests(tmp_path):
    exe = tmp_path / "host_tests"
    subprocess.run(
        ["g++", "-std=c++14", "-O1", "-Wall", "-Wextra", "-Werror",
         os.path.join(REPO, "tests", "cpp", "host_tests.cpp"), "-o", str(exe),
         "-I", os.path.join(PKG, "include"), "-I", os.path.join(REPO, "include"),
         "-I", os.path.join(PKG, "hip"), "-L", os.path.join(PKG, "lib"), "-laccblas_hip",
         "-Wl,-rpath," + os.path.join(PKG, "lib")], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host tests ok" in out.stdout


def test_host_unit_tests_under_sanitizers(tmp_path):
    """AddressSanitizer + UBSan on the CPU build (the GPU pool has no sanitizer
    support): accessor indexing, Memory<T> ownership, float16 bit tricks."""
    exe = tmp_path / "host_tests_san"
    subprocess.run(
        ["g++", "-std=c++14", "-O1", "-g", "-fsanitize=address,undefined",
         "-fno-sanitize-recover=all",
         os.path.join(REPO, "tests", "cpp", "host_tests.cpp"), "-o", str(exe),
         "-I", os.path.join(PKG, "include"), "-I", os.path.join(REPO, "include"),
         "-I", os.path.join(PKG, "hip"), "-L", os.path.join(PKG, "lib"), "-laccblas_hip",
         "-Wl,-rpath," + os.path.join(PKG, "lib")], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host tests ok" in out.stdout


def test_accessor_headers_compile_for_gfx950(tmp_path):
    """The same headers must compile as device code (hipcc cross-compiles)."""
    src = tmp_path / "dev.hip"
    src.write_text(
        "#include <accessor/range.hpp>\n#include <accessor/reduced_row_major.hpp>\n"
        "using acc = gko::acc::reduced_row_major<2, double, float>;\n"
        "__global__ void k(gko::acc::range<acc> r, gko::acc::range<acc::const_accessor> c)"
        "{ r(0, 0) = c(0, 0) * r(1, 1) + c(1, 0); }\nint main() { return 0; }\n")
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++14", "--offload-arch=gfx950", "-c",
                    str(src), "-o", str(tmp_path / "dev.o"), "-I",
                    os.path.join(PKG, "include")], check=True)


@pytest.mark.parametrize("name", ["gemv_benchmark", "dot_benchmark", "trsv_benchmark"])
def test_cli_flag_errors(name):
    exe = os.path.join(BIN, name)
    bad = subprocess.run([exe, "--bogus"], capture_output=True, text=True)
    assert bad.returncode == 1
    assert bad.stderr.startswith("Unsupported parameter: --bogus\nUsage: ")
    assert "[--error] [--size=SIZE]" in bad.stderr and bad.stdout == ""
    small = subprocess.run([exe, "--size=10"], capture_output=True, text=True)
    assert small.returncode == 1 and "needs to be at least" in small.stderr


def test_config_c1_dot_cpu_baseline():
    """BASELINE config 1: dot n=1M, fp64/fp64, host CPU path, no GPU."""
    out = subprocess.run([os.path.join(BIN, "dot_benchmark"), "--cpu-baseline",
                          "--size=1000000"], capture_output=True, text=True, check=True)
    header, row = out.stdout.strip().split("\n")
    names = header.split(";")
    values = row.split(";")
    assert names[0] == "Vector Size" and values[0] == "1000000"
    result = float(values[names.index("Result CPU DOT fp64")])
    assert abs(result - 154.655753144421740) <= 1e-13 * 154.655753144421740
    err = float(values[names.index("Error CPU DOT Acc<fp64, fp32>")])
    assert 1e-8 < err < 5e-7  # 6.9e-8: the storage-rounding floor for fp32


def test_cmake_configures_and_lists_the_reference_targets(tmp_path):
    """CMakeLists.txt (top level + accessor-blas_amd/) is the ROCm counterpart of
    the reference's CMake build (`/root/reference/CMakeLists.txt:1-48`,
    `cuda/CMakeLists.txt:29-31`): configure with the HIP language enabled and
    check that the three benchmark executables and the libraries are targets.
    (The full `ninja` build takes as long as `make`; it is done by hand, see
    DESIGN.md.)"""
    import shutil
    if shutil.which("cmake") is None or shutil.which("ninja") is None:
        pytest.skip("cmake / ninja not installed")
    build = tmp_path / "build"
    subprocess.run(["cmake", "-S", REPO, "-B", str(build), "-G", "Ninja"], check=True,
                   capture_output=True, timeout=300)
    targets = subprocess.run(["ninja", "-C", str(build), "-t", "targets", "all"], check=True,
                             capture_output=True, text=True, timeout=60).stdout
    for name in ("gemv_benchmark", "dot_benchmark", "trsv_benchmark", "multi_gpu_benchmark",
                 "libaccblas_hip.so", "libaccblas_vendor.so", "libaccblas_rccl.so"):
        assert name in targets, name
