"""C-ABI checks that need no GPU: the library loads, exports exactly what
include/accblas.h declares, the Python prototypes cover the same set, and
argument validation (which happens before any HIP call) returns the documented
status codes."""
import ctypes
import os
import re
import subprocess

import pytest

import accblas

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
HEADER = os.path.join(REPO, "include", "accblas.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(accblas_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = accblas.load()
    names = declared_symbols()
    assert len(names) >= 35
    for name in names:
        assert hasattr(lib, name), f"{name} declared in accblas.h but not exported"
    exported = subprocess.run(["nm", "-D", "--defined-only", accblas.LIB_PATH],
                              check=True, capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (accblas_[a-z0-9_]+)\b", exported))
    assert exported == set(names), exported.symmetric_difference(names)


def test_python_prototypes_match_header():
    assert sorted(accblas.SIGNATURES) == declared_symbols()


def test_header_is_plain_c():
    """The boundary must be consumable from C (and so from cgo/JNI/ctypes)."""
    src = os.path.join(REPO, "gpurun_out", "abi_check.c")
    os.makedirs(os.path.dirname(src), exist_ok=True)
    with open(src, "w") as f:
        f.write('#include "accblas.h"\nint main(void){return accblas_version() == 0;}\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-c", src,
                    "-I", os.path.join(REPO, "include"), "-o", src + ".o"], check=True)


def test_host_only_entry_points():
    lib = accblas.load()
    assert lib.accblas_version() == 100
    assert [lib.accblas_dtype_size(t) for t in (0, 1, 2)] == [2, 4, 8]
    assert isinstance(lib.accblas_last_error_string(), bytes)


def test_argument_validation_statuses():
    lib = accblas.load()
    dummy = ctypes.c_void_p(0x1000)
    # negative size
    assert lib.accblas_acc_gemv(2, 1, -1, 4, 1.0, dummy, 4, dummy, 1, 1.0, dummy, 1,
                                None) == -2
    assert b"negative" in lib.accblas_last_error_string()
    # row stride smaller than cols ("Wrong use of stride", matrix_helper.cuh:31-33)
    assert lib.accblas_gemv(2, 4, 8, 1.0, dummy, 4, dummy, 1, 1.0, dummy, 1, None) == -2
    # NULL operands
    assert lib.accblas_gemv(2, 4, 4, 1.0, None, 4, dummy, 1, 1.0, dummy, 1, None) == -3
    # dot without a handle
    assert lib.accblas_dot(None, 2, 10, dummy, 1, dummy, 1, dummy, None) == -3
    # trsv: helper is required, lda >= n, increments >= 1
    assert lib.accblas_trsv(2, 0, 1, 8, dummy, 8, dummy, 1, None, None) == -3
    assert lib.accblas_trsv(2, 0, 1, 8, dummy, 4, dummy, 1, dummy, None) == -2
    assert lib.accblas_trsv(2, 0, 1, 8, dummy, 8, dummy, 0, dummy, None) == -2
    assert lib.accblas_trsv(2, 7, 1, 8, dummy, 8, dummy, 1, dummy, None) == -2
    with pytest.raises(accblas.AccblasError) as info:
        accblas.gemv(accblas.F64, 4, 4, 1.0, 0x1000, 2, 0x1000, 1, 1.0, 0x1000, 1)
    assert info.value.status == -2 and "stride" in str(info.value)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(accblas, "_lib", None)
    monkeypatch.setattr(accblas, "LIB_PATH", "/nonexistent/libaccblas_hip.so")
    with pytest.raises(FileNotFoundError, match="no CPU fallback"):
        accblas.load()
